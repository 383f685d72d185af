"""Which copy of the HIP runtime a Python process ends up with (run on the GPU box).  jn_cuclark_amd._lib imports torch before it
loads libmcclark.so, because the other order gives the process two runtimes -- /opt/rocm's for the library, torch's own for torch --
and the one that starts second finds no device.  MC_PROBE_RAW=1 loads the library the raw way first (ctypes, no torch) to show it."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("MC_PROBE_RAW"):
    C.CDLL(os.path.join(ROOT, "jn_cuclark_amd", "libmcclark.so"))
    print("libmcclark.so loaded by hand, before torch", flush=True)
from jn_cuclark_amd import _lib  # noqa: E402
print("mc_index_plan through the package:", _lib.index_plan(6_450_000_000, 1, 288 * 10**9)["fill"], flush=True)
import torch  # noqa: E402
print("torch.cuda.is_available():", torch.cuda.is_available(), flush=True)
from jn_cuclark_amd import CuClarkDB  # noqa: E402
try:
    with CuClarkDB(k=21, numBatches=1, numTargets=6, device=0, htsize=1000003, maxhits=15):
        print("mc_open after that: ok", flush=True)
except Exception as e:
    print("mc_open after that: FAILED:", e, flush=True)
maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
print("copies of libamdhip64 in this process:", sorted(set(maps)), flush=True)
