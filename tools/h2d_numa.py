"""H2D / D2H rate of pinned buffers allocated while the process is bound to the CPUs of each NUMA node (first touch
decides where the pages live): is the 30 GB/s of the pipelined path a placement problem?  Run on the GPU box."""
import glob
import os
import time

import torch


def cpus_of(node):
    txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in txt.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("numa nodes:", nodes, "| gpu numa:", [open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")][:8], flush=True)
allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus:", len(allowed), allowed[:4], "...", allowed[-4:], flush=True)
dev = torch.device("cuda:0")
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device=dev)
for node in nodes:
    cp = [c for c in cpus_of(node) if c in allowed]
    if not cp:
        print("node", node, "no allowed cpus"); continue
    os.sched_setaffinity(0, cp)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    h.fill_(1)
    for direction in ("h2d", "d2h"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            if direction == "h2d":
                d.copy_(h, non_blocking=True)
            else:
                h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("node %d (%d cpus) %s: %.1f GB/s" % (node, len(cp), direction, 5 * n / dt / 1e9), flush=True)
    del h
os.sched_setaffinity(0, allowed)

# the library's own pinned batch buffers (mc_alloc_batches binds them to the GPU's node)
import ctypes as C
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jn_cuclark_amd import _lib
lib = _lib.load_library()
hip = C.CDLL("libamdhip64.so")
for bind in (nodes + [None]):
    os.sched_setaffinity(0, [c for c in cpus_of(bind) if c in allowed] if bind is not None else allowed)
    h = C.c_void_p()
    _lib.check(lib.mc_open(C.byref(h), 0, 31, 1610612741, 16, 15))
    _lib.check(lib.mc_alloc_batches(h, 1, 25_000_000, 500_000_000, 0))
    p, c_, f, r = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.mc_batch_buffers(h, 0, C.byref(p), C.byref(c_), C.byref(f), C.byref(r)))
    C.memset(c_.value, 1, 1_000_000_000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        hip.hipMemcpy(C.c_void_p(d.data_ptr()), c_, C.c_size_t(1_000_000_000), C.c_int(1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("library buffers, thread on node %s: h2d %.1f GB/s" % (bind, 5 * 1e9 / dt / 1e9), flush=True)
    lib.mc_close(h)
