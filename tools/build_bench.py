#!/usr/bin/env python3
"""Database build timing on the GPU box: CPU builder (sort of all occurrences) vs
--gpu-build (counting sort by bucket on the device).  Synthetic genomes."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jn_cuclark_amd import synth  # noqa: E402


def main():
    n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    glen = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    threads = sys.argv[3] if len(sys.argv) > 3 else "16"
    work = "/tmp/bb"
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(1)
    tl = []
    for i in range(n_gen):
        codes = rng.integers(0, 4, glen, dtype=np.uint8)
        p = "%s/g%d.fa" % (work, i)
        open(p, "wb").write(synth.fasta_text([b"g%d" % i], [synth.codes_to_ascii(codes)], width=80))
        tl.append("%s\tT%d\n" % (p, i))
    open(work + "/targets.txt", "w").write("".join(tl))
    reads = work + "/r.fa"
    open(reads, "wb").write(b">r\n" + b"ACGT" * 40 + b"\n")
    for tag, extra in (("gpu", ["--gpu-build"]), ("cpu", [])):
        d = "%s/db_%s" % (work, tag)
        subprocess.run(["rm", "-rf", d]); os.makedirs(d)
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "bin", "cuCLARK"), "-k", "31", "-T", work + "/targets.txt", "-D", d, "-O", reads,
                            "-R", work + "/o" + tag, "-n", threads] + extra, capture_output=True, text=True)
        dt = time.time() - t0
        stored = [l for l in r.stderr.split("\n") if "successfully stored" in l]
        print("%s build+load+classify: %.1fs rc=%d  %s" % (tag, dt, r.returncode, stored[-1] if stored else r.stderr[-300:]), flush=True)
    same = all(open("%s/db_cpu/%s" % (work, f), "rb").read() == open("%s/db_gpu/%s" % (work, f), "rb").read()
               for f in os.listdir(work + "/db_cpu") if f.endswith((".ky", ".lb")))
    print("identical .ky/.lb:", same, " total bases: %.0fM" % (n_gen * glen / 1e6))


if __name__ == "__main__":
    main()
