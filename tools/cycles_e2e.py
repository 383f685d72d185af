"""Database cycles at the metric's size (run on the GPU box): the bench's 6.45e9-k-mer table as .sz/.ky/.lb files, a FASTQ file of N
reads, `bin/cuCLARK -k 31` once against the resident table and once with the table cut into C parts that are loaded one at a time
(MC_GROUP_CYCLES=C: the path a table larger than all devices takes -- the reference's swapDbParts loop, src/CuCLARK_hh.hh:1765-1772);
the two CSV files must be byte-identical.
    python tools/cycles_e2e.py [--reads 10000000] [--cycles 2] [--dir D]"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jn_cuclark_amd import synth_gpu

K, HT, T, LAM, GLEN = 31, 1610612741, 4096, 3.75, 100_000


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--cycles", type=int, default=2)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    work = a.dir or ("/dev/shm/mc_cycles" if os.path.isdir("/dev/shm") else "/tmp/mc_cycles")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    try:
        dev = torch.device("cuda", 0)
        genomes = synth_gpu.make_genomes(T, GLEN, seed=31, device=dev)
        base = os.path.join(work, "db_central_k%d_t%d_s%d_m0.tsk" % (K, T, HT))
        ranges = [(HT * j // 16, HT * (j + 1) // 16) for j in range(16)]

        def chunks():
            for b0, b1 in ranges:
                d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, K, HT, T, LAM, genomes=genomes, shard=(b0, b1))
                yield d_sz, d_keys, d_labels, b0, b1

        t0 = time.time()
        n_keys, nbytes = synth_gpu.write_db_files(base, chunks())
        print("database files: %.2fe9 k-mers, %.1f GB, written in %.1f s" % (n_keys / 1e9, nbytes / 1e9, time.time() - t0), flush=True)
        fq = os.path.join(work, "reads.fq")
        truth = synth_gpu.write_fastq(fq, genomes, a.reads, seed=91).numpy()
        del genomes
        torch.cuda.empty_cache()
        exe = os.path.join(ROOT, "bin", "cuCLARK")
        digests = {}
        for tag, env in (("resident", {}), ("%d cycles" % a.cycles, {"MC_GROUP_CYCLES": str(a.cycles)})):
            for k_, v in env.items():
                os.environ[k_] = v
            r = synth_gpu.host_driver_run(exe, work, K, T, fq, a.reads, truth=truth, timeout=1000)
            for k_ in env:
                os.environ.pop(k_)
            assert r["csv_lines"] == a.reads and r["assigned_to_their_genome"] > 0.995 * r["checked"], r
            digests[tag] = sha(os.path.join(work, "res.csv"))
            print("%-10s wall %.1f s | %s | %s | sha256(csv) %s" % (tag, r["wall_s"], r["line"], " | ".join(r["timing"]), digests[tag][:16]), flush=True)
        assert len(set(digests.values())) == 1, digests
        print("the CSV of the cycled run is byte-identical to the resident run's (%d reads)" % a.reads, flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
