#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary's streaming interface (mc_alloc_batches /
mc_submit / mc_wait): packed reads start in pinned HOST memory, results end there.
Same table and reads as bench.py; 4 batches of 2.5 M reads in flight (bench.py reports the same figure as `pipelined`)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jn_cuclark_amd import CuClarkDB, synth_gpu  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ht, k, T = 1610612741, 31, 4096
    genomes = synth_gpu.make_genomes(T, 100_000, seed=31, device=dev)
    raw = synth_gpu.build_db(dev, 31, k, ht, T, 3.75, genomes=genomes)
    nb, per = 4, 2_500_000
    db = CuClarkDB(k=k, numBatches=nb, numTargets=T, device=0, htsize=ht, maxhits=15)
    db.read_device(*raw)
    del raw
    rp_l, con_l, fin_l, _ = db.malloc(per, per * 20)
    for b in range(nb):
        rp, con = synth_gpu.make_reads(genomes, per, 150, seed=100 + b)
        rp_l[b][: per + 1] = rp.cpu().numpy().view(np.uint32)
        con_l[b][: per * 20] = con.cpu().numpy().view(np.uint16)
        db.readyBatch(b, per, per * 20)
    torch.cuda.synchronize()
    for rounds in (1, 5):
        t0 = time.perf_counter()
        for i in range(rounds * nb):          # steady state: a batch is resubmitted once its previous results were waited for
            b = i % nb
            if i >= nb:
                db.waitForBatch(b)
            db.queryBatch(b)
        for b in range(nb):
            db.waitForBatch(b)
        dt = time.perf_counter() - t0
        print("rounds=%d: %.1f Mreads/s host-to-host (%.1f GB/s H2D + %.1f GB/s D2H)"
              % (rounds, rounds * nb * per / dt / 1e6, rounds * nb * per * 44 / dt / 1e9, rounds * nb * per * 10 / dt / 1e9), flush=True)
    assert (fin_l[0][: per * 5].reshape(per, 5)[: per // 2, 1] > 0).mean() > 0.99
    db.close()


if __name__ == "__main__":
    main()
