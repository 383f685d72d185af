"""Worker for tests/test_distributed.py: runs the sharded classification protocol of
jn_cuclark_amd.dist on `gloo` (CPU).  The per-rank compute is played by the ORACLE
(test double for the HIP backend): what is under test is the host logic -- shard
ranges, the all_to_all reduce-scatter by read range, merge order, top-2, gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from jn_cuclark_amd import synth                                    # noqa: E402
from jn_cuclark_amd.dist import ShardedClassifier, shard_range      # noqa: E402
from oracle import pyoracle                                         # noqa: E402
from helpers import small_db                                        # noqa: E402

K, HT, MAXHITS = 21, 1000003, 15


class OracleBackend:
    def __init__(self, odb, part):
        self.odb, self.part, self.row_len = odb, part, 2 * MAXHITS + 2

    def query_rows(self, rp, con, n):
        rows, _ = self.odb.query_rows(K, rp.numpy().view(np.uint32), con.numpy().view(np.uint16), MAXHITS, part=self.part)
        return torch.from_numpy(rows.view(np.int16))

    def merge_rows(self, a, b, n):
        m = pyoracle.merge_rows(a[:n].numpy().view(np.uint16), b[:n].numpy().view(np.uint16))
        a[:n] = torch.from_numpy(m.view(np.int16))

    def result_rows(self, rows, n):
        return torch.from_numpy(pyoracle.result_rows(rows[:n].numpy().view(np.uint16)).view(np.int16))


def main():
    out = sys.argv[1]
    n_reads = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    genomes, sz, ky, lb = small_db(n_targets=10)
    codes, _ = synth.sample_reads(genomes, n_reads, 150, seed=6)
    rp, con = synth.pack_uniform(codes)
    odb = pyoracle.OracleDB.from_arrays(HT, sz, ky, lb)
    sc = ShardedClassifier(OracleBackend(odb, shard_range(HT, rank, world)))
    fin = sc.classify_gathered(torch.from_numpy(rp.view(np.int32)), torch.from_numpy(con.view(np.int16)), n_reads)
    part, (lo, hi) = sc.classify(torch.from_numpy(rp.view(np.int32)), torch.from_numpy(con.view(np.int16)), n_reads)
    assert torch.equal(part, fin[lo:hi])
    if rank == 0:
        want, _ = odb.classify(K, rp, con, MAXHITS)
        np.savez(out, got=fin.numpy().view(np.uint16), want=want)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
