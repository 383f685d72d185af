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
from jn_cuclark_amd.dist import ShardedClassifier, shard_range, shard_groups, dense_allreduce_classify      # noqa: E402
from oracle import pyoracle                                         # noqa: E402
from helpers import small_db                                        # noqa: E402

K, HT, MAXHITS = 21, 1000003, 15


class OracleBackend:
    def __init__(self, odb, part):
        self.odb, self.part, self.row_len = odb, part, 2 * MAXHITS + 2
        self.device = torch.device("cpu")

    def query_rows_into(self, rp, con, r0, r1, out):
        rows, _ = self.odb.query_rows(K, rp[r0:r1 + 1].numpy().view(np.uint32).copy(), con.numpy().view(np.uint16), MAXHITS, part=self.part)
        out[: r1 - r0] = torch.from_numpy(rows.view(np.int16))

    def merge_result(self, srcs, n):
        acc = srcs[0][:n].numpy().view(np.uint16).copy()
        for s in srcs[1:]:
            acc = pyoracle.merge_rows(acc, s[:n].numpy().view(np.uint16).copy())
        return torch.from_numpy(pyoracle.result_rows(acc).view(np.int16)).reshape(n, 5)


def main():
    out = sys.argv[1]
    n_reads = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    genomes, sz, ky, lb = small_db(n_targets=10)
    odb = pyoracle.OracleDB.from_arrays(HT, sz, ky, lb)
    # S parts x G groups (argv[4], default: one group of `world` parts).  Every group holds the whole table and
    # classifies ITS OWN batch (seed 6 + group); the exchange stays inside the group.
    S = int(sys.argv[4]) if len(sys.argv) > 4 else world
    group, gi, part_i, G = shard_groups(S)
    if group is not None:
        codes, _ = synth.sample_reads(genomes, n_reads, 150, seed=6 + gi)
        rp, con = synth.pack_uniform(codes)
        sc = ShardedClassifier(OracleBackend(odb, shard_range(HT, part_i, S)), group=group, n_chunks=int(sys.argv[3]) if len(sys.argv) > 3 else 4)
        assert sc.world == S and sc.rank == part_i
        fin = sc.classify_gathered(torch.from_numpy(rp.view(np.int32)), torch.from_numpy(con.view(np.int16)), n_reads)
        part, ranges = sc.classify(torch.from_numpy(rp.view(np.int32)), torch.from_numpy(con.view(np.int16)), n_reads)
        assert torch.equal(part, torch.cat([fin[lo:hi] for lo, hi in ranges]))
        # the dense all-reduce of per-target vectors (BASELINE's wording of the combine) gives the same rows
        dense = dense_allreduce_classify(sc.be, torch.from_numpy(rp.view(np.int32)), torch.from_numpy(con.view(np.int16)), n_reads, 10, group=group)
        assert torch.equal(dense, fin)
        if part_i == 0:
            want, _ = odb.classify(K, rp, con, MAXHITS)
            np.savez(out if gi == 0 else out + ".g%d.npz" % gi, got=fin.numpy().view(np.uint16), want=want)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
