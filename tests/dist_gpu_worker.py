"""Worker for tests/test_gpu_dist.py: the sharded protocol of jn_cuclark_amd.dist with the REAL backend
(HipBackend over libmcclark.so) on every rank; ranks share the one card of the test box and talk over gloo
(no RCCL peers there).  Rank r holds line-range part r of the table in the files."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from jn_cuclark_amd import CuClarkDB                                  # noqa: E402
from jn_cuclark_amd.dist import ShardedClassifier, HipBackend, dense_allreduce_classify, shard_groups        # noqa: E402


def main():
    inp, out = sys.argv[1], sys.argv[2]
    d = np.load(inp, allow_pickle=False)
    k, ht, base = int(d["k"]), int(d["htsize"]), str(d["base"])
    backend = os.environ.get("DIST_TEST_BACKEND", "gloo")
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        # RCCL for real (world_size 1 on the one card of the test box: init, all_to_all_single, all_reduce and
        # all_gather on device buffers, the side-stream ordering of ShardedClassifier.classify)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")                  # before anything touches the GPU
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(dev)
    S = int(d["shards"]) if "shards" in d.files and int(d["shards"]) else world
    group, gi, part, G = shard_groups(S)
    rp = torch.from_numpy(d["rp"].view(np.int32)).to(dev)
    con = torch.from_numpy(d["con"].view(np.int16)).to(dev)
    n = rp.numel() - 1
    full = None
    if group is not None:
      with CuClarkDB(k=k, numBatches=1, numTargets=int(d["targets"]), device=0, htsize=ht, maxhits=15) as db:
        assert db.read_part(base, part, S) is True
        info = db.db_info()
        assert info["part"] == part and info["n_parts"] == S and info["index_kind"] == 1
        sc = ShardedClassifier(HipBackend(db, dev), group=group, n_chunks=int(d["chunks"]))
        assert sc._gloo == (backend == "gloo")
        full = sc.classify_gathered(rp, con, n)
        again, ranges = sc.classify(rp, con, n)          # buffers are reused: same answer
        torch.cuda.synchronize()
        assert torch.equal(again, torch.cat([full[lo:hi] for lo, hi in ranges]))
        dense = dense_allreduce_classify(sc.be, rp, con, n, int(d["targets"]), group=group)
        assert torch.equal(dense, full)
        owned = torch.tensor([info["n_keys_owned"]], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(owned, group=group)
        assert int(owned.item()) == info["n_keys"]
    # every group ends with the whole result (each holds the whole table): the last group's leader reports too
    if full is not None and part == 0 and gi == G - 1 and gi != 0:
        np.save(out + ".last.npy", full.cpu().numpy().view(np.uint16))
    if rank == 0:
        np.save(out, full.cpu().numpy().view(np.uint16))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
