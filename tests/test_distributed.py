"""CPU suite, part 4: the N>1 path (DB sharded by bucket range, reduce-scatter of sparse
rows by read range, merge, top-2) with world_size 2 and 3 on gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n_reads,chunks", [(2, 501, 4), (3, 200, 3), (2, 1, 4), (2, 7, 1)])
def test_sharded_protocol_on_gloo(world, n_reads, chunks, tmp_path):
    out = str(tmp_path / "res.npz")
    port = 29600 + world * 7 + n_reads % 50 + chunks
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n_reads), str(chunks)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    assert d["got"].shape == (n_reads, 5)
    assert np.array_equal(d["got"], d["want"])


def test_ranges_cover_everything():
    from jn_cuclark_amd.dist import shard_range, read_range
    for world in (1, 2, 3, 8):
        ht = 1610612741
        edges = [shard_range(ht, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == ht
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        for n in (0, 1, 7, 10_000_000):
            rr = [read_range(n, r, world) for r in range(world)]
            assert sum(hi - lo for lo, hi, _ in rr) == n
