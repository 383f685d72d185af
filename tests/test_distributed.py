"""CPU suite, part 4: the N>1 path (DB sharded by bucket range, reduce-scatter of sparse
rows by read range, merge, top-2) with world_size 2, 3 and 8 on gloo; S parts x G groups with 4, 5 and 8 ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n_reads,chunks", [(2, 501, 4), (3, 200, 3), (2, 1, 4), (2, 7, 1), (8, 333, 3)])      # (8: the node's rank count, configs[3]/[4])
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


@pytest.mark.parametrize("world,shards", [(4, 2), (5, 2), (8, 2), (8, 4)])
def test_shard_groups_on_gloo(world, shards, tmp_path):
    """S parts x G groups: a table that needs `shards` cards, `world` ranks -> world // shards groups that each hold
    the whole table and classify their own batch; a rank past the last full group stays idle (5 ranks, 2 parts)"""
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, "301", "3", str(shards)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = []
    for g in range(world // shards):
        d = np.load(out if g == 0 else out + ".g%d.npz" % g)
        assert d["got"].shape == (301, 5) and np.array_equal(d["got"], d["want"])
        seen.append(d["got"])
    assert len(seen) == world // shards
    for a in range(len(seen)):
        for b in range(a + 1, len(seen)):
            assert not np.array_equal(seen[a], seen[b])        # the groups worked on different batches


def test_plan_shards():
    from jn_cuclark_amd.dist import plan_shards
    card = 288 * 10**9
    assert plan_shards(6_450_000_000, 8, card) == (1, 8)            # fits one card: replicas
    assert plan_shards(16_000_000_000, 8, card) == (2, 4)           # needs two: 4 groups of 2
    assert plan_shards(32_000_000_000, 8, card) == (4, 2)           # needs three: 2 groups, spread over 4 each
    assert plan_shards(60_000_000_000, 8, card) == (8, 1)
    assert plan_shards(16_000_000_000, 7, card) == (2, 3)           # one rank idle
    assert plan_shards(10**12, 8, card) == (8, 1)                   # fits no way: one group, the load says so


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
