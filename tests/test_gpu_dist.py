"""GPU suite: the N > 1 path with the real backend class -- ShardedClassifier(HipBackend) on every rank,
line-range parts read from the database files, chunked exchange on a second stream, k-way merge + top-2 --
against the oracle on the whole table.  world_size 2 and 3 on the ONE card of the test box (gloo between the
ranks; on a multi-GPU node the same code runs over RCCL, bench.py --mode shard)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta, pack_with_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,chunks,shards,backend", [
    (2, 4, 0, "gloo"), (3, 1, 0, "gloo"),
    (4, 2, 2, "gloo"),       # 2 parts x 2 groups: every group holds the whole table, the exchange stays inside it
    (1, 2, 0, "nccl"),       # first contact with RCCL on the box there is: one rank, the collectives for real
])
def test_hip_backend_ranks_reproduce_the_unsharded_result(oracle, tmp_path, world, chunks, shards, backend):
    k, ht = 21, 1000003
    genomes, sz, ky, lb = small_db(n_targets=9, glen=7000)
    nzb = np.flatnonzero(sz)
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(ht)
    base = str(tmp_path / "db")
    oracle.db_write(base, ht, 4, canon, lb)
    names, seqs = mixed_fasta(genomes, k, n=5003)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), k)
    want, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).classify(k, rp, con, 15)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npy")
    np.savez(inp, k=k, htsize=ht, base=base, rp=rp, con=con, targets=9, chunks=chunks, shards=shards)
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", DIST_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), inp, out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    assert np.array_equal(got, want)
    if shards:
        assert np.array_equal(np.load(out + ".last.npy"), want)
    assert (want[:, 2] > 0).sum() > 2000
