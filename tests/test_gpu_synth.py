"""GPU suite: the on-device workload generator used by bench.py (libmcsynth + torch)
produces a valid database, and the HIP path is bit-exact on it against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_generated_db_roundtrip_and_parity(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from jn_cuclark_amd import CuClarkDB, synth_gpu, synth
    dev = torch.device("cuda:0")
    k, ht, T, lam = 21, 1000003, 64, 6.0
    genomes = synth_gpu.make_genomes(T, 3000, seed=5, device=dev)
    d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 7, k, ht, T, lam, genomes=genomes)
    sz = d_sz.cpu().numpy()
    ky = d_keys.cpu().numpy().view(np.uint32)
    lb = d_labels.cpu().numpy().view(np.uint16)
    assert sz.sum() == ky.size == lb.size
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    # ascending inside every bucket (what hashTable's sortall guarantees, hashTable_hh.hh:203-216)
    for b in range(0, ht, 101):
        seg = ky[off[b]:off[b + 1]].astype(np.int64)
        assert np.all(np.diff(seg) >= 0)
    # the background part equals the host twin, bucket by bucket
    g_np = genomes.cpu().numpy()
    app = synth.canonical(np.concatenate([synth.kmers_of(g, k) for g in g_np]), k)
    app_r = set((app % np.uint64(ht)).tolist())
    checked = 0
    for b in range(0, ht, 997):
        if b in app_r:
            continue
        kk, ll = synth_gpu.bucket_host(7, k, ht, T, lam, b)
        assert np.array_equal(ky[off[b]:off[b + 1]], kk) and np.array_equal(lb[off[b]:off[b + 1]], ll)
        checked += 1
    assert checked > 500
    # every genome k-mer is in the table
    odb = oracle.OracleDB.from_arrays(ht, sz, ky, lb)
    for x in app[::211].tolist():
        assert odb.lookup(k, x)[0]
    # reads: device generator -> HIP path == oracle
    rp_t, con_t = synth_gpu.make_reads(genomes, 20000, 150, seed=9)
    rp = rp_t.cpu().numpy().view(np.uint32)
    con = con_t.cpu().numpy().view(np.uint16)
    want, _ = odb.classify(k, rp, con, 15)
    with CuClarkDB(k=k, numBatches=1, numTargets=T, device=0, htsize=ht, maxhits=15) as db:
        db.read_device(d_sz, d_keys, d_labels)
        fin = torch.zeros((20000, 5), dtype=torch.int16, device=dev)
        db.query_device(rp_t, con_t, final_t=fin, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    got = fin.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, want)
    hit = want[:10000, 0].mean() / (150 - k + 1)
    assert 0.6 < hit < 0.95          # planted half: (1 - 0.01)^k of the k-mers survive
    assert want[10000:, 0].mean() < 2
