"""GPU suite at BASELINE's full table size (HTSIZE 1610612741, k=31, ~6.4e9 k-mers, 103 GB
of bucket lines): size-independent properties instead of an oracle pass over everything.
(The oracle spot-check at this size is part of every bench.py run: --verify.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HT, K, T = 1610612741, 31, 4096


@pytest.fixture(scope="module")
def world():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from jn_cuclark_amd import synth_gpu
    dev = torch.device("cuda:0")
    genomes = synth_gpu.make_genomes(T, 100_000, seed=31, device=dev)
    raw = synth_gpu.build_db(dev, 31, K, HT, T, 3.75, genomes=genomes)
    n = 1_000_000
    rp, con, truth = synth_gpu.make_reads(genomes, n, 150, seed=77, return_truth=True)
    return dev, genomes, raw, (rp, con, truth, n)


def _classify(dev, raw, reads, shard=(0, 0), rows=False):
    import torch
    from jn_cuclark_amd import CuClarkDB
    rp, con, _, n = reads
    d_sz, d_keys, d_labels = raw
    with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
        if shard == (0, 0):
            db.read_device(d_sz, d_keys, d_labels)
        else:
            # keys of the shard = one contiguous run of the arrays
            off0 = int(d_sz[:shard[0]].to(torch.int64).sum().item())
            cnt = int(d_sz[shard[0]:shard[1]].to(torch.int64).sum().item())
            db.read_device(d_sz[shard[0]:shard[1]], d_keys[off0:off0 + cnt], d_labels[off0:off0 + cnt], shard=shard)
        st = torch.cuda.current_stream().cuda_stream
        info = db.db_info()
        if rows:
            out = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
            db.query_device(rp, con, rows_t=out, stream=st)
        else:
            out = torch.zeros((n, 5), dtype=torch.int16, device=dev)
            db.query_device(rp, con, final_t=out, stream=st)
        torch.cuda.synchronize()
    return out, info


def test_full_size_table_properties(world):
    import torch
    from jn_cuclark_amd import CuClarkDB
    dev, genomes, raw, reads = world
    rp, con, truth, n = reads
    fin_t, info = _classify(dev, raw, reads)
    assert info["line_bytes"] in (64, 128) and info["n_keys"] > 6_000_000_000
    assert info["device_bytes"] > 100e9
    fin = fin_t.cpu().numpy().view(np.uint16)
    npl = truth.numel()
    tr = truth.cpu().numpy()
    # ground truth: a read sampled from genome g is assigned to target g (id g+1)
    assigned = fin[:npl, 1]
    ok = assigned == tr + 1
    assert ok.mean() > 0.995
    assert np.all(assigned[~ok] == 0) or (assigned[~ok] != 0).mean() < 0.01     # the rest: no clean 31-mer left
    assert 0.60 < fin[:npl, 0].mean() / 120 < 0.85                              # (1-0.01)^31 = 0.73 of the k-mers hit
    assert np.all(fin[:, 0] <= 120) and np.all(fin[:, 2] <= fin[:, 0]) and np.all(fin[:, 4] <= fin[:, 2])
    # uniform random reads almost never hit (6.4e9 k-mers out of 4^31)
    assert fin[npl:, 0].astype(np.int64).sum() < 100
    # determinism / idempotence
    fin2, _ = _classify(dev, raw, reads)
    assert torch.equal(fin_t, fin2)
    # two bucket-range shards -> sparse rows -> merge -> top-2 == the fused single-shard result
    half = HT // 2
    r0, _ = _classify(dev, raw, reads, shard=(0, half), rows=True)
    r1, _ = _classify(dev, raw, reads, shard=(half, HT), rows=True)
    with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
        st = torch.cuda.current_stream().cuda_stream
        db.merge_rows_device(r0, r1, r0, n, stream=st)
        out = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.result_rows_device(r0, out, n, stream=st)
        torch.cuda.synchronize()
    assert torch.equal(out, fin_t)


def test_eight_shards_with_paired_250bp_reads_equal_unsharded(world):
    """BASELINE configs[3]/[4] shape on one card: the full-size table as 8 bucket-range shards played
    in turn (CuClarkDB.cu:552-559), every shard sees every read (:842-851), 2 x 250 bp pairs joined
    into one two-part read (file.cc:250: 501 nt, 66 containers, 440 k-mers); rows merged in shard
    order, top-2 on the merged rows == the fused single-table result."""
    import torch
    from jn_cuclark_amd import CuClarkDB, synth_gpu
    dev, genomes, raw, _ = world
    n = 200_000
    p1, c1 = synth_gpu.make_reads(genomes, n, 250, seed=521)
    _, c2 = synth_gpu.make_reads(genomes, n, 250, seed=522)
    per = c1.numel() // n                                        # [250][32 containers]
    con = torch.cat([c1.view(n, per), c2.view(n, per)], dim=1).reshape(-1).contiguous()
    rp = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (2 * per)).to(torch.int32)
    reads = (rp, con, None, n)
    fin_t, _ = _classify(dev, raw, reads)
    fin = fin_t.cpu().numpy().view(np.uint16)
    assert (fin[: n // 2, 0] > 250).mean() > 0.5                 # both mates hit
    acc = None
    for s in range(8):
        a, b = HT * s // 8, HT * (s + 1) // 8
        rows, info = _classify(dev, raw, reads, shard=(a, b), rows=True)
        assert info["shard_begin"] == a and info["shard_end"] == b
        if acc is None:
            acc = rows
        else:
            with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
                db.merge_rows_device(acc, rows, acc, n, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
    with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
        out = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.result_rows_device(acc, out, n, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert torch.equal(out, fin_t)


def test_eight_minimizer_parts_with_paired_250bp_reads_equal_unsharded(world):
    """The same with the DEFAULT partition of a sharded table: 8 parts by minimizer (mc_index_begin(part, 8):
    every part streams the whole table and keeps the k-mers whose minimizer hashes to it), played in turn on
    the one card; every part sees every 2 x 250 bp pair; ONE k-way merge + top-2 over the 8 row sets (what the
    owner of a read range runs after the exchange) == the fused single-table result."""
    import torch
    from jn_cuclark_amd import CuClarkDB, synth_gpu
    dev, genomes, raw, _ = world
    n = 200_000
    p1, c1 = synth_gpu.make_reads(genomes, n, 250, seed=521)
    _, c2 = synth_gpu.make_reads(genomes, n, 250, seed=522)
    per = c1.numel() // n
    con = torch.cat([c1.view(n, per), c2.view(n, per)], dim=1).reshape(-1).contiguous()
    rp = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (2 * per)).to(torch.int32)
    fin_t, _ = _classify(dev, raw, (rp, con, None, n))
    d_sz, d_keys, d_labels = raw
    n_keys = int(d_keys.numel())
    st = torch.cuda.current_stream().cuda_stream
    parts, owned, lines = [], 0, None
    for p in range(8):
        with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
            db.read_chunks(lambda: [(d_sz, d_keys, d_labels, 0, HT)], n_keys, part=p, n_parts=8, device=True)
            info = db.db_info()
            assert info["part"] == p and info["n_parts"] == 8 and info["n_keys"] == n_keys
            assert lines is None or lines == info["line_end"] - info["line_begin"]       # every part: the same line count
            lines = info["line_end"] - info["line_begin"]
            owned += info["n_keys_owned"]
            assert 0.11 < info["n_keys_owned"] / n_keys < 0.14                            # an eighth each
            rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
            db.query_device(rp, con, rows_t=rows, stream=st)
            torch.cuda.synchronize()
            parts.append(rows)
    assert owned == n_keys
    with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=15) as db:
        out = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.merge_result_device(parts, n, final_t=out, stream=st)
        torch.cuda.synchronize()
    assert torch.equal(out, fin_t)
    assert all(int((p_[:, 0] != 0).sum()) > n // 4 for p_ in parts)                     # every part contributes
