"""GPU suite of the super-k-mer index (MC_INDEX=skm, csrc/mc_skm.hpp): records of up to 9 k-mers stored relative to
their minimizer, a read's run matched against a record once.  Everything is compared with the oracle on the same
table and reads -- sparse rows and final rows, bit for bit -- as the other two indexes are (tests/test_gpu_parity.py):
the answer is the reference's (src/CuClarkDB.cu:999-1254), the layout is ours."""
import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta, pack_with_oracle

pytestmark = pytest.mark.gpu

HT = 1000003


@pytest.fixture(autouse=True)
def skm(monkeypatch):
    monkeypatch.setenv("MC_INDEX", "skm")


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    from jn_cuclark_amd import CuClarkDB
    return CuClarkDB


def _check(gpu, oracle, k, sz, ky, lb, rp, con, n_targets, ht=HT, maxhits=15, expect_kind=2):
    want_rows, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).query_rows(k, rp, con, maxhits)
    want = oracle.result_rows(want_rows)
    with gpu(k=k, numBatches=1, numTargets=n_targets, device=0, htsize=ht, maxhits=maxhits) as db:
        db.read_arrays(sz, ky, lb)
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    assert info["index_kind"] == expect_kind and info["n_keys"] == ky.size
    bad = np.flatnonzero((rows != want_rows).any(axis=1))
    assert bad.size == 0, "first differing reads %r: got %r want %r" % (bad[:5], rows[bad[:3]], want_rows[bad[:3]])
    assert np.array_equal(got, want)
    return info, want


@pytest.mark.parametrize("k", [25, 26, 27, 28, 29, 30, 31])
def test_mixed_reads_bit_exact_for_every_k_the_index_serves(gpu, oracle, k):
    """ragged input (N-split reads, short reads and parts, lower case, U, all-N) on a table of related genomes"""
    genomes, sz, ky, lb = small_db(k=k, glen=6000, shared=400)
    names, seqs = mixed_fasta(genomes, k, n=3000)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), k)
    info, want = _check(gpu, oracle, k, sz, ky, lb, rp, con, 6)
    assert (want[:, 2] > 0).sum() > 1500
    assert info["line_bytes"] == 128 and info["n_spilled_keys"] < ky.size / 2        # records hold several k-mers each


def test_k_outside_the_range_falls_back_to_the_minimizer_index(gpu, oracle):
    genomes, sz, ky, lb = small_db(k=21)
    codes, _ = synth.sample_reads(genomes, 500, 150, seed=4)
    rp, con = synth.pack_uniform(codes)
    _check(gpu, oracle, 21, sz, ky, lb, rp, con, 6, expect_kind=1)


@pytest.mark.parametrize("length", [31, 32, 38, 39, 40, 150, 151, 157, 158, 159, 160, 166, 167, 285, 286, 287, 288, 501, 1000, 5000])
def test_read_lengths_around_step_boundaries(gpu, oracle, length):
    """k-mers per read around 128 (one step of the kernel), runs cut at step boundaries, the m-mers behind the last k-mer"""
    k = 31
    genomes, sz, ky, lb = small_db(k=k, glen=8000)
    codes, _ = synth.sample_reads(genomes, 300, length, seed=length)
    rp, con = synth.pack_uniform(codes)
    _check(gpu, oracle, k, sz, ky, lb, rp, con, 6)


def test_low_complexity_sequence(gpu, oracle):
    """poly-A, short tandem repeats, (AT)n: every window of a k-mer reaches the smallest hash, m-mers that are their own
    reverse complement (k = 28: m = 20)"""
    for k in (28, 31):
        rng = np.random.default_rng(k)
        genomes = []
        for t in range(5):
            g = rng.integers(0, 4, size=4000).astype(np.uint8)
            g[300:420] = 3                                              # poly-A
            g[800:1000] = np.tile(np.array([0, 1, 2], dtype=np.uint8) + (t & 1), 67)[:200]
            g[1500:1600] = np.tile(np.array([3, 0], dtype=np.uint8), 50)   # (AT)n
            g[2000:2000 + 60] = np.tile(np.array([1, 2], dtype=np.uint8), 30)  # (CG)n
            genomes.append(g)
        genomes = np.stack(genomes)
        sz, ky, lb = synth.genome_db(genomes, k, HT)
        codes, _ = synth.sample_reads(genomes, 3000, 150, seed=k)
        # plus reads cut exactly out of the repeats, both strands
        extra = np.stack([genomes[t][s:s + 150] for t in range(5) for s in (250, 300, 330, 780, 850, 1450, 1500, 1980)])
        extra = np.concatenate([extra, 3 - extra[:, ::-1]])
        rp, con = synth.pack_uniform(np.concatenate([codes, extra]))
        _check(gpu, oracle, k, sz, ky, lb, rp, con, 5)


def _sk_key_bits(cw, m):
    B = 2 * m - 32
    lo, hi = cw & 0xFFFFFFFF, cw >> 32
    a = (lo * 0x9E3779B1) & 0xFFFFFFFF
    h = hi ^ (a >> (32 - B))
    b = (((a + (h & 0xFFFFFF) * 0x85EBCB) & 0xFFFFFFFF) * 0xC2B2AE35) & 0xFFFFFFFF
    return (b << 20) | (h << (20 - B))


def test_one_minimizer_shared_by_60000_kmers_goes_to_hashed_chains(gpu, oracle):
    """A conserved m-mer in tens of thousands of contexts: the line's entries go, one by one, to the chain line their own hash
    picks; a run that lands there is answered k-mer by k-mer.  Absent k-mers with the same minimizer miss."""
    k, ht, m = 31, 1000003, 23
    rng = np.random.default_rng(7)
    cand = rng.integers(0, 1 << (2 * m), size=20000, dtype=np.uint64)
    canon = np.minimum(cand, synth.revcomp(cand, m))
    keys = np.array([_sk_key_bits(int(c), m) for c in canon], dtype=np.uint64)
    X = int(cand[int(np.argmin(keys))])
    free = rng.choice(9 << 16, size=80000, replace=False).astype(np.uint64)
    off, bits = free >> np.uint64(16), free & np.uint64(0xFFFF)
    rbits = np.uint64(2) * (np.uint64(8) - off)
    left, right = bits >> rbits, bits & ((np.uint64(1) << rbits) - np.uint64(1))
    kmers = (left << (rbits + np.uint64(2 * m))) | (np.uint64(X) << rbits) | right
    ck = np.unique(synth.canonical(kmers, k))
    stored, absent = ck[:60000], ck[60000:]
    labels = (np.arange(stored.size) % 7).astype(np.uint16)
    genomes, sz0, ky0, lb0 = small_db(k=k, n_targets=7, glen=3000)
    nzb = np.flatnonzero(sz0)
    base_canon = np.repeat(nzb, sz0[nzb]).astype(np.uint64) + ky0.astype(np.uint64) * np.uint64(ht)
    keep = ~np.isin(base_canon, ck)
    sz, ky, lb = synth.db_from_kmers(np.concatenate([stored, base_canon[keep]]), np.concatenate([labels, lb0[keep]]), ht)
    q = np.concatenate([stored[::3], absent, synth.revcomp(stored[1::50], k)])
    codes = np.zeros((q.size, k), dtype=np.uint8)
    for j in range(k):
        codes[:, j] = ((q >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.uint8)
    rp1, con1 = synth.pack_uniform(codes)
    c2, _ = synth.sample_reads(genomes, 500, 150, seed=3)
    rp2, con2 = synth.pack_uniform(c2)
    rp = np.concatenate([rp1[:-1], rp2 + rp1[-1]]).astype(np.uint32)
    info, want = _check(gpu, oracle, k, sz, ky, lb, rp, np.concatenate([con1, con2]), 7, ht=ht)
    assert info["largest_line"] >= 55000 and info["n_lines_crowded"] >= 1
    assert int((want[:q.size, 2] > 0).sum()) == stored[::3].size + stored[1::50].size


@pytest.mark.parametrize("n_parts", [2, 3])
def test_parts_by_minimizer_add_up_to_the_whole_table(gpu, oracle, n_parts):
    import torch
    k = 31
    genomes, sz, ky, lb = small_db(k=k, n_targets=8, glen=6000)
    names, seqs = mixed_fasta(genomes, k, n=3000)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), k)
    want_rows, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(k, rp, con, 15)
    n = rp.size - 1
    dev = torch.device("cuda:0")
    rp_t, con_t = torch.from_numpy(rp.view(np.int32)).to(dev), torch.from_numpy(con.view(np.int16)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    parts, dbs, owned = [], [], 0
    for p in range(n_parts):
        db = gpu(k=k, numBatches=1, numTargets=8, device=0, htsize=HT, maxhits=15)
        db.read_chunks(lambda: [(sz, ky, lb, 0, HT)], ky.size, part=p, n_parts=n_parts)
        info = db.db_info()
        assert info["index_kind"] == 2 and info["part"] == p and info["n_parts"] == n_parts
        owned += info["n_keys_owned"]
        rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
        db.query_device(rp_t, con_t, rows_t=rows, stream=st)
        parts.append(rows)
        dbs.append(db)
    torch.cuda.synchronize()
    assert owned >= ky.size
    out_rows = torch.zeros_like(parts[0])
    fin = torch.zeros((n, 5), dtype=torch.int16, device=dev)
    dbs[0].merge_result_device(parts, n, rows_t=out_rows, final_t=fin, stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(out_rows.cpu().numpy().view(np.uint16), want_rows)
    assert np.array_equal(fin.cpu().numpy().view(np.uint16), oracle.result_rows(want_rows))
    assert all(int((p_[:, 0] != 0).sum()) > 100 for p_ in parts)
    for db in dbs:
        db.close()


def test_genome_shaped_table_every_read_against_the_oracle(gpu, oracle):
    """512 structured genomes (genera of 4 with shared sequence, a conserved block in every genome, tandem repeats, poly-A:
    jn_cuclark_amd.synth_gpu.make_structured_genomes) -- ~1e8 k-mers, built from chunks on the device, 400 000 reads, every
    row against the oracle; the lines hold several k-mers per record and few overflow"""
    import torch
    from jn_cuclark_amd import synth_gpu
    k, ht, T = 31, 1610612741, 512
    dev = torch.device("cuda:0")
    genomes = synth_gpu.make_structured_genomes(T, 200_000, seed=5, device=dev)
    chunks, n_keys = synth_gpu.build_genome_db(genomes, k, ht)
    n = 400_000
    rp, con = synth_gpu.make_reads(genomes, n, 150, seed=6)
    host = [(c[0].cpu().numpy(), c[1].cpu().numpy(), c[2].cpu().numpy(), c[3], c[4]) for c in chunks]
    raw = tuple(np.concatenate([c[i] for c in host]) for i in range(3))
    odb = oracle.OracleDB.from_arrays(ht, raw[0], raw[1].view(np.uint32), raw[2].view(np.uint16))
    oracle.set_num_threads(oracle.host_cores())
    want_rows, _ = odb.query_rows(k, rp.cpu().numpy().view(np.uint32), con.cpu().numpy().view(np.uint16), 15)
    odb.close()
    with gpu(k=k, numBatches=1, numTargets=T, device=0, htsize=ht, maxhits=15) as db:
        db.read_chunks(lambda: chunks, n_keys, device=True)
        info = db.db_info()
        rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
        fin = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.query_device(rp, con, final_t=fin, rows_t=rows, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert info["index_kind"] == 2 and info["n_keys"] == n_keys
    got = rows.cpu().numpy().view(np.uint16)
    bad = np.flatnonzero((got != want_rows).any(axis=1))
    assert bad.size == 0, "reads %r differ: got %r want %r" % (bad[:5], got[bad[:3]], want_rows[bad[:3]])
    assert np.array_equal(fin.cpu().numpy().view(np.uint16), oracle.result_rows(want_rows))
    # several k-mers per record, a fraction of the bytes the minimizer index takes (37-41 per k-mer), few overflowing lines
    assert info["n_spilled_keys"] * 3 < n_keys
    assert info["device_bytes"] < 25 * n_keys
    assert info["n_lines_overflowing"] < 0.03 * (info["line_end"] - info["line_begin"])


def test_auto_picks_records_for_genomes_and_lines_for_isolated_kmers(gpu, oracle, monkeypatch):
    """MC_INDEX=auto counts for both indexes in the first build pass and looks at how the k-mers clump around their
    minimizers: a table made of genomes gets super-k-mer records, a table of isolated random k-mers minimizer lines"""
    monkeypatch.setenv("MC_INDEX", "auto")
    k = 31
    genomes, sz, ky, lb = small_db(k=k, glen=6000)
    codes, _ = synth.sample_reads(genomes, 2000, 150, seed=12)
    rp, con = synth.pack_uniform(codes)
    _check(gpu, oracle, k, sz, ky, lb, rp, con, 6, expect_kind=2)
    sz, ky, lb = synth.random_db(seed=3, htsize=HT, n_keys=200000, n_targets=40, k=k)
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    stored = np.concatenate([ky[off[r]:off[r + 1]].astype(np.uint64) * np.uint64(HT) + np.uint64(r) for r in range(0, HT, 997)])
    kcodes = np.zeros((stored.size, k), dtype=np.uint8)
    for j in range(k):
        kcodes[:, j] = ((stored >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.uint8)
    rp, con = synth.pack_uniform(kcodes)
    info, want = _check(gpu, oracle, k, sz, ky, lb, rp, con, 40, expect_kind=1)
    assert (want[:, 2] > 0).mean() > 0.3        # (a random quotient is the canonical form of its k-mer about half the time)


def test_file_loader_builds_the_records_from_sz_ky_lb(gpu, oracle, tmp_path):
    """mc_load_db: the files streamed in chunks, twice, through the same build"""
    k = 27
    genomes, sz, ky, lb = small_db(k=k, n_targets=8, glen=6000)
    nzb = np.flatnonzero(sz)
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(HT)
    base = str(tmp_path / "db")
    oracle.db_write(base, HT, 8, canon, lb)          # (k = 27 over 1000003 buckets: the quotients need more than 32 bits)
    names, seqs = mixed_fasta(genomes, k, n=2000)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), k)
    want_rows, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(k, rp, con, 15)
    with gpu(k=k, numBatches=1, numTargets=8, device=0, htsize=HT, maxhits=15) as db:
        assert db.read(base, key_bytes=8) is True
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    assert info["index_kind"] == 2 and info["n_keys"] == ky.size
    assert np.array_equal(rows, want_rows) and np.array_equal(got, oracle.result_rows(want_rows))


def test_entries_of_the_build_in_host_memory(gpu, oracle, monkeypatch):
    """a table whose entries would crowd the card is built with its entries in pinned host memory (MC_SKM_ENTRIES=host forces
    it): the same lines"""
    k = 31
    genomes, sz, ky, lb = small_db(k=k, glen=6000)
    codes, _ = synth.sample_reads(genomes, 2000, 150, seed=21)
    rp, con = synth.pack_uniform(codes)
    info_dev, _ = _check(gpu, oracle, k, sz, ky, lb, rp, con, 6)
    monkeypatch.setenv("MC_SKM_ENTRIES", "host")
    info_host, _ = _check(gpu, oracle, k, sz, ky, lb, rp, con, 6)
    assert info_host == info_dev
