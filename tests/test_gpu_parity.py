"""GPU suite: the HIP path (through the C ABI of libmcclark.so) against the oracle on
the same seeded inputs.  Bit-exact: everything on this path is integer."""
import os

import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta, pack_with_oracle

pytestmark = pytest.mark.gpu

K, HT = 21, 1000003


@pytest.fixture(params=["minimizer", "lines"], autouse=True)
def index_mode(request, monkeypatch):
    """every test runs on both in-HBM indexes: the minimizer index (default) and the direct
    bucket-line table (MC_INDEX=lines)"""
    monkeypatch.setenv("MC_INDEX", request.param)
    return request.param


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    from jn_cuclark_amd import CuClarkDB
    return CuClarkDB


def _open(gpu, sz, ky, lb, k=K, ht=HT, maxhits=15, ntargets=16, shard=(0, 0)):
    db = gpu(k=k, numBatches=2, numTargets=ntargets, device=0, htsize=ht, maxhits=maxhits)
    db.read_arrays(sz, ky, lb, shard=shard)
    return db


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_mixed_reads_bit_exact(gpu, oracle, fmt, index_mode):
    """ragged input: N-split reads, short reads, short parts, lower case, U, all-N"""
    genomes, sz, ky, lb = small_db()
    names, seqs = mixed_fasta(genomes, K, n=2000)
    text = synth.fastq_text(names, seqs) if fmt == "fastq" else synth.fasta_text(names, seqs, width=70)
    _, rp, con = pack_with_oracle(oracle, text, K)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    want_rows, _ = odb.query_rows(K, rp, con, 15)
    want = oracle.result_rows(want_rows)
    with _open(gpu, sz, ky, lb) as db:
        got, rows = db.classify(rp, con, extended=True)
        info = db.db_info()
    assert np.array_equal(got, want)
    assert np.array_equal(rows, want_rows)
    assert info["n_keys"] == ky.size and info["line_bytes"] == (64 if index_mode == "lines" else 128)
    assert (want[:, 2] > 0).sum() > 1000


@pytest.mark.parametrize("length", [21, 22, 84, 85, 86, 139, 140, 141, 142, 143, 144, 145, 148, 149, 150, 151, 269, 272, 277, 501, 1000])
def test_read_lengths_around_wave_boundaries(gpu, oracle, length):
    """k-mers per read around 64 / 128 (one / two wave steps) and long reads"""
    genomes, sz, ky, lb = small_db(glen=6000)
    codes, _ = synth.sample_reads(genomes, 300, length, seed=length)
    rp, con = synth.pack_uniform(codes)
    want, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).classify(K, rp, con, 15)
    with _open(gpu, sz, ky, lb) as db:
        got = db.classify(rp, con)
    assert np.array_equal(got, want)


def test_empty_and_tiny_batches(gpu, oracle):
    genomes, sz, ky, lb = small_db()
    with _open(gpu, sz, ky, lb) as db:
        got = db.classify(np.zeros(1, dtype=np.uint32), np.zeros(0, dtype=np.uint16))
        assert got.shape == (0, 5)
        # reads without any container (all shorter than k)
        rp = np.zeros(6, dtype=np.uint32)
        got = db.classify(rp, np.zeros(0, dtype=np.uint16))
        assert np.array_equal(got, np.zeros((5, 5), dtype=np.uint16))
        # a single read
        codes, _ = synth.sample_reads(genomes, 1, 150, seed=1)
        rp, con = synth.pack_uniform(codes)
        want, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).classify(K, rp, con, 15)
        assert np.array_equal(db.classify(rp, con), want)


def test_long_multi_part_read_uses_unstaged_path(gpu, oracle):
    """one read larger than a wave's LDS slice (contig-like, several parts)"""
    genomes, sz, ky, lb = small_db(glen=30000, n_targets=4)
    seq = b"N".join(synth.codes_to_ascii(g[:9000 + 13 * i]) for i, g in enumerate(genomes))
    short = synth.codes_to_ascii(genomes[1][50:200])
    text = synth.fasta_text([b"contig", b"r1", b"contig2", b"r2"], [seq, short, seq[::-1], short], width=80)
    _, rp, con = pack_with_oracle(oracle, text, K)
    want_rows, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(K, rp, con, 15)
    with _open(gpu, sz, ky, lb) as db:
        got, rows = db.classify(rp, con, extended=True)
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    assert got[0, 0] > 30000


def test_more_targets_than_maxhits(gpu, oracle):
    """reference: undefined; ours: keep the maxhits smallest ids, count the read"""
    genomes, sz, ky, lb = small_db(n_targets=12, glen=1500, shared=0)
    seq = b"N".join(synth.codes_to_ascii(g[100:140]) for g in genomes)
    text = synth.fasta_text([b"chimera", b"plain"], [seq, synth.codes_to_ascii(genomes[3][:150])])
    _, rp, con = pack_with_oracle(oracle, text, K)
    want_rows, ovf = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(K, rp, con, 5)
    assert ovf == 1
    with _open(gpu, sz, ky, lb, maxhits=5) as db:
        got, rows = db.classify(rp, con, extended=True)
        st = db.stats()
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    assert st["reads_over_maxhits"] == 1


def test_many_targets_per_read_above_64(gpu, oracle):
    """more distinct targets than accumulator lanes: the 64 smallest ids survive"""
    genomes, sz, ky, lb = small_db(n_targets=90, glen=400, shared=0)
    order = [(i * 37) % 90 for i in range(90)]
    seq = b"N".join(synth.codes_to_ascii(genomes[i][50:50 + 30 + (i % 7)]) for i in order)
    text = synth.fasta_text([b"zoo"], [seq])
    _, rp, con = pack_with_oracle(oracle, text, K)
    want_rows, ovf = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(K, rp, con, 63)
    with _open(gpu, sz, ky, lb, maxhits=63, ntargets=90) as db:
        got, rows = db.classify(rp, con, extended=True)
    assert ovf == 1 and np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))


@pytest.mark.parametrize("line", [64, 128])
def test_dense_buckets_overflow_table(gpu, oracle, line, monkeypatch, index_mode):
    """heavily loaded table: buckets beyond a line's capacity go to the side table"""
    ht = 4099
    sz, ky, lb = synth.random_db(seed=2, htsize=ht, n_keys=60000, n_targets=40, k=K)   # ~14.6 / bucket
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    # reads made of stored k-mers (hits) and of random sequence (misses)
    kmers = np.concatenate([ky[off[r]:off[r + 1]].astype(np.uint64) * np.uint64(ht) + np.uint64(r) for r in range(0, ht, 7)])
    codes = np.zeros((kmers.size, 50), dtype=np.uint8)
    rnd = synth.random_codes(77, kmers.size * 50).reshape(kmers.size, 50)
    codes[:] = rnd
    for j in range(K):
        codes[:, 10 + j] = ((kmers >> np.uint64(2 * (K - 1 - j))) & np.uint64(3)).astype(np.uint8)
    rp, con = synth.pack_uniform(codes)
    want, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).classify(K, rp, con, 15)
    monkeypatch.setenv("MC_LINE_BYTES", str(line))
    with _open(gpu, sz, ky, lb, ht=ht, ntargets=40) as db:
        info = db.db_info()
        got = db.classify(rp, con)
    # (the minimizer index spreads these random k-mers evenly: whether a line overflows there is up to the hash;
    # its extra lines have their own tests, test_minimizer_extra_line_chains and tests/test_gpu_index.py)
    assert index_mode == "minimizer" or (info["line_bytes"] == line and info["n_overflow_buckets"] > 0)
    assert np.array_equal(got, want)
    assert (want[:, 2] > 0).mean() > 0.45


@pytest.mark.parametrize("k", [16, 17, 18])
def test_smallest_k_of_the_minimizer_index(gpu, oracle, k, index_mode):
    """k = 16 is the smallest k the minimizer index takes (m = k - 8 = 8); the kernel's second reverse complement has a
    code path of its own below k = 17 (the complemented base lands in the low word).  Reads with N, lower case, short
    parts; rows and results against the oracle."""
    ht = 100003
    genomes = synth.toy_genomes(4, 1500, 13, shared=100)
    sz, ky, lb = synth.genome_db(genomes, k, ht)
    names, seqs = mixed_fasta(genomes, k, seed=4, n=800)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), k)
    want_rows, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).query_rows(k, rp, con, 15)
    with _open(gpu, sz, ky, lb, k=k, ht=ht) as db:
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    if index_mode == "minimizer":
        assert info["index_kind"] == 1 and info["index_fallback"] == 0
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    assert (got[:, 2] > 0).sum() > 300


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_sharded_db_merge_result_equals_unsharded(gpu, oracle, shards):
    """bucket-range shards -> sparse rows -> merge -> top-2 == single shard (one GPU
    plays every shard in turn; the exchange itself is covered by test_distributed)"""
    import torch
    genomes, sz, ky, lb = small_db(n_targets=10)
    codes, _ = synth.sample_reads(genomes, 3000, 150, seed=6)
    rp, con = synth.pack_uniform(codes)
    want, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).classify(K, rp, con, 15)
    dev = torch.device("cuda:0")
    rp_t = torch.from_numpy(rp.view(np.int32)).to(dev)
    con_t = torch.from_numpy(con.view(np.int16)).to(dev)
    n = rp.size - 1
    acc = None
    bounds = [HT * i // shards for i in range(shards + 1)]
    for s in range(shards):
        with _open(gpu, sz, ky, lb, shard=(bounds[s], bounds[s + 1])) as db:
            rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
            db.query_device(rp_t, con_t, rows_t=rows, stream=torch.cuda.current_stream().cuda_stream)
            if acc is None:
                acc = rows
            else:
                db.merge_rows_device(acc, rows, acc, n, stream=torch.cuda.current_stream().cuda_stream)
            if s == shards - 1:
                fin = torch.zeros((n, 5), dtype=torch.int16, device=dev)
                db.result_rows_device(acc, fin, n, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
    assert np.array_equal(fin.cpu().numpy().view(np.uint16), want)


@pytest.mark.parametrize("fixture,maxhits", [("db_light_k27.npz", 23), ("db_full_k31.npz", 15)])
def test_reference_built_tables_against_reference_find(gpu, oracle, golden_dir, fixture, maxhits):
    """The databases the REFERENCE's own code built and the lookups the reference's hTable::find
    answered (hashTable_hh.hh:358-396; fixtures by tests/golden/make_golden.py): cuCLARK-l's
    HTSIZE 57777779 / k=27 and the metric's own HTSIZE 1610612741 / k=31, on both indexes."""
    g = np.load(os.path.join(golden_dir, fixture), allow_pickle=False)
    k, ht = int(g["k"]), int(g["htsize"])
    sz = np.zeros(ht, dtype=np.uint8)
    sz[g["nonzero_buckets"]] = g["nonzero_sizes"]
    q = g["query_kmers"]
    # one k-mer per read
    codes = np.zeros((q.size, k), dtype=np.uint8)
    for j in range(k):
        codes[:, j] = ((q >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.uint8)
    rp, con = synth.pack_uniform(codes)
    found = g["query_found"].astype(bool)
    qlab = g["query_label"].astype(np.int64)
    per = 1000                                   # k-mers per long read
    n_long = q.size // per
    perm = np.random.default_rng(5).permutation(q.size)      # the fixture lists its k-mers target by target: mix them
    with _open(gpu, sz, g["keys"], g["labels"], k=k, ht=ht, maxhits=maxhits, ntargets=3) as db:
        got = db.classify(rp, con)
        # (2) the same k-mers as long reads of 1000 PARTS each ('N' between the k-mers: a part = one k-mer), so the
        #     per-target sums of the reference's answers are the expected counts: multi-part reads, the accumulator
        part = con.reshape(q.size, -1)[perm]                 # [k][containers] per k-mer
        words = part.shape[1]
        rp_n = (np.arange(n_long + 1, dtype=np.uint64) * np.uint64(per * words)).astype(np.uint32)
        got_n = db.classify(rp_n, part[: n_long * per].reshape(-1))
        # (3) ... and as ONE part per long read (the k-mers back to back, no 'N'): consecutive positions share
        #     minimizers, lines, runs and steps; the k - 1 k-mers across every junction are looked up in the
        #     reference-built table here, in numpy
        long_codes = codes[perm][: n_long * per].reshape(n_long, per * k)
        rp_c, con_c = synth.pack_uniform(long_codes)
        got_c = db.classify(rp_c, con_c)
    assert found.sum() > 5000 and (~found).sum() > 5000
    assert np.array_equal(got[:, 2] > 0, found)
    assert np.array_equal(got[found, 1] - 1, g["query_label"][found])
    assert np.all(got[found, 0] == 1)

    def final_of(counts):        # [n, T] hit counts -> the 5 result columns (ascending scan, strict '>': CuClarkDB.cu:1380-1397)
        out = np.zeros((counts.shape[0], 5), dtype=np.uint16)
        out[:, 0] = counts.sum(axis=1) & 0xFFFF
        best = counts.argmax(axis=1)                       # first maximum = smallest id
        bc = counts[np.arange(counts.shape[0]), best]
        rest = counts.copy()
        rest[np.arange(counts.shape[0]), best] = -1
        sec = rest.argmax(axis=1)
        sc = rest[np.arange(counts.shape[0]), sec]
        out[:, 1] = np.where(bc > 0, best + 1, 0)
        out[:, 2] = bc
        out[:, 3] = np.where(sc > 0, sec + 1, 0)
        out[:, 4] = np.maximum(sc, 0)
        return out

    T = 3
    cnt = np.zeros((n_long, T), dtype=np.int64)
    f_l, l_l = found[perm][: n_long * per].reshape(n_long, per), qlab[perm][: n_long * per].reshape(n_long, per)
    for t in range(T):
        cnt[:, t] = (f_l & (l_l == t)).sum(axis=1)
    assert cnt.sum() == f_l.sum() and (cnt > 50).all()            # every long read hits all three targets
    assert np.array_equal(got_n, final_of(cnt))
    # junction k-mers of (3): every position of the long read that is not a multiple of k
    nzb = g["nonzero_buckets"].astype(np.uint64)
    canon_db = np.repeat(nzb, g["nonzero_sizes"]) + g["keys"].astype(np.uint64) * np.uint64(ht)
    order = np.argsort(canon_db)
    canon_sorted, lab_sorted = canon_db[order], g["labels"].astype(np.int64)[order]
    cnt_c = cnt.copy()
    extra = 0
    for r in range(n_long):
        km = synth.canonical(synth.kmers_of(long_codes[r], k), k)
        pos = np.arange(km.size)
        km = km[pos % k != 0]
        at = np.searchsorted(canon_sorted, km)
        at[at >= canon_sorted.size] = 0
        hit = canon_sorted[at] == km
        extra += int(hit.sum())
        for t in range(T):
            cnt_c[r, t] += int((hit & (lab_sorted[at] == t)).sum())
    assert np.array_equal(got_c, final_of(cnt_c))


def test_file_loader_and_sampling(gpu, oracle, tmp_path):
    genomes, sz, ky, lb = small_db()
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    nzb = np.flatnonzero(sz)
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(HT)
    base = str(tmp_path / "db_central_k21_t6_s1000003_m0.tsk")
    oracle.db_write(base, HT, 4, canon, lb)
    codes, _ = synth.sample_reads(genomes, 1500, 150, seed=12)
    rp, con = synth.pack_uniform(codes)
    from jn_cuclark_amd import CuClarkDB
    for s in (1, 3):
        want, _ = oracle.OracleDB.load(base, HT, 4, sampling=s).classify(K, rp, con, 15)
        with CuClarkDB(k=K, numBatches=1, numTargets=6, device=0, htsize=HT, maxhits=15) as db:
            assert db.read(base, modCollision=s) is True
            got = db.classify(rp, con)
        assert np.array_equal(got, want)
        assert (want[:, 2] > 0).sum() > 500
    with CuClarkDB(k=K, numBatches=1, numTargets=6, device=0, htsize=HT, maxhits=15) as db:
        assert db.read(str(tmp_path / "nope")) is False     # caller rebuilds (CuCLARK_hh.hh:622-684)


def test_file_loader_over_several_chunks_with_sampling(gpu, oracle, tmp_path):
    """cuCLARK-l's table size (57 777 779 buckets = four chunks of the loader's pipeline: reader thread, two pinned
    buffers, staging sets) with and without the -s sampling rule (CuClarkDB.cu:503-513: the counter runs over the
    non-empty buckets of the WHOLE table, the unsampled buckets are cut out of every chunk); on both indexes"""
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(6, 30000, seed=77, shared=500)
    sz, ky, lb = synth.genome_db(genomes, k, ht)
    nzb = np.flatnonzero(sz)
    assert nzb[0] < (1 << 24) and nzb[-1] > 3 * (1 << 24)            # k-mers in the first and in the last chunk
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(ht)
    base = str(tmp_path / "db_central_k27_t6_s57777779_m0_light_4.tsk")
    oracle.db_write(base, ht, 4, canon, lb)
    codes, _ = synth.sample_reads(genomes, 3000, 150, seed=13)
    rp, con = synth.pack_uniform(codes)
    from jn_cuclark_amd import CuClarkDB
    for s_ in (1, 2, 5):
        want, _ = oracle.OracleDB.load(base, ht, 4, sampling=s_).classify(k, rp, con, 23)
        with CuClarkDB(k=k, numBatches=1, numTargets=6, device=0, htsize=ht, maxhits=23) as db:
            assert db.read(base, modCollision=s_) is True
            got = db.classify(rp, con)
        assert np.array_equal(got, want), s_
        assert (want[:, 2] > 0).sum() > (2500 if s_ == 1 else 1500)


def test_batched_streaming_interface(gpu, oracle):
    """several batches in flight through malloc/readyBatch/queryBatch/waitForBatch"""
    genomes, sz, ky, lb = small_db()
    nb = 5
    from jn_cuclark_amd import CuClarkDB
    with CuClarkDB(k=K, numBatches=nb, numTargets=6, device=0, htsize=HT, maxhits=15) as db:
        db.read_arrays(sz, ky, lb)
        assert db.swapDbParts() is True
        batches = []
        for b in range(nb):
            codes, _ = synth.sample_reads(genomes, 700 + 13 * b, 150, seed=20 + b)
            batches.append(synth.pack_uniform(codes))
        rp_l, con_l, fin_l, _ = db.malloc(max(p.size for p, _ in batches), max(c.size for _, c in batches))
        for b, (p, c) in enumerate(batches):
            rp_l[b][: p.size] = p
            con_l[b][: c.size] = c
            db.readyBatch(b, p.size - 1, c.size)
            assert db.queryBatch(b) is True
        assert db.swapDbParts() is False
        odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
        for b, (p, c) in enumerate(batches):
            db.waitForBatch(b)
            n = p.size - 1
            want, _ = odb.classify(K, p, c, 15)
            assert np.array_equal(fin_l[b][: n * 5].reshape(n, 5), want)
        db.freeBatchMemory()


def test_errors_are_reported_not_fatal(gpu):
    from jn_cuclark_amd import CuClarkDB, McError
    with pytest.raises(McError):
        CuClarkDB(k=40, numBatches=1, numTargets=3)
    with CuClarkDB(k=K, numBatches=1, numTargets=3, htsize=HT) as db:
        with pytest.raises(McError):
            db.malloc(10, 100)
            db.readyBatch(0, 1, 3)
            db.queryBatch(0)                                              # no database yet


def test_two_byte_keys_and_sharded_file_load(gpu, oracle, tmp_path):
    """k small enough for 2-byte quotients (reference T16 regime, main.cc:255-263): the
    .ky file holds u16, widened on the device; also a bucket-range shard read from files."""
    k, ht = 17, 1000003
    genomes = synth.toy_genomes(5, 4000, seed=91, shared=100)
    sz, ky, lb = synth.genome_db(genomes, k, ht)
    assert ky.max() < 65536
    nzb = np.flatnonzero(sz)
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(ht)
    base = str(tmp_path / "db16")
    oracle.db_write(base, ht, 2, canon, lb)
    assert os.path.getsize(base + ".ky") == 2 * ky.size
    codes, _ = synth.sample_reads(genomes, 1500, 120, seed=3)
    rp, con = synth.pack_uniform(codes)
    odb = oracle.OracleDB.load(base, ht, 2)
    want, _ = odb.classify(k, rp, con, 15)
    assert (want[:, 2] > 0).sum() > 1000
    with gpu(k=k, numBatches=1, numTargets=5, device=0, htsize=ht, maxhits=15) as db:
        assert db.read(base, key_bytes=2) is True
        assert np.array_equal(db.classify(rp, con), want)
    with gpu(k=k, numBatches=1, numTargets=5, device=0, htsize=ht, maxhits=15) as db:
        db.read_arrays(sz, ky.astype(np.uint16), lb)
        assert np.array_equal(db.classify(rp, con), want)
    # shard [a, b) straight from the files == oracle restricted to the same bucket range
    a, b = ht // 3, 2 * ht // 3
    rows_want, _ = odb.query_rows(k, rp, con, 15, part=(a, b))
    with gpu(k=k, numBatches=1, numTargets=5, device=0, htsize=ht, maxhits=15) as db:
        assert db.read(base, key_bytes=2, shard=(a, b)) is True
        _, rows = db.classify(rp, con, extended=True)
    assert np.array_equal(rows, rows_want)


@pytest.mark.parametrize("line", [64, 128])
def test_wide_keys_k_and_table_size_beyond_32_bit_quotients(gpu, oracle, line, monkeypatch, index_mode):
    """the reference's T64 regime (k = 32 with the full table, main.cc:277-286): quotients
    need 64 bits; here reached with k = 25 on a 100003-bucket table (4^25 / 100003 > 2^32)"""
    k, ht = 25, 100003
    genomes = synth.toy_genomes(6, 30000, seed=95, shared=400)
    sz, ky, lb = synth.genome_db(genomes, k, ht)
    assert ky.dtype == np.uint64 and ky.max() > 0xFFFFFFFF
    names, seqs = mixed_fasta(genomes, k, seed=5, n=1500)
    text = synth.fasta_text(names, seqs)
    _, rp, con = pack_with_oracle(oracle, text, k)
    want_rows, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).query_rows(k, rp, con, 15)
    monkeypatch.setenv("MC_LINE_BYTES", str(line))
    with _open(gpu, sz, ky, lb, k=k, ht=ht, ntargets=6) as db:
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    if index_mode == "lines":
        assert info["line_bytes"] == line and info["line_capacity"] == (6 if line == 64 else 12)
        assert info["n_overflow_buckets"] > 0 or line == 128
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    assert (got[:, 2] > 0).sum() > 800


def test_paired_250bp_reads_are_staged_in_pieces(gpu, oracle):
    """BASELINE config 5 shape: 2 x 250 bp pairs joined by N (501 bases, 66 containers per
    read): 16 of them exceed a wave's LDS slice, so the group is staged in pieces"""
    genomes, sz, ky, lb = small_db(glen=8000, n_targets=5)
    c1, _ = synth.sample_reads(genomes, 700, 250, seed=21)
    c2, _ = synth.sample_reads(genomes, 700, 250, seed=22)
    seqs = [synth.codes_to_ascii(a) + b"N" + synth.codes_to_ascii(b) for a, b in zip(c1, c2)]
    seqs[5] = synth.codes_to_ascii(genomes[0][:7000])           # one contig-like read in the middle
    text = synth.fasta_text([b"p%d" % i for i in range(len(seqs))], seqs)
    _, rp, con = pack_with_oracle(oracle, text, K)
    want, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).classify(K, rp, con, 15)
    with _open(gpu, sz, ky, lb) as db:
        got = db.classify(rp, con)
    assert np.array_equal(got, want)
    assert (want[:, 0] > 300).sum() > 500


@pytest.mark.parametrize("fill", ["12", "40"])
def test_minimizer_extra_line_chains(gpu, oracle, monkeypatch, fill, index_mode):
    """crowded minimizer lines (MC_MZ_FILL k-mers per 12-slot line on average): most lookups
    walk the extra-line chain, guarded by the header's Bloom word"""
    if index_mode != "minimizer":
        pytest.skip("minimizer index only")
    monkeypatch.setenv("MC_MZ_FILL", fill)
    genomes, sz, ky, lb = small_db()
    names, seqs = mixed_fasta(genomes, K, n=1500)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), K)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    want_rows, _ = odb.query_rows(K, rp, con, 15)
    with _open(gpu, sz, ky, lb) as db:
        got, rows = db.classify(rp, con, extended=True)
        info = db.db_info()
    assert info["n_overflow_buckets"] > 0          # extra lines exist
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_random_ragged_batches(gpu, oracle, seed):
    """fuzz: reads of random lengths (k-3 .. 700) cut from the genomes or random, with random runs of N and
    other non-ACGT bytes at random places -- every boundary between parts, steps (128 positions) and
    staged pieces moves around; rows and results must equal the oracle's on both indexes"""
    rng = np.random.default_rng(seed)
    genomes, sz, ky, lb = small_db(glen=7000, n_targets=6)
    names, seqs = [], []
    for i in range(1200):
        L = int(rng.choice([K - 3, K, K + 1, 60, 100, 127 + K, 128 + K, 129 + K, 150, 251, 400, 700]))
        if rng.random() < 0.6:
            g = int(rng.integers(0, len(genomes)))
            p = int(rng.integers(0, genomes[g].size - L))
            codes = genomes[g][p:p + L].copy()
            mut = rng.random(L) < 0.02
            codes[mut] = (codes[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
        else:
            codes = rng.integers(0, 4, size=L).astype(np.uint8)
        s = bytearray(synth.codes_to_ascii(codes))
        for _ in range(int(rng.integers(0, 4))):               # runs of N / IUPAC / '-' of length 1..5
            at = int(rng.integers(0, L))
            for t in range(at, min(L, at + int(rng.integers(1, 6)))):
                s[t] = rng.choice(list(b"NRY-nx"))
        names.append(b"f%d" % i)
        seqs.append(bytes(s))
    text = synth.fasta_text(names, seqs, width=int(rng.choice([60, 80, 1000])))
    _, rp, con = pack_with_oracle(oracle, text, K)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    want_rows, _ = odb.query_rows(K, rp, con, 15)
    with _open(gpu, sz, ky, lb) as db:
        got, rows = db.classify(rp, con, extended=True)
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    assert (got[:, 2] > 0).sum() > 400


def test_counts_saturate_at_65535_in_every_output(gpu, oracle):
    """One rule for per-target counts above 65 535 (the reference's packed u16 atomics
    carry into the neighbouring target there, CuClarkDB.cu:1104-1108): they saturate -- in the fused
    final row, in the sparse row, through result_rows, and through a 2-shard merge whose halves are
    each below the limit.  A contig-like read with > 66 000 hits on one target (two passes over the
    same genome) next to a second target with a few hits."""
    import torch
    genomes, sz, ky, lb = small_db(glen=40000, n_targets=3, shared=0)
    g0 = synth.codes_to_ascii(genomes[0][:35000])
    seq = g0 + b"N" + g0 + b"N" + synth.codes_to_ascii(genomes[2][100:400])
    text = synth.fasta_text([b"contig", b"plain"], [seq, synth.codes_to_ascii(genomes[1][:150])], width=100)
    _, rp, con = pack_with_oracle(oracle, text, K)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    want_rows, _ = odb.query_rows(K, rp, con, 15)
    want = oracle.result_rows(want_rows)
    assert want_rows[0, 0] == 2 and want_rows[0, 2] == 65535 and 200 < want_rows[0, 4] <= 280
    assert want[0, 2] == 65535 and want[0, 1] == 1 and want[0, 0] == (65535 + int(want_rows[0, 4])) % 65536
    n = rp.size - 1
    dev = torch.device("cuda:0")
    rp_t = torch.from_numpy(rp.view(np.int32)).to(dev)
    con_t = torch.from_numpy(con.view(np.int16)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    with _open(gpu, sz, ky, lb) as db:
        got, rows = db.classify(rp, con, extended=True)                 # fused final + rows of one launch
        assert np.array_equal(rows, want_rows) and np.array_equal(got, want)
        fin_t = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.result_rows_device(torch.from_numpy(rows.view(np.int16)).to(dev), fin_t, n, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(fin_t.cpu().numpy().view(np.uint16), want)      # rows -> result_rows
    # two bucket-range shards, each with fewer than 65 536 hits on target 0, merged
    half = HT // 2
    parts = []
    for a, b in ((0, half), (half, HT)):
        pr_want, _ = odb.query_rows(K, rp, con, 15, part=(a, b))
        assert 20000 < pr_want[0, 2] < 50000
        with _open(gpu, sz, ky, lb, shard=(a, b)) as db:
            rt = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
            db.query_device(rp_t, con_t, rows_t=rt, stream=st)
            torch.cuda.synchronize()
            assert np.array_equal(rt.cpu().numpy().view(np.uint16), pr_want)
            parts.append(rt)
    assert np.array_equal(oracle.merge_rows(parts[0].cpu().numpy().view(np.uint16), parts[1].cpu().numpy().view(np.uint16)), want_rows)
    with _open(gpu, sz, ky, lb) as db:
        db.merge_rows_device(parts[0], parts[1], parts[0], n, stream=st)
        fin_t = torch.zeros((n, 5), dtype=torch.int16, device=dev)
        db.result_rows_device(parts[0], fin_t, n, stream=st)
        torch.cuda.synchronize()
    assert np.array_equal(parts[0].cpu().numpy().view(np.uint16), want_rows)
    assert np.array_equal(fin_t.cpu().numpy().view(np.uint16), want)


def test_sharded_kernel_with_integer_division_remainder(gpu, oracle, index_mode):
    """the shard filter's `c mod HTSIZE`: the floating-point remainder needs HTSIZE > 1024 and
    4^k <= HTSIZE * 2^32; a 1009-bucket table takes the 64-bit magic division instead"""
    k, ht = 21, 1009
    genomes = synth.toy_genomes(5, 3000, seed=17, shared=100)
    sz, ky, lb = synth.genome_db(genomes, k, ht)
    names, seqs = mixed_fasta(genomes, k, seed=4, n=800)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs), k)
    odb = oracle.OracleDB.from_arrays(ht, sz, ky, lb)
    for a, b in ((0, 400), (400, 1009)):
        want_rows, _ = odb.query_rows(k, rp, con, 15, part=(a, b))
        with _open(gpu, sz, ky, lb, k=k, ht=ht, ntargets=5, shard=(a, b)) as db:
            _, rows = db.classify(rp, con, extended=True)
        assert np.array_equal(rows, want_rows)
        assert (want_rows[:, 0] > 0).sum() > 200
