"""GPU suite: the in-HBM minimizer index beyond the plain case -- line-range parts (what the GPUs of a
multi-GPU job hold), the streamed two-pass build, crowded minimizers (extra lines, hashed chains).
Everything is compared with the oracle on the unsharded table: bit-exact."""
import os

import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta, pack_with_oracle

pytestmark = pytest.mark.gpu

K, HT = 21, 1000003


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    from jn_cuclark_amd import CuClarkDB
    return CuClarkDB


def _files(oracle, tmp_path, sz, ky, lb, ht=HT, name="db"):
    nzb = np.flatnonzero(sz)
    canon = np.repeat(nzb, sz[nzb]).astype(np.uint64) + ky.astype(np.uint64) * np.uint64(ht)
    base = str(tmp_path / name)
    oracle.db_write(base, ht, 4, canon, lb)
    return base


def _to_dev(rp, con):
    import torch
    dev = torch.device("cuda:0")
    return torch.from_numpy(rp.view(np.int32)).to(dev), torch.from_numpy(con.view(np.int16)).to(dev)


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_line_range_parts_add_up_to_the_unsharded_result(gpu, oracle, tmp_path, n_parts):
    """every part sees every read and answers for the k-mers of ITS minimizer lines; the k-way merge of
    the parts' rows and the top-2 on it equal the oracle on the whole table (rows and final)"""
    import torch
    genomes, sz, ky, lb = small_db(n_targets=8, glen=6000)
    base = _files(oracle, tmp_path, sz, ky, lb)
    names, seqs = mixed_fasta(genomes, K, n=3000)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs, width=70), K)
    want_rows, _ = oracle.OracleDB.from_arrays(HT, sz, ky, lb).query_rows(K, rp, con, 15)
    want = oracle.result_rows(want_rows)
    n = rp.size - 1
    rp_t, con_t = _to_dev(rp, con)
    st = torch.cuda.current_stream().cuda_stream
    parts, owned, lines = [], 0, 0
    dbs = []
    for p in range(n_parts):
        db = gpu(k=K, numBatches=1, numTargets=8, device=0, htsize=HT, maxhits=15)
        assert db.read_part(base, p, n_parts) is True
        info = db.db_info()
        assert info["index_kind"] == 1 and info["part"] == p and info["n_parts"] == n_parts
        assert info["n_keys"] == ky.size                       # the whole table streamed past
        owned += info["n_keys_owned"]
        lines += info["line_end"] - info["line_begin"]
        rows = torch.zeros((n, db.row_len), dtype=torch.int16, device=rp_t.device)
        db.query_device(rp_t, con_t, rows_t=rows, stream=st)
        parts.append(rows)
        dbs.append(db)
    torch.cuda.synchronize()
    assert owned == ky.size and lines == dbs[0].db_info()["n_lines"]
    out_rows = torch.zeros_like(parts[0])
    fin = torch.zeros((n, 5), dtype=torch.int16, device=rp_t.device)
    dbs[0].merge_result_device(parts, n, rows_t=out_rows, final_t=fin, stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(out_rows.cpu().numpy().view(np.uint16), want_rows)
    assert np.array_equal(fin.cpu().numpy().view(np.uint16), want)
    # the parts are not trivial: each holds hits
    assert all(int((p_[:, 0] != 0).sum()) > 100 for p_ in parts)
    # the same combine through the pairwise kernels
    acc = parts[0].clone()
    for p_ in parts[1:]:
        dbs[0].merge_rows_device(acc, p_, acc, n, stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy().view(np.uint16), want_rows)
    for db in dbs:
        db.close()


def test_streamed_build_from_uneven_chunks_equals_the_one_shot_build(gpu, oracle):
    """mc_index_begin / add / next_pass / add / end with ragged host chunks, whole table and one of three
    line parts; device chunks as well"""
    import torch
    genomes, sz, ky, lb = small_db(n_targets=5, glen=5000)
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    cuts = [0, 1, 17, 4096, 4097, 300000, 300001, 999000, HT]

    def chunks():
        for a, b in zip(cuts[:-1], cuts[1:]):
            yield sz[a:b], ky[off[a]:off[b]], lb[off[a]:off[b]], a, b

    names, seqs = mixed_fasta(genomes, K, n=1500)
    _, rp, con = pack_with_oracle(oracle, synth.fasta_text(names, seqs), K)
    odb = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    want_rows, _ = odb.query_rows(K, rp, con, 15)
    with gpu(k=K, numBatches=1, numTargets=5, device=0, htsize=HT, maxhits=15) as db:
        db.read_chunks(chunks, ky.size)
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    assert info["n_keys"] == ky.size and info["n_keys_owned"] == ky.size and info["shard_begin"] == 0 and info["shard_end"] == HT
    assert np.array_equal(rows, want_rows) and np.array_equal(got, oracle.result_rows(want_rows))
    dev = torch.device("cuda:0")

    def dchunks():
        for s_, k_, l_, a, b in chunks():
            yield (torch.from_numpy(s_.copy()).to(dev), torch.from_numpy(k_.view(np.int32).copy()).to(dev),
                   torch.from_numpy(l_.view(np.int16).copy()).to(dev), a, b)

    rp_t, con_t = _to_dev(rp, con)
    n = rp.size - 1
    parts = []
    for p in range(3):
        with gpu(k=K, numBatches=1, numTargets=5, device=0, htsize=HT, maxhits=15) as db:
            db.read_chunks(dchunks, ky.size, part=p, n_parts=3, device=True)
            rows_t = torch.zeros((n, db.row_len), dtype=torch.int16, device=dev)
            db.query_device(rp_t, con_t, rows_t=rows_t, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            parts.append(rows_t)
    with gpu(k=K, numBatches=1, numTargets=5, device=0, htsize=HT, maxhits=15) as db:
        out = torch.zeros_like(parts[0])
        db.merge_result_device(parts, n, rows_t=out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want_rows)
    # feeding different tables in the two passes is refused
    from jn_cuclark_amd import McError
    with gpu(k=K, numBatches=1, numTargets=5, device=0, htsize=HT, maxhits=15) as db:
        state = {"n": 0}

        def bad():
            state["n"] += 1
            lim = len(cuts) - (1 if state["n"] == 1 else 2)
            for a, b in zip(cuts[:lim], cuts[1:lim + 1]):
                yield sz[a:b], ky[off[a]:off[b]], lb[off[a]:off[b]], a, b
        with pytest.raises(McError):
            db.read_chunks(bad, ky.size)


def _mmer_key_model(cw):
    """python twin of mc::mz::key_of_canonical (order only): (t, top 20 bits of the product)"""
    lo, hi = cw & 0xFFFFFFFF, cw >> 32
    p = lo * 0x9E3779B1
    t = (p & 0xFFFFFFFF) ^ ((hi * 0x85EBCA6B) & 0xFFFFFFFF)
    return (t << 20) | (p >> 44)


def test_one_minimizer_shared_by_100000_kmers_is_spread_over_hashed_chains(gpu, oracle):
    """A conserved m-mer in thousands of contexts (a 16S-like region across many genomes): > 10^5 stored
    k-mers share ONE minimizer line.  A chain is capped at 3 extra lines; the crowded line gets 2^s chains and
    a k-mer's chain is picked by a hash of the k-mer, so a lookup reads a bounded number of lines; the index
    is kept for the whole table and every lookup is still exact."""
    k, ht, m = 21, 1000003, 13           # 9 windows per k-mer: m = k - 8 (csrc/mc_minimizer.hpp MZ_MAXW)
    rng = np.random.default_rng(7)
    # an m-mer whose key is far below anything a random window offers
    cand = rng.integers(0, 1 << (2 * m), size=200000, dtype=np.uint64)
    rc = synth.revcomp(cand, m)
    canon = np.minimum(cand, rc)
    keys = np.array([_mmer_key_model(int(c)) for c in canon[:20000]], dtype=np.uint64)
    X = int(cand[int(np.argmin(keys))])
    # k-mers = o free bases | X | 8 - o free bases for every offset o = 0..8, all distinct (free part enumerated)
    free = rng.choice(9 << 16, size=130000, replace=False).astype(np.uint64)
    off, bits = free >> np.uint64(16), free & np.uint64(0xFFFF)
    rbits = np.uint64(2) * (np.uint64(8) - off)                     # bits of the free bases behind X
    left, right = bits >> rbits, bits & ((np.uint64(1) << rbits) - np.uint64(1))
    kmers = (left << (rbits + np.uint64(2 * m))) | (np.uint64(X) << rbits) | right
    ck = np.unique(synth.canonical(kmers, k))
    stored, absent = ck[:110000], ck[110000:]
    labels = (np.arange(stored.size) % 7).astype(np.uint16)
    genomes, sz0, ky0, lb0 = small_db(n_targets=7, glen=3000)       # plus an ordinary table around it
    nzb = np.flatnonzero(sz0)
    base_canon = np.repeat(nzb, sz0[nzb]).astype(np.uint64) + ky0.astype(np.uint64) * np.uint64(ht)
    keep = ~np.isin(base_canon, ck)
    allc = np.concatenate([stored, base_canon[keep]])
    alll = np.concatenate([labels, lb0[keep]])
    sz, ky, lb = synth.db_from_kmers(allc, alll, ht)
    # reads: one k-mer each -- stored ones (hits), absent ones with the same minimizer (misses that walk the
    # hashed chain to its end), plus ordinary reads
    q = np.concatenate([stored[::3], absent, synth.revcomp(stored[1::50], k)])
    codes = np.zeros((q.size, k), dtype=np.uint8)
    for j in range(k):
        codes[:, j] = ((q >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.uint8)
    rp1, con1 = synth.pack_uniform(codes)
    c2, _ = synth.sample_reads(genomes, 500, 150, seed=3)
    rp2, con2 = synth.pack_uniform(c2)
    rp = np.concatenate([rp1[:-1], rp2 + rp1[-1]]).astype(np.uint32)
    con = np.concatenate([con1, con2])
    want_rows, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).query_rows(k, rp, con, 15)
    with gpu(k=k, numBatches=1, numTargets=7, device=0, htsize=ht, maxhits=15) as db:
        db.read_arrays(sz, ky, lb)
        info = db.db_info()
        got, rows = db.classify(rp, con, extended=True)
    assert info["index_kind"] == 1 and info["index_fallback"] == 0
    assert info["largest_line"] >= 100000 and info["n_spilled_keys"] >= 100000 - 48 and info["n_lines_crowded"] >= 1
    assert 3 * 8192 <= info["n_extra_lines"] < 3 * 40000             # 2^s chains of 3 lines at ~9 k-mers per chain
    assert np.array_equal(rows, want_rows)
    assert np.array_equal(got, oracle.result_rows(want_rows))
    n_hit = (stored[::3].size + stored[1::50].size)
    assert int((got[:q.size, 2] > 0).sum()) == n_hit


@pytest.mark.parametrize("part,n_parts", [(0, 4096), (5, 8)])
def test_a_part_of_a_table_that_needs_sharding_gets_its_lines(gpu, part, n_parts):
    """configs[3]/[4] scale: 32e9 k-mers.  Round 2 refused this ("too many lines": 8e9 global lines in a 32-bit
    index); a part now has a line space of its own.  The part is opened, fed an empty bucket range in both passes
    and closed: fill, line count and HBM are what mc_index_plan says for a card of this size (part 5 of 8 really
    allocates its 170 GB of lines).  Reference: a table is cut into as many parts as memory dictates and runs
    whatever its size (src/CuClarkDB.cu:516-559)."""
    import torch
    from jn_cuclark_amd import _lib
    n_keys = 32_000_000_000
    total = torch.cuda.get_device_properties(0).total_memory
    plan = _lib.index_plan(n_keys, n_parts, total)
    assert plan["fits"] == 1 and plan["fill"] == 3.0
    empty = (np.zeros(4096, dtype=np.uint8), np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint16), 0, 4096)
    with gpu(k=31, numBatches=1, numTargets=8192, device=0, htsize=1610612741, maxhits=15) as db:
        db.read_chunks(lambda: [empty], n_keys, part=part, n_parts=n_parts)
        info = db.db_info()
        assert info["index_kind"] == 1 and info["part"] == part and info["n_parts"] == n_parts
        assert info["line_end"] - info["line_begin"] == plan["lines_per_part"] and info["line_begin"] == part * plan["lines_per_part"]
        assert info["n_lines"] == plan["lines_per_part"] * n_parts > 2 ** 32        # more lines than one 32-bit space
        assert plan["lines_per_part"] * 128 <= info["device_bytes"] <= plan["bytes_per_part"]     # lines + the room reserved for extra lines
        assert info["n_keys_owned"] == 0
        # and it answers (nothing stored: no hits)
        got = db.classify(np.array([0, 21], dtype=np.uint32), np.concatenate([[150], np.arange(20)]).astype(np.uint16))
        assert np.array_equal(got, np.zeros((1, 5), dtype=np.uint16))


def test_a_second_pass_with_other_kmers_is_refused_not_written_out_of_bounds(gpu, oracle):
    """The placing pass trusts the counts of the first.  Same NUMBER of k-mers, other k-mers (a generator that is not
    deterministic, a file that changed): a line that outgrows the chain sized for it sets a flag instead of
    writing past its extra lines; mc_index_end reports MC_EINVAL."""
    from jn_cuclark_amd import McError
    k, ht, m = 21, 1000003, 13
    rng = np.random.default_rng(11)
    n = 60000
    a = np.unique(synth.canonical(rng.integers(0, 1 << (2 * k), size=n + 500, dtype=np.uint64), k))[:n]
    # n k-mers around ONE m-mer: they all share its minimizer line
    X = np.uint64(rng.integers(0, 1 << (2 * m)))
    free = rng.choice(1 << 16, size=n + 500, replace=False).astype(np.uint64)
    b = np.unique(synth.canonical(((free >> np.uint64(8)) << np.uint64(2 * m + 8)) | (X << np.uint64(8)) | (free & np.uint64(0xFF)), k))
    assert b.size >= 40000
    b = np.concatenate([b, a])[:n]
    lab = np.zeros(n, dtype=np.uint16)
    ta, tb = synth.db_from_kmers(a, lab, ht), synth.db_from_kmers(b, lab, ht)
    state = {"n": 0}

    def chunks():
        state["n"] += 1
        sz, ky, lb = ta if state["n"] == 1 else tb
        yield sz, ky, lb, 0, ht

    with gpu(k=k, numBatches=1, numTargets=1, device=0, htsize=ht, maxhits=15) as db:
        with pytest.raises(McError) as e:
            db.read_chunks(chunks, n)
        assert "second pass" in str(e.value)
        # the context is usable afterwards
        db.read_arrays(*ta)
        assert db.db_info()["n_keys"] == n


def test_a_chunk_with_fewer_kmers_than_its_bucket_sizes_say_is_refused(gpu, oracle):
    """mc_index_add_* with n_keys below the sum of the chunk's bucket sizes: the build kernels must not read past the
    arrays the caller passed (they stop at n_keys), the mismatch is reported as MC_EINVAL at the pass boundary, and the
    context loads the next table as if nothing had happened."""
    import ctypes as C
    from jn_cuclark_amd import McError
    from jn_cuclark_amd._lib import check
    _, sz, ky, lb = small_db(n_targets=4, glen=5000)
    n = int(ky.size)
    short = n - 1000
    with gpu(k=K, numBatches=1, numTargets=4, device=0, htsize=HT, maxhits=15) as db:
        check(db._lib.mc_index_begin(db._h, short, 0, 1))
        # the arrays really are `short` long: anything read behind them is out of bounds
        ky_s, lb_s = np.ascontiguousarray(ky[:short]), np.ascontiguousarray(lb[:short])
        check(db._lib.mc_index_add_host(db._h, sz.ctypes.data, ky_s.ctypes.data, ky_s.dtype.itemsize, lb_s.ctypes.data, short, 0, HT))
        with pytest.raises(McError) as e:
            check(db._lib.mc_index_next_pass(db._h))
        assert "do not sum" in str(e.value)
        db.read_arrays(sz, ky, lb)
        assert db.db_info()["n_keys"] == n
