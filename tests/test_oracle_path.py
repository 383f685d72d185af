"""CPU suite, part 2: the restated per-read path (packer, scoring, rows, merge, top-2,
CSV).  The reference ships no fixtures for these stages, so they
are checked by construction (independent numpy model) and by the domain's invariants."""
import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta, pack_with_oracle

K = 21       # 4^21 / 1000003 fits the 4-byte keys of the test-size table
HT = 1000003


def _py_parts(seq, k):
    """independent model of the packer: maximal ACGTU runs (newlines ignored), >= k"""
    parts, cur = [], []
    for ch in seq:
        c = chr(ch)
        if c in "ACGTUacgtu":
            cur.append({"A": 3, "C": 2, "G": 1, "T": 0, "U": 0}[c.upper()])
        elif c == "\n":
            continue
        else:
            if len(cur) >= k:
                parts.append(cur)
            cur = []
    if len(cur) >= k:
        parts.append(cur)
    return parts


def _py_pack(seqs, k):
    ptr, con = [0], []
    for s in seqs:
        if len(s.replace(b"\n", b"")) >= k:
            for p in _py_parts(s, k):
                con.append(len(p))
                for i in range(0, len(p), 8):
                    chunk = p[i:i + 8]
                    v = 0
                    for c in chunk:
                        v = (v << 2) | c
                    v <<= 2 * (8 - len(chunk))
                    con.append(v)
        ptr.append(len(con))
    return np.array(ptr, dtype=np.uint32), np.array(con, dtype=np.uint16)


@pytest.mark.parametrize("fmt", ["fasta", "fasta_multiline", "fastq"])
def test_packer_matches_independent_model(oracle, fmt):
    genomes, *_ = small_db()
    names, seqs = mixed_fasta(genomes, K)
    if fmt == "fastq":
        text = synth.fastq_text(names, seqs)
    else:
        text = synth.fasta_text(names, seqs, width=60 if fmt == "fasta_multiline" else 0)
    (ns, ne, sp, ep, ln), rp, con = pack_with_oracle(oracle, text, K)
    assert ln.size == len(seqs)
    for i in (0, 1, 17, len(seqs) - 1):
        assert text[int(ns[i]):int(ne[i])] == names[i].split(b" ")[0]
        assert int(ln[i]) == len(seqs[i])
    erp, econ = _py_pack(seqs, K)
    assert np.array_equal(rp, erp)
    assert np.array_equal(con, econ)


def test_pack_uniform_twin(oracle):
    codes = synth.random_codes(3, 40 * 150).reshape(40, 150)
    seqs = [synth.codes_to_ascii(c) for c in codes]
    text = synth.fasta_text([b"r%d" % i for i in range(40)], seqs)
    _, rp, con = pack_with_oracle(oracle, text, K)
    rp2, con2 = synth.pack_uniform(codes)
    assert np.array_equal(rp, rp2) and np.array_equal(con, con2)


def _dense_model(genomes_db, rp, con, k, htsize):
    """hits per read from a python dict of the database -- independent of the oracle's CSR"""
    sz, ky, lb = genomes_db
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    table = {}
    for r in np.flatnonzero(sz):
        for j in range(off[r], off[r + 1]):
            table[int(ky[j]) * htsize + int(r)] = int(lb[j])
    out = []
    for i in range(rp.size - 1):
        hits = {}
        p = int(rp[i])
        while p < int(rp[i + 1]):
            L = int(con[p]); nc = (L - 1) // 8 + 1
            codes = []
            for c in con[p + 1:p + 1 + nc]:
                codes += [(int(c) >> (14 - 2 * j)) & 3 for j in range(8)]
            codes = np.array(codes[:L], dtype=np.uint8)
            for x in synth.canonical(synth.kmers_of(codes, k), k).tolist():
                if x in table:
                    hits[table[x]] = hits.get(table[x], 0) + 1
            p += 1 + nc
        out.append(hits)
    return out


def _top2(hits):
    best = sbest = ib = isb = 0
    for t in sorted(hits):
        sc = hits[t]
        if sc > best:
            sbest, isb, best, ib = best, ib, sc, t + 1
        elif sc > sbest:
            sbest, isb = sc, t + 1
    return [sum(hits.values()) & 0xFFFF, ib, best, isb, sbest]


def test_query_rows_and_results_match_dense_model(oracle):
    genomes, sz, ky, lb = small_db()
    names, seqs = mixed_fasta(genomes, K, n=120)
    text = synth.fasta_text(names, seqs)
    _, rp, con = pack_with_oracle(oracle, text, K)
    db = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    rows, ovf = db.query_rows(K, rp, con, maxhits=15)
    model = _dense_model((sz, ky, lb), rp, con, K, HT)
    assert ovf == 0
    n_hit_reads = 0
    for i, hits in enumerate(model):
        n = int(rows[i, 0])
        got = {int(rows[i, 1 + 2 * j]): int(rows[i, 2 + 2 * j]) for j in range(n)}
        assert got == hits
        assert list(rows[i, 1:1 + 2 * n:2]) == sorted(hits)
        n_hit_reads += bool(hits)
    assert n_hit_reads > 60
    res = oracle.result_rows(rows)
    assert np.array_equal(res, np.array([_top2(h) for h in model], dtype=np.uint16))
    res2, _ = db.classify(K, rp, con, maxhits=15)
    assert np.array_equal(res, res2)


def test_result_tie_breaks():
    from oracle import pyoracle
    def row(pairs, maxhits=15):
        r = np.zeros((1, 2 * maxhits + 2), dtype=np.uint16)
        r[0, 0] = len(pairs)
        for j, (t, h) in enumerate(pairs):
            r[0, 1 + 2 * j], r[0, 2 + 2 * j] = t, h
        return r
    # strict '>' on an ascending-id scan (reference CuClarkDB.cu:1385-1397)
    assert pyoracle.result_rows(row([(2, 5), (7, 5)]))[0].tolist() == [10, 3, 5, 8, 5]
    assert pyoracle.result_rows(row([(2, 3), (7, 5), (9, 3)]))[0].tolist() == [11, 8, 5, 3, 3]
    assert pyoracle.result_rows(row([(2, 3), (4, 3), (9, 5)]))[0].tolist() == [11, 10, 5, 3, 3]
    assert pyoracle.result_rows(row([]))[0].tolist() == [0, 0, 0, 0, 0]
    assert pyoracle.result_rows(row([(0, 1)]))[0].tolist() == [1, 1, 1, 0, 0]
    # u16 wrap of the sum (reference :1370-1372, :1398)
    assert pyoracle.result_rows(row([(1, 40000), (2, 30000)]))[0].tolist() == [70000 - 65536, 2, 40000, 3, 30000]


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_sharded_merge_equals_unsharded(oracle, shards):
    """bucket-range shards + mergeKernel == one part (reference CuClarkDB.cu:1212-1214, :884-928)"""
    genomes, sz, ky, lb = small_db(n_targets=9)
    codes, _ = synth.sample_reads(genomes, 200, 150, seed=4)
    rp, con = synth.pack_uniform(codes)
    db = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    full, _ = db.query_rows(K, rp, con, maxhits=15)
    bounds = [HT * i // shards for i in range(shards + 1)]
    acc = None
    for s in range(shards):
        part, _ = db.query_rows(K, rp, con, maxhits=15, part=(bounds[s], bounds[s + 1]))
        acc = part if acc is None else oracle.merge_rows(acc, part)
    assert np.array_equal(acc, full)
    assert np.array_equal(oracle.result_rows(acc), db.classify(K, rp, con, 15)[0])


def test_overflow_keeps_smallest_ids_and_is_shard_invariant(oracle):
    """> maxhits distinct targets is UB in the reference; ours: keep the smallest ids."""
    genomes, sz, ky, lb = small_db(n_targets=12, glen=1500, shared=0)
    # one read visiting all 12 genomes: 12 parts of 40 bases joined by N
    seq = b"N".join(synth.codes_to_ascii(g[100:140]) for g in genomes)
    text = synth.fasta_text([b"chimera"], [seq])
    _, rp, con = pack_with_oracle(oracle, text, K)
    db = oracle.OracleDB.from_arrays(HT, sz, ky, lb)
    rows, ovf = db.query_rows(K, rp, con, maxhits=5)
    assert ovf == 1 and rows[0, 0] == 5
    assert rows[0, 1:11:2].tolist() == [0, 1, 2, 3, 4]
    a, _ = db.query_rows(K, rp, con, maxhits=5, part=(0, HT // 2))
    b, _ = db.query_rows(K, rp, con, maxhits=5, part=(HT // 2, HT))
    assert np.array_equal(oracle.merge_rows(a, b), rows)


def test_sampling_factor_load(oracle, tmp_path):
    """-s keeps every s-th non-empty bucket (reference CuClarkDB.cu:503-513)"""
    genomes, sz, ky, lb = small_db()
    off = np.concatenate([[0], np.cumsum(sz.astype(np.int64))])
    canon = np.concatenate([ky[off[r]:off[r + 1]].astype(np.uint64) * np.uint64(HT) + np.uint64(r)
                            for r in np.flatnonzero(sz)])
    base = str(tmp_path / "db")
    oracle.db_write(base, HT, 4, canon, lb)
    db1 = oracle.OracleDB.load(base, HT, 4, sampling=1)
    db3 = oracle.OracleDB.load(base, HT, 4, sampling=3)
    nz = np.flatnonzero(sz)
    for idx, r in enumerate(nz[:60]):
        x = int(ky[off[r]]) * HT + int(r)
        assert db1.lookup(K, x)[0]
        assert db3.lookup(K, x)[0] == ((idx + 1) % 3 == 0)


def test_csv_line_format(oracle):
    # name,gamma,assignment,best,confidence with %g (reference CuCLARK_hh.hh:2110-2118)
    line = oracle.csv_line(b"read1", [90, 3, 60, 5, 30], 150, 31, "562")
    assert line == "read1,0.75,562,60,0.666667\n"
    assert oracle.csv_line(b"x" * 50, [0, 0, 0, 0, 0], 150, 31, "NA") == "x" * 39 + ",0,NA,0,0\n"
    # paired reads are normalised by length-1 (NBN), done by the caller
    assert oracle.csv_line(b"p", [120, 1, 120, 0, 0], 501 - 1, 31, "a").startswith("p,0.255319,a,120,1")


def test_the_restatement_itself_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """oracle/clark_oracle.c is the checker of every GPU test: built with -fsanitize=address,undefined and driven through its
    whole surface in a process of its own (the sanitizer runtimes preloaded into Python) -- indexer and packer on FASTQ and
    multi-line FASTA with N runs, short reads, lower case and other symbols; table from arrays, written to files and read back;
    lookups, sparse rows whole and in two bucket-range parts, merge, top-2, the discriminative-set builder -- no report, and the
    same answers as the normal build gives in this process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liboracle.so")
    r = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-std=c11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared",
                        "-o", so, os.path.join(root, "oracle", "clark_oracle.c")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    pre = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(x) and os.path.exists(x) for x in pre):
        pytest.skip("sanitizer runtimes not found as shared libraries: %r" % (pre,))
    script = tmp_path / "drive.py"
    script.write_text(
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle import pyoracle\n"
        "pyoracle._LIB_PATH = %r\n"
        "pyoracle.build = lambda force=False: pyoracle._LIB_PATH\n"
        "from jn_cuclark_amd import synth\n"
        "from helpers import small_db, mixed_fasta\n"
        "k, ht = %d, %d\n"
        "genomes, sz, ky, lb = small_db(k=k, glen=6000)\n"
        "names, seqs = mixed_fasta(genomes, k, n=3000)\n"
        "out = []\n"
        "for text in (synth.fastq_text(names, seqs), synth.fasta_text(names, seqs, width=70)):\n"
        "    ns, ne, sp, ep, ln = pyoracle.index_reads(text)\n"
        "    rp, con = pyoracle.pack_reads(text, sp, ep, ln, k)\n"
        "    odb = pyoracle.OracleDB.from_arrays(ht, sz, ky, lb)\n"
        "    fin, over = odb.classify(k, rp, con, 15)\n"
        "    rows, _ = odb.query_rows(k, rp, con, 15)\n"
        "    a, _ = odb.query_rows(k, rp, con, 15, part=(0, ht // 2))\n"
        "    b, _ = odb.query_rows(k, rp, con, 15, part=(ht // 2, ht))\n"
        "    m = pyoracle.merge_rows(a, b)\n"
        "    assert np.array_equal(m, rows) and np.array_equal(pyoracle.result_rows(m), fin)\n"
        "    odb.close()\n"
        "    out.append(fin)\n"
        "km = np.concatenate([synth.kmers_of(g, k) for g in genomes])\n"
        "tg = np.concatenate([np.full(g.size - k + 1, i, dtype=np.uint16) for i, g in enumerate(genomes)])\n"
        "canon, lab = pyoracle.build_discriminative(km, tg, k, ht)\n"
        "base = sys.argv[1]\n"
        "pyoracle.db_write(base, ht, 4, canon, lab)\n"
        "db = pyoracle.OracleDB.load(base, ht, 4)\n"
        "hits = sum(int(db.lookup(k, int(v))[0]) for v in km[:2000])\n"
        "db.close()\n"
        "np.savez(sys.argv[2], fq=out[0], fa=out[1], n=canon.size, hits=hits)\n"
        "print('driven')\n" % (root, os.path.join(root, "tests"), so, K, HT))
    env = dict(os.environ, LD_PRELOAD=":".join(pre), ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    res = str(tmp_path / "res.npz")
    q = subprocess.run([sys.executable, str(script), str(tmp_path / "db"), res], capture_output=True, text=True, timeout=600, env=env)
    assert q.returncode == 0 and "driven" in q.stdout, q.stderr[-1500:]
    assert "Sanitizer" not in q.stderr and "runtime error" not in q.stderr, q.stderr[-1500:]
    # the same drive through the normal build, here
    from oracle import pyoracle
    genomes, sz, ky, lb = small_db(k=K, glen=6000)
    names, seqs = mixed_fasta(genomes, K, n=3000)
    d = np.load(res)
    for key, text in (("fq", synth.fastq_text(names, seqs)), ("fa", synth.fasta_text(names, seqs, width=70))):
        ns, ne, sp, ep, ln = pyoracle.index_reads(text)
        rp, con = pyoracle.pack_reads(text, sp, ep, ln, K)
        fin, _ = pyoracle.OracleDB.from_arrays(HT, sz, ky, lb).classify(K, rp, con, 15)
        assert np.array_equal(fin, d[key]) and (fin[:, 2] > 0).sum() > 1000
    assert int(d["n"]) > 1000 and int(d["hits"]) > 500
