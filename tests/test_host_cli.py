"""The C++ host driver (bin/cuCLARK, bin/cuCLARK-l): command line, database build from
targets.txt (BASELINE config 1 plumbing, CPU), packer, CSV -- against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from jn_cuclark_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def _build():
    if not (os.path.exists(os.path.join(BIN, "cuCLARK")) and os.path.exists(os.path.join(BIN, "cuCLARK-l"))):
        import __graft_entry__
        __graft_entry__.build()


def _write_targets(tmp_path, genomes, labels, n_mask=True):
    lines = []
    for i, g in enumerate(genomes):
        s = bytearray(synth.codes_to_ascii(g))
        if n_mask and i == 0:
            s[500] = ord("N")                 # an ambiguous base resets the k-mer window
        p = tmp_path / ("genome%d.fa" % i)
        p.write_bytes(synth.fasta_text([b"seq%d some description" % i], [bytes(s)], width=70))
        lines.append("%s\t%s\n" % (p, labels[i]))
    t = tmp_path / "targets.txt"
    t.write_text("".join(lines))
    return str(t)


def _expected_db(genomes, labels, k, light, gap=4, n_mask=True):
    """numpy model of the builder: overlapping windows (full) or every gap-th of the
    consecutive non-overlapping windows (light, reference CuCLARK_hh.hh:707-760)."""
    uniq = []
    for l in labels:
        if l not in uniq:
            uniq.append(l)
    km, tg = [], []
    for i, g in enumerate(genomes):
        runs = [g]
        if n_mask and i == 0:
            runs = [g[:500], g[501:]]
        it = 0
        for r in runs:
            if not light:
                x = synth.kmers_of(r, k)
            else:
                starts = np.arange(0, r.size - k + 1, k)
                x = synth.kmers_of(r, k)[starts]
                keep = (it + np.arange(starts.size)) % gap == 0
                it += starts.size
                x = x[keep]
            km.append(x)
            tg.append(np.full(x.size, uniq.index(labels[i]), dtype=np.uint16))
    return synth.discriminative(np.concatenate(km), np.concatenate(tg), k), uniq


def _run(exe, args, check_gpu=True, env=None):
    return subprocess.run([os.path.join(BIN, exe)] + args, capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, **env) if env else None)


@pytest.mark.parametrize("variant", ["light", "full"])
def test_database_build_from_targets_matches_model(oracle, tmp_path, variant):
    """-T targets.txt -> db_central_k*_t*_s*_m*.tsk.{sz,ky,lb}; runs on the CPU, before
    any device is opened (on a machine without GPU the run then stops with an error)."""
    _build()
    light = variant == "light"
    k = 27 if light else 31
    ht = 57777779 if light else 1610612741
    genomes = synth.toy_genomes(3, 4000, seed=41, shared=600)
    labels = ["562", "1280", "562"]            # two files share one label (one target)
    targets = _write_targets(tmp_path, genomes, labels)
    reads = tmp_path / "reads.fa"
    reads.write_bytes(synth.fasta_text([b"r0"], [synth.codes_to_ascii(genomes[0][:150])]))
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    exe = "cuCLARK-l" if light else "cuCLARK"
    r = _run(exe, ["-k", str(k), "-T", targets, "-D", str(dbdir), "-O", str(reads), "-R", str(tmp_path / "out")])
    name = "db_central_k%d_t2_s%d_m0%s.tsk" % (k, ht, "_light_4" if light else "")
    base = str(dbdir / name)
    assert os.path.exists(base + ".sz"), r.stderr
    (canon, lab), uniq = _expected_db(genomes, labels, k, light)
    order = np.lexsort((canon // np.uint64(ht), canon % np.uint64(ht)))
    canon, lab = canon[order], lab[order]
    ky = np.fromfile(base + ".ky", dtype=np.uint32)
    lb = np.fromfile(base + ".lb", dtype=np.uint16)
    assert os.path.getsize(base + ".sz") == ht
    assert np.array_equal(ky, (canon // np.uint64(ht)).astype(np.uint32))
    assert np.array_equal(lb, lab)
    sz = np.fromfile(base + ".sz", dtype=np.uint8)
    nz, cnt = np.unique((canon % np.uint64(ht)).astype(np.int64), return_counts=True)
    assert np.array_equal(np.flatnonzero(sz), nz) and np.array_equal(sz[nz], cnt.astype(np.uint8))
    os.remove(base + ".sz")
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "No HIP devices" in r.stderr      # no silent CPU fallback


def test_the_driver_s_start_and_its_cpu_builder_are_clean_under_the_sanitizers(tmp_path):
    """host/main.cc (-DMC_LIGHT) built with -fsanitize=address,undefined and run up to the point where it opens a device, with the
    card hidden from it: command line, targets file, the CPU database builder (host/dbbuild.hpp: scan, sort, one-target rule, the
    three files, the --tsk text files) -- no report, and the files are byte for byte those of the normal build"""
    _build()
    exe = str(tmp_path / "cuCLARK-l_san")
    lib_dir = os.path.join(ROOT, "jn_cuclark_amd")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-fopenmp", "-pthread", "-DMC_LIGHT",
                        "-o", exe, os.path.join(lib_dir, "host", "main.cc"), "-L" + lib_dir, "-lmcclark", "-lz", "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr[-1500:]
    genomes = synth.toy_genomes(5, 6000, seed=91, shared=700)
    labels = ["A", "B", "B", "C", "D"]                     # two files of one target
    targets = _write_targets(tmp_path, genomes, labels)
    hidden = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = {}
    for tag, prog in (("san", exe), ("plain", os.path.join(BIN, "cuCLARK-l"))):
        d = tmp_path / ("db_" + tag)
        d.mkdir()
        r = subprocess.run([prog, "-T", targets, "-D", str(d) + "/", "-O", str(tmp_path / "genome0.fa"), "-R", str(tmp_path / ("res_" + tag)), "--tsk", "--verbose"],
                           capture_output=True, text=True, timeout=900, env=hidden)
        assert r.returncode != 0 and "No HIP devices" in r.stderr, r.stderr[-800:]          # (the build is done by then)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-1500:]
        out[tag] = {f: open(str(d / f), "rb").read() for f in sorted(os.listdir(str(d)))}
    assert sorted(out["san"]) == sorted(out["plain"]) and len(out["san"]) >= 3 + 4          # .sz/.ky/.lb and one .ht per target
    for f in out["plain"]:
        assert out["san"][f] == out["plain"][f], f


def test_mates_packed_from_two_files_equal_the_joined_records(tmp_path):
    """host/reads.hpp pack_mates (what the streamed plan of -P uses: no joined text) against the packer run on the
    joined records ">id\\nR1NR2" of the sequential join: mates with N, lower case, U, mates shorter than k, empty
    mates, lengths around the 32-base blocks of the vectorised packer -- same offsets, same containers"""
    exe = _input_harness(tmp_path)
    rng = np.random.default_rng(11)
    k, n = 27, 4000
    alpha = np.frombuffer(b"ACGTacgtUNn-", dtype=np.uint8)
    weights = np.array([20, 20, 20, 20, 2, 2, 2, 2, 1, 1, 1, 1], dtype=np.float64)
    weights /= weights.sum()

    def mate(i):
        L = int(rng.choice([0, 5, 26, 27, 28, 31, 32, 33, 63, 64, 65, 100, 150, 151, 250]))
        return alpha[rng.choice(alpha.size, size=L, p=weights)].tobytes()

    m1, m2 = [mate(i) for i in range(n)], [mate(i) for i in range(n)]
    f1, f2 = tmp_path / "m_1.fq", tmp_path / "m_2.fq"
    f1.write_bytes(b"".join(b"@p%d/1\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(m1)))
    f2.write_bytes(b"".join(b"@p%d/2\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(m2)))
    joined = subprocess.run([exe, "pair", str(f1), str(f2)], capture_output=True, timeout=300)
    assert joined.returncode == 0, joined.stderr
    fj = tmp_path / "joined.fa"
    fj.write_bytes(joined.stdout)
    want = subprocess.run([exe, "pack", str(fj), str(k), "1"], capture_output=True, timeout=300)
    got = subprocess.run([exe, "packm", str(f1), str(f2), str(k)], capture_output=True, timeout=300)
    assert want.returncode == 0 and got.returncode == 0, (want.stderr, got.stderr)
    assert got.stdout == want.stdout
    nn, cc = np.frombuffer(got.stdout, dtype=np.uint64, count=2)
    assert int(nn) == n and int(cc) > 10 * n


def test_ranges_of_file_1_are_paired_up_with_file_2(tmp_path):
    """host/pairs.hpp find_mate (the record with the same id around the same relative place of file 2) and align_mates
    (by counting records): for cuts of file 1 at record starts both must return the byte offset of the record with the
    same index in file 2 -- checked against offsets computed here; with mates that drift (short in the first half, long
    in the second) and a 1 KB search window the id search must give up while the count still pairs them up; with one
    record more in file 2 the count must refuse"""
    exe = _input_harness(tmp_path)
    rng = np.random.default_rng(8)
    n = 6000
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]

    def write(path, tag, lens):
        offs, out = [], bytearray()
        for i in range(n):
            offs.append(len(out))
            L = int(lens[i])
            out += b"@q%06d/%s x\n" % (i, tag) + seq[i, :L].tobytes() + b"\n+\n" + b"I" * L + b"\n"
        path.write_bytes(bytes(out))
        return offs, len(out)

    f1, f2, f3 = tmp_path / "a_1.fq", tmp_path / "a_2.fq", tmp_path / "a_3.fq"
    o1, n1 = write(f1, b"1", np.full(n, 150))
    o2, n2 = write(f2, b"2", 60 + rng.integers(0, 80, n))                 # no drift: the id search finds every mate
    o3, n3 = write(f3, b"2", np.where(np.arange(n) < n // 2, 40, 140))    # drift

    def run(fa, fb, window):
        r = subprocess.run([exe, "mates", str(fa), str(fb), "7", str(window)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        rows = {l.split()[0]: [int(v) for v in l.split()[1:]] for l in r.stdout.strip().split("\n")}
        return rows

    rows = run(f1, f2, 262144)
    idx = [o1.index(c) if c < n1 else n for c in rows["cuts"]]          # record index of every cut of file 1
    want = [o2[i] if i < n else n2 for i in idx]
    assert rows["by_id"][0] == 1 and rows["by_id"][1:] == want
    assert rows["by_count"][0] == 1 and rows["by_count"][1:] == want
    rows = run(f1, f3, 1024)
    want3 = [o3[i] if i < n else n3 for i in idx]
    assert rows["by_id"][0] == 0
    assert rows["by_count"][0] == 1 and rows["by_count"][1:] == want3
    f4 = tmp_path / "a_4.fq"
    f4.write_bytes(f2.read_bytes() + b"@lonely/2\nACGT\n+\nIIII\n")
    rows = run(f1, f4, 262144)
    assert rows["by_count"][0] == 0


def test_parallel_mate_join_equals_the_sequential_one(tmp_path):
    """host/pairs.hpp: paired FASTQ files joined on several threads (byte ranges of file 1 at record starts, the
    matching record of file 2 found from record counts) against the sequential join that restates mergePairedFiles
    (src/file.cc:205-268): a regular pair of 5 MB files (quality lines that start with '@' included), the same without
    the final newline, and three pairs the parallel path must hand to the sequential one or fail like it -- a blank
    line in file 1, a changed id, a truncated file 2: same bytes or the same message"""
    exe = _input_harness(tmp_path)
    rng = np.random.default_rng(3)
    n = 45000

    def fq(tag, L):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, L))]
        out = bytearray()
        for i in range(n):
            q = (b"@" + b"I" * (L - 1)) if i % 3 == 0 else b"I" * L
            out += b"@r%07d/%s\n" % (i, tag) + seq[i].tobytes() + b"\n+\n" + q + b"\n"
        return bytes(out)

    a, b = fq(b"1", 100), fq(b"2", 80)
    cut = a.index(b"@r0030000/1")
    cases = {
        "regular": (a, b),
        "no_final_newline": (a, b[:-1]),
        "blank_line": (a[:cut] + b"\n" + a[cut:], b),
        "changed_id": (a, b.replace(b"@r0020000/2", b"@r0020001/2", 1)),
        "truncated": (a, b[:len(b) // 2 + 7]),
    }
    for name, (x, y) in cases.items():
        f1, f2 = tmp_path / (name + "_1.fq"), tmp_path / (name + "_2.fq")
        f1.write_bytes(x)
        f2.write_bytes(y)
        seq = subprocess.run([exe, "pair", str(f1), str(f2)], capture_output=True, timeout=300)
        par = subprocess.run([exe, "pairp", str(f1), str(f2), "4"], capture_output=True, timeout=300)
        assert par.returncode == seq.returncode, name
        assert par.stdout == seq.stdout, name
        assert par.stderr == seq.stderr, name
        if name in ("regular", "no_final_newline"):
            assert seq.returncode == 0 and seq.stdout.count(b">") == n and seq.stdout.startswith(b">r0000000\n")


def test_tsk_writes_the_reference_s_target_specific_kmer_files(tmp_path):
    """--tsk (createTargetFilesNames, src/CuCLARK_hh.hh:342-378; SaveMultiple, src/HashTableStorage_hh.hh:282-327): one
    text file per target, "<value>\\t<count>\\t<k-mer>" for every k-mer seen in that target only, in the reference
    table's iteration order (buckets ascending; inside a bucket the order of first insertion -- the fixture plants
    three k-mers of one bucket met in the order 2, 0, 1).  Fixtures: what the REFERENCE's host driver wrote for the
    same three genomes (tests/golden/make_golden.py tsk_golden), plus the sha256 of its .sz/.ky/.lb.  The builder runs
    before any device is opened, so this needs no GPU (without one the run then stops with "No HIP devices")."""
    import hashlib
    _build()
    gold = os.path.join(ROOT, "tests", "golden", "tsk")
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    r = subprocess.run([os.path.join(BIN, "cuCLARK"), "-k", "31", "-T", os.path.join(gold, "targets.txt"), "-D", str(dbdir) + "/",
                        "-O", os.path.join(gold, "g0.fa"), "-R", str(tmp_path / "res"), "--tsk", "--verbose"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert "Creation of targets specific k-mers files requested" in r.stderr
    for name in ("T0_k31.ht", "T1_k31.ht", "S9_k31.ht"):
        assert open(str(dbdir / name), "rb").read() == open(os.path.join(gold, name), "rb").read(), name
    want = dict(l.split()[::-1] for l in open(os.path.join(gold, "db_sha256.txt")))
    base = str(dbdir / "db_central_k31_t3_s1610612741_m0.tsk")
    for ext, digest in want.items():
        assert hashlib.sha256(open(base + ext, "rb").read()).hexdigest() == digest, ext


def test_tsk_recovery_rebuilds_the_database_from_the_ht_files(tmp_path):
    """The database files are gone, the per-target .ht files of an earlier --tsk run are still there
    (src/CuCLARK_hh.hh:633-684): with --tsk the database is put back together from them -- byte-identical to what the
    REFERENCE's builder wrote for these genomes (tests/golden/tsk/db_sha256.txt) -- and the program leaves with
    exit(-1), as the reference does; without --tsk it says "Failed to find the database." (the reference does not
    rebuild from the target files when the .ht files exist, getTargetsData :1826-1836).  Before any device is
    opened: no GPU needed.  (tests/test_ref_host.py runs the reference's own recovery next to this one.)"""
    import hashlib
    import shutil
    _build()
    gold = os.path.join(ROOT, "tests", "golden", "tsk")
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    for name in ("T0_k31.ht", "T1_k31.ht", "S9_k31.ht"):
        shutil.copy(os.path.join(gold, name), str(dbdir / name))
    args = ["-k", "31", "-T", os.path.join(gold, "targets.txt"), "-D", str(dbdir) + "/", "-O", os.path.join(gold, "g0.fa"),
            "-R", str(tmp_path / "res")]
    base = str(dbdir / "db_central_k31_t3_s1610612741_m0.tsk")
    r = subprocess.run([os.path.join(BIN, "cuCLARK")] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 255 and "Failed to find the database." in r.stderr and not os.path.exists(base + ".ky")
    r = subprocess.run([os.path.join(BIN, "cuCLARK")] + args + ["--tsk"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 255, r.stderr
    for msg in ("The database will be recovered from saved targets-specific data.", "Dataset 3 loaded.",
                "31-mers finally loaded. Creating database in disk...", "Central Hashtable successfully stored in disk."):
        assert msg in r.stderr, (msg, r.stderr)
    want = dict(l.split()[::-1] for l in open(os.path.join(gold, "db_sha256.txt")))
    for ext, digest in want.items():
        assert hashlib.sha256(open(base + ext, "rb").read()).hexdigest() == digest, ext
    n_kmers = sum(1 for name in ("T0_k31.ht", "T1_k31.ht", "S9_k31.ht") for ln in open(os.path.join(gold, name)) if not ln.startswith("#"))
    assert ("%d 31-mers finally loaded." % n_kmers) in r.stderr and os.path.getsize(base + ".ky") == 4 * n_kmers


def test_cli_errors_match_reference_messages(tmp_path):
    _build()
    r = _run("cuCLARK", ["-k", "40", "-T", "x", "-D", "y", "-O", "z", "-R", "w"])
    assert r.returncode == 1 and "The k-mer length should be in [2,32]." in r.stderr
    r = _run("cuCLARK", ["-k", "31", "-T", str(tmp_path / "missing"), "-D", "y", "-O", "z", "-R", "w"])
    assert r.returncode == 1 and "Failed to find/read the file of the targets definition" in r.stderr
    r = _run("cuCLARK", ["--version"])
    assert r.returncode == 0 and "Version: 1.1" in r.stdout
    r = _run("cuCLARK", ["-k", "31", "--bogus", "1", "2", "3", "4"])
    assert r.returncode == 1 and "Failed to recognize option: --bogus" in r.stderr


def _expected_csv(oracle, text, k, ht, base, names, paired=False, extended=False, maxhits=23):
    ns, ne, sp, ep, ln = oracle.index_reads(text)
    rp, con = oracle.pack_reads(text, sp, ep, ln, k)
    odb = oracle.OracleDB.load(base, ht, 4)
    rows, _ = odb.query_rows(k, rp, con, maxhits)
    res = oracle.result_rows(rows)
    out = ["Object_ID" + ("," + ",".join(names[1:]) if extended else "") + ",Gamma,Assignment,Score,Confidence\n"]
    for i in range(ln.size):
        line = oracle.csv_line(text[int(ns[i]):int(ne[i])], res[i], int(ln[i]) - (1 if paired else 0), k, names[int(res[i, 1])])
        if extended:
            dense = np.zeros(len(names) - 1, dtype=np.int64)
            for j in range(int(rows[i, 0])):
                dense[int(rows[i, 1 + 2 * j])] = int(rows[i, 2 + 2 * j])
            nm, rest = line.split(",", 1)
            line = nm + "".join(",%d" % v for v in dense) + "," + rest
        out.append(line)
    return "".join(out), (rp, con)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fasta", "fastq_batches", "paired", "extended", "fastq_gz", "paired_gz"])
def test_end_to_end_csv_is_byte_identical_to_oracle(oracle, tmp_path, mode):
    """file -> CSV through bin/cuCLARK-l; the *_gz modes feed gzip files directly (inflated in memory; the
    reference's wrapper script gunzips a copy first), paired mates are joined in memory"""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    _build()
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(4, 5000, seed=51, shared=500)
    labels = ["Ecoli", "Saureus", "Bsub", "Paer"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    names, seqs = mixed_fasta(genomes, k, seed=13, n=1500)
    args = ["-T", targets, "-D", str(dbdir), "-R", str(tmp_path / "res"), "--dump-batches", str(tmp_path / "dump.bin")]
    paired = mode in ("paired", "paired_gz")
    if paired:
        m1 = [s[:100].replace(b"\n", b"") for s in seqs]
        m2 = [s[50:150] for s in seqs]
        nm = [n.split(b" ")[0] for n in names]
        f1, f2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
        t1, t2 = synth.fastq_text([n + b"/1" for n in nm], m1), synth.fastq_text([n + b"/2" for n in nm], m2)
        if mode == "paired_gz":
            f1, f2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
            from helpers import bgzf_compress
            f1.write_bytes(bgzf_compress(t1, block=5000))          # BGZF: inflated by -n / 2 threads, next to file 2
            args += ["-n", "4"]
            # two concatenated gzip members (what `cat a.gz b.gz` or bgzip produce)
            cut = t2.index(b"\n@", len(t2) // 2) + 1
            f2.write_bytes(gzip.compress(t2[:cut]) + gzip.compress(t2[cut:]))
        else:
            f1.write_bytes(t1)
            f2.write_bytes(t2)
        args += ["-P", str(f1), str(f2)]
        text = synth.fasta_text(nm, [a + b"N" + b for a, b in zip(m1, m2)])
    elif mode in ("fastq_batches", "fastq_gz"):
        text = synth.fastq_text(names, seqs)
        p = tmp_path / ("reads.fq.gz" if mode == "fastq_gz" else "reads.fq")
        p.write_bytes(gzip.compress(text) if mode == "fastq_gz" else text)
        args += ["-O", str(p), "-n", "4", "-b", "7"]
    else:
        text = synth.fasta_text(names, seqs, width=60)
        p = tmp_path / "reads.fa"
        p.write_bytes(text)
        args += ["-O", str(p)] + (["--extended"] if mode == "extended" else [])
    r = _run("cuCLARK-l", args)
    assert r.returncode == 0, r.stderr
    base = str(dbdir / ("db_central_k27_t4_s%d_m0_light_4.tsk" % ht))
    want, (rp, con) = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels, paired=paired,
                                    extended=mode == "extended")
    got = open(str(tmp_path / "res.csv")).read()
    assert got == want
    # the packed batches equal the oracle's restatement of the reference packer
    raw = open(str(tmp_path / "dump.bin"), "rb").read()
    off, ptrs, cons = 0, [], []
    while off < len(raw):
        n, c = np.frombuffer(raw, dtype=np.uint64, count=2, offset=off)
        off += 16
        ptrs.append(np.frombuffer(raw, dtype=np.uint32, count=int(n) + 1, offset=off)); off += 4 * (int(n) + 1)
        cons.append(np.frombuffer(raw, dtype=np.uint16, count=int(c), offset=off)); off += 2 * int(c)
    assert np.array_equal(np.concatenate(cons), con)
    starts = np.concatenate([[0], np.cumsum([c.size for c in cons])[:-1]])
    flat = np.concatenate([p[:-1].astype(np.int64) + s for p, s in zip(ptrs, starts)] + [[con.size]])
    assert np.array_equal(flat, rp.astype(np.int64))
    assert "Done in" in r.stderr and "reads/min" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fastq", "fastq_host", "fastq_hybrid", "fasta_70", "paired", "paired_drift", "paired_extra_mate", "paired_changed_id",
                                  "extended", "gives_up", "gives_up_host", "gives_up_hybrid", "odd_record", "odd_record_hybrid"])
def test_streamed_ingest_equals_the_oracle(oracle, tmp_path, mode):
    """Large files are cut into byte ranges at record starts and every range is indexed, packed and submitted by
    one task (host/main.cc classify_image, streamed plan; MC_STREAM_MIN_BYTES lowers the size it starts at).  The
    CSV must not depend on it: FASTQ, multi-line FASTA, joined mates and the extended table against the oracle
    with 9 ranges on 4 threads; `gives_up`: long reads first, then ten times as many short ones -- the ranges at
    the end hold more reads than the buffers guessed from the head take, and the run starts over with the plan
    that indexes the whole file first.  `paired`: the mates are classified straight from their two files (byte ranges
    of file 1, the matching records of file 2 found by id: no joined text); `paired_extra_mate`: file 2 holds one
    record more -- the ranges do not pair up and the mates are joined first, as for small files;
    `paired_changed_id`: the reference's message and exit status; `paired_drift`: the mates of the first half are short,
    those of the second long, so a record's mate is NOT at the same relative place of file 2 (search window cut to 2 KB
    for the test) and the ranges are paired up by counting records instead.
    Round 4: plain FASTQ goes to the card as TEXT (mc_text_*: records cut, packed and classified there; `fastq`, `gives_up`);
    `fastq_host` / `gives_up_host` (MC_GPU_INGEST=0) keep the host's indexer and packer under test; `odd_record`: a header line
    the card does not vouch for (a name that starts with a blank) -- the batch comes back unclassified and the host does the
    file; FASTA, mates and the extended table never leave the host path.  `*_hybrid` (MC_CARD_SHARE=50): every second range goes
    up as text, the others are indexed and packed on the host -- two rings of buffers, one CSV; a range either way may be the one
    that makes the run start over."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    _build()
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(4, 5000, seed=52, shared=500)
    labels = ["Ecoli", "Saureus", "Bsub", "Paer"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    names, seqs = mixed_fasta(genomes, k, seed=17, n=3000)
    args = ["-T", targets, "-D", str(dbdir), "-R", str(tmp_path / "res"), "-n", "4", "-b", "9", "--verbose"]
    paired = mode.startswith("paired")
    if paired:
        m1 = [s[:100].replace(b"\n", b"") for s in seqs]
        m2 = [s[50:150] for s in seqs]
        if mode == "paired_drift":
            m2 = [s[50:90] if i < len(seqs) // 2 else s[20:150] for i, s in enumerate(seqs)]
        nm = [n.split(b" ")[0] for n in names]
        f1, f2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
        f1.write_bytes(synth.fastq_text([n + b"/1" for n in nm], m1))
        t2 = synth.fastq_text([n + b"/2" for n in nm], m2)
        if mode == "paired_extra_mate":
            t2 += b"@lonely/2\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
        if mode == "paired_changed_id":
            t2 = t2.replace(nm[2000] + b"/2", nm[2000] + b"x/2", 1)
        f2.write_bytes(t2)
        args += ["-P", str(f1), str(f2)]
        text = synth.fasta_text(nm, [a + b"N" + b for a, b in zip(m1, m2)])
    elif mode == "fasta_70":
        text = synth.fasta_text(names, seqs, width=70)
        p = tmp_path / "reads.fa"
        p.write_bytes(text)
        args += ["-O", str(p)]
    else:
        if mode.startswith("gives_up"):
            rng = np.random.default_rng(5)
            long_ones = [synth.codes_to_ascii(genomes[i % 4][:4000]) for i in range(300)]
            short_ones = [synth.codes_to_ascii(genomes[i % 4][int(s):int(s) + 40]) for i, s in enumerate(rng.integers(0, 4900, 30000))]
            seqs = long_ones + short_ones
            names = [b"r%d" % i for i in range(len(seqs))]
        text = synth.fastq_text(names, seqs)
        if mode.startswith("odd_record"):
            text = text.replace(b"@" + names[1500], b"@ " + names[1500], 1)
        p = tmp_path / "reads.fq"
        p.write_bytes(text)
        args += ["-O", str(p)] + (["--extended"] if mode == "extended" else [])
    hybrid = mode.endswith("_hybrid")
    r = _run("cuCLARK-l", args, env={"MC_STREAM_MIN_BYTES": "1", "MC_MATE_WINDOW": "2048" if mode == "paired_drift" else "262144",
                                     "MC_GPU_INGEST": "0" if mode.endswith("_host") else "1", "MC_CARD_SHARE": "50" if hybrid else "100"})
    if mode == "paired_changed_id":
        assert r.returncode != 0 and "Error: read id does not match between files!" in r.stderr, r.stderr
        return
    assert r.returncode == 0, r.stderr
    if hybrid:
        # (which range meets the trouble first -- one on the card, one on the host -- is a matter of timing; the odd record may sit in
        #  a range the host indexes, and then nothing is given up at all)
        gave_up = ("streamed ingest given up" in r.stderr) or ("ingest on the card given up" in r.stderr)
        assert gave_up == (mode == "gives_up_hybrid") or mode == "odd_record_hybrid", r.stderr
        assert ("50 % of them as text to the card" in r.stderr) == (not gave_up), r.stderr
    else:
        assert ("streamed ingest given up" in r.stderr) == (mode == "gives_up_host"), r.stderr
        assert ("ingest on the card given up" in r.stderr) == (mode in ("gives_up", "odd_record")), r.stderr
        assert ("classified on the card" in r.stderr) == (mode == "fastq"), r.stderr
        assert ("timing: streamed" in r.stderr) == (not mode.startswith("gives_up") and mode != "odd_record"), r.stderr
    assert ("streamed ingest of the two files given up" in r.stderr) == (mode == "paired_extra_mate"), r.stderr
    assert ("byte ranges of both files" in r.stderr) == (mode in ("paired", "paired_drift")), r.stderr
    assert ("mates located by counting records" in r.stderr) == (mode == "paired_drift"), r.stderr
    base = str(dbdir / ("db_central_k27_t4_s%d_m0_light_4.tsk" % ht))
    want, _ = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels, paired=paired, extended=mode == "extended")
    assert open(str(tmp_path / "res.csv")).read() == want
    assert "%d reads)" % len(seqs) in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fastq", "fastq_card", "fasta_70", "extended", "gives_up", "contig", "members", "bgzf", "truncated"])
def test_gzip_files_are_classified_in_segments(oracle, tmp_path, mode):
    """A gzip file is inflated by a thread of its own into segments of whole records and classified segment by
    segment, the lines of a segment written behind those of the one before (host/gzstream.hpp, host/main.cc
    run_gz_segments; MC_GZ_SEGMENT_BYTES cuts the 256 MB of a segment down to 60 KB here, so a 0.5-1.3 MB text is a
    dozen segments or more).  The CSV must be the one the oracle gives for the whole text: FASTQ through the host's
    indexer and through the card's (`fastq_card`), multi-line FASTA, the extended table (its MIN/MAX/AVG line adds up
    over the segments); `gives_up`: 400 KB segments, long reads first, then many short ones -- inside the segment that holds
    both the streamed plan gives up and that segment alone is done again by the plan that indexes it whole; `contig`: a record of 300 KB, five
    segments long, makes its segment grow; `members`: three concatenated gzip members (cat a.gz b.gz, bgzip);
    `bgzf`: bgzip's 64 KB blocks, inflated by the four -n threads at once;
    `truncated`: the file stops inside a member -- the reads of the segments before that are in the CSV, the run says
    so and fails (the whole-file path, which inflates everything first, classifies nothing: MC_GZ_WHOLE=1)."""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta, bgzf_compress
    _build()
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(4, 5000, seed=53, shared=500)
    labels = ["Ecoli", "Saureus", "Bsub", "Paer"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    names, seqs = mixed_fasta(genomes, k, seed=19, n=3000)
    args = ["-T", targets, "-D", str(dbdir), "-R", str(tmp_path / "res"), "-n", "4", "-b", "5", "--verbose"]
    if mode == "gives_up":
        rng = np.random.default_rng(6)
        long_ones = [synth.codes_to_ascii(genomes[i % 4][:4000]) for i in range(40)]          # 320 KB of the first 400 KB segment
        short_ones = [synth.codes_to_ascii(genomes[i % 4][int(s):int(s) + 40]) for i, s in enumerate(rng.integers(0, 4900, 12000))]
        seqs = long_ones + short_ones + long_ones
        names = [b"r%d" % i for i in range(len(seqs))]
    if mode == "contig":
        big = np.concatenate([genomes[i % 4] for i in range(60)])
        seqs = seqs[:1000] + [synth.codes_to_ascii(big)] + seqs[1000:]
        names = names[:1000] + [b"contig_of_300k"] + names[1000:]
    if mode in ("fasta_70", "contig"):
        text = synth.fasta_text(names, seqs, width=70)
    else:
        text = synth.fastq_text(names, seqs)
    if mode == "members":
        a, b = len(text) // 3, 2 * len(text) // 3            # member ends fall inside records
        z = gzip.compress(text[:a]) + gzip.compress(text[a:b]) + gzip.compress(text[b:])
    elif mode == "bgzf":
        z = bgzf_compress(text, block=20000)
        assert gzip.decompress(z) == text
    else:
        z = gzip.compress(text)
    if mode == "truncated":
        z = z[:len(z) * 2 // 3]
    p = tmp_path / "reads.gz"
    p.write_bytes(z)
    args += ["-O", str(p)] + (["--extended"] if mode == "extended" else [])
    env = {"MC_STREAM_MIN_BYTES": "1", "MC_GZ_SEGMENT_BYTES": "400000" if mode == "gives_up" else "60000",
           "MC_GPU_INGEST": "1" if mode == "fastq_card" else "0"}
    r = _run("cuCLARK-l", args, env=env)
    base = str(dbdir / ("db_central_k27_t4_s%d_m0_light_4.tsk" % ht))
    want, _ = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels, paired=False, extended=mode == "extended")
    got = open(str(tmp_path / "res.csv")).read()
    if mode == "truncated":
        assert r.returncode != 0 and "zlib: truncated gzip input" in r.stderr and "reads only" in r.stderr, r.stderr
        n_lines = got.count("\n")
        assert 500 < n_lines < len(seqs) and want.startswith(got)
        os.remove(str(tmp_path / "res.csv"))
        r = _run("cuCLARK-l", args, env=dict(env, MC_GZ_WHOLE="1"))
        assert "zlib: truncated gzip input" in r.stderr and not os.path.exists(str(tmp_path / "res.csv")), r.stderr
        return
    assert r.returncode == 0, r.stderr
    n_seg = int(r.stderr.split("gzip input classified in ")[1].split()[0])
    assert n_seg >= (3 if mode == "gives_up" else 5 if mode == "contig" else 8), r.stderr
    assert got == want
    assert "%d reads)" % len(seqs) in r.stderr and r.stderr.count("Done.") == 1
    # ("Writing results..." comes with the segment that opens the CSV: once, or twice when that very segment is done again;
    #  the segment that holds the contig may give the streamed plan up too -- one range with the contig, the others with short reads)
    assert r.stderr.count("Writing") == 1 or (mode == "gives_up" and r.stderr.count("Writing") == 2), r.stderr
    assert ("streamed ingest given up" in r.stderr) == (mode == "gives_up") or mode == "contig", r.stderr
    assert ("classified on the card" in r.stderr) == (mode == "fastq_card"), r.stderr
    assert ("BGZF: " in r.stderr) == (mode == "bgzf"), r.stderr
    if mode == "bgzf":
        assert int(r.stderr.split("BGZF: ")[1].split()[0]) == (len(text) + 19999) // 20000 + 1
    if mode == "contig":
        assert int(r.stderr.split("segment(s) of at most ")[1].split()[0]) > 300000
    if mode == "extended":
        # the closing statistics are those of the whole file: the run on the inflated text prints the same line
        q = tmp_path / "reads.fq"
        q.write_bytes(text)
        r2 = _run("cuCLARK-l", args[:-3] + ["-O", str(q), "--extended"], env=env)
        line = [l for l in r.stderr.split("\n") if l.startswith("MIN targets")]
        assert r2.returncode == 0 and len(line) == 1 and line == [l for l in r2.stderr.split("\n") if l.startswith("MIN targets")]
        assert open(str(tmp_path / "res.csv")).read() == want


STAND_IN_SCRIPT = """#!/bin/sh
# stand-in for the exec line of the reference's scripts/classify_metagenome.sh (:84-87 prepend the contents of
# .settings, :155-159 exec ../bin/cuCLARK or, with --light, ../bin/cuCLARK-l with the caller's arguments)
PARAMS="$(tr '\\n' ' ' < ./.settings)"; EXE=cuCLARK
for a in "$@"; do case "$a" in --light) EXE=cuCLARK-l;; *) PARAMS="$PARAMS $a";; esac; done
exec ../bin/$EXE $PARAMS
"""


@pytest.mark.gpu
@pytest.mark.parametrize("front", ["script", "kent_script", "kent_direct"])
def test_runs_under_the_classify_script_and_kent(oracle, tmp_path, front):
    """drop-in under scripts/classify_metagenome.sh (reference :155-159 execs ../bin/cuCLARK[-l] with .settings +
    its own arguments) and under `kent -c` (app/kent.cpp:452-553: cd scripts && ./classify_metagenome.sh ...
    --light, results under ./results/).  Where /root/reference exists the reference's own script is used (from a
    scratch copy: it needs a writable .settings next to it); on the GPU box a stand-in for its exec line.
    kent_direct: our kent without any script (-T/-D)."""
    import shutil
    _build()
    ref_script = "/root/reference/scripts/classify_metagenome.sh"
    (tmp_path / "scripts").mkdir()
    (tmp_path / "bin").mkdir()
    (tmp_path / "results").mkdir()
    if front != "kent_direct":
        dst = tmp_path / "scripts" / "classify_metagenome.sh"
        if os.path.exists(ref_script):
            shutil.copy(ref_script, str(dst))
        else:
            dst.write_text(STAND_IN_SCRIPT)
        os.chmod(str(dst), 0o755)
    for exe in ("cuCLARK", "cuCLARK-l", "kent", "getAbundance"):
        os.symlink(os.path.join(BIN, exe), str(tmp_path / "bin" / exe))
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(3, 3000, seed=81)
    labels = ["A", "B", "C"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "custom_0"
    dbdir.mkdir()
    (tmp_path / "scripts" / ".settings").write_text("-T %s\n-D %s/\n" % (targets, dbdir))
    names = [b"r%d" % i for i in range(200)]
    seqs = [synth.codes_to_ascii(genomes[i % 3][10 * i:10 * i + 150]) for i in range(200)]
    text = synth.fasta_text(names, seqs)
    reads = tmp_path / "reads.fa"
    reads.write_bytes(text)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "jn_cuclark_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    if front == "script":
        r = subprocess.run(["sh", "./classify_metagenome.sh", "-O", str(reads), "-R", str(tmp_path / "results" / "out"),
                            "-n", "2", "-b", "4", "--light"], cwd=str(tmp_path / "scripts"),
                           capture_output=True, text=True, timeout=600, env=env)
    elif front == "kent_script":
        r = subprocess.run([str(tmp_path / "bin" / "kent"), "-c", "-O", "reads.fa", "-R", "out", "-n", "2", "-b", "4"],
                           cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
    else:
        os.remove(str(tmp_path / "scripts" / ".settings"))
        r = subprocess.run([os.path.join(BIN, "kent"), "-c", "-O", "reads.fa", "-R", "out", "-n", "2", "-b", "4",
                            "-T", targets, "-D", str(dbdir)], cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    base = str(dbdir / ("db_central_k27_t3_s%d_m0_light_4.tsk" % ht))
    want, _ = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels)
    assert open(str(tmp_path / "results" / "out.csv")).read() == want
    os.remove(base + ".sz")


@pytest.mark.gpu
def test_k32_full_table_end_to_end(oracle, tmp_path):
    """k = 32: 8-byte keys on disk (reference T64, main.cc:277-286) and in HBM"""
    _build()
    k, ht = 32, 1610612741
    genomes = synth.toy_genomes(3, 5000, seed=97, shared=300)
    labels = ["A", "B", "C"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    names, seqs = mixed_fasta(genomes, k, seed=3, n=600)
    text = synth.fasta_text(names, seqs)
    p = tmp_path / "reads.fa"
    p.write_bytes(text)
    r = _run("cuCLARK", ["-k", "32", "-T", targets, "-D", str(dbdir), "-O", str(p), "-R", str(tmp_path / "res")])
    assert r.returncode == 0, r.stderr
    base = str(dbdir / ("db_central_k32_t3_s%d_m0.tsk" % ht))
    n = os.path.getsize(base + ".lb") // 2
    assert os.path.getsize(base + ".ky") == 8 * n
    ns, ne, sp, ep, ln = oracle.index_reads(text)
    rp, con = oracle.pack_reads(text, sp, ep, ln, k)
    odb = oracle.OracleDB.load(base, ht, 8)
    res, _ = odb.classify(k, rp, con, 15)
    want = "Object_ID,Gamma,Assignment,Score,Confidence\n" + "".join(
        oracle.csv_line(text[int(ns[i]):int(ne[i])], res[i], int(ln[i]), k, (["NA"] + labels)[int(res[i, 1])])
        for i in range(ln.size))
    assert open(str(tmp_path / "res.csv")).read() == want
    assert (res[:, 2] > 0).sum() > 300
    os.remove(base + ".sz")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["light", "full"])
def test_gpu_database_build_is_byte_identical_to_cpu_build(tmp_path, variant):
    """--gpu-build (include/mc_build.h: count -> scan -> scatter -> per-bucket sort ->
    one-target rule -> compact, on the device) writes the same three files as the CPU
    builder, which tests/test_ref_host.py pins against the reference's own builder."""
    import filecmp
    _build()
    light = variant == "light"
    k = 27 if light else 31
    ht = 57777779 if light else 1610612741
    genomes = synth.toy_genomes(6, 60000, seed=33, shared=3000)
    genomes[4] = genomes[4].copy()
    genomes[4][10000:20000] = genomes[4][30000:40000]          # repeats inside one target
    genomes[5] = genomes[5].copy()
    genomes[5][:5000] = 3                                      # poly-A: one k-mer thousands of times
    labels = ["a", "b", "c", "a", "d", "e"]
    targets = _write_targets(tmp_path, genomes, labels)
    reads = tmp_path / "reads.fa"
    reads.write_bytes(synth.fasta_text([b"r0"], [synth.codes_to_ascii(genomes[0][:150])]))
    exe = "cuCLARK-l" if light else "cuCLARK"
    name = "db_central_k%d_t5_s%d_m0%s.tsk" % (k, ht, "_light_4" if light else "")
    dirs = {}
    for tag, extra in (("cpu", []), ("gpu", ["--gpu-build", "-n", "4"])):
        d = tmp_path / tag
        d.mkdir()
        r = _run(exe, ["-k", str(k), "-T", targets, "-D", str(d), "-O", str(reads), "-R", str(tmp_path / ("o" + tag))] + extra)
        assert r.returncode == 0, r.stderr
        dirs[tag] = d
    for ext in (".sz", ".ky", ".lb"):
        a, b = str(dirs["cpu"] / (name + ext)), str(dirs["gpu"] / (name + ext))
        assert os.path.getsize(a) > 0 and filecmp.cmp(a, b, shallow=False), ext
    assert open(str(tmp_path / "ocpu.csv")).read() == open(str(tmp_path / "ogpu.csv")).read()
    for d in dirs.values():
        os.remove(str(d / (name + ".sz")))


def _input_harness(tmp_path):
    exe = str(tmp_path / "host_input")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fopenmp", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "harness", "host_input.cc"), "-lz"],
                   check=True)
    return exe


@pytest.mark.parametrize("fmt", ["fastq", "fasta_wide", "fasta_70"])
def test_indexer_and_vectorised_packer_equal_the_oracle(oracle, tmp_path, fmt):
    """host/reads.hpp + host/simd.hpp on the CPU: parallel indexer (byte ranges cut at record starts) and the
    packer whose runs of bases go through 32-base AVX2 blocks -- ragged reads, lower case, U, N runs, reads
    and parts shorter than k, long reads; against the oracle's restatement of the reference packer
    (src/CuCLARK_hh.hh:1629-1707)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    exe = _input_harness(tmp_path)
    k = 27
    genomes = synth.toy_genomes(4, 9000, seed=77)
    names, seqs = mixed_fasta(genomes, k, seed=31, n=6000)
    rng = np.random.default_rng(3)
    for i in range(0, 6000, 97):                       # long reads: many whole blocks, odd tails
        L = int(rng.integers(300, 4000))
        s = bytearray(synth.codes_to_ascii(genomes[i % 4][:L]))
        if i % 2:
            s[int(rng.integers(0, L))] = ord("n")
        seqs[i] = bytes(s)
    text = (synth.fastq_text(names, seqs) if fmt == "fastq" else
            synth.fasta_text(names, seqs, width=0 if fmt == "fasta_wide" else 70))
    f = tmp_path / "reads.txt"
    f.write_bytes(text)
    ns, ne, sp, ep, ln = oracle.index_reads(text)
    rp, con = oracle.pack_reads(text, sp, ep, ln, k)
    for threads in ("1", "5"):
        r = subprocess.run([exe, "pack", str(f), str(k), threads], capture_output=True)
        assert r.returncode == 0, r.stderr
        n, c = np.frombuffer(r.stdout, dtype=np.uint64, count=2)
        got_rp = np.frombuffer(r.stdout, dtype=np.uint32, count=int(n) + 1, offset=16)
        got_con = np.frombuffer(r.stdout, dtype=np.uint16, count=int(c), offset=16 + 4 * (int(n) + 1))
        assert int(n) == ln.size and np.array_equal(got_rp, rp) and np.array_equal(got_con, con)


def test_one_sweep_fastq_indexer_equals_the_line_by_line_indexer(tmp_path):
    """host/reads.hpp index_fastq_avx2 / index_fasta_avx2 (newline bitmaps of 64-byte blocks, one state machine) against the
    line-by-line indexer it replaces for FASTQ, which the test above pins to the oracle's restatement of
    src/CuCLARK_hh.hh:1476-1533: 60 000 hostile texts (soups of '@', newlines and blanks, truncated records,
    overwritten separators), every index column equal"""
    exe = _input_harness(tmp_path)
    r = subprocess.run([exe, "indexfuzz", "60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() in ("60000", "no avx2")
    # the same for FASTA (index_fasta_avx2): multi-line records, '>' inside sequences, blank lines, empty names
    r = subprocess.run([exe, "indexfuzz_fasta", "60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() in ("60000", "no avx2")


def test_input_images_gzip_and_pairing(tmp_path):
    """host/input.hpp on the CPU: gzip members are inflated in memory, mates are joined as the
    reference's mergePairedFiles does (src/file.cc:205-268), errors keep the reference's wording"""
    import gzip
    exe = _input_harness(tmp_path)
    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=int(n))) for n in rng.integers(30, 260, size=400)]
    names = [b"read%d extra words" % i for i in range(len(seqs))]
    fq = synth.fastq_text(names, seqs)
    plain, gz, gz2, bad = (tmp_path / n for n in ("a.fq", "a.fq.gz", "a2.fq.gz", "bad.fq.gz"))
    plain.write_bytes(fq)
    gz.write_bytes(gzip.compress(fq))
    cut = len(fq) // 3
    gz2.write_bytes(gzip.compress(fq[:cut]) + gzip.compress(fq[cut:]))          # concatenated members
    bad.write_bytes(gzip.compress(fq)[:-200])                                    # truncated
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import bgzf_compress
    bg, bg2, bg3 = (tmp_path / n for n in ("b.fq.gz", "b2.fq.gz", "b3.fq.gz"))
    bg.write_bytes(bgzf_compress(fq, block=3000))
    bg2.write_bytes(bgzf_compress(fq[:cut], block=3000, eof_block=False) + gzip.compress(fq[cut:]))      # blocks, then a plain member
    bg3.write_bytes(bgzf_compress(fq, block=3000) + b"\0\0xyz")                                         # trailing bytes
    for f in (plain, gz, gz2, bg, bg2, bg3):
        for threads in ("1", "4"):                       # 4: BGZF blocks are inflated side by side
            r = subprocess.run([exe, "load", str(f), threads], capture_output=True)
            assert r.returncode == 0 and r.stdout == fq, (f, threads)
    dam = bytearray(bgzf_compress(fq, block=3000))
    dam[len(dam) // 2] ^= 0x41
    bad.write_bytes(bytes(dam))
    r = subprocess.run([exe, "load", str(bad), "4"], capture_output=True)
    assert r.returncode == 2 and b"gzip" in r.stderr
    bad.write_bytes(gzip.compress(fq)[:-200])
    r = subprocess.run([exe, "load", str(bad)], capture_output=True)
    assert r.returncode == 2 and b"gzip" in r.stderr
    r = subprocess.run([exe, "load", str(tmp_path / "missing.fq")], capture_output=True)
    assert r.returncode == 2 and b"Failed to open" in r.stderr

    # mates: ids are cut at ' ', '/', '\t', '@' and must agree
    ids = [b"r%d" % i for i in range(len(seqs))]
    m1 = synth.fastq_text([i + b"/1 lane=3" for i in ids], [s[:100] for s in seqs])
    m2 = synth.fastq_text([i + b"/2" for i in ids], [s[-80:] for s in seqs])
    (tmp_path / "m1.fq").write_bytes(m1)
    (tmp_path / "m2.fq.gz").write_bytes(gzip.compress(m2))
    r = subprocess.run([exe, "pair", str(tmp_path / "m1.fq"), str(tmp_path / "m2.fq.gz")], capture_output=True)
    want = b"".join(b">" + i + b"\n" + s[:100] + b"N" + s[-80:] + b"\n" for i, s in zip(ids, seqs))
    assert r.returncode == 0 and r.stdout == want
    (tmp_path / "m3.fq").write_bytes(m2.replace(b"@r7/", b"@r700/", 1))
    r = subprocess.run([exe, "pair", str(tmp_path / "m1.fq"), str(tmp_path / "m3.fq")], capture_output=True)
    assert r.returncode == 2 and b"read id does not match between files" in r.stderr
    (tmp_path / "m4.fa").write_bytes(synth.fasta_text(ids, seqs))
    r = subprocess.run([exe, "pair", str(tmp_path / "m1.fq"), str(tmp_path / "m4.fa")], capture_output=True)
    assert r.returncode == 2 and b"different format" in r.stderr


def test_gzip_segments_are_whole_records_and_add_up_to_the_text(tmp_path):
    """host/gzstream.hpp on the CPU: whatever the segment size (64 bytes: every record a segment of its own, up to
    one segment for the file), the segments of a gzip file concatenate to its text, every segment opens with a record
    start and ends with a newline, the last one is marked; FASTQ whose quality lines start with '@', multi-line and
    single-line FASTA, records many segments long, a text without a final newline, a text that is no read file (one
    segment, for the indexer to refuse), one and three gzip members, BGZF blocks inflated side by side; truncated, corrupt
    and empty input"""
    import gzip
    exe = _input_harness(tmp_path)
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def fq(n, long_at=()):
        out = bytearray()
        for i in range(n):
            L = 40000 if i in long_at else int(rng.integers(30, 260))
            q = (b"@" + b"I" * (L - 1)) if i % 3 == 0 else b"I" * L
            out += b"@r%d x\n" % i + acgt[rng.integers(0, 4, L)].tobytes() + b"\n+\n" + q + b"\n"
        return bytes(out)

    def fa(n, width, long_at=()):
        out = bytearray()
        for i in range(n):
            L = 50000 if i in long_at else int(rng.integers(30, 600))
            seq = acgt[rng.integers(0, 4, L)].tobytes()
            out += b">c%d y\n" % i
            out += b"".join(seq[j:j + width] + b"\n" for j in range(0, L, width)) if width else seq + b"\n"
        return bytes(out)

    cases = {"fq": fq(3000, {5, 1500, 2999}), "fa70": fa(1500, 70, {0, 700}), "fa0": fa(1500, 0, {1499}),
             "junk": b"hello world\n" * 5000, "no_final_newline": fq(100)[:-1]}
    f = tmp_path / "x.gz"
    for name, t in cases.items():
        for seg in (64, 1000, 4096, 70000, 10 ** 7):
            for members in (1, 3):
                a, b = len(t) // 3, 2 * len(t) // 3
                f.write_bytes(gzip.compress(t) if members == 1 else gzip.compress(t[:a]) + gzip.compress(t[a:b]) + gzip.compress(t[b:]))
                r = subprocess.run([exe, "gzseg", str(f), str(seg)], capture_output=True, timeout=120)
                assert r.returncode == 0, (name, seg, r.stderr)
                assert r.stdout == t, (name, seg, members)
                w = r.stderr.split()
                n_seg, largest, bad = int(w[1]), int(w[3]), int(w[5])
                assert bad == 0, (name, seg, r.stderr)
                if name == "junk" or seg == 10 ** 7:
                    assert n_seg == 1
                elif seg <= 4096:
                    assert n_seg > len(t) // (3 * max(seg, 700)), (name, seg, r.stderr)      # segments stay segment-sized after a long record
    # BGZF: the blocks inflated by four threads (one thread: the plain inflater does it, member by member); blocks
    # followed by a member that is no block, by trailing bytes; a damaged block, a file that stops inside a block
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import bgzf_compress
    t = cases["fq"]
    zb = bgzf_compress(t)
    assert gzip.decompress(zb) == t
    half = len(t) // 2
    for name, z, blocks in (("bgzf", zb, (len(t) + 65279) // 65280 + 1), ("no_eof_block", bgzf_compress(t, eof_block=False), (len(t) + 65279) // 65280),
                            ("small_blocks", bgzf_compress(t, block=1000), (len(t) + 999) // 1000 + 1),
                            ("then_a_plain_member", bgzf_compress(t[:half], eof_block=False) + gzip.compress(t[half:]), (half + 65279) // 65280),
                            ("trailing_bytes", zb + b"\0\0\0\0garbage", (len(t) + 65279) // 65280 + 1)):
        f.write_bytes(z)
        for seg in (64, 5000, 200000, 10 ** 8):
            for threads in (1, 4):
                r = subprocess.run([exe, "gzseg", str(f), str(seg), str(threads)], capture_output=True, timeout=120)
                assert r.returncode == 0, (name, seg, threads, r.stderr)
                assert r.stdout == t, (name, seg, threads)
                w = r.stderr.split()
                assert int(w[5]) == 0 and int(w[7]) == (blocks if threads == 4 else -1), (name, seg, threads, r.stderr)
    dam = bytearray(zb)
    dam[len(dam) // 2] ^= 0x41
    f.write_bytes(bytes(dam))
    r = subprocess.run([exe, "gzseg", str(f), "200000", "4"], capture_output=True)
    assert r.returncode == 2 and b"corrupt gzip input (bgzf block" in r.stderr
    f.write_bytes(zb[:len(zb) // 2])
    r = subprocess.run([exe, "gzseg", str(f), "200000", "4"], capture_output=True)
    assert r.returncode == 2 and b"truncated gzip input" in r.stderr
    z = gzip.compress(cases["fq"])
    f.write_bytes(z[:-200])
    r = subprocess.run([exe, "gzseg", str(f), "4096"], capture_output=True)
    assert r.returncode == 2 and b"truncated gzip input" in r.stderr
    zz = bytearray(z)
    zz[len(zz) // 2] ^= 0xFF
    zz[len(zz) // 2 + 1] ^= 0x55
    f.write_bytes(bytes(zz))
    r = subprocess.run([exe, "gzseg", str(f), "4096"], capture_output=True)
    assert r.returncode == 2 and b"gzip input" in r.stderr
    f.write_bytes(gzip.compress(b""))
    r = subprocess.run([exe, "gzseg", str(f), "4096"], capture_output=True)
    assert r.returncode == 0 and r.stdout == b"" and b"segments 1 largest 0" in r.stderr


def test_host_input_code_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The host's text code (host/reads.hpp, simd.hpp, input.hpp, pairs.hpp, gzstream.hpp) built with
    -fsanitize=address,undefined (CPU build only; the GPU pool has no sanitizer runs): both indexer fuzzers, index + pack
    of FASTQ / multi-line FASTA on one and five threads, the sequential and the parallel mate join, mates packed from
    two files, gzip segments of plain and BGZF files incl. the truncated and the damaged one -- no report, same exit codes"""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta, bgzf_compress
    exe = str(tmp_path / "host_input_san")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-fopenmp", "-pthread", "-o", exe,
                        os.path.join(ROOT, "tests", "harness", "host_input.cc"), "-lz"], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("no sanitizer runtime in this toolchain: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")

    def run(args, rc=0):
        q = subprocess.run([exe] + [str(a) for a in args], capture_output=True, timeout=300, env=env)
        assert q.returncode == rc, (args, q.returncode, q.stderr[-600:])
        assert b"AddressSanitizer" not in q.stderr and b"LeakSanitizer" not in q.stderr and b"runtime error" not in q.stderr, (args, q.stderr[-900:])
        return q

    run(["indexfuzz", 8000])
    run(["indexfuzz_fasta", 8000])
    genomes = synth.toy_genomes(4, 9000, seed=78)
    names, seqs = mixed_fasta(genomes, 27, seed=32, n=2500)
    f = tmp_path / "r.txt"
    for text in (synth.fastq_text(names, seqs), synth.fasta_text(names, seqs, width=70)):
        f.write_bytes(text)
        a = run(["pack", f, 27, 1]).stdout
        assert a == run(["pack", f, 27, 5]).stdout and len(a) > 50000
    ids = [b"r%d" % i for i in range(len(seqs))]
    clean = [s.replace(b"\n", b"") for s in seqs]
    m1, m2 = tmp_path / "m1.fq", tmp_path / "m2.fq"
    m1.write_bytes(synth.fastq_text([i + b"/1" for i in ids], [s[:100] for s in clean]))
    m2.write_bytes(synth.fastq_text([i + b"/2" for i in ids], [s[-80:] for s in clean]))
    assert run(["pair", m1, m2]).stdout == run(["pairp", m1, m2, 5]).stdout
    run(["packm", m1, m2, 27])
    run(["mates", m1, m2, 7, 262144])
    fq = synth.fastq_text(names, seqs)
    z = tmp_path / "x.gz"
    for blob in (gzip.compress(fq), gzip.compress(fq[:9999]) + gzip.compress(fq[9999:]), bgzf_compress(fq, block=3000),
                 bgzf_compress(fq[:9999], block=3000, eof_block=False) + gzip.compress(fq[9999:])):
        z.write_bytes(blob)
        for seg in (64, 5000, 10 ** 7):
            for threads in (1, 4):
                assert run(["gzseg", z, seg, threads]).stdout == fq
        assert run(["load", z, 4]).stdout == fq
    z.write_bytes(bgzf_compress(fq, block=3000)[:-500])
    run(["gzseg", z, 5000, 4], rc=2)
    run(["load", z, 4], rc=2)
    dam = bytearray(bgzf_compress(fq, block=3000))
    dam[len(dam) // 2] ^= 0x41
    z.write_bytes(bytes(dam))
    run(["gzseg", z, 5000, 4], rc=2)
    run(["load", z, 4], rc=2)


def test_threaded_input_code_is_clean_under_the_thread_sanitizer(tmp_path):
    """-fsanitize=thread over the parts of the host's input code that run on std::thread: the gzip segmenter (a producer
    thread, the consumer, and four threads inflating BGZF blocks), the whole-file BGZF inflater and the parallel mate join
    (the OpenMP loops are left out: libgomp is not instrumented and reports races that are not there)"""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import bgzf_compress
    exe = str(tmp_path / "host_input_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-fopenmp", "-pthread", "-o", exe,
                        os.path.join(ROOT, "tests", "harness", "host_input.cc"), "-lz"], capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("no thread sanitizer runtime in this toolchain: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(2)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = b"".join(b"@r%d/1\n" % i + acgt[rng.integers(0, 4, 150)].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(20000))

    def run(args, rc=0):
        q = subprocess.run([exe] + [str(a) for a in args], capture_output=True, timeout=300)
        if b"FATAL: ThreadSanitizer" in q.stderr:          # (the runtime could not set itself up in this container: nothing learnt)
            pytest.skip("thread sanitizer runtime does not start here: " + q.stderr[-200:].decode(errors="replace"))
        assert q.returncode == rc and b"ThreadSanitizer" not in q.stderr, (args, q.returncode, q.stderr[-900:])
        return q

    z = tmp_path / "t.gz"
    for blob in (bgzf_compress(t), gzip.compress(t)):
        z.write_bytes(blob)
        for seg in (5000, 2000000):
            assert run(["gzseg", z, seg, 4]).stdout == t
        assert run(["load", z, 4]).stdout == t
    f1, f2 = tmp_path / "p1.fq", tmp_path / "p2.fq"
    f1.write_bytes(t)
    f2.write_bytes(t.replace(b"/1\n", b"/2\n"))
    assert run(["pairp", f1, f2, 5]).stdout.count(b">") == 20000


def test_fast_g_format_matches_printf(tmp_path):
    """host/format.hpp: the CSV writer's printf-free "%g" of a ratio agrees with snprintf for every
    a <= b <= 2000 and a million random pairs up to 2^20, and declines what it does not cover"""
    exe = str(tmp_path / "host_format")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "harness", "host_format.cc")], check=True)
    r = subprocess.run([exe, "2000", "1000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("checked ")
    # and a smaller sweep of the same under the address / UB sanitizers
    san = str(tmp_path / "host_format_san")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-std=c++17", "-o", san, os.path.join(ROOT, "tests", "harness", "host_format.cc")],
                       capture_output=True, text=True)
    if b.returncode == 0:
        r = subprocess.run([san, "600", "100000"], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("checked "), r.stderr[-800:]
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-800:]


@pytest.mark.gpu
@pytest.mark.parametrize("members,env,expect", [
    ("0,0", {"MC_GROUP_MODE": "shards"}, "shards by minimizer line range"),
    ("0,0,0", {"MC_GROUP_HBM_BYTES": "100000000"}, "shards by minimizer line range"),      # "does not fit": AUTO picks shards
    ("0,0,0", {"MC_GROUP_MODE": "shards", "MC_GROUP_SHARD": "buckets"}, "shards by bucket range"),
    ("0,0", {"MC_GROUP_MODE": "shards", "MC_INDEX": "lines"}, "shards by bucket range"),
    ("0,0,0", {}, "replicas"),
    # as few parts as the budget dictates, replicated: 4 members, a table that needs 2 -> 2 parts x 2 groups
    ("0,0,0,0", {"MC_GROUP_HBM_BYTES": "fit2"}, "shards by minimizer line range: 2 parts x 2 groups"),
    ("0,0,0,0,0", {"MC_GROUP_MODE": "shards", "MC_GROUP_PARTS": "2"}, "shards by minimizer line range: 2 parts x 2 groups"),
    # the minimizer lines "do not fit" anywhere (MC_MZ_ALLOC_LIMIT): one member falls back to the bucket-line table, two
    # members first cut the table into parts, then fall back to the reference's bucket ranges
    # the super-k-mer index (MC_INDEX=skm) behind the same group interface: replicas, and parts by minimizer whose members
    # take the first member's line count
    ("0,0,0", {"MC_INDEX": "skm"}, "replicas"),
    ("0,0", {"MC_GROUP_MODE": "shards", "MC_INDEX": "skm"}, "shards by minimizer line range"),
    ("0,0,0", {"MC_GROUP_MODE": "shards", "MC_INDEX": "auto"}, "shards by minimizer line range"),
    ("0", {"MC_MZ_ALLOC_LIMIT": "65536"}, "replicas"),
    ("0,0", {"MC_MZ_ALLOC_LIMIT": "65536"}, "shards by bucket range"),
    # a table larger than all devices together (forced: MC_GROUP_CYCLES): C x N parts, N at a time, every batch classified once
    # per cycle and the rows of the cycles merged on the card (the reference's swapDbParts loop, CuClarkDB.cu:775-815)
    ("0", {"MC_GROUP_CYCLES": "2"}, "shards by minimizer line range: 1 parts x 1 groups, 2 database cycles per file"),
    ("0,0", {"MC_GROUP_CYCLES": "3"}, "shards by minimizer line range: 2 parts x 1 groups, 3 database cycles per file"),
    ("0,0", {"MC_GROUP_CYCLES": "2", "MC_INDEX": "skm"}, "shards by minimizer line range: 2 parts x 1 groups, 2 database cycles per file"),
    # not forced: the minimizer lines of the whole table (1140 lines) "do not fit" the one member, the bucket-line table is kept
    # out of the way -- the loader goes to cycles, and half the lines fit
    ("0", {"MC_MZ_ALLOC_LIMIT": "100000", "MC_GROUP_NO_LINES_FALLBACK": "1", "MC_INDEX": "minimizer"},
     "shards by minimizer line range: 1 parts x 1 groups, 2 database cycles per file"),
])
@pytest.mark.parametrize("extended", [False, True])
def test_several_devices_produce_the_single_device_csv(oracle, tmp_path, members, env, expect, extended):
    """-d N through mc_group (include/mc_group.h), rehearsed with N contexts on the one card of the test box
    (MC_GROUP_DEVICES): a table that "does not fit" is cut into line-range shards (bucket ranges as in the
    reference, CuClarkDB.cu:552-559, when asked for or when the minimizer index is not in use), every shard
    sees every batch (:842-851), rows are exchanged device-to-device by read range and merged by the owner
    (:909-928 is the reference's tree).  The CSV must be byte-identical to the one-device run -- with
    several batches in flight, submitted out of order by four packer threads."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    _build()
    k, ht = 27, 57777779
    genomes = synth.toy_genomes(6, 6000, seed=61, shared=400)
    labels = ["A", "B", "C", "D", "E", "F"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    names, seqs = mixed_fasta(genomes, k, seed=23, n=4000)
    text = synth.fastq_text(names, seqs)
    (tmp_path / "reads.fq").write_bytes(text)
    common = ["-T", targets, "-D", str(dbdir), "-O", str(tmp_path / "reads.fq"), "-n", "4", "-b", "9", "--verbose"]
    if extended:
        common.append("--extended")
    one = subprocess.run([os.path.join(BIN, "cuCLARK-l")] + common + ["-R", str(tmp_path / "one"), "-d", "1"],
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr
    assert "Devices: 1 (replicas" in one.stderr
    if env.get("MC_GROUP_HBM_BYTES") == "fit2":
        # a per-device budget that holds half of this table but not all of it (mc_index_plan is the loader's
        # own arithmetic: min_parts must come out as 2); the database was built by the run above
        from jn_cuclark_amd import _lib
        n_keys = os.path.getsize(str(dbdir / ("db_central_k27_t6_s%d_m0_light_4.tsk.ky" % ht))) // 4
        lo, hi = 1 << 30, 64 << 30
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if 1 <= _lib.index_plan(n_keys, 1, mid)["min_parts"] <= 2:
                hi = mid
            else:
                lo = mid
        assert _lib.index_plan(n_keys, 1, hi)["min_parts"] == 2
        env = dict(env, MC_GROUP_HBM_BYTES=str(hi))
    e = dict(os.environ, MC_GROUP_DEVICES=members, **env)
    many = subprocess.run([os.path.join(BIN, "cuCLARK-l")] + common + ["-R", str(tmp_path / "many")],
                          capture_output=True, text=True, timeout=900, env=e)
    assert many.returncode == 0, many.stderr
    assert ("Devices: %d (%s" % (len(members.split(",")), expect)) in many.stderr, many.stderr
    if env.get("MC_INDEX") == "skm":          # ("auto" may take either: this toy table holds a few hundred k-mers)
        assert "super-k-mer index" in many.stderr, many.stderr
    if "MC_GROUP_CYCLES" in env:
        c, n = int(env["MC_GROUP_CYCLES"]), len(members.split(","))
        assert ("%d parts, %d at a time (%d cycles per file)" % (c * n, n, c)) in many.stderr, many.stderr
        assert ("database cycle %d of %d" % (c - 1, c)) in many.stderr, many.stderr
    if "MC_GROUP_NO_LINES_FALLBACK" in env:
        assert "database cycles per file" in many.stderr and "at a time (" in many.stderr, many.stderr
    elif "MC_MZ_ALLOC_LIMIT" in env:
        assert "falling back to the bucket-line table" in many.stderr and "[fallback: the minimizer index did not fit]" in many.stderr, many.stderr
    if "replicas" not in expect and "parts x" not in expect:
        assert ("%d parts x 1 groups" % len(members.split(","))) in many.stderr, many.stderr
    a, b = open(str(tmp_path / "one.csv")).read(), open(str(tmp_path / "many.csv")).read()
    assert a == b
    base = str(dbdir / ("db_central_k27_t6_s%d_m0_light_4.tsk" % ht))
    want, _ = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels, extended=extended)
    assert b == want
    assert sum(ln.split(",")[-3] != "NA" for ln in b.split("\n")[1:-1]) > 1500


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mates", "fasta", "two_files"])
def test_database_cycles_with_mates_and_fasta(tmp_path, kind):
    """The cycled database (see above) with the other inputs: mates of two FASTQ files -- joined first, the file is classified
    once per cycle from the joined text --, a multi-line FASTA file, and a list of two files (the second starts with the parts the
    first ended with); byte-identical to the run on the resident table."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    _build()
    k = 27
    genomes = synth.toy_genomes(5, 5000, seed=67, shared=300)
    labels = ["A", "B", "C", "D", "E"]
    targets = _write_targets(tmp_path, genomes, labels, n_mask=False)
    (tmp_path / "db").mkdir()
    names, seqs = mixed_fasta(genomes, k, seed=29, n=3000)
    if kind == "mates":
        nm = [n.split(b" ")[0] for n in names]
        (tmp_path / "r_1.fq").write_bytes(synth.fastq_text([n + b"/1" for n in nm], [s[:90].replace(b"\n", b"") for s in seqs]))
        (tmp_path / "r_2.fq").write_bytes(synth.fastq_text([n + b"/2" for n in nm], [s[50:150] for s in seqs]))
        inp = ["-P", str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")]
    elif kind == "two_files":
        (tmp_path / "a.fq").write_bytes(synth.fastq_text(names[:1700], seqs[:1700]))
        (tmp_path / "b.fa").write_bytes(synth.fasta_text(names[1700:], seqs[1700:], width=60))
        (tmp_path / "objects.txt").write_text("%s\n%s\n" % (tmp_path / "a.fq", tmp_path / "b.fa"))
        inp = ["-O", str(tmp_path / "objects.txt")]
    else:
        (tmp_path / "reads.fa").write_bytes(synth.fasta_text(names, seqs, width=60))
        inp = ["-O", str(tmp_path / "reads.fa")]
    common = ["-T", targets, "-D", str(tmp_path / "db")] + inp + ["-n", "4", "-b", "7", "--verbose"]
    out = {}
    for tag, env in (("resident", {}), ("cycled", {"MC_GROUP_CYCLES": "3", "MC_STREAM_MIN_BYTES": "1000"})):
        res = str(tmp_path / tag)
        if kind == "two_files":          # a list of result names next to the list of inputs (reference run(), src/CuCLARK_hh.hh:383-506)
            (tmp_path / (tag + "_results.txt")).write_text("%s_a\n%s_b\n" % (res, res))
            res = str(tmp_path / (tag + "_results.txt"))
        r = subprocess.run([os.path.join(BIN, "cuCLARK-l")] + common + ["-R", res, "-d", "1"],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert ("3 database cycles per file" in r.stderr) == (tag == "cycled"), r.stderr
        if kind == "two_files":
            out[tag] = open(str(tmp_path / (tag + "_a.csv"))).read() + open(str(tmp_path / (tag + "_b.csv"))).read()
        else:
            out[tag] = open(str(tmp_path / (tag + ".csv"))).read()
    assert out["cycled"] == out["resident"] and out["cycled"].count("\n") == (3002 if kind == "two_files" else 3001)
    assert sum(ln.split(",")[-3] != "NA" for ln in out["cycled"].split("\n")[1:-1]) > 1000


@pytest.mark.gpu
def test_more_devices_than_present_is_refused(tmp_path):
    _build()
    genomes = synth.toy_genomes(2, 2000, seed=71)
    targets = _write_targets(tmp_path, genomes, ["A", "B"], n_mask=False)
    (tmp_path / "db").mkdir()
    (tmp_path / "r.fa").write_bytes(synth.fasta_text([b"r"], [synth.codes_to_ascii(genomes[0][:150])]))
    r = _run("cuCLARK-l", ["-T", targets, "-D", str(tmp_path / "db"), "-O", str(tmp_path / "r.fa"), "-R", str(tmp_path / "o"), "-d", "99"])
    assert r.returncode != 0 and "99 devices requested. Insufficient devices found. Abort." in r.stderr
