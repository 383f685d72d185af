#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own CPU code.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py            # light table (HTSIZE 57777779, k=27)
    python tests/golden/make_golden.py --full     # + full table (HTSIZE 1610612741, k=31, ~30 GB RAM)

What the reference computes here (oracle/ref_ht_driver.cc is only a harness around it):
  * kmer_vectors.npz   string -> value / reverse complement
                       (vectorToIndex + getReverseComplement, src/kmersConversion.cc:39-87)
  * db_<variant>.npz   occurrences -> discriminative database files
                       (EHashtable::addElement -> SortAllHashTable -> RemoveCommon -> Write,
                        src/HashTableStorage_hh.hh:421-461, :229-280, src/hashTable_hh.hh:473-546)
                       and lookups (EHashtable::Read + queryElement(uint64) = hTable::find,
                        src/hashTable_hh.hh:358-396) -- the function the GPU lookup mirrors
                        (src/CuClarkDB.cu:1185-1254).
The fixtures are data (inputs + the reference's outputs); no reference source is stored.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jn_cuclark_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.dirname(os.path.abspath(__file__))


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw).stdout


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def kmer_vectors(tmp):
    rows = {}
    for k, exe in ((27, "ref_ht_light"), (31, "ref_ht_full"), (32, "ref_ht_full"), (20, "ref_ht_light")):
        codes = synth.random_codes(1000 + k, 64 * k).reshape(64, k)
        strs = [synth.codes_to_ascii(c).decode() for c in codes]
        # a few hand-picked ones: homopolymers, palindromes, lower case
        strs += ["A" * k, "C" * k, "G" * k, "T" * k, ("AC" * k)[:k], ("acgt" * k)[:k], ("TTAA" * k)[:k]]
        p = os.path.join(tmp, "km%d.txt" % k)
        open(p, "w").write("\n".join(strs) + "\n")
        out = run([os.path.join(REF, exe), "kmer", str(k), p])
        fw, rv = [], []
        for line in out.strip().split("\n"):
            s, a, b = line.split("\t")
            fw.append(int(a)); rv.append(int(b))
        rows["k%d_str" % k] = np.array(strs)
        rows["k%d_fwd" % k] = np.array(fw, dtype=np.uint64)
        rows["k%d_rc" % k] = np.array(rv, dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "kmer_vectors.npz"), **rows)
    print("kmer_vectors.npz written")


def db_fixture(tmp, variant, exe, k, htsize, seed):
    # three toy genomes; genome 1 shares a block with genome 0 (non-discriminative),
    # genome 2 contains an internal repeat (same k-mers twice in ONE target: kept)
    genomes = synth.toy_genomes(3, 3000, seed, shared=400)
    genomes[2] = genomes[2].copy()
    genomes[2][2000:2200] = genomes[2][100:300]
    # plus the reverse complement of a stretch of genome 0 placed in genome 1
    rc_block = (3 - genomes[0][1000:1200])[::-1]
    genomes[1] = genomes[1].copy()
    genomes[1][2500:2700] = rc_block
    km = np.concatenate([synth.kmers_of(g, k) for g in genomes])
    tg = np.concatenate([np.full(g.size - k + 1, i, dtype=np.uint16) for i, g in enumerate(genomes)])
    tsv = os.path.join(tmp, "occ_%s.tsv" % variant)
    with open(tsv, "w") as f:
        for x, t in zip(km.tolist(), tg.tolist()):
            f.write("%d\tt%d\n" % (x, t))
    base = os.path.join(tmp, "db_%s" % variant)
    out = run([os.path.join(REF, exe), "build", str(k), "0", tsv, base])
    stored = int([ln for ln in out.split("\n") if ln.startswith("stored")][0].split("\t")[1])
    sz = np.fromfile(base + ".sz", dtype=np.uint8)
    ky = np.fromfile(base + ".ky", dtype=np.uint32)
    lb = np.fromfile(base + ".lb", dtype=np.uint16)
    assert sz.size == htsize and ky.size == stored and lb.size == stored
    nz = np.flatnonzero(sz).astype(np.uint32)

    # lookups: every genome k-mer, its reverse complement, neighbours and random values
    q = np.concatenate([km, synth.revcomp(km, k), km ^ np.uint64(1),
                        synth.rand_u64(seed + 77, 3000) & np.uint64((1 << (2 * k)) - 1)])
    qp = os.path.join(tmp, "q_%s.txt" % variant)
    np.savetxt(qp, q, fmt="%d")
    res = run([os.path.join(REF, exe), "query", str(k), base, qp])
    found, label = [], []
    for line in res.strip().split("\n"):
        _, a, b = line.split("\t")
        found.append(int(a)); label.append(int(b))
    np.savez_compressed(
        os.path.join(OUT, "db_%s.npz" % variant),
        k=np.int64(k), htsize=np.int64(htsize),
        occ_kmers=km, occ_targets=tg,
        nonzero_buckets=nz, nonzero_sizes=sz[nz], keys=ky, labels=lb,
        sha256=np.array([sha(base + ".sz"), sha(base + ".ky"), sha(base + ".lb")]),
        query_kmers=q, query_found=np.array(found, dtype=np.uint8), query_label=np.array(label, dtype=np.uint16))
    for ext in (".sz", ".ky", ".lb"):
        os.remove(base + ext)
    print("db_%s.npz written: %d stored k-mers, %d lookups, %d found" % (variant, stored, q.size, sum(found)))


def config0_targets(tmp):
    """BASELINE configs[0] plumbing: files-to-taxonomy table -> targets.txt through the reference's
    getTargetsDef (src/getTargetsDef.cc:38-96, run by scripts/set_targets.sh:117).  The input table
    is ours (three genome files of the toy database, one file without taxonomy, one with an UNKNOWN
    species); the outputs for rank 0 (species) and rank 1 (genus) and files_excluded.txt are the
    reference's."""
    rows = [
        "Custom/genome0.fa\t511145\t562\t561\t543\t91347\t1236\t1224",
        "Custom/genome1.fa,93061,1280,1279,90964,1385,91061,1239",
        "Custom/plasmid_x.fa -1 -1 -1 -1 -1 -1 -1",
        "Custom/genome2.fa\t224308\t1423\t1386\t186817\t1385\t91061\t1239",
        "Custom/unplaced.fa\t12345\tUNKNOWN\t1386\t186817\t1385\t91061\t1239",
    ]
    d = os.path.join(OUT, "config0")
    os.makedirs(d, exist_ok=True)
    src = os.path.join(d, "custom.fileToTaxIDs")
    open(src, "w").write("\n".join(rows) + "\n")
    for rank in (0, 1):
        out = run([os.path.join(REF, "ref_getTargetsDef"), src, str(rank)], cwd=tmp)
        open(os.path.join(d, "targets_rank%d.txt" % rank), "w").write(out)
    open(os.path.join(d, "files_excluded.txt"), "w").write(open(os.path.join(tmp, "files_excluded.txt")).read())
    print("config0/: targets_rank0.txt, targets_rank1.txt, files_excluded.txt written")


def abundance_inputs(d):
    """a toy NCBI-style taxonomy (two phyla, a species group, a strain below a species, an id the tree does not
    know) and two per-read result files in the classifier's CSV format"""
    os.makedirs(os.path.join(d, "db", "taxonomy"), exist_ok=True)
    nodes = [  # id, parent, rank
        (1, 1, "no rank"), (131567, 1, "no rank"), (2, 131567, "superkingdom"),
        (1224, 2, "phylum"), (1236, 1224, "class"), (91347, 1236, "order"), (543, 91347, "family"),
        (561, 543, "genus"), (562, 561, "species"), (511145, 562, "no rank"),
        (590, 543, "genus"), (28901, 590, "species"),
        (1239, 2, "phylum"), (91061, 1239, "class"), (1385, 91061, "order"), (90964, 1385, "family"),
        (1279, 90964, "genus"), (1280, 1279, "species"),
        (186817, 1385, "family"), (1386, 186817, "genus"), (653685, 1386, "species group"), (1423, 653685, "species"),
        (10239, 1, "superkingdom"), (10508, 10239, "family"), (10509, 10508, "genus"), (129951, 10509, "species"),
    ]
    names = {1: "root", 131567: "cellular organisms", 2: "Bacteria", 1224: "Proteobacteria", 1236: "Gammaproteobacteria",
             91347: "Enterobacterales", 543: "Enterobacteriaceae", 561: "Escherichia", 562: "Escherichia coli",
             511145: "Escherichia coli str. K-12 substr. MG1655", 590: "Salmonella", 28901: "Salmonella enterica",
             1239: "Firmicutes", 91061: "Bacilli", 1385: "Bacillales", 90964: "Staphylococcaceae", 1279: "Staphylococcus",
             1280: "Staphylococcus aureus", 186817: "Bacillaceae", 1386: "Bacillus", 653685: "Bacillus subtilis group",
             1423: "Bacillus subtilis", 10239: "Viruses", 10508: "Adenoviridae", 10509: "Mastadenovirus",
             129951: "Human mastadenovirus C"}
    with open(os.path.join(d, "db", "taxonomy", "nodes.dmp"), "w") as f:
        for i, p_, r in nodes:
            f.write("%d\t|\t%d\t|\t%s\t|\t\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t0\t|\t0\t|\t\t|\n" % (i, p_, r))
    with open(os.path.join(d, "db", "taxonomy", "names.dmp"), "w") as f:
        for i in sorted(names):
            if i == 562:
                f.write("562\t|\tBacillus coli\t|\t\t|\tsynonym\t|\n")
            f.write("%d\t|\t%s\t|\t\t|\tscientific name\t|\n" % (i, names[i]))
    rng = np.random.default_rng(3)
    labels = ["562", "562", "562", "1280", "1423", "28901", "129951", "511145", "999999", "NA"]
    for fi, n in ((1, 400), (2, 150)):
        with open(os.path.join(d, "result%d.csv" % fi), "w") as f:
            f.write("Object_ID,Gamma,Assignment,Score,Confidence\n")
            for i in range(n):
                lab = labels[int(rng.integers(0, len(labels)))]
                gamma = float(rng.choice([0.0, 0.01, 0.02, 0.05, 0.3, 0.85]))
                best = int(rng.integers(1, 120))
                second = int(rng.choice([0, 0, 1, best // 3, best]))
                conf = best / (best + second)
                if lab == "NA":
                    f.write("read%d_%d,0,NA,0,0\n" % (fi, i))
                else:
                    f.write("read%d_%d,%g,%s,%d,%g\n" % (fi, i, gamma, lab, best, conf))
    with open(os.path.join(d, "result_ext.csv"), "w") as f:                    # --extended layout: per-target columns
        f.write("Object_ID,562,1280,1423,Gamma,Assignment,Score,Confidence\n")
        for i in range(60):
            lab = ["562", "1280", "1423", "NA"][i % 4]
            f.write("x%d,%d,%d,%d,%g,%s,%d,%g\n" % (i, i % 7, i % 5, i % 3, 0.5 if lab != "NA" else 0, lab, 9 if lab != "NA" else 0,
                                                 0.9 if i % 3 else 0.6))


def abundance_golden(tmp):
    """BASELINE's consumers of the CSV: what the reference's getAbundance (src/getAbundance.cc) and
    kent -m / -r (app/kent.cpp:605-820) write for the inputs above.  Inputs and the reference's outputs are
    stored under tests/golden/abundance/."""
    d = os.path.join(OUT, "abundance")
    os.makedirs(d, exist_ok=True)
    abundance_inputs(d)
    ga, kent = os.path.join(REF, "ref_getAbundance"), os.path.join(REF, "ref_kent")
    cases = {
        "plain": ["-F", "result1.csv"],
        "two_files": ["-F", "result1.csv", "result2.csv"],
        "taxonomy": ["-D", "db", "-F", "result1.csv", "result2.csv"],
        "highconf": ["--highconfidence", "-D", "db", "-F", "result1.csv"],
        "filters": ["-c", "0.8", "-g", "0.02", "-a", "5", "-D", "db", "-F", "result1.csv", "result2.csv"],
        "extended": ["-D", "db", "-F", "result_ext.csv"],
        "exports": ["-D", "db", "-F", "result1.csv", "--krona", "--mpa"],
    }
    for name, args in cases.items():
        r = subprocess.run([ga] + args, cwd=d, check=True, capture_output=True, text=True)
        open(os.path.join(d, "out_%s.csv" % name), "w").write(r.stdout)
        if name == "exports":
            for f in ("results.krn", "results.mpa"):
                os.replace(os.path.join(d, f), os.path.join(d, "out_" + f))
    # kent -m / -r work on abundance tables; they write under ./results/
    os.makedirs(os.path.join(tmp, "results"), exist_ok=True)
    for a, b, out in (("out_taxonomy.csv", "out_highconf.csv", "merged_lineage.csv"), ("out_plain.csv", "out_two_files.csv", "merged_plain.csv")):
        subprocess.run([kent, "-m", os.path.join(d, a), os.path.join(d, b), "-o", out], cwd=tmp, check=True, capture_output=True)
        os.replace(os.path.join(tmp, "results", out), os.path.join(d, out))
    for src, out in (("out_taxonomy.csv", "report_taxonomy.txt"), ("merged_lineage.csv", "report_merged.txt")):
        subprocess.run([kent, "-r", os.path.join(d, src)], cwd=tmp, check=True, capture_output=True)
        os.replace(os.path.join(tmp, "results", "report.txt"), os.path.join(d, out))
    print("abundance/: %d tables, 2 merges, 2 reports written" % len(cases))


def tsk_golden(tmp):
    """--tsk: the per-target text files of target-specific k-mers (createTargetFilesNames, src/CuCLARK_hh.hh:342-378;
    EHashtable::SaveMultiple, src/HashTableStorage_hh.hh:282-327), written by the REFERENCE's host driver
    (oracle/_ref/ref_host_mc_full = src/main.cc + src/CuCLARK_hh.hh compiled where they lie): its builder runs on the
    CPU before any device is opened, so the run ends with "No HIP devices" after the files are on disk.  Full variant,
    k = 31 (the chained table of 1610612741 buckets: ~40 GB for a minute); the light variant's file names index an
    empty vector in the reference (:367) and cannot be produced.  Inputs (three small genomes: a shared block,
    a repeat inside one genome, an N) and outputs are the fixture."""
    import numpy as np
    d = os.path.join(OUT, "tsk")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(2024)
    base = "ACGT"
    g = ["".join(base[i] for i in rng.integers(0, 4, n)) for n in (700, 650, 720)]
    shared = g[0][100:180]
    g[1] = g[1][:300] + shared + g[1][380:]               # in two targets: not specific
    g[2] = g[2][:200] + g[2][50:120] + g[2][270:]         # twice in one target: count 2
    g[0] = g[0][:400] + "N" + g[0][401:]
    # three k-mers of ONE bucket (values v, v + HTSIZE, v + 2 HTSIZE, each its own canonical form), met in the order
    # 2, 0, 1: the reference lists a bucket in insertion order, not in key order
    H, k = 1610612741, 31
    code = {"A": 3, "C": 2, "G": 1, "T": 0}
    text = lambda v: "".join("TGCA"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))
    rcv = lambda v: sum((3 - ((v >> (2 * i)) & 3)) << (2 * (k - 1 - i)) for i in range(k))
    v = None
    while v is None:
        c = int(rng.integers(0, 1 << 61))
        if all(x <= rcv(x) for x in (c, c + H, c + 2 * H)):
            v = c
    assert all(sum(code[ch] << (2 * (k - 1 - i)) for i, ch in enumerate(text(x))) == x for x in (v, v + H))
    g[0] += "N" + text(v + 2 * H) + "N" + text(v) + "N" + text(v + H)
    lines = []
    for i, seq in enumerate(g):
        fa = os.path.join(d, "g%d.fa" % i)
        open(fa, "w").write(">g%d toy\n" % i + "\n".join(seq[j:j + 70] for j in range(0, len(seq), 70)) + "\n")
        lines.append("%s\t%s\n" % (os.path.join("tests", "golden", "tsk", "g%d.fa" % i), ("T%d" % i) if i < 2 else "S9"))
    open(os.path.join(d, "targets.txt"), "w").write("".join(lines))
    db = os.path.join(tmp, "tskdb")
    os.makedirs(db, exist_ok=True)
    reads = os.path.join(tmp, "r.fa")
    open(reads, "w").write(">r\n" + g[0][:100] + "\n")
    r = subprocess.run([os.path.join(REF, "ref_host_mc_full"), "-k", "31", "-T", os.path.join(d, "targets.txt"), "-D", db + "/",
                        "-O", reads, "-R", os.path.join(tmp, "res"), "--tsk"], cwd=ROOT, capture_output=True, text=True)
    print(r.stderr[-600:])
    got = sorted(f for f in os.listdir(db) if f.endswith(".ht"))
    assert got, "the reference wrote no .ht files"
    for f in got:
        shutil.copy(os.path.join(db, f), os.path.join(d, f))
    open(os.path.join(d, "db_sha256.txt"), "w").write("".join(
        "%s  %s\n" % (sha(os.path.join(db, f)), f.split(".tsk")[-1]) for f in sorted(os.listdir(db)) if ".tsk." in f))
    assert "maximum number of collisions: 3" in r.stderr, "the crafted bucket did not collide"
    print("tsk/: %s written" % ", ".join(got))


def main():
    full = "--full" in sys.argv
    if "--tsk-only" in sys.argv:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            tsk_golden(tmp)
        return
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        config0_targets(tmp)
        abundance_golden(tmp)
        if "--config0-only" in sys.argv or "--text-only" in sys.argv:
            return
        kmer_vectors(tmp)
        db_fixture(tmp, "light_k27", "ref_ht_light", 27, 57777779, 11)
        if full:
            db_fixture(tmp, "full_k31", "ref_ht_full", 31, 1610612741, 12)


if __name__ == "__main__":
    main()
