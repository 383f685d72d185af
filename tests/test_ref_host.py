"""GPU suite: the REFERENCE's own host driver (src/main.cc + src/CuCLARK_hh.hh: targets
parser, database builder, FASTA/FASTQ indexer, read packer, CSV writer), linked against
our HIP backend through integration/CuClarkDB_mc.cc (oracle/_ref/ref_host_mc_light, built
by `make -C oracle ref_host` where /root/reference exists), next to our own host driver
bin/cuCLARK-l on the same inputs.  Database files and CSV must be byte-identical.

This pins our host side (builder, indexer, packer, CSV formatting) against the reference's
real code, and shows the C ABI is a working drop-in for src/CuClarkDB.cu."""
import filecmp
import os
import subprocess
import sys

import pytest

from jn_cuclark_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_host_mc_light")
OURS = os.path.join(ROOT, "bin", "cuCLARK-l")


def _targets(tmp_path, genomes, labels):
    lines = []
    for i, g in enumerate(genomes):
        s = bytearray(synth.codes_to_ascii(g))
        s[777] = ord("N")
        s[2000:2010] = b"nnnnnRYKMS"
        p = tmp_path / ("genome%d.fa" % i)
        p.write_bytes(synth.fasta_text([b"contig%d x" % i, b"contig%d_b" % i], [bytes(s[:3000]), bytes(s[3000:])], width=80))
        lines.append("%s\t%s\n" % (p, labels[i]))
    t = tmp_path / "targets.txt"
    t.write_text("".join(lines))
    return str(t)


@pytest.mark.parametrize("mode", ["fasta_multiline", "fastq_3batches", "paired", "extended", "spectrum_targets",
                                  "fastq_3batches_cycles", "extended_cycles", "paired_cycles"])
def test_reference_host_and_our_host_agree_byte_for_byte(tmp_path, mode):
    # *_cycles: the reference's host drives a database that "does not fit" (MC_GROUP_CYCLES=3: three times one part of three)
    # through its own swapDbParts / queryBatch(.., followup) loop (src/CuCLARK_hh.hh:1765-1772); our host on the resident table
    cycles = mode.endswith("_cycles")
    if cycles:
        mode = mode[:-len("_cycles")]
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/ref_host_mc_light not built (needs /root/reference at build time)")
    if not os.path.exists(OURS):
        import __graft_entry__
        __graft_entry__.build()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import mixed_fasta
    k = 27
    genomes = synth.toy_genomes(5, 6000, seed=61, shared=700)
    labels = ["Ecoli", "Saureus", "Bsub", "Ecoli", "Paer"]
    targets = _targets(tmp_path, genomes, labels)
    if mode == "spectrum_targets":
        # two of the five targets as k-mer spectra ("<k-mer> <count>" per line, CuCLARK_hh.hh:861-876): the k-mers of
        # the genome in file order with small counts, a few of them repeated
        lines = open(targets).read().split("\n")[:-1]
        for i in (1, 4):
            km = synth.kmers_of(genomes[i], k)
            rows = []
            for j, x in enumerate(km[: 4000].tolist()):
                s_ = "".join("TGCA"[(x >> (2 * (k - 1 - t))) & 3] for t in range(k))
                rows.append("%s %d" % (s_ if j % 3 else s_.lower(), 1 + j % 4))
            p_ = tmp_path / ("spectrum%d.txt" % i)
            p_.write_text("\n".join(rows + rows[:50]) + "\n")
            lines[i] = "%s\t%s" % (p_, labels[i])
        open(targets, "w").write("\n".join(lines) + "\n")
    names, seqs = mixed_fasta(genomes, k, seed=17, n=1200)
    names = [n + b" trailing words" for n in names]
    extra = []
    if mode == "paired":
        nm = [n.split(b" ")[0] for n in names]
        m1 = [s[:100].replace(b"\n", b"") for s in seqs]
        m2 = [s[40:150] for s in seqs]
        f1, f2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
        f1.write_bytes(synth.fastq_text([n + b"/1" for n in nm], m1))
        f2.write_bytes(synth.fastq_text([n + b"/2" for n in nm], m2))
        inp = ["-P", str(f1), str(f2)]
    elif mode == "fastq_3batches":
        p = tmp_path / "reads.fq"
        p.write_bytes(synth.fastq_text(names, seqs))
        inp = ["-O", str(p)]
        extra = ["-b", "3"]
    else:
        p = tmp_path / "reads.fa"
        p.write_bytes(synth.fasta_text(names, seqs, width=50))
        inp = ["-O", str(p)]
        if mode == "extended":
            extra = ["--extended"]
    outs = {}
    for tag, exe in (("ref", REF), ("ours", OURS)):
        d = tmp_path / ("db_" + tag)
        d.mkdir()
        env = dict(os.environ, MC_GROUP_CYCLES="3") if cycles and tag == "ref" else None
        r = subprocess.run([exe, "-T", targets, "-D", str(d)] + inp + ["-R", str(tmp_path / ("res_" + tag))] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        if env:
            assert "3 parts, 1 at a time (3 cycles per file)" in r.stderr, r.stderr[-1500:]
        outs[tag] = d
    name = "db_central_k27_t4_s57777779_m0_light_4.tsk"
    for ext in (".sz", ".ky", ".lb"):
        a, b = str(outs["ref"] / (name + ext)), str(outs["ours"] / (name + ext))
        assert os.path.getsize(a) > 0
        assert filecmp.cmp(a, b, shallow=False), ext
    ref_csv = open(str(tmp_path / "res_ref.csv")).read()
    our_csv = open(str(tmp_path / "res_ours.csv")).read()
    assert ref_csv.count("\n") == 1201
    assert our_csv == ref_csv
    assigned = sum(1 for ln in ref_csv.split("\n")[1:-1] if ",NA," not in ln)
    assert assigned > 500


def test_reference_recovery_from_ht_files_and_ours_write_the_same_database(tmp_path):
    """--tsk with the database files gone and the per-target .ht files still there: the REFERENCE's host driver
    (full variant, oracle/_ref/ref_host_mc_full) puts the database back together with its own code
    (loadSpecificTargetSets, src/CuCLARK_hh.hh:633-684: EHashtable::Load + SortAllHashTable + Write) and leaves with
    exit(-1); bin/cuCLARK does the same.  The three files must be byte-identical, and equal to what the reference's
    builder wrote for these genomes in the first place (tests/golden/tsk/db_sha256.txt)."""
    import hashlib
    import shutil
    ref_full = os.path.join(ROOT, "oracle", "_ref", "ref_host_mc_full")
    ours_full = os.path.join(ROOT, "bin", "cuCLARK")
    if not os.path.exists(ref_full):
        pytest.skip("oracle/_ref/ref_host_mc_full not built (needs /root/reference at build time)")
    gold = os.path.join(ROOT, "tests", "golden", "tsk")
    digests = {}
    for tag, exe in (("ref", ref_full), ("ours", ours_full)):
        d = tmp_path / ("db_" + tag)
        d.mkdir()
        for name in ("T0_k31.ht", "T1_k31.ht", "S9_k31.ht"):
            shutil.copy(os.path.join(gold, name), str(d / name))
        r = subprocess.run([exe, "-k", "31", "-T", os.path.join(gold, "targets.txt"), "-D", str(d) + "/", "-O", os.path.join(gold, "g0.fa"),
                            "-R", str(tmp_path / ("res_" + tag)), "--tsk"], cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 255, (tag, r.stderr[-1500:])
        assert "The database will be recovered from saved targets-specific data." in r.stderr, (tag, r.stderr[-1500:])
        assert "Central Hashtable successfully stored in disk." in r.stderr, (tag, r.stderr[-1500:])
        base = str(d / "db_central_k31_t3_s1610612741_m0.tsk")
        digests[tag] = {ext: hashlib.sha256(open(base + ext, "rb").read()).hexdigest() for ext in (".sz", ".ky", ".lb")}
    assert digests["ref"] == digests["ours"]
    want = dict(l.split()[::-1] for l in open(os.path.join(gold, "db_sha256.txt")))
    assert digests["ours"] == want
