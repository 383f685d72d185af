"""BASELINE configs[0]: 3-genome toy database + 1 000 x 150 bp synthetic reads, k = 31, through the
plumbing the reference uses for it (set_targets.sh -> getTargetsDef -> targets.txt -> cuCLARK builds
the database from the target files -> classification -> CSV).

  * the targets definition is pinned by the reference's own getTargetsDef: tests/golden/config0/ holds
    the input table and what the reference tool printed for it (tests/golden/make_golden.py);
  * the database builder is pinned against the reference's builder in tests/test_ref_host.py;
  * the CSV is compared with the oracle's (800 reads sampled from the genomes
    with 1 % substitutions, 100 uniform random, 50 with one N, 50 shorter than k)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from jn_cuclark_amd import synth
from test_host_cli import _build, _expected_csv, BIN

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config0")


@pytest.mark.parametrize("rank", [0, 1])
def test_targets_definition_equals_the_reference_tools_output(tmp_path, rank):
    _build()
    r = subprocess.run([os.path.join(BIN, "getTargetsDef"), os.path.join(GOLD, "custom.fileToTaxIDs"), str(rank)],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLD, "targets_rank%d.txt" % rank)).read()
    assert open(str(tmp_path / "files_excluded.txt")).read() == open(os.path.join(GOLD, "files_excluded.txt")).read()
    r = subprocess.run([os.path.join(BIN, "getTargetsDef"), os.path.join(GOLD, "custom.fileToTaxIDs"), "6"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to recognize the rank" in r.stderr


def test_targets_definition_tool_is_clean_under_the_sanitizers(tmp_path):
    """host/getTargetsDef.cc with -fsanitize=address,undefined: both ranks and the refused rank, same bytes, no report"""
    exe = str(tmp_path / "getTargetsDef_san")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-std=c++17", "-o", exe,
                        os.path.join(root, "jn_cuclark_amd", "host", "getTargetsDef.cc")], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr
    for rank in (0, 1):
        r = subprocess.run([exe, os.path.join(GOLD, "custom.fileToTaxIDs"), str(rank)], cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
        assert r.stdout == open(os.path.join(GOLD, "targets_rank%d.txt" % rank)).read()
    r = subprocess.run([exe, os.path.join(GOLD, "custom.fileToTaxIDs"), "6"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def config0_genomes():
    """3 genomes x 100 000 nt uniform ACGT (seeds 1, 2, 3); genome 1 carries a 5 000-nt block of
    genome 0 (non-discriminative region)"""
    g = [synth.random_codes(s, 100_000) for s in (1, 2, 3)]
    g[1] = g[1].copy()
    g[1][40_000:45_000] = g[0][10_000:15_000]
    return g


def config0_reads(genomes, k):
    codes, _ = synth.sample_reads(genomes, 800, 150, seed=11)
    names = [b"smp%d" % i for i in range(800)]
    seqs = [synth.codes_to_ascii(c) for c in codes]
    for i in range(100):
        names.append(b"rnd%d" % i)
        seqs.append(synth.codes_to_ascii(synth.random_codes(1100 + i, 150)))
    withn, _ = synth.sample_reads(genomes, 50, 150, seed=12)
    pos = synth.rand_u64(13, 50) % np.uint64(150)
    for i in range(50):
        s = bytearray(synth.codes_to_ascii(withn[i]))
        s[int(pos[i])] = ord("N")
        names.append(b"n%d" % i)
        seqs.append(bytes(s))
    short, _ = synth.sample_reads(genomes, 50, 150, seed=14)
    for i in range(50):
        names.append(b"short%d" % i)
        seqs.append(synth.codes_to_ascii(short[i][: 1 + (i * 7) % (k - 1)]))
    return names, seqs


@pytest.mark.gpu
def test_config0_end_to_end_csv_equals_oracle(oracle, tmp_path):
    _build()
    k, ht = 31, 1610612741
    genomes = config0_genomes()
    (tmp_path / "Custom").mkdir()
    for i, g in enumerate(genomes):
        (tmp_path / "Custom" / ("genome%d.fa" % i)).write_bytes(
            synth.fasta_text([b"NC_%06d.1 synthetic genome %d" % (i, i)], [synth.codes_to_ascii(g)], width=80))
    # targets.txt exactly as the reference's getTargetsDef printed it (relative paths: run from tmp_path)
    shutil.copy(os.path.join(GOLD, "targets_rank0.txt"), str(tmp_path / "targets.txt"))
    labels = [ln.split("\t")[1] for ln in open(os.path.join(GOLD, "targets_rank0.txt")).read().split("\n") if ln]
    assert labels == ["562", "1280", "1423"]
    names, seqs = config0_reads(genomes, k)
    text = synth.fastq_text(names, seqs)
    (tmp_path / "reads.fq").write_bytes(text)
    (tmp_path / "db").mkdir()
    r = subprocess.run([os.path.join(BIN, "cuCLARK"), "-k", "31", "-T", "targets.txt", "-D", "db/", "-O", "reads.fq",
                        "-R", "out", "-n", "4", "-b", "5"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    base = str(tmp_path / "db" / ("db_central_k31_t3_s%d_m0.tsk" % ht))
    # the shared block is not discriminative: 2 x (5000 - 30) k-mers of it are in no target's set
    n_db = os.path.getsize(base + ".lb") // 2
    assert 3 * (100_000 - 30) - 2 * 5000 - 100 < n_db <= 3 * (100_000 - 30) - 2 * (5000 - 30)
    want, _ = _expected_csv(oracle, text, k, ht, base, ["NA"] + labels, maxhits=15)
    got = open(str(tmp_path / "out.csv")).read()
    assert got == want
    rows = got.split("\n")[1:-1]
    assert len(rows) == 1000
    assigned = [ln.split(",")[2] for ln in rows]
    assert sum(a != "NA" for a in assigned[:800]) > 700 and all(a == "NA" for a in assigned[800:900])
    assert all(a == "NA" for a in assigned[950:])                       # shorter than k: no k-mer
    os.remove(base + ".sz")
