"""The consumers of the per-read CSV: bin/getAbundance and bin/kent -a / -m / -r against what
the REFERENCE's own tools wrote for the same inputs (tests/golden/abundance/: inputs + outputs of
src/getAbundance.cc and app/kent.cpp compiled from source in the build container, tests/golden/make_golden.py).
Byte for byte; no GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")
GOLD = os.path.join(ROOT, "tests", "golden", "abundance")

CASES = {
    "plain": ["-F", "result1.csv"],
    "two_files": ["-F", "result1.csv", "result2.csv"],
    "taxonomy": ["-D", "db", "-F", "result1.csv", "result2.csv"],
    "highconf": ["--highconfidence", "-D", "db", "-F", "result1.csv"],
    "filters": ["-c", "0.8", "-g", "0.02", "-a", "5", "-D", "db", "-F", "result1.csv", "result2.csv"],
    "extended": ["-D", "db", "-F", "result_ext.csv"],
    "exports": ["-D", "db", "-F", "result1.csv", "--krona", "--mpa"],
}


def _build():
    if not all(os.path.exists(os.path.join(BIN, b)) for b in ("getAbundance", "kent")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "jn_cuclark_amd", "host"),
                        os.path.join("..", "..", "bin", "getAbundance"), os.path.join("..", "..", "bin", "kent")], check=True)


def _workdir(tmp_path):
    w = tmp_path / "w"
    shutil.copytree(GOLD, str(w))
    return w


@pytest.mark.parametrize("case", sorted(CASES))
def test_abundance_tables_equal_the_reference_tools(tmp_path, case):
    _build()
    w = _workdir(tmp_path)
    r = subprocess.run([os.path.join(BIN, "getAbundance")] + CASES[case], cwd=str(w), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLD, "out_%s.csv" % case)).read()
    if case == "exports":
        for f in ("results.krn", "results.mpa"):
            assert open(str(w / f)).read() == open(os.path.join(GOLD, "out_" + f)).read(), f


def test_abundance_tools_are_clean_under_address_and_ub_sanitizers(tmp_path):
    """host/getAbundance.cc and host/kent.cc built with -fsanitize=address,undefined (CPU build only): every golden table
    again, the merge and the report of kent, the argument errors -- the same bytes, no sanitizer report"""
    exes = {}
    for name in ("getAbundance", "kent"):
        exes[name] = str(tmp_path / (name + "_san"))
        r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-fopenmp", "-pthread",
                            "-o", exes[name], os.path.join(ROOT, "jn_cuclark_amd", "host", name + ".cc")], capture_output=True, text=True)
        if r.returncode != 0 and "sanitize" in r.stderr.lower():
            pytest.skip("no sanitizer runtime in this toolchain")
        assert r.returncode == 0, r.stderr
    w = _workdir(tmp_path)

    def run(exe, args, rc=0):
        q = subprocess.run([exes[exe]] + args, cwd=str(w), capture_output=True, text=True)
        assert q.returncode == rc, (args, q.returncode, q.stderr[-500:])
        assert "Sanitizer" not in q.stderr and "runtime error" not in q.stderr, (args, q.stderr[-900:])
        return q

    for case in sorted(CASES):
        assert run("getAbundance", CASES[case]).stdout == open(os.path.join(GOLD, "out_%s.csv" % case)).read(), case
    run("getAbundance", ["-c", "0.2", "-F", "result1.csv"], rc=1)
    run("getAbundance", ["-F", "missing.csv", "x"], rc=1)
    (w / "results").mkdir()
    for a, b, out in (("out_taxonomy.csv", "out_highconf.csv", "merged_lineage.csv"), ("out_plain.csv", "out_two_files.csv", "merged_plain.csv")):
        run("kent", ["-m", str(w / a), str(w / b), "-o", out])
        assert open(str(w / "results" / out)).read() == open(os.path.join(GOLD, out)).read()
    for src, out in (("out_taxonomy.csv", "report_taxonomy.txt"), ("merged_lineage.csv", "report_merged.txt")):
        run("kent", ["-r", str(w / src)])
        assert open(str(w / "results" / "report.txt")).read() == open(os.path.join(GOLD, out)).read()
    run("kent", ["-m", str(w / "out_plain.csv")], rc=1)


def test_abundance_argument_errors(tmp_path):
    _build()
    w = _workdir(tmp_path)
    exe = os.path.join(BIN, "getAbundance")
    for args, msg in ((["-c", "0.2", "-F", "result1.csv"], "confidence score between 0.5 and 1"),
                      (["-g", "2", "-F", "result1.csv"], "Gamma score between 0 and 1"),
                      (["-a", "101", "-F", "result1.csv"], "abundance between 0 and 100"),
                      (["--bogus", "-F", "result1.csv"], "Failed to recognize option: --bogus"),
                      (["-F", "missing.csv", "x"], "Failed to open missing.csv")):
        r = subprocess.run([exe] + args, cwd=str(w), capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)


def test_kent_abundance_merge_report(tmp_path):
    """kent -a runs the abundance tool into results/, -m adds up tables of split runs, -r writes the report"""
    _build()
    w = _workdir(tmp_path)
    (w / "results").mkdir()
    kent = os.path.join(BIN, "kent")
    env = dict(os.environ, HOME=str(w))
    r = subprocess.run([kent, "-a", str(w / "db"), "result1.csv", "-o", "ab1.csv"], cwd=str(w), capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "Abundance estimation completed successfully." in r.stdout, r.stderr
    assert open(str(w / "results" / "ab1.csv")).read() == subprocess.run(
        [os.path.join(BIN, "getAbundance"), "-D", "db", "-F", "result1.csv"], cwd=str(w), capture_output=True, text=True).stdout
    for a, b, out in (("out_taxonomy.csv", "out_highconf.csv", "merged_lineage.csv"), ("out_plain.csv", "out_two_files.csv", "merged_plain.csv")):
        r = subprocess.run([kent, "-m", str(w / a), str(w / b), "-o", out], cwd=str(w), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(str(w / "results" / out)).read() == open(os.path.join(GOLD, out)).read()
        assert r.stdout.startswith("Merged 2 abundance files (")
    for src, out in (("out_taxonomy.csv", "report_taxonomy.txt"), ("merged_lineage.csv", "report_merged.txt")):
        r = subprocess.run([kent, "-r", str(w / src)], cwd=str(w), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(str(w / "results" / "report.txt")).read() == open(os.path.join(GOLD, out)).read()
    r = subprocess.run([kent, "-m", str(w / "out_plain.csv")], cwd=str(w), capture_output=True, text=True)
    assert r.returncode == 1 and "At least 2 abundance files are required." in r.stderr
    r = subprocess.run([kent, "-c", "-R", "x"], cwd=str(w), capture_output=True, text=True)
    assert r.returncode == 1 and "Classification requires -O <fastq> or -P <file1> <file2>" in r.stderr
    r = subprocess.run([kent, "-c", "-O", "nope.fq", "-R", "x", "-b", "0"], cwd=str(w), capture_output=True, text=True)
    assert r.returncode == 1 and "Missing or invalid argument for -b" in r.stderr
