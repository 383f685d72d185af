"""The super-k-mer index (csrc/mc_skm.hpp) without a GPU: its host-callable core -- the bijective m-mer hash, the entries
of a stored k-mer (every window that reaches the smallest hash, both strands of a palindromic m-mer), record formation,
the run descriptor the kernel's front half leaves, the record match -- compiled for the host and compared with a plain
set lookup of canonical k-mers on genome-shaped data: related genomes, poly-A, short tandem repeats, both strands,
substitutions, k = 25 .. 31, through records and through the per-k-mer path of the hashed chains
(tests/cpu/skm_model.cc).  The answer must be the reference's (src/CuClarkDB.cu:1216-1247): a hit iff the canonical
k-mer is stored, counted once per position."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_super_kmer_records_answer_like_a_set_of_canonical_kmers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not here")
    exe = str(tmp_path / "skm_model")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-x", "hip",
                           os.path.join(ROOT, "tests", "cpu", "skm_model.cc"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" 0 reads differ") == 14, r.stdout


def test_the_same_under_address_and_ub_sanitizers(tmp_path):
    """the host side of csrc/mc_skm.hpp (hash, entries, records, descriptors, match) compiled with -fsanitize=address,undefined
    (host code only; the GPU pool runs no sanitizers): the same 14 cases, no report"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not here")
    exe = str(tmp_path / "skm_model_san")
    r = subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address,undefined",
                        "-Xarch_host", "-fno-omit-frame-pointer", "-x", "hip", os.path.join(ROOT, "tests", "cpu", "skm_model.cc"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    assert r.stdout.count(" 0 reads differ") == 14, r.stdout
