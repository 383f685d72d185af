"""CPU suite, part 3: the C-ABI library loads and exports every symbol include/mc_api.h
declares.  No compute calls: there is no GPU in the CPU test environment."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="mc_api.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from jn_cuclark_amd import _lib
    bound = sorted(n for n, _, _ in _lib.SYMBOLS)
    assert bound == _declared_symbols()


def test_library_exports_every_declared_symbol():
    from jn_cuclark_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.library_path())
    for name in _declared_symbols() + _declared_symbols("mc_group.h"):
        assert hasattr(lib, name), name
    assert len(_declared_symbols("mc_group.h")) == 16          # (round 4: + the four mc_group_text_* entry points, + mc_group_set_cycle)
    assert _lib.load_library().mc_api_version() == 4


def test_builder_symbols_exported():
    txt = open(os.path.join(ROOT, "include", "mc_build.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(mc_builder_[a-z_0-9]+)\s*\(", txt)))
    assert len(names) == 7
    from jn_cuclark_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.library_path())
    for n in names:
        assert hasattr(lib, n), n


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "jn_cuclark_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".cc")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in src and "clark_oracle" not in src and "liboracle" not in src, f


def test_fails_loudly_without_gpu():
    """With no device the library must refuse, not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from jn_cuclark_amd import CuClarkDB, McError
    with pytest.raises(McError):
        CuClarkDB(k=31, numBatches=1, numTargets=3)


GB = 1 << 30
CARD = 288 * 10**9          # one MI355X


@pytest.mark.parametrize("n_keys", [6_450_000_000, 20_000_000_000, 32_000_000_000, 50_000_000_000, 90_000_000_000])
def test_index_plan_respects_the_line_space_and_the_budget(n_keys):
    """The loader's arithmetic without a device (mc_index_plan = choose_fill / lines_per_part / min_parts of
    csrc/mc_api.hip).  Round 2 computed n_keys / fill GLOBAL lines in 32 bits: 32e9 k-mers over 8 cards came out at
    fill 4 = 8e9 lines = "too many lines".  A part now has its own 32-bit line space (mc_minimizer.hpp part_of):
    every table whose share fits a card's HBM must come out with a plan.  Reference behaviour: a table is cut
    into as many parts as memory dictates and runs whatever its size (src/CuClarkDB.cu:516-559)."""
    from jn_cuclark_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    for n_parts in range(1, 9):
        p = _lib.index_plan(n_keys, n_parts, CARD)
        assert 3.0 <= p["fill"] <= 12.0
        share = n_keys / n_parts
        if p["fits"]:
            assert 0 < p["lines_per_part"] < 0xFFFFFFF0
            assert p["bytes_per_part"] + 16 * GB <= CARD
            # the lines really hold the part's k-mers at that fill (+ the 1024 lines of slack, rounded up)
            assert p["lines_per_part"] * n_parts * p["fill"] >= n_keys
            assert p["lines_per_part"] * p["fill"] <= share + 1024 * p["fill"] + p["fill"] * n_parts
        else:
            # does not fit even at 12 per line: more than ~12e9 k-mers per card
            assert share > 11e9, (n_keys, n_parts, p)
        # the sparsest fill is taken: one step sparser would not fit (or we are at 3)
        if p["fits"] and p["fill"] > 3.0:
            q = _lib.index_plan(n_keys, n_parts, CARD - 1)      # monotone in the budget
            assert q["fill"] >= p["fill"]
    p8 = _lib.index_plan(n_keys, 8, CARD)
    assert p8["fits"] == 1, "8 cards hold up to 90e9 k-mers"
    # the smallest part count: one less must not fit at fill 10, and the count itself does
    s = p8["min_parts"]
    assert 1 <= s <= 8
    assert _lib.index_plan(n_keys, s, CARD)["fits"] == 1
    if s > 1:
        lower = _lib.index_plan(n_keys, s - 1, CARD)
        assert lower["fits"] == 0 or lower["fill"] > 10.0


def test_index_plan_examples_of_the_review():
    """configs[3]/[4] scale: 32e9 k-mers.  8 cards: fill 3, 1.33e9 lines each; one card cannot hold it; the
    line limit itself (2^32 - 16 lines per part) is only reached beyond what any card holds."""
    from jn_cuclark_amd import _lib
    p = _lib.index_plan(32_000_000_000, 8, CARD)
    assert p["fits"] == 1 and p["fill"] == 3.0 and p["lines_per_part"] == (int(32_000_000_000 / 3.0) + 1024 + 7) // 8
    assert p["min_parts"] == 3                                        # 32e9 / 3 = 10.7e9 per card at 10 per line
    assert _lib.index_plan(32_000_000_000, 1, CARD)["fits"] == 0
    assert _lib.index_plan(32_000_000_000, 4096, CARD)["fits"] == 1    # part 0 of 4096: a few MB
    # a budget with room for more than 2^32 lines: the fill is clamped by the line space, not refused
    huge = _lib.index_plan(40_000_000_000, 1, 4000 * 10**9)
    assert huge["fits"] == 1 and huge["fill"] > 9.0 and huge["lines_per_part"] < 0xFFFFFFF0
    assert _lib.index_plan(6_450_000_000, 1, CARD)["min_parts"] == 1


def test_one_copy_of_the_hip_runtime_in_a_python_process():
    """A fresh process that touches the package BEFORE it imports torch (mc_index_plan needs no device) and imports torch afterwards
    maps ONE libamdhip64 -- torch's: jn_cuclark_amd._lib imports torch before it loads libmcclark.so.  The other order gave the process
    /opt/rocm's copy next to torch's own, and on the GPU box the copy that started second found no device (mc_open: "no HIP device
    visible", profiles/r04_two_hip_runtimes.txt).  The mapping can be checked without a card."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from jn_cuclark_amd import _lib\n"
        "assert _lib.index_plan(6450000000, 1, 288 * 10**9)['fits'] == 1\n"
        "import torch\n"
        "copies = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln})\n"
        "print('\\n'.join(copies))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    copies = [c for c in r.stdout.strip().split("\n") if c]
    assert len(copies) == 1 and os.sep + "torch" + os.sep in copies[0], copies
