"""CPU suite, part 3: the C-ABI library loads and exports every symbol include/mc_api.h
declares.  No compute calls: there is no GPU in the CPU test environment."""
import ctypes
import os
import re

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
    assert len(_declared_symbols("mc_group.h")) == 11
    assert _lib.load_library().mc_api_version() == 2


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
