"""jn_cuclark_amd -- MI355X-native classification core for cuCLARK-format databases.

The product is the HIP library ``libmcclark.so`` (C ABI: ``include/mc_api.h``) plus the
C++ host driver under ``host/``.  This Python package is plumbing for tests and
``bench.py``: a ctypes binding of the C ABI (``_lib``), a mirror of the reference's
``CuClarkDB`` interface (``classifier``) and synthetic workload generators (``synth``).
It never imports anything from ``oracle/``.
"""
from ._lib import load_library, library_path, McError  # noqa: F401
from .classifier import CuClarkDB  # noqa: F401

__all__ = ["load_library", "library_path", "McError", "CuClarkDB"]
