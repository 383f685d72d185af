import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """torch is imported (and, on a box with a card, the HIP runtime started) before any test runs: torch must be the one that
    loads libamdhip64 -- it ships a copy of its own, and a process in which libmcclark.so (RUNPATH /opt/rocm) came first ends up
    with BOTH copies, of which the one that starts second finds no device.  Seen on the GPU box with tests/test_host_cli.py run before
    the first in-process GPU test (it loads the library for mc_index_plan): mc_open said "no HIP device visible"
    (profiles/r04_two_hip_runtimes.txt).  jn_cuclark_amd._lib imports torch before it loads the library for the same reason; this
    hook keeps the suite independent of the order of its files whatever a test loads by hand.  Without a card (the build container)
    it is `import torch` and nothing else."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:          # (no torch, no card: the gpu tests say so themselves)
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/).  Test infrastructure: the checker, never the product."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
