import os
import sys

import pytest

try:                # torch bundles its own HIP runtime: load it BEFORE libapd_hip.so pulls in /opt/rocm's,
    import torch    # so that one process never holds two HIP runtimes (only matters for tests using both)
except Exception:   # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libapd_oracle.so), built on demand.  Checker only."""
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def apd():
    """The product library through its C ABI (ctypes), no compute call made here."""
    from audio_pattern_discovery_amd import _lib
    return _lib
