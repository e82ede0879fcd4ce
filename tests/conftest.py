import os
import sys

import pytest

# No torch here: the GPU suite runs on the HIP / RCCL runtime libapd_hip.so is linked against (/opt/rocm), with device
# buffers from the library itself (apd_device_alloc).  The one test that shares a stream and tensors with torch
# (tests/test_gpu_runtime.py) does so in a child process, where torch's bundled runtime is loaded first.

# Every allocation, event record and launch inside the library checks that the calling thread is bound to the context it works on
# (include/apd.h, APD_DEBUG_AFFINITY): on for the whole suite -- and, through the environment, for the bench.py / harness child
# processes the tests start -- so that the several-ranks-on-one-GPU rehearsals of tests/test_gpu_multi.py can see a missing bind.
os.environ.setdefault("APD_DEBUG_AFFINITY", "1")

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
