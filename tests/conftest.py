import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# a fatal signal in ANY native thread of the product library prints that thread's stack before Python's faulthandler
# reports the Python side (cap_amd/csrc/capgpu.hip: CAPGPU_SEGV_BACKTRACE; round 6 met one unexplained crash in ~12 suite runs)
os.environ.setdefault("CAPGPU_SEGV_BACKTRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cg():
    """The product library bound to cuda:0.  GPU tests fail loudly (not skip) when it cannot initialise."""
    from cap_amd import lib
    lib.init(0)
    return lib


@pytest.fixture(scope="session")
def tau():
    from oracle import bn254 as bn
    return bn.SplitMix64(0xCA9).field(bn.R)
