import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import cc
    cc.build()
    return cc


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY CPU emulation build of the product's HIP sources (kernel-logic checks without a GPU)."""
    d = os.path.join(ROOT, "tests", "hipemu")
    subprocess.check_call(["make", "-s", "-C", d])
    from lecturemath_amd import _lib
    return _lib.load(os.path.join(d, "liblecturemath_emu.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library on a real GPU; fails (not skips) when it is missing or no GPU is visible."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    from lecturemath_amd import _lib
    lib = _lib.load()
    assert lib.is_device_build
    return lib
