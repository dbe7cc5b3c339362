import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu_lib():
    """TEST HARNESS: the product's C-ABI sources compiled for the host against the stand-in
    hip_runtime.h (tests/host_emu).  Lets the CPU tier check solver logic + host-side argument
    handling against the oracle.  Never used by the product or by the -m gpu tests."""
    d = os.path.join(ROOT, "tests", "host_emu")
    from optimization_dynamics_amd import _lib
    if os.environ.get("OD_EMU_LIB"):          # e.g. a sanitizer build of the same sources (tools/asan_tier.sh)
        return _lib.Library(os.environ["OD_EMU_LIB"])
    subprocess.check_call(["make", "-C", d, "-j", "8"], stdout=subprocess.DEVNULL)
    return _lib.Library(os.path.join(d, "libod_emu.so"))


@pytest.fixture(scope="session")
def gpu_lib():
    """The shipped HIP library; fails loudly if it was not built."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from optimization_dynamics_amd import _lib
    return _lib.default_library()
