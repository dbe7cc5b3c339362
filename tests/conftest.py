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


# Collection order of the GPU tier under `pytest -x`: the parity tests proper first (HIP path vs oracle through the C ABI, the five
# BASELINE configs, the plain-C caller), then the remaining kernel / solver tests, then the start-sensitive example solves, and the
# environment-sensitive tests (an RCCL subprocess, hipcc on the box) last -- so that one hiccup of the environment cannot leave the
# parity tests unreached.  Within a class the file order is kept.
_ORDER_FIRST = (
    "test_gpu_parity.py::",
    "test_kernel_instantiations.py::test_config4_8192_rollouts_full_horizon",
    "test_kernel_instantiations.py::test_plumbing_config_callbacks",
    "test_ilqr.py::test_config5_rocket_projection_ilqr_as_stated",
    "test_abi.py::test_c_caller_computes_what_the_python_mirror_and_the_oracle_compute",
    "test_gpu_parity_sweep.py::",
)
_ORDER_LATE = ("test_examples.py::", "test_fd_validator.py::", "test_ilqr.py::test_rocket_example_as_shipped_nominal",
               "test_ilqr.py::test_cartpole_friction_example_on_the_device", "test_reference_golden.py::")
_ORDER_LAST = ("test_model_generator.py::", "test_comm.py::test_one_rank_allgather_over_rccl_gpu", "test_julia_shim_calls.py::test_communicator_sequence_gpu",
               "test_distributed.py::test_bench_force_dist_over_rccl_gpu")


def _order_class(nodeid):
    for rank, pats in ((0, _ORDER_FIRST), (3, _ORDER_LAST), (2, _ORDER_LATE)):
        for i, p in enumerate(pats):
            if p in nodeid:
                return (rank, i if rank == 0 else 0)
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: _order_class(it.nodeid))          # stable: file order inside a class
