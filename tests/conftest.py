import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "default_precision: runs WITHOUT the suite-wide TT_PRECISION=bf16 (checks what an unchanged reference call gets)")
    # the CPU oracle mostly runs small shapes: on a 256-core GPU host torch's default (one thread per core) makes every
    # small matmul a 256-way barrier (a 3 s test took 260 s); the full-depth oracle raises the count for itself
    try:
        import torch

        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(autouse=True)
def _restore_torch_threads():
    """A test may raise torch's CPU thread count for a large oracle matmul (256 on the GPU host); left in place it makes
    every later small CPU op a 256-way barrier (the x3 tests took 100 s instead of 1 s behind the full-size scan test)."""
    try:
        import torch
    except Exception:  # noqa: BLE001
        yield
        return
    before = torch.get_num_threads()
    yield
    if torch.get_num_threads() != before:
        torch.set_num_threads(before)


@pytest.fixture(autouse=True)
def _suite_names_bf16(request, monkeypatch):
    """The product's default precision is the reference's own (fp32 semantics: ``precision.DEFAULT_MODE``, round 4).  The
    suites of rounds 1-3 build the plugin surfaces without naming a dtype and hold them to bf16-grade bounds, i.e. they test the
    bf16 mode: they NAME it here, through the process setting (``TT_PRECISION=bf16``).  Tests marked ``default_precision`` run
    without the variable and check what the unchanged reference calls get."""
    if request.node.get_closest_marker("default_precision") is None:
        monkeypatch.setenv("TT_PRECISION", "bf16")
    else:
        monkeypatch.delenv("TT_PRECISION", raising=False)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    """Path of libtt_hip.so, building it if this checkout has not been built yet."""
    from tensor_truth_amd import _lib

    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.lib_path()


@pytest.fixture(scope="session")
def diag_lib_env(built_lib):
    """Environment of a CHILD process that loads the diagnostic build of the library (csrc `make DIAG=1` ->
    libtt_hip_diag.so): the only build in which the kernels' A/B switches exist -- the product library reads no environment
    variable (tests/test_lib_abi.py).  A/B tests run their "other side" in such a child and compare bits with the product."""
    import subprocess

    from tensor_truth_amd import _lib

    path = os.path.join(os.path.dirname(_lib.lib_path()), "libtt_hip_diag.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tensor-truth_amd", "csrc"), "DIAG=1", "-j4"], check=True)
    return dict(os.environ, TT_LIB_NAME="libtt_hip_diag.so")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test selected but no HIP device is visible")
    return torch.device("cuda:0")

TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:      # helper modules next to the tests (rank_checks.py)
    sys.path.insert(0, TESTS)
