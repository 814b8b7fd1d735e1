import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _cpu_workers(config):
    """The CPU suite is ~12 minutes of single-threaded work (the lock-step emulator of the HIP kernels, the torch-CPU oracle): when it
    is run as the whole `-m "not gpu"` selection, pytest-xdist is installed and nobody asked for a worker count, spread it over the
    host's cores.  Never for GPU runs (one device), never inside a worker; MDM_TEST_SERIAL=1 switches it off."""
    try:
        if os.environ.get("MDM_TEST_SERIAL") or os.environ.get("PYTEST_XDIST_WORKER"):
            return 0
        if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None):
            return 0
        if (getattr(config.option, "markexpr", "") or "").strip() != "not gpu":
            return 0
        if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
            return 0
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        return min(16, n - 1) if n >= 4 else 0
    except Exception:       # whatever goes wrong here must not keep the suite from running serially
        return 0


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    n = _cpu_workers(config)
    if n >= 2:
        config.option.numprocesses = n      # (registered after the xdist plugin, so this runs before its own cmdline hook reads it)
        if getattr(config.option, "dist", "no") in ("no", "load"):
            config.option.dist = "worksteal"    # the emulator cases last 1-170 s each: idle workers take over the queues' tails
    return None


if os.environ.get("PYTEST_XDIST_WORKER"):     # one of several workers: the oracle's torch-CPU ops must not spin up a thread pool each
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    try:
        import torch
        torch.set_num_threads(1)
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    # The GPU suite's wall time is the CHECKER's time: the torch-CPU oracle replays whole sampling loops (1000 steps at batch 1 in
    # tests/test_gpu_round2.py) and torch's default intra-op pool is every hardware thread of the GPU box (256), which at these sizes is
    # 5-10x SLOWER than a few cores (bench.py cpu_baseline: 1.4 s per 50-step loop at 16 threads, 5.3 s at 64): 423 s of a 926 s run
    # were that one replay.  Bound the pool on a GPU box; the oracle's results do not depend on the thread count (bit-identical
    # at 1 and 8 threads).
    try:
        import torch
        if torch.cuda.is_available() and not os.environ.get("PYTEST_XDIST_WORKER"):
            torch.set_num_threads(min(16, torch.get_num_threads()))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


GEMM_PATHS = {"small": {}, "small64": {"small_gemm_row_tiles": 2}, "big": {"small_gemm_max_seqs": 0}}


@pytest.fixture
def engine_options(monkeypatch):
    """Pin the kernel route of every engine built during a test: `engine_options(small_gemm_max_seqs=0)` replaces the defaults new
    engines start with (mdm_amd/_engine.py DEFAULT_OPTIONS -> include/mdm_hip.h mdm_set_option) until the test ends.  An explicit
    in-process setter: the library reads no environment variable (rounds 3-4 steered it through MDM_X3S_* read per launch)."""
    from mdm_amd import _engine

    def set_defaults(**opts):
        monkeypatch.setattr(_engine, "DEFAULT_OPTIONS", dict(opts))
    set_defaults()
    return set_defaults


@pytest.fixture(params=["small", "big"])
def gemm_path(request, engine_options):
    """Which of the two split-precision encoder GEMM kernels a small-batch test runs on: csrc/gemm_x3s.h's 32-row tiles (the
    default up to 80 sequences) or csrc/gemm_x3.h's sequence-sized tiles (small_gemm_max_seqs = 0; what large batches run).  Both
    are product code."""
    engine_options(**GEMM_PATHS[request.param])
    return request.param
