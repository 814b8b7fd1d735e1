import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


GEMM_PATHS = {"small": None, "small64": ("MDM_X3S_RT", "2"), "big": ("MDM_X3S_MAX_SEQS", "0")}


@pytest.fixture(params=["small", "big"])
def gemm_path(request, monkeypatch):
    """Which of the two split-precision encoder GEMM kernels a small-batch test runs on: csrc/gemm_x3s.h's 32-row tiles (the
    default below 32 sequences) or csrc/gemm_x3.h's sequence-sized tiles (MDM_X3S_MAX_SEQS=0; what large batches run).  Both
    are product code; the environment switch is read per launch."""
    monkeypatch.delenv("MDM_X3S_MAX_SEQS", raising=False)
    monkeypatch.delenv("MDM_X3S_RT", raising=False)
    kv = GEMM_PATHS[request.param]
    if kv is not None:
        monkeypatch.setenv(*kv)
    return request.param
