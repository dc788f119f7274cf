"""Tests of kernel families that live in libspmv_hip_experiments.so only (retired from the product library: csrc/internal.hpp).
The files are named exp_*.py so that `pytest tests/` does not collect them into a process that has loaded the product library;
tests/test_gpu_experiments.py runs them in a child process with SPMV_HIP_EXPERIMENTS=1."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "spmv-cache-trace_amd", "python"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")


@pytest.fixture(autouse=True)
def _experiments_library_only():
    if os.environ.get("SPMV_HIP_EXPERIMENTS", "") in ("", "0"):
        pytest.skip("tests/experiments needs SPMV_HIP_EXPERIMENTS=1 (libspmv_hip_experiments.so)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()
