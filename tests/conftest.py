"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI export
checks.  `-m gpu` runs on the MI355X box: parity of the HIP path against the oracle,
always through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def reflib():
    import oracle_py
    if not oracle_py.RefLib.available():
        pytest.skip("oracle/_ref/libref_spmv.so not built (needs /root/reference)")
    return oracle_py.RefLib()


@pytest.fixture(scope="session")
def golden():
    import helpers
    return helpers.load_golden()
