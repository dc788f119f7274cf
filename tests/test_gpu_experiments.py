"""Two kernel families were measured slower than the paths they were meant to replace and were retired from the product
library in round 5 (csrc/internal.hpp): a lane group per row for stencil rows of 17 ... 64 entries
(tools/experiments/csr_rowgroup.hpp) and hub columns for web graphs (tools/experiments/csr_hub.hpp).  The product library refuses their flags; their parity
tests (against the oracle's CSR loop, src/matrix/csr-matrix-spmv.cpp:21-33) still run -- in a child process that loads
libspmv_hip_experiments.so."""
import os
import subprocess
import sys

import numpy as np
import pytest

from spmv_amd import capi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_refuses_the_retired_flags():
    p = np.arange(0, 11, dtype=np.int32)
    for bit in (capi.FLAG_ROW_GROUPS, capi.FLAG_HUB_COLUMNS):
        with pytest.raises(capi.SpmvHipError) as e:
            capi.CsrPlan(10, 10, p, capi.CSR_AUTO, 0, bit)
        assert e.value.code == capi.ERR_INVALID and "unknown flag" in str(e.value)
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("spmv_hip.h", "spmv_hip_tuning.h", "spmv_hip_plan.h"))
    assert "define SPMV_HIP_FLAG_ROW_GROUPS" not in text and "define SPMV_HIP_FLAG_HUB_COLUMNS" not in text


@pytest.mark.parametrize("name", ["exp_gpu_rowgroup.py", "exp_gpu_hub.py"])
def test_retired_kernels_in_the_experiments_library(name):
    env = dict(os.environ, SPMV_HIP_EXPERIMENTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "experiments", name), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
