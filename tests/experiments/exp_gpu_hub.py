"""Hub columns (tools/experiments/csr_hub.hpp) -- a kernel family that was measured slower than the balanced tiles it extends (26.6 against
23.9 us on the web graph) and lives in libspmv_hip_experiments.so only since round 5.  Run in a process of its
own with SPMV_HIP_EXPERIMENTS=1 (tests/test_gpu_experiments.py does that); the product library refuses the flag."""
import numpy as np
import pytest

from helpers import assert_bitexact, assert_close, abs_products
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_hub_columns_of_a_graph_matrix(oracle):
    """(Opt-in, SPMV_HIP_FLAG_HUB_COLUMNS.)  A web-like matrix whose x exceeds an XCD's L2: the columns many rows refer to become hubs -- the plan keeps its own column
    stream, every multiply first copies the hubs' x entries into a dense array, the tiles read them from there.  Same bits as
    the plan without hubs; x may change between multiplies (the copy is made by every multiply); y_out != y_in; with a value
    dictionary too; another column array at multiply time uses nothing of it."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(31)
    rows, cols = 400000, 700000
    lens = np.where(rng.random(rows) < 0.02, rng.integers(17, 500, rows), rng.integers(1, 6, rows))  # (no row above 512 entries: those are split and meet in atomics, whose order is not reproducible)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    Z = int(p[-1])
    r = np.repeat(np.arange(rows), lens)
    popular = rng.choice(cols, size=20000, replace=False)
    local = np.clip(r * cols // rows + rng.integers(-3000, 3001, size=Z), 0, cols - 1)
    c = np.where(rng.random(Z) < 0.3, popular[(rng.pareto(1.2, size=Z) * 40).astype(np.int64) % len(popular)], local)
    order = np.lexsort((c, r))
    c = np.ascontiguousarray(c[order].astype(np.int32))
    p = p.astype(np.int32)
    assert Z >= (1 << 20)
    for values in ("hashed", "ones"):
        v = rng.uniform(-1.0, 1.0, size=Z) if values == "hashed" else np.ones(Z)
        x1, x2 = synth.x_vector(cols, seed=3), synth.x_vector(cols, seed=33)
        y0 = synth.x_vector(rows, seed=4)
        tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
        tx1, tx2 = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
        got = {}
        for flags in (capi.FLAG_HUB_COLUMNS, 0):
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
            info = plan.info()
            assert info["balanced"] == 1, info
            if flags:
                assert 1000 < info["hub_columns"] <= 60000 and info["hub_entries"] > 0.1 * Z, info
                assert info["indexed_values"] == (1 if values == "ones" else 0)
            else:
                assert info["hub_columns"] == 0
            ty = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx1.data_ptr(), ty.data_ptr(), stream)
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx2.data_ptr(), ty.data_ptr(), stream)  # another x: the dense copy follows
            tout = torch.full((rows,), np.nan, dtype=torch.float64, device=dev)
            plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx1.data_ptr(), ty.data_ptr(), tout.data_ptr(), stream)
            tc2 = tc.clone()  # the same columns at another address: the plan's derived streams must not be used
            tz = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc2.data_ptr(), tv.data_ptr(), tx2.data_ptr(), tz.data_ptr(), stream)
            torch.cuda.synchronize()
            got[flags] = (ty.cpu().numpy(), tout.cpu().numpy(), tz.cpu().numpy())
            plan.close()
        for a, b, what in zip(got[capi.FLAG_HUB_COLUMNS], got[0], ("two multiplies, two x", "y_out", "other column array")):
            assert_bitexact(a, b, "%s, %s: hubs against no hubs" % (values, what))
        want = oracle.csr_spmv(rows, p, c, v, x1, y=y0, num_threads=4)
        want = oracle.csr_spmv(rows, p, c, v, x2, y=want, num_threads=4)
        scale = abs_products(rows, p, c, v, x1) + abs_products(rows, p, c, v, x2) + np.abs(y0)
        assert_close(got[0][0], want, scale, what=values + ", two multiplies")
        assert_close(got[0][1], oracle.csr_spmv(rows, p, c, v, x1, y=want, num_threads=4), scale + abs_products(rows, p, c, v, x1), what=values + ", y_out")
