#!/usr/bin/env python3
"""Soak of the tile classifier and every CSR kernel path on adversarial matrices (tests/test_gpu_parity.py's
patchwork_matrix: shifted blocks of any row length, perturbed ones, equally long unrelated rows, ragged / empty / very
long rows), many seeds, with the values as they are and drawn from dictionaries of 1, 2, 5 and 100 values, under the
plan flags that select different paths.  Every result is compared with the oracle (bit for bit under EXACT_ORDER,
1e-10 relative otherwise).  Kept under tests/ because it uses the checker library (oracle/); not collected by pytest.
    python3 tests/soak_patchwork.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_parity import patchwork_matrix
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    checked = 0
    for seed in range(first, first + count):
        rows, cols, p, c, v0 = patchwork_matrix(seed)
        rng = np.random.default_rng(seed)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        for pool in (0, 1, 2, 5, 100):
            v = v0 if pool == 0 else rng.uniform(-1, 1, size=pool)[rng.integers(0, pool, size=len(v0))]
            want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=1)
            scale = abs_products(rows, p, c, v, x) + np.abs(y0)
            tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
            for f in (0, capi.FLAG_ROWS128, capi.FLAG_EXACT_ORDER, capi.FLAG_ROWS64 | capi.FLAG_NO_X_WINDOW, capi.FLAG_XCD_REMAP):
                plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, f)
                plan.compress(tc.data_ptr(), stream)
                plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
                if pool:
                    plan.index_values(tv.data_ptr(), stream)
                ty = torch.from_numpy(y0.copy()).to(dev)
                plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
                torch.cuda.synchronize()
                got = ty.cpu().numpy()
                what = "seed %d pool %d flags %x %r" % (seed, pool, f, plan.info())
                plan.close()
                if f & capi.FLAG_EXACT_ORDER:
                    assert_bitexact(got, want, what)
                else:
                    assert_close(got, want, scale, what=what)
                checked += 1
        if (seed - first) % 10 == 9:
            print("seeds %d..%d ok (%d multiplies checked)" % (first, seed, checked), flush=True)
    print("soak ok: %d multiplies" % checked)


if __name__ == "__main__":
    main()
