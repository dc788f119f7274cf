#!/usr/bin/env python3
"""Column panels: what does the width of a panel's x slice cost?  4 M rows x 24 uniformly random columns, the column count varied
(x = 32 / 16 / 8 / 4 MB: panels of 4 / 2 / 1 / 0.5 MB in each XCD's 4 MB L2), default plan (values read).

    python tools/panel_width_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    import torch
    from spmv_amd import capi, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows, k = 4000000, 24
    for cols in (4000000, 2000000, 1000000, 500000):
        _, _, p, c, v = synth.random_uniform(rows, cols, k, seed=5)
        nnz = int(p[-1])
        tp, tc, tv = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v))
        tx = torch.from_numpy(synth.x_vector(cols, seed=3)).to(dev)
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        for flags, name in ((capi.FLAG_NO_VALUE_INDEX, "default"), (capi.FLAG_NO_VALUE_INDEX | capi.FLAG_NO_COLUMN_PANELS, "no panels")):
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            info = plan.info()
            ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
            best = None
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    plan.spmv(*ptrs, stream)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 10 * 1e3
                if rnd:
                    best = us if best is None else min(best, us)
            print("cols %8d (x %5.1f MB) %-10s: %7.1f us  %6.1f G gathers/s  frac(8d) %.3f  panel tiles %d of %d" % (
                cols, cols * 8 / 1e6, name, best, nnz / best / 1e3, synth.csr_bytes(rows, cols, nnz) / (best * 1e-6) / 8e12,
                info["panel_tiles"], info["row_blocks"]), flush=True)
            plan.close()
        del tp, tc, tv, tx, ty
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
