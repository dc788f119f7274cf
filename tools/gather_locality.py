#!/usr/bin/env python3
"""How much of the random-gather cost is x residency?  The same 2 M x 24 random rows with the
columns drawn from 256 K, 512 K, 1 M, 2 M and 4 M columns (x = 2 ... 32 MB): if a small x (fits one
XCD's 4 MB L2) is much faster, column panels per XCD would pay for scattered matrices."""
import os
import sys
os.environ.setdefault("SPMV_HIP_EXPERIMENTS", "1")  # --force needs the experiments build

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    import torch
    from spmv_amd import capi, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows = 2000000
    per_row = int(sys.argv[sys.argv.index("--per-row") + 1]) if "--per-row" in sys.argv else 24
    sizes = (2097152,) if "--per-row" in sys.argv else (65536, 262144, 524288, 1048576, 2097152, 4194304)
    for cols in sizes:
        _, _, p, c, v = synth.random_uniform(rows, cols, per_row, seed=3)
        tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
        tx = torch.from_numpy(synth.x_vector(cols)).to(dev)
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, 0x4000 if "--force" in sys.argv else 0)
        plan.compress(tc.data_ptr(), stream)
        if "--panels" in sys.argv:
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        for _ in range(3):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        info = plan.info()
        print("cols %8d (x %5.1f MB)  nnz %9d  %8.1f us  %6.1f GFLOP/s  narrow tiles %d of %d  panel tiles %d" % (
            cols, cols * 8 / 1e6, int(p[-1]), us, 2 * int(p[-1]) / us / 1e3, info["narrow_tiles"], info["row_blocks"],
            info["panel_tiles"]))
        plan.close()
        del tp, tc, tv, tx, ty


if __name__ == "__main__":
    main()
