#!/usr/bin/env python3
"""Does the launch time depend on WHERE the value array lies?  Two processes on one box gave 621 and 676 us for the same
queen-like launch; this places the values at a range of byte offsets inside one big allocation and times the same plan on
each, in one process.

    python tools/align_probe.py --matrix synthetic:queen
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="synthetic:queen")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, hostapi, synth
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c, v = A.rows, A.cols, A.row_ptr, A.column_index, A.value
    nnz = int(p[-1])
    dev = torch.device("cuda:0")
    tp, tc = (torch.from_numpy(np.asarray(t)).to(dev) for t in (p, c))
    tx = torch.from_numpy(synth.x_vector(cols, "uniform", seed=12345)).to(dev)
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    offsets = [0, 128, 256, 1024, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 65536, 16 << 20, (16 << 20) + 2048]
    slack = max(offsets) + 64
    big = torch.empty(nnz * 8 + slack, dtype=torch.uint8, device=dev)
    hv = torch.from_numpy(np.asarray(v))
    print("%s: %d rows, %d entries; big allocation at 0x%x, row_ptr 0x%x, columns 0x%x, x 0x%x, y 0x%x" % (
        args.matrix, rows, nnz, big.data_ptr(), tp.data_ptr(), tc.data_ptr(), tx.data_ptr(), ty.data_ptr()))
    for off in offsets:
        tv = big[off:off + nnz * 8].view(torch.float64)
        tv.copy_(hv)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
        times = []
        for rnd in range(args.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times.append(e0.elapsed_time(e1) / args.reps * 1e3)
        print("values at +%-9d (0x%x)  %8.2f us  (min %.2f)" % (off, tv.data_ptr(), float(np.median(times)), min(times)))
        plan.close()


if __name__ == "__main__":
    main()
