#!/usr/bin/env python3
"""Time the rank-local multiply of a row-partitioned matrix on ONE GPU: what each of G ranks
would do per step (tools/slice_timing.py [--grid 4096]).  The collective is not part of this."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--flags", type=lambda t: int(t, 0), default=0, help="plan flags (capi.FLAG_*)")
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, synth, partition
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    n = args.grid
    rows = n * n
    x = torch.from_numpy(synth.x_vector(rows, seed=12345)).to(dev)
    for world in (1, 2, 4, 8):
        rank = world // 2
        b, e = partition.row_range(rows, rank, world)
        nr, cols, p, c, v = synth.poisson2d(n, b, e)
        tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
        y = torch.zeros(nr, dtype=torch.float64, device=dev)
        plan = capi.CsrPlan(nr, cols, p, capi.CSR_AUTO, 0, args.flags)
        plan.compress(tc.data_ptr(), stream)
        for _ in range(5):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), x.data_ptr(), y.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), x.data_ptr(), y.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        nnz = int(p[-1])
        by = synth.csr_bytes(nr, cols, nnz)
        print("G=%d  rank %d rows %9d nnz %9d  %8.2f us  %7.1f GB/s  (%.1f GFLOP/s per rank; gathered y segment %.1f MB, "
              "received per rank %.1f MB)" % (world, rank, nr, nnz, us, by / us / 1e3, 2 * nnz / us / 1e3,
                                               8 * partition.row_chunk(rows, world) / 1e6,
                                               8 * partition.row_chunk(rows, world) * (world - 1) / 1e6))
        plan.close()
        del tp, tc, tv, y


if __name__ == "__main__":
    main()
