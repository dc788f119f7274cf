#!/usr/bin/env python3
"""Where is the floor for a cache-resident problem?  Times (one event pair around 200 back-to-back
launches) the STREAM triad at several sizes -- 65 MB of traffic is what a webbase-like SpMV moves --
next to the CSR algorithms on the webbase-like matrix with and without its remote links."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def timeit(fn, reps=200):
    import torch
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    import torch
    from spmv_amd import capi, hostapi, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    lib = capi.load()
    for n in (1 << 17, 1 << 20, 2720000, 1 << 23, 1 << 26):
        a, b, c = (torch.ones(n, dtype=torch.float64, device=dev) for _ in range(3))
        pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
        us = timeit(lambda: lib.spmv_hip_triad(n, pa, pb, pc, 3.1, stream))
        print("triad n=%9d  %7.1f MB  %8.2f us  %7.1f GB/s" % (n, 24e-6 * n, us, 24.0 * n / us / 1e3))
        del a, b, c
    for spec in ("synthetic:webbase", "synthetic:webbase:1000005,3105536,4700,100", "synthetic:powerlaw"):
        A = hostapi.load(spec)
        tp, tc, tv = (torch.from_numpy(np.asarray(t)).to(dev) for t in (A.row_ptr, A.column_index, A.value))
        tx = torch.from_numpy(synth.x_vector(A.cols)).to(dev)
        ty = torch.zeros(A.rows, dtype=torch.float64, device=dev)
        ty2 = torch.zeros(A.rows, dtype=torch.float64, device=dev)
        for name, algo, lanes, flags in (("balanced", capi.CSR_AUTO, 0, 0), ("row-owned tiles", capi.CSR_AUTO, 0, capi.FLAG_NO_BALANCED_TILES),
                                         ("scalar", capi.CSR_SCALAR, 0, 0), ("vector2", capi.CSR_VECTOR, 2, 0), ("vector4", capi.CSR_VECTOR, 4, 0),
                                         ("adaptive", capi.CSR_ADAPTIVE, 0, 0)):
            plan = capi.CsrPlan(A.rows, A.cols, A.row_ptr, algo, lanes, flags)
            plan.compress(tc.data_ptr(), stream)
            h = plan.h
            args = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
            us = timeit(lambda: lib.spmv_hip_csr_spmv(h, args[0], args[1], args[2], args[3], args[4], stream))
            line = "%-40s %-16s %8.2f us  tiles %d" % (spec[10:], name, us, plan.info()["row_blocks"])
            if name == "balanced":
                y2 = ty2.data_ptr()
                us2 = timeit(lambda: lib.spmv_hip_csr_spmv_out(h, args[0], args[1], args[2], args[3], args[4], y2, stream))
                line += "   (y_in != y_out: %.2f us)" % us2
            print(line)
            plan.close()
        A.close()


if __name__ == "__main__":
    main()
