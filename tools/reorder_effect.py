#!/usr/bin/env python3
"""What the reordering suffixes buy on the GPU (VERDICT r03 task 4): launch time of `y += A x` for a matrix as
it is, after `__RCM` (the reference's reverse Cuthill-McKee) and after `__GPX<n>` (EXTENSION: the repo's k-way partition order; `__GP<n>` without METIS reorders nothing, as in the reference), with the
time the reordering itself took on the host.

    python tools/reorder_effect.py [SPEC ...]        (default: a scrambled band of 27 per row and uniformly random columns)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    import torch
    from spmv_amd import capi, hostapi, synth
    specs = sys.argv[1:] or ["synthetic:scrambled:2000000,13", "synthetic:random:1000000,24,3", "synthetic:webbase"]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    print("| matrix | order | load + reorder s | bandwidth | mean abs(i-j) | launch us | GFLOP/s | frac algorithmic | 16-bit tiles / tiles | column panels |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for spec in specs:
        for suffix in ("", "__RCM", "__GPX64", "__GPX4096"):
            t0 = time.perf_counter()
            A = hostapi.load(spec + suffix, "csr")
            t_load = time.perf_counter() - t0
            rows, cols, nnz = A.rows, A.cols, A.num_entries
            p, c, v = A.row_ptr, A.column_index, A.value
            r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(p))
            dist = np.abs(r - c)
            tp, tc, tv = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v))
            tx = torch.from_numpy(synth.x_vector(cols, seed=1)).to(dev)
            ty = torch.zeros(rows, dtype=torch.float64, device=dev)
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
            ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
            for _ in range(5):
                plan.spmv(*ptrs, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            info = plan.info()
            alg = synth.csr_bytes(rows, cols, nnz)
            print("| %s | %s | %.1f | %d | %.0f | %.1f | %.0f | %.3f | %d / %d | %s |" % (
                spec, suffix or "as generated", t_load, int(dist.max()), float(dist.mean()), us, 2.0 * nnz / us / 1e3,
                alg / (us * 1e-6) / 8e12, info["narrow_tiles"], info["row_blocks"], "yes" if info["panel_tiles"] else "no"), flush=True)
            plan.close()
            A.close()
            del tp, tc, tv, tx, ty
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
