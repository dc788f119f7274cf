#!/usr/bin/env python3
"""What Kernel::init costs through the Level-1 context (spmv_hip_create + spmv_hip_upload_csr: copy, tiles, classification,
block-tile confirmation, dictionary) for a few matrices, next to one multiply:  python tools/init_time.py [SPEC ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    from spmv_amd import capi, hostapi, synth
    specs = sys.argv[1:] or ["synthetic:poisson2d:4096", "synthetic:queen", "synthetic:queen:110,71,177,3,20,1000", "synthetic:kkt:200"]
    for spec in specs:
        M = hostapi.load(spec, "csr")
        p, c, v = np.array(M.row_ptr), np.array(M.column_index), np.array(M.value)
        rows, cols = M.rows, M.cols
        M.close()
        x = synth.x_vector(cols, seed=3)
        for rep in range(2):
            fresh = (p.copy(), c.copy(), v.copy())
            t0 = time.perf_counter()
            ctx = capi.Context(0, capi.FLAG_NO_RUN_EVENTS | int(os.environ.get("INIT_FLAGS", "0"), 0))  # INIT_FLAGS: extra SPMV_HIP_FLAG_* bits
            t1 = time.perf_counter()
            ctx.upload_csr(rows, cols, *fresh)
            t2 = time.perf_counter()
            ctx.set_x(x)
            ctx.run(1)
            t3 = time.perf_counter()
            for _ in range(20):
                ctx.run(1, sync=False)
            ctx.run(1)
            t4 = time.perf_counter()
            info = ctx.info()
            ctx.close()
            gb = (12.0 * len(c) + 4.0 * (rows + 1)) / 1e9
            print("%-40s rep %d: create %.1f ms, upload_csr %.1f ms (%.2f GB: %.1f GB/s), first multiply %.1f ms, then %.0f us per multiply; tiles %d"
                  % (spec, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, gb, gb / (t2 - t1), (t3 - t2) * 1e3, (t4 - t3) / 21 * 1e6, info["row_blocks"]), flush=True)


if __name__ == "__main__":
    main()
