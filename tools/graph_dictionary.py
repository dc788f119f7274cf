#!/usr/bin/env python3
"""A web graph as a PATTERN matrix (every stored entry 1.0) or with weights 1 / out-degree (capped to 100 distinct values): the
balanced-tile kernel with the value dictionary (one index byte per entry) against 8-byte values.

    python tools/graph_dictionary.py [--matrix synthetic:webbase]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="synthetic:webbase")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, hostapi, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c = A.rows, A.cols, np.asarray(A.row_ptr), np.asarray(A.column_index)
    nnz = int(p[-1])
    lens = np.diff(p)
    cases = {"pattern (all ones)": np.ones(nnz), "1 / min(out-degree, 100)": np.repeat(1.0 / np.minimum(np.maximum(lens, 1), 100), lens)}
    tp, tc = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c))
    tx = torch.from_numpy(synth.x_vector(cols, "uniform", seed=12345)).to(dev)
    out = {}
    for name, v in cases.items():
        tv = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
        res, ys = {}, {}
        for label, flags in (("dictionary", 0), ("8-byte values", capi.FLAG_NO_VALUE_INDEX)):
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
            ty = torch.zeros(rows, dtype=torch.float64, device=dev)
            ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
            plan.spmv(*ptrs, stream)
            torch.cuda.synchronize()
            ys[label] = ty.cpu().numpy()
            for _ in range(5):
                plan.spmv(*ptrs, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            info = plan.info()
            res[label] = {"us": round(us, 2), "gflops": round(2 * nnz / us / 1e3, 1), "streamed_bytes": info["streamed_bytes"],
                          "dictionary": info["indexed_values"], "balanced": info["balanced"]}
            plan.close()
        same = bool(np.array_equal(ys["dictionary"].view(np.uint64), ys["8-byte values"].view(np.uint64)))
        rel = float(np.max(np.abs(ys["dictionary"] - ys["8-byte values"])) / max(1e-300, np.max(np.abs(ys["8-byte values"]))))
        print("%-26s %s: %s  %s" % (name, args.matrix, "; ".join("%s %.2f us (%d values, %.1f MB streamed)" % (
            k, r["us"], r["dictionary"], r["streamed_bytes"] / 1e6) for k, r in res.items()), "bit-identical" if same else "max rel diff %.1e (chunks of long rows meet in atomics: the order varies from launch to launch)" % rel))
        out[name] = {"results": res, "bit_identical": same, "max_rel_diff": rel}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
