#!/usr/bin/env python3
"""Constant-coefficient stencils beyond 5 points: a 7-point and a 27-point operator on an n^3 grid (and 27 plain diagonals)
with the same coefficients in every row (centre 26 / 6, neighbours -1), timed with the value dictionary + constant-row
tiles (default plan) and with 8-byte values (SPMV_HIP_FLAG_NO_VALUE_INDEX); the two results are compared bit for bit.

    python tools/constant_stencil.py [--n 200]
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
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pool", type=int, default=0, help="draw every entry's value from a pool of this many (no constant rows)")
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    n = args.n
    cases = {
        "7-point %d^3" % n: [dz * n * n + dy * n + dx for dz, dy, dx in ((0, 0, 0), (0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0))],
        "27-point %d^3" % n: [dz * n * n + dy * n + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)],
        "27 diagonals": list(range(-13, 14)),
    }
    out = {}
    for name, offs in cases.items():
        rows, cols, p, c, v = synth.banded(n ** 3, offs, seed=1)
        r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(p))
        v = np.where(c == r, float(len(offs) - 1), -1.0)
        del r
        nnz = int(p[-1])
        tp, tc, tv = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v))
        tx = torch.from_numpy(synth.x_vector(cols, "uniform", seed=12345)).to(dev)
        res = {}
        ys = {}
        plans = [("dictionary + constant rows", 0), ("8-byte values", capi.FLAG_NO_VALUE_INDEX)]
        if args.pool:  # the same structure, every entry's value drawn from a pool: no constant rows
            rng = np.random.default_rng(5)
            v = rng.uniform(-1, 1, size=args.pool)[rng.integers(0, args.pool, size=nnz)]
            tv = torch.from_numpy(v).to(dev)
            plans = [("default plan", 0), ("no x window (dictionary, indexed)", capi.FLAG_NO_X_WINDOW), ("8-byte values", capi.FLAG_NO_VALUE_INDEX)]
        for label, flags in plans:
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
            ty = torch.zeros(rows, dtype=torch.float64, device=dev)
            ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
            plan.spmv(*ptrs, stream)
            torch.cuda.synchronize()
            ys[label] = ty.cpu().numpy()
            for _ in range(3):
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
                          "tiles": info["row_blocks"], "constant_row_tiles": info["value_row_tiles"], "launch_tiles": info["dictionary_launch_tiles"],
                          "dictionary": info["indexed_values"], "xwin_tiles": info["xwin_tiles"]}
            plan.close()
            del ty
        a, b = ys[plans[0][0]], ys["8-byte values"]
        same = bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))
        rel = float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))
        print("%-16s rows %9d entries %10d: %s" % (name, rows, nnz, "; ".join(
            "%s %.1f us %.0f GFLOP/s (%d tiles launched, dictionary %d, x-window tiles %d)" % (k, r["us"], r["gflops"], r["launch_tiles"] or r["tiles"], r["dictionary"], r["xwin_tiles"]) for k, r in res.items())),
            "bit-identical" if same else "max rel diff %.1e" % rel)
        out[name] = {"rows": rows, "nnz": nnz, "results": res, "bit_identical": same, "max_rel_diff": rel}
        del tp, tc, tv, tx
    print(json.dumps(out))


if __name__ == "__main__":
    main()
