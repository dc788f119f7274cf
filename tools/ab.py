#!/usr/bin/env python3
"""Interleaved A/B timing of CSR plans that differ in their flags (and, with the experiments build, in
plan-time environment switches) on one matrix in one process; every variant's y is compared with the
first one's.

    python tools/ab.py --matrix synthetic:queen  base=0  nowin=0x800  "tiles16=0;SPMV_HIP_SEGWIN_TILES=16"

A variant is NAME=FLAGS[;ENV=VALUE...]; the environment entries are set while its plan is built
(libspmv_hip_experiments.so reads them at plan time; the product library ignores them).  The pseudo entry
OUT=1 times the variant as y_out = y_in + A x between two vectors that swap after every launch.
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
    ap.add_argument("--matrix", required=True, help="file or synthetic: spec (host library)")
    ap.add_argument("--expand-symmetric", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--experiments", action="store_true", help="load libspmv_hip_experiments.so")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    if args.experiments:
        os.environ["SPMV_HIP_EXPERIMENTS"] = "1"
    import torch
    from spmv_amd import capi, hostapi, synth

    if args.matrix.startswith("delaunay:"):  # delaunay:<points>,<unknowns per node>[,seed[,rcm|random]] (python generator, scipy)
        q = args.matrix[9:].split(",")
        rows, cols, p, c, v = synth.delaunay_mesh(int(q[0]), int(q[1]), seed=int(q[2]) if len(q) > 2 else 1, order=q[3] if len(q) > 3 else "rcm")
    else:
        A = hostapi.load(args.matrix, "csr", expand_symmetric=args.expand_symmetric)
        rows, cols, p, c, v = A.rows, A.cols, A.row_ptr, A.column_index, A.value
    nnz = int(p[-1])
    nbytes = synth.csr_bytes(rows, cols, nnz)
    dev = torch.device("cuda:0")
    tp, tc, tv = (torch.from_numpy(np.asarray(t)).to(dev) for t in (p, c, v))
    tx = torch.from_numpy(synth.x_vector(cols, "uniform", seed=12345)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    plans, ys = {}, {}
    out_of_place = set()
    for spec in args.variants:
        name, rest = spec.split("=", 1)
        parts = rest.split(";")
        flags = int(parts[0], 0)
        env = dict(e.split("=", 1) for e in parts[1:])
        if env.pop("OUT", None):  # y_out = y_in + A x into a second vector, the two swapped after every launch
            out_of_place.add(name)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
        if not (flags & capi.FLAG_NO_INDEX_COMPRESSION):
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        plans[name] = plan
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys[name] = ty.cpu().numpy()
        del ty
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    ty2 = torch.zeros(rows, dtype=torch.float64, device=dev) if out_of_place else None
    times = {k: [] for k in plans}
    for rnd in range(args.rounds + 1):
        for k, plan in plans.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if k in out_of_place:
                a, b = ty.data_ptr(), ty2.data_ptr()
                for _ in range(args.reps):
                    plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), a, b, stream)
                    a, b = b, a
            else:
                for _ in range(args.reps):
                    plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[k].append(e0.elapsed_time(e1) / args.reps * 1e3)
    first = next(iter(ys))
    ref = ys[first]
    scale = max(float(np.max(np.abs(ref))), 1e-300)
    print("matrix %s rows %d nnz %d (%.1f/row) algorithmic bytes %.3f GB" % (args.matrix, rows, nnz, nnz / max(1, rows), nbytes / 1e9))
    out = {}
    for k, t in times.items():
        med, mn = float(np.median(t)), float(np.min(t))
        info = plans[k].info()
        diff = float(np.max(np.abs(ys[k] - ref))) / scale
        same = bool(np.array_equal(ys[k].view(np.uint64), ref.view(np.uint64)))
        out[k] = {"us_median": round(med, 2), "us_min": round(mn, 2), "frac_algorithmic": round(nbytes / med / 1e3 / 8000, 4),
                  "frac_streamed": round(info["streamed_bytes"] / med / 1e3 / 8000, 4), "gflops": round(2 * nnz / med / 1e3, 1),
                  "vs_first_max_rel": diff, "vs_first_bitexact": same, "plan": info}
        print("%-20s median %9.2f us  min %9.2f us  algorithmic %.3f  streamed %.3f  %7.1f GFLOP/s  vs %s: %.1e%s  tiles %d blockwin %d segwin %d (%d slots) shifted %d narrow %d panels %d dict %d" % (
            k, med, mn, out[k]["frac_algorithmic"], out[k]["frac_streamed"], out[k]["gflops"], first, diff, " (bit-exact)" if same else "",
            info["row_blocks"], info["blockwin_tiles"], info.get("segwin_tiles", 0), info.get("segwin_slots", 0), info["shifted_tiles"],
            info["narrow_tiles"], info["panel_tiles"], info["indexed_values"]))
    print(json.dumps({"matrix": args.matrix, "rows": rows, "nnz": nnz, "results": out}))


if __name__ == "__main__":
    main()
