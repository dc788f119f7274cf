#!/usr/bin/env python3
"""Interleaved A/B timing of the CSR kernel variants on one matrix, one process
(perf deltas only count when measured this way: cdna_hip_programming.md rule 24).

    python tools/kernel_sweep.py [--workload poisson2d --grid 4096 --rounds 5 --reps 20]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
# the A/B and ablation variants live in libspmv_hip_experiments.so only (same sources, -DSPMV_HIP_EXPERIMENTS)
os.environ.setdefault("SPMV_HIP_EXPERIMENTS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="poisson2d")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", default="")
    ap.add_argument("--triad-variants", action="store_true")
    ap.add_argument("--formats", default="csr", help="comma list of csr,coo,coo_shuffled,ell")
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, synth

    keep = None
    if args.workload.startswith("synthetic:"):
        # the generators of host/matrix/synthetic.cpp: synthetic:queen, synthetic:kkt[:n], synthetic:webbase, synthetic:powerlaw
        from spmv_amd import hostapi
        keep = hostapi.load(args.workload, "csr")
        rows, cols, p, c, v = keep.rows, keep.cols, keep.row_ptr, keep.column_index, keep.value
    elif args.workload == "poisson2d":
        rows, cols, p, c, v = synth.poisson2d(args.grid)
    elif args.workload == "stencil27":
        rows, cols, p, c, v = synth.stencil27_like(160, 160, 160)
    elif args.workload == "random":
        rows, cols, p, c, v = synth.random_uniform(2000000, 2000000, 24, seed=3)
    elif args.workload in ("scrambled", "scrambled_rcm"):
        # a 27-point stencil whose unknowns were numbered at random, and the same matrix after reverse
        # Cuthill-McKee (scipy's here, to avoid a text round trip; the product's RCM is
        # host/matrix/matrix-reorder.cpp behind the "__RCM" path suffix): what reordering buys the gather
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        rows, cols, p, c, v = synth.stencil27_like(100, 100, 100)
        perm = np.random.default_rng(11).permutation(rows)
        A = sp.csr_matrix((v, c, p), shape=(rows, cols))[perm][:, perm].tocsr()
        if args.workload == "scrambled_rcm":
            order = reverse_cuthill_mckee(A, symmetric_mode=True)
            A = A[order][:, order].tocsr()
        A.sort_indices()
        p, c, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    elif args.workload == "queen":  # ~80 entries/row, banded-ish
        rows, cols, p, c, v = synth.banded(2000000, list(range(-40, 41)), seed=5)
    elif args.workload == "longrows_dense":
        # the same shape with consecutive columns (a dense band): the long-row path without gather effects
        rows = cols = 60000
        L = 1500
        cm = (np.arange(rows, dtype=np.int64)[:, None] + np.arange(L, dtype=np.int64)[None, :]) % cols
        cm.sort(axis=1)
        p = (np.arange(rows + 1, dtype=np.int64) * L).astype(np.int32)
        c = cm.reshape(-1).astype(np.int32)
        v = np.random.default_rng(9).uniform(-1.0, 1.0, size=c.shape[0])
        del cm
    elif args.workload == "longrows":
        # 60000 rows of 1500 entries each (longer than a tile, shorter than the split threshold),
        # columns within +-30000 of the diagonal
        rng = np.random.default_rng(9)
        rows = cols = 60000
        L = 1500
        cm = np.sort((np.arange(rows, dtype=np.int64)[:, None] + rng.integers(-30000, 30000, size=(rows, L))) % cols, axis=1)
        p = (np.arange(rows + 1, dtype=np.int64) * L).astype(np.int32)
        c = cm.reshape(-1).astype(np.int32)
        v = rng.uniform(-1.0, 1.0, size=c.shape[0])
        del cm
    elif args.workload == "fem_mesh":
        # a jittered 3-D mesh: 80 x 80 x 78 nodes, 3 unknowns per node, every node coupled to 27 nodes
        # near its 27 grid neighbours (each moved by up to 2 nodes): rows of a node share their
        # columns, neighbouring nodes share most of theirs -- the locality of a real finite-element
        # matrix -- but no two rows are shifted copies of each other
        rng = np.random.default_rng(8)
        nx, ny, nz = 80, 80, 78
        nodes = nx * ny * nz
        offs = np.array([dz * nx * ny + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], dtype=np.int64)
        nb = np.arange(nodes, dtype=np.int64)[:, None] + offs[None, :] + rng.integers(-2, 3, size=(nodes, 27))
        nb = np.clip(nb, 0, nodes - 1)
        nb.sort(axis=1)
        cm_node = (3 * nb[:, :, None] + np.arange(3, dtype=np.int64)[None, None, :]).reshape(nodes, 81)
        rows = cols = 3 * nodes
        cm = np.repeat(cm_node, 3, axis=0)
        cm.sort(axis=1)
        p = (np.arange(rows + 1, dtype=np.int64) * 81).astype(np.int32)
        c = cm.reshape(-1).astype(np.int32)
        v = rng.uniform(-1.0, 1.0, size=c.shape[0])
        del cm, cm_node, nb
    elif args.workload == "fem3d":
        # like "fem" below but the 27 blocks of a row lie within +-40000 of the diagonal (a 3-D mesh's
        # bandwidth): too wide for 16-bit offsets and for an LDS window, yet far from scattered
        rng = np.random.default_rng(7)
        rows = cols = 1500000
        starts = rng.integers(-40000, 40000, size=(rows, 27), dtype=np.int64) + np.arange(rows, dtype=np.int64)[:, None]
        cm = (starts[:, :, None] + np.arange(3, dtype=np.int64)[None, None, :]).reshape(rows, 81)
        cm = np.clip(cm, 0, cols - 1)
        cm.sort(axis=1)
        p = (np.arange(rows + 1, dtype=np.int64) * 81).astype(np.int32)
        c = cm.reshape(-1).astype(np.int32)
        v = rng.uniform(-1.0, 1.0, size=c.shape[0])
    elif args.workload == "fem":
        # an unstructured band, closer to a finite-element matrix (Queen_4147) than the perfect band
        # above: 27 blocks of 3 consecutive columns per row, placed at random within +-3000 of the diagonal
        rng = np.random.default_rng(6)
        rows = cols = 1500000
        starts = rng.integers(-3000, 3000, size=(rows, 27), dtype=np.int64) + np.arange(rows, dtype=np.int64)[:, None]
        cm = (starts[:, :, None] + np.arange(3, dtype=np.int64)[None, None, :]).reshape(rows, 81)
        cm = np.clip(cm, 0, cols - 1)
        cm.sort(axis=1)
        p = (np.arange(rows + 1, dtype=np.int64) * 81).astype(np.int32)
        c = cm.reshape(-1).astype(np.int32)
        v = rng.uniform(-1.0, 1.0, size=c.shape[0])
    else:
        rows, cols, p, c, v = synth.powerlaw(1000005, 1000005, seed=4)
    nnz = int(p[-1])
    nbytes = synth.csr_bytes(rows, cols, nnz)
    dev = torch.device("cuda:0")
    tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
    tx = torch.from_numpy(synth.x_vector(cols)).to(dev)
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    variants = {
        "scalar": (capi.CSR_SCALAR, 0, 0),
        "vector2": (capi.CSR_VECTOR, 2, 0), "vector4": (capi.CSR_VECTOR, 4, 0),
        "vector8": (capi.CSR_VECTOR, 8, 0), "vector16": (capi.CSR_VECTOR, 16, 0),
        "vector32": (capi.CSR_VECTOR, 32, 0), "vector64": (capi.CSR_VECTOR, 64, 0),
        "adaptive": (capi.CSR_ADAPTIVE, 0, 0),
        "adaptive_xcd": (capi.CSR_ADAPTIVE, 0, capi.FLAG_XCD_REMAP),
        "adaptive_exact": (capi.CSR_ADAPTIVE, 0, capi.FLAG_EXACT_ORDER),
        "wavetile": (capi.CSR_WAVETILE, 0, 0),
        "wavetile_xcd": (capi.CSR_WAVETILE, 0, capi.FLAG_XCD_REMAP),
        "wavetile_big": (capi.CSR_WAVETILE, 0, capi.FLAG_BIG_TILE),
        "wavetile_c16": (capi.CSR_WAVETILE, 0, 0x100000),  # 0x100000: sweep-local marker = compress the plan
        "wavetile_c16_noshift": (capi.CSR_WAVETILE, 0, capi.FLAG_NO_SHIFTED_TILES | 0x100000),
        "wavetile_c16_noshift_rows128": (capi.CSR_WAVETILE, 0, capi.FLAG_NO_SHIFTED_TILES | capi.FLAG_ROWS128 | 0x100000),
        "wavetile_c16_noxwin": (capi.CSR_WAVETILE, 0, capi.FLAG_NO_X_WINDOW | 0x100000),
        "wavetile_c16_ring": (capi.CSR_WAVETILE, 0, 0x100000 | 0x400000),  # 0x400000: sweep-local marker = no segment windows (the one-ring block window instead)
        "wavetile_c16_noxwin_noshift": (capi.CSR_WAVETILE, 0, capi.FLAG_NO_X_WINDOW | capi.FLAG_NO_SHIFTED_TILES | 0x100000),
        "wavetile_c16_blockwin_simple": (capi.CSR_WAVETILE, 0, 0x2000 | 0x100000),
        "wavetile_c16_panels": (capi.CSR_WAVETILE, 0, 0x100000 | 0x200000),
        "wavetile_c16_panels_forced": (capi.CSR_WAVETILE, 0, 0x4000 | 0x100000 | 0x200000),
        "wavetile_c16_big": (capi.CSR_WAVETILE, 0, capi.FLAG_BIG_TILE | 0x100000),
        "wavetile_c16_rowptr": (capi.CSR_WAVETILE, 0, capi.FLAG_READ_ROW_PTR | 0x100000),
        "wavetile_c16_xcd": (capi.CSR_WAVETILE, 0, capi.FLAG_XCD_REMAP | 0x100000),
        "wavetile_c16_rows64_xcd": (capi.CSR_WAVETILE, 0, capi.FLAG_ROWS64 | capi.FLAG_XCD_REMAP | 0x100000),
        "wavetile_c16_rows64": (capi.CSR_WAVETILE, 0, capi.FLAG_ROWS64 | 0x100000),
        "wavetile_c16_rows128": (capi.CSR_WAVETILE, 0, capi.FLAG_ROWS128 | 0x100000),
        # timing experiments (wrong results by design): where the time of a tile goes
        "abl_noshift_no_gather": (capi.CSR_WAVETILE, 0, 0x10000 | 0x100000 | capi.FLAG_NO_SHIFTED_TILES | capi.FLAG_NO_X_WINDOW),
        "abl_noshift_no_rowsum": (capi.CSR_WAVETILE, 0, 0x20000 | 0x100000 | capi.FLAG_NO_SHIFTED_TILES | capi.FLAG_NO_X_WINDOW),
        "abl_noshift_neither": (capi.CSR_WAVETILE, 0, 0x30000 | 0x100000 | capi.FLAG_NO_SHIFTED_TILES | capi.FLAG_NO_X_WINDOW),
        "abl_no_gather": (capi.CSR_WAVETILE, 0, 0x10000 | 0x100000),
        "abl_no_rowsum": (capi.CSR_WAVETILE, 0, 0x20000 | 0x100000),
        "abl_neither": (capi.CSR_WAVETILE, 0, 0x30000 | 0x100000),
    }
    if args.variants:
        variants = {k: variants[k] for k in args.variants.split(",")}
    plans = {k: capi.CsrPlan(rows, cols, p, a, l, (f & 0x3FFFF) | (capi.FLAG_NO_SEGMENT_WINDOW if f & 0x400000 else 0))
             for k, (a, l, f) in variants.items()}  # sweep-local markers are above bit 17
    for k, (a, l, f) in variants.items():
        if f & 0x100000:
            plans[k].compress(tc.data_ptr(), stream)
        if f & 0x200000:  # sweep-local marker: column panels where the matrix qualifies
            plans[k].repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    times = {k: [] for k in plans}
    for rnd in range(args.rounds + 1):
        for k, plan in plans.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:  # round 0 is warm-up
                times[k].append(e0.elapsed_time(e1) / args.reps * 1e3)
    # empirical roofline: STREAM triad on 3 x 512 MiB (past the 256 MiB Infinity Cache)
    nt = 64 * 1024 * 1024
    ta = torch.zeros(nt, dtype=torch.float64, device=dev)
    tb = torch.ones(nt, dtype=torch.float64, device=dev)
    tcv = torch.ones(nt, dtype=torch.float64, device=dev)
    tt = []
    for rnd in range(args.rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            capi.triad(nt, ta.data_ptr(), tb.data_ptr(), tcv.data_ptr(), 3.1, stream)
        e1.record()
        torch.cuda.synchronize()
        if rnd > 0:
            tt.append(e0.elapsed_time(e1) / args.reps * 1e3)
    if args.triad_variants:
        import ctypes as C
        fn = capi.load().spmv_hip_triad_variant
        fn.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        for var in range(5):
            tv = []
            for rnd in range(args.rounds + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn(nt, ta.data_ptr(), tb.data_ptr(), tcv.data_ptr(), 3.1, stream, var)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    tv.append(e0.elapsed_time(e1) / args.reps * 1e3)
            print("triad variant %d: median %.2f us = %.1f GB/s" % (var, float(np.median(tv)), 24.0 * nt / float(np.median(tv)) / 1e3))
        # read-only and copy rates with torch's own kernels, for orientation
        for name, fn2, nbytes_ in (("torch copy", lambda: ta.copy_(tb), 16.0 * nt), ("torch sum", lambda: tb.sum(), 8.0 * nt)):
            tv = []
            for rnd in range(args.rounds + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn2()
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    tv.append(e0.elapsed_time(e1) / args.reps * 1e3)
            print("%s: median %.2f us = %.1f GB/s" % (name, float(np.median(tv)), nbytes_ / float(np.median(tv)) / 1e3))
    triad_gbs = 24.0 * nt / float(np.median(tt)) / 1e3
    print("triad (3 x 512 MiB): median %.2f us = %.1f GB/s (%.1f%% of 8 TB/s)" % (
        float(np.median(tt)), triad_gbs, triad_gbs / 80))
    del ta, tb, tcv
    print("workload %s rows %d nnz %d (%.1f/row) algorithmic bytes %.3f GB" % (
        args.workload, rows, nnz, nnz / rows, nbytes / 1e9))
    res = {}
    for k, t in times.items():
        med, mn = float(np.median(t)), float(np.min(t))
        res[k] = {"us_median": round(med, 2), "us_min": round(mn, 2), "gbs": round(nbytes / med / 1e3, 1),
                  "frac_of_8TBs": round(nbytes / med / 1e3 / 8000, 4), "gflops": round(2 * nnz / med / 1e3, 1),
                  "plan": plans[k].info()}
        print("%-16s median %9.2f us  min %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s, %.1f%% of triad)  %7.1f GFLOP/s" % (
            k, med, mn, nbytes / med / 1e3, 100 * nbytes / med / 1e3 / 8000,
            100 * nbytes / med / 1e3 / triad_gbs, 2 * nnz / med / 1e3))
    # ---- the other two formats on the same matrix ------------------------------------------
    fmts = args.formats.split(",")
    others = {}

    def time_it(fn):
        t = []
        for rnd in range(args.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                t.append(e0.elapsed_time(e1) / args.reps * 1e3)
        return float(np.median(t))

    if "coo" in fmts or "coo_shuffled" in fmts:
        ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(p.astype(np.int64)))
        cb = synth.coo_bytes(rows, cols, nnz)
        for name in ("coo", "coo_shuffled"):
            if name not in fmts:
                continue
            perm = np.arange(nnz) if name == "coo" else np.random.default_rng(1).permutation(nnz)
            tr = torch.from_numpy(ri[perm]).to(dev)
            tcc = torch.from_numpy(c[perm]).to(dev)
            tvv = torch.from_numpy(v[perm]).to(dev)
            med = time_it(lambda: capi.coo_spmv(rows, nnz, tr.data_ptr(), tcc.data_ptr(), tvv.data_ptr(),
                                                tx.data_ptr(), ty.data_ptr(), stream))
            others[name] = {"us_median": round(med, 2), "gbs": round(cb / med / 1e3, 1), "gflops": round(2 * nnz / med / 1e3, 1)}
            print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  %7.1f GFLOP/s  [bytes %.3f GB]" % (
                name, med, cb / med / 1e3, cb / med / 1e3 / 80, 2 * nnz / med / 1e3, cb / 1e9))
            capi.coo_variant(1)  # the 64-entries-per-wave kernel, same arrays, same process
            med1 = time_it(lambda: capi.coo_spmv(rows, nnz, tr.data_ptr(), tcc.data_ptr(), tvv.data_ptr(),
                                                 tx.data_ptr(), ty.data_ptr(), stream))
            capi.coo_variant(0)
            others[name + "_64_per_wave"] = {"us_median": round(med1, 2), "gbs": round(cb / med1 / 1e3, 1)}
            print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  [64 entries per wave]" % (
                name + "_64", med1, cb / med1 / 1e3, cb / med1 / 1e3 / 80))
            if name == "coo_shuffled":  # what the upload does by default: stable sort by row, once
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                capi.coo_sort_by_row(rows, nnz, tr.data_ptr(), tcc.data_ptr(), tvv.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                med = time_it(lambda: capi.coo_spmv(rows, nnz, tr.data_ptr(), tcc.data_ptr(), tvv.data_ptr(),
                                                    tx.data_ptr(), ty.data_ptr(), stream))
                others["coo_shuffled_sorted"] = {"us_median": round(med, 2), "sort_ms": round(e0.elapsed_time(e1), 2)}
                print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  after a one-time %.1f ms sort by row" % (
                    "coo_shuf_sorted", med, cb / med / 1e3, cb / med / 1e3 / 80, e0.elapsed_time(e1)))
            del tr, tcc, tvv
    if "ell" in fmts:
        L = int(np.diff(p).max())
        if rows * L < 2**31 and rows * L * 12 < 40e9:
            lens = np.diff(p.astype(np.int64))
            ec = np.zeros((rows, L), dtype=np.int32)
            ev = np.zeros((rows, L))
            mask = np.arange(L)[None, :] < lens[:, None]
            ec[mask] = c
            ev[mask] = v
            tec = torch.from_numpy(np.ascontiguousarray(ec.T)).to(dev)  # column-major
            tev = torch.from_numpy(np.ascontiguousarray(ev.T)).to(dev)
            eb = synth.ell_bytes(rows, cols, L)
            med = time_it(lambda: capi.ell_spmv(rows, L, tec.data_ptr(), tev.data_ptr(), tx.data_ptr(),
                                                ty.data_ptr(), stream))
            others["ell"] = {"us_median": round(med, 2), "row_length": L, "gbs": round(eb / med / 1e3, 1),
                             "gflops": round(2 * nnz / med / 1e3, 1)}
            print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  %7.1f GFLOP/s  [L=%d, bytes %.3f GB]" % (
                "ell", med, eb / med / 1e3, eb / med / 1e3 / 80, 2 * nnz / med / 1e3, L, eb / 1e9))
            if L <= 256:  # what the ctx does for these rows: the row-major arrays in place, as uniform wave tiles
                del tec, tev
                ter = torch.from_numpy(np.ascontiguousarray(ec)).to(dev)
                tvr = torch.from_numpy(np.ascontiguousarray(ev)).to(dev)
                pe = (np.arange(rows + 1, dtype=np.int64) * L).astype(np.int32)
                tpe = torch.from_numpy(pe).to(dev)
                # one lane per row (the reference's order) whatever the row length
                plan_e = capi.CsrPlan(rows, cols, pe, capi.CSR_WAVETILE, 0, capi.FLAG_EXACT_ORDER)
                plan_e.compress(ter.data_ptr(), stream)
                med = time_it(lambda: plan_e.spmv(tpe.data_ptr(), ter.data_ptr(), tvr.data_ptr(), tx.data_ptr(),
                                                  ty.data_ptr(), stream))
                others["ell_as_tiles"] = {"us_median": round(med, 2), "gbs": round(eb / med / 1e3, 1)}
                print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  %7.1f GFLOP/s  [row-major in place]" % (
                    "ell_as_tiles", med, eb / med / 1e3, eb / med / 1e3 / 80, 2 * nnz / med / 1e3))
                plan_n = capi.CsrPlan(rows, cols, pe, capi.CSR_WAVETILE, 0, capi.FLAG_EXACT_ORDER | capi.FLAG_NO_X_WINDOW)
                plan_n.compress(ter.data_ptr(), stream)
                if plan_n.info()["xwin_tiles"] * 2 > plan_n.info()["row_blocks"]:
                    med = time_it(lambda: plan_n.spmv(tpe.data_ptr(), ter.data_ptr(), tvr.data_ptr(), tx.data_ptr(),
                                                      ty.data_ptr(), stream))
                    others["ell_as_tiles_no_x_window"] = {"us_median": round(med, 2), "gbs": round(eb / med / 1e3, 1)}
                    print("%-16s median %9.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)  [row-major in place, x gathered]" % (
                        "ell_tiles_noxw", med, eb / med / 1e3, eb / med / 1e3 / 80))
        else:
            print("ell: skipped (rows*row_length = %d x %d too large)" % (rows, L))
    res["other_formats"] = others
    print(json.dumps({"workload": args.workload, "rows": rows, "nnz": nnz, "triad_gbs": round(triad_gbs, 1), "results": res}))


if __name__ == "__main__":
    main()
