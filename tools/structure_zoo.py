#!/usr/bin/env python3
"""A zoo of matrix structures that are NOT among BASELINE's configurations, each through the default plan (Level 2, values read):
launch time, SURVEY 8(d)'s fraction (algorithmic bytes / time / 8 TB/s), streamed bytes of the triad, and the tile classes the plan
chose -- a look-out for cliffs between the classes (a structure one step away from a stand-in that falls to half its speed).

    python tools/structure_zoo.py [name ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def stencil3d(n, offsets, seed=1):
    """n^3 grid, one unknown per cell, the given (dz, dy, dx) neighbours where they exist."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    z, y, x = z.ravel(), y.ravel(), x.ravel()
    cols, ok = [], []
    for dz, dy, dx in sorted(offsets):
        zz, yy, xx = z + dz, y + dy, x + dx
        good = (zz >= 0) & (zz < n) & (yy >= 0) & (yy < n) & (xx >= 0) & (xx < n)
        cols.append(np.where(good, (zz * n + yy) * n + xx, 0))
        ok.append(good)
    cols, ok = np.stack(cols, axis=1), np.stack(ok, axis=1)
    lens = ok.sum(axis=1)
    p = np.zeros(n ** 3 + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = cols[ok].astype(np.int32)
    return n ** 3, n ** 3, p.astype(np.int32), c, rng.uniform(-1, 1, size=len(c))


def stencil3d_2d(n, seed=1):
    """5-point stencil on an n x n grid (lines of n cells)."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    y, x = y.ravel(), x.ravel()
    cols, ok = [], []
    for dy, dx in [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)]:
        yy, xx = y + dy, x + dx
        good = (yy >= 0) & (yy < n) & (xx >= 0) & (xx < n)
        cols.append(np.where(good, yy * n + xx, 0))
        ok.append(good)
    cols, ok = np.stack(cols, axis=1), np.stack(ok, axis=1)
    p = np.zeros(n * n + 1, dtype=np.int64)
    np.cumsum(ok.sum(axis=1), out=p[1:])
    c = cols[ok].astype(np.int32)
    return n * n, n * n, p.astype(np.int32), c, rng.uniform(-1, 1, size=len(c))


def zoo():
    from spmv_amd import synth
    seven = [(0, 0, 0), (0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)]
    nineteen = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if abs(a) + abs(b) + abs(c) <= 2]
    out = {
        "stencil7_256^3": lambda: stencil3d(256, seven),
        "stencil19_160^3": lambda: stencil3d(160, nineteen),
        "stencil27_160^3": lambda: synth.stencil27_like(160, 160, 160),
        "stencil27_128^3_real": lambda: stencil3d(128, [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]),  # HPCG's matrix
        "stencil7_100^3": lambda: stencil3d(100, seven),
        "stencil5_2d_1500^2": lambda: stencil3d_2d(1500),
        "band9": lambda: synth.banded(8000000, list(range(-4, 5)), seed=2),
        "band17": lambda: synth.banded(6000000, list(range(-8, 9)), seed=2),
        "band18": lambda: synth.banded(6000000, list(range(-8, 10)), seed=2),
        "band65": lambda: synth.banded(2000000, list(range(-32, 33)), seed=2),
        "band129": lambda: synth.banded(1000000, list(range(-64, 65)), seed=2),
        "band161": lambda: synth.banded(800000, list(range(-80, 81)), seed=2),
        "far_diagonals_11": lambda: synth.banded(6000000, [-2000000, -70000, -300, -2, -1, 0, 1, 2, 300, 70000, 2000000], seed=5),
    }

    def ragged(rows, lo, hi, reach, seed):
        rng = np.random.default_rng(seed)
        lens = rng.integers(lo, hi + 1, size=rows)
        p = np.zeros(rows + 1, dtype=np.int64)
        np.cumsum(lens, out=p[1:])
        base = np.repeat(np.arange(rows, dtype=np.int64), lens)
        c = np.clip(base + rng.integers(-reach, reach + 1, size=int(p[-1])), 0, rows - 1)
        # ascending within a row
        order = np.lexsort((c, base))
        c = c[order].astype(np.int32)
        return rows, rows, p.astype(np.int32), c, rng.uniform(-1, 1, size=len(c))
    def queen_dof_major():
        # the queen-like mesh (3 unknowns per node) numbered dof by dof -- all x unknowns, then all y, then all z -- instead of node by
        # node: the same matrix, no 3 x 3 blocks in consecutive rows, three clusters of columns a third of the matrix apart
        from spmv_amd import hostapi
        import scipy.sparse as sp
        Q = hostapi.load("synthetic:queen:100,80,70", "csr")
        A = sp.csr_matrix((np.array(Q.value), np.array(Q.column_index), np.array(Q.row_ptr)), shape=(Q.rows, Q.cols))
        Q.close()
        n = A.shape[0] // 3
        perm = np.concatenate([np.arange(n) * 3 + d for d in range(3)])
        B = A[perm][:, perm].tocsr()
        B.sort_indices()
        return B.shape[0], B.shape[1], B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data
    out["queen_dof_major"] = queen_dof_major

    def queen_dof_major_rcm():
        # ... and what reverse Cuthill-McKee makes of it (scipy's here; the product's is the "__RCM" path suffix): the three unknowns of
        # a node have the same neighbours, so they come out next to each other again -- groups of three rows with the same columns
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        rows, cols, p, c, v = queen_dof_major()
        A = sp.csr_matrix((v, c, p), shape=(rows, cols))
        order = reverse_cuthill_mckee(A, symmetric_mode=True)
        B = A[order][:, order].tocsr()
        B.sort_indices()
        return rows, cols, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data
    out["queen_dof_major_rcm"] = queen_dof_major_rcm
    out["mesh_1dof_160x120x100"] = lambda: synth.mesh_dofs((160, 120, 100), 1)
    out["mesh_2dof_100x80x70"] = lambda: synth.mesh_dofs((100, 80, 70), 2)
    out["mesh_3dof_100x80x70"] = lambda: synth.mesh_dofs((100, 80, 70), 3)
    out["mesh_4dof_100x80x60"] = lambda: synth.mesh_dofs((100, 80, 60), 4)
    out["mesh_5dof_80x60x60"] = lambda: synth.mesh_dofs((80, 60, 60), 5)
    out["mesh_6dof_80x60x50"] = lambda: synth.mesh_dofs((80, 60, 50), 6)
    # round 6: unstructured meshes with VARIABLE valence -- Delaunay tetrahedra of random points, numbered by
    # reverse Cuthill-McKee -- a structure that no generator of the host library shaped (4 ... ~40 neighbours per node, no two
    # neighbourhoods alike); 3 unknowns per node = the honest twin of Queen_4147, 1 per node = the commonest SuiteSparse shape
    out["delaunay_3dof_700k"] = lambda: synth.delaunay_mesh(700000, 3, seed=1)     # 104 M entries, ~49.5 per row
    def tril(make):
        # the stored lower triangle of a symmetric matrix (what the reference multiplies from a `symmetric` Matrix Market file)
        import scipy.sparse as sp
        rows, cols, p, c, v = make()
        A = sp.tril(sp.csr_matrix((v, c, p), shape=(rows, cols)), format="csr")
        A.sort_indices()
        return rows, cols, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    out["delaunay_3dof_700k_tril"] = lambda: tril(lambda: synth.delaunay_mesh(700000, 3, seed=1))  # ... as a symmetric file stores it: 53 M entries
    def tril_dropped(make, drop):
        # ... with a share of its off-diagonal entries missing (explicit zeros the assembly left out)
        import scipy.sparse as sp
        rows, cols, p, c, v = tril(make)
        r = np.repeat(np.arange(rows), np.diff(p))
        keep = (c == r) | (np.random.default_rng(1).random(len(c)) >= drop)
        B = sp.csr_matrix((v[keep], (r[keep], c[keep])), shape=(rows, cols))
        B.sort_indices()
        return rows, cols, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data
    out["delaunay_3dof_700k_tril_drop2pct"] = lambda: tril_dropped(lambda: synth.delaunay_mesh(700000, 3, seed=1), 0.02)
    out["delaunay_3dof_700k_tril_drop10pct"] = lambda: tril_dropped(lambda: synth.delaunay_mesh(700000, 3, seed=1), 0.10)
    out["delaunay_6dof_250k"] = lambda: synth.delaunay_mesh(250000, 6, seed=6)     # shells: 6 x 6 blocks = 3 x 3 blocks, 148 M entries, ~99 per row
    out["delaunay_1dof_2M"] = lambda: synth.delaunay_mesh(2000000, 1, seed=2)      # 33 M entries, ~16.5 per row
    out["delaunay_1dof_6M"] = lambda: synth.delaunay_mesh(6200000, 1, seed=2)      # 102 M entries (minutes of qhull)
    out["delaunay_2dof_1M"] = lambda: synth.delaunay_mesh(1000000, 2, seed=3)      # 66 M entries, ~33 per row
    out["delaunay_2d_1dof_8M"] = lambda: synth.delaunay_mesh(8000000, 1, seed=4, dim=2)  # triangles: 7 per row, 56 M entries
    out["delaunay_3dof_random_order"] = lambda: synth.delaunay_mesh(300000, 3, seed=5, order="random")  # the same mesh numbered at random
    out["ragged_1-8_near"] = lambda: ragged(8000000, 1, 8, 2000, 3)
    out["ragged_4-40_near"] = lambda: ragged(3000000, 4, 40, 5000, 4)
    out["ragged_20-100_near"] = lambda: ragged(1200000, 20, 100, 20000, 5)
    out["ragged_4-40_far"] = lambda: ragged(3000000, 4, 40, 1500000, 6)
    return out


def main():
    import torch
    from spmv_amd import capi, synth
    import perf_floor
    names = sys.argv[1:]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    triad = perf_floor.measure_triad()
    print("triad %.0f GB/s" % triad)
    for name, make in zoo().items():
        if names and name not in names:
            continue
        t0 = time.perf_counter()
        rows, cols, p, c, v = make()
        nnz = int(p[-1])
        tp, tc, tv = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v))
        tx = torch.from_numpy(synth.x_vector(cols, seed=3)).to(dev)
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, capi.FLAG_NO_VALUE_INDEX | int(os.environ.get("ZOO_FLAGS", "0"), 0))
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        info = plan.info()
        ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
        best = None
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            if rnd > 0:
                best = us if best is None else min(best, us)
        alg = synth.csr_bytes(rows, cols, nnz)
        print("%-22s %9d rows %6.1f/row  %8.1f us  frac(8d) %.3f  streamed/triad %.3f  tiles %d: narrow %d shifted %d xwin %d blockwin %d segwin %d panels %d block %d group %d multi %d long %d balanced %d stencil-masked %d  (setup %.0f s)"
              % (name, rows, nnz / rows, best, alg / (best * 1e-6) / 8e12, info["streamed_bytes"] / (best * 1e-6) / 1e9 / triad, info["row_blocks"],
                 info["narrow_tiles"], info["shifted_tiles"], info["xwin_tiles"], info["blockwin_tiles"], info["segwin_tiles"], info["panel_tiles"],
                 info["block_tiles"], info["group_tiles"], info["multi_window_tiles"], info["long_blocks"], info["balanced"], info["stencil_mask_tiles"], time.perf_counter() - t0), flush=True)
        plan.close()
        del tp, tc, tv, tx, ty
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
