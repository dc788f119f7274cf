#!/usr/bin/env python3
"""Soak of the x-window paths (csr_wavetile_kernel's XW variant: window of runs and contiguous window, the window living in the
end of the wave's product slice) and of the opt-in row-group kernel: stencil-like matrices with EVERY row length 2 ... 176, random
first-row patterns in 1 ... 9 runs or one band, a random number of short rows in front (so that tiles start at every offset
within a 16-byte quad) and behind, many seeds; default plan, SPMV_HIP_FLAG_ROW_GROUPS, SPMV_HIP_FLAG_NO_X_WINDOW and
SPMV_HIP_FLAG_EXACT_ORDER (bit-exact) against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/).
Kept under tests/ because it uses the checker library; not collected by pytest.
    python3 tests/soak_windows.py [first_seed] [count]"""
import os
import sys

import numpy as np

EXPERIMENTS = os.environ.get("SPMV_HIP_EXPERIMENTS", "") not in ("", "0")  # the row-group flag exists in libspmv_hip_experiments.so only

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def stencil_with_edges(n, offsets, front, back, rng):
    """`front` rows of 1 ... 3 entries, then n rows with columns i + offsets (clipped to the matrix), then `back` short rows."""
    offsets = np.unique(np.asarray(offsets, dtype=np.int64))
    rows = front + n + back
    lens, cols = [], []
    for r in range(front):
        k = int(rng.integers(1, 4))
        lens.append(k)
        cols.append(np.sort(rng.choice(rows, size=k, replace=False)))
    body = np.arange(front, front + n, dtype=np.int64)[:, None] + offsets[None, :]
    ok = (body >= 0) & (body < rows)
    lens.extend(ok.sum(axis=1).tolist())
    cols.append(body[ok])
    for r in range(back):
        k = int(rng.integers(1, 4))
        lens.append(k)
        cols.append(np.sort(rng.choice(rows, size=k, replace=False)))
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(np.asarray(lens, dtype=np.int64), out=p[1:])
    c = np.concatenate(cols).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    return rows, rows, p.astype(np.int32), c, v


def main():
    import torch
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    checked = windows = groups = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        length = int(rng.integers(2, 177))
        if rng.random() < 0.5:
            offsets = np.arange(length) - int(rng.integers(0, length + 1))  # one band
        else:
            nruns = int(rng.integers(1, min(9, length) + 1))
            cuts = np.sort(rng.choice(np.arange(1, length), size=nruns - 1, replace=False)) if nruns > 1 else np.array([], dtype=np.int64)
            sizes = np.diff(np.concatenate([[0], cuts, [length]]))
            starts = np.sort(rng.choice(np.arange(-3000, 3000, 70), size=nruns, replace=False))
            offsets = np.concatenate([s + np.arange(k) for s, k in zip(starts, sizes)])
        n = int(rng.integers(4000, 20000))
        rows, cols, p, c, v = stencil_with_edges(n, offsets, int(rng.integers(0, 8)), int(rng.integers(0, 8)), rng)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=1)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
        for f in (0, capi.FLAG_NO_X_WINDOW, capi.FLAG_EXACT_ORDER) + ((capi.FLAG_ROW_GROUPS,) if EXPERIMENTS else ()):
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, f | capi.FLAG_NO_VALUE_INDEX)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            info = plan.info()
            ty = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            torch.cuda.synchronize()
            got = ty.cpu().numpy()
            what = "seed %d length %d flags %x %r" % (seed, length, f, info)
            plan.close()
            if f & capi.FLAG_EXACT_ORDER:
                assert_bitexact(got, want, what)
            else:
                assert_close(got, want, scale, what=what, nterms=length)
            checked += 1
            if f == 0 and 2 * info["xwin_tiles"] > info["row_blocks"]:
                windows += 1
            if info["row_group_tiles"] > 0:
                groups += 1
        if (seed - first + 1) % 50 == 0:
            print("seeds %d..%d ok (%d multiplies; %d plans on the x-window variant, %d with row groups)" % (first, seed, checked, windows, groups), flush=True)
    print("soak ok: %d multiplies, %d plans on the x-window variant, %d with row groups" % (checked, windows, groups))


if __name__ == "__main__":
    main()
