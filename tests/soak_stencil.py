#!/usr/bin/env python3
"""Soak of the masked stencil tiles (csr_stenciltile.hpp) on the GPU, every result against the oracle
(src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c); kept under tests/ because it uses the checker library,
not collected by pytest.

Random structured grids: 2 or 3 dimensions, edges of 3 ... 90 cells (short lines: almost every tile holds the end of one), a random
stencil of 3 ... 16 neighbours within a radius of 1 ... 3 cells, cells removed at random (0 ... 20 %), and three kinds of damage that a
tile must survive by NOT being taken: rows with a foreign column (no neighbour of the stencil), rows with a neighbour twice removed
(a column of the stencil but of another cell), rows cut to their diagonal.  Values random or from a small set (the value dictionary
meets the same tiles).  Rows have at most 16 entries, one lane adds a row left to right: everything is compared BIT FOR BIT --
default plan, values read (no dictionary), SPMV_HIP_FLAG_NO_SHIFTED_TILES, exact order, three accumulating runs, y_out != y_in,
another column array, and the context's CSR / COO / ELLPACK uploads.

    python3 tests/soak_stencil.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def random_grid(seed):
    """(rows, cols, p, c, v, description)"""
    rng = np.random.default_rng(seed)
    dims = int(rng.integers(2, 4))
    if dims == 2:
        shape = tuple(int(s) for s in rng.integers(20, 400, size=2))
    else:
        shape = tuple(int(s) for s in rng.integers(3, 90, size=3))
        while np.prod(shape) > 400000:
            shape = tuple(max(3, s * 3 // 4) for s in shape)
    radius = int(rng.integers(1, 4))
    cand = [tuple(int(t) for t in o) for o in np.indices((2 * radius + 1,) * dims).reshape(dims, -1).T - radius if any(o != radius)]
    k = int(rng.integers(2, 16))
    pick = rng.choice(len(cand), size=min(k, len(cand)), replace=False)
    offsets = [tuple([0] * dims)] + [cand[i] for i in pick]
    n = int(np.prod(shape))
    idx = np.indices(shape).reshape(dims, -1)
    strides = np.array([int(np.prod(shape[d + 1:])) for d in range(dims)])
    hole_share = float(rng.choice([0.0, 0.0, 0.01, 0.05, 0.2]))
    hole = rng.random(n) < hole_share
    offsets.sort(key=lambda o: int(np.dot(o, strides)))
    cols, ok = [], []
    for off in offsets:
        nb = idx + np.array(off)[:, None]
        good = np.all((nb >= 0) & (nb < np.array(shape)[:, None]), axis=0)
        c = np.where(good, (nb * strides[:, None]).sum(axis=0), 0)
        good &= ~hole[c]
        cols.append(c)
        ok.append(good)
    cols, ok = np.stack(cols, axis=1), np.stack(ok, axis=1)
    diag = [int(np.dot(o, strides)) == 0 for o in offsets].index(True)
    ok[:, diag] = True
    # two offsets may give the same column distance (a wide stencil on a narrow grid): keep the first of equal columns in a row
    for a in range(len(offsets)):
        for b in range(a + 1, len(offsets)):
            ok[:, b] &= ~(ok[:, a] & (cols[:, a] == cols[:, b]))
    # damage: rows cut to their diagonal
    cut = rng.random(n) < float(rng.choice([0.0, 0.0, 0.002, 0.02]))
    ok[cut] = False
    ok[cut, diag] = True
    # ascending columns per row (offsets sorted by distance are ascending except where the grid is narrower than the stencil)
    rows_list_c = np.where(ok, cols, np.iinfo(np.int64).max)
    order = np.argsort(rows_list_c, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    ok = np.take_along_axis(ok, order, axis=1)
    # damage: a foreign column / a stencil column of another cell in some rows (kept ascending and distinct)
    damaged = 0
    share = float(rng.choice([0.0, 0.0, 0.001, 0.01]))
    if share > 0:
        for r in np.nonzero(rng.random(n) < share)[0]:
            have = cols[r][ok[r]]
            slot = int(rng.integers(0, len(have)))
            lo = have[slot - 1] + 1 if slot > 0 else 0
            hi = have[slot + 1] - 1 if slot + 1 < len(have) else n - 1
            if hi < lo:
                continue
            new = int(rng.integers(lo, hi + 1))
            pos = np.nonzero(ok[r])[0][slot]
            cols[r, pos] = new
            damaged += 1
    lens = ok.sum(axis=1)
    p = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = cols[ok].astype(np.int32)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        v = rng.uniform(-1.0, 1.0, size=len(c))
    elif kind == 1:  # a small set of values: the value dictionary applies
        v = rng.choice(rng.uniform(-2.0, 2.0, size=int(rng.integers(2, 40))), size=len(c))
    else:  # constant coefficients per offset
        v = np.where(c == np.repeat(np.arange(n), lens), 2.0 * dims, -1.0)
    what = "seed %d grid %s, %d-point radius %d, holes %.2f, %d rows cut, %d rows damaged, values %s" % (
        seed, "x".join(str(s) for s in shape), len(offsets), radius, hole_share, int(cut.sum()), damaged, ("random", "few", "constant")[kind])
    return n, n, p.astype(np.int32), c, v, what


def main():
    from spmv_amd import capi, synth
    from helpers import assert_bitexact
    from test_gpu_stenciltiles import run_plan
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    masked = shifted = tiles = multiplies = with_masked = 0
    for seed in range(first, first + count):
        rows, cols, p, c, v, what = random_grid(seed)
        assert int(np.diff(p).max()) <= 16 and int(np.diff(p).min()) >= 1
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        got, info = run_plan(rows, cols, p, c, v, x, y0)
        assert_bitexact(got, want, what)
        got_v, info_v = run_plan(rows, cols, p, c, v, x, y0, index_values=False)
        assert_bitexact(got_v, want, what + ", values read")
        got_n, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_SHIFTED_TILES, index_values=False)
        assert_bitexact(got_n, want, what + ", no shifted tiles")
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
        assert_bitexact(got_e, want, what + ", exact order")
        got3, _ = run_plan(rows, cols, p, c, v, x, y0, runs=3)
        assert_bitexact(got3, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3), what + ", three runs")
        got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
        assert_bitexact(got_o, want, what + ", y_out")
        got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True, index_values=bool(seed & 1))
        assert_bitexact(got_c, want, what + ", other column array")
        multiplies += 9
        want0 = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
        with capi.Context(0) as ctx:
            ctx.upload_csr(rows, cols, p, c, v)
            ctx.set_x(x)
            ctx.run()
            assert_bitexact(ctx.get_y(), want0, what + ", csr upload")
            i, j, a = synth.csr_to_coordinate(rows, p, c, v)
            ctx.upload_coo(rows, cols, i - 1, j - 1, a)
            ctx.set_x(x)
            ctx.run()
            assert_bitexact(ctx.get_y(), want0, what + ", coo upload")
            rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
            assert rc == 0
            ctx.upload_ell(rows, cols, L, ec, ev)
            ctx.set_x(x)
            ctx.run()
            assert_bitexact(ctx.get_y(), oracle.ell_spmv(rows, L, ec, ev, x), what + ", ellpack upload")
        multiplies += 3
        masked += info_v["stencil_mask_tiles"]
        shifted += info_v["shifted_tiles"]
        tiles += info_v["row_blocks"]
        with_masked += info_v["stencil_mask_tiles"] > 0
        if (seed - first) % 10 == 9:
            print("seed %d: %d multiplies; %d masked stencil + %d shifted of %d tiles; %d of %d matrices with masked tiles"
                  % (seed, multiplies, masked, shifted, tiles, with_masked, seed - first + 1), flush=True)
    assert masked > 0
    print("soak ok: %d seeds, %d multiplies, none off by a bit; %d masked stencil + %d shifted of %d tiles; %d matrices with masked tiles"
          % (count, multiplies, masked, shifted, tiles, with_masked))


if __name__ == "__main__":
    main()
