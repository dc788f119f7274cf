#!/usr/bin/env python3
"""Soak of the group tiles (csr_blocktile.hpp: rows in groups of 2 or 4 with the same columns) on the GPU, every result against the
oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c); kept under tests/ because it uses the checker library,
not collected by pytest.

Random meshes: grids of 6 ... 40 nodes per edge, 2, 4 or 8 unknowns per node, 0 ... 60 % of the links jittered, a random share of the
groups damaged (a foreign column in one row; a row one entry short; a row one entry long), the matrix cut at a random first row (the grid
of groups then starts anywhere), sometimes a stretch of short or scattered rows in the middle.  Per matrix: default plan (contract
tolerance), SPMV_HIP_FLAG_NO_BLOCK_TILES (tolerance), exact order (bit for bit), another column array and y_out != y_in (the same bits
as the default plan), two accumulating runs.

    python3 tests/soak_groups.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def damaged_mesh(seed):
    from spmv_amd import synth
    rng = np.random.default_rng(seed)
    d = int(rng.choice([2, 2, 4, 4, 8]))
    grid = tuple(int(g) for g in rng.integers(6, 41 if d < 8 else 21, size=3))
    while np.prod(grid) * d * 27 * d > 40_000_000:
        grid = tuple(max(6, g * 3 // 4) for g in grid)
    rows, cols, p, c, v = synth.mesh_dofs(grid, d, seed=seed, jitter_share=float(rng.choice([0.0, 0.3, 0.6])))
    lens = np.diff(p).astype(np.int64)
    share = float(rng.choice([0.0, 0.0, 0.01, 0.05, 0.1]))
    keep = np.ones(len(c), dtype=bool)
    c = c.copy()
    extra_rows, extra_cols = [], []
    damaged = 0
    if share > 0:
        for n in rng.choice(rows // d, size=max(1, int(share * rows / d)), replace=False):
            r = int(n) * d + int(rng.integers(0, d))
            kind = int(rng.integers(0, 3))
            k_last = p[r + 1] - 1
            if kind == 0 and c[k_last] + 1 < cols:       # a foreign column (same length)
                c[k_last] += 1
            elif kind == 1 and lens[r] > 2:               # one entry short
                keep[k_last] = False
            elif c[k_last] + 2 < cols:                    # one entry long
                extra_rows.append(r)
                extra_cols.append(c[k_last] + 2)
            damaged += 1
    rr = np.repeat(np.arange(rows), lens)
    rr, cc, vv = rr[keep], c[keep].astype(np.int64), v[keep]
    if extra_rows:
        rr = np.concatenate([rr, np.array(extra_rows)])
        cc = np.concatenate([cc, np.array(extra_cols, dtype=np.int64)])
        vv = np.concatenate([vv, rng.uniform(-1, 1, size=len(extra_rows))])
    # sometimes a stretch of other rows in the middle: short rows (a lane per row), or scattered long ones
    if rng.random() < 0.3:
        at = int(rng.integers(0, rows))
        n_other = int(rng.integers(1, 200))
        other_len = int(rng.choice([3, 9, 40]))
        rr = np.where(rr >= at, rr + n_other, rr)
        orow = np.repeat(np.arange(at, at + n_other), other_len)
        ocol = np.concatenate([np.sort(rng.choice(cols, size=other_len, replace=False)) for _ in range(n_other)])
        rr = np.concatenate([rr, orow])
        cc = np.concatenate([cc, ocol])
        vv = np.concatenate([vv, rng.uniform(-1, 1, size=len(orow))])
        rows += n_other
    order = np.lexsort((cc, rr))
    rr, cc, vv = rr[order], cc[order], vv[order]
    first = int(rng.integers(0, 2 * d))  # the matrix starts at another row: a rank's row block
    sel = rr >= first
    rr, cc, vv = rr[sel] - first, cc[sel], vv[sel]
    rows -= first
    p2 = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(rr, minlength=rows), out=p2[1:])
    what = "seed %d mesh %s, %d per node, %d groups damaged, first row %d, %d rows" % (seed, "x".join(map(str, grid)), d, damaged, first, rows)
    return rows, cols, p2.astype(np.int32), cc.astype(np.int32), vv, what, d


def main():
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_grouptiles import run_plan, same_bits
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    group_tiles = tiles = multiplies = with_groups = block_tiles = 0
    for seed in range(first, first + count):
        rows, cols, p, c, v, what, d = damaged_mesh(seed)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        nterms = 27 * d + d + 2
        got, info = run_plan(rows, cols, p, c, v, x, y0)
        assert_close(got, want, scale, what=what, nterms=nterms)
        got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
        assert info_n["group_tiles"] == 0
        assert_close(got_n, want, scale, what=what + ", no group tiles", nterms=nterms)
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
        assert_bitexact(got_e, want, what + ", exact order")
        got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
        if info["block_tiles"] == 0:  # (a small mesh whose row lengths are multiples of 3 may get masked 3 x 3 block tiles: another order)
            same_bits(got_c, got, what + ", other column array")
        else:
            assert_close(got_c, want, scale, what=what + ", other column array", nterms=nterms)
        got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
        same_bits(got_o, got, what + ", y_out")
        block_tiles += info["block_tiles"]
        got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=2 * nterms)
        multiplies += 7
        group_tiles += info["group_tiles"]
        tiles += info["row_blocks"]
        with_groups += info["group_tiles"] > 0
        if (seed - first) % 10 == 9:
            print("seed %d: %d multiplies; %d group tiles + %d block tiles of %d; %d of %d matrices with group tiles" % (
                seed, multiplies, group_tiles, block_tiles, tiles, with_groups, seed - first + 1), flush=True)
    assert group_tiles > 0
    print("soak ok: %d seeds, %d multiplies, none off; %d group tiles + %d block tiles of %d; %d matrices with group tiles" % (
        count, multiplies, group_tiles, block_tiles, tiles, with_groups))


if __name__ == "__main__":
    main()
