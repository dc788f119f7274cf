#!/usr/bin/env python3
"""Soak of round 5's tile classes on the GPU, every result against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in
oracle/spmv_oracle.c); kept under tests/ because it uses the checker library, not collected by pytest.

  * masked block tiles (csr_blocktile.hpp): meshes of 3 x 3 blocks with a random share of entries dropped (0 ... 30 %), nodes
    with one or two unknowns at a random spacing, random neighbour counts; default plan, SPMV_HIP_FLAG_NO_MASKED_BLOCKS, exact
    order (bit-exact), two accumulating runs;
  * long rows (tile_common.hpp: long_row_sum, tile_rows_long_registers): matrices whose rows are 1 ... 9000 entries long in random
    mixtures -- runs of long rows of equal and of different lengths (several per wave, in registers), single long rows among
    short ones (a wave each, chunks meeting in atomics in these small matrices), columns within 65536 of each other (16-bit
    columns) and scattered wider (32-bit) -- default plan, SPMV_HIP_FLAG_NO_MULTI_WINDOW, exact order, two runs, and the same rows as
    an ELLPACK upload where the padded size allows.

    python3 tests/soak_round5.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def long_row_matrix(seed):
    rng = np.random.default_rng(seed)
    rows = int(rng.integers(300, 1500))
    wide = bool(rng.integers(0, 2))
    cols = int(rng.integers(120000, 400000)) if wide else int(rng.integers(9500, 40000))
    lens = np.zeros(rows, dtype=np.int64)
    r = 0
    while r < rows:
        kind = rng.integers(0, 5)
        run = int(rng.integers(1, 12))
        if kind == 0:    # short rows
            lens[r:r + run] = rng.integers(0, 40, size=min(run, rows - r))
        elif kind == 1:  # a run of equally long rows
            lens[r:r + run] = int(rng.choice([513, 600, 1001, 1024, 1025, 1536, 1537, 2049, 3001, 4099]))
        elif kind == 2:  # a run of long rows of different lengths
            lens[r:r + run] = rng.integers(513, 5000, size=min(run, rows - r))
        elif kind == 3:  # medium rows (multi-window tiles through LDS) next to long ones
            lens[r:r + run] = rng.integers(161, 1100, size=min(run, rows - r))
        else:            # one very long row
            lens[r] = int(rng.integers(5000, 9000))
            run = 1
        r += run
    lens = np.minimum(lens, cols // 2)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    parts = []
    for i, n in enumerate(lens):
        if n == 0:
            continue
        if wide and rng.random() < 0.5:
            parts.append(np.sort(rng.choice(cols, size=int(n), replace=False)))
        else:  # a row whose columns stay within 65536 of each other
            span = int(min(cols, max(n + 10, rng.integers(n + 10, 60000))))
            lo = int(rng.integers(0, cols - span + 1))
            parts.append(lo + np.sort(rng.choice(span, size=int(n), replace=False)))
    c = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, dtype=np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    return rows, cols, p.astype(np.int32), c, v


def main():
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_blocktiles import fem_ragged, run_plan
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    masked = dense = tiles = multi = long_rows = multiplies = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        # ---- masked block tiles
        lo = int(rng.integers(6, 40))
        hi = lo + int(rng.integers(0, 16))
        drop = float(rng.choice([0.0, 0.01, 0.03, 0.1, 0.2, 0.3]))
        odd = int(rng.choice([0, 0, 7, 40, 333]))
        rows, cols, p, c, v = fem_ragged(int(rng.integers(1500, 4000)), lo, hi, seed=seed, drop=drop, odd_every=odd)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        what = "seed %d blocks %d-%d drop %.2f odd %d" % (seed, lo, hi, drop, odd)
        got, info = run_plan(rows, cols, p, c, v, x, y0)
        assert_close(got, want, scale, what=what, nterms=3 * hi + 3)
        got_n, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_MASKED_BLOCKS)
        assert_close(got_n, want, scale, what=what + ", no masked blocks", nterms=3 * hi + 3)
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
        assert_bitexact(got_e, want, what + ", exact order")
        got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=6 * hi + 6)
        masked += info["masked_block_tiles"]
        dense += info["block_tiles"] - info["masked_block_tiles"]
        tiles += info["row_blocks"]
        multiplies += 5
        # ---- long rows
        rows, cols, p, c, v = long_row_matrix(seed)
        x = synth.x_vector(cols, seed=seed + 3)
        y0 = synth.x_vector(rows, seed=seed + 4)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        what = "seed %d long rows (%d rows, %d entries)" % (seed, rows, len(c))
        got, info = run_plan(rows, cols, p, c, v, x, y0, index_values=False)
        assert_close(got, want, scale, what=what, nterms=9000)
        got_n, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_MULTI_WINDOW, index_values=False)
        assert_close(got_n, want, scale, what=what + ", no shared tiles", nterms=9000)
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER, index_values=False)
        assert_bitexact(got_e, want, what + ", exact order")
        got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True, index_values=False)
        assert_close(got_c, want, scale, what=what + ", other column array", nterms=9000)
        got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2, index_values=False)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=18000)
        multi += info["multi_window_tiles"]
        long_rows += info["long_blocks"]
        multiplies += 6
        # the same rows as an ELLPACK upload (padded to the longest row) where that stays small
        L = int(np.diff(p).max())
        if rows * L <= 6_000_000 and np.diff(p)[0] > 0:
            i, j, a = synth.csr_to_coordinate(rows, p, c, v)
            rc, Lr, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
            assert rc == 0 and Lr == L
            wante = oracle.ell_spmv(rows, L, ec, ev, x, y=y0)
            with capi.Context(0) as ctx:
                ctx.upload_ell(rows, cols, L, ec, ev)
                ctx.set_x(x)
                ctx.set_y(y0)
                ctx.run()
                gote = ctx.get_y()
            escale = (np.abs(ev.reshape(rows, L)) * np.abs(x[ec.reshape(rows, L)])).sum(axis=1) + np.abs(y0)
            assert_close(gote, wante, escale, what=what + ", as ELLPACK L=%d" % L, nterms=L)
            multiplies += 1
        if (seed - first) % 10 == 9:
            print("seed %d: %d multiplies; block tiles %d dense + %d masked of %d; %d shared long-row / multi-window tiles, %d long-row tiles"
                  % (seed, multiplies, dense, masked, tiles, multi, long_rows), flush=True)
    assert masked > 0 and multi > 0 and long_rows > 0
    print("soak ok: %d seeds, %d multiplies, none off; block tiles %d dense + %d masked of %d; %d shared long-row / multi-window tiles, %d long-row tiles"
          % (count, multiplies, dense, masked, tiles, multi, long_rows))


if __name__ == "__main__":
    main()
