#!/usr/bin/env python3
"""Soak of the block-tile classifier and kernel path (csr_blocktile.hpp) on tests/test_gpu_blocktiles.py's block_patchwork:
many seeds, each multiplied with block tiles, without them and in exact order, every result against the oracle.  Prints how
many tiles were block tiles over the whole soak (a soak that never marks a tile tests nothing).  Kept under tests/ because it
uses the checker library (oracle/); not collected by pytest.
    python3 tests/soak_blocktiles.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_blocktiles import block_patchwork, run_plan
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    marked = tiles = hinted = 0
    for seed in range(first, first + count):
        rows, cols, p, c, v = block_patchwork(seed, nodes=3000)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        got, info = run_plan(rows, cols, p, c, v, x, y0)
        assert_close(got, want, scale, what="seed %d" % seed, nterms=2100)
        got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what="seed %d, two runs" % seed, nterms=4200)
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
        assert_bitexact(got_e, want, "seed %d exact" % seed)
        marked += info["block_tiles"]
        tiles += info["row_blocks"]
        hinted += info["block_tiles"] > 0
        if (seed - first) % 10 == 9:
            print("seed %d: %d of %d tiles were block tiles so far, %d of %d matrices had some" % (seed, marked, tiles, hinted, seed - first + 1), flush=True)
    assert marked > 0
    print("soak ok: %d matrices, %d block tiles of %d" % (count, marked, tiles))


if __name__ == "__main__":
    main()
