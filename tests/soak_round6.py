#!/usr/bin/env python3
"""Soak of round 6's tile classes on the GPU, every result against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in
oracle/spmv_oracle.c); kept under tests/ because it uses the checker library, not collected by pytest.

  * WIDE block tiles and STORED TRIANGLES (csr_blocktile.hpp, plan_csr.hip): meshes of 3 x 3 blocks (random neighbour counts, a
    random share of entries dropped, nodes with one or two unknowns) whose column nodes are scattered over 1 ... 3000 times as many
    nodes (tiles narrow, wide within 22 bits, wide beyond), as they are / as their lower / as their upper triangle; default plan,
    SPMV_HIP_FLAG_NO_BLOCK_TILES, exact order (bit-exact), two accumulating runs, another column array;
  * WIDE group tiles: meshes with 2 / 4 unknowns per node, column nodes scattered likewise;
  * the PIPELINED gather of the single-process multi-GPU path (SPMV_HIP_FLAG_PIPELINE_GATHER): 2 ... 8 parts on this device, 1 ... 7
    back-to-back runs, every part's copy compared (VERIFY_PLAN), the bits of the serial order.

    python3 tests/soak_round6.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def scatter(rows, cols, p, c, v, d, factor, seed):
    """column node m -> the m-th of a sorted random subset of factor * nodes nodes (rows stay ascending, blocks aligned)"""
    if factor == 1:
        return rows, cols, p, c, v
    rng = np.random.default_rng(seed)
    nodes = cols // d
    pick = np.sort(rng.choice(nodes * factor, size=nodes, replace=False)).astype(np.int64)
    c2 = d * pick[c // d] + c % d
    return rows, cols * factor, p, c2.astype(np.int32), v


def triangle(rows, cols, p, c, v, which):
    if which == "full" or rows != cols:
        return rows, cols, p, c, v
    import scipy.sparse as sp
    A = sp.csr_matrix((v, c, p), shape=(rows, cols))
    A = sp.tril(A, format="csr") if which == "lower" else sp.triu(A, format="csr")
    A.sort_indices()
    return rows, cols, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data


def main():
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_blocktiles import fem_ragged, run_plan
    from test_gpu_grouptiles import run_plan as run_group_plan
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    block = masked = wide_plans = tiles = group = gtiles = multiplies = pipelined = 0
    os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        # ---- block matrices: as they are / stored triangles, narrow / wide / beyond 22 bits
        lo = int(rng.integers(4, 36))
        hi = lo + int(rng.integers(0, 20))
        drop = float(rng.choice([0.0, 0.0, 0.02, 0.1]))
        odd = int(rng.choice([0, 0, 0, 0, 50, 400]))
        which = str(rng.choice(["full", "full", "lower", "upper"]))
        factor = int(rng.choice([1, 1, 30, 200, 3000]))
        rows, cols, p, c, v = fem_ragged(int(rng.integers(1500, 4000)), lo, hi, seed=seed, drop=drop, odd_every=odd)
        if which != "full":
            rows, cols, p, c, v = triangle(rows, cols, p, c, v, which)
        elif odd == 0:
            rows, cols, p, c, v = scatter(rows, cols, p, c, v, 3, factor, seed)
        base = capi.FLAG_NO_COLUMN_PANELS  # (scattered twins have a large x: the panel copy would hide the tiles under test)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        what = "seed %d blocks %d-%d drop %.2f odd %d %s x%d" % (seed, lo, hi, drop, odd, which, factor)
        got, info = run_plan(rows, cols, p, c, v, x, y0, flags=base, index_values=False)
        assert_close(got, want, scale, what=what, nterms=3 * hi + 6)
        got_n, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base | capi.FLAG_NO_BLOCK_TILES, index_values=False)
        assert_close(got_n, want, scale, what=what + ", no block tiles", nterms=3 * hi + 6)
        got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base | capi.FLAG_EXACT_ORDER, index_values=False)
        assert_bitexact(got_e, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=1), what + ", exact order")
        got2, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base, runs=2, index_values=False)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=6 * hi + 12)
        got_c, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base, other_columns=True, index_values=False)
        assert_close(got_c, want, scale, what=what + ", other column array", nterms=3 * hi + 6)
        block += info["block_tiles"]
        masked += info["masked_block_tiles"]
        tiles += info["row_blocks"]
        wide_plans += int(info["block_tiles"] > 0 and info["narrow_tiles"] < info["row_blocks"] // 2)
        multiplies += 6
        # ---- group tiles, narrow and wide
        d = int(rng.choice([2, 4]))
        grid = (int(rng.integers(8, 20)), int(rng.integers(8, 16)), int(rng.integers(6, 14)))
        gfac = int(rng.choice([1, 60, 500]))
        rows, cols, p, c, v = scatter(*synth.mesh_dofs(grid, d, seed=seed), d, gfac, seed)
        x = synth.x_vector(cols, seed=seed + 3)
        y0 = synth.x_vector(rows, seed=seed + 4)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        what = "seed %d mesh %s x %d dof, columns x%d" % (seed, grid, d, gfac)
        got, ginfo = run_group_plan(rows, cols, p, c, v, x, y0, flags=base)
        assert_close(got, want, scale, what=what)
        got2, _ = run_group_plan(rows, cols, p, c, v, x, y0, flags=base, runs=2)
        assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs")
        got_o, _ = run_group_plan(rows, cols, p, c, v, x, y0, flags=base, out_of_place=True)
        assert_bitexact(got_o, got, what + ", y_out")
        group += ginfo["group_tiles"]
        gtiles += ginfo["row_blocks"]
        multiplies += 4
        # ---- the pipelined gather of the single-process multi-GPU path, on the same matrix
        parts = int(rng.integers(2, 9))
        runs = int(rng.integers(1, 8))
        ys = {}
        for name, extra in (("serial", 0), ("pipelined", capi.FLAG_PIPELINE_GATHER)):
            with capi.Context(num_gpus=parts, flags=capi.FLAG_PEER_GATHER | capi.FLAG_VERIFY_PLAN | capi.FLAG_NO_COLUMN_PANELS | extra) as ctx:
                ctx.upload_csr(rows, cols, p, c, v)
                ctx.set_x(x)
                ctx.set_y(y0)
                ctx.run(runs)
                ys[name] = ctx.get_y()
                if extra:
                    pipelined += int(ctx.info()["pipelined"])
        assert_bitexact(ys["pipelined"], ys["serial"], what + ", %d parts, %d runs: pipelined against serial" % (parts, runs))
        assert_close(ys["pipelined"], oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=runs), runs * scale, what=what + ", %d parts" % parts)
        multiplies += 2 * runs
        if (seed - first) % 10 == 9:
            print("seed %d: %d multiplies; block tiles %d (%d masked) of %d, %d plans mostly wide; group tiles %d of %d; %d pipelined contexts"
                  % (seed, multiplies, block, masked, tiles, wide_plans, group, gtiles, pipelined), flush=True)
    assert block > 0 and masked > 0 and wide_plans > 0 and group > 0 and pipelined > 0
    print("soak ok: %d seeds, %d multiplies, none off; block tiles %d (%d masked) of %d, %d plans mostly wide; group tiles %d of %d; %d pipelined contexts"
          % (count, multiplies, block, masked, tiles, wide_plans, group, gtiles, pipelined))


if __name__ == "__main__":
    main()
