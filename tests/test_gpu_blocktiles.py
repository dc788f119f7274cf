"""Block tiles (csr_blocktile.hpp): matrices of dense 3 x 3 blocks -- three unknowns per mesh node, Queen_4147's kind -- read
one 16-bit number per BLOCK instead of a column index per entry, and no row_ptr.  The plan only takes a hint from row_ptr
(rows in triples of equal length); spmv_hip_plan_csr_repack then checks every tile entry by entry before it marks it.

Checked here against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c): meshes with 6 ... 56
blocks per row (1 ... 9 block rows per tile), tiles whose structure is broken in one place (one column moved, one row of a
triple longer, a block off the 3-grid) and must fall back alone, rows of <= 16 entries (never block tiles: bit-exact), the
exact-order flag, accumulation, y_out != y_in, another column array at spmv time (nothing derived may be used), the
context uploads (CSR, COO), and a matrix with a value dictionary (its launch ignores the marks)."""
import numpy as np
import pytest

from helpers import assert_bitexact, assert_close, abs_products
from spmv_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu


def fem3(nodes, nbr_lo, nbr_hi, seed, reach=1500):
    """3 unknowns per node; node n is linked to a random sorted set of nbr_lo..nbr_hi nodes within `reach` (itself included):
    rows 3n .. 3n+2 carry the columns 3m .. 3m+2 of every linked node m."""
    rng = np.random.default_rng(seed)
    cnt = rng.integers(nbr_lo, nbr_hi + 1, size=nodes)
    ptr = np.zeros(nodes + 1, dtype=np.int64)
    np.cumsum(cnt, out=ptr[1:])
    nbr = np.empty(int(ptr[-1]), dtype=np.int64)
    for n in range(nodes):
        lo, hi = max(0, n - reach), min(nodes, n + reach + 1)
        pick = rng.choice(hi - lo, size=min(cnt[n], hi - lo), replace=False) + lo
        pick[0] = n
        pick = np.unique(pick)
        while len(pick) < cnt[n]:  # duplicates of n removed: top up
            extra = rng.integers(lo, hi)
            pick = np.unique(np.append(pick, extra))
        nbr[ptr[n]:ptr[n + 1]] = pick[:cnt[n]]
    lens = np.repeat(3 * cnt, 3)
    p = np.zeros(3 * nodes + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = np.empty(int(p[-1]), dtype=np.int32)
    for n in range(nodes):
        cols = (3 * nbr[ptr[n]:ptr[n + 1]][:, None] + np.arange(3)[None, :]).ravel()
        for a in range(3):
            c[p[3 * n + a]:p[3 * n + a + 1]] = cols
    v = rng.uniform(-1.0, 1.0, size=len(c))
    return 3 * nodes, 3 * nodes, p.astype(np.int32), c, v


def run_plan(rows, cols, p, c, v, x, y0, flags=0, runs=1, other_columns=False, out_of_place=False, index_values=True):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
    plan.compress(tc.data_ptr(), stream)
    plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    if index_values:
        plan.index_values(tv.data_ptr(), stream)
    info = plan.info()
    cols_now = tc.clone() if other_columns else tc
    ty = torch.from_numpy(y0.copy()).to(dev)
    if out_of_place:
        tout = torch.full((rows,), np.nan, dtype=torch.float64, device=dev)
        plan.spmv_out(tp.data_ptr(), cols_now.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), tout.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_bitexact(ty.cpu().numpy(), y0, "y_in untouched")
        ty = tout
    else:
        for _ in range(runs):
            plan.spmv(tp.data_ptr(), cols_now.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    got = ty.cpu().numpy()
    plan.close()
    return got, info


@pytest.mark.parametrize("name,lo,hi,nodes", [("27 per node", 27, 27, 6000), ("20-34 per node", 20, 34, 6000), ("6-12 per node", 6, 12, 9000),
                                              ("6 per node", 6, 6, 9000), ("40-56 per node", 40, 56, 3000), ("mixed 6-56", 6, 56, 5000)])
def test_block_tiles_against_oracle(oracle, name, lo, hi, nodes):
    rows, cols, p, c, v = fem3(nodes, lo, hi, seed=len(name))
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["block_tiles"] >= 0.95 * info["row_blocks"], (name, info)
    assert info["block_entries"] >= 0.95 * info["nnz"], (name, info)
    assert_close(got, want, scale, what=name, nterms=3 * hi)
    # fewer bytes than the same plan without them: 8 + 2/9 instead of 8 + 2 (+ row_ptr) per entry
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info_n["block_tiles"] == 0
    assert info["streamed_bytes"] < info_n["streamed_bytes"] - 1.5 * info["nnz"], (info["streamed_bytes"], info_n["streamed_bytes"])
    assert_close(got_n, want, scale, what=name + ", no block tiles", nterms=3 * hi)
    # the exact-order flag: no block tiles, one lane per row, the reference's bits
    got_e, info_e = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    assert info_e["block_tiles"] == 0
    assert_bitexact(got_e, want, name + ", exact order")
    # accumulating twice, and y_out = y_in + A x
    got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
    assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=name + ", two runs", nterms=6 * hi)
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    assert np.array_equal(got_o.view(np.uint64), got.view(np.uint64)), name + ": y_out differs from the in-place result"
    # another column array at the same multiply: nothing derived from the plan's columns may be used
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    assert_close(got_c, want, scale, what=name + ", other column array", nterms=3 * hi)


def test_broken_tiles_fall_back_alone(oracle):
    """One column moved inside a block, one block off the 3-grid, one row of a triple a copy of ANOTHER row pattern: the tile that
    holds the damage loses its mark (the check is entry by entry), its neighbours keep theirs, y matches the oracle."""
    rows, cols, p, c, v = fem3(6000, 24, 30, seed=11)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    _, clean = run_plan(rows, cols, p, c, v, x, y0)
    assert clean["block_tiles"] >= 0.95 * clean["row_blocks"]
    damaged = 0
    for what, node in (("column moved", 1000), ("block off the grid", 2500), ("row differs from its triple", 4000)):
        c2 = c.copy()
        r = 3 * node + 1
        k = int(p[r]) + 4  # second entry of the row's second block
        if what == "column moved":
            c2[k] = c2[k - 1]  # a duplicate column: still a valid CSR matrix, no longer a run c, c+1, c+2
        elif what == "block off the grid":
            # move a whole block (in all three rows) one column to the right where there is room
            for a in range(3):
                q = int(p[3 * node + a]) + 3 * (int(p[3 * node + 1] - p[3 * node]) // 3 - 1)  # the row's last block
                c2[q:q + 3] = np.minimum(c2[q:q + 3] + 1, cols - 1)
        else:
            q = int(p[r])
            c2[q:q + 3] = c2[q + 3:q + 6]  # first block repeats the second one's columns in this row only
        # (round 4's rule, SPMV_HIP_FLAG_NO_MASKED_BLOCKS: any damage demotes the tile)
        got, info = run_plan(rows, cols, p, c2, v, x, y0, flags=capi.FLAG_NO_MASKED_BLOCKS)
        assert clean["block_tiles"] - 2 <= info["block_tiles"] < clean["block_tiles"], (what, clean["block_tiles"], info["block_tiles"])
        assert info["masked_block_tiles"] == 0
        damaged += 1
        want2 = oracle.csr_spmv(rows, p, c2, v, x, y=y0, num_threads=4)
        assert_close(got, want2, abs_products(rows, p, c2, v, x) + np.abs(y0), what=what)
        # default (round 5): a block off the grid is still a block -- the tile becomes a MASKED block tile; so does the row that
        # repeats a block's columns (the greedy cover simply opens the same block twice: every entry still meets its own x);
        # a column twice INSIDE one block cannot be expressed by a mask: that tile falls back as before
        got, info = run_plan(rows, cols, p, c2, v, x, y0)
        if what == "column moved":
            assert clean["block_tiles"] - 2 <= info["block_tiles"] < clean["block_tiles"] and info["masked_block_tiles"] == 0, (what, clean, info)
        else:
            assert info["block_tiles"] == clean["block_tiles"] and 1 <= info["masked_block_tiles"] <= 2, (what, clean, info)
        assert_close(got, want2, abs_products(rows, p, c2, v, x) + np.abs(y0), what=what + " (masked allowed)")
    assert damaged == 3
    # a triple whose rows differ in LENGTH: the hint survives (one triple in 6000), the tile does not qualify
    lens = np.diff(p).astype(np.int64)
    r = 3 * 3000
    keep = np.ones(len(c), dtype=bool)
    keep[int(p[r + 2]) + 3:int(p[r + 2]) + 6] = False  # third row of the triple loses a block
    lens[r + 2] -= 3
    p3 = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p3[1:])
    p3 = p3.astype(np.int32)
    c3, v3 = c[keep], v[keep]
    got, info = run_plan(rows, cols, p3, c3, v3, x, y0, flags=capi.FLAG_NO_MASKED_BLOCKS)
    assert info["block_tiles"] > 0.9 * clean["block_tiles"], (clean, info)
    assert info["row_blocks"] - info["block_tiles"] > clean["row_blocks"] - clean["block_tiles"], (clean, info)
    want3 = oracle.csr_spmv(rows, p3, c3, v3, x, y=y0, num_threads=4)
    assert_close(got, want3, abs_products(rows, p3, c3, v3, x) + np.abs(y0), what="ragged triple")
    # ... and by default that tile is a masked block tile (the third row's mask lacks one block)
    got, info = run_plan(rows, cols, p3, c3, v3, x, y0)
    assert info["masked_block_tiles"] >= 1 and info["row_blocks"] - info["block_tiles"] <= clean["row_blocks"] - clean["block_tiles"] + 1, (clean, info)
    assert_close(got, want3, abs_products(rows, p3, c3, v3, x) + np.abs(y0), what="ragged triple (masked)")


@pytest.mark.parametrize("first", [1, 2, 3001, 4499])
def test_row_block_that_starts_inside_the_grid_of_triples(oracle, first):
    """One rank's rows of a partitioned matrix start wherever ceil(rows / G) puts them (queen-like at G = 8: row 518 389, not a
    multiple of 3): the plan finds the grid of triples at any of the three offsets and cuts its tiles on it."""
    from spmv_amd import partition
    rows, cols, p, c, v = fem3(3000, 20, 30, seed=3)
    lp, lc, lv = partition.csr_slice(p, c, v, first, rows)
    lrows = rows - first
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(lrows, seed=4)
    got, info = run_plan(lrows, cols, lp, lc, lv, x, y0)
    assert info["block_tiles"] >= 0.9 * info["row_blocks"], (first, info)
    assert_close(got, oracle.csr_spmv(lrows, lp, lc, lv, x, y=y0, num_threads=4), abs_products(lrows, lp, lc, lv, x) + np.abs(y0), what="rows %d.." % first, nterms=90)


def test_short_block_rows_stay_bit_exact(oracle):
    """Rows of at most 16 entries (5 blocks) keep their one-lane-per-row tiles: no block tiles, the reference's bits."""
    rows, cols, p, c, v = fem3(9000, 3, 5, seed=5)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["block_tiles"] == 0, info
    assert_bitexact(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), "rows of <= 15 entries")


def test_no_hint_no_block_tiles_and_scalar_meshes(oracle):
    """A matrix whose rows do not come in equal triples gets no hint (the tiles are cut as ever), and one that does but has no
    3 x 3 blocks (a scalar 27-point mesh with 3 N rows) gets the hint, no marks, and the same y."""
    rows, cols, p, c, v = synth.stencil27_like(30, 30, 30)
    x = synth.x_vector(cols, seed=3)
    y0 = np.zeros(rows)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["block_tiles"] == 0
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, num_threads=4), abs_products(rows, p, c, v, x), what="27-point")
    rows, cols, p, c, v = synth.banded(30000, list(range(-13, 14)), seed=2)  # 27 per row: triples of equal length, divisible by 3
    x = synth.x_vector(cols, seed=3)
    got, info = run_plan(rows, cols, p, c, v, x, np.zeros(rows), flags=capi.FLAG_NO_SHIFTED_TILES)
    assert info["block_tiles"] == 0, info
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, num_threads=4), abs_products(rows, p, c, v, x), what="band of 27")


def test_wrong_hint_rebuilds_the_tiles(oracle):
    """Rows of 120 entries in equal triples but NO blocks (a scalar matrix): the hint cuts 4-row tiles down to 3 rows, the check
    finds no block anywhere, and repack builds the tiles once more without the hint -- the plan ends up exactly as if the
    hint had never been taken: same tiles, same bits."""
    rng = np.random.default_rng(8)
    rows = cols = 30000
    L = 120
    c = np.sort(np.clip(np.arange(rows)[:, None] + rng.choice(np.arange(-4000, 4000), size=(rows, L)), 0, cols - 1), axis=1)
    c = c.astype(np.int32).ravel()
    p = (np.arange(rows + 1, dtype=np.int64) * L).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info["block_tiles"] == 0 and info["row_blocks"] == info_n["row_blocks"] == rows // 4, (info, info_n)
    assert_bitexact(got, got_n, "rebuilt plan against the plan without the hint")
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="scalar rows of 120", nterms=L)


def test_dictionary_built_before_a_repack_that_cuts_the_tiles_anew(oracle):
    """compress, index_values, THEN repack -- an order the header allows.  A constant-coefficient
    30-point stencil has rows of 30 entries in equal triples (the block hint) that are no 3 x 3 blocks (the hint is wrong and
    cost tile fill -- 15 rows per tile instead of 17: repack cuts the tiles anew) and few distinct values (a dictionary with constant-row tiles).  The dictionary
    built on the old tiling must not survive the re-cut: repack drops it and builds it again on the new tiles."""
    import torch
    n = 40
    idx = np.arange(n ** 3).reshape(n, n, n)
    inner = idx[1:-1, 1:-1, 1:-1].ravel()
    # 27 points + 3 more: rows of 30 entries, of which 17 fit a 512-entry tile -- the hint cuts that down to 15 (block_cuts > 0)
    offs = np.array([dz * n * n + dy * n + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)] + [-2, 2, 2 * n])
    L = len(offs)
    coef = np.where(offs == 0, 26.0, -1.0) * np.array([1.0 + 0.25 * (k % 3) for k in range(L)])  # 5 distinct values
    rows = cols = len(inner)
    remap = -np.ones(n ** 3, dtype=np.int64)
    remap[inner] = np.arange(rows)
    cc = remap[inner[:, None] + offs[None, :]]
    keep = cc >= 0
    # boundary rows of the interior block lose entries: pad them to the full length with zeros on their own diagonal
    cc = np.where(keep, cc, remap[inner][:, None])
    vv = np.where(keep, coef[None, :], 0.0)
    order = np.argsort(cc, axis=1, kind="stable")
    c = np.take_along_axis(cc, order, axis=1).astype(np.int32).ravel()
    v = np.take_along_axis(vv, order, axis=1).ravel()
    p = (np.arange(rows + 1, dtype=np.int64) * L).astype(np.int32)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got = {}
    for name, order_of_calls in (("index_values before repack", ("compress", "index_values", "repack")),
                                 ("the default order", ("compress", "repack", "index_values"))):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
        before = None
        for call in order_of_calls:
            if call == "compress":
                plan.compress(tc.data_ptr(), stream)
            elif call == "index_values":
                plan.index_values(tv.data_ptr(), stream)
            else:
                before = plan.info()
                plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        info = plan.info()
        assert info["block_tiles"] == 0 and info["row_blocks"] < before["row_blocks"], (name, before["row_blocks"], info["row_blocks"])
        assert info["indexed_values"] > 0, (name, info)  # the dictionary is there AFTER the re-cut, whatever the order
        ty = torch.from_numpy(y0.copy()).to(dev)
        for _ in range(2):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        got[name] = ty.cpu().numpy()
        plan.close()
        want2 = oracle.csr_spmv(rows, p, c, v, x, y=want.copy(), num_threads=4)
        assert_close(got[name], want2, 2 * scale, what=name, nterms=L)
    assert_bitexact(got["index_values before repack"], got["the default order"], "both orders end in the same plan")


def test_queen_like_generator_and_context_uploads(oracle):
    """The Queen_4147 stand-in at a small size through the Level-2 plan and through the context API (CSR and COO uploads: the
    sorted triplets run as the same row-major tiles)."""
    M = hostapi.load("synthetic:queen:30,24,20", "csr")
    rows, cols, p, c, v = M.rows, M.cols, np.array(M.row_ptr), np.array(M.column_index), np.array(M.value)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["block_tiles"] >= 0.9 * info["row_blocks"], info
    assert_close(got, want, scale, what="queen-like, plan")
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what="queen-like, context CSR")
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        rng = np.random.default_rng(1)
        perm = rng.permutation(len(a))
        ctx.upload_coo(rows, cols, np.ascontiguousarray((i - 1)[perm].astype(np.int32)), np.ascontiguousarray((j - 1)[perm].astype(np.int32)),
                       np.ascontiguousarray(a[perm]))
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what="queen-like, context COO (shuffled)")
    M.close()


def test_block_structure_with_a_value_dictionary(oracle):
    """Few distinct values: the dictionary launch has no block path and must ignore the marks (16-bit columns intact)."""
    rows, cols, p, c, v = fem3(5000, 20, 30, seed=21)
    rng = np.random.default_rng(2)
    v = np.array([-1.0, 0.5, 2.0, 0.125])[rng.integers(0, 4, size=len(v))]
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["indexed_values"] == 4 and info["block_tiles"] == 0, info  # (reported as 0: the launch does not use them)
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="dictionary")
    got_g, info_g = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_VALUE_INDEX)
    assert info_g["block_tiles"] > 0
    assert_close(got_g, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="values read")


def block_patchwork(seed, nodes=4000):
    """Sections of 3 x 3 block rows (2 ... 60 blocks per row, any mix: 1 ... 9 block rows per tile) interleaved with sections
    that are NOT blocks (rows of any length, sometimes a number of rows that is no multiple of 3, so that the triples behind
    them sit off the tiler's 3-row grid), with damage sprinkled over the block sections: a column moved, a row of a triple
    made longer, a block off the 3-grid.  Columns stay within 60 000 of the row (16-bit tiles) or anywhere (32-bit)."""
    rng = np.random.default_rng(seed)
    cols = 3 * nodes + 300
    rows_c, rows_v = [], []
    node = 0
    while node < nodes:
        kind = rng.integers(0, 10)
        if kind <= 6:  # a run of block rows
            n = int(rng.integers(5, 400))
            lo_b, hi_b = sorted(rng.integers(2, 61, size=2).tolist())
            wide = rng.integers(0, 8) == 0
            for _ in range(min(n, nodes - node)):
                nb = int(rng.integers(lo_b, hi_b + 1))
                reach = nodes if wide else 6000
                lo, hi = max(0, node - reach), min(nodes, node + reach + 1)
                nbrs = np.unique(rng.integers(lo, hi, size=nb))
                cc = (3 * nbrs[:, None] + np.arange(3)[None, :]).ravel()
                trio = [cc.copy(), cc.copy(), cc.copy()]
                dmg = rng.integers(0, 60)
                if dmg == 0 and len(cc) > 3:
                    trio[int(rng.integers(0, 3))][int(rng.integers(1, len(cc)))] += 1  # may collide with its neighbour: a duplicate
                elif dmg == 1:
                    trio[int(rng.integers(0, 3))] = np.append(cc, [cols - 3, cols - 2, cols - 1])  # one row a block longer
                elif dmg == 2 and len(cc) > 3:
                    for t in trio:
                        t[-3:] = np.minimum(t[-3:] + 1, cols - 1)  # last block off the grid
                for t in trio:
                    rows_c.append(np.sort(t))
                node += 1
        elif kind <= 8:  # rows that are no blocks; one time in four knocking the triples behind them off the 3-row grid
            nr = int(rng.integers(1, 13))
            if rng.integers(0, 4):
                nr = 3 * ((nr + 2) // 3)
            for _ in range(nr):
                rows_c.append(np.sort(rng.integers(0, cols, size=int(rng.integers(0, 120)))))
        else:
            for _ in range(3 if rng.integers(0, 4) else 1):
                rows_c.append(np.sort(rng.integers(0, cols, size=int(rng.choice([513, 700, 2049])))))
    lens = np.array([len(r) for r in rows_c], dtype=np.int64)
    p = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = np.concatenate(rows_c).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    return len(lens), cols, p.astype(np.int32), c, v


@pytest.mark.parametrize("seed", range(10))
def test_block_patchwork(oracle, seed):
    rows, cols, p, c, v = block_patchwork(seed)
    x = synth.x_vector(cols, seed=seed + 50)
    y0 = synth.x_vector(rows, seed=seed + 60)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert_close(got, want, scale, what="block patchwork %d" % seed, nterms=2100)
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info_n["block_tiles"] == 0
    assert_close(got_n, want, scale, what="block patchwork %d, no block tiles" % seed, nterms=2100)
    got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    assert_bitexact(got_e, want, "block patchwork %d, exact order" % seed)


def fem_ragged(nodes, nbr_lo, nbr_hi, seed, drop=0.0, odd_every=0, reach=1500):
    """fem3 with what real files do: every stored entry of an off-diagonal block is dropped with probability `drop` (explicit
    zeros the assembly left out), and with odd_every = K every K-th node has one or two unknowns instead of three (the grid of
    row and column triples moves behind it)."""
    rng = np.random.default_rng(seed)
    dofs = np.full(nodes, 3, dtype=np.int64)
    if odd_every:
        odd = np.arange(odd_every // 2, nodes, odd_every)
        dofs[odd] = rng.integers(1, 3, size=len(odd))
    first = np.zeros(nodes + 1, dtype=np.int64)
    np.cumsum(dofs, out=first[1:])
    rows = int(first[-1])
    row_cols, row_vals = [], []
    for n in range(nodes):
        lo, hi = max(0, n - reach), min(nodes, n + reach + 1)
        k = int(rng.integers(nbr_lo, nbr_hi + 1))
        pick = np.unique(np.append(rng.choice(hi - lo, size=min(k, hi - lo), replace=False) + lo, n))
        cols_n = np.concatenate([first[m] + np.arange(dofs[m]) for m in pick])
        diag = np.concatenate([np.full(dofs[m], m == n) for m in pick])
        for a in range(dofs[n]):
            keep = diag | (rng.random(len(cols_n)) >= drop)
            row_cols.append(cols_n[keep])
            row_vals.append(rng.uniform(-1.0, 1.0, size=int(keep.sum())))
    lens = np.array([len(r) for r in row_cols], dtype=np.int64)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    return rows, rows, p.astype(np.int32), np.concatenate(row_cols).astype(np.int32), np.concatenate(row_vals)


@pytest.mark.parametrize("name,lo,hi,nodes,drop,odd", [
    ("2 % of the entries dropped", 24, 30, 6000, 0.02, 0),
    ("10 % dropped", 20, 34, 6000, 0.10, 0),
    ("25 % dropped", 27, 27, 5000, 0.25, 0),
    ("an odd node every 40", 24, 30, 6000, 0.0, 40),
    ("odd nodes and 5 % dropped", 20, 30, 6000, 0.05, 97),
    ("few blocks per row, 10 % dropped", 8, 12, 9000, 0.10, 0),
    ("many blocks per row, 3 % dropped", 40, 52, 3000, 0.03, 0)])
def test_masked_block_tiles_against_oracle(oracle, name, lo, hi, nodes, drop, odd):
    """Blocks with entries missing, rows of a triple that differ in length, nodes with one or two unknowns: the tiles are covered
    with blocks of three consecutive columns and a 9-bit mask each (csr_blocktile.hpp) and multiplied with one lane per block,
    the values read in place.  Against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33), against the same plan without masked
    blocks, accumulating, y_out != y_in, with another column array, and never under EXACT_ORDER."""
    rows, cols, p, c, v = fem_ragged(nodes, lo, hi, seed=len(name), drop=drop, odd_every=odd)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["masked_block_tiles"] > 0 and info["block_tiles"] >= 0.8 * info["row_blocks"], (name, info)
    assert_close(got, want, scale, what=name, nterms=3 * hi + 3)
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_MASKED_BLOCKS)
    assert info_n["masked_block_tiles"] == 0
    assert info["streamed_bytes"] < info_n["streamed_bytes"] - 0.8 * info["masked_block_entries"], (info, info_n)
    assert_close(got_n, want, scale, what=name + ", no masked blocks", nterms=3 * hi + 3)
    got_e, info_e = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    assert info_e["block_tiles"] == 0
    assert_bitexact(got_e, want, name + ", exact order")
    got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
    assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=name + ", two runs", nterms=6 * hi + 6)
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    assert np.array_equal(got_o.view(np.uint64), got.view(np.uint64)), name + ": y_out differs from the in-place result"
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    assert_close(got_c, want, scale, what=name + ", other column array", nterms=3 * hi + 3)


def test_candidate_without_blocks_pays_for_a_sample_only(oracle):
    """Rows in triples of SIMILAR length whose columns have nothing to do with each other (a scalar mesh: 20 ... 26 random columns
    per row): the plan notes a candidate, repack's sample of triples finds three entries per block, and nothing else happens --
    the tiles are the ones the plan without block tiles has, the result its bits."""
    rng = np.random.default_rng(12)
    rows = cols = 30000
    lens = np.repeat(rng.integers(20, 27, size=rows // 3), 3) + rng.integers(0, 2, size=rows)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = np.concatenate([np.sort(rng.choice(np.arange(max(0, r - 3000), min(cols, r + 3000)), size=n, replace=False)) for r, n in enumerate(lens)]).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    p = p.astype(np.int32)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info["block_tiles"] == 0 and info["row_blocks"] == info_n["row_blocks"], (info, info_n)
    assert_bitexact(got, got_n, "candidate turned down against the plan without block tiles")
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="scalar mesh", nterms=30)


@pytest.mark.parametrize("spec", ["synthetic:queen:14,12,10,3,20", "synthetic:queen:14,12,10,3,300", "synthetic:queen:14,12,10,3,0,50",
                                  "synthetic:queen:14,12,10,6,20,97"])
def test_queen_twins_through_the_context(oracle, spec):
    """The less tidy twins of the Queen_4147 stand-in (blocks with dropped entries, nodes with one or two unknowns) through
    the context API, CSR and COO uploads, against the oracle."""
    Q = hostapi.load(spec, "csr")
    rows, cols = Q.rows, Q.cols
    p, c, v = np.array(Q.row_ptr), np.array(Q.column_index), np.array(Q.value)
    Q.close()
    x = synth.x_vector(cols, seed=3)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    scale = abs_products(rows, p, c, v, x)
    _, info = run_plan(rows, cols, p, c, v, x, np.zeros(rows))
    assert info["masked_block_tiles"] > 0 and info["block_tiles"] >= 0.7 * info["row_blocks"], (spec, info)
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what=spec + " csr upload", nterms=110)
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        ctx.upload_coo(rows, cols, i - 1, j - 1, a)
        ctx.set_x(x)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what=spec + " coo upload", nterms=110)


def test_confirm_blocks_before_compress_gives_the_same_plan(oracle):
    """spmv_hip_plan_csr_confirm_blocks between plan and compress (what spmv_hip_upload_csr does): the candidate's tiles are cut on
    the row groups BEFORE they are classified, so that the classification runs once; the plan ends up with the same tiles and the
    same bits as when spmv_hip_plan_csr_repack finds the groups after the classification.  A no-op on a matrix without blocks."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows, cols, p, c, v = fem_ragged(5000, 20, 30, seed=21, drop=0.08, odd_every=61)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    got, infos = {}, {}
    for how in ("repack finds the groups", "confirm_blocks first", "confirm_blocks, row_ptr fetched back"):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, capi.FLAG_NO_VALUE_INDEX)
        if how != "repack finds the groups":
            before = plan.info()["row_blocks"]
            plan.confirm_blocks(tp.data_ptr(), tc.data_ptr(), p if how == "confirm_blocks first" else None, stream)
            assert plan.info()["row_blocks"] != before  # the tiles were cut anew
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        infos[how] = plan.info()
        ty = torch.from_numpy(y0.copy()).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        got[how] = ty.cpu().numpy()
        plan.close()
    ref = infos["repack finds the groups"]
    assert ref["masked_block_tiles"] > 0
    for how in got:
        assert (infos[how]["row_blocks"], infos[how]["block_tiles"], infos[how]["masked_block_tiles"]) == (ref["row_blocks"], ref["block_tiles"], ref["masked_block_tiles"]), how
        assert_bitexact(got[how], got["repack finds the groups"], how)
    assert_close(got["confirm_blocks first"], want, abs_products(rows, p, c, v, x) + np.abs(y0), what="confirm_blocks", nterms=93)
    # no candidate, nothing to do
    r2, c2n, p2, cc2, v2 = synth.poisson2d(80)
    plan = capi.CsrPlan(r2, c2n, p2, capi.CSR_AUTO, 0, 0)
    tp2, tc2 = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p2, cc2))
    before = plan.info()["row_blocks"]
    plan.confirm_blocks(tp2.data_ptr(), tc2.data_ptr(), p2, stream)
    assert plan.info()["row_blocks"] == before
    plan.close()


# ---- round 6: WIDE block tiles -- a tile whose columns span 64 K or more (an unstructured mesh numbered by reverse Cuthill-McKee has
# a band of ~7 n^(2/3) nodes; numbered at random, of the whole matrix) has no 16-bit columns, but its 3 x 3 blocks are blocks all the
# same: masked block words carry 22 bits of column since round 6 --------------------------------------------------------------------
def _scatter_nodes(rows, cols, p, c, v, factor, seed):
    """The same blocks with their COLUMN nodes scattered over `factor` times as many nodes (a rectangular matrix: rows x factor * cols):
    column node m -> a sorted random subset's m-th element, so that rows stay ascending and the triples stay aligned."""
    rng = np.random.default_rng(seed)
    nodes = cols // 3
    pick = np.sort(rng.choice(nodes * factor, size=nodes, replace=False)).astype(np.int64)
    c2 = (3 * pick[c // 3] + c % 3)
    return rows, cols * factor, p, c2.astype(np.int32), v


@pytest.mark.parametrize("name,make,expect_wide", [
    ("delaunay 3 dof, 30 K points in random order", lambda: synth.delaunay_mesh(30000, 3, seed=5, order="random"), True),
    ("delaunay 3 dof, 60 K points, rcm", lambda: synth.delaunay_mesh(60000, 3, seed=6), None),  # narrow and wide tiles side by side
    ("fem3 with its column nodes spread over 3 M columns", lambda: _scatter_nodes(*fem3(6000, 20, 34, seed=9), factor=170, seed=1), True),
    ("fem3 with its column nodes spread over 9 M columns", lambda: _scatter_nodes(*fem3(6000, 20, 34, seed=9), factor=500, seed=2), None),  # some tiles fit 22 bits, some do not
    ("fem3 with its column nodes spread over 54 M columns", lambda: _scatter_nodes(*fem3(6000, 20, 34, seed=9), factor=3000, seed=2), False)])
def test_wide_block_tiles(oracle, name, make, expect_wide):
    """Whole vector against the oracle; the tiles are block tiles although they have no 16-bit columns (masked words, 22 bits of
    column) -- up to a span of 4 M columns, beyond which a tile keeps its 32-bit column indices (and the same y)."""
    rows, cols, p, c, v = make()
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    # (the scattered twins have an x of 24 / 72 MB and columns all over it: the plan would multiply a column-panel copy, whose launch
    # ignores the block marks -- the panels are switched off to get at the tiles this test is about)
    base = capi.FLAG_NO_COLUMN_PANELS if name.startswith("fem3") else 0
    got, info = run_plan(rows, cols, p, c, v, x, y0, flags=base, index_values=False)
    assert info["panel_tiles"] == 0
    assert_close(got, want, scale, what=name)
    if expect_wide is True:
        assert info["narrow_tiles"] < 0.2 * info["row_blocks"], info
        assert info["masked_block_tiles"] > 0.8 * info["row_blocks"], (name, info["masked_block_tiles"], info["row_blocks"])
    elif expect_wide is False:
        assert info["block_tiles"] < 0.2 * info["row_blocks"], info  # spans beyond 22 bits: plain tiles with 32-bit columns
    elif not name.startswith("fem3"):
        assert info["block_tiles"] > 0.8 * info["row_blocks"], (name, info["block_tiles"], info["masked_block_tiles"], info["narrow_tiles"], info["row_blocks"])
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=base | capi.FLAG_NO_BLOCK_TILES, index_values=False)
    assert info_n["block_tiles"] == 0
    assert_close(got_n, want, scale, what=name + ", no block tiles")
    if expect_wide is True or not name.startswith("fem3"):
        # (the scattered twin's x -- 24 MB, read once -- is most of what its launch streams)
        assert info["streamed_bytes"] < (0.9 if name.startswith("fem3") else 0.85) * info_n["streamed_bytes"], (info["streamed_bytes"], info_n["streamed_bytes"])
    got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER, index_values=False)
    assert_bitexact(got_e, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=1), name + ", exact order")
    got3, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base, runs=3, index_values=False)
    assert_close(got3, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3), 3 * scale, what=name + ", three runs")
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base, out_of_place=True, index_values=False)
    assert_close(got_o, want, scale, what=name + ", y_out")
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, flags=base, other_columns=True, index_values=False)
    assert_close(got_c, want, scale, what=name + ", other column array")
    with capi.Context(0) as ctx:  # the context API (what the adapters use) builds the same plan
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what=name + ", context upload")


def _tril(rows, cols, p, c, v):
    import scipy.sparse as sp
    A = sp.tril(sp.csr_matrix((v, c, p), shape=(rows, cols)), format="csr")
    A.sort_indices()
    return rows, cols, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data


def _tril_dropped(rows, cols, p, c, v, drop, seed=1):
    """the stored lower triangle with a share of its off-diagonal entries missing (explicit zeros the assembly left out)"""
    import scipy.sparse as sp
    A = sp.tril(sp.csr_matrix((v, c, p), shape=(rows, cols)), format="csr")
    A.sort_indices()
    r = np.repeat(np.arange(rows), np.diff(A.indptr))
    keep = (A.indices == r) | (np.random.default_rng(seed).random(len(A.data)) >= drop)
    B = sp.csr_matrix((A.data[keep], (r[keep], A.indices[keep])), shape=(rows, cols))
    B.sort_indices()
    return rows, cols, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data


def _triu(rows, cols, p, c, v):
    import scipy.sparse as sp
    A = sp.triu(sp.csr_matrix((v, c, p), shape=(rows, cols)), format="csr")
    A.sort_indices()
    return rows, cols, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data


@pytest.mark.parametrize("name,make", [
    ("delaunay 3 dof, 60 K points, rcm, lower triangle", lambda: _tril(*synth.delaunay_mesh(60000, 3, seed=6))),
    ("delaunay 3 dof, 60 K points, rcm, upper triangle", lambda: _triu(*synth.delaunay_mesh(60000, 3, seed=6))),
    ("delaunay 3 dof, 30 K points, random order, lower triangle", lambda: _tril(*synth.delaunay_mesh(30000, 3, seed=5, order="random"))),
    # real files drop explicit zeros: a fifth to two thirds of the triples still go exactly (m + 1, m + 2, m + 3) -- the hint's peak
    ("delaunay 3 dof, 60 K points, rcm, lower triangle, 0.5 % dropped", lambda: _tril_dropped(*synth.delaunay_mesh(60000, 3, seed=6), 0.005)),
    ("delaunay 3 dof, 60 K points, rcm, lower triangle, 2 % dropped", lambda: _tril_dropped(*synth.delaunay_mesh(60000, 3, seed=6), 0.02)),
    ("queen-like 30 x 24 x 20, lower triangle", lambda: (lambda M: (M.rows, M.cols, np.array(M.row_ptr), np.array(M.column_index), np.array(M.value)))(
        hostapi.load("synthetic:queen:30,24,20:tril", "csr"))),
    ("delaunay 6 dof, 25 K points", lambda: synth.delaunay_mesh(25000, 6, seed=7))])
def test_stored_triangles_of_unstructured_block_matrices(oracle, name, make):
    """Round 6: the STORED TRIANGLE of a matrix of 3 x 3 blocks -- what the reference multiplies from a `symmetric` file
    (src/matrix/matrix-market.cpp:530-555) -- for an unstructured mesh: the rows of a node are (m + 1, m + 2, m + 3) entries long, m
    anything from 0 to ~100.  spmv_hip_plan_csr reads that off row_ptr (the skewed-triple hint), cuts its tiles by the block tile's
    limits instead of falling back to balanced tiles, and repack marks them as masked block tiles (triangular diagonal blocks, rows of
    1 ... 16 entries beside longer ones).  Whole vector against the oracle; and the meshes with 6 unknowns per node (6 x 6 blocks are
    3 x 3 blocks) that segment windows used to claim first."""
    rows, cols, p, c, v = make()
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0, index_values=False)
    assert_close(got, want, scale, what=name)
    assert info["balanced"] == 0, info
    assert info["block_tiles"] > 0.7 * info["row_blocks"], (name, info["block_tiles"], info["masked_block_tiles"], info["row_blocks"])
    assert info["block_entries"] > 0.9 * int(p[-1]), (name, info["block_entries"], int(p[-1]))
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES, index_values=False)
    assert info_n["block_tiles"] == 0
    assert_close(got_n, want, scale, what=name + ", no block tiles")
    got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER, index_values=False)
    assert_bitexact(got_e, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=1), name + ", exact order")
    got3, _ = run_plan(rows, cols, p, c, v, x, y0, runs=3, index_values=False)
    assert_close(got3, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3), 3 * scale, what=name + ", three runs")
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what=name + ", context upload")


def test_skewed_triples_that_are_no_blocks_fall_back(oracle):
    """Row lengths that go (m + 1, m + 2, m + 3) without any block structure behind them (random columns): the hint is taken, repack
    finds no blocks, the tiles are cut anew without it -- and y is the oracle's either way."""
    rng = np.random.default_rng(11)
    nodes = 4000
    m = 3 * rng.integers(2, 30, size=nodes)
    lens = (m[:, None] + np.arange(1, 4)[None, :]).ravel()
    rows = cols = 3 * nodes
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) for n in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, size=len(c))
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p.astype(np.int32), c, v, x, y0, index_values=False)
    assert info["block_tiles"] == 0
    assert_close(got, oracle.csr_spmv(rows, p.astype(np.int32), c, v, x, y=y0, num_threads=4), abs_products(rows, p.astype(np.int32), c, v, x) + np.abs(y0),
                 what="skewed triples without blocks")
