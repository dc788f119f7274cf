"""Group tiles (csr_blocktile.hpp, late in round 5): a mesh with 2 or 4 unknowns per node has no 3 x 3 blocks, but its rows come in
groups of 2 or 4 with identical column lists.  Such a tile reads ONE 16-bit column list and gathers x ONCE per group; the products
land where the plain tile would have put them and the row sums are the plain tile's (same lanes per row, same order) -- but the
hint cuts the tiles on group boundaries, so a plan without it has other tiles and, for rows of more than 16 entries, other lanes per
row: every test compares with the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c) under the contract's
tolerance, the plan without group tiles (SPMV_HIP_FLAG_NO_BLOCK_TILES) likewise, and the variants of ONE plan bitwise.  Covered: 2, 4 and 8 unknowns per node (8: rows of 216 entries, groups of 2 -- four such rows do
not fit a tile), 6 (3 x 3 blocks, not groups), groups that start at row 1 (a rank's row block), groups broken by a foreign column or a missing entry (their
tiles stay plain), rows too long for a tile, another column array, y_out != y_in, accumulation, exact order, a value dictionary
(no group tiles then), and the context's uploads."""
import numpy as np
import pytest

from helpers import assert_close, abs_products
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_plan(rows, cols, p, c, v, x, y0, flags=0, runs=1, other_columns=False, out_of_place=False, index_values=False):
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
        ty = tout
    else:
        for _ in range(runs):
            plan.spmv(tp.data_ptr(), cols_now.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    got = ty.cpu().numpy()
    plan.close()
    return got, info


def same_bits(a, b, what):
    assert np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64)), what


@pytest.mark.parametrize("grid,d,group_rows", [((30, 24, 20), 2, 2), ((24, 20, 16), 4, 4), ((16, 14, 12), 8, 2)])
def test_group_tiles_against_the_oracle(oracle, grid, d, group_rows):
    rows, cols, p, c, v = synth.mesh_dofs(grid, d, seed=d)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    what = "mesh %s with %d unknowns per node" % ("x".join(map(str, grid)), d)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["group_rows"] == group_rows and info["block_tiles"] == 0, info
    assert info["group_tiles"] > (0.8 if d < 8 else 0.5) * info["row_blocks"], (what, info)  # (d = 8: the boundary nodes' shorter rows come three to a tile)
    plain, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info_n["group_tiles"] == 0 and info_n["group_rows"] == 0
    assert info["streamed_bytes"] < info_n["streamed_bytes"]
    assert_close(got, want, scale, what=what, nterms=27 * d + d)
    assert_close(plain, want, scale, what=what + ", no group tiles", nterms=27 * d + d)
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    same_bits(got_c, got, what + ", other column array")
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    same_bits(got_o, got, what + ", y_out")
    got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
    assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=2 * (27 * d + d))
    got_e, info_e = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    assert info_e["group_tiles"] == 0
    same_bits(got_e, want, what + ", exact order")


def test_six_unknowns_per_node_are_block_tiles_not_groups(oracle):
    rows, cols, p, c, v = synth.mesh_dofs((16, 14, 12), 6, seed=6)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["group_tiles"] == 0 and info["block_tiles"] > 0.8 * info["row_blocks"], info
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="6 dofs", nterms=200)


@pytest.mark.parametrize("d", [2, 4])
def test_groups_that_start_at_another_row(oracle, d):
    """A rank's row block: the grid of groups starts at row 1 (d = 2) / row 3 (d = 4)."""
    rows, cols, p, c, v = synth.mesh_dofs((24, 20, 16), d, seed=11)
    first = 1 if d == 2 else 3
    p2 = (p[first:] - p[first]).astype(np.int32)
    c2, v2 = c[p[first]:], v[p[first]:]
    r2 = rows - first
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(r2, seed=4)
    got, info = run_plan(r2, cols, p2, c2, v2, x, y0)
    assert info["group_rows"] == d and info["group_tiles"] > 0.8 * info["row_blocks"], info
    assert_close(got, oracle.csr_spmv(r2, p2, c2, v2, x, y=y0, num_threads=4), abs_products(r2, p2, c2, v2, x) + np.abs(y0), what="offset", nterms=200)


@pytest.mark.parametrize("d", [2, 4])
def test_broken_groups_keep_their_tiles_plain(oracle, d):
    """Some groups get a foreign column in one of their rows (same length, other column), some rows lose an entry (the group's rows
    differ in length): those tiles must not be marked, the others are, and y is the plain plan's."""
    rows, cols, p, c, v = synth.mesh_dofs((24, 20, 16), d, seed=21)
    rng = np.random.default_rng(5)
    c = c.copy()
    lens = np.diff(p)
    # a foreign column: the last entry of a group's SECOND row moved up by one where that stays ascending and in range
    nodes = rng.choice(rows // d, size=rows // d // 50, replace=False)
    changed = 0
    for n in nodes:
        r = int(n) * d + 1
        k = p[r + 1] - 1
        if c[k] + 1 < cols:
            c[k] += 1
            changed += 1
    assert changed > 10
    # rows that lose their last entry
    drop = np.zeros(len(c), dtype=bool)
    for n in rng.choice(rows // d, size=rows // d // 80, replace=False):
        drop[p[int(n) * d + d - 1 + 1] - 1] = True
    keep = ~drop
    lens2 = lens.copy()
    rr = np.repeat(np.arange(rows), lens)
    lens2 -= np.bincount(rr[drop], minlength=rows)
    p2 = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens2, out=p2[1:])
    c2, v2 = c[keep], v[keep]
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p2, c2, v2, x, y0)
    assert info["group_rows"] == d, info
    assert 0.3 * info["row_blocks"] < info["group_tiles"] < info["row_blocks"], info
    assert_close(got, oracle.csr_spmv(rows, p2, c2, v2, x, y=y0, num_threads=4), abs_products(rows, p2, c2, v2, x) + np.abs(y0), what="broken groups", nterms=200)


def test_rows_in_pairs_without_shared_columns_are_no_groups(oracle):
    """The hint from row_ptr holds (rows in pairs of equal, even length) but the columns differ: no tile is marked, and a hint that cut
    tiles shorter is taken back."""
    rng = np.random.default_rng(9)
    rows, cols = 40000, 40000
    lens = np.repeat(2 * rng.integers(9, 30, size=rows // 2), 2)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    base = np.repeat(np.arange(rows), lens)
    off = np.concatenate([np.sort(rng.choice(4000, size=int(n), replace=False)) for n in lens])
    c = (np.clip(base - 2000, 0, cols - 4000) + off).astype(np.int32)  # ascending and distinct within a row, different from row to row
    v = rng.uniform(-1, 1, size=len(c))
    p = p.astype(np.int32)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["group_tiles"] == 0 and info["group_rows"] == 0, info
    plain, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    same_bits(got, plain, "pairs without shared columns")
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0), what="pairs", nterms=100)


def test_a_value_dictionary_takes_precedence_and_context_uploads(oracle):
    rows, cols, p, c, v = synth.mesh_dofs((24, 20, 16), 2, seed=31)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    vq = np.round(v * 4) / 4 + 0.125  # nine distinct values: the dictionary applies
    want = oracle.csr_spmv(rows, p, c, vq, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, vq, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, vq, x, y0, index_values=True)
    assert info["indexed_values"] > 0 and info["group_tiles"] == 0, info  # (reported as 0 under a dictionary: its launch reads the 16-bit columns)
    assert_close(got, want, scale, what="dictionary", nterms=100)
    want0 = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.run()
        assert_close(ctx.get_y(), want0, scale, what="csr upload", nterms=100)
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        ctx.upload_coo(rows, cols, i - 1, j - 1, a)
        ctx.set_x(x)
        ctx.run()
        assert_close(ctx.get_y(), want0, scale, what="coo upload", nterms=100)


@pytest.mark.parametrize("name,make,d", [
    ("delaunay 2 dof, 40 K points in random order (every tile wide)", lambda: synth.delaunay_mesh(40000, 2, seed=8, order="random"), 2),
    ("delaunay 2 dof, 120 K points, rcm (narrow and wide tiles)", lambda: synth.delaunay_mesh(120000, 2, seed=9), 2),
    ("delaunay 4 dof, 25 K points in random order", lambda: synth.delaunay_mesh(25000, 4, seed=10, order="random"), 4),
    ("delaunay 2-d triangles, 2 dof, 60 K points in random order", lambda: synth.delaunay_mesh(60000, 2, seed=11, order="random", dim=2), 2)])
def test_wide_group_tiles(oracle, name, make, d):
    """Round 6: a tile whose columns span 64 K or more has no 16-bit columns -- its group columns are kept as 32-bit absolute columns
    in the tile's own (unused) slots of the 16-bit stream.  Unstructured meshes with 2 / 4 unknowns per node (dense 2 x 2 / 4 x 4
    blocks: one column per pair of adjacent columns): whole vector against the oracle, variants of one plan bitwise."""
    rows, cols, p, c, v = make()
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert_close(got, want, scale, what=name)
    if "2-d triangles" not in name:  # (7 neighbours per node: rows of 14 entries, mostly below the 16 a group tile starts at)
        assert info["group_rows"] == d and info["group_tiles"] > 0.7 * info["row_blocks"], (name, info["group_tiles"], info["narrow_tiles"], info["row_blocks"])
        if "random order" in name:
            assert info["narrow_tiles"] < 0.2 * info["row_blocks"], info
    plain, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BLOCK_TILES)
    assert info_n["group_tiles"] == 0
    assert_close(plain, want, scale, what=name + ", no group tiles")
    if info["group_tiles"] > 0.7 * info["row_blocks"]:
        assert info["streamed_bytes"] < 0.92 * info_n["streamed_bytes"], (info["streamed_bytes"], info_n["streamed_bytes"])
    got3, _ = run_plan(rows, cols, p, c, v, x, y0, runs=3)
    assert_close(got3, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3), 3 * scale, what=name + ", three runs")
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    same_bits(got_o, got, name + ", y_out")
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    assert_close(got_c, want, scale, what=name + ", other column array")
    got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    same_bits(got_e, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=1), name + ", exact order")
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what=name + ", context upload")
