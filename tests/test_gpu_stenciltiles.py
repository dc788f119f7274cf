"""Masked stencil tiles (csr_stenciltile.hpp): the boundary rows of a structured grid -- rows whose columns are a subset of the
interior rows' stencil -- keep the lane-per-row, no-column-index path: a 16-bit mask per row instead of column indices and
row_ptr.  One lane adds a row left to right in column order, so everything here is BIT-EXACT against the oracle
(src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c): 3-D 7-point grids with short lines (every tile of 73
rows holds the end of a grid line), 2-D 5- and 9-point grids, a 3-D 13-point star, grids with holes (Dirichlet cells removed),
coefficient dictionaries (the dictionary launch meets the same tiles), accumulation, y_out != y_in, another column array (nothing
derived may be used), exact order, the context uploads (CSR, COO), and tiles that must NOT be taken (a foreign column in a
boundary row)."""
import numpy as np
import pytest

from helpers import assert_bitexact
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu


def grid_stencil(shape, offsets, seed=1, hole_share=0.0, values="random"):
    """Cells of a grid of the given shape (2 or 3 dims), one unknown each, the given neighbour offsets where the neighbour exists
    (and is no hole): ascending columns."""
    rng = np.random.default_rng(seed)
    dims = len(shape)
    idx = np.indices(shape).reshape(dims, -1)
    n = idx.shape[1]
    strides = np.array([int(np.prod(shape[d + 1:])) for d in range(dims)])
    hole = rng.random(n) < hole_share
    cols, ok = [], []
    for off in sorted(offsets, key=lambda o: int(np.dot(o, strides))):
        nb = idx + np.array(off)[:, None]
        good = np.all((nb >= 0) & (nb < np.array(shape)[:, None]), axis=0)
        c = np.where(good, (nb * strides[:, None]).sum(axis=0), 0)
        good &= ~hole[c]
        cols.append(c)
        ok.append(good)
    cols, ok = np.stack(cols, axis=1), np.stack(ok, axis=1)
    ok[:, [int(np.dot(o, strides)) == 0 for o in sorted(offsets, key=lambda o: int(np.dot(o, strides)))]] = True  # the diagonal stays
    lens = ok.sum(axis=1)
    p = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = cols[ok].astype(np.int32)
    if values == "random":
        v = rng.uniform(-1.0, 1.0, size=len(c))
    else:  # constant coefficients: the diagonal 2 * dims, the others -1 (a dictionary of two values)
        v = np.where(c == np.repeat(np.arange(n), lens), 2.0 * dims, -1.0)
    return n, n, p.astype(np.int32), c, v


def run_plan(rows, cols, p, c, v, x, y0, flags=0, runs=1, other_columns=False, out_of_place=False, index_values=True,
             index_values_first=False):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
    plan.compress(tc.data_ptr(), stream)
    if index_values and index_values_first:  # (compress, index_values, repack: an order the header allows)
        plan.index_values(tv.data_ptr(), stream)
    plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    if index_values and not index_values_first:
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


STAR7 = [(0, 0, 0), (0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)]
STAR13 = STAR7 + [(0, 0, 2), (0, 0, -2), (0, 2, 0), (0, -2, 0), (2, 0, 0), (-2, 0, 0)]
FIVE = [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)]
NINE = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]


@pytest.mark.parametrize("name,shape,offsets,holes,values", [
    ("7-point 40^3", (40, 40, 40), STAR7, 0.0, "random"),
    ("7-point 33 x 47 x 29", (33, 47, 29), STAR7, 0.0, "random"),
    ("7-point 64^3, constant coefficients", (64, 64, 64), STAR7, 0.0, "constant"),
    ("13-point star 36^3", (36, 36, 36), STAR13, 0.0, "random"),
    ("5-point 300 x 210", (300, 210), FIVE, 0.0, "random"),
    ("9-point 257 x 129", (257, 129), NINE, 0.0, "random"),
    ("7-point 40^3 with 3 % of the cells removed", (40, 40, 40), STAR7, 0.03, "random"),
    ("5-point 200^2, constant coefficients, 1 % removed", (200, 200), FIVE, 0.01, "constant")])
def test_masked_stencil_tiles_bitexact(oracle, name, shape, offsets, holes, values):
    rows, cols, p, c, v = grid_stencil(shape, offsets, seed=len(name), hole_share=holes, values=values)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["stencil_mask_tiles"] > 0, (name, info)
    # with them (almost) every tile reads no column index: shifted tiles proper + masked ones
    assert info["shifted_tiles"] + info["stencil_mask_tiles"] >= 0.9 * info["row_blocks"], (name, info)
    assert_bitexact(got, want, name)
    got_v, info_v = run_plan(rows, cols, p, c, v, x, y0, index_values=False)
    assert_bitexact(got_v, want, name + ", values read")
    got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_SHIFTED_TILES, index_values=False)
    assert info_n["stencil_mask_tiles"] == 0
    assert_bitexact(got_n, want, name + ", no shifted tiles")
    assert info_v["streamed_bytes"] < info_n["streamed_bytes"]  # (both with the values read: no column index, no row_ptr)
    got_e, _ = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_EXACT_ORDER)
    assert_bitexact(got_e, want, name + ", exact order")
    got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=3)
    assert_bitexact(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3), name + ", three runs")
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    assert_bitexact(got_o, want, name + ", y_out")
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    assert_bitexact(got_c, want, name + ", other column array")


@pytest.mark.parametrize("name,shape,offsets,holes", [
    ("7-point 64^3", (64, 64, 64), STAR7, 0.0),
    ("7-point 33 x 47 x 29", (33, 47, 29), STAR7, 0.0),
    ("5-point 200^2, 1 % removed", (200, 200), FIVE, 0.01),
    ("5-point 300 x 210", (300, 210), FIVE, 0.0)])
def test_dictionary_built_before_the_repack_that_marks_stencil_tiles(oracle, name, shape, offsets, holes):
    """compress, index_values, THEN repack on a constant-coefficient grid.  The dictionary (with its
    constant-row tile list) is built while the boundary tiles still hold 16-bit columns; repack then turns those slots into row
    masks.  The dictionary must not survive that: repack drops it and builds it again on the marked tiles, and both orders end in
    the same plan and the reference's bits."""
    rows, cols, p, c, v = grid_stencil(shape, offsets, seed=len(name), hole_share=holes, values="constant")
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2)
    got_d, info_d = run_plan(rows, cols, p, c, v, x, y0, runs=2)
    got_f, info_f = run_plan(rows, cols, p, c, v, x, y0, runs=2, index_values_first=True)
    assert info_d["stencil_mask_tiles"] > 0 and info_d["indexed_values"] == 2, info_d
    for k in ("stencil_mask_tiles", "indexed_values", "row_blocks", "value_row_tiles", "dictionary_launch_tiles", "streamed_bytes"):
        assert info_f[k] == info_d[k], (name, k, info_f[k], info_d[k])
    assert_bitexact(got_d, want, name + ", default order")
    assert_bitexact(got_f, want, name + ", index_values before repack")


def test_a_foreign_column_keeps_its_tile_out(oracle):
    """One boundary row gets a column that is no neighbour of the stencil: its tile must keep its column indices (and the same y),
    the other boundary tiles are taken as before."""
    rows, cols, p, c, v = grid_stencil((40, 40, 40), STAR7, seed=5)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    _, clean = run_plan(rows, cols, p, c, v, x, y0, index_values=False)
    c2 = c.copy()
    r = (7 * 40 + 5) * 40 + 39  # the last cell of an interior grid line: 6 entries
    assert p[r + 1] - p[r] == 6
    c2[p[r + 1] - 1] = min(cols - 1, c2[p[r + 1] - 1] + 5)  # its last column moved: ascending still, no stencil neighbour
    got, info = run_plan(rows, cols, p, c2, v, x, y0, index_values=False)
    assert clean["stencil_mask_tiles"] - 1 == info["stencil_mask_tiles"], (clean["stencil_mask_tiles"], info["stencil_mask_tiles"])
    assert_bitexact(got, oracle.csr_spmv(rows, p, c2, v, x, y=y0, num_threads=4), "foreign column")


def test_context_uploads_of_a_short_line_grid(oracle):
    rows, cols, p, c, v = grid_stencil((48, 40, 36), STAR7, seed=9)
    x = synth.x_vector(cols, seed=3)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.run()
        assert_bitexact(ctx.get_y(), want, "csr upload")
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        ctx.upload_coo(rows, cols, i - 1, j - 1, a)
        ctx.set_x(x)
        ctx.run()
        assert_bitexact(ctx.get_y(), want, "coo upload")
