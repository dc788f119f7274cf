"""Row-group tiles (csr_rowgroup.hpp; opt-in, SPMV_HIP_FLAG_ROW_GROUPS -- measured slower than the default path, kept for the
record and for matrices it may suit): the interior of a stencil or band whose rows hold 17 ... 64 entries is multiplied by
2 ... 8 lanes per row, each adding up to twelve consecutive entries in registers; the plan keeps a list of those tiles and one
of the others (a second launch of csr_wavetile_kernel's x-window variant).

Checked here against the oracle (src/matrix/csr-matrix-spmv.cpp:21-33 restated in oracle/spmv_oracle.c): EVERY row length
17 ... 64 with first-row columns in a few runs (window of runs) and in one band (contiguous window), the same plan without the
flag (the default path: the two must agree to the contract, and both with the oracle), accumulation, y_out != y_in, another column array at spmv
time (nothing derived may be used), rows of <= 16 and >= 65 entries (never row-group tiles), a matrix whose tiles are mostly
NOT of this kind (no second launch: the plan stays as it was), the exact-order flag (bit-exact, no row groups), and the ragged
end of the value array (the last tiles stay with csr_wavetile_kernel: no read past the end)."""
import numpy as np
import pytest

from helpers import assert_bitexact, assert_close, abs_products
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu


def shifted_rows(n, offsets, seed, tail=0):
    """Row i holds the columns i + offsets (sorted, unique) that fall inside [0, n): the interior rows are copies of each other
    moved along the diagonal; `tail` extra rows of 3 entries follow (a ragged end)."""
    offsets = np.unique(np.asarray(offsets, dtype=np.int64))
    rng = np.random.default_rng(seed)
    cols = np.arange(n, dtype=np.int64)[:, None] + offsets[None, :]
    ok = (cols >= 0) & (cols < n)
    lens = ok.sum(axis=1)
    if tail:
        lens = np.concatenate([lens, np.full(tail, 3, dtype=lens.dtype)])
    p = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    c = cols[ok]
    if tail:
        c = np.concatenate([c, (np.arange(tail)[:, None] * 7 % n + np.arange(3)[None, :]).ravel()])
    v = rng.uniform(-1.0, 1.0, size=len(c))
    rows = n + tail
    return rows, max(n, rows), p.astype(np.int32), c.astype(np.int32), v


def run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_ROW_GROUPS, runs=1, other_columns=False, out_of_place=False):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags | capi.FLAG_NO_VALUE_INDEX)
    plan.compress(tc.data_ptr(), stream)
    plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
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


def offsets_in_runs(length, seed):
    """`length` offsets in 1 ... 9 runs of consecutive columns, the runs up to a few thousand columns apart (a stencil in two or
    three dimensions seen from one row)."""
    rng = np.random.default_rng(seed)
    # (the window of runs holds `length + runs * (rows - 1)` entries of x: at most 256, and used twice on average)
    rows = 512 // length
    most = max(1, (min(250, rows * length // 2) - length) // (rows - 1))
    nruns = int(rng.integers(1, min(9, most) + 1))
    cuts = np.sort(rng.choice(np.arange(1, length), size=nruns - 1, replace=False)) if nruns > 1 else np.array([], dtype=np.int64)
    sizes = np.diff(np.concatenate([[0], cuts, [length]]))
    starts = np.sort(rng.choice(np.arange(-4000, 4000, 80), size=nruns, replace=False))
    return np.concatenate([s + np.arange(k) for s, k in zip(starts, sizes)])


@pytest.mark.parametrize("length", list(range(17, 65)))
def test_every_row_length_against_oracle(oracle, length):
    n = 24000
    for kind in ("runs", "band"):
        offsets = offsets_in_runs(length, seed=length) if kind == "runs" else np.arange(length) - length // 2
        rows, cols, p, c, v = shifted_rows(n, offsets, seed=100 + length)
        x = synth.x_vector(cols, seed=3)
        y0 = synth.x_vector(rows, seed=4)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        what = "%d per row, %s" % (length, kind)
        got, info = run_plan(rows, cols, p, c, v, x, y0)
        assert info["row_group_tiles"] > 0.5 * info["row_blocks"], (what, info)
        assert_close(got, want, scale, what=what, nterms=length)
        got_n, info_n = run_plan(rows, cols, p, c, v, x, y0, flags=0)
        assert info_n["row_group_tiles"] == 0
        assert_close(got_n, want, scale, what=what + ", no row groups", nterms=length)


@pytest.mark.parametrize("length,kind", [(27, "runs"), (33, "runs"), (45, "band"), (64, "runs"), (17, "band")])
def test_accumulate_out_of_place_other_columns_exact_order(oracle, length, kind):
    n = 30000
    offsets = offsets_in_runs(length, seed=7 * length) if kind == "runs" else np.arange(length) - 3
    rows, cols, p, c, v = shifted_rows(n, offsets, seed=length)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    what = "%d per row, %s" % (length, kind)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["row_group_tiles"] > 0
    assert_close(got, want, scale, what=what, nterms=length)
    got2, _ = run_plan(rows, cols, p, c, v, x, y0, runs=2)
    assert_close(got2, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * scale, what=what + ", two runs", nterms=2 * length)
    got_o, _ = run_plan(rows, cols, p, c, v, x, y0, out_of_place=True)
    assert np.array_equal(got_o.view(np.uint64), got.view(np.uint64)), what + ": y_out differs from the in-place result"
    got_c, _ = run_plan(rows, cols, p, c, v, x, y0, other_columns=True)
    assert_close(got_c, want, scale, what=what + ", other column array", nterms=length)
    got_e, info_e = run_plan(rows, cols, p, c, v, x, y0, flags=capi.FLAG_ROW_GROUPS | capi.FLAG_EXACT_ORDER)
    assert info_e["row_group_tiles"] == 0
    assert_bitexact(got_e, want, what + ", exact order")
    # the same bits on every run (no atomics on this path)
    again, _ = run_plan(rows, cols, p, c, v, x, y0)
    assert np.array_equal(again.view(np.uint64), got.view(np.uint64)), what


@pytest.mark.parametrize("length", [5, 16, 65, 81])
def test_other_row_lengths_are_left_alone(oracle, length):
    rows, cols, p, c, v = shifted_rows(20000, np.arange(length) * 3 - length, seed=length)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["row_group_tiles"] == 0, info
    if length <= 16:
        assert_bitexact(got, want, "%d per row" % length)
    else:
        assert_close(got, want, abs_products(rows, p, c, v, x) + np.abs(y0), what="%d per row" % length, nterms=length)


def test_minority_of_tiles_keeps_the_plan(oracle):
    """A third of the rows are a 27-entry stencil interior, the others irregular: no row-group launch."""
    rng = np.random.default_rng(5)
    n = 30000
    rows, cols, p, c, v = shifted_rows(n, offsets_in_runs(27, seed=1), seed=2)
    p = p.astype(np.int64)
    keep = np.ones(len(c), dtype=bool)
    for r in range(n // 3, n):  # drop one random entry of every later row: no two rows alike
        keep[p[r] + rng.integers(0, p[r + 1] - p[r])] = False
    lens = np.add.reduceat(keep.astype(np.int64), p[:-1])
    p2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=p2[1:])
    c2, v2 = c[keep], v[keep]
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p2.astype(np.int32), c2, v2, x, y0)
    assert info["row_group_tiles"] == 0, info
    want = oracle.csr_spmv(rows, p2.astype(np.int32), c2, v2, x, y=y0, num_threads=4)
    assert_close(got, want, abs_products(rows, p2.astype(np.int32), c2, v2, x) + np.abs(y0), what="minority", nterms=27)


def test_mixed_lengths_and_ragged_end(oracle):
    """Blocks of rows of 27 and 33 entries (the KKT-like matrix's two kinds) and a ragged end of short rows right behind the
    last stencil tile: the tiles within 16 entries of the end of the value array stay with the other kernel."""
    n = 26000
    a = shifted_rows(n, offsets_in_runs(27, seed=3), seed=1)
    b = shifted_rows(n, offsets_in_runs(33, seed=4), seed=2, tail=2)
    rows = a[0] + b[0]
    cols = max(a[1], b[1])
    p = np.concatenate([a[2].astype(np.int64), b[2][1:].astype(np.int64) + int(a[2][-1])]).astype(np.int32)
    c = np.concatenate([a[3], b[3]])
    v = np.concatenate([a[4], b[4]])
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert 0 < info["row_group_tiles"] < info["row_blocks"], info
    assert_close(got, want, abs_products(rows, p, c, v, x) + np.abs(y0), what="27 and 33 per row", nterms=33)
    # no tail at all: the matrix ends with a stencil tile
    rows, cols, p, c, v = shifted_rows(n, np.arange(27) - 13, seed=9)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    got, info = run_plan(rows, cols, p, c, v, x, y0)
    assert info["row_group_tiles"] > 0
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4), abs_products(rows, p, c, v, x) + np.abs(y0),
                 what="band of 27 to the last row", nterms=27)
