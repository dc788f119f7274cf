"""Balanced tiles (csr_segtile_kernel): tiles filled by entries, row sums by segmented reduction.

Chosen by the plan for matrices whose row-owned tiles come out less than half full because the
rows are skewed.  Checked here against the oracle on matrices built to hit the corners of the
reduction: rows that begin / end exactly on a lane's 8-entry boundary, rows spanning many lanes,
runs of empty rows, single-entry rows, tiles capped by rows (256) and by entries (512), long rows
(own tile / split with atomics) in between, tiles that start off a 4-entry boundary, the ragged end
of the arrays; also the COO and hybrid uploads that now run through the same kernel.
"""
import numpy as np
import pytest

from helpers import assert_ell as helpers_assert_ell
from helpers import assert_bitexact, assert_close, abs_products
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu


def _csr_from_lens(lens, cols, rng, local=None):
    lens = np.asarray(lens, dtype=np.int64)
    rows = len(lens)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    Z = int(p[-1])
    r = np.repeat(np.arange(rows), lens)
    if local is None:
        c = rng.integers(0, cols, size=Z)
    else:
        c = np.clip(r * cols // max(rows, 1) + rng.integers(-local, local + 1, size=Z), 0, cols - 1)
    order = np.lexsort((c, r))
    c = c[order].astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=Z)
    return rows, cols, p.astype(np.int32), c, v


def _multiply(rows, cols, p, c, v, x, y0, flags=0, compress=True, runs=1):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c if len(c) else np.zeros(4, np.int32), v if len(v) else np.zeros(4), x))
    if compress:
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    ty = torch.from_numpy(y0.copy()).to(dev)
    for _ in range(runs):
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    info = plan.info()
    plan.close()
    return ty.cpu().numpy(), info


CASES = {
    # most rows 1..5 entries, a long one every so often: what a web graph looks like
    "web": lambda rng: np.where(rng.random(40000) < 0.02, rng.integers(17, 400, 40000), rng.integers(1, 6, 40000)),
    # rows of exactly 8 and 16 entries (lane boundaries of the 8-entry shares) mixed with long ones
    "lane_aligned": lambda rng: np.where(rng.random(20000) < 0.12, 8 * rng.integers(3, 60, 20000), 8 * rng.integers(1, 3, 20000)),
    # runs of empty rows, single entries, and rows of 100-500
    "holes": lambda rng: np.where(rng.random(30000) < 0.6, 0, np.where(rng.random(30000) < 0.05, rng.integers(100, 513, 30000), 1)),
    # rows of 1 and 2 entries with a 300-entry row now and then: tiles capped by 256 rows
    "row_cap": lambda rng: np.where(rng.random(60000) < 0.002, 300, rng.integers(1, 3, 60000)),
    # long rows in between: exactly a tile, more than a tile (own wave), more than 2048 (split, atomics)
    "long_between": lambda rng: np.where(rng.random(40000) < 0.0015, rng.choice([509, 512, 513, 700, 2048, 2049, 5000], 40000), rng.integers(0, 7, 40000)),
    # odd lengths so tiles start off a 4-entry boundary all the time
    "odd_starts": lambda rng: np.where(rng.random(30000) < 0.04, 2 * rng.integers(10, 120, 30000) + 1, 2 * rng.integers(0, 3, 30000) + 1),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("seed", [1, 2])
def test_balanced_tiles_against_oracle(oracle, name, seed):
    rng = np.random.default_rng(1000 * seed + len(name))
    lens = CASES[name](rng)
    rows, cols, p, c, v = _csr_from_lens(lens, 50000, rng, local=None if seed == 1 else 20000)
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    got, info = _multiply(rows, cols, p, c, v, x, y0)
    assert info["balanced"] == 1, (name, info)
    assert_close(got, want, scale, what="balanced %s seed %d" % (name, seed))
    # far fewer tiles than the row-owned tiling of the same matrix, and the same y up to rounding
    got_r, info_r = _multiply(rows, cols, p, c, v, x, y0, flags=capi.FLAG_NO_BALANCED_TILES | capi.FLAG_NO_MULTI_WINDOW)  # (the plain row-owned tiling)
    assert info_r["balanced"] == 0 and info_r["row_blocks"] > 1.5 * info["row_blocks"], (info, info_r)
    assert_close(got_r, want, scale, what="row-owned %s seed %d" % (name, seed))
    # without index compression (32-bit columns) and accumulating twice
    got2, info2 = _multiply(rows, cols, p, c, v, x, y0, compress=False, runs=2)
    assert info2["balanced"] == 1
    want2 = oracle.csr_spmv(rows, p, c, v, x, y=want, num_threads=4)
    assert_close(got2, want2, 2 * scale, what="balanced uncompressed x2 %s" % name)
    # rows whose entries lie inside one lane's share are summed in the reference's order: with single-entry
    # and two-entry rows in the majority, most of y is bit-identical
    if name in ("holes", "row_cap"):
        same = got.view(np.uint64) == want.view(np.uint64)
        assert same.mean() > 0.7, same.mean()


def test_balanced_is_deterministic_and_not_chosen_for_regular_matrices(oracle):
    rng = np.random.default_rng(5)
    lens = CASES["web"](rng)
    rows, cols, p, c, v = _csr_from_lens(lens, 80000, rng)
    x = synth.x_vector(cols, seed=3)
    y0 = np.zeros(rows)
    a, info = _multiply(rows, cols, p, c, v, x, y0)
    b, _ = _multiply(rows, cols, p, c, v, x, y0)
    assert info["balanced"] == 1
    assert_bitexact(a, b, "two runs of the balanced kernel")
    for gen in (lambda: synth.poisson2d(300), lambda: synth.banded(40000, range(-40, 41), seed=2),
                lambda: synth.stencil27_like(30, 30, 30), lambda: _csr_from_lens(np.full(50000, 2), 50000, rng)):
        rows, cols, p, c, v = gen()
        _, info = _multiply(rows, cols, p, c, v, synth.x_vector(cols, seed=1), np.zeros(rows))
        assert info["balanced"] == 0, info
    # the exact-order flag keeps row-owned tiles (and the reference's bits) on a skewed matrix
    rows, cols, p, c, v = _csr_from_lens(lens, 80000, rng)
    got, info = _multiply(rows, cols, p, c, v, x, np.zeros(rows), flags=capi.FLAG_EXACT_ORDER)
    assert info["balanced"] == 0
    assert_bitexact(got, oracle.csr_spmv(rows, p, c, v, x, num_threads=4), "exact order on a skewed matrix")


@pytest.mark.parametrize("order", ["row", "column", "shuffled"])
def test_coo_upload_runs_as_row_major_tiles(oracle, order):
    """spmv_hip_upload_coo: triplets in any order are sorted by row once (stably) and multiplied through
    the CSR plan built from the device-side row_ptr; KEEP_ORDER keeps the atomic COO kernel."""
    rng = np.random.default_rng(9)
    lens = CASES["web"](rng)[:20000]
    rows, cols, p, c, v = _csr_from_lens(lens, 30000, rng)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    r, cc = (i - 1).astype(np.int32), (j - 1).astype(np.int32)
    perm = {"row": np.arange(len(a)), "column": np.lexsort((r, cc)), "shuffled": rng.permutation(len(a))}[order]
    r, cc, vv = np.ascontiguousarray(r[perm]), np.ascontiguousarray(cc[perm]), np.ascontiguousarray(a[perm])
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.coo_spmv(rows, r, cc, vv, x, y=y0, runs=2)
    scale = 2 * abs_products(rows, p, c, v, x) + np.abs(y0)
    for flags in (0, capi.FLAG_COO_KEEP_ORDER):
        with capi.Context(0, flags=flags) as ctx:
            ctx.upload_coo(rows, cols, r, cc, vv)
            info = ctx.info()
            assert (info["row_blocks"] > 0) == (flags == 0), info  # tiles only on the default path
            ctx.set_x(x)
            ctx.set_y(y0)
            ctx.run(2)
            assert_close(ctx.get_y(), want, scale, what="coo %s flags %x" % (order, flags))
    # bad indices are found on the device, and the context is left empty
    bad = r.copy()
    bad[len(bad) // 2] = rows
    with capi.Context(0) as ctx:
        with pytest.raises(capi.SpmvHipError) as e:
            ctx.upload_coo(rows, cols, bad, cc, vv)
        assert e.value.code == capi.ERR_INVALID
        assert ctx.info()["format"] == 0


def test_hybrid_upload_is_one_fused_multiply(oracle):
    """spmv_hip_upload_hybrid merges the ELL part (padding included) and the COO remainder into one
    row-major matrix on the device: one launch per run, same y as hybrid_matrix::spmv within 1e-10;
    NaN / Inf in x reach y through the padding exactly like in the reference (0.0 * Inf = NaN)."""
    rows, cols, p, c, v = synth.powerlaw(30000, 30000, seed=4, max_len=600)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    H = oracle.hybrid_from_coordinate(rows, i, j, a)
    assert H["row_length"] >= 1 and len(H["coo_val"]) > 0
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.hybrid_spmv(rows, H, x, y=y0, runs=2)
    scale = 2 * abs_products(rows, p, c, v, x) + np.abs(y0)
    with capi.Context(0) as ctx:
        ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
        info = ctx.info()
        assert info["format"] == 4 and info["row_blocks"] > 0
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run(2)
        assert_close(ctx.get_y(), want, scale, what="hybrid merged")
        # a non-finite x entry: every row whose ELL padding points at that column becomes NaN in the reference
        x2 = x.copy()
        hot = int(H["ell_col"][H["row_length"] - 1])  # the last (possibly padded) slot of row 0
        x2[hot] = np.inf
        ctx.set_x(x2)
        ctx.set_y(np.zeros(rows))
        ctx.run()
        got = ctx.get_y()
        ref = oracle.hybrid_spmv(rows, H, x2)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    # KEEP_ORDER: the two-launch path (ELL tiles, then the atomic COO kernel)
    with capi.Context(0, flags=capi.FLAG_COO_KEEP_ORDER) as ctx:
        ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run(2)
        assert_close(ctx.get_y(), want, scale, what="hybrid two launches")


def test_value_dictionary_bit_identical_and_guarded(oracle):
    """spmv_hip_plan_csr_index_values: matrices with at most 128 distinct values stream one byte per entry and take
    the double from a table.  Same bits as without; -0.0 / 0.0 and NaN payloads stay distinct; 129 values -> no
    dictionary; a changed value array is found (ERR_STATE) unless refresh_values follows the change; the context
    API builds the dictionary by itself for CSR, COO, ELLPACK and hybrid uploads."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(77)

    def run(rows, cols, p, c, v, x, y0, index=True, flags=0):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
        tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        if index:
            plan.index_values(tv.data_ptr(), stream)
        ty = torch.from_numpy(y0.copy()).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        return ty.cpu().numpy(), plan.info(), (plan, tp, tc, tv, tx)

    special = np.array([0.0, -0.0, 1.0, -1.0, 4.0, 1e-300, -2.5, np.inf, 3.0, 0.5])
    for name, gen, nvals in (("poisson", lambda: synth.poisson2d(200), 2),
                             ("band, 10 special values", lambda: synth.banded(30000, [-700, -300, -1, 0, 1, 500], seed=2), 10),
                             ("random columns, 128 values", lambda: synth.random_uniform(20000, 20000, 6, seed=5), 128),
                             ("random columns, 129 values", lambda: synth.random_uniform(20000, 20000, 6, seed=6), 129)):
        rows, cols, p, c, v = gen()
        if name != "poisson":
            pool = special if nvals == 10 else rng.uniform(-1, 1, size=nvals)
            v = pool[rng.integers(0, len(pool), size=len(v))]
            v[:len(pool)] = pool  # every value occurs
        x = synth.x_vector(cols, seed=3)
        if nvals == 10:
            x = np.abs(x) + 0.5  # inf * x stays inf: no NaN from inf - inf in a row with one infinite entry ... mostly
        y0 = synth.x_vector(rows, seed=4)
        got, info, keep = run(rows, cols, p, c, v, x, y0)
        ref, info0, _ = run(rows, cols, p, c, v, x, y0, index=False)
        assert info["indexed_values"] == (nvals if nvals <= 128 else 0), (name, info)
        assert info0["indexed_values"] == 0
        if nvals <= 128:
            assert info["streamed_bytes"] < info0["streamed_bytes"]
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), name  # NaN-safe bit comparison
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) or name.startswith("random"), name
        plan, tp, tc, tv, tx = keep
        if nvals == 2:
            # values changed in place without telling the plan: found on request (VERIFY_PLAN) ...
            plan2 = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, capi.FLAG_VERIFY_PLAN)
            plan2.compress(tc.data_ptr(), stream)
            plan2.index_values(tv.data_ptr(), stream)
            ty = torch.zeros(rows, dtype=torch.float64, device=dev)
            plan2.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            tv[5] = 7.25
            with pytest.raises(capi.SpmvHipError) as e:
                plan2.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            assert e.value.code == capi.ERR_STATE
            # ... and brought up to date by refresh_values (three values now)
            plan2.refresh_values(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            assert plan2.info()["indexed_values"] == 3
            ty.zero_()
            plan2.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            torch.cuda.synchronize()
            v2 = v.copy()
            v2[5] = 7.25
            assert_bitexact(ty.cpu().numpy(), oracle.csr_spmv(rows, p, c, v2, x, num_threads=4), "after refresh_values")
            plan2.close()
        plan.close()
    # the context API: dictionary for every format, same y as the oracle
    rows, cols, p, c, v = synth.poisson2d(150)
    x = synth.x_vector(cols, seed=9)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    with capi.Context(0) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        ctx.run()
        assert_bitexact(ctx.get_y(), want, "ctx csr with dictionary")
        assert ctx.info()["streamed_bytes"] < 6 * len(v) + 24 * rows
        ctx.upload_coo(rows, cols, i - 1, j - 1, a)
        ctx.set_x(x)
        ctx.run()
        assert_bitexact(ctx.get_y(), want, "ctx coo with dictionary")
        rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
        ctx.upload_ell(rows, cols, L, ec, ev)
        ctx.set_x(x)
        ctx.run()
        helpers_assert_ell(ctx.get_y(), oracle.ell_spmv(rows, L, ec, ev, x), L, 0, ec, ev, x, what="ctx ell with dictionary (padding zeros are values too)")
    with capi.Context(0, flags=capi.FLAG_NO_VALUE_INDEX) as ctx:
        ctx.upload_csr(rows, cols, p, c, v)
        assert ctx.info()["streamed_bytes"] > 8 * len(v)


def test_value_dictionary_on_short_row_stencils_bit_identical(oracle):
    """Shifted tiles with a value dictionary on stencils and bands whose tiles have 33 ... 128 rows, at the matrix
    boundaries, with 64- and 128-row tiles, with and without x windows: the oracle's bits every time."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    cases = [("poisson 5-pt 300^2", lambda: synth.poisson2d(300)), ("poisson 5-pt 67^2", lambda: synth.poisson2d(67)),
             ("tridiagonal", lambda: synth.banded(50000, [-1, 0, 1], seed=1)), ("diagonal", lambda: synth.banded(30000, [0], seed=2)),
             ("7-point 40^3", lambda: synth.banded(64000, [-1600, -40, -1, 0, 1, 40, 1600], seed=3)),
             ("8 diagonals", lambda: synth.banded(40000, [-900, -500, -200, -1, 0, 1, 300, 800], seed=4)),
             ("9 diagonals (not staged)", lambda: synth.banded(40000, [-900, -500, -200, -1, 0, 1, 300, 800, 1100], seed=5)),
             ("upper bidiagonal, x ends with the last run", lambda: synth.banded(20000, [0, 1], seed=6)),
             # one lane per row (EXACT_ORDER below) with rows longer than one round of gathers, and longer than a wave
             ("27 diagonals", lambda: synth.banded(30000, sorted(set(int(o) for o in np.linspace(-1300, 1300, 27))), seed=7)),
             ("70 diagonals", lambda: synth.banded(20000, list(range(-35, 35)), seed=8))]
    vals = np.array([-1.0, 4.0, 0.5, -0.25, 2.0, 1e-3, -7.0])
    rng = np.random.default_rng(5)
    for name, gen in cases:
        rows, cols, p, c, v = gen()
        v = vals[rng.integers(0, len(vals), size=len(v))]
        x = synth.x_vector(cols, seed=3)
        y0 = synth.x_vector(rows, seed=4)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
        out = {}
        for flags in (capi.FLAG_ROWS128, capi.FLAG_ROWS64, capi.FLAG_ROWS128 | capi.FLAG_NO_X_WINDOW,
                      capi.FLAG_EXACT_ORDER | capi.FLAG_NO_X_WINDOW):
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
            plan.compress(tc.data_ptr(), stream)
            plan.index_values(tv.data_ptr(), stream)
            ty = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            torch.cuda.synchronize()
            out[flags] = (ty.cpu().numpy(), plan.info())
            plan.close()
        for flags, (got, info) in out.items():
            what = "%s flags %x (dictionary %d)" % (name, flags, info["indexed_values"])
            if "diagonals" in name and int(name.split()[0]) > 16 and not (flags & capi.FLAG_EXACT_ORDER):
                # rows of more than 16 entries are summed by several lanes unless EXACT_ORDER asks for the reference's order
                assert_close(got, want, abs_products(rows, p, c, v, x) + np.abs(y0), what=what)
            else:
                assert_bitexact(got, want, what)
        assert out[capi.FLAG_ROWS128 | capi.FLAG_NO_X_WINDOW][1]["indexed_values"] == len(vals), name


def test_lane_per_row_stencil_tiles_fuzz(oracle):
    """The one-lane-per-row path of the value-dictionary kernel (uniform shifted tiles): bands of 1 ... 64 diagonals at
    random offsets, so that tiles have anything from 8 to 128 rows, start at any of the four positions inside an
    aligned quad, end ragged, and come with and without a shared pattern; rows of up to 16 entries take it by
    default, longer ones under EXACT_ORDER.  y must equal the oracle's bit for bit every time."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(2024)
    vals = np.array([1.0, -1.0, 4.0, 0.5, -0.125, 3.0, 1e-8, -2.0e5])
    took_path = 0
    for case in range(24):
        ndiag = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16, 17, 23, 32, 47, 63, 64]))
        rows = int(rng.integers(2500, 9000))
        reach = int(rng.integers(ndiag, max(ndiag + 1, rows // 3)))
        offs = np.sort(rng.choice(np.arange(-reach, reach + 1), size=ndiag, replace=False)).tolist()
        r, cols, p, c, v = synth.banded(rows, offs, seed=case)
        pool = vals[:(1, 2, 3, 8)[case % 4]]  # one or two values: selected from registers; more: looked up in the LDS table
        v = pool[rng.integers(0, len(pool), size=len(v))]
        x = synth.x_vector(cols, seed=case + 100)
        y0 = synth.x_vector(r, seed=case + 200)
        want = oracle.csr_spmv(r, p, c, v, x, y=y0, num_threads=2)
        tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
        for flags in ((capi.FLAG_EXACT_ORDER | capi.FLAG_NO_X_WINDOW), capi.FLAG_NO_X_WINDOW, 0):
            if ndiag > 16 and not (flags & capi.FLAG_EXACT_ORDER):
                continue  # several lanes per row: another path, another summation order
            for index in (True, False):  # with the dictionary, and the same scheme on the 8-byte values
                plan = capi.CsrPlan(r, cols, p, capi.CSR_AUTO, 0, flags)
                plan.compress(tc.data_ptr(), stream)
                if index:
                    plan.index_values(tv.data_ptr(), stream)
                info = plan.info()
                ty = torch.from_numpy(y0.copy()).to(dev)
                plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
                torch.cuda.synchronize()
                assert_bitexact(ty.cpu().numpy(), want, "case %d: %d diagonals, %d rows, flags %x, dictionary %s, %r" % (
                    case, ndiag, r, flags, index, info))
                if index and info["indexed_values"] > 0 and info["shifted_tiles"] > 0 and info["uniform_tiles"] > 0:
                    took_path += 1
                plan.close()
    assert took_path >= 20  # the cases did exercise dictionary + shifted + uniform tiles


def test_balanced_tiles_with_a_value_dictionary(oracle):
    """A graph as a pattern matrix (all ones) or with a few distinct weights: the balanced-tile kernel reads one index byte per
    entry (csr_segtile_kernel<VI>); 1e-10 against the oracle like the 8-byte path, and bit-identical to it where no row is cut
    into atomically added chunks."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, max_len in (("web graph", 3000), ("web graph without split rows", 400)):
        rows, cols, p, c, v = synth.powerlaw(40000, 40000, max_len=max_len, seed=21)
        lens = np.diff(p)
        for what, vals in (("pattern", np.ones(len(v))), ("five weights", np.array([0.5, 1.0, -2.0, 0.125, 3.0])[np.arange(len(v)) % 5]),
                           ("1/degree", np.repeat(1.0 / np.minimum(np.maximum(lens, 1), 90), lens))):
            x = synth.x_vector(cols, seed=22)
            y0 = synth.x_vector(rows, seed=23)
            tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, vals, x))
            got = {}
            for flags in (0, capi.FLAG_NO_VALUE_INDEX):
                plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
                plan.compress(tc.data_ptr(), stream)
                plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
                plan.index_values(tv.data_ptr(), stream)
                info = plan.info()
                assert info["balanced"] == 1, (name, info)
                assert info["indexed_values"] == (0 if flags else len(np.unique(vals))), (name, what, info)
                ty = torch.from_numpy(y0.copy()).to(dev)
                for _ in range(2):
                    plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
                torch.cuda.synchronize()
                got[flags] = ty.cpu().numpy()
                if not flags:
                    assert info["streamed_bytes"] < 12 * len(v) + 28 * rows, (name, info)
                plan.close()
            want = oracle.csr_spmv(rows, p, c, vals, x, y=y0, num_threads=4, runs=2)
            assert_close(got[0], want, 2 * abs_products(rows, p, c, vals, x) + np.abs(y0), what="%s, %s" % (name, what))
            if max_len <= 512:
                assert np.array_equal(got[0].view(np.uint64), got[capi.FLAG_NO_VALUE_INDEX].view(np.uint64)), (name, what)



def test_constant_row_tiles_read_no_value_stream(oracle):
    """Stencil tiles whose rows all carry the first row's dictionary indices (a constant-coefficient stencil) read only
    the first row's bytes (plan_info[23], kTileMetaValueRows).  Same bits as the indexed path and as the reference's
    loop; a single differing coefficient sends ITS tile back to the indexed path; refresh_values re-marks the tiles;
    a plan without a dictionary ignores the marks."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, (rows, cols, p, c, v) in (("5-point", synth.poisson2d(300)),
                                        ("9 diagonals, one value", synth.banded(40000, [-900, -301, -300, -1, 0, 1, 300, 301, 900], seed=3)),
                                        ("13 diagonals, three values", synth.banded(50000, [-2000, -700, -300, -2, -1, 0, 1, 2, 5, 300, 700, 2000, 2500], seed=4))):
        if name == "9 diagonals, one value":
            v = np.full(len(v), 0.375)
        elif name != "5-point":
            # the value depends on the position within the row only (interior rows are equally long): constant coefficients
            pool = np.array([-1.5, 2.0, 0.25])
            pos = np.arange(len(v)) - np.repeat(p[:-1], np.diff(p))
            v = pool[pos % 3]
        x = synth.x_vector(cols, seed=5)
        y0 = synth.x_vector(rows, seed=6)
        tp, tc, tx = (torch.from_numpy(t).to(dev) for t in (p, c, x))
        tv = torch.from_numpy(v).to(dev)

        def multiply(plan, values):
            ty = torch.from_numpy(y0.copy()).to(dev)
            for _ in range(2):
                plan.spmv(tp.data_ptr(), tc.data_ptr(), values.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            torch.cuda.synchronize()
            return ty.cpu().numpy()

        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        info = plan.info()
        assert info["indexed_values"] == len(np.unique(v)), (name, info)
        assert 0.8 * info["shifted_tiles"] < info["value_row_tiles"] <= info["shifted_tiles"], (name, info)
        if name != "5-point":  # one long run of interior rows: the dictionary launch re-cuts it into tiles of 128 rows
            assert 0 < info["dictionary_launch_tiles"] < info["row_blocks"], (name, info)
        got = multiply(plan, tv)
        assert_bitexact(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), name)
        # one coefficient in the middle of the matrix takes another value of the dictionary: its tile reads its index bytes again
        k = int(p[rows // 2 + (150 if name == "5-point" else 7)]) + 1  # (5-point: the middle of a grid line, away from its 4-entry rows)
        other = np.unique(v)[0] if v[k] != np.unique(v)[0] else (np.unique(v)[-1] if len(np.unique(v)) > 1 else 9.0)
        v2 = v.copy()
        v2[k] = other
        tv2 = torch.from_numpy(v2).to(dev)
        plan.refresh_values(tp.data_ptr(), tc.data_ptr(), tv2.data_ptr(), stream)
        info2 = plan.info()
        assert info2["value_row_tiles"] == info["value_row_tiles"] - 1, (name, info, info2)
        assert info2["streamed_bytes"] > info["streamed_bytes"]
        assert_bitexact(multiply(plan, tv2), oracle.csr_spmv(rows, p, c, v2, x, y=y0, num_threads=4, runs=2), name + ", one coefficient changed")
        # ... and back
        plan.refresh_values(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        assert plan.info()["value_row_tiles"] == info["value_row_tiles"]
        assert np.array_equal(multiply(plan, tv), got)
        plan.close()
        # the marks mean nothing to a plan without a dictionary
        plain = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, capi.FLAG_NO_VALUE_INDEX)
        plain.compress(tc.data_ptr(), stream)
        plain.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plain.index_values(tv.data_ptr(), stream)
        assert plain.info()["value_row_tiles"] == 0 and plain.info()["indexed_values"] == 0
        assert np.array_equal(multiply(plain, tv), got), name
        plain.close()


def test_constant_rows_of_long_stencils_in_recut_tiles(oracle):
    """Constant-coefficient stencils with MORE than 16 entries per row (27-point, 27 plain diagonals): the dictionary launch
    re-cuts their runs of constant-row tiles into tiles of 64 ... 128 rows that read neither values nor LDS, a lane adding
    two adjacent rows left to right -- the reference's order, so bit for bit its result, although the plan would otherwise
    stage x through LDS and sum such rows with two lanes.  Runs too short for 64-row tiles stay with the plan's tiles; the same
    structure with values that differ from row to row takes the indexed path."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def run(rows, cols, p, c, v, x, y0, flags=0):
        tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        ty = torch.from_numpy(y0.copy()).to(dev)
        for _ in range(2):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        info = plan.info()
        plan.close()
        return ty.cpu().numpy(), info

    n = 24
    stencil = [dz * n * n + dy * n + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    for name, N, offs in (("27-point 24^3", n ** 3, stencil), ("27 diagonals", 30000, list(range(-13, 14))),
                          ("27 diagonals, 300 rows", 300, list(range(-13, 14))), ("27 diagonals, 150 rows", 150, list(range(-13, 14))),
                          ("33 diagonals, odd run", 64 * 3 + 33 + 32, list(range(-16, 17)))):
        rows, cols, p, c, v = synth.banded(N, offs, seed=9)
        r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(p))
        v = np.where(c == r, float(len(offs) - 1), -1.0)
        x = synth.x_vector(cols, seed=10)
        y0 = synth.x_vector(rows, seed=11)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2)
        got, info = run(rows, cols, p, c, v, x, y0)
        assert_close(got, want, 2 * abs_products(rows, p, c, v, x) + np.abs(y0), what=name)
        assert info["indexed_values"] == 2 and info["value_row_tiles"] > 0, (name, info)
        assert 0 < info["dictionary_launch_tiles"] < info["row_blocks"], (name, info)
        # interior rows are summed by one lane in the reference's order: bit-identical there (the few boundary rows keep the
        # general path, which gives a 27-entry row two lanes)
        reach = max(abs(o) for o in offs) + 64  # (rows within reach of the ends are shorter; the tile around each end of the run is mixed)
        inner = slice(reach, rows - reach)
        assert inner.stop - inner.start > 20 or rows < 300, name
        assert np.array_equal(got[inner].view(np.uint64), want[inner].view(np.uint64)), name
        # the same structure, values drawn per entry from a dictionary of three: no constant rows, the indexed path (one byte per entry)
        rng = np.random.default_rng(12)
        v3 = np.array([-1.0, 2.5, 0.125])[rng.integers(0, 3, size=len(v))]
        got3, info3 = run(rows, cols, p, c, v3, x, y0)
        assert info3["indexed_values"] == 3 and info3["dictionary_launch_tiles"] == 0, (name, info3)
        assert_close(got3, oracle.csr_spmv(rows, p, c, v3, x, y=y0, num_threads=4, runs=2), 2 * abs_products(rows, p, c, v3, x) + np.abs(y0), what=name + ", three values")
    # a run of 60 interior rows of 27 cannot give a 64-row tile: left alone
    rows, cols, p, c, v = synth.banded(86, list(range(-13, 14)), seed=9)
    v = np.where(c == np.repeat(np.arange(rows), np.diff(p)), 26.0, -1.0)
    x, y0 = synth.x_vector(cols, seed=10), synth.x_vector(rows, seed=11)
    got, info = run(rows, cols, p, c, v, x, y0)
    assert info["dictionary_launch_tiles"] == 0, info
    assert_close(got, oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), 2 * abs_products(rows, p, c, v, x) + np.abs(y0), what="short run")


def test_constant_row_tiles_with_a_coefficient_jump_on_a_tile_boundary(oracle):
    """A piecewise-constant stencil: rows < b carry one coefficient set, rows >= b another.  Wherever b falls on a plan tile
    boundary both neighbours are constant-row tiles (each against its OWN first row) with different sets, and the
    dictionary launch must not merge them into one re-cut tile, which would multiply the second tile's rows with the first
    tile's coefficients (round 3 did: ADVICE r03).  Every b in a window of 128 consecutive rows -- 9-entry rows give plan
    tiles of 56 rows, so the window holds at least two tile boundaries -- bit for bit against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows, cols, p, c, v = synth.banded(12000, [-900, -301, -300, -1, 0, 1, 300, 301, 900], seed=3)
    pos = np.arange(len(v)) - np.repeat(p[:-1], np.diff(p))
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(p))
    pool1, pool2 = np.array([-1.5, 2.0, 0.25]), np.array([2.0, -1.5, 0.75])  # (shared values: the sets differ by position)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    tp, tc, tx = (torch.from_numpy(t).to(dev) for t in (p, c, x))
    merged_somewhere = 0
    marked = []
    for b in range(6000, 6128):
        vb = np.where(r < b, pool1[pos % 3], pool2[pos % 3])
        tv = torch.from_numpy(vb).to(dev)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
        plan.compress(tc.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        info = plan.info()
        assert info["indexed_values"] == 4, info
        merged_somewhere += info["dictionary_launch_tiles"] > 0
        marked.append(info["value_row_tiles"])
        ty = torch.from_numpy(y0.copy()).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        plan.close()
        assert_bitexact(ty.cpu().numpy(), oracle.csr_spmv(rows, p, c, vb, x, y=y0, num_threads=4), "jump at row %d" % b)
    assert merged_somewhere == 128, "the runs on either side of the jump are still re-cut into 128-row tiles"
    # the jump is on a tile boundary exactly when no tile lost its mark to it (one marked tile more than otherwise)
    assert max(marked) == min(marked) + 1 and 2 <= marked.count(max(marked)) <= 4, marked


def test_value_dictionary_out_of_place(oracle):
    """y_out = y_in + A x (what the partitioned multiply uses) through the lane-per-row path with a two-value dictionary
    held in registers and with a seven-value one in the LDS table: y_out bit for bit, y_in untouched."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows, cols, p, c, v2 = synth.poisson2d(300)
    rng = np.random.default_rng(9)
    v7 = np.array([-1.0, 4.0, 0.5, -0.25, 2.0, 1e-3, -7.0])[rng.integers(0, 7, size=len(v2))]
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    for v in (v2, v7):
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=2)
        tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
        plan.compress(tc.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        assert plan.info()["indexed_values"] == len(np.unique(v))
        yin = torch.from_numpy(y0.copy()).to(dev)
        yout = torch.full((rows,), np.nan, dtype=torch.float64, device=dev)
        plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), yin.data_ptr(), yout.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_bitexact(yout.cpu().numpy(), want, "y_out, %d values" % len(np.unique(v)))
        assert_bitexact(yin.cpu().numpy(), y0, "y_in untouched")
        plan.close()
