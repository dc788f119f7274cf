"""Shared helpers for the tests: golden-fixture loading, tolerances, small parsers."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json: "y within 1e-10 relative of CPU reference".  Paths that add a row's
# products in the reference's order must be bit-exact instead (tolerance 0).
RTOL = 1e-10


def unhex(a):
    return np.array([float.fromhex(s) for s in a], dtype=np.float64)


def load_golden():
    g = {}
    g["kat"] = json.load(open(os.path.join(GOLDEN, "kat.json")))
    ref = json.load(open(os.path.join(GOLDEN, "ref_vectors.json")))
    g["cases"] = ref["cases"]
    g["print_sample"] = ref["print_sample"]
    g["poisson2D_mtx"] = open(os.path.join(GOLDEN, "poisson2D.mtx")).read()
    g["poisson2D_b"] = np.array([float(t) for t in open(os.path.join(GOLDEN, "poisson2D_b.txt")).read().split()])
    g["poisson2D_result"] = np.array(
        [float(t) for t in open(os.path.join(GOLDEN, "poisson2D_result.txt")).read().split()])
    return g


def case_mtx(g, case):
    return g["poisson2D_mtx"] if case["mtx"] == "@poisson2D.mtx" else case["mtx"]


def parse_mtx_text(text):
    """Tiny Matrix Market reader for test inputs (coordinate only): the pure-Python
    oracle of the loader's accept set.  Returns rows, cols, i, j, a (1-based; values per
    reference src/matrix/matrix-market.cpp:243-277: complex -> real part, pattern -> 1.0)."""
    lines = text.split("\n")
    hdr = lines[0].split()
    assert hdr[0] == "%%MatrixMarket" and hdr[1].lower() == "matrix" and hdr[2].lower() == "coordinate"
    field = hdr[3].lower()
    k = 1
    while lines[k].startswith("%"):
        k += 1
    rows, cols, n = (int(t) for t in lines[k].split()[:3])
    toks = " ".join(lines[k + 1:]).split()
    per = {"real": 3, "integer": 3, "complex": 4, "pattern": 2}[field]
    i = np.array([int(toks[per * e]) for e in range(n)], dtype=np.int32)
    j = np.array([int(toks[per * e + 1]) for e in range(n)], dtype=np.int32)
    if field == "pattern":
        a = np.ones(n)
    else:
        a = np.array([float(toks[per * e + 2]) for e in range(n)], dtype=np.float64)
    return rows, cols, i, j, a, field, hdr[4].lower()


def abs_products(rows, row_ptr, col, val, x):
    """(|A||x|)_i: the scale a reordered sum's rounding error is proportional to."""
    lens = np.diff(np.asarray(row_ptr, dtype=np.int64))
    r = np.repeat(np.arange(rows, dtype=np.int32 if rows < 2**31 else np.int64), lens)
    return np.bincount(r, weights=np.abs(val) * np.abs(x[col]), minlength=rows)


# Rows that needed the second clause of assert_close (below): appended to $SPMV_TOLERANCE_REPORT (default
# gpurun_out/tolerance_report.jsonl when that directory exists) so that a run says which tests passed on the contract's
# formula alone and which needed the summation bound, and by how much.
_REPORT = os.environ.get("SPMV_TOLERANCE_REPORT") or (
    os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out", "tolerance_report.jsonl")
    if os.path.isdir(os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out")) else None)


def _report(entry):
    if not _REPORT:
        return
    entry["test"] = os.environ.get("PYTEST_CURRENT_TEST", "")
    try:
        with open(_REPORT, "a") as f:
            f.write(json.dumps(entry) + "\n")
    except OSError:
        pass


def row_tolerance(y_cpu, rtol=RTOL):
    """SURVEY section 8(d), verbatim: per row |y_gpu - y_cpu| <= 1e-10 * max(|y_cpu_i|, 1e-6 * ||y_cpu||_inf)."""
    y_cpu = np.asarray(y_cpu)
    ninf = np.max(np.abs(y_cpu)) if y_cpu.size else 0.0
    return rtol * np.maximum(np.abs(y_cpu), 1e-6 * ninf)


def assert_close(y_gpu, y_cpu, scale=None, rtol=RTOL, what="", nterms=4096):
    """The contract's tolerance (BASELINE.json "within 1e-10 relative"):

      * inf-norm and 2-norm of the error <= 1e-10 of the reference's;
      * per row |y_gpu_i - y_cpu_i| <= 1e-10 * max(|y_cpu_i|, 1e-6 * ||y_cpu||_inf).

    The per-row floor is 1e-16 * ||y||_inf -- below one unit in the last place of the largest element -- so a row whose
    products cancel can miss it by rounding alone whenever its sum is formed in another order than the reference's (several
    lanes per row, segmented sums).  For exactly those rows, and only when the caller hands over `scale` = (|A||x|)_i (times
    the runs, plus |y0_i|), a second clause applies: the a-priori bound on the difference of two summation orders of the
    same products, 2 * nterms * 2^-53 * scale_i (nterms >= the row's products; default 4096) -- six orders of magnitude
    tighter than the 1e-10 * (|A||x|)_i that rounds 1-3 allowed.  Whatever needs the second clause is reported (see
    _report) with the largest ratio to the contract's bound."""
    y_gpu = np.asarray(y_gpu)
    y_cpu = np.asarray(y_cpu)
    assert y_gpu.shape == y_cpu.shape, what
    if y_cpu.size == 0:
        return
    err = np.abs(y_gpu - y_cpu)
    ninf = np.max(np.abs(y_cpu))
    assert np.max(err) <= rtol * max(ninf, np.finfo(float).tiny), \
        "%s: inf-norm rel err %.3e" % (what, np.max(err) / max(ninf, 1e-300))
    n2 = np.linalg.norm(y_cpu)
    assert np.linalg.norm(y_gpu - y_cpu) <= rtol * max(n2, np.finfo(float).tiny), what
    contract = row_tolerance(y_cpu, rtol) + 1e-300
    over = np.nonzero(err > contract)[0]
    if over.size == 0:
        return
    if scale is None:
        raise AssertionError("%s: %d rows outside 1e-10*max(|y_i|, 1e-6*||y||inf), first %d: gpu=%r cpu=%r" % (
            what, over.size, over[0], y_gpu[over[0]], y_cpu[over[0]]))
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), y_cpu.shape)
    summation = 2.0 * nterms * 2.0 ** -53 * np.maximum(scale[over], np.abs(y_cpu[over])) + 1e-300
    bad = over[err[over] > summation]
    _report({"what": what, "rows": int(y_cpu.size), "rows_needing_summation_bound": int(over.size),
             "worst_ratio_to_contract": float(np.max(err[over] / contract[over])),
             "worst_ratio_to_summation_bound": float(np.max(err[over] / summation)), "failed": int(bad.size)})
    assert bad.size == 0, "%s: %d rows off (contract AND summation bound), first %d: gpu=%r cpu=%r" % (
        what, bad.size, bad[0], y_gpu[bad[0]], y_cpu[bad[0]])


def assert_ell(y_gpu, y_cpu, L, flags, ell_col, ell_val, x, y0=None, runs=1, what=""):
    """The ELLPACK tolerance classes: rows of <= 16 entries, SPMV_HIP_FLAG_EXACT_ORDER (0x2) and
    SPMV_HIP_FLAG_ELL_COLUMN_MAJOR (0x200) sum every row with one lane in the reference's order
    (src/matrix/ell-matrix.cpp:243-258): bit-exact.  Longer rows are summed by 2..64 lanes by default:
    1e-10 relative, per row against the magnitude of its products (BASELINE.json)."""
    if L <= 16 or (flags & (0x2 | 0x200)):
        return assert_bitexact(y_gpu, y_cpu, what)
    rows = len(np.asarray(y_cpu))
    scale = (np.abs(np.asarray(ell_val).reshape(rows, L)) * np.abs(np.asarray(x)[np.asarray(ell_col).reshape(rows, L)])).sum(axis=1) * runs
    if y0 is not None:
        scale = scale + np.abs(y0)
    assert_close(y_gpu, y_cpu, scale, what=what)


def assert_bitexact(y_gpu, y_cpu, what=""):
    y_gpu = np.ascontiguousarray(y_gpu, dtype=np.float64)
    y_cpu = np.ascontiguousarray(y_cpu, dtype=np.float64)
    assert y_gpu.shape == y_cpu.shape, what
    same = y_gpu.view(np.uint64) == y_cpu.view(np.uint64)
    if not same.all():
        k = int(np.nonzero(~same)[0][0])
        raise AssertionError("%s: %d of %d values differ bitwise, first at %d: %r vs %r" % (
            what, int((~same).sum()), same.size, k, y_gpu[k].hex(), y_cpu[k].hex()))


def have_gpu():
    try:
        from spmv_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False
