"""The generated stand-ins of BASELINE.json's SuiteSparse configurations (host/matrix/synthetic.cpp)
and the host C ABI that hands them -- and files -- to non-C++ callers (include/spmv_host.h).
CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spmv_amd import hostapi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_header_symbols_all_exported():
    text = open(os.path.join(ROOT, "include", "spmv_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(spmv_host_[a-z0-9_]+)\s*\(", text)))
    assert len(syms) == 6, syms
    lib = C.CDLL(hostapi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libspmv_host.so does not export %s" % s
    # the test hooks are NOT part of the product library any more
    assert not hasattr(lib, "host_mm_from_buffer")


def test_poisson2d_matches_the_numpy_generator():
    A = hostapi.load("synthetic:poisson2d:37")
    rows, cols, p, c, v = synth.poisson2d(37)
    assert (A.rows, A.cols) == (rows, cols)
    assert np.array_equal(A.row_ptr, p) and np.array_equal(A.column_index, c) and np.array_equal(A.value, v)


def _check_sorted_unique(A):
    d = np.diff(A.column_index.astype(np.int64))
    inner = np.ones(len(d), dtype=bool)
    starts = A.row_ptr[1:-1]
    inner[starts[(starts > 0) & (starts <= len(d))] - 1] = False
    assert (d[inner] > 0).all()
    assert A.column_index.min() >= 0 and A.column_index.max() < A.cols


@pytest.mark.parametrize("spec,rows", [("synthetic:kkt:7", 2 * 343 + 6 * 49), ("synthetic:queen:6,5,7", 3 * 210),
                                       ("synthetic:queen:6,5,7,3,0,0,2", 2 * 210), ("synthetic:queen:6,5,7,3,0,0,4", 4 * 210)])
def test_kkt_and_queen_are_symmetric_with_ascending_columns(spec, rows):
    import scipy.sparse as sp
    A = hostapi.load(spec)
    assert A.rows == rows == A.cols
    _check_sorted_unique(A)
    M = sp.csr_matrix((A.value, A.column_index, A.row_ptr), shape=(A.rows, A.cols))
    assert abs(M - M.T).max() == 0.0  # structure AND values
    if spec.count(",") == 6:  # d unknowns per node: the d rows of a node have the same columns, in runs of d
        d = int(spec.rsplit(",", 1)[1])
        lens = np.diff(A.row_ptr)
        assert (lens % d == 0).all()
        for node in (0, 17, 209):
            rows_of = [A.column_index[A.row_ptr[node * d + a]:A.row_ptr[node * d + a + 1]] for a in range(d)]
            assert all(np.array_equal(rows_of[0], r) for r in rows_of[1:])
            assert np.array_equal(rows_of[0][:d], rows_of[0][0] + np.arange(d))
        with pytest.raises(Exception):
            hostapi.load("synthetic:queen:6,5,7,3,20,0,2")  # broken blocks are defined for 3 unknowns per node only
    if "kkt" in spec:
        n = 7
        lens = np.diff(A.row_ptr)
        assert (lens[343:343 + 6 * 49] == 2).all()                # control rows
        assert lens[:343].max() == 28 and lens[343 + 294:].max() <= 30  # state rows: 1 + 27; constraint rows: 27 + faces
        assert M[343 + 294:, 343 + 294:].nnz == 0                 # the zero block of [H A'; A 0]


def test_row_ranges_equal_slices_of_the_whole_matrix():
    for spec in ("synthetic:kkt:6", "synthetic:queen:5,4,6", "synthetic:poisson2d:19", "synthetic:webbase:5000,16000,90,70"):
        A = hostapi.load(spec)
        for b, e in ((0, 0), (0, A.rows), (A.rows // 3, 2 * A.rows // 3), (A.rows - 1, A.rows)):
            S = hostapi.load_csr_rows(spec, b, e)
            assert S.rows == e - b and S.rows_total == A.rows and S.cols == A.cols
            k0, k1 = int(A.row_ptr[b]), int(A.row_ptr[e])
            assert np.array_equal(S.row_ptr, A.row_ptr[b:e + 1] - k0)
            assert np.array_equal(S.column_index, A.column_index[k0:k1]) and np.array_equal(S.value, A.value[k0:k1])


def test_webbase_like_has_the_published_shape():
    """N = 1 000 005, Z = 3 105 536, every row at least one entry, longest row 4700 (webbase-1M's
    published figures); the scattered variant has the same row lengths."""
    A = hostapi.load("synthetic:webbase")
    B = hostapi.load("synthetic:powerlaw")
    for M in (A, B):
        lens = np.diff(M.row_ptr)
        assert M.rows == M.cols == 1000005 and M.stored == 3105536
        assert lens.min() == 1 and lens.max() == 4700 and (lens == 4700).sum() == 1
        assert 0.6 < (lens <= 3).mean() < 0.95          # mostly short rows ...
        assert (lens > 64).sum() > 1000                  # ... with a heavy tail
        _check_sorted_unique(M)
    assert np.array_equal(np.diff(A.row_ptr), np.diff(B.row_ptr))
    r = np.repeat(np.arange(A.rows), np.diff(A.row_ptr))
    near_a = (np.abs(A.column_index.astype(np.int64) - r) < 50000).mean()
    near_b = (np.abs(B.column_index.astype(np.int64) - r) < 50000).mean()
    assert near_a > 0.6 and near_b < 0.2  # host-local links vs uniformly scattered columns


def test_other_formats_and_errors_through_the_host_abi(tmp_path):
    M = hostapi.load("synthetic:webbase:3000,9500,80,75", "hybrid")
    assert M.format == "hybrid" and M.stored == M.rows * M.row_length and M.num_coo_entries > 0
    assert M.num_entries == 9500
    C_ = hostapi.load("synthetic:webbase:3000,9500,80,75", "coo")
    assert C_.stored == 9500 and len(C_.row_index) == 9500
    with pytest.raises(hostapi.HostError) as e:
        hostapi.load("synthetic:nosuchfamily")
    assert "unknown synthetic matrix family" in str(e.value)
    with pytest.raises(hostapi.HostError):
        hostapi.load(str(tmp_path / "missing.mtx"))
    with pytest.raises(hostapi.HostError) as e:
        hostapi.load("synthetic:webbase", "ell")  # rows * 4700 > 2^31 - 1, like the reference's converter
    assert "Integer overflow" in str(e.value)
    # a file goes through the same entry point, symmetric ones unexpanded unless asked
    path = tmp_path / "s.mtx"
    path.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2.0\n2 1 -1.0\n3 2 -1.0\n3 3 2.0\n")
    A = hostapi.load(str(path))
    assert A.stored == 4 and not A.expanded
    E = hostapi.load(str(path), expand_symmetric=True)
    assert E.stored == 6 and E.expanded
    # and the reordering suffix works on generated matrices too
    R = hostapi.load("synthetic:poisson2d:12__RCM")
    assert R.rows == 144 and R.stored == hostapi.load("synthetic:poisson2d:12").stored


@pytest.mark.parametrize("spec,suffix", [("synthetic:kkt:12", ".mtx"), ("synthetic:webbase:20000,62000,300,75", ".mtx.gz"),
                                         ("synthetic:queen:6,5,7", ".mtx"), ("synthetic:queen:6,5,7:tril", ".mtx")])
def test_write_mtx_reads_back_bit_for_bit(tmp_path, spec, suffix):
    """--write-mtx: a generated matrix written as a Matrix Market file (plain or gzip) and read back through the
    loader gives the same CSR arrays, the values bit for bit (shortest round-trip decimals)."""
    import json
    import hostlib
    path = str(tmp_path / ("m" + suffix))
    code, out, err = hostlib.run_cli("--matrix", spec, "--write-mtx", path)
    assert code == 0, err
    info = json.loads(out)
    a, b = hostapi.load(spec, "csr"), hostapi.load(path, "csr")
    assert info["rows"] == a.rows == b.rows and a.cols == b.cols and info["entries"] == len(np.asarray(a.value))
    assert np.array_equal(np.asarray(a.row_ptr), np.asarray(b.row_ptr))
    assert np.array_equal(np.asarray(a.column_index), np.asarray(b.column_index))
    assert np.array_equal(np.asarray(a.value).view(np.uint64), np.asarray(b.value).view(np.uint64))
    if spec.endswith(":tril"):
        # a stored triangle is written as the `symmetric` file it stands for, and mirrored on request like any such file
        assert open(path).readline().split()[-1] == "symmetric"
        full, mirrored = hostapi.load(spec[:-5], "csr"), hostapi.load(path, "csr", expand_symmetric=True)
        assert np.array_equal(np.asarray(full.row_ptr), np.asarray(mirrored.row_ptr))
        assert np.array_equal(np.asarray(full.column_index), np.asarray(mirrored.column_index))
    # needs a matrix; refuses an unwritable path with one line on stderr
    code, out, err = hostlib.run_cli("--write-mtx", path)
    assert code != 0
    code, out, err = hostlib.run_cli("--matrix", spec, "--write-mtx", str(tmp_path / "no" / "such" / "dir.mtx"))
    assert code != 0 and "cannot open" in err


@pytest.mark.parametrize("fmt", ["csr", "coo", "ell", "hybrid"])
def test_matrix_arrays_are_page_aligned(fmt):
    """The reference keeps every matrix array in 4096-byte aligned storage (src/util/aligned-allocator.hpp; its tests
    ask for 64: test/test_ell-matrix.cpp:97-102, test/test_hybrid-matrix.cpp:123-130).  The device upload relies on
    at least 16 (16-byte vector loads of the tiles)."""
    m = hostapi.load("synthetic:webbase:5000,15000,200,75" if fmt != "ell" else "synthetic:poisson2d:40", fmt)
    names = [n for n in ("row_ptr", "row_index", "column_index", "value", "coo_row_index", "coo_column_index", "coo_value") if hasattr(m, n)]
    assert len(names) >= 2
    for n in names:
        a = getattr(m, n)
        if len(a):
            assert a.ctypes.data % 4096 == 0, (fmt, n)
    m.close()


def test_banded_and_random_families():
    """SURVEY 8d's S-banded(N, b, seed) and S-random(N, k, seed): shapes, ascending distinct columns, values in (-1, 1),
    the same arrays whatever the row range, different seeds different matrices."""
    b = hostapi.load("synthetic:banded:5000,13", "csr")
    p, c = np.asarray(b.row_ptr), np.asarray(b.column_index)
    lens = np.diff(p)
    assert b.rows == b.cols == 5000 and lens.max() == 27 and lens[0] == 14 and lens[-1] == 14 and lens[2500] == 27
    assert np.array_equal(c[p[2500]:p[2501]], np.arange(2500 - 13, 2500 + 14))
    r = hostapi.load("synthetic:random:4000,24,3", "csr")
    p, c, v = np.asarray(r.row_ptr), np.asarray(r.column_index), np.asarray(r.value)
    assert np.all(np.diff(p) == 24) and c.min() >= 0 and c.max() < 4000 and np.all(np.abs(v) < 1)
    cc = c.reshape(4000, 24)
    assert np.all(np.diff(cc, axis=1) > 0)  # distinct, ascending
    assert len(np.unique(cc[:, 0])) > 100   # and not all the same
    part = hostapi.load_csr_rows("synthetic:random:4000,24,3", 1000, 1500)
    assert np.array_equal(np.asarray(part.column_index), c[p[1000]:p[1500]]) and np.array_equal(np.asarray(part.value), v[p[1000]:p[1500]])
    other = hostapi.load("synthetic:random:4000,24,4", "csr")
    assert not np.array_equal(np.asarray(other.column_index), c)
    tiny = hostapi.load("synthetic:random:30,30", "csr")  # k = N: every row holds every column
    assert np.array_equal(np.asarray(tiny.column_index).reshape(30, 30), np.tile(np.arange(30), (30, 1)))
    for bad in ("synthetic:banded:10", "synthetic:random:10,11", "synthetic:banded:10,17000"):
        with pytest.raises(hostapi.HostError):
            hostapi.load(bad, "csr")


@pytest.mark.parametrize("base,twin", [("synthetic:poisson2d:40", "synthetic:poisson2d:40,1"), ("synthetic:kkt:12", "synthetic:kkt:12,50"),
                                       ("synthetic:kkt:12", "synthetic:kkt:12,100"), ("synthetic:queen:10,9,8", "synthetic:queen:10,9,8,6")])
def test_pessimistic_twins_keep_the_shape_and_drop_the_friendly_structure(base, twin):
    """The "worst plausible" companions of the stand-ins (profiles/r03_results.md): same size and row populations,
    columns ascending and distinct in every row -- but hashed coefficients (no value dictionary), jittered stencil links
    (no two rows shifted copies of each other), links moved further (wider clusters)."""
    A, B = hostapi.load(base), hostapi.load(twin)
    assert (A.rows, A.cols) == (B.rows, B.cols)
    _check_sorted_unique(B)
    assert B.column_index.min() >= 0 and B.column_index.max() < B.cols
    la, lb = np.diff(A.row_ptr), np.diff(B.row_ptr)
    if "queen" in base:
        assert abs(int(lb.sum()) - int(la.sum())) < 0.05 * la.sum()  # jittered links may coincide at mesh borders
    else:
        assert np.array_equal(la, lb)  # the same row lengths exactly
    if "poisson2d" in base:
        assert np.array_equal(A.column_index, B.column_index)
        assert len(np.unique(A.value)) == 2 and len(np.unique(B.value)) == len(B.value)
    else:
        assert not np.array_equal(A.column_index, B.column_index)
    if "kkt" in base:  # interior rows are shifted copies in the base, not in the twin
        def shifted_pairs(M):
            n, hits = 0, 0
            for r in range(200, 400):
                a = M.column_index[M.row_ptr[r]:M.row_ptr[r + 1]]
                b = M.column_index[M.row_ptr[r + 1]:M.row_ptr[r + 2]]
                if len(a) == len(b):
                    n += 1
                    hits += int(np.array_equal(a + 1, b))
            return hits / max(1, n)
        assert shifted_pairs(A) > 0.7 and shifted_pairs(B) < 0.2
    A.close()
    B.close()


@pytest.mark.parametrize("spec", ["synthetic:poisson2d:9999999999", "synthetic:queen:99999,99999,99999", "synthetic:kkt:99999999999",
                                  "synthetic:webbase:99999999999", "synthetic:kkt:12,101", "synthetic:queen:10,9,8,2", "synthetic:poisson2d:4,1,1"])
def test_out_of_range_specs_are_refused_before_any_arithmetic_on_them(spec):
    """products of numbers parsed from the command line are formed only after each factor has been bounded."""
    with pytest.raises(hostapi.HostError):
        hostapi.load(spec)


def test_poisson3d_family():
    """synthetic:poisson3d:<n>: the 7-point Laplacian on an n^3 grid, x fastest; rows on the faces lack the neighbours outside."""
    n = 7
    A = hostapi.load("synthetic:poisson3d:%d" % n, "csr")
    p, c, v = np.asarray(A.row_ptr), np.asarray(A.column_index), np.asarray(A.value)
    assert A.rows == A.cols == n ** 3 and p[-1] == 7 * n ** 3 - 6 * n * n
    lens = np.diff(p)
    r = (3 * n + 3) * n + 3  # an interior cell
    assert lens[r] == 7 and np.array_equal(c[p[r]:p[r + 1]], [r - n * n, r - n, r - 1, r, r + 1, r + n, r + n * n])
    assert lens[0] == 4 and np.array_equal(c[p[0]:p[1]], [0, 1, n, n * n])
    assert all(np.all(np.diff(c[p[i]:p[i + 1]]) > 0) for i in range(A.rows))
    # symmetric structure, positive diagonal
    import scipy.sparse as sp
    M = sp.csr_matrix((v, c, p), shape=(A.rows, A.cols))
    assert (abs(M) > 0).astype(int).T.tocsr().nnz == M.nnz and ((abs(M) > 0) != (abs(M.T) > 0)).nnz == 0
    assert np.all(M.diagonal() > 4.0)
    part = hostapi.load("synthetic:poisson3d:%d" % n, "csr", row_begin=100, row_end=200) if "row_begin" in hostapi.load.__code__.co_varnames else None
    if part is not None:
        assert np.array_equal(np.asarray(part.column_index), c[p[100]:p[200]])
        part.close()
    A.close()
    with pytest.raises(hostapi.HostError):
        hostapi.load("synthetic:poisson3d:5000")


@pytest.mark.parametrize("spec", ["synthetic:queen:9,8,7", "synthetic:kkt:7", "synthetic:poisson2d:13", "synthetic:banded:500,6,3",
                                  "synthetic:queen:9,8,7,3,100,13"])
def test_stored_triangle_modifier(spec, tmp_path):
    """`<spec>:tril` (round 6): the stored lower triangle of the generated matrix -- what a `symmetric` Matrix Market file holds
    and the reference multiplies as it stands (src/matrix/matrix-market.cpp:396-414 parses the word, :530-555 keeps the entries as
    read; README.md:106).  Entries with column <= row, in place; as coordinate entries the matrix carries the `symmetric` header
    word, so SPMV_HOST_EXPAND_SYMMETRIC gives the whole (structurally symmetric) matrix back; row ranges are slices of it; a
    written file reads back as the same triangle."""
    F = hostapi.load(spec, "csr")
    T = hostapi.load(spec + ":tril", "csr")
    fp, fc, fv = np.array(F.row_ptr), np.array(F.column_index), np.array(F.value)
    tp, tc, tv = np.array(T.row_ptr), np.array(T.column_index), np.array(T.value)
    r = np.repeat(np.arange(F.rows), np.diff(fp))
    keep = fc <= r
    assert (T.rows, T.cols) == (F.rows, F.cols) and T.num_entries == int(keep.sum())
    assert np.array_equal(tc, fc[keep]) and np.array_equal(tv, fv[keep])
    assert np.array_equal(np.diff(tp), np.bincount(r[keep], minlength=F.rows))
    # a row range of the triangle = the rows of the triangle
    b, e = F.rows // 3, 2 * F.rows // 3 + 1
    S = hostapi.load_csr_rows(spec + ":tril", b, e)
    assert S.rows_total == F.rows and np.array_equal(np.array(S.row_ptr), tp[b:e + 1] - tp[b])
    assert np.array_equal(np.array(S.column_index), tc[tp[b]:tp[e]])
    # mirrored again: the whole matrix, where its structure is symmetric (dropped entries of the broken twin are not)
    E = hostapi.load(spec + ":tril", "csr", expand_symmetric=True)
    assert E.expanded
    if ",100," not in spec:
        assert np.array_equal(np.array(E.row_ptr), fp) and np.array_equal(np.array(E.column_index), fc)
    # COO keeps the triangle too (the converters see a symmetric header and mirror nothing)
    Ccoo = hostapi.load(spec + ":tril", "coo")
    assert Ccoo.num_entries == T.num_entries
    for m in (F, T, S, E, Ccoo):
        m.close()


def test_stored_triangle_of_the_block_matrices_has_the_rows_the_plan_looks_for():
    """queen:tril / kkt:tril at a small size: the rows of a node are 3 k + 1, 3 k + 2, 3 k + 3 entries long (what spmv_hip_plan_csr's
    skewed-triple hint reads off row_ptr), the KKT states and controls hold their diagonal only."""
    T = hostapi.load("synthetic:queen:10,9,8:tril", "csr")
    l = np.diff(np.array(T.row_ptr)).reshape(-1, 3)
    assert np.all(l[:, 0] % 3 == 1) and np.all(l[:, 1] == l[:, 0] + 1) and np.all(l[:, 2] == l[:, 0] + 2)
    T.close()
    K = hostapi.load("synthetic:kkt:8:tril", "csr")
    l = np.diff(np.array(K.row_ptr))
    ny, nu = 8 ** 3, 6 * 8 ** 2
    assert np.all(l[:ny + nu] == 1) and l[ny + nu:].min() >= 8 and l[ny + nu:].max() <= 27 + 3
    K.close()
