#!/usr/bin/env python3
"""Regenerate tests/golden/ (run in the build container, where /root/reference exists).

Two kinds of fixture are produced; both are DATA (inputs + expected outputs):

1. Data the reference's own tests hold, extracted verbatim:
   * poisson2D.mtx / poisson2D_b.txt / poisson2D_result.txt
         <- test/poisson2D.hpp:7-2437 (matrix), :2439-2806 (b), :2808-3175 (A*b)
   * test_mtx.gz / test_mtx.tar / test_mtx.tar.gz
         <- the byte arrays of test/test_matrix-market.cpp:109-118, :202-1056, :1078-1094
   * kat.json: the hand-written known-answer cases of test/test_{csr,coo,ell}-matrix.cpp
     and test/test_matrix-market.cpp (matrix text in, expected arrays / vectors out)

2. Vectors produced by running the reference library itself
   (oracle/_ref/libref_spmv.so, built by oracle/Makefile from /root/reference):
   * ref_vectors.json: for each case the Matrix Market text (or synthetic arrays),
     the converted CSR/COO/ELL arrays and y after `runs` accumulating SpMVs with
     `threads` threads, doubles as C99 hex-float strings (bit exact).

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
REF = "/root/reference"

import oracle_py  # noqa: E402
from spmv_amd import synth  # noqa: E402


def hexf(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def ints(a):
    return [int(v) for v in np.asarray(a).ravel()]


# --------------------------------------------------------------------------
# 1. data held by the reference's tests
# --------------------------------------------------------------------------
def extract_poisson2d():
    src = open(os.path.join(REF, "test", "poisson2D.hpp")).read()
    m = re.search(r'poisson2D\{R"\((.*?)\)"\}', src, re.S)
    open(os.path.join(HERE, "poisson2D.mtx"), "w").write(m.group(1) + "\n")
    for name in ("poisson2D_b", "poisson2D_result"):
        m = re.search(name + r" = std::vector<double>\{\s*\{(.*?)\}\};", src, re.S)
        vals = [t.strip() for t in m.group(1).split(",") if t.strip()]
        open(os.path.join(HERE, name + ".txt"), "w").write("\n".join(vals) + "\n")
        print(name, len(vals))


def extract_byte_arrays():
    src = open(os.path.join(REF, "test", "test_matrix-market.cpp")).read()
    for cname, fname in (("test_mtx_gz", "test_mtx.gz"), ("test_mtx_tar", "test_mtx.tar"),
                         ("test_mtx_tar_gz", "test_mtx.tar.gz")):
        m = re.search(r"unsigned char " + cname + r" \[\] = \{(.*?)\};", src, re.S)
        data = bytes(int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]{2}", m.group(1)))
        open(os.path.join(HERE, fname), "wb").write(data)
        print(fname, len(data))


MM_4x5_7 = ("%%MatrixMarket matrix coordinate real general\n% Test matrix\n4 5 7\n"
            "1 1 1.0\n1 2 2.0\n2 2 1.0\n3 3 3.0\n4 1 -1.0\n4 4 2.0\n4 5 1.0\n")
MM_4x5_6 = ("%%MatrixMarket matrix coordinate real general\n% Test matrix\n4 5 6\n"
            "1 1 1.0\n1 2 2.0\n2 2 1.0\n3 3 3.0\n4 4 2.0\n4 5 1.0\n")
MM_4x5_8 = ("%%MatrixMarket matrix coordinate real general\n% Test matrix\n4 5 8\n"
            "1 1 1.0\n1 2 2.0\n1 4 3.0\n2 1 4.0\n2 2 1.0\n3 3 3.0\n4 4 2.0\n4 5 1.0\n")
MM_4x5_8_UNSORTED = ("%%MatrixMarket matrix coordinate real general\n% Test matrix\n4 5 8\n"
                     "1 1 1.0\n1 4 3.0\n2 1 4.0\n2 2 1.0\n1 2 2.0\n3 3 3.0\n4 4 2.0\n4 5 1.0\n")


def known_answer_tests():
    """Transcribed from the reference's gtest files (inputs and expected values only)."""
    x = [5.0, 2.0, 3.0, 3.0, 1.0]
    kat = {
        "_source": "reference test/test_{csr,coo,ell}-matrix.cpp, test/test_matrix-market.cpp",
        "csr_from_matrix_market": {  # test_csr-matrix.cpp:22-29,58-75
            "mtx": MM_4x5_7, "rows": 4, "columns": 5, "num_entries": 7,
            "row_ptr": [0, 2, 3, 4, 7], "column_index": [0, 1, 1, 2, 0, 3, 4],
            "value": [1.0, 2.0, 1.0, 3.0, -1.0, 2.0, 1.0]},
        "csr_from_matrix_market_row_aligned": {  # test_csr-matrix.cpp:31-40,77-94
            "mtx": MM_4x5_7, "row_alignment": 2, "row_ptr": [0, 2, 4, 6, 10],
            "column_index": [0, 1, 1, 0, 2, 0, 0, 3, 4, 0],
            "value": [1.0, 2.0, 1.0, 0.0, 3.0, 0.0, -1.0, 2.0, 1.0, 0.0]},
        "csr_spmv": {"mtx": MM_4x5_7, "x": x, "y": [9.0, 2.0, 9.0, 2.0]},  # :96-103
        "coo_from_matrix_market": {  # test_coo-matrix.cpp:22-29,47-63
            "mtx": MM_4x5_6, "row_index": [0, 0, 1, 2, 3, 3],
            "column_index": [0, 1, 1, 2, 3, 4], "value": [1.0, 2.0, 1.0, 3.0, 2.0, 1.0]},
        "coo_spmv": {"mtx": MM_4x5_6, "x": x, "y": [9.0, 2.0, 9.0, 7.0]},  # :65-72
        "coo_spmv_column_major": {"mtx": MM_4x5_8, "x": x, "y": [18.0, 22.0, 9.0, 7.0]},  # :77-107
        "ell_from_matrix_market": {  # test_ell-matrix.cpp:18-33,53-71
            "mtx": MM_4x5_8, "row_length": 3, "num_entries": 8,
            "column_index": [0, 1, 3, 0, 1, 1, 2, 2, 2, 3, 4, 4],
            "value": [1.0, 2.0, 3.0, 4.0, 1.0, 0.0, 3.0, 0.0, 0.0, 2.0, 1.0, 0.0]},
        "ell_spmv": {"mtx": MM_4x5_8, "x": x, "y": [18.0, 22.0, 9.0, 7.0]},  # :73-82
        "mm_real": {"mtx": "%%MatrixMarket matrix coordinate real general\n% Test matrix\n1 1 1\n1 1 .5\n",
                    "field": "real", "i": [1], "j": [1], "a": [0.5]},  # test_matrix-market.cpp:15-36
        "mm_complex": {"mtx": "%%MatrixMarket matrix coordinate complex general\n% Test matrix\n1 1 1\n1 1 .5 -0.5\n",
                       "field": "complex", "i": [1], "j": [1], "a": [0.5]},  # :38-60 (real part kept)
        "mm_integer": {"mtx": "%%MatrixMarket matrix coordinate integer general\n% Test matrix\n1 1 1\n1 1 5\n",
                       "field": "integer", "i": [1], "j": [1], "a": [5.0]},  # :62-83
        "mm_pattern": {"mtx": "%%MatrixMarket matrix coordinate pattern general\n% Test matrix\n1 1 1\n1 1 5\n",
                       "field": "pattern", "i": [1], "j": [1], "a": [1.0]},  # :85-105 (pattern -> 1.0)
        "mm_max_row_length": {"mtx": MM_4x5_8_UNSORTED, "max_row_length": 3},  # :137-158
        "mm_sort_row_major": {"mtx": MM_4x5_8_UNSORTED,  # :160-179
                              "i": [1, 1, 1, 2, 2, 3, 4, 4], "j": [1, 2, 4, 1, 2, 3, 4, 5],
                              "a": [1.0, 2.0, 3.0, 4.0, 1.0, 3.0, 2.0, 1.0]},
        "mm_compressed": {"files": ["test_mtx.gz", "test_mtx.tar", "test_mtx.tar.gz"],
                          "member": "test.mtx", "rows": 1, "columns": 1,
                          "i": [1], "j": [1], "a": [0.5], "comment": "% Test matrix"},  # :107-135,1058-1115
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)


# --------------------------------------------------------------------------
# 2. vectors generated by the reference library
# --------------------------------------------------------------------------
def mtx_text(rows, cols, i, j, a, field="real", symmetry="general", extra=None):
    lines = ["%%%%MatrixMarket matrix coordinate %s %s" % (field, symmetry), "% generated"]
    lines.append("%d %d %d" % (rows, cols, len(i)))
    for k in range(len(i)):
        if field == "pattern":
            lines.append("%d %d" % (i[k], j[k]))
        elif field == "integer":
            lines.append("%d %d %d" % (i[k], j[k], int(a[k])))
        elif field == "complex":
            lines.append("%d %d %.17g %.17g" % (i[k], j[k], a[k], extra[k]))
        else:
            lines.append("%d %d %.17g" % (i[k], j[k], a[k]))
    return "\n".join(lines) + "\n"


def ref_case(R, name, text, x, runs=1, threads=1, ell=True, row_alignment=1):
    h = R.mm_from_string(text)
    info = R.mm_info(h)
    i, j, a = R.mm_entries(h)
    case = {"name": name, "mtx": text, "rows": info["rows"], "columns": info["columns"],
            "num_entries": info["num_entries"], "field": info["field"], "symmetry": info["symmetry"],
            "x": hexf(x), "runs": runs, "threads": threads,
            "entries": {"i": ints(i), "j": ints(j), "a": hexf(a)},
            "max_row_length": R.mm_max_row_length(h)}
    A = R.csr_from_mm(h, row_alignment)
    p, cj, ca = R.csr_arrays(A)
    case["csr"] = {"row_alignment": row_alignment, "row_ptr": ints(p), "column_index": ints(cj),
                   "value": hexf(ca), "size": R.csr_info(A)["size"],
                   "y": hexf(R.csr_spmv(A, x, num_threads=threads, runs=runs))}
    R.csr_free(A)
    A = R.coo_from_mm(h)
    r, c, v = R.coo_arrays(A)
    case["coo"] = {"row_index": ints(r), "column_index": ints(c), "value": hexf(v),
                   "size": R.coo_info(A)["size"],
                   "y": hexf(R.coo_spmv(A, x, num_threads=threads, runs=runs))}
    R.coo_free(A)
    if ell:
        A = R.ell_from_mm(h)
        c, v = R.ell_arrays(A)
        inf = R.ell_info(A)
        case["ell"] = {"row_length": inf["row_length"], "column_index": ints(c), "value": hexf(v),
                       "size": inf["size"],
                       "y": hexf(R.ell_spmv(A, x, num_threads=threads, runs=runs))}
        R.ell_free(A)
    A = R.hybrid_from_mm(h)
    inf = R.hybrid_info(A)
    ej, ea, cr, cc, cv = R.hybrid_arrays(A)
    case["hybrid"] = {"row_length": inf["row_length"], "ell_column_index": ints(ej), "ell_value": hexf(ea),
                      "coo_row_index": ints(cr), "coo_column_index": ints(cc), "coo_value": hexf(cv),
                      "size": inf["size"],
                      "y": hexf(R.hybrid_spmv(A, x, num_threads=threads, runs=runs))}
    R.hybrid_free(A)
    R.mm_free(h)
    return case


def reference_vectors():
    oracle_py.build(ref=True)
    R = oracle_py.RefLib()
    rng = np.random.default_rng(2024)
    cases = []

    poisson = open(os.path.join(HERE, "poisson2D.mtx")).read()
    b = np.array([float(t) for t in open(os.path.join(HERE, "poisson2D_b.txt")).read().split()])
    cases.append(ref_case(R, "poisson2D_b", poisson, b))
    cases.append(ref_case(R, "poisson2D_random_x", poisson, synth.x_vector(367)))
    cases.append(ref_case(R, "poisson2D_ones_3runs", poisson, np.ones(367), runs=3))
    cases.append(ref_case(R, "poisson2D_2threads", poisson, synth.x_vector(367), threads=2))
    # COO workspace recurrence with >= 2 threads: 3 runs give 6*A*x
    cases.append(ref_case(R, "poisson2D_2threads_3runs", poisson, synth.x_vector(367), threads=2, runs=3))
    cases.append(ref_case(R, "poisson2D_row_aligned4", poisson, synth.x_vector(367), row_alignment=4))
    # drop the bulky poisson text from all but the first case
    for c in cases[1:]:
        c["mtx"] = "@poisson2D.mtx"
        c.pop("entries")
        for fmt in ("csr", "coo", "ell", "hybrid"):
            for k in list(c[fmt].keys()):
                if k not in ("y", "row_alignment", "row_length", "size") and not (
                        fmt == "csr" and c["name"] == "poisson2D_row_aligned4"):
                    c[fmt].pop(k)
    first = cases[0]
    first["mtx"] = "@poisson2D.mtx"
    first.pop("entries")

    # symmetric header: entries are NOT mirrored
    i = np.array([1, 2, 2, 3, 4, 4, 5, 5]); j = np.array([1, 1, 2, 3, 2, 4, 1, 5])
    a = rng.uniform(-1, 1, 8)
    cases.append(ref_case(R, "symmetric_not_expanded", mtx_text(5, 5, i, j, a, symmetry="symmetric"),
                          rng.uniform(-1, 1, 5)))
    # empty rows in the middle and at the end (row 1 non-empty so ELL is defined)
    i = np.array([1, 1, 4, 4, 4, 6]); j = np.array([2, 7, 1, 3, 8, 8])
    cases.append(ref_case(R, "empty_rows", mtx_text(8, 8, i, j, rng.uniform(-1, 1, 6)),
                          rng.uniform(-1, 1, 8)))
    # duplicate (i,j) entries with EQUAL values (so std::sort's unspecified order cannot matter)
    i = np.array([2, 1, 2, 2, 3, 1]); j = np.array([2, 1, 2, 3, 3, 1])
    a = np.array([0.5, 0.25, 0.5, -1.5, 2.0, 0.25])
    cases.append(ref_case(R, "duplicates", mtx_text(3, 3, i, j, a), rng.uniform(-1, 1, 3)))
    # unsorted, rectangular, wider than tall
    N, M = 7, 19
    i = rng.integers(1, N + 1, 40); j = rng.integers(1, M + 1, 40)
    key = i.astype(np.int64) * 100 + j
    _, uniq = np.unique(key, return_index=True)
    i, j = i[np.sort(uniq)], j[np.sort(uniq)]
    i[0] = 1
    cases.append(ref_case(R, "rect_unsorted", mtx_text(N, M, i, j, rng.uniform(-1, 1, len(i))),
                          rng.uniform(-1, 1, M)))
    # fields: pattern -> 1.0, integer -> double, complex -> real part
    i = np.array([1, 2, 3, 3]); j = np.array([1, 3, 1, 2])
    cases.append(ref_case(R, "field_pattern", mtx_text(3, 3, i, j, None, field="pattern"),
                          rng.uniform(-1, 1, 3)))
    cases.append(ref_case(R, "field_integer", mtx_text(3, 3, i, j, [3, -2, 7, 1], field="integer"),
                          rng.uniform(-1, 1, 3)))
    cases.append(ref_case(R, "field_complex", mtx_text(3, 3, i, j, [0.5, -1.25, 2.0, 3.5], field="complex",
                                                       extra=[9.0, 8.0, 7.0, 6.0]),
                          rng.uniform(-1, 1, 3)))
    # one long row among short ones (exercises the long-row kernel path)
    N = 40
    ii = [1] * 33 + list(range(2, N + 1))
    jj = list(range(1, 34)) + [((3 * r) % N) + 1 for r in range(2, N + 1)]
    cases.append(ref_case(R, "one_long_row", mtx_text(N, N, ii, jj, rng.uniform(-1, 1, len(ii))),
                          rng.uniform(-1, 1, N), threads=3))
    # small 5-point stencil, multi-run accumulate
    n = 6
    Np, Mp, p, cj, ca = synth.poisson2d(n)
    si, sj, sa = synth.csr_to_coordinate(Np, p, cj, ca)
    cases.append(ref_case(R, "stencil6_5runs", mtx_text(Np, Mp, si, sj, sa), rng.uniform(-1, 1, Mp), runs=5))

    # print_sample through the reference's JSON stream buffer
    samples = {}
    for name, v in (("readme", [14155, 14201, 14252, 14190, 14300, 14260, 15100, 14400, 21658, 16321]),
                    ("single", [1234]), ("pair", [10, 30]), ("five", [5, 3, 9, 1, 7])):
        samples[name] = {"v": v, "json": R.print_sample(v)}

    json.dump({"_generated_by": "tests/golden/make_golden.py via oracle/_ref/libref_spmv.so",
               "cases": cases, "print_sample": samples},
              open(os.path.join(HERE, "ref_vectors.json"), "w"), indent=0)
    print("ref_vectors.json:", len(cases), "cases")


def trace_config_echoes():
    """Reference parse + print of the committed trace-config fixtures."""
    R = oracle_py.RefLib()
    out = {}
    for name in ("trace_config_2threads.json", "trace_config_1thread.json"):
        text, info = R.trace_config_echo(os.path.join(HERE, name))
        out[name] = {"echo": text, "info": info}
    json.dump(out, open(os.path.join(HERE, "trace_config_echo.json"), "w"), indent=1)


if __name__ == "__main__":
    trace_config_echoes()
    extract_poisson2d()
    extract_byte_arrays()
    known_answer_tests()
    reference_vectors()
