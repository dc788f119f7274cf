#!/usr/bin/env python3
"""Regenerate tests/golden/reorder_vectors.json (run in the build container, where /root/reference exists and
oracle/_ref/libref_spmv.so has been built from it by oracle/Makefile).

For each case: the Matrix Market text that goes in (or the name of a fixture file next to this script) and what the REFERENCE
LIBRARY itself returns for `<file>__RCM` -- matrix_market::load_matrix with the reordering suffix
(src/matrix/matrix-market.cpp:782-802 -> find_new_order_RCM, src/matrix/matrix-market-reorder.cpp:60-170 -> Matrix::permute):
the permuted (i, j) of every entry, in file order.  Data only: inputs and expected outputs.

Usage:  python tests/golden/make_reorder_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))

import oracle_py  # noqa: E402
from spmv_amd import synth  # noqa: E402


def mtx_text(rows, cols, i, j, a, symmetry="general"):
    lines = ["%%MatrixMarket matrix coordinate real " + symmetry, "%d %d %d" % (rows, cols, len(a))]
    lines += ["%d %d %s" % (ii, jj, repr(float(v))) for ii, jj, v in zip(i, j, a)]
    return "\n".join(lines) + "\n"


def scrambled_band(n, seed):
    rows, cols, p, c, v = synth.banded(n, [-7, -2, -1, 0, 1, 2, 7], seed=seed)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    perm = np.random.default_rng(seed).permutation(n) + 1
    return perm[i - 1].astype(np.int32), perm[j - 1].astype(np.int32), a


def main():
    R = oracle_py.RefLib()
    cases = []
    inputs = [("poisson2D", "@poisson2D.mtx"), ("bus1138_like", "@bus1138_like.mtx")]
    i, j, a = scrambled_band(700, 5)
    inputs.append(("scrambled_band_700", mtx_text(700, 700, i, j, a)))
    i = np.array([1, 2, 2, 3, 5, 6, 6, 7, 7, 9, 10, 10, 1, 12, 12])
    j = np.array([2, 1, 3, 2, 6, 5, 7, 6, 6, 9, 11, 10, 2, 12, 4])
    inputs.append(("components_isolated_duplicates", mtx_text(12, 12, i, j, np.arange(1.0, 16.0))))
    i, j, a = scrambled_band(300, 9)
    keep = i >= j
    inputs.append(("one_triangle_symmetric_header", mtx_text(300, 300, i[keep], j[keep], a[keep], "symmetric")))
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in inputs:
            body = open(os.path.join(HERE, text[1:])).read() if text.startswith("@") else text
            path = os.path.join(tmp, name + ".mtx")
            open(path, "w").write(body)
            h = R.mm_load(path + "__RCM")
            ri, rj, ra = R.mm_entries(h)
            o = R.mm_load(path)
            oi, oj, oa = R.mm_entries(o)
            assert np.array_equal(ra, oa), "permute keeps the file order of the entries"
            cases.append({"name": name, "mtx": text, "rows": int(R.mm_info(o)["rows"]), "entries": int(len(oa)),
                          "rcm_i": [int(v) for v in ri], "rcm_j": [int(v) for v in rj]})
            R.mm_free(h)
            R.mm_free(o)
            print(name, len(oa), "entries; bandwidth", int(np.max(np.abs(oi - oj))), "->", int(np.max(np.abs(ri - rj))))
        # "__GP<n>" as THIS reference build does it (no METIS: src/matrix/matrix-market-reorder.cpp:172-181 -- a warning on stdout and
        # the identity order; after "__RCM" the RCM order stays), also for a rectangular pattern matrix (no shape / field check there)
        gp_cases = []
        rect = "%%MatrixMarket matrix coordinate pattern general\n3 5 4\n1 5\n3 1\n2 2\n1 1\n"
        for name, text, suffix in [("components_isolated_duplicates", dict(inputs)["components_isolated_duplicates"], "__GP"),
                                   ("components_isolated_duplicates", dict(inputs)["components_isolated_duplicates"], "__GP16"),
                                   ("one_triangle_symmetric_header", dict(inputs)["one_triangle_symmetric_header"], "__GP4"),
                                   ("one_triangle_symmetric_header", dict(inputs)["one_triangle_symmetric_header"], "__RCM__GP8"),
                                   ("one_triangle_symmetric_header", dict(inputs)["one_triangle_symmetric_header"], "__GP8__RCM"),
                                   ("rectangular_pattern", rect, "__GP2")]:
            path = os.path.join(tmp, name + ".mtx")
            open(path, "w").write(text)
            h = R.mm_load(path + suffix)
            gi, gj, _ = R.mm_entries(h)
            gp_cases.append({"name": name, "suffix": suffix, "mtx": text, "i": [int(v) for v in gi], "j": [int(v) for v in gj]})
            R.mm_free(h)
            print(name + suffix, len(gi), "entries")
    json.dump({"_generated_by": "tests/golden/make_reorder_golden.py with oracle/_ref/libref_spmv.so (the reference's own "
                                "load_matrix on <file>__RCM)", "cases": cases, "gp_cases": gp_cases},
              open(os.path.join(HERE, "reorder_vectors.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
