#!/usr/bin/env python3
"""Time Matrix Market ingest: this repo's loader + CSR converter against the
reference library's (oracle/_ref, when present) on a generated 5-point stencil file.

    python tests/ingest_bench.py [--grid 1024] [--keep]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1024)
    args = ap.parse_args()
    import hostlib
    import oracle_py
    from spmv_amd import synth

    rows, cols, p, c, v = synth.poisson2d(args.grid)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    perm = np.random.default_rng(0).permutation(len(a))  # unsorted file: the converters must sort
    d = tempfile.mkdtemp()
    path = os.path.join(d, "stencil.mtx")
    t = time.perf_counter()
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (rows, cols, len(a)))
        np.savetxt(f, np.column_stack([i[perm], j[perm], a[perm]]), fmt="%d %d %.17g")
    size = os.path.getsize(path)
    print("file: %d entries, %.1f MB (written in %.1f s)" % (len(a), size / 1e6, time.perf_counter() - t))

    def best_of(n, load, convert, free_a, free_h):
        best = None
        for _ in range(n):
            t0 = time.perf_counter()
            h = load()
            t1 = time.perf_counter()
            A = convert(h)
            t2 = time.perf_counter()
            free_a(A)
            free_h(h)
            if best is None or t2 - t0 < best[2]:
                best = (t1 - t0, t2 - t1, t2 - t0)
        return best

    H = hostlib.Host()
    ld, cv, tot = best_of(3, lambda: H.mm_load(path), lambda h: H.lib.host_csr_from_mm(h, 1),
                          H.lib.host_csr_free, H.mm_free)
    print("this repo : load %.2f s (%.0f MB/s), to CSR %.2f s, total %.2f s  [%d OpenMP threads, best of 3]" % (
        ld, size / 1e6 / ld, cv, tot, os.cpu_count()))
    if oracle_py.RefLib.available():
        R = oracle_py.RefLib()
        ld2, cv2, tot2 = best_of(3, lambda: R.mm_load(path), lambda h: R.csr_from_mm(h), R.csr_free, R.mm_free)
        print("reference : load %.2f s (%.0f MB/s), to CSR %.2f s, total %.2f s  [best of 3]" % (
            ld2, size / 1e6 / ld2, cv2, tot2))
        print("speed-up  : load %.1fx, convert %.1fx, total %.1fx" % (ld2 / ld, cv2 / cv, tot2 / tot))
    os.remove(path)
    os.rmdir(d)


if __name__ == "__main__":
    main()
