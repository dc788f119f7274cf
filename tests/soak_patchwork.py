#!/usr/bin/env python3
"""Soak of the tile classifier and every CSR kernel path on adversarial matrices (tests/test_gpu_parity.py's
patchwork_matrix: shifted blocks of any row length, perturbed ones, equally long unrelated rows, ragged / empty / very
long rows), many seeds, with the values as they are and drawn from dictionaries of 1, 2, 5 and 100 values, under the
plan flags that select different paths.  Every result is compared with the oracle (bit for bit under EXACT_ORDER,
1e-10 relative otherwise).  Kept under tests/ because it uses the checker library (oracle/); not collected by pytest.
    python3 tests/soak_patchwork.py [first_seed] [count]"""
import os
import sys

import numpy as np

EXPERIMENTS = os.environ.get("SPMV_HIP_EXPERIMENTS", "") not in ("", "0")  # the row-group flag exists in libspmv_hip_experiments.so only

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    from spmv_amd import capi, synth
    from helpers import assert_bitexact, assert_close, abs_products
    from test_gpu_parity import patchwork_matrix
    import oracle_py
    oracle = oracle_py.Oracle()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    checked = 0
    for seed in range(first, first + count):
        rows, cols, p, c, v0 = patchwork_matrix(seed)
        rng = np.random.default_rng(seed)
        x = synth.x_vector(cols, seed=seed + 1)
        y0 = synth.x_vector(rows, seed=seed + 2)
        for pool in (0, 1, 2, 5, 100):
            v = v0 if pool == 0 else rng.uniform(-1, 1, size=pool)[rng.integers(0, pool, size=len(v0))]
            want = y0 + oracle.csr_spmv(rows, p, c, v, x, num_threads=1)
            scale = abs_products(rows, p, c, v, x) + np.abs(y0)
            tp, tc, tv, tx = (torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (p, c, v, x))
            for f in (0, capi.FLAG_ROWS128, capi.FLAG_EXACT_ORDER, capi.FLAG_ROWS64 | capi.FLAG_NO_X_WINDOW, capi.FLAG_XCD_REMAP) + ((capi.FLAG_ROW_GROUPS,) if EXPERIMENTS else ()):
                plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, f)
                plan.compress(tc.data_ptr(), stream)
                plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
                if pool:
                    plan.index_values(tv.data_ptr(), stream)
                ty = torch.from_numpy(y0.copy()).to(dev)
                plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
                torch.cuda.synchronize()
                got = ty.cpu().numpy()
                what = "seed %d pool %d flags %x %r" % (seed, pool, f, plan.info())
                plan.close()
                if f & capi.FLAG_EXACT_ORDER:
                    assert_bitexact(got, want, what)
                else:
                    assert_close(got, want, scale, what=what)
                checked += 1
        if seed % 5 == 0:
            # the context API on the same matrix: shuffled COO triplets, ELLPACK where it fits, three row blocks of equal
            # entries rehearsed on this one device (peer-store gather)
            want2 = oracle.csr_spmv(rows, p, c, v0, x, y=y0.copy(), num_threads=1, runs=2)
            scale2 = 2 * abs_products(rows, p, c, v0, x) + np.abs(y0)
            i, j, a = synth.csr_to_coordinate(rows, p, c, v0)
            perm = rng.permutation(len(a))
            with capi.Context() as ctx:
                ctx.upload_coo(rows, cols, i[perm] - 1, j[perm] - 1, a[perm])
                ctx.set_x(x); ctx.set_y(y0); ctx.run(2)
                assert_close(ctx.get_y(), want2, scale2, what="seed %d COO" % seed)
            lens = np.diff(p)
            L = int(lens.max())
            if rows * L < 40_000_000 and lens[0] > 0:
                ec = np.zeros((rows, L), dtype=np.int32); ev = np.zeros((rows, L), dtype=np.float64)
                slot = np.arange(len(c)) - np.repeat(p[:-1], lens); rix = np.repeat(np.arange(rows), lens)
                ec[rix, slot] = c; ev[rix, slot] = v0
                last = np.where(lens > 0, c[np.maximum(p[1:] - 1, 0)], 0)
                pad = np.arange(L)[None, :] >= lens[:, None]
                ec[pad] = np.broadcast_to(last[:, None], (rows, L))[pad]
                with capi.Context() as ctx:
                    ctx.upload_ell(rows, cols, L, ec.ravel(), ev.ravel())
                    ctx.set_x(x); ctx.set_y(y0); ctx.run(2)
                    # rows of <= 16 entries: one lane per row, the reference's bits; longer rows: several lanes, 1e-10
                    if L <= 16:
                        assert_bitexact(ctx.get_y(), want2, "seed %d ELLPACK L=%d" % (seed, L))
                    else:
                        assert_close(ctx.get_y(), want2, scale2, what="seed %d ELLPACK L=%d" % (seed, L))
                with capi.Context(0, flags=capi.FLAG_EXACT_ORDER) as ctx:
                    ctx.upload_ell(rows, cols, L, ec.ravel(), ev.ravel())
                    ctx.set_x(x); ctx.set_y(y0); ctx.run(2)
                    assert_bitexact(ctx.get_y(), want2, "seed %d ELLPACK L=%d, exact order" % (seed, L))
            if seed % 4 == 0:
                # segment windows: a KKT-like matrix with jittered stencils (columns in clusters > 65536 apart), grid size
                # and jitter varied by the seed; with and without the windows the bits must be the same
                from spmv_amd import hostapi
                K = hostapi.load("synthetic:kkt:%d,%d" % (41 + seed % 9, (25, 50, 75, 100)[(seed // 4) % 4]), "csr")
                kr, kc, kp, kj, kv = K.rows, K.cols, np.array(K.row_ptr), np.array(K.column_index), np.array(K.value)
                K.close()
                kx = synth.x_vector(kc, seed=seed + 3)
                kwant = oracle.csr_spmv(kr, kp, kj, kv, kx, num_threads=4)
                ktp, ktc, ktv, ktx = (torch.from_numpy(t).to(dev) for t in (kp, kj, kv, kx))
                got = {}
                for f in (0, capi.FLAG_NO_SEGMENT_WINDOW | capi.FLAG_NO_COLUMN_PANELS):
                    plan = capi.CsrPlan(kr, kc, kp, capi.CSR_AUTO, 0, f)
                    plan.compress(ktc.data_ptr(), stream)
                    plan.repack(ktp.data_ptr(), ktc.data_ptr(), ktv.data_ptr(), stream)
                    info = plan.info()
                    assert (info["segwin_tiles"] > 0.5 * info["row_blocks"]) == (f == 0), info
                    ty = torch.zeros(kr, dtype=torch.float64, device=dev)
                    plan.spmv(ktp.data_ptr(), ktc.data_ptr(), ktv.data_ptr(), ktx.data_ptr(), ty.data_ptr(), stream)
                    torch.cuda.synchronize()
                    got[f] = ty.cpu().numpy()
                    plan.close()
                    assert_close(got[f], kwant, abs_products(kr, kp, kj, kv, kx), what="seed %d kkt flags %x" % (seed, f))
                    checked += 1
                a, b = got.values()
                assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), "seed %d: segment windows changed bits" % seed
                del ktp, ktc, ktv, ktx
            os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
            with capi.Context(num_gpus=3, flags=capi.FLAG_PEER_GATHER | capi.FLAG_BALANCE_ENTRIES) as ctx:
                ctx.upload_csr(rows, cols, p, c, v0)
                ctx.set_x(x); ctx.set_y(y0); ctx.run(2)
                assert_close(ctx.get_y(), want2, scale2, what="seed %d three blocks" % seed)
            os.environ.pop("SPMV_HIP_SHARE_DEVICES", None)
            checked += 3
        if (seed - first) % 10 == 9:
            print("seeds %d..%d ok (%d multiplies checked)" % (first, seed, checked), flush=True)
    print("soak ok: %d multiplies" % checked)


if __name__ == "__main__":
    main()
