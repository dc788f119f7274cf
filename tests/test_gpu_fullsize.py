"""BASELINE.json's configurations at FULL size on the MI355X, whole vector against the oracle.

configs[1] Poisson 4096^2 exactly as bench.py times it (compressed + repacked plan: 16-bit /
shifted / pattern / 128-row tiles), and the same matrix as ELLPACK (L = 5) and COO through the
context API; configs[2..4] through the structure-faithful generators of
host/matrix/synthetic.cpp (queen-, kkt-, webbase-like; the SuiteSparse files cannot be fetched
here), every row of y compared with the oracle's multi-threaded CSR loop.

Tolerances: bit-exact where every row is summed by one lane in the reference's order
(rows of <= 16 entries in CSR wave tiles, ELLPACK); 1e-10 relative otherwise (BASELINE.json).
"""
import os

import numpy as np
import pytest

from helpers import assert_bitexact, assert_close, abs_products
from spmv_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu

THREADS = 16


def _plan_multiply(A, x, compress=True, flags=0, runs=1, y0=None):
    """y0 + runs * A x through the device-pointer API with the plan bench.py builds."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(A.rows, A.cols, A.row_ptr, capi.CSR_AUTO, 0, flags)
    tp, tc, tv = (torch.from_numpy(np.asarray(t)).to(dev) for t in (A.row_ptr, A.column_index, A.value))
    tx = torch.from_numpy(x).to(dev)
    if compress:
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        plan.index_values(tv.data_ptr(), stream)
    ty = torch.zeros(A.rows, dtype=torch.float64, device=dev) if y0 is None else torch.from_numpy(y0).to(dev)
    for _ in range(runs):
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
    torch.cuda.synchronize()
    info = plan.info()
    y = ty.cpu().numpy()
    plan.close()
    del tp, tc, tv, tx, ty
    torch.cuda.empty_cache()
    return y, info


def test_poisson4096_compressed_plan_whole_vector_bitexact(oracle):
    """The configuration bench.py times: CSR_AUTO + compress + repack.  > 90 % of the tiles must be
    shifted ones (columns not read), and all 16 777 216 rows must equal the oracle bit for bit."""
    A = hostapi.load("synthetic:poisson2d:4096", "csr")
    assert A.rows == 16777216 and A.stored == 83869696
    x = synth.x_vector(A.cols, seed=12345)
    y, info = _plan_multiply(A, x)
    assert info["shifted_tiles"] > 0.9 * info["row_blocks"], info
    assert info["narrow_tiles"] == info["row_blocks"] and info["panel_tiles"] == 0
    assert info["shifted_entries"] > 0.9 * A.stored
    assert info["indexed_values"] == 2  # the stencil's two values, -1 and 4: one byte per entry instead of eight
    # the tile classes stream less than the algorithmic bytes (1 instead of 12 B per entry, no row_ptr)
    assert info["streamed_bytes"] < 0.4 * synth.csr_bytes(A.rows, A.cols, A.stored)
    want = oracle.csr_spmv(A.rows, A.row_ptr, A.column_index, A.value, x, num_threads=THREADS)
    assert_bitexact(y, want, "poisson 4096^2, compressed plan")
    # and twice more on top (y += A x accumulates)
    y3, _ = _plan_multiply(A, x, runs=2, y0=y)
    want3 = oracle.csr_spmv(A.rows, A.row_ptr, A.column_index, A.value, x, y=want, num_threads=THREADS, runs=2)
    assert_bitexact(y3, want3, "poisson 4096^2, three accumulating runs")
    A.close()


@pytest.mark.parametrize("fmt", ["ell", "coo"])
def test_poisson4096_alt_formats_context_api(oracle, fmt):
    """The same matrix as ELLPACK (L = 5, row-major in place as uniform wave tiles: bit-exact with
    ell_spmv_inner_loop) and as COO (row-sorted triplets: 1e-10)."""
    M = hostapi.load("synthetic:poisson2d:4096", fmt)
    x = synth.x_vector(M.cols, seed=12345)
    with capi.Context(0) as ctx:
        if fmt == "ell":
            assert M.row_length == 5
            ctx.upload_ell(M.rows, M.cols, M.row_length, M.column_index, M.value)
        else:
            ctx.upload_coo(M.rows, M.cols, M.row_index, M.column_index, M.value)
        ctx.set_x(x)
        ctx.run()
        y = ctx.get_y()
        info = ctx.info()
    if fmt == "ell":
        want = oracle.ell_spmv(M.rows, M.row_length, M.column_index, M.value, x, num_threads=THREADS)
        assert info["row_blocks"] > 0  # ran as wave tiles
        assert_bitexact(y, want, "poisson 4096^2 ELL")
    else:
        want = oracle.coo_spmv(M.rows, M.row_index, M.column_index, M.value, x)
        assert_close(y, want, scale=np.full(M.rows, 8.0), what="poisson 4096^2 COO")
    M.close()


@pytest.mark.parametrize("spec,min_rows,min_nnz", [
    ("synthetic:queen", 4147110, 320000000),      # configs[2]: Queen_4147-like, ~329.5 M entries, ~79/row
    ("synthetic:queen:110,71,177,3,20,1000", 4140900, 320000000),  # ... 2 % of its blocks with entries dropped, a node with 1 or 2 unknowns every 1000: masked block tiles
    ("synthetic:kkt:200", 16240000, 430000000),   # configs[3]: nlpkkt200-like, N = 16.24 M, ~436 M entries
    ("synthetic:webbase", 1000005, 3105536),      # configs[4]: webbase-1M-like
    ("synthetic:powerlaw", 1000005, 3105536),     # the same row lengths, uniformly scattered columns
    # round 6: what the reference multiplies when it is handed the SuiteSparse FILES -- the stored lower triangle of a `symmetric`
    # Matrix Market file, nothing mirrored (src/matrix/matrix-market.cpp:396-414, :530-555; README.md:106)
    ("synthetic:queen:tril", 4147110, 166000000),   # configs[2] as stored: 166.8 M entries, triangular diagonal blocks, rows of 1 ... 84
    ("synthetic:kkt:200:tril", 16240000, 222000000),  # configs[3] as stored: 8.24 M rows that hold their diagonal only, 8 M rows of <= 27
])
def test_baseline_configs_full_size_whole_vector(oracle, spec, min_rows, min_nnz):
    A = hostapi.load(spec, "csr")
    assert (A.rows == min_rows or (spec.count(",") > 3 and min_rows <= A.rows < min_rows + 6300)) and A.stored >= min_nnz
    x = synth.x_vector(A.cols, seed=12345)
    y, info = _plan_multiply(A, x)
    if spec.count(",") > 3:
        assert info["masked_block_tiles"] > 0.5 * info["row_blocks"], info
    want = oracle.csr_spmv(A.rows, A.row_ptr, A.column_index, A.value, x, num_threads=THREADS)
    lens = np.diff(A.row_ptr)
    if lens.max() <= 16 and info["panel_tiles"] == 0:
        assert_bitexact(y, want, spec)
    else:
        assert_close(y, want, scale=abs_products(A.rows, A.row_ptr, A.column_index, A.value, x), what=spec)
        # rows of <= 16 entries are summed by one lane in the reference's order whenever their tile holds no longer row;
        # at least the tiles of short rows only must be exact: most rows
        short = lens <= 16
        same = (y.view(np.uint64) == want.view(np.uint64))
        # (where short rows are a population of their own -- the kkt-like matrix's control rows, a web graph -- not the three corner
        # rows of a mesh, which share their tiles with long rows and, since round 6, may sit in a block tile)
        if info["panel_tiles"] == 0 and short.mean() > 0.01:
            assert same[short].mean() > 0.5, (spec, same[short].mean())
    A.close()


@pytest.mark.parametrize("fmt", ["coo", "hybrid"])
def test_webbase_alt_formats(oracle, fmt):
    """configs[4]: webbase-1M-like in COO and in the hybrid ELL+COO format; ELLPACK itself overflows
    int32 exactly like the reference's converter (rows * 4700 > 2^31 - 1)."""
    M = hostapi.load("synthetic:webbase", fmt)
    x = synth.x_vector(M.cols, seed=12345)
    A = hostapi.load("synthetic:webbase", "csr")
    want = oracle.csr_spmv(A.rows, A.row_ptr, A.column_index, A.value, x, num_threads=THREADS)
    scale = abs_products(A.rows, A.row_ptr, A.column_index, A.value, x)
    with capi.Context(0) as ctx:
        if fmt == "coo":
            ctx.upload_coo(M.rows, M.cols, M.row_index, M.column_index, M.value)
        else:
            ctx.upload_hybrid(M.rows, M.cols, M.row_length, M.column_index, M.value,
                              M.coo_row_index, M.coo_column_index, M.coo_value)
        ctx.set_x(x)
        ctx.run(2)
        y = ctx.get_y()
    assert_close(y, 2.0 * want, scale=2.0 * scale, what="webbase " + fmt)
    M.close()
    A.close()


def test_webbase_ellpack_overflows_like_the_reference():
    with pytest.raises(hostapi.HostError) as e:
        hostapi.load("synthetic:webbase", "ell")
    assert "Integer overflow" in str(e.value)  # src/matrix/ell-matrix.cpp:199-205


def test_y_in_y_out_and_plan_guards(oracle):
    """spmv_hip_csr_spmv_out (two segment buffers, as the partitioned multiply uses them), the plan's
    content guard, and the refresh of a column-panel value snapshot."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, gen in (("poisson", lambda: synth.poisson2d(300)), ("powerlaw", lambda: synth.powerlaw(40000, 40000, seed=3)),
                      ("band", lambda: synth.banded(30000, range(-20, 21), seed=2))):
        rows, cols, p, c, v = gen()
        x = synth.x_vector(cols, seed=5)
        y0 = synth.x_vector(rows, seed=6)
        want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4)
        scale = abs_products(rows, p, c, v, x) + np.abs(y0)
        tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
        for algo in (capi.CSR_AUTO, capi.CSR_SCALAR, capi.CSR_ADAPTIVE):
            plan = capi.CsrPlan(rows, cols, p, algo)
            plan.compress(tc.data_ptr(), stream)
            ya = torch.from_numpy(y0).to(dev)
            yb = torch.full((rows,), 123.0, dtype=torch.float64, device=dev)
            plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ya.data_ptr(), yb.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(ya.cpu().numpy(), y0), "y_in must not change"
            assert_close(yb.cpu().numpy(), want, scale, what="%s y_out algo %d" % (name, algo))
            # and back: y_a = y_b + A x
            plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), yb.data_ptr(), ya.data_ptr(), stream)
            torch.cuda.synchronize()
            want2 = oracle.csr_spmv(rows, p, c, v, x, y=want, num_threads=4)
            assert_close(ya.cpu().numpy(), want2, 2 * scale, what="%s y_out twice algo %d" % (name, algo))
            with pytest.raises(capi.SpmvHipError) as e:  # overlapping buffers are refused
                plan.spmv_out(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ya.data_ptr(), ya.data_ptr() + 8, stream)
            assert e.value.code == capi.ERR_INVALID
            plan.close()
        # content guard: same address, different columns -> ERR_STATE, not a wrong y
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO)
        plan.compress(tc.data_ptr(), stream)
        plan.verify(tc.data_ptr(), stream)
        saved = tc.clone()
        tc[: min(64, len(c))] = torch.flip(tc[: min(64, len(c))], dims=[0]) if len(c) > 1 else tc
        tc[0] = (tc[0] + 1) % cols
        torch.cuda.synchronize()
        with pytest.raises(capi.SpmvHipError) as e:
            plan.verify(tc.data_ptr(), stream)
        assert e.value.code == capi.ERR_STATE
        plan.close()
        # the first multiply after compress checks by itself (no explicit verify in between) ...
        tc.copy_(saved)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO)
        plan.compress(tc.data_ptr(), stream)
        tc[0] = (tc[0] + 1) % cols
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        with pytest.raises(capi.SpmvHipError) as e:
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        assert e.value.code == capi.ERR_STATE
        plan.close()
        # ... and with SPMV_HIP_FLAG_VERIFY_PLAN every multiply does
        tc.copy_(saved)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, capi.FLAG_VERIFY_PLAN)
        plan.compress(tc.data_ptr(), stream)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        tc[0] = (tc[0] + 1) % cols
        with pytest.raises(capi.SpmvHipError) as e:
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        assert e.value.code == capi.ERR_STATE
        tc.copy_(saved)
        plan.close()
    # unknown flag bits are refused
    with pytest.raises(capi.SpmvHipError) as e:
        capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0x2000)
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.SpmvHipError) as e:
        capi.Context(0, flags=0x40000000)
    assert e.value.code == capi.ERR_INVALID


def test_column_panel_value_snapshot_refresh(oracle):
    """A scattered matrix gets column panels (a copy of the values inside the plan): plan_info says so,
    and after the caller changes the values refresh_values brings the copy up to date."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rows, cols, p, c, v = synth.random_uniform(600000, 600000, 8, seed=3)
    x = synth.x_vector(cols, seed=5)
    tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO)
    plan.compress(tc.data_ptr(), stream)
    plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    info = plan.info()
    assert info["panel_tiles"] > 0 and info["value_snapshot"] == 1, info
    scale = abs_products(rows, p, c, v, x)

    def mul():
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr(), stream)
        torch.cuda.synchronize()
        return ty.cpu().numpy()

    assert_close(mul(), oracle.csr_spmv(rows, p, c, v, x, num_threads=8), scale, what="panels")
    v2 = v * 3.0 + 0.25
    tv.copy_(torch.from_numpy(v2).to(dev))
    plan.refresh_values(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    assert_close(mul(), oracle.csr_spmv(rows, p, c, v2, x, num_threads=8), abs_products(rows, p, c, v2, x), what="panels after refresh")
    plan.close()


def test_context_on_callers_stream(oracle):
    """spmv_hip_set_stream: the context's launches are ordered on the caller's stream (torch events see them)."""
    import torch
    rows, cols, p, c, v = synth.poisson2d(200)
    x = synth.x_vector(cols, seed=1)
    s = torch.cuda.Stream()
    with capi.Context(0) as ctx:
        ctx.set_stream(s.cuda_stream)
        ctx.upload_csr(rows, cols, p, c, v)
        ctx.set_x(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
            ctx.run(50, sync=False)
            e1.record()
        s.synchronize()
        assert e0.elapsed_time(e1) > 0.0
        y = ctx.get_y()
        ctx.set_stream(None)  # back on its own stream
        ctx.run()
        y2 = ctx.get_y()
    want = oracle.csr_spmv(rows, p, c, v, x, runs=50)
    assert_bitexact(y, want, "50 runs on the caller's stream")
    assert_bitexact(y2, oracle.csr_spmv(rows, p, c, v, x, y=want), "51st run on the own stream")


def test_multi_gpu_context_with_one_device(oracle):
    """spmv_hip_create_multi: the in-process row partition + in-place RCCL all-gather.  This box has one
    GPU, so G = 1: without RCCL (the gather is a no-op) and, with SPMV_HIP_FORCE_RCCL=1, with
    ncclCommInitAll + a grouped ncclAllGather over one device -- the calls the G > 1 path makes."""
    import os
    rows, cols, p, c, v = synth.powerlaw(30000, 30000, seed=3)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3)
    scale = 3 * abs_products(rows, p, c, v, x) + np.abs(y0)
    for force in ("0", "1"):
        os.environ["SPMV_HIP_FORCE_RCCL"] = force
        try:
            with capi.Context(num_gpus=1) as ctx:
                ctx.upload_csr(rows, cols, p, c, v)
                ctx.set_x(x)
                ctx.set_y(y0)
                ctx.run(3)
                assert_close(ctx.get_y(), want, scale, what="multi ctx, G=1, rccl=%s" % force)
                info = ctx.info()
                assert info["devices"] == 1 and info["format"] == 1 and info["rows"] == rows
                k_ns, g_ns = ctx.last_run_times()
                assert k_ns > 0 and g_ns < 5_000_000
                i, j, a = synth.csr_to_coordinate(rows, p, c, v)
                ctx.upload_coo(rows, cols, i - 1, j - 1, a)  # a second upload (another format) replaces the first
                ctx.set_x(x)
                ctx.run()
                assert_close(ctx.get_y(), oracle.csr_spmv(rows, p, c, v, x, num_threads=4), scale, what="COO in a multi ctx")
                ctx.upload_csr(rows, cols, p, c, v)
                ctx.set_x(x)
                ctx.run()
                assert_close(ctx.get_y(), oracle.csr_spmv(rows, p, c, v, x, num_threads=4), scale, what="re-upload")
        finally:
            os.environ.pop("SPMV_HIP_FORCE_RCCL", None)
    with pytest.raises(capi.SpmvHipError) as e:
        capi.Context(num_gpus=capi.device_count() + 1)
    assert e.value.code == capi.ERR_INVALID
    # round 6: the PIPELINED RCCL gather -- ncclAllGather on the second stream, behind an event, beside the next run's multiply -- with
    # the one device this box has (only under SPMV_HIP_FORCE_RCCL=1; otherwise one device never pipelines): the serial order's bits
    os.environ["SPMV_HIP_FORCE_RCCL"] = "1"
    try:
        # (a 27-point stencil: row-owned tiles, no atomics -- the same bits on every run; the power-law matrix above has rows whose
        # chunks meet in atomics and is compared within the tolerance)
        srows, scols, sp, sc, sv = synth.stencil27_like(29, 31, 23)
        sx, sy0 = synth.x_vector(scols, seed=5), synth.x_vector(srows, seed=6)
        ys = {}
        for name, extra in (("serial", 0), ("pipelined", capi.FLAG_PIPELINE_GATHER)):
            with capi.Context(num_gpus=1, flags=extra) as ctx:
                ctx.upload_csr(srows, scols, sp, sc, sv)
                ctx.set_x(sx)
                ctx.set_y(sy0)
                ctx.run(5)
                ys[name] = ctx.get_y()
                info = ctx.info()
                assert info["pipelined"] == (1 if extra else 0) and info["rccl_ranks"] == 1, info
                ctx.upload_csr(rows, cols, p, c, v)
                ctx.set_x(x)
                ctx.set_y(y0)
                ctx.run(2)
                assert_close(ctx.get_y(), oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2), scale, what="%s rccl, G=1, power-law rows" % name)
        assert_bitexact(ys["pipelined"], ys["serial"], "pipelined RCCL gather with one device")
        assert_close(ys["pipelined"], oracle.csr_spmv(srows, sp, sc, sv, sx, y=sy0, num_threads=4, runs=5), 5 * abs_products(srows, sp, sc, sv, sx) + np.abs(sy0),
                     what="pipelined rccl, G=1, five runs")
    finally:
        os.environ.pop("SPMV_HIP_FORCE_RCCL", None)
    with capi.Context(num_gpus=1, flags=capi.FLAG_PIPELINE_GATHER) as ctx:  # without the forced collective: nothing to pipeline
        ctx.upload_csr(rows, cols, p, c, v)
        assert ctx.info()["pipelined"] == 0


def test_cli_synthetic_full_size_gpus_and_check():
    """The C++ CLI at full size without a file: --synthetic poisson2d:4096 on the GPU, the reference's
    timed loop (sync per run), --gpus 1 (multi-GPU context), --x uniform --check against the CPU kernel."""
    import json
    import subprocess
    import hostlib
    r = subprocess.run([hostlib.CLI, "--synthetic", "poisson2d:4096", "--spmv-format", "hip-csr", "--gpus", "1", "--threads", "1",
                        "--profile", "5", "--x", "uniform", "--check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["kernel"]["name"] == "hip-csr-spmv" and d["kernel"]["rows"] == 16777216 and d["kernel"]["nonzeros"] == 83869696
    assert d["kernel"]["device"]["gpus"] == 1 and "last_run_all_gather_ns" in d["kernel"]["device"]
    assert d["execution_time"]["samples"] == 5 and 50_000 < d["execution_time"]["median"] < 2_000_000
    assert d["parity"]["pass"] is True and d["parity"]["max_relative_error"] <= 1e-10
    # SPMV_DEVICE=hip flips the default of the README spelling; NaN in the result fails the gate (not hides in it)
    r = subprocess.run([hostlib.CLI, "--synthetic", "webbase:20000,62000,300,75", "--spmv-format", "coo", "--threads", "1", "--profile", "2",
                        "--check"], env=dict(os.environ, SPMV_DEVICE="hip"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["kernel"]["name"] == "hip-coo-spmv"
    # --gpus 4 --peer-gather, the four row blocks rehearsed on this box's one device: the G > 1 flow end to end
    r = subprocess.run([hostlib.CLI, "--synthetic", "kkt:30", "--spmv-format", "hip-csr", "--gpus", "4", "--peer-gather", "--threads", "1",
                        "--profile", "3", "--x", "uniform", "--check"], env=dict(os.environ, SPMV_HIP_SHARE_DEVICES="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["kernel"]["device"]["gpus"] == 4 and d["parity"]["pass"] is True and d["parity"]["max_relative_error"] <= 1e-10


@pytest.mark.parametrize("gather", ["push", "fused", "push-pipelined"])
@pytest.mark.parametrize("parts,balance", [(2, False), (3, False), (8, False), (3, True), (8, True)])
def test_multi_gpu_context_partition_rehearsed_on_one_device(oracle, parts, balance, gather):
    """The G > 1 flow of spmv_hip_create_multi -- ceil(rows / G) row blocks (src/matrix/csr-matrix.cpp:77-95), one
    plan per block, y slots, the gather, set_y / get_y -- with every part on this box's one device
    (SPMV_HIP_SHARE_DEVICES=1) and the gather done by the peer-push kernel (SPMV_HIP_FLAG_PEER_GATHER; RCCL cannot put
    two ranks on a device).  Rows are not a multiple of G, the last block is short, one block is empty of entries.
    balance: SPMV_HIP_FLAG_BALANCE_ENTRIES, blocks of equal stored entries (unequal rows, padded y slots).
    gather "fused": SPMV_HIP_FLAG_FUSED_PEER_STORE -- every part's run delivers its rows itself; on a 27-point stencil
    (the default kernel: row sums forwarded by the multiply) and on power-law rows (balanced tiles: pushed behind it).
    gather "push-pipelined" (round 6): SPMV_HIP_FLAG_PIPELINE_GATHER -- two alternating copies of y per part, the push of run k on a
    second stream beside the multiply of run k + 1; the same bits as the serial order."""
    import os
    if gather == "fused" and not balance:
        rows, cols, p, c, v = synth.stencil27_like(31, 29, 33)
    else:
        rows, cols, p, c, v = synth.powerlaw(30011, 30011, seed=13)
    # rows of the second block hold nothing: a part without entries
    chunk = -(-rows // parts)
    lens = np.diff(p).astype(np.int64)
    if parts >= 3:
        keep = np.ones(len(c), dtype=bool)
        keep[p[chunk]:p[2 * chunk]] = False
        lens[chunk:2 * chunk] = 0
        c, v = c[keep], v[keep]
        p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=3)
    scale = 3 * abs_products(rows, p, c, v, x) + np.abs(y0)
    os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
    try:
        gflag = {"push": capi.FLAG_PEER_GATHER, "fused": capi.FLAG_FUSED_PEER_STORE,
                 "push-pipelined": capi.FLAG_PEER_GATHER | capi.FLAG_PIPELINE_GATHER}[gather]
        # VERIFY_PLAN on a multi-GPU context: get_y compares EVERY part's copy of y with part 0's, bit for bit
        with capi.Context(num_gpus=parts, flags=gflag | capi.FLAG_VERIFY_PLAN | (capi.FLAG_BALANCE_ENTRIES if balance else 0)) as ctx:
            ctx.upload_csr(rows, cols, p, c, v)
            ctx.set_x(x)
            ctx.set_y(y0)
            ctx.run(3)
            assert_close(ctx.get_y(), want, scale, what="G=%d on one device, balance=%s, gather %s" % (parts, balance, gather))
            info = ctx.info()
            assert info["devices"] == parts and info["rows"] == rows and info["stored"] == len(c)
            k_ns, g_ns = ctx.last_run_times()
            assert k_ns > 0
            # a second y replaces the first on every part; the next run starts from it everywhere
            ctx.set_y(np.zeros(rows))
            ctx.run()
            assert_close(ctx.get_y(), oracle.csr_spmv(rows, p, c, v, x, num_threads=4), scale, what="after set_y")
        with pytest.raises(capi.SpmvHipError) as e:  # sharing devices is a rehearsal of the peer gather only
            capi.Context(num_gpus=capi.device_count() + 1)
        assert e.value.code == capi.ERR_INVALID
    finally:
        os.environ.pop("SPMV_HIP_SHARE_DEVICES", None)


def test_pipelined_gather_gives_the_serial_bits(oracle):
    """SPMV_HIP_FLAG_PIPELINE_GATHER over many back-to-back runs (the case it exists for: K runs, one sync): 5 row blocks of a
    27-point stencil and of a COO upload (row-major tiles), 1 ... 9 runs from a given y, odd and even counts (the current copy of y
    alternates), set_y in between, every part's copy compared under VERIFY_PLAN -- and the SAME BITS as the context without the
    flag.  A hybrid upload that keeps its COO remainder apart has no y_in != y_out form: the flag is ignored, the serial order kept."""
    import os
    rows, cols, p, c, v = synth.stencil27_like(37, 23, 29)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    i, j, a = synth.csr_to_coordinate(rows, p, c, v)
    os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
    try:
        got = {}
        for name, extra in (("serial", 0), ("pipelined", capi.FLAG_PIPELINE_GATHER)):
            with capi.Context(num_gpus=5, flags=capi.FLAG_PEER_GATHER | capi.FLAG_VERIFY_PLAN | extra) as ctx:
                ctx.upload_csr(rows, cols, p, c, v)
                ctx.set_x(x)
                out = []
                for runs in (1, 2, 9, 4):
                    ctx.set_y(y0)
                    ctx.run(runs)
                    out.append(ctx.get_y())
                    ctx.run(3)  # ... and on from there without a set_y in between
                    out.append(ctx.get_y())
                ctx.upload_coo(rows, cols, i - 1, j - 1, a)
                ctx.set_x(x)
                ctx.set_y(y0)
                ctx.run(7)
                out.append(ctx.get_y())
                got[name] = out
        for k, (a_, b_) in enumerate(zip(got["serial"], got["pipelined"])):
            assert_bitexact(b_, a_, "pipelined against serial, vector %d" % k)
        for runs, y in zip((1, 4, 2, 5, 9, 12, 4, 7), got["pipelined"][:8]):
            want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=runs)
            assert_close(y, want, runs * abs_products(rows, p, c, v, x) + np.abs(y0), what="%d pipelined runs" % runs)
    finally:
        os.environ.pop("SPMV_HIP_SHARE_DEVICES", None)


@pytest.mark.parametrize("spec,gather", [("synthetic:kkt:200", "fused"), ("synthetic:kkt:199", "fused"), ("synthetic:kkt:199", "push"),
                                         ("synthetic:kkt:199", "push-pipelined")])
def test_eight_row_blocks_of_the_kkt_matrix_on_one_device(oracle, spec, gather):
    """BASELINE configs[3] as the first 8-GPU run will see it -- nlpkkt200's stand-in in EIGHT row blocks (ceil(rows / 8) rows
    each, src/matrix/csr-matrix.cpp:77-95), eight plans, eight copies of y, the fused peer store / the push -- rehearsed with all
    eight parts on this box's one device.  kkt:200 divides evenly (8 x 2 030 000 rows); kkt:199 does not (15 998 804 rows: the
    last block is 4 rows short), which is the uneven-last-chunk case at G = 8.  Whole vector against the CPU kernel; under
    VERIFY_PLAN get_y also compares all eight copies of y bit for bit."""
    import os
    A = hostapi.load(spec, "csr")
    rows, cols = A.rows, A.cols
    assert (rows % 8 == 0) == (spec.endswith("200")) and A.stored > 400000000
    x = synth.x_vector(cols, seed=12345)
    want = oracle.csr_spmv(rows, A.row_ptr, A.column_index, A.value, x, num_threads=THREADS, runs=2)
    os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
    try:
        gflag = {"fused": capi.FLAG_FUSED_PEER_STORE, "push": capi.FLAG_PEER_GATHER, "push-pipelined": capi.FLAG_PEER_GATHER | capi.FLAG_PIPELINE_GATHER}[gather]
        with capi.Context(num_gpus=8, flags=gflag | capi.FLAG_VERIFY_PLAN) as ctx:
            ctx.upload_csr(rows, cols, A.row_ptr, A.column_index, A.value)
            ctx.set_x(x)
            ctx.run(2)
            got = ctx.get_y()
            info = ctx.info()
            assert info["devices"] == 8 and info["rows"] == rows and info["stored"] == A.stored
            k_ns, g_ns = ctx.last_run_times()
            assert k_ns > 0
    finally:
        os.environ.pop("SPMV_HIP_SHARE_DEVICES", None)
    assert_close(got, want, scale=2 * abs_products(rows, A.row_ptr, A.column_index, A.value, x), what="%s in 8 row blocks, %s" % (spec, gather))
    A.close()


@pytest.mark.parametrize("fmt,parts,balance", [("coo", 3, False), ("coo", 8, True), ("ell", 2, False), ("ell", 5, False),
                                               ("hybrid", 3, False), ("hybrid", 4, True)])
def test_multi_gpu_context_coo_and_ellpack_blocks(oracle, fmt, parts, balance):
    """COO and ELLPACK through spmv_hip_create_multi ("COO: split the row-sorted stream at row boundaries;
    ELL: row range"), rehearsed with every block on this box's one device: the triplets come shuffled, are dealt to
    the blocks of their rows and rebased; the ELLPACK arrays are cut by rows.  y against the oracle on the whole
    matrix, two accumulating runs from a given y."""
    import os
    rng = np.random.default_rng(5)
    if fmt in ("coo", "hybrid"):
        rows, cols, p, c, v = synth.powerlaw(20011, 20011, seed=21)
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        perm = rng.permutation(len(a))
        i, j, a = i[perm] - 1, j[perm] - 1, a[perm]
    else:
        rows, cols, p, c, v = synth.banded(15013, [-700, -3, -1, 0, 1, 2, 900], seed=22)
    x = synth.x_vector(cols, seed=5)
    y0 = synth.x_vector(rows, seed=6)
    want = oracle.csr_spmv(rows, p, c, v, x, y=y0, num_threads=4, runs=2)
    scale = 2 * abs_products(rows, p, c, v, x) + np.abs(y0)
    os.environ["SPMV_HIP_SHARE_DEVICES"] = "1"
    try:
        with capi.Context(num_gpus=parts, flags=capi.FLAG_PEER_GATHER | (capi.FLAG_BALANCE_ENTRIES if balance else 0)) as ctx:
            if fmt == "coo":
                ctx.upload_coo(rows, cols, i, j, a)
            elif fmt == "hybrid":
                # the reference's hybrid converter (src/matrix/hybrid-matrix.cpp:316-417) on the whole matrix; the context cuts
                # the ELLPACK part by rows, deals the remainder to the blocks of its rows, and every part merges its two
                H = oracle.hybrid_from_coordinate(rows, i + 1, j + 1, a)
                ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"])
            else:
                # row-major ELLPACK like ell_matrix::from_matrix_market (src/matrix/ell-matrix.cpp:190-238): zero
                # padding whose column repeats the row's last real one
                lens = np.diff(p)
                L = int(lens.max())
                ec = np.zeros((rows, L), dtype=np.int32)
                ev = np.zeros((rows, L), dtype=np.float64)
                slot = np.arange(len(c)) - np.repeat(p[:-1], lens)
                rix = np.repeat(np.arange(rows), lens)
                ec[rix, slot] = c
                ev[rix, slot] = v
                last = np.where(lens > 0, c[np.maximum(p[1:] - 1, 0)], 0)
                pad = np.arange(L)[None, :] >= lens[:, None]
                ec[pad] = np.broadcast_to(last[:, None], (rows, L))[pad]
                ctx.upload_ell(rows, cols, L, ec.ravel(), ev.ravel())
            ctx.set_x(x)
            ctx.set_y(y0)
            ctx.run(2)
            assert_close(ctx.get_y(), want, scale, what="%s over %d blocks" % (fmt, parts))
            info = ctx.info()
            assert info["devices"] == parts and info["format"] == {"coo": 2, "ell": 3, "hybrid": 4}[fmt] and info["rows"] == rows
            with pytest.raises(capi.SpmvHipError) as e:  # a bad row index is refused before anything is dealt
                ctx.upload_coo(rows, cols, np.array([rows], dtype=np.int32), np.array([0], dtype=np.int32), np.array([1.0]))
            assert e.value.code == capi.ERR_INVALID
    finally:
        os.environ.pop("SPMV_HIP_SHARE_DEVICES", None)


def test_bench_on_a_matrix_market_file(tmp_path):
    """bench.py --matrix FILE: what a box that has the SuiteSparse files runs.  A symmetric file (the 1138_bus
    stand-in) packed the way SuiteSparse ships (NAME.tar.gz holding NAME/NAME.mtx), stored triangle only and with
    --expand-symmetric, CSR and COO: one JSON line each, named after the file, parity against the CPU kernel."""
    import json
    import shutil
    import subprocess
    import sys
    import tarfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "bus_like"
    member = tmp_path / name
    member.mkdir()
    shutil.copy(os.path.join(root, "tests", "golden", "bus1138_like.mtx"), member / (name + ".mtx"))
    archive = tmp_path / (name + ".tar.gz")
    with tarfile.open(archive, "w:gz") as t:
        t.add(member, arcname=name)
    stored = None
    for extra, fmt in (([], "csr"), (["--expand-symmetric"], "csr"), (["--expand-symmetric"], "coo")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--matrix", str(archive), "--format", fmt, "--steps", "3",
                            "--warmup", "1", "--cpu-seconds", "0.2", "--no-reference-protocol"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["config"]["workload"].startswith(name + ".tar.gz") and d["config"]["rows"] == 1138
        assert d["parity"]["pass"] is True and d["roofline"]["kernel_us"] > 0 and d["cpu_baseline"]["value"] > 0
        assert d["config"]["symmetric_file_expanded"] is bool(extra)
        if not extra:
            stored = d["config"]["nnz"]
            assert stored == 2596  # the reference multiplies what the file stores
        else:
            assert d["config"]["nnz"] == 2 * stored - 1138  # mirrored off-diagonal entries


def test_cli_flush_caches_reaches_the_device():
    """--flush-caches (src/profile-kernel.cpp:181-192, :264) with a hip-* kernel also evicts the device's L2 and
    Infinity Cache between the timed runs: a 65 MB problem that otherwise lives in the Infinity Cache takes
    measurably longer, the result stays the same."""
    import json
    import subprocess
    import hostlib
    spec = "webbase"
    med = {}
    for flush in (False, True):
        r = subprocess.run([hostlib.CLI, "--synthetic", spec, "--spmv-format", "hip-csr", "--threads", "1", "--profile", "15", "--check"]
                           + (["--flush-caches"] if flush else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout)
        assert d["parity"]["pass"] is True
        med[flush] = d["execution_time"]["median"]
    assert med[True] > 1.05 * med[False], med  # cold: everything comes from HBM (measured: 42.9 -> 48.5 us)
