#!/usr/bin/env python3
"""What does a wave of the headline launch spend its life on?  With a library built from
tools/probes/wave_trace.patch (experiments build, -DSPMV_WAVE_TRACE) every wave of the lane-per-row dictionary path
writes seven time stamps of the chip's 100 MHz constant clock:

    0 wave started   1 descriptor pair back   2 (indexed stencil tiles) index bytes back
    3 (indexed stencil tiles) first round of x back / (general path) products parked in LDS: streams and x are back
    4 all products added / row sums done   5 old y back   6 y stores issued

    SPMV_HIP_EXPERIMENTS=tools/ablate/wave_trace.so python tools/wave_trace.py [--matrix synthetic:poisson2d:4096]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="synthetic:poisson2d:4096")
    args = ap.parse_args()
    import torch
    from spmv_amd import capi, hostapi, synth
    lib = capi.load()
    if not hasattr(lib, "spmv_hip_experiment_wave_trace"):
        sys.exit("wave_trace.py: the loaded library has no trace hook (build it from tools/probes/wave_trace.patch)")
    fn = lib.spmv_hip_experiment_wave_trace
    fn.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c, v = A.rows, A.cols, A.row_ptr, A.column_index, A.value
    dev = torch.device("cuda:0")
    tp, tc, tv = (torch.from_numpy(np.asarray(t)).to(dev) for t in (p, c, v))
    tx = torch.from_numpy(synth.x_vector(cols, "uniform", seed=12345)).to(dev)
    ty = torch.zeros(rows, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, 0)
    plan.compress(tc.data_ptr(), stream)
    plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
    plan.index_values(tv.data_ptr(), stream)
    info = plan.info()
    ntiles = info["row_blocks"]
    ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
    for _ in range(5):
        plan.spmv(*ptrs, stream)
    torch.cuda.synchronize()
    assert fn(None, ntiles, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.spmv(*ptrs, stream)
    e1.record()
    torch.cuda.synchronize()
    n = min(ntiles, 1 << 20)
    out = np.zeros((n, 8), dtype=np.uint64)
    assert fn(out.ctypes.data, n, 0) == 0
    t = out[:, :7].astype(np.int64)
    ok = t[:, 0] > 0  # waves of the traced path (the others took another branch)
    t = t[ok]
    tick = 10.0  # ns per tick of the 100 MHz clock
    t0 = t[:, 0].min()
    names = ["descriptor back", "index bytes back", "x back / products parked", "products added / row sums done", "old y back", "stores issued"]
    print("%s: %d tiles, %d traced waves, launch %.1f us by events, first wave start to last store %.1f us" % (
        args.matrix, ntiles, len(t), e0.elapsed_time(e1) * 1e3, (t[:, 6].max() - t0) * tick / 1e3))
    life = (t[:, 6] - t[:, 0]) * tick / 1e3
    res = {"waves": int(len(t)), "lifetime_us": {}}
    print("wave lifetime (start -> stores issued): mean %.2f us, median %.2f, p10 %.2f, p90 %.2f" % (
        life.mean(), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
    res["lifetime_us"] = {"mean": float(life.mean()), "median": float(np.median(life))}
    prev = t[:, 0]
    for i, name in enumerate(names, start=1):
        have = t[:, i] > 0
        if not have.any():
            continue
        since_start = (t[have, i] - t[have, 0]) * tick / 1e3
        step = (t[have, i] - prev[have]) * tick / 1e3
        print("  %-32s mean %.2f us after start (median %.2f)   %+.2f us after the stamp before   (%d waves)" % (
            name, since_start.mean(), np.median(since_start), step.mean(), int(have.sum())))
        res[name] = float(since_start.mean())
        prev = np.where(have, np.maximum(prev, t[:, i]), prev)
    # how many traced waves are alive at a time
    starts, ends = np.sort(t[:, 0]), np.sort(t[:, 6])
    grid = np.linspace(t0, t[:, 6].max(), 200)
    alive = np.searchsorted(starts, grid, side="right") - np.searchsorted(ends, grid, side="right")
    print("waves alive (between start and stores issued): mean %.0f, max %d of 8192 slots" % (alive[10:-10].mean(), alive.max()))
    # order in which tiles start
    order = np.argsort(t[:, 0], kind="stable")
    idx = np.nonzero(ok)[0][order]
    disp = np.abs(idx - np.arange(len(idx)) * (ntiles / max(1, len(idx))))
    print("tile number vs start order: mean |distance| %.0f tiles" % disp.mean())
    res["alive_mean"] = float(alive[10:-10].mean())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
