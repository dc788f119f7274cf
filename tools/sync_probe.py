#!/usr/bin/env python3
"""What does the reference's protocol (launch, wait, launch, wait: src/profile-kernel.cpp:137-179) cost on top of the
kernel?  Wall time of spmv_hip_run + spmv_hip_sync per run through the context API, with and without the
event pair around the launch, and with the device's sync mode set to spin / yield / blocking.
    python3 tools/sync_probe.py [spec ...]         (each variant runs in its own process)"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))

SCHED = {"auto": 0, "spin": 1, "yield": 2, "blocking": 4}


def one(spec, sched, events):
    from spmv_amd import capi, hostapi, synth
    lib = capi.load()
    hip = C.CDLL(capi.hip_runtime_path or "libamdhip64.so")
    if sched != "auto":
        rc = hip.hipSetDeviceFlags(C.c_uint(SCHED[sched]))
        assert rc == 0, rc
    A = hostapi.load(spec)
    with capi.Context(flags=0 if events else capi.FLAG_NO_RUN_EVENTS) as ctx:
        ctx.upload_csr(A.rows, A.cols, np.asarray(A.row_ptr), np.asarray(A.column_index), np.asarray(A.value))
        ctx.set_x(synth.x_vector(A.cols))
        ctx.run(20)
        run, sync, h = lib.spmv_hip_run, lib.spmv_hip_sync, ctx.h
        poll = os.environ.get("SYNC_PROBE_POLL") == "1"
        if poll:  # the caller's own stream, polled with hipStreamQuery instead of a blocking hipStreamSynchronize
            st = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(st), C.c_uint(1)) == 0
            ctx.set_stream(st.value)
            hip.hipStreamQuery.argtypes = [C.c_void_p]
            query = hip.hipStreamQuery
        t = []
        for _ in range(300):
            t0 = time.perf_counter_ns()
            run(h)
            if poll:
                while query(st) != 0:
                    pass
            else:
                sync(h)
            t.append(time.perf_counter_ns() - t0)
        t = np.sort(np.array(t)) / 1e3
        dev = ctx.last_run_ns() / 1e3 if events else float("nan")
        print("%-28s sched %-8s events %d %s  wall median %7.1f us  min %7.1f  p90 %7.1f   device %7.1f us" % (
            spec[10:], sched, events, "poll" if poll else "sync", t[len(t) // 2], t[0], t[int(len(t) * 0.9)], dev), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "--one":
        one(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    specs = sys.argv[1:] or ["synthetic:webbase", "synthetic:poisson2d:4096"]
    for spec in specs:
        for sched in ("auto", "spin", "yield", "blocking"):
            for events in (1, 0):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", spec, sched, str(events)], check=False)
