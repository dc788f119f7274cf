#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SpMV path.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path, y += A*x, over the workload with all inputs
already resident in HBM.  Workload (BASELINE.json configs[1]): Poisson 5-point stencil
CSR on a 4096 x 4096 grid, N = 16 777 216 rows, Z = 83 869 696 stored entries, fp64
values / int32 indices.  With N > 1 GPUs the SAME matrix is row-partitioned
(ceil(rows/N) rows per rank, the reference's static chunk rule), x is replicated, and
every step ends with ONE all-gather of the y segments (RCCL): total work is fixed, so
scaling is "strong".

Prints ONE JSON line on rank 0.  `value` = 2*Z*K / t in GFLOP/s (whole job).
`roofline` prices the local SpMV kernel: algorithmic bytes of one launch
(CSR: 12*Z + 4*(rows+1) + 16*rows + 8*cols, BASELINE.md section 3, for the rank's row
slice) divided by its mean duration measured with HIP events on the launch stream.
`cpu_baseline` (rank 0, N = 1 only) times the reference's own OpenMP CSR kernel
(oracle/_ref, kind "reference") or the C oracle (kind "port") on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
# the host driver only supports dmabuf IPC: without this RCCL cannot share buffers across ranks
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="poisson2d", choices=["poisson2d", "stencil27", "random"])
    ap.add_argument("--grid", type=int, default=4096, help="poisson2d grid edge (4096 = BASELINE configs[1])")
    ap.add_argument("--algorithm", default="auto", choices=["auto", "scalar", "vector", "adaptive", "wavetile"])
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--xcd-remap", action="store_true")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0, help="extra SPMV_HIP_FLAG_* bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores visible to this process")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--events", choices=["launch", "region"], default="region",
                    help="one HIP event pair around the K timed launches (default; mean launch duration = span / K, "
                         "launch gaps included), or a pair around every launch (adds ~5 us of gap per step)")
    ap.add_argument("--partition", choices=["rows", "nnz"], default="rows",
                    help="N > 1: the reference's static row chunks (default), or a split on row boundaries with "
                         "equal stored entries per rank (uneven rows; only for workloads generated whole)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for each all-gather before the next multiply (default: gather k overlaps multiply k+1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo is for rehearsing N > 1 on a box with one GPU")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: every rank uses device 0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the all-gather even with one rank (rehearsal)")
    return ap.parse_args()


def make_rows(args, begin, end):
    """This rank's rows [begin, end) of the workload, plus global sizes."""
    from spmv_amd import synth, partition
    if args.workload == "poisson2d":
        n = args.grid
        rows = n * n
        nr, cols, p, c, v = synth.poisson2d(n, begin, rows if end is None else end)
        return rows, cols, 5 * rows - 4 * n, p, c, v, "poisson2d-5pt-%dx%d-csr" % (n, n)
    if args.workload == "stencil27":
        # nlpkkt200-like stand-in (configs[3]): 27-point stencil, ~16.2M rows, ~436M entries
        rows, cols, p, c, v = synth.stencil27_like(253, 253, 253)
        name = "stencil27-253^3-csr"
    else:
        rows, cols, p, c, v = synth.random_uniform(4000000, 4000000, 24, seed=3)
        name = "random-4M-24perrow-csr"
    nnz = int(p[-1])
    if end is not None:
        p, c, v = partition.csr_slice(p, c, v, begin, end)
    return rows, cols, nnz, p, c, v, name


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota
    (a GPU box exposes all host CPUs but grants one GPU's share of them)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], txt[1]
            else:
                quota, period = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def pmc_traffic(kernel_name, algorithmic_bytes):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_summary.json, written by tools/profile_gpu.sh: FETCH_SIZE x2 + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be read from inside this process;
    the figure is only reported when the summary was taken on the same kernel and workload."""
    import glob
    best = None
    seq = -1
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            rl = (d.get("bench_line") or {}).get("roofline", {})
            if rl.get("algorithmic_bytes_per_launch") != algorithmic_bytes:
                continue
            for k in d["kernels"]:
                if kernel_name in k["kernel"] and "hbm_traffic_bytes_per_launch" in k and d.get("sequence", 0) > seq:
                    seq = d.get("sequence", 0)
                    best = (k["hbm_traffic_bytes_per_launch"], os.path.basename(f))
        except (OSError, ValueError, KeyError):
            continue
    return best


def cpu_baseline(args, rows, cols, p, c, v, x, y_gpu=None):
    """Reference OpenMP CSR kernel (or the C oracle) on the host cores, bounded sample.  Also the
    parity gate: one CPU multiply from y = 0 is compared with the GPU's (y_gpu), whole vector,
    tolerance 1e-10 relative (BASELINE.json).  Returns (cpu_baseline, parity)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    threads = args.cpu_threads or host_cores()
    nnz = int(p[-1])
    budget = args.cpu_seconds
    if oracle_py.RefLib.available():
        R = oracle_py.RefLib()
        A = R.csr_from_arrays(rows, cols, p, c, v)
        t = time.perf_counter()
        if not args.cpu_threads:
            # the box may expose more CPUs than it grants: probe a few team sizes briefly
            best = None
            for cand in sorted({min(threads, 16), min(threads, 32), min(threads, 64), threads}):
                ns, _ = R.csr_spmv_timed(A, x, cand, 2)
                if best is None or np.median(ns) < best[0]:
                    best = (float(np.median(ns)), cand)
            threads = best[1]
        ns, _ = R.csr_spmv_timed(A, x, threads, 2)  # 1 warm-up + 2 timed, to size the sample
        per = max(float(np.median(ns)) * 1e-9, 1e-4)
        runs = int(max(3, min(200, (budget - (time.perf_counter() - t)) / per)))
        ns, _ = R.csr_spmv_timed(A, x, threads, runs)
        ns1, _ = R.csr_spmv_timed(A, x, 1, 3)  # one thread, as BASELINE configs[0] is defined
        y_cpu = R.csr_spmv(A, x, num_threads=threads) if y_gpu is not None else None
        R.csr_free(A)
        kind = "reference"
    else:
        O = oracle_py.Oracle()
        y = np.zeros(rows)
        O.csr_spmv_inplace(rows, p, c, v, x, y, threads)  # warm-up
        ns = []
        t_end = time.perf_counter() + budget
        while len(ns) < 3 or (time.perf_counter() < t_end and len(ns) < 200):
            t0 = time.perf_counter_ns()
            O.csr_spmv_inplace(rows, p, c, v, x, y, threads)
            ns.append(time.perf_counter_ns() - t0)
        ns = np.array(ns)
        ns1 = []
        for _ in range(3):
            t0 = time.perf_counter_ns()
            O.csr_spmv_inplace(rows, p, c, v, x, y, 1)
            ns1.append(time.perf_counter_ns() - t0)
        ns1 = np.array(ns1)
        y_cpu = O.csr_spmv(rows, p, c, v, x, num_threads=threads) if y_gpu is not None else None
        kind = "port"
    med = float(np.median(ns)) * 1e-9
    parity = None
    if y_gpu is not None:
        err = float(np.max(np.abs(y_gpu - y_cpu)) / max(float(np.max(np.abs(y_cpu))), 1e-300))
        parity = {"against": "cpu_baseline kernel (%s), one multiply from y = 0" % kind, "rows_checked": int(rows),
                  "max_rel_err": err, "tolerance": 1e-10, "pass": bool(err <= 1e-10),
                  "bitexact": bool(np.array_equal(y_gpu, y_cpu))}
    return {"value": round(2.0 * nnz / med / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": kind,
            "sample": "full workload, %d timed runs after 1 warm-up, median %.2f ms (min %.2f ms), %d OpenMP threads"
                      % (len(ns), med * 1e3, float(np.min(ns)) * 1e-6, threads),
            "gbs": round((12.0 * nnz + 4 * (rows + 1) + 16.0 * rows + 8.0 * cols) / med / 1e9, 2),
            "single_thread_gflops": round(2.0 * nnz / (float(np.median(ns1)) * 1e-9) / 1e9, 3)}, parity


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from spmv_amd import capi, partition, synth
    from spmv_amd.distributed import DistributedCsrSpmv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; this benchmark has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- workload: this rank's rows, global x ------------------------------------------
    t_setup = time.perf_counter()
    ranges = None
    if args.workload == "poisson2d":
        rows = args.grid * args.grid
    else:
        rows = None
    if rows is not None and world > 1:
        begin, end = partition.row_range(rows, rank, world)
        rows, cols, nnz, p, c, v, wname = make_rows(args, begin, end)
    elif world > 1:
        rows_g, cols, nnz, p, c, v, wname = make_rows(args, 0, None)
        if args.partition == "nnz":
            ranges = partition.nnz_balanced_ranges(p, world)
            begin, end = ranges[rank]
        else:
            begin, end = partition.row_range(rows_g, rank, world)
        p, c, v = partition.csr_slice(p, c, v, begin, end)
        rows = rows_g
    else:
        rows, cols, nnz, p, c, v, wname = make_rows(args, 0, None)
        begin, end = 0, rows
    x = synth.x_vector(cols, "uniform", seed=12345)
    algo = {"auto": capi.CSR_AUTO, "scalar": capi.CSR_SCALAR, "vector": capi.CSR_VECTOR,
            "adaptive": capi.CSR_ADAPTIVE, "wavetile": capi.CSR_WAVETILE}[args.algorithm]
    flags = (capi.FLAG_XCD_REMAP if args.xcd_remap else 0) | args.flags
    op = DistributedCsrSpmv.on_gpu(rows, cols, rank, world, device, p, c, v, x, algo, args.lanes, flags,
                                   overlap=not args.no_overlap, ranges=ranges)
    local_rows, local_nnz = end - begin, int(p[-1])
    local_bytes = synth.csr_bytes(local_rows, cols, local_nnz)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    # ---- one multiply into a zero y, kept on the device: the cpu_baseline leg (the only place the
    # checker libraries under oracle/ are loaded) compares it with the CPU kernel's y -----------------
    y_check = None
    if world == 1 and not args.no_cpu_baseline and not args.no_parity_check:
        op.multiply_local()
        torch.cuda.synchronize()
        y_check = op.y_local[:local_rows].clone()
        op.y_local.zero_()

    # ---- warm-up, then K timed steps -------------------------------------------------------
    for _ in range(args.warmup):
        op.step()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    # with a collective in the step the compute stream also carries the snapshot copy and the wait
    # for the previous gather, so the span is no longer K launches: time every launch there
    per_launch = args.events == "launch" or use_dist
    if not per_launch:
        ev0[0].record()
    for k in range(args.steps):
        if per_launch:
            ev0[k].record()
        op.multiply_local()
        if per_launch:
            ev1[k].record()
        if use_dist:
            if op.overlap:
                op.gather_async()
            else:
                op.gather()
    if not per_launch:
        ev1[0].record()
    op.finish()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if per_launch:
        kernel_ms = np.array([a.elapsed_time(b) for a, b in zip(ev0, ev1)])
    else:
        kernel_ms = np.array([ev0[0].elapsed_time(ev1[0]) / args.steps])

    stats = torch.tensor([elapsed, float(kernel_ms.mean())], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    elapsed_max, kern_ms_max = stats.tolist()

    # N > 1: the collective on its own (after the timed region, not part of `value`): a few
    # blocking all-gathers, max over ranks, so the line shows where a step's time goes.
    gather_us = None
    if use_dist:
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            dist.all_gather_into_tensor(op.y_full, op.send_buf if op.overlap else op.y_local, group=op.group)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - g0)
        gt = torch.tensor([float(np.median(times))], dtype=torch.float64, device=device)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_us = gt.item() * 1e6

    # N > 1: the gathered y must hold what the OTHER ranks computed.  Rank 0 recomputes a strip of
    # the last rank's rows on its own GPU (product kernel, one multiply) and compares it with the
    # gathered segment, which has accumulated warm-up + K multiplies.
    gather_check = None
    if use_dist and world > 1 and rank == 0 and args.workload == "poisson2d":
        ob, oe = partition.row_range(rows, world - 1, world)
        strip = min(4096, oe - ob)
        _, _, ps, cs, vs = synth.poisson2d(args.grid, ob, ob + strip)
        tps, tcs, tvs = (torch.from_numpy(t).to(device) for t in (ps, cs, vs))
        ys = torch.zeros(strip, dtype=torch.float64, device=device)
        plan_s = capi.CsrPlan(strip, cols, ps, algo, args.lanes, flags)
        plan_s.spmv(tps.data_ptr(), tcs.data_ptr(), tvs.data_ptr(), op._keep[3].data_ptr(), ys.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = ys * float(args.steps + args.warmup)
        got = op.y()[ob:ob + strip]
        err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-300))
        gather_check = {"rows_checked": strip, "of_rank": world - 1, "max_rel_err": err, "pass": bool(err <= 1e-10)}
        plan_s.close()

    # N > 1, for information only (never `value`): the same K multiplies with ONE all-gather after the
    # last of them -- what a caller pays who, like the reference's timed loop, looks at y only after
    # the loop.  `value` above gathers after every multiply.
    deferred = None
    if use_dist:
        torch.cuda.synchronize()
        dist.barrier()
        d0 = time.perf_counter()
        for _ in range(args.steps):
            op.multiply_local()
        op.gather()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - d0], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        deferred = {"steps": args.steps, "all_gathers": 1, "ms_total": round(dt.item() * 1e3, 4),
                    "gflops": round(2.0 * nnz * args.steps / dt.item() / 1e9, 2)}

    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        gflops = 2.0 * nnz * args.steps / elapsed_max / 1e9
        kern_s = kern_ms_max * 1e-3  # slowest rank's mean launch duration
        achieved = local_bytes / kern_s / 1e9
        info = op.plan.info()
        out = {
            "metric": "spmv_csr_gflops", "value": round(gflops, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "rows": rows, "cols": cols, "nnz": nnz, "format": "csr",
                       "index_dtype": "int32", "x": "uniform(-1,1) seed 12345",
                       "algorithm": capi.CSR_ALGORITHM_NAMES[info["algorithm"]],
                       "lanes_per_row": info["lanes_per_row"], "workgroups": info["workgroups"],
                       "tiles": info["row_blocks"], "tiles_with_16bit_columns": info["narrow_tiles"],
                       "uniform_tiles": info["uniform_tiles"], "shifted_tiles": info["shifted_tiles"],
                       "tiles_with_x_window": info["xwin_tiles"], "block_window_tiles": info["blockwin_tiles"],
                       "column_panel_tiles": info["panel_tiles"],
                       "partition": ("%s, x replicated, 1 all-gather(y)/step%s" % (
                           ("rows/%d static chunks" % world) if ranges is None else ("%d row ranges of equal stored entries" % world), ", gather k overlaps multiply k+1" if op.overlap else ""))
                       if use_dist else "single GPU", "backend": args.backend if use_dist else None,
                       "rehearsal_shared_gpu": bool(args.share_gpu)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "csr_%s" % capi.CSR_ALGORITHM_NAMES[info["algorithm"]],
                         "kernel_us": round(kern_s * 1e6, 2), "kernel_us_min": round(float(kernel_ms.min()) * 1e3, 2) if per_launch else None,
                         "events": "per launch" if per_launch else "one pair around the %d timed launches" % args.steps,
                         "algorithmic_bytes_per_launch": int(local_bytes),
                         "gflops_kernel_only": round(2.0 * local_nnz / kern_s / 1e9, 1)},
            "hbm_gbs_whole_step": round(synth.csr_bytes(rows, cols, nnz) / (ms_per_step * 1e-3) / 1e9, 1),
            "setup_s": round(setup_s, 1),
        }
        if gather_us is not None:
            recv = 8.0 * op.chunk * (world - 1)
            out["multi_gpu"] = {"local_kernel_us": round(kern_s * 1e6, 2), "all_gather_us": round(gather_us, 2),
                                "all_gather_bytes_received_per_rank": int(recv),
                                "all_gather_gbs_received_per_rank": round(recv / (gather_us * 1e-6) / 1e9, 1) if world > 1 else None,
                                "overlap": bool(op.overlap),
                                "note": "all_gather_us: blocking collective alone, median of 5 after the timed region",
                                "one_all_gather_after_the_k_multiplies": deferred}
        if gather_check:
            out["gather_check"] = gather_check
            if not gather_check["pass"]:
                print(json.dumps(out), flush=True)
                sys.exit("bench.py: gathered y does not match the owning rank's rows")
        tr = pmc_traffic(out["roofline"]["kernel"], int(local_bytes))
        if tr:
            out["roofline"]["traffic"] = tr[0]
            out["roofline"]["traffic_source"] = "profiles/" + tr[1]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(
                args, rows, cols, p, c, v, x, None if y_check is None else y_check.cpu().numpy())
            if out["parity"] and not out["parity"]["pass"]:
                print(json.dumps(out), flush=True)
                sys.exit("bench.py: parity check failed: max relative error %.3e > 1e-10"
                         % out["parity"]["max_rel_err"])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
