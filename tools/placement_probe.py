#!/usr/bin/env python3
"""Does the launch time of one matrix depend on WHERE its arrays lie?  (profiles/r05_ab_segs12.log: the same library on the same
box ran synthetic:kkt:200 in 747 / 775 / 798 us in three processes.)  One process, the matrix uploaded several times -- each time
behind spacer allocations of another size, the earlier copies kept or freed -- the default plan built and timed on every copy;
prints the time with the addresses of the arrays.

    python tools/placement_probe.py [--matrix synthetic:kkt:200] [--copies 6]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="synthetic:kkt:200")
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--keep", type=int, default=1, help="keep the earlier copies allocated (1) or free them (0)")
    ap.add_argument("--flags", type=lambda s: int(s, 0), default=0x100000)
    ap.add_argument("--vary", default="", help="comma list of y,x,v,plan: one upload, then only the named array re-allocated per trial")
    ap.add_argument("--policies", default="", help="semicolon list of policies, each a comma list of array=byte offset (arrays p c v x y), "
                    "e.g. ';y=1114112;x=65536,y=1114112': every policy gets --copies fresh uploads behind spacers")
    args = ap.parse_args()
    if args.vary:
        return vary(args)
    if args.policies:
        return policies(args)
    import torch
    from spmv_amd import capi, hostapi, synth
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c, v = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
    A.close()
    x = synth.x_vector(cols, seed=3)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(1)
    kept = []
    for copy in range(args.copies):
        spacer = torch.empty(int(rng.integers(1, 4000)) * 4096 + (0 if copy % 2 == 0 else 1 << 20), dtype=torch.uint8, device=dev) if copy else None
        tp, tc, tv, tx = (torch.from_numpy(t).to(dev) for t in (p, c, v, x))
        ty = torch.zeros(rows, dtype=torch.float64, device=dev)
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, args.flags)
        plan.compress(tc.data_ptr(), stream)
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
        times = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 20 * 1e3)
        print("copy %d: %8.1f us (rounds %s)  p %#x c %#x v %#x x %#x y %#x" % (
            copy, min(times[1:]), " ".join("%.1f" % t for t in times), *ptrs), flush=True)
        if args.keep:
            kept.append((plan, tp, tc, tv, tx, ty, spacer))
        else:
            plan.close()
            del tp, tc, tv, tx, ty, spacer
            torch.cuda.empty_cache()
    # the first copy again, after everything else
    if args.keep:
        plan, tp, tc, tv, tx, ty, _ = kept[0]
        ptrs = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), ty.data_ptr())
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            print("copy 0 again: %8.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)


def policies(args):
    import torch
    from spmv_amd import capi, hostapi, synth
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c, v = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
    A.close()
    host = {"p": p, "c": c, "v": v, "x": synth.x_vector(cols, seed=3), "y": np.zeros(rows)}
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(4)
    spacers = []
    for pol in args.policies.split(";"):
        offs = dict((kv.split("=")[0], int(kv.split("=")[1], 0)) for kv in pol.split(",") if kv)
        times, ypass, xpass = [], [], []
        for copy in range(args.copies):
            spacers.append(torch.empty(int(rng.integers(1, 3000)) * 4096 * 17, dtype=torch.uint8, device=dev))
            t, raw = {}, []
            for k in "pcvxy":
                h = host[k]
                buf = torch.empty(h.nbytes + offs.get(k, 0), dtype=torch.uint8, device=dev)
                view = buf[offs.get(k, 0):].view(torch.from_numpy(h[:1]).dtype)
                view.copy_(torch.from_numpy(h))
                raw.append(buf)
                t[k] = view
            plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, args.flags)
            plan.compress(t["c"].data_ptr(), stream)
            plan.repack(t["p"].data_ptr(), t["c"].data_ptr(), t["v"].data_ptr(), stream)
            ptrs = tuple(t[k].data_ptr() for k in "pcvxy")

            def launch_time():
                best = None
                for rnd in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        plan.spmv(*ptrs, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / 20 * 1e3
                    if rnd:
                        best = us if best is None else min(best, us)
                return best
            best = launch_time()
            times.append(best)
            # the arrays by themselves: y read + written, x read, by plain elementwise kernels (are the slow copies slow on their own?)
            def plain(fn, n=30):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            ypass.append(plain(lambda: t["y"].add_(0.0)))
            xpass.append(plain(lambda: t["x"].add_(0.0)))
            plan.close()
            del t, raw, plan
            torch.cuda.empty_cache()
        print("policy %-28s: %s   median %.1f  spread %.1f%%" % (pol or "(all 2 MB aligned)", " ".join("%.1f" % u for u in times), float(np.median(times)),
                                                           100.0 * (max(times) - min(times)) / min(times)), flush=True)
        print("    y += 0 alone: %s\n    x += 0 alone: %s" % (" ".join("%.1f" % u for u in ypass), " ".join("%.1f" % u for u in xpass)), flush=True)


def vary(args):
    import torch
    from spmv_amd import capi, hostapi, synth
    A = hostapi.load(args.matrix, "csr")
    rows, cols, p, c, v = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
    A.close()
    x = synth.x_vector(cols, seed=3)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(2)
    t = {"p": torch.from_numpy(p).to(dev), "c": torch.from_numpy(c).to(dev), "v": torch.from_numpy(v).to(dev),
         "x": torch.from_numpy(x).to(dev), "y": torch.zeros(rows, dtype=torch.float64, device=dev)}

    def build():
        plan = capi.CsrPlan(rows, cols, p, capi.CSR_AUTO, 0, args.flags)
        plan.compress(t["c"].data_ptr(), stream)
        plan.repack(t["p"].data_ptr(), t["c"].data_ptr(), t["v"].data_ptr(), stream)
        return plan

    def timed(plan):
        ptrs = tuple(t[k].data_ptr() for k in "pcvxy")
        best = None
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.spmv(*ptrs, stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            if rnd:
                best = us if best is None else min(best, us)
        return best

    plan = build()
    print("base: %.1f us" % timed(plan), flush=True)
    hold = []
    for what in args.vary.split(","):
        for trial in range(args.copies):
            hold.append(torch.empty(int(rng.integers(1, 3000)) * 4096 * 17, dtype=torch.uint8, device=dev))  # a spacer: the next allocation lands elsewhere
            if what == "plan":
                hold.append(plan)
                plan = build()
                where = ""
            else:
                old = t[what]
                t[what] = old.clone()
                hold.append(old)
                where = "%#x" % t[what].data_ptr()
            print("new %-4s trial %d: %8.1f us  %s" % (what, trial, timed(plan), where), flush=True)


if __name__ == "__main__":
    main()
