#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_gpu.sh into one small JSON + markdown.

Per kernel: calls, total / average duration (kernel-trace + --stats), and HBM traffic per launch
from the PMC passes.  Counter handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read, so the read side is doubled before it is compared with a byte count
(WRITE_SIZE is exact for streaming stores).  Both the raw and the corrected figure are kept.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_stats(out):
    rows = []
    for f in find(os.path.join(out, "stats"), "*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    return rows


def kernel_trace(out):
    """Fallback / cross-check: per-dispatch durations from the trace itself."""
    agg = defaultdict(list)
    for f in find(os.path.join(out, "stats"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            try:
                agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            except (KeyError, ValueError):
                pass
    return agg


def pmc(out, sub, counter):
    """Counter values per kernel: list of per-dispatch values."""
    agg = defaultdict(list)
    for f in find(os.path.join(out, sub), "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    out, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(out)
    stats = kernel_stats(out)
    trace = kernel_trace(out)
    fetch = pmc(out, "pmc_fetch", "FETCH_SIZE")
    write = pmc(out, "pmc_write", "WRITE_SIZE")
    bench = None
    for line in open(os.path.join(out, "stats.log")):
        if line.startswith("{") and '"metric"' in line:
            bench = json.loads(line)
    kernels = []
    names = set(trace) | set(fetch) | set(write)
    for name in sorted(names):
        if "spmv" not in name:
            continue
        d = trace.get(name, [])
        k = {"kernel": short(name), "calls": len(d)}
        if d:
            d_sorted = sorted(d)
            k["avg_us"] = round(sum(d) / len(d) / 1e3, 3)
            k["median_us"] = round(d_sorted[len(d) // 2] / 1e3, 3)
            k["min_us"] = round(d_sorted[0] / 1e3, 3)
            k["total_ms"] = round(sum(d) / 1e6, 3)
        for r in stats:
            if r.get("Name") == name:
                k["stats_calls"] = int(r["Calls"])
                k["stats_avg_us"] = round(float(r["AverageNs"]) / 1e3, 3)
                k["stats_pct"] = float(r["Percentage"])
        if name in fetch and fetch[name]:
            f = fetch[name]
            k["FETCH_SIZE_KiB_per_launch"] = round(sum(f) / len(f), 1)
            k["fetch_bytes_raw"] = int(sum(f) / len(f) * 1024)
            k["fetch_bytes_corrected_x2"] = int(sum(f) / len(f) * 1024 * 2)
        if name in write and write[name]:
            w = write[name]
            k["WRITE_SIZE_KiB_per_launch"] = round(sum(w) / len(w), 1)
            k["write_bytes"] = int(sum(w) / len(w) * 1024)
        # optional diagnostic passes (PROFILE_EXTRA=1): per-launch means of every counter collected
        for sub in ("pmc_l2", "pmc_l1", "pmc_sq"):
            for f in find(os.path.join(out, sub), "*counter_collection.csv"):
                acc = defaultdict(list)
                for r in csv.DictReader(open(f)):
                    if r.get("Kernel_Name") == name:
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for cname, vals in acc.items():
                    k.setdefault("counters", {})[cname] = round(sum(vals) / len(vals), 1)
        c = k.get("counters", {})
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            k["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if c.get("SQ_WAVE_CYCLES"):
            k["wave_cycle_split"] = {"waiting_on_memory": round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
                                     "waiting_to_issue": round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
                                     "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)}
        if "fetch_bytes_corrected_x2" in k and "write_bytes" in k:
            k["hbm_traffic_bytes_per_launch"] = k["fetch_bytes_corrected_x2"] + k["write_bytes"]
            k["hbm_traffic_bytes_per_launch_uncorrected"] = k["fetch_bytes_raw"] + k["write_bytes"]
        kernels.append(k)
    import time
    summary = {"tag": tag, "sequence": int(time.time()), "build": (bench or {}).get("build"), "bench_line": bench, "kernels": kernels,
               "note": "durations from rocprofv3 --kernel-trace --stats; traffic from separate --pmc passes; "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"}
    json.dump(summary, open(os.path.join(root, tag + "_summary.json"), "w"), indent=1)
    with open(os.path.join(root, tag + "_summary.md"), "w") as f:
        f.write("# rocprofv3 summary `%s`\n\n" % tag)
        if bench:
            rl = bench.get("roofline", {})
            f.write("bench.py: %s = %.2f %s, ms/step %.5f, roofline achieved %.1f GB/s (frac %.4f; algorithmic %.4f), "
                    "kernel_us (HIP events) %.2f, streamed bytes/launch %d, algorithmic bytes/launch %d\n\n" % (
                        bench["metric"], bench["value"], bench["unit"], bench["ms_per_step"],
                        rl.get("achieved", 0), rl.get("frac", 0), rl.get("frac_algorithmic", 0), rl.get("kernel_us", 0),
                        rl.get("streamed_bytes_per_launch", 0), rl.get("algorithmic_bytes_per_launch", 0)))
            b = bench.get("build") or {}
            f.write("build: source_sha256 %s, lib_sha256 %s, git %s\n\n" % (b.get("source_sha256"), b.get("lib_sha256"), b.get("git_head")))
        f.write("| kernel | calls | avg us | median us | min us | FETCH KiB | WRITE KiB | HBM bytes/launch (fetch x2 + write) |\n")
        f.write("|---|---|---|---|---|---|---|---|\n")
        for k in kernels:
            f.write("| %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
                k["kernel"], k.get("calls"), k.get("avg_us"), k.get("median_us"), k.get("min_us"),
                k.get("FETCH_SIZE_KiB_per_launch"), k.get("WRITE_SIZE_KiB_per_launch"),
                k.get("hbm_traffic_bytes_per_launch")))
        extra = [k for k in kernels if "counters" in k]
        if extra:
            f.write("\nDiagnostic counters (per-launch means; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); wave-cycle split per MI355X_MICROARCH.md \"rocprofv3 PMC slots\"):\n\n")
            for k in extra:
                f.write("* `%s`: L2 hit rate %s, wave cycles %s, counters %s\n" % (
                    k["kernel"], k.get("l2_hit_rate"), json.dumps(k.get("wave_cycle_split")), json.dumps(k["counters"])))
        f.write("\n" + summary["note"] + "\n")
    print(open(os.path.join(root, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
