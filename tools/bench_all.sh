#!/bin/bash
# tools/bench_all.sh TAG -- on the GPU box: bench.py over every BASELINE configuration and format
# (non-profiled; each line carries roofline + cpu_baseline + parity); lines go to gpurun_out/TAG_*.log
TAG=${1:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { name=$1; shift; python3 bench.py --steps 50 --warmup 10 --cpu-seconds 4 "$@" > gpurun_out/${TAG}_$name.log 2> gpurun_out/${TAG}_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/${TAG}_$name.err; }; python3 - gpurun_out/${TAG}_$name.log $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-16s %8.1f us  %7.1f GFLOP/s  frac %.3f  streamed %.3f  of-triad %s  cpu %s GFLOP/s  parity %s  refproto %s" % (
        sys.argv[2], r["kernel_us"], d["value"], r["frac"], r["frac_streamed"], r["frac_of_triad"],
        (d.get("cpu_baseline") or {}).get("value"), (d.get("parity") or {}).get("pass"),
        ((d.get("reference_protocol") or {}).get("execution_time_ns") or {}).get("median")))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
run poisson_csr
run queen_csr --workload queen
run kkt_csr --workload kkt
run webbase_csr --workload webbase
run powerlaw_csr --workload powerlaw
run webbase_coo --workload webbase --format coo
run webbase_hybrid --workload webbase --format hybrid
run powerlaw_coo --workload powerlaw --format coo
run powerlaw_hybrid --workload powerlaw --format hybrid
run poisson_ell --format ell
run poisson_coo --format coo
run random_csr --workload random --no-reference-protocol
