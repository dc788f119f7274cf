#!/bin/bash
# tools/bench_all.sh TAG -- on the GPU box: bench.py over every BASELINE configuration and format
# (non-profiled; each line carries roofline + cpu_baseline + parity); lines go to gpurun_out/TAG_*.log
TAG=${1:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { name=$1; shift; python3 bench.py --steps 50 --warmup 10 --cpu-seconds 4 --no-companions --no-config3 --no-drop-in ${LIVE_PMC:---no-live-pmc} "$@" > gpurun_out/${TAG}_$name.log 2> gpurun_out/${TAG}_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/${TAG}_$name.err; }; python3 - gpurun_out/${TAG}_$name.log $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    st = r["streamed"]
    comp = r.get("compressed") if isinstance(r.get("compressed"), dict) else None
    print("%-20s %8.1f us (cold %s)  %7.1f GFLOP/s  frac(8d) %.3f  streamed %.3f (%.2f of triad)  cpu %s GFLOP/s  parity %s  refproto %s%s" % (
        sys.argv[2], r["kernel_us"], (r.get("cold") or {}).get("kernel_us_median"), d["value"], r["frac"], st["frac"], st["frac_of_triad"] or 0,
        (d.get("cpu_baseline") or {}).get("value"), (d.get("parity") or {}).get("pass"),
        ((d.get("reference_protocol") or {}).get("execution_time_ns") or {}).get("median"),
        ("  | compressed %.1f us %.0f GFLOP/s" % (comp["kernel_us"], comp["gflops"])) if comp else ""))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
run poisson_csr
run poisson_csr_product --headline product
run queen_csr --workload queen
run kkt_csr --workload kkt
# round 6: the stored lower triangles -- what the reference multiplies when it is handed the symmetric SuiteSparse files
run queen_stored_csr --workload queen_stored
run kkt_stored_csr --workload kkt_stored
run webbase_csr --workload webbase
run powerlaw_csr --workload powerlaw
run webbase_coo --workload webbase --format coo
run webbase_hybrid --workload webbase --format hybrid
run powerlaw_coo --workload powerlaw --format coo
run powerlaw_hybrid --workload powerlaw --format hybrid
run poisson_ell --format ell
run poisson_coo --format coo
run random_csr --workload random --no-reference-protocol
run banded_csr --workload banded
run random24_csr --workload random24 --no-reference-protocol
# pessimistic twins of the stand-ins (VERDICT r02 task 1b): what the friendlier assumptions are worth
run poisson_csr_hashed --matrix synthetic:poisson2d:4096,1
run kkt_csr_noshift --workload kkt --flags 0x400
run kkt_csr_jitter50 --matrix synthetic:kkt:200,50
run queen_csr_jitter6 --matrix synthetic:queen:110,71,177,6
run queen_csr_noblocks --workload queen --flags 0x2000000
# round 5: the less tidy twins of the queen-like stand-in (masked block tiles) and the 7-point Laplacian on a 256^3 grid (masked stencil tiles)
run queen_csr_broken2pct --matrix synthetic:queen:110,71,177,3,20
run queen_csr_oddnodes --matrix synthetic:queen:110,71,177,3,20,1000
run mesh_2dof_csr --matrix synthetic:queen:100,80,70,3,0,0,2
run mesh_4dof_csr --matrix synthetic:queen:100,80,60,3,0,0,4
run poisson3d_csr --matrix synthetic:poisson3d:256
# the launches of tests/test_gpu_perf_floor.py measured on this box -- measured and logged only: updating the committed table
# (tests/golden/perf_floor.json) is an explicit, reviewed step (python3 tools/perf_floor.py --write), never a side effect of a
# sweep, or a regressed build run through this script would loosen the very floor that exists to catch it
python3 tools/perf_floor.py > gpurun_out/${TAG}_perf_floor.log 2>&1
# the SuiteSparse files themselves, where a box has them (same switch as tests/test_gpu_realfiles.py)
if [ -n "$SPMV_SUITESPARSE_DIR" ]; then
  for name in 1138_bus Queen_4147 nlpkkt200 webbase-1M; do
    for f in "$SPMV_SUITESPARSE_DIR/$name.mtx" "$SPMV_SUITESPARSE_DIR/$name.mtx.gz" "$SPMV_SUITESPARSE_DIR/$name.tar.gz" "$SPMV_SUITESPARSE_DIR/$name/$name.mtx"; do
      if [ -e "$f" ]; then
        run file_${name}_csr --matrix "$f"
        [ "$name" != "webbase-1M" ] && run file_${name}_csr_expanded --matrix "$f" --expand-symmetric
        [ "$name" = "webbase-1M" ] && { run file_${name}_coo --matrix "$f" --format coo; run file_${name}_hybrid --matrix "$f" --format hybrid; }
        break
      fi
    done
  done
fi
