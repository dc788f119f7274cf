#!/bin/bash
# tools/profile_gpu.sh TAG [bench.py args...]
# Runs on the GPU box.  Separate rocprofv3 passes over the same bench.py command: kernel trace +
# stats; FETCH_SIZE; WRITE_SIZE (the TCC block cannot hold both in one pass, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"); and, with PROFILE_EXTRA=1, the L2 hit rate, the L1->L2 request counts and
# the wave-cycle split (waiting on memory / waiting to issue / issuing).  tools/summarize_rocprof.py
# condenses them into gpurun_out/TAG_summary.{json,md}.  Copy those into profiles/ to keep them.
# Counter passes never run together with a trace domain other than --kernel-trace.
set -o pipefail
TAG=${1:-prof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps ${PROFILE_STEPS:-40} --warmup 5 --no-cpu-baseline --no-reference-protocol --no-live-pmc $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -20 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -20 "$OUT/pmc_write.log"; exit 1; }
if [ "${PROFILE_EXTRA:-0}" = "1" ]; then
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_l2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_l2.log" 2>&1 || tail -5 "$OUT/pmc_l2.log"
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d "$OUT/pmc_l1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_l1.log" 2>&1 || tail -5 "$OUT/pmc_l1.log"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1 || tail -5 "$OUT/pmc_sq.log"
fi
python3 "$ROOT/tools/summarize_rocprof.py" "$OUT" "$TAG" > "$OUT/summarize.log" 2>&1 || { tail -20 "$OUT/summarize.log"; exit 1; }
cat "$ROOT/gpurun_out/${TAG}_summary.md"
# keep the merged-back payload small: drop the raw per-dispatch traces
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
