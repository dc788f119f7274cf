#!/bin/bash
# tools/profile_gpu.sh TAG [bench.py args...]
# Runs on the GPU box.  Three separate rocprofv3 passes over the same bench.py command
# (kernel trace + stats; FETCH_SIZE; WRITE_SIZE -- the TCC block cannot hold both counters in one
# pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), then tools/summarize_rocprof.py condenses
# them into gpurun_out/TAG_summary.{json,md}.  Copy those into profiles/ to keep them.
set -o pipefail
TAG=${1:-prof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps 40 --warmup 5 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -20 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -20 "$OUT/pmc_write.log"; exit 1; }
python3 "$ROOT/tools/summarize_rocprof.py" "$OUT" "$TAG" > "$OUT/summarize.log" 2>&1 || { tail -20 "$OUT/summarize.log"; exit 1; }
cat "$ROOT/gpurun_out/${TAG}_summary.md"
# keep the merged-back payload small: drop the raw per-dispatch traces
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
