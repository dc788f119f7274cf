#!/bin/bash
# tools/ingest_fullsize.sh [SPEC] [TAG] -- on the GPU box: the file path of the drop-in at the scale of BASELINE's
# SuiteSparse configurations.  Writes the generated stand-in as a Matrix Market file (--write-mtx), then runs the
# C++ CLI on that FILE: load (parallel tokeniser) -> CSR (counting sort) -> upload + plan -> timed loop on the
# GPU -> --check against the CPU kernel.  Default: synthetic:kkt:125 (3.9 M rows, 106 M lines, ~3 GB of text).
set -o pipefail
SPEC=${1:-synthetic:kkt:125}
TAG=${2:-r02_ingest}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/spmv-cache-trace_amd/spmv-cache-trace-hip
OUT=$ROOT/gpurun_out/$TAG.log
DIR=$(mktemp -d /tmp/ingest.XXXXXX)
trap 'rm -rf "$DIR"' EXIT
{
echo "# $SPEC, $(nproc) CPUs visible (the CLI takes its cgroup share: --threads 0), $(df -h /tmp | tail -1 | awk '{print $4}') free in /tmp"
t0=$(date +%s.%N)
"$CLI" --matrix "$SPEC" --write-mtx "$DIR/m.mtx" || exit 1
t1=$(date +%s.%N)
ls -l "$DIR/m.mtx" | awk '{printf "file: %.1f MB\n", $5 / 1e6}'
echo "generated + written in $(python3 -c "print(round($t1 - $t0, 2))") s"
t2=$(date +%s.%N)
"$CLI" --threads 0 --csr "$DIR/m.mtx" --device hip --profile 10 --x uniform --check > "$DIR/run.json" || { cat "$DIR/run.json" | tail -5; exit 1; }
t3=$(date +%s.%N)
echo "CLI from the file, whole process: $(python3 -c "print(round($t3 - $t2, 2))") s"
python3 - "$DIR/run.json" "$DIR/m.mtx" <<'PY'
import json, os, sys
d = json.load(open(sys.argv[1]))
k, dev = d["kernel"], d["kernel"]["device"]
size = os.path.getsize(sys.argv[2])
ini = dev.get("init_seconds", {})
print("rows %d  entries %d  lines/s %.1f M  MB/s %.0f (load + convert %.2f s)  upload + plan %.2f s" % (
    k["rows"], k["nonzeros"], k["nonzeros"] / max(ini.get("load_and_convert", 1e-9), 1e-9) / 1e6,
    size / max(ini.get("load_and_convert", 1e-9), 1e-9) / 1e6, ini.get("load_and_convert", -1), ini.get("upload_and_plan", -1)))
print("kernel: median %.1f us over %d runs (sync per run), device %.1f us; parity %s, max rel err %.3g" % (
    d["execution_time"]["median"] / 1e3, d["execution_time"]["samples"], dev["last_run_device_ns"] / 1e3,
    d["parity"]["pass"], d["parity"]["max_relative_error"]))
PY
} 2>&1 | tee "$OUT"
