#!/bin/bash
# tools/gpu.sh [--timeout S] -- '<command>'   (build container only)
# gpurun with the commit stamped first: the GPU box gets a snapshot without .git, and every bench line /
# profile summary wants to say which commit it was taken from (spmv_amd/buildinfo.py).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT/spmv-cache-trace_amd/python")
from spmv_amd import buildinfo
print("stamped", buildinfo.write_head_stamp(), "sources", buildinfo.source_sha256()[:16])
PY
exec /usr/local/graft/bin/gpurun "$@"
