#!/bin/bash
# tools/ell_row_lengths.sh TAG -- on the GPU box: ELLPACK (and the queen-like hybrid) through the context API over a range
# of row lengths, default flags against SPMV_HIP_FLAG_EXACT_ORDER (0x2: one lane per row) and
# SPMV_HIP_FLAG_ELL_COLUMN_MAJOR (0x200).  One line per case: kernel us / fraction of 8 TB/s on 12 B per padded entry.
TAG=${1:-ell}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
one() { name=$1; fmt=$2; spec=$3; flags=$4
  python3 bench.py --matrix "$spec" --format $fmt --flags $flags --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol \
      > gpurun_out/${TAG}_tmp.log 2> gpurun_out/${TAG}_tmp.err || { echo "$name flags $flags FAILED"; tail -2 gpurun_out/${TAG}_tmp.err; return; }
  python3 - "$name" "$fmt" "$flags" gpurun_out/${TAG}_tmp.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
r = d["roofline"]
print("%-22s %-6s flags %-6s %9.1f us  algorithmic %.3f  streamed %.3f  L=%s tiles=%s" % (sys.argv[1], sys.argv[2], sys.argv[3], r["kernel_us"],
      r["frac_algorithmic"], r["frac"], d["config"].get("ell_row_length"), d["config"].get("tiles")))
PY
}
for b in 13 16 24 32 40 48 64 88 100 128 150 180 220 300; do
  for flags in 0 0x2 0x200; do one "banded b=$b" ell synthetic:banded:1000000,$b $flags; done
done
for flags in 0 0x2 0x200; do one "queen-like" ell synthetic:queen $flags; done
one "queen-like" hybrid synthetic:queen 0
for flags in 0 0x2; do one "kkt-like" ell synthetic:kkt:200 $flags; done
one "random k=32" ell synthetic:random:1000000,32 0
one "random k=32" ell synthetic:random:1000000,32 0x200
