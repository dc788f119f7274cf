#!/bin/bash
# tools/pmc_probe.sh TAG "<kernel_sweep args>" -- diagnostic PMC passes (issue vs wait split,
# instruction mix, L2 hit rate) over tools/kernel_sweep.py.  Raw per-dispatch rows are averaged
# per kernel and counter into gpurun_out/TAG_pmc.txt.
set -o pipefail
TAG=${1:-pmc}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/kernel_sweep.py" --rounds 1 --reps 3 $* > "$OUT/pass$i.log" 2>&1 || { tail -5 "$OUT/pass$i.log"; }
  i=$((i+1))
done
python3 - "$OUT" > "$ROOT/gpurun_out/${TAG}_pmc.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-32s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
cat "$ROOT/gpurun_out/${TAG}_pmc.txt"
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
