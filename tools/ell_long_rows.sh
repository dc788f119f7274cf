# tools/ell_long_rows.sh -- on the GPU box: ELLPACK bands of 141 ... 8191 entries per row through the context API: default flags
# (in place; rows of 161..2048 entries in multi-window tiles, longer rows a wave each in registers -- round 5),
# SPMV_HIP_FLAG_NO_MULTI_WINDOW (0x8000000: the column-major kernel for 161 ... 2048) and SPMV_HIP_FLAG_ELL_COLUMN_MAJOR (0x200).
# SPECS="rows,half_bandwidth ..." and FLAGS="0 0x200" override the lists.
cd ${GRAFT_REPO_ROOT:-.}
SPECS=${SPECS:-"1000000,70 1000000,88 1000000,100 1000000,128 1000000,150 1000000,180 1000000,220 1000000,239 1000000,255 400000,300 400000,400 400000,500 400000,750 400000,1000 200000,1100 100000,1024 100000,1500 50000,2000 50000,3000 25000,4095"}
FLAGS=${FLAGS:-"0 0x8000000"}
for spec in $SPECS; do
  for flags in $FLAGS; do
  python3 bench.py --matrix synthetic:banded:$spec --format ell --flags $flags --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > gpurun_out/ell_tmp.log 2> gpurun_out/ell_tmp.err || { echo FAILED; tail -3 gpurun_out/ell_tmp.err; }
  python3 - $spec $flags <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ell_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
print("banded %s flags %s: %.1f us frac %.3f L=%s tiles=%s" % (sys.argv[1], sys.argv[2], r["kernel_us"], r["frac"], d["config"].get("ell_row_length"), d["config"].get("tiles")))
PY
  done
done
