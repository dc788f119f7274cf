# tools/ell_long_rows.sh -- on the GPU box: ELLPACK bands of 177 ... 601 entries per row through the context API: default flags
# (in place, multi-window tiles), SPMV_HIP_FLAG_NO_MULTI_WINDOW (0x8000000: the column-major kernel for 161..512) and
# SPMV_HIP_FLAG_ELL_COLUMN_MAJOR (0x200)
cd ${GRAFT_REPO_ROOT:-.}
for b in 70 88 100 128 150 180 220 255 300; do
  for flags in 0 0x8000000 0x200; do
  python3 bench.py --matrix synthetic:banded:1000000,$b --format ell --flags $flags --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > gpurun_out/ell_tmp.log 2> gpurun_out/ell_tmp.err || { echo FAILED; tail -3 gpurun_out/ell_tmp.err; }
  python3 - $b $flags <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ell_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
print("banded b=%s flags %s: %.1f us frac %.3f L=%s tiles=%s" % (sys.argv[1], sys.argv[2], r["kernel_us"], r["frac"], d["config"].get("ell_row_length"), d["config"].get("tiles")))
PY
  done
done
