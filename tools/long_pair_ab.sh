# tools/long_pair_ab.sh -- on the GPU box, one box for all lines: ELLPACK rows of more than 512 entries sharing a wave (in registers)
# against a wave each, and how large such a tile may be (SPMV_HIP_REGISTER_TILE_CAP_DIV: at most 1/DIV of the matrix), through the
# experiments library's plan-time switches.
cd ${GRAFT_REPO_ROOT:-.}
for spec in ${SPECS:-100000,1500 50000,2000 50000,3000 25000,4095 400000,1000}; do
  for div in ${DIVS:-32768 8192 16384 32768 8192 16384}; do
  SPMV_HIP_EXPERIMENTS=1 SPMV_HIP_LONG_PAIR_GAIN=${GAIN:--1} SPMV_HIP_REGISTER_TILE_CAP_DIV=$div python3 bench.py --matrix synthetic:banded:$spec --format ell --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol --no-cold --no-host-boundary > gpurun_out/ell_tmp.log 2> gpurun_out/ell_tmp.err || { echo FAILED; tail -3 gpurun_out/ell_tmp.err; }
  python3 - $spec $div <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ell_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
print("banded %s cap 1/%s: %.1f us frac %.3f L=%s tiles=%s" % (sys.argv[1], sys.argv[2], r["kernel_us"], r["frac"], d["config"].get("ell_row_length"), d["config"].get("tiles")))
PY
  done
done
