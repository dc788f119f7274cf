# tools/csr_long_rows.sh -- on the GPU box: CSR bands of 513 ... 8191 entries per row, default plan (rows of up to 2048 entries in
# multi-window tiles, longer rows a wave each in registers) against SPMV_HIP_FLAG_NO_MULTI_WINDOW (0x8000000: every row of more
# than 512 entries a wave -- or a few -- in registers).  SPECS / FLAGS override the lists.
cd ${GRAFT_REPO_ROOT:-.}
SPECS=${SPECS:-"400000,300 400000,400 400000,500 400000,750 200000,1000 100000,1024 100000,1500 50000,2000 25000,4095"}
FLAGS=${FLAGS:-"0 0x8000000"}
for spec in $SPECS; do
  for flags in $FLAGS; do
  python3 bench.py --matrix synthetic:banded:$spec --format csr --flags $flags --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > gpurun_out/csr_tmp.log 2> gpurun_out/csr_tmp.err || { echo FAILED; tail -3 gpurun_out/csr_tmp.err; }
  python3 - $spec $flags <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/csr_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
print("banded %s csr flags %s: %.1f us frac %.3f streamed/triad %.3f tiles=%s" % (sys.argv[1], sys.argv[2], r["kernel_us"], r["frac"], r["streamed"]["frac_of_triad"], d["config"].get("tiles")))
PY
  done
done
