cd ${GRAFT_REPO_ROOT:-.}
for b in 88 100 128 150 180 220 239; do
  python3 bench.py --matrix synthetic:banded:1000000,$b --format ell --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > gpurun_out/ell_tmp.log 2> gpurun_out/ell_tmp.err || { echo FAILED; tail -3 gpurun_out/ell_tmp.err; }
  python3 - $b <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ell_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
print("banded b=%s default: %.1f us frac %.3f L=%s tiles=%s" % (sys.argv[1], r["kernel_us"], r["frac"], d["config"].get("ell_row_length"), d["config"].get("tiles")))
PY
done
