# tools/queen_twins.sh -- on the GPU box: the Queen_4147 stand-in and its less tidy twins: 2 % of the blocks
# with one or two entries dropped, nodes with one or two unknowns, links jittered by 6 nodes; each with masked block tiles
# (default) and without (SPMV_HIP_FLAG_NO_MASKED_BLOCKS 0x20000000), plus block tiles off altogether (0x2000000).
cd ${GRAFT_REPO_ROOT:-.}
SPECS=${SPECS:-"110,71,177 110,71,177,3,20 110,71,177,3,100 110,71,177,3,0,1000 110,71,177,3,20,1000 110,71,177,6 110,71,177,6,20"}
FLAGS=${FLAGS:-"0 0x20000000 0x2000000"}
for spec in $SPECS; do
  for flags in $FLAGS; do
  python3 bench.py --matrix synthetic:queen:$spec --flags $flags --steps 30 --warmup 5 --no-cpu-baseline --no-reference-protocol --no-cold > gpurun_out/queen_tmp.log 2> gpurun_out/queen_tmp.err || { echo FAILED; tail -3 gpurun_out/queen_tmp.err; }
  python3 - $spec $flags <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/queen_tmp.log") if l.startswith("{")][-1])
r = d["roofline"]
pm = r.get("plan_ms") or {}
print("queen %-24s flags %-10s %7.1f us  frac(8d) %.3f  streamed/triad %.3f  plan %.1f ms  parity %s  tiles=%s" % (
    sys.argv[1], sys.argv[2], r["kernel_us"], r["frac"], r["streamed"]["frac_of_triad"], pm.get("total", float("nan")),
    (d.get("parity") or {}).get("pass"), d["config"].get("tiles")))
PY
  done
done
