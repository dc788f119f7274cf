#!/bin/bash
# on the GPU box: previous build (spmv-cache-trace_amd/libspmv_hip_prev.so) against the new one on the dictionary launches:
# Poisson as CSR (2 values), as ELLPACK (3 values: the padding's 0.0) and a 4-value graph-like matrix
mkdir -p gpurun_out
out=gpurun_out/${1:-ab_dictionary}.log
: > $out
for rep in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/libspmv_hip_prev.so; else unset SPMV_HIP_EXPERIMENTS; fi
    for args in "--headline product" "--format ell" "--format coo"; do
      python3 bench.py $args --steps 50 --warmup 10 --no-cpu-baseline --no-reference-protocol --no-companions --no-config3 --no-cold --no-host-boundary > gpurun_out/ab_tmp.log 2> gpurun_out/ab_tmp.err || tail -3 gpurun_out/ab_tmp.err
      python3 - "$which" "$args" >> $out <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_tmp.log") if l.startswith("{")][-1])
print("%-5s %-20s %.2f us" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_us"]))
PY
    done
  done
done
cat $out
