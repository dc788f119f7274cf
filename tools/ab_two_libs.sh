#!/bin/bash
# A/B of two library builds on one box: prev (libspmv_hip_prev.so) and new, alternating processes
set -e
mkdir -p gpurun_out
out=gpurun_out/r04_window_alias_ab.log
: > $out
for m in synthetic:kkt:200 synthetic:banded:4000000,13 synthetic:banded:2000000,30; do
  for rep in 1 2; do
    echo "== $m prev (rep $rep)" >> $out
    SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/libspmv_hip_prev.so timeout -k 10 200 python tools/ab.py --matrix $m base=0x100000 2>&1 | grep -E "^base" >> $out
    echo "== $m new (rep $rep)" >> $out
    timeout -k 10 200 python tools/ab.py --matrix $m base=0x100000 2>&1 | grep -E "^base" >> $out
  done
done
cat $out
