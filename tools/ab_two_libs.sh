#!/bin/bash
# tools/ab_two_libs.sh [TAG] -- on the GPU box: A/B of two library BUILDS, alternating processes (tools/ab.py compares plans of
# one build).  Before gpurun: copy the build to compare against to spmv-cache-trace_amd/libspmv_hip_prev.so (git-ignored,
# travels with the snapshot), then build the new sources.  Log: gpurun_out/TAG.log
set -e
mkdir -p gpurun_out
out=gpurun_out/${1:-ab_two_libs}.log
[ -e spmv-cache-trace_amd/libspmv_hip_prev.so ] || { echo "no spmv-cache-trace_amd/libspmv_hip_prev.so"; exit 1; }
: > $out
for m in ${MATRICES:-synthetic:kkt:200 synthetic:banded:4000000,13 synthetic:banded:2000000,30}; do
  for rep in 1 2; do
    echo "== $m prev (rep $rep)" >> $out
    SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/libspmv_hip_prev.so timeout -k 10 300 python tools/ab.py --matrix $m base=0x100000 2>&1 | grep -E "^base" | cut -c1-150 >> $out
    echo "== $m new (rep $rep)" >> $out
    timeout -k 10 300 python tools/ab.py --matrix $m base=0x100000 2>&1 | grep -E "^base" | cut -c1-150 >> $out
  done
done
cat $out
