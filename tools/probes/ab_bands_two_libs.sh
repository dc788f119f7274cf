#!/bin/bash
# on the GPU box: previous build (spmv-cache-trace_amd/libspmv_hip_prev.so) against the new one on bands of 27 ... 163 entries per row,
# Poisson and the kkt-like matrix (alternating processes)
mkdir -p gpurun_out
out=gpurun_out/${1:-ab_bands}.log
: > $out
for rep in 1 2; do
  for m in synthetic:banded:4000000,13 synthetic:banded:800000,32 synthetic:banded:400000,64 synthetic:banded:350000,70 synthetic:banded:310000,80 synthetic:banded:300000,81 synthetic:poisson2d:4096 synthetic:kkt:200; do
    for which in prev new; do
      if [ $which = prev ]; then export SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/libspmv_hip_prev.so; else unset SPMV_HIP_EXPERIMENTS; fi
      echo -n "$which $m " >> $out
      timeout -k 10 200 python3 tools/ab.py --matrix $m --rounds 5 base=0x100000 2>&1 | grep -E "^base" | cut -c1-110 >> $out
    done
  done
done
cat $out
