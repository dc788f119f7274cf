#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r04_packed_kkt_ab.log
: > $out
for rep in 1 2 3 4 5; do
  for which in prev new; do
    if [ $which = prev ]; then export SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/libspmv_hip_prev.so; else unset SPMV_HIP_EXPERIMENTS; fi
    for m in synthetic:kkt:200 synthetic:kkt:125; do
      echo -n "$which $m " >> $out
      timeout -k 10 200 python tools/ab.py --matrix $m --rounds 7 base=0x100000 2>&1 | grep -E "^base" | cut -c1-100 >> $out
    done
  done
done
cat $out
