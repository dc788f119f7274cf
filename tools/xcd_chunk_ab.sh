#!/bin/bash
# tools/xcd_chunk_ab.sh [TAG] -- on the GPU box (experiments build): the default launch with its workgroups dealt to the XCDs in CHUNKS
# of 2^k consecutive workgroups (tile_common.hpp: xcd_chunk_remap) against launch order (k = 0), interleaved in one process per matrix.
TAG=${1:-xcd_chunk_ab}
out=gpurun_out/$TAG.log
: > $out
V='base=0x100000'
for k in 2 3 4 5 6 8; do V="$V c$k=0x100000;SPMV_HIP_XCD_CHUNK_LOG2=$k"; done
for m in ${MATRICES:-synthetic:queen:160,120,100,3,0,0,1 delaunay:2000000,1,2 delaunay:700000,3,1 synthetic:queen synthetic:queen:tril synthetic:kkt:200 synthetic:poisson2d:4096 synthetic:banded:4000000,13 synthetic:webbase}; do
  echo "== $m" >> $out
  timeout -k 10 600 python tools/ab.py --experiments --matrix "$m" $V 2>&1 | grep -E "^(base|c[0-9]|matrix)" | cut -c1-120 >> $out
done
cat $out
