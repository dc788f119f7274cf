#!/bin/bash
# tools/gather_layout_ab.sh [TAG] -- on the GPU box (experiments build): the products of plain narrow / wide tiles in the quad layout
# (a lane owns four consecutive entries) against the lane-major layouts of round 6 (tile_products_lanes: neighbouring lanes take
# neighbouring entries, singly or in pairs), interleaved in one process per matrix, y compared bit for bit.
TAG=${1:-gather_layout_ab}
out=gpurun_out/$TAG.log
: > $out
V='quad=0x100000 lanes=0x100000;SPMV_HIP_GATHER_LAYOUT=1 pairs=0x100000;SPMV_HIP_GATHER_LAYOUT=2'
for m in synthetic:queen:160,120,100,3,0,0,1 delaunay:2000000,1,2 delaunay:1000000,2,3 synthetic:webbase synthetic:random:4000000,24,3 synthetic:queen:100,80,60,3,0,0,5 "synthetic:kkt:200,50"; do
  echo "== $m" >> $out
  timeout -k 10 600 python tools/ab.py --experiments --matrix "$m" $V 2>&1 | grep -E "^(quad|lanes|pairs|matrix)" >> $out
done
cat $out
