#!/bin/bash
# tools/row_length_sweep.sh [TAG] -- on the GPU box: band matrices of 3 ... 2049 entries per row (about 50 M entries each) through the
# default CSR plan (values read): one line per row length with the launch time and the SURVEY 8(d) fraction.  A hole in the plan's
# heuristics (a row length that falls between two tile classes) shows up as a line far below its neighbours.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
out=gpurun_out/${1:-row_length_sweep}.log
: > $out
for b in 1 2 3 4 6 8 9 12 13 16 20 24 28 32 40 48 64 80 81 85 96 100 128 150 160 192 255 256 300 384 450 512 600 768 900 1023 1024; do
  len=$((2 * b + 1))
  n=$((50000000 / len))
  [ $n -gt 8000000 ] && n=8000000
  echo -n "len $len rows $n " >> $out
  timeout -k 10 120 python3 tools/ab.py --matrix synthetic:banded:$n,$b --rounds 3 --reps 10 base=0x100000 2>&1 | grep -E "^base" | cut -c1-200 >> $out
done
cat $out
