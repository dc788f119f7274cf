# tools/ab_misc.sh [TAG] -- on the GPU box: two library BUILDS side by side in alternating processes (tools/ab.py compares plans of
# one build): libspmv_hip_experiments.so against spmv-cache-trace_amd/libspmv_hip_variant.so (built by hand from the same sources
# with one -D switch; git-ignored, travels with the snapshot).  MATRICES / FLAGS override the lists.
out=gpurun_out/${1:-r05_ab_variant}.log
: > $out
for m in ${MATRICES:-synthetic:kkt:200 synthetic:banded:4000000,13 synthetic:queen synthetic:poisson2d:4096}; do
for rep in 1 2 3; do
  for lib in libspmv_hip_experiments.so libspmv_hip_variant.so; do
    echo "-- $m $lib" >> $out
    SPMV_HIP_EXPERIMENTS=$PWD/spmv-cache-trace_amd/$lib python tools/ab.py --matrix $m base=${FLAGS:-0x100000} 2>&1 | grep -E "^base" | cut -c1-140 >> $out
  done
done
done
cat $out
