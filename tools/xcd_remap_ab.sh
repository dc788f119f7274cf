out=gpurun_out/r06_xcd_remap_ab.log
: > $out
for m in synthetic:queen:160,120,100,3,0,0,1 delaunay:2000000,1,2 delaunay:700000,3,1 synthetic:queen synthetic:queen:tril synthetic:kkt:200 synthetic:poisson2d:4096 synthetic:banded:4000000,13 synthetic:webbase; do
  echo "== $m" >> $out
  timeout -k 10 600 python tools/ab.py --matrix "$m" base=0x100000 xcd=0x100001 2>&1 | grep -E "^(base|xcd|matrix)" >> $out
done
cat $out
