#!/bin/bash
# tools/ablate.sh build   -- here: libraries of the value-dictionary kernel with parts of its work removed
#                            (tools/ablate/ablate_N.so, N = bit mask: 1 no x traffic, 2 no y read, 4 no row sums,
#                            8 no y write); they compute wrong results by design and are never shipped
# tools/ablate.sh run     -- on the GPU box: bench.py on the headline workload with each of them
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
if [ "$1" = build ]; then
    mkdir -p tools/ablate
    for n in ${ABLATE_MASKS:-1 2 4 8 15}; do
        (cd spmv-cache-trace_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics \
            -I../include -DSPMV_HIP_EXPERIMENTS -DSPMV_VI_ABLATE=$n -shared csrc/*.hip -o ../tools/ablate/ablate_$n.so) &
    done
    wait
else
    for n in 0 ${ABLATE_MASKS:-1 2 4 8 15}; do
        if [ $n = 0 ]; then export SPMV_HIP_EXPERIMENTS=0; else export SPMV_HIP_EXPERIMENTS=tools/ablate/ablate_$n.so; fi
        python3 bench.py --steps 100 --warmup 10 --no-reference-protocol --no-cpu-baseline "${@:2}" > gpurun_out/ablate_$n.log 2> gpurun_out/ablate_$n.err
        echo "ablate mask $n: $(python3 tools/jline.py gpurun_out/ablate_$n.log roofline.kernel_us)"
    done
fi
