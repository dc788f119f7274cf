#!/bin/bash
# tools/sanitize_host.sh -- the C++ host side (loader, converters, CPU kernels, timed loop, CLI) built with
# AddressSanitizer + UndefinedBehaviorSanitizer and run over the CPU paths of the CLI (sanitizers are for the
# CPU build only: no GPU ASan on this pool).  Prints one line per command; any report is a failure.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/spmv-cache-trace_amd
OUT=${TMPDIR:-/tmp}/spmv_sanitize
mkdir -p "$OUT"
cd "$PKG" || exit 1
SRCS="host/util/json-value.cpp host/util/cpu-budget.cpp host/trace-config.cpp host/matrix/matrix-market.cpp host/matrix/matrix-cache.cpp \
 host/matrix/csr-matrix.cpp host/matrix/coo-matrix.cpp host/matrix/ell-matrix.cpp host/matrix/hybrid-matrix.cpp host/matrix/matrix-reorder.cpp \
 host/matrix/synthetic.cpp host/kernels/spmv-kernels.cpp host/kernels/triad-kernel.cpp host/profile-kernel.cpp host/host-api.cpp host/main.cpp"
g++ -std=c++17 -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -D__HIP_PLATFORM_AMD__ \
    -I"$ROOT/include" -I/opt/rocm/include -Ihost $SRCS -o "$OUT/cli" -L. -lspmv_hip -L/opt/rocm/lib -lamdhip64 -lz \
    -Wl,-rpath,"$PKG" -Wl,-rpath,/opt/rocm/lib || exit 1
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
G=$ROOT/tests/golden
bad=0
run() {
    "$OUT/cli" "$@" > "$OUT/out.txt" 2> "$OUT/err.txt"
    n=$(grep -c -E 'runtime error|AddressSanitizer' "$OUT/err.txt")
    echo "reports: $n   $*"
    [ "$n" = 0 ] || { bad=1; grep -E 'runtime error|AddressSanitizer' "$OUT/err.txt" | head -5; }
}
run --threads 2 --csr "$G/bus1138_like.mtx" --profile 2 --check
run --threads 3 --ell "$G/poisson2D.mtx" --profile 2 --check
run --threads 2 --coo "$G/test_mtx.gz" --profile 2
run --threads 2 --spmv-format hybrid --matrix "$G/poisson2D.mtx" --profile 2 --check
run --threads 1 --matrix synthetic:webbase:20000,62000,300,75 --spmv-format coo --profile 2 --check
run --matrix synthetic:kkt:10 --write-mtx "$OUT/k.mtx.gz"
run --threads 2 --matrix synthetic:queen:9,8,7,3,150,40 --spmv-format csr --profile 2 --check
run --threads 2 --matrix synthetic:queen:9,8,7,3,0,11 --spmv-format ell --profile 2 --check
run --threads 2 --csr "$OUT/k.mtx.gz" --profile 1 --expand-symmetric
run --threads 2 --csr "$G/bus1138_like.mtx__RCM" --profile 1
run --threads 1 --synthetic queen:5,4,6 --spmv-format csr --profile 1 --x uniform
run --threads 2 --csr "$G/kat.json" --profile 1
# the pessimistic twins and the specs whose products used to overflow before the friendly error
run --threads 1 --synthetic kkt:8,50 --spmv-format csr --profile 1
run --threads 1 --synthetic queen:5,4,6,6 --spmv-format csr --profile 1
run --threads 1 --synthetic poisson2d:20,1 --spmv-format csr --profile 1
run --synthetic poisson2d:9999999999 --spmv-format csr --profile 1
run --synthetic queen:99999,99999,99999 --spmv-format csr --profile 1
run --synthetic kkt:99999999999 --spmv-format csr --profile 1
exit $bad
