// tools/probes/gather_fetch.hip -- what does ONE scattered 8-byte load cost the fabric?  (MI355X_MICROARCH.md: FETCH_SIZE is
// calibrated for wide coalesced reads only -- "calibrate on a known byte count in your own access pattern".)
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_fetch tools/probes/gather_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./gather_fetch
//
// Three kernels over a 2 GiB table (8 x the Infinity Cache), each reading n = 4 Mi doubles, every one in a 128-byte line of its
// own that no other load of the launch touches:
//   gather_once    lane i reads element 16 * perm(i)            (scattered lines, 8 useful bytes per line)
//   gather_pairs   lane i reads elements 16 * perm(i) and + 8   (both 64-byte halves of its line)
//   stream         lane i reads 16 consecutive bytes             (the calibrated case: FETCH_SIZE x 2 = bytes)
// FETCH_SIZE (KiB) per launch x 1024 / n = bytes the counter tallies per scattered load; compared with `stream` that tells whether a
// scattered 8-byte gather moves 128, 64 or 32 bytes across the fabric -- and how to read the PMC traffic of the web-graph launch
// (DESIGN.md section 3.3).  The program prints the timings; the counters come from rocprofv3.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                         \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

// a bijection of [0, 2^24): lines of the table in a scrambled order (multiplication by an odd number modulo 2^24, xor-shifted)
__device__ __forceinline__ unsigned scramble(unsigned i)
{
    i = (i * 0x9E3779B1u) & 0xFFFFFFu;
    i ^= i >> 11;
    i = (i * 0x85EBCA6Bu) & 0xFFFFFFu;
    return i;
}

__global__ __launch_bounds__(256) void gather_once(long long n, const double * __restrict__ t, double * __restrict__ out)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = t[16ull * scramble((unsigned) i)];
}

__global__ __launch_bounds__(256) void gather_pairs(long long n, const double * __restrict__ t, double * __restrict__ out)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const unsigned long long e = 16ull * scramble((unsigned) i);
        out[i] = t[e] + t[e + 8];
    }
}

__global__ __launch_bounds__(256) void stream(long long n2, const double2 * __restrict__ t, double * __restrict__ out)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i < n2) {
        const double2 v = t[i];
        if (v.x == 12345.678)
            out[0] = v.y; // never true: keeps the load
    }
}

int main()
{
    const long long lines = 1ll << 24, n = 1ll << 22; // 16 Mi lines of 128 B = 2 GiB; 4 Mi of them read
    double * t = nullptr, * out = nullptr;
    CHECK(hipMalloc((void **) &t, (size_t) lines * 128));
    CHECK(hipMalloc((void **) &out, (size_t) n * 8));
    CHECK(hipMemset(t, 0, (size_t) lines * 128));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned) (n / 256);
    for (int k = 0; k < 3; ++k) {
        float ms[3];
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_once, dim3(grid), dim3(256), 0, 0, n, t, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[0], e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_pairs, dim3(grid), dim3(256), 0, 0, n, t, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[1], e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream, dim3((unsigned) (n * 8 / 256)), dim3(256), 0, 0, n * 8, (const double2 *) t, out); // 512 MiB of 16-byte loads
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[2], e0, e1));
        std::printf("round %d: gather_once %.1f us (%.1f G loads/s), gather_pairs %.1f us, stream 512 MiB %.1f us (%.2f TB/s)\n", k, ms[0] * 1e3,
                    n / (ms[0] * 1e-3) / 1e9, ms[1] * 1e3, ms[2] * 1e3, 536870912.0 / (ms[2] * 1e-3) / 1e12);
    }
    std::printf("expected if a scattered 8-byte load fetches its whole 128-byte line: %lld KiB per gather_once launch (64-byte half: %lld KiB)\n",
                n * 128 / 1024, n * 64 / 1024);
    return 0;
}
