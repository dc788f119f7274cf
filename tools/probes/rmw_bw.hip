// What does it cost to update a vector in place?  y[i] += c as the SpMV kernel does it (8 bytes per lane; a wave
// per 102 consecutive rows, i.e. 64 + 38 lanes, tiles not aligned to cache lines) against the same with aligned
// 128-row pieces, against out of place (z[i] = y[i] + c), with and without non-temporal accesses.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rmw_bw.hip -o tools/probes/rmw_bw && tools/probes/rmw_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int ROWS, bool NT, bool INPLACE>
__global__ __launch_bounds__(256) void update_kernel(const double * __restrict__ yin, double * yout, long long n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r0 = ((long long) blockIdx.x * 4 + wave) * ROWS;
    const double * src = INPLACE ? yout : yin;
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long r = r0 + lane + 64 * h;
        if (lane + 64 * h < ROWS && r < n)
            v[h] = NT ? __builtin_nontemporal_load(src + r) : src[r];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long r = r0 + lane + 64 * h;
        if (lane + 64 * h < ROWS && r < n) {
            if (NT)
                __builtin_nontemporal_store(v[h] + 1.0, yout + r);
            else
                yout[r] = v[h] + 1.0;
        }
    }
}

template <typename F>
double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        launch();
    double best = 1e30;
    for (int t = 0; t < 3; ++t) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i)
            launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms * 1e3 / reps < best ? ms * 1e3 / reps : best;
    }
    return best;
}

int main()
{
    for (long long n : {16777216ll, 134217728ll}) {
        double *y, *z;
        CHECK(hipMalloc((void **) &y, n * 8 + 1024));
        CHECK(hipMalloc((void **) &z, n * 8 + 1024));
        CHECK(hipMemset(y, 0, n * 8));
        CHECK(hipMemset(z, 0, n * 8));
#define RUN(ROWS, NT, INPLACE) { const unsigned grid = (unsigned) ((n + 4 * ROWS - 1) / (4 * ROWS)); \
        const double us = time_us([&] { hipLaunchKernelGGL((update_kernel<ROWS, NT, INPLACE>), dim3(grid), dim3(256), 0, 0, y, INPLACE ? y : z, n); }, 20); \
        std::printf("n = %10lld  %3d rows per wave  %-13s %-12s %9.1f us  %7.1f GB/s (read + write)\n", n, ROWS, NT ? "non-temporal" : "plain", INPLACE ? "in place" : "out of place", us, 16.0 * n / us / 1e3); }
        RUN(102, false, true) RUN(102, true, true) RUN(102, false, false) RUN(102, true, false)
        RUN(128, false, true) RUN(128, true, true) RUN(128, false, false) RUN(128, true, false)
        RUN(96, false, true) RUN(96, true, true) RUN(64, true, true) RUN(64, true, false)
        CHECK(hipFree(y));
        CHECK(hipFree(z));
    }
    return 0;
}
