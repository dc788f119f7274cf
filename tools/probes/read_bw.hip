// How fast can this chip READ?  The STREAM triad (2 reads + 1 write) is the repo's empirical bandwidth ceiling;
// an SpMV launch is almost all reads (12 of every 12.x bytes), so this probe times pure read streams of the
// shapes the wave-tile kernel issues: 16-byte loads, a wave per contiguous 4 KB / 8 KB piece, 1..8 loads in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_bw.hip -o tools/probes/read_bw && tools/probes/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// every wave reads UNROLL consecutive 1 KB lines-of-64-lanes per step; workgroups are dealt pieces round-robin
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const v2d * __restrict__ a, long long n16, double * out)
{
    const long long per_block = 256ll * UNROLL;
    double s = 0.0;
    for (long long base = (long long) blockIdx.x * per_block; base < n16; base += (long long) gridDim.x * per_block) {
        v2d v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long i = base + u * 256 + threadIdx.x;
            v[u] = i < n16 ? a[i] : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            s += v[u].x + v[u].y;
    }
    if (s == 1.2345e300)
        out[blockIdx.x] = s; // keeps the loads alive, never taken
}

// one piece per workgroup, no grid-stride loop: the shape of the SpMV launch (a fresh wave per tile)
template <int UNROLL>
__global__ __launch_bounds__(256) void read_once_kernel(const v2d * __restrict__ a, long long n16, double * out)
{
    const long long base = (long long) blockIdx.x * 256 * UNROLL;
    v2d v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        v[u] = i < n16 ? a[i] : v2d{0.0, 0.0};
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
        s += v[u].x + v[u].y;
    if (s == 1.2345e300)
        out[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void triad_kernel(v2d * __restrict__ a, const v2d * __restrict__ b, const v2d * __restrict__ c, long long n16)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        const v2d r = b[i] + 3.1 * c[i];
        __builtin_nontemporal_store(r, a + i);
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
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (double gb : {0.5, 1.5, 3.4}) {
        const long long n16 = (long long) (gb * 1e9 / 16);
        v2d *a, *b, *c;
        double * out;
        CHECK(hipMalloc((void **) &a, n16 * 16));
        CHECK(hipMalloc((void **) &b, n16 * 16));
        CHECK(hipMalloc((void **) &c, n16 * 16));
        CHECK(hipMalloc((void **) &out, 1 << 20));
        CHECK(hipMemset(a, 0, n16 * 16));
        CHECK(hipMemset(b, 0, n16 * 16));
        CHECK(hipMemset(c, 0, n16 * 16));
        const double bytes = (double) n16 * 16;
        {
            const unsigned grid = (unsigned) ((n16 + 255) / 256);
            const double us = time_us([&] { hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(256), 0, 0, a, b, c, n16); }, 20);
            std::printf("%.1f GB arrays  triad (3 arrays)          %9.1f us  %7.1f GB/s\n", gb, us, 3 * bytes / us / 1e3);
        }
#define ONCE(U) { const unsigned grid = (unsigned) ((n16 + 256 * U - 1) / (256 * U)); \
            const double us = time_us([&] { hipLaunchKernelGGL(read_once_kernel<U>, dim3(grid), dim3(256), 0, 0, a, n16, out); }, 20); \
            std::printf("%.1f GB read once, %d x 16 B per lane, %8u workgroups  %9.1f us  %7.1f GB/s\n", gb, U, grid, us, bytes / us / 1e3); }
        ONCE(1) ONCE(2) ONCE(4) ONCE(8)
#define LOOP(U, PER_CU) { const unsigned grid = (unsigned) (cus * PER_CU); \
            const double us = time_us([&] { hipLaunchKernelGGL(read_kernel<U>, dim3(grid), dim3(256), 0, 0, a, n16, out); }, 20); \
            std::printf("%.1f GB read loop, %d x 16 B per lane, %2d workgroups per CU   %9.1f us  %7.1f GB/s\n", gb, U, PER_CU, us, bytes / us / 1e3); }
        LOOP(2, 8) LOOP(4, 8) LOOP(8, 8) LOOP(4, 4) LOOP(8, 4) LOOP(4, 2)
        CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(c)); CHECK(hipFree(out));
    }
    return 0;
}
