// What does the LOAD MIX of a wave tile cost by itself?  One wave per 512-entry tile, four waves per workgroup, like
// csr_wavetile_kernel: 4 KB of values (two quads: 2 x 2 loads of 16 B per lane) and 1 KB of 16-bit column offsets per
// tile -- the latter as two loads of 8 B per lane (what the kernel issues) or as ONE load of 16 B per lane.  Nothing else:
// no x, no y, no LDS.  The gap between this and the real launch is what gathering, parking and summing cost.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_stream.hip -o tools/probes/tile_stream && tools/probes/tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int MODE> // 0: values only; 1: + columns as 2 x 8 B per lane; 2: + columns as 1 x 16 B per lane; 3: + columns as 32-bit (2 x 16 B)
                    // 4: like 1, but where the tile starts comes out of a 16-byte descriptor first (a dependent scalar load, as in the real kernel)
                    // 5: like 4, plus y read and written (8 rows per tile) and the products parked in LDS and read back once
__global__ __launch_bounds__(256, 8) void tile_stream_kernel(long long ntiles, const double * __restrict__ a, const uint16_t * __restrict__ j16,
                                                              const int32_t * __restrict__ j32, double * out, const int4 * __restrict__ desc = nullptr,
                                                              double * y = nullptr)
{
    __shared__ double prod_all[4][516];
    long long w = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= ntiles)
        return;
    const int lane = threadIdx.x & 63;
    if (MODE >= 4) {
        const int4 d0 = desc[w], d1 = desc[w + 1];
        w = __builtin_amdgcn_readfirstlane(d0.y) / 512 + (__builtin_amdgcn_readfirstlane(d1.y) & 0); // = w, but only known once the loads are back
    }
    const double * at = a + w * 512;
    double yv = 0.0;
    if (MODE == 5)
        yv = y[w * 8 + (lane >> 3)];
    v2d va[2], vb[2];
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = 256 * q + 4 * lane;
        va[q] = *reinterpret_cast<const v2d *>(at + o);
        vb[q] = *reinterpret_cast<const v2d *>(at + o + 2);
        if (MODE == 1 || MODE >= 4) {
            const v2u c = *reinterpret_cast<const v2u *>(j16 + w * 512 + o);
            acc += c.x ^ c.y;
        }
        if (MODE == 3) {
            const v4u c = *reinterpret_cast<const v4u *>(j32 + w * 512 + o);
            acc += c.x ^ c.y ^ c.z ^ c.w;
        }
    }
    if (MODE == 2) {
        const v4u c = *reinterpret_cast<const v4u *>(j16 + w * 512 + 8 * lane);
        acc += c.x ^ c.y ^ c.z ^ c.w;
    }
    double s = va[0].x + va[0].y + vb[0].x + vb[0].y + va[1].x + va[1].y + vb[1].x + vb[1].y + (double) acc;
    if (MODE == 5) {
        double * prod = prod_all[threadIdx.x >> 6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v2d * dst = reinterpret_cast<v2d *>(prod + 256 * q + 4 * lane);
            dst[0] = va[q];
            dst[1] = vb[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double z = 0.0;
        for (int t = 0; t < 8; ++t)
            z += prod[(lane >> 3) * 64 + (lane & 7) + 8 * t];
        if ((lane & 7) == 0)
            y[w * 8 + (lane >> 3)] = yv + z;
        s += z;
    }
    if (s == 1.2345e300)
        out[blockIdx.x] = s; // keeps the loads alive, never taken
}

template <typename F>
double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
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
    const long long ntiles = 644000; // 329.7 M entries: the queen-like matrix
    const long long n = ntiles * 512;
    double * a, * out;
    uint16_t * j16;
    int32_t * j32;
    CHECK(hipMalloc((void **) &a, n * 8));
    CHECK(hipMalloc((void **) &j16, n * 2));
    CHECK(hipMalloc((void **) &j32, n * 4));
    CHECK(hipMalloc((void **) &out, 1 << 22));
    CHECK(hipMemset(a, 0, n * 8));
    CHECK(hipMemset(j16, 0, n * 2));
    CHECK(hipMemset(j32, 0, n * 4));
    const unsigned grid = (unsigned) ((ntiles + 3) / 4);
    const char * what[4] = {"values only (8 B/entry)", "values + 16-bit columns, 2 x 8 B per lane (10 B/entry)", "values + 16-bit columns, 1 x 16 B per lane (10 B/entry)",
                            "values + 32-bit columns, 2 x 16 B per lane (12 B/entry)"};
    const double bytes[4] = {8.0 * n, 10.0 * n, 10.0 * n, 12.0 * n};
    double us[4];
    us[0] = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<0>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out); }, 10);
    us[1] = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<1>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out); }, 10);
    us[2] = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<2>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out); }, 10);
    us[3] = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<3>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out); }, 10);
    {
        int4 * desc;
        double * y;
        CHECK(hipMalloc((void **) &desc, (ntiles + 1) * sizeof(int4)));
        CHECK(hipMalloc((void **) &y, ntiles * 8 * sizeof(double)));
        CHECK(hipMemset(y, 0, ntiles * 8 * sizeof(double)));
        int4 * h = (int4 *) std::malloc((ntiles + 1) * sizeof(int4));
        for (long long t = 0; t <= ntiles; ++t)
            h[t] = make_int4((int) (t * 8), (int) (t * 512), 0, 0);
        CHECK(hipMemcpy(desc, h, (ntiles + 1) * sizeof(int4), hipMemcpyHostToDevice));
        const double u4 = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<4>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out, desc, y); }, 10);
        const double u5 = time_us([&] { hipLaunchKernelGGL(tile_stream_kernel<5>, dim3(grid), dim3(256), 0, 0, ntiles, a, j16, j32, out, desc, y); }, 10);
        std::printf("%-62s %8.1f us  %7.1f GB/s\n", "values + 16-bit columns behind a 16-byte descriptor load", u4, (10.0 * n + 16.0 * ntiles) / u4 / 1e3);
        std::printf("%-62s %8.1f us  %7.1f GB/s\n", "... + y read / written, products through LDS, 8-lane sums", u5, (10.0 * n + 16.0 * ntiles + 128.0 * ntiles) / u5 / 1e3);
    }
    for (int m = 0; m < 4; ++m)
        std::printf("%-62s %8.1f us  %7.1f GB/s\n", what[m], us[m], bytes[m] / us[m] / 1e3);
    return 0;
}
