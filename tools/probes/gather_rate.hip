// tools/probes/gather_rate.hip -- how fast does the chip turn scattered 8-byte loads around, and does the cache policy of the load
// change it?  (uniformly scattered columns run at ~130-170 G gathers/s "however small the L2-resident slice";
// every such load moves a whole 128-byte line from the L2 into the vector L1.  The scope / non-temporal bits of a gfx950 load decide
// whether the line is kept in the L1 and how the L2 treats it -- do they also change what crosses between the two?)
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/probes/gather_rate.hip && ./gather_rate
//
// Every lane makes 8 independent loads per trip at hashed indices into a table of T doubles (T = 32 Ki ... 64 Mi: 256 KB, inside
// one L2; 4 MB; 32 MB, the size of x for 4 M columns; 512 MB, beyond the Infinity Cache), 2048 workgroups of 256 threads, 64 trips.
// Variants: plain load; non-temporal (nt); sc0; sc1; sc0 sc1 (system scope: not kept in the L1); nt sc0 sc1.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(call)                                                         \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

__device__ __forceinline__ unsigned mix(unsigned i)
{
    i *= 0x9E3779B1u;
    i ^= i >> 15;
    i *= 0x85EBCA6Bu;
    i ^= i >> 13;
    return i;
}

#define ASM_LOAD(bits)                                                                                                  \
    asm volatile("global_load_dwordx2 %0, %8, off " bits "\n\t"                                                         \
                 "global_load_dwordx2 %1, %9, off " bits "\n\t"                                                         \
                 "global_load_dwordx2 %2, %10, off " bits "\n\t"                                                        \
                 "global_load_dwordx2 %3, %11, off " bits "\n\t"                                                        \
                 "global_load_dwordx2 %4, %12, off " bits "\n\t"                                                        \
                 "global_load_dwordx2 %5, %13, off " bits "\n\t"                                                        \
                 "global_load_dwordx2 %6, %14, off " bits "\n\t"                                                        \
                 "global_load_dwordx2 %7, %15, off " bits "\n\t"                                                        \
                 "s_waitcnt vmcnt(0)"                                                                                   \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
                 : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])                 \
                 : "memory")

template <int VARIANT>
__global__ __launch_bounds__(256) void gather_kernel(const double * __restrict__ t, unsigned mask, int trips, double * __restrict__ out)
{
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    double acc = 0.0;
    for (int trip = 0; trip < trips; ++trip) {
        const double * q[8];
        double v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g)
            q[g] = t + (mix((tid * 64u + (unsigned) trip) * 8u + (unsigned) g) & mask);
        if (VARIANT == 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
                v[g] = *q[g];
        } else if (VARIANT == 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
                v[g] = __builtin_nontemporal_load(q[g]);
        } else if (VARIANT == 2) {
            ASM_LOAD("sc0");
        } else if (VARIANT == 3) {
            ASM_LOAD("sc1");
        } else if (VARIANT == 4) {
            ASM_LOAD("sc0 sc1");
        } else {
            ASM_LOAD("sc0 sc1 nt");
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
            acc += v[g];
    }
    if (acc == 12345.678)
        out[0] = acc; // never true: keeps the loads
}

// The shape of a multiply: the indices come from a stream (4 bytes per gather, read once), a tile of 512 entries per wave, each lane
// 8 of them (two 16-byte loads), then the 8 dependent gathers.  ONE tile per wave (the wave ends, the next wave starts: what the
// multiply's kernels do) against persistent waves that request the NEXT tile's indices before they wait for this tile's gathers.
template <bool PERSISTENT>
__global__ __launch_bounds__(256) void tile_kernel(const int4 * __restrict__ idx, int ntiles, const double * __restrict__ t, double * __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int stride = PERSISTENT ? (int) gridDim.x * 4 : ntiles;
    int w = (int) blockIdx.x * 4 + ((int) threadIdx.x >> 6);
    if (w >= ntiles)
        return;
    double acc = 0.0;
    int4 c0 = idx[(size_t) w * 128 + lane], c1 = idx[(size_t) w * 128 + 64 + lane];
    while (w < ntiles) {
        const int wn = w + stride;
        int4 n0 = c0, n1 = c1;
        if (PERSISTENT && wn < ntiles) {
            n0 = idx[(size_t) wn * 128 + lane];
            n1 = idx[(size_t) wn * 128 + 64 + lane];
        }
        const double v0 = t[c0.x], v1 = t[c0.y], v2 = t[c0.z], v3 = t[c0.w];
        const double v4 = t[c1.x], v5 = t[c1.y], v6 = t[c1.z], v7 = t[c1.w];
        acc += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
        c0 = n0;
        c1 = n1;
        w = wn;
    }
    if (acc == 12345.678)
        out[0] = acc;
}

// ... and what a column-panel multiply adds to that: the value stream (8 bytes per gather), and per tile 128 row sums that leave
// as fp64 atomics into y (Y = 1: what csr_wavetile_kernel<PANELS> does -- workgroup b works on panel b % 8, all panels sweep the
// rows in the same order, so eight XCDs add into the same lines of y at about the same time), as plain stores into a partial
// vector of the panel's own (Y = 2), or not at all (Y = 0).
template <int Y>
__global__ __launch_bounds__(256) void panel_kernel(const int4 * __restrict__ idx, const double * __restrict__ val, int ntiles,
                                                    const double * __restrict__ t, double * __restrict__ y, int rows, double * __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int w = (int) blockIdx.x * 4 + ((int) threadIdx.x >> 6);
    if (w >= ntiles)
        return;
    const int4 c0 = idx[(size_t) w * 128 + lane], c1 = idx[(size_t) w * 128 + 64 + lane];
    const double2 * a = reinterpret_cast<const double2 *>(val) + (size_t) w * 256;
    const double2 a0 = a[2 * lane], a1 = a[2 * lane + 1], a2 = a[128 + 2 * lane], a3 = a[128 + 2 * lane + 1];
    const double z0 = (a0.x * t[c0.x] + a0.y * t[c0.y]) + (a1.x * t[c0.z] + a1.y * t[c0.w]);
    const double z1 = (a2.x * t[c1.x] + a2.y * t[c1.y]) + (a3.x * t[c1.z] + a3.y * t[c1.w]);
    const int panel = (int) blockIdx.x & 7;
    const int r0 = (int) (((long long) (w >> 3) * 128) % (rows - 128));
    if (Y == 1) {
        unsafeAtomicAdd(y + r0 + lane, z0);
        unsafeAtomicAdd(y + r0 + 64 + lane, z1);
    } else if (Y == 2) {
        double * yp = y + (size_t) panel * rows;
        yp[r0 + lane] = z0;
        yp[r0 + 64 + lane] = z1;
    } else if (z0 + z1 == 12345.678) {
        out[0] = z0;
    }
}

template <int Y>
static double run_panel(const int * idx, const double * val, int ntiles, const double * t, double * y, int rows, double * out)
{
    const int blocks = (ntiles + 3) / 4;
    hipLaunchKernelGGL((panel_kernel<Y>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const int4 *>(idx), val, ntiles, t, y, rows, out);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL((panel_kernel<Y>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const int4 *>(idx), val, ntiles, t, y, rows, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 5.0 * ntiles * 512.0 / (ms * 1e-3) / 1e9;
}

__global__ __launch_bounds__(256) void fill_idx_kernel(int * __restrict__ idx, size_t n, unsigned mask)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        idx[i] = (int) (mix((unsigned) i) & mask);
}

template <bool PERSISTENT>
static double run_tiles(const int * idx, int ntiles, const double * t, double * out, int waves_per_simd)
{
    const int blocks = PERSISTENT ? 256 * waves_per_simd : (ntiles + 3) / 4;
    hipLaunchKernelGGL((tile_kernel<PERSISTENT>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const int4 *>(idx), ntiles, t, out);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL((tile_kernel<PERSISTENT>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const int4 *>(idx), ntiles, t, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 5.0 * ntiles * 512.0 / (ms * 1e-3) / 1e9;
}

template <int VARIANT>
static double run(const double * t, unsigned mask, double * out)
{
    const int blocks = 2048, trips = 64;
    hipLaunchKernelGGL((gather_kernel<VARIANT>), dim3(blocks), dim3(256), 0, 0, t, mask, trips, out);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL((gather_kernel<VARIANT>), dim3(blocks), dim3(256), 0, 0, t, mask, trips, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double gathers = 5.0 * blocks * 256.0 * trips * 8.0;
    return gathers / (ms * 1e-3) / 1e9;
}

int main()
{
    const size_t max_elems = 64u << 20;
    double * t = nullptr, * out = nullptr;
    CHECK(hipMalloc((void **) &t, max_elems * sizeof(double)));
    CHECK(hipMalloc((void **) &out, 64));
    CHECK(hipMemset(t, 0, max_elems * sizeof(double)));
    const char * names[6] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};
    std::printf("%-12s", "table");
    for (auto n : names)
        std::printf(" %12s", n);
    std::printf("   (G gathers/s: scattered 8-byte loads, 8 in flight per lane)\n");
    for (size_t elems : {(size_t) 32 << 10, (size_t) 512 << 10, (size_t) 4 << 20, (size_t) 64 << 20}) {
        const unsigned mask = (unsigned) (elems - 1);
        std::printf("%8.2f MB ", elems * 8.0 / 1e6);
        std::printf(" %12.1f", run<0>(t, mask, out));
        std::printf(" %12.1f", run<1>(t, mask, out));
        std::printf(" %12.1f", run<2>(t, mask, out));
        std::printf(" %12.1f", run<3>(t, mask, out));
        std::printf(" %12.1f", run<4>(t, mask, out));
        std::printf(" %12.1f", run<5>(t, mask, out));
        std::printf("\n");
        std::fflush(stdout);
    }
    // indices from a stream: 96 M gathers (the random matrix of north_star: 4 M rows x 24)
    const int ntiles = 187500;
    int * idx = nullptr;
    CHECK(hipMalloc((void **) &idx, (size_t) ntiles * 512 * sizeof(int)));
    std::printf("\nindices streamed (4 B per gather), 512 per wave-tile, %d tiles:\n%-12s %14s %14s %14s %14s\n", ntiles, "table", "tile per wave",
                "persistent x8", "persistent x4", "persistent x2");
    for (size_t elems : {(size_t) 64 << 10, (size_t) 512 << 10, (size_t) 4 << 20}) {
        const size_t n = (size_t) ntiles * 512;
        hipLaunchKernelGGL(fill_idx_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, idx, n, (unsigned) (elems - 1));
        CHECK(hipDeviceSynchronize());
        std::printf("%8.2f MB  %14.1f %14.1f %14.1f %14.1f\n", elems * 8.0 / 1e6, run_tiles<false>(idx, ntiles, t, out, 0),
                    run_tiles<true>(idx, ntiles, t, out, 8), run_tiles<true>(idx, ntiles, t, out, 4), run_tiles<true>(idx, ntiles, t, out, 2));
        std::fflush(stdout);
    }
    // the same with the values, and with 128 row sums per tile leaving as atomics / as plain stores into per-panel partial vectors
    {
        const int rows = 4000000;
        double * val = nullptr, * y = nullptr;
        CHECK(hipMalloc((void **) &val, (size_t) ntiles * 512 * sizeof(double)));
        CHECK(hipMalloc((void **) &y, (size_t) 8 * rows * sizeof(double)));
        CHECK(hipMemset(val, 0, (size_t) ntiles * 512 * sizeof(double)));
        CHECK(hipMemset(y, 0, (size_t) 8 * rows * sizeof(double)));
        std::printf("\n... + values (8 B per gather), 128 row sums per tile:\n%-12s %14s %14s %14s\n", "table", "no y", "atomics into y", "stores, 8 partial y");
        for (size_t elems : {(size_t) 64 << 10, (size_t) 512 << 10, (size_t) 4 << 20}) {
            const size_t n = (size_t) ntiles * 512;
            hipLaunchKernelGGL(fill_idx_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, idx, n, (unsigned) (elems - 1));
            CHECK(hipDeviceSynchronize());
            std::printf("%8.2f MB  %14.1f %14.1f %14.1f\n", elems * 8.0 / 1e6, run_panel<0>(idx, val, ntiles, t, y, rows, out),
                        run_panel<1>(idx, val, ntiles, t, y, rows, out), run_panel<2>(idx, val, ntiles, t, y, rows, out));
            std::fflush(stdout);
        }
        CHECK(hipFree(val));
        CHECK(hipFree(y));
    }
    CHECK(hipFree(idx));
    CHECK(hipFree(t));
    CHECK(hipFree(out));
    return 0;
}
