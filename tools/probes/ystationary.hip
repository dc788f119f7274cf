// How fast can a matrix with uniformly scattered columns go if y never leaves the chip and x is gathered from the L2?
// y of 4 M rows is 32 MB = the LDS of 256 CUs at 128 KB each: workgroup g keeps rows [g R, (g+1) R) of y in its LDS for the
// whole launch and adds products into it with LDS atomics; the entries are stored panel-major (all workgroups' entries
// with columns in panel 0, then panel 1, ...), so that at any time the whole chip gathers from ONE slice of x that fits
// every XCD's 4 MB L2 (time-multiplexed panels instead of csr_wavetile_kernel<PANELS>'s one panel per XCD).  An entry
// is 14 bytes: column 4, value 8, row within the workgroup's block 2.  No row_ptr, no virtual rows, no global atomics.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/ystationary.hip -o tools/probes/ystationary && tools/probes/ystationary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned short v4h __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_kernel(long long n, long long chunk_len, int nwg, int rows_per_wg, long long width, int32_t * col, double * val, uint16_t * rowoff)
{
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long) gridDim.x * blockDim.x) {
        const long long chunk = k / chunk_len;
        const long long p = chunk / nwg;
        const uint64_t h = mix((uint64_t) k);
        col[k] = (int32_t) (p * width + (long long) (h % (uint64_t) width));
        rowoff[k] = (uint16_t) ((h >> 40) % (uint64_t) rows_per_wg);
        val[k] = 1.0 + (double) ((h >> 20) & 1023) / 1024.0;
    }
}

__global__ void reference_kernel(long long n, long long chunk_len, int nwg, int rows_per_wg, const int32_t * col, const double * val, const uint16_t * rowoff,
                                 const double * x, double * y)
{
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long) gridDim.x * blockDim.x) {
        const long long g = (k / chunk_len) % nwg;
        unsafeAtomicAdd(y + g * rows_per_wg + rowoff[k], val[k] * x[col[k]]);
    }
}

// what scattered fp64 atomic adds into global memory cost (x-stationary alternative: x slice in LDS, y by global atomics):
// GATHER false: x is not read at all
template <bool GATHER>
__global__ __launch_bounds__(256) void atomics_kernel(long long n, long long chunk_len, int nwg, int rows_per_wg, const int32_t * col, const double * val,
                                                       const uint16_t * rowoff, const double * x, double * y)
{
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long) gridDim.x * blockDim.x) {
        // rows scattered over the whole of y: the workgroup block from the column hash instead of the entry's place
        const long long g = GATHER ? (k / chunk_len) % nwg : (long long) ((unsigned) col[k] * 2654435761u % (unsigned) nwg);
        unsafeAtomicAdd(y + g * rows_per_wg + rowoff[k], GATHER ? val[k] * x[col[k]] : val[k]);
    }
}

// MODE 0: as described; 1: no gather (x[lane]); 2: no LDS atomics (plain sum into a register); 3: streams only
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void ystat_kernel(int nwg, int npanels, int rows_per_wg, long long chunk_len, const int32_t * __restrict__ col,
                                                        const double * __restrict__ val, const uint16_t * __restrict__ rowoff,
                                                        const double * __restrict__ x, double * __restrict__ y)
{
    extern __shared__ double ylds[];
    for (int i = threadIdx.x; i < rows_per_wg; i += THREADS)
        ylds[i] = 0.0;
    __syncthreads();
    double sink = 0.0;
    for (int p = 0; p < npanels; ++p) {
        const long long base = ((long long) p * nwg + blockIdx.x) * chunk_len;
        const int32_t * cp = col + base;
        const double * vp = val + base;
        const uint16_t * rp = rowoff + base;
        for (long long o = (long long) threadIdx.x * 4; o < chunk_len; o += (long long) THREADS * 8) {
            const long long o2 = o + (long long) THREADS * 4; // chunk_len is a multiple of THREADS * 8
            const int4 c0 = *reinterpret_cast<const int4 *>(cp + o);
            const int4 c1 = *reinterpret_cast<const int4 *>(cp + o2);
            const v2d a0 = *reinterpret_cast<const v2d *>(vp + o), a1 = *reinterpret_cast<const v2d *>(vp + o + 2);
            const v2d a2 = *reinterpret_cast<const v2d *>(vp + o2), a3 = *reinterpret_cast<const v2d *>(vp + o2 + 2);
            const v4h r0 = *reinterpret_cast<const v4h *>(rp + o);
            const v4h r1 = *reinterpret_cast<const v4h *>(rp + o2);
            double x0, x1, x2, x3, x4, x5, x6, x7;
            if (MODE == 1 || MODE == 3) {
                x0 = x1 = x2 = x3 = x4 = x5 = x6 = x7 = (double) (c0.x ^ c0.y ^ c0.z ^ c0.w ^ c1.x ^ c1.y ^ c1.z ^ c1.w);
            } else {
                x0 = x[c0.x]; x1 = x[c0.y]; x2 = x[c0.z]; x3 = x[c0.w];
                x4 = x[c1.x]; x5 = x[c1.y]; x6 = x[c1.z]; x7 = x[c1.w];
            }
            if (MODE >= 2) {
                sink += a0.x * x0 + a0.y * x1 + a1.x * x2 + a1.y * x3 + a2.x * x4 + a2.y * x5 + a3.x * x6 + a3.y * x7
                        + (double) (r0.x + r0.y + r0.z + r0.w + r1.x + r1.y + r1.z + r1.w);
            } else {
                unsafeAtomicAdd(ylds + r0.x, a0.x * x0);
                unsafeAtomicAdd(ylds + r0.y, a0.y * x1);
                unsafeAtomicAdd(ylds + r0.z, a1.x * x2);
                unsafeAtomicAdd(ylds + r0.w, a1.y * x3);
                unsafeAtomicAdd(ylds + r1.x, a2.x * x4);
                unsafeAtomicAdd(ylds + r1.y, a2.y * x5);
                unsafeAtomicAdd(ylds + r1.z, a3.x * x6);
                unsafeAtomicAdd(ylds + r1.w, a3.y * x7);
            }
        }
    }
    __syncthreads();
    if (MODE >= 2)
        ylds[threadIdx.x] = sink;
    __syncthreads();
    double * yt = y + (long long) blockIdx.x * rows_per_wg;
    for (int i = threadIdx.x; i < rows_per_wg; i += THREADS)
        yt[i] += ylds[i];
}

template <int THREADS, int MODE>
static double run(const char * what, int nwg, int npanels, int rows_per_wg, long long chunk_len, const int32_t * col, const double * val,
                  const uint16_t * rowoff, const double * x, double * y, int reps)
{
    const size_t lds = (size_t) rows_per_wg * sizeof(double);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ystat_kernel<THREADS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((ystat_kernel<THREADS, MODE>), dim3(nwg), dim3(THREADS), lds, 0, nwg, npanels, rows_per_wg, chunk_len, col, val, rowoff, x, y);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((ystat_kernel<THREADS, MODE>), dim3(nwg), dim3(THREADS), lds, 0, nwg, npanels, rows_per_wg, chunk_len, col, val, rowoff, x, y);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double n = (double) chunk_len * nwg * npanels;
    std::printf("%-44s threads %4d  wg %5d x %5d rows  panels %2d  %9.1f us  %6.1f GFLOP/s  %5.2f TB/s of 14 B entries\n", what, THREADS, nwg, rows_per_wg,
                npanels, us, 2.0 * n / us / 1e3, 14.0 * n / us / 1e6);
    return us;
}

int main(int argc, char ** argv)
{
    const long long rows = 1LL << 22;
    const int per_row = 32; // (24 in bench.py's random workload: scale the times by 3/4)
    const long long n = rows * per_row;
    int32_t * col; double * val; uint16_t * rowoff; double * x; double * y; double * yref;
    CHECK(hipMalloc((void **) &col, n * 4 + 64));
    CHECK(hipMalloc((void **) &val, n * 8 + 64));
    CHECK(hipMalloc((void **) &rowoff, n * 2 + 64));
    CHECK(hipMalloc((void **) &x, rows * 8));
    CHECK(hipMalloc((void **) &y, rows * 8));
    CHECK(hipMalloc((void **) &yref, rows * 8));
    std::vector<double> hx((size_t) rows);
    for (long long i = 0; i < rows; ++i)
        hx[(size_t) i] = (double) (mix((uint64_t) i + 77) >> 11) / 9007199254740992.0 - 0.5;
    CHECK(hipMemcpy(x, hx.data(), rows * 8, hipMemcpyHostToDevice));
    const int reps = 10;
    const int panel_counts[] = {8, 16, 32, 64, 128};
    for (int variant = 0; variant < 5; ++variant) {
        const int rows_per_wg = 16384;
        const int npanels = panel_counts[variant];
        const int nwg = (int) (rows / rows_per_wg);
        const long long chunk_len = n / ((long long) nwg * npanels); // 16384 * 32 / 64 = 8192: a multiple of 1024 * 8
        if (chunk_len % 8192)
            continue;
        const long long width = rows / npanels;
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, n, chunk_len, nwg, rows_per_wg, width, col, val, rowoff);
        CHECK(hipMemset(y, 0, rows * 8));
        CHECK(hipMemset(yref, 0, rows * 8));
        hipLaunchKernelGGL(reference_kernel, dim3(4096), dim3(256), 0, 0, n, chunk_len, nwg, rows_per_wg, col, val, rowoff, x, yref);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ystat_kernel<1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, rows_per_wg * 8));
        hipLaunchKernelGGL((ystat_kernel<1024, 0>), dim3(nwg), dim3(1024), (size_t) rows_per_wg * 8, 0, nwg, npanels, rows_per_wg, chunk_len, col, val, rowoff, x, y);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        std::vector<double> hy((size_t) rows), hr((size_t) rows);
        CHECK(hipMemcpy(hy.data(), y, rows * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hr.data(), yref, rows * 8, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (long long i = 0; i < rows; ++i) {
            worst = std::fmax(worst, std::fabs(hy[(size_t) i] - hr[(size_t) i]));
            scale = std::fmax(scale, std::fabs(hr[(size_t) i]));
        }
        std::printf("%d panels of %lld KB of x: max |y - y_ref| / max |y_ref| = %.2e\n", npanels, width * 8 / 1024, worst / scale);
        if (variant == 0) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float ms;
            hipLaunchKernelGGL(atomics_kernel<false>, dim3(8192), dim3(256), 0, 0, n, chunk_len, nwg, rows_per_wg, col, val, rowoff, x, yref);
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 3; ++i)
                hipLaunchKernelGGL(atomics_kernel<false>, dim3(8192), dim3(256), 0, 0, n, chunk_len, nwg, rows_per_wg, col, val, rowoff, x, yref);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("global fp64 atomic adds, rows scattered over all of y, no x:        %9.1f us for %lld entries = %.1f G/s\n", ms * 1e3 / 3, n, n / (ms * 1e3 / 3) / 1e3);
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 3; ++i)
                hipLaunchKernelGGL(atomics_kernel<true>, dim3(8192), dim3(256), 0, 0, n, chunk_len, nwg, rows_per_wg, col, val, rowoff, x, yref);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("global gather + global atomic add, rows within one 16 K block at a time: %9.1f us = %.1f G/s\n", ms * 1e3 / 3, n / (ms * 1e3 / 3) / 1e3);
        }
        run<1024, 0>("y in LDS, x from L2, LDS atomics", nwg, npanels, rows_per_wg, chunk_len, col, val, rowoff, x, y, reps);
        run<1024, 3>("  streams only", nwg, npanels, rows_per_wg, chunk_len, col, val, rowoff, x, y, reps);
    }
    return 0;
}
