// How long does the chip take just to START and retire the waves of the headline launch?  Poisson 4096^2 is 164 467 tiles =
// 41 117 workgroups of four waves with 19.6 KB of LDS each.  An empty kernel of that shape (every wave reads its 16-byte
// descriptor and leaves, or leaves at once) gives the floor under any one-tile-per-wave design; the same number of waves in
// bigger or smaller workgroups says what the workgroup size is worth to the dispatcher.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_rate.hip -o tools/probes/dispatch_rate && tools/probes/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int THREADS, int LDS_BYTES, int MODE> // MODE 0: leave at once; 1: read the wave's descriptor, store one word if it is odd (never)
__global__ __launch_bounds__(THREADS) void empty_kernel(const int4 * __restrict__ desc, int * out)
{
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
    if (LDS_BYTES > 0 && out == nullptr) // keeps the allocation alive
        lds[threadIdx.x] = 1;
    if (MODE == 1) {
        const int w = (int) blockIdx.x * (THREADS / 64) + (int) (threadIdx.x >> 6);
        const int4 d = desc[w];
        if (d.x & 1)
            out[w] = d.y + (LDS_BYTES > 0 ? lds[0] : 0);
    }
}

template <int THREADS, int LDS_BYTES, int MODE>
static void run(long long waves, const int4 * desc, int * out)
{
    const long long per_wg = THREADS / 64;
    const unsigned grid = (unsigned) ((waves + per_wg - 1) / per_wg);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((empty_kernel<THREADS, LDS_BYTES, MODE>), dim3(grid), dim3(THREADS), 0, 0, desc, out);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((empty_kernel<THREADS, LDS_BYTES, MODE>), dim3(grid), dim3(THREADS), 0, 0, desc, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%7lld waves as %6u workgroups of %4d threads, %5d B LDS, %s: %7.2f us per launch = %.2f waves per ns\n", waves, grid, THREADS, LDS_BYTES,
                MODE ? "descriptor read" : "leave at once  ", ms * 1e3 / reps, waves / (ms * 1e6 / reps));
}

int main()
{
    const long long waves = 164468;
    int4 * desc;
    int * out;
    const long long most = 700328 + 2048; // every run's waves (rounded up to whole workgroups) stay inside the arrays
    CHECK(hipMalloc((void **) &desc, most * sizeof(int4)));
    CHECK(hipMemset(desc, 0, most * sizeof(int4)));
    CHECK(hipMalloc((void **) &out, most * sizeof(int)));
    run<256, 0, 0>(waves, desc, out);
    run<256, 19584, 0>(waves, desc, out);
    run<256, 19584, 1>(waves, desc, out);
    run<64, 4896, 1>(waves, desc, out);
    run<128, 9792, 1>(waves, desc, out);
    run<512, 39168, 1>(waves, desc, out);
    run<1024, 65536, 1>(waves, desc, out);
    run<256, 19584, 1>(700328, desc, out); // the queen-like launch's wave count
    return 0;
}
