// tools/probes/xgmi_bw.hip -- what one GPU can STORE into its peers' memory over xGMI, the traffic pattern of the fused peer
// store (csr_wavetile_kernel<PEER>): so that the first line from a real multi-GPU node explains itself.
//
//   hipcc --offload-arch=gfx950 -O3 -o xgmi_bw tools/probes/xgmi_bw.hip && ./xgmi_bw [MiB per segment, default 16]
//
// Prints, for G visible devices:
//   (1) one writer, one reader: GB/s of a kernel on device i storing a segment into device j's memory, every ordered pair;
//   (2) the all-gather pattern: EVERY device stores its segment into ALL others at the same time (each link carries one
//       segment per direction), max over devices, and what that means for y of nlpkkt200 (16.24 M rows: 16.24 MB segments at
//       G = 8) and of the Poisson 4096^2 headline (16.8 M rows);
//   (3) with one device: the same kernel storing into local memory (the ceiling a link can never exceed).
// Stand-alone: no library, no torch.  Every kernel is a grid-stride copy with an exit every lane reaches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(call)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                  \
            std::exit(1);                                                                    \
        }                                                                                    \
    } while (0)

constexpr int kMaxPeers = 7;
struct Targets {
    double * dst[kMaxPeers];
    int n;
};

// every element of src goes to all targets (16 bytes per lane and store, like the multiply's y stores when two rows share a lane)
__global__ __launch_bounds__(256) void push_kernel(long long n2, const double2 * __restrict__ src, Targets t)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const double2 v = src[i];
#pragma unroll
        for (int k = 0; k < kMaxPeers; ++k)
            if (k < t.n)
                reinterpret_cast<double2 *>(t.dst[k])[i] = v;
    }
}

int main(int argc, char ** argv)
{
    const long long mib = argc > 1 ? std::max(1, std::atoi(argv[1])) : 16;
    const long long n = mib * 1024 * 1024 / 8; // doubles per segment
    int G = 0;
    CHECK(hipGetDeviceCount(&G));
    if (G < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 1;
    }
    G = std::min(G, 8);
    std::vector<double *> seg((size_t) G), full((size_t) G);
    std::vector<hipStream_t> stream((size_t) G);
    std::vector<hipEvent_t> e0((size_t) G), e1((size_t) G);
    for (int g = 0; g < G; ++g) {
        CHECK(hipSetDevice(g));
        for (int h = 0; h < G; ++h)
            if (h != g) {
                int can = 0;
                CHECK(hipDeviceCanAccessPeer(&can, g, h));
                if (!can) {
                    std::fprintf(stderr, "device %d cannot access device %d: no peer stores on this node\n", g, h);
                    return 1;
                }
                hipError_t e = hipDeviceEnablePeerAccess(h, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    CHECK(e);
                (void) hipGetLastError();
            }
        CHECK(hipMalloc((void **) &seg[(size_t) g], (size_t) n * 8));
        CHECK(hipMalloc((void **) &full[(size_t) g], (size_t) n * 8 * (size_t) G));
        CHECK(hipMemset(seg[(size_t) g], 0, (size_t) n * 8));
        CHECK(hipMemset(full[(size_t) g], 0, (size_t) n * 8 * (size_t) G));
        CHECK(hipStreamCreate(&stream[(size_t) g]));
        CHECK(hipEventCreate(&e0[(size_t) g]));
        CHECK(hipEventCreate(&e1[(size_t) g]));
        CHECK(hipDeviceSynchronize());
    }
    const int reps = 20;
    auto time_one = [&](int g, Targets t) { // ms per launch of device g storing its segment into the targets
        CHECK(hipSetDevice(g));
        const unsigned grid = 256 * 8;
        for (int w = 0; w < 3; ++w)
            hipLaunchKernelGGL(push_kernel, dim3(grid), dim3(256), 0, stream[(size_t) g], n / 2, (const double2 *) seg[(size_t) g], t);
        CHECK(hipEventRecord(e0[(size_t) g], stream[(size_t) g]));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(push_kernel, dim3(grid), dim3(256), 0, stream[(size_t) g], n / 2, (const double2 *) seg[(size_t) g], t);
        CHECK(hipEventRecord(e1[(size_t) g], stream[(size_t) g]));
        CHECK(hipEventSynchronize(e1[(size_t) g]));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0[(size_t) g], e1[(size_t) g]));
        return (double) ms / reps;
    };
    std::printf("segment: %lld MiB (%lld doubles), %d device(s)\n", mib, n, G);
    {
        Targets t{};
        t.n = 1;
        t.dst[0] = full[0];
        const double ms = time_one(0, t);
        std::printf("(3) device 0 -> its own memory: %.1f GB/s stored (%.3f ms)\n", (double) n * 8 / ms / 1e6, ms);
    }
    if (G == 1) {
        std::printf("one device visible: nothing crosses a link here; run on a multi-GPU node for (1) and (2)\n");
        return 0;
    }
    std::printf("(1) one writer, one target: GB/s stored, row = writer, column = target\n      ");
    for (int h = 0; h < G; ++h)
        std::printf("%8d", h);
    std::printf("\n");
    for (int g = 0; g < G; ++g) {
        std::printf("  %2d  ", g);
        for (int h = 0; h < G; ++h) {
            if (h == g) {
                std::printf("%8s", "-");
                continue;
            }
            Targets t{};
            t.n = 1;
            t.dst[0] = full[(size_t) h] + (size_t) g * (size_t) n;
            std::printf("%8.1f", (double) n * 8 / time_one(g, t) / 1e6);
        }
        std::printf("\n");
    }
    // (2) everybody stores into everybody at once
    std::vector<Targets> all((size_t) G);
    for (int g = 0; g < G; ++g) {
        all[(size_t) g].n = 0;
        for (int h = 0; h < G; ++h)
            if (h != g && all[(size_t) g].n < kMaxPeers)
                all[(size_t) g].dst[all[(size_t) g].n++] = full[(size_t) h] + (size_t) g * (size_t) n;
    }
    const unsigned grid = 256 * 8;
    for (int round = 0; round < 2; ++round) { // the first round warms up
        for (int g = 0; g < G; ++g) {
            CHECK(hipSetDevice(g));
            CHECK(hipEventRecord(e0[(size_t) g], stream[(size_t) g]));
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(push_kernel, dim3(grid), dim3(256), 0, stream[(size_t) g], n / 2, (const double2 *) seg[(size_t) g], all[(size_t) g]);
            CHECK(hipEventRecord(e1[(size_t) g], stream[(size_t) g]));
        }
        double worst = 0.0;
        for (int g = 0; g < G; ++g) {
            CHECK(hipSetDevice(g));
            CHECK(hipEventSynchronize(e1[(size_t) g]));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0[(size_t) g], e1[(size_t) g]));
            worst = std::max(worst, (double) ms / reps);
        }
        if (round == 1) {
            const double per_link = (double) n * 8 / worst / 1e6; // GB/s each link carries in one direction
            std::printf("(2) all %d devices store their segment into the %d others at once: %.3f ms per exchange (slowest device), "
                        "%.1f GB/s per link and direction, %.1f GB/s received per device\n", G, G - 1, worst, per_link, per_link * (G - 1));
            for (double rows : {16240000.0, 16777216.0}) {
                const double seg_bytes = rows * 8 / G;
                std::printf("    y of %.2f M rows at G = %d: %.2f MB per segment -> %.0f us per exchange at that rate\n", rows / 1e6, G, seg_bytes / 1e6,
                            seg_bytes / (per_link * 1e9) * 1e6);
            }
        }
    }
    return 0;
}
