// h2d_bw.hip -- what does the host boundary (spmv_hip_upload_*: pageable host arrays -> HBM) cost, and what would pinned staging buy?
//   hipcc -O2 --offload-arch=gfx950 -pthread tools/probes/h2d_bw.hip -o tools/probes/h2d_bw && tools/probes/h2d_bw [MiB]
// Prints GB/s for: hipMemcpy from pageable memory; hipHostRegister of the caller's array + copy (+ what registering costs);
// a pipeline of T threads, each copying chunks into its own two pinned buffers and sending them on its own stream.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double staged(char * dst, const char * src, size_t bytes, int T, size_t chunk, std::vector<void *> & pinned, bool to_device)
{
    const size_t items = (bytes + chunk - 1) / chunk;
    std::atomic<size_t> next{0};
    const double t0 = now();
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t] {
            CK(hipSetDevice(0));
            hipStream_t s;
            hipEvent_t ev[2];
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
            bool used[2] = {false, false};
            size_t pend_i[2] = {0, 0};
            int b = 0;
            for (size_t i = next.fetch_add(1); i < items; i = next.fetch_add(1)) {
                const size_t off = i * chunk, n = std::min(chunk, bytes - off);
                char * pb = (char *) pinned[(size_t) 2 * t + b];
                if (used[b]) {
                    CK(hipEventSynchronize(ev[b]));
                    if (!to_device) { const size_t o = pend_i[b] * chunk; std::memcpy(dst + o, pb, std::min(chunk, bytes - o)); }
                }
                if (to_device) {
                    std::memcpy(pb, src + off, n);
                    CK(hipMemcpyAsync(dst + off, pb, n, hipMemcpyHostToDevice, s));
                } else {
                    CK(hipMemcpyAsync(pb, src + off, n, hipMemcpyDeviceToHost, s));
                    pend_i[b] = i;
                }
                CK(hipEventRecord(ev[b], s));
                used[b] = true;
                b ^= 1;
            }
            CK(hipStreamSynchronize(s));
            if (!to_device)
                for (int k = 0; k < 2; ++k)
                    if (used[k]) { const size_t o = pend_i[k] * chunk; std::memcpy(dst + o, (char *) pinned[(size_t) 2 * t + k], std::min(chunk, bytes - o)); }
            CK(hipEventDestroy(ev[0]));
            CK(hipEventDestroy(ev[1]));
            CK(hipStreamDestroy(s));
        });
    for (auto & th : pool)
        th.join();
    return now() - t0;
}

int main(int argc, char ** argv)
{
    const size_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024;
    const size_t bytes = mib << 20;
    char * host = (char *) std::aligned_alloc(4096, bytes);
    for (size_t i = 0; i < bytes; i += 4096) host[i] = (char) i;  // touched: the pages exist
    std::memset(host, 1, bytes);
    char * back = (char *) std::aligned_alloc(4096, bytes);
    std::memset(back, 0, bytes);
    char * dev;
    CK(hipSetDevice(0));
    CK(hipMalloc((void **) &dev, bytes));
    CK(hipMemset(dev, 0, bytes));
    CK(hipDeviceSynchronize());
    std::printf("%zu MiB, %u hardware threads\n", mib, std::thread::hardware_concurrency());
    for (int r = 0; r < 3; ++r) {
        double t0 = now();
        CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        std::printf("pageable hipMemcpy H2D        %7.2f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    {
        double t0 = now();
        CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost));
        double t1 = now();
        std::printf("pageable hipMemcpy D2H        %7.2f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    for (int r = 0; r < 2; ++r) {
        double t0 = now();
        CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        double t2 = now();
        CK(hipHostUnregister(host));
        double t3 = now();
        std::printf("hipHostRegister %7.2f ms + copy %7.2f ms (%6.1f GB/s) + unregister %7.2f ms = %6.1f GB/s in all\n", (t1 - t0) * 1e3,
                    (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, bytes / (t3 - t0) / 1e9);
    }
    for (size_t chunk_mib : {1, 2, 4, 8}) {
        for (int T : {2, 4, 8, 12}) {
            const size_t chunk = chunk_mib << 20;
            std::vector<void *> pinned((size_t) 2 * T);
            double a0 = now();
            for (auto & p : pinned)
                CK(hipHostMalloc(&p, chunk, hipHostMallocDefault));
            double a1 = now();
            double best = 1e9, bestd = 1e9;
            for (int r = 0; r < 3; ++r)
                best = std::min(best, staged(dev, host, bytes, T, chunk, pinned, true));
            for (int r = 0; r < 2; ++r)
                bestd = std::min(bestd, staged(back, dev, bytes, T, chunk, pinned, false));
            std::printf("staged T=%2d chunk=%zu MiB: pinned alloc %6.2f ms (%zu MiB); H2D %7.2f ms %6.1f GB/s; D2H %7.2f ms %6.1f GB/s\n", T, chunk_mib,
                        (a1 - a0) * 1e3, (2 * T * chunk) >> 20, best * 1e3, bytes / best / 1e9, bestd * 1e3, bytes / bestd / 1e9);
            for (auto & p : pinned)
                CK(hipHostFree(p));
        }
    }
    int ok = std::memcmp(host, back, bytes) == 0;
    std::printf("round trip %s\n", ok ? "intact" : "CORRUPT");
    return ok ? 0 : 1;
}
