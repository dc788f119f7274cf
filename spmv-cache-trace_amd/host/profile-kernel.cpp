#include "profile-kernel.hpp"

#include "util/sample.hpp"

#include <atomic>
#include <cerrno>
#include <exception>
#include <mutex>
#include <ostream>
#include <system_error>

#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __linux__
#include <sched.h>
#endif

Profiling::Profiling(TraceConfig const & trace_config, Kernel const & kernel, std::vector<ProfilingRun> runs)
    : trace_config_(trace_config), kernel_(kernel), runs_(std::move(runs))
{
    bool any_device_time = false;
    for (auto const & r : runs_) {
        execution_time_.push_back(r.execution_time);
        any_device_time = any_device_time || r.device_time != 0;
    }
    if (any_device_time)
        for (auto const & r : runs_)
            device_time_.push_back(r.device_time);
}

namespace {

// First exception of any thread, kept until the team has left the parallel region.
class FirstError
{
public:
    void capture()
    {
        std::lock_guard<std::mutex> lock(m_);
        if (!error_)
            error_ = std::current_exception();
        failed_.store(true, std::memory_order_release);
    }
    bool failed() const { return failed_.load(std::memory_order_acquire); }
    void rethrow()
    {
        if (error_)
            std::rethrow_exception(error_);
    }

private:
    std::mutex m_;
    std::exception_ptr error_;
    std::atomic<bool> failed_{false};
};

// Touch ten times the largest cache (src/profile-kernel.cpp:181-192); worksharing loops, so
// every thread of the team must call it.
void flush_cache(cache_size_type cache_size)
{
    static std::vector<double> scratch;
    static double sink = 0.0;
    long long const n = 10 * cache_size / (long long) sizeof(double);
#pragma omp single
    scratch.assign((std::size_t) n, 0.0);
#pragma omp for
    for (long long i = 0; i < n; ++i)
        scratch[(std::size_t) i] = 1.1;
    double sum = 0.0;
#pragma omp for nowait
    for (long long i = 0; i < n; ++i)
        sum += scratch[(std::size_t) i];
#pragma omp atomic
    sink += sum;
#pragma omp barrier
}

} // namespace

Profiling profile_kernel(TraceConfig const & trace_config, Kernel & kernel, bool warmup, bool flush_caches,
                         int runs, std::ostream & o, bool verbose)
{
    auto const & pins = trace_config.thread_affinities();
    int const num_threads = (int) pins.size();
    if (num_threads < 1)
        throw trace_config_error("Expected at least one thread in \"thread_affinities\"");
#ifdef _OPENMP
    omp_set_dynamic(0);
    omp_set_num_threads(num_threads);
#else
    if (num_threads > 1)
        throw trace_config_error("Multi-threaded profiling failed: Please re-build with OpenMP enabled");
#endif
    if (verbose)
        o << "Profiling " << kernel.name() << ": " << runs << " runs, " << num_threads << " threads\n";

    std::vector<ProfilingRun> results((std::size_t) (runs > 0 ? runs : 0));
    FirstError error;

    // Every thread executes the same sequence of barriers whatever happens: a failing step is
    // recorded, and the whole team leaves the loop together at the next check.
#pragma omp parallel num_threads(num_threads)
    {
#ifdef _OPENMP
        int const thread = omp_get_thread_num();
#else
        int const thread = 0;
#endif
        try {
#ifdef __linux__
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(pins[(std::size_t) thread].cpu, &set);
            if (sched_setaffinity(0, sizeof set, &set) < 0)
                throw std::system_error(errno, std::generic_category(), "sched_setaffinity");
#endif
        } catch (...) {
            error.capture();
        }
#pragma omp barrier
        if (!error.failed()) {
            try {
                kernel.prepare(trace_config);
            } catch (...) {
                error.capture();
            }
        }
#pragma omp barrier
        if (warmup && !error.failed()) {
            try {
                kernel.run(trace_config);
            } catch (...) {
                error.capture();
            }
        }
#pragma omp barrier
        for (int run = 0; run < runs; ++run) {
            if (error.failed())
                break; // uniform: the flag was last written before the previous barrier
            if (flush_caches) {
                flush_cache(trace_config.max_cache_size());
#pragma omp master
                {
                    try {
                        kernel.flush_caches(); // a device kernel's own caches (no-op for the CPU kernels)
                    } catch (...) {
                        error.capture();
                    }
                }
            }

            profiling_clock::time_point t0, t1;
#pragma omp barrier
#pragma omp master
            t0 = profiling_clock::now();
#pragma omp barrier
            try {
                kernel.run(trace_config);
            } catch (...) {
                error.capture();
            }
#pragma omp barrier
#pragma omp master
            {
                t1 = profiling_clock::now();
                results[(std::size_t) run].execution_time = (t1 - t0).count();
                results[(std::size_t) run].device_time = kernel.last_device_ns();
            }
#pragma omp barrier
        }
    }
    error.rethrow();
    return Profiling(trace_config, kernel, std::move(results));
}

std::ostream & operator<<(std::ostream & o, Profiling const & profiling)
{
    o << "{\n"
      << "\"trace_config\": " << profiling.trace_config() << ",\n"
      << "\"kernel\": " << profiling.kernel() << ",\n"
      << "\"execution_time\": ";
    print_sample(o, profiling.execution_time(), "ns");
    if (!profiling.device_time().empty()) {
        o << ",\n\"device_time\": ";
        print_sample(o, profiling.device_time(), "ns");
    }
    o << ",\n\"profiling_events\": []";
    // additive: rates derived from the median wall time and, for GPU kernels, the median device time
    double const flops = profiling.kernel().flops_per_run(), bytes = profiling.kernel().bytes_per_run();
    if (flops > 0.0 && !profiling.execution_time().empty()) {
        auto const wall = sample_stats(profiling.execution_time());
        o << ",\n\"throughput\": {\"flops_per_run\": " << flops << ", \"algorithmic_bytes_per_run\": " << bytes
          << ", \"gflops_median\": " << flops / wall.median << ", \"gbs_median\": " << bytes / wall.median;
        if (!profiling.device_time().empty()) {
            auto const dev = sample_stats(profiling.device_time());
            o << ", \"device_gflops_median\": " << flops / dev.median << ", \"device_gbs_median\": " << bytes / dev.median
              << ", \"device_fraction_of_8TBs\": " << bytes / dev.median / 8000.0;
        }
        o << "}";
    }
    return o << profiling.extra() << "\n}";
}
