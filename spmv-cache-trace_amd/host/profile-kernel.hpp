// profile-kernel.hpp -- the timed loop of `--profile=N` and its JSON report.
//
// Same protocol as the reference (src/profile-kernel.cpp:137-179, 197-313): one OpenMP team
// sized and pinned by the trace configuration; prepare(); one warm-up run; then N runs, each
// bracketed barrier / master t0 / barrier / run / barrier / master t1 with steady_clock, the
// durations in nanoseconds.  Hardware performance counters (libpfm4) are the reference's CPU
// cache study and are not collected: "profiling_events" is always [].
// Added per run: the kernel's own device time when it reports one (HIP event pair).
#pragma once

#include "kernels/kernel.hpp"

#include <chrono>
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

using profiling_clock = std::chrono::steady_clock;
using duration_type = profiling_clock::duration::rep;

struct ProfilingRun
{
    duration_type execution_time = 0; // wall ns around run(), measured on the master thread
    std::uint64_t device_time = 0;    // kernel-reported device ns (0 for CPU kernels)
};

class Profiling
{
public:
    Profiling(TraceConfig const & trace_config, Kernel const & kernel, std::vector<ProfilingRun> runs);

    TraceConfig const & trace_config() const { return trace_config_; }
    Kernel const & kernel() const { return kernel_; }
    std::vector<ProfilingRun> const & profiling_runs() const { return runs_; }
    std::vector<duration_type> const & execution_time() const { return execution_time_; }
    std::vector<std::uint64_t> const & device_time() const { return device_time_; }
    // extra members spliced in front of the report's closing brace (text starting with ",\n")
    void set_extra(std::string extra) { extra_ = std::move(extra); }
    std::string const & extra() const { return extra_; }

private:
    // references, as in the reference: both must outlive the report
    TraceConfig const & trace_config_;
    Kernel const & kernel_;
    std::vector<ProfilingRun> runs_;
    std::vector<duration_type> execution_time_;
    std::vector<std::uint64_t> device_time_;
    std::string extra_;
};

Profiling profile_kernel(TraceConfig const & trace_config, Kernel & kernel, bool warmup, bool flush_caches,
                         int runs, std::ostream & o, bool verbose);

std::ostream & operator<<(std::ostream & o, Profiling const & profiling);
