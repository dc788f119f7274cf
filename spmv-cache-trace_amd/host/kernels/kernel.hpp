// kernel.hpp -- the operator / plug-in interface every benchmark kernel implements.
//
// Same contract as the reference's `Kernel` (src/kernels/kernel.hpp:18-45): the driver calls
// init() once (single-threaded: load the matrix, build x = 1 and y = 0), then from INSIDE an
// OpenMP parallel region prepare() once and run() once per timed repetition -- every thread of
// the team calls them (src/profile-kernel.cpp:227,262-264,160).  print() writes the "kernel"
// object of the result JSON.  Errors are kernel_error exceptions.
#pragma once

#include "../trace-config.hpp"

#include <cstdint>
#include <iosfwd>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

class kernel_error : public std::runtime_error
{
public:
    explicit kernel_error(std::string const & message) : std::runtime_error(message) {}
};

// (address, NUMA domain) pairs of one thread's memory accesses: the input of the reference's
// cache simulation (src/cache-simulation/replacement.hpp).  The simulation is not part of this
// engine; the type is kept so that the interface stays source-compatible.
typedef std::vector<std::pair<uintptr_t, int>> MemoryReferenceString;
// the reference spells the type replacement::MemoryReferenceString (src/cache-simulation/replacement.hpp:29):
// a Kernel subclass written against the reference's header compiles against this one unchanged
namespace replacement { using MemoryReferenceString = ::MemoryReferenceString; }

class Kernel
{
public:
    virtual ~Kernel() {}

    virtual void init(TraceConfig const & trace_config, std::ostream & o, bool verbose) = 0;
    virtual void prepare(TraceConfig const & trace_config) = 0;
    virtual void run(TraceConfig const & trace_config) = 0;
    virtual MemoryReferenceString memory_reference_string(TraceConfig const & trace_config, int thread,
                                                          int num_threads) const = 0;
    virtual std::string name() const = 0;
    virtual std::ostream & print(std::ostream & o) const = 0;

    // Additive hooks (not in the reference): device time of the last run, 0 if not measured,
    // and the result vector for parity checks (the reference never exposes y).
    virtual std::uint64_t last_device_ns() const { return 0; }
    // flops and algorithmic bytes of one run() (SURVEY 8d formulas), 0 if not defined
    virtual double flops_per_run() const { return 0.0; }
    virtual double bytes_per_run() const { return 0.0; }
    // number of entries of x (0 if the kernel has no such vector)
    virtual std::size_t columns() const { return 0; }
    // Not pure: a Kernel written against the reference's header (src/kernels/kernel.hpp:18-45)
    // compiles against this one unchanged; it simply cannot take part in --check.
    virtual std::vector<double> result() const { throw kernel_error(name() + ": result() is not implemented"); }
    // --flush-caches: what the kernel has to add to the harness's flush of the CPU caches (a device kernel evicts the
    // device's caches); called by one thread between two timed runs, never inside the timed window.
    virtual void flush_caches() {}
    // Replace x (default: all ones, src/kernels/csr-spmv.cpp:35) before prepare().
    virtual void set_x(std::vector<double> const &) { throw kernel_error(name() + ": set_x() is not implemented"); }
};

inline std::ostream & operator<<(std::ostream & o, Kernel const & kernel) { return kernel.print(o); }
