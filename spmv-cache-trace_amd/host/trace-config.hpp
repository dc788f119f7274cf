// trace-config.hpp -- the trace configuration: caches, NUMA domains, thread -> cpu pins.
//
// Kept from the reference as the CLI's required input (src/main.cpp:152-153): it supplies
// the size of the OpenMP team and the CPU each thread is pinned to in the timed loop, and it
// is echoed verbatim into the result JSON.  Accept set and echo format follow
// src/trace-config.cpp:198-243 (caches), :263-343 (thread affinities), :579-597 (echo).
#pragma once

#include <cstdint>
#include <iosfwd>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

class trace_config_error : public std::runtime_error
{
public:
    explicit trace_config_error(std::string const & message) : std::runtime_error(message) {}
};

typedef int64_t cache_size_type;

struct Cache
{
    std::string name;
    cache_size_type size = 0;
    cache_size_type line_size = 0;
    double bandwidth = 0.0; // 0 = not given (null)
    std::vector<double> bandwidth_per_numa_domain;
    std::string cache_miss_event; // empty = null
    std::string parent;           // empty = null (last-level cache)
};

struct EventGroup
{
    int pid = 0;
    int cpu = 0;
    std::vector<std::string> events;
};

struct ThreadAffinity
{
    int thread = 0;
    int cpu = 0;
    std::string cache;
    int numa_domain = 0;
    std::vector<EventGroup> event_groups;
};

class TraceConfig
{
public:
    TraceConfig() = default;
    // validates: size % line_size, parents exist, thread caches exist, NUMA domain range
    TraceConfig(std::string name, std::string description, int num_numa_domains,
                std::vector<double> bandwidth_per_numa_domain, std::map<std::string, Cache> caches,
                std::vector<ThreadAffinity> thread_affinities);

    std::string const & name() const { return name_; }
    std::string const & description() const { return description_; }
    int num_numa_domains() const { return num_numa_domains_; }
    std::vector<double> const & bandwidth_per_numa_domain() const { return bandwidth_per_numa_domain_; }
    std::map<std::string, Cache> const & caches() const { return caches_; }
    std::vector<ThreadAffinity> const & thread_affinities() const { return thread_affinities_; }
    cache_size_type max_cache_size() const;

private:
    std::string name_, description_;
    int num_numa_domains_ = 0;
    std::vector<double> bandwidth_per_numa_domain_;
    std::map<std::string, Cache> caches_;
    std::vector<ThreadAffinity> thread_affinities_;
};

TraceConfig parse_trace_config(std::string const & json_text);
TraceConfig read_trace_config(std::string const & path);
// A configuration for `threads` unpinned-by-topology threads (cpu = thread index): what the
// --csr/--coo/--ell shorthands use when no --trace-config is given.
TraceConfig default_trace_config(int threads);

std::ostream & operator<<(std::ostream & o, TraceConfig const & trace_config);
