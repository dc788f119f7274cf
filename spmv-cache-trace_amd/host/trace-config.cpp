#include "trace-config.hpp"

#include "util/json-value.hpp"

#include <cerrno>
#include <cstring>
#include <fstream>
#include <iterator>
#include <ostream>
#include <sstream>

TraceConfig::TraceConfig(std::string name, std::string description, int num_numa_domains,
                         std::vector<double> bandwidth_per_numa_domain,
                         std::map<std::string, Cache> caches,
                         std::vector<ThreadAffinity> thread_affinities)
    : name_(std::move(name))
    , description_(std::move(description))
    , num_numa_domains_(num_numa_domains)
    , bandwidth_per_numa_domain_(std::move(bandwidth_per_numa_domain))
    , caches_(std::move(caches))
    , thread_affinities_(std::move(thread_affinities))
{
    for (auto const & kv : caches_) {
        Cache const & c = kv.second;
        if (c.line_size == 0 || c.size % c.line_size != 0) {
            std::ostringstream s;
            s << c.name << ": Expected size (" << c.size << ") to be a multiple of line_size ("
              << c.line_size << ")";
            throw trace_config_error(s.str());
        }
        if (!c.parent.empty() && caches_.find(c.parent) == caches_.end()) {
            std::ostringstream s;
            s << kv.first << ": \"parent\": Expected a cache or numa domain, got \"" << c.parent << "\"";
            throw trace_config_error(s.str());
        }
    }
    for (std::size_t i = 0; i < thread_affinities_.size(); ++i) {
        ThreadAffinity const & t = thread_affinities_[i];
        if (caches_.find(t.cache) == caches_.end()) {
            std::ostringstream s;
            s << "\"thread_affinities\": " << i << ": Expected a first-level cache, got \"" << t.cache << "\"";
            throw trace_config_error(s.str());
        }
        if (t.numa_domain >= num_numa_domains_) {
            std::ostringstream s;
            s << "\"thread_affinities\": " << i << ": Expected a NUMA domain in the range [0,"
              << num_numa_domains_ << "), got \"" << t.numa_domain << "\"";
            throw trace_config_error(s.str());
        }
    }
}

cache_size_type TraceConfig::max_cache_size() const
{
    cache_size_type m = 0;
    for (auto const & kv : caches_)
        m = std::max(m, kv.second.size);
    return m;
}

namespace {

using json::Value;

// number-or-null array -> doubles (null -> empty)
std::vector<double> numbers_or_null(Value const & v)
{
    std::vector<double> out;
    if (v.is_null())
        return out;
    for (Value const & e : v.array) {
        if (!e.is_number())
            throw trace_config_error("Expected '\"bandwidth_per_numa_domain\": to be an array of numbers");
        out.push_back(e.number);
    }
    return out;
}

Cache cache_from_json(std::string const & name, Value const & v)
{
    // every key must be present; bandwidth / bandwidth_per_numa_domain / cache_miss_event /
    // parent may be null (src/trace-config.cpp:210-231)
    Value const * size = v.get("size");
    if (!size || !size->is_number())
        throw trace_config_error("Expected \"size\": (number)");
    Value const * line = v.get("line_size");
    if (!line || !line->is_number())
        throw trace_config_error("Expected \"line_size\": (number)");
    Value const * bw = v.get("bandwidth");
    if (!bw || !(bw->is_number() || bw->is_null()))
        throw trace_config_error("Expected \"bandwidth\": (number) or null");
    Value const * bwn = v.get("bandwidth_per_numa_domain");
    if (!bwn || !(bwn->is_array() || bwn->is_null()))
        throw trace_config_error("Expected \"bandwidth_per_numa_domain\": (array) or null");
    Value const * ev = v.get("cache_miss_event");
    if (!ev || !(ev->is_string() || ev->is_null()))
        throw trace_config_error("Expected \"cache_miss_event\": (string) or null");
    Value const * parent = v.get("parent");
    if (!parent || !(parent->is_string() || parent->is_null()))
        throw trace_config_error("Expected \"parent\": (string) or null");

    Cache c;
    c.name = name;
    c.size = size->to_int();
    c.line_size = line->to_int();
    c.bandwidth = bw->is_number() ? bw->number : 0.0;
    c.bandwidth_per_numa_domain = numbers_or_null(*bwn);
    c.cache_miss_event = ev->is_string() ? ev->string : "";
    c.parent = parent->is_string() ? parent->string : "";
    return c;
}

ThreadAffinity affinity_from_json(int thread, Value const & v)
{
    if (!v.is_object())
        throw trace_config_error("Expected '\"thread_affinities\": {\"cache\": ..., \"numa_domain\": ...}");
    Value const * cpu = v.get("cpu");
    if (!cpu || !cpu->is_number())
        throw trace_config_error("Expected \"cpu\": (number)");
    Value const * cache = v.get("cache");
    if (!cache || !cache->is_string())
        throw trace_config_error("Expected \"cache\": (string)");
    Value const * numa = v.get("numa_domain");
    if (!numa || !numa->is_number())
        throw trace_config_error("Expected \"numa_domain\": (number)");
    Value const * groups = v.get("event_groups");
    if (groups && !groups->is_array())
        throw trace_config_error("Expected \"event_groups\": (array)");

    ThreadAffinity t;
    t.thread = thread;
    t.cpu = (int) cpu->to_int();
    t.cache = cache->string;
    t.numa_domain = (int) numa->to_int();
    if (groups) {
        for (Value const & g : groups->array) {
            if (!g.is_object())
                throw trace_config_error("Expected '\"event_groups\": {\"pid\": ..., \"cpu\": ..., \"events\": ...}");
            Value const * pid = g.get("pid");
            if (!pid || !pid->is_number())
                throw trace_config_error("Expected \"pid\": (number)");
            Value const * gcpu = g.get("cpu");
            if (!gcpu || !gcpu->is_number())
                throw trace_config_error("Expected \"cpu\": (number)");
            Value const * events = g.get("events");
            if (!events || !events->is_array())
                throw trace_config_error("Expected \"events\": (array)");
            EventGroup eg;
            eg.pid = (int) pid->to_int();
            eg.cpu = (int) gcpu->to_int();
            for (Value const & e : events->array) {
                if (!e.is_string())
                    throw trace_config_error("Expected \"event\": (string)");
                eg.events.push_back(e.string);
            }
            t.event_groups.push_back(std::move(eg));
        }
    }
    return t;
}

} // namespace

TraceConfig parse_trace_config(std::string const & json_text)
{
    Value root;
    try {
        root = json::parse(json_text);
    } catch (json::parse_error const & e) {
        throw trace_config_error(e.what());
    }

    std::string name, description;
    if (Value const * v = root.get("name"); v && v->is_string())
        name = v->string;
    if (Value const * v = root.get("description"); v && v->is_string())
        description = v->string;
    int numa_domains = 0;
    if (Value const * v = root.get("num_numa_domains"); v && v->is_number())
        numa_domains = (int) v->to_int();
    std::vector<double> bandwidth;
    if (Value const * v = root.get("bandwidth_per_numa_domain"); v && v->is_array())
        bandwidth = numbers_or_null(*v);

    Value const * jcaches = root.get("caches");
    if (!jcaches || !jcaches->is_object())
        throw trace_config_error("Expected \"caches\" object");
    std::map<std::string, Cache> caches;
    for (auto const & kv : jcaches->object)
        caches.emplace(kv.first, cache_from_json(kv.first, kv.second));

    Value const * jthreads = root.get("thread_affinities");
    if (!jthreads || !jthreads->is_array())
        throw trace_config_error("Expected \"thread_affinities\" array");
    std::vector<ThreadAffinity> threads;
    for (Value const & v : jthreads->array)
        threads.push_back(affinity_from_json((int) threads.size(), v));

    return TraceConfig(name, description, numa_domains, bandwidth, caches, threads);
}

TraceConfig read_trace_config(std::string const & path)
{
    std::ifstream f(path);
    if (!f)
        throw trace_config_error(std::strerror(errno));
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse_trace_config(text);
}

TraceConfig default_trace_config(int threads)
{
    std::map<std::string, Cache> caches;
    Cache c;
    c.name = "L1";
    c.size = 32768;
    c.line_size = 64;
    caches.emplace(c.name, c);
    std::vector<ThreadAffinity> t;
    for (int i = 0; i < threads; ++i) {
        ThreadAffinity a;
        a.thread = i;
        a.cpu = i;
        a.cache = "L1";
        t.push_back(a);
    }
    return TraceConfig("default", "generated: no --trace-config given", 1, {}, caches, t);
}

namespace {

void put_doubles(std::ostream & o, std::vector<double> const & v)
{
    if (v.empty()) {
        o << "null";
        return;
    }
    o << '[';
    for (std::size_t i = 0; i < v.size(); ++i)
        o << (i ? ", " : "") << v[i];
    o << ']';
}

std::string quoted_or_null(std::string const & s) { return s.empty() ? "null" : "\"" + s + "\""; }

void put_cache(std::ostream & o, Cache const & c)
{
    o << "{\"size\": " << c.size << ", \"line_size\": " << c.line_size << ", \"bandwidth\": "
      << (c.bandwidth == 0.0 ? std::string("null") : std::to_string(c.bandwidth))
      << ", \"bandwidth_per_numa_domain\": ";
    put_doubles(o, c.bandwidth_per_numa_domain);
    o << ", \"cache_miss_event\": " << quoted_or_null(c.cache_miss_event)
      << ", \"parent\": " << quoted_or_null(c.parent) << '}';
}

void put_event_groups(std::ostream & o, std::vector<EventGroup> const & groups)
{
    if (groups.empty()) {
        o << "[]";
        return;
    }
    o << "[\n";
    for (std::size_t i = 0; i < groups.size(); ++i) {
        EventGroup const & g = groups[i];
        o << "{\"pid\": " << g.pid << ", \"cpu\": " << g.cpu << ", \"events\": ";
        if (g.events.empty()) {
            o << "[]";
        } else {
            o << '[';
            for (std::size_t k = 0; k < g.events.size(); ++k)
                o << (k ? ", " : "") << '"' << g.events[k] << '"';
            o << ']';
        }
        o << '}' << (i + 1 < groups.size() ? ",\n" : "\n");
    }
    o << ']';
}

} // namespace

std::ostream & operator<<(std::ostream & o, TraceConfig const & tc)
{
    o << "{\n"
      << "\"name\": \"" << tc.name() << "\",\n"
      << "\"description\": \"" << tc.description() << "\",\n"
      << "\"num_numa_domains\": " << tc.num_numa_domains() << ",\n"
      << "\"bandwidth_per_numa_domain\": ";
    put_doubles(o, tc.bandwidth_per_numa_domain());
    o << ",\n\"caches\": ";
    if (tc.caches().empty()) {
        o << "{}";
    } else {
        o << "{\n";
        std::size_t i = 0;
        for (auto const & kv : tc.caches()) {
            o << '"' << kv.first << "\": ";
            put_cache(o, kv.second);
            o << (++i < tc.caches().size() ? ",\n" : "\n");
        }
        o << '}';
    }
    o << ",\n\"thread_affinities\": ";
    auto const & threads = tc.thread_affinities();
    if (threads.empty()) {
        o << "[]";
    } else {
        o << "[\n";
        for (std::size_t i = 0; i < threads.size(); ++i) {
            ThreadAffinity const & t = threads[i];
            o << "{\"cpu\": " << t.cpu << ", \"cache\": \"" << t.cache << "\", \"numa_domain\": "
              << t.numa_domain << ", \"event_groups\": ";
            put_event_groups(o, t.event_groups);
            o << '}' << (i + 1 < threads.size() ? ",\n" : "\n");
        }
        o << ']';
    }
    return o << "\n}";
}
