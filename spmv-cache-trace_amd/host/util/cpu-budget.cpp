#include "cpu-budget.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <sched.h>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

bool read_two(char const * path, long long & a, long long & b, bool & a_is_max)
{
    std::FILE * f = std::fopen(path, "r");
    if (!f)
        return false;
    char first[64] = {0};
    int const n = std::fscanf(f, "%63s %lld", first, &b);
    std::fclose(f);
    if (n < 1)
        return false;
    a_is_max = !std::strcmp(first, "max");
    a = a_is_max ? -1 : std::atoll(first);
    if (n < 2)
        b = 0;
    return true;
}

} // namespace

int cpu_budget()
{
    int n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        n = std::max(1, CPU_COUNT(&set));
    long long quota = -1, period = 0;
    bool is_max = false;
    if (read_two("/sys/fs/cgroup/cpu.max", quota, period, is_max)) { // cgroup v2: "<quota|max> <period>"
        if (!is_max && quota > 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, quota / period));
    } else {
        long long q = -1, p = 0, unused = 0;
        bool m = false;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, unused, m) && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p, unused, m)
            && q > 0 && p > 0)
            n = std::min<long long>(n, std::max<long long>(1, q / p));
    }
    return n;
}

#ifdef _OPENMP
namespace {
struct OmpBudget {
    OmpBudget()
    {
        if (std::getenv("OMP_NUM_THREADS"))
            return; // the user's word stands
        int const budget = cpu_budget();
        if (budget < omp_get_max_threads())
            omp_set_num_threads(budget);
    }
} const omp_budget_at_load;
} // namespace
#endif
