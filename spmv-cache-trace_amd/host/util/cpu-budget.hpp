#ifndef SPMV_HOST_CPU_BUDGET_HPP
#define SPMV_HOST_CPU_BUDGET_HPP

// Cores this process may really use: the affinity mask capped by the cgroup CPU quota.  A container on a big host
// (a one-GPU share of a 256-thread machine) sees every CPU but is granted a fraction of them; OpenMP's default team
// of one thread per visible CPU then spends its time being descheduled -- the loader and the converters, and above
// all the barriers of the timed loop.  When OMP_NUM_THREADS is not set, the host library caps OpenMP's default team
// size at this number once, at load time (cpu-budget.cpp).
int cpu_budget();

#endif
