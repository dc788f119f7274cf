/*
 * ref_profile.cpp -- the reference's CSR kernel timed the way SURVEY 8(d) prescribes for the CPU baseline
 * (TEST INFRASTRUCTURE ONLY; bench.py's cpu_baseline leg, through oracle/cpu_baseline_child.py).
 *
 * This file is ours.  oracle/Makefile compiles it with the reference's own sources where they lie under
 * /root/reference/src and with -DHAVE_LIBNUMA into oracle/_ref/libref_profile.so, so that the page placement of
 * csr_spmv_kernel::prepare (src/kernels/csr-spmv.cpp:48-62) is the reference's own distribute_pages
 * (src/util/aligned-allocator.hpp:216-271): the arrays are allocated and first touched by the master thread, as the
 * reference's init does (aligned-allocator.hpp:113-118 runs serially outside a parallel region), then every page is
 * moved to the NUMA node of the thread whose row block it belongs to.  The timed loop is profile_kernel_run's
 * (src/profile-kernel.cpp:137-179): barrier, t0 on the master, barrier, csr_matrix::spmv from every thread of the
 * team, barrier, t1.  Threads are placed by the OpenMP runtime (the parent sets OMP_PROC_BIND=close OMP_PLACES=cores
 * before this process starts, so libgomp sees them); each thread reports the CPU it runs on, and those CPUs are what
 * distribute_pages is given, as profile_kernel gives it the configured thread affinities.
 */
#include "matrix/csr-matrix.hpp"
#include "util/aligned-allocator.hpp"

#include <numa.h>
#include <numaif.h>
#include <omp.h>
#include <sched.h>

#include <chrono>
#include <cstring>
#include <string>
#include <system_error>
#include <vector>

namespace {
thread_local std::string last_error;
}

extern "C" {

const char * ref_profile_last_error() { return last_error.c_str(); }

void * ref_profile_csr_create(int32_t rows, int32_t columns, int32_t num_entries, const int32_t * row_ptr,
                              const int32_t * column_index, const double * value)
{
    try {
        csr_matrix::size_array_type p(row_ptr, row_ptr + rows + 1);
        int32_t n = row_ptr[rows];
        csr_matrix::index_array_type j(column_index, column_index + n);
        csr_matrix::value_array_type a(value, value + n);
        return new csr_matrix::Matrix(rows, columns, num_entries, 1, p, j, a);
    } catch (std::exception const & e) {
        last_error = e.what();
        return nullptr;
    }
}

void ref_profile_csr_free(void * h) { delete static_cast<csr_matrix::Matrix *>(h); }

/* Can this process move its own pages?  (A container may forbid move_pages; then the arrays stay where the master
 * touched them and the caller is told.)  0 = yes, otherwise errno. */
int ref_profile_can_move_pages()
{
    if (numa_available() < 0)
        return ENOSYS;
    void * page = nullptr;
    if (posix_memalign(&page, 4096, 4096) != 0)
        return ENOMEM;
    std::memset(page, 0, 4096);
    int node = numa_node_of_cpu(sched_getcpu());
    int status = 0;
    int err = numa_move_pages(0, 1, &page, &node, &status, MPOL_MF_MOVE);
    int code = err < 0 ? errno : (status < 0 ? -status : 0);
    std::free(page);
    return code;
}

/* ns[runs]: wall time of each timed run; cpus[num_threads], nodes[num_threads]: where the team ran;
 * info[0] = threads the runtime really gave, info[1] = 1 if the pages were distributed, info[2] = NUMA nodes configured.
 * y_out (rows doubles, may be null): y after warm-up + runs multiplies from zero. */
int ref_profile_csr_run(void * h, const double * x, int num_threads, int runs, int distribute, int64_t * ns, int32_t * cpus,
                        int32_t * nodes, int64_t * info, double * y_out)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    try {
        omp_set_dynamic(0);
        omp_set_num_threads(num_threads);
        csr_matrix::value_array_type xv(x, x + A.columns), yv(A.rows, 0.0);
        std::vector<int> cpu_of(num_threads, 0);
        int got = 0;
        #pragma omp parallel
        {
            #pragma omp master
            got = omp_get_num_threads();
        }
        if (got != num_threads) {
            last_error = "the OpenMP runtime gave " + std::to_string(got) + " threads instead of " + std::to_string(num_threads);
            return -1;
        }
        const bool move = distribute && ref_profile_can_move_pages() == 0;
        #pragma omp parallel
        {
            int t = omp_get_thread_num();
            cpu_of[t] = sched_getcpu();
            #pragma omp barrier
            if (move) {  /* csr-spmv.cpp:48-62, every thread of the team calls it */
                distribute_pages(A.row_ptr.data(), A.row_ptr.size(), num_threads, cpu_of.data());
                distribute_pages(A.column_index.data(), A.column_index.size(), num_threads, cpu_of.data());
                distribute_pages(A.value.data(), A.value.size(), num_threads, cpu_of.data());
                distribute_pages(xv.data(), xv.size(), num_threads, cpu_of.data());
                distribute_pages(yv.data(), yv.size(), num_threads, cpu_of.data());
            }
            csr_matrix::spmv(A, xv, yv); /* the warm-up run (main.cpp:251 always asks for it) */
            for (int r = 0; r < runs; r++) {
                std::chrono::steady_clock::time_point t0, t1;
                #pragma omp barrier
                #pragma omp master
                t0 = std::chrono::steady_clock::now();
                #pragma omp barrier
                csr_matrix::spmv(A, xv, yv);
                #pragma omp barrier
                #pragma omp master
                {
                    t1 = std::chrono::steady_clock::now();
                    ns[r] = (t1 - t0).count();
                }
                #pragma omp barrier
            }
        }
        for (int t = 0; t < num_threads; t++) {
            cpus[t] = cpu_of[t];
            nodes[t] = numa_available() < 0 ? -1 : numa_node_of_cpu(cpu_of[t]);
        }
        info[0] = got;
        info[1] = move ? 1 : 0;
        info[2] = numa_available() < 0 ? 0 : numa_num_configured_nodes();
        if (y_out)
            std::memcpy(y_out, yv.data(), yv.size() * sizeof(double));
        return 0;
    } catch (std::exception const & e) {
        last_error = e.what();
        return -1;
    }
}

}
