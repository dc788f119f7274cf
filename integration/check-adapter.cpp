/*
 * check-adapter -- the drop-in, end to end, inside the reference's own program text.
 *
 * Built (oracle/Makefile, target _ref/check-adapter) from a temporary copy of the reference tree with
 * integration/reference.patch applied: the reference's loader, converters, trace-config reader and its timed loop
 * (profile_kernel, src/profile-kernel.cpp:197-313) drive a hip_*_spmv_kernel (integration/src/kernels/hip-spmv.cpp);
 * afterwards the vector left on the device is compared with the same number of multiplies by the reference's own CPU
 * kernel on the same matrix (csr_matrix::spmv, coo_matrix::spmv, ell_matrix::spmv).
 *
 *     check-adapter csr|coo|ell MATRIX CONFIG.json RUNS
 *
 * stdout: the reference's Profiling document (operator<<, src/profile-kernel.cpp:340-391).
 * stderr: one line {"check": ...}.  Exit 0 when every row is within SURVEY 8(d)'s tolerance.
 * Test infrastructure (tests/test_gpu_reference_binary.py); nothing in the product links it.
 */
#include "kernels.hpp"
#include "profile-kernel.hpp"
#include "trace-config.hpp"
#include "matrix/matrix-market.hpp"
#include "util/json-ostreambuf.hpp"
#include "util/perf-events.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace
{

struct Outcome
{
    double worst_rel = 0.0, worst_over_bound = 0.0, norm = 0.0;
    long rows = 0, outside = 0, by_summation_bound = 0;
    bool bitexact = true;
};

/* What two summation orders of one row may differ by, a priori, after `multiplies` accumulating multiplies:
 * 2 k (n_i + 1) 2^-53 (|A||x|)_i  (tests/helpers.py::assert_close's second clause). */
std::vector<double> summation_slack(matrix_market::Matrix const & mm, int multiplies)
{
    csr_matrix::Matrix A = csr_matrix::from_matrix_market(mm);
    for (auto & v : A.value)
        v = std::fabs(v);
    csr_matrix::value_array_type x(A.columns, 1.0), a(A.rows, 0.0);
    csr_matrix::spmv(A, x, a);
    std::vector<double> slack(A.rows);
    for (csr_matrix::index_type i = 0; i < A.rows; i++)
        slack[i] = 2.0 * multiplies * (A.row_ptr[i + 1] - A.row_ptr[i] + 1) * std::ldexp(1.0, -53) * a[i];
    return slack;
}

/* |y_gpu - y_cpu| <= 1e-10 * max(|y_cpu_i|, 1e-6 * ||y_cpu||_inf) per row.  That floor is one unit in the
 * last place of the largest element, so a row whose products cancel can miss it by rounding alone when its sum is formed
 * in another order: such a row must then be within the a-priori summation bound, and is counted. */
Outcome compare(std::vector<double> const & got, double const * want, long n, std::vector<double> const & slack)
{
    Outcome o;
    o.rows = n;
    for (long i = 0; i < n; i++)
        o.norm = std::max(o.norm, std::fabs(want[i]));
    for (long i = 0; i < n; i++) {
        double d = std::fabs(got[i] - want[i]);
        if (!(d == 0.0))
            o.bitexact = false;
        double bound = 1e-10 * std::max(std::fabs(want[i]), 1e-6 * o.norm);
        if (bound > 0.0)
            o.worst_over_bound = std::max(o.worst_over_bound, d / bound);
        else if (d > 0.0)
            o.worst_over_bound = INFINITY;
        if (!(d <= bound)) {
            if (d <= slack[i])
                o.by_summation_bound++;
            else
                o.outside++;
        }
        if (o.norm > 0.0)
            o.worst_rel = std::max(o.worst_rel, d / o.norm);
    }
    return o;
}

template <class HipKernel>
int drive(std::string const & format, std::string const & path, TraceConfig const & config, int runs)
{
    HipKernel kernel(path);
    kernel.init(config, std::cerr, false);
    perf::libpfm_context libpfm;  /* never asked for an event group by a configuration without any */
    Profiling profiling = profile_kernel(config, kernel, true, false, runs, libpfm, std::cerr, false);
    {
        auto o = json_ostreambuf(std::cout);
        std::cout << profiling << '\n';
    }
    std::vector<double> got = kernel.result();

    /* the same multiplies on the CPU: one warm-up run + `runs` timed ones, y accumulating from zero */
    matrix_market::Matrix mm = matrix_market::load_matrix(path, std::cerr, false);
    Outcome out;
    int const multiplies = runs + 1;
    std::vector<double> slack = summation_slack(mm, multiplies);
    if (format == "csr") {
        csr_matrix::Matrix A = csr_matrix::from_matrix_market(mm);
        csr_matrix::value_array_type x(A.columns, 1.0), y(A.rows, 0.0);
        for (int k = 0; k < multiplies; k++)
            csr_matrix::spmv(A, x, y);
        out = compare(got, y.data(), A.rows, slack);
    } else if (format == "coo") {
        coo_matrix::Matrix A = coo_matrix::from_matrix_market(mm);
        coo_matrix::value_array_type x(A.columns, 1.0), y(A.rows, 0.0), workspace;
        for (int k = 0; k < multiplies; k++)
            coo_matrix::spmv(1, A, x, y, workspace);
        out = compare(got, y.data(), A.rows, slack);
    } else {
        ell_matrix::Matrix A = ell_matrix::from_matrix_market(mm);
        ell_matrix::value_array_type x(A.columns, 1.0), y(A.rows, 0.0);
        for (int k = 0; k < multiplies; k++)
            ell_matrix::spmv(A, x, y);
        out = compare(got, y.data(), A.rows, slack);
    }
    std::fprintf(stderr,
                 "{\"check\": {\"kernel\": \"%s\", \"against\": \"%s_matrix::spmv of the reference, %d multiplies from y = 0\", "
                 "\"rows\": %ld, \"max_rel_err\": %.3e, \"worst_row_over_8d_bound\": %.3e, \"rows_within_summation_bound_only\": %ld, "
                 "\"rows_outside_both_bounds\": %ld, \"bitexact\": %s, \"pass\": %s}}\n",
                 kernel.name().c_str(), format.c_str(), multiplies, out.rows, out.worst_rel, out.worst_over_bound, out.by_summation_bound,
                 out.outside, out.bitexact ? "true" : "false", out.outside == 0 ? "true" : "false");
    return out.outside == 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}

}

int main(int argc, char ** argv)
{
    if (argc != 5) {
        std::cerr << "usage: check-adapter csr|coo|ell MATRIX CONFIG.json RUNS\n";
        return 2;
    }
    std::string format = argv[1], path = argv[2], config_path = argv[3];
    int runs = std::atoi(argv[4]);
    if (runs < 1 || (format != "csr" && format != "coo" && format != "ell")) {
        std::cerr << "check-adapter: bad format or run count\n";
        return 2;
    }
    try {
        TraceConfig config = read_trace_config(config_path);
        if (format == "csr")
            return drive<hip_csr_spmv_kernel>(format, path, config, runs);
        if (format == "coo")
            return drive<hip_coo_spmv_kernel>(format, path, config, runs);
        return drive<hip_ell_spmv_kernel>(format, path, config, runs);
    } catch (trace_config_error const & e) {
        std::cerr << config_path << ": " << e.what() << '\n';
    } catch (kernel_error const & e) {
        std::cerr << "check-adapter: " << e.what() << '\n';
    } catch (perf::perf_error const & e) {
        std::cerr << e.what() << '\n';
    } catch (std::exception const & e) {
        std::cerr << "check-adapter: " << e.what() << '\n';
    }
    return EXIT_FAILURE;
}
