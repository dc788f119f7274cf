/*
 * ref_shim.cpp -- C-ABI wrapper around the REFERENCE library itself
 * (TEST INFRASTRUCTURE ONLY).
 *
 * This file is ours; it contains no reference code.  oracle/Makefile compiles
 * it together with the reference's own sources *where they lie* under
 * /root/reference/src (matrix/ and the two util streambufs) into
 * oracle/_ref/libref_spmv.so.  That .so is git-ignored, travels to the GPU box
 * with the snapshot, and is used (a) to pin oracle/spmv_oracle.c bit-for-bit,
 * (b) to generate tests/golden/ fixtures, (c) optionally as bench.py's
 * cpu_baseline of kind "reference".  Nothing here reads /root/reference at run
 * time.
 *
 * Wrapped reference entry points:
 *   matrix_market::fromStream / load_matrix      src/matrix/matrix-market.cpp:530-555, 777-861
 *   {csr,coo,ell}_matrix::from_matrix_market     csr-matrix.cpp:187-243, coo-matrix.cpp:220-243, ell-matrix.cpp:190-238
 *   {csr,coo,ell}_matrix::spmv                   csr-matrix-spmv.cpp:148-167, coo-matrix.cpp:313-335, ell-matrix.cpp:311-335
 *   print_sample                                 src/util/sample.hpp:137-165
 *   read_trace_config / operator<<(TraceConfig)  src/trace-config.cpp:386-404, 579-597
 */
#include "matrix/coo-matrix.hpp"
#include "matrix/csr-matrix.hpp"
#include "matrix/ell-matrix.hpp"
#include "matrix/hybrid-matrix.hpp"
#include "matrix/matrix-error.hpp"
#include "matrix/matrix-market.hpp"
#include "util/json-ostreambuf.hpp"
#include "util/sample.hpp"
#include "trace-config.hpp"

#include <omp.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {
thread_local std::string last_error;

template <typename F> int guarded(F && f)
{
    try {
        f();
        return 0;
    } catch (std::exception const & e) {
        last_error = e.what();
        return -1;
    }
}
}

extern "C" {

const char * ref_last_error() { return last_error.c_str(); }

/* ---- Matrix Market ---------------------------------------------------- */

void * ref_mm_from_string(const char * text, int64_t len)
{
    matrix_market::Matrix * m = nullptr;
    int rc = guarded([&] {
        std::istringstream s(std::string(text, (size_t) len));
        m = new matrix_market::Matrix(matrix_market::fromStream(s));
    });
    return rc == 0 ? m : nullptr;
}

void * ref_mm_load(const char * path)
{
    matrix_market::Matrix * m = nullptr;
    int rc = guarded([&] {
        std::ostringstream log;
        m = new matrix_market::Matrix(matrix_market::load_matrix(path, log, false));
    });
    return rc == 0 ? m : nullptr;
}

void ref_mm_free(void * h) { delete static_cast<matrix_market::Matrix *>(h); }

/* out[0..5] = rows, columns, num_entries, format, field, symmetry */
void ref_mm_info(void * h, int32_t * out)
{
    auto & m = *static_cast<matrix_market::Matrix *>(h);
    out[0] = m.rows();
    out[1] = m.columns();
    out[2] = m.num_entries();
    out[3] = (int32_t) m.format();
    out[4] = (int32_t) m.field();
    out[5] = (int32_t) m.symmetry();
}

void ref_mm_entries(void * h, int32_t * i, int32_t * j, double * a)
{
    auto & m = *static_cast<matrix_market::Matrix *>(h);
    auto ri = m.row_indices();
    auto ci = m.column_indices();
    auto v = m.values_real();
    std::memcpy(i, ri.data(), ri.size() * sizeof(int32_t));
    std::memcpy(j, ci.data(), ci.size() * sizeof(int32_t));
    std::memcpy(a, v.data(), v.size() * sizeof(double));
}

int32_t ref_mm_max_row_length(void * h)
{
    return static_cast<matrix_market::Matrix *>(h)->max_row_length();
}

/* ---- CSR ---------------------------------------------------------------- */

void * ref_csr_from_mm(void * h, int32_t row_alignment)
{
    csr_matrix::Matrix * A = nullptr;
    int rc = guarded([&] {
        A = new csr_matrix::Matrix(csr_matrix::from_matrix_market_row_aligned(
            *static_cast<matrix_market::Matrix *>(h), row_alignment));
    });
    return rc == 0 ? A : nullptr;
}

void * ref_csr_from_arrays(int32_t rows, int32_t columns, int32_t num_entries,
                           const int32_t * row_ptr, const int32_t * column_index,
                           const double * value)
{
    csr_matrix::size_array_type p(row_ptr, row_ptr + rows + 1);
    int32_t n = row_ptr[rows];
    csr_matrix::index_array_type j(column_index, column_index + n);
    csr_matrix::value_array_type a(value, value + n);
    return new csr_matrix::Matrix(rows, columns, num_entries, 1, p, j, a);
}

void ref_csr_free(void * h) { delete static_cast<csr_matrix::Matrix *>(h); }

/* out[0..5] = rows, columns, num_entries, row_alignment, stored entries, size() bytes */
void ref_csr_info(void * h, int64_t * out)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    out[0] = A.rows;
    out[1] = A.columns;
    out[2] = A.num_entries;
    out[3] = A.row_alignment;
    out[4] = (int64_t) A.value.size();
    out[5] = (int64_t) A.size();
}

void ref_csr_arrays(void * h, int32_t * row_ptr, int32_t * column_index, double * value)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    std::memcpy(row_ptr, A.row_ptr.data(), A.row_ptr.size() * sizeof(int32_t));
    std::memcpy(column_index, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(value, A.value.data(), A.value.size() * sizeof(double));
}

/* y += A*x, `runs` times, called the way src/profile-kernel.cpp:227,160 calls
 * it: from every thread of one parallel region. */
int ref_csr_spmv(void * h, const double * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    return guarded([&] {
        csr_matrix::value_array_type xv(x, x + A.columns);
        csr_matrix::value_array_type yv(y, y + A.rows);
        omp_set_num_threads(num_threads);
        #pragma omp parallel
        {
            for (int r = 0; r < runs; r++) {
                csr_matrix::spmv(A, xv, yv);
                #pragma omp barrier
            }
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

/* ---- COO ---------------------------------------------------------------- */

void * ref_coo_from_mm(void * h)
{
    coo_matrix::Matrix * A = nullptr;
    int rc = guarded([&] {
        A = new coo_matrix::Matrix(coo_matrix::from_matrix_market(
            *static_cast<matrix_market::Matrix *>(h)));
    });
    return rc == 0 ? A : nullptr;
}

void * ref_coo_from_arrays(int32_t rows, int32_t columns, int32_t num_entries,
                           const int32_t * row_index, const int32_t * column_index,
                           const double * value)
{
    coo_matrix::index_array_type i(row_index, row_index + num_entries);
    coo_matrix::index_array_type j(column_index, column_index + num_entries);
    coo_matrix::value_array_type a(value, value + num_entries);
    return new coo_matrix::Matrix(rows, columns, num_entries, i, j, a);
}

void ref_coo_free(void * h) { delete static_cast<coo_matrix::Matrix *>(h); }

void ref_coo_info(void * h, int64_t * out)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    out[0] = A.rows;
    out[1] = A.columns;
    out[2] = A.num_entries;
    out[3] = (int64_t) A.size();
}

void ref_coo_arrays(void * h, int32_t * row_index, int32_t * column_index, double * value)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    std::memcpy(row_index, A.row_index.data(), A.row_index.size() * sizeof(int32_t));
    std::memcpy(column_index, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(value, A.value.data(), A.value.size() * sizeof(double));
}

/* The workspace is allocated zeroed once and reused across `runs`, as
 * src/kernels/coo-spmv.cpp:41-48,76-81 does. */
int ref_coo_spmv(void * h, const double * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    return guarded([&] {
        coo_matrix::value_array_type xv(x, x + A.columns);
        coo_matrix::value_array_type yv(y, y + A.rows);
        coo_matrix::value_array_type ws((size_t) num_threads * A.rows, 0.0);
        omp_set_num_threads(num_threads);
        #pragma omp parallel
        {
            for (int r = 0; r < runs; r++) {
                coo_matrix::spmv(num_threads, A, xv, yv, ws);
                #pragma omp barrier
            }
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

/* ---- ELL ---------------------------------------------------------------- */

void * ref_ell_from_mm(void * h, int skip_padding)
{
    ell_matrix::Matrix * A = nullptr;
    int rc = guarded([&] {
        A = new ell_matrix::Matrix(ell_matrix::from_matrix_market(
            *static_cast<matrix_market::Matrix *>(h), skip_padding != 0));
    });
    return rc == 0 ? A : nullptr;
}

void ref_ell_free(void * h) { delete static_cast<ell_matrix::Matrix *>(h); }

void ref_ell_info(void * h, int64_t * out)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    out[0] = A.rows;
    out[1] = A.columns;
    out[2] = A.num_entries;
    out[3] = A.row_length;
    out[4] = (int64_t) A.value.size();
    out[5] = (int64_t) A.size();
}

void ref_ell_arrays(void * h, int32_t * column_index, double * value)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    std::memcpy(column_index, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(value, A.value.data(), A.value.size() * sizeof(double));
}

int ref_ell_spmv(void * h, const double * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    return guarded([&] {
        ell_matrix::value_array_type xv(x, x + A.columns);
        ell_matrix::value_array_type yv(y, y + A.rows);
        omp_set_num_threads(num_threads);
        #pragma omp parallel
        {
            for (int r = 0; r < runs; r++) {
                ell_matrix::spmv(A, xv, yv);
                #pragma omp barrier
            }
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

/* ---- HYBRID ------------------------------------------------------------------- */

void * ref_hybrid_from_mm(void * h, int skip_padding)
{
    hybrid_matrix::Matrix * A = nullptr;
    int rc = guarded([&] {
        std::ostringstream log;
        A = new hybrid_matrix::Matrix(hybrid_matrix::from_matrix_market(
            *static_cast<matrix_market::Matrix *>(h), skip_padding != 0, log, false));
    });
    return rc == 0 ? A : nullptr;
}

void ref_hybrid_free(void * h) { delete static_cast<hybrid_matrix::Matrix *>(h); }

/* out[0..6] = rows, columns, num_entries, ell_row_length, stored ell entries, coo entries, size() */
void ref_hybrid_info(void * h, int64_t * out)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    out[0] = A.rows;
    out[1] = A.columns;
    out[2] = A.num_entries;
    out[3] = A.ell_row_length;
    out[4] = (int64_t) A.ell_value.size();
    out[5] = A.num_coo_entries;
    out[6] = (int64_t) A.size();
}

void ref_hybrid_arrays(void * h, int32_t * ej, double * ea, int32_t * cr, int32_t * cc, double * cv)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    std::memcpy(ej, A.ell_column_index.data(), A.ell_column_index.size() * sizeof(int32_t));
    std::memcpy(ea, A.ell_value.data(), A.ell_value.size() * sizeof(double));
    std::memcpy(cr, A.coo_row_index.data(), A.coo_row_index.size() * sizeof(int32_t));
    std::memcpy(cc, A.coo_column_index.data(), A.coo_column_index.size() * sizeof(int32_t));
    std::memcpy(cv, A.coo_value.data(), A.coo_value.size() * sizeof(double));
}

int ref_hybrid_spmv(void * h, const double * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    return guarded([&] {
        hybrid_matrix::value_array_type xv(x, x + A.columns);
        hybrid_matrix::value_array_type yv(y, y + A.rows);
        hybrid_matrix::value_array_type ws((size_t) num_threads * A.rows, 0.0);
        omp_set_num_threads(num_threads);
        #pragma omp parallel
        {
            for (int r = 0; r < runs; r++) {
                hybrid_matrix::spmv(num_threads, A, xv, yv, ws);
                #pragma omp barrier
            }
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

/* ---- Timed CSR loop for bench.py's cpu_baseline (kind "reference") -------
 * Mirrors src/profile-kernel.cpp:137-179: barrier, master t0, barrier, run,
 * barrier, master t1.  One warm-up run first (src/main.cpp:251).  ns[] gets
 * `runs` wall times in nanoseconds. */
int ref_csr_spmv_timed(void * h, const double * x, double * y, int num_threads,
                       int runs, int64_t * ns)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    return guarded([&] {
        omp_set_num_threads(num_threads);
        csr_matrix::value_array_type xv(A.columns, 0.0), yv(A.rows, 0.0);
        std::memcpy(xv.data(), x, xv.size() * sizeof(double));
        std::memcpy(yv.data(), y, yv.size() * sizeof(double));
        #pragma omp parallel
        {
            csr_matrix::spmv(A, xv, yv);
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
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

/* ---- print_sample through the reference's JSON stream buffer ------------- */

int64_t ref_print_sample(const int64_t * v, int64_t n, char * out, int64_t cap)
{
    std::ostringstream s;
    {
        json_ostreambuf buf(s);
        std::vector<int64_t> vv(v, v + n);
        print_sample(s, vv, std::string("ns"));
    }
    std::string str = s.str();
    if ((int64_t) str.size() + 1 > cap)
        return -(int64_t) str.size() - 1;
    std::memcpy(out, str.c_str(), str.size() + 1);
    return (int64_t) str.size();
}

/* ---- trace-config: parse a file and echo it through the JSON stream buffer ------
 * Returns the length written, or a negative number: -1 = trace_config_error (message in
 * ref_last_error), -(n+2) = buffer too small for n bytes. */
int64_t ref_trace_config_echo(const char * path, char * out, int64_t cap, int32_t * info)
{
    std::string str;
    int rc = guarded([&] {
        TraceConfig tc = read_trace_config(path);
        std::ostringstream s;
        {
            json_ostreambuf buf(s);
            s << tc;
        }
        str = s.str();
        info[0] = (int32_t) tc.thread_affinities().size();
        info[1] = tc.num_numa_domains();
        info[2] = (int32_t) tc.caches().size();
        info[3] = (int32_t) tc.max_cache_size();
    });
    if (rc != 0)
        return -1;
    if ((int64_t) str.size() + 1 > cap)
        return -(int64_t) str.size() - 2;
    std::memcpy(out, str.c_str(), str.size() + 1);
    return (int64_t) str.size();
}

}
