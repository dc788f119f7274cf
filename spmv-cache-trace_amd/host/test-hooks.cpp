// test-hooks.cpp -- C entry points into the host library for the Python test-suite.
//
// Not part of the product surface (the CLI does not call these): they let tests/ feed golden
// Matrix Market text to the loader and converters and read back the arrays, statistics and
// JSON echo, so that they can be compared with the oracle and the reference library.
#include "matrix/coo-matrix.hpp"
#include "matrix/csr-matrix.hpp"
#include "matrix/ell-matrix.hpp"
#include "matrix/hybrid-matrix.hpp"
#include "matrix/matrix-error.hpp"
#include "matrix/matrix-market.hpp"
#include "trace-config.hpp"
#include "util/json-ostreambuf.hpp"
#include "util/sample.hpp"

#include <cstring>
#include <sstream>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#endif

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

long long put_string(std::string const & s, char * out, long long cap)
{
    if ((long long) s.size() + 1 > cap)
        return -(long long) s.size() - 2;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (long long) s.size();
}
}

extern "C" {

char const * host_last_error() { return last_error.c_str(); }

void * host_mm_from_buffer(char const * text, long long len)
{
    matrix_market::Matrix * m = nullptr;
    return guarded([&] { m = new matrix_market::Matrix(matrix_market::fromBuffer(text, (std::size_t) len)); }) == 0 ? m : nullptr;
}
void * host_mm_load(char const * path)
{
    matrix_market::Matrix * m = nullptr;
    return guarded([&] {
        std::ostringstream log;
        m = new matrix_market::Matrix(matrix_market::load_matrix(path, log, false));
    }) == 0 ? m : nullptr;
}
void * host_mm_load_tar_gz_member(char const * path, char const * member)
{
    matrix_market::Matrix * m = nullptr;
    return guarded([&] { m = new matrix_market::Matrix(matrix_market::load_tar_gz_member(path, member)); }) == 0 ? m : nullptr;
}
void * host_mm_expand_symmetry(void * h)
{
    matrix_market::Matrix * m = nullptr;
    return guarded([&] { m = new matrix_market::Matrix(matrix_market::expand_symmetry(*static_cast<matrix_market::Matrix *>(h))); }) == 0 ? m : nullptr;
}
void host_mm_free(void * h) { delete static_cast<matrix_market::Matrix *>(h); }
void host_mm_info(void * h, int32_t * out)
{
    auto & m = *static_cast<matrix_market::Matrix *>(h);
    out[0] = m.rows(); out[1] = m.columns(); out[2] = m.num_entries();
    out[3] = (int32_t) m.format(); out[4] = (int32_t) m.field(); out[5] = (int32_t) m.symmetry();
    out[6] = (int32_t) m.comments().size();
}
int host_mm_entries(void * h, int32_t * i, int32_t * j, double * a)
{
    auto & m = *static_cast<matrix_market::Matrix *>(h);
    return guarded([&] {
        auto const v = m.values_real();
        std::memcpy(i, m.row_indices().data(), m.row_indices().size() * sizeof(int32_t));
        std::memcpy(j, m.column_indices().data(), m.column_indices().size() * sizeof(int32_t));
        std::memcpy(a, v.data(), v.size() * sizeof(double));
    });
}
long long host_mm_comment(void * h, int k, char * out, long long cap)
{
    return put_string(static_cast<matrix_market::Matrix *>(h)->comments().at((std::size_t) k), out, cap);
}
int32_t host_mm_max_row_length(void * h)
{
    int32_t r = -1;
    guarded([&] { r = static_cast<matrix_market::Matrix *>(h)->max_row_length(); });
    return r;
}
int host_mm_sorted_entries(void * h, int column_major, int32_t * i, int32_t * j, double * a)
{
    auto & m = *static_cast<matrix_market::Matrix *>(h);
    return guarded([&] {
        auto s = column_major ? matrix_market::sort_matrix_column_major(m) : matrix_market::sort_matrix_row_major(m);
        auto const v = s.values_real();
        std::memcpy(i, s.row_indices().data(), s.row_indices().size() * sizeof(int32_t));
        std::memcpy(j, s.column_indices().data(), s.column_indices().size() * sizeof(int32_t));
        std::memcpy(a, v.data(), v.size() * sizeof(double));
    });
}

// ---- CSR --------------------------------------------------------------------------------
void * host_csr_from_mm(void * h, int32_t row_alignment)
{
    csr_matrix::Matrix * A = nullptr;
    return guarded([&] {
        A = new csr_matrix::Matrix(csr_matrix::from_matrix_market_row_aligned(*static_cast<matrix_market::Matrix *>(h), row_alignment));
    }) == 0 ? A : nullptr;
}
void host_csr_free(void * h) { delete static_cast<csr_matrix::Matrix *>(h); }
void host_csr_info(void * h, long long * out)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    out[0] = A.rows; out[1] = A.columns; out[2] = A.num_entries; out[3] = A.row_alignment;
    out[4] = (long long) A.value.size(); out[5] = (long long) A.size();
}
void host_csr_arrays(void * h, int32_t * p, int32_t * j, double * a)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    std::memcpy(p, A.row_ptr.data(), A.row_ptr.size() * sizeof(int32_t));
    std::memcpy(j, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(a, A.value.data(), A.value.size() * sizeof(double));
}
int host_csr_spmv(void * h, double const * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<csr_matrix::Matrix *>(h);
    return guarded([&] {
        csr_matrix::value_array_type xv(x, x + A.columns), yv(y, y + A.rows);
#pragma omp parallel num_threads(num_threads)
        for (int r = 0; r < runs; ++r) {
            csr_matrix::spmv(A, xv, yv);
#pragma omp barrier
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

// ---- COO --------------------------------------------------------------------------------
void * host_coo_from_mm(void * h)
{
    coo_matrix::Matrix * A = nullptr;
    return guarded([&] { A = new coo_matrix::Matrix(coo_matrix::from_matrix_market(*static_cast<matrix_market::Matrix *>(h))); }) == 0 ? A : nullptr;
}
void host_coo_free(void * h) { delete static_cast<coo_matrix::Matrix *>(h); }
void host_coo_info(void * h, long long * out)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    out[0] = A.rows; out[1] = A.columns; out[2] = A.num_entries; out[3] = (long long) A.size();
}
void host_coo_arrays(void * h, int32_t * r, int32_t * c, double * v)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    std::memcpy(r, A.row_index.data(), A.row_index.size() * sizeof(int32_t));
    std::memcpy(c, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(v, A.value.data(), A.value.size() * sizeof(double));
}
int host_coo_spmv(void * h, double const * x, double * y, int num_threads, int runs, int atomic)
{
    auto & A = *static_cast<coo_matrix::Matrix *>(h);
    return guarded([&] {
        coo_matrix::value_array_type xv(x, x + A.columns), yv(y, y + A.rows);
        coo_matrix::value_array_type ws((std::size_t) num_threads * (std::size_t) A.rows, 0.0);
#pragma omp parallel num_threads(num_threads)
        for (int r = 0; r < runs; ++r) {
            if (atomic)
                coo_matrix::spmv_atomic(num_threads, A, xv, yv);
            else
                coo_matrix::spmv(num_threads, A, xv, yv, ws);
#pragma omp barrier
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

// ---- ELL --------------------------------------------------------------------------------
void * host_ell_from_mm(void * h, int skip_padding)
{
    ell_matrix::Matrix * A = nullptr;
    return guarded([&] { A = new ell_matrix::Matrix(ell_matrix::from_matrix_market(*static_cast<matrix_market::Matrix *>(h), skip_padding != 0)); }) == 0 ? A : nullptr;
}
void host_ell_free(void * h) { delete static_cast<ell_matrix::Matrix *>(h); }
void host_ell_info(void * h, long long * out)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    out[0] = A.rows; out[1] = A.columns; out[2] = A.num_entries; out[3] = A.row_length;
    out[4] = (long long) A.value.size(); out[5] = (long long) A.size();
}
void host_ell_arrays(void * h, int32_t * c, double * v)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    std::memcpy(c, A.column_index.data(), A.column_index.size() * sizeof(int32_t));
    std::memcpy(v, A.value.data(), A.value.size() * sizeof(double));
}
int host_ell_spmv(void * h, double const * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<ell_matrix::Matrix *>(h);
    return guarded([&] {
        ell_matrix::value_array_type xv(x, x + A.columns), yv(y, y + A.rows);
#pragma omp parallel num_threads(num_threads)
        for (int r = 0; r < runs; ++r) {
            ell_matrix::spmv(A, xv, yv);
#pragma omp barrier
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

// ---- HYBRID ------------------------------------------------------------------------------
void * host_hybrid_from_mm(void * h, int skip_padding)
{
    hybrid_matrix::Matrix * A = nullptr;
    return guarded([&] { A = new hybrid_matrix::Matrix(hybrid_matrix::from_matrix_market(*static_cast<matrix_market::Matrix *>(h), skip_padding != 0)); }) == 0 ? A : nullptr;
}
void host_hybrid_free(void * h) { delete static_cast<hybrid_matrix::Matrix *>(h); }
void host_hybrid_info(void * h, long long * out)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    out[0] = A.rows; out[1] = A.columns; out[2] = A.num_entries; out[3] = A.ell_row_length;
    out[4] = (long long) A.ell_value.size(); out[5] = A.num_coo_entries; out[6] = (long long) A.size();
}
void host_hybrid_arrays(void * h, int32_t * ej, double * ea, int32_t * cr, int32_t * cc, double * cv)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    std::memcpy(ej, A.ell_column_index.data(), A.ell_column_index.size() * sizeof(int32_t));
    std::memcpy(ea, A.ell_value.data(), A.ell_value.size() * sizeof(double));
    std::memcpy(cr, A.coo_row_index.data(), A.coo_row_index.size() * sizeof(int32_t));
    std::memcpy(cc, A.coo_column_index.data(), A.coo_column_index.size() * sizeof(int32_t));
    std::memcpy(cv, A.coo_value.data(), A.coo_value.size() * sizeof(double));
}
int host_hybrid_spmv(void * h, double const * x, double * y, int num_threads, int runs)
{
    auto & A = *static_cast<hybrid_matrix::Matrix *>(h);
    return guarded([&] {
        hybrid_matrix::value_array_type xv(x, x + A.columns), yv(y, y + A.rows);
        hybrid_matrix::value_array_type ws((std::size_t) num_threads * (std::size_t) A.rows, 0.0);
#pragma omp parallel num_threads(num_threads)
        for (int r = 0; r < runs; ++r) {
            hybrid_matrix::spmv(num_threads, A, xv, yv, ws);
#pragma omp barrier
        }
        std::memcpy(y, yv.data(), yv.size() * sizeof(double));
    });
}

// ---- statistics + JSON ---------------------------------------------------------------------
long long host_print_sample(long long const * v, long long n, char * out, long long cap)
{
    std::ostringstream s;
    {
        json_ostreambuf pretty(s);
        std::vector<long long> vv(v, v + n);
        print_sample(s, vv, "ns");
    }
    return put_string(s.str(), out, cap);
}

long long host_trace_config_echo(char const * path, char * out, long long cap, int32_t * info)
{
    std::string text;
    int rc = guarded([&] {
        TraceConfig tc = read_trace_config(path);
        std::ostringstream s;
        {
            json_ostreambuf pretty(s);
            s << tc;
        }
        text = s.str();
        info[0] = (int32_t) tc.thread_affinities().size();
        info[1] = tc.num_numa_domains();
        info[2] = (int32_t) tc.caches().size();
        info[3] = (int32_t) tc.max_cache_size();
    });
    return rc == 0 ? put_string(text, out, cap) : -1;
}

} // extern "C"
