// host-api.cpp -- implementation of include/spmv_host.h: the loader and the format converters
// behind a C ABI (what Kernel::init does in the reference, src/kernels/csr-spmv.cpp:26-46, for
// front ends that are not C++).
#include "spmv_host.h"

#include "matrix/coo-matrix.hpp"
#include "matrix/csr-matrix.hpp"
#include "matrix/ell-matrix.hpp"
#include "matrix/hybrid-matrix.hpp"
#include "matrix/matrix-error.hpp"
#include "matrix/matrix-market.hpp"
#include "matrix/synthetic.hpp"

#include <algorithm>
#include <new>
#include <sstream>
#include <string>
#include <system_error>

struct spmv_host_matrix
{
    int format = 0;
    long long rows_total = 0;
    bool expanded = false;
    csr_matrix::Matrix csr;
    coo_matrix::Matrix coo;
    ell_matrix::Matrix ell;
    hybrid_matrix::Matrix hybrid;
};

namespace {

thread_local std::string g_error;

template <typename F> int guarded(F && f)
{
    try {
        f();
        return SPMV_HOST_OK;
    } catch (matrix::matrix_error const & e) {
        g_error = e.what();
        return SPMV_HOST_ERR_MATRIX;
    } catch (std::bad_alloc const & e) {
        g_error = e.what();
        return SPMV_HOST_ERR_SYSTEM;
    } catch (std::system_error const & e) {
        g_error = e.what();
        return SPMV_HOST_ERR_SYSTEM;
    } catch (std::exception const & e) {
        g_error = e.what();
        return SPMV_HOST_ERR_MATRIX;
    }
}

int invalid(char const * what)
{
    g_error = what;
    return SPMV_HOST_ERR_INVALID;
}

matrix_market::Matrix load_mm(std::string const & path, unsigned flags, bool & expanded)
{
    std::ostringstream log;
    matrix_market::Matrix mm = matrix_market::load_matrix(path, log, false);
    expanded = false;
    if ((flags & SPMV_HOST_EXPAND_SYMMETRIC) && mm.symmetry() != matrix_market::Symmetry::general) {
        expanded = true;
        return matrix_market::expand_symmetry(mm);
    }
    return mm;
}

// a generated matrix with no reordering suffix goes straight to CSR (no coordinate intermediate)
bool plain_spec(std::string const & path)
{
    return synthetic::is_spec(path) && path.find("__RCM") == std::string::npos && path.find("__GP") == std::string::npos;
}

csr_matrix::Matrix slice(csr_matrix::Matrix const & A, long long rb, long long re)
{
    std::size_t const k0 = (std::size_t) A.row_ptr[(std::size_t) rb], k1 = (std::size_t) A.row_ptr[(std::size_t) re];
    csr_matrix::size_array_type p((std::size_t) (re - rb) + 1);
    for (long long r = rb; r <= re; ++r)
        p[(std::size_t) (r - rb)] = A.row_ptr[(std::size_t) r] - (csr_matrix::size_type) k0;
    csr_matrix::index_array_type c(A.column_index.begin() + (std::ptrdiff_t) k0, A.column_index.begin() + (std::ptrdiff_t) k1);
    csr_matrix::value_array_type v(A.value.begin() + (std::ptrdiff_t) k0, A.value.begin() + (std::ptrdiff_t) k1);
    return csr_matrix::Matrix((csr_matrix::index_type) (re - rb), A.columns, (csr_matrix::size_type) (k1 - k0), 1,
                              std::move(p), std::move(c), std::move(v));
}

} // namespace

extern "C" {

const char * spmv_host_last_error(void) { return g_error.c_str(); }

int spmv_host_load(const char * path, int format, unsigned flags, spmv_host_matrix ** out)
{
    if (!out)
        return invalid("out is null");
    *out = nullptr;
    if (!path)
        return invalid("path is null");
    if (format < SPMV_HOST_FORMAT_CSR || format > SPMV_HOST_FORMAT_HYBRID)
        return invalid("unknown format");
    if (flags & ~SPMV_HOST_EXPAND_SYMMETRIC)
        return invalid("unknown flag bits");
    spmv_host_matrix * m = new (std::nothrow) spmv_host_matrix;
    if (!m)
        return invalid("allocation failed");
    m->format = format;
    int const rc = guarded([&] {
        std::string const p = path;
        // (a stored triangle that is to be expanded takes the loader's way: generate -> mirror -> convert)
        if (format == SPMV_HOST_FORMAT_CSR && plain_spec(p) && !((flags & SPMV_HOST_EXPAND_SYMMETRIC) && synthetic::is_stored_triangle(p))) {
            m->csr = synthetic::generate_csr(p);
            m->rows_total = m->csr.rows;
            return;
        }
        matrix_market::Matrix const mm = load_mm(p, flags, m->expanded);
        m->rows_total = mm.rows();
        switch (format) {
        case SPMV_HOST_FORMAT_CSR: m->csr = csr_matrix::from_matrix_market(mm); break;
        case SPMV_HOST_FORMAT_COO: m->coo = coo_matrix::from_matrix_market(mm); break;
        case SPMV_HOST_FORMAT_ELL: m->ell = ell_matrix::from_matrix_market(mm); break;
        default: m->hybrid = hybrid_matrix::from_matrix_market(mm); break;
        }
    });
    if (rc != SPMV_HOST_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return SPMV_HOST_OK;
}

int spmv_host_load_csr_rows(const char * path, unsigned flags, int64_t row_begin, int64_t row_end, spmv_host_matrix ** out)
{
    if (!out)
        return invalid("out is null");
    *out = nullptr;
    if (!path)
        return invalid("path is null");
    if (flags & ~SPMV_HOST_EXPAND_SYMMETRIC)
        return invalid("unknown flag bits");
    if (row_begin < 0 || row_end < row_begin)
        return invalid("bad row range");
    spmv_host_matrix * m = new (std::nothrow) spmv_host_matrix;
    if (!m)
        return invalid("allocation failed");
    m->format = SPMV_HOST_FORMAT_CSR;
    int const rc = guarded([&] {
        std::string const p = path;
        if (plain_spec(p) && !((flags & SPMV_HOST_EXPAND_SYMMETRIC) && synthetic::is_stored_triangle(p))) {
            m->csr = synthetic::generate_csr(p, row_begin, row_end, &m->rows_total);
            return;
        }
        csr_matrix::Matrix const A = csr_matrix::from_matrix_market(load_mm(p, flags, m->expanded));
        if (row_end > A.rows)
            throw matrix::matrix_error("row range out of bounds");
        m->rows_total = A.rows;
        m->csr = slice(A, row_begin, row_end);
    });
    if (rc != SPMV_HOST_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return SPMV_HOST_OK;
}

void spmv_host_matrix_free(spmv_host_matrix * m) { delete m; }

int spmv_host_matrix_info(const spmv_host_matrix * m, int64_t * out, int n)
{
    if (!m || !out || n < 0)
        return invalid("matrix/out null");
    int64_t v[10] = {m->format, 0, 0, 0, 0, 0, 0, m->rows_total, 0, m->expanded ? 1 : 0};
    switch (m->format) {
    case SPMV_HOST_FORMAT_CSR:
        v[1] = m->csr.rows; v[2] = m->csr.columns; v[3] = m->csr.num_entries;
        v[4] = m->csr.row_ptr.empty() ? 0 : m->csr.row_ptr[(std::size_t) m->csr.rows];
        v[8] = (int64_t) m->csr.size();
        break;
    case SPMV_HOST_FORMAT_COO:
        v[1] = m->coo.rows; v[2] = m->coo.columns; v[3] = v[4] = m->coo.num_entries;
        v[8] = (int64_t) m->coo.size();
        break;
    case SPMV_HOST_FORMAT_ELL:
        v[1] = m->ell.rows; v[2] = m->ell.columns; v[3] = m->ell.num_entries;
        v[4] = (int64_t) m->ell.rows * m->ell.row_length; v[5] = m->ell.row_length;
        v[8] = (int64_t) m->ell.size();
        break;
    default:
        v[1] = m->hybrid.rows; v[2] = m->hybrid.columns; v[3] = m->hybrid.num_entries;
        v[4] = m->hybrid.num_ell_entries; v[5] = m->hybrid.ell_row_length; v[6] = m->hybrid.num_coo_entries;
        v[8] = (int64_t) m->hybrid.size();
        break;
    }
    for (int i = 0; i < n && i < 10; ++i)
        out[i] = v[i];
    return SPMV_HOST_OK;
}

const void * spmv_host_matrix_array(const spmv_host_matrix * m, int which)
{
    if (!m)
        return nullptr;
    switch (m->format) {
    case SPMV_HOST_FORMAT_CSR:
        return which == 0 ? (const void *) m->csr.row_ptr.data() : which == 1 ? (const void *) m->csr.column_index.data()
            : which == 2 ? (const void *) m->csr.value.data() : nullptr;
    case SPMV_HOST_FORMAT_COO:
        return which == 0 ? (const void *) m->coo.row_index.data() : which == 1 ? (const void *) m->coo.column_index.data()
            : which == 2 ? (const void *) m->coo.value.data() : nullptr;
    case SPMV_HOST_FORMAT_ELL:
        return which == 1 ? (const void *) m->ell.column_index.data() : which == 2 ? (const void *) m->ell.value.data() : nullptr;
    case SPMV_HOST_FORMAT_HYBRID:
        switch (which) {
        case 1: return m->hybrid.ell_column_index.data();
        case 2: return m->hybrid.ell_value.data();
        case 3: return m->hybrid.coo_row_index.data();
        case 4: return m->hybrid.coo_column_index.data();
        case 5: return m->hybrid.coo_value.data();
        default: return nullptr;
        }
    }
    return nullptr;
}

} // extern "C"
