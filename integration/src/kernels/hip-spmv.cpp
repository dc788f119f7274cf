#include "hip-spmv.hpp"

#include "matrix/matrix-error.hpp"
#include "matrix/matrix-market.hpp"

#include "spmv_hip.h"

#ifdef USE_OPENMP
#include <omp.h>
#endif

#include <ostream>
#include <system_error>

namespace hip_spmv
{

namespace
{

/* prepare() and run() are called by every thread of the team (profile-kernel.cpp:262-264, :160); one of
 * them talks to the device. */
bool master_thread()
{
#ifdef USE_OPENMP
    return omp_get_thread_num() == 0;
#else
    return true;
#endif
}

}

csr_format::matrix_type csr_format::convert(matrix_market::Matrix const & mm)
{
    return csr_matrix::from_matrix_market(mm);
}

int csr_format::upload(spmv_hip_ctx * ctx, matrix_type const & A)
{
    return spmv_hip_upload_csr(
        ctx, A.rows, A.columns, A.row_ptr[A.rows],
        A.row_ptr.data(), A.column_index.data(), A.value.data());
}

coo_format::matrix_type coo_format::convert(matrix_market::Matrix const & mm)
{
    return coo_matrix::from_matrix_market(mm);
}

int coo_format::upload(spmv_hip_ctx * ctx, matrix_type const & A)
{
    return spmv_hip_upload_coo(
        ctx, A.rows, A.columns, A.num_entries,
        A.row_index.data(), A.column_index.data(), A.value.data());
}

ell_format::matrix_type ell_format::convert(matrix_market::Matrix const & mm)
{
    return ell_matrix::from_matrix_market(mm);
}

int ell_format::upload(spmv_hip_ctx * ctx, matrix_type const & A)
{
    return spmv_hip_upload_ell(
        ctx, A.rows, A.columns, A.row_length,
        A.column_index.data(), A.value.data());
}

template <class Format>
kernel<Format>::kernel(std::string const & matrix_path)
    : Kernel()
    , matrix_path(matrix_path)
    , ctx(nullptr)
{
}

template <class Format>
kernel<Format>::~kernel()
{
    if (ctx)
        spmv_hip_destroy(ctx);
}

/* The library never throws across its C boundary: a negative code and a detail string per thread. */
template <class Format>
void kernel<Format>::check(int code) const
{
    if (code == SPMV_HIP_OK)
        return;
    std::string what = spmv_hip_strerror(code);
    char const * detail = spmv_hip_last_error();
    if (detail && *detail && std::string(detail).find(what) == 0)
        what = detail;  /* the detail repeats the code's text and adds to it */
    else if (detail && *detail)
        what += std::string(": ") + detail;
    throw kernel_error(matrix_path + ": " + what);
}

template <class Format>
void kernel<Format>::init(TraceConfig const &, std::ostream & o, bool verbose)
{
    try {
        matrix_market::Matrix mm = matrix_market::load_matrix(matrix_path, o, verbose);
        A = Format::convert(mm);
    } catch (matrix::matrix_error const & e) {
        throw kernel_error(matrix_path + ": " + e.what());
    } catch (std::system_error const & e) {
        throw kernel_error(matrix_path + ": " + e.what());
    }
    x.assign(A.columns, 1.0);
    y.assign(A.rows, 0.0);
    check(spmv_hip_create(&ctx, 0, 0));
    check(Format::upload(ctx, A));
}

/* In place of the NUMA page distribution: x and the starting y go to the device. */
template <class Format>
void kernel<Format>::prepare(TraceConfig const &)
{
    if (master_thread()) {
        check(spmv_hip_set_x(ctx, x.data()));
        check(spmv_hip_set_y(ctx, y.data()));
    }
}

/* y += A*x on the device.  Returns when the device is idle: the caller's next statement is the barrier
 * in front of the second time stamp (profile-kernel.cpp:161-165). */
template <class Format>
void kernel<Format>::run(TraceConfig const &)
{
    if (master_thread()) {
        check(spmv_hip_run(ctx));
        check(spmv_hip_sync(ctx));
    }
}

/* A device kernel has no host memory reference string (as mkl-csr-spmv.cpp:74-81). */
template <class Format>
replacement::MemoryReferenceString kernel<Format>::memory_reference_string(
    TraceConfig const &, int, int) const
{
    throw kernel_error("Not implemented");
}

template <class Format>
std::string kernel<Format>::name() const
{
    return Format::kernel_name();
}

template <class Format>
std::vector<double> kernel<Format>::result() const
{
    std::vector<double> out(y.size());
    check(spmv_hip_get_y(ctx, out.data()));
    return out;
}

namespace
{

template <class T>
void field(std::ostream & o, char const * key, T const & v, bool last = false)
{
    o << '"' << key << '"' << ": " << v << (last ? "\n" : ",\n");
}

std::string quoted(std::string const & s)
{
    return '"' + s + '"';
}

}

/* The fields of csr-spmv.cpp:97-112 with the same meaning. */
template <class Format>
std::ostream & kernel<Format>::print(std::ostream & o) const
{
    o << "{\n";
    field(o, "name", quoted(name()));
    field(o, "matrix_path", quoted(matrix_path));
    field(o, "matrix_format", quoted(Format::format_name()));
    field(o, "rows", A.rows);
    field(o, "columns", A.columns);
    field(o, "nonzeros", A.num_entries);
    field(o, "matrix_size", A.size());
    field(o, "x_size", sizeof(double) * A.columns);
    field(o, "y_size", sizeof(double) * A.rows, true);
    return o << "}";
}

template class kernel<csr_format>;
template class kernel<coo_format>;
template class kernel<ell_format>;

}
