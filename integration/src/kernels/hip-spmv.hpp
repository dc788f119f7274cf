#ifndef HIP_SPMV_HPP
#define HIP_SPMV_HPP

/*
 * MI355X kernels behind the Kernel interface (kernel.hpp:18-45): one class template, instantiated for the
 * CSR, COO and ELLPACK containers of src/matrix.  The matrix is loaded and converted by the reference's own
 * code, handed once to libspmv_hip.so (include/spmv_hip.h, a C ABI) and multiplied there; x and y stay on
 * the device between runs, as csr_spmv_kernel keeps them in DRAM.
 *
 * Stands where csr_spmv_kernel / coo_spmv_kernel / ell_spmv_kernel stand (csr-spmv.cpp:15-112,
 * coo-spmv.cpp:15-126, ell-spmv.cpp:15-111); selected by --spmv-format hip-csr | hip-coo | hip-ell.
 */

#include "kernel.hpp"
#include "trace-config.hpp"
#include "cache-simulation/replacement.hpp"
#include "matrix/coo-matrix.hpp"
#include "matrix/csr-matrix.hpp"
#include "matrix/ell-matrix.hpp"
#include "matrix/matrix-market.hpp"

#include <iosfwd>
#include <string>
#include <vector>

struct spmv_hip_ctx;

namespace hip_spmv
{

/* What differs between the formats: the container, its converter, the upload call, two names. */
struct csr_format
{
    typedef csr_matrix::Matrix matrix_type;
    static char const * kernel_name() { return "hip-csr-spmv"; }
    static char const * format_name() { return "csr"; }
    static matrix_type convert(matrix_market::Matrix const & mm);
    static int upload(spmv_hip_ctx * ctx, matrix_type const & A);
};

struct coo_format
{
    typedef coo_matrix::Matrix matrix_type;
    static char const * kernel_name() { return "hip-coo-spmv"; }
    static char const * format_name() { return "coo"; }
    static matrix_type convert(matrix_market::Matrix const & mm);
    static int upload(spmv_hip_ctx * ctx, matrix_type const & A);
};

struct ell_format
{
    typedef ell_matrix::Matrix matrix_type;
    static char const * kernel_name() { return "hip-ell-spmv"; }
    static char const * format_name() { return "ell"; }
    static matrix_type convert(matrix_market::Matrix const & mm);
    static int upload(spmv_hip_ctx * ctx, matrix_type const & A);
};

template <class Format>
class kernel : public Kernel
{
public:
    explicit kernel(std::string const & matrix_path);
    ~kernel();

    void init(TraceConfig const & trace_config, std::ostream & o, bool verbose) override;
    void prepare(TraceConfig const & trace_config) override;
    void run(TraceConfig const & trace_config) override;

    replacement::MemoryReferenceString memory_reference_string(
        TraceConfig const & trace_config, int thread, int num_threads) const override;

    std::string name() const override;
    std::ostream & print(std::ostream & o) const override;

    /* for checks: the device's y after the runs so far */
    std::vector<double> result() const;

private:
    void check(int code) const;

    std::string matrix_path;
    typename Format::matrix_type A;
    std::vector<double> x;
    std::vector<double> y;
    spmv_hip_ctx * ctx;
};

extern template class kernel<csr_format>;
extern template class kernel<coo_format>;
extern template class kernel<ell_format>;

}

typedef hip_spmv::kernel<hip_spmv::csr_format> hip_csr_spmv_kernel;
typedef hip_spmv::kernel<hip_spmv::coo_format> hip_coo_spmv_kernel;
typedef hip_spmv::kernel<hip_spmv::ell_format> hip_ell_spmv_kernel;

#endif
