#include "spmv-kernels.hpp"

#include "../matrix/matrix-error.hpp"
#include "../matrix/matrix-market.hpp"
#include "../matrix/matrix-reorder.hpp"
#include "../matrix/synthetic.hpp"

#include "spmv_hip_tuning.h" // (spmv_hip.h + the CSR algorithm choice and ctx_info of the CLI)

#include <chrono>
#include <ostream>
#include <sstream>
#include <system_error>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// Load + optional symmetry expansion, with the reference's error wrapping
// (src/kernels/csr-spmv.cpp:37-45): "<path>: <what>".
template <typename F>
void guarded_init(std::string const & path, F && body)
{
    try {
        body();
    } catch (matrix::matrix_error const & e) {
        throw kernel_error(path + ": " + e.what());
    } catch (std::system_error const & e) {
        throw kernel_error(path + ": " + e.what());
    } catch (std::bad_alloc const & e) {
        throw kernel_error(path + ": " + e.what());
    }
}

matrix_market::Matrix load(std::string const & path, SpmvOptions const & opt, std::ostream & o, bool verbose)
{
    matrix_market::Matrix mm = matrix_market::load_matrix(path, o, verbose);
    if (opt.expand_symmetric && mm.symmetry() != matrix_market::Symmetry::general) {
        if (verbose)
            o << "Expanding " << mm.num_entries() << " stored entries of a symmetric matrix\n";
        return matrix_market::expand_symmetry(mm);
    }
    return mm;
}

// CSR straight from a generator spec (no coordinate intermediate: a third of the memory at 450 M
// entries); files and reordered specs go through the loader and the converter like in the reference
csr_matrix::Matrix load_csr(std::string const & path, SpmvOptions const & opt, std::ostream & o, bool verbose)
{
    if (synthetic::is_spec(path) && path.find("__RCM") == std::string::npos && path.find("__GP") == std::string::npos &&
        !(opt.expand_symmetric && synthetic::is_stored_triangle(path))) {
        if (verbose)
            o << "Generating matrix " << path << '\n';
        return synthetic::generate_csr(path);
    }
    return csr_matrix::from_matrix_market(load(path, opt, o, verbose));
}

std::ostream & print_common(std::ostream & o, std::string const & name, std::string const & path,
                            char const * format, long long rows, long long columns, long long nonzeros,
                            std::size_t matrix_size)
{
    // (additive, only for a path with a reordering suffix: which order the rows got -- "__GP<n>" without METIS changes nothing,
    // "__GPX<n>" names this build's own partitioner, so that neither is taken for a METIS ordering)
    std::string const reordering = matrix_market::reordering_of(path);
    o << "{\n"
      << "\"name\": \"" << name << "\",\n"
      << "\"matrix_path\": \"" << path << "\",\n";
    if (!reordering.empty())
        o << "\"reordering\": \"" << reordering << "\",\n";
    return o << "\"matrix_format\": \"" << format << "\",\n"
             << "\"rows\": " << rows << ",\n"
             << "\"columns\": " << columns << ",\n"
             << "\"nonzeros\": " << nonzeros << ",\n"
             << "\"matrix_size\": " << matrix_size << ",\n"
             << "\"x_size\": " << sizeof(double) * (std::size_t) columns << ",\n"
             << "\"y_size\": " << sizeof(double) * (std::size_t) rows;
}

bool is_master()
{
#ifdef _OPENMP
    return omp_get_thread_num() == 0;
#else
    return true;
#endif
}

[[noreturn]] void no_reference_string()
{
    // as the reference's MKL kernel does for a kernel it cannot trace (mkl-csr-spmv.cpp:74-81)
    throw kernel_error("Not implemented");
}

// ------------------------------------------------------------------------------------
// CPU kernels
// ------------------------------------------------------------------------------------
class csr_spmv_kernel : public Kernel
{
public:
    csr_spmv_kernel(std::string path, SpmvOptions opt) : matrix_path(std::move(path)), options(opt) {}

    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        guarded_init(matrix_path, [&] {
            A = load_csr(matrix_path, options, o, verbose);
            x = csr_matrix::value_array_type((std::size_t) A.columns, 1.0);
            y = csr_matrix::value_array_type((std::size_t) A.rows, 0.0);
        });
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const &) override { csr_matrix::spmv(A, x, y); }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override { no_reference_string(); }
    std::string name() const override { return "csr-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        return print_common(o, name(), matrix_path, "csr", A.rows, A.columns, A.num_entries, A.size()) << "\n}";
    }
    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * A.row_ptr[(std::size_t) A.rows] + 4.0 * (A.rows + 1.0) + 16.0 * A.rows + 8.0 * A.columns; }
    std::vector<double> result() const override { return std::vector<double>(y.begin(), y.end()); }
    std::size_t columns() const override { return x.size(); }
    void set_x(std::vector<double> const & v) override
    {
        if (v.size() != x.size())
            throw kernel_error("set_x: size mismatch");
        std::copy(v.begin(), v.end(), x.begin());
    }

private:
    std::string matrix_path;
    SpmvOptions options;
    csr_matrix::Matrix A;
    csr_matrix::value_array_type x, y;
};

class coo_spmv_kernel : public Kernel
{
public:
    coo_spmv_kernel(std::string path, SpmvOptions opt, bool atomic)
        : matrix_path(std::move(path)), options(opt), atomic(atomic)
    {
    }

    void init(TraceConfig const & trace_config, std::ostream & o, bool verbose) override
    {
        int const num_threads = (int) trace_config.thread_affinities().size();
        guarded_init(matrix_path, [&] {
            A = coo_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x = coo_matrix::value_array_type((std::size_t) A.columns, 1.0);
            y = coo_matrix::value_array_type((std::size_t) A.rows, 0.0);
            if (!atomic) {
                std::size_t n;
                if (__builtin_mul_overflow((std::size_t) num_threads, (std::size_t) A.rows, &n))
                    throw matrix::matrix_error(
                        "Failed to compute COO SpMV: Integer overflow when computing workspace size");
                workspace = coo_matrix::value_array_type(n, 0.0); // zeroed here, never again
            }
        });
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const & trace_config) override
    {
        int const num_threads = (int) trace_config.thread_affinities().size();
        if (atomic)
            coo_matrix::spmv_atomic(num_threads, A, x, y);
        else
            coo_matrix::spmv(num_threads, A, x, y, workspace);
    }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override { no_reference_string(); }
    std::string name() const override { return atomic ? "coo-spmv-atomic" : "coo-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        return print_common(o, name(), matrix_path, "coo", A.rows, A.columns, A.num_entries, A.size()) << "\n}";
    }
    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 16.0 * A.num_entries + 16.0 * A.rows + 8.0 * A.columns; }
    std::vector<double> result() const override { return std::vector<double>(y.begin(), y.end()); }
    std::size_t columns() const override { return x.size(); }
    void set_x(std::vector<double> const & v) override
    {
        if (v.size() != x.size())
            throw kernel_error("set_x: size mismatch");
        std::copy(v.begin(), v.end(), x.begin());
    }

private:
    std::string matrix_path;
    SpmvOptions options;
    bool atomic;
    coo_matrix::Matrix A;
    coo_matrix::value_array_type x, y, workspace;
};

class ell_spmv_kernel : public Kernel
{
public:
    ell_spmv_kernel(std::string path, SpmvOptions opt) : matrix_path(std::move(path)), options(opt) {}

    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        guarded_init(matrix_path, [&] {
            A = ell_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x = ell_matrix::value_array_type((std::size_t) A.columns, 1.0);
            y = ell_matrix::value_array_type((std::size_t) A.rows, 0.0);
        });
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const &) override { ell_matrix::spmv(A, x, y); }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override { no_reference_string(); }
    std::string name() const override { return "ell-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        return print_common(o, name(), matrix_path, "ell", A.rows, A.columns, A.num_entries, A.size()) << "\n}";
    }
    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * (double) A.rows * A.row_length + 16.0 * A.rows + 8.0 * A.columns; }
    std::vector<double> result() const override { return std::vector<double>(y.begin(), y.end()); }
    std::size_t columns() const override { return x.size(); }
    void set_x(std::vector<double> const & v) override
    {
        if (v.size() != x.size())
            throw kernel_error("set_x: size mismatch");
        std::copy(v.begin(), v.end(), x.begin());
    }

private:
    std::string matrix_path;
    SpmvOptions options;
    ell_matrix::Matrix A;
    ell_matrix::value_array_type x, y;
};

class hybrid_spmv_kernel : public Kernel
{
public:
    hybrid_spmv_kernel(std::string path, SpmvOptions opt) : matrix_path(std::move(path)), options(opt) {}

    void init(TraceConfig const & trace_config, std::ostream & o, bool verbose) override
    {
        int const num_threads = (int) trace_config.thread_affinities().size();
        guarded_init(matrix_path, [&] {
            if (verbose)
                o << "Converting matrix to hybrid format" << std::endl;
            A = hybrid_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x = hybrid_matrix::value_array_type((std::size_t) A.columns, 1.0);
            y = hybrid_matrix::value_array_type((std::size_t) A.rows, 0.0);
            std::size_t n;
            if (__builtin_mul_overflow((std::size_t) num_threads, (std::size_t) A.rows, &n))
                throw matrix::matrix_error(
                    "Failed to compute HYBRID SpMV: Integer overflow when computing workspace size");
            workspace = hybrid_matrix::value_array_type(n, 0.0);
        });
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const & trace_config) override
    {
        hybrid_matrix::spmv((int) trace_config.thread_affinities().size(), A, x, y, workspace);
    }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override { no_reference_string(); }
    std::string name() const override { return "hybrid-spmv"; }
    std::ostream & print(std::ostream & o) const override { return print_hybrid(o, name(), matrix_path, A) << "\n}"; }
    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * (double) A.num_ell_entries + 16.0 * A.num_coo_entries + 16.0 * A.rows + 8.0 * A.columns; }
    std::vector<double> result() const override { return std::vector<double>(y.begin(), y.end()); }
    std::size_t columns() const override { return x.size(); }
    void set_x(std::vector<double> const & v) override
    {
        if (v.size() != x.size())
            throw kernel_error("set_x: size mismatch");
        std::copy(v.begin(), v.end(), x.begin());
    }

    // the reference's object (src/kernels/hybrid-spmv.cpp:111-131) without its stray second comma
    // after "matrix_size", which makes the reference's own output invalid JSON
    static std::ostream & print_hybrid(std::ostream & o, std::string const & name, std::string const & path,
                                       hybrid_matrix::Matrix const & A)
    {
        return print_common(o, name, path, "hybrid", A.rows, A.columns, A.num_entries, A.size())
            << ",\n\"ell_row_length\": " << A.ell_row_length << ",\n\"num_ell_entries\": " << A.num_ell_entries
            << ",\n\"num_coo_entries\": " << A.num_coo_entries;
    }

private:
    std::string matrix_path;
    SpmvOptions options;
    hybrid_matrix::Matrix A;
    hybrid_matrix::value_array_type x, y, workspace;
};

// ------------------------------------------------------------------------------------
// HIP kernels: the host object keeps A, x, y like its CPU sibling; the device copies live in
// a spmv_hip_ctx.  run() is executed by the master thread only and returns after the device
// is idle, so the harness's closing barrier really brackets the multiplication.
// ------------------------------------------------------------------------------------
class hip_kernel_base : public Kernel
{
public:
    hip_kernel_base(std::string path, SpmvOptions opt) : matrix_path(std::move(path)), options(opt) {}
    ~hip_kernel_base() override
    {
        if (ctx)
            spmv_hip_destroy(ctx);
    }

    void prepare(TraceConfig const &) override
    {
        // the reference moves pages between NUMA domains here; this kernel puts x and the
        // starting y on the device instead (master only, the others wait at the barrier)
        // every thread reaches the barrier even if the master's upload fails; the failure is
        // then reported by all of them
        if (is_master()) {
            prepare_error.clear();
            try {
                check(spmv_hip_set_x(ctx, x.data()), "set_x");
                check(spmv_hip_set_y(ctx, y.data()), "set_y");
            } catch (kernel_error const & e) {
                prepare_error = e.what();
            }
        }
#pragma omp barrier
        if (!prepare_error.empty())
            throw kernel_error(prepare_error);
    }

    void run(TraceConfig const &) override
    {
        if (is_master()) {
            check(spmv_hip_run(ctx), "run");
            check(spmv_hip_sync(ctx), "sync");
            std::uint64_t ns = 0, gns = 0;
            if (spmv_hip_last_run_times(ctx, &ns, &gns) == SPMV_HIP_OK) {
                device_ns = ns;
                gather_ns = gns;
            }
        }
    }

    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override { no_reference_string(); }
    std::uint64_t last_device_ns() const override { return device_ns; }
    void flush_caches() override { check(spmv_hip_flush_caches(ctx), "flush_caches"); }

    std::vector<double> result() const override
    {
        std::vector<double> out(y.size());
        if (!out.empty())
            check(spmv_hip_get_y(ctx, out.data()), "get_y");
        return out;
    }

    std::size_t columns() const override { return x.size(); }
    void set_x(std::vector<double> const & v) override
    {
        if (v.size() != x.size())
            throw kernel_error("set_x: size mismatch");
        std::copy(v.begin(), v.end(), x.begin());
    }

protected:
    void create_context()
    {
        // --gpus G: one context over G devices (row blocks by the reference's static rule, one in-place
        // RCCL all-gather of y per run); otherwise one device
        int rc = options.num_gpus > 0 ? spmv_hip_create_multi(&ctx, options.num_gpus, options.hip_flags)
                                      : spmv_hip_create(&ctx, options.device, options.hip_flags);
        if (rc != SPMV_HIP_OK)
            throw kernel_error(matrix_path + ": " + spmv_hip_strerror(rc) + ": " + spmv_hip_last_error());
        check(spmv_hip_set_csr_algorithm(ctx, options.csr_algorithm, options.csr_lanes_per_row), "set_csr_algorithm");
    }

    void check(int rc, char const * what) const
    {
        if (rc != SPMV_HIP_OK) {
            std::string detail = spmv_hip_last_error();
            throw kernel_error(matrix_path + ": " + spmv_hip_strerror(rc) + (detail.empty() ? "" : ": " + detail) +
                               " (" + what + ")");
        }
    }

    std::ostream & print_device(std::ostream & o) const
    {
        std::int64_t info[17] = {0};
        spmv_hip_ctx_info(ctx, info, 17);
        static char const * const algo[] = {"auto", "scalar", "vector", "adaptive", "wavetile"};
        o << ",\n\"device\": {\"backend\": \"hip\", \"index\": " << options.device;
        if (info[0] == 1)
            o << ", \"csr_algorithm\": \"" << algo[info[4] >= 0 && info[4] <= 4 ? info[4] : 0]
              << "\", \"lanes_per_row\": " << info[5] << ", \"tiles\": " << info[7] << ", \"long_rows\": " << info[8]
              << ", \"tiles_16bit_columns\": " << info[10] << ", \"tiles_shifted\": " << info[11]
              << ", \"tiles_x_window\": " << info[12] << ", \"tiles_block_window\": " << info[13]
              << ", \"tiles_column_panels\": " << info[14];
        o << ", \"workgroups\": " << info[6] << ", \"device_bytes\": " << info[9]
          << ", \"streamed_bytes_per_run\": " << info[15] << ", \"gpus\": " << info[16]
          << ", \"last_run_device_ns\": " << device_ns;
        if (info[16] > 1 || options.num_gpus > 0)
            o << ", \"last_run_all_gather_ns\": " << gather_ns << ", \"partition\": \""
              << ((options.hip_flags & SPMV_HIP_FLAG_BALANCE_ENTRIES) ? "row blocks of equal stored entries over " : "static chunks of ceil(rows/G) rows over ")
              << info[16] << " devices, x replicated, 1 in-place all-gather(y) per run ("
              << ((options.hip_flags & SPMV_HIP_FLAG_FUSED_PEER_STORE) ? "row sums stored into every device's y by the multiply itself"
                  : (options.hip_flags & SPMV_HIP_FLAG_PEER_GATHER) ? "remote stores over xGMI" : "RCCL")
              << (((options.hip_flags & SPMV_HIP_FLAG_PIPELINE_GATHER) && !(options.hip_flags & SPMV_HIP_FLAG_FUSED_PEER_STORE))
                      ? "; back-to-back runs pipelined: gather k beside multiply k + 1" : "") << ")\"";
        if (init_load_seconds > 0.0 || init_upload_seconds > 0.0)
            o << ", \"init_seconds\": {\"load_and_convert\": " << init_load_seconds << ", \"upload_and_plan\": " << init_upload_seconds << "}";
        o << "}";
        return o;
    }

    std::string matrix_path;
    SpmvOptions options;
    spmv_hip_ctx * ctx = nullptr;
    aligned_vector<double> x, y;
    std::uint64_t device_ns = 0, gather_ns = 0;
    double init_load_seconds = 0.0, init_upload_seconds = 0.0; // wall time of Kernel::init: file -> host arrays -> device + plan
    std::string prepare_error;
};

class hip_csr_spmv_kernel : public hip_kernel_base
{
public:
    using hip_kernel_base::hip_kernel_base;
    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        auto const t0 = std::chrono::steady_clock::now();
        guarded_init(matrix_path, [&] {
            A = load_csr(matrix_path, options, o, verbose);
            x.assign((std::size_t) A.columns, 1.0);
            y.assign((std::size_t) A.rows, 0.0);
        });
        auto const t1 = std::chrono::steady_clock::now();
        create_context();
        check(spmv_hip_upload_csr(ctx, A.rows, A.columns, A.row_ptr[(std::size_t) A.rows], A.row_ptr.data(),
                                  A.column_index.data(), A.value.data()), "upload_csr");
        init_load_seconds = std::chrono::duration<double>(t1 - t0).count();
        init_upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    }
    std::string name() const override { return "hip-csr-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        print_common(o, name(), matrix_path, "csr", A.rows, A.columns, A.num_entries, A.size());
        return print_device(o) << "\n}";
    }

    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * A.row_ptr[(std::size_t) A.rows] + 4.0 * (A.rows + 1.0) + 16.0 * A.rows + 8.0 * A.columns; }

private:
    csr_matrix::Matrix A;
};

class hip_coo_spmv_kernel : public hip_kernel_base
{
public:
    using hip_kernel_base::hip_kernel_base;
    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        guarded_init(matrix_path, [&] {
            A = coo_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x.assign((std::size_t) A.columns, 1.0);
            y.assign((std::size_t) A.rows, 0.0);
        });
        create_context();
        check(spmv_hip_upload_coo(ctx, A.rows, A.columns, A.num_entries, A.row_index.data(),
                                  A.column_index.data(), A.value.data()), "upload_coo");
    }
    std::string name() const override { return "hip-coo-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        print_common(o, name(), matrix_path, "coo", A.rows, A.columns, A.num_entries, A.size());
        return print_device(o) << "\n}";
    }

    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 16.0 * A.num_entries + 16.0 * A.rows + 8.0 * A.columns; }

private:
    coo_matrix::Matrix A;
};

class hip_ell_spmv_kernel : public hip_kernel_base
{
public:
    using hip_kernel_base::hip_kernel_base;
    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        guarded_init(matrix_path, [&] {
            A = ell_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x.assign((std::size_t) A.columns, 1.0);
            y.assign((std::size_t) A.rows, 0.0);
        });
        create_context();
        check(spmv_hip_upload_ell(ctx, A.rows, A.columns, A.row_length, A.column_index.data(), A.value.data()),
              "upload_ell");
    }
    std::string name() const override { return "hip-ell-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        print_common(o, name(), matrix_path, "ell", A.rows, A.columns, A.num_entries, A.size());
        return print_device(o) << "\n}";
    }

    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * (double) A.rows * A.row_length + 16.0 * A.rows + 8.0 * A.columns; }

private:
    ell_matrix::Matrix A;
};

class hip_hybrid_spmv_kernel : public hip_kernel_base
{
public:
    using hip_kernel_base::hip_kernel_base;
    void init(TraceConfig const &, std::ostream & o, bool verbose) override
    {
        guarded_init(matrix_path, [&] {
            A = hybrid_matrix::from_matrix_market(load(matrix_path, options, o, verbose));
            x.assign((std::size_t) A.columns, 1.0);
            y.assign((std::size_t) A.rows, 0.0);
        });
        create_context();
        check(spmv_hip_upload_hybrid(ctx, A.rows, A.columns, A.ell_row_length, A.ell_column_index.data(),
                                     A.ell_value.data(), A.num_coo_entries, A.coo_row_index.data(),
                                     A.coo_column_index.data(), A.coo_value.data()), "upload_hybrid");
    }
    std::string name() const override { return "hip-hybrid-spmv"; }
    std::ostream & print(std::ostream & o) const override
    {
        hybrid_spmv_kernel::print_hybrid(o, name(), matrix_path, A);
        return print_device(o) << "\n}";
    }

    double flops_per_run() const override { return 2.0 * A.num_entries; }
    double bytes_per_run() const override { return 12.0 * (double) A.num_ell_entries + 16.0 * A.num_coo_entries + 16.0 * A.rows + 8.0 * A.columns; }

private:
    hybrid_matrix::Matrix A;
};

} // namespace

std::unique_ptr<Kernel> make_spmv_kernel(SpmvFormat format, bool hip, std::string const & path,
                                         SpmvOptions const & opt)
{
    switch (format) {
    case SpmvFormat::csr:
        if (hip) return std::make_unique<hip_csr_spmv_kernel>(path, opt);
        return std::make_unique<csr_spmv_kernel>(path, opt);
    case SpmvFormat::coo:
        if (hip) return std::make_unique<hip_coo_spmv_kernel>(path, opt);
        return std::make_unique<coo_spmv_kernel>(path, opt, false);
    case SpmvFormat::coo_atomic:
        if (hip) return std::make_unique<hip_coo_spmv_kernel>(path, opt); // the GPU COO kernel IS the atomic form
        return std::make_unique<coo_spmv_kernel>(path, opt, true);
    case SpmvFormat::ell:
        if (hip) return std::make_unique<hip_ell_spmv_kernel>(path, opt);
        return std::make_unique<ell_spmv_kernel>(path, opt);
    case SpmvFormat::hybrid:
        if (hip) return std::make_unique<hip_hybrid_spmv_kernel>(path, opt);
        return std::make_unique<hybrid_spmv_kernel>(path, opt);
    }
    throw kernel_error("unknown kernel type");
}
