#include "triad-kernel.hpp"

#include "../matrix/aligned-vector.hpp"

#include "spmv_hip_plan.h"

#include <hip/hip_runtime_api.h>

#include <ostream>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

std::ostream & print_triad(std::ostream & o, std::string const & name, std::size_t n)
{
    // name and the (quoted) entry count as the reference prints them (src/kernels/triad.cpp:87-95),
    // plus the bytes one run moves: 24 per element (src/main.cpp:184)
    return o << "{\n"
             << "\"name\": \"" << name << "\",\n"
             << "\"num_entries\": \"" << n << "\",\n"
             << "\"bytes_per_run\": " << 24 * n << "\n}";
}

class triad_kernel : public Kernel
{
public:
    explicit triad_kernel(std::size_t n) : n(n) {}
    void init(TraceConfig const &, std::ostream &, bool) override
    {
        a.assign(n, 1.0);
        b.assign(n, 0.0);
        c.assign(n, 0.0);
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const &) override
    {
        double const d = 3.1;
        double * const pa = a.data();
        double const * const pb = b.data();
        double const * const pc = c.data();
#pragma omp for
        for (long long i = 0; i < (long long) n; ++i)
            pa[i] = pb[i] + d * pc[i];
    }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override
    {
        throw kernel_error("Not implemented");
    }
    std::string name() const override { return "triad"; }
    std::ostream & print(std::ostream & o) const override { return print_triad(o, name(), n); }
    std::vector<double> result() const override { return std::vector<double>(a.begin(), a.end()); }
    void set_x(std::vector<double> const &) override {}

private:
    std::size_t n;
    aligned_vector<double> a, b, c;
};

class hip_triad_kernel : public Kernel
{
public:
    hip_triad_kernel(std::size_t n, int device) : n(n), device(device) {}
    ~hip_triad_kernel() override
    {
        for (double * p : {da, db, dc})
            if (p)
                (void) hipFree(p);
        if (e0) (void) hipEventDestroy(e0);
        if (e1) (void) hipEventDestroy(e1);
    }
    void init(TraceConfig const &, std::ostream &, bool) override
    {
        int count = 0;
        spmv_hip_device_count(&count);
        if (count < 1)
            throw kernel_error("no HIP device available (the GPU kernels have no CPU fallback)");
        hip(hipSetDevice(device), "hipSetDevice");
        hip(hipMalloc((void **) &da, bytes()), "hipMalloc");
        hip(hipMalloc((void **) &db, bytes()), "hipMalloc");
        hip(hipMalloc((void **) &dc, bytes()), "hipMalloc");
        hip(hipMemset(da, 0, bytes()), "hipMemset");
        hip(hipMemset(db, 0, bytes()), "hipMemset");
        hip(hipMemset(dc, 0, bytes()), "hipMemset");
        hip(hipEventCreate(&e0), "hipEventCreate");
        hip(hipEventCreate(&e1), "hipEventCreate");
    }
    void prepare(TraceConfig const &) override {}
    void run(TraceConfig const &) override
    {
#ifdef _OPENMP
        if (omp_get_thread_num() != 0)
            return;
#endif
        hip(hipEventRecord(e0, nullptr), "hipEventRecord");
        int rc = spmv_hip_triad((int64_t) n, da, db, dc, 3.1, nullptr);
        if (rc != SPMV_HIP_OK)
            throw kernel_error(std::string(spmv_hip_strerror(rc)) + ": " + spmv_hip_last_error());
        hip(hipEventRecord(e1, nullptr), "hipEventRecord");
        hip(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0;
        hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        device_ns = (std::uint64_t) (ms * 1e6 + 0.5);
    }
    MemoryReferenceString memory_reference_string(TraceConfig const &, int, int) const override
    {
        throw kernel_error("Not implemented");
    }
    std::string name() const override { return "hip-triad"; }
    std::ostream & print(std::ostream & o) const override { return print_triad(o, name(), n); }
    std::uint64_t last_device_ns() const override { return device_ns; }
    std::vector<double> result() const override
    {
        std::vector<double> out(n);
        if (n)
            hip(hipMemcpy(out.data(), da, bytes(), hipMemcpyDeviceToHost), "hipMemcpy");
        return out;
    }
    void set_x(std::vector<double> const &) override {}

private:
    std::size_t bytes() const { return (n ? n : 1) * sizeof(double); }
    static void hip(hipError_t e, char const * what)
    {
        if (e != hipSuccess)
            throw kernel_error(std::string(what) + ": " + hipGetErrorString(e));
    }
    std::size_t n;
    int device;
    double *da = nullptr, *db = nullptr, *dc = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::uint64_t device_ns = 0;
};

} // namespace

std::unique_ptr<Kernel> make_triad_kernel(std::size_t num_entries, bool hip, int device)
{
    if (hip)
        return std::make_unique<hip_triad_kernel>(num_entries, device);
    return std::make_unique<triad_kernel>(num_entries);
}
