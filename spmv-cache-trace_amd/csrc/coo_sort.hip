// coo_sort.hip -- stable sort of COO triplets by row index, on the device (upload time only).
//
// The reference keeps COO entries in file order, and SuiteSparse files are column-major: in that
// order consecutive entries hit 64 different rows, every lane of the COO kernel issues its own fp64
// atomic, and the kernel runs at the atomic rate (0.5 TB/s measured) instead of the HBM rate.
// Sorting by row once, STABLY (entries of a row keep their file order, so each row is still summed in
// file order), puts the same triplets into the layout the kernel is fast on: runs of equal rows that
// a wave adds up before issuing one atomic per run.  rocPRIM's radix sort (through hipCUB) does the
// sorting; this is library plumbing, not a hot-path kernel.
#include "spmv_hip.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>

namespace {

__global__ __launch_bounds__(256) void iota_kernel(int n, int32_t * v)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        v[i] = (int32_t) i;
}

__global__ __launch_bounds__(256) void gather_kernel(int n, const int32_t * __restrict__ perm,
                                                     const int32_t * __restrict__ col_in, const double * __restrict__ val_in,
                                                     int32_t * __restrict__ col_out, double * __restrict__ val_out)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int32_t k = perm[i];
        col_out[i] = col_in[k];
        val_out[i] = val_in[k];
    }
}

} // namespace

// defined in spmv_hip.hip
int spmv_hip_internal_fail_hip(hipError_t e, const char * call);

extern "C" int spmv_hip_coo_sort_by_row(int32_t rows, int32_t nnz, int32_t * d_row, int32_t * d_col, double * d_val,
                                        void * stream)
{
    if (rows < 0 || nnz < 0)
        return SPMV_HIP_ERR_INVALID;
    if (nnz < 2 || rows < 2)
        return SPMV_HIP_OK;
    if (!d_row || !d_col || !d_val)
        return SPMV_HIP_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int bits = 1;
    while (bits < 31 && (1LL << bits) < (long long) rows)
        ++bits;

    int32_t *keys_out = nullptr, *idx_in = nullptr, *idx_out = nullptr, *col_out = nullptr;
    double * val_out = nullptr;
    void * temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = hipSuccess;
    auto alloc = [&](void ** p, size_t bytes) {
        if (e == hipSuccess)
            e = hipMalloc(p, bytes);
    };
    alloc((void **) &keys_out, (size_t) nnz * 4);
    alloc((void **) &idx_in, (size_t) nnz * 4);
    alloc((void **) &idx_out, (size_t) nnz * 4);
    alloc((void **) &col_out, (size_t) nnz * 4);
    alloc((void **) &val_out, (size_t) nnz * 8);
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_row, keys_out, idx_in, idx_out, nnz, 0, bits, s);
    alloc(&temp, temp_bytes ? temp_bytes : 16);
    if (e == hipSuccess) {
        const int grid = (int) std::min<long long>(((long long) nnz + 255) / 256, 4096);
        hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, s, nnz, idx_in);
        e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, d_row, keys_out, idx_in, idx_out, nnz, 0, bits, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, s, nnz, idx_out, d_col, d_val, col_out, val_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(d_row, keys_out, (size_t) nnz * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_col, col_out, (size_t) nnz * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_val, val_out, (size_t) nnz * 8, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    for (void * p : {(void *) keys_out, (void *) idx_in, (void *) idx_out, (void *) col_out, (void *) val_out, temp})
        if (p)
            (void) hipFree(p);
    return e == hipSuccess ? SPMV_HIP_OK : spmv_hip_internal_fail_hip(e, "COO sort by row");
}


// Exclusive prefix sum of n int32 values (column panels: counts -> virtual row pointers).
extern "C" int spmv_hip_internal_exclusive_scan_i32(const int32_t * d_in, int32_t * d_out, long long n, hipStream_t s)
{
    if (n <= 0)
        return SPMV_HIP_OK;
    void * temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_in, d_out, (int) n, s);
    if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, d_in, d_out, (int) n, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (temp)
        (void) hipFree(temp);
    return e == hipSuccess ? SPMV_HIP_OK : SPMV_HIP_ERR_HIP;
}
