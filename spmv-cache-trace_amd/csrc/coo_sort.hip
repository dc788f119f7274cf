// coo_sort.hip -- stable sort of COO triplets by row index, on the device (upload time only).
//
// The reference keeps COO entries in file order, and SuiteSparse files are column-major: in that
// order consecutive entries hit 64 different rows, every lane of the COO kernel issues its own fp64
// atomic, and the kernel runs at the atomic rate (0.5 TB/s measured) instead of the HBM rate.
// Sorting by row once, STABLY (entries of a row keep their file order, so each row is still summed in
// file order), puts the same triplets into the layout the kernel is fast on: runs of equal rows that
// a wave adds up before issuing one atomic per run.  rocPRIM's radix sort (through hipCUB) does the
// sorting; this is library plumbing, not a hot-path kernel.
#include "spmv_hip_plan.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
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


namespace {

__global__ __launch_bounds__(256) void panel_key_kernel(int n, int width, const int32_t * __restrict__ col,
                                                        int32_t * __restrict__ key, int32_t * __restrict__ idx)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        key[i] = col[i] / width;
        idx[i] = (int32_t) i;
    }
}

// first[k] = position of the first sorted entry with key >= k is derived on the host from these
__global__ __launch_bounds__(256) void panel_first_kernel(int n, const int32_t * __restrict__ key_sorted, int32_t * __restrict__ first)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        if (i == 0 || key_sorted[i] != key_sorted[i - 1])
            first[key_sorted[i]] = (int32_t) i;
}

struct PanelPlace {
    int ufirst[9];      // unpadded start of every panel in the sorted order
    long long start[9]; // padded start in the output arrays
};

__global__ __launch_bounds__(256) void panel_place_kernel(int n, const int32_t * __restrict__ key_sorted,
                                                          const int32_t * __restrict__ idx_sorted,
                                                          const int32_t * __restrict__ row, const int32_t * __restrict__ col,
                                                          const double * __restrict__ val, int32_t * __restrict__ prow,
                                                          int32_t * __restrict__ pcol, double * __restrict__ pval, PanelPlace pp)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int k = key_sorted[i];
        const long long dst = pp.start[k] + (i - pp.ufirst[k]);
        const int32_t src = idx_sorted[i];
        prow[dst] = row[src];
        pcol[dst] = col[src];
        pval[dst] = val[src];
    }
}

} // namespace

// Column panels for COO triplets (row-sorted on entry): a copy grouped by column panel, stable inside
// a panel, every panel padded to a multiple of 1024 entries with (row -1, column 0, value 0).
// start[0..8] receives the panel boundaries in the copy; the three output arrays are hipMalloc'ed here.
extern "C" int spmv_hip_internal_coo_panels(int32_t cols, int32_t nnz, const int32_t * d_row, const int32_t * d_col,
                                            const double * d_val, int32_t ** out_row, int32_t ** out_col,
                                            double ** out_val, long long * start, hipStream_t s)
{
    *out_row = *out_col = nullptr;
    *out_val = nullptr;
    const int width = (cols + 7) / 8;
    int32_t *key = nullptr, *key_sorted = nullptr, *idx = nullptr, *idx_sorted = nullptr, *d_first = nullptr;
    void * temp = nullptr;
    size_t temp_bytes = 0;
    const size_t nb = (size_t) nnz * sizeof(int32_t);
    hipError_t e = hipMalloc((void **) &key, nb);
    if (e == hipSuccess) e = hipMalloc((void **) &key_sorted, nb);
    if (e == hipSuccess) e = hipMalloc((void **) &idx, nb);
    if (e == hipSuccess) e = hipMalloc((void **) &idx_sorted, nb);
    if (e == hipSuccess) e = hipMalloc((void **) &d_first, 9 * sizeof(int32_t));
    const int grid = (int) std::min<long long>(((long long) nnz + 255) / 256, 256 * 64);
    int32_t first[9];
    if (e == hipSuccess) {
        hipLaunchKernelGGL(panel_key_kernel, dim3(grid), dim3(256), 0, s, nnz, width, d_col, key, idx);
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, key, key_sorted, idx, idx_sorted, nnz, 0, 3, s);
    }
    if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key_sorted, idx, idx_sorted, nnz, 0, 3, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_first, 0xFF, 9 * sizeof(int32_t), s); // -1 = panel without entries
    if (e == hipSuccess) {
        hipLaunchKernelGGL(panel_first_kernel, dim3(grid), dim3(256), 0, s, nnz, key_sorted, d_first);
        e = hipMemcpyAsync(first, d_first, 9 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    PanelPlace pp;
    long long total = 0;
    if (e == hipSuccess) {
        first[8] = nnz;
        for (int k = 7; k >= 0; --k)
            if (first[k] < 0)
                first[k] = first[k + 1];
        for (int k = 0; k < 8; ++k) {
            pp.ufirst[k] = first[k];
            pp.start[k] = total;
            const long long len = (long long) first[k + 1] - first[k];
            total += (len + 1023) / 1024 * 1024;
        }
        pp.ufirst[8] = nnz;
        pp.start[8] = total;
        for (int k = 0; k <= 8; ++k)
            start[k] = pp.start[k];
        e = hipMalloc((void **) out_row, (size_t) total * sizeof(int32_t) + 64);
        if (e == hipSuccess) e = hipMalloc((void **) out_col, (size_t) total * sizeof(int32_t) + 64);
        if (e == hipSuccess) e = hipMalloc((void **) out_val, (size_t) total * sizeof(double) + 64);
        if (e == hipSuccess) e = hipMemsetAsync(*out_row, 0xFF, (size_t) total * sizeof(int32_t) + 64, s);
        if (e == hipSuccess) e = hipMemsetAsync(*out_col, 0, (size_t) total * sizeof(int32_t) + 64, s);
        if (e == hipSuccess) e = hipMemsetAsync(*out_val, 0, (size_t) total * sizeof(double) + 64, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(panel_place_kernel, dim3(grid), dim3(256), 0, s, nnz, key_sorted, idx_sorted, d_row, d_col, d_val,
                               *out_row, *out_col, *out_val, pp);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    void * frees[] = {key, key_sorted, idx, idx_sorted, d_first, temp};
    for (void * q : frees)
        if (q)
            (void) hipFree(q);
    if (e != hipSuccess) {
        if (*out_row) (void) hipFree(*out_row);
        if (*out_col) (void) hipFree(*out_col);
        if (*out_val) (void) hipFree(*out_val);
        *out_row = *out_col = nullptr;
        *out_val = nullptr;
        return SPMV_HIP_ERR_HIP;
    }
    return SPMV_HIP_OK;
}
