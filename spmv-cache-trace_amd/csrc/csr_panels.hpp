// csr_panels.hpp -- plan-time kernels of the column panels (csr_wavetile_kernel<..., PANELS> multiplies them).
#pragma once

#include "tile_common.hpp"

namespace spmv {

// Plan-time kernels of the column panels.  csr_panel_count_kernel: entries of row r in panel k
// -> count[k * rows + r] (one thread per row); after an exclusive scan over the 8 * rows counts,
// csr_panel_scatter_kernel copies every entry to its panel's place, rows in order, entries of a
// row in their original order.
static __global__ __launch_bounds__(256) void csr_panel_count_kernel(
    int rows, int width, const int32_t * __restrict__ p, const int32_t * __restrict__ j, int32_t * __restrict__ count)
{
    const long long r = (long long) blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    int n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = p[r]; k < p[r + 1]; ++k) {
        const int pk = j[k] / width;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            n[q] += (pk == q);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        count[(size_t) q * rows + r] = n[q];
}

static __global__ __launch_bounds__(256) void csr_panel_scatter_kernel(
    int rows, int width, const int32_t * __restrict__ p, const int32_t * __restrict__ j, const double * __restrict__ a,
    const int32_t * __restrict__ vrow_ptr, int32_t * __restrict__ pj, double * __restrict__ pa)
{
    const long long r = (long long) blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    int cur[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        cur[q] = vrow_ptr[(size_t) q * rows + r];
    for (int k = p[r]; k < p[r + 1]; ++k) {
        const int c = j[k];
        const int pk = c / width;
        int dst = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (pk == q)
                dst = cur[q]++;
        pj[dst] = c;
        pa[dst] = a[k];
    }
}

} // namespace spmv
