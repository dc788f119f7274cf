// triad_kernels.hpp -- STREAM triad (src/kernels/triad.cpp:48-54), the empirical bandwidth roofline.
#pragma once

#include "tile_common.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// STREAM triad a = b + q*c (reference src/kernels/triad.cpp:48-54): two doubles per
// lane per step (16-byte loads/stores), grid-stride.  The measured rate of this
// kernel is the empirical HBM roofline the SpMV kernels are compared against.
// ---------------------------------------------------------------------------------
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void triad_kernel(
    long long n, double * __restrict__ a, const double * __restrict__ b,
    const double * __restrict__ c, double q)
{
    const long long n2 = n >> 1; // double2 elements
    const double2 * __restrict__ b2 = reinterpret_cast<const double2 *>(b);
    const double2 * __restrict__ c2 = reinterpret_cast<const double2 *>(c);
    double2 * __restrict__ a2 = reinterpret_cast<double2 *>(a);
    const long long stride = (long long) gridDim.x * BLOCK;
    const long long gid = (long long) blockIdx.x * BLOCK + threadIdx.x;
    long long i = gid;
    // UNROLL independent 16-byte loads per array in flight per lane
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        double2 vb[UNROLL], vc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            vb[u] = b2[i + u * stride];
            vc[u] = c2[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            a2[i + u * stride] = make_double2(vb[u].x + q * vc[u].x, vb[u].y + q * vc[u].y);
    }
    for (; i < n2; i += stride) {
        const double2 vb = b2[i], vc = c2[i];
        a2[i] = make_double2(vb.x + q * vc.x, vb.y + q * vc.y);
    }
    if ((n & 1) && gid == 0)
        a[n - 1] = b[n - 1] + q * c[n - 1];
}

// Experiment variants of the triad (tools/kernel_sweep.py --triad-variants): not part of the ABI.
template <int BLOCK, bool NT_STORE>
__global__ __launch_bounds__(BLOCK) void triad_flat_kernel(
    long long n2, double * __restrict__ a, const double * __restrict__ b,
    const double * __restrict__ c, double q)
{
    const long long i = (long long) blockIdx.x * BLOCK + threadIdx.x; // one 16-byte element per lane
    if (i < n2) {
        const v2d vb = reinterpret_cast<const v2d *>(b)[i];
        const v2d vc = reinterpret_cast<const v2d *>(c)[i];
        const v2d r = v2d{vb.x + q * vc.x, vb.y + q * vc.y};
        if (NT_STORE)
            __builtin_nontemporal_store(r, reinterpret_cast<v2d *>(a) + i);
        else
            reinterpret_cast<v2d *>(a)[i] = r;
    }
}

} // namespace spmv
