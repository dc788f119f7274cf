// ell_kernels.hpp -- ELLPACK, column-major, one lane per row (src/matrix/ell-matrix.cpp:243-258) + the upload transpose.
#pragma once

#include "tile_common.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// ELLPACK.  The reference stores row-major (k = i*L + l), which on a GPU would make
// lanes read with stride 12*L bytes; the upload transposes to column-major
// (k = l*rows + i) so lane i reads consecutive addresses for each l.
// One lane per row, l ascending, padded entries multiplied like real ones:
// bit-exact with ell_spmv_inner_loop (src/matrix/ell-matrix.cpp:243-258).
// ---------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ell_kernel(
    int rows, int row_length, const int32_t * __restrict__ j, const double * __restrict__ a,
    const double * __restrict__ x, double * __restrict__ y)
{
    const long long stride = (long long) gridDim.x * BLOCK;
    for (long long i = (long long) blockIdx.x * BLOCK + threadIdx.x; i < rows; i += stride) {
        double z = 0.0;
        int l = 0;
        for (; l + 4 <= row_length; l += 4) {
            const long long k = (long long) l * rows + i;
            const int c0 = j[k], c1 = j[k + rows], c2 = j[k + 2LL * rows], c3 = j[k + 3LL * rows];
            const double v0 = a[k], v1 = a[k + rows], v2 = a[k + 2LL * rows], v3 = a[k + 3LL * rows];
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            z += v0 * x0;
            z += v1 * x1;
            z += v2 * x2;
            z += v3 * x3;
        }
        for (; l < row_length; ++l) {
            const long long k = (long long) l * rows + i;
            z += a[k] * x[j[k]];
        }
        y[i] += z;
    }
}

// Row-major -> column-major (upload time only).  A wave reads 64 consecutive
// row-major elements (coalesced) and scatters them; the scatter is absorbed by L2.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ell_transpose_kernel(
    int rows, int row_length, const int32_t * __restrict__ j_rm, const double * __restrict__ a_rm,
    int32_t * __restrict__ j_cm, double * __restrict__ a_cm)
{
    const long long n = (long long) rows * row_length;
    const long long stride = (long long) gridDim.x * BLOCK;
    for (long long k = (long long) blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
        const long long i = k / row_length;
        const long long l = k - i * row_length;
        j_cm[l * rows + i] = j_rm[k];
        a_cm[l * rows + i] = a_rm[k];
    }
}

} // namespace spmv
