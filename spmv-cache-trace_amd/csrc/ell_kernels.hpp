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
// Round 4: the loop is software-pipelined -- while the U entries of one batch gather x and are added, the column / value loads
// of the NEXT batch are already in flight (round 3's loop issued four loads, waited, gathered, waited, added: the streams were
// requested half of the time and the kernel sat at 0.68-0.70 of the roofline for every row length).  Same additions in the
// same order: bit-exact as before.
template <int BLOCK, int U = 8>
__global__ __launch_bounds__(BLOCK) void ell_kernel(
    int rows, int row_length, const int32_t * __restrict__ j, const double * __restrict__ a,
    const double * __restrict__ x, double * __restrict__ y)
{
    const long long stride = (long long) gridDim.x * BLOCK;
    const int batches = row_length / U;
    for (long long i = (long long) blockIdx.x * BLOCK + threadIdx.x; i < rows; i += stride) {
        const int32_t * jp = j + i;
        const double * ap = a + i;
        const double y0 = y[i];
        int c[U];
        double v[U];
        if (batches > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = jp[(long long) u * rows];
                v[u] = ap[(long long) u * rows];
            }
        }
        double z = 0.0;
        // (the last batch is peeled off: a conditional prefetch inside the loop makes the compiler wait for ALL loads at the join)
        for (int b = 0; b + 1 < batches; ++b) {
            // the gathers of this batch FIRST, the next batch's streams behind them: loads return in the order they were issued,
            // so the additions below wait for the gathers only (s_waitcnt vmcnt(2 U + ...)) while the streams stay in flight
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                xv[u] = x[c[u]];
            int cn[U];
            double vn[U];
            const long long base = (long long) (b + 1) * U * rows;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cn[u] = jp[base + (long long) u * rows];
                vn[u] = ap[base + (long long) u * rows];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                z += v[u] * xv[u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = cn[u];
                v[u] = vn[u];
            }
        }
        if (batches > 0) {
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < U; ++u)
                z += v[u] * xv[u];
        }
        for (int l = batches * U; l < row_length; ++l) {
            const long long k = (long long) l * rows;
            z += ap[k] * x[jp[k]];
        }
        y[i] = y0 + z;
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

// (Round 4 also tried the matrix in SLICES of 64 rows -- slice s column-major, entry (l, lane) at (s * L + l) * 64 + lane, so
// that a wave streams its 768 L bytes front to back instead of touching L planes rows * 8 bytes apart: 0.65-0.72 against
// 0.69-0.73 for the layout above with the same pipelined loop (profiles/r04_ell_long_rows_*.log): not the page locality
// either.  Removed again.  One lane per row keeps L dependent rounds of "streams back, gather, add" per wave with at most
// two batches in flight; the in-place row-major path, which gives a row several lanes, is the faster one wherever its tiles
// fill: DESIGN.md section 3.4.)

} // namespace spmv
