// csr_basic.hpp -- the selectable CSR kernels besides the wave tiles: one lane per row, L lanes per row, adaptive row blocks.
#pragma once

#include "tile_common.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// CSR, one lane per row ("scalar").  Reference order: bit-exact.
// ---------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void csr_scalar_kernel(
    int rows, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    const double * __restrict__ a, const double * __restrict__ x, double * __restrict__ y)
{
    const long long stride = (long long) gridDim.x * BLOCK;
    for (long long i = (long long) blockIdx.x * BLOCK + threadIdx.x; i < rows; i += stride) {
        const int k1 = p[i + 1];
        double z = 0.0;
        for (int k = p[i]; k < k1; ++k)
            z += a[k] * x[j[k]];
        y[i] += z;
    }
}

// ---------------------------------------------------------------------------------
// CSR, LPR lanes per row ("vector").  A wave owns 64/LPR consecutive rows; the
// lanes of a row walk its entries with unit stride (coalesced col/val loads), then
// a DPP / ds_swizzle butterfly adds the LPR partial sums.
// ---------------------------------------------------------------------------------
template <int LPR, int BLOCK>
__global__ __launch_bounds__(BLOCK) void csr_vector_kernel(
    int rows, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    const double * __restrict__ a, const double * __restrict__ x, double * __restrict__ y)
{
    constexpr int ROWS_PER_BLOCK = BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int sub = threadIdx.x / LPR;
    const long long stride = (long long) gridDim.x * ROWS_PER_BLOCK;
    // trip count is uniform per workgroup, so every lane reaches group_sum
    for (long long base = (long long) blockIdx.x * ROWS_PER_BLOCK; base < rows; base += stride) {
        const long long row = base + sub;
        const bool valid = row < rows;
        int k0 = 0, k1 = 0;
        if (valid) {
            k0 = p[row];
            k1 = p[row + 1];
        }
        double z = 0.0;
        for (int k = k0 + lane; k < k1; k += LPR)
            z += a[k] * x[j[k]];
        z = group_sum<LPR>(z);
        if (valid && lane == 0)
            y[row] += z;
    }
}

// ---------------------------------------------------------------------------------
// CSR, adaptive row blocks.
//
// The host cuts the rows into blocks [blk_row[b], blk_row[b+1]) holding at most
// TILE stored entries (counted from the 4-aligned start) and at most BLOCK rows;
// a row longer than TILE is a block by itself.
//
// Stream block: the workgroup reads its contiguous slice of column_index / value
// with 16-byte-per-lane loads (int4 + 2 x double2, perfectly coalesced whatever the
// row lengths are), gathers x, and parks the rounded products in LDS.  After one
// barrier each row is summed from LDS by L lanes (L chosen per block from its
// entries-per-row, L = 1 gives the reference's left-to-right order exactly).
//
// Long row: the whole workgroup strides the row, wave butterfly + LDS combine.
// ---------------------------------------------------------------------------------
template <int L, int BLOCK>
__device__ __forceinline__ void sum_rows_from_lds(
    const double * prod, const int32_t * __restrict__ p, double * __restrict__ y,
    int r0, int nrows, int kb)
{
    constexpr int ROWS_PER_PASS = BLOCK / L;
    const int lane = threadIdx.x % L;
    const int sub = threadIdx.x / L;
    for (int rb = 0; rb < nrows; rb += ROWS_PER_PASS) {
        const int r = rb + sub;
        const bool valid = r < nrows;
        int s = 0, e = 0;
        if (valid) {
            s = p[r0 + r] - kb;
            e = p[r0 + r + 1] - kb;
        }
        double z = 0.0;
        for (int k = s + lane; k < e; k += L)
            z += prod[k];
        z = group_sum<L>(z);
        if (valid && lane == 0)
            y[r0 + r] += z;
    }
}

template <int BLOCK, int TILE>
__global__ __launch_bounds__(BLOCK) void csr_adaptive_kernel(
    int nblk, const int32_t * __restrict__ blk_row, const int32_t * __restrict__ p,
    const int32_t * __restrict__ j, const double * __restrict__ a,
    const double * __restrict__ x, double * __restrict__ y, int nnz_total, int xcd_aware,
    int exact_order)
{
    __shared__ __attribute__((aligned(16))) double prod[TILE + 4];
    __shared__ double wave_part[BLOCK / kWave];

    const int b = xcd_remap(blockIdx.x, nblk, xcd_aware != 0);
    const int r0 = blk_row[b];
    const int r1 = blk_row[b + 1];
    const int nrows = r1 - r0;
    const int k0 = p[r0];
    const int k1 = p[r1];
    const int kb = k0 & ~3;

    if (k1 - kb <= TILE) {
        // ---- stream: products to LDS -------------------------------------------
        for (int e = kb + 4 * (int) threadIdx.x; e < k1; e += 4 * BLOCK) {
            int c0, c1, c2, c3;
            double v0, v1, v2, v3;
            if (e + 3 < nnz_total) {
                const int4 c = *reinterpret_cast<const int4 *>(j + e);
                const double2 va = *reinterpret_cast<const double2 *>(a + e);
                const double2 vb = *reinterpret_cast<const double2 *>(a + e + 2);
                c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
                v0 = va.x; v1 = va.y; v2 = vb.x; v3 = vb.y;
            } else { // last (partial) quad of the arrays
                c0 = j[e];
                v0 = a[e];
                c1 = (e + 1 < nnz_total) ? j[e + 1] : 0;
                v1 = (e + 1 < nnz_total) ? a[e + 1] : 0.0;
                c2 = (e + 2 < nnz_total) ? j[e + 2] : 0;
                v2 = (e + 2 < nnz_total) ? a[e + 2] : 0.0;
                c3 = 0;
                v3 = 0.0;
            }
            // entries before k0 / after k1 belong to neighbouring blocks
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
            if (e >= k0) q0 = v0 * x[c0];
            if (e + 1 >= k0 && e + 1 < k1) q1 = v1 * x[c1];
            if (e + 2 >= k0 && e + 2 < k1) q2 = v2 * x[c2];
            if (e + 3 >= k0 && e + 3 < k1) q3 = v3 * x[c3];
            double2 * dst = reinterpret_cast<double2 *>(prod + (e - kb));
            dst[0] = make_double2(q0, q1);
            dst[1] = make_double2(q2, q3);
        }
        __syncthreads();

        // ---- per-row sums from LDS ------------------------------------------------
        // lanes per row: enough to keep the workgroup busy, never more than the
        // rows are long; one lane per row keeps the reference's summation order
        int lanes = 1;
        if (!exact_order && nrows > 0) {
            const int avg = (k1 - k0) / nrows;
            int cap = BLOCK / nrows; // >= 1 because nrows <= BLOCK
            if (cap > kWave) cap = kWave;
            while (lanes * 2 <= cap && lanes * 8 <= avg)
                lanes *= 2;
        }
        switch (lanes) {
        case 1: sum_rows_from_lds<1, BLOCK>(prod, p, y, r0, nrows, kb); break;
        case 2: sum_rows_from_lds<2, BLOCK>(prod, p, y, r0, nrows, kb); break;
        case 4: sum_rows_from_lds<4, BLOCK>(prod, p, y, r0, nrows, kb); break;
        case 8: sum_rows_from_lds<8, BLOCK>(prod, p, y, r0, nrows, kb); break;
        case 16: sum_rows_from_lds<16, BLOCK>(prod, p, y, r0, nrows, kb); break;
        case 32: sum_rows_from_lds<32, BLOCK>(prod, p, y, r0, nrows, kb); break;
        default: sum_rows_from_lds<64, BLOCK>(prod, p, y, r0, nrows, kb); break;
        }
    } else if (!exact_order) {
        // ---- one long row: whole workgroup strides it --------------------------------
        double z = 0.0;
        for (int k = k0 + (int) threadIdx.x; k < k1; k += BLOCK)
            z += a[k] * x[j[k]];
        z = group_sum<kWave>(z);
        if ((threadIdx.x & (kWave - 1)) == 0)
            wave_part[threadIdx.x / kWave] = z;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < BLOCK / kWave; ++w)
                t += wave_part[w];
            y[r0] += t;
        }
    } else {
        // ---- one long row, reference order: tiles of products, lane 0 adds them ------
        double z = 0.0;
        for (int t0 = k0; t0 < k1; t0 += TILE) {
            const int t1 = (t0 + TILE < k1) ? t0 + TILE : k1;
            for (int k = t0 + (int) threadIdx.x; k < t1; k += BLOCK)
                prod[k - t0] = a[k] * x[j[k]];
            __syncthreads();
            if (threadIdx.x == 0)
                for (int k = 0; k < t1 - t0; ++k)
                    z += prod[k];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            y[r0] += z;
    }
}

} // namespace spmv
