// spmv_kernels.hpp -- hand-written gfx950 kernels for y += A*x (fp64 values, int32 indices).
//
// Built with -ffp-contract=off: a product is rounded before it is added, as in the
// reference's x86-64 -O3 build (no FMA), so every path that adds a row's products
// left to right with one lane is bit-identical to the reference loop
// (src/matrix/csr-matrix-spmv.cpp:29-32, src/matrix/ell-matrix.cpp:251-257).
//
// None of this is GEMM-shaped: ~0.13 flop/byte, HBM-bound.  No MFMA on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_ops.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// XCD-aware workgroup order.  Workgroups are dealt round-robin to the 8 XCDs
// (blockIdx b and b+8 share an L2).  Row blocks that are neighbours in the matrix
// read overlapping windows of x, so give each XCD one contiguous run of blocks:
// logical = (b % 8) * ceil-ish(n/8) + b / 8, bijective for any n.  Placement is a
// speed matter only; any mapping gives the same y.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_remap(int bid, int nblk, bool enable)
{
    if (!enable || nblk < 16)
        return bid;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

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

// ---------------------------------------------------------------------------------
// COO in any order.  Each wave takes 64 consecutive entries per step, forms the
// products, adds runs of equal row index inside the wave (segmented inclusive scan
// over head flags, ds_bpermute moves) and issues ONE fp64 atomic per run, so a
// row-sorted file costs ~1 atomic per row per wave and an unsorted one degrades to
// one atomic per entry -- the semantics of the reference's coo_spmv_atomic
// (src/matrix/coo-matrix.cpp:287-309).
// ---------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void coo_kernel(
    int nnz, const int32_t * __restrict__ ri, const int32_t * __restrict__ ci,
    const double * __restrict__ v, const double * __restrict__ x, double * __restrict__ y)
{
    const int lane = (int) __lane_id();
    const long long total = (long long) gridDim.x * BLOCK;
    const long long gid = (long long) blockIdx.x * BLOCK + threadIdx.x;
    for (long long base = 0; base < nnz; base += total) { // uniform trip count
        const long long k = base + gid;
        const bool valid = k < nnz;
        int r = -1;
        double s = 0.0;
        if (valid) {
            r = ri[k];
            s = v[k] * x[ci[k]];
        }
        const int rprev = lane_up(r, 1);
        int head = (lane == 0) || (rprev != r);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const double sp = lane_up(s, d);
            const int hp = lane_up(head, d);
            if (lane >= d && !head) {
                s += sp;
                head |= hp;
            }
        }
        const int rnext = lane_down1(r);
        const bool tail = (lane == kWave - 1) || (rnext != r);
        if (valid && tail)
            unsafeAtomicAdd(y + r, s);
    }
}

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
