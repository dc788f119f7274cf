// csr_blocktile.hpp -- block-row tiles: matrices with 3 unknowns per mesh node (finite-element elasticity: Queen_4147 and its
// relatives), whose rows come in triples with identical columns and whose columns come in aligned triples -- dense 3 x 3 blocks.
//
// The reference's loop (src/matrix/csr-matrix-spmv.cpp:21-33) reads one column index per stored entry; here a tile reads ONE
// 16-bit number per 3 x 3 BLOCK (2/9 of a byte per entry instead of 2 or 4) and no row_ptr: the CSR arrays themselves are
// untouched (values are read in place, 8 bytes per entry), the plan only adds a side stream of block columns,
//     bcol[block] = (first column of the block - the tile's smallest column) / 3  |  0x8000 where a block row begins,
// written at plan time by csr_block3_mark_kernel after it has CHECKED the tile entry by entry (rows in triples of equal
// length, every row a copy of its triple's first row, columns in runs c, c+1, c+2 with c congruent to the tile's base mod 3).
//
// Multiply: ONE LANE PER BLOCK (a 512-entry tile holds at most 56).  The lanes load the block stream (128 bytes per wave);
// a ballot of the begin marks tells every lane its block row's first block and length -- and with them where its nine
// values lie, because every block in front of it holds exactly nine entries: entry of (block row b, row a, block j) =
// k0 + 9 * first_b + a * 3 * nb_b + 3 * j.  So nothing but the descriptor and the block stream stands between a wave and its
// value loads: three pieces of 24 consecutive bytes per lane, consecutive across the lanes of a block row (coalesced, every
// value byte fetched once), and three consecutive x entries per lane instead of nine gathers.  A lane forms its block's three
// partial row sums; they meet through the wave's LDS slice (three doubles per lane), where LPR lanes per row add them up.
// No values and no products are parked in LDS at all.
//
// Summation order: inside a block left to right, then LPR interleaved chains and a butterfly -- not the reference's order:
// 1e-10 class, like every row of more than 16 entries on the default path.  Block tiles are therefore only formed from rows of
// MORE than 16 entries (shorter rows keep their one-lane-per-row, bit-exact path) and never under SPMV_HIP_FLAG_EXACT_ORDER.
#pragma once

#include "tile_common.hpp"

namespace spmv {

constexpr int kTileMetaBlock3 = 1 << 23; // set by csr_block3_mark_kernel; only the kernels that know block tiles look at it
constexpr unsigned kBlockRowBegin = 0x8000u;
constexpr int kBlockTileMaxRows = 30;    // 10 block rows (the row lanes below: 2 per row)

// where the block stream lives: behind the 16-bit column stream, in the same allocation (no extra kernel argument)
__host__ __device__ __forceinline__ size_t block_stream_offset(long long nnz_total)
{
    return (size_t) ((nnz_total + 63) & ~63LL) + 64;
}
// ... and where a tile's blocks start in it: every tile in front of it holds at most (its entries / 9) blocks
__host__ __device__ __forceinline__ int block_stream_index(int k0) { return k0 / 9; }

template <int LPR>
__device__ __forceinline__ double block_row_sum(const double * part, int first, int nb, int t)
{
    double z = 0.0;
    for (int i = t; i < nb; i += LPR)
        z += part[first + i];
    return z;
}

typedef double v2d_u8 __attribute__((ext_vector_type(2), aligned(8)));

template <typename YStore>
__device__ __forceinline__ void tile_rows_block3(
    double * lds, const uint16_t * __restrict__ bc, const double * __restrict__ a, const double * __restrict__ xt /* x + tile base */,
    const double * y_in, int r0, int k0, int k1, int nrows, int lane, YStore && store)
{
    const int nblk = __builtin_amdgcn_readfirstlane((k1 - k0) / 9); // 2 .. 56
    const bool mine = lane < nblk;
    const int L = mine ? lane : nblk - 1; // idle lanes repeat the last block's loads (same addresses: no traffic) and add nothing
    const unsigned e = bc[L];
    const unsigned long long starts = __ballot(mine && (e & kBlockRowBegin)); // bit 0 is always set
    const unsigned long long upto = starts & (~0ull >> (63 - L));
    const int first = 63 - __builtin_clzll(upto);
    const unsigned long long after = L < 63 ? (starts >> (L + 1)) : 0ull;
    const int next = after ? L + 1 + __builtin_ctzll(after) : nblk;
    const int nb = next - first;
    // (1) everything that depends on the block stream only: the lane's nine values and three x entries
    const char * va = reinterpret_cast<const char *>(a + k0);
    const unsigned o0 = (unsigned) (9 * first + 3 * (L - first)) * 8u, step = (unsigned) (3 * nb) * 8u;
    const v2d_u8 p0 = *reinterpret_cast<const v2d_u8 *>(va + o0);
    const double q0 = *reinterpret_cast<const double *>(va + o0 + 16);
    const v2d_u8 p1 = *reinterpret_cast<const v2d_u8 *>(va + o0 + step);
    const double q1 = *reinterpret_cast<const double *>(va + o0 + step + 16);
    const v2d_u8 p2 = *reinterpret_cast<const v2d_u8 *>(va + o0 + 2 * step);
    const double q2 = *reinterpret_cast<const double *>(va + o0 + 2 * step + 16);
    const unsigned xo = 3u * (e & 0x7FFFu) * 8u;
    const v2d_u8 x01 = *reinterpret_cast<const v2d_u8 *>(reinterpret_cast<const char *>(xt) + xo);
    const double x2 = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(xt) + xo + 16);
    // (2) the row lanes: LPR lanes per row, old y requested now
    const int lpr_log2 = nrows <= 3 ? 4 : (nrows <= 6 ? 3 : (nrows <= 15 ? 2 : 1)); // wave-uniform; rows * LPR <= 64
    const int row = lane >> lpr_log2, t = lane & ((1 << lpr_log2) - 1);
    const bool row_lane = row < nrows;
    const int rowc = row_lane ? row : nrows - 1;
    const double yv = y_in[r0 + rowc];
    // (3) the block's three partial sums, left to right
    double s0 = p0.x * x01.x, s1 = p1.x * x01.x, s2 = p2.x * x01.x;
    s0 += p0.y * x01.y;
    s1 += p1.y * x01.y;
    s2 += p2.y * x01.y;
    s0 += q0 * x2;
    s1 += q1 * x2;
    s2 += q2 * x2;
    int * info = reinterpret_cast<int *>(lds + 3 * kWave);
    if (mine) {
        lds[lane] = s0;
        lds[kWave + lane] = s1;
        lds[2 * kWave + lane] = s2;
        if (lane == first)
            info[__builtin_popcountll(upto) - 1] = first | (nb << 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (4) row sums: row 3 b + a adds the a-th partial sums of block row b's lanes
    const int bi = info[rowc / 3];
    const int rfirst = bi & 0xFF, rnb = row_lane ? (bi >> 8) : 0;
    const double * part = lds + (rowc % 3) * kWave;
    double z;
    switch (lpr_log2) {
    case 4: z = group_sum<16>(block_row_sum<16>(part, rfirst, rnb, t)); break;
    case 3: z = group_sum<8>(block_row_sum<8>(part, rfirst, rnb, t)); break;
    case 2: z = group_sum<4>(block_row_sum<4>(part, rfirst, rnb, t)); break;
    default: z = group_sum<2>(block_row_sum<2>(part, rfirst, rnb, t)); break;
    }
    if (row_lane && t == 0)
        store(r0 + row, yv + z);
}

// Plan time, one wave per tile (spmv_hip_plan_csr_repack: the pass that has row_ptr on the device): a stream tile with
// 16-bit columns, rows of more than 16 entries and 3, 6, ... kBlockTileMaxRows rows is checked ENTRY BY ENTRY for the block
// structure described above; where it holds the tile's block stream is written and the tile marked.  count[0] += tiles,
// count[1] += their entries, count[2], count[3]: the same for the tiles no block window has claimed.
static __global__ __launch_bounds__(256) void csr_block3_mark_kernel(
    int ntiles, int tile, int4 * __restrict__ desc, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    uint16_t * __restrict__ bstream, unsigned long long * __restrict__ count)
{
    const int wave = (int) threadIdx.x >> 6;
    const int lane = (int) __lane_id();
    const int w = blockIdx.x * 4 + wave;
    if (w >= ntiles)
        return;
    const int4 d0 = desc[w];
    const int4 d1 = desc[w + 1];
    const int r0 = d0.x & ~kTileFlagPartial, r1 = d1.x & ~kTileFlagPartial;
    const int k0 = d0.y, k1 = d1.y;
    const int meta = d0.z;
    const int nrows = r1 - r0, n = k1 - k0;
    // (tiles marked for a block window are checked too: where block tiles turn out to be the majority the plan drops the windows)
    const int other = kTileMetaShifted | kTileMetaXWin | kTileMetaXSeg | kTileMetaPattern | (1 << 21) /* balanced tiles */;
    if ((d0.x & kTileFlagPartial) || !(meta & kTileMetaFast) || !(meta & kTileMetaNarrow) || (meta & other)
        || nrows < 3 || nrows > kBlockTileMaxRows || nrows % 3 != 0 || n % 9 != 0 || n < 18 || n / 9 > kWave || k1 - (k0 & ~3) > tile)
        return;
    const int cmin = d0.w;
    // row starts in lanes 0 .. nrows (nrows <= 30)
    const int ps = p[r0 + (lane <= nrows ? lane : nrows)];
    const int len = __shfl_down(ps, 1) - ps; // lanes < nrows
    const int len0 = __shfl(len, lane - lane % 3);
    int ok = lane >= nrows || (len == len0 && len % 3 == 0 && len > 16);
    ok = __all(ok);
    if (!ok)
        return;
    // row by row (the row number is wave-uniform: its bounds come out of lane r with a readlane, not a shuffle per entry --
    // the first version looked every entry's row up with a loop of shuffles: 38.8 ms for the queen-like matrix's 808 K tiles)
    for (int r = 0; r < nrows; ++r) {
        const int start = __shfl(ps, r), rl = __shfl(len, r), a = r % 3;
        for (int k = start + lane; k < start + rl; k += kWave) {
            const int pos = k - start;
            const int c = j[k];
            int good = 1;
            if (a != 0)
                good &= c == j[k - a * rl];
            if (pos % 3 != 0)
                good &= c == j[k - 1] + 1;
            else
                good &= (c - cmin) % 3 == 0 && (c - cmin) / 3 < 0x8000;
            ok &= good;
        }
    }
    ok = __all(ok);
    if (!ok)
        return;
    // the block stream: lane = block, block rows one after the other
    const int nblk = n / 9;
    {
        int cum = 0, entry = k0, begin = 0;
        for (int r = 0; r < nrows; r += 3) {
            const int nb = __shfl(len, r) / 3, st = __shfl(ps, r);
            if (lane >= cum && lane < cum + nb) {
                entry = st + 3 * (lane - cum);
                begin = lane == cum;
            }
            cum += nb;
        }
        if (lane < nblk)
            bstream[block_stream_index(k0) + lane] = (uint16_t) (((j[entry] - cmin) / 3) | (begin ? kBlockRowBegin : 0u));
    }
    if (lane == 0) {
        desc[w].z = meta | kTileMetaBlock3;
        striped_add(count, 0, 1ull);
        striped_add(count, 1, (unsigned long long) n);
        if (!(meta & kTileMetaBlockWin)) {
            striped_add(count, 2, 1ull);
            striped_add(count, 3, (unsigned long long) n);
        }
    }
}

// the plan gave up its block windows for block tiles: every tile belongs to csr_wavetile_kernel again
static __global__ __launch_bounds__(256) void csr_clear_blockwin_kernel(int ntiles, int4 * __restrict__ desc)
{
    const int w = (int) (blockIdx.x * 256 + threadIdx.x);
    if (w < ntiles && (desc[w].z & kTileMetaBlockWin))
        desc[w].z &= ~kTileMetaBlockWin;
}

} // namespace spmv
