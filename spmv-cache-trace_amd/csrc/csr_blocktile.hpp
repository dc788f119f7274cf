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

// MASKED block tiles (round 5): what real finite-element files do to the picture above -- explicit zeros dropped from a block
// (7 or 8 of its 9 entries stored), a node with one or two unknowns that shifts the grid of column triples, rows of a triple
// that differ in length.  Such a tile keeps one lane per block, but a block is now ANY three consecutive columns [c, c + 3) in
// the tile's three-row strip together with a 9-bit mask of the entries that are stored, one 32-bit word per block:
//     bits 0-21  c - the tile's smallest column      bits 22-30  mask (bit 3 a + b: row a of the triple has column c + b)
//     bit 31     a block row begins here             (a word with an empty mask ends the tile's list: fewer than 64 blocks)
// (round 6: 22 bits of column instead of 16 -- the tile's columns may span 4 M, not 64 K -- so that a tile whose columns are too
// far apart for 16-bit offsets can be a block tile too: an UNSTRUCTURED mesh numbered by reverse Cuthill-McKee has a band of
// ~7 n^(2/3) nodes, and the Delaunay twin of the queen-like stand-in -- 700 K nodes, 2.1 M rows -- had 8 % narrow tiles, no block
// tile and ran at 0.66 of the roofline: profiles/r06_zoo_delaunay_first.log.  Such a WIDE tile carries kTileMetaBlock3 |
// kTileMetaBlock3Masked without kTileMetaNarrow, and its descriptor's .w is set to its smallest column by the mark kernel.)
// The blocks of a block row are the greedy cover of the union of its three rows' columns (csr_block3m_mark_kernel): for dense
// aligned blocks that is the blocks themselves, otherwise a cover that is merely a little less full.  Where a block's entries
// lie in the value array follows from the masks in front of it: a prefix sum over the lanes of the three per-row counts
// (three 10-bit fields in one integer, one DPP scan), so the values are still read IN PLACE (8 bytes per stored entry) and the
// column information costs 4 bytes per block -- 0.44 ... 0.67 bytes per entry instead of 2.  Tiles whose blocks are all dense
// and aligned keep the 16-bit stream above (0.22 bytes per entry, no scan).
constexpr int kTileMetaBlock3Masked = 1 << 29; // only together with kTileMetaBlock3 (a block tile has no x window: the window bits are free)
constexpr unsigned kMaskedRowBegin = 0x80000000u;
constexpr int kMaskedColBits = 22;
constexpr unsigned kMaskedColMask = (1u << kMaskedColBits) - 1u;
constexpr int kMaskedMinFill = 6; // entries per block a masked tile must reach to be worth it (and <= 64 blocks)
__host__ __device__ __forceinline__ size_t mask_stream_offset(long long nnz_total) // in 16-bit units from d_col16; 128-byte aligned
{
    return (block_stream_offset(nnz_total) + (size_t) (nnz_total / 9) + 128 + 63) & ~(size_t) 63;
}
__host__ __device__ __forceinline__ size_t mask_stream_words(long long nnz_total) { return (size_t) (nnz_total / 4) + 192; }
// a tile's words start at k0 / 4: every tile in front of it holds at most (its entries / kMaskedMinFill) < (its entries / 4) blocks
__host__ __device__ __forceinline__ int mask_stream_index(int k0) { return k0 >> 2; }

// GROUP tiles (late in round 5): a mesh with 2 or 4 unknowns per node has no 3 x 3 blocks, but its rows still come in groups of
// D = 2 or 4 with IDENTICAL column lists.  Such a tile keeps its values in place and its usual row sums; what changes is where the
// products' columns come from: ONE 16-bit column list per group (the first row's, copied to a plan-owned stream -- 2 / D bytes
// per entry instead of 2) and ONE gather of x per column of a group, multiplied with the D values that share it
// (tile_products_grouped in csr_wavetile.hpp).  The products land where tile_products_narrow would have put them, so the row sums
// -- lanes per row, order, bits -- are those of the plain tile.  D is the plan's (the strict hint of spmv_hip_plan_csr: all rows in
// groups of D equally long rows); the mark: kTileMetaBlock3 together with kTileMetaGroupRows.
// (structure sweep, profiles/r05_zoo_mesh_dofs.log: meshes with 2 / 4 unknowns per node 0.79 / 0.80 of the roofline against 1.05 / 1.10
// with 3 / 6)
constexpr int kTileMetaGroupRows = 1 << 30; // only together with kTileMetaBlock3 (a block tile has no x window: the window bits are free)
// ... and, with kTileMetaGroupRows: every PAIR of group columns the multiply takes is (c, c + 1) -- the dense 2 x 2 / 4 x 4 blocks of a mesh
// with 2 / 4 unknowns per node -- so x is read 16 bytes at a time, one gather per pair, and the group stream holds one column per pair:
// 1 / D bytes per entry (the bit that marks masked 3 x 3 blocks otherwise)
constexpr int kTileMetaGroupPairs = 1 << 29;
// the group stream shares the block stream's place behind the 16-bit columns (a plan has 3 x 3 blocks or row groups, not both);
// a tile's group entries start at k0 / D: every tile in front of it holds at most (its entries / D)
__host__ __device__ __forceinline__ int group_stream_index(int k0, int d) { return k0 / d; }

// inclusive prefix sum over the 64 lanes (every lane active): DPP row shifts inside the rows of 16, then the two row broadcasts
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return v;
}

template <int LPR>
__device__ __forceinline__ double block_row_sum(const double * part, int first, int nb, int t)
{
    double z = 0.0;
    for (int i = t; i < nb; i += LPR)
        z += part[first + i];
    return z;
}

typedef double v2d_u8 __attribute__((ext_vector_type(2), aligned(8)));

template <bool MASKED, typename YStore>
__device__ __forceinline__ void tile_rows_block3(
    double * lds, const void * __restrict__ stream /* uint16 per block, or (MASKED) uint32 per block */, const double * __restrict__ a,
    const double * __restrict__ xt /* x + tile base */, const double * y_in, int r0, int k0, int k1, int nrows, int lane, YStore && store)
{
    unsigned e;
    int nblk;
    if (MASKED) {
        e = reinterpret_cast<const unsigned *>(stream)[lane]; // (words past the tile's last block: allocated; the first of them is 0)
        const unsigned long long none = __ballot(((e >> kMaskedColBits) & 0x1FFu) == 0u); // the terminator, or nothing: 64 blocks
        nblk = none ? (int) __builtin_ctzll(none) : kWave;
    } else {
        nblk = __builtin_amdgcn_readfirstlane((k1 - k0) / 9); // 2 .. 56
    }
    const bool mine = lane < nblk;
    const int L = mine ? lane : nblk - 1; // idle lanes repeat the last block's loads (same addresses: no traffic) and add nothing
    if (!MASKED)
        e = reinterpret_cast<const uint16_t *>(stream)[L];
    const unsigned long long starts = __ballot(mine && (e & (MASKED ? kMaskedRowBegin : kBlockRowBegin))); // bit 0 is always set
    const unsigned long long upto = starts & (~0ull >> (63 - L));
    const int first = 63 - __builtin_clzll(upto);
    const unsigned long long after = L < 63 ? (starts >> (L + 1)) : 0ull;
    const int next = after ? L + 1 + __builtin_ctzll(after) : nblk;
    const int nb = next - first;
    // (1) everything that depends on the block stream only: the lane's nine values and three x entries
    // (scalars, not arrays: an array indexed in an unrolled loop cost this kernel 80 bytes of scratch per lane)
    const char * va = reinterpret_cast<const char *>(a + k0);
    unsigned mask = 0x1FFu;
    unsigned b0, b1, b2; // byte offsets of the three rows' pieces
    if (MASKED) {
        mask = mine ? (e >> kMaskedColBits) & 0x1FFu : 0u;
        // stored entries of this block per row, three 10-bit fields; their prefix sums say where the block's values lie
        const int cnt = __builtin_popcount(mask & 7u) | (__builtin_popcount(mask & 0x38u) << 10) | (__builtin_popcount(mask & 0x1C0u) << 20);
        const int incl = wave_inclusive_scan(cnt);
        const int before_row = __builtin_amdgcn_ds_bpermute(first << 2, incl - cnt); // what lies in front of my block row
        const int row_total = __builtin_amdgcn_ds_bpermute((next - 1) << 2, incl) - before_row; // my block row's three row lengths
        const int mine_before = incl - cnt - before_row;                                  // ... and my place in each of them
        const int base = (before_row & 0x3FF) + ((before_row >> 10) & 0x3FF) + ((before_row >> 20) & 0x3FF);
        // (idle lanes: the tile's first entries -- their own offsets would point past the tile's end)
        // three consecutive doubles per row whatever the mask says (a row's next block, or the next row, follows: the mark
        // kernel makes sure two doubles past the tile's end are still inside the array); the mask picks below
        b0 = mine ? (unsigned) (base + (mine_before & 0x3FF)) * 8u : 0u;
        b1 = mine ? (unsigned) (base + (row_total & 0x3FF) + ((mine_before >> 10) & 0x3FF)) * 8u : 0u;
        b2 = mine ? (unsigned) (base + (row_total & 0x3FF) + ((row_total >> 10) & 0x3FF) + ((mine_before >> 20) & 0x3FF)) * 8u : 0u;
    } else {
        const unsigned step = (unsigned) (3 * nb) * 8u;
        b0 = (unsigned) (9 * first + 3 * (L - first)) * 8u;
        b1 = b0 + step;
        b2 = b1 + step;
    }
    const v2d_u8 p0 = *reinterpret_cast<const v2d_u8 *>(va + b0);
    const double q0 = *reinterpret_cast<const double *>(va + b0 + 16);
    const v2d_u8 p1 = *reinterpret_cast<const v2d_u8 *>(va + b1);
    const double q1 = *reinterpret_cast<const double *>(va + b1 + 16);
    const v2d_u8 p2 = *reinterpret_cast<const v2d_u8 *>(va + b2);
    const double q2 = *reinterpret_cast<const double *>(va + b2 + 16);
    const unsigned xo = MASKED ? (mine ? (e & kMaskedColMask) * 8u : 0u) : 3u * (e & 0x7FFFu) * 8u;
    const v2d_u8 x01 = *reinterpret_cast<const v2d_u8 *>(reinterpret_cast<const char *>(xt) + xo);
    const double x2 = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(xt) + xo + 16);
    // (2) the row lanes: LPR lanes per row, old y requested now
    const int lpr_log2 = nrows <= 3 ? 4 : (nrows <= 6 ? 3 : (nrows <= 15 ? 2 : 1)); // wave-uniform; rows * LPR <= 64
    const int row = lane >> lpr_log2, t = lane & ((1 << lpr_log2) - 1);
    const bool row_lane = row < nrows;
    const int rowc = row_lane ? row : nrows - 1;
    const double yv = y_in[r0 + rowc];
    // (3) the block's three partial sums, left to right
    double s0, s1, s2;
    if (MASKED) {
        // a row's stored entries are w0, w1, w2 in column order: column b holds the (number of stored columns in front of b)-th
        // of them.  A product is formed only where an entry is stored (x may hold anything elsewhere).
        auto masked_row = [&](unsigned m, double w0, double w1, double w2) {
            const double c1 = (m & 1u) ? w1 : w0;
            const double c2 = (m & 1u) ? ((m & 2u) ? w2 : w1) : ((m & 2u) ? w1 : w0);
            double z = (m & 1u) ? w0 * x01.x : 0.0;
            z += (m & 2u) ? c1 * x01.y : 0.0;
            z += (m & 4u) ? c2 * x2 : 0.0;
            return z;
        };
        s0 = masked_row(mask & 7u, p0.x, p0.y, q0);
        s1 = masked_row((mask >> 3) & 7u, p1.x, p1.y, q1);
        s2 = masked_row((mask >> 6) & 7u, p2.x, p2.y, q2);
    } else {
        s0 = p0.x * x01.x, s1 = p1.x * x01.x, s2 = p2.x * x01.x;
        s0 += p0.y * x01.y;
        s1 += p1.y * x01.y;
        s2 += p2.y * x01.y;
        s0 += q0 * x2;
        s1 += q1 * x2;
        s2 += q2 * x2;
    }
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

// The greedy cover of one block row (three rows whose columns, minus the tile's smallest, lie in `col` at [pr[a], er[a])): blocks
// [c, c + 3) with c = the smallest column not covered yet (never beyond `limit`, so that three x entries can always be read),
// one word each (see above) into `out` if given.  Returns the number of blocks; *ok = 0 for a row with descending columns or a
// column twice, or when more than 64 blocks are needed.  One lane; always terminates (every round consumes its smallest column).
__device__ __forceinline__ int block3_greedy_cover(const int * col, int (&pr)[3], const int (&er)[3], int limit, uint32_t * out, int * ok_out)
{
    int cnt = 0, ok = 1;
    while (ok && (pr[0] < er[0] || pr[1] < er[1] || pr[2] < er[2])) {
        int s = 0x7FFFFFFF;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (pr[a] < er[a])
                s = min(s, col[pr[a]]);
        s = min(s, limit);
        unsigned mask = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int t = 0; t < 3 && pr[a] < er[a]; ++t) {
                const int d = col[pr[a]] - s;
                if (d > 2)
                    break;
                const unsigned bit = 1u << (3 * a + (d < 0 ? 0 : d));
                if (d < 0 || (mask & bit))
                    ok = 0; // descending columns, or a column twice
                mask |= bit;
                ++pr[a];
            }
        if (cnt >= kWave || s < 0 || s > (int) kMaskedColMask)
            ok = 0;
        else if (out)
            out[cnt] = (unsigned) s | (mask << kMaskedColBits) | (cnt == 0 ? kMaskedRowBegin : 0u);
        ++cnt;
    }
    *ok_out = ok;
    return cnt;
}

// Plan time: which rows belong to the same mesh node?  For a CANDIDATE matrix (rows of similar length, spmv_hip_plan_csr) one
// thread per row compares row r with row r - 1: the same node if both are longer than 16 entries, their lengths within a
// quarter, and their first and their last columns within 2 of each other (identical unless an explicit zero was dropped at
// an end; two different nodes of a mesh differ by a whole node -- 3 columns -- at least at one end; the rows of a band or of a
// stencil differ by 1 at both and form one endless group, i.e. no triples).  Bit r of `starts` = row r begins a group.
static __global__ __launch_bounds__(256) void csr_row_group_kernel(
    int rows, const int32_t * __restrict__ p, const int32_t * __restrict__ j, uint32_t * __restrict__ starts)
{
    const int r = (int) (blockIdx.x * 256 + threadIdx.x); // (the grid covers whole 64-row words: lanes past the end vote 0)
    int start = 0;
    if (r < rows) {
        start = 1;
        if (r > 0) {
            const int a0 = p[r - 1], a1 = p[r], a2 = p[r + 1];
            const int la = a1 - a0, lb = a2 - a1;
            const int hi = la > lb ? la : lb, lo = la > lb ? lb : la;
            if (lo > 16 && hi - lo <= hi / 4) {
                const int fa = j[a0], fb = j[a1], ea = j[a1 - 1], eb = j[a2 - 1];
                const int df = fa > fb ? fa - fb : fb - fa, de = ea > eb ? ea - eb : eb - ea;
                if (df <= 2 && de <= 2)
                    start = 0;
            }
        }
    }
    const unsigned long long vote = __ballot(start);
    if ((threadIdx.x & 63) == 0 && r < rows) {
        starts[r >> 5] = (uint32_t) vote; // (the words are allocated in pairs: 2 * ceil(rows / 64))
        starts[(r >> 5) + 1] = (uint32_t) (vote >> 32);
    }
}

// ... and how many groups are TRIPLES (a start, two rows that continue, a start -- or the end of the matrix -- behind them)
__device__ __forceinline__ bool group_start_bit(const uint32_t * starts, int rows, int q)
{
    return q >= rows || ((starts[q >> 5] >> (q & 31)) & 1u);
}
static __global__ __launch_bounds__(256) void csr_row_triple_count_kernel(int rows, const uint32_t * __restrict__ starts, unsigned long long * __restrict__ count)
{
    const int q = (int) (blockIdx.x * 256 + threadIdx.x);
    const bool triple = q + 3 <= rows && group_start_bit(starts, rows, q) && !group_start_bit(starts, rows, q + 1)
        && !group_start_bit(starts, rows, q + 2) && group_start_bit(starts, rows, q + 3);
    const unsigned long long vote = __ballot(triple);
    if ((threadIdx.x & 63) == 0 && vote)
        striped_add(count, 0, (unsigned long long) __builtin_popcountll(vote));
}

// Plan time, one wave per tile (spmv_hip_plan_csr_repack: the pass that has row_ptr on the device): a stream tile with
// 16-bit columns, rows of more than 16 entries and 3, 6, ... kBlockTileMaxRows rows is checked ENTRY BY ENTRY for the block
// structure described above; where it holds the tile's block stream is written and the tile marked.  A tile that is not made
// of dense aligned blocks gets a second chance as a MASKED block tile: lane b walks the three rows of block row b through the
// tile's columns (staged in LDS) and covers them greedily with blocks [c, c + 3), c = the smallest column not covered yet;
// the tile is taken if that needs at most 64 blocks holding kMaskedMinFill entries on average.  Rows with a column twice or
// with descending columns never qualify (a mask has one bit per place).
// count[0] += tiles, count[1] += their entries, count[2], count[3]: the same for the tiles no block window has claimed;
// count[4], count[5]: masked tiles among them and their entries.
static __global__ __launch_bounds__(256) void csr_block3_mark_kernel(
    int ntiles, int tile, int4 * __restrict__ desc, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    uint16_t * __restrict__ bstream, uint32_t * __restrict__ mstream, int nnz_total, int cols, int allow_masked,
    unsigned long long * __restrict__ count)
{
    __shared__ int col_all[4][512];
    __shared__ uint32_t word_all[4][kBlockTileMaxRows / 3][kWave];
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
    // WIDE: the tile's columns span 64 K or more (no 16-bit columns; its descriptor's .w is free): a candidate for masked blocks
    // with 22-bit columns
    const bool wide = !(meta & kTileMetaNarrow);
    // (a tile a window kernel has claimed is checked all the same: where block tiles turn out to be the majority the plan gives the
    // windows up -- plan_csr.hip)
    if ((d0.x & kTileFlagPartial) || !(meta & kTileMetaFast) || (meta & other) || (wide && !allow_masked)
        || nrows < 3 || nrows > kBlockTileMaxRows || nrows % 3 != 0 || n < 18 || k1 - (k0 & ~3) > tile || tile > 512)
        return;
    int cmin = d0.w;
    if (wide) { // the compress pass kept no smallest column for it
        cmin = 0x7FFFFFFF;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const int c = j[k];
            cmin = c < cmin ? c : cmin;
        }
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_xor(cmin, d);
            cmin = o < cmin ? o : cmin;
        }
    }
    // row starts in lanes 0 .. nrows (nrows <= 30)
    const int ps = p[r0 + (lane <= nrows ? lane : nrows)];
    const int len = __shfl_down(ps, 1) - ps; // lanes < nrows
    // Rows of up to 16 entries keep their one-lane-per-row, bit-exact path: a tile of such rows only is never a block tile.  (Until
    // round 6 EVERY row of a block tile had to be longer; but a short row that shares a plain tile with a longer one is added by that
    // tile's several lanes per row anyway -- 1e-10 class -- and the stored triangle of a mesh matrix mixes rows of 1 ... 100 entries.)
    if (!__any(lane < nrows && len > 16))
        return;
    const int len0 = __shfl(len, lane - lane % 3);
    int dense = !wide && n % 9 == 0 && n / 9 <= kWave && __all(lane >= nrows || (len == len0 && len % 3 == 0));
    if (dense) {
        // row by row (the row number is wave-uniform: its bounds come out of lane r with a readlane, not a shuffle per entry --
        // the first version looked every entry's row up with a loop of shuffles: 38.8 ms for the queen-like matrix's 808 K tiles)
        int ok = 1;
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
        dense = __all(ok);
    }
    if (dense) {
        // the block stream: lane = block, block rows one after the other
        const int nblk = n / 9;
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
    } else {
        // ---- masked blocks -------------------------------------------------------------------------------------------
        // (two doubles past the tile's end must be inside the value array: the multiply reads three per row and block)
        const int limit = cols - 3 - cmin; // the last place a block may start
        if (!allow_masked || (long long) k1 + 2 > (long long) nnz_total || limit < 0 || n > 512)
            return;
        int * col = col_all[wave];
        for (int k = k0 + lane; k < k1; k += kWave)
            col[k - k0] = j[k] - cmin;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nbr = nrows / 3;
        const int b = lane < nbr ? lane : nbr - 1;
        int pr[3], er[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pr[a] = __shfl(ps, 3 * b + a) - k0;
            er[a] = __shfl(ps, 3 * b + a + 1) - k0;
        }
        int cnt = 0, ok = 1;
        if (lane < nbr)
            cnt = block3_greedy_cover(col, pr, er, limit, word_all[wave][lane], &ok);
        if (!__all(ok))
            return;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { // nbr <= 10
            const int up = __shfl_up(incl, d);
            if (lane >= d)
                incl += up;
        }
        const int total = __shfl(incl, nbr - 1);
        if (total > kWave || total * kMaskedMinFill > n)
            return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t word = 0;
        for (int q = 0; q < nbr; ++q) {
            const int e1 = __shfl(incl, q), c1 = __shfl(cnt, q);
            if (lane >= e1 - c1 && lane < e1)
                word = word_all[wave][q][lane - (e1 - c1)];
        }
        // the list ends with a word whose mask is empty (the tile's place in the stream holds n / 4 > total words)
        if (lane <= total && lane < kWave)
            mstream[mask_stream_index(k0) + lane] = lane < total ? word : 0u;
    }
    if (lane == 0) {
        desc[w].z = meta | kTileMetaBlock3 | (dense ? 0 : kTileMetaBlock3Masked);
        if (wide)
            desc[w].w = cmin; // what x + cbase of the multiply starts from (a tile without 16-bit columns has no other use for it)
        striped_add(count, 0, 1ull);
        striped_add(count, 1, (unsigned long long) n);
        if (!(meta & kTileMetaBlockWin)) {
            striped_add(count, 2, 1ull);
            striped_add(count, 3, (unsigned long long) n);
        }
        if (!dense) {
            striped_add(count, 4, 1ull);
            striped_add(count, 5, (unsigned long long) n);
        }
    }
}

// Products of a group tile.  A "group entry" e = one column of one group: the tile's groups hold E = entries / D of them, numbered
// through the tile (group g starts at G_g = (its first row's start - the tile's start) / D).  Row lengths are even (the mark kernel
// checks), so a PAIR of group entries 2 t, 2 t + 1 never straddles two groups: lane l takes the pairs t = l, l + 64 (4 / D steps at
// most).  Which group a pair belongs to follows from the row starts the lanes already hold (`ps_rel`: the start of the lane's row
// relative to the tile's 4-aligned base kb; row h sits in lane h << lanes_log2): a wave-uniform walk over the groups with
// readlanes, no memory access.  Then: two 16-bit columns from the group stream, two x, D pairs of values (16-byte loads,
// consecutive across the lanes of a group row -- the load mix of tile_products_narrow with 1 / D of its gathers) and D pairs of
// products into the wave's LDS slice at the entries' own places.
// WIDE group tiles (round 6): a tile whose columns span 64 K or more has no 16-bit columns -- and no use for its slots in the 16-bit
// column stream, which therefore hold its group columns as 32-bit ABSOLUTE columns (`wide`: gt points at them, xt = x, limit =
// cols - 1).  The slots hold n / 2 words: enough for one column per pair of adjacent columns (any D) or per column with D = 4.
// An unstructured mesh with 2 unknowns per node in RCM order (Delaunay, 1 M nodes) ran at 0.65 of the roofline in plain 32-bit tiles.
template <int D>
__device__ __forceinline__ void tile_products_grouped(
    double * prod, const uint16_t * __restrict__ gt /* the tile's group stream */, const double * __restrict__ at /* a + kb */,
    const double * __restrict__ xt /* x + tile base */, unsigned limit, int ps_rel, int row_len, int lanes_log2, int nrows, int first_rel /* k0 - kb */,
    int entries, int lane, bool adjacent /* every pair is (c, c + 1): kTileMetaGroupPairs */, bool wide = false)
{
    static_assert(D == 2 || D == 4, "row groups of 2 or 4");
    constexpr int STEPS = 4 / D; // a tile holds at most 512 / D group entries = 256 / D pairs: D = 2: 2 x 64, D = 4: 1 x 64
    const int pairs = entries / (2 * D);
    int off[STEPS], len[STEPS]; // 2 t + off = the pair's place (relative to kb) in the group's FIRST row; len = the group's row length
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        off[i] = first_rel;
        len[i] = 0;
    }
    const int ngroups = nrows / D;
    for (int h = 0; h < ngroups; ++h) { // wave-uniform
        const int src = (h * D) << lanes_log2;
        const int s_h = __builtin_amdgcn_readlane(ps_rel, src), m_h = __builtin_amdgcn_readlane(row_len, src);
        const int G_h = (s_h - first_rel) / D;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const bool here = 2 * (lane + 64 * i) >= G_h;
            off[i] = here ? s_h - G_h : off[i];
            len[i] = here ? m_h : len[i];
        }
    }
    unsigned c0[STEPS], c1[STEPS];
    v2d_u8 v[STEPS][D];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        int t = lane + 64 * i;
        t = t < pairs ? t : pairs - 1; // idle lanes repeat the last pair's loads and store nothing
        if (wide) { // (wave-uniform)
            const uint32_t * g32 = reinterpret_cast<const uint32_t *>(gt);
            c0[i] = adjacent ? g32[t] : g32[2 * t];
            c1[i] = adjacent ? 0u : g32[2 * t + 1];
        } else {
            c0[i] = adjacent ? gt[t] : gt[2 * t]; // (adjacent pairs: the stream holds the first column of every pair only)
            c1[i] = adjacent ? 0u : gt[2 * t + 1];
        }
#pragma unroll
        for (int a = 0; a < D; ++a)
            v[i][a] = *reinterpret_cast<const v2d_u8 *>(at + 2 * t + off[i] + a * len[i]);
    }
    const char * xb = reinterpret_cast<const char *>(xt);
    double x0[STEPS], x1[STEPS];
    if (adjacent) { // wave-uniform: one 16-byte gather per pair of columns
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const v2d_u8 xx = *reinterpret_cast<const v2d_u8 *>(xb + (min(c0[i], limit - 1u) << 3));
            x0[i] = xx.x;
            x1[i] = xx.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            x0[i] = *reinterpret_cast<const double *>(xb + (min(c0[i], limit) << 3));
            x1[i] = *reinterpret_cast<const double *>(xb + (min(c1[i], limit) << 3));
        }
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int t = lane + 64 * i;
        if (t < pairs) {
#pragma unroll
            for (int a = 0; a < D; ++a)
                *reinterpret_cast<v2d_u8 *>(prod + 2 * t + off[i] + a * len[i]) = v2d_u8{v[i][a].x * x0[i], v[i][a].y * x1[i]};
        }
    }
}

// Plan time, one wave per tile: a stream tile with 16-bit columns and rows of more than 16 entries is a GROUP tile if its rows come in
// groups of D equally long rows with the same columns, entry by entry; the first row's columns (offsets from the tile's smallest
// column) are then copied to the group stream and the tile marked.  Counts as csr_block3_mark_kernel's: count[0], [1] tiles and
// entries, count[2], [3] those no block window has claimed.
// where a WIDE group tile keeps its 32-bit group columns: its own slots of the 16-bit column stream, from the first even one on
__host__ __device__ __forceinline__ int wide_group_first_slot(int k0) { return (k0 + 1) & ~1; }

static __global__ __launch_bounds__(256) void csr_group_mark_kernel(
    int ntiles, int tile, int d, int4 * __restrict__ desc, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    uint16_t * __restrict__ gstream, unsigned long long * __restrict__ count, uint16_t * __restrict__ j16 = nullptr, int cols = 0)
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
    const int other = kTileMetaShifted | kTileMetaXWin | kTileMetaXSeg | kTileMetaPattern | (1 << 21) /* balanced tiles */;
    // WIDE: no 16-bit columns (the tile's columns span 64 K or more): its group columns go, 32 bits each and absolute, into its own
    // unused slots of the 16-bit stream -- unless a window kernel has claimed the tile (segment windows keep their slots there)
    const bool wide = !(meta & kTileMetaNarrow);
    if ((d0.x & kTileFlagPartial) || !(meta & kTileMetaFast) || (meta & other) || (wide && (!j16 || (meta & kTileMetaBlockWin) || cols >= (1 << 29)))
        || nrows < d || nrows > kBlockTileMaxRows || nrows % d != 0 || k1 - (k0 & ~3) > tile || tile > 512 || (d != 2 && d != 4))
        return;
    const int cmin = wide ? 0 : d0.w;
    const int ps = p[r0 + (lane <= nrows ? lane : nrows)]; // row starts in lanes 0 .. nrows (nrows <= 30)
    const int len = __shfl_down(ps, 1) - ps;                // lanes < nrows
    const int len0 = __shfl(len, lane - lane % d);
    if (!__all(lane >= nrows || (len > 16 && len == len0 && len % 2 == 0)))
        return; // (rows of up to 16 entries keep their one-lane-per-row, bit-exact path; even lengths: the multiply takes the columns in pairs)
    int ok = 1;
    for (int r = 0; r < nrows; ++r) {
        const int a = r % d;
        if (a == 0)
            continue;
        const int start = __shfl(ps, r), rl = __shfl(len, r);
        for (int k = start + lane; k < start + rl; k += kWave)
            ok &= j[k] == j[k - a * rl];
    }
    if (!__all(ok))
        return;
    // is every pair of the first rows' columns (c, c + 1)?
    int adjacent = 1;
    for (int r = 0; r < nrows; r += d) {
        const int start = __shfl(ps, r), rl = __shfl(len, r);
        for (int q = 2 * lane + 1; q < rl; q += 2 * kWave)
            adjacent &= j[start + q] == j[start + q - 1] + 1;
    }
    adjacent = __all(adjacent);
    uint32_t * g32 = nullptr;
    if (wide) { // room for one 32-bit column per group column (or per pair) in the tile's own 16-bit slots?
        const int first = wide_group_first_slot(k0), words = ((k1 & ~1) - first) / 2, need = adjacent ? n / (2 * d) : n / d;
        if (need > words)
            return;
        g32 = reinterpret_cast<uint32_t *>(j16 + first);
    }
    // the group stream: the first rows' columns, group after group -- every column, or (adjacent pairs) the first of each pair
    const int base = group_stream_index(k0, d);
    for (int r = 0; r < nrows; r += d) {
        const int start = __shfl(ps, r), rl = __shfl(len, r);
        const int g0 = (start - k0) / d; // what the groups in front of this one hold (even: row lengths are)
        if (adjacent) {
            for (int q = 2 * lane; q < rl; q += 2 * kWave) {
                if (wide)
                    g32[(g0 + q) / 2] = (uint32_t) j[start + q];
                else
                    gstream[base + (g0 + q) / 2] = (uint16_t) (j[start + q] - cmin);
            }
        } else {
            for (int q = lane; q < rl; q += kWave) {
                if (wide)
                    g32[g0 + q] = (uint32_t) j[start + q];
                else
                    gstream[base + g0 + q] = (uint16_t) (j[start + q] - cmin);
            }
        }
    }
    if (lane == 0) {
        desc[w].z = meta | kTileMetaBlock3 | kTileMetaGroupRows | (adjacent ? kTileMetaGroupPairs : 0);
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
