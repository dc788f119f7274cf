// csr_rowgroup.hpp -- row-group tiles: the interior of a stencil or a band whose rows hold 17 ... 64 entries (the 27-point
// stencil, the KKT-like matrix's rows of 27 and 33, bands of 21 ... 61).
//
// The reference's loop (src/matrix/csr-matrix-spmv.cpp:21-33) walks a row left to right.  csr_wavetile_kernel deals such a
// tile's entries to the lanes four at a time, whatever row they belong to, parks the products in LDS and has 2 or 4 lanes per
// row read them back -- 36 LDS instructions per wave (and, when this kernel was written, 26.6 KB of LDS per workgroup: 6 waves
// per SIMD; 18.5 KB and 8 since the window lives inside the product slice)
// (profiles/r04_prof_kkt_csr_extra_summary.md: two thirds of the wave cycles wait for memory with that few waves in flight).
//
// Here a tile that is uniform (all rows equally long), shifted (every row has the first row's columns moved along with it) and
// has an x window -- the marks csr_tile_compress_kernel / csr_pattern_assign_kernel set -- is multiplied by G lanes per row,
// G = as many as the tile's rows leave room for in the wave (2 ... 8): a lane loads E = ceil(len / G) <= 12 CONSECUTIVE values
// of its row (straight from the value array, 16 bytes at a time), multiplies them by x from the wave's window and adds them up
// in registers; the G partial sums of a row meet through the lane crossbar.  No product is parked: 5 LDS writes (window,
// positions) + 3 + E reads + 3 crossbar steps per wave, 9 KB of LDS per workgroup, 8 waves per SIMD.  The window and the
// positions come from the tile's pattern record exactly as in tile_products_xseg / tile_products_xwin (csr_wavetile.hpp).
//
// OPT-IN (SPMV_HIP_FLAG_ROW_GROUPS): measured SLOWER than csr_wavetile_kernel's x-window variant -- KKT-like matrix 797 vs 740
// us, 27 diagonals 181 vs 176 us (profiles/r04_rowgroup_ab_*.log).  A lane that owns 72 consecutive bytes makes each of the
// wave's five value-load instructions touch all 36 cache lines of the tile: 180 line look-ups per tile instead of 36, and with
// 32 waves per CU the 4.6 KB tiles in flight (147 KB) do not stay in the 32 KB vector L1 between the first and the fifth touch.
// (Block tiles, csr_blocktile.hpp, own 24 consecutive bytes per lane -- twice over, not five times -- and do win.)
//
// Which tiles: the plan keeps a LIST of them (plan_account, plan_csr.hip) and one of the others; the kernel below runs over
// the first, csr_wavetile_kernel's LIST variant over the second.  Only plans where the row-group tiles are the majority do.
//
// Summation order: E entries left to right in a lane, then a scan over the row's G lanes -- not the reference's order:
// 1e-10 class, like every row of more than 16 entries on the default path (never under SPMV_HIP_FLAG_EXACT_ORDER).
#pragma once

#include "csr_blocktile.hpp" // v2d_u8
#include "csr_wavetile.hpp"  // PeerY, y_store

namespace spmv {

constexpr int kRowGroupMinLen = 17, kRowGroupMaxLen = 64;
constexpr int kRowGroupPerLane = 12; // most entries of a row one lane takes
// the last lanes of a tile may load up to this many values past the tile's end (never used): tiles that close to the end of
// the value array stay with csr_wavetile_kernel
constexpr int kRowGroupOverRead = 32;

// lanes per row: as many as the tile's rows leave room for in the wave, at most 8 (rows <= 30: at least 2)
__host__ __device__ __forceinline__ int rowgroup_lanes(int rows) { return rows <= 8 ? 8 : kWave / rows; }

// what plan_account asks of a tile (meta = descriptor .z, entries / rows of the tile, k1 = its end in the arrays)
inline bool rowgroup_tile(int row_field, int meta, long long entries, long long rows, long long k1, long long nnz)
{
    const int len = meta & 0xFFFF;
    if ((row_field & kTileFlagPartial) || !(meta & kTileMetaFast) || !(meta & kTileMetaUniform) || !(meta & kTileMetaShifted)
        || !(meta & kTileMetaPattern) || !(meta & (kTileMetaXSeg | kTileMetaXWin)) || (meta & (kTileMetaBlockWin | kTileMetaBlock3)))
        return false;
    if (len < kRowGroupMinLen || len > kRowGroupMaxLen || rows < 2 || rows > 32 || entries != rows * len || entries > 512)
        return false;
    const int G = rowgroup_lanes((int) rows);
    return (len + G - 1) / G <= kRowGroupPerLane && k1 + kRowGroupOverRead <= nnz;
}

// One tile with G lanes per row and E <= EMAX entries per lane (EMAX = 9: four 16-byte loads and one of 8 bytes; 12: six).
template <bool PEER, int EMAX>
__device__ __forceinline__ void rowgroup_tile_body(
    double * xw, unsigned long long * tab, const int32_t * __restrict__ pat, const double * __restrict__ a, const double * __restrict__ x,
    const double * y_in, double * y, const PeerY & peers, int cols, int r0, int k0, int nrows, int len, int chunks, bool runs, int G, int E,
    unsigned inv_g, int lane)
{
    static_assert(EMAX == 9 || EMAX == 12, "value loads: pairs, and one single for an odd count");
    const unsigned inv_e = 65536u / (unsigned) E + 1u;
    const int rho = (int) (((unsigned) lane * inv_g) >> 16);
    const int t = lane - rho * G;
    const int row = rho < nrows ? rho : nrows - 1; // idle lanes repeat the last row's loads and store nothing
    const int p0 = t * E;
    const int left = len - p0;
    const int cnt = left < 0 ? 0 : (left < E ? left : E); // (G lanes of E entries may be more than the row needs)
    const int tq = (int) (((unsigned) lane * inv_e) >> 16), ti = lane - tq * E;
    const int tab_at = lane < G * E ? 12 * tq + ti : 96 + (lane & 3);

    // (1) every load that depends on the descriptor only.  No branches: all four window chunks (the record's unused slots
    // name a valid x entry), both forms of the positions, EMAX values (those past the lane's share are not used)
    const int relmin = __builtin_amdgcn_readfirstlane(pat[3]);
    int so[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
        so[ch] = pat[kPatSrc + 64 * ch + lane];
    const int pl = lane < len ? lane : len - 1;
    const unsigned tv_runs = reinterpret_cast<const uint16_t *>(pat + kPatXoff)[pl];
    const int tv_rel = pat[kPatRel + pl];
    const double yv = y_in[r0 + row];
    // (2) the window of x, then the values: the pattern record is cache-resident, x mostly (neighbouring tiles share it), the
    // values never -- they are requested last and are still on their way while the window is written
    __builtin_amdgcn_sched_barrier(0);
    double xs[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        int c = r0 + (runs ? so[ch] : relmin + 64 * ch + lane);
        c = ch < chunks ? c : r0 + relmin;             // chunks the window does not have: one address for the whole wave
        c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c); // padding slots of the last chunk
        xs[ch] = x[c];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double * ap = a + k0 + row * len + p0;
    v2d_u8 va[EMAX / 2];
#pragma unroll
    for (int i = 0; i < EMAX / 2; ++i)
        va[i] = *reinterpret_cast<const v2d_u8 *>(ap + 2 * i);
    const double v_last = (EMAX & 1) ? ap[EMAX - 1] : 0.0; // (a pair here would leave a register half unused, and the compiler
                                                            // would reuse it while the load is in flight: a full wait)
    __builtin_amdgcn_sched_barrier(0);
    // positions: lane l holds row position l; the lane (tq) that multiplies it finds it as its entry ti, in a 24-byte record of
    // twelve 16-bit window positions per lane of a row (three aligned 8-byte reads instead of up to twelve 2-byte ones)
    // (lanes past G * E <= 64 write a spare record: a store every lane executes keeps the loads above where they are)
    reinterpret_cast<uint16_t *>(tab)[tab_at] = (uint16_t) (runs ? tv_runs : (unsigned) (tv_rel - relmin));
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
        xw[64 * ch + lane] = xs[ch];
    // same-wave LDS operations execute in order; the fences only pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) the lane's entries, left to right (entries past its share add +0.0: an identity)
    const unsigned long long w0 = tab[3 * t], w1 = tab[3 * t + 1], w2 = tab[3 * t + 2];
    double xv[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const unsigned long long wq = i < 4 ? w0 : (i < 8 ? w1 : w2);
        const unsigned slot = ((unsigned) (wq >> (16 * (i & 3))) & 0xFFFFu) + (unsigned) row;
        xv[i] = xw[slot & 255u];
    }
    double z = 0.0;
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const double v = ((EMAX & 1) && i == EMAX - 1) ? v_last : ((i & 1) ? va[i / 2].y : va[i / 2].x);
        const double q = v * xv[i];
        z += i < cnt ? q : 0.0;
    }
    // (4) the row's G partial sums: an inclusive scan over the row's lanes, its last lane holds the total
    double s = z;
    {
        const double o = lane_up(s, 1);
        s += t >= 1 ? o : 0.0;
    }
    if (G > 2) {
        const double o = lane_up(s, 2);
        s += t >= 2 ? o : 0.0;
    }
    if (G > 4) {
        const double o = lane_up(s, 4);
        s += t >= 4 ? o : 0.0;
    }
    if (t == G - 1 && rho < nrows)
        y_store<PEER, false>(y, peers, r0 + rho, yv + s);
}

template <bool PEER>
__global__ __launch_bounds__(256, 8) void csr_rowgroup_kernel(
    int nlist, const int32_t * __restrict__ list, const int4 * __restrict__ desc, const double * __restrict__ a,
    const double * __restrict__ x, const double * y_in, double * y, int cols, const int32_t * __restrict__ patterns,
    PeerY peers = PeerY{})
{
    __shared__ double xwin_all[4][256];
    __shared__ unsigned long long tab_all[4][3 * 8 + 1]; // 8 lanes per row x 12 window positions of 16 bits, one spare
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    int w = (int) blockIdx.x * 4 + wave;
    if (w >= nlist)
        return; // whole wave leaves; no workgroup barrier in this kernel
    w = __builtin_amdgcn_readfirstlane(list[w]);
    const TilePair dp = load_tile_pair(desc, w);
    const int r0 = __builtin_amdgcn_readfirstlane(dp.d0.x & ~kTileFlagPartial);
    const int k0 = __builtin_amdgcn_readfirstlane(dp.d0.y);
    const int meta = __builtin_amdgcn_readfirstlane(dp.d0.z);
    const int32_t * pat = patterns + (size_t) __builtin_amdgcn_readfirstlane(dp.d0.w) * kPatStride;
    const int nrows = __builtin_amdgcn_readfirstlane(dp.d1.x & ~kTileFlagPartial) - r0;
    const int len = meta & 0xFFFF;
    const int chunks = ((meta >> kTileMetaXChunksShift) & 3) + 1;
    const bool runs = (meta & kTileMetaXSeg) != 0; // window of runs; otherwise one contiguous range of x
    // the lanes of a row: G of them, E entries each (wave-uniform; q / G and q / E for q < 128 by multiplication)
    const int G = rowgroup_lanes(nrows);
    const unsigned inv_g = 65536u / (unsigned) G + 1u;
    const int E = (int) (((unsigned) (len + G - 1) * inv_g) >> 16);
    if (E <= 9)
        rowgroup_tile_body<PEER, 9>(xwin_all[wave], tab_all[wave], pat, a, x, y_in, y, peers, cols, r0, k0, nrows, len, chunks, runs, G, E, inv_g, lane);
    else
        rowgroup_tile_body<PEER, 12>(xwin_all[wave], tab_all[wave], pat, a, x, y_in, y, peers, cols, r0, k0, nrows, len, chunks, runs, G, E, inv_g, lane);
}

} // namespace spmv
