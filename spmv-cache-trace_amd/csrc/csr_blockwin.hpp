// csr_blockwin.hpp -- the one-ring block window: x staged through LDS for 16 tiles at a time (unstructured bands).
#pragma once

#include "csr_wavetile.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// Block window: x staged through LDS for a whole workgroup.  A row whose columns are scattered
// over a band (a finite-element matrix: 27 blocks of 3 columns anywhere within +-3000 of the
// diagonal) gets nothing from a per-tile window -- no x entry is used twice inside a tile --
// and its gather runs at the rate at which 128-byte lines come out of L2 (measured: 260 of
// 465 us).  Sixteen consecutive tiles, however, share one window of a few thousand columns:
// the workgroup (16 waves, one tile each) reads it once with coalesced loads into 48 KB of LDS
// and every wave gathers from there.  One workgroup per CU fits (48 KB window + 16 product
// slices), so the stream runs at half the usual occupancy: only blocks whose window fits and
// whose tiles have no cheaper path are marked (csr_blockwin_mark_kernel), and the kernel is
// only launched when they are the majority; csr_wavetile_kernel skips the marked tiles.
// ---------------------------------------------------------------------------------
constexpr int kBlockWinSlots = 8192; // doubles (the ring of csr_blockwin_stream_kernel)
constexpr int kBlockWinTiles = 16;

template <int TILE>
__global__ __launch_bounds__(1024) void csr_blockwin_kernel(
    int ntiles, const int4 * __restrict__ desc, const int2 * __restrict__ blocks,
    const int32_t * __restrict__ p, const uint16_t * __restrict__ j16, const double * __restrict__ a,
    const double * __restrict__ x, const double * y_in, double * y)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr int QUADS = TILE / 256;
    __shared__ double xwin[kBlockWinSlots];
    __shared__ __attribute__((aligned(16))) double prod_all[kBlockWinTiles][TILE + 4];

    const int2 bd = blocks[blockIdx.x];
    const int xbase = __builtin_amdgcn_readfirstlane(bd.x);
    const int span = __builtin_amdgcn_readfirstlane(bd.y);
    if (span <= 0)
        return; // not a window block: its tiles went through csr_wavetile_kernel
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    const int w = blockIdx.x * kBlockWinTiles + wave;
    const bool active = w < ntiles; // wave-uniform
    double * prod = prod_all[wave];

    // (1) the tile's own loads first: they do not depend on the window
    int r0 = 0, k0 = 0, meta = 0, cbase = 0, nrows = 1, kb = 0, last = 0, maxlen = 1, lanes_log2 = 0;
    int ps = 0, pe = 0, psB = 0, peB = 0;
    double yv = 0.0, yvB = 0.0;
    bool second = false;
    v2u c[QUADS];
    v2d va[QUADS], vb[QUADS];
    if (active) {
        const int4 d0 = desc[w];
        const int4 d1 = desc[w + 1];
        r0 = __builtin_amdgcn_readfirstlane(d0.x & ~kTileFlagPartial);
        k0 = __builtin_amdgcn_readfirstlane(d0.y);
        meta = __builtin_amdgcn_readfirstlane(d0.z);
        cbase = __builtin_amdgcn_readfirstlane(d0.w);
        const int r1 = __builtin_amdgcn_readfirstlane(d1.x & ~kTileFlagPartial);
        const int k1 = __builtin_amdgcn_readfirstlane(d1.y);
        maxlen = meta & 0xFFFF;
        lanes_log2 = (meta >> kTileMetaLanesShift) & 0x7;
        nrows = r1 - r0;
        kb = k0 & ~3;
        last = (k1 - 1 - kb) & ~3;
        const int sub = lane >> lanes_log2;
        const int rowi = sub < nrows ? sub : nrows - 1;
        if (meta & kTileMetaUniform) {
            ps = k0 + rowi * maxlen;
            pe = ps + maxlen;
        } else {
            ps = p[r0 + rowi];
            pe = p[r0 + rowi + 1];
        }
        yv = y_in[r0 + rowi];
        second = nrows > kWave;
        if (second) {
            const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
            if (meta & kTileMetaUniform) {
                psB = k0 + rowB * maxlen;
                peB = psB + maxlen;
            } else {
                psB = p[r0 + rowB];
                peB = p[r0 + rowB + 1];
            }
            yvB = y_in[r0 + rowB];
        }
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            int o = 256 * q + 4 * lane;
            o = o < last ? o : last;
            c[q] = *reinterpret_cast<const v2u *>(j16 + kb + o);
            va[q] = *reinterpret_cast<const v2d *>(a + kb + o);
            vb[q] = *reinterpret_cast<const v2d *>(a + kb + o + 2);
        }
    }
    // (2) the block's window of x, by all 1024 threads
    for (int i = (int) threadIdx.x; i < span; i += 1024)
        xwin[i] = x[xbase + i];
    __syncthreads();
    if (!active)
        return;
    // (3) products from the window
    const unsigned off = (unsigned) (cbase - xbase), wlimit = (unsigned) (span - 1);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            const unsigned c0 = min(off + (c[q].x & 0xFFFFu), wlimit), c1 = min(off + (c[q].x >> 16), wlimit);
            const unsigned c2 = min(off + (c[q].y & 0xFFFFu), wlimit), c3 = min(off + (c[q].y >> 16), wlimit);
            const double q0 = va[q].x * xwin[c0];
            const double q1 = va[q].y * xwin[c1];
            const double q2 = vb[q].x * xwin[c2];
            const double q3 = vb[q].y * xwin[c3];
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (4) row sums, as in csr_wavetile_kernel
    const int sub = lane >> lanes_log2;
    const int part = lane & ((1 << lanes_log2) - 1);
    const int s = ps - kb, e_row = pe - kb;
    double z;
    if (lanes_log2 == 0) {
        z = tile_row_sum<1>(prod, s, e_row, 0, maxlen);
    } else {
        const int trips = (maxlen + (1 << lanes_log2) - 1) >> lanes_log2;
        switch (lanes_log2) {
        case 1: z = tile_row_sum<2>(prod, s, e_row, part, trips); break;
        case 2: z = tile_row_sum<4>(prod, s, e_row, part, trips); break;
        case 3: z = tile_row_sum<8>(prod, s, e_row, part, trips); break;
        case 4: z = tile_row_sum<16>(prod, s, e_row, part, trips); break;
        case 5: z = tile_row_sum<32>(prod, s, e_row, part, trips); break;
        default: z = tile_row_sum<64>(prod, s, e_row, part, trips); break;
        }
    }
    if (sub < nrows && part == 0)
        y[r0 + sub] = yv + z;
    if (second) {
        const double zB = tile_row_sum<1>(prod, psB - kb, peB - kb, 0, maxlen);
        if (lane + kWave < nrows)
            y[r0 + lane + kWave] = yvB + zB;
    }
}

// The same with persistent workgroups and a sliding window.  Consecutive blocks of a band need
// almost the same columns (the window moves on by the block's rows), so a workgroup that walks
// through consecutive blocks keeps x in a ring of 8192 LDS slots (slot = column mod 8192) and only
// loads what is new; and because the next block's streams -- and that window increment -- are
// requested before the current block is multiplied, something is always in flight although only
// one workgroup fits a CU.  Two barriers per block: before the ring is written (the previous
// block's gathers are done) and after.
constexpr int kBlockRing = 8192;

template <int QUADS>
struct BwTile {
    int r0, k0, kb, last, nrows, maxlen, lanes_log2, cbase;
    int ps, pe, psB, peB;
    double yv, yvB;
    unsigned cx[QUADS], cy[QUADS];
    v2d va[QUADS], vb[QUADS];
    bool valid, second;
};

template <int TILE>
__device__ __forceinline__ void bw_load_tile(
    BwTile<TILE / 256> & t, int w, int ntiles, const int4 * __restrict__ desc, const int32_t * __restrict__ p,
    const uint16_t * __restrict__ j16, const double * __restrict__ a, const double * y, int lane)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr int QUADS = TILE / 256;
    t.valid = w < ntiles;
    if (!t.valid)
        return;
    const int4 d0 = desc[w];
    const int4 d1 = desc[w + 1];
    const int meta = __builtin_amdgcn_readfirstlane(d0.z);
    t.valid = (meta & kTileMetaBlockWin) != 0;
    if (!t.valid)
        return;
    t.r0 = __builtin_amdgcn_readfirstlane(d0.x & ~kTileFlagPartial);
    t.k0 = __builtin_amdgcn_readfirstlane(d0.y);
    t.cbase = __builtin_amdgcn_readfirstlane(d0.w);
    const int r1 = __builtin_amdgcn_readfirstlane(d1.x & ~kTileFlagPartial);
    const int k1 = __builtin_amdgcn_readfirstlane(d1.y);
    t.maxlen = meta & 0xFFFF;
    t.lanes_log2 = (meta >> kTileMetaLanesShift) & 0x7;
    t.nrows = r1 - t.r0;
    t.kb = t.k0 & ~3;
    t.last = (k1 - 1 - t.kb) & ~3;
    const int sub = lane >> t.lanes_log2;
    const int rowi = sub < t.nrows ? sub : t.nrows - 1;
    const bool uniform = (meta & kTileMetaUniform) != 0;
    if (uniform) {
        t.ps = t.k0 + rowi * t.maxlen;
        t.pe = t.ps + t.maxlen;
    } else {
        t.ps = p[t.r0 + rowi];
        t.pe = p[t.r0 + rowi + 1];
    }
    t.yv = y[t.r0 + rowi];
    t.second = t.nrows > kWave;
    t.psB = t.peB = 0;
    t.yvB = 0.0;
    if (t.second) {
        const int rowB = lane + kWave < t.nrows ? lane + kWave : t.nrows - 1;
        if (uniform) {
            t.psB = t.k0 + rowB * t.maxlen;
            t.peB = t.psB + t.maxlen;
        } else {
            t.psB = p[t.r0 + rowB];
            t.peB = p[t.r0 + rowB + 1];
        }
        t.yvB = y[t.r0 + rowB];
    }
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < t.last ? o : t.last;
        const v2u c = *reinterpret_cast<const v2u *>(j16 + t.kb + o);
        t.cx[q] = c.x;
        t.cy[q] = c.y;
        t.va[q] = *reinterpret_cast<const v2d *>(a + t.kb + o);
        t.vb[q] = *reinterpret_cast<const v2d *>(a + t.kb + o + 2);
    }
}

template <int TILE>
__device__ __forceinline__ void bw_compute_tile(
    const BwTile<TILE / 256> & t, double * prod, const double * xring, double * y, int lane)
{
    constexpr int QUADS = TILE / 256;
    const unsigned base = (unsigned) t.cbase;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= t.last) {
            // columns of the tile proper lie inside the ring's window; entries of neighbouring tiles
            // that share a quad read some slot or other and are never summed
            const unsigned c0 = (base + (t.cx[q] & 0xFFFFu)) & (kBlockRing - 1), c1 = (base + (t.cx[q] >> 16)) & (kBlockRing - 1);
            const unsigned c2 = (base + (t.cy[q] & 0xFFFFu)) & (kBlockRing - 1), c3 = (base + (t.cy[q] >> 16)) & (kBlockRing - 1);
            const double q0 = t.va[q].x * xring[c0];
            const double q1 = t.va[q].y * xring[c1];
            const double q2 = t.vb[q].x * xring[c2];
            const double q3 = t.vb[q].y * xring[c3];
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int sub = lane >> t.lanes_log2;
    const int part = lane & ((1 << t.lanes_log2) - 1);
    const int s = t.ps - t.kb, e_row = t.pe - t.kb;
    double z;
    if (t.lanes_log2 == 0) {
        z = tile_row_sum<1>(prod, s, e_row, 0, t.maxlen);
    } else {
        const int trips = (t.maxlen + (1 << t.lanes_log2) - 1) >> t.lanes_log2;
        switch (t.lanes_log2) {
        case 1: z = tile_row_sum<2>(prod, s, e_row, part, trips); break;
        case 2: z = tile_row_sum<4>(prod, s, e_row, part, trips); break;
        case 3: z = tile_row_sum<8>(prod, s, e_row, part, trips); break;
        case 4: z = tile_row_sum<16>(prod, s, e_row, part, trips); break;
        case 5: z = tile_row_sum<32>(prod, s, e_row, part, trips); break;
        default: z = tile_row_sum<64>(prod, s, e_row, part, trips); break;
        }
    }
    if (sub < t.nrows && part == 0)
        y[t.r0 + sub] = t.yv + z;
    if (t.second) {
        const double zB = tile_row_sum<1>(prod, t.psB - t.kb, t.peB - t.kb, 0, t.maxlen);
        if (lane + kWave < t.nrows)
            y[t.r0 + lane + kWave] = t.yvB + zB;
    }
    // the product slice is reused by this wave's next tile: its reads above come first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int TILE>
__global__ __launch_bounds__(1024) void csr_blockwin_stream_kernel(
    int ntiles, int nblocks, int blocks_per_group, const int4 * __restrict__ desc, const int2 * __restrict__ blocks,
    const int32_t * __restrict__ p, const uint16_t * __restrict__ j16, const double * __restrict__ a,
    const double * __restrict__ x, const double * y_in, double * y)
{
    constexpr int XS = kBlockWinSlots / 1024; // window increments a thread may have to carry
    __shared__ double xring[kBlockRing];
    __shared__ __attribute__((aligned(16))) double prod_all[kBlockWinTiles][TILE + 4];
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    const int tid = (int) threadIdx.x;
    double * prod = prod_all[wave];
    const int b_begin = (int) blockIdx.x * blocks_per_group;
    const int b_end = min(nblocks, b_begin + blocks_per_group);
    if (b_begin >= b_end)
        return;

    int wlo = 0, whi = 0; // columns [wlo, whi) are in the ring (wave-uniform)
    BwTile<TILE / 256> cur, nxt;
    double xs[XS];
    int xs_from = 0, xs_hi = 0, xs_lo = 0; // the increment carried in xs belongs to window [xs_lo, xs_hi)
    // prologue: the first block's streams and its whole window
    {
        const int2 bd = blocks[b_begin];
        const int span = __builtin_amdgcn_readfirstlane(bd.y);
        xs_lo = xs_from = __builtin_amdgcn_readfirstlane(bd.x);
        xs_hi = xs_lo + (span > 0 ? span : 0);
        bw_load_tile<TILE>(nxt, b_begin * kBlockWinTiles + wave, ntiles, desc, p, j16, a, y_in, lane);
#pragma unroll
        for (int k = 0; k < XS; ++k) {
            const int i = xs_from + tid + 1024 * k;
            xs[k] = i < xs_hi ? x[i] : 0.0;
        }
    }
    for (int b = b_begin; b < b_end; ++b) {
        cur = nxt;
        const bool window = xs_hi > xs_lo; // this block has a window
        __syncthreads(); // the previous block's gathers are done: ring slots may be overwritten
        if (window) {
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const int i = xs_from + tid + 1024 * k;
                if (i < xs_hi)
                    xring[i & (kBlockRing - 1)] = xs[k];
            }
            wlo = xs_lo;
            whi = xs_hi;
        } else {
            wlo = whi = 0;
        }
        __syncthreads();
        // requests for the next block: its tiles' streams and what its window adds to the ring
        if (b + 1 < b_end) {
            const int2 bd = blocks[b + 1];
            const int span = __builtin_amdgcn_readfirstlane(bd.y);
            xs_lo = __builtin_amdgcn_readfirstlane(bd.x);
            xs_hi = xs_lo + (span > 0 ? span : 0);
            // columns already in the ring stay valid if the new window starts inside the old one
            xs_from = (whi > wlo && xs_lo >= wlo && xs_lo <= whi) ? max(whi, xs_lo) : xs_lo;
            bw_load_tile<TILE>(nxt, (b + 1) * kBlockWinTiles + wave, ntiles, desc, p, j16, a, y_in, lane);
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const int i = xs_from + tid + 1024 * k;
                xs[k] = i < xs_hi ? x[i] : 0.0;
            }
        } else {
            nxt.valid = false;
            xs_lo = xs_hi = xs_from = 0;
        }
        if (window && cur.valid)
            bw_compute_tile<TILE>(cur, prod, xring, y, lane);
    }
}

// Plan-time: one workgroup per 16 consecutive tiles.  The block gets a window if every tile is a
// plain narrow fast tile (no shifted tile, no per-tile window: those are cheaper), the union of
// their column ranges fits kBlockWinSlots and has at least as many entries as slots.  With
// apply == 0 only counts[3] += tiles that would be marked; with apply != 0 the tiles are marked
// and blocks[b] = {first column, slots} (0 slots = no window).
static __global__ __launch_bounds__(1024) void csr_blockwin_mark_kernel(
    int ntiles, int tile, int4 * __restrict__ desc, const uint16_t * __restrict__ j16,
    int2 * __restrict__ blocks, int * __restrict__ counts, int apply)
{
    __shared__ int s_min[kBlockWinTiles], s_max[kBlockWinTiles], s_ok[kBlockWinTiles], s_entries[kBlockWinTiles];
    __shared__ int s_decision[2];
    const int wave = (int) threadIdx.x >> 6;
    const int lane = (int) __lane_id();
    const int w = blockIdx.x * kBlockWinTiles + wave;
    int ok = 1, cmin = 0x7FFFFFFF, cmax = -1, entries = 0;
    if (w < ntiles) {
        const int4 d0 = desc[w];
        const int k0 = d0.y, k1 = desc[w + 1].y;
        const int m = d0.z;
        ok = !(d0.x & kTileFlagPartial) && (m & kTileMetaNarrow) && (m & kTileMetaFast)
             && !(m & (kTileMetaShifted | kTileMetaXWin | kTileMetaXSeg | kTileMetaPattern)) && k1 - (k0 & ~3) <= tile;
        if (ok) {
            int mx = 0;
            for (int k = k0 + lane; k < k1; k += kWave)
                mx = max(mx, (int) j16[k]);
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1)
                mx = max(mx, __shfl_xor(mx, d));
            cmin = d0.w;
            cmax = d0.w + mx;
            entries = k1 - k0;
        }
    }
    if (lane == 0) {
        s_min[wave] = cmin;
        s_max[wave] = cmax;
        s_ok[wave] = ok;
        s_entries[wave] = entries;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int all = 1, lo = 0x7FFFFFFF, hi = -1, n = 0, tiles = 0;
        for (int t = 0; t < kBlockWinTiles; ++t) {
            all &= s_ok[t];
            if (s_max[t] >= 0) {
                lo = min(lo, s_min[t]);
                hi = max(hi, s_max[t]);
                n += s_entries[t];
                ++tiles;
            }
        }
        const int span = hi - lo + 1;
        const int yes = all && tiles > 0 && span <= kBlockWinSlots && n >= span;
        s_decision[0] = yes ? lo : 0;
        s_decision[1] = yes ? span : 0;
        if (yes)
            striped_add(counts, 3, tiles);
        if (apply)
            blocks[blockIdx.x] = make_int2(yes ? lo : 0, yes ? span : 0);
    }
    __syncthreads();
    if (apply && s_decision[1] > 0 && w < ntiles && lane == 0)
        desc[w].z |= kTileMetaBlockWin;
}

} // namespace spmv
