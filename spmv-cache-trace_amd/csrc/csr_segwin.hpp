// csr_segwin.hpp -- segment windows: x staged through LDS for rows whose columns cluster in a few far-apart
// column ranges (any mesh in natural ordering; replaces the gather of src/matrix/csr-matrix-spmv.cpp:21-33's
// x[j[k]] for such rows).
//
// A finite-element or finite-difference row on a 3-D mesh touches three planes of the mesh: its columns sit in
// three clusters tens of thousands of columns apart (Queen_4147-like: 23 K; a KKT system's state rows: a diagonal
// entry plus three planes eight million columns away).  No x entry is used twice inside a 512-entry tile, so a
// per-wave window gains nothing, and the one-ring block window of csr_blockwin_stream_kernel needs the whole
// column range of 16 tiles inside 8192 slots.  But a few hundred consecutive ROWS share each cluster almost
// entirely: the union of the columns of 32 consecutive tiles is a handful of SEGMENTS of a few hundred to a
// thousand columns each, every slot of which is used 4-6 times.
//
// Plan time (csr_segwin_mark_kernel, one workgroup per block of `tiles_per_block` tiles): a bitmap over the
// column space (65536 bits, 2^shift columns per bit) collects the block's columns, runs of set bits become
// segments (the closest ones merged until at most kSegWinMaxSegs remain), a second pass over the entries gives
// every segment its exact first and last column, and -- if the segments together fit the window and every slot
// is used at least twice -- a third pass rewrites the block's 16-bit column stream to WINDOW SLOTS:
// slot = segment's place in the window + (column - segment's first column).  Whatever the columns' range (a KKT
// row spans millions), such a tile streams 2 bytes of index per entry.
//
// Multiply (csr_segwin_kernel): workgroups of 8 waves, one per block, two or three resident per CU -- while one
// waits at its barrier for the window the others multiply; no hand-rolled double buffering across blocks.  All
// 512 threads read the segments with coalesced loads (every load of a thread is issued before the first is waited
// for) into the window; each wave meanwhile has its first tile's streams in flight; after ONE barrier every wave
// walks through its tiles (the next tile's streams requested before the current one is multiplied), takes x
// from LDS by slot -- no address arithmetic, no vector-L1 look-up -- and adds the rows up exactly like
// csr_wavetile_kernel: same lanes per row, same order, same bits.
#pragma once

#include "csr_wavetile.hpp"
#include "csr_segtile.hpp"

namespace spmv {

constexpr int kSegWinWaves = 8;
// (8 until round 5; 12: a mesh of 3 unknowns per node numbered dof by dof has NINE clusters per block of rows -- three unknowns x
// three mesh planes -- and fell back to column panels: 321 -> 304 us, profiles/r05_segwin_dof.log; kkt-like and queen-like plans
// unchanged, profiles/r05_ab_segs12.log)
constexpr int kSegWinMaxSegs = 12;
constexpr int kSegWinMaxRuns = 64;
constexpr int kSegWinBitmapWords = 2048; // 65536 bits

struct SegWinBlock {
    int nseg;  // 0: no window, the block's tiles go through csr_wavetile_kernel
    int slots; // doubles of x the block stages
    int first_tile, ntiles;
    int start[kSegWinMaxSegs];  // first column of every segment
    int offset[kSegWinMaxSegs]; // its first window slot (segments lie back to back: offset[s + 1] = offset[s] + its length)
};

template <int QUADS>
struct SwTile {
    int r0, kb, last, nrows, maxlen, lanes_log2;
    int ps, pe, psB, peB;
    double yv, yvB;
    unsigned cx[QUADS], cy[QUADS];
    v2d va[QUADS], vb[QUADS];
    bool second;
};

template <int TILE>
__device__ __forceinline__ void sw_load_tile(
    SwTile<TILE / 256> & t, int w, const int4 * __restrict__ desc, const int32_t * __restrict__ p,
    const uint16_t * __restrict__ j16, const double * __restrict__ a, const double * y, int lane)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr int QUADS = TILE / 256;
    const TilePair dp = load_tile_pair(desc, w);
    const int4 d0 = dp.d0, d1 = dp.d1;
    const int meta = __builtin_amdgcn_readfirstlane(d0.z);
    t.r0 = __builtin_amdgcn_readfirstlane(d0.x & ~kTileFlagPartial);
    const int k0 = __builtin_amdgcn_readfirstlane(d0.y);
    const int r1 = __builtin_amdgcn_readfirstlane(d1.x & ~kTileFlagPartial);
    const int k1 = __builtin_amdgcn_readfirstlane(d1.y);
    t.maxlen = meta & 0xFFFF;
    t.lanes_log2 = (meta >> kTileMetaLanesShift) & 0x7;
    t.nrows = r1 - t.r0;
    t.kb = k0 & ~3;
    t.last = (k1 - 1 - t.kb) & ~3;
    const int sub = lane >> t.lanes_log2;
    const int rowi = sub < t.nrows ? sub : t.nrows - 1;
    const bool uniform = (meta & kTileMetaUniform) != 0;
    if (uniform) {
        t.ps = k0 + rowi * t.maxlen;
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
            t.psB = k0 + rowB * t.maxlen;
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

template <int TILE, bool PEER>
__device__ __forceinline__ void sw_compute_tile(
    const SwTile<TILE / 256> & t, double * prod, const double * win, unsigned wlimit, double * y, const PeerY & peers, int lane)
{
    constexpr int QUADS = TILE / 256;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= t.last) {
            // slots of the tile proper lie inside the window; entries of neighbouring tiles that share a quad
            // (their 16 bits may mean something else) are clamped into it and never summed
            const unsigned c0 = min(t.cx[q] & 0xFFFFu, wlimit), c1 = min(t.cx[q] >> 16, wlimit);
            const unsigned c2 = min(t.cy[q] & 0xFFFFu, wlimit), c3 = min(t.cy[q] >> 16, wlimit);
            const double q0 = t.va[q].x * win[c0];
            const double q1 = t.va[q].y * win[c1];
            const double q2 = t.vb[q].x * win[c2];
            const double q3 = t.vb[q].y * win[c3];
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
        y_store<PEER, false>(y, peers, t.r0 + sub, t.yv + z);
    if (t.second) {
        const double zB = tile_row_sum<1>(prod, t.psB - t.kb, t.peB - t.kb, 0, t.maxlen);
        if (lane + kWave < t.nrows)
            y_store<PEER, false>(y, peers, t.r0 + lane + kWave, t.yvB + zB);
    }
    // the product slice is reused by this wave's next tile: its reads above come first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// W: window slots (doubles) the kernel variant holds; the plan launches the smallest variant its blocks fit.
// PEER: every row sum is also stored into the other ranks' copies of y (csr_wavetile.hpp, PeerY).
template <int TILE, int W, bool PEER = false>
__global__ __launch_bounds__(kSegWinWaves * kWave, (W <= 2688 ? 6 : 4)) void csr_segwin_kernel(
    const int4 * __restrict__ desc, const SegWinBlock * __restrict__ blocks,
    const int32_t * __restrict__ p, const uint16_t * __restrict__ j16, const double * __restrict__ a,
    const double * __restrict__ x, const double * y_in, double * y, PeerY peers = PeerY{})
{
    constexpr int THREADS = kSegWinWaves * kWave;
    constexpr int XS = (W + THREADS - 1) / THREADS; // window slots a thread loads
    __shared__ double win[W];
    __shared__ __attribute__((aligned(16))) double prod_all[kSegWinWaves][TILE + 4];
    const SegWinBlock * bd = blocks + blockIdx.x;
    const int nseg = __builtin_amdgcn_readfirstlane(bd->nseg);
    if (nseg == 0)
        return; // no window: the block's tiles went through csr_wavetile_kernel
    const int slots = __builtin_amdgcn_readfirstlane(bd->slots);
    const int t_begin = __builtin_amdgcn_readfirstlane(bd->first_tile);
    const int t_end = t_begin + __builtin_amdgcn_readfirstlane(bd->ntiles);
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    const int tid = (int) threadIdx.x;
    double * prod = prod_all[wave];

    // (1) this wave's first tile: its loads do not depend on the window
    SwTile<TILE / 256> cur, nxt;
    int t = t_begin + wave;
    if (t < t_end)
        sw_load_tile<TILE>(cur, t, desc, p, j16, a, y_in, lane);
    // (2) the window: slot i belongs to the last segment whose offset is <= i and holds x[i + (start - offset)]
    {
        int col[XS];
#pragma unroll
        for (int k = 0; k < XS; ++k)
            col[k] = tid + THREADS * k + __builtin_amdgcn_readfirstlane(bd->start[0]);
        for (int s = 1; s < nseg; ++s) { // wave-uniform trip count
            const int off = __builtin_amdgcn_readfirstlane(bd->offset[s]);
            const int delta = __builtin_amdgcn_readfirstlane(bd->start[s]) - off;
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const int i = tid + THREADS * k;
                col[k] = i >= off ? i + delta : col[k];
            }
        }
        double xs[XS];
#pragma unroll
        for (int k = 0; k < XS; ++k)
            if (THREADS * k < slots) // wave-uniform
                xs[k] = (tid + THREADS * k < slots) ? x[col[k]] : 0.0;
#pragma unroll
        for (int k = 0; k < XS; ++k)
            if (THREADS * k < slots && tid + THREADS * k < slots)
                win[tid + THREADS * k] = xs[k];
    }
    __syncthreads(); // the only one
    const unsigned wlimit = (unsigned) (slots - 1);
    // (3) this wave's tiles, the next one's streams requested before the current one is multiplied
    while (t < t_end) {
        const int tn = t + kSegWinWaves;
        if (tn < t_end)
            sw_load_tile<TILE>(nxt, tn, desc, p, j16, a, y_in, lane);
        sw_compute_tile<TILE, PEER>(cur, prod, win, wlimit, y, peers, lane);
        cur = nxt;
        t = tn;
    }
}

// Plan-time: one workgroup per block of tiles_per_block consecutive tiles.  apply == 0: counts[3] += tiles of the
// blocks that would get a window.  apply != 0: the block records are written, the tiles marked
// (kTileMetaBlockWin: csr_wavetile_kernel skips them) and their 16-bit column stream rewritten to window slots.
static __global__ __launch_bounds__(kSegWinWaves * kWave) void csr_segwin_mark_kernel(
    int ntiles, int tile, int tiles_per_block, int4 * __restrict__ desc, const int32_t * __restrict__ j,
    uint16_t * __restrict__ j16, SegWinBlock * __restrict__ blocks, int * __restrict__ counts, int apply,
    int shift, int max_slots, int take_narrow_blocks)
{
    constexpr int THREADS = kSegWinWaves * kWave;
    __shared__ unsigned bitmap[kSegWinBitmapWords];
    __shared__ int s_ok, s_entries, s_nseg, s_slots, s_narrow;
    __shared__ int run_lo[kSegWinMaxRuns], run_hi[kSegWinMaxRuns]; // in bits, inclusive
    __shared__ int seg_lo[kSegWinMaxSegs], seg_hi[kSegWinMaxSegs], seg_off[kSegWinMaxSegs];
    const int tid = (int) threadIdx.x;
    const int wave = tid >> 6;
    const int lane = (int) __lane_id();
    const int t0 = (int) blockIdx.x * tiles_per_block;
    const int t1 = min(ntiles, t0 + tiles_per_block);
    for (int i = tid; i < kSegWinBitmapWords; i += THREADS)
        bitmap[i] = 0u;
    if (tid == 0) {
        s_ok = 1;
        s_entries = 0;
        s_nseg = 0;
        s_slots = 0;
        s_narrow = 0;
    }
    __syncthreads();
    // (1) the block's columns into the bitmap; every tile must be a plain stream tile (shifted tiles and per-tile
    // windows are cheaper; long rows and ragged tiles have their own paths)
    for (int t = t0 + wave; t < t1; t += kSegWinWaves) {
        const int4 d0 = desc[t];
        const int k0 = d0.y, k1 = desc[t + 1].y;
        const int m = d0.z;
        const bool ok = !(d0.x & kTileFlagPartial) && (m & kTileMetaFast)
            && !(m & (kTileMetaShifted | kTileMetaXWin | kTileMetaXSeg | kTileMetaPattern | kTileMetaSeg | kTileMetaBlockWin))
            && k1 - (k0 & ~3) <= tile;
        if (!ok) {
            if (lane == 0)
                s_ok = 0;
            continue;
        }
        for (int k = k0 + lane; k < k1; k += kWave) {
            const unsigned b = (unsigned) j[k] >> shift;
            atomicOr(&bitmap[(b >> 5) & (kSegWinBitmapWords - 1)], 1u << (b & 31u));
        }
        if (lane == 0) {
            atomicAdd(&s_entries, k1 - k0);
            if (m & kTileMetaNarrow)
                atomicAdd(&s_narrow, 1);
        }
    }
    __syncthreads();
    // A block whose tiles all have 16-bit columns already keeps them: their gather goes through the vector L1 at
    // no measurable cost where consecutive rows share their columns (queen-like mesh, 2442 slots per 32 tiles,
    // same box and process: 657 us without windows, 659-677 us with), and unstructured bands have the one-ring block
    // window.  What the segment windows are for is the tiles that could NOT be compressed -- columns in clusters
    // more than 65536 apart -- which otherwise stream 4-byte indices and gather from all over x (KKT-like with
    // jittered stencils: 1510 -> 926 us).
    // (2) runs of set bits -> at most kSegWinMaxSegs segments (the closest runs merged first)
    // (take_narrow_blocks: the A/B switch of the experiments build that produced those numbers)
    if (tid == 0 && s_ok && (take_narrow_blocks || s_narrow < t1 - t0)) {
        int n = 0, open = -1, prev = -2;
        bool fits = true;
        for (int w = 0; w < kSegWinBitmapWords && fits; ++w) {
            unsigned bits = bitmap[w];
            while (bits) {
                const int b = w * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                if (b != prev + 1) { // a new run begins
                    if (open >= 0) {
                        if (n == kSegWinMaxRuns) { fits = false; break; }
                        run_lo[n] = open;
                        run_hi[n++] = prev;
                    }
                    open = b;
                }
                prev = b;
            }
        }
        if (fits && open >= 0) {
            if (n == kSegWinMaxRuns)
                fits = false;
            else {
                run_lo[n] = open;
                run_hi[n++] = prev;
            }
        }
        while (fits && n > kSegWinMaxSegs) {
            int best = 1, gap = run_lo[1] - run_hi[0];
            for (int i = 2; i < n; ++i)
                if (run_lo[i] - run_hi[i - 1] < gap) {
                    gap = run_lo[i] - run_hi[i - 1];
                    best = i;
                }
            run_hi[best - 1] = run_hi[best];
            for (int i = best; i + 1 < n; ++i) {
                run_lo[i] = run_lo[i + 1];
                run_hi[i] = run_hi[i + 1];
            }
            --n;
        }
        s_nseg = (fits && n > 0) ? n : 0;
        for (int i = 0; i < kSegWinMaxSegs; ++i) {
            seg_lo[i] = 0x7FFFFFFF;
            seg_hi[i] = -1;
        }
    }
    __syncthreads();
    const int nseg = s_nseg;
    if (nseg == 0) { // uniform
        if (apply && tid == 0)
            blocks[blockIdx.x].nseg = 0;
        return;
    }
    // (3) exact first / last column of every segment
    for (int t = t0 + wave; t < t1; t += kSegWinWaves) {
        const int k0 = desc[t].y, k1 = desc[t + 1].y;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const int c = j[k];
            const int b = (int) ((unsigned) c >> shift);
            int s = 0;
            for (int i = 1; i < nseg; ++i)
                s = b >= run_lo[i] ? i : s;
            atomicMin(&seg_lo[s], c);
            atomicMax(&seg_hi[s], c);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0;
        for (int i = 0; i < nseg; ++i) {
            seg_off[i] = off;
            off += seg_hi[i] - seg_lo[i] + 1;
            if (off > max_slots)
                break;
        }
        // the window pays when its slots are used at least twice (as for the per-tile windows)
        s_slots = (off <= max_slots && s_entries >= 2 * off) ? off : 0;
    }
    __syncthreads();
    const int slots = s_slots;
    if (slots == 0) {
        if (apply && tid == 0)
            blocks[blockIdx.x].nseg = 0;
        return;
    }
    if (tid == 0)
        striped_add(counts, 3, t1 - t0);
    if (!apply)
        return;
    if (tid == 0) {
        SegWinBlock bd;
        bd.nseg = nseg;
        bd.slots = slots;
        bd.first_tile = t0;
        bd.ntiles = t1 - t0;
        for (int i = 0; i < kSegWinMaxSegs; ++i) {
            bd.start[i] = i < nseg ? seg_lo[i] : 0;
            bd.offset[i] = i < nseg ? seg_off[i] : slots;
        }
        blocks[blockIdx.x] = bd;
    }
    // (4) mark the tiles and rewrite their column stream to window slots
    for (int t = t0 + wave; t < t1; t += kSegWinWaves) {
        const int k0 = desc[t].y, k1 = desc[t + 1].y;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const int c = j[k];
            const int b = (int) ((unsigned) c >> shift);
            int s = 0;
            for (int i = 1; i < nseg; ++i)
                s = b >= run_lo[i] ? i : s;
            j16[k] = (uint16_t) (seg_off[s] + (c - seg_lo[s]));
        }
        if (lane == 0)
            desc[t].z |= kTileMetaBlockWin;
    }
}

} // namespace spmv
