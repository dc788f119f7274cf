// csr_wavetile.hpp -- the default CSR kernel (per-wavefront row ownership, src/matrix/csr-matrix-spmv.cpp:21-33, 63-76)
// and the tile classes it multiplies: shifted tiles, lane-per-row stencil tiles, x windows, column panels, value dictionary.
#pragma once

#include "tile_common.hpp"
#include "csr_blocktile.hpp"

namespace spmv {

// A "shifted" tile: entry t of the tile (row t / len, position t % len) has column
// first_row[t % len] + t / len, so the column stream shrinks to the first row's `len` columns,
// read from the original 32-bit array (the tile's columns may span any range: a 253^3 grid's
// 27-point rows reach 128 K columns) and parked in the wave's LDS table (len <= 128).  Holding
// them one per lane and fetching with ds_bpermute measured the same
// (profiles/r01_sweep_shifted_*.log) and stops at 64.  t / len uses a 22-bit reciprocal, exact
// while t * len < 2^22 (t < 1024, len <= 512), with t * magic < 2^32.
constexpr int kShiftedMaxLen = 128;

// Timing experiments of the value-dictionary path (bits 1-8: parts of the work compiled out, wrong results by design;
// 16: y read with a plain instead of a non-temporal load, 32: y written with a plain store -- right results).  They exist
// only in libraries built by tools/ablate.sh with -DSPMV_HIP_EXPERIMENTS -DSPMV_VI_ABLATE=n; everywhere else the constant
// is 0 and every test on it folds away.
#if defined(SPMV_HIP_EXPERIMENTS) && defined(SPMV_VI_ABLATE)
constexpr int kViAblate = SPMV_VI_ABLATE;
#else
constexpr int kViAblate = 0;
#endif
// A dword at a WAVE-UNIFORM address through the scalar data cache (s_load_dword): the constant address space is how
// the compiler is told that the load may go there; the data is read-only for the whole launch.
__device__ __forceinline__ int scalar_load_i32(const int32_t * p)
{
    typedef const int32_t __attribute__((address_space(4))) * const_ptr;
    return *reinterpret_cast<const_ptr>(reinterpret_cast<uintptr_t>(p));
}
} // namespace spmv
#include "csr_stenciltile.hpp" // (uses scalar_load_i32)
namespace spmv {

template <int QUADS, bool X32, bool VI = false>
__device__ __forceinline__ void tile_products_shifted(
    double * prod, uint32_t * tab, const int32_t * __restrict__ first_row, int first_row_base,
    const double * __restrict__ at, const double * __restrict__ x, unsigned limit, int last, int lane,
    int len, int lead, const uint8_t * __restrict__ vit = nullptr, ValueLookup vtab = ValueLookup{nullptr, false, 0.0, 0.0})
{
    static_assert(QUADS * 256 <= 1024, "reciprocal below is exact for t < 1024 only");
    TileValues<QUADS, VI> vals;
    // first_row: the tile's own first row in the column array (base 0), or its pattern's columns
    // relative to the first row index (base = that index; cache-resident, no per-tile read)
    for (int i = lane; i < len; i += kWave)
        tab[i] = (uint32_t) (first_row[i] + first_row_base);
    vals.load(at, vit, last, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned magic = ((1u << 22) + (unsigned) len - 1u) / (unsigned) len; // wave-uniform
    // the x gathers depend on the descriptor and the first row only: all of them are issued before anything
    // waits for the value stream (with a value dictionary the table look-ups below need the index loads back)
    double xg[QUADS][4];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // entries in front of the tile (they share its first quad) are multiplied and never
                // read back, like the ones behind its end; both only need a valid column
                const int ti = o + i - lead;
                const unsigned t = ti > 0 ? (unsigned) ti : 0u;
                const unsigned r = (t * magic) >> 22;
                if (VI && (kViAblate & 1))
                    xg[q][i] = gather_x<X32>(x, (int) ((tab[t - r * (unsigned) len] + r) & 15u)); // no x traffic
                else
                    xg[q][i] = gather_x<X32>(x, (int) min(tab[t - r * (unsigned) len] + r, limit));
            }
        }
    }
    vals.resolve(vtab);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{vals.va[q].x * xg[q][0], vals.va[q].y * xg[q][1]};
            dst[1] = v2d{vals.vb[q].x * xg[q][2], vals.vb[q].y * xg[q][3]};
        }
    }
}

// A shifted tile whose rows are all equally long (the interior of a stencil), under a value dictionary: ONE LANE
// PER ROW.  Row r of the tile has the columns first_row[pos] + r, so for a given pos the lanes of a wave read
// x[first_row[pos] + lane]: 512 contiguous bytes, 8 accesses of the vector L1 -- where the entry-major layout of
// tile_products_shifted (lane = four consecutive entries) lands the 64 lanes of every gather on all the
// diagonals at once, ~35 different 64-byte pieces per instruction.  The counters of the value-dictionary launch
// (71 M L1 accesses in 143 us: 0.85 per clock and CU, profiles/r02_prof_poisson_csr_vi_summary.md) say that this
// look-up rate, not memory, was what it ran at.  first_row is read position by position with scalar loads
// (wave-uniform address); the tile's index bytes go through the wave's LDS slice (two coalesced
// dwords per lane in, the row's bytes out); the doubles come from the table and are added left to right from
// +0.0: the reference's order, bit for bit.  Lanes own a second row 64 further on when the tile has more than 64.
// A lane per row pays while the tile has rows for at least half the wave: rows of up to 16 entries (32+ rows per 512-entry
// tile).  Longer rows reach this test only under SPMV_HIP_FLAG_EXACT_ORDER (ELLPACK): 33 entries per row would leave
// 15 lanes gathering in seven dependent rounds -- measured on an ELLPACK band of 33: 199 us against 157 for the
// entry-major path.
constexpr int kLanePerRowMaxLen = 16;

template <bool X32>
__device__ __forceinline__ void tile_rows_uniform_indexed(
    double * prod, const int32_t * __restrict__ first_row, int first_row_base,
    const uint8_t * __restrict__ vit, ValueLookup vtab, const double * __restrict__ x, int last, int lane,
    int len, int lead, int nrows, bool second, double & zA, double & zB)
{
    // The first row's columns come through the SCALAR data cache (their address is wave-uniform: a pattern record shared by
    // every interior tile, or the tile's own first row): a scalar-cache hit returns sooner than the vector load from the
    // L2 that round 2 used (one column per lane, broadcast with v_readlane), and the x gathers below wait for nothing
    // else -- Poisson 4096^2 with its dictionary 119.3-120.1 -> 113.5-116.2 us in the same process (round 3).
    unsigned * vw = reinterpret_cast<unsigned *>(prod);
    unsigned vi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last; // lanes past the tile's end re-read its last dword (and park it where nobody looks)
        vi[q] = *reinterpret_cast<const unsigned *>(vit + o);
    }
    const int rowA = lane < nrows ? lane : nrows - 1;
    const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
    zA = 0.0;
    zB = 0.0;
    constexpr int CH = 5; // positions per round: a 5-point row in one go
    double xa[CH], xb[CH];
    // first round of gathers: they depend on first_row only and leave before the index bytes are back
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (i < len) { // wave-uniform
            const int c = scalar_load_i32(first_row + i) + first_row_base;
            xa[i] = gather_x<X32>(x, (kViAblate & 1) ? ((c + rowA) & 15) : c + rowA);
            if (second)
                xb[i] = gather_x<X32>(x, (kViAblate & 1) ? ((c + rowB) & 15) : c + rowB);
        }
    }
    vw[lane] = vi[0];
    vw[64 + lane] = vi[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t * vA = reinterpret_cast<const uint8_t *>(prod) + lead + rowA * len;
    const uint8_t * vB = reinterpret_cast<const uint8_t *>(prod) + lead + rowB * len;
    for (int p0 = 0; p0 < len; p0 += CH) {
        if (p0 > 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (p0 + i < len) {
                    const int c = scalar_load_i32(first_row + p0 + i) + first_row_base;
                    xa[i] = gather_x<X32>(x, c + rowA);
                    if (second)
                        xb[i] = gather_x<X32>(x, c + rowB);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                zA += vtab[vA[p0 + i] & 0x7Fu] * xa[i];
                if (second)
                    zB += vtab[vB[p0 + i] & 0x7Fu] * xb[i];
            }
        }
    }
}

// ... and when every row of the tile carries the FIRST row's dictionary indices (kTileMetaValueRows: a constant-coefficient
// stencil), nothing of the index stream is read beyond those `len` bytes, which come through the scalar cache like the first
// row's columns: no vector load of values, no LDS, the coefficients sit in scalar registers.  Same products, same order,
// same bits.  Poisson 4096^2: 113.8-114.3 -> 106.9-107.6 us, 411 instead of 495 MB streamed (round 3).
template <bool X32>
__device__ __forceinline__ void tile_rows_uniform_constant(
    const int32_t * __restrict__ first_row, int first_row_base, const uint8_t * __restrict__ vit, ValueLookup vtab,
    const double * __restrict__ x, int lane, int len, int lead, int nrows, bool second, double & zA, double & zB)
{
    const int rowA = lane < nrows ? lane : nrows - 1;
    const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
    zA = 0.0;
    zB = 0.0;
    constexpr int CH = 5;
    double xa[CH], xb[CH];
    for (int p0 = 0; p0 < len; p0 += CH) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) { // wave-uniform
                const int c = scalar_load_i32(first_row + p0 + i) + first_row_base;
                xa[i] = gather_x<X32>(x, c + rowA);
                if (second)
                    xb[i] = gather_x<X32>(x, c + rowB);
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                const int bi = lead + p0 + i; // vit is 4-byte aligned: the dword that holds byte bi, through the scalar cache
                const int dw = scalar_load_i32(reinterpret_cast<const int32_t *>(vit) + (bi >> 2));
                const double val = vtab[(unsigned) (dw >> ((bi & 3) * 8)) & 0x7Fu];
                zA += val * xa[i];
                if (second)
                    zB += val * xb[i];
            }
        }
    }
}

// The same lane-per-row scheme with the values themselves (no dictionary): the tile's values are loaded as ever --
// two coalesced 16-byte loads per lane and quad -- and parked in the wave's LDS slice where the products used to
// go; a lane then reads its row's values back (the access pattern the row sums had) and multiplies them with x
// read 512 contiguous bytes at a time.  Same bits as the reference's loop.
template <int QUADS, bool X32>
__device__ __forceinline__ void tile_rows_uniform_values(
    double * prod, const int32_t * __restrict__ first_row, int first_row_base, const double * __restrict__ at,
    const double * __restrict__ x, int last, int lane, int len, int lead, int nrows, bool second, double & zA, double & zB)
{
    TileValues<QUADS, false> vals;
    vals.load(at, nullptr, last, lane);
    const int rowA = lane < nrows ? lane : nrows - 1;
    const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
    zA = 0.0;
    zB = 0.0;
    constexpr int CH = 5;
    double xa[CH], xb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (i < len) { // wave-uniform
            const int c = scalar_load_i32(first_row + i) + first_row_base; // scalar data cache, see tile_rows_uniform_indexed
            xa[i] = gather_x<X32>(x, c + rowA);
            if (second)
                xb[i] = gather_x<X32>(x, c + rowB);
        }
    }
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = vals.va[q];
            dst[1] = vals.vb[q];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double * vA = prod + lead + rowA * len;
    const double * vB = prod + lead + rowB * len;
    for (int p0 = 0; p0 < len; p0 += CH) {
        if (p0 > 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (p0 + i < len) {
                    const int c = scalar_load_i32(first_row + p0 + i) + first_row_base;
                    xa[i] = gather_x<X32>(x, c + rowA);
                    if (second)
                        xb[i] = gather_x<X32>(x, c + rowB);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                zA += vA[p0 + i] * xa[i];
                if (second)
                    zB += vB[p0 + i] * xb[i];
            }
        }
    }
}

// The window slots of a quad's four entries from the packed position table (tile_common.hpp: one 64-bit word per row position):
// tf = tile-relative index of the quad's first entry (negative in the tile's first quad when the tile does not start on a
// quad boundary: those entries are the previous tile's, their products are never read).
// ... kept in LDS with the four residues of the position apart: the lanes of a quad loop read positions 4 apart (8-byte words 32
// bytes apart would meet in the same banks eight at a time), this way consecutive words
__device__ __forceinline__ unsigned packed_slot_index(unsigned pos) { return ((pos & 3u) << 4) | (pos >> 2); } // pos < 64

__device__ __forceinline__ void packed_window_slots(unsigned (&cc)[4], const unsigned long long * tab, int tf, int len, unsigned magic, unsigned wlimit)
{
    const unsigned t = tf > 0 ? (unsigned) tf : 0u;
    const unsigned r = (t * magic) >> 22;
    unsigned long long w = tab[packed_slot_index(t - r * (unsigned) len)];
    w <<= tf < 0 ? (unsigned) (-16 * tf) : 0u; // entry i of the quad is entry tf + i of the tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
        cc[i] = min(((unsigned) (w >> (16 * i)) & 0xFFFFu) + r, wlimit);
}

// x staged through LDS (kernel variant XW > 0, tiles marked kTileMetaXWin): the tile's column
// range [base, base + 64 * chunks) is read once with coalesced loads into the wave's window and the
// products take x from there (ds_read_b64) instead of gathering it through the vector L1.  All
// global loads -- window, column offsets or first row, values -- are issued before the first wait.
template <int QUADS, int XW>
__device__ __forceinline__ void tile_products_xwin(
    double * prod, double * xw, uint32_t * tab, const uint16_t * __restrict__ jt,
    const int32_t * __restrict__ first_row, int first_row_base, const double * __restrict__ at,
    const double * __restrict__ xt, int cbase, unsigned limit, int last, int lane, int chunks, bool shifted,
    int len, int lead, const int32_t * __restrict__ win4 = nullptr /* shifted tile with a pattern, len <= 64: its packed positions */)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    static_assert(XW % 64 == 0 && XW <= 256, "window is staged in at most four 64-entry chunks");
    double xs[XW / 64];
#pragma unroll
    for (int ch = 0; ch < XW / 64; ++ch)
        if (ch < chunks)
            xs[ch] = xt[min((unsigned) (64 * ch + lane), limit)];
    v2u c[QUADS];
    v2d va[QUADS], vb[QUADS];
    const bool packed = shifted && win4 != nullptr;
    if (packed) {
        reinterpret_cast<unsigned long long *>(tab)[packed_slot_index((unsigned) lane)] = reinterpret_cast<const unsigned long long *>(win4)[lane < len ? lane : len - 1];
    } else if (shifted) {
        for (int i = lane; i < len; i += kWave)
            tab[i] = (uint32_t) (first_row[i] + first_row_base - cbase);
    }
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last;
        if (!shifted)
            c[q] = *reinterpret_cast<const v2u *>(jt + o);
        va[q] = *reinterpret_cast<const v2d *>(at + o);
        vb[q] = *reinterpret_cast<const v2d *>(at + o + 2);
    }
#pragma unroll
    for (int ch = 0; ch < XW / 64; ++ch)
        if (ch < chunks)
            xw[64 * ch + lane] = xs[ch];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned magic = ((1u << 22) + (unsigned) len - 1u) / (unsigned) len;
    const unsigned wlimit = (unsigned) (64 * chunks - 1); // garbage entries of shared quads stay inside the window
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            unsigned cc[4];
            if (packed) {
                packed_window_slots(cc, reinterpret_cast<const unsigned long long *>(tab), o - lead, len, magic, wlimit);
            } else if (shifted) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ti = o + i - lead;
                    const unsigned t = ti > 0 ? (unsigned) ti : 0u;
                    const unsigned r = (t * magic) >> 22;
                    cc[i] = min(tab[t - r * (unsigned) len] + r, wlimit);
                }
            } else {
                cc[0] = min(c[q].x & 0xFFFFu, wlimit);
                cc[1] = min(c[q].x >> 16, wlimit);
                cc[2] = min(c[q].y & 0xFFFFu, wlimit);
                cc[3] = min(c[q].y >> 16, wlimit);
            }
            const double q0 = va[q].x * xw[cc[0]];
            const double q1 = va[q].y * xw[cc[1]];
            const double q2 = vb[q].x * xw[cc[2]];
            const double q3 = vb[q].y * xw[cc[3]];
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
}

// x window of a shifted tile whose columns are too far apart for one contiguous window (any
// stencil in 2 or 3 dimensions): entry (row r, position pos) reads x[first_row[pos] + r], i.e. the
// tile needs `len` runs of `rows` consecutive x entries.  Runs that touch or overlap are merged,
// and the layout -- where each position's run starts in the window (xoff), which x entry each
// window slot holds relative to the tile's first row (src) -- comes from the tile's pattern
// record, which is shared by all tiles of the same shape and therefore cache-resident: the
// window loads can be issued as soon as the descriptor is there (27-point stencil: 180 slots in
// 9 runs instead of 486 gathered entries touching ~50 lines per instruction); the products then
// read x from LDS.  Per-tile tables instead of patterns measured 201 vs 176 us (768 B per tile
// and one more dependent round trip).
template <int QUADS, int XW>
__device__ __forceinline__ void tile_products_xseg(
    double * prod, double * xw, unsigned long long * tab, const int32_t * __restrict__ pat, int r0,
    const double * __restrict__ at, const double * __restrict__ x,
    int limit, int last, int lane, int chunks, int len, int lead)
{
    static_assert(XW % 64 == 0 && XW <= 256, "window is staged in at most four 64-entry chunks");
    int so[XW / 64];
#pragma unroll
    for (int ch = 0; ch < XW / 64; ++ch)
        if (ch < chunks)
            so[ch] = pat[kPatSrc + 64 * ch + lane];
    // four window positions per row position (tile_common.hpp): len <= 64
    const unsigned long long xo4 = reinterpret_cast<const unsigned long long *>(pat + kPatXoff4)[lane < len ? lane : len - 1];
    v2d va[QUADS], vb[QUADS];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last;
        va[q] = *reinterpret_cast<const v2d *>(at + o);
        vb[q] = *reinterpret_cast<const v2d *>(at + o + 2);
    }
    double xs[XW / 64];
#pragma unroll
    for (int ch = 0; ch < XW / 64; ++ch)
        if (ch < chunks) {
            int c = r0 + so[ch];
            c = c < 0 ? 0 : (c > limit ? limit : c); // padding slots of the last chunk
            xs[ch] = x[c];
        }
    tab[packed_slot_index((unsigned) lane)] = xo4;
#pragma unroll
    for (int ch = 0; ch < XW / 64; ++ch)
        if (ch < chunks)
            xw[64 * ch + lane] = xs[ch];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned magic = ((1u << 22) + (unsigned) len - 1u) / (unsigned) len;
    const unsigned wlimit = (unsigned) (64 * chunks - 1);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            unsigned cc[4];
            packed_window_slots(cc, tab, o - lead, len, magic, wlimit);
            const double q0 = va[q].x * xw[cc[0]];
            const double q1 = va[q].y * xw[cc[1]];
            const double q2 = vb[q].x * xw[cc[2]];
            const double q3 = vb[q].y * xw[cc[3]];
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
}

// ABL: timing experiments that switch parts of the work off (results are wrong by design):
// 1 = x gather collapsed to two entries, 2 = row sums reduced to one LDS read per row.
// Column panels (kernel variant PANELS): the matrix handed to the kernel is the plan's own copy, cut
// into 8 column panels and stored panel by panel, "row" v = panel * rows + r holding row r's
// entries of that panel.  Workgroups b, b + 8, b + 16, ... share an XCD (observed dispatch order,
// used for speed only), so workgroup b works on panel b % 8: every XCD then gathers from one eighth
// of x, which stays in its private 4 MB L2, instead of dragging all of x through it (2 M rows x 24
// random columns: 694 us with x = 16 MB, 270 us with x = 2 MB).  A row's eight partial sums meet in
// y through fp64 atomics.
struct PanelInfo {
    int first[9]; // tiles [first[k], first[k+1]) belong to panel k
    int rows;     // rows of the matrix (= virtual rows per panel)
};

// Kernel variant PEER (one process per GPU, rows partitioned, src/matrix/csr-matrix.cpp:77-95): every row sum is stored
// not only into this rank's y but straight into the same place of up to kMaxPeers OTHER ranks' copies of y (their
// memory, mapped here through hipIpcOpenMemHandle; the stores leave over the xGMI link to each peer).  The all-gather
// of the y segments then costs no launch and no collective of its own: it travels while the tiles are multiplied.
constexpr int kMaxPeers = 7;
struct PeerY {
    double * y[kMaxPeers]; // where this rank's FIRST row lives in each peer's y
    int n;
};

// NT: the store goes past the caches (non-temporal).  y is read once and written once per launch; written normally it takes room
// in the L2 that x and the streams want: the dictionary variant and the value-reading stencil tiles always store it that way
// (143 -> 139 us, 192.2 -> 189.9 us), every other tile class when the launch says so (`nt`: the matrix streams from HBM --
// 27 diagonals 171.6 -> 165.7 us, kkt-like and queen-like unchanged: profiles/r05_ab_y_nt_all.log; a cache-resident matrix keeps
// its y in the caches between launches).
template <bool PEER, bool NT>
__device__ __forceinline__ void y_store(double * y, const PeerY & peers, long long idx, double v, bool nt = false)
{
    if (NT || nt)
        __builtin_nontemporal_store(v, y + idx);
    else
        y[idx] = v;
    if (PEER) {
#pragma unroll
        for (int k = 0; k < kMaxPeers; ++k) // static indices: the pointers stay in scalar registers
            if (k < peers.n)
                peers.y[k][idx] = v;
    }
}

// Constant-row tiles with a lane owning two ADJACENT rows (2l, 2l + 1): x, old y and new y move as ONE 16-byte access per lane
// instead of two 8-byte ones -- half the vector-memory instructions of the tile (the wave traces show a stencil wave spending
// the first and the last microsecond of its life issuing those into a full memory pipeline); every row is still added left to
// right by one lane: same bits.  The accesses are 8-byte aligned only (a stencil row has odd and even columns): global memory
// takes that.  A tile with an odd number of rows: the lane that would start at the last row starts one row earlier and
// stores only its second sum.  Poisson 4096^2: 106.8-107.3 -> 98.9-100.7 us on three boxes (round 3; the same idea on the
// indexed path, before the constant rows, had been -3 % ... +7 % depending on the box).
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
template <bool X32, bool PEER>
__device__ __forceinline__ void tile_rows_pairs_constant(
    const int32_t * __restrict__ first_row, int first_row_base, const uint8_t * __restrict__ vit, ValueLookup vtab,
    const double * __restrict__ x, const double * y_in, double * y, const PeerY & peers, int r0, int lane, int len, int lead, int nrows)
{
    const int top = nrows - 2; // nrows >= 2
    const int base = 2 * lane < top ? 2 * lane : top;
    const bool both = 2 * lane <= top, only_second = 2 * lane == nrows - 1;
    const v2d_a8 yv = __builtin_nontemporal_load(reinterpret_cast<const v2d_a8 *>(y_in + r0 + base));
    constexpr int CH = 5;
    v2d_a8 xv[CH];
    double zA = 0.0, zB = 0.0;
    for (int p0 = 0; p0 < len; p0 += CH) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                const int c = scalar_load_i32(first_row + p0 + i) + first_row_base;
                xv[i] = X32 ? *reinterpret_cast<const v2d_a8 *>(reinterpret_cast<const char *>(x) + ((unsigned) (c + base) << 3))
                            : *reinterpret_cast<const v2d_a8 *>(x + c + base);
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                const int bi = lead + p0 + i;
                const int dw = scalar_load_i32(reinterpret_cast<const int32_t *>(vit) + (bi >> 2));
                const double val = vtab[(unsigned) (dw >> ((bi & 3) * 8)) & 0x7Fu];
                zA += val * xv[i].x;
                zB += val * xv[i].y;
            }
        }
    }
    if (both) {
        v2d_a8 out = {yv.x + zA, yv.y + zB};
        __builtin_nontemporal_store(out, reinterpret_cast<v2d_a8 *>(y + r0 + base));
        if (PEER) {
#pragma unroll
            for (int k = 0; k < kMaxPeers; ++k)
                if (k < peers.n) {
                    peers.y[k][r0 + base] = out.x;
                    peers.y[k][r0 + base + 1] = out.y;
                }
        }
    } else if (only_second) {
        y_store<PEER, true>(y, peers, r0 + base + 1, yv.y + zB);
    }
}

// Multi-window tiles: SEVERAL rows of 129 ... 512 entries owned by one wave that holds more than TILE entries (the host forms
// them where such rows would leave a 512-entry tile badly filled, plan_csr.hip).  The wave walks the tile's entries in windows
// of TILE: loads, gathers and parks one window's products exactly like a plain tile, then every row's lanes add what the window
// holds of THEIR row to a partial sum they keep in registers; one butterfly per row at the very end.  Every window is full
// but the last: 7 rows of 361 entries = 4.94 windows instead of 7 tiles 70 % full.
template <int LPR>
__device__ __forceinline__ double window_row_sum(const double * prod, int s, int e, int part)
{
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
    int k = s + part;
    for (; k + 3 * LPR < e; k += 4 * LPR) {
        const double a = prod[k], b = prod[k + LPR], c = prod[k + 2 * LPR], d = prod[k + 3 * LPR];
        z0 += a;
        z1 += b;
        z2 += c;
        z3 += d;
    }
    for (; k < e; k += LPR)
        z0 += prod[k];
    return (z0 + z1) + (z2 + z3);
}

template <int TILE, int QUADS, bool C16, bool X32, bool VI, typename YStore>
__device__ __forceinline__ void tile_rows_multi_window(
    double * prod, const int32_t * __restrict__ p, const int32_t * __restrict__ j, const uint16_t * __restrict__ j16,
    const double * __restrict__ a, const double * __restrict__ x, const double * y_in, int r0, int k0, int k1, int nrows, int meta,
    int cbase, int cols, int lane, const uint8_t * __restrict__ vidx, ValueLookup vtab, YStore && store)
{
    const int kb = k0 & ~3;
    const int lanes_log2 = (meta >> kTileMetaLanesShift) & 0x7; // 3, 4 or 5: rows * lanes <= 64
    const int sub = lane >> lanes_log2, part = lane & ((1 << lanes_log2) - 1);
    const int rowi = sub < nrows ? sub : nrows - 1;
    int ps, pe;
    if (meta & kTileMetaUniform) {
        ps = k0 + rowi * (meta & 0xFFFF);
        pe = ps + (meta & 0xFFFF);
    } else {
        ps = p[r0 + rowi];
        pe = p[r0 + rowi + 1];
    }
    const double yv = y_in[r0 + rowi];
    const bool narrow = C16 && (meta & kTileMetaNarrow);
    double z = 0.0;
    for (int base = kb; base < k1; base += TILE) { // wave-uniform
        const int left = k1 - base;
        const int last = ((left < TILE ? left : TILE) - 1) & ~3;
        if (narrow)
            tile_products_narrow<QUADS, 0, VI>(prod, j16 + base, a + base, x + cbase, (unsigned) (cols - 1 - cbase), last, lane, vidx + base, vtab);
        else
            tile_products_wide<QUADS, X32, VI>(prod, j + base, a + base, x, last, lane, vidx + base, vtab);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int s = (ps > base ? ps : base) - base;
        const int e = (pe < base + TILE ? pe : base + TILE) - base;
        // (four reads in flight per lane: a window holds one or two of the tile's rows, so only their lanes work here, each
        // through 30 ... 60 products -- one dependent LDS round trip per product was most of a window's time)
        switch (lanes_log2) {
        case 3: z += window_row_sum<8>(prod, s, e, part); break;
        case 4: z += window_row_sum<16>(prod, s, e, part); break;
        default: z += window_row_sum<32>(prod, s, e, part); break;
        }
        // (the next window overwrites the slice: same-wave LDS operations execute in order; the fences pin the compiler)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    switch (lanes_log2) {
    case 3: z = group_sum<8>(z); break;
    case 4: z = group_sum<16>(z); break;
    case 5: z = group_sum<32>(z); break;
    default: z = group_sum<64>(z); break;
    }
    if (sub < nrows && part == 0)
        store(r0 + sub, yv + z);
}

// VI: the plan holds a value dictionary (see TileValues): vidx = one byte per stored entry, vtable = the
// <= kMaxIndexedValues distinct values; the workgroup copies the table into LDS before anything else (the
// only workgroup barrier of this kernel, passed by every wave before any of them can leave).
// LIST: the launch covers only the tiles named in tile_list (`ntiles` of them) -- what is left for this kernel of a plan
// whose other tiles belong to a window kernel (a launch over ALL tiles in which 99.8 % of the waves read a descriptor
// and leave cost the KKT-like matrix 61 of 1040 us).
template <int TILE, bool C16, bool X32, bool XCD, int ABL = 0, int XW = 0, bool PANELS = false, bool VI = false, bool PEER = false, bool LIST = false>
__global__ __launch_bounds__(256, (TILE <= 512 ? 8 : 4)) void csr_wavetile_kernel(
    int ntiles, const int4 * __restrict__ desc, const int32_t * __restrict__ p,
    const int32_t * __restrict__ j, const uint16_t * __restrict__ j16,
    const double * __restrict__ a, const double * __restrict__ x, const double * y_in_arg, double * y_arg,
    int nnz_total, int cols, int exact_order_and_nt, const int32_t * __restrict__ patterns, PanelInfo pinfo,
    const uint8_t * __restrict__ vidx = nullptr, const double * __restrict__ vtable = nullptr, int nvalues = 0,
    PeerY peers = PeerY{}, const int32_t * __restrict__ tile_list = nullptr)
{
    static_assert(!LIST || (!PANELS && !VI && !XCD), "the tile list is for the plain variants that follow a window kernel");
    static_assert(!PEER || !PANELS, "column panels add partial sums atomically: nothing to forward");
    // y_out = y_in + A*x.  The two may be the same array (y += A*x, the reference's form) or two
    // different ones (a partitioned multiply whose previous result is still being gathered); every
    // row is read and written by the same lane, so the in-place case needs no ordering.
    constexpr int QUADS = TILE / 256; // 16-byte column loads per lane
    // exact_order_and_nt: bit 0 = SPMV_HIP_FLAG_EXACT_ORDER, bit 1 = write y non-temporally (the matrix streams from HBM)
    const int exact_order = exact_order_and_nt & 1;
    const bool nt_y = (exact_order_and_nt & 2) != 0;
    const int group_rows = (exact_order_and_nt >> 8) & 15; // bits 8-11: rows per group of the plan's group tiles (csr_blocktile.hpp), 0 = none
    __shared__ __attribute__((aligned(16))) double prod_all[4][TILE + 4];
    __shared__ __attribute__((aligned(16))) uint32_t first_row_all[C16 ? 4 : 1][C16 ? kShiftedMaxLen : 1]; // shifted tiles: the first row's columns
    __shared__ double vtab_lds[VI ? kMaxIndexedValues : 1];             // VI variant: the value dictionary

    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    int w;
    double * y = y_arg;
    const double * y_in = y_in_arg;
    if (PANELS) {
        const int pk = (int) blockIdx.x & 7;
        w = pinfo.first[pk] + ((int) blockIdx.x >> 3) * 4 + wave;
        if (w >= pinfo.first[pk + 1])
            return;
        y = y_arg - (size_t) pk * (size_t) pinfo.rows; // virtual row v of panel pk is row v - pk * rows
    } else {
        w = (XCD ? xcd_remap(blockIdx.x, (ntiles + 3) >> 2, true) : (int) blockIdx.x) * 4 + wave;
        if (!VI && w >= ntiles)
            return; // whole wave leaves; no workgroup barrier in the kernels without a value dictionary
        if (LIST)
            w = __builtin_amdgcn_readfirstlane(tile_list[w]);
    }
    double * prod = prod_all[wave];
    // XW variant: the tile's window of x lives in the END of the wave's product slice (round 4; it had 2 KB of its own: 26.6 KB per
    // workgroup, 6 waves per SIMD -- now 18.5 KB, 8).  The slots it covers belong to the products of the tile's LAST quad only
    // (256 * (QUADS - 1) <= TILE + 4 - XW), and a lane has read the x entries of that quad before any lane writes a product of
    // it: same-wave LDS operations execute in order, and the compiler keeps a store behind the loads that may alias it.
    static_assert(XW == 0 || TILE + 4 - XW >= 256 * (QUADS - 1), "the window may only overlap the last quad's product slots");
    double * xwin = prod + (XW ? TILE + 4 - XW : 0);

    // (VI: waves past the last tile read its descriptor and leave after the table barrier)
    const int wd = VI ? (w < ntiles ? w : ntiles - 1) : w;
    const TilePair dp = load_tile_pair(desc, wd);
    const int4 d0 = dp.d0, d1 = dp.d1;
    ValueLookup vtab{vtab_lds, false, 0.0, 0.0};
    if (VI) {
        vtab.tiny = nvalues <= 2; // kernel-uniform
        if (vtab.tiny) {
            vtab.t0 = vtable[0]; // scalar loads (the table is padded to kMaxIndexedValues entries)
            vtab.t1 = vtable[1];
        } else {
            // the table load travels together with the descriptor loads; the only workgroup barrier of this kernel
            if (threadIdx.x < kMaxIndexedValues)
                vtab_lds[threadIdx.x] = vtable[threadIdx.x];
            __syncthreads();
        }
        if (w >= ntiles)
            return;
    }
    const int r0 = __builtin_amdgcn_readfirstlane(d0.x & ~kTileFlagPartial);
    const int partial = __builtin_amdgcn_readfirstlane(d0.x & kTileFlagPartial);
    const int k0 = __builtin_amdgcn_readfirstlane(d0.y);
    const int meta = __builtin_amdgcn_readfirstlane(d0.z);
    if (C16 && (meta & kTileMetaBlockWin))
        return; // done by csr_blockwin_kernel (second launch of the same multiply)
    const int maxlen = meta & 0xFFFF;
    const int lanes_log2 = (meta >> kTileMetaLanesShift) & 0x7;
    const int cbase = __builtin_amdgcn_readfirstlane(d0.w);
    const int r1 = __builtin_amdgcn_readfirstlane(d1.x & ~kTileFlagPartial);
    const int k1 = __builtin_amdgcn_readfirstlane(d1.y);
    const int nrows = r1 - r0;
    const int kb = k0 & ~3;

    // kTileMetaFast (set by the host): a non-empty stream tile whose last quad lies inside the
    // arrays, i.e. everything but long rows, tiles of empty rows and the ragged end of the matrix
    if (meta & kTileMetaFast) {
        // ---- stream tile, fast path ----------------------------------------------------
        if (VI && !kViAblate && C16 && TILE == 512 && !PANELS && (meta & kTileMetaValueRows) && (meta & kTileMetaShifted) && (meta & kTileMetaUniform)
            && (maxlen <= kLanePerRowMaxLen ? (lanes_log2 == 0 && nrows >= 2) : nrows >= kConstantRowMinRows)) {
            const bool pattern = (meta & kTileMetaPattern) != 0;
            tile_rows_pairs_constant<X32, PEER>(pattern ? patterns + (size_t) cbase * kPatStride + kPatRel : j + k0, pattern ? r0 : 0,
                                                vidx + kb, vtab, x, y_in, y, peers, r0, lane, maxlen, k0 - kb, nrows);
            return;
        }
        if (C16 && TILE == 512 && !PANELS && is_masked_stencil_tile(meta)) {
            // rows that follow the stencil's pattern with some positions missing (the boundary of a structured grid): a lane per row,
            // the rows' masks in the tile's 16-bit column slots (csr_stenciltile.hpp); under a value dictionary an index byte per entry
            const double * yin_t = y_in + r0;
            const double yA = yin_t[lane < nrows ? lane : nrows - 1];
            const double yB = yin_t[lane + kWave < nrows ? lane + kWave : nrows - 1];
            const int32_t * pat = patterns + (size_t) cbase * kPatStride;
            double zA, zB;
            tile_rows_masked_stencil<QUADS, X32, VI && !kViAblate>(prod, pat + kPatRel, __builtin_amdgcn_readfirstlane(pat[0]), j16 + k0, a + kb, vidx + kb,
                                                                   vtab, x, cols, r0, (k1 - 1 - kb) & ~3, lane, k0 - kb, nrows, zA, zB);
            if (lane < nrows)
                y_store<PEER, false>(y, peers, r0 + lane, yA + zA, nt_y);
            if (lane + kWave < nrows)
                y_store<PEER, false>(y, peers, r0 + lane + kWave, yB + zB, nt_y);
            return;
        }
        if (TILE == 512 && !PANELS && nrows > 1 && k1 - kb > TILE && lanes_log2 == 6) {
            // several rows of more than 512 entries each: in registers, one butterfly per row (tile_common.hpp)
            const bool narrow = C16 && (meta & kTileMetaNarrow);
            tile_rows_long_registers<X32>(p, j, j16, narrow, a, x, narrow ? cbase : 0, (unsigned) (cols - 1 - (narrow ? cbase : 0)), y_in, r0, k0, k1,
                                          nrows, lane, [&](int idx, double v) { y_store<PEER, false>(y, peers, idx, v, nt_y); });
            return;
        }
        if (TILE == 512 && !PANELS && nrows > 1 && k1 - kb > TILE) {
            tile_rows_multi_window<TILE, QUADS, C16, X32, VI>(prod, p, j, j16, a, x, y_in, r0, k0, k1, nrows, meta, cbase, cols, lane, vidx, vtab,
                                                              [&](int idx, double v) { y_store<PEER, false>(y, peers, idx, v, nt_y); });
            return;
        }
        if (C16 && !VI && !PANELS && TILE == 512 && (meta & kTileMetaBlock3) && !(meta & kTileMetaGroupRows) && !exact_order) {
            // dense 3 x 3 blocks (csr_blocktile.hpp): one 16-bit number per block instead of a column per entry, no row_ptr
            if (meta & kTileMetaBlock3Masked) // blocks with entries missing, off the grid of column triples: a 32-bit word per block
                tile_rows_block3<true>(prod, reinterpret_cast<const uint32_t *>(j16 + mask_stream_offset(nnz_total)) + mask_stream_index(k0), a,
                                       x + cbase, y_in, r0, k0, k1, nrows, lane, [&](int idx, double v) { y_store<PEER, false>(y, peers, idx, v, nt_y); });
            else
                tile_rows_block3<false>(prod, j16 + block_stream_offset(nnz_total) + block_stream_index(k0), a, x + cbase, y_in, r0, k0, k1, nrows,
                                        lane, [&](int idx, double v) { y_store<PEER, false>(y, peers, idx, v, nt_y); });
            return;
        }
        // (1) loads nobody waits for yet: row_ptr pair and old y of this lane's row
        const int sub = lane >> lanes_log2;
        const int part = lane & ((1 << lanes_log2) - 1);
        const int rowi = sub < nrows ? sub : nrows - 1; // clamp instead of branching
        double * yt = y + r0;
        int ps, pe;
        if (meta & kTileMetaUniform) {
            // all rows equally long (the interior of any stencil): row bounds follow from the
            // descriptor, row_ptr is not read at all
            ps = k0 + rowi * maxlen;
            pe = ps + maxlen;
        } else {
            const int32_t * pt = p + r0;
            ps = pt[rowi];
            pe = pt[rowi + 1];
        }
        const double * yin_t = y_in + r0;
        // (value-dictionary variant: y is read once and written once per launch -- non-temporal, to keep it out of
        // the way of x in the caches: 143 -> 139 us)
        // (the value-reading stencil tiles WRITE y non-temporally too, round 5; reading it that way as well was measured 0.8 % slower)
        const bool stencil_values = !VI && C16 && TILE == 512 && !PANELS && XW == 0 && ABL == 0 && (meta & kTileMetaShifted) && (meta & kTileMetaUniform)
            && lanes_log2 == 0 && maxlen <= kLanePerRowMaxLen;
        const double yv = (PANELS || (VI && (kViAblate & 2))) ? 0.0 // panels: the partial sums are added atomically
            : ((VI && !(kViAblate & 16)) ? __builtin_nontemporal_load(yin_t + rowi) : yin_t[rowi]);
        // a tile of short rows may hold up to 128 of them: lanes then own a second row, 64 further on
        const bool second = nrows > kWave; // wave-uniform; implies one lane per row
        int psB = 0, peB = 0;
        double yvB = 0.0;
        if (second) {
            const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
            if (meta & kTileMetaUniform) {
                psB = k0 + rowB * maxlen;
                peB = psB + maxlen;
            } else {
                psB = p[r0 + rowB];
                peB = p[r0 + rowB + 1];
            }
            if (!PANELS && !(VI && (kViAblate & 2)))
                yvB = (VI && !(kViAblate & 16)) ? __builtin_nontemporal_load(yin_t + rowB) : yin_t[rowB];
        }
        const int last = (k1 - 1 - kb) & ~3;
        if (VI && C16 && TILE == 512 && !PANELS && (meta & kTileMetaShifted) && (meta & kTileMetaUniform)
            && lanes_log2 == 0 && maxlen <= kLanePerRowMaxLen) {
            // equally long shifted rows under a value dictionary: a lane per row, nothing parked in LDS
            const bool pattern = (meta & kTileMetaPattern) != 0;
            double zA, zB;
            if ((meta & kTileMetaValueRows) && !kViAblate)
                tile_rows_uniform_constant<X32>(pattern ? patterns + (size_t) cbase * kPatStride + kPatRel : j + k0, pattern ? r0 : 0,
                                                vidx + kb, vtab, x, lane, maxlen, k0 - kb, nrows, second, zA, zB);
            else
                tile_rows_uniform_indexed<X32>(prod, pattern ? patterns + (size_t) cbase * kPatStride + kPatRel : j + k0, pattern ? r0 : 0,
                                               vidx + kb, vtab, x, last, lane, maxlen, k0 - kb, nrows, second, zA, zB);
            if (lane < nrows && !((kViAblate & 8) && lane > 0))
                y_store<PEER, !(kViAblate & 32)>(y, peers, r0 + lane, yv + zA);
            if (second && lane + kWave < nrows && !(kViAblate & 8))
                y_store<PEER, !(kViAblate & 32)>(y, peers, r0 + lane + kWave, yvB + zB);
            return;
        }
        if (stencil_values) {
            const bool pattern = (meta & kTileMetaPattern) != 0;
            double zA, zB;
            tile_rows_uniform_values<QUADS, X32>(prod, pattern ? patterns + (size_t) cbase * kPatStride + kPatRel : j + k0, pattern ? r0 : 0,
                                                 a + kb, x, last, lane, maxlen, k0 - kb, nrows, second, zA, zB);
            // (y written past the caches -- non-temporal, like the dictionary variant does: it is read once and written once per
            // launch, and x wants the room: Poisson 4096^2 192.2 -> 189.9 us in alternating processes, profiles/r05_ab_misc.log,
            // r05_ab_y_nt.log)
            if (lane < nrows)
                y_store<PEER, true>(y, peers, r0 + lane, yv + zA);
            if (second && lane + kWave < nrows)
                y_store<PEER, true>(y, peers, r0 + lane + kWave, yvB + zB);
            return;
        }
        // (2) the tile's column/value quads, (3) gather x and park the rounded products; entries
        // of neighbouring tiles that share the first/last quad are multiplied as well and never
        // read back
        if (XW > 0 && C16 && (meta & kTileMetaXSeg)) {
            tile_products_xseg<QUADS, (XW > 0 ? XW : 64)>(prod, xwin,
                                      reinterpret_cast<unsigned long long *>(first_row_all[C16 ? wave : 0]),
                                      patterns + (size_t) cbase * kPatStride, r0,
                                      a + kb, x, cols - 1, last, lane,
                                      ((meta >> kTileMetaXChunksShift) & 3) + 1, maxlen > 0 ? maxlen : 1, k0 - kb);
        } else if (XW > 0 && C16 && (meta & kTileMetaXWin)) {
            // with a pattern, desc.w is its number and the smallest column follows from it
            const bool pattern = (meta & kTileMetaPattern) != 0;
            const int32_t * pat = patterns + (size_t) (pattern ? cbase : 0) * kPatStride;
            const int cb = pattern ? r0 + __builtin_amdgcn_readfirstlane(pat[3]) : cbase;
            tile_products_xwin<QUADS, (XW > 0 ? XW : 64)>(prod, xwin, first_row_all[C16 ? wave : 0], j16 + kb,
                                      pattern ? pat + kPatRel : j + k0, pattern ? r0 : 0,
                                      a + kb, x + cb, cb, (unsigned) (cols - 1 - cb), last, lane,
                                      ((meta >> kTileMetaXChunksShift) & 3) + 1, (meta & kTileMetaShifted) != 0,
                                      maxlen > 0 ? maxlen : 1, k0 - kb, pattern && maxlen <= kPatPackedMaxLen ? pat + kPatWin4 : nullptr);
        }
        else if (C16 && (meta & kTileMetaShifted)) {
            const bool pattern = (meta & kTileMetaPattern) != 0;
            tile_products_shifted<QUADS, X32, VI>(prod, first_row_all[C16 ? wave : 0],
                                              pattern ? patterns + (size_t) cbase * kPatStride + kPatRel : j + k0, pattern ? r0 : 0,
                                              a + kb, x, (unsigned) (cols - 1), last, lane, maxlen, k0 - kb, vidx + kb, vtab);
        }
        else if (C16 && !VI && !PANELS && TILE == 512 && ABL == 0 && (meta & kTileMetaBlock3) && (meta & kTileMetaGroupRows) && !exact_order
                 && (group_rows == 2 || group_rows == 4)) {
            // rows in groups of 2 or 4 with the same columns (csr_blocktile.hpp): one column list and one x per group; the row
            // sums below are the plain tile's
            // (a WIDE group tile -- no 16-bit columns -- keeps 32-bit absolute group columns in its own slots of the 16-bit stream)
            const bool gwide = !(meta & kTileMetaNarrow);
            const uint16_t * gt = gwide ? j16 + wide_group_first_slot(k0) : j16 + block_stream_offset(nnz_total) + group_stream_index(k0, group_rows);
            const double * gx = gwide ? x : x + cbase;
            const unsigned glimit = (unsigned) (cols - 1 - (gwide ? 0 : cbase));
            if (group_rows == 2)
                tile_products_grouped<2>(prod, gt, a + kb, gx, glimit, ps - kb, pe - ps, lanes_log2, nrows, k0 - kb, k1 - k0, lane, (meta & kTileMetaGroupPairs) != 0, gwide);
            else
                tile_products_grouped<4>(prod, gt, a + kb, gx, glimit, ps - kb, pe - ps, lanes_log2, nrows, k0 - kb, k1 - k0, lane, (meta & kTileMetaGroupPairs) != 0, gwide);
        }
        else if (C16 && (meta & kTileMetaNarrow))
            tile_products_narrow<QUADS, ABL, VI>(prod, j16 + kb, a + kb, x + cbase, (unsigned) (cols - 1 - cbase), last, lane, vidx + kb, vtab);
        else
            tile_products_wide<QUADS, X32, VI>(prod, j + kb, a + kb, x, last, lane, vidx + kb, vtab);
        // same-wave LDS operations execute in order; the fences only pin the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (4) row sums from LDS
        const int s = ps - kb;
        const int e_row = pe - kb;
        double z;
        if ((ABL & 2) || (VI && (kViAblate & 4))) {
            z = prod[s];
        } else if (lanes_log2 == 0) { // short rows: one lane per row, the reference's order
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
        if (sub < nrows && part == 0 && !(VI && (kViAblate & 8) && lane > 0)) {
            if (PANELS)
                unsafeAtomicAdd(yt + sub, z);
            else if (VI)
                y_store<PEER, !(kViAblate & 32)>(y, peers, r0 + sub, yv + z);
            else
                y_store<PEER, false>(y, peers, r0 + sub, yv + z, nt_y);
        }
        if (second) {
            const double zB = ((ABL & 2) || (VI && (kViAblate & 4))) ? prod[psB - kb] : tile_row_sum<1>(prod, psB - kb, peB - kb, 0, maxlen);
            if (lane + kWave < nrows && !(VI && (kViAblate & 8))) {
                if (PANELS)
                    unsafeAtomicAdd(yt + lane + kWave, zB);
                else if (VI)
                    y_store<PEER, !(kViAblate & 32)>(y, peers, r0 + lane + kWave, yvB + zB);
                else
                    y_store<PEER, false>(y, peers, r0 + lane + kWave, yvB + zB, nt_y);
            }
        }
    } else if (!partial && k1 - kb <= TILE) {
        // ---- stream tile at the ragged end of the arrays, or a tile of empty rows: scalar
        // loads, one lane per row
        for (int k = k0 + lane; k < k1; k += kWave)
            prod[k - kb] = a[k] * x[j[k]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int r = lane; r < nrows; r += kWave) {
            const int s = p[r0 + r] - kb, e_row = p[r0 + r + 1] - kb;
            double z = 0.0;
            for (int k = s; k < e_row; ++k)
                z += prod[k];
            if (PANELS)
                unsafeAtomicAdd(y + r0 + r, z);
            else
                y_store<PEER, false>(y, peers, r0 + r, y_in[r0 + r] + z, nt_y);
        }
    } else if (!exact_order) {
        // ---- one long row, or one chunk of a very long row: the whole wave, in registers (tile_common.hpp) ----------
        const bool narrow = C16 && (meta & kTileMetaNarrow);
        const double z = long_row_sum<X32>(j, j16, narrow, a, x, narrow ? cbase : 0, (unsigned) (cols - 1 - (narrow ? cbase : 0)), k0, k1, lane);
        if (lane == 0) {
            if (partial || PANELS)
                unsafeAtomicAdd(y + r0, z); // the host made y_out a copy of y_in first if they differ (never under PEER)
            else
                y_store<PEER, false>(y, peers, r0, y_in[r0] + z, nt_y);
        }
    } else {
        // ---- one long row in the reference's order: lane 0 adds tiles of products ---------
        double z = 0.0;
        for (int t0 = k0; t0 < k1; t0 += TILE) {
            const int t1 = (t0 + TILE < k1) ? t0 + TILE : k1;
            for (int k = t0 + lane; k < t1; k += kWave)
                prod[k - t0] = a[k] * x[j[k]];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0)
                for (int k = 0; k < t1 - t0; ++k)
                    z += prod[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane == 0) {
            if (PANELS)
                unsafeAtomicAdd(y + r0, z);
            else
                y_store<PEER, false>(y, peers, r0, y_in[r0] + z, nt_y);
        }
    }
}

} // namespace spmv
