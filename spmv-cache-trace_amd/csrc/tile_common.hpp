// tile_common.hpp -- what the tile kernels share: the XCD-aware workgroup order, the tile descriptor format, and the
// helpers every tile class uses (stream loads, the x gather, row sums from the wave's LDS slice, where a tile's
// values come from, products of tiles with 32-bit and 16-bit columns).
//
// Built with -ffp-contract=off: a product is rounded before it is added, as in the reference's x86-64 -O3 build
// (no FMA), so every path that adds a row's products left to right with one lane is bit-identical to the
// reference loop (src/matrix/csr-matrix-spmv.cpp:29-32, src/matrix/ell-matrix.cpp:251-257).
// None of this is GEMM-shaped: ~0.13 flop/byte, HBM-bound.  No MFMA on purpose.
//
// Kernels that are not templates are `static`: the headers are included by several translation units.
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
// CSR, wave tiles ("wavetile"): per-wavefront row ownership, no workgroup barrier.
//
// The host cuts the rows into tiles owned by ONE wave: up to 128 consecutive rows (two per lane
// when rows are short) holding at most TILE stored entries (counted from the 4-aligned start).
// A tile is described by an int4 {first row | flags, first entry, meta, column base} with
//   meta = longest row | log2(lanes per row) << 16 | narrow << 24 | fast << 25 | uniform << 26
//          | shifted << 27 | x window << 28 | (window chunks - 1) << 29 | window of runs << 31;
// tile w ends where tile w+1 starts.  In a uniform tile (all rows equally long, e.g. the
// interior of a stencil) the row bounds follow from the descriptor and row_ptr is not read.
// A wave reads its descriptor pair and then has everything it needs to issue ALL its
// independent loads back to back -- the row_ptr pair and old y of the lane's row, then the
// column/value quads (16 B per lane, coalesced whatever the row lengths are) -- so a tile costs
// three dependent memory round trips (descriptor -> streams -> x) instead of the six of a
// row_ptr-driven kernel.  The rounded products are parked in the wave's private LDS slice
// (same-wave LDS operations execute in order: no barrier, no wait beyond the data dependence),
// then each row is added up by L lanes, L chosen by the host from the tile's longest row
// (<= 16 entries per lane); L = 1 walks the row left to right exactly like the reference loop.
//
// The kernel is also kept lean in issued instructions, which at 5 entries per row is
// what bounds it next to HBM: no per-entry predicates (entries of neighbouring tiles
// that share a 16-byte quad are multiplied too, their products are simply never
// read), clamped indices instead of divergent branches, the per-tile integer
// divisions done once on the host (descriptor .z/.w), and a row loop whose trip count
// is wave-uniform (the tile's longest row).
//
// Compressed column indices: when all columns of a tile lie within 65536 of the tile's
// smallest column (any banded matrix), the plan keeps them as 16-bit offsets from that
// base in a second index stream, and the tile reads 2 instead of 4 bytes per entry
// (10 instead of 12 with the value) and gathers x through a scalar base + 32-bit offset.
// Entries of neighbouring tiles that share a boundary quad decode against the wrong base;
// their offset is clamped into x so that the (never used) gather stays in bounds.
//
// A row longer than TILE is a tile by itself (the wave strides it); rows longer than
// kSplitThreshold (512 entries) are cut into chunks spread over several waves (bit 31 of the row
// field), each adding its partial sum with one fp64 atomic.
// ---------------------------------------------------------------------------------
constexpr int kTileFlagPartial = (int) 0x80000000u;
constexpr int kTileMetaLanesShift = 16;
constexpr int kTileMetaBlockWin = 1 << 20; // the tile belongs to csr_blockwin_kernel; csr_wavetile_kernel skips it
constexpr int kTileMetaPattern = 1 << 19; // shifted tile: desc.w is a pattern number (first-row columns = first row + pattern)
// value dictionary only: every row of this (uniform, shifted, one-lane-per-row) tile carries the FIRST row's dictionary
// indices -- a constant-coefficient stencil -- so the tile's index bytes are not read at all (set / cleared for every tile
// by value_rows_mark_kernel whenever a dictionary is built; kernels without a dictionary never look at it)
constexpr int kTileMetaValueRows = 1 << 22;
// ... rows of up to this many entries; rows longer than the one-lane-per-row limit (16) take the constant-row path only in tiles of
// at least kConstantRowMinRows rows (the dictionary launch's re-cut tiles of 128): a plan tile of 18 rows of 27 would leave 55 lanes idle
constexpr int kConstantRowMaxLen = 64;
constexpr int kConstantRowMinRows = 64;
constexpr int kTileMetaNarrow = 1 << 24;
constexpr int kTileMetaFast = 1 << 25;
constexpr int kTileMetaUniform = 1 << 26; // every row of the tile has exactly `longest row` entries
// uniform + every row has the columns of the tile's first row shifted by its distance from it
// (the interior of a stencil, a band matrix): only the first row's columns are read
constexpr int kTileMetaShifted = 1 << 27;
// narrow, and the tile's whole column range fits the x window of the XW kernel variant:
// bits 29-30 hold the number of 64-entry chunks of x to stage, minus one
constexpr int kTileMetaXWin = 1 << 28;
constexpr int kTileMetaXChunksShift = 29;
// shifted tile whose x entries -- `len` runs of `rows` consecutive entries, runs that touch or
// overlap merged -- fit the window: the plan keeps, in the tile's unused 16-bit column slots,
// the window position of every first-row column and the x offset of every window slot
constexpr int kTileMetaXSeg = (int) 0x80000000u;
// A window-of-runs tile refers (desc.w) to a pattern shared by all tiles with the same row count
// and the same first-row columns relative to the first row index -- the whole interior of a
// stencil is one pattern -- so the window tables cost no HBM traffic and no per-tile round trip.
// Record, in 32-bit words: [0] row length, [1] rows, [2] window slots a window of runs would use
// (2^20 = none worked out), [3] smallest first-row column - first row index;
// [16..144) first-row columns - first row index; [144..176) window position of each row position
// (16 bits each); [176..432) x index - first row index of every window slot.
// Rows of up to 64 entries also get, per row position p, one 64-bit word of FOUR 16-bit window positions: those of the entries
// p, p + 1, p + 2, p + 3 of the tile's row-major entry sequence relative to the row of entry p (an entry past the row's end is
// the next row's: that position's value + the rows passed), so that a lane finds the four window slots of its quad with ONE
// division and ONE LDS read (slot = field + the row of the quad's first entry) instead of four of each: [432..560) for the
// window of runs, [560..688) for the contiguous window (column - smallest column of the first row).  Round 4.
constexpr int kPatStride = 688;
constexpr int kPatRel = 16, kPatXoff = 144, kPatSrc = 176, kPatXoff4 = 432, kPatWin4 = 560;
constexpr int kPatPackedMaxLen = 64;
constexpr int kMaxPatterns = 64;

// Plan-time counters are kept in kCountStripes copies -- a workgroup adds to copy (its number mod kCountStripes) -- and summed on
// the host: one address takes about 10^8 atomic adds per second, and the plan passes issue one or more per TILE (the queen-like
// matrix's 808 K tiles x 4 counters cost its block check 39 ms of its 40; round 4).
constexpr int kCountStripes = 128;
constexpr int kCountWidth = 8; // counters per copy
template <typename T>
__device__ __forceinline__ void striped_add(T * counters, int idx, T v)
{
    atomicAdd(counters + (size_t) (blockIdx.x & (kCountStripes - 1)) * kCountWidth + idx, v);
}

// native vector types: __builtin_nontemporal_load wants these, not HIP's wrapper structs
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// A tile's descriptor and its successor's (row / entry bounds come from the pair) at a WAVE-UNIFORM index through the
// scalar data cache, eight dwords at once: the constant address space tells the compiler that the load may go there (the
// descriptors are read-only for the whole launch).  As `desc[w]`, `desc[w + 1]` the compiler issued VECTOR loads and, in the
// kernels with an early exit between the uses of the two, fetched the pair in two dependent round trips (round 3:
// tools/wave_trace.py showed 1.55 us of a 4.9 us wave lifetime waiting for the descriptor).
struct TilePair {
    int4 d0, d1;
};
__device__ __forceinline__ TilePair load_tile_pair(const int4 * __restrict__ desc, int w)
{
    typedef int v8i_a16 __attribute__((ext_vector_type(8), aligned(16)));
    typedef const v8i_a16 __attribute__((address_space(4))) * const_ptr;
    const v8i_a16 v = *reinterpret_cast<const_ptr>(reinterpret_cast<uintptr_t>(desc + w));
    TilePair t;
    t.d0 = make_int4(v[0], v[1], v[2], v[3]);
    t.d1 = make_int4(v[4], v[5], v[6], v[7]);
    return t;
}

template <typename T, bool NT>
__device__ __forceinline__ T stream_load(const T * ptr)
{
    if (NT)
        return __builtin_nontemporal_load(ptr);
    return *ptr;
}

// x[c] with a 32-bit byte offset from a scalar base when x is smaller than 4 GiB
// (global_load saddr + voffset: one shift instead of 64-bit address arithmetic)
template <bool X32>
__device__ __forceinline__ double gather_x(const double * __restrict__ x, int c)
{
    if (X32)
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + ((unsigned) c << 3));
    return x[c];
}

// Sum of one row's products from the wave's LDS slice by L lanes; the trip count is wave-uniform
// (the tile's longest row), lanes whose row is finished add +0.0 without reading LDS.  That is an
// identity: z starts at +0.0 and can never become -0.0 (a sum that cancels rounds to +0.0), so the
// bits match a loop that simply stops at the end of the row.  (Reading a shared zero slot instead
// of predicating the read was measured slower: 245 vs 222 us on the 27-point stencil.)
template <int L>
__device__ __forceinline__ double tile_row_sum(const double * prod, int s, int e_row, int part, int trips)
{
    double z = 0.0;
    int k = s + part;
    for (int t = 0; t < trips; ++t, k += L) {
        const double v = (k < e_row) ? prod[k] : 0.0;
        z += v;
    }
    return group_sum<L>(z);
}

// Where a tile's values come from.  VI = false: the value array (two 16-byte loads per lane and quad).
// VI = true (the plan holds a value dictionary: the matrix has at most kMaxIndexedValues distinct values --
// a pattern / graph matrix, a constant-coefficient stencil, a mesh of identical elements): one BYTE per
// entry from the plan's index stream (one dword per lane and quad) and the value itself out of a table
// in LDS.  The doubles are the stored ones bit for bit; the tile streams 1 instead of 8 bytes per entry.
constexpr int kMaxIndexedValues = 128;

// Where an index byte finds its double: the dictionary in LDS -- or, for a dictionary of one or two values (a
// pattern or graph matrix; the 5-point stencil's -1 and 4), two scalar registers and a select: no table, no look-up,
// and no workgroup barrier at the start of the kernel.
struct ValueLookup {
    const double * tab;
    bool tiny;
    double t0, t1;
    __device__ __forceinline__ double operator[](unsigned b) const { return tiny ? (b ? t1 : t0) : tab[b]; }
};

template <int QUADS, bool VI>
struct TileValues {
    v2d va[QUADS], vb[QUADS];
    unsigned vi[VI ? QUADS : 1];

    // at / vit already point at the tile's 4-aligned first entry
    __device__ __forceinline__ void load(const double * __restrict__ at, const uint8_t * __restrict__ vit, int last, int lane)
    {
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            int o = 256 * q + 4 * lane;
            o = o < last ? o : last; // lanes past the tile's end re-read its last quad
            if (VI) {
                vi[q] = *reinterpret_cast<const unsigned *>(vit + o);
            } else {
                va[q] = *reinterpret_cast<const v2d *>(at + o);
                vb[q] = *reinterpret_cast<const v2d *>(at + o + 2);
            }
        }
    }
    __device__ __forceinline__ void resolve(ValueLookup vtab)
    {
        if (VI) {
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                va[q] = v2d{vtab[vi[q] & 0x7Fu], vtab[(vi[q] >> 8) & 0x7Fu]};
                vb[q] = v2d{vtab[(vi[q] >> 16) & 0x7Fu], vtab[(vi[q] >> 24) & 0x7Fu]};
            }
        }
    }
};

// Products of one quad-set with 32-bit column indices.
template <int QUADS, bool X32, bool VI = false>
__device__ __forceinline__ void tile_products_wide(
    double * prod, const int32_t * __restrict__ jt, const double * __restrict__ at,
    const double * __restrict__ x, int last, int lane, const uint8_t * __restrict__ vit = nullptr, ValueLookup vtab = ValueLookup{nullptr, false, 0.0, 0.0})
{
    v4i c[QUADS];
    TileValues<QUADS, VI> vals;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last; // lanes past the tile's end re-read its last quad
        c[q] = *reinterpret_cast<const v4i *>(jt + o);
    }
    vals.load(at, vit, last, lane);
    vals.resolve(vtab);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            const double q0 = vals.va[q].x * gather_x<X32>(x, c[q].x);
            const double q1 = vals.va[q].y * gather_x<X32>(x, c[q].y);
            const double q2 = vals.vb[q].x * gather_x<X32>(x, c[q].z);
            const double q3 = vals.vb[q].y * gather_x<X32>(x, c[q].w);
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
}

// The same with 16-bit column offsets from the tile's base: xt = x + base (scalar), limit =
// last valid offset from the base (cols - 1 - base).
template <int QUADS, int ABL, bool VI = false>
__device__ __forceinline__ void tile_products_narrow(
    double * prod, const uint16_t * __restrict__ jt, const double * __restrict__ at,
    const double * __restrict__ xt, unsigned limit, int last, int lane, const uint8_t * __restrict__ vit = nullptr,
    ValueLookup vtab = ValueLookup{nullptr, false, 0.0, 0.0})
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u c[QUADS];
    TileValues<QUADS, VI> vals;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last;
        c[q] = *reinterpret_cast<const v2u *>(jt + o); // four 16-bit offsets
    }
    vals.load(at, vit, last, lane);
    vals.resolve(vtab);
    const char * xb = reinterpret_cast<const char *>(xt);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            unsigned c0 = min(c[q].x & 0xFFFFu, limit), c1 = min(c[q].x >> 16, limit);
            unsigned c2 = min(c[q].y & 0xFFFFu, limit), c3 = min(c[q].y >> 16, limit);
            if (ABL & 1) { // timing experiment only: every lane gathers the same four x entries
                c0 &= 1; c1 &= 1; c2 &= 1; c3 &= 1;
            }
            const double q0 = vals.va[q].x * *reinterpret_cast<const double *>(xb + (c0 << 3));
            const double q1 = vals.va[q].y * *reinterpret_cast<const double *>(xb + (c1 << 3));
            const double q2 = vals.vb[q].x * *reinterpret_cast<const double *>(xb + (c2 << 3));
            const double q3 = vals.vb[q].y * *reinterpret_cast<const double *>(xb + (c3 << 3));
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
}

// One LONG ROW (or one chunk of it), entries [k0, k1), summed by the whole wave IN REGISTERS (round 5).  Every lane of the wave
// works for the same row, so nothing has to be parked: the 4-aligned interior [ka, kz) is read exactly like a tile's streams --
// per lane and step one 16-byte quad of columns (8 bytes where the plan holds 16-bit offsets for this tile) and two 16-byte
// loads of values, two quads per lane in flight, 512 entries per wave and step -- the products go straight into four
// accumulators per lane, and ONE butterfly over the wave closes the row.  The up to three entries in front of ka and behind kz
// are taken by single lanes with scalar-width loads from the 32-bit columns (always there).  This replaces a loop of 4-byte
// column / 8-byte value loads (a quarter of the bytes per instruction) and, for ELLPACK rows of more than 2048 entries, the
// column-major copy (src/matrix/ell-matrix.cpp:243-258: the row loop the reference runs; 1e-10 class, not its order).
template <bool X32>
__device__ __forceinline__ double long_row_sum(
    const int32_t * __restrict__ j, const uint16_t * __restrict__ j16, bool narrow, const double * __restrict__ a,
    const double * __restrict__ x, int cbase, unsigned limit /* last valid offset from cbase */, int k0, int k1, int lane)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
    const int ka = (k0 + 3) & ~3, kz = k1 & ~3;
    if (ka >= kz) { // fewer than one whole quad: the lanes take an entry each
        for (int k = k0 + lane; k < k1; k += kWave)
            z0 += a[k] * x[j[k]];
        return group_sum<kWave>(z0);
    }
    if (lane < ka - k0)
        z0 += a[k0 + lane] * x[j[k0 + lane]];
    if (lane >= 4 && lane - 4 < k1 - kz)
        z1 += a[kz + lane - 4] * x[j[kz + lane - 4]];
    const char * xb = reinterpret_cast<const char *>(x + cbase);
    for (int o = ka + 4 * lane; o < kz; o += 2 * 4 * kWave) {
        const bool two = o + 4 * kWave < kz;          // per lane: the second quad of this step exists
        const int o2 = two ? o + 4 * kWave : o;        // (otherwise the first is read again and nothing added)
        unsigned c[8];
        if (narrow) {
            const v2u ca = *reinterpret_cast<const v2u *>(j16 + o), cb = *reinterpret_cast<const v2u *>(j16 + o2);
            c[0] = ca.x & 0xFFFFu; c[1] = ca.x >> 16; c[2] = ca.y & 0xFFFFu; c[3] = ca.y >> 16;
            c[4] = cb.x & 0xFFFFu; c[5] = cb.x >> 16; c[6] = cb.y & 0xFFFFu; c[7] = cb.y >> 16;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                c[i] = min(c[i], limit);
        } else {
            const v4i ca = *reinterpret_cast<const v4i *>(j + o), cb = *reinterpret_cast<const v4i *>(j + o2);
            c[0] = (unsigned) ca.x; c[1] = (unsigned) ca.y; c[2] = (unsigned) ca.z; c[3] = (unsigned) ca.w;
            c[4] = (unsigned) cb.x; c[5] = (unsigned) cb.y; c[6] = (unsigned) cb.z; c[7] = (unsigned) cb.w;
        }
        const v2d a0 = *reinterpret_cast<const v2d *>(a + o), a1 = *reinterpret_cast<const v2d *>(a + o + 2);
        const v2d b0 = *reinterpret_cast<const v2d *>(a + o2), b1 = *reinterpret_cast<const v2d *>(a + o2 + 2);
        double xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            xv[i] = narrow ? *reinterpret_cast<const double *>(xb + (c[i] << 3)) : gather_x<X32>(x, (int) c[i]);
        z0 += a0.x * xv[0];
        z1 += a0.y * xv[1];
        z2 += a1.x * xv[2];
        z3 += a1.y * xv[3];
        if (two) {
            z0 += b0.x * xv[4];
            z1 += b0.y * xv[5];
            z2 += b1.x * xv[6];
            z3 += b1.y * xv[7];
        }
    }
    return group_sum<kWave>((z0 + z1) + (z2 + z3));
}

// SEVERAL long rows (each of more than 512 entries) owned by one wave, in registers (round 5): the multi-window tiles' fill --
// five rows of 601 entries are 5.9 windows of 512, where one such row alone leaves its second window 83 % empty -- without
// their LDS round trip.  The wave walks the tile's entries in steps of 512 like long_row_sum; a lane adds those of its eight
// products that belong to the row that is currently open to ONE accumulator; where a row ends inside a step (wave-uniform: at
// most twice per step for rows of this length) the accumulator is summed over the wave, lane `row` keeps the result, and the
// next row opens on the same products.  One butterfly per row, no LDS, no atomics, the same y on every run.
template <bool X32, typename YStore>
__device__ __forceinline__ void tile_rows_long_registers(
    const int32_t * __restrict__ p, const int32_t * __restrict__ j, const uint16_t * __restrict__ j16, bool narrow,
    const double * __restrict__ a, const double * __restrict__ x, int cbase, unsigned limit, const double * y_in,
    int r0, int k0, int k1, int nrows, int lane, YStore && store)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int kb = k0 & ~3, lastq = (k1 - 1) & ~3;
    const double yv = y_in[r0 + (lane < nrows ? lane : nrows - 1)];
    const char * xb = reinterpret_cast<const char *>(x + cbase);
    int row = 0;
    int ps = k0, pe = __builtin_amdgcn_readfirstlane(p[r0 + 1]);
    double acc = 0.0, zmine = 0.0;
    for (int base = kb; base < k1; base += 2 * 4 * kWave) { // wave-uniform
        const int oA = base + 4 * lane, oB = oA + 4 * kWave;   // this lane's two quads (their entries: the masks below)
        const int cA = oA < lastq ? oA : lastq, cB = oB < lastq ? oB : lastq; // (past the tile's end: its last quad again, masked out)
        unsigned c[8];
        if (narrow) {
            const v2u ca = *reinterpret_cast<const v2u *>(j16 + cA), cb = *reinterpret_cast<const v2u *>(j16 + cB);
            c[0] = ca.x & 0xFFFFu; c[1] = ca.x >> 16; c[2] = ca.y & 0xFFFFu; c[3] = ca.y >> 16;
            c[4] = cb.x & 0xFFFFu; c[5] = cb.x >> 16; c[6] = cb.y & 0xFFFFu; c[7] = cb.y >> 16;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                c[i] = min(c[i], limit);
        } else {
            const v4i ca = *reinterpret_cast<const v4i *>(j + cA), cb = *reinterpret_cast<const v4i *>(j + cB);
            c[0] = (unsigned) ca.x; c[1] = (unsigned) ca.y; c[2] = (unsigned) ca.z; c[3] = (unsigned) ca.w;
            c[4] = (unsigned) cb.x; c[5] = (unsigned) cb.y; c[6] = (unsigned) cb.z; c[7] = (unsigned) cb.w;
        }
        const v2d a0 = *reinterpret_cast<const v2d *>(a + cA), a1 = *reinterpret_cast<const v2d *>(a + cA + 2);
        const v2d b0 = *reinterpret_cast<const v2d *>(a + cB), b1 = *reinterpret_cast<const v2d *>(a + cB + 2);
        double q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            q[i] = narrow ? *reinterpret_cast<const double *>(xb + (c[i] << 3)) : gather_x<X32>(x, (int) c[i]);
        q[0] *= a0.x; q[1] *= a0.y; q[2] *= a1.x; q[3] *= a1.y;
        q[4] *= b0.x; q[5] *= b0.y; q[6] *= b1.x; q[7] *= b1.y;
        for (;;) { // the rows that have entries in this step, one after the other
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                part += (oA + i >= ps && oA + i < pe) ? q[i] : 0.0;
                part += (oB + i >= ps && oB + i < pe) ? q[4 + i] : 0.0;
            }
            acc += part;
            if (pe > base + 2 * 4 * kWave)
                break; // the open row goes on in the next step
            const double z = group_sum<kWave>(acc);
            if (lane == row)
                zmine = z;
            acc = 0.0;
            ++row;
            if (row >= nrows)
                break;
            ps = pe;
            pe = __builtin_amdgcn_readfirstlane(p[r0 + row + 1]);
            if (ps >= base + 2 * 4 * kWave)
                break; // the next row begins with the next step
        }
    }
    if (lane < nrows)
        store(r0 + lane, yv + zmine);
}

} // namespace spmv
