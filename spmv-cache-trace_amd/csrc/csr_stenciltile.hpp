// csr_stenciltile.hpp -- MASKED STENCIL TILES (round 5): the boundary rows of a structured grid.
//
// A stencil's interior rows are shifted copies of each other (kTileMetaShifted: the lane-per-row tiles of csr_wavetile.hpp read no
// column index at all), but a tile that holds a boundary row -- the cell at the end of a grid line lacks its +x neighbour -- is
// neither uniform nor shifted and fell back to 32-bit columns, row_ptr and row sums through LDS.  On a 4096^2 grid that is one
// tile in 250; on a 256^3 grid every grid line of 256 cells ends inside a tile of 73 rows: 28 % of the tiles, and the 7-point
// Laplacian ran at 0.82 of the roofline where the 2-D one runs at 0.95 (profiles/r05_structure_zoo_before.log).
//
// Such a row still follows the stencil: its columns are a SUBSET of {row + rel[pos]}, rel = the relative columns of the interior
// rows (a pattern record the plan already holds).  A masked stencil tile keeps, per row, a 16-bit mask of the positions that are
// stored -- written at plan time into the tile's own 16-bit column slots, which it does not read as columns -- and multiplies with
// ONE LANE PER ROW like tile_rows_uniform_values: the tile's values are loaded coalesced and parked in the wave's LDS slice, where
// a row's values start follows from a prefix sum of the rows' mask counts (one DPP scan), x is read 512 contiguous bytes per
// position (clamped into x where the neighbour does not exist), and every row is added left to right by one lane in column order:
// the reference's bits (src/matrix/csr-matrix-spmv.cpp:29-32).  No column index, no row_ptr: 8 bytes per entry + 2 per row (under a
// value dictionary: 1 byte per entry + 2 per row).
#pragma once

#include "tile_common.hpp"
#include "csr_blocktile.hpp" // wave_inclusive_scan

namespace spmv {

// tile marks: kTileMetaShifted WITHOUT kTileMetaUniform (a shifted tile proper is always uniform) + kTileMetaPattern, desc.w = pattern
__device__ __host__ __forceinline__ bool is_masked_stencil_tile(int meta)
{
    return (meta & kTileMetaShifted) && !(meta & kTileMetaUniform) && (meta & kTileMetaPattern);
}
constexpr int kStencilMaskMaxLen = 16; // positions of the pattern = bits of a row's mask

// VI: the plan holds a value dictionary (a constant-coefficient Laplacian: two values): the tile's values are one index BYTE per entry
// (vit, 512 bytes per tile parked as two dwords per lane) and the doubles come out of the dictionary, like tile_rows_uniform_indexed.
template <int QUADS, bool X32, bool VI>
__device__ __forceinline__ void tile_rows_masked_stencil(
    double * prod, const int32_t * __restrict__ rel /* the pattern's relative columns */, int len, const uint16_t * __restrict__ rowmask /* j16 + k0 */,
    const double * __restrict__ at, const uint8_t * __restrict__ vit, ValueLookup vtab, const double * __restrict__ x, int cols, int r0, int last,
    int lane, int lead, int nrows, double & zA, double & zB)
{
    static_assert(!VI || QUADS == 2, "the index bytes of a 512-entry tile are two dwords per lane");
    TileValues<QUADS, false> vals;
    unsigned vi[2] = {0u, 0u};
    if (VI) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int o = 256 * q + 4 * lane;
            o = o < last ? o : last; // lanes past the tile's end re-read its last dword (and park it where nobody looks)
            vi[q] = *reinterpret_cast<const unsigned *>(vit + o);
        }
    } else {
        vals.load(at, nullptr, last, lane);
    }
    const bool second = nrows > kWave; // wave-uniform
    const int rowA = lane < nrows ? lane : nrows - 1;
    const int rowB = lane + kWave < nrows ? lane + kWave : nrows - 1;
    const unsigned mA = lane < nrows ? rowmask[rowA] : 0u;
    const unsigned mB = (second && lane + kWave < nrows) ? rowmask[rowB] : 0u;
    // where each row's values start in the tile: rows 0 .. 63 first (A), then rows 64 .. (B); two 16-bit fields, one scan
    const int cnt = __builtin_popcount(mA) | (__builtin_popcount(mB) << 16);
    const int incl = wave_inclusive_scan(cnt);
    const int total_a = __builtin_amdgcn_readlane(incl, kWave - 1) & 0xFFFF;
    int kA = lead + (incl & 0xFFFF) - __builtin_popcount(mA);
    int kB = lead + total_a + (incl >> 16) - __builtin_popcount(mB);
    if (VI) {
        unsigned * vw = reinterpret_cast<unsigned *>(prod);
        vw[lane] = vi[0];
        vw[kWave + lane] = vi[1];
    } else {
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            const int o = 256 * q + 4 * lane;
            if (o <= last) {
                v2d * dst = reinterpret_cast<v2d *>(prod + o);
                dst[0] = vals.va[q];
                dst[1] = vals.vb[q];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t * vbytes = reinterpret_cast<const uint8_t *>(prod);
    zA = 0.0;
    zB = 0.0;
    constexpr int CH = 4;
    const int top = cols - 1;
    for (int p0 = 0; p0 < len; p0 += CH) { // wave-uniform
        double xa[CH], xb[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                const int c = scalar_load_i32(rel + p0 + i) + r0;
                // (a neighbour that does not exist lies outside x, or wraps into the next grid line: read SOMETHING in bounds, never used)
                const int ca = c + rowA, cb = c + rowB;
                xa[i] = gather_x<X32>(x, ca < 0 ? 0 : (ca > top ? top : ca));
                if (second)
                    xb[i] = gather_x<X32>(x, cb < 0 ? 0 : (cb > top ? top : cb));
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (p0 + i < len) {
                if ((mA >> (p0 + i)) & 1u) {
                    zA += (VI ? vtab[vbytes[kA] & 0x7Fu] : prod[kA]) * xa[i];
                    ++kA;
                }
                if (second && ((mB >> (p0 + i)) & 1u)) {
                    zB += (VI ? vtab[vbytes[kB] & 0x7Fu] : prod[kB]) * xb[i];
                    ++kB;
                }
            }
        }
    }
}

// Plan time (spmv_hip_plan_csr_repack: row_ptr next to the columns), one wave per tile: a stream tile of 2 ... 128 non-empty rows of
// at most 16 entries that is NOT a shifted tile, and that no window kernel has claimed, is tried against the patterns of `list`
// (the stencils found by sampling rows, longest first, then the plan's own most frequent ones): if every row's columns are among {row + rel[pos]} in ascending position, the rows'
// masks go to the tile's 16-bit column slots (slot k0 + row: a tile holds at least one entry per row) and the tile is marked.
// count[0] += tiles, count[1] += their entries.
struct StencilTryList {
    int n;
    int pattern[8];
};

// What do the rows of the matrix look like?  `nsamples` rows scattered over the matrix, one thread each: a row of 2 ... 16
// entries reports its length and its columns relative to its own index (out[17 i] = length, then the columns); any other row 0.
// The host counts equal reports: the frequent ones are the stencils of the matrix -- also where NO tile is a shifted tile because
// every tile holds the end of a grid line (a 40^3 grid: lines of 40 cells, tiles of 73 rows), so that the plan has no pattern yet.
static __global__ __launch_bounds__(256) void csr_row_pattern_sample_kernel(
    int rows, int nsamples, const int32_t * __restrict__ p, const int32_t * __restrict__ j, int32_t * __restrict__ out)
{
    const int i = (int) (blockIdx.x * 256 + threadIdx.x);
    if (i >= nsamples)
        return;
    // (scattered by a multiplicative hash, not evenly spaced: rows 128 apart in a 64^3 grid are ALL the first cell of a grid line)
    const int r = (int) (((unsigned long long) i * 0x9E3779B97F4A7C15ull >> 20) % (unsigned long long) rows);
    const int k0 = p[r], len = p[r + 1] - k0;
    int32_t * o = out + (size_t) i * (kStencilMaskMaxLen + 1);
    if (len < 2 || len > kStencilMaskMaxLen) {
        o[0] = 0;
        return;
    }
    o[0] = len;
    for (int k = 0; k < len; ++k)
        o[1 + k] = j[k0 + k] - r;
}

static __global__ __launch_bounds__(256) void csr_stencil_mask_kernel(
    int ntiles, int tile, int4 * __restrict__ desc, const int32_t * __restrict__ p, const int32_t * __restrict__ j,
    uint16_t * __restrict__ j16, const int32_t * __restrict__ patterns, StencilTryList list, int dry_run, unsigned long long * __restrict__ count)
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
    const int nrows = r1 - r0, maxlen = meta & 0xFFFF;
    // (dry_run: nothing is written, and the tiles a BLOCK WINDOW has claimed are counted as well -- where masked stencil tiles and
    // shifted tiles together are most of the matrix the plan gives its block windows up first, like it does for block tiles)
    const int claimed = kTileMetaShifted | kTileMetaXWin | kTileMetaXSeg | kTileMetaPattern | (dry_run ? 0 : kTileMetaBlockWin) | (1 << 21) /* balanced */
        | (1 << 23) /* block */;
    if ((d0.x & kTileFlagPartial) || !(meta & kTileMetaFast) || (meta & claimed) || nrows < 2 || nrows > 2 * kWave || maxlen > kStencilMaskMaxLen
        || maxlen < 1 || ((meta >> kTileMetaLanesShift) & 7) != 0 || k1 - (k0 & ~3) > tile || k1 - k0 < nrows)
        return;
    // this lane's rows: lane and lane + 64
    int ps[2], pe[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lane + kWave * h;
        const int rc = row < nrows ? row : nrows - 1;
        ps[h] = p[r0 + rc];
        pe[h] = p[r0 + rc + 1];
    }
    for (int t = 0; t < list.n; ++t) { // wave-uniform
        const int q = list.pattern[t];
        const int32_t * pat = patterns + (size_t) q * kPatStride;
        const int len = pat[0];
        if (len > kStencilMaskMaxLen || len < maxlen)
            continue;
        unsigned mask[2] = {0u, 0u};
        int ok = 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = lane + kWave * h;
            if (row < nrows) {
                int pos = 0;
                if (pe[h] <= ps[h])
                    ok = 0; // an empty row has no slot for its mask
                for (int k = ps[h]; k < pe[h] && ok; ++k) {
                    const int o = j[k] - (r0 + row);
                    while (pos < len && pat[kPatRel + pos] != o)
                        ++pos;
                    if (pos >= len)
                        ok = 0; // not a column of the stencil (or not in ascending position)
                    else
                        mask[h] |= 1u << pos++;
                }
            }
        }
        if (!__all(ok))
            continue;
        if (!dry_run) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = lane + kWave * h;
                if (row < nrows)
                    j16[k0 + row] = (uint16_t) mask[h];
            }
        }
        if (lane == 0) {
            if (!dry_run) {
                desc[w].z = (meta & ~kTileMetaUniform & ~kTileMetaNarrow) | kTileMetaShifted | kTileMetaPattern;
                desc[w].w = q;
            }
            striped_add(count, 0, 1ull);
            striped_add(count, 1, (unsigned long long) (k1 - k0));
        }
        return;
    }
}

// the plan gave up its SEGMENT windows for stencil tiles: their tiles' 16-bit stream holds window slots, not column offsets, so
// whichever of them is not taken as a masked stencil tile goes back to its 32-bit columns
static __global__ __launch_bounds__(256) void csr_clear_segwin_kernel(int ntiles, int4 * __restrict__ desc)
{
    const int w = (int) (blockIdx.x * 256 + threadIdx.x);
    if (w < ntiles && (desc[w].z & kTileMetaBlockWin))
        desc[w].z &= ~(kTileMetaBlockWin | kTileMetaNarrow);
}

} // namespace spmv
