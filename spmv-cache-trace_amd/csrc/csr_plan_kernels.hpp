// csr_plan_kernels.hpp -- plan-time passes over the device columns: tile classes, 16-bit column stream, tile patterns.
#pragma once

#include "csr_wavetile.hpp"

namespace spmv {

// Plan-time pass (one wave per tile) that classifies the stream tiles and writes the second index
// stream; counts[0..2] receive the number of narrow / shifted / windowed tiles.
//  narrow:  the columns span < 65536: their offsets from the smallest one go to j16 (10 instead of
//           12 bytes per entry), desc.w = that column;
//  shifted: at least two equally long rows (<= kShiftedMaxLen entries) that all repeat the first
//           row's columns moved right by the row distance -- whatever range they span: the kernel
//           reads the first row's 32-bit columns and no others (8 bytes per entry);
//  window:  x staged through LDS by the XW kernel variant, for a narrow tile whose columns span
//           < 256 (kTileMetaXWin) or a shifted tile whose merged runs of x fit 256 slots
//           (kTileMetaXSeg; the run tables go to the tile's j16 slots, which a shifted tile does not
//           read), in both cases only if every slot is used at least twice (measured: 81/row band
//           5.6 uses per slot 284 -> 257 us, 27-point stencil 2.7 uses 216 -> 199 us, 5-point
//           stencil 1.65 uses 44.9 -> 51.3 us on a cache-resident 2048^2 grid).
static __global__ __launch_bounds__(256) void csr_tile_compress_kernel(
    int ntiles, int tile, int4 * __restrict__ desc, const int32_t * __restrict__ j,
    uint16_t * __restrict__ j16, int * __restrict__ counts, int detect_shifted,
    unsigned long long * __restrict__ fingerprint, int panel_width)
{
    const int wave = (int) threadIdx.x >> 6;
    const int lane = (int) __lane_id();
    const int w = blockIdx.x * 4 + wave;
    if (w >= ntiles)
        return;
    const int4 d0 = desc[w];
    const int4 d1 = desc[w + 1];
    const int k0 = d0.y, k1 = d1.y;
    // (a tile of SEVERAL rows that holds more than `tile` entries is a multi-window tile, csr_wavetile.hpp: it may have 16-bit
    // columns like any other, but none of the classes below that park a whole tile's columns or x in LDS)
    const bool big = k1 - (k0 & ~3) > tile;
    if (k1 <= k0)
        return; // empty tiles
    // one long row, or one chunk of it (long_row_sum, tile_common.hpp): 16-bit columns where the chunk's columns allow, nothing else
    const bool long_row = (d0.x & kTileFlagPartial) || (big && (d1.x & ~kTileFlagPartial) - (d0.x & ~kTileFlagPartial) < 2);
    int cmin = 0x7FFFFFFF, cmax = -1;
    for (int k = k0 + lane; k < k1; k += kWave) {
        const int c = j[k];
        cmin = c < cmin ? c : cmin;
        cmax = c > cmax ? c : cmax;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int omin = __shfl_xor(cmin, d), omax = __shfl_xor(cmax, d);
        cmin = omin < cmin ? omin : cmin;
        cmax = omax > cmax ? omax : cmax;
    }
    if (cmin < 0)
        return;
    // counts[4]: tiles whose columns reach further than one column panel (an eighth of the matrix):
    // what "scattered" means for spmv_hip_plan_csr_repack
    if (lane == 0 && cmax - cmin >= panel_width)
        striped_add(counts, 4, 1);
    const bool narrow = cmax - cmin < 65536;
    if (narrow)
        for (int k = k0 + lane; k < k1; k += kWave)
            j16[k] = (uint16_t) (j[k] - cmin);
    const int len = d0.z & 0xFFFF;
    int shifted = detect_shifted && !big && !long_row && (d0.z & kTileMetaFast) && (d0.z & kTileMetaUniform) && len >= 1
                  && len <= kShiftedMaxLen && k1 - k0 >= 2 * len && tile <= 1024;
    if (shifted) {
        int ok = 1;
        for (int k = k0 + lane; k < k1; k += kWave) {
            const int t = k - k0, r = t / len;
            ok &= (j[k] == j[k0 + (t - r * len)] + r);
        }
        shifted = __all(ok);
    }
    int xwin = 0;
    if (narrow && !big && !long_row && cmax - cmin < 256 && k1 - k0 >= 2 * (cmax - cmin + 1))
        xwin = kTileMetaXWin | (((cmax - cmin) >> 6) << kTileMetaXChunksShift);
    if (shifted && fingerprint) {
        // the tile's shape: (row length, rows, first-row columns relative to the first row);
        // csr_pattern_assign_kernel gives it its pattern later (and a window of runs if it pays)
        const int r0 = d0.x & ~kTileFlagPartial;
        unsigned long long h = 0;
        for (int pos = lane; pos < len; pos += kWave) {
            unsigned long long t = (unsigned long long) (unsigned) (j[k0 + pos] - r0) + 0x9E3779B97F4A7C15ull * (unsigned long long) (pos + 1);
            t ^= t >> 29;
            t *= 0xBF58476D1CE4E5B9ull;
            t ^= t >> 32;
            h += t;
        }
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1)
            h += __shfl_xor(h, s);
        h += 0x94D049BB133111EBull * (unsigned long long) len + 0xD6E8FEB86659FD93ull * (unsigned long long) ((k1 - k0) / len);
        if (lane == 0)
            fingerprint[w] = h | 1ull; // 0 = no shape
    }
    if (!narrow && !shifted)
        return;
    if (lane == 0) {
        desc[w].z = d0.z | (narrow ? kTileMetaNarrow : 0) | (shifted ? kTileMetaShifted : 0) | xwin;
        if (narrow) {
            desc[w].w = cmin;
            striped_add(counts, 0, 1);
        }
        if (shifted)
            striped_add(counts, 1, 1);
        if (xwin)
            striped_add(counts, 2, 1);
    }
}

// One wave per pattern: write the record of pattern p from its representative tile.  Rows of up
// to 64 entries also get the layout of a window of runs: lane = position in the row; a position
// whose column is within `rows` of the previous one continues its run, so the runs' x ranges
// [column, column + rows) are merged where they touch or overlap.
static __global__ __launch_bounds__(64) void csr_pattern_build_kernel(
    const int * __restrict__ rep_tile, const int4 * __restrict__ desc, const int32_t * __restrict__ j,
    int32_t * __restrict__ patterns)
{
    const int lane = (int) __lane_id();
    const int w = rep_tile[blockIdx.x];
    int32_t * pat = patterns + (size_t) blockIdx.x * kPatStride;
    const int4 d0 = desc[w];
    const int r0 = d0.x & ~kTileFlagPartial;
    const int k0 = d0.y, k1 = desc[w + 1].y;
    const int len = d0.z & 0xFFFF;
    const int nrows = (k1 - k0) / len;
    int relmin = 0x7FFFFFFF;
    for (int pos = lane; pos < len; pos += kWave) {
        const int rel = j[k0 + pos] - r0;
        pat[kPatRel + pos] = rel;
        relmin = rel < relmin ? rel : relmin;
    }
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) {
        const int o = __shfl_xor(relmin, s);
        relmin = o < relmin ? o : relmin;
    }
    int total = 1 << 20;
    if (len <= kWave) {
        const int col = lane < len ? j[k0 + lane] : 0;
        const int d = col - __shfl_up(col, 1);
        const int fresh = lane == 0 || d < 0 || d > nrows;
        int xo = (lane == 0 || lane >= len) ? 0 : (fresh ? nrows : d);
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1) {
            const int up = __shfl_up(xo, s);
            if (lane >= s)
                xo += up;
        }
        total = __shfl(xo, len - 1) + nrows;
        const int d_next = __shfl_down(d, 1), fresh_next = __shfl_down(fresh, 1);
        uint16_t * xoff = reinterpret_cast<uint16_t *>(pat + kPatXoff);
        const int col0 = __shfl(col, 0);
        for (int i = (total < 256 ? total : 256) + lane; i < 256; i += kWave)
            pat[kPatSrc + i] = col0 - r0; // unused slots: any valid entry
        if (lane < len) {
            xoff[lane] = (uint16_t) (xo < 65535 ? xo : 65535);
            const int cnt = (lane == len - 1 || fresh_next) ? nrows : d_next;
            for (int i = 0; i < cnt && xo + i < 256; ++i)
                pat[kPatSrc + xo + i] = col - r0 + i;
        }
        // four window positions per row position (see tile_common.hpp), for both window forms
        unsigned f_runs[4], f_win[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane + i, rr = q / len, qq = q - rr * len; // (every lane takes part in the shuffles)
            const int xq = __shfl(xo, qq), cq = __shfl(col, qq);
            f_runs[i] = (unsigned) min(xq + rr, 65535);
            f_win[i] = (unsigned) min(max(cq - r0 - relmin + rr, 0), 65535);
        }
        if (lane < len) {
            pat[kPatXoff4 + 2 * lane] = (int32_t) (f_runs[0] | (f_runs[1] << 16));
            pat[kPatXoff4 + 2 * lane + 1] = (int32_t) (f_runs[2] | (f_runs[3] << 16));
            pat[kPatWin4 + 2 * lane] = (int32_t) (f_win[0] | (f_win[1] << 16));
            pat[kPatWin4 + 2 * lane + 1] = (int32_t) (f_win[2] | (f_win[3] << 16));
        }
    }
    if (lane == 0) {
        pat[0] = len;
        pat[1] = nrows;
        pat[2] = total;
        pat[3] = relmin;
    }
}

// One wave per tile: a candidate (fingerprint != 0) whose fingerprint is among the patterns' and
// whose shape really equals that pattern's gets the pattern number in desc.w and is marked
// kTileMetaXSeg; counts[2] += 1.
static __global__ __launch_bounds__(256) void csr_pattern_assign_kernel(
    int ntiles, int4 * __restrict__ desc, const int32_t * __restrict__ j,
    const unsigned long long * __restrict__ fingerprint, const unsigned long long * __restrict__ pattern_fp,
    int npatterns, const int32_t * __restrict__ patterns, int * __restrict__ counts)
{
    const int wave = (int) threadIdx.x >> 6;
    const int lane = (int) __lane_id();
    const int w = blockIdx.x * 4 + wave;
    if (w >= ntiles)
        return;
    const unsigned long long fp = fingerprint[w];
    if (fp == 0)
        return;
    const unsigned long long hit = __ballot(lane < npatterns && pattern_fp[lane < npatterns ? lane : 0] == fp);
    if (hit == 0)
        return;
    const int p = __builtin_ctzll(hit);
    const int32_t * pat = patterns + (size_t) p * kPatStride;
    const int4 d0 = desc[w];
    const int r0 = d0.x & ~kTileFlagPartial;
    const int k0 = d0.y, k1 = desc[w + 1].y;
    const int len = d0.z & 0xFFFF;
    int same = len == pat[0] && (k1 - k0) == pat[1] * len;
    if (same)
        for (int pos = lane; pos < len; pos += kWave)
            same &= (j[k0 + pos] - r0) == pat[kPatRel + pos];
    if (!__all(same))
        return;
    if (lane == 0) {
        // a window of runs only where every slot is used at least twice (see above), and a
        // contiguous window (already marked) is the better one where both apply
        const int total = pat[2];
        const bool window = !(d0.z & kTileMetaXWin) && total <= 256 && 2 * total <= k1 - k0;
        desc[w].z = d0.z | kTileMetaPattern
                    | (window ? (kTileMetaXSeg | (((total - 1) >> 6) << kTileMetaXChunksShift)) : 0);
        desc[w].w = p;
        if (window)
            striped_add(counts, 2, 1);
    }
}

} // namespace spmv
