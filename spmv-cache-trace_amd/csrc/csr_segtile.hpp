// csr_segtile.hpp -- balanced tiles (filled by entries, row sums by segmented reduction) for skewed rows.
#pragma once

#include "tile_common.hpp"
#ifdef SPMV_HIP_EXPERIMENTS
#include "csr_hub.hpp" // retired from the product library (internal.hpp)
#endif

namespace spmv {

// ---------------------------------------------------------------------------------
// CSR, balanced tiles ("segmented" row sums): for matrices whose rows are short on average but
// skewed (a web graph: 3 entries per row, a few rows of hundreds).  The wave tiles above give a
// row at least one lane and a long row several, so one 300-entry row confines its tile to 4 rows
// and the matrix falls apart into tiles a fifth full: 29 556 waves for 3.1 M entries, 3.6 rounds
// of waves that each wait out the same chain of memory round trips (measured, webbase-like:
// 29 us, 61 % of the wave cycles waiting on memory; profiles/r02_prof_webbase_csr_summary.md).
// Here a tile is filled by ENTRIES: up to 512 of them in up to 256 whole rows, whatever their
// lengths (rows of more than 512 entries keep the long-row path).  The products go to the wave's
// LDS slice as before; then every lane takes 8 CONSECUTIVE products and the row sums come out of a
// segmented reduction whose cost does not depend on the row lengths:
//   * every non-empty row marks its first entry's slot with its number (rowat[], 16 bit);
//   * a lane adds its 8 products run by run, left to right (a row that begins and ends inside the
//     lane is summed in the reference's order, bit for bit);
//   * runs that cross lanes meet in one segmented inclusive scan over the lanes' last runs
//     (ds_bpermute moves), and the lane in which the next row starts closes the row before it;
//   * the sums are parked in LDS by row number (the product slots are free by then: every lane
//     has its 8 products in registers), and the lanes write y for the rows they loaded y for.
// No atomics, the same result on every run.  Rows that span two or more lanes are added in a
// different order than the reference's loop: within 1e-10, not bit-identical
// (SPMV_HIP_FLAG_EXACT_ORDER keeps the one-lane-per-row tiles).
// ---------------------------------------------------------------------------------
constexpr int kSegMaxRows = 256;
constexpr int kTileMetaSeg = 1 << 21;

// VI: the plan holds a value dictionary (a pattern / graph matrix: all ones; few distinct weights): the stream tiles read one index
// byte per entry instead of eight bytes of value (see csr_wavetile_kernel); long rows keep reading the values themselves.
// HUB: the plan holds hub columns (csr_hub.hpp): tiles with 32-bit columns read the plan's own column stream `jh`, in which a hub
// column is 0x80000000 | its number in the dense copy `hubx` (filled by hub_gather_kernel just before this launch).
template <bool C16, bool X32, bool XCD, bool VI = false, bool HUB = false>
__global__ __launch_bounds__(256, 6) void csr_segtile_kernel(
    int ntiles, const int4 * __restrict__ desc, const int32_t * __restrict__ p,
    const int32_t * __restrict__ j, const uint16_t * __restrict__ j16,
    const double * __restrict__ a, const double * __restrict__ x, const double * y_in, double * y,
    int nnz_total, int cols, const uint8_t * __restrict__ vidx = nullptr, const double * __restrict__ vtable = nullptr, int nvalues = 0,
    const int32_t * __restrict__ jh = nullptr, const double * __restrict__ hubx = nullptr)
{
    constexpr int TILE = 512, QUADS = 2, RPL = kSegMaxRows / kWave; // rows per lane
    __shared__ __attribute__((aligned(16))) double prod_all[4][TILE + 4];
    __shared__ __attribute__((aligned(16))) uint16_t rowat_all[4][TILE];
    __shared__ double vtab_lds[VI ? kMaxIndexedValues : 1];

    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    // XCD: workgroups b, b + 8, b + 16, ... share an XCD and its L2; give each XCD one contiguous run of
    // tiles, so that the x entries its rows refer to (a web graph links mostly within the neighbourhood
    // of the row) collect in ONE L2 instead of being fetched over the fabric into all eight
    const int w = (XCD ? xcd_remap((int) blockIdx.x, (ntiles + 3) >> 2, true) : (int) blockIdx.x) * 4 + wave;
    if (!VI && w >= ntiles)
        return; // whole wave leaves; no workgroup barrier in the kernel without a value dictionary
    double * prod = prod_all[wave];
    uint16_t * rowat = rowat_all[wave];
    ValueLookup vtab{vtab_lds, false, 0.0, 0.0};
    if (VI) {
        vtab.tiny = nvalues <= 2; // kernel-uniform
        if (vtab.tiny) {
            vtab.t0 = vtable[0];
            vtab.t1 = vtable[1];
        } else {
            if (threadIdx.x < kMaxIndexedValues)
                vtab_lds[threadIdx.x] = vtable[threadIdx.x];
            __syncthreads(); // the only workgroup barrier; passed by every wave before any can leave
        }
        if (w >= ntiles)
            return;
    }

    const TilePair dp = load_tile_pair(desc, w);
    const int4 d0 = dp.d0, d1 = dp.d1;
    const int r0 = __builtin_amdgcn_readfirstlane(d0.x & ~kTileFlagPartial);
    const int partial = __builtin_amdgcn_readfirstlane(d0.x & kTileFlagPartial);
    const int k0 = __builtin_amdgcn_readfirstlane(d0.y);
    const int meta = __builtin_amdgcn_readfirstlane(d0.z);
    const int cbase = __builtin_amdgcn_readfirstlane(d0.w);
    const int r1 = __builtin_amdgcn_readfirstlane(d1.x & ~kTileFlagPartial);
    const int k1 = __builtin_amdgcn_readfirstlane(d1.y);
    const int nrows = r1 - r0;
    const int kb = k0 & ~3;

    if (meta & kTileMetaFast) {
        // (1) row bounds and old y of up to four rows per lane: nobody waits for these yet
        int ps[RPL], pe[RPL];
        double yv[RPL];
        const int32_t * pt = p + r0;
        const double * yin_t = y_in + r0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = lane + kWave * i;
            const int rc = row < nrows ? row : nrows - 1; // clamp instead of branching
            ps[i] = pt[rc];
            pe[i] = pt[rc + 1];
            yv[i] = yin_t[rc];
        }
        // (2) the tile's column/value quads, gather x, park the rounded products
        const int last = (k1 - 1 - kb) & ~3;
        if (C16 && (meta & kTileMetaNarrow))
            tile_products_narrow<QUADS, 0, VI>(prod, j16 + kb, a + kb, x + cbase, (unsigned) (cols - 1 - cbase), last, lane, vidx + kb, vtab);
#ifdef SPMV_HIP_EXPERIMENTS
        else if (HUB)
            tile_products_wide_hub<QUADS, VI>(prod, jh + kb, a + kb, x, hubx, last, lane, vidx + kb, vtab);
#endif
        else
            tile_products_wide<QUADS, X32, VI>(prod, j + kb, a + kb, x, last, lane, vidx + kb, vtab);
        // (3) every non-empty row marks the slot of its first entry
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        *reinterpret_cast<v4u *>(rowat + 8 * lane) = v4u{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = lane + kWave * i;
            if (row < nrows && pe[i] > ps[i])
                rowat[ps[i] - kb] = (uint16_t) row;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (4) this lane's 8 consecutive products and marks
        const int e0 = 8 * lane;
        double q[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v2d t = *reinterpret_cast<const v2d *>(prod + e0 + 2 * i);
            q[2 * i] = t.x;
            q[2 * i + 1] = t.y;
        }
        const v4u rm = *reinterpret_cast<const v4u *>(rowat + e0);
        int ra[8];
        ra[0] = (int) (rm.x & 0xFFFFu); ra[1] = (int) (rm.x >> 16);
        ra[2] = (int) (rm.y & 0xFFFFu); ra[3] = (int) (rm.y >> 16);
        ra[4] = (int) (rm.z & 0xFFFFu); ra[5] = (int) (rm.z >> 16);
        ra[6] = (int) (rm.w & 0xFFFFu); ra[7] = (int) (rm.w >> 16);
        // the row my first entry belongs to = the last mark in the lanes before me (-1: none, i.e.
        // the entries in front of the tile that share its first quad)
        int mylast = -1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (ra[i] != 0xFFFF)
                mylast = ra[i];
        int incl = mylast;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int up = lane_up(incl, d);
            if (lane >= d && incl < 0)
                incl = up;
        }
        int carry = lane_up(incl, 1);
        if (lane == 0)
            carry = -1;
        // everything below only reads registers and writes row sums: the product slots are free
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (5) runs inside the lane, left to right
        const int nend = k1 - kb;
        int cur = carry;
        double s = 0.0, s_first = 0.0;
        bool multi = false; // a row starts somewhere in this lane
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ra[i] != 0xFFFF) {
                if (!multi) {
                    s_first = s; // my part of the row that came in from the left (possibly nothing)
                    multi = true;
                } else {
                    prod[cur] = s; // began and ended inside this lane: the reference's order
                }
                cur = ra[i];
                s = 0.0;
            }
            if (e0 + i < nend)
                s += q[i];
        }
        // (6) runs that cross lanes: segmented inclusive scan over the lanes' last runs
        int head = multi ? 1 : 0;
        double sc = s;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const double sp = lane_up(sc, d);
            const int hp = lane_up(head, d);
            if (lane >= d && !head) {
                sc += sp;
                head |= hp;
            }
        }
        double s_prev = lane_up(sc, 1);
        if (lane == 0)
            s_prev = 0.0;
        if (multi && carry >= 0)
            prod[carry] = s_prev + s_first; // the row before my first mark ends here
        if (lane == kWave - 1 && cur >= 0)
            prod[cur] = sc; // the tile's last row
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (7) y for the rows this lane loaded it for (empty rows: y unchanged but copied to y_out)
        double * yt = y + r0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = lane + kWave * i;
            if (row < nrows) {
                const double z = pe[i] > ps[i] ? prod[row] : 0.0;
                yt[row] = yv[i] + z;
            }
        }
    } else if (!partial && k1 - kb <= TILE) {
        // ---- tile at the ragged end of the arrays, or a tile of empty rows: scalar loads, one lane per row
        for (int k = k0 + lane; k < k1; k += kWave)
            prod[k - kb] = a[k] * x[j[k]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int r = lane; r < nrows; r += kWave) {
            const int s0 = p[r0 + r] - kb, e_row = p[r0 + r + 1] - kb;
            double z = 0.0;
            for (int k = s0; k < e_row; ++k)
                z += prod[k];
            y[r0 + r] = y_in[r0 + r] + z;
        }
    } else {
        // ---- one long row, or one chunk of a very long row: the whole wave, in registers (tile_common.hpp)
        const bool narrow = C16 && (meta & kTileMetaNarrow);
        const double z = long_row_sum<X32>(j, j16, narrow, a, x, narrow ? cbase : 0, (unsigned) (cols - 1 - (narrow ? cbase : 0)), k0, k1, lane);
        if (lane == 0) {
            if (partial)
                unsafeAtomicAdd(y + r0, z); // the host made y_out a copy of y_in first if they differ
            else
                y[r0] = y_in[r0] + z;
        }
    }
}

} // namespace spmv
