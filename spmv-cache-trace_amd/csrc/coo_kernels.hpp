// coo_kernels.hpp -- COO in any order (the semantics of coo_spmv_atomic, src/matrix/coo-matrix.cpp:287-309).
#pragma once

#include "tile_common.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// COO in any order.  Each wave takes 64 consecutive entries per step, forms the
// products, adds runs of equal row index inside the wave (segmented inclusive scan
// over head flags, ds_bpermute moves) and issues ONE fp64 atomic per run, so a
// row-sorted file costs ~1 atomic per row per wave and an unsorted one degrades to
// one atomic per entry -- the semantics of the reference's coo_spmv_atomic
// (src/matrix/coo-matrix.cpp:287-309).
// ---------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void coo_kernel(
    int nnz, const int32_t * __restrict__ ri, const int32_t * __restrict__ ci,
    const double * __restrict__ v, const double * __restrict__ x, double * __restrict__ y)
{
    const int lane = (int) __lane_id();
    const long long total = (long long) gridDim.x * BLOCK;
    const long long gid = (long long) blockIdx.x * BLOCK + threadIdx.x;
    for (long long base = 0; base < nnz; base += total) { // uniform trip count
        const long long k = base + gid;
        const bool valid = k < nnz;
        int r = -1;
        double s = 0.0;
        if (valid) {
            r = ri[k];
            s = v[k] * x[ci[k]];
        }
        const int rprev = lane_up(r, 1);
        int head = (lane == 0) || (rprev != r);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const double sp = lane_up(s, d);
            const int hp = lane_up(head, d);
            if (lane >= d && !head) {
                s += sp;
                head |= hp;
            }
        }
        const int rnext = lane_down1(r);
        const bool tail = (lane == kWave - 1) || (rnext != r);
        if (valid && tail)
            unsafeAtomicAdd(y + r, s);
    }
}

// The same semantics with 256 consecutive entries per wave, four per lane (one 16-byte load of
// each index stream and two of the values per lane).  A lane first adds its own entries run by
// run; runs that begin and end inside the lane are complete.  Across lanes only one (row, sum)
// pair per lane takes part in the segmented scan: the lane's last run.  A lane's first run is
// closed by that lane (carry of the preceding lanes + its own part), its last run by the lane
// where the row changes next.  Row-sorted input with 5 entries per row thus issues one atomic
// instruction with ~51 active lanes per 256 entries instead of four with ~13: fp64 atomics are
// paid per wave instruction (MI355X_MICROARCH.md, "Global float atomics").
// Entries past nnz (last wave only) are loaded one by one and carry row -1.
// PANELS: the triplets are the context's own copy, grouped by column panel (an eighth of the
// columns each; inside a panel in row order), every panel padded with row -1 entries to whole
// workgroups.  Workgroup b takes its 1024 entries from panel b % 8, so each XCD gathers from one
// eighth of x out of its own L2 (see csr_wavetile_kernel, PANELS).
struct CooPanels {
    long long start[9]; // entries [start[k], start[k+1]) are panel k; multiples of 1024
};

template <bool PANELS>
__global__ __launch_bounds__(256) void coo_wide_kernel(
    int nnz, const int32_t * __restrict__ ri, const int32_t * __restrict__ ci,
    const double * __restrict__ v, const double * __restrict__ x, double * __restrict__ y, CooPanels cp)
{
    const int lane = (int) __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    long long base;
    if (PANELS) {
        const int pk = (int) blockIdx.x & 7;
        base = cp.start[pk] + ((long long) (blockIdx.x >> 3) * 4 + wave) * 256;
        if (base >= cp.start[pk + 1])
            return;
    } else {
        base = ((long long) blockIdx.x * 4 + wave) * 256;
    }
    if (base >= nnz)
        return; // whole wave
    const int o = 4 * lane;
    int r[4];
    double q[4];
    if (base + 256 <= nnz) {
        const v4i rr = *reinterpret_cast<const v4i *>(ri + base + o);
        const v4i cc = *reinterpret_cast<const v4i *>(ci + base + o);
        const v2d va = *reinterpret_cast<const v2d *>(v + base + o);
        const v2d vb = *reinterpret_cast<const v2d *>(v + base + o + 2);
        r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; r[3] = rr.w;
        q[0] = va.x * x[cc.x];
        q[1] = va.y * x[cc.y];
        q[2] = vb.x * x[cc.z];
        q[3] = vb.y * x[cc.w];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long k = base + o + i;
            const bool valid = k < nnz;
            r[i] = valid ? ri[k] : -1;
            q[i] = valid ? v[k] * x[ci[k]] : 0.0;
        }
    }
    // runs inside the lane
    const int r_first = r[0];
    int r_cur = r[0];
    double s_cur = q[0], s_first = 0.0;
    bool multi = false;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (r[i] == r_cur) {
            s_cur += q[i];
        } else {
            if (!multi) {
                s_first = s_cur;
                multi = true;
            } else if (r_cur >= 0) {
                unsafeAtomicAdd(y + r_cur, s_cur); // began and ended in this lane
            }
            r_cur = r[i];
            s_cur = q[i];
        }
    }
    // segmented inclusive scan over the lanes' last runs
    const int r_last = r_cur;
    const int r_prev = lane_up(r_last, 1);
    const bool cont = lane > 0 && r_prev == r_first; // my first run continues the previous lane's last
    int head = multi || !cont;
    double s = s_cur;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const double sp = lane_up(s, d);
        const int hp = lane_up(head, d);
        if (lane >= d && !head) {
            s += sp;
            head |= hp;
        }
    }
    const double s_prev = lane_up(s, 1);
    const int next_cont = lane_down1((int) cont);
    const bool tail = lane == kWave - 1 || !next_cont;
    // one (row, sum) per lane in the common case: the end of its first run, or of its only run
    const int r_out = multi ? r_first : (tail ? r_last : -1);
    const double s_out = multi ? (cont ? s_prev + s_first : s_first) : s;
    if (r_out >= 0)
        unsafeAtomicAdd(y + r_out, s_out);
    if (multi && tail && r_last >= 0)
        unsafeAtomicAdd(y + r_last, s);
}

// Plan-time: how many 256-entry chunks of the (row-sorted) triplets have columns that reach further
// than one column panel -- what "scattered" means for the COO panels.
static __global__ __launch_bounds__(256) void coo_chunk_spread_kernel(
    int nnz, int width, const int32_t * __restrict__ ci, int * __restrict__ count)
{
    const int lane = (int) __lane_id();
    const long long base = ((long long) blockIdx.x * 4 + (threadIdx.x >> 6)) * 256;
    if (base >= nnz)
        return;
    int lo = 0x7FFFFFFF, hi = -1;
    for (int i = lane; i < 256 && base + i < nnz; i += kWave) {
        const int c = ci[base + i];
        lo = min(lo, c);
        hi = max(hi, c);
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        lo = min(lo, __shfl_xor(lo, d));
        hi = max(hi, __shfl_xor(hi, d));
    }
    if (lane == 0 && hi - lo >= width)
        atomicAdd(count, 1);
}

} // namespace spmv
