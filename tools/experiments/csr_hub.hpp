// csr_hub.hpp -- hub columns of a graph matrix: a dense copy of the x entries that many rows refer to.
//
// A web graph's links go mostly to pages nearby -- and, a quarter of them, to a few tens of thousands of POPULAR pages
// scattered over the whole index space.  Every XCD's private L2 has to fetch the 128-byte line of each such x entry once
// per launch for 8 useful bytes: on the webbase-like matrix 84 of the 149 MB a launch moves across the fabric
// (profiles/r03_prof_webbase_csr_summary.md: 2.29 x the algorithmic bytes, L2 hit rate 0.44).  Plan time: in-degrees by one
// pass of atomics over the columns; the columns with at least `threshold` references become hubs, numbered densely; the plan
// keeps its own 32-bit column stream in which a hub column is replaced by 0x80000000 | its number.  Multiply: a small launch
// copies x[hub column] into the plan's dense array (each scattered line read once, by one XCD), and the tile kernel takes a
// hub entry's x from there -- sixteen hubs to a line, a few hundred KB that stay in every L2.  Same products, same order,
// same bits as without.
//
// OPT-IN (SPMV_HIP_FLAG_HUB_COLUMNS): built in round 4 on round 3's estimate that the 84 MB were hub lines, and measured on the
// webbase-like matrix: tile kernel 24.55 vs 24.82 us, PMC traffic 143.0 vs 149.2 MB -- and 5 us of second launch on top (26.6 vs
// 23.9 us per multiply).  The traffic is the long TAIL: 82 % of the popular links go to pages with fewer than 8 references, spread
// over the whole index space, so that nearly every line of x (8 MB) is fetched by every one of the eight L2s (profiles/
// r04_prof_webbase_hub_summary.md, r04_prof_webbase_nohub_summary.md).  A dense copy cannot help a tail.
#pragma once

#include "tile_common.hpp"

namespace spmv {

constexpr int kHubBit = (int) 0x80000000u;

static __global__ __launch_bounds__(256) void hub_count_kernel(long long n, const int32_t * __restrict__ j, int32_t * __restrict__ degree)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride)
        atomicAdd(degree + j[k], 1);
}

// flag[c] = degree[c] >= threshold; stats[0] += hubs, stats[1] += the entries that refer to them
static __global__ __launch_bounds__(256) void hub_flag_kernel(int cols, const int32_t * __restrict__ degree, int threshold,
                                                               int32_t * __restrict__ flag, unsigned long long * __restrict__ stats)
{
    const int c = (int) (blockIdx.x * 256 + threadIdx.x);
    int d = 0;
    if (c < cols) {
        d = degree[c];
        d = d >= threshold ? d : 0;
        flag[c] = d > 0;
    }
    unsigned long long hubs = d > 0, entries = (unsigned long long) d;
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) {
        hubs += __shfl_xor(hubs, s);
        entries += __shfl_xor(entries, s);
    }
    if ((int) __lane_id() == 0 && hubs) {
        atomicAdd(stats, hubs);
        atomicAdd(stats + 1, entries);
    }
}

// slot = exclusive scan of flag: hub_column[slot[c]] = c for every hub
static __global__ __launch_bounds__(256) void hub_list_kernel(int cols, const int32_t * __restrict__ flag, const int32_t * __restrict__ slot,
                                                               int32_t * __restrict__ hub_column)
{
    const int c = (int) (blockIdx.x * 256 + threadIdx.x);
    if (c < cols && flag[c])
        hub_column[slot[c]] = c;
}

static __global__ __launch_bounds__(256) void hub_remap_kernel(long long n, const int32_t * __restrict__ j, const int32_t * __restrict__ flag,
                                                                const int32_t * __restrict__ slot, int32_t * __restrict__ jh)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const int c = j[k];
        jh[k] = flag[c] ? (kHubBit | slot[c]) : c;
    }
}

// every multiply: the dense copy of the hub entries of x
static __global__ __launch_bounds__(256) void hub_gather_kernel(int nhubs, const int32_t * __restrict__ hub_column, const double * __restrict__ x,
                                                                 double * __restrict__ hubx)
{
    const int s = (int) (blockIdx.x * 256 + threadIdx.x);
    if (s < nhubs)
        hubx[s] = x[hub_column[s]];
}

// x for a column of the plan's own stream: a hub's from the dense copy
__device__ __forceinline__ double gather_x_hub(const double * __restrict__ x, const double * __restrict__ hubx, int c)
{
    const double * base = c < 0 ? hubx : x;
    return base[c & 0x7FFFFFFF];
}

// tile_products_wide with the plan's hub-aware column stream
template <int QUADS, bool VI>
__device__ __forceinline__ void tile_products_wide_hub(
    double * prod, const int32_t * __restrict__ jt, const double * __restrict__ at, const double * __restrict__ x,
    const double * __restrict__ hubx, int last, int lane, const uint8_t * __restrict__ vit, ValueLookup vtab)
{
    v4i c[QUADS];
    TileValues<QUADS, VI> vals;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        int o = 256 * q + 4 * lane;
        o = o < last ? o : last;
        c[q] = *reinterpret_cast<const v4i *>(jt + o);
    }
    vals.load(at, vit, last, lane);
    vals.resolve(vtab);
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int o = 256 * q + 4 * lane;
        if (o <= last) {
            const double q0 = vals.va[q].x * gather_x_hub(x, hubx, c[q].x);
            const double q1 = vals.va[q].y * gather_x_hub(x, hubx, c[q].y);
            const double q2 = vals.vb[q].x * gather_x_hub(x, hubx, c[q].z);
            const double q3 = vals.vb[q].y * gather_x_hub(x, hubx, c[q].w);
            v2d * dst = reinterpret_cast<v2d *>(prod + o);
            dst[0] = v2d{q0, q1};
            dst[1] = v2d{q2, q3};
        }
    }
}

} // namespace spmv
