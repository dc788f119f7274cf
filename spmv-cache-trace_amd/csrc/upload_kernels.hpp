// upload_kernels.hpp -- upload- and plan-time passes: index checks, row_ptr of sorted triplets, hybrid merge, value dictionary, checksums.
#pragma once

#include "tile_common.hpp"
#include "csr_wavetile.hpp"

namespace spmv {

// ---------------------------------------------------------------------------------
// Upload-time checks on the device (the host arrays are never walked entry by entry).
// index_check_kernel: flags[0] |= 1 if any idx[k] is outside [0, limit); with `sorted_flag`,
// flags[1] |= 1 if idx is not non-decreasing.  column_checksum_kernel: out += sum over k of
// hash(k, j[k]) -- the plan's content guard (a different array at the same address changes it).
// ---------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void index_check_kernel(
    long long n, int limit, const int32_t * __restrict__ idx, int * __restrict__ flags, int sorted_flag)
{
    const long long stride = (long long) gridDim.x * 256;
    int bad = 0, unsorted = 0;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const int v = idx[k];
        bad |= (v < 0) | (v >= limit);
        if (sorted_flag && k > 0)
            unsorted |= v < idx[k - 1];
    }
    if (__any(bad) && (int) __lane_id() == 0)
        atomicOr(flags, 1);
    if (sorted_flag && __any(unsorted) && (int) __lane_id() == 0)
        atomicOr(flags + 1, 1);
}

// row_ptr of row-sorted triplets: row_ptr[r] = first k with row[k] >= r (run-length of the row stream),
// for r = 0 .. rows; thread k fills the rows in (row[k-1], row[k]], thread nnz the tail.
static __global__ __launch_bounds__(256) void rowptr_from_sorted_kernel(
    long long nnz, int rows, const int32_t * __restrict__ row, int32_t * __restrict__ row_ptr)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k <= nnz; k += stride) {
        const int lo = k == 0 ? 0 : row[k - 1] + 1;
        const int hi = k == nnz ? rows : row[k];
        for (int r = lo; r <= hi; ++r)
            row_ptr[r] = (int32_t) k;
    }
}

// Hybrid ELLPACK + COO -> one row-major matrix: row r = its ELL entries (row_length of them, padding
// included: a padded entry multiplies 0.0 by x like in the reference) followed by its COO entries.
// One thread per row (upload time only).
static __global__ __launch_bounds__(256) void hybrid_merge_kernel(
    int rows, int row_length, const int32_t * __restrict__ ell_col, const double * __restrict__ ell_val,
    const int32_t * __restrict__ coo_ptr, const int32_t * __restrict__ coo_col, const double * __restrict__ coo_val,
    int32_t * __restrict__ out_col, double * __restrict__ out_val)
{
    const long long r = (long long) blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    long long dst = r * row_length + coo_ptr[r];
    for (long long k = r * row_length; k < (r + 1) * row_length; ++k, ++dst) {
        out_col[dst] = ell_col[k];
        out_val[dst] = ell_val[k];
    }
    for (int k = coo_ptr[r]; k < coo_ptr[r + 1]; ++k, ++dst) {
        out_col[dst] = coo_col[k];
        out_val[dst] = coo_val[k];
    }
}

// Value dictionary, plan time.  value_dict_insert_kernel: every distinct bit pattern among the n values goes
// into an open-addressing table of kDictSlots 64-bit keys (kDictEmpty = free); state[0] counts the distinct
// values, state[1] is raised when there are more than `limit` (or a value equals the free marker) and
// everybody stops.  Almost every probe ends on its first load: a matrix that qualifies has few values.
constexpr int kDictSlots = 1024;
constexpr unsigned long long kDictEmpty = 0x7FF8DEADBEEF0001ull; // a NaN payload nobody stores

static __global__ __launch_bounds__(256) void value_dict_insert_kernel(
    long long n, const double * __restrict__ a, unsigned long long * __restrict__ keys, int * __restrict__ state, int limit)
{
    const long long stride = (long long) gridDim.x * 256;
    int round = 0;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride, ++round) {
        // a matrix with more distinct values than the dictionary holds is found out within the first few thousand
        // entries: every thread looks at the verdict every eighth round and leaves
        if ((round & 7) == 0 && __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return;
        const unsigned long long key = (unsigned long long) __double_as_longlong(a[k]);
        if (key == kDictEmpty) {
            atomicOr(state + 1, 1);
            return;
        }
        unsigned long long h = key * 0x9E3779B97F4A7C15ull;
        unsigned slot = (unsigned) (h >> 54) & (kDictSlots - 1);
        for (int probe = 0; probe < kDictSlots; ++probe, slot = (slot + 1) & (kDictSlots - 1)) {
            unsigned long long cur = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == key)
                break;
            if (cur == kDictEmpty) {
                cur = atomicCAS(keys + slot, kDictEmpty, key);
                if (cur == kDictEmpty) {
                    if (atomicAdd(state, 1) + 1 > limit)
                        atomicOr(state + 1, 1);
                    break;
                }
                if (cur == key)
                    break;
            }
        }
    }
}

// value_index_kernel: idx[k] = position of a[k] in the dictionary `table` (nvalues bit patterns, ascending as
// unsigned 64-bit integers); state[1] is raised if a value is not in it (the array changed under the plan).
static __global__ __launch_bounds__(256) void value_index_kernel(
    long long n, const double * __restrict__ a, const unsigned long long * __restrict__ table, int nvalues,
    uint8_t * __restrict__ idx, int * __restrict__ state)
{
    __shared__ unsigned long long t[kMaxIndexedValues];
    if (threadIdx.x < kMaxIndexedValues)
        t[threadIdx.x] = threadIdx.x < (unsigned) nvalues ? table[threadIdx.x] : ~0ull;
    __syncthreads();
    const long long stride = (long long) gridDim.x * 256;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const unsigned long long key = (unsigned long long) __double_as_longlong(a[k]);
        int lo = 0, hi = nvalues - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (t[lo] != key)
            atomicOr(state + 1, 1);
        idx[k] = (uint8_t) lo;
    }
}

// value_rows_mark_kernel: one wave per tile.  A stencil tile (fast, uniform, shifted, rows of at most max_len entries)
// whose rows all repeat the first row's index bytes gets
// kTileMetaValueRows; every other tile loses the bit.  count[0] = tiles marked, count[1] = their entries.
// same_prev[w] (if asked for) = 1 when tile w's first row carries exactly the index bytes of tile w - 1's first `len`
// entries: two constant-row tiles may only be merged into one (merge_constant_row_tiles: the re-cut tile reads ITS first
// row's bytes for all its rows) when their coefficient sets are the same -- a piecewise-constant stencil whose coefficients
// jump exactly on a tile boundary has two constant tiles with different sets next to each other.  Compared byte by byte,
// not hashed: a wrong "same" would be a wrong y.
static __global__ __launch_bounds__(256) void value_rows_mark_kernel(
    int ntiles, int4 * __restrict__ desc, const uint8_t * __restrict__ idx, int max_len, unsigned long long * __restrict__ count,
    uint8_t * __restrict__ same_prev)
{
    const int w = (int) blockIdx.x * 4 + (int) (threadIdx.x >> 6);
    const int lane = (int) (threadIdx.x & 63);
    if (w >= ntiles)
        return;
    const int4 d0 = desc[w];
    const int k0 = d0.y, k1 = desc[w + 1].y;
    const int meta = d0.z;
    const int len = meta & 0xFFFF;
    const bool candidate = !(d0.x & kTileFlagPartial) && (meta & kTileMetaFast) && (meta & kTileMetaUniform) && (meta & kTileMetaShifted)
        && len >= 1 && len <= max_len && k1 > k0;
    bool same = candidate;
    if (candidate) {
        int pos = lane % len; // position of entry k0 + lane within its row
        const int step = kWave % len;
        for (int k = k0 + lane; k < k1; k += kWave) {
            same = same && idx[k] == idx[k0 + pos];
            pos += step;
            pos -= pos >= len ? len : 0;
        }
        same = __all(same);
    }
    if (same_prev) {
        // (only .y of the neighbour's descriptor is read: .z is being rewritten by the neighbour's wave)
        bool eq = candidate && w > 0;
        if (eq) {
            const int kp = desc[w - 1].y;
            eq = k0 - kp >= len; // the neighbour holds at least one row of this length (the host checks that it is such a tile)
            for (int pos = lane; eq && pos < len; pos += kWave)
                eq = idx[k0 + pos] == idx[kp + pos];
            eq = __all(eq);
        }
        if (lane == 0)
            same_prev[w] = eq ? 1 : 0;
    }
    if (lane == 0) {
        const int now = same ? (meta | kTileMetaValueRows) : (meta & ~kTileMetaValueRows);
        if (now != meta)
            desc[w].z = now;
        if (same) {
            striped_add(count, 0, 1ull);
            striped_add(count, 1, (unsigned long long) (k1 - k0));
        }
    }
}

static __global__ __launch_bounds__(256) void value_checksum_kernel(
    long long n, const double * __restrict__ a, unsigned long long * __restrict__ out)
{
    const long long stride = (long long) gridDim.x * 256;
    unsigned long long h = 0;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        unsigned long long t = (unsigned long long) k * 0x9E3779B97F4A7C15ull ^ (unsigned long long) __double_as_longlong(a[k]);
        t ^= t >> 29;
        t *= 0xBF58476D1CE4E5B9ull;
        t ^= t >> 32;
        h += t;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1)
        h += __shfl_xor(h, d);
    if ((int) __lane_id() == 0)
        atomicAdd(out, h);
}

static __global__ __launch_bounds__(256) void column_checksum_kernel(
    long long n, const int32_t * __restrict__ j, unsigned long long * __restrict__ out)
{
    const long long stride = (long long) gridDim.x * 256;
    unsigned long long h = 0;
    for (long long k = (long long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        unsigned long long t = (unsigned long long) k * 0x9E3779B97F4A7C15ull + (unsigned long long) (unsigned) j[k];
        t ^= t >> 29;
        t *= 0xBF58476D1CE4E5B9ull;
        t ^= t >> 32;
        h += t;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1)
        h += __shfl_xor(h, d);
    if ((int) __lane_id() == 0)
        atomicAdd(out, h);
}

} // namespace spmv
