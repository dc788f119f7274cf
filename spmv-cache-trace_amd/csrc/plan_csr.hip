// plan_csr.hip -- CSR launch plans (Level 2 of include/spmv_hip.h): wave tiles from the host row_ptr, tile classes
// from the device columns (16-bit columns, shifted tiles, patterns, x windows, block / segment windows), column
// panels, the value dictionary, and the content guards that tie a plan to the arrays it was derived from.
#include "internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <utility>

using namespace spmvi;

namespace {

// Plan-time counters live in kCountStripes copies on the device (tile_common.hpp: striped_add); these read them back summed.
constexpr size_t kStripedInts = (size_t) spmv::kCountStripes * spmv::kCountWidth;

template <typename T>
hipError_t read_striped(const T * d_counters, T * out, int n, hipStream_t s)
{
    T host[kStripedInts];
    hipError_t e = hipMemcpyAsync(host, d_counters, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    for (int i = 0; i < n; ++i) {
        out[i] = 0;
        for (int k = 0; k < spmv::kCountStripes; ++k)
            out[i] += host[(size_t) k * spmv::kCountWidth + i];
    }
    return e;
}

} // namespace

namespace spmvi {

// flags[0]: an index outside [0, limit); flags[1] (if asked): not non-decreasing
int device_index_check(const int32_t * d_idx, long long n, int limit, bool want_sorted, bool * bad, bool * sorted, hipStream_t s)
{
    *bad = false;
    if (sorted)
        *sorted = true;
    if (n <= 0)
        return SPMV_HIP_OK;
    int * d_flags = nullptr;
    int flags[2] = {0, 0};
    HIP_TRY(hipMalloc((void **) &d_flags, sizeof(flags)));
    hipError_t e = hipMemsetAsync(d_flags, 0, sizeof(flags), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::index_check_kernel, dim3((unsigned) grid_for(n, kBlock, cu_count() * 16)), dim3(256), 0, s, n, limit,
                           d_idx, d_flags, want_sorted ? 1 : 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(d_flags);
    if (e != hipSuccess)
        return fail_hip(e, "index check");
    *bad = flags[0] != 0;
    if (sorted)
        *sorted = flags[1] == 0;
    return SPMV_HIP_OK;
}

int device_column_checksum(const int32_t * d_col, long long n, unsigned long long * out, hipStream_t s)
{
    *out = 0;
    if (n <= 0)
        return SPMV_HIP_OK;
    unsigned long long * d_sum = nullptr;
    HIP_TRY(hipMalloc((void **) &d_sum, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::column_checksum_kernel, dim3((unsigned) grid_for(n, kBlock, cu_count() * 16)), dim3(256), 0, s, n, d_col, d_sum);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_sum, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(d_sum);
    return e == hipSuccess ? SPMV_HIP_OK : fail_hip(e, "column checksum");
}

int device_value_checksum(const double * d_val, long long n, unsigned long long * out, hipStream_t s)
{
    *out = 0;
    if (n <= 0)
        return SPMV_HIP_OK;
    unsigned long long * d_sum = nullptr;
    HIP_TRY(hipMalloc((void **) &d_sum, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::value_checksum_kernel, dim3((unsigned) grid_for(n, kBlock, cu_count() * 16)), dim3(256), 0, s, n, d_val, d_sum);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_sum, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(d_sum);
    return e == hipSuccess ? SPMV_HIP_OK : fail_hip(e, "value checksum");
}

// the value dictionary belongs to ONE value array with ONE content: same rule as for the columns below
int verify_plan_values(const spmv_hip_plan * pl, const double * d_value, hipStream_t s)
{
    if (pl->nvalues == 0 || pl->values_from != d_value)
        return SPMV_HIP_OK; // another array: its values are read as they are
    unsigned long long sum = 0;
    int rc = device_value_checksum(d_value, pl->nnz, &sum, s);
    if (rc != SPMV_HIP_OK)
        return rc;
    if (sum != pl->value_checksum)
        return fail(SPMV_HIP_ERR_STATE, "the value array changed since spmv_hip_plan_csr_index_values: call "
                                        "spmv_hip_plan_csr_refresh_values after changing values");
    return SPMV_HIP_OK;
}

// The plan's derived data (16-bit column stream, tile marks, patterns) belong to ONE column array.
// Pointer identity alone cannot tell a new matrix that an allocator placed at the old address, so
// the contents are checked: on the first multiply after compress, on every multiply with
// SPMV_HIP_FLAG_VERIFY_PLAN, and on demand (spmv_hip_plan_verify).
int verify_plan(const spmv_hip_plan * pl, const int32_t * d_column_index, hipStream_t s)
{
    if (!pl->d_col16 || pl->compressed_from != d_column_index)
        return SPMV_HIP_OK; // another array: the plan falls back to its 32-bit path, nothing derived is used
    unsigned long long sum = 0;
    int rc = device_column_checksum(d_column_index, pl->nnz, &sum, s);
    if (rc != SPMV_HIP_OK)
        return rc;
    if (sum != pl->column_checksum)
        return fail(SPMV_HIP_ERR_STATE, "the column array at this address is not the one the plan was compressed from "
                                        "(contents changed): make a new plan");
    return SPMV_HIP_OK;
}

// Bytes one multiply streams with the tile classes chosen (bookkeeping for the roofline report):
// values 8 B per entry; columns 4 B (wide), 2 B (16-bit), one first row (shifted) or nothing
// (shifted with a pattern); row_ptr 4 B per row of a non-uniform tile; y 16 B per row; x once;
// 16 B of descriptor per tile.
int plan_account(spmv_hip_plan * pl, bool compressed)
{
    const long long algorithmic = 12LL * pl->nnz + 4LL * (pl->rows + 1LL) + 16LL * pl->rows + 8LL * pl->cols;
    pl->streamed_bytes = algorithmic;
    pl->shifted_entries = pl->narrow_entries = pl->uniform_rows = 0;
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->ntiles == 0)
        return SPMV_HIP_OK;
    // (the dictionary launch may have its own descriptors: runs of constant-row tiles re-cut into tiles of 128 rows)
    const bool vi_tiles = pl->nvalues > 0 && pl->d_tiles_vi;
    const int ntiles = vi_tiles ? pl->ntiles_vi : pl->ntiles;
    std::vector<int4> d((size_t) ntiles + 1);
    HIP_TRY(hipMemcpy(d.data(), vi_tiles ? pl->d_tiles_vi : pl->d_tiles, d.size() * sizeof(int4), hipMemcpyDeviceToHost));
    long long bytes = 8LL * pl->cols;
    for (int w = 0; w < ntiles; ++w) {
        const long long entries = (long long) d[(size_t) w + 1].y - d[(size_t) w].y;
        const long long rows = (long long) (d[(size_t) w + 1].x & 0x7FFFFFFF) - (d[(size_t) w].x & 0x7FFFFFFF);
        const int meta = d[(size_t) w].z;
        const bool stream_tile = !(d[(size_t) w].x & 0x80000000) && entries > 0 && (meta & spmv::kTileMetaFast);
        const bool shifted = compressed && stream_tile && (meta & spmv::kTileMetaShifted);
        const bool narrow = compressed && stream_tile && (meta & spmv::kTileMetaNarrow);
        const bool uniform = stream_tile && (meta & spmv::kTileMetaUniform);
        const bool block3 = compressed && stream_tile && (meta & spmv::kTileMetaBlock3) && !(meta & spmv::kTileMetaBlockWin) && pl->nvalues == 0
            && !(pl->flags & SPMV_HIP_FLAG_EXACT_ORDER);
        if (compressed && stream_tile && spmv::is_masked_stencil_tile(meta)) { // no columns, no row_ptr: a 16-bit mask per row
            pl->shifted_entries += entries;
            bytes += (pl->nvalues > 0 ? 1 : 8) * entries + 2 * rows + 16 + 16 * rows; // (a dictionary launch reads an index byte per entry)
            continue;
        }
        if (block3 && (meta & spmv::kTileMetaGroupRows)) { // group tile: one 16-bit column per column of a group of block_hint rows; row_ptr read
            pl->narrow_entries += entries;
            // (a wide group tile's columns are 32-bit ones)
            bytes += 8 * entries + ((meta & spmv::kTileMetaGroupPairs) ? 1 : 2) * ((meta & spmv::kTileMetaNarrow) ? 1 : 2) * (entries / std::max(1, pl->block_hint)) + 16
                + 16 * rows + (uniform ? 0 : 4 * (rows + 1));
            if (uniform)
                pl->uniform_rows += rows;
            continue;
        }
        if (block3) { // one 16-bit number per 3 x 3 block (masked block tiles: a 32-bit word per block, about one per 8 entries), no row_ptr
            pl->narrow_entries += entries;
            bytes += 8 * entries + ((meta & spmv::kTileMetaBlock3Masked) ? 4 * ((entries + 7) / 8) : 2 * (entries / 9)) + 16 + 16 * rows;
            continue;
        }
        long long col_bytes = 4 * entries;
        if (shifted) {
            col_bytes = (meta & spmv::kTileMetaPattern) ? 0 : 4LL * (meta & 0xFFFF);
            pl->shifted_entries += entries;
        } else if (narrow || (compressed && stream_tile && (meta & spmv::kTileMetaBlockWin))) {
            col_bytes = 2 * entries; // 16-bit offsets from the tile's base, or window slots (segment windows: any column range)
            pl->narrow_entries += entries;
        } else if (compressed && !(meta & spmv::kTileMetaFast) && (meta & spmv::kTileMetaNarrow) && entries > 0
                   && ((d[(size_t) w].x & 0x80000000) || (long long) d[(size_t) w + 1].y - (d[(size_t) w].y & ~3) > pl->tile)) {
            col_bytes = 2 * entries; // a long row (or a chunk of one) whose columns span less than 65536: long_row_sum reads the 16-bit stream
            pl->narrow_entries += entries;
        }
        if (uniform)
            pl->uniform_rows += rows;
        // with a value dictionary the stream tiles of the default kernel read one byte per entry
        const int len = meta & 0xFFFF;
        const bool value_rows = pl->nvalues > 0 && stream_tile && (meta & spmv::kTileMetaValueRows) && shifted && uniform // as the kernel decides
            && (len <= spmv::kLanePerRowMaxLen ? (((meta >> spmv::kTileMetaLanesShift) & 7) == 0 && rows >= 2) : rows >= spmv::kConstantRowMinRows);
        const long long val_bytes = value_rows ? (meta & 0xFFFF) // only the first row's index bytes are read
            : (pl->nvalues > 0 && stream_tile) ? entries : 8 * entries;
        bytes += val_bytes + col_bytes + 16 + 16 * rows + (uniform ? 0 : 4 * (rows + 1));
    }
    if (pl->nhubs > 0) // the dense copy: hub columns read, their x entries read and written, then read by the tiles (once)
        bytes += 20LL * pl->nhubs;
    pl->streamed_bytes = bytes;
    // plans with block or segment windows: the list of tiles left to csr_wavetile_kernel (made once, right after marking)
    if (!vi_tiles && compressed && (pl->d_blocks || pl->d_segblocks) && !pl->d_rest_tiles && pl->blockwin_tiles < pl->ntiles) {
        std::vector<int32_t> rest;
        rest.reserve((size_t) (pl->ntiles - pl->blockwin_tiles));
        for (int w = 0; w < pl->ntiles; ++w)
            if (!(d[(size_t) w].z & spmv::kTileMetaBlockWin))
                rest.push_back(w);
        rest.resize((rest.size() + 3) & ~(size_t) 3, rest.empty() ? 0 : rest.back()); // padded to whole workgroups (never read past nrest)
        if (!rest.empty()) {
            HIP_TRY(hipMalloc((void **) &pl->d_rest_tiles, rest.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(pl->d_rest_tiles, rest.data(), rest.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            pl->meta_bytes += rest.size() * sizeof(int32_t);
        }
        pl->nrest_tiles = 0;
        for (int w = 0; w < pl->ntiles; ++w)
            pl->nrest_tiles += !(d[(size_t) w].z & spmv::kTileMetaBlockWin);
    }
    // row-group plans (csr_rowgroup.hpp, opt-in): made anew after every change of the tile marks
    for (int32_t ** q : {&pl->d_group_tiles, &pl->d_group_rest})
        if (*q) {
            (void) hipFree(*q);
            *q = nullptr;
        }
    pl->ngroup_tiles = pl->ngroup_rest = 0;
#ifdef SPMV_HIP_EXPERIMENTS
    if ((pl->flags & SPMV_HIP_FLAG_ROW_GROUPS) && compressed && pl->nvalues == 0 && !pl->d_blocks && !pl->d_segblocks && pl->tile == 512 && !pl->balanced && pl->cols < (1 << 29)
        && !(pl->flags & (SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_NO_X_WINDOW | SPMV_HIP_FLAG_XCD_REMAP))) {
        std::vector<int32_t> group, rest;
        for (int w = 0; w < pl->ntiles; ++w) {
            const long long k1 = d[(size_t) w + 1].y, entries = k1 - d[(size_t) w].y;
            const long long rows = (long long) (d[(size_t) w + 1].x & 0x7FFFFFFF) - (d[(size_t) w].x & 0x7FFFFFFF);
            (spmv::rowgroup_tile(d[(size_t) w].x, d[(size_t) w].z, entries, rows, k1, pl->nnz) ? group : rest).push_back(w);
        }
        if (2 * group.size() > (size_t) pl->ntiles) {
            pl->ngroup_tiles = (int) group.size();
            pl->ngroup_rest = (int) rest.size();
            HIP_TRY(hipMalloc((void **) &pl->d_group_tiles, group.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(pl->d_group_tiles, group.data(), group.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            if (!rest.empty()) {
                HIP_TRY(hipMalloc((void **) &pl->d_group_rest, rest.size() * sizeof(int32_t)));
                HIP_TRY(hipMemcpy(pl->d_group_rest, rest.data(), rest.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            }
        }
    }
#endif
    return SPMV_HIP_OK;
}

} // namespace spmvi

namespace {

int pick_lanes(double mean_len)
{
    int l = 2;
    while (l < 64 && l < mean_len)
        l *= 2;
    return l;
}

} // namespace

extern "C" {

/* ================================ Level 2 ======================================= */

int spmv_hip_plan_csr(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p,
                      int algorithm, int lanes_per_row, unsigned flags)
{
    return plan_csr_internal(out, rows, cols, p, algorithm, lanes_per_row, flags, 0);
}

} // extern "C"

// Wave tiles of a plan from the HOST row_ptr: <= 64 (128) rows and <= tile entries (from the 4-aligned start) per wave.  Used by
// plan_csr_internal and, once more, by spmv_hip_plan_csr_repack when the block hint taken here turned out to be wrong.

// Runs `work` on up to n of the host's threads, the caller being one of them.  A thread that cannot be started (std::system_error
// under a container's pid limit, std::bad_alloc) is simply not there: both users hand their items out through an atomic counter,
// so whoever runs -- at worst the caller alone -- does all of them, and nothing is thrown across the C boundary for it.
template <class Work>
static void run_on_host_threads(int n, Work && work)
{
    std::vector<std::thread> pool;
    try {
        pool.reserve((size_t) std::max(0, n - 1));
        for (int t = 1; t < n; ++t)
            pool.emplace_back(work);
    } catch (std::system_error const &) {
    } catch (std::bad_alloc const &) {
    }
    try {
        work();
    } catch (...) {
        for (auto & th : pool)
            th.join();
        throw;
    }
    for (auto & th : pool)
        th.join();
}

// (7 entries per row: 73 rows per tile, the second pass 14 % full.  3-D 7-point Laplacian on a 256^3 grid, same process pair:
// 278.0 / 278.9 -> 255.9 / 252.8 us (0.84 -> 0.92 of the roofline) with 64-row tiles, which also sit four to a grid line there;
// 7 diagonals through the x window 169.0 -> 167.5 us: profiles/r05_results.md)
constexpr int kSecondPassMinRows = 80;

static int build_wave_tiles(spmv_hip_plan * pl, const int32_t * p, unsigned flags, int32_t break_rows, int split_threshold, int split_chunk)
{
    const int32_t rows = pl->rows;
    // wave tiles: <= 64 rows and <= tile entries (from the 4-aligned start) per wave
    const int tile = (flags & SPMV_HIP_FLAG_BIG_TILE) ? 1024 : 512;
    const bool exact = (flags & SPMV_HIP_FLAG_EXACT_ORDER) != 0;
    pl->tile = tile;
    // A row longer than a tile is walked by ONE wave in registers (long_row_sum, tile_common.hpp: 512 entries per step).  It is cut
    // into chunks -- one wave each, meeting in one fp64 atomic per chunk -- only as far as a wave walking it alone would be the
    // tail of the launch: a chunk is 1/8192 of the matrix (the chip runs 8192 waves side by side), at least split_chunk entries.
    // A web graph of 3 M entries keeps its chunks of 512; the 4096-entry rows of a 33 M-entry ELLPACK matrix stay whole (no
    // atomics, the same y on every run), as do the rows of any matrix with more than 8192 equally long rows.
    {
        const long long per_wave = (((long long) p[rows] / 8192) + 511) & ~511LL;
        if (per_wave > split_chunk) {
            split_chunk = (int) std::min<long long>(per_wave, 1LL << 24);
            split_threshold = std::max(split_threshold, split_chunk);
        }
    }
    // Block hint (csr_blocktile.hpp): rows in triples of equal length, divisible by 3 and longer than 16 entries -- three
    // unknowns per mesh node.  Only a hint: tiles are then cut on triple boundaries, and spmv_hip_plan_csr_repack checks the
    // columns of every tile before it marks it.
    // (The triples need not start at row 0: a rank's row block of a partitioned matrix starts wherever ceil(rows / G) puts it, so
    // the three possible offsets are tried and the tiles cut on THAT grid.)
    // Real files are less tidy: explicit zeros dropped from some blocks make the rows of a triple differ in length, nothing
    // obliges a row length to be divisible by 3, and a node with one or two unknowns moves the grid of triples behind it.  A
    // matrix most of whose rows are longer than 16 entries and about as long as the row behind them is therefore remembered as
    // a CANDIDATE: its tiles are cut without the hint, and spmv_hip_plan_csr_repack -- which sees the columns -- works out
    // which rows belong to the same node (csr_row_group_kernel); only if half of the rows then stand in groups of three does
    // it cut the tiles once more, on the boundaries of those groups (pl->group_bits), for the masked block tiles of
    // csr_blocktile.hpp.  A matrix that merely has rows of similar length pays for that one pass over row_ptr, nothing else.
    auto similar_triple = [p, rows](int32_t q) {
        if (q < 0 || q + 3 > rows)
            return false;
        const int l0 = p[q + 1] - p[q], l1 = p[q + 2] - p[q + 1], l2 = p[q + 3] - p[q + 2];
        const int lo = std::min(l0, std::min(l1, l2)), hi = std::max(l0, std::max(l1, l2));
        return lo > 16 && hi - lo <= hi / 4;
    };
    const uint32_t * const gb = pl->group_bits; // bit r: row r begins a group of rows with the same columns (repack's second cut only)
    auto group_start = [gb, rows](int32_t q) { return q >= rows || ((gb[q >> 5] >> (q & 31)) & 1u) != 0; };
    auto triple_at = [&](int32_t q) { return q + 3 <= rows && group_start(q) && !group_start(q + 1) && !group_start(q + 2) && group_start(q + 3); };
    // ... and the STORED TRIANGLE of such a matrix (round 6) -- what a `symmetric` Matrix Market file holds and the reference multiplies
    // as it stands (src/matrix/matrix-market.cpp:530-555): the rows of a node are (m + 1, m + 2, m + 3) entries long with m a multiple
    // of 3 (lower triangle: m / 3 whole blocks and the node's own triangular one; upper triangle: the lengths run the other way).
    // Row lengths then grow with the number of neighbours numbered in front of a node -- in an unstructured mesh anything from
    // 1 to ~100 -- which a plain tile takes badly (its longest row fixes the lanes per row: 39 % full, and the plan fell back to
    // balanced tiles at 0.57 of the roofline, profiles/r06_structure_zoo.log) and a block tile does not mind (a lane per block).
    auto skewed_triple = [p, rows](int32_t q) {
        if (q < 0 || q + 3 > rows)
            return false;
        const int l0 = p[q + 1] - p[q], l1 = p[q + 2] - p[q + 1], l2 = p[q + 3] - p[q + 2];
        return (l0 % 3 == 1 && l1 == l0 + 1 && l2 == l0 + 2) || (l2 % 3 == 1 && l1 == l2 + 1 && l0 == l2 + 2);
    };
    pl->block_candidate = 0;
    pl->block_skewed = false;
    pl->hint_from_bits = gb != nullptr;
    if (gb) {
        pl->block_hint = 3;
        pl->block_offset = 0;
    } else {
        pl->block_hint = 0;
        pl->block_offset = 0;
        if (!exact && tile == 512 && break_rows == 0 && rows >= 192 && !(flags & (SPMV_HIP_FLAG_NO_BLOCK_TILES | SPMV_HIP_FLAG_NO_MASKED_BLOCKS))
            && !(pl->hints_tried & 4)) {
            // Real files drop explicit zeros: with 0.5 % of the entries gone only two thirds of the triples still go EXACTLY
            // (m + 1, m + 2, m + 3), with 2 % a fifth (the Delaunay stored triangle fell from 0.99 to 0.67 with 0.5 % dropped:
            // balanced tiles).  What survives is the PEAK: at the right offset a tenth of the triples or more are exact, at the two
            // wrong ones next to none (an exact triple wants 3 | m at that very offset), and nearly all triples are LOOSE ones --
            // each row -2 ... +4 entries longer than the one above.  A glance (1024 triples per offset, spread over the matrix)
            // finds the offset, one pass over row_ptr at that offset confirms; a matrix without the peak -- scalar meshes: 0.7 % of
            // the triples exact at every offset, stencils: none -- pays for the glance only.
            auto loose = [p](int32_t q) {
                const int l0 = p[q + 1] - p[q], d1 = (p[q + 2] - p[q + 1]) - l0, d2 = (p[q + 3] - p[q + 2]) - (p[q + 2] - p[q + 1]);
                return l0 >= 1 && d1 >= -2 && d1 <= 4 && d2 >= -2 && d2 <= 4;
            };
            int glance[3] = {0, 0, 0};
            const long long triples_all = (rows - 2) / 3;
            for (int o = 0; o < 3; ++o)
                for (int t = 0; t < 1024; ++t) {
                    const int32_t q = o + 3 * (int32_t) ((triples_all - 1) * t / 1023);
                    glance[o] += skewed_triple(q) ? 1 : 0;
                }
            const int o = glance[0] >= glance[1] && glance[0] >= glance[2] ? 0 : (glance[1] >= glance[2] ? 1 : 2);
            const int other = std::max(glance[(o + 1) % 3], glance[(o + 2) % 3]);
            if (glance[o] >= 80 && glance[o] >= 4 * other) {
                const long long triples = (rows - o) / 3;
                long long exact3 = 0, near3 = 0, entries = 0;
                for (int32_t q = o; q + 2 < rows; q += 3) {
                    const bool e3 = skewed_triple(q), n3 = e3 || loose(q);
                    exact3 += e3;
                    near3 += n3;
                    entries += n3 ? p[q + 3] - p[q] : 0;
                }
                // (on average more than two blocks per block row: a tridiagonal matrix, whose interior rows "go" 2, 3, 3, is not one)
                if (exact3 * 10 >= triples && near3 * 5 >= triples * 4 && entries >= 24 * near3) {
                    pl->block_hint = 3;
                    pl->block_offset = o;
                    pl->block_skewed = true;
                }
            }
        }
        if (!pl->block_hint && !exact && tile == 512 && break_rows == 0 && rows >= 192 && !(flags & SPMV_HIP_FLAG_NO_BLOCK_TILES)) {
            // (pl->hints_tried: a hint that repack found wrong is not taken again when the tiles are cut anew -- the next one gets its turn)
            for (int o = 0; o < 3 && !pl->block_hint && !(pl->hints_tried & 1); ++o) {
                // (an offset is given up as soon as a fifth of all triples have failed: a matrix without blocks -- most -- pays for a
                // fifth of one pass per offset, not for three passes over row_ptr)
                const long long triples = (rows - o) / 3, allowed_bad = triples / 5;
                long long good = 0, bad = 0;
                for (int32_t q = o; q + 2 < rows && bad <= allowed_bad; q += 3) {
                    const int l0 = p[q + 1] - p[q];
                    const bool ok = l0 > 16 && l0 % 3 == 0 && p[q + 2] - p[q + 1] == l0 && p[q + 3] - p[q + 2] == l0;
                    good += ok;
                    bad += !ok;
                }
                if (bad <= allowed_bad && good * 5 >= triples * 4) {
                    pl->block_hint = 3;
                    pl->block_offset = o;
                }
            }
            // Rows in groups of 4 or of 2 equally long rows (a mesh with 4 or 2 unknowns per node; 8 or 10: groups of 4 / 2 as well):
            // no 3 x 3 blocks, but the rows of a group may share their column list -- group tiles, csr_blocktile.hpp.  The same
            // strict rule as for triples (four fifths of the groups: rows longer than 16 entries, all equally long, the length a
            // multiple of the group size, a whole group no longer than a tile), tried for 4 before 2; tiles are then cut on multiples of
            // that many rows.
            for (int d = 4; d >= 2 && !pl->block_hint && !(pl->hints_tried & 2); d -= 2)
                for (int o = 0; o < d && !pl->block_hint; ++o) {
                    const long long groups = (rows - o) / d, allowed_bad = groups / 5;
                    long long good = 0, bad = 0;
                    // (a glance first -- 512 groups spread over the matrix -- so that a matrix without groups costs next to nothing)
                    int glance = 0;
                    for (int t = 0; t < 512; ++t) {
                        const int32_t q = o + (int32_t) ((groups - 1) * t / 511) * d;
                        const int l0 = p[q + 1] - p[q];
                        bool ok = l0 > 16 && l0 % d == 0 && (long long) l0 * d <= tile; // (a group must fit a tile)
                        for (int a = 1; a < d && ok; ++a)
                            ok = p[q + a + 1] - p[q + a] == l0;
                        glance += ok;
                    }
                    if (glance < 384)
                        continue;
                    for (int32_t q = o; q + d <= rows && bad <= allowed_bad; q += d) {
                        const int l0 = p[q + 1] - p[q];
                        bool ok = l0 > 16 && l0 % d == 0 && (long long) l0 * d <= tile;
                        for (int a = 1; a < d && ok; ++a)
                            ok = p[q + a + 1] - p[q + a] == l0;
                        good += ok;
                        bad += !ok;
                    }
                    if (bad <= allowed_bad && good * 5 >= groups * 4) {
                        pl->block_hint = d;
                        pl->block_offset = o;
                    }
                }
            if (!pl->block_hint && !(flags & SPMV_HIP_FLAG_NO_MASKED_BLOCKS)) {
                // (given up as soon as two fifths of the rows have failed)
                const long long allowed_bad = 2LL * rows / 5;
                long long bad = 0;
                for (int32_t q = 0; q + 1 < rows && bad <= allowed_bad; ++q) {
                    const int l0 = p[q + 1] - p[q], l1 = p[q + 2] - p[q + 1];
                    const int lo = std::min(l0, l1), hi = std::max(l0, l1);
                    bad += !(lo > 16 && hi - lo <= hi / 4);
                }
                pl->block_candidate = bad <= allowed_bad ? 1 : 0;
            }
        }
    }
    // The rows are cut into tiles in up to kTileChunks independent ranges, side by side on the host's threads (a tile never
    // crosses a range boundary; the boundaries depend on the row count alone -- and, with a block hint, lie on the grid of
    // triples -- so the tiling is the same on every machine).  Round 4: this walk over row_ptr was 29 of the 35 ms a Poisson
    // 4096^2 plan costs.  Column panels (break_rows > 0) keep one range: their tiles are numbered per panel.
    struct Range {
        std::vector<int4> desc;
        long long stream_tiles = 0, stream_tile_entries = 0, multi_candidates = 0; // tiles of whole rows (not long-row tiles) and what they hold
        int uniform_tiles = 0, long_blocks = 0, split_rows = 0, longest_tile_row = 0, multi_window_tiles = 0, block_cuts = 0, longest = 0;
    };
    int next_panel = 0;
    auto tile_range = [&](int32_t cb, int32_t ce, bool allow_multi, Range & o) {
    std::vector<int4> & desc = o.desc;
    desc.reserve((size_t) (ce - cb) / 48 + 16);
    int32_t r = cb;
    // With repack's groups (gb): a tile of long rows starts on a triple and holds whole triples only; rows of other groups (a node
    // with one or two unknowns, a boundary row) get tiles of their own up to the next triple.  With the hint from row_ptr alone:
    // the grid of triples at pl->block_offset, followed locally where the row lengths show that it moved.
    int phase = pl->block_offset;
    while (r < ce) {
        int32_t row_limit = ce;
        bool triples_only = false;
        if (gb) {
            if (triple_at(r)) {
                triples_only = true;
                phase = r % 3;
            } else {
                int32_t q = r + 1;
                while (q < ce && q - r < 128 && !triple_at(q))
                    ++q;
                row_limit = q;
            }
        } else if (pl->block_hint == 3 && ((r - phase) % 3 + 3) % 3 == 0 && r + 8 <= ce
                   && !(pl->block_skewed ? (skewed_triple(r) || skewed_triple(r + 3)) : similar_triple(r))) {
            for (int d = 1; d <= 2; ++d)
                if (pl->block_skewed ? (skewed_triple(r + d) && skewed_triple(r + d + 3)) : (similar_triple(r + d) && similar_triple(r + d + 3))) {
                    phase = (r + d) % 3;
                    row_limit = r + d;
                    break;
                }
        }
        if (break_rows > 0) // the first tile of each panel (tiles never straddle a panel boundary)
            while (next_panel <= 8 && next_panel <= r / break_rows)
                pl->pinfo.first[next_panel++] = (int) desc.size();
        const int32_t kb = p[r] & ~3;
        int32_t r1 = r;
        int32_t maxlen = 0, minlen = INT32_MAX;
        // rows per tile: 128 (two short rows per lane, fuller quads) pays once the matrix streams from
        // HBM (twice the 256 MiB Infinity Cache); below that more, smaller tiles win (measured:
        // Poisson 4096^2 223 vs 238 us, half of it 103 vs 107 us, a quarter 48 vs 43 us, 2048^2 57 vs 50 us)
        const double footprint = 12.0 * (double) p[rows] + 20.0 * (double) rows;
        const int row_cap = (flags & SPMV_HIP_FLAG_ROWS64) ? 64
            : (flags & SPMV_HIP_FLAG_ROWS128) ? 128 : (footprint >= 512e6 ? 128 : 64);
        // lanes per row follow the tile's LONGEST row (<= 16 entries per lane), and a tile takes
        // only as many rows as the wave has lanes for: a 400-entry row among 63 short ones would
        // otherwise be summed by one lane while the others wait (power-law rows 3/row: 50 -> 44 us)
        int per_lane = 16;
#ifdef SPMV_HIP_EXPERIMENTS
        // (tools/ab.py, round 5: 8 entries per lane -- 4 lanes per row of 28 -- kkt-like 730.1 against 734.6 us, 27 diagonals 161.8 against
        // 163.0, queen-like without block tiles 768.7 against 624.6: profiles/r05_ab_misc.log; 16 stays)
        if (const char * v = std::getenv("SPMV_HIP_ENTRIES_PER_LANE")) per_lane = std::max(4, std::min(64, std::atoi(v)));
#endif
        auto lanes_for = [per_lane](int len) {
            int l = 0;
            while (l < 6 && (per_lane << l) < len)
                ++l;
            return l;
        };
        while (r1 < row_limit && (r1 - r) < row_cap && (long long) p[r1 + 1] - kb <= tile) {
            if (break_rows > 0 && r1 > r && r1 % break_rows == 0)
                break;
            if (triples_only && r1 > r && (r1 - r) % 3 == 0 && !triple_at(r1))
                break;
            const int len = p[r1 + 1] - p[r1];
            o.longest = std::max(o.longest, len);
            // A long run of equally long rows (the interior of a stencil line) starts its own tile: the rows in front of it
            // (a grid line's boundary rows) would make the run's first tile non-uniform and send it down the general path --
            // Poisson 4096^2: 2.5 % of the tiles, each holding its wave slot twice as long as a stencil tile.  "Long" = at
            // least four full tiles of such rows, so that a matrix cannot fall apart into small tiles; only for rows short
            // enough for the one-lane-per-row stencil path (with 27 entries per row a tile is 18 rows: cutting in front of every
            // run leaves a partly filled tile per grid line -- KKT-like 758 -> 750 us, but its twin without shifted rows
            // 927 -> 956 us: not done there).
            if (r1 > r && len != p[r1] - p[r1 - 1] && len > 0 && len <= spmv::kLanePerRowMaxLen) {
                const long long need = 4LL * std::max(1, std::min(row_cap, tile / len));
                if (r1 + need <= rows) {
                    bool run = true;
                    for (long long q = r1 + 1; q < r1 + need && run; ++q)
                        run = p[q + 1] - p[q] == len;
                    if (run)
                        break;
                }
            }
            if (!exact && r1 > r && !pl->block_skewed) {
                const int l = lanes_for(std::max(maxlen, len));
                if (l > 0 && ((r1 - r + 1) << l) > 64)
                    break;
            }
            // (stored-triangle plans: the tile is cut by the block tile's limits below -- 30 rows, a lane per block; should the
            // columns not bear the blocks out, the plain path adds its rows with the lanes the wave has: fewer than 16 entries per
            // lane would want, correct all the same)
            if (pl->block_skewed && r1 - r >= 64)
                break;
            maxlen = std::max(maxlen, len);
            minlen = std::min(minlen, len);
            ++r1;
        }
        // A tile of short rows with a handful of rows beyond the wave's 64 lanes: the lane-per-row paths then run every position twice
        // (rows 0 .. 63, rows 64 ...) for those few rows -- 7 entries per row: 73 rows, the second pass 14 % full.  Below
        // kSecondPassMinRows rows such a tile ends at 64 rows instead.
        {
            int second_min = kSecondPassMinRows;
#ifdef SPMV_HIP_EXPERIMENTS
            if (const char * v = std::getenv("SPMV_HIP_SECOND_PASS_MIN_ROWS")) second_min = std::atoi(v); // tools/ab.py
#endif
            if (r1 - r > 64 && r1 - r < second_min && maxlen <= spmv::kLanePerRowMaxLen && break_rows == 0 && !triples_only) {
                r1 = r + 64;
                maxlen = 0;
                minlen = INT32_MAX;
                for (int32_t q = r; q < r1; ++q) {
                    maxlen = std::max(maxlen, p[q + 1] - p[q]);
                    minlen = std::min(minlen, p[q + 1] - p[q]);
                }
            }
        }
        // block hint: a tile of long rows ends on a triple boundary and holds at most kBlockTileMaxRows rows
        if (pl->block_hint && (maxlen > 16 || pl->block_skewed) && r1 > r + 1) {
            const int d = pl->block_hint; // 3, or the rows per group of a group-tile plan
            // (a stored-triangle plan's tile of short rows only -- no block tile -- still ends on the grid of triples, for its successor's sake)
            int32_t cut = maxlen > 16 ? std::min(r1, r + spmv::kBlockTileMaxRows - spmv::kBlockTileMaxRows % d) : r1;
            if (cut < ce)
                cut -= ((cut - phase) % d + d) % d;
            if (cut > r && cut < r1) {
                o.block_cuts++;
                r1 = cut;
                maxlen = 0;
                minlen = INT32_MAX;
                for (int32_t q = r; q < r1; ++q) {
                    maxlen = std::max(maxlen, p[q + 1] - p[q]);
                    minlen = std::min(minlen, p[q + 1] - p[q]);
                }
            }
        }
        // Multi-window tiles (csr_wavetile.hpp): rows of 129 ... 512 entries fill a 512-entry tile badly -- one row of 361 leaves
        // 30 % of the wave's load slots empty, two of 201 leave 21 % -- so a wave takes up to 8 such rows and walks them in
        // windows of 512 entries, carrying the row sums in registers: 7 rows of 361 are 4.94 windows (99 % full).  Chosen: the
        // row count (at most 8, all rows <= 512 entries, at most 8 windows) with the fullest windows, if that beats the plain
        // tile by a tenth.  Not in exact order (one lane per row), not for column panels, not at the ragged end of the arrays.
        int multi_lanes_log2 = -1;
        // (measured on ELLPACK bands, profiles/r04_ell_long_rows.md: 177 ... 441 per row 0.69-0.72 -> 0.76-0.84 of the roofline; rows
        // of up to 160 keep the plain tile -- three rows of 141 fill it to 83 % and have their x window, 0.93 against 0.90)
        // (... and rows of 513 ... 1024 entries are taken two to eight at a time the same way)
        // (round 5: rows of more than kMultiWindowMaxRow = 1024 entries are better off with a wave each, in registers -- bands of
        // 2001 per row 0.75 in multi-window tiles, 0.80 a wave per row; 1001 and 1501 per row the same either way; 601 and 801 per row
        // 0.75 against 0.66 / 0.71: profiles/r05_csr_long_rows.log)
        // (later in round 5: SEVERAL rows of more than 512 entries each per wave, in registers -- tile_rows_long_registers, one
        // butterfly per row and no LDS: the fill of a multi-window tile without its round trip.  ELLPACK bands of 601 / 801 / 1001
        // per row 0.76 / 0.77 / 0.81 -> 0.82 / 0.83 / 0.86.  Such a tile may hold rows of up to 16384 entries in up to 40 steps of
        // 512 (profiles/r05_long_pair_ab.log, r05_long_cap_ab.log) -- but never more than 1/8192 of the matrix, so that a small
        // matrix keeps its waves; rows of up to 1024 entries that the cap keeps out of the registers still share a tile through LDS)
        constexpr int kMultiWindowMaxRow = 1024, kRegisterTileMaxRow = 16384, kRegisterTileMaxSteps = 40;
        long long cap_div = 8192;
#ifdef SPMV_HIP_EXPERIMENTS
        if (const char * v = std::getenv("SPMV_HIP_REGISTER_TILE_CAP_DIV")) cap_div = std::max(1LL, std::atoll(v)); // tools/long_pair_ab.sh
#endif
        const long long register_tile_cap = std::max<long long>(2 * tile, (long long) p[rows] / cap_div);
        const bool multi_start = r1 > r ? maxlen > 160 : (long long) p[r + 1] - p[r] <= kRegisterTileMaxRow;
        // (under the block hint only rows that no block tile could hold: three rows of more than 170 entries exceed a tile)
        const bool hint_allows = !pl->block_hint || (r1 > r ? maxlen > 170 : true);
        if (!exact && tile == 512 && break_rows == 0 && hint_allows && multi_start && !(flags & SPMV_HIP_FLAG_NO_MULTI_WINDOW))
            ++o.multi_candidates;
        if (allow_multi && !exact && tile == 512 && break_rows == 0 && hint_allows && multi_start
            && !(flags & SPMV_HIP_FLAG_NO_MULTI_WINDOW)) {
            // what the rows would fill without this: the plain tile, or -- a row longer than a tile -- the steps of its own wave
            const long long alone = (long long) p[r + 1] - kb;
            const double plain = r1 > r ? (double) ((long long) p[r1] - kb) / tile : (double) alone / (double) (((alone + tile - 1) / tile) * tile);
            int best = 0;
            // (rows of 257: no row count up to 8 reaches 0.9 -- 7 rows are 0.88 -- and one row per tile is 0.50.  A plain tile of ONE
            // row is given up for any two rows that fill their windows to 3/4: a row of 479 has a tile 94 % full to itself and runs
            // at 0.67 of the roofline -- a descriptor, a y access and a 64-lane sum per row -- where two rows of 511 in two windows
            // run at 0.85, profiles/r04_ell_long_rows.md)
            // (a row of up to 1536 entries with steps to itself shares a tile whenever that tile is 3/4 full -- 1001 / 1501 per row:
            // 0.79 / 0.81 a wave each, 0.85 / 0.85 two per wave at the same fill; longer rows only where sharing fills the steps
            // better: in CSR, same process, 2001 / 3001 / 4001 per row ran 8 / 4 / 8 % SLOWER shared, profiles/r05_csr_shared_vs_alone.log)
            double alone_gain = (long long) p[r + 1] - p[r] <= 1536 ? -1.0 : 0.04;
#ifdef SPMV_HIP_EXPERIMENTS
            if (const char * v = std::getenv("SPMV_HIP_LONG_PAIR_GAIN")) alone_gain = std::atof(v); // tools/ell_long_rows.sh: -1 = share a tile whenever it is 3/4 full
#endif
            double best_fill = r1 - r >= 2 ? std::max(plain + 0.1, 0.75) : (r1 > r ? 0.75 : std::max(plain + alone_gain, 0.75));
            int mx = 0, mn = INT32_MAX;
            bool best_in_registers = false;
            for (int32_t q = r; q < row_limit && q - r < 8; ++q) {
                mx = std::max(mx, p[q + 1] - p[q]);
                mn = std::min(mn, p[q + 1] - p[q]);
                const long long e = (long long) p[q + 1] - kb;
                const long long windows = (e + tile - 1) / tile;
                // in registers: every row so far is longer than a step (at most one row ends per step); through LDS: round 4's limits
                const bool in_registers = mn > tile && mx <= kRegisterTileMaxRow && windows <= kRegisterTileMaxSteps && (q == r || e <= register_tile_cap);
                const bool through_lds = mx <= kMultiWindowMaxRow && windows <= 8;
                if (!in_registers && !through_lds)
                    break;
                const bool inside = ((e + kb - 1) | 3) < (long long) p[rows]; // the last quad's 16-byte loads stay inside the arrays
                const double fill = (double) e / (double) (windows * tile);
                if (q + 1 > r1 && q + 1 - r >= 2 && inside && fill > best_fill) {
                    best_fill = fill;
                    best = (int) (q + 1 - r);
                    best_in_registers = in_registers;
                }
            }
            if (best > 0) {
                r1 = r + best;
                maxlen = 0;
                minlen = INT32_MAX;
                for (int32_t q = r; q < r1; ++q) {
                    maxlen = std::max(maxlen, p[q + 1] - p[q]);
                    minlen = std::min(minlen, p[q + 1] - p[q]);
                }
                multi_lanes_log2 = best <= 2 ? 5 : (best <= 4 ? 4 : 3); // 64 / (rows rounded up to a power of two)
                if (best_in_registers)
                    multi_lanes_log2 = 6; // "all 64 lanes": the tile is walked in registers
                o.multi_window_tiles++;
            }
        }
#ifdef SPMV_HIP_EXPERIMENTS
        // tools/ab.py: tiles of short rows end on a multiple of SPMV_HIP_TILE_ROW_ALIGN rows (whole 128-byte lines of y per tile)
        if (const char * al = std::getenv("SPMV_HIP_TILE_ROW_ALIGN")) {
            const int align = std::atoi(al);
            if (align > 1 && r1 - r > 2 * align && r1 < rows && (r1 % align) != 0 && (r1 / align) * align > r) {
                r1 = (r1 / align) * align;
                maxlen = 0;
                minlen = INT32_MAX;
                for (int32_t q = r; q < r1; ++q) {
                    maxlen = std::max(maxlen, p[q + 1] - p[q]);
                    minlen = std::min(minlen, p[q + 1] - p[q]);
                }
            }
        }
#endif
        if (r1 == r) { // one row longer than a tile
            const long long len = (long long) p[r + 1] - p[r];
            o.longest = std::max(o.longest, (int) std::min<long long>(len, INT32_MAX));
            o.long_blocks++;
            if (!exact && len > split_threshold) {
                o.split_rows++;
                for (long long k = p[r]; k < p[r + 1]; k += split_chunk)
                    desc.push_back(make_int4((int) (r | 0x80000000u), (int) k, 0, 0));
            } else {
                desc.push_back(make_int4(r, p[r], 0, 0));
            }
            r1 = r + 1;
        } else {
            // one lane per row (rows of <= 16 entries) keeps the reference's summation order
            int lanes_log2 = multi_lanes_log2 >= 0 ? multi_lanes_log2 : (exact ? 0 : lanes_for(maxlen));
            while (multi_lanes_log2 < 0 && lanes_log2 > 0 && ((r1 - r) << lanes_log2) > 64)
                --lanes_log2; // (only in stored-triangle plans: more rows than the longest row's lanes leave room for)
            o.longest_tile_row = std::max(o.longest_tile_row, (int) maxlen);
            // "fast": non-empty, and 16-byte loads of its last quad stay inside the arrays
            const bool fast = p[r1] > p[r] && (((long long) p[r1] - 1) | 3) < (long long) p[rows];
            const bool uniform = minlen == maxlen && !(flags & SPMV_HIP_FLAG_READ_ROW_PTR);
            if (uniform)
                o.uniform_tiles++;
            o.stream_tiles++;
            o.stream_tile_entries += (long long) p[r1] - p[r];
            desc.push_back(make_int4(r, p[r], maxlen | (lanes_log2 << 16) | (fast ? (1 << 25) : 0) |
                                                  (uniform ? (1 << 26) : 0), 0));
        }
        r = r1;
    }
    };
    constexpr int kTileChunks = 64;
    const int nchunks = (break_rows > 0 || rows < (1 << 18)) ? 1 : kTileChunks;
    std::vector<int32_t> bound((size_t) nchunks + 1);
    for (int c = 0; c <= nchunks; ++c) {
        long long bnd = (long long) rows * c / nchunks;
        if (pl->block_hint && c > 0 && c < nchunks)
            bnd -= ((bnd - pl->block_offset) % pl->block_hint + pl->block_hint) % pl->block_hint;
        bound[(size_t) c] = (int32_t) bnd;
    }
    std::vector<int4> desc;
    long long stream_tiles = 0, stream_tile_entries = 0;
    int longest = 0;
    // Two passes at most: the first cuts plain tiles and decides on them whether the matrix gets balanced tiles instead (below);
    // only if it does not, and rows of 161 ... 512 entries exist, the second cuts again with multi-window tiles allowed.
    bool want_balanced = false;
    long long multi_candidates = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const bool allow_multi = pass == 1;
        std::vector<Range> part((size_t) nchunks);
        next_panel = 0;
        {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const int nthreads = (int) std::min<unsigned>({(unsigned) nchunks, hw, 16u});
            // (column panels -- break_rows > 0 -- share next_panel and the Range vector across tile_range calls: one chunk only)
            if (break_rows > 0 && nchunks != 1)
                return fail(SPMV_HIP_ERR_STATE, "internal: a panel tiling must be cut in one chunk");
            std::atomic<int> next{0};
            run_on_host_threads(nthreads, [&] {
                for (int c = next.fetch_add(1); c < nchunks; c = next.fetch_add(1))
                    tile_range(bound[(size_t) c], bound[(size_t) c + 1], allow_multi, part[(size_t) c]);
            });
        }
        desc.clear();
        stream_tiles = stream_tile_entries = 0;
        pl->uniform_tiles = pl->long_blocks = pl->split_rows = pl->longest_tile_row = pl->multi_window_tiles = pl->block_cuts = 0;
        size_t total = 0;
        for (auto const & o : part)
            total += o.desc.size();
        desc.reserve(total + 1);
        for (auto const & o : part) {
            desc.insert(desc.end(), o.desc.begin(), o.desc.end());
            stream_tiles += o.stream_tiles;
            stream_tile_entries += o.stream_tile_entries;
            if (pass == 0)
                multi_candidates += o.multi_candidates;
            pl->uniform_tiles += o.uniform_tiles;
            pl->long_blocks += o.long_blocks;
            pl->split_rows += o.split_rows;
            pl->longest_tile_row = std::max(pl->longest_tile_row, o.longest_tile_row);
            pl->multi_window_tiles += o.multi_window_tiles;
            pl->block_cuts += o.block_cuts;
            longest = std::max(longest, o.longest);
        }
        if (pass == 0) {
            // (rows with a wave or more to themselves are the same in both tilings and do not count)
            want_balanced = !exact && tile == 512 && break_rows == 0 && !(flags & SPMV_HIP_FLAG_NO_BALANCED_TILES)
                && longest > 16 && 2 * stream_tile_entries < stream_tiles * tile;
            if (want_balanced || multi_candidates == 0)
                break;
        }
    }
    int32_t r = 0;
    // Balanced tiles: when the tiles above come out mostly empty BECAUSE rows are skewed (a long row
    // limits its tile to the rows the wave has lanes for), fill tiles by entries instead -- up to 512 in
    // up to 256 whole rows -- and let csr_segtile_kernel add the rows up by segmented reduction.
    // Regular matrices (every tile already full, or only short rows) keep the tiles above and with
    // them the reference's summation order.
    {
        const bool want = want_balanced;
        if (want) {
            desc.clear();
            pl->uniform_tiles = pl->long_blocks = pl->split_rows = pl->longest_tile_row = pl->multi_window_tiles = 0;
            r = 0;
            while (r < rows) {
                const int32_t kb = p[r] & ~3;
                int32_t r1 = r;
                int32_t maxlen = 0;
                while (r1 < rows && (r1 - r) < spmv::kSegMaxRows && (long long) p[r1 + 1] - kb <= tile) {
                    maxlen = std::max(maxlen, p[r1 + 1] - p[r1]);
                    ++r1;
                }
                if (r1 == r) { // one row longer than a tile
                    const long long len = (long long) p[r + 1] - p[r];
                    pl->long_blocks++;
                    if (len > split_threshold) {
                        pl->split_rows++;
                        for (long long k = p[r]; k < p[r + 1]; k += split_chunk)
                            desc.push_back(make_int4((int) (r | 0x80000000u), (int) k, 0, 0));
                    } else {
                        desc.push_back(make_int4(r, p[r], 0, 0));
                    }
                    r1 = r + 1;
                } else {
                    pl->longest_tile_row = std::max(pl->longest_tile_row, (int) maxlen);
                    const bool fast = p[r1] > p[r] && (((long long) p[r1] - 1) | 3) < (long long) p[rows];
                    desc.push_back(make_int4(r, p[r], maxlen | (fast ? (1 << 25) : 0) | spmv::kTileMetaSeg, 0));
                }
                r = r1;
            }
            pl->balanced = true;
            pl->block_hint = 0;
        }
    }
    pl->ntiles = (int) desc.size();
    if (break_rows > 0) {
        while (next_panel <= 8)
            pl->pinfo.first[next_panel++] = pl->ntiles;
        pl->pinfo.rows = break_rows;
    }
    desc.push_back(make_int4(rows, p[rows], 0, 0));
    pl->nblk = pl->ntiles;
    pl->workgroups = (pl->ntiles + 3) / 4;
    if (pl->ntiles > 0) {
        pl->meta_bytes = desc.size() * sizeof(int4);
        hipError_t e = hipMalloc((void **) &pl->d_tiles, pl->meta_bytes);
        if (e == hipSuccess)
            e = hipMemcpy(pl->d_tiles, desc.data(), pl->meta_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            int rc = fail_hip(e, "plan metadata upload");
            if (pl->d_tiles)
                (void) hipFree(pl->d_tiles);
            pl->d_tiles = nullptr;
            return rc;
        }
    }
    return SPMV_HIP_OK;
}

// break_rows > 0: no tile may contain a row index that is a multiple of break_rows except as its
// first row (column panels: tiles stay inside one panel)
namespace spmvi {
int plan_csr_internal(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p,
                             int algorithm, int lanes_per_row, unsigned flags, int32_t break_rows)
{
    if (!out)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    *out = nullptr;
    if (rows < 0 || cols < 0 || !p)
        return fail(SPMV_HIP_ERR_INVALID, "rows/cols negative or row_ptr null");
    if (p[0] < 0)
        return fail(SPMV_HIP_ERR_INVALID, "row_ptr[0] is negative");
    {
        // (one pass over row_ptr; side by side on the host's threads once it is long enough to matter)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int nthreads = rows < (1 << 18) ? 1 : (int) std::min(hw, 16u);
        std::atomic<int> decreasing{0};
        const int nparts = nthreads <= 1 ? 1 : 4 * nthreads;
        std::atomic<int> next{0};
        run_on_host_threads(nthreads, [&] {
            for (int c = next.fetch_add(1); c < nparts; c = next.fetch_add(1)) {
                const int32_t cb = (int32_t) ((long long) rows * c / nparts), ce = (int32_t) ((long long) rows * (c + 1) / nparts);
                int bad = 0;
                for (int32_t r = cb; r < ce; ++r)
                    bad |= p[r + 1] < p[r];
                if (bad)
                    decreasing.store(1);
            }
        });
        if (decreasing.load())
            return fail(SPMV_HIP_ERR_INVALID, "row_ptr is not non-decreasing");
    }
    if (algorithm < SPMV_HIP_CSR_AUTO || algorithm > SPMV_HIP_CSR_WAVETILE)
        return fail(SPMV_HIP_ERR_INVALID, "unknown CSR algorithm");
    if (flags & ~kKnownFlags)
        return fail(SPMV_HIP_ERR_INVALID, "unknown flag bits");
    if (lanes_per_row != 0 &&
        (lanes_per_row < 2 || lanes_per_row > 64 || (lanes_per_row & (lanes_per_row - 1))))
        return fail(SPMV_HIP_ERR_INVALID, "lanes_per_row must be 0 or a power of two in 2..64");

    spmv_hip_plan * pl = new (std::nothrow) spmv_hip_plan;
    if (!pl)
        return fail(SPMV_HIP_ERR_ALLOC, "plan allocation failed");
    pl->rows = rows;
    pl->cols = cols;
    pl->nnz = p[rows];
    pl->flags = flags;
    const double mean = rows > 0 ? double(p[rows] - p[0]) / rows : 0.0;

    if (algorithm == SPMV_HIP_CSR_AUTO)
        algorithm = SPMV_HIP_CSR_WAVETILE;
    if (flags & SPMV_HIP_FLAG_EXACT_ORDER) {
        if (algorithm == SPMV_HIP_CSR_VECTOR)
            algorithm = SPMV_HIP_CSR_WAVETILE;
    }
    pl->algorithm = algorithm;

    int split_threshold = kSplitThreshold, split_chunk = kSplitChunk;
#ifdef SPMV_HIP_EXPERIMENTS
    if (const char * v = std::getenv("SPMV_HIP_SPLIT_THRESHOLD")) split_threshold = std::max(512, std::atoi(v)); // tools/ab.py
    if (const char * v = std::getenv("SPMV_HIP_SPLIT_CHUNK")) split_chunk = std::max(64, std::atoi(v));
#endif
    if (algorithm == SPMV_HIP_CSR_SCALAR) {
        pl->workgroups = grid_for(rows, kBlock);
    } else if (algorithm == SPMV_HIP_CSR_VECTOR) {
        pl->lanes_per_row = lanes_per_row ? lanes_per_row : pick_lanes(mean);
        pl->workgroups = grid_for((long long) rows * pl->lanes_per_row, kBlock, cu_count() * 32);
    } else if (algorithm == SPMV_HIP_CSR_WAVETILE) {
        pl->break_rows = break_rows;
        int rc = build_wave_tiles(pl, p, flags, break_rows, split_threshold, split_chunk);
        if (rc != SPMV_HIP_OK) {
            delete pl;
            return rc;
        }
    } else {
        // adaptive: cut rows into blocks of <= kTile entries (from the 4-aligned
        // start of the block's first row) and <= kBlock rows
        std::vector<int32_t> blk;
        blk.reserve((size_t) rows / 64 + 16);
        blk.push_back(0);
        int32_t r = 0;
        while (r < rows) {
            const int32_t kb = p[r] & ~3;
            int32_t r1 = r;
            while (r1 < rows && (r1 - r) < kBlock && (long long) p[r1 + 1] - kb <= kTile)
                ++r1;
            if (r1 == r) { // one row longer than a tile
                r1 = r + 1;
                pl->long_blocks++;
            }
            blk.push_back(r1);
            r = r1;
        }
        pl->nblk = (int) blk.size() - 1;
        pl->workgroups = pl->nblk;
        if (pl->nblk > 0) {
            pl->meta_bytes = blk.size() * sizeof(int32_t);
            hipError_t e = hipMalloc((void **) &pl->d_blk_row, pl->meta_bytes);
            if (e == hipSuccess)
                e = hipMemcpy(pl->d_blk_row, blk.data(), pl->meta_bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                int rc = fail_hip(e, "plan metadata upload");
                if (pl->d_blk_row)
                    (void) hipFree(pl->d_blk_row);
                delete pl;
                return rc;
            }
        }
    }
    int rc_acc = plan_account(pl, false);
    if (rc_acc != SPMV_HIP_OK) {
        spmv_hip_plan_destroy(pl);
        return rc_acc;
    }
    *out = pl;
    return SPMV_HIP_OK;
}

} // namespace spmvi

extern "C" {

void spmv_hip_plan_destroy(spmv_hip_plan * pl)
{
    if (!pl)
        return;
    if (pl->d_blk_row)
        (void) hipFree(pl->d_blk_row);
    if (pl->d_tiles)
        (void) hipFree(pl->d_tiles);
    if (pl->d_col16)
        (void) hipFree(pl->d_col16);
    if (pl->d_patterns)
        (void) hipFree(pl->d_patterns);
    if (pl->d_blocks)
        (void) hipFree(pl->d_blocks);
    if (pl->d_segblocks)
        (void) hipFree(pl->d_segblocks);
    if (pl->d_rest_tiles)
        (void) hipFree(pl->d_rest_tiles);
    if (pl->d_group_tiles)
        (void) hipFree(pl->d_group_tiles);
    if (pl->d_group_rest)
        (void) hipFree(pl->d_group_rest);
    if (pl->d_vidx)
        (void) hipFree(pl->d_vidx);
    if (pl->d_vtab)
        (void) hipFree(pl->d_vtab);
    if (pl->d_tiles_vi)
        (void) hipFree(pl->d_tiles_vi);
    if (pl->d_colh)
        (void) hipFree(pl->d_colh);
    if (pl->d_hub_column)
        (void) hipFree(pl->d_hub_column);
    if (pl->d_hubx)
        (void) hipFree(pl->d_hubx);
    if (pl->inner)
        spmv_hip_plan_destroy(pl->inner);
    if (pl->d_vrow_ptr)
        (void) hipFree(pl->d_vrow_ptr);
    if (pl->d_pcol)
        (void) hipFree(pl->d_pcol);
    if (pl->d_pval)
        (void) hipFree(pl->d_pval);
    delete pl;
}

int spmv_hip_internal_exclusive_scan_i32(const int32_t * d_in, int32_t * d_out, long long n, hipStream_t s);

#ifdef SPMV_HIP_EXPERIMENTS // retired from the product library (internal.hpp)
// Hub columns (csr_hub.hpp; opt-in, SPMV_HIP_FLAG_HUB_COLUMNS) for balanced plans -- graph matrices -- whose x does not fit an XCD's L2: in-degrees, the columns
// with at least `threshold` references (8, doubled while more than 2^18 qualify), the plan's own column stream.  Nothing is kept
// unless the hubs are few (<= 2^18: 2 MB of x, resident in every L2) and carry a share of the entries worth a second launch
// per multiply (>= 10 %).
static int plan_hub_columns(spmv_hip_plan * pl, const int32_t * d_column_index, hipStream_t s)
{
    if (!pl->balanced || pl->tile != 512 || !(pl->flags & SPMV_HIP_FLAG_HUB_COLUMNS) || (pl->flags & SPMV_HIP_FLAG_XCD_REMAP) || pl->nnz < (1 << 20)
        || (long long) pl->cols * 8 < 4LL * 1024 * 1024 || pl->cols >= (1 << 29))
        return SPMV_HIP_OK;
    int32_t * d_degree = nullptr, * d_flag = nullptr, * d_slot = nullptr;
    unsigned long long * d_stats = nullptr;
    unsigned long long stats[2] = {0, 0};
    const int cols = pl->cols;
    const unsigned cgrid = (unsigned) ((cols + 255) / 256);
    hipError_t e = hipMalloc((void **) &d_degree, (size_t) cols * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &d_flag, ((size_t) cols + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &d_slot, ((size_t) cols + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &d_stats, sizeof(stats));
    if (e == hipSuccess) e = hipMemsetAsync(d_degree, 0, (size_t) cols * sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, ((size_t) cols + 1) * sizeof(int32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::hub_count_kernel, dim3((unsigned) grid_for(pl->nnz, kBlock, cu_count() * 16)), dim3(256), 0, s, (long long) pl->nnz,
                           d_column_index, d_degree);
        e = hipGetLastError();
    }
    int threshold = 8;
    int rc = SPMV_HIP_OK;
    for (int attempt = 0; e == hipSuccess && attempt < 6; ++attempt, threshold *= 2) {
        e = hipMemsetAsync(d_stats, 0, sizeof(stats), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spmv::hub_flag_kernel, dim3(cgrid), dim3(256), 0, s, cols, d_degree, threshold, d_flag, d_stats);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess || stats[0] <= (1u << 18))
            break;
    }
    const bool take = e == hipSuccess && stats[0] >= 256 && stats[0] <= (1u << 18) && 10 * stats[1] >= (unsigned long long) pl->nnz;
    if (take) {
        const int nhubs = (int) stats[0];
        if (spmv_hip_internal_exclusive_scan_i32(d_flag, d_slot, (long long) cols + 1, s) != 0)
            e = hipErrorUnknown;
        if (e == hipSuccess) e = hipMalloc((void **) &pl->d_colh, (size_t) pl->nnz * sizeof(int32_t) + 64);
        if (e == hipSuccess) e = hipMalloc((void **) &pl->d_hub_column, (size_t) nhubs * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **) &pl->d_hubx, (size_t) nhubs * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(pl->d_colh, 0, (size_t) pl->nnz * sizeof(int32_t) + 64, s);
        if (e == hipSuccess) e = hipMemsetAsync(pl->d_hubx, 0, (size_t) nhubs * sizeof(double), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spmv::hub_list_kernel, dim3(cgrid), dim3(256), 0, s, cols, d_flag, d_slot, pl->d_hub_column);
            hipLaunchKernelGGL(spmv::hub_remap_kernel, dim3((unsigned) grid_for(pl->nnz, kBlock, cu_count() * 16)), dim3(256), 0, s, (long long) pl->nnz,
                               d_column_index, d_flag, d_slot, pl->d_colh);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            pl->nhubs = nhubs;
            pl->hub_threshold = threshold;
            pl->hub_entries = (long long) stats[1];
            pl->meta_bytes += (size_t) pl->nnz * sizeof(int32_t) + 64 + (size_t) nhubs * 12;
        } else {
            for (void * q : {(void *) pl->d_colh, (void *) pl->d_hub_column, (void *) pl->d_hubx})
                if (q)
                    (void) hipFree(q);
            pl->d_colh = nullptr;
            pl->d_hub_column = nullptr;
            pl->d_hubx = nullptr;
        }
    }
    for (void * q : {(void *) d_degree, (void *) d_flag, (void *) d_slot, (void *) d_stats})
        if (q)
            (void) hipFree(q);
    if (e != hipSuccess)
        rc = fail_hip(e, "hub columns");
    return rc;
}
#endif

int spmv_hip_plan_csr_compress(spmv_hip_plan * pl, const int32_t * d_column_index, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->ntiles == 0 || pl->nnz == 0)
        return SPMV_HIP_OK; // nothing to compress for the other algorithms
    if (!d_column_index)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    if (pl->d_col16)
        return fail(SPMV_HIP_ERR_STATE, "plan is already compressed");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (with a block hint the block stream of csr_blocktile.hpp lives behind the 16-bit columns, in the same allocation)
    // (... and behind that the 32-bit words of the masked block tiles)
    const size_t bytes = pl->block_hint ? spmv::mask_stream_offset(pl->nnz) * sizeof(uint16_t) + spmv::mask_stream_words(pl->nnz) * sizeof(uint32_t)
                                        : (size_t) pl->nnz * sizeof(uint16_t) + 64;
    const bool want_patterns = !(pl->flags & SPMV_HIP_FLAG_NO_SHIFTED_TILES);
    int * d_count = nullptr;
    unsigned long long * d_fp = nullptr;
    HIP_TRY(hipMalloc((void **) &pl->d_col16, bytes));
    int counts[5] = {0, 0, 0, 0, 0};
    hipError_t e = hipMalloc((void **) &d_count, kStripedInts * sizeof(int));
    if (e == hipSuccess && want_patterns) {
        e = hipMalloc((void **) &d_fp, (size_t) pl->ntiles * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d_fp, 0, (size_t) pl->ntiles * sizeof(unsigned long long), s);
    }
    if (e == hipSuccess) e = hipMemsetAsync(pl->d_col16, 0, bytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, kStripedInts * sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::csr_tile_compress_kernel, dim3((pl->ntiles + 3) / 4), dim3(256), 0, s,
                           pl->ntiles, pl->tile, pl->d_tiles, d_column_index, pl->d_col16, d_count,
                           (pl->flags & SPMV_HIP_FLAG_NO_SHIFTED_TILES) ? 0 : 1, d_fp,
                           // "scattered" (counts[4]): a tile whose columns reach further than one column panel -- an eighth of the matrix --
                           // AND further than an XCD's L2 holds of x beside the matrix streams (3 MB: the same figure as the matrix-level
                           // test in repack).  Round 6: a mesh in RCM order has a band of ~7 n^(2/3) nodes, which for 0.3 ... 1 M nodes is
                           // more than an eighth of the matrix and still only ~1 MB of x (the Delaunay twin with 300 K nodes was given
                           // column panels and ran at 0.56 where 100 K and 500 K nodes ran at 0.90: profiles/r06_delaunay_panel_cliff.log)
                           std::max(std::max(1, (pl->cols + 7) / 8), 3 * 1024 * 1024 / 8));
        e = hipGetLastError();
    }
    // patterns: the shifted tiles' shape fingerprints come back to the host, the most frequent
    // shapes become patterns, and the tiles that really have one of those shapes are marked
    // (first-row columns from the pattern; a window of runs where it pays)
    if (e == hipSuccess && d_fp) {
        std::vector<unsigned long long> fp((size_t) pl->ntiles);
        e = hipMemcpyAsync(fp.data(), d_fp, fp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        std::vector<std::pair<unsigned long long, std::pair<int, int>>> shapes; // fingerprint, (count, first tile)
        if (e == hipSuccess) {
            std::unordered_map<unsigned long long, size_t> index;
            for (int w = 0; w < pl->ntiles; ++w) {
                if (!fp[(size_t) w])
                    continue;
                auto it = index.find(fp[(size_t) w]);
                if (it == index.end()) {
                    index.emplace(fp[(size_t) w], shapes.size());
                    shapes.push_back({fp[(size_t) w], {1, w}});
                } else {
                    shapes[it->second].second.first++;
                }
            }
            std::sort(shapes.begin(), shapes.end(),
                      [](auto const & a, auto const & b) { return a.second.first > b.second.first; });
            if (shapes.size() > (size_t) spmv::kMaxPatterns)
                shapes.resize((size_t) spmv::kMaxPatterns);
        }
        if (e == hipSuccess && !shapes.empty()) {
            const int np = (int) shapes.size();
            std::vector<int> rep((size_t) np);
            std::vector<unsigned long long> pfp((size_t) np);
            for (int i = 0; i < np; ++i) {
                rep[(size_t) i] = shapes[(size_t) i].second.second;
                pfp[(size_t) i] = shapes[(size_t) i].first;
            }
            int * d_rep = nullptr;
            unsigned long long * d_pfp = nullptr;
            const size_t pat_bytes = (size_t) np * spmv::kPatStride * sizeof(int32_t);
            e = hipMalloc((void **) &pl->d_patterns, pat_bytes);
            if (e == hipSuccess) e = hipMalloc((void **) &d_rep, (size_t) np * sizeof(int));
            if (e == hipSuccess) e = hipMalloc((void **) &d_pfp, (size_t) np * sizeof(unsigned long long));
            if (e == hipSuccess) e = hipMemcpyAsync(d_rep, rep.data(), (size_t) np * sizeof(int), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(d_pfp, pfp.data(), (size_t) np * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::csr_pattern_build_kernel, dim3(np), dim3(64), 0, s, d_rep, pl->d_tiles,
                                   d_column_index, pl->d_patterns);
                hipLaunchKernelGGL(spmv::csr_pattern_assign_kernel, dim3((pl->ntiles + 3) / 4), dim3(256), 0, s,
                                   pl->ntiles, pl->d_tiles, d_column_index, d_fp, d_pfp, np, pl->d_patterns, d_count);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (d_rep) (void) hipFree(d_rep);
            if (d_pfp) (void) hipFree(d_pfp);
            if (e == hipSuccess) {
                pl->npatterns = np;
                pl->meta_bytes += pat_bytes;
            }
        }
    }
    if (e == hipSuccess) e = read_striped(d_count, counts, 5, s);
    pl->narrow_tiles = counts[0];
    pl->shifted_tiles = counts[1];
    pl->xwin_tiles = counts[2];
    pl->spread_tiles = counts[4];
    // segment windows (x staged through LDS per block of 32 tiles, in up to 12 far-apart column segments: meshes in
    // natural ordering, KKT systems) for what has no cheaper path: first count the tiles that would qualify, and
    // only if they are the majority mark them and rewrite their 16-bit column stream to window slots
    if (e == hipSuccess && pl->tile == 512 && !pl->balanced && pl->ntiles >= 4 * 32
        && !(pl->flags & (SPMV_HIP_FLAG_NO_X_WINDOW | SPMV_HIP_FLAG_NO_SEGMENT_WINDOW | SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_XCD_REMAP))) {
        int per_block = 32, max_slots = 4096, take_narrow = 0;
#ifdef SPMV_HIP_EXPERIMENTS
        if (const char * v = std::getenv("SPMV_HIP_SEGWIN_NARROW")) take_narrow = std::atoi(v) != 0; // windows for blocks of narrow tiles too
        if (const char * v = std::getenv("SPMV_HIP_SEGWIN_TILES")) per_block = std::max(8, std::min(256, std::atoi(v)));
        if (const char * v = std::getenv("SPMV_HIP_SEGWIN_SLOTS")) max_slots = std::max(512, std::min(4096, std::atoi(v)));
#endif
        const int nb = (pl->ntiles + per_block - 1) / per_block;
        int shift = 0; // columns per bitmap bit = 2^shift: the whole column space in 65536 bits
        while (((long long) pl->cols - 1) >> shift >= 65536)
            ++shift;
        // (counts[3] only ever grows on the device: every pass is read as the difference to the reading before it)
        int before3 = counts[3];
        {
            hipLaunchKernelGGL(spmv::csr_segwin_mark_kernel, dim3(nb), dim3(512), 0, s, pl->ntiles, pl->tile, per_block, pl->d_tiles,
                               d_column_index, pl->d_col16, (spmv::SegWinBlock *) nullptr, d_count, 0, shift, max_slots, take_narrow);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = read_striped(d_count, counts, 5, s);
        const int seg_candidates = counts[3] - before3;
        if (e == hipSuccess && 2 * (long long) seg_candidates > pl->ntiles) {
            e = hipMalloc((void **) &pl->d_segblocks, (size_t) nb * sizeof(spmv::SegWinBlock));
            if (e == hipSuccess) e = hipMemsetAsync(pl->d_segblocks, 0, (size_t) nb * sizeof(spmv::SegWinBlock), s);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::csr_segwin_mark_kernel, dim3(nb), dim3(512), 0, s, pl->ntiles, pl->tile, per_block, pl->d_tiles,
                                   d_column_index, pl->d_col16, pl->d_segblocks, d_count, 1, shift, max_slots, take_narrow);
                e = hipGetLastError();
            }
            std::vector<spmv::SegWinBlock> hb;
            if (e == hipSuccess) {
                hb.resize((size_t) nb);
                e = hipMemcpyAsync(hb.data(), pl->d_segblocks, hb.size() * sizeof(spmv::SegWinBlock), hipMemcpyDeviceToHost, s);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess) {
                pl->nsegblocks = nb;
                pl->seg_tiles_per_block = per_block;
                pl->segwin_tiles = 0;
                pl->segwin_slots = 0;
                for (auto const & b : hb)
                    if (b.nseg > 0) {
                        pl->segwin_tiles += b.ntiles;
                        pl->segwin_slots = std::max(pl->segwin_slots, b.slots);
                    }
                pl->blockwin_tiles = pl->segwin_tiles;
                pl->meta_bytes += (size_t) nb * sizeof(spmv::SegWinBlock);
            }
        }
        if (e == hipSuccess) e = read_striped(d_count, counts, 5, s);
    }
    // block windows (x staged through LDS per 16 tiles) for what has no cheaper path: first count
    // the tiles that would qualify, and only if they are the majority mark them
    if (e == hipSuccess && !pl->d_segblocks && pl->tile == 512 && !pl->balanced && pl->ntiles >= 4 * spmv::kBlockWinTiles
        && !(pl->flags & (SPMV_HIP_FLAG_NO_X_WINDOW | SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_XCD_REMAP))) {
        const int nb = (pl->ntiles + spmv::kBlockWinTiles - 1) / spmv::kBlockWinTiles;
        const int before3 = counts[3];
        hipLaunchKernelGGL(spmv::csr_blockwin_mark_kernel, dim3(nb), dim3(1024), 0, s, pl->ntiles, pl->tile, pl->d_tiles,
                           pl->d_col16, (int2 *) nullptr, d_count, 0);
        e = hipGetLastError();
        if (e == hipSuccess) e = read_striped(d_count, counts, 5, s);
        const int win_candidates = counts[3] - before3;
        if (e == hipSuccess && 2 * (long long) win_candidates > pl->ntiles) {
            e = hipMalloc((void **) &pl->d_blocks, (size_t) nb * sizeof(int2));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::csr_blockwin_mark_kernel, dim3(nb), dim3(1024), 0, s, pl->ntiles, pl->tile,
                                   pl->d_tiles, pl->d_col16, pl->d_blocks, d_count, 1);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess) {
                pl->nblocks16 = nb;
                pl->blockwin_tiles = win_candidates;
                pl->meta_bytes += (size_t) nb * sizeof(int2);
            }
        }
    }
    if (d_count)
        (void) hipFree(d_count);
    if (d_fp)
        (void) hipFree(d_fp);
    if (e != hipSuccess) {
        (void) hipFree(pl->d_col16);
        pl->d_col16 = nullptr;
        if (pl->d_patterns) {
            (void) hipFree(pl->d_patterns);
            pl->d_patterns = nullptr;
        }
        if (pl->d_blocks) {
            (void) hipFree(pl->d_blocks);
            pl->d_blocks = nullptr;
            pl->nblocks16 = 0;
        }
        if (pl->d_segblocks) {
            (void) hipFree(pl->d_segblocks);
            pl->d_segblocks = nullptr;
            pl->nsegblocks = 0;
        }
        return fail_hip(e, "index compression");
    }
    pl->meta_bytes += bytes;
    pl->compressed_from = d_column_index;
    int rc = device_column_checksum(d_column_index, pl->nnz, &pl->column_checksum, s);
#ifdef SPMV_HIP_EXPERIMENTS
    if (rc == SPMV_HIP_OK)
        rc = plan_hub_columns(pl, d_column_index, s);
#endif
    if (rc == SPMV_HIP_OK)
        rc = plan_account(pl, true);
    pl->verify_pending = true;
    return rc;
}

int spmv_hip_plan_csr_confirm_blocks(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index,
                                          const int32_t * host_row_ptr, void * stream);

int spmv_hip_plan_verify(spmv_hip_plan * pl, const int32_t * d_column_index, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    pl->verify_pending = false; // an explicit check stands in for the first multiply's
    return verify_plan(pl, d_column_index, static_cast<hipStream_t>(stream));
}

// defined in coo_sort.hip (hipCUB): out[i] = sum of in[0..i), n elements
int spmv_hip_internal_exclusive_scan_i32(const int32_t * d_in, int32_t * d_out, long long n, hipStream_t s);

static void drop_value_dictionary(spmv_hip_plan * pl);
static int repack_stages(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index, const double * d_value,
                         void * stream, const double ** reindex);

int spmv_hip_plan_csr_repack(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index,
                             const double * d_value, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    if (pl->inner)
        return fail(SPMV_HIP_ERR_STATE, "plan is already repacked");
    // A value dictionary built BEFORE this call (compress, index_values, repack -- the header allows the order) describes the
    // tiling it was built on: when a stage below cuts the tiles anew it is dropped first and built again at the end.
    const double * reindex = nullptr;
    const bool had_windows = pl->d_blocks || pl->d_segblocks;
    int rc = repack_stages(pl, d_row_ptr, d_column_index, d_value, stream, &reindex);
    // ... and a dictionary that was ASKED FOR before this call but could not be built then -- the plan ran a window kernel, which
    // has no dictionary variant -- is asked for again when a stage above gave the windows up (a structured grid whose boundary tiles
    // became masked stencil tiles): both orders of the calls end in the same plan (tests/test_gpu_stenciltiles.py)
    if (rc == SPMV_HIP_OK && !reindex && pl->values_wanted && pl->nvalues == 0 && had_windows && !pl->d_blocks && !pl->d_segblocks)
        reindex = pl->values_wanted;
    if (rc == SPMV_HIP_OK && reindex && !pl->inner)
        rc = spmv_hip_plan_csr_index_values(pl, reindex, stream);
    return rc;
}

// The tiles of a compressed plan cut once more from row_ptr fetched back from the device, and classified again (plan time only):
// without the block hint (it was wrong and cost tile fill), or on the grid of triples a candidate matrix turned out to have.
static int rebuild_tiles(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index, void * stream,
                         const uint32_t * group_bits /* null: without block tiles */, const double ** reindex,
                         const int32_t * host_row_ptr = nullptr /* the caller still has it: no copy back */)
{
    const bool with_blocks = group_bits != nullptr;
    const int wrong_hint = pl->block_hint;
    const bool wrong_skewed = pl->block_skewed;
    const bool wrong_from_bits = pl->hint_from_bits;
    const bool was_compressed = pl->d_col16 != nullptr; // (a plan that has not been classified yet is not classified here either)
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<int32_t> hp;
    if (!host_row_ptr) {
        hp.resize((size_t) pl->rows + 1);
        HIP_TRY(hipMemcpyAsync(hp.data(), d_row_ptr, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        host_row_ptr = hp.data();
    }
    // the dictionary's tile list (d_tiles_vi) and its constant-row marks number the OLD tiles and patterns
    if (pl->values_from)
        *reindex = pl->values_from;
    drop_value_dictionary(pl);
    for (void * q : {(void *) pl->d_tiles, (void *) pl->d_col16, (void *) pl->d_patterns, (void *) pl->d_blocks, (void *) pl->d_segblocks,
                     (void *) pl->d_rest_tiles})
        if (q)
            (void) hipFree(q);
    pl->d_tiles = nullptr; pl->d_col16 = nullptr; pl->d_patterns = nullptr; pl->d_blocks = nullptr; pl->d_segblocks = nullptr;
    pl->d_rest_tiles = nullptr;
    pl->ntiles = pl->nblk = pl->workgroups = 0;
    pl->narrow_tiles = pl->shifted_tiles = pl->xwin_tiles = pl->longest_tile_row = pl->spread_tiles = 0;
    pl->nblocks16 = pl->blockwin_tiles = pl->nrest_tiles = pl->nsegblocks = pl->segwin_tiles = pl->segwin_slots = pl->npatterns = 0;
    pl->uniform_tiles = pl->split_rows = pl->long_blocks = pl->multi_window_tiles = 0;
    pl->balanced = false;
    pl->block_hint = pl->block_cuts = pl->block_tiles = pl->masked_block_tiles = pl->block_candidate = pl->stencil_mask_tiles = pl->colshare_tiles = 0;
    pl->colshare_entries = 0;
    pl->stencil_mask_entries = 0;
    pl->block_entries = pl->masked_block_entries = 0;
    pl->meta_bytes = 0;
    pl->compressed_from = nullptr;
    if (!with_blocks) {
        // a hint read from row_ptr alone was wrong: it is not taken again, the others (triples / groups of 2 or 4 / the candidate
        // whose columns are looked at) still get their turn; groups found in the columns themselves that made no block tiles end it
        if (wrong_from_bits || wrong_hint == 0)
            pl->flags |= SPMV_HIP_FLAG_NO_BLOCK_TILES;
        else
            pl->hints_tried |= wrong_hint == 3 ? (wrong_skewed ? 4 : 1) : 2;
    }
    pl->group_bits = group_bits;
    int rc = build_wave_tiles(pl, host_row_ptr, pl->flags, pl->break_rows, kSplitThreshold, kSplitChunk);
    pl->group_bits = nullptr;
    if (rc == SPMV_HIP_OK)
        rc = plan_account(pl, false);
    if (rc == SPMV_HIP_OK && was_compressed)
        rc = spmv_hip_plan_csr_compress(pl, d_column_index, stream);
    return rc;
}

// A CANDIDATE for block tiles (rows of similar length, spmv_hip_plan_csr): which rows have the same columns as the row in front of
// them (csr_row_group_kernel)?  If half of the rows stand in groups of three the tiles are cut again on those groups; the block
// stage of repack marks them later.  Asked once per plan: by spmv_hip_plan_csr_repack, or -- where the caller owns all the
// arrays (spmv_hip_upload_csr) -- BEFORE the tiles are classified, which saves classifying them twice.
static int confirm_block_candidate(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index, void * stream,
                                   const double ** reindex, const int32_t * host_row_ptr)
{
    if (!pl->block_candidate || pl->block_hint || !d_row_ptr || !d_column_index
        || pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->tile != 512 || pl->balanced || pl->ntiles == 0 || pl->rows < 192 || pl->nnz == 0
        || (pl->flags & (SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_NO_BLOCK_TILES | SPMV_HIP_FLAG_NO_MASKED_BLOCKS)))
        return SPMV_HIP_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    pl->block_candidate = 0; // asked once
    const size_t words = 2 * (((size_t) pl->rows + 63) / 64);
    uint32_t * d_bits = nullptr;
    unsigned long long * d_count = nullptr;
    unsigned long long triples[1] = {0};
    HIP_TRY(hipMalloc((void **) &d_bits, words * sizeof(uint32_t)));
    hipError_t e = hipMalloc((void **) &d_count, kStripedInts * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, kStripedInts * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned) (((size_t) pl->rows + 255) / 256);
        hipLaunchKernelGGL(spmv::csr_row_group_kernel, dim3(grid), dim3(256), 0, s, pl->rows, d_row_ptr, d_column_index, d_bits);
        hipLaunchKernelGGL(spmv::csr_row_triple_count_kernel, dim3(grid), dim3(256), 0, s, pl->rows, d_bits, d_count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = read_striped(d_count, triples, 1, s);
    std::vector<uint32_t> bits;
    const bool confirmed = e == hipSuccess && 6 * triples[0] >= (unsigned long long) pl->rows; // half of the rows in triples
    if (confirmed) {
        bits.resize(words);
        e = hipMemcpyAsync(bits.data(), d_bits, words * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void) hipFree(d_bits);
    if (d_count) (void) hipFree(d_count);
    if (e != hipSuccess)
        return fail_hip(e, "row groups");
    if (confirmed)
        return rebuild_tiles(pl, d_row_ptr, d_column_index, stream, bits.data(), reindex, host_row_ptr);
    return SPMV_HIP_OK;
}

int spmv_hip_plan_csr_confirm_blocks(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index,
                                          const int32_t * host_row_ptr, void * stream)
{
    if (!pl || pl->d_col16)
        return SPMV_HIP_OK;
    const double * reindex = nullptr; // (no dictionary yet)
    return confirm_block_candidate(pl, d_row_ptr, d_column_index, stream, &reindex, host_row_ptr);
}

static int repack_stages(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index, const double * d_value,
                         void * stream, const double ** reindex)
{
    // (a hint that turns out wrong makes the tiles anew, and the next hint -- or the candidate -- gets its turn: at most four rounds)
    for (int attempt = 0; attempt < 4; ++attempt) {
    bool hint_was_wrong = false;
    if (pl->d_col16 && pl->compressed_from == d_column_index) {
        int rc = confirm_block_candidate(pl, d_row_ptr, d_column_index, stream, reindex, nullptr);
        if (rc != SPMV_HIP_OK)
            return rc;
    }
    // block tiles (csr_blocktile.hpp): the one structural pass that needs row_ptr next to the columns
    if (pl->block_hint && pl->block_tiles == 0 && pl->colshare_tiles == 0 && pl->d_col16 && pl->compressed_from == d_column_index && d_row_ptr
        && pl->algorithm == SPMV_HIP_CSR_WAVETILE && pl->tile == 512 && !pl->balanced && pl->ntiles > 0
        && !(pl->flags & (SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_NO_BLOCK_TILES))) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        unsigned long long * d_count = nullptr;
        unsigned long long count[6] = {0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMalloc((void **) &d_count, kStripedInts * sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(d_count, 0, kStripedInts * sizeof(unsigned long long), s);
        const bool groups = pl->block_hint != 3; // rows in groups of 2 or 4 with one column list each, not 3 x 3 blocks
        if (e == hipSuccess) {
            if (groups)
                hipLaunchKernelGGL(spmv::csr_group_mark_kernel, dim3((unsigned) ((pl->ntiles + 3) / 4)), dim3(256), 0, s, pl->ntiles, pl->tile,
                                   pl->block_hint, pl->d_tiles, d_row_ptr, d_column_index, pl->d_col16 + spmv::block_stream_offset(pl->nnz), d_count,
                                   pl->d_col16, pl->cols);
            else
                hipLaunchKernelGGL(spmv::csr_block3_mark_kernel, dim3((unsigned) ((pl->ntiles + 3) / 4)), dim3(256), 0, s, pl->ntiles, pl->tile,
                                   pl->d_tiles, d_row_ptr, d_column_index, pl->d_col16 + spmv::block_stream_offset(pl->nnz),
                                   reinterpret_cast<uint32_t *>(pl->d_col16 + spmv::mask_stream_offset(pl->nnz)), pl->nnz, pl->cols,
                                   (pl->flags & SPMV_HIP_FLAG_NO_MASKED_BLOCKS) ? 0 : 1, d_count);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = read_striped(d_count, count, 6, s);
        pl->masked_block_tiles = (int) count[4];
        pl->masked_block_entries = (long long) count[5];
        (void) hipFree(d_count);
        if (e != hipSuccess)
            return fail_hip(e, "block tiles");
        pl->block_tiles = (int) count[2];
        pl->block_entries = (long long) count[3];
        // Block tiles and block windows (x staged through LDS per 16 tiles, 10 bytes per entry) want the same narrow tiles: where
        // most of the matrix consists of blocks the windows are given up -- 8.2 bytes per entry and no second launch
        if (pl->d_blocks && 2 * (long long) count[0] > pl->ntiles) {
            hipLaunchKernelGGL(spmv::csr_clear_blockwin_kernel, dim3((unsigned) ((pl->ntiles + 255) / 256)), dim3(256), 0, s, pl->ntiles, pl->d_tiles);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            (void) hipFree(pl->d_blocks);
            pl->d_blocks = nullptr;
            pl->meta_bytes -= std::min(pl->meta_bytes, (size_t) pl->nblocks16 * sizeof(int2));
            pl->nblocks16 = 0;
            pl->blockwin_tiles = 0;
            if (pl->d_rest_tiles) {
                (void) hipFree(pl->d_rest_tiles);
                pl->d_rest_tiles = nullptr;
                pl->nrest_tiles = 0;
            }
            pl->block_tiles = (int) count[0];
            pl->block_entries = (long long) count[1];
        }
        // ... and likewise the segment windows (x staged through LDS per block of 32 tiles; their tiles' 16-bit stream holds window
        // slots, so whichever of them is no block tile goes back to its 32-bit columns).  Round 6: the Delaunay mesh with 6 unknowns
        // per node -- every tile a block tile, more than half of them claimed by segment windows first -- ran at 0.62 of the
        // roofline with the block hint declared wrong (profiles/r06_structure_zoo.log)
        if (pl->d_segblocks && 2 * (long long) count[0] > pl->ntiles) {
            hipLaunchKernelGGL(spmv::csr_clear_segwin_kernel, dim3((unsigned) ((pl->ntiles + 255) / 256)), dim3(256), 0, s, pl->ntiles, pl->d_tiles);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            (void) hipFree(pl->d_segblocks);
            pl->d_segblocks = nullptr;
            pl->meta_bytes -= std::min(pl->meta_bytes, (size_t) pl->nsegblocks * sizeof(spmv::SegWinBlock));
            pl->nsegblocks = pl->segwin_tiles = pl->segwin_slots = 0;
            pl->blockwin_tiles = 0;
            if (pl->d_rest_tiles) {
                (void) hipFree(pl->d_rest_tiles);
                pl->d_rest_tiles = nullptr;
                pl->nrest_tiles = 0;
            }
            pl->block_tiles = (int) count[0];
            pl->block_entries = (long long) count[1];
        }
        if (groups) { // (the launch hands the group size to the kernel only if there are group tiles)
            pl->colshare_tiles = pl->block_tiles;
            pl->colshare_entries = pl->block_entries;
            pl->block_tiles = 0;
            pl->block_entries = 0;
        }
        if (pl->block_cuts > 0 && 2 * (long long) count[0] < pl->ntiles) {
            // The hint was wrong (rows in equal triples, but no 3 x 3 blocks: a scalar mesh, a band) AND it made tiles shorter
            // than they would have been: the tiles are built once more without it, from row_ptr fetched back from the device,
            // and classified again -- plan time only, and only for such matrices.
            int rc = rebuild_tiles(pl, d_row_ptr, d_column_index, stream, nullptr, reindex);
            if (rc != SPMV_HIP_OK)
                return rc;
            hint_was_wrong = true;
        } else if (pl->block_tiles > 0 || pl->colshare_tiles > 0) {
            int rc = plan_account(pl, true);
            if (rc != SPMV_HIP_OK)
                return rc;
        }
    }
    if (!hint_was_wrong)
        break;
    }
    // masked stencil tiles (csr_stenciltile.hpp): the boundary rows of a structured grid (after the block stage: should that one
    // cut the tiles anew -- a wrong hint -- the marks made here would go with the old tiles)
    if (pl->stencil_mask_tiles == 0 && pl->d_col16 && pl->compressed_from == d_column_index && d_row_ptr && pl->rows >= 64
        && pl->algorithm == SPMV_HIP_CSR_WAVETILE && pl->tile == 512 && !pl->balanced && pl->ntiles > 0 && pl->shifted_tiles < pl->ntiles
        && !(pl->flags & (SPMV_HIP_FLAG_NO_SHIFTED_TILES | SPMV_HIP_FLAG_READ_ROW_PTR))) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        // (1) the stencils of the matrix, from a sample of its rows
        constexpr int kSamples = 2048, kRecord = spmv::kStencilMaskMaxLen + 1;
        const int nsamples = std::min(kSamples, (int) pl->rows);
        std::vector<int32_t> sample((size_t) nsamples * kRecord);
        int32_t * d_sample = nullptr;
        HIP_TRY(hipMalloc((void **) &d_sample, sample.size() * sizeof(int32_t)));
        hipLaunchKernelGGL(spmv::csr_row_pattern_sample_kernel, dim3((unsigned) ((nsamples + 255) / 256)), dim3(256), 0, s, pl->rows, nsamples, d_row_ptr,
                           d_column_index, d_sample);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(sample.data(), d_sample, sample.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void) hipFree(d_sample);
        if (e != hipSuccess)
            return fail_hip(e, "row pattern sample");
        std::vector<std::pair<int, std::vector<int32_t>>> found; // (count, {len, rel...})
        for (int i = 0; i < nsamples; ++i) {
            const int32_t * o = sample.data() + (size_t) i * kRecord;
            if (o[0] < 2)
                continue;
            std::vector<int32_t> key(o, o + 1 + o[0]);
            auto it = std::find_if(found.begin(), found.end(), [&](auto const & f) { return f.second == key; });
            if (it == found.end())
                found.push_back({1, key});
            else
                it->first++;
        }
        // frequent = at least 2 % of the sample; the longest first (a boundary row's stencil is a subset of the interior one's)
        found.erase(std::remove_if(found.begin(), found.end(), [&](auto const & f) { return f.first * 50 < nsamples; }), found.end());
        std::sort(found.begin(), found.end(), [](auto const & a, auto const & b) {
            return a.second[0] != b.second[0] ? a.second[0] > b.second[0] : a.first > b.first;
        });
        if (found.size() > 4)
            found.resize(4);
        // (2) the patterns to try: the sampled ones (appended to the plan's records unless they are there already), then the plan's own
        std::vector<int32_t> pat;
        if (pl->npatterns > 0) {
            pat.resize((size_t) pl->npatterns * spmv::kPatStride);
            HIP_TRY(hipMemcpy(pat.data(), pl->d_patterns, pat.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        spmv::StencilTryList list{0, {0, 0, 0, 0, 0, 0, 0, 0}};
        int np = pl->npatterns;
        for (auto const & f : found) {
            const int len = f.second[0];
            int at = -1;
            for (int q = 0; q < np && at < 0; ++q)
                if (pat[(size_t) q * spmv::kPatStride] == len
                    && std::equal(f.second.begin() + 1, f.second.end(), pat.begin() + (size_t) q * spmv::kPatStride + spmv::kPatRel))
                    at = q;
            if (at < 0) {
                at = np++;
                pat.resize((size_t) np * spmv::kPatStride, 0);
                int32_t * rec = pat.data() + (size_t) at * spmv::kPatStride;
                rec[0] = len;
                rec[1] = 0;       // (no tile shape: a record for masked stencil tiles only)
                rec[2] = 1 << 20; // no window of runs worked out
                rec[3] = *std::min_element(f.second.begin() + 1, f.second.end());
                std::copy(f.second.begin() + 1, f.second.end(), rec + spmv::kPatRel);
            }
            list.pattern[list.n++] = at;
        }
        for (int q = 0; q < pl->npatterns && list.n < 8; ++q)
            if (pat[(size_t) q * spmv::kPatStride] <= spmv::kStencilMaskMaxLen
                && std::find(list.pattern, list.pattern + list.n, q) == list.pattern + list.n)
                list.pattern[list.n++] = q;
        if (list.n > 0) {
            if (np > pl->npatterns) { // the records grew: a new array replaces the plan's
                int32_t * d_new = nullptr;
                HIP_TRY(hipMalloc((void **) &d_new, pat.size() * sizeof(int32_t)));
                e = hipMemcpy(d_new, pat.data(), pat.size() * sizeof(int32_t), hipMemcpyHostToDevice);
                if (e != hipSuccess) {
                    (void) hipFree(d_new);
                    return fail_hip(e, "stencil patterns");
                }
                if (pl->d_patterns)
                    (void) hipFree(pl->d_patterns);
                pl->d_patterns = d_new;
                pl->meta_bytes += (size_t) (np - pl->npatterns) * spmv::kPatStride * sizeof(int32_t);
                pl->npatterns = np;
            }
            // (3) the tiles whose rows all follow one of them.  Where block windows have claimed tiles (a grid with lines so short
            // that 16 tiles span fewer than 8192 columns), a dry run counts first: if stencil tiles -- shifted and masked -- would be
            // nine tenths of the matrix, the windows are given up (8 bytes per entry and no second launch against 10)
            unsigned long long * d_count = nullptr;
            unsigned long long count[2] = {0, 0};
            HIP_TRY(hipMalloc((void **) &d_count, kStripedInts * sizeof(unsigned long long)));
            // A value dictionary built BEFORE this stage (compress, index_values, repack: an order the header allows) lists the
            // boundary tiles in d_tiles_vi as plain narrow tiles, while the marks below turn their 16-bit column slots into row
            // masks: its launch would multiply with masks for columns.  Such a plan gets a dry run first;
            // if any tile would be marked, the dictionary is dropped and built again on the marked tiles (as rebuild_tiles does).
            const bool had_dictionary = pl->values_from != nullptr || pl->d_tiles_vi != nullptr;
            for (int dry = (pl->d_blocks || pl->d_segblocks || had_dictionary) ? 1 : 0; dry >= 0; --dry) {
                e = hipMemsetAsync(d_count, 0, kStripedInts * sizeof(unsigned long long), s);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(spmv::csr_stencil_mask_kernel, dim3((unsigned) ((pl->ntiles + 3) / 4)), dim3(256), 0, s, pl->ntiles, pl->tile,
                                       pl->d_tiles, d_row_ptr, d_column_index, pl->d_col16, pl->d_patterns, list, dry, d_count);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = read_striped(d_count, count, 2, s);
                if (e != hipSuccess)
                    break;
                if (dry && had_dictionary && count[0] > 0) {
                    if (pl->values_from)
                        *reindex = pl->values_from;
                    drop_value_dictionary(pl);
                }
                if (dry && (pl->d_blocks || pl->d_segblocks) && 10 * ((long long) count[0] + pl->shifted_tiles) >= 9LL * pl->ntiles) {
                    if (pl->d_blocks)
                        hipLaunchKernelGGL(spmv::csr_clear_blockwin_kernel, dim3((unsigned) ((pl->ntiles + 255) / 256)), dim3(256), 0, s, pl->ntiles, pl->d_tiles);
                    else // (segment windows rewrote their tiles' 16-bit stream to window slots: those tiles go back to 32-bit columns)
                        hipLaunchKernelGGL(spmv::csr_clear_segwin_kernel, dim3((unsigned) ((pl->ntiles + 255) / 256)), dim3(256), 0, s, pl->ntiles, pl->d_tiles);
                    e = hipGetLastError();
                    if (e == hipSuccess) e = hipStreamSynchronize(s);
                    if (e != hipSuccess)
                        break;
                    if (pl->d_blocks) {
                        (void) hipFree(pl->d_blocks);
                        pl->d_blocks = nullptr;
                        pl->meta_bytes -= std::min(pl->meta_bytes, (size_t) pl->nblocks16 * sizeof(int2));
                        pl->nblocks16 = 0;
                    }
                    if (pl->d_segblocks) {
                        (void) hipFree(pl->d_segblocks);
                        pl->d_segblocks = nullptr;
                        pl->meta_bytes -= std::min(pl->meta_bytes, (size_t) pl->nsegblocks * sizeof(spmv::SegWinBlock));
                        pl->nsegblocks = pl->segwin_tiles = pl->segwin_slots = 0;
                    }
                    pl->blockwin_tiles = 0;
                    if (pl->d_rest_tiles) {
                        (void) hipFree(pl->d_rest_tiles);
                        pl->d_rest_tiles = nullptr;
                        pl->nrest_tiles = 0;
                    }
                }
            }
            (void) hipFree(d_count);
            if (e != hipSuccess)
                return fail_hip(e, "masked stencil tiles");
            pl->stencil_mask_tiles = (int) count[0];
            pl->stencil_mask_entries = (long long) count[1];
            if (count[0] > 0) {
                int rc = plan_account(pl, true);
                if (rc != SPMV_HIP_OK)
                    return rc;
            }
        }
    }
    // column panels pay when x does not fit one XCD's L2 and the columns are scattered; they cost
    // a copy of the matrix, one virtual row per (row, panel) and atomic y updates
    // (never where the tile classes that read no column index per entry -- block, group, masked stencil tiles -- are the majority:
    // their launch streams 8.2 ... 9 bytes per entry at the triad's rate, a panel copy 12 with atomics at a quarter of it)
    const bool scattered = 2 * (long long) pl->spread_tiles > pl->ntiles && 2 * (long long) pl->shifted_tiles < pl->ntiles
        && 2 * ((long long) pl->block_tiles + pl->colshare_tiles + pl->stencil_mask_tiles) < pl->ntiles;
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->tile != 512 || pl->balanced || pl->nnz == 0 || pl->rows < 1024
        || (pl->flags & (SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_NO_COLUMN_PANELS | SPMV_HIP_FLAG_XCD_REMAP))
        || !pl->d_col16 /* not compressed: the tile classes are unknown */
        || pl->d_segblocks /* the gather already comes out of LDS */
        || (long long) pl->rows * 8 + 1 > 0x7FFFFFF0LL)
        return SPMV_HIP_OK;
#ifdef SPMV_HIP_EXPERIMENTS
    const bool force = (pl->flags & 0x4000u) != 0; // tools/gather_locality.py: panels whatever the shape
#else
    const bool force = false;
#endif
    if (!force && (!scattered || (long long) pl->cols * 8 < 3 * 1024 * 1024 || (long long) pl->nnz < 4LL * pl->rows
                   || pl->nnz < (1 << 20) /* too small for the gather to matter; keeps small matrices bit-exact */))
        return SPMV_HIP_OK;
    if (!d_row_ptr || !d_column_index || !d_value)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t rows = pl->rows;
    const long long vrows = 8LL * rows;
    const int width = (pl->cols + 7) / 8;
    int32_t * d_count = nullptr;
    hipError_t e = hipMalloc((void **) &d_count, (size_t) (vrows + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &pl->d_vrow_ptr, (size_t) (vrows + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **) &pl->d_pcol, (size_t) pl->nnz * sizeof(int32_t) + 64);
    if (e == hipSuccess) e = hipMalloc((void **) &pl->d_pval, (size_t) pl->nnz * sizeof(double) + 64);
    if (e == hipSuccess) e = hipMemsetAsync(d_count + vrows, 0, sizeof(int32_t), s);
    std::vector<int32_t> vrow_ptr;
    int rc = SPMV_HIP_OK;
    if (e == hipSuccess) {
        const unsigned grid = (unsigned) ((rows + 255) / 256);
        hipLaunchKernelGGL(spmv::csr_panel_count_kernel, dim3(grid), dim3(256), 0, s, rows, width, d_row_ptr, d_column_index, d_count);
        e = hipGetLastError();
        if (e == hipSuccess && spmv_hip_internal_exclusive_scan_i32(d_count, pl->d_vrow_ptr, vrows + 1, s) != 0)
            e = hipErrorUnknown;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spmv::csr_panel_scatter_kernel, dim3(grid), dim3(256), 0, s, rows, width, d_row_ptr, d_column_index,
                               d_value, pl->d_vrow_ptr, pl->d_pcol, pl->d_pval);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            vrow_ptr.resize((size_t) vrows + 1);
            e = hipMemcpyAsync(vrow_ptr.data(), pl->d_vrow_ptr, vrow_ptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (d_count)
        (void) hipFree(d_count);
    if (e == hipSuccess && vrow_ptr.back() != pl->nnz)
        rc = fail(SPMV_HIP_ERR_INVALID, "column index out of range while forming column panels");
    if (e == hipSuccess && rc == SPMV_HIP_OK) {
        // the panel-major matrix is a CSR matrix of 8 * rows virtual rows: plan and classify it like any other
        // (no x windows: its kernel variant has none)
        rc = plan_csr_internal(&pl->inner, (int32_t) vrows, pl->cols, vrow_ptr.data(), SPMV_HIP_CSR_WAVETILE, 0,
                               (pl->flags | SPMV_HIP_FLAG_NO_X_WINDOW | SPMV_HIP_FLAG_NO_COLUMN_PANELS) & ~SPMV_HIP_FLAG_ROWS128, rows);
        if (rc == SPMV_HIP_OK)
            rc = spmv_hip_plan_csr_compress(pl->inner, pl->d_pcol, stream);
    }
    if (e != hipSuccess || rc != SPMV_HIP_OK) {
        if (pl->inner) { spmv_hip_plan_destroy(pl->inner); pl->inner = nullptr; }
        if (pl->d_vrow_ptr) { (void) hipFree(pl->d_vrow_ptr); pl->d_vrow_ptr = nullptr; }
        if (pl->d_pcol) { (void) hipFree(pl->d_pcol); pl->d_pcol = nullptr; }
        if (pl->d_pval) { (void) hipFree(pl->d_pval); pl->d_pval = nullptr; }
        return e != hipSuccess ? fail_hip(e, "column panels") : rc;
    }
    pl->pinfo = pl->inner->pinfo;
    int most = 0;
    for (int k = 0; k < 8; ++k)
        most = std::max(most, pl->pinfo.first[k + 1] - pl->pinfo.first[k]);
    pl->panel_blocks = (most + 3) / 4;
    pl->panels_from_col = d_column_index;
    pl->panels_from_val = d_value;
    pl->meta_bytes += (size_t) (vrows + 1) * sizeof(int32_t) + (size_t) pl->nnz * 12 + pl->inner->meta_bytes;
    // what the panel copy streams: its own tiles, with y counted once per row and panel that has entries
    {
        long long nonempty = 0;
        for (long long v = 0; v < vrows; ++v)
            nonempty += vrow_ptr[(size_t) v + 1] > vrow_ptr[(size_t) v];
        pl->streamed_bytes = pl->inner->streamed_bytes - 16LL * vrows + 16LL * nonempty;
        pl->shifted_entries = pl->inner->shifted_entries;
        pl->narrow_entries = pl->inner->narrow_entries;
        pl->uniform_rows = 0;
    }
    return SPMV_HIP_OK;
}

// The dictionary launch's own descriptor array: consecutive constant-row tiles (kTileMetaValueRows) of one stencil -- same row
// length, same first-row columns relative to the first row (one pattern) -- are re-cut into tiles of 128 rows.  Such a tile reads
// nothing but its first row's columns / index bytes, x and y, so the 512-entry limit of the LDS slice does not bind it, and with
// two adjacent rows per lane 128 rows keep all 64 lanes busy (a 5-point tile of 512 entries has 102 rows: 51 lanes).  Every
// other tile is copied as it is.  Only tiles that refer to a shared pattern are merged (their first-row columns are known to be
// the same without looking at the column array).
static hipError_t merge_constant_row_tiles(spmv_hip_plan * pl, const uint8_t * d_same_prev)
{
#if defined(SPMV_HIP_EXPERIMENTS) && defined(SPMV_VI_ABLATE)
    return hipSuccess; // the ablation builds switch the constant-row path off: every tile has to fit the general path
#endif
    std::vector<int4> d((size_t) pl->ntiles + 1);
    hipError_t e = hipMemcpy(d.data(), pl->d_tiles, d.size() * sizeof(int4), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return e;
    // same_prev[w]: tile w's first row carries the dictionary bytes of tile w - 1's (value_rows_mark_kernel, byte by byte)
    std::vector<uint8_t> same_prev((size_t) pl->ntiles, 0);
    e = hipMemcpy(same_prev.data(), d_same_prev, same_prev.size(), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return e;
    std::vector<int32_t> pat;
    if (pl->d_patterns && pl->npatterns > 0) {
        pat.resize((size_t) pl->npatterns * spmv::kPatStride);
        e = hipMemcpy(pat.data(), pl->d_patterns, pat.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return e;
    }
    const int want = spmv::kTileMetaValueRows | spmv::kTileMetaPattern | spmv::kTileMetaFast | spmv::kTileMetaUniform | spmv::kTileMetaShifted;
    auto mergeable = [&](int w) {
        return !(d[(size_t) w].x & spmv::kTileFlagPartial) && (d[(size_t) w].z & want) == want && d[(size_t) w].w >= 0
            && d[(size_t) w].w < pl->npatterns;
    };
    auto same_stencil = [&](int a, int b) { // same row length and the same first-row columns relative to the first row
        const int len = d[(size_t) a].z & 0xFFFF;
        if (len != (d[(size_t) b].z & 0xFFFF))
            return false;
        const int32_t * pa = pat.data() + (size_t) d[(size_t) a].w * spmv::kPatStride + spmv::kPatRel;
        const int32_t * pb = pat.data() + (size_t) d[(size_t) b].w * spmv::kPatStride + spmv::kPatRel;
        return d[(size_t) a].w == d[(size_t) b].w || std::equal(pa, pa + len, pb);
    };
    std::vector<int4> out;
    out.reserve(d.size());
    int merged_runs = 0;
    for (int w = 0; w < pl->ntiles;) {
        if (!mergeable(w)) {
            out.push_back(d[(size_t) w++]);
            continue;
        }
        int e_run = w + 1;
        // (a run also ends where the COEFFICIENTS change: two constant-row tiles next to each other may carry different sets --
        // a piecewise-constant stencil whose jump falls on a tile boundary -- and a re-cut tile multiplies all its rows with
        // its own first row's)
        while (e_run < pl->ntiles && mergeable(e_run) && same_stencil(w, e_run) && same_prev[(size_t) e_run])
            ++e_run;
        const int len = d[(size_t) w].z & 0xFFFF;
        const long long r_begin = d[(size_t) w].x, r_end = d[(size_t) e_run].x & 0x7FFFFFFF, k_begin = d[(size_t) w].y;
        if (e_run - w < 2 || (long long) d[(size_t) e_run].y - k_begin != (r_end - r_begin) * len) { // (uniform rows: cannot fail)
            for (; w < e_run; ++w)
                out.push_back(d[(size_t) w]);
            continue;
        }
        ++merged_runs;
        // n tiles of as equal a number of rows as possible, none above 128: a run of 130 rows becomes 65 + 65, not 128 + 2
        // (rows longer than the one-lane-per-row limit need kConstantRowMinRows rows per tile to take the constant-row path at
        // all, and a shorter re-cut tile of such rows would not even fit the general path's LDS slice: runs that cannot give
        // every tile that many rows are left alone)
        const long long run_rows = r_end - r_begin;
        const long long n = (run_rows + 127) / 128;
        if (len > spmv::kLanePerRowMaxLen && run_rows / n < spmv::kConstantRowMinRows) {
            --merged_runs;
            for (; w < e_run; ++w)
                out.push_back(d[(size_t) w]);
            continue;
        }
        long long r = r_begin;
        for (long long i = 0; i < n; ++i) {
            out.push_back(make_int4((int) r, (int) (k_begin + (r - r_begin) * len), d[(size_t) w].z, d[(size_t) w].w));
            r += run_rows / n + (i < run_rows % n ? 1 : 0);
        }
        w = e_run;
    }
    if (merged_runs == 0 || out.size() >= (size_t) pl->ntiles)
        return hipSuccess; // nothing gained: the launch keeps the plan's tiles
    const int n = (int) out.size();
    out.push_back(d[(size_t) pl->ntiles]); // the sentinel: {rows, nnz}
    e = hipMalloc((void **) &pl->d_tiles_vi, out.size() * sizeof(int4));
    if (e == hipSuccess)
        e = hipMemcpy(pl->d_tiles_vi, out.data(), out.size() * sizeof(int4), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl->d_tiles_vi) (void) hipFree(pl->d_tiles_vi);
        pl->d_tiles_vi = nullptr;
        return e;
    }
    pl->ntiles_vi = n;
    return hipSuccess;
}

static void drop_value_dictionary(spmv_hip_plan * pl)
{
    if (pl->d_vidx) (void) hipFree(pl->d_vidx);
    if (pl->d_vtab) (void) hipFree(pl->d_vtab);
    if (pl->d_tiles_vi) (void) hipFree(pl->d_tiles_vi);
    pl->d_vidx = nullptr;
    pl->d_vtab = nullptr;
    pl->d_tiles_vi = nullptr;
    pl->ntiles_vi = 0;
    pl->nvalues = 0;
    pl->values_from = nullptr;
    pl->verify_values_pending = false;
}

int spmv_hip_plan_csr_index_values(spmv_hip_plan * pl, const double * d_value, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t before = (pl->d_vidx ? (size_t) pl->nnz + 64 + spmv::kMaxIndexedValues * sizeof(double) : 0);
    drop_value_dictionary(pl);
    pl->values_wanted = d_value;
    pl->meta_bytes -= std::min(pl->meta_bytes, before);
    // only the default kernel reads the dictionary (row-owned wave tiles with 16-bit-capable plans, x below 4 GiB)
    // (not with column panels, block windows or a majority of x-window tiles: those launches have their own variants)
    // (a plan whose launch would stage x through LDS gives that up for the dictionary launch: one byte instead of eight per entry --
    // or none at all in constant-row tiles -- is worth more than the window: 27-point stencil with 100 distinct values 196.7 ->
    // 175.6 us, 27 diagonals 180.3 -> 129.5 us, constant coefficients 187.5 -> 43.1 us; round 3, tools/constant_stencil.py)
    const bool other_variant = pl->inner || pl->d_blocks || pl->d_segblocks;
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->tile != 512 || pl->nnz == 0 || pl->ntiles == 0
        || (pl->balanced && (pl->flags & SPMV_HIP_FLAG_XCD_REMAP)) /* the balanced-tile kernel's dictionary variant has no XCD deal */
        || !pl->d_col16 || other_variant || pl->cols >= (1 << 29)
        || (pl->flags & SPMV_HIP_FLAG_NO_VALUE_INDEX))
        return (pl->inner || before == 0) ? SPMV_HIP_OK // nothing was dropped: the account of plan / compress still holds
                                          : plan_account(pl, pl->d_col16 != nullptr);
    if (!d_value)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    unsigned long long * d_keys = nullptr;
    int * d_state = nullptr;
    std::vector<unsigned long long> keys((size_t) spmv::kDictSlots, spmv::kDictEmpty);
    int state[2] = {0, 0};
    hipError_t e = hipMalloc((void **) &d_keys, keys.size() * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **) &d_state, sizeof(state));
    if (e == hipSuccess) e = hipMemcpyAsync(d_keys, keys.data(), keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_state, 0, sizeof(state), s);
    // A matrix with more distinct values than the dictionary holds gives itself away within its first few thousand entries:
    // one workgroup looks at a prefix first.  (Without it the whole grid's first entries all insert at once -- half a million
    // threads contending for the table's 1024 slots before anybody sees the verdict: 8.5 ms on a 3 M-entry web graph, 11 ms on
    // the queen-like matrix, for a "no".)
    if (e == hipSuccess && pl->nnz > 16384) {
        hipLaunchKernelGGL(spmv::value_dict_insert_kernel, dim3(1), dim3(256), 0, s, 8192LL, d_value, d_keys, d_state, spmv::kMaxIndexedValues);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(state, d_state, sizeof(state), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess && state[1] == 0) {
        hipLaunchKernelGGL(spmv::value_dict_insert_kernel, dim3((unsigned) grid_for(pl->nnz, kBlock, cu_count() * 8)), dim3(256), 0, s,
                           (long long) pl->nnz, d_value, d_keys, d_state, spmv::kMaxIndexedValues);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(keys.data(), d_keys, keys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(state, d_state, sizeof(state), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int rc = SPMV_HIP_OK;
    if (e == hipSuccess && state[1] == 0 && state[0] >= 1 && state[0] <= spmv::kMaxIndexedValues) {
        // few enough distinct values: sort the bit patterns, index every entry
        std::vector<unsigned long long> table;
        for (unsigned long long k : keys)
            if (k != spmv::kDictEmpty)
                table.push_back(k);
        std::sort(table.begin(), table.end());
        std::vector<double> values((size_t) spmv::kMaxIndexedValues, 0.0);
        for (size_t i = 0; i < table.size(); ++i)
            std::memcpy(&values[i], &table[i], sizeof(double));
        unsigned long long * d_table = nullptr;
        e = hipMalloc((void **) &d_table, (size_t) spmv::kMaxIndexedValues * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void **) &pl->d_vtab, (size_t) spmv::kMaxIndexedValues * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **) &pl->d_vidx, (size_t) pl->nnz + 64);
        if (e == hipSuccess) e = hipMemsetAsync(pl->d_vidx, 0, (size_t) pl->nnz + 64, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(pl->d_vtab, values.data(), values.size() * sizeof(double), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_state, 0, sizeof(state), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spmv::value_index_kernel, dim3((unsigned) grid_for(pl->nnz, kBlock, cu_count() * 16)), dim3(256), 0, s,
                               (long long) pl->nnz, d_value, d_table, (int) table.size(), pl->d_vidx, d_state);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(state, d_state, sizeof(state), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (d_table) (void) hipFree(d_table);
        pl->value_row_tiles = 0;
        if (e == hipSuccess && state[1] == 0) {
            // tiles whose rows all repeat the first row's indices (constant-coefficient stencils) need no index stream
            unsigned long long * d_count = nullptr;
            uint8_t * d_same_prev = nullptr;
            unsigned long long count[2] = {0, 0};
            e = hipMalloc((void **) &d_count, kStripedInts * sizeof(unsigned long long));
            if (e == hipSuccess) e = hipMalloc((void **) &d_same_prev, (size_t) pl->ntiles);
            if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, kStripedInts * sizeof(unsigned long long), s);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::value_rows_mark_kernel, dim3((unsigned) ((pl->ntiles + 3) / 4)), dim3(256), 0, s, pl->ntiles, pl->d_tiles,
                                   pl->d_vidx, spmv::kConstantRowMaxLen, d_count, d_same_prev);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = read_striped(d_count, count, 2, s);
            if (d_count) (void) hipFree(d_count);
            pl->value_row_tiles = (int) count[0];
            if (e == hipSuccess && count[0] > 0)
                e = merge_constant_row_tiles(pl, d_same_prev);
            if (d_same_prev) (void) hipFree(d_same_prev);
        }
        if (e == hipSuccess && state[1] == 0) {
            pl->nvalues = (int) table.size();
            pl->values_from = d_value;
            rc = device_value_checksum(d_value, pl->nnz, &pl->value_checksum, s);
            pl->verify_values_pending = true;
            pl->meta_bytes += (size_t) pl->nnz + 64 + spmv::kMaxIndexedValues * sizeof(double);
        } else {
            drop_value_dictionary(pl); // the values changed between the two passes, or a HIP error
        }
    }
    if (d_keys) (void) hipFree(d_keys);
    if (d_state) (void) hipFree(d_state);
    if (e != hipSuccess) {
        drop_value_dictionary(pl);
        return fail_hip(e, "value dictionary");
    }
    if (rc != SPMV_HIP_OK) {
        drop_value_dictionary(pl);
        return rc;
    }
    if (pl->nvalues == 0 && before == 0)
        return SPMV_HIP_OK; // more than 128 distinct values and nothing dropped: the account is unchanged (no descriptor download)
    return plan_account(pl, pl->d_col16 != nullptr);
}

int spmv_hip_plan_csr_refresh_values(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index,
                                     const double * d_value, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    if (pl->nvalues > 0 || pl->d_vidx) {
        // the dictionary is rebuilt from the new values (and dropped if they are no longer few)
        int rc = spmv_hip_plan_csr_index_values(pl, d_value, stream);
        if (rc != SPMV_HIP_OK)
            return rc;
    }
    if (!pl->inner)
        return SPMV_HIP_OK; // no snapshot: the multiply reads the caller's values
    if (!d_row_ptr || !d_column_index || !d_value)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    if (d_column_index != pl->panels_from_col)
        return fail(SPMV_HIP_ERR_STATE, "the column panels were made from another column array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned) ((pl->rows + 255) / 256);
    hipLaunchKernelGGL(spmv::csr_panel_scatter_kernel, dim3(grid), dim3(256), 0, s, pl->rows, (pl->cols + 7) / 8, d_row_ptr,
                       d_column_index, d_value, pl->d_vrow_ptr, pl->d_pcol, pl->d_pval);
    HIP_TRY(hipGetLastError());
    pl->panels_from_val = d_value;
    return SPMV_HIP_OK;
}

int spmv_hip_plan_info(const spmv_hip_plan * pl, int64_t * out, int n)
{
    if (!pl || !out || n < 0)
        return fail(SPMV_HIP_ERR_INVALID, "plan/out null");
    const int64_t v[38] = {pl->algorithm, pl->lanes_per_row, pl->workgroups, pl->nblk,
                           pl->long_blocks, pl->rows, pl->nnz, (int64_t) pl->meta_bytes, pl->narrow_tiles,
                           pl->uniform_tiles, pl->shifted_tiles, pl->xwin_tiles, pl->blockwin_tiles,
                           pl->inner ? pl->inner->ntiles : 0, pl->streamed_bytes, pl->shifted_entries,
                           pl->narrow_entries, pl->uniform_rows, pl->inner ? 1 : 0, pl->balanced ? 1 : 0, pl->nvalues,
                           pl->segwin_tiles, pl->segwin_slots, pl->nvalues > 0 ? pl->value_row_tiles : 0,
                           pl->nvalues > 0 && pl->d_tiles_vi ? pl->ntiles_vi : 0, pl->nvalues > 0 ? 0 : pl->block_tiles,
                           pl->nvalues > 0 ? 0 : pl->block_entries, pl->nhubs, pl->hub_entries, pl->multi_window_tiles,
                           pl->ngroup_tiles, pl->nvalues > 0 ? 0 : pl->masked_block_tiles, pl->nvalues > 0 ? 0 : pl->masked_block_entries,
                           pl->stencil_mask_tiles, pl->stencil_mask_entries,
                           pl->nvalues > 0 ? 0 : pl->colshare_tiles, pl->nvalues > 0 ? 0 : pl->colshare_entries,
                           pl->colshare_tiles > 0 ? pl->block_hint : 0};
    for (int i = 0; i < n && i < 38; ++i)
        out[i] = v[i];
    return SPMV_HIP_OK;
}

} // extern "C"
