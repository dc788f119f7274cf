// spmv_hip.hip -- implementation of the C ABI declared in include/spmv_hip.h.
//
// Host side of the device library: argument checks, launch plans (row blocks for
// the adaptive CSR kernel), kernel launches, and the Level-1 context that owns
// device copies of A, x, y.  No CPU compute path exists here: if HIP cannot run,
// the entry points return an error.
#include "spmv_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl.so is dlopen'ed by spmv_hip_create_multi when G > 1

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "spmv_kernels.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char * what)
{
    g_last_error = what ? what : "";
    return code;
}

int fail_hip(hipError_t e, const char * call)
{
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%s)", call, hipGetErrorString(e), hipGetErrorName(e));
    g_last_error = buf;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
        return SPMV_HIP_ERR_NO_DEVICE;
    if (e == hipErrorOutOfMemory)
        return SPMV_HIP_ERR_ALLOC;
    return SPMV_HIP_ERR_HIP;
}

} // namespace
int spmv_hip_internal_fail_hip(hipError_t e, const char * call) { return fail_hip(e, call); }
namespace {

#define HIP_TRY(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess)                           \
            return fail_hip(e_, #call);                 \
    } while (0)

// every documented SPMV_HIP_FLAG_* bit; anything else is refused (SPMV_HIP_ERR_INVALID)
constexpr unsigned kKnownFlags = SPMV_HIP_FLAG_XCD_REMAP | SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_BIG_TILE |
    SPMV_HIP_FLAG_NO_INDEX_COMPRESSION | SPMV_HIP_FLAG_COO_KEEP_ORDER | SPMV_HIP_FLAG_READ_ROW_PTR | SPMV_HIP_FLAG_ROWS64 |
    SPMV_HIP_FLAG_ROWS128 | SPMV_HIP_FLAG_ELL_COLUMN_MAJOR | SPMV_HIP_FLAG_NO_SHIFTED_TILES | SPMV_HIP_FLAG_NO_X_WINDOW |
    SPMV_HIP_FLAG_NO_COLUMN_PANELS | SPMV_HIP_FLAG_VERIFY_PLAN | SPMV_HIP_FLAG_NO_BALANCED_TILES | SPMV_HIP_FLAG_NO_RUN_EVENTS | SPMV_HIP_FLAG_NO_VALUE_INDEX |
    SPMV_HIP_FLAG_PEER_GATHER | SPMV_HIP_FLAG_BALANCE_ENTRIES
#ifdef SPMV_HIP_EXPERIMENTS
    | 0x2000u | 0x4000u | 0x30000u // timing experiments of tools/kernel_sweep.py (libspmv_hip_experiments.so only)
#endif
    ;

constexpr int kEllInPlaceMaxLength = 80;

bool aligned16(const void * p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kBlock = 256;
constexpr int kTile = 2048;
// compute units of the current device (MI355X: 256); asked once per device, 256 if the query fails
int cu_count()
{
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void) hipGetLastError();
        return 256;
    }
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void) hipGetLastError();
            n = 256;
        }
        cached[dev] = n;
    }
    return cached[dev];
}
// wavetile: rows longer than kSplitThreshold entries are cut into kSplitChunk-entry
// chunks handled by different waves (each adds its partial sum with one atomic)
constexpr int kSplitThreshold = 2048;
constexpr int kSplitChunk = 1024;

int grid_for(long long work_items, int per_block, int max_blocks = 0)
{
    if (max_blocks <= 0)
        max_blocks = cu_count() * 8;
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int) g;
}

} // namespace

struct spmv_hip_plan {
    int32_t rows = 0, cols = 0, nnz = 0;
    int algorithm = SPMV_HIP_CSR_ADAPTIVE;
    int lanes_per_row = 0;
    unsigned flags = 0;
    int workgroups = 0;
    int nblk = 0;
    int long_blocks = 0;
    int32_t * d_blk_row = nullptr;
    int4 * d_tiles = nullptr; // wavetile descriptors {first row | partial flag, first entry, longest row, log2 lanes/row}
    int ntiles = 0;
    int tile = 0;
    uint16_t * d_col16 = nullptr;       // 16-bit column offsets of the narrow tiles (index compression)
    const int32_t * compressed_from = nullptr; // the column array d_col16 was derived from
    int narrow_tiles = 0;
    int shifted_tiles = 0;
    int xwin_tiles = 0; // tiles whose whole column range fits a 256-entry window of x, or with a window of runs
    int longest_tile_row = 0; // longest row inside a stream tile
    int spread_tiles = 0;     // tiles whose columns reach further than an eighth of the matrix
    // column panels: the plan's own panel-major copy of the matrix, multiplied through `inner`
    spmv_hip_plan * inner = nullptr;     // plan of the 8 * rows virtual rows
    int32_t * d_vrow_ptr = nullptr;      // [8 * rows + 1]
    int32_t * d_pcol = nullptr;          // [nnz]
    double * d_pval = nullptr;           // [nnz]
    const int32_t * panels_from_col = nullptr; // the arrays the copy was made from
    const double * panels_from_val = nullptr;
    spmv::PanelInfo pinfo{};
    int panel_blocks = 0;                // workgroups per panel (grid = 8 * panel_blocks)
    int2 * d_blocks = nullptr; // block windows: {first column, slots} per 16 tiles (csr_blockwin_kernel)
    int nblocks16 = 0;
    int blockwin_tiles = 0;
    int32_t * d_patterns = nullptr; // shared window-of-runs layouts (kernels: kPatStride words each)
    int npatterns = 0;
    int uniform_tiles = 0; // tiles whose rows are all equally long: row_ptr is not read for them
    int split_rows = 0;    // rows cut into chunks that are added to y with atomics
    bool balanced = false; // tiles filled by entries, row sums by segmented reduction (csr_segtile_kernel)
    size_t meta_bytes = 0;
    // what one multiply streams with the tile classes chosen (plan_account): roofline bookkeeping
    long long streamed_bytes = 0, shifted_entries = 0, narrow_entries = 0, uniform_rows = 0;
    // value dictionary (spmv_hip_plan_csr_index_values): one byte per stored entry + the distinct values
    uint8_t * d_vidx = nullptr;
    double * d_vtab = nullptr;            // kMaxIndexedValues doubles
    int nvalues = 0;                      // 0 = no dictionary
    const double * values_from = nullptr; // the value array it was made from
    unsigned long long value_checksum = 0;
    mutable std::atomic<bool> verify_values_pending{false}; // claimed (exchange) by the one multiply that re-checks
    // content guard: checksum of the column array the 16-bit stream and the tile marks were derived from
    unsigned long long column_checksum = 0;
    mutable std::atomic<bool> verify_pending{false}; // the first multiply after compress re-checks the checksum
};

struct spmv_hip_ctx {
    int device = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;     // where everything of this context is enqueued
    hipStream_t own_stream = nullptr; // the stream spmv_hip_create made (destroyed with the context)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int format = 0; // 0 none, 1 csr, 2 coo, 3 ell, 4 hybrid (ell + coo remainder)
    int32_t rows = 0, cols = 0, nnz = 0, row_length = 0, nnz2 = 0;
    int csr_algorithm = SPMV_HIP_CSR_AUTO;
    int csr_lanes = 0;
    spmv_hip_plan * plan = nullptr;
    int32_t *d_ptr = nullptr, *d_idx = nullptr, *d_col = nullptr, *d_col2 = nullptr;
    double *d_val = nullptr, *d_val2 = nullptr, *d_x = nullptr, *d_y = nullptr;
    size_t bytes = 0;
    bool coo_sorted_on_device = false;
    // COO column panels (scattered triplets): the context's panel-major, padded copy
    int32_t *d_prow = nullptr, *d_pcol = nullptr;
    double * d_pval = nullptr;
    spmv::CooPanels coo_panels{};
    int coo_panel_blocks = 0;
    bool ell_as_tiles = false; // ELLPACK runs as uniform CSR tiles (row-major, in place)
    bool as_csr = false;       // COO / hybrid were turned into one row-major matrix on the device: run = the CSR plan
    double * d_flush = nullptr; // scratch of spmv_hip_flush_caches (4 x the Infinity Cache), allocated on first use
    bool ell_in_place_any_length = false; // set by upload_hybrid around its ELLPACK upload: the parts are merged into one
                                          // row-major matrix afterwards, which wants the row-major arrays whatever the row length
    bool y_borrowed = false;   // d_y points into memory owned by a multi-GPU front context
    double * borrowed_y = nullptr;
    // ---- multi-GPU front (spmv_hip_create_multi): parts[g] is an ordinary context on device g that holds
    // the rows [g * chunk, min(rows, (g + 1) * chunk)) of the matrix, a full x, and -- as its y -- slot g of
    // yfull[g], that device's copy of the whole y.  A run multiplies on every device and then gathers the
    // slots with ONE in-place all-gather (RCCL), after which every yfull[g] holds the same y.
    bool multi = false;
    std::vector<spmv_hip_ctx *> parts;
    std::vector<double *> yfull;
    std::vector<hipEvent_t> ev_gather; // recorded after the all-gather on each part's stream
    int32_t chunk = 0;              // doubles per y slot: the longest row block
    std::vector<int32_t> row_begin; // G + 1 block boundaries; block g sits at yfull[.] + g * chunk
    bool packed = true;             // every block but the last fills its slot: yfull IS y (the static rule)
    bool peer_gather = false; // SPMV_HIP_FLAG_PEER_GATHER: slots are pushed to the other devices by a kernel, no RCCL
    void * rccl_lib = nullptr;
    std::vector<ncclComm_t> comms;
    ncclResult_t (*p_all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_group_start)() = nullptr;
    ncclResult_t (*p_group_end)() = nullptr;
    ncclResult_t (*p_comm_destroy)(ncclComm_t) = nullptr;
    const char * (*p_error_string)(ncclResult_t) = nullptr;
};

namespace {

int pick_lanes(double mean_len)
{
    int l = 2;
    while (l < 64 && l < mean_len)
        l *= 2;
    return l;
}

template <int LPR>
void launch_vector(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                   const double * x, double * y, hipStream_t s)
{
    hipLaunchKernelGGL((spmv::csr_vector_kernel<LPR, kBlock>), dim3(pl->workgroups), dim3(kBlock), 0, s,
                       pl->rows, p, j, a, x, y);
}

void free_ctx_matrix(spmv_hip_ctx * c)
{
    if (c->plan) {
        spmv_hip_plan_destroy(c->plan);
        c->plan = nullptr;
    }
    if (c->y_borrowed)
        c->d_y = nullptr;
    void * ptrs[] = {c->d_ptr, c->d_idx, c->d_col, c->d_col2, c->d_val, c->d_val2, c->d_x, c->d_y, c->d_prow, c->d_pcol, c->d_pval};
    for (void * p : ptrs)
        if (p)
            (void) hipFree(p);
    c->d_prow = c->d_pcol = nullptr;
    c->d_pval = nullptr;
    c->coo_panel_blocks = 0;
    c->d_ptr = c->d_idx = c->d_col = c->d_col2 = nullptr;
    c->d_val = c->d_val2 = c->d_x = c->d_y = nullptr;
    c->format = 0;
    c->rows = c->cols = c->nnz = c->row_length = c->nnz2 = 0;
    c->coo_sorted_on_device = false;
    c->ell_as_tiles = false;
    c->as_csr = false;
    c->bytes = 0;
}

// device allocation padded so that 16-byte vector loads at the tail stay inside it
template <typename T>
int dev_alloc(spmv_hip_ctx * c, T ** out, size_t n)
{
    size_t bytes = n * sizeof(T) + 64;
    void * p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail_hip(e, "hipMalloc");
    *out = static_cast<T *>(p);
    c->bytes += bytes;
    return SPMV_HIP_OK;
}

int ctx_common_vectors(spmv_hip_ctx * c)
{
    int rc;
    if ((rc = dev_alloc(c, &c->d_x, (size_t) c->cols)) != 0) return rc;
    HIP_TRY(hipMemsetAsync(c->d_x, 0, (size_t) c->cols * sizeof(double), c->stream));
    if (c->y_borrowed) {
        c->d_y = c->borrowed_y; // a slot of the front context's gathered y (already zeroed there)
        return SPMV_HIP_OK;
    }
    if ((rc = dev_alloc(c, &c->d_y, (size_t) c->rows)) != 0) return rc;
    HIP_TRY(hipMemsetAsync(c->d_y, 0, (size_t) c->rows * sizeof(double), c->stream));
    return SPMV_HIP_OK;
}

} // namespace

extern "C" {

int spmv_hip_version(void) { return SPMV_HIP_VERSION; }

const char * spmv_hip_strerror(int code)
{
    switch (code) {
    case SPMV_HIP_OK: return "success";
    case SPMV_HIP_ERR_INVALID: return "invalid argument";
    case SPMV_HIP_ERR_NO_DEVICE: return "no HIP device available";
    case SPMV_HIP_ERR_HIP: return "HIP runtime error";
    case SPMV_HIP_ERR_ALLOC: return "out of memory";
    case SPMV_HIP_ERR_STATE: return "invalid call sequence";
    case SPMV_HIP_ERR_OVERFLOW: return "Integer overflow when computing number of non-zeros";
    case SPMV_HIP_ERR_ALIGN: return "device pointer is not 16-byte aligned";
    default: return "unknown error";
    }
}

const char * spmv_hip_last_error(void) { return g_last_error.c_str(); }

int spmv_hip_device_count(int * count)
{
    if (!count)
        return fail(SPMV_HIP_ERR_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        n = 0;
    }
    *count = n;
    return SPMV_HIP_OK;
}

} // extern "C"

namespace {

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
    std::vector<int4> d((size_t) pl->ntiles + 1);
    HIP_TRY(hipMemcpy(d.data(), pl->d_tiles, d.size() * sizeof(int4), hipMemcpyDeviceToHost));
    long long bytes = 8LL * pl->cols;
    for (int w = 0; w < pl->ntiles; ++w) {
        const long long entries = (long long) d[(size_t) w + 1].y - d[(size_t) w].y;
        const long long rows = (long long) (d[(size_t) w + 1].x & 0x7FFFFFFF) - (d[(size_t) w].x & 0x7FFFFFFF);
        const int meta = d[(size_t) w].z;
        const bool stream_tile = !(d[(size_t) w].x & 0x80000000) && entries > 0 && (meta & spmv::kTileMetaFast);
        const bool shifted = compressed && stream_tile && (meta & spmv::kTileMetaShifted);
        const bool narrow = compressed && stream_tile && (meta & spmv::kTileMetaNarrow);
        const bool uniform = stream_tile && (meta & spmv::kTileMetaUniform);
        long long col_bytes = 4 * entries;
        if (shifted) {
            col_bytes = (meta & spmv::kTileMetaPattern) ? 0 : 4LL * (meta & 0xFFFF);
            pl->shifted_entries += entries;
        } else if (narrow) {
            col_bytes = 2 * entries;
            pl->narrow_entries += entries;
        }
        if (uniform)
            pl->uniform_rows += rows;
        // with a value dictionary the stream tiles of the default kernel read one byte per entry
        const long long val_bytes = (pl->nvalues > 0 && stream_tile && !pl->balanced) ? entries : 8 * entries;
        bytes += val_bytes + col_bytes + 16 + 16 * rows + (uniform ? 0 : 4 * (rows + 1));
    }
    pl->streamed_bytes = bytes;
    return SPMV_HIP_OK;
}

} // namespace

extern "C" {

/* ================================ Level 2 ======================================= */

static int plan_csr_internal(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p,
                             int algorithm, int lanes_per_row, unsigned flags, int32_t break_rows);

int spmv_hip_plan_csr(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p,
                      int algorithm, int lanes_per_row, unsigned flags)
{
    return plan_csr_internal(out, rows, cols, p, algorithm, lanes_per_row, flags, 0);
}

// break_rows > 0: no tile may contain a row index that is a multiple of break_rows except as its
// first row (column panels: tiles stay inside one panel)
static int plan_csr_internal(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p,
                             int algorithm, int lanes_per_row, unsigned flags, int32_t break_rows)
{
    if (!out)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    *out = nullptr;
    if (rows < 0 || cols < 0 || !p)
        return fail(SPMV_HIP_ERR_INVALID, "rows/cols negative or row_ptr null");
    if (p[0] < 0)
        return fail(SPMV_HIP_ERR_INVALID, "row_ptr[0] is negative");
    for (int32_t r = 0; r < rows; ++r)
        if (p[r + 1] < p[r])
            return fail(SPMV_HIP_ERR_INVALID, "row_ptr is not non-decreasing");
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

    if (algorithm == SPMV_HIP_CSR_SCALAR) {
        pl->workgroups = grid_for(rows, kBlock);
    } else if (algorithm == SPMV_HIP_CSR_VECTOR) {
        pl->lanes_per_row = lanes_per_row ? lanes_per_row : pick_lanes(mean);
        pl->workgroups = grid_for((long long) rows * pl->lanes_per_row, kBlock, cu_count() * 32);
    } else if (algorithm == SPMV_HIP_CSR_WAVETILE) {
        // wave tiles: <= 64 rows and <= tile entries (from the 4-aligned start) per wave
        const int tile = (flags & SPMV_HIP_FLAG_BIG_TILE) ? 1024 : 512;
        const bool exact = (flags & SPMV_HIP_FLAG_EXACT_ORDER) != 0;
        pl->tile = tile;
        std::vector<int4> desc;
        desc.reserve((size_t) rows / 48 + 16);
        int32_t r = 0;
        int next_panel = 0;
        long long stream_tiles = 0, stream_tile_entries = 0; // tiles of whole rows (not long-row tiles) and what they hold
        while (r < rows) {
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
            auto lanes_for = [](int len) {
                int l = 0;
                while (l < 6 && (16 << l) < len)
                    ++l;
                return l;
            };
            while (r1 < rows && (r1 - r) < row_cap && (long long) p[r1 + 1] - kb <= tile) {
                if (break_rows > 0 && r1 > r && r1 % break_rows == 0)
                    break;
                const int len = p[r1 + 1] - p[r1];
                if (!exact && r1 > r) {
                    const int l = lanes_for(std::max(maxlen, len));
                    if (l > 0 && ((r1 - r + 1) << l) > 64)
                        break;
                }
                maxlen = std::max(maxlen, len);
                minlen = std::min(minlen, len);
                ++r1;
            }
            if (r1 == r) { // one row longer than a tile
                const long long len = (long long) p[r + 1] - p[r];
                pl->long_blocks++;
                if (!exact && len > kSplitThreshold) {
                    pl->split_rows++;
                    for (long long k = p[r]; k < p[r + 1]; k += kSplitChunk)
                        desc.push_back(make_int4((int) (r | 0x80000000u), (int) k, 0, 0));
                } else {
                    desc.push_back(make_int4(r, p[r], 0, 0));
                }
                r1 = r + 1;
            } else {
                // one lane per row (rows of <= 16 entries) keeps the reference's summation order
                const int lanes_log2 = exact ? 0 : lanes_for(maxlen);
                pl->longest_tile_row = std::max(pl->longest_tile_row, (int) maxlen);
                // "fast": non-empty, and 16-byte loads of its last quad stay inside the arrays
                const bool fast = p[r1] > p[r] && (((long long) p[r1] - 1) | 3) < (long long) p[rows];
                const bool uniform = minlen == maxlen && !(flags & SPMV_HIP_FLAG_READ_ROW_PTR);
                if (uniform)
                    pl->uniform_tiles++;
                stream_tiles++;
                stream_tile_entries += (long long) p[r1] - p[r];
                desc.push_back(make_int4(r, p[r], maxlen | (lanes_log2 << 16) | (fast ? (1 << 25) : 0) |
                                                      (uniform ? (1 << 26) : 0), 0));
            }
            r = r1;
        }
        // Balanced tiles: when the tiles above come out mostly empty BECAUSE rows are skewed (a long row
        // limits its tile to the rows the wave has lanes for), fill tiles by entries instead -- up to 512 in
        // up to 256 whole rows -- and let csr_segtile_kernel add the rows up by segmented reduction.
        // Regular matrices (every tile already full, or only short rows) keep the tiles above and with
        // them the reference's summation order.
        {
            int longest = 0;
            for (int32_t q = 0; q < rows; ++q)
                longest = std::max(longest, (int) (p[q + 1] - p[q]));
            // (rows with a wave or more to themselves are the same in both tilings and do not count)
            const bool want = !exact && tile == 512 && break_rows == 0 && !(flags & SPMV_HIP_FLAG_NO_BALANCED_TILES)
                && longest > 16 && 2 * stream_tile_entries < stream_tiles * tile;
            if (want) {
                desc.clear();
                pl->uniform_tiles = pl->long_blocks = pl->split_rows = pl->longest_tile_row = 0;
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
                        if (len > kSplitThreshold) {
                            pl->split_rows++;
                            for (long long k = p[r]; k < p[r + 1]; k += kSplitChunk)
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
                delete pl;
                return rc;
            }
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
    if (pl->d_vidx)
        (void) hipFree(pl->d_vidx);
    if (pl->d_vtab)
        (void) hipFree(pl->d_vtab);
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
    const size_t bytes = (size_t) pl->nnz * sizeof(uint16_t) + 64;
    const bool want_patterns = !(pl->flags & SPMV_HIP_FLAG_NO_SHIFTED_TILES);
    int * d_count = nullptr;
    unsigned long long * d_fp = nullptr;
    HIP_TRY(hipMalloc((void **) &pl->d_col16, bytes));
    int counts[5] = {0, 0, 0, 0, 0};
    hipError_t e = hipMalloc((void **) &d_count, sizeof(counts));
    if (e == hipSuccess && want_patterns) {
        e = hipMalloc((void **) &d_fp, (size_t) pl->ntiles * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d_fp, 0, (size_t) pl->ntiles * sizeof(unsigned long long), s);
    }
    if (e == hipSuccess) e = hipMemsetAsync(pl->d_col16, 0, bytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, sizeof(counts), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::csr_tile_compress_kernel, dim3((pl->ntiles + 3) / 4), dim3(256), 0, s,
                           pl->ntiles, pl->tile, pl->d_tiles, d_column_index, pl->d_col16, d_count,
                           (pl->flags & SPMV_HIP_FLAG_NO_SHIFTED_TILES) ? 0 : 1, d_fp, std::max(1, (pl->cols + 7) / 8));
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
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_count, sizeof(counts), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pl->narrow_tiles = counts[0];
    pl->shifted_tiles = counts[1];
    pl->xwin_tiles = counts[2];
    pl->spread_tiles = counts[4];
    // block windows (x staged through LDS per 16 tiles) for what has no cheaper path: first count
    // the tiles that would qualify, and only if they are the majority mark them
    if (e == hipSuccess && pl->tile == 512 && !pl->balanced && pl->ntiles >= 4 * spmv::kBlockWinTiles
        && !(pl->flags & (SPMV_HIP_FLAG_NO_X_WINDOW | SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_XCD_REMAP))) {
        const int nb = (pl->ntiles + spmv::kBlockWinTiles - 1) / spmv::kBlockWinTiles;
        hipLaunchKernelGGL(spmv::csr_blockwin_mark_kernel, dim3(nb), dim3(1024), 0, s, pl->ntiles, pl->tile, pl->d_tiles,
                           pl->d_col16, (int2 *) nullptr, d_count, 0);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(counts, d_count, sizeof(counts), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess && 2 * (long long) counts[3] > pl->ntiles) {
            e = hipMalloc((void **) &pl->d_blocks, (size_t) nb * sizeof(int2));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::csr_blockwin_mark_kernel, dim3(nb), dim3(1024), 0, s, pl->ntiles, pl->tile,
                                   pl->d_tiles, pl->d_col16, pl->d_blocks, d_count, 1);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess) {
                pl->nblocks16 = nb;
                pl->blockwin_tiles = counts[3];
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
        return fail_hip(e, "index compression");
    }
    pl->meta_bytes += bytes;
    pl->compressed_from = d_column_index;
    int rc = device_column_checksum(d_column_index, pl->nnz, &pl->column_checksum, s);
    if (rc == SPMV_HIP_OK)
        rc = plan_account(pl, true);
    pl->verify_pending = true;
    return rc;
}

int spmv_hip_plan_verify(spmv_hip_plan * pl, const int32_t * d_column_index, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    pl->verify_pending = false; // an explicit check stands in for the first multiply's
    return verify_plan(pl, d_column_index, static_cast<hipStream_t>(stream));
}

// defined in coo_sort.hip (hipCUB): out[i] = sum of in[0..i), n elements
int spmv_hip_internal_exclusive_scan_i32(const int32_t * d_in, int32_t * d_out, long long n, hipStream_t s);

int spmv_hip_plan_csr_repack(spmv_hip_plan * pl, const int32_t * d_row_ptr, const int32_t * d_column_index,
                             const double * d_value, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    if (pl->inner)
        return fail(SPMV_HIP_ERR_STATE, "plan is already repacked");
    // column panels pay when x does not fit one XCD's L2 and the columns are scattered; they cost
    // a copy of the matrix, one virtual row per (row, panel) and atomic y updates
    const bool scattered = 2 * (long long) pl->spread_tiles > pl->ntiles && 2 * (long long) pl->shifted_tiles < pl->ntiles;
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->tile != 512 || pl->balanced || pl->nnz == 0 || pl->rows < 1024
        || (pl->flags & (SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_NO_COLUMN_PANELS | SPMV_HIP_FLAG_XCD_REMAP))
        || !pl->d_col16 /* not compressed: the tile classes are unknown */
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

static void drop_value_dictionary(spmv_hip_plan * pl)
{
    if (pl->d_vidx) (void) hipFree(pl->d_vidx);
    if (pl->d_vtab) (void) hipFree(pl->d_vtab);
    pl->d_vidx = nullptr;
    pl->d_vtab = nullptr;
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
    pl->meta_bytes -= std::min(pl->meta_bytes, before);
    // only the default kernel reads the dictionary (row-owned wave tiles with 16-bit-capable plans, x below 4 GiB)
    // (not with column panels, block windows or a majority of x-window tiles: those launches have their own variants)
    const bool other_variant = pl->inner || pl->d_blocks
        || (!(pl->flags & SPMV_HIP_FLAG_NO_X_WINDOW) && 2 * (long long) pl->xwin_tiles > pl->ntiles);
    if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->tile != 512 || pl->nnz == 0 || pl->ntiles == 0 || pl->balanced
        || !pl->d_col16 || other_variant || pl->cols >= (1 << 29)
        || (pl->flags & SPMV_HIP_FLAG_NO_VALUE_INDEX))
        return pl->inner ? SPMV_HIP_OK : plan_account(pl, pl->d_col16 != nullptr);
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
    if (e == hipSuccess) {
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
    const int64_t v[21] = {pl->algorithm, pl->lanes_per_row, pl->workgroups, pl->nblk,
                           pl->long_blocks, pl->rows, pl->nnz, (int64_t) pl->meta_bytes, pl->narrow_tiles,
                           pl->uniform_tiles, pl->shifted_tiles, pl->xwin_tiles, pl->blockwin_tiles,
                           pl->inner ? pl->inner->ntiles : 0, pl->streamed_bytes, pl->shifted_entries,
                           pl->narrow_entries, pl->uniform_rows, pl->inner ? 1 : 0, pl->balanced ? 1 : 0, pl->nvalues};
    for (int i = 0; i < n && i < 21; ++i)
        out[i] = v[i];
    return SPMV_HIP_OK;
}

int spmv_hip_csr_spmv(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j,
                      const double * a, const double * x, double * y, void * stream)
{
    return spmv_hip_csr_spmv_out(pl, p, j, a, x, y, y, stream);
}

int spmv_hip_csr_spmv_out(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                          const double * x, const double * y_in, double * y, void * stream)
{
    if (!pl)
        return fail(SPMV_HIP_ERR_INVALID, "plan is null");
    if (pl->rows == 0)
        return SPMV_HIP_OK;
    if (!p || !y || !y_in || (pl->nnz > 0 && (!j || !a || !x)))
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    if (!aligned16(j) || !aligned16(a))
        return fail(SPMV_HIP_ERR_ALIGN, "column_index/value must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool panels_now = pl->algorithm == SPMV_HIP_CSR_WAVETILE && pl->inner && pl->panels_from_col == j
        && pl->panels_from_val == a && pl->panel_blocks > 0;
    if (y_in != y) {
        const double * lo = y_in < y ? y_in : y, * hi = y_in < y ? y : y_in;
        if (lo + pl->rows > hi)
            return fail(SPMV_HIP_ERR_INVALID, "y_in and y_out overlap");
        // kernels that read y_in and write y_out row by row take the pair as it is; the others
        // (atomic partial sums: split long rows, column panels; the non-default algorithms) get
        // y_out = y_in first and then accumulate in place
        if (pl->algorithm != SPMV_HIP_CSR_WAVETILE || pl->split_rows > 0 || panels_now) {
            HIP_TRY(hipMemcpyAsync(y, y_in, (size_t) pl->rows * sizeof(double), hipMemcpyDeviceToDevice, s));
            y_in = y;
        }
    }
    // One-time content check of the first multiply after compress / index_values (and every multiply under
    // SPMV_HIP_FLAG_VERIFY_PLAN): a checksum pass + hipStreamSynchronize, documented in spmv_hip.h.  The pending
    // marks are atomics claimed by exchange, so two host threads sharing a plan do not both run it, and the plan is
    // not otherwise modified here; while the stream is being captured into a graph the check is left pending
    // (a synchronize would invalidate the capture).
    const bool every = (pl->flags & SPMV_HIP_FLAG_VERIFY_PLAN) != 0;
    if (every || pl->verify_pending.load(std::memory_order_relaxed) || pl->verify_values_pending.load(std::memory_order_relaxed)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
            (void) hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        if (cap == hipStreamCaptureStatusNone) {
            if (pl->verify_pending.exchange(false) || every) {
                int rc = verify_plan(pl, j, s);
                if (rc != SPMV_HIP_OK)
                    return rc;
            }
            if (pl->verify_values_pending.exchange(false) || (every && pl->nvalues > 0)) {
                int rc = verify_plan_values(pl, a, s);
                if (rc != SPMV_HIP_OK)
                    return rc;
            }
        }
    }
    switch (pl->algorithm) {
    case SPMV_HIP_CSR_SCALAR:
        hipLaunchKernelGGL((spmv::csr_scalar_kernel<kBlock>), dim3(pl->workgroups), dim3(kBlock), 0, s,
                           pl->rows, p, j, a, x, y);
        break;
    case SPMV_HIP_CSR_VECTOR:
        switch (pl->lanes_per_row) {
        case 2: launch_vector<2>(pl, p, j, a, x, y, s); break;
        case 4: launch_vector<4>(pl, p, j, a, x, y, s); break;
        case 8: launch_vector<8>(pl, p, j, a, x, y, s); break;
        case 16: launch_vector<16>(pl, p, j, a, x, y, s); break;
        case 32: launch_vector<32>(pl, p, j, a, x, y, s); break;
        default: launch_vector<64>(pl, p, j, a, x, y, s); break;
        }
        break;
    case SPMV_HIP_CSR_WAVETILE:
        if (panels_now) {
            // column panels: the plan's panel-major copy, one panel per XCD label, atomic partial sums
            const spmv_hip_plan * in = pl->inner;
            const bool x32 = pl->cols < (1 << 29);
            const dim3 grid((unsigned) (8 * pl->panel_blocks));
            if (x32)
                hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, true>), grid, dim3(256), 0, s, in->ntiles,
                                   in->d_tiles, pl->d_vrow_ptr, pl->d_pcol, in->d_col16, pl->d_pval, x, y, y, pl->nnz, pl->cols, 0,
                                   in->d_patterns, pl->pinfo);
            else
                hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, false, false, 0, 0, true>), grid, dim3(256), 0, s, in->ntiles,
                                   in->d_tiles, pl->d_vrow_ptr, pl->d_pcol, in->d_col16, pl->d_pval, x, y, y, pl->nnz, pl->cols, 0,
                                   in->d_patterns, pl->pinfo);
        } else if (pl->balanced && pl->ntiles > 0) {
            // tiles filled by entries, row sums by segmented reduction (skewed rows)
            const bool c16 = pl->d_col16 != nullptr && pl->compressed_from == j;
            const bool x32 = pl->cols < (1 << 29);
            const dim3 grid((unsigned) pl->workgroups);
            const bool xcd = (pl->flags & SPMV_HIP_FLAG_XCD_REMAP) != 0;
#define SPMV_SEG_LAUNCH(C, X, R) \
    hipLaunchKernelGGL((spmv::csr_segtile_kernel<C, X, R>), grid, dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols)
#define SPMV_SEG_X(C, R) do { if (x32) SPMV_SEG_LAUNCH(C, true, R); else SPMV_SEG_LAUNCH(C, false, R); } while (0)
#define SPMV_SEG_C(R) do { if (c16) SPMV_SEG_X(true, R); else SPMV_SEG_X(false, R); } while (0)
            if (xcd) SPMV_SEG_C(true); else SPMV_SEG_C(false);
#undef SPMV_SEG_C
#undef SPMV_SEG_X
#undef SPMV_SEG_LAUNCH
        } else if (pl->ntiles > 0) {
            const int xcd = (pl->flags & SPMV_HIP_FLAG_XCD_REMAP) ? 1 : 0;
            const int exact = (pl->flags & SPMV_HIP_FLAG_EXACT_ORDER) ? 1 : 0;
            // the 16-bit index stream is only valid for the column array it was derived from
            const bool c16 = pl->d_col16 != nullptr && pl->compressed_from == j;
            // x below 4 GiB: 32-bit gather offsets from a scalar base
            const bool x32 = pl->cols < (1 << 29);
#define SPMV_WT_LAUNCH(T, C, X, R)                                                                    \
    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<T, C, X, R>), dim3(pl->workgroups), dim3(256), 0, s, \
                       pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{})
#define SPMV_WT_X(T, C, R)  do { if (x32) SPMV_WT_LAUNCH(T, C, true, R); else SPMV_WT_LAUNCH(T, C, false, R); } while (0)
#define SPMV_WT_C(T, R)     do { if (c16) SPMV_WT_X(T, true, R); else SPMV_WT_X(T, false, R); } while (0)
#ifdef SPMV_HIP_EXPERIMENTS
            const int abl = (int) ((pl->flags >> 16) & 3); // timing experiments (kernel_sweep.py): wrong results by design
#endif
            // every tile belongs to the block-window kernel below: nothing for this launch to do
            const bool all_blockwin = c16 && pl->d_blocks && pl->blockwin_tiles == pl->ntiles;
            if (all_blockwin) {
            } else
            // x staged through LDS when most tiles have a window.  With one lane per row (EXACT_ORDER,
            // the in-place ELLPACK path) long row sums want the occupancy more than the gather wants
            // the window (L = 81: 339 vs 333 us; L = 27: 199 vs 223 us), so only up to 32 entries per row
            if (!(pl->flags & SPMV_HIP_FLAG_NO_X_WINDOW) && (!exact || pl->longest_tile_row <= 32) && c16 && x32
                && pl->tile == 512 && !xcd && 2 * (long long) pl->xwin_tiles > pl->ntiles) {
                hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 256>), dim3(pl->workgroups), dim3(256), 0, s,
                                   pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{});
            }
#ifdef SPMV_HIP_EXPERIMENTS
            else if (abl && c16 && x32 && pl->tile == 512) {
                if (abl == 1) hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 1>), dim3(pl->workgroups), dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{});
                else if (abl == 2) hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 2>), dim3(pl->workgroups), dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{});
                else hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 3>), dim3(pl->workgroups), dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{});
            }
#endif
            else if (pl->tile == 1024) {
                if (xcd) SPMV_WT_C(1024, true); else SPMV_WT_C(1024, false);
            } else if (c16 && x32 && pl->nvalues > 0 && pl->values_from == a) {
                // the default kernel with the value dictionary: one byte per entry instead of eight
                if (xcd)
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, true, 0, 0, false, true>), dim3(pl->workgroups), dim3(256), 0, s,
                                       pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, pl->d_vidx, pl->d_vtab, pl->nvalues);
                else
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, true>), dim3(pl->workgroups), dim3(256), 0, s,
                                       pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, pl->d_vidx, pl->d_vtab, pl->nvalues);
            } else {
                if (xcd) SPMV_WT_C(512, true); else SPMV_WT_C(512, false);
            }
#undef SPMV_WT_C
#undef SPMV_WT_X
#undef SPMV_WT_LAUNCH
            // the tiles marked for a block window were skipped above (only when the 16-bit column
            // stream is valid for this column array, like the marks themselves)
            if (c16 && pl->d_blocks) {
#ifdef SPMV_HIP_EXPERIMENTS
                if (pl->flags & 0x2000u) { // tools/kernel_sweep.py: one workgroup per block, no sliding window
                    hipLaunchKernelGGL((spmv::csr_blockwin_kernel<512>), dim3(pl->nblocks16), dim3(1024), 0, s, pl->ntiles,
                                       pl->d_tiles, pl->d_blocks, p, pl->d_col16, a, x, y_in, y);
                } else
#endif
                {
                    // persistent workgroups, one per CU, each walking through consecutive blocks
                    const int groups = std::min(pl->nblocks16, cu_count());
                    const int per_group = (pl->nblocks16 + groups - 1) / groups;
                    hipLaunchKernelGGL((spmv::csr_blockwin_stream_kernel<512>), dim3(groups), dim3(1024), 0, s, pl->ntiles,
                                       pl->nblocks16, per_group, pl->d_tiles, pl->d_blocks, p, pl->d_col16, a, x, y_in, y);
                }
            }
        }
        break;
    default:
        if (pl->nblk > 0)
            hipLaunchKernelGGL((spmv::csr_adaptive_kernel<kBlock, kTile>), dim3(pl->nblk), dim3(kBlock), 0, s,
                               pl->nblk, pl->d_blk_row, p, j, a, x, y, pl->nnz,
                               (pl->flags & SPMV_HIP_FLAG_XCD_REMAP) ? 1 : 0,
                               (pl->flags & SPMV_HIP_FLAG_EXACT_ORDER) ? 1 : 0);
        break;
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

#ifdef SPMV_HIP_EXPERIMENTS
/* libspmv_hip_experiments.so only (tools/kernel_sweep.py): 0 = default choice, 1 = always the
 * 64-entries-per-wave kernel. */
static int g_coo_variant = 0;
void spmv_hip_coo_variant(int variant) { g_coo_variant = variant; }
#else
static const int g_coo_variant = 0;
#endif

int spmv_hip_coo_spmv(int32_t rows, int32_t nnz, const int32_t * ri, const int32_t * ci,
                      const double * v, const double * x, double * y, void * stream)
{
    if (rows < 0 || nnz < 0)
        return fail(SPMV_HIP_ERR_INVALID, "negative size");
    if (nnz == 0 || rows == 0)
        return SPMV_HIP_OK;
    if (!ri || !ci || !v || !x || !y)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (aligned16(ri) && aligned16(ci) && aligned16(v) && g_coo_variant == 0) {
        // 16-byte loads, 256 entries per wave
        const unsigned grid = (unsigned) (((long long) nnz + 1023) / 1024);
        hipLaunchKernelGGL((spmv::coo_wide_kernel<false>), dim3(grid), dim3(256), 0, s, nnz, ri, ci, v, x, y, spmv::CooPanels{});
    } else {
        const int grid = grid_for(nnz, kBlock, cu_count() * 16);
        hipLaunchKernelGGL((spmv::coo_kernel<kBlock>), dim3(grid), dim3(kBlock), 0, s, nnz, ri, ci, v, x, y);
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

int spmv_hip_ell_to_column_major(int32_t rows, int32_t row_length, const int32_t * j_rm,
                                 const double * a_rm, int32_t * j_cm, double * a_cm, void * stream)
{
    if (rows < 0 || row_length < 0)
        return fail(SPMV_HIP_ERR_INVALID, "negative size");
    int32_t n;
    if (__builtin_mul_overflow(rows, row_length, &n))
        return fail(SPMV_HIP_ERR_OVERFLOW, "rows*row_length overflows int32");
    if (n == 0)
        return SPMV_HIP_OK;
    if (!j_rm || !a_rm || !j_cm || !a_cm)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n, kBlock, cu_count() * 16);
    hipLaunchKernelGGL((spmv::ell_transpose_kernel<kBlock>), dim3(grid), dim3(kBlock), 0, s, rows,
                       row_length, j_rm, a_rm, j_cm, a_cm);
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

int spmv_hip_ell_spmv(int32_t rows, int32_t row_length, const int32_t * j, const double * a,
                      const double * x, double * y, void * stream)
{
    if (rows < 0 || row_length < 0)
        return fail(SPMV_HIP_ERR_INVALID, "negative size");
    int32_t n;
    if (__builtin_mul_overflow(rows, row_length, &n))
        return fail(SPMV_HIP_ERR_OVERFLOW, "rows*row_length overflows int32");
    if (rows == 0)
        return SPMV_HIP_OK;
    if (!y || (n > 0 && (!j || !a || !x)))
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(rows, kBlock, cu_count() * 16);
    hipLaunchKernelGGL((spmv::ell_kernel<kBlock>), dim3(grid), dim3(kBlock), 0, s, rows, row_length, j, a, x, y);
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

int spmv_hip_triad(int64_t n, double * a, const double * b, const double * c, double q, void * stream)
{
    if (n < 0)
        return fail(SPMV_HIP_ERR_INVALID, "negative size");
    if (n == 0)
        return SPMV_HIP_OK;
    if (!a || !b || !c)
        return fail(SPMV_HIP_ERR_INVALID, "null device pointer");
    if (!aligned16(a) || !aligned16(b) || !aligned16(c))
        return fail(SPMV_HIP_ERR_ALIGN, "triad arrays must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one 16-byte element per lane over a flat grid, non-temporal stores: measured 6.15 TB/s against
    // 4.9 TB/s for a grid-stride loop with 4 loads in flight (profiles/r01_triad_variants.log)
    const long long n2 = n / 2;
    if (n2 > 0) {
        const long long grid = (n2 + kBlock - 1) / kBlock;
        hipLaunchKernelGGL((spmv::triad_flat_kernel<kBlock, true>), dim3((unsigned) grid), dim3(kBlock), 0, s, n2, a, b, c, q);
    }
    if (n & 1)
        hipLaunchKernelGGL((spmv::triad_kernel<64, 1>), dim3(1), dim3(64), 0, s, 1LL, a + (n - 1), b + (n - 1), c + (n - 1), q);
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

#ifdef SPMV_HIP_EXPERIMENTS
/* Not in the header: A/B variants of the triad for tools/kernel_sweep.py.
 * 0 = grid-stride unroll 4 on 8 workgroups per CU, 1 = one element per lane (flat grid),
 * 2 = flat + non-temporal stores (the shipped kernel),
 * 3 = grid-stride unroll 4 on 16 workgroups per CU, 4 = unroll 8. */
int spmv_hip_triad_variant(int64_t n, double * a, const double * b, const double * c, double q,
                           void * stream, int variant)
{
    if (n & 1)
        return spmv_hip_triad(n, a, b, c, q, stream);
    if (variant == 0) {
        hipLaunchKernelGGL((spmv::triad_kernel<kBlock, 4>), dim3(cu_count() * 8), dim3(kBlock), 0, static_cast<hipStream_t>(stream), (long long) n, a, b, c, q);
        HIP_TRY(hipGetLastError());
        return SPMV_HIP_OK;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n2 = n / 2;
    if (variant == 1 || variant == 2) {
        const long long grid = (n2 + kBlock - 1) / kBlock;
        if (variant == 1)
            hipLaunchKernelGGL((spmv::triad_flat_kernel<kBlock, false>), dim3((unsigned) grid), dim3(kBlock), 0, s, n2, a, b, c, q);
        else
            hipLaunchKernelGGL((spmv::triad_flat_kernel<kBlock, true>), dim3((unsigned) grid), dim3(kBlock), 0, s, n2, a, b, c, q);
    } else if (variant == 3) {
        hipLaunchKernelGGL((spmv::triad_kernel<kBlock, 4>), dim3(cu_count() * 16), dim3(kBlock), 0, s, (long long) n, a, b, c, q);
    } else {
        hipLaunchKernelGGL((spmv::triad_kernel<kBlock, 8>), dim3(cu_count() * 8), dim3(kBlock), 0, s, (long long) n, a, b, c, q);
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}
#endif

} // extern "C"
namespace {
void multi_free_matrix(spmv_hip_ctx * c);
int multi_upload_csr(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_ptr,
                     const int32_t * column_index, const double * value);
int multi_upload_ell(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length, const int32_t * column_index, const double * value);
int multi_upload_coo(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_index, const int32_t * column_index,
                     const double * value);
int multi_set_x(spmv_hip_ctx * c, const double * x);
int multi_set_y(spmv_hip_ctx * c, const double * y);
int multi_get_y(spmv_hip_ctx * c, double * y);
int multi_run(spmv_hip_ctx * c);
int multi_sync(spmv_hip_ctx * c);
int multi_times(spmv_hip_ctx * c, uint64_t * kernel_ns, uint64_t * gather_ns);
} // namespace
extern "C" {

/* ================================ Level 1 ======================================= */

int spmv_hip_create(spmv_hip_ctx ** out, int device, unsigned flags)
{
    if (!out)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    if (device < 0 || device >= n)
        return fail(SPMV_HIP_ERR_INVALID, "device index out of range");
    if (flags & ~kKnownFlags)
        return fail(SPMV_HIP_ERR_INVALID, "unknown flag bits");
    HIP_TRY(hipSetDevice(device));
    spmv_hip_ctx * c = new (std::nothrow) spmv_hip_ctx;
    if (!c)
        return fail(SPMV_HIP_ERR_ALLOC, "ctx allocation failed");
    c->device = device;
    c->flags = flags;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    c->stream = c->own_stream;
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        int rc = fail_hip(e, "stream/event creation");
        spmv_hip_destroy(c);
        return rc;
    }
    *out = c;
    return SPMV_HIP_OK;
}

void spmv_hip_destroy(spmv_hip_ctx * c)
{
    if (!c)
        return;
    if (c->multi) {
        multi_free_matrix(c);
        for (size_t g = 0; g < c->comms.size(); ++g)
            if (c->comms[g] && c->p_comm_destroy)
                (void) c->p_comm_destroy(c->comms[g]);
        for (hipEvent_t ev : c->ev_gather)
            if (ev)
                (void) hipEventDestroy(ev);
        for (spmv_hip_ctx * part : c->parts)
            spmv_hip_destroy(part);
        // librccl.so stays loaded (dlclose of a library with live device state is not safe)
        delete c;
        return;
    }
    (void) hipSetDevice(c->device);
    if (c->own_stream)
        (void) hipStreamSynchronize(c->stream);
    free_ctx_matrix(c);
    if (c->d_flush) (void) hipFree(c->d_flush);
    if (c->ev0) (void) hipEventDestroy(c->ev0);
    if (c->ev1) (void) hipEventDestroy(c->ev1);
    if (c->own_stream) (void) hipStreamDestroy(c->own_stream);
    delete c;
}

int spmv_hip_set_stream(spmv_hip_ctx * c, void * stream, int use_own)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return fail(SPMV_HIP_ERR_STATE, "a multi-GPU context runs on its own per-device streams");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream)); // nothing of the old stream is left behind
    c->stream = use_own ? c->own_stream : static_cast<hipStream_t>(stream);
    return SPMV_HIP_OK;
}

int spmv_hip_set_csr_algorithm(spmv_hip_ctx * c, int algorithm, int lanes_per_row)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (algorithm < SPMV_HIP_CSR_AUTO || algorithm > SPMV_HIP_CSR_WAVETILE)
        return fail(SPMV_HIP_ERR_INVALID, "unknown CSR algorithm");
    c->csr_algorithm = algorithm;
    c->csr_lanes = lanes_per_row;
    return SPMV_HIP_OK;
}

int spmv_hip_upload_csr(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz,
                        const int32_t * row_ptr, const int32_t * column_index, const double * value)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return multi_upload_csr(c, rows, cols, nnz, row_ptr, column_index, value);
    if (rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!column_index || !value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad CSR arguments");
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz)
        return fail(SPMV_HIP_ERR_INVALID, "row_ptr[0] must be 0 and row_ptr[rows] must equal nnz");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_ctx_matrix(c);
    int rc = spmv_hip_plan_csr(&c->plan, rows, cols, row_ptr, c->csr_algorithm, c->csr_lanes, c->flags);
    if (rc != 0)
        return rc;
    c->rows = rows;
    c->cols = cols;
    c->nnz = nnz;
    if ((rc = dev_alloc(c, &c->d_ptr, (size_t) rows + 1)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_col, (size_t) nnz)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_val, (size_t) nnz)) != 0) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_ptr, row_ptr, ((size_t) rows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_col, column_index, (size_t) nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_val, value, (size_t) nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = ctx_common_vectors(c)) != 0) return rc;
    // the column indices are range-checked where they now are, at HBM speed (row_ptr was checked by
    // the plan builder); a bad file must not become an out-of-bounds gather
    bool bad = false;
    if ((rc = device_index_check(c->d_col, nnz, cols, false, &bad, nullptr, c->stream)) != 0) return rc;
    if (bad) {
        free_ctx_matrix(c);
        return fail(SPMV_HIP_ERR_INVALID, "column index out of range");
    }
    if (!(c->flags & SPMV_HIP_FLAG_NO_INDEX_COMPRESSION)) {
        if ((rc = spmv_hip_plan_csr_compress(c->plan, c->d_col, c->stream)) != 0) return rc;
        // scattered columns and an x that does not fit one XCD's L2: column panels (the context owns
        // the arrays, so the snapshot of the values cannot go stale)
        if ((rc = spmv_hip_plan_csr_repack(c->plan, c->d_ptr, c->d_col, c->d_val, c->stream)) != 0) return rc;
        // the context owns the device copy of the values, so a value dictionary cannot go stale
        if ((rc = spmv_hip_plan_csr_index_values(c->plan, c->d_val, c->stream)) != 0) return rc;
    }
    c->format = 1;
    return SPMV_HIP_OK;
}

// defined in coo_sort.hip
int spmv_hip_internal_coo_panels(int32_t cols, int32_t nnz, const int32_t * d_row, const int32_t * d_col, const double * d_val,
                                 int32_t ** out_row, int32_t ** out_col, double ** out_val, long long * start, hipStream_t s);

// Column panels for row-sorted device triplets whose columns are scattered (most 256-entry chunks
// reach further than an eighth of the columns), with x larger than one XCD's L2: see
// spmv::coo_wide_kernel<true>.  Leaves the context without panels when the triplets do not qualify.
static int ctx_coo_panels(spmv_hip_ctx * c, const int32_t * d_row, const int32_t * d_col, const double * d_val, int32_t nnz)
{
    // the same bounds as for CSR (spmv_hip_plan_csr_repack); with fewer than 4 entries per row the
    // extra atomics of rows cut into panels cost what the gather gains (power law 3/row: 42 -> 44 us)
    if ((c->flags & (SPMV_HIP_FLAG_NO_COLUMN_PANELS | SPMV_HIP_FLAG_COO_KEEP_ORDER)) || nnz < (1 << 20)
        || (long long) c->cols * 8 < 3 * 1024 * 1024 || (long long) nnz < 4LL * c->rows)
        return SPMV_HIP_OK;
    int * d_count = nullptr;
    int spread = 0;
    HIP_TRY(hipMalloc((void **) &d_count, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(int), c->stream);
    const unsigned chunks = (unsigned) (((long long) nnz + 255) / 256);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spmv::coo_chunk_spread_kernel, dim3((chunks + 3) / 4), dim3(256), 0, c->stream, nnz,
                           std::max(1, (c->cols + 7) / 8), d_col, d_count);
        e = hipMemcpyAsync(&spread, d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void) hipFree(d_count);
    if (e != hipSuccess)
        return fail_hip(e, "COO panels");
    if (2LL * spread <= (long long) chunks)
        return SPMV_HIP_OK;
    int rc = spmv_hip_internal_coo_panels(c->cols, nnz, d_row, d_col, d_val, &c->d_prow, &c->d_pcol, &c->d_pval,
                                          c->coo_panels.start, c->stream);
    if (rc != SPMV_HIP_OK)
        return fail(rc, "COO panels");
    long long most = 0;
    for (int k = 0; k < 8; ++k)
        most = std::max(most, c->coo_panels.start[k + 1] - c->coo_panels.start[k]);
    c->coo_panel_blocks = (int) (most / 1024);
    c->bytes += (size_t) c->coo_panels.start[8] * 16;
    return SPMV_HIP_OK;
}

static int ctx_coo_run(spmv_hip_ctx * c, int32_t nnz, const int32_t * d_row, const int32_t * d_col, const double * d_val)
{
    if (c->d_prow && c->coo_panel_blocks > 0) {
        hipLaunchKernelGGL((spmv::coo_wide_kernel<true>), dim3((unsigned) (8 * c->coo_panel_blocks)), dim3(256), 0, c->stream,
                           (int) c->coo_panels.start[8], c->d_prow, c->d_pcol, c->d_pval, c->d_x, c->d_y, c->coo_panels);
        HIP_TRY(hipGetLastError());
        return SPMV_HIP_OK;
    }
    return spmv_hip_coo_spmv(c->rows, nnz, d_row, d_col, d_val, c->d_x, c->d_y, c->stream);
}

// Row-sorted triplets ARE a CSR matrix whose row_ptr is the run-length of the row stream: build it on
// the device, fetch it (rows + 1 integers) for the tile builder, and give the context a CSR plan for
// (d_ptr, d_col, d_val).  After that the row-index stream is not needed any more: the multiply
// streams 12 instead of 16 bytes per entry, needs no atomics, and gets every tile class of the CSR
// path (16-bit columns, shifted tiles, balanced tiles, column panels).
static int ctx_csr_from_sorted_rows(spmv_hip_ctx * c, const int32_t * d_rows_sorted, int32_t nnz, std::vector<int32_t> * host_ptr_out)
{
    int rc;
    if ((rc = dev_alloc(c, &c->d_ptr, (size_t) c->rows + 1)) != 0) return rc;
    hipLaunchKernelGGL(spmv::rowptr_from_sorted_kernel, dim3((unsigned) grid_for((long long) nnz + 1, kBlock, cu_count() * 16)), dim3(256), 0,
                       c->stream, (long long) nnz, (int) c->rows, d_rows_sorted, c->d_ptr);
    HIP_TRY(hipGetLastError());
    host_ptr_out->resize((size_t) c->rows + 1);
    HIP_TRY(hipMemcpyAsync(host_ptr_out->data(), c->d_ptr, host_ptr_out->size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

static int ctx_plan_device_csr(spmv_hip_ctx * c, const std::vector<int32_t> & host_ptr, unsigned extra_flags)
{
    int rc = spmv_hip_plan_csr(&c->plan, c->rows, c->cols, host_ptr.data(), SPMV_HIP_CSR_WAVETILE, 0, c->flags | extra_flags);
    if (rc != 0)
        return rc;
    if (!(c->flags & SPMV_HIP_FLAG_NO_INDEX_COMPRESSION)) {
        if ((rc = spmv_hip_plan_csr_compress(c->plan, c->d_col, c->stream)) != 0) return rc;
        if ((rc = spmv_hip_plan_csr_repack(c->plan, c->d_ptr, c->d_col, c->d_val, c->stream)) != 0) return rc;
        if ((rc = spmv_hip_plan_csr_index_values(c->plan, c->d_val, c->stream)) != 0) return rc;
    }
    return SPMV_HIP_OK;
}

int spmv_hip_upload_coo(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz,
                        const int32_t * row_index, const int32_t * column_index, const double * value)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return multi_upload_coo(c, rows, cols, nnz, row_index, column_index, value);
    if (rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!row_index || !column_index || !value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad COO arguments");
    bool row_sorted = true;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_ctx_matrix(c);
    c->rows = rows;
    c->cols = cols;
    c->nnz = nnz;
    int rc;
    if ((rc = dev_alloc(c, &c->d_idx, (size_t) nnz)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_col, (size_t) nnz)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_val, (size_t) nnz)) != 0) return rc;
    if (nnz > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_idx, row_index, (size_t) nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_col, column_index, (size_t) nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_val, value, (size_t) nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = ctx_common_vectors(c)) != 0) return rc;
    {
        bool bad_row = false, bad_col = false;
        if ((rc = device_index_check(c->d_idx, nnz, rows, true, &bad_row, &row_sorted, c->stream)) != 0) return rc;
        if ((rc = device_index_check(c->d_col, nnz, cols, false, &bad_col, nullptr, c->stream)) != 0) return rc;
        if (bad_row || bad_col) {
            free_ctx_matrix(c);
            return fail(SPMV_HIP_ERR_INVALID, "row or column index out of range");
        }
    }
    if (!(c->flags & SPMV_HIP_FLAG_COO_KEEP_ORDER) && nnz > 0 && rows > 0) {
        // default: sort by row once if need be (stably: a row keeps its file order), then multiply the
        // triplets as the row-major matrix they are (see ctx_csr_from_sorted_rows)
        if (!row_sorted) {
            if ((rc = spmv_hip_coo_sort_by_row(rows, nnz, c->d_idx, c->d_col, c->d_val, c->stream)) != 0) return rc;
            c->coo_sorted_on_device = true;
        }
        std::vector<int32_t> host_ptr;
        if ((rc = ctx_csr_from_sorted_rows(c, c->d_idx, nnz, &host_ptr)) != 0) return rc;
        if ((rc = ctx_plan_device_csr(c, host_ptr, 0)) != 0) return rc;
        (void) hipFree(c->d_idx); // the row stream has done its work
        c->d_idx = nullptr;
        c->bytes -= (size_t) nnz * sizeof(int32_t) + 64;
        c->as_csr = true;
    }
    c->format = 2;
    return SPMV_HIP_OK;
}

int spmv_hip_upload_ell(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length,
                        const int32_t * column_index, const double * value)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return multi_upload_ell(c, rows, cols, row_length, column_index, value);
    if (rows < 0 || cols < 0 || row_length < 0)
        return fail(SPMV_HIP_ERR_INVALID, "bad ELL arguments");
    int32_t n;
    if (__builtin_mul_overflow(rows, row_length, &n))
        return fail(SPMV_HIP_ERR_OVERFLOW, "Integer overflow when computing number of non-zeros");
    if (n > 0 && (!column_index || !value))
        return fail(SPMV_HIP_ERR_INVALID, "null ELL arrays");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_ctx_matrix(c);
    c->rows = rows;
    c->cols = cols;
    c->nnz = n;
    c->row_length = row_length;
    int rc;
    if ((rc = dev_alloc(c, &c->d_col, (size_t) n)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_val, (size_t) n)) != 0) return rc;
    // The row-major ELLPACK arrays ARE a CSR matrix with row_ptr[i] = i*row_length, every tile of
    // which is uniform (row bounds from the descriptor, no row_ptr traffic) and eligible for 16-bit
    // columns and shifted tiles: while at least two rows fit a tile they run through the wave-tile
    // kernel in place, one lane per row -- the reference's order -- with no transposed copy
    // (measured against the column-major kernel: L=5 202 vs 265 us, L=27 229 vs 285, L=81 337 vs 368).
    // Longer rows take the column-major one-lane-per-row kernel, which keeps the order for any length.
    // Rows of more than kEllInPlaceMaxLength entries go through the column-major kernel after all: with one lane per
    // row (the reference's order) a tile of such rows keeps a handful of lanes busy -- measured in place / column-major,
    // fraction of the roofline: L = 65 0.70 / 0.64, L = 93 (queen-like) 0.56 / 0.66, L = 97 0.59 / 0.65, L = 301 0.20 / 0.65,
    // L = 601 0.39 / 0.59 (profiles/r02_ell_row_lengths.log).
    c->ell_as_tiles = n > 0 && !(c->flags & SPMV_HIP_FLAG_ELL_COLUMN_MAJOR) && (row_length <= kEllInPlaceMaxLength || c->ell_in_place_any_length);
    if (c->ell_as_tiles) {
        std::vector<int32_t> row_ptr((size_t) rows + 1);
        for (int32_t i = 0; i <= rows; ++i)
            row_ptr[(size_t) i] = i * row_length;
        if ((rc = spmv_hip_plan_csr(&c->plan, rows, cols, row_ptr.data(), SPMV_HIP_CSR_WAVETILE, 0,
                                    c->flags | SPMV_HIP_FLAG_EXACT_ORDER)) != 0)
            return rc;
        if ((rc = dev_alloc(c, &c->d_ptr, (size_t) rows + 1)) != 0) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_ptr, row_ptr.data(), ((size_t) rows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_col, column_index, (size_t) n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_val, value, (size_t) n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        bool bad = false;
        if ((rc = device_index_check(c->d_col, n, cols, false, &bad, nullptr, c->stream)) != 0) return rc;
        if (bad) {
            free_ctx_matrix(c);
            return fail(SPMV_HIP_ERR_INVALID, "column index out of range");
        }
        if (!(c->flags & SPMV_HIP_FLAG_NO_INDEX_COMPRESSION)) {
            if ((rc = spmv_hip_plan_csr_compress(c->plan, c->d_col, c->stream)) != 0) return rc;
            if ((rc = spmv_hip_plan_csr_index_values(c->plan, c->d_val, c->stream)) != 0) return rc;
        }
    } else if (n > 0) {
        int32_t * t_col = nullptr;
        double * t_val = nullptr;
        HIP_TRY(hipMalloc((void **) &t_col, (size_t) n * sizeof(int32_t)));
        hipError_t e = hipMalloc((void **) &t_val, (size_t) n * sizeof(double));
        if (e != hipSuccess) {
            (void) hipFree(t_col);
            return fail_hip(e, "hipMalloc");
        }
        e = hipMemcpyAsync(t_col, column_index, (size_t) n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(t_val, value, (size_t) n * sizeof(double), hipMemcpyHostToDevice, c->stream);
        bool bad = false;
        if (e == hipSuccess && device_index_check(t_col, n, cols, false, &bad, nullptr, c->stream) != 0)
            e = hipErrorUnknown;
        if (e == hipSuccess && bad) {
            (void) hipFree(t_col);
            (void) hipFree(t_val);
            free_ctx_matrix(c);
            return fail(SPMV_HIP_ERR_INVALID, "column index out of range");
        }
        if (e == hipSuccess) {
            rc = spmv_hip_ell_to_column_major(rows, row_length, t_col, t_val, c->d_col, c->d_val, c->stream);
            e = hipStreamSynchronize(c->stream);
        }
        (void) hipFree(t_col);
        (void) hipFree(t_val);
        if (e != hipSuccess)
            return fail_hip(e, "ELL upload");
        if (rc != 0)
            return rc;
    }
    if ((rc = ctx_common_vectors(c)) != 0) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->format = 3;
    return SPMV_HIP_OK;
}

int spmv_hip_upload_hybrid(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t ell_row_length,
                           const int32_t * ell_column_index, const double * ell_value,
                           int32_t num_coo_entries, const int32_t * coo_row_index,
                           const int32_t * coo_column_index, const double * coo_value)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return fail(SPMV_HIP_ERR_STATE, "a multi-GPU context takes CSR, COO or ELLPACK (row blocks: src/matrix/csr-matrix.cpp:77-95)");
    if (num_coo_entries < 0 || (num_coo_entries > 0 && (!coo_row_index || !coo_column_index || !coo_value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad hybrid COO arguments");
    // the ELL part is uploaded (validated) exactly like a plain ELLPACK matrix ...
    c->ell_in_place_any_length = !(c->flags & (SPMV_HIP_FLAG_COO_KEEP_ORDER | SPMV_HIP_FLAG_EXACT_ORDER)); // ... to be merged below
    int rc = spmv_hip_upload_ell(c, rows, cols, ell_row_length, ell_column_index, ell_value);
    c->ell_in_place_any_length = false;
    if (rc != 0)
        return rc;
    // ... and the COO remainder rides along; should that fail, the context is left without a matrix
    c->format = 0;
    c->nnz2 = num_coo_entries;
    auto remainder = [&]() -> int {
        int r;
        if ((r = dev_alloc(c, &c->d_idx, (size_t) num_coo_entries)) != 0) return r;
        if ((r = dev_alloc(c, &c->d_col2, (size_t) num_coo_entries)) != 0) return r;
        if ((r = dev_alloc(c, &c->d_val2, (size_t) num_coo_entries)) != 0) return r;
        if (num_coo_entries > 0) {
            HIP_TRY(hipMemcpyAsync(c->d_idx, coo_row_index, (size_t) num_coo_entries * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_col2, coo_column_index, (size_t) num_coo_entries * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_val2, coo_value, (size_t) num_coo_entries * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        bool bad_row = false, bad_col = false, sorted = true;
        if ((r = device_index_check(c->d_idx, num_coo_entries, rows, true, &bad_row, &sorted, c->stream)) != 0) return r;
        if ((r = device_index_check(c->d_col2, num_coo_entries, cols, false, &bad_col, nullptr, c->stream)) != 0) return r;
        if (bad_row || bad_col)
            return fail(SPMV_HIP_ERR_INVALID, "row or column index out of range");
        // One fused multiply: the reference adds the ELL part and then the remainder into the same y
        // (hybrid_matrix::spmv, src/matrix/hybrid-matrix.cpp:535-567); here the two parts are merged on
        // the device into ONE row-major matrix -- row r = its ELL entries, padding included, then its
        // remainder entries -- so a run is one launch over balanced tiles instead of an ELL launch
        // followed by an atomic COO launch (webbase-like: 10.8 + 18.9 us before).  The remainder is in
        // (row, column) order (hybrid-matrix.cpp:316-417); any other order is sorted by row first.
        const long long merged = (long long) c->nnz + num_coo_entries;
        if (c->ell_as_tiles && !(c->flags & (SPMV_HIP_FLAG_COO_KEEP_ORDER | SPMV_HIP_FLAG_EXACT_ORDER)) && merged <= INT32_MAX && rows > 0) {
            if (!sorted && (r = spmv_hip_coo_sort_by_row(rows, num_coo_entries, c->d_idx, c->d_col2, c->d_val2, c->stream)) != 0) return r;
            // the ELL arrays sit in d_col / d_val with their plan; keep them aside, build the merged ones
            int32_t * ell_col = c->d_col, * coo_ptr = nullptr;
            double * ell_val = c->d_val;
            spmv_hip_plan_destroy(c->plan);
            c->plan = nullptr;
            (void) hipFree(c->d_ptr);
            c->d_ptr = nullptr;
            c->d_col = nullptr;
            c->d_val = nullptr;
            auto cleanup = [&] { (void) hipFree(ell_col); (void) hipFree(ell_val); };
            std::vector<int32_t> host_ptr;
            if ((r = ctx_csr_from_sorted_rows(c, c->d_idx, num_coo_entries, &host_ptr)) != 0) { cleanup(); return r; }
            coo_ptr = c->d_ptr; // row_ptr of the remainder alone
            c->d_ptr = nullptr;
            for (int32_t q = 0; q <= rows; ++q)
                host_ptr[(size_t) q] += q * ell_row_length;
            hipError_t e = hipSuccess;
            if ((r = dev_alloc(c, &c->d_ptr, (size_t) rows + 1)) != 0 || (r = dev_alloc(c, &c->d_col, (size_t) merged)) != 0
                || (r = dev_alloc(c, &c->d_val, (size_t) merged)) != 0) {
                cleanup();
                (void) hipFree(coo_ptr);
                return r;
            }
            e = hipMemcpyAsync(c->d_ptr, host_ptr.data(), host_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(spmv::hybrid_merge_kernel, dim3((unsigned) ((rows + 255) / 256)), dim3(256), 0, c->stream, (int) rows,
                                   (int) ell_row_length, ell_col, ell_val, coo_ptr, c->d_col2, c->d_val2, c->d_col, c->d_val);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            cleanup();
            (void) hipFree(coo_ptr);
            if (e != hipSuccess)
                return fail_hip(e, "hybrid merge");
            void * parts[] = {c->d_idx, c->d_col2, c->d_val2};
            for (void * q : parts)
                (void) hipFree(q);
            c->d_idx = c->d_col2 = nullptr;
            c->d_val2 = nullptr;
            c->ell_as_tiles = false;
            c->as_csr = true;
            // what the context holds now: the merged matrix, x and y (the parts it was made from are gone)
            c->bytes = ((size_t) rows + 1) * sizeof(int32_t) + (size_t) merged * (sizeof(int32_t) + sizeof(double))
                + ((size_t) cols + (size_t) rows) * sizeof(double) + 5 * 64;
            return ctx_plan_device_csr(c, host_ptr, 0);
        }
        // two launches (file order kept, exact ELL order asked for, or the merged matrix would not fit
        // int32): scattered remainders get column panels
        return ctx_coo_panels(c, c->d_idx, c->d_col2, c->d_val2, num_coo_entries);
    };
    rc = remainder();
    if (rc != 0) {
        std::string const keep = g_last_error;
        free_ctx_matrix(c);
        g_last_error = keep;
        return rc;
    }
    c->format = 4;
    return SPMV_HIP_OK;
}

int spmv_hip_set_x(spmv_hip_ctx * c, const double * x)
{
    if (!c || !x)
        return fail(SPMV_HIP_ERR_INVALID, "ctx/x null");
    if (c->format == 0)
        return fail(SPMV_HIP_ERR_STATE, "no matrix uploaded");
    if (c->multi)
        return multi_set_x(c, x);
    HIP_TRY(hipSetDevice(c->device));
    if (c->cols > 0)
        HIP_TRY(hipMemcpyAsync(c->d_x, x, (size_t) c->cols * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

int spmv_hip_set_y(spmv_hip_ctx * c, const double * y)
{
    if (!c || !y)
        return fail(SPMV_HIP_ERR_INVALID, "ctx/y null");
    if (c->format == 0)
        return fail(SPMV_HIP_ERR_STATE, "no matrix uploaded");
    if (c->multi)
        return multi_set_y(c, y);
    HIP_TRY(hipSetDevice(c->device));
    if (c->rows > 0)
        HIP_TRY(hipMemcpyAsync(c->d_y, y, (size_t) c->rows * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

int spmv_hip_get_y(spmv_hip_ctx * c, double * y)
{
    if (!c || !y)
        return fail(SPMV_HIP_ERR_INVALID, "ctx/y null");
    if (c->format == 0)
        return fail(SPMV_HIP_ERR_STATE, "no matrix uploaded");
    if (c->multi)
        return multi_get_y(c, y);
    HIP_TRY(hipSetDevice(c->device));
    if (c->rows > 0)
        HIP_TRY(hipMemcpyAsync(y, c->d_y, (size_t) c->rows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

int spmv_hip_run(spmv_hip_ctx * c)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->format == 0)
        return fail(SPMV_HIP_ERR_STATE, "no matrix uploaded");
    if (c->multi)
        return multi_run(c);
    HIP_TRY(hipSetDevice(c->device));
    const bool timed = !(c->flags & SPMV_HIP_FLAG_NO_RUN_EVENTS);
    if (timed)
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
    int rc = SPMV_HIP_OK;
    switch (c->format) {
    case 1: rc = spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream); break;
    case 2:
        rc = c->as_csr ? spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream)
                       : ctx_coo_run(c, c->nnz, c->d_idx, c->d_col, c->d_val);
        break;
    case 3:
        rc = c->ell_as_tiles
            ? spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream)
            : spmv_hip_ell_spmv(c->rows, c->row_length, c->d_col, c->d_val, c->d_x, c->d_y, c->stream);
        break;
    case 4:
        if (c->as_csr) { // ELL part and remainder merged into one row-major matrix: one launch
            rc = spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream);
            break;
        }
        rc = c->ell_as_tiles
            ? spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream)
            : spmv_hip_ell_spmv(c->rows, c->row_length, c->d_col, c->d_val, c->d_x, c->d_y, c->stream);
        if (rc == 0)
            rc = ctx_coo_run(c, c->nnz2, c->d_idx, c->d_col2, c->d_val2);
        break;
    }
    if (rc != 0)
        return rc;
    if (timed) {
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    return SPMV_HIP_OK;
}

int spmv_hip_sync(spmv_hip_ctx * c)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return multi_sync(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

int spmv_hip_flush_caches(spmv_hip_ctx * c)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi) {
        for (spmv_hip_ctx * part : c->parts) {
            int rc = spmv_hip_flush_caches(part);
            if (rc != 0)
                return rc;
        }
        return SPMV_HIP_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const long long n = 1LL << 27; // 2^27 doubles = 1 GiB: four times the 256 MB Infinity Cache
    if (!c->d_flush)
        HIP_TRY(hipMalloc((void **) &c->d_flush, (size_t) n * sizeof(double)));
    // a = b + q * c over the halves of the scratch: reads 512 MiB, writes 512 MiB (values are irrelevant)
    const long long n2 = n / 4; // 16-byte elements per array, two arrays read from the upper half
    hipLaunchKernelGGL((spmv::triad_flat_kernel<kBlock, false>), dim3((unsigned) ((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, n2,
                       c->d_flush, c->d_flush + n / 2, c->d_flush + n / 2, 0.0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPMV_HIP_OK;
}

int spmv_hip_last_run_ns(spmv_hip_ctx * c, uint64_t * kernel_ns)
{
    if (!c || !kernel_ns)
        return fail(SPMV_HIP_ERR_INVALID, "ctx/out null");
    if (c->multi)
        return multi_times(c, kernel_ns, nullptr);
    if (!c->timed)
        return fail(SPMV_HIP_ERR_STATE, "no run recorded");
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *kernel_ns = (uint64_t) (ms * 1.0e6 + 0.5);
    return SPMV_HIP_OK;
}

int spmv_hip_ctx_info(spmv_hip_ctx * c, int64_t * out, int n)
{
    if (!c || !out || n < 0)
        return fail(SPMV_HIP_ERR_INVALID, "ctx/out null");
    if (c->multi) {
        // the whole matrix: sizes from the front, tile counts / bytes summed over the devices, [16] = devices
        int64_t v[17] = {c->format, c->rows, c->cols, c->nnz, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (int64_t) c->parts.size()};
        for (size_t g = 0; g < c->parts.size(); ++g) {
            int64_t w[16] = {0};
            if (c->parts[g]->format != 0)
                spmv_hip_ctx_info(c->parts[g], w, 16);
            v[4] = w[4] ? w[4] : v[4];
            v[5] = w[5];
            for (int i : {6, 7, 8, 9, 10, 11, 12, 13, 14, 15})
                v[i] += w[i];
            if (c->yfull[g])
                v[9] += (int64_t) c->chunk * (int64_t) c->parts.size() * 8;
        }
        for (int i = 0; i < n && i < 17; ++i)
            out[i] = v[i];
        return SPMV_HIP_OK;
    }
    int64_t v[17] = {c->format, c->rows, c->cols, c->nnz, 0, 0, 0, 0, 0, (int64_t) c->bytes, 0, 0, 0, 0, 0, 0, 1};
    // [15] bytes one run streams: the plan's count where tiles are used, else the format's algorithmic bytes
    switch (c->format) {
    case 2: v[15] = 16LL * c->nnz + 16LL * c->rows + 8LL * c->cols; break;
    case 3: v[15] = 12LL * c->nnz + 16LL * c->rows + 8LL * c->cols; break;
    case 4: v[15] = 12LL * c->nnz + 16LL * c->nnz2 + 16LL * c->rows + 8LL * c->cols; break;
    default: break;
    }
    if (c->plan) {
        v[15] = c->plan->streamed_bytes + (c->format == 4 && !c->as_csr ? 16LL * c->nnz2 : 0);
        v[4] = c->plan->algorithm;
        v[5] = c->plan->lanes_per_row;
        v[6] = c->plan->workgroups;
        v[7] = c->plan->nblk;
        v[8] = c->plan->long_blocks;
        v[9] += (int64_t) c->plan->meta_bytes;
        v[10] = c->plan->narrow_tiles;
        v[11] = c->plan->shifted_tiles;
        v[12] = c->plan->xwin_tiles;
        v[13] = c->plan->blockwin_tiles;
        v[14] = c->plan->inner ? c->plan->inner->ntiles : 0;
    }
    if (c->d_prow)
        v[14] += c->coo_panel_blocks; // COO (part) in column panels: workgroups per panel
    for (int i = 0; i < n && i < 17; ++i)
        out[i] = v[i];
    return SPMV_HIP_OK;
}

} // extern "C"

/* ---- multi-GPU front ------------------------------------------------------------------------------
 * One process, G devices: rows are cut by the reference's static rule,
 * chunk = ceil(rows / G) (src/matrix/csr-matrix.cpp:77-95, devices take the place of threads), x is
 * replicated, and a run is G local multiplies followed by ONE in-place ncclAllGather of the y slots
 * inside a group call.  librccl.so is loaded with dlopen only when G > 1 (or when
 * SPMV_HIP_FORCE_RCCL=1 asks for the collective with one device): a single-GPU build has no
 * dependency on it, and a process that already holds another RCCL (PyTorch's) is not handed a second
 * one behind its back. */
namespace {

int multi_fail_nccl(spmv_hip_ctx * c, ncclResult_t r, const char * what)
{
    std::string msg = std::string(what) + ": " + (c->p_error_string ? c->p_error_string(r) : "RCCL error");
    return fail(SPMV_HIP_ERR_HIP, msg.c_str());
}

int multi_load_rccl(spmv_hip_ctx * c, int num_gpus)
{
    const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char * n : names)
        if ((c->rccl_lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr)
            break;
    if (!c->rccl_lib)
        return fail(SPMV_HIP_ERR_STATE, "librccl.so could not be loaded: a multi-GPU context needs RCCL");
    auto p_init_all = reinterpret_cast<ncclResult_t (*)(ncclComm_t *, int, const int *)>(dlsym(c->rccl_lib, "ncclCommInitAll"));
    c->p_all_gather = reinterpret_cast<decltype(c->p_all_gather)>(dlsym(c->rccl_lib, "ncclAllGather"));
    c->p_group_start = reinterpret_cast<decltype(c->p_group_start)>(dlsym(c->rccl_lib, "ncclGroupStart"));
    c->p_group_end = reinterpret_cast<decltype(c->p_group_end)>(dlsym(c->rccl_lib, "ncclGroupEnd"));
    c->p_comm_destroy = reinterpret_cast<decltype(c->p_comm_destroy)>(dlsym(c->rccl_lib, "ncclCommDestroy"));
    c->p_error_string = reinterpret_cast<decltype(c->p_error_string)>(dlsym(c->rccl_lib, "ncclGetErrorString"));
    if (!p_init_all || !c->p_all_gather || !c->p_group_start || !c->p_group_end || !c->p_comm_destroy)
        return fail(SPMV_HIP_ERR_STATE, "librccl.so lacks ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd");
    std::vector<int> devs((size_t) num_gpus);
    for (int g = 0; g < num_gpus; ++g)
        devs[(size_t) g] = g;
    c->comms.assign((size_t) num_gpus, nullptr);
    ncclResult_t r = p_init_all(c->comms.data(), num_gpus, devs.data());
    if (r != ncclSuccess) {
        c->comms.clear();
        return multi_fail_nccl(c, r, "ncclCommInitAll");
    }
    return SPMV_HIP_OK;
}

void multi_free_matrix(spmv_hip_ctx * c)
{
    for (size_t g = 0; g < c->parts.size(); ++g) {
        (void) hipSetDevice(c->parts[g]->device);
        (void) hipStreamSynchronize(c->parts[g]->stream);
        free_ctx_matrix(c->parts[g]);
        c->parts[g]->borrowed_y = nullptr;
        if (g < c->yfull.size() && c->yfull[g]) {
            (void) hipFree(c->yfull[g]);
            c->yfull[g] = nullptr;
        }
    }
    c->format = 0;
    c->rows = c->cols = c->nnz = 0;
    c->chunk = 0;
    c->row_begin.clear();
    c->packed = true;
    c->timed = false;
}

// Row blocks of a multi-GPU context and each device's copy of y.  row_ptr (rows + 1 entries, any base) gives the
// stored entries in front of every row: the reference's static rule needs only `rows`, SPMV_HIP_FLAG_BALANCE_ENTRIES
// cuts where the entries divide evenly (boundary g = the first row whose row_ptr reaches g * nnz / G).
int multi_layout(spmv_hip_ctx * c, int32_t rows, const long long * entries_before_row /* rows + 1, or null */)
{
    multi_free_matrix(c);
    const int G = (int) c->parts.size();
    c->row_begin.assign((size_t) G + 1, 0);
    if ((c->flags & SPMV_HIP_FLAG_BALANCE_ENTRIES) && entries_before_row) {
        const long long nnz = entries_before_row[rows] - entries_before_row[0];
        for (int g = 1; g < G; ++g) {
            const long long target = entries_before_row[0] + (nnz * g) / G;
            const int32_t r = (int32_t) (std::lower_bound(entries_before_row, entries_before_row + rows + 1, target) - entries_before_row);
            c->row_begin[(size_t) g] = std::max(c->row_begin[(size_t) g - 1], std::min(r, rows));
        }
    } else {
        const long long per = std::max<long long>(1, ((long long) rows + G - 1) / G); // ceil(rows / G): src/matrix/csr-matrix.cpp:77-95
        for (int g = 1; g < G; ++g)
            c->row_begin[(size_t) g] = (int32_t) std::min<long long>(rows, g * per);
    }
    c->row_begin[(size_t) G] = rows;
    int32_t chunk = 1;
    for (int g = 0; g < G; ++g)
        chunk = std::max(chunk, c->row_begin[(size_t) g + 1] - c->row_begin[(size_t) g]);
    c->chunk = chunk; // slots are equally long (the all-gather wants equal counts); shorter blocks leave padding
    c->packed = true;
    for (int g = 0; g < G; ++g) {
        const int32_t b = c->row_begin[(size_t) g], e = c->row_begin[(size_t) g + 1];
        if (e > b && (b != (long long) g * chunk || (e - b != chunk && e != rows)))
            c->packed = false;
    }
    for (int g = 0; g < G; ++g) {
        spmv_hip_ctx * part = c->parts[(size_t) g];
        HIP_TRY(hipSetDevice(part->device));
        // the device's copy of the whole y: G slots of `chunk` doubles (the last ones padded), zeroed
        const size_t ybytes = (size_t) chunk * (size_t) G * sizeof(double) + 64;
        HIP_TRY(hipMalloc((void **) &c->yfull[(size_t) g], ybytes));
        HIP_TRY(hipMemsetAsync(c->yfull[(size_t) g], 0, ybytes, part->stream));
        part->y_borrowed = true;
        part->borrowed_y = c->yfull[(size_t) g] + (size_t) g * (size_t) chunk;
        part->csr_algorithm = c->csr_algorithm;
        part->csr_lanes = c->csr_lanes;
    }
    return SPMV_HIP_OK;
}

int multi_upload_failed(spmv_hip_ctx * c, int rc)
{
    std::string const keep = g_last_error;
    multi_free_matrix(c);
    g_last_error = keep;
    return rc;
}

int multi_upload_csr(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_ptr,
                     const int32_t * column_index, const double * value)
{
    if (rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!column_index || !value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad CSR arguments");
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz)
        return fail(SPMV_HIP_ERR_INVALID, "row_ptr[0] must be 0 and row_ptr[rows] must equal nnz");
    std::vector<long long> before;
    if (c->flags & SPMV_HIP_FLAG_BALANCE_ENTRIES)
        before.assign(row_ptr, row_ptr + rows + 1);
    int rc = multi_layout(c, rows, before.empty() ? nullptr : before.data());
    if (rc != 0)
        return multi_upload_failed(c, rc);
    std::vector<int32_t> local_ptr;
    for (size_t g = 0; g < c->parts.size(); ++g) {
        const int32_t b = c->row_begin[g], e = c->row_begin[g + 1];
        local_ptr.resize((size_t) (e - b) + 1);
        for (int32_t r = b; r <= e; ++r)
            local_ptr[(size_t) (r - b)] = row_ptr[r] - row_ptr[b];
        rc = spmv_hip_upload_csr(c->parts[g], e - b, cols, row_ptr[e] - row_ptr[b], local_ptr.data(),
                                 column_index ? column_index + row_ptr[b] : nullptr, value ? value + row_ptr[b] : nullptr);
        if (rc != 0)
            return multi_upload_failed(c, rc);
    }
    c->rows = rows;
    c->cols = cols;
    c->nnz = nnz;
    c->format = 1;
    return SPMV_HIP_OK;
}

// ELLPACK across the devices ("ELL: row range"): every row has row_length slots, so the blocks of the
// static rule are also the blocks of equal entries; device g gets rows [b, e) of the row-major arrays as they are.
int multi_upload_ell(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length, const int32_t * column_index, const double * value)
{
    if (rows < 0 || cols < 0 || row_length < 0 || ((long long) rows * row_length > 0 && (!column_index || !value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad ELL arguments");
    if ((long long) rows * row_length > INT32_MAX)
        return fail(SPMV_HIP_ERR_OVERFLOW, "Integer overflow when computing number of non-zeros");
    int rc = multi_layout(c, rows, nullptr);
    if (rc != 0)
        return multi_upload_failed(c, rc);
    for (size_t g = 0; g < c->parts.size(); ++g) {
        const int32_t b = c->row_begin[g], e = c->row_begin[g + 1];
        const size_t off = (size_t) b * (size_t) row_length;
        rc = spmv_hip_upload_ell(c->parts[g], e - b, cols, row_length, column_index ? column_index + off : nullptr, value ? value + off : nullptr);
        if (rc != 0)
            return multi_upload_failed(c, rc);
    }
    c->rows = rows;
    c->cols = cols;
    c->nnz = (int32_t) ((long long) rows * row_length);
    c->format = 3;
    return SPMV_HIP_OK;
}

// COO across the devices ("split the row-sorted stream at row boundaries"): the triplets may come in any
// order (file order: src/matrix/coo-matrix.cpp:220-243); they are dealt to the row blocks by a stable counting pass --
// every device gets its rows' triplets in their original relative order, row indices rebased to the block -- and
// each device then treats its share like any COO upload (sorted by row on the device, multiplied as row-major tiles).
int multi_upload_coo(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_index, const int32_t * column_index,
                     const double * value)
{
    if (rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && (!row_index || !column_index || !value)))
        return fail(SPMV_HIP_ERR_INVALID, "bad COO arguments");
    for (int32_t k = 0; k < nnz; ++k)
        if (row_index[k] < 0 || row_index[k] >= rows)
            return fail(SPMV_HIP_ERR_INVALID, "row or column index out of range");
    std::vector<long long> before;
    if (c->flags & SPMV_HIP_FLAG_BALANCE_ENTRIES) { // entries in front of every row, from a histogram of the row indices
        before.assign((size_t) rows + 1, 0);
        for (int32_t k = 0; k < nnz; ++k)
            ++before[(size_t) row_index[k] + 1];
        for (int32_t r = 0; r < rows; ++r)
            before[(size_t) r + 1] += before[(size_t) r];
    }
    int rc = multi_layout(c, rows, before.empty() ? nullptr : before.data());
    if (rc != 0)
        return multi_upload_failed(c, rc);
    const size_t G = c->parts.size();
    // block of a row: binary search over the G + 1 boundaries (blocks may be empty)
    auto block_of = [&](int32_t r) {
        return (size_t) (std::upper_bound(c->row_begin.begin() + 1, c->row_begin.end(), r) - (c->row_begin.begin() + 1));
    };
    std::vector<size_t> start(G + 1, 0);
    for (int32_t k = 0; k < nnz; ++k)
        ++start[block_of(row_index[k]) + 1];
    for (size_t g = 0; g < G; ++g)
        start[g + 1] += start[g];
    std::vector<int32_t> ri((size_t) nnz), ci((size_t) nnz);
    std::vector<double> va((size_t) nnz);
    std::vector<size_t> fill(start.begin(), start.end() - 1);
    for (int32_t k = 0; k < nnz; ++k) {
        const size_t g = block_of(row_index[k]);
        const size_t at = fill[g]++;
        ri[at] = row_index[k] - c->row_begin[g];
        ci[at] = column_index[k];
        va[at] = value[k];
    }
    for (size_t g = 0; g < G; ++g) {
        const int32_t b = c->row_begin[g], e = c->row_begin[g + 1];
        const size_t off = start[g], cnt = start[g + 1] - start[g];
        rc = spmv_hip_upload_coo(c->parts[g], e - b, cols, (int32_t) cnt, cnt ? ri.data() + off : nullptr, cnt ? ci.data() + off : nullptr,
                                 cnt ? va.data() + off : nullptr);
        if (rc != 0)
            return multi_upload_failed(c, rc);
    }
    c->rows = rows;
    c->cols = cols;
    c->nnz = nnz;
    c->format = 2;
    return SPMV_HIP_OK;
}

int multi_set_x(spmv_hip_ctx * c, const double * x)
{
    for (spmv_hip_ctx * part : c->parts) {
        int rc = spmv_hip_set_x(part, x);
        if (rc != 0)
            return rc;
    }
    return SPMV_HIP_OK;
}

int multi_set_y(spmv_hip_ctx * c, const double * y)
{
    int rc0 = multi_sync(c); // a peer's push of an earlier run may still be writing into the vectors replaced here
    if (rc0 != 0)
        return rc0;
    for (size_t g = 0; g < c->parts.size(); ++g) { // every device gets the whole y, as after a gather
        spmv_hip_ctx * part = c->parts[g];
        HIP_TRY(hipSetDevice(part->device));
        if (c->rows > 0 && c->packed)
            HIP_TRY(hipMemcpyAsync(c->yfull[g], y, (size_t) c->rows * sizeof(double), hipMemcpyHostToDevice, part->stream));
        for (size_t h = 0; h < c->parts.size() && !c->packed; ++h) { // block by block into the slots
            const int32_t b = c->row_begin[h], e = c->row_begin[h + 1];
            if (e > b)
                HIP_TRY(hipMemcpyAsync(c->yfull[g] + h * (size_t) c->chunk, y + b, (size_t) (e - b) * sizeof(double), hipMemcpyHostToDevice,
                                       part->stream));
        }
        HIP_TRY(hipStreamSynchronize(part->stream));
    }
    return SPMV_HIP_OK;
}

int multi_get_y(spmv_hip_ctx * c, double * y)
{
    if (c->peer_gather) { // device 0's y is complete once every OTHER device's push has finished
        int rc0 = multi_sync(c);
        if (rc0 != 0)
            return rc0;
    }
    spmv_hip_ctx * part = c->parts[0];
    HIP_TRY(hipSetDevice(part->device));
    if (c->rows > 0 && c->packed)
        HIP_TRY(hipMemcpyAsync(y, c->yfull[0], (size_t) c->rows * sizeof(double), hipMemcpyDeviceToHost, part->stream));
    for (size_t h = 0; h < c->parts.size() && !c->packed; ++h) {
        const int32_t b = c->row_begin[h], e = c->row_begin[h + 1];
        if (e > b)
            HIP_TRY(hipMemcpyAsync(y + b, c->yfull[0] + h * (size_t) c->chunk, (size_t) (e - b) * sizeof(double), hipMemcpyDeviceToHost,
                                   part->stream));
    }
    HIP_TRY(hipStreamSynchronize(part->stream));
    return SPMV_HIP_OK;
}

// SPMV_HIP_FLAG_PEER_GATHER: the all-gather as remote stores.  Device g reads its slot once and writes it into
// slot g of up to kPeerFanout other devices' y (coalesced stores that leave over the xGMI link to each peer: on a
// fully connected node all seven links of the device carry one copy each, which is what a direct all-gather
// over point-to-point links amounts to).  One launch per device and run; nothing is received by a kernel --
// the stores of the peers land in memory this device does not touch until the streams have been synchronised.
constexpr int kPeerFanout = 8;
struct PeerTargets {
    double * dst[kPeerFanout];
    int n;
};

__global__ __launch_bounds__(256) void peer_push_kernel(const double * __restrict__ src, PeerTargets t, long long n)
{
    // a slot starts at g * chunk doubles: 8-byte aligned only, hence one double per lane (a wave still writes
    // 512 contiguous bytes per store instruction)
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = src[i];
#pragma unroll
        for (int k = 0; k < kPeerFanout; ++k)
            if (k < t.n)
                t.dst[k][i] = v;
    }
}

int multi_peer_gather(spmv_hip_ctx * c)
{
    const int G = (int) c->parts.size();
    const size_t chunk = (size_t) c->chunk;
    const long long pairs = (long long) chunk;
    if (pairs == 0)
        return SPMV_HIP_OK;
    for (int g = 0; g < G; ++g) {
        spmv_hip_ctx * part = c->parts[(size_t) g];
        HIP_TRY(hipSetDevice(part->device));
        const unsigned blocks = (unsigned) std::min<long long>((pairs + 255) / 256, 8ll * cu_count());
        for (int h0 = 0; h0 < G; h0 += kPeerFanout) {
            PeerTargets t;
            t.n = 0;
            for (int h = h0; h < G && h < h0 + kPeerFanout; ++h)
                if (h != g)
                    t.dst[t.n++] = c->yfull[(size_t) h] + (size_t) g * chunk;
            for (int k = t.n; k < kPeerFanout; ++k)
                t.dst[k] = nullptr;
            if (t.n > 0)
                hipLaunchKernelGGL(peer_push_kernel, dim3(blocks), dim3(256), 0, part->stream,
                                   c->yfull[(size_t) g] + (size_t) g * chunk, t, pairs);
        }
        HIP_TRY(hipGetLastError());
    }
    return SPMV_HIP_OK;
}

int multi_enable_peers(spmv_hip_ctx * c)
{
    const int G = (int) c->parts.size();
    for (int g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(c->parts[(size_t) g]->device));
        for (int h = 0; h < G; ++h) {
            const int dg = c->parts[(size_t) g]->device, dh = c->parts[(size_t) h]->device;
            if (dg == dh)
                continue;
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, dg, dh));
            if (!can)
                return fail(SPMV_HIP_ERR_STATE, "SPMV_HIP_FLAG_PEER_GATHER: a device cannot access a peer's memory");
            const hipError_t e = hipDeviceEnablePeerAccess(dh, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return fail_hip(e, "hipDeviceEnablePeerAccess");
            (void) hipGetLastError();
        }
    }
    return SPMV_HIP_OK;
}

int multi_run(spmv_hip_ctx * c)
{
    const int G = (int) c->parts.size();
    for (spmv_hip_ctx * part : c->parts) { // every device multiplies its rows into its slot of its y
        int rc = spmv_hip_run(part);
        if (rc != 0)
            return rc;
    }
    if (c->peer_gather) {
        int rc = multi_peer_gather(c);
        if (rc != 0)
            return rc;
    } else if (!c->comms.empty()) {
        // the one collective of the path: every device sends its slot and receives the others', in place
        ncclResult_t r = c->p_group_start();
        if (r != ncclSuccess)
            return multi_fail_nccl(c, r, "ncclGroupStart");
        for (int g = 0; g < G && r == ncclSuccess; ++g)
            r = c->p_all_gather(c->yfull[(size_t) g] + (size_t) g * (size_t) c->chunk, c->yfull[(size_t) g], (size_t) c->chunk, ncclDouble,
                                c->comms[(size_t) g], c->parts[(size_t) g]->stream);
        ncclResult_t r2 = c->p_group_end();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return multi_fail_nccl(c, r != ncclSuccess ? r : r2, "ncclAllGather");
    }
    if (!(c->flags & SPMV_HIP_FLAG_NO_RUN_EVENTS)) {
        for (int g = 0; g < G; ++g) {
            HIP_TRY(hipSetDevice(c->parts[(size_t) g]->device));
            HIP_TRY(hipEventRecord(c->ev_gather[(size_t) g], c->parts[(size_t) g]->stream));
        }
    }
    c->timed = true;
    return SPMV_HIP_OK;
}

int multi_sync(spmv_hip_ctx * c)
{
    for (spmv_hip_ctx * part : c->parts) {
        HIP_TRY(hipSetDevice(part->device));
        HIP_TRY(hipStreamSynchronize(part->stream));
    }
    return SPMV_HIP_OK;
}

// slowest device's multiply, and the longest wait from the end of a device's multiply to the end of its gather
int multi_times(spmv_hip_ctx * c, uint64_t * kernel_ns, uint64_t * gather_ns)
{
    if (!c->timed)
        return fail(SPMV_HIP_ERR_STATE, "no run recorded");
    if (c->flags & SPMV_HIP_FLAG_NO_RUN_EVENTS)
        return fail(SPMV_HIP_ERR_STATE, "the context was created with SPMV_HIP_FLAG_NO_RUN_EVENTS: no run is timed");
    float kmax = 0.f, gmax = 0.f;
    for (size_t g = 0; g < c->parts.size(); ++g) {
        spmv_hip_ctx * part = c->parts[g];
        HIP_TRY(hipSetDevice(part->device));
        HIP_TRY(hipEventSynchronize(c->ev_gather[g]));
        float k = 0.f, ga = 0.f;
        HIP_TRY(hipEventElapsedTime(&k, part->ev0, part->ev1));
        HIP_TRY(hipEventElapsedTime(&ga, part->ev1, c->ev_gather[g]));
        kmax = std::max(kmax, k);
        gmax = std::max(gmax, ga);
    }
    if (kernel_ns) *kernel_ns = (uint64_t) (kmax * 1.0e6 + 0.5);
    if (gather_ns) *gather_ns = (uint64_t) (gmax * 1.0e6 + 0.5);
    return SPMV_HIP_OK;
}

} // namespace

extern "C" {

int spmv_hip_create_multi(spmv_hip_ctx ** out, int num_gpus, unsigned flags)
{
    if (!out)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    if (flags & ~kKnownFlags)
        return fail(SPMV_HIP_ERR_INVALID, "unknown flag bits");
    // SPMV_HIP_SHARE_DEVICES=1 (rehearsals on fewer devices than parts): part g runs on device g mod visible.
    // Only with the peer gather -- RCCL refuses two ranks on one device.
    const char * share_env = std::getenv("SPMV_HIP_SHARE_DEVICES");
    const bool share = share_env && share_env[0] == '1' && (flags & SPMV_HIP_FLAG_PEER_GATHER);
    if (num_gpus < 1 || (num_gpus > n && !share) || num_gpus > 64)
        return fail(SPMV_HIP_ERR_INVALID, "num_gpus must be between 1 and the number of visible devices");
    spmv_hip_ctx * c = new (std::nothrow) spmv_hip_ctx;
    if (!c)
        return fail(SPMV_HIP_ERR_ALLOC, "ctx allocation failed");
    c->multi = true;
    c->flags = flags;
    c->peer_gather = (flags & SPMV_HIP_FLAG_PEER_GATHER) != 0;
    c->yfull.assign((size_t) num_gpus, nullptr);
    int rc = SPMV_HIP_OK;
    for (int g = 0; g < num_gpus && rc == SPMV_HIP_OK; ++g) {
        spmv_hip_ctx * part = nullptr;
        rc = spmv_hip_create(&part, g % n, flags);
        if (rc == SPMV_HIP_OK) {
            c->parts.push_back(part);
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) != hipSuccess)
                rc = fail(SPMV_HIP_ERR_HIP, "hipEventCreate");
            c->ev_gather.push_back(ev);
        }
    }
    const char * force = std::getenv("SPMV_HIP_FORCE_RCCL");
    if (rc == SPMV_HIP_OK && c->peer_gather)
        rc = multi_enable_peers(c);
    else if (rc == SPMV_HIP_OK && (num_gpus > 1 || (force && force[0] == '1')))
        rc = multi_load_rccl(c, num_gpus);
    if (rc != SPMV_HIP_OK) {
        std::string const keep = g_last_error;
        spmv_hip_destroy(c);
        g_last_error = keep;
        return rc;
    }
    *out = c;
    return SPMV_HIP_OK;
}

int spmv_hip_last_run_times(spmv_hip_ctx * c, uint64_t * kernel_ns, uint64_t * gather_ns)
{
    if (!c)
        return fail(SPMV_HIP_ERR_INVALID, "ctx is null");
    if (c->multi)
        return multi_times(c, kernel_ns, gather_ns);
    if (gather_ns)
        *gather_ns = 0;
    uint64_t k = 0;
    int rc = spmv_hip_last_run_ns(c, &k);
    if (rc == SPMV_HIP_OK && kernel_ns)
        *kernel_ns = k;
    return rc;
}

} // extern "C"
