// internal.hpp -- what the translation units of libspmv_hip.so share: error reporting, the plan and context
// structures behind the opaque handles of include/spmv_hip.h, and the helpers that cross file boundaries.
//   common.hip       errors, library queries, device properties
//   plan_csr.hip     CSR launch plans: tiles, tile classes, column panels, value dictionary, content guards
//   launch.hip       the multiplies on caller-owned device arrays (Level 2) and their kernel instantiations
//   context.hip      Level 1: a context that owns device copies of A, x, y
//   multi_gpu.hip    one process, G devices: row blocks + one all-gather
//   peer_gather.hip  one process per GPU: y segments stored straight into the other ranks' vectors
//   coo_sort.hip     device sort of COO triplets, scans (hipCUB plumbing)
#pragma once

#include "spmv_hip_plan.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl.so is dlopen'ed by spmv_hip_create_multi when G > 1

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "spmv_kernels.hpp"
#include "csr_segwin.hpp"

namespace spmvi {

// last error text of the calling thread (spmv_hip_last_error); fail() returns `code` after recording `what`
int fail(int code, const char * what);
int fail_hip(hipError_t e, const char * call);
std::string last_error_text();
void set_last_error_text(std::string const & text);

#define HIP_TRY(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess)                           \
            return ::spmvi::fail_hip(e_, #call);        \
    } while (0)

// Two kernel families were built, measured slower than the paths they were meant to replace, and retired from the product
// library in round 5: hub columns for web graphs (tools/experiments/csr_hub.hpp: 26.6 against 23.9 us) and a lane group per
// row for stencil rows of 17 ... 64 entries (tools/experiments/csr_rowgroup.hpp: 797 against 740 us).  They live on in
// libspmv_hip_experiments.so (-DSPMV_HIP_EXPERIMENTS), with their parity tests (tests/experiments/); the product library
// refuses both bits like any unknown flag.
#define SPMV_HIP_FLAG_HUB_COLUMNS 0x4000000u
#define SPMV_HIP_FLAG_ROW_GROUPS 0x10000000u

// every documented SPMV_HIP_FLAG_* bit; anything else is refused (SPMV_HIP_ERR_INVALID)
constexpr unsigned kKnownFlags = SPMV_HIP_FLAG_XCD_REMAP | SPMV_HIP_FLAG_EXACT_ORDER | SPMV_HIP_FLAG_BIG_TILE |
    SPMV_HIP_FLAG_NO_INDEX_COMPRESSION | SPMV_HIP_FLAG_COO_KEEP_ORDER | SPMV_HIP_FLAG_READ_ROW_PTR | SPMV_HIP_FLAG_ROWS64 |
    SPMV_HIP_FLAG_ROWS128 | SPMV_HIP_FLAG_ELL_COLUMN_MAJOR | SPMV_HIP_FLAG_NO_SHIFTED_TILES | SPMV_HIP_FLAG_NO_X_WINDOW |
    SPMV_HIP_FLAG_NO_COLUMN_PANELS | SPMV_HIP_FLAG_VERIFY_PLAN | SPMV_HIP_FLAG_NO_BALANCED_TILES | SPMV_HIP_FLAG_NO_RUN_EVENTS | SPMV_HIP_FLAG_NO_VALUE_INDEX |
    SPMV_HIP_FLAG_PEER_GATHER | SPMV_HIP_FLAG_BALANCE_ENTRIES | SPMV_HIP_FLAG_NO_SEGMENT_WINDOW | SPMV_HIP_FLAG_FUSED_PEER_STORE |
    SPMV_HIP_FLAG_NO_BLOCK_TILES | SPMV_HIP_FLAG_NO_MULTI_WINDOW | SPMV_HIP_FLAG_NO_MASKED_BLOCKS | SPMV_HIP_FLAG_PIPELINE_GATHER
#ifdef SPMV_HIP_EXPERIMENTS
    | 0x2000u | 0x4000u | 0x30000u // timing experiments of tools/kernel_sweep.py (libspmv_hip_experiments.so only)
    | SPMV_HIP_FLAG_HUB_COLUMNS | SPMV_HIP_FLAG_ROW_GROUPS // kernel families that were measured SLOWER than the default path (below)
#endif
    ;


constexpr int kEllInPlaceMaxLength = 80;
constexpr int kBlock = 256;
constexpr int kTile = 2048;
// wavetile: rows longer than kSplitThreshold entries are cut into kSplitChunk-entry
// chunks handled by different waves (each adds its partial sum with one atomic)
// (round 3: 2048 / 1024 -> 512 / 512.  One wave striding a 2000-entry row -- four gathers in flight per lane, eight dependent
// trips -- was the tail of the whole launch on a web graph: webbase-like 27.1 -> 24.1 us, power law 43.1 -> 40.9 us,
// profiles/r03_split_rows.log)
constexpr int kSplitThreshold = 512;
constexpr int kSplitChunk = 512;

inline bool aligned16(const void * p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// compute units of the current device (MI355X: 256); asked once per device, 256 if the query fails
int cu_count();
int grid_for(long long work_items, int per_block, int max_blocks = 0);

} // namespace spmvi

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
    int32_t * d_rest_tiles = nullptr; // with block / segment windows: the tiles NOT marked for them (what csr_wavetile_kernel<LIST> multiplies)
    int nrest_tiles = 0;
    // row-group plans (csr_rowgroup.hpp): the tiles csr_rowgroup_kernel multiplies, and the others (csr_wavetile_kernel<LIST>)
    int32_t * d_group_tiles = nullptr, * d_group_rest = nullptr;
    int ngroup_tiles = 0, ngroup_rest = 0;
    // segment windows (csr_segwin.hpp): x staged through LDS per block of seg_tiles_per_block tiles, in up to 12 column segments
    spmv::SegWinBlock * d_segblocks = nullptr;
    int nsegblocks = 0, seg_tiles_per_block = 0, segwin_tiles = 0, segwin_slots = 0;
    int32_t * d_patterns = nullptr; // shared window-of-runs layouts (kernels: kPatStride words each)
    int npatterns = 0;
    int uniform_tiles = 0; // tiles whose rows are all equally long: row_ptr is not read for them
    int split_rows = 0;    // rows cut into chunks that are added to y with atomics
    bool balanced = false; // tiles filled by entries, row sums by segmented reduction (csr_segtile_kernel)
    // hub columns (csr_hub.hpp; balanced plans of graph matrices): the plan's own column stream with hubs renumbered, the hub
    // columns, and the dense copy of their x entries -- SCRATCH written by every multiply: multiplies through one plan must be
    // ordered (one stream, or the caller's own ordering), like the runs of a context
    int32_t * d_colh = nullptr;
    int32_t * d_hub_column = nullptr;
    double * d_hubx = nullptr;
    int nhubs = 0, hub_threshold = 0;
    long long hub_entries = 0;
    // block tiles (csr_blocktile.hpp): rows in triples of equal length (a hint from row_ptr at plan time), checked against
    // the columns and marked by spmv_hip_plan_csr_repack; their block stream lives behind the 16-bit columns in d_col16
    int multi_window_tiles = 0; // tiles of several long rows walked in windows of 512 entries
    int block_hint = 0;   // 3: rows in triples (3 x 3 blocks); 2 or 4: rows in groups of that many equally long rows (group tiles)
    int block_offset = 0; // the row (0 ... block_hint - 1) at which the grid of triples / groups starts
    bool block_skewed = false; // block_hint == 3 read from rows whose lengths go (m + 1, m + 2, m + 3) or back: the stored triangle of a matrix of
                               // 3 x 3 blocks (a `symmetric` Matrix Market file, multiplied as stored) -- row lengths grow with the neighbours
                               // numbered in front of a node, so tiles are cut by the block tile's limits, not by the lanes a plain tile needs
    int hints_tried = 0;  // bit 2: the triangular triples were wrong; bit 0: the triples read from row_ptr were wrong (repack found no blocks), bit 1: the groups of 2 / 4 were
    bool hint_from_bits = false; // block_hint stands on row groups found in the columns (group_bits), not on row_ptr alone
    int colshare_tiles = 0; // group tiles (csr_blocktile.hpp): tiles whose rows share one column list per group of block_hint rows
    long long colshare_entries = 0;
    int block_candidate = 0; // rows in triples of merely similar length: repack samples the columns before it believes in blocks
    const uint32_t * group_bits = nullptr; // ... and has confirmed them: bit r = row r begins a group of rows with the same columns
                                           // (host memory, alive only while repack cuts the tiles once more)
    int block_cuts = 0; // tiles the hint made shorter (0: the tiling is what it would have been without the hint)
    int break_rows = 0; // (what the tiles were built with: a rebuild needs them again)
    int block_tiles = 0;
    long long block_entries = 0;
    int stencil_mask_tiles = 0; // tiles whose rows follow a stencil pattern with positions missing (csr_stenciltile.hpp), and their entries
    long long stencil_mask_entries = 0;
    int masked_block_tiles = 0; // ... of which: blocks with entries missing or off the grid of column triples (a 32-bit word per block)
    long long masked_block_entries = 0;
    size_t meta_bytes = 0;
    // what one multiply streams with the tile classes chosen (plan_account): roofline bookkeeping
    long long streamed_bytes = 0, shifted_entries = 0, narrow_entries = 0, uniform_rows = 0;
    // value dictionary (spmv_hip_plan_csr_index_values): one byte per stored entry + the distinct values
    uint8_t * d_vidx = nullptr;
    double * d_vtab = nullptr;            // kMaxIndexedValues doubles
    int nvalues = 0;                      // 0 = no dictionary
    int value_row_tiles = 0; // tiles of the dictionary kernel that read no index stream (kTileMetaValueRows)
    int4 * d_tiles_vi = nullptr; // the dictionary launch's own descriptors: runs of constant-row tiles re-cut into tiles of 128 rows
    int ntiles_vi = 0;           // (0 = it uses d_tiles)
    const double * values_from = nullptr; // the value array it was made from
    const double * values_wanted = nullptr; // the array spmv_hip_plan_csr_index_values was last called with, dictionary built or not:
                                            // a repack that frees the plan of its window kernels asks again (round 6)
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
    // SPMV_HIP_FLAG_FUSED_PEER_STORE: where this part's rows live in the OTHER devices' copies of y; a run then delivers
    // them itself (the multiply kernel's own stores, or a push behind it)
    std::vector<double *> peer_y;
    // SPMV_HIP_FLAG_PIPELINE_GATHER: a part's CSR run reads the old y from here and writes the new one to d_y (null: in place)
    const double * y_in_override = nullptr;
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
    // SPMV_HIP_FLAG_PIPELINE_GATHER: a second copy of y per device, a second (gather) stream per device, and the events that
    // order them.  Run k reads slot g of ybuf(cur) and writes slot g of ybuf(cur ^ 1) on the part's stream, then the gather
    // of ybuf(cur ^ 1) is enqueued on comm[g] behind ev_mul[g]; run k + 1's multiply only waits for the gather that last
    // SENT the copy it is about to write (ev_sent[b][g]: recorded on comm[g] after the gather out of copy b).
    bool pipeline = false;          // asked for and possible for the current upload
    std::vector<double *> yfull2;
    int cur = 0;                    // which copy holds the current y (0: yfull, 1: yfull2)
    std::vector<hipStream_t> comm;
    std::vector<hipEvent_t> ev_mul;
    std::vector<hipEvent_t> ev_sent[2];
    bool sent_recorded[2] = {false, false};
    bool peer_gather = false; // SPMV_HIP_FLAG_PEER_GATHER: slots are pushed to the other devices by a kernel, no RCCL
    void * rccl_lib = nullptr;
    std::vector<ncclComm_t> comms;
    ncclResult_t (*p_all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_group_start)() = nullptr;
    ncclResult_t (*p_group_end)() = nullptr;
    ncclResult_t (*p_comm_destroy)(ncclComm_t) = nullptr;
    const char * (*p_error_string)(ncclResult_t) = nullptr;
    ncclResult_t (*p_comm_count)(const ncclComm_t, int *) = nullptr;
};

namespace spmvi {

// plan_csr.hip
int plan_csr_internal(spmv_hip_plan ** out, int32_t rows, int32_t cols, const int32_t * p, int algorithm, int lanes_per_row,
                      unsigned flags, int32_t break_rows);
int device_index_check(const int32_t * d_idx, long long n, int limit, bool want_sorted, bool * bad, bool * sorted, hipStream_t s);
int verify_plan(const spmv_hip_plan * pl, const int32_t * d_column_index, hipStream_t s);
int verify_plan_values(const spmv_hip_plan * pl, const double * d_value, hipStream_t s);

// context.hip
void free_ctx_matrix(spmv_hip_ctx * c);

// multi_gpu.hip
void multi_free_matrix(spmv_hip_ctx * c);
int multi_upload_csr(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_ptr,
                     const int32_t * column_index, const double * value);
int multi_upload_ell(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length, const int32_t * column_index, const double * value);
int multi_upload_coo(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t nnz, const int32_t * row_index, const int32_t * column_index,
                     const double * value);
int multi_upload_hybrid(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length, const int32_t * ell_col, const double * ell_val,
                        int32_t ncoo, const int32_t * coo_row, const int32_t * coo_col, const double * coo_val);
int multi_set_x(spmv_hip_ctx * c, const double * x);
int multi_set_y(spmv_hip_ctx * c, const double * y);
int multi_get_y(spmv_hip_ctx * c, double * y);
int multi_run(spmv_hip_ctx * c);
int multi_sync(spmv_hip_ctx * c);
int multi_times(spmv_hip_ctx * c, uint64_t * kernel_ns, uint64_t * gather_ns);

} // namespace spmvi

// coo_sort.hip
extern "C" int spmv_hip_internal_exclusive_scan_i32(const int32_t * d_in, int32_t * d_out, long long n, hipStream_t s);
extern "C" int spmv_hip_internal_coo_panels(int32_t cols, int32_t nnz, const int32_t * d_row, const int32_t * d_col, const double * d_val,
                                            int32_t ** out_row, int32_t ** out_col, double ** out_val, long long * start, hipStream_t s);
