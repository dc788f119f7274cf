/*
 * spmv_hip.h -- C ABI of the MI355X (gfx950) SpMV engine.
 *
 * This is the drop-in boundary for ONE path of jamtrott/spmv-cache-trace: the
 * `y += A*x` kernels behind its `Kernel::run()` plug-in interface
 * (reference src/kernels/kernel.hpp:18-45).  Each entry point names the
 * reference function it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - plain C types only; opaque handles; int32 indices and fp64 values exactly
 *     as the reference stores them (src/matrix/csr-matrix.hpp:15-17)
 *   - every function returns 0 (SPMV_HIP_OK) or a negative SPMV_HIP_ERR_* code;
 *     nothing throws across this boundary; spmv_hip_last_error() gives the
 *     detail string of the calling thread's last failure
 *   - all kernels ACCUMULATE: y += A*x (csr-matrix-spmv.cpp:32,
 *     coo-matrix.cpp:268, ell-matrix.cpp:257); y is never zeroed here
 *   - "host" pointers are borrowed for the duration of the call; "device"
 *     pointers must be 16-byte aligned hipMalloc'ed (or torch) memory on the
 *     current device
 *   - there is NO CPU fallback: without a usable GPU every compute entry point
 *     fails with SPMV_HIP_ERR_NO_DEVICE / SPMV_HIP_ERR_HIP
 *   - a ctx / plan is not thread-safe; call it from one thread (the reference's
 *     harness calls run() from every OpenMP thread, src/profile-kernel.cpp:160:
 *     the adapters in host/ funnel that to the master thread)
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_HIP_VERSION 130 /* 1.3.0: segment windows, several lanes per long ELLPACK row, peer stores for one process per GPU
                                (spmv_hip_ipc_*, spmv_hip_peer_push, spmv_hip_csr_spmv_out_peers) */

/* ---- error codes ---------------------------------------------------------- */
#define SPMV_HIP_OK 0
#define SPMV_HIP_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, bad row_ptr) */
#define SPMV_HIP_ERR_NO_DEVICE (-2) /* no HIP device visible */
#define SPMV_HIP_ERR_HIP (-3)       /* a HIP runtime call failed; see spmv_hip_last_error() */
#define SPMV_HIP_ERR_ALLOC (-4)     /* host or device allocation failed */
#define SPMV_HIP_ERR_STATE (-5)     /* call order: no matrix uploaded, wrong format, ... */
#define SPMV_HIP_ERR_OVERFLOW (-6)  /* rows*row_length does not fit int32 (ell-matrix.cpp:199-205) */
#define SPMV_HIP_ERR_ALIGN (-7)     /* device pointer not 16-byte aligned */

/* ---- CSR algorithms --------------------------------------------------------- */
#define SPMV_HIP_CSR_AUTO 0     /* pick from the row-length statistics of the matrix */
#define SPMV_HIP_CSR_SCALAR 1   /* one lane per row; sums in the reference's order: bit-exact */
#define SPMV_HIP_CSR_VECTOR 2   /* 2..64 lanes per row + DPP/ds_swizzle wave reduce */
#define SPMV_HIP_CSR_ADAPTIVE 3 /* row blocks: coalesced stream of col/val -> products in LDS ->
                                   per-row sums (reference order when a row gets one lane);
                                   rows longer than a tile get a whole workgroup */
#define SPMV_HIP_CSR_WAVETILE 4 /* per-wavefront row ownership: tiles of <= 128 rows owned by one
                                   wave, descriptor-driven so all of a tile's loads issue at once,
                                   products in the wave's LDS slice, no workgroup barrier; very
                                   long rows are split over several waves (fp64 atomics).  After
                                   spmv_hip_plan_csr_compress / _repack the tiles are specialised by
                                   structure: 16-bit columns, shifted tiles and patterns, x windows,
                                   block windows, column panels */

/* plan / ctx flags */
#define SPMV_HIP_FLAG_XCD_REMAP 0x1u   /* give each XCD one contiguous run of tiles instead of the round-robin
                                           deal (measured SLOWER on MI355X for streaming SpMV: off by default) */
#define SPMV_HIP_FLAG_EXACT_ORDER 0x2u  /* force one lane per row everywhere (bit-exact, slower on long rows); for ELLPACK
                                           uploads: rows of more than 16 entries too (default: 2..64 lanes per such row,
                                           1e-10 class; rows of <= 16 entries are bit-exact either way) */
#define SPMV_HIP_FLAG_NO_INDEX_COMPRESSION 0x10u /* ctx: keep 32-bit column indices for every tile */
#define SPMV_HIP_FLAG_COO_KEEP_ORDER 0x20u /* ctx: keep COO triplets in file order on the device */
#define SPMV_HIP_FLAG_READ_ROW_PTR 0x40u /* wavetile: read row_ptr even for tiles whose rows are all equally long
                                            (by default their row bounds come from the tile descriptor) */
#define SPMV_HIP_FLAG_ROWS64 0x80u       /* wavetile: at most 64 rows per tile ... */
#define SPMV_HIP_FLAG_ROWS128 0x100u     /* ... or up to 128 (lanes own two short rows); default: 128 once the matrix
                                            exceeds ~512 MB (streams from HBM), 64 while it is cache-resident */
#define SPMV_HIP_FLAG_ELL_COLUMN_MAJOR 0x200u /* ctx: always transpose ELLPACK to column-major and use the one-lane-per-row
                                                kernel (bit-exact for any row length).  Default: the row-major arrays in
                                                place as wave tiles for EVERY row length (several lanes per row of more than
                                                16 entries: 1e-10 class; rows of 161..1024 entries in multi-window tiles,
                                                longer rows a wave each in registers -- round 5; until round 4 rows of more
                                                than 2048 entries took the column-major kernel); with
                                                SPMV_HIP_FLAG_EXACT_ORDER the column-major kernel takes rows of
                                                more than 80 entries.  Which path an upload took: spmv_hip_ctx_info [17] */
#define SPMV_HIP_FLAG_NO_SHIFTED_TILES 0x400u /* plan_csr_compress: do not look for tiles whose rows all repeat the first
                                                 row's columns shifted by the row distance (stencil interiors, bands);
                                                 such tiles read one row of column offsets instead of all of them */
#define SPMV_HIP_FLAG_NO_X_WINDOW 0x800u /* wavetile: never stage x through LDS.  Default after plan_csr_compress: per-wave
                                            windows where most tiles' x entries fit 256 slots and are used twice (narrow
                                            bands, stencils), and a per-workgroup ring for unstructured bands whose 16-tile
                                            blocks span <= 8192 columns (a second kernel launch per multiply) */
#define SPMV_HIP_FLAG_NO_COLUMN_PANELS 0x1000u /* plan_csr_repack / upload_csr / upload_coo / upload_hybrid: never form
                                                  column panels (a copy of a scattered matrix cut into 8 column ranges,
                                                  one per group of workgroups that share an XCD's L2) */
#define SPMV_HIP_FLAG_BIG_TILE 0x8u     /* wavetile: 1024-entry tiles instead of 512 */
#define SPMV_HIP_FLAG_VERIFY_PLAN 0x8000u /* spmv_hip_csr_spmv: re-check on EVERY call that the column array still has the
                                             contents the plan was compressed from (one extra pass over it per multiply;
                                             by default this is checked on the first multiply only, see spmv_hip_plan_verify).
                                             On a spmv_hip_create_multi context also: spmv_hip_get_y fetches EVERY device's copy of
                                             y and returns SPMV_HIP_ERR_STATE unless they are identical bit for bit */
#define SPMV_HIP_FLAG_NO_BALANCED_TILES 0x40000u /* wavetile: never switch to tiles filled by entries (up to 512 in up to 256
                                             rows, row sums by segmented reduction: csr_segtile_kernel).  By default a matrix
                                             whose row-owned tiles come out less than half full because its rows are skewed
                                             (longest row > 16 entries) gets them: a web graph runs in a fifth of the waves.
                                             Rows that span lanes are then added in another order than the reference's
                                             (1e-10, not bit-identical; SPMV_HIP_FLAG_EXACT_ORDER also keeps row-owned tiles) */
#define SPMV_HIP_FLAG_NO_RUN_EVENTS 0x80000u /* ctx: spmv_hip_run does not bracket the launch with a HIP event pair (each
                                             record is a barrier packet between back-to-back runs, ~5 us per run);
                                             spmv_hip_last_run_ns then returns SPMV_HIP_ERR_STATE.  For callers that time a
                                             whole region themselves (bench.py); the Kernel adapters keep the events */
#define SPMV_HIP_FLAG_NO_VALUE_INDEX 0x100000u /* never build a value dictionary (spmv_hip_plan_csr_index_values is a no-op; the
                                              context does not build one for its uploads) */
#define SPMV_HIP_FLAG_PEER_GATHER 0x200000u /* spmv_hip_create_multi: gather the y slots with remote stores instead of RCCL: after
                                             its multiply every device runs one kernel that writes its slot into the
                                             other devices' y over xGMI (peer access must be available between all
                                             devices; librccl.so is then never loaded).  Same result, bit for bit. */
#define SPMV_HIP_FLAG_BALANCE_ENTRIES 0x400000u /* spmv_hip_create_multi: cut the rows where the stored entries divide evenly
                                             (boundary g = first row whose row_ptr reaches g * nnz / G) instead of
                                             the reference's ceil(rows / G) rows per device; y slots are as long as
                                             the longest block.  For matrices whose row lengths differ between the
                                             top and the bottom (a KKT system's two row populations). */
#define SPMV_HIP_FLAG_FUSED_PEER_STORE 0x1000000u /* spmv_hip_create_multi: like SPMV_HIP_FLAG_PEER_GATHER (no RCCL, peer access needed), but
                                             where a device's multiply is the default CSR kernel it stores every row sum into all
                                             G copies of y itself as each tile finishes -- the gather overlaps the SAME multiply
                                             and needs no launch of its own; other kernels are followed by the push kernel */
#define SPMV_HIP_FLAG_NO_SEGMENT_WINDOW 0x800000u /* plan_csr_compress: no segment windows (x staged through LDS per block of 32
                                             tiles in up to 12 far-apart column segments: rows of a 3-D mesh in natural
                                             ordering, KKT systems; the tiles' 16-bit column stream then holds window
                                             slots).  Unstructured bands fall back to the one-ring block window. */
#define SPMV_HIP_FLAG_NO_BLOCK_TILES 0x2000000u /* plan_csr / plan_csr_repack: no block tiles.  By default a matrix whose rows come in
                                             triples of equal length (3 unknowns per mesh node: finite-element elasticity) gets its
                                             tiles cut on triple boundaries, and plan_csr_repack marks every tile that really consists
                                             of dense 3 x 3 blocks (checked entry by entry): such a tile reads one 16-bit number per
                                             BLOCK instead of a column index per entry and no row_ptr (8.2 instead of 10 bytes per
                                             entry).  Only rows of more than 16 entries (1e-10 class either way); never under
                                             SPMV_HIP_FLAG_EXACT_ORDER.  The flag also switches off the GROUP TILES of meshes with 2 or 4
                                             unknowns per node (rows in groups of 2 or 4 equally long rows with the same columns: one
                                             16-bit column list per group, one 16-byte gather of x per pair of adjacent columns;
                                             plan_info[35..37]) and every hint read from row_ptr for either */
/* 0x4000000u and 0x10000000u are not flags of this library: two kernel families that were measured SLOWER than the paths they
 * were meant to replace (hub columns for web graphs, 26.6 vs 23.9 us; a lane group per row for stencil rows of 17 ... 64
 * entries, 797 vs 740 us: DESIGN.md sections 3.3, 3.1b) were retired from the product library in round 5 and are
 * refused like any unknown bit.  They live on in libspmv_hip_experiments.so (csrc/internal.hpp) for tools/ and
 * tests/experiments/. */
#define SPMV_HIP_FLAG_NO_MULTI_WINDOW 0x8000000u /* plan_csr: no multi-window tiles.  By default rows of 161 ... 1024 entries, which fill a
                                             512-entry tile badly (one row of 361: 70 %) or do not fit one at all,
                                             are taken two to eight at a time by one wave that walks
                                             them in windows of 512 entries and carries the row sums in registers (7 rows of 361 =
                                             4.94 windows): no atomics, the same y on every run.  1e-10 class like every row of more than 16 entries;
                                             never under SPMV_HIP_FLAG_EXACT_ORDER */
#define SPMV_HIP_FLAG_NO_MASKED_BLOCKS 0x20000000u /* plan_csr_repack: no MASKED block tiles.  By default a tile of row triples that is
                                             not made of dense, aligned 3 x 3 blocks -- explicit zeros dropped from some blocks, a node
                                             with one or two unknowns that shifts the grid of column triples, rows of a triple that
                                             differ in length -- is covered greedily with blocks of three consecutive columns and a
                                             9-bit mask each (a 32-bit word per block instead of a column index per entry) if that
                                             takes at most 64 blocks holding 6 stored entries on average; with this flag such a tile
                                             keeps its 16-bit columns (round 4's behaviour: one broken block demotes its tile) */
/* Any other bit is refused with SPMV_HIP_ERR_INVALID by spmv_hip_create and spmv_hip_plan_csr. */

typedef struct spmv_hip_ctx spmv_hip_ctx;
typedef struct spmv_hip_plan spmv_hip_plan;

/* ---- library ------------------------------------------------------------------ */
int spmv_hip_version(void);
const char *spmv_hip_strerror(int code);
const char *spmv_hip_last_error(void);
/* Number of visible HIP devices (0 and SPMV_HIP_OK when there are none). */
int spmv_hip_device_count(int *count);

/* =================================================================================
 * Level 1 -- context API: host arrays in, host arrays out.
 * Bound by the hip_{csr,coo,ell,hybrid}_spmv_kernel adapters (host/kernels/spmv-kernels.cpp),
 * which stand where the reference's csr_spmv_kernel / coo_spmv_kernel /
 * ell_spmv_kernel stand (src/kernels/{csr,coo,ell}-spmv.cpp).
 * ============================================================================== */

/* Create a context on `device` with its own stream.  Replaces nothing in the
 * reference (it has no device); called from Kernel::init. */
int spmv_hip_create(spmv_hip_ctx **ctx, int device, unsigned flags);
void spmv_hip_destroy(spmv_hip_ctx *ctx);

/* One context over `num_gpus` devices (0 .. num_gpus-1) of this process: SURVEY 8b / 8e.
 * spmv_hip_upload_csr then cuts the rows into num_gpus contiguous blocks by the reference's static
 * rule, chunk = ceil(rows / num_gpus) (src/matrix/csr-matrix.cpp:77-95, src/matrix/csr-matrix-spmv.cpp:154-161:
 * devices take the place of OpenMP threads), device g holds block g, the whole x, and its own copy of
 * the whole y.  spmv_hip_run multiplies on every device (y_g += A_g x into slot g of that device's y)
 * and then assembles y everywhere with ONE in-place ncclAllGather (RCCL over xGMI) issued for all
 * devices inside a group call; spmv_hip_sync waits for all of them; spmv_hip_get_y reads device 0's y.
 * spmv_hip_upload_ell cuts the row-major ELLPACK arrays into the same row blocks, spmv_hip_upload_coo deals the
 * triplets (any order) to the blocks of their rows by a stable pass and rebases the row indices;
 * spmv_hip_upload_hybrid does both (ELLPACK part by rows, remainder dealt to the blocks) and every device merges its two
 * parts into one row-major matrix like a single-device hybrid upload.  spmv_hip_set_stream is refused.  librccl.so is loaded at run time, and only when num_gpus > 1 (or SPMV_HIP_FORCE_RCCL=1
 * is set, which runs the collective with one device too); num_gpus = 1 is an ordinary context with
 * the same calling sequence.
 * With SPMV_HIP_FLAG_PEER_GATHER the gather is done without RCCL: every device pushes its slot into the other
 * devices' y with one kernel of remote stores (one copy per xGMI link); y is complete on every device after
 * spmv_hip_sync.  Under that flag the environment variable SPMV_HIP_SHARE_DEVICES=1 lets num_gpus exceed the
 * number of visible devices (part g then runs on device g mod visible): a rehearsal of the G-way partition,
 * slots and gather on a smaller machine -- the arithmetic is the same, the timing means nothing. */
int spmv_hip_create_multi(spmv_hip_ctx **ctx, int num_gpus, unsigned flags);

/* Enqueue everything this context does from now on on the caller's `stream` (a hipStream_t on the
 * context's device; NULL = the default stream), or, with use_own != 0, on the context's own stream
 * again.  Lets a host program order the multiply against its own work (and time it with its own
 * events: bench.py).  Waits for the stream in use before switching. */
int spmv_hip_set_stream(spmv_hip_ctx *ctx, void *stream, int use_own);

/* CSR algorithm for later uploads (default SPMV_HIP_CSR_AUTO);
 * lanes_per_row: 0 = choose, else 2,4,...,64 for SPMV_HIP_CSR_VECTOR. */
int spmv_hip_set_csr_algorithm(spmv_hip_ctx *ctx, int algorithm, int lanes_per_row);

/* Copy a CSR matrix to the device and build its launch plan.
 * Takes what csr_matrix::Matrix holds (src/matrix/csr-matrix.hpp:58-64):
 * row_ptr[rows+1], column_index[row_ptr[rows]], value[row_ptr[rows]].
 * `nnz` is row_ptr[rows] (padding entries of a row-aligned matrix included). */
int spmv_hip_upload_csr(spmv_hip_ctx *ctx, int32_t rows, int32_t cols, int32_t nnz,
                        const int32_t *row_ptr, const int32_t *column_index,
                        const double *value);

/* COO in file order (src/matrix/coo-matrix.hpp:65-70). */
int spmv_hip_upload_coo(spmv_hip_ctx *ctx, int32_t rows, int32_t cols, int32_t nnz,
                        const int32_t *row_index, const int32_t *column_index,
                        const double *value);

/* ELLPACK in the reference's ROW-MAJOR padded layout, k = i*row_length + l
 * (src/matrix/ell-matrix.cpp:253-256).  The arrays are used in place as uniform wave tiles, padding multiplied
 * like real entries.  Summation order: rows of up to 16 entries are added by one lane in the reference's order
 * (bit-exact); longer rows by 2..64 lanes (1e-10 class) -- rows of 161..1024 entries in multi-window tiles, longer
 * rows a wave each (a small matrix: a few waves per row meeting in fp64 atomics).  The reference's order for every
 * row length: SPMV_HIP_FLAG_EXACT_ORDER or SPMV_HIP_FLAG_ELL_COLUMN_MAJOR (a column-major copy, one lane per row).
 * Behaviour changes: round 4 (rows of 161..2048 entries lost bit-exactness by default), round 5 (rows of more than
 * 2048 entries too: they no longer take the column-major kernel unless asked). */
int spmv_hip_upload_ell(spmv_hip_ctx *ctx, int32_t rows, int32_t cols, int32_t row_length,
                        const int32_t *column_index, const double *value);

/* Hybrid ELLPACK + COO (src/matrix/hybrid-matrix.hpp:83-96): the ELL part in the reference's
 * row-major layout (rows*ell_row_length entries), the remainder as COO triplets.  One run
 * launches the ELL kernel and then the COO kernel on the same stream
 * (hybrid_matrix::spmv, src/matrix/hybrid-matrix.cpp:535-567). */
int spmv_hip_upload_hybrid(spmv_hip_ctx *ctx, int32_t rows, int32_t cols, int32_t ell_row_length,
                           const int32_t *ell_column_index, const double *ell_value,
                           int32_t num_coo_entries, const int32_t *coo_row_index,
                           const int32_t *coo_column_index, const double *coo_value);

/* x has `cols` doubles, y has `rows` doubles (src/kernels/csr-spmv.cpp:35-36). */
int spmv_hip_set_x(spmv_hip_ctx *ctx, const double *x);
int spmv_hip_set_y(spmv_hip_ctx *ctx, const double *y);
int spmv_hip_get_y(spmv_hip_ctx *ctx, double *y);

/* Enqueue one y += A*x on the context's stream.  Replaces
 * csr_matrix::spmv (csr-matrix-spmv.cpp:148-167), coo_matrix::spmv
 * (coo-matrix.cpp:313-335), ell_matrix::spmv (ell-matrix.cpp:311-335). */
int spmv_hip_run(spmv_hip_ctx *ctx);
/* Block until the stream is idle: must precede the harness's closing barrier
 * (src/profile-kernel.cpp:161). */
int spmv_hip_sync(spmv_hip_ctx *ctx);
/* --flush-caches on the device: what flush_cache(10 x the largest cache) does for the CPU between two timed runs
 * (src/profile-kernel.cpp:181-192, :264).  Streams through a scratch buffer of four times the 256 MB Infinity Cache
 * (allocated on first use) on every device of the context and waits, so that the next run finds neither its
 * matrix nor its vectors in the L2 or the Infinity Cache.  Not part of a run; never timed. */
int spmv_hip_flush_caches(spmv_hip_ctx *ctx);
/* Device time of the last spmv_hip_run (hipEvent pair), valid after a sync. */
int spmv_hip_last_run_ns(spmv_hip_ctx *ctx, uint64_t *kernel_ns);
/* The same with the collective apart: kernel_ns = the slowest device's multiply, gather_ns
 * = the longest span from the end of a device's multiply to the end of its all-gather (0 for a
 * single-device context).  Either pointer may be NULL. */
int spmv_hip_last_run_times(spmv_hip_ctx *ctx, uint64_t *kernel_ns, uint64_t *gather_ns);

/* Descriptive numbers for JSON output / tests.  out[] receives up to n of:
 * [0] format (0 none, 1 csr, 2 coo, 3 ell, 4 hybrid)  [1] rows  [2] cols  [3] stored entries
 * [4] csr algorithm in use  [5] lanes per row (vector)  [6] workgroups per launch
 * [7] row blocks / tiles  [8] long-row blocks  [9] device bytes held  [10] tiles with 16-bit columns
 * [11] shifted tiles  [12] tiles with an x window  [13] block-window tiles  [14] tiles of the
 * column-panel copy (see spmv_hip_plan_info)  [15] bytes one run streams with the tile classes in
 * use (see spmv_hip_plan_info [14]; formats without tiles: their algorithmic bytes)
 * [16] devices (1, or the num_gpus of spmv_hip_create_multi: [6..15] are then sums over the devices)
 * [17] ELLPACK path of the upload: 0 = not ELLPACK, 1 = the row-major arrays in place (wave tiles), 2 = column-major
 *      copy (one lane per row, the reference's order) */
int spmv_hip_ctx_info(spmv_hip_ctx *ctx, int64_t *out, int n);

/* =================================================================================
 * Level 2 -- device-pointer API: the caller owns device memory and the stream
 * (bench.py and the tests pass torch tensors' data_ptr() and torch's stream).
 * `stream` is a hipStream_t (NULL = default stream).
 * ============================================================================== */

/* Build the launch plan of a CSR matrix from its HOST row_ptr (the only part of
 * the matrix the schedule depends on).  Allocates a few KB of device metadata on
 * the current device. */
int spmv_hip_plan_csr(spmv_hip_plan **plan, int32_t rows, int32_t cols,
                      const int32_t *host_row_ptr, int algorithm, int lanes_per_row,
                      unsigned flags);
/* Optional second planning step for the wave-tile algorithm (done automatically by
 * spmv_hip_upload_csr): one pass over the device column indices that classifies every tile.
 *  - all columns within 65536 of the smallest one: kept as 16-bit offsets in a plan-owned index
 *    stream (2 extra bytes per entry of device memory); the tile reads 10 instead of 12 bytes/entry;
 *  - equally long rows that repeat the first row's columns shifted by the row distance (stencil
 *    interiors, bands; any column range): only the first row's columns are read, 8 bytes/entry;
 *  - tiles whose x entries fit 256 LDS slots and are each used at least twice: x staged through LDS;
 *  - blocks of 16 plain narrow tiles whose columns span <= 8192: marked for the block-window kernel.
 * Results are unchanged bit for bit.  The plan then expects the same d_column_index in spmv_hip_csr_spmv (a different
 * pointer falls back to the 32-bit indices and uses nothing derived here).  Because a pointer can be
 * the same while the contents are not (an allocator reusing the address for another matrix), the plan
 * keeps a 64-bit checksum of the column array: it is re-computed and compared on the first multiply
 * after this call, on every multiply with SPMV_HIP_FLAG_VERIFY_PLAN, and by spmv_hip_plan_verify;
 * a mismatch is SPMV_HIP_ERR_STATE, never a silent wrong result.  Synchronises `stream`.
 * THE FIRST MULTIPLY after this call (and after spmv_hip_plan_csr_index_values) therefore contains one checksum
 * pass and a hipStreamSynchronize; call spmv_hip_plan_verify beforehand to have the check outside a timed or
 * latency-sensitive first call.  The check is skipped (left pending) while `stream` is being captured into a
 * graph.  Two host threads may share a plan: the pending check is claimed atomically by one of them. */
int spmv_hip_plan_csr_compress(spmv_hip_plan *plan, const int32_t *d_column_index, void *stream);
/* Optional, BEFORE spmv_hip_plan_csr_compress (spmv_hip_upload_csr does it): a plan whose rows are mostly longer than 16 entries and about
 * as long as their successors is a CANDIDATE for (masked) block tiles -- three unknowns per mesh node, with or without entries
 * missing.  This call looks at the columns (which rows have the same columns as the row in front of them?) and, if half of the
 * rows stand in groups of three, cuts the plan's tiles once more on those groups, so that they are classified ONCE; without it
 * spmv_hip_plan_csr_repack does the same after the classification and classifies again (queen-like with broken blocks: 51 instead
 * of ~35 ms of plan time).  host_row_ptr: the array given to spmv_hip_plan_csr, or NULL (row_ptr is then fetched back from the
 * device).  A no-op for every other plan, and after spmv_hip_plan_csr_compress.  Synchronises `stream`. */
int spmv_hip_plan_csr_confirm_blocks(spmv_hip_plan *plan, const int32_t *d_row_ptr, const int32_t *d_column_index,
                                     const int32_t *host_row_ptr, void *stream);
/* Content guard on demand: SPMV_HIP_OK if d_column_index is not the array the plan was compressed from
 * (nothing derived will be used) or still has the same contents; SPMV_HIP_ERR_STATE if the contents
 * changed.  One pass over the array; synchronises `stream`. */
int spmv_hip_plan_verify(spmv_hip_plan *plan, const int32_t *d_column_index, void *stream);
/* Optional third planning step (done automatically by spmv_hip_upload_csr), after
 * spmv_hip_plan_csr_compress: for a matrix whose columns are scattered (in most tiles they reach
 * further than an eighth of the matrix, and the tiles are not shifted ones), with at least 4 entries per row, at least 2^20 entries and an x larger than one XCD's L2, the plan makes its
 * own copy of the matrix cut into 8 column panels -- one per group of workgroups that share an XCD --
 * so that every XCD gathers from one eighth of x out of its private L2; a row's partial sums are
 * added to y with fp64 atomics (order not reproducible; within the usual tolerance).  The copy is
 * used by spmv_hip_csr_spmv when it is called with the same d_column_index and d_value; the VALUES
 * ARE SNAPSHOTTED: after changing them call spmv_hip_plan_csr_repack on a fresh plan, or pass
 * SPMV_HIP_FLAG_NO_COLUMN_PANELS.  Does nothing (returns 0) when the matrix does not qualify;
 * plan_info[13] tells.  Costs 12 bytes per entry + 32 bytes per row of device memory.  Synchronises.
 * The same step -- it is the one that sees row_ptr next to the columns -- marks BLOCK TILES (plan_info[25]): in a matrix
 * whose rows come in triples of equal length (spmv_hip_plan_csr noticed that from row_ptr and cut its tiles on triple
 * boundaries) every tile of rows longer than 16 entries is checked entry by entry for dense 3 x 3 blocks; a tile that
 * has them reads one 16-bit number per block from a plan-owned side stream instead of a column index per entry, and no
 * row_ptr.  The CSR arrays are read in place; y stays within 1e-10 (such rows are summed by several lanes either way).
 * Likewise GROUP TILES (plan_info[35]): where the rows come in groups of 2 or 4 equally long rows (2 or 4 unknowns per mesh node)
 * every tile is checked entry by entry for identical column lists within its groups; a tile that has them reads the first
 * row's 16-bit columns of each group from the side stream (one per PAIR where the pairs are adjacent columns) and sums its rows
 * exactly as before.  A hint from row_ptr that the columns do not bear out costs a second cut of the tiles (plan time only) and
 * hands over to the next one: triples, groups of 4 / 2, then the rows of merely similar length whose columns are looked at.
 * SPMV_HIP_FLAG_NO_BLOCK_TILES / SPMV_HIP_FLAG_EXACT_ORDER switch it off. */
int spmv_hip_plan_csr_repack(spmv_hip_plan *plan, const int32_t *d_row_ptr, const int32_t *d_column_index,
                             const double *d_value, void *stream);
/* Optional planning step for matrices with FEW DISTINCT VALUES (at most 128 different bit patterns among
 * the stored entries: pattern / graph matrices, constant-coefficient stencils, meshes of identical
 * elements): the plan keeps the distinct values in a table and one BYTE per entry saying which, and the
 * default kernel then streams 1 instead of 8 bytes of value per entry, taking the double itself from the
 * table -- the stored bits, so y is unchanged bit for bit.  Stencil tiles whose rows all carry the first row's values
 * (plan_info[23]) read those few bytes only, and the dictionary launch re-cuts runs of them into tiles of up to 128 rows
 * (plan_info[24]).  A plan whose launch would stage x through LDS runs the dictionary launch instead (measured faster).  Does nothing (returns 0, plan_info[20] == 0)
 * when the matrix has more distinct values or the plan uses another kernel (column panels, block / segment windows; balanced
 * tiles have their own dictionary variant).  BY CALLING THIS THE CALLER
 * PROMISES that d_value keeps its contents while the plan lives, or that
 * spmv_hip_plan_csr_refresh_values follows every change; the promise is checked like the one for the
 * columns (checksum on the first multiply, on every multiply with SPMV_HIP_FLAG_VERIFY_PLAN: a changed
 * array is SPMV_HIP_ERR_STATE, not a wrong y).  spmv_hip_upload_* do this by themselves: the context
 * owns its copy of the values.  Costs nnz bytes of device memory; synchronises `stream`. */
int spmv_hip_plan_csr_index_values(spmv_hip_plan *plan, const double *d_value, void *stream);
/* After changing the VALUES of a matrix whose plan holds a value dictionary (plan_info[20] > 0) or column
 * panels (plan_info[18] == 1): bring both up to date (structure unchanged; d_value may be a new array,
 * which the plan then expects; a dictionary is dropped if the values are no longer few).  Does nothing
 * when the plan has neither.  The panel copy is asynchronous on `stream`, the dictionary synchronises it. */
int spmv_hip_plan_csr_refresh_values(spmv_hip_plan *plan, const int32_t *d_row_ptr, const int32_t *d_column_index,
                                     const double *d_value, void *stream);
void spmv_hip_plan_destroy(spmv_hip_plan *plan);
/* out[]: [0] algorithm  [1] lanes per row  [2] workgroups  [3] row blocks
 *        [4] long-row blocks  [5] rows  [6] nnz  [7] metadata bytes on device
 *        [8] tiles with 16-bit column offsets (after spmv_hip_plan_csr_compress)
 *        [9] uniform tiles (all rows equally long: row_ptr not read)
 *        [10] shifted tiles (column offsets read for the first row only)
 *        [11] tiles whose column range fits a 256-entry window of x (x staged through LDS when most tiles qualify)
 *        [12] tiles multiplied by the block-window kernel (x staged through LDS per 16 tiles)
 *        [13] tiles of the column-panel copy (0 = no panels; see spmv_hip_plan_csr_repack)
 *        [14] bytes one multiply streams with the tile classes chosen: 8 B per value; per column 4 B
 *             (32-bit), 2 B (16-bit) or nothing (shifted tiles: one first row, or a cached pattern);
 *             row_ptr 4 B per row of a non-uniform tile; y 16 B per row; x once; 16 B per tile.  The
 *             ALGORITHMIC bytes of SURVEY 8(d), 12 nnz + 4 (rows + 1) + 16 rows + 8 cols, never shrink
 *        [15] stored entries in shifted tiles  [16] stored entries in 16-bit tiles
 *        [17] rows in uniform tiles  [18] 1 if the plan holds a snapshot of the values (column panels)
 *        [19] 1 if the tiles are balanced ones (filled by entries; see SPMV_HIP_FLAG_NO_BALANCED_TILES)
 *        [20] size of the value dictionary (0 = none; see spmv_hip_plan_csr_index_values)
 *        [21] tiles multiplied by the segment-window kernel (a subset of [12]; x staged through LDS per block of 32 tiles
 *             in up to 12 column segments)  [22] the largest window among its blocks, in doubles
 *        [23] with a value dictionary: tiles whose rows all repeat the first row's values (constant-coefficient stencils) --
 *             they read no index stream at all, only the first row's bytes
 *        [24] tiles of the dictionary launch when runs of such tiles were re-cut into tiles of 128 rows (0: it uses [3])
 *        [25] block tiles (dense 3 x 3 blocks: one 16-bit number per block, see spmv_hip_plan_csr_repack)  [26] their entries
 *        [27], [28] 0 in this library (hub columns and their entries in libspmv_hip_experiments.so)
 *        [29] multi-window tiles (several rows of 161 ... 1024 entries walked in windows of 512: SPMV_HIP_FLAG_NO_MULTI_WINDOW),
 *        [30] 0 in this library (row-group tiles in libspmv_hip_experiments.so)
 *        [31] of the block tiles [25]: MASKED ones (blocks with entries missing or off the grid of column triples: a 32-bit word
 *             per block, SPMV_HIP_FLAG_NO_MASKED_BLOCKS)  [32] their entries
 *        [33] masked stencil tiles (the boundary rows of a structured grid: rows that follow a stencil pattern of at most 16
 *             positions with some of them missing -- a 16-bit mask per row instead of column indices and row_ptr; marked by
 *             spmv_hip_plan_csr_repack; never with SPMV_HIP_FLAG_NO_SHIFTED_TILES)  [34] their entries
 *        [35] group tiles (rows in groups of 2 or 4 equally long rows with the same columns -- a mesh with 2 or 4 unknowns per node:
 *             one 16-bit column list and one gather of x per group instead of per row, the values in place, the row sums of the
 *             plain tile bit for bit; a hint from row_ptr in spmv_hip_plan_csr, checked against the columns and marked by
 *             spmv_hip_plan_csr_repack; never with SPMV_HIP_FLAG_NO_BLOCK_TILES or a value dictionary)  [36] their entries
 *        [37] the rows per group of those tiles (0 = none) */
int spmv_hip_plan_info(const spmv_hip_plan *plan, int64_t *out, int n);

/* y += A*x, CSR.  Replaces csr_spmv / csr_spmv_inner_loop
 * (src/matrix/csr-matrix-spmv.cpp:21-33, 63-76). */
int spmv_hip_csr_spmv(const spmv_hip_plan *plan, const int32_t *d_row_ptr,
                      const int32_t *d_column_index, const double *d_value,
                      const double *d_x, double *d_y, void *stream);
/* y_out = y_in + A*x: the same multiply reading the old y from one array and writing the new one to
 * another (they must not overlap; y_in == y_out is spmv_hip_csr_spmv).  For a row-partitioned multiply
 * whose previous y segment is still being sent (the all-gather of src/matrix/csr-matrix.cpp:77-95's
 * row blocks across GPUs): two segment buffers alternate and no copy is needed.  Plans that add
 * partial sums with atomics (split rows > 512 entries, column panels) and the non-default
 * algorithms copy y_in to y_out first. */
int spmv_hip_csr_spmv_out(const spmv_hip_plan *plan, const int32_t *d_row_ptr,
                          const int32_t *d_column_index, const double *d_value,
                          const double *d_x, const double *d_y_in, double *d_y_out, void *stream);

/* ---- one process per GPU: the all-gather of the y segments as stores into the other ranks' memory ------------------
 * The row blocks of src/matrix/csr-matrix.cpp:77-95 on G GPUs driven by G processes (bench.py --gpus G,
 * python/spmv_amd/distributed.py).  Every rank allocates its copy of the whole y with spmv_hip_ipc_alloc, the ranks
 * exchange the 64-byte handles over whatever channel they share, and spmv_hip_ipc_open maps the other ranks' copies
 * into this process (peer access between the devices must be possible; on this driver HSA_ENABLE_IPC_MODE_LEGACY=0
 * must be in the environment).  Close the mappings (spmv_hip_ipc_close) before their owner frees the memory
 * (spmv_hip_ipc_free).  The memory comes back zeroed. */
int spmv_hip_ipc_alloc(void **d_ptr, size_t bytes, void *handle64);
int spmv_hip_ipc_open(const void *handle64, void **d_ptr);
int spmv_hip_ipc_close(void *d_ptr);
int spmv_hip_ipc_free(void *d_ptr);
/* d_dst[k][i] = d_src[i], i < n, for every k < ndst: one kernel that reads the segment once and writes it into up to 8
 * peers per launch (coalesced stores that leave over the xGMI link to each peer). */
int spmv_hip_peer_push(const double *d_src, double *const *d_dst, int ndst, int64_t n, void *stream);
/* spmv_hip_csr_spmv_out that ALSO delivers this rank's rows to the other ranks: peer_y[k] (host array of npeers DEVICE
 * pointers) is where this rank's first row lives in rank k's copy of y.  Where the plan runs the default kernel
 * (row-owned wave tiles, with or without a value dictionary, no split rows, no window kernels) every row sum is stored
 * into all copies by the multiply itself as each tile finishes (*fused = 1, up to 7 peers): the transfer overlaps the
 * same multiply and costs no launch of its own.  Otherwise the multiply is followed by spmv_hip_peer_push of
 * d_y_out on the same stream (*fused = 0).  Either way the peers' copies are complete once this stream has been
 * synchronised; a reader on another rank additionally needs to know that (a barrier between the processes).
 * fused may be NULL. */
int spmv_hip_csr_spmv_out_peers(const spmv_hip_plan *plan, const int32_t *d_row_ptr, const int32_t *d_column_index,
                                const double *d_value, const double *d_x, const double *d_y_in, double *d_y_out,
                                double *const *peer_y, int npeers, int *fused, void *stream);

/* y += A*x, COO in any order: wave-level segmented sums + fp64 atomics, i.e. the
 * semantics of coo_spmv_atomic (src/matrix/coo-matrix.cpp:287-309); equals
 * coo_spmv (:248-285) up to summation order. */
int spmv_hip_coo_spmv(int32_t rows, int32_t nnz, const int32_t *d_row_index,
                      const int32_t *d_column_index, const double *d_value,
                      const double *d_x, double *d_y, void *stream);

/* Stable sort of COO triplets by row index, in place on the device (entries of a row keep their
 * file order, so every row is still summed in file order).  Done automatically by
 * spmv_hip_upload_coo / _hybrid when the triplets are not row-sorted (unless
 * SPMV_HIP_FLAG_COO_KEEP_ORDER): in column-major file order -- what SuiteSparse ships -- every
 * entry would cost its own atomic.  Allocates temporaries, synchronises `stream`. */
int spmv_hip_coo_sort_by_row(int32_t rows, int32_t nnz, int32_t *d_row_index, int32_t *d_column_index,
                             double *d_value, void *stream);

/* Row-major (reference layout) -> column-major (k = l*rows + i) on the device. */
int spmv_hip_ell_to_column_major(int32_t rows, int32_t row_length,
                                 const int32_t *d_col_row_major, const double *d_val_row_major,
                                 int32_t *d_col_col_major, double *d_val_col_major,
                                 void *stream);

/* y += A*x, ELLPACK, column-major device layout; padded entries are multiplied
 * like real ones (0.0 * x[j]), as ell_spmv_inner_loop does
 * (src/matrix/ell-matrix.cpp:243-258).  Sums in the reference's order: bit-exact. */
int spmv_hip_ell_spmv(int32_t rows, int32_t row_length, const int32_t *d_col_col_major,
                      const double *d_val_col_major, const double *d_x, double *d_y,
                      void *stream);

/* STREAM triad a[i] = b[i] + q*c[i] on device arrays of n doubles (24 B and 2 flop
 * per element).  Replaces triad_kernel::run (src/kernels/triad.cpp:48-54, q = 3.1);
 * used as the EMPIRICAL HBM roofline next to the 8 TB/s spec peak.
 * mul then add, not fused: bit-exact with the reference loop. */
int spmv_hip_triad(int64_t n, double *d_a, const double *d_b, const double *d_c, double q,
                   void *stream);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_HIP_H */
