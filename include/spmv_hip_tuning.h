/*
 * spmv_hip_tuning.h -- the part of the C ABI of libspmv_hip.so that an adapter of the reference does NOT need
 * (include/spmv_hip.h is the drop-in boundary): which CSR algorithm an upload uses, switches that turn single tile
 * classes of the default kernel off -- they exist for A/B measurements (tools/ab.py, tools/bench_all.sh) and for the
 * parity tests of the fallback paths, not for production callers -- the stream a context enqueues on, and descriptive
 * numbers of an upload.  Same conventions as spmv_hip.h.
 */
#ifndef SPMV_HIP_TUNING_H
#define SPMV_HIP_TUNING_H

#include "spmv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* ---- plan / ctx flags (besides those of spmv_hip.h) ------------------------------------------------------------ */
#define SPMV_HIP_FLAG_XCD_REMAP 0x1u   /* give each XCD one contiguous run of tiles instead of the round-robin
                                           deal (measured SLOWER on MI355X for streaming SpMV: off by default) */
#define SPMV_HIP_FLAG_NO_INDEX_COMPRESSION 0x10u /* ctx: keep 32-bit column indices for every tile */
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
#define SPMV_HIP_FLAG_NO_VALUE_INDEX 0x100000u /* never build a value dictionary (spmv_hip_plan_csr_index_values is a no-op; the
                                              context does not build one for its uploads) */
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
/* 0x4000000u and 0x10000000u are not flags of this library: two kernel families that were measured SLOWER than the paths they
 * were meant to replace (hub columns for web graphs, 26.6 vs 23.9 us; a lane group per row for stencil rows of 17 ... 64
 * entries, 797 vs 740 us: DESIGN.md sections 3.3, 3.1b) were retired from the product library in round 5 and are
 * refused like any unknown bit.  They live on in libspmv_hip_experiments.so (csrc/internal.hpp) for tools/ and
 * tests/experiments/. */
/* Any bit that neither this header nor spmv_hip.h defines is refused with SPMV_HIP_ERR_INVALID by spmv_hip_create and spmv_hip_plan_csr. */

/* Enqueue everything this context does from now on on the caller's `stream` (a hipStream_t on the
 * context's device; NULL = the default stream), or, with use_own != 0, on the context's own stream
 * again.  Lets a host program order the multiply against its own work (and time it with its own
 * events: bench.py).  Waits for the stream in use before switching. */
int spmv_hip_set_stream(spmv_hip_ctx *ctx, void *stream, int use_own);

/* CSR algorithm for later uploads (default SPMV_HIP_CSR_AUTO);
 * lanes_per_row: 0 = choose, else 2,4,...,64 for SPMV_HIP_CSR_VECTOR. */
int spmv_hip_set_csr_algorithm(spmv_hip_ctx *ctx, int algorithm, int lanes_per_row);

/* Descriptive numbers for JSON output / tests.  out[] receives up to n of:
 * [0] format (0 none, 1 csr, 2 coo, 3 ell, 4 hybrid)  [1] rows  [2] cols  [3] stored entries
 * [4] csr algorithm in use  [5] lanes per row (vector)  [6] workgroups per launch
 * [7] row blocks / tiles  [8] long-row blocks  [9] device bytes held  [10] tiles with 16-bit columns
 * [11] shifted tiles  [12] tiles with an x window  [13] block-window tiles  [14] tiles of the
 * column-panel copy (see spmv_hip_plan_info)  [15] bytes one run streams with the tile classes in
 * use (see spmv_hip_plan_info [14]; formats without tiles: their algorithmic bytes)
 * [16] devices (1, or the num_gpus of spmv_hip_create_multi: [6..15] are then sums over the devices)
 * [17] ELLPACK path of the upload: 0 = not ELLPACK, 1 = the row-major arrays in place (wave tiles), 2 = column-major
 *      copy (one lane per row, the reference's order)
 * [18] ranks of the context's RCCL communicator as ncclCommCount reports them (0: the context holds none -- one device, or the
 *      gather is done by peer stores)  [19] 1 if back-to-back runs are pipelined (SPMV_HIP_FLAG_PIPELINE_GATHER asked for AND
 *      possible for the current upload) */
int spmv_hip_ctx_info(spmv_hip_ctx *ctx, int64_t *out, int n);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_HIP_TUNING_H */
