/*
 * spmv_hip_plan.h -- Level 2 of the C ABI of libspmv_hip.so: launch plans and multiplies on CALLER-OWNED device memory and
 * streams (bench.py and the tests pass torch tensors' data_ptr() and torch's stream), the entry points of the
 * one-process-per-GPU operators, and the stand-alone COO / ELLPACK / triad kernels.  The adapters of the reference bind
 * include/spmv_hip.h only (the context API builds the same plans by itself).  Same conventions as spmv_hip.h.
 */
#ifndef SPMV_HIP_PLAN_H
#define SPMV_HIP_PLAN_H

#include <stddef.h>

#include "spmv_hip_tuning.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_hip_plan spmv_hip_plan;

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
/* THE row partition, in one place: row_begin[0 .. parts] receives the block boundaries of `rows` rows over `parts` devices or
 * ranks -- the reference's static rule chunk = ceil(rows / parts), block g = [g * chunk, min(rows, (g + 1) * chunk))
 * (src/matrix/csr-matrix.cpp:77-95, src/matrix/csr-matrix-spmv.cpp:154-161), or, with balance_entries != 0 and a host row_ptr
 * (rows + 1 entries, any base), blocks of equal stored entries: boundary g = the first row whose row_ptr reaches g * nnz / parts
 * (SPMV_HIP_FLAG_BALANCE_ENTRIES).  What spmv_hip_create_multi's uploads cut by and what the one-process-per-GPU operators
 * (python/spmv_amd/partition.py) call: both process models share this one function.  No device needed. */
int spmv_hip_partition_rows(int32_t rows, int parts, const int32_t *host_row_ptr, int balance_entries, int32_t *row_begin);

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

#endif /* SPMV_HIP_PLAN_H */
