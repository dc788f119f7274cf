/*
 * spmv_hip.h -- C ABI of the MI355X (gfx950) SpMV engine: THE DROP-IN BOUNDARY.
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
 *
 * This header is what an adapter of the reference binds (integration/src/kernels/hip-spmv.cpp compiles against it alone):
 * a context that owns device copies of A, x and y -- create, upload_{csr,coo,ell,hybrid}, set_x / set_y / get_y, run, sync,
 * flush_caches, last_run_ns / last_run_times, strerror, destroy -- the error codes, and the few flags an adapter has a use for.
 * Everything else lives in two headers that include this one:
 *   spmv_hip_tuning.h  CSR algorithm selection, the switches that turn single tile classes off (for A/B measurements and tests),
 *                      spmv_hip_set_stream, spmv_hip_ctx_info
 *   spmv_hip_plan.h    Level 2: launch plans and multiplies on caller-owned device memory (bench.py, the tests), the
 *                      one-process-per-GPU peer entry points, the stand-alone COO / ELLPACK / triad kernels
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

/* ---- context flags an adapter may pass to spmv_hip_create / spmv_hip_create_multi --------------------------------
 * (the tile-class and tuning switches are in spmv_hip_tuning.h; any bit that neither header defines is refused with
 * SPMV_HIP_ERR_INVALID) */
#define SPMV_HIP_FLAG_EXACT_ORDER 0x2u  /* force one lane per row everywhere (bit-exact, slower on long rows); for ELLPACK
                                           uploads: rows of more than 16 entries too (default: 2..64 lanes per such row,
                                           1e-10 class; rows of <= 16 entries are bit-exact either way) */
#define SPMV_HIP_FLAG_COO_KEEP_ORDER 0x20u /* ctx: keep COO triplets in file order on the device */
#define SPMV_HIP_FLAG_NO_RUN_EVENTS 0x80000u /* ctx: spmv_hip_run does not bracket the launch with a HIP event pair (each
                                             record is a barrier packet between back-to-back runs, ~5 us per run);
                                             spmv_hip_last_run_ns then returns SPMV_HIP_ERR_STATE.  For callers that time a
                                             whole region themselves (bench.py); the Kernel adapters keep the events */
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
#define SPMV_HIP_FLAG_PIPELINE_GATHER 0x4u /* spmv_hip_create_multi, with the RCCL all-gather or SPMV_HIP_FLAG_PEER_GATHER: back-to-back runs
                                             overlap -- the gather of run k travels on a second stream per device while run k + 1
                                             multiplies (every device keeps TWO copies of y that alternate: a run reads its slot of one
                                             and writes the other, so a slot is never written while it is still being sent).  A run
                                             followed by spmv_hip_sync (the reference's timed loop, src/profile-kernel.cpp:159-161)
                                             costs what it costs without the flag; K runs and one sync cost ~K x max(multiply, gather)
                                             instead of K x (multiply + gather).  CSR-planned uploads only (CSR, and COO / ELLPACK /
                                             hybrid uploads that run as row-major tiles); ignored -- the serial order is kept -- for
                                             one device, for SPMV_HIP_FLAG_FUSED_PEER_STORE (whose multiply IS the transfer) and for
                                             uploads that run another kernel.  Same y, bit for bit. */

typedef struct spmv_hip_ctx spmv_hip_ctx;

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

#ifdef __cplusplus
}
#endif

#endif /* SPMV_HIP_H */
