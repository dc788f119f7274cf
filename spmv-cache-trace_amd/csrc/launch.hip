// launch.hip -- the multiplies on caller-owned device arrays (Level 2 of include/spmv_hip.h): which kernel variant a
// plan launches, and the COO / ELLPACK / triad entry points.  Every kernel template is instantiated here.
#include "internal.hpp"
#ifdef SPMV_HIP_EXPERIMENTS
#include "csr_pipe.hpp"
#include <cstdlib>
#endif

#include <algorithm>

using namespace spmvi;

namespace {

template <int LPR>
void launch_vector(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                   const double * x, double * y, hipStream_t s)
{
    hipLaunchKernelGGL((spmv::csr_vector_kernel<LPR, kBlock>), dim3(pl->workgroups), dim3(kBlock), 0, s,
                       pl->rows, p, j, a, x, y);
}

} // namespace

extern "C" {

int spmv_hip_csr_spmv(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j,
                      const double * a, const double * x, double * y, void * stream)
{
    return spmv_hip_csr_spmv_out(pl, p, j, a, x, y, y, stream);
}

static int csr_spmv_launch(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                           const double * x, const double * y_in, double * y, void * stream, const spmv::PeerY * peers, int * fused);

int spmv_hip_csr_spmv_out(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                          const double * x, const double * y_in, double * y, void * stream)
{
    return csr_spmv_launch(pl, p, j, a, x, y_in, y, stream, nullptr, nullptr);
}

// defined in peer_gather.hip
int spmv_hip_peer_push(const double * d_src, double * const * d_dst, int ndst, int64_t n, void * stream);

int spmv_hip_csr_spmv_out_peers(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                                const double * x, const double * y_in, double * y, double * const * peer_y, int npeers,
                                int * fused_out, void * stream)
{
    if (npeers < 0 || (npeers > 0 && !peer_y))
        return fail(SPMV_HIP_ERR_INVALID, "bad peer list");
    for (int k = 0; k < npeers; ++k)
        if (!peer_y[k])
            return fail(SPMV_HIP_ERR_INVALID, "null peer pointer");
    if (fused_out)
        *fused_out = 0;
    if (npeers == 0)
        return csr_spmv_launch(pl, p, j, a, x, y_in, y, stream, nullptr, nullptr);
    int fused = 0;
    spmv::PeerY py{};
    if (npeers <= spmv::kMaxPeers) {
        for (int k = 0; k < npeers; ++k)
            py.y[k] = peer_y[k];
        py.n = npeers;
    }
    int rc = csr_spmv_launch(pl, p, j, a, x, y_in, y, stream, npeers <= spmv::kMaxPeers ? &py : nullptr, &fused);
    if (rc != SPMV_HIP_OK)
        return rc;
    if (fused_out)
        *fused_out = fused;
    if (!fused && pl && pl->rows > 0) // this plan's kernel has no forwarding variant: one more launch pushes the segment
        return spmv_hip_peer_push(y, peer_y, npeers, pl->rows, stream);
    return SPMV_HIP_OK;
}

static int csr_spmv_launch(const spmv_hip_plan * pl, const int32_t * p, const int32_t * j, const double * a,
                           const double * x, const double * y_in, double * y, void * stream, const spmv::PeerY * peers, int * fused)
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
            const bool vi = c16 && x32 && !xcd && pl->nvalues > 0 && pl->values_from == a;
#ifdef SPMV_HIP_EXPERIMENTS
            if (c16 && x32 && !xcd && pl->nhubs > 0) {
                // hub columns: the dense copy of their x entries first (every multiply: x is the caller's), then the tiles
                hipLaunchKernelGGL(spmv::hub_gather_kernel, dim3((unsigned) ((pl->nhubs + 255) / 256)), dim3(256), 0, s, pl->nhubs, pl->d_hub_column, x,
                                   pl->d_hubx);
                if (vi)
                    hipLaunchKernelGGL((spmv::csr_segtile_kernel<true, true, false, true, true>), grid, dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j,
                                       pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, pl->d_vidx, pl->d_vtab, pl->nvalues, pl->d_colh, pl->d_hubx);
                else
                    hipLaunchKernelGGL((spmv::csr_segtile_kernel<true, true, false, false, true>), grid, dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j,
                                       pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, (const uint8_t *) nullptr, (const double *) nullptr, 0, pl->d_colh,
                                       pl->d_hubx);
            } else
#endif
            if (vi)
                // a value dictionary (pattern / graph matrices): one index byte per entry instead of eight bytes of value
                hipLaunchKernelGGL((spmv::csr_segtile_kernel<true, true, false, true>), grid, dim3(256), 0, s, pl->ntiles, pl->d_tiles, p, j, pl->d_col16,
                                   a, x, y_in, y, pl->nnz, pl->cols, pl->d_vidx, pl->d_vtab, pl->nvalues);
            else if (xcd) SPMV_SEG_C(true); else SPMV_SEG_C(false);
#undef SPMV_SEG_C
#undef SPMV_SEG_X
#undef SPMV_SEG_LAUNCH
        } else if (pl->ntiles > 0) {
            const int xcd = (pl->flags & SPMV_HIP_FLAG_XCD_REMAP) ? 1 : 0;
            const int exact_order = (pl->flags & SPMV_HIP_FLAG_EXACT_ORDER) ? 1 : 0;
            // what the kernels get: bit 0 = exact order, bit 1 = y written non-temporally -- when the matrix streams from HBM (twice
            // the Infinity Cache: the same threshold as for the 128-row tiles); a cache-resident matrix keeps its y in the caches
            // (bits 8-11: the rows per group of the plan's group tiles, csr_blocktile.hpp)
            const int exact = exact_order | ((12.0 * (double) pl->nnz + 20.0 * (double) pl->rows >= 512e6) ? 2 : 0)
                | ((pl->colshare_tiles > 0 ? pl->block_hint : 0) << 8);
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
#ifdef SPMV_HIP_EXPERIMENTS
            if (const char * pipe = std::getenv("SPMV_HIP_PIPE")) { // timing experiment (csr_pipe.hpp): persistent, software-pipelined narrow tiles
                const int per_cu = std::max(1, std::atoi(pipe));
                if (c16 && pl->tile == 512) {
                    hipLaunchKernelGGL((spmv::csr_pipe_kernel<512>), dim3((unsigned) (cu_count() * per_cu)), dim3(256), 0, s, pl->ntiles, pl->d_tiles, p,
                                       pl->d_col16, a, x, y_in, y, pl->cols);
                    HIP_TRY(hipGetLastError());
                    return SPMV_HIP_OK;
                }
            }
#endif
            // the dictionary launch has its own descriptors when runs of constant-row tiles were re-cut (plan_csr.hip)
            const int4 * vi_tiles = pl->d_tiles_vi ? pl->d_tiles_vi : pl->d_tiles;
            const int vi_ntiles = pl->d_tiles_vi ? pl->ntiles_vi : pl->ntiles;
            const unsigned vi_groups = (unsigned) ((vi_ntiles + 3) / 4);
            const bool all_blockwin = c16 && (pl->d_blocks || pl->d_segblocks) && pl->blockwin_tiles == pl->ntiles;
            // segment-window plans of a one-process-per-GPU operator: the window kernel and the launch over the leftover
            // tiles both forward their row sums (x below 4 GiB, no split rows: their partial sums meet in atomics)
            const bool rest_forwards = peers && c16 && x32 && pl->d_segblocks && pl->split_rows == 0 && pl->tile == 512 && !xcd
                && (pl->blockwin_tiles == pl->ntiles || (pl->d_rest_tiles && pl->nrest_tiles > 0));
            if (all_blockwin) {
            } else if (rest_forwards) {
                hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, false, true, true>), dim3((unsigned) ((pl->nrest_tiles + 3) / 4)),
                                   dim3(256), 0, s, pl->nrest_tiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                   spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, *peers, pl->d_rest_tiles);
            } else if (c16 && (pl->d_blocks || pl->d_segblocks) && pl->d_rest_tiles && pl->nrest_tiles > 0 && pl->tile == 512 && !xcd) {
                // the few tiles a window kernel did not take: a launch over exactly those
                const dim3 grid((unsigned) ((pl->nrest_tiles + 3) / 4));
                if (x32)
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, false, false, true>), grid, dim3(256), 0, s,
                                       pl->nrest_tiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, spmv::PeerY{}, pl->d_rest_tiles);
                else
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, false, false, 0, 0, false, false, false, true>), grid, dim3(256), 0, s,
                                       pl->nrest_tiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, spmv::PeerY{}, pl->d_rest_tiles);
            } else
#ifdef SPMV_HIP_EXPERIMENTS
            if (pl->d_group_tiles && c16 && x32 && !xcd && !exact_order && pl->tile == 512 && (!peers || pl->split_rows == 0)) {
                // most tiles are rows of 17 ... 64 entries of a stencil or band: a group of lanes per row (csr_rowgroup.hpp),
                // then the other tiles (the x-window variant over a list)
                const dim3 grid((unsigned) ((pl->ngroup_tiles + 3) / 4)), rest((unsigned) ((pl->ngroup_rest + 3) / 4));
                if (peers) { // one process per GPU: both launches forward their row sums
                    hipLaunchKernelGGL((spmv::csr_rowgroup_kernel<true>), grid, dim3(256), 0, s, pl->ngroup_tiles, pl->d_group_tiles, pl->d_tiles, a, x, y_in,
                                       y, pl->cols, pl->d_patterns, *peers);
                    if (pl->ngroup_rest > 0)
                        hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 256, false, false, true, true>), rest, dim3(256), 0, s,
                                           pl->ngroup_rest, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                           spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, *peers, pl->d_group_rest);
                    *fused = 1;
                } else {
                    hipLaunchKernelGGL((spmv::csr_rowgroup_kernel<false>), grid, dim3(256), 0, s, pl->ngroup_tiles, pl->d_group_tiles, pl->d_tiles, a, x, y_in,
                                       y, pl->cols, pl->d_patterns);
                    if (pl->ngroup_rest > 0)
                        hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 256, false, false, false, true>), rest, dim3(256), 0, s,
                                           pl->ngroup_rest, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                           spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, spmv::PeerY{}, pl->d_group_rest);
                }
            } else
#endif
            // x staged through LDS when most tiles have a window.  With one lane per row (EXACT_ORDER,
            // the in-place ELLPACK path) long row sums want the occupancy more than the gather wants
            // the window (L = 81: 339 vs 333 us; L = 27: 199 vs 223 us), so only up to 32 entries per row
            if (!(pl->flags & SPMV_HIP_FLAG_NO_X_WINDOW) && (!exact_order || pl->longest_tile_row <= 32) && c16 && x32
                && pl->tile == 512 && !xcd && 2 * (long long) pl->xwin_tiles > pl->ntiles
                && !(pl->nvalues > 0 && pl->values_from == a) /* a dictionary kept in spite of the windows: constant-row tiles (plan_csr.hip) */) {
                if (peers && pl->split_rows == 0 && !pl->d_blocks && !pl->d_segblocks) { // one process per GPU: row sums forwarded (see below)
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 256, false, false, true>), dim3(pl->workgroups), dim3(256), 0, s,
                                       pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns, spmv::PanelInfo{},
                                       (const uint8_t *) nullptr, (const double *) nullptr, 0, *peers);
                    *fused = 1;
                } else
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
            } else if (peers && c16 && x32 && !xcd && pl->split_rows == 0 && !pl->d_blocks && !pl->d_segblocks) {
                // one process per GPU: the default kernel (with or without the value dictionary) stores every row sum
                // into the other ranks' copies of y as well (split long rows add partial sums atomically and keep the
                // separate push; so do the plans whose tiles are shared with a window kernel)
                if (pl->nvalues > 0 && pl->values_from == a)
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, true, true>), dim3(vi_groups), dim3(256), 0, s,
                                       vi_ntiles, vi_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, pl->d_vidx, pl->d_vtab, pl->nvalues, *peers);
                else
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, false, true>), dim3(pl->workgroups), dim3(256), 0, s,
                                       pl->ntiles, pl->d_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, (const uint8_t *) nullptr, (const double *) nullptr, 0, *peers);
                *fused = 1;
            } else if (c16 && x32 && pl->nvalues > 0 && pl->values_from == a) {
                // the default kernel with the value dictionary: one byte per entry instead of eight
                if (xcd)
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, true, 0, 0, false, true>), dim3(vi_groups), dim3(256), 0, s,
                                       vi_ntiles, vi_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, pl->d_vidx, pl->d_vtab, pl->nvalues);
                else
                    hipLaunchKernelGGL((spmv::csr_wavetile_kernel<512, true, true, false, 0, 0, false, true>), dim3(vi_groups), dim3(256), 0, s,
                                       vi_ntiles, vi_tiles, p, j, pl->d_col16, a, x, y_in, y, pl->nnz, pl->cols, exact, pl->d_patterns,
                                       spmv::PanelInfo{}, pl->d_vidx, pl->d_vtab, pl->nvalues);
            } else {
                if (xcd) SPMV_WT_C(512, true); else SPMV_WT_C(512, false);
            }
#undef SPMV_WT_C
#undef SPMV_WT_X
#undef SPMV_WT_LAUNCH
            // the tiles marked for a block window were skipped above (only when the 16-bit column
            // stream is valid for this column array, like the marks themselves)
            if (c16 && pl->d_segblocks) {
                // segment windows: one workgroup of 8 waves per block of tiles, the smallest window variant the plan's blocks fit
                // (2688 doubles + 8 product slices = 53.25 KB: three workgroups per CU; 4096: two)
                if (rest_forwards) { // one process per GPU: both launches of this multiply forward their row sums
                    if (pl->segwin_slots <= 2688)
                        hipLaunchKernelGGL((spmv::csr_segwin_kernel<512, 2688, true>), dim3(pl->nsegblocks), dim3(512), 0, s, pl->d_tiles,
                                           pl->d_segblocks, p, pl->d_col16, a, x, y_in, y, *peers);
                    else
                        hipLaunchKernelGGL((spmv::csr_segwin_kernel<512, 4096, true>), dim3(pl->nsegblocks), dim3(512), 0, s, pl->d_tiles,
                                           pl->d_segblocks, p, pl->d_col16, a, x, y_in, y, *peers);
                    *fused = 1;
                } else if (pl->segwin_slots <= 2688)
                    hipLaunchKernelGGL((spmv::csr_segwin_kernel<512, 2688>), dim3(pl->nsegblocks), dim3(512), 0, s, pl->d_tiles,
                                       pl->d_segblocks, p, pl->d_col16, a, x, y_in, y);
                else
                    hipLaunchKernelGGL((spmv::csr_segwin_kernel<512, 4096>), dim3(pl->nsegblocks), dim3(512), 0, s, pl->d_tiles,
                                       pl->d_segblocks, p, pl->d_col16, a, x, y_in, y);
            } else if (c16 && pl->d_blocks) {
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
