// context.hip -- Level 1 of include/spmv_hip.h: a context that owns device copies of A, x and y (what the
// hip_{csr,coo,ell,hybrid}_spmv_kernel adapters of host/kernels/spmv-kernels.cpp bind).
#include "internal.hpp"

#include <algorithm>
#include <new>
#include <system_error>
#include <thread>

using namespace spmvi;

namespace spmvi {

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

} // namespace spmvi

namespace {

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

// ---- host arrays -> HBM beside the host tiler (round 5) ---------------------------------------------------------------------------
// A copy from pageable host memory occupies the thread that issues it until the data has left (hipMemcpyAsync from pageable
// memory is synchronous for the host), and Kernel::init used to do everything in a row: cut the tiles from row_ptr (11 ms for
// Poisson 4096^2), THEN copy the arrays (1.07 GB: 20 ms at the 49-56 GB/s a pageable copy reaches on these boxes).  The copies
// now run on a helper thread while the caller's thread cuts the tiles (src/kernels/csr-spmv.cpp:26-62 is what this stands in
// for: load, convert, first touch).  Measured and rejected: page-locked staging (four threads copying 2 MiB chunks into pinned
// buffers, each with its own stream) reaches 35-43 GB/s -- SLOWER than the plain pageable copy; what made the first pageable
// copy of a process look slow in the probe (14 GB/s) was the runtime's own start-up, not the pages
// (profiles/r05_h2d_probe.log, profiles/r05_host_boundary.md).  If the helper thread cannot be started the caller copies itself.
struct UploadJob {
    void * dst;
    const void * src;
    size_t bytes;
};

class StagedUpload {
  public:
    StagedUpload(int device, hipStream_t stream) : device_(device), stream_(stream) {}
    ~StagedUpload() { (void) finish(); }
    void start(std::vector<UploadJob> jobs)
    {
        jobs_ = std::move(jobs);
        size_t total = 0;
        for (auto const & j : jobs_)
            total += j.bytes;
        if (total >= (8u << 20)) {
            try {
                worker_ = std::thread([this] { copy_all(); });
                return;
            } catch (std::system_error const &) {
            } catch (std::bad_alloc const &) {
            }
        }
        copy_all();
    }
    hipError_t finish()
    {
        if (worker_.joinable())
            worker_.join();
        return error_;
    }

  private:
    void copy_all()
    {
        hipError_t e = hipSetDevice(device_);
        for (auto const & j : jobs_)
            if (j.bytes > 0 && e == hipSuccess)
                e = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream_);
        error_ = e;
    }
    int device_;
    hipStream_t stream_;
    std::vector<UploadJob> jobs_;
    std::thread worker_;
    hipError_t error_ = hipSuccess;
};

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
        for (size_t g = 0; g < c->comm.size(); ++g) { // SPMV_HIP_FLAG_PIPELINE_GATHER: the gather streams and their events
            if (g < c->parts.size())
                (void) hipSetDevice(c->parts[g]->device);
            for (hipEvent_t ev : {g < c->ev_mul.size() ? c->ev_mul[g] : nullptr, g < c->ev_sent[0].size() ? c->ev_sent[0][g] : nullptr,
                                  g < c->ev_sent[1].size() ? c->ev_sent[1][g] : nullptr})
                if (ev)
                    (void) hipEventDestroy(ev);
            if (c->comm[g])
                (void) hipStreamDestroy(c->comm[g]);
        }
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
    c->rows = rows;
    c->cols = cols;
    c->nnz = nnz;
    int rc;
    if ((rc = dev_alloc(c, &c->d_ptr, (size_t) rows + 1)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_col, (size_t) nnz)) != 0) return rc;
    if ((rc = dev_alloc(c, &c->d_val, (size_t) nnz)) != 0) return rc;
    {
        // the arrays travel on a helper thread while this thread cuts the tiles from row_ptr
        StagedUpload up(c->device, c->stream);
        up.start({{c->d_ptr, row_ptr, ((size_t) rows + 1) * sizeof(int32_t)},
                  {c->d_col, column_index, (size_t) nnz * sizeof(int32_t)},
                  {c->d_val, value, (size_t) nnz * sizeof(double)}});
        rc = spmv_hip_plan_csr(&c->plan, rows, cols, row_ptr, c->csr_algorithm, c->csr_lanes, c->flags);
        const hipError_t e = up.finish();
        if (rc != 0 || e != hipSuccess) {
            const std::string why = rc != 0 ? last_error_text() : std::string();
            free_ctx_matrix(c);
            if (rc != 0) {
                set_last_error_text(why);
                return rc;
            }
            return fail_hip(e, "upload (host arrays -> device)");
        }
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
        // a candidate for (masked) block tiles is confirmed -- and its tiles cut on the row groups -- before they are classified
        if ((rc = spmv_hip_plan_csr_confirm_blocks(c->plan, c->d_ptr, c->d_col, row_ptr, c->stream)) != 0) return rc;
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
        if ((rc = spmv_hip_plan_csr_confirm_blocks(c->plan, c->d_ptr, c->d_col, host_ptr.data(), c->stream)) != 0) return rc;
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
        StagedUpload up(c->device, c->stream);
        up.start({{c->d_idx, row_index, (size_t) nnz * sizeof(int32_t)},
                  {c->d_col, column_index, (size_t) nnz * sizeof(int32_t)},
                  {c->d_val, value, (size_t) nnz * sizeof(double)}});
        HIP_TRY(up.finish());
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
    // columns and shifted tiles: they run through the wave-tile kernel in place, with no transposed copy
    // (measured against the column-major kernel: L=5 202 vs 265 us, L=27 229 vs 285, L=81 337 vs 368).
    // Rows of up to 16 entries are summed by one lane each: the reference's order, bit for bit
    // (src/matrix/ell-matrix.cpp:243-258).  Longer rows get the lanes the CSR path gives a row of that length
    // (2 .. 64, <= 16 entries per lane): with one lane per row a tile of 93-entry rows keeps 5 of the wave's 64
    // lanes busy through 93 dependent additions (queen-like L = 93: 0.62 of the roofline against 0.71-0.80 for the
    // same matrix in CSR; bands of 65-301 per row 0.63-0.69, profiles/r02_ell_row_lengths.log); the sums then
    // differ from the reference's in the last bits (1e-10 class, like CSR rows of that length).
    // SPMV_HIP_FLAG_EXACT_ORDER keeps one lane per row for every length: in place up to kEllInPlaceMaxLength
    // entries per row, through the column-major kernel beyond (measured crossover, same log).
    // One exception, measured (profiles/r03_ell_row_lengths.log, fraction of the roofline in place / column-major):
    // L = 129 0.88 / 0.72, but L = 201 0.58 / 0.69, 361 0.59 / 0.68, 441 0.66 / 0.70 -- one or two whole rows per
    // 512-entry tile leave load slots idle and the x window no longer applies -- and L = 601 (a wave strides the row)
    // 0.69 / 0.69.  So rows of 161..512 entries take the column-major kernel.
    const bool exact = (c->flags & SPMV_HIP_FLAG_EXACT_ORDER) != 0;
    // (round 4: multi-window tiles -- a wave takes up to 8 such rows and walks them in windows of 512 entries -- fill the tiles of
    // these lengths again; SPMV_HIP_FLAG_NO_MULTI_WINDOW brings the column-major kernel back for them)
    // Rows of more than 2048 entries (round 5): a wave per row walks them in place, in registers (long_row_sum, tile_common.hpp),
    // with 16-bit columns where the row's columns allow; the column-major copy (0.60-0.69 of the roofline,
    // profiles/r04_ell_long_rows.md) is left to SPMV_HIP_FLAG_ELL_COLUMN_MAJOR and to EXACT_ORDER (it keeps the reference's order).
    const bool poor_fill = row_length > 160 && row_length <= 2048 && (c->flags & SPMV_HIP_FLAG_NO_MULTI_WINDOW);
    c->ell_as_tiles = n > 0 && !(c->flags & SPMV_HIP_FLAG_ELL_COLUMN_MAJOR)
        && (exact ? row_length <= kEllInPlaceMaxLength : !poor_fill || c->ell_in_place_any_length);
    if (c->ell_in_place_any_length)
        c->ell_as_tiles = n > 0 && !(c->flags & SPMV_HIP_FLAG_ELL_COLUMN_MAJOR);
    if (c->ell_as_tiles) {
        std::vector<int32_t> row_ptr((size_t) rows + 1);
        for (int32_t i = 0; i <= rows; ++i)
            row_ptr[(size_t) i] = i * row_length;
        if ((rc = spmv_hip_plan_csr(&c->plan, rows, cols, row_ptr.data(), SPMV_HIP_CSR_WAVETILE, 0,
                                    c->flags | SPMV_HIP_FLAG_NO_BLOCK_TILES /* padded rows are no 3 x 3 blocks, and this upload has no repack step */
                                        | (row_length <= 16 || c->ell_in_place_any_length ? SPMV_HIP_FLAG_EXACT_ORDER : 0u))) != 0)
            return rc;
        if ((rc = dev_alloc(c, &c->d_ptr, (size_t) rows + 1)) != 0) return rc;
        {
            StagedUpload up(c->device, c->stream);
            up.start({{c->d_ptr, row_ptr.data(), ((size_t) rows + 1) * sizeof(int32_t)},
                      {c->d_col, column_index, (size_t) n * sizeof(int32_t)},
                      {c->d_val, value, (size_t) n * sizeof(double)}});
            HIP_TRY(up.finish());
        }
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
        return multi_upload_hybrid(c, rows, cols, ell_row_length, ell_column_index, ell_value, num_coo_entries, coo_row_index, coo_column_index,
                                   coo_value);
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
        std::string const keep = last_error_text();
        free_ctx_matrix(c);
        set_last_error_text(keep);
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
    bool delivered = false;
    // a part of a multi-GPU context with SPMV_HIP_FLAG_FUSED_PEER_STORE: the CSR multiply delivers its rows to the other
    // devices' copies of y itself (row sums forwarded by the kernel, or pushed behind it: spmv_hip_csr_spmv_out_peers)
    auto csr_run = [&]() {
        if (c->y_in_override) // a part of a pipelined multi-GPU context: the old y comes from the other copy (multi_gpu.hip)
            return spmv_hip_csr_spmv_out(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->y_in_override, c->d_y, c->stream);
        if (c->peer_y.empty())
            return spmv_hip_csr_spmv(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->stream);
        delivered = true;
        return spmv_hip_csr_spmv_out_peers(c->plan, c->d_ptr, c->d_col, c->d_val, c->d_x, c->d_y, c->d_y, c->peer_y.data(),
                                           (int) c->peer_y.size(), nullptr, c->stream);
    };
    switch (c->format) {
    case 1: rc = csr_run(); break;
    case 2:
        rc = c->as_csr ? csr_run() : ctx_coo_run(c, c->nnz, c->d_idx, c->d_col, c->d_val);
        break;
    case 3:
        rc = c->ell_as_tiles ? csr_run() : spmv_hip_ell_spmv(c->rows, c->row_length, c->d_col, c->d_val, c->d_x, c->d_y, c->stream);
        break;
    case 4:
        if (c->as_csr) { // ELL part and remainder merged into one row-major matrix: one launch
            rc = csr_run();
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
    if (!c->peer_y.empty() && c->rows > 0 && !delivered) // kernels without a forwarding variant: the push kernel behind them
        if ((rc = spmv_hip_peer_push(c->d_y, c->peer_y.data(), (int) c->peer_y.size(), c->rows, c->stream)) != 0)
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
        int64_t v[20] = {c->format, c->rows, c->cols, c->nnz, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (int64_t) c->parts.size(), 0, 0, c->pipeline ? 1 : 0};
        if (!c->comms.empty() && c->comms[0] && c->p_comm_count) { // what RCCL itself says the communicator holds
            int ranks = 0;
            if (c->p_comm_count(c->comms[0], &ranks) == ncclSuccess)
                v[18] = ranks;
        }
        for (size_t g = 0; g < c->parts.size(); ++g) {
            int64_t w[18] = {0};
            if (c->parts[g]->format != 0)
                spmv_hip_ctx_info(c->parts[g], w, 18);
            v[17] = w[17] ? w[17] : v[17];
            v[4] = w[4] ? w[4] : v[4];
            v[5] = w[5];
            for (int i : {6, 7, 8, 9, 10, 11, 12, 13, 14, 15})
                v[i] += w[i];
            if (c->yfull[g])
                v[9] += (int64_t) c->chunk * (int64_t) c->parts.size() * 8 * (c->pipeline ? 2 : 1);
        }
        for (int i = 0; i < n && i < 20; ++i)
            out[i] = v[i];
        return SPMV_HIP_OK;
    }
    int64_t v[20] = {c->format, c->rows, c->cols, c->nnz, 0, 0, 0, 0, 0, (int64_t) c->bytes, 0, 0, 0, 0, 0, 0, 1,
                     c->format == 3 ? (c->ell_as_tiles ? 1 : 2) : 0, 0, 0};
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
    for (int i = 0; i < n && i < 20; ++i)
        out[i] = v[i];
    return SPMV_HIP_OK;
}

} // extern "C"
