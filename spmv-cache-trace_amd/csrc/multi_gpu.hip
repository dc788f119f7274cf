// multi_gpu.hip -- one process, G devices (spmv_hip_create_multi): row blocks by the reference's static rule
// (src/matrix/csr-matrix.cpp:77-95), x replicated, one in-place all-gather of the y slots per run.
#include "internal.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace spmvi;

/* ---- multi-GPU front ------------------------------------------------------------------------------
 * One process, G devices: rows are cut by the reference's static rule,
 * chunk = ceil(rows / G) (src/matrix/csr-matrix.cpp:77-95, devices take the place of threads), x is
 * replicated, and a run is G local multiplies followed by ONE in-place ncclAllGather of the y slots
 * inside a group call.  librccl.so is loaded with dlopen only when G > 1 (or when
 * SPMV_HIP_FORCE_RCCL=1 asks for the collective with one device): a single-GPU build has no
 * dependency on it, and a process that already holds another RCCL (PyTorch's) is not handed a second
 * one behind its back. */
namespace spmvi {

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
    c->p_comm_count = reinterpret_cast<decltype(c->p_comm_count)>(dlsym(c->rccl_lib, "ncclCommCount")); // (optional: ctx_info [18])
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

// the copy of y that holds the current vector on device g (SPMV_HIP_FLAG_PIPELINE_GATHER alternates two)
static inline double * ybuf(const spmv_hip_ctx * c, int which, size_t g) { return which ? c->yfull2[g] : c->yfull[g]; }

// does a run of this part go through its CSR plan (the only launches with a y_in != y_out form)?
static bool part_runs_csr_plan(const spmv_hip_ctx * part)
{
    return part->plan && (part->format == 1 || (part->format == 2 && part->as_csr) || (part->format == 3 && part->ell_as_tiles)
                          || (part->format == 4 && part->as_csr));
}

void multi_free_matrix(spmv_hip_ctx * c)
{
    for (size_t g = 0; g < c->parts.size(); ++g) {
        (void) hipSetDevice(c->parts[g]->device);
        (void) hipStreamSynchronize(c->parts[g]->stream);
        if (g < c->comm.size() && c->comm[g])
            (void) hipStreamSynchronize(c->comm[g]);
    }
    for (size_t g = 0; g < c->parts.size(); ++g) {
        (void) hipSetDevice(c->parts[g]->device);
        free_ctx_matrix(c->parts[g]);
        c->parts[g]->borrowed_y = nullptr;
        c->parts[g]->y_in_override = nullptr;
        c->parts[g]->peer_y.clear();
        if (g < c->yfull.size() && c->yfull[g]) {
            (void) hipFree(c->yfull[g]);
            c->yfull[g] = nullptr;
        }
        if (g < c->yfull2.size() && c->yfull2[g]) {
            (void) hipFree(c->yfull2[g]);
            c->yfull2[g] = nullptr;
        }
    }
    c->pipeline = false;
    c->cur = 0;
    c->sent_recorded[0] = c->sent_recorded[1] = false;
    c->format = 0;
    c->rows = c->cols = c->nnz = 0;
    c->chunk = 0;
    c->row_begin.clear();
    c->packed = true;
    c->timed = false;
}

// THE row partition (spmv_hip_partition_rows; both process models cut by it): the reference's static rule, or blocks of
// equal stored entries when the entries in front of every row are given.
template <class T>
static void partition_rows(int32_t rows, int G, const T * entries_before_row /* rows + 1, or null */, int32_t * row_begin /* G + 1 */)
{
    row_begin[0] = 0;
    if (entries_before_row) {
        const long long nnz = (long long) entries_before_row[rows] - (long long) entries_before_row[0];
        for (int g = 1; g < G; ++g) {
            const T target = (T) ((long long) entries_before_row[0] + (nnz * g) / G);
            const int32_t r = (int32_t) (std::lower_bound(entries_before_row, entries_before_row + rows + 1, target) - entries_before_row);
            row_begin[g] = std::max(row_begin[g - 1], std::min(r, rows));
        }
    } else {
        const long long per = std::max<long long>(1, ((long long) rows + G - 1) / G); // ceil(rows / G): src/matrix/csr-matrix.cpp:77-95
        for (int g = 1; g < G; ++g)
            row_begin[g] = (int32_t) std::min<long long>(rows, g * per);
    }
    row_begin[G] = rows;
}

// Row blocks of a multi-GPU context and each device's copy of y.  row_ptr (rows + 1 entries, any base) gives the
// stored entries in front of every row: the reference's static rule needs only `rows`, SPMV_HIP_FLAG_BALANCE_ENTRIES
// cuts where the entries divide evenly (boundary g = the first row whose row_ptr reaches g * nnz / G).
int multi_layout(spmv_hip_ctx * c, int32_t rows, const long long * entries_before_row /* rows + 1, or null */)
{
    multi_free_matrix(c);
    const int G = (int) c->parts.size();
    c->row_begin.assign((size_t) G + 1, 0);
    partition_rows(rows, G, (c->flags & SPMV_HIP_FLAG_BALANCE_ENTRIES) ? entries_before_row : nullptr, c->row_begin.data());
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
    for (int g = 0; g < G; ++g) { // fused peer store: where part g's rows live in the other devices' copies
        c->parts[(size_t) g]->peer_y.clear();
        if (c->flags & SPMV_HIP_FLAG_FUSED_PEER_STORE)
            for (int h = 0; h < G; ++h)
                if (h != g)
                    c->parts[(size_t) g]->peer_y.push_back(c->yfull[(size_t) h] + (size_t) g * (size_t) chunk);
    }
    return SPMV_HIP_OK;
}

// After every upload: can back-to-back runs overlap (SPMV_HIP_FLAG_PIPELINE_GATHER)?  Needs more than one device, a gather that
// is a step of its own (not the fused store), and parts whose runs have a y_in != y_out form.  Then every device gets its
// second copy of y (zeroed like the first).
static int multi_after_upload(spmv_hip_ctx * c)
{
    const size_t G = c->parts.size();
    c->pipeline = false;
    c->cur = 0;
    c->sent_recorded[0] = c->sent_recorded[1] = false;
    // (one device: only in the rehearsal that forces the collective, SPMV_HIP_FORCE_RCCL=1 -- the comm streams exist only then)
    if (!(c->flags & SPMV_HIP_FLAG_PIPELINE_GATHER) || (c->flags & SPMV_HIP_FLAG_FUSED_PEER_STORE) || c->comm.size() != G || G < 1
        || (G < 2 && c->comms.empty()))
        return SPMV_HIP_OK;
    for (size_t g = 0; g < G; ++g)
        if (c->parts[g]->rows > 0 && !part_runs_csr_plan(c->parts[g]))
            return SPMV_HIP_OK;
    c->yfull2.assign(G, nullptr);
    const size_t ybytes = (size_t) c->chunk * G * sizeof(double) + 64;
    for (size_t g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(c->parts[g]->device));
        HIP_TRY(hipMalloc((void **) &c->yfull2[g], ybytes));
        HIP_TRY(hipMemsetAsync(c->yfull2[g], 0, ybytes, c->parts[g]->stream));
        HIP_TRY(hipStreamSynchronize(c->parts[g]->stream));
    }
    c->pipeline = true;
    return SPMV_HIP_OK;
}

int multi_upload_failed(spmv_hip_ctx * c, int rc)
{
    std::string const keep = last_error_text();
    multi_free_matrix(c);
    set_last_error_text(keep);
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
    if ((rc = multi_after_upload(c)) != 0)
        return multi_upload_failed(c, rc);
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
    if ((rc = multi_after_upload(c)) != 0)
        return multi_upload_failed(c, rc);
    return SPMV_HIP_OK;
}

// Hybrid ELLPACK + COO across the devices: the ELLPACK part is cut by rows like a plain ELLPACK matrix, the remainder's
// triplets are dealt to the blocks of their rows (stably: a row keeps its remainder in file order), and every device
// merges its two parts into one row-major matrix exactly like a single-device hybrid upload (spmv_hip_upload_hybrid).
int multi_upload_hybrid(spmv_hip_ctx * c, int32_t rows, int32_t cols, int32_t row_length, const int32_t * ell_col, const double * ell_val,
                        int32_t ncoo, const int32_t * coo_row, const int32_t * coo_col, const double * coo_val)
{
    if (rows < 0 || cols < 0 || row_length < 0 || ncoo < 0 || ((long long) rows * row_length > 0 && (!ell_col || !ell_val))
        || (ncoo > 0 && (!coo_row || !coo_col || !coo_val)))
        return fail(SPMV_HIP_ERR_INVALID, "bad hybrid arguments");
    if ((long long) rows * row_length > INT32_MAX)
        return fail(SPMV_HIP_ERR_OVERFLOW, "Integer overflow when computing number of non-zeros");
    for (int32_t k = 0; k < ncoo; ++k)
        if (coo_row[k] < 0 || coo_row[k] >= rows)
            return fail(SPMV_HIP_ERR_INVALID, "row or column index out of range");
    std::vector<long long> before;
    if (c->flags & SPMV_HIP_FLAG_BALANCE_ENTRIES) { // stored entries in front of every row: its ELLPACK slots + its remainder
        before.assign((size_t) rows + 1, 0);
        for (int32_t k = 0; k < ncoo; ++k)
            ++before[(size_t) coo_row[k] + 1];
        for (int32_t r = 0; r < rows; ++r)
            before[(size_t) r + 1] += before[(size_t) r] + row_length;
    }
    int rc = multi_layout(c, rows, before.empty() ? nullptr : before.data());
    if (rc != 0)
        return multi_upload_failed(c, rc);
    const size_t G = c->parts.size();
    std::vector<uint8_t> block_of_row((size_t) rows);
    for (size_t g = 0; g < G; ++g)
        for (int32_t r = c->row_begin[g]; r < c->row_begin[g + 1]; ++r)
            block_of_row[(size_t) r] = (uint8_t) g;
    std::vector<size_t> start(G + 1, 0);
    for (int32_t k = 0; k < ncoo; ++k)
        ++start[(size_t) block_of_row[(size_t) coo_row[k]] + 1];
    for (size_t g = 0; g < G; ++g)
        start[g + 1] += start[g];
    std::vector<int32_t> ri((size_t) ncoo), ci((size_t) ncoo);
    std::vector<double> va((size_t) ncoo);
    std::vector<size_t> fill(start.begin(), start.end() - 1);
    for (int32_t k = 0; k < ncoo; ++k) {
        const size_t g = block_of_row[(size_t) coo_row[k]];
        const size_t at = fill[g]++;
        ri[at] = coo_row[k] - c->row_begin[g];
        ci[at] = coo_col[k];
        va[at] = coo_val[k];
    }
    for (size_t g = 0; g < G; ++g) {
        const int32_t b = c->row_begin[g], e = c->row_begin[g + 1];
        const size_t eoff = (size_t) b * (size_t) row_length, off = start[g], cnt = start[g + 1] - start[g];
        rc = spmv_hip_upload_hybrid(c->parts[g], e - b, cols, row_length, ell_col ? ell_col + eoff : nullptr, ell_val ? ell_val + eoff : nullptr,
                                    (int32_t) cnt, cnt ? ri.data() + off : nullptr, cnt ? ci.data() + off : nullptr, cnt ? va.data() + off : nullptr);
        if (rc != 0)
            return multi_upload_failed(c, rc);
    }
    c->rows = rows;
    c->cols = cols;
    c->nnz = (int32_t) ((long long) rows * row_length);
    c->nnz2 = ncoo;
    c->format = 4;
    if ((rc = multi_after_upload(c)) != 0)
        return multi_upload_failed(c, rc);
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
    // block of every ROW, once (rows form contiguous blocks; blocks may be empty), then one counting pass and one
    // dealing pass over the entries: no search per entry
    std::vector<uint8_t> block_of_row((size_t) rows);
    for (size_t g = 0; g < G; ++g)
        for (int32_t r = c->row_begin[g]; r < c->row_begin[g + 1]; ++r)
            block_of_row[(size_t) r] = (uint8_t) g;
    std::vector<size_t> start(G + 1, 0);
    for (int32_t k = 0; k < nnz; ++k)
        ++start[(size_t) block_of_row[(size_t) row_index[k]] + 1];
    for (size_t g = 0; g < G; ++g)
        start[g + 1] += start[g];
    std::vector<int32_t> ri((size_t) nnz), ci((size_t) nnz);
    std::vector<double> va((size_t) nnz);
    std::vector<size_t> fill(start.begin(), start.end() - 1);
    for (int32_t k = 0; k < nnz; ++k) {
        const size_t g = block_of_row[(size_t) row_index[k]];
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
    if ((rc = multi_after_upload(c)) != 0)
        return multi_upload_failed(c, rc);
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
    c->cur = 0; // (pipelined contexts: the given vector goes into the first copy, which is the current one from here on)
    c->sent_recorded[0] = c->sent_recorded[1] = false;
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
    if (c->peer_gather || c->pipeline) { // device 0's y is complete once every OTHER device's push (every gather stream) has finished
        int rc0 = multi_sync(c);
        if (rc0 != 0)
            return rc0;
    }
    spmv_hip_ctx * part = c->parts[0];
    HIP_TRY(hipSetDevice(part->device));
    const int cur = c->pipeline ? c->cur : 0;
    if (c->rows > 0 && c->packed)
        HIP_TRY(hipMemcpyAsync(y, ybuf(c, cur, 0), (size_t) c->rows * sizeof(double), hipMemcpyDeviceToHost, part->stream));
    for (size_t h = 0; h < c->parts.size() && !c->packed; ++h) {
        const int32_t b = c->row_begin[h], e = c->row_begin[h + 1];
        if (e > b)
            HIP_TRY(hipMemcpyAsync(y + b, ybuf(c, cur, 0) + h * (size_t) c->chunk, (size_t) (e - b) * sizeof(double), hipMemcpyDeviceToHost,
                                   part->stream));
    }
    HIP_TRY(hipStreamSynchronize(part->stream));
    // SPMV_HIP_FLAG_VERIFY_PLAN on a multi-GPU context: after a gather EVERY device must hold the same y.  The other
    // copies are fetched and compared with device 0's, slot by slot (SPMV_HIP_ERR_STATE names the first that differs).
    if ((c->flags & SPMV_HIP_FLAG_VERIFY_PLAN) && c->rows > 0) {
        int rc0 = multi_sync(c);
        if (rc0 != 0)
            return rc0;
        std::vector<double> other((size_t) c->chunk);
        for (size_t g = 1; g < c->parts.size(); ++g) {
            HIP_TRY(hipSetDevice(c->parts[g]->device));
            for (size_t h = 0; h < c->parts.size(); ++h) {
                const int32_t b = c->row_begin[h], e = c->row_begin[h + 1];
                if (e <= b)
                    continue;
                HIP_TRY(hipMemcpy(other.data(), ybuf(c, cur, g) + h * (size_t) c->chunk, (size_t) (e - b) * sizeof(double), hipMemcpyDeviceToHost));
                if (std::memcmp(other.data(), y + b, (size_t) (e - b) * sizeof(double)) != 0) {
                    char msg[160];
                    std::snprintf(msg, sizeof msg, "after the gather device %zu's copy of y differs from device 0's in the rows of block %zu", g, h);
                    return fail(SPMV_HIP_ERR_STATE, msg);
                }
            }
        }
    }
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

int multi_peer_gather(spmv_hip_ctx * c, int which /* copy of y */, bool on_comm_streams)
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
                    t.dst[t.n++] = ybuf(c, which, (size_t) h) + (size_t) g * chunk;
            for (int k = t.n; k < kPeerFanout; ++k)
                t.dst[k] = nullptr;
            if (t.n > 0)
                hipLaunchKernelGGL(peer_push_kernel, dim3(blocks), dim3(256), 0, on_comm_streams ? c->comm[(size_t) g] : part->stream,
                                   ybuf(c, which, (size_t) g) + (size_t) g * chunk, t, pairs);
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

// the one collective of the path: every device sends its slot of copy `which` and receives the others', in place
static int multi_rccl_gather(spmv_hip_ctx * c, int which, bool on_comm_streams)
{
    const int G = (int) c->parts.size();
    ncclResult_t r = c->p_group_start();
    if (r != ncclSuccess)
        return multi_fail_nccl(c, r, "ncclGroupStart");
    for (int g = 0; g < G && r == ncclSuccess; ++g)
        r = c->p_all_gather(ybuf(c, which, (size_t) g) + (size_t) g * (size_t) c->chunk, ybuf(c, which, (size_t) g), (size_t) c->chunk, ncclDouble,
                            c->comms[(size_t) g], on_comm_streams ? c->comm[(size_t) g] : c->parts[(size_t) g]->stream);
    ncclResult_t r2 = c->p_group_end();
    if (r != ncclSuccess || r2 != ncclSuccess)
        return multi_fail_nccl(c, r != ncclSuccess ? r : r2, "ncclAllGather");
    return SPMV_HIP_OK;
}

// SPMV_HIP_FLAG_PIPELINE_GATHER.  Run k: every device multiplies y_out = y_in + A_g x with y_in = its slot of the current copy
// and y_out = its slot of the OTHER copy, then the gather of that other copy goes to the device's second stream behind an
// event -- so that run k + 1's multiply (which reads what run k wrote, on the same stream, and writes the copy run k read from)
// runs beside gather k.  The one hazard: run k + 1 overwrites slot g of the copy that gather k - 1 may still be sending; its
// multiply therefore waits for ev_sent of that copy (recorded behind gather k - 1 on the gather stream).  Remote writes of a
// gather land in slots h != g of a copy on device g, which no multiply of device g ever touches.
static int multi_run_pipelined(spmv_hip_ctx * c)
{
    const int G = (int) c->parts.size();
    const int in = c->cur, out = c->cur ^ 1;
    for (int g = 0; g < G; ++g) {
        spmv_hip_ctx * part = c->parts[(size_t) g];
        HIP_TRY(hipSetDevice(part->device));
        if (c->sent_recorded[out])
            HIP_TRY(hipStreamWaitEvent(part->stream, c->ev_sent[out][(size_t) g], 0));
        part->y_in_override = ybuf(c, in, (size_t) g) + (size_t) g * (size_t) c->chunk;
        part->d_y = ybuf(c, out, (size_t) g) + (size_t) g * (size_t) c->chunk;
        int rc = spmv_hip_run(part);
        part->y_in_override = nullptr;
        if (rc != 0)
            return rc;
        HIP_TRY(hipEventRecord(c->ev_mul[(size_t) g], part->stream));
        HIP_TRY(hipStreamWaitEvent(c->comm[(size_t) g], c->ev_mul[(size_t) g], 0));
    }
    int rc = c->peer_gather ? multi_peer_gather(c, out, true) : multi_rccl_gather(c, out, true);
    if (rc != 0)
        return rc;
    for (int g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(c->parts[(size_t) g]->device));
        HIP_TRY(hipEventRecord(c->ev_sent[out][(size_t) g], c->comm[(size_t) g]));
        if (!(c->flags & SPMV_HIP_FLAG_NO_RUN_EVENTS))
            HIP_TRY(hipEventRecord(c->ev_gather[(size_t) g], c->comm[(size_t) g]));
    }
    c->sent_recorded[out] = true;
    c->cur = out;
    c->timed = true;
    return SPMV_HIP_OK;
}

int multi_run(spmv_hip_ctx * c)
{
    const int G = (int) c->parts.size();
    if (c->pipeline)
        return multi_run_pipelined(c);
    for (spmv_hip_ctx * part : c->parts) { // every device multiplies its rows into its slot of its y
        int rc = spmv_hip_run(part);
        if (rc != 0)
            return rc;
    }
    if (c->flags & SPMV_HIP_FLAG_FUSED_PEER_STORE) {
        // every part's run has delivered its rows already (spmv_hip_run of a part with peer_y)
    } else if (c->peer_gather) {
        int rc = multi_peer_gather(c, 0, false);
        if (rc != 0)
            return rc;
    } else if (!c->comms.empty()) {
        int rc = multi_rccl_gather(c, 0, false);
        if (rc != 0)
            return rc;
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
    for (size_t g = 0; g < c->parts.size(); ++g) {
        spmv_hip_ctx * part = c->parts[g];
        HIP_TRY(hipSetDevice(part->device));
        HIP_TRY(hipStreamSynchronize(part->stream));
        if (g < c->comm.size() && c->comm[g])
            HIP_TRY(hipStreamSynchronize(c->comm[g]));
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

} // namespace spmvi

extern "C" {

int spmv_hip_partition_rows(int32_t rows, int parts, const int32_t * host_row_ptr, int balance_entries, int32_t * row_begin)
{
    if (rows < 0 || parts < 1 || parts > 4096 || !row_begin || (balance_entries && !host_row_ptr))
        return fail(SPMV_HIP_ERR_INVALID, "spmv_hip_partition_rows: rows >= 0, 1 <= parts <= 4096, row_begin[parts + 1]; row_ptr when the entries are to be balanced");
    if (balance_entries)
        for (int32_t r = 0; r < rows; ++r)
            if (host_row_ptr[r + 1] < host_row_ptr[r])
                return fail(SPMV_HIP_ERR_INVALID, "row_ptr must be non-decreasing");
    partition_rows<int32_t>(rows, parts, balance_entries ? host_row_ptr : nullptr, row_begin);
    return SPMV_HIP_OK;
}

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
    const bool share = share_env && share_env[0] == '1' && (flags & (SPMV_HIP_FLAG_PEER_GATHER | SPMV_HIP_FLAG_FUSED_PEER_STORE));
    if (num_gpus < 1 || (num_gpus > n && !share) || num_gpus > 64)
        return fail(SPMV_HIP_ERR_INVALID, "num_gpus must be between 1 and the number of visible devices");
    spmv_hip_ctx * c = new (std::nothrow) spmv_hip_ctx;
    if (!c)
        return fail(SPMV_HIP_ERR_ALLOC, "ctx allocation failed");
    c->multi = true;
    c->flags = flags;
    c->peer_gather = (flags & (SPMV_HIP_FLAG_PEER_GATHER | SPMV_HIP_FLAG_FUSED_PEER_STORE)) != 0;
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
    const char * force_env = std::getenv("SPMV_HIP_FORCE_RCCL");
    const bool force_one = num_gpus == 1 && force_env && force_env[0] == '1' && !(flags & SPMV_HIP_FLAG_PEER_GATHER);
    if (rc == SPMV_HIP_OK && (flags & SPMV_HIP_FLAG_PIPELINE_GATHER) && (num_gpus > 1 || force_one) && !(flags & SPMV_HIP_FLAG_FUSED_PEER_STORE)) {
        for (int g = 0; g < num_gpus && rc == SPMV_HIP_OK; ++g) {
            hipStream_t st = nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            if (hipSetDevice(c->parts[(size_t) g]->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess
                || hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess
                || hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess)
                rc = fail(SPMV_HIP_ERR_HIP, "stream / events of SPMV_HIP_FLAG_PIPELINE_GATHER");
            c->comm.push_back(st);
            c->ev_mul.push_back(e0);
            c->ev_sent[0].push_back(e1);
            c->ev_sent[1].push_back(e2);
        }
    }
    const char * force = std::getenv("SPMV_HIP_FORCE_RCCL");
    if (rc == SPMV_HIP_OK && c->peer_gather)
        rc = multi_enable_peers(c);
    else if (rc == SPMV_HIP_OK && (num_gpus > 1 || (force && force[0] == '1')))
        rc = multi_load_rccl(c, num_gpus);
    if (rc != SPMV_HIP_OK) {
        std::string const keep = last_error_text();
        spmv_hip_destroy(c);
        set_last_error_text(keep);
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
