// peer_gather.hip -- one process per GPU (python/spmv_amd/distributed.py, bench.py --gpus N): the y segments reach the
// other ranks' vectors as plain stores into THEIR memory, mapped into this process with HIP's inter-process handles --
// the "direct all-gather over point-to-point links" of SURVEY section 5, without a collective call.
//
//   every rank:  spmv_hip_ipc_alloc   its copy of the whole y (device memory + a 64-byte handle)
//   exchange the handles (the callers' process group), then
//                spmv_hip_ipc_open    the other ranks' copies -> device pointers valid in this process
//   per multiply: spmv_hip_csr_spmv_out_peers (launch.hip): the kernel itself stores every row sum into all copies,
//                or -- plans whose kernels cannot -- spmv_hip_peer_push after the multiply: one kernel that reads the
//                segment once and writes it to every peer.
// Nothing is received by a kernel: a rank's copy is complete when every rank has synchronised its stream and the
// ranks have met at a barrier (DistributedCsrSpmv.finish).  Row blocks are disjoint, so ranks that drift apart never
// touch the same doubles.
#include "internal.hpp"

#include <algorithm>
#include <cstring>

using namespace spmvi;

namespace {

constexpr int kPushFanout = 8;
struct PushTargets {
    double * dst[kPushFanout];
    int n;
};

// one double per lane (a segment starts at rank * chunk doubles: 8-byte aligned only); a wave still writes 512
// contiguous bytes per store instruction and peer
__global__ __launch_bounds__(256) void segment_push_kernel(const double * __restrict__ src, PushTargets t, long long n)
{
    const long long stride = (long long) gridDim.x * 256;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = src[i];
#pragma unroll
        for (int k = 0; k < kPushFanout; ++k)
            if (k < t.n)
                t.dst[k][i] = v;
    }
}

} // namespace

extern "C" {

int spmv_hip_ipc_alloc(void ** d_ptr, size_t bytes, void * handle64)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands the handle over as 64 opaque bytes");
    if (!d_ptr || !handle64 || bytes == 0)
        return fail(SPMV_HIP_ERR_INVALID, "null argument or zero size");
    *d_ptr = nullptr;
    void * p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    // a device memset may return before it has run: nothing orders another rank's stores (other process, other device)
    // behind it, so the zeroes must be in memory before the handle exists -- "the memory comes back zeroed"
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess)
        e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void) hipFree(p);
        return fail_hip(e, "hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this driver)");
    }
    std::memcpy(handle64, &h, sizeof h);
    *d_ptr = p;
    return SPMV_HIP_OK;
}

int spmv_hip_ipc_open(const void * handle64, void ** d_ptr)
{
    if (!handle64 || !d_ptr)
        return fail(SPMV_HIP_ERR_INVALID, "null argument");
    *d_ptr = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void * p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *d_ptr = p;
    return SPMV_HIP_OK;
}

int spmv_hip_ipc_close(void * d_ptr)
{
    if (!d_ptr)
        return SPMV_HIP_OK;
    HIP_TRY(hipIpcCloseMemHandle(d_ptr));
    return SPMV_HIP_OK;
}

int spmv_hip_ipc_free(void * d_ptr)
{
    if (!d_ptr)
        return SPMV_HIP_OK;
    HIP_TRY(hipFree(d_ptr));
    return SPMV_HIP_OK;
}

int spmv_hip_peer_push(const double * d_src, double * const * d_dst, int ndst, int64_t n, void * stream)
{
    if (ndst < 0 || n < 0 || (ndst > 0 && n > 0 && (!d_src || !d_dst)))
        return fail(SPMV_HIP_ERR_INVALID, "bad push arguments");
    if (ndst == 0 || n == 0)
        return SPMV_HIP_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned) std::min<long long>((n + 255) / 256, 8ll * cu_count());
    for (int k0 = 0; k0 < ndst; k0 += kPushFanout) {
        PushTargets t{};
        for (int k = k0; k < ndst && k < k0 + kPushFanout; ++k) {
            if (!d_dst[k])
                return fail(SPMV_HIP_ERR_INVALID, "null peer pointer");
            t.dst[t.n++] = d_dst[k];
        }
        hipLaunchKernelGGL(segment_push_kernel, dim3(blocks), dim3(256), 0, s, d_src, t, (long long) n);
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

} // extern "C"
