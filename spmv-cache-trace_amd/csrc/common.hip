// common.hip -- error reporting, library queries and device properties of libspmv_hip.so (include/spmv_hip.h).
// No CPU compute path exists in this library: if HIP cannot run, the entry points return an error.
#include "internal.hpp"

#include <cstdio>

using namespace spmvi;

namespace {
thread_local std::string g_last_error;
}

namespace spmvi {

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

std::string last_error_text() { return g_last_error; }
void set_last_error_text(std::string const & text) { g_last_error = text; }

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

int grid_for(long long work_items, int per_block, int max_blocks)
{
    if (max_blocks <= 0)
        max_blocks = cu_count() * 8;
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int) g;
}

} // namespace spmvi

int spmv_hip_internal_fail_hip(hipError_t e, const char * call) { return spmvi::fail_hip(e, call); }

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
