// Error reporting, ABI version and the fp64 log-likelihood accumulator.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <unordered_set>

namespace dpk {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int kMaxDevices = 64;
int device_cus() {
    static int cus[kMaxDevices] = {0};   // (a racing first call writes the same value twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

int ensure_dynamic_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_set<uint64_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t key = (uint64_t)(uintptr_t)kernel * 64u + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return DPK_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(max dynamic LDS = %d): %s", bytes, hipGetErrorString(e));
        return DPK_ELAUNCH;
    }
    done.insert(key);
    return DPK_OK;
}

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local int g_ev_kernel = 0;
void profile_take(hipEvent_t *a, hipEvent_t *b, int kernel_id) {
    if (g_ev_kernel != 0 && g_ev_kernel != kernel_id) {
        *a = *b = nullptr;
        return;
    }
    *a = g_ev_start;
    *b = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
}

// acc[0] += sum(ll), acc[1] += n  -- the per-rank partial of the mean-LL all-reduce
// (the reference averages on the host: deeprob/torch/routines.py:419-425).
__global__ void ll_accumulate_kernel(const float *__restrict__ ll, int64_t n, double *acc) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        s += (double)ll[i];
    s = wave_reduce_sum(s);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
        if (blockIdx.x == 0) atomicAdd(acc + 1, (double)n);
    }
}
}  // namespace dpk

extern "C" const char *dpk_last_error(void) { return dpk::g_err; }
extern "C" int dpk_abi_version(void) { return 1; }
extern "C" int dpk_profile_next_kernel(void *ev_start, void *ev_stop) {
    dpk::g_ev_start = (hipEvent_t)ev_start;
    dpk::g_ev_stop = (hipEvent_t)ev_stop;
    dpk::g_ev_kernel = 0;
    return DPK_OK;
}
extern "C" int dpk_profile_next_kernel_of(void *ev_start, void *ev_stop, int32_t kernel_id) {
    dpk::g_ev_start = (hipEvent_t)ev_start;
    dpk::g_ev_stop = (hipEvent_t)ev_stop;
    dpk::g_ev_kernel = kernel_id;
    return DPK_OK;
}

extern "C" int dpk_ll_accumulate(const float *ll, int64_t n, double *acc, void *stream) {
    DPK_REQUIRE(ll && acc && n >= 0, DPK_EINVAL, "ll_accumulate: bad argument");
    if (n == 0) return DPK_OK;
    int grid = dpk::cdiv(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    DPK_LAUNCH(dpk::ll_accumulate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ll, n, acc);
    DPK_CHECK_LAUNCH("ll_accumulate_kernel");
    return DPK_OK;
}

#ifdef DPK_TIMELINE
// measurement builds: copy the timeline buffer of the last forward launch to the host
extern "C" int dpk_debug_read(void *dev, void *host, int64_t bytes) {
    return hipMemcpy(host, dev, (size_t)bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
