// Error reporting, ABI version and the fp64 log-likelihood accumulator.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <unordered_set>
#include <unordered_map>
#include <vector>

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

namespace {
constexpr int kHintSlots = 1024;
struct HintPool {
    std::mutex mu;
    int *host = nullptr, *dev = nullptr;
    bool failed = false;
    std::unordered_map<const void *, int> slot;
    std::vector<int> free_list;
    unsigned seq[kHintSlots] = {};
    int next = 0;
};
HintPool &hint_pool() {
    static HintPool p;
    return p;
}
}  // namespace

bool slow_hint_next(const void *ws_key, int **dev_word, int *launch_seq) {
    HintPool &hp = hint_pool();
    *dev_word = nullptr;
    *launch_seq = 0;
    std::lock_guard<std::mutex> lock(hp.mu);
    if (hp.failed) return false;
    if (hp.host == nullptr) {
        int *p = nullptr, *d = nullptr;
        if (hipHostMalloc((void **)&p, kHintSlots * sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !p ||
            hipHostGetDevicePointer((void **)&d, p, 0) != hipSuccess || !d) {
            (void)hipGetLastError();
            hp.failed = true;
            return false;
        }
        for (int i = 0; i < kHintSlots; ++i) p[i] = -1000;
        hp.host = p;
        hp.dev = d;
    }
    auto it = hp.slot.find(ws_key);
    if (it == hp.slot.end()) {
        int idx;
        if (!hp.free_list.empty()) {
            idx = hp.free_list.back();
            hp.free_list.pop_back();
        } else if (hp.next < kHintSlots) {
            idx = hp.next++;
        } else {
            return false;   // (no slot: this workspace runs without the hint -- the default variants, same results)
        }
        hp.host[idx] = -1000;
        hp.seq[idx] = 0;
        it = hp.slot.emplace(ws_key, idx).first;
    }
    const int idx = it->second;
    *dev_word = hp.dev + idx;
    *launch_seq = (int)++hp.seq[idx];
    // the host runs ahead of the device by its launch queue: "recent" = within 256 launches
    return (unsigned)(*launch_seq - *(volatile int *)(hp.host + idx)) <= 256u;
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

// ---- parameter fingerprints (common.h: params_gate) -----------------------------------------------------------------
struct FpArgs {
    FpSeg seg[kFpMaxSegs];
    int nseg;
    int verify;
    unsigned long long *state;   // [0] hash of the bytes the tables were built from, [1] accumulator, [2] lo: tickets, hi: gate
};
constexpr int64_t kFpSpan = 16384;   // bytes hashed per work-group


__global__ __launch_bounds__(256) void params_fingerprint_kernel(const FpArgs a) {
    // block b hashes bytes [b, b + 1) * kFpSpan of the concatenated segments: sum over 4-byte words (single bytes
    // where a segment is not word aligned) of mix(word ^ mix(position)): order independent, so the blocks meet in one
    // 64-bit atomic sum
    const int64_t lo = (int64_t)blockIdx.x * kFpSpan, hi = lo + kFpSpan;
    unsigned long long h = 0ull;
    int64_t base = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const int64_t n = a.seg[i].bytes;
        const int64_t s0 = lo > base ? lo - base : 0, s1 = (hi - base) < n ? (hi - base) : n;   // span within the segment
        if (s0 < s1) {
            const unsigned char *p = (const unsigned char *)a.seg[i].p;
            const bool words = (((uintptr_t)p | (uintptr_t)n) & 3) == 0;   // (kFpSpan is a multiple of 4)
            if (words) {
                const unsigned *w = (const unsigned *)p;
                for (int64_t e = (s0 >> 2) + threadIdx.x; e < (s1 >> 2); e += blockDim.x)
                    h += fp_mix((unsigned long long)w[e] ^ fp_mix(((unsigned long long)(i + 1) << 48) ^ (unsigned long long)e));
            } else {
                for (int64_t e = s0 + threadIdx.x; e < s1; e += blockDim.x)
                    h += fp_mix((unsigned long long)p[e] ^ fp_mix(((unsigned long long)(i + 65) << 48) ^ (unsigned long long)e));
            }
        }
        base += n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += (unsigned long long)__shfl_xor((long long)h, o, 64);
    __shared__ unsigned long long part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *tick = reinterpret_cast<unsigned *>(a.state + 2);
        atomicAdd(a.state + 1, part[0] + part[1] + part[2] + part[3]);
        // (the partial results above are device-scope atomics: their acknowledgement is all the release there is to wait for;
        // a __threadfence() writes back this XCD's L2 -- 17 .. 40 us behind a kernel that left it dirty, round-4 measurement)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (atomicAdd(tick, 1u) == gridDim.x - 1) {   // last block: every partial sum has arrived (read back with atomics)
            const unsigned long long sum = atomicExch(a.state + 1, 0ull);
            tick[1] = (!a.verify || sum != a.state[0]) ? 1u : 0u;
            a.state[0] = sum;
            tick[0] = 0u;
        }
    }
}

__global__ void gated_zero_kernel(uint4 *p, int64_t n16, const unsigned *gate) {
    if (gate_closed(gate)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = uint4{0u, 0u, 0u, 0u};
}

int gated_zero(void *p, int64_t bytes, const unsigned *gate, hipStream_t st) {
    DPK_REQUIRE(((uintptr_t)p & 15) == 0 && (bytes & 15) == 0, DPK_EINVAL, "gated_zero: unaligned");
    if (bytes == 0) return DPK_OK;
    int grid = cdiv(bytes / 16, 256 * 4);
    if (grid > 2048) grid = 2048;
    DPK_LAUNCH(gated_zero_kernel, dim3(grid), dim3(256), 0, st, (uint4 *)p, bytes / 16, gate);
    DPK_CHECK_LAUNCH("gated_zero_kernel");
    return DPK_OK;
}

namespace {
struct GatePool {
    unsigned long long *base = nullptr;
    std::unordered_map<const void *, int> slot;
    std::vector<int> free_list;
    int next = 0;
    bool failed = false;
};
struct GatePools {
    std::mutex mu;
    GatePool pools[kMaxDevices];
};
GatePools &gate_pools() {
    static GatePools g;
    return g;
}
}  // namespace

// Slots are keyed by an address inside the workspace that holds the table set; a workspace that goes away hands them
// back (deeprob.hip.Workspace.__del__ -> dpk_workspace_forget), so a long-lived process that builds many models does not
// run the 4096-slot pool dry (round-3 advice: exhausted, every verifying call silently became a rebuild).  A recycled
// slot's first use is an unconditional build, whatever its old contents.  One module evaluates on one stream at a time:
// the slot's accumulator and ticket are per table set, not per stream.
void workspace_forget(const void *base, int64_t bytes) {
    const char *lo = (const char *)base, *hi = lo + bytes;
    {
        GatePools &gp = gate_pools();
        std::lock_guard<std::mutex> lock(gp.mu);
        for (int d = 0; d < kMaxDevices; ++d) {
            GatePool &pl = gp.pools[d];
            for (auto it = pl.slot.begin(); it != pl.slot.end();) {
                const char *k = (const char *)it->first;
                if (k >= lo && k < hi) {
                    pl.free_list.push_back(it->second);
                    it = pl.slot.erase(it);
                } else {
                    ++it;
                }
            }
        }
    }
    {
        HintPool &hp = hint_pool();
        std::lock_guard<std::mutex> lock(hp.mu);
        for (auto it = hp.slot.begin(); it != hp.slot.end();) {
            const char *k = (const char *)it->first;
            if (k >= lo && k < hi) {
                hp.free_list.push_back(it->second);
                it = hp.slot.erase(it);
            } else {
                ++it;
            }
        }
    }
}

const unsigned *params_gate(const void *key, const FpSeg *segs, int nseg, bool verify, hipStream_t st) {
    // state slots: one 32-byte record per table set, carved from a per-device pool the library owns (the first use of
    // a device allocates it -- outside any stream capture: captured steps are preceded by eager ones)
    constexpr int kSlots = 4096;
    GatePools &gp = gate_pools();
    std::mutex &mu = gp.mu;
    GatePool *pools = gp.pools;
    if (nseg < 1 || nseg > kFpMaxSegs) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    unsigned long long *state = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        GatePool &pl = pools[dev];
        if (pl.failed) return nullptr;
        if (pl.base == nullptr) {
            if (hipMalloc(&pl.base, (size_t)kSlots * 32) != hipSuccess || hipMemset(pl.base, 0, (size_t)kSlots * 32) != hipSuccess) {
                (void)hipGetLastError();
                pl.base = nullptr;
                pl.failed = true;
                return nullptr;
            }
        }
        auto it = pl.slot.find(key);
        if (it == pl.slot.end()) {
            int idx;
            if (!pl.free_list.empty()) {
                idx = pl.free_list.back();
                pl.free_list.pop_back();
            } else if (pl.next < kSlots) {
                idx = pl.next++;
            } else {
                return nullptr;
            }
            it = pl.slot.emplace(key, idx).first;
            verify = false;   // a fresh (or recycled) slot holds no valid hash: this call's build is unconditional
        }
        state = pl.base + (int64_t)it->second * 4;
    }
    FpArgs a{};
    int64_t total = 0;
    for (int i = 0; i < nseg; ++i) {
        a.seg[i] = segs[i];
        if (segs[i].p == nullptr) a.seg[i].bytes = 0;
        total += a.seg[i].bytes;
    }
    a.nseg = nseg;
    a.verify = verify ? 1 : 0;
    a.state = state;
    const int grid = total > 0 ? cdiv(total, kFpSpan) : 1;
    DPK_LAUNCH(params_fingerprint_kernel, dim3(grid), dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return reinterpret_cast<const unsigned *>(state + 2) + 1;
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
extern "C" int dpk_workspace_forget(const void *base, int64_t bytes) {
    if (base == nullptr || bytes <= 0) return DPK_OK;
    dpk::workspace_forget(base, bytes);
    return DPK_OK;
}
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
