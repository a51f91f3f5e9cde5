// Training route of the RealNVP-1D path on gfx950: backward of CouplingLayer1d.apply_backward (depth-1
// conditioner), train-mode BatchNormLayer1d (batch statistics) forward / backward, Normal base backward.
// Formulas: SURVEY 8a "Backward formulas" (checked there against the reference's autograd in float64).
//
// The conditioner's five backward GEMMs (recomputed forward H, Z; dW2 = dZ^T H; dH = dZ W2; dW1 = dH^T xm;
// dx += mask * dH W1) all go through ONE generic fp32-MFMA kernel (v_mfma_f32_32x32x2_f32, 64x64 tile per
// 4-wave work-group, operands addressed through strides so the transposed products need no copies).  Training
// batches are hundreds of samples, so this route is correctness-first; the 64k-sample density path is the
// fused forward kernel in coupling.hip.
#include "common.h"
#include <mutex>
#include <type_traits>
#include <math.h>

namespace dpk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *A, *Bm;
    float *C;
    int M, N, K;
    int64_t sam, sak, sbk, sbn, ldc;
    const float *kscale;   // A(m,k) *= kscale[k]
    const float *nscale;   // result(m,n) *= nscale[n]
    const float *bias;     // + bias[n] (before relu)
    const float *gate;     // result zeroed where gate[m*ldg + n] <= 0
    int64_t ldg;
    int relu, accumulate;
    int ksplit, kchunk;    // > 1: blockIdx.z owns K range [z*kchunk, (z+1)*kchunk)
    float *partials;       // split-K: [tile][slice][16][256] partial tiles; the last slice to finish a tile sums them in slice
    unsigned *tickets;     // order and runs the epilogue ([tile] arrival counts, zero between launches).  Null: the partial
};                         // sums meet by atomicAdd into a zeroed C and gemm_epilogue_kernel follows (fallback)

constexpr int kGT = 64, kGK = 64;

// 64 x 64 output tile per work-group, K walked in slabs of 64: the next slab's operands are fetched into registers
// (16 + 16 loads in flight per thread) before the MFMAs of the current one, so a slab costs max(load latency, MFMA
// time) instead of their sum; products too small to fill the chip are split over K by launch_gemm.
// The operand orientations are template parameters (A contiguous along k or along m, B along n or along k): a thread's
// 16 + 16 loads per slab are one base pointer plus constant steps, advanced by one add per slab.
template <bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ float As[kGT][kGK + 1];
    __shared__ float Bs[kGK][kGT + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Work-groups are dealt round-robin to the 8 XCDs (each with its own L2): renumber them so that every XCD walks
    // a contiguous run of tiles (n fastest), i.e. the tiles sharing a row block of A meet in one L2.
    const unsigned n_tx = gridDim.x, n_tiles = gridDim.x * gridDim.y, bid = blockIdx.x + n_tx * blockIdx.y;
    const unsigned xcd = bid & 7u, per = n_tiles >> 3, rem = n_tiles & 7u;
    const unsigned tile = xcd * per + min(xcd, rem) + (bid >> 3);
    const int m0 = (int)(tile / n_tx) * kGT, n0 = (int)(tile % n_tx) * kGT;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kbeg = (g.ksplit > 1) ? (int)blockIdx.z * g.kchunk : 0;
    const int kend = (g.ksplit > 1) ? min(g.K, kbeg + g.kchunk) : g.K;
    constexpr int kPer = kGT * kGK / 256;       // 16 elements of each operand per thread and slab
    // element i of this thread: fast index f (the operand's contiguous axis), slow index s0 + 4 i.
    // Every load is unconditional at a clamped (in-range) address and zeroed afterwards: the first version predicated
    // each load with a branch, and hipcc waits for ALL outstanding loads where such a branch joins -- the 32 loads of a
    // slab were 32 dependent round trips, 9 us per slab (round-4 trace: 18 us for a two-slab product).
    const int f = lane, s0 = wave;
    // (a second slab of operands in flight in registers was measured: no gain -- the slab is not waiting for its loads)
    float ar[kPer], br[kPer];
    auto fetch_with = [&](int k0, auto has_kscale) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int am = A_KFAST ? m0 + s0 + 4 * i : m0 + f, ak = A_KFAST ? k0 + f : k0 + s0 + 4 * i;
            const int bk = B_NFAST ? k0 + s0 + 4 * i : k0 + f, bn = B_NFAST ? n0 + f : n0 + s0 + 4 * i;
            const bool aok = am < g.M && ak < kend, bok = bk < kend && bn < g.N;
            const int amc = min(am, g.M - 1), akc = min(ak, g.K - 1), bkc = min(bk, g.K - 1), bnc = min(bn, g.N - 1);
            float av = g.A[(int64_t)amc * g.sam + (int64_t)akc * g.sak];
            if constexpr (decltype(has_kscale)::value) av *= g.kscale[akc];
            const float bv = g.Bm[(int64_t)bkc * g.sbk + (int64_t)bnc * g.sbn];
            ar[i] = aok ? av : 0.f;
            br[i] = bok ? bv : 0.f;
        }
    };
    auto fetch = [&](int k0) {
        if (g.kscale != nullptr) fetch_with(k0, std::true_type{});
        else fetch_with(k0, std::false_type{});
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += kGK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (A_KFAST) As[s0 + 4 * i][f] = ar[i]; else As[f][s0 + 4 * i] = ar[i];
            if (B_NFAST) Bs[s0 + 4 * i][f] = br[i]; else Bs[f][s0 + 4 * i] = br[i];
        }
        __syncthreads();
        if (k0 + kGK < kend) fetch(k0 + kGK);
#pragma unroll 8
        for (int kk = 0; kk < kGK; kk += 2) {
            const float a = As[wm + (lane & 31)][kk + (lane >> 5)];
            const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn + (lane & 31);
    if (g.ksplit > 1 && g.partials != nullptr) {
        // Split-K without atomics on C (round 4: 0.8 - 1.6 M same-address float atomics per product were most of these
        // launches, plus a memset in front and an epilogue launch behind): every slice stores its 64 x 64 partial tile
        // device-coherently, the LAST slice of a tile to arrive adds them up in slice order (deterministic) and runs the
        // ordinary epilogue.  Release = acknowledged stores (no L2 write-back fence: ratspn_level_bwd.hip).
        __shared__ unsigned last_s;
        float *P = g.partials + ((int64_t)tile * g.ksplit) * 4096 + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __hip_atomic_store(P + (int64_t)blockIdx.z * 4096 + r * 256, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) last_s = (atomicAdd(g.tickets + tile, 1u) == (unsigned)g.ksplit - 1u) ? 1u : 0u;
        __syncthreads();
        if (last_s == 0u) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // (four slices' loads in flight at a time: one slice per trip was one round trip per slice; summed in slice order)
        for (int z0 = 0; z0 < g.ksplit; z0 += 4) {
            float v[4][16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int z = min(z0 + q, g.ksplit - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[q][r] = __hip_atomic_load(P + (int64_t)z * 4096 + r * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = z0 + q < g.ksplit;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += in ? v[q][r] : 0.f;
            }
        }
        if (tid == 0) __hip_atomic_store(g.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (g.ksplit > 1) {
        if (n >= g.N) return;
        const float ns = g.nscale ? g.nscale[n] : 1.f;
        // fallback: raw partial sums (times the per-column scale, which is linear) are added into C; bias / ReLU /
        // gate are applied by gemm_epilogue_kernel once every slice has landed
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < g.M) atomicAdd(g.C + (int64_t)m * g.ldc + n, acc[r] * ns);
        }
        return;
    }
    if (n >= g.N) return;
    const float ns = g.nscale ? g.nscale[n] : 1.f, bs = g.bias ? g.bias[n] : 0.f;
    // gate values and the old C (accumulate) of all 16 rows are requested together, each under ONE uniform branch: a load
    // behind a per-row condition was one dependent round trip per row (16 of them: the gated product was the slowest)
    float gv[16], cv[16];
    if (g.gate != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), g.M - 1);
            gv[r] = g.gate[(int64_t)m * g.ldg + n];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[r] = 1.f;
    }
    if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), g.M - 1);
            cv[r] = g.C[(int64_t)m * g.ldc + n];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r] + bs;
        if (g.relu) v = fmaxf(v, 0.f);
        v *= ns;
        v = (gv[r] > 0.f) ? v : 0.f;
        if (m < g.M) g.C[(int64_t)m * g.ldc + n] = cv[r] + v;
    }
}

static void launch_gemm_kernel(const GemmArgs &g, dim3 grid, hipStream_t st) {
    const bool ak = g.sak == 1, bn = g.sbn == 1;   // any other stride pair runs as the "slow axis" orientation
    if (ak && bn) DPK_LAUNCH((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (ak) DPK_LAUNCH((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (bn) DPK_LAUNCH((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
    else DPK_LAUNCH((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
}

// bias / ReLU / gate of a split-K product (the per-column scale was applied to the partial sums; with a bias the
// scale is 1 in every call site, so the order bias -> relu -> scale of the unsplit epilogue is preserved)
__global__ void gemm_epilogue_kernel(const GemmArgs g) {
    const int64_t total = (int64_t)g.M * g.N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % g.N);
        const int64_t m = e / g.N;
        float v = g.C[m * g.ldc + n] + (g.bias ? g.bias[n] : 0.f);
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.gate && !(g.gate[m * g.ldg + n] > 0.f)) v = 0.f;
        g.C[m * g.ldc + n] = v;
    }
}

// Few output tiles and a long K (the training-batch products dH = dZ W2, H = x W1^T, dW = dOut^T In): split K over
// blockIdx.z so that the grid fills the chip.  Scratch for the partial tiles and the arrival counts comes from a pool PER
// DEVICE (64 MB ring + 64 K counters on the device that is current at the launch, allocated at the first product that
// needs it there; a first use inside a stream capture cannot allocate and takes the atomicAdd fallback).  A process that
// trains on a second GPU must not store its partials and tickets into device 0's memory: that faults without peer access,
// and with it the agent-scope ticket protocol does not order memory across devices.  A product takes a fresh region of the
// ring, so products of one stream never meet.  Restriction: one stream per device at a time -- more than ~8 split products
// in flight at once on DIFFERENT streams of one device would wrap the ring into regions still in use.
struct GemmPool {
    std::mutex mu;
    float *partials = nullptr;
    unsigned *tickets = nullptr;
    int64_t cursor = 0, tcursor = 0;
    bool failed = false;
};
constexpr int64_t kGemmPoolFloats = 16ll << 20;   // 64 MB
constexpr int kGemmPoolTickets = 1 << 16;
constexpr int kGemmPoolDevices = 16;
static bool gemm_pool_take(int64_t floats, int tiles, float **p, unsigned **t, hipStream_t st) {
    static GemmPool pools[kGemmPoolDevices];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kGemmPoolDevices) {
        (void)hipGetLastError();
        return false;              // (no pool for this device: the atomicAdd fallback)
    }
    GemmPool &pool = pools[dev];
    std::lock_guard<std::mutex> lock(pool.mu);
    if (pool.failed || floats > kGemmPoolFloats / 2 || tiles > kGemmPoolTickets / 2) return false;
    if (pool.partials == nullptr) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return false;          // (an allocation would invalidate the capture: the fallback this once)
        }
        void *a = nullptr, *b = nullptr;
        if (hipMalloc(&a, (size_t)kGemmPoolFloats * 4) != hipSuccess || hipMalloc(&b, (size_t)kGemmPoolTickets * 4) != hipSuccess ||
            hipMemset(b, 0, (size_t)kGemmPoolTickets * 4) != hipSuccess) {
            (void)hipGetLastError();
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            (void)hipGetLastError();
            return false;          // (e.g. inside a stream capture: ask again next time)
        }
        pool.partials = (float *)a;
        pool.tickets = (unsigned *)b;
    }
    if (pool.cursor + floats > kGemmPoolFloats) pool.cursor = 0;
    if (pool.tcursor + tiles > kGemmPoolTickets) pool.tcursor = 0;
    *p = pool.partials + pool.cursor;
    *t = pool.tickets + pool.tcursor;
    pool.cursor += (floats + 63) / 64 * 64;
    pool.tcursor += tiles;
    return true;
}

static void launch_gemm(const GemmArgs &g_in, hipStream_t st) {
    if (g_in.M <= 0 || g_in.N <= 0) return;
    GemmArgs g = g_in;
    const int tiles = cdiv(g.N, kGT) * cdiv(g.M, kGT);
    int ksplit = 1;
    const int target = device_cus();                          // one work-group per compute unit: a slab is ~1.5 us
    if (tiles < target / 2 + target / 4 && g.K >= 2 * kGK) {
        ksplit = cdiv(target, tiles);
        const int max_split = cdiv(g.K, kGK);
        if (ksplit > max_split) ksplit = max_split;
        if (ksplit > 8) ksplit = 8;   // (the last slice reads them all back: 13 slices of one slab measured 21 us, 7 of two 14)
    }
    if (ksplit > 1) {
        g.kchunk = (int)align_up(cdiv(g.K, ksplit), kGK);
        g.ksplit = cdiv(g.K, g.kchunk);
    }
    if (g.ksplit > 1) {
        if (gemm_pool_take((int64_t)tiles * g.ksplit * 4096, tiles, &g.partials, &g.tickets, st)) {
            launch_gemm_kernel(g, dim3(cdiv(g.N, kGT), cdiv(g.M, kGT), g.ksplit), st);
            return;
        }
        g.partials = nullptr;
        g.tickets = nullptr;
        if (!(g.bias && g.nscale)) {   // (the fallback's epilogue order: bias -> relu, the scale already applied)
            if (!g.accumulate) {
                if (g.ldc == g.N) (void)hipMemsetAsync(g.C, 0, (size_t)g.M * g.N * 4, st);
                else (void)hipMemset2DAsync(g.C, (size_t)g.ldc * 4, 0, (size_t)g.N * 4, (size_t)g.M, st);
            }
            launch_gemm_kernel(g, dim3(cdiv(g.N, kGT), cdiv(g.M, kGT), g.ksplit), st);
            if (g.bias || g.relu || g.gate) {
                const int64_t total = (int64_t)g.M * g.N;
                const int64_t nb = (total + 255) / 256;
                DPK_LAUNCH(gemm_epilogue_kernel, dim3((int)(nb > 4096 ? 4096 : nb)), dim3(256), 0, st, g);
            }
            return;
        }
    }
    g.ksplit = 1;
    launch_gemm_kernel(g, dim3(cdiv(g.N, kGT), cdiv(g.M, kGT)), st);
}

// Column reducers over the batch: a work-group of 1024 threads owns kRC = 16 adjacent columns, its 64 row groups
// stride the batch (a training batch of 512 rows leaves 8 loads per thread; 64-column blocks left the chip to
// a dozen work-groups).  colblock_sum returns the column total to every thread of that column.
constexpr int kRC = 16;
constexpr int kRThreads = 1024;
__device__ inline float colblock_sum(float v, float (*red)[kRC]) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane < kRC) red[wave][lane] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kRThreads / 64; ++w) s += red[w][threadIdx.x & (kRC - 1)];
    return s;
}

// out[n] = sum_m src[m*ld + n]
__global__ __launch_bounds__(kRThreads) void colsum_kernel(const float *__restrict__ src, int64_t M, int N, int64_t ld,
                                                           float *__restrict__ out) {
    __shared__ float red[kRThreads / 64][kRC];
    const int c = blockIdx.x * kRC + (threadIdx.x & (kRC - 1)), rg = threadIdx.x >> 4;
    float s = 0.f;
    if (c < N)
        for (int64_t m = rg; m < M; m += kRThreads / kRC) s += src[m * ld + c];
    s = colblock_sum(s, red);
    if (rg == 0 && c < N) out[c] = s;
}

// element-wise core of the coupling backward.  Z holds [t_hat | s_hat] (affine) or z (NICE); it is
// overwritten with dZ.  gx receives the direct term g_u * exp(-s).
//   ds = -g_u u - g_ildj ; dt = -g_u e^{-s} ; ds_hat = inv_mask ds a (1 - tanh^2) ; da = sum inv_mask ds tanh
// inverse != 0: the sampling direction (coupling.py:89-104), x_out = x e^{s} + t, ldj = +sum s:
//   ds = g x e^{s} + g_ldj ; dt = g ; direct term g e^{s}
// A wave per sample row (rows strided over the grid), lanes over the columns four at a time with every load of the four
// requested before anything is used: the first version (a thread per element, `e % D`, loads behind `if (live)`) was a
// chain of dependent round trips and 1024 same-address atomics per launch -- 17 us for 400 k elements (round-4 trace).
__global__ __launch_bounds__(256) void coupling_bwd_elem_kernel(const float *__restrict__ x, float *__restrict__ Z,
                                                                const float *__restrict__ inv_mask,
                                                                const float *__restrict__ act_weight,
                                                                const float *__restrict__ gu,
                                                                const float *__restrict__ gildj, int64_t B, int D,
                                                                int affine, int inverse, float *__restrict__ gx,
                                                                float *__restrict__ gact) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float a = affine ? act_weight[0] : 0.f;
    float da = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
        const float gl = gildj ? gildj[b] : 0.f;
        for (int d0 = lane; d0 < D; d0 += 64 * U) {
            float g[U], mk[U], xv[U], zt[U], zs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int dc = min(d0 + 64 * u, D - 1);
                g[u] = gu ? gu[b * D + dc] : 0.f;
                mk[u] = inv_mask[dc];
                xv[u] = x[b * D + dc];
                zt[u] = affine ? Z[b * 2 * D + dc] : 0.f;
                zs[u] = affine ? Z[b * 2 * D + D + dc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int d = d0 + 64 * u;
                if (d >= D) continue;
                const bool live = mk[u] != 0.f;
                if (affine) {
                    const float th = tanhf(zs[u]);
                    const float s = a * th;
                    const float es = expf(inverse ? s : -s);
                    const float ds = inverse ? g[u] * xv[u] * es + gl : -g[u] * ((xv[u] - zt[u]) * es) - gl;
                    Z[b * 2 * D + d] = live ? (inverse ? g[u] : -g[u] * es) : 0.f;
                    Z[b * 2 * D + D + d] = live ? ds * a * (1.f - th * th) : 0.f;
                    if (live) da += ds * th;
                    gx[b * D + d] = live ? g[u] * es : g[u];
                } else {
                    Z[b * D + d] = live ? (inverse ? g[u] : -g[u]) : 0.f;
                    gx[b * D + d] = g[u];
                }
            }
        }
    }
    if (affine && gact) {   // one atomic per work-group (every wave adding on its own serialised on the one address)
        __shared__ float wsum[4];
        da = wave_reduce_sum(da);
        if (lane == 0) wsum[wave] = da;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            if (t != 0.f) atomicAdd(gact, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNormLayer1d, training mode (deeprob/flows/utils.py:118-139)
// ---------------------------------------------------------------------------------------------------------
// per-column mean and unbiased variance over the batch (two passes), running statistics updated in place,
// and the affine (scale, shift) + constant log-det of the transformation with these statistics.
__global__ __launch_bounds__(kRThreads) void bn1d_stats_kernel(const float *__restrict__ x, int64_t B, int D,
                                                               const float *__restrict__ weight,
                                                               const float *__restrict__ bias, float momentum, float eps,
                                                               float *__restrict__ running_var,
                                                               float *__restrict__ running_mean,
                                                               float *__restrict__ mean_out, float *__restrict__ var_out,
                                                               float *__restrict__ scale_out,
                                                               float *__restrict__ shift_out,
                                                               float *__restrict__ ldj_const) {
    __shared__ float red[kRThreads / 64][kRC];
    const int c = blockIdx.x * kRC + (threadIdx.x & (kRC - 1)), rg = threadIdx.x >> 4;
    constexpr int kStep = kRThreads / kRC;
    float s = 0.f;
    if (c < D)
        for (int64_t b = rg; b < B; b += kStep) s += x[b * D + c];
    const float mean = colblock_sum(s, red) / (float)B;
    float q = 0.f;
    if (c < D)
        for (int64_t b = rg; b < B; b += kStep) {
            const float dlt = x[b * D + c] - mean;
            q = fmaf(dlt, dlt, q);
        }
    q = colblock_sum(q, red);
    float term = 0.f;
    if (rg == 0 && c < D) {
        const float var = q / (float)(B - 1);
        running_var[c] = running_var[c] * momentum + var * (1.f - momentum);
        running_mean[c] = running_mean[c] * momentum + mean * (1.f - momentum);
        mean_out[c] = mean;
        var_out[c] = var;
        const float ve = var + eps;
        const float sc = expf(weight[c]) / sqrtf(ve);
        scale_out[c] = sc;
        shift_out[c] = bias[c] - mean * sc;
        term = weight[c] - 0.5f * logf(ve);
    }
    if (threadIdx.x < 64) {   // rg == 0 lives in lanes 0..15 of wave 0
        term = wave_reduce_sum(term);
        if (threadIdx.x == 0) atomicAdd(ldj_const, term);
    }
}

// column reductions of the backward: s1[d] = sum_b g_u, s2[d] = sum_b g_u * xhat
__global__ __launch_bounds__(kRThreads) void bn1d_bwd_reduce_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ gu, int64_t B, int D,
                                                                    const float *__restrict__ mean,
                                                                    const float *__restrict__ var, float eps,
                                                                    float *__restrict__ s1, float *__restrict__ s2) {
    __shared__ float red[kRThreads / 64][kRC];
    const int c = blockIdx.x * kRC + (threadIdx.x & (kRC - 1)), rg = threadIdx.x >> 4;
    float a1 = 0.f, a2 = 0.f;
    if (c < D) {
        const float mu = mean[c], is = 1.f / sqrtf(var[c] + eps);
        for (int64_t b = rg; b < B; b += kRThreads / kRC) {
            const float g = gu[b * D + c];
            a1 += g;
            a2 = fmaf(g, (x[b * D + c] - mu) * is, a2);
        }
    }
    a1 = colblock_sum(a1, red);
    a2 = colblock_sum(a2, red);
    if (rg == 0 && c < D) {
        s1[c] = a1;
        s2[c] = a2;
    }
}

// gx and the parameter gradients.  train != 0: the statistics depend on x (unbiased variance).
//   u = xhat e^w + bias, ildj = sum_d (w_d - 0.5 log(var_d + eps))
__global__ void bn1d_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gu,
                                      const float *__restrict__ sg, int64_t B, int D, const float *__restrict__ weight,
                                      const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                      const float *__restrict__ s1, const float *__restrict__ s2, int train,
                                      float *__restrict__ gx, float *__restrict__ gw, float *__restrict__ gb) {
    const int64_t total = B * D;
    const float gsum = sg ? sg[0] : 0.f;   // sum_b g_ildj[b]
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        const float ve = var[d] + eps, is = 1.f / sqrtf(ve), ew = expf(weight[d]);
        float v = gu[e] * ew * is;
        if (train) {
            const float xhat = (x[e] - mean[d]) * is;
            // d/dvar: through u (-0.5 e^w s2 / ve) and through ildj (-0.5 gsum / ve); var = sum (x-mean)^2/(B-1)
            const float dvar = -0.5f * (ew * s2[d] + gsum) / ve;
            const float dmean = -ew * is * s1[d];
            v += dvar * 2.f * (xhat / is) / (float)(B - 1) + dmean / (float)B;
        }
        gx[e] = v;
        if (e < D) {   // first row's threads also finish the parameter gradients
            if (gw) gw[d] = ew * s2[d] + gsum;
            if (gb) gb[d] = s1[d];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Batch-sharded (synchronised) training-mode BatchNormLayer1d: every rank holds a slice of the batch, the
// statistics are those of the WHOLE batch (what the single-process reference computes, flows/utils.py:122-128).
// Ranks exchange {count, mean, M2} per column (all-gather of 2D+1 floats) and combine them in rank order with the
// pairwise update of Chan et al., so every rank derives bit-identical statistics.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRThreads) void bn1d_local_moments_kernel(const float *__restrict__ x, int64_t B, int D,
                                                                       float *__restrict__ mom) {
    __shared__ float red[kRThreads / 64][kRC];
    const int c = blockIdx.x * kRC + (threadIdx.x & (kRC - 1)), rg = threadIdx.x >> 4;
    constexpr int kStep = kRThreads / kRC;
    float s = 0.f;
    if (c < D)
        for (int64_t b = rg; b < B; b += kStep) s += x[b * D + c];
    const float mean = B > 0 ? colblock_sum(s, red) / (float)B : 0.f;
    float q = 0.f;
    if (c < D)
        for (int64_t b = rg; b < B; b += kStep) {
            const float dlt = x[b * D + c] - mean;
            q = fmaf(dlt, dlt, q);
        }
    q = colblock_sum(q, red);
    if (rg == 0 && c < D) {
        mom[1 + c] = mean;
        mom[1 + D + c] = q;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) mom[0] = (float)B;
}

__global__ void bn1d_combine_kernel(const float *__restrict__ gathered, int W, int D, const float *__restrict__ weight,
                                    const float *__restrict__ bias, float momentum, float eps,
                                    float *__restrict__ running_var, float *__restrict__ running_mean,
                                    float *__restrict__ mean_out, float *__restrict__ var_out,
                                    float *__restrict__ scale_out, float *__restrict__ shift_out,
                                    float *__restrict__ ldj_const) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (c < D) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int r = 0; r < W; ++r) {
            const float *g = gathered + (int64_t)r * (2 * D + 1);
            const float nr = g[0];
            if (nr > 0.f) {
                const float dlt = g[1 + c] - mean, nn = n + nr;
                mean += dlt * (nr / nn);
                m2 += g[1 + D + c] + dlt * dlt * (n * nr / nn);
                n = nn;
            }
        }
        const float var = m2 / (n - 1.f);
        running_var[c] = running_var[c] * momentum + var * (1.f - momentum);
        running_mean[c] = running_mean[c] * momentum + mean * (1.f - momentum);
        mean_out[c] = mean;
        var_out[c] = var;
        const float ve = var + eps;
        const float sc = expf(weight[c]) / sqrtf(ve);
        scale_out[c] = sc;
        shift_out[c] = bias[c] - mean * sc;
        term = weight[c] - 0.5f * logf(ve);
    }
    // fixed-order sum of the D log-det terms: one block, LDS tree
    __shared__ float red[1024];
    red[threadIdx.x] = term;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(ldj_const, red[0]);
}

// gx with the whole-batch column sums (sx = {s1, s2, sum g_ildj}, already scaled to this rank's loss), parameter
// gradients with this rank's own sums (sp); Bg = samples of the whole batch.
__global__ void bn1d_sync_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gu, int64_t B, int64_t Bg,
                                           int D, const float *__restrict__ weight, const float *__restrict__ mean,
                                           const float *__restrict__ var, float eps, const float *__restrict__ sx,
                                           const float *__restrict__ sp, float *__restrict__ gx, float *__restrict__ gw,
                                           float *__restrict__ gb) {
    const int64_t total = B * D;
    const float gsum = sx[2 * D];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        const float ve = var[d] + eps, is = 1.f / sqrtf(ve), ew = expf(weight[d]);
        const float xhat = (x[e] - mean[d]) * is;
        const float dvar = -0.5f * (ew * sx[D + d] + gsum) / ve;
        const float dmean = -ew * is * sx[d];
        gx[e] = gu[e] * ew * is + dvar * 2.f * (xhat / is) / (float)(Bg - 1) + dmean / (float)Bg;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < D) {
        const float ew = expf(weight[t]);
        if (gw) gw[t] = ew * sp[D + t] + sp[2 * D];
        if (gb) gb[t] = sp[t];
    }
}

__global__ void fill_kernel(float *__restrict__ p, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void vecsum_kernel(const float *__restrict__ v, int64_t n, float *__restrict__ out) {
    // one block; fp32 pairwise-ish: per-thread strided partials, LDS tree
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// d/du of sum_d log N(u; loc, scale) weighted by g[b]
__global__ void normal_base_bwd_kernel(const float *__restrict__ u, const float *__restrict__ loc,
                                       const float *__restrict__ scale, const float *__restrict__ g, int64_t B, int D,
                                       float *__restrict__ gu) {
    const int64_t total = B * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        const float sg = scale[d];
        gu[e] = -g[e / D] * (u[e] - loc[d]) / (sg * sg);
    }
}


// element-wise tail of the generic (any-depth) coupling route: Z = conditioner output [t_hat | s_hat] (affine)
// or z (NICE).  One wave per sample: writes out[b,:] and the row's log-det.
__global__ __launch_bounds__(256) void coupling_fwd_elem_kernel(const float *__restrict__ x, const float *__restrict__ Z,
                                                                const float *__restrict__ inv_mask,
                                                                const float *__restrict__ act_weight, int64_t B, int D,
                                                                int affine, int inverse, float *__restrict__ out,
                                                                float *__restrict__ ldj) {
    constexpr int U = 4;      // columns per lane in flight: all their loads are requested before the first is used
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float a = affine ? act_weight[0] : 0.f;
    float acc = 0.f;
    for (int d0 = lane; d0 < D; d0 += 64 * U) {
        float xv[U], mk[U], zt[U], zs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int dc = min(d0 + 64 * u, D - 1);
            xv[u] = x[b * D + dc];
            mk[u] = inv_mask[dc];
            zt[u] = affine ? Z[b * 2 * D + dc] : Z[b * D + dc];
            zs[u] = affine ? Z[b * 2 * D + D + dc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = d0 + 64 * u;
            if (d >= D) continue;
            const bool live = mk[u] != 0.f;
            float r;
            if (affine) {
                const float s = a * tanhf(zs[u]);
                r = inverse ? fmaf(xv[u], expf(s), zt[u]) : (xv[u] - zt[u]) * expf(-s);
                if (live) acc += s;
            } else {
                r = inverse ? xv[u] + zt[u] : xv[u] - zt[u];
            }
            out[b * D + d] = live ? r : xv[u];
        }
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) ldj[b] = inverse ? acc : -acc;
}

static inline int grid1d(int64_t total, int block = 256, int cap = 8192) {
    int64_t n = (total + block - 1) / block;
    return (int)(n < 1 ? 1 : (n > cap ? cap : n));
}

}  // namespace dpk

using namespace dpk;

// workspace: H [B,units], Z [B,zc], dH [B,units]   (zc = 2D affine, D NICE)
extern "C" int64_t dpk_coupling1d_backward_workspace_bytes(int64_t B, int32_t D, int32_t units, int32_t affine) {
    if (B < 0 || D <= 0 || units <= 0) return DPK_EINVAL;
    const int64_t zc = affine ? 2 * (int64_t)D : D;
    return 2 * align_up(B * units * 4, 256) + align_up(B * zc * 4, 256) + 256;
}

extern "C" int dpk_coupling1d_backward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                                       const float *W1, const float *b1, const float *W2, const float *b2,
                                       int32_t units, const float *act_weight, int32_t affine, const float *grad_u,
                                       const float *grad_ildj, float *grad_x, float *grad_W1, float *grad_b1,
                                       float *grad_W2, float *grad_b2, float *grad_act, void *ws, int64_t ws_bytes,
                                       void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && units > 0, DPK_EINVAL, "coupling1d_backward: bad sizes");
    DPK_REQUIRE(mask && inv_mask && W1 && b1 && W2 && b2 && ws, DPK_EINVAL, "coupling1d_backward: null pointer");
    DPK_REQUIRE(!affine || act_weight, DPK_EINVAL, "coupling1d_backward: affine coupling needs the ScaledTanh weight");
    const int64_t need = dpk_coupling1d_backward_workspace_bytes(B, D, units, affine);
    DPK_REQUIRE(ws_bytes >= need, DPK_EWORKSPACE, "coupling1d_backward: workspace %lld < %lld", (long long)ws_bytes,
                (long long)need);
    hipStream_t st = (hipStream_t)stream;
    const int zc = affine ? 2 * D : D;
    if (grad_act) DPK_REQUIRE(hipMemsetAsync(grad_act, 0, 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B == 0) {
        if (grad_W1) (void)hipMemsetAsync(grad_W1, 0, (size_t)units * D * 4, st);
        if (grad_b1) (void)hipMemsetAsync(grad_b1, 0, (size_t)units * 4, st);
        if (grad_W2) (void)hipMemsetAsync(grad_W2, 0, (size_t)zc * units * 4, st);
        if (grad_b2) (void)hipMemsetAsync(grad_b2, 0, (size_t)zc * 4, st);
        return DPK_OK;
    }
    DPK_REQUIRE(x && grad_x && (grad_u || grad_ildj), DPK_EINVAL, "coupling1d_backward: null pointer");
    char *p = (char *)ws;
    float *H = (float *)p;
    p += align_up(B * units * 4, 256);
    float *dH = (float *)p;
    p += align_up(B * units * 4, 256);
    float *Z = (float *)p;

    GemmArgs g{};
    // H = relu((mask * x) W1^T + b1)
    g = GemmArgs{};
    g.A = x; g.sam = D; g.sak = 1; g.kscale = mask;
    g.Bm = W1; g.sbk = 1; g.sbn = D;
    g.C = H; g.ldc = units; g.M = (int)B; g.N = units; g.K = D; g.bias = b1; g.relu = 1;
    launch_gemm(g, st);
    // Z = H W2^T + b2
    g = GemmArgs{};
    g.A = H; g.sam = units; g.sak = 1;
    g.Bm = W2; g.sbk = 1; g.sbn = units;
    g.C = Z; g.ldc = zc; g.M = (int)B; g.N = zc; g.K = units; g.bias = b2;
    launch_gemm(g, st);
    // element-wise core: Z <- dZ, grad_x <- direct term
    DPK_LAUNCH(coupling_bwd_elem_kernel, dim3(grid1d(B * 64, 256, 1024)), dim3(256), 0, st, x, Z, inv_mask, act_weight,
                       grad_u, grad_ildj, B, D, affine, 0, grad_x, grad_act);
    // dW2 = dZ^T H, db2 = colsum(dZ)
    if (grad_W2) {
        g = GemmArgs{};
        g.A = Z; g.sam = 1; g.sak = zc;
        g.Bm = H; g.sbk = units; g.sbn = 1;
        g.C = grad_W2; g.ldc = units; g.M = zc; g.N = units; g.K = (int)B;
        launch_gemm(g, st);
    }
    if (grad_b2) DPK_LAUNCH(colsum_kernel, dim3(cdiv(zc, kRC)), dim3(kRThreads), 0, st, Z, B, zc, (int64_t)zc, grad_b2);
    // dH = (dZ W2) * [H > 0]
    g = GemmArgs{};
    g.A = Z; g.sam = zc; g.sak = 1;
    g.Bm = W2; g.sbk = units; g.sbn = 1;
    g.C = dH; g.ldc = units; g.M = (int)B; g.N = units; g.K = zc; g.gate = H; g.ldg = units;
    launch_gemm(g, st);
    // dW1 = dH^T (mask * x), db1 = colsum(dH)
    if (grad_W1) {
        g = GemmArgs{};
        g.A = dH; g.sam = 1; g.sak = units;
        g.Bm = x; g.sbk = D; g.sbn = 1; g.nscale = mask;
        g.C = grad_W1; g.ldc = D; g.M = units; g.N = D; g.K = (int)B;
        launch_gemm(g, st);
    }
    if (grad_b1)
        DPK_LAUNCH(colsum_kernel, dim3(cdiv(units, kRC)), dim3(kRThreads), 0, st, dH, B, units, (int64_t)units, grad_b1);
    // grad_x += mask * (dH W1)
    g = GemmArgs{};
    g.A = dH; g.sam = units; g.sak = 1;
    g.Bm = W1; g.sbk = D; g.sbn = 1; g.nscale = mask;
    g.C = grad_x; g.ldc = D; g.M = (int)B; g.N = D; g.K = units; g.accumulate = 1;
    launch_gemm(g, st);
    DPK_CHECK_LAUNCH("coupling1d_backward");
    return DPK_OK;
}

// Training-mode BatchNormLayer1d.apply_backward (deeprob/flows/utils.py:118-139): batch var_mean (unbiased),
// running statistics updated in place with `momentum`, u = (x - mean)/sqrt(var + eps) * exp(weight) + bias,
// ildj_const[0] = sum_d (weight_d - 0.5 log(var_d + eps)).  save_mean / save_var [D] feed the backward.
extern "C" int dpk_bn1d_train_forward(const float *x, int64_t B, int32_t D, const float *weight, const float *bias,
                                      float *running_var, float *running_mean, float momentum, float eps, float *out,
                                      float *ildj_const, float *save_mean, float *save_var, void *ws, int64_t ws_bytes,
                                      void *stream) {
    DPK_REQUIRE(B >= 2 && D > 0, DPK_EINVAL, "bn1d_train_forward: needs at least 2 samples (unbiased variance)");
    DPK_REQUIRE(x && weight && bias && running_var && running_mean && out && ildj_const && save_mean && save_var && ws,
                DPK_EINVAL, "bn1d_train_forward: null pointer");
    DPK_REQUIRE(ws_bytes >= (int64_t)2 * D * 4, DPK_EWORKSPACE, "bn1d_train_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *sc = (float *)ws, *sh = sc + D;
    DPK_REQUIRE(hipMemsetAsync(ildj_const, 0, 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    DPK_LAUNCH(bn1d_stats_kernel, dim3(cdiv(D, kRC)), dim3(kRThreads), 0, st, x, B, D, weight, bias, momentum, eps,
                       running_var, running_mean, save_mean, save_var, sc, sh, ildj_const);
    DPK_CHECK_LAUNCH("bn1d_stats_kernel");
    return dpk_affine1d_forward(x, sc, sh, B, D, out, stream);
}

// Backward of BatchNormLayer1d.apply_backward.  train=1: mean/var are the saved batch statistics and the
// gradient flows through them; train=0: mean/var are the running statistics (constants).
// grad_ildj [B] may be NULL.  Workspace: 3*D + 1 floats.
extern "C" int dpk_bn1d_backward(const float *x, const float *grad_u, const float *grad_ildj, int64_t B, int32_t D,
                                 const float *weight, const float *mean, const float *var, float eps, int32_t train,
                                 float *grad_x, float *grad_weight, float *grad_bias, void *ws, int64_t ws_bytes,
                                 void *stream) {
    DPK_REQUIRE(B >= 1 && D > 0 && (!train || B >= 2), DPK_EINVAL, "bn1d_backward: bad sizes");
    DPK_REQUIRE(x && grad_u && weight && mean && var && grad_x && ws, DPK_EINVAL, "bn1d_backward: null pointer");
    DPK_REQUIRE(ws_bytes >= (int64_t)(2 * D + 64) * 4, DPK_EWORKSPACE, "bn1d_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *s1 = (float *)ws, *s2 = s1 + D, *sg = s2 + D;
    if (grad_ildj) DPK_LAUNCH(vecsum_kernel, dim3(1), dim3(256), 0, st, grad_ildj, B, sg);
    DPK_LAUNCH(bn1d_bwd_reduce_kernel, dim3(cdiv(D, kRC)), dim3(kRThreads), 0, st, x, grad_u, B, D, mean, var, eps, s1,
                       s2);
    DPK_LAUNCH(bn1d_bwd_apply_kernel, dim3(grid1d(B * D)), dim3(256), 0, st, x, grad_u,
                       grad_ildj ? sg : nullptr, B, D, weight, mean, var, eps, s1, s2, train, grad_x, grad_weight,
                       grad_bias);
    DPK_CHECK_LAUNCH("bn1d_backward");
    return DPK_OK;
}


// ---- batch-sharded BatchNormLayer1d (see the kernels above) ------------------------------------------------
extern "C" int dpk_bn1d_local_moments(const float *x, int64_t B, int32_t D, float *moments, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && moments && (B == 0 || x), DPK_EINVAL, "bn1d_local_moments: bad arguments");
    DPK_LAUNCH(bn1d_local_moments_kernel, dim3(cdiv(D, kRC)), dim3(kRThreads), 0, (hipStream_t)stream, x, B, D,
                       moments);
    DPK_CHECK_LAUNCH("bn1d_local_moments_kernel");
    return DPK_OK;
}

extern "C" int dpk_bn1d_sync_forward(const float *x, int64_t B, int32_t D, const float *weight, const float *bias,
                                     const float *gathered, int32_t world, float *running_var, float *running_mean,
                                     float momentum, float eps, float *out, float *ildj_const, float *save_mean,
                                     float *save_var, void *ws, int64_t ws_bytes, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && world >= 1, DPK_EINVAL, "bn1d_sync_forward: bad sizes");
    DPK_REQUIRE(weight && bias && gathered && running_var && running_mean && ildj_const && save_mean && save_var && ws &&
                    (B == 0 || (x && out)),
                DPK_EINVAL, "bn1d_sync_forward: null pointer");
    DPK_REQUIRE(ws_bytes >= (int64_t)2 * D * 4, DPK_EWORKSPACE, "bn1d_sync_forward: workspace too small");
    DPK_REQUIRE(D <= 1024 * 1024, DPK_EUNSUPPORTED, "bn1d_sync_forward: D too large");
    hipStream_t st = (hipStream_t)stream;
    float *sc = (float *)ws, *sh = sc + D;
    DPK_REQUIRE(hipMemsetAsync(ildj_const, 0, 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    DPK_LAUNCH(bn1d_combine_kernel, dim3(cdiv(D, 1024)), dim3(1024), 0, st, gathered, world, D, weight, bias,
                       momentum, eps, running_var, running_mean, save_mean, save_var, sc, sh, ildj_const);
    DPK_CHECK_LAUNCH("bn1d_combine_kernel");
    if (B == 0) return DPK_OK;
    return dpk_affine1d_forward(x, sc, sh, B, D, out, stream);
}

extern "C" int dpk_bn1d_backward_sums(const float *x, const float *grad_u, const float *grad_ildj, int64_t B, int32_t D,
                                      const float *mean, const float *var, float eps, float *sums, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && mean && var && sums && (B == 0 || (x && grad_u)), DPK_EINVAL,
                "bn1d_backward_sums: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || !grad_ildj) DPK_REQUIRE(hipMemsetAsync(sums + 2 * D, 0, 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B > 0 && grad_ildj) DPK_LAUNCH(vecsum_kernel, dim3(1), dim3(256), 0, st, grad_ildj, B, sums + 2 * D);
    DPK_LAUNCH(bn1d_bwd_reduce_kernel, dim3(cdiv(D, kRC)), dim3(kRThreads), 0, st, x, grad_u, B, D, mean, var, eps,
                       sums, sums + D);
    DPK_CHECK_LAUNCH("bn1d_backward_sums");
    return DPK_OK;
}

extern "C" int dpk_bn1d_sync_backward(const float *x, const float *grad_u, int64_t B, int64_t B_total, int32_t D,
                                      const float *weight, const float *mean, const float *var, float eps,
                                      const float *sums_x, const float *sums_p, float *grad_x, float *grad_weight,
                                      float *grad_bias, void *stream) {
    DPK_REQUIRE(B >= 0 && B_total >= 2 && D > 0 && weight && mean && var && sums_x && sums_p &&
                    (B == 0 || (x && grad_u && grad_x)),
                DPK_EINVAL, "bn1d_sync_backward: bad arguments");
    const int64_t work = B * D > D ? B * D : D;
    DPK_LAUNCH(bn1d_sync_bwd_apply_kernel, dim3(grid1d(work)), dim3(256), 0, (hipStream_t)stream, x, grad_u, B,
                       B_total, D, weight, mean, var, eps, sums_x, sums_p, grad_x, grad_weight, grad_bias);
    DPK_CHECK_LAUNCH("bn1d_sync_bwd_apply_kernel");
    return DPK_OK;
}

// Backward of the eval-statistics inverse BatchNormLayer1d.apply_forward (flows/utils.py:141-153):
//   x = (u - bias) G + mean,  G = exp(-weight) sqrt(var + eps),  ldj = sum_d (-weight + 0.5 log(var + eps))
//   grad_u = g G ;  grad_weight = -G sum_b g (u - bias) - sum_b g_ldj ;  grad_bias = -G sum_b g
__global__ void bn1d_inverse_bwd_finish_kernel(const float *__restrict__ weight, const float *__restrict__ var, float eps,
                                               int D, const float *__restrict__ s1, const float *__restrict__ s2,
                                               const float *__restrict__ sg, float *__restrict__ G,
                                               float *__restrict__ gw, float *__restrict__ gb) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float g = expf(-weight[d]) * sqrtf(var[d] + eps);
    G[d] = g;
    if (gw) gw[d] = -g * s2[d] - (sg ? sg[0] : 0.f);
    if (gb) gb[d] = -g * s1[d];
}

extern "C" int dpk_bn1d_inverse_backward(const float *u, const float *grad_x, const float *grad_ldj, int64_t B, int32_t D,
                                         const float *weight, const float *bias, const float *running_var, float eps,
                                         float *grad_u, float *grad_weight, float *grad_bias, void *ws, int64_t ws_bytes,
                                         void *stream) {
    DPK_REQUIRE(B >= 1 && D > 0, DPK_EINVAL, "bn1d_inverse_backward: bad sizes");
    DPK_REQUIRE(u && grad_x && weight && bias && running_var && grad_u && ws, DPK_EINVAL,
                "bn1d_inverse_backward: null pointer");
    DPK_REQUIRE(ws_bytes >= (int64_t)(5 * D + 64) * 4, DPK_EWORKSPACE, "bn1d_inverse_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *s1 = (float *)ws, *s2 = s1 + D, *G = s2 + D, *ones = G + D, *zero = ones + D, *sg = zero + D;
    // column sums of g and g (u - bias): the reducer of the forward direction with mean := bias, 1 / sqrt(var + eps) := 1
    DPK_REQUIRE(hipMemsetAsync(zero, 0, (size_t)D * 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    DPK_LAUNCH(fill_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ones, 1.0f, (int64_t)D);
    if (grad_ldj) DPK_LAUNCH(vecsum_kernel, dim3(1), dim3(256), 0, st, grad_ldj, B, sg);
    DPK_LAUNCH(bn1d_bwd_reduce_kernel, dim3(cdiv(D, kRC)), dim3(kRThreads), 0, st, u, grad_x, B, D, bias, ones, 0.f,
                       s1, s2);
    DPK_LAUNCH(bn1d_inverse_bwd_finish_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, weight, running_var, eps, D,
                       s1, s2, grad_ldj ? sg : nullptr, G, grad_weight, grad_bias);
    DPK_CHECK_LAUNCH("bn1d_inverse_backward");
    return dpk_affine1d_forward(grad_x, G, zero, B, D, grad_u, stream);
}

// d/du of dpk_normal_base_logprob (no incoming affine): grad_u[b,d] = -g[b] (u - loc)/scale^2
extern "C" int dpk_normal_base_backward(const float *u, const float *loc, const float *scale, const float *g, int64_t B,
                                        int32_t D, float *grad_u, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0, DPK_EINVAL, "normal_base_backward: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(u && loc && scale && g && grad_u, DPK_EINVAL, "normal_base_backward: null pointer");
    DPK_LAUNCH(normal_base_bwd_kernel, dim3(grid1d(B * D)), dim3(256), 0, (hipStream_t)stream, u, loc, scale, g,
                       B, D, grad_u);
    DPK_CHECK_LAUNCH("normal_base_bwd_kernel");
    return DPK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Generic conditioner depth (CouplingLayer1d(depth = n_hidden), flows/layers/coupling.py:45-56): the MLP is
// chained through the generic GEMM kernel, layer by layer.  W[i] [widths[i], in_i], b[i] [widths[i]] for
// i = 0..n_hidden (host arrays of device pointers); widths[n_hidden] = 2D (affine) or D.
// ---------------------------------------------------------------------------------------------------------
struct MlpWs {
    float *H[9];   // hidden activations, H[i] [B, widths[i]]
    float *Z;      // [B, zc]
    float *dA, *dB;
    int64_t bytes;
};
static MlpWs carve_mlp_ws(void *base, int64_t B, int n_hidden, const int32_t *widths, bool backward) {
    MlpWs w{};
    int64_t o = 0;
    auto take = [&](int64_t n_bytes) {
        char *p = base ? (char *)base + o : nullptr;
        o = align_up(o + n_bytes, 256);
        return (float *)p;
    };
    int maxw = 0;
    for (int i = 0; i < n_hidden; ++i) {
        w.H[i] = take(B * widths[i] * 4);
        maxw = widths[i] > maxw ? widths[i] : maxw;
    }
    w.Z = take(B * widths[n_hidden] * 4);
    if (backward) {
        w.dA = take(B * maxw * 4);
        w.dB = take(B * maxw * 4);
    }
    w.bytes = o + 256;
    return w;
}

static int mlp_check(int64_t B, int D, int n_hidden, const float *const *W, const float *const *b,
                     const int32_t *widths, int affine, const char *who) {
    DPK_REQUIRE(B >= 0 && D > 0 && n_hidden >= 1 && n_hidden <= 8, DPK_EINVAL, "%s: bad sizes (1..8 hidden layers)", who);
    DPK_REQUIRE(W && b && widths, DPK_EINVAL, "%s: null pointer", who);
    for (int i = 0; i <= n_hidden; ++i) {
        DPK_REQUIRE(W[i] && b[i] && widths[i] > 0, DPK_EINVAL, "%s: layer %d missing", who, i);
    }
    DPK_REQUIRE(widths[n_hidden] == (affine ? 2 * D : D), DPK_EINVAL, "%s: last layer must have %d outputs", who,
                affine ? 2 * D : D);
    return DPK_OK;
}

// forward of the MLP into ws (H[i], Z)
static void mlp_forward(const MlpWs &w, const float *x, int64_t B, int D, const float *mask, int n_hidden,
                        const float *const *W, const float *const *b, const int32_t *widths, hipStream_t st) {
    const float *in = x;
    int in_w = D;
    for (int i = 0; i <= n_hidden; ++i) {
        GemmArgs g{};
        g.A = in; g.sam = in_w; g.sak = 1; g.kscale = (i == 0) ? mask : nullptr;
        g.Bm = W[i]; g.sbk = 1; g.sbn = in_w;
        g.C = (i < n_hidden) ? w.H[i] : w.Z; g.ldc = widths[i];
        g.M = (int)B; g.N = widths[i]; g.K = in_w; g.bias = b[i]; g.relu = (i < n_hidden);
        launch_gemm(g, st);
        in = g.C;
        in_w = widths[i];
    }
}

extern "C" int64_t dpk_coupling1d_mlp_workspace_bytes(int64_t B, int32_t n_hidden, const int32_t *widths,
                                                      int32_t backward) {
    if (B < 0 || n_hidden < 1 || n_hidden > 8 || !widths) return DPK_EINVAL;
    return carve_mlp_ws(nullptr, B, n_hidden, widths, backward != 0).bytes;
}

extern "C" int dpk_coupling1d_mlp_forward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                                          int32_t n_hidden, const float *const *W, const float *const *b,
                                          const int32_t *widths, const float *act_weight, int32_t affine,
                                          int32_t inverse, float *out, float *ldj, void *ws, int64_t ws_bytes,
                                          void *stream) {
    int rc = mlp_check(B, D, n_hidden, W, b, widths, affine, "coupling1d_mlp_forward");
    if (rc) return rc;
    DPK_REQUIRE(mask && inv_mask && ws, DPK_EINVAL, "coupling1d_mlp_forward: null pointer");
    DPK_REQUIRE(!affine || act_weight, DPK_EINVAL, "coupling1d_mlp_forward: affine coupling needs the ScaledTanh weight");
    MlpWs w = carve_mlp_ws(ws, B, n_hidden, widths, false);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "coupling1d_mlp_forward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out && ldj, DPK_EINVAL, "coupling1d_mlp_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    mlp_forward(w, x, B, D, mask, n_hidden, W, b, widths, st);
    DPK_LAUNCH(coupling_fwd_elem_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, w.Z, inv_mask, act_weight, B, D,
                       affine, inverse, out, ldj);
    DPK_CHECK_LAUNCH("coupling1d_mlp_forward");
    return DPK_OK;
}

// grad_W[i] / grad_b[i] may be NULL pointers inside the arrays (or the arrays themselves NULL).
static int mlp_backward_dir(const float *x, int64_t B, int32_t D, const float *mask,
                                           const float *inv_mask, int32_t n_hidden, const float *const *W,
                                           const float *const *b, const int32_t *widths, const float *act_weight,
                                           int32_t affine, int32_t inverse, const float *grad_u, const float *grad_ildj, float *grad_x,
                                           float *const *grad_W, float *const *grad_b, float *grad_act,
                                           int32_t ws_holds_forward, void *ws, int64_t ws_bytes, void *stream) {
    int rc = mlp_check(B, D, n_hidden, W, b, widths, affine, "coupling1d_mlp_backward");
    if (rc) return rc;
    DPK_REQUIRE(mask && inv_mask && ws, DPK_EINVAL, "coupling1d_mlp_backward: null pointer");
    DPK_REQUIRE(!affine || act_weight, DPK_EINVAL, "coupling1d_mlp_backward: affine coupling needs the ScaledTanh weight");
    MlpWs w = carve_mlp_ws(ws, B, n_hidden, widths, true);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "coupling1d_mlp_backward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (grad_act) DPK_REQUIRE(hipMemsetAsync(grad_act, 0, 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B == 0) {
        int in_w = D;
        for (int i = 0; i <= n_hidden; ++i) {
            if (grad_W && grad_W[i]) (void)hipMemsetAsync(grad_W[i], 0, (size_t)widths[i] * in_w * 4, st);
            if (grad_b && grad_b[i]) (void)hipMemsetAsync(grad_b[i], 0, (size_t)widths[i] * 4, st);
            in_w = widths[i];
        }
        return DPK_OK;
    }
    DPK_REQUIRE(x && grad_x && (grad_u || grad_ildj), DPK_EINVAL, "coupling1d_mlp_backward: null pointer");
    if (!ws_holds_forward) mlp_forward(w, x, B, D, mask, n_hidden, W, b, widths, st);
    DPK_LAUNCH(coupling_bwd_elem_kernel, dim3(grid1d(B * 64, 256, 1024)), dim3(256), 0, st, x, w.Z, inv_mask, act_weight,
                       grad_u, grad_ildj, B, D, affine, inverse, grad_x, grad_act);
    // back through the layers: dOut starts as dZ (in w.Z)
    float *dout = w.Z;
    float *spare[2] = {w.dA, w.dB};
    for (int i = n_hidden; i >= 0; --i) {
        const int ow = widths[i];
        const int in_w = (i == 0) ? D : widths[i - 1];
        const float *in = (i == 0) ? x : w.H[i - 1];
        GemmArgs g{};
        if (grad_W && grad_W[i]) {   // dW = dOut^T In  (In = mask * x for the first layer)
            g = GemmArgs{};
            g.A = dout; g.sam = 1; g.sak = ow;
            g.Bm = in; g.sbk = in_w; g.sbn = 1; g.nscale = (i == 0) ? mask : nullptr;
            g.C = grad_W[i]; g.ldc = in_w; g.M = ow; g.N = in_w; g.K = (int)B;
            launch_gemm(g, st);
        }
        if (grad_b && grad_b[i])
            DPK_LAUNCH(colsum_kernel, dim3(cdiv(ow, kRC)), dim3(kRThreads), 0, st, dout, B, ow, (int64_t)ow, grad_b[i]);
        g = GemmArgs{};
        g.A = dout; g.sam = ow; g.sak = 1;
        g.Bm = W[i]; g.sbk = in_w; g.sbn = 1;
        g.M = (int)B; g.N = in_w; g.K = ow;
        if (i == 0) {   // grad_x += mask * (dOut W0)
            g.C = grad_x; g.ldc = D; g.nscale = mask; g.accumulate = 1;
            launch_gemm(g, st);
        } else {        // dIn = (dOut W_i) * [H_{i-1} > 0]
            float *dst = spare[i & 1];
            g.C = dst; g.ldc = in_w; g.gate = w.H[i - 1]; g.ldg = in_w;
            launch_gemm(g, st);
            dout = dst;
        }
    }
    DPK_CHECK_LAUNCH("coupling1d_mlp_backward");
    return DPK_OK;
}

extern "C" int dpk_coupling1d_mlp_backward(const float *x, int64_t B, int32_t D, const float *mask,
                                           const float *inv_mask, int32_t n_hidden, const float *const *W,
                                           const float *const *b, const int32_t *widths, const float *act_weight,
                                           int32_t affine, const float *grad_u, const float *grad_ildj, float *grad_x,
                                           float *const *grad_W, float *const *grad_b, float *grad_act,
                                           int32_t ws_holds_forward, void *ws, int64_t ws_bytes, void *stream) {
    return mlp_backward_dir(x, B, D, mask, inv_mask, n_hidden, W, b, widths, act_weight, affine, 0, grad_u, grad_ildj,
                            grad_x, grad_W, grad_b, grad_act, ws_holds_forward, ws, ws_bytes, stream);
}

// Backward of the SAMPLING direction (CouplingLayer1d.apply_forward, coupling.py:89-104; what NormalizingFlow.rsample
// differentiates, flows/models/base.py:159-180): x = the input u of apply_forward, grad_u = gradient w.r.t. its
// output, grad_ildj = gradient w.r.t. its log-det.  Everything else as dpk_coupling1d_mlp_backward.
extern "C" int dpk_coupling1d_mlp_backward_inverse(const float *x, int64_t B, int32_t D, const float *mask,
                                                   const float *inv_mask, int32_t n_hidden, const float *const *W,
                                                   const float *const *b, const int32_t *widths,
                                                   const float *act_weight, int32_t affine, const float *grad_out,
                                                   const float *grad_ldj, float *grad_x, float *const *grad_W,
                                                   float *const *grad_b, float *grad_act, int32_t ws_holds_forward,
                                                   void *ws, int64_t ws_bytes, void *stream) {
    return mlp_backward_dir(x, B, D, mask, inv_mask, n_hidden, W, b, widths, act_weight, affine, 1, grad_out, grad_ldj,
                            grad_x, grad_W, grad_b, grad_act, ws_holds_forward, ws, ws_bytes, stream);
}
