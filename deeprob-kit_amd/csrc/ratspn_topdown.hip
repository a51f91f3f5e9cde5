// Top-down pass of a RAT-SPN in ONE launch: RatSpn.mpe and RatSpn.sample.
//
// reference: deeprob/spn/models/ratspn.py:124-162 (mpe) and :164-182 (sample) -- a python loop over the layers, root to
// leaves, each layer a handful of index operations on [B, groups] tensors: RootLayer.mpe / .sample
// (deeprob/spn/layers/ratspn.py:460-474 / :476-490), SumLayer.mpe / .sample (:380-399 / :401-417), ProductLayer.mpe / .sample
// (:288-304 / :306-330) and RegionGraphLayer.mpe / .sample with unpad_samples (:118-136 / :138-157, :68-85).
//
// Here a wave walks one sample's induced tree.  What it needs from the bottom-up pass are the leaf layer's and the sum
// layers' outputs only (the product tensors [B, P, N^2] the reference stores are re-formed on the fly as
// fl(in[2p, i] + in[2p+1, j]) -- the same fp32 value); the argmax adds the log-softmax weight in the reference's order,
// fl(fl(a + c) + lw), and resolves ties to the smallest index like torch.argmax on the reference's CPU path.  The choice
// of a node is kept per region of the chosen repetition in LDS (2^depth ints at the leaves), the product layers are index
// arithmetic (offset -> offset / N, offset % N), and the leaf step writes x[b, f] for every variable straight in variable
// order through `src` (= the reference's inv_mask with the dummy variables dropped, unpad_samples).
//
// Sampling draws are counter based (the library's splitmix64 hash, common.h: dropout_hit uses the same), so that a test can
// replay them: u(ctr) = (splitmix64(seed + ctr * golden) >> 40) / 2^24 with ctr = b * K + slot, K = 2^depth + 2 D;
// slot 1 = the root's choice, slot G + g = the choice of region g (of G) of a sum level, slots 2^depth + 2 f, + 1 = the two
// uniforms of variable f's leaf (Box-Muller: z = sqrt(-2 log(1 - u1)) cos(2 pi u2); Bernoulli: u1 < p).  A categorical
// choice is the inverse CDF over exp(log-softmax weights) in index order.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace dpk {

constexpr int kTdMaxDepth = 10;      // 2^depth <= in_features (RegionGraph): 784 variables allow depth 9

struct TopDownArgs {
    int mode, dist;                  // 0 = mpe, 1 = sample; 0 = Gaussian, 1 = Bernoulli leaves
    int64_t B;
    int D, depth, reps, I, S, C, d;
    const float *x;                  // [B, D] evidence, NaN = to be completed; null: nothing observed
    const int64_t *y;                // [B] class of the root to descend from; null: class 0
    const float *act[kTdMaxDepth];   // [0] leaf layer output [B, reps 2^depth, I]; [t] sum level t output [B, reps 2^(depth-t), S]  (mpe)
    const float *logw[kTdMaxDepth + 1];   // [t], 1 <= t < depth: log_softmax of sum level t's weight [P_t, S, N^2]; [depth]: root [C, reps N^2]
    const int *src;                  // [reps, D]: position (region within the repetition) * d + j holding variable f
    const float *p0, *p1;            // leaf parameters [reps 2^depth, I, d]: loc, scale / logits, null
    unsigned long long seed;
    float *out;                      // [B, D]
    int *choice;                     // optional [B, 1 + 2^depth]: repetition, then the leaf channel chosen per region
};

__device__ __forceinline__ float td_uniform(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed + ctr * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

// first maximum over the wave: (v, n) pairs, larger v wins, equal v -> smaller n
__device__ __forceinline__ void wave_argmax(float &v, int &n) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int on = __shfl_xor(n, o, 64);
        const bool take = (ov > v) || (ov == v && on < n);
        v = take ? ov : v;
        n = take ? on : n;
    }
}

// One node's choice among `count` inputs: value(n) = its score (mpe) / its log-weight (sample).  Returns the chosen n in
// every lane.
template <typename ScoreFn, typename LogwFn>
__device__ __forceinline__ int td_choose(int mode, int count, int lane, float u, ScoreFn score, LogwFn logw) {
    if (mode == 0) {
        float bv = -INFINITY;
        int bn = 0x7fffffff;
        for (int n = lane; n < count; n += 64) {
            const float v = score(n);
            if (bn == 0x7fffffff || v > bv) {
                bv = v;
                bn = n;
            }
        }
        wave_argmax(bv, bn);
        return bn;
    }
    // inverse CDF in index order: lane l owns the contiguous chunk [l ch, (l + 1) ch)
    const int ch = (count + 63) / 64;
    const int n0 = min(lane * ch, count), n1 = min(n0 + ch, count);
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += expf(logw(n));
    float inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const float total = __shfl(inc, 63, 64);
    const float target = u * total;
    const unsigned long long hit = __ballot(target < inc && n1 > n0);
    int pick = count - 1;                          // (target >= total by rounding: the last input)
    if (hit != 0ull) {
        const int owner = __ffsll((long long)hit) - 1;
        int mine = n1 - 1;
        if (lane == owner) {
            float c = inc - s;
            for (int n = n0; n < n1; ++n) {
                c += expf(logw(n));
                if (target < c) {
                    mine = n;
                    break;
                }
            }
        }
        pick = __shfl(mine, owner, 64);
    }
    return pick;
}

__global__ __launch_bounds__(256) void ratspn_topdown_kernel(const TopDownArgs a) {
    extern __shared__ int td_nodes[];               // [wave][2][2^depth]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G0 = 1 << a.depth;                    // leaf regions per repetition
    int *cur = td_nodes + (size_t)wave * 2 * G0, *nxt = cur + G0;
    const unsigned long long K = (unsigned long long)G0 + 2ull * (unsigned long long)a.D;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < a.B; b += (int64_t)gridDim.x * 4) {
        const unsigned long long ctr0 = (unsigned long long)b * K;
        // (a label outside [0, C) indexes past the root's weight rows in the reference and raises there; here it is clamped:
        // a kernel cannot raise, and must not read past the table)
        const int cls = a.y ? min(max((int)a.y[b], 0), a.C - 1) : 0;
        // ---- root: one of the reps * N^2 inputs (partition = repetition, (i, j) = nodes of its two regions) ----
        int rep;
        {
            const int t = a.depth - 1;                                  // the level the root consumes
            const int N = t == 0 ? a.I : a.S, NN = N * N, R = 2 * a.reps;
            const float *A = a.act[t] ? a.act[t] + b * (int64_t)R * N : nullptr;
            const float *lw = a.logw[a.depth] + (int64_t)cls * a.reps * NN;
            const int n = td_choose(
                a.mode, a.reps * NN, lane, a.mode ? td_uniform(a.seed, ctr0 + 1) : 0.f,
                [&](int q) {
                    const int p = q / NN, e = q - p * NN, i = e / N, j = e - i * N;
                    return (A[(2 * p) * N + i] + A[(2 * p + 1) * N + j]) + lw[q];
                },
                [&](int q) { return lw[q]; });
            rep = n / NN;
            const int e = n - rep * NN;
            if (lane == 0) {
                cur[0] = e / N;
                cur[1] = e - (e / N) * N;
            }
        }
        // ---- sum levels, top to bottom: level t has G = 2^(depth - t) regions in the repetition ----
        for (int t = a.depth - 1; t >= 1; --t) {
            const int G = 1 << (a.depth - t);
            const int N = t == 1 ? a.I : a.S, NN = N * N;               // nodes of the level below = inputs per child region
            const int Rb = 2 * G * a.reps;                              // regions of the level below
            const float *A = a.act[t - 1] ? a.act[t - 1] + b * (int64_t)Rb * N : nullptr;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int gl = 0; gl < G; ++gl) {
                const int g = rep * G + gl, o = cur[gl];
                const float *lw = a.logw[t] + ((int64_t)g * a.S + o) * NN;
                const int n = td_choose(
                    a.mode, NN, lane, a.mode ? td_uniform(a.seed, ctr0 + (unsigned long long)(G + gl)) : 0.f,
                    [&](int q) {
                        const int i = q / N, j = q - i * N;
                        return (A[(2 * g) * N + i] + A[(2 * g + 1) * N + j]) + lw[q];
                    },
                    [&](int q) { return lw[q]; });
                if (lane == 0) {
                    nxt[2 * gl] = n / N;
                    nxt[2 * gl + 1] = n - (n / N) * N;
                }
            }
            int *sw = cur;
            cur = nxt;
            nxt = sw;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- leaves: every variable from the chosen channel of its region, in variable order ----
        if (a.choice != nullptr) {
            int *c = a.choice + b * (int64_t)(1 + G0);
            if (lane == 0) c[0] = rep;
            for (int q = lane; q < G0; q += 64) c[1 + q] = cur[q];
        }
        const int *src = a.src + (int64_t)rep * a.D;
        for (int f = lane; f < a.D; f += 64) {
            const float xv = a.x ? a.x[b * a.D + f] : NAN;
            float v = xv;
            if (xv != xv) {
                const int s = src[f], rl = s / a.d, j = s - rl * a.d;
                const int64_t po = ((int64_t)(rep * G0 + rl) * a.I + cur[rl]) * a.d + j;
                const float q0 = a.p0[po];
                if (a.mode == 0) {
                    // the mode: Normal -> loc; Bernoulli -> [sigmoid(logit) >= 0.5] (ratspn.py:130, distribution means)
                    v = a.dist == 0 ? q0 : ((1.f / (1.f + expf(-q0))) >= 0.5f ? 1.f : 0.f);
                } else {
                    const float u1 = td_uniform(a.seed, ctr0 + (unsigned long long)G0 + 2ull * (unsigned long long)f);
                    if (a.dist == 0) {
                        const float u2 = td_uniform(a.seed, ctr0 + (unsigned long long)G0 + 2ull * (unsigned long long)f + 1ull);
                        const float z = sqrtf(-2.f * logf(1.f - u1)) * cosf(6.28318530717958647692f * u2);
                        v = fmaf(a.p1[po], z, q0);
                    } else {
                        v = u1 < (1.f / (1.f + expf(-q0))) ? 1.f : 0.f;
                    }
                }
            }
            a.out[b * a.D + f] = v;
        }
        // (cur / nxt are rewritten by lane 0 only after every lane has passed the reads above: same wave, program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

}  // namespace dpk

using namespace dpk;

extern "C" int dpk_ratspn_topdown(int32_t mode, int32_t dist, int64_t B, int32_t D, int32_t depth, int32_t reps, int32_t I,
                                  int32_t S, int32_t C, int32_t d, const float *x, const int64_t *y,
                                  const float *const *act, const float *const *logw, const int32_t *src,
                                  const float *p0, const float *p1, uint64_t seed, float *out, int32_t *choice,
                                  void *stream) {
    DPK_REQUIRE(mode == 0 || mode == 1, DPK_EINVAL, "ratspn_topdown: mode %d", mode);
    DPK_REQUIRE(dist == 0 || dist == 1, DPK_EINVAL, "ratspn_topdown: dist %d", dist);
    DPK_REQUIRE(B >= 0 && D > 0 && depth >= 1 && depth <= kTdMaxDepth && reps > 0 && I > 0 && S > 0 && C > 0 && d > 0,
                DPK_EINVAL, "ratspn_topdown: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(logw && src && p0 && out && (dist == 1 || mode == 0 || p1), DPK_EINVAL, "ratspn_topdown: null pointer");
    DPK_REQUIRE(mode == 1 || act, DPK_EINVAL, "ratspn_topdown: mpe needs the bottom-up activations");
    TopDownArgs a{};
    a.mode = mode; a.dist = dist; a.B = B; a.D = D; a.depth = depth; a.reps = reps; a.I = I; a.S = S; a.C = C; a.d = d;
    a.x = x; a.y = y; a.src = src; a.p0 = p0; a.p1 = p1; a.seed = seed; a.out = out; a.choice = choice;
    for (int t = 0; t < depth; ++t) {
        a.act[t] = (mode == 0) ? act[t] : nullptr;
        DPK_REQUIRE(mode == 1 || a.act[t], DPK_EINVAL, "ratspn_topdown: activations of level %d missing", t);
    }
    for (int t = 1; t <= depth; ++t) {
        a.logw[t] = logw[t];
        DPK_REQUIRE(a.logw[t], DPK_EINVAL, "ratspn_topdown: log-weights of level %d missing", t);
    }
    const size_t lds = (size_t)4 * 2 * ((size_t)1 << depth) * sizeof(int);
    const int64_t groups = (B + 3) / 4;
    const unsigned grid = (unsigned)std::min<int64_t>(groups, (int64_t)device_cus() * 16);
    DPK_LAUNCH(ratspn_topdown_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    DPK_CHECK_LAUNCH("ratspn_topdown_kernel");
    return DPK_OK;
}
