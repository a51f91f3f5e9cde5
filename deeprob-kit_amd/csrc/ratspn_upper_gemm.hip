// ProductLayer folded into the SumLayer / RootLayer above it, on the f16 matrix cores (8 or 16 nodes per region).
//
// reference: ProductLayer.forward (deeprob/spn/layers/ratspn.py:272-286) + SumLayer.forward (:363-378) /
//            RootLayer.forward (:446-458); eval route of models wider than the fused kernels (rg_batch = rg_sum = 16 of
//            examples/ratspn_mnist.py, the 8-channel default of experiments/ratspn.py below 32k samples)
//
//   out[b,p,o] = logsumexp_ij(a_i + c_j + log_softmax(W)[p,o,i,j]),  a = in[b,2p,:], c = in[b,2p+1,:]
//              = max a + max c + log sum_i ea_i (sum_j softmax(W)[p,o,i,j] ec_j),   ea = e^{a - max a}, ec likewise
//
// The inner sum is a GEMM per partition: T[(o,i), b] = sum_j Wp[(o,i), j] ec[j, b] with K = N (one K-step of 16; N = 8
// uses half of it).  It runs as three v_mfma_f32_32x32x16_f16 on two-way f16 splits of both operands (weights and
// exponentials are in [0, 1]; >= 22 significant bits per product, fp32 accumulation) instead of S N^2 fp32 FMAs per
// sample and partition on the VALU: 135k flop per sample at N = S = 16, the layer that bounded that model.  The row
// order (o, i) is chosen so that lane (sample l & 31, half h = l >> 5) receives all N rows i of the outputs o it
// owns: the outer sum over i is a dot product with the lane's own ea, in registers.  A wave owns 32 samples; weight
// fragments (256 KB at N = S = 16, L2 resident) are read straight from global memory, no LDS.  A node whose scaled
// sum falls below 1e-8 (dominant pair under a small weight: the split's absolute error would show) is redone in the
// exact two-pass log domain.
#include "common.h"
#include "ratspn_gemm_common.h"
#include <math.h>

namespace dpk {

typedef const __attribute__((address_space(1))) half8 ug_h8;
typedef const __attribute__((address_space(1))) gf32x4 ug_f4;

// Both MFMA operands live in [0, 1] (softmax weights, exponentials scaled to their maximum).  An f16 split keeps 22
// significant bits only while the LOW half is a normal f16 number: unscaled, a weight of 1/64 would be carried to 5e-6
// relative, and the error is ABSOLUTE (3e-8, half a subnormal step) against a sum v that can itself be as small as the
// dominant pair's weight.  Both operands are therefore scaled by 2^15 before the split (exact; <= 32768, inside the f16
// range) and the accumulated sum by 2^-30 afterwards: absolute error 1e-12 per operand.  A node whose sum falls below
// kUpExactBelow (where that absolute error would reach 1e-4 relative) is redone in the exact log domain.
constexpr float kUpScale = 32768.f, kUpScaleLog2 = 15.f, kUpUnscale = 1.f / (32768.f * 32768.f), kUpExactBelow = 1e-8f;

// A-fragment tables from the linear softmax weights
//   sum layer : Wl [P][S][N*N]      -> frag [P][S*N/32 tiles][2][64][8]
//   root layer: Wl [C][P][N*N]      -> frag [P][ceil(C*N/32) tiles][2][64][8]
__global__ __launch_bounds__(256) void upper_pack_kernel(const float *__restrict__ Wl, int P, int N, int S, int root,
                                                         int tiles, uint16_t *__restrict__ frag, const unsigned *gate) {
    if (gate_closed(gate)) return;   // (tables still match the live weights: common.h params_gate)
    const int64_t total = (int64_t)P * tiles * 64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const int t = (int)((e >> 6) % tiles), p = (int)((e >> 6) / tiles);
        const int row = l & 31, hg = l >> 5;
        int o, i;
        bool ok = true;
        if (root) {   // natural order: global row = class * N + i
            const int g = 32 * t + row;
            o = g / N;
            i = g - o * N;
            ok = o < S;
        } else if (N == 16) {   // lane half h of the C layout gets o = 2 t + h, i = its register index
            const int hrow = (row >> 2) & 1, u = (row & 3) + 4 * (row >> 3);
            o = 2 * t + hrow;
            i = u;
        } else {                // N == 8: o = 4 t + 2 h + (u >> 3), i = u & 7
            const int hrow = (row >> 2) & 1, u = (row & 3) + 4 * (row >> 3);
            o = 4 * t + 2 * hrow + (u >> 3);
            i = u & 7;
        }
        half8 vh, vl;
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            const int j = 8 * hg + el;
            float v = 0.f;
            if (ok && j < N) {
                const int64_t src = root ? (((int64_t)o * P + p) * N + i) * N + j : (((int64_t)p * S + o) * N + i) * N + j;
                v = Wl[src] * kUpScale;
            }
            _Float16 hi, lo;
            split_f16(v, hi, lo);
            vh[el] = hi; vl[el] = lo;
        }
        uint16_t *dst = frag + ((int64_t)p * tiles + t) * 1024 + l * 8;
        *reinterpret_cast<half8 *>(dst) = vh;
        *reinterpret_cast<half8 *>(dst + 512) = vl;
    }
}

template <int N>
__device__ __forceinline__ void upper_load(const float *xa, float (&a)[N], float (&c)[N]) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const gf32x4 va = *(ug_f4 *)(xa + 4 * q), vc = *(ug_f4 *)(xa + N + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[4 * q + i] = va[i];
            c[4 * q + i] = vc[i];
        }
    }
}

template <int N> __device__ __forceinline__ float upper_max(const float (&x)[N]) {
    float m = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = fmaxf(m, x[i]);
    return (m == -INFINITY) ? 0.f : m;
}

// exact log-domain (m, s) of logsumexp_ij(a_i + c_j + lw[i*N + j])
template <int N>
__device__ __noinline__ void upper_exact_ms(const float *xa, const float *lw, float &m, float &sum) {
    m = -INFINITY;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) m = fmaxf(m, xa[i] + xa[N + j] + lw[i * N + j]);
    sum = 0.f;
    if (m > -INFINITY)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) sum += expf(xa[i] + xa[N + j] + lw[i * N + j] - m);
}

struct UpperArgs {
    const float *in;          // [B, R, N]
    const uint16_t *frag;
    const float *LW;          // log-softmax weights (exact fallback)
    float *out;
    int64_t B;
    int R, S, tiles, ppb;     // S = outputs per partition (sum) / classes (root); ppb = partitions per block (sum)
};

constexpr float kUpLn2 = 0.6931471805599453f, kUpL2E = 1.4426950408889634f;

// ---- sum layer ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void prodsum_mfma_kernel(const UpperArgs a) {
    constexpr int OPT = 32 / N;                 // outputs per tile: 2 (N = 16) or 4 (N = 8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int P = a.R / 2, S = a.S;
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + s;
    const bool row_ok = b_raw < a.B;
    const int64_t b = row_ok ? b_raw : a.B - 1;
    const int p0 = blockIdx.y * a.ppb, p1 = min(P, p0 + a.ppb);
    for (int p = p0; p < p1; ++p) {
        const float *xa = a.in + (b * a.R + 2 * p) * N;
        float av[N], cv[N];
        upper_load<N>(xa, av, cv);
        const float ma = upper_max<N>(av), mc = upper_max<N>(cv);
        float ea[N];
#pragma unroll
        for (int i = 0; i < N; ++i) ea[i] = __builtin_amdgcn_exp2f((av[i] - ma) * kUpL2E);
        // B fragment: K slot (hg = h, el) <-> j = 8 h + el (N = 16); N = 8: the upper half of K is unused
        float eb[8];
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            float cj;
            if (N == 16) cj = h ? cv[8 + el] : cv[el]; else cj = cv[el];
            const float e = __builtin_amdgcn_exp2f(fmaf(cj - mc, kUpL2E, kUpScaleLog2));   // e^{c - max c} * 2^15
            eb[el] = (N == 8 && h) ? 0.f : e;
        }
        half8 eh, el8;
        split8(eb, eh, el8);
        const ug_h8 *fp = (ug_h8 *)(a.frag + (int64_t)p * a.tiles * 1024) + lane;
        bool vanished = false;
        for (int t = 0; t < a.tiles; ++t) {
            const half8 wh = fp[t * 128], wl = fp[t * 128 + 64];
            gf32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, eh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, el8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, eh, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < OPT / 2; ++q) {   // the lane's outputs of this tile
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) v = fmaf(ea[i], acc[q * N + i], v);
                v *= kUpUnscale;
                const int o = (N == 16) ? 2 * t + h : 4 * t + 2 * h + q;
                vanished = vanished || (v < kUpExactBelow);
                const float r = fmaf(__builtin_amdgcn_logf(v), kUpLn2, ma + mc);
                if (row_ok && o < S) a.out[(b * P + p) * S + o] = r;
            }
        }
        if (__any(vanished)) {   // rare: the wave redoes this partition exactly (every output of its samples)
            if (row_ok)
                for (int o = h; o < S; o += 2) {
                    float m, sum;
                    upper_exact_ms<N>(xa, a.LW + ((int64_t)p * S + o) * N * N, m, sum);
                    a.out[(b * P + p) * S + o] = (m > -INFINITY) ? m + logf(sum) : -INFINITY;
                }
        }
    }
}

// ---- root layer: out[b, c] = logsumexp over (p, i, j); classes c <= CT * 32 / N -------------------------------------
template <int N, int CT>
__global__ __launch_bounds__(256) void prodroot_mfma_kernel(const UpperArgs a) {
    constexpr int CPT = 32 / N;                // classes per tile
    constexpr int HALF = N / 2;                // i's of a class held by one lane half
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int P = a.R / 2, C = a.S;
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + s;
    const bool row_ok = b_raw < a.B;
    const int64_t b = row_ok ? b_raw : a.B - 1;
    float rm[CT * CPT], rs[CT * CPT];
#pragma unroll
    for (int k = 0; k < CT * CPT; ++k) {
        rm[k] = -INFINITY;
        rs[k] = 0.f;
    }
    bool vanished = false;
    for (int p = 0; p < P; ++p) {
        const float *xa = a.in + (b * a.R + 2 * p) * N;
        float av[N], cv[N];
        upper_load<N>(xa, av, cv);
        const float ma = upper_max<N>(av), mc = upper_max<N>(cv);
        // this lane's rows of a tile are i = (u & 3) + 8 ((u >> 2) & 1) + 4 h (N = 16) / (u & 3) + 4 h (N = 8)
        float es[HALF];
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
            const int i0 = (N == 16) ? (k & 3) + 8 * (k >> 2) : k;
            const float ai = h ? av[i0 + 4] : av[i0];
            es[k] = __builtin_amdgcn_exp2f((ai - ma) * kUpL2E);
        }
        float eb[8];
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            float cj;
            if (N == 16) cj = h ? cv[8 + el] : cv[el]; else cj = cv[el];
            const float e = __builtin_amdgcn_exp2f(fmaf(cj - mc, kUpL2E, kUpScaleLog2));   // e^{c - max c} * 2^15
            eb[el] = (N == 8 && h) ? 0.f : e;
        }
        half8 eh, el8;
        split8(eb, eh, el8);
        const ug_h8 *fp = (ug_h8 *)(a.frag + (int64_t)p * CT * 1024) + lane;
        const float m = ma + mc;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const half8 wh = fp[t * 128], wl = fp[t * 128 + 64];
            gf32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, eh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, el8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, eh, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < HALF; ++k) v = fmaf(es[k], acc[q * HALF + k], v);
                v += __shfl_xor(v, 32, 64);              // the other half of the i's
                v *= kUpUnscale;
                const int k = t * CPT + q;
                vanished = vanished || (v < kUpExactBelow && k < C && m > -INFINITY);
                // running (max, scaled sum) over the partitions
                const float mm = fmaxf(rm[k], m);
                const float mm0 = (mm == -INFINITY) ? 0.f : mm;
                rs[k] = rs[k] * __builtin_amdgcn_exp2f((rm[k] - mm0) * kUpL2E) + v * __builtin_amdgcn_exp2f((m - mm0) * kUpL2E);
                rm[k] = mm;
            }
        }
    }
    const bool redo = __any(vanished);
#pragma unroll
    for (int k = 0; k < CT * CPT; ++k) {
        if (k < C) {
            float r = (rm[k] > -INFINITY) ? fmaf(__builtin_amdgcn_logf(rs[k]), kUpLn2, rm[k]) : -INFINITY;
            if (redo) {   // rare: exact log-domain root over all partitions
                float mm = -INFINITY, ss = 0.f;
                for (int p = 0; p < P; ++p) {
                    float pm, ps;
                    upper_exact_ms<N>(a.in + (b * a.R + 2 * p) * N, a.LW + ((int64_t)k * P + p) * N * N, pm, ps);
                    if (ps > 0.f && pm > -INFINITY) {
                        if (pm > mm) {
                            ss = ss * expf(mm - pm) + ps;
                            mm = pm;
                        } else {
                            ss += ps * expf(pm - mm);
                        }
                    }
                }
                r = (mm > -INFINITY) ? mm + logf(ss) : -INFINITY;
            }
            if (row_ok && h == 0) a.out[b * C + k] = r;
        }
    }
}

static int root_ct(int N, int C) {   // class tiles of the root kernel: 1, 2, 4 or 8
    const int t = cdiv((int64_t)C * N, 32);
    return t <= 1 ? 1 : (t <= 2 ? 2 : (t <= 4 ? 4 : 8));
}
bool upper_mfma_shape_ok(bool root, int N, int S) {
    if (!(N == 8 || N == 16)) return false;
    if (root) return cdiv((int64_t)S * N, 32) <= 8;
    return (S * N) % 32 == 0;
}
int64_t upper_mfma_frag_bytes(int R, int N, int S) {   // (covers the sum and the root layout)
    const int64_t t = cdiv((int64_t)S * N, 32);
    return align_up((int64_t)(R / 2) * (t > 8 ? t : 8) * 2048, 256);
}

// W / LW: linear and log softmax weights (already computed by the caller), frag: upper_mfma_frag_bytes() of scratch
int upper_mfma_forward(bool root, const float *in, const float *W, const float *LW, int64_t B, int R, int N, int S,
                       float *out, void *frag, bool frag_cached, hipStream_t st, const unsigned *gate) {
    const int P = R / 2, tiles = root ? root_ct(N, S) : cdiv((int64_t)S * N, 32);
    if (!frag_cached)
        DPK_LAUNCH(upper_pack_kernel, dim3(cdiv((int64_t)P * tiles * 64, 256)), dim3(256), 0, st, W, P, N, S,
                       root ? 1 : 0, tiles, (uint16_t *)frag, gate);
    UpperArgs a{};
    a.in = in; a.frag = (const uint16_t *)frag; a.LW = LW; a.out = out; a.B = B; a.R = R; a.S = S; a.tiles = tiles;
    const int gx = cdiv(B, 128);
    if (!root) {
        // partitions per block: enough blocks to fill the chip at small batches, all of them in one block at large ones
        int ppb = P;
        while (ppb > 1 && (int64_t)gx * cdiv(P, ppb) < 1024) ppb = (ppb + 1) / 2;
        a.ppb = ppb;
        const dim3 grid(gx, cdiv(P, ppb));
        if (N == 16) DPK_LAUNCH(prodsum_mfma_kernel<16>, grid, dim3(256), 0, st, a);
        else DPK_LAUNCH(prodsum_mfma_kernel<8>, grid, dim3(256), 0, st, a);
    } else {
#define DPK_ROOT(NN, CTT) DPK_LAUNCH((prodroot_mfma_kernel<NN, CTT>), dim3(gx), dim3(256), 0, st, a)
        if (N == 16) {
            if (tiles <= 1) DPK_ROOT(16, 1); else if (tiles <= 2) DPK_ROOT(16, 2); else if (tiles <= 4) DPK_ROOT(16, 4);
            else DPK_ROOT(16, 8);
        } else {
            if (tiles <= 1) DPK_ROOT(8, 1); else if (tiles <= 2) DPK_ROOT(8, 2); else if (tiles <= 4) DPK_ROOT(8, 4);
            else DPK_ROOT(8, 8);
        }
#undef DPK_ROOT
    }
    DPK_CHECK_LAUNCH("upper_mfma_forward");
    return DPK_OK;
}

}  // namespace dpk
