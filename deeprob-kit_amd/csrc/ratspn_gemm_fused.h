// Pieces shared by the two fused RAT-SPN kernels with the leaf layer on the matrix cores (ratspn_gemm.hip: persistent
// 128-sample tiles behind an LDS-DMA ring, large batches; ratspn_gemm_small.hip: 32-sample tiles with the features
// split over the waves, small batches): the argument block, the exact per-element evaluation of a wave's 32 samples
// and the in-register upper layers (product + sum nodes, root) on a lane's accumulators.
//
// reference: ProductLayer / SumLayer / RootLayer (deeprob/spn/layers/ratspn.py:272-286, :363-378, :446-458) as chained
// by RatSpn.forward (deeprob/spn/models/ratspn.py:105-122)
#pragma once
#include "common.h"
#include "ratspn_nodes.h"
#include "ratspn_gemm_common.h"
#include <math.h>

namespace dpk {

struct GemmArgs {
    const float *x;
    int64_t B;
    int D, d, reps, C, NCH, ntiles;
    const uint16_t *mtab, *ctab;
    const float *biasT;   // [2][NT][16] whole-row constants in the accumulator order of a lane
    const float *biasC;   // [NCH][2][NT][16] the same per chunk (tiles with marginalised evidence)
    const float *biasK;   // [NKS][2][NT][16] per K-step, [8][2][NT][16] per wave slice (small-batch kernel)
    const float *biasS;
    const int *elig;
    const float *W0;   // [reps*2][S][I*I] linear softmax weights (copied into LDS)
    const float *LW0;  // log-softmax weights (exact fallback of a node, exact evaluation)
    cfloat_p Wr, LWr;  // [C][reps*S*S]
    float *out;
    double *ll_sum;
    int ll_cnt;        // index of the count behind ll_sum: 1 ({sum, count}) or 16 (DPK_FLAG_LL_SUM_SPREAD: sixteen partial sums, then the count)
    // exact evaluation
    const int64_t *mask;
    const uint8_t *pad;
    const float *loc, *scale;
#ifdef DPK_TIMELINE
    unsigned long long *dbg;   // [blocks][waves][64][8] s_memtime stamps (measurement builds)
#endif
    int ablate;       // measurement only (DPK_GEMM_ABLATE): 1 no compute, 2 no table DMA, 4 no x DMA
    int *slow_flag;   // host-mapped hint word (may be null): launch number of the last launch that met NaN evidence
    int launch_seq;
    int marginal;     // host: a launch within the last 256 met NaN evidence (selects the variant built for it)
    const uint16_t *upfrag;   // 8-channel kernels: MFMA fragments of the first sum layer (ratspn_gemm_prep.h)
    // Training forward (dpk_ratspn_forward_train, EMIT instantiations of the 32-sample kernels): what the layers' backward
    // kernels read, written on the way up.  All three are RELATIVE to the sample's common quadratic term: the GEMM form
    // carries -1/2 sum x^2 to the root instead of through the regions, and the backward of a sum / root layer only ever
    // sees in - out (ratspn_layers.hip sum_bwd_kernel), so the shift cancels level by level.
    float *emit_leaf;   // [B, R, I]        leaf layer outputs (+ 1/2 sum x^2 over the region's variables)
    float *emit_sum;    // [B, 2 reps, S]   first sum layer outputs (+ 1/2 sum x^2 over the partition's variables)
    float *emit_out;    // [B, C]           root outputs (+ 1/2 sum x^2 over all variables)
};

__device__ __forceinline__ void lse_merge(float &m, float &s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    if (mm == -INFINITY) {
        s = 0.f;
        return;
    }
    s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
}


// 16-lane (one sample's partitions) exchanges on the DPP path (VALU operand modifiers, a few cycles; ds_bpermute-based
// shuffles cost an LDS round trip each, and four dependent ones per reduction were a third of phase 2).
// quad_perm [1,0,3,2] = lane ^ 1, [2,3,0,1] = lane ^ 2; row_half_mirror: i <-> 7 - i; row_mirror: i <-> 15 - i -- after
// the four steps every lane of the row holds the reduction over all 16.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    v += dpp_f<kDppHalfMirror>(v);
    v += dpp_f<kDppMirror>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f<kDppMirror>(v));
    return v;
}
__device__ __forceinline__ unsigned row16_or(unsigned v) {
    v |= dpp_u<kDppXor1>(v);
    v |= dpp_u<kDppXor2>(v);
    v |= dpp_u<kDppHalfMirror>(v);
    v |= dpp_u<kDppMirror>(v);
    return v;
}

__device__ __forceinline__ void row16_lse_merge(float &m, float &sum) {
    // (m, sum) pairs of the 16 lanes of a row -> their log-sum-exp combination in every lane
    float om, os;
    om = dpp_f<kDppXor1>(m); os = dpp_f<kDppXor1>(sum); lse_merge(m, sum, om, os);
    om = dpp_f<kDppXor2>(m); os = dpp_f<kDppXor2>(sum); lse_merge(m, sum, om, os);
    om = dpp_f<kDppHalfMirror>(m); os = dpp_f<kDppHalfMirror>(sum); lse_merge(m, sum, om, os);
    om = dpp_f<kDppMirror>(m); os = dpp_f<kDppMirror>(sum); lse_merge(m, sum, om, os);
}


// ---- exp-domain helpers of the fast upper layers ---------------------------------------------------------------
// e[i] = 2^((x[i] - max) log2 e); returns max (0 for an all -inf input, whose exponentials are then 0)
template <int NI> __device__ __forceinline__ float exp2_children(const float (&x)[NI], float (&e)[NI]) {
    float m = x[0];
#pragma unroll
    for (int i = 1; i < NI; ++i) m = fmaxf(m, x[i]);
    const float m0 = (m == -INFINITY) ? 0.f : m;
#pragma unroll
    for (int i = 0; i < NI; ++i) e[i] = __builtin_amdgcn_exp2f((x[i] - m0) * 1.4426950408889634f);
    return m0;
}

// Exact per-element evaluation of the 32 samples of a wave (any scale, any evidence): lane (s, h) takes the
// repetitions rho = 2m + h, the two lanes of a sample meet in one shuffle per class.  Slow by design.
template <int I, int S, int NT>
__device__ __forceinline__ void gemm_exact_body(const GemmArgs &a, int64_t bw0, int lane, LseScratch sc, unsigned gmask = 0xFu) {
    // gmask: bit g set = the samples 8 g .. 8 g + 7 of the wave are stored (and summed); the slice mapping leaves the
    // fast path in groups of 8 samples
    constexpr int RPT = 8 / I;
    constexpr int RH = (NT * RPT + 1) / 2;   // repetitions per lane half
    const int s = lane & 31, h = lane >> 5;
    const int64_t b = bw0 + s;
    const bool valid = b < a.B && ((gmask >> (s >> 3)) & 1u) != 0u;
    const float *xr = a.x + (b < a.B ? b : a.B - 1) * a.D;
    const int d = a.d;
    float n1[RH][2][S];
#pragma unroll
    for (int m = 0; m < RH; ++m) {
        const int rho = 2 * m + h;
        float leaf[4][I];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < I; ++k) leaf[q][k] = 0.f;
        if (rho < a.reps) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rho * 4 + q;
                for (int j = 0; j < d; ++j) {
                    const int64_t o = (int64_t)r * d + j;
                    if (a.pad != nullptr && a.pad[o]) continue;
                    const float xv = xr[a.mask[o]];
#pragma unroll
                    for (int k = 0; k < I; ++k) {
                        const int64_t po = ((int64_t)r * I + k) * d + j;
                        const float mu = a.loc[po], sg = a.scale[po];
                        const float dlt = xv - mu;
                        leaf[q][k] += nan_to_num_f(fmaf(dlt * dlt, -0.5f / (sg * sg), -logf(sg) - kLogSqrt2Pi));
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int64_t wo = ((int64_t)rho * 2 + p) * S * I * I;
                prodsum_node<I, S>(leaf[2 * p], leaf[2 * p + 1], a.W0 + wo, a.LW0 + wo, sc, n1[m][p]);
            }
        }
    }
    const int M = a.reps * S * S;
    double part = 0.0;
    for (int cl = 0; cl < a.C; ++cl) {
        float mm = -INFINITY, ss = 0.f;
#pragma unroll
        for (int m = 0; m < RH; ++m) {
            const int rho = 2 * m + h;
            if (rho < a.reps) {
                float ea[S], ec[S], ma, mc, pm, ps;
                exp_children<S>(n1[m][0], ea, ma);
                exp_children<S>(n1[m][1], ec, mc);
                const float *wr = (const float *)a.Wr + (int64_t)cl * M + rho * S * S;
                const float *lwr = (const float *)a.LWr + (int64_t)cl * M + rho * S * S;
                root_partial<S>(n1[m][0], n1[m][1], ea, ec, ma, mc, wr, lwr, sc, pm, ps);
                lse_merge(mm, ss, pm, ps);
            }
        }
        const float om = __shfl_xor(mm, 32, 64), os = __shfl_xor(ss, 32, 64);
        lse_merge(mm, ss, om, os);
        const float ll = (mm > -INFINITY) ? mm + logf(ss) : -INFINITY;
        if (h == 0 && valid) {
            a.out[b * a.C + cl] = ll;
            part += (double)ll;
        }
    }
    if (a.ll_sum != nullptr) {
        part = wave_reduce_sum(part);
        if (lane == 0) atomicAdd(a.ll_sum, part);   // (the count: once per launch, at the end of the kernel)
    }
}

// out of line: the ring kernel's hot loop keeps its registers (the small-batch kernel inlines the body instead: a call
// needs a stack, and a kernel with scratch memory pays for it at every dispatch)
template <int I, int S, int NT>
__device__ __noinline__ void gemm_exact_wave(const GemmArgs &a, int64_t bw0, int lane, LseScratch sc) {
    gemm_exact_body<I, S, NT>(a, bw0, lane, sc);
}

// The upper layers of ONE sample on the lane pair (s, h = 0 / 1) that holds its leaf sums: acc[t][u] + cst[t][u] are the
// leaf outputs (minus the common -1/2 sum x^2, which reaches the root as qtot) of regions {2h, 2h+1} of every repetition
// in the accumulator order of the MFMA (see ratspn_gemm.hip "Mapping").  Writes out[b, :] from the h == 0 lane and
// returns the lane's fp64 share of their sum in part_out.  Returns true when some node of the WAVE vanished in the exp
// domain (dominant pair under a vanishing weight): the caller then evaluates the wave's samples exactly.
template <int I, int S, int NT>
__device__ __forceinline__ bool gemm_upper_fast(const GemmArgs &a, const gf32x16 (&acc)[NT], const float (&cst)[NT][16],
                                                const lfloat *w0_l, float qtot, int h, int64_t b, double &part_out) {
    constexpr int RPT = 8 / I;
    // Upper layers in the exp domain on the hardware's base-2 transcendentals; a node whose scaled sum
    // vanishes (dominant pair under a vanishing weight) is redone exactly, out of line (gemm_node_exact).
    constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    // phase A, branch free so that the independent nodes interleave (one wave per SIMD: a dependent chain of
    // transcendentals would otherwise run at its latency): every product + sum node of the lane's partitions
    float n1[NT * RPT][S];
    bool vanished = false;   // some node's scaled sum fell below 1e-30 (dominant pair under a vanishing weight)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ap = 0; ap < RPT; ++ap) {
            const int rho = t * RPT + ap;
            float va[I], vc[I];
#pragma unroll
            for (int k = 0; k < I; ++k) {
                va[k] = acc[t][(ap * 2) * I + k] + cst[t][(ap * 2) * I + k];
                vc[k] = acc[t][(ap * 2 + 1) * I + k] + cst[t][(ap * 2 + 1) * I + k];
            }
            float ea[I], ec[I];
            const float ma = exp2_children<I>(va, ea), mc = exp2_children<I>(vc, ec);
            const int wo = (min(rho, a.reps - 1) * 2 + h) * S * I * I;
#pragma unroll
            for (int o = 0; o < S; ++o) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < I; ++i) {
                    float tt = 0.f;
#pragma unroll
                    for (int j = 0; j < I; ++j) tt = fmaf(w0_l[wo + (o * I + i) * I + j], ec[j], tt);
                    v = fmaf(ea[i], tt, v);
                }
                n1[rho][o] = fmaf(__builtin_amdgcn_logf(v), kLn2, ma + mc);
                vanished = vanished || (v < 1e-30f && rho < a.reps);
            }
        }
    }
    // both lanes of a sample finish every repetition (the root weights stay wave-uniform):
    // v_permlane32_swap leaves partition 0's outputs in one register and partition 1's in the other
    float ta[NT * RPT][S], tc[NT * RPT][S];
#pragma unroll
    for (int rho = 0; rho < NT * RPT; ++rho)
#pragma unroll
        for (int o = 0; o < S; ++o) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const unsigned bits = __float_as_uint(n1[rho][o]);
            const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
            ta[rho][o] = __uint_as_float(sw2[0]);
            tc[rho][o] = __uint_as_float(sw2[1]);
        }
    // root: per repetition (m, s) with logsumexp = m + ln s; the exponentials do not depend on the class
    float ea[NT * RPT][S], ec[NT * RPT][S], mr[NT * RPT];
    float mtop = -INFINITY;
#pragma unroll
    for (int rho = 0; rho < NT * RPT; ++rho) {   // (branch free: a column tile's spare repetitions get -inf)
        const float m2 = exp2_children<S>(ta[rho], ea[rho]) + exp2_children<S>(tc[rho], ec[rho]);
        mr[rho] = (rho < a.reps) ? m2 : -INFINITY;
        mtop = fmaxf(mtop, mr[rho]);
    }
    const float mtop0 = (mtop == -INFINITY) ? 0.f : mtop;
    float scale[NT * RPT];
#pragma unroll
    for (int rho = 0; rho < NT * RPT; ++rho) scale[rho] = __builtin_amdgcn_exp2f((mr[rho] - mtop0) * kL2E);
    const int M = a.reps * S * S;
    const float qterm = -0.5f * qtot;
    double part = 0.0;
    // a vanished node anywhere in the wave: the wave's samples go through the exact evaluation instead
    // (rare: a softmax weight below e^-69 on the dominant pair)
    if (__any(vanished)) return true;
    for (int cl = 0; cl < a.C; ++cl) {
        float tot = 0.f;
#pragma unroll
        for (int rho = 0; rho < NT * RPT; ++rho) {
            const int wo = cl * M + min(rho, a.reps - 1) * S * S;   // (spare repetitions: scale == 0)
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                float tt = 0.f;
#pragma unroll
                for (int j = 0; j < S; ++j) tt = fmaf(a.Wr[wo + i * S + j], ec[rho][j], tt);
                v = fmaf(ea[rho][i], tt, v);
            }
            vanished = vanished || (v < 1e-30f && mr[rho] > -INFINITY);
            tot = fmaf(v, scale[rho], tot);
        }
        const float ll = ((mtop > -INFINITY) ? fmaf(__builtin_amdgcn_logf(tot), kLn2, mtop) : -INFINITY) + qterm;
        if (h == 0 && b < a.B) {
            a.out[b * a.C + cl] = ll;
            part += (double)ll;
        }
    }
    part_out = part;
    return __any(vanished);   // (the caller's exact evaluation then overwrites what this wave stored)
}

}  // namespace dpk
