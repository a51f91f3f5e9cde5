// In-register product / sum / root nodes of a RAT-SPN repetition, shared by the forward kernels
// (ratspn_fwd.hip: one wave per repetition; ratspn_gemm.hip: MFMA leaf layer, lanes own half a repetition).
#pragma once
#include "common.h"
#include <math.h>

namespace dpk {

// Tables that only earlier kernels write are read through the constant address space: a
// wave-uniform load from it is always selected as s_load (scalar cache), which is the whole
// point of the lane <-> sample mapping.
#define DPK_CONST __attribute__((address_space(4)))
typedef const DPK_CONST float *cfloat_p;
typedef const DPK_CONST int *cint_p;
template <typename T> __host__ __device__ __forceinline__ const DPK_CONST T *as_const(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const DPK_CONST T *)p;
#pragma clang diagnostic pop
}

// --------------------------------------------------------------------------------------
// in-register product+sum node:  out[o] = logsumexp_{i,j}(a[i] + c[j] + lw[o,i,j])
// (ProductLayer.forward ratspn.py:280-285 followed by SumLayer.forward :375-377).
// Fast path in the exp domain with linear softmax weights; when the scaled sum falls
// below 1e-30 (dominant pair far from (argmax a, argmax c) AND a vanishing weight) the
// exact two-pass form with the true maximum is used, which is what torch.logsumexp does.
// --------------------------------------------------------------------------------------
struct LseScratch {
    float *slot;  // per-lane LDS slice, 2*NI floats
};

template <int NI, class LWP>
__device__ __forceinline__ void exact_lse(const float (&a)[NI], const float (&c)[NI],
                                          LWP lw, LseScratch sc, float &m_out, float &s_out) {
    // rare, lane-divergent: keep it small (rolled loops over an LDS copy)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        sc.slot[i] = a[i];
        sc.slot[NI + i] = c[i];
    }
    float m = -INFINITY;
#pragma unroll 1
    for (int i = 0; i < NI; ++i)
#pragma unroll 1
        for (int j = 0; j < NI; ++j) m = fmaxf(m, sc.slot[i] + sc.slot[NI + j] + lw[i * NI + j]);
    float s = 0.f;
    if (m > -INFINITY) {
#pragma unroll 1
        for (int i = 0; i < NI; ++i)
#pragma unroll 1
            for (int j = 0; j < NI; ++j)
                s += expf(sc.slot[i] + sc.slot[NI + j] + lw[i * NI + j] - m);
    }
    m_out = m;
    s_out = s;
}

template <int NI>
__device__ __forceinline__ void exp_children(const float (&a)[NI], float (&ea)[NI], float &ma) {
    float m = a[0];
#pragma unroll
    for (int i = 1; i < NI; ++i) m = fmaxf(m, a[i]);
    const float m0 = (m == -INFINITY) ? 0.f : m;
#pragma unroll
    for (int i = 0; i < NI; ++i) ea[i] = __expf(a[i] - m0);
    ma = m0;
}

template <int NI, int NO, class WP, class LWP>
__device__ __forceinline__ void prodsum_node(const float (&a)[NI], const float (&c)[NI],
                                             WP W, LWP LW, LseScratch sc,
                                             float (&out)[NO]) {
    float ea[NI], ec[NI], ma, mc;
    exp_children<NI>(a, ea, ma);
    exp_children<NI>(c, ec, mc);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) t = fmaf(W[(o * NI + i) * NI + j], ec[j], t);
            v = fmaf(ea[i], t, v);
        }
        if (v < 1e-30f) {
            float m, s;
            exact_lse<NI>(a, c, LW + o * NI * NI, sc, m, s);
            out[o] = (m > -INFINITY) ? m + logf(s) : -INFINITY;
        } else {
            out[o] = ma + mc + __logf(v);
        }
    }
}

// partial of the root log-sum-exp contributed by one repetition: (m, s) with
// logsumexp = m + log s   (RootLayer.forward ratspn.py:454-457 restricted to one repetition)
template <int NI, class WP, class LWP>
__device__ __forceinline__ void root_partial(const float (&a)[NI], const float (&c)[NI],
                                             const float (&ea)[NI], const float (&ec)[NI], float ma,
                                             float mc, WP W, LWP LW, LseScratch sc,
                                             float &m_out, float &s_out) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) t = fmaf(W[i * NI + j], ec[j], t);
        v = fmaf(ea[i], t, v);
    }
    if (v < 1e-30f) {
        exact_lse<NI>(a, c, LW, sc, m_out, s_out);
    } else {
        m_out = ma + mc;
        s_out = v;
    }
}

}  // namespace dpk
