// RAT-SPN fused forward for 8-CHANNEL models (depth 2, unit-scale Gaussian leaves, rg_batch = 8, <= 8 repetitions):
// leaf layer on the matrix cores, product / sum / root layers in registers, one launch for any batch size.
//
// reference: RatSpn.forward (deeprob/spn/models/ratspn.py:105-122) = RegionGraphLayer.forward + GaussianLayer
// (deeprob/spn/layers/ratspn.py:87-108, :160-213), ProductLayer :272-286, SumLayer :363-378, RootLayer :446-458 at the
// sizes of experiments/ratspn.py (rg_batch = rg_sum = 8) -- BASELINE config 2's second line.
//
// Same formulation, tables and arithmetic as ratspn_gemm.hip (read its header first).  With 8 channels a repetition
// fills one 32-column MFMA tile (4 regions x 8 channels), so the mapping is by REPETITION instead of by feature slice:
//   * a work-group owns 32 samples (one MFMA column block); wave w owns repetition w: the whole K range of its column
//     tile, 16 accumulator registers, no partial sums to exchange.  The MFMA leaves lane (sample s, half h) with the 8
//     channels of regions {2h, 2h+1} of repetition w: exactly one partition's inputs, so the product + sum node of that
//     partition runs in the lane's registers, the partner partition arrives by one v_permlane32_swap per value and the
//     repetition's share of the root follows; only the root's log-sum-exp over the repetitions crosses waves (LDS);
//   * the x tile (100 KB at D = 784) is staged once in LDS by DMA ([K-step][row][64 bytes], source-side XOR swizzle as
//     in ratspn_gemm_small.hip), shared by the eight waves; each wave streams ITS tile's table fragments from L2 into
//     registers, a dozen K-steps ahead (the accumulators being 16 registers, the file is free for that);
//   * three kernels (leaf | product+sum | product+root), two [B, 32, 8] tensors between them and three table builds
//     become one launch and one table check.
// Marginalised evidence: while the NaN hint is up (slow_hint) the negated-constant fragments travel with the mean
// fragments; otherwise a K-step that holds NaN fetches them on demand.  +-inf / huge evidence, large sums of squares
// and models outside the expanded square's envelope: the wave evaluates its repetition's leaf sums exactly, per element
// (every wave sees the same x tile, so all eight take that decision together).  Vanished sum nodes fall back to the
// log domain inside prodsum_node / root_partial (ratspn_nodes.h).
#include "ratspn_gemm_fused.h"
#include <stdlib.h>

namespace dpk {

constexpr int kWideWaves = 8;      // waves per work-group = repetitions (column tiles) it can hold
constexpr int kWideI = 8;
constexpr int kWideMaxC = 32;      // classes: the repetitions' root partials are exchanged through LDS

template <int S, bool MARG>
__global__ __launch_bounds__(kWideWaves * 64) void ratspn_gemm_wide_kernel(const GemmArgs a) {
    constexpr int I = kWideI;
    constexpr int PF = MARG ? 6 : 12;                 // K-steps of table fragments in flight per wave
    typedef const __attribute__((address_space(1))) half8 gh8;
    typedef __attribute__((address_space(3))) const gf32x4 lf4;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D, NT = a.reps;
    const int NKS = (D + 15) >> 4;
    const int64_t b0 = (int64_t)blockIdx.x * 32;
    const int64_t b = b0 + s;
    const int nvalid = (int)min((int64_t)32, a.B - b0);
    const bool mine = wave < NT;                       // (a model with fewer repetitions leaves waves without a tile)
    const int rho = mine ? wave : NT - 1;
    // LDS: [0, NKS * 2048) the x tile; behind it the wave's copy of its two partitions' sum weights (2 x S x 64 floats)
    lfloat *w0_l = (lfloat *)(smem + NKS * 2048) + wave * (2 * S * I * I);

    // ---- requests: the x tile (every wave its share of the K-steps), then the first PF K-steps of fragments ----------
    {
        const gcchar_p xt = (gcchar_p)(a.x + b0 * D);
        unsigned voff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int P = i * 64 + lane, row = P >> 2, c = (P & 3) ^ ((row >> 2) & 3);
            voff[i] = (unsigned)(min(row, nvalid - 1) * D + c * 4) * 4u;
        }
        for (int ks = wave; ks < NKS; ks += kWideWaves) {
            const int f0 = ks * 16;
            const unsigned dst = (unsigned)(uintptr_t)smem + ks * 2048;
            if (f0 + 16 <= D) {
                glds16(voff[0] + f0 * 4u, xt, dst);
                glds16(voff[1] + f0 * 4u, xt, dst + 1024);
            } else {   // ragged last K-step: pieces beyond the row re-fetch its first one (zeroed by the consumer)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int P = i * 64 + lane, row = P >> 2, c = (P & 3) ^ ((row >> 2) & 3);
                    const bool in = f0 + c * 4 + 4 <= D;
                    glds16(in ? voff[i] + f0 * 4u : (unsigned)(min(row, nvalid - 1) * D) * 4u, xt, dst + i * 1024);
                }
            }
        }
    }
    half8 mh[PF], ml[PF], ch[MARG ? PF : 1], cl[MARG ? PF : 1];
    const gcchar_p tbase = (gcchar_p)a.mtab + ((int64_t)rho * 2048 + lane * 16);
    const gcchar_p cbase = (gcchar_p)a.ctab + ((int64_t)rho * 2048 + lane * 16);
    auto load_frags = [&](int slot, int ks) {
        const int64_t o = (int64_t)min(ks, NKS - 1) * NT * 2048;
        mh[slot] = *(gh8 *)(tbase + o);
        ml[slot] = *(gh8 *)(tbase + o + 1024);
        if constexpr (MARG) {
            ch[slot] = *(gh8 *)(cbase + o);
            cl[slot] = *(gh8 *)(cbase + o + 1024);
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load_frags(k, k);
    // the wave's sum weights (linear softmax rows of its two partitions) into its LDS slice
    {
        const float *wp = a.W0 + (int64_t)rho * 2 * S * I * I;
        for (int e = lane; e < 2 * S * I * I; e += 64) w0_l[e] = wp[e];
    }
    bool model_ok = a.elig[rho] != 0;
    __syncthreads();   // (drains every request above: the x tile is in LDS for everyone)

    // ---- phase 1: P^T = M^T x^T over the whole K range of the wave's tile --------------------------------------------
    gf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float qsum = 0.f;
    bool need_exact = false, saw_nan = false;
    unsigned long long odd_mask = 0ull;   // K-steps whose constants the validity GEMM accumulated
    const int sw = (s >> 2) & 3;
    const unsigned xo0 = (unsigned)(s * 64 + (((h * 2) ^ sw) << 4)), xo1 = (unsigned)(s * 64 + (((h * 2 + 1) ^ sw) << 4));
    // (a.ablate: measurement only, DPK_GEMM_ABLATE -- 1 no K loop, 2 no upper layers)
    for (int k0 = 0; k0 < ((a.ablate & 1) ? 0 : NKS); k0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int ks = k0 + k;
            if (ks < NKS) {   // (wave-uniform)
                const lchar *xb = smem + ks * 2048;
                const gf32x4 x0 = *(lf4 *)(xb + xo0), x1 = *(lf4 *)(xb + xo1);
                float v[8];
                {
                    const int f0 = ks * 16 + h * 8;
                    const bool in0 = f0 + 4 <= D, in1 = f0 + 8 <= D;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = in0 ? x0[i] : 0.f;
                        v[4 + i] = in1 ? x1[i] : 0.f;
                    }
                }
                gf32x2 tq2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const gf32x2 pv = {v[i], v[i + 1]};
                    tq2 = __builtin_elementwise_fma(pv, pv, tq2);
                }
                float tq = tq2[0] + tq2[1];
                const bool odd = __any(!(tq < kGemmStepBound));
                half8 valid;
                if (odd) {
                    odd_mask |= 1ull << ks;
                    tq = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float vi = v[i];
                        const bool isn = vi != vi;
                        const bool big = !isn && !(fabsf(vi) < kGemmAbsBound);
                        need_exact = need_exact || big;
                        saw_nan = saw_nan || isn;
                        v[i] = (isn || big) ? 0.f : vi;
                        valid[i] = isn ? (_Float16)0.0f : (_Float16)1.0f;
                        tq = fmaf(v[i], v[i], tq);
                    }
                }
                qsum += tq;
                half8 xh, xl;
                split8(v, xh, xl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[k], xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[k], xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[k], xh, acc, 0, 0, 0);
                if (odd) {
                    half8 c0, c1;
                    if constexpr (MARG) {
                        c0 = ch[k];
                        c1 = cl[k];
                    } else {
                        const int64_t o = (int64_t)ks * NT * 2048;
                        c0 = *(gh8 *)(cbase + o);
                        c1 = *(gh8 *)(cbase + o + 1024);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c0, valid, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1, valid, acc, 0, 0, 0);
                }
                load_frags(k, ks + PF);   // (the slot is free again: the fragments PF K-steps ahead)
            }
        }
    }
    const float qtot = qsum + __shfl_xor(qsum, 32, 64);
    // the expanded square is within the 1e-5 bar while sum x^2 <= 36 D (|mu| <= 6: DESIGN 3.3)
    const bool lane_exact = need_exact || !(qtot <= kExpandBound * kExpandBound * (float)D);
    model_ok = __all(model_ok);
    const bool exact = !model_ok || __any(lane_exact);   // (the same x tile in every wave: the same verdict in every wave)

    // ---- leaf sums of the lane's partition: regions 2h (a) and 2h + 1 (c) of repetition rho -------------------------
    float va[I], vc[I];
    if (!exact) {
        float cst[16];
        if (odd_mask == 0ull) {
            const float *bt = a.biasT + (h * NT + rho) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) cst[i] = bt[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) cst[i] = 0.f;
            for (int ks = 0; ks < NKS; ++ks) {
                if ((odd_mask >> ks) & 1ull) continue;
                const float *bk = a.biasK + ((ks * 2 + h) * NT + rho) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) cst[i] += bk[i];
            }
        }
#pragma unroll
        for (int k = 0; k < I; ++k) {
            va[k] = acc[k] + cst[k];                       // (the common -1/2 sum x^2 reaches the root as qtot)
            vc[k] = acc[I + k] + cst[I + k];
        }
    } else {
        // exact per-element evaluation of the partition's two regions (any scale, any evidence: nan_to_num_ at ratspn.py:103)
        const float *xr = a.x + (b < a.B ? b : a.B - 1) * D;
        const int d = a.d;
#pragma unroll
        for (int k = 0; k < I; ++k) {
            va[k] = 0.f;
            vc[k] = 0.f;
        }
        for (int qq = 0; qq < 2; ++qq) {
            const int r = rho * 4 + 2 * h + qq;
            float t[I];
#pragma unroll
            for (int k = 0; k < I; ++k) t[k] = 0.f;
            for (int j = 0; j < d; ++j) {
                const int64_t o = (int64_t)r * d + j;
                if (a.pad != nullptr && a.pad[o]) continue;
                const float xv = xr[a.mask[o]];
#pragma unroll
                for (int k = 0; k < I; ++k) {
                    const int64_t po = ((int64_t)r * I + k) * d + j;
                    const float mu = a.loc[po], sg = a.scale[po];
                    const float dlt = xv - mu;
                    t[k] += nan_to_num_f(fmaf(dlt * dlt, -0.5f / (sg * sg), -logf(sg) - kLogSqrt2Pi));
                }
            }
#pragma unroll
            for (int k = 0; k < I; ++k) {
                if (qq == 0) va[k] = t[k]; else vc[k] = t[k];
            }
        }
    }
    __syncthreads();   // every wave is done with the x tile: its LDS becomes scratch and the root exchange buffer
    if (a.ablate & 2) return;

    // ---- the partition's product + sum node, then the repetition's share of the root ---------------------------------
    LseScratch sc{reinterpret_cast<float *>(smem_generic) + tid * (2 * I)};                  // [512][16] floats = 32 KB
    float *xch = reinterpret_cast<float *>(smem_generic) + kWideWaves * 64 * 2 * I;           // [reps][32][2 C]
    float n1[S];
    {
        const lfloat *wl = w0_l + h * S * I * I;
        const float *lw = a.LW0 + ((int64_t)rho * 2 + h) * S * I * I;
        prodsum_node<I, S>(va, vc, wl, lw, sc, n1);
    }
    float ta[S], tc[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned bits = __float_as_uint(n1[o]);
        const u32x2 sw2 = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        ta[o] = __uint_as_float(sw2[0]);
        tc[o] = __uint_as_float(sw2[1]);
    }
    float ea[S], ec[S], ma, mc;
    exp_children<S>(ta, ea, ma);
    exp_children<S>(tc, ec, mc);
    const int M = NT * S * S, C = a.C;
    if (mine) {
        // the two lanes of a sample hold the same (ta, tc): they split the classes
        for (int cl = h; cl < C; cl += 2) {
            float pm, ps;
            const cfloat_p wr = a.Wr + (int64_t)cl * M + rho * S * S;
            const cfloat_p lwr = a.LWr + (int64_t)cl * M + rho * S * S;
            root_partial<S>(ta, tc, ea, ec, ma, mc, wr, lwr, sc, pm, ps);
            if (!(ps > 0.f)) pm = -INFINITY;
            xch[((rho * 32 + s) * C + cl) * 2] = pm;
            xch[((rho * 32 + s) * C + cl) * 2 + 1] = ps;
        }
    }
    __syncthreads();
    // ---- root: log-sum-exp over the repetitions; thread = (sample, class slot) ---------------------------------------
    double part = 0.0;
    {
        const int smp = tid >> 4, slot = tid & 15;
        const int64_t bs = b0 + smp;
        // sum x^2 of the sample: computed by the lanes (smp, h = 0 / 1) of every wave; take wave 0's through LDS
        float *qx = xch + NT * 32 * C * 2;
        if (wave == 0 && h == 0) qx[s] = exact ? 0.f : -0.5f * qtot;   // (the exact leaf sums already carry it)
        __syncthreads();
        const float qterm = qx[smp];
        for (int cl = slot; cl < C; cl += 16) {
            float mm = -INFINITY, ss = 0.f;
            for (int r = 0; r < NT; ++r) lse_merge(mm, ss, xch[((r * 32 + smp) * C + cl) * 2], xch[((r * 32 + smp) * C + cl) * 2 + 1]);
            const float ll = ((mm > -INFINITY) ? mm + logf(ss) : -INFINITY) + qterm;
            if (bs < a.B) {
                a.out[bs * C + cl] = ll;
                part += (double)ll;
            }
        }
    }
    if (a.ll_sum != nullptr) {
        double *red = reinterpret_cast<double *>(xch + NT * 32 * C * 2 + 32);
        part = wave_reduce_sum(part);
        __syncthreads();
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kWideWaves; ++w) tot += red[w];
            atomicAdd(a.ll_sum, tot);
            if (blockIdx.x == 0) atomicAdd(a.ll_sum + 1, (double)a.B * (double)a.C);
        }
    }
    if (saw_nan && tid == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool gemm_wide_shape_ok(int D, int reps, int I, int S, int C) {
    if (I != kWideI || !(S == 2 || S == 4 || S == 8) || reps < 1 || reps > kWideWaves || C > kWideMaxC) return false;
    const int NKS = cdiv(D, 16);
    if (NKS > 64) return false;   // K-step bit mask
    // LDS: the x tile + the waves' sum weights; afterwards scratch (32 KB) + the root exchange buffer in its place
    const size_t p1 = (size_t)NKS * 2048 + (size_t)kWideWaves * 2 * S * kWideI * kWideI * 4;
    const size_t p2 = (size_t)kWideWaves * 64 * 2 * kWideI * 4 + ((size_t)reps * 32 * C * 2 + 32) * 4 + kWideWaves * 8 + 64;
    return p1 <= 160 * 1024 && p2 <= (size_t)NKS * 2048;
}

template <int S, bool MARG>
static int gemm_wide_launch(const GemmArgs &a, hipStream_t st) {
    const int NKS = cdiv(a.D, 16);
    const size_t lds = (size_t)NKS * 2048 + (size_t)kWideWaves * 2 * S * kWideI * kWideI * 4;
    auto kern = ratspn_gemm_wide_kernel<S, MARG>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_FUSED);
    if (ev0) (void)hipEventRecord(ev0, st);
    DPK_LAUNCH(kern, dim3(cdiv(a.B, 32)), dim3(kWideWaves * 64), lds, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_gemm_wide_kernel");
    return DPK_OK;
}

// The caller (ratspn_gemm_forward) has built the tables and filled the argument block.
int ratspn_gemm_wide_forward(const GemmArgs &a, int S, hipStream_t st) {
    const bool marg = a.marginal != 0;
    switch (S) {
        case 2: return marg ? gemm_wide_launch<2, true>(a, st) : gemm_wide_launch<2, false>(a, st);
        case 4: return marg ? gemm_wide_launch<4, true>(a, st) : gemm_wide_launch<4, false>(a, st);
        case 8: return marg ? gemm_wide_launch<8, true>(a, st) : gemm_wide_launch<8, false>(a, st);
    }
    set_error("ratspn_gemm_wide: sums=%d not built", S);
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk
