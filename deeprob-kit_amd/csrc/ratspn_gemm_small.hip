// RAT-SPN fused forward for SMALL batches (depth 2, unit-scale Gaussian leaves, leaf layer on the matrix cores).
//
// reference: RatSpn.forward (deeprob/spn/models/ratspn.py:105-122) = RegionGraphLayer.forward + GaussianLayer
// (deeprob/spn/layers/ratspn.py:87-108, :160-213), ProductLayer :272-286, SumLayer :363-378, RootLayer :446-458 --
// at the batch sizes of BASELINE config 2 (4096 samples per call).
//
// Same formulation, tables and arithmetic as ratspn_gemm.hip (read its header first).  What differs is the mapping.
// The ring kernel gives a work-group 128 samples and streams their features through a three-stage LDS ring: at 4096
// samples that is 32 work-groups on 256 compute units, each walking 13 chunks at one HBM round trip per chunk pair
// (18 us, 0.085 of the HBM roofline in round 2).  Here a work-group owns 32 samples -- one MFMA column block -- and its
// EIGHT waves split the FEATURE axis: wave w takes K-steps [w NKS/8, (w+1) NKS/8) of 16 features, requests its slice of
// the 32 rows (plain 16-byte loads, 32 bytes per lane and K-step, every byte of the tile in flight at once) and its
// fragments of the mean table (L2) straight into registers, runs its 3 NT MFMAs per K-step and leaves a partial
// accumulator in LDS.  One barrier later wave 0 adds the eight partials in a fixed order and evaluates the upper layers
// exactly like a compute wave of the ring kernel (gemm_upper_fast).  No ring, no loader waves, one HBM round trip per
// work-group; 128 work-groups at 4096 samples, the kernel is as long as that round trip plus ~1 us of arithmetic.
//
// Marginalised evidence: a K-step that holds NaN runs the validity GEMM against the negated-constant table (as in the
// ring kernel, at K-step instead of chunk granularity); the constants of a wave's clean K-steps come ready-made per
// slice (all clean: the whole-row table) or per K-step.  +-inf / huge evidence, large sums of squares, models outside
// the expanded square's envelope and vanished sum nodes send the 32 samples through gemm_exact_wave.
//
// Results agree with the ring kernel to fp32 rounding, not bit for bit: the K-steps are summed in eight partial
// accumulators here and in one there.  A given batch size always takes the same kernel.
#include "ratspn_gemm_fused.h"
#include "ratspn_gemm_prep.h"
#include <stdlib.h>

namespace dpk {

// bytes of a wave's private LDS region: its x pieces in phase 1 ([K-step][32 rows][64 bytes]); in phase 2 the partial
// sums it hands over ([NT*16 + 2 rows][65 floats]) followed by the exact path's per-lane scratch (64 x 8 floats)
__host__ __device__ constexpr int gemm_small_region_bytes(int NT, int MAXK) {
    const int x = MAXK * 2048, p2 = ((NT * 16 + 2) * 65 + 64 * 8) * 4;
    return ((x > p2 ? x : p2) + 255) / 256 * 256;
}

#ifdef DPK_TIMELINE
// measurement builds (make ../lib/libdeeprob_hip_timeline.so; tools/timeline_small.py): s_memtime stamps per wave
#define SM_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (a.dbg && lane == 0 && blockIdx.x < 256) a.dbg[((int64_t)blockIdx.x * kGemmSmallWaves + wave) * 16 + (i)] = (i) == 15 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SM_STAMP(i) do { } while (0)
#endif

// Exact per-element evaluation of the 4 samples of a wave in the phase-2 roles (16 lanes per sample, lane = one
// (repetition, partition)): any scale, any evidence -- the reference's formula term by term (nan_to_num_ at
// ratspn.py:103), log-domain fallbacks inside the nodes.  Slow by design; returns the sample's log-likelihoods through
// `store` (called by every lane with identical values, class by class).
// `raw0` / `rawr` non-null (a launch that found its tables stale, ratspn_gemm_prep.h): the nodes take their log-softmax
// weights straight from the raw sum / root weights instead of the (being rebuilt) tables.
template <int I, int S, class Store>
__device__ __forceinline__ void small_exact_wave(const GemmArgs &a, const float *xr, int64_t emit_b, int rc, int p, bool active,
                                                 LseScratch sc, const float *raw0, const float *rawr, Store store) {
    const int d = a.d;
    float leaf[2][I];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
        for (int k = 0; k < I; ++k) leaf[qq][k] = 0.f;
        const int r = rc * 4 + 2 * p + qq;
        for (int jx = 0; jx < d; ++jx) {
            const int64_t o = (int64_t)r * d + jx;
            if (a.pad != nullptr && a.pad[o]) continue;
            const float xv = xr[a.mask[o]];
#pragma unroll
            for (int k = 0; k < I; ++k) {
                const int64_t po = ((int64_t)r * I + k) * d + jx;
                const float mu = a.loc[po], sg = a.scale[po];
                const float dlt = xv - mu;
                leaf[qq][k] += nan_to_num_f(fmaf(dlt * dlt, -0.5f / (sg * sg), -logf(sg) - kLogSqrt2Pi));
            }
        }
    }
    float n1[S];
    const int64_t wo = ((int64_t)rc * 2 + p) * S * I * I;
    if (raw0 != nullptr) prodsum_node_raw<I, S>(leaf[0], leaf[1], raw0 + wo, sc.slot, n1);
    else prodsum_node<I, S>(leaf[0], leaf[1], a.W0 + wo, a.LW0 + wo, sc, n1);
    if (a.emit_leaf != nullptr && active && emit_b >= 0) {   // (training forward: the exact values, no shift)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int k = 0; k < I; ++k) a.emit_leaf[(emit_b * (4 * a.reps) + rc * 4 + 2 * p + qq) * I + k] = leaf[qq][k];
#pragma unroll
        for (int o = 0; o < S; ++o) a.emit_sum[(emit_b * (2 * a.reps) + rc * 2 + p) * S + o] = n1[o];
    }
    float ta[S], tc[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
        const float other = dpp_f<kDppXor1>(n1[o]);
        ta[o] = p == 0 ? n1[o] : other;
        tc[o] = p == 0 ? other : n1[o];
    }
    float ea[S], ec[S], ma, mc;
    exp_children<S>(ta, ea, ma);
    exp_children<S>(tc, ec, mc);
    const int M = a.reps * S * S;
    for (int cl = 0; cl < a.C; ++cl) {
        float pm = -INFINITY, ps = 0.f;
        if (active && p == 0) {
            if (rawr != nullptr) {
                root_partial_raw<S>(ta, tc, rawr + (int64_t)cl * M + rc * S * S, raw_row_lse(rawr + (int64_t)cl * M, M),
                                    sc.slot, pm, ps);
            } else {
                const float *wr = (const float *)a.Wr + (int64_t)cl * M + rc * S * S;
                const float *lwr = (const float *)a.LWr + (int64_t)cl * M + rc * S * S;
                root_partial<S>(ta, tc, ea, ec, ma, mc, wr, lwr, sc, pm, ps);
            }
            if (!(ps > 0.f)) pm = -INFINITY;
        }
        row16_lse_merge(pm, ps);
        store(cl, (pm > -INFINITY) ? pm + logf(ps) : -INFINITY);
    }
}

template <int I, int S, int NT, int MAXK>
__global__ __launch_bounds__(kGemmSmallWaves * 64) void ratspn_gemm_small_kernel(const GemmArgs a, const GemmPrepArgs pa) {
    constexpr int RPT = 8 / I;
    constexpr int NMAX = (I > S ? I : S);
    constexpr int NR = NT * 16;                      // accumulator registers of a lane
    constexpr int RSTR = 65;                         // floats between two accumulator rows in LDS (bank skew, see phase 2)
    constexpr int SLOT = (NR + 2) * RSTR;            // floats a wave leaves in LDS: partials, sum x^2, flags
    constexpr int REG = gemm_small_region_bytes(NT, MAXK);   // a wave's private LDS region
    constexpr int REGF = REG / 4;
    constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    typedef const __attribute__((address_space(1))) half8 gh8;
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    static_assert(MAXK <= kGemmSmallMaxK, "K-steps per wave");
    static_assert(NT * RPT <= 8, "a sample's partitions fill at most 16 lanes");
    static_assert(SLOT * 4 + 64 * 2 * NMAX * 4 <= REG && MAXK * 2048 <= REG, "region layout");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A launch that checks its parameter tables itself (kPrepInline, ratspn_gemm_prep.h): the first pa.np work-groups are
    // the table work-groups -- fingerprint, verdict, rebuild if stale -- the others evaluate tile blockIdx.x - pa.np.
    const int np = pa.np;
    if ((int)blockIdx.x < np) {
        gemm_prep_block_inline<I>(pa, (int)blockIdx.x, reinterpret_cast<int *>(smem_generic));
        return;
    }
    const int D = a.D;
    const int NKS = (D + 15) >> 4;
    const int64_t b0 = (int64_t)((int)blockIdx.x - np) * 32;
    SM_STAMP(0);
    SM_STAMP(15);

    // =========================== phase 1: the wave's slice of the feature axis on the matrix cores =================
    // LDS region of the wave, phase 1: its slice of the x tile, [K-step][row][4 pieces of 16 bytes], the pieces of a row
    // XOR-swizzled by (row >> 2) & 3 so that the MFMA-shaped ds_read_b128 (a lane reads two pieces of its row) is
    // conflict free; written by LDS-DMA, 128 pieces = two instructions per K-step, four consecutive lanes fetching 64
    // contiguous bytes of a row (a lane-per-row load touches 64 cache lines per instruction: measured ~64 cycles of
    // address processing each, the small-batch kernel's first bottleneck).
    lchar *my = smem + wave * REG;
    gf32x16 acc[NT];
    float qsum = 0.f;
    bool need_exact = false, saw_nan = false;
    unsigned odd_mask = 0u;   // K-steps of the slice whose constants the validity GEMM accumulated
    const int s = lane & 31, h = lane >> 5;          // MFMA roles: sample of the tile, half of the K-step / of the columns
    {
        const int k0 = wave * NKS / kGemmSmallWaves, k1 = (wave + 1) * NKS / kGemmSmallWaves;
        const int nk = k1 - k0;                      // <= MAXK (checked by the host)
        const int nvalid = (int)min((int64_t)32, a.B - b0);
        // (a.ablate: measurement only, DPK_GEMM_ABLATE -- 1 no K loop, 2 no upper layers, 4 every lane reads row 0, 8 one fragment set)
        const gcchar_p xt = (gcchar_p)(a.x + b0 * D);
        half8 mh[MAXK][NT], ml[MAXK][NT];
        auto load_frags = [&](int kk) {
            const int ks = (a.ablate & 8) ? 0 : min(k0 + kk, NKS - 1);
            const gcchar_p tb = (gcchar_p)a.mtab + ((int64_t)ks * NT * 2048 + lane * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mh[kk][t] = *(gh8 *)(tb + t * 2048);
                ml[kk][t] = *(gh8 *)(tb + t * 2048 + 1024);
            }
        };
        // Every request of the slice goes out before anything waits, K-step by K-step: the two DMA instructions of
        // its x pieces, then its table fragments (plain loads, L2).  Loads retire in order, so once a K-step's fragments
        // are in registers its x pieces are in LDS; the first MFMAs start under the rest of the stream.
        unsigned voff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int P = i * 64 + lane, row = P >> 2, c = (P & 3) ^ ((row >> 2) & 3);
            voff[i] = (unsigned)(((a.ablate & 4) ? 0 : min(row, nvalid - 1)) * D + c * 4) * 4u;
        }
#pragma unroll
        for (int kk = 0; kk < MAXK; ++kk) {
            if (kk < nk) {   // (wave-uniform)
                const int f0 = (k0 + kk) * 16;
                const unsigned dst = (unsigned)(uintptr_t)my + kk * 2048;
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
            if (kk < nk) load_frags(kk);   // (wave-uniform)
            __builtin_amdgcn_sched_barrier(0);   // (keep the request order)
        }
        SM_STAMP(1);   // loads requested
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const int sw = (s >> 2) & 3;
        const unsigned xo0 = (unsigned)(s * 64 + (((h * 2) ^ sw) << 4)), xo1 = (unsigned)(s * 64 + (((h * 2 + 1) ^ sw) << 4));
#pragma unroll
        for (int kk = 0; kk < MAXK; ++kk) {
            if (kk < nk && !(a.ablate & 1)) {   // (wave-uniform)
                // the K-step's fragments have arrived => so have its x pieces (older requests): the opaque zero ties the
                // LDS reads to that wait without a hand-counted vmcnt
                unsigned zero;
                {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 bits = __builtin_bit_cast(u32x4, ml[kk][NT - 1]);
                    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(bits[3]) : "memory");
                }
                const lchar *xb = my + kk * 2048;
                const gf32x4 x0 = *(lf4 *)(xb + xo0 + zero), x1 = *(lf4 *)(xb + xo1 + zero);
                float v[8];
                {
                    const int f0 = (k0 + kk) * 16 + h * 8;
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
                // NaN / +-inf / huge evidence anywhere in the wave's share of this K-step?
                const bool odd = __any(!(tq < kGemmStepBound));
                half8 valid;
                if (odd) {
                    // NaN (marginalised) entries count as 0 and drop out of the constants (validity indicator below);
                    // +-inf / huge entries send the samples through the exact evaluation
                    odd_mask |= 1u << kk;
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
                if (kk == 0) SM_STAMP(2);   // first K-step's operands are here
                half8 xh, xl;
                split8(v, xh, xl);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xh, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xl, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[kk][t], xh, acc[t], 0, 0, 0);
                if (odd) {
                    // - (mu^2/2 + log sqrt(2 pi)) of the variables that ARE observed (table of negated constants)
                    const gcchar_p cb = (gcchar_p)a.ctab + ((int64_t)(k0 + kk) * NT * 2048 + lane * 16);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const half8 ch = *(gh8 *)(cb + t * 2048);
                        const half8 cl = *(gh8 *)(cb + t * 2048 + 1024);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, valid, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, valid, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        SM_STAMP(3);   // K loop done (MFMAs issued)
        if (odd_mask != 0u) {
            // this slice met marginalised evidence: the constants of its CLEAN K-steps join the partial sums here
            // (the validity GEMM has accumulated those of the others); phase 2 then leaves the slice's table out
            for (int kk = 0; kk < nk; ++kk) {
                if ((odd_mask >> kk) & 1u) continue;
                const float *bk = a.biasK + (((k0 + kk) * 2 + h) * NT) * 16;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += bk[t * 16 + i];
            }
        }
    }

    // =========================== phase 2 roles: one (sample, repetition, partition) per thread =====================
    // 16 consecutive lanes own a sample: slot j = 2 rho + p, p = the partition (= the MFMA lane half that accumulated
    // its two regions).  A wave finishes 4 samples on its own; the 8 waves the tile.
    const int sl = lane >> 4, j = lane & 15;
    const int rho = j >> 1, p = j & 1;
    const int s2 = wave * 4 + sl;                    // sample of the tile
    const int64_t b2 = b0 + s2;
    const bool active = rho < a.reps;
    const int rc = active ? rho : a.reps - 1;        // (spare slots compute on a copy and are masked out)
    const int t2 = rc / RPT, ap = rc - t2 * RPT;
    // requested now, consumed after the barrier: the partition's sum weights, its leaf constants, the root weights of
    // class 0 and the eligibility flags
    float w0[S][I * I];
    {
        const float *wp = a.W0 + ((int64_t)(rc * 2 + p) * S) * I * I;
#pragma unroll
        for (int o = 0; o < S; ++o)
#pragma unroll
            for (int e = 0; e < I * I; ++e) w0[o][e] = wp[o * I * I + e];
    }
    float cst[2][I];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int k = 0; k < I; ++k) cst[qq][k] = a.biasT[(p * NT + t2) * 16 + (ap * 2 + qq) * I + k];
    const int elig = a.elig[rc];
    const int M = a.reps * S * S;
    const float *wr0 = (const float *)a.Wr + rc * S * S;
    float wr_c[S * S];          // root weights of the class evaluated next
#pragma unroll
    for (int e = 0; e < S * S; ++e) wr_c[e] = wr0[e];

    // ---- partial sums of the slice -> LDS (over the wave's own, now idle, x pieces): row r, column 2 s + h -------
    {
        lfloat *mine = (lfloat *)my + (2 * s + h);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[(t * 16 + i) * RSTR] = acc[t][i];
        mine[NR * RSTR] = qsum;
        mine[(NR + 1) * RSTR] = __uint_as_float((need_exact ? 1u : 0u) | (odd_mask != 0u ? 2u : 0u) | (saw_nan ? 4u : 0u));
    }
    SM_STAMP(4);   // partials written
    // the launch's verdict on its tables (np > 0): by now the table work-groups published it long ago -- one L2 round
    // trip for thread 0, under the other waves' MFMA tails; the barrier hands it to everyone
    lunsigned *verdict_l = (lunsigned *)(smem + kGemmSmallWaves * REG + kGemmSmallWaves * 8);
    unsigned vi_ticket = 0u;
    if (np > 0 && tid == 0) {
        const bool dirty = vi_wait(pa.ctl, np, vi_ticket);
        verdict_l[0] = dirty ? 1u : 0u;
    }
    __syncthreads();
    const bool tables_stale = np > 0 && verdict_l[0] != 0u;
    SM_STAMP(5);   // barrier passed
    if (a.ablate & 2) return;

    // ---- leaf sums of the thread's two regions: eight partials each, fixed order (launches agree bit for bit).
    // Banks: a half-wave reads row r0 + 4 rho (I = 2; 8 rho for I = 4) at column 2 s2 + p: (r + column) mod 32 takes 32
    // distinct values over its 2 samples x 8 repetitions x 2 partitions with the row stride of 65 -- conflict free.
    const lfloat *part_l = (const lfloat *)smem;
    float leaf[2][I];
    {
        const lfloat *src = part_l + (t2 * 16 + ap * 2 * I) * RSTR + (2 * s2 + p);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int k = 0; k < I; ++k) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < kGemmSmallWaves; ++w) v += src[w * REGF + (qq * I + k) * RSTR];
                leaf[qq][k] = v;
            }
    }
    SM_STAMP(6);   // leaf sums
    // sum of squares and flags of the sample: 16 partials (8 slices x 2 K halves), one per slot
    const float qtot = row16_sum(part_l[(j >> 1) * REGF + NR * RSTR + 2 * s2 + (j & 1)]);
    const unsigned fl = __float_as_uint(part_l[(j >> 1) * REGF + (NR + 1) * RSTR + 2 * s2 + (j & 1)]);
    const unsigned flags = row16_or(fl);
    const unsigned odd_waves = row16_or(((fl >> 1) & 1u) << (j >> 1));   // bit w: slice w accumulated its own constants
    if (odd_waves != 0u) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int k = 0; k < I; ++k) cst[qq][k] = 0.f;
        for (int w = 0; w < kGemmSmallWaves; ++w) {
            if ((odd_waves >> w) & 1u) continue;
            const float *bs = a.biasS + ((w * 2 + p) * NT + t2) * 16 + ap * 2 * I;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int k = 0; k < I; ++k) cst[qq][k] += bs[qq * I + k];
        }
    }
    // the expanded square is within the 1e-5 bar while sum x^2 <= 36 D (|mu| <= 6: DESIGN 3.3)
    bool bad = (flags & 1u) != 0u || !(qtot <= kExpandBound * kExpandBound * (float)D) || (active && elig == 0) || tables_stale;
    SM_STAMP(7);   // flags, constants

    // ---- product + sum node of the partition (exp domain, base-2 transcendentals) -------------------------------
    float va[I], vc[I];
#pragma unroll
    for (int k = 0; k < I; ++k) {
        va[k] = leaf[0][k] + cst[0][k];
        vc[k] = leaf[1][k] + cst[1][k];
    }
    // training forward (dpk_ratspn_forward_train): the leaf / sum layer outputs relative to the sample's quadratic term; a
    // wave that turns out to need the exact evaluation overwrites its four samples' values below (same lanes, program order)
    const bool emitting = a.emit_leaf != nullptr;
    if (emitting && active && b2 < a.B) {
        float *dst = a.emit_leaf + (b2 * (4 * a.reps) + rc * 4 + 2 * p) * I;
#pragma unroll
        for (int k = 0; k < I; ++k) {
            dst[k] = va[k];
            dst[I + k] = vc[k];
        }
    }
    float ea[I], ec[I];
    const float ma = exp2_children<I>(va, ea), mc = exp2_children<I>(vc, ec);
    float n1[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) {
            float tt = 0.f;
#pragma unroll
            for (int jj = 0; jj < I; ++jj) tt = fmaf(w0[o][i * I + jj], ec[jj], tt);
            v = fmaf(ea[i], tt, v);
        }
        n1[o] = fmaf(__builtin_amdgcn_logf(v), kLn2, ma + mc);
        bad = bad || (v < 1e-30f && active);   // vanished: dominant pair under a vanishing weight
    }
    if (emitting && active && b2 < a.B) {
#pragma unroll
        for (int o = 0; o < S; ++o) a.emit_sum[(b2 * (2 * a.reps) + rc * 2 + p) * S + o] = n1[o];
    }
    SM_STAMP(8);   // node
    // ---- root: the two partitions of a repetition meet (lane ^ 1), then the repetitions of the sample ------------
    float ta[S], tc[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
        const float other = dpp_f<kDppXor1>(n1[o]);
        ta[o] = p == 0 ? n1[o] : other;
        tc[o] = p == 0 ? other : n1[o];
    }
    float ra[S], rcx[S];
    const float m2 = exp2_children<S>(ta, ra) + exp2_children<S>(tc, rcx);
    const float mr = active ? m2 : -INFINITY;
    const float mtop = row16_max(mr);
    const float mtop0 = (mtop == -INFINITY) ? 0.f : mtop;
    const float scale = (active && p == 0) ? __builtin_amdgcn_exp2f((mr - mtop0) * kL2E) : 0.f;
    const float qterm = -0.5f * qtot;
    const bool writer = j == 0 && b2 < a.B;
    double part = 0.0;
    float ll_keep[4];           // the first classes' results wait for the wave's verdict in registers
    for (int cl = 0; cl < a.C; ++cl) {
        float wr[S * S];
#pragma unroll
        for (int e = 0; e < S * S; ++e) wr[e] = wr_c[e];
        if (cl + 1 < a.C) {     // (the next class's weights travel under this class's arithmetic)
#pragma unroll
            for (int e = 0; e < S * S; ++e) wr_c[e] = wr0[(int64_t)(cl + 1) * M + e];
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float tt = 0.f;
#pragma unroll
            for (int jj = 0; jj < S; ++jj) tt = fmaf(wr[i * S + jj], rcx[jj], tt);
            v = fmaf(ra[i], tt, v);
        }
        bad = bad || (v < 1e-30f && mr > -INFINITY);
        const float tot = row16_sum(v * scale);
        const float rel = (mtop > -INFINITY) ? fmaf(__builtin_amdgcn_logf(tot), kLn2, mtop) : -INFINITY;
        const float ll = rel + qterm;
        if (emitting && writer) a.emit_out[b2 * a.C + cl] = rel;
        if (cl < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == cl) ll_keep[q] = ll;
        } else if (writer) {
            a.out[b2 * a.C + cl] = ll;      // (an exact verdict rewrites these below: same lane, program order)
        }
        part += (double)ll;
    }
    SM_STAMP(9);   // classes
    // ---- the wave's verdict: a sample outside the fast path's envelope -> its 4 samples evaluated exactly ---------
    if (__any(bad)) {
        LseScratch sc{reinterpret_cast<float *>(smem_generic + wave * REG + SLOT * 4) + lane * (2 * NMAX)};
        const float *xr = a.x + (b2 < a.B ? b2 : a.B - 1) * D;
        part = 0.0;
        small_exact_wave<I, S>(a, xr, b2 < a.B ? b2 : (int64_t)-1, rc, p, active, sc, tables_stale ? pa.w[0] : nullptr,
                               tables_stale ? pa.w[1] : nullptr, [&](int cl, float ll) {
            if (writer) {
                a.out[b2 * a.C + cl] = ll;
                if (emitting) a.emit_out[b2 * a.C + cl] = ll;
            }
            part += (double)ll;
        });
    } else if (writer) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < a.C) a.out[b2 * a.C + q] = ll_keep[q];
    }
    SM_STAMP(11);  // stored
    if (a.ll_sum != nullptr) {
        // {sum of LLs, count}: one atomic per work-group (and one for the count per launch)
        double *red_l = reinterpret_cast<double *>(smem_generic + kGemmSmallWaves * REG);
        part = wave_reduce_sum(writer ? part : 0.0);
        if (lane == 0) red_l[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kGemmSmallWaves; ++w) tot += red_l[w];
            atomicAdd(a.ll_sum + (a.ll_cnt > 1 ? ((int)blockIdx.x & 15) : 0), tot);
            if ((int)blockIdx.x == np) atomicAdd(a.ll_sum + a.ll_cnt, (double)a.B * (double)a.C);
        }
    }
    if ((flags & 4u) != 0u && lane == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
    if (np > 0 && tid == 0) vi_done(pa.ctl, vi_ticket, pa.readers);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool gemm_small_shape_ok(int D, int NT) {
    return NT <= 2 && cdiv(cdiv(D, 16), kGemmSmallWaves) <= kGemmSmallMaxK;
}

// samples per launch up to which the small-batch kernels are taken (dpk_ratspn_small_batch_max; DPK_GEMM_SMALL_MAX in
// the environment sets the initial value, 0 switches them off)
constexpr int64_t kSmallBatchDefault = 16384;
static int64_t small_batch_initial() {
    const char *e = getenv("DPK_GEMM_SMALL_MAX");
    return e ? (int64_t)atoll(e) : kSmallBatchDefault;
}
static int64_t &small_batch_max_ref() {
    static int64_t v = small_batch_initial();
    return v;
}
int64_t gemm_small_max_batch() { return small_batch_max_ref(); }

template <int I, int S, int NT, int MAXK>
static int gemm_small_launch(const GemmArgs &a, const GemmPrepArgs &p, hipStream_t st) {
    size_t lds = (size_t)kGemmSmallWaves * gemm_small_region_bytes(NT, MAXK) + kGemmSmallWaves * 8 + 16;
    if (p.np > 0 && gemm_prep_lds_bytes(a.D, I, a.d) > lds) lds = gemm_prep_lds_bytes(a.D, I, a.d);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "ratspn_gemm_small: %zu bytes of LDS", lds);
    auto kern = ratspn_gemm_small_kernel<I, S, NT, MAXK>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
#ifdef DPK_TIMELINE
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, (size_t)256 * kGemmSmallWaves * 16 * 8);
        const_cast<GemmArgs &>(a).dbg = dbg;
        FILE *f = fopen("/tmp/dpk_timeline_small_ptr.txt", "w");
        if (f) { fprintf(f, "%p %d\n", (void *)dbg, cdiv(a.B, 32)); fclose(f); }
    }
#endif
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_FUSED);
    if (ev0) (void)hipEventRecord(ev0, st);
    GemmPrepArgs pp = p;
    pp.readers = p.np + (int)cdiv(a.B, 32);
    DPK_LAUNCH(kern, dim3(p.np + cdiv(a.B, 32)), dim3(kGemmSmallWaves * 64), lds, st, a, pp);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_gemm_small_kernel");
    return DPK_OK;
}

template <int I, int S, int NT>
static int gemm_small_launch_k(const GemmArgs &a, const GemmPrepArgs &p, hipStream_t st) {
    // K-steps of 16 features per wave: the smallest instantiation that holds the slice
    const int per_wave = cdiv(cdiv(a.D, 16), kGemmSmallWaves);
    if (per_wave <= 4) return gemm_small_launch<I, S, NT, 4>(a, p, st);
    if (per_wave <= 7) return gemm_small_launch<I, S, NT, 7>(a, p, st);
    return gemm_small_launch<I, S, NT, 8>(a, p, st);
}

// The caller (ratspn_gemm_forward) has built the tables and filled the argument block.
int ratspn_gemm_small_forward(const GemmArgs &a, const GemmPrepArgs &p, int reps, int I, int S, int NT, hipStream_t st) {
#define DPK_SMALL(II, SS)                                                 \
    if (I == II && S == SS)                                               \
        return NT == 1 ? gemm_small_launch_k<II, SS, 1>(a, p, st) : gemm_small_launch_k<II, SS, 2>(a, p, st)
    DPK_SMALL(2, 2);
    DPK_SMALL(2, 4);
    DPK_SMALL(4, 2);
    DPK_SMALL(4, 4);
#undef DPK_SMALL
    set_error("ratspn_gemm_small: (channels=%d, sums=%d) not built", I, S);
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk

extern "C" int64_t dpk_ratspn_small_batch_max(int64_t samples) {
    int64_t &v = dpk::small_batch_max_ref();
    const int64_t prev = v;
    v = samples < 0 ? dpk::small_batch_initial() : samples;
    return prev;
}
