// RAT-SPN fused forward with the leaf layer on the matrix cores (depth 2, unit-scale Gaussian leaves).
//
// reference: RegionGraphLayer.forward + GaussianLayer (deeprob/spn/layers/ratspn.py:87-108, :160-213),
// ProductLayer :272-286, SumLayer :363-378, RootLayer :446-458, chained by RatSpn.forward
// (deeprob/spn/models/ratspn.py:105-122).
//
// With sigma == 1 the leaf sum of region r, channel k is
//     sum_f [f in r] ( x_f mu_rkf - mu_rkf^2/2 - log sqrt(2 pi) )  -  1/2 sum_f [f in r] x_f^2 .
// The first part is a GEMM  P[b, n] = sum_f x[b, f] M[f, n]  over the n = (repetition, region, channel)
// columns, with M[f, n] = mu where variable f belongs to the region of column n and 0 elsewhere (one
// region per repetition holds f: M is 1/4 dense at depth 2).  The second part is common to the channels
// of a region, hence factors out of every sum node above it; every repetition covers each variable once,
// so what reaches the root is -1/2 sum_f x_f^2: one scalar per sample.
//
// The GEMM runs on v_mfma_f32_32x32x16_f16 with both operands split into two f16 halves (x = xh + xl,
// mu = mh + ml, each half carrying 11 significant bits): x mu ~= mh xh + mh xl + ml xh, accumulated in
// fp32; the dropped ml xl term is below 2^-22 |x mu|, i.e. the product keeps fp32 accuracy.  Three f16
// MFMAs cost 3/16 of one fp32 MFMA, so even the dense form of the 1/4-dense matrix is ~5x cheaper than
// the VALU form -- and, unlike it, leaves the kernel bound by the HBM stream of x.
//
// The constants: a chunk of 64 features without marginalised evidence adds its per-column sums of
// -(mu^2/2 + log sqrt(2 pi)) ready-made (prepared table).  In a chunk that holds NaN evidence the NaN
// entries become 0 and the constants of the OBSERVED variables only are accumulated by a second GEMM, a 0/1
// validity indicator against the table of negated constants -- marginalisation is exact in the same form,
// nothing is added and subtracted again, and an all-NaN row comes out as exactly 0 like the reference's.
// Anything the expanded square cannot carry within the 1e-5 bar (+-inf or huge evidence, a sample whose
// sum of squares is large, means beyond kExpandBound, non-unit scales) sends the 32 samples of the wave
// through an exact per-element evaluation (gemm_exact_wave) -- correctness never depends on the hint.
//
// Mapping: a work-group of 4 waves owns 128 samples, a wave 32 of them.  The x tile streams through LDS
// in chunks of 64 features by LDS-DMA (global_load_lds_dwordx4: full 256-byte row segments from HBM, the
// 16-byte pieces XOR-swizzled on the SOURCE side so that the MFMA-shaped ds_read_b128 is conflict free),
// three stages deep, counted vmcnt + raw s_barrier (no drain across the barrier).  The mean-table chunk
// (16 KB of ready-made MFMA A-fragments, L2 resident) rides in the same stages.  The MFMA computes
// P^T = M^T x^T, so a lane ends up holding, for ONE sample, the 16 columns of half a repetition set:
// lane l (sample l & 31, half h = l >> 5) owns regions {2h, 2h+1} of every repetition, evaluates that
// partition's product + sum nodes in registers, swaps the S outputs with lane l ^ 32 and finishes the
// root.  Work-groups are persistent (grid = min(tiles, CUs)); the DMA ring runs across tile boundaries.
#include "common.h"
#include "ratspn_gemm_fused.h"
#include "ratspn_gemm_prep.h"
#include <math.h>
#include <stdlib.h>

namespace dpk {

// ------------------------------------------------------------------------------------------------
// tables, rebuilt from the live parameters (ratspn_gemm_prep.h): one work-group per repetition slot + the softmax rows
// behind them.  This is the stand-alone launch (kPrepBuild / kPrepVerify); the small-batch and 8-channel kernels run the
// same work-groups inside their own launch (kPrepInline).
// ------------------------------------------------------------------------------------------------
template <int I>
__global__ __launch_bounds__(kGemmPrepThreads) void ratspn_gemm_prep_kernel(const GemmPrepArgs a) {
    extern __shared__ int prep_dyn[];
    if (a.mode == kPrepBuild && blockIdx.x == 0 && threadIdx.x == 0) {   // (no launch of this module is in flight: stream order)
        a.ctl->word = 0ull;
        a.ctl->readers = 0u;
    }
    gemm_prep_block<I>(a, (int)blockIdx.x, prep_dyn);
}

// The ring kernel below is kept exactly as round 2 left it, in a namespace of its own with its own argument block and
// its own copy of the upper layers: its hand-balanced schedule (one compute wave per SIMD, LDS-DMA ring, counted waits)
// moved by 3-7 us of 46 when the upper layers were factored into a function shared with the other two mappings, when a
// template flag for the marginalised-evidence variant was threaded through it, and by 0.7 us when three fields were
// appended to its argument block (same-box A/B runs, round 3).  The small-batch kernel (ratspn_gemm_small.hip) and the
// marginalised-evidence variant (ratspn_gemm_nan.hip) share ratspn_gemm_fused.h instead.
namespace ring {

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float *x;
    int64_t B;
    int D, d, reps, C, NCH, ntiles;
    const uint16_t *mtab, *ctab;
    const float *biasT;   // [2][NT][16] whole-row constants in the accumulator order of a lane
    const float *biasC;   // [NCH][2][NT][16] the same per chunk (tiles with marginalised evidence)
    const int *elig;
    const float *W0;   // [reps*2][S][I*I] linear softmax weights (copied into LDS)
    const float *LW0;  // log-softmax weights (exact fallback of a node, exact evaluation)
    cfloat_p Wr, LWr;  // [C][reps*S*S]
    float *out;
    double *ll_sum;
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
__device__ __noinline__ void gemm_exact_wave(const GemmArgs &a, int64_t bw0, int lane, LseScratch sc) {
    constexpr int RPT = 8 / I;
    constexpr int RH = (NT * RPT + 1) / 2;   // repetitions per lane half
    const int s = lane & 31, h = lane >> 5;
    const int64_t b = bw0 + s;
    const bool valid = b < a.B;
    const float *xr = a.x + (valid ? b : a.B - 1) * a.D;
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

#ifdef DPK_TIMELINE
#define GEMM_STAMP(row, slot) do { __builtin_amdgcn_sched_barrier(0); if (a.dbg && lane == 0 && !loader && (row) < 64) a.dbg[(((int64_t)blockIdx.x * kGemmWaves + wave) * 64 + (row)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GEMM_STAMP(row, slot) do { } while (0)
#endif

template <int I, int S, int NT>
__global__ __launch_bounds__(2 * kGemmWaves * 64) void ratspn_gemm_kernel(const GemmArgs a) {
    constexpr int RPT = 8 / I;                           // repetitions per column tile
    constexpr int KS = gemm_ks(NT);
    constexpr int KC = 16 * KS;                          // features per chunk
    constexpr int W = 4 * KS;                            // 16-byte pieces per staged row
    constexpr int ROWB = KC * 4;
    constexpr int RPI = 64 / W;                          // rows per x DMA instruction
    constexpr int SWS = (W == 16) ? 0 : (W == 8 ? 1 : 2);  // swizzle: piece ^= (row >> SWS) & (W-1)
    constexpr int XB = kGemmTile * ROWB;                 // x chunk bytes
    constexpr int BB = KS * NT * 2 * 1024;               // mean-table bytes per chunk
    constexpr int STAGE = XB + BB;
    constexpr int NS = kGemmStages;
    constexpr int PX = 32 / RPI;                         // x DMA instructions per loader wave and chunk
    constexpr int PB = BB / (kGemmWaves * 1024);         // table DMA instructions per loader wave and chunk
    constexpr int P = PX + PB;                           // DMA instructions per loader wave and chunk
    static_assert(BB % (kGemmWaves * 1024) == 0, "table chunk must split over the waves");
    static_assert(NS == 3 && P <= 63, "the counted waits leave exactly one chunk in flight");
    constexpr int NMAX = (I > S ? I : S);
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) const half8 lh8;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    lfloat *bias_l = (lfloat *)(smem + NS * STAGE);                  // [2][NT][16] constants of a whole row
    lfloat *w0_l = bias_l + 2 * NT * 16;                             // [reps*2][S*I*I]
    float *scr_l = reinterpret_cast<float *>(smem_generic + NS * STAGE) + 2 * NT * 16 +
                   a.reps * 2 * S * I * I;                           // [256][2*NMAX] exact_lse scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Waves 0-3 compute (one per SIMD), waves 4-7 only feed the LDS ring: wave 4+w copies the 32 rows of wave w and a
    // quarter of the mean-table chunk.  A compute wave never issues a DMA (an LDS-DMA instruction costs its wave
    // 60-150 issue cycles), a loader never touches a VALU; the two meet at one s_barrier per chunk.
    const bool loader = wave8 >= kGemmWaves;
    const int wave = wave8 & (kGemmWaves - 1);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D, NCH = a.NCH;
    GEMM_STAMP(63, 0);
#ifdef DPK_TIMELINE
    if (a.dbg && lane == 0 && !loader) a.dbg[(((int64_t)blockIdx.x * kGemmWaves + wave) * 64 + 63) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif

    // this work-group's tiles: blockIdx.x, + gridDim.x, ... (persistent); every counter below is wave-uniform
    const int grid = (int)gridDim.x;
    const int ntiles = a.ntiles;

    double red_ll = 0.0;
    bool saw_nan_any = false;
    if (loader) {
        gemm_loader_run<KS, PB>(a.x, a.B, D, NCH, ntiles, (int)blockIdx.x, grid, (gcchar_p)a.mtab, BB, wave * PB,
                                (unsigned)(uintptr_t)smem, STAGE, wave, lane);
    } else {
    // ================================================ compute waves =========================================
    // constants into LDS
    for (int e = tid; e < 2 * NT * 16; e += kGemmWaves * 64) bias_l[e] = a.biasT[e];
    for (int e = tid; e < a.reps * 2 * S * I * I; e += kGemmWaves * 64) w0_l[e] = a.W0[e];
    bool model_ok = true;
    for (int e = lane; e < NT * RPT; e += 64) model_ok = model_ok && (a.elig[e] != 0);
    model_ok = __all(model_ok);
    LseScratch sc{scr_l + tid * (2 * NMAX)};
    __syncthreads();

    // LDS byte offsets (within a stage) of the lane's x pieces and of its table fragments
    const int rl_own = wave * 32 + s;
    const int sw = (rl_own >> SWS) & (W - 1);
    unsigned xoff[2 * KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int pcs = ks * 4 + h * 2;
        xoff[2 * ks] = (unsigned)(rl_own * ROWB + ((pcs ^ sw) << 4));
        xoff[2 * ks + 1] = (unsigned)(rl_own * ROWB + (((pcs | 1) ^ sw) << 4));
    }
    const unsigned foff = (unsigned)(XB + lane * 16);

    GEMM_STAMP(63, 1);
    [[maybe_unused]] int grow = 0;   // timeline row = chunk count of this work-group
    bool saw_nan = false;
    double ll_part = 0.0;   // this lane's share of the sum of the LLs written by the fast path (all tiles)
    int cstage = 0;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
        gf32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        float qsum = 0.f;
        bool need_exact = false;
        unsigned odd_mask = 0u;   // chunks that met NaN evidence: their constants were accumulated by the validity GEMM
        for (int c = 0; c < NCH; ++c) {
            GEMM_STAMP(grow, 0);
            GEMM_STAMP(grow, 1);
            gemm_lds_barrier();   // the loaders have seen this chunk land; everyone is done reading the previous one
            GEMM_STAMP(grow, 2);
            const lchar *st = smem + cstage * STAGE;
            cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
            if (a.ablate & 1) continue;
            const lchar *tb = st + foff;
            // the chunk's table fragments and the lane's 8*KS values: every LDS read of the chunk is issued up front
            half8 mh[KS][NT], ml[KS][NT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    mh[ks][t] = *(lh8 *)(tb + (ks * NT + t) * 2048);
                    ml[ks][t] = *(lh8 *)(tb + (ks * NT + t) * 2048 + 1024);
                }
            // the lane's 8*KS values of this chunk
            float v[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const gf32x4 x0 = *(lf4 *)(st + xoff[2 * ks]);
                const gf32x4 x1 = *(lf4 *)(st + xoff[2 * ks + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[ks][i] = x0[i];
                    v[ks][4 + i] = x1[i];
                }
            }
            const bool partial = (c + 1) * KC > D;
            if (partial) {   // last chunk: slots beyond D hold clamped copies (or nothing this chunk wrote)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int f0 = c * KC + ks * 16 + h * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[ks][i] = (f0 + i < D) ? v[ks][i] : 0.f;
                }
            }
            gf32x2 tq2 = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const gf32x2 pv = {v[ks][i], v[ks][i + 1]};
                    tq2 = __builtin_elementwise_fma(pv, pv, tq2);
                }
            float tq = tq2[0] + tq2[1];
            // NaN / +-inf / huge evidence anywhere in the wave's share of the chunk?
            const bool odd_chunk = __any(!(tq < kGemmStepBound));
            if (!odd_chunk && !partial) {
                // ---- hot path: clean, complete chunk -------------------------------------------------------
                qsum += tq;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    half8 xh, xl;
                    split8(v[ks], xh, xl);
                    // independent accumulators alternate (a dependent 32x32x16 chain would stall on its own latency)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[ks][t], xh, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[ks][t], xl, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[ks][t], xh, acc[t], 0, 0, 0);
                }
            } else {
                // ---- generic path: last (partial) chunk, or NaN / inf / huge evidence ----------------------------
                const int nks = min(KS, (D - c * KC + 15) >> 4);
                half8 valid[KS];
                if (odd_chunk) {
                    // NaN (marginalised) entries count as 0 and drop out of the constants (validity indicator below);
                    // +-inf / huge entries send the wave through the exact evaluation at the end of the tile
                    odd_mask |= 1u << c;
                    tq = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float vi = v[ks][i];
                            const bool isn = vi != vi;
                            const bool big = !isn && !(fabsf(vi) < kGemmAbsBound);
                            need_exact = need_exact || big;
                            saw_nan = saw_nan || isn;
                            v[ks][i] = (isn || big) ? 0.f : vi;
                            valid[ks][i] = isn ? (_Float16)0.0f : (_Float16)1.0f;
                            tq = fmaf(v[ks][i], v[ks][i], tq);
                        }
                }
                qsum += tq;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks < nks) {
                        half8 xh, xl;
                        split8(v[ks], xh, xl);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[ks][t], xh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[ks][t], xl, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[ks][t], xh, acc[t], 0, 0, 0);
                        }
                        if (odd_chunk) {
                            // - (mu^2/2 + log sqrt(2 pi)) of the variables that ARE observed (table of negated constants)
                            typedef const __attribute__((address_space(1))) half8 gh8;
                            const gcchar_p cb = (gcchar_p)a.ctab + ((((int64_t)(c * KS + ks) * NT) * 2) * 512 + lane * 8) * 2;
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const half8 ch = *(gh8 *)(cb + t * 2048);
                                const half8 cl = *(gh8 *)(cb + t * 2048 + 1024);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, valid[ks], acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, valid[ks], acc[t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            GEMM_STAMP(grow, 4);
            ++grow;
        }
        GEMM_STAMP(grow - 1, 5);
        if (!(a.ablate & 8)) {
            // ---- upper layers of the tile ------------------------------------------------------------
            const int64_t b0 = (int64_t)tile * kGemmTile;
            const int64_t bw0 = b0 + wave * 32;
            const int64_t b = bw0 + s;
            const float qtot = qsum + __shfl_xor(qsum, 32, 64);
            // the expanded square is within the 1e-5 bar while sum x^2 <= 36 D (|mu| <= 6: DESIGN 3.3)
            const bool lane_exact = need_exact || !(qtot <= kExpandBound * kExpandBound * (float)D);
            if (!model_ok || __any(lane_exact)) {
                // (a private copy: handing the kernel argument block itself to a call would move it, and with it
                // every loop counter derived from it, out of the scalar registers)
                const GemmArgs ac = a;
                gemm_exact_wave<I, S, NT>(ac, bw0, lane, sc);
            } else {
                // per-column constants: the whole-row sums, or -- after chunks with marginalised evidence, whose
                // constants the validity GEMM accumulated -- the sums of the clean chunks only
                float cst[NT][16];
                if (odd_mask == 0u) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const gf32x4 q4 = *(lf4 *)(bias_l + (h * NT + t) * 16 + 4 * i);
#pragma unroll
                            for (int j = 0; j < 4; ++j) cst[t][4 * i + j] = q4[j];
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) cst[t][i] = 0.f;
                    for (int c = 0; c < NCH; ++c) {
                        if ((odd_mask >> c) & 1u) continue;
                        const float *bc = a.biasC + ((c * 2 + h) * NT) * 16;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int i = 0; i < 16; ++i) cst[t][i] += bc[t * 16 + i];
                    }
                }
                // Upper layers in the exp domain on the hardware's base-2 transcendentals; a node whose scaled sum
                // vanishes (dominant pair under a vanishing weight) is redone exactly, out of line (gemm_node_exact).
                constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
                // phase A, branch free so that the independent nodes interleave (one wave per SIMD: a dependent chain of
                // transcendentals would otherwise run at its latency): every product + sum node of the lane's partitions
                GEMM_STAMP(grow - 1, 0);
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
                GEMM_STAMP(grow - 1, 1);
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
                GEMM_STAMP(grow - 1, 7);
                // a vanished node anywhere in the wave: the wave's samples go through the exact evaluation instead
                // (rare: a softmax weight below e^-69 on the dominant pair)
                if (__any(vanished)) {
                    const GemmArgs ac = a;
                    gemm_exact_wave<I, S, NT>(ac, bw0, lane, sc);
                    continue;
                }
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
                if (__any(vanished)) {   // (the exact evaluation overwrites what this wave stored and adds its own sum)
                    const GemmArgs ac = a;
                    gemm_exact_wave<I, S, NT>(ac, bw0, lane, sc);
                    continue;
                }
                GEMM_STAMP(grow - 1, 6);
                ll_part += part;
            }
            // (measured: sending the sums of all tiles but the last as one atomic per wave here, while the work-group
            // is still streaming, costs 8 us per launch -- the compute waves' next LDS-DMA-fed chunk waits behind it)
        }
    }
    GEMM_STAMP(63, 2);
    red_ll = wave_reduce_sum(ll_part);
    saw_nan_any = saw_nan;
    }   // compute waves
    // {sum LL, count}: one atomic per work-group, issued when no counted wait is left to trip over it (an atomic is a
    // VMEM operation: inside the ring it would sit in every wave's vmcnt until the L2 has serialised thousands of them)
    if (a.ll_sum != nullptr && !(a.ablate & 16)) {
        double *red = reinterpret_cast<double *>(smem_generic);   // the stages are idle now
        __syncthreads();
        if (lane == 0 && !loader) red[wave] = red_ll;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kGemmWaves; ++w) tot += red[w];
            atomicAdd(a.ll_sum, tot);
            // every sample of the launch is evaluated by exactly one path: the count needs no per-work-group atomic
            // (256 same-address fp64 atomics at the very end of the kernel cost it 1.5 us)
            if (blockIdx.x == 0) atomicAdd(a.ll_sum + 1, (double)a.B * (double)a.C);
        }
    }
    if (saw_nan_any && lane == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
    GEMM_STAMP(63, 3);
#ifdef DPK_TIMELINE
    if (a.dbg && lane == 0 && !loader) a.dbg[(((int64_t)blockIdx.x * kGemmWaves + wave) * 64 + 63) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace ring

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int I, int S, int NT>
static int gemm_launch(const ring::GemmArgs &a, int reps, hipStream_t st) {
    constexpr int KS = gemm_ks(NT);
    constexpr int BB = KS * NT * 2 * 1024;
    constexpr int NMAX = (I > S ? I : S);
    const size_t lds = (size_t)kGemmStages * (kGemmTile * 64 * KS + BB) +
                       (size_t)(2 * NT * 16 + reps * 2 * S * I * I) * 4 + (size_t)kGemmWaves * 64 * 2 * NMAX * 4;
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "ratspn_gemm: %zu bytes of LDS", lds);
    auto kern = ring::ratspn_gemm_kernel<I, S, NT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    const int grid = a.ntiles < cus ? a.ntiles : cus;
#ifdef DPK_TIMELINE
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, (size_t)1024 * kGemmWaves * 64 * 8 * 8);
        const_cast<ring::GemmArgs &>(a).dbg = dbg;
        FILE *f = fopen("/tmp/dpk_timeline_ptr.txt", "w");
        if (f) { fprintf(f, "%p %d %d\n", (void *)dbg, grid, a.NCH); fclose(f); }
    }
#endif
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_FUSED);
    if (ev0) (void)hipEventRecord(ev0, st);
    DPK_LAUNCH(kern, dim3(grid), dim3(2 * kGemmWaves * 64), lds, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_gemm_kernel");
    return DPK_OK;
}

template <int I, int S>
static int gemm_dispatch_nt(const ring::GemmArgs &a, int reps, int NT, hipStream_t st) {
    switch (NT) {
        case 1: return gemm_launch<I, S, 1>(a, reps, st);
        case 2: return gemm_launch<I, S, 2>(a, reps, st);
        case 3: return gemm_launch<I, S, 3>(a, reps, st);
        case 4: return gemm_launch<I, S, 4>(a, reps, st);
    }
    set_error("ratspn_gemm: %d column tiles not built", NT);
    return DPK_EUNSUPPORTED;
}

// ratspn_gemm_small.hip: 32-sample tiles, features split over the waves (small batches);
// ratspn_gemm_nan.hip: the ring kernel's variant for marginalised evidence
bool gemm_small_shape_ok(int D, int NT);
int64_t gemm_small_max_batch();
int ratspn_gemm_small_forward(const GemmArgs &a, const GemmPrepArgs &p, int reps, int I, int S, int NT, hipStream_t st);
bool gemm_marginal_shape_ok(int D, int NT);
// ratspn_gemm_wide.hip: 8-channel models, a wave per repetition
bool gemm_wide_shape_ok(int D, int reps, int I, int S, int C);
int ratspn_gemm_wide_forward(const GemmArgs &a, const GemmPrepArgs &p, int S, hipStream_t st);
bool gemm_wide_takes_tile32(int64_t B, int D, int reps, int C, bool marginal);
int ratspn_gemm_marginal_forward(const GemmArgs &a, int reps, int I, int S, int NT, hipStream_t st);

// The caller (dpk_ratspn_forward) has validated the arguments and carved the workspace.
int ratspn_gemm_forward(const RatWs &w, const float *x, int64_t B, int D, const int64_t *mask, const uint8_t *pad,
                        const float *loc, const float *scale, const float *sum_weight0, const float *root_weight,
                        int reps, int I, int S, int C, float *out, double *ll_sum, uint32_t flags, hipStream_t st) {
    const int d = (D + (4 - D % 4) % 4) / 4;
    const int NT = w.g_nt;
    int *slow_word = nullptr;
    int launch_seq = 0;
    bool marginal = slow_hint_next(&slow_word, &launch_seq);
    static const int ablate = [] { const char *e = getenv("DPK_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    {   // (measurement: DPK_GEMM_MARGINAL=0 / 1 pins the variant)
        static const int force = [] { const char *e = getenv("DPK_GEMM_MARGINAL"); return e ? atoi(e) : -1; }();
        if (force >= 0) marginal = force != 0;
    }
    // Three mappings of the same arithmetic: below ~one 128-sample tile per compute unit the persistent ring kernel
    // runs at the latency of its chunk walk and 32-sample tiles with the feature axis split over the waves take over
    // (ratspn_gemm_small.hip); while recent launches met NaN evidence the ring variant that stages both tables runs
    // (ratspn_gemm_nan.hip); otherwise the ring kernel below.
    const bool wide = I == 8;
    const bool small = !wide && B <= gemm_small_max_batch() && gemm_small_shape_ok(D, NT);
    // The tables.  "Believed current -- check" (DPK_FLAG_PARAMS_VERIFY) costs the 32-sample mappings nothing extra: their
    // launch carries the table work-groups itself (kPrepInline, ratspn_gemm_prep.h).  The ring kernels keep the
    // stand-alone check launch in front of them (DPK_VERIFY_INLINE=0 forces it everywhere: A/B measurements).
    static const bool inline_allowed = [] { const char *e = getenv("DPK_VERIFY_INLINE"); return !(e && atoi(e) == 0); }();
    GemmPrepArgs p{};
    p.mask = mask; p.pad = pad; p.loc = loc; p.scale = scale;
    p.D = D; p.d = d; p.reps = reps; p.NT = NT; p.NKSP = w.g_nksp; p.KS = gemm_ks(NT);
    p.mtab = w.gm_tab; p.ctab = w.gc_tab; p.bias = w.gbias; p.bias_row = w.gbias_row; p.bias_ks = w.gbias_ks;
    p.bias_sl = w.gbias_sl; p.elig = w.gelig;
    p.w[0] = sum_weight0; p.W[0] = w.w[0]; p.LW[0] = w.lw[0]; p.rows[0] = reps * 2 * S; p.n[0] = I * I;
    p.w[1] = root_weight; p.W[1] = w.w[2]; p.LW[1] = w.lw[2]; p.rows[1] = C; p.n[1] = reps * S * S;
    p.hash = w.ghash; p.ctl = w.gctl;
    p.upfrag = (I == 8 && S >= 2) ? w.gup : nullptr; p.up_S = S;
    const int np = gemm_prep_blocks(NT, I, p.rows[0] + p.rows[1], kGemmPrepThreads);
    const size_t prep_lds = gemm_prep_lds_bytes(D, I, d);
    DPK_REQUIRE(prep_lds <= 60 * 1024, DPK_EUNSUPPORTED, "ratspn_gemm: in_features=%d too large for the table kernel", D);
    const bool verify_inline = inline_allowed && (flags & DPK_FLAG_PARAMS_VERIFY) && !(flags & DPK_FLAG_PARAMS_CACHED) &&
                               ((wide && gemm_wide_takes_tile32(B, D, reps, C, marginal)) || small);
    if (verify_inline) {
        p.mode = kPrepInline;
        p.np = np;
    } else if (!(flags & DPK_FLAG_PARAMS_CACHED)) {
        p.mode = (flags & DPK_FLAG_PARAMS_VERIFY) ? kPrepVerify : kPrepBuild;
        if (I == 2) DPK_LAUNCH(ratspn_gemm_prep_kernel<2>, dim3(np), dim3(kGemmPrepThreads), prep_lds, st, p);
        else if (I == 4) DPK_LAUNCH(ratspn_gemm_prep_kernel<4>, dim3(np), dim3(kGemmPrepThreads), prep_lds, st, p);
        else DPK_LAUNCH(ratspn_gemm_prep_kernel<8>, dim3(np), dim3(kGemmPrepThreads), prep_lds, st, p);
        DPK_CHECK_LAUNCH("ratspn_gemm_prep_kernel");
        p.np = 0;
    }
    if (wide || small || (marginal && gemm_marginal_shape_ok(D, NT))) {
        GemmArgs a{};
        a.x = x; a.B = B; a.D = D; a.d = d; a.reps = reps; a.C = C;
        a.NCH = cdiv(D, 16 * gemm_ks(NT));
        a.ntiles = cdiv(B, kGemmTile);
        a.mtab = w.gm_tab; a.ctab = w.gc_tab; a.biasT = w.gbias_row; a.biasC = w.gbias; a.elig = w.gelig;
        a.biasK = w.gbias_ks; a.biasS = w.gbias_sl;
        a.W0 = w.w[0]; a.LW0 = w.lw[0]; a.Wr = as_const(w.w[2]); a.LWr = as_const(w.lw[2]);
        a.out = out; a.ll_sum = ll_sum;
        a.mask = mask; a.pad = pad; a.loc = loc; a.scale = scale;
        a.slow_flag = slow_word; a.launch_seq = launch_seq; a.marginal = marginal ? 1 : 0;
        a.ablate = ablate;
        a.upfrag = p.upfrag;
        if (wide) return ratspn_gemm_wide_forward(a, p, S, st);
        if (small) return ratspn_gemm_small_forward(a, p, reps, I, S, NT, st);
        return ratspn_gemm_marginal_forward(a, reps, I, S, NT, st);
    }
    ring::GemmArgs a{};
    a.x = x; a.B = B; a.D = D; a.d = d; a.reps = reps; a.C = C;
    a.NCH = cdiv(D, 16 * gemm_ks(NT));
    a.ntiles = cdiv(B, kGemmTile);
    a.mtab = w.gm_tab; a.ctab = w.gc_tab; a.biasT = w.gbias_row; a.biasC = w.gbias; a.elig = w.gelig;
    a.W0 = w.w[0]; a.LW0 = w.lw[0]; a.Wr = as_const(w.w[2]); a.LWr = as_const(w.lw[2]);
    a.out = out; a.ll_sum = ll_sum;
    a.mask = mask; a.pad = pad; a.loc = loc; a.scale = scale;
    a.slow_flag = slow_word; a.launch_seq = launch_seq;
    a.ablate = ablate;
    if (I == 2) {
        if (S == 2) return gemm_dispatch_nt<2, 2>(a, reps, NT, st);
        return gemm_dispatch_nt<2, 4>(a, reps, NT, st);
    }
    if (S == 2) return gemm_dispatch_nt<4, 2>(a, reps, NT, st);
    return gemm_dispatch_nt<4, 4>(a, reps, NT, st);
}

}  // namespace dpk
