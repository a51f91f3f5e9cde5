// RAT-SPN fused forward, third mapping of the matrix-core route: persistent 32-sample blocks, the FEATURE axis split
// over seven waves that keep their slice of the mean table in REGISTERS for the whole launch.
//
// reference: RatSpn.forward (deeprob/spn/models/ratspn.py:105-122) = RegionGraphLayer.forward + GaussianLayer
// (deeprob/spn/layers/ratspn.py:87-108, :160-213), ProductLayer :272-286, SumLayer :363-378, RootLayer :446-458.
//
// Same formulation, tables and arithmetic as ratspn_gemm.hip (read its header first).  Why a third mapping: the ring
// kernel there streams 48 KB stages of which a third is the mean-table chunk, re-fetched from L2 for every 128-sample
// tile by the same four loader waves that fetch x (its ring floor, 38.8 us at 65 536 samples, is those waves' DMA issue
// rate), and a tile's upper layers run with the ring stalled (+4.6 us).  Here
//   * the table never moves: 49 K-steps of 16 features = 7 waves x 7 K-steps; a wave holds its 7 x NT x {hi, lo}
//     A-fragments (112 VGPRs at NT = 2) from the prologue to the end of the launch;
//   * x is the only stream.  A block is 32 CONSECUTIVE rows = one contiguous 100 KB range of x.  Wave w copies ITS OWN
//     slice (rows x features [112 w, 112 w + 112)) into its private 14 KB of LDS by LDS-DMA (14 instructions, four
//     consecutive lanes fetching 64 contiguous bytes of a row, 16-byte pieces XOR-swizzled on the source side so that the
//     MFMA-shaped ds_read_b128 is conflict free -- the small-batch kernel's layout), reads it into registers, and
//     re-issues the DMA of its next block BEFORE it converts and multiplies: producer and consumer of a slot are the same
//     wave, so the x stream needs no barrier at all -- a counted vmcnt is the whole protocol -- and 98 KB per compute
//     unit are in flight while the matrix cores work;
//   * the seven partial accumulators of a block meet in LDS (56 KB) and the EIGHTH wave adds them in a fixed order and
//     evaluates the upper layers (gemm_upper_fast, the ring kernel's code) while the seven are already on the next
//     block: the upper layers leave the critical path.  Two s_barriers per block hand the partial buffer back and forth.
// A block outside the fast path's envelope (NaN / +-inf / huge evidence, large sum of squares, model outside the expanded
// square's bound, vanished sum node) is evaluated exactly by the eighth wave (gemm_exact_body) -- correctness never
// depends on a hint; launches that meet NaN evidence set the hint that routes the following ones to the variant built
// for marginalised evidence (ratspn_gemm_nan.hip).
//
// Results agree with the other two mappings to fp32 rounding, not bit for bit (seven partial sums in a fixed order).
#include "ratspn_gemm_fused.h"
#include "ratspn_gemm_prep.h"
#include <stdlib.h>
#include <algorithm>

namespace dpk {

constexpr int kSliceCompute = 7;                 // waves that own a slice of the feature axis
constexpr int kSliceKW = 7;                      // K-steps of 16 features per slice (registers: 4 NT VGPRs x 2 per K-step)
constexpr int kSliceThreads = (kSliceCompute + 1) * 64;
constexpr int kSliceSlot = kSliceKW * 2048;      // bytes of a wave's x slot: [K-step][32 rows][4 pieces of 16 bytes]

__host__ __device__ constexpr int slice_part_bytes(int NT) { return NT * 4 * 1024; }   // a wave's partial accumulators
__host__ __device__ constexpr int slice_lds_bytes(int NT, int w0_floats) {
    return kSliceCompute * kSliceSlot + kSliceCompute * slice_part_bytes(NT) + kSliceCompute * 256 + w0_floats * 4 + 64;
}

#ifdef DPK_TIMELINE
#define SL_STAMP(row, slot) do { __builtin_amdgcn_sched_barrier(0); if (a.dbg && lane == 0 && (row) < 16 && blockIdx.x < 256) a.dbg[(((int64_t)blockIdx.x * 8 + wave) * 16 + (row)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SL_STAMP(row, slot) do { } while (0)
#endif

template <int I, int S, int NT>
__global__ __launch_bounds__(kSliceThreads) void ratspn_gemm_slice_kernel(const GemmArgs a) {
    constexpr int RPT = 8 / I;
    constexpr int NMAX = (I > S ? I : S);
    constexpr int KW = kSliceKW;
    constexpr int PW = slice_part_bytes(NT);
    typedef const __attribute__((address_space(1))) half8 gh8;
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) gf32x4 lf4w;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    lchar *part_l = smem + kSliceCompute * kSliceSlot;                        // [7][NT*4][64 lanes][16 bytes]
    lfloat *q_l = (lfloat *)(part_l + kSliceCompute * PW);                    // [7][64] sums of squares
    lfloat *w0_l = q_l + kSliceCompute * 64;                                  // [reps*2][S*I*I]
    unsigned long long *exact_l = reinterpret_cast<unsigned long long *>(
        reinterpret_cast<float *>(smem_generic + kSliceCompute * kSliceSlot + kSliceCompute * PW + kSliceCompute * 256) +
        a.reps * 2 * S * I * I);                                              // blocks left to the exact evaluation

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D;
    // (host: D == 16 * 7 * 7 -- every slice holds exactly KW K-steps; a run-time slice length made hipcc clone the loop
    // body per length and spill 184 registers)
    const int nblk = a.ntiles;                                                // blocks of 32 samples
    const int first = (int)blockIdx.x, stride = (int)gridDim.x;
    SL_STAMP(15, 0);

    if (wave < kSliceCompute) {
        // ============================================ slice waves ==================================================
        const int k0 = wave * KW;
        lchar *my = smem + wave * kSliceSlot;
        const unsigned my_u = (unsigned)(uintptr_t)my;
        // DMA roles: piece P = i*64 + lane of a K-step's [32 rows][4 pieces]: row P >> 2, LDS piece P & 3 holds source
        // piece (P & 3) ^ ((row >> 2) & 3)
        int drow[2], dcol[2];
        unsigned voff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int P = i * 64 + lane;
            drow[i] = P >> 2;
            dcol[i] = (P & 3) ^ ((drow[i] >> 2) & 3);
            voff[i] = (unsigned)(drow[i] * D + k0 * 16 + dcol[i] * 4) * 4u;
        }
        auto issue = [&](int blk) {
            const int64_t b0 = (int64_t)blk * 32;
            const gcchar_p xt = (gcchar_p)(a.x + b0 * D);
            if (b0 + 32 <= a.B) {
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    glds16<kGemmXNonTemporal>(voff[0] + kk * 64u, xt, my_u + kk * 2048);
                    glds16<kGemmXNonTemporal>(voff[1] + kk * 64u, xt, my_u + kk * 2048 + 1024);
                }
            } else {   // ragged last block: rows beyond the batch re-fetch its last row (never stored)
                const int nvalid = (int)(a.B - b0);
#pragma unroll
                for (int kk = 0; kk < KW; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        glds16<kGemmXNonTemporal>((unsigned)(min(drow[i], nvalid - 1) * D + (k0 + kk) * 16 + dcol[i] * 4) * 4u, xt,
                                                  my_u + kk * 2048 + i * 1024);
            }
        };
        if (first < nblk && !(a.ablate & 4)) issue(first);
        // the slice of the mean table: registers for the whole launch (plain loads, L2; hipcc's waits for them also cover
        // the older DMA requests above -- loads retire in order)
        half8 mh[KW][NT], ml[KW][NT];
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int ks = k0 + kk;
            const gcchar_p tb = (gcchar_p)a.mtab + ((int64_t)ks * NT * 2048 + lane * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mh[kk][t] = *(gh8 *)(tb + t * 2048);
                ml[kk][t] = *(gh8 *)(tb + t * 2048 + 1024);
            }
        }
        // The fragments must have ARRIVED before the loop: a wait that hipcc placed at their first use inside it would be a
        // vmcnt(0) behind the next block's DMA requests, every iteration.  An empty asm that reads each fragment pins the
        // wait here (and keeps hipcc from re-materialising the loads inside the loop).
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                asm volatile("" : "+v"(mh[kk][t]));
                asm volatile("" : "+v"(ml[kk][t]));
            }
        const int sw = (s >> 2) & 3;
        const lchar *xr0 = my + s * 64 + (((h * 2) ^ sw) << 4);
        const lchar *xr1 = my + s * 64 + (((h * 2 + 1) ^ sw) << 4);
        lf4w *pw = (lf4w *)(part_l + wave * PW + lane * 16);
        SL_STAMP(15, 1);
        [[maybe_unused]] int row = 0;
        for (int blk = first; blk < nblk; blk += stride) {
            SL_STAMP(row, 0);
            // the block's slice has landed (only this wave's own DMA requests are outstanding)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SL_STAMP(row, 1);
            float v[KW][8];
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const gf32x4 x0 = *(lf4 *)(xr0 + kk * 2048), x1 = *(lf4 *)(xr1 + kk * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[kk][i] = x0[i];
                    v[kk][4 + i] = x1[i];
                }
            }
            // ... and is in registers: the slot is free, the next block's slice travels under this block's arithmetic
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SL_STAMP(row, 2);
            if (blk + stride < nblk && !(a.ablate & 4)) issue(blk + stride);
            SL_STAMP(row, 3);
            gf32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            gf32x2 tq2 = {0.f, 0.f};
            if (!(a.ablate & 1)) {
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const gf32x2 pv = {v[kk][i], v[kk][i + 1]};
                        tq2 = __builtin_elementwise_fma(pv, pv, tq2);
                    }
                    half8 xh, xl;
                    split8(v[kk], xh, xl);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xh, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xl, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[kk][t], xh, acc[t], 0, 0, 0);
                }
            }
            SL_STAMP(row, 4);
            // the eighth wave has read the previous block's partials
            gemm_lds_barrier();
            SL_STAMP(row, 5);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const gf32x4 o = {acc[t][4 * i4], acc[t][4 * i4 + 1], acc[t][4 * i4 + 2], acc[t][4 * i4 + 3]};
                    pw[(t * 4 + i4) * 64] = o;
                }
            q_l[wave * 64 + lane] = tq2[0] + tq2[1];
            gemm_lds_barrier();   // this block's partials are complete
            SL_STAMP(row, 6);
            ++row;
        }
    } else {
        // ============================================ the eighth wave ==============================================
        for (int e = lane; e < a.reps * 2 * S * I * I; e += 64) w0_l[e] = a.W0[e];
        bool model_ok = true;
        for (int e = lane; e < NT * RPT; e += 64) model_ok = model_ok && (a.elig[e] != 0);
        model_ok = __all(model_ok);
        float cst[NT][16];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) cst[t][i] = a.biasT[(h * NT + t) * 16 + i];
        const lf4 *pr = (const lf4 *)(part_l + lane * 16);
        double ll_part = 0.0;
        bool saw_nan = false;
        unsigned long long exact_mask = 0ull;   // bit i: the i-th block of this work-group (host: at most 64 per launch)
        [[maybe_unused]] int row = 0;
        int it = 0;
        for (int blk = first; blk < nblk; blk += stride, ++it) {
            gemm_lds_barrier();   // (hands the partial buffer to the slice waves)
            gemm_lds_barrier();   // the block's seven partials are in LDS
            SL_STAMP(row, 0);
            gf32x16 acc[NT];
            float qsum;
            {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        gf32x4 sum4 = pr[(t * 4 + i4) * 64];
#pragma unroll
                        for (int w = 1; w < kSliceCompute; ++w) sum4 += pr[(w * (PW / 16)) + (t * 4 + i4) * 64];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[t][4 * i4 + j] = sum4[j];
                    }
                qsum = q_l[lane];
#pragma unroll
                for (int w = 1; w < kSliceCompute; ++w) qsum += q_l[w * 64 + lane];
            }
            // (every read above has returned before this wave arrives at the next barrier: gemm_lds_barrier waits)
            SL_STAMP(row, 1);
            if (a.ablate & 2) { ++row; continue; }
            const int64_t bw0 = (int64_t)blk * 32;
            const int64_t b = bw0 + s;
            const float qtot = qsum + __shfl_xor(qsum, 32, 64);
            saw_nan = saw_nan || (qtot != qtot);
            // the f16 split and the expanded square hold while sum x^2 <= 36 D (|mu| <= 6: DESIGN 3.3): NaN, +-inf and
            // huge evidence fail the same test
            const bool lane_exact = !(qtot <= kExpandBound * kExpandBound * (float)D);
            bool exact = !model_ok || __any(lane_exact);
            if (!exact) {
                double part = 0.0;
                exact = gemm_upper_fast<I, S, NT>(a, acc, cst, w0_l, qtot, h, b, part);
                if (!exact) ll_part += part;
            }
            // a block outside the fast path's envelope is evaluated exactly AFTER the stream, by all eight waves (what the
            // fast path stored for it is overwritten there).  Inside this loop the exact evaluation cost 150 spilled
            // registers -- and a kernel with scratch pays for it at every dispatch, used or not.
            if (exact) exact_mask |= 1ull << it;
            SL_STAMP(row, 2);
            ++row;
        }
        if (a.ll_sum != nullptr) {
            const double red = wave_reduce_sum(ll_part);
            if (lane == 0) {
                atomicAdd(a.ll_sum, red);
                if (blockIdx.x == 0) atomicAdd(a.ll_sum + 1, (double)a.B * (double)a.C);
            }
        }
        if (__any(saw_nan) && lane == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
        if (lane == 0) *exact_l = exact_mask;
    }
    SL_STAMP(15, 2);
    // ---- blocks that left the fast path: exact evaluation, a wave per block, the x slots as the nodes' scratch ----------
    __syncthreads();
    const unsigned long long todo = *exact_l;
    if (todo != 0ull) {
        LseScratch sc{reinterpret_cast<float *>(smem_generic) + tid * (2 * NMAX)};
        int n = 0;
        for (int it = 0; it < 64; ++it) {
            if (!((todo >> it) & 1ull)) continue;
            if ((n++ & 7) != wave) continue;
            gemm_exact_body<I, S, NT>(a, (int64_t)(first + it * stride) * 32, lane, sc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool gemm_slice_shape_ok(int D, int reps, int I, int S, int NT) {
    if (!(I == 2 && S == 2) || NT > 2) return false;   // (instantiations built; the mapping itself is general in I, S)
    if (D != 16 * kSliceCompute * kSliceKW) return false;   // 784 = 7 slices x 7 K-steps of 16 features
    return slice_lds_bytes(NT, reps * 2 * S * I * I) <= 160 * 1024;
}

// samples per launch from which the slice mapping is taken (dpk_ratspn_slice_batch_min; DPK_GEMM_SLICE_MIN in the
// environment sets the initial value; a negative value switches the mapping off)
constexpr int64_t kSliceBatchDefault = 16385;
static int64_t slice_batch_initial() {
    const char *e = getenv("DPK_GEMM_SLICE_MIN");
    return e ? (int64_t)atoll(e) : kSliceBatchDefault;
}
static int64_t &slice_batch_min_ref() {
    static int64_t v = slice_batch_initial();
    return v;
}
int64_t gemm_slice_min_batch() { return slice_batch_min_ref(); }

template <int I, int S, int NT>
static int gemm_slice_launch(const GemmArgs &a0, hipStream_t st) {
    const size_t lds = (size_t)slice_lds_bytes(NT, a0.reps * 2 * S * I * I);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "ratspn_gemm_slice: %zu bytes of LDS", lds);
    auto kern = ratspn_gemm_slice_kernel<I, S, NT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_FUSED);
    if (ev0) (void)hipEventRecord(ev0, st);
    // a work-group remembers the blocks it leaves to the exact evaluation in a 64-bit mask: at most 64 blocks per
    // work-group and launch, i.e. 524 288 samples per launch on 256 compute units -- larger batches take several launches
    const int64_t per_launch = (int64_t)64 * 32 * cus;
    for (int64_t off = 0; off < a0.B; off += per_launch) {
        GemmArgs a = a0;
        a.x = a0.x + off * a0.D;
        a.out = a0.out + off * a0.C;
        a.B = std::min(per_launch, a0.B - off);
        a.ntiles = cdiv(a.B, 32);
        const int grid = a.ntiles < cus ? a.ntiles : cus;
#ifdef DPK_TIMELINE
        {
            static unsigned long long *dbg = nullptr;
            if (!dbg) (void)hipMalloc(&dbg, (size_t)256 * 8 * 16 * 8 * 8);
            a.dbg = dbg;
            FILE *f = fopen("/tmp/dpk_timeline_slice_ptr.txt", "w");
            if (f) { fprintf(f, "%p %d\n", (void *)dbg, grid); fclose(f); }
        }
#endif
        DPK_LAUNCH(kern, dim3(grid), dim3(kSliceThreads), lds, st, a);
        DPK_CHECK_LAUNCH("ratspn_gemm_slice_kernel");
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    return DPK_OK;
}

// The caller (ratspn_gemm_forward) has built / checked the tables and filled the argument block.
int ratspn_gemm_slice_forward(const GemmArgs &a, int I, int S, int NT, hipStream_t st) {
    if (I == 2 && S == 2) return NT == 1 ? gemm_slice_launch<2, 2, 1>(a, st) : gemm_slice_launch<2, 2, 2>(a, st);
    set_error("ratspn_gemm_slice: (channels=%d, sums=%d) not built", I, S);
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk

extern "C" int64_t dpk_ratspn_slice_batch_min(int64_t samples) {
    int64_t &v = dpk::slice_batch_min_ref();
    const int64_t prev = v;
    v = samples < -1 ? dpk::slice_batch_initial() : samples;   // (-1: off, below: back to the initial value)
    return prev;
}
