// RAT-SPN fused forward, third mapping of the matrix-core route: persistent 32-sample blocks, the FEATURE axis split
// over seven waves that keep their slice of the mean table in REGISTERS for the whole launch.
//
// reference: RatSpn.forward (deeprob/spn/models/ratspn.py:105-122) = RegionGraphLayer.forward + GaussianLayer
// (deeprob/spn/layers/ratspn.py:87-108, :160-213), ProductLayer :272-286, SumLayer :363-378, RootLayer :446-458.
//
// Same formulation, tables and arithmetic as ratspn_gemm.hip (read its header first).  Why a third mapping: the ring
// kernel there streams 48 KB stages of which a third is the mean-table chunk, re-fetched from L2 for every 128-sample
// tile by the same four loader waves that fetch x, and a tile's upper layers run with the ring stalled (+4.6 us of 47 at
// 65 536 samples); the small-batch kernel re-reads the whole table for every 32 samples.  Here
//   * the table never moves: 49 K-steps of 16 features = 7 waves x 7 K-steps; a wave holds its 7 x NT x {hi, lo}
//     A-fragments (112 VGPRs at NT = 2) from the first block to the end of the launch;
//   * x is the only stream.  A block is 32 CONSECUTIVE rows = one contiguous 100 KB range of x.  Wave w copies ITS OWN
//     slice (rows x features [112 w, 112 w + 112)) into its private 14 KB of LDS by LDS-DMA: producer and consumer of a
//     slot are the same wave, so the x stream needs no barrier -- a counted vmcnt is the whole protocol.  The unit of the
//     stream is the K-step (2 KB per wave): K-step kk of block b goes to registers and K-step kk of block b + 1 is
//     requested into the same bytes, so every piece has a whole block period to land;
//   * a compute unit's request path takes ~37 cycles per 1 KB request whoever issues it (measured: 98 requests of a block
//     in the K loop made the loop 6k cycles), so the requests of the next block are SPREAD over the block period: three
//     K-steps inside the K loop, two in front of barrier A (where the wave would wait anyway), two behind barrier B;
//   * the seven partial accumulators of a block meet in LDS (56 KB, a layout that is conflict free for the MFMA-shaped
//     writers and for the readers) and all EIGHT waves evaluate the upper layers, 16 lanes per sample, one (repetition,
//     partition) per lane, seven partials added in a fixed order; two s_barriers per block;
//   * the prologue is the request path again: the table travels WITHOUT its structural zeros (a variable belongs to one of a
//     repetition's four regions: 7 KB + 1 KB of keep-masks per wave instead of 28 KB, ratspn_gemm_prep.h: stab / smask) by
//     LDS-DMA into the wave's idle partial-accumulator block and is expanded from there, K-step kk + 1 under K-step kk's
//     products of the first block (broadcast ds_read_b128 + AND);
//   * a launch that checks its parameter tables (the default of model(x)) does so on the EIGHTH wave of its first np
//     work-groups while the other seven run the first block; counters that only grow, nothing reset (slice_verify_share).
// What was measured on the way and not kept (git history of this file): an eighth wave evaluating the upper
// layers alone (15k cycles per block: the bottleneck); a loader wave + LDS counters instead of barriers (a wave holds at
// most 63 requests -- vmcnt is 6 bits -- and becomes latency bound); mixed forms of the two; the slot as the unit of the
// stream (period >= request latency + issue of 98 requests + copy); a row-run slot layout (3 % cheaper requests, does not
// split by K-step); staggered work-group starts (no effect); non-temporal requests (slower).
// A block in which a sample leaves the fast path's envelope (NaN / +-inf / huge evidence, large sum of squares, model outside
// the expanded square's bound, vanished sum node) is evaluated exactly after the stream, a wave per block (gemm_exact_body)
// -- correctness never depends on a hint; launches that meet NaN evidence set the hint that routes the following ones to
// the variant built for marginalised evidence (ratspn_gemm_nan.hip).
//
// Results agree with the other two mappings to fp32 rounding, not bit for bit (seven partial sums in a fixed order).
#include "ratspn_gemm_fused.h"
#include "ratspn_gemm_prep.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace dpk {

constexpr int kSliceCompute = 7;                 // waves that own a slice of the feature axis
constexpr int kSliceKW = 7;                      // K-steps of 16 features per slice (registers: 4 NT VGPRs x 2 per K-step)
constexpr int kSliceThreads = (kSliceCompute + 1) * 64;
constexpr int kSliceSlot = kSliceKW * 2048;      // bytes of a wave's x slot: [K-step][32 rows][4 pieces of 16 bytes]

__host__ __device__ constexpr int slice_part_bytes(int NT) { return NT * 4 * 1024; }   // a wave's partial accumulators
__host__ __device__ constexpr int slice_lds_bytes(int NT, int w0_floats) {
    return kSliceCompute * kSliceSlot + kSliceCompute * slice_part_bytes(NT) + kSliceCompute * 256 + 128 + 1024 + 0 * w0_floats;
}

#ifdef DPK_TIMELINE
#define SL_STAMP(row, slot) do { __builtin_amdgcn_sched_barrier(0); if (a.dbg && lane == 0 && (row) < 16 && blockIdx.x < 256) a.dbg[(((int64_t)blockIdx.x * 8 + wave) * 16 + (row)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SL_STAMP(row, slot) do { } while (0)
#endif

// The table-free exact evaluation of 32 samples (a launch that found its parameter tables stale, ratspn_gemm_prep.h):
// gemm_exact_body with the log-softmax weights taken straight from the raw sum / root weights.
template <int I, int S, int NT>
__device__ __forceinline__ void slice_exact_raw(const GemmArgs &a, const float *raw0, const float *rawr, int64_t bw0, int lane,
                                                LseScratch sc) {
    constexpr int RPT = 8 / I;
    constexpr int RH = (NT * RPT + 1) / 2;
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
            for (int p = 0; p < 2; ++p)
                prodsum_node_raw<I, S>(leaf[2 * p], leaf[2 * p + 1], raw0 + ((int64_t)rho * 2 + p) * S * I * I, sc.slot, n1[m][p]);
        }
    }
    const int M = a.reps * S * S;
    double part = 0.0;
    for (int cl = 0; cl < a.C; ++cl) {
        const float lse_row = raw_row_lse(rawr + (int64_t)cl * M, M);
        float mm = -INFINITY, ss = 0.f;
#pragma unroll
        for (int m = 0; m < RH; ++m) {
            const int rho = 2 * m + h;
            if (rho < a.reps) {
                float pm, ps;
                root_partial_raw<S>(n1[m][0], n1[m][1], rawr + (int64_t)cl * M + rho * S * S, lse_row, sc.slot, pm, ps);
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
        if (lane == 0) atomicAdd(a.ll_sum, part);
    }
}

// The in-launch check of the parameter tables (DPK_FLAG_PARAMS_VERIFY, ratspn_gemm_prep.h) for THIS mapping.  The small
// kernels put np table work-groups in front of their tiles; here every work-group is a persistent model work-group that
// owns its compute unit (13 more would start 13 of the 256 late by the check's 4-5 us), and ANY work-group that reaches
// its first barrier late ends the launch late.  So the check is the eighth wave's -- idle until the first barrier, ~8 us
// away -- in the first np work-groups: wave 7 of work-group g fingerprints ALL of table work-group g's inputs (20 KB for a
// repetition of the headline model: every load requested before the first is consumed, one round trip), compares with the
// stored fingerprint and counts itself in; the last of the np publishes the verdict.  Every work-group reads the verdict
// under a later K loop, or at its tail if it has no later block; a stale launch discards what it computed, evaluates on
// the table-free exact route, and its first np work-groups rebuild the tables in place.
//
// Nothing is reset and nobody leaves last (the VerifyCtl protocol's 256 reader tickets on one address were 3-4 us at
// the tail of every work-group here; 256 arrivals on one address, and a last arriver with two more round trips in front
// of ITS first barrier, +6.5 us at every batch size -- measured, both).  All counters only grow.  Launches on a stream
// follow one another, so the tickets launch k draws from shard s are exactly [T_s(k-1), T_s(k)), and T_s(k-1) is what
// verdict[s] holds until launch k's last hasher replaces it with T_s(k) = T_s(k-1) + (this launch's work-groups in
// shard s): a work-group with ticket n knows its launch's verdict is out as soon as verdict[s]'s total exceeds n.  The
// same for the hashers' own count (`hashed`, its pre-launch value in `hashed_base`; mismatches are counted in its upper
// half, and a carry out of the lower half after 2^32 arrivals reads as one spurious stale launch: a rebuild, same results).
constexpr int kSliceShards = 16;
struct SliceVerify {               // behind the VerifyCtl in the workspace (zeroed with the tables' first build)
    unsigned long long verdict[kSliceShards];   // T_s << 1 | the launch found a stale table
    unsigned long long hashed, hashed_base;     // arrivals of the hashing waves | mismatches << 32
    unsigned long long pad[14];
    struct { unsigned long long n, pad[15]; } tickets[kSliceShards];   // (an address per shard, 128 bytes apart)
};
static_assert(sizeof(SliceVerify) == kSliceVerifyBytes, "the workspace block behind the VerifyCtl (common.h)");
__device__ __forceinline__ SliceVerify *slice_verify_of(VerifyCtl *c) {
    return reinterpret_cast<SliceVerify *>(reinterpret_cast<char *>(c) + 64);
}
// A thread's share (tid of nthreads) of the fingerprint of `nwords` 4-byte words at p -- the value fp_range_n gives -- in
// two halves, so that the loads of SEVERAL ranges are all requested before any is consumed (fp_range_n's loops are one
// dependent round trip per trip).  Ranges of more than kFpK * nthreads words: fp_range_n (the host does not route those here).
constexpr int kFpK = 32;
template <int K>
__device__ __forceinline__ void fpw_issue(const unsigned *p, int nwords, int tid, int nthreads, unsigned (&w)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + nthreads * k;
        w[k] = p[e < nwords ? e : 0];
    }
}
template <int K>
__device__ __forceinline__ unsigned long long fpw_sum(const unsigned (&w)[K], int nwords, unsigned tag, int tid, int nthreads) {
    unsigned long long h = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + nthreads * k;
        if (e < nwords) h += fp_word(w[k], (unsigned)e * 8u + tag);
    }
    return h;
}
// the same with 16-byte loads (nwords a multiple of 4, p 16-byte aligned): a quarter of the requests -- the eighth wave's
// requests take turns with the seven slice waves' 1 KB table and evidence requests on the compute unit's one request path
typedef unsigned fp_u4 __attribute__((ext_vector_type(4)));
template <int K>
__device__ __forceinline__ void fpw_issue4(const unsigned *p, int nwords, int tid, int nthreads, fp_u4 (&w)[K]) {
    const fp_u4 *p4 = (const fp_u4 *)p;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + nthreads * k;
        w[k] = p4[4 * e < nwords ? e : 0];
    }
}
template <int K>
__device__ __forceinline__ unsigned long long fpw_sum4(const fp_u4 (&w)[K], int nwords, unsigned tag, int tid, int nthreads) {
    unsigned long long h = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = 4 * (tid + nthreads * k);
        if (e < nwords)
            h += (fp_word(w[k].x, (unsigned)e * 8u + tag) + fp_word(w[k].y, (unsigned)(e + 1) * 8u + tag)) +
                 (fp_word(w[k].z, (unsigned)(e + 2) * 8u + tag) + fp_word(w[k].w, (unsigned)(e + 3) * 8u + tag));
    }
    return h;
}
// gemm_prep_hash_share's value for table work-group blk, all loads up front (word-aligned ranges: what the tables' inputs are)
template <int I>
__device__ __forceinline__ unsigned long long slice_hash_share(const GemmPrepArgs &a, int blk, int tid, int nthreads) {
    constexpr int RPT = 8 / I;
    const int nrb = a.NT * RPT, d = a.d;
    if (blk < nrb) {
        if (blk >= a.reps) return 0ull;
        const int rho = blk;
        const int nm = 8 * d, npd = a.pad != nullptr ? d : 0, nl = 4 * I * d;
        const unsigned *pm = (const unsigned *)(a.mask + (int64_t)rho * 4 * d);
        const unsigned *pl = (const unsigned *)(a.loc + (int64_t)rho * 4 * I * d);
        const unsigned *ps = (const unsigned *)(a.scale + (int64_t)rho * 4 * I * d);
        const unsigned *pp = a.pad != nullptr ? (const unsigned *)(a.pad + (int64_t)rho * 4 * d) : pm;
        if (nl > kFpK * nthreads || nm > kFpK * nthreads || npd > 8 * nthreads || ((uintptr_t)pp & 3) != 0 ||
            (((uintptr_t)pm | (uintptr_t)pl | (uintptr_t)ps) & 15) != 0)
            return gemm_prep_hash_share<I>(a, blk, tid, nthreads);   // (long, byte-wise or unaligned ranges: the general form)
        fp_u4 wm[kFpK / 4], wl[kFpK / 4], ws[kFpK / 4];
        unsigned wp[8];
        fpw_issue4(pm, nm, tid, nthreads, wm);
        fpw_issue4(pl, nl, tid, nthreads, wl);
        fpw_issue4(ps, nl, tid, nthreads, ws);
        fpw_issue(pp, npd > 0 ? npd : 1, tid, nthreads, wp);
        return (fpw_sum4(wm, nm, 1, tid, nthreads) + fpw_sum(wp, npd, 2, tid, nthreads)) +
               (fpw_sum4(wl, nl, 3, tid, nthreads) + fpw_sum4(ws, nl, 4, tid, nthreads));
    }
    constexpr int rpb = kGemmPrepThreads / 64;
    unsigned long long h = 0ull;
    unsigned wv[rpb];
    bool in[rpb];
    unsigned tg[rpb];
#pragma unroll
    for (int r = 0; r < rpb; ++r) {   // (rows of up to nthreads weights: one load per thread and row, all in flight together)
        const int row = (blk - nrb) * rpb + r;
        int rr = row, m = -1;
#pragma unroll
        for (int mm = 0; mm < 3; ++mm) {
            if (m < 0) {
                if (rr < a.rows[mm]) m = mm;
                else rr -= a.rows[mm];
            }
        }
        const int n = m >= 0 ? a.n[m] : 0;
        if (n > nthreads) return gemm_prep_hash_share<I>(a, blk, tid, nthreads);   // (wave-uniform: the general form)
        in[r] = tid < n;
        const float *base = m >= 0 ? a.w[m] + (int64_t)rr * n : a.w[0];
        wv[r] = __float_as_uint(base[in[r] ? tid : 0]);
        tg[r] = (unsigned)tid * 8u + (5u + 8192u * (unsigned)row);
    }
#pragma unroll
    for (int r = 0; r < rpb; ++r)
        if (in[r]) h += fp_word(wv[r], tg[r]);
    return h;
}
// ctl_l (LDS, this work-group, written and read by the eighth wave's lane 0 only until the tail's barrier):
// [0] = 2 | stale once the verdict is known, [2..3] the work-group's ticket
template <int I>
__device__ __forceinline__ void slice_verify_share(const GemmPrepArgs &pa, int lane, lunsigned *ctl_l) {
    const int np = pa.np, G = (int)gridDim.x, g = (int)blockIdx.x;
    SliceVerify *sv = slice_verify_of(pa.ctl);
    const int shard = g & (kSliceShards - 1);
    if (g >= np) {   // (wave-uniform) not a hashing work-group: the ticket, nothing else
        if (lane == 0) {
            const unsigned long long t = __hip_atomic_fetch_add(&sv->tickets[shard].n, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl_l[0] = 0u;
            ctl_l[2] = (unsigned)t;
            ctl_l[3] = (unsigned)(t >> 32);
        }
        return;
    }
    // (what the last hasher needs, requested with the inputs: the shards' totals before this launch, lane s = shard s)
    const unsigned long long vold = __hip_atomic_load(&sv->verdict[lane & (kSliceShards - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hb = __hip_atomic_load(&sv->hashed_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long stored = pa.hash[g];
    unsigned long long h = slice_hash_share<I>(pa, g, lane, 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += (unsigned long long)__shfl_xor((long long)h, o, 64);
    const bool mismatch = kPrepHashBase + (unsigned long long)g + h != stored;
    unsigned long long t = 0ull, r = 0ull;
    if (lane == 0) {
        // (requests return in order: the ticket is drawn here, behind the inputs, not in front of them)
        t = __hip_atomic_fetch_add(&sv->tickets[shard].n, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = __hip_atomic_fetch_add(&sv->hashed, 1ull + (mismatch ? (1ull << 32) : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl_l[0] = 0u;
        ctl_l[2] = (unsigned)t;
        ctl_l[3] = (unsigned)(t >> 32);
    }
    const unsigned rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r >> 32));
    const unsigned hblo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb);
    const unsigned hbhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(hb >> 32));
    // (a hasher that read the NEW hashed_base is not the last one: its count is below that total)
    if (rlo - hblo != (unsigned)(np - 1)) return;
    const unsigned mism_now = rhi + (mismatch ? 1u : 0u);
    const bool stale = mism_now != hbhi;
    if (lane == 0)
        __hip_atomic_store(&sv->hashed_base, ((unsigned long long)mism_now << 32) | (unsigned long long)(rlo + 1u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    // (hashed_base is read by the NEXT launch's hashers only: no wait between it and the verdicts)
    if (lane < kSliceShards) {
        const int cnt = lane < G ? (G - 1 - lane) / kSliceShards + 1 : 0;
        __hip_atomic_store(&sv->verdict[lane], (((vold >> 1) + (unsigned long long)cnt) << 1) | (stale ? 1ull : 0ull), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) ctl_l[0] = 2u | (stale ? 1u : 0u);
}
// The eighth wave has nothing to do until its work-group's first barrier (5.8 us after entry at one block per work-group,
// 7.7 us otherwise): it polls for the verdict until `until` (s_memrealtime, 100 MHz), so that the tail finds it in LDS.
// (A look per block inside the block loop -- requested in front of the upper layers, examined behind them -- cost the
// FROZEN-model launch 2.2 us at 65 536 samples through the loop's register allocation: same-box A/B, 47.6 against 45.4.)
__device__ __forceinline__ void slice_verdict_poll(const GemmPrepArgs &pa, lunsigned *ctl_l, unsigned long long until) {
    const int shard = (int)blockIdx.x & (kSliceShards - 1);
    const unsigned long long *vp = &slice_verify_of(pa.ctl)->verdict[shard];
    const unsigned long long n = ((unsigned long long)ctl_l[3] << 32) | ctl_l[2];
    while ((ctl_l[0] & 2u) == 0u && __builtin_amdgcn_s_memrealtime() < until) {
        const unsigned long long v = __hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 1) > n) ctl_l[0] = 2u | (unsigned)(v & 1ull);
        else __builtin_amdgcn_s_sleep(16);
    }
}
// ... and the wait for it where the work-group cannot go on without (stale also if nothing came for 1 s)
__device__ __forceinline__ void slice_verdict_wait(const GemmPrepArgs &pa, lunsigned *ctl_l) {
    if ((ctl_l[0] & 2u) != 0u) return;
    const int shard = (int)blockIdx.x & (kSliceShards - 1);
    const unsigned long long *vp = &slice_verify_of(pa.ctl)->verdict[shard];
    const unsigned long long n = ((unsigned long long)ctl_l[3] << 32) | ctl_l[2];
    unsigned long long v = __hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool timed_out = false;
    if ((v >> 1) <= n) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        do {
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull;
        } while ((v >> 1) <= n && !timed_out);
    }
    ctl_l[0] = 2u | ((v & 1ull) != 0ull || timed_out ? 1u : 0u);
}

template <int I, int S, int NT>
__global__ __launch_bounds__(kSliceThreads) void ratspn_gemm_slice_kernel(const GemmArgs a, const GemmPrepArgs pa) {
    constexpr int RPT = 8 / I;
    constexpr int NMAX = (I > S ? I : S);
    constexpr int KW = kSliceKW;
    constexpr int PW = slice_part_bytes(NT);
    constexpr float kL2E = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    static_assert(I == 2 && NT == 2 && NT * RPT == 8, "phase-2 roles: 8 repetitions x 2 partitions = the 16 lanes of a sample; one b128 per partition");
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) gf32x4 lf4w;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    lchar *part_l = smem + kSliceCompute * kSliceSlot;                        // [7 slices][32 samples][16 units of 16 bytes]
    lfloat *q_l = (lfloat *)(part_l + kSliceCompute * PW);                    // [7][64] sums of squares
    lunsigned *flag_l = (lunsigned *)(q_l + kSliceCompute * 64);              // [2][8] a wave's verdict on its 4 samples of a block
    lunsigned *ctl_l = flag_l + 16;                                           // the table check: slice_verify_share
    lfloat *c2_l = (lfloat *)(flag_l + 32);                                   // [16 slots][16]: sum weights, leaf constants, root weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D;
    // (host: D == 16 * 7 * 7 -- every slice holds exactly KW K-steps; a run-time slice length made hipcc clone the loop
    // body per length and spill 184 registers)
    const int nblk = a.ntiles;                                                // blocks of 32 samples
    const int first = (int)blockIdx.x, stride = (int)gridDim.x;
    const bool slicer = wave < kSliceCompute;                                 // (the eighth wave only joins phase 2)
    SL_STAMP(15, 0);
    // (the eighth wave, idle until the first barrier: the launch's table check, before anything of the stream is live)
    unsigned long long t_poll = 0ull;
    if (!slicer && pa.np > 0) {
        const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_setprio(3);   // (its requests in front of the same SIMD's slice wave's: the chain below is three round trips)
        slice_verify_share<I>(pa, lane, ctl_l);
        __builtin_amdgcn_s_setprio(0);
        t_poll = t_entry + (a.ntiles > (int)gridDim.x ? 500ull : 450ull);
    }

    // ---- phase-1 roles (slice waves): MFMA lane (sample s, half h) ----------------------------------------------------
    const int k0 = wave * KW;
    lchar *my = smem + (slicer ? wave : 0) * kSliceSlot;
    const unsigned my_u = (unsigned)(uintptr_t)my;
    // Slot layout [K-step][32 rows][4 pieces of 16 bytes]: a K-step is two DMA instructions (piece P = i*64 + lane: row
    // P >> 2, LDS piece P & 3 holds source piece (P & 3) ^ ((row >> 2) & 3), so that the MFMA-shaped ds_read_b128 -- 16 rows
    // x one piece -- covers 16 different 16-byte bank groups).  The K-step is the unit of the stream: the wave copies K-step
    // kk of block b into registers and at once requests K-step kk of block b + 1 into the same 2 KB, so every piece has a
    // whole block period to land.  (With the slot as the unit -- all 14 requests after the whole copy -- the period could
    // not fall below request latency + 98 instructions of issue + copy: measured 11k cycles; the row-run layout, 3 % cheaper
    // to request, does not split by K-step.)
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = i * 64 + lane, drow = P >> 2, dcol = (P & 3) ^ ((drow >> 2) & 3);
        voff[i] = (unsigned)(drow * D + k0 * 16 + dcol * 4) * 4u;
    }
    // the two requests of K-step kk of the 32 rows at xt (rows beyond nvalid re-fetch the last one: never stored)
    auto issue_kk = [&](gcchar_p xt, int nvalid, int kk) {
        if (nvalid >= 32) {
            glds16<kGemmXNonTemporal>(voff[0] + kk * 64u, xt, my_u + kk * 2048);
            glds16<kGemmXNonTemporal>(voff[1] + kk * 64u, xt, my_u + kk * 2048 + 1024);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = i * 64 + lane, drow = P >> 2, dcol = (P & 3) ^ ((drow >> 2) & 3);
                glds16<kGemmXNonTemporal>((unsigned)(min(drow, nvalid - 1) * D + (k0 + kk) * 16 + dcol * 4) * 4u, xt,
                                          my_u + kk * 2048 + i * 1024);
            }
        }
    };
    auto issue = [&](int blk) {
        const int64_t b0 = (int64_t)blk * 32;
        const gcchar_p xt = (gcchar_p)(a.x + b0 * D);
        const int nvalid = (int)min((int64_t)32, a.B - b0);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) issue_kk(xt, nvalid, kk);
    };
    const int nit = first < nblk ? (nblk - first + stride - 1) / stride : 0;   // blocks of this work-group (>= 1)
    // The slice of the mean table: registers for the whole launch.  Its 28 KB per wave were 28 requests of 1 KB on the compute
    // unit's request path (196 per compute unit, 37 cycles each, in front of the first product -- and 50 MB out of the
    // L2s per launch); three quarters of the entries are structural zeros (a variable belongs to ONE region of a
    // repetition), so the table travels without them (ratspn_gemm_prep.h: stab, 7 KB per wave + 1 KB of keep-masks) by
    // LDS-DMA into the wave's own partial-accumulator block, idle until the first barrier, and is expanded from there:
    // ds_read_b128 (four lanes -- the four regions of a (repetition, channel) -- share an address: broadcast), then an AND
    // with the lane's keep-mask.  All of it while the first block's x is on its way from HBM.
    // (measured with a quarter of the table requests and garbage values: -1.5 / -2.5 / -4.5 / -4.3 us at 8192 / 16384 /
    // 32768 / 65536 samples)
    lchar *tl = part_l + (slicer ? wave : 0) * PW;
    static_assert(PW >= KW * NT * kSliceTabBytes + 1024, "compact table + keep-masks in the wave's partial block");
    if (slicer && nit > 0) {
        // (the table in front of x: interleaved K-step by K-step -- the first product then needs the first four requests
        // only -- measured the same; x first puts the table behind x's HBM latency: requests complete in order)
        const unsigned tl_u = (unsigned)(uintptr_t)tl;
        glds16<false>((unsigned)lane * 16u, (gcchar_p)pa.smask + wave * 1024, tl_u + KW * NT * kSliceTabBytes);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
            glds16<false>((unsigned)lane * 16u, (gcchar_p)pa.stab + (int64_t)(k0 + kk) * NT * kSliceTabBytes,
                          tl_u + kk * NT * kSliceTabBytes);
        issue(first);
    }
    half8 mh[KW][NT], ml[KW][NT];
    typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const gu32x4 lu4;
    typedef __attribute__((address_space(3))) const unsigned short lushort;
#define DPK_SL_TABLE_EXPAND(kk)                                                                                                \
    do {                                                                                                                      \
        const unsigned bits = *(lushort *)(tl + KW * NT * kSliceTabBytes + lane * 16 + (kk) * 2);                             \
        const lchar *te = tl + (kk) * NT * kSliceTabBytes + ((lane >> 5) * 8 + ((lane >> 3) & 3) * 2 + (lane & 1)) * 16;      \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                      \
            gu32x4 hi = *(lu4 *)(te + t * kSliceTabBytes);                                                                    \
            gu32x4 lo = *(lu4 *)(te + t * kSliceTabBytes + kSliceTabBytes / 2);                                               \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                \
                const int e0 = __builtin_amdgcn_sbfe((int)bits, t * 8 + 2 * jj, 1);                                           \
                const int e1 = __builtin_amdgcn_sbfe((int)bits, t * 8 + 2 * jj + 1, 1);                                       \
                const unsigned keep = ((unsigned)e0 & 0xFFFFu) | ((unsigned)e1 & 0xFFFF0000u);                                \
                hi[jj] &= keep;                                                                                               \
                lo[jj] &= keep;                                                                                               \
            }                                                                                                                 \
            mh[kk][t] = __builtin_bit_cast(half8, hi);                                                                        \
            ml[kk][t] = __builtin_bit_cast(half8, lo);                                                                        \
        }                                                                                                                     \
    } while (0)
    // ---- phase-2 roles (all eight waves): 16 consecutive lanes own a sample, slot j = 2 rho + p ------------------------
    const int sl = lane >> 4, j = lane & 15;
    const int rho = j >> 1, p = j & 1;
    const int s2 = wave * 4 + sl;                    // sample of the block
    const bool active = rho < a.reps;
    const int rc = active ? rho : a.reps - 1;        // (spare slots compute on a copy and are masked out)
    const int t2 = rc / RPT, ap = rc - t2 * RPT;
    // phase-2 constants of slot j (sum weights of its partition, leaf constants of its two regions, root weights of
    // class 0) live in LDS, 64 bytes per slot: four b128 reads per block instead of 16 registers held through phase 1
    static_assert(S * I * I == 8 && 2 * I == 4 && S * S == 4, "slot record layout");
    if (wave == 7 && sl == 0) {
        const float *wp = a.W0 + ((int64_t)(rc * 2 + p) * S) * I * I;
#pragma unroll
        for (int e = 0; e < S * I * I; ++e) c2_l[j * 16 + e] = wp[e];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int k = 0; k < I; ++k) c2_l[j * 16 + 8 + qq * I + k] = a.biasT[(p * NT + t2) * 16 + (ap * 2 + qq) * I + k];
#pragma unroll
        for (int e = 0; e < S * S; ++e) c2_l[j * 16 + 12 + e] = ((const float *)a.Wr)[rc * S * S + e];
    }
    // (the eighth wave polls for the verdict up to 4.5 us after entry -- 5 at two or more blocks per work-group: the first
    // barrier, 5.8 / 7.7 us after entry, is not kept waiting)
    if (!slicer && pa.np > 0 && lane == 0) slice_verdict_poll(pa, ctl_l, t_poll);
    bool model_ok;
    {   // (through the scalar cache: a vector load here would put a compiler-counted wait into the request queue above)
        cint_p ce = as_const(a.elig);
        unsigned bits = 0u;
#pragma unroll
        for (int e = 0; e < NT * RPT; ++e) bits |= (ce[e] != 0 ? 1u : 0u) << e;
        model_ok = ((bits >> rc) & 1u) != 0u;
    }
    const int M = a.reps * S * S;
    const float *wr0 = (const float *)a.Wr + rc * S * S;
    if (tid < 16) flag_l[tid] = 0u;

    // partial accumulators in LDS: slice w, unit (16 bytes) U(s, h, g) = s*16 + ((2 g + h) ^ (s & 7)) for the four values
    // of partition (repetition g, half h) of sample s.  Writers (fixed g; 8 consecutive lanes = 8 samples per
    // ds_write_b128 group) cover 8 different units mod 8, readers (16-lane ds_read_b128 groups = two samples x 8 of the 16
    // (g, h) slots) 16 different units mod 16: both sides conflict free.
    const int sw = (s >> 2) & 3;
    constexpr int XROW = 64, XKK = 2048;             // bytes between rows / between K-steps of a row
    const lchar *xr0 = my + s * XROW + (((h * 2) ^ sw) << 4);
    const lchar *xr1 = my + s * XROW + (((h * 2 + 1) ^ sw) << 4);
    lchar *pw = part_l + (slicer ? wave : 0) * PW + s * 256;

    double ll_part = 0.0, pend_part = 0.0;
    bool saw_nan = false;
    unsigned long long exact_mask = 0ull;   // bit i: the i-th block of this work-group (host: at most 64 per launch)
    gf32x16 acc[NT];
    float qlane = 0.f;
#define DPK_SL_KSTEP(kk, v, tq2)                                                                                              \
    do {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; i += 2) {                                                                    \
            const gf32x2 pv = {v[i], v[i + 1]};                                                                               \
            tq2 = __builtin_elementwise_fma(pv, pv, tq2);                                                                     \
        }                                                                                                                     \
        half8 xh, xl;                                                                                                         \
        split8(v, xh, xl);                                                                                                    \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xh, acc[t], 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[kk][t], xl, acc[t], 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[kk][t], xh, acc[t], 0, 0, 0); \
    } while (0)
    // ---- the work-group's first block ------------------------------------------------------------------------------------------
    // Request queue (in order): keep-masks (1), table K-steps (7), x of block 0 (14) | per step kk <= 2: x of block 1, K-step kk
    // (2).  "All but the youngest N have completed" with N = what was issued after the awaited request: the table: 14; x
    // K-step 0: 12; x K-step kk + 1 at step kk (it is copied a step ahead): 12, 12, 12, 10, 8, 6 (10, 8, 6, 4, 2, 0 when there
    // is no second block).  K-step kk + 1 of the table is expanded under K-step kk's products.
    if (slicer && nit > 0) {
        const bool more = nit > 1;
        const int64_t nb0 = (int64_t)(first + stride) * 32;
        const gcchar_p nxt = (gcchar_p)(a.x + nb0 * D);
        const int nnv = (int)min((int64_t)32, a.B - nb0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        gf32x2 tq2 = {0.f, 0.f};
        gf32x4 xa0, xa1, xb0, xb1;
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");   // the table and the masks (x of block 0 is younger)
        DPK_SL_TABLE_EXPAND(0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // K-step 0 of x
        xa0 = *(lf4 *)(xr0);
        xa1 = *(lf4 *)(xr1);
#define DPK_SL_FIRST_STEP(kk, NW_MORE, NW_LAST)                                                                               \
    do {                                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
        if (more && (kk) <= 2) issue_kk(nxt, nnv, kk);                                                                        \
        float v[8];                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
            v[i] = xa0[i];                                                                                                    \
            v[4 + i] = xa1[i];                                                                                                \
        }                                                                                                                     \
        if ((kk) + 1 < KW) {                                                                                                  \
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW_MORE) : "memory");                                          \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW_LAST) : "memory");                                               \
            xb0 = *(lf4 *)(xr0 + ((kk) + 1) * XKK);                                                                           \
            xb1 = *(lf4 *)(xr1 + ((kk) + 1) * XKK);                                                                           \
        }                                                                                                                     \
        if ((kk) + 1 < KW) DPK_SL_TABLE_EXPAND(((kk) + 1 < KW ? (kk) + 1 : 0));                                               \
        DPK_SL_KSTEP(kk, v, tq2);                                                                                             \
        xa0 = xb0;                                                                                                            \
        xa1 = xb1;                                                                                                            \
    } while (0)
        DPK_SL_FIRST_STEP(0, 12, 10);
        DPK_SL_FIRST_STEP(1, 12, 8);
        DPK_SL_FIRST_STEP(2, 12, 6);
        DPK_SL_FIRST_STEP(3, 10, 4);
        DPK_SL_FIRST_STEP(4, 8, 2);
        DPK_SL_FIRST_STEP(5, 6, 0);
        DPK_SL_FIRST_STEP(6, 0, 0);
#undef DPK_SL_FIRST_STEP
#undef DPK_SL_TABLE_EXPAND
        qlane = tq2[0] + tq2[1];
        if (more) {
            issue_kk(nxt, nnv, 3);
            issue_kk(nxt, nnv, 4);
        }
    }
    SL_STAMP(15, 1);
    [[maybe_unused]] int row = 0;
    int it = 0;
    for (; it < nit; ++it) {
        const int blk = first + it * stride;
        const bool more = it + 1 < nit;                 // a next block exists: its K loop runs at the end of this iteration
        const bool more2 = it + 2 < nit;                // ... and requests the block after it
        SL_STAMP(row, 0);
        // everyone has read the previous block's partials (and published its verdict on it)
        gemm_lds_barrier();
        SL_STAMP(row, 1);
        if (slicer) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int g = t * 4 + i4;
                    const gf32x4 o = {acc[t][4 * i4], acc[t][4 * i4 + 1], acc[t][4 * i4 + 2], acc[t][4 * i4 + 3]};
                    *(lf4w *)(pw + (((g * 2 + h) ^ (s & 7)) << 4)) = o;
                }
            q_l[wave * 64 + lane] = qlane;
        }
        // the previous block's verdict: its sum joins the total only if no wave left the fast path on it (the exact
        // evaluation at the end then covers -- and sums -- all 32 samples of the block)
        if (it > 0) {
            const lunsigned *fl = flag_l + ((it - 1) & 1) * 8;
            const unsigned any_bad = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)((fl[0] | fl[1]) | (fl[2] | fl[3]) | (fl[4] | fl[5]) | (fl[6] | fl[7])));
            if (any_bad) exact_mask |= 1ull << (it - 1);
            else ll_part += pend_part;
        }
        gemm_lds_barrier();   // this block's partials are complete
        if (slicer && more) {                           // K-steps 5, 6 of the next block
            const int64_t nb0 = (int64_t)(first + (it + 1) * stride) * 32;
            const gcchar_p nxt = (gcchar_p)(a.x + nb0 * D);
            const int nnv = (int)min((int64_t)32, a.B - nb0);
            issue_kk(nxt, nnv, 5);
            issue_kk(nxt, nnv, 6);
        }
        SL_STAMP(row, 2);

        // =========================== phase 2: one (sample, repetition, partition) per lane ==============================
        // leaf sums of the lane's two regions: seven partials each, fixed order (launches agree bit for bit)
        // (the phase-2 addresses are recomputed per block from an opaque copy of the lane id: held through phase 1 they
        // cost the registers that decide between 256 and a spill)
        int jo = j;
        asm volatile("" : "+v"(jo));
        const lchar *pr = part_l + s2 * 256 + ((jo ^ (s2 & 7)) << 4);
        const lfloat *qr = q_l + (jo < 2 * kSliceCompute ? (jo >> 1) * 64 + (jo & 1) * 32 + s2 : 0);
        gf32x4 lf = *(lf4 *)pr;
#pragma unroll
        for (int w = 1; w < kSliceCompute; ++w) lf += *(lf4 *)(pr + w * PW);
        float qv = *qr;
        qv = j < 2 * kSliceCompute ? qv : 0.f;
        const float qtot = row16_sum(qv);
        const int64_t b2 = (int64_t)blk * 32 + s2;
        saw_nan = saw_nan || (qtot != qtot);
        // the f16 split and the expanded square hold while sum x^2 <= 36 D (|mu| <= 6: DESIGN 3.3): NaN, +-inf and huge
        // evidence fail the same test
        bool bad = !(qtot <= kExpandBound * kExpandBound * (float)D) || (active && !model_ok);
        const lf4 *c2 = (const lf4 *)(c2_l + jo * 16);
        const gf32x4 w0a = c2[0], w0b = c2[1], cs4 = c2[2], wr4 = c2[3];
        const float w0[S][I * I] = {{w0a[0], w0a[1], w0a[2], w0a[3]}, {w0b[0], w0b[1], w0b[2], w0b[3]}};
        float va[I], vc[I];
#pragma unroll
        for (int k = 0; k < I; ++k) {
            va[k] = lf[k] + cs4[k];
            vc[k] = lf[I + k] + cs4[I + k];
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
        // root: the two partitions of a repetition meet (lane ^ 1), then the repetitions of the sample
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
        for (int cl = 0; cl < a.C; ++cl) {
            float wr[S * S];
            if (cl == 0) {
#pragma unroll
                for (int e = 0; e < S * S; ++e) wr[e] = wr4[e];
            } else {
#pragma unroll
                for (int e = 0; e < S * S; ++e) wr[e] = wr0[(int64_t)cl * M + e];
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
            const float ll = ((mtop > -INFINITY) ? fmaf(__builtin_amdgcn_logf(tot), kLn2, mtop) : -INFINITY) + qterm;
            if (writer) {
                a.out[b2 * a.C + cl] = ll;      // (an exact verdict rewrites these at the end of the kernel)
                part += (double)ll;
            }
        }
        pend_part = part;
        // the wave's verdict on its 4 samples of this block, read by everyone behind the next barrier
        const bool wave_bad = __any(bad);
        if (lane == 0) flag_l[(it & 1) * 8 + wave] = wave_bad ? 1u : 0u;
        SL_STAMP(row, 3);
        // =========================== the next block's K loop ================================================================
        if (slicer && more) {
            const int64_t nb2 = (int64_t)(first + (it + 2) * stride) * 32;
            const gcchar_p nxt = (gcchar_p)(a.x + nb2 * D);
            const int nnv = (int)min((int64_t)32, a.B - nb2);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            gf32x2 tq2 = {0.f, 0.f};
            // K-step p: landed -> registers -> its 2 KB requested again for the block after.  Requests complete in order, so
            // "all but the youngest N" is K-step p with N = the requests younger than it.  Those are spread over the whole
            // block period -- K-steps 0..2 right after their copy, 3 and 4 behind the K loop (in front of barrier A, where
            // the wave would wait anyway), 5 and 6 behind barrier B -- because a compute unit's request path takes ~37
            // cycles per DMA instruction: all 98 in the K loop made it 6k cycles long.  Hence N = 2 (6 - p) of this block +
            // 2 min(p, 3) of the next; the last block requests nothing and counts down.  (Stores of the upper layers still in
            // flight only make the wait longer: k completed operations include k - #stores loads.)
            gf32x4 xa0, xa1, xb0, xb1;
#define DPK_SL_LAND(p)                                                                                                        \
    do {                                                                                                                      \
        if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (6 - (p)) + 2 * ((p) < 3 ? (p) : 3)) : "memory");              \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 - 2 * (p)) : "memory");                                              \
    } while (0)
            DPK_SL_LAND(0);
            xa0 = *(lf4 *)(xr0);
            xa1 = *(lf4 *)(xr1);
#define DPK_SL_STEP(kk)                                                                                                       \
    do {                                                                                                                      \
        /* (K-step kk is on its way to the registers; once there, its slot is requested again) */                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
        if (more2 && (kk) <= 2) issue_kk(nxt, nnv, kk);                                                                       \
        float v[8];                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
            v[i] = xa0[i];                                                                                                    \
            v[4 + i] = xa1[i];                                                                                                \
        }                                                                                                                     \
        if ((kk) + 1 < KW) {                                                                                                  \
            DPK_SL_LAND(((kk) + 1 < KW ? (kk) + 1 : 6));                                                                      \
            xb0 = *(lf4 *)(xr0 + ((kk) + 1) * XKK);                                                                           \
            xb1 = *(lf4 *)(xr1 + ((kk) + 1) * XKK);                                                                           \
        }                                                                                                                     \
        DPK_SL_KSTEP(kk, v, tq2);                                                                                             \
        xa0 = xb0;                                                                                                            \
        xa1 = xb1;                                                                                                            \
    } while (0)
            DPK_SL_STEP(0);
            DPK_SL_STEP(1);
            DPK_SL_STEP(2);
            DPK_SL_STEP(3);
            DPK_SL_STEP(4);
            DPK_SL_STEP(5);
            DPK_SL_STEP(6);
#undef DPK_SL_STEP
#undef DPK_SL_LAND
            qlane = tq2[0] + tq2[1];
            if (more2) {
                issue_kk(nxt, nnv, 3);
                issue_kk(nxt, nnv, 4);
            }
        }
        SL_STAMP(row, 4);
        ++row;
    }
    SL_STAMP(15, 2);
    // ---- the launch's verdict on its tables, the last block's verdict, then what left the fast path ------------------------
    if (pa.np > 0 && !slicer && lane == 0) slice_verdict_wait(pa, ctl_l);
    __syncthreads();
    const bool tables_stale = pa.np > 0 && (ctl_l[0] & 1u) != 0u;
    if (it > 0) {
        const lunsigned *fl = flag_l + ((it - 1) & 1) * 8;
        const unsigned any_bad = (fl[0] | fl[1]) | (fl[2] | fl[3]) | (fl[4] | fl[5]) | (fl[6] | fl[7]);
        if (any_bad) exact_mask |= 1ull << (it - 1);
        else ll_part += pend_part;
    }
    if (tables_stale) {   // (nothing computed from the stale tables counts: every block goes through the table-free route)
        ll_part = 0.0;
        exact_mask = nit >= 64 ? ~0ull : ((1ull << nit) - 1ull);
    }
    if (a.ll_sum != nullptr) {
        // {sum of LLs, count}: one atomic per work-group (and one for the count per launch)
        double *red_l = reinterpret_cast<double *>(smem_generic + kSliceCompute * kSliceSlot);   // (the partials are idle now)
        // (only the lanes 0, 16, 32, 48 -- the writers of the wave's four samples -- hold anything: two exchange steps, not six;
        // this chain sits at the very end of every work-group, and at one block per work-group -- a strong-scaling shard --
        // the end of the work-group is the end of the launch)
        double red = ll_part + __shfl_xor(ll_part, 16, 64);
        red += __shfl_xor(red, 32, 64);
        if (lane == 0) red_l[wave] = red;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += red_l[w];
            // (DPK_FLAG_LL_SUM_SPREAD: sixteen partial sums -- 256 work-groups that finish together are 256 same-address
            // fp64 atomics in series, 1.6 us behind a 12 us launch at one block per work-group; the caller adds the sixteen)
            atomicAdd(a.ll_sum + (a.ll_cnt > 1 ? ((int)blockIdx.x & 15) : 0), tot);
            if (blockIdx.x == 0) atomicAdd(a.ll_sum + a.ll_cnt, (double)a.B * (double)a.C);
        }
    }
    if (__any(saw_nan) && lane == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
    if (exact_mask != 0ull) {   // (every wave holds the same mask)
        int tid_x = (int)threadIdx.x;   // (an opaque copy: nothing of this tail is worth a register held through the loop)
        asm volatile("" : "+v"(tid_x));
        const int lane_x = tid_x & 63;
        LseScratch sc{reinterpret_cast<float *>(smem_generic) + tid_x * (2 * NMAX)};   // (the x slots are idle now)
        int n = 0;
        for (int e = 0; e < 64; ++e) {
            if (!((exact_mask >> e) & 1ull)) continue;
            if ((n++ & 7) != wave) continue;
            if (tables_stale) slice_exact_raw<I, S, NT>(a, pa.w[0], pa.w[1], (int64_t)(first + e * stride) * 32, lane_x, sc);
            else gemm_exact_body<I, S, NT>(a, (int64_t)(first + e * stride) * 32, lane_x, sc);
        }
    }
    // a stale launch: the first np work-groups rebuild their table work-group's outputs in place (nobody consumes a table
    // value any more) and store the new fingerprints -- the next launch is clean again
    if (tables_stale && (int)blockIdx.x < pa.np) {
        __syncthreads();   // (the exact evaluation's scratch is the rebuild's staging area)
        // (mode through the override: a by-value copy of the argument block costs 264 bytes of scratch per lane, every launch)
        gemm_prep_block<I>(pa, (int)blockIdx.x, reinterpret_cast<int *>(smem_generic), kPrepBuild);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool gemm_slice_shape_ok(int D, int reps, int I, int S, int NT) {
    if (!(I == 2 && S == 2) || NT != 2) return false;   // (the instantiation built: two channels, 5..8 repetitions)
    if (D != 16 * kSliceCompute * kSliceKW) return false;   // 784 = 7 slices x 7 K-steps of 16 features
    return slice_lds_bytes(NT, reps * 2 * S * I * I) <= 160 * 1024;
}

// samples per launch from which the slice mapping is taken (dpk_ratspn_slice_batch_min; DPK_GEMM_SLICE_MIN in the
// environment sets the initial value; a negative value switches the mapping off)
// (7681 = 240 blocks: up to there the small-batch kernel's 13 in-launch table work-groups still fit beside its tiles in one
// round of 256 compute units; at 8192 it needs a second round -- 25.7 us against 15.0, profiles/r05_shard8192_kernel_stats.txt)
constexpr int64_t kSliceBatchDefault = 7681;
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
static int gemm_slice_launch(const GemmArgs &a0, const GemmPrepArgs &p, hipStream_t st) {
    size_t lds = (size_t)slice_lds_bytes(NT, a0.reps * 2 * S * I * I);
    if (p.np > 0 && gemm_prep_lds_bytes(a0.D, I, a0.d) > lds) lds = gemm_prep_lds_bytes(a0.D, I, a0.d);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "ratspn_gemm_slice: %zu bytes of LDS", lds);
    auto kern = ratspn_gemm_slice_kernel<I, S, NT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_FUSED);
    if (ev0) (void)hipEventRecord(ev0, st);
    // a work-group remembers the blocks it leaves to the exact evaluation in a 64-bit mask: at most 64 blocks per
    // work-group and launch, i.e. 524 288 samples per launch on 256 compute units -- larger batches take several launches
    // (the caller keeps the in-launch table check to single-launch batches: gemm_slice_checks_inline)
    const int64_t per_launch = (int64_t)64 * 32 * cus;
    for (int64_t off = 0; off < a0.B; off += per_launch) {
        GemmArgs a = a0;
        a.x = a0.x + off * a0.D;
        a.out = a0.out + off * a0.C;
        a.B = std::min(per_launch, a0.B - off);
        a.ntiles = cdiv(a.B, 32);
        const int grid = a.ntiles < cus ? a.ntiles : cus;
        GemmPrepArgs pp = p;
        pp.readers = grid;
#ifdef DPK_TIMELINE
        {
            static unsigned long long *dbg = nullptr;
            if (!dbg) (void)hipMalloc(&dbg, (size_t)256 * 8 * 16 * 8 * 8);
            a.dbg = dbg;
            FILE *f = fopen("/tmp/dpk_timeline_slice_ptr.txt", "w");
            if (f) { fprintf(f, "%p %d\n", (void *)dbg, grid); fclose(f); }
        }
#endif
        DPK_LAUNCH(kern, dim3(grid), dim3(kSliceThreads), lds, st, a, pp);
        DPK_CHECK_LAUNCH("ratspn_gemm_slice_kernel");
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    return DPK_OK;
}

// whether a launch of B samples can carry its own table check: one launch, at least np work-groups (the eighth wave of
// work-group g fingerprints table work-group g's inputs: ranges of at most kFpK x 64 words -- d <= 256 features per region)
bool gemm_slice_checks_inline(int64_t B, int np, int d) {
    const int cus = device_cus();
    return B <= (int64_t)64 * 32 * cus && cdiv(B, 32) >= np && 8 * d <= kFpK * 64;
}

// The caller (ratspn_gemm_forward) has built / checked the tables and filled the argument block.
int ratspn_gemm_slice_forward(const GemmArgs &a, const GemmPrepArgs &p, int I, int S, int NT, hipStream_t st) {
    if (I == 2 && S == 2 && NT == 2) return gemm_slice_launch<2, 2, 2>(a, p, st);
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
