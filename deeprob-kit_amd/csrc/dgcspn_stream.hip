// DGC-SPN, evaluation route: streaming kernels of the fused depthwise product + sum level for 8 -> 8 channels
// (reference: deeprob/spn/models/dgcspn.py:146-150 chaining layers/dgcspn.py:224-236 and :289-304).
//
//   out[b,o,p] = logsumexp_c( sum_taps in[b,c,tap(p)] + log W[o,c,p] )
//
// The sum layer's weights depend on the pixel, so a sample needs 64 weights per output pixel against 8 inputs and 8
// outputs: the weights have to be amortised over the batch.  Here a thread owns one output pixel for the whole
// launch and keeps its 8 x 8 softmaxed weights in registers; a work-group owns a tile of pixels and walks a slice of
// the batch.  The rows of the input map that the tile's taps touch are copied into LDS by loader waves with LDS-DMA
// (16 bytes per lane, nothing passes through registers) a few samples ahead of the compute waves, which read their
// taps with ds_read_b32 -- no tap is fetched through the vector cache, no address is recomputed per sample, and the
// zero padding of the product layer is a zeroed guard word that out-of-map taps point at.  One s_barrier per sample.
//
//   LDS, per stage (= one sample): 8 channel slots of CS floats; a slot holds the tile's row bands back to back
//   (each band starts on a 16-byte piece of global memory; band starts are congruent mod 4 rows so that the
//   sub-piece shift of a channel is the same in every band) and ends with 4 guard floats that stay zero.
//   Address of tap t in channel c: offb[t] (per thread, set once) + Kb[c] (per channel, wave-uniform).
//
// Round 6: PIXEL-MAJOR maps between the streaming levels (IN_PM / OUT_PM; torch's channels_last: [B, H, W, 8]).  The
// channel-major stage costs a thread 32 ds_read_b32 + 32 address adds per sample for its 4 taps x 8 channels, and a
// level 8 scattered stores: a third of the ~200 instructions of a pixel, on kernels that are bound by instruction issue
// (the whole model: ~32 SIMD cycles per pixel and sample at every level).  With the 8 channels of a pixel adjacent, a tap
// is two ds_read_b128 (8 LDS instructions and 4 address adds per sample), the level's result two 16-byte stores, a
// stage one contiguous copy per row band (rows are 32 W bytes: no sub-piece shifts), and the loaders' piece map a
// subtraction.  The first level of a model still reads the leaf layer's channel-major map, the caller of a single
// level gets the layout it asks for.
//
// MODE 0 writes the level's map.  MODE 1 is the model's last sum level: its map (8 x 59 x 59 per sample for 28 x 28
// inputs, the largest of the model) is consumed in registers by the last product layer (the four taps of a root
// pixel sit in four neighbouring lanes, one DPP quad reduction) and the root layer's log-sum-exp, which leaves one
// (max, sum) pair per wave and sample for stream_root_combine_kernel.
#include "dgcspn_stream.h"
#include "ratspn_gemm_common.h"
#include <algorithm>
#include <mutex>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace dpk {

constexpr int kStreamMaxTiles = 16;
constexpr int kStreamMaxBands = 4;
constexpr int kStreamLds = 160 * 1024;
constexpr int kStreamMaxWaves = 16;     // 1024 threads; 128 VGPRs each
constexpr int kStreamC = 8;
// measured (s_memtime ticks): a wave's sample start to end when it has its SIMD to itself, the SIMD time of a sample
// when several waves share it, one DMA instruction of a loader
constexpr double kLatencyTicks = 2900, kComputeTicks = 900, kDmaTicks = 250;
constexpr int kStreamMaxDma = 24;      // DMA instructions per loader wave and sample (24 KiB)

struct StreamTile {
    int first, count, nb;               // pixels (MODE 0) / root pixels (MODE 1) of the tile; row bands staged
    int r0[kStreamMaxBands], nr[kStreamMaxBands], lb[kStreamMaxBands];   // first row, rows, float offset in the slot
};

struct StreamArgs {
    const float *in, *Wl, *LW, *LWr;
    float *out;
    const float *loc, *scale;   // INK = 2: leaf parameters [8, Cx, H, W]
    int Cx;
    int B, per_wg, slices, T, cw, nl, nst, CS, stage_bytes, K;
    long long *dbg;   // measurement only (DPK_DGC_STREAM_TIMELINE): s_memtime stamps of work-group 0
    ProdGeom q5, q6;
    StreamTile tile[kStreamMaxTiles];
};

__device__ __forceinline__ void stream_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wait until at most k of this wave's DMA instructions are in flight (a smaller immediate only waits longer)
__device__ __forceinline__ void stream_wait(int k) {
    if (k >= 48)
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (k >= 32)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (k >= 24)
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (k >= 16)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (k >= 12)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (k >= 8)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (k >= 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ float dpp_quad_sum(float v) {
    // lanes 4i .. 4i+3 all end with the sum of the four
    float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    return v + t;
}

#ifdef DPK_STREAM_TIMELINE
#define DPK_STREAM_TL 1
#else
#define DPK_STREAM_TL 0
#endif
#define STREAM_STAMP(slot)                                                                          \
    do {                                                                                           \
        if (DPK_STREAM_TL && a.dbg && bid == 0 && lane == 0 && i < 16)                                              \
            a.dbg[(wave * 16 + i) * 8 + (slot)] = (long long)__builtin_readcyclecounter();        \
    } while (0)

// wave-wide max / sum on the VALU's DPP path (no LDS round trips): quads, half rows, rows, then the row results walk
// to lane 63, which is read back as a wave-uniform value
#define DPK_DPP(x, old, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float dpp_wave_max(float v) {
    v = fmaxf(v, DPK_DPP(v, v, 0xB1, 0xf));    // quad_perm [1,0,3,2]
    v = fmaxf(v, DPK_DPP(v, v, 0x4E, 0xf));    // quad_perm [2,3,0,1]
    v = fmaxf(v, DPK_DPP(v, v, 0x141, 0xf));   // row_half_mirror
    v = fmaxf(v, DPK_DPP(v, v, 0x140, 0xf));   // row_mirror
    v = fmaxf(v, DPK_DPP(v, v, 0x142, 0xa));   // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, DPK_DPP(v, v, 0x143, 0xc));   // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v += DPK_DPP(v, 0.f, 0xB1, 0xf);
    v += DPK_DPP(v, 0.f, 0x4E, 0xf);
    v += DPK_DPP(v, 0.f, 0x141, 0xf);
    v += DPK_DPP(v, 0.f, 0x140, 0xf);
    v += DPK_DPP(v, 0.f, 0x142, 0xa);
    v += DPK_DPP(v, 0.f, 0x143, 0xc);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// (the leaf variant: at most 12 waves -- 170 registers; at 128 it spilled 8 of the 64 weight registers into the sample loop: 143 us
// against 124 for the unpacked form that fits)
constexpr int kStreamLeafWaves = 12;
template <int MODE, int INK, bool OUT_PM>
__global__ __launch_bounds__(INK == 2 ? kStreamLeafWaves * 64 : 1024) void spatial_stream_kernel(const StreamArgs a) {
    // INK: what the stage holds -- 0 the input map channel-major, 1 the input map pixel-major, 2 the IMAGE (round 6: the Gaussian
    // leaf layer folded into the model's first level: a tap's 8 leaf values are evaluated from the staged pixel and the leaf
    // parameters of its position, which the work-group keeps in LDS; the [B, 8, H, W] leaf map -- 205 MB written and read back
    // at B = 8192 -- never exists)
    constexpr bool IN_PM = INK == 1, IN_LEAF = INK == 2;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    if (DPK_STREAM_TL && a.dbg && threadIdx.x == 0) a.dbg[16 * 16 * 8 + 8 + bid * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    // the T tiles of a batch slice sit 8 work-groups apart: same XCD (work-groups go round-robin over the 8 XCDs),
    // about the same time, so the rows two tiles share are served by that XCD's L2
    const int tile_i = (bid >> 3) % a.T, slice = ((bid >> 3) / a.T) * 8 + (bid & 7);
    if (slice >= a.slices) return;
    const int s0 = slice * a.per_wg, n = min(a.per_wg, a.B - s0);
    const int nb = a.tile[tile_i].nb, first = a.tile[tile_i].first, count = a.tile[tile_i].count;
    int r0[kStreamMaxBands], nr[kStreamMaxBands], lb[kStreamMaxBands];
#pragma unroll
    for (int k = 0; k < kStreamMaxBands; ++k) {
        r0[k] = a.tile[tile_i].r0[k];
        nr[k] = a.tile[tile_i].nr[k];
        lb[k] = a.tile[tile_i].lb[k];
    }
    const int H = a.q5.H, W = a.q5.W, HW = H * W, CS = a.CS, nst = a.nst, stage_bytes = a.stage_bytes;
    const int cw = a.cw, nl = a.nl;
    const int nin = IN_LEAF ? a.Cx : kStreamC;      // channel slots of a stage
    const unsigned smem_base = (unsigned)(uintptr_t)smem;

    // pixel-major stage: the bands back to back, band k at float offset lbp[k]; 8 zero floats (the guard pixel) behind them
    int lbp[kStreamMaxBands], pm_floats = 0;
#pragma unroll
    for (int k = 0; k < kStreamMaxBands; ++k) {
        lbp[k] = pm_floats;
        if (k < nb) pm_floats += nr[k] * W * kStreamC;
    }
    // guard words (never written again: the DMA pieces of the last band end before them)
    if (IN_PM) {
        for (int i = tid; i < nst * kStreamC; i += blockDim.x)
            *(lfloat *)(smem + (i / kStreamC) * stage_bytes + 4 * (pm_floats + (i % kStreamC))) = 0.f;
    } else {
        for (int i = tid; i < nst * nin * 4; i += blockDim.x) {
            const int stg = i / (nin * 4), c = (i >> 2) % nin;
            *(lfloat *)(smem + stg * stage_bytes + 4 * (c * CS + CS - 4 + (i & 3))) = 0.f;
        }
    }
    // INK = 2: the leaf parameters of the tile's staged positions, [slot offset][image channel][mu, 1/(2 s^2), -log s - log sqrt(2 pi)][8
    // leaf channels] behind the stages; positions that hold no pixel (alignment slack, the guard) keep zeros: value 0 = log 1
    lfloat *ptab = (lfloat *)(smem + nst * stage_bytes);
    typedef __attribute__((address_space(3))) int lint;
    lint *leaf_odd = (lint *)(ptab + (IN_LEAF ? (size_t)CS * a.Cx * 24 : 0));
    if (IN_LEAF) {
        const int Cx = a.Cx;
        if (tid == 0) *leaf_odd = 0;
        __syncthreads();
        for (int e = tid; e < CS * Cx * kStreamC; e += blockDim.x) {
            const int k8 = e % kStreamC, cx = (e / kStreamC) % Cx, o = e / (kStreamC * Cx);
            float mu = 0.f, iv = 0.f, cs = 0.f;
#pragma unroll
            for (int k = 0; k < kStreamMaxBands; ++k)
                if (k < nb && o >= lb[k] && o < lb[k] + nr[k] * W) {
                    const int pin = r0[k] * W + (o - lb[k]);
                    const float sg = a.scale[((size_t)k8 * Cx + cx) * HW + pin];
                    mu = a.loc[((size_t)k8 * Cx + cx) * HW + pin];
                    iv = 0.5f / (sg * sg);
                    cs = -logf(sg) - kLogSqrt2Pi;
                }
            lfloat *row = ptab + ((size_t)o * Cx + cx) * 24;
            row[k8] = mu;
            row[8 + k8] = iv;
            row[16 + k8] = cs;
            // (parameters with which a finite pixel could still give a non-finite term -- a zero / NaN / inf scale, a
            // non-finite mean: the tile then keeps the term-by-term nan_to_num_ form for every sample)
            if (!(fabsf(mu) < 1e18f) || !(iv < 1e30f) || !(fabsf(cs) < 1e30f)) *leaf_odd = 1;
        }
    }

    if (wave >= cw) {
        // ---- loader waves ------------------------------------------------------------------------------------------
        // A stage is 8 * CS / 4 pieces of 16 bytes; DMA instruction j fills pieces 64 j .. 64 j + 63 (one per lane,
        // the LDS side of an LDS-DMA is contiguous) and belongs to loader (j mod nl).  Which global piece a lane
        // fetches (or none: alignment slack, band gaps, the guard) never changes, so it is worked out here once and
        // a sample costs the loader nothing but the DMA instructions themselves.
        const int lw = wave - cw;
        const int PPS = CS >> 2, NI = IN_PM ? ((pm_floats >> 2) + 63) >> 6 : (nin * PPS + 63) >> 6;
        unsigned voff[kStreamMaxDma];
        unsigned live = 0;   // instructions with at least one lane to fetch
        int ninstr = 0;
#pragma unroll
        for (int jj = 0; jj < kStreamMaxDma; ++jj) {
            const int j = lw + jj * nl;
            const int piece = j * 64 + lane, c = piece / PPS, f = 4 * (piece - c * PPS);   // float offset in the slot
            unsigned v = 0xffffffffu;
            if (IN_PM) {
                // pixel-major: piece `piece` of the stage is floats [4 piece, 4 piece + 4) of the bands laid back to back; a
                // band is one contiguous range of the sample's map (rows r0 .. r0 + nr - 1, 8 W floats each)
                const int fp = 4 * piece;
#pragma unroll
                for (int k = 0; k < kStreamMaxBands; ++k)
                    if (k < nb && fp >= lbp[k] && fp < lbp[k] + nr[k] * W * kStreamC)
                        v = (unsigned)(r0[k] * W * kStreamC + (fp - lbp[k])) * 4u;
            } else if (j < NI && c < nin) {
#pragma unroll
                for (int k = 0; k < kStreamMaxBands; ++k) {
                    const int gfl = c * HW + r0[k] * W, sh = gfl & 3, np = (sh + nr[k] * W + 3) >> 2;
                    if (k < nb && f >= lb[k] && f < lb[k] + 4 * np) v = (unsigned)(gfl - sh + (f - lb[k])) * 4u;
                }
            }
            voff[jj] = v;
            if (__builtin_amdgcn_ballot_w64(v != 0xffffffffu) != 0) {
                live |= 1u << jj;
                ++ninstr;
            }
        }
        const float *in = a.in;
        auto issue = [&](int i, int stg) {
            const gcchar_p sb = (gcchar_p)(in + (int64_t)(s0 + i) * nin * HW);
            const unsigned st = smem_base + stg * stage_bytes + lw * 1024;
#pragma unroll
            for (int jj = 0; jj < kStreamMaxDma; ++jj)
                if (live & (1u << jj)) {
                    if (voff[jj] != 0xffffffffu) glds16(voff[jj], sb, st + jj * nl * 1024);
                }
        };
        int pst = 0;   // stage of the next sample to issue
        for (int g = 0; g < nst - 1; ++g)
            if (g < n) {
                issue(g, pst);
                pst = (pst + 1 == nst) ? 0 : pst + 1;
            }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            // sample i has landed once only the later ones (at most nst - 2 of them) are in flight
            STREAM_STAMP(0);
            stream_wait(min(n - 1 - i, nst - 2) * ninstr);
            STREAM_STAMP(1);
            stream_barrier();   // sample i is in LDS for everyone; everyone is done with sample i - 1
            STREAM_STAMP(2);
            if (i + nst - 1 < n) {
                issue(i + nst - 1, pst);
                pst = (pst + 1 == nst) ? 0 : pst + 1;
            }
            STREAM_STAMP(3);
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------------------
    const int j = wave * 64 + lane;
    const int OHW = a.q5.OH * a.q5.OW;
    bool act;          // this thread's value is used
    int p;             // its pixel in the level's output map
    int q = 0;         // MODE 1: its root pixel
    bool root_q = false;   // MODE 1: the root pixel exists (all four lanes of its quad)
    if (MODE == 0) {
        act = j < count;
        p = first + min(j, count - 1);
    } else {
        const int ql = j >> 2, t6 = j & 3;
        const bool qv = ql < count;
        q = first + min(ql, count - 1);
        const int oh6 = q / a.q6.OW, ow6 = q - oh6 * a.q6.OW;
        const int th6 = t6 / a.q6.kw, tw6 = t6 - th6 * a.q6.kw;
        const int ph = oh6 * a.q6.sh - a.q6.pt + th6 * a.q6.dh, pw = ow6 * a.q6.sw - a.q6.pl + tw6 * a.q6.dw;
        act = qv && t6 < a.q6.kh * a.q6.kw && ph >= 0 && ph < a.q6.H && pw >= 0 && pw < a.q6.W;
        p = act ? ph * a.q5.OW + pw : 0;
        root_q = qv;
    }
    unsigned offb[4];
    {
        const int oh = p / a.q5.OW, ow = p - oh * a.q5.OW, T5 = a.q5.kh * a.q5.kw;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int th = t / a.q5.kw, tw = t - th * a.q5.kw;
            const int ih = oh * a.q5.sh - a.q5.pt + th * a.q5.dh, iw = ow * a.q5.sw - a.q5.pl + tw * a.q5.dw;
            int off = IN_PM ? pm_floats : CS - 4;   // guard: zero padding (log 1) and taps beyond kh*kw
            if (act && t < T5 && ih >= 0 && ih < H && iw >= 0 && iw < W) {
#pragma unroll
                for (int k = 0; k < kStreamMaxBands; ++k)
                    if (k < nb && ih >= r0[k] && ih < r0[k] + nr[k])
                        off = IN_PM ? lbp[k] + ((ih - r0[k]) * W + iw) * kStreamC : lb[k] + (ih - r0[k]) * W + iw;
            }
            offb[t] = 4u * (unsigned)off;
        }
    }
    unsigned Kb[kStreamC];
#pragma unroll
    for (int c = 0; c < kStreamC; ++c) Kb[c] = 4u * (unsigned)(c * CS + ((c * HW + r0[0] * W) & 3));
    gf32x2 w[kStreamC][kStreamC / 2];
#pragma unroll
    for (int o = 0; o < kStreamC; ++o)
#pragma unroll
        for (int c2 = 0; c2 < kStreamC / 2; ++c2) {
            w[o][c2].x = a.Wl[((size_t)o * kStreamC + 2 * c2) * OHW + p];
            w[o][c2].y = a.Wl[((size_t)o * kStreamC + 2 * c2 + 1) * OHW + p];
        }
    const int OHW6 = (MODE == 1) ? a.q6.OH * a.q6.OW : 0;
    // root layer: the four lanes of a quad hold the same product values; lane t6 takes output channels 2 t6, 2 t6 + 1
    float lwr[2];
    if (MODE == 1) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) lwr[jj] = root_q ? a.LWr[(2 * (j & 3) + jj) * OHW6 + q] : -INFINITY;
    }
    // The weights are consumed here, once: hipcc otherwise puts the vmcnt waits of these loads at their first use
    // INSIDE the sample loop, where they count down to vmcnt(0) in every iteration and wait for the previous sample's
    // stores.
#pragma unroll
    for (int o = 0; o < kStreamC; ++o)
#pragma unroll
        for (int c2 = 0; c2 < kStreamC / 2; ++c2) asm volatile("" : "+v"(w[o][c2].x), "+v"(w[o][c2].y));
    if (MODE == 1) asm volatile("" : "+v"(lwr[0]), "+v"(lwr[1]));
    __syncthreads();

    if (DPK_STREAM_TL && a.dbg && tid == 0) a.dbg[16 * 16 * 8 + 8 + bid * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    if (DPK_STREAM_TL && a.dbg && bid == 0 && tid == 0) {
        a.dbg[16 * 16 * 8 + 0] = (long long)__builtin_readcyclecounter();
        a.dbg[16 * 16 * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    const bool leaf_odd_tile = IN_LEAF && *leaf_odd != 0;
    int stg = 0;
    for (int i = 0; i < n; ++i) {
        STREAM_STAMP(0);
        stream_barrier();
        STREAM_STAMP(1);
        const unsigned sb = stg * stage_bytes;
        stg = (stg + 1 == nst) ? 0 : stg + 1;
        float acc[kStreamC], m = -INFINITY;
        if (IN_PM) {
            // a tap = the 8 channels of one input pixel: two 16-byte reads
            typedef __attribute__((address_space(3))) const gf32x4 lf4;
            // (channels 0-3 of the four taps first, then channels 4-7: sixteen registers of taps in flight instead of
            // thirty-two -- the MODE 1 build spilled twelve at its 128-register budget)
            gf32x4 lo[4], hi[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) lo[t] = *(lf4 *)(smem + (sb + offb[t]));
            const gf32x4 slo = (lo[0] + lo[1]) + (lo[2] + lo[3]);
            if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) hi[t] = *(lf4 *)(smem + (sb + offb[t] + 16u));
            const gf32x4 shi = (hi[0] + hi[1]) + (hi[2] + hi[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = slo[c];
                acc[4 + c] = shi[c];
            }
#pragma unroll
            for (int c = 0; c < kStreamC; ++c) m = fmaxf(m, acc[c]);
        } else if (IN_LEAF) {
            // SpatialGaussianLayer.forward (layers/dgcspn.py:101-120) at the four tap positions, then the product (their sum):
            // leaf[k] = sum over the image channels of nan_to_num(-(x - mu)^2 / (2 s^2) - log s - log sqrt(2 pi)); a NaN pixel
            // (marginalised) gives 0 through nan_to_num like the reference, a padding tap reads the guard row (all zeros)
            typedef __attribute__((address_space(3))) const gf32x4 lf4;
            const int Cx = a.Cx;
            gf32x4 alo = {0.f, 0.f, 0.f, 0.f}, ahi = {0.f, 0.f, 0.f, 0.f};
            for (int cx = 0; cx < Cx; ++cx) {
                const unsigned xb = sb + 4u * (unsigned)(cx * CS + ((cx * HW + r0[0] * W) & 3));
                float xv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[t] = *(lfloat *)(smem + (xb + offb[t]));
                // finite pixels of moderate size under sane parameters give finite terms: nan_to_num_ is the identity and
                // the eight channels go through packed arithmetic (3 instead of 8 instructions per value); a NaN (marginalised)
                // or huge pixel anywhere in the wave, or odd parameters in the tile, take the term-by-term form
                const bool plain = (fabsf(xv[0]) < 1e18f) && (fabsf(xv[1]) < 1e18f) && (fabsf(xv[2]) < 1e18f) && (fabsf(xv[3]) < 1e18f);
                const bool packed = __all(plain) && !leaf_odd_tile;       // (wave-uniform)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // (the variant's launch bound leaves 170 registers: the taps' parameter reads may overlap)
                    const lchar *row = (const lchar *)ptab + ((size_t)(offb[t] >> 2) * Cx + cx) * 96;
                    const float xt = *(lfloat *)(smem + (xb + offb[t]));        // (read again: four registers less across the taps)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {                            // leaf channels 0-3, then 4-7: twelve registers of parameters
                        const gf32x4 mu4 = *(lf4 *)(row + hf * 16), iv4 = *(lf4 *)(row + 32 + hf * 16), cs4 = *(lf4 *)(row + 64 + hf * 16);
                        // (the same arithmetic on both paths -- a sample's value must not depend on its wave-mates --, the
                        // reference's nan_to_num_ on top where a term can be non-finite)
                        const gf32x4 d4 = xt - mu4;
                        gf32x4 v4 = cs4 - (d4 * d4) * iv4;
                        if (!packed) {
#pragma unroll
                            for (int k8 = 0; k8 < 4; ++k8) v4[k8] = nan_to_num_f(v4[k8]);
                        }
                        if (hf == 0) alo += v4;
                        else ahi += v4;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = alo[c];
                acc[4 + c] = ahi[c];
            }
#pragma unroll
            for (int c = 0; c < kStreamC; ++c) m = fmaxf(m, acc[c]);
        } else {
#pragma unroll
            for (int c = 0; c < kStreamC; ++c) {
                const unsigned kc = sb + Kb[c];
                const float t0 = *(lfloat *)(smem + (kc + offb[0])), t1 = *(lfloat *)(smem + (kc + offb[1]));
                const float t2 = *(lfloat *)(smem + (kc + offb[2])), t3 = *(lfloat *)(smem + (kc + offb[3]));
                acc[c] = (t0 + t1) + (t2 + t3);
                m = fmaxf(m, acc[c]);
            }
        }
        STREAM_STAMP(2);
        const float m0 = (m == -INFINITY) ? 0.f : m;
        const float m0l = -m0 * 1.44269504088896340736f;
        gf32x2 e[kStreamC / 2];
#pragma unroll
        for (int c2 = 0; c2 < kStreamC / 2; ++c2) {
            e[c2].x = __builtin_amdgcn_exp2f(fmaf(acc[2 * c2], 1.44269504088896340736f, m0l));
            e[c2].y = __builtin_amdgcn_exp2f(fmaf(acc[2 * c2 + 1], 1.44269504088896340736f, m0l));
        }
        // the 8 output channels advance together: 8 independent FMA chains, 8 independent logs
        gf32x2 v2[kStreamC];
#pragma unroll
        for (int o = 0; o < kStreamC; ++o) v2[o] = w[o][0] * e[0];
#pragma unroll
        for (int c2 = 1; c2 < kStreamC / 2; ++c2)
#pragma unroll
            for (int o = 0; o < kStreamC; ++o) v2[o] = w[o][c2] * e[c2] + v2[o];
        float r[kStreamC], vmin = INFINITY;
#pragma unroll
        for (int o = 0; o < kStreamC; ++o) {
            const float v = v2[o].x + v2[o].y;
            vmin = fminf(vmin, v);
            r[o] = fmaf(__builtin_amdgcn_logf(v), 0.69314718055994530942f, m0);
        }
        if (__builtin_expect(vmin < 1e-30f, 0)) {
            // exact log-domain pass (rare: every term of some channel underflowed against the largest product)
#pragma unroll
            for (int o = 0; o < kStreamC; ++o)
                if (v2[o].x + v2[o].y < 1e-30f) {
                    // (the tap sums are read again from the stage: keeping them live across the hot path costs it
                    // 8 registers)
                    // (the pixel passed through an opaque copy: the address math of this rare path is otherwise hoisted out of
                    // the sample loop and its eight pointers spilled to scratch, which the kernel then pays for at every launch)
                    int pq = p;
                    asm volatile("" : "+v"(pq));
                    const float *lp = a.LW + (size_t)o * kStreamC * OHW + pq;
                    auto taps = [&](int c) {
                        if (IN_LEAF) {       // the leaf values of channel c at the four taps, once more
                            float tsum = 0.f;
                            for (int t = 0; t < 4; ++t)
                                for (int cx = 0; cx < a.Cx; ++cx) {
                                    const float xv = *(lfloat *)(smem + (sb + 4u * (unsigned)(cx * CS + ((cx * HW + r0[0] * W) & 3)) + offb[t]));
                                    const lfloat *row = ptab + ((size_t)(offb[t] >> 2) * a.Cx + cx) * 24;
                                    const float d = xv - row[c];
                                    tsum += nan_to_num_f(row[16 + c] - (d * d) * row[8 + c]);
                                }
                            return tsum;
                        }
                        const unsigned kc = IN_PM ? sb + 4u * (unsigned)c
                                                  : sb + 4u * (unsigned)(c * CS + ((c * HW + r0[0] * W) & 3));
                        return (*(lfloat *)(smem + (kc + offb[0])) + *(lfloat *)(smem + (kc + offb[1]))) +
                               (*(lfloat *)(smem + (kc + offb[2])) + *(lfloat *)(smem + (kc + offb[3])));
                    };
                    float mm = -INFINITY;
#pragma unroll 1
                    for (int c = 0; c < kStreamC; ++c) mm = fmaxf(mm, taps(c) + lp[(size_t)c * OHW]);
                    if (mm > -INFINITY) {
                        float sx = 0.f;
#pragma unroll 1
                        for (int c = 0; c < kStreamC; ++c) sx += expf(taps(c) + lp[(size_t)c * OHW] - mm);
                        r[o] = mm + logf(sx);
                    } else {
                        r[o] = -INFINITY;
                    }
                }
        }
        STREAM_STAMP(3);
        if (MODE == 0) {
            if (act) {
                float *ob = a.out + (size_t)(s0 + i) * kStreamC * OHW;   // wave-uniform base + one lane offset
                if (OUT_PM) {
                    gf32x4 *op = reinterpret_cast<gf32x4 *>(ob + (size_t)p * kStreamC);
                    op[0] = gf32x4{r[0], r[1], r[2], r[3]};
                    op[1] = gf32x4{r[4], r[5], r[6], r[7]};
                } else {
#pragma unroll
                    for (int o = 0; o < kStreamC; ++o) ob[(size_t)o * OHW + p] = r[o];
                }
            }
        } else {
            // last product layer: the four taps of a root pixel are the four lanes of a quad (padding taps add log 1)
            float P[kStreamC];
#pragma unroll
            for (int o = 0; o < kStreamC; ++o) P[o] = dpp_quad_sum(act ? r[o] : 0.f);
            const int t6l = j & 3;
            const float pa = t6l == 0 ? P[0] : t6l == 1 ? P[2] : t6l == 2 ? P[4] : P[6];
            const float pb = t6l == 0 ? P[1] : t6l == 1 ? P[3] : t6l == 2 ? P[5] : P[7];
            for (int k = 0; k < a.K; ++k) {
                float l0 = lwr[0], l1 = lwr[1];
                if (k > 0) {
                    l0 = root_q ? a.LWr[((size_t)k * kStreamC + 2 * t6l) * OHW6 + q] : -INFINITY;
                    l1 = root_q ? a.LWr[((size_t)k * kStreamC + 2 * t6l + 1) * OHW6 + q] : -INFINITY;
                }
                const float tv0 = pa + l0, tv1 = pb + l1;   // -inf outside the map of root pixels
                const float wm = dpp_wave_max(fmaxf(tv0, tv1));
                const float wml = (wm == -INFINITY) ? 0.f : -wm * 1.44269504088896340736f;
                float ts = __builtin_amdgcn_exp2f(fmaf(tv0, 1.44269504088896340736f, wml)) +
                           __builtin_amdgcn_exp2f(fmaf(tv1, 1.44269504088896340736f, wml));
                ts = dpp_wave_sum(ts);
                if (lane == 0) {
                    float *pp = a.out + ((((size_t)(s0 + i) * a.T + tile_i) * cw + wave) * a.K + k) * 2;
                    pp[0] = wm;
                    pp[1] = ts;
                }
            }
        }
        STREAM_STAMP(4);
    }
    if (DPK_STREAM_TL && a.dbg && tid == 0) a.dbg[16 * 16 * 8 + 8 + bid * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    if (DPK_STREAM_TL && a.dbg && bid == 0 && tid == 0) {
        a.dbg[16 * 16 * 8 + 2] = (long long)__builtin_readcyclecounter();
        a.dbg[16 * 16 * 8 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// out[b,k] = log-sum-exp of the (max, sum) pairs the waves of every tile left for sample b and class k: one wave per
// (b, k), lanes over the pairs
__global__ __launch_bounds__(256) void stream_root_combine_kernel(const float *__restrict__ part, int64_t B, int np,
                                                                   int K, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= B * K) return;
    const int64_t b = e / K;
    const int k = (int)(e - b * K);
    const float *pp = part + ((size_t)b * np * K + k) * 2;
    float m = -INFINITY;
    for (int i = lane; i < np; i += 64) m = fmaxf(m, pp[(size_t)i * K * 2]);
    m = wave_reduce_max(m);
    float s = 0.f;
    if (m > -INFINITY)
        for (int i = lane; i < np; i += 64) s += pp[(size_t)i * K * 2 + 1] * expf(pp[(size_t)i * K * 2] - m);
    s = wave_reduce_sum(s);
    if (lane == 0) out[e] = (m > -INFINITY) ? m + logf(s) : -INFINITY;
}

// ----------------------------------------------------------------------------------------------------------------
// host: tiling plan
// ----------------------------------------------------------------------------------------------------------------
struct StreamPlan {
    int T, cw, nl, nst, CS, stage_bytes, wg_per_cu;
    StreamTile tile[kStreamMaxTiles];
};

// Batches below this run the batch-independent kernels (a work-group of the streaming kernels needs a few dozen samples
// to amortise loading its weights).  DPK_DGC_STREAM_MIN_B overrides it, read at every call: 0 forces the streaming
// route (parity tests on small batches), a huge value disables it.
static int64_t stream_min_batch() {
    const char *e = getenv("DPK_DGC_STREAM_MIN_B");
    const long long v = e ? atoll(e) : 256;   // measured: the streaming route wins from 256 samples (0.103 vs 0.132 ms)
    return v < 0 ? 0 : v;
}

// rows of the input map touched by the tile's taps -> at most kStreamMaxBands bands whose first rows are congruent mod 4
static void tile_bands(int mode, const ProdGeom &q5, const ProdGeom *q6, StreamTile &t) {
    std::vector<char> need(q5.H, 0);
    auto mark5 = [&](int r) {
        for (int th = 0; th < q5.kh; ++th) {
            const int ih = r * q5.sh - q5.pt + th * q5.dh;
            if (ih >= 0 && ih < q5.H) need[ih] = 1;
        }
    };
    if (mode == 0) {
        for (int r = t.first / q5.OW; r <= (t.first + t.count - 1) / q5.OW; ++r) mark5(r);
    } else {
        for (int qr = t.first / q6->OW; qr <= (t.first + t.count - 1) / q6->OW; ++qr)
            for (int th6 = 0; th6 < q6->kh; ++th6) {
                const int ph = qr * q6->sh - q6->pt + th6 * q6->dh;
                if (ph >= 0 && ph < q6->H) mark5(ph);
            }
    }
    std::vector<std::pair<int, int>> runs;   // [first, last]
    for (int r = 0; r < q5.H; ++r)
        if (need[r]) {
            if (!runs.empty() && runs.back().second == r - 1)
                runs.back().second = r;
            else
                runs.push_back({r, r});
        }
    if (runs.empty()) runs.push_back({0, 0});   // a tile of padding only: stage one row, every tap points at the guard
    while ((int)runs.size() > kStreamMaxBands) {   // close the smallest gap
        size_t best = 1;
        for (size_t i = 2; i < runs.size(); ++i)
            if (runs[i].first - runs[i - 1].second < runs[best].first - runs[best - 1].second) best = i;
        runs[best - 1].second = runs[best].second;
        runs.erase(runs.begin() + best);
    }
    std::vector<std::pair<int, int>> bands;
    for (auto &r : runs) {
        int s = r.first;
        if (!bands.empty()) {
            s -= (s - bands[0].first) & 3;
            if (s <= bands.back().second + 1) {
                bands.back().second = std::max(bands.back().second, r.second);
                continue;
            }
        }
        bands.push_back({s, r.second});
    }
    t.nb = (int)bands.size();
    int lb = 0;
    for (int k = 0; k < kStreamMaxBands; ++k) {
        if (k < t.nb) {
            t.r0[k] = bands[k].first;
            t.nr[k] = bands[k].second - bands[k].first + 1;
            t.lb[k] = lb;
            lb += (int)align_up(3 + t.nr[k] * q5.W, 4);
        } else {
            t.r0[k] = t.nr[k] = t.lb[k] = 0;
        }
    }
}

static int tile_slot_floats(const StreamTile &t, int W) {
    int lb = 0;
    for (int k = 0; k < t.nb; ++k) lb += (int)align_up(3 + t.nr[k] * W, 4);
    return lb + 4;
}

static bool stream_plan_build(int mode, const ProdGeom &q5, const ProdGeom *q6, StreamPlan &best, int max_waves) {
    const int N = mode == 0 ? q5.OH * q5.OW : q6->OH * q6->OW;   // pixels / root pixels to distribute
    const int per = mode == 0 ? 1 : 4;                           // thread slots per unit
    double best_score = -1;
    const char *force = getenv("DPK_DGC_STREAM_T");   // measurement only: "<mode>:<T>" pins the tile count of a mode
    const int forced_T = (force && force[0] - '0' == mode && force[1] == ':') ? atoi(force + 2) : 0;
    for (int T = 1; T <= kStreamMaxTiles; ++T) {
        if (forced_T && T != forced_T) continue;
        StreamPlan pl;
        const int tile = cdiv(N, T);
        if (cdiv(N, tile) != T) continue;   // same tiling as a smaller T
        int cs = 0, dma = 0;
        for (int i = 0; i < T; ++i) {
            pl.tile[i].first = i * tile;
            pl.tile[i].count = std::min(tile, N - i * tile);
            tile_bands(mode, q5, q6, pl.tile[i]);
            cs = std::max(cs, tile_slot_floats(pl.tile[i], q5.W));
            int d = 0;
            for (int k = 0; k < pl.tile[i].nb; ++k) d += cdiv(cdiv(3 + pl.tile[i].nr[k] * q5.W, 4), 64);
            dma = std::max(dma, d * kStreamC);
        }
        pl.T = T;
        pl.CS = cs;
        pl.stage_bytes = kStreamC * cs * 4;
        pl.cw = cdiv(tile * per, 64);
        // loaders: an LDS-DMA instruction holds its wave for a few hundred cycles, a compute wave needs about
        // kComputeTicks per sample -- enough loaders that the copy of a sample is not the longer of the two
        const int NI = cdiv(kStreamC * cs / 4, 64);
        pl.nl = std::min(4, std::max(cdiv(NI, kStreamMaxDma), cdiv(NI, 8)));
        while (pl.nl > cdiv(NI, kStreamMaxDma) && pl.nl > 1 && pl.cw + pl.nl > max_waves) --pl.nl;
        if (pl.cw + pl.nl > max_waves || cdiv(NI, pl.nl) > kStreamMaxDma) continue;
        // the waves of a work-group go round-robin over the 4 SIMDs starting at the same one: a SIMD (4 waves of 128
        // VGPRs) ends up with ceil(waves / 4) of every resident work-group
        const int by_waves = 4 / cdiv(pl.cw + pl.nl, 4);
        const int by_lds = kStreamLds / (2 * pl.stage_bytes);
        if (by_waves < 1 || by_lds < 1) continue;
        pl.wg_per_cu = std::min(by_waves, by_lds);
        pl.nst = std::min(4, kStreamLds / (pl.wg_per_cu * pl.stage_bytes));
        // samples of the whole map per tick and CU: wg work-groups each finish 1/T of a sample per period
        const double t_compute = std::max(kLatencyTicks, kComputeTicks * cdiv(pl.wg_per_cu * pl.cw, 4));
        const double t_dma = kDmaTicks * cdiv(NI, pl.nl);
        const double score = pl.wg_per_cu / (T * std::max(t_compute, t_dma)) * (pl.nst >= 3 ? 1.0 : 0.8);
        if (score > best_score * 1.02) {
            best_score = score;
            best = pl;
        }
    }
    return best_score > 0;
}

// The plan depends on the geometry only: a model reuses a handful of them at every call (planning costs tens of
// microseconds of host time: band search over up to 16 tilings), so the last few are kept.
static bool stream_plan(int mode, const ProdGeom &q5, const ProdGeom *q6, StreamPlan &best, int max_waves = kStreamMaxWaves) {
    struct Entry {
        int mode, ok, max_waves;
        ProdGeom q5, q6;
        StreamPlan plan;
    };
    static std::mutex mu;
    static std::vector<Entry> cache;
    const char *force = getenv("DPK_DGC_STREAM_T");
    const ProdGeom q6v = q6 ? *q6 : q5;
    if (!force) {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry &e : cache)
            if (e.mode == mode && e.max_waves == max_waves && memcmp(&e.q5, &q5, sizeof(ProdGeom)) == 0 && memcmp(&e.q6, &q6v, sizeof(ProdGeom)) == 0) {
                best = e.plan;
                return e.ok != 0;
            }
    }
    const bool ok = stream_plan_build(mode, q5, q6, best, max_waves);
    if (!force) {
        std::lock_guard<std::mutex> lock(mu);
        if (cache.size() >= 64) cache.erase(cache.begin());
        cache.push_back(Entry{mode, ok ? 1 : 0, max_waves, q5, q6v, best});
    }
    return ok;
}


static bool stream_shape_ok(const ProdGeom &q, int Cout, int64_t B, const float *in) {
    return q.depthwise && q.C == kStreamC && Cout == kStreamC && q.kh * q.kw <= 4 && B >= stream_min_batch() &&
           B >= 1 && ((uintptr_t)in & 15) == 0 && (int64_t)kStreamC * q.H * q.W * 4 < ((int64_t)1 << 30);
}

bool stream_prodsum_ok(const ProdGeom &q, int Cout, int64_t B, const float *in) {
    if (!stream_shape_ok(q, Cout, B, in)) return false;
    StreamPlan pl;
    return stream_plan(0, q, nullptr, pl);
}

template <int MODE>
static int stream_launch(StreamArgs &a, const StreamPlan &pl, int64_t B, hipStream_t st, int kernel_id, bool in_pm, bool out_pm) {
    a.B = (int)B;
    a.T = pl.T;
    a.cw = pl.cw;
    a.nl = pl.nl;
    a.nst = pl.nst;
    a.CS = pl.CS;
    a.stage_bytes = pl.stage_bytes;
    for (int i = 0; i < pl.T; ++i) a.tile[i] = pl.tile[i];
    // one slice of the batch per resident work-group
    // (all T tiles of a slice run on one XCD, see the kernel: 8 XCDs, each with an eighth of the work-group slots)
    int wg_per_cu = pl.wg_per_cu;
    if (a.loc != nullptr) {   // the leaf variant: 141 registers (3 waves per SIMD), its parameter table in LDS beside the stages
        const size_t lds_leaf = (size_t)pl.nst * align_up((int64_t)a.Cx * pl.CS * 4, 16) + (size_t)pl.CS * a.Cx * 24 * 4 + 16;
        wg_per_cu = std::max(1, std::min({wg_per_cu, 3 / cdiv(pl.cw + pl.nl, 4), (int)((size_t)kStreamLds / lds_leaf)}));
    }
    int64_t slices = 8 * std::max<int64_t>(1, (int64_t)(device_cus() / 8) * wg_per_cu / pl.T);
    if (slices > B) slices = B;
    a.per_wg = (int)cdiv(B, slices);
    a.slices = cdiv(B, a.per_wg);
    void (*kern)(const StreamArgs) = nullptr;
    const bool leaf = a.loc != nullptr;
    if (MODE == 1) kern = in_pm ? spatial_stream_kernel<1, 1, false> : spatial_stream_kernel<1, 0, false>;
    else if (leaf) kern = out_pm ? spatial_stream_kernel<0, 2, true> : spatial_stream_kernel<0, 2, false>;
    else if (in_pm) kern = out_pm ? spatial_stream_kernel<0, 1, true> : spatial_stream_kernel<0, 1, false>;
    else kern = out_pm ? spatial_stream_kernel<0, 0, true> : spatial_stream_kernel<0, 0, false>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), kStreamLds)) return rc;
    size_t lds_bytes = (size_t)pl.nst * pl.stage_bytes;
    if (leaf) {   // a stage holds Cx image channels instead of 8 map channels; the parameter table sits behind the stages
        a.stage_bytes = (int)align_up((int64_t)a.Cx * pl.CS * 4, 16);
        // a stage is a kilobyte or three and ONE LDS-DMA instruction: with the plan's 4 stages the loader has 3 samples in
        // flight against several microseconds of HBM latency under load -- the level was bound by that (5.7 k cycles per
        // wave and sample for ~250 instructions); 16 stages
        const size_t tab = (size_t)pl.CS * a.Cx * 24 * 4 + 16;
        a.nst = (int)std::max<size_t>(pl.nst, std::min<size_t>(16, ((size_t)kStreamLds / wg_per_cu - tab) / a.stage_bytes));
        lds_bytes = (size_t)a.nst * a.stage_bytes + tab;
    }
    const unsigned grid = (unsigned)(pl.T * align_up(a.slices, 8));
    static const bool debug = getenv("DPK_DGC_STREAM_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "spatial_stream<%d>: in %dx%d out %dx%d  T=%d cw=%d nl=%d nst=%d CS=%d stage=%d B wg/cu=%d "
                        "per_wg=%d slices=%d grid=%u bands(tile0)=%d rows=%d\n",
                MODE, a.q5.H, a.q5.W, a.q5.OH, a.q5.OW, pl.T, pl.cw, pl.nl, pl.nst, pl.CS, pl.stage_bytes,
                pl.wg_per_cu, a.per_wg, a.slices, grid, pl.tile[0].nb,
                pl.tile[0].nr[0] + pl.tile[0].nr[1] + pl.tile[0].nr[2] + pl.tile[0].nr[3]);
    hipEvent_t pev0, pev1;
    profile_take(&pev0, &pev1, kernel_id);
    if (pev0) (void)hipEventRecord(pev0, st);
    static const bool timeline = DPK_STREAM_TL && getenv("DPK_DGC_STREAM_TIMELINE") != nullptr;
    a.dbg = nullptr;
    if (timeline) (void)hipMalloc(&a.dbg, 16 * 16 * 8 * 8 + 64 + 4096 * 32);
    if (a.dbg) (void)hipMemset(a.dbg, 0, 16 * 16 * 8 * 8 + 64 + 4096 * 32);
    DPK_LAUNCH(kern, dim3(grid), dim3((pl.cw + pl.nl) * 64), lds_bytes, st, a);
    if (pev1) (void)hipEventRecord(pev1, st);
    DPK_CHECK_LAUNCH("spatial_stream_kernel");
    if (a.dbg) {   // measurement only: synchronous read-back of work-group 0's stamps
        std::vector<long long> h(16 * 16 * 8 + 8 + 4096 * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(a.dbg);
        {
            const long long *g = h.data() + 16 * 16 * 8 + 8;
            long long first = -1;
            for (unsigned b = 0; b < grid && b < 4096; ++b)
                if (g[b * 4] && (first < 0 || g[b * 4] < first)) first = g[b * 4];
            fprintf(stderr, "timeline<%d> work-group (start, loop start, end) in us after the first start:", MODE);
            for (unsigned b = 0; b < grid && b < 4096; b += (grid > 64 ? grid / 32 : 1))
                if (g[b * 4])
                    fprintf(stderr, " %u:(%.1f %.1f %.1f)", b, (g[b * 4] - first) * 0.01, (g[b * 4 + 1] - first) * 0.01,
                            (g[b * 4 + 2] - first) * 0.01);
            fprintf(stderr, "\n");
        }
        const long long t0 = h[(0 * 16 + 0) * 8 + 0];
        fprintf(stderr, "timeline<%d> loop of work-group 0: %lld s_memtime ticks, %lld s_memrealtime ticks (100 MHz), n=%d\n", MODE,
                h[16 * 16 * 8 + 2] - h[16 * 16 * 8 + 0], h[16 * 16 * 8 + 3] - h[16 * 16 * 8 + 1], a.per_wg);
        for (int w : {0, pl.cw - 1, pl.cw}) {
            fprintf(stderr, "timeline<%d> %dx%d wave %d:", MODE, a.q5.H, a.q5.W, w);
            for (int i = 0; i < 6; ++i) {
                fprintf(stderr, "  [");
                for (int sl = 0; sl < 5; ++sl) fprintf(stderr, " %lld", h[(w * 16 + i) * 8 + sl] ? h[(w * 16 + i) * 8 + sl] - t0 : -1);
                fprintf(stderr, " ]");
            }
            fprintf(stderr, "\n");
        }
    }
    return DPK_OK;
}

int stream_prodsum_forward(const float *in, int64_t B, const ProdGeom &q, const float *Wl, const float *LW, float *out,
                           hipStream_t st, bool in_pm, bool out_pm) {
    StreamPlan pl;
    DPK_REQUIRE(stream_plan(0, q, nullptr, pl), DPK_EUNSUPPORTED, "spatial_prodsum: no streaming plan");
    StreamArgs a{};
    a.in = in;
    a.Wl = Wl;
    a.LW = LW;
    a.LWr = nullptr;
    a.out = out;
    a.K = 0;
    a.q5 = q;
    a.q6 = q;
    return stream_launch<0>(a, pl, B, st, DPK_KERNEL_SPATIAL_PRODSUM, in_pm, out_pm);
}

// the model's first level with the Gaussian leaf layer folded in (INK = 2): x [B, Cx, H, W], loc / scale [8, Cx, H, W]
bool stream_leaf_prodsum_ok(const ProdGeom &q, int Cout, int64_t B, const float *x, int Cx) {
    if (!stream_shape_ok(q, Cout, B, x) || Cx < 1 || Cx > 4) return false;
    StreamPlan pl;
    if (!stream_plan(0, q, nullptr, pl, kStreamLeafWaves)) return false;
    const size_t lds = (size_t)pl.nst * align_up((int64_t)Cx * pl.CS * 4, 16) + (size_t)pl.CS * Cx * 24 * 4 + 16;
    return lds <= (size_t)kStreamLds && pl.cw + pl.nl <= kStreamLeafWaves;
}

int stream_leaf_prodsum_forward(const float *x, const float *loc, const float *scale, int Cx, int64_t B, const ProdGeom &q,
                                const float *Wl, const float *LW, float *out, hipStream_t st, bool out_pm) {
    StreamPlan pl;
    DPK_REQUIRE(stream_plan(0, q, nullptr, pl, kStreamLeafWaves), DPK_EUNSUPPORTED, "spatial_leaf_prodsum: no streaming plan");
    StreamArgs a{};
    a.in = x;
    a.loc = loc;
    a.scale = scale;
    a.Cx = Cx;
    a.Wl = Wl;
    a.LW = LW;
    a.LWr = nullptr;
    a.out = out;
    a.K = 0;
    a.q5 = q;
    a.q6 = q;
    return stream_launch<0>(a, pl, B, st, DPK_KERNEL_SPATIAL_PRODSUM, false, out_pm);
}

int64_t stream_sumprodroot_partial_bytes(const ProdGeom &q5, int Cout, const ProdGeom &q6, int K, int64_t B) {
    if (!(q5.depthwise && q5.C == kStreamC && Cout == kStreamC && q5.kh * q5.kw <= 4 && q6.kh * q6.kw <= 4 &&
          B >= stream_min_batch() && B >= 1 && K >= 1))
        return 0;
    StreamPlan pl;
    if (!stream_plan(1, q5, &q6, pl)) return 0;
    return align_up(B * pl.T * pl.cw * K * 2 * 4, 256);
}

int stream_sumprodroot_forward(const float *in, int64_t B, const ProdGeom &q5, const float *Wl, const float *LW,
                               const ProdGeom &q6, const float *LWr, int K, float *out, void *partials,
                               hipStream_t st, bool in_pm) {
    StreamPlan pl;
    DPK_REQUIRE(stream_plan(1, q5, &q6, pl), DPK_EUNSUPPORTED, "spatial_sumprodroot: no streaming plan");
    StreamArgs a{};
    a.in = in;
    a.Wl = Wl;
    a.LW = LW;
    a.LWr = LWr;
    a.out = (float *)partials;
    a.K = K;
    a.q5 = q5;
    a.q6 = q6;
    int rc = stream_launch<1>(a, pl, B, st, DPK_KERNEL_SPATIAL_SUMPRODROOT, in_pm, false);
    if (rc) return rc;
    DPK_LAUNCH(stream_root_combine_kernel, dim3(cdiv(B * K, 4)), dim3(256), 0, st, (const float *)partials, B,
               pl.T * pl.cw, K, out);
    DPK_CHECK_LAUNCH("stream_root_combine_kernel");
    return DPK_OK;
}

}  // namespace dpk
