// RegionGraphLayer.forward with unit-scale Gaussian leaves on the matrix cores: the leaf layer ALONE, any depth /
// repetition count, I in {2, 4, 8, 16} channels -- the per-layer (autograd) route, the training forward and the
// evaluation of models wider than the fused kernels (e.g. rg_batch = rg_sum = 16) start here.
//
// reference: RegionGraphLayer.forward + GaussianLayer (deeprob/spn/layers/ratspn.py:87-108, :160-213)
//
// Same formulation as ratspn_gemm.hip (read its header first): out[b, r, k] = P[b, (r,k)] + const[(r,k)] - Q[b, r] / 2
// with P = x . M (means, split-f16 MFMA) over the natural column order n = r*I + k, and -- because the leaf outputs
// themselves are wanted -- the per-region sums of squares Q[b, r] = sum_f [f in r] x_f^2 as one more column tile: a
// GEMM of the split squares (x^2 = qh + ql) against the 0/1 region-indicator matrix.  Columns are processed in
// groups of NTG <= 4 tiles of 32 (blockIdx.y = group; the x tile is streamed once per group), so that a wave's
// accumulators (NTG + 1 tiles of 16 registers) stay in the register file.  Marginalised evidence, the per-chunk
// constants, the loader / compute wave split and the LDS ring are those of the fused kernel.  A wave that meets
// +-inf / huge evidence, or a sample whose sum of squares is beyond the expanded square's accuracy envelope, or a model
// outside it (non-unit scale, |mu| > kExpandBound) evaluates its 32 samples exactly, element by element.
#include "common.h"
#include "ratspn_gemm_common.h"
#include <math.h>

namespace dpk {

int prepare_leaf_structure(const RatWs &w, const int64_t *mask, const uint8_t *pad, int R, int d, uint32_t flags,
                           hipStream_t st);

// ------------------------------------------------------------------------------------------------
// tables: one block per region
//   table of group g, K-step ks: [NTG + 1 tiles][2 (hi, lo)][64 lanes][8 halves]; tile NTG = region indicators
// ------------------------------------------------------------------------------------------------
struct LeafPrepArgs {
    const int64_t *mask;
    const uint8_t *pad;
    const float *loc, *scale;
    int D, d, R, I, NTG, NG, NKSP, KS;
    uint16_t *mtab, *ctab;   // [NG][NKSP][NTG+1][2][512]
    float *bias;             // [NG][NCH][2][NTG][16]  per-(chunk, column) constants in accumulator order
    float *bias_row;         // [NG][2][NTG][16]
    int *elig;               // [R]
    int verify;                 // DPK_FLAG_PARAMS_VERIFY: a block rebuilds only if the bytes it depends on changed
    unsigned long long *hash;   // [R][kLeafPrepParts] fingerprint of the bytes each region's tables were built from
};

// Grid (R, kLeafPrepParts): a region's tables are a latency chain (fingerprint -> variable positions -> fragments /
// constants), 17 us on R = 32 work-groups when one work-group did all of it (round-4 trace of the training step, where every
// launch rebuilds).  Parts 0 .. P-2 share the fragment stores, part P-1 owns the per-chunk constants; every part keeps its
// own fingerprint word (a part that finished must not make its neighbours skip their share).
__global__ __launch_bounds__(256) void ratspn_leaf_gemm_prep_kernel(const LeafPrepArgs a) {
    extern __shared__ int featpos[];                            // [D] position j of variable f in this region, -1 if absent
    float *locs = reinterpret_cast<float *>(featpos + a.D);     // [I][d]
    float *csum = locs + a.I * a.d;                             // [NCH][I]
    __shared__ int bad_s;
    __shared__ unsigned long long red_s[17];
    constexpr int P = kLeafPrepParts;
    const int r = blockIdx.x, part = blockIdx.y, D = a.D, d = a.d, I = a.I, NTG = a.NTG;
    // Every global read of the region's parameters is issued up front and used twice (fingerprint + tables): the earlier
    // form (fingerprint loops, then `if (pad[j]) continue; .. scale[o]` per element) was ~25 dependent round trips to
    // memory, 17 us for 12 KB.  Fingerprint = sum of fp_word(word, position * 8 + tag) as in fp_range, over mask (tag 1),
    // pad_mask (2, one byte per word), loc (3), scale (4).
    unsigned char *padl = reinterpret_cast<unsigned char *>(csum + ((D + 16 * a.KS - 1) / (16 * a.KS)) * I);   // [d]
    const unsigned long long stored = a.verify ? a.hash[r * P + part] : 0ull;   // (requested first: back when the hash is)
    unsigned long long hsum = 0ull;
    const int tid = threadIdx.x, n = I * d;
    const unsigned *wl = reinterpret_cast<const unsigned *>(a.loc + (int64_t)r * n);
    const unsigned *wsc = reinterpret_cast<const unsigned *>(a.scale + (int64_t)r * n);
    constexpr int KB = 8;
    unsigned vl[KB], vs[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int e = k * 256 + tid;
        vl[k] = e < n ? wl[e] : 0u;
        vs[k] = e < n ? wsc[e] : 0u;
    }
    for (int f = tid; f < D; f += 256) featpos[f] = -1;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    for (int j0 = 0; j0 < d; j0 += 512) {
        long long mk[2];
        unsigned pd[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = j0 + k * 256 + tid;
            mk[k] = j < d ? a.mask[(int64_t)r * d + j] : -1;
            pd[k] = (j < d && a.pad != nullptr) ? a.pad[(int64_t)r * d + j] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = j0 + k * 256 + tid;
            if (j < d) {
                hsum += fp_word((unsigned)mk[k], (unsigned)j * 16u + 1u) + fp_word((unsigned)(mk[k] >> 32), (unsigned)j * 16u + 9u) +
                        fp_word(pd[k], (unsigned)j * 8u + 2u);
                padl[j] = (unsigned char)(pd[k] != 0u);
                if (!pd[k] && mk[k] >= 0 && mk[k] < D) featpos[(int)mk[k]] = j;
            }
        }
    }
    __syncthreads();
    bool bad = false;
    for (int e0 = 0; e0 < n; e0 += KB * 256) {
        if (e0 > 0) {
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int e = e0 + k * 256 + tid;
                vl[k] = e < n ? wl[e] : 0u;
                vs[k] = e < n ? wsc[e] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = e0 + k * 256 + tid;
            if (e < n) {
                hsum += fp_word(vl[k], (unsigned)e * 8u + 3u) + fp_word(vs[k], (unsigned)e * 8u + 4u);
                const float mu = __uint_as_float(vl[k]);
                locs[e] = mu;
                if (!padl[e % d]) bad = bad || !(fabsf(mu) <= kExpandBound) || (__uint_as_float(vs[k]) != 1.0f);
            }
        }
    }
    hsum = block_sum_u64(hsum + 0x9E3779B97F4A7C15ull * (tid == 0 ? (unsigned long long)(r + 1) : 0ull), red_s);
    if (a.verify && stored == hsum) return;
    if (tid == 0) a.hash[r * P + part] = hsum;
    if (bad) bad_s = 1;
    __syncthreads();
    const int tile_cols = 32 * NTG;
    const int64_t tile_halves = 2 * 512, ks_halves = (int64_t)(NTG + 1) * tile_halves;
    if (part < P - 1) {
        const int tid = part * (int)blockDim.x + (int)threadIdx.x, nth = (P - 1) * (int)blockDim.x;
        // mean / constant fragments: (K-step, lane half, channel) -> 8 consecutive variables
        for (int e = tid; e < a.NKSP * 2 * I; e += nth) {
            const int k = e % I, hg = (e / I) & 1, ks = e / (2 * I);
            const int n = r * I + k, g = n / tile_cols, nl = n - g * tile_cols;
            const int t = nl >> 5, idx = nl & 31, h = idx >> 4, u = idx & 15;
            const int row = (u & 3) + 8 * (u >> 2) + 4 * h;
            half8 mh, ml, ch, cl;
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int f = ks * 16 + hg * 8 + el;
                float mu = 0.f, cc = 0.f;
                if (f < D) {
                    const int j = featpos[f];
                    if (j >= 0) {
                        mu = locs[k * d + j];
                        cc = -fmaf(0.5f * mu, mu, kLogSqrt2Pi);
                    }
                }
                _Float16 hi, lo;
                split_f16(mu, hi, lo);
                mh[el] = hi; ml[el] = lo;
                split_f16(cc, hi, lo);
                ch[el] = hi; cl[el] = lo;
            }
            const int64_t o = ((int64_t)g * a.NKSP + ks) * ks_halves + t * tile_halves + (hg * 32 + row) * 8;
            *reinterpret_cast<half8 *>(a.mtab + o) = mh;
            *reinterpret_cast<half8 *>(a.mtab + o + 512) = ml;
            *reinterpret_cast<half8 *>(a.ctab + o) = ch;
            *reinterpret_cast<half8 *>(a.ctab + o + 512) = cl;
        }
        // the region's row of the indicator tile (tile NTG of its group): lane half h owns the regions whose columns it
        // holds, in the order it meets them
        const int g = (r * I) / tile_cols, nl = r * I - g * tile_cols;
        const int t = nl >> 5, h = (nl & 31) >> 4, j = (nl & 15) / I;
        const int up = t * (16 / I) + j;
        const int row = (up & 3) + 8 * (up >> 2) + 4 * h;
        for (int e = tid; e < a.NKSP * 2; e += nth) {
            const int hg = e & 1, ks = e >> 1;
            half8 ind;
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int f = ks * 16 + hg * 8 + el;
                ind[el] = (f < D && featpos[f] >= 0) ? (_Float16)1.0f : (_Float16)0.0f;
            }
            const int64_t o = ((int64_t)g * a.NKSP + ks) * ks_halves + NTG * tile_halves + (hg * 32 + row) * 8;
            *reinterpret_cast<half8 *>(a.mtab + o) = ind;   // (the lo half of this tile stays zero: memset by the host)
        }
        return;
    }
    // per-(chunk, channel) constants and their sum, fixed summation order (launches must agree bit for bit): 16 lanes
    // per constant, lane l takes variables l, l + 16, .. of the chunk, then a butterfly over the 16 lanes
    const int KC = 16 * a.KS, NCH = (D + KC - 1) / KC;
    const int sub = threadIdx.x & 15;
    for (int e0 = 0; e0 < NCH * I; e0 += (int)blockDim.x / 16) {
        const int e = e0 + ((int)threadIdx.x >> 4);
        const bool live = e < NCH * I;
        const int k = live ? e % I : 0, c = live ? e / I : 0;
        float sum = 0.f;
        const int f1 = min(D, (c + 1) * KC);
        if (live)
            for (int f = c * KC + sub; f < f1; f += 16) {
                const int j = featpos[f];
                if (j >= 0) {
                    const float mu = locs[k * d + j];
                    sum -= fmaf(0.5f * mu, mu, kLogSqrt2Pi);
                }
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (live && sub == 0) {
            const int n = r * I + k, g = n / tile_cols, nl = n - g * tile_cols;
            const int t = nl >> 5, idx = nl & 31, h = idx >> 4, u = idx & 15;
            a.bias[((((int64_t)g * NCH + c) * 2 + h) * NTG + t) * 16 + u] = sum;
            csum[e] = sum;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < I) {
        const int k = threadIdx.x;
        float sum = 0.f;
        for (int c = 0; c < NCH; ++c) sum += csum[c * I + k];
        const int n = r * I + k, g = n / tile_cols, nl = n - g * tile_cols;
        const int t = nl >> 5, idx = nl & 31, h = idx >> 4, u = idx & 15;
        a.bias_row[(((int64_t)g * 2 + h) * NTG + t) * 16 + u] = sum;
    }
    if (threadIdx.x == 0) a.elig[r] = bad_s ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct LeafGemmArgs {
    const float *x;
    int64_t B;
    int D, d, R, I, NCH, ntiles, NKSP;
    const uint16_t *mtab, *ctab;
    const float *biasC, *biasT;
    const int *elig;
    float *out;   // [B, R, I]
    const int64_t *mask;
    const uint8_t *pad;
    const float *loc, *scale;
    int ablate;   // measurement only (DPK_LEAF_ABLATE): 1 = no output stores
};

// exact per-element evaluation of the wave's 32 samples for the columns of this group (any scale, any evidence)
template <int I>
__device__ __noinline__ void leaf_exact_wave(const LeafGemmArgs &a, int64_t bw0, int lane, int r0, int r1) {
    const int s = lane & 31, h = lane >> 5;
    const int64_t b = bw0 + s;
    if (b >= a.B) return;
    const float *xr = a.x + b * a.D;
    const int d = a.d;
    for (int r = r0 + h; r < r1; r += 2) {
        float acc[I];
#pragma unroll
        for (int k = 0; k < I; ++k) acc[k] = 0.f;
        for (int j = 0; j < d; ++j) {
            const int64_t o = (int64_t)r * d + j;
            if (a.pad != nullptr && a.pad[o]) continue;
            const float xv = xr[a.mask[o]];
#pragma unroll
            for (int k = 0; k < I; ++k) {
                const int64_t po = ((int64_t)r * I + k) * d + j;
                const float mu = a.loc[po], sg = a.scale[po];
                const float dlt = xv - mu;
                acc[k] += nan_to_num_f(fmaf(dlt * dlt, -0.5f / (sg * sg), -logf(sg) - kLogSqrt2Pi));
            }
        }
#pragma unroll
        for (int k = 0; k < I; ++k) a.out[(b * a.R + r) * I + k] = acc[k];
    }
}

template <int I, int NTG>
__global__ __launch_bounds__(2 * kGemmWaves * 64) void ratspn_leaf_gemm_kernel(const LeafGemmArgs a) {
    constexpr int KS = kLeafGemmKS;
    constexpr int KC = 16 * KS, W = 4 * KS, ROWB = KC * 4;
    constexpr int SWS = (W == 16) ? 0 : (W == 8 ? 1 : 2);
    constexpr int XB = kGemmTile * ROWB;
    constexpr int TKB = KS * (2 * NTG + 1);          // table KiB per chunk (the indicator tile's zero half is not staged)
    constexpr int KSB = (2 * NTG + 2) * 1024;        // table bytes per K-step in global memory
    constexpr int STAGE = XB + KS * KSB;             // (LDS keeps the global layout; the unstaged KiB is simply unused)
    constexpr int NS = kGemmStages;
    constexpr int RPL = 16 / I;                      // regions per lane half and tile
    static_assert(I == 2 || I == 4 || I == 8 || I == 16, "channels");
    static_assert(NTG * RPL <= 16, "one indicator tile per group");
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) const half8 lh8;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= kGemmWaves;
    const int wave = wave8 & (kGemmWaves - 1);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D, NCH = a.NCH;
    const int grid = (int)gridDim.x, ntiles = a.ntiles;
    const int g = (int)blockIdx.y;                                   // column group
    const int64_t gtab = (int64_t)g * a.NKSP * KSB;

    if (loader) {
        // the KS K-steps of a chunk are contiguous in the table; every K-step's last KiB (zero half of the indicator
        // tile) is skipped: wave w copies KiB pieces w, w+4, ... of the chunk's staged pieces
        // -> simplest exact split: pieces are dealt in blocks per wave
        constexpr int Q4 = TKB / 4, R4 = TKB % 4;
        // piece p (0 .. TKB-1) of a chunk lives at byte (p / (2 NTG + 1)) * KSB + (p % (2 NTG + 1)) * 1024
        // (handled inside the loader through a per-piece offset table would cost registers; instead the table chunk
        // is staged WITH the zero KiB when that divides evenly)
        (void)Q4; (void)R4;
        constexpr int PBF = KS * (2 * NTG + 2) / 4;                  // pieces per wave, zero KiB included
        static_assert((KS * (2 * NTG + 2)) % 4 == 0, "table chunk must split over the loader waves");
        gemm_loader_run<KS, PBF>(a.x, a.B, D, NCH, ntiles, (int)blockIdx.x, grid, (gcchar_p)a.mtab + gtab, KS * KSB,
                                 wave * PBF, (unsigned)(uintptr_t)smem, STAGE, wave, lane);
        return;
    }
    // ================================================ compute waves =========================================
    const int tile_cols = 32 * NTG;
    const int r_lo = g * tile_cols / I, r_hi = min(a.R, (g + 1) * tile_cols / I);
    bool model_ok = true;
    for (int e = r_lo + lane; e < r_hi; e += 64) model_ok = model_ok && (a.elig[e] != 0);
    model_ok = __all(model_ok);
    __syncthreads();

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

    int cstage = 0;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
        gf32x16 acc[NTG + 1];
#pragma unroll
        for (int t = 0; t <= NTG; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        float qsum = 0.f;
        bool need_exact = false;
        unsigned odd_lo = 0u, odd_hi = 0u;   // chunks (< 64) whose constants the validity GEMM accumulated
        for (int c = 0; c < NCH; ++c) {
            gemm_lds_barrier();
            const lchar *st = smem + cstage * STAGE;
            cstage = (cstage + 1 == NS) ? 0 : cstage + 1;
            const lchar *tb = st + foff;
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
            if (partial) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int f0 = c * KC + ks * 16 + h * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[ks][i] = (f0 + i < D) ? v[ks][i] : 0.f;
                }
            }
            float tq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) tq = fmaf(v[ks][i], v[ks][i], tq);
            const bool odd_chunk = __any(!(tq < kGemmStepBound));
            half8 valid[KS];
            if (odd_chunk) {
                if (c < 32) odd_lo |= 1u << c; else odd_hi |= 1u << (c - 32);
                tq = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float vi = v[ks][i];
                        const bool isn = vi != vi;
                        const bool big = !isn && !(fabsf(vi) < kGemmAbsBound);
                        need_exact = need_exact || big;
                        v[ks][i] = (isn || big) ? 0.f : vi;
                        valid[ks][i] = isn ? (_Float16)0.0f : (_Float16)1.0f;
                        tq = fmaf(v[ks][i], v[ks][i], tq);
                    }
            }
            qsum += tq;
            const int nks = min(KS, (D - c * KC + 15) >> 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < nks) {
                    half8 xh, xl, qh, ql;
                    split8(v[ks], xh, xl);
                    float sq[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) sq[i] = v[ks][i] * v[ks][i];
                    split8(sq, qh, ql);
                    const lchar *tk = tb + ks * KSB;
                    const half8 ind = *(lh8 *)(tk + NTG * 2048);
                    acc[NTG] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ind, qh, acc[NTG], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NTG; ++t) {
                        const half8 mh = *(lh8 *)(tk + t * 2048);
                        const half8 ml = *(lh8 *)(tk + t * 2048 + 1024);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, xh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, xl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml, xh, acc[t], 0, 0, 0);
                    }
                    acc[NTG] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ind, ql, acc[NTG], 0, 0, 0);
                    if (odd_chunk) {
                        typedef const __attribute__((address_space(1))) half8 gh8;
                        const gcchar_p cb = (gcchar_p)a.ctab + gtab + (int64_t)(c * KS + ks) * KSB + lane * 16;
#pragma unroll
                        for (int t = 0; t < NTG; ++t) {
                            const half8 ch = *(gh8 *)(cb + t * 2048);
                            const half8 cl = *(gh8 *)(cb + t * 2048 + 1024);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, valid[ks], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, valid[ks], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- outputs of the tile -----------------------------------------------------------------------------
        const int64_t bw0 = (int64_t)tile * kGemmTile + wave * 32;
        const int64_t b = bw0 + s;
        const float qtot = qsum + __shfl_xor(qsum, 32, 64);
        const bool lane_exact = need_exact || !(qtot <= kExpandBound * kExpandBound * (float)D);
        if (!model_ok || __any(lane_exact)) {
            const LeafGemmArgs ac = a;
            leaf_exact_wave<I>(ac, bw0, lane, r_lo, r_hi);
            continue;
        }
        const int ncol = a.R * I;
        const bool clean = (odd_lo | odd_hi) == 0u;
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            float cst[16];
            if (clean) {
                const float *bc = a.biasT + (((int64_t)g * 2 + h) * NTG + t) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) cst[i] = bc[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) cst[i] = 0.f;
                for (int c = 0; c < NCH; ++c) {
                    const bool odd = c < 32 ? ((odd_lo >> c) & 1u) : ((odd_hi >> (c - 32)) & 1u);
                    if (odd) continue;
                    const float *bc = a.biasC + ((((int64_t)g * NCH + c) * 2 + h) * NTG + t) * 16;
#pragma unroll
                    for (int i = 0; i < 16; ++i) cst[i] += bc[i];
                }
            }
            const int n0 = g * tile_cols + 32 * t + 16 * h;   // first column of this lane's 16
            if (b < a.B && n0 < ncol) {
                float *o = a.out + b * ncol + n0;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    gf32x4 w4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = 4 * q4 + i;
                        w4[i] = acc[t][u] + cst[u] - 0.5f * acc[NTG][t * RPL + u / I];
                    }
                    if (n0 + 4 * q4 < ncol && !(a.ablate & 1)) *reinterpret_cast<gf32x4 *>(o + 4 * q4) = w4;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int I, int NTG>
static int leaf_gemm_launch(const LeafGemmArgs &a, int NG, hipStream_t st) {
    constexpr int KS = kLeafGemmKS;
    const size_t lds = (size_t)kGemmStages * (kGemmTile * 64 * KS + KS * (2 * NTG + 2) * 1024);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "leaf_gemm: %zu bytes of LDS", lds);
    auto kern = ratspn_leaf_gemm_kernel<I, NTG>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    const int per_group = cus / NG > 0 ? cus / NG : 1;
    const int gx = a.ntiles < per_group ? a.ntiles : per_group;
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_RATSPN_LEAF);
    if (ev0) (void)hipEventRecord(ev0, st);
    DPK_LAUNCH(kern, dim3(gx, NG), dim3(2 * kGemmWaves * 64), lds, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_leaf_gemm_kernel");
    return DPK_OK;
}

// `ws` = the leaf_gemm_ws_bytes() segment of the caller's workspace
int ratspn_leaf_gemm_forward(void *ws, const float *x, int64_t B, int D, const int64_t *mask, const uint8_t *pad,
                             const float *loc, const float *scale, int R, int I, int d, float *out, uint32_t flags,
                             hipStream_t st) {
    const int NTG = leaf_ntg(I), NG = cdiv((int64_t)R * I, 32 * NTG);
    const int KS = kLeafGemmKS;
    const int NCH = cdiv(D, 16 * KS), NKSP = NCH * KS;
    const int64_t tab = align_up((int64_t)NG * NKSP * (2 * NTG + 2) * 1024, 256);
    const int64_t bias = align_up((int64_t)NG * NCH * 2 * NTG * 16 * 4, 256);
    const int64_t brow = align_up((int64_t)NG * 2 * NTG * 16 * 4, 256);
    char *p = (char *)ws;
    uint16_t *mtab = (uint16_t *)p, *ctab = (uint16_t *)(p + tab);
    float *biasC = (float *)(p + 2 * tab), *biasT = (float *)(p + 2 * tab + bias);
    int *elig = (int *)(p + 2 * tab + bias + brow);
    if (!(flags & DPK_FLAG_PARAMS_CACHED)) {
        // The slots no region block writes (padding columns of the last tile, the zero half of the indicator tile) are
        // zeroed by the unconditional build; a verifying call keeps them (every region block rewrites all of its own)
        const bool verify = (flags & DPK_FLAG_PARAMS_VERIFY) != 0;
        if (!verify)
            DPK_REQUIRE(hipMemsetAsync(p, 0, (size_t)(2 * tab + bias + brow), st) == hipSuccess, DPK_ELAUNCH, "memset");
        LeafPrepArgs pa{};
        pa.verify = verify ? 1 : 0;
        pa.hash = (unsigned long long *)(p + 2 * tab + bias + brow + align_up((int64_t)R * 4, 256));
        pa.mask = mask; pa.pad = pad; pa.loc = loc; pa.scale = scale;
        pa.D = D; pa.d = d; pa.R = R; pa.I = I; pa.NTG = NTG; pa.NG = NG; pa.NKSP = NKSP; pa.KS = KS;
        pa.mtab = mtab; pa.ctab = ctab; pa.bias = biasC; pa.bias_row = biasT; pa.elig = elig;
        const size_t lds = ((size_t)D + (size_t)I * d + (size_t)NCH * I) * 4 + (size_t)align_up(d, 4);
        DPK_REQUIRE(lds <= 60 * 1024, DPK_EUNSUPPORTED, "leaf_gemm: in_features=%d too large for the table kernel", D);
        DPK_LAUNCH(ratspn_leaf_gemm_prep_kernel, dim3(R, kLeafPrepParts), dim3(256), lds, st, pa);
        DPK_CHECK_LAUNCH("ratspn_leaf_gemm_prep_kernel");
    }
    LeafGemmArgs a{};
    {
        static const int ab = getenv("DPK_LEAF_ABLATE") ? atoi(getenv("DPK_LEAF_ABLATE")) : 0;
        a.ablate = ab;
    }
    a.x = x; a.B = B; a.D = D; a.d = d; a.R = R; a.I = I; a.NCH = NCH; a.NKSP = NKSP;
    a.ntiles = cdiv(B, kGemmTile);
    a.mtab = mtab; a.ctab = ctab; a.biasC = biasC; a.biasT = biasT; a.elig = elig;
    a.out = out; a.mask = mask; a.pad = pad; a.loc = loc; a.scale = scale;
    switch (I) {
        case 2: return leaf_gemm_launch<2, 2>(a, NG, st);
        case 4: return leaf_gemm_launch<4, 4>(a, NG, st);
        case 8: return leaf_gemm_launch<8, 4>(a, NG, st);
        case 16: return leaf_gemm_launch<16, 4>(a, NG, st);
    }
    set_error("leaf_gemm: channels=%d not built", I);
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk
