// RealNVP-2D evaluation path: the convolutional conditioners (weight-normalised 3x3 / 1x1 convolutions
// with eval-mode BatchNorm2d + ReLU folded into the operand load), the checkerboard / channel-wise coupling
// transformation, the BatchNormLayer2d bijector and the squeeze / multi-scale permutations.
//   reference: flows/layers/coupling.py:107-272 (CouplingLayer2d), :275-408 (CouplingBlock2d),
//              flows/layers/resnet.py:9-90, flows/layers/densenet.py, torch/utils.py:86-121 (WeightNormConv2d),
//              flows/utils.py:11-38 (squeeze), :165-222 (BatchNormLayer2d), flows/models/realnvp.py:75-220 (RealNVP2d)
//
// Convolutions, fp32 throughout (the 1e-5 parity bar): 3x3 layers on v_mfma_f32_32x32x2_f32 with the activations staged
// through LDS (conv3x3_lds_kernel), 1x1 layers on the same MFMA with operands from global memory (conv2d_mfma_kernel),
// masked / narrow layers on the vector ALUs (conv2d_kernel: a thread owns 4 pixels x 16 output channels, wave-uniform
// weights arrive through the scalar unit and feed v_pk_fma_f32 as SGPR pairs).  The conditioners are compute-bound
// (14.4 MFLOP per 3x3 convolution of 32 channels on 28x28 against 300 KB of activation traffic).
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace dpk {

constexpr int kConvCO = 16;   // output channels per thread
constexpr int kConvPix = 4;   // pixels per thread (one row segment)
typedef float float2_t __attribute__((ext_vector_type(2)));

// ---- weight normalisation + packing -------------------------------------------------------------------------------
// w[co,ci,ky,kx] = g[co] * v[co,ci,ky,kx] / ||v[co]||  (torch.nn.utils.weight_norm, dim 0) -> wpack[ci][tap][CoutPad]
// block co < Cout: one output channel; blocks Cout..CoutPad-1 write zeros; the last block folds the BatchNorm2d that
// precedes the convolution into pre[0:Cin] = gamma / sqrt(var + eps), pre[Cin:2Cin] = beta - mean * pre[0:Cin].
__global__ __launch_bounds__(256) void conv2d_prepare_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                                             int Cout, int CoutPad, int Cin, int taps,
                                                             const float *__restrict__ bn_w,
                                                             const float *__restrict__ bn_b,
                                                             const float *__restrict__ bn_mean,
                                                             const float *__restrict__ bn_var, float bn_eps,
                                                             float *__restrict__ wpack, float *__restrict__ pre) {
    const int co = blockIdx.x, n = Cin * taps;
    if (co == CoutPad) {
        for (int c = threadIdx.x; c < Cin; c += 256) {
            const float a = bn_w[c] / sqrtf(bn_var[c] + bn_eps);
            pre[c] = a;
            pre[Cin + c] = bn_b[c] - bn_mean[c] * a;
        }
        return;
    }
    if (co >= Cout) {
        for (int i = threadIdx.x; i < n; i += 256) wpack[(int64_t)i * CoutPad + co] = 0.f;
        return;
    }
    const float *vc = v + (int64_t)co * n;
    float scale = 1.f;
    if (g) {
        __shared__ float part[256];
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s = fmaf(vc[i], vc[i], s);
        part[threadIdx.x] = s;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
            __syncthreads();
        }
        scale = g[co] / sqrtf(part[0]);
    }
    for (int i = threadIdx.x; i < n; i += 256) wpack[(int64_t)i * CoutPad + co] = vc[i] * scale;
}

// ---- convolution ---------------------------------------------------------------------------------------------------
struct Conv2dArgs {
    const float *in;
    int64_t in_bs;
    int B, Cin, H, W;
    const float *w;
    int Cout, CoutPad;
    const float *pre;   // [2*Cin] or null: relu(a*x + b) applied to the operand (zero padding stays zero)
    const float *mask;  // [H*W] or null: multiplies the operand (checkerboard coupling mask)
    const float *bias;  // [Cout] or null
    const float *res;   // added to the result, or null
    int64_t res_bs;
    float *out;
    int64_t out_bs;
};

template <int KS, bool PRE, bool MASK>
__global__ __launch_bounds__(256) void conv2d_kernel(Conv2dArgs a) {
    constexpr int P = KS / 2, NV = kConvPix + KS - 1, CO = kConvCO, NL = KS * NV;
    const int QW = (a.W + kConvPix - 1) / kConvPix;
    const int per = a.H * QW;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(t / per);
    if (b >= a.B) return;
    const int rem = (int)(t - (int64_t)b * per);
    const int y = rem / QW, x0 = (rem - y * QW) * kConvPix;
    const int co0 = blockIdx.y * CO;
    const int HW = a.H * a.W;

    // the NL = KS * (4 + KS - 1) operand positions of this thread: clamped offsets (every load is in bounds and
    // unconditional, so the loads of a channel go out back to back) and the padding / mask factor of each
    int off[NL];
    float fac[NL];   // MASK builds only
    bool okb[NL];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        const int yy = y + ky - P;
        const int yc = min(max(yy, 0), a.H - 1);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int xx = x0 + j - P;
            const int xc = min(max(xx, 0), a.W - 1);
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            off[ky * NV + j] = yc * a.W + xc;
            okb[ky * NV + j] = ok;
            fac[ky * NV + j] = ok ? (MASK ? a.mask[yc * a.W + xc] : 1.f) : 0.f;
        }
    }

    // accumulators as pairs of neighbouring output channels: one v_pk_fma_f32 per pair (weights = an SGPR pair, the
    // operand broadcast to both halves)
    float2_t acc[kConvPix][CO / 2];
#pragma unroll
    for (int co = 0; co < CO; co += 2) {
        float2_t bv;
        bv.x = (a.bias && co0 + co < a.Cout) ? a.bias[co0 + co] : 0.f;
        bv.y = (a.bias && co0 + co + 1 < a.Cout) ? a.bias[co0 + co + 1] : 0.f;
#pragma unroll
        for (int p = 0; p < kConvPix; ++p) acc[p][co / 2] = bv;
    }
    const float *ip = a.in + (int64_t)b * a.in_bs;
    const float *wp = a.w + co0;
    float nxt[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) nxt[i] = ip[off[i]];
    // weights: the 16 floats of a tap are wave-uniform (scalar loads); the next tap's are requested before this tap's
    // FMAs so that the scalar-cache latency is covered by them
    float2_t wc[CO / 2], wn[CO / 2];
#pragma unroll
    for (int co = 0; co < CO; co += 2) {
        wc[co / 2].x = wp[co];
        wc[co / 2].y = wp[co + 1];
    }
    for (int ci = 0; ci < a.Cin; ++ci) {
        float v[NL];
        float pa = 1.f, pb = 0.f;
        if (PRE) {
            pa = a.pre[ci];
            pb = a.pre[a.Cin + ci];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float r = nxt[i];
            if (PRE) r = fmaxf(fmaf(r, pa, pb), 0.f);
            v[i] = MASK ? r * fac[i] : (okb[i] ? r : 0.f);
        }
        // the next channel's operands are requested before this channel's FMAs (the last iteration re-reads channel
        // Cin - 1: in bounds, unused)
        const bool more = ci + 1 < a.Cin;
        ip += more ? HW : 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) nxt[i] = ip[off[i]];
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            // (after the last tap of the last channel: re-reads that tap, unused)
            wp += (tap + 1 < KS * KS || more) ? a.CoutPad : 0;
#pragma unroll
            for (int co = 0; co < CO; co += 2) {
                wn[co / 2].x = wp[co];
                wn[co / 2].y = wp[co + 1];
            }
#pragma unroll
            for (int co = 0; co < CO; co += 2) {
#pragma unroll
                for (int p = 0; p < kConvPix; ++p) {
                    float2_t vv;
                    vv.x = vv.y = v[ky * NV + p + kx];
                    acc[p][co / 2] = __builtin_elementwise_fma(vv, wc[co / 2], acc[p][co / 2]);
                }
            }
#pragma unroll
            for (int co = 0; co < CO / 2; ++co) wc[co] = wn[co];
        }
    }
    const int64_t pix = (int64_t)y * a.W + x0;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co0 + co >= a.Cout) break;
        float *op = a.out + (int64_t)b * a.out_bs + (int64_t)(co0 + co) * HW + pix;
        const float *rp = a.res ? a.res + (int64_t)b * a.res_bs + (int64_t)(co0 + co) * HW + pix : nullptr;
#pragma unroll
        for (int p = 0; p < kConvPix; ++p) {
            const float r = (co & 1) ? acc[p][co / 2].y : acc[p][co / 2].x;
            if (x0 + p < a.W) op[p] = rp ? r + rp[p] : r;
        }
    }
}

// ---- convolution on the matrix cores ------------------------------------------------------------------------------
// D[cout][pixel] = sum_k W[cout][k] X[k][pixel] with v_mfma_f32_32x32x2_f32 (fp32 products and sums: the 1e-5 parity
// rules out narrower operands); k runs over (input-channel pair, tap).  Lane l supplies W[cout = l & 31][k-half = l >> 5]
// from the fragment-ordered table (one coalesced dword per k-step, straight from L2) and X[k-half][pixel = l & 31]: its own
// pixel of channel 2*jp + (l >> 5), one tap per k-step, BatchNorm2d + ReLU + zero padding applied in registers.  A wave
// owns 64 consecutive pixels of the flattened [B, H*W] axis (two 32-pixel tiles) times 32 output channels; the operands
// of the next channel pair are requested before the 18 MFMAs of the current one.  Results leave as 128-byte runs (32
// consecutive pixels of one channel per store).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int kMfmaTiles = 2;
// DPK_C2D_ABLATE (measurement builds only): 1 = activations loaded for the first channel pair only, 2 = one MFMA per tile
// and pair instead of nine, 4 = weight fragments loaded for the first pair only
#ifndef DPK_C2D_ABLATE
#define DPK_C2D_ABLATE 0
#endif

// fragment order: wfrag[((cog * nJ + jp) * taps + tap) * 64 + lane] = w[co = cog*32 + (lane & 31)][ci = 2*jp + (lane >> 5)][tap]
__global__ __launch_bounds__(256) void conv2d_frag_kernel(const float *__restrict__ wpack, int Cout, int CoutPad, int Cin,
                                                          int taps, float *__restrict__ wfrag, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int lane = (int)(e & 63);
    const int64_t q = e >> 6;
    const int tap = (int)(q % taps);
    const int nJ = (Cin + 1) / 2;
    const int jp = (int)((q / taps) % nJ);
    const int cog = (int)(q / ((int64_t)taps * nJ));
    const int ci = 2 * jp + (lane >> 5), co = cog * 32 + (lane & 31);
    wfrag[e] = (ci < Cin && co < Cout) ? wpack[((int64_t)ci * taps + tap) * CoutPad + co] : 0.f;
}

template <int KS, bool PRE>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(Conv2dArgs a, const float *__restrict__ wfrag) {
    constexpr int TAPS = KS * KS, P = KS / 2, T = kMfmaTiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int HW = a.H * a.W;
    const int64_t total = (int64_t)a.B * HW;
    const int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * T);
    if (slot0 >= total) return;
    const int cog = blockIdx.y;
    const int nJ = (a.Cin + 1) / 2;

    const float *ptr[T];
    int offs[T][TAPS];
    uint64_t vmask = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int sl = (int)min(slot0 + t * 32 + col, total - 1);   // (the entry point keeps B * H * W below 2^31)
        const int b = sl / HW;
        const int pix = sl - b * HW;
        const int y = pix / a.W, x = pix - y * a.W;
        ptr[t] = a.in + (int64_t)b * a.in_bs;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int yy = y + tap / KS - P, xx = x + tap % KS - P;
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            offs[t][tap] = min(max(yy, 0), a.H - 1) * a.W + min(max(xx, 0), a.W - 1);
            vmask |= (uint64_t)(ok ? 1 : 0) << (t * TAPS + tap);
        }
    }
    const float *wf = wfrag + (int64_t)cog * nJ * TAPS * 64 + lane;

    f32x16_t acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // two operand sets used in turn (no register copies between iterations): while the MFMAs of one channel pair run,
    // the other set receives the next pair's activations, weight fragments and BatchNorm factors
    float xa[2][T][TAPS], wb[2][TAPS], pa[2] = {1.f, 1.f}, pb[2] = {0.f, 0.f};
    auto request = [&](int set, int jp) {
        const int ci = min(2 * jp + half, a.Cin - 1);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
#if DPK_C2D_ABLATE == 1
                if (jp == 0)
#endif
                xa[set][t][tap] = ptr[t][ci * HW + offs[t][tap]];
            }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
#if DPK_C2D_ABLATE == 4
            if (jp == 0)
#endif
            wb[set][tap] = wf[((int64_t)jp * TAPS + tap) * 64];
        }
        if (PRE) {
            pa[set] = a.pre[ci];
            pb[set] = a.pre[a.Cin + ci];
        }
    };
    auto step = [&](int set, int jp) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                float r = xa[set][t][tap];
                if (PRE) r = fmaxf(fmaf(r, pa[set], pb[set]), 0.f);
                xa[set][t][tap] = ((vmask >> (t * TAPS + tap)) & 1) ? r : 0.f;
            }
        // the requests stay above the MFMAs (left alone, the scheduler sinks every load to just before its use one
        // iteration later -- shortest live ranges -- and each MFMA pair then waits a full memory latency: measured 3x);
        // the last iteration re-reads its own operands: in bounds, unused
        __builtin_amdgcn_sched_barrier(0);
        request(set ^ 1, min(jp + 1, nJ - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int t = 0; t < T; ++t) {
#if DPK_C2D_ABLATE == 2
                if (tap > 0) continue;
#endif
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[set][tap], xa[set][t][tap], acc[t], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    request(0, 0);
    for (int jp = 0; jp < nJ; jp += 2) {
        step(0, jp);
        if (jp + 1 < nJ) step(1, jp + 1);
    }
    // epilogue: every residual / bias value is requested before the first addition (clamped addresses, no branches around
    // the loads: one memory latency per tile instead of one per value), then 128-byte store runs under the lane predicate
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int sl = (int)min(slot0 + t * 32 + col, total - 1);
        const bool live = slot0 + t * 32 + col < total;
        const int b = sl / HW;
        const int pix = sl - b * HW;
        float add[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = 0.f;
        if (a.res) {
            const float *rp = a.res + (int64_t)b * a.res_bs + pix;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min(cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
                add[r] = rp[(int64_t)co * HW];
            }
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min(cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
                add[r] += a.bias[co];
            }
        }
        float *op = a.out + (int64_t)b * a.out_bs + pix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (live && co < a.Cout) op[(int64_t)co * HW] = acc[t][r] + add[r];
        }
    }
}

// ---- 3x3 convolution on the matrix cores, activations staged through LDS ------------------------------------------------
// A work-group owns G consecutive samples (G * H * W <= 1024 pixels) and 32 output channels.  The input is staged 4 channels
// at a time into zero-bordered (H+2) x (W+2) planes in LDS -- each activation is fetched, normalised and rectified ONCE (the
// per-tap global loads of conv2d_mfma_kernel<3> fetch it nine times and do not overlap with the MFMAs) and the zero padding
// is the border, not a mask -- two buffers: the next 4 channels travel global -> registers while the MFMAs of the current 4
// run, and are written to the other buffer afterwards.  MFMA operands: lane l reads its pixel of channel 2*jp + (l >> 5) at
// the tap's constant offset (ds_read_b32, one per MFMA and tile, requested one tap ahead); weights as in conv2d_mfma_kernel.
// A wave holds up to 8 tiles of 32 pixels (128 accumulator registers); tiles are dealt round-robin to the 4 waves.
constexpr int kLdsCC = 4;        // channels per staged chunk
constexpr int kLdsPix = 1024;    // pixels per work-group at most (32 tiles)

struct ConvLdsGeom {
    int G, plane, NP;            // samples per work-group, floats per padded plane, G * H * W
};

// NW waves per work-group, TW tiles per wave (tile = wave + NW * t): (4, 8) covers 1024 pixels (32x32; five 14x14 samples = 31 tiles), (4, 7) 28 tiles (28x28: 25), (4, 6) 24 tiles
template <bool PRE, int NW, int TW>
__global__ __launch_bounds__(NW * 64, NW > 4 ? 3 : 2) void conv3x3_lds_kernel(Conv2dArgs a, const float *__restrict__ wfrag, ConvLdsGeom q) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int HW = a.H * a.W, PW = a.W + 2;
    const int b0 = blockIdx.x * q.G;
    const int nb = min(q.G, a.B - b0);
    const int npv = nb * HW;                      // live pixels of this work-group
    const int cog = blockIdx.y;
    const int nJ = (a.Cin + 1) / 2;
    const int nchunk = (a.Cin + kLdsCC - 1) / kLdsCC;
    const int chunk_floats = kLdsCC * q.G * q.plane;
    constexpr int NT = NW * 64, NS = (kLdsPix + NT - 1) / NT;   // threads, staged pixel slots per thread

    // zero both buffers once: the borders stay zero for the whole kernel
    for (int i = tid; i < 2 * chunk_floats; i += NT) lds[i] = 0.f;

    // staging roles: pixel slots tid + NT * k
    int goff[NS], loff[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int p = tid + NT * k;
        const int pc = min(p, npv - 1);
        const int g = pc / HW, pix = pc - g * HW;
        const int y = pix / a.W, x = pix - y * a.W;
        goff[k] = p < npv ? (int)(g * a.in_bs) + pix : -1;      // (the entry point keeps G * in_bstride below 2^31)
        loff[k] = g * q.plane + (y + 1) * PW + (x + 1);
    }
    const float *inb = a.in + (int64_t)b0 * a.in_bs;
    float stg[kLdsCC][NS];
    auto fetch = [&](int c) {
#pragma unroll
        for (int ch = 0; ch < kLdsCC; ++ch) {
            const int ci = min(c * kLdsCC + ch, a.Cin - 1);
#pragma unroll
            for (int k = 0; k < NS; ++k) stg[ch][k] = inb[(int64_t)ci * HW + max(goff[k], 0)];
        }
    };
    auto stash = [&](int c, float *buf) {
#pragma unroll
        for (int ch = 0; ch < kLdsCC; ++ch) {
            const int ci = c * kLdsCC + ch;
            const bool real = ci < a.Cin;
            float pa = 1.f, pb = 0.f;
            if (PRE && real) {
                pa = a.pre[ci];
                pb = a.pre[a.Cin + ci];
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float r = stg[ch][k];
                if (PRE) r = fmaxf(fmaf(r, pa, pb), 0.f);
                if (goff[k] >= 0) buf[ch * q.G * q.plane + loff[k]] = real ? r : 0.f;
            }
        }
    };

    // MFMA roles
    int lbase[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int p = min((wave + NW * t) * 32 + col, npv - 1);
        const int g = p / HW, pix = p - g * HW;
        const int y = pix / a.W, x = pix - y * a.W;
        lbase[t] = half * q.G * q.plane + g * q.plane + (y + 1) * PW + (x + 1);
    }
    const float *wf = wfrag + (int64_t)cog * nJ * 9 * 64 + lane;

    f32x16_t acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float wc[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wc[tap] = wf[tap * 64];
    fetch(0);
    __syncthreads();                              // zero fill done
    stash(0, lds);
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const float *cur = lds + (c & 1) * chunk_floats;
        if (c + 1 < nchunk) fetch(c + 1);
        const int njp = min(kLdsCC / 2, nJ - c * (kLdsCC / 2));
        // k-steps of this chunk: (jp, tap).  The LDS operands of the next tap are requested before the MFMAs of this one;
        // the nine weight fragments of the next channel pair (an L2 round trip: longer than one tap's MFMAs) before the
        // nine taps of this pair
        float xa[2][TW];
        auto request = [&](int set, int jp, int tap) {
            const int so = 2 * jp * q.G * q.plane + (tap / 3 - 1) * PW + (tap % 3 - 1);
#pragma unroll
            for (int t = 0; t < TW; ++t) xa[set][t] = cur[lbase[t] + so];
        };
        request(0, 0, 0);
        // (nine taps per pair: the operand set of a tap is (tap + PAR) & 1 with PAR alternating between pairs)
        auto pair = [&](auto par, int jp) {
            constexpr int PAR = decltype(par)::value;
            const int jg = c * (kLdsCC / 2) + jp;
            const int jn = min(jg + 1, nJ - 1);
            float wn[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wn[tap] = wf[((int64_t)jn * 9 + tap) * 64];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_sched_barrier(0);
                if (tap < 8) request((tap + 1 + PAR) & 1, jp, tap + 1);
                else if (jp + 1 < njp) request((tap + 1 + PAR) & 1, jp + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TW; ++t)   // (every wave runs all 7 tiles: straight-line code, exact wait counts;
                    // tiles past the work-group's pixels compute on a clamped pixel and are not stored)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[tap], xa[(tap + PAR) & 1][t], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wc[tap] = wn[tap];
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int jp = 0; jp < njp; jp += 2) {
            pair(std::integral_constant<int, 0>{}, jp);
            if (jp + 1 < njp) pair(std::integral_constant<int, 1>{}, jp + 1);
        }
        if (c + 1 < nchunk) stash(c + 1, lds + ((c + 1) & 1) * chunk_floats);
        __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int p = (wave + NW * t) * 32 + col;
        const bool live = p < npv;
        const int pc = min(p, npv - 1);
        const int g = pc / HW, pix = pc - g * HW;
        const int b = b0 + g;
        float add[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = 0.f;
        if (a.res) {
            const float *rp = a.res + (int64_t)b * a.res_bs + pix;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min(cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
                add[r] = rp[(int64_t)co * HW];
            }
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min(cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.Cout - 1);
                add[r] += a.bias[co];
            }
        }
        float *op = a.out + (int64_t)b * a.out_bs + pix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (live && co < a.Cout) op[(int64_t)co * HW] = acc[t][r] + add[r];
        }
    }
}

// ---- coupling transformation (one work-group per sample) ---------------------------------------------------------
__device__ inline float block_sum_256(float v, float *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// density direction: u = (x - t) exp(-s), ldj -= sum s;  inverse: x = u exp(s) + t, ldj += sum s.
// checkerboard (inv_mask != null): z [B, 2C or C, H, W], t and s multiplied by inv_mask[H*W];
// channel-wise: x = [my | mx] (reverse: [mx | my]) halves of Ch = C/2 channels, z [B, 2Ch or Ch, H, W] conditions on mx;
// mx is copied through.
__global__ __launch_bounds__(256) void coupling2d_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ inv_mask, int C, int HW,
                                                         int affine, int reverse, int inverse,
                                                         const float *__restrict__ ldj_in, float *__restrict__ out,
                                                         float *__restrict__ ldj_out) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const bool chw = inv_mask == nullptr;
    const int Ch = chw ? C / 2 : C;
    const int n = Ch * HW;
    const int64_t zb = (int64_t)b * (affine ? 2 : 1) * n;
    // offset of the transformed half inside x / out
    const int off = chw ? (reverse ? n : 0) : 0;
    const float *xb = x + (int64_t)b * C * HW;
    float *ob = out + (int64_t)b * C * HW;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / HW, p = i - c * HW;
        const float m = chw ? 1.f : inv_mask[p];
        float t = z[zb + i] * m, s = 0.f;
        if (affine) s = scale[c] * tanhf(z[zb + n + i]) * m;
        const float xv = xb[off + i];
        ob[off + i] = inverse ? fmaf(xv, expf(s), t) : (xv - t) * expf(-s);
        acc += s;
        if (chw) ob[(n - off) + i] = xb[(n - off) + i];
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) ldj_out[b] = (ldj_in ? ldj_in[b] : 0.f) + (inverse ? acc : -acc);
}

// BatchNormLayer2d with running statistics (flows/utils.py:186-222).
// density: u = (x - mean) / sqrt(var + eps) * exp(w) + bias, ldj += HW * sum_c (w - 0.5 log(var + eps));  inverse: the inverse.
__global__ __launch_bounds__(256) void bn2d_bijector_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps, int C, int HW,
                                                            int inverse, const float *__restrict__ ldj_in,
                                                            float *__restrict__ out, float *__restrict__ ldj_out) {
    const int b = blockIdx.x;
    const int n = C * HW;
    const float *xb = x + (int64_t)b * n;
    float *ob = out + (int64_t)b * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / HW;
        const float sd = sqrtf(var[c] + eps);
        ob[i] = inverse ? (xb[i] - bias[c]) * expf(-w[c]) * sd + mean[c] : (xb[i] - mean[c]) / sd * expf(w[c]) + bias[c];
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += w[c] - 0.5f * logf(var[c] + eps);
        s *= (float)HW;
        ldj_out[b] = (ldj_in ? ldj_in[b] : 0.f) + (inverse ? -s : s);
    }
}

// ---- squeeze / multi-scale permutations -----------------------------------------------------------------------------
// out channel o takes in[b, table[o] >> 2, 2h + ((table[o] >> 1) & 1), 2w + (table[o] & 1)];
// channels [0, Ca) go to out_a, [Ca, 4C) to out_b.
__global__ __launch_bounds__(256) void space_to_depth_kernel(float *__restrict__ in, int64_t total, int C, int H,
                                                             int W, const int *__restrict__ table,
                                                             float *__restrict__ out_a, int Ca,
                                                             float *__restrict__ out_b, int inverse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int h2 = H / 2, w2 = W / 2, hw2 = h2 * w2;
    const int w = (int)(i % w2);
    const int h = (int)((i / w2) % h2);
    const int o = (int)((i / hw2) % (4 * C));
    const int64_t b = i / ((int64_t)hw2 * 4 * C);
    const int e = table[o];
    const int64_t full = ((b * C + (e >> 2)) * H + 2 * h + ((e >> 1) & 1)) * W + 2 * w + (e & 1);
    float *dst = o < Ca ? out_a + ((b * Ca + o) * hw2 + h * w2 + w) : out_b + ((b * (4 * C - Ca) + (o - Ca)) * hw2 + h * w2 + w);
    if (inverse) in[full] = *dst;
    else *dst = in[full];
}

}  // namespace dpk

using namespace dpk;

extern "C" {

// packed table = the [ci][tap][CoutPad] table of the vector-ALU kernel, then the fragment-ordered table of the MFMA kernel
static int64_t conv_valu_floats(int32_t Cout, int32_t Cin, int32_t ks) {
    return (int64_t)Cin * ks * ks * align_up(Cout, kConvCO);
}
static int64_t conv_frag_floats(int32_t Cout, int32_t Cin, int32_t ks) {
    return (int64_t)cdiv(Cout, 32) * ((Cin + 1) / 2) * ks * ks * 64;
}
int64_t dpk_conv2d_pack_floats(int32_t Cout, int32_t Cin, int32_t ks) {
    return conv_valu_floats(Cout, Cin, ks) + conv_frag_floats(Cout, Cin, ks);
}

int dpk_conv2d_prepare(const float *weight_v, const float *weight_g, int32_t Cout, int32_t Cin, int32_t ks,
                       const float *bn_weight, const float *bn_bias, const float *bn_mean, const float *bn_var,
                       float bn_eps, float *wpack, float *pre, void *stream) {
    DPK_REQUIRE(weight_v && wpack, DPK_EINVAL, "conv2d_prepare: null pointer");
    DPK_REQUIRE(Cout > 0 && Cin > 0 && (ks == 1 || ks == 3), DPK_EINVAL, "conv2d_prepare: Cout=%d Cin=%d ks=%d", Cout,
                Cin, ks);
    const bool bn = bn_weight != nullptr;
    DPK_REQUIRE(!bn || (bn_bias && bn_mean && bn_var && pre), DPK_EINVAL, "conv2d_prepare: incomplete BatchNorm2d");
    const int CoutPad = (int)align_up(Cout, kConvCO);
    DPK_LAUNCH(conv2d_prepare_kernel, dim3(CoutPad + (bn ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, weight_v, weight_g,
               Cout, CoutPad, Cin, ks * ks, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, wpack, pre);
    DPK_CHECK_LAUNCH("conv2d_prepare_kernel");
    const int64_t nf = conv_frag_floats(Cout, Cin, ks);
    DPK_LAUNCH(conv2d_frag_kernel, dim3((unsigned)cdiv(nf, 256)), dim3(256), 0, (hipStream_t)stream, wpack, Cout, CoutPad, Cin,
               ks * ks, wpack + conv_valu_floats(Cout, Cin, ks), nf);
    DPK_CHECK_LAUNCH("conv2d_frag_kernel");
    return DPK_OK;
}

int dpk_conv2d_forward(const float *in, int64_t in_bstride, int64_t B, int32_t Cin, int32_t H, int32_t W,
                       const float *wpack, int32_t Cout, int32_t ks, const float *pre, const float *in_mask,
                       const float *bias, const float *res, int64_t res_bstride, float *out, int64_t out_bstride,
                       void *stream) {
    DPK_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DPK_EINVAL, "conv2d_forward: bad sizes");
    DPK_REQUIRE(ks == 1 || ks == 3, DPK_EUNSUPPORTED, "conv2d_forward: kernel size %d not built (1, 3)", ks);
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && wpack && out, DPK_EINVAL, "conv2d_forward: null pointer");
    DPK_REQUIRE(in_bstride >= (int64_t)Cin * H * W && out_bstride >= (int64_t)Cout * H * W &&
                    (!res || res_bstride >= (int64_t)Cout * H * W),
                DPK_EINVAL, "conv2d_forward: batch stride below the tensor size");
    DPK_REQUIRE((int64_t)Cin * H * W < INT32_MAX && (int64_t)Cout * H * W < INT32_MAX, DPK_EUNSUPPORTED,
                "conv2d_forward: image too large");
    Conv2dArgs a{};
    a.in = in; a.in_bs = in_bstride; a.B = (int)B; a.Cin = Cin; a.H = H; a.W = W;
    a.w = wpack; a.Cout = Cout; a.CoutPad = (int)align_up(Cout, kConvCO);
    a.pre = pre; a.mask = in_mask; a.bias = bias; a.res = res; a.res_bs = res_bstride;
    a.out = out; a.out_bs = out_bstride;
    DPK_REQUIRE(B <= INT32_MAX / 2, DPK_EUNSUPPORTED, "conv2d_forward: batch too large");
    const int64_t threads = B * H * ((W + kConvPix - 1) / kConvPix);
    const dim3 grid((unsigned)cdiv(threads, 256), (unsigned)(a.CoutPad / kConvCO));
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(256);
    // matrix-core kernel: the unmasked 1x1 convolutions with enough input channels to fill the k-steps (skip / transition /
    // layers of >= 16 output channels: one pass over the activations for all 32 output channels, 278 against 351 us at 4096 x 32 x 28 x 28).
    // For 3x3 it is measured SLOWER than the vector-ALU kernel (1156 against 967 us): its nine per-tap operand
    // loads per MFMA pair do not overlap with the MFMAs; DPK_CONV_MFMA3=1 selects it anyway, DPK_CONV_VALU=1 never uses it
    static const bool force_valu = getenv("DPK_CONV_VALU") != nullptr && getenv("DPK_CONV_VALU")[0] == '1';
    static const bool mfma3 = getenv("DPK_CONV_MFMA3") != nullptr && getenv("DPK_CONV_MFMA3")[0] == '1';
    // 3x3 with the activations staged through LDS (images of at most 1024 pixels; DPK_CONV_LDS=0 switches it off)
    static const bool no_lds = getenv("DPK_CONV_LDS") != nullptr && getenv("DPK_CONV_LDS")[0] == '0';
    if (!in_mask && ks == 3 && Cin >= 8 && Cout >= 16 && !force_valu && !mfma3 && !no_lds && H * W <= kLdsPix &&
        B < INT32_MAX / 2) {
        ConvLdsGeom q;
        q.plane = (H + 2) * (W + 2);
        q.G = (int)std::max<int64_t>(1, std::min<int64_t>(kLdsPix / (H * W), 2048 / q.plane));
        q.G = (int)std::min<int64_t>(q.G, B);
        q.NP = q.G * H * W;
        if ((int64_t)q.G * in_bstride < INT32_MAX) {
            const size_t lds_bytes = (size_t)2 * kLdsCC * q.G * q.plane * sizeof(float);
            const float *wfrag = wpack + conv_valu_floats(Cout, Cin, ks);
            const dim3 lgrid((unsigned)cdiv(B, q.G), (unsigned)cdiv(Cout, 32));
            const int ntile = cdiv((int64_t)q.NP, 32);
#define DPK_LDS_CASE(PRE_, NW_, TW_) \
    DPK_LAUNCH((conv3x3_lds_kernel<PRE_, NW_, TW_>), lgrid, dim3(NW_ * 64), lds_bytes, st, a, wfrag, q)
            if (ntile <= 24) {
                if (pre) DPK_LDS_CASE(true, 4, 6); else DPK_LDS_CASE(false, 4, 6);
            } else if (ntile <= 28) {   // (25 tiles on 5 waves x 5 tiles -- no padded tile -- measured 910 against 645 us:
                // uneven SIMD load)
                if (pre) DPK_LDS_CASE(true, 4, 7); else DPK_LDS_CASE(false, 4, 7);
            } else {
                if (pre) DPK_LDS_CASE(true, 4, 8); else DPK_LDS_CASE(false, 4, 8);
            }
#undef DPK_LDS_CASE
            DPK_CHECK_LAUNCH("conv3x3_lds_kernel");
            return DPK_OK;
        }
    }
    if (!in_mask && Cin >= 8 && Cout >= 16 && !force_valu && (ks == 1 || mfma3) && B * H * W < INT32_MAX) {
        const float *wfrag = wpack + conv_valu_floats(Cout, Cin, ks);
        const int64_t slots = B * H * W;
        const dim3 mgrid((unsigned)cdiv(slots, 4 * 32 * kMfmaTiles), (unsigned)cdiv(Cout, 32));
        if (ks == 3) {
            if (pre) DPK_LAUNCH((conv2d_mfma_kernel<3, true>), mgrid, blk, 0, st, a, wfrag);
            else DPK_LAUNCH((conv2d_mfma_kernel<3, false>), mgrid, blk, 0, st, a, wfrag);
        } else {
            if (pre) DPK_LAUNCH((conv2d_mfma_kernel<1, true>), mgrid, blk, 0, st, a, wfrag);
            else DPK_LAUNCH((conv2d_mfma_kernel<1, false>), mgrid, blk, 0, st, a, wfrag);
        }
        DPK_CHECK_LAUNCH("conv2d_mfma_kernel");
        return DPK_OK;
    }
#define DPK_CONV_CASE(KS_, PRE_, MASK_) DPK_LAUNCH((conv2d_kernel<KS_, PRE_, MASK_>), grid, blk, 0, st, a)
    if (ks == 3) {
        if (pre && in_mask) DPK_CONV_CASE(3, true, true);
        else if (pre) DPK_CONV_CASE(3, true, false);
        else if (in_mask) DPK_CONV_CASE(3, false, true);
        else DPK_CONV_CASE(3, false, false);
    } else {
        if (pre && in_mask) DPK_CONV_CASE(1, true, true);
        else if (pre) DPK_CONV_CASE(1, true, false);
        else if (in_mask) DPK_CONV_CASE(1, false, true);
        else DPK_CONV_CASE(1, false, false);
    }
#undef DPK_CONV_CASE
    DPK_CHECK_LAUNCH("conv2d_kernel");
    return DPK_OK;
}

int dpk_coupling2d_transform(const float *x, const float *z, const float *scale, const float *inv_mask, int64_t B,
                             int32_t C, int32_t H, int32_t W, int32_t affine, int32_t reverse, int32_t inverse,
                             const float *ldj_in, float *out, float *ldj_out, void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "coupling2d_transform: bad sizes");
    DPK_REQUIRE(inv_mask || C % 2 == 0, DPK_EINVAL, "coupling2d_transform: channel-wise coupling needs an even C=%d", C);
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX && B < INT32_MAX, DPK_EUNSUPPORTED, "coupling2d_transform: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && z && out && ldj_out && (!affine || scale), DPK_EINVAL, "coupling2d_transform: null pointer");
    DPK_LAUNCH(coupling2d_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, z, scale, inv_mask, C, H * W,
               affine, reverse, inverse, ldj_in, out, ldj_out);
    DPK_CHECK_LAUNCH("coupling2d_kernel");
    return DPK_OK;
}

int dpk_bn2d_bijector(const float *x, const float *weight, const float *bias, const float *mean, const float *var,
                      float eps, int64_t B, int32_t C, int32_t H, int32_t W, int32_t inverse, const float *ldj_in,
                      float *out, float *ldj_out, void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "bn2d_bijector: bad sizes");
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX && B < INT32_MAX, DPK_EUNSUPPORTED, "bn2d_bijector: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && weight && bias && mean && var && out && ldj_out, DPK_EINVAL, "bn2d_bijector: null pointer");
    DPK_LAUNCH(bn2d_bijector_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, weight, bias, mean, var,
               eps, C, H * W, inverse, ldj_in, out, ldj_out);
    DPK_CHECK_LAUNCH("bn2d_bijector_kernel");
    return DPK_OK;
}

static int depth_perm(const float *full, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *table,
                      float *part_a, int32_t Ca, float *part_b, int inverse, void *stream, const char *who) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, DPK_EINVAL,
                "%s: needs even H, W (C=%d H=%d W=%d)", who, C, H, W);
    DPK_REQUIRE(Ca > 0 && Ca <= 4 * C, DPK_EINVAL, "%s: split %d outside (0, %d]", who, Ca, 4 * C);
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(full && table && part_a && (Ca == 4 * C || part_b), DPK_EINVAL, "%s: null pointer", who);
    const int64_t total = B * C * H * W;
    DPK_REQUIRE(total / 256 < INT32_MAX, DPK_EUNSUPPORTED, "%s: too large", who);
    DPK_LAUNCH(space_to_depth_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
               const_cast<float *>(full), total, C,
               H, W, table, part_a, Ca, part_b, inverse);
    DPK_CHECK_LAUNCH("space_to_depth_kernel");
    return DPK_OK;
}

int dpk_space_to_depth(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *table, float *out_a,
                       int32_t Ca, float *out_b, void *stream) {
    return depth_perm(in, B, C, H, W, table, out_a, Ca, out_b, 0, stream, "space_to_depth");
}

int dpk_depth_to_space(const float *in_a, int32_t Ca, const float *in_b, int64_t B, int32_t C, int32_t H, int32_t W,
                       const int32_t *table, float *out, void *stream) {
    return depth_perm(out, B, C, H, W, table, const_cast<float *>(in_a), Ca, const_cast<float *>(in_b), 1, stream,
                      "depth_to_space");
}

}  // extern "C"
