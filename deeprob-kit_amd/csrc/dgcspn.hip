// DGC-SPN spatial layers for gfx950 (reference: deeprob/spn/layers/dgcspn.py).
//
// All tensors are NCHW fp32 like the reference.  Lanes run over pixels (w fastest), so activations
// and the position-dependent sum weights [Cout, Cin, H, W] are read coalesced; the batch is the outer
// (grid-stride) dimension.  These are streaming kernels: the roofline is HBM bandwidth on the
// activation tensors (588 KB/sample for config 4).
//   spatial_gaussian : SpatialGaussianLayer.forward  dgcspn.py:101-120
//   spatial_product  : SpatialProductLayer.forward   dgcspn.py:224-236 (F.pad + F.conv2d with ones /
//                      one-hot kernels == a sum of kh*kw dilated taps)
//   spatial_sum      : SpatialSumLayer.forward       dgcspn.py:289-304
// SpatialRootLayer.forward (:343-355) is dpk_root_forward on the flattened map.
#include "common.h"
#include "dgcspn_stream.h"
#include <math.h>

namespace dpk {

static inline int grid_cap(int64_t total, int block, int cap = 16384) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------
// Gaussian leaves: y[b,k,p] = sum_c nan_to_num(log N(x[b,c,p]; loc[k,c,p], scale[k,c,p]))
// ------------------------------------------------------------------------------------------------
// Evaluation form (no dropout) for up to kGaussC input channels: thread = (pixel, leaf channel k) over a slice of
// samples; the pixel's parameters become (mu, 1 / (2 sigma^2), -log sigma - log sqrt(2 pi)) once, then every
// sample costs one load per input channel, three flops and one store.
constexpr int kGaussC = 4;
__global__ __launch_bounds__(256) void spatial_gaussian_fwd_fast_kernel(const float *__restrict__ x,
                                                                        const float *__restrict__ loc,
                                                                        const float *__restrict__ scale, int64_t B,
                                                                        int K, int C, int HW, int bslice,
                                                                        float *__restrict__ out) {
    const int p = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (p >= HW || k >= K) return;
    float mu[kGaussC], iv[kGaussC], cs[kGaussC];
#pragma unroll
    for (int c = 0; c < kGaussC; ++c) {
        const bool live = c < C;
        const float sg = live ? scale[((int64_t)k * C + c) * HW + p] : 1.f;
        mu[c] = live ? loc[((int64_t)k * C + c) * HW + p] : 0.f;
        iv[c] = 0.5f / (sg * sg);
        cs[c] = -logf(sg) - kLogSqrt2Pi;
    }
    const int64_t b0 = (int64_t)blockIdx.z * bslice, b1 = min(b0 + bslice, B);
#pragma unroll 4
    for (int64_t b = b0; b < b1; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kGaussC; ++c) {
            if (c < C) {
                const float d = x[(b * C + c) * HW + p] - mu[c];
                acc += nan_to_num_f(fmaf(-(d * d), iv[c], cs[c]));
            }
        }
        out[(b * K + k) * HW + p] = acc;
    }
}

// The same with four consecutive pixels per thread (H*W a multiple of 4, 16-byte aligned tensors): 16-byte loads and
// stores, 1 KB per wave and instruction in flight instead of 256 B -- the scalar form is latency-bound at ~2 TB/s.
typedef float gauss_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void spatial_gaussian_fwd_fast4_kernel(const float *__restrict__ x,
                                                                         const float *__restrict__ loc,
                                                                         const float *__restrict__ scale, int64_t B,
                                                                         int K, int C, int HW, int bslice,
                                                                         float *__restrict__ out) {
    // (leaf channel, pixel quad) flattened over the grid: no partly filled waves at the end of every channel plane
    const int e = blockIdx.x * 256 + threadIdx.x, Q = HW >> 2;
    if (e >= K * Q) return;
    const int k = e / Q, p = (e - k * Q) * 4;
    gauss_f4 mu[kGaussC], iv[kGaussC], cs[kGaussC];
#pragma unroll
    for (int c = 0; c < kGaussC; ++c) {
        const bool live = c < C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = live ? scale[((int64_t)k * C + c) * HW + p + j] : 1.f;
            mu[c][j] = live ? loc[((int64_t)k * C + c) * HW + p + j] : 0.f;
            iv[c][j] = 0.5f / (sg * sg);
            cs[c][j] = -logf(sg) - kLogSqrt2Pi;
        }
    }
    const int64_t b0 = (int64_t)blockIdx.z * bslice, b1 = min(b0 + bslice, B);
#pragma unroll 8
    for (int64_t b = b0; b < b1; ++b) {
        gauss_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < kGaussC; ++c) {
            if (c < C) {
                const gauss_f4 xv = *reinterpret_cast<const gauss_f4 *>(x + (b * C + c) * HW + p);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xv[j] - mu[c][j];
                    acc[j] += nan_to_num_f(fmaf(-(d * d), iv[c][j], cs[c][j]));
                }
            }
        }
        *reinterpret_cast<gauss_f4 *>(out + (b * K + k) * HW + p) = acc;
    }
}

__global__ void spatial_gaussian_fwd_kernel(const float *__restrict__ x, const float *__restrict__ loc,
                                            const float *__restrict__ scale, int64_t B, int K, int C, int HW,
                                            float *__restrict__ out, float drop_p, uint64_t seed) {
    const int64_t total = B * K * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const int64_t bk = e / HW;
        const int k = (int)(bk % K);
        const int64_t b = bk / K;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            // training-mode input dropout (dgcspn.py:113-114) on the [B,K,C,H,W] element
            if (drop_p > 0.f && dropout_hit(seed, ((uint64_t)bk * C + c) * HW + p, drop_p)) continue;
            const float xv = x[(b * C + c) * HW + p];
            const float mu = loc[((int64_t)k * C + c) * HW + p], sg = scale[((int64_t)k * C + c) * HW + p];
            const float d = xv - mu;
            acc += nan_to_num_f(-(d * d) / (2.f * sg * sg) - logf(sg) - kLogSqrt2Pi);
        }
        out[e] = acc;
    }
}

// d/dx: gx[b,c,p] = sum_k g[b,k,p] * (-(x-mu)/s^2) (0 where x is NaN)
__global__ void spatial_gaussian_bwd_x_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                              const float *__restrict__ loc, const float *__restrict__ scale,
                                              int64_t B, int K, int C, int HW, float *__restrict__ gx, float drop_p,
                                              uint64_t seed) {
    const int64_t total = B * C * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const int64_t bc = e / HW;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        const float xv = x[e];
        float acc = 0.f;
        if (xv == xv) {
            for (int k = 0; k < K; ++k) {
                if (drop_p > 0.f && dropout_hit(seed, (((uint64_t)b * K + k) * C + c) * HW + p, drop_p)) continue;
                const float sg = scale[((int64_t)k * C + c) * HW + p];
                acc = fmaf(g[(b * K + k) * HW + p], -(xv - loc[((int64_t)k * C + c) * HW + p]) / (sg * sg), acc);
            }
        }
        gx[e] = acc;
    }
}

// parameter gradients: one thread per (k,c,p), a slice of the batch per blockIdx.y, one atomic per slice
__global__ void spatial_gaussian_bwd_p_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                              const float *__restrict__ loc, const float *__restrict__ scale,
                                              int64_t B, int K, int C, int HW, int bslice,
                                              float *__restrict__ gloc, float *__restrict__ gscale, float drop_p,
                                              uint64_t seed) {
    const int64_t n = (int64_t)K * C * HW;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int p = (int)(e % HW);
    const int64_t kc = e / HW;
    const int c = (int)(kc % C), k = (int)(kc / C);
    const float mu = loc[e], sg = scale[e];
    const float iv = 1.f / (sg * sg), is = 1.f / sg;
    const int64_t b0 = (int64_t)blockIdx.y * bslice, b1 = min(b0 + bslice, B);
    float a0 = 0.f, a1 = 0.f;
    for (int64_t b = b0; b < b1; ++b) {
        const float xv = x[(b * C + c) * HW + p];
        if (xv == xv && !(drop_p > 0.f && dropout_hit(seed, (((uint64_t)b * K + k) * C + c) * HW + p, drop_p))) {
            const float gv = g[(b * K + k) * HW + p], d = xv - mu;
            a0 = fmaf(gv, d * iv, a0);
            a1 = fmaf(gv, d * d * iv * is - is, a1);
        }
    }
    if (gloc) atomicAdd(gloc + e, a0);
    if (gscale) atomicAdd(gscale + e, a1);
}

// (ProdGeom, the geometry of a product layer: dgcspn_stream.h)

__device__ __forceinline__ int ipow(int base, int e) {
    int r = 1;
    for (int i = 0; i < e; ++i) r *= base;
    return r;
}

__global__ void spatial_product_fwd_kernel(const float *__restrict__ in, int64_t B, ProdGeom q,
                                           float *__restrict__ out) {
    const int OHW = q.OH * q.OW, T = q.kh * q.kw;
    const int64_t total = B * q.OC * OHW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int op = (int)(e % OHW);
        const int64_t bo = e / OHW;
        const int oc = (int)(bo % q.OC);
        const int64_t b = bo / q.OC;
        const int oh = op / q.OW, ow = op - oh * q.OW;
        float acc = 0.f;
        int div = q.depthwise ? 1 : ipow(q.C, T - 1);
        for (int t = 0; t < T; ++t) {
            const int th = t / q.kw, tw = t - th * q.kw;
            const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
            // non-depthwise: output channel = combination (c_0..c_{T-1}) in itertools.product order
            const int c = q.depthwise ? oc : (oc / div) % q.C;
            if (!q.depthwise) div /= q.C;
            if (ih >= 0 && ih < q.H && iw >= 0 && iw < q.W) acc += in[((b * q.C + c) * q.H + ih) * q.W + iw];
        }
        out[e] = acc;
    }
}

// gin[b,c,ih,iw] = sum of g over every (output channel, tap, output pixel) that read it
__global__ void spatial_product_bwd_kernel(const float *__restrict__ g, int64_t B, ProdGeom q,
                                           float *__restrict__ gin) {
    const int HW = q.H * q.W, T = q.kh * q.kw;
    const int64_t total = B * q.C * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int ip = (int)(e % HW);
        const int64_t bc = e / HW;
        const int c = (int)(bc % q.C);
        const int64_t b = bc / q.C;
        const int ih = ip / q.W, iw = ip - ih * q.W;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const int th = t / q.kw, tw = t - th * q.kw;
            const int nh = ih + q.pt - th * q.dh, nw = iw + q.pl - tw * q.dw;
            if (nh < 0 || nw < 0 || nh % q.sh || nw % q.sw) continue;
            const int oh = nh / q.sh, ow = nw / q.sw;
            if (oh >= q.OH || ow >= q.OW) continue;
            if (q.depthwise) {
                acc += g[((b * q.OC + c) * q.OH + oh) * q.OW + ow];
            } else {
                // all combinations whose digit t equals c: hi * C^(T-t) + c * C^(T-1-t) + lo
                const int lo_n = ipow(q.C, T - 1 - t), hi_n = ipow(q.C, t);
                for (int hi = 0; hi < hi_n; ++hi)
                    for (int lo = 0; lo < lo_n; ++lo) {
                        const int oc = (hi * q.C + c) * lo_n + lo;
                        acc += g[((b * q.OC + oc) * q.OH + oh) * q.OW + ow];
                    }
            }
        }
        gin[e] = acc;
    }
}

// Depthwise, <= 4 taps (every product layer of the reference's DGC-SPN): thread = one input pixel, its (up to four)
// source positions in the output-gradient plane worked out once, then a run of (sample, channel) planes with no
// integer division and unconditional loads at clamped offsets.
__global__ __launch_bounds__(256) void spatial_product_bwd_dw_kernel(const float *__restrict__ g, int64_t planes,
                                                                     ProdGeom q, int pslice,
                                                                     float *__restrict__ gin) {
    const int HW = q.H * q.W, OHW = q.OH * q.OW, T = q.kh * q.kw;
    const int ip = blockIdx.x * 256 + threadIdx.x;
    if (ip >= HW) return;
    const int ih = ip / q.W, iw = ip - ih * q.W;
    int off[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int th = t / q.kw, tw = t - th * q.kw;
        const int nh = ih + q.pt - th * q.dh, nw = iw + q.pl - tw * q.dw;
        const int oh = nh / q.sh, ow = nw / q.sw;
        ok[t] = t < T && nh >= 0 && nw >= 0 && nh % q.sh == 0 && nw % q.sw == 0 && oh < q.OH && ow < q.OW;
        off[t] = ok[t] ? oh * q.OW + ow : 0;
    }
    const int64_t p0 = (int64_t)blockIdx.y * pslice, p1 = min(p0 + pslice, planes);
#pragma unroll 4
    for (int64_t pl = p0; pl < p1; ++pl) {
        const float *gp = g + pl * OHW;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float v = gp[off[t]];
            acc += ok[t] ? v : 0.f;
        }
        gin[pl * HW + ip] = acc;
    }
}
static void launch_product_bwd(const float *g, int64_t B, const ProdGeom &q, float *gin, hipStream_t st) {
    if (q.depthwise && q.kh * q.kw <= 4) {
        const int64_t planes = B * q.C, cols = cdiv(q.H * q.W, 256);
        int64_t slices = cdiv(4096, cols);                      // about 16 k waves
        int64_t pslice = cdiv(planes, slices);
        if (pslice < 4) pslice = 4;
        slices = cdiv(planes, pslice);
        DPK_LAUNCH(spatial_product_bwd_dw_kernel, dim3((unsigned)cols, (unsigned)slices), dim3(256), 0, st, g, planes, q,
                   (int)pslice, gin);
    } else {
        DPK_LAUNCH(spatial_product_bwd_kernel, dim3(grid_cap(B * q.C * q.H * q.W, 256)), dim3(256), 0, st, g, B, q, gin);
    }
}

// ------------------------------------------------------------------------------------------------
// Sum layer: out[b,o,p] = logsumexp_c(x[b,c,p] + log_softmax(weight, 1)[o,c,p])
// ------------------------------------------------------------------------------------------------
// softmax over the input channels for every (o, p): W, LW [Cout, Cin, HW]
// (bid / nblocks: the block's place among the blocks that share this table -- the whole grid for the per-level launch,
// a slice of it for dpk_spatial_tables)
__device__ __forceinline__ void spatial_softmax_body(const float *__restrict__ w, int Cout, int Cin, int HW,
                                                     float *__restrict__ Wl, float *__restrict__ LW, int bid, int nblocks) {
    const int64_t n = (int64_t)Cout * HW;
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < n; e += (int64_t)nblocks * blockDim.x) {
        const int p = (int)(e % HW), o = (int)(e / HW);
        const float *src = w + (int64_t)o * Cin * HW + p;
        float m = -INFINITY;
        for (int c = 0; c < Cin; ++c) m = fmaxf(m, src[(int64_t)c * HW]);
        float s = 0.f;
        for (int c = 0; c < Cin; ++c) s += expf(src[(int64_t)c * HW] - m);
        // W by (correctly rounded) division: rows sum to 1 without the common-mode error that
        // exp(w - m - log s) inherits from the rounding of log s
        const float ls = logf(s);
        for (int c = 0; c < Cin; ++c) {
            const float d = src[(int64_t)c * HW] - m;
            LW[((int64_t)o * Cin + c) * HW + p] = d - ls;
            Wl[((int64_t)o * Cin + c) * HW + p] = expf(d) / s;
        }
    }
}
__global__ void spatial_softmax_kernel(const float *__restrict__ w, int Cout, int Cin, int HW,
                                       float *__restrict__ Wl, float *__restrict__ LW, const unsigned *gate = nullptr) {
    if (gate_closed(gate)) return;   // (tables still match the live weights: common.h params_gate)
    spatial_softmax_body(w, Cout, Cin, HW, Wl, LW, (int)blockIdx.x, (int)gridDim.x);
}

// thread per (b, p); exp-domain with the row maximum, exact two-pass fallback when the scaled sum
// vanishes (same policy as the RAT-SPN sum layer).  CMAX > 0: the Cin <= CMAX exponentials live in
// registers and are computed once; CMAX == 0: any Cin, exponentials recomputed per output channel.
// expf / logf are the correctly-rounded-to-1-ulp OCML versions on purpose: the fast intrinsics are
// biased near 1, and the bias is amplified 4x by every product level above.
template <int CMAX>
__global__ void spatial_sum_fwd_kernel(const float *__restrict__ x, const float *__restrict__ Wl,
                                       const float *__restrict__ LW, int64_t B, int Cin, int Cout, int HW,
                                       float *__restrict__ out) {
    const int64_t total = B * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const int64_t b = e / HW;
        const float *xp = x + b * Cin * HW + p;
        float ev[CMAX > 0 ? CMAX : 1];
        float m = -INFINITY;
        if (CMAX > 0) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                ev[c] = c < Cin ? xp[(int64_t)c * HW] : -INFINITY;
                m = fmaxf(m, ev[c]);
            }
        } else {
            for (int c = 0; c < Cin; ++c) m = fmaxf(m, xp[(int64_t)c * HW]);
        }
        const float m0 = (m == -INFINITY) ? 0.f : m;
        if (CMAX > 0) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) ev[c] = expf(ev[c] - m0);
        }
        for (int o = 0; o < Cout; ++o) {
            const float *wp = Wl + (int64_t)o * Cin * HW + p;
            float v = 0.f;
            if (CMAX > 0) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < Cin) v = fmaf(wp[(int64_t)c * HW], ev[c], v);
            } else {
                for (int c = 0; c < Cin; ++c) v = fmaf(wp[(int64_t)c * HW], expf(xp[(int64_t)c * HW] - m0), v);
            }
            float r;
            if (v < 1e-30f) {
                const float *lp = LW + (int64_t)o * Cin * HW + p;
                float mm = -INFINITY;
                for (int c = 0; c < Cin; ++c) mm = fmaxf(mm, xp[(int64_t)c * HW] + lp[(int64_t)c * HW]);
                if (mm > -INFINITY) {
                    float s = 0.f;
                    for (int c = 0; c < Cin; ++c) s += expf(xp[(int64_t)c * HW] + lp[(int64_t)c * HW] - mm);
                    r = mm + logf(s);
                } else {
                    r = -INFINITY;
                }
            } else {
                r = m0 + logf(v);
            }
            out[(b * Cout + o) * HW + p] = r;
        }
    }
}

// backward: pi = exp(x_c + lw[o,c,p] - out[o]);  gx[b,c,p] = sum_o g pi;  glw[o,c,p] = sum_b g pi
// thread = (c, p) for a slice of the batch: samples outer, output channels inner with the per-output weights and
// the batch sums of glw in registers (kSpOB outputs at a time), so x and gx are touched once per sample
constexpr int kSpOB = 16;
__global__ void spatial_sum_bwd_kernel(const float *__restrict__ x, const float *__restrict__ LW,
                                       const float *__restrict__ out, const float *__restrict__ g, int64_t B,
                                       int Cin, int Cout, int HW, int bslice, float *__restrict__ gx,
                                       float *__restrict__ glw) {
    const int64_t n = (int64_t)Cin * HW;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (c, p)
    if (e >= n) return;
    const int p = (int)(e % HW), c = (int)(e / HW);
    const int64_t b0 = (int64_t)blockIdx.y * bslice, b1 = min(b0 + bslice, B);
    for (int ob = 0; ob < Cout; ob += kSpOB) {
        float lw[kSpOB], acc[kSpOB];
#pragma unroll
        for (int q = 0; q < kSpOB; ++q) {
            lw[q] = (ob + q < Cout) ? LW[((int64_t)(ob + q) * Cin + c) * HW + p] : 0.f;
            acc[q] = 0.f;
        }
        for (int64_t b = b0; b < b1; ++b) {
            const float xv = x[(b * Cin + c) * HW + p];
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < kSpOB; ++q) {
                if (ob + q < Cout) {
                    const float xo = out[(b * Cout + ob + q) * HW + p];
                    const float t = (xo > -INFINITY) ? g[(b * Cout + ob + q) * HW + p] * expf(xv + lw[q] - xo) : 0.f;
                    acc[q] += t;
                    tot += t;
                }
            }
            if (gx) {
                float *dst = gx + (b * Cin + c) * HW + p;
                *dst = (ob == 0) ? tot : (*dst + tot);
            }
        }
        if (glw) {
#pragma unroll
            for (int q = 0; q < kSpOB; ++q)
                if (ob + q < Cout) atomicAdd(glw + ((int64_t)(ob + q) * Cin + c) * HW + p, acc[q]);
        }
    }
}

// Cin, Cout <= 8 (the DGC-SPN defaults): thread = pixel x half of the input channels x slice of samples, with the
// pixel's 8 x 4 linear weights and the batch sums of glw in registers.  pi[o,c] = W[o,c] e^{x_c - m} e^{m - out_o}
// with m = max_c x_c: 12 exponentials per thread and sample instead of 32, gx written once.  m - out_o is bounded
// by -log(max_c W[o,c]); an output whose bound leaves the fp32 range takes the exact log-domain expression.
// TAPS: the sum layer's input is the depthwise product of the map `x` [B,Cin,q.H,q.W] (dpk_spatial_prodsum_backward:
// the product map of the training forward is never stored); x_c is then the sum of the pixel's taps, gx the gradient
// of that product map.
constexpr int kSp8C = 4;
template <bool TAPS>
__global__ __launch_bounds__(256) void spatial_sum_bwd8_kernel(const float *__restrict__ x, const float *__restrict__ Wl,
                                                               const float *__restrict__ LW,
                                                               const float *__restrict__ out,
                                                               const float *__restrict__ g, int64_t B, int Cin,
                                                               int Cout, int HW, int bslice, float *__restrict__ gx,
                                                               float *__restrict__ glw, ProdGeom q) {
    const int p = min((int)(blockIdx.x * 64 + threadIdx.x), HW - 1);   // tail lanes shadow the last pixel, stores masked
    const bool own = (int)(blockIdx.x * 64 + threadIdx.x) < HW;
    const int c0 = blockIdx.z * kSp8C;
    // TAPS: offsets of the pixel's taps inside an input plane (clamped; padding taps contribute log 1 = 0)
    int tclamp[4] = {0, 0, 0, 0};
    bool tval[4] = {false, false, false, false};
    const int inHW = TAPS ? q.H * q.W : HW;
    if (TAPS) {
        const int oh = p / q.OW, ow = p - oh * q.OW, T = q.kh * q.kw;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int th = t / q.kw, tw = t - th * q.kw;
            const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
            tval[t] = t < T && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W;
            tclamp[t] = tval[t] ? ih * q.W + iw : 0;
        }
    }
    float w[8][kSp8C], acc[8][kSp8C];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < kSp8C; ++q) {
            w[o][q] = (o < Cout && c0 + q < Cin) ? Wl[((int64_t)o * Cin + c0 + q) * HW + p] : 0.f;
            acc[o][q] = 0.f;
        }
    // the slice index is uniform over the wave (blockDim.x = 64): sample offsets stay on the scalar unit
    const int slice = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * blockDim.y + threadIdx.y));
    const int64_t b0 = (int64_t)slice * bslice, b1 = min(b0 + bslice, B);
    float nx[8], nxo[8], ng[8];   // inputs of the next sample, in flight under the arithmetic of the current one
    const int cin1 = Cin - 1, cout1 = Cout - 1;
    // padding rows (c >= Cin, o >= Cout) read a valid row and are replaced when consumed: no branch per load
    auto fetch = [&](int64_t b) {
        const float *ob = out + b * Cout * HW + p, *gb = g + b * Cout * HW + p;
        if (TAPS) {
            const float *xb = x + b * Cin * inHW;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float *xc = xb + (int64_t)min(c, cin1) * inHW;
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float v = xc[tclamp[t]];
                    a += tval[t] ? v : 0.f;
                }
                nx[c] = a;
            }
        } else {
            const float *xb = x + b * Cin * HW + p;
#pragma unroll
            for (int c = 0; c < 8; ++c) nx[c] = xb[(int64_t)min(c, cin1) * HW];
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            nxo[o] = ob[(int64_t)min(o, cout1) * HW];
            ng[o] = gb[(int64_t)min(o, cout1) * HW];
        }
    };
    if (b0 < b1) fetch(b0);
    for (int64_t b = b0; b < b1; ++b) {
        float xa[8], xo8[8], g8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xa[i] = (i < Cin) ? nx[i] : -INFINITY;
            xo8[i] = (i < Cout) ? nxo[i] : -INFINITY;
            g8[i] = ng[i];
        }
        if (b + 1 < b1) fetch(b + 1);
        float xv[kSp8C], ex[kSp8C], tot[kSp8C];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) m = fmaxf(m, xa[c]);
#pragma unroll
        for (int q = 0; q < kSp8C; ++q) xv[q] = (blockIdx.z == 0) ? xa[q] : xa[kSp8C + q];
        const float mm = (m > -INFINITY) ? m : 0.f;
#pragma unroll
        for (int q = 0; q < kSp8C; ++q) {
            ex[q] = __expf(xv[q] - mm);
            tot[q] = 0.f;
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float xo = xo8[o], gv = g8[o];
            if (!(xo > -INFINITY)) continue;   // also the padding outputs o >= Cout
            const float d = mm - xo;
            if (d < 80.f) {
                const float eo = gv * __expf(d);
#pragma unroll
                for (int q = 0; q < kSp8C; ++q) {
                    const float t = w[o][q] * ex[q] * eo;
                    acc[o][q] += t;
                    tot[q] += t;
                }
            } else {
                const float *lwp = LW;
                asm volatile("" : "+s"(lwp));   // keeps the rare path's 32 addresses out of the loop's registers
#pragma unroll
                for (int q = 0; q < kSp8C; ++q) {
                    if (c0 + q >= Cin) break;
                    const float t = gv * expf(xv[q] + lwp[((int64_t)o * Cin + c0 + q) * HW + p] - xo);
                    acc[o][q] += t;
                    tot[q] += t;
                }
            }
        }
        if (gx) {
            float *gxb = gx + b * Cin * HW;
#pragma unroll
            for (int q = 0; q < kSp8C; ++q)
                if (own && c0 + q < Cin) gxb[(int64_t)(c0 + q) * HW + p] = tot[q];
        }
    }
    if (glw) {
        // the four sample slices of the work-group meet in LDS; one wave's worth of atomics per work-group
        __shared__ float red[3][8 * kSp8C][64];
        if (threadIdx.y > 0) {
#pragma unroll
            for (int o = 0; o < 8; ++o)
#pragma unroll
                for (int q = 0; q < kSp8C; ++q) red[threadIdx.y - 1][o * kSp8C + q][threadIdx.x] = acc[o][q];
        }
        __syncthreads();
        if (threadIdx.y == 0) {
#pragma unroll
            for (int o = 0; o < 8; ++o)
#pragma unroll
                for (int q = 0; q < kSp8C; ++q) {
                    const float v = acc[o][q] + red[0][o * kSp8C + q][threadIdx.x] +
                                    red[1][o * kSp8C + q][threadIdx.x] + red[2][o * kSp8C + q][threadIdx.x];
                    if (own && o < Cout && c0 + q < Cin) atomicAdd(glw + ((int64_t)o * Cin + c0 + q) * HW + p, v);
                }
        }
    }
}

// gW[o,c,p] = glw[o,c,p] - W[o,c,p] * sum_c glw[o,c,p]
__global__ void spatial_softmax_jacobian_kernel(const float *__restrict__ glw, const float *__restrict__ Wl,
                                                int Cout, int Cin, int HW, float *__restrict__ gW) {
    const int64_t n = (int64_t)Cout * HW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW), o = (int)(e / HW);
        float s = 0.f;
        for (int c = 0; c < Cin; ++c) s += glw[((int64_t)o * Cin + c) * HW + p];
        for (int c = 0; c < Cin; ++c) {
            const int64_t i = ((int64_t)o * Cin + c) * HW + p;
            gW[i] = glw[i] - Wl[i] * s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused depthwise product + sum level (the eval route of DgcSpn.forward): the [B,C,OH,OW] product map is
// never written.  Thread = one output pixel for NB consecutive samples, so each softmaxed weight
// W[o,c,p] is loaded once per NB samples; lanes run over pixels (coalesced taps, weights and stores).
//   out[b,o,p] = logsumexp_c( sum_taps in[b,c,tap(p)] + lw[o,c,p] )
// HBM traffic per level and sample: C*H*W read (the 4 taps of a pixel hit L2) + Cout*OH*OW written.
// ------------------------------------------------------------------------------------------------
template <int CMAX, int NB>
__global__ __launch_bounds__(256) void spatial_prodsum_fwd_kernel(const float *__restrict__ in,
                                                                   const float *__restrict__ Wl,
                                                                   const float *__restrict__ LW, int B, ProdGeom q,
                                                                   int Cout, float *__restrict__ out, int slots) {
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    // small maps (the pooled levels: 14 x 14, 7 x 7): `slots` groups of samples share a work-group, so that the pixel's
    // weights -- Cout x C floats per thread and sample group, re-read from L2 by every work-group -- serve slots x NB samples
    // (pt = pixels per work-group: 256 / slots, or the whole map when it is smaller)
    const int pt = slots > 1 ? min(256 / slots, OHW) : 256;
    const int slot = threadIdx.x / pt;
    const int p = blockIdx.x * pt + (threadIdx.x - slot * pt);
    if (slot >= slots || p >= OHW) return;
    const int b0 = (blockIdx.y * slots + slot) * NB;
    if (b0 >= B) return;
    const int oh = p / q.OW, ow = p - oh * q.OW;
    // tap offsets inside one channel plane, -1 = padding (contributes log 1 = 0)
    int toff[4];
    {
        const int T = q.kh * q.kw;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int th = t / q.kw, tw = t - th * q.kw;
            const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
            toff[t] = (t < T && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W) ? ih * q.W + iw : -1;
        }
    }
    // unconditional loads at clamped offsets + a select: no per-load exec-mask juggling at the map borders
    int tclamp[4];
    bool tval[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        tval[t] = toff[t] >= 0;
        tclamp[t] = max(toff[t], 0);
    }
    const bool full_c = (q.C == CMAX);
    float ev[NB][CMAX], m0[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const float *src = in + (size_t)min(b0 + s, B - 1) * q.C * HW;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const int cc = full_c ? c : min(c, q.C - 1);
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = src[cc * HW + tclamp[t]];
                a += tval[t] ? v : 0.f;
            }
            if (!full_c && c >= q.C) a = -INFINITY;
            ev[s][c] = a;
            m = fmaxf(m, a);
        }
        m0[s] = (m == -INFINITY) ? 0.f : m;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) ev[s][c] = __expf(ev[s][c] - m0[s]);
    }
    for (int o = 0; o < Cout; ++o) {
        float w[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) w[c] = c < q.C ? Wl[((size_t)o * q.C + c) * OHW + p] : 0.f;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            if (b0 + s >= B) break;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) v = fmaf(w[c], ev[s][c], v);
            float r;
            if (v < 1e-30f) {
                // exact log-domain pass (rare): rebuild the products from the taps
                const float *src = in + (size_t)(b0 + s) * q.C * HW;
                const float *lp = LW + (size_t)o * q.C * OHW + p;
                float mm = -INFINITY;
                for (int c = 0; c < q.C; ++c) {
                    float a = 0.f;
                    for (int t = 0; t < 4; ++t)
                        if (toff[t] >= 0) a += src[c * HW + toff[t]];
                    mm = fmaxf(mm, a + lp[(size_t)c * OHW]);
                }
                if (mm > -INFINITY) {
                    float acc = 0.f;
                    for (int c = 0; c < q.C; ++c) {
                        float a = 0.f;
                        for (int t = 0; t < 4; ++t)
                            if (toff[t] >= 0) a += src[c * HW + toff[t]];
                        acc += expf(a + lp[(size_t)c * OHW] - mm);
                    }
                    r = mm + logf(acc);
                } else {
                    r = -INFINITY;
                }
            } else {
                r = m0[s] + logf(v);
            }
            out[((size_t)(b0 + s) * Cout + o) * OHW + p] = r;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The same level for WIDE layers (16 / 32 input channels, up to 32 sum channels), round 5.  In the kernel above a thread
// reads the Cout x C weights of its pixel from global memory for every group of NB samples: one load per multiply-add,
// 3.3 GB of L2 traffic per launch at 32 -> 32 channels (c4b, the example model: 504 + 284 + 2 x 191 us for four levels
// whose multiply-adds are 70 us of vector-ALU time).  Here a work-group owns 16 pixels and keeps their softmaxed
// weights in LDS for a slice of the batch: [output][4 inputs][pixel][4] floats, so that the 16 pixels of a wave read 16
// consecutive 16-byte pieces (four lanes -- four sample slots -- share a piece: broadcast); a thread = (pixel, sample slot)
// walks the slice two samples at a time: 2 C exponentials in registers, 8 ds_read_b128 per output against 64
// multiply-adds.  The rare vanished node takes the exact log-domain form from the taps, as above.
// ------------------------------------------------------------------------------------------------
constexpr int kWidePix = 16;          // pixels per work-group (256 threads = 16 pixels x 16 sample slots)
// Work-group -> (pixel tile, slice of the batch).  The pixel tiles of ONE slice read the same input planes (a tile touches
// a tenth of each), and consecutive work-groups go to different XCDs, each with its own L2: dispatched tile-fastest, the 13
// tiles of a slice pulled every plane into up to eight L2s (c4b: 4.6 GB of traffic per step for 1.2 GB of maps).  The grid
// is one-dimensional: work-group L lands on XCD L mod 8 and takes slice (L / 8 / tiles) * 8 + L mod 8 -- all tiles of a
// slice on one XCD, one after the other.
// The tile's weights into LDS, [output][4 inputs][pixel][4]: a thread takes (output, 4 inputs) x 4 pixels -- four 16-byte
// loads along the pixels (when the map's rows of the weight tensor are 16-byte aligned: OHW % 4 == 0), transposed in
// registers, four 16-byte LDS writes.  (One 4-byte load and one 4-byte LDS write per element was 50 us of a 243 us level:
// every work-group loads 64 KB.)
template <int CIN>
__device__ __forceinline__ void wide_tile_weights(const float *__restrict__ Wl, float *wide_w, int C, int Cout, int OHW, int p0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int PT = kWidePix, C4 = CIN / 4;
    const bool vec = (OHW & 3) == 0 && p0 + PT <= OHW && C == CIN && (reinterpret_cast<uintptr_t>(Wl) & 15) == 0;
    if (vec) {
        for (int e = threadIdx.x; e < Cout * C4 * (PT / 4); e += 256) {
            const int pq = e % (PT / 4), oc4 = e / (PT / 4);   // oc4 = o * C4 + c4
            const int o = oc4 / C4, c4 = oc4 - o * C4;
            f32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                r[i] = *reinterpret_cast<const f32x4 *>(Wl + ((size_t)o * C + 4 * c4 + i) * OHW + p0 + 4 * pq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = {r[0][j], r[1][j], r[2][j], r[3][j]};
                *reinterpret_cast<f32x4 *>(wide_w + ((size_t)oc4 * PT + 4 * pq + j) * 4) = t;
            }
        }
    } else {
        for (int e = threadIdx.x; e < Cout * CIN * PT; e += 256) {
            const int pp = e % PT, oc = e / PT;                  // oc = o * CIN + c
            const int o = oc / CIN, c = oc - o * CIN;
            const float w = (p0 + pp < OHW && c < C) ? Wl[((size_t)o * C + c) * OHW + p0 + pp] : 0.f;
            wide_w[((o * C4 + (c >> 2)) * PT + pp) * 4 + (c & 3)] = w;
        }
    }
}
__device__ __forceinline__ bool wide_block_of(const ProdGeom &q, int &tile, int &chunk, int B, int per_wg) {
    const int tiles = (q.OH * q.OW + kWidePix - 1) / kWidePix;
    const int L = (int)blockIdx.x, x = L & 7, i = L >> 3;
    tile = i % tiles;
    chunk = (i / tiles) * 8 + x;
    return chunk * per_wg < B;
}
template <int CIN, bool PAIRS>
__global__ __launch_bounds__(256) void spatial_prodsum_wide_kernel(const float *__restrict__ in, const float *__restrict__ Wl,
                                                                    const float *__restrict__ LW, int B, ProdGeom q, int Cout,
                                                                    float *__restrict__ out, int per_wg) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float wide_w[];   // [Cout][CIN / 4][kWidePix][4]
    constexpr int PT = kWidePix, C4 = CIN / 4;
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    const int px = threadIdx.x & (PT - 1), slot = threadIdx.x / PT;
    int tile_x, chunk_y;
    if (!wide_block_of(q, tile_x, chunk_y, B, per_wg)) return;
    const int p0 = tile_x * PT;
    const int p = p0 + px;
    const bool has_p = p < OHW;
    // ---- the tile's weights: thread -> (output, input) rows of 16 consecutive pixels ---------------------------------------
    wide_tile_weights<CIN>(Wl, wide_w, q.C, Cout, OHW, p0);
    __syncthreads();
    if (!has_p) return;
    const int oh = p / q.OW, ow = p - oh * q.OW;
    int tclamp[4];
    bool tval[4];
    {
        const int T = q.kh * q.kw;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int th = t / q.kw, tw = t - th * q.kw;
            const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
            tval[t] = t < T && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W;
            tclamp[t] = tval[t] ? ih * q.W + iw : 0;
        }
    }
    const int b_begin = chunk_y * per_wg, b_end = min(B, b_begin + per_wg);
    // Horizontally adjacent taps (2 x 2 window, dilation 1 along the row, no padding: the pooling levels) travel as ONE
    // 8-byte load per tap row: the level is bound by the number of load instructions (a compute unit's request path takes
    // ~37 cycles each), not by bytes.  (With padding the pair has to be clamped into the row and its elements selected
    // per lane: measured slower than four loads.)
    constexpr bool paired = PAIRS;
    int pbase[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) pbase[r] = (oh * q.sh + r * q.dh) * q.W + ow * q.sw;
    constexpr int NB = 2;
    for (int b0 = b_begin + slot * NB; b0 < b_end; b0 += (256 / PT) * NB) {
        float ev[NB][CIN], m0[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const float *src = in + (size_t)min(b0 + s, B - 1) * q.C * HW;
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                float a = 0.f;
                if (paired) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                        const f32x2u v = *reinterpret_cast<const f32x2u *>(src + (size_t)c * HW + pbase[r]);
                        a += v[0] + v[1];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float v = src[(size_t)c * HW + tclamp[t]];
                        a += tval[t] ? v : 0.f;
                    }
                }
                ev[s][c] = a;
                m = fmaxf(m, a);
            }
            m0[s] = (m == -INFINITY) ? 0.f : m;
#pragma unroll
            for (int c = 0; c < CIN; ++c) ev[s][c] = __expf(ev[s][c] - m0[s]);
        }
#pragma unroll 4
        for (int o = 0; o < Cout; ++o) {
            float v[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) v[s] = 0.f;
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(wide_w) + (size_t)o * C4 * PT + px;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
                const f32x4 w = wp[c4 * PT];
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    v[s] = fmaf(w[0], ev[s][4 * c4], v[s]);
                    v[s] = fmaf(w[1], ev[s][4 * c4 + 1], v[s]);
                    v[s] = fmaf(w[2], ev[s][4 * c4 + 2], v[s]);
                    v[s] = fmaf(w[3], ev[s][4 * c4 + 3], v[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                if (b0 + s >= b_end) break;
                float r;
                if (v[s] < 1e-30f) {
                    // exact log-domain pass (rare): rebuild the products from the taps
                    const float *src = in + (size_t)(b0 + s) * q.C * HW;
                    const float *lp = LW + (size_t)o * q.C * OHW + p;
                    float mm = -INFINITY;
                    for (int c = 0; c < q.C; ++c) {
                        float a = 0.f;
                        for (int t = 0; t < 4; ++t)
                            if (tval[t]) a += src[(size_t)c * HW + tclamp[t]];
                        mm = fmaxf(mm, a + lp[(size_t)c * OHW]);
                    }
                    if (mm > -INFINITY) {
                        float acc = 0.f;
                        for (int c = 0; c < q.C; ++c) {
                            float a = 0.f;
                            for (int t = 0; t < 4; ++t)
                                if (tval[t]) a += src[(size_t)c * HW + tclamp[t]];
                            acc += expf(a + lp[(size_t)c * OHW] - mm);
                        }
                        r = mm + logf(acc);
                    } else {
                        r = -INFINITY;
                    }
                } else {
                    r = fmaf(__builtin_amdgcn_logf(v[s]), 0.6931471805599453f, m0[s]);   // (v >= 1e-30: a normal number; v_log_f32 is good to 1 ulp)
                }
                out[((size_t)(b0 + s) * Cout + o) * OHW + p] = r;
            }
        }
    }
}

// The pooling level with an even output width (2 x 2 window, stride 2, no padding: the model's first level): a thread owns
// TWO horizontally adjacent output pixels of one sample -- their eight taps are two aligned 16-byte loads per channel, their
// results one 8-byte store per output: half the memory instructions of the kernel above per pixel and sample (the level is
// bound by their number).  256 threads = 8 pixel pairs x 32 sample slots; weights in LDS as above.
template <int CIN>
__global__ __launch_bounds__(256) void spatial_prodsum_wide_pool_kernel(const float *__restrict__ in, const float *__restrict__ Wl,
                                                                         const float *__restrict__ LW, int B, ProdGeom q, int Cout,
                                                                         float *__restrict__ out, int per_wg) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float wide_w[];   // [Cout][CIN / 4][kWidePix][4]
    constexpr int PT = kWidePix, C4 = CIN / 4, NP = PT / 2;
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    const int pr = threadIdx.x & (NP - 1), slot = threadIdx.x / NP;
    int tile_x, chunk_y;
    if (!wide_block_of(q, tile_x, chunk_y, B, per_wg)) return;
    const int p0 = tile_x * PT;
    const int p = p0 + 2 * pr;                               // (OW even, p0 even: p and p + 1 share a row)
    wide_tile_weights<CIN>(Wl, wide_w, q.C, Cout, OHW, p0);
    __syncthreads();
    if (p >= OHW) return;
    const int oh = p / q.OW, ow = p - oh * q.OW;
    const int base0 = (oh * 2) * q.W + ow * 2, base1 = base0 + q.W;      // four consecutive inputs per tap row
    const int b_begin = chunk_y * per_wg, b_end = min(B, b_begin + per_wg);
    for (int b = b_begin + slot; b < b_end; b += 256 / NP) {
        const float *src = in + (size_t)b * q.C * HW;
        float ev[2][CIN], m0[2];
        float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(src + (size_t)c * HW + base0);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(src + (size_t)c * HW + base1);
            ev[0][c] = (r0[0] + r0[1]) + (r1[0] + r1[1]);
            ev[1][c] = (r0[2] + r0[3]) + (r1[2] + r1[3]);
            ma = fmaxf(ma, ev[0][c]);
            mb = fmaxf(mb, ev[1][c]);
        }
        m0[0] = (ma == -INFINITY) ? 0.f : ma;
        m0[1] = (mb == -INFINITY) ? 0.f : mb;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            ev[0][c] = __expf(ev[0][c] - m0[0]);
            ev[1][c] = __expf(ev[1][c] - m0[1]);
        }
        for (int o = 0; o < Cout; ++o) {
            float v[2] = {0.f, 0.f};
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(wide_w) + (size_t)o * C4 * PT + 2 * pr;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x4 w = wp[c4 * PT + s];
                    v[s] = fmaf(w[0], ev[s][4 * c4], v[s]);
                    v[s] = fmaf(w[1], ev[s][4 * c4 + 1], v[s]);
                    v[s] = fmaf(w[2], ev[s][4 * c4 + 2], v[s]);
                    v[s] = fmaf(w[3], ev[s][4 * c4 + 3], v[s]);
                }
            }
            f32x2 r;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (v[s] < 1e-30f) {
                    // exact log-domain pass (rare): rebuild the products from the taps
                    const float *lp = LW + (size_t)o * q.C * OHW + p + s;
                    float mm = -INFINITY;
                    for (int c = 0; c < q.C; ++c) {
                        const float *t0 = src + (size_t)c * HW + base0 + 2 * s;
                        const float a = (t0[0] + t0[1]) + (t0[q.W] + t0[q.W + 1]);
                        mm = fmaxf(mm, a + lp[(size_t)c * OHW]);
                    }
                    if (mm > -INFINITY) {
                        float acc = 0.f;
                        for (int c = 0; c < q.C; ++c) {
                            const float *t0 = src + (size_t)c * HW + base0 + 2 * s;
                            const float a = (t0[0] + t0[1]) + (t0[q.W] + t0[q.W + 1]);
                            acc += expf(a + lp[(size_t)c * OHW] - mm);
                        }
                        r[s] = mm + logf(acc);
                    } else {
                        r[s] = -INFINITY;
                    }
                } else {
                    r[s] = fmaf(__builtin_amdgcn_logf(v[s]), 0.6931471805599453f, m0[s]);   // (v >= 1e-30: a normal number; v_log_f32 is good to 1 ulp)
                }
            }
            *reinterpret_cast<f32x2 *>(out + ((size_t)b * Cout + o) * OHW + p) = r;
        }
    }
}

// The model's FIRST level with the Gaussian leaf layer folded in (dpk_spatial_leaf_prodsum_forward): the leaf map --
// [B, K, H, W], the largest tensor of the example model, 411 MB written by the leaf kernel and read back by this level at
// B = 8192 -- is never formed.  Same mapping as the pooling kernel above (a thread = two adjacent output pixels of one
// sample); the eight input pixels of the pair come from the IMAGE (two 16-byte loads per image channel), and the leaf
// parameters of those eight pixels (mean, 1 / 2 s^2, -log s - log sqrt(2 pi): 16 bytes per leaf channel, image channel and
// pixel) live in LDS beside the tile's sum weights.  leaf = sum over the image channels of nan_to_num(log N(x; mu, s))
// (layers/dgcspn.py:101-120), then the window's four taps are added as the product layer does.
template <int CIN>
__global__ __launch_bounds__(256) void spatial_leaf_pool_prodsum_kernel(const float *__restrict__ x, const float *__restrict__ loc,
                                                                         const float *__restrict__ scale, int Cx,
                                                                         const float *__restrict__ Wl, const float *__restrict__ LW,
                                                                         int B, ProdGeom q, int Cout, float *__restrict__ out,
                                                                         int per_wg) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float wide_w[];   // [Cout][CIN / 4][kWidePix][4], then the leaf parameters
    constexpr int PT = kWidePix, C4 = CIN / 4, NP = PT / 2;
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    f32x4 *lp = reinterpret_cast<f32x4 *>(wide_w + (size_t)Cout * CIN * PT);   // [CIN * Cx][pair][row][4 pixels]: (mu, iv, cs, -)
    const int pr = threadIdx.x & (NP - 1), slot = threadIdx.x / NP;
    int tile_x, chunk_y;
    if (!wide_block_of(q, tile_x, chunk_y, B, per_wg)) return;
    const int p0 = tile_x * PT;
    const int p = p0 + 2 * pr;                               // (OW even, p0 even: p and p + 1 share a row)
    wide_tile_weights<CIN>(Wl, wide_w, q.C, Cout, OHW, p0);
    for (int e = threadIdx.x; e < CIN * Cx * NP * 8; e += 256) {
        const int j = e & 3, row = (e >> 2) & 1, pp = (e >> 3) % NP, kc = e / (8 * NP);
        const int pe = p0 + 2 * pp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pe < OHW) {
            const int oh = pe / q.OW, ow = pe - oh * q.OW;
            const int64_t o = (int64_t)kc * HW + (oh * 2 + row) * q.W + ow * 2 + j;
            const float sg = scale[o];
            v[0] = loc[o];
            v[1] = 0.5f / (sg * sg);
            v[2] = -logf(sg) - kLogSqrt2Pi;
        }
        lp[e] = v;
    }
    __syncthreads();
    if (p >= OHW) return;
    const int oh = p / q.OW, ow = p - oh * q.OW;
    const int base0 = (oh * 2) * q.W + ow * 2, base1 = base0 + q.W;      // four consecutive inputs per tap row
    const int b_begin = chunk_y * per_wg, b_end = min(B, b_begin + per_wg);
    for (int b = b_begin + slot; b < b_end; b += 256 / NP) {
        const float *src = x + (size_t)b * Cx * HW;
        float ev[2][CIN], m0[2];
#pragma unroll
        for (int k = 0; k < CIN; ++k) ev[0][k] = ev[1][k] = 0.f;
        for (int cx = 0; cx < Cx; ++cx) {
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(src + (size_t)cx * HW + base0);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(src + (size_t)cx * HW + base1);
#pragma unroll
            for (int k = 0; k < CIN; ++k) {
                const f32x4 *pk = lp + ((size_t)(k * Cx + cx) * NP + pr) * 8;
                float t[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = pk[j], c = pk[4 + j];
                    const float d0 = r0[j] - a[0], d1 = r1[j] - c[0];
                    t[j] = nan_to_num_f(fmaf(-(d0 * d0), a[1], a[2]));
                    t[4 + j] = nan_to_num_f(fmaf(-(d1 * d1), c[1], c[2]));
                }
                ev[0][k] += (t[0] + t[1]) + (t[4] + t[5]);
                ev[1][k] += (t[2] + t[3]) + (t[6] + t[7]);
            }
        }
        float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
        for (int k = 0; k < CIN; ++k) {
            ma = fmaxf(ma, ev[0][k]);
            mb = fmaxf(mb, ev[1][k]);
        }
        m0[0] = (ma == -INFINITY) ? 0.f : ma;
        m0[1] = (mb == -INFINITY) ? 0.f : mb;
        float la[2][CIN];                                    // (the log-domain products, kept for the rare exact form)
#pragma unroll
        for (int k = 0; k < CIN; ++k) {
            la[0][k] = ev[0][k];
            la[1][k] = ev[1][k];
            ev[0][k] = __expf(ev[0][k] - m0[0]);
            ev[1][k] = __expf(ev[1][k] - m0[1]);
        }
        for (int o = 0; o < Cout; ++o) {
            float v[2] = {0.f, 0.f};
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(wide_w) + (size_t)o * C4 * PT + 2 * pr;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x4 w = wp[c4 * PT + s];
                    v[s] = fmaf(w[0], ev[s][4 * c4], v[s]);
                    v[s] = fmaf(w[1], ev[s][4 * c4 + 1], v[s]);
                    v[s] = fmaf(w[2], ev[s][4 * c4 + 2], v[s]);
                    v[s] = fmaf(w[3], ev[s][4 * c4 + 3], v[s]);
                }
            }
            f32x2 r;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (v[s] < 1e-30f) {
                    // exact log-domain pass (rare)
                    const float *lw = LW + (size_t)o * q.C * OHW + p + s;
                    float mm = -INFINITY;
#pragma unroll
                    for (int k = 0; k < CIN; ++k) mm = fmaxf(mm, la[s][k] + lw[(size_t)k * OHW]);
                    if (mm > -INFINITY) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < CIN; ++k) acc += expf(la[s][k] + lw[(size_t)k * OHW] - mm);
                        r[s] = mm + logf(acc);
                    } else {
                        r[s] = -INFINITY;
                    }
                } else {
                    r[s] = fmaf(__builtin_amdgcn_logf(v[s]), 0.6931471805599453f, m0[s]);   // (v >= 1e-30: a normal number; v_log_f32 is good to 1 ulp)
                }
            }
            *reinterpret_cast<f32x2 *>(out + ((size_t)b * Cout + o) * OHW + p) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Last level of the eval route: depthwise product ('final' padding) folded into the root sum nodes.
//   out[b,k] = logsumexp_m( prod[b,m] + log_softmax(weight,1)[k,m] ),  m = (c, oh, ow) flattened,
//   prod[b,m] = sum of the window taps of in[b,c] -- never written.  One wave per sample, lanes stride m,
//   per-lane online log-sum-exp in the log domain for up to kRootK classes at a time, then a wave combine.
// ------------------------------------------------------------------------------------------------
// one 256-thread block per row (rows are thousands of entries long, there are only a few of them)
__device__ __forceinline__ void rowwise_logsoftmax_body(const float *__restrict__ w, int row, int n, float *__restrict__ LW) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *src = w + (int64_t)row * n;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, src[i]);
    mx = wave_reduce_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += expf(src[i] - mx);
    sum = wave_reduce_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float ls = logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = tid; i < n; i += 256) LW[(int64_t)row * n + i] = src[i] - mx - ls;
}
__global__ __launch_bounds__(256) void rowwise_logsoftmax_kernel(const float *__restrict__ w, int rows, int n,
                                                                  float *__restrict__ LW, const unsigned *gate = nullptr) {
    if (gate_closed(gate)) return;
    rowwise_logsoftmax_body(w, (int)blockIdx.x, n, LW);
}

// The softmaxed-weight tables of several levels of a model (and the root's log-softmax rows) in ONE launch
// (dpk_spatial_tables): DgcSpn.forward walks its levels in a Python loop (reference dgcspn.py:134-151) and every level's
// entry point rebuilt its table with a 9 us launch of its own before its main kernel.
constexpr int kSpTablesMax = 8;
struct SpTablesArgs {
    const float *w[kSpTablesMax];
    float *Wl[kSpTablesMax], *LW[kSpTablesMax];
    int Cout[kSpTablesMax], Cin[kSpTablesMax], HW[kSpTablesMax], blk0[kSpTablesMax + 1];
    int n;
    const float *rw;     // root weight [K, M] or null
    float *rLW;
    int K, M;
};
__global__ __launch_bounds__(256) void spatial_tables_many_kernel(const SpTablesArgs a) {
    const int bid = (int)blockIdx.x;
    if (bid >= a.blk0[a.n]) {   // (uniform per block) the root's rows
        rowwise_logsoftmax_body(a.rw, bid - a.blk0[a.n], a.M, a.rLW);
        return;
    }
    int l = 0;
    while (l + 1 < a.n && bid >= a.blk0[l + 1]) ++l;
    spatial_softmax_body(a.w[l], a.Cout[l], a.Cin[l], a.HW[l], a.Wl[l], a.LW[l], bid - a.blk0[l], a.blk0[l + 1] - a.blk0[l]);
}

constexpr int kRootK = 4;
__global__ __launch_bounds__(256) void spatial_prodroot_fwd_kernel(const float *__restrict__ in,
                                                                    const float *__restrict__ LW, int64_t B,
                                                                    ProdGeom q, int K, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int OHW = q.OH * q.OW, HW = q.H * q.W, M = q.C * OHW, T = q.kh * q.kw;
    const float *src = in + b * q.C * HW;
    for (int k0 = 0; k0 < K; k0 += kRootK) {
        float mx[kRootK], sm[kRootK];
#pragma unroll
        for (int j = 0; j < kRootK; ++j) {
            mx[j] = -INFINITY;
            sm[j] = 0.f;
        }
        for (int m = lane; m < M; m += 64) {
            const int c = m / OHW, p = m - c * OHW;
            const int oh = p / q.OW, ow = p - oh * q.OW;
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int th = t / q.kw, tw = t - th * q.kw;
                const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
                const bool ok = t < T && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W;
                const float v = src[c * HW + (ok ? ih * q.W + iw : 0)];
                a += ok ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kRootK; ++j) {
                if (k0 + j < K) {
                    const float t = a + LW[(int64_t)(k0 + j) * M + m];
                    if (t > mx[j]) {
                        sm[j] *= expf(mx[j] - t);
                        mx[j] = t;
                    }
                    if (mx[j] > -INFINITY) sm[j] += expf(t - mx[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kRootK; ++j) {
            if (k0 + j < K) {
                const float Mx = wave_reduce_max(mx[j]);
                const float part = (mx[j] > -INFINITY) ? sm[j] * expf(mx[j] - Mx) : 0.f;
                const float tot = wave_reduce_sum(part);
                if (lane == 0) out[b * K + k0 + j] = (Mx > -INFINITY) ? Mx + logf(tot) : -INFINITY;
            }
        }
    }
}

}  // namespace dpk

using namespace dpk;

static int spatial_gaussian_forward_impl(const float *x, const float *loc, const float *scale, int64_t B, int32_t K,
                                         int32_t C, int32_t H, int32_t W, float *out, void *stream, float drop_p,
                                         uint64_t seed) {
    DPK_REQUIRE(B >= 0 && K > 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "spatial_gaussian: bad sizes");
    DPK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DPK_EINVAL, "spatial_gaussian: dropout rate must be in [0, 1)");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && loc && scale && out, DPK_EINVAL, "spatial_gaussian: null pointer");
    const int64_t total = B * K * H * W;
    if (drop_p == 0.f && C <= kGaussC) {
        const int HW = H * W;
        if (HW % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
            const int64_t blocks = cdiv((int64_t)K * (HW / 4), 256);
            int64_t slices = cdiv(2048, blocks);                // about 8 k waves
            int64_t bslice = cdiv(B, slices);
            if (bslice < 8) bslice = 8;
            slices = cdiv(B, bslice);
            DPK_LAUNCH(spatial_gaussian_fwd_fast4_kernel, dim3((unsigned)blocks, 1, (unsigned)slices), dim3(256),
                       0, (hipStream_t)stream, x, loc, scale, B, K, C, HW, (int)bslice, out);
            DPK_CHECK_LAUNCH("spatial_gaussian_fwd_fast4_kernel");
            return DPK_OK;
        }
        const int64_t cols = cdiv(HW, 64), kb = cdiv(K, 4);
        int64_t slices = cdiv(4096, cols * kb);                 // about 16 k waves
        int64_t bslice = cdiv(B, slices);
        if (bslice < 8) bslice = 8;
        slices = cdiv(B, bslice);
        DPK_LAUNCH(spatial_gaussian_fwd_fast_kernel, dim3((unsigned)cols, (unsigned)kb, (unsigned)slices),
                           dim3(256), 0, (hipStream_t)stream, x, loc, scale, B, K, C, HW, (int)bslice, out);
        DPK_CHECK_LAUNCH("spatial_gaussian_fwd_fast_kernel");
        return DPK_OK;
    }
    DPK_LAUNCH(spatial_gaussian_fwd_kernel, dim3(grid_cap(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, loc, scale, B, K, C, H * W, out, drop_p, seed);
    DPK_CHECK_LAUNCH("spatial_gaussian_fwd_kernel");
    return DPK_OK;
}
extern "C" int dpk_spatial_gaussian_forward(const float *x, const float *loc, const float *scale, int64_t B,
                                            int32_t K, int32_t C, int32_t H, int32_t W, float *out, void *stream) {
    return spatial_gaussian_forward_impl(x, loc, scale, B, K, C, H, W, out, stream, 0.f, 0);
}
extern "C" int dpk_spatial_gaussian_forward_dropout(const float *x, const float *loc, const float *scale, int64_t B,
                                                    int32_t K, int32_t C, int32_t H, int32_t W, float drop_p,
                                                    uint64_t seed, float *out, void *stream) {
    return spatial_gaussian_forward_impl(x, loc, scale, B, K, C, H, W, out, stream, drop_p, seed);
}

static int spatial_gaussian_backward_impl(const float *x, const float *g, const float *loc, const float *scale,
                                          int64_t B, int32_t K, int32_t C, int32_t H, int32_t W, float *grad_loc,
                                          float *grad_scale, float *grad_x, void *stream, float drop_p, uint64_t seed) {
    DPK_REQUIRE(B >= 0 && K > 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "spatial_gaussian_backward: bad sizes");
    DPK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DPK_EINVAL, "spatial_gaussian_backward: dropout rate must be in [0, 1)");
    DPK_REQUIRE(loc && scale, DPK_EINVAL, "spatial_gaussian_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const size_t pbytes = (size_t)K * C * HW * 4;
    if (grad_loc) DPK_REQUIRE(hipMemsetAsync(grad_loc, 0, pbytes, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (grad_scale) DPK_REQUIRE(hipMemsetAsync(grad_scale, 0, pbytes, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && g, DPK_EINVAL, "spatial_gaussian_backward: null pointer");
    if (grad_x)
        DPK_LAUNCH(spatial_gaussian_bwd_x_kernel, dim3(grid_cap(B * C * HW, 256)), dim3(256), 0, st, x, g,
                           loc, scale, B, K, C, HW, grad_x, drop_p, seed);
    if (grad_loc || grad_scale) {
        const int bslice = 64;
        DPK_LAUNCH(spatial_gaussian_bwd_p_kernel, dim3(cdiv((int64_t)K * C * HW, 256), cdiv(B, bslice)),
                           dim3(256), 0, st, x, g, loc, scale, B, K, C, HW, bslice, grad_loc, grad_scale, drop_p, seed);
    }
    DPK_CHECK_LAUNCH("spatial_gaussian_bwd");
    return DPK_OK;
}
extern "C" int dpk_spatial_gaussian_backward(const float *x, const float *g, const float *loc, const float *scale,
                                             int64_t B, int32_t K, int32_t C, int32_t H, int32_t W, float *grad_loc,
                                             float *grad_scale, float *grad_x, void *stream) {
    return spatial_gaussian_backward_impl(x, g, loc, scale, B, K, C, H, W, grad_loc, grad_scale, grad_x, stream, 0.f, 0);
}
extern "C" int dpk_spatial_gaussian_backward_dropout(const float *x, const float *g, const float *loc,
                                                     const float *scale, int64_t B, int32_t K, int32_t C, int32_t H,
                                                     int32_t W, float drop_p, uint64_t seed, float *grad_loc,
                                                     float *grad_scale, float *grad_x, void *stream) {
    return spatial_gaussian_backward_impl(x, g, loc, scale, B, K, C, H, W, grad_loc, grad_scale, grad_x, stream, drop_p,
                                          seed);
}

static int make_geom(ProdGeom &q, int C, int H, int W, int OC, int OH, int OW, int kh, int kw, int sh, int sw,
                     int dh, int dw, int pt, int pl, int depthwise) {
    DPK_REQUIRE(C > 0 && H > 0 && W > 0 && OC > 0 && OH > 0 && OW > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 &&
                    dh > 0 && dw > 0 && pt >= 0 && pl >= 0,
                DPK_EINVAL, "spatial_product: bad geometry");
    if (depthwise) {
        DPK_REQUIRE(OC == C, DPK_EINVAL, "spatial_product: depthwise needs OC == C");
    } else {
        int64_t want = 1;
        for (int t = 0; t < kh * kw; ++t) want *= C;
        DPK_REQUIRE(want == OC, DPK_EINVAL, "spatial_product: OC must be C^(kh*kw)");
    }
    q = ProdGeom{C, H, W, OC, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, depthwise};
    return DPK_OK;
}

extern "C" int dpk_spatial_product_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OC,
                                           int32_t OH, int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                           int32_t dh, int32_t dw, int32_t pad_top, int32_t pad_left,
                                           int32_t depthwise, float *out, void *stream) {
    ProdGeom q;
    int rc = make_geom(q, C, H, W, OC, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, depthwise);
    if (rc) return rc;
    if (B <= 0) return B == 0 ? DPK_OK : DPK_EINVAL;
    DPK_REQUIRE(in && out, DPK_EINVAL, "spatial_product: null pointer");
    DPK_LAUNCH(spatial_product_fwd_kernel, dim3(grid_cap(B * OC * OH * OW, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, B, q, out);
    DPK_CHECK_LAUNCH("spatial_product_fwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_spatial_product_backward(const float *g, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OC,
                                            int32_t OH, int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                            int32_t dh, int32_t dw, int32_t pad_top, int32_t pad_left,
                                            int32_t depthwise, float *grad_in, void *stream) {
    ProdGeom q;
    int rc = make_geom(q, C, H, W, OC, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, depthwise);
    if (rc) return rc;
    if (B <= 0) return B == 0 ? DPK_OK : DPK_EINVAL;
    DPK_REQUIRE(g && grad_in, DPK_EINVAL, "spatial_product_backward: null pointer");
    launch_product_bwd(g, B, q, grad_in, (hipStream_t)stream);
    DPK_CHECK_LAUNCH("spatial_product_bwd_kernel");
    return DPK_OK;
}

// workspace: W, LW, glw of [Cout, Cin, H, W]
extern "C" int64_t dpk_spatial_sum_workspace_bytes(int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DPK_EINVAL;
    return 3 * align_up((int64_t)Cout * Cin * H * W * 4, 256);
}

extern "C" int dpk_spatial_sum_forward(const float *x, const float *weight, int64_t B, int32_t Cin, int32_t Cout,
                                       int32_t H, int32_t W, float *out, void *ws, int64_t ws_bytes, void *stream) {
    DPK_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DPK_EINVAL, "spatial_sum: bad sizes");
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "spatial_sum: null pointer");
    const int HW = H * W;
    const int64_t seg = align_up((int64_t)Cout * Cin * HW * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_sum: workspace too small");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out, DPK_EINVAL, "spatial_sum: null pointer");
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg);
    hipStream_t st = (hipStream_t)stream;
    DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * HW, 256)), dim3(256), 0, st, weight,
                       Cout, Cin, HW, Wl, LW);
    const dim3 grid(grid_cap(B * HW, 256)), block(256);
    if (Cin <= 8)
        DPK_LAUNCH(spatial_sum_fwd_kernel<8>, grid, block, 0, st, x, Wl, LW, B, Cin, Cout, HW, out);
    else if (Cin <= 16)
        DPK_LAUNCH(spatial_sum_fwd_kernel<16>, grid, block, 0, st, x, Wl, LW, B, Cin, Cout, HW, out);
    else if (Cin <= 32)
        DPK_LAUNCH(spatial_sum_fwd_kernel<32>, grid, block, 0, st, x, Wl, LW, B, Cin, Cout, HW, out);
    else
        DPK_LAUNCH(spatial_sum_fwd_kernel<0>, grid, block, 0, st, x, Wl, LW, B, Cin, Cout, HW, out);
    DPK_CHECK_LAUNCH("spatial_sum_fwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_spatial_sum_backward(const float *x, const float *weight, const float *out, const float *g,
                                        int64_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, float *grad_x,
                                        float *grad_weight, void *ws, int64_t ws_bytes, void *stream) {
    DPK_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DPK_EINVAL, "spatial_sum_backward: bad sizes");
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "spatial_sum_backward: null pointer");
    const int HW = H * W;
    const int64_t seg = align_up((int64_t)Cout * Cin * HW * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_sum_backward: workspace too small");
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg), *glw = (float *)((char *)ws + 2 * seg);
    hipStream_t st = (hipStream_t)stream;
    DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * HW, 256)), dim3(256), 0, st, weight,
                       Cout, Cin, HW, Wl, LW);
    if (grad_weight)
        DPK_REQUIRE(hipMemsetAsync(glw, 0, (size_t)Cout * Cin * HW * 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B > 0) {
        DPK_REQUIRE(x && out && g, DPK_EINVAL, "spatial_sum_backward: null pointer");
        if (Cin <= 8 && Cout <= 8) {
            // about 8192 waves: (pixels / 64) columns x channel halves x sample slices of at least 8
            const int64_t cols = cdiv(HW, 64), halves = cdiv(Cin, kSp8C);
            int64_t slices = cdiv(8192, cols * halves);
            int64_t bslice = cdiv(B, slices);
            if (bslice < 8) bslice = 8;
            slices = cdiv(B, bslice);
            DPK_LAUNCH(spatial_sum_bwd8_kernel<false>, dim3((unsigned)cols, (unsigned)cdiv(slices, 4), (unsigned)halves),
                       dim3(64, 4), 0, st, x, Wl, LW, out, g, B, Cin, Cout, HW, (int)bslice, grad_x,
                       grad_weight ? glw : nullptr, ProdGeom{});
        } else {
            const int bslice = 16;
            DPK_LAUNCH(spatial_sum_bwd_kernel, dim3(cdiv((int64_t)Cin * HW, 256), cdiv(B, bslice)), dim3(256),
                               0, st, x, LW, out, g, B, Cin, Cout, HW, bslice, grad_x, grad_weight ? glw : nullptr);
        }
    }
    if (grad_weight)
        DPK_LAUNCH(spatial_softmax_jacobian_kernel, dim3(grid_cap((int64_t)Cout * HW, 256)), dim3(256), 0, st,
                           glw, Wl, Cout, Cin, HW, grad_weight);
    DPK_CHECK_LAUNCH("spatial_sum_bwd_kernel");
    return DPK_OK;
}

// Autograd of the fused depthwise product + sum level (training route of a DGC-SPN level whose forward ran as
// dpk_spatial_prodsum_forward): the product map is recomputed from the taps, its gradient lands in grad_prod
// [B,C,OH,OW] (caller scratch) and is scattered to grad_in by the product layer's backward kernel.
extern "C" int dpk_spatial_prodsum_backward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH,
                                            int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh,
                                            int32_t dw, int32_t pad_top, int32_t pad_left, const float *weight,
                                            int32_t Cout, const float *out, const float *g, float *grad_prod,
                                            float *grad_in, float *grad_weight, void *ws, int64_t ws_bytes,
                                            uint32_t flags, void *stream) {
    ProdGeom q;
    int rc = make_geom(q, C, H, W, C, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, 1);
    if (rc) return rc;
    DPK_REQUIRE(B >= 0 && Cout > 0, DPK_EINVAL, "spatial_prodsum_backward: bad sizes");
    DPK_REQUIRE(kh * kw <= 4 && C <= 8 && Cout <= 8, DPK_EUNSUPPORTED,
                "spatial_prodsum_backward: taps=%d channels=%d/%d outside the fused kernel", kh * kw, C, Cout);
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "spatial_prodsum_backward: null pointer");
    const int HW = OH * OW;
    const int64_t seg = align_up((int64_t)Cout * C * HW * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_prodsum_backward: workspace too small");
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg), *glw = (float *)((char *)ws + 2 * seg);
    hipStream_t st = (hipStream_t)stream;
    // (DPK_FLAG_PARAMS_CACHED: the forward of this very node left its tables in the workspace)
    if (!(flags & DPK_FLAG_PARAMS_CACHED))
        DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * HW, 256)), dim3(256), 0, st, weight, Cout, C, HW,
                   Wl, LW);
    if (grad_weight)
        DPK_REQUIRE(hipMemsetAsync(glw, 0, (size_t)Cout * C * HW * 4, st) == hipSuccess, DPK_ELAUNCH, "memset");
    if (B > 0) {
        DPK_REQUIRE(in && out && g && (grad_prod || !grad_in), DPK_EINVAL, "spatial_prodsum_backward: null pointer");
        const int64_t cols = cdiv(HW, 64), halves = cdiv(C, kSp8C);
        int64_t slices = cdiv(8192, cols * halves);
        int64_t bslice = cdiv(B, slices);
        if (bslice < 8) bslice = 8;
        slices = cdiv(B, bslice);
        DPK_LAUNCH(spatial_sum_bwd8_kernel<true>, dim3((unsigned)cols, (unsigned)cdiv(slices, 4), (unsigned)halves),
                   dim3(64, 4), 0, st, in, Wl, LW, out, g, B, C, Cout, HW, (int)bslice, grad_in ? grad_prod : nullptr,
                   grad_weight ? glw : nullptr, q);
        if (grad_in) launch_product_bwd(grad_prod, B, q, grad_in, st);
    }
    if (grad_weight)
        DPK_LAUNCH(spatial_softmax_jacobian_kernel, dim3(grid_cap((int64_t)Cout * HW, 256)), dim3(256), 0, st, glw, Wl,
                   Cout, C, HW, grad_weight);
    DPK_CHECK_LAUNCH("spatial_prodsum_backward");
    return DPK_OK;
}

static bool sumprodroot_shape_ok(const ProdGeom &q5, const ProdGeom &q6, int C, int Cout);
// Will a level of these shapes run on the streaming route (the one that takes / leaves pixel-major maps)?
extern "C" int32_t dpk_spatial_level_streams(int32_t last, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *geom5,
                                            int32_t Cout, const int32_t *geom6, int32_t K) {
    if (!geom5 || B <= 0) return 0;
    ProdGeom q5, q6;
    if (make_geom(q5, C, H, W, C, geom5[0], geom5[1], geom5[2], geom5[3], geom5[4], geom5[5], geom5[6], geom5[7], geom5[8],
                  geom5[9], 1))
        return 0;
    if (!last) return stream_prodsum_ok(q5, Cout, B, nullptr) ? 1 : 0;
    if (!geom6 ||
        make_geom(q6, Cout, geom5[0], geom5[1], Cout, geom6[0], geom6[1], geom6[2], geom6[3], geom6[4], geom6[5], geom6[6],
                  geom6[7], geom6[8], geom6[9], 1))
        return 0;
    return (sumprodroot_shape_ok(q5, q6, C, Cout) && stream_sumprodroot_partial_bytes(q5, Cout, q6, K, B) > 0) ? 1 : 0;
}

// SpatialProductLayer (depthwise, <= 4 taps) followed by SpatialSumLayer, product map kept in registers.
// DPK_EUNSUPPORTED for non-depthwise products, more than 4 taps or more than 32 channels: the caller chains
// dpk_spatial_product_forward + dpk_spatial_sum_forward instead.
extern "C" int dpk_spatial_prodsum_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH,
                                           int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh,
                                           int32_t dw, int32_t pad_top, int32_t pad_left, const float *weight,
                                           int32_t Cout, float *out, void *ws, int64_t ws_bytes, uint32_t flags,
                                           void *stream) {
    ProdGeom q;
    int rc = make_geom(q, C, H, W, C, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, 1);
    if (rc) return rc;
    DPK_REQUIRE(B >= 0 && Cout > 0, DPK_EINVAL, "spatial_prodsum: bad sizes");
    DPK_REQUIRE(kh * kw <= 4 && C <= 32, DPK_EUNSUPPORTED, "spatial_prodsum: taps=%d channels=%d not fused",
                kh * kw, C);
    DPK_REQUIRE(B <= INT32_MAX / 2 && (int64_t)B * (C > Cout ? C : Cout) * (int64_t)(H * W > OH * OW ? H * W : OH * OW) <
                                          ((int64_t)1 << 46),
                DPK_EUNSUPPORTED, "spatial_prodsum: tensor too large");
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "spatial_prodsum: null pointer");
    const int OHW = OH * OW;
    const int64_t seg = align_up((int64_t)Cout * C * OHW * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_prodsum: workspace too small");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && out, DPK_EINVAL, "spatial_prodsum: null pointer");
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg);
    hipStream_t st = (hipStream_t)stream;
    // (DPK_FLAG_PARAMS_CACHED: the workspace still holds the tables an earlier call built from this very weight;
    // DPK_FLAG_PARAMS_VERIFY: believed so, checked on the device)
    // (rebuilding IS the check here: the softmax pass costs what a fingerprint of the same bytes would)
    if (!(flags & DPK_FLAG_PARAMS_CACHED))
        DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * OHW, 256)), dim3(256), 0, st, weight,
                   Cout, C, OHW, Wl, LW);
    // large batches of the 8 -> 8 channel level: pixel-resident weights, taps staged through LDS (dgcspn_stream.hip)
    const bool in_pm = (flags & DPK_FLAG_IN_PIXEL_MAJOR) != 0, out_pm = (flags & DPK_FLAG_OUT_PIXEL_MAJOR) != 0;
    if (stream_prodsum_ok(q, Cout, B, in)) return stream_prodsum_forward(in, B, q, Wl, LW, out, st, in_pm, out_pm);
    DPK_REQUIRE(!in_pm && !out_pm, DPK_EUNSUPPORTED, "spatial_prodsum: pixel-major maps only on the streaming route (dpk_spatial_level_streams)");
    const int Bi = (int)B;
    hipEvent_t pev0, pev1;
    profile_take(&pev0, &pev1, DPK_KERNEL_SPATIAL_PRODSUM);
    if (pev0) (void)hipEventRecord(pev0, st);
    // Wide levels (C >= 16): a thread reads Cout x C weights of its pixel per NB samples -- 800 KB per work-group at
    // 32 -> 32 channels on a 14 x 14 map, 3.3 GB of L2 traffic per launch with one sample group per work-group (round-4
    // trace: 865 us).  There a work-group is 16 pixels x 16 groups of samples: the 16 groups read the same weights.
    // wide levels with exactly 16 / 32 input channels: the tile's weights in LDS (spatial_prodsum_wide_kernel)
    static const bool wide_off = [] { const char *e = getenv("DPK_DGC_WIDE"); return e && atoi(e) == 0; }();
    if (!wide_off && (C == 16 || C == 32) && Cout <= 32) {
        const int ptiles = cdiv(OHW, kWidePix);
        // samples per work-group: enough work-groups for four rounds of the chip, at least 32 samples each
        int per_wg = (int)std::max<int64_t>(32, cdiv((int64_t)Bi * ptiles, 4 * (int64_t)device_cus()));
        per_wg = (int)align_up(per_wg, 32);
        const size_t lds = (size_t)Cout * C * kWidePix * 4;
        const dim3 wgrid(ptiles * (int)align_up(cdiv(Bi, per_wg), 8));   // (one-dimensional: wide_block_of)
        // (every tap inside the map, the two of a row adjacent)
        const bool pairs = kh == 2 && kw == 2 && dw == 1 && pad_top == 0 && pad_left == 0 && (OW - 1) * sw + 1 < W &&
                           (OH - 1) * sh + dh < H;
#define DPK_WIDE(CIN, PR)                                                                                                  \
    do {                                                                                                                   \
        if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&spatial_prodsum_wide_kernel<CIN, PR>), 64 * 1024)) return lrc; \
        DPK_LAUNCH((spatial_prodsum_wide_kernel<CIN, PR>), wgrid, dim3(256), lds, st, in, Wl, LW, Bi, q, Cout, out, per_wg);   \
    } while (0)
        // (the pooling level with an even output width: two pixels per thread, 16-byte tap loads -- rows 16-byte aligned)
        const bool pool2 = pairs && sh == 2 && sw == 2 && dh == 1 && (OW % 2) == 0 && (W % 4) == 0 && (OHW % 2) == 0 &&
                           (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;
        if (pool2 && (C == 16 || C == 32)) {
            if (C == 16) {
                if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&spatial_prodsum_wide_pool_kernel<16>), 64 * 1024)) return lrc;
                DPK_LAUNCH((spatial_prodsum_wide_pool_kernel<16>), wgrid, dim3(256), lds, st, in, Wl, LW, Bi, q, Cout, out, per_wg);
            } else {
                if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&spatial_prodsum_wide_pool_kernel<32>), 64 * 1024)) return lrc;
                DPK_LAUNCH((spatial_prodsum_wide_pool_kernel<32>), wgrid, dim3(256), lds, st, in, Wl, LW, Bi, q, Cout, out, per_wg);
            }
        } else if (C == 16) {
            if (pairs) DPK_WIDE(16, true); else DPK_WIDE(16, false);
        } else {
            if (pairs) DPK_WIDE(32, true); else DPK_WIDE(32, false);
        }
#undef DPK_WIDE
        if (pev1) (void)hipEventRecord(pev1, st);
        DPK_CHECK_LAUNCH("spatial_prodsum_wide_kernel");
        return DPK_OK;
    }
    int slots = 1, ptile = 256;
    if (C >= 16) {
        ptile = OHW < 16 ? OHW : 16;
        slots = 256 / ptile;
    } else if (OHW <= 128) {
        ptile = OHW;
        slots = 256 / OHW;
    }
#define DPK_PRODSUM(CMAX, NB)                                                                                      \
    DPK_LAUNCH((spatial_prodsum_fwd_kernel<CMAX, NB>), dim3(cdiv(OHW, ptile), cdiv(Bi, NB * slots)), dim3(256), 0, st, \
                       in, Wl, LW, Bi, q, Cout, out, slots)
    if (C <= 4)
        DPK_PRODSUM(4, 4);
    else if (C <= 8)
        DPK_PRODSUM(8, 4);
    else if (C <= 16)
        DPK_PRODSUM(16, 2);
    else
        DPK_PRODSUM(32, 1);   // (NB = 2 / 4: 256 registers or scratch, one wave per SIMD in a latency chain over the outputs -- measured slower)
#undef DPK_PRODSUM
    if (pev1) (void)hipEventRecord(pev1, st);
    DPK_CHECK_LAUNCH("spatial_prodsum_fwd_kernel");
    return DPK_OK;
}

// The same fusion for any 2 x 2 window (stride, dilation, padding as the product layer has them; 8, 16 or 32 leaf channels):
// thread = (pixel, sample slot) as in spatial_prodsum_wide_kernel, two samples at a time; the four taps of the pixel are
// four 4-byte loads of the IMAGE per image channel, their leaf parameters 16 bytes per (leaf channel, image channel, tap)
// in LDS ([..][pixel][tap]: a lane's four taps are one 64-byte run); a tap in the padding has zero parameters and adds
// nan_to_num(-(d^2) 0 + 0) = 0, the log 1 of the product layer's padding.
template <int CIN>
__global__ __launch_bounds__(256) void spatial_leaf_prodsum_kernel(const float *__restrict__ x, const float *__restrict__ loc,
                                                                    const float *__restrict__ scale, int Cx,
                                                                    const float *__restrict__ Wl, const float *__restrict__ LW, int B,
                                                                    ProdGeom q, int Cout, float *__restrict__ out, int per_wg) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float wide_w[];   // [Cout][CIN / 4][kWidePix][4], then the leaf parameters
    constexpr int PT = kWidePix, C4 = CIN / 4;
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    f32x4 *lp = reinterpret_cast<f32x4 *>(wide_w + (size_t)Cout * CIN * PT);   // [CIN * Cx][pixel][tap]: (mu, iv, cs, -)
    const int px = threadIdx.x & (PT - 1), slot = threadIdx.x / PT;
    int tile_x, chunk_y;
    if (!wide_block_of(q, tile_x, chunk_y, B, per_wg)) return;
    const int p0 = tile_x * PT;
    const int p = p0 + px;
    wide_tile_weights<CIN>(Wl, wide_w, q.C, Cout, OHW, p0);
    for (int e = threadIdx.x; e < CIN * Cx * PT * 4; e += 256) {
        const int t = e & 3, pp = (e >> 2) & (PT - 1), kc = e / (4 * PT);
        const int pe = p0 + pp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pe < OHW) {
            const int oh = pe / q.OW, ow = pe - oh * q.OW;
            const int th = t / q.kw, tw = t - th * q.kw;
            const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
            if (t < q.kh * q.kw && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W) {
                const int64_t o = (int64_t)kc * HW + ih * q.W + iw;
                const float sg = scale[o];
                v[0] = loc[o];
                v[1] = 0.5f / (sg * sg);
                v[2] = -logf(sg) - kLogSqrt2Pi;
            }
        }
        lp[e] = v;
    }
    __syncthreads();
    if (p >= OHW) return;
    const int oh = p / q.OW, ow = p - oh * q.OW;
    int tclamp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int th = t / q.kw, tw = t - th * q.kw;
        const int ih = oh * q.sh - q.pt + th * q.dh, iw = ow * q.sw - q.pl + tw * q.dw;
        tclamp[t] = (t < q.kh * q.kw && ih >= 0 && ih < q.H && iw >= 0 && iw < q.W) ? ih * q.W + iw : 0;
    }
    const int b_begin = chunk_y * per_wg, b_end = min(B, b_begin + per_wg);
    constexpr int NB = 2;
    for (int b0 = b_begin + slot * NB; b0 < b_end; b0 += (256 / PT) * NB) {
        float ev[NB][CIN], la[NB][CIN], m0[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const float *src = x + (size_t)min(b0 + s, B - 1) * Cx * HW;
#pragma unroll
            for (int k = 0; k < CIN; ++k) ev[s][k] = 0.f;
            for (int cx = 0; cx < Cx; ++cx) {
                float xv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[t] = src[(size_t)cx * HW + tclamp[t]];
#pragma unroll
                for (int k = 0; k < CIN; ++k) {
                    const f32x4 *pk = lp + ((size_t)(k * Cx + cx) * PT + px) * 4;
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 pr = pk[t];
                        const float d = xv[t] - pr[0];
                        a += nan_to_num_f(fmaf(-(d * d), pr[1], pr[2]));
                    }
                    ev[s][k] += a;
                }
            }
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < CIN; ++k) m = fmaxf(m, ev[s][k]);
            m0[s] = (m == -INFINITY) ? 0.f : m;
#pragma unroll
            for (int k = 0; k < CIN; ++k) {
                la[s][k] = ev[s][k];
                ev[s][k] = __expf(ev[s][k] - m0[s]);
            }
        }
        for (int o = 0; o < Cout; ++o) {
            float v[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) v[s] = 0.f;
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(wide_w) + (size_t)o * C4 * PT + px;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
                const f32x4 w = wp[c4 * PT];
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    v[s] = fmaf(w[0], ev[s][4 * c4], v[s]);
                    v[s] = fmaf(w[1], ev[s][4 * c4 + 1], v[s]);
                    v[s] = fmaf(w[2], ev[s][4 * c4 + 2], v[s]);
                    v[s] = fmaf(w[3], ev[s][4 * c4 + 3], v[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                if (b0 + s >= b_end) break;
                float r;
                if (v[s] < 1e-30f) {
                    // exact log-domain pass (rare)
                    const float *lw = LW + (size_t)o * q.C * OHW + p;
                    float mm = -INFINITY;
#pragma unroll
                    for (int k = 0; k < CIN; ++k) mm = fmaxf(mm, la[s][k] + lw[(size_t)k * OHW]);
                    if (mm > -INFINITY) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < CIN; ++k) acc += expf(la[s][k] + lw[(size_t)k * OHW] - mm);
                        r = mm + logf(acc);
                    } else {
                        r = -INFINITY;
                    }
                } else {
                    r = fmaf(__builtin_amdgcn_logf(v[s]), 0.6931471805599453f, m0[s]);
                }
                out[((size_t)(b0 + s) * Cout + o) * OHW + p] = r;
            }
        }
    }
}

// smallest leaf channel count that takes the general (non-pooling) form of the fused first level: measurement knob
static int &leaf_fuse_min_k_ref() {
    static int v = [] { const char *e = getenv("DPK_DGC_LEAF_FUSE_MIN_K"); return e ? atoi(e) : 16; }();
    return v;
}
extern "C" int32_t dpk_spatial_leaf_fuse_min_k(int32_t k) {
    int &v = leaf_fuse_min_k_ref();
    const int prev = v;
    if (k > 0) v = k;
    return prev;
}

// SpatialGaussianLayer + the first depthwise product + sum level in one launch (eval route; models/dgcspn.py:134-147 for
// i = 0, 1): 2 x 2 windows, 8 / 16 / 32 leaf channels, up to 32 sum channels -- DPK_EUNSUPPORTED otherwise (the caller runs
// dpk_spatial_gaussian_forward + dpk_spatial_prodsum_forward).  DPK_DGC_LEAF_FUSE_MIN_K (measurement only) raises the
// smallest channel count that takes this route.
extern "C" int dpk_spatial_leaf_prodsum_forward(const float *x, const float *loc, const float *scale, int64_t B, int32_t Cx,
                                                int32_t K, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t kh, int32_t kw,
                                                int32_t sh, int32_t sw, int32_t dh, int32_t dw, int32_t pad_top,
                                                int32_t pad_left, const float *weight, int32_t Cout, float *out, void *ws,
                                                int64_t ws_bytes, uint32_t flags, void *stream) {
    ProdGeom q;
    int rc = make_geom(q, K, H, W, K, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, 1);
    if (rc) return rc;
    DPK_REQUIRE(B >= 0 && Cout > 0 && Cx > 0, DPK_EINVAL, "spatial_leaf_prodsum: bad sizes");
    DPK_REQUIRE(weight && ws && loc && scale, DPK_EINVAL, "spatial_leaf_prodsum: null pointer");
    const int OHW = OH * OW;
    const size_t lds = (size_t)Cout * K * kWidePix * 4 + (size_t)K * Cx * kWidePix * 4 * 16;   // (both kernels: 64 parameter slots per (k, cx))
    const bool pool = (K == 16 || K == 32) && kh == 2 && kw == 2 && sh == 2 && sw == 2 && dh == 1 && dw == 1 &&
                      pad_top == 0 && pad_left == 0 && (OW - 1) * 2 + 1 < W && (OH - 1) * 2 + 1 < H && (OW % 2) == 0 &&
                      (W % 4) == 0 && (OHW % 2) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 7) == 0;
    // (the general form evaluates an input pixel's leaf once per window that taps it -- four times at stride 1: for 8 leaf
    // channels that costs what the leaf map's round trip saves, 186 us against 76 + 110 on config 4's first level, so the
    // 8-channel models keep the leaf kernel + the streaming level; DPK_DGC_LEAF_FUSE_MIN_K: measurement only)
    const int leaf_min_k = leaf_fuse_min_k_ref();   // (dpk_spatial_leaf_fuse_min_k: read once from the environment, set by tests)
    // Round 6: 8 -> 8 channels at streaming batch sizes -- the leaf layer inside the streaming level kernel (dgcspn_stream.hip,
    // INK = 2): the image is what a stage holds (1/8 of the DMA), a tap's leaf values are evaluated from it and the position's
    // parameters in LDS.  The leaf map's round trip (410 MB at B = 8192) and the leaf launch go; the level becomes bound by
    // its arithmetic instead of HBM.  DPK_DGC_LEAF_STREAM=0 (measurement) keeps leaf kernel + streaming level.
    static const bool leaf_stream_off = [] { const char *e = getenv("DPK_DGC_LEAF_STREAM"); return e && atoi(e) == 0; }();
    const bool leaf_stream = !leaf_stream_off && K == 8 && Cout == 8 && B > 0 && stream_leaf_prodsum_ok(q, Cout, B, x, Cx);
    DPK_REQUIRE(leaf_stream || !(flags & DPK_FLAG_OUT_PIXEL_MAJOR), DPK_EUNSUPPORTED,
                "spatial_leaf_prodsum: a pixel-major output only on the streaming route");
    if (leaf_stream) {
        const int64_t seg = align_up((int64_t)Cout * K * OHW * 4, 256);
        DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_leaf_prodsum: workspace too small");
        DPK_REQUIRE(x && out, DPK_EINVAL, "spatial_leaf_prodsum: null pointer");
        float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg);
        hipStream_t st = (hipStream_t)stream;
        if (!(flags & DPK_FLAG_PARAMS_CACHED))
            DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * OHW, 256)), dim3(256), 0, st, weight, Cout, K, OHW, Wl, LW);
        return stream_leaf_prodsum_forward(x, loc, scale, Cx, B, q, Wl, LW, out, st, (flags & DPK_FLAG_OUT_PIXEL_MAJOR) != 0);
    }
    const bool ok = (K == 8 || K == 16 || K == 32) && (pool || K >= leaf_min_k) && Cout <= 32 && kh == 2 && kw == 2 &&
                    lds <= 150 * 1024 && B <= INT32_MAX / 2;
    if (!ok) {
        set_error("spatial_leaf_prodsum: outside the fused first level's envelope");
        return DPK_EUNSUPPORTED;
    }
    const int64_t seg = align_up((int64_t)Cout * K * OHW * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "spatial_leaf_prodsum: workspace too small");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out, DPK_EINVAL, "spatial_leaf_prodsum: null pointer");
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & DPK_FLAG_PARAMS_CACHED))
        DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * OHW, 256)), dim3(256), 0, st, weight, Cout, K, OHW, Wl, LW);
    const int Bi = (int)B;
    const int ptiles = cdiv(OHW, kWidePix);
    int per_wg = (int)std::max<int64_t>(32, cdiv((int64_t)Bi * ptiles, 4 * (int64_t)device_cus()));
    per_wg = (int)align_up(per_wg, 32);
    const dim3 wgrid(ptiles * (int)align_up(cdiv(Bi, per_wg), 8));   // (one-dimensional: wide_block_of)
    hipEvent_t pev0, pev1;
    profile_take(&pev0, &pev1, DPK_KERNEL_SPATIAL_PRODSUM);
    if (pev0) (void)hipEventRecord(pev0, st);
#define DPK_LEAF_FUSED(KERN)                                                                                               \
    do {                                                                                                                   \
        if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&KERN), 160 * 1024)) return lrc;                  \
        DPK_LAUNCH((KERN), wgrid, dim3(256), lds, st, x, loc, scale, Cx, Wl, LW, Bi, q, Cout, out, per_wg);              \
    } while (0)
    if (pool && K == 16) DPK_LEAF_FUSED(spatial_leaf_pool_prodsum_kernel<16>);
    else if (pool) DPK_LEAF_FUSED(spatial_leaf_pool_prodsum_kernel<32>);
    else if (K == 8) DPK_LEAF_FUSED(spatial_leaf_prodsum_kernel<8>);
    else if (K == 16) DPK_LEAF_FUSED(spatial_leaf_prodsum_kernel<16>);
    else DPK_LEAF_FUSED(spatial_leaf_prodsum_kernel<32>);
#undef DPK_LEAF_FUSED
    if (pev1) (void)hipEventRecord(pev1, st);
    DPK_CHECK_LAUNCH("spatial_leaf_pool_prodsum_kernel");
    return DPK_OK;
}

// Depthwise SpatialProductLayer (<= 4 taps) followed by SpatialRootLayer (models/dgcspn.py:146-150 in eval mode):
// weight [K, C*OH*OW]; workspace = K*C*OH*OW floats (+256 B).
// ------------------------------------------------------------------------------------------------
// Last sum level + last product + root in one launch (models/dgcspn.py:146-150 at i = n-3 .. n-1).
// The last product ('final' padding, dilation d) reads the last sum layer's map at four positions a distance d
// apart and its d x d output only feeds the root: a thread owns one of those output pixels, evaluates the fused
// product + sum level (as spatial_prodsum_fwd_kernel) at its (up to) four tap positions, adds them, and the
// work-group (= NB samples) finishes the root's log-sum-exp in the log domain.  The largest map of the model
// (8 x 59 x 59 per sample at 28 x 28 inputs: 0.9 GB written and read back at B = 8192) never reaches memory.
// ------------------------------------------------------------------------------------------------
template <int CMAX, int NB>
__global__ __launch_bounds__(1024) void spatial_sumprodroot_fwd_kernel(const float *__restrict__ in,
                                                                       const float *__restrict__ Wl,
                                                                       const float *__restrict__ LW, int B,
                                                                       ProdGeom q5, int Cout, ProdGeom q6,
                                                                       const float *__restrict__ LWr, int K,
                                                                       float *__restrict__ out) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int OHW5 = q5.OH * q5.OW, HW = q5.H * q5.W, OHW6 = q6.OH * q6.OW;
    const bool act = tid < OHW6;
    const int b0 = blockIdx.x * NB;
    const bool full_c = (q5.C == CMAX);
    float P[NB][CMAX];
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int o = 0; o < CMAX; ++o) P[s][o] = 0.f;
    if (act) {
        const int oh6 = tid / q6.OW, ow6 = tid - oh6 * q6.OW;
        const int T6 = q6.kh * q6.kw;
#pragma unroll 1
        for (int t6 = 0; t6 < T6; ++t6) {
            const int th6 = t6 / q6.kw, tw6 = t6 - th6 * q6.kw;
            const int ph = oh6 * q6.sh - q6.pt + th6 * q6.dh, pw = ow6 * q6.sw - q6.pl + tw6 * q6.dw;
            if (ph < 0 || ph >= q6.H || pw < 0 || pw >= q6.W) continue;   // zero padding of the last product
            const int p = ph * q5.OW + pw;                                 // pixel of the last sum layer's map
            int tclamp[4];
            bool tval[4];
            {
                const int T = q5.kh * q5.kw;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int th = t / q5.kw, tw = t - th * q5.kw;
                    const int ih = ph * q5.sh - q5.pt + th * q5.dh, iw = pw * q5.sw - q5.pl + tw * q5.dw;
                    tval[t] = t < T && ih >= 0 && ih < q5.H && iw >= 0 && iw < q5.W;
                    tclamp[t] = tval[t] ? ih * q5.W + iw : 0;
                }
            }
            float ev[NB][CMAX], m0[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const float *src = in + (size_t)min(b0 + s, B - 1) * q5.C * HW;
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const int cc = full_c ? c : min(c, q5.C - 1);
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float v = src[cc * HW + tclamp[t]];
                        a += tval[t] ? v : 0.f;
                    }
                    if (!full_c && c >= q5.C) a = -INFINITY;
                    ev[s][c] = a;
                    m = fmaxf(m, a);
                }
                m0[s] = (m == -INFINITY) ? 0.f : m;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) ev[s][c] = __expf(ev[s][c] - m0[s]);
            }
#pragma unroll
            for (int o = 0; o < CMAX; ++o) {
                if (o >= Cout) break;
                float w[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) w[c] = c < q5.C ? Wl[((size_t)o * q5.C + c) * OHW5 + p] : 0.f;
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) v = fmaf(w[c], ev[s][c], v);
                    float r;
                    if (v < 1e-30f) {
                        // exact log-domain pass (rare): rebuild the products from the taps
                        const float *src = in + (size_t)min(b0 + s, B - 1) * q5.C * HW;
                        const float *lp = LW + (size_t)o * q5.C * OHW5 + p;
                        float mm = -INFINITY;
                        for (int c = 0; c < q5.C; ++c) {
                            float a = 0.f;
                            for (int t = 0; t < 4; ++t)
                                if (tval[t]) a += src[c * HW + tclamp[t]];
                            mm = fmaxf(mm, a + lp[(size_t)c * OHW5]);
                        }
                        if (mm > -INFINITY) {
                            float acc = 0.f;
                            for (int c = 0; c < q5.C; ++c) {
                                float a = 0.f;
                                for (int t = 0; t < 4; ++t)
                                    if (tval[t]) a += src[c * HW + tclamp[t]];
                                acc += expf(a + lp[(size_t)c * OHW5] - mm);
                            }
                            r = mm + logf(acc);
                        } else {
                            r = -INFINITY;
                        }
                    } else {
                        r = m0[s] + logf(v);
                    }
                    P[s][o] += r;
                }
            }
        }
    }
    // root (layers/dgcspn.py:343-355): log-sum-exp over (channel, pixel) of P + log_softmax(weight), per class
    const int M = Cout * OHW6;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            float tv[CMAX], tmax = -INFINITY;
#pragma unroll
            for (int o = 0; o < CMAX; ++o) {
                tv[o] = (act && o < Cout) ? P[s][o] + LWr[(size_t)k * M + o * OHW6 + tid] : -INFINITY;
                tmax = fmaxf(tmax, tv[o]);
            }
            tmax = wave_reduce_max(tmax);
            __syncthreads();
            if (lane == 0) red[wave] = tmax;
            __syncthreads();
            float bmax = -INFINITY;
            for (int w2 = 0; w2 < n_waves; ++w2) bmax = fmaxf(bmax, red[w2]);
            float part = 0.f;
            if (bmax > -INFINITY) {
#pragma unroll
                for (int o = 0; o < CMAX; ++o) part += expf(tv[o] - bmax);   // exp(-inf) = 0 for the padding
            }
            part = wave_reduce_sum(part);
            __syncthreads();
            if (lane == 0) red[wave] = part;
            __syncthreads();
            if (tid == 0 && b0 + s < B) {
                float tot = 0.f;
                for (int w2 = 0; w2 < n_waves; ++w2) tot += red[w2];
                out[(size_t)(b0 + s) * K + k] = (bmax > -INFINITY) ? bmax + logf(tot) : -INFINITY;
            }
        }
    }
}

extern "C" int64_t dpk_spatial_sumprodroot_workspace_bytes(int32_t C, int32_t Cout, int32_t OH5, int32_t OW5,
                                                           int32_t OH6, int32_t OW6, int32_t K) {
    if (C <= 0 || Cout <= 0 || OH5 <= 0 || OW5 <= 0 || OH6 <= 0 || OW6 <= 0 || K <= 0) return DPK_EINVAL;
    return 2 * align_up((int64_t)Cout * C * OH5 * OW5 * 4, 256) + align_up((int64_t)K * Cout * OH6 * OW6 * 4, 256) + 256;
}

// Workspace of dpk_spatial_sumprodroot_forward for a batch of B samples: the tables above plus, where the streaming
// kernel applies, one (max, sum) pair per sample, class and compute wave for the root's log-sum-exp.  geom5 / geom6 as
// in dpk_spatial_sumprodroot_forward.  With only dpk_spatial_sumprodroot_workspace_bytes() bytes the entry point
// runs the batch-independent kernel.
// the one statement of dpk_spatial_sumprodroot_forward's shape envelope
static bool sumprodroot_shape_ok(const ProdGeom &q5, const ProdGeom &q6, int C, int Cout) {
    return q5.kh * q5.kw <= 4 && q6.kh * q6.kw <= 4 && C <= 8 && Cout <= 8 && q6.OH * q6.OW <= 1024;
}

extern "C" int64_t dpk_spatial_sumprodroot_workspace_bytes_batch(int64_t B, int32_t C, int32_t H, int32_t W,
                                                                 const int32_t *geom5, int32_t Cout,
                                                                 const int32_t *geom6, int32_t K) {
    if (!geom5 || !geom6 || B < 0) return DPK_EINVAL;
    const int64_t base = dpk_spatial_sumprodroot_workspace_bytes(C, Cout, geom5[0], geom5[1], geom6[0], geom6[1], K);
    if (base < 0) return base;
    ProdGeom q5, q6;
    if (make_geom(q5, C, H, W, C, geom5[0], geom5[1], geom5[2], geom5[3], geom5[4], geom5[5], geom5[6], geom5[7],
                  geom5[8], geom5[9], 1) ||
        make_geom(q6, Cout, geom5[0], geom5[1], Cout, geom6[0], geom6[1], geom6[2], geom6[3], geom6[4], geom6[5],
                  geom6[6], geom6[7], geom6[8], geom6[9], 1))
        return DPK_EINVAL;
    // (the forward's envelope, answered here so that no caller keeps a copy of these constants: ADVICE r05)
    if (!sumprodroot_shape_ok(q5, q6, C, Cout)) return DPK_EUNSUPPORTED;
    return base + stream_sumprodroot_partial_bytes(q5, Cout, q6, K, B);
}

extern "C" int dpk_spatial_sumprodroot_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W,
                                               const int32_t *geom5, const float *sum_weight, int32_t Cout,
                                               const int32_t *geom6, const float *root_weight, int32_t K, float *out,
                                               void *ws, int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(geom5 && geom6, DPK_EINVAL, "spatial_sumprodroot: null geometry");
    // geom = {OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left} of a depthwise product layer
    ProdGeom q5, q6;
    int rc = make_geom(q5, C, H, W, C, geom5[0], geom5[1], geom5[2], geom5[3], geom5[4], geom5[5], geom5[6], geom5[7],
                       geom5[8], geom5[9], 1);
    if (rc) return rc;
    rc = make_geom(q6, Cout, geom5[0], geom5[1], Cout, geom6[0], geom6[1], geom6[2], geom6[3], geom6[4], geom6[5],
                   geom6[6], geom6[7], geom6[8], geom6[9], 1);
    if (rc) return rc;
    DPK_REQUIRE(B >= 0 && Cout > 0 && K > 0, DPK_EINVAL, "spatial_sumprodroot: bad sizes");
    DPK_REQUIRE(sumprodroot_shape_ok(q5, q6, C, Cout),
                DPK_EUNSUPPORTED, "spatial_sumprodroot: shape outside the fused kernel (<= 8 channels, <= 1024 pixels)");
    DPK_REQUIRE(B <= INT32_MAX / 2 && (int64_t)B * C * H * W < ((int64_t)1 << 46), DPK_EUNSUPPORTED,
                "spatial_sumprodroot: tensor too large");
    DPK_REQUIRE(sum_weight && root_weight && ws, DPK_EINVAL, "spatial_sumprodroot: null pointer");
    DPK_REQUIRE(ws_bytes >= dpk_spatial_sumprodroot_workspace_bytes(C, Cout, q5.OH, q5.OW, q6.OH, q6.OW, K),
                DPK_EWORKSPACE, "spatial_sumprodroot: workspace too small");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && out, DPK_EINVAL, "spatial_sumprodroot: null pointer");
    const int OHW5 = q5.OH * q5.OW, OHW6 = q6.OH * q6.OW;
    const int64_t seg = align_up((int64_t)Cout * C * OHW5 * 4, 256);
    float *Wl = (float *)ws, *LW = (float *)((char *)ws + seg), *LWr = (float *)((char *)ws + 2 * seg);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & DPK_FLAG_PARAMS_CACHED)) {   // (flags as in dpk_spatial_prodsum_forward: VERIFY = rebuild)
        DPK_LAUNCH(spatial_softmax_kernel, dim3(grid_cap((int64_t)Cout * OHW5, 256)), dim3(256), 0, st, sum_weight,
                   Cout, C, OHW5, Wl, LW);
        DPK_LAUNCH(rowwise_logsoftmax_kernel, dim3(K), dim3(256), 0, st, root_weight, K, Cout * OHW6, LWr);
    }
    {
        // streaming kernel when the workspace carries its per-wave partials (.._workspace_bytes_batch)
        const int64_t base = dpk_spatial_sumprodroot_workspace_bytes(C, Cout, q5.OH, q5.OW, q6.OH, q6.OW, K);
        const int64_t part = ((uintptr_t)in & 15) == 0 ? stream_sumprodroot_partial_bytes(q5, Cout, q6, K, B) : 0;
        if (part > 0 && ws_bytes >= base + part)
            return stream_sumprodroot_forward(in, B, q5, Wl, LW, q6, LWr, K, out, (char *)ws + base, st,
                                              (flags & DPK_FLAG_IN_PIXEL_MAJOR) != 0);
    }
    DPK_REQUIRE(!(flags & DPK_FLAG_IN_PIXEL_MAJOR), DPK_EUNSUPPORTED,
                "spatial_sumprodroot: a pixel-major input only on the streaming route (dpk_spatial_level_streams)");
    constexpr int kNB = 2;
    const int threads = (int)align_up(OHW6, 64);
    hipEvent_t pev0, pev1;
    profile_take(&pev0, &pev1, DPK_KERNEL_SPATIAL_SUMPRODROOT);
    if (pev0) (void)hipEventRecord(pev0, st);
    DPK_LAUNCH((spatial_sumprodroot_fwd_kernel<8, kNB>), dim3(cdiv((int)B, kNB)), dim3(threads), 0, st, in, Wl,
                       LW, (int)B, q5, Cout, q6, LWr, K, out);
    if (pev1) (void)hipEventRecord(pev1, st);
    DPK_CHECK_LAUNCH("spatial_sumprodroot_fwd_kernel");
    return DPK_OK;
}

extern "C" int64_t dpk_spatial_prodroot_workspace_bytes(int32_t C, int32_t OH, int32_t OW, int32_t K) {
    if (C <= 0 || OH <= 0 || OW <= 0 || K <= 0) return DPK_EINVAL;
    return align_up((int64_t)K * C * OH * OW * 4, 256) + 256;
}

extern "C" int dpk_spatial_prodroot_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH,
                                            int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh,
                                            int32_t dw, int32_t pad_top, int32_t pad_left, const float *weight,
                                            int32_t K, float *out, void *ws, int64_t ws_bytes, void *stream) {
    ProdGeom q;
    int rc = make_geom(q, C, H, W, C, OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left, 1);
    if (rc) return rc;
    DPK_REQUIRE(B >= 0 && K > 0, DPK_EINVAL, "spatial_prodroot: bad sizes");
    DPK_REQUIRE(kh * kw <= 4, DPK_EUNSUPPORTED, "spatial_prodroot: taps=%d not fused", kh * kw);
    DPK_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 30) && (int64_t)C * OH * OW < ((int64_t)1 << 30), DPK_EUNSUPPORTED,
                "spatial_prodroot: map too large");
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "spatial_prodroot: null pointer");
    const int M = C * OH * OW;
    DPK_REQUIRE(ws_bytes >= dpk_spatial_prodroot_workspace_bytes(C, OH, OW, K), DPK_EWORKSPACE,
                "spatial_prodroot: workspace too small");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && out, DPK_EINVAL, "spatial_prodroot: null pointer");
    hipStream_t st = (hipStream_t)stream;
    float *LW = (float *)ws;
    DPK_LAUNCH(rowwise_logsoftmax_kernel, dim3(K), dim3(256), 0, st, weight, K, M, LW);
    DPK_LAUNCH(spatial_prodroot_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, in, LW, B, q, K, out);
    DPK_CHECK_LAUNCH("spatial_prodroot_fwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_spatial_tables(int32_t n, const dpk_spatial_tables_args *levels, void *stream) {
    DPK_REQUIRE(n >= 0 && n <= kSpTablesMax, DPK_EINVAL, "spatial_tables: n = %d (0..%d)", n, kSpTablesMax);
    if (n == 0) return DPK_OK;
    DPK_REQUIRE(levels, DPK_EINVAL, "spatial_tables: null pointer");
    SpTablesArgs a{};
    a.n = n;
    int blocks = 0;
    for (int l = 0; l < n; ++l) {
        const dpk_spatial_tables_args &q = levels[l];
        DPK_REQUIRE(q.sum_weight && q.ws && q.C > 0 && q.Cout > 0 && q.OHW > 0, DPK_EINVAL, "spatial_tables: bad level %d", l);
        const int64_t seg = align_up((int64_t)q.Cout * q.C * q.OHW * 4, 256);
        DPK_REQUIRE(q.ws_bytes >= 2 * seg, DPK_EWORKSPACE, "spatial_tables: workspace of level %d too small", l);
        a.w[l] = q.sum_weight;
        a.Wl[l] = (float *)q.ws;
        a.LW[l] = (float *)((char *)q.ws + seg);
        a.Cout[l] = q.Cout; a.Cin[l] = q.C; a.HW[l] = q.OHW;
        a.blk0[l] = blocks;
        blocks += grid_cap((int64_t)q.Cout * q.OHW, 256, 64);
        if (q.root_weight) {
            DPK_REQUIRE(a.rw == nullptr && q.K > 0 && q.M > 0, DPK_EINVAL, "spatial_tables: one root per call");
            DPK_REQUIRE(q.ws_bytes >= 2 * seg + (int64_t)q.K * q.M * 4, DPK_EWORKSPACE, "spatial_tables: root workspace too small");
            a.rw = q.root_weight;
            a.rLW = (float *)((char *)q.ws + 2 * seg);
            a.K = q.K; a.M = q.M;
        }
    }
    a.blk0[n] = blocks;
    DPK_LAUNCH(spatial_tables_many_kernel, dim3(blocks + (a.rw ? a.K : 0)), dim3(256), 0, (hipStream_t)stream, a);
    DPK_CHECK_LAUNCH("spatial_tables_many_kernel");
    return DPK_OK;
}
