// RealNVP-2D training direction: the kernels autograd needs next to the evaluation kernels of flows2d.hip.
//
// The reference trains the 2-D flows through ATen's autograd over nn.BatchNorm2d (batch statistics), ReLU,
// weight-normalised F.conv2d (torch/utils.py:86-121 inside flows/layers/resnet.py:9-90), the coupling transformation
// (flows/layers/coupling.py:181-226) and BatchNormLayer2d's training branch (flows/utils.py:186-208).  Here:
//   * batch statistics of a [B, C, H, W] tensor per channel (sum, sum of squares in fp64) and their backward,
//   * the per-channel affine map a[c] x + b[c] (BatchNormLayer2d with batch statistics) and its backward, which is also
//     the backward of the BatchNorm2d + ReLU (+ mask) that dpk_conv2d_forward folds into its operand load,
//   * the weight gradient of the convolution (the input gradient is dpk_conv2d_forward with transposed, flipped weights),
//   * the backward of the coupling transformation in the density direction.
// NCHW fp32; reductions over the batch accumulate in fp64 atomics (order-independent to fp32 rounding).
#include "common.h"

namespace dpk {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

constexpr int kStatSpan = 256 * 16;  // elements of one channel per work-group

// sums[c] += sum x, sums[C + c] += sum x^2 over the channel's B*HW elements; grid (chunks, C)
__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ x, int64_t bstride, int64_t n_el,
                                                            int HW, int C, double *__restrict__ sums, int want_sq) {
    __shared__ double sh[8];
    const int c = blockIdx.y;
    const int64_t e0 = (int64_t)blockIdx.x * kStatSpan;
    // fp64 from the first addition on: var = E[x^2] - mean^2 cancels when |mean| >> std (a few samples of a 2x2 map),
    // and fp32 partial sums of squares would leave 1e-7 * mean^2 / var of relative error in it
    double s = 0.0, q = 0.0;
    for (int k = 0; k < 16; ++k) {
        const int64_t e = e0 + k * 256 + threadIdx.x;
        if (e < n_el) {
            const int64_t b = e / HW;
            const int p = (int)(e - b * HW);
            const double v = (double)x[b * bstride + (int64_t)c * HW + p];
            s += v;
            q = fma(v, v, q);
        }
    }
    double ds = wave_sum_d(s), dq = wave_sum_d(q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = ds; sh[4 + w] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], sh[0] + sh[1] + sh[2] + sh[3]);
        if (want_sq) atomicAdd(&sums[C + c], sh[4] + sh[5] + sh[6] + sh[7]);
    }
}

// mean = sum / n, var = sumsq / n - mean^2 (biased) seen by autograd:  dx = (dmean + 2 dvar (x - mean)) / n
__global__ __launch_bounds__(256) void channel_stats_bwd_kernel(const float *__restrict__ x, int64_t bstride,
                                                                int64_t total, int C, int HW,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ dmean,
                                                                const float *__restrict__ dvar, float inv_n,
                                                                int accumulate, float *__restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int chw = C * HW;
    const int64_t b = i / chw;
    const int r = (int)(i - b * chw);
    const int c = r / HW;
    const float v = x[b * bstride + r];
    const float t = (dmean[c] + 2.f * dvar[c] * (v - mean[c])) * inv_n;
    dx[i] = accumulate ? dx[i] + t : t;
}

// nn.BatchNorm2d in training mode as the operand map of the convolution behind it: from the batch sums,
// mean = S1/n, var = S2/n - mean^2 (biased), rstd = (var + eps)^-1/2, pre = [gamma rstd | beta - mean gamma rstd],
// stat = [mean | rstd]; the running statistics move as torch's module moves them (unbiased variance).
__global__ __launch_bounds__(256) void bn2d_fold_train_kernel(const double *__restrict__ sums, double n, int C,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, float momentum,
                                                              float *__restrict__ running_mean,
                                                              float *__restrict__ running_var, float *__restrict__ pre,
                                                              float *__restrict__ stat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / n;
    double var = sums[C + c] / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float mf = (float)mean, vf = (float)var;
    const float rstd = 1.f / sqrtf(vf + eps);
    const float a = gamma[c] * rstd;
    pre[c] = a;
    pre[C + c] = beta[c] - mf * a;
    stat[c] = mf;
    stat[C + c] = rstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
    }
}

// gradients of (a, b) = (gamma rstd, beta - mean a) back to gamma, beta and the statistics:
// da' = da - mean db,  dgamma = da' rstd,  dbeta = db,  dmean = -a db,  dvar = -da' gamma rstd^3 / 2
__global__ __launch_bounds__(256) void bn2d_fold_bwd_kernel(const double *__restrict__ dab, int C,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ stat, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, float *__restrict__ dstat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mean = stat[c], rstd = stat[C + c], g = gamma[c];
    const double da = dab[c] - mean * dab[C + c], db = dab[C + c];
    dgamma[c] = (float)(da * rstd);
    dbeta[c] = (float)db;
    dstat[c] = (float)(-g * rstd * db);
    dstat[C + c] = (float)(-0.5 * da * g * rstd * rstd * rstd);
}

// out = a[c] x + b[c]
__global__ __launch_bounds__(256) void channel_affine_kernel(const float *__restrict__ x, int64_t total, int C, int HW,
                                                             const float *__restrict__ ab, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    out[i] = fmaf(ab[c], x[i], ab[C + c]);
}

// y = f(a[c] x + b[c]) * mask[p], f = relu or identity;  g = dy * mask * f'(.):  dx = g a, dab[c] += sum g x,
// dab[C + c] += sum g.  ab == null: y = x * mask (dx = dy * mask, no parameter sums).  grid (chunks, C)
__global__ __launch_bounds__(256) void channel_affine_bwd_kernel(const float *__restrict__ x, int64_t x_bstride,
                                                                 const float *__restrict__ dy, int64_t n_el, int HW,
                                                                 int C, const float *__restrict__ ab, int relu,
                                                                 const float *__restrict__ mask,
                                                                 float *__restrict__ dx, double *__restrict__ dab) {
    __shared__ double sh[8];
    const int c = blockIdx.y;
    const int64_t e0 = (int64_t)blockIdx.x * kStatSpan;
    const float a = ab ? ab[c] : 1.f, bb = ab ? ab[C + c] : 0.f;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < 16; ++k) {
        const int64_t e = e0 + k * 256 + threadIdx.x;
        if (e < n_el) {
            const int64_t b = e / HW;
            const int p = (int)(e - b * HW);
            const int64_t o = (b * C + c) * HW + p;
            const float v = x[b * x_bstride + (int64_t)c * HW + p];
            float g = dy[o];
            if (mask) g *= mask[p];
            if (relu && !(fmaf(a, v, bb) > 0.f)) g = 0.f;
            dx[o] = g * a;
            s = fmaf(g, v, s);
            q += g;
        }
    }
    if (!dab) return;
    double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = ds; sh[4 + w] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&dab[c], sh[0] + sh[1] + sh[2] + sh[3]);
        atomicAdd(&dab[C + c], sh[4] + sh[5] + sh[6] + sh[7]);
    }
}

// dw[co][ci][ky][kx] += sum_{b,p} dout[b,co,p] * h[b,ci,p + (ky-r, kx-r)],  h = mask * f(pre_a x + pre_b) (zero outside
// the image).  Work-group: input channel ci, COT (8 or 16) output channels, a slice of the batch; every thread walks its
// pixels over the samples of the slice with KS*KS*COT accumulators, then the group reduces and adds its partial to dw with fp32 atomics.
// (Measured and dropped in round 4: two input channels per 512-thread work-group so that the halves share their dout reads
// through L1 -- 356 -> 470 us per call.)
template <int KS, int COT>
__global__ __launch_bounds__(256) void conv2d_bwd_weight_kernel(const float *__restrict__ x, int64_t x_bstride,
                                                                const float *__restrict__ dout, int64_t B, int Cin,
                                                                int Cout, int H, int W,
                                                                const float *__restrict__ pre,
                                                                const float *__restrict__ mask, int b_per_group,
                                                                float *__restrict__ dw) {
    constexpr int T = KS * KS, R = KS / 2;
    __shared__ float sh[4][T * COT];
    const int tid = threadIdx.x;
    const int ci = blockIdx.x, co0 = blockIdx.y * COT;
    const int HW = H * W;
    const int64_t b0 = (int64_t)blockIdx.z * b_per_group;
    const int64_t b1 = b0 + b_per_group < B ? b0 + b_per_group : B;
    const float pa = pre ? pre[ci] : 1.f, pb = pre ? pre[Cin + ci] : 0.f;
    // accumulators as pairs of output channels: the inner product runs on v_pk_fma_f32 (two FMAs per lane and issue slot;
    // the scalar form reached 40 TFLOP/s, half of the vector ALU's unpacked fp32 rate)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(COT % 2 == 0, "output channels in pairs");
    f32x2 acc2[T][COT / 2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < COT / 2; ++j) acc2[t][j] = f32x2{0.f, 0.f};
    // a thread keeps its pixels (p = tid, tid + 256, ...) over the samples of the slice: tap offsets / validity per pixel
    // are computed once per pixel, not per sample
    for (int p = tid; p < HW; p += 256) {
        const int h = p / W, w = p - h * W;
        int off[T];
        float keep[T];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int hh = h + ky - R, ww = w + kx - R;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                const int o = ok ? hh * W + ww : p;
                off[ky * KS + kx] = o;
                keep[ky * KS + kx] = ok ? (mask ? mask[o] : 1.f) : 0.f;
            }
        for (int64_t b = b0; b < b1; ++b) {
            const float *xc = x + b * x_bstride + (int64_t)ci * HW;
            float hv[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float v = xc[off[t]];
                if (pre) v = fmaxf(fmaf(pa, v, pb), 0.f);
                hv[t] = v * keep[t];
            }
            const float *dc = dout + (b * Cout + co0) * HW + p;
#pragma unroll
            for (int j = 0; j < COT / 2; ++j) {
                const f32x2 d = {co0 + 2 * j < Cout ? dc[(int64_t)(2 * j) * HW] : 0.f,
                                 co0 + 2 * j + 1 < Cout ? dc[(int64_t)(2 * j + 1) * HW] : 0.f};
#pragma unroll
                for (int t = 0; t < T; ++t) acc2[t][j] = __builtin_elementwise_fma(d, f32x2{hv[t], hv[t]}, acc2[t][j]);
            }
        }
    }
    const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            const float v = wave_sum_f(acc2[t][j >> 1][j & 1]);
            if (ln == 0) sh[wv][t * COT + j] = v;
        }
    __syncthreads();
    if (tid < T * COT) {
        const int t = tid / COT, j = tid - t * COT;
        if (co0 + j < Cout) {
            const float v = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
            atomicAdd(&dw[((int64_t)(co0 + j) * Cin + ci) * T + t], v);
        }
    }
}

// Backward of coupling2d_kernel in the density direction (u = (x - t) exp(-s), ldj = -sum s; t = z_t m,
// s = scale[c] tanh(z_s) m):  dx = du e^{-s} (+ du copied through on the conditioning half),  dz_t = -du e^{-s} m,
// ds = -du (x - t) e^{-s} - dldj[b],  dz_s = ds scale[c] m (1 - tanh^2),  dscale[c] += ds tanh m.
__global__ __launch_bounds__(256) void coupling2d_bwd_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ inv_mask, int C, int HW,
                                                             int affine, int reverse, const float *__restrict__ dout,
                                                             const float *__restrict__ dldj, float *__restrict__ dx,
                                                             float *__restrict__ dz, double *__restrict__ dscale) {
    extern __shared__ float sc[];  // [Ch] partial scale gradients of this sample
    const int b = blockIdx.x;
    const bool chw = inv_mask == nullptr;
    const int Ch = chw ? C / 2 : C;
    const int n = Ch * HW;
    const int64_t zb = (int64_t)b * (affine ? 2 : 1) * n;
    const int off = chw ? (reverse ? n : 0) : 0;
    const float *xb = x + (int64_t)b * C * HW;
    const float *db = dout + (int64_t)b * C * HW;
    float *dxb = dx + (int64_t)b * C * HW;
    const float dl = dldj ? dldj[b] : 0.f;
    if (affine)
        for (int c = threadIdx.x; c < Ch; c += 256) sc[c] = 0.f;
    __syncthreads();
    // a thread walks consecutive pixels of a channel: i = c * HW + p with p strided by 256 inside the channel
    for (int c = 0; c < Ch; ++c) {
        float part = 0.f;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int i = c * HW + p;
            const float m = chw ? 1.f : inv_mask[p];
            const float t = z[zb + i] * m;
            const float du = db[off + i];
            if (affine) {
                const float th = tanhf(z[zb + n + i]);
                const float k = scale[c];
                const float s = k * th * m;
                const float e = expf(-s);
                const float ds = -du * (xb[off + i] - t) * e - dl;
                dxb[off + i] = du * e;
                dz[zb + i] = -du * e * m;
                dz[zb + n + i] = ds * k * m * (1.f - th * th);
                part = fmaf(ds, th * m, part);
            } else {
                dxb[off + i] = du;
                dz[zb + i] = -du * m;
            }
            if (chw) dxb[(n - off) + i] = db[(n - off) + i];
        }
        if (affine) {
            part = wave_sum_f(part);
            if ((threadIdx.x & 63) == 0) atomicAdd(&sc[c], part);
        }
    }
    if (affine) {
        __syncthreads();
        for (int c = threadIdx.x; c < Ch; c += 256) atomicAdd(&dscale[c], (double)sc[c]);
    }
}

}  // namespace dpk

using namespace dpk;

extern "C" {

int dpk_channel_stats(const float *x, int64_t x_bstride, int64_t B, int32_t C, int32_t H, int32_t W, int32_t want_sq,
                      double *sums, void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "channel_stats: bad sizes");
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX && C <= 65535, DPK_EUNSUPPORTED, "channel_stats: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && sums, DPK_EINVAL, "channel_stats: null pointer");
    DPK_REQUIRE(x_bstride >= (int64_t)C * H * W, DPK_EINVAL, "channel_stats: batch stride below C*H*W");
    const int64_t n_el = B * H * W;
    DPK_REQUIRE(n_el / kStatSpan < INT32_MAX, DPK_EUNSUPPORTED, "channel_stats: too large");
    DPK_LAUNCH(channel_stats_kernel, dim3((unsigned)cdiv(n_el, kStatSpan), (unsigned)C), dim3(256), 0,
               (hipStream_t)stream, x, x_bstride, n_el, H * W, C, sums, want_sq);
    DPK_CHECK_LAUNCH("channel_stats_kernel");
    return DPK_OK;
}

int dpk_channel_stats_backward(const float *x, int64_t x_bstride, int64_t B, int32_t C, int32_t H, int32_t W,
                               const float *mean, const float *dmean, const float *dvar, int32_t accumulate, float *dx,
                               void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "channel_stats_backward: bad sizes");
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX, DPK_EUNSUPPORTED, "channel_stats_backward: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && mean && dmean && dvar && dx, DPK_EINVAL, "channel_stats_backward: null pointer");
    const int64_t total = B * C * H * W;
    DPK_REQUIRE(total / 256 < INT32_MAX, DPK_EUNSUPPORTED, "channel_stats_backward: too large");
    DPK_LAUNCH(channel_stats_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
               x_bstride, total, C, H * W, mean, dmean, dvar, 1.f / (float)(B * H * W), accumulate, dx);
    DPK_CHECK_LAUNCH("channel_stats_bwd_kernel");
    return DPK_OK;
}

int dpk_bn2d_fold_train(const double *sums, int64_t n, int32_t C, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *pre, float *stat, void *stream) {
    DPK_REQUIRE(C > 0 && n > 0, DPK_EINVAL, "bn2d_fold_train: bad sizes");
    DPK_REQUIRE(sums && gamma && beta && pre && stat && (!running_mean == !running_var), DPK_EINVAL,
                "bn2d_fold_train: null pointer");
    DPK_LAUNCH(bn2d_fold_train_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, (double)n,
               C, gamma, beta, eps, momentum, running_mean, running_var, pre, stat);
    DPK_CHECK_LAUNCH("bn2d_fold_train_kernel");
    return DPK_OK;
}

int dpk_bn2d_fold_backward(const double *dab, int32_t C, const float *gamma, const float *stat, float *dgamma,
                           float *dbeta, float *dstat, void *stream) {
    DPK_REQUIRE(C > 0, DPK_EINVAL, "bn2d_fold_backward: bad sizes");
    DPK_REQUIRE(dab && gamma && stat && dgamma && dbeta && dstat, DPK_EINVAL, "bn2d_fold_backward: null pointer");
    DPK_LAUNCH(bn2d_fold_bwd_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dab, C, gamma, stat,
               dgamma, dbeta, dstat);
    DPK_CHECK_LAUNCH("bn2d_fold_bwd_kernel");
    return DPK_OK;
}

int dpk_channel_affine_forward(const float *x, int64_t B, int32_t C, int32_t H, int32_t W, const float *ab, float *out,
                               void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "channel_affine_forward: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && ab && out, DPK_EINVAL, "channel_affine_forward: null pointer");
    const int64_t total = B * C * H * W;
    DPK_REQUIRE(total / 256 < INT32_MAX, DPK_EUNSUPPORTED, "channel_affine_forward: too large");
    DPK_LAUNCH(channel_affine_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, total, C,
               H * W, ab, out);
    DPK_CHECK_LAUNCH("channel_affine_kernel");
    return DPK_OK;
}

int dpk_channel_affine_backward(const float *x, int64_t x_bstride, const float *dy, int64_t B, int32_t C, int32_t H,
                                int32_t W, const float *ab, int32_t relu, const float *mask, float *dx, double *dab,
                                void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "channel_affine_backward: bad sizes");
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX && C <= 65535, DPK_EUNSUPPORTED, "channel_affine_backward: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && dy && dx && (!ab || dab), DPK_EINVAL, "channel_affine_backward: null pointer");
    DPK_REQUIRE(x_bstride >= (int64_t)C * H * W, DPK_EINVAL, "channel_affine_backward: batch stride below C*H*W");
    const int64_t n_el = B * H * W;
    DPK_REQUIRE(n_el / kStatSpan < INT32_MAX, DPK_EUNSUPPORTED, "channel_affine_backward: too large");
    DPK_LAUNCH(channel_affine_bwd_kernel, dim3((unsigned)cdiv(n_el, kStatSpan), (unsigned)C), dim3(256), 0,
               (hipStream_t)stream, x, x_bstride, dy, n_el, H * W, C, ab, relu, mask, dx, ab ? dab : nullptr);
    DPK_CHECK_LAUNCH("channel_affine_bwd_kernel");
    return DPK_OK;
}

int dpk_conv2d_backward_weight(const float *in, int64_t in_bstride, const float *dout, int64_t B, int32_t Cin,
                               int32_t Cout, int32_t H, int32_t W, int32_t ks, const float *pre, const float *in_mask,
                               float *dw, void *stream) {
    DPK_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DPK_EINVAL, "conv2d_backward_weight: bad sizes");
    DPK_REQUIRE(ks == 1 || ks == 3, DPK_EUNSUPPORTED, "conv2d_backward_weight: kernel size %d (1 and 3 are built)", ks);
    DPK_REQUIRE((int64_t)Cin * H * W < INT32_MAX && (int64_t)Cout * H * W < INT32_MAX && Cout / 8 < 65535,
                DPK_EUNSUPPORTED, "conv2d_backward_weight: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && dout && dw, DPK_EINVAL, "conv2d_backward_weight: null pointer");
    DPK_REQUIRE(in_bstride >= (int64_t)Cin * H * W, DPK_EINVAL, "conv2d_backward_weight: batch stride below Cin*H*W");
    // 16 output channels per work-group when the layer has them (half the operand loads per FMA), else 8; enough
    // work-groups to fill 256 CUs a few times over, at least 2048 pixel-samples each
    const int cot = Cout >= 16 ? 16 : 8;
    const int groups = Cin * cdiv(Cout, cot);
    int64_t split = cdiv(2048, groups);
    const int64_t min_b = cdiv(2048, (int64_t)H * W);
    if (split > cdiv(B, min_b)) split = cdiv(B, min_b);
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    const int b_per_group = cdiv(B, split);
    const dim3 grid((unsigned)Cin, (unsigned)cdiv(Cout, cot), (unsigned)cdiv(B, b_per_group));
#define DPK_BW_CASE(KS_, COT_)                                                                                      \
    DPK_LAUNCH((conv2d_bwd_weight_kernel<KS_, COT_>), grid, dim3(256), 0, (hipStream_t)stream, in, in_bstride, dout, B, \
               Cin, Cout, H, W, pre, in_mask, b_per_group, dw)
    if (ks == 3) {
        if (cot == 16) DPK_BW_CASE(3, 16); else DPK_BW_CASE(3, 8);
    } else {
        if (cot == 16) DPK_BW_CASE(1, 16); else DPK_BW_CASE(1, 8);
    }
#undef DPK_BW_CASE
    DPK_CHECK_LAUNCH("conv2d_bwd_weight_kernel");
    return DPK_OK;
}

int dpk_coupling2d_transform_backward(const float *x, const float *z, const float *scale, const float *inv_mask,
                                      int64_t B, int32_t C, int32_t H, int32_t W, int32_t affine, int32_t reverse,
                                      const float *dout, const float *dldj, float *dx, float *dz, double *dscale,
                                      void *stream) {
    DPK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DPK_EINVAL, "coupling2d_transform_backward: bad sizes");
    DPK_REQUIRE(inv_mask || C % 2 == 0, DPK_EINVAL, "coupling2d_transform_backward: channel-wise coupling needs an even C=%d", C);
    DPK_REQUIRE((int64_t)C * H * W < INT32_MAX && B < INT32_MAX && C <= 8192, DPK_EUNSUPPORTED,
                "coupling2d_transform_backward: too large");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && z && dout && dx && dz && (!affine || (scale && dscale)), DPK_EINVAL,
                "coupling2d_transform_backward: null pointer");
    const int Ch = inv_mask ? C : C / 2;
    DPK_LAUNCH(coupling2d_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)Ch * sizeof(float), (hipStream_t)stream, x,
               z, scale, inv_mask, C, H * W, affine, reverse, dout, dldj, dx, dz, dscale);
    DPK_CHECK_LAUNCH("coupling2d_bwd_kernel");
    return DPK_OK;
}

}  // extern "C"
