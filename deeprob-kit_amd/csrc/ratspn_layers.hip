// Per-layer RAT-SPN operators for gfx950: ProductLayer, SumLayer / RootLayer (forward and
// backward) and the backward of the leaf layers.  These are the general-shape operators behind
// the individual reference modules; the inference fast path is the fused kernel in
// ratspn_fwd.hip.
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <math.h>

namespace dpk {

#define DPK_CONST __attribute__((address_space(4)))
typedef const DPK_CONST float *cfloat_p;
template <typename T> __host__ __device__ __forceinline__ const DPK_CONST T *as_const(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const DPK_CONST T *)p;
#pragma clang diagnostic pop
}

__global__ void softmax_rows_kernel2(const float *__restrict__ w, int rows, int n, float *__restrict__ W,
                                     float *__restrict__ LW, const unsigned *gate = nullptr) {
    if (gate_closed(gate)) return;   // (tables still match the live weights: common.h params_gate)
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *src = w + (int64_t)row * n;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, src[i]);
    m = wave_reduce_max(m);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(src[i] - m);
    s = wave_reduce_sum(s);
    const float ls = logf(s);
    for (int i = lane; i < n; i += 64) {
        const float l = src[i] - m - ls;
        LW[(int64_t)row * n + i] = l;
        W[(int64_t)row * n + i] = expf(l);
    }
}

// Few, long rows (a DGC-SPN root: one row of C*H*W = 8192 weights per class): one 1024-thread block per row instead of
// one wave -- the single-wave form takes 48 us for an 8192-entry row.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = is_max ? wave_reduce_max(v) : wave_reduce_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = is_max ? -INFINITY : 0.f;
    for (int w = 0; w < nw; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}
__global__ __launch_bounds__(1024) void softmax_rows_wide_kernel(const float *__restrict__ w, int rows, int n,
                                                                 float *__restrict__ W, float *__restrict__ LW,
                                                                 const unsigned *gate = nullptr) {
    __shared__ float red[16];
    if (gate_closed(gate)) return;
    const int row = blockIdx.x;
    const float *src = w + (int64_t)row * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, src[i]);
    m = block_reduce(m, true, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += expf(src[i] - m);
    s = block_reduce(s, false, red);
    const float ls = logf(s);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float l = src[i] - m - ls;
        LW[(int64_t)row * n + i] = l;
        W[(int64_t)row * n + i] = expf(l);
    }
}
__global__ __launch_bounds__(1024) void logsoftmax_jacobian_wide_kernel(const float *__restrict__ glw,
                                                                        const float *__restrict__ W, int rows, int n,
                                                                        float *__restrict__ gW) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += glw[(int64_t)row * n + i];
    s = block_reduce(s, false, red);
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        gW[(int64_t)row * n + i] = glw[(int64_t)row * n + i] - W[(int64_t)row * n + i] * s;
}
static void launch_softmax_rows(const float *weight, int rows, int n, float *W, float *LW, hipStream_t st,
                                const unsigned *gate = nullptr) {
    if (n >= 2048 && rows <= 1024)
        DPK_LAUNCH(softmax_rows_wide_kernel, dim3(rows), dim3(1024), 0, st, weight, rows, n, W, LW, gate);
    else
        DPK_LAUNCH(softmax_rows_kernel2, dim3(cdiv(rows, 4)), dim3(256), 0, st, weight, rows, n, W, LW, gate);
}

// ------------------------------------------------------------------------------------
// ProductLayer (reference: deeprob/spn/layers/ratspn.py:272-286)
// ------------------------------------------------------------------------------------
__global__ void product_fwd_kernel(const float *__restrict__ in, int64_t total, int R, int N,
                                   float *__restrict__ out) {
    const int NN = N * N, Ph = R / 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int ij = (int)(e % NN);
        const int64_t bp = e / NN;
        const int p = (int)(bp % Ph);
        const int64_t b = bp / Ph;
        const int i = ij / N, j = ij - i * N;
        const float *row = in + (b * R + 2 * p) * N;
        out[e] = row[i] + row[N + j];
    }
}

// gx[b,2p,i] = sum_j g[b,p,i,j];  gx[b,2p+1,j] = sum_i g[b,p,i,j]
__global__ void product_bwd_kernel(const float *__restrict__ g, int64_t total, int R, int N,
                                   float *__restrict__ gx) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % N);
        const int64_t br = e / N;
        const int r = (int)(br % R);
        const int64_t b = br / R;
        const float *gp = g + (b * (R / 2) + (r >> 1)) * N * N;
        float s = 0.f;
        if ((r & 1) == 0)
            for (int j = 0; j < N; ++j) s += gp[n * N + j];
        else
            for (int i = 0; i < N; ++i) s += gp[i * N + n];
        gx[e] = s;
    }
}

// ------------------------------------------------------------------------------------
// SumLayer / RootLayer forward (reference: ratspn.py:363-378, :446-458)
//   out[b,p,o] = logsumexp_n(x[b,p,n] + lw[p,o,n])
// One wave per (64 samples, partition).  x is staged through LDS in chunks of kNC columns
// (coalesced rows in, conflict-free column reads out); per sample the row maximum m is taken
// once and v_o = sum_n softmax(W)[o,n] * exp(x_n - m) accumulated for OB outputs at a time with
// the weights on the scalar path.  v_o < 1e-30 falls back to the exact two-pass form with the
// true maximum of x_n + lw_on (what torch.logsumexp computes).
// ------------------------------------------------------------------------------------
constexpr int kNC = 128;   // columns of x staged per pass
constexpr int kOG = 4;     // output groups per work-group (one wave each)
constexpr int kOB = 4;     // outputs accumulated at a time per thread

// Work-group = 4 waves = 64 samples of one partition.  Per 128-column chunk: the four waves stage the
// [64 x 128] slice of x into LDS (coalesced rows), then lane = sample, wave = output group: every wave reads
// the exponentials of its lane's row from LDS (row stride kNC+1: conflict-free) and accumulates its share of
// the S outputs, the softmaxed weights arriving on the scalar path (uniform per wave).  The row maxima come
// from a first pass over the same chunks (the second read of x hits L2).
__global__ __launch_bounds__(256) void sum_fwd_kernel(const float *__restrict__ in, cfloat_p W, cfloat_p LW,
                                                     int64_t B, int P, int N, int S,
                                                     float *__restrict__ out) {
    __shared__ float tile[64 * (kNC + 1)];
    __shared__ float rowmax[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: keeps the weight loads scalar
    const int p = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int64_t b = b0 + lane;
    const int64_t row_stride = (int64_t)P * N;
    const float *xin = in + (int64_t)p * N;

    // pass A: row maxima (wave w owns rows w, w+4, ...; lanes over the columns)
    for (int r = wave; r < 64; r += 4) {
        const float *src = xin + min(b0 + r, B - 1) * row_stride;
        float m = -INFINITY;
        for (int n = lane; n < N; n += 64) m = fmaxf(m, src[n]);
        m = wave_reduce_max(m);
        if (lane == 0) rowmax[r] = (m == -INFINITY) ? 0.f : m;
    }
    __syncthreads();
    const float m0 = rowmax[lane];

    // outputs of this wave: o = wave, wave + 4, ... (kOB at a time)
    const int n_own = (S > wave) ? (S - wave + kOG - 1) / kOG : 0;
    for (int ob = 0; ob < (S + kOG - 1) / kOG; ob += kOB) {   // uniform trip count: the barriers are inside
        float v[kOB];
#pragma unroll
        for (int q = 0; q < kOB; ++q) v[q] = 0.f;
        for (int n0 = 0; n0 < N; n0 += kNC) {
            __syncthreads();
            for (int r = wave; r < 64; r += 4) {
                const float *src = xin + min(b0 + r, B - 1) * row_stride + n0;
                const float mr = rowmax[r];
                for (int c = lane; c < kNC; c += 64)
                    tile[r * (kNC + 1) + c] = (n0 + c < N) ? __expf(src[c] - mr) : 0.f;
            }
            __syncthreads();
            const int nn = min(kNC, N - n0);
            if (ob < n_own) {
                int oq[kOB];
#pragma unroll
                for (int q = 0; q < kOB; ++q) oq[q] = min(wave + (ob + q) * kOG, S - 1);
                for (int c = 0; c < nn; ++c) {
                    const float e = tile[lane * (kNC + 1) + c];
#pragma unroll
                    for (int q = 0; q < kOB; ++q) v[q] = fmaf(W[((int64_t)p * S + oq[q]) * N + n0 + c], e, v[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kOB; ++q) {
            const int o = wave + (ob + q) * kOG;
            if (ob + q < n_own && b < B) {
                float r;
                if (v[q] < 1e-30f) {
                    const float *xr = xin + b * row_stride;
                    cfloat_p lw = LW + ((int64_t)p * S + o) * N;
                    float mm = -INFINITY;
                    for (int n = 0; n < N; ++n) mm = fmaxf(mm, xr[n] + lw[n]);
                    if (mm > -INFINITY) {
                        float s = 0.f;
                        for (int n = 0; n < N; ++n) s += expf(xr[n] + lw[n] - mm);
                        r = mm + logf(s);
                    } else {
                        r = -INFINITY;
                    }
                } else {
                    r = m0 + __logf(v[q]);
                }
                out[(b * P + p) * S + o] = r;
            }
        }
    }
}

// Wide single-partition rows (SpatialRootLayer: N = Cin*H*W in the thousands, few classes): one wave
// per sample, lanes stride the row (coalesced), per-lane online log-sum-exp in the log domain, then a
// wave combine.  The row is re-read once per class (it stays in L2).
__global__ __launch_bounds__(256) void root_wide_kernel(const float *__restrict__ in, const float *__restrict__ LW,
                                                        int64_t B, int N, int S, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *xr = in + b * N;
    for (int o = 0; o < S; ++o) {
        const float *lw = LW + (int64_t)o * N;
        float m = -INFINITY, s = 0.f;
        int n = lane;
        for (; n + 192 < N; n += 256) {
            const float t0 = xr[n] + lw[n], t1 = xr[n + 64] + lw[n + 64];
            const float t2 = xr[n + 128] + lw[n + 128], t3 = xr[n + 192] + lw[n + 192];
            const float cm = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
            if (cm > m) {
                s *= expf(m - cm);   // m == -inf: s == 0 stays 0 (exp(-inf) == 0)
                m = cm;
            }
            if (m > -INFINITY) s += (expf(t0 - m) + expf(t1 - m)) + (expf(t2 - m) + expf(t3 - m));
        }
        for (; n < N; n += 64) {
            const float t = xr[n] + lw[n];
            if (t > m) {
                s *= expf(m - t);
                m = t;
            }
            if (m > -INFINITY) s += expf(t - m);
        }
        const float M = wave_reduce_max(m);
        const float part = (m > -INFINITY) ? s * expf(m - M) : 0.f;
        const float tot = wave_reduce_sum(part);
        if (lane == 0) out[b * S + o] = (M > -INFINITY) ? M + logf(tot) : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------
// SumLayer / RootLayer backward.
//   pi[b,p,o,n] = exp(x[b,p,n] + lw[p,o,n] - out[b,p,o])
//   gx[b,p,n]   = sum_o g[b,p,o] pi ;  glw[p,o,n] = sum_b g[b,p,o] pi
//   gW          = glw - softmax(W) * sum_n glw            (log_softmax Jacobian)
// Lanes run over the (p,n) columns, so neither sum needs a cross-lane reduction: gx is a sum over
// o inside the thread and glw a sum over the samples of the tile, flushed with one atomic per
// (tile, column, o).
// ------------------------------------------------------------------------------------
// The stored output of a sum node, out = fl(logsumexp), carries half an ulp of its magnitude (3e-5 at 560, 6e-5 at 1100),
// and exp(x + lw - out) inherits it as a RELATIVE error common to all responsibilities of the node (the reference does not
// show it: its logsumexp backward normalises the very terms its forward summed).  corr[b,p,o] = log sum_n exp((x - out) + lw)
// -- the residual of the stored output against the sum the backward is about to form, a number of magnitude 1e-5 and
// therefore exact to 1e-12 as an fp32 -- is subtracted in the exponent: the responsibilities then sum to one.
// thread = one (b, p, o) row for short rows, a wave for long ones.
__global__ __launch_bounds__(256) void lse_residual_kernel(const float *__restrict__ x, const float *__restrict__ LW,
                                                          const float *__restrict__ out, int64_t B, int P, int N, int S,
                                                          float *__restrict__ corr) {
    const int64_t rows = B * P * S;
    if (N <= 32) {
        for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
            const int o = (int)(r % S);
            const int64_t bp = r / S;
            const int p = (int)(bp % P);
            const float xo = out[r];
            const float *xr = x + bp * N, *lw = LW + ((int64_t)p * S + o) * N;
            float sum = 0.f;
            for (int n = 0; n < N; ++n) sum += expf(fminf((xr[n] - xo) + lw[n], kRespArgMax));
            corr[r] = (xo > -INFINITY && sum > 0.f) ? logf(sum) : 0.f;
        }
    } else {
        const int lane = threadIdx.x & 63;
        for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
            const int o = (int)(r % S);
            const int64_t bp = r / S;
            const int p = (int)(bp % P);
            const float xo = out[r];
            const float *xr = x + bp * N, *lw = LW + ((int64_t)p * S + o) * N;
            float sum = 0.f;
            for (int n = lane; n < N; n += 64) sum += expf(fminf((xr[n] - xo) + lw[n], kRespArgMax));
            sum = wave_reduce_sum(sum);
            if (lane == 0) corr[r] = (xo > -INFINITY && sum > 0.f) ? logf(sum) : 0.f;
        }
    }
}

constexpr int kBwdTile = 32;   // samples per block in the sum backward (16 / 8 while the grid would not cover the chip)
constexpr int kBwdOB = 16;     // outputs held in registers at a time
constexpr int kBwdWaves = 4;

// block = (sample tile, 64 columns): its four waves take a quarter of the tile's samples each (a training batch of 512
// is 16 tiles: one wave per CU walking 32 samples in a chain of dependent loads was 55 us for the (8,8) sum layer) and
// meet in LDS before the one atomic per (tile, column, output).
__global__ __launch_bounds__(64 * kBwdWaves) void sum_bwd_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ LW,
                                                                const float *__restrict__ out,
                                                                const float *__restrict__ g, int64_t B, int P, int N,
                                                                int S, float *__restrict__ gx,
                                                                float *__restrict__ glw, int tile,
                                                                const float *__restrict__ corr) {
    __shared__ float red[kBwdWaves][kBwdOB][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + lane;  // (p, n) flattened
    const bool live = col < P * N;
    const int cc = live ? col : P * N - 1;
    const int p = cc / N, n = cc - p * N;
    const int per = tile / kBwdWaves;
    const int64_t b0 = (int64_t)blockIdx.x * tile + wave * per;
    const int64_t b1 = min(b0 + per, B);
    // samples outer, outputs inner: x and gx are touched once per sample, the per-output weights and the
    // batch sums of glw stay in registers (kBwdOB outputs at a time)
    for (int ob = 0; ob < S; ob += kBwdOB) {
        float lw[kBwdOB], acc[kBwdOB];
#pragma unroll
        for (int q = 0; q < kBwdOB; ++q) {
            lw[q] = (ob + q < S) ? LW[((int64_t)p * S + ob + q) * N + n] : 0.f;
            acc[q] = 0.f;
        }
        for (int64_t b = b0; b < b1; ++b) {
            const float xv = x[b * P * N + cc];
            const float *op = out + (b * P + p) * S + ob, *gp = g + (b * P + p) * S + ob;
            const float *cp = corr + (b * P + p) * S + ob;
            float tot = 0.f;
            // (outputs beyond S read the last valid one and are dropped by a select: `if (ob + q < S) { loads }` was a
            // dependent round trip per output and sample)
            float xo8[kBwdOB], g8[kBwdOB], c8[kBwdOB];
#pragma unroll
            for (int q = 0; q < kBwdOB; ++q) {
                const int qc = min(q, S - 1 - ob);
                xo8[q] = op[qc];
                g8[q] = gp[qc];
                c8[q] = cp[qc];
            }
#pragma unroll
            for (int q = 0; q < kBwdOB; ++q) {
                const float xo = xo8[q];
                // an all -inf row has out = -inf: its gradient is defined as zero (reference: the
                // masked_fill guard inside torch.logsumexp's backward gives the same)
                const float t = (ob + q < S && xo > -INFINITY) ? g8[q] * expf(fminf((xv - xo) + lw[q], kRespArgMax) - c8[q]) : 0.f;   // (large magnitudes cancel first, exactly; then the residual)
                acc[q] += t;
                tot += t;
            }
            if (gx != nullptr && live) {
                float *dst = gx + b * P * N + col;
                *dst = (ob == 0) ? tot : (*dst + tot);
            }
        }
        if (glw != nullptr) {
#pragma unroll
            for (int q = 0; q < kBwdOB; ++q) red[wave][q][lane] = acc[q];
            __syncthreads();
            // wave w flushes the outputs q = w, w + 4, ...
            for (int q = wave; q < kBwdOB && ob + q < S; q += kBwdWaves) {
                const float t = (red[0][q][lane] + red[1][q][lane]) + (red[2][q][lane] + red[3][q][lane]);
                if (live) atomicAdd(glw + ((int64_t)p * S + ob + q) * N + n, t);
            }
            __syncthreads();
        }
    }
}

__global__ void logsoftmax_jacobian_kernel(const float *__restrict__ glw, const float *__restrict__ W,
                                           int rows, int n, float *__restrict__ gW) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += glw[(int64_t)row * n + i];
    s = wave_reduce_sum(s);
    for (int i = lane; i < n; i += 64)
        gW[(int64_t)row * n + i] = glw[(int64_t)row * n + i] - W[(int64_t)row * n + i] * s;
}

// ------------------------------------------------------------------------------------
// Leaf backward.  Parameter-stationary: a thread owns one table entry (region r,
// variable j) for CBK channels and walks the samples of its tile, so the parameter gradients are
// plain register sums; upstream g[b,r,k] is wave-uniform (scalar path); x is gathered from the
// row.  Tiles meet through one atomic per (tile, entry, channel).
//   Gaussian: dmu = sum_b g m (x-mu)/s^2 ; dsigma = sum_b g m ((x-mu)^2/s^3 - 1/s)
//   Bernoulli: dlogit = sum_b g m (x - sigmoid(l))
// A second kernel, one thread per (sample, variable), forms d/dx = sum over the repetitions.
// ------------------------------------------------------------------------------------
constexpr int kLeafBwdTile = 64;

// block = (sample tile, region group, channel block of CBK): thread = one table entry, CBK channels in registers,
// samples outer so that x[b, f] is gathered once for the CBK channels
template <int DIST, int CBK>
__global__ __launch_bounds__(256) void leaf_bwd_param_kernel(
    const float *__restrict__ x, const float *__restrict__ g, int64_t B, int D, int R, int I, int d, int SP,
    const int *__restrict__ feat, const int *__restrict__ srcr, const float *__restrict__ p0,
    const float *__restrict__ p1, float *__restrict__ gp0, float *__restrict__ gp1, float drop_p,
    uint64_t seed, int tile) {
    const int grp = blockIdx.y;  // region group: its entry stream holds (variable, r*d+j) pairs
    const int kb = blockIdx.z * CBK;
    const int64_t b0 = (int64_t)blockIdx.x * tile;
    const int64_t b1 = min(b0 + tile, B);
    for (int e = threadIdx.x; e < SP; e += blockDim.x) {
        const int rj = srcr[(int64_t)grp * SP + e];
        if (rj < 0) continue;
        const int r = rj / d, j = rj - r * d;
        const int f = feat[(int64_t)grp * SP + e];
        float c0[CBK], c1[CBK], c2[CBK], a0[CBK], a1[CBK];
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            const int64_t po = ((int64_t)r * I + kb + k) * d + j;
            a0[k] = a1[k] = 0.f;
            if (DIST == 0) {
                const float sg = p1[po];
                c0[k] = p0[po];           // mu
                c1[k] = 1.f / (sg * sg);  // 1/s^2
                c2[k] = 1.f / sg;
            } else {
                c0[k] = 1.f / (1.f + expf(-p0[po]));  // sigmoid(logit)
                c1[k] = c2[k] = 0.f;
            }
        }
        // the loads of four samples are issued together (the loop is a chain of dependent latencies otherwise)
#pragma unroll 4
        for (int64_t b = b0; b < b1; ++b) {
            const float xr = x[b * D + f];
            const bool live = (xr == xr);   // marginalised: no contribution
            const float xv = live ? xr : 0.f;
            const float *gp = g + (b * R + r) * I + kb;
#pragma unroll
            for (int k = 0; k < CBK; ++k) {
                // training-mode input dropout: the same (seed, element) decision as the forward kernel
                if (drop_p > 0.f && dropout_hit(seed, (((uint64_t)b * R + r) * I + kb + k) * d + j, drop_p)) continue;
                const float gv = live ? gp[k] : 0.f;
                if (DIST == 0) {
                    const float dl = xv - c0[k];
                    a0[k] = fmaf(gv, dl * c1[k], a0[k]);
                    a1[k] = fmaf(gv, dl * dl * c1[k] * c2[k] - c2[k], a1[k]);
                } else {
                    a0[k] = fmaf(gv, xv - c0[k], a0[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            const int64_t po = ((int64_t)r * I + kb + k) * d + j;
            if (gp0) atomicAdd(gp0 + po, a0[k]);
            if (DIST == 0 && gp1) atomicAdd(gp1 + po, a1[k]);
        }
    }
}

// The same for small batches (16-sample tiles so that the grid covers the chip), with the tile's rows of x and of the
// upstream gradient staged in LDS first: a thread's x[b, f] is a gather along the row (f follows the region graph's
// permutation) and g[b, r, k] depends on the entry's region -- from global memory those were 5 uncoalesced loads per entry
// and sample, 53 us of the (8,8) training step at B = 512; staged, the rows arrive coalesced once per block and the
// gathers are LDS reads.
// A work-group walks `passes` consecutive 16-sample tiles with its entries' sums in registers (up to 4 entries per thread)
// and adds them to memory once: the atomics -- 32 per parameter at B = 512 with one tile per work-group, 1.6 M in all --
// were a third of the kernel.
template <int DIST, int CBK, bool WANT1>
__global__ __launch_bounds__(256) void leaf_bwd_param_lds_kernel(
    const float *__restrict__ x, const float *__restrict__ g, int64_t B, int D, int R, int I, int d, int SP,
    const int *__restrict__ feat, const int *__restrict__ srcr, const float *__restrict__ p0,
    const float *__restrict__ p1, float *__restrict__ gp0, float *__restrict__ gp1, int tile, int passes) {
    extern __shared__ float leaf_bwd_sm[];
    float *xs = leaf_bwd_sm, *gs = leaf_bwd_sm + (size_t)tile * D;   // xs[tile][D], gs[tile][R][CBK]
    constexpr int EMAX = 4;             // entries per thread (the host checks SP <= 4 * 256 before asking for passes > 1)
    const int grp = blockIdx.y;
    const int kb = blockIdx.z * CBK;
    // the thread's entries: (region, position), variable, parameters
    int rr[EMAX], ff[EMAX];
    int64_t po0[EMAX];
    float c0[EMAX][CBK], c1[EMAX][CBK], c2[EMAX][CBK], a0[EMAX][CBK], a1[EMAX][CBK];
#pragma unroll
    for (int q = 0; q < EMAX; ++q) {
        const int e = threadIdx.x + q * 256;
        const int rj = e < SP ? srcr[(int64_t)grp * SP + e] : -1;
        rr[q] = rj < 0 ? -1 : rj / d;
        const int j = rj < 0 ? 0 : rj - rr[q] * d;
        ff[q] = rj < 0 ? 0 : feat[(int64_t)grp * SP + e];
        po0[q] = ((int64_t)(rj < 0 ? 0 : rr[q]) * I + kb) * d + j;
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            a0[q][k] = a1[q][k] = 0.f;
            c0[q][k] = c1[q][k] = c2[q][k] = 0.f;
            if (rj >= 0) {
                const int64_t po = po0[q] + (int64_t)k * d;
                if (DIST == 0) {
                    const float sg = p1[po];
                    c0[q][k] = p0[po];           // mu
                    c1[q][k] = 1.f / (sg * sg);  // 1/s^2
                    c2[q][k] = 1.f / sg;
                } else {
                    c0[q][k] = 1.f / (1.f + expf(-p0[po]));  // sigmoid(logit)
                }
            }
        }
    }
    for (int pass = 0; pass < passes; ++pass) {
        const int64_t b0 = ((int64_t)blockIdx.x * passes + pass) * tile;
        if (b0 >= B) break;
        const int nb = (int)(min(b0 + tile, B) - b0);
        if (pass > 0) __syncthreads();
        {
            const float *xsrc = x + b0 * D;
            const int tot = nb * D;
            if ((D & 3) == 0) {
                for (int i = threadIdx.x * 4; i < tot; i += blockDim.x * 4)
                    *reinterpret_cast<float4 *>(xs + i) = *reinterpret_cast<const float4 *>(xsrc + i);
            } else {
                for (int i = threadIdx.x; i < tot; i += blockDim.x) xs[i] = xsrc[i];
            }
            const int gt = nb * R * CBK;
            for (int i = threadIdx.x; i < gt; i += blockDim.x) {
                const int bl = i / (R * CBK), rem = i - bl * (R * CBK);
                const int r = rem / CBK, k = rem - r * CBK;
                gs[i] = g[((b0 + bl) * R + r) * I + kb + k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            if (rr[q] < 0) continue;
#pragma unroll 4
            for (int bl = 0; bl < nb; ++bl) {
                const float xr = xs[bl * D + ff[q]];
                const bool live = (xr == xr);   // marginalised: no contribution
                const float xv = live ? xr : 0.f;
                const float *gp = gs + (bl * R + rr[q]) * CBK;
#pragma unroll
                for (int k = 0; k < CBK; ++k) {
                    const float gv = live ? gp[k] : 0.f;
                    if (DIST == 0) {
                        const float dl = xv - c0[q][k];
                        a0[q][k] = fmaf(gv, dl * c1[q][k], a0[q][k]);
                        if (WANT1) a1[q][k] = fmaf(gv, dl * dl * c1[q][k] * c2[q][k] - c2[q][k], a1[q][k]);
                    } else {
                        a0[q][k] = fmaf(gv, xv - c0[q][k], a0[q][k]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < EMAX; ++q) {
        if (rr[q] < 0) continue;
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            const int64_t po = po0[q] + (int64_t)k * d;
            if (gp0) atomicAdd(gp0 + po, a0[q][k]);
            if (DIST == 0 && WANT1 && gp1) atomicAdd(gp1 + po, a1[q][k]);
        }
    }
}

// Round 5: the same sums as three small GEMMs on the matrix cores.  With the upstream gradient of the live samples
// gl[b, (r,k)] = g[b,r,k] [x[b,f] observed],
//     S0[(r,k), f] = sum_b gl,   S1 = sum_b gl x[b,f],   S2 = sum_b gl x[b,f]^2      (G^T [R I x B] times [B x D] operands)
// and  d/dmu = (S1 - mu S0) / s^2,   d/ds = ((S2 - 2 mu S1 + mu^2 S0) / s^2 - S0) / s  (Bernoulli: d/dlogit = S1 - p S0):
// no gather along the row of x, no per-parameter atomics, no zeroing of the gradients.  A work-group owns 32 rows
// (32 / I regions) x 32 variables; four waves a quarter of the batch each on v_mfma_f32_32x32x2_f32 (exact fp32 products),
// summed through LDS; the epilogue keeps the (row, variable) pairs whose variable belongs to the row's region -- one in
// 2^depth -- through a map built from the region masks.  Measured at B = 512, (8,8): 26.9 us (staged VALU kernel) ->
// see DESIGN.md.  Batches up to kLeafMomentMaxB (the K loop is not split over work-groups).
constexpr int kLeafMomentMaxB = 2048;
constexpr int kLeafMomentChunk = 64;               // samples a wave stages at a time
constexpr int kLeafMomentStride = 36;              // floats between staged rows (32 + 4: the b128 writes of 8 rows spread over the banks)
constexpr int kLeafMomentExtra = (32 + 4 * 32 + 4 * 32 + 4) * 4;   // pivots, per-wave max |x - c| and sum |g|, the tile's verdict
constexpr int kLeafMomentLds = 4 * 2 * kLeafMomentChunk * kLeafMomentStride * 4 + 32 * 32 * 2 + kLeafMomentExtra;   // staging (reused for the partial sums) + map + extras
// Conditioning: raw moments lose d/ds = ((S2 - 2 mu S1 + mu^2 S0)/s^2 - S0)/s to cancellation once |mu| / s is
// large (1e-3 .. 1e-2 relative for mu = 3, s = 0.1 or uint8-range data).  Two measures:
//  (1) the moments are taken about a per-variable pivot c_f = midpoint of the means of the tile's rows that hold f
//      (x - c_f as the B operand, mu - c_f in the epilogue): unstandardised data / small learned scales are back at the
//      direct form's error (fp32 emulation: 8e-2 -> 1e-5);
//  (2) what a pivot shared by the rows cannot fix -- channels of a region sitting on clusters many scales apart -- is
//      DETECTED per entry from the moments themselves (the terms that cancel, N = |S2| + 2 |d| |S1| + d^2 |S0|, against the
//      result and the entry's natural scale s^2 sum|g|; for mixed-sign g the bound sum|g| (max|x - c| + |d|)^2 instead), and
//      a tile with such an entry evaluates its entries again in the direct form sum_b g ((x - mu)^2 / s^2 - 1) / s:
//      correctness never depends on the data being well scaled, the fast path is only as fast as it is.
constexpr float kLeafMomentTol = 2.5e-5f;          // accepted relative error of an entry from the cancellation (GRAD_TOL / 4)
template <int DIST, bool WANT1>
__global__ __launch_bounds__(256) void leaf_bwd_moment_kernel(
    const float *__restrict__ x, const float *__restrict__ g, int64_t B, int D, int R, int I, int d,
    const int64_t *__restrict__ mask, const uint8_t *__restrict__ pad, const float *__restrict__ p0,
    const float *__restrict__ p1, float *__restrict__ gp0, float *__restrict__ gp1) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int CH = kLeafMomentChunk, ST = kLeafMomentStride;
    extern __shared__ __attribute__((aligned(16))) float leaf_mom_sm[];
    static_assert(4 * 2 * CH * ST >= 4 * 3 * 32 * 33, "the partial sums reuse the staging area");
    float *stage = leaf_mom_sm;                                                  // [wave][g, x][CH][ST]
    float (*part)[3][32 * 33] = reinterpret_cast<float (*)[3][32 * 33]>(leaf_mom_sm);   // [wave][S0, S1, S2][32 x 33]
    short (*jmap)[32] = reinterpret_cast<short (*)[32]>(leaf_mom_sm + 4 * 2 * CH * ST);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const int RI = R * I;
    const int nreg = 32 / I, r0 = row0 / I;            // regions of this row tile (32 % I == 0, R I % 32 == 0: the host checks)
    // ---- which variable of the column tile sits where in the tile's regions -------------------------------------------
    for (int e = tid; e < 32 * 32; e += 256) jmap[e >> 5][e & 31] = -1;
    __syncthreads();
    for (int e = tid; e < nreg * d; e += 256) {
        const int rl = e / d, j = e - rl * d;
        const int64_t o = (int64_t)(r0 + rl) * d + j;
        if (pad != nullptr && pad[o]) continue;
        const int f = (int)mask[o];
        if (f >= f0 && f < f0 + 32) jmap[rl][f - f0] = (short)j;
    }
    float *piv = reinterpret_cast<float *>(reinterpret_cast<char *>(jmap) + 32 * 32 * 2);   // [32]
    float (*devw)[32] = reinterpret_cast<float (*)[32]>(piv + 32);                            // [wave][column]
    float (*gabw)[32] = reinterpret_cast<float (*)[32]>(piv + 32 + 4 * 32);                   // [wave][row]
    int *redo = reinterpret_cast<int *>(piv + 32 + 8 * 32);
    __syncthreads();
    {
        // the pivot of a column: midpoint of the means of the rows that hold it.  thread = (column, one of 8 row groups): four
        // rows each, their loads unconditional at a clamped position (a load behind a branch is a round trip of its own) and
        // all in flight together; the groups meet in LDS (the staging area is not in use yet)
        float (*pmm)[2][32] = reinterpret_cast<float (*)[2][32]>(stage);      // [row group][min, max][column]
        const int col = tid & 31, part = tid >> 5;
        float lo = INFINITY, hi = -INFINITY;
        if (DIST == 0) {
            float mu[4];
            int jj[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = part + 8 * q, rl = i / I, k = i - rl * I;
                jj[q] = jmap[rl][col];
                mu[q] = p0[((int64_t)(r0 + rl) * I + k) * d + max(jj[q], 0)];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo = jj[q] >= 0 ? fminf(lo, mu[q]) : lo;
                hi = jj[q] >= 0 ? fmaxf(hi, mu[q]) : hi;
            }
        }
        pmm[part][0][col] = lo;
        pmm[part][1][col] = hi;
        __syncthreads();
        if (tid < 32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                lo = fminf(lo, pmm[q][0][tid]);
                hi = fmaxf(hi, pmm[q][1][tid]);
            }
            piv[tid] = hi >= lo ? 0.5f * (lo + hi) : 0.f;
        }
    }
    if (tid == 0) *redo = 0;
    __syncthreads();
    // ---- the wave's quarter of the batch, staged CH samples at a time: 16-byte loads (a compute unit's request path
    // takes ~37 cycles per load INSTRUCTION: one 4-byte load per operand and K-step was 512 of them, 23 us) ----------------
    const int m = lane & 31, kh = lane >> 5;
    const int64_t per = (B + 3) / 4;
    const int64_t bq0 = wave * per, bq1 = min(B, bq0 + per);
    float *gst = stage + (size_t)wave * 2 * CH * ST, *xst = gst + CH * ST;
    const int lr = lane >> 3, lc = (lane & 7) * 4;      // the lane's row (of 8 per instruction) and columns of a staged piece
    const bool xvec = (D % 4) == 0 && f0 + 32 <= D && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    f32x16 a0, a1, a2;
#pragma unroll
    for (int i = 0; i < 16; ++i) a0[i] = a1[i] = a2[i] = 0.f;
    const float cm = piv[m];        // pivot of the lane's column
    float dev = 0.f, gab = 0.f;     // max |x - c| of the lane's column, sum |g| of the lane's row (this K half)
    for (int64_t b = bq0; b < bq1; b += CH) {
        f32x4 gq[CH / 8], xq[CH / 8];
#pragma unroll
        for (int it = 0; it < CH / 8; ++it) {
            const int64_t bb = b + it * 8 + lr;
            const int64_t bc = min(bb, B - 1);
            gq[it] = *reinterpret_cast<const f32x4 *>(g + bc * RI + row0 + lc);          // (R I % 32 == 0: 16-byte aligned)
            if (xvec) {
                xq[it] = *reinterpret_cast<const f32x4 *>(x + bc * D + f0 + lc);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) xq[it][q] = x[bc * D + min(f0 + lc + q, D - 1)];
            }
            if (bb >= bq1) gq[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < CH / 8; ++it) {
            *reinterpret_cast<f32x4 *>(gst + (it * 8 + lr) * ST + lc) = gq[it];
            *reinterpret_cast<f32x4 *>(xst + (it * 8 + lr) * ST + lc) = xq[it];
        }
        // (the wave reads what it wrote itself: no barrier)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 8
        for (int u = 0; u < CH / 2; ++u) {
            const float gv = gst[(2 * u + kh) * ST + m];
            const float xr = xst[(2 * u + kh) * ST + m];
            const bool live = xr == xr;                 // marginalised: no contribution
            const float xl = live ? xr - cm : 0.f;
            dev = fmaxf(dev, fabsf(xl));
            gab += fabsf(gv);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, live ? 1.f : 0.f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xl, a1, 0, 0, 0);
            if (DIST == 0 && WANT1) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xl * xl, a2, 0, 0, 0);
        }
    }
    __syncthreads();   // (every wave is done with its staging area: the partial sums take its place)
    dev = fmaxf(dev, __shfl_xor(dev, 32, 64));
    gab += __shfl_xor(gab, 32, 64);
    if (lane < 32) {
        devw[wave][m] = dev;
        gabw[wave][m] = gab;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int i = (v >> 2) * 8 + kh * 4 + (v & 3);
        part[wave][0][i * 33 + m] = a0[v];
        part[wave][1][i * 33 + m] = a1[v];
        if (DIST == 0 && WANT1) part[wave][2][i * 33 + m] = a2[v];
    }
    __syncthreads();
    // ---- epilogue: thread = (row i, 4 variables) ---------------------------------------------------------------------------
    {
        const int i = tid >> 3, c0 = (tid & 7) * 4;
        const int rl = i / I, k = i - rl * I;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = c0 + cc;
            const int j = jmap[rl][c];
            if (j < 0) continue;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s0 += part[w][0][i * 33 + c];
                s1 += part[w][1][i * 33 + c];
                if (DIST == 0 && WANT1) s2 += part[w][2][i * 33 + c];
            }
            const int64_t po = ((int64_t)(r0 + rl) * I + k) * d + j;
            if (DIST == 0) {
                const float mu = p0[po], sg = p1[po];
                const float is2 = 1.f / (sg * sg);
                const float dl = mu - piv[c], adl = fabsf(dl);
                const float core1 = s1 - dl * s0;
                const float core2 = fmaf(dl, fmaf(dl, s0, -2.f * s1), s2);
                if (gp0) gp0[po] = core1 * is2;
                if (WANT1 && gp1) gp1[po] = (core2 * is2 - s0) / sg;
                // ---- did the pivot form cancel?  (header: measure (2)) ----
                const float gabs = (gabw[0][i] + gabw[1][i]) + (gabw[2][i] + gabw[3][i]);
                const float span = fmaxf(fmaxf(devw[0][c], devw[1][c]), fmaxf(devw[2][c], devw[3][c])) + adl;
                const bool mixed = gabs > fabsf(s0) * 1.001f;          // g of both signs (or marginalised samples): the signed moments under-count
                const float n1 = mixed ? gabs * span : fabsf(s1) + adl * fabsf(s0);
                const float n2 = mixed ? gabs * span * span : fabsf(s2) + 2.f * adl * fabsf(s1) + dl * dl * fabsf(s0);
                const float kap = fmaxf(4.f, (float)B * (1.f / 128.f)) * 5.96e-8f;   // accumulated rounding of a quarter-batch chain
                bool bad = gp0 && kap * n1 > kLeafMomentTol * fmaxf(fabsf(core1), sg * gabs);
                if (WANT1 && gp1) bad = bad || kap * n2 > kLeafMomentTol * fmaxf(fabsf(core2), sg * sg * gabs);
                if (bad) *redo = 1;
            } else {
                if (gp0) gp0[po] = s1 - s0 / (1.f + expf(-p0[po]));
            }
        }
    }
    if (DIST != 0) return;
    __syncthreads();
    if (*redo == 0) return;
    // ---- a tile with an ill-conditioned entry: its entries once more in the direct form (rare; x and g are L2 hits) ------
    {
        const int i = tid >> 3, c0 = (tid & 7) * 4;
        const int rl = i / I, k = i - rl * I;
        float mu[4], is2[4], sg[4], acc0[4], acc1[4];
        int64_t po[4];
        bool any = false;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int j = jmap[rl][c0 + cc];
            po[cc] = j >= 0 ? ((int64_t)(r0 + rl) * I + k) * d + j : -1;
            any = any || j >= 0;
            const int64_t pc = j >= 0 ? po[cc] : 0;
            mu[cc] = p0[pc];
            sg[cc] = p1[pc];
            is2[cc] = 1.f / (sg[cc] * sg[cc]);
            acc0[cc] = acc1[cc] = 0.f;
        }
        if (!any) return;
        const float *gcol = g + row0 + i;
        for (int64_t b = 0; b < B; ++b) {
            const float gv = gcol[b * RI];
            float xv[4];
            if (xvec) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(x + b * D + f0 + c0);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) xv[cc] = q[cc];
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) xv[cc] = x[b * D + min(f0 + c0 + cc, D - 1)];
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const bool live = xv[cc] == xv[cc];
                const float dlt = live ? xv[cc] - mu[cc] : 0.f;
                const float gl = live ? gv : 0.f;
                acc0[cc] = fmaf(gl, dlt, acc0[cc]);
                acc1[cc] = fmaf(gl, fmaf(dlt * dlt, is2[cc], -1.f), acc1[cc]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            if (po[cc] < 0) continue;
            if (gp0) gp0[po[cc]] = acc0[cc] * is2[cc];
            if (WANT1 && gp1) gp1[po[cc]] = acc1[cc] / sg[cc];
        }
    }
}

// inverse structure: for repetition rho and variable f, the (region, position) r*d+j holding it
__global__ void leaf_inverse_kernel(const int *__restrict__ feat, const int *__restrict__ srcr, int d, int SP,
                                    int D, int regions_per_rep, int *__restrict__ inv) {
    const int grp = blockIdx.x;
    for (int e = threadIdx.x; e < SP; e += blockDim.x) {
        const int rj = srcr[(int64_t)grp * SP + e];
        if (rj >= 0) {
            const int rho = (rj / d) / regions_per_rep;
            inv[(int64_t)rho * D + feat[(int64_t)grp * SP + e]] = rj;
        }
    }
}

__global__ __launch_bounds__(256) void gaussian_leaf_bwd_x_kernel(
    const float *__restrict__ x, const float *__restrict__ g, int64_t B, int D, int R, int I, int d, int reps,
    const int *__restrict__ inv, const float *__restrict__ loc, const float *__restrict__ scale,
    float *__restrict__ gx, float drop_p, uint64_t seed, int dist) {
    const int64_t total = B * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(e % D);
        const int64_t b = e / D;
        const float xv = x[e];
        float acc = 0.f;
        if (xv == xv) {
            for (int rho = 0; rho < reps; ++rho) {
                const int rj = inv[(int64_t)rho * D + f];
                if (rj < 0) continue;
                const int r = rj / d, j = rj - r * d;
                for (int k = 0; k < I; ++k) {
                    if (drop_p > 0.f && dropout_hit(seed, (((uint64_t)b * R + r) * I + k) * d + j, drop_p)) continue;
                    const int64_t po = ((int64_t)r * I + k) * d + j;
                    if (dist == 0) {
                        const float sg = scale[po];
                        acc = fmaf(g[(b * R + r) * I + k], -(xv - loc[po]) / (sg * sg), acc);
                    } else {   // Bernoulli: d/dx (x l - softplus(l)) = l   (ratspn.py:243, -BCEWithLogits)
                        acc = fmaf(g[(b * R + r) * I + k], loc[po], acc);
                    }
                }
            }
        }
        gx[e] = acc;
    }
}

// structure kernels live in ratspn_fwd.hip
int prepare_leaf_structure(const RatWs &w, const int64_t *mask, const uint8_t *pad, int R, int d, uint32_t flags,
                           hipStream_t st);

}  // namespace dpk

using namespace dpk;

static int grid_for(int64_t total, int block, int cap = 8192) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int dpk_product_forward(const float *in, int64_t B, int32_t R, int32_t N, float *out, void *stream) {
    DPK_REQUIRE(in && out, DPK_EINVAL, "product_forward: null pointer");
    DPK_REQUIRE(B >= 0 && R > 0 && (R % 2) == 0 && N > 0, DPK_EINVAL, "product_forward: bad sizes");
    const int64_t total = B * (R / 2) * N * N;
    if (total == 0) return DPK_OK;
    DPK_LAUNCH(product_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       total, R, N, out);
    DPK_CHECK_LAUNCH("product_fwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_product_backward(const float *g, int64_t B, int32_t R, int32_t N, float *grad_in,
                                    void *stream) {
    DPK_REQUIRE(g && grad_in, DPK_EINVAL, "product_backward: null pointer");
    DPK_REQUIRE(B >= 0 && R > 0 && (R % 2) == 0 && N > 0, DPK_EINVAL, "product_backward: bad sizes");
    const int64_t total = B * R * N;
    if (total == 0) return DPK_OK;
    DPK_LAUNCH(product_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       total, R, N, grad_in);
    DPK_CHECK_LAUNCH("product_bwd_kernel");
    return DPK_OK;
}

// workspace of the sum / root operators: W, LW, glw  (each P*S*N floats)
extern "C" int64_t dpk_sum_workspace_bytes(int64_t B, int32_t P, int32_t N, int32_t S) {
    if (P <= 0 || N <= 0 || S <= 0) return DPK_EINVAL;
    // (+ the backward's per-output normalisation residuals, [B, P, S] floats: lse_residual_kernel)
    return 3 * align_up((int64_t)P * S * N * 4, 256) + align_up(std::max<int64_t>(B, 0) * P * S * 4, 256);
}

static int sum_forward_impl(const float *in, const float *weight, int64_t B, int P, int N, int S, float *out,
                            void *ws, int64_t ws_bytes, void *stream, const char *who) {
    DPK_REQUIRE(B >= 0 && P > 0 && N > 0 && S > 0, DPK_EINVAL, "%s: bad sizes", who);
    DPK_REQUIRE(weight && ws && (B == 0 || (in && out)), DPK_EINVAL, "%s: null pointer", who);
    DPK_REQUIRE(P <= 65535, DPK_EUNSUPPORTED, "%s: partitions=%d > 65535", who, P);
    const int64_t seg = align_up((int64_t)P * S * N * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "%s: workspace too small", who);
    if (B == 0) return DPK_OK;
    float *W = (float *)ws, *LW = (float *)((char *)ws + seg);
    hipStream_t st = (hipStream_t)stream;
    launch_softmax_rows(weight, P * S, N, W, LW, st);
    if (P == 1 && N >= 1024)
        DPK_LAUNCH(root_wide_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, in, LW, B, N, S, out);
    else
        DPK_LAUNCH(sum_fwd_kernel, dim3(cdiv(B, 64), P), dim3(256), 0, st, in, as_const(W), as_const(LW), B,
                           P, N, S, out);
    DPK_CHECK_LAUNCH("sum_fwd_kernel");
    return DPK_OK;
}

static int sum_backward_impl(const float *in, const float *weight, const float *out, const float *g, int64_t B,
                             int P, int N, int S, float *grad_in, float *grad_weight, void *ws,
                             int64_t ws_bytes, void *stream, const char *who) {
    DPK_REQUIRE(B >= 0 && P > 0 && N > 0 && S > 0, DPK_EINVAL, "%s: bad sizes", who);
    DPK_REQUIRE(weight && ws && (B == 0 || (in && out && g)), DPK_EINVAL, "%s: null pointer", who);
    const int64_t seg = align_up((int64_t)P * S * N * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg + align_up(B * P * S * 4, 256), DPK_EWORKSPACE, "%s: workspace too small", who);
    float *W = (float *)ws, *LW = (float *)((char *)ws + seg), *glw = (float *)((char *)ws + 2 * seg);
    float *corr = (float *)((char *)ws + 3 * seg);
    hipStream_t st = (hipStream_t)stream;
    launch_softmax_rows(weight, P * S, N, W, LW, st);
    if (B > 0) {
        const int64_t rows = B * P * S;
        const int64_t blocks = N <= 32 ? (rows + 255) / 256 : (rows + 3) / 4;
        DPK_LAUNCH(lse_residual_kernel, dim3((unsigned)std::min<int64_t>(blocks, 65535 * 4)), dim3(256), 0, st, in, LW, out, B, P, N,
                   S, corr);
    }
    if (grad_weight) {
        hipError_t e = hipMemsetAsync(glw, 0, (size_t)P * S * N * 4, st);
        DPK_REQUIRE(e == hipSuccess, DPK_ELAUNCH, "%s: memset: %s", who, hipGetErrorString(e));
    }
    if (B > 0) {
        const int cols = P * N;
        int tile = kBwdTile;   // (shorter sample slices while the grid would leave compute units idle)
        while (tile > 8 && cdiv(B, tile) * cdiv(cols, 64) < 2 * device_cus()) tile /= 2;
        DPK_LAUNCH(sum_bwd_kernel, dim3(cdiv(B, tile), cdiv(cols, 64)), dim3(64 * kBwdWaves), 0, st, in, LW,
                           out, g, B, P, N, S, grad_in, grad_weight ? glw : nullptr, tile, corr);
    } else if (grad_in) {
        // nothing to write
    }
    if (grad_weight)
    {
        if (N >= 2048 && P * S <= 1024)
            DPK_LAUNCH(logsoftmax_jacobian_wide_kernel, dim3(P * S), dim3(1024), 0, st, glw, W, P * S, N, grad_weight);
        else
            DPK_LAUNCH(logsoftmax_jacobian_kernel, dim3(cdiv(P * S, 4)), dim3(256), 0, st, glw, W, P * S, N,
                       grad_weight);
    }
    DPK_CHECK_LAUNCH("sum_bwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_sum_forward(const float *in, const float *weight, int64_t B, int32_t P, int32_t N, int32_t S,
                               float *out, void *ws, int64_t ws_bytes, void *stream) {
    return sum_forward_impl(in, weight, B, P, N, S, out, ws, ws_bytes, stream, "sum_forward");
}
extern "C" int dpk_sum_backward(const float *in, const float *weight, const float *out, const float *g,
                                int64_t B, int32_t P, int32_t N, int32_t S, float *grad_in,
                                float *grad_weight, void *ws, int64_t ws_bytes, void *stream) {
    return sum_backward_impl(in, weight, out, g, B, P, N, S, grad_in, grad_weight, ws, ws_bytes, stream,
                             "sum_backward");
}
extern "C" int dpk_root_forward(const float *in, const float *weight, int64_t B, int32_t M, int32_t C,
                                float *out, void *ws, int64_t ws_bytes, void *stream) {
    return sum_forward_impl(in, weight, B, 1, M, C, out, ws, ws_bytes, stream, "root_forward");
}
extern "C" int dpk_root_backward(const float *in, const float *weight, const float *out, const float *g,
                                 int64_t B, int32_t M, int32_t C, float *grad_in, float *grad_weight, void *ws,
                                 int64_t ws_bytes, void *stream) {
    return sum_backward_impl(in, weight, out, g, B, 1, M, C, grad_in, grad_weight, ws, ws_bytes, stream,
                             "root_backward");
}

static int leaf_backward_common(int dist, const float *x, const float *g, int64_t B, int32_t D,
                                const int64_t *mask, const uint8_t *pad_mask, const float *p0, const float *p1,
                                int32_t R, int32_t I, int32_t d, float *gp0, float *gp1, float *gx, void *ws,
                                int64_t ws_bytes, uint32_t flags, void *stream, float drop_p = 0.f,
                                uint64_t seed = 0) {
    DPK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DPK_EINVAL, "leaf_backward: dropout rate must be in [0, 1)");
    DPK_REQUIRE(x && g && mask && p0 && ws, DPK_EINVAL, "leaf_backward: null pointer");
    DPK_REQUIRE(dist == 1 || p1, DPK_EINVAL, "leaf_backward: null scale");
    DPK_REQUIRE(B >= 0 && D > 0 && R > 0 && I > 0 && d > 0, DPK_EINVAL, "leaf_backward: bad sizes");
    DPK_REQUIRE(R <= 65535, DPK_EUNSUPPORTED, "leaf_backward: regions=%d > 65535", R);
    DPK_REQUIRE((R % 2) == 0, DPK_EINVAL, "leaf_backward: odd number of regions");
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, leaf_group(R), 0, 0, 0, 0);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "leaf_backward: workspace %lld < %lld", (long long)ws_bytes,
                (long long)w.bytes);
    hipStream_t st = (hipStream_t)stream;
    int rc = prepare_leaf_structure(w, mask, pad_mask, R, d, flags, st);
    if (rc) return rc;
    const size_t pbytes = (size_t)R * I * d * 4;
    // small batches: the parameter gradients as moment GEMMs (leaf_bwd_moment_kernel) -- every entry that has a variable
    // is written exactly once, so only a padded model's gradients are zeroed first
    static const bool moment_off = [] { const char *e = getenv("DPK_LEAF_MOMENT"); return e && atoi(e) == 0; }();
    const bool moment = !moment_off && B > 0 && B <= kLeafMomentMaxB && drop_p == 0.f && I <= 32 && (32 % I) == 0 &&
                        ((R * I) % 32) == 0 && d <= 32767 && (gp0 || gp1) &&
                        (reinterpret_cast<uintptr_t>(g) & 15) == 0;
    if (!moment || pad_mask != nullptr) {
        if (gp0) DPK_REQUIRE(hipMemsetAsync(gp0, 0, pbytes, st) == hipSuccess, DPK_ELAUNCH, "leaf_backward: memset");
        if (gp1) DPK_REQUIRE(hipMemsetAsync(gp1, 0, pbytes, st) == hipSuccess, DPK_ELAUNCH, "leaf_backward: memset");
    }
    if (moment) {
        const dim3 mgrid(R * I / 32, cdiv(D, 32)), mblock(256);
#define DPK_LEAF_MOMENT(DIST, W1)                                                                                          \
    do {                                                                                                                   \
        if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&leaf_bwd_moment_kernel<DIST, W1>), 96 * 1024)) return lrc; \
        DPK_LAUNCH((leaf_bwd_moment_kernel<DIST, W1>), mgrid, mblock, kLeafMomentLds, st, x, g, B, D, R, I, d, mask, pad_mask, p0, \
                   p1, gp0, gp1);                                                                                          \
    } while (0)
        if (dist == 0 && gp1) DPK_LEAF_MOMENT(0, true);
        else if (dist == 0) DPK_LEAF_MOMENT(0, false);
        else DPK_LEAF_MOMENT(1, false);
#undef DPK_LEAF_MOMENT
        DPK_CHECK_LAUNCH("leaf_bwd_moment_kernel");
    } else if (B > 0 && (gp0 || gp1)) {
        const int cbk = (I % 4 == 0) ? 4 : ((I % 2 == 0) ? 2 : 1);
        // small batches: shorter sample slices so that the grid still covers the chip (more atomics per parameter)
        const int tile = (B > 1024) ? kLeafBwdTile : 16;
        const dim3 grid(cdiv(B, tile), w.G, I / cbk), block(256);
        // (no dropout, rows 16-byte aligned, the tile's rows within 64 KB of LDS: the staged kernel)
        const size_t stage_bytes = (size_t)tile * ((size_t)D + (size_t)R * cbk) * 4;
        const bool staged = tile == 16 && drop_p == 0.f && stage_bytes <= 80 * 1024 && w.SP <= 4 * 256 &&
                            ((D & 3) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) == 0);
        // tiles per work-group of the staged kernel: as many as keep two work-groups per compute unit busy
        int passes = 1;
        if (staged)
            while (passes < 8 && cdiv(B, tile * passes * 2) * w.G * (I / cbk) >= 2 * device_cus()) passes *= 2;   // (B = 512: one pass; two measured 32 vs 27 us)
        const dim3 sgrid(cdiv(B, tile * passes), w.G, I / cbk);
#define DPK_LEAF_BWD(DIST, CBK)                                                                                  \
    do {                                                                                                         \
        if (staged && gp1) {                                                                                     \
            if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&leaf_bwd_param_lds_kernel<DIST, CBK, true>), 96 * 1024)) return lrc; \
            DPK_LAUNCH((leaf_bwd_param_lds_kernel<DIST, CBK, true>), sgrid, block, stage_bytes, st, x, g, B, D, R, I, d, \
                       w.SP, w.feat, w.srcr, p0, p1, gp0, gp1, tile, passes);                                    \
        } else if (staged) {                                                                                     \
            if (int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&leaf_bwd_param_lds_kernel<DIST, CBK, false>), 96 * 1024)) return lrc; \
            DPK_LAUNCH((leaf_bwd_param_lds_kernel<DIST, CBK, false>), sgrid, block, stage_bytes, st, x, g, B, D, R, I, d, \
                       w.SP, w.feat, w.srcr, p0, p1, gp0, gp1, tile, passes);                                    \
        } else                                                                                                     \
            DPK_LAUNCH((leaf_bwd_param_kernel<DIST, CBK>), grid, block, 0, st, x, g, B, D, R, I, d, w.SP, w.feat, \
                       w.srcr, p0, p1, gp0, gp1, drop_p, seed, tile);                                            \
    } while (0)
        if (dist == 0) {
            if (cbk == 4) DPK_LEAF_BWD(0, 4);
            else if (cbk == 2) DPK_LEAF_BWD(0, 2);
            else DPK_LEAF_BWD(0, 1);
        } else {
            if (cbk == 4) DPK_LEAF_BWD(1, 4);
            else if (cbk == 2) DPK_LEAF_BWD(1, 2);
            else DPK_LEAF_BWD(1, 1);
        }
#undef DPK_LEAF_BWD
        DPK_CHECK_LAUNCH("leaf_bwd_param_kernel");
    }
    if (gx && B > 0) {
        // regions per repetition = 2^depth = (D + pad) / d with pad < 2^depth <= D (RegionGraph: depth <= log2 D), i.e.
        // the smallest power of two p with p * d >= D -- NOT ceil(D / d), which is smaller whenever pad >= d
        // (D = 9, depth 3: d = 2, 8 regions, ceil(9/2) = 5)
        int per_rep = 1;
        while ((int64_t)per_rep * d < D) per_rep *= 2;
        const int reps = R / per_rep;
        DPK_REQUIRE(reps * per_rep == R, DPK_EINVAL, "leaf_backward: R=%d is not reps*%d", R, per_rep);
        // the inverse table reuses the `par` segment (parameter tables are not needed here)
        int *inv = (int *)w.par;
        DPK_REQUIRE((int64_t)reps * D * 4 <= (int64_t)w.G * w.SP * 2 * I * 4, DPK_EWORKSPACE,
                    "leaf_backward: inverse table does not fit");
        DPK_REQUIRE(hipMemsetAsync(inv, 0xff, (size_t)reps * D * 4, st) == hipSuccess, DPK_ELAUNCH,
                    "leaf_backward: memset");
        DPK_LAUNCH(leaf_inverse_kernel, dim3(w.G), dim3(256), 0, st, w.feat, w.srcr, d, w.SP, D, per_rep,
                           inv);
        DPK_LAUNCH(gaussian_leaf_bwd_x_kernel, dim3(grid_for(B * D, 256)), dim3(256), 0, st, x, g, B, D,
                           R, I, d, reps, inv, p0, p1, gx, drop_p, seed, dist);
        DPK_CHECK_LAUNCH("gaussian_leaf_bwd_x_kernel");
    }
    return DPK_OK;
}

extern "C" int dpk_gaussian_leaf_backward(const float *x, const float *g, int64_t B, int32_t D,
                                          const int64_t *mask, const uint8_t *pad_mask, const float *loc,
                                          const float *scale, int32_t R, int32_t I, int32_t d, float *grad_loc,
                                          float *grad_scale, float *grad_x, void *ws, int64_t ws_bytes,
                                          uint32_t flags, void *stream) {
    return leaf_backward_common(0, x, g, B, D, mask, pad_mask, loc, scale, R, I, d, grad_loc, grad_scale, grad_x,
                                ws, ws_bytes, flags, stream);
}

// d/dx of the Bernoulli leaf layer (the reference's autograd returns it, ratspn.py:243): grad_x[b,f] = sum over the
// (region, position) pairs holding variable f and the channels k of g[b,r,k] * logits[r,k,j]; 0 where x is NaN.
extern "C" int dpk_bernoulli_leaf_backward_input(const float *x, const float *g, int64_t B, int32_t D,
                                                 const int64_t *mask, const uint8_t *pad_mask, const float *logits,
                                                 int32_t R, int32_t I, int32_t d, float *grad_x, void *ws,
                                                 int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(grad_x, DPK_EINVAL, "bernoulli_leaf_backward_input: null pointer");
    return leaf_backward_common(1, x, g, B, D, mask, pad_mask, logits, nullptr, R, I, d, nullptr, nullptr, grad_x, ws,
                                ws_bytes, flags, stream);
}

extern "C" int dpk_bernoulli_leaf_backward(const float *x, const float *g, int64_t B, int32_t D,
                                           const int64_t *mask, const uint8_t *pad_mask, const float *logits,
                                           int32_t R, int32_t I, int32_t d, float *grad_logits, void *ws,
                                           int64_t ws_bytes, uint32_t flags, void *stream) {
    return leaf_backward_common(1, x, g, B, D, mask, pad_mask, logits, nullptr, R, I, d, grad_logits, nullptr,
                                nullptr, ws, ws_bytes, flags, stream);
}

// ------------------------------------------------------------------------------------
// Training-mode input dropout of the leaf layer (reference: ratspn.py:98-100): every element of the
// [B,R,I,d] log-density tensor is dropped (-> NaN -> 0) with probability p before the sum over d.
// Plain kernel for training batches: thread per (b, r, k).
// ------------------------------------------------------------------------------------
template <int DIST>
__global__ void leaf_fwd_dropout_kernel(const float *__restrict__ x, int64_t B, int D, const int64_t *__restrict__ mask,
                                        const uint8_t *__restrict__ pad_mask, const float *__restrict__ p0,
                                        const float *__restrict__ p1, int R, int I, int d, float drop_p, uint64_t seed,
                                        float *__restrict__ out) {
    const int64_t total = B * R * I;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % I);
        const int64_t br = e / I;
        const int r = (int)(br % R);
        const int64_t b = br / R;
        float acc = 0.f;
        for (int j = 0; j < d; ++j) {
            if (pad_mask && pad_mask[(int64_t)r * d + j]) continue;
            if (dropout_hit(seed, (uint64_t)e * d + j, drop_p)) continue;
            const float xv = x[b * D + mask[(int64_t)r * d + j]];
            const int64_t po = ((int64_t)r * I + k) * d + j;
            float t;
            if (DIST == 0) {
                const float sg = p1[po], dl = xv - p0[po];
                t = -(dl * dl) / (2.f * sg * sg) - logf(sg) - kLogSqrt2Pi;
            } else {
                const float l = p0[po];
                t = -(fmaxf(l, 0.f) - l * xv + log1pf(expf(-fabsf(l))));   // -BCEWithLogits(l, x)
            }
            acc += nan_to_num_f(t);
        }
        out[e] = acc;
    }
}

extern "C" int dpk_leaf_forward_dropout(int32_t dist, const float *x, int64_t B, int32_t D, const int64_t *mask,
                                        const uint8_t *pad_mask, const float *p0, const float *p1, int32_t R,
                                        int32_t I, int32_t d, float drop_p, uint64_t seed, float *out, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && R > 0 && I > 0 && d > 0, DPK_EINVAL, "leaf_forward_dropout: bad sizes");
    DPK_REQUIRE(dist == 0 || dist == 1, DPK_EINVAL, "leaf_forward_dropout: dist must be 0 (Normal) or 1 (Bernoulli)");
    DPK_REQUIRE(drop_p > 0.f && drop_p < 1.f, DPK_EINVAL, "leaf_forward_dropout: dropout rate must be in (0, 1)");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && mask && p0 && out && (dist == 1 || p1), DPK_EINVAL, "leaf_forward_dropout: null pointer");
    const int64_t total = B * R * I;
    if (dist == 0)
        DPK_LAUNCH(leaf_fwd_dropout_kernel<0>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           x, B, D, mask, pad_mask, p0, p1, R, I, d, drop_p, seed, out);
    else
        DPK_LAUNCH(leaf_fwd_dropout_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           x, B, D, mask, pad_mask, p0, p1, R, I, d, drop_p, seed, out);
    DPK_CHECK_LAUNCH("leaf_fwd_dropout_kernel");
    return DPK_OK;
}

extern "C" int dpk_leaf_backward_dropout(int32_t dist, const float *x, const float *g, int64_t B, int32_t D,
                                         const int64_t *mask, const uint8_t *pad_mask, const float *p0,
                                         const float *p1, int32_t R, int32_t I, int32_t d, float drop_p,
                                         uint64_t seed, float *grad_p0, float *grad_p1, float *grad_x, void *ws,
                                         int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(dist == 0 || dist == 1, DPK_EINVAL, "leaf_backward_dropout: dist must be 0 (Normal) or 1 (Bernoulli)");
    return leaf_backward_common(dist, x, g, B, D, mask, pad_mask, p0, p1, R, I, d, grad_p0, dist == 0 ? grad_p1 : nullptr,
                                grad_x, ws, ws_bytes, flags, stream, drop_p, seed);
}

// Sum-layer dropout (reference: ratspn.py:371-372, dgcspn.py:297-298): out = (dropped ? fill : x) on the flat
// element index; backward: grad_in = (dropped ? 0 : g).
__global__ void dropout_fill_kernel(const float *__restrict__ x, int64_t n, float drop_p, uint64_t seed, float fill,
                                    float *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = dropout_hit(seed, (uint64_t)e, drop_p) ? fill : x[e];
}

extern "C" int dpk_dropout_fill(const float *x, int64_t n, float drop_p, uint64_t seed, float fill, float *out,
                                void *stream) {
    DPK_REQUIRE(n >= 0 && drop_p >= 0.f && drop_p < 1.f, DPK_EINVAL, "dropout_fill: bad arguments");
    if (n == 0) return DPK_OK;
    DPK_REQUIRE(x && out, DPK_EINVAL, "dropout_fill: null pointer");
    DPK_LAUNCH(dropout_fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, drop_p,
                       seed, fill, out);
    DPK_CHECK_LAUNCH("dropout_fill_kernel");
    return DPK_OK;
}

// ------------------------------------------------------------------------------------
// Eval route for shapes the single-launch model kernel does not cover (e.g. I = S = 16, the reference's MNIST
// example): ProductLayer folded into the SumLayer / RootLayer above it, so the [B, P, N^2] product tensor
// (1 GB at B = 65536, N = 16) is never written.
//   prodsum : out[b,p,o] = logsumexp_{i,j}( x[b,2p,i] + x[b,2p+1,j] + lw[p,o,i,j] )
//   prodroot: out[b,k]   = logsumexp_{p,i,j}( x[b,2p,i] + x[b,2p+1,j] + lw[k,p,i,j] )
// Exp domain: ea_i = exp(a_i - max a), ec_j likewise, v = sum_i ea_i sum_j W[i,j] ec_j with the linear softmax
// weights on the scalar path (uniform per wave); v < 1e-30 falls back to the exact log-domain double loop.
// lane = sample, wave = partition (prodsum) / class block (prodroot).
// ------------------------------------------------------------------------------------
template <int NMAX>
__device__ __forceinline__ void load_children(const float *__restrict__ xa, const float *__restrict__ xc, int N,
                                              float (&a)[NMAX], float (&c)[NMAX], float (&ea)[NMAX],
                                              float (&ec)[NMAX], float &ma, float &mc) {
    float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        a[i] = i < N ? xa[i] : -INFINITY;
        c[i] = i < N ? xc[i] : -INFINITY;
        m1 = fmaxf(m1, a[i]);
        m2 = fmaxf(m2, c[i]);
    }
    ma = (m1 == -INFINITY) ? 0.f : m1;
    mc = (m2 == -INFINITY) ? 0.f : m2;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        ea[i] = __expf(a[i] - ma);
        ec[i] = __expf(c[i] - mc);
    }
}

template <int NMAX>
__device__ __forceinline__ float bilinear(cfloat_p W, int N, const float (&ea)[NMAX], const float (&ec)[NMAX]) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {   // fully unrolled: ea / ec stay in registers
        if (i < N) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (j < N) t = fmaf(W[i * N + j], ec[j], t);
            v = fmaf(ea[i], t, v);
        }
    }
    return v;
}

// exact log-domain (m, s) of logsumexp_{i,j}(a_i + c_j + lw[i,j]); rare and lane-divergent
__device__ __forceinline__ void exact_pair_lse(const float *__restrict__ xa, const float *__restrict__ xc,
                                               cfloat_p lw, int N, float &m, float &sum) {
    m = -INFINITY;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) m = fmaxf(m, xa[i] + xc[j] + lw[i * N + j]);
    sum = 0.f;
    if (m > -INFINITY)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) sum += expf(xa[i] + xc[j] + lw[i * N + j] - m);
}

template <int NMAX>
__global__ __launch_bounds__(256) void prodsum_fwd_kernel(const float *__restrict__ in, cfloat_p W, cfloat_p LW,
                                                         int64_t B, int R, int N, int S, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int P = R / 2;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const int64_t b = min((int64_t)blockIdx.x * 64 + lane, B - 1);
    if (p >= P) return;
    const float *xa = in + (b * R + 2 * p) * N, *xc = xa + N;
    float a[NMAX], c[NMAX], ea[NMAX], ec[NMAX], ma, mc;
    load_children<NMAX>(xa, xc, N, a, c, ea, ec, ma, mc);
    const int NN = N * N;
    for (int o = 0; o < S; ++o) {
        const float v = bilinear<NMAX>(W + ((int64_t)p * S + o) * NN, N, ea, ec);
        float r;
        if (v < 1e-30f) {
            float m, sum;
            exact_pair_lse(xa, xc, LW + ((int64_t)p * S + o) * NN, N, m, sum);
            r = (m > -INFINITY) ? m + logf(sum) : -INFINITY;
        } else {
            r = ma + mc + __logf(v);
        }
        if ((int64_t)blockIdx.x * 64 + lane < B) out[(b * P + p) * S + o] = r;
    }
}

// classes over the waves of the block (k = wave, wave + 4, ...), partitions inside the thread
template <int NMAX>
__global__ __launch_bounds__(256) void prodroot_fwd_kernel(const float *__restrict__ in, cfloat_p W, cfloat_p LW,
                                                          int64_t B, int R, int N, int C, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = R / 2, NN = N * N;
    const int64_t b = min((int64_t)blockIdx.x * 64 + lane, B - 1);
    for (int k = wave; k < C; k += 4) {
        float rm = -INFINITY, rs = 0.f;   // running (max, scaled sum) over the partitions
        for (int p = 0; p < P; ++p) {
            const float *xa = in + (b * R + 2 * p) * N, *xc = xa + N;
            float a[NMAX], c[NMAX], ea[NMAX], ec[NMAX], ma, mc;
            load_children<NMAX>(xa, xc, N, a, c, ea, ec, ma, mc);
            float m = ma + mc, v = bilinear<NMAX>(W + ((int64_t)k * P + p) * NN, N, ea, ec);
            if (v < 1e-30f) exact_pair_lse(xa, xc, LW + ((int64_t)k * P + p) * NN, N, m, v);
            if (v > 0.f && m > -INFINITY) {
                if (m > rm) {
                    rs = rs * __expf(rm - m) + v;
                    rm = m;
                } else {
                    rs += v * __expf(m - rm);
                }
            }
        }
        if ((int64_t)blockIdx.x * 64 + lane < B) out[b * C + k] = (rm > -INFINITY) ? rm + logf(rs) : -INFINITY;
    }
}

// ratspn_upper_gemm.hip: the same layers on the f16 matrix cores (8 / 16 nodes per region)
namespace dpk {
bool upper_mfma_shape_ok(bool root, int N, int S);
int64_t upper_mfma_frag_bytes(int R, int N, int S);
int upper_mfma_forward(bool root, const float *in, const float *W, const float *LW, int64_t B, int R, int N, int S,
                       float *out, void *frag, bool frag_cached, hipStream_t st, const unsigned *gate = nullptr);
int upper_mfma_tables(bool root, const float *weight, float *W, float *LW, int R, int N, int S, void *frag, hipStream_t st);
}

static int prod_fused_common(bool root, const float *in, const float *weight, int64_t B, int R, int N, int S,
                             float *out, void *ws, int64_t ws_bytes, uint32_t flags, void *stream, const char *who) {
    DPK_REQUIRE(B >= 0 && R > 0 && (R % 2) == 0 && N > 0 && S > 0, DPK_EINVAL, "%s: bad sizes", who);
    DPK_REQUIRE(N <= 32, DPK_EUNSUPPORTED, "%s: %d nodes per region > 32", who, N);
    DPK_REQUIRE(weight && ws, DPK_EINVAL, "%s: null pointer", who);
    const int P = R / 2;
    const int rows = root ? S : P * S;                 // S = classes for the root
    const int n = root ? P * N * N : N * N;
    const int64_t seg = align_up((int64_t)rows * n * 4, 256);
    DPK_REQUIRE(ws_bytes >= 2 * seg, DPK_EWORKSPACE, "%s: workspace too small", who);
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(in && out, DPK_EINVAL, "%s: null pointer", who);
    float *W = (float *)ws, *LW = (float *)((char *)ws + seg);
    hipStream_t st = (hipStream_t)stream;
    // DPK_FLAG_PARAMS_CACHED: softmax rows (and MFMA fragments) of an earlier call from this very weight are in ws;
    // DPK_FLAG_PARAMS_VERIFY: believed so, checked on the device (table kernels gated on the verdict)
    // (VERIFY = rebuild: the softmax rows + fragment pack cost what fingerprinting the weights would)
    const bool cached = (flags & DPK_FLAG_PARAMS_CACHED) != 0;
    {
        static const bool mfma = [] {
            const char *e = getenv("DPK_RATSPN_GEMM");
            return !(e && e[0] == '0');
        }();
        const int64_t fb = upper_mfma_frag_bytes(R, N, S);
        if (mfma && upper_mfma_shape_ok(root, N, S) && ws_bytes >= 2 * seg + fb &&
            (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
            // softmax rows and MFMA fragments in ONE launch (round 4; two before)
            if (!cached) {
                if (int rc = upper_mfma_tables(root, weight, W, LW, R, N, S, (char *)ws + 2 * seg, st)) return rc;
            }
            return upper_mfma_forward(root, in, W, LW, B, R, N, S, out, (char *)ws + 2 * seg, true, st);
        }
    }
    if (!cached) launch_softmax_rows(weight, rows, n, W, LW, st);
    const dim3 block(256);
#define DPK_LAUNCH_PS(NMAX)                                                                                         \
    do {                                                                                                            \
        if (root)                                                                                                   \
            DPK_LAUNCH(prodroot_fwd_kernel<NMAX>, dim3(cdiv(B, 64)), block, 0, st, in, as_const(W),         \
                               as_const(LW), B, R, N, S, out);                                                      \
        else                                                                                                        \
            DPK_LAUNCH(prodsum_fwd_kernel<NMAX>, dim3(cdiv(B, 64), cdiv(P, 4)), block, 0, st, in,           \
                               as_const(W), as_const(LW), B, R, N, S, out);                                         \
    } while (0)
    if (N <= 4) DPK_LAUNCH_PS(4);
    else if (N <= 8) DPK_LAUNCH_PS(8);
    else if (N <= 16) DPK_LAUNCH_PS(16);
    else DPK_LAUNCH_PS(32);
#undef DPK_LAUNCH_PS
    DPK_CHECK_LAUNCH(who);
    return DPK_OK;
}

// The softmax rows + MFMA fragments of a depth-2 model's sum layer and root layer in ONE launch, into the two layers' own
// workspaces (same layout as prod_fused_common carves): the following dpk_prodsum_forward / dpk_prodroot_forward calls take
// DPK_FLAG_PARAMS_CACHED.  DPK_EUNSUPPORTED when either layer is outside the MFMA route (the caller lets the layers build
// their own tables).
namespace dpk {
int upper_mfma_tables_pair(const float *w0, float *W0, float *LW0, int R0, int N0, int S0, void *frag0, const float *w1,
                           float *W1, float *LW1, int R1, int N1, int C, void *frag1, hipStream_t st);
}
extern "C" int dpk_upper_tables_pair(const float *sum_weight, int32_t R0, int32_t N0, int32_t S0, void *ws0, int64_t ws0_bytes,
                                     const float *root_weight, int32_t R1, int32_t N1, int32_t C, void *ws1,
                                     int64_t ws1_bytes, void *stream) {
    DPK_REQUIRE(sum_weight && root_weight && ws0 && ws1, DPK_EINVAL, "upper_tables_pair: null pointer");
    DPK_REQUIRE(R0 > 0 && (R0 % 2) == 0 && R1 > 0 && (R1 % 2) == 0 && N0 > 0 && N1 > 0 && S0 > 0 && C > 0, DPK_EINVAL,
                "upper_tables_pair: bad sizes");
    static const bool mfma = [] {
        const char *e = getenv("DPK_RATSPN_GEMM");
        return !(e && e[0] == '0');
    }();
    const int64_t seg0 = align_up((int64_t)(R0 / 2) * S0 * N0 * N0 * 4, 256), seg1 = align_up((int64_t)C * (R1 / 2) * N1 * N1 * 4, 256);
    const int64_t fb0 = upper_mfma_frag_bytes(R0, N0, S0), fb1 = upper_mfma_frag_bytes(R1, N1, C);
    if (!mfma || !upper_mfma_shape_ok(false, N0, S0) || !upper_mfma_shape_ok(true, N1, C) || ws0_bytes < 2 * seg0 + fb0 ||
        ws1_bytes < 2 * seg1 + fb1) {
        set_error("upper_tables_pair: a layer is outside the MFMA route");
        return DPK_EUNSUPPORTED;
    }
    return upper_mfma_tables_pair(sum_weight, (float *)ws0, (float *)((char *)ws0 + seg0), R0, N0, S0, (char *)ws0 + 2 * seg0,
                                  root_weight, (float *)ws1, (float *)((char *)ws1 + seg1), R1, N1, C, (char *)ws1 + 2 * seg1,
                                  (hipStream_t)stream);
}

extern "C" int64_t dpk_prodsum_workspace_bytes(int32_t R, int32_t N, int32_t S) {
    if (R <= 0 || N <= 0 || S <= 0) return DPK_EINVAL;
    return 2 * align_up((int64_t)(R / 2) * S * N * N * 4, 256) + upper_mfma_frag_bytes(R, N, S) + 256;
}
extern "C" int dpk_prodsum_forward(const float *in, const float *weight, int64_t B, int32_t R, int32_t N, int32_t S,
                                   float *out, void *ws, int64_t ws_bytes, uint32_t flags, void *stream) {
    return prod_fused_common(false, in, weight, B, R, N, S, out, ws, ws_bytes, flags, stream, "prodsum_forward");
}
extern "C" int dpk_prodroot_forward(const float *in, const float *weight, int64_t B, int32_t R, int32_t N, int32_t C,
                                    float *out, void *ws, int64_t ws_bytes, uint32_t flags, void *stream) {
    return prod_fused_common(true, in, weight, B, R, N, C, out, ws, ws_bytes, flags, stream, "prodroot_forward");
}
