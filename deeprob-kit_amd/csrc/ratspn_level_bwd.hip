// Backward of one RAT-SPN level -- ProductLayer followed by a SumLayer or the RootLayer -- in one launch (+ the log-softmax
// Jacobian of the weight gradient).
//
// reference: autograd through ProductLayer.forward (deeprob/spn/layers/ratspn.py:272-286: out[b,p,i*N+j] = x[b,2p,i] +
// x[b,2p+1,j]) and SumLayer.forward :363-378 / RootLayer.forward :446-458 (logsumexp(in + log_softmax(weight)) over the
// inputs), as the training loop runs it (torch/routines.py:150-170).
//
// The per-layer chain (dpk_product_forward to recompute the [B, P, N^2] product tensor, softmax rows, dpk_sum_backward,
// dpk_product_backward, Jacobian) is five launches and two [B, P, N^2] round trips through HBM per level: 48 us of a
// 150 us RAT-SPN (8,8) training step at B = 512 (round-4 trace), all of it launch and latency.  Here a lane owns one
// (i, j) pair of a partition: t[s] = g[b,p,s] exp(x[b,2p,i] + x[b,2p+1,j] + lw[p,s,i,j] - out[b,p,s]) -- the same term the
// layer kernels evaluate (sum_bwd_kernel) -- feeds the input gradient (sum over s, then over j for child a / over i for
// child c: lane reductions) and the weight gradient (sum over the samples of the work-group, one atomic per work-group).
// The product tensor never exists, the log-softmax of the partition's weight rows is a lane reduction in the prologue.
// (in, out) may carry a common per-sample shift (dpk_ratspn_forward_train's relative tensors): only in - out is used.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace dpk {

// sum / max over the G = 2^k consecutive lanes of a group (G <= 64), result in every lane of the group
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

struct LevelBwdArgs {
    const float *x;      // [B, 2P, N]
    const float *w;      // sum: [P, S, N*N] raw weights; root: [S, P*N*N]
    const float *out;    // sum: [B, P, S]; root: [B, S]
    const float *g;      // like out
    int64_t B;
    int P, S;
    float *gx;           // [B, 2P, N] or null
    float *glw;          // like w: gradient w.r.t. the log-softmax weights, accumulated with atomics (zeroed by the host); or null
    float *W;            // like w: softmax(w), for the Jacobian launch (written by the first sample tile)
    int tile;            // samples per work-group
    float *gw;           // like w: gradient w.r.t. the raw weights, written by the LAST work-group to finish (null with glw)
    unsigned *ticket;    // [P] (sum) / [1] (root): work-groups that have finished (zeroed by the host with glw)
    int rows, n;         // shape of w as rows of one softmax each
};

// d/dW of log_softmax rows [row0, row0 + nrows), gW = glw - softmax(W) * rowsum(glw), by the LAST of the `peers` work-groups
// that add into those rows: the atomics of the others are complete (device-scope fence before each ticket) and are read back
// through L2.  A separate launch for this costs 4 us per level; a single last work-group walking all 128 rows of the (8,8)
// sum layer cost 50 (dependent round trips): hence one ticket per partition.
__device__ __forceinline__ void level_jacobian_tail(const LevelBwdArgs &a, unsigned *ticket, unsigned peers, int row0, int nrows,
                                                    unsigned *flag_s) {
    // Everything the tail reads was written with device-scope atomics (performed at the coherence point, acknowledged through
    // vmcnt) and is read back with device-scope atomic loads: waiting for the acknowledgements is the whole release.  A
    // __threadfence() here writes back the XCD's L2 -- measured 17 .. 40 us per launch.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) *flag_s = (atomicAdd(ticket, 1u) == peers - 1u) ? 1u : 0u;
    __syncthreads();
    if (*flag_s == 0u) return;
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, n = a.n;
    for (int r = threadIdx.x >> 6; r < nrows; r += nw) {
        const float *gl = a.glw + (int64_t)(row0 + r) * n;
        const float *wr = a.W + (int64_t)(row0 + r) * n;
        constexpr int U = 8;                          // (rows of up to 512 weights: every load in flight at once)
        float s = 0.f;
        for (int i0 = 0; i0 < n; i0 += 64 * U) {
            float v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = i0 + k * 64 + lane;
                v[k] = i < n ? __hip_atomic_load(gl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) s += v[k];
        }
        s = wave_reduce_sum(s);
        for (int i0 = 0; i0 < n; i0 += 64 * U) {
            float v[U], w[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = i0 + k * 64 + lane;
                v[k] = i < n ? __hip_atomic_load(gl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                w[k] = i < n ? __hip_atomic_load(wr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = i0 + k * 64 + lane;
                if (i < n) a.gw[(int64_t)(row0 + r) * n + i] = v[k] - w[k] * s;
            }
        }
    }
}

// ---- a SumLayer level: work-group = (sample tile, partition); a group of N*N lanes per sample ---------------------------
template <int N, int S>
__global__ __launch_bounds__(256) void prodsum_bwd_kernel(const LevelBwdArgs a) {
    constexpr int NN = N * N, GPW = 64 / NN, WAVES = 4;
    __shared__ float red[WAVES][S][64];
    __shared__ unsigned last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = lane % NN, grp = lane / NN;           // (i, j) pair; sample slot of the wave
    const int i = e / N, j = e % N;
    const int p = blockIdx.y, P = a.P;
    // log-softmax of the partition's S weight rows: a row = the N*N lanes of a group
    float lw[S], acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float wv = a.w[((int64_t)p * S + s) * NN + e];
        const float mx = group_max<NN>(wv);
        const float sum = group_sum<NN>(expf(wv - mx));
        lw[s] = wv - mx - logf(sum);
        acc[s] = 0.f;
        if (blockIdx.x == 0 && wave == 0 && grp == 0)
            __hip_atomic_store(a.W + ((int64_t)p * S + s) * NN + e, expf(lw[s]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int64_t b0 = (int64_t)blockIdx.x * a.tile;
    const int64_t b1 = min(b0 + a.tile, a.B);
    // samples of the tile: slot (wave, grp) takes b0 + wave * GPW + grp, + WAVES * GPW, ...
    constexpr int UN = 2;
    for (int64_t bb = b0 + wave * GPW + grp; bb < b1; bb += (int64_t)UN * WAVES * GPW) {
        float xa[UN], xc[UN], o[UN][S], gg[UN][S];
        bool in[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t b = bb + (int64_t)u * WAVES * GPW;
            in[u] = b < b1;
            const int64_t bc = in[u] ? b : b0;
            xa[u] = a.x[(bc * 2 * P + 2 * p) * N + i];
            xc[u] = a.x[(bc * 2 * P + 2 * p + 1) * N + j];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                o[u][s] = a.out[(bc * P + p) * S + s];
                gg[u][s] = a.g[(bc * P + p) * S + s];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t b = bb + (int64_t)u * WAVES * GPW;
            float tot = 0.f;
            const TwoSum ac = two_sum(xa[u], xc[u]);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // an all -inf row has out = -inf: its gradient is defined as zero (sum_bwd_kernel, and the masked_fill
                // guard inside torch.logsumexp's backward).  The responsibilities of a sum node are normalised over its
                // N*N inputs (the lanes of the group) instead of trusting the stored `out`, whose own rounding (half an ulp
                // of ~560) would otherwise scale all of them alike.
                const float e = (in[u] && o[u][s] > -INFINITY) ? expf(resp_arg(ac, lw[s], o[u][s])) : 0.f;
                const float z = dpp_group_sum<NN>(e);
                const float t = z > 0.f ? gg[u][s] * (e * __builtin_amdgcn_rcpf(z)) : 0.f;   // (hardware reciprocal, 1 ulp: the IEEE division is ten instructions)
                acc[s] += t;
                tot += t;
            }
            if (a.gx != nullptr) {
                // child a (region 2p) node i: sum over j = the N consecutive lanes; child c node j: sum over i = stride N
                float ga = tot, gc = tot;
#pragma unroll
                for (int of = N / 2; of > 0; of >>= 1) ga += __shfl_xor(ga, of, 64);
#pragma unroll
                for (int of = NN / 2; of >= N; of >>= 1) gc += __shfl_xor(gc, of, 64);
                if (in[u] && j == 0) a.gx[(b * 2 * P + 2 * p) * N + i] = ga;
                if (in[u] && i == 0) a.gx[(b * 2 * P + 2 * p + 1) * N + j] = gc;
            }
        }
    }
    if (a.glw != nullptr) {
        // the wave's sample slots, then the work-group's waves, then one atomic per (s, i, j)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float v = acc[s];
#pragma unroll
            for (int of = 32; of >= NN; of >>= 1) v += __shfl_xor(v, of, 64);
            red[wave][s][lane] = v;
        }
        __syncthreads();
        for (int q = tid; q < S * NN; q += 256) {
            const int s = q / NN, ee = q % NN;
            const float t = (red[0][s][ee] + red[1][s][ee]) + (red[2][s][ee] + red[3][s][ee]);
            atomicAdd(a.glw + ((int64_t)p * S + s) * NN + ee, t);
        }
        level_jacobian_tail(a, a.ticket + p, gridDim.x, p * S, S, &last_s);
    }
}

// ---- a SumLayer level with 16 nodes per region: the (i, j) pairs of a partition are the 256 threads of a work-group -----
// (the example model of examples/ratspn_mnist.py: rg_batch = rg_sum = 16, depth 3 -- its levels were the per-layer chain:
// product recomputed 13 us + softmax rows 5 + sum backward 33 + product backward 11 + Jacobian 5 per level.)
// Same term as prodsum_bwd_kernel; what differs is that a row's log-softmax and the column sums over i cross the four
// waves (LDS), and the samples of the tile are walked one at a time by the whole work-group.
template <int S>
__global__ __launch_bounds__(256) void prodsum16_bwd_kernel(const LevelBwdArgs a) {
    constexpr int N = 16, NN = 256;
    __shared__ float red[4][S];
    __shared__ float bc[2][S];
    __shared__ float colp[2][4][N];
    __shared__ float zred[2][4][2][S];      // [trip parity][wave][sample of the trip][sum node]: normalisation partials
    __shared__ unsigned last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid >> 4, j = tid & 15;
    const int p = blockIdx.y, P = a.P;
    float lw[S], acc[S];
    {
        float wv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) wv[s] = a.w[((int64_t)p * S + s) * NN + tid];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float mx = wave_reduce_max(wv[s]);
            if (lane == 0) red[wave][s] = mx;
        }
        __syncthreads();
        if (tid < S) bc[0][tid] = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float sum = wave_reduce_sum(expf(wv[s] - bc[0][s]));
            if (lane == 0) red[wave][s] = sum;      // (the maxima were read behind the barrier above)
        }
        __syncthreads();
        if (tid < S) bc[1][tid] = logf((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            lw[s] = wv[s] - bc[0][s] - bc[1][s];
            acc[s] = 0.f;
            if (blockIdx.x == 0)
                __hip_atomic_store(a.W + ((int64_t)p * S + s) * NN + tid, expf(lw[s]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int64_t b0 = (int64_t)blockIdx.x * a.tile, b1 = min(b0 + a.tile, a.B);
    // two samples per trip, their loads requested together.  (Measured and dropped: the factored form W e^{xa-ma} e^{xc-mc}
    // g e^{ma+mc-out} with 3 instead of 16 exponentials per thread and sample -- same 63-65 us: the launch is bound by its
    // 2.1 M weight-gradient atomics and the serial walk over the tile, not by the vector ALU.)
    constexpr int UN = 2;
    for (int64_t bb = b0; bb < b1; bb += UN) {
        float xa[UN], xc[UN], o[UN][S], gg[UN][S];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t b = min(bb + u, b1 - 1);
            xa[u] = a.x[(b * 2 * P + 2 * p) * N + i];
            xc[u] = a.x[(b * 2 * P + 2 * p + 1) * N + j];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                o[u][s] = a.out[(b * P + p) * S + s];
                gg[u][s] = a.g[(b * P + p) * S + s];
            }
        }
        float ga[UN], gc[UN];
        // responsibilities normalised over the node's 256 inputs = the work-group (not by the stored `out`, whose rounding
        // would scale them all alike: see two_sum in common.h); the partial sums of the four waves meet in LDS,
        // double-buffered by trip parity (a wave reaches trip k + 2 only through the barrier of trip k + 1)
        float e[UN][S];
        const int zp = (int)(((bb - b0) / UN) & 1);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool in = bb + u < b1;
            const TwoSum ac = two_sum(xa[u], xc[u]);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                e[u][s] = (in && o[u][s] > -INFINITY) ? expf(resp_arg(ac, lw[s], o[u][s])) : 0.f;
                const float zw = dpp_group_sum<64>(e[u][s]);
                if (lane == 0) zred[zp][wave][u][s] = zw;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float tot = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float z = (zred[zp][0][u][s] + zred[zp][1][u][s]) + (zred[zp][2][u][s] + zred[zp][3][u][s]);
                const float t = z > 0.f ? gg[u][s] * (e[u][s] * __builtin_amdgcn_rcpf(z)) : 0.f;
                acc[s] += t;
                tot += t;
            }
            // child a (region 2p), node i: sum over j = 16 consecutive lanes; child c, node j: sum over i = the wave's four
            // rows by shuffle, then the four waves through LDS
            ga[u] = tot;
            gc[u] = tot;
#pragma unroll
            for (int of = 8; of > 0; of >>= 1) ga[u] += __shfl_xor(ga[u], of, 64);
            gc[u] += __shfl_xor(gc[u], 16, 64);
            gc[u] += __shfl_xor(gc[u], 32, 64);
        }
        if (a.gx != nullptr) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (bb + u < b1 && j == 0) a.gx[((bb + u) * 2 * P + 2 * p) * N + i] = ga[u];
                if (lane < N) colp[u][wave][lane] = gc[u];
            }
            __syncthreads();
            if (tid < UN * N) {
                const int u = tid >> 4, jj = tid & 15;
                if (bb + u < b1)
                    a.gx[((bb + u) * 2 * P + 2 * p + 1) * N + jj] = (colp[u][0][jj] + colp[u][1][jj]) + (colp[u][2][jj] + colp[u][3][jj]);
            }
            __syncthreads();
        }
    }
    if (a.glw != nullptr) {
#pragma unroll
        for (int s = 0; s < S; ++s) atomicAdd(a.glw + ((int64_t)p * S + s) * NN + tid, acc[s]);
        level_jacobian_tail(a, a.ticket + p, gridDim.x, p * S, S, &last_s);
    }
}

// ---- the RootLayer level: one work-group per sample tile, a lane per (partition, i, j) input of the root ----------------
// CB classes at a time (their log-softmax rows over all M = P N^2 inputs are work-group reductions).
template <int N, int CB>
__global__ __launch_bounds__(1024) void prodroot_bwd_kernel(const LevelBwdArgs a) {
    constexpr int NN = N * N;
    __shared__ float red[16][CB];
    __shared__ float bc[2][CB];
    __shared__ float zred[16][4 * CB];
    __shared__ float zsum[4 * CB];
    __shared__ unsigned last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int P = a.P, M = P * NN, C = a.S;
    const bool live = tid < M;
    const int m = live ? tid : 0;
    const int p = m / NN, e = m % NN, i = e / N, j = e % N;
    constexpr int TB = 4;                       // samples per pass; a.tile = TB * passes
    const bool keep = C <= CB;                  // one chunk of classes: the weight gradient waits in registers for the last pass
    float acc_keep[CB];
#pragma unroll
    for (int q = 0; q < CB; ++q) acc_keep[q] = 0.f;
    const int64_t t0 = (int64_t)blockIdx.x * a.tile, t1 = min(t0 + a.tile, a.B);
    for (int64_t b0 = t0; b0 < t1; b0 += TB) {
        const int nb = (int)min((int64_t)TB, t1 - b0);
        float xa[TB], xc[TB], tot[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
            const int64_t b = b0 + (u < nb ? u : 0);
            xa[u] = a.x[(b * 2 * P + 2 * p) * N + i];
            xc[u] = a.x[(b * 2 * P + 2 * p + 1) * N + j];
            tot[u] = 0.f;
        }
        for (int c0 = 0; c0 < C; c0 += CB) {
            float wv[CB], lw[CB];
#pragma unroll
            for (int q = 0; q < CB; ++q) wv[q] = (live && c0 + q < C) ? a.w[(int64_t)(c0 + q) * M + m] : -INFINITY;
            float o[TB][CB], gg[TB][CB];
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    const bool ok = u < nb && c0 + q < C;
                    o[u][q] = ok ? a.out[(b0 + u) * C + c0 + q] : -INFINITY;
                    gg[u][q] = ok ? a.g[(b0 + u) * C + c0 + q] : 0.f;
                }
            // log-softmax of the CB rows over the work-group (recomputed per pass: a dozen barriers against a pass's loads)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CB; ++q) {
                const float mx = wave_reduce_max(wv[q]);
                if (lane == 0) red[wave][q] = mx;
            }
            __syncthreads();
            if (tid < CB) {
                float mx = -INFINITY;
                for (int w = 0; w < nw; ++w) mx = fmaxf(mx, red[w][tid]);
                bc[0][tid] = mx;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CB; ++q) {
                const float sum = wave_reduce_sum(live ? expf(wv[q] - bc[0][q]) : 0.f);
                if (lane == 0) red[wave][q] = sum;      // (the maxima were read behind the barrier above)
            }
            __syncthreads();
            if (tid < CB) {
                float sum = 0.f;
                for (int w = 0; w < nw; ++w) sum += red[w][tid];
                bc[1][tid] = logf(sum);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CB; ++q) {
                lw[q] = wv[q] - bc[0][q] - bc[1][q];
                if (blockIdx.x == 0 && b0 == t0 && live && c0 + q < C)
                    __hip_atomic_store(a.W + (int64_t)(c0 + q) * M + m, expf(lw[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float acc[CB];
#pragma unroll
            for (int q = 0; q < CB; ++q) acc[q] = 0.f;
            // responsibilities of the root's M inputs, normalised over the work-group (= all inputs of a class) instead of
            // trusting the stored `out`: its rounding -- half an ulp of ~1100, 6e-5 -- would scale every gradient of the
            // sample alike (see two_sum in common.h)
            float e[TB][CB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const TwoSum ac = two_sum(xa[u], xc[u]);
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    e[u][q] = (o[u][q] > -INFINITY) ? expf(resp_arg(ac, lw[q], o[u][q])) : 0.f;
                    if (c0 + q < C) {      // (uniform: the classes that exist -- one, for the generative models)
                        const float zw = dpp_group_sum<64>(e[u][q]);
                        if (lane == 0) zred[wave][u * CB + q] = zw;
                    }
                }
            }
            __syncthreads();
            if (tid < TB * CB) {
                float z = 0.f;
                for (int w = 0; w < nw; ++w) z += zred[w][tid];
                zsum[tid] = z;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    const float z = zsum[u * CB + q];
                    const float t = (c0 + q < C && z > 0.f) ? gg[u][q] * (e[u][q] * __builtin_amdgcn_rcpf(z)) : 0.f;   // (classes beyond C: their LDS slots were never written)
                    acc[q] += t;
                    tot[u] += t;
                }
            if (keep) {
#pragma unroll
                for (int q = 0; q < CB; ++q) acc_keep[q] += acc[q];
            } else if (a.glw != nullptr && live) {
#pragma unroll
                for (int q = 0; q < CB; ++q)
                    if (c0 + q < C) atomicAdd(a.glw + (int64_t)(c0 + q) * M + m, acc[q]);
            }
        }
        if (a.gx != nullptr) {
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                float ga = live ? tot[u] : 0.f, gc = ga;
#pragma unroll
                for (int of = N / 2; of > 0; of >>= 1) ga += __shfl_xor(ga, of, 64);
#pragma unroll
                for (int of = NN / 2; of >= N; of >>= 1) gc += __shfl_xor(gc, of, 64);
                if (live && u < nb && j == 0) a.gx[((b0 + u) * 2 * P + 2 * p) * N + i] = ga;
                if (live && u < nb && i == 0) a.gx[((b0 + u) * 2 * P + 2 * p + 1) * N + j] = gc;
            }
        }
    }
    if (keep && a.glw != nullptr && live) {
#pragma unroll
        for (int q = 0; q < CB; ++q)
            if (q < C) atomicAdd(a.glw + (int64_t)q * M + m, acc_keep[q]);
    }
    if (a.glw != nullptr) level_jacobian_tail(a, a.ticket, gridDim.x, 0, C, &last_s);
}

template <int N>
static int launch_sum_level(const LevelBwdArgs &a, int S, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(a.B, a.tile), (unsigned)a.P);
    switch (S) {
        case 2: DPK_LAUNCH((prodsum_bwd_kernel<N, 2>), grid, dim3(256), 0, st, a); return DPK_OK;
        case 4: DPK_LAUNCH((prodsum_bwd_kernel<N, 4>), grid, dim3(256), 0, st, a); return DPK_OK;
        case 8: DPK_LAUNCH((prodsum_bwd_kernel<N, 8>), grid, dim3(256), 0, st, a); return DPK_OK;
    }
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk

using namespace dpk;

extern "C" int dpk_prodsum_backward(const float *in, const float *weight, const float *out, const float *g, int64_t B,
                                    int32_t R, int32_t N, int32_t S, int32_t root, float *grad_in, float *grad_weight,
                                    void *ws, int64_t ws_bytes, void *stream) {
    DPK_REQUIRE(B >= 0 && R > 0 && (R % 2) == 0 && N > 0 && S > 0, DPK_EINVAL, "prodsum_backward: bad sizes");
    DPK_REQUIRE(weight && ws && (B == 0 || (in && out && g)), DPK_EINVAL, "prodsum_backward: null pointer");
    const int P = R / 2, NN = N * N;
    const bool wide16 = !root && N == 16 && (S == 8 || S == 16);
    const bool shape_ok = ((N == 2 || N == 4 || N == 8) && (root ? (P * NN <= 1024) : (S == 2 || S == 4 || S == 8)) && P <= 65535) ||
                          (wide16 && P <= 65535);
    if (!shape_ok) {
        set_error("prodsum_backward: (nodes=%d, sums=%d, partitions=%d) not built; chain the layers' backward entry points", N, S, P);
        return DPK_EUNSUPPORTED;
    }
    const int64_t cells = (int64_t)P * S * NN;             // weight entries, sum and root alike
    const int64_t seg = align_up(cells * 4, 256);
    DPK_REQUIRE(ws_bytes >= 3 * seg, DPK_EWORKSPACE, "prodsum_backward: workspace %lld < %lld", (long long)ws_bytes,
                (long long)(3 * seg));
    hipStream_t st = (hipStream_t)stream;
    // ws: [W | (unused: the layer kernels' LW) ... tickets | glw]; the tickets sit at the end of the LW segment, in front of
    // glw, so that one memset clears both
    float *W = (float *)ws, *glw = (float *)((char *)ws + 2 * seg);
    const int64_t tick_bytes = align_up((int64_t)(root ? 1 : P) * 4, 256);   // (<= seg: a partition has >= 4 weights)
    unsigned *ticket = (unsigned *)((char *)ws + 2 * seg - tick_bytes);
    if (grad_weight) {
        hipError_t e = hipMemsetAsync(B > 0 ? (void *)ticket : (void *)grad_weight, 0,
                                      B > 0 ? (size_t)(cells * 4 + tick_bytes) : (size_t)cells * 4, st);
        DPK_REQUIRE(e == hipSuccess, DPK_ELAUNCH, "prodsum_backward: memset: %s", hipGetErrorString(e));
    }
    LevelBwdArgs a{};
    a.x = in; a.w = weight; a.out = out; a.g = g; a.B = B; a.P = P; a.S = S; a.gx = grad_in;
    a.glw = grad_weight ? glw : nullptr; a.W = W; a.gw = grad_weight; a.ticket = ticket;
    a.rows = root ? S : P * S; a.n = root ? P * NN : NN;
    if (B > 0) {
        if (root) {
            // 4 samples per pass; more passes per work-group once the grid covers the chip twice (fewer atomics per weight)
            int passes = (int)std::min<int64_t>(64, std::max<int64_t>(1, B / (4 * 2 * (int64_t)device_cus())));
            a.tile = 4 * passes;
            const int threads = (int)align_up(P * NN, 64);
            const dim3 grid((unsigned)cdiv(B, a.tile));
            if (N == 2) DPK_LAUNCH((prodroot_bwd_kernel<2, 8>), grid, dim3(threads), 0, st, a);
            else if (N == 4) DPK_LAUNCH((prodroot_bwd_kernel<4, 8>), grid, dim3(threads), 0, st, a);
            else DPK_LAUNCH((prodroot_bwd_kernel<8, 8>), grid, dim3(threads), 0, st, a);
        } else {
            // sample tiles: short while the grid would leave compute units idle (more atomics per weight)
            int tile = 256;
            while (tile > 8 && cdiv(B, tile) * P < 2 * device_cus()) tile /= 2;
            a.tile = tile;
            int rc = DPK_OK;
            if (wide16) {
                // (a work-group per partition walks its samples one at a time: tiles short enough to cover the chip)
                int t16 = 64;
                while (t16 > 4 && cdiv(B, t16) * P < 2 * device_cus()) t16 /= 2;   // (longer tiles, fewer atomics: 91 against 63 us -- the walk over the tile is serial)
                a.tile = t16;
                const dim3 grid16((unsigned)cdiv(B, t16), (unsigned)P);
                if (S == 8) DPK_LAUNCH((prodsum16_bwd_kernel<8>), grid16, dim3(256), 0, st, a);
                else DPK_LAUNCH((prodsum16_bwd_kernel<16>), grid16, dim3(256), 0, st, a);
            } else {
                rc = N == 2 ? launch_sum_level<2>(a, S, st) : (N == 4 ? launch_sum_level<4>(a, S, st) : launch_sum_level<8>(a, S, st));
            }
            if (rc) return rc;
        }
        DPK_CHECK_LAUNCH("prodsum_bwd_kernel");
    }
    return DPK_OK;
}
