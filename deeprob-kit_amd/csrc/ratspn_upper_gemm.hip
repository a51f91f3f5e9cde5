// ProductLayer folded into the SumLayer / RootLayer above it, on the f16 matrix cores (8 or 16 nodes per region).
//
// reference: ProductLayer.forward (deeprob/spn/layers/ratspn.py:272-286) + SumLayer.forward (:363-378) /
//            RootLayer.forward (:446-458); eval route of models wider than the fused kernels (rg_batch = rg_sum = 16 of
//            examples/ratspn_mnist.py, the 8-channel default of experiments/ratspn.py below 32k samples)
//
//   out[b,p,o] = logsumexp_ij(a_i + c_j + log_softmax(W)[p,o,i,j]),  a = in[b,2p,:], c = in[b,2p+1,:]
//              = max a + max c + log sum_i ea_i (sum_j softmax(W)[p,o,i,j] ec_j),   ea = e^{a - max a}, ec likewise
//
// The inner sum is a GEMM per partition: T[(o,i), b] = sum_j Wp[(o,i), j] ec[j, b] with K = N (one K-step of 16; N = 8
// uses half of it).  It runs as three v_mfma_f32_32x32x16_f16 on two-way f16 splits of both operands (weights and
// exponentials are in [0, 1]; >= 22 significant bits per product, fp32 accumulation) instead of S N^2 fp32 FMAs per
// sample and partition on the VALU: 135k flop per sample at N = S = 16, the layer that bounded that model.  The row
// order (o, i) is chosen so that lane (sample l & 31, half h = l >> 5) receives all N rows i of the outputs o it
// owns: the outer sum over i is a dot product with the lane's own ea, in registers.  A wave owns 32 samples; weight
// fragments (256 KB at N = S = 16, L2 resident) are read straight from global memory, no LDS.  A node whose scaled
// sum falls below 1e-8 (dominant pair under a small weight: the split's absolute error would show) is redone in the
// exact two-pass log domain.
#include "common.h"
#include "ratspn_gemm_common.h"
#include <math.h>
#include <stdlib.h>

namespace dpk {

typedef const __attribute__((address_space(1))) half8 ug_h8;
typedef const __attribute__((address_space(1))) gf32x4 ug_f4;

// Both MFMA operands live in [0, 1] (softmax weights, exponentials scaled to their maximum).  An f16 split keeps 22
// significant bits only while the LOW half is a normal f16 number: unscaled, a weight of 1/64 would be carried to 5e-6
// relative, and the error is ABSOLUTE (3e-8, half a subnormal step) against a sum v that can itself be as small as the
// dominant pair's weight.  Both operands are therefore scaled by 2^15 before the split (exact; <= 32768, inside the f16
// range) and the accumulated sum by 2^-30 afterwards: absolute error 1e-12 per operand.  A node whose sum falls below
// kUpExactBelow (where that absolute error would reach 1e-4 relative) is redone in the exact log domain.
constexpr float kUpScale = 32768.f, kUpScaleLog2 = 15.f, kUpUnscale = 1.f / (32768.f * 32768.f), kUpExactBelow = 1e-8f;

// A-fragment tables from the linear softmax weights
//   sum layer : Wl [P][S][N*N]      -> frag [P][S*N/32 tiles][2][64][8]
//   root layer: Wl [C][P][N*N]      -> frag [P][ceil(C*N/32) tiles][2][64][8]
__global__ __launch_bounds__(256) void upper_pack_kernel(const float *__restrict__ Wl, int P, int N, int S, int root,
                                                         int tiles, uint16_t *__restrict__ frag, const unsigned *gate) {
    if (gate_closed(gate)) return;   // (tables still match the live weights: common.h params_gate)
    const int64_t total = (int64_t)P * tiles * 64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const int t = (int)((e >> 6) % tiles), p = (int)((e >> 6) / tiles);
        const int row = l & 31, hg = l >> 5;
        int o, i;
        bool ok = true;
        if (root) {   // natural order: global row = class * N + i
            const int g = 32 * t + row;
            o = g / N;
            i = g - o * N;
            ok = o < S;
        } else if (N == 16) {   // lane half h of the C layout gets o = 2 t + h, i = its register index
            const int hrow = (row >> 2) & 1, u = (row & 3) + 4 * (row >> 3);
            o = 2 * t + hrow;
            i = u;
        } else {                // N == 8: o = 4 t + 2 h + (u >> 3), i = u & 7
            const int hrow = (row >> 2) & 1, u = (row & 3) + 4 * (row >> 3);
            o = 4 * t + 2 * hrow + (u >> 3);
            i = u & 7;
        }
        half8 vh, vl;
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            const int j = 8 * hg + el;
            float v = 0.f;
            if (ok && j < N) {
                const int64_t src = root ? (((int64_t)o * P + p) * N + i) * N + j : (((int64_t)p * S + o) * N + i) * N + j;
                v = Wl[src] * kUpScale;
            }
            _Float16 hi, lo;
            split_f16(v, hi, lo);
            vh[el] = hi; vl[el] = lo;
        }
        uint16_t *dst = frag + ((int64_t)p * tiles + t) * 1024 + l * 8;
        *reinterpret_cast<half8 *>(dst) = vh;
        *reinterpret_cast<half8 *>(dst + 512) = vl;
    }
}

// Softmax rows AND their fragments in one launch (round 4: the folded route's per-call table rebuild was two launches per
// layer -- softmax rows, then upper_pack_kernel over them; at B = 4096 the four small launches of a (16,16) model cost a
// third of its kernels' time).  One block per softmax row: W = softmax(w[row]), LW = log_softmax(w[row]) and every
// fragment entry the row owns (the inverse of upper_pack_kernel's mapping); the root layout's padding rows (class tiles
// beyond C N rows) are zeroed by one extra block.
struct UpperTabJob {
    const float *w;
    float *W, *LW;
    uint16_t *frag;
    int rows, n, P, N, S, root, tiles, blocks;
};
__device__ __forceinline__ void upper_tables_body(const float *__restrict__ w, int rows, int n, float *__restrict__ W,
                                                  float *__restrict__ LW, uint16_t *__restrict__ frag, int P, int N, int S,
                                                  int root, int tiles, int blk, float *red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blk >= rows) {   // root layout: rows S N .. tiles * 32 of every partition are zero
        const int g0 = S * N, g1 = tiles * 32;
        const int per_p = (g1 - g0) * 16;
        for (int e = tid; e < P * per_p; e += blockDim.x) {
            const int p = e / per_p, q = e - p * per_p;
            const int g = g0 + q / 16, k = q % 16;
            uint16_t *dst = frag + ((int64_t)p * tiles + g / 32) * 1024 + ((k >> 3) * 32 + (g & 31)) * 8 + (k & 7);
            dst[0] = 0;
            dst[512] = 0;
        }
        return;
    }
    const int row = blk;
    const float *src = w + (int64_t)row * n;
    // rows of up to 2048 weights are read ONCE, every load in flight together (three passes over global memory were three
    // dependent round trips of this 7 us launch)
    constexpr int RV = 8;
    const bool fits = n <= 256 * RV;
    float v[RV];
    if (fits) {
#pragma unroll
        for (int k = 0; k < RV; ++k) v[k] = (tid + 256 * k < n) ? src[tid + 256 * k] : -INFINITY;
    }
    float m = -INFINITY;
    if (fits) {
#pragma unroll
        for (int k = 0; k < RV; ++k) m = fmaxf(m, v[k]);
    } else {
        for (int i = tid; i < n; i += blockDim.x) m = fmaxf(m, src[i]);
    }
    m = wave_reduce_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    if (fits) {
#pragma unroll
        for (int k = 0; k < RV; ++k)
            if (tid + 256 * k < n) sum += expf(v[k] - m);
    } else {
        for (int i = tid; i < n; i += blockDim.x) sum += expf(src[i] - m);
    }
    sum = wave_reduce_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float ls = logf((red[4] + red[5]) + (red[6] + red[7]));
    const int NN = N * N;
    for (int e = tid, kk = 0; e < n; e += blockDim.x, ++kk) {
        float sv;
        if (fits) {
            sv = v[0];
#pragma unroll
            for (int k = 1; k < RV; ++k) sv = (kk == k) ? v[k] : sv;
        } else {
            sv = src[e];
        }
        const float l = sv - m - ls, wl = expf(l);
        LW[(int64_t)row * n + e] = l;
        W[(int64_t)row * n + e] = wl;
        int p, o, t, r;
        const int ij = root ? e % NN : e, i = ij / N, j = ij - i * N;
        if (root) {   // global row = class * N + i
            o = row;
            p = e / NN;
            const int g = o * N + i;
            t = g >> 5;
            r = g & 31;
        } else {
            p = row / S;
            o = row - p * S;
            int hrow, u;
            if (N == 16) { t = o >> 1; hrow = o & 1; u = i; }
            else { t = o >> 2; hrow = (o >> 1) & 1; u = (o & 1) * 8 + i; }
            r = (u & 3) + 8 * (u >> 2) + 4 * hrow;
        }
        _Float16 hi, lo;
        split_f16(wl * kUpScale, hi, lo);
        uint16_t *dst = frag + ((int64_t)p * tiles + t) * 1024 + ((j >> 3) * 32 + r) * 8 + (j & 7);
        dst[0] = __builtin_bit_cast(uint16_t, hi);
        dst[512] = __builtin_bit_cast(uint16_t, lo);
        if (N == 8) {   // the upper half of the K-step is unused
            dst[256] = 0;
            dst[256 + 512] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void upper_tables_kernel(const float *__restrict__ w, int rows, int n, float *__restrict__ W,
                                                           float *__restrict__ LW, uint16_t *__restrict__ frag, int P, int N,
                                                           int S, int root, int tiles) {
    __shared__ float red[8];
    upper_tables_body(w, rows, n, W, LW, frag, P, N, S, root, tiles, (int)blockIdx.x, red);
}
// the tables of a model's sum layer AND root layer in one launch (the folded route's default mode rebuilds both per call:
// two 6 us launches in front of two 10 us kernels at B = 4096)
__global__ __launch_bounds__(256) void upper_tables_pair_kernel(const UpperTabJob j0, const UpperTabJob j1) {
    __shared__ float red[8];
    const int blk = (int)blockIdx.x;
    if (blk < j0.blocks) upper_tables_body(j0.w, j0.rows, j0.n, j0.W, j0.LW, j0.frag, j0.P, j0.N, j0.S, j0.root, j0.tiles, blk, red);
    else upper_tables_body(j1.w, j1.rows, j1.n, j1.W, j1.LW, j1.frag, j1.P, j1.N, j1.S, j1.root, j1.tiles, blk - j0.blocks, red);
}

template <int N>
__device__ __forceinline__ void upper_load(const float *xa, float (&a)[N], float (&c)[N]) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const gf32x4 va = *(ug_f4 *)(xa + 4 * q), vc = *(ug_f4 *)(xa + N + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[4 * q + i] = va[i];
            c[4 * q + i] = vc[i];
        }
    }
}

template <int N> __device__ __forceinline__ float upper_max(const float (&x)[N]) {
    float m = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = fmaxf(m, x[i]);
    return (m == -INFINITY) ? 0.f : m;
}

// exact log-domain (m, s) of logsumexp_ij(a_i + c_j + lw[i*N + j]) for ONE node, by the whole wave (every lane calls it with the same arguments): the N^2 pairs are dealt over the
// 64 lanes, two wave reductions.  The vanished-node fallbacks of the kernels below call it for the flagged (sample, node)
// pairs only: a launch of a few thousand samples meets a handful of them (a softmax weight of 1e-8 on the dominant pair),
// and redoing every node of the wave's 32 samples in the serial form above made those few waves 20 us long -- the
// duration of the whole launch at small batches (round 4).
template <int N>
__device__ __forceinline__ void upper_exact_ms_wave(const float *xa, const float *lw, int lane, float &m, float &sum) {
    float t[(N * N + 63) / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < (N * N + 63) / 64; ++k) {
        const int e = k * 64 + lane;
        t[k] = (e < N * N) ? xa[e / N] + xa[N + e % N] + lw[e] : -INFINITY;
        mx = fmaxf(mx, t[k]);
    }
    mx = wave_reduce_max(mx);
    float sm = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
        for (int k = 0; k < (N * N + 63) / 64; ++k) sm += expf(t[k] - mx);
    }
    m = mx;
    sum = wave_reduce_sum(sm);
}

struct UpperArgs {
    const float *in;          // [B, R, N]
    const uint16_t *frag;
    const float *LW;          // log-softmax weights (exact fallback)
    float *out;
    int64_t B;
    int R, S, tiles, ppb;     // S = outputs per partition (sum) / classes (root); ppb = partitions per block (sum)
};

constexpr float kUpLn2 = 0.6931471805599453f, kUpL2E = 1.4426950408889634f;

// ---- sum layer ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void prodsum_mfma_kernel(const UpperArgs a) {
    constexpr int OPT = 32 / N;                 // outputs per tile: 2 (N = 16) or 4 (N = 8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int P = a.R / 2, S = a.S;
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + s;
    const bool row_ok = b_raw < a.B;
    const int64_t b = row_ok ? b_raw : a.B - 1;
    const int p0 = blockIdx.y * a.ppb, p1 = min(P, p0 + a.ppb);
    extern __shared__ float upper_res[];                 // [4 waves][32][S + 1]: a wave's outputs on their way to whole rows
    const int SP = S + 1;                                // (row stride: the 32 samples of a half-wave hit 32 banks)
    float *res_l = upper_res + wave * 32 * SP;
    const int64_t b0w = ((int64_t)blockIdx.x * 4 + wave) * 32;
    for (int p = p0; p < p1; ++p) {
        const float *xa = a.in + (b * a.R + 2 * p) * N;
        float av[N], cv[N];
        upper_load<N>(xa, av, cv);
        const float ma = upper_max<N>(av), mc = upper_max<N>(cv);
        float ea[N];
#pragma unroll
        for (int i = 0; i < N; ++i) ea[i] = __builtin_amdgcn_exp2f((av[i] - ma) * kUpL2E);
        // B fragment: K slot (hg = h, el) <-> j = 8 h + el (N = 16); N = 8: the upper half of K is unused
        float eb[8];
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            float cj;
            if (N == 16) cj = h ? cv[8 + el] : cv[el]; else cj = cv[el];
            const float e = __builtin_amdgcn_exp2f(fmaf(cj - mc, kUpL2E, kUpScaleLog2));   // e^{c - max c} * 2^15
            eb[el] = (N == 8 && h) ? 0.f : e;
        }
        half8 eh, el8;
        split8(eb, eh, el8);
        const ug_h8 *fp = (ug_h8 *)(a.frag + (int64_t)p * a.tiles * 1024) + lane;
        bool vanished = false, vmask_over = false;
        unsigned long long vmask = 0ull;   // outputs of this lane's sample whose scaled sum vanished
        // Tile loop, two tiles per trip: the fragments of tile t + 1 are requested before the MFMAs of tile t, into a
        // second register set (a copy between the sets would make hipcc wait in the trip that issued the request).  The
        // outputs go to a wave-private LDS tile and leave as whole rows after the loop.  What this replaces (round 4): a
        // request AND a store per tile inside the loop -- on this hardware a store counts in vmcnt like a load, so every
        // tile waited for the previous tile's store to complete and for its own L2 round trip -- and, when unrolled, a
        // straight-line body of 4000 instructions that each wave executes once (the instruction fetch of a cold kernel
        // was most of its 15-22 us at B = 4096).
        const int T = a.tiles;
        auto tile_body = [&](int t, const half8 &wh, const half8 &wl) {
            gf32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, eh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, el8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, eh, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < OPT / 2; ++q) {   // the lane's outputs of this tile
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) v = fmaf(ea[i], acc[q * N + i], v);
                v *= kUpUnscale;
                const int o = (N == 16) ? 2 * t + h : 4 * t + 2 * h + q;
                if (v < kUpExactBelow) {
                    vanished = true;
                    if (o < 64) vmask |= 1ull << o;
                    else vmask_over = true;
                }
                if (o < S) res_l[s * SP + o] = fmaf(__builtin_amdgcn_logf(v), kUpLn2, ma + mc);
            }
        };
        half8 ah = fp[0], al = fp[64], bh, bl;
#pragma clang loop unroll(disable)
        for (int t = 0; t < T; t += 2) {
            {
                const int tn = min(t + 1, T - 1);
                bh = fp[tn * 128];
                bl = fp[tn * 128 + 64];
            }
            tile_body(t, ah, al);
            if (t + 1 < T) {
                const int tn = min(t + 2, T - 1);
                ah = fp[tn * 128];
                al = fp[tn * 128 + 64];
                tile_body(t + 1, bh, bl);
            }
        }
        // the wave's [32 samples][S] outputs of this partition: whole rows (S floats = 64 bytes at S = 16) per sample
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private tile: no barrier, only the wave's own writes)
        for (int e = lane; e < 32 * S; e += 64) {
            const int rs = e / S, o = e - rs * S;
            if (b0w + rs < a.B) a.out[((b0w + rs) * P + p) * S + o] = res_l[rs * SP + o];
        }
        if (__any(vanished)) {   // rare: the flagged (sample, output) nodes are redone exactly, by the whole wave each
            {
                // (outputs beyond the 64-bit mask: such a sample redoes all of its outputs)
                unsigned long long lanes = __ballot((vmask != 0ull || vmask_over) && row_ok);
                while (lanes) {
                    const int src = __builtin_ctzll(lanes);
                    lanes &= lanes - 1;
                    const int64_t bs = __shfl(b, src, 64);
                    unsigned long long om = __shfl(vmask, src, 64);
                    const bool all = __shfl((int)vmask_over, src, 64) != 0;
                    const float *xs = a.in + (bs * a.R + 2 * p) * N;
                    const int hs = src >> 5;
                    for (int o = 0; o < S; ++o) {
                        const int owner = (N == 16) ? (o & 1) : ((o >> 1) & 1);   // lane half that computed output o
                        if (!(all ? owner == hs : (o < 64 && ((om >> o) & 1ull)))) continue;
                        float m, sum;
                        upper_exact_ms_wave<N>(xs, a.LW + ((int64_t)p * S + o) * N * N, lane, m, sum);
                        if (lane == src) a.out[(bs * P + p) * S + o] = (m > -INFINITY) ? m + logf(sum) : -INFINITY;
                    }
                }
            }
        }
    }
}

// ---- root layer: out[b, c] = logsumexp over (p, i, j); classes c <= CT * 32 / N -------------------------------------
template <int N, int CT>
__global__ __launch_bounds__(256) void prodroot_mfma_kernel(const UpperArgs a) {
    constexpr int CPT = 32 / N;                // classes per tile
    constexpr int HALF = N / 2;                // i's of a class held by one lane half
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int P = a.R / 2, C = a.S;
    const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + s;
    const bool row_ok = b_raw < a.B;
    const int64_t b = row_ok ? b_raw : a.B - 1;
    float rm[CT * CPT], rs[CT * CPT];
#pragma unroll
    for (int k = 0; k < CT * CPT; ++k) {
        rm[k] = -INFINITY;
        rs[k] = 0.f;
    }
    bool vanished = false;
    unsigned cmask = 0u;   // classes of this lane's sample with a vanished partition share (CT * CPT <= 32)
    // Partition p + 1's inputs and fragments are requested before the arithmetic of partition p, into a SECOND register
    // set (two partitions per trip: a copy between the sets makes hipcc wait for the request in the trip that issued it).
    // A wave used to pay two dependent L2 round trips per partition: 18 us for the 8 partitions of the (16,16) root at
    // B = 4096 (round 4).
    struct Operands {
        float av[N], cv[N];
        half8 fh[CT], fl[CT];
    };
    auto request = [&](int p, Operands &o) {
        const int pc = min(p, P - 1);
        upper_load<N>(a.in + (b * a.R + 2 * pc) * N, o.av, o.cv);
        const ug_h8 *fp = (ug_h8 *)(a.frag + (int64_t)pc * CT * 1024) + lane;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            o.fh[t] = fp[t * 128];
            o.fl[t] = fp[t * 128 + 64];
        }
    };
    auto partition = [&](const Operands &o) {
        const float ma = upper_max<N>(o.av), mc = upper_max<N>(o.cv);
        // this lane's rows of a tile are i = (u & 3) + 8 ((u >> 2) & 1) + 4 h (N = 16) / (u & 3) + 4 h (N = 8)
        float es[HALF];
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
            const int i0 = (N == 16) ? (k & 3) + 8 * (k >> 2) : k;
            const float ai = h ? o.av[i0 + 4] : o.av[i0];
            es[k] = __builtin_amdgcn_exp2f((ai - ma) * kUpL2E);
        }
        float eb[8];
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            float cj;
            if (N == 16) cj = h ? o.cv[8 + el] : o.cv[el]; else cj = o.cv[el];
            const float e = __builtin_amdgcn_exp2f(fmaf(cj - mc, kUpL2E, kUpScaleLog2));   // e^{c - max c} * 2^15
            eb[el] = (N == 8 && h) ? 0.f : e;
        }
        half8 eh, el8;
        split8(eb, eh, el8);
        const float m = ma + mc;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const half8 wh = o.fh[t], wl = o.fl[t];
            gf32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, eh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, el8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, eh, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < HALF; ++k) v = fmaf(es[k], acc[q * HALF + k], v);
                v += __shfl_xor(v, 32, 64);              // the other half of the i's
                v *= kUpUnscale;
                const int k = t * CPT + q;
                if (v < kUpExactBelow && k < C && m > -INFINITY) {
                    vanished = true;
                    cmask |= 1u << k;
                }
                // running (max, scaled sum) over the partitions
                const float mm = fmaxf(rm[k], m);
                const float mm0 = (mm == -INFINITY) ? 0.f : mm;
                rs[k] = rs[k] * __builtin_amdgcn_exp2f((rm[k] - mm0) * kUpL2E) + v * __builtin_amdgcn_exp2f((m - mm0) * kUpL2E);
                rm[k] = mm;
            }
        }
    };
    Operands opa, opb;
    request(0, opa);
    for (int p = 0; p < P; p += 2) {
        request(p + 1, opb);
        partition(opa);
        if (p + 1 < P) {
            request(p + 2, opa);
            partition(opb);
        }
    }
    // results of the fast path; then, rare, the flagged (sample, class) roots are redone exactly over all partitions, by
    // the whole wave each (upper_exact_ms_wave)
#pragma unroll
    for (int k = 0; k < CT * CPT; ++k) {
        if (k < C) {
            const float r = (rm[k] > -INFINITY) ? fmaf(__builtin_amdgcn_logf(rs[k]), kUpLn2, rm[k]) : -INFINITY;
            if (row_ok && h == 0) a.out[b * C + k] = r;
        }
    }
    if (__any(vanished)) {
        cmask |= (unsigned)__shfl_xor((int)cmask, 32, 64);          // (both lane halves of a sample saw the same sums)
        unsigned long long lanes = __ballot(cmask != 0u && row_ok && h == 0);
        while (lanes) {
            const int src = __builtin_ctzll(lanes);
            lanes &= lanes - 1;
            const int64_t bs = __shfl(b, src, 64);
            unsigned cm = (unsigned)__shfl((int)cmask, src, 64);
            while (cm) {
                const int k = __builtin_ctz(cm);
                cm &= cm - 1;
                float mm = -INFINITY, ss = 0.f;
                for (int p = 0; p < P; ++p) {
                    float pm, ps;
                    upper_exact_ms_wave<N>(a.in + (bs * a.R + 2 * p) * N, a.LW + ((int64_t)k * P + p) * N * N, lane, pm, ps);
                    if (ps > 0.f && pm > -INFINITY) {
                        if (pm > mm) {
                            ss = ss * expf(mm - pm) + ps;
                            mm = pm;
                        } else {
                            ss += ps * expf(pm - mm);
                        }
                    }
                }
                if (lane == src) a.out[bs * C + k] = (mm > -INFINITY) ? mm + logf(ss) : -INFINITY;
            }
        }
    }
}

static int root_ct(int N, int C) {   // class tiles of the root kernel: 1, 2, 4 or 8
    const int t = cdiv((int64_t)C * N, 32);
    return t <= 1 ? 1 : (t <= 2 ? 2 : (t <= 4 ? 4 : 8));
}
bool upper_mfma_shape_ok(bool root, int N, int S) {
    if (!(N == 8 || N == 16)) return false;
    if (root) return cdiv((int64_t)S * N, 32) <= 8;
    return (S * N) % 32 == 0;
}
int64_t upper_mfma_frag_bytes(int R, int N, int S) {   // (covers the sum and the root layout)
    const int64_t t = cdiv((int64_t)S * N, 32);
    return align_up((int64_t)(R / 2) * (t > 8 ? t : 8) * 2048, 256);
}

// W / LW: linear and log softmax weights (already computed by the caller), frag: upper_mfma_frag_bytes() of scratch
// softmax rows + fragments of a layer from its raw weights (the per-call rebuild of the folded route)
int upper_mfma_tables(bool root, const float *weight, float *W, float *LW, int R, int N, int S, void *frag, hipStream_t st) {
    const int P = R / 2, tiles = root ? root_ct(N, S) : cdiv((int64_t)S * N, 32);
    const int rows = root ? S : P * S, n = root ? P * N * N : N * N;
    const int extra = (root && S * N < tiles * 32) ? 1 : 0;
    DPK_LAUNCH(upper_tables_kernel, dim3(rows + extra), dim3(256), 0, st, weight, rows, n, W, LW, (uint16_t *)frag, P, N, S,
               root ? 1 : 0, tiles);
    DPK_CHECK_LAUNCH("upper_tables_kernel");
    return DPK_OK;
}

static UpperTabJob upper_tab_job(bool root, const float *weight, float *W, float *LW, int R, int N, int S, void *frag) {
    UpperTabJob j{};
    const int P = R / 2;
    j.w = weight; j.W = W; j.LW = LW; j.frag = (uint16_t *)frag; j.P = P; j.N = N; j.S = S; j.root = root ? 1 : 0;
    j.tiles = root ? root_ct(N, S) : cdiv((int64_t)S * N, 32);
    j.rows = root ? S : P * S;
    j.n = root ? P * N * N : N * N;
    j.blocks = j.rows + ((root && S * N < j.tiles * 32) ? 1 : 0);
    return j;
}
int upper_mfma_tables_pair(const float *w0, float *W0, float *LW0, int R0, int N0, int S0, void *frag0, const float *w1,
                           float *W1, float *LW1, int R1, int N1, int C, void *frag1, hipStream_t st) {
    const UpperTabJob j0 = upper_tab_job(false, w0, W0, LW0, R0, N0, S0, frag0), j1 = upper_tab_job(true, w1, W1, LW1, R1, N1, C, frag1);
    DPK_LAUNCH(upper_tables_pair_kernel, dim3(j0.blocks + j1.blocks), dim3(256), 0, st, j0, j1);
    DPK_CHECK_LAUNCH("upper_tables_pair_kernel");
    return DPK_OK;
}

int upper_mfma_forward(bool root, const float *in, const float *W, const float *LW, int64_t B, int R, int N, int S,
                       float *out, void *frag, bool frag_cached, hipStream_t st, const unsigned *gate) {
    const int P = R / 2, tiles = root ? root_ct(N, S) : cdiv((int64_t)S * N, 32);
    if (!frag_cached)
        DPK_LAUNCH(upper_pack_kernel, dim3(cdiv((int64_t)P * tiles * 64, 256)), dim3(256), 0, st, W, P, N, S,
                       root ? 1 : 0, tiles, (uint16_t *)frag, gate);
    UpperArgs a{};
    a.in = in; a.frag = (const uint16_t *)frag; a.LW = LW; a.out = out; a.B = B; a.R = R; a.S = S; a.tiles = tiles;
    const int gx = cdiv(B, 128);
    if (!root) {
        // partitions per block: a wave's partitions run one after the other, each behind its own L2 round trip, so they
        // are spread over the grid until it is a few waves deep per SIMD (DPK_UPPER_BLOCKS overrides: measurements)
        static const int64_t want = [] { const char *e = getenv("DPK_UPPER_BLOCKS"); return e ? (int64_t)atoll(e) : (int64_t)8192; }();
        int ppb = P;
        while (ppb > 1 && (int64_t)gx * cdiv(P, ppb) < want) ppb = (ppb + 1) / 2;
        a.ppb = ppb;
        const dim3 grid(gx, cdiv(P, ppb));
        const size_t lds = (size_t)4 * 32 * (S + 1) * 4;
        if (N == 16) DPK_LAUNCH(prodsum_mfma_kernel<16>, grid, dim3(256), lds, st, a);
        else DPK_LAUNCH(prodsum_mfma_kernel<8>, grid, dim3(256), lds, st, a);
    } else {
#define DPK_ROOT(NN, CTT) DPK_LAUNCH((prodroot_mfma_kernel<NN, CTT>), dim3(gx), dim3(256), 0, st, a)
        if (N == 16) {
            if (tiles <= 1) DPK_ROOT(16, 1); else if (tiles <= 2) DPK_ROOT(16, 2); else if (tiles <= 4) DPK_ROOT(16, 4);
            else DPK_ROOT(16, 8);
        } else {
            if (tiles <= 1) DPK_ROOT(8, 1); else if (tiles <= 2) DPK_ROOT(8, 2); else if (tiles <= 4) DPK_ROOT(8, 4);
            else DPK_ROOT(8, 8);
        }
#undef DPK_ROOT
    }
    DPK_CHECK_LAUNCH("upper_mfma_forward");
    return DPK_OK;
}

}  // namespace dpk
