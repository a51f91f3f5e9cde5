// RealNVP-1D coupling with the reference's alternating masks (coupling.py:58-60: mask = arange(D) % 2, swapped
// for every other layer), depth-1 conditioner, on the f16 matrix cores with fp32-grade products.
//
// reference: CouplingLayer1d.apply_backward / apply_forward, deeprob/flows/layers/coupling.py:72-104
//
//   z = W2 relu(W1 (mask*x) + b1) + b2 ;  t, s = chunk(z) ; s = a tanh(s) ; t, s *= inv_mask
//   density direction:  u = (x - t) exp(-s),  ildj = -sum_d s ;   sampling direction:  x = u exp(s) + t
//
// Both GEMMs run on v_mfma_f32_32x32x16_f16 with every operand split in two f16 halves (v = vh + vl, 11 + 11
// significant bits while the low half is a normal f16 number, i.e. for |v| >= 0.25; below that the split is exact to
// 3e-8 ABSOLUTE -- |dz| <= sum_k |h_k| 3e-8 ~ 4e-6 at worst for this conditioner, 1e-7 of a log-likelihood;
// a b ~= ah bh + ah bl + al bh, fp32 accumulation): the product keeps >= 22 bits -- the
// accuracy class of the fp32 MFMA this replaces (coupling.hip, still used for other masks) -- at 3/16 of its cost,
// which turns the layer from matrix-core-bound (0.40 of the fp32 MFMA peak, 311 us per 65536 x 784 layer) into a
// stream of x, out and the L2-resident weight tables.
//
// Mapping (as ratspn_gemm.hip: 4 compute waves + 4 loader waves, 3-stage LDS ring, one s_barrier per chunk):
//  * a work-group owns 128 samples, a compute wave 32 of them and ALL hidden units: GEMM 1 is computed transposed,
//    H^T = W1m X^T, so lane (sample s = l & 31, half h = l >> 5) ends up holding 16 of every 32 hidden units of its
//    own sample.  The K order of an MFMA is free as long as both operands agree, so those accumulator registers ARE
//    the B fragments of GEMM 2 (Z^T = W2 H^T) once bias + ReLU + split are applied: the hidden activations never
//    leave the registers, the W2 fragments are packed in that unit order.
//  * phase 1 chunks: 64 raw columns of x (the 32 masked ones feed the MFMAs) + the matching W1 fragments (16 KB);
//    phase 2 chunks: the W2 fragments of 32 transformed variables (t and s rows, hi and lo: 32 KB) + their biases
//    and the folded input affine.  Lane (s, h) then holds t and s of 16 of those 32 variables for its sample,
//    reads the (pass-through, transformed) column pairs as 32-byte runs, applies the epilogue and writes the pairs.
//  * an eval-mode BatchNormLayer1d in front (in_scale / in_shift) is folded into W1 / b1 when the tables are packed
//    and applied to the pairs in the epilogue.
#include "common.h"
#include "ratspn_gemm_common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace dpk {

constexpr int kX3Tile = kGemmTile;     // 128 samples per work-group
constexpr int kX3XB = kX3Tile * 256;   // x chunk: 64 raw columns per row

struct X3Geom {
    int NU, NCH1, NPT, K1, N2, W1CH, W2CH;   // hidden tiles, phase-1 / phase-2 chunks, table chunk bytes
};
static inline X3Geom x3_geom(int D, int U) {
    X3Geom g{};
    g.NU = U / 32;
    g.K1 = D / 2;
    g.N2 = D / 2;
    g.NCH1 = cdiv(D, 64);
    g.NPT = cdiv(g.N2, 32);
    g.W1CH = 2 * g.NU * 2 * 1024;                                 // 2 K-steps x NU tiles x (hi, lo) KiB
    g.W2CH = (int)align_up((int64_t)(U / 16) * 4 * 1024 + 1024, 4096);  // K-steps x (t,s) x (hi,lo) KiB + extras
    return g;
}

// ------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------
struct X3PackArgs {
    const float *W1, *b1, *W2, *b2, *in_scale, *in_shift;
    int D, U, pm;        // pm: parity of the masked (conditioning) columns; the transformed ones are 1 - pm
    int affine;
    X3Geom g;
    uint16_t *w1t;       // [NCH1][2][NU][2][512]
    char *w2t;           // [NPT][W2CH bytes]
    float *b1f;          // [U] b1 + W1 (mask * in_shift)
    float *scales;       // [2] power-of-two scales of the W1 / W2 tables (largest entry -> [2^12, 2^13))
    const unsigned *gate;   // table kernels return at once while *gate == 0 (common.h: params_gate)
};

// The f16 halves of a split are exact to 2^-22 relative only while the low half is a normal f16 number (|v| >= 0.25);
// conditioner weights are a few hundredths, so unscaled they carry an ABSOLUTE error of 3e-8 each -- invisible on
// standardised data, 1e-3 relative on a log-likelihood once BatchNorm scales of 100 and |x| of 30 multiply it (measured
// against an fp64 evaluation of the reference formulas).  Each table is therefore multiplied by the power of two that puts its largest entry into
// [2^12, 2^13); the kernel divides the accumulators by it again.
//
// One pass over the parameters does two things (round 3): the maxima behind those scales, and a fingerprint of every byte
// the tables depend on -- the used columns of W1 with the folded input affine, b1, the used rows of W2 and b2.  The last
// work-group to finish compares the fingerprint with the one the tables were built from and opens the gate of the pack
// kernel only if it differs (DPK_FLAG_PARAMS_VERIFY; an unconditional build opens it always): a verifying call costs
// this pass + an empty pack launch instead of the generic fingerprint + scale + pack launches.
struct X3Check {   // 32 bytes in the workspace, zeroed by the unconditional build
    unsigned long long stored, acc;
    unsigned m1, m2, tickets, gate;
};
constexpr int kX3CheckBlocks = 48;
// (bid / nblocks: the block's place among the blocks that share this layer -- the whole grid for the per-layer launch,
// a slice of it for dpk_coupling1d_pairs_tables)
__device__ __forceinline__ void x3_check_body(const X3PackArgs &a, X3Check *st, int verify, int bid, int nblocks) {
    __shared__ float red[2][4];
    __shared__ unsigned long long hred[4];
    const int D = a.D, U = a.U, K1 = a.g.K1, N2 = a.g.N2;
    // (32-bit indices: U K1 and 2 N2 U are a few hundred thousand; a 64-bit division per element tripled this pass)
    const int gtid = (int)(bid * blockDim.x + threadIdx.x), gsz = (int)(nblocks * blockDim.x);
    float m1 = 0.f, m2 = 0.f;
    unsigned long long h = 0ull;
    // (four elements per trip, their loads requested together: one load per trip left this pass latency bound)
    const int n1 = U * K1;
    for (int e0 = gtid; e0 < n1; e0 += 4 * gsz) {
        float v[4], sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q * gsz, n1 - 1);
            const int unit = e / K1, col = 2 * (e - unit * K1) + a.pm;
            v[q] = a.W1[unit * D + col];
            sc[q] = a.in_scale ? a.in_scale[col] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * gsz;
            if (e < n1) {
                h += fp_word(__float_as_uint(v[q]), (unsigned)e * 8u + 1u);
                m1 = fmaxf(m1, fabsf(v[q] * sc[q]));
            }
        }
    }
    const int rows2 = a.affine ? 2 : 1;
    const int n2 = rows2 * N2 * U;
    for (int e0 = gtid; e0 < n2; e0 += 4 * gsz) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q * gsz, n2 - 1);
            const int r = e / U, unit = e - r * U;
            const int ts = r / N2, var = 2 * (r - ts * N2) + (1 - a.pm);
            v[q] = a.W2[(ts * D + var) * U + unit];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * gsz;
            if (e < n2) {
                h += fp_word(__float_as_uint(v[q]), (unsigned)e * 8u + 2u);
                m2 = fmaxf(m2, fabsf(v[q]));
            }
        }
    }
    for (int e = gtid; e < U; e += gsz) h += fp_word(__float_as_uint(a.b1[e]), (unsigned)e * 8u + 3u);
    for (int e = gtid; e < rows2 * D; e += gsz) h += fp_word(__float_as_uint(a.b2[e]), (unsigned)e * 8u + 4u);
    if (a.in_scale)
        for (int e = gtid; e < D; e += gsz)
            h += fp_word(__float_as_uint(a.in_scale[e]), (unsigned)e * 8u + 5u) + fp_word(__float_as_uint(a.in_shift[e]), (unsigned)e * 8u + 6u);
    m1 = wave_reduce_max(m1);
    m2 = wave_reduce_max(m2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += (unsigned long long)__shfl_xor((long long)h, o, 64);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = m1;
        red[1][threadIdx.x >> 6] = m2;
        hred[threadIdx.x >> 6] = h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (non-negative floats order like their bit patterns: an unsigned atomic max is the float max)
        atomicMax(&st->m1, __float_as_uint(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]))));
        atomicMax(&st->m2, __float_as_uint(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]))));
        atomicAdd(&st->acc, (hred[0] + hred[1]) + (hred[2] + hred[3]) + (unsigned long long)(a.pm + 2 * a.affine + 1));
        // (the partial results above are device-scope atomics: their acknowledgement is all the release there is to wait for;
        // a __threadfence() writes back this XCD's L2 -- 17 .. 40 us behind a kernel that left it dirty, round-4 measurement)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (atomicAdd(&st->tickets, 1u) == (unsigned)nblocks - 1u) {   // last block: every partial result has arrived (read back with atomics)
            const unsigned long long sum = atomicExch(&st->acc, 0ull);
            const unsigned b1 = atomicExch(&st->m1, 0u), b2 = atomicExch(&st->m2, 0u);
            const bool rebuild = !verify || sum != st->stored;
            if (rebuild) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float m = __uint_as_float(t ? b2 : b1);
                    // m * scale in [2^12, 2^13); degenerate tables (all zero, inf, NaN) keep scale 1
                    float sc = 1.f;
                    if (m > 0.f && m < 3.0e38f) sc = exp2f(fminf(fmaxf(12.f - floorf(log2f(m)), -60.f), 60.f));
                    a.scales[t] = sc;
                }
            }
            st->stored = sum;
            st->gate = rebuild ? 1u : 0u;
            st->tickets = 0u;
        }
    }
}

__global__ __launch_bounds__(256) void coupling_x3_check_kernel(const X3PackArgs a, X3Check *st, int verify) {
    x3_check_body(a, st, verify, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void x3_pack_body(const X3PackArgs &a, int bid, int nblocks) {
    if (gate_closed(a.gate)) return;
    const int D = a.D, U = a.U, NU = a.g.NU, K1 = a.g.K1, N2 = a.g.N2;
    const int64_t n1 = (int64_t)a.g.NCH1 * 2 * NU * 64;           // W1 fragment entries (hi + lo written together)
    const int64_t n2 = (int64_t)a.g.NPT * (U / 16) * 2 * 64;      // W2 fragment entries
    const int64_t n3 = (int64_t)a.g.NPT * 32;                     // per-variable extras
    const int64_t total = n1 + n2 + n3 + U;
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += (int64_t)nblocks * blockDim.x) {
        if (e < n1) {
            const int l = (int)(e & 63);
            const int64_t r = e >> 6;
            const int T = (int)(r % NU), ksg = (int)(r / NU);      // ksg = 2 * chunk + k-step in chunk
            const int unit = 32 * T + (l & 31), hg = l >> 5;
            half8 vh, vl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 16 * ksg + 8 * hg + i, col = 2 * m + a.pm;
                float v = 0.f;
                if (m < K1 && col < D) {
                    v = a.W1[(int64_t)unit * D + col];
                    if (a.in_scale) v *= a.in_scale[col];
                    v *= a.scales[0];
                }
                _Float16 hi, lo;
                split_f16(v, hi, lo);
                vh[i] = hi; vl[i] = lo;
            }
            uint16_t *o = a.w1t + ((int64_t)ksg * NU + T) * 1024 + l * 8;
            *reinterpret_cast<half8 *>(o) = vh;
            *reinterpret_cast<half8 *>(o + 512) = vl;
        } else if (e < n1 + n2) {
            const int64_t f = e - n1;
            const int l = (int)(f & 63);
            const int64_t r = f >> 6;
            const int ts = (int)(r & 1), kk = (int)((r >> 1) % (U / 16)), pt = (int)((r >> 1) / (U / 16));
            const int n = 32 * pt + (l & 31), hg = l >> 5;
            const int var = 2 * n + (1 - a.pm);
            half8 vh, vl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // K slot (kk, hg, i) <-> the hidden unit a lane of GEMM 1 holds in accumulator register 8*(kk&1) + i
                const int reg = 8 * (kk & 1) + i;
                const int unit = 32 * (kk >> 1) + (reg & 3) + 8 * (reg >> 2) + 4 * hg;
                float v = 0.f;
                if (n < N2 && (ts == 0 || a.affine)) v = a.W2[((int64_t)ts * D + var) * U + unit] * a.scales[1];
                _Float16 hi, lo;
                split_f16(v, hi, lo);
                vh[i] = hi; vl[i] = lo;
            }
            uint16_t *o = reinterpret_cast<uint16_t *>(a.w2t + (int64_t)pt * a.g.W2CH) + ((int64_t)kk * 2 + ts) * 1024 + l * 8;
            *reinterpret_cast<half8 *>(o) = vh;
            *reinterpret_cast<half8 *>(o + 512) = vl;
        } else if (e < n1 + n2 + n3) {
            const int64_t f = e - n1 - n2;
            const int pt = (int)(f >> 5), j = (int)(f & 31);
            const int n = 32 * pt + j, var = 2 * n + (1 - a.pm), par = var ^ 1;
            float *x = reinterpret_cast<float *>(a.w2t + (int64_t)pt * a.g.W2CH + (int64_t)(U / 16) * 4096);
            const bool ok = n < N2;
            x[j] = ok ? a.b2[var] : 0.f;
            x[32 + j] = (ok && a.affine) ? a.b2[D + var] : 0.f;
            x[64 + j] = (ok && a.in_scale) ? a.in_scale[var] : 1.f;
            x[96 + j] = (ok && a.in_shift) ? a.in_shift[var] : 0.f;
            x[128 + j] = (ok && a.in_scale) ? a.in_scale[par] : 1.f;
            x[160 + j] = (ok && a.in_shift) ? a.in_shift[par] : 0.f;
        } else {
            const int unit = (int)(e - n1 - n2 - n3);
            float v = a.b1[unit];
            if (a.in_shift)
                for (int m = 0; m < K1; ++m) {
                    const int col = 2 * m + a.pm;
                    v = fmaf(a.W1[(int64_t)unit * D + col], a.in_shift[col], v);
                }
            a.b1f[unit] = v;
        }
    }
}

__global__ __launch_bounds__(256) void coupling_x3_pack_kernel(const X3PackArgs a) {
    x3_pack_body(a, (int)blockIdx.x, (int)gridDim.x);
}

// Several layers per launch (dpk_coupling1d_pairs_tables): kX3CheckBlocks / kX3PackBlocks blocks per layer.  The
// layers' argument blocks travel as kernel arguments (16 x ~140 bytes).
constexpr int kX3ManyMax = 16;
constexpr int kX3PackBlocks = 96;
struct X3ManyArgs {
    X3PackArgs L[kX3ManyMax];
    X3Check *st[kX3ManyMax];
    int verify[kX3ManyMax];
};
__global__ __launch_bounds__(256) void coupling_x3_check_many_kernel(const X3ManyArgs m) {
    const int l = (int)blockIdx.x / kX3CheckBlocks;
    x3_check_body(m.L[l], m.st[l], m.verify[l], (int)blockIdx.x - l * kX3CheckBlocks, kX3CheckBlocks);
}
__global__ __launch_bounds__(256) void coupling_x3_pack_many_kernel(const X3ManyArgs m) {
    const int l = (int)blockIdx.x / kX3PackBlocks;
    x3_pack_body(m.L[l], (int)blockIdx.x - l * kX3PackBlocks, kX3PackBlocks);
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct X3Args {
    const float *x;
    float *out, *ldj;
    int64_t B;
    int D, U, pm, inverse, accumulate, ntiles;
    X3Geom g;
    const uint16_t *w1t;
    const char *w2t;
    const float *b1f, *act_weight, *scales;
    const float *in_scale, *in_shift;   // the folded input affine per raw column (x-once kernel; null: identity)
    // BASE mode (last coupling of a flow + the affine behind it + the Normal base, dpk_coupling1d_pairs_logprob): the
    // per-column (a_d, c_d) pairs with -t^2 = -(u' - loc)^2 / (2 sigma^2), t = a_d u_d + c_d, their constant, the
    // log-det accumulated so far and the log-likelihoods written instead of `out`
    const float *base_ac;      // [2][D] then [1] constant
    const float *ildj_in;      // [B] or null
    float *ll_out;             // [B]
    long long *dbg;   // measurement only (-DDPK_X3_TIMELINE + DPK_X3_TIMELINE=1): s_memtime stamps of work-group 0
};

__device__ __forceinline__ float x3_tanh(float v) {
    // tanh(v) = (e^{2v} - 1) / (e^{2v} + 1) on the hardware exp2 / rcp (absolute error ~1e-7, as coupling.hip)
    const float c = fminf(fmaxf(v, -15.f), 15.f);
    const float t = __builtin_amdgcn_exp2f(c * 2.8853900817779268f);
    return (t - 1.f) * __builtin_amdgcn_rcpf(t + 1.f);
}

// s_memtime stamps of work-group 0 (row = chunk count of the wave; slots: 0 before the chunk's barrier, 1 after it,
// 2 end of the chunk, 4 phase 2: MFMAs done; loaders: 0 before the counted wait, 1 after it, 2 after the barrier,
// 3 after issuing the next chunk).  Compiled in with -DDPK_X3_TIMELINE only.
#ifdef DPK_X3_TIMELINE
#define X3_STAMP(row, slot)                                                                                  \
    do {                                                                                                     \
        if (a.dbg && blockIdx.x == 0 && lane == 0 && (row) < 64)                                             \
            a.dbg[((wave8 * 64) + (row)) * 8 + (slot)] = (long long)__builtin_readcyclecounter();            \
    } while (0)
#else
#define X3_STAMP(row, slot) do { } while (0)
#endif

template <bool AFFINE, int NU, bool BASE = false>
__global__ __launch_bounds__(2 * kGemmWaves * 64) void coupling_x3_kernel(const X3Args a) {
    constexpr int W1CH = 2 * NU * 2 * 1024;
    constexpr int KK = NU * 2;                                   // K-steps of GEMM 2 (16 hidden units each)
    constexpr int W2CH = ((KK * 4 * 1024 + 1024 + 4095) / 4096) * 4096;
    constexpr int STAGE = (kX3XB + W1CH) > W2CH ? (kX3XB + W1CH) : W2CH;
    constexpr int P1 = 8 + W1CH / (4 * 1024);                    // DMA instructions per loader wave: phase-1 chunk
    constexpr int P2 = W2CH / (4 * 1024);                        //                                   phase-2 chunk
    static_assert(W1CH % 4096 == 0 && P1 <= 63 && P2 <= 63, "chunk split");
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) const half8 lh8;
    typedef const __attribute__((address_space(1))) gf32x4 gf4;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    lfloat *b1_l = (lfloat *)(smem + kGemmStages * STAGE);       // [U]
    lfloat *base_l = b1_l + ((a.U + 3) & ~3);                    // BASE: [2][D] (a_d, c_d), 16-byte aligned (D % 8 == 0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= kGemmWaves;
    const int wave = wave8 & (kGemmWaves - 1);
    const int s = lane & 31, h = lane >> 5;
    const int D = a.D, NCH1 = a.g.NCH1, NPT = a.g.NPT;
    const int grid = (int)gridDim.x, ntiles = a.ntiles;
    const int nchunks = NCH1 + NPT;

    if (loader) {
        const unsigned smem_base = (unsigned)(uintptr_t)smem;
        unsigned voff[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = wave * 32 + j * 4 + (lane >> 4);
            const int gp = (lane & 15) ^ (rl & 15);
            voff[j] = (unsigned)(rl * D + gp * 4) * 4u;
        }
        int ptile = (int)blockIdx.x, pc = 0, pstage = 0;
        auto issue_next = [&]() {
            const unsigned st = smem_base + pstage * STAGE;
            if (pc < NCH1) {
                const int64_t b0 = (int64_t)ptile * kX3Tile;
                const gcchar_p xt = (gcchar_p)a.x + (b0 * D + pc * 64) * 4;
                const bool full = (b0 + kX3Tile <= a.B) && ((pc + 1) * 64 <= D);
                if (full) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) glds16(voff[j], xt, st + (wave * 32 + j * 4) * 256);
                } else {
                    const int nvalid = (int)min((int64_t)kX3Tile, a.B - b0);
                    const int vp = min(16, (D - pc * 64) >> 2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int rl = wave * 32 + j * 4 + (lane >> 4);
                        const int gp = min((lane & 15) ^ (rl & 15), vp - 1);
                        glds16((unsigned)(min(rl, nvalid - 1) * D + gp * 4) * 4u, xt, st + (wave * 32 + j * 4) * 256);
                    }
                }
                const gcchar_p tsrc = (gcchar_p)a.w1t + (int64_t)pc * W1CH;
                const unsigned t0 = (unsigned)(wave * (W1CH / 4));
#pragma unroll
                for (int j = 0; j < W1CH / 4096; ++j)
                    glds16(t0 + j * 1024 + lane * 16, tsrc, st + kX3XB + t0 + j * 1024);
            } else {
                const gcchar_p tsrc = (gcchar_p)a.w2t + (int64_t)(pc - NCH1) * W2CH;
                const unsigned t0 = (unsigned)(wave * (W2CH / 4));
#pragma unroll
                for (int j = 0; j < P2; ++j) glds16(t0 + j * 1024 + lane * 16, tsrc, st + t0 + j * 1024);
            }
            pstage = (pstage + 1 == kGemmStages) ? 0 : pstage + 1;
            if (++pc == nchunks) {
                pc = 0;
                ptile += grid;
            }
        };
#pragma unroll
        for (int g = 0; g < kGemmStages - 1; ++g)
            if (ptile < ntiles) issue_next();
        __syncthreads();
        [[maybe_unused]] int lrow = 0;
        for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
            for (int c = 0; c < nchunks; ++c) {
                X3_STAMP(lrow, 0);
                // chunk c has landed once only the chunk issued after it (if any) is still in flight
                const bool last = (c + 1 == nchunks) && !(tile + grid < ntiles);
                if (last) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if ((c + 1) % nchunks < NCH1) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P1) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P2) : "memory");
                }
                X3_STAMP(lrow, 1);
                gemm_lds_barrier();
                X3_STAMP(lrow, 2);
                if (ptile < ntiles) issue_next();
                X3_STAMP(lrow, 3);
                ++lrow;
            }
        }
        return;
    }
    // ================================================ compute waves =========================================
    for (int e = tid; e < a.U; e += kGemmWaves * 64) b1_l[e] = a.b1f[e];
    if (BASE)
        for (int e = tid; e < 2 * a.D; e += kGemmWaves * 64) base_l[e] = a.base_ac[e];
    const float base_cst = BASE ? a.base_ac[2 * a.D] : 0.f;
    const float act = AFFINE ? a.act_weight[0] : 0.f;
    const float w1sc = a.scales[0], w2sc = a.scales[1];
    __syncthreads();

    const int rl_own = wave * 32 + s;
    const int sw = rl_own & 15;
    // LDS byte offsets of the lane's 16 raw columns of each of the chunk's two K-steps (4 pieces of 16 bytes each)
    unsigned xoff[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) xoff[j][q] = (unsigned)(rl_own * 256 + (((8 * j + 4 * h + q) ^ sw) << 4));
    const unsigned foff = (unsigned)(lane * 16);
    const int pm = a.pm;

    int cstage = 0;
    [[maybe_unused]] int crow = 0;   // timeline row (chunk count of this work-group)
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
        gf32x16 acc[NU];
#pragma unroll
        for (int T = 0; T < NU; ++T)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[T][i] = 0.f;
        // f16 range: the B operands (x, then the hidden activations) are carried per SAMPLE with a power-of-two scale
        // that shrinks whenever a value would leave the f16 range (the split halves are f16: 65504 at most) -- exact,
        // and free for ordinary data (one max + one compare per K-step); without it evidence beyond 6.5e4, or a badly
        // conditioned flow, gave inf - inf = NaN where the reference's fp32 arithmetic stays finite.  A column of the
        // transposed GEMM is one sample, so the scale is a per-lane scalar; the two lanes of a sample share the
        // operand's K slots and therefore agree on it.
        constexpr float kBig = 16384.f;
        float xsc = 1.f;
        // ---- phase 1: H^T = W1m X^T ----------------------------------------------------------------------
        for (int c = 0; c < NCH1; ++c) {
            X3_STAMP(crow, 0);
            gemm_lds_barrier();
            X3_STAMP(crow, 1);
            const lchar *st = smem + cstage * STAGE;
            cstage = (cstage + 1 == kGemmStages) ? 0 : cstage + 1;
            const lchar *tb = st + kX3XB + foff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (c * 64 + j * 32 < D) {
                    float raw[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const gf32x4 p4 = *(lf4 *)(st + xoff[j][q]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) raw[4 * q + i] = p4[i];
                    }
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m = pm ? raw[2 * i + 1] : raw[2 * i];
                        // (columns beyond D hold clamped copies; their W1 entries are zero, the values must be finite)
                        v[i] = (c * 64 + j * 32 + 16 * h + 2 * i < D) ? m : 0.f;
                    }
                    float m8 = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) m8 = fmaxf(m8, fabsf(v[i]));
                    m8 = fmaxf(m8, __shfl_xor(m8, 32, 64));
                    if (__builtin_expect(m8 * xsc > kBig && m8 < 3.0e38f, 0)) {
                        const float f = exp2f(-ceilf(log2f(m8 * xsc * (1.f / kBig))));
#pragma unroll
                        for (int T = 0; T < NU; ++T)
#pragma unroll
                            for (int i = 0; i < 16; ++i) acc[T][i] *= f;
                        xsc *= f;
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= xsc;
                    half8 xh, xl;
                    split8(v, xh, xl);
#pragma unroll
                    for (int T = 0; T < NU; ++T) {
                        const half8 wh = *(lh8 *)(tb + (j * NU + T) * 2048);
                        const half8 wl = *(lh8 *)(tb + (j * NU + T) * 2048 + 1024);
                        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[T], 0, 0, 0);
                        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[T], 0, 0, 0);
                        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[T], 0, 0, 0);
                    }
                }
            }
            X3_STAMP(crow, 2);
            ++crow;
        }
        // ---- bias + ReLU + split: the accumulators become the B fragments of GEMM 2 ---------------------------
        half8 hh[KK], hl[KK];
        const float xinv = 1.f / (xsc * w1sc);     // (powers of two: the operand's scale and the W1 table's)
        float hmax = 0.f;
#pragma unroll
        for (int T = 0; T < NU; ++T)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int unit = 32 * T + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                acc[T][reg] = fmaxf(fmaf(acc[T][reg], xinv, b1_l[unit]), 0.f);
                hmax = fmaxf(hmax, acc[T][reg]);
            }
        hmax = fmaxf(hmax, __shfl_xor(hmax, 32, 64));
        float hsc = 1.f;
        if (__builtin_expect(hmax > kBig && hmax < 3.0e38f, 0)) hsc = exp2f(-ceilf(log2f(hmax * (1.f / kBig))));
        const float hinv = 1.f / (hsc * w2sc);
#pragma unroll
        for (int T = 0; T < NU; ++T)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[T][8 * j + i] * hsc;
                split8(v, hh[2 * T + j], hl[2 * T + j]);
            }
        // ---- phase 2: Z^T = W2 H^T per tile of 32 transformed variables, fused epilogue ------------------------
        const int64_t b = (int64_t)tile * kX3Tile + rl_own;
        const bool row_ok = b < a.B;
        const float *xrow = a.x + (row_ok ? b : a.B - 1) * D;
        float *orow = a.out + (row_ok ? b : a.B - 1) * D;
        float ssum = 0.f, bacc = 0.f;
        for (int pt = 0; pt < NPT; ++pt) {
            X3_STAMP(crow, 0);
            gemm_lds_barrier();
            X3_STAMP(crow, 1);
            const lchar *st = smem + cstage * STAGE;
            cstage = (cstage + 1 == kGemmStages) ? 0 : cstage + 1;
            const lchar *tb = st + foff;
            // the lane's pairs: accumulator register r <-> variable n = 32 pt + (r & 3) + 8 (r >> 2) + 4 h; the four
            // registers of a group are four consecutive variables = 8 consecutive raw columns
            gf32x4 xin[4][2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n0 = 32 * pt + 8 * g4 + 4 * h;
                const int col = min(2 * n0, D - 8);   // (a group beyond the last variable re-reads the last run: unused)
                xin[g4][0] = *(gf4 *)(xrow + col);
                xin[g4][1] = *(gf4 *)(xrow + col + 4);
            }
            gf32x16 zt, zs;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                zt[i] = 0.f;
                zs[i] = 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const half8 th = *(lh8 *)(tb + (kk * 2 + 0) * 2048);
                const half8 tl = *(lh8 *)(tb + (kk * 2 + 0) * 2048 + 1024);
                zt = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, hh[kk], zt, 0, 0, 0);
                if (AFFINE) {
                    const half8 sh8 = *(lh8 *)(tb + (kk * 2 + 1) * 2048);
                    zs = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh8, hh[kk], zs, 0, 0, 0);
                }
                zt = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, hl[kk], zt, 0, 0, 0);
                if (AFFINE) {
                    const half8 sh8 = *(lh8 *)(tb + (kk * 2 + 1) * 2048);
                    zs = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh8, hl[kk], zs, 0, 0, 0);
                }
                zt = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, hh[kk], zt, 0, 0, 0);
                if (AFFINE) {
                    const half8 sl8 = *(lh8 *)(tb + (kk * 2 + 1) * 2048 + 1024);
                    zs = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl8, hh[kk], zs, 0, 0, 0);
                }
            }
            X3_STAMP(crow, 4);
            const lchar *ex = st + KK * 4096;   // extras: bt, bs, sc_t, sh_t, sc_p, sh_p (32 floats each)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j0 = 8 * g4 + 4 * h;                  // first of the group's four variables in the tile
                const int n0 = 32 * pt + j0;
                const gf32x4 bt = *(lf4 *)(ex + (0 + j0) * 4), bs = *(lf4 *)(ex + (32 + j0) * 4);
                const gf32x4 sct = *(lf4 *)(ex + (64 + j0) * 4), sht = *(lf4 *)(ex + (96 + j0) * 4);
                const gf32x4 scp = *(lf4 *)(ex + (128 + j0) * 4), shp = *(lf4 *)(ex + (160 + j0) * 4);
                gf32x4 o[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g4 + i;
                    // raw columns 2 (n0 + i) + {0, 1}: element 2 i + (1 - pm) is transformed, 2 i + pm passes through
                    const float e0 = xin[g4][(2 * i) >> 2][(2 * i) & 3], e1 = xin[g4][(2 * i + 1) >> 2][(2 * i + 1) & 3];
                    const float xt = pm ? e0 : e1, xp = pm ? e1 : e0;
                    const float xv = fmaf(xt, sct[i], sht[i]);
                    const float pv = fmaf(xp, scp[i], shp[i]);
                    const float tv = fmaf(zt[r], hinv, bt[i]);
                    float ov;
                    if (AFFINE) {
                        const float sv = act * x3_tanh(fmaf(zs[r], hinv, bs[i]));
                        const float es = __builtin_amdgcn_exp2f((a.inverse ? sv : -sv) * 1.4426950408889634f);
                        ov = a.inverse ? fmaf(xv, es, tv) : (xv - tv) * es;
                        if (n0 + i < a.g.N2) ssum += sv;
                    } else {
                        ov = a.inverse ? xv + tv : xv - tv;
                    }
                    const float lo = pm ? ov : pv, hi = pm ? pv : ov;
                    o[(2 * i) >> 2][(2 * i) & 3] = lo;
                    o[(2 * i + 1) >> 2][(2 * i + 1) & 3] = hi;
                }
                if (BASE) {
                    // the Normal base on the 8 raw columns of the group: -t^2 with t = a_d u_d + c_d (no store of u)
                    if (2 * n0 < D) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            if (2 * n0 + 4 * q < D) {
                                const gf32x4 ba = *(lf4 *)(base_l + 2 * n0 + 4 * q), bc = *(lf4 *)(base_l + D + 2 * n0 + 4 * q);
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float t = fmaf(o[q][i], ba[i], bc[i]);
                                    bacc = fmaf(-t, t, bacc);
                                }
                            }
                        }
                    }
                } else if (row_ok && 2 * n0 < D) {
                    *reinterpret_cast<gf32x4 *>(orow + 2 * n0) = o[0];
                    if (2 * n0 + 4 < D) *reinterpret_cast<gf32x4 *>(orow + 2 * n0 + 4) = o[1];
                }
            }
            X3_STAMP(crow, 2);
            ++crow;
        }
        // ---- log-det: the two lanes of a sample hold disjoint halves of the transformed variables -----------------
        const float tot = ssum + __shfl_xor(ssum, 32, 64);
        if (BASE) {
            const float btot = bacc + __shfl_xor(bacc, 32, 64);
            if (h == 0 && row_ok)
                a.ll_out[b] = btot + base_cst + (a.ildj_in ? a.ildj_in[b] : 0.f) + (AFFINE ? -tot : 0.f);
        } else if (h == 0 && row_ok) {
            const float v = AFFINE ? (a.inverse ? tot : -tot) : 0.f;
            if (a.accumulate) a.ldj[b] += v; else a.ldj[b] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// round 3: the same layer with x read ONCE (coupling_x1_kernel)
// ------------------------------------------------------------------------------------------------
// coupling_x3_kernel reads every row twice (phase 1 through the ring, phase 2 as column pairs from global memory) and runs
// its MFMAs and its epilogue in series on the same four waves while the four loader waves idle.  This kernel keeps the
// tables and the fragment mapping and changes the roles:
//  * a work-group owns 64 samples.  Four HOLDER waves copy every landed x chunk from the ring into registers, input
//    affine applied (a lane keeps 16 raw columns of one sample per chunk: 13 chunks x 16 = 208 VGPRs hold the 64 x 832
//    tile between them -- the one place on the CU with room for it: 200 KB against 160 KB of LDS), and run the whole
//    epilogue in phase 2 from those registers: no second read of x.
//  * four MFMA waves = (sample block of 32) x (role 0 / 1).  In phase 1 a role computes every other 32-unit block of the
//    hidden layer; the two roles of a sample block swap their B fragments through LDS once per tile.  In phase 2 role 0
//    runs the t rows and role 1 the s rows of GEMM 2 and hands z = acc / scale + bias to the holders through a
//    double-buffered 18 KB LDS slab; the holders work on variable tile p - 1 while the MFMA waves run tile p.
//    One s_barrier per chunk, as before, plus one per tile behind the fragment swap.
//  * the ring's LDS-DMA is issued by whoever has the time: by the holders during phase-1 steps (they only copy 256 bytes
//    per lane there), by the MFMA waves during phase-2 steps (they have no other vector-memory traffic, and the holders'
//    stores would spoil a counted wait).  A chunk's issuer waits for it (counted vmcnt) before the chunk's barrier.
constexpr int kX1Tile = 64;
constexpr int kX1XB = kX1Tile * 256;    // x chunk: 64 raw columns of 64 rows
constexpr int kX1ZRow = 36 * 4;         // bytes per sample row of a z slab (32 variables + 16 bytes: conflict-free b128 writes)
constexpr int kX1ZSlab = kX1Tile * kX1ZRow;
#ifndef DPK_X1_NT
#define DPK_X1_NT 1
#endif
constexpr bool kX1XNonTemporal = DPK_X1_NT != 0;   // x is read once: non-temporal LDS-DMA (the L2 keeps the tables)
constexpr int kX1SwapWave = 8192;       // fragment-swap bytes per MFMA wave (<= 2 blocks x 2 K-steps x (hi, lo) KiB)

template <int NU>
struct X1Cfg {
    static constexpr int W1CH = 2 * NU * 2 * 1024;
    static constexpr int KK = NU * 2;
    static constexpr int W2CH = ((KK * 4 * 1024 + 1024 + 4095) / 4096) * 4096;
    static constexpr int STAGE = (kX1XB + W1CH) > W2CH ? (kX1XB + W1CH) : W2CH;
    static constexpr int P1 = 4 + W1CH / (4 * 1024);   // DMA instructions per issuing wave: phase-1 chunk
    static constexpr int P2 = W2CH / (4 * 1024);       //                                    phase-2 chunk
    static_assert(W1CH % 4096 == 0 && P1 <= 63 && P2 <= 63, "chunk split");
};

// N (<= 4) consecutive 1-KiB pieces of a table with ONE M0 setting: the instruction offset advances the global and the
// LDS address alike (glds16 per piece costs five scalar instructions around every load; a wave issues one instruction per
// four cycles, and the issuing waves have none to spare).
template <int N>
__device__ __forceinline__ void x1_glds_run(unsigned voff, gcchar_p sbase_in, unsigned lds_dst_in) {
    static_assert(N >= 1 && N <= 4, "instruction offset field");
    const uint64_t sb = (uint64_t)(uintptr_t)sbase_in;
    const uint64_t sbase = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32)) << 32) |
                           (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_in);
    unsigned keep;
    if constexpr (N == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void x1_glds_table(unsigned voff, gcchar_p sbase, unsigned lds_dst) {
    if constexpr (N >= 4) {
        x1_glds_run<4>(voff, sbase, lds_dst);
        if constexpr (N > 4) x1_glds_table<N - 4>(voff, sbase + 4096, lds_dst + 4096);
    } else if constexpr (N >= 1) {
        x1_glds_run<N>(voff, sbase, lds_dst);
    }
}

// Issue this wave's quarter of chunk (tile, c) into ring stage `stage` (wave = 0..3 within its group of four).
template <int NU>
__device__ __forceinline__ void x1_issue(const X3Args &a, int tile, int c, int stage, int wave, int lane, unsigned smem_base,
                                         const unsigned (&voff)[4]) {
    typedef X1Cfg<NU> C;
    const int D = a.D, NCH1 = a.g.NCH1;
    const unsigned st = smem_base + stage * C::STAGE;
    if (c < NCH1) {
        const int64_t b0 = (int64_t)tile * kX1Tile;
        const gcchar_p xt = (gcchar_p)a.x + (b0 * D + c * 64) * 4;
        const bool full = (b0 + kX1Tile <= a.B) && ((c + 1) * 64 <= D);
        if (full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16<kX1XNonTemporal>(voff[j], xt, st + (wave * 16 + j * 4) * 256);
        } else {   // ragged tile / last chunk: clamp to rows and pieces that exist (clamped slots are never consumed)
            const int nvalid = (int)min((int64_t)kX1Tile, a.B - b0);
            const int vp = min(16, (D - c * 64) >> 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wave * 16 + j * 4 + (lane >> 4);
                const int gp = min((lane & 15) ^ (r & 15), vp - 1);
                glds16<kX1XNonTemporal>((unsigned)(min(r, nvalid - 1) * D + gp * 4) * 4u, xt, st + (wave * 16 + j * 4) * 256);
            }
        }
        const gcchar_p tsrc = (gcchar_p)a.w1t + (int64_t)c * C::W1CH;
        const unsigned t0 = (unsigned)(wave * (C::W1CH / 4));
        x1_glds_table<C::W1CH / 4096>(t0 + lane * 16, tsrc, st + kX1XB + t0);
    } else {
        const gcchar_p tsrc = (gcchar_p)a.w2t + (int64_t)(c - NCH1) * C::W2CH;
        const unsigned t0 = (unsigned)(wave * (C::W2CH / 4));
        x1_glds_table<C::P2>(t0 + lane * 16, tsrc, st + t0);
    }
}

// s_waitcnt vmcnt(n) for the three counts a ring wait can need (0 = nothing younger in flight)
template <int NU>
__device__ __forceinline__ void x1_wait(int kind) {   // kind: 0 -> 0, 1 -> P1, 2 -> P2
    typedef X1Cfg<NU> C;
    if (kind == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (kind == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::P1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::P2) : "memory");
    }
}

// max over the two lanes (l, l + 32) that share a sample, as a vector-ALU operation (v_permlane32_swap, gfx950):
// __shfl_xor(v, 32) is an LDS round trip in the middle of the K loop's dependency chain
__device__ __forceinline__ float x1_pair_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ---- holder / epilogue waves ------------------------------------------------------------------------------------------
// One epilogue step: NP pieces of four raw columns (two variables each) of one row, 16 columns apart: the four lanes of
// a row own interleaved pieces, so one store instruction writes 64 contiguous bytes per row (a lane owning 16 contiguous
// columns wrote four 16-byte fragments 64 bytes apart per row and instruction).  xk: the kept pieces (input affine
// applied), zb: the lane's z values (t; s one slab further), col0: the first raw column.  Pairs of variables go through
// the packed fp32 instructions; PM (the parity of the masked columns) is a template parameter: as a run-time value every
// element select became a chain of v_cndmask (310 instructions per step).
template <bool AFFINE, bool BASE, int PM, bool INV, int NP>
__device__ __forceinline__ void x1_epilogue(const gf32x4 *xk, const lchar *zb, float act, const lfloat *base_l, int D, int col0,
                                            bool row_ok, float *orow, float &ssum, float &bacc) {
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) const gf32x2 lf2;
    const float c2 = (INV ? act : -act) * 1.4426950408889634f;
    gf32x2 ssum2 = {0.f, 0.f};
    // (every z value requested before the first is used: one LDS round trip per step, not one per piece)
    gf32x2 tz[NP], sz[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        tz[k] = *(lf2 *)(zb + k * 32);
        sz[k] = tz[k];
        if (AFFINE) sz[k] = *(lf2 *)(zb + kX1ZSlab + k * 32);
    }
    gf32x4 yo[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        gf32x4 y = xk[k];
        // elements 2 i + (1 - PM) are transformed
        const gf32x2 xv = {y[1 - PM], y[3 - PM]};
        const gf32x2 tv = tz[k];
        gf32x2 ov;
        if (AFFINE) {
            const gf32x2 zz = sz[k];
            // tanh(v) = (e^{2v} - 1) / (e^{2v} + 1) on the hardware exp2 / rcp (absolute error ~1e-7, as x3_tanh)
            gf32x2 e;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float c = __builtin_amdgcn_fmed3f(zz[i], -15.f, 15.f);   // (one instruction; NaN -> -15 like fmaxf first)
                e[i] = __builtin_amdgcn_exp2f(c * 2.8853900817779268f);
            }
            const gf32x2 ep = e + 1.f;
            gf32x2 r;
#pragma unroll
            for (int i = 0; i < 2; ++i) r[i] = __builtin_amdgcn_rcpf(ep[i]);
            const gf32x2 th = (e - 1.f) * r;
            ssum2 += th;                              // (x act at the end; D % 16 == 0: the variables exist)
            const gf32x2 ea = th * c2;
            gf32x2 es;
#pragma unroll
            for (int i = 0; i < 2; ++i) es[i] = __builtin_amdgcn_exp2f(ea[i]);
            ov = INV ? xv * es + tv : (xv - tv) * es;
        } else {
            ov = INV ? xv + tv : xv - tv;
        }
        y[1 - PM] = ov[0];
        y[3 - PM] = ov[1];
        if (BASE) {
            const gf32x4 ba = *(lf4 *)(base_l + col0 + 16 * k), bc = *(lf4 *)(base_l + D + col0 + 16 * k);
            const gf32x4 t = y * ba + bc;
#pragma unroll
            for (int i = 0; i < 4; ++i) bacc = fmaf(-t[i], t[i], bacc);
        }
        yo[k] = y;
        if (BASE && NP > 1 && (k & 1)) __builtin_amdgcn_sched_barrier(0);   // (register pressure: eight columns at a time)
    }
    // (the stores behind ONE branch after the arithmetic of all pieces: a guarded store per piece split the step into
    // basic blocks and every piece's exp -> rcp -> exp chain ran alone, 1850 cycles for ~100 instructions)
    if (!BASE && row_ok) {
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<gf32x4 *>(orow + col0 + 16 * k) = yo[k];
    }
    if (AFFINE) ssum += act * (ssum2[0] + ssum2[1]);
}

// The register file of the four holder waves is the tile's home: kX1FullCh full chunks (16 columns per lane) and, when D
// ends with a 16-column chunk (784 = 12 x 64 + 16), that chunk spread over the four lanes of a row (one piece each).
constexpr int kX1FullCh = 12;

template <bool AFFINE, int NU, bool BASE, int PM>
__device__ __forceinline__ void x1_holder(const X3Args &a, lchar *smem, lchar *zbuf, const lfloat *aff_l, const lfloat *base_l,
                                          int wave, int lane_in, int wave8) {
    typedef X1Cfg<NU> C;
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    const int D = a.D, NCH1 = a.g.NCH1, NPT = a.g.NPT;
    const int NFULL = D >> 6;                       // full chunks; NCH1 == NFULL + 1: a 16-column tail chunk follows
    const int grid = (int)gridDim.x, ntiles = a.ntiles;
    const float base_cst = BASE ? a.base_ac[2 * D] : 0.f;
    const float act = AFFINE ? a.act_weight[0] : 0.f;
    const bool inv = a.inverse != 0;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    gf32x4 xr[kX1FullCh][4];
    gf32x4 xtail = {0.f, 0.f, 0.f, 0.f};
    int hstage = 0;
    [[maybe_unused]] int hrow = 0;
    [[maybe_unused]] const int lane = lane_in;     // (X3_STAMP)
    __syncthreads();
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
        float ssum = 0.f, bacc = 0.f;
        // ---- phase-1 steps: issue chunk c + 2 (chunks 2 .. NCH1 + 1 are the holders'), keep chunk c ----
        // (unrolled: the register file is indexed statically; the lane number made opaque per step so that the offsets
        // derived from it are recomputed instead of living in registers -- every register here is wanted for x)
#pragma unroll
        for (int c = 0; c <= kX1FullCh; ++c) {
            if (c < NCH1) {
                if (c > 0) {   // (the barrier of chunk 0 is the last step of the previous tile / the opening one below)
                    X3_STAMP(hrow, 0);
                    // chunks 2.. were issued by this wave; chunk c + 1 (issued one step ago) may stay in flight
                    if (c >= 2) x1_wait<NU>(c + 1 < NCH1 ? 1 : 2);
                    gemm_lds_barrier();
                    X3_STAMP(hrow, 1);
                } else if (tile == (int)blockIdx.x) {
                    gemm_lds_barrier();
                }
                int lo = lane_in;
                asm volatile("" : "+v"(lo));
                {
                    int is = hstage + 2;
                    is = is >= kGemmStages ? is - kGemmStages : is;
                    // (the chunk number made opaque: thirteen unrolled copies of the issue code with constant chunk numbers
                    // had their ragged-edge offsets hoisted out of the tile loop -- 120 spilled registers)
                    int cn = c + 2;
                    asm volatile("" : "+s"(cn));
                    unsigned voff[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = wave * 16 + j * 4 + (lo >> 4);
                        voff[j] = (unsigned)(r * D + ((lo & 15) ^ (r & 15)) * 4) * 4u;
                    }
                    x1_issue<NU>(a, tile, cn, is, wave, lo, smem_base, voff);
                }
                const lchar *st = smem + hstage * C::STAGE;
                hstage = (hstage + 1 == kGemmStages) ? 0 : hstage + 1;
                const int rl = wave * 16 + (lo >> 2), q = lo & 3;
                if (c < kX1FullCh && c < NFULL) {
                    const int col0 = 64 * c + 4 * q;   // piece i of the lane: raw columns col0 + 16 i .. + 3
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const gf32x4 v = *(lf4 *)(st + (unsigned)(rl * 256 + (((4 * i + q) ^ (rl & 15)) << 4)));
                        const gf32x4 sc = *(lf4 *)(aff_l + col0 + 16 * i), sh = *(lf4 *)(aff_l + D + col0 + 16 * i);
                        xr[c < kX1FullCh ? c : 0][i] = v * sc + sh;
                        // (register pressure: one 16-byte piece at a time once most of the tile is held; the early chunks'
                        // twelve LDS reads go out together)
                        if (c >= 7) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {   // the 16-column tail chunk: piece q of row rl
                    const int col0 = 64 * c + 4 * q;
                    const gf32x4 v = *(lf4 *)(st + (unsigned)(rl * 256 + ((q ^ (rl & 15)) << 4)));
                    const gf32x4 sc = *(lf4 *)(aff_l + col0), sh = *(lf4 *)(aff_l + D + col0);
                    xtail = v * sc + sh;
                }
                X3_STAMP(hrow, 2);
                ++hrow;
            }
        }
        // ---- phase-2 steps ----
#pragma unroll
        for (int pt = 0; pt <= kX1FullCh + 1; ++pt) {
            if (pt <= NPT) {
                // pt < NPT: this tile's chunk NCH1 + pt; pt == NPT: chunk 0 of the next tile, or the closing barrier
                X3_STAMP(hrow, 0);
                if (pt == 0) x1_wait<NU>(2);        // chunk NCH1 (ours); chunk NCH1 + 1 may stay in flight
                else if (pt == 1) x1_wait<NU>(0);   // chunk NCH1 + 1, the last one the holders issue (no store is in flight yet)
                gemm_lds_barrier();
                X3_STAMP(hrow, 1);
                if (pt == 0) gemm_lds_barrier();    // behind the MFMA waves' fragment swap
                if (pt < NPT) hstage = (hstage + 1 == kGemmStages) ? 0 : hstage + 1;
                if (pt > 0) {
                    // ---- epilogue of variable tile p = pt - 1 ----
                    const int pc_ = pt > 0 ? pt - 1 : 0;   // (a constant after unrolling: xr stays in registers)
                    int lo = lane_in;
                    asm volatile("" : "+v"(lo));
                    const int rl = wave * 16 + (lo >> 2), q = lo & 3;
                    const int64_t b = (int64_t)tile * kX1Tile + rl;
#ifdef DPK_X3_TIMELINE
                    // measurement: DPK_X3_TIMELINE=2 drops the stores of the stamped work-group (results are wrong)
                    const bool row_ok = b < a.B && !(a.dbg && blockIdx.x == 0 && a.accumulate == 7);
#else
                    const bool row_ok = b < a.B;
#endif
                    float *orow = a.out + (row_ok ? b : 0) * D;
                    const lchar *zb = zbuf + (pc_ & 1) * 2 * kX1ZSlab + rl * kX1ZRow;
                    if (pc_ < kX1FullCh && pc_ < NFULL) {   // raw columns 64 p + 16 k + 4 q .. + 3 (k = 0..3) of row rl
                        const gf32x4 *xk = xr[pc_ < kX1FullCh ? pc_ : 0];
                        if (inv) x1_epilogue<AFFINE, BASE, PM, true, 4>(xk, zb + q * 8, act, base_l, D, 64 * pc_ + 4 * q, row_ok, orow, ssum, bacc);
                        else x1_epilogue<AFFINE, BASE, PM, false, 4>(xk, zb + q * 8, act, base_l, D, 64 * pc_ + 4 * q, row_ok, orow, ssum, bacc);
                    } else {                                // the tail: raw columns 64 p + 4 q .. + 3
                        if (inv) x1_epilogue<AFFINE, BASE, PM, true, 1>(&xtail, zb + q * 8, act, base_l, D, 64 * pc_ + 4 * q, row_ok, orow, ssum, bacc);
                        else x1_epilogue<AFFINE, BASE, PM, false, 1>(&xtail, zb + q * 8, act, base_l, D, 64 * pc_ + 4 * q, row_ok, orow, ssum, bacc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                X3_STAMP(hrow, 2);
                ++hrow;
            }
        }
        // ---- the four lanes of a sample hold disjoint quarters of its variables ----
        const int rl = wave * 16 + (lane_in >> 2), q = lane_in & 3;
        const int64_t b = (int64_t)tile * kX1Tile + rl;
        const bool row_ok = b < a.B;
        float tot = ssum + __shfl_xor(ssum, 1, 64);
        tot += __shfl_xor(tot, 2, 64);
        if (BASE) {
            float btot = bacc + __shfl_xor(bacc, 1, 64);
            btot += __shfl_xor(btot, 2, 64);
            if (q == 0 && row_ok)
                a.ll_out[b] = btot + base_cst + (a.ildj_in ? a.ildj_in[b] : 0.f) + (AFFINE ? -tot : 0.f);
        } else if (q == 0 && row_ok) {
            const float v = AFFINE ? (inv ? tot : -tot) : 0.f;
            if (a.accumulate) a.ldj[b] += v; else a.ldj[b] = v;
        }
    }
}

// ---- MFMA waves ---------------------------------------------------------------------------------------------------------
template <bool AFFINE, int NU, int ROLE, int PM>
__device__ __forceinline__ void x1_mfma(const X3Args &a, lchar *smem, lchar *zbuf, const lfloat *b1_l, int wave, int lane,
                                        int wave8) {
    typedef X1Cfg<NU> C;
    constexpr int KK = C::KK;
    constexpr int NT = (NU + 1 - ROLE) / 2;          // hidden blocks of this role: T = ROLE, ROLE + 2, ...
    constexpr int NTP = (NU + ROLE) / 2;             // ... and of the partner: T = 1 - ROLE, 3 - ROLE, ...
    constexpr int NTA = NT > 0 ? NT : 1;
    typedef __attribute__((address_space(3))) const gf32x4 lf4;
    typedef __attribute__((address_space(3))) gf32x4 lf4w;
    typedef __attribute__((address_space(3))) const half8 lh8;
    typedef __attribute__((address_space(3))) half8 lh8w;
    const int D = a.D, NCH1 = a.g.NCH1, NPT = a.g.NPT;
    const int grid = (int)gridDim.x, ntiles = a.ntiles;
    const int nchunks = NCH1 + NPT;
    const int s = lane & 31, h = lane >> 5;
    const int sb = wave >> 1;
    const float w1sc = a.scales[0], w2sc = a.scales[1];
    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 16 + j * 4 + (lane >> 4);
        voff[j] = (unsigned)(r * D + ((lane & 15) ^ (r & 15)) * 4) * 4u;
    }
    // (the prologue's table loads are compiler-counted and complete at this barrier; the DMAs start after it)
    __syncthreads();
    x1_issue<NU>(a, (int)blockIdx.x, 0, 0, wave, lane, smem_base, voff);
    x1_issue<NU>(a, (int)blockIdx.x, 1, 1, wave, lane, smem_base, voff);

    const int rl_own = sb * 32 + s;
    const int sw = rl_own & 15;
    unsigned xoff[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) xoff[j][qq] = (unsigned)(rl_own * 256 + (((8 * j + 4 * h + qq) ^ sw) << 4));
    const unsigned foff = (unsigned)(lane * 16);
    constexpr bool z_active = AFFINE || ROLE == 0;
    lchar *zrow = zbuf + ROLE * kX1ZSlab + rl_own * kX1ZRow + h * 16;
    lchar *swap_own = zbuf + wave * kX1SwapWave + foff;           // [2 NT K-steps][hi, lo][64 lanes x 16 B]
    const lchar *swap_par = zbuf + (wave ^ 1) * kX1SwapWave + foff;
    lfloat *hsc_own = (lfloat *)(zbuf + 4 * kX1SwapWave) + wave * 64 + lane;
    const lfloat *hsc_par = (lfloat *)(zbuf + 4 * kX1SwapWave) + (wave ^ 1) * 64 + lane;

    int cstage = 0;
    [[maybe_unused]] int crow = 0;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += grid) {
        const bool more = tile + grid < ntiles;
        gf32x16 acc[NTA];
#pragma unroll
        for (int t = 0; t < NTA; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        constexpr float kBig = 16384.f;
        float xsc = 1.f;
        // ---- phase 1: this role's blocks of H^T = W1m X^T ------------------------------------------------------
        for (int c = 0; c < NCH1; ++c) {
            X3_STAMP(crow, 0);
            // chunks 0 and 1 of a tile are the MFMA waves' (issued two steps back); chunk 1 may stay in flight behind 0
            if (c == 0) x1_wait<NU>(1);
            else if (c == 1) x1_wait<NU>(0);
            X3_STAMP(crow, 1);
            gemm_lds_barrier();
            X3_STAMP(crow, 2);
            const lchar *st = smem + cstage * C::STAGE;
            cstage = (cstage + 1 == kGemmStages) ? 0 : cstage + 1;
            const lchar *tb = st + kX1XB + foff;
            if (NT > 0) {
                // both K-steps of the chunk at once: one wait for the x pieces, one range check (max over the 16 values,
                // the partner lane's through v_permlane32_swap), then conversions and MFMAs -- as two K-steps in series the
                // chain ds_read -> max -> LDS shuffle -> branch -> split -> MFMA ran twice per chunk
                // (the chunk's W1 fragments are requested with the x pieces: behind the range-check branch they were a second
                // LDS round trip in the step's dependency chain)
                half8 wh[2][NTA], wl[2][NTA];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int T = ROLE + 2 * t;
                        wh[j][t] = *(lh8 *)(tb + (j * NU + T) * 2048);
                        wl[j][t] = *(lh8 *)(tb + (j * NU + T) * 2048 + 1024);
                    }
                float v[2][8];
                float m8 = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float raw[16];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const gf32x4 p4 = *(lf4 *)(st + xoff[j][qq]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) raw[4 * qq + i] = p4[i];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m = PM ? raw[2 * i + 1] : raw[2 * i];
                        // (columns beyond D hold clamped copies; their W1 entries are zero, the values must be finite)
                        v[j][i] = (c * 64 + j * 32 + 16 * h + 2 * i < D) ? m : 0.f;
                        m8 = fmaxf(m8, fabsf(v[j][i]));
                    }
                }
                m8 = x1_pair_max(m8);
                if (__builtin_expect(m8 * xsc > kBig && m8 < 3.0e38f, 0)) {
                    const float f = exp2f(-ceilf(log2f(m8 * xsc * (1.f / kBig))));
#pragma unroll
                    for (int t = 0; t < NTA; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[t][i] *= f;
                    xsc *= f;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (c * 64 + j * 32 < D) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[j][i] *= xsc;
                        half8 xh, xl;
                        split8(v[j], xh, xl);
                        // independent accumulators alternate (a dependent 32x32x16 MFMA waits out its predecessor's
                        // 64-cycle latency; back-to-back independent ones issue every 32)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][t], xh, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][t], xl, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j][t], xh, acc[t], 0, 0, 0);
                    }
                }
            }
            X3_STAMP(crow, 4);
            ++crow;
        }
        // ---- bias + ReLU + split of this role's blocks; swap with the partner role ----------------------------
        half8 hh[KK], hl[KK];
        const float xinv = 1.f / (xsc * w1sc);
        float hmax = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int unit = 32 * (ROLE + 2 * t) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                acc[t][reg] = fmaxf(fmaf(acc[t][reg], xinv, b1_l[unit]), 0.f);
                hmax = fmaxf(hmax, acc[t][reg]);
            }
        hmax = x1_pair_max(hmax);
        float hsc = 1.f;
        if (__builtin_expect(hmax > kBig && hmax < 3.0e38f, 0)) hsc = exp2f(-ceilf(log2f(hmax * (1.f / kBig))));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[t][8 * j + i] * hsc;
                const int kk = 2 * (ROLE + 2 * t) + j;
                split8(v, hh[kk], hl[kk]);
                *(lh8w *)(swap_own + (2 * t + j) * 2048) = hh[kk];
                *(lh8w *)(swap_own + (2 * t + j) * 2048 + 1024) = hl[kk];
            }
        *hsc_own = hsc;
        float hinv = 1.f;
        // ---- phase 2: this role's rows of Z^T = W2 H^T per tile of 32 transformed variables -> z slab ----------
        for (int pt = 0; pt < NPT; ++pt) {
            X3_STAMP(crow, 0);
            // chunks NCH1 + 2 .. are the MFMA waves'; the chunk issued one step later (the next tile's first one after
            // this tile's last) may stay in flight
            if (pt >= 2) {
                if (pt + 1 < NPT) x1_wait<NU>(2);
                else x1_wait<NU>(more ? 1 : 0);
            }
            X3_STAMP(crow, 1);
            gemm_lds_barrier();
            X3_STAMP(crow, 2);
            const lchar *st = smem + cstage * C::STAGE;
            // this step's W2 fragments are requested BEFORE the DMA issue (650 cycles: their LDS latency hides under it)
            half8 th[KK], tl[KK];
            if (z_active) {
                const lchar *tb = st + foff + ROLE * 2048;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    th[kk] = *(lh8 *)(tb + kk * 4096);
                    tl[kk] = *(lh8 *)(tb + kk * 4096 + 1024);
                }
            }
            {   // issue the chunk two steps ahead: this tile's, then chunks 0 and 1 of the next tile
                int is = cstage + 2;
                is = is >= kGemmStages ? is - kGemmStages : is;
                const int cn = NCH1 + pt + 2;
                if (cn < nchunks) x1_issue<NU>(a, tile, cn, is, wave, lane, smem_base, voff);
                else if (more) x1_issue<NU>(a, tile + grid, cn - nchunks, is, wave, lane, smem_base, voff);
            }
            X3_STAMP(crow, 3);
            cstage = (cstage + 1 == kGemmStages) ? 0 : cstage + 1;
            if (pt == 0) {
                // the partner's fragments (scaled by ITS power of two) and the common scale of the sample
                const float hp = *hsc_par;
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kk = 2 * (1 - ROLE + 2 * t) + j;
                        hh[kk] = *(lh8 *)(swap_par + (2 * t + j) * 2048);
                        hl[kk] = *(lh8 *)(swap_par + (2 * t + j) * 2048 + 1024);
                    }
                const float hc = fminf(hsc, hp);
                if (__builtin_expect(hsc != hp, 0)) {   // (rare: activations beyond the f16 range in one role only)
                    const _Float16 fo = (_Float16)(hc / hsc), fp = (_Float16)(hc / hp);
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const bool own = ((kk >> 1) & 1) == ROLE;
                        const _Float16 f = own ? fo : fp;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            hh[kk][i] *= f;
                            hl[kk][i] *= f;
                        }
                    }
                }
                hinv = 1.f / (hc * w2sc);
                gemm_lds_barrier();   // every fragment is read: the z slabs (same LDS) may be written
            }
            if (z_active) {
                // three accumulator chains, one per product term: a dependent 32x32x16 MFMA waits out its predecessor's
                // 64-cycle latency, independent ones issue every 32 (two chains left z -> z pairs back to back: 1550
                // cycles for 24 MFMAs)
                gf32x16 z, z1, z2;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    z[i] = 0.f;
                    z1[i] = 0.f;
                    z2[i] = 0.f;
                }
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[kk], hh[kk], z, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[kk], hl[kk], z1, 0, 0, 0);
                    z2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[kk], hh[kk], z2, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] += z1[i] + z2[i];
                const lchar *ex = st + KK * 4096 + ROLE * 128;   // extras: bt (role 0) / bs (role 1), 32 floats each
                lchar *zw = zrow + (pt & 1) * 2 * kX1ZSlab;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const gf32x4 bb = *(lf4 *)(ex + (8 * g4 + 4 * h) * 4);
                    gf32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaf(z[4 * g4 + i], hinv, bb[i]);
                    *(lf4w *)(zw + g4 * 32) = o;
                }
            }
            X3_STAMP(crow, 4);
            ++crow;
        }
    }
    gemm_lds_barrier();   // closing barrier: the holders' epilogue of the last variable tile follows it
}

template <bool AFFINE, int NU, bool BASE, int PM>
__global__ __launch_bounds__(512) void coupling_x1_kernel(const X3Args a) {
    typedef X1Cfg<NU> C;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    lchar *zbuf = smem + kGemmStages * C::STAGE;                 // z slabs [2 buffers][2 roles][64 rows][36 floats] / swap area
    lfloat *b1_l = (lfloat *)(zbuf + 4 * kX1ZSlab);              // [U]
    lfloat *aff_l = b1_l + ((a.U + 3) & ~3);                     // [2][D] input affine (scale, shift) per raw column
    lfloat *base_l = aff_l + 2 * a.D;                            // BASE: [2][D] (a_d, c_d)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const int D = a.D;
    for (int e = tid; e < a.U; e += 512) b1_l[e] = a.b1f[e];
    for (int e = tid; e < D; e += 512) {
        aff_l[e] = a.in_scale ? a.in_scale[e] : 1.f;
        aff_l[D + e] = a.in_shift ? a.in_shift[e] : 0.f;
    }
    if (BASE)
        for (int e = tid; e < 2 * D; e += 512) base_l[e] = a.base_ac[e];
    if (wave8 >= 4) x1_holder<AFFINE, NU, BASE, PM>(a, smem, zbuf, aff_l, base_l, wave, lane, wave8);
    else if (wave & 1) x1_mfma<AFFINE, NU, 1, PM>(a, smem, zbuf, b1_l, wave, lane, wave8);
    else x1_mfma<AFFINE, NU, 0, PM>(a, smem, zbuf, b1_l, wave, lane, wave8);
}

// (a_d, c_d) of the fused Normal base: t_d = a_d u_d + c_d with -t_d^2 = -(sc_d u_d + sh_d - loc_d)^2 / (2 sigma_d^2), and the
// constant sum_d (-log sigma_d - log sqrt(2 pi)) + ildj_const (reference: flows/models/base.py:139-143 behind the affine
// of an eval-mode BatchNormLayer1d, flows/utils.py:118-139).  One block, every call (D-sized, live parameters).
__global__ __launch_bounds__(256) void x3_base_prep_kernel(const float *__restrict__ sc, const float *__restrict__ sh,
                                                           const float *__restrict__ loc, const float *__restrict__ scale,
                                                           const float *__restrict__ ildj_const, int D, float *__restrict__ ac) {
    __shared__ float part[4];
    float csum = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float sg = scale[d], r = 0.70710678118654752440f / sg;
        ac[d] = (sc ? sc[d] : 1.f) * r;
        ac[D + d] = ((sc ? sh[d] : 0.f) - loc[d]) * r;
        csum += -logf(sg) - kLogSqrt2Pi;
    }
    csum = wave_reduce_sum(csum);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = csum;
    __syncthreads();
    if (threadIdx.x == 0) ac[2 * D] = (part[0] + part[1]) + (part[2] + part[3]) + (ildj_const ? *ildj_const : 0.f);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct X3Ws {
    uint16_t *w1t;
    char *w2t;
    float *b1f, *scales;
    float *base_ac;   // [2][D] + constant: the fused Normal base of dpk_coupling1d_pairs_logprob (rebuilt per call)
    X3Check *check;   // fingerprint / gate state of the tables
    int64_t bytes;
};
static X3Ws x3_carve(void *base, const X3Geom &g, int U) {
    X3Ws w{};
    char *p = (char *)base;
    int64_t o = 0;
    auto take = [&](int64_t n) {
        char *q = p ? p + o : nullptr;
        o = align_up(o + n, 256);
        return q;
    };
    w.w1t = (uint16_t *)take((int64_t)g.NCH1 * g.W1CH);
    w.w2t = take((int64_t)g.NPT * g.W2CH);
    w.b1f = (float *)take((int64_t)U * 4);
    w.scales = (float *)take(8);
    w.base_ac = (float *)take((int64_t)(4 * g.K1 + 4) * 4);
    w.check = (X3Check *)take(sizeof(X3Check));
    w.bytes = o;
    return w;
}

// (D % 8: a lane's four consecutive transformed variables are one aligned run of 8 raw columns)
static bool x3_shape_ok(int D, int U) { return D >= 8 && (D % 8) == 0 && (U == 32 || U == 64 || U == 96 || U == 128); }

template <bool AFFINE, int NU, bool BASE = false>
static int x3_launch(const X3Args &a, hipStream_t st) {
    constexpr int W1CH = 2 * NU * 2 * 1024, KK = NU * 2;
    constexpr int W2CH = ((KK * 4 * 1024 + 1024 + 4095) / 4096) * 4096;
    constexpr int STAGE = (kX3XB + W1CH) > W2CH ? (kX3XB + W1CH) : W2CH;
    const size_t lds = (size_t)kGemmStages * STAGE + (size_t)((a.U + 3) & ~3) * 4 + (BASE ? (size_t)2 * a.D * 4 : 0);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "coupling1d_pairs: %zu bytes of LDS", lds);
    auto kern = coupling_x3_kernel<AFFINE, NU, BASE>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_COUPLING1D);
    if (ev0) (void)hipEventRecord(ev0, st);
#ifdef DPK_X3_TIMELINE
    X3Args at = a;
    at.dbg = nullptr;
    if (getenv("DPK_X3_TIMELINE")) {
        (void)hipMalloc(&at.dbg, 8 * 64 * 8 * 8);
        (void)hipMemset(at.dbg, 0, 8 * 64 * 8 * 8);
    }
    DPK_LAUNCH(kern, dim3(grid), dim3(2 * kGemmWaves * 64), lds, st, at);
    if (at.dbg) {   // synchronous read-back: measurement builds only
        std::vector<long long> hb(8 * 64 * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb.data(), at.dbg, hb.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(at.dbg);
        const long long t0 = hb[0];
        for (int w : {0, 3, 4}) {
            fprintf(stderr, "x3 timeline wave %d:", w);
            for (int r = 0; r < 56; ++r) {
                fprintf(stderr, " [");
                for (int sl = 0; sl < 5; ++sl)
                    fprintf(stderr, " %lld", hb[(w * 64 + r) * 8 + sl] ? hb[(w * 64 + r) * 8 + sl] - t0 : -1);
                fprintf(stderr, " ]");
            }
            fprintf(stderr, "\n");
        }
    }
#else
    DPK_LAUNCH(kern, dim3(grid), dim3(2 * kGemmWaves * 64), lds, st, a);
#endif
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("coupling_x3_kernel");
    return DPK_OK;
}

// x-once kernel (coupling_x1_kernel) where its holders can keep the tile.
// DPK_X3_TWO_PASS=1 keeps the two-pass kernel (A/B measurements).
static bool x1_shape_ok(int D) {   // D = 64 n or 64 n + 16, 2 chunks .. 12 full chunks + tail (784 = 12 x 64 + 16)
    static const bool two_pass = getenv("DPK_X3_TWO_PASS") != nullptr;
    // (two chunks at least: ring protocol; the holders keep twelve full chunks and a 16-column tail)
    return !two_pass && D > 64 && ((D % 64) == 0 || (D % 64) == 16) && (D >> 6) <= kX1FullCh;
}

template <bool AFFINE, int NU, bool BASE = false>
static int x1_launch(const X3Args &a_in, hipStream_t st) {
    constexpr int STAGE = X1Cfg<NU>::STAGE;
    X3Args a = a_in;
    a.ntiles = (int)cdiv(a.B, (int64_t)kX1Tile);
    const size_t lds = (size_t)kGemmStages * STAGE + (size_t)4 * kX1ZSlab + (size_t)((a.U + 3) & ~3) * 4 +
                       (size_t)2 * a.D * 4 + (BASE ? (size_t)2 * a.D * 4 : 0);
    DPK_REQUIRE(lds <= 160 * 1024, DPK_EUNSUPPORTED, "coupling1d_pairs: %zu bytes of LDS", lds);
    auto kern = a.pm ? coupling_x1_kernel<AFFINE, NU, BASE, 1> : coupling_x1_kernel<AFFINE, NU, BASE, 0>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    const int cus = device_cus();
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DPK_KERNEL_COUPLING1D);
    if (ev0) (void)hipEventRecord(ev0, st);
#ifdef DPK_X3_TIMELINE
    if (getenv("DPK_X3_TIMELINE")) {
        (void)hipMalloc(&a.dbg, 8 * 64 * 8 * 8);
        (void)hipMemset(a.dbg, 0, 8 * 64 * 8 * 8);
        if (atoi(getenv("DPK_X3_TIMELINE")) == 2) a.accumulate = 7;
    }
#endif
    DPK_LAUNCH(kern, dim3(grid), dim3(512), lds, st, a);
#ifdef DPK_X3_TIMELINE
    if (a.dbg) {   // synchronous read-back: measurement builds only
        std::vector<long long> hb(8 * 64 * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb.data(), a.dbg, hb.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(a.dbg);
        const long long t0 = hb[0];
        for (int w : {0, 1, 4}) {
            fprintf(stderr, "x1 timeline wave %d:", w);
            for (int r = 0; r < 60; ++r) {
                fprintf(stderr, " [");
                for (int sl = 0; sl < 5; ++sl)
                    fprintf(stderr, " %lld", hb[(w * 64 + r) * 8 + sl] ? hb[(w * 64 + r) * 8 + sl] - t0 : -1);
                fprintf(stderr, " ]");
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("coupling_x1_kernel");
    return DPK_OK;
}

}  // namespace dpk

using namespace dpk;

extern "C" int64_t dpk_coupling1d_pairs_workspace_bytes(int32_t D, int32_t units) {
    if (D <= 0 || units <= 0) return DPK_EINVAL;
    if (!x3_shape_ok(D, units)) return DPK_EUNSUPPORTED;
    return x3_carve(nullptr, x3_geom(D, units), units).bytes;
}

struct X3Base {   // the fused tail of dpk_coupling1d_pairs_logprob (all null: plain coupling)
    const float *out_scale, *out_shift, *loc, *scale, *ildj_in, *ildj_const;
    float *ll;
};
static int x3_forward_common(const float *x, int64_t B, int32_t D, int32_t masked_parity,
                             const float *W1, const float *b1, const float *W2, const float *b2,
                             int32_t units, const float *act_weight, const float *in_scale,
                             const float *in_shift, int32_t affine, int32_t inverse, float *out,
                             float *ldj, int32_t accumulate_ldj, void *ws, int64_t ws_bytes,
                             uint32_t flags, void *stream, const X3Base *base) {
    DPK_REQUIRE(B >= 0 && D > 0 && units > 0 && (masked_parity == 0 || masked_parity == 1), DPK_EINVAL,
                "coupling1d_pairs: bad sizes");
    DPK_REQUIRE(x3_shape_ok(D, units), DPK_EUNSUPPORTED, "coupling1d_pairs: D=%d units=%d not built", D, units);
    DPK_REQUIRE(W1 && b1 && W2 && b2 && ws, DPK_EINVAL, "coupling1d_pairs: null pointer");
    DPK_REQUIRE(!affine || act_weight, DPK_EINVAL, "coupling1d_pairs: affine coupling needs the ScaledTanh weight");
    DPK_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DPK_EINVAL, "coupling1d_pairs: scale/shift mismatch");
    if (B == 0) return DPK_OK;
    if (base) {
        DPK_REQUIRE(x && base->loc && base->scale && base->ll && !inverse, DPK_EINVAL, "coupling1d_pairs_logprob: null pointer");
        DPK_REQUIRE((base->out_scale == nullptr) == (base->out_shift == nullptr), DPK_EINVAL,
                    "coupling1d_pairs_logprob: scale/shift mismatch");
        DPK_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, DPK_EUNSUPPORTED, "coupling1d_pairs: x must be 16-byte aligned");
    } else {
        DPK_REQUIRE(x && out && ldj, DPK_EINVAL, "coupling1d_pairs: null pointer");
        DPK_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                    DPK_EUNSUPPORTED, "coupling1d_pairs: x / out must be 16-byte aligned");
    }
    const X3Geom g = x3_geom(D, units);
    const X3Ws w = x3_carve(ws, g, units);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "coupling1d_pairs: workspace %lld < %lld", (long long)ws_bytes,
                (long long)w.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & DPK_FLAG_PARAMS_CACHED)) {
        const bool verify = (flags & DPK_FLAG_PARAMS_VERIFY) != 0;
        if (!verify)   // (the workspace may be fresh memory: the check state starts from zero)
            DPK_REQUIRE(hipMemsetAsync(w.check, 0, sizeof(X3Check), st) == hipSuccess, DPK_ELAUNCH, "memset");
        X3PackArgs p{};
        p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.in_scale = in_scale; p.in_shift = in_shift;
        p.D = D; p.U = units; p.pm = masked_parity; p.affine = affine; p.g = g;
        p.w1t = w.w1t; p.w2t = w.w2t; p.b1f = w.b1f; p.scales = w.scales;
        p.gate = &w.check->gate;
        DPK_LAUNCH(coupling_x3_check_kernel, dim3(kX3CheckBlocks), dim3(256), 0, st, p, w.check, verify ? 1 : 0);
        const int64_t total = (int64_t)g.NCH1 * 2 * g.NU * 64 + (int64_t)g.NPT * (units / 16) * 2 * 64 + (int64_t)g.NPT * 32 +
                              units;
        DPK_LAUNCH(coupling_x3_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p);
        DPK_CHECK_LAUNCH("coupling_x3_pack_kernel");
    }
    X3Args a{};
    a.x = x; a.out = out; a.ldj = ldj; a.B = B; a.D = D; a.U = units; a.pm = masked_parity; a.inverse = inverse;
    a.accumulate = accumulate_ldj; a.ntiles = cdiv(B, kX3Tile); a.g = g;
    a.w1t = w.w1t; a.w2t = w.w2t; a.b1f = w.b1f; a.act_weight = act_weight; a.scales = w.scales;
    a.in_scale = in_scale; a.in_shift = in_shift;
    const bool once = x1_shape_ok(D);
    if (base) {
        DPK_LAUNCH(x3_base_prep_kernel, dim3(1), dim3(256), 0, st, base->out_scale, base->out_shift, base->loc, base->scale,
                   base->ildj_const, D, w.base_ac);
        DPK_CHECK_LAUNCH("x3_base_prep_kernel");
        a.base_ac = w.base_ac; a.ildj_in = base->ildj_in; a.ll_out = base->ll;
        if (once) switch (units / 32) {
            case 1: return affine ? x1_launch<true, 1, true>(a, st) : x1_launch<false, 1, true>(a, st);
            case 2: return affine ? x1_launch<true, 2, true>(a, st) : x1_launch<false, 2, true>(a, st);
            case 3: return affine ? x1_launch<true, 3, true>(a, st) : x1_launch<false, 3, true>(a, st);
            default: return affine ? x1_launch<true, 4, true>(a, st) : x1_launch<false, 4, true>(a, st);
        }
        switch (units / 32) {
            case 1: return affine ? x3_launch<true, 1, true>(a, st) : x3_launch<false, 1, true>(a, st);
            case 2: return affine ? x3_launch<true, 2, true>(a, st) : x3_launch<false, 2, true>(a, st);
            case 3: return affine ? x3_launch<true, 3, true>(a, st) : x3_launch<false, 3, true>(a, st);
            default: return affine ? x3_launch<true, 4, true>(a, st) : x3_launch<false, 4, true>(a, st);
        }
    }
    if (once) switch (units / 32) {
        case 1: return affine ? x1_launch<true, 1>(a, st) : x1_launch<false, 1>(a, st);
        case 2: return affine ? x1_launch<true, 2>(a, st) : x1_launch<false, 2>(a, st);
        case 3: return affine ? x1_launch<true, 3>(a, st) : x1_launch<false, 3>(a, st);
        default: return affine ? x1_launch<true, 4>(a, st) : x1_launch<false, 4>(a, st);
    }
    switch (units / 32) {
        case 1: return affine ? x3_launch<true, 1>(a, st) : x3_launch<false, 1>(a, st);
        case 2: return affine ? x3_launch<true, 2>(a, st) : x3_launch<false, 2>(a, st);
        case 3: return affine ? x3_launch<true, 3>(a, st) : x3_launch<false, 3>(a, st);
        default: return affine ? x3_launch<true, 4>(a, st) : x3_launch<false, 4>(a, st);
    }
}

extern "C" int dpk_coupling1d_pairs_tables(int32_t n, const dpk_pairs_tables_args *layers, void *stream) {
    DPK_REQUIRE(n >= 0 && n <= kX3ManyMax, DPK_EINVAL, "coupling1d_pairs_tables: n = %d (0..%d)", n, kX3ManyMax);
    if (n == 0) return DPK_OK;
    DPK_REQUIRE(layers, DPK_EINVAL, "coupling1d_pairs_tables: null pointer");
    hipStream_t st = (hipStream_t)stream;
    X3ManyArgs m{};
    for (int l = 0; l < n; ++l) {
        const dpk_pairs_tables_args &q = layers[l];
        DPK_REQUIRE(q.D > 0 && q.units > 0 && (q.masked_parity == 0 || q.masked_parity == 1), DPK_EINVAL,
                    "coupling1d_pairs_tables: bad sizes in layer %d", l);
        DPK_REQUIRE(x3_shape_ok(q.D, q.units), DPK_EUNSUPPORTED, "coupling1d_pairs_tables: D=%d units=%d not built", q.D, q.units);
        DPK_REQUIRE(q.W1 && q.b1 && q.W2 && q.b2 && q.ws, DPK_EINVAL, "coupling1d_pairs_tables: null pointer in layer %d", l);
        DPK_REQUIRE((q.in_scale == nullptr) == (q.in_shift == nullptr), DPK_EINVAL, "coupling1d_pairs_tables: scale/shift mismatch");
        const X3Geom g = x3_geom(q.D, q.units);
        const X3Ws w = x3_carve(q.ws, g, q.units);
        DPK_REQUIRE(q.ws_bytes >= w.bytes, DPK_EWORKSPACE, "coupling1d_pairs_tables: workspace %lld < %lld",
                    (long long)q.ws_bytes, (long long)w.bytes);
        const bool verify = (q.flags & DPK_FLAG_PARAMS_VERIFY) != 0;
        if (!verify)   // (the workspace may be fresh memory: the check state starts from zero)
            DPK_REQUIRE(hipMemsetAsync(w.check, 0, sizeof(X3Check), st) == hipSuccess, DPK_ELAUNCH, "memset");
        X3PackArgs &p = m.L[l];
        p.W1 = q.W1; p.b1 = q.b1; p.W2 = q.W2; p.b2 = q.b2; p.in_scale = q.in_scale; p.in_shift = q.in_shift;
        p.D = q.D; p.U = q.units; p.pm = q.masked_parity; p.affine = q.affine; p.g = g;
        p.w1t = w.w1t; p.w2t = w.w2t; p.b1f = w.b1f; p.scales = w.scales;
        p.gate = &w.check->gate;
        m.st[l] = w.check;
        m.verify[l] = verify ? 1 : 0;
    }
    DPK_LAUNCH(coupling_x3_check_many_kernel, dim3(n * kX3CheckBlocks), dim3(256), 0, st, m);
    DPK_LAUNCH(coupling_x3_pack_many_kernel, dim3(n * kX3PackBlocks), dim3(256), 0, st, m);
    DPK_CHECK_LAUNCH("coupling_x3_pack_many_kernel");
    return DPK_OK;
}

extern "C" int dpk_coupling1d_pairs_forward(const float *x, int64_t B, int32_t D, int32_t masked_parity,
                                            const float *W1, const float *b1, const float *W2, const float *b2,
                                            int32_t units, const float *act_weight, const float *in_scale,
                                            const float *in_shift, int32_t affine, int32_t inverse, float *out,
                                            float *ldj, int32_t accumulate_ldj, void *ws, int64_t ws_bytes,
                                            uint32_t flags, void *stream) {
    return x3_forward_common(x, B, D, masked_parity, W1, b1, W2, b2, units, act_weight, in_scale, in_shift, affine, inverse,
                             out, ldj, accumulate_ldj, ws, ws_bytes, flags, stream, nullptr);
}

extern "C" int dpk_coupling1d_pairs_logprob(const float *x, int64_t B, int32_t D, int32_t masked_parity,
                                            const float *W1, const float *b1, const float *W2, const float *b2,
                                            int32_t units, const float *act_weight, const float *in_scale,
                                            const float *in_shift, int32_t affine, const float *out_scale,
                                            const float *out_shift, const float *base_loc, const float *base_scale,
                                            const float *ildj_in, const float *ildj_const, float *ll, void *ws,
                                            int64_t ws_bytes, uint32_t flags, void *stream) {
    const X3Base base{out_scale, out_shift, base_loc, base_scale, ildj_in, ildj_const, ll};
    return x3_forward_common(x, B, D, masked_parity, W1, b1, W2, b2, units, act_weight, in_scale, in_shift, affine, 0,
                             nullptr, nullptr, 0, ws, ws_bytes, flags, stream, &base);
}
