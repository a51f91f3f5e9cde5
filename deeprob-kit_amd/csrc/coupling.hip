// RealNVP-1D affine / NICE coupling for gfx950: conditioner MLP (Linear -> ReLU -> Linear) on the
// fp32 matrix cores, fused with the tanh / exp / affine / log-det epilogue.
//
// reference: CouplingLayer1d.apply_backward / apply_forward, deeprob/flows/layers/coupling.py:72-104
//            (conditioner :45-56, ScaledTanh deeprob/torch/utils.py:52-70)
//
//   z = W2 relu(W1 (mask*x) + b1) + b2 ;  t, s = chunk(z) ; s = a tanh(s) ; t, s *= inv_mask
//   backward (density) direction:  u = (x - t) exp(-s),  ildj = -sum_d s
//   forward (sampling) direction:  x = u exp(s) + t,     ldj  = +sum_d s
//
// Mapping: a work-group of 4 waves owns 64 samples.  Only the columns where mask != 0 enter the first
// GEMM (K1 = nnz(mask)) and only the rows where inv_mask != 0 leave the second one (N2 = nnz(inv_mask)):
// the weights are re-packed per call into MFMA B-fragment order (coupling_pack_kernel), so a wave
// fetches one float4 per four v_mfma_f32_32x32x2_f32 straight from L2 and no weight ever sits in LDS.
// The x tile (A operand of GEMM 1) is staged through LDS in chunks of 64 masked columns, the hidden
// activations H (A operand of GEMM 2) stay in LDS; t and s tiles of the same 32 variables accumulate
// in the same wave, so the epilogue is register-local.  fp32 MFMA = exact fp32 FMA chains (1e-5 parity
// rules out bf16).  An optional per-variable affine (in_scale, in_shift) is applied to x on load: this
// is how an eval-mode BatchNormLayer1d in front of the coupling is folded in (flows/utils.py:118-139).
#include "common.h"
#include <math.h>

// DPK_CPL_ABLATE (measurement builds only): 1 = no phase-0 copy, 2 = no staging of x (GEMM 1 on stale LDS),
// 3 = no GEMM-1 MFMAs, 4 = no GEMM-2 MFMAs, 5 = no epilogue (math + x gather + stores)
#ifndef DPK_CPL_ABLATE
#define DPK_CPL_ABLATE 0
#endif

namespace dpk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCM = 64;        // samples per work-group
constexpr int kCKC = 64;       // masked columns staged per chunk
constexpr int kCWaves = 4;
#ifndef DPK_CPL_PF
#define DPK_CPL_PF 2
#endif
#ifndef DPK_CPL_XPRE
#define DPK_CPL_XPRE 1
#endif
constexpr int kPF = DPK_CPL_PF;  // B-fragment prefetch distance (float4 per lane)

struct CouplingWs {
    int *kidx;      // [K1p] variable id of the k-th masked column (-1 = padding)
    int *nidx;      // [N2p] variable id of the n-th transformed variable (-1 = padding)
    float *w1p;     // [U/32][K1p/8][64][4]   B fragments of GEMM 1
    float *w2tp;    // [N2p/32][U/8][64][4]   B fragments of GEMM 2, translation rows
    float *w2sp;    // same, log-scale rows (affine only)
    float *b2tp;    // [N2p]
    float *b2sp;    // [N2p]
    int64_t bytes;
    int K1p, N2p;
};

static inline CouplingWs carve_coupling_ws(void *base, int D, int U, int K1, int N2) {
    CouplingWs w{};
    char *p = (char *)base;
    int64_t o = 0;
    auto take = [&](int64_t n) {
        char *q = p ? p + o : nullptr;
        o = align_up(o + n, 256);
        return q;
    };
    w.K1p = (int)align_up(K1 > 0 ? K1 : 1, kCKC);
    w.N2p = (int)align_up(N2 > 0 ? N2 : 1, 32);
    w.kidx = (int *)take((int64_t)w.K1p * 4);
    w.nidx = (int *)take((int64_t)w.N2p * 4);
    w.w1p = (float *)take((int64_t)U * w.K1p * 4);
    w.w2tp = (float *)take((int64_t)w.N2p * U * 4);
    w.w2sp = (float *)take((int64_t)w.N2p * U * 4);
    w.b2tp = (float *)take((int64_t)w.N2p * 4);
    w.b2sp = (float *)take((int64_t)w.N2p * 4);
    w.bytes = o;
    (void)D;
    return w;
}

// index lists of the non-zero entries of mask / inv_mask: one wave, ballot compaction (order preserved)
__global__ __launch_bounds__(64) void coupling_index_kernel(const float *__restrict__ mask,
                                                           const float *__restrict__ inv_mask, int D, int K1p,
                                                           int N2p, int *__restrict__ kidx,
                                                           int *__restrict__ nidx, int *__restrict__ bad) {
    const int lane = threadIdx.x;
    int kbase = 0, nbase = 0, nonbinary = 0;
    int even_ok = 1, odd_ok = 1;   // transformed variables = exactly the even / the odd columns
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        const float m = d < D ? mask[d] : 0.f, im = d < D ? inv_mask[d] : 0.f;
        nonbinary |= !((m == 0.f || m == 1.f) && (im == 0.f || im == 1.f));
        if (d < D) {
            even_ok &= (im != 0.f) == ((d & 1) == 0);
            odd_ok &= (im != 0.f) == ((d & 1) == 1);
        }
        const unsigned long long bm = __ballot(m != 0.f), bi = __ballot(im != 0.f);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (m != 0.f) {
            const int pos = kbase + __popcll(bm & below);
            if (pos < K1p) kidx[pos] = d;
        }
        if (im != 0.f) {
            const int pos = nbase + __popcll(bi & below);
            if (pos < N2p) nidx[pos] = d;
        }
        kbase += __popcll(bm);
        nbase += __popcll(bi);
    }
    for (int k = kbase + lane; k < K1p; k += 64) kidx[k] = -1;
    for (int n = nbase + lane; n < N2p; n += 64) nidx[n] = -1;
    const int any_bad = __any(nonbinary), all_even = __all(even_ok), all_odd = __all(odd_ok);
    if (lane == 0) {
        bad[0] = any_bad ? 1 : 0;
        // bad[1]: 1 / 2 = the transformed variables are the even / odd columns of an even-width input (the
        // reference's CouplingLayer1d masks, coupling.py:58-60): the epilogue then owns (pass-through, transformed)
        // column pairs as 8-byte accesses and the tile is never copied separately; 0 = any other binary mask
        bad[1] = ((D & 1) == 0 && !any_bad) ? (all_even ? 1 : (all_odd ? 2 : 0)) : 0;
    }
}

// B-fragment order of v_mfma_f32_32x32x2_f32: lane l holds B[k = l>>5][j = l&31]; four consecutive
// k-steps are packed into one float4 per lane.
__global__ void coupling_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                     const float *__restrict__ b2, const int *__restrict__ kidx,
                                     const int *__restrict__ nidx, int D, int U, int K1p, int N2p, int affine,
                                     float *__restrict__ w1p, float *__restrict__ w2tp,
                                     float *__restrict__ w2sp, float *__restrict__ b2tp,
                                     float *__restrict__ b2sp) {
    const int64_t n1 = (int64_t)U * K1p, n2 = (int64_t)N2p * U;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n1 + n2 + N2p;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (e < n1) {
            // e = ((nt * (K1p/8) + ks4) * 64 + lane) * 4 + q
            const int q = (int)(e & 3), lane = (int)((e >> 2) & 63);
            const int64_t r = e >> 8;
            const int ks4 = (int)(r % (K1p / 8)), nt = (int)(r / (K1p / 8));
            const int k = 2 * (4 * ks4 + q) + (lane >> 5), j = nt * 32 + (lane & 31);
            const int col = kidx[k];
            w1p[e] = (col >= 0) ? W1[(int64_t)j * D + col] : 0.f;
        } else if (e < n1 + n2) {
            const int64_t f = e - n1;
            const int q = (int)(f & 3), lane = (int)((f >> 2) & 63);
            const int64_t r = f >> 8;
            const int ks4 = (int)(r % (U / 8)), pt = (int)(r / (U / 8));
            const int k = 2 * (4 * ks4 + q) + (lane >> 5);
            const int var = nidx[pt * 32 + (lane & 31)];
            w2tp[f] = (var >= 0) ? W2[(int64_t)var * U + k] : 0.f;
            if (affine) w2sp[f] = (var >= 0) ? W2[(int64_t)(D + var) * U + k] : 0.f;
        } else {
            const int n = (int)(e - n1 - n2);
            const int var = nidx[n];
            b2tp[n] = (var >= 0) ? b2[var] : 0.f;
            if (affine) b2sp[n] = (var >= 0) ? b2[D + var] : 0.f;
        }
    }
}

struct CouplingArgs {
    const float *x;
    float *out;
    float *ldj;          // [B]
    int64_t B;
    int D, U, K1p, N2p;
    const int *kidx, *nidx;
    const float *w1p, *b1, *w2tp, *w2sp, *b2tp, *b2sp;
    const float *in_scale, *in_shift;  // nullable: x <- x*in_scale + in_shift on load
    const float *act_weight;           // ScaledTanh weight (1 float), affine only
    int inverse;                       // 0: density direction (u, ildj); 1: sampling direction (x, ldj)
    int accumulate;                    // ldj[b] += ... instead of =
    const int *flags;                  // [1]: column-pair mode (coupling_index_kernel)
};

// tanh(v) = (e^{2v} - 1) / (e^{2v} + 1) on the hardware exp2 / rcp: absolute error ~1e-7 (the scale s = a tanh(.)
// enters exp(-s) and the log-det sum, where 1e-7 absolute is 1e-7 relative); |v| clamped so e^{2v} stays finite
__device__ __forceinline__ float fast_tanh(float v) {
    const float c = fminf(fmaxf(v, -15.f), 15.f);
    const float t = __expf(2.f * c);
    return __fdividef(t - 1.f, t + 1.f);
}

// row of accumulator register `reg` for this lane (32x32 MFMA C/D layout)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Four work-groups per CU (34 KB of LDS each at U = 128) need <= 128 VGPRs: the register budget is pinned to four
// waves per SIMD (a few cold values spill); measured 452 -> 369 us per layer against the compiler's default of 168
// registers / three waves.
#ifndef DPK_CPL_WAVES
#define DPK_CPL_WAVES 4
#endif
template <bool AFFINE>
__global__ __launch_bounds__(kCWaves * 64) __attribute__((amdgpu_waves_per_eu(DPK_CPL_WAVES, DPK_CPL_WAVES)))
void coupling1d_kernel(const CouplingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int U = a.U, D = a.D;
    const int HS = U + 1;                      // row stride of H
    float *hs = lds;                           // [kCM][U+1]
    // x chunk [kCM][kCKC+1], phase 1 only.  With a single pass over the hidden tiles (64 <= U <= 32 * waves) H is
    // only written after the last chunk has been consumed, so the chunk buffer aliases H (34 KB of LDS per
    // work-group at U = 128: four work-groups per CU); otherwise it lives behind H.
    const bool alias_xs = (U >= 64 && U <= 32 * kCWaves);
    float *ldj_lds = hs + kCM * HS;            // [kCWaves][kCM] per-wave partial sums (fixed order)
    float *xs = alias_xs ? lds : ldj_lds + kCWaves * kCM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * kCM;
    const int rows = (int)min((int64_t)kCM, a.B - b0);
    const bool has_aff = a.in_scale != nullptr;
    // column-pair mode: 0 = generic mask; 1 / 2 = the transformed variables are the even / odd columns.  Then the
    // epilogue reads and writes (pass-through, transformed) pairs and phase 0 is skipped.
#ifdef DPK_CPL_NOPAIR
    const int pair_mode = 0;
#else
    const int pair_mode = __builtin_amdgcn_readfirstlane(a.flags[1]);
#endif
    const bool paired = pair_mode != 0;

    // ---- phase 0: pass-through copy of the tile (every variable; the transformed ones are overwritten by
    // the epilogue), log-det scratch.  Rows over the waves, columns over the lanes: no index division, and
    // 16-byte accesses when the rows are 16-byte aligned.
    if (DPK_CPL_ABLATE != 1 && !paired) {
        const float *xb = a.x + b0 * D;
        float *ob = a.out + b0 * D;
        if ((D & 3) == 0) {
            const int D4 = D >> 2;
            for (int r = wave; r < rows; r += kCWaves) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(xb + (int64_t)r * D);
                f32x4 *dst = reinterpret_cast<f32x4 *>(ob + (int64_t)r * D);
                for (int c = lane; c < D4; c += 64) {
                    f32x4 v = src[c];
                    if (has_aff) {
                        const f32x4 sc = reinterpret_cast<const f32x4 *>(a.in_scale)[c];
                        const f32x4 sh = reinterpret_cast<const f32x4 *>(a.in_shift)[c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q], sc[q], sh[q]);
                    }
                    dst[c] = v;
                }
            }
        } else {
            for (int r = wave; r < rows; r += kCWaves)
                for (int c = lane; c < D; c += 64) {
                    float v = xb[(int64_t)r * D + c];
                    if (has_aff) v = fmaf(v, a.in_scale[c], a.in_shift[c]);
                    ob[(int64_t)r * D + c] = v;
                }
        }
    }
    ldj_lds[tid] = 0.f;  // kCWaves * kCM == blockDim.x

    // ---- phase 1: H = relu(Xm W1^T + b1); wave w owns hidden units [32w', 32w'+32) for w' = w, w+4, ..
    const int n_ntiles = U / 32;
    for (int nt0 = 0; nt0 < n_ntiles; nt0 += kCWaves) {
        const int nt = nt0 + wave;
        const bool nt_ok = nt < n_ntiles;
        f32x16 acc0 = {0}, acc1 = {0};
        for (int kc = 0; kc < a.K1p; kc += kCKC) {
            __syncthreads();
            for (int e = tid; e < (DPK_CPL_ABLATE == 2 ? 0 : kCM * kCKC); e += kCWaves * 64) {
                const int r = e / kCKC, c = e - r * kCKC;
                const int col = a.kidx[kc + c];
                float v = 0.f;
                if (col >= 0 && r < rows) {
                    v = a.x[(b0 + r) * D + col];
                    if (has_aff) v = fmaf(v, a.in_scale[col], a.in_shift[col]);
                }
                xs[r * (kCKC + 1) + c] = v;
            }
            __syncthreads();
            if (nt_ok && DPK_CPL_ABLATE != 3) {
                const f32x4 *bp = reinterpret_cast<const f32x4 *>(a.w1p) +
                                  ((int64_t)nt * (a.K1p / 8) + kc / 8) * 64 + lane;
                const float *ap = xs + (lane & 31) * (kCKC + 1) + (lane >> 5);
                // B fragments run kPF float4 ahead of the MFMAs that consume them (an L2 round trip is ~4x the
                // 8 MFMAs one fragment feeds)
                f32x4 bq[kPF];
#pragma unroll
                for (int j = 0; j < kPF; ++j) bq[j] = bp[j * 64];
#pragma unroll
                for (int ks4 = 0; ks4 < kCKC / 8; ++ks4) {
                    const f32x4 b = bq[ks4 % kPF];
                    if (ks4 + kPF < kCKC / 8) bq[ks4 % kPF] = bp[(ks4 + kPF) * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = 2 * (4 * ks4 + q);
                        const float a0 = ap[k], a1 = ap[32 * (kCKC + 1) + k];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[q], acc1, 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();   // every wave is done reading the last x chunk: H may overwrite it
        if (nt_ok) {
            const int col = nt * 32 + (lane & 31);
            const float bias = a.b1[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, lane);
                hs[row * HS + col] = fmaxf(acc0[r] + bias, 0.f);
                hs[(32 + row) * HS + col] = fmaxf(acc1[r] + bias, 0.f);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: z = H W2^T + b2 on the transformed variables, fused epilogue.  Work item = (32-variable tile,
    // 32-row half): two 32x32 accumulators (t, s) per wave keep the kernel near 100 VGPRs (4 work-groups per CU),
    // and 2 * n_pairs items spread evenly over the 4 waves.
    const float act = AFFINE ? a.act_weight[0] : 0.f;
    // items it = wave, wave + 4, ..: the row half (it & 1) is the same for every item of a wave, so one set of
    // log-det partial sums per wave
    float ssum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[r] = 0.f;
    static_assert((kCWaves & 1) == 0, "a wave's items must share their row half");
    const int n_items = 2 * (a.N2p / 32);
    for (int it = wave; it < n_items; it += kCWaves) {
        const int pt = it >> 1, half = it & 1;   // wave-uniform
        f32x16 t0 = {0}, s0 = {0};
        // the x values this lane transforms are requested BEFORE the MFMA loop: one exposed memory latency per
        // item instead of one per accumulator register in the epilogue
        const int var_p = a.nidx[pt * 32 + (lane & 31)];
        float xpre[16];
        {
            const float *xb = a.x + b0 * D;
            const int lane_off = 4 * (lane >> 5) * D + max(var_p, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = half * 32 + (r & 3) + 8 * (r >> 2);
                const int rowc = min(row0 + 4 * (lane >> 5), rows - 1) - 4 * (lane >> 5);  // clamp ragged tiles
                xpre[r] = (DPK_CPL_ABLATE == 5 || !DPK_CPL_XPRE || paired) ? 0.f : xb[lane_off + rowc * D];
            }
        }
        const f32x4 *btp = reinterpret_cast<const f32x4 *>(a.w2tp) + (int64_t)pt * (U / 8) * 64 + lane;
        const f32x4 *bsp = reinterpret_cast<const f32x4 *>(a.w2sp) + (int64_t)pt * (U / 8) * 64 + lane;
        const float *ap = hs + (half * 32 + (lane & 31)) * HS + (lane >> 5);
        const int nk4 = (DPK_CPL_ABLATE == 4) ? 0 : U / 8;   // a multiple of 4 (U % 32 == 0)
        f32x4 btq[kPF], bsq[kPF];
#pragma unroll
        for (int j = 0; j < kPF; ++j) {
            btq[j] = btp[j * 64];
            bsq[j] = AFFINE ? bsp[j * 64] : f32x4{0, 0, 0, 0};
        }
        for (int ks0 = 0; ks0 < nk4; ks0 += kPF) {
#pragma unroll
            for (int j = 0; j < kPF; ++j) {
                const int ks4 = ks0 + j;
                const f32x4 bt = btq[j], bs = bsq[j];
                if (ks4 + kPF < nk4) {
                    btq[j] = btp[(ks4 + kPF) * 64];
                    if (AFFINE) bsq[j] = bsp[(ks4 + kPF) * 64];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = ap[2 * (4 * ks4 + q)];
                    t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bt[q], t0, 0, 0, 0);
                    if (AFFINE) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bs[q], s0, 0, 0, 0);
                }
            }
        }
        const int n = pt * 32 + (lane & 31);
        const int var = var_p;
        if (var >= 0 && DPK_CPL_ABLATE != 5) {
            const float bt = a.b2tp[n], bs = AFFINE ? a.b2sp[n] : 0.f;
            float sc = 1.f, sh = 0.f, scp = 1.f, shp = 0.f;
            if (has_aff) {
                sc = a.in_scale[var];
                sh = a.in_shift[var];
                if (paired) {
                    scp = a.in_scale[var ^ 1];
                    shp = a.in_shift[var ^ 1];
                }
            }
            // 32-bit offsets inside the tile: lane part (column, +4 rows for the upper half-wave) + a uniform row part
            float *ob = a.out + b0 * D;
            const int lane_off = 4 * (lane >> 5) * D + var;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = half * 32 + (r & 3) + 8 * (r >> 2);   // + 4 * (lane >> 5) = mfma_row
                const int off = lane_off + row0 * D;
                if (row0 + 4 * (lane >> 5) < rows) {
                    const float tv = t0[r] + bt;
                    float xraw, praw = 0.f;
                    if (paired) {   // 8-byte access to the (even, odd) column pair of this lane's variable
                        const float2 v2 = *reinterpret_cast<const float2 *>(a.x + b0 * D + (off & ~1));
                        xraw = (pair_mode == 1) ? v2.x : v2.y;
                        praw = (pair_mode == 1) ? v2.y : v2.x;
                    } else {
                        xraw = DPK_CPL_XPRE ? xpre[r] : a.x[b0 * D + off];
                    }
                    const float xv = fmaf(xraw, sc, sh);
                    float o;
                    if (AFFINE) {
                        const float sv = act * fast_tanh(s0[r] + bs);
                        o = a.inverse ? fmaf(xv, __expf(sv), tv) : (xv - tv) * __expf(-sv);
                        ssum[r] += sv;
                    } else {
                        o = a.inverse ? xv + tv : xv - tv;
                    }
                    if (paired) {
                        const float pv = fmaf(praw, scp, shp);
                        *reinterpret_cast<float2 *>(ob + (off & ~1)) = (pair_mode == 1) ? make_float2(o, pv) : make_float2(pv, o);
                    } else {
                        ob[off] = o;
                    }
                }
            }
        }
    }

    // ---- phase 3: log-det = -/+ sum over the transformed variables of s, per sample
    if (AFFINE) {
        const int half_w = wave & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0 = ssum[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v0 += __shfl_xor(v0, o, 64);  // over the 32 columns of this half-wave
            if ((lane & 31) == 0)  // one writer per (wave, row): no atomics, deterministic
                ldj_lds[wave * kCM + half_w * 32 + mfma_row(r, lane)] = v0;
        }
    }
    __syncthreads();
    if (tid < rows) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kCWaves; ++w) tot += ldj_lds[w * kCM + tid];
        const float v = AFFINE ? (a.inverse ? tot : -tot) : 0.f;
        if (a.accumulate) a.ldj[b0 + tid] += v; else a.ldj[b0 + tid] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Eval-mode BatchNormLayer1d as a per-variable affine (flows/utils.py:118-153):
//   backward: u = (x - mean)/sqrt(var+eps) * exp(weight) + bias,  ildj = sum(weight - log(var+eps)/2)
//   forward : x = (u - bias) exp(-weight) sqrt(var+eps) + mean,   ldj  = -ildj
// The kernel composes it with an incoming affine (sc_in, sh_in):  y = (x*sc_in + sh_in)*g + h.
// ------------------------------------------------------------------------------------------------
__global__ void bn1d_fold_kernel(const float *__restrict__ weight, const float *__restrict__ bias,
                                 const float *__restrict__ var, const float *__restrict__ mean, float eps, int D,
                                 int inverse, const float *__restrict__ sc_in, const float *__restrict__ sh_in,
                                 float *__restrict__ sc_out, float *__restrict__ sh_out,
                                 float *__restrict__ ldj_const, int accumulate) {
    float part = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float v = var[d] + eps;
        float g, h;
        if (!inverse) {
            g = expf(weight[d]) / sqrtf(v);
            h = bias[d] - mean[d] * g;
            part += weight[d] - 0.5f * logf(v);
        } else {
            g = expf(-weight[d]) * sqrtf(v);
            h = mean[d] - bias[d] * g;
            part += -weight[d] + 0.5f * logf(v);
        }
        const float s0 = sc_in ? sc_in[d] : 1.f, h0 = sh_in ? sh_in[d] : 0.f;
        sc_out[d] = s0 * g;
        sh_out[d] = fmaf(h0, g, h);
    }
    part = wave_reduce_sum(part);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        if (accumulate) *ldj_const += tot; else *ldj_const = tot;
    }
}

// y = x*scale + shift (per variable), grid-stride, coalesced
__global__ void affine1d_kernel(const float *__restrict__ x, const float *__restrict__ sc,
                                const float *__restrict__ sh, int64_t total, int D, float *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        out[e] = fmaf(x[e], sc[d], sh[d]);
    }
}

// out[b] = sum_d log N(u[b,d]*sc[d]+sh[d]; loc[d], scale[d]) + ildj[b] + ildj_const
// (NormalizingFlow.forward, flows/models/base.py:139-143 with the default Normal base)
// log N(v sc + sh; loc, s) = -t^2 + c_d with t = v sc' + sh', sc' = sc / (s sqrt 2), sh' = (sh - loc) / (s sqrt 2),
// c_d = -log s - log sqrt(2 pi): a work-group turns the per-variable parameters into (sc', sh') pairs in LDS and
// the constant sum_d c_d once, then its waves stream kBaseRows rows each at two FMAs per element.
constexpr int kBaseRows = 8;
__global__ __launch_bounds__(256) void normal_base_logprob_kernel(
    const float *__restrict__ u, const float *__restrict__ sc, const float *__restrict__ sh,
    const float *__restrict__ loc, const float *__restrict__ scale, const float *__restrict__ ildj,
    const float *__restrict__ ildj_const, int64_t B, int D, float *__restrict__ out, int rows) {
    extern __shared__ __attribute__((aligned(16))) float base_lds[];   // [2][D4*4] sc', sh' (zero padded), [4] sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Dp = (D + 3) & ~3;
    float *scp = base_lds, *shp = base_lds + Dp, *part = base_lds + 2 * Dp;
    float csum = 0.f;
    for (int d = tid; d < Dp; d += 256) {
        // (clamped, unconditional loads + selects: a load behind `if (d < D)` is a dependent round trip per trip of this loop)
        const int dc = min(d, D - 1);
        const float s = scale[dc], r = 0.70710678118654752440f / s;
        const float scv = sc ? sc[dc] : 1.f, shv = sc ? sh[dc] : 0.f, lc = loc[dc];
        const bool in = d < D;
        scp[d] = in ? scv * r : 0.f;
        shp[d] = in ? (shv - lc) * r : 0.f;
        csum += in ? -logf(s) - kLogSqrt2Pi : 0.f;
    }
    csum = wave_reduce_sum(csum);
    if (lane == 0) part[wave] = csum;
    __syncthreads();
    const float cst = (part[0] + part[1]) + (part[2] + part[3]) + (ildj_const ? *ildj_const : 0.f);
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows;
    const bool vec = (D & 3) == 0;
    for (int64_t b = r0; b < min(r0 + rows, B); ++b) {
        const float *row = u + b * D;
        float acc = 0.f;
        if (vec) {
            for (int q = lane; q < (D >> 2); q += 64) {
                const f32x4 v = reinterpret_cast<const f32x4 *>(row)[q];
                const f32x4 a = reinterpret_cast<const f32x4 *>(scp)[q], c = reinterpret_cast<const f32x4 *>(shp)[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = fmaf(v[j], a[j], c[j]);
                    acc = fmaf(-t, t, acc);
                }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                const float t = fmaf(row[d], scp[d], shp[d]);
                acc = fmaf(-t, t, acc);
            }
        }
        acc = wave_reduce_sum(acc);
        if (lane == 0) out[b] = acc + cst + (ildj ? ildj[b] : 0.f);
    }
}

}  // namespace dpk

using namespace dpk;

extern "C" int64_t dpk_coupling1d_workspace_bytes(int32_t D, int32_t units, int32_t n_masked,
                                                  int32_t n_transformed) {
    if (D <= 0 || units <= 0 || n_masked < 0 || n_transformed < 0) return DPK_EINVAL;
    return carve_coupling_ws(nullptr, D, units, n_masked, n_transformed).bytes + 256;
}

extern "C" int dpk_coupling1d_forward(const float *x, int64_t B, int32_t D, const float *mask,
                                      const float *inv_mask, int32_t n_masked, int32_t n_transformed,
                                      const float *W1, const float *b1, const float *W2, const float *b2,
                                      int32_t units, const float *act_weight, const float *in_scale,
                                      const float *in_shift, int32_t affine, int32_t inverse, float *out,
                                      float *ldj, int32_t accumulate_ldj, void *ws, int64_t ws_bytes,
                                      void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && units > 0, DPK_EINVAL, "coupling1d: bad sizes");
    DPK_REQUIRE(mask && inv_mask && W1 && b1 && W2 && b2 && ws, DPK_EINVAL, "coupling1d: null pointer");
    DPK_REQUIRE(!affine || act_weight, DPK_EINVAL, "coupling1d: affine coupling needs the ScaledTanh weight");
    DPK_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DPK_EINVAL, "coupling1d: scale/shift mismatch");
    DPK_REQUIRE(units % 32 == 0 && units <= 512, DPK_EUNSUPPORTED,
                "coupling1d: hidden units=%d not built (multiple of 32, <= 512)", units);
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out && ldj, DPK_EINVAL, "coupling1d: null pointer");
    CouplingWs w = carve_coupling_ws(ws, D, units, n_masked, n_transformed);
    DPK_REQUIRE(ws_bytes >= w.bytes + 256, DPK_EWORKSPACE, "coupling1d: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes + 256);
    int *bad = (int *)((char *)ws + w.bytes);
    hipStream_t st = (hipStream_t)stream;
    DPK_LAUNCH(coupling_index_kernel, dim3(1), dim3(64), 0, st, mask, inv_mask, D, w.K1p, w.N2p, w.kidx,
                       w.nidx, bad);
    const int64_t n_pack = (int64_t)units * w.K1p + (int64_t)w.N2p * units + w.N2p;
    DPK_LAUNCH(coupling_pack_kernel, dim3(cdiv(n_pack, 256) > 2048 ? 2048 : cdiv(n_pack, 256)), dim3(256),
                       0, st, W1, W2, b2, w.kidx, w.nidx, D, units, w.K1p, w.N2p, affine, w.w1p, w.w2tp, w.w2sp,
                       w.b2tp, w.b2sp);
    DPK_CHECK_LAUNCH("coupling_pack_kernel");
    CouplingArgs a{};
    a.x = x; a.out = out; a.ldj = ldj; a.B = B; a.D = D; a.U = units; a.K1p = w.K1p; a.N2p = w.N2p;
    a.kidx = w.kidx; a.nidx = w.nidx; a.w1p = w.w1p; a.b1 = b1; a.w2tp = w.w2tp; a.w2sp = w.w2sp;
    a.b2tp = w.b2tp; a.b2sp = w.b2sp; a.in_scale = in_scale; a.in_shift = in_shift; a.act_weight = act_weight;
    a.inverse = inverse; a.accumulate = accumulate_ldj; a.flags = bad;
    const size_t hs_floats = (size_t)kCM * (units + 1), xs_floats = (size_t)kCM * (kCKC + 1);
    const bool alias_xs = units >= 64 && units <= 32 * kCWaves;   // as in the kernel
    const size_t lds = (hs_floats + kCWaves * kCM + (alias_xs ? 0 : xs_floats)) * sizeof(float);
    const int grid = cdiv(B, kCM);
    if (affine) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(coupling1d_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t ev0, ev1;
        profile_take(&ev0, &ev1, DPK_KERNEL_COUPLING1D);
        if (ev0) (void)hipEventRecord(ev0, st);
        DPK_LAUNCH(coupling1d_kernel<true>, dim3(grid), dim3(kCWaves * 64), lds, st, a);
        if (ev1) (void)hipEventRecord(ev1, st);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(coupling1d_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t ev0, ev1;
        profile_take(&ev0, &ev1, DPK_KERNEL_COUPLING1D);
        if (ev0) (void)hipEventRecord(ev0, st);
        DPK_LAUNCH(coupling1d_kernel<false>, dim3(grid), dim3(kCWaves * 64), lds, st, a);
        if (ev1) (void)hipEventRecord(ev1, st);
    }
    DPK_CHECK_LAUNCH("coupling1d_kernel");
    return DPK_OK;
}

extern "C" int dpk_bn1d_fold(const float *weight, const float *bias, const float *running_var,
                             const float *running_mean, float eps, int32_t D, int32_t inverse,
                             const float *scale_in, const float *shift_in, float *scale_out, float *shift_out,
                             float *ldj_const, int32_t accumulate, void *stream) {
    DPK_REQUIRE(weight && bias && running_var && running_mean && scale_out && shift_out && ldj_const, DPK_EINVAL,
                "bn1d_fold: null pointer");
    DPK_REQUIRE(D > 0 && (scale_in == nullptr) == (shift_in == nullptr), DPK_EINVAL, "bn1d_fold: bad arguments");
    DPK_LAUNCH(bn1d_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weight, bias, running_var,
                       running_mean, eps, D, inverse, scale_in, shift_in, scale_out, shift_out, ldj_const,
                       accumulate);
    DPK_CHECK_LAUNCH("bn1d_fold_kernel");
    return DPK_OK;
}

// every eval-mode BatchNormLayer1d of a flow in one launch (n x D elements: the work is nothing, the launches were the
// cost), plus the sum of their constants
struct Bn1dFoldMany {
    dpk_bn1d_fold_args L[16];
    int n;
    float *total;
};
__global__ __launch_bounds__(1024) void bn1d_fold_many_kernel(const Bn1dFoldMany a) {
    // four groups of 256 threads, a layer per group and round, each with the arithmetic AND the summation order of
    // bn1d_fold_kernel: the constants are bit-identical to the per-layer entry's
    __shared__ float red[4][4];
    __shared__ float cst[16];
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
    for (int l0 = 0; l0 < a.n; l0 += 4) {
        const int l = l0 + grp;
        const bool active = l < a.n;
        float part = 0.f;
        if (active) {
            const dpk_bn1d_fold_args &q = a.L[l];
            for (int d = t; d < q.D; d += 256) {
                const float v = q.running_var[d] + q.eps;
                float g, h;
                if (!q.inverse) {
                    g = expf(q.weight[d]) / sqrtf(v);
                    h = q.bias[d] - q.running_mean[d] * g;
                    part += q.weight[d] - 0.5f * logf(v);
                } else {
                    g = expf(-q.weight[d]) * sqrtf(v);
                    h = q.running_mean[d] - q.bias[d] * g;
                    part += -q.weight[d] + 0.5f * logf(v);
                }
                const float s0 = q.scale_in ? q.scale_in[d] : 1.f, h0 = q.shift_in ? q.shift_in[d] : 0.f;
                q.scale_out[d] = s0 * g;
                q.shift_out[d] = fmaf(h0, g, h);
            }
        }
        part = wave_reduce_sum(part);
        if ((t & 63) == 0) red[grp][t >> 6] = part;
        __syncthreads();
        if (active && t == 0) {
            const float tot = red[grp][0] + red[grp][1] + red[grp][2] + red[grp][3];
            const dpk_bn1d_fold_args &q = a.L[l];
            if (q.accumulate) *q.ldj_const += tot; else *q.ldj_const = tot;
            cst[l] = q.accumulate ? 0.f : tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.total) {
        float total = 0.f;
        for (int l = 0; l < a.n; ++l) total += cst[l];
        *a.total = total;
    }
}

extern "C" int dpk_bn1d_fold_many(int32_t n, const dpk_bn1d_fold_args *layers, float *ldj_total, void *stream) {
    DPK_REQUIRE(n >= 0 && n <= 16, DPK_EINVAL, "bn1d_fold_many: n = %d (0..16)", n);
    if (n == 0) return DPK_OK;
    DPK_REQUIRE(layers, DPK_EINVAL, "bn1d_fold_many: null pointer");
    Bn1dFoldMany a{};
    a.n = n;
    a.total = ldj_total;
    for (int l = 0; l < n; ++l) {
        const dpk_bn1d_fold_args &q = layers[l];
        DPK_REQUIRE(q.weight && q.bias && q.running_var && q.running_mean && q.scale_out && q.shift_out && q.ldj_const,
                    DPK_EINVAL, "bn1d_fold_many: null pointer in layer %d", l);
        DPK_REQUIRE(q.D > 0 && (q.scale_in == nullptr) == (q.shift_in == nullptr), DPK_EINVAL,
                    "bn1d_fold_many: bad arguments in layer %d", l);
        // (the total is the sum of the constants written by this call: accumulating entries would need the old values)
        DPK_REQUIRE(!(ldj_total && q.accumulate), DPK_EINVAL, "bn1d_fold_many: ldj_total with an accumulating layer");
        a.L[l] = q;
    }
    DPK_LAUNCH(bn1d_fold_many_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    DPK_CHECK_LAUNCH("bn1d_fold_many_kernel");
    return DPK_OK;
}

extern "C" int dpk_affine1d_forward(const float *x, const float *scale, const float *shift, int64_t B, int32_t D,
                                    float *out, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0, DPK_EINVAL, "affine1d: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && scale && shift && out, DPK_EINVAL, "affine1d: null pointer");
    const int64_t total = B * D;
    int grid = cdiv(total, 256);
    if (grid > 8192) grid = 8192;
    DPK_LAUNCH(affine1d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, scale, shift, total, D,
                       out);
    DPK_CHECK_LAUNCH("affine1d_kernel");
    return DPK_OK;
}

// LogitLayer (deeprob/flows/utils.py:257-294), one pass: a wave per row.
//   density direction (apply_backward): p = alpha + (1 - 2 alpha) x, u = log p - log(1 - p),
//                                       ildj = -(sum_d (log p + log(1 - p)) + ldj_const)
//   sampling direction (apply_forward): p = sigmoid(u), x = (p - alpha) / (1 - 2 alpha),
//                                       ldj = sum_d (log p + log(1 - p)) + ldj_const
__global__ __launch_bounds__(256) void logit1d_kernel(const float *__restrict__ x, int64_t B, int D, float alpha,
                                                      float ldj_const, int inverse, float *__restrict__ out,
                                                      float *__restrict__ ldj) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *xr = x + b * D;
    float *orow = out + b * D;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = xr[d];
        if (!inverse) {
            const float p = alpha + (1.0f - 2.0f * alpha) * v;
            const float lp = logf(p), lq = logf(1.0f - p);
            orow[d] = lp - lq;
            acc += lp + lq;
        } else {
            const float p = 1.0f / (1.0f + expf(-v));
            orow[d] = (p - alpha) / (1.0f - 2.0f * alpha);
            acc += logf(p) + logf(1.0f - p);
        }
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) ldj[b] = inverse ? (acc + ldj_const) : -(acc + ldj_const);
}

extern "C" int dpk_logit1d_forward(const float *x, int64_t B, int32_t D, float alpha, float ldj_const, int32_t inverse,
                                   float *out, float *ldj, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && alpha > 0.f && alpha < 1.f, DPK_EINVAL, "logit1d: bad arguments");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out && ldj, DPK_EINVAL, "logit1d: null pointer");
    DPK_LAUNCH(logit1d_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, B, D, alpha, ldj_const,
                       inverse, out, ldj);
    DPK_CHECK_LAUNCH("logit1d_kernel");
    return DPK_OK;
}

extern "C" int dpk_normal_base_logprob(const float *u, const float *scale_in, const float *shift_in,
                                       const float *loc, const float *scale, const float *ildj,
                                       const float *ildj_const, int64_t B, int32_t D, float *out, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0, DPK_EINVAL, "normal_base_logprob: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(u && loc && scale && out, DPK_EINVAL, "normal_base_logprob: null pointer");
    DPK_REQUIRE((scale_in == nullptr) == (shift_in == nullptr), DPK_EINVAL, "normal_base_logprob: scale/shift");
    const size_t lds = (size_t)(2 * ((D + 3) & ~3) + 4) * sizeof(float);
    DPK_REQUIRE(lds <= 64 * 1024, DPK_EUNSUPPORTED, "normal_base_logprob: D=%d above the on-chip parameter table", D);
    // rows per wave: 8 once the grid covers the chip a few times over (the per-work-group parameter table amortised), 1 for
    // training batches (B = 512 was 16 work-groups walking 8 rows each: 15 us)
    const int rows = B >= 4 * kBaseRows * 4 * (int64_t)device_cus() ? kBaseRows : 1;
    DPK_LAUNCH(normal_base_logprob_kernel, dim3(cdiv(B, 4 * rows)), dim3(256), lds, (hipStream_t)stream,
                       u, scale_in, shift_in, loc, scale, ildj, ildj_const, B, D, out, rows);
    DPK_CHECK_LAUNCH("normal_base_logprob_kernel");
    return DPK_OK;
}
