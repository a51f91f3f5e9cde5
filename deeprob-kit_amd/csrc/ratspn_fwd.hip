// RAT-SPN forward path for gfx950: table preparation, the leaf (input distribution)
// kernel and the whole-model fused kernel.
//
// Mapping (see DESIGN.md "RAT-SPN forward"):
//   * a work-group of 8 waves owns a tile of T = 64*SPL samples; a LANE owns SPL samples,
//     so everything that depends only on the circuit (variable ids, loc/scale, sum
//     weights) is wave-uniform and travels on the scalar path (s_load), never in VGPRs/LDS;
//   * the x tile is streamed once from HBM in chunks of 128 features, transposed into LDS
//     ([feature][sample], conflict-free rows) so that "x[:, mask]" becomes an LDS read at a
//     scalar offset; the next chunk is already in flight in registers while the current
//     one is consumed (issue-early / write-late);
//   * wave w evaluates repetition w of the region graph: its 2^depth leaf regions are
//     accumulated in registers across the chunks, then the product / sum / root layers of
//     that repetition run in registers in the exp domain; the 8 repetitions meet in one
//     LDS log-sum-exp for the root.
#include "common.h"
#include <math.h>

namespace dpk {

// Tables that only earlier kernels write are read through the constant address space: a
// wave-uniform load from it is always selected as s_load (scalar cache), which is the whole
// point of the lane <-> sample mapping.
#define DPK_CONST __attribute__((address_space(4)))
typedef const DPK_CONST float *cfloat_p;
typedef const DPK_CONST int *cint_p;
template <typename T> __host__ __device__ __forceinline__ const DPK_CONST T *as_const(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const DPK_CONST T *)p;
#pragma clang diagnostic pop
}

// --------------------------------------------------------------------------------------
// structure tables: sort every region's variable ids (the sum over a region is order
// free) so that the entries falling in LDS chunk c are one contiguous range.
// reference: RegionGraphLayer.__init__ mask / pad_mask, deeprob/spn/layers/ratspn.py:42-56
// --------------------------------------------------------------------------------------
__global__ void ratspn_struct_kernel(const int64_t *__restrict__ mask,
                                     const uint8_t *__restrict__ pad, int R, int d, int NC, int dP,
                                     int *__restrict__ fl, int *__restrict__ src,
                                     int *__restrict__ feat, int *__restrict__ cb) {
    extern __shared__ int sm_i[];
    int *ids = sm_i;            // [d]
    int *sid = ids + d;         // [d] sorted ids
    int *ssrc = sid + d;        // [d] original position (or -1) of the sorted entry
    int *lo = ssrc + d;         // [NC+1] first sorted position of every chunk
    int *ps = lo + NC + 1;      // [NC+1] first padded position of every chunk
    const int r = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += blockDim.x) ids[j] = (int)mask[(int64_t)r * d + j];
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        const int key = ids[j];
        int rank = 0;
        for (int jj = 0; jj < d; ++jj) {
            const int kk = ids[jj];
            rank += (kk < key) || (kk == key && jj < j);
        }
        sid[rank] = key;
        ssrc[rank] = (pad != nullptr && pad[(int64_t)r * d + j]) ? -1 : j;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= NC; c += blockDim.x) {
        const int lim = c * kChunk;
        int l = 0, h = d;  // first position with sid[pos] >= lim
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (sid[mid] < lim) l = mid + 1; else h = mid;
        }
        lo[c] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c <= NC; ++c) {
            ps[c] = acc;
            cb[r * (NC + 1) + c] = acc;
            if (c < NC) acc += (lo[c + 1] - lo[c] + kBlock - 1) / kBlock * kBlock;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < dP; q += blockDim.x) {
        int c = 0;
        while (c + 1 <= NC && ps[c + 1] <= q) ++c;  // segment holding padded position q
        int f = -1, sj = -1, l = kChunk;
        if (c < NC) {
            const int p = lo[c] + (q - ps[c]);
            if (p < lo[c + 1]) {
                f = sid[p];
                sj = ssrc[p];
                l = f % kChunk;
            }
        }
        const int64_t o = (int64_t)r * dP + q;
        feat[o] = f;
        src[o] = sj;
        fl[o] = l;
    }
    if (r == 0)  // slack behind the last region: neutral entries
        for (int q = threadIdx.x; q < kTableSlack; q += blockDim.x) {
            const int64_t o = (int64_t)R * dP + q;
            feat[o] = -1;
            src[o] = -1;
            fl[o] = kChunk;
        }
}

// --------------------------------------------------------------------------------------
// parameter tables, rebuilt on every call from the live nn.Parameter storage.
//   Gaussian : term = a*(x-mu)^2 + c,  a = -1/(2 sigma^2), c = -log sigma - log sqrt(2 pi)
//              (torch.distributions.Normal.log_prob as used at ratspn.py:96)
//   Bernoulli: term = x*l - softplus(l) = -BCEWithLogits(l, x)   (ratspn.py:243)
// --------------------------------------------------------------------------------------
template <int DIST>
__global__ void leaf_param_kernel(const float *__restrict__ p0, const float *__restrict__ p1,
                                  const int *__restrict__ src, const int *__restrict__ cb, int R,
                                  int I, int d, int dP, int NC, float *__restrict__ par,
                                  float *__restrict__ cel, float *__restrict__ biasc) {
    const int r = blockIdx.x;
    const int n_ent = dP + (r == R - 1 ? kTableSlack : 0);
    const int n = n_ent * I;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int pidx = e / I, k = e - pidx * I;
        const int j = src[(int64_t)r * dP + pidx];
        float A = 0.f, Bv = 0.f, Cc = 0.f;
        if (j >= 0) {
            const int64_t o = ((int64_t)r * I + k) * d + j;
            if (DIST == 0) {
                const float mu = p0[o], sg = p1[o];
                A = mu;
                Bv = -0.5f / (sg * sg);
                Cc = -logf(sg) - kLogSqrt2Pi;
            } else {
                const float l = p0[o];
                A = l;
                Cc = -(fmaxf(l, 0.f) + log1pf(expf(-fabsf(l))));
            }
        }
        const int64_t po = ((int64_t)r * dP + pidx) * 2 * I;
        par[po + k] = A;
        par[po + I + k] = Bv;
        cel[((int64_t)r * dP + pidx) * I + k] = Cc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NC * I; e += blockDim.x) {
        const int c = e / I, k = e - c * I;
        const int j0 = cb[r * (NC + 1) + c], j1 = cb[r * (NC + 1) + c + 1];
        float s = 0.f;
        for (int p = j0; p < j1; ++p) s += cel[((int64_t)r * dP + p) * I + k];
        biasc[((int64_t)r * NC + c) * I + k] = s;
    }
}

// softmax / log_softmax of every row of a [rows, n] weight matrix (one wave per row).
// reference: torch.log_softmax at ratspn.py:375 and :455
__global__ void softmax_rows_kernel(const float *__restrict__ w, int rows, int n,
                                    float *__restrict__ W, float *__restrict__ LW) {
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

// --------------------------------------------------------------------------------------
// in-register product+sum node:  out[o] = logsumexp_{i,j}(a[i] + c[j] + lw[o,i,j])
// (ProductLayer.forward ratspn.py:280-285 followed by SumLayer.forward :375-377).
// Fast path in the exp domain with linear softmax weights; when the scaled sum falls
// below 1e-30 (dominant pair far from (argmax a, argmax c) AND a vanishing weight) the
// exact two-pass form with the true maximum is used, which is what torch.logsumexp does.
// --------------------------------------------------------------------------------------
struct LseScratch {
    float *slot;  // per-lane LDS slice, 2*NI floats
};

template <int NI>
__device__ __forceinline__ void exact_lse(const float (&a)[NI], const float (&c)[NI],
                                          cfloat_p lw, LseScratch sc, float &m_out, float &s_out) {
    // rare, lane-divergent: keep it small (rolled loops over an LDS copy)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        sc.slot[i] = a[i];
        sc.slot[NI + i] = c[i];
    }
    float m = -INFINITY;
#pragma unroll 1
    for (int i = 0; i < NI; ++i)
#pragma unroll 1
        for (int j = 0; j < NI; ++j) m = fmaxf(m, sc.slot[i] + sc.slot[NI + j] + lw[i * NI + j]);
    float s = 0.f;
    if (m > -INFINITY) {
#pragma unroll 1
        for (int i = 0; i < NI; ++i)
#pragma unroll 1
            for (int j = 0; j < NI; ++j)
                s += expf(sc.slot[i] + sc.slot[NI + j] + lw[i * NI + j] - m);
    }
    m_out = m;
    s_out = s;
}

template <int NI>
__device__ __forceinline__ void exp_children(const float (&a)[NI], float (&ea)[NI], float &ma) {
    float m = a[0];
#pragma unroll
    for (int i = 1; i < NI; ++i) m = fmaxf(m, a[i]);
    const float m0 = (m == -INFINITY) ? 0.f : m;
#pragma unroll
    for (int i = 0; i < NI; ++i) ea[i] = __expf(a[i] - m0);
    ma = m0;
}

template <int NI, int NO>
__device__ __forceinline__ void prodsum_node(const float (&a)[NI], const float (&c)[NI],
                                             cfloat_p W, cfloat_p LW, LseScratch sc,
                                             float (&out)[NO]) {
    float ea[NI], ec[NI], ma, mc;
    exp_children<NI>(a, ea, ma);
    exp_children<NI>(c, ec, mc);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) t = fmaf(W[(o * NI + i) * NI + j], ec[j], t);
            v = fmaf(ea[i], t, v);
        }
        if (v < 1e-30f) {
            float m, s;
            exact_lse<NI>(a, c, LW + o * NI * NI, sc, m, s);
            out[o] = (m > -INFINITY) ? m + logf(s) : -INFINITY;
        } else {
            out[o] = ma + mc + __logf(v);
        }
    }
}

// partial of the root log-sum-exp contributed by one repetition: (m, s) with
// logsumexp = m + log s   (RootLayer.forward ratspn.py:454-457 restricted to one repetition)
template <int NI>
__device__ __forceinline__ void root_partial(const float (&a)[NI], const float (&c)[NI],
                                             const float (&ea)[NI], const float (&ec)[NI], float ma,
                                             float mc, cfloat_p W, cfloat_p LW, LseScratch sc,
                                             float &m_out, float &s_out) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) t = fmaf(W[i * NI + j], ec[j], t);
        v = fmaf(ea[i], t, v);
    }
    if (v < 1e-30f) {
        exact_lse<NI>(a, c, LW, sc, m_out, s_out);
    } else {
        m_out = ma + mc;
        s_out = v;
    }
}

// --------------------------------------------------------------------------------------
// kernel arguments
// --------------------------------------------------------------------------------------
struct LeafArgs {
    const float *x;
    int64_t B;
    int D, R, I, d, NC, dP;
    cint_p fl;
    cint_p cb;
    cfloat_p par;
    cfloat_p cel;
    cfloat_p biasc;
    float *leaf_out;  // [B,R,I] or nullptr
    // fused model
    int reps, C;
    cfloat_p W0, LW0, W1, LW1, Wr, LWr;
    float *out;      // [B,C]
    double *ll_sum;  // [2] or nullptr
};

template <int SPL> struct TileGeom {
    static constexpr int T = 64 * SPL;                 // samples per work-group
    static constexpr int ROW = (SPL == 2) ? 130 : 65;  // LDS dwords per feature row
    static constexpr int ROWB = ROW * 4;
    static constexpr int NLD = T * kChunk / (kLeafWaves * 64);  // staged dwords per thread
    static constexpr int CHUNK_BYTES = (kChunk + 1) * ROW * 4;  // + the all-zero row
    // epilogue reuse of the chunk buffer: 16 floats per thread + root exchange [2][waves][T]
    static constexpr int EPI_BYTES = kLeafWaves * 64 * 16 * 4 + 2 * kLeafWaves * T * 4;
    static constexpr int BUF_BYTES = CHUNK_BYTES > EPI_BYTES ? CHUNK_BYTES : EPI_BYTES;
};

template <int SPL> struct XVec;
template <> struct XVec<1> { using type = float; };
template <> struct XVec<2> { using type = float2; };

// lgkmcnt(0) with vmcnt / expcnt left alone (gfx9 encoding: vmcnt 0x3f, expcnt 7, lgkmcnt 0)
#define DPK_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// One block of kBlock table entries: parameters (SGPRs) and the x values (VGPRs).
template <int CB, int SPL, bool SLOW> struct LeafBlock {
    float p0[kBlock][CB], p1[kBlock][CB], pc[SLOW ? kBlock : 1][CB];
    float x[kBlock][SPL];
};

__device__ __forceinline__ void leaf_load_off(int (&off)[kBlock], cint_p flp, int j, int rowb) {
#pragma unroll
    for (int u = 0; u < kBlock; ++u) off[u] = flp[j + u] * rowb;
}
template <int CB, int SPL, bool SLOW>
__device__ __forceinline__ void leaf_load_par(LeafBlock<CB, SPL, SLOW> &blk, cfloat_p pp, cfloat_p cp,
                                              int I, int j) {
#pragma unroll
    for (int u = 0; u < kBlock; ++u) {
        cfloat_p pr = pp + (int64_t)(j + u) * 2 * I;
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            blk.p0[u][k] = pr[k];
            blk.p1[u][k] = pr[I + k];
            if (SLOW) blk.pc[u][k] = cp[(int64_t)(j + u) * I + k];
        }
    }
}
template <int CB, int SPL, bool SLOW>
__device__ __forceinline__ void leaf_read_x(LeafBlock<CB, SPL, SLOW> &blk, const char *lane_base,
                                            const int (&off)[kBlock]) {
#pragma unroll
    for (int u = 0; u < kBlock; ++u) {
        if (SPL == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(lane_base + off[u]);
            blk.x[u][0] = v.x;
            blk.x[u][SPL - 1] = v.y;
        } else {
            blk.x[u][0] = *reinterpret_cast<const float *>(lane_base + off[u]);
        }
    }
}
template <int DIST, int CB, int SPL, bool SLOW>
__device__ __forceinline__ void leaf_block_compute(float (&acc)[CB][SPL],
                                                   const LeafBlock<CB, SPL, SLOW> &blk) {
#pragma unroll
    for (int u = 0; u < kBlock; ++u)
#pragma unroll
        for (int k = 0; k < CB; ++k)
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                if (DIST == 0) {
                    const float dlt = blk.x[u][s] - blk.p0[u][k];
                    if (!SLOW) acc[k][s] = fmaf(dlt * dlt, blk.p1[u][k], acc[k][s]);
                    else acc[k][s] += nan_to_num_f(fmaf(dlt * dlt, blk.p1[u][k], blk.pc[u][k]));
                } else {
                    if (!SLOW) acc[k][s] = fmaf(blk.x[u][s], blk.p0[u][k], acc[k][s]);
                    else acc[k][s] += nan_to_num_f(fmaf(blk.x[u][s], blk.p0[u][k], blk.pc[u][k]));
                }
            }
}

// Accumulate the table entries [j0, j1) (a multiple of kBlock) of one region over the chunk
// held in LDS.  Software-pipelined by hand over two register sets: while block b is consumed,
// the x reads and parameters of block b+1 and the row offsets of block b+2 are in flight.  The
// single lgkmcnt(0) sits at the END of each step: SMEM returns out of order, so a wait placed
// at first use (what the compiler would do) drains the loads it has just issued.
template <int DIST, int CB, int SPL, bool SLOW>
__device__ __forceinline__ void leaf_accum(float (&acc)[CB][SPL], const char *lane_base, cint_p flp,
                                           cfloat_p pp, cfloat_p cp, int I, int j0, int j1) {
    constexpr int ROWB = TileGeom<SPL>::ROWB;
    if (j0 >= j1) return;
    LeafBlock<CB, SPL, SLOW> A, Bk;
    int off[kBlock];
    leaf_load_off(off, flp, j0, ROWB);
    leaf_load_par<CB, SPL, SLOW>(A, pp, cp, I, j0);
    DPK_WAIT_LGKM0();
    leaf_read_x<CB, SPL, SLOW>(A, lane_base, off);
    __builtin_amdgcn_sched_barrier(0);
    leaf_load_off(off, flp, j0 + kBlock, ROWB);
    DPK_WAIT_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    for (int j = j0;; j += 2 * kBlock) {
        leaf_read_x<CB, SPL, SLOW>(Bk, lane_base, off);
        __builtin_amdgcn_sched_barrier(0);
        leaf_load_off(off, flp, j + 2 * kBlock, ROWB);
        leaf_load_par<CB, SPL, SLOW>(Bk, pp, cp, I, j + kBlock);
        __builtin_amdgcn_sched_barrier(0);
        leaf_block_compute<DIST, CB, SPL, SLOW>(acc, A);
        DPK_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        if (j + kBlock >= j1) break;
        leaf_read_x<CB, SPL, SLOW>(A, lane_base, off);
        __builtin_amdgcn_sched_barrier(0);
        leaf_load_off(off, flp, j + 3 * kBlock, ROWB);
        leaf_load_par<CB, SPL, SLOW>(A, pp, cp, I, j + 2 * kBlock);
        __builtin_amdgcn_sched_barrier(0);
        leaf_block_compute<DIST, CB, SPL, SLOW>(acc, Bk);
        DPK_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        if (j + 2 * kBlock >= j1) break;
    }
}

// DEPTH == 0: leaf only (QB regions x CB channels per wave item, written to leaf_out)
// DEPTH >= 1: fused model, QB == 2^DEPTH, CB == I, S sum nodes
template <int DIST, int QB, int CB, int SPL, int DEPTH, int S>
__global__ __launch_bounds__(kLeafWaves * 64, 4) void ratspn_leaf_kernel(const LeafArgs a) {
    using G = TileGeom<SPL>;
    constexpr int T = G::T, ROW = G::ROW, NLD = G::NLD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs_lds = reinterpret_cast<float *>(smem);
    float *run_m = reinterpret_cast<float *>(smem + G::BUF_BYTES);  // [C][T] fused only
    float *run_s = run_m + (DEPTH > 0 ? a.C * T : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * T;
    const int I = (DEPTH > 0) ? CB : a.I;  // static in the fused kernels
    const int R = a.R, dP = a.dP, NC = a.NC, D = a.D;

    const int n_cblk = (DEPTH > 0) ? 1 : I / CB;
    const int n_items = (DEPTH > 0) ? a.reps : ((R + QB - 1) / QB) * n_cblk;
    const int n_pass = (n_items + kLeafWaves - 1) / kLeafWaves;

    const char *lane_base = smem + lane * (4 * SPL);

    if constexpr (DEPTH > 0) {
        for (int e = tid; e < a.C * T; e += kLeafWaves * 64) {
            run_m[e] = -INFINITY;
            run_s[e] = 0.f;
        }
    }

    for (int pass = 0; pass < n_pass; ++pass) {
        const int item = pass * kLeafWaves + wave;
        const bool active = item < n_items;
        const int g = item / n_cblk;
        const int kb = (item - g * n_cblk) * CB;

        float acc[QB][CB][SPL];
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int k = 0; k < CB; ++k)
#pragma unroll
                for (int s = 0; s < SPL; ++s) acc[q][k][s] = 0.f;

        if (tid < ROW) xs_lds[kChunk * ROW + tid] = 0.f;  // the neutral row (see RatWs)

        // staging map: thread -> feature-in-chunk flc and sample quad sq; load i covers sample
        // s_i = 4*i + sq, so a wave reads 64 consecutive floats of one row (coalesced) and the
        // LDS image [feature][sample] is written at a compile-time stride.
        float pre[NLD];
        const int flc = tid & (kChunk - 1);
        const int sq = tid >> 7;
        const bool full_tile = (b0 + T <= a.B);
        const float *xt = a.x + b0 * D;
        auto load_chunk = [&](int c) {
            // per-thread 32-bit offset + uniform base; the asm keeps the compiler from hoisting
            // NLD loop-invariant 64-bit addresses out of the chunk loop (and spilling them)
            int vo = sq * D + flc;
            asm volatile("" : "+v"(vo));
            if (full_tile && (c + 1) * kChunk <= D) {
                const float *xc = xt + c * kChunk;
#pragma unroll
                for (int i = 0; i < NLD; ++i) pre[i] = (xc + (int64_t)i * 4 * D)[vo];
            } else {  // ragged tile / last chunk: clamp (clamped slots are never consumed)
                const int f = min(c * kChunk + flc, D - 1);
                int sqv = sq;
                asm volatile("" : "+v"(sqv));
                const int nv1 = (int)min((int64_t)T, a.B - b0) - 1;
#pragma unroll
                for (int i = 0; i < NLD; ++i) pre[i] = xt[min(4 * i + sqv, nv1) * D + f];
            }
        };
        load_chunk(0);

        for (int c = 0; c < NC; ++c) {
            __syncthreads();  // every wave is done with the previous chunk
            float chk = 0.f;
            float *wr = xs_lds + flc * ROW + (SPL == 2 ? 2 * sq : sq);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const float v = pre[i];
                chk = fmaf(v, 0.f, chk);  // NaN iff v is NaN or +-inf
                const int pos = (SPL == 2) ? ((i < 16) ? 8 * i : 8 * (i - 16) + 1) : 4 * i;
                wr[pos] = v;
            }
            if (c + 1 < NC) load_chunk(c + 1);
            const int slow = __syncthreads_or(chk != chk);

            if (active) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const int r = g * QB + q;
                    if (r < R) {
                        const int j0 = a.cb[r * (NC + 1) + c], j1 = a.cb[r * (NC + 1) + c + 1];
                        cint_p flp = a.fl + (int64_t)r * dP;
                        cfloat_p pp = a.par + (int64_t)r * dP * 2 * I + kb;
                        cfloat_p cp = a.cel + (int64_t)r * dP * I + kb;
                        if (!slow) {
                            leaf_accum<DIST, CB, SPL, false>(acc[q], lane_base, flp, pp, cp, I, j0, j1);
                            cfloat_p bp = a.biasc + ((int64_t)r * NC + c) * I + kb;
#pragma unroll
                            for (int k = 0; k < CB; ++k) {
                                const float bk = bp[k];
#pragma unroll
                                for (int s = 0; s < SPL; ++s) acc[q][k][s] += bk;
                            }
                        } else {
                            leaf_accum<DIST, CB, SPL, true>(acc[q], lane_base, flp, pp, cp, I, j0, j1);
                        }
                    }
                }
            }
        }

        // ---- leaf outputs ---------------------------------------------------------
        if (a.leaf_out != nullptr && active) {
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int64_t b = b0 + lane + 64 * s;
                if (b < a.B) {
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const int r = g * QB + q;
                        if (r < R) {
                            float *o = a.leaf_out + (b * R + r) * I + kb;
#pragma unroll
                            for (int k = 0; k < CB; ++k) o[k] = acc[q][k][s];
                        }
                    }
                }
            }
        }

        // ---- product / sum / root layers of this repetition, in registers ------------
        if constexpr (DEPTH > 0) {
            constexpr int NI = (DEPTH >= 2) ? S : CB;  // inputs of the last product
            __syncthreads();  // chunk buffer is free: reuse it (exact-path slices, root exchange)
            LseScratch sc{reinterpret_cast<float *>(smem) + tid * (2 * 8)};
            float *cm = reinterpret_cast<float *>(smem) + kLeafWaves * 64 * 16;  // [waves][T]
            float *cs = cm + kLeafWaves * T;
            const int rep = item;
            float ta[SPL][NI], tc[SPL][NI];
            if (active) {
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    if constexpr (DEPTH == 1) {
#pragma unroll
                        for (int k = 0; k < NI; ++k) {
                            ta[s][k] = acc[0][k][s];
                            tc[s][k] = acc[1][k][s];
                        }
                    } else {
                        float v[QB][CB];
#pragma unroll
                        for (int q = 0; q < QB; ++q)
#pragma unroll
                            for (int k = 0; k < CB; ++k) v[q][k] = acc[q][k][s];
                        float n1[QB / 2][S];
#pragma unroll
                        for (int p = 0; p < QB / 2; ++p) {
                            const int64_t wo = ((int64_t)rep * (QB / 2) + p) * S * CB * CB;
                            prodsum_node<CB, S>(v[2 * p], v[2 * p + 1], a.W0 + wo, a.LW0 + wo, sc, n1[p]);
                        }
                        if constexpr (DEPTH == 2) {
#pragma unroll
                            for (int k = 0; k < NI; ++k) {
                                ta[s][k] = n1[0][k];
                                tc[s][k] = n1[1][k];
                            }
                        } else {
                            float n2[2][S];
#pragma unroll
                            for (int p = 0; p < 2; ++p) {
                                const int64_t wo = ((int64_t)rep * 2 + p) * S * S * S;
                                prodsum_node<S, S>(n1[2 * p], n1[2 * p + 1],
                                                   a.W1 + wo, a.LW1 + wo, sc, n2[p]);
                            }
#pragma unroll
                            for (int k = 0; k < NI; ++k) {
                                ta[s][k] = n2[0][k];
                                tc[s][k] = n2[1][k];
                            }
                        }
                    }
                }
            }
            const int M = a.reps * NI * NI;
            for (int cl = 0; cl < a.C; ++cl) {
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    float m = -INFINITY, sv = 0.f;
                    if (active) {
                        float ea[NI], ec[NI], ma, mc;
                        exp_children<NI>(ta[s], ea, ma);
                        exp_children<NI>(tc[s], ec, mc);
                        const int64_t wo = (int64_t)cl * M + (int64_t)rep * NI * NI;
                        root_partial<NI>(ta[s], tc[s], ea, ec, ma, mc, a.Wr + wo, a.LWr + wo, sc, m, sv);
                    }
                    cm[wave * T + lane + 64 * s] = m;
                    cs[wave * T + lane + 64 * s] = sv;
                }
                __syncthreads();
                if (tid < T) {
                    float mm = run_m[cl * T + tid];
#pragma unroll
                    for (int w = 0; w < kLeafWaves; ++w) mm = fmaxf(mm, cm[w * T + tid]);
                    float tot = 0.f;
                    if (mm > -INFINITY) {
                        tot = run_s[cl * T + tid] * __expf(run_m[cl * T + tid] - mm);
#pragma unroll
                        for (int w = 0; w < kLeafWaves; ++w) tot += cs[w * T + tid] * __expf(cm[w * T + tid] - mm);
                    }
                    run_m[cl * T + tid] = mm;
                    run_s[cl * T + tid] = tot;
                }
                __syncthreads();
            }
        }
    }

    if constexpr (DEPTH > 0) {
        double part = 0.0;
        if (tid < T) {
            const int64_t b = b0 + tid;
            if (b < a.B) {
                for (int cl = 0; cl < a.C; ++cl) {
                    const float mm = run_m[cl * T + tid];
                    const float ll = (mm > -INFINITY) ? mm + __logf(run_s[cl * T + tid]) : -INFINITY;
                    a.out[b * a.C + cl] = ll;
                    part += (double)ll;
                }
            }
        }
        if (a.ll_sum != nullptr && tid < T) {
            part = wave_reduce_sum(part);
            if (lane == 0) {
                atomicAdd(a.ll_sum, part);
                int64_t nvalid = a.B - (b0 + (tid & ~63));
                nvalid = nvalid < 0 ? 0 : (nvalid > 64 ? 64 : nvalid);
                atomicAdd(a.ll_sum + 1, (double)(nvalid * a.C));
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
int prepare_leaf_structure(const RatWs &w, const int64_t *mask, const uint8_t *pad, int R, int d,
                           uint32_t flags, hipStream_t st) {
    if (flags & DPK_FLAG_STRUCT_CACHED) return DPK_OK;
    const size_t lds = (size_t)(3 * d + 2 * (w.NC + 1)) * sizeof(int);
    DPK_REQUIRE(lds <= 64 * 1024, DPK_EUNSUPPORTED, "region dimension %d too large for the structure kernel", d);
    hipLaunchKernelGGL(ratspn_struct_kernel, dim3(R), dim3(256), lds, st, mask, pad, R, d, w.NC, w.dP, w.fl,
                       w.src, w.feat, w.cb);
    DPK_CHECK_LAUNCH("ratspn_struct_kernel");
    return DPK_OK;
}

static int prepare_leaf_tables(int dist, const RatWs &w, const int64_t *mask, const uint8_t *pad,
                               const float *p0, const float *p1, int R, int I, int d, uint32_t flags,
                               hipStream_t st) {
    int rc = prepare_leaf_structure(w, mask, pad, R, d, flags, st);
    if (rc) return rc;
    if (dist == 0)
        hipLaunchKernelGGL(leaf_param_kernel<0>, dim3(R), dim3(256), 0, st, p0, p1, w.src, w.cb, R, I, d,
                           w.dP, w.NC, w.par, w.cel, w.biasc);
    else
        hipLaunchKernelGGL(leaf_param_kernel<1>, dim3(R), dim3(256), 0, st, p0, p1, w.src, w.cb, R, I, d,
                           w.dP, w.NC, w.par, w.cel, w.biasc);
    DPK_CHECK_LAUNCH("leaf_param_kernel");
    return DPK_OK;
}

template <int DIST, int QB, int CB, int SPL, int DEPTH, int S>
static int launch_leaf(const LeafArgs &a, hipStream_t st) {
    using G = TileGeom<SPL>;
    const int grid = cdiv(a.B, G::T);
    size_t lds = G::BUF_BYTES;
    if (DEPTH > 0) lds += (size_t)2 * a.C * G::T * sizeof(float);
    auto kern = ratspn_leaf_kernel<DIST, QB, CB, SPL, DEPTH, S>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e));
            return DPK_ELAUNCH;
        }
    }
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1);
    if (ev0) (void)hipEventRecord(ev0, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kLeafWaves * 64), lds, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_leaf_kernel");
    return DPK_OK;
}

template <int DIST>
static int leaf_forward_dispatch(const LeafArgs &a, hipStream_t st) {
    // largest channel block in {8,4,2,1} dividing I; two samples per lane while registers allow
    const int I = a.I;
    const bool q4 = (a.R % 4) == 0;
    if (I % 8 == 0) return q4 ? launch_leaf<DIST, 4, 8, 1, 0, 1>(a, st) : launch_leaf<DIST, 2, 8, 1, 0, 1>(a, st);
    if (I % 4 == 0) return q4 ? launch_leaf<DIST, 4, 4, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 4, 2, 0, 1>(a, st);
    if (I % 2 == 0) return q4 ? launch_leaf<DIST, 4, 2, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 2, 2, 0, 1>(a, st);
    return q4 ? launch_leaf<DIST, 4, 1, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 1, 2, 0, 1>(a, st);
}

static int leaf_forward_common(int dist, const float *x, int64_t B, int32_t D, const int64_t *mask,
                               const uint8_t *pad_mask, const float *p0, const float *p1, int32_t R,
                               int32_t I, int32_t d, float *out, void *ws, int64_t ws_bytes,
                               uint32_t flags, void *stream) {
    DPK_REQUIRE(x && mask && p0 && out && ws, DPK_EINVAL, "leaf_forward: null pointer");
    DPK_REQUIRE(dist == 1 || p1, DPK_EINVAL, "leaf_forward: null scale");
    DPK_REQUIRE(B >= 0 && D > 0 && R > 0 && I > 0 && d > 0, DPK_EINVAL, "leaf_forward: bad sizes");
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, 0, 0, 0, 0);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "leaf_forward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    if (B == 0) return DPK_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = prepare_leaf_tables(dist, w, mask, pad_mask, p0, p1, R, I, d, flags, st);
    if (rc) return rc;
    LeafArgs a{};
    a.x = x; a.B = B; a.D = D; a.R = R; a.I = I; a.d = d; a.NC = w.NC; a.dP = w.dP;
    a.fl = as_const(w.fl); a.cb = as_const(w.cb); a.par = as_const(w.par); a.cel = as_const(w.cel);
    a.biasc = as_const(w.biasc);
    a.leaf_out = out;
    return dist == 0 ? leaf_forward_dispatch<0>(a, st) : leaf_forward_dispatch<1>(a, st);
}

template <int DEPTH, int I, int S>
static int fused_launch(const LeafArgs &a, hipStream_t st) {
    constexpr int SPL = (I <= 4) ? 2 : 1;
    return launch_leaf<0, (1 << DEPTH), I, SPL, DEPTH, S>(a, st);
}

template <int DEPTH, int I>
static int fused_dispatch_s(const LeafArgs &a, int S, hipStream_t st) {
    if (DEPTH == 1) return fused_launch<DEPTH, I, I>(a, st);  // no sum layer: S unused
    switch (S) {
        case 2: return fused_launch<DEPTH, I, 2>(a, st);
        case 4: return fused_launch<DEPTH, I, 4>(a, st);
        case 8: return fused_launch<DEPTH, I, 8>(a, st);
    }
    set_error("ratspn_forward: sums=%d not built (2,4,8)", S);
    return DPK_EUNSUPPORTED;
}

template <int DEPTH>
static int fused_dispatch_i(const LeafArgs &a, int S, hipStream_t st) {
    switch (a.I) {
        case 2: return fused_dispatch_s<DEPTH, 2>(a, S, st);
        case 4: return fused_dispatch_s<DEPTH, 4>(a, S, st);
        case 8: return fused_dispatch_s<DEPTH, 8>(a, S, st);
    }
    set_error("ratspn_forward: channels=%d not built (2,4,8)", a.I);
    return DPK_EUNSUPPORTED;
}

}  // namespace dpk

using namespace dpk;

extern "C" int64_t dpk_ratspn_workspace_bytes(int32_t in_features, int32_t regions, int32_t dimension,
                                              int32_t channels, int32_t depth, int32_t reps,
                                              int32_t sums, int32_t classes) {
    if (in_features <= 0 || regions <= 0 || dimension <= 0 || channels <= 0) return DPK_EINVAL;
    return carve_ratspn_ws(nullptr, in_features, regions, dimension, channels, depth, reps, sums, classes)
        .bytes;
}

extern "C" int dpk_gaussian_leaf_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                                         const uint8_t *pad_mask, const float *loc, const float *scale,
                                         int32_t R, int32_t I, int32_t d, float *out, void *ws,
                                         int64_t ws_bytes, uint32_t flags, void *stream) {
    return leaf_forward_common(0, x, B, D, mask, pad_mask, loc, scale, R, I, d, out, ws, ws_bytes, flags,
                               stream);
}

extern "C" int dpk_bernoulli_leaf_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                                          const uint8_t *pad_mask, const float *logits, int32_t R,
                                          int32_t I, int32_t d, float *out, void *ws, int64_t ws_bytes,
                                          uint32_t flags, void *stream) {
    return leaf_forward_common(1, x, B, D, mask, pad_mask, logits, nullptr, R, I, d, out, ws, ws_bytes,
                               flags, stream);
}

extern "C" int dpk_ratspn_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                                  const uint8_t *pad_mask, const float *loc, const float *scale,
                                  const float *sum_weight0, const float *sum_weight1,
                                  const float *root_weight, int32_t depth, int32_t reps, int32_t I,
                                  int32_t S, int32_t C, float *out, float *leaf_out, double *ll_sum,
                                  void *ws, int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(x && mask && loc && scale && root_weight && out && ws, DPK_EINVAL,
                "ratspn_forward: null pointer");
    DPK_REQUIRE(B >= 0 && D > 0 && reps > 0 && I > 0 && S > 0 && C > 0, DPK_EINVAL,
                "ratspn_forward: bad sizes");
    DPK_REQUIRE(depth >= 1 && depth <= 3, DPK_EUNSUPPORTED, "ratspn_forward: depth=%d not built (1..3)",
                depth);
    DPK_REQUIRE(C <= 64, DPK_EUNSUPPORTED, "ratspn_forward: classes=%d > 64", C);
    DPK_REQUIRE(depth < 2 || sum_weight0, DPK_EINVAL, "ratspn_forward: null sum_weight0");
    DPK_REQUIRE(depth < 3 || sum_weight1, DPK_EINVAL, "ratspn_forward: null sum_weight1");
    const int Q = 1 << depth;
    const int R = reps * Q;
    const int pad = (Q - D % Q) % Q;
    const int d = (D + pad) / Q;
    DPK_REQUIRE(pad == 0 || pad_mask, DPK_EINVAL, "ratspn_forward: padded model needs pad_mask");
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, depth, reps, S, C);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "ratspn_forward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    if (!(I == 2 || I == 4 || I == 8) || (depth >= 2 && !(S == 2 || S == 4 || S == 8))) {
        set_error("ratspn_forward: (channels=%d, sums=%d) not built; use the per-layer entry points", I, S);
        return DPK_EUNSUPPORTED;
    }
    if (B == 0) return DPK_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = prepare_leaf_tables(0, w, mask, pad_mask, loc, scale, R, I, d, flags, st);
    if (rc) return rc;
    const int nlast = depth >= 2 ? S : I;
    if (depth >= 2) {
        const int rows = reps * (Q / 2) * S;
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, sum_weight0, rows,
                           I * I, w.w[0], w.lw[0]);
    }
    if (depth >= 3) {
        const int rows = reps * (Q / 4) * S;
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, sum_weight1, rows,
                           S * S, w.w[1], w.lw[1]);
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, root_weight, C,
                       reps * nlast * nlast, w.w[2], w.lw[2]);
    DPK_CHECK_LAUNCH("softmax_rows_kernel");

    LeafArgs a{};
    a.x = x; a.B = B; a.D = D; a.R = R; a.I = I; a.d = d; a.NC = w.NC; a.dP = w.dP;
    a.fl = as_const(w.fl); a.cb = as_const(w.cb); a.par = as_const(w.par); a.cel = as_const(w.cel);
    a.biasc = as_const(w.biasc);
    a.leaf_out = leaf_out;
    a.reps = reps; a.C = C;
    a.W0 = as_const(w.w[0]); a.LW0 = as_const(w.lw[0]); a.W1 = as_const(w.w[1]);
    a.LW1 = as_const(w.lw[1]); a.Wr = as_const(w.w[2]); a.LWr = as_const(w.lw[2]);
    a.out = out; a.ll_sum = ll_sum;
    switch (depth) {
        case 1: return fused_dispatch_i<1>(a, S, st);
        case 2: return fused_dispatch_i<2>(a, S, st);
        default: return fused_dispatch_i<3>(a, S, st);
    }
}
