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
#include "ratspn_nodes.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace dpk {


// --------------------------------------------------------------------------------------
// structure tables: sort every region's variable ids (the sum over a region is order
// free) so that the entries falling in LDS chunk c are one contiguous range.
// reference: RegionGraphLayer.__init__ mask / pad_mask, deeprob/spn/layers/ratspn.py:42-56
// --------------------------------------------------------------------------------------
__global__ void ratspn_struct_kernel(const int64_t *__restrict__ mask,
                                     const uint8_t *__restrict__ pad, int R, int d, int NC, int QB, int SP,
                                     int *__restrict__ fl1, int *__restrict__ fl2,
                                     int *__restrict__ srcr, int *__restrict__ feat,
                                     int *__restrict__ nblk, int *__restrict__ segoff_out) {
    // One block per group g of QB consecutive regions (for the fused model: one repetition).
    // The group's entries are laid out in CONSUMPTION order of the wave that owns it:
    //   for chunk c: for region slot q: the variables of region g*QB+q that fall in chunk c,
    // every (c,q) segment padded to a multiple of kBlock with neutral entries.  One contiguous
    // stream per group lets the software pipeline run across segment and chunk boundaries.
    extern __shared__ int sm_i[];
    int *sid = sm_i;                    // [QB*d] sorted ids per region
    int *ssrc = sid + QB * d;           // [QB*d] r*d + j of the sorted entry, -1 for a dummy variable
    int *lo = ssrc + QB * d;            // [QB*(NC+1)] first sorted position of every chunk
    int *segoff = lo + QB * (NC + 1);   // [NC*QB+1] first stream position of every segment
    const int g = blockIdx.x;
    for (int e = threadIdx.x; e < QB * d; e += blockDim.x) {
        const int q = e / d, j = e - q * d;
        const int r = g * QB + q;
        const int64_t *row = mask + (int64_t)r * d;
        const int key = (int)row[j];
        int rank = 0;
        for (int jj = 0; jj < d; ++jj) {
            const int kk = (int)row[jj];
            rank += (kk < key) || (kk == key && jj < j);
        }
        sid[q * d + rank] = key;
        ssrc[q * d + rank] = (pad != nullptr && pad[(int64_t)r * d + j]) ? -1 : r * d + j;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < QB * (NC + 1); e += blockDim.x) {
        const int q = e / (NC + 1), c = e - q * (NC + 1);
        const int lim = c * kChunk;
        int l = 0, h = d;  // first position with sid[q][pos] >= lim
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (sid[q * d + mid] < lim) l = mid + 1; else h = mid;
        }
        lo[e] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < NC; ++c)
            for (int q = 0; q < QB; ++q) {
                const int n = lo[q * (NC + 1) + c + 1] - lo[q * (NC + 1) + c];
                const int nb = (n + kBlock - 1) / kBlock;
                segoff[c * QB + q] = run;
                nblk[((int64_t)g * NC + c) * QB + q] = nb;
                segoff_out[((int64_t)g * NC + c) * QB + q] = run;
                run += nb * kBlock;
            }
        segoff[NC * QB] = run;
    }
    __syncthreads();
    const int total = segoff[NC * QB];
    for (int p = threadIdx.x; p < SP; p += blockDim.x) {
        int f = -1, sj = -1, l = kChunk;
        if (p < total) {
            int s0 = 0, s1 = NC * QB;  // last segment with segoff[s] <= p
            while (s1 - s0 > 1) {
                const int mid = (s0 + s1) >> 1;
                if (segoff[mid] <= p) s0 = mid; else s1 = mid;
            }
            const int c = s0 / QB, q = s0 - c * QB;
            const int idx = lo[q * (NC + 1) + c] + (p - segoff[s0]);
            if (idx < lo[q * (NC + 1) + c + 1]) {
                f = sid[q * d + idx];
                sj = ssrc[q * d + idx];
                // a dummy (padding) variable reads the all-zero row like a neutral entry: the
                // unit-scale path applies no per-entry multiplier that could mask it
                if (sj >= 0) l = f % kChunk;
            }
        }
        const int64_t o = (int64_t)g * SP + p;
        feat[o] = f;
        srcr[o] = sj;
        fl1[o] = l * 65 * 4;   // TileGeom<1>::ROWB
        fl2[o] = l * 130 * 4;  // TileGeom<2>::ROWB
    }
}

// --------------------------------------------------------------------------------------
// parameter tables, rebuilt on every call from the live nn.Parameter storage.
//   Gaussian : term = a*(x-mu)^2 + c,  a = -1/(2 sigma^2), c = -log sigma - log sqrt(2 pi)
//              (torch.distributions.Normal.log_prob as used at ratspn.py:96)
//   Bernoulli: term = x*l - softplus(l) = -BCEWithLogits(l, x)   (ratspn.py:243)
// --------------------------------------------------------------------------------------
// One launch prepares everything that depends on the live parameters: blocks [0, G*kPrepSlices) build
// the leaf tables (kPrepSlices blocks per region group, each a slice of the group's entry stream),
// the blocks behind them take the rows of the sum / root weight matrices (softmax + log_softmax).
constexpr int kPrepSlices = 32;

struct PrepArgs {
    // leaf tables
    const float *p0, *p1;
    const int *srcr, *nblk, *segoff, *fl2;
    int R, I, CB, d, NC, QB, SP, G;
    float *par, *cel, *biasc, *biasx, *rec;
    int rec_compact;   // 1: 48-byte records {mean[4][2], u16 row offset[4], pad} (unit-scale hint, I == 2)
    int *unit;
    // up to three weight matrices [rows, n] -> W (softmax), LW (log_softmax)
    const float *w[3];
    float *W[3], *LW[3];
    int rows[3], n[3];
};

__device__ __forceinline__ void leaf_entry_params(int dist, const float *p0, const float *p1, int64_t o, float &A,
                                                  float &Bv, float &Cc) {
    if (dist == 0) {
        const float mu = p0[o], sg = p1[o];
        A = mu;
        Bv = -0.5f / (sg * sg);
        Cc = -logf(sg) - kLogSqrt2Pi;
    } else {
        const float l = p0[o];
        A = l;
        Bv = 0.f;
        Cc = -(fmaxf(l, 0.f) + log1pf(expf(-fabsf(l))));
    }
}

template <int DIST>
__global__ __launch_bounds__(256) void ratspn_prep_kernel(const PrepArgs a) {
    // block roles (all independent, so their dependent-load chains overlap instead of adding up):
    //   [0, nT)            leaf tables, slice sl of group g
    //   [nT, 2 nT)         per-(region, chunk) constants, slice sl of group g
    //   [2 nT, 2 nT + G)   unit-scale / bounded-mean flags of group g
    //   rest               softmax rows
    const int nT = a.G * kPrepSlices;
    const int n_leaf_blocks = 2 * nT + a.G;
    if ((int)blockIdx.x >= n_leaf_blocks) {
        // ---- softmax rows: one wave per row (reference: torch.log_softmax at ratspn.py:375 and :455)
        int row = (blockIdx.x - n_leaf_blocks) * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (row < a.rows[m]) {
                const int n = a.n[m];
                const float *src = a.w[m] + (int64_t)row * n;
                float mx = -INFINITY;
                for (int i = lane; i < n; i += 64) mx = fmaxf(mx, src[i]);
                mx = wave_reduce_max(mx);
                float sum = 0.f;
                for (int i = lane; i < n; i += 64) sum += expf(src[i] - mx);
                sum = wave_reduce_sum(sum);
                const float ls = logf(sum);
                for (int i = lane; i < n; i += 64) {
                    const float l = src[i] - mx - ls;
                    a.LW[m][(int64_t)row * n + i] = l;
                    a.W[m][(int64_t)row * n + i] = expf(l);
                }
                return;
            }
            row -= a.rows[m];
        }
        return;
    }
    // ---- leaf tables: [group][channel block][stream position]{p0[CB], p1[CB]}; the kBlock entries of
    // a block are one contiguous run for the wave that owns (group, channel block)
    const int role = (int)blockIdx.x < nT ? 0 : ((int)blockIdx.x < 2 * nT ? 1 : 2);
    const int bid = (int)blockIdx.x - role * nT;
    const int g = role == 2 ? bid : bid / kPrepSlices, sl = role == 2 ? 0 : bid - g * kPrepSlices;
    const int I = a.I, CB = a.CB, d = a.d, SP = a.SP, QB = a.QB, NC = a.NC;
    const int ncb = I / CB;
    const int per = ((SP / kBlock + kPrepSlices - 1) / kPrepSlices) * kBlock;  // positions per slice
    const int p_lo = sl * per, p_hi = min(SP, p_lo + per);
    for (int e = threadIdx.x; role == 0 && e < (p_hi - p_lo) * I; e += blockDim.x) {
        const int pidx = p_lo + e / I, k = e % I;
        const int rj = a.srcr[(int64_t)g * SP + pidx];
        float A = 0.f, Bv = 0.f, Cc = 0.f;
        if (rj >= 0) {
            const int r = rj / d, j = rj - r * d;
            leaf_entry_params(DIST, a.p0, a.p1, ((int64_t)r * I + k) * d + j, A, Bv, Cc);
        }
        const int kbi = k / CB, kk = k - kbi * CB;
        const int64_t ent = ((int64_t)g * ncb + kbi) * SP + pidx;
        a.par[ent * 2 * CB + kk] = A;
        a.par[ent * 2 * CB + CB + kk] = Bv;
        a.cel[ent * CB + kk] = Cc;
        if (a.rec != nullptr && a.rec_compact) {
            // compact records (CompactPipe): {mean[4][2], row offsets as 4 x u16, 8 bytes of padding}
            const int blk = pidx / kBlock, u = pidx - blk * kBlock;
            float *rp = a.rec + ((int64_t)g * (SP / kBlock) + blk) * (kCompactRec / 4);
            rp[u * 2 + k] = A;
            if (k == 0 && (u & 1) == 0) {
                const int o0 = a.fl2[(int64_t)g * SP + pidx], o1 = a.fl2[(int64_t)g * SP + pidx + 1];
                rp[8 + (u >> 1)] = __int_as_float((o0 & 0xffff) | (o1 << 16));
            }
        } else if (a.rec != nullptr) {
            // block records for the LDS-resident tables (I == CB <= 2): {row offsets[4], p0[4][CB],
            // p1[4][CB]} -- see LdsPipe
            const int RECB = 4 + 8 * CB;
            const int blk = pidx / kBlock, u = pidx - blk * kBlock;
            float *rp = a.rec + ((int64_t)g * (SP / kBlock) + blk) * RECB;
            rp[4 + u * CB + k] = A;
            rp[4 + 4 * CB + u * CB + k] = Bv;
            if (k == 0) rp[u] = __int_as_float(a.fl2[(int64_t)g * SP + pidx]);
        }
    }
    // per-(region, chunk) constants: segments are dealt round-robin to the slices, one wave per
    // (segment, channel), lanes over the entries, summed from the parameters directly (no dependence
    // on what the other slices write)
    if (role == 1) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int e = wv; e < NC * QB * I; e += 4) {
            const int k = e % I, cq = e / I;
            if (cq % kPrepSlices != sl) continue;
            const int c = cq / QB, q = cq - c * QB;
            const int64_t so = ((int64_t)g * NC + c) * QB + q;
            const int j0 = a.segoff[so], j1 = j0 + a.nblk[so] * kBlock;
            float s = 0.f, s2 = 0.f;
            for (int p = j0 + lane; p < j1; p += 64) {
                const int rj = a.srcr[(int64_t)g * SP + p];
                if (rj >= 0) {
                    const int r = rj / d, j = rj - r * d;
                    float A, Bv, Cc;
                    leaf_entry_params(DIST, a.p0, a.p1, ((int64_t)r * I + k) * d + j, A, Bv, Cc);
                    s += Cc;
                    s2 = fmaf(A, A, s2);
                }
            }
            s = wave_reduce_sum(s);
            s2 = wave_reduce_sum(s2);
            if (lane == 0) {
                a.biasc[((int64_t)(g * QB + q) * NC + c) * I + k] = s;
                a.biasx[((int64_t)(g * QB + q) * NC + c) * I + k] = s - 0.5f * s2;
            }
        }
    }
    // unit-scale flag per region (one block scans the group's scales and means)
    if (role == 2) {
        __shared__ int not_unit[8], big_mean[8];
        if (threadIdx.x < 8) {
            not_unit[threadIdx.x] = (DIST != 0);
            big_mean[threadIdx.x] = 0;
        }
        __syncthreads();
        if (DIST == 0) {
            for (int e = threadIdx.x; e < QB * I * d; e += blockDim.x) {
                const int q = e / (I * d);
                if (a.p1[(int64_t)g * QB * I * d + e] != 1.0f) not_unit[q] = 1;
                if (!(fabsf(a.p0[(int64_t)g * QB * I * d + e]) <= kExpandBound)) big_mean[q] = 1;
            }
        }
        __syncthreads();
        if (threadIdx.x < QB)
            a.unit[g * QB + threadIdx.x] = not_unit[threadIdx.x] ? 0 : (big_mean[threadIdx.x] ? 1 : 2);
    }
}


// --------------------------------------------------------------------------------------
// kernel arguments
// --------------------------------------------------------------------------------------
struct LeafArgs {
    const float *x;
    int64_t B;
    int D, R, I, d, NC, SP;
    cint_p fl1, fl2;  // LDS byte offsets of the rows for SPL = 1 / 2
    cint_p nblk;      // [G][NC][QB] blocks per segment
    cint_p segoff;    // [G][NC][QB] first stream position of every segment
    const float *rec; // [G][SP/4] block records for the LDS-resident tables (CB <= 2)
    int tabcap;       // bytes of LDS per wave for the records of one chunk
    int tabcap_c;     // the same for compact records (host side only: launch_leaf moves it into tabcap)
    int unit_hint;    // host hint: every scale is 1 (DPK_FLAG_UNIT_SCALE)
#ifdef DPK_TIMELINE
    unsigned long long *dbg;  // [blocks][waves][NC+2][6] s_memtime stamps (measurement builds)
#endif
    cfloat_p par;
    cfloat_p cel;
    cfloat_p biasc;
    cfloat_p biasx;   // biasc - 0.5 * sum mu^2 (expanded unit-scale form)
    cint_p unit;      // [R] 1: every scale of the region is exactly 1; 2: and every |mean| <= kExpandBound
    float *leaf_out;  // [B,R,I] or nullptr
    // fused model
    int reps, C;
    cfloat_p W0, LW0, W1, LW1, Wr, LWr;
    float *out;      // [B,C]
    double *ll_sum;  // [2] ([17] with DPK_FLAG_LL_SUM_SPREAD) or nullptr
    int ll_cnt;      // index of the count: 1 or 16
    // last, so that the offsets of everything above stay what the kernels were tuned with
    int *slow_flag;   // host-mapped word: a work-group that meets a slow chunk stores launch_seq there
    int launch_seq;
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

// Work-group barrier that orders LDS traffic only: lgkmcnt(0) + s_barrier.  __syncthreads()
// carries a release fence that also waits vmcnt(0), which would stall every wave on the HBM
// prefetch of the NEXT chunk that is deliberately in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// lgkmcnt(0) with vmcnt / expcnt left alone (gfx9 encoding: vmcnt 0x3f, expcnt 7, lgkmcnt 0)
#define DPK_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// DPK_ABLATE (measurement builds only, never shipped): 1 = constant parameters, 2 = constant
// parameters and row offsets (no SMEM in the inner loop), 3 = no LDS reads of x, 4 = no HBM staging,
// 5 = no LDS reads of the block records in the expanded pipeline
#ifndef DPK_ABLATE
#define DPK_ABLATE 0
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Scalar-memory loads issued as inline asm.  The compiler schedules a C++ load "as late as
// legal" and, with SMEM (out-of-order return) in flight, turns every wait into lgkmcnt(0) right
// behind the issue; hand-issued loads plus ONE explicit lgkmcnt(0) at the end of each pipeline
// step keep a full step of VALU work between issue and use.  hipcc does not count these loads
// (cdna_hip_programming.md 5.7): every destination is consumed only behind DPK_WAIT_LGKM0().
template <int OFF> __device__ __forceinline__ i32x4 sload_i4(cint_p p) {
    i32x4 r;
    if (DPK_ABLATE == 2) return (i32x4){0, 520, 1040, 1560};
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF) : "memory");
    return r;
}
template <int NDW> struct SVec;
template <> struct SVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct SVec<4> { typedef f32x4 type; };
template <> struct SVec<8> { typedef f32x8 type; };
template <> struct SVec<16> { typedef f32x16 type; };
template <int NDW, int OFF> __device__ __forceinline__ typename SVec<NDW>::type sload_f(cfloat_p p) {
    typename SVec<NDW>::type r;
    if constexpr (NDW == 2) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF) : "memory");
    if constexpr (NDW == 4) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF) : "memory");
    if constexpr (NDW == 8) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF) : "memory");
    if constexpr (NDW == 16) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r) : "s"(p), "i"(OFF) : "memory");
    return r;
}

// One block of kBlock table entries.  Parameters of the block are one contiguous run of
// kBlock*2*CB dwords ({p0[CB], p1[CB]} per entry) fetched in pieces of at most 16 dwords.
template <int CB> struct ParBlock {
    static constexpr int NDW = kBlock * 2 * CB;           // 8, 16, 32, 64
    static constexpr int PIECE = NDW < 16 ? NDW : 16;
    static constexpr int NP = NDW / PIECE;
    typename SVec<PIECE>::type v[NP];
    __device__ __forceinline__ float get(int i) const { return v[i / PIECE][i % PIECE]; }
    __device__ __forceinline__ float p0(int u, int k) const { return get(u * 2 * CB + k); }
    __device__ __forceinline__ float p1(int u, int k) const { return get(u * 2 * CB + CB + k); }
};
template <int CB, int PIDX = 0>
__device__ __forceinline__ void par_load(ParBlock<CB> &b, cfloat_p pp) {
    if (DPK_ABLATE == 1 || DPK_ABLATE == 2) return;
    if constexpr (PIDX < ParBlock<CB>::NP) {
        b.v[PIDX] = sload_f<ParBlock<CB>::PIECE, PIDX * ParBlock<CB>::PIECE * 4>(pp);
        par_load<CB, PIDX + 1>(b, pp);
    }
}
// additive constants of the block (exact path only): kBlock*CB dwords
template <int CB> struct CelBlock {
    static constexpr int NDW = kBlock * CB;               // 4, 8, 16, 32
    static constexpr int PIECE = NDW < 16 ? NDW : 16;
    static constexpr int NP = NDW / PIECE;
    typename SVec<PIECE>::type v[NP];
    __device__ __forceinline__ float pc(int u, int k) const { return v[(u * CB + k) / PIECE][(u * CB + k) % PIECE]; }
};
template <int CB, int PIDX = 0>
__device__ __forceinline__ void cel_load(CelBlock<CB> &b, cfloat_p cp) {
    if constexpr (PIDX < CelBlock<CB>::NP) {
        b.v[PIDX] = sload_f<CelBlock<CB>::PIECE, PIDX * CelBlock<CB>::PIECE * 4>(cp);
        cel_load<CB, PIDX + 1>(b, cp);
    }
}

// Make the accumulators opaque at this point: the FMAs that produce them cannot be sunk below
// (IR-level code motion would otherwise carry a whole block of compute across the wait that
// follows and expose the load latency again).
template <int CB, int SPL> __device__ __forceinline__ void pin_acc(float (&acc)[CB][SPL]) {
#pragma unroll
    for (int k = 0; k < CB; ++k)
#pragma unroll
        for (int s = 0; s < SPL; ++s) asm volatile("" : "+v"(acc[k][s]));
}

template <int SPL>
__device__ __forceinline__ void leaf_read_x(float (&x)[kBlock][SPL], const char *lane_base, const i32x4 off) {
#pragma unroll
    for (int u = 0; u < kBlock; ++u) {
        if (DPK_ABLATE == 3) {
#pragma unroll
            for (int s = 0; s < SPL; ++s) x[u][s] = __int_as_float(off[u] + s);
            continue;
        }
        if (SPL == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(lane_base + off[u]);
            x[u][0] = v.x;
            x[u][SPL - 1] = v.y;
        } else {
            x[u][0] = *reinterpret_cast<const float *>(lane_base + off[u]);
        }
    }
}

// MODE 0: finite inputs, acc += a (x-mu)^2            (3 VALU / element)
// MODE 1: exact per-element form with nan_to_num (NaN / inf evidence in the chunk)
// MODE 2: finite inputs and sigma == 1 for the whole region, acc += (x-mu)^2 (2 VALU / element;
//         the caller scales by -1/2 once per chunk)
template <int DIST, int CB, int SPL, int MODE>
__device__ __forceinline__ void leaf_block_compute(float (&acc)[CB][SPL], const float (&x)[kBlock][SPL],
                                                   const ParBlock<CB> &par, const CelBlock<CB> &cel) {
    constexpr bool SLOW = (MODE == 1);
#pragma unroll
    for (int u = 0; u < kBlock; ++u)
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const float p0 = (DPK_ABLATE == 1 || DPK_ABLATE == 2) ? 0.25f * (float)(k + 1) : par.p0(u, k);
            const float p1 = (DPK_ABLATE == 1 || DPK_ABLATE == 2) ? -0.5f : par.p1(u, k);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                if (DIST == 0) {
                    const float dlt = x[u][s] - p0;
                    if (MODE == 2) acc[k][s] = fmaf(dlt, dlt, acc[k][s]);
                    else if (!SLOW) acc[k][s] = fmaf(dlt * dlt, p1, acc[k][s]);
                    else acc[k][s] += nan_to_num_f(fmaf(dlt * dlt, p1, cel.pc(u, k)));
                } else {
                    if (!SLOW) acc[k][s] = fmaf(x[u][s], p0, acc[k][s]);
                    else acc[k][s] += nan_to_num_f(fmaf(x[u][s], p0, cel.pc(u, k)));
                }
            }
        }
}

// Accumulate `nblk` blocks of table entries of one region over the chunk held in LDS.
// flp: byte offsets of the LDS rows (premultiplied), pp / cp: parameters / constants, all
// advancing by one block per step.  Software-pipelined over two register sets: while block b is
// consumed, the x reads + parameters of block b+1 and the row offsets of block b+2 are in flight;
// the only wait is the explicit lgkmcnt(0) that ends each step.
// Plain version: compiler-issued scalar loads, one block at a time.  Used for the exact (NaN / inf)
// path and for wide channel blocks, whose parameter sets do not fit twice in the SGPR file (the
// hand-pipelined version below would make the register allocator spill SGPRs that an asm load is
// still writing).
template <int DIST, int CB, int SPL, int MODE>
__device__ __forceinline__ void leaf_accum_plain(float (&acc)[CB][SPL], const char *lane_base, cint_p flp,
                                                 cfloat_p pp, cfloat_p cp, int nblk) {
    constexpr bool SLOW = (MODE == 1);
    for (int b = 0; b < nblk; ++b) {
        float x[kBlock][SPL];
        i32x4 off;
#pragma unroll
        for (int u = 0; u < kBlock; ++u) off[u] = flp[u];
        leaf_read_x<SPL>(x, lane_base, off);
#pragma unroll
        for (int u = 0; u < kBlock; ++u)
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const float p0 = pp[u * 2 * CB + k];
                const float p1 = pp[u * 2 * CB + CB + k];
                const float pc = SLOW ? cp[u * CB + k] : 0.f;
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    if (DIST == 0) {
                        const float dlt = x[u][s] - p0;
                        if (MODE == 2) acc[k][s] = fmaf(dlt, dlt, acc[k][s]);
                        else if (!SLOW) acc[k][s] = fmaf(dlt * dlt, p1, acc[k][s]);
                        else acc[k][s] += nan_to_num_f(fmaf(dlt * dlt, p1, pc));
                    } else {
                        if (!SLOW) acc[k][s] = fmaf(x[u][s], p0, acc[k][s]);
                        else acc[k][s] += nan_to_num_f(fmaf(x[u][s], p0, pc));
                    }
                }
            }
        flp += kBlock;
        pp += kBlock * 2 * CB;
        cp += kBlock * CB;
    }
}

// Expanded unit-scale form on the scalar-cache tables (wide channel blocks): acc[k] += x mu_k, one FMA per
// (entry, channel, sample) with the mean as a scalar operand, instead of sub / mul / fma.  WITHQ: also q += x^2
// (leaf-only launches; the fused model factors the x^2 sums out, see qfree in the kernel).
template <int CB, int SPL, bool WITHQ>
__device__ __forceinline__ void leaf_accum_linear(float (&acc)[CB][SPL], float (&q)[SPL], const char *lane_base,
                                                  cint_p flp, cfloat_p pp, int nblk) {
    for (int b = 0; b < nblk; ++b) {
        float x[kBlock][SPL];
        i32x4 off;
#pragma unroll
        for (int u = 0; u < kBlock; ++u) off[u] = flp[u];
        leaf_read_x<SPL>(x, lane_base, off);
#pragma unroll
        for (int u = 0; u < kBlock; ++u) {
#pragma unroll
            for (int s = 0; s < SPL; ++s)
                if (WITHQ) q[s] = fmaf(x[u][s], x[u][s], q[s]);
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const float mu = pp[u * 2 * CB + k];
#pragma unroll
                for (int s = 0; s < SPL; ++s) acc[k][s] = fmaf(x[u][s], mu, acc[k][s]);
            }
        }
        flp += kBlock;
        pp += kBlock * 2 * CB;
    }
}

// Software pipeline over the table blocks of one (wave, chunk), tables resident in LDS (CB <= 2).
//
// Why LDS and not the scalar cache for the hot loop: an s_load that misses costs ~750-900 cycles
// here (PMC SmemLatency), returns out of order (so every wait is a full lgkmcnt(0)) and one step of
// four entries is only ~140 cycles of VALU work per wave -- with four waves per SIMD the scalar
// latency cannot be covered.  An LDS broadcast read (all lanes, same address) returns in ~130
// cycles, in order, so the compiler's counted waits pipeline it, and the operands arrive in VGPRs
// (plain VOP2 encodings).  The block record (see leaf_param_kernel):
//     dwords 0..3            LDS byte offsets of the 4 entries' x rows
//     dwords 4..4+4CB-1      p0 (mean)        [entry][channel]
//     dwords 4+4CB..4+8CB-1  p1 (-1/2sigma^2) [entry][channel]
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CB> struct RecGeom {
    static constexpr int RECB = 4 + 8 * CB;   // dwords per block record
    static constexpr int RECBB = RECB * 4;    // bytes
};

template <int CB, int SPL, bool GENERAL> struct LdsPipe {
    typedef float pvec __attribute__((ext_vector_type(4 * CB)));  // p0 or p1 of one block
    static constexpr int RECBB = RecGeom<CB>::RECBB;
    float x[2][kBlock][SPL];
    pvec mu[2], av[2];
    i32x4 offc, offn;
    const char *tbn;  // record of block b+1

    template <int SET> __device__ __forceinline__ void load_par(const char *rec) {
        if (DPK_ABLATE == 5) {   // measurement: no record reads
#pragma unroll
            for (int i = 0; i < 4 * CB; ++i) mu[SET][i] = 0.25f * (float)(i + 1);
            return;
        }
        mu[SET] = *reinterpret_cast<const pvec *>(rec + 16);
        if (GENERAL) av[SET] = *reinterpret_cast<const pvec *>(rec + 16 + 16 * CB);
    }
    __device__ __forceinline__ void prime(const char *tb0, const char *lane_base) {
        offc = *reinterpret_cast<const i32x4 *>(tb0);
        load_par<0>(tb0);
        offn = *reinterpret_cast<const i32x4 *>(tb0 + RECBB);
        leaf_read_x<SPL>(x[0], lane_base, offc);
        tbn = tb0 + RECBB;
    }
    template <int DIST, int MODE, int CUR>
    __device__ __forceinline__ void step(float (&acc)[CB][SPL], const char *lane_base) {
        constexpr int OTH = 1 - CUR;
        leaf_read_x<SPL>(x[OTH], lane_base, offn);
        offc = offn;
        offn = *reinterpret_cast<const i32x4 *>(tbn + RECBB);
        load_par<OTH>(tbn);
        tbn += RECBB;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kBlock; ++u)
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const float p0 = mu[CUR][u * CB + k];
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    if (DIST == 0) {
                        const float dlt = x[CUR][u][s] - p0;
                        if (MODE == 2) acc[k][s] = fmaf(dlt, dlt, acc[k][s]);
                        else acc[k][s] = fmaf(dlt * dlt, av[CUR][u * CB + k], acc[k][s]);
                    } else {
                        acc[k][s] = fmaf(x[CUR][u][s], p0, acc[k][s]);
                    }
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // Expanded unit-scale form (MODE 3, SPL == 2): sum (x-mu)^2 = Q - 2 P + sum mu^2 with Q = sum x^2 (one
    // packed FMA per entry, shared by the channels) and P[k] = sum x mu_k (one packed FMA per entry and
    // channel, the mean broadcast to both samples by op_sel): 4 + 4*CB v_pk_fma_f32 per block instead of
    // 16*CB VOP2.  Only taken when |x| and |mu| are bounded by kExpandBound (cancellation).
    template <int CUR, bool WITHQ>
    __device__ __forceinline__ void step3(f32x2 (&P)[CB], f32x2 &Q, const char *lane_base) {
        constexpr int OTH = 1 - CUR;
        leaf_read_x<SPL>(x[OTH], lane_base, offn);
        offc = offn;
        if (DPK_ABLATE == 5) offn = (i32x4){0, 520, 1040, 1560};
        else offn = *reinterpret_cast<const i32x4 *>(tbn + RECBB);
        load_par<OTH>(tbn);
        tbn += RECBB;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kBlock; ++u) {
            const f32x2 xv = {x[CUR][u][0], x[CUR][u][SPL - 1]};
            if (WITHQ) Q = __builtin_elementwise_fma(xv, xv, Q);
            if constexpr (CB == 2) {
                // the two channel means of an entry sit in one aligned register pair: broadcast its low / high
                // half to both samples with op_sel (the compiler otherwise copies the high half into a fresh pair)
                const f32x2 mp = {mu[CUR][2 * u], mu[CUR][2 * u + 1]};
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P[0]) : "v"(xv), "v"(mp));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(P[1]) : "v"(xv), "v"(mp));
            } else {
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const float m = mu[CUR][u * CB + k];
                    P[k] = __builtin_elementwise_fma(xv, (f32x2){m, m}, P[k]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <bool WITHQ>
    __device__ __forceinline__ void run3(f32x2 (&P)[CB], f32x2 &Q, const char *lane_base, int nb) {
        for (int i = nb >> 1; i > 0; --i) {
            step3<0, WITHQ>(P, Q, lane_base);
            step3<1, WITHQ>(P, Q, lane_base);
        }
        if (nb & 1) {
            step3<0, WITHQ>(P, Q, lane_base);
            mu[0] = mu[1];
#pragma unroll
            for (int u = 0; u < kBlock; ++u)
#pragma unroll
                for (int s = 0; s < SPL; ++s) x[0][u][s] = x[1][u][s];
        }
    }
    // nb blocks into acc; register sets alternate statically, an odd count ends with one set move
    template <int DIST, int MODE>
    __device__ __forceinline__ void run(float (&acc)[CB][SPL], const char *lane_base, int nb) {
        for (int i = nb >> 1; i > 0; --i) {
            step<DIST, MODE, 0>(acc, lane_base);
            step<DIST, MODE, 1>(acc, lane_base);
        }
        if (nb & 1) {
            step<DIST, MODE, 0>(acc, lane_base);
            mu[0] = mu[1];
            if (GENERAL) av[0] = av[1];
#pragma unroll
            for (int u = 0; u < kBlock; ++u)
#pragma unroll
                for (int s = 0; s < SPL; ++s) x[0][u][s] = x[1][u][s];
        }
    }
};

// The unit-scale two-channel pipeline on compact records.  The accumulate loop is LDS-return-bandwidth bound
// (tools/ubench/leaf_loop.hip: 256 B/clk per CU whether or not the lanes share an address; a broadcast
// ds_read_b128 costs ~4.6 LDS cycles, a ds_read_b64 ~1.9), so the record is cut from 3 x b128 to 2 x b128 (the 8
// means) + 1 x b64 (the 4 row offsets as u16), the unpack folded into the address add by SDWA: 18.7 instead of
// 21.4 LDS cycles per block, 173 instead of 205 ns per block and wave in the microbenchmark.
__device__ __forceinline__ int sdwa_add_w0(int base, int pk) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(r) : "v"(base), "v"(pk));
    return r;
}
__device__ __forceinline__ int sdwa_add_w1(int base, int pk) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(r) : "v"(base), "v"(pk));
    return r;
}
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct CompactPipe {
    f32x2 x[2][kBlock];
    f32x8 mu[2];
    i32x2 offn;        // packed row offsets of block b+1
    const char *tbn;   // record of block b+1

    __device__ __forceinline__ void read_x(f32x2 (&dst)[kBlock], const char *smem0, int lane_off, i32x2 o) {
        dst[0] = *reinterpret_cast<const f32x2 *>(smem0 + sdwa_add_w0(lane_off, o[0]));
        dst[1] = *reinterpret_cast<const f32x2 *>(smem0 + sdwa_add_w1(lane_off, o[0]));
        dst[2] = *reinterpret_cast<const f32x2 *>(smem0 + sdwa_add_w0(lane_off, o[1]));
        dst[3] = *reinterpret_cast<const f32x2 *>(smem0 + sdwa_add_w1(lane_off, o[1]));
    }
    __device__ __forceinline__ void prime(const char *tb0, const char *smem0, int lane_off) {
        mu[0] = *reinterpret_cast<const f32x8 *>(tb0);
        read_x(x[0], smem0, lane_off, *reinterpret_cast<const i32x2 *>(tb0 + 32));
        offn = *reinterpret_cast<const i32x2 *>(tb0 + kCompactRec + 32);
        tbn = tb0 + kCompactRec;
    }
    template <int CUR, bool WITHQ>
    __device__ __forceinline__ void step(f32x2 (&P)[2], f32x2 &Q, const char *smem0, int lane_off) {
        constexpr int OTH = 1 - CUR;
        read_x(x[OTH], smem0, lane_off, offn);
        offn = *reinterpret_cast<const i32x2 *>(tbn + kCompactRec + 32);
        mu[OTH] = *reinterpret_cast<const f32x8 *>(tbn);
        tbn += kCompactRec;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kBlock; ++u) {
            const f32x2 xv = x[CUR][u];
            if (WITHQ) Q = __builtin_elementwise_fma(xv, xv, Q);
            const f32x2 mp = {mu[CUR][2 * u], mu[CUR][2 * u + 1]};


            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P[0]) : "v"(xv), "v"(mp));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(P[1]) : "v"(xv), "v"(mp));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // Exact per-entry form on the same records, for chunks that hold NaN / inf / out-of-bound evidence:
    // t = -(x-mu)^2/2 - log sqrt(2 pi) with nan_to_num semantics (NaN -> 0, -inf -> -FLT_MAX), complete terms (no
    // bias afterwards).  The caller fills the neutral LDS row with NaN for the chunk, so padding entries add 0.
    template <int CUR>
    __device__ __forceinline__ void step_exact(f32x2 (&A)[2], const char *smem0, int lane_off) {
        constexpr int OTH = 1 - CUR;
        read_x(x[OTH], smem0, lane_off, offn);
        offn = *reinterpret_cast<const i32x2 *>(tbn + kCompactRec + 32);
        mu[OTH] = *reinterpret_cast<const f32x8 *>(tbn);
        tbn += kCompactRec;
#pragma unroll
        for (int u = 0; u < kBlock; ++u) {
            const f32x2 xv = x[CUR][u];
            const bool n0 = xv[0] != xv[0], n1 = xv[1] != xv[1];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float mk = mu[CUR][2 * u + k];
                const float d0 = xv[0] - mk, d1 = xv[1] - mk;
                const float t0 = fmaxf(fmaf(d0 * d0, -0.5f, -kLogSqrt2Pi), -FLT_MAX);
                const float t1 = fmaxf(fmaf(d1 * d1, -0.5f, -kLogSqrt2Pi), -FLT_MAX);
                A[k][0] += n0 ? 0.f : t0;
                A[k][1] += n1 ? 0.f : t1;
            }
        }
    }
    __device__ __forceinline__ void run_exact(f32x2 (&A)[2], const char *smem0, int lane_off, int nb) {
        for (int i = nb >> 1; i > 0; --i) {
            step_exact<0>(A, smem0, lane_off);
            step_exact<1>(A, smem0, lane_off);
        }
        if (nb & 1) {
            step_exact<0>(A, smem0, lane_off);
            mu[0] = mu[1];
#pragma unroll
            for (int u = 0; u < kBlock; ++u) x[0][u] = x[1][u];
        }
    }
    template <bool WITHQ>
    __device__ __forceinline__ void run(f32x2 (&P)[2], f32x2 &Q, const char *smem0, int lane_off, int nb) {
        for (int i = nb >> 1; i > 0; --i) {
            step<0, WITHQ>(P, Q, smem0, lane_off);
            step<1, WITHQ>(P, Q, smem0, lane_off);
        }
        if (nb & 1) {
            step<0, WITHQ>(P, Q, smem0, lane_off);
            mu[0] = mu[1];
#pragma unroll
            for (int u = 0; u < kBlock; ++u) x[0][u] = x[1][u];
        }
    }
};


// DEPTH == 0: leaf only (QB regions x CB channels per wave item, written to leaf_out)
// DEPTH >= 1: fused model, QB == 2^DEPTH, CB == I, S sum nodes
#ifndef DPK_NO_EXPAND
#define DPK_NO_EXPAND 0
#endif
#ifdef DPK_FORCE_SPL1
#define DPK_MINW(SPL) 8
#else
#define DPK_MINW(SPL) 4
#endif
// GEN: the LDS-table pipeline carries the per-entry scale factors (general sigma).  With GEN = false
// (host hint "scale is frozen at 1", checked on the device per region) only the means travel, which
// keeps the kernel under 128 VGPRs without scratch; a group whose scales are not all 1 then takes
// the scalar-cache path, so the hint can only cost speed, never correctness.
template <int DIST, int QB, int CB, int SPL, int DEPTH, int S, bool GEN, bool XLDS = false>
__global__ __launch_bounds__(kLeafWaves * 64, DPK_MINW(SPL)) void ratspn_leaf_kernel(const LeafArgs a) {
    using G = TileGeom<SPL>;
    constexpr int T = G::T, ROW = G::ROW, NLD = G::NLD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs_lds = reinterpret_cast<float *>(smem);
    int *flags_lds = reinterpret_cast<int *>(smem + G::BUF_BYTES);  // [waves]
    float *run_m = reinterpret_cast<float *>(smem + G::BUF_BYTES + 64 + kLeafWaves * 2 * a.tabcap);  // [C][T] fused only
    float *run_s = run_m + (DEPTH > 0 ? a.C * T : 0);
    float *qlds = run_s + (DEPTH > 0 ? a.C * T : 0);   // [waves][T] partial sums of x^2 (fused model, see qfree)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * T;
    const int I = (DEPTH > 0) ? CB : a.I;  // static in the fused kernels
    const int R = a.R, NC = a.NC, D = a.D;

    const int n_cblk = (DEPTH > 0) ? 1 : I / CB;
    const int n_items = (DEPTH > 0) ? a.reps : (R / QB) * n_cblk;
    const int n_pass = (n_items + kLeafWaves - 1) / kLeafWaves;

    const char *lane_base = smem + lane * (4 * SPL);

#ifdef DPK_TIMELINE
    if (a.dbg && lane == 0) {
        unsigned long long *row = a.dbg + (((int64_t)blockIdx.x * kLeafWaves + wave) * (NC + 2) + NC) * 6;
        row[0] = __builtin_amdgcn_s_memtime();
        row[2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if constexpr (DEPTH > 0) {
        for (int e = tid; e < a.C * T; e += kLeafWaves * 64) {
            run_m[e] = -INFINITY;
            run_s[e] = 0.f;
        }
    }

    // Expanded unit-scale form (LdsPipe::step3): only when every region's means are bounded, and then every
    // staged |x| above the bound sends its tile to the exact path like a non-finite value does.
    // kernels with the LDS-table pipeline (unit hint, CB <= 2) or with wide channel blocks on the scalar-cache tables
    constexpr bool kWide = (CB > 2);
    bool expand_all;
    if constexpr (kWide)
        expand_all = (DIST == 0) && (DPK_NO_EXPAND == 0) && (DEPTH == 0 || a.leaf_out == nullptr);
    else
        expand_all = (DIST == 0) && !GEN && (SPL == 2) && a.tabcap > 0 && (DPK_NO_EXPAND == 0) &&
                     (DEPTH == 0 || a.leaf_out == nullptr);
    if (expand_all) {
        // lanes read the flags in parallel (a serial scalar loop costs ~150 ns per region)
        bool ok = true;
        for (int r = lane; r < R; r += 64) ok = ok && (a.unit[r] == 2);
        expand_all = __all(ok);
    }
    // a staged value v is "bad" iff bits(v*v) > thr_bits: NaN, +-inf, |v| > bound (or so large that
    // (v - mu)^2 could overflow when the direct form is used)
    const unsigned thr_bits = __float_as_uint(expand_all ? kExpandBound * kExpandBound : 1.0e37f);
    // Fused model, expanded form: the -1/2 sum_f x_f^2 of a region is common to all its channels, so it factors
    // out of every sum node above it; every repetition covers each variable once, so what reaches the root is
    // -1/2 sum over ALL variables of x^2 -- one scalar per sample, the same for every repetition.  The waves
    // therefore accumulate only P = sum x mu (8 instead of 12 packed FMAs per block) and share out the x^2 sums
    // (8 rows of each chunk per wave); the scalar is added to the root output.  Chunks that take the exact path
    // carry their complete terms and are left out of the scalar.  A fused call that also wants the leaf outputs
    // does not take the expanded form at all.
    const bool qfree = expand_all && (DEPTH > 0);   // expand_all excludes leaf_out for the fused model (below)
    f32x2 qpart = {0.f, 0.f};

    // leaf-only launches spread the passes over blockIdx.y (small training batches would otherwise occupy
    // B/T compute units); the fused model keeps them in the work-group (the root exchange spans them)
    bool saw_slow = false;
    const int pass_lo = (DEPTH == 0) ? (int)blockIdx.y : 0;
    const int pass_hi = (DEPTH == 0) ? min(pass_lo + 1, n_pass) : n_pass;
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
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
        constexpr int SQN = kLeafWaves * 64 / kChunk;  // samples staged per pass of the work-group
        const int flc = tid & (kChunk - 1);
        const int sq = tid / kChunk;
        const bool full_tile = (b0 + T <= a.B);
        const float *xt = a.x + b0 * D;
        auto load_chunk = [&](int c) {
            // per-thread 32-bit offset + uniform base; the asm keeps the compiler from hoisting
            // NLD loop-invariant 64-bit addresses out of the chunk loop (and spilling them)
            int vo = sq * D + flc;
            asm volatile("" : "+v"(vo));
            if (DPK_ABLATE == 4) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) pre[i] = (float)(vo + i);
            } else if (full_tile && (c + 1) * kChunk <= D) {
                const float *xc = xt + c * kChunk;
#pragma unroll
                for (int i = 0; i < NLD; ++i) pre[i] = (xc + (int64_t)i * SQN * D)[vo];
            } else {  // ragged tile / last chunk: clamp (clamped slots are never consumed)
                const int f = min(c * kChunk + flc, D - 1);
                int sqv = sq;
                asm volatile("" : "+v"(sqv));
                const int nv1 = (int)min((int64_t)T, a.B - b0) - 1;
#pragma unroll
                for (int i = 0; i < NLD; ++i) pre[i] = xt[min(SQN * i + sqv, nv1) * D + f];
            }
        };
        // this wave's entry stream (consumption order, see ratspn_struct_kernel)
        constexpr bool kLdsTables = (CB <= 2) && (SPL == 2);
        bool use_lds = kLdsTables && a.tabcap > 0;
        cint_p fl_g = (SPL == 2 ? a.fl2 : a.fl1) + (int64_t)g * a.SP;
        cfloat_p par_g = a.par + ((int64_t)g * n_cblk + kb / CB) * a.SP * 2 * CB;
        cfloat_p cel_g = a.cel + ((int64_t)g * n_cblk + kb / CB) * a.SP * CB;
        int pos = 0;

        // LDS-resident block records of this wave for the current chunk (kLdsTables): every wave
        // copies its own records, so no other wave ever reads them and only lgkmcnt orders them
        // the unit-scale two-channel kernel reads compact records (CompactPipe), everything else the full ones
        constexpr bool kCompact = (DIST == 0) && !GEN && (CB == 2) && (SPL == 2) && (DPK_NO_EXPAND == 0);
        constexpr int RECBB = kCompact ? kCompactRec : RecGeom<(kLdsTables ? CB : 1)>::RECBB;
        constexpr int NTL = (QB <= 4) ? 2 : 3;  // float4 per lane: 2 / 3 KiB of records per (wave, chunk)
        // two buffers per wave; the records of chunk c+1 are copied by LDS-DMA (no VGPRs) while chunk
        // c is consumed.  Ordering: the DMA is issued BEFORE the x prefetch of the same chunk, vmcnt
        // retires in order, and the x prefetch is waited for before the chunk is staged.
        char *tab_lds = smem + G::BUF_BYTES + 64 + wave * (2 * a.tabcap);
        const char *rec_g = reinterpret_cast<const char *>(a.rec) + (int64_t)g * (a.SP / kBlock) * RECBB;
        bool grp_unit = (DIST == 0);
        if (kLdsTables && active) {
#pragma unroll
            for (int q = 0; q < QB; ++q) grp_unit = grp_unit && (a.unit[g * QB + q] != 0);
        }
        if (DIST == 0 && !GEN && !grp_unit) use_lds = false;  // hint was wrong for this group
        // GEN = false on the LDS route: acc holds sum (x-mu)^2 - 2*(constants) until the end of the pass
        bool expand = expand_all && (use_lds || kWide);
        // the unit-scale kernel only carries the expanded pipeline: unbounded means take the scalar-cache path
        if (DIST == 0 && !GEN && SPL == 2 && !DPK_NO_EXPAND && !expand) use_lds = false;
        bool acc_is_squares = (DIST == 0) && !GEN && use_lds && !expand;
        auto load_tab = [&](int c) {
            if (!use_lds || !active) return;
            // records of chunk c plus two blocks of run-ahead for the pipeline (the stream is
            // contiguous, so these are the first blocks of the next chunk or neutral slack)
            const int64_t so = ((int64_t)g * NC + c) * QB;
            const int blk0 = a.segoff[so] / kBlock;
            int nbc = 2;
#pragma unroll
            for (int q = 0; q < QB; ++q) nbc += a.nblk[so + q];
            const int bytes = min(nbc * RECBB, a.tabcap);
            const char *src = rec_g + (int64_t)blk0 * RECBB;
            char *dst = tab_lds + (c & 1) * a.tabcap;
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
                const int o = (lane + 64 * i) * 16;
                if (o < bytes)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + o),
                        (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
            }
        };
        bool qgo = qfree && pass == 0;
        bool nan_row = false;
        load_tab(0);
        load_chunk(0);

#ifdef DPK_TIMELINE
#define DPK_STAMP(slot) do { if (a.dbg && lane == 0) a.dbg[(((int64_t)blockIdx.x * kLeafWaves + wave) * (NC + 2) + c) * 6 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DPK_STAMP(slot) do { } while (0)
#endif
        for (int c = 0; c < NC; ++c) {
            DPK_STAMP(0);
            lds_barrier();  // every wave is done with the previous chunk
            DPK_STAMP(1);
            unsigned chk = 0u;
            if (XLDS && nan_row) {   // the previous chunk ran in the exact LDS form: the neutral row holds zeros again
                if (tid < ROW) xs_lds[kChunk * ROW + tid] = 0.f;
                nan_row = false;
            }
            float *wr = xs_lds + flc * ROW + (SPL == 2 ? 2 * sq : sq);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const float v = pre[i];
                chk = max(chk, __float_as_uint(v * v));  // v*v >= 0: its bit pattern is monotone, NaN on top
                constexpr int H = 64 / SQN;  // passes covering the first 64 samples of the tile
                const int pos_i = (SPL == 2) ? ((i < H) ? 2 * SQN * i : 2 * SQN * (i - H) + 1) : SQN * i;
                wr[pos_i] = v;
            }
            // work-group OR of "non-finite value staged" through 8 LDS words (no __syncthreads_or:
            // its release fence is a vmcnt(0) that would drain the prefetch just issued)
            const bool wave_bad = __any(chk > thr_bits);
            if (lane == 0) flags_lds[wave] = wave_bad ? 1 : 0;
            DPK_STAMP(2);
            lds_barrier();
            DPK_STAMP(3);
            int slow = 0;
#pragma unroll
            for (int w = 0; w < kLeafWaves; ++w) slow |= flags_lds[w];
            slow = __builtin_amdgcn_readfirstlane(slow);
            // prefetch of the next chunk: issued AFTER the barrier so that no wave holds the others
            // up with the address arithmetic and the scalar loads behind the record copy
            if (c + 1 < NC) {
                load_tab(c + 1);
                load_chunk(c + 1);
            }

            // compact-record kernel: a slow chunk stays on the LDS pipeline in the exact per-entry form (complete
            // terms, neutral row = NaN), and the chunks after it carry on in the expanded form
            bool slow_lds = false;
            if constexpr (kCompact && XLDS) {
                slow_lds = slow && expand && use_lds;
                if (slow_lds) {
                    if (tid < ROW) xs_lds[kChunk * ROW + tid] = __int_as_float(0x7fc00000);
                    if (tid == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;   // host hint (launch_leaf)
                    nan_row = true;
                    lds_barrier();
                }
            }
            if (slow && !slow_lds) qgo = false;
            if (qgo && !slow) {
                // this wave's share of sum_f x_f^2 for the chunk: rows 8*wave .. 8*wave+7 (see qfree)
                const char *qb = lane_base + wave * (kChunk / kLeafWaves) * G::ROWB;
                const int nvalid = min(kChunk, D - c * kChunk) - wave * (kChunk / kLeafWaves);
#pragma unroll
                for (int j = 0; j < kChunk / kLeafWaves; ++j) {
                    if (j < nvalid) {
                        if constexpr (SPL == 2) {
                            const float2 v = *reinterpret_cast<const float2 *>(qb + j * G::ROWB);
                            const f32x2 xv = {v.x, v.y};
                            qpart = __builtin_elementwise_fma(xv, xv, qpart);
                        } else {
                            const float v = *reinterpret_cast<const float *>(qb + j * G::ROWB);
                            qpart[0] = fmaf(v, v, qpart[0]);
                        }
                    }
                }
            }

            if (active) {
                cint_p nbp = a.nblk + ((int64_t)g * NC + c) * QB;
                if (slow && acc_is_squares) {
                    // first non-finite chunk of a unit-scale tile: bring acc to final units (the exact
                    // terms can be as large as -FLT_MAX and must not be rescaled) and finish this tile on
                    // the scalar-cache paths
#pragma unroll
                    for (int q = 0; q < QB; ++q)
#pragma unroll
                        for (int k = 0; k < CB; ++k)
#pragma unroll
                            for (int s = 0; s < SPL; ++s) acc[q][k][s] *= -0.5f;
                    acc_is_squares = false;
                    use_lds = false;
                }
                if (slow && expand && !slow_lds) {  // acc is already in final units: just leave the LDS pipeline
                    expand = false;
                    use_lds = false;
                    saw_slow = true;
                }
                if ((slow && !slow_lds) || !use_lds) {
                    // exact per-element path (NaN / inf evidence) or wide channel blocks: tables
                    // through the scalar cache
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const int r = g * QB + q;
                        const int nb = nbp[q];
                        cfloat_p bp = a.biasc + ((int64_t)r * NC + c) * I + kb;
                        if (slow) {
                            leaf_accum_plain<DIST, CB, SPL, 1>(acc[q], lane_base, fl_g + pos,
                                                               par_g + (int64_t)pos * 2 * CB,
                                                               cel_g + (int64_t)pos * CB, nb);
                        } else if (kWide && expand) {
                            float qr[SPL];
#pragma unroll
                            for (int s = 0; s < SPL; ++s) qr[s] = 0.f;
                            leaf_accum_linear<CB, SPL, (DEPTH == 0)>(acc[q], qr, lane_base, fl_g + pos,
                                                                     par_g + (int64_t)pos * 2 * CB, nb);
                            cfloat_p bx = a.biasx + ((int64_t)r * NC + c) * I + kb;
#pragma unroll
                            for (int k = 0; k < CB; ++k) {
                                const float bk = bx[k];
#pragma unroll
                                for (int s = 0; s < SPL; ++s) acc[q][k][s] += fmaf(-0.5f, qr[s], bk);
                            }
                        } else {
                            leaf_accum_plain<DIST, CB, SPL, 0>(acc[q], lane_base, fl_g + pos,
                                                               par_g + (int64_t)pos * 2 * CB,
                                                               cel_g + (int64_t)pos * CB, nb);
#pragma unroll
                            for (int k = 0; k < CB; ++k) {
                                const float bk = bp[k];
#pragma unroll
                                for (int s = 0; s < SPL; ++s) acc[q][k][s] += bk;
                            }
                        }
                        pos += nb * kBlock;
                    }
                } else {
                    if constexpr (kLdsTables) {
                        auto chunk_lds = [&](auto &pipe, auto mode_tag) {
                            constexpr int MODE = decltype(mode_tag)::value;
                            pipe.prime(tab_lds + (c & 1) * a.tabcap, lane_base);
#pragma unroll
                            for (int q = 0; q < QB; ++q) {
                                const int r = g * QB + q;
                                const int nb = nbp[q];
                                cfloat_p bp = a.biasc + ((int64_t)r * NC + c) * I + kb;
                                // unit-scale groups keep acc = sum (x-mu)^2 - 2*(constants), finished with
                                // one multiply by -1/2 after the last chunk (kAccIsSquares)
                                pipe.template run<DIST, MODE>(acc[q], lane_base, nb);
#pragma unroll
                                for (int k = 0; k < CB; ++k) {
                                    const float bk = (MODE == 2) ? -2.0f * bp[k] : bp[k];
#pragma unroll
                                    for (int s = 0; s < SPL; ++s) acc[q][k][s] += bk;
                                }
                                pos += nb * kBlock;
                            }
                        };
                        LdsPipe<CB, SPL, GEN> pipe;
                        if constexpr (DIST == 0 && !GEN && SPL == 2) {
                            if (slow_lds) {
                                if constexpr (kCompact && XLDS) {
                                    CompactPipe cp;
                                    const int lane_off = lane * (4 * SPL);
                                    cp.prime(tab_lds + (c & 1) * a.tabcap, smem, lane_off);
#pragma unroll
                                    for (int q = 0; q < QB; ++q) {
                                        const int nb = nbp[q];
                                        f32x2 A[2] = {{acc[q][0][0], acc[q][0][1]}, {acc[q][1][0], acc[q][1][1]}};
                                        cp.run_exact(A, smem, lane_off, nb);
                                        acc[q][0][0] = A[0][0]; acc[q][0][1] = A[0][1];
                                        acc[q][1][0] = A[1][0]; acc[q][1][1] = A[1][1];
                                        pos += nb * kBlock;
                                    }
                                }
                            } else if (expand) {
                                CompactPipe cp;
                                const int lane_off = lane * (4 * SPL);
                                if constexpr (kCompact) cp.prime(tab_lds + (c & 1) * a.tabcap, smem, lane_off);
                                else pipe.prime(tab_lds + (c & 1) * a.tabcap, lane_base);
#pragma unroll
                                for (int q = 0; q < QB; ++q) {
                                    const int r = g * QB + q;
                                    const int nb = nbp[q];
                                    cfloat_p bp = a.biasx + ((int64_t)r * NC + c) * I + kb;
                                    f32x2 P[CB], Q = {0.f, 0.f};
#pragma unroll
                                    for (int k = 0; k < CB; ++k) P[k] = (f32x2){acc[q][k][0], acc[q][k][1]};
                                    // fused model: the x^2 sums are factored out (qfree)
                                    if constexpr (kCompact) {
                                        f32x2 (&P2)[2] = reinterpret_cast<f32x2 (&)[2]>(P);
                                        cp.template run<(DEPTH == 0)>(P2, Q, smem, lane_off, nb);
                                    } else {
                                        pipe.template run3<(DEPTH == 0)>(P, Q, lane_base, nb);
                                    }
                                    // log-density sum = P - Q/2 + (constants - sum mu^2 / 2); Q == 0 when factored out
#pragma unroll
                                    for (int k = 0; k < CB; ++k) {
                                        const float bk = bp[k];
                                        acc[q][k][0] = fmaf(-0.5f, Q[0], P[k][0]) + bk;
                                        acc[q][k][1] = fmaf(-0.5f, Q[1], P[k][1]) + bk;
                                    }
                                    pos += nb * kBlock;
                                }
                            } else if (DPK_NO_EXPAND) {
                                chunk_lds(pipe, std::integral_constant<int, 2>{});
                            }
                        } else if (GEN || DIST != 0) {
                            chunk_lds(pipe, std::integral_constant<int, 0>{});
                        } else {
                            chunk_lds(pipe, std::integral_constant<int, 2>{});
                        }
                        DPK_STAMP(4);
#ifdef DPK_TIMELINE
                        if (a.dbg && lane == 0) a.dbg[(((int64_t)blockIdx.x * kLeafWaves + wave) * (NC + 2) + c) * 6 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
                    }
                }
            }
        }

        if (acc_is_squares) {
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int k = 0; k < CB; ++k)
#pragma unroll
                    for (int s = 0; s < SPL; ++s) acc[q][k][s] *= -0.5f;
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
        float qterm = 0.f;
        if (qfree) {
            qlds[wave * T + lane] = qpart[0];
            if (SPL == 2) qlds[wave * T + lane + 64] = qpart[1];
            __syncthreads();
            if (tid < T) {
                float qs = 0.f;
#pragma unroll
                for (int w = 0; w < kLeafWaves; ++w) qs += qlds[w * T + tid];
                qterm = -0.5f * qs;
            }
        }
        double part = 0.0;
        if (tid < T) {
            const int64_t b = b0 + tid;
            if (b < a.B) {
                for (int cl = 0; cl < a.C; ++cl) {
                    const float mm = run_m[cl * T + tid];
                    const float ll = (mm > -INFINITY) ? mm + __logf(run_s[cl * T + tid]) + qterm : -INFINITY;
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
                atomicAdd(a.ll_sum + a.ll_cnt, (double)(nvalid * a.C));
            }
        }
    }
    // host hint (launch_leaf): raised at the very end, where no register is under pressure -- a store inside the
    // chunk loop's slow branch changes the SGPR allocation of the whole loop (+1.5 % on clean inputs)
    if (saw_slow && lane == 0 && a.slow_flag != nullptr) *a.slow_flag = a.launch_seq;
#ifdef DPK_TIMELINE
    if (a.dbg && lane == 0) {
        unsigned long long *row = a.dbg + (((int64_t)blockIdx.x * kLeafWaves + wave) * (NC + 2) + NC) * 6;
        row[1] = __builtin_amdgcn_s_memtime();
        row[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
int prepare_leaf_structure(const RatWs &w, const int64_t *mask, const uint8_t *pad, int R, int d,
                           uint32_t flags, hipStream_t st) {
    if (flags & DPK_FLAG_STRUCT_CACHED) return DPK_OK;
    const size_t lds = (size_t)(2 * w.QB * d + w.QB * (w.NC + 1) + w.NC * w.QB + 1) * sizeof(int);
    DPK_REQUIRE(lds <= 64 * 1024, DPK_EUNSUPPORTED, "region dimension %d too large for the structure kernel", d);
    DPK_LAUNCH(ratspn_struct_kernel, dim3(w.G), dim3(256), lds, st, mask, pad, R, d, w.NC, w.QB, w.SP,
                       w.fl1, w.fl2, w.srcr, w.feat, w.nblk, w.segoff);
    DPK_CHECK_LAUNCH("ratspn_struct_kernel");
    return DPK_OK;
}

// channel block the kernels use for `I` channels: the largest of {8,4,2,1} dividing I
static int channel_block(int I) { return (I % 8 == 0) ? 8 : (I % 4 == 0) ? 4 : (I % 2 == 0) ? 2 : 1; }

struct SoftmaxJob {
    const float *w;
    float *W, *LW;
    int rows, n;
};

static int prepare_leaf_tables(int dist, const RatWs &w, const int64_t *mask, const uint8_t *pad,
                               const float *p0, const float *p1, int R, int I, int CB, int d, uint32_t flags,
                               hipStream_t st, const SoftmaxJob *jobs = nullptr, int n_jobs = 0) {
    int rc = prepare_leaf_structure(w, mask, pad, R, d, flags, st);
    if (rc) return rc;
    PrepArgs a{};
    a.p0 = p0; a.p1 = p1; a.srcr = w.srcr; a.nblk = w.nblk; a.segoff = w.segoff; a.fl2 = w.fl2;
    a.R = R; a.I = I; a.CB = CB; a.d = d; a.NC = w.NC; a.QB = w.QB; a.SP = w.SP; a.G = w.G;
    a.par = w.par; a.cel = w.cel; a.biasc = w.biasc; a.biasx = w.biasx; a.unit = w.unit;
    a.rec = (CB == I && CB <= 2) ? w.rec : nullptr;
    // the unit-scale two-channel kernel (launch_leaf: hint && Gaussian && CB == 2) reads compact records
    a.rec_compact = (dist == 0 && I == 2 && CB == 2 && (flags & DPK_FLAG_UNIT_SCALE) != 0 && DPK_NO_EXPAND == 0) ? 1 : 0;
    int rows = 0;
    for (int m = 0; m < 3; ++m) {
        if (m < n_jobs) {
            a.w[m] = jobs[m].w; a.W[m] = jobs[m].W; a.LW[m] = jobs[m].LW;
            a.rows[m] = jobs[m].rows; a.n[m] = jobs[m].n;
            rows += jobs[m].rows;
        }
    }
    const int grid = 2 * w.G * kPrepSlices + w.G + cdiv(rows, 4);   // block roles: see ratspn_prep_kernel
    if (dist == 0) DPK_LAUNCH(ratspn_prep_kernel<0>, dim3(grid), dim3(256), 0, st, a);
    else DPK_LAUNCH(ratspn_prep_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    DPK_CHECK_LAUNCH("ratspn_prep_kernel");
    return DPK_OK;
}

static void fill_leaf_args(LeafArgs &a, const RatWs &w) {
    a.NC = w.NC; a.SP = w.SP;
    a.fl1 = as_const(w.fl1); a.fl2 = as_const(w.fl2); a.nblk = as_const(w.nblk);
    a.segoff = as_const(w.segoff); a.rec = w.rec; a.tabcap = w.tabcap; a.tabcap_c = w.tabcap_c;
    a.par = as_const(w.par); a.cel = as_const(w.cel); a.biasc = as_const(w.biasc); a.biasx = as_const(w.biasx);
    a.unit = as_const(w.unit);
}

template <int DIST, int QB, int CB, int SPL, int DEPTH, int S, bool GEN, bool XLDS = false>
static int launch_leaf_gen(const LeafArgs &a, hipStream_t st) {
    using G = TileGeom<SPL>;
    const int grid = cdiv(a.B, G::T);
    size_t lds = G::BUF_BYTES + 64 + (size_t)kLeafWaves * 2 * a.tabcap;
    if (DEPTH > 0) lds += (size_t)(2 * a.C + kLeafWaves) * G::T * sizeof(float);
    auto kern = ratspn_leaf_kernel<DIST, QB, CB, SPL, DEPTH, S, GEN, XLDS>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e));
            return DPK_ELAUNCH;
        }
    }
#ifdef DPK_TIMELINE
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, (size_t)4096 * kLeafWaves * 64 * 6 * 8);
        const_cast<LeafArgs &>(a).dbg = dbg;
        FILE *f = fopen("/tmp/dpk_timeline_ptr.txt", "w");
        if (f) { fprintf(f, "%p %d %d\n", (void *)dbg, grid, a.NC); fclose(f); }
    }
#endif
    hipEvent_t ev0, ev1;
    profile_take(&ev0, &ev1, DEPTH > 0 ? DPK_KERNEL_RATSPN_FUSED : DPK_KERNEL_RATSPN_LEAF);
    if (ev0) (void)hipEventRecord(ev0, st);
    int gy = 1;
    if (DEPTH == 0) {
        const int n_items = (a.R / QB) * (a.I / CB);
        gy = cdiv(n_items, kLeafWaves);
    }
    DPK_LAUNCH(kern, dim3(grid, gy), dim3(kLeafWaves * 64), lds, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    DPK_CHECK_LAUNCH("ratspn_leaf_kernel");
    return DPK_OK;
}

template <int DIST, int QB, int CB, int SPL, int DEPTH, int S>
static int launch_leaf(const LeafArgs &a, hipStream_t st) {
    // the means-only variant exists where the LDS-table pipeline does (Gaussian, CB <= 2, SPL == 2)
    if constexpr (DIST == 0 && CB <= 2 && SPL == 2) {
        if (a.unit_hint) {
            if (CB == 2 && DPK_NO_EXPAND == 0) {   // compact records (prepare_leaf_tables wrote them for this case)
                LeafArgs c = a;
                c.tabcap = a.tabcap_c;
                // Two builds of this kernel, both exact for any input: the default sends a tile that meets NaN / inf /
                // out-of-bound evidence to the scalar-cache path; the other keeps such chunks on the LDS record
                // pipeline in the exact per-entry form (1.2-1.6x faster on marginalised inputs, 2.6 % slower on clean
                // ones because of the extra code in the chunk loop).  A work-group that meets a slow chunk stores the
                // launch number in a host-mapped word; a launch takes the second build while one of the recent
                // launches did so.  The word is read without synchronising: a stale value only costs speed.
                const bool marginal = slow_hint_next(a.rec, &c.slow_flag, &c.launch_seq);
                if (marginal) return launch_leaf_gen<DIST, QB, CB, SPL, DEPTH, S, false, true>(c, st);
                return launch_leaf_gen<DIST, QB, CB, SPL, DEPTH, S, false>(c, st);
            }
            return launch_leaf_gen<DIST, QB, CB, SPL, DEPTH, S, false>(a, st);
        }
    }
    return launch_leaf_gen<DIST, QB, CB, SPL, DEPTH, S, true>(a, st);
}

template <int DIST>
static int leaf_forward_dispatch(const LeafArgs &a, hipStream_t st) {
    // channel block = largest of {8,4,2,1} dividing I; two samples per lane while registers allow
    const int I = a.I;
    const bool q4 = leaf_group(a.R) == 4;
#ifdef DPK_HEADLINE_ONLY
    return launch_leaf<DIST, 4, 2, 2, 0, 1>(a, st);
#endif
    if (I % 8 == 0) {
        // two samples per lane once there are more than two 64-sample tiles per CU (see fused_launch)
        if (a.B > 2 * 256 * 64) return q4 ? launch_leaf<DIST, 4, 8, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 8, 2, 0, 1>(a, st);
        return q4 ? launch_leaf<DIST, 4, 8, 1, 0, 1>(a, st) : launch_leaf<DIST, 2, 8, 1, 0, 1>(a, st);
    }
    if (I % 4 == 0) return q4 ? launch_leaf<DIST, 4, 4, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 4, 2, 0, 1>(a, st);
    if (I % 2 == 0) return q4 ? launch_leaf<DIST, 4, 2, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 2, 2, 0, 1>(a, st);
    return q4 ? launch_leaf<DIST, 4, 1, 2, 0, 1>(a, st) : launch_leaf<DIST, 2, 1, 2, 0, 1>(a, st);
}

// ratspn_leaf_gemm.hip: the leaf layer alone on the matrix cores
int ratspn_leaf_gemm_forward(void *ws, const float *x, int64_t B, int D, const int64_t *mask, const uint8_t *pad,
                             const float *loc, const float *scale, int R, int I, int d, float *out, uint32_t flags,
                             hipStream_t st);
static int &mfma_route_ref() {
    static int enabled = [] {
        const char *e = getenv("DPK_RATSPN_GEMM");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return enabled;
}
static bool mfma_enabled() { return mfma_route_ref() != 0; }
// measurement knob (bench.py's fp32_exact figure, A/B runs): the matrix-core route of the RAT-SPN forward on / off
extern "C" int32_t dpk_ratspn_mfma_route(int32_t enable) {
    int &v = mfma_route_ref();
    const int prev = v;
    if (enable >= 0) v = enable ? 1 : 0;
    return prev;
}
static bool leaf_gemm_route(int dist, const float *x, const float *out, int D, int R, int I, int d, uint32_t flags) {
    return mfma_enabled() && dist == 0 && (flags & DPK_FLAG_UNIT_SCALE) != 0 && leaf_gemm_shape_ok(D, R, I, d) &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

static int leaf_forward_common(int dist, const float *x, int64_t B, int32_t D, const int64_t *mask,
                               const uint8_t *pad_mask, const float *p0, const float *p1, int32_t R,
                               int32_t I, int32_t d, float *out, void *ws, int64_t ws_bytes,
                               uint32_t flags, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && R > 0 && (R % 2) == 0 && I > 0 && d > 0, DPK_EINVAL, "leaf_forward: bad sizes");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && mask && p0 && out && ws, DPK_EINVAL, "leaf_forward: null pointer");
    DPK_REQUIRE(dist == 1 || p1, DPK_EINVAL, "leaf_forward: null scale");
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, leaf_group(R), 0, 0, 0, 0);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "leaf_forward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (leaf_gemm_route(dist, x, out, D, R, I, d, flags) && w.lg != nullptr) {
        // (the structure tables of the VALU route stay current: a later call with the cached structure may take it)
        int rc = prepare_leaf_structure(w, mask, pad_mask, R, d, flags, st);
        if (rc) return rc;
        return ratspn_leaf_gemm_forward(w.lg, x, B, D, mask, pad_mask, p0, p1, R, I, d, out, flags, st);
    }
    int rc = prepare_leaf_tables(dist, w, mask, pad_mask, p0, p1, R, I, channel_block(I), d, flags, st);
    if (rc) return rc;
    LeafArgs a{};
    a.x = x; a.B = B; a.D = D; a.R = R; a.I = I; a.d = d;
    fill_leaf_args(a, w);
    a.unit_hint = (flags & DPK_FLAG_UNIT_SCALE) != 0;
    a.leaf_out = out;
    return dist == 0 ? leaf_forward_dispatch<0>(a, st) : leaf_forward_dispatch<1>(a, st);
}

template <int DEPTH, int I, int S>
static int fused_launch(const LeafArgs &a, hipStream_t st) {
#ifdef DPK_FORCE_SPL1
    return launch_leaf<0, (1 << DEPTH), I, 1, DEPTH, S>(a, st);
#else
    // Two samples per lane halve the per-sample cost of fetching the table entries (LDS records for <= 4 channels,
    // scalar-cache loads for 8).  With 8 channels the 64 accumulators per wave fit the register budget up to depth 2,
    // and the 128-sample tiles only pay off once there are more than two 64-sample tiles per CU to begin with.
    if constexpr (I <= 4) {
        return launch_leaf<0, (1 << DEPTH), I, 2, DEPTH, S>(a, st);
    } else {
        if (DEPTH <= 2 && a.B > 2 * 256 * 64) return launch_leaf<0, (1 << (DEPTH <= 2 ? DEPTH : 2)), I, 2, (DEPTH <= 2 ? DEPTH : 2), S>(a, st);
        return launch_leaf<0, (1 << DEPTH), I, 1, DEPTH, S>(a, st);
    }
#endif
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

// ratspn_gemm.hip: fused forward with the leaf layer on the matrix cores
int ratspn_gemm_forward(const RatWs &w, const float *x, int64_t B, int D, const int64_t *mask, const uint8_t *pad,
                        const float *loc, const float *scale, const float *sum_weight0, const float *root_weight,
                        int reps, int I, int S, int C, float *out, double *ll_sum, uint32_t flags, hipStream_t st,
                        const GemmEmit *emit = nullptr);

bool gemm_wide_shape_ok(int D, int reps, int I, int S, int C);   // ratspn_gemm_wide.hip

// The MFMA route takes a fused evaluation when the model is in its envelope (depth 2, 2 or 4 channels / sums),
// the caller hints unit scales (checked on the device), the rows of x are 16-byte aligned (LDS-DMA) and the leaf
// outputs are not wanted.  DPK_RATSPN_GEMM=0 in the environment keeps the VALU kernels (A/B measurements).
static bool gemm_route(const float *x, int D, int depth, int reps, int I, int S, int C, bool want_leaf, uint32_t flags) {
    return mfma_enabled() && !want_leaf && (flags & DPK_FLAG_UNIT_SCALE) != 0 && gemm_shape_ok(D, depth, reps, I, S) &&
           C <= 64 && (I != 8 || gemm_wide_shape_ok(D, reps, I, S, C)) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

}  // namespace dpk

using namespace dpk;

extern "C" int dpk_ratspn_forward_on_mfma(const float *x, int32_t D, int32_t depth, int32_t reps, int32_t I,
                                          int32_t S, int32_t C, int32_t want_leaf_out, uint32_t flags) {
    return gemm_route(x, D, depth, reps, I, S, C, want_leaf_out != 0, flags) ? 1 : 0;
}

extern "C" int dpk_gaussian_leaf_forward_on_mfma(const float *x, const float *out, int32_t D, int32_t R, int32_t I,
                                                 int32_t d, uint32_t flags) {
    return leaf_gemm_route(0, x, out, D, R, I, d, flags) ? 1 : 0;
}

extern "C" int64_t dpk_ratspn_workspace_bytes(int32_t in_features, int32_t regions, int32_t dimension,
                                              int32_t channels, int32_t depth, int32_t reps,
                                              int32_t sums, int32_t classes) {
    if (in_features <= 0 || regions <= 0 || (regions % 2) != 0 || dimension <= 0 || channels <= 0)
        return DPK_EINVAL;
    // the per-layer operators group regions by leaf_group(R), the fused model by repetition: size for both
    const int64_t a = carve_ratspn_ws(nullptr, in_features, regions, dimension, channels, leaf_group(regions), 0,
                                      0, 0, 0).bytes;
    int64_t b = 0;
    if (depth >= 1 && depth <= 3 && reps >= 1 && regions == reps * (1 << depth))
        b = carve_ratspn_ws(nullptr, in_features, regions, dimension, channels, 1 << depth, depth, reps,
                            sums > 0 ? sums : 1, classes > 0 ? classes : 1).bytes;
    return a > b ? a : b;
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

extern "C" int dpk_ratspn_forward_train(const float *x, int64_t B, int32_t D, const int64_t *mask, const uint8_t *pad_mask,
                                        const float *loc, const float *scale, const float *sum_weight0,
                                        const float *root_weight, int32_t depth, int32_t reps, int32_t I, int32_t S,
                                        int32_t C, float *out, float *leaf_rel, float *sum_rel, float *out_rel, void *ws,
                                        int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && reps > 0 && I > 0 && S > 0 && C > 0, DPK_EINVAL, "ratspn_forward_train: bad sizes");
    DPK_REQUIRE(mask && loc && scale && sum_weight0 && root_weight && ws, DPK_EINVAL, "ratspn_forward_train: null pointer");
    DPK_REQUIRE(B == 0 || (x && out && leaf_rel && sum_rel && out_rel), DPK_EINVAL, "ratspn_forward_train: null pointer");
    if (depth != 2 || !gemm_route(x, D, depth, reps, I, S, C, false, flags)) {
        set_error("ratspn_forward_train: outside the fused MFMA route (depth 2, unit scales, 16-byte aligned rows)");
        return DPK_EUNSUPPORTED;
    }
    const int Q = 4, R = reps * Q;
    const int pad = (Q - D % Q) % Q, d = (D + pad) / Q;
    DPK_REQUIRE(pad == 0 || pad_mask, DPK_EINVAL, "ratspn_forward_train: padded model needs pad_mask");
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, Q, depth, reps, S, C);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "ratspn_forward_train: workspace %lld < %lld", (long long)ws_bytes,
                (long long)w.bytes);
    if (w.g_nt <= 0) {
        set_error("ratspn_forward_train: shape outside the MFMA tables");
        return DPK_EUNSUPPORTED;
    }
    if (B == 0) return DPK_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = prepare_leaf_structure(w, mask, pad_mask, R, d, flags, st);
    if (rc) return rc;
    const GemmEmit emit{leaf_rel, sum_rel, out_rel};
    return ratspn_gemm_forward(w, x, B, D, mask, pad_mask, loc, scale, sum_weight0, root_weight, reps, I, S, C, out, nullptr,
                               flags, st, &emit);
}

extern "C" int dpk_ratspn_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                                  const uint8_t *pad_mask, const float *loc, const float *scale,
                                  const float *sum_weight0, const float *sum_weight1,
                                  const float *root_weight, int32_t depth, int32_t reps, int32_t I,
                                  int32_t S, int32_t C, float *out, float *leaf_out, double *ll_sum,
                                  void *ws, int64_t ws_bytes, uint32_t flags, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && reps > 0 && I > 0 && S > 0 && C > 0, DPK_EINVAL,
                "ratspn_forward: bad sizes");
    DPK_REQUIRE(B == 0 || (x && out), DPK_EINVAL, "ratspn_forward: null pointer");
    DPK_REQUIRE(mask && loc && scale && root_weight && ws, DPK_EINVAL, "ratspn_forward: null pointer");
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
    RatWs w = carve_ratspn_ws(ws, D, R, d, I, Q, depth, reps, S, C);
    DPK_REQUIRE(ws_bytes >= w.bytes, DPK_EWORKSPACE, "ratspn_forward: workspace %lld < %lld",
                (long long)ws_bytes, (long long)w.bytes);
    if (!(I == 2 || I == 4 || I == 8) || (depth >= 2 && !(S == 2 || S == 4 || S == 8))) {
        set_error("ratspn_forward: (channels=%d, sums=%d) not built; use the per-layer entry points", I, S);
        return DPK_EUNSUPPORTED;
    }
    if (B == 0) return DPK_OK;
    hipStream_t st = (hipStream_t)stream;
    if (gemm_route(x, D, depth, reps, I, S, C, leaf_out != nullptr, flags) && w.g_nt > 0) {
        // keep the structure tables of the VALU route current too: a later call with the same (cached) structure
        // may take that route (leaf outputs wanted, unaligned input)
        int rc = prepare_leaf_structure(w, mask, pad_mask, R, d, flags, st);
        if (rc) return rc;
        return ratspn_gemm_forward(w, x, B, D, mask, pad_mask, loc, scale, sum_weight0, root_weight, reps, I, S, C,
                                   out, ll_sum, flags, st);
    }
    const int nlast = depth >= 2 ? S : I;
    SoftmaxJob jobs[3];
    int n_jobs = 0;
    if (depth >= 2) jobs[n_jobs++] = SoftmaxJob{sum_weight0, w.w[0], w.lw[0], reps * (Q / 2) * S, I * I};
    if (depth >= 3) jobs[n_jobs++] = SoftmaxJob{sum_weight1, w.w[1], w.lw[1], reps * (Q / 4) * S, S * S};
    jobs[n_jobs++] = SoftmaxJob{root_weight, w.w[2], w.lw[2], C, reps * nlast * nlast};
    int rc = prepare_leaf_tables(0, w, mask, pad_mask, loc, scale, R, I, I, d, flags, st, jobs, n_jobs);
    if (rc) return rc;

    LeafArgs a{};
    a.x = x; a.B = B; a.D = D; a.R = R; a.I = I; a.d = d;
    fill_leaf_args(a, w);
    a.unit_hint = (flags & DPK_FLAG_UNIT_SCALE) != 0;
    a.leaf_out = leaf_out;
    a.reps = reps; a.C = C;
    a.W0 = as_const(w.w[0]); a.LW0 = as_const(w.lw[0]); a.W1 = as_const(w.w[1]);
    a.LW1 = as_const(w.lw[1]); a.Wr = as_const(w.w[2]); a.LWr = as_const(w.lw[2]);
    a.out = out; a.ll_sum = ll_sum; a.ll_cnt = (flags & DPK_FLAG_LL_SUM_SPREAD) ? 16 : 1;
#ifdef DPK_HEADLINE_ONLY  // measurement builds: only the headline instantiation
    if (depth == 2 && I == 2 && S == 2) return fused_launch<2, 2, 2>(a, st);
    set_error("measurement build: only depth=2, channels=2, sums=2");
    return DPK_EUNSUPPORTED;
#else
    switch (depth) {
        case 1: return fused_dispatch_i<1>(a, S, st);
        case 2: return fused_dispatch_i<2>(a, S, st);
        default: return fused_dispatch_i<3>(a, S, st);
    }
#endif
}
