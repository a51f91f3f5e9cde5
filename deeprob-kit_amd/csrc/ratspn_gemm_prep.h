// Parameter tables of the fused RAT-SPN kernels with the leaf layer on the matrix cores (ratspn_gemm*.hip), and the
// check that keeps them honest.
//
// reference: what the tables cache is RegionGraphLayer's gather + GaussianLayer's parameters
// (deeprob/spn/layers/ratspn.py:42-66, :87-108, :160-213) and the log_softmax of the sum / root weights (:375, :455).
//
// The tables (f16 MFMA fragments of the means, per-chunk / per-K-step constants, softmax rows) are a pure function of
// mask / pad_mask / loc / scale / sum weights / root weights.  The host keys them on (address, version counter), but a
// write through `param.data` moves neither, so by default every call CHECKS them on the device
// (DPK_FLAG_PARAMS_VERIFY).  Three ways to run the work of one table work-group (`gemm_prep_block`):
//   kPrepBuild   unconditional rebuild (the host knows the parameters changed): its own launch, before the model kernel;
//   kPrepVerify  its own launch: fingerprint the bytes the block's outputs depend on, rebuild if they differ from the
//                fingerprint stored at the last build (block local; round 3) -- the large-batch ring kernels;
//   kPrepInline  round 4: the table work-groups are the LEADING work-groups of the model kernel's own launch.  They
//                fingerprint and publish ONE verdict for the launch (VerifyCtl); the model work-groups run
//                speculatively on the cached tables and read the verdict before they produce anything.  Clean (every
//                call but the first after a `.data` write): the tables were right, nothing else happens -- the check
//                costs no launch and hides under the x stream.  Dirty: every model work-group evaluates its samples on
//                the table-free exact route (per-element leaves, log-sum-exp straight from the raw weights), while the
//                table work-groups rebuild IN PLACE (nobody consumes a table value in a dirty launch) and store the new
//                fingerprints: the next launch is clean again.  No host round trip, replays from a HIP graph heal
//                themselves.
// Protocol of a launch (np table work-groups, nm model work-groups; all counters zero between launches):
//   table wg:  h = fingerprint; arrive: word += 1 + (h != stored) << 32;  wait until low32(word) == np;  dirty = hi32 != 0;
//              ticket = readers++;  [dirty: rebuild, store h];  last ticket (np + nm - 1) zeroes word and readers
//   model wg:  ... main loop ...; wait until low32(word) == np; dirty = hi32 != 0; ticket = readers++; ...; the same reset
// A work-group only ever waits for table work-groups, which have lower block indices (dispatched first) and never wait
// for a model work-group: no circular wait, whatever the residency.  One module evaluates on one stream at a time (as
// its workspace already requires); a wait that does not end within ~1 s is taken as "dirty" (exact route), never a hang.
#pragma once
#include "common.h"
#include "ratspn_gemm_common.h"
#include <math.h>

namespace dpk {

struct VerifyCtl {
    unsigned long long word;   // low 32: table work-groups that have published; high 32: those that found changed bytes
    unsigned readers;          // work-groups that have read the verdict
    unsigned pad;
};

enum { kPrepBuild = 0, kPrepVerify = 1, kPrepInline = 2 };

struct GemmPrepArgs {
    const int64_t *mask;
    const uint8_t *pad;
    const float *loc, *scale;
    int D, d, reps, NT, NKSP, KS;
    uint16_t *mtab, *ctab;
    float *bias, *bias_row, *bias_ks, *bias_sl;
    int *elig;
    const float *w[3];
    float *W[3], *LW[3];
    int rows[3], n[3];
    uint16_t *upfrag;           // 8-channel models: MFMA A-fragments of the first sum layer (wide_upfrag_*), else null
    int up_S;                   // ... its sum nodes per region
    int mode;                   // kPrepBuild / kPrepVerify / kPrepInline
    unsigned long long *hash;   // [NT*RPT] per repetition, then one per softmax-row work-group (kPrepInline)
    VerifyCtl *ctl;
    int np;                     // table work-groups of the launch
    int readers;                // work-groups that read the verdict (np + model work-groups)
    int ablate;                 // measurement only (DPK_PREP_ABLATE): 1 stop after the verdict, 2 no fragments, 4 no constants
    uint16_t *stab;             // two-channel models: the compact mean table of the slice mapping (slice_tab_*), else null
    unsigned char *smask;       // ... and the keep-masks of its lanes
    float *wx_part;             // 8-channel models: exchange of two work-groups that share a block (ratspn_gemm_wide.hip), else null
    unsigned *wx_tick;          // ... the blocks' tickets
};

// table work-groups a launch needs: one per repetition slot + one per (threads / 64) softmax rows
__host__ __device__ inline int gemm_prep_blocks(int NT, int I, int rows_total, int threads) {
    return NT * (8 / I) + cdiv(rows_total, threads / 64);
}

// ---- the verdict word ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void vi_arrive(VerifyCtl *c, bool mismatch) {
    __hip_atomic_fetch_add(&c->word, 1ull + (mismatch ? (1ull << 32) : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread: wait for the np arrivals; returns the verdict (true = some table is stale) and takes a reader ticket
__device__ __forceinline__ bool vi_wait(VerifyCtl *c, int np, unsigned &ticket) {
    unsigned long long v = __hip_atomic_load(&c->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool timed_out = false;
    if ((unsigned)v < (unsigned)np) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        do {
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(&c->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull;
        } while ((unsigned)v < (unsigned)np && !timed_out);
    }
    ticket = __hip_atomic_fetch_add(&c->readers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (v >> 32) != 0ull || timed_out;
}
// The model work-groups' side, latency off the critical path: `vi_peek` is an ordinary (device-coherent) load that every
// thread may issue early, straight-line, under its K loop; `vi_verdict` uses it if all table work-groups had arrived by
// then and only otherwise goes back to memory ((8,8) at B = 4096, default mode: 22.4 -> 21.4 us per call).  The reader
// ticket is requested right behind the verdict and looked at when the work-group leaves; taking it only at the very end
// (`vi_leave`) exposes its round trip at the tail of every work-group: 0.3 us slower on the (2,2) kernel, measured.
__device__ __forceinline__ unsigned long long vi_peek(const VerifyCtl *c) {
    return __hip_atomic_load(&c->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool vi_verdict(VerifyCtl *c, int np, unsigned long long peeked) {
    unsigned long long v = peeked;
    bool timed_out = false;
    if ((unsigned)v < (unsigned)np) {
        v = __hip_atomic_load(&c->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while ((unsigned)v < (unsigned)np && !timed_out) {
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(&c->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull;
        }
    }
    return (v >> 32) != 0ull || timed_out;
}
__device__ __forceinline__ void vi_done(VerifyCtl *c, unsigned ticket, int readers);
__device__ __forceinline__ void vi_leave(VerifyCtl *c, int readers) {
    const unsigned ticket = __hip_atomic_fetch_add(&c->readers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    vi_done(c, ticket, readers);
}
// the last reader of the launch leaves the counters as it found them
__device__ __forceinline__ void vi_done(VerifyCtl *c, unsigned ticket, int readers) {
    if (ticket == (unsigned)(readers - 1)) {
        __hip_atomic_store(&c->word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&c->readers, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- 8-channel models: the first sum layer on the matrix cores (ratspn_gemm_wide.hip: wide_block_upper) -------------
// Per repetition the two partitions' product + sum nodes are ONE small GEMM  T[(h, o, i), s] = sum_j W[h][o][i][j] ec_h[j, s]
// with the partition h in the K index: K-slots 8h .. 8h+7 carry partition h's exponentials (lane half h of the B operand
// holds exactly its own partition's values after the leaf GEMM -- no exchange), and the rows that land in lane half h of
// the accumulator belong to partition h and have zeros in the other partition's K-slots.  Tile t holds outputs
// o = 2t, 2t+1: accumulator register u = (o & 1) * 8 + i.  Fragments [rep][S/2 tiles][hi, lo][64 lanes][8 halves] of
// the softmaxed weights scaled by 2^15 (ratspn_upper_gemm.hip: both operands are in [0, 1]; the scale keeps the low
// halves of the f16 split normal).
constexpr float kWideUpScale = 32768.f;
__host__ __device__ inline int64_t wide_upfrag_halves(int reps, int S) { return (int64_t)reps * (S / 2 > 0 ? S / 2 : 1) * 1024; }
// softmax row `row` = (rep * 2 + h) * S + o of the first sum layer, entry e = i * 8 + j, value wl
__device__ __forceinline__ void wide_upfrag_store(uint16_t *frag, int S, int row, int e, float wl) {
    const int o = row % S, ph = row / S, h = ph & 1, rep = ph >> 1;
    const int i = e >> 3, j = e & 7;
    const int tiles = S / 2 > 0 ? S / 2 : 1, t = o >> 1, u = (o & 1) * 8 + i;
    const int r = (u & 3) + 8 * (u >> 2) + 4 * h;          // A-fragment row = accumulator row of lane half h
    _Float16 hi, lo;
    split_f16(wl * kWideUpScale, hi, lo);
    uint16_t *base = frag + ((int64_t)rep * tiles + t) * 1024;
    const int own = (h * 32 + r) * 8 + j, other = ((1 - h) * 32 + r) * 8 + j;
    base[own] = __builtin_bit_cast(uint16_t, hi);
    base[512 + own] = __builtin_bit_cast(uint16_t, lo);
    base[other] = 0;                                       // the other partition's K-slots of this row
    base[512 + other] = 0;
}

// ---- one table work-group -------------------------------------------------------------------------------------------
// blk < NT * RPT: the fragments / constants / eligibility flag of repetition slot blk; otherwise (blockDim.x / 64)
// softmax rows, one wave per row (torch.log_softmax at ratspn.py:375 and :455).  dyn = the launch's dynamic LDS
// (>= gemm_prep_lds_bytes).  Every thread of the work-group calls this.
constexpr int kGemmPrepScratchInts = 64;   // reduction / verdict scratch in front of the tables' staging area
__host__ __device__ inline size_t gemm_prep_lds_bytes(int D, int I, int d) {
    return ((size_t)kGemmPrepScratchInts + (size_t)D + (size_t)4 * I * d + (size_t)cdiv(D, 32) * 4 * I +
            (size_t)cdiv(D, 16) * 4 * I + (size_t)d) * 4;   // (+ 4 d padding flags, one byte each)
}

// This thread's share (tid of nthreads) of the fingerprint of table work-group blk's inputs; the work-group's fingerprint is
// kPrepHashBase + blk + the sum of the shares (position-tagged words: any split of the threads gives the same sum).
constexpr unsigned long long kPrepHashBase = 0x9E3779B97F4A7C15ull;
template <int I>
__device__ __forceinline__ unsigned long long gemm_prep_hash_share(const GemmPrepArgs &a, int blk, int tid, int nthreads) {
    constexpr int RPT = 8 / I;
    const int nrb = a.NT * RPT, d = a.d;
    unsigned long long h = 0ull;
    if (blk < nrb) {
        if (blk < a.reps) {
            const int rho = blk;
            h += fp_range_n<I == 8>(a.mask + (int64_t)rho * 4 * d, (int64_t)4 * d * 8, 1, tid, nthreads);
            h += fp_range_n<I == 8>(a.pad ? a.pad + (int64_t)rho * 4 * d : nullptr, (int64_t)4 * d, 2, tid, nthreads);
            h += fp_range_n<I == 8>(a.loc + (int64_t)rho * 4 * I * d, (int64_t)4 * I * d * 4, 3, tid, nthreads);
            h += fp_range_n<I == 8>(a.scale + (int64_t)rho * 4 * I * d, (int64_t)4 * I * d * 4, 4, tid, nthreads);
        }
    } else {
        // the raw weight rows this work-group normalises (fingerprints are per work-group of kGemmPrepThreads / 64 rows)
        constexpr int rpb = kGemmPrepThreads / 64;
        int row = (blk - nrb) * rpb;
        for (int r = 0; r < rpb; ++r, ++row) {
            int rr = row;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (rr < a.rows[m]) {
                    // (position-dependent through the row number: rows that trade places change the sum)
                    h += fp_range_n<I == 8>(a.w[m] + (int64_t)rr * a.n[m], (int64_t)a.n[m] * 4, 5u + 8192u * (unsigned)row, tid, nthreads);
                    break;
                }
                rr -= a.rows[m];
            }
        }
    }
    return h;
}

// Round 4 (training forward: every launch rebuilds): every global read of the work-group's inputs is requested up front and
// used twice -- for the fingerprint and for the tables -- and the per-K-step / per-chunk constant sums read LDS sixteen
// variables at a time.  The first form (fingerprint passes, then `if (pad[o]) continue; .. scale[o]` per element, then
// serial 16- and 32-term LDS chains per constant) was ~60 dependent round trips: 46 us for the (8,8) model's 8 repetitions.
// Same fingerprint values (gemm_prep_hash_share) and the same summation orders as before: tables bit for bit.
template <int I>
__device__ __forceinline__ void gemm_prep_block(const GemmPrepArgs &a, int blk, int *dyn, int mode_override = -1) {
    constexpr int RPT = 8 / I;       // repetitions per 32-column tile (4 regions x I channels each)
    const int mode = mode_override >= 0 ? mode_override : a.mode;   // (a copy of the argument block with another mode lives in scratch)
    const int nrb = a.NT * RPT;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int rpb = nth >> 6;        // softmax rows per work-group
    // (no static LDS: the model kernels that carry these work-groups ask for the whole 160 KB as dynamic LDS)
    unsigned long long *red_s = reinterpret_cast<unsigned long long *>(dyn);   // [17]
    unsigned *vi_s = reinterpret_cast<unsigned *>(dyn) + 34;                    // [2]
    int &bad_s = dyn[36];
    dyn += kGemmPrepScratchInts;
    const bool rows_blk = blk >= nrb;
    const int rho = blk;
    const bool real = !rows_blk && rho < a.reps;
    const int D = a.D, d = a.d;

    // fingerprint of the bytes this work-group's outputs depend on (block local: a repetition's tables depend on its own
    // slice of mask / pad_mask / loc / scale only; a write through `param.data` moves no version counter on the host,
    // DESIGN 3.9) -> does this work-group rebuild?
    const unsigned long long stored_early = mode != kPrepBuild ? a.hash[blk] : 0ull;   // (requested first)
    unsigned long long h = tid == 0 ? kPrepHashBase + (unsigned long long)blk : 0ull;    // (once per work-group)
    auto decide = [&](unsigned long long share) -> bool {
        const unsigned long long hb = block_sum_u64(share, red_s);
        if (mode == kPrepVerify) {
            if (stored_early == hb) return false;             // nothing this work-group's outputs depend on has changed
        } else if (mode == kPrepInline) {
            if (tid == 0) {
                vi_arrive(a.ctl, stored_early != hb);
                unsigned ticket;
                const bool dirty = vi_wait(a.ctl, a.np, ticket);
                vi_done(a.ctl, ticket, a.readers);
                vi_s[0] = dirty ? 1u : 0u;
            }
            __syncthreads();
            if (vi_s[0] == 0u) return false;                  // clean launch: nothing to do
        }
        __syncthreads();
        if (tid == 0) a.hash[blk] = hb;
        return true;
    };

    if (rows_blk) {
        // a wave per softmax row (torch.log_softmax at ratspn.py:375 and :455); rows of up to 512 weights live in registers
        const int lane = tid & 63;
        const int grow = (blk - nrb) * rpb + (tid >> 6);      // row number over the (up to three) weight matrices
        int row = grow, m = 0;
        bool has = false;
#pragma unroll
        for (int mm = 0; mm < 3; ++mm) {
            if (!has) {
                if (row < a.rows[mm]) { has = true; m = mm; }
                else row -= a.rows[mm];
            }
        }
        constexpr int RV = 8;
        const int n = has ? a.n[m] : 0;
        const float *src = has ? a.w[m] + (int64_t)row * n : nullptr;
        const bool fits = n <= 64 * RV;
        const unsigned tag = 5u + 8192u * (unsigned)grow;     // (position-dependent through the row number)
        float v[RV];
        if (has && fits) {
#pragma unroll
            for (int k = 0; k < RV; ++k) v[k] = (lane + 64 * k < n) ? src[lane + 64 * k] : -INFINITY;
#pragma unroll
            for (int k = 0; k < RV; ++k)
                if (lane + 64 * k < n) h += fp_word(__float_as_uint(v[k]), (unsigned)(lane + 64 * k) * 8u + tag);
        } else if (has) {
            h += fp_range_n(src, (int64_t)n * 4, tag, lane, 64);
        }
        if (!decide(h)) return;
        if (!has) return;
        float mx = -INFINITY, sum = 0.f;
        if (fits) {
#pragma unroll
            for (int k = 0; k < RV; ++k) mx = fmaxf(mx, v[k]);
            mx = wave_reduce_max(mx);
#pragma unroll
            for (int k = 0; k < RV; ++k)
                if (lane + 64 * k < n) sum += expf(v[k] - mx);
        } else {
            for (int i = lane; i < n; i += 64) mx = fmaxf(mx, src[i]);
            mx = wave_reduce_max(mx);
            for (int i = lane; i < n; i += 64) sum += expf(src[i] - mx);
        }
        sum = wave_reduce_sum(sum);
        const float ls = logf(sum);
        auto put = [&](int i, float sv) {
            const float l = sv - mx - ls;
            const float wl = expf(l);
            a.LW[m][(int64_t)row * n + i] = l;
            a.W[m][(int64_t)row * n + i] = wl;
            if (m == 0 && a.upfrag != nullptr) wide_upfrag_store(a.upfrag, a.up_S, row, i, wl);
        };
        if (fits) {
#pragma unroll
            for (int k = 0; k < RV; ++k)
                if (lane + 64 * k < n) put(lane + 64 * k, v[k]);
        } else {
            for (int i = lane; i < n; i += 64) put(i, src[i]);
        }
        return;
    }

    // ---- a repetition's fragments, constants and eligibility flag -----------------------------------------------------
    constexpr int PSH = 20;           // posrow: (region of the repetition << 20) | position in the region
    const int KC = 16 * a.KS;
    const int NCH = (D + KC - 1) / KC, NKS = (D + 15) / 16;
    int *posrow = dyn;               // [D] packed position of variable f in this repetition, -1 if absent
    float *locs = reinterpret_cast<float *>(posrow + a.D);   // [4][I][d] the repetition's means
    float *csum = locs + 4 * I * d;   // [NCH][4I]
    float *ksum = csum + NCH * 4 * I;   // [NKS][4I]
    unsigned char *padl = reinterpret_cast<unsigned char *>(ksum + NKS * 4 * I);   // [4 d] padding flags
    const int n4 = 4 * I * d;
    constexpr int KB = 2 * I;       // first batch = the whole repetition at D <= 1024 with 512 threads (predicated-off loads still cost issue slots and code)
    unsigned vl[KB], vs[KB];
    long long mk[2];
    unsigned pd[2];
    const unsigned *wl = reinterpret_cast<const unsigned *>(a.loc) + (int64_t)rho * n4;
    const unsigned *wsc = reinterpret_cast<const unsigned *>(a.scale) + (int64_t)rho * n4;
    const int64_t m0 = (int64_t)rho * 4 * d;
    // ---- the fingerprint first (a checking launch's verdict waits for it): every first-batch load in flight at once, the
    // values stay in registers for the tables
    if (real) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = k * nth + tid;
            vl[k] = e < n4 ? wl[e] : 0u;
            vs[k] = e < n4 ? wsc[e] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = k * nth + tid;
            mk[k] = e < 4 * d ? a.mask[m0 + e] : -1;
            pd[k] = (e < 4 * d && a.pad != nullptr) ? a.pad[m0 + e] : 0u;
        }
        h += fp_range_n(a.pad ? a.pad + m0 : nullptr, (int64_t)4 * d, 2, tid, nth);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = k * nth + tid;
            if (e < n4) h += fp_word(vl[k], (unsigned)e * 8u + 3u) + fp_word(vs[k], (unsigned)e * 8u + 4u);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = k * nth + tid;
            if (e < 4 * d)
                h += fp_word((unsigned)mk[k], (unsigned)(2 * e) * 8u + 1u) + fp_word((unsigned)(mk[k] >> 32), (unsigned)(2 * e + 1) * 8u + 1u);
        }
        // (larger models: the batches behind the first)
        for (int e = KB * nth + tid; e < n4; e += nth) h += fp_word(wl[e], (unsigned)e * 8u + 3u) + fp_word(wsc[e], (unsigned)e * 8u + 4u);
        for (int e = 2 * nth + tid; e < 4 * d; e += nth) {
            const long long mv = a.mask[m0 + e];
            h += fp_word((unsigned)mv, (unsigned)(2 * e) * 8u + 1u) + fp_word((unsigned)(mv >> 32), (unsigned)(2 * e + 1) * 8u + 1u);
        }
    }
    if (!decide(h)) return;
    // ---- rebuilding: positions, padding flags and means into LDS -------------------------------------------------------
    for (int f = tid; f < D; f += nth) posrow[f] = -1;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    if (real) {
        for (int e0 = 0; e0 < 4 * d; e0 += 2 * nth) {
            if (e0 > 0) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = e0 + k * nth + tid;
                    mk[k] = e < 4 * d ? a.mask[m0 + e] : -1;
                    pd[k] = (e < 4 * d && a.pad != nullptr) ? a.pad[m0 + e] : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = e0 + k * nth + tid;
                if (e < 4 * d) {
                    padl[e] = (unsigned char)(pd[k] != 0u);
                    const int q = e / d;
                    if (!pd[k] && mk[k] >= 0 && mk[k] < D) posrow[(int)mk[k]] = (q << PSH) | (e - q * d);
                }
            }
        }
    }
    __syncthreads();
    // eligibility of the repetition for the expanded form: scale == 1 everywhere, |mu| <= kExpandBound
    bool bad = false;
    if (real) {
        for (int e0 = 0; e0 < n4; e0 += KB * nth) {
            if (e0 > 0) {
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int e = e0 + k * nth + tid;
                    vl[k] = e < n4 ? wl[e] : 0u;
                    vs[k] = e < n4 ? wsc[e] : 0u;
                }
            }
            // (e / d and e % d stepped from one division per batch: sixteen unrolled runtime divisions were 800 instructions,
            // and this code runs once per wave -- instruction fetch is what it costs)
            int quo = (e0 + tid) / d, rem = (e0 + tid) - quo * d;
            const int sq = nth / d, sr = nth - sq * d;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int e = e0 + k * nth + tid;
                if (e < n4) {
                    const float mu = __uint_as_float(vl[k]);
                    locs[e] = mu;
                    const int rr = quo / I;                   // region of the repetition: e / (I d)
                    if (!padl[rr * d + rem]) bad = bad || !(fabsf(mu) <= kExpandBound) || (__uint_as_float(vs[k]) != 1.0f);
                }
                quo += sq; rem += sr;
                if (rem >= d) { rem -= d; ++quo; }
            }
        }
    }
    if (a.ablate & 1) return;
    if (bad) bad_s = 1;
    __syncthreads();
    const int t = rho / RPT, ap = rho - t * RPT;
    // fragment entries: (K-step, lane half, column of this repetition) -> 8 consecutive variables
    for (int e = tid; e < ((a.ablate & 2) ? 0 : a.NKSP * 2 * 4 * I); e += nth) {
        const int col = e % (4 * I);
        const int hg = (e / (4 * I)) & 1;
        const int ks = e / (8 * I);
        const int q = col / I, k = col - q * I;
        const int hh = q >> 1, qq = q & 1;
        const int u = (ap * 2 + qq) * I + k;               // accumulator register of the lane half
        const int row = (u & 3) + 8 * (u >> 2) + 4 * hh;   // MFMA output row = A-fragment row
        int pp[8];
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            const int f = ks * 16 + hg * 8 + el;
            pp[el] = (real && f < D) ? posrow[f] : -1;
        }
        half8 mh, ml, ch, cl;
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            float mu = 0.f, cc = 0.f;
            if ((pp[el] >> PSH) == q) {                    // (-1 >> 20 = -1: absent)
                mu = locs[(q * I + k) * d + (pp[el] & ((1 << PSH) - 1))];
                cc = -fmaf(0.5f * mu, mu, kLogSqrt2Pi);
            }
            _Float16 hi, lo;
            split_f16(mu, hi, lo);
            mh[el] = hi; ml[el] = lo;
            split_f16(cc, hi, lo);
            ch[el] = hi; cl[el] = lo;
        }
        const int64_t o = (((int64_t)ks * a.NT + t) * 2) * 512 + (hg * 32 + row) * 8;
        *reinterpret_cast<half8 *>(a.mtab + o) = mh;
        *reinterpret_cast<half8 *>(a.mtab + o + 512) = ml;
        *reinterpret_cast<half8 *>(a.ctab + o) = ch;
        *reinterpret_cast<half8 *>(a.ctab + o + 512) = cl;
    }
    // The slice mapping's compact table (ratspn_gemm_slice.hip): three quarters of a repetition's fragment entries are
    // structural zeros -- variable f belongs to ONE of the repetition's four regions -- so entry (K-step, lane half,
    // channel) holds f's mean in whichever region that is, and a lane's keep-mask says which of its 8 variables belong to
    // its own column's region.  A quarter of the bytes and of the requests on the compute unit's request path.
    if (a.stab != nullptr && !(a.ablate & 2)) {
        for (int e = tid; e < a.NKSP * 2 * I; e += nth) {
            const int k = e % I, hg = (e / I) & 1, ks = e / (2 * I);
            half8 mh, ml;
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int f = ks * 16 + hg * 8 + el;
                const int pp = (real && f < D) ? posrow[f] : -1;
                const float mu = pp >= 0 ? locs[((pp >> PSH) * I + k) * d + (pp & ((1 << PSH) - 1))] : 0.f;
                _Float16 hi, lo;
                split_f16(mu, hi, lo);
                mh[el] = hi; ml[el] = lo;
            }
            const int64_t o = ((int64_t)ks * a.NT + t) * (kSliceTabBytes / 2) + (hg * 4 * I + ap * I + k) * 8;
            *reinterpret_cast<half8 *>(a.stab + o) = mh;
            *reinterpret_cast<half8 *>(a.stab + o + kSliceTabBytes / 4) = ml;
        }
        // keep-masks [slice wave][lane][K-step of the wave (8 slots)][tile (2)]: bit el = variable el of the lane's 8
        const int nwv = a.NKSP / 7;
        for (int e = tid; e < nwv * 7 * 2 * 4 * I; e += nth) {
            const int k = e % I, q = (e / I) & 3, hg = (e / (4 * I)) & 1, kk = (e / (8 * I)) % 7, wv = e / (56 * I);
            const int hh = q >> 1, qq = q & 1;
            const int u = (ap * 2 + qq) * I + k;
            const int row = (u & 3) + 8 * (u >> 2) + 4 * hh;
            unsigned bits = 0u;
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int f = (wv * 7 + kk) * 16 + hg * 8 + el;
                const int pp = (real && f < D) ? posrow[f] : -1;
                if ((pp >> PSH) == q) bits |= 1u << el;
            }
            a.smask[((wv * 64 + hg * 32 + row) * 8 + kk) * 2 + (t & 1)] = (unsigned char)bits;
        }
    }
    // - sum_f (mu^2/2 + log sqrt(2 pi)) over the variables [f0, f1) that belong to column (q, k)'s region, ascending f (fixed
    // summation order: launches must agree bit for bit); the positions of 16 variables are read together
    // One loop over both families of constants: per (chunk, column) [bias, csum] and per (K-step of 16 features, column)
    // [bias_ks, ksum; also the per-slice sums of the small-batch kernel below].
    const int n_c = NCH * 4 * I, n_k = NKS * 4 * I;
    for (int e = tid; e < ((a.ablate & 4) ? 0 : n_c + n_k); e += nth) {
        const bool chunk = e < n_c;
        const int ee = chunk ? e : e - n_c;
        const int col = ee % (4 * I), c = ee / (4 * I);
        const int q = col / I, k = col - q * I;
        const int span = chunk ? KC : 16;
        const int f0 = c * span, f1 = min(D, f0 + span);
        float sum = 0.f;
        if (real) {
            for (int fb = f0; fb < f1; fb += 16) {
                int pp[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) pp[i] = (fb + i < f1) ? posrow[fb + i] : -1;
                float mu[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool ok = (pp[i] >> PSH) == q;
                    mu[i] = locs[(q * I + k) * d + (ok ? (pp[i] & ((1 << PSH) - 1)) : 0)];
                    mu[i] = ok ? -fmaf(0.5f * mu[i], mu[i], kLogSqrt2Pi) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += mu[i];     // (absent variables add an exact 0)
            }
        }
        const int hh = q >> 1, qq = q & 1;
        const int u = (ap * 2 + qq) * I + k;
        if (chunk) {
            a.bias[((c * 2 + hh) * a.NT + t) * 16 + u] = sum;
            csum[ee] = sum;
        } else {
            a.bias_ks[((c * 2 + hh) * a.NT + t) * 16 + u] = sum;
            ksum[ee] = sum;
        }
    }
    __syncthreads();
    if (tid < 4 * I) {
        const int col = tid;
        const int q = col / I, k = col - q * I;
        float sum = 0.f;
        for (int c = 0; c < NCH; ++c) sum += csum[c * 4 * I + col];
        const int hh = q >> 1, qq = q & 1;
        const int u = (ap * 2 + qq) * I + k;
        a.bias_row[(hh * a.NT + t) * 16 + u] = sum;
    }
    for (int e = tid; e < kGemmSmallWaves * 4 * I; e += nth) {
        const int col = e % (4 * I), w = e / (4 * I);
        const int q = col / I, k = col - q * I;
        float sum = 0.f;
        for (int ks = w * NKS / kGemmSmallWaves; ks < (w + 1) * NKS / kGemmSmallWaves; ++ks) sum += ksum[ks * 4 * I + col];
        const int hh = q >> 1, qq = q & 1;
        const int u = (ap * 2 + qq) * I + k;
        a.bias_sl[((w * 2 + hh) * a.NT + t) * 16 + u] = sum;
    }
    if (tid == 0) a.elig[rho] = bad_s ? 0 : 1;
}

// ---- table work-groups inside a model kernel's launch (kPrepInline) ------------------------------------------------------
// A clean launch -- every launch but the one after a parameter changed -- only needs the fingerprint and the verdict: the
// compact loops of gemm_prep_hash_share, nothing staged, nothing kept in registers for a rebuild that does not happen.
// Measured, same box, (2,2) at B = 4096 per graph-replayed model(x): this form 9.87 us; gemm_prep_block's own
// loads-first fingerprint (built for the rebuilding launches of a training step) 10.40 us; the rebuild behind a noinline
// call 15.6 us default AND 12.9 us trusted -- the call makes the kernel use 536 bytes of scratch per lane, which costs
// 3.5 us per launch whether or not the call is ever made (the 0.2 us of tools/ubench/scratch_cost.hip was for 64 bytes).
template <int I>
__device__ __forceinline__ void gemm_prep_rebuild_cold(const GemmPrepArgs *a, int blk, int *dyn) {
    gemm_prep_block<I>(*a, blk, dyn, kPrepBuild);
}
template <int I>
__device__ __forceinline__ void gemm_prep_block_inline(const GemmPrepArgs &a, int blk, int *dyn) {
    unsigned long long *red_s = reinterpret_cast<unsigned long long *>(dyn);   // [17]
    unsigned *vi_s = reinterpret_cast<unsigned *>(dyn) + 34;                    // [2]
    const unsigned long long stored = a.hash[blk];                             // (requested first)
    unsigned long long h = threadIdx.x == 0 ? kPrepHashBase + (unsigned long long)blk : 0ull;
    h += gemm_prep_hash_share<I>(a, blk, (int)threadIdx.x, (int)blockDim.x);
    h = block_sum_u64(h, red_s);
    if (threadIdx.x == 0) {
        vi_arrive(a.ctl, stored != h);
        unsigned ticket;
        const bool dirty = vi_wait(a.ctl, a.np, ticket);
        vi_done(a.ctl, ticket, a.readers);
        vi_s[0] = dirty ? 1u : 0u;
    }
    __syncthreads();
    if (vi_s[0] == 0u) return;                                                 // clean launch: nothing to do
    __syncthreads();
    gemm_prep_rebuild_cold<I>(&a, blk, dyn);
}

// ---- the table-free nodes of a dirty launch --------------------------------------------------------------------------
// out[o] = logsumexp_{i,j}(a[i] + c[j] + log_softmax(w[o, :])[i, j]) straight from the RAW weights w [NO][NI*NI]
// (ProductLayer.forward ratspn.py:280-285 followed by SumLayer.forward :375-377): two log-sum-exps per node, rolled
// loops -- slow by design, this is the route of the one launch that finds its tables stale.
template <int NI, int NO>
__device__ __forceinline__ void prodsum_node_raw(const float (&a)[NI], const float (&c)[NI], const float *w, float *slot,
                                                 float (&out)[NO]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        slot[i] = a[i];
        slot[NI + i] = c[i];
    }
#pragma unroll 1
    for (int o = 0; o < NO; ++o) {
        const float *wo = w + o * NI * NI;
        float mw = -INFINITY, m = -INFINITY;
#pragma unroll 1
        for (int e = 0; e < NI * NI; ++e) {
            const float we = wo[e];
            mw = fmaxf(mw, we);
            m = fmaxf(m, slot[e / NI] + slot[NI + e % NI] + we);
        }
        float sw = 0.f, sv = 0.f;
#pragma unroll 1
        for (int e = 0; e < NI * NI; ++e) {
            const float we = wo[e];
            sw += expf(we - mw);
            if (m > -INFINITY) sv += expf(slot[e / NI] + slot[NI + e % NI] + we - m);
        }
        const float res = (m > -INFINITY) ? (m + logf(sv)) - (mw + logf(sw)) : -INFINITY;
#pragma unroll
        for (int q = 0; q < NO; ++q)
            if (q == o) out[q] = res;
    }
}

// log-sum-exp of one raw weight row (the normaliser of torch.log_softmax), by one thread
__device__ __forceinline__ float raw_row_lse(const float *w, int n) {
    float m = -INFINITY;
#pragma unroll 1
    for (int e = 0; e < n; ++e) m = fmaxf(m, w[e]);
    float s = 0.f;
#pragma unroll 1
    for (int e = 0; e < n; ++e) s += expf(w[e] - m);
    return m + logf(s);
}

// (m, s) of one repetition's share of the root: logsumexp_{i,j}(a[i] + c[j] + w[i, j] - lse_row) = m + log s
template <int NI>
__device__ __forceinline__ void root_partial_raw(const float (&a)[NI], const float (&c)[NI], const float *w, float lse_row,
                                                 float *slot, float &m_out, float &s_out) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        slot[i] = a[i];
        slot[NI + i] = c[i];
    }
    float m = -INFINITY;
#pragma unroll 1
    for (int e = 0; e < NI * NI; ++e) m = fmaxf(m, slot[e / NI] + slot[NI + e % NI] + (w[e] - lse_row));
    float s = 0.f;
    if (m > -INFINITY) {
#pragma unroll 1
        for (int e = 0; e < NI * NI; ++e) s += expf(slot[e / NI] + slot[NI + e % NI] + (w[e] - lse_row) - m);
    }
    m_out = m;
    s_out = s;
}

}  // namespace dpk
