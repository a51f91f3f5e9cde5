// Shared host/device helpers for libdeeprob_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <float.h>
#include "../../include/deeprob_hip.h"

// gfx950 only, and not as a formality: the "last work-group finishes" protocols of this library (params_fingerprint_kernel,
// level_jacobian_tail, the split-K tickets of coupling_bwd.hip, the in-launch table verdict of ratspn_gemm_prep.h) order
// their data with device-scope relaxed atomics + a work-group-scope release + s_waitcnt 0 in front of the ticket instead of
// __threadfence() (which writes back the XCD's L2: 17-40 us here).  That relies on gfx950's acknowledgement semantics: an
// agent-scope atomic (sc1) is performed at the memory side, and s_waitcnt vmcnt(0) returns only when it has been.  Another
// architecture needs the release / acquire fences back.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdeeprob_hip is written for gfx950 (MI355X): see the note on fence-free ticket protocols in common.h"
#endif

namespace dpk {

void set_error(const char *fmt, ...);
// one-shot measurement hook (dpk_profile_next_kernel[_of]): the events the caller wants recorded around the next launch
// of the kernel with this id (DPK_KERNEL_* in deeprob_hip.h), or nulls
void profile_take(hipEvent_t *start, hipEvent_t *stop, int kernel_id = 1);
// Per-DEVICE launch state (a process may drive several GPUs): the compute-unit count of the calling thread's current
// device, and the raised dynamic-LDS limit of a kernel, set once per (kernel, device) -- hipFuncSetAttribute and the
// CU count both belong to a device, not to the process.  ensure_dynamic_lds returns DPK_OK or DPK_ELAUNCH (error set).
int device_cus();
int ensure_dynamic_lds(const void *kernel, int bytes);

// Marginalised-evidence hint of the RAT-SPN forward kernels: a work-group that meets NaN evidence stores the launch
// number in a host-mapped word; the host reads it without synchronising (a stale value only costs speed) and picks the
// kernel build / variant that keeps such inputs fast while one of the recent launches did so.  One word PER WORKSPACE
// (round 4; a process-wide word before: two models -- or a clean and a marginalised stream -- in one process steered each
// other's variant): slots of one host-mapped page per process, keyed by the workspace's address, released by
// dpk_workspace_forget.  A captured HIP graph freezes the variant that was current at capture time (results identical).
// Returns whether a launch of this workspace within its last 256 met marginalised evidence; *dev_word = nullptr (and
// false) when no slot could be had.
bool slow_hint_next(const void *ws_key, int **dev_word, int *launch_seq);
// releases what the library keeps per workspace address (hint slot, params_gate slots) for keys in [base, base + bytes)
void workspace_forget(const void *base, int64_t bytes);

// ---- device-side check of cached parameter tables (DPK_FLAG_PARAMS_VERIFY) ------------------------------------------
// The host can only tell that a parameter MAY have changed from its address / version counter; a write through
// `param.data` bumps neither.  An entry point that keeps tables derived from parameters therefore fingerprints the live
// parameter bytes on the device on every call (one small launch: 64-bit position-dependent hash, last block compares it
// with the hash of the bytes the tables were built from) and its table kernels start with `if (gate_closed(gate))
// return;` -- they run for real only when the bytes differ.  params_gate() returns the device word the kernels test
// (1 = rebuild), or nullptr when no slot could be had (the caller then rebuilds unconditionally).  `key` names the table
// set (a pointer into the workspace that holds it); verify = false records the hash of a build that happens anyway.
struct FpSeg {
    const void *p;
    int64_t bytes;
};
constexpr int kFpMaxSegs = 10;
const unsigned *params_gate(const void *key, const FpSeg *segs, int nseg, bool verify, hipStream_t st);
__device__ __forceinline__ bool gate_closed(const unsigned *gate) { return gate != nullptr && *gate == 0u; }
// zero `bytes` of device memory unless the gate is closed (the stream-ordered memset of a gated table build)
int gated_zero(void *p, int64_t bytes, const unsigned *gate, hipStream_t st);
// What an entry point does with its tables for these flags: DPK_FLAG_PARAMS_CACHED -> nothing; DPK_FLAG_PARAMS_VERIFY ->
// fingerprint, table kernels gated on the verdict; neither -> table kernels unconditionally (the hash is recorded).
struct TablePlan {
    bool run;
    const unsigned *gate;
};
inline TablePlan plan_tables(uint32_t flags, const void *key, const FpSeg *segs, int nseg, hipStream_t st) {
    if (flags & DPK_FLAG_PARAMS_CACHED) return {false, nullptr};
    const bool verify = (flags & DPK_FLAG_PARAMS_VERIFY) != 0;
    const unsigned *g = params_gate(key, segs, nseg, verify, st);
    return {true, verify ? g : nullptr};   // (no slot to be had: nullptr = rebuild unconditionally)
}

#define DPK_REQUIRE(cond, code, ...)       \
    do {                                   \
        if (!(cond)) {                     \
            dpk::set_error(__VA_ARGS__);   \
            return (code);                 \
        }                                  \
    } while (0)

#define DPK_CHECK_LAUNCH(what)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            dpk::set_error("%s: %s", (what), hipGetErrorString(e__));               \
            return DPK_ELAUNCH;                                                     \
        }                                                                           \
    } while (0)

// A launch reports its own status: whatever an earlier runtime call on this thread (the caller's own included, e.g. a
// failed hipEventElapsedTime) left in the last-error slot is dropped first, so DPK_CHECK_LAUNCH never blames a kernel
// for it.
#define DPK_LAUNCH(...)                    \
    do {                                   \
        (void)hipGetLastError();           \
        hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kCompactRec = 48;      // bytes of a compact block record: 8 means + 4 u16 row offsets + pad
constexpr float kExpandBound = 6.0f; // |x|, |mu| bound under which the unit-scale leaf uses the expanded square
constexpr int kChunk = 64;          // features staged in LDS per chunk (FC)
constexpr int kLeafWaves = 8;       // waves per work-group of the leaf / fused kernels
constexpr float kLogSqrt2Pi = 0.918938533204672741780329736406f;

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Workspace carve-up shared by every RAT-SPN entry point.
//
// Regions are handled in groups of QB consecutive regions (fused model: QB = 2^depth = one
// repetition).  A group's table entries form ONE stream in the order its wave consumes them: for
// chunk c (features [c*kChunk, (c+1)*kChunk)), for region slot q, the sorted variables of region
// g*QB+q inside the chunk; every (c,q) segment is padded to a multiple of kBlock entries with neutral
// entries (LDS row kChunk = all zeros, parameters 0).  SP = stream capacity per group.
constexpr int kBlock = 4;
constexpr int kTableSlack = 16;  // neutral entries the software-pipelined readers may run into
struct VerifyCtl;
constexpr int kWideSplitBlocks = 128;     // 8-channel models, blocks of 32 samples shared by two work-groups: up to this many per launch
constexpr int kSliceTabBytes = 512;       // compact mean table: bytes per (K-step, column tile) -- [hi, lo][lane half][repetition, channel][8 variables]
constexpr int kSliceVerifyBytes = 2304;   // ratspn_gemm_slice.hip SliceVerify, in the workspace behind the VerifyCtl
struct RatWs {
    // structure tables (depend on mask / pad_mask only)
    int *fl1;     // [G*SP] LDS byte offset of the entry's row for 1 sample per lane
    int *fl2;     //        ... for 2 samples per lane
    int *srcr;    // [G*SP] r*d + j of the entry (position in loc/scale), -1 for dummy / neutral
    int *feat;    // [G*SP] variable id, -1 for neutral entries
    int *nblk;    // [G*NC*QB] blocks of every segment
    int *segoff;  // [G*NC*QB] first stream position of every segment
    // parameter tables (rebuilt on every call)
    float *par;   // [G*ncb*SP*2CB] {p0[CB], p1[CB]} per entry, channel-block major
    float *cel;   // [G*ncb*SP*CB]  additive constant per entry
    float *biasc; // [R*NC*I] per-(region, chunk) sum of cel
    float *biasx; // [R*NC*I] the same minus half the sum of squared means (expanded unit-scale form)
    int *unit;    // [R] Gaussian leaves: 1 if every scale of the region equals 1, 2 if additionally every
                  // |mean| <= kExpandBound (the expanded form x^2 - 2 x mu + mu^2 is then safe)
    float *rec;   // [G*(SP/4)*(4+8I)] block records staged into LDS by the kernels (I <= 2 only)
    int tabcap;   // LDS bytes per wave for the records of one chunk, 0 = records not used
    int tabcap_c; // the same for the compact 48-byte records of the unit-scale two-channel kernel
    float *w[3];  // linear softmax weights: sum layer 0, sum layer 1, root
    float *lw[3]; // log-softmax weights
    // MFMA leaf layer of the fused depth-2 model (ratspn_gemm.hip): the means as f16 (hi, lo) MFMA fragments
    uint16_t *gm_tab;  // [NKSP][NT][2][64 lanes][8 halves]  mean table (x . mu GEMM)
    uint16_t *gc_tab;  //  same layout: mu^2/2 + log sqrt(2 pi)  (marginalised-evidence correction GEMM)
    float *gbias;      // [NCH][2][NT][16] per-(chunk, column) constants in the accumulator order of a lane
    float *gbias_row;  // [2][NT][16] the sums over all chunks
    float *gbias_ks;   // [NKS][2][NT][16] the same per K-step of 16 features (small-batch kernel, ratspn_gemm_small.hip)
    float *gbias_sl;   // [8][2][NT][16] ... per feature slice of its 8 waves (K-steps [w NKS/8, (w+1) NKS/8))
    int *gelig;        // [NT*RPT] 1: repetition is unit-scale with bounded means
    unsigned long long *ghash;   // [NT*RPT] fingerprint of the parameter bytes each repetition's tables were built from,
                                 // then one per softmax-row work-group of the table build (ratspn_gemm_prep.h)
    struct VerifyCtl *gctl;      // the verdict word of a launch that checks its tables itself (ratspn_gemm_prep.h)
    uint16_t *gs_tab;            // two-channel models: the mean table without its structural zeros (ratspn_gemm_slice.hip)
    unsigned char *gs_mask;      // ... and which of a lane's entries its column keeps, per slice wave
    uint16_t *gup;               // 8-channel models: MFMA fragments of the first sum layer (ratspn_gemm_prep.h: wide_upfrag_*)
    float *gwx_part;             // ... root partials of the two work-groups that share a block (ratspn_gemm_wide.hip, kWideSplit*)
    unsigned *gwx_tick;          // ... and the blocks' tickets (only ever counted up)
    void *lg;          // tables of the leaf-only MFMA kernel (leaf_gemm_ws_bytes), null when the shape is outside it
    int g_nt, g_nksp;  // column tiles of 32, K-steps of 16 features (padded to whole chunks); 0 = not built
    int64_t bytes;
    int NC, SP, G, QB;
};

// geometry of the MFMA leaf layer (ratspn_gemm.hip)
constexpr int kGemmKS = 4;                // K-steps of 16 features per staged chunk
constexpr int kGemmKC = 16 * kGemmKS;     // features per chunk
constexpr int kGemmMaxNT = 4;             // column tiles of 32 the fused kernel is built for
constexpr int kGemmSmallWaves = 8;        // small-batch kernel: waves per 32-sample tile = slices of the feature axis
constexpr int kGemmPrepThreads = 512;     // threads of a table work-group (stand-alone launch and in-launch alike: the
                                          // softmax-row fingerprints are per work-group of kGemmPrepThreads / 64 rows)
constexpr int kGemmSmallMaxK = 8;         // ... and the K-steps (of 16 features) one of them can hold in registers
static inline bool gemm_shape_ok(int D, int depth, int reps, int I, int S) {
    if (depth != 2 || (D % 4) != 0 || reps < 1) return false;
    // 8 channels: one column tile per repetition, a wave per tile (ratspn_gemm_wide.hip); x tile + weights in LDS
    if (I == 8) return (S == 2 || S == 4 || S == 8) && reps <= 8 && (D + 15) / 16 <= 64;
    if (!(I == 2 || I == 4) || !(S == 2 || S == 4)) return false;
    return (reps * 4 * I + 31) / 32 <= kGemmMaxNT;
}

// geometry of the leaf-only MFMA kernel (ratspn_leaf_gemm.hip): column groups of NTG tiles, chunks of 32 features
// training forward of the fused MFMA route (dpk_ratspn_forward_train): see GemmArgs::emit_* in ratspn_gemm_fused.h
struct GemmEmit { float *leaf, *sum, *out; };

constexpr int kLeafGemmKS = 2;
constexpr int kLeafPrepParts = 4;   // work-groups per region of the table kernel (the last one owns the constants)
static inline int leaf_ntg(int I) { return I >= 4 ? 4 : I; }
static inline bool leaf_gemm_shape_ok(int D, int R, int I, int d) {
    if (!(I == 2 || I == 4 || I == 8 || I == 16) || (D % 4) != 0 || R < 1 || d < 1) return false;
    return (D + 16 * kLeafGemmKS - 1) / (16 * kLeafGemmKS) <= 64;   // chunk bit masks of the kernel
}
static inline int64_t leaf_gemm_ws_bytes(int D, int R, int I) {
    const int NTG = leaf_ntg(I), NG = (int)(((int64_t)R * I + 32 * NTG - 1) / (32 * NTG));
    const int NCH = (D + 16 * kLeafGemmKS - 1) / (16 * kLeafGemmKS), NKSP = NCH * kLeafGemmKS;
    const int64_t tab = align_up((int64_t)NG * NKSP * (2 * NTG + 2) * 1024, 256);
    const int64_t bias = align_up((int64_t)NG * NCH * 2 * NTG * 16 * 4, 256);
    const int64_t brow = align_up((int64_t)NG * 2 * NTG * 16 * 4, 256);
    return 2 * tab + bias + brow + align_up((int64_t)R * 4, 256) + align_up((int64_t)R * 8 * kLeafPrepParts, 256);   // (+ fingerprints)
}

// region-group size used by the per-layer leaf operators (the fused model uses 2^depth)
static inline int leaf_group(int R) { return (R % 4 == 0) ? 4 : 2; }

inline RatWs carve_ratspn_ws(void *base, int D, int R, int d, int I, int QB, int depth, int reps, int S,
                             int C) {
    RatWs w{};
    char *p = (char *)base;
    int64_t o = 0;
    auto take = [&](int64_t n_bytes) {
        char *q = p ? p + o : nullptr;
        o = align_up(o + n_bytes, 256);
        return q;
    };
    const int NC = cdiv(D, kChunk);
    w.NC = NC;
    w.QB = QB;
    w.G = R / QB;
    w.SP = (int)align_up((int64_t)QB * d + (kBlock - 1) * QB * NC, kBlock) + kTableSlack;
    const int64_t GS = (int64_t)w.G * w.SP;
    w.fl1 = (int *)take(GS * 4);
    w.fl2 = (int *)take(GS * 4);
    w.srcr = (int *)take(GS * 4);
    w.feat = (int *)take(GS * 4);
    w.nblk = (int *)take((int64_t)w.G * NC * QB * 4);
    w.segoff = (int *)take((int64_t)w.G * NC * QB * 4);
    w.par = (float *)take(GS * 2 * I * 4);
    w.cel = (float *)take(GS * I * 4);
    w.biasc = (float *)take((int64_t)R * NC * I * 4);
    w.biasx = (float *)take((int64_t)R * NC * I * 4);
    w.unit = (int *)take((int64_t)R * 4);
    w.rec = nullptr;
    w.tabcap = 0;
    w.tabcap_c = 0;
    if (I <= 2) {
        const int recb = 4 + 8 * I;
        w.rec = (float *)take((int64_t)w.G * (w.SP / kBlock) * recb * 4);
        // blocks of one (group, chunk): the regions of one repetition are disjoint, so a group spanning
        // `rpg` repetitions holds at most rpg*(kChunk + padding) entries of a chunk, plus < 1 block of
        // padding per region and two blocks of pipeline run-ahead
        const int per_rep = cdiv(D, d);
        const int rpg = cdiv(QB, per_rep);
        const int blocks = rpg * (kChunk / kBlock + 1) + QB + 2;
        const int cap = (int)align_up((int64_t)blocks * recb * 4, 16);
        if (cap <= 3 * 1024) w.tabcap = cap;  // three 16-byte loads per lane (NTL in the kernel)
        w.tabcap_c = (int)align_up((int64_t)blocks * kCompactRec, 16);
    }
    // sum layers (only meaningful for the fused model entry point)
    int64_t n0 = 0, n1 = 0, nr = 0;
    if (depth >= 1 && reps >= 1) {
        const int Q = 1 << depth;
        if (depth >= 2) n0 = (int64_t)reps * (Q / 2) * S * I * I;
        if (depth >= 3) n1 = (int64_t)reps * (Q / 4) * S * S * S;
        const int nlast = depth >= 2 ? S : I;
        nr = (int64_t)C * reps * nlast * nlast;
    }
    const int64_t n[3] = {n0, n1, nr};
    for (int i = 0; i < 3; ++i) {
        w.w[i] = (float *)take(n[i] * 4);
        w.lw[i] = (float *)take(n[i] * 4);
    }
    w.g_nt = 0;
    w.g_nksp = 0;
    if (gemm_shape_ok(D, depth, reps, I, S)) {
        w.g_nt = (reps * 4 * I + 31) / 32;
        w.g_nksp = ((D + kGemmKC - 1) / kGemmKC) * kGemmKS;
        const int64_t tab = (int64_t)w.g_nksp * w.g_nt * 2 * 1024;
        w.gm_tab = (uint16_t *)take(tab);
        w.gc_tab = (uint16_t *)take(tab);
        w.gbias = (float *)take((int64_t)((D + 31) / 32) * 2 * w.g_nt * 16 * 4);
        w.gbias_row = (float *)take((int64_t)2 * w.g_nt * 16 * 4);
        w.gbias_ks = (float *)take((int64_t)((D + 15) / 16) * 2 * w.g_nt * 16 * 4);
        w.gbias_sl = (float *)take((int64_t)kGemmSmallWaves * 2 * w.g_nt * 16 * 4);
        w.gelig = (int *)take((int64_t)w.g_nt * 8 * 4);
        w.ghash = (unsigned long long *)take(((int64_t)w.g_nt * 8 + cdiv(reps * 2 * S + C, 4) + 4) * 8);
        w.gctl = (struct VerifyCtl *)take(64 + kSliceVerifyBytes);   // (+ the slice mapping's counters behind it)
        w.gup = nullptr;
        w.gs_tab = nullptr;
        w.gs_mask = nullptr;
        if (I == 2) {
            w.gs_tab = (uint16_t *)take((int64_t)w.g_nksp * w.g_nt * kSliceTabBytes);
            w.gs_mask = (unsigned char *)take((int64_t)cdiv(w.g_nksp, 7) * 64 * 16);
        }
        if (I == 8) w.gup = (uint16_t *)take((int64_t)reps * (S / 2 > 0 ? S / 2 : 1) * 1024 * 2);
        w.gwx_part = nullptr;
        w.gwx_tick = nullptr;
        if (I == 8 && reps > 4) {
            w.gwx_part = (float *)take((int64_t)kWideSplitBlocks * 2 * 32 * C * 2 * 4);
            w.gwx_tick = (unsigned *)take((int64_t)kWideSplitBlocks * 4);
        }
    }
    w.lg = nullptr;
    if (leaf_gemm_shape_ok(D, R, I, d)) w.lg = take(leaf_gemm_ws_bytes(D, R, I));
    w.bytes = o;
    return w;
}

// ---- device helpers -------------------------------------------------------
// Counter-based dropout decision shared by the training forward and backward kernels: element `idx` of a call
// with seed `seed` is dropped iff the top 24 bits of splitmix64(seed + idx * golden) fall below p * 2^24.
// The same (seed, idx) gives the same answer in every kernel, so no mask tensor is stored.
__host__ __device__ __forceinline__ bool dropout_hit(uint64_t seed, uint64_t idx, float p) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) < p * 16777216.0f;
}


__device__ __forceinline__ float nan_to_num_f(float t) {
    // torch.nan_to_num_ defaults: nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
    if (t != t) return 0.0f;
    return fminf(fmaxf(t, -FLT_MAX), FLT_MAX);
}

// The argument of a responsibility, in_a + in_c + lw - out, for log-likelihoods of magnitude 10^2..10^3: added left to
// right each of the three roundings is half an ulp of the LARGE operands (3e-5 at 560, 6e-5 at 1120) and lands in the
// responsibility as a relative error of that size -- which the reference does not show, because its logsumexp normalises
// the very same rounded terms (forward and backward errors cancel for the dominant input).  Here the large magnitudes
// cancel exactly first: (hi, lo) = in_a + in_c without error (Knuth's two-sum), hi - out is exact when the two are within
// a factor of two (Sterbenz; otherwise the term is e^{-|large|} = 0 either way), and what is rounded is small.
struct TwoSum { float hi, lo; };
__device__ __forceinline__ TwoSum two_sum(float a, float b) {
    const float s = a + b, bb = s - a;
    const float lo = (a - (s - bb)) + (b - bb);
    return {s, fabsf(s) < INFINITY ? lo : 0.f};      // (a dead input, -inf: the error term would be inf - inf)
}
// (clamped: where the evidence is so large that the stored `out` is off by whole units -- ulp(4e8) = 32 -- the terms of a node
// may exceed it by more than exp() can hold; the normalisation that follows only needs them finite)
constexpr float kRespArgMax = 60.f;
__device__ __forceinline__ float resp_arg(const TwoSum ac, float lw, float out) { return fminf((ac.hi - out) + (lw + ac.lo), kRespArgMax); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the G = 4 / 16 / 64 consecutive lanes of a group on the vector ALU's DPP path (no LDS crossbar: __shfl_xor is a
// ds_bpermute round trip per step), result in every lane of the group: quads, half rows, rows (16 lanes), then the row
// results walk to lane 63, which is read back wave-uniform.
#define DPK_DPP_ADD(x, ctrl, rmask) \
    ((x) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), ctrl, rmask, 0xf, false)))
template <int G> __device__ __forceinline__ float dpp_group_sum(float v) {
    static_assert(G == 1 || G == 4 || G == 16 || G == 64, "quad, row or wave");
    if (G >= 4) {
        v = DPK_DPP_ADD(v, 0xB1, 0xf);     // quad_perm [1,0,3,2]
        v = DPK_DPP_ADD(v, 0x4E, 0xf);     // quad_perm [2,3,0,1]
    }
    if (G >= 16) {
        v = DPK_DPP_ADD(v, 0x141, 0xf);    // row_half_mirror
        v = DPK_DPP_ADD(v, 0x140, 0xf);    // row_mirror
    }
    if (G == 64) {
        v = DPK_DPP_ADD(v, 0x142, 0xa);    // row_bcast:15 into rows 1 and 3
        v = DPK_DPP_ADD(v, 0x143, 0xc);    // row_bcast:31 into rows 2 and 3
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    }
    return v;
}
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- fingerprints of parameter bytes (DPK_FLAG_PARAMS_VERIFY) ------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long fp_mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// This thread's share of the fingerprint of `bytes` bytes at p (4-byte words where the range is word aligned, single
// bytes otherwise), position dependent through `tag`; the block's fingerprint is the sum over its threads (block_sum_u64).
// Two independent 32-bit multiply-rotate mixes per word packed into one 64-bit sum (a 64-bit multiply costs four VALU
// multiplies here, and the first version of this check spent 5 us hashing 20 KB): change detection, not cryptography.
__device__ __forceinline__ unsigned long long fp_word(unsigned w, unsigned pos) {
    unsigned a = (w ^ (pos * 0x9E3779B1u)) * 0x85EBCA6Bu;
    a ^= a >> 15;
    a *= 0xC2B2AE35u;
    unsigned b = (w + pos) * 0xCC9E2D51u;
    b = (b << 13) | (b >> 19);
    b = b * 0x1B873593u + (pos ^ 0x27D4EB2Fu);
    return ((unsigned long long)(b ^ (b >> 16)) << 32) | (unsigned long long)(a ^ (a >> 13));
}
// (tid of nthreads: the threads that share the range -- the whole block by default; the sum over them is what counts)
template <bool PRED_TAIL = false>
__device__ __forceinline__ unsigned long long fp_range_n(const void *p, int64_t bytes, unsigned tag, int tid, int nthreads) {
    unsigned long long h = 0ull;
    if (p == nullptr) return h;
    if ((((uintptr_t)p | (uintptr_t)bytes) & 3) == 0) {
        // (four loads in flight per thread: one load per trip made this pass latency bound -- 11 us for 26 KB)
        const unsigned *w = (const unsigned *)p;
        const int n = (int)(bytes >> 2), step = nthreads;
        // PRED_TAIL: the ragged end predicated instead of a one-load-per-trip tail loop (1568 words over 512 threads are three
        // dependent round trips in the tail for all but 32 threads).  Same sum.  Measured on the in-launch table check at
        // B = 4096, same box: 8-channel kernel 20.7 -> 20.25 us per call with it, the (2,2) small-batch kernel 9.9 -> 10.45 us
        // (its work-groups have every load of their own in flight in the first microseconds; a fingerprint that spreads its
        // requests out disturbs them less) -- so the caller chooses.
        if constexpr (PRED_TAIL) {
        for (int e = tid; e < n; e += 4 * step) {
            const bool p1 = e + step < n, p2 = e + 2 * step < n, p3 = e + 3 * step < n;
            const unsigned w0 = w[e], w1 = p1 ? w[e + step] : 0u, w2 = p2 ? w[e + 2 * step] : 0u, w3 = p3 ? w[e + 3 * step] : 0u;
            h += fp_word(w0, (unsigned)e * 8u + tag);
            if (p1) h += fp_word(w1, (unsigned)(e + step) * 8u + tag);
            if (p2) h += fp_word(w2, (unsigned)(e + 2 * step) * 8u + tag);
            if (p3) h += fp_word(w3, (unsigned)(e + 3 * step) * 8u + tag);
        }
        } else {
        int e = tid;
        for (; e + 3 * step < n; e += 4 * step) {
            const unsigned w0 = w[e], w1 = w[e + step], w2 = w[e + 2 * step], w3 = w[e + 3 * step];
            h += fp_word(w0, (unsigned)e * 8u + tag) + fp_word(w1, (unsigned)(e + step) * 8u + tag) +
                 fp_word(w2, (unsigned)(e + 2 * step) * 8u + tag) + fp_word(w3, (unsigned)(e + 3 * step) * 8u + tag);
        }
        for (; e < n; e += step) h += fp_word(w[e], (unsigned)e * 8u + tag);
        }
    } else {
        const unsigned char *b = (const unsigned char *)p;
        for (int64_t e = tid; e < bytes; e += nthreads) h += fp_word(b[e], (unsigned)e * 8u + tag + 4u);
    }
    return h;
}
__device__ __forceinline__ unsigned long long fp_range(const void *p, int64_t bytes, unsigned tag) {
    return fp_range_n<false>(p, bytes, tag, (int)threadIdx.x, (int)blockDim.x);
}
// sum of a 64-bit value over the block, returned to every thread (red: 17 words of LDS; blockDim a multiple of 64)
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long h, unsigned long long *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += (unsigned long long)__shfl_xor((long long)h, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
        for (int w = 0; w < nw; ++w) t += red[w];
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}

}  // namespace dpk
