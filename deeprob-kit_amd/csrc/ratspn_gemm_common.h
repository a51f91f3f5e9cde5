// Shared pieces of the MFMA leaf-layer kernels (ratspn_gemm.hip: fused depth-2 model; ratspn_leaf_gemm.hip: the leaf
// layer alone): types, the LDS-DMA instruction, the f16 split of the x operand and the loader-wave loop of the ring.
#pragma once
#include "common.h"
#include <math.h>

namespace dpk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));
typedef float gf32x2 __attribute__((ext_vector_type(2)));
typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef float gf32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmWaves = 4;
constexpr int kGemmTile = 32 * kGemmWaves;           // samples per work-group tile
constexpr int kGemmStages = 3;
#ifndef DPK_X_NT
#define DPK_X_NT 0
#endif
constexpr bool kGemmXNonTemporal = DPK_X_NT != 0;   // x pieces of the ring: non-temporal LDS-DMA
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(3))) unsigned lunsigned;
typedef __attribute__((address_space(3))) char lchar;
typedef const __attribute__((address_space(1))) char *gcchar_p;
typedef const __attribute__((address_space(1))) void *gvoid_p;
// K-steps of 16 features per staged chunk: 64-feature chunks (256-byte row segments) while the mean table of a
// chunk fits beside them, 32-feature chunks for wide column sets
__host__ __device__ constexpr int gemm_ks(int NT) { return NT <= 2 ? 4 : 2; }
constexpr float kGemmStepBound = 1.0e6f;             // a K-step whose 8 squares sum above this is examined
constexpr float kGemmAbsBound = 1.0e3f;              // |x| above this: exact evaluation of the wave

__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

__device__ __forceinline__ void gemm_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One LDS-DMA instruction: LDS[lds_dst + lane*16 .. +15] <- global[sbase + voff .. +15].  Issued as inline asm so that
// hipcc neither counts it (the ring below is ordered by hand-counted vmcnt + s_barrier) nor drains it with a
// vmcnt(0) in front of an unrelated load; M0 (the DMA's LDS base) is compiler-reserved, hence saved and restored.
// The leading s_nop covers the SALU-write -> VMEM-read hazard of a freshly computed base (cdna_hip_programming 5.7).
// NT: the non-temporal form for data that ONE compute unit reads once (the x stream): measured on this ring, see DESIGN.
template <bool NT = false>
__device__ __forceinline__ void glds16(unsigned voff, gcchar_p sbase_in, unsigned lds_dst_in) {
    // (readfirstlane: a no-op for values hipcc already holds in SGPRs, a guarantee where it has moved them to VGPRs)
    const uint64_t sb = (uint64_t)(uintptr_t)sbase_in;
    const uint64_t sbase = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32)) << 32) |
                           (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_in);
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}

// x (8 values of one sample) -> f16 halves xh + xl = x to 2^-22: xh = rn16(x), xl = rn16(x - xh).
// Plain vector conversions: on gfx950 hipcc selects v_cvt_pk_f16_f32 / v_cvt_f32_f16 (+ SDWA) / v_pk_add_f32 for them,
// the sequence this function used to spell as inline asm -- but an asm statement's register writes are invisible to the
// compiler's hazard recognizer: with a conversion scheduled two instructions in front of the MFMA that reads its result
// (and an SDWA result consumed by the next VALU) the small-batch kernel computed wrong products for three of every eight
// features, in K-steps whose schedule happened to place them so (found with a one-hot feature probe, round 3).
__device__ __forceinline__ void split8(const float (&v)[8], half8 &xh, half8 &xl) {
#ifdef DPK_SPLIT8_ASM   // (measurement: the round-2 inline-asm form, for A/B runs of the ring kernel)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hp, lp;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned h2, l2;
        float b0, b1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h2) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
        asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(b0) : "v"(h2));
        asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(b1) : "v"(h2));
        const float d0 = v[2 * p] - b0, d1 = v[2 * p + 1] - b1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l2) : "v"(d0), "v"(d1));
        hp[p] = h2;
        lp[p] = l2;
    }
    xh = __builtin_bit_cast(half8, hp);
    xl = __builtin_bit_cast(half8, lp);
    return;
#endif
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const gf32x2 a = {v[2 * p], v[2 * p + 1]};
        const half2 h = __builtin_convertvector(a, half2);
        const gf32x2 b = __builtin_convertvector(h, gf32x2);
        const half2 l = __builtin_convertvector(a - b, half2);
        xh[2 * p] = h[0];
        xh[2 * p + 1] = h[1];
        xl[2 * p] = l[0];
        xl[2 * p + 1] = l[1];
    }
}


// The loader waves' side of the LDS ring (see ratspn_gemm.hip "Mapping"): wave `wave` (0..3) copies, for every chunk of
// every tile of this work-group, the KC = 16*KS features of its 32 rows of x (XOR-swizzled 16-byte pieces) and PB
// 1-KiB pieces of the chunk's table, starting at piece tab_kb0, into stage (chunk % 3); one s_barrier per chunk with
// the compute waves.  Tiles: tile0, tile0 + tstride, ... < ntiles.  The caller has already passed the work-group's
// opening __syncthreads() count into account: this function executes exactly ONE __syncthreads() (after its first two
// chunks are issued) and then one s_barrier per chunk.
// PB2 > 0: a second table with the same chunk geometry (the negated-constant table of the marginalised-evidence variant)
// is staged behind the first one, PB2 pieces per wave starting at piece wave * PB2.
template <int KS, int PB, int PB2 = 0>
__device__ __forceinline__ void gemm_loader_run(const float *x, int64_t B, int D, int NCH, int ntiles, int tile0,
                                                int tstride, gcchar_p table, int chunk_table_bytes, int tab_kb0,
                                                unsigned smem_base, int stage_bytes, int wave, int lane,
                                                gcchar_p table2 = nullptr) {
    constexpr int KC = 16 * KS, W = 4 * KS, ROWB = KC * 4, RPI = 64 / W;
    constexpr int SWS = (W == 16) ? 0 : (W == 8 ? 1 : 2);
    constexpr int XB = kGemmTile * ROWB;
    constexpr int PX = 32 / RPI;
    constexpr int P = PX + PB + PB2;
    static_assert(P <= 63, "vmcnt field");
    unsigned voff[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int rl = wave * 32 + j * RPI + lane / W;
        const int gp = (lane & (W - 1)) ^ ((rl >> SWS) & (W - 1));
        voff[j] = (unsigned)(rl * D + gp * 4) * 4u;
    }
    const unsigned toff = (unsigned)(tab_kb0 * 1024 + lane * 16);
    int ptile = tile0, pc = 0, pstage = 0;   // next chunk to stage
    auto issue_next = [&]() {
        const int64_t b0 = (int64_t)ptile * kGemmTile;
        const gcchar_p xt = (gcchar_p)x + (b0 * D + pc * KC) * 4;
        const gcchar_p tsrc = table + (int64_t)pc * chunk_table_bytes;
        const unsigned st = smem_base + pstage * stage_bytes;
        const bool full = (b0 + kGemmTile <= B) && ((pc + 1) * KC <= D);
        if (full) {
#pragma unroll
            for (int j = 0; j < PX; ++j) glds16<kGemmXNonTemporal>(voff[j], xt, st + (wave * 32 + j * RPI) * ROWB);
        } else {   // ragged tile / last chunk: clamp to rows and pieces that exist (clamped slots are never consumed)
            const int nvalid = (int)min((int64_t)kGemmTile, B - b0);
            const int vp = min(W, (D - pc * KC) >> 2);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int rl = wave * 32 + j * RPI + lane / W;
                const int gp = min((lane & (W - 1)) ^ ((rl >> SWS) & (W - 1)), vp - 1);
                glds16<kGemmXNonTemporal>((unsigned)(min(rl, nvalid - 1) * D + gp * 4) * 4u, xt, st + (wave * 32 + j * RPI) * ROWB);
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) glds16(toff + j * 1024, tsrc, st + XB + tab_kb0 * 1024 + j * 1024);
        if constexpr (PB2 > 0) {
            const gcchar_p tsrc2 = table2 + (int64_t)pc * chunk_table_bytes;
#pragma unroll
            for (int j = 0; j < PB2; ++j)
                glds16((unsigned)((wave * PB2 + j) * 1024 + lane * 16), tsrc2, st + XB + chunk_table_bytes + (wave * PB2 + j) * 1024);
        }
        pstage = (pstage + 1 == kGemmStages) ? 0 : pstage + 1;
        if (++pc == NCH) {
            pc = 0;
            ptile += tstride;
        }
    };
    static_assert(kGemmStages == 3, "the counted waits leave exactly one chunk in flight");
#pragma unroll
    for (int g = 0; g < kGemmStages - 1; ++g)
        if (ptile < ntiles) issue_next();
    __syncthreads();   // (the compute waves fill their constants meanwhile; hipcc does not count the asm DMAs)
    for (int tile = tile0; tile < ntiles; tile += tstride) {
        for (int c = 0; c < NCH; ++c) {
            // chunk (tile, c) has landed once at most one later chunk is still in flight (none exists at the very end)
            if (c + 1 < NCH || tile + tstride < ntiles) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            gemm_lds_barrier();   // this chunk is in LDS for everyone; everyone is done reading the previous one
            if (ptile < ntiles) issue_next();
        }
    }
}

}  // namespace dpk
