// Microbenchmark (measurement only): the inner block loop of the expanded leaf pipeline in isolation.
// Per block: 3 broadcast ds_read_b128 (record), 4 ds_read_b64 (x rows at lane_base + offset), 4 v_add, NPK packed FMAs.
// DEPTH = software-pipeline lookahead in blocks.  16 waves per CU (two 512-thread work-groups), like the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 65, kRowB = 520, kRecB = 80, kNRec = 24;

template <int DEPTH, int NPK, bool XREAD, bool RECREAD>
__global__ __launch_bounds__(512, 4) void k(float *out, int nblk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRows * kRowB / 4; i += 512) ((float *)smem)[i] = 1e-3f * (float)(i & 255);
    char *tab = smem + (kRows * kRowB + 15) / 16 * 16 + wave * (kNRec * kRecB);   // 16-byte aligned records
    for (int i = lane; i < kNRec * kRecB / 4; i += 64) {
        const int rec = i / (kRecB / 4), dw = i % (kRecB / 4);
        ((int *)tab)[i] = dw < 4 ? ((rec * 4 + dw) % 64) * kRowB : __float_as_int(0.5f + 0.01f * dw);
    }
    __syncthreads();
    const char *lane_base = smem + lane * 8;
    f2 P0 = {0, 0}, P1 = {0, 0}, Q = {0, 0};
    for (int it = 0; it < iters; ++it) {
        // like the real pipeline: while block b is consumed, x rows + means of block b+1 and the row offsets of
        // block b+2 are in flight (DEPTH = 1); DEPTH = 2 shifts everything one block further ahead
        constexpr int R = DEPTH + 1;          // ring size of x / mu
        i4 off_nx;                            // offsets of block b + DEPTH (already requested)
        f8 mu[R];
        f2 x[R][4];
        off_nx = *(const i4 *)(tab);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            mu[d] = *(const f8 *)(tab + d * kRecB + 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) x[d][u] = XREAD ? *(const f2 *)(lane_base + off_nx[u]) : (f2){1.f * u, 2.f};
            off_nx = *(const i4 *)(tab + (d + 1) * kRecB);
        }
        int rofs = DEPTH * kRecB;
        for (int b0 = 0; b0 < nblk; b0 += R) {
#pragma unroll
            for (int ph = 0; ph < R; ++ph) {
                const int sn = (ph + DEPTH) % R;
                const char *rec = tab + rofs;
                rofs = (rofs + kRecB >= kNRec * kRecB) ? 0 : rofs + kRecB;
                // x rows of block b + DEPTH (offsets arrived one step ago), then the next offsets and means
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int o = off_nx[u];
                    x[sn][u] = XREAD ? *(const f2 *)(lane_base + o) : (f2){__int_as_float(o), 2.f};
                }
                if (RECREAD) {
                    off_nx = *(const i4 *)(tab + ((rofs + kRecB >= kNRec * kRecB) ? 0 : rofs + kRecB));
                    mu[sn] = *(const f8 *)(rec + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f2 xv = x[ph][u];
                    const f2 mp = {mu[ph][2 * u], mu[ph][2 * u + 1]};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P0) : "v"(xv), "v"(mp));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(P1) : "v"(xv), "v"(mp));
                    if (NPK == 12) Q = __builtin_elementwise_fma(xv, xv, Q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const f2 r = P0 + P1 + Q;
    out[blockIdx.x * 512 + tid] = r.x + r.y;
}

template <int DEPTH, int NPK, bool XREAD, bool RECREAD> void run(const char *name, float *out) {
    const int nblk = 240, iters = 40, grid = 512;
    const size_t lds = (kRows * kRowB + 15) / 16 * 16 + 8 * kNRec * kRecB;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<DEPTH, NPK, XREAD, RECREAD><<<grid, 512, lds>>>(out, nblk, 2);
    hipEventRecord(a);
    k<DEPTH, NPK, XREAD, RECREAD><<<grid, 512, lds>>>(out, nblk, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // one wave executes nblk*iters blocks; 4 waves share a SIMD
    printf("%-46s %7.3f ms   %.1f ns per block per wave (wall), %.1f ns per block-round per SIMD\n", name, ms,
           ms * 1e6 / (nblk * iters), ms * 1e6 / (nblk * iters));
}

// variant: the record arrives as ONE ds_read_b32 (lane i holds dword i), offsets and means move to SGPRs by v_readlane
template <int NPK>
__global__ __launch_bounds__(512, 4) void k_sgpr(float *out, int nblk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRows * kRowB / 4; i += 512) ((float *)smem)[i] = 1e-3f * (float)(i & 255);
    constexpr int kRec2 = 48;
    char *tab = smem + (kRows * kRowB + 15) / 16 * 16 + wave * (kNRec * kRecB);
    for (int i = lane; i < kNRec * kRec2 / 4; i += 64) {
        const int rec = i / (kRec2 / 4), dw = i % (kRec2 / 4);
        ((int *)tab)[i] = dw < 4 ? ((rec * 4 + dw) % 64) * kRowB : __float_as_int(0.5f + 0.01f * dw);
    }
    __syncthreads();
    const char *lane_base = smem + lane * 8;
    const char *rp = tab + (lane < 12 ? lane : 11) * 4;
    f2 P0 = {0, 0}, P1 = {0, 0}, Q = {0, 0};
    for (int it = 0; it < iters; ++it) {
        int rec_cur = *(const int *)rp, rec_nx = *(const int *)(rp + kRec2);
        f2 x[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[0][u] = *(const f2 *)(lane_base + __builtin_amdgcn_readlane(rec_cur, u));
        int rofs = 2 * kRec2;
        for (int b0 = 0; b0 < nblk; b0 += 2) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int rec_n2 = *(const int *)(rp + rofs);
                rofs = (rofs + kRec2 >= kNRec * kRec2) ? 0 : rofs + kRec2;
#pragma unroll
                for (int u = 0; u < 4; ++u) x[1 - ph][u] = *(const f2 *)(lane_base + __builtin_amdgcn_readlane(rec_nx, u));
                float m[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) m[i] = __int_as_float(__builtin_amdgcn_readlane(rec_cur, 4 + i));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f2 xv = x[ph][u];
                    P0 = __builtin_elementwise_fma(xv, (f2){m[2 * u], m[2 * u]}, P0);
                    P1 = __builtin_elementwise_fma(xv, (f2){m[2 * u + 1], m[2 * u + 1]}, P1);
                    if (NPK == 12) Q = __builtin_elementwise_fma(xv, xv, Q);
                }
                __builtin_amdgcn_sched_barrier(0);
                rec_cur = rec_nx;
                rec_nx = rec_n2;
            }
        }
    }
    const f2 r = P0 + P1 + Q;
    out[blockIdx.x * 512 + tid] = r.x + r.y;
}

template <int NPK> void run_sgpr(const char *name, float *out) {
    const int nblk = 240, iters = 40, grid = 512;
    const size_t lds = (kRows * kRowB + 15) / 16 * 16 + 8 * kNRec * kRecB;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_sgpr<NPK><<<grid, 512, lds>>>(out, nblk, 2);
    hipEventRecord(a);
    k_sgpr<NPK><<<grid, 512, lds>>>(out, nblk, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-46s %7.3f ms   %.1f ns per block per wave (wall)\n", name, ms, ms * 1e6 / (nblk * iters));
}

// variant: compact 48-byte record = 8 means (2 x ds_read_b128) + 4 row offsets packed as u16 (1 x ds_read_b64),
// the unpack folded into the address add by SDWA
__device__ __forceinline__ int add_w0(int base, int pk) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(pk));
    return r;
}
__device__ __forceinline__ int add_w1(int base, int pk) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(pk));
    return r;
}
typedef int i2 __attribute__((ext_vector_type(2)));
template <int DEPTH>
__global__ __launch_bounds__(512, 4) void k_compact(float *out, int nblk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRows * kRowB / 4; i += 512) ((float *)smem)[i] = 1e-3f * (float)(i & 255);
    constexpr int kRec2 = 48;
    char *tab = smem + (kRows * kRowB + 15) / 16 * 16 + wave * (kNRec * kRecB);
    for (int i = lane; i < kNRec * kRec2 / 4; i += 64) {
        const int rec = i / (kRec2 / 4), dw = i % (kRec2 / 4);
        int v = __float_as_int(0.5f + 0.01f * dw);
        if (dw == 8) v = (((rec * 4 + 0) % 64) * kRowB) | ((((rec * 4 + 1) % 64) * kRowB) << 16);
        if (dw == 9) v = (((rec * 4 + 2) % 64) * kRowB) | ((((rec * 4 + 3) % 64) * kRowB) << 16);
        ((int *)tab)[i] = v;
    }
    __syncthreads();
    const int lane_off = (int)(size_t)0 + lane * 8;   // byte offset of the lane inside a row
    f2 P0 = {0, 0}, P1 = {0, 0};
    for (int it = 0; it < iters; ++it) {
        constexpr int R = DEPTH + 1;
        i2 off_nx = *(const i2 *)(tab + 32);
        f8 mu[R];
        f2 x[R][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            mu[d] = *(const f8 *)(tab + d * kRec2);
            x[d][0] = *(const f2 *)(smem + add_w0(lane_off, off_nx[0]));
            x[d][1] = *(const f2 *)(smem + add_w1(lane_off, off_nx[0]));
            x[d][2] = *(const f2 *)(smem + add_w0(lane_off, off_nx[1]));
            x[d][3] = *(const f2 *)(smem + add_w1(lane_off, off_nx[1]));
            off_nx = *(const i2 *)(tab + (d + 1) * kRec2 + 32);
        }
        int rofs = DEPTH * kRec2;
        for (int b0 = 0; b0 < nblk; b0 += R) {
#pragma unroll
            for (int ph = 0; ph < R; ++ph) {
                const int sn = (ph + DEPTH) % R;
                const char *rec = tab + rofs;
                rofs = (rofs + kRec2 >= kNRec * kRec2) ? 0 : rofs + kRec2;
                x[sn][0] = *(const f2 *)(smem + add_w0(lane_off, off_nx[0]));
                x[sn][1] = *(const f2 *)(smem + add_w1(lane_off, off_nx[0]));
                x[sn][2] = *(const f2 *)(smem + add_w0(lane_off, off_nx[1]));
                x[sn][3] = *(const f2 *)(smem + add_w1(lane_off, off_nx[1]));
                off_nx = *(const i2 *)(tab + ((rofs + kRec2 >= kNRec * kRec2) ? 0 : rofs + kRec2) + 32);
                mu[sn] = *(const f8 *)(rec);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f2 xv = x[ph][u];
                    const f2 mp = {mu[ph][2 * u], mu[ph][2 * u + 1]};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(P0) : "v"(xv), "v"(mp));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(P1) : "v"(xv), "v"(mp));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const f2 r = P0 + P1;
    out[blockIdx.x * 512 + tid] = r.x + r.y;
}

template <int DEPTH> void run_compact(const char *name, float *out) {
    const int nblk = 240, iters = 40, grid = 512;
    const size_t lds = (kRows * kRowB + 15) / 16 * 16 + 8 * kNRec * kRecB;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_compact<DEPTH><<<grid, 512, lds>>>(out, nblk, 2);
    hipEventRecord(a);
    k_compact<DEPTH><<<grid, 512, lds>>>(out, nblk, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-46s %7.3f ms   %.1f ns per block per wave (wall)\n", name, ms, ms * 1e6 / (nblk * iters));
}

int main() {
    float *out; hipMalloc(&out, 512 * 512 * 4);
    run<1, 12, true, true>("depth 1, 12 pk (Q in loop)", out);
    run<1, 8, true, true>("depth 1, 8 pk", out);
    run<2, 8, true, true>("depth 2, 8 pk", out);
    run<3, 8, true, true>("depth 3, 8 pk", out);
    run<1, 8, false, true>("depth 1, 8 pk, no x reads", out);
    run<1, 8, true, false>("depth 1, 8 pk, no record reads", out);
    run<1, 8, false, false>("depth 1, 8 pk, no LDS reads at all", out);
    run_compact<1>("compact record (b64 offsets + sdwa), depth 1", out);
    run_compact<2>("compact record (b64 offsets + sdwa), depth 2", out);
    run_sgpr<8>("b32 record + 12 readlanes, 8 pk", out);
    run_sgpr<12>("b32 record + 12 readlanes, 12 pk", out);
    return 0;
}
