// LDS-DMA layout check (gfx950): two global_load_lds_dwordx4 per 16-feature K-step of a 32-row tile, four consecutive
// lanes fetching 64 contiguous bytes of a row, pieces XOR-swizzled on the source side; read back in the MFMA B-operand
// roles (lane = (row s, K half h)) and compared on the host.  Also tries several LDS base offsets (> 64 KiB included).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_layout.hip -o tools/ubench/dma_layout.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char *gcchar_p;
typedef __attribute__((address_space(3))) char lchar;
__device__ __forceinline__ void glds16(unsigned voff, gcchar_p sbase_in, unsigned lds_dst_in) {
    const uint64_t sb = (uint64_t)(uintptr_t)sbase_in;
    const uint64_t sbase = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32)) << 32) |
                           (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_in);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(512) void k(const float *x, int D, int base, float *out) {
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lchar *my = smem + base + wave * 2048;
    const int f0 = wave * 16;
    for (int i = 0; i < 2; ++i) {
        const int P = i * 64 + lane, row = P >> 2, c = (P & 3) ^ ((row >> 2) & 3);
        glds16((unsigned)(row * D + c * 4 + f0) * 4u, (gcchar_p)x, (unsigned)(uintptr_t)my + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int s = lane & 31, h = lane >> 5, sw = (s >> 2) & 3;
    typedef __attribute__((address_space(3))) const f4 lf4;
    const f4 x0 = *(lf4 *)(my + s * 64 + (((h * 2) ^ sw) << 4));
    const f4 x1 = *(lf4 *)(my + s * 64 + (((h * 2 + 1) ^ sw) << 4));
    float *o = out + (wave * 64 + lane) * 8;
    for (int e = 0; e < 4; ++e) { o[e] = x0[e]; o[4 + e] = x1[e]; }
}
// second kernel: the small-batch kernel's request pattern -- per K-step two DMAs then four plain 16-byte loads of a table,
// 7 K-steps per wave, reads ordered only by the compiler's wait for the K-step's last table load (opaque zero)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k2(const float *x, int D, const u4 *tab, float *out, unsigned *sink, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lchar *my = smem + wave * 14336;
    const int k0 = wave * 6;
    u4 fr[7][4];
    unsigned voff[2];
    for (int i = 0; i < 2; ++i) {
        const int P = i * 64 + lane, row = P >> 2, c = (P & 3) ^ ((row >> 2) & 3);
        voff[i] = (unsigned)(row * D + c * 4) * 4u;
    }
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        const int f0 = (k0 + kk) * 16;
        const unsigned dst = (unsigned)(uintptr_t)my + kk * 2048;
        glds16(voff[0] + f0 * 4u, (gcchar_p)x, dst);
        glds16(voff[1] + f0 * 4u, (gcchar_p)x, dst + 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) fr[kk][t] = tab[((k0 + kk) * 4 + t) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int s = lane & 31, h = lane >> 5, sw = (s >> 2) & 3;
    typedef __attribute__((address_space(3))) const f4 lf4;
    unsigned acc = 0;
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        unsigned zero;
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(fr[kk][3][3]) : "memory");
        const lchar *xb = my + kk * 2048;
        const f4 x0 = *(lf4 *)(xb + s * 64 + (((h * 2) ^ sw) << 4) + zero);
        const f4 x1 = *(lf4 *)(xb + s * 64 + (((h * 2 + 1) ^ sw) << 4) + zero);
        float *o = out + ((wave * 7 + kk) * 64 + lane) * 8;
        for (int e = 0; e < 4; ++e) { o[e] = x0[e]; o[4 + e] = x1[e]; }
        for (int t = 0; t < 4; ++t) acc += fr[kk][t][0] + fr[kk][t][1] + fr[kk][t][2] + fr[kk][t][3];
    }
    sink[threadIdx.x] = acc;
}
int main2() {
    const int D = 784;
    std::vector<float> hx(32 * D);
    for (int r = 0; r < 32; ++r) for (int f = 0; f < D; ++f) hx[r * D + f] = r * 1000.f + f;
    float *x, *out; unsigned *sink; u4 *tab;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&out, 8 * 7 * 64 * 8 * 4); hipMalloc(&sink, 2048); hipMalloc(&tab, 64 * 4 * 64 * 16);
    hipMemset(tab, 1, 64 * 4 * 64 * 16);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, 8 * 7 * 64 * 8 * 4);
        hipLaunchKernelGGL(k2, dim3(1), dim3(512), 8 * 14336, 0, x, D, tab, out, sink, mode);
        std::vector<float> ho(8 * 7 * 64 * 8);
        hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 8; ++w) for (int kk = 0; kk < 7; ++kk) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
            const int s = l & 31, h = l >> 5;
            const float want = s * 1000.f + ((w * 6 + kk) * 16 + h * 8 + e);
            const float got = ho[((w * 7 + kk) * 64 + l) * 8 + e];
            if (got != want) { if (bad < 6) printf("  mode %d wave %d kk %d lane %d e %d: got %.0f want %.0f\n", mode, w, kk, l, e, got, want); ++bad; }
        }
        printf("k2 mode %d: %d mismatches\n", mode, bad);
    }
    return 0;
}
int main() {
    main2();
    const int D = 784;
    std::vector<float> hx(32 * D);
    for (int r = 0; r < 32; ++r) for (int f = 0; f < D; ++f) hx[r * D + f] = r * 1000.f + f;
    float *x, *out; hipMalloc(&x, hx.size() * 4); hipMalloc(&out, 512 * 8 * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int base : {0, 14336, 65536, 100352, 131072}) {
        hipMemset(out, 0, 512 * 8 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), base + 8 * 2048, 0, x, D, base, out);
        std::vector<float> ho(512 * 8);
        hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 8; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
            const int s = l & 31, h = l >> 5;
            const float want = s * 1000.f + (w * 16 + h * 8 + e);
            if (ho[(w * 64 + l) * 8 + e] != want) { if (bad < 4) printf("  base %d wave %d lane %d e %d: got %.0f want %.0f\n", base, w, l, e, ho[(w * 64 + l) * 8 + e], want); ++bad; }
        }
        printf("base %6d: %d mismatches of 4096\n", base, bad);
    }
    return 0;
}
