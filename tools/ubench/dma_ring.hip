// How fast can ONE work-group per CU stream x through an LDS-DMA ring, as a function of the bytes it keeps in flight?
// The headline kernel's ring (ratspn_gemm.hip) has 3 stages of 48 KB (32 KB of x + 16 KB of mean table): two chunks =
// 64 KB of x in flight per CU, 5.0 TB/s.  This bench runs the same access pattern (128-row tiles, 64-feature chunks =
// 256-byte row segments at a 3136-byte stride, XOR-swizzled 16-byte pieces, one s_barrier per chunk with four consumer
// waves that read the chunk back from LDS) with x-only stages of 32 KB and NS = 3, 4, 5 stages (NS - 1 chunks in flight),
// and with 16 KB of table riding along as in the kernel (TAB = 1).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_ring.hip -o tools/ubench/dma_ring.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
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
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kTile = 128, kKC = 64, kRowB = kKC * 4, kXB = kTile * kRowB;   // 32 KB of x per chunk

template <int NS, int TAB>
__global__ __launch_bounds__(512) void ring(const float *x, const char *tab, int64_t B, int D, int ntiles, float *sink) {
    constexpr int PX = 8, PT = TAB ? 4 : 0, P = PX + PT, STAGE = kXB + (TAB ? 16384 : 0);
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lchar *smem = (lchar *)smem_generic;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCH = (D + kKC - 1) / kKC;
    int nmine = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++nmine;
    const int total = nmine * NCH;   // chunks of this work-group
    if (wave >= 4) {
        const int w = wave - 4;
        unsigned voff[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int rl = w * 32 + j * 4 + (lane >> 4);
            voff[j] = (unsigned)(rl * D + (((lane & 15) ^ (rl & 15)) << 2)) * 4u;
        }
        int ptile = blockIdx.x, pc = 0, pstage = 0, issued = 0;
        auto issue = [&]() {
            const gcchar_p xt = (gcchar_p)(x + ((int64_t)ptile * kTile * D + pc * kKC));
            const unsigned st = (unsigned)(uintptr_t)smem + pstage * STAGE;
            const bool full = (pc + 1) * kKC <= D;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                unsigned vo = voff[j];
                if (!full) {   // ragged last chunk: clamp the piece
                    const int rl = w * 32 + j * 4 + (lane >> 4);
                    const int gp = min((lane & 15) ^ (rl & 15), ((D - pc * kKC) >> 2) - 1);
                    vo = (unsigned)(rl * D + gp * 4) * 4u;
                }
                glds16(vo, xt, st + (w * 32 + j * 4) * kRowB);
            }
#pragma unroll
            for (int j = 0; j < PT; ++j)
                glds16((unsigned)((w * PT + j) * 1024 + lane * 16), (gcchar_p)(tab + (int64_t)pc * 16384), st + kXB + (w * PT + j) * 1024);
            pstage = (pstage + 1 == NS) ? 0 : pstage + 1;
            if (++pc == NCH) { pc = 0; ptile += gridDim.x; }
            ++issued;
        };
        for (int g = 0; g < NS - 1 && issued < total; ++g) issue();
        __syncthreads();
        for (int c = 0; c < total; ++c) {
            // chunk c has landed once at most (issued - c - 1) later chunks are in flight
            const int later = issued - c - 1;
            if (later >= NS - 2 && NS >= 3) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * P) : "memory");
            } else if (NS >= 4 && later == NS - 3) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS >= 4 ? NS - 3 : 0) * P) : "memory");
            } else if (NS >= 5 && later == NS - 4) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS >= 5 ? NS - 4 : 0) * P) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();
            if (issued < total) issue();
        }
        return;
    }
    // consumer waves: read the lane's two pieces of each of the chunk's 4 K-steps for their 32-row block
    typedef __attribute__((address_space(3))) const f4 lf4;
    const int s = lane & 31, h = lane >> 5, rl = wave * 32 + s, sw = rl & 15;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    int stage = 0;
    for (int c = 0; c < total; ++c) {
        lds_barrier();
        const lchar *st = smem + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int pcs = kk * 4 + h * 2;
            acc += *(lf4 *)(st + rl * kRowB + ((pcs ^ sw) << 4));
            acc += *(lf4 *)(st + rl * kRowB + (((pcs | 1) ^ sw) << 4));
        }
        stage = (stage + 1 == NS) ? 0 : stage + 1;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) sink[threadIdx.x] = acc[0];
}

template <int NS, int TAB>
static void run(float *const *xs, const char *tab, int64_t B, int D, float *sink, int grid) {
    constexpr int NX = 4;   // inputs used in turn: one 205 MB buffer would be served by the 256 MB Infinity Cache
    const int ntiles = (int)(B / kTile);
    const size_t lds = (size_t)NS * (kXB + (TAB ? 16384 : 0));
    auto kern = ring<NS, TAB>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, xs[i % NX], tab, B, D, ntiles, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, xs[i % NX], tab, B, D, ntiles, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / 12 < best ? ms / 12 : best;
    }
    const double bytes = (double)B * D * 4;
    printf("stages %d table %d grid %d lds %3zu KB: %7.2f us  %6.2f TB/s of x  (err %s)\n", NS, TAB, grid, lds / 1024, best * 1e3,
           bytes / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char **argv) {
    const int64_t B = 65536;
    const int D = 784;
    float *xs[4], *sink;
    char *tab;
    for (int i = 0; i < 4; ++i) {
        hipMalloc(&xs[i], (size_t)B * D * 4);
        hipMemset(xs[i], 0, (size_t)B * D * 4);
    }
    hipMalloc(&tab, 13 * 16384);
    hipMalloc(&sink, 4096);
    hipMemset(tab, 0, 13 * 16384);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int grid : {cus, 2 * cus}) {
        if (grid == 2 * cus) {   // two work-groups per CU need <= 80 KB each
            run<2, 0>(xs, tab, B, D, sink, grid);
            continue;
        }
        run<3, 1>(xs, tab, B, D, sink, grid);
        run<3, 0>(xs, tab, B, D, sink, grid);
        run<4, 0>(xs, tab, B, D, sink, grid);
        run<5, 0>(xs, tab, B, D, sink, grid);
    }
    return 0;
}
