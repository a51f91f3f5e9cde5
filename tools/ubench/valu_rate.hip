// Microbenchmark (measurement only): issue rate of the f32 VALU forms the leaf kernel uses.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *par, int iters) {
    // par is uniform -> SGPR operands
    const float s0 = par[0], s1 = par[1], s2 = par[2], s3 = par[3];
    f2 a[8];
    float x = threadIdx.x * 1e-3f;
    float tabv = par[threadIdx.x & 15];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f2){x + i, x - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {  // scalar fma, vgpr operands
                a[i].x = __builtin_fmaf(a[i].x, a[i].y, a[i].x);
            } else if (MODE == 1) {  // scalar fma with SGPR operand
                a[i].x = __builtin_fmaf(a[i].x, s0, a[i].y);
            } else if (MODE == 2) {  // pk fma vgpr
                a[i] = __builtin_elementwise_fma(a[i], a[i], a[i]);
            } else if (MODE == 3) {  // pk fma with SGPR pair
                a[i] = __builtin_elementwise_fma(a[i], (f2){s0, s1}, a[i]);
            } else if (MODE == 4) {  // pk add with SGPR pair
                a[i] = a[i] - (f2){s2, s3};
            } else if (MODE == 5) {  // pk mul
                a[i] = a[i] * a[i];
            } else if (MODE == 6) {  // scalar sub sgpr + scalar fma (unit-scale pair), 2 instr
                float t = a[i].x - s0;
                a[i].y = __builtin_fmaf(t, t, a[i].y);
            } else if (MODE == 8) {  // dpp-broadcast sub (asm) + fma: the unit-scale pair with a VGPR table
                float t;
                asm volatile("v_sub_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "=v"(t) : "v"(tabv), "v"(a[i].x));
                a[i].y = __builtin_fmaf(t, t, a[i].y);
            } else if (MODE == 9) {  // dpp-broadcast v_fmac (general scale third op)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a[i].y) : "v"(tabv), "v"(a[i].x));
            } else if (MODE == 7) {  // pk sub + pk fma, 2 instr for 2 elements
                f2 t = (f2){a[i].x, a[i].x} - (f2){s0, s1};
                a[i] = __builtin_elementwise_fma(t, t, a[i]);
            }
            asm volatile("" : "+v"(a[i]));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, int instr_per_iter, float *out, float *par) {
    const int iters = 20000, blocks = 256 * 4;  // 4 blocks of 256 / CU = 4 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, par, 100);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, par, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD = 4 waves * iters * 8 * instr_per_iter ; cycles at 2.4 GHz nominal
    double winstr = 4.0 * iters * 8 * instr_per_iter;
    printf("%-36s %8.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / winstr);
}

int main() {
    float *out, *par;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMalloc(&par, 64);
    float h[16] = {1.0001f, 0.9999f, 1e-7f, -1e-7f, 0.5f, 1e-3f};
    hipMemcpy(par, h, 64, hipMemcpyHostToDevice);
    run<0>("v_fma_f32 vgpr", 1, out, par);
    run<1>("v_fma_f32 sgpr operand", 1, out, par);
    run<2>("v_pk_fma_f32 vgpr", 1, out, par);
    run<3>("v_pk_fma_f32 sgpr pair", 1, out, par);
    run<4>("v_pk_add_f32 sgpr pair", 1, out, par);
    run<5>("v_pk_mul_f32", 1, out, par);
    run<6>("sub+fma scalar (2 instr)", 2, out, par);
    run<7>("pk_sub+pk_fma (2 instr, 2 elem)", 2, out, par);
    run<8>("sub_dpp(row_newbcast)+fma (2 instr)", 2, out, par);
    run<9>("fmac_dpp(row_newbcast)", 1, out, par);
    return 0;
}
