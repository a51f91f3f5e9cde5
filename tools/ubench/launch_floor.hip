// Launch-floor microbenchmark: what rocprofv3 --kernel-trace reports for kernels that do (almost) nothing, at the
// small-batch kernel's geometry (128 work-groups of 512 threads, 70 KB of dynamic LDS).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(float *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1.f; }
__global__ __launch_bounds__(512) void k_lds(float *p) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = 1.f;
    __syncthreads();
    if (p == nullptr && sm[(threadIdx.x + 1) & 511] == 3.f) p[0] = 1.f;
}
__global__ __launch_bounds__(512) void k_load1(const float4 *x, float *p) {   // one L2-resident 16-byte load per lane, waited for
    const float4 v = x[threadIdx.x];
    if (v.x == 12345.f) p[0] = v.y;
}
__global__ __launch_bounds__(512) void k_load32(const float4 *x, float *p) {  // 32 loads per lane in flight, then one wait
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += x[(blockIdx.x * 32 + i) * 512 + threadIdx.x].x;
    if (acc == 12345.f) p[0] = acc;
}
__global__ __launch_bounds__(512, 2) void k_regs(const float4 *x, float *p) {  // forces a 256-VGPR allocation
    float v[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) v[i] = x[i].x * threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 200; ++i) acc = fmaf(acc, v[i], v[199 - i]);
    if (acc == 12345.f) p[0] = acc;
}
int main() {
    float4 *x; float *p;
    hipMalloc(&x, 128 * 32 * 512 * 16); hipMemset(x, 0, 128 * 32 * 512 * 16); hipMalloc(&p, 16);
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int it = 0; it < 50; ++it) {
        hipLaunchKernelGGL(k_empty, dim3(128), dim3(512), 0, 0, p);
        hipLaunchKernelGGL(k_lds, dim3(128), dim3(512), 70 * 1024, 0, p);
        hipLaunchKernelGGL(k_load1, dim3(128), dim3(512), 0, 0, x, p);
        hipLaunchKernelGGL(k_load32, dim3(128), dim3(512), 0, 0, x, p);
        hipLaunchKernelGGL(k_regs, dim3(128), dim3(512), 0, 0, x, p);
        hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, 0, p);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
