// Measurement: what do 8-byte stores at a 16-byte stride cost against full 16-byte stores?  (a coupling layer
// that writes the pass-through and the transformed columns of a row at different times)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/partial_writes.hip -o gpurun_out/partial_writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void full16(float4 *o, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// half 0: floats {0, 2} of every float4 (the even columns), half 1: floats {1, 3}; half 2: the low 8 bytes, half 3: the high
__global__ void part8(float *o, size_t n4, int half) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (half < 2) {
            o[4 * i + half] = 1.f;
            o[4 * i + 2 + half] = 2.f;
        } else {
            *reinterpret_cast<float2 *>(o + 4 * i + 2 * (half - 2)) = make_float2(1.f, 2.f);
        }
    }
}
// both halves from one kernel, `lag` tiles apart: a block owns 200 KB tiles; it writes the low halves of tile t and the high
// halves of tile t - lag in the same pass (what a work-group of the layer would do, phase 1 of a tile / phase 2 of the one before)
__global__ void lagged(float *o, size_t tile4, int ntiles, int lag) {
    for (int t = blockIdx.x; t < ntiles + lag * (int)gridDim.x; t += gridDim.x) {
        for (size_t i = threadIdx.x; i < tile4; i += blockDim.x) {
            if (t < ntiles) *reinterpret_cast<float2 *>(o + 4 * ((size_t)t * tile4 + i)) = make_float2(1.f, 2.f);
            const int tp = t - lag * (int)gridDim.x;
            if (tp >= 0) *reinterpret_cast<float2 *>(o + 4 * ((size_t)tp * tile4 + i) + 2) = make_float2(3.f, 4.f);
        }
    }
}
int main() {
    const size_t B = 65536, D = 784, n4 = B * D / 4;
    float *buf[4];
    for (auto &b : buf) { hipMalloc(&b, n4 * 16); hipMemset(b, 0, n4 * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto fn) {
        for (int i = 0; i < 4; ++i) fn(buf[i & 3]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) fn(buf[i & 3]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  (%.2f TB/s of the 205 MB buffer)\n", name, ms / 20 * 1e3, n4 * 16 / (ms / 20 * 1e-3) / 1e12);
    };
    time("full 16-byte stores", [&](float *b) { full16<<<2048, 256>>>((float4 *)b, n4); });
    time("even floats only (4-byte stores)", [&](float *b) { part8<<<2048, 256>>>(b, n4, 0); });
    time("low 8 bytes of every 16", [&](float *b) { part8<<<2048, 256>>>(b, n4, 2); });
    time("low 8 then high 8 (two kernels)", [&](float *b) { part8<<<2048, 256>>>(b, n4, 2); part8<<<2048, 256>>>(b, n4, 3); });
    time("even then odd floats (two kernels)", [&](float *b) { part8<<<2048, 256>>>(b, n4, 0); part8<<<2048, 256>>>(b, n4, 1); });
    const size_t tile4 = 64 * D / 4;
    time("one kernel, halves 1 tile apart", [&](float *b) { lagged<<<256, 512>>>(b, tile4, (int)(B / 64), 1); });
    time("one kernel, halves 0 tiles apart", [&](float *b) { lagged<<<256, 512>>>(b, tile4, (int)(B / 64), 0); });
    return 0;
}
