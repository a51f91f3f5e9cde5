// Measurement: achievable HBM READ bandwidth of one MI355X with plain streaming kernels (16-byte loads, every
// work-group a contiguous slab, or the 256-byte row segments at a 3136-byte stride the RAT-SPN kernel's chunks read).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/read_bw.hip -o tools/ubench/read_bw.bin && tools/ubench/read_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_linear(const f4 *x, size_t n4, float *out) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
// rows of 784 floats; a work-group owns 128 rows and walks them in chunks of 64 columns (256-byte segments per row)
__global__ __launch_bounds__(256) void read_chunks(const float *x, int rows, float *out) {
    f4 acc = {0, 0, 0, 0};
    const int tiles = rows / 128;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const float *base = x + (size_t)t * 128 * 784;
        for (int c = 0; c < 12; ++c)            // 12 full chunks of 64 columns (the 13th is partial)
            for (int r = threadIdx.x / 16; r < 128; r += 16) acc += *(const f4 *)(base + (size_t)r * 784 + c * 64 + (threadIdx.x & 15) * 4);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

int main() {
    const size_t rows = 65536 * 4, n = rows * 784;   // 822 MB: beyond the 256 MB Infinity Cache
    float *x, *out;
    (void)hipMalloc(&x, n * 4);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(x, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, (const f4 *)x, n / 4, out);
        (void)hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, (const f4 *)x, n / 4, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("linear 16-byte loads, grid %5d: %.2f TB/s\n", grid, n * 4.0 * 5 / (ms * 1e-3) / 1e12);
    }
    for (int grid : {256, 512, 2048}) {
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(read_chunks, dim3(grid), dim3(256), 0, 0, x, (int)rows, out);
        (void)hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(read_chunks, dim3(grid), dim3(256), 0, 0, x, (int)rows, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("256-byte row segments (12 of 12.25 chunks), grid %5d: %.2f TB/s\n", grid,
               rows * 768.0 * 4 * 5 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
