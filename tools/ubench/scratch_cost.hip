// Microbenchmark: what does a kernel's scratch (private stack) allocation cost per dispatch on gfx950?
// Two kernels with the same trivial body, 256 work-groups of 512 threads: one keeps everything in registers, the other
// is forced to own a 224-byte-per-lane private array (the size hipcc reports for ring::ratspn_gemm_kernel).  Run under
// rocprofv3 --kernel-trace --stats and compare the average durations; also alternated (A B A B) to see whether the cost
// is paid when the scratch size changes between consecutive dispatches.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void plain_kernel(float *out, int n) {
    float v = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    if (v == -1.f) out[0] = v;
}

__global__ __launch_bounds__(512) void scratch_kernel(float *out, int n, int idx) {
    float v = (float)threadIdx.x;
    if (idx < 0) {                                 // never taken: the scratch is allocated for the dispatch, not touched
        volatile float priv[56];                   // 224 bytes per lane, dynamically indexed: lives in scratch
        for (int i = 0; i < 56; ++i) priv[i] = (float)(i + threadIdx.x);
        v = priv[idx & 55];
    }
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    if (v == -1.f) out[0] = v;
}

int main() {
    float *d;
    hipMalloc(&d, 4096);
    hipStream_t st;
    hipStreamCreate(&st);
    for (int rep = 0; rep < 200; ++rep) plain_kernel<<<256, 512, 0, st>>>(d, 64);
    hipStreamSynchronize(st);
    for (int rep = 0; rep < 200; ++rep) scratch_kernel<<<256, 512, 0, st>>>(d, 64, rep);
    hipStreamSynchronize(st);
    for (int rep = 0; rep < 200; ++rep) {
        plain_kernel<<<256, 512, 0, st>>>(d, 64);
        scratch_kernel<<<256, 512, 0, st>>>(d, 64, rep);
    }
    hipStreamSynchronize(st);
    // wall time per dispatch of back-to-back launches
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int rep = 0; rep < 1000; ++rep) {
            if (which) scratch_kernel<<<256, 512, 0, st>>>(d, 64, rep);
            else plain_kernel<<<256, 512, 0, st>>>(d, 64);
        }
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per back-to-back dispatch\n", which ? "scratch" : "plain", ms);
    }
    return 0;
}
