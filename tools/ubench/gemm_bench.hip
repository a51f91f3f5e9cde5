// Microbenchmark (measurement only): the stride-addressed fp32-MFMA GEMM of the flows' training route on the six
// product shapes of one RealNVP1d(784, units 128) coupling layer at a given batch.  usage: gemm_bench.bin [B] [iters]
#include "../../deeprob-kit_amd/csrc/coupling_bwd.hip"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 200;
    const int D = 784, U = 128, Z = 2 * D;
    float *x, *H, *Zb, *W1, *W2, *dH, *gW1, *gW2, *gx, *vec;
    hipMalloc(&x, (size_t)B * D * 4); hipMalloc(&H, (size_t)B * U * 4); hipMalloc(&Zb, (size_t)B * Z * 4);
    hipMalloc(&W1, (size_t)U * D * 4); hipMalloc(&W2, (size_t)Z * U * 4); hipMalloc(&dH, (size_t)B * U * 4);
    hipMalloc(&gW1, (size_t)U * D * 4); hipMalloc(&gW2, (size_t)Z * U * 4); hipMalloc(&gx, (size_t)B * D * 4);
    hipMalloc(&vec, (size_t)Z * 4);
    hipMemset(x, 0, (size_t)B * D * 4); hipMemset(H, 0, (size_t)B * U * 4); hipMemset(Zb, 0, (size_t)B * Z * 4);
    hipMemset(W1, 0, (size_t)U * D * 4); hipMemset(W2, 0, (size_t)Z * U * 4); hipMemset(dH, 0, (size_t)B * U * 4);
    hipMemset(vec, 0, (size_t)Z * 4); hipMemset(gx, 0, (size_t)B * D * 4);
    GemmArgs gs[6];
    const char *names[6] = {"H  = x W1^T   ", "Z  = H W2^T   ", "dW2= dZ^T H   ", "dH = dZ W2    ", "dW1= dH^T x   ", "gx+= dH W1    "};
    GemmArgs g{};
    g.A = x; g.sam = D; g.sak = 1; g.kscale = vec; g.Bm = W1; g.sbk = 1; g.sbn = D; g.C = H; g.ldc = U; g.M = B; g.N = U; g.K = D; g.bias = vec; g.relu = 1; gs[0] = g;
    g = GemmArgs{}; g.A = H; g.sam = U; g.sak = 1; g.Bm = W2; g.sbk = 1; g.sbn = U; g.C = Zb; g.ldc = Z; g.M = B; g.N = Z; g.K = U; g.bias = vec; gs[1] = g;
    g = GemmArgs{}; g.A = Zb; g.sam = 1; g.sak = Z; g.Bm = H; g.sbk = U; g.sbn = 1; g.C = gW2; g.ldc = U; g.M = Z; g.N = U; g.K = B; gs[2] = g;
    g = GemmArgs{}; g.A = Zb; g.sam = Z; g.sak = 1; g.Bm = W2; g.sbk = U; g.sbn = 1; g.C = dH; g.ldc = U; g.M = B; g.N = U; g.K = Z; g.gate = H; g.ldg = U; gs[3] = g;
    g = GemmArgs{}; g.A = dH; g.sam = 1; g.sak = U; g.Bm = x; g.sbk = D; g.sbn = 1; g.nscale = vec; g.C = gW1; g.ldc = D; g.M = U; g.N = D; g.K = B; gs[4] = g;
    g = GemmArgs{}; g.A = dH; g.sam = U; g.sak = 1; g.Bm = W1; g.sbk = D; g.sbn = 1; g.nscale = vec; g.C = gx; g.ldc = D; g.M = B; g.N = D; g.K = U; g.accumulate = 1; gs[5] = g;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int s = 0; s < 6; ++s) {
        for (int i = 0; i < 20; ++i) launch_gemm(gs[s], 0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch_gemm(gs[s], 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * gs[s].M * gs[s].N * gs[s].K;
        printf("%s M=%5d N=%5d K=%5d  %8.2f us  %7.2f TFLOP/s\n", names[s], gs[s].M, gs[s].N, gs[s].K, us, fl / us * 1e-6);
    }
    return 0;
}
