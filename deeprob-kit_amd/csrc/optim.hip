// Adam update of every parameter tensor of a model in ONE launch, many work-groups.
//
// reference: the training loops call `optimizer.step()` of torch.optim.Adam (deeprob/torch/routines.py:164, :280; optimizer
// chosen by name at deeprob/torch/utils.py:32-49).  torch's own fused Adam hands each tensor to work-groups in chunks of
// 65 536 elements: the models of this path (50 k .. 1.5 M parameters in 3 .. 30 tensors) end up on 3 .. 30 work-groups that
// walk their chunk serially -- 30 us per step for a RAT-SPN whose whole forward + backward is 200 us (round-3 profile).
// Here a work-group owns 2048 elements; same update rule, fp32 arithmetic in the same order as torch's (non-amsgrad,
// decoupled weight decay off):
//     m = m + (g - m) (1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t lives on the device (fp32, like torch's capturable Adam) so that the launch can be captured in a HIP
// graph: every work-group reads it, the LAST work-group to finish increments it for the next launch.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace dpk {

constexpr int kAdamChunk = 2048;
constexpr int kAdamMaxTensors = 96;

struct AdamArgs {
    float *p[kAdamMaxTensors];
    const float *g[kAdamMaxTensors];
    float *m[kAdamMaxTensors], *v[kAdamMaxTensors];
    int first_block[kAdamMaxTensors + 1];   // prefix sums of the tensors' work-group counts
    int numel[kAdamMaxTensors];
    int n;
    float lr, b1, b2, eps, weight_decay;
    int maximize;
    float *step;          // [1] number of updates done so far
    unsigned *ticket;     // [1] work-groups of this launch that have finished (zero between launches)
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
    // which tensor: the prefix table is tiny and wave-uniform
    const int blk = (int)blockIdx.x;
    int t = 0;
    while (t + 1 < a.n && blk >= a.first_block[t + 1]) ++t;
    const int e0 = (blk - a.first_block[t]) * kAdamChunk;
    const int n = a.numel[t];
    float *p = a.p[t], *m = a.m[t], *v = a.v[t];
    const float *g = a.g[t];
    constexpr int K = kAdamChunk / 256;
    // all of the chunk's reads in flight before anything waits on the step count
    float gr[K], pr[K], mr[K], vr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = e0 + k * 256 + (int)threadIdx.x;
        const bool in = e < n;
        gr[k] = in ? g[e] : 0.f; pr[k] = in ? p[e] : 0.f; mr[k] = in ? m[e] : 0.f; vr[k] = in ? v[e] : 0.f;
    }
    const float tstep = *a.step + 1.0f;
    const float bc1 = 1.0f - powf(a.b1, tstep), bc2 = 1.0f - powf(a.b2, tstep);
    const float step_size = a.lr / bc1, bc2_sqrt = sqrtf(bc2);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = e0 + k * 256 + (int)threadIdx.x;
        if (e < n) {
            float grad = a.maximize ? -gr[k] : gr[k];
            const float param = pr[k];
            if (a.weight_decay != 0.f) grad = fmaf(param, a.weight_decay, grad);
            const float mn = mr[k] + (grad - mr[k]) * (1.0f - a.b1);
            const float vn = a.b2 * vr[k] + (1.0f - a.b2) * grad * grad;
            m[e] = mn;
            v[e] = vn;
            const float denom = sqrtf(vn) / bc2_sqrt + a.eps;
            p[e] = param - step_size * (mn / denom);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every work-group read *step before it got here; the last one publishes t for the next launch)
        if (atomicAdd(a.ticket, 1u) == gridDim.x - 1) {
            *a.step = tstep;
            *a.ticket = 0u;
        }
    }
}

// loss = -mean(x) (RatSpn.loss / DgcSpn.loss / NormalizingFlow.loss with one class: models/ratspn.py:184-191) and its
// gradient -g / n: torch runs mean, neg and their two backward nodes as four launches -- 12 us of a 140 us training step.
__global__ __launch_bounds__(1024) void neg_mean_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)x[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = (float)(-t / (double)n);
    }
}
__global__ void neg_mean_bwd_kernel(const float *__restrict__ gout, int64_t n, float *__restrict__ gx) {
    const float v = -gout[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) gx[i] = v;
}

}  // namespace dpk

using namespace dpk;

extern "C" int dpk_neg_mean_forward(const float *x, int64_t n, float *out, void *stream) {
    DPK_REQUIRE(x && out && n > 0, DPK_EINVAL, "neg_mean_forward: bad arguments");
    DPK_LAUNCH(neg_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    DPK_CHECK_LAUNCH("neg_mean_kernel");
    return DPK_OK;
}
extern "C" int dpk_neg_mean_backward(const float *grad_out, int64_t n, float *grad_x, void *stream) {
    DPK_REQUIRE(grad_out && grad_x && n > 0, DPK_EINVAL, "neg_mean_backward: bad arguments");
    const int blocks = (int)std::min<int64_t>(1024, cdiv(n, 256));
    DPK_LAUNCH(neg_mean_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_out, n, grad_x);
    DPK_CHECK_LAUNCH("neg_mean_bwd_kernel");
    return DPK_OK;
}

extern "C" int dpk_adam_step(int32_t n, const dpk_adam_tensor *tensors, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t maximize, float *step, uint32_t *ticket, void *stream) {
    DPK_REQUIRE(n >= 0 && n <= kAdamMaxTensors, DPK_EUNSUPPORTED, "adam_step: %d tensors (0..%d per call)", n, kAdamMaxTensors);
    if (n == 0) return DPK_OK;
    DPK_REQUIRE(tensors && step && ticket, DPK_EINVAL, "adam_step: null pointer");
    AdamArgs a{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const dpk_adam_tensor &q = tensors[i];
        DPK_REQUIRE(q.param && q.grad && q.exp_avg && q.exp_avg_sq && q.numel >= 0 && q.numel < (1ll << 31), DPK_EINVAL,
                    "adam_step: bad tensor %d", i);
        a.p[i] = q.param; a.g[i] = q.grad; a.m[i] = q.exp_avg; a.v[i] = q.exp_avg_sq;
        a.numel[i] = (int)q.numel;
        a.first_block[i] = blocks;
        blocks += (int)cdiv(q.numel, kAdamChunk);
    }
    a.first_block[n] = blocks;
    if (blocks == 0) return DPK_OK;
    a.n = n; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.maximize = maximize;
    a.step = step; a.ticket = ticket;
    DPK_LAUNCH(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    DPK_CHECK_LAUNCH("adam_step_kernel");
    return DPK_OK;
}
