/*
 * deeprob_hip.h -- C ABI of libdeeprob_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the tensorized density-evaluation path of DeeProb-kit
 * (RAT-SPN / DGC-SPN layers, RealNVP-1D coupling).  The reference is pure
 * Python on PyTorch and has no FFI of its own; every entry point below takes
 * the place of one `forward` / `apply_backward` body (or its autograd
 * backward) of a reference nn.Module and is what a ctypes binding placed in
 * that method would call.  The reference location replaced by each entry is
 * cited as `file:line` relative to the reference tree.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA storage),
 *     fp32 tensors are contiguous row-major, last index fastest;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued
 *     asynchronously on it and no entry point synchronises the device or the
 *     stream.  No device memory is allocated behind the caller's back.  Process
 *     state beyond the thread-local error string, all of it created lazily and
 *     once: (a) dpk_ratspn_forward's VALU route keeps ONE 64-byte host-mapped
 *     word (hipHostMalloc) in which a kernel that met marginalised evidence
 *     leaves the launch number -- a speed hint only, read without waiting: a
 *     stale value never changes results; (b) the kernels that need more than
 *     64 KB of LDS set hipFuncAttributeMaxDynamicSharedMemorySize on first use
 *     and cache the device's compute-unit count; (c) dpk_profile_next_kernel
 *     arms a thread-local, one-shot pair of events;
 *   - scratch memory is handed in by the caller (`ws`, `ws_bytes`); the
 *     matching *_workspace_bytes() query gives the required size;
 *   - the return value is 0 on success and a negative DPK_E* code otherwise;
 *     dpk_last_error() returns a thread-local message for the last failure.
 *     No C++ exception crosses the boundary.
 */
#ifndef DEEPROB_HIP_H
#define DEEPROB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPK_OK 0
#define DPK_EINVAL (-1)   /* bad argument (null pointer, size out of domain)   */
#define DPK_EWORKSPACE (-2) /* workspace too small                               */
#define DPK_ELAUNCH (-3)  /* hipLaunch / runtime error                          */
#define DPK_EUNSUPPORTED (-4) /* shape outside what the kernels are built for    */

/* flags shared by the RAT-SPN entry points */
#define DPK_FLAG_STRUCT_CACHED 1u /* structure tables in `ws` (built from mask /
                                     pad_mask by an earlier call with the same
                                     shapes) are still valid: skip rebuilding   */

#define DPK_FLAG_UNIT_SCALE 2u    /* hint: every Gaussian scale equals 1 (optimize_scale=False, the
                                     reference default).  Selects kernels that move only the means;
                                     the device re-checks and takes the general path where the hint
                                     is wrong, so it never changes results                         */

#define DPK_FLAG_PARAMS_CACHED 4u  /* dpk_ratspn_forward on the MFMA route (dpk_ratspn_forward_on_mfma):
                                     the parameter tables in `ws` were built by an earlier call from the
                                     same, unchanged loc / scale / weights: skip rebuilding them.  Ignored
                                     on every other route (tables are rebuilt per call there)              */

#define DPK_FLAG_IN_PIXEL_MAJOR 16u  /* dpk_spatial_prodsum_forward / dpk_spatial_sumprodroot_forward: the input map is
                                       [B, H, W, C] (torch's channels_last) instead of [B, C, H, W]; only on the streaming
                                       route (dpk_spatial_level_streams), DPK_EUNSUPPORTED elsewhere                      */
#define DPK_FLAG_OUT_PIXEL_MAJOR 32u /* dpk_spatial_prodsum_forward: write the output map as [B, OH, OW, Cout]; as above */
#define DPK_FLAG_LL_SUM_SPREAD 64u   /* dpk_ratspn_forward: `ll_sum` points at 17 doubles -- sixteen partial sums of the
                                       log-likelihoods (a work-group adds into one of them: 256 work-groups that finish
                                       together are otherwise 256 same-address fp64 atomics in series) and then the count;
                                       the caller adds the sixteen.  Without it: {sum, count}, 2 doubles.              */
#define DPK_FLAG_PARAMS_VERIFY 8u  /* the same belief, checked on the device: the entry point fingerprints the live
                                     parameter bytes (one small launch) and rebuilds its tables only if they differ
                                     from the bytes the tables were built from.  What a caller passes when all it
                                     knows is that addresses and version counters are unchanged -- a write through
                                     `param.data` moves neither.  DPK_FLAG_PARAMS_CACHED remains the caller's own
                                     guarantee that the bytes are unchanged (no check, no launch).  How an entry
                                     point checks is its own business: the RAT-SPN table work-groups fingerprint the
                                     slice of parameters each depends on -- inside the model kernel's own launch for
                                     the 32-sample kernels (round 4), as a ~5 us launch in front of the ring kernels --, the coupling tables use a fingerprint kernel + gated table
                                     kernels, entry points whose tables are cheap simply rebuild them.              */

const char *dpk_last_error(void);
int dpk_abi_version(void);

/* A workspace handed to the entry points below is about to be released (or its memory reused for something else): the
 * library forgets what it keeps per workspace ADDRESS -- the fingerprint slots of the cached-table checks and the
 * marginalised-evidence hint word (one host-mapped word per workspace; a performance hint only).  Optional: a process that
 * builds a few models never needs it; one that builds thousands would otherwise run the 4096-slot pool dry (every check
 * then degrades to a rebuild).  No reference counterpart (the reference keeps no derived tables).  Host side only, no
 * stream work. */
int dpk_workspace_forget(const void *workspace, int64_t bytes);

/* ------------------------------------------------------------------------ *
 * RAT-SPN                                                                   *
 * ------------------------------------------------------------------------ */

/* Scratch size for every dpk_*leaf* / dpk_ratspn_* call of a model with R leaf
 * regions of `dimension` features, `channels` distributions per region, sum
 * layers of `sums` nodes, `classes` root outputs. */
int64_t dpk_ratspn_workspace_bytes(int32_t in_features, int32_t regions, int32_t dimension,
                                   int32_t channels, int32_t depth, int32_t reps, int32_t sums,
                                   int32_t classes);

/* RegionGraphLayer.forward with GaussianLayer, eval mode
 * (deeprob/spn/layers/ratspn.py:87-108, distribution :160-213).
 *   x        [B, D]        inputs, NaN = marginalised
 *   mask     [R, d] int64  region variable ids (buffer `mask`)
 *   pad_mask [R, d] uint8  1 = dummy variable (buffer `pad_mask`, may be NULL)
 *   loc,scale[R, I, d]
 *   out      [B, R, I]     sum_j nan_to_num(Normal(loc,scale).log_prob(x[:,mask]))   */
/* 1 when dpk_gaussian_leaf_forward with these arguments runs on the matrix cores (unit-scale hint, channels in
 * {2,4,8,16}, 16-byte aligned x rows and out): DPK_FLAG_PARAMS_CACHED is honoured there as in dpk_ratspn_forward. */
int dpk_gaussian_leaf_forward_on_mfma(const float *x, const float *out, int32_t D, int32_t R, int32_t I, int32_t d,
                                      uint32_t flags);
int dpk_gaussian_leaf_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                              const uint8_t *pad_mask, const float *loc, const float *scale,
                              int32_t R, int32_t I, int32_t d, float *out, void *ws,
                              int64_t ws_bytes, uint32_t flags, void *stream);

/* Same with BernoulliLayer (ratspn.py:216-247): log p = -BCEWithLogits(logits, x). */
int dpk_bernoulli_leaf_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                               const uint8_t *pad_mask, const float *logits, int32_t R,
                               int32_t I, int32_t d, float *out, void *ws, int64_t ws_bytes,
                               uint32_t flags, void *stream);

/* Autograd backward of the Gaussian leaf.  g [B,R,I] upstream.
 * grad_loc / grad_scale [R,I,d] are OVERWRITTEN (may be NULL to skip),
 * grad_x [B,D] is OVERWRITTEN (may be NULL).  Marginalised (NaN) inputs get a
 * zero gradient.                                                            */
int dpk_gaussian_leaf_backward(const float *x, const float *g, int64_t B, int32_t D,
                               const int64_t *mask, const uint8_t *pad_mask, const float *loc,
                               const float *scale, int32_t R, int32_t I, int32_t d,
                               float *grad_loc, float *grad_scale, float *grad_x, void *ws,
                               int64_t ws_bytes, uint32_t flags, void *stream);

int dpk_bernoulli_leaf_backward(const float *x, const float *g, int64_t B, int32_t D,
                                const int64_t *mask, const uint8_t *pad_mask,
                                const float *logits, int32_t R, int32_t I, int32_t d,
                                float *grad_logits, void *ws, int64_t ws_bytes, uint32_t flags,
                                void *stream);

/* d/dx of the Bernoulli leaf layer (ratspn.py:243 through autograd): grad_x [B, D], 0 at marginalised inputs. */
int dpk_bernoulli_leaf_backward_input(const float *x, const float *g, int64_t B, int32_t D,
                                      const int64_t *mask, const uint8_t *pad_mask, const float *logits,
                                      int32_t R, int32_t I, int32_t d, float *grad_x, void *ws,
                                      int64_t ws_bytes, uint32_t flags, void *stream);

/* ProductLayer.forward (ratspn.py:272-286): in [B,R,N] -> out [B,R/2,N*N],
 * out[b,p,i*N+j] = in[b,2p,i] + in[b,2p+1,j].                                */
int dpk_product_forward(const float *in, int64_t B, int32_t R, int32_t N, float *out,
                        void *stream);
/* backward: g [B,R/2,N*N] -> grad_in [B,R,N] (overwritten). */
int dpk_product_backward(const float *g, int64_t B, int32_t R, int32_t N, float *grad_in,
                         void *stream);

/* SumLayer.forward, eval mode (ratspn.py:363-378): in [B,P,N], weight [P,S,N],
 * out[b,p,o] = logsumexp_n(in[b,p,n] + log_softmax(weight,2)[p,o,n]).        */
int dpk_sum_forward(const float *in, const float *weight, int64_t B, int32_t P, int32_t N,
                    int32_t S, float *out, void *ws, int64_t ws_bytes, void *stream);
/* backward: grad_in [B,P,N] and grad_weight [P,S,N] overwritten (either may be
 * NULL).  `out` is the forward result.                                       */
int dpk_sum_backward(const float *in, const float *weight, const float *out, const float *g,
                     int64_t B, int32_t P, int32_t N, int32_t S, float *grad_in,
                     float *grad_weight, void *ws, int64_t ws_bytes, void *stream);
int64_t dpk_sum_workspace_bytes(int64_t B, int32_t P, int32_t N, int32_t S);

/* RootLayer.forward (ratspn.py:446-458): in [B,M] (flattened), weight [C,M],
 * out[b,c] = logsumexp_n(in[b,n] + log_softmax(weight,1)[c,n]).  It is the sum
 * layer with P = 1.                                                          */
int dpk_root_forward(const float *in, const float *weight, int64_t B, int32_t M, int32_t C,
                     float *out, void *ws, int64_t ws_bytes, void *stream);
int dpk_root_backward(const float *in, const float *weight, const float *out, const float *g,
                      int64_t B, int32_t M, int32_t C, float *grad_in, float *grad_weight,
                      void *ws, int64_t ws_bytes, void *stream);

/* RatSpn.forward, eval mode, Gaussian leaves, whole model in ONE launch
 * (deeprob/spn/models/ratspn.py:105-122).  Built for depth in {1,2,3} and
 * channels, sums in {2,4,8}; anything else returns DPK_EUNSUPPORTED and the
 * caller chains the per-layer entry points instead.
 *   sum_weight0    [reps*2^(depth-1), S, I*I]   first sum layer (depth >= 2)
 *   sum_weight1    [reps*2^(depth-2), S, S*S]   second sum layer (depth == 3)
 *   root_weight    [C, reps*N_last]
 *   out            [B, C]
 *   leaf_out       [B, R, I] or NULL -- leaf log-likelihoods kept for backward
 *   ll_sum         double[2] or NULL -- += {sum of out, number of entries}    */
/* 1 when dpk_ratspn_forward with these arguments evaluates the leaf layer on the matrix cores
 * (depth 2, channels / sums in {2,4}, unit-scale hint, 16-byte aligned rows, no leaf_out): only then is
 * DPK_FLAG_PARAMS_CACHED honoured.  Pure function of its arguments, no device work.                     */
int dpk_ratspn_forward_on_mfma(const float *x, int32_t D, int32_t depth, int32_t reps, int32_t I, int32_t S,
                               int32_t C, int32_t want_leaf_out, uint32_t flags);
/* Batch size up to which the MFMA route takes its small-batch kernels (32-sample tiles, features split over the
 * waves of a work-group: csrc/ratspn_gemm_small.hip, the small-batch path of csrc/ratspn_leaf_gemm.hip and of the
 * folded product + sum / root layers) instead of the persistent 128-sample ring kernels.  Sets the threshold
 * (0: ring kernels always; negative: back to the built-in default, also given by DPK_GEMM_SMALL_MAX) and returns
 * the previous one.  Process-wide tuning knob: results of the two mappings agree to fp32 rounding.          */
int64_t dpk_ratspn_small_batch_max(int64_t samples);
/* Batch size FROM which the MFMA route of the two-channel depth-2 models over 784 variables takes its third mapping
 * (csrc/ratspn_gemm_slice.hip: persistent 32-sample blocks, the feature axis split over seven waves that keep their
 * slice of the mean table in registers; replaces RatSpn.forward, deeprob/spn/models/ratspn.py:105-122, like the other
 * two).  It has precedence over the small-batch kernels where both thresholds admit a batch.  Sets the threshold
 * (-1: never; below -1: back to the built-in default, also given by DPK_GEMM_SLICE_MIN) and returns the previous one.
 * Process-wide tuning knob: results of the mappings agree to fp32 rounding.                                          */
int64_t dpk_ratspn_slice_batch_min(int64_t samples);
/* Whether RatSpn.forward / RegionGraphLayer.forward (deeprob/spn/models/ratspn.py:105-122, layers/ratspn.py:87-108) may take
 * the matrix-core route (split-f16 MFMA leaf GEMM, >= 22 bits per product, guarded) at all: 1 = yes (default, also
 * DPK_RATSPN_GEMM in the environment), 0 = every launch on the exact fp32 vector-ALU kernels; negative = query only.
 * Returns the previous setting.  Process-wide measurement knob: bench.py quotes the exact-fp32 step beside the headline. */
int32_t dpk_ratspn_mfma_route(int32_t enable);
int dpk_ratspn_forward(const float *x, int64_t B, int32_t D, const int64_t *mask,
                       const uint8_t *pad_mask, const float *loc, const float *scale,
                       const float *sum_weight0, const float *sum_weight1,
                       const float *root_weight,
                       int32_t depth, int32_t reps, int32_t I, int32_t S, int32_t C,
                       float *out, float *leaf_out, double *ll_sum, void *ws, int64_t ws_bytes,
                       uint32_t flags, void *stream);

/* The log-softmax tables (torch.log_softmax(weight) at ratspn.py:375 and :455) and matrix-core fragments of a depth-2
 * model's SumLayer (weight [R0/2, S0, N0*N0]) and RootLayer (weight [C, (R1/2)*N1*N1]) in ONE launch, into the two layers'
 * own workspaces (dpk_prodsum_workspace_bytes each): the dpk_prodsum_forward / dpk_prodroot_forward calls that follow take
 * DPK_FLAG_PARAMS_CACHED.  DPK_EUNSUPPORTED when a layer is outside the matrix-core route (the layers then build their own). */
int dpk_upper_tables_pair(const float *sum_weight, int32_t R0, int32_t N0, int32_t S0, void *ws0, int64_t ws0_bytes,
                          const float *root_weight, int32_t R1, int32_t N1, int32_t C, void *ws1, int64_t ws1_bytes,
                          void *stream);

/* Backward of a whole level -- ProductLayer (ratspn.py:272-286) under a SumLayer (:363-378; root = 0: weight
 * [R/2, S, N*N], out / g [B, R/2, S]) or under the RootLayer (:446-458; root = 1: weight [S, (R/2)*N*N], out / g [B, S]) --
 * from the level's INPUT in [B, R, N]: the [B, R/2, N*N] product tensor is never formed.  grad_in [B, R, N] and
 * grad_weight (like weight) may each be NULL.  Equivalent to dpk_product_forward + dpk_sum_backward (dpk_root_backward)
 * + dpk_product_backward; like them it only uses in - out, so a common per-sample shift of (in, out) is allowed.
 * ws as dpk_sum_workspace_bytes(B, R/2, N*N, S).  Built for N in {2,4,8}, S in {2,4,8} (sum) / (R/2)*N*N <= 1024 (root)
 * and for N = 16, S in {8,16} (sum); DPK_EUNSUPPORTED otherwise.                                                                                       */
int dpk_prodsum_backward(const float *in, const float *weight, const float *out, const float *g, int64_t B, int32_t R,
                         int32_t N, int32_t S, int32_t root, float *grad_in, float *grad_weight, void *ws,
                         int64_t ws_bytes, void *stream);

/* RatSpn.forward of a TRAINING step (models/ratspn.py:105-122 under autograd; the loop of torch/routines.py:150-170):
 * the same single launch as dpk_ratspn_forward on the MFMA route, which on its way up also writes what the layers'
 * backward entry points read -- the three launches + three table builds of the per-layer chain become one launch.
 *   leaf_rel [B, R, I]       leaf layer outputs      \  each RELATIVE to the sample: plus 1/2 sum x^2 over the variables
 *   sum_rel  [B, 2 reps, S]  first sum layer outputs  } below the node (the GEMM form carries that common term to the root);
 *   out_rel  [B, C]          root outputs            /  dpk_sum_backward / dpk_root_backward only use in - out, so
 *                                                       (leaf_rel, sum_rel) and (sum_rel, out_rel) are valid (in, out) pairs.
 *   out      [B, C]          the log-likelihoods themselves.
 * Built for depth 2, Gaussian leaves with the unit-scale hint, 8 channels (<= 8 repetitions, <= 32 classes) or 2 / 4
 * channels up to dpk_ratspn_small_batch_max samples; DPK_EUNSUPPORTED otherwise (the caller chains the layers).        */
int dpk_ratspn_forward_train(const float *x, int64_t B, int32_t D, const int64_t *mask, const uint8_t *pad_mask,
                             const float *loc, const float *scale, const float *sum_weight0, const float *root_weight,
                             int32_t depth, int32_t reps, int32_t I, int32_t S, int32_t C, float *out, float *leaf_rel,
                             float *sum_rel, float *out_rel, void *ws, int64_t ws_bytes, uint32_t flags, void *stream);

/* ------------------------------------------------------------------------ *
 * RealNVP-1D                                                                *
 * ------------------------------------------------------------------------ */

/* CouplingLayer1d.apply_backward (inverse = 0) / apply_forward (inverse = 1),
 * conditioner depth 1 (deeprob/flows/layers/coupling.py:72-104, network :45-56,
 * ScaledTanh deeprob/torch/utils.py:52-70).  One fused launch on the fp32
 * matrix cores: Linear(D,units) -> ReLU -> Linear(units, 2D or D) -> epilogue.
 *   x [B,D];  mask, inv_mask [D] binary floats with n_masked / n_transformed
 *   non-zeros;  W1 [units,D], b1 [units], W2 [2D or D, units], b2 [2D or D];
 *   act_weight [1] (ScaledTanh, affine only);  affine = 1 RealNVP, 0 NICE.
 *   in_scale / in_shift [D] or NULL: per-variable affine applied to x on load
 *   (an eval-mode BatchNormLayer1d in front of the layer, see dpk_bn1d_fold).
 *   out [B,D] (must not alias x);  ldj [B] = (-/+) sum_d s, overwritten or
 *   accumulated (accumulate_ldj).                                              */
int64_t dpk_coupling1d_workspace_bytes(int32_t D, int32_t units, int32_t n_masked, int32_t n_transformed);
int dpk_coupling1d_forward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                           int32_t n_masked, int32_t n_transformed, const float *W1, const float *b1,
                           const float *W2, const float *b2, int32_t units, const float *act_weight,
                           const float *in_scale, const float *in_shift, int32_t affine, int32_t inverse,
                           float *out, float *ldj, int32_t accumulate_ldj, void *ws, int64_t ws_bytes,
                           void *stream);

/* The same layer for the reference's own masks (coupling.py:58-60: mask = arange(D) % 2, or its complement for
 * every other layer) with a depth-1 conditioner of 32 / 64 / 96 / 128 units and D % 8 == 0: both GEMMs on the f16
 * matrix cores with two-way f16 splits of every operand and fp32 accumulation (>= 22 significant bits per product),
 * hidden activations kept in registers.  masked_parity: parity of the columns where mask != 0.  flags:
 * DPK_FLAG_PARAMS_CACHED = the packed tables in `ws` were built by an earlier call from the same, unchanged
 * W1 / b1 / W2 / b2 / in_scale / in_shift.  x and out must be 16-byte aligned and must not alias.        */
int64_t dpk_coupling1d_pairs_workspace_bytes(int32_t D, int32_t units);
int dpk_coupling1d_pairs_forward(const float *x, int64_t B, int32_t D, int32_t masked_parity, const float *W1,
                                 const float *b1, const float *W2, const float *b2, int32_t units,
                                 const float *act_weight, const float *in_scale, const float *in_shift,
                                 int32_t affine, int32_t inverse, float *out, float *ldj, int32_t accumulate_ldj,
                                 void *ws, int64_t ws_bytes, uint32_t flags, void *stream);
/* The LAST coupling of a flow fused with what follows it in NormalizingFlow.log_prob (reference: flows/models/base.py:
 * 123-143): the density-direction coupling above, the per-variable affine behind it (out_scale / out_shift: an eval-mode
 * BatchNormLayer1d folded by dpk_bn1d_fold, or NULL) and the diagonal Normal base --
 *   ll[b] = sum_d log N(out_scale_d u_d + out_shift_d; base_loc_d, base_scale_d) + ildj_in[b] - sum s + *ildj_const.
 * u is never written: the layer reads x once more than the plain coupling writes nothing, and the separate
 * dpk_normal_base_logprob pass (one more read of u) disappears.  ildj_in / ildj_const may be NULL.  Same shapes,
 * workspace and flags as dpk_coupling1d_pairs_forward.                                                        */
int dpk_coupling1d_pairs_logprob(const float *x, int64_t B, int32_t D, int32_t masked_parity, const float *W1,
                                 const float *b1, const float *W2, const float *b2, int32_t units,
                                 const float *act_weight, const float *in_scale, const float *in_shift, int32_t affine,
                                 const float *out_scale, const float *out_shift, const float *base_loc,
                                 const float *base_scale, const float *ildj_in, const float *ildj_const, float *ll,
                                 void *ws, int64_t ws_bytes, uint32_t flags, void *stream);

/* The parameter tables of n alternating-mask couplings verified / rebuilt in TWO launches (one fingerprint pass, one
 * gated pack pass over all layers) instead of two per layer: afterwards each layer's dpk_coupling1d_pairs_forward /
 * _logprob may be called with DPK_FLAG_PARAMS_CACHED.  Entry i carries the table-related arguments of those calls;
 * flags = DPK_FLAG_PARAMS_VERIFY: rebuild only if the parameters' fingerprint differs from the one the tables in ws
 * were built from; 0: rebuild.  n <= 16.                                                                          */
typedef struct dpk_pairs_tables_args {
    const float *W1, *b1, *W2, *b2, *in_scale, *in_shift;
    void *ws;
    int64_t ws_bytes;
    int32_t D, units, masked_parity, affine;
    uint32_t flags;
} dpk_pairs_tables_args;
int dpk_coupling1d_pairs_tables(int32_t n, const dpk_pairs_tables_args *layers, void *stream);

/* Eval-mode BatchNormLayer1d.apply_backward (inverse = 0) / apply_forward (1)
 * (deeprob/flows/utils.py:118-153) as a per-variable affine y = x*scale_out + shift_out
 * composed with an optional incoming affine; ldj_const [1] = its (constant) log-det,
 * overwritten or accumulated.  All vectors have D entries (the reference's [1,D]).   */
int dpk_bn1d_fold(const float *weight, const float *bias, const float *running_var,
                  const float *running_mean, float eps, int32_t D, int32_t inverse, const float *scale_in,
                  const float *shift_in, float *scale_out, float *shift_out, float *ldj_const,
                  int32_t accumulate, void *stream);
/* Every eval-mode BatchNormLayer1d of a flow in ONE launch (round 3): NormalizingFlow.log_prob walks its layers in a
 * Python loop (flows/models/base.py:182-193); per layer the fold above is a 5 us launch of a D-element kernel.  Entry i
 * is the argument list of dpk_bn1d_fold (scale_in / shift_in must not be outputs of another entry of the same call);
 * ldj_total (may be NULL) receives the sum of the n constants.                                                    */
typedef struct dpk_bn1d_fold_args {
    const float *weight, *bias, *running_var, *running_mean, *scale_in, *shift_in;
    float *scale_out, *shift_out, *ldj_const;
    float eps;
    int32_t D, inverse, accumulate;
} dpk_bn1d_fold_args;
int dpk_bn1d_fold_many(int32_t n, const dpk_bn1d_fold_args *layers, float *ldj_total, void *stream);
/* out[b,d] = x[b,d]*scale[d] + shift[d]  (materialises a folded BatchNormLayer1d). */
int dpk_affine1d_forward(const float *x, const float *scale, const float *shift, int64_t B, int32_t D,
                         float *out, void *stream);
/* LogitLayer.apply_backward (inverse = 0) / apply_forward (1) in one pass (deeprob/flows/utils.py:276-294):
 * x [B, D] (any trailing shape flattened to D), out [B, D], ldj [B] overwritten; ldj_const = the layer's `ldj`
 * buffer, -D log(1 - 2 alpha).                                                                        */
int dpk_logit1d_forward(const float *x, int64_t B, int32_t D, float alpha, float ldj_const, int32_t inverse,
                        float *out, float *ldj, void *stream);
/* NormalizingFlow.forward tail with the default Normal base (flows/models/base.py:139-143):
 * out[b] = sum_d log N(u[b,d]*scale_in[d]+shift_in[d]; loc[d], scale[d]) + ildj[b] + ildj_const.
 * scale_in/shift_in, ildj, ildj_const may be NULL.                                     */
int dpk_normal_base_logprob(const float *u, const float *scale_in, const float *shift_in, const float *loc,
                            const float *scale, const float *ildj, const float *ildj_const, int64_t B,
                            int32_t D, float *out, void *stream);

/* Eval route for RAT-SPN shapes outside dpk_ratspn_forward's envelope (e.g. I = S = 16): a ProductLayer folded
 * into the SumLayer / RootLayer above it (ratspn.py:272-286 + :363-378 / :446-458), the [B,P,N^2] product tensor
 * never reaches HBM.  in [B,R,N]; prodsum: weight [R/2,S,N^2] -> out [B,R/2,S]; prodroot: weight [C,(R/2)*N^2] ->
 * out [B,C].  N <= 32.  Workspace: dpk_prodsum_workspace_bytes(R,N,S) (S = C for the root).                 */
int64_t dpk_prodsum_workspace_bytes(int32_t R, int32_t N, int32_t S);
/* flags: DPK_FLAG_PARAMS_CACHED = the workspace still holds the softmax rows (and MFMA fragments) that an earlier
 * call of the same entry point built from this very weight tensor, unchanged since: they are not rebuilt.     */
int dpk_prodsum_forward(const float *in, const float *weight, int64_t B, int32_t R, int32_t N, int32_t S, float *out,
                        void *ws, int64_t ws_bytes, uint32_t flags, void *stream);
int dpk_prodroot_forward(const float *in, const float *weight, int64_t B, int32_t R, int32_t N, int32_t C, float *out,
                         void *ws, int64_t ws_bytes, uint32_t flags, void *stream);

/* RatSpn.mpe (mode 0; deeprob/spn/models/ratspn.py:124-162) and RatSpn.sample (mode 1; :164-182): the whole top-down pass
 * -- RootLayer.mpe / .sample (deeprob/spn/layers/ratspn.py:460-474 / :476-490), SumLayer (:380-399 / :401-417), ProductLayer
 * (:288-304 / :306-330), RegionGraphLayer.mpe / .sample with unpad_samples (:118-136 / :138-157, :68-85) -- in one launch,
 * a wave per sample.  dist: 0 = Gaussian, 1 = Bernoulli leaves.  x [B, D]: evidence, NaN entries are completed (NULL:
 * everything is generated); y [B] int64: class of the root to descend from (NULL: class 0).  act / logw are HOST arrays of
 * device pointers: act[0] = leaf layer output [B, reps 2^depth, I], act[t] (1 <= t < depth) = output of sum level t
 * [B, reps 2^(depth-t), S] (mode 0 only; the product tensors are not needed); logw[t] (1 <= t < depth) = log_softmax of sum
 * level t's weight [reps 2^(depth-t), S, N^2] (N = I for t = 1, else S), logw[depth] = log_softmax of the root weight
 * [C, reps N^2]; logw[0] is ignored.  src [reps, D] int32: (region within the repetition) * d + position that holds
 * variable f (inv_mask without the dummy variables).  p0 / p1: loc / scale or logits / NULL, [reps 2^depth, I, d].
 * Sampling draws are counter based on `seed` (csrc/ratspn_topdown.hip states the counter layout: a test can replay them).
 * out [B, D]; choice (optional, [B, 1 + 2^depth] int32): the repetition and the leaf channel chosen per region.           */
int dpk_ratspn_topdown(int32_t mode, int32_t dist, int64_t B, int32_t D, int32_t depth, int32_t reps, int32_t I,
                       int32_t S, int32_t C, int32_t d, const float *x, const int64_t *y,
                       const float *const *act, const float *const *logw, const int32_t *src,
                       const float *p0, const float *p1, uint64_t seed, float *out, int32_t *choice,
                       void *stream);

/* ---- DGC-SPN spatial layers (NCHW fp32; deeprob/spn/layers/dgcspn.py) ------------------ */
/* SpatialGaussianLayer.forward (dgcspn.py:101-120):
 * out[b,k,h,w] = sum_c nan_to_num(log N(x[b,c,h,w]; loc[k,c,h,w], scale[k,c,h,w])); NaN x = marginalised. */
int dpk_spatial_gaussian_forward(const float *x, const float *loc, const float *scale, int64_t B, int32_t K,
                                 int32_t C, int32_t H, int32_t W, float *out, void *stream);
/* its autograd: grad_loc / grad_scale [K,C,H,W], grad_x [B,C,H,W]; any may be NULL. */
int dpk_spatial_gaussian_backward(const float *x, const float *g, const float *loc, const float *scale,
                                  int64_t B, int32_t K, int32_t C, int32_t H, int32_t W, float *grad_loc,
                                  float *grad_scale, float *grad_x, void *stream);
/* SpatialProductLayer.forward (dgcspn.py:224-236): F.pad(zero; pad_left/pad_top, the right/bottom
 * padding is implied by OH/OW) then F.conv2d with the all-ones depthwise kernel (depthwise=1,
 * OC == C) or the one-hot kernels enumerating every channel combination in itertools.product
 * order (depthwise=0, OC == C^(kh*kw); dgcspn.py:187-193).                                    */
int dpk_spatial_product_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OC,
                                int32_t OH, int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                int32_t dh, int32_t dw, int32_t pad_top, int32_t pad_left, int32_t depthwise,
                                float *out, void *stream);
int dpk_spatial_product_backward(const float *g, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OC,
                                 int32_t OH, int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                 int32_t dh, int32_t dw, int32_t pad_top, int32_t pad_left, int32_t depthwise,
                                 float *grad_in, void *stream);
/* SpatialSumLayer.forward (dgcspn.py:289-304):
 * out[b,o,h,w] = logsumexp_c(x[b,c,h,w] + log_softmax(weight,1)[o,c,h,w]), weight [Cout,Cin,H,W]. */
int64_t dpk_spatial_sum_workspace_bytes(int32_t Cin, int32_t Cout, int32_t H, int32_t W);
int dpk_spatial_sum_forward(const float *x, const float *weight, int64_t B, int32_t Cin, int32_t Cout, int32_t H,
                            int32_t W, float *out, void *ws, int64_t ws_bytes, void *stream);
int dpk_spatial_sum_backward(const float *x, const float *weight, const float *out, const float *g, int64_t B,
                             int32_t Cin, int32_t Cout, int32_t H, int32_t W, float *grad_x, float *grad_weight,
                             void *ws, int64_t ws_bytes, void *stream);

/* Eval route of one DGC-SPN level: depthwise SpatialProductLayer (<= 4 taps, C <= 32) fused with the
 * SpatialSumLayer that follows it (models/dgcspn.py:146-147), the product map never reaches HBM.
 * Geometry as dpk_spatial_product_forward, weight [Cout,C,OH,OW], workspace of
 * dpk_spatial_sum_workspace_bytes(C,Cout,OH,OW).  DPK_EUNSUPPORTED outside that envelope.
 * flags: DPK_FLAG_PARAMS_CACHED = the workspace still holds the softmaxed weight tables a previous call built
 * from this very weight tensor (unchanged since), they are not rebuilt.  8 -> 8 channel levels on batches of
 * 256 samples and more (input 16-byte aligned) take the streaming kernel: the 64 weights of an output pixel
 * stay in registers while its work-group walks a slice of the batch staged through LDS.                     */
int dpk_spatial_prodsum_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH,
                                int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh,
                                int32_t dw, int32_t pad_top, int32_t pad_left, const float *weight, int32_t Cout,
                                float *out, void *ws, int64_t ws_bytes, uint32_t flags, void *stream);
/* Whether a fused level of these shapes runs on the streaming route of dpk_spatial_prodsum_forward (last = 0) /
 * dpk_spatial_sumprodroot_forward (last = 1; geom6, K as there) at this batch size -- the route that accepts
 * DPK_FLAG_IN_PIXEL_MAJOR / DPK_FLAG_OUT_PIXEL_MAJOR, so that DgcSpn.forward (deeprob/spn/models/dgcspn.py:134-151) can keep
 * the maps between two such levels pixel-major.  geom = {OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left}.  1 / 0.     */
int32_t dpk_spatial_level_streams(int32_t last, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *geom5,
                                  int32_t Cout, const int32_t *geom6, int32_t K);

/* SpatialGaussianLayer followed by the FIRST depthwise product + sum level, eval route, in one launch
 * (models/dgcspn.py:134-147 at i = 0, 1; layers/dgcspn.py:101-120 for the leaf): x [B,Cx,H,W] (NaN = marginalised),
 * loc / scale [K,Cx,H,W], weight [Cout,K,OH,OW], workspace and flags as dpk_spatial_prodsum_forward.  The [B,K,H,W] leaf
 * map is never written.  2 x 2 windows (any stride, dilation, padding), K = 8, 16 or 32, Cout <= 32; the pooling level
 * (stride 2, no padding, even output width, W % 4 == 0) takes a form with 16-byte image loads.  DPK_EUNSUPPORTED
 * otherwise: the caller runs the two entry points it replaces.                                                       */
int dpk_spatial_leaf_prodsum_forward(const float *x, const float *loc, const float *scale, int64_t B, int32_t Cx, int32_t K,
                                     int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t kh, int32_t kw, int32_t sh,
                                     int32_t sw, int32_t dh, int32_t dw, int32_t pad_top, int32_t pad_left,
                                     const float *weight, int32_t Cout, float *out, void *ws, int64_t ws_bytes,
                                     uint32_t flags, void *stream);
/* Smallest leaf channel count for which dpk_spatial_leaf_prodsum_forward takes the general (stride-1) form of the fused
 * first level of DgcSpn.forward (deeprob/spn/models/dgcspn.py:134-147; default 16, DPK_DGC_LEAF_FUSE_MIN_K in the
 * environment sets the initial value): k > 0 sets it, k <= 0 only queries.  Returns the previous value.  Measurement knob. */
int32_t dpk_spatial_leaf_fuse_min_k(int32_t k);

/* ---- RealNVP-1D training route (autograd of the flow) ------------------- */
/* Backward of CouplingLayer1d.apply_backward (flows/layers/coupling.py:72-87), depth-1 conditioner:
 * given grad_u [B,D] and grad_ildj [B] (either may be NULL = zeros) writes grad_x [B,D] and, where
 * non-NULL, grad_W1 [units,D], grad_b1 [units], grad_W2 [2D|D,units], grad_b2 [2D|D], grad_act [1].
 * The forward activations are recomputed.  Binary masks.                                       */
int64_t dpk_coupling1d_backward_workspace_bytes(int64_t B, int32_t D, int32_t units, int32_t affine);
int dpk_coupling1d_backward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                            const float *W1, const float *b1, const float *W2, const float *b2, int32_t units,
                            const float *act_weight, int32_t affine, const float *grad_u, const float *grad_ildj,
                            float *grad_x, float *grad_W1, float *grad_b1, float *grad_W2, float *grad_b2,
                            float *grad_act, void *ws, int64_t ws_bytes, void *stream);
/* Any conditioner depth (CouplingLayer1d(depth = n_hidden), coupling.py:45-56): W[i] [widths[i], in_i] and
 * b[i] [widths[i]] for i = 0..n_hidden are HOST arrays of device pointers, widths[n_hidden] = 2D (affine) or D;
 * the MLP is chained through the generic fp32-MFMA GEMM kernel.  forward: inverse = 0 apply_backward, 1
 * apply_forward; ldj is overwritten.  backward: as dpk_coupling1d_backward, grad_W / grad_b host arrays of
 * device pointers (entries or the arrays may be NULL).  1 <= n_hidden <= 8.
 * The forward's workspace layout (hidden activations, conditioner output) is a prefix of the backward's:
 * ws_holds_forward != 0 tells the backward that `ws` (sized for backward = 1) still holds what
 * dpk_coupling1d_mlp_forward(inverse = 0) left there for the same x and parameters, so the conditioner is
 * not evaluated again; 0 recomputes it.                                                                   */
int64_t dpk_coupling1d_mlp_workspace_bytes(int64_t B, int32_t n_hidden, const int32_t *widths, int32_t backward);
int dpk_coupling1d_mlp_forward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                               int32_t n_hidden, const float *const *W, const float *const *b, const int32_t *widths,
                               const float *act_weight, int32_t affine, int32_t inverse, float *out, float *ldj,
                               void *ws, int64_t ws_bytes, void *stream);
int dpk_coupling1d_mlp_backward(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                                int32_t n_hidden, const float *const *W, const float *const *b, const int32_t *widths,
                                const float *act_weight, int32_t affine, const float *grad_u, const float *grad_ildj,
                                float *grad_x, float *const *grad_W, float *const *grad_b, float *grad_act,
                                int32_t ws_holds_forward, void *ws, int64_t ws_bytes, void *stream);
/* Backward of the SAMPLING direction, CouplingLayer1d.apply_forward (coupling.py:89-104): what
 * NormalizingFlow.rsample differentiates (flows/models/base.py:159-180).  x = the input of apply_forward, grad_out /
 * grad_ldj = gradients w.r.t. its two outputs; everything else as dpk_coupling1d_mlp_backward.               */
int dpk_coupling1d_mlp_backward_inverse(const float *x, int64_t B, int32_t D, const float *mask, const float *inv_mask,
                                        int32_t n_hidden, const float *const *W, const float *const *b,
                                        const int32_t *widths, const float *act_weight, int32_t affine,
                                        const float *grad_out, const float *grad_ldj, float *grad_x,
                                        float *const *grad_W, float *const *grad_b, float *grad_act,
                                        int32_t ws_holds_forward, void *ws, int64_t ws_bytes, void *stream);
/* Backward of BatchNormLayer1d.apply_forward (flows/utils.py:141-153, running statistics):
 * x = (u - bias) exp(-weight) sqrt(var + eps) + mean.  grad_ldj [B] may be NULL, grad_weight / grad_bias [D] may be
 * NULL.  Workspace >= (5 D + 64) floats.                                                                     */
int dpk_bn1d_inverse_backward(const float *u, const float *grad_x, const float *grad_ldj, int64_t B, int32_t D,
                              const float *weight, const float *bias, const float *running_var, float eps,
                              float *grad_u, float *grad_weight, float *grad_bias, void *ws, int64_t ws_bytes,
                              void *stream);
/* Training-mode BatchNormLayer1d.apply_backward (flows/utils.py:118-139): torch.var_mean over the batch
 * (unbiased), running_var / running_mean updated IN PLACE with `momentum`, out = (x-mean)/sqrt(var+eps)
 * * exp(weight) + bias, ildj_const[0] = sum_d(weight_d - 0.5 log(var_d+eps)); save_mean/save_var [D]
 * are the batch statistics for dpk_bn1d_backward.  Workspace >= 2*D floats.  B >= 2.            */
int dpk_bn1d_train_forward(const float *x, int64_t B, int32_t D, const float *weight, const float *bias,
                           float *running_var, float *running_mean, float momentum, float eps, float *out,
                           float *ildj_const, float *save_mean, float *save_var, void *ws, int64_t ws_bytes,
                           void *stream);
/* Batch-sharded (synchronised) training-mode BatchNormLayer1d: every rank holds B rows of a batch of B_total rows
 * and all ranks derive the statistics of the WHOLE batch, i.e. what the single-process reference computes
 * (flows/utils.py:122-128).  No reference counterpart for the exchange itself (the reference is single device).
 *   1. dpk_bn1d_local_moments: moments[2D+1] = {B, mean_d of the local rows, sum_b (x - mean_d)^2}
 *   2. the caller all-gathers them: gathered[world][2D+1]
 *   3. dpk_bn1d_sync_forward: combines in rank order (Chan's pairwise update: bit-identical on every rank), updates
 *      the running statistics, writes out / ildj_const / save_mean / save_var like dpk_bn1d_train_forward
 *   backward: dpk_bn1d_backward_sums: sums[2D+1] = {sum_b g_u, sum_b g_u xhat, sum_b g_ildj} of the local rows; the
 *   caller all-reduces them weighted by B/B_total and rescales by B_total/B (each rank's loss is the mean over ITS
 *   rows) -> sums_x; dpk_bn1d_sync_backward: grad_x from sums_x (whole batch), grad_weight / grad_bias from the
 *   local sums_p, so that the sample-weighted gradient average over the ranks equals the single-process gradient.  */
int dpk_bn1d_local_moments(const float *x, int64_t B, int32_t D, float *moments, void *stream);
int dpk_bn1d_sync_forward(const float *x, int64_t B, int32_t D, const float *weight, const float *bias,
                          const float *gathered, int32_t world, float *running_var, float *running_mean,
                          float momentum, float eps, float *out, float *ildj_const, float *save_mean,
                          float *save_var, void *ws, int64_t ws_bytes, void *stream);
int dpk_bn1d_backward_sums(const float *x, const float *grad_u, const float *grad_ildj, int64_t B, int32_t D,
                           const float *mean, const float *var, float eps, float *sums, void *stream);
int dpk_bn1d_sync_backward(const float *x, const float *grad_u, int64_t B, int64_t B_total, int32_t D,
                           const float *weight, const float *mean, const float *var, float eps,
                           const float *sums_x, const float *sums_p, float *grad_x, float *grad_weight,
                           float *grad_bias, void *stream);
/* Backward of BatchNormLayer1d.apply_backward: train=1 with the saved batch statistics (gradient flows
 * through them), train=0 with the running statistics as constants.  grad_ildj [B] may be NULL;
 * grad_weight / grad_bias [D] may be NULL.  Workspace >= (2*D + 64) floats.                     */
int dpk_bn1d_backward(const float *x, const float *grad_u, const float *grad_ildj, int64_t B, int32_t D,
                      const float *weight, const float *mean, const float *var, float eps, int32_t train,
                      float *grad_x, float *grad_weight, float *grad_bias, void *ws, int64_t ws_bytes,
                      void *stream);
/* d/du of dpk_normal_base_logprob without incoming affine: grad_u[b,d] = -g[b](u-loc)/scale^2.  */
int dpk_normal_base_backward(const float *u, const float *loc, const float *scale, const float *g, int64_t B,
                             int32_t D, float *grad_u, void *stream);

/* ---- training-mode probabilistic dropout (RAT-SPN ratspn.py:98-100, :371-372; DGC-SPN dgcspn.py:113-114,
 * :297-298).  The reference draws torch.rand_like masks; here element `idx` of a call is dropped iff
 * splitmix64(seed + idx * 0x9E3779B97F4A7C15) >> 40 < p * 2^24, evaluated identically by the forward and the
 * backward kernels (no mask tensor).  idx = flat index of the reference's masked tensor: [B,R,I,d] for the
 * RAT-SPN leaf layer, [B,K,C,H,W] for the spatial Gaussian layer, the sum layer's input for dpk_dropout_fill. */
int dpk_leaf_forward_dropout(int32_t dist /* 0 Normal, 1 Bernoulli */, const float *x, int64_t B, int32_t D,
                             const int64_t *mask, const uint8_t *pad_mask, const float *p0, const float *p1,
                             int32_t R, int32_t I, int32_t d, float drop_p, uint64_t seed, float *out, void *stream);
int dpk_leaf_backward_dropout(int32_t dist, const float *x, const float *g, int64_t B, int32_t D, const int64_t *mask,
                              const uint8_t *pad_mask, const float *p0, const float *p1, int32_t R, int32_t I,
                              int32_t d, float drop_p, uint64_t seed, float *grad_p0, float *grad_p1, float *grad_x,
                              void *ws, int64_t ws_bytes, uint32_t flags, void *stream);
int dpk_spatial_gaussian_forward_dropout(const float *x, const float *loc, const float *scale, int64_t B, int32_t K,
                                         int32_t C, int32_t H, int32_t W, float drop_p, uint64_t seed, float *out,
                                         void *stream);
int dpk_spatial_gaussian_backward_dropout(const float *x, const float *g, const float *loc, const float *scale,
                                          int64_t B, int32_t K, int32_t C, int32_t H, int32_t W, float drop_p,
                                          uint64_t seed, float *grad_loc, float *grad_scale, float *grad_x,
                                          void *stream);
/* out[i] = dropped(i) ? fill : x[i]  (sum-layer input dropout, fill = -inf); its backward is the same call on the
 * upstream gradient with fill = 0.                                                                           */
int dpk_dropout_fill(const float *x, int64_t n, float drop_p, uint64_t seed, float fill, float *out, void *stream);

/* Eval route of the last DGC-SPN level: depthwise product (<= 4 taps) folded into SpatialRootLayer
 * (models/dgcspn.py:146-150, layers/dgcspn.py:343-355): out[b,k] = logsumexp_m(prod[b,m] + log_softmax(weight,1)[k,m]),
 * weight [K, C*OH*OW], the product map never reaches HBM.  DPK_EUNSUPPORTED outside that envelope.        */
int64_t dpk_spatial_prodroot_workspace_bytes(int32_t C, int32_t OH, int32_t OW, int32_t K);
int dpk_spatial_prodroot_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH,
                                 int32_t OW, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh,
                                 int32_t dw, int32_t pad_top, int32_t pad_left, const float *weight, int32_t K,
                                 float *out, void *ws, int64_t ws_bytes, void *stream);

/* Measurement hook: the NEXT dominant-kernel launch made from this thread (the fused / leaf
 * forward kernel) is bracketed by hipEventRecord(ev_start) / hipEventRecord(ev_stop) on its stream,
 * so a harness can time that kernel alone inside a longer step.  One-shot; pass NULLs to cancel. */
int dpk_profile_next_kernel(void *ev_start, void *ev_stop);
/* The same for the next launch of ONE kernel family (a step of several kernels: time the dominant one). */
#define DPK_KERNEL_RATSPN_FUSED 1        /* dpk_ratspn_forward (either route)                    */
#define DPK_KERNEL_RATSPN_LEAF 2         /* dpk_gaussian_leaf_forward / dpk_bernoulli_leaf_forward */
#define DPK_KERNEL_COUPLING1D 3          /* dpk_coupling1d_forward, fused kernel                 */
#define DPK_KERNEL_SPATIAL_PRODSUM 4     /* dpk_spatial_prodsum_forward                          */
#define DPK_KERNEL_SPATIAL_SUMPRODROOT 5 /* dpk_spatial_sumprodroot_forward                      */
int dpk_profile_next_kernel_of(void *ev_start, void *ev_stop, int32_t kernel_id);

/* torch.optim.Adam's update (the optimiser the reference's training loops build by name: torch/utils.py:32-49, stepped at
 * torch/routines.py:164 / :280) for up to 96 parameter tensors in ONE launch: m = m + (g - m)(1 - b1),
 * v = b2 v + (1 - b2) g^2, p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), fp32, non-amsgrad; weight_decay
 * is the L2 form (added to the gradient), maximize negates the gradient.  `step` [1] fp32 holds the number of updates
 * done (read by every work-group, incremented by the last to finish: the launch is HIP-graph capturable), `ticket` [1]
 * a zeroed word the library uses for that hand-over.  The tensor table is host memory, copied into the kernel arguments. */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    int64_t numel;
} dpk_adam_tensor;
int dpk_adam_step(int32_t n, const dpk_adam_tensor *tensors, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t maximize, float *step, uint32_t *ticket, void *stream);

/* The generative loss of the three model families, loss = -mean(x) over all n entries (models/ratspn.py:184-191,
 * models/dgcspn.py loss, flows/models/base.py loss), accumulated in fp64, and its gradient grad_x[i] = -grad_out[0] / n:
 * one launch each instead of torch's mean + neg and their two backward nodes.  out / grad_out: one float on the device. */
int dpk_neg_mean_forward(const float *x, int64_t n, float *out, void *stream);
int dpk_neg_mean_backward(const float *grad_out, int64_t n, float *grad_x, void *stream);

/* sum and count of a vector of log-likelihoods in fp64 (the per-rank partial of
 * the mean-LL all-reduce): acc[0] += sum(ll), acc[1] += n.                   */
int dpk_ll_accumulate(const float *ll, int64_t n, double *acc, void *stream);

/* Last three stages of an eval-mode DGC-SPN in one launch (models/dgcspn.py:146-150): depthwise product + sum level
 * (geom5, sum_weight [Cout, C, OH5, OW5]), the last depthwise product (geom6, on the [Cout, OH5, OW5] map) and the
 * root (root_weight [K, Cout*OH6*OW6]); the largest activation map never reaches memory.
 * geom = {OH, OW, kh, kw, sh, sw, dh, dw, pad_top, pad_left} (host array of 10 ints) of a SpatialProductLayer
 * (layers/dgcspn.py:151-198).  DPK_EUNSUPPORTED outside <= 8 channels / <= 1024 final pixels / <= 4 taps.     */
int64_t dpk_spatial_sumprodroot_workspace_bytes(int32_t C, int32_t Cout, int32_t OH5, int32_t OW5, int32_t OH6,
                                                int32_t OW6, int32_t K);
/* Autograd of dpk_spatial_prodsum_forward (the training route of a depthwise level, C, Cout <= 8, <= 4 taps): the
 * product map (dgcspn.py:224-236) is recomputed from the taps instead of being kept from the forward; its gradient is
 * written to grad_prod [B,C,OH,OW] (caller scratch) and scattered to grad_in [B,C,H,W]; grad_weight [Cout,C,OH,OW].
 * out = the forward's output, g = its gradient.  Workspace of dpk_spatial_sum_workspace_bytes(C,Cout,OH,OW).
 * flags: DPK_FLAG_PARAMS_CACHED = the softmaxed weight tables of the forward call are still in the workspace.
 * DPK_EUNSUPPORTED outside that envelope (the caller chains the two layers' own backward entries).            */
int dpk_spatial_prodsum_backward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                 int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t dh, int32_t dw,
                                 int32_t pad_top, int32_t pad_left, const float *weight, int32_t Cout,
                                 const float *out, const float *g, float *grad_prod, float *grad_in,
                                 float *grad_weight, void *ws, int64_t ws_bytes, uint32_t flags, void *stream);

/* Workspace for a batch of B samples: the tables of ..._workspace_bytes plus, where the streaming kernel applies
 * (8 -> 8 channels, B >= 256, `in` 16-byte aligned), one (max, sum) pair per sample, class and compute wave for the
 * root's log-sum-exp.  With this much workspace the entry point streams the batch through LDS with the sum layer's
 * weights resident in registers; with only ..._workspace_bytes it runs the batch-independent kernel.               */
int64_t dpk_spatial_sumprodroot_workspace_bytes_batch(int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *geom5,
                                                      int32_t Cout, const int32_t *geom6, int32_t K);
int dpk_spatial_sumprodroot_forward(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *geom5,
                                    const float *sum_weight, int32_t Cout, const int32_t *geom6,
                                    const float *root_weight, int32_t K, float *out, void *ws, int64_t ws_bytes,
                                    uint32_t flags, void *stream);
/* The softmaxed-weight tables of up to 8 fused levels (dpk_spatial_prodsum_forward / dpk_spatial_sumprodroot_forward
 * workspaces; for the last one also the root's log-softmax rows, root_weight [K, M] or NULL) in ONE launch, after
 * which those entry points may be called with DPK_FLAG_PARAMS_CACHED (round 3: one table launch per DgcSpn.forward
 * instead of one per level).  sum_weight [Cout, C, OHW]; ws / ws_bytes: the workspace the level's entry point gets. */
typedef struct dpk_spatial_tables_args {
    const float *sum_weight;
    void *ws;
    int64_t ws_bytes;
    const float *root_weight;
    int32_t C, Cout, OHW, K, M;
} dpk_spatial_tables_args;
int dpk_spatial_tables(int32_t n, const dpk_spatial_tables_args *levels, void *stream);
   /* flags: DPK_FLAG_PARAMS_CACHED as above */

/* ---- vanilla (node-graph) SPN, flattened (BASELINE config 1) ---------------------------------------------
 * Bottom-up log-likelihood of deeprob/spn/algorithms/inference.py:37-58 (eval_bottom_up, evaluation.py:37-96;
 * node values clamped at -1e31 and kept in float32, inference.py:94-103) for a circuit given as arrays over
 * node ids 0..n_nodes-1 (the ids of the reference's JSON export, deeprob/spn/structure/io.py:133-175):
 *   order [n_nodes]   evaluation order, children before parents
 *   kind  [n_nodes]   0 Sum (node.py:116-117), 1 Product (node.py:152-153), 2 Bernoulli (leaf.py:182-186),
 *                     3 Categorical (:301-305), 4 Uniform (:475-479), 5 Gaussian (:553-557)
 *   inner nodes: arg0 = first entry in child_index / child_weight, arg1 = number of children;
 *                child_weight = the sum node's (linear) weight per child, unused for products
 *   leaves:      arg0 = input column; par0 / par1 (double) = log p, log1p(-p) | start, width | mean, stddev;
 *                Categorical: arg1 = first entry in cat_value / cat_logp, arg2 = number of categories
 * x [B, D] with NaN = marginalised.  out [B] = value of `root`; node_values (optional, [n_nodes, B]) receives
 * every node's value (return_results = True).  Without node_values the values live in an on-chip table of
 * n_slots rows: node_slot [n_nodes] = row of each node, child_slot (parallel to child_index) = row of each
 * child; a row may be reused once every parent of its node has been evaluated.  n_slots = 0 or > 256: the
 * circuit is evaluated through a workspace of dpk_flat_spn_workspace_bytes instead.                        */
int64_t dpk_flat_spn_workspace_bytes(int64_t B, int32_t n_nodes, int32_t n_slots);
int dpk_flat_spn_forward(const float *x, int64_t B, int32_t D, int32_t n_nodes, int32_t root, const int32_t *order,
                         const int32_t *kind, const int32_t *arg0, const int32_t *arg1, const int32_t *arg2,
                         const double *par0, const double *par1, const int32_t *child_index,
                         const float *child_weight, const int32_t *cat_value, const float *cat_logp,
                         int32_t n_slots, const int32_t *node_slot, const int32_t *child_slot, float *out,
                         float *node_values, void *ws, int64_t ws_bytes, void *stream);

/* ---- RealNVP-2D evaluation path (density and sampling directions, running statistics) -----------
 * Conditioner convolutions (torch/utils.py:86-121 WeightNormConv2d inside flows/layers/resnet.py:9-90 and
 * flows/layers/densenet.py): NCHW fp32, kernel 1x1 or 3x3, stride 1, "same" zero padding.
 * dpk_conv2d_prepare: w[co] = weight_g[co] * weight_v[co] / ||weight_v[co]|| (weight_g NULL: w = weight_v) packed as
 *   wpack[ci][ky*ks+kx][CoutPad] (dpk_conv2d_pack_floats floats); bn_* non-NULL additionally folds the eval-mode
 *   nn.BatchNorm2d that precedes the convolution into pre[0:Cin] = gamma/sqrt(var+eps), pre[Cin:2Cin] = beta - mean*pre[0:Cin].
 * dpk_conv2d_forward: out[b,co] = bias[co] + sum_ci w[co,ci] * f(in[b,ci]) (+ res[b,co]);
 *   f(v) = relu(pre_a v + pre_b) when pre != NULL (the BatchNorm2d + ReLU in front of the convolution), then * in_mask[H*W]
 *   when in_mask != NULL (CouplingLayer2d's mask * x, coupling.py:209-210); padding pixels contribute 0.
 *   in / res / out rows of a sample start in_bstride / res_bstride / out_bstride floats apart (channel slices of
 *   larger tensors: torch.chunk inputs, DenseNet concatenations); res == out is allowed (z += skip(x)).      */
int64_t dpk_conv2d_pack_floats(int32_t Cout, int32_t Cin, int32_t ks);
int dpk_conv2d_prepare(const float *weight_v, const float *weight_g, int32_t Cout, int32_t Cin, int32_t ks,
                       const float *bn_weight, const float *bn_bias, const float *bn_mean, const float *bn_var,
                       float bn_eps, float *wpack, float *pre, void *stream);
int dpk_conv2d_forward(const float *in, int64_t in_bstride, int64_t B, int32_t Cin, int32_t H, int32_t W,
                       const float *wpack, int32_t Cout, int32_t ks, const float *pre, const float *in_mask,
                       const float *bias, const float *res, int64_t res_bstride, float *out, int64_t out_bstride,
                       void *stream);
/* CouplingLayer2d.apply_backward (inverse = 0, coupling.py:181-226) / apply_forward (inverse = 1, :228-272) given the
 * conditioner output z [B, 2*Ch or Ch, H, W] (t | s; NICE mode: t only).  inv_mask [H*W] != NULL: checkerboard
 * coupling, Ch = C; NULL: channel-wise, Ch = C/2, x = [my | mx] (reverse: [mx | my]), mx copied through.
 * scale [Ch] = ScaledTanh weight.  ldj_out[b] = (ldj_in ? ldj_in[b] : 0) -/+ sum s.                         */
int dpk_coupling2d_transform(const float *x, const float *z, const float *scale, const float *inv_mask, int64_t B,
                             int32_t C, int32_t H, int32_t W, int32_t affine, int32_t reverse, int32_t inverse,
                             const float *ldj_in, float *out, float *ldj_out, void *stream);
/* BatchNormLayer2d with running statistics (flows/utils.py:186-222): weight / bias / mean / var [C].          */
int dpk_bn2d_bijector(const float *x, const float *weight, const float *bias, const float *mean, const float *var,
                      float eps, int64_t B, int32_t C, int32_t H, int32_t W, int32_t inverse, const float *ldj_in,
                      float *out, float *ldj_out, void *stream);
/* squeeze_depth2d (flows/utils.py:11-23) and RealNVP2d's one-hot permutation convolution followed by torch.chunk
 * (flows/models/realnvp.py:182-185), and their inverses (:26-38, :188-191): output channel o of the [B,4C,H/2,W/2]
 * tensor = in[b, table[o]>>2, 2h + ((table[o]>>1)&1), 2w + (table[o]&1)]; channels [0,Ca) live in *_a, the rest in *_b. */
int dpk_space_to_depth(const float *in, int64_t B, int32_t C, int32_t H, int32_t W, const int32_t *table, float *out_a,
                       int32_t Ca, float *out_b, void *stream);
int dpk_depth_to_space(const float *in_a, int32_t Ca, const float *in_b, int64_t B, int32_t C, int32_t H, int32_t W,
                       const int32_t *table, float *out, void *stream);

/* ---- RealNVP-2D training direction (csrc/flows2d_train.hip) ---------------------------------------------------
 * The reference trains RealNVP2d through ATen's autograd; these are the device passes our autograd nodes call.
 * Reductions over the batch are accumulated with fp64 atomics into buffers the caller zeroed.
 * dpk_channel_stats: sums[c] += sum_{b,h,w} x, sums[C+c] += sum x^2 (want_sq) -- the batch statistics of
 *   nn.BatchNorm2d in training mode (flows/layers/resnet.py:19-33) and of BatchNormLayer2d (flows/utils.py:190-198);
 *   also the bias gradient of a convolution (sum of the output gradient per channel).
 * dpk_channel_stats_backward: with mean = sum/n and var = sumsq/n - mean^2: dx (+)= (dmean + 2 dvar (x - mean)) / n.*/
int dpk_channel_stats(const float *x, int64_t x_bstride, int64_t B, int32_t C, int32_t H, int32_t W, int32_t want_sq,
                      double *sums, void *stream);
int dpk_channel_stats_backward(const float *x, int64_t x_bstride, int64_t B, int32_t C, int32_t H, int32_t W,
                               const float *mean, const float *dmean, const float *dvar, int32_t accumulate, float *dx,
                               void *stream);
/* nn.BatchNorm2d in training mode (flows/layers/resnet.py:19-33) folded into the operand map of the convolution behind
 * it: from the sums of dpk_channel_stats over n = B*H*W values per channel, pre = [gamma rstd | beta - mean gamma rstd],
 * stat = [mean | rstd]; running_mean / running_var (NULL: left alone) move by `momentum` as torch's module moves them
 * (unbiased variance).  dpk_bn2d_fold_backward: gradients dab [2C] of pre back to gamma, beta and to dstat =
 * [dmean | dvar], which dpk_channel_stats_backward (accumulate = 1: added to dx) carries to the input.             */
int dpk_bn2d_fold_train(const double *sums, int64_t n, int32_t C, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *pre, float *stat, void *stream);
int dpk_bn2d_fold_backward(const double *dab, int32_t C, const float *gamma, const float *stat, float *dgamma,
                           float *dbeta, float *dstat, void *stream);
/* out = ab[c] x + ab[C+c] (BatchNormLayer2d.apply_backward with the batch statistics folded into ab,
 * flows/utils.py:200-207).  Backward: g = dy * mask[h,w] (mask != NULL) * [ab[c] x + ab[C+c] > 0] (relu), dx = g ab[c],
 * dab[c] += sum g x, dab[C+c] += sum g; ab == NULL: dx = dy * mask only.  With relu / mask this is the backward of the
 * operand map dpk_conv2d_forward applies on load (BatchNorm2d + ReLU, CouplingLayer2d's mask * x).                  */
int dpk_channel_affine_forward(const float *x, int64_t B, int32_t C, int32_t H, int32_t W, const float *ab, float *out,
                               void *stream);
int dpk_channel_affine_backward(const float *x, int64_t x_bstride, const float *dy, int64_t B, int32_t C, int32_t H,
                                int32_t W, const float *ab, int32_t relu, const float *mask, float *dx, double *dab,
                                void *stream);
/* Weight gradient of dpk_conv2d_forward (F.conv2d's, torch/utils.py:117-121): dw[co][ci][ky][kx] += sum over samples
 * and pixels of dout[b,co] * f(in[b,ci]) shifted by the tap, f as in dpk_conv2d_forward (pre / in_mask).  dw zeroed by
 * the caller.  The input gradient is dpk_conv2d_forward itself on dout with the transposed, flipped weights.        */
int dpk_conv2d_backward_weight(const float *in, int64_t in_bstride, const float *dout, int64_t B, int32_t Cin,
                               int32_t Cout, int32_t H, int32_t W, int32_t ks, const float *pre, const float *in_mask,
                               float *dw, void *stream);
/* Backward of dpk_coupling2d_transform with inverse = 0 (flows/layers/coupling.py:181-226): gradients of (out, ldj)
 * = (dout, dldj[B] or NULL) w.r.t. x (the direct path only: the path through the conditioner is dz), z and the
 * ScaledTanh weight (dscale [Ch] fp64, zeroed by the caller).                                                        */
int dpk_coupling2d_transform_backward(const float *x, const float *z, const float *scale, const float *inv_mask,
                                      int64_t B, int32_t C, int32_t H, int32_t W, int32_t affine, int32_t reverse,
                                      const float *dout, const float *dldj, float *dx, float *dz, double *dscale,
                                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPROB_HIP_H */
