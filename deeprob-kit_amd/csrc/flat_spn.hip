// Bottom-up log-likelihood of a vanilla (node-graph) SPN flattened into arrays -- BASELINE config 1.
//
// Replaces, for one batch, the node-by-node numpy pass of deeprob/spn/algorithms/evaluation.py:37-96 with the
// node functions of deeprob/spn/algorithms/inference.py:94-103 (every node value clamped at -1e31, float32):
//   Sum      node.py:116-117   logsumexp(children, b = weights)
//   Product  node.py:152-153   sum(children)
//   leaves   leaf.py:182-186 (Bernoulli), :301-305 (Categorical), :475-479 (Uniform), :553-557 (Gaussian);
//            0 where the input is NaN (marginalised)
// Layout: lane = sample, the node loop is uniform over the wave (node records come through scalar loads); the
// table of node values is [row][sample], so every access is one 256-byte row per wave.  In LDS (one 64-sample wave
// per work-group) a node's row is a slot that the host recycles after the node's last parent has been evaluated
// (a tree-like circuit needs a handful of slots, so the wave count per CU is not bounded by LDS); a circuit whose
// live set does not fit, or a caller that wants every node's value, uses a [n_nodes, B] buffer with row = node id.
#include "common.h"

namespace dpk {

enum FlatKind : int32_t { kFlatSum = 0, kFlatProduct = 1, kFlatBernoulli = 2, kFlatCategorical = 3, kFlatUniform = 4,
                          kFlatGaussian = 5 };

struct FlatSpnArgs {
    const float *x;
    int64_t B;
    int D, n_nodes, root;
    const int32_t *order, *kind, *arg0, *arg1, *arg2;
    const double *par0, *par1;
    const int32_t *child_index;   // node ids of the children (global table) or their slots (LDS table)
    const int32_t *node_slot;     // LDS table: row of each node
    const float *child_weight;
    const int32_t *cat_value;
    const float *cat_logp;
    float *table;       // global table (or nullptr when the LDS table is used)
    int64_t tstride;
    float *out;
};

constexpr float kFlatFloor = -1e31f;            // inference.py:103
constexpr int kFlatLdsSlots = 256;              // 256 rows x 64 samples x 4 B = 64 KB of LDS

template <bool kLds>
__global__ __launch_bounds__(64) void flat_spn_kernel(const FlatSpnArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const bool own = b < a.B;
    const int64_t bb = own ? b : a.B - 1;       // tail lanes shadow the last sample, global stores masked
    const float *xrow = a.x + bb * a.D;
    auto load = [&](int row) -> float { return kLds ? lds[row * 64 + lane] : a.table[row * a.tstride + bb]; };
    for (int t = 0; t < a.n_nodes; ++t) {
        const int i = a.order[t];
        const int kind = a.kind[i];
        float v;
        if (kind == kFlatSum) {
            // scipy logsumexp with weights: zero-weight terms are dropped, m = max of the rest (0 if not finite)
            const int c0 = a.arg0[i], nc = a.arg1[i];
            float m = -INFINITY;
            for (int j = 0; j < nc; ++j)
                if (a.child_weight[c0 + j] != 0.f) m = fmaxf(m, load(a.child_index[c0 + j]));
            if (!(fabsf(m) < INFINITY)) m = 0.f;
            float s = 0.f;
            for (int j = 0; j < nc; ++j) {
                const float w = a.child_weight[c0 + j];
                if (w != 0.f) s += w * expf(load(a.child_index[c0 + j]) - m);
            }
            v = logf(s) + m;
        } else if (kind == kFlatProduct) {
            const int c0 = a.arg0[i], nc = a.arg1[i];
            v = 0.f;
            for (int j = 0; j < nc; ++j) v += load(a.child_index[c0 + j]);
        } else {
            const float xv = xrow[a.arg0[i]];
            if (xv != xv) {
                v = 0.f;
            } else if (kind == kFlatBernoulli) {          // par0 = log p, par1 = log1p(-p)
                v = (xv == 1.f) ? (float)a.par0[i] : (xv == 0.f) ? (float)a.par1[i] : -INFINITY;
            } else if (kind == kFlatCategorical) {        // x.astype(int64): truncation toward zero
                const int c0 = a.arg1[i], nc = a.arg2[i];
                const long long cat = (long long)xv;
                v = -INFINITY;
                for (int j = 0; j < nc; ++j)
                    if ((long long)a.cat_value[c0 + j] == cat) v = a.cat_logp[c0 + j];
            } else if (kind == kFlatUniform) {            // par0 = start, par1 = width
                const double z = ((double)xv - a.par0[i]) / a.par1[i];
                v = (z >= 0.0 && z <= 1.0) ? (float)(-log(a.par1[i])) : -INFINITY;
            } else {                                      // Gaussian: par0 = mean, par1 = stddev
                const double z = ((double)xv - a.par0[i]) / a.par1[i];
                v = (float)(-0.5 * z * z - 0.91893853320467274178 - log(a.par1[i]));
            }
        }
        v = fmaxf(v, kFlatFloor);
        if (kLds)
            lds[a.node_slot[i] * 64 + lane] = v;
        else if (own)
            a.table[i * a.tstride + bb] = v;
        // a lane only ever reads its own column of the table: no barrier between nodes
    }
    if (own) a.out[b] = load(kLds ? a.node_slot[a.root] : a.root);
}

}  // namespace dpk

using namespace dpk;

extern "C" int64_t dpk_flat_spn_workspace_bytes(int64_t B, int32_t n_nodes, int32_t n_slots) {
    if (B < 0 || n_nodes <= 0 || n_slots < 0) return DPK_EINVAL;
    return (n_slots > 0 && n_slots <= kFlatLdsSlots) ? 0 : (int64_t)n_nodes * B * 4 + 256;
}

extern "C" int dpk_flat_spn_forward(const float *x, int64_t B, int32_t D, int32_t n_nodes, int32_t root,
                                    const int32_t *order, const int32_t *kind, const int32_t *arg0,
                                    const int32_t *arg1, const int32_t *arg2, const double *par0, const double *par1,
                                    const int32_t *child_index, const float *child_weight, const int32_t *cat_value,
                                    const float *cat_logp, int32_t n_slots, const int32_t *node_slot,
                                    const int32_t *child_slot, float *out, float *node_values, void *ws,
                                    int64_t ws_bytes, void *stream) {
    DPK_REQUIRE(B >= 0 && D > 0 && n_nodes > 0 && root >= 0 && root < n_nodes, DPK_EINVAL,
                "flat_spn_forward: bad sizes");
    DPK_REQUIRE(order && kind && arg0 && arg1 && arg2 && par0 && par1, DPK_EINVAL,
                "flat_spn_forward: null node arrays");
    if (B == 0) return DPK_OK;
    DPK_REQUIRE(x && out, DPK_EINVAL, "flat_spn_forward: null pointer");
    FlatSpnArgs a{};
    a.x = x; a.B = B; a.D = D; a.n_nodes = n_nodes; a.root = root;
    a.order = order; a.kind = kind; a.arg0 = arg0; a.arg1 = arg1; a.arg2 = arg2; a.par0 = par0; a.par1 = par1;
    a.child_index = child_index; a.child_weight = child_weight; a.cat_value = cat_value; a.cat_logp = cat_logp;
    a.out = out; a.tstride = B;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)cdiv(B, 64);
    if (node_values) {
        a.table = node_values;
        DPK_LAUNCH(flat_spn_kernel<false>, dim3(blocks), dim3(64), 0, st, a);
    } else if (n_slots > 0 && n_slots <= kFlatLdsSlots) {
        DPK_REQUIRE(node_slot && child_slot, DPK_EINVAL, "flat_spn_forward: slot arrays missing");
        a.node_slot = node_slot;
        a.child_index = child_slot;
        DPK_LAUNCH(flat_spn_kernel<true>, dim3(blocks), dim3(64), (size_t)n_slots * 256, st, a);
    } else {
        const int64_t need = dpk_flat_spn_workspace_bytes(B, n_nodes, n_slots);
        DPK_REQUIRE(ws && ws_bytes >= need, DPK_EWORKSPACE, "flat_spn_forward: workspace %lld < %lld",
                    (long long)ws_bytes, (long long)need);
        a.table = (float *)ws;
        DPK_LAUNCH(flat_spn_kernel<false>, dim3(blocks), dim3(64), 0, st, a);
    }
    DPK_CHECK_LAUNCH("flat_spn_forward");
    return DPK_OK;
}
