// DGC-SPN: geometry of a product layer and the streaming (LDS-staged) product + sum level kernels.
#pragma once
#include "common.h"

namespace dpk {

// Product layer geometry (kh x kw taps, dilation, stride, zero padding on the left / top)
struct ProdGeom {
    int C, H, W;        // input
    int OC, OH, OW;     // output
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int depthwise;
};

// Streaming route of the fused depthwise product + sum level for 8 -> 8 channels (dgcspn_stream.hip).
//   stream_prodsum_ok       : shape / batch inside the route's envelope
//   stream_prodsum_forward  : out[b,o,p] = logsumexp_c(sum_taps in[b,c,tap(p)] + log W[o,c,p]);  Wl / LW are the
//                             softmaxed weights and their logs, [Cout, C, OH*OW]
//   stream_sumprodroot_*    : the same level followed by the last product and the root layer; `partials` holds
//                             stream_sumprodroot_partial_bytes() bytes
bool stream_prodsum_ok(const ProdGeom &q, int Cout, int64_t B, const float *in);
//   in_pm / out_pm          : the input / output map is pixel-major ([B, H, W, 8], torch's channels_last) instead of [B, 8, H, W]
int stream_prodsum_forward(const float *in, int64_t B, const ProdGeom &q, const float *Wl, const float *LW, float *out,
                           hipStream_t st, bool in_pm = false, bool out_pm = false);
//   stream_leaf_prodsum_*   : the first level with the Gaussian leaf layer folded in: x is the image [B, Cx, H, W]
bool stream_leaf_prodsum_ok(const ProdGeom &q, int Cout, int64_t B, const float *x, int Cx);
int stream_leaf_prodsum_forward(const float *x, const float *loc, const float *scale, int Cx, int64_t B, const ProdGeom &q,
                                const float *Wl, const float *LW, float *out, hipStream_t st, bool out_pm);
int64_t stream_sumprodroot_partial_bytes(const ProdGeom &q5, int Cout, const ProdGeom &q6, int K, int64_t B);
int stream_sumprodroot_forward(const float *in, int64_t B, const ProdGeom &q5, const float *Wl, const float *LW,
                               const ProdGeom &q6, const float *LWr, int K, float *out, void *partials,
                               hipStream_t st, bool in_pm = false);

}  // namespace dpk
