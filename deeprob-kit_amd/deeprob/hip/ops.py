"""Operator layer: one ``torch.autograd.Function`` per C-ABI forward/backward pair.

Each ``forward`` enqueues hand-written gfx950 kernels on the caller's current stream through
``ctypes``; each ``backward`` does the same with the matching ``*_backward`` entry point.  Nothing
in here computes with PyTorch ops.
"""
from typing import Optional, Tuple

import torch

from deeprob.hip import (
    load_library, check, ptr, stream_ptr, require_device_f32, Workspace, HipError, DPK_FLAG_STRUCT_CACHED,
    DPK_FLAG_UNIT_SCALE, DPK_FLAG_PARAMS_CACHED, DPK_FLAG_LL_SUM_SPREAD, LL_SPREAD, cached_tables_flag,
)


def _buffers_key(*tensors) -> tuple:
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) if t is not None else None for t in tensors)


class LeafContext:
    """Static description of a region-graph leaf layer handed to the kernels."""

    def __init__(self, in_features: int, regions: int, channels: int, dimension: int, depth: int = 0,
                 reps: int = 0, sums: int = 0, classes: int = 0):
        self.D, self.R, self.I, self.d = in_features, regions, channels, dimension
        self.depth, self.reps, self.S, self.C = depth, reps, sums, classes
        self.ws = Workspace()

    def workspace(self, device, mask, pad_mask, scale=None) -> Tuple[torch.Tensor, int]:
        lib = load_library()
        n = lib.dpk_ratspn_workspace_bytes(self.D, self.R, self.d, self.I, self.depth, self.reps,
                                           max(self.S, 1), max(self.C, 1))
        if n < 0:
            check(int(n), 'dpk_ratspn_workspace_bytes')
        buf = self.ws.get(n, device)
        key = _buffers_key(mask, pad_mask)
        flags = DPK_FLAG_STRUCT_CACHED if self.ws.struct_key == key else 0
        self.ws.struct_key = key
        # a frozen scale parameter is the reference's "scale == 1" configuration (optimize_scale=False);
        # only a hint: the kernels verify it on the device
        if scale is not None and not scale.requires_grad:
            flags |= DPK_FLAG_UNIT_SCALE
        return buf, flags


def _params_flag(lib, lctx: 'LeafContext', x_ptr, flags: int, tensors) -> int:
    """The cached-tables flag (``hip.cached_tables_flag``: checked on the device by default) when this fused call runs
    on the MFMA route and the tables in the workspace were built by an earlier call on that route from parameters at the
    same addresses with the same version counters (a structure rebuild invalidates them too)."""
    if not lib.dpk_ratspn_forward_on_mfma(x_ptr, lctx.D, lctx.depth, lctx.reps, lctx.I, lctx.S, lctx.C, 0, flags):
        lctx.ws.params_key = None
        return 0
    key = (_buffers_key(*tensors), lctx.ws.struct_key)
    if lctx.ws.params_key == key and (flags & DPK_FLAG_STRUCT_CACHED):
        return cached_tables_flag()
    lctx.ws.params_key = key
    return 0


def _pad_u8(pad_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    # torch.bool is one byte per element: reinterpret, no copy
    return None if pad_mask is None else pad_mask.contiguous().view(torch.uint8)


class GaussianLeafFn(torch.autograd.Function):
    """RegionGraphLayer.forward with Normal leaves (reference: deeprob/spn/layers/ratspn.py:87-108)."""

    @staticmethod
    def forward(ctx, x, loc, scale, mask, pad_mask, lctx: LeafContext):
        lib = load_library()
        x = require_device_f32(x, 'x')
        loc_c, scale_c = require_device_f32(loc, 'loc'), require_device_f32(scale, 'scale')
        B = x.shape[0]
        out = torch.empty((B, lctx.R, lctx.I), dtype=torch.float32, device=x.device)
        pad = _pad_u8(pad_mask)
        ws, flags = lctx.workspace(x.device, mask, pad_mask, scale)
        if lib.dpk_gaussian_leaf_forward_on_mfma(ptr(x), ptr(out), lctx.D, lctx.R, lctx.I, lctx.d, flags):
            # MFMA route: its parameter tables survive between calls while loc / scale are unchanged
            key = (_buffers_key(loc_c, scale_c), lctx.ws.struct_key)
            if lctx.ws.params_key == key and (flags & DPK_FLAG_STRUCT_CACHED):
                flags |= cached_tables_flag()
            lctx.ws.params_key = key
        else:
            lctx.ws.params_key = None
        check(lib.dpk_gaussian_leaf_forward(ptr(x), B, lctx.D, ptr(mask), ptr(pad), ptr(loc_c), ptr(scale_c),
                                            lctx.R, lctx.I, lctx.d, ptr(out), ptr(ws), ws.numel(), flags,
                                            stream_ptr(x.device)), 'dpk_gaussian_leaf_forward')
        ctx.save_for_backward(x, loc_c, scale_c, mask, pad_mask)
        ctx.lctx = lctx
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, loc, scale, mask, pad_mask = ctx.saved_tensors
        lctx = ctx.lctx
        g = require_device_f32(g, 'grad')
        need_x, need_loc, need_scale = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gx = torch.empty_like(x) if need_x else None
        gloc = torch.empty_like(loc) if need_loc else None
        gscale = torch.empty_like(scale) if need_scale else None
        ws, flags = lctx.workspace(x.device, mask, pad_mask)
        check(lib.dpk_gaussian_leaf_backward(ptr(x), ptr(g), x.shape[0], lctx.D, ptr(mask),
                                             ptr(_pad_u8(pad_mask)), ptr(loc), ptr(scale), lctx.R, lctx.I,
                                             lctx.d, ptr(gloc), ptr(gscale), ptr(gx), ptr(ws), ws.numel(),
                                             flags, stream_ptr(x.device)), 'dpk_gaussian_leaf_backward')
        return gx, gloc, gscale, None, None, None


class BernoulliLeafFn(torch.autograd.Function):
    """RegionGraphLayer.forward with Bernoulli leaves (reference: ratspn.py:87-108, :216-247)."""

    @staticmethod
    def forward(ctx, x, logits, mask, pad_mask, lctx: LeafContext):
        lib = load_library()
        x = require_device_f32(x, 'x')
        logits_c = require_device_f32(logits, 'logits')
        B = x.shape[0]
        out = torch.empty((B, lctx.R, lctx.I), dtype=torch.float32, device=x.device)
        ws, flags = lctx.workspace(x.device, mask, pad_mask)
        check(lib.dpk_bernoulli_leaf_forward(ptr(x), B, lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)),
                                             ptr(logits_c), lctx.R, lctx.I, lctx.d, ptr(out), ptr(ws),
                                             ws.numel(), flags, stream_ptr(x.device)),
              'dpk_bernoulli_leaf_forward')
        ctx.save_for_backward(x, logits_c, mask, pad_mask)
        ctx.lctx = lctx
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, logits, mask, pad_mask = ctx.saved_tensors
        lctx = ctx.lctx
        g = require_device_f32(g, 'grad')
        glog = torch.empty_like(logits) if ctx.needs_input_grad[1] else None
        ws, flags = lctx.workspace(x.device, mask, pad_mask)
        check(lib.dpk_bernoulli_leaf_backward(ptr(x), ptr(g), x.shape[0], lctx.D, ptr(mask),
                                              ptr(_pad_u8(pad_mask)), ptr(logits), lctx.R, lctx.I, lctx.d,
                                              ptr(glog), ptr(ws), ws.numel(), flags, stream_ptr(x.device)),
              'dpk_bernoulli_leaf_backward')
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            check(lib.dpk_bernoulli_leaf_backward_input(ptr(x), ptr(g), x.shape[0], lctx.D, ptr(mask),
                                                        ptr(_pad_u8(pad_mask)), ptr(logits), lctx.R, lctx.I, lctx.d,
                                                        ptr(gx), ptr(ws), ws.numel(), flags | DPK_FLAG_STRUCT_CACHED,
                                                        stream_ptr(x.device)), 'dpk_bernoulli_leaf_backward_input')
        return gx, glog, None, None, None


def draw_seed() -> int:
    """A fresh 62-bit dropout seed from torch's CPU generator (so torch.manual_seed makes training reproducible)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class LeafDropoutFn(torch.autograd.Function):
    """Training-mode RegionGraphLayer.forward with input dropout (reference: ratspn.py:94-108): the element-wise
    Bernoulli(p) decisions are a counter-based hash of (seed, element), replayed by the backward kernels."""

    @staticmethod
    def forward(ctx, x, p0, p1, mask, pad_mask, lctx: LeafContext, dist: int, rate: float, seed: int):
        lib = load_library()
        x = require_device_f32(x, 'x')
        p0_c = require_device_f32(p0, 'loc' if dist == 0 else 'logits')
        p1_c = require_device_f32(p1, 'scale') if p1 is not None else None
        B = x.shape[0]
        out = torch.empty((B, lctx.R, lctx.I), dtype=torch.float32, device=x.device)
        check(lib.dpk_leaf_forward_dropout(dist, ptr(x), B, lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)), ptr(p0_c),
                                           ptr(p1_c), lctx.R, lctx.I, lctx.d, float(rate), seed, ptr(out),
                                           stream_ptr(x.device)), 'dpk_leaf_forward_dropout')
        ctx.save_for_backward(x, p0_c, p1_c, mask, pad_mask)
        ctx.meta = (lctx, dist, float(rate), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, p0, p1, mask, pad_mask = ctx.saved_tensors
        lctx, dist, rate, seed = ctx.meta
        g = require_device_f32(g, 'grad')
        if dist == 1 and ctx.needs_input_grad[0]:
            raise NotImplementedError("d/dx through Bernoulli leaves is not defined by the reference use")
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g0 = torch.empty_like(p0) if ctx.needs_input_grad[1] else None
        g1 = torch.empty_like(p1) if (p1 is not None and ctx.needs_input_grad[2]) else None
        ws, flags = lctx.workspace(x.device, mask, pad_mask)
        check(lib.dpk_leaf_backward_dropout(dist, ptr(x), ptr(g), x.shape[0], lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)),
                                            ptr(p0), ptr(p1), lctx.R, lctx.I, lctx.d, rate, seed, ptr(g0), ptr(g1),
                                            ptr(gx), ptr(ws), ws.numel(), flags, stream_ptr(x.device)),
              'dpk_leaf_backward_dropout')
        return gx, g0, g1, None, None, None, None, None, None


class DropoutFillFn(torch.autograd.Function):
    """Sum-layer input dropout (reference: ratspn.py:371-372, dgcspn.py:297-298): dropped inputs become -inf;
    their gradient is zero."""

    @staticmethod
    def forward(ctx, x, rate: float, seed: int):
        lib = load_library()
        x = require_device_f32(x, 'x')
        out = torch.empty_like(x)
        check(lib.dpk_dropout_fill(ptr(x), x.numel(), float(rate), seed, float('-inf'), ptr(out), stream_ptr(x.device)),
              'dpk_dropout_fill')
        ctx.meta = (float(rate), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        g = require_device_f32(g, 'grad')
        rate, seed = ctx.meta
        gx = torch.empty_like(g)
        check(lib.dpk_dropout_fill(ptr(g), g.numel(), rate, seed, 0.0, ptr(gx), stream_ptr(g.device)),
              'dpk_dropout_fill')
        return gx, None, None


class ProductFn(torch.autograd.Function):
    """ProductLayer.forward (reference: ratspn.py:272-286)."""

    @staticmethod
    def forward(ctx, x):
        lib = load_library()
        x = require_device_f32(x, 'x')
        B, R, N = x.shape
        out = torch.empty((B, R // 2, N * N), dtype=torch.float32, device=x.device)
        check(lib.dpk_product_forward(ptr(x), B, R, N, ptr(out), stream_ptr(x.device)), 'dpk_product_forward')
        ctx.shape = (B, R, N)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        B, R, N = ctx.shape
        g = require_device_f32(g, 'grad')
        gx = torch.empty((B, R, N), dtype=torch.float32, device=g.device)
        check(lib.dpk_product_backward(ptr(g), B, R, N, ptr(gx), stream_ptr(g.device)), 'dpk_product_backward')
        return gx


def _sum_ws(ws: Workspace, B, P, N, S, device):
    lib = load_library()
    n = lib.dpk_sum_workspace_bytes(B, P, N, S)
    if n < 0:
        check(int(n), 'dpk_sum_workspace_bytes')
    ws.params_key = None   # the per-layer route lays its own tables over the folded route's cached ones
    return ws.get(n, device)


class SumFn(torch.autograd.Function):
    """SumLayer.forward in eval mode (reference: ratspn.py:363-378)."""

    @staticmethod
    def forward(ctx, x, weight, ws: Workspace):
        lib = load_library()
        x = require_device_f32(x, 'x')
        w = require_device_f32(weight, 'weight')
        B, P, N = x.shape
        S = w.shape[1]
        out = torch.empty((B, P, S), dtype=torch.float32, device=x.device)
        buf = _sum_ws(ws, 0, P, N, S, x.device)      # (B sizes the backward's residual segment only)
        check(lib.dpk_sum_forward(ptr(x), ptr(w), B, P, N, S, ptr(out), ptr(buf), buf.numel(),
                                  stream_ptr(x.device)), 'dpk_sum_forward')
        ctx.save_for_backward(x, w, out)
        ctx.ws = ws
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, w, out = ctx.saved_tensors
        g = require_device_f32(g, 'grad')
        B, P, N = x.shape
        S = w.shape[1]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        buf = _sum_ws(ctx.ws, B, P, N, S, x.device)
        check(lib.dpk_sum_backward(ptr(x), ptr(w), ptr(out), ptr(g), B, P, N, S, ptr(gx), ptr(gw), ptr(buf),
                                   buf.numel(), stream_ptr(x.device)), 'dpk_sum_backward')
        return gx, gw, None


class RootFn(torch.autograd.Function):
    """RootLayer.forward (reference: ratspn.py:446-458)."""

    @staticmethod
    def forward(ctx, x, weight, ws: Workspace):
        lib = load_library()
        x = require_device_f32(x, 'x')
        w = require_device_f32(weight, 'weight')
        B = x.shape[0]
        M, C = w.shape[1], w.shape[0]
        x2 = x.reshape(B, M)
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        buf = _sum_ws(ws, 0, 1, M, C, x.device)
        check(lib.dpk_root_forward(ptr(x2), ptr(w), B, M, C, ptr(out), ptr(buf), buf.numel(),
                                   stream_ptr(x.device)), 'dpk_root_forward')
        ctx.save_for_backward(x2, w, out)
        ctx.ws = ws
        ctx.in_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x2, w, out = ctx.saved_tensors
        g = require_device_f32(g, 'grad')
        B, M = x2.shape
        C = w.shape[0]
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        buf = _sum_ws(ctx.ws, B, 1, M, C, x2.device)
        check(lib.dpk_root_backward(ptr(x2), ptr(w), ptr(out), ptr(g), B, M, C, ptr(gx), ptr(gw), ptr(buf),
                                    buf.numel(), stream_ptr(x2.device)), 'dpk_root_backward')
        return (gx.reshape(ctx.in_shape) if gx is not None else None), gw, None


class ProdSumFn(torch.autograd.Function):
    """ProductLayer + SumLayer (or + RootLayer) as ONE autograd node, the training route of a RAT-SPN level
    (reference: ratspn.py:272-286 chained with :363-378 / :446-458): the forward is the folded evaluation kernel (on
    the matrix cores for 8 / 16 nodes per region), the ``[B, P, N^2]`` product tensor is neither written nor kept; the
    backward recomputes it and runs the two layers' own backward kernels."""

    @staticmethod
    def forward(ctx, x, weight, ws: Workspace, root: bool):
        out = prodroot_forward(x, weight, ws) if root else prodsum_forward(x, weight, ws)
        if out is None:
            raise HipError("ProdSumFn: shape outside the folded kernels (checked by prodsum_autograd)")
        ctx.save_for_backward(x, weight, out)
        ctx.ws, ctx.root = ws, root
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        gx, gw = _prodsum_backward(x, w, out, g, ctx.ws, ctx.root, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None


_LEVEL_BACKWARD = [True]     # (tests and A/B measurements switch the single-launch level backward off here)


def _prodsum_backward(x, w, out, g, ws: Workspace, root: bool, need_gx: bool, need_gw: bool):
    """Backward of a folded level from its saved (input, weight, output): the product tensor recomputed, then the Sum /
    Root layer's and the ProductLayer's own backward kernels.  ``(x, out)`` may carry a common per-sample shift (the
    training forward's relative tensors): the kernels only use ``in - out``."""
    lib = load_library()
    g = require_device_f32(g, 'grad')
    B, R, N = x.shape
    P, st = R // 2, stream_ptr(x.device)
    if _LEVEL_BACKWARD[0] and x.is_contiguous() and out.is_contiguous():
        # the level's backward in one launch (+ the Jacobian of the weight gradient); DPK_EUNSUPPORTED: the chain below
        S = w.shape[0] if root else w.shape[1]
        gx = torch.empty_like(x) if need_gx else None
        gw = torch.empty_like(w) if need_gw else None
        buf = _sum_ws(ws, 0, P, N * N, S, x.device)
        rc = lib.dpk_prodsum_backward(ptr(x), ptr(w), ptr(out), ptr(g), B, R, N, S, int(root), ptr(gx), ptr(gw), ptr(buf),
                                      buf.numel(), st)
        if rc != -4:
            check(rc, 'dpk_prodsum_backward')
            return gx, gw
    prod = torch.empty((B, P, N * N), dtype=torch.float32, device=x.device)
    check(lib.dpk_product_forward(ptr(x), B, R, N, ptr(prod), st), 'dpk_product_forward')
    gprod = torch.empty_like(prod) if need_gx else None
    gw = torch.empty_like(w) if need_gw else None
    if root:
        M, C = P * N * N, w.shape[0]
        buf = _sum_ws(ws, B, 1, M, C, x.device)
        check(lib.dpk_root_backward(ptr(prod), ptr(w), ptr(out), ptr(g), B, M, C, ptr(gprod), ptr(gw), ptr(buf),
                                    buf.numel(), st), 'dpk_root_backward')
    else:
        S = w.shape[1]
        buf = _sum_ws(ws, B, P, N * N, S, x.device)
        check(lib.dpk_sum_backward(ptr(prod), ptr(w), ptr(out), ptr(g), B, P, N * N, S, ptr(gprod), ptr(gw), ptr(buf),
                                   buf.numel(), st), 'dpk_sum_backward')
    gx = None
    if gprod is not None:
        gx = torch.empty_like(x)
        check(lib.dpk_product_backward(ptr(gprod), B, R, N, ptr(gx), st), 'dpk_product_backward')
    return gx, gw


class RatSpnTrainFn(torch.autograd.Function):
    """RatSpn.forward of a training step as ONE autograd node (reference: models/ratspn.py:105-122 under autograd): the
    forward is the single-launch kernel of the evaluation path, which also writes the leaf and sum layer outputs the
    backward needs (``dpk_ratspn_forward_train``); the backward chains the layers' own backward kernels.  Depth 2,
    Gaussian leaves with a frozen (unit) scale, no dropout, no gradient with respect to the evidence."""

    @staticmethod
    def forward(ctx, x, loc, scale, w0, wr, mask, pad_mask, lctx: LeafContext, leaf_lctx: LeafContext, ws0: Workspace,
                wsr: Workspace):
        lib = load_library()
        x = require_device_f32(x, 'x')
        loc_c, scale_c = require_device_f32(loc, 'loc'), require_device_f32(scale, 'scale')
        w0_c, wr_c = require_device_f32(w0, 'sum weight'), require_device_f32(wr, 'root weight')
        B, dev = x.shape[0], x.device
        out = torch.empty((B, lctx.C), dtype=torch.float32, device=dev)
        out_rel = torch.empty_like(out)
        leaf_rel = torch.empty((B, lctx.R, lctx.I), dtype=torch.float32, device=dev)
        sum_rel = torch.empty((B, 2 * lctx.reps, lctx.S), dtype=torch.float32, device=dev)
        ws, flags = lctx.workspace(dev, mask, pad_mask, scale)
        flags |= _params_flag(lib, lctx, ptr(x), flags, [loc_c, scale_c, w0_c, wr_c])
        rc = lib.dpk_ratspn_forward_train(ptr(x), B, lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)), ptr(loc_c), ptr(scale_c),
                                          ptr(w0_c), ptr(wr_c), lctx.depth, lctx.reps, lctx.I, lctx.S, lctx.C, ptr(out),
                                          ptr(leaf_rel), ptr(sum_rel), ptr(out_rel), ptr(ws), ws.numel(), flags,
                                          stream_ptr(dev))
        if rc:
            lctx.ws.params_key = None
            if rc == -4:  # DPK_EUNSUPPORTED
                lctx.ws.struct_key = None
                raise _TrainForwardUnsupported()
        check(rc, 'dpk_ratspn_forward_train')
        ctx.save_for_backward(x, loc_c, scale_c, w0_c, wr_c, mask, pad_mask, leaf_rel, sum_rel, out_rel)
        ctx.leaf_lctx, ctx.ws0, ctx.wsr = leaf_lctx, ws0, wsr
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, loc, scale, w0, wr, mask, pad_mask, leaf_rel, sum_rel, out_rel = ctx.saved_tensors
        need_loc, need_w0, need_wr = ctx.needs_input_grad[1], ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        need_below = need_loc or need_w0
        gsum, gwr = _prodsum_backward(sum_rel, wr, out_rel, g, ctx.wsr, True, need_below, need_wr)
        gleaf, gw0, gloc = None, None, None
        if need_below:
            gleaf, gw0 = _prodsum_backward(leaf_rel, w0, sum_rel, gsum, ctx.ws0, False, need_loc, need_w0)
        if need_loc:
            lctx = ctx.leaf_lctx
            gloc = torch.empty_like(loc)
            ws, flags = lctx.workspace(x.device, mask, pad_mask)
            check(lib.dpk_gaussian_leaf_backward(ptr(x), ptr(gleaf), x.shape[0], lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)),
                                                 ptr(loc), ptr(scale), lctx.R, lctx.I, lctx.d, ptr(gloc), None, None,
                                                 ptr(ws), ws.numel(), flags, stream_ptr(x.device)),
                  'dpk_gaussian_leaf_backward')
        return (None, gloc, None, gw0, gwr) + (None,) * 6


class _TrainForwardUnsupported(Exception):
    pass


def ratspn_forward_train(x, mask, pad_mask, loc, scale, sum_weight, root_weight, lctx: LeafContext,
                         leaf_lctx: LeafContext, ws0: Workspace, wsr: Workspace) -> Optional[torch.Tensor]:
    """``RatSpnTrainFn`` or None when the model / batch is outside the single-launch training forward."""
    try:
        return RatSpnTrainFn.apply(x, loc, scale, sum_weight, root_weight, mask, pad_mask, lctx, leaf_lctx, ws0, wsr)
    except _TrainForwardUnsupported:
        return None


def prodsum_autograd(x: torch.Tensor, weight: torch.Tensor, ws: Workspace, root: bool = False) -> Optional[torch.Tensor]:
    """The folded level with an autograd graph; None outside the folded kernels' envelope (more than 32 nodes per
    region): the caller chains the two layers."""
    if x.dim() != 3 or x.shape[2] > 32 or x.shape[1] % 2:
        return None
    return ProdSumFn.apply(x, weight, ws, root)


def ratspn_forward_fused(x, mask, pad_mask, loc, scale, sum_weights, root_weight, lctx: LeafContext,
                         ll_acc: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """RatSpn.forward in one launch (reference: deeprob/spn/models/ratspn.py:105-122).

    Returns None when the shape is outside what the fused kernel is built for (the caller then
    chains the per-layer operators); raises on any other failure.
    """
    lib = load_library()
    x = require_device_f32(x, 'x')
    loc_c, scale_c = require_device_f32(loc, 'loc'), require_device_f32(scale, 'scale')
    sw = [require_device_f32(w, 'sum weight') for w in sum_weights]
    rw = require_device_f32(root_weight, 'root weight')
    B = x.shape[0]
    out = torch.empty((B, lctx.C), dtype=torch.float32, device=x.device)
    ws, flags = lctx.workspace(x.device, mask, pad_mask, scale)
    flags |= _params_flag(lib, lctx, ptr(x), flags, [loc_c, scale_c] + sw + [rw])
    if ll_acc is not None and ll_acc.numel() > LL_SPREAD:      # a spread slot: sixteen partial sums, then the count
        flags |= DPK_FLAG_LL_SUM_SPREAD
    rc = lib.dpk_ratspn_forward(ptr(x), B, lctx.D, ptr(mask), ptr(_pad_u8(pad_mask)), ptr(loc_c), ptr(scale_c),
                                ptr(sw[0]) if len(sw) > 0 else None, ptr(sw[1]) if len(sw) > 1 else None,
                                ptr(rw), lctx.depth, lctx.reps, lctx.I, lctx.S, lctx.C, ptr(out), None,
                                ptr(ll_acc), ptr(ws), ws.numel(), flags, stream_ptr(x.device))
    if rc == -4:  # DPK_EUNSUPPORTED
        lctx.ws.struct_key = None
        lctx.ws.params_key = None
        return None
    if rc:
        lctx.ws.params_key = None
    check(rc, 'dpk_ratspn_forward')
    return out


class FusedForwardPlan:
    """A bound ``dpk_ratspn_forward`` call for one resident input buffer: the pointer / size validation and the
    argument marshalling of :func:`ratspn_forward_fused` are done once, every later step is a single C call
    (host cost per step matters when the kernel itself takes ~0.1 ms).

    The plan reads the live parameter storage like every other call (in-place updates are seen) but it pins the
    *addresses*: it is valid while the input buffer, the parameters and the module's workspace stay allocated
    where they are -- ``valid()`` re-checks that cheaply.  The output tensor is reused by every ``run``.
    """

    def __init__(self, x, mask, pad_mask, loc, scale, sum_weights, root_weight, lctx: LeafContext,
                 static_params: bool = False):
        # static_params: the caller guarantees that the parameter BYTES do not change while the plan lives (a frozen
        # model serving / evaluating): later runs pass DPK_FLAG_PARAMS_CACHED and skip the device-side fingerprint
        self.static_params = bool(static_params)
        self.lib = load_library()
        x = require_device_f32(x, 'x')
        self.tensors = [x, mask, _pad_u8(pad_mask), require_device_f32(loc, 'loc'), require_device_f32(scale, 'scale')]
        self.tensors += [require_device_f32(w, 'sum weight') for w in sum_weights]
        self.tensors.append(require_device_f32(root_weight, 'root weight'))
        sw = self.tensors[5:-1]
        B = x.shape[0]
        self.device = x.device
        self.out = torch.empty((B, lctx.C), dtype=torch.float32, device=x.device)
        ws, flags = lctx.workspace(x.device, mask, pad_mask, scale)
        self.lctx, self.ws = lctx, ws
        self.args = [ptr(x), B, lctx.D, ptr(mask), ptr(self.tensors[2]), ptr(self.tensors[3]), ptr(self.tensors[4]),
                     ptr(sw[0]) if len(sw) > 0 else None, ptr(sw[1]) if len(sw) > 1 else None,
                     ptr(self.tensors[-1]), lctx.depth, lctx.reps, lctx.I, lctx.S, lctx.C, ptr(self.out), None,
                     None, ptr(ws), ws.numel(), flags, None]
        self.ptrs = self._addresses()
        self.struct_key = lctx.ws.struct_key
        self._struct_tensors = (mask, pad_mask)
        self.params = self.tensors[3:]
        # the first call also builds (or re-validates) the structure tables and tells whether the shape is covered
        self.args[-1] = stream_ptr(self.device)
        self.args[-2] = flags | _params_flag(self.lib, lctx, self.args[0], flags, self.params)
        rc = self.lib.dpk_ratspn_forward(*self.args)
        self.supported = rc != -4
        if self.supported:
            check(rc, 'dpk_ratspn_forward')
            self.base_flags = flags | DPK_FLAG_STRUCT_CACHED
        else:
            lctx.ws.struct_key = None
            lctx.ws.params_key = None

    def _addresses(self):
        return tuple(t.data_ptr() for t in self.tensors if t is not None) + (self.ws.data_ptr(),)

    def valid(self) -> bool:
        # addresses AND the structure buffers' version counters (an in-place load_state_dict of another region
        # graph keeps the addresses): the module's cached structure key must still be the one this plan bound
        return (self.lctx.ws.buf is self.ws and self._addresses() == self.ptrs
                and self.lctx.ws.struct_key == self.struct_key
                and _buffers_key(*self._struct_tensors) == self.struct_key)

    def run(self, ll_acc: Optional[torch.Tensor] = None) -> torch.Tensor:
        args = self.args
        args[17] = None if ll_acc is None else ll_acc.data_ptr()
        args[-1] = torch.cuda.current_stream(self.device).cuda_stream
        pf = _params_flag(self.lib, self.lctx, args[0], self.base_flags, self.params)
        if pf and self.static_params:
            pf = DPK_FLAG_PARAMS_CACHED
        args[-2] = self.base_flags | pf | (DPK_FLAG_LL_SUM_SPREAD if ll_acc is not None and ll_acc.numel() > LL_SPREAD else 0)
        rc = self.lib.dpk_ratspn_forward(*args)
        if rc:
            check(rc, 'dpk_ratspn_forward')
        return self.out


def ll_accumulate(ll: torch.Tensor, acc: torch.Tensor):
    """acc[0] += sum(ll) (fp64), acc[1] += ll.numel(); a spread slot (17 doubles, hip.LL_SPREAD) takes the sum in its last
    partial and the count behind it -- the same two adjacent doubles."""
    lib = load_library()
    ll = require_device_f32(ll, 'll')
    if acc.numel() > LL_SPREAD:
        acc = acc.view(-1)[LL_SPREAD - 1:LL_SPREAD + 1]
    assert acc.dtype == torch.float64 and acc.numel() == 2 and acc.is_cuda
    check(lib.dpk_ll_accumulate(ptr(ll), ll.numel(), ptr(acc), stream_ptr(ll.device)), 'dpk_ll_accumulate')


def _prodsum_ws(ws: Workspace, R, N, S, device):
    lib = load_library()
    n = lib.dpk_prodsum_workspace_bytes(R, N, S)
    if n < 0:
        check(int(n), 'dpk_prodsum_workspace_bytes')
    return ws.get(n, device)


def _upper_tables_flag(ws: Workspace, route: str, w: torch.Tensor) -> int:
    """DPK_FLAG_PARAMS_CACHED when the layer's workspace still holds the softmax rows and MFMA fragments that the
    same folded entry point built from this very weight tensor (address, shape, version counter).  The per-layer
    operators sharing the workspace drop the key (_sum_ws), so does a replaced buffer (Workspace.get)."""
    key = (route, w.data_ptr(), tuple(w.shape), w._version)
    if ws.params_key == key:
        return cached_tables_flag()
    ws.params_key = key
    return 0


def upper_tables_pair(sum_weight: torch.Tensor, ws0: Workspace, R0: int, N0: int, root_weight: torch.Tensor, ws1: Workspace,
                      R1: int, N1: int, device) -> bool:
    """The tables (softmax rows + MFMA fragments) of a depth-2 model's sum layer and root layer in ONE launch, written into
    the two layers' workspaces; True when done -- the caller then passes ``tables_current=True`` to ``prodsum_forward`` /
    ``prodroot_forward``.  The default mode rebuilds these tables on every call (a write through ``weight.data`` must be
    seen): two launches before, one now."""
    lib = load_library()
    w0, w1 = require_device_f32(sum_weight, 'sum weight'), require_device_f32(root_weight, 'root weight')
    S0, C = w0.shape[1], w1.shape[0]
    b0, b1 = _prodsum_ws(ws0, R0, N0, S0, device), _prodsum_ws(ws1, R1, N1, C, device)
    key0 = ('prodsum', w0.data_ptr(), tuple(w0.shape), w0._version)
    key1 = ('prodroot', w1.data_ptr(), tuple(w1.shape), w1._version)
    if cached_tables_flag() == DPK_FLAG_PARAMS_CACHED and ws0.params_key == key0 and ws1.params_key == key1:
        return False     # (trusting the version counters and nothing moved: the layers skip their tables themselves)
    rc = lib.dpk_upper_tables_pair(ptr(w0), R0, N0, S0, ptr(b0), b0.numel(), ptr(w1), R1, N1, C, ptr(b1), b1.numel(),
                                   stream_ptr(device))
    if rc == -4:
        return False
    check(rc, 'dpk_upper_tables_pair')
    ws0.params_key, ws1.params_key = key0, key1
    return True


def prodsum_forward(x: torch.Tensor, weight: torch.Tensor, ws: Workspace, tables_current: bool = False) -> Optional[torch.Tensor]:
    """ProductLayer + SumLayer in one launch, eval mode, no autograd graph (reference: ratspn.py:272-286, :363-378).
    x [B,R,N], weight [R/2,S,N*N] -> [B,R/2,S]; None when N is beyond what the kernel is built for."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    w = require_device_f32(weight, 'weight')
    B, R, N = x.shape
    S = w.shape[1]
    out = torch.empty((B, R // 2, S), dtype=torch.float32, device=x.device)
    buf = _prodsum_ws(ws, R, N, S, x.device)
    flags = DPK_FLAG_PARAMS_CACHED if tables_current else _upper_tables_flag(ws, 'prodsum', w)
    rc = lib.dpk_prodsum_forward(ptr(x), ptr(w), B, R, N, S, ptr(out), ptr(buf), buf.numel(), flags,
                                 stream_ptr(x.device))
    if rc:
        ws.params_key = None
    if rc == -4:
        return None
    check(rc, 'dpk_prodsum_forward')
    return out


def prodroot_forward(x: torch.Tensor, weight: torch.Tensor, ws: Workspace, tables_current: bool = False) -> Optional[torch.Tensor]:
    """Last ProductLayer + RootLayer in one launch (reference: ratspn.py:272-286, :446-458). x [B,R,N],
    weight [C,(R/2)*N*N] -> [B,C]."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    w = require_device_f32(weight, 'weight')
    B, R, N = x.shape
    C = w.shape[0]
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    buf = _prodsum_ws(ws, R, N, C, x.device)
    flags = DPK_FLAG_PARAMS_CACHED if tables_current else _upper_tables_flag(ws, 'prodroot', w)
    rc = lib.dpk_prodroot_forward(ptr(x), ptr(w), B, R, N, C, ptr(out), ptr(buf), buf.numel(), flags,
                                  stream_ptr(x.device))
    if rc:
        ws.params_key = None
    if rc == -4:
        return None
    check(rc, 'dpk_prodroot_forward')
    return out


def ratspn_topdown(mode: int, dist: int, n_samples: int, lctx: LeafContext, x: Optional[torch.Tensor], y: Optional[torch.Tensor],
                   acts, logws, src: torch.Tensor, p0: torch.Tensor, p1: Optional[torch.Tensor], seed: int = 0,
                   want_choice: bool = False):
    """RatSpn.mpe (mode 0) / RatSpn.sample (mode 1) top-down in one launch (reference: deeprob/spn/models/ratspn.py:124-182
    and the layers' mpe / sample methods).  ``acts``: [leaf output, sum level 1 output, ...] (mode 0); ``logws``: log-softmax
    weights of the sum levels 1 .. depth-1, then of the root.  Returns the completed / generated ``[B, D]`` tensor (and the
    ``[B, 1 + 2^depth]`` choices when asked)."""
    import ctypes
    lib = load_library()
    depth = lctx.depth
    device = p0.device
    B = int(n_samples)
    keep = []

    def dev(t, name):
        t = require_device_f32(t, name)
        keep.append(t)
        return t

    act_arr = (ctypes.c_void_p * depth)()
    if mode == 0:
        if len(acts) != depth:
            raise ValueError('ratspn_topdown: %d activation tensors for depth %d' % (len(acts), depth))
        for t, a in enumerate(acts):
            act_arr[t] = ptr(dev(a, 'act'))
    if len(logws) != depth:
        raise ValueError('ratspn_topdown: %d weight tensors for depth %d' % (len(logws), depth))
    logw_arr = (ctypes.c_void_p * (depth + 1))()
    for t, w in enumerate(logws):
        logw_arr[t + 1] = ptr(dev(w, 'logw'))
    xd = dev(x, 'x') if x is not None else None
    yd = None
    if y is not None:
        yd = y.to(device=device, dtype=torch.int64).contiguous()
        if not yd.is_cuda:
            raise HipError('ratspn_topdown: y is not on a HIP device')
    p0d, p1d = dev(p0, 'p0'), (dev(p1, 'p1') if p1 is not None else None)
    out = torch.empty((B, lctx.D), dtype=torch.float32, device=device)
    choice = torch.empty((B, 1 + (1 << depth)), dtype=torch.int32, device=device) if want_choice else None
    check(lib.dpk_ratspn_topdown(mode, dist, B, lctx.D, depth, lctx.reps, lctx.I, lctx.S, lctx.C, lctx.d, ptr(xd), ptr(yd),
                                 ctypes.cast(act_arr, ctypes.c_void_p), ctypes.cast(logw_arr, ctypes.c_void_p), ptr(src),
                                 ptr(p0d), ptr(p1d), seed & 0xFFFFFFFFFFFFFFFF, ptr(out), ptr(choice), stream_ptr(device)),
          'dpk_ratspn_topdown')
    return (out, choice) if want_choice else out


class NegMeanFn(torch.autograd.Function):
    """``-torch.mean(x)``, the generative loss (reference: models/ratspn.py:184-191), one launch forward (fp64
    accumulation) and one backward."""

    @staticmethod
    def forward(ctx, x):
        lib = load_library()
        x = require_device_f32(x, 'x')
        out = torch.empty((), dtype=torch.float32, device=x.device)
        check(lib.dpk_neg_mean_forward(ptr(x), x.numel(), ptr(out), stream_ptr(x.device)), 'dpk_neg_mean_forward')
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        g = require_device_f32(g, 'grad')
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib.dpk_neg_mean_backward(ptr(g), gx.numel(), ptr(gx), stream_ptr(g.device)), 'dpk_neg_mean_backward')
        return gx


def neg_mean(x: torch.Tensor) -> torch.Tensor:
    """``-torch.mean(x)``; torch's own ops outside the kernel's range (CPU tensors, other dtypes, more than 2^20 entries)."""
    if x.is_cuda and x.dtype == torch.float32 and 0 < x.numel() <= (1 << 20):
        return NegMeanFn.apply(x)
    return -torch.mean(x)
