"""Autograd operators over the DGC-SPN spatial kernels (csrc/dgcspn.hip).

Every operator calls the C ABI directly and raises when the library or a device tensor is missing;
there is no CPU path.
"""
import ctypes

import torch

from deeprob.hip import (load_library, check, ptr, stream_ptr, require_device_f32, Workspace, DPK_FLAG_PARAMS_CACHED,
                         DPK_FLAG_IN_PIXEL_MAJOR, DPK_FLAG_OUT_PIXEL_MAJOR,
                         cached_tables_flag)


class SpatialGaussianFn(torch.autograd.Function):
    """SpatialGaussianLayer.forward (reference: deeprob/spn/layers/dgcspn.py:101-120)."""

    @staticmethod
    def forward(ctx, x, loc, scale, rate=0.0, seed=0):
        """rate > 0: training-mode input dropout (reference :113-114), decisions hashed from (seed, element)."""
        lib = load_library()
        x = require_device_f32(x, 'x')
        loc_c, scale_c = require_device_f32(loc, 'loc'), require_device_f32(scale, 'scale')
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(loc_c.shape[1:]):
            raise ValueError(f"expected input [B, {', '.join(map(str, loc_c.shape[1:]))}], got {tuple(x.shape)}")
        B, C, H, W = x.shape
        K = loc_c.shape[0]
        out = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
        if rate > 0.0:
            check(lib.dpk_spatial_gaussian_forward_dropout(ptr(x), ptr(loc_c), ptr(scale_c), B, K, C, H, W, float(rate),
                                                           seed, ptr(out), stream_ptr(x.device)),
                  'dpk_spatial_gaussian_forward_dropout')
        else:
            check(lib.dpk_spatial_gaussian_forward(ptr(x), ptr(loc_c), ptr(scale_c), B, K, C, H, W, ptr(out),
                                                   stream_ptr(x.device)), 'dpk_spatial_gaussian_forward')
        ctx.save_for_backward(x, loc_c, scale_c)
        ctx.drop = (float(rate), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, loc, scale = ctx.saved_tensors
        g = require_device_f32(g, 'grad')
        B, C, H, W = x.shape
        K = loc.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gl = torch.empty_like(loc) if ctx.needs_input_grad[1] else None
        gs = torch.empty_like(scale) if ctx.needs_input_grad[2] else None
        rate, seed = ctx.drop
        check(lib.dpk_spatial_gaussian_backward_dropout(ptr(x), ptr(g), ptr(loc), ptr(scale), B, K, C, H, W, rate, seed,
                                                        ptr(gl), ptr(gs), ptr(gx), stream_ptr(x.device)),
              'dpk_spatial_gaussian_backward_dropout')
        return gx, gl, gs, None, None


def _geom(layer):
    """Geometry of a SpatialProductLayer as the C ABI takes it."""
    C, H, W = layer.in_features
    OC, OH, OW = layer.out_features
    kh, kw = layer.kernel_size
    return (C, H, W, OC, OH, OW, kh, kw, layer.stride[0], layer.stride[1], layer.dilation[0], layer.dilation[1],
            layer.pad[2], layer.pad[0], 1 if layer.depthwise else 0)


class SpatialProductFn(torch.autograd.Function):
    """SpatialProductLayer.forward (reference: dgcspn.py:224-236)."""

    @staticmethod
    def forward(ctx, x, layer):
        lib = load_library()
        x = require_device_f32(x, 'x')
        geom = _geom(layer)
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(layer.in_features):
            raise ValueError(f"expected input [B, {layer.in_features}], got {tuple(x.shape)}")
        B = x.shape[0]
        out = torch.empty((B,) + tuple(layer.out_features), dtype=torch.float32, device=x.device)
        check(lib.dpk_spatial_product_forward(ptr(x), B, *geom, ptr(out), stream_ptr(x.device)),
              'dpk_spatial_product_forward')
        ctx.geom = geom
        ctx.in_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        g = require_device_f32(g, 'grad')
        gin = torch.empty(ctx.in_shape, dtype=torch.float32, device=g.device)
        check(lib.dpk_spatial_product_backward(ptr(g), ctx.in_shape[0], *ctx.geom, ptr(gin), stream_ptr(g.device)),
              'dpk_spatial_product_backward')
        return gin, None


def _spatial_sum_ws(ws: Workspace, Cin, Cout, H, W, device):
    lib = load_library()
    n = lib.dpk_spatial_sum_workspace_bytes(Cin, Cout, H, W)
    if n < 0:
        check(int(n), 'dpk_spatial_sum_workspace_bytes')
    return ws.get(n, device)


class SpatialSumFn(torch.autograd.Function):
    """SpatialSumLayer.forward in eval mode (reference: dgcspn.py:289-304)."""

    @staticmethod
    def forward(ctx, x, weight, ws: Workspace):
        lib = load_library()
        x = require_device_f32(x, 'x')
        w = require_device_f32(weight, 'weight')
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(w.shape[1:]):
            raise ValueError(f"expected input [B, {', '.join(map(str, w.shape[1:]))}], got {tuple(x.shape)}")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        buf = _spatial_sum_ws(ws, Cin, Cout, H, W, x.device)
        ws.params_key = None   # (the folded eval route caches its tables in the same workspace)
        check(lib.dpk_spatial_sum_forward(ptr(x), ptr(w), B, Cin, Cout, H, W, ptr(out), ptr(buf), buf.numel(),
                                          stream_ptr(x.device)), 'dpk_spatial_sum_forward')
        ctx.save_for_backward(x, w, out)
        ctx.ws = ws
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, w, out = ctx.saved_tensors
        g = require_device_f32(g, 'grad')
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        buf = _spatial_sum_ws(ctx.ws, Cin, Cout, H, W, x.device)
        ctx.ws.params_key = None
        check(lib.dpk_spatial_sum_backward(ptr(x), ptr(w), ptr(out), ptr(g), B, Cin, Cout, H, W, ptr(gx), ptr(gw),
                                           ptr(buf), buf.numel(), stream_ptr(x.device)),
              'dpk_spatial_sum_backward')
        return gx, gw, None


# ---- one table launch per DgcSpn.forward (round 3) ----------------------------------------------------------------------
# DgcSpn.forward (evaluation) calls tables_prepare once with the levels its loop is about to take: their softmaxed-weight
# tables (and the root's log-softmax rows) are rebuilt by ONE launch (dpk_spatial_tables) -- rebuilding is the check of
# these tables, see _tables_flag -- and the level operators below recognise the forward's token and pass
# DPK_FLAG_PARAMS_CACHED instead of launching a softmax kernel each.
_prep_token = None


class _SpatialTablesArgs(ctypes.Structure):     # dpk_spatial_tables_args
    _fields_ = [('sum_weight', ctypes.c_void_p), ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_int64),
                ('root_weight', ctypes.c_void_p), ('C', ctypes.c_int32), ('Cout', ctypes.c_int32), ('OHW', ctypes.c_int32),
                ('K', ctypes.c_int32), ('M', ctypes.c_int32)]


def _weights_key(route, *weights):
    return (route,) + tuple((w.data_ptr(), tuple(w.shape), w._version) for w in weights)


def tables_release():
    global _prep_token
    _prep_token = None


def tables_prepare(levels, x: torch.Tensor):
    """levels: [('prodsum', product layer, sum layer)] ... optionally ending with ('sumprodroot', product layer, sum
    layer, last product layer, root layer) -- what DgcSpn.forward is about to evaluate through spatial_prodsum /
    spatial_sumprodroot.  No-op when the version counters are trusted (unchanged tables are not rebuilt then)."""
    global _prep_token
    _prep_token = None
    from deeprob import hip
    if hip._trust_versions or not x.is_cuda or not levels or len(levels) > 8:
        return
    lib = load_library()
    dev, B = x.device, x.shape[0]
    entries, marked = [], []
    for lv in levels:
        prod, sm = lv[1], lv[2]
        w = sm.weight
        if not (prod.depthwise and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = _geom(prod)
        Cout = w.shape[0]
        if tuple(w.shape[1:]) != (C, OH, OW):
            continue
        if lv[0] == 'prodsum':
            ws = sm._ws
            buf = _spatial_sum_ws(ws, C, Cout, OH, OW, dev)
            key = _weights_key('prodsum', w)
            entries.append(_SpatialTablesArgs(ptr(w), ptr(buf), buf.numel(), None, C, Cout, OH * OW, 0, 0))
        else:
            prod6, root = lv[3], lv[4]
            wr = root.weight
            if not (prod6.depthwise and wr.is_cuda and wr.dtype == torch.float32 and wr.is_contiguous()):
                continue
            C6, H6, W6, _, OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6, _ = _geom(prod6)
            K = wr.shape[0]
            if (C6, H6, W6) != (Cout, OH, OW) or wr.shape[1] != Cout * OH6 * OW6:
                continue
            g5 = (ctypes.c_int32 * 10)(OH, OW, kh, kw, sh, sw, dh, dw, pt, pl)
            g6 = (ctypes.c_int32 * 10)(OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6)
            n = lib.dpk_spatial_sumprodroot_workspace_bytes_batch(B, C, H, W, g5, Cout, g6, K)
            if n < 0:   # (DPK_EUNSUPPORTED: outside dpk_spatial_sumprodroot_forward's envelope -- the library's answer, no copy of its constants here)
                # outside the last-level kernel (wide models): the forward will run this level through spatial_prodsum --
                # its tables join this launch instead of costing one of their own
                ws = sm._ws
                buf = _spatial_sum_ws(ws, C, Cout, OH, OW, dev)
                key = _weights_key('prodsum', w)
                entries.append(_SpatialTablesArgs(ptr(w), ptr(buf), buf.numel(), None, C, Cout, OH * OW, 0, 0))
                marked.append((ws, key))
                continue
            ws = root._ws3
            buf = ws.get(n, dev)
            key = _weights_key('sumprodroot', w, wr)
            entries.append(_SpatialTablesArgs(ptr(w), ptr(buf), buf.numel(), ptr(wr), C, Cout, OH * OW, K, wr.shape[1]))
        marked.append((ws, key))
    if not entries:
        return
    arr = (_SpatialTablesArgs * len(entries))(*entries)
    check(lib.dpk_spatial_tables(len(entries), ctypes.cast(arr, ctypes.c_void_p), stream_ptr(dev)), 'dpk_spatial_tables')
    token = object()
    for ws, key in marked:
        ws.params_key = key
        ws._prep_token = token
        ws._prep_key = key
    _prep_token = token


def _tables_flag(ws: Workspace, route: str, *weights) -> int:
    """The cached-tables flag (``hip.cached_tables_flag``: checked on the device by default, so that a write through
    ``weight.data`` is seen) when the workspace's softmaxed-weight tables were built by an earlier call of the same entry
    point from these very tensors (address, shape, version counter; Workspace.get() drops the key when the buffer is
    replaced)."""
    key = (route,) + tuple((w.data_ptr(), tuple(w.shape), w._version) for w in weights)
    if _prep_token is not None and getattr(ws, '_prep_token', None) is _prep_token and getattr(ws, '_prep_key', None) == key:
        return DPK_FLAG_PARAMS_CACHED      # rebuilt by this forward's tables_prepare
    if ws.params_key == key:
        return cached_tables_flag()
    ws.params_key = key
    return 0


def _is_pixel_major(x: torch.Tensor) -> bool:
    """A [B, C, H, W] view of a dense [B, H, W, C] buffer (torch's channels_last) that is NOT also plainly contiguous."""
    return (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[1] > 1 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last))


def _dense_f32(x: torch.Tensor, name: str) -> torch.Tensor:
    """require_device_f32 that leaves a pixel-major map as it is."""
    return x if _is_pixel_major(x) else require_device_f32(x, name)


def level_streams(prod_layer, B: int, Cout: int, last_prod=None, K: int = 0) -> bool:
    """Whether the fused level runs on the streaming route at this batch size -- the route that takes / leaves pixel-major
    maps (the library's answer, dpk_spatial_level_streams)."""
    if not prod_layer.depthwise:
        return False
    C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = _geom(prod_layer)
    g5 = (ctypes.c_int32 * 10)(OH, OW, kh, kw, sh, sw, dh, dw, pt, pl)
    g6 = None
    if last_prod is not None:
        if not last_prod.depthwise:
            return False
        _, _, _, _, OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6, _ = _geom(last_prod)
        g6 = (ctypes.c_int32 * 10)(OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6)
    return bool(load_library().dpk_spatial_level_streams(0 if last_prod is None else 1, B, C, H, W, g5, Cout, g6, K))


def spatial_prodsum(x, prod_layer, weight, ws: Workspace, out_pixel_major: bool = False):
    """One eval-mode DGC-SPN level, depthwise SpatialProductLayer + SpatialSumLayer in a single launch
    (reference: deeprob/spn/models/dgcspn.py:146-147).  No autograd graph is recorded.  Returns None when
    the level is outside what the fused kernel covers (the caller chains the two layer operators).

    A pixel-major ``x`` (torch's channels_last) is consumed as it is by the streaming route; ``out_pixel_major`` asks that
    route for a channels_last result (same shape, same values, other strides) -- the layout in which the next streaming
    level reads a tap's 8 channels with two 16-byte LDS reads.  Outside the streaming route both fall back to the plain
    layout (one conversion)."""
    lib = load_library()
    x = _dense_f32(x, 'x')
    w = require_device_f32(weight, 'weight')
    if not prod_layer.depthwise:
        return None
    if x.dim() != 4 or tuple(x.shape[1:]) != tuple(prod_layer.in_features):
        raise ValueError(f"expected input [B, {prod_layer.in_features}], got {tuple(x.shape)}")
    C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = _geom(prod_layer)
    B, Cout = x.shape[0], w.shape[0]
    in_pm = _is_pixel_major(x)
    if (in_pm or out_pixel_major) and not level_streams(prod_layer, B, Cout):
        x, in_pm, out_pixel_major = x.contiguous(), False, False
    if out_pixel_major:
        out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    else:
        out = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    buf = _spatial_sum_ws(ws, C, Cout, OH, OW, x.device)
    flags = _tables_flag(ws, 'prodsum', w) | (DPK_FLAG_IN_PIXEL_MAJOR if in_pm else 0) | \
        (DPK_FLAG_OUT_PIXEL_MAJOR if out_pixel_major else 0)
    rc = lib.dpk_spatial_prodsum_forward(ptr(x), B, C, H, W, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, ptr(w), Cout,
                                         ptr(out), ptr(buf), buf.numel(), flags, stream_ptr(x.device))
    if rc == -4 and (in_pm or out_pixel_major):      # (e.g. a misaligned view: the plain layout once more)
        return spatial_prodsum(x.contiguous(), prod_layer, weight, ws)
    if rc:
        ws.params_key = None
    if rc == -4:  # DPK_EUNSUPPORTED
        return None
    check(rc, 'dpk_spatial_prodsum_forward')
    return out


def spatial_leaf_prodsum(x, leaf_layer, prod_layer, weight, ws: Workspace, out_pixel_major: bool = False):
    """SpatialGaussianLayer + the first depthwise product + sum level of the eval route in ONE launch (reference:
    deeprob/spn/models/dgcspn.py:134-147): the [B, K, H, W] leaf map is never written.  None when the level is outside
    the fused kernel's envelope (the caller evaluates the leaf layer and then ``spatial_prodsum``)."""
    lib = load_library()
    if not prod_layer.depthwise or not x.is_cuda or x.dim() != 4:
        return None
    x = require_device_f32(x, 'x')
    w = require_device_f32(weight, 'weight')
    loc, scale = require_device_f32(leaf_layer.loc, 'loc'), require_device_f32(leaf_layer.scale, 'scale')
    if loc.dim() != 4 or tuple(loc.shape) != tuple(scale.shape) or tuple(x.shape[1:]) != tuple(loc.shape[1:]):
        return None
    K, Cx = loc.shape[0], loc.shape[1]
    C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = _geom(prod_layer)
    if C != K or (H, W) != tuple(x.shape[2:]):
        return None
    B, Cout = x.shape[0], w.shape[0]
    if out_pixel_major:      # (honoured by the streaming route only: anything else answers DPK_EUNSUPPORTED, plain layout then)
        out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    else:
        out = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    buf = _spatial_sum_ws(ws, C, Cout, OH, OW, x.device)
    flags = _tables_flag(ws, 'prodsum', w) | (DPK_FLAG_OUT_PIXEL_MAJOR if out_pixel_major else 0)
    rc = lib.dpk_spatial_leaf_prodsum_forward(ptr(x), ptr(loc), ptr(scale), B, Cx, K, H, W, OH, OW, kh, kw, sh, sw, dh, dw, pt,
                                              pl, ptr(w), Cout, ptr(out), ptr(buf), buf.numel(), flags, stream_ptr(x.device))
    if rc == -4 and out_pixel_major:     # (not on the streaming route: the plain layout once more)
        return spatial_leaf_prodsum(x, leaf_layer, prod_layer, weight, ws)
    if rc:
        ws.params_key = None
    if rc == -4:  # DPK_EUNSUPPORTED
        return None
    check(rc, 'dpk_spatial_leaf_prodsum_forward')
    return out


class SpatialProdSumFn(torch.autograd.Function):
    """Depthwise SpatialProductLayer + SpatialSumLayer as ONE autograd node (training route of a DGC-SPN level,
    reference: deeprob/spn/models/dgcspn.py:146-147 chaining layers/dgcspn.py:224-236 and :289-304): the forward is the
    fused evaluation kernel, the backward recomputes the product map from the taps -- the [B,C,OH,OW] product tensor is
    neither written nor kept for the backward."""

    @staticmethod
    def forward(ctx, x, weight, geom, ws: Workspace):
        lib = load_library()
        x = require_device_f32(x, 'x')
        w = require_device_f32(weight, 'weight')
        C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = geom
        B, Cout = x.shape[0], w.shape[0]
        out = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
        buf = _spatial_sum_ws(ws, C, Cout, OH, OW, x.device)
        flags = _tables_flag(ws, 'prodsum', w)   # (same tables as the evaluation route builds)
        check(lib.dpk_spatial_prodsum_forward(ptr(x), B, C, H, W, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, ptr(w), Cout,
                                              ptr(out), ptr(buf), buf.numel(), flags, stream_ptr(x.device)),
              'dpk_spatial_prodsum_forward')
        ctx.save_for_backward(x, w, out)
        ctx.geom, ctx.ws = geom, ws
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, w, out = ctx.saved_tensors
        g = require_device_f32(g, 'grad')
        C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = ctx.geom
        B, Cout = x.shape[0], w.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gprod = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device) if gx is not None else None
        buf = _spatial_sum_ws(ctx.ws, C, Cout, OH, OW, x.device)
        # the forward's tables (softmax(W), log softmax(W)) are still there unless the weight or the workspace changed.
        # "Believed current -- check on the device" becomes "current" here: the gradient wanted is that of the function the
        # forward evaluated, with the tables IT used (a rebuild was five softmax launches per DGC-SPN step, 42 us of 840)
        flags = _tables_flag(ctx.ws, 'prodsum', w)
        if flags:
            flags = DPK_FLAG_PARAMS_CACHED
        check(lib.dpk_spatial_prodsum_backward(ptr(x), B, C, H, W, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, ptr(w), Cout,
                                               ptr(out), ptr(g), ptr(gprod), ptr(gx), ptr(gw), ptr(buf), buf.numel(),
                                               flags, stream_ptr(x.device)), 'dpk_spatial_prodsum_backward')
        return gx, gw, None, None


def spatial_prodsum_autograd(x, prod_layer, weight, ws: Workspace):
    """The fused level with an autograd graph; None when the level is outside the fused kernels' envelope (non-depthwise
    product, more than 4 taps, more than 8 channels): the caller chains the two layers."""
    if not prod_layer.depthwise:
        return None
    geom = _geom(prod_layer)
    C, kh, kw = geom[0], geom[6], geom[7]
    if kh * kw > 4 or C > 8 or weight.shape[0] > 8 or x.dim() != 4 or tuple(x.shape[1:]) != tuple(prod_layer.in_features):
        return None
    return SpatialProdSumFn.apply(x, weight, geom, ws)


def spatial_prodroot(x, prod_layer, weight, ws: Workspace):
    """Last level of the eval route: depthwise SpatialProductLayer + SpatialRootLayer in one launch (reference:
    deeprob/spn/models/dgcspn.py:146-150).  No autograd graph; None when outside the fused kernel's envelope."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    w = require_device_f32(weight, 'weight')
    if not prod_layer.depthwise:
        return None
    if x.dim() != 4 or tuple(x.shape[1:]) != tuple(prod_layer.in_features):
        raise ValueError(f"expected input [B, {prod_layer.in_features}], got {tuple(x.shape)}")
    C, H, W, _, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, _ = _geom(prod_layer)
    B, K = x.shape[0], w.shape[0]
    if w.shape[1] != C * OH * OW:
        raise ValueError("root weight does not match the product layer's output")
    n = lib.dpk_spatial_prodroot_workspace_bytes(C, OH, OW, K)
    if n < 0:
        check(int(n), 'dpk_spatial_prodroot_workspace_bytes')
    buf = ws.get(n, x.device)
    out = torch.empty((B, K), dtype=torch.float32, device=x.device)
    rc = lib.dpk_spatial_prodroot_forward(ptr(x), B, C, H, W, OH, OW, kh, kw, sh, sw, dh, dw, pt, pl, ptr(w), K,
                                          ptr(out), ptr(buf), buf.numel(), stream_ptr(x.device))
    if rc == -4:  # DPK_EUNSUPPORTED
        return None
    check(rc, 'dpk_spatial_prodroot_forward')
    return out


def spatial_sumprodroot(x, prod5, sum_weight, prod6, root_weight, ws: Workspace):
    """Last three stages of the eval route in one launch: depthwise SpatialProductLayer + SpatialSumLayer, the final
    depthwise SpatialProductLayer and the SpatialRootLayer (reference: deeprob/spn/models/dgcspn.py:146-150).  The
    largest activation map of the model is never written.  No autograd graph; None when outside the fused kernel's
    envelope (the caller falls back to spatial_prodsum + spatial_prodroot)."""
    import ctypes
    lib = load_library()
    x = _dense_f32(x, 'x')
    w5 = require_device_f32(sum_weight, 'weight')
    wr = require_device_f32(root_weight, 'weight')
    if not (prod5.depthwise and prod6.depthwise):
        return None
    if _is_pixel_major(x) and not level_streams(prod5, x.shape[0], w5.shape[0], prod6, wr.shape[0]):
        x = x.contiguous()
    in_pm = _is_pixel_major(x)
    if x.dim() != 4 or tuple(x.shape[1:]) != tuple(prod5.in_features):
        raise ValueError(f"expected input [B, {prod5.in_features}], got {tuple(x.shape)}")
    C, H, W, _, OH5, OW5, kh5, kw5, sh5, sw5, dh5, dw5, pt5, pl5, _ = _geom(prod5)
    C6, H6, W6, _, OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6, _ = _geom(prod6)
    Cout, K = w5.shape[0], wr.shape[0]
    if tuple(w5.shape[1:]) != (C, OH5, OW5) or (C6, H6, W6) != (Cout, OH5, OW5) or wr.shape[1] != Cout * OH6 * OW6:
        raise ValueError("layer shapes do not chain")
    g5 = (ctypes.c_int32 * 10)(OH5, OW5, kh5, kw5, sh5, sw5, dh5, dw5, pt5, pl5)
    g6 = (ctypes.c_int32 * 10)(OH6, OW6, kh6, kw6, sh6, sw6, dh6, dw6, pt6, pl6)
    B = x.shape[0]
    n = lib.dpk_spatial_sumprodroot_workspace_bytes_batch(B, C, H, W, g5, Cout, g6, K)
    if n == -4:  # DPK_EUNSUPPORTED: outside the fused kernel's envelope
        return None
    if n < 0:
        check(int(n), 'dpk_spatial_sumprodroot_workspace_bytes_batch')
    buf = ws.get(n, x.device)
    out = torch.empty((B, K), dtype=torch.float32, device=x.device)
    flags = _tables_flag(ws, 'sumprodroot', w5, wr) | (DPK_FLAG_IN_PIXEL_MAJOR if in_pm else 0)
    rc = lib.dpk_spatial_sumprodroot_forward(ptr(x), B, C, H, W, g5, ptr(w5), Cout, g6, ptr(wr), K, ptr(out), ptr(buf),
                                             buf.numel(), flags, stream_ptr(x.device))
    if rc == -4 and in_pm:
        return spatial_sumprodroot(x.contiguous(), prod5, sum_weight, prod6, root_weight, ws)
    if rc:
        ws.params_key = None
    if rc == -4:  # DPK_EUNSUPPORTED
        return None
    check(rc, 'dpk_spatial_sumprodroot_forward')
    return out
