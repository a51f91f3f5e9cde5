"""RealNVP-2D training direction: autograd nodes over the HIP kernels of csrc/flows2d.hip and csrc/flows2d_train.hip.

The reference trains these models through ATen's autograd (flows/layers/resnet.py, densenet.py, coupling.py:181-226,
flows/utils.py:186-208).  Here every pass over an image tensor is a HIP kernel behind an ``autograd.Function``:

* batch statistics of nn.BatchNorm2d / BatchNormLayer2d (``ChannelStatsFn``), whose mean and variance stay in the graph;
  the BatchNorm2d + ReLU in front of a convolution stays folded into the convolution's operand load, now with the scale /
  shift vectors computed from the batch statistics -- in training mode inside one node with the convolution
  (``BnConv2dFn``: statistics, fold, convolution and the whole BatchNorm gradient as device passes), with running
  statistics and gradients wanted as small [C] tensor arithmetic left to torch's autograd,
* the convolution (``Conv2dFn``): input gradient = the forward kernel on the output gradient with transposed, flipped
  weights, then the backward of the folded operand map; weight gradient = ``dpk_conv2d_backward_weight``; the weight
  normalisation ``g v / |v|`` is [Cout, Cin, k, k] parameter arithmetic, left to torch's autograd,
* the coupling transformation, BatchNormLayer2d's affine map, squeeze and multi-scale permutations.

Only the density direction has a backward (what ``train_model`` / ``loss`` need).
"""
from typing import Optional, Tuple

import torch

from deeprob.hip import load_library, check, ptr, stream_ptr, require_device_f32, HipError
from deeprob.hip import ops_flows2d as ev


def wants_graph(module, *tensors) -> bool:
    """Training-mode batch statistics or a gradient through the module is asked for."""
    if module.training:
        return True
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors) or \
        any(p.requires_grad for p in module.parameters())


def _grad_image(g: torch.Tensor) -> torch.Tensor:
    return g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()


def _sync_counts(group, B: int, HW: int, sums: torch.Tensor):
    """Synchronised batch statistics (deeprob.parallel.synchronize_batchnorm): all-reduce the per-channel fp64 sums and
    return (elements of the whole batch, this rank's weight n_r / N).  The counts come from the training loop's shard
    sizes (host integers, no read-back); without them the count travels with the sums and is read back once."""
    import torch.distributed as dist
    from deeprob import parallel
    sizes = parallel.shard_sizes()
    if sizes is not None and sizes[0] == B:
        dist.all_reduce(sums, group=group)
        return sizes[1] * HW, float(sizes[0]) / float(sizes[1])
    buf = torch.cat([sums, torch.tensor([float(B * HW)], dtype=torch.float64, device=sums.device)])
    dist.all_reduce(buf, group=group)
    sums.copy_(buf[:-1])
    n_total = int(round(float(buf[-1].item())))
    return n_total, float(B * HW) / float(n_total)


def _sync_stat_grads(group, weight: float, dstat: torch.Tensor) -> torch.Tensor:
    """d(whole-batch loss) / d(mean, var) from the ranks' d(local loss) / d(mean, var): their n_r / N weighted sum."""
    import torch.distributed as dist
    g = dstat * weight
    dist.all_reduce(g, group=group)
    return g


class ChannelStatsFn(torch.autograd.Function):
    """Per-channel mean and biased variance of a [B, C, H, W] tensor over (B, H, W) -- over the whole sharded batch when a
    process group is given (synchronised batch normalisation)."""

    @staticmethod
    def forward(ctx, x, group=None):
        lib = load_library()
        x = ev._image(x, 'x')
        B, C, H, W = x.shape
        if B * H * W == 0:
            raise HipError("batch statistics of an empty batch")
        sums = torch.zeros(2 * C, dtype=torch.float64, device=x.device)
        check(lib.dpk_channel_stats(ptr(x), x.stride(0), B, C, H, W, 1, ptr(sums), stream_ptr(x.device)),
              'dpk_channel_stats')
        n = float(B * H * W)
        ctx.group, ctx.weight = group, 1.0
        if group is not None:
            n_total, ctx.weight = _sync_counts(group, B, H * W, sums)
            n = float(n_total)
        ctx.n_total = n
        mean64 = sums[:C] / n
        var64 = (sums[C:] / n - mean64 * mean64).clamp_min_(0.0)
        mean, var = mean64.float(), var64.float()
        ctx.save_for_backward(x, mean)
        ctx.mark_non_differentiable()
        return mean, var

    @staticmethod
    def backward(ctx, dmean, dvar):
        lib = load_library()
        x, mean = ctx.saved_tensors
        B, C, H, W = x.shape
        dmean = torch.zeros_like(mean) if dmean is None else _grad_image(dmean)
        dvar = torch.zeros_like(mean) if dvar is None else _grad_image(dvar)
        if ctx.group is not None:
            # (the kernel divides by the LOCAL element count: with the weighted sum over the ranks that is exactly
            # d(whole-batch loss)/dx as this rank's autograd must see it before the n_r / N weighted gradient exchange)
            g = _sync_stat_grads(ctx.group, ctx.weight, torch.cat([dmean, dvar]))
            dmean, dvar = g[:C].contiguous(), g[C:].contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        check(lib.dpk_channel_stats_backward(ptr(x), x.stride(0), B, C, H, W, ptr(mean), ptr(dmean), ptr(dvar), 0,
                                             ptr(dx), stream_ptr(x.device)), 'dpk_channel_stats_backward')
        return dx, None


class ChannelAffineFn(torch.autograd.Function):
    """out = ab[c] x + ab[C + c]."""

    @staticmethod
    def forward(ctx, x, ab):
        lib = load_library()
        x = require_device_f32(x, 'x')
        ab = require_device_f32(ab, 'ab')
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.dpk_channel_affine_forward(ptr(x), B, C, H, W, ptr(ab), ptr(out), stream_ptr(x.device)),
              'dpk_channel_affine_forward')
        ctx.save_for_backward(x, ab)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, ab = ctx.saved_tensors
        B, C, H, W = x.shape
        g = _grad_image(g)
        dx = torch.empty_like(x)
        dab = torch.zeros(2 * C, dtype=torch.float64, device=x.device)
        check(lib.dpk_channel_affine_backward(ptr(x), x.stride(0), ptr(g), B, C, H, W, ptr(ab), 0, None, ptr(dx),
                                              ptr(dab), stream_ptr(x.device)), 'dpk_channel_affine_backward')
        return dx, dab.float()


def _pack(w: torch.Tensor) -> torch.Tensor:
    """Kernel-side layout of an explicit [Cout, Cin, k, k] weight tensor."""
    lib = load_library()
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    wpack = torch.empty(lib.dpk_conv2d_pack_floats(cout, cin, ks), dtype=torch.float32, device=w.device)
    check(lib.dpk_conv2d_prepare(ptr(w), None, cout, cin, ks, None, None, None, None, 0.0, ptr(wpack), None,
                                 stream_ptr(w.device)), 'dpk_conv2d_prepare')
    return wpack


def _conv(x, wpack, cout, ks, pre, mask, bias, res):
    lib = load_library()
    B, cin, H, W = x.shape
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    check(lib.dpk_conv2d_forward(ptr(x), x.stride(0), B, cin, H, W, ptr(wpack), cout, ks, ptr(pre), ptr(mask),
                                 ptr(bias), ptr(res), 0 if res is None else res.stride(0), ptr(out), out.stride(0),
                                 stream_ptr(x.device)), 'dpk_conv2d_forward')
    return out


class Conv2dFn(torch.autograd.Function):
    """out = bias + conv(mask * relu(pre_a x + pre_b), w) + res (pre / mask / bias / res optional)."""

    @staticmethod
    def forward(ctx, x, w, bias, pre, mask, res):
        x = ev._image(x, 'x')
        w = require_device_f32(w, 'weight')
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        if x.shape[1] != cin:
            raise HipError("conv2d: input has {} channels, the layer expects {}".format(x.shape[1], cin))
        bias = None if bias is None else require_device_f32(bias, 'bias')
        pre = None if pre is None else require_device_f32(pre, 'pre')
        mask = None if mask is None else require_device_f32(mask, 'mask')
        res = None if res is None else ev._image(res, 'res')
        out = _conv(x, _pack(w), cout, ks, pre, mask, bias, res)
        ctx.save_for_backward(x, w, pre, mask)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, w, pre, mask = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout, ks = w.shape[0], w.shape[2]
        g = _grad_image(g)
        dev = x.device
        dx = dpre = dw = dbias = None
        if ctx.needs_input_grad[0] or (pre is not None and ctx.needs_input_grad[3]):
            wt = w.transpose(0, 1).flip(2, 3).contiguous()
            dh = _conv(g, _pack(wt), cin, ks, None, None, None, None)
            if pre is None and mask is None:
                dx = dh
            else:
                dx = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev)
                dab = None if pre is None else torch.zeros(2 * cin, dtype=torch.float64, device=dev)
                check(lib.dpk_channel_affine_backward(ptr(x), x.stride(0), ptr(dh), B, cin, H, W, ptr(pre),
                                                      int(pre is not None), ptr(mask), ptr(dx), ptr(dab),
                                                      stream_ptr(dev)), 'dpk_channel_affine_backward')
                dpre = None if dab is None else dab.float()
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            check(lib.dpk_conv2d_backward_weight(ptr(x), x.stride(0), ptr(g), B, cin, cout, H, W, ks, ptr(pre),
                                                 ptr(mask), ptr(dw), stream_ptr(dev)), 'dpk_conv2d_backward_weight')
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sums = torch.zeros(cout, dtype=torch.float64, device=dev)
            check(lib.dpk_channel_stats(ptr(g), g.stride(0), B, cout, H, W, 0, ptr(sums), stream_ptr(dev)),
                  'dpk_channel_stats')
            dbias = sums.float()
        return dx, dw, dbias, dpre, None, (g if ctx.has_res else None)


class BnConv2dFn(torch.autograd.Function):
    """Training-mode ``conv(mask * relu(bn(x)), w) + bias + res`` as ONE node: batch statistics, the fold of
    (gamma, beta, mean, var) into the operand map, the convolution, and in the backward the whole BatchNorm2d gradient
    (operand-map gradient -> gamma / beta / statistics -> input) without [C]-sized torch operators in between -- a
    training step of the default MNIST flow is ~70 such nodes and the small launches between the kernels were what it
    spent its time on.  The running statistics of `bn` are updated in the fold kernel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, bias, mask, res, bn):
        lib = load_library()
        x = ev._image(x, 'x')
        w = require_device_f32(w, 'weight')
        gamma = require_device_f32(gamma, 'batch norm weight')
        beta = require_device_f32(beta, 'batch norm bias')
        B, cin, H, W = x.shape
        cout, ks = w.shape[0], w.shape[2]
        if w.shape[1] != cin:
            raise HipError("conv2d: input has {} channels, the layer expects {}".format(cin, w.shape[1]))
        if B * H * W == 0:
            raise HipError("batch statistics of an empty batch")
        dev = x.device
        st = stream_ptr(dev)
        sums = torch.zeros(2 * cin, dtype=torch.float64, device=dev)
        check(lib.dpk_channel_stats(ptr(x), x.stride(0), B, cin, H, W, 1, ptr(sums), st), 'dpk_channel_stats')
        pre = torch.empty(2 * cin, dtype=torch.float32, device=dev)
        stat = torch.empty(2 * cin, dtype=torch.float32, device=dev)
        bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
        n_elems = B * H * W
        ctx.group, ctx.weight = getattr(bn, 'sync_group', None), 1.0
        if ctx.group is not None:       # statistics of the whole sharded batch (deeprob.parallel.synchronize_batchnorm)
            n_elems, ctx.weight = _sync_counts(ctx.group, B, H * W, sums)
        check(lib.dpk_bn2d_fold_train(ptr(sums), n_elems, cin, ptr(gamma), ptr(beta), float(bn.eps), float(momentum),
                                      ptr(bn.running_mean), ptr(bn.running_var), ptr(pre), ptr(stat), st),
              'dpk_bn2d_fold_train')
        bias = None if bias is None else require_device_f32(bias, 'bias')
        mask = None if mask is None else require_device_f32(mask, 'mask')
        res = None if res is None else ev._image(res, 'res')
        out = _conv(x, _pack(w), cout, ks, pre, mask, bias, res)
        ctx.save_for_backward(x, w, pre, stat, gamma, mask)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load_library()
        x, w, pre, stat, gamma, mask = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout, ks = w.shape[0], w.shape[2]
        g = _grad_image(g)
        dev = x.device
        st = stream_ptr(dev)
        wt = w.transpose(0, 1).flip(2, 3).contiguous()
        dh = _conv(g, _pack(wt), cin, ks, None, None, None, None)
        dx = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev)
        dab = torch.zeros(2 * cin, dtype=torch.float64, device=dev)
        check(lib.dpk_channel_affine_backward(ptr(x), x.stride(0), ptr(dh), B, cin, H, W, ptr(pre), 1, ptr(mask), ptr(dx),
                                              ptr(dab), st), 'dpk_channel_affine_backward')
        small = torch.empty(4 * cin, dtype=torch.float32, device=dev)     # dgamma | dbeta | dmean | dvar
        dgamma, dbeta, dstat = small[:cin], small[cin:2 * cin], small[2 * cin:]
        check(lib.dpk_bn2d_fold_backward(ptr(dab), cin, ptr(gamma), ptr(stat), ptr(dgamma), ptr(dbeta), ptr(dstat), st),
              'dpk_bn2d_fold_backward')
        if ctx.group is not None:
            dstat = _sync_stat_grads(ctx.group, ctx.weight, dstat)
        check(lib.dpk_channel_stats_backward(ptr(x), x.stride(0), B, cin, H, W, ptr(stat), ptr(dstat), ptr(dstat[cin:]),
                                             1, ptr(dx), st), 'dpk_channel_stats_backward')
        dw = dbias = None
        if ctx.needs_input_grad[3]:
            dw = torch.zeros_like(w)
            check(lib.dpk_conv2d_backward_weight(ptr(x), x.stride(0), ptr(g), B, cin, cout, H, W, ks, ptr(pre), ptr(mask),
                                                 ptr(dw), st), 'dpk_conv2d_backward_weight')
        if ctx.has_bias and ctx.needs_input_grad[4]:
            sums = torch.zeros(cout, dtype=torch.float64, device=dev)
            check(lib.dpk_channel_stats(ptr(g), g.stride(0), B, cout, H, W, 0, ptr(sums), st), 'dpk_channel_stats')
            dbias = sums.float()
        return dx, dgamma, dbeta, dw, dbias, None, (g if ctx.has_res else None), None


class WeightNormFn(torch.autograd.Function):
    """``w = g v / |v|`` per output channel (torch.nn.utils.weight_norm with dim 0, reference torch/utils.py:103-115) with
    a hand-written backward: 9 small launches instead of the ~19 of autograd over reshape / norm / div / mul -- a
    training step has 180 of these.  dg = <dw, v> / |v|,  dv = (g / |v|) dw - (g <dw, v> / |v|^3) v."""

    @staticmethod
    def forward(ctx, v, g):
        cout = v.shape[0]
        norm = v.reshape(cout, -1).norm(dim=1)
        scale = g.reshape(-1) / norm
        ctx.save_for_backward(v, scale, norm)
        ctx.g_shape = g.shape
        return v * scale.view(-1, 1, 1, 1)

    @staticmethod
    def backward(ctx, dw):
        v, scale, norm = ctx.saved_tensors
        cout = v.shape[0]
        s = (dw * v).reshape(cout, -1).sum(dim=1) / norm
        dv = torch.addcmul(dw * scale.view(-1, 1, 1, 1), v, (-(scale * s / norm)).view(-1, 1, 1, 1))
        return dv, s.reshape(ctx.g_shape)


def effective_weight(p) -> torch.Tensor:
    """The weight-normalised kernel of a WeightNormConv2d's parameters."""
    return WeightNormFn.apply(p.weight_v, p.weight_g)


def batchnorm_operand_map(bn, x: torch.Tensor) -> torch.Tensor:
    """[a | b] with ``bn(x) = a x + b`` per channel for an nn.BatchNorm2d: batch statistics (and the running-statistics
    update of torch's module) in training mode, running statistics otherwise."""
    if bn.weight is None or bn.running_mean is None:
        raise HipError("conv2d: BatchNorm2d without affine parameters / running statistics is not built")
    if bn.training:
        group = getattr(bn, 'sync_group', None)
        mean, var = ChannelStatsFn.apply(x, group)
        with torch.no_grad():
            n = x.shape[0] * x.shape[2] * x.shape[3]
            if group is not None:
                from deeprob import parallel
                sizes = parallel.shard_sizes()
                if sizes is not None and sizes[0] == x.shape[0]:
                    n = sizes[1] * x.shape[2] * x.shape[3]
            bn.num_batches_tracked.add_(1)
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
            bn.running_mean.mul_(1.0 - m).add_(mean.detach(), alpha=m)
            bn.running_var.mul_(1.0 - m).add_(var.detach(), alpha=m * n / max(n - 1, 1))
    else:
        mean, var = bn.running_mean, bn.running_var
    a = bn.weight * torch.rsqrt(var + bn.eps)
    return torch.cat([a, bn.bias - mean * a])


def conv2d(x: torch.Tensor, conv, bn=None, in_mask: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The graph-building form of :func:`deeprob.hip.ops_flows2d.conv2d`."""
    p = conv.conv
    mask = None if in_mask is None else in_mask.reshape(-1)
    if bn is not None and bn.training:
        if bn.weight is None or bn.running_mean is None:
            raise HipError("conv2d: BatchNorm2d without affine parameters / running statistics is not built")
        return BnConv2dFn.apply(x, bn.weight, bn.bias, effective_weight(p), p.bias, mask, res, bn)
    pre = None if bn is None else batchnorm_operand_map(bn, x)
    return Conv2dFn.apply(x, effective_weight(p), p.bias, pre, mask, res)


class CouplingTransformFn(torch.autograd.Function):
    """(u, ildj) of CouplingLayer2d.apply_backward given the conditioner output z."""

    @staticmethod
    def forward(ctx, x, z, scale, inv_mask, affine, reverse):
        lib = load_library()
        x = require_device_f32(x, 'x')
        z = require_device_f32(z, 'z')
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        ldj = torch.empty(B, dtype=torch.float32, device=x.device)
        sc = None if scale is None else require_device_f32(scale, 'scale_act.weight').view(-1)
        check(lib.dpk_coupling2d_transform(ptr(x), ptr(z), ptr(sc), ptr(inv_mask), B, C, H, W, int(affine), int(reverse),
                                           0, None, ptr(out), ptr(ldj), stream_ptr(x.device)),
              'dpk_coupling2d_transform')
        ctx.save_for_backward(x, z, sc, inv_mask)
        ctx.affine, ctx.reverse = affine, reverse
        ctx.scale_shape = None if scale is None else scale.shape
        return out, ldj

    @staticmethod
    def backward(ctx, gout, gldj):
        lib = load_library()
        x, z, sc, inv_mask = ctx.saved_tensors
        B, C, H, W = x.shape
        gout = torch.zeros_like(x) if gout is None else _grad_image(gout)
        gldj = None if gldj is None else _grad_image(gldj)
        dx = torch.empty_like(x)
        dz = torch.empty_like(z)
        dscale = None if sc is None else torch.zeros(sc.numel(), dtype=torch.float64, device=x.device)
        check(lib.dpk_coupling2d_transform_backward(ptr(x), ptr(z), ptr(sc), ptr(inv_mask), B, C, H, W, int(ctx.affine),
                                                    int(ctx.reverse), ptr(gout), ptr(gldj), ptr(dx), ptr(dz),
                                                    ptr(dscale), stream_ptr(x.device)),
              'dpk_coupling2d_transform_backward')
        return dx, dz, (None if dscale is None else dscale.float().reshape(ctx.scale_shape)), None, None, None


def coupling2d(x: torch.Tensor, z: torch.Tensor, layer) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = layer.scale_act.weight if layer.affine else None
    inv_mask = None if layer.channelwise else require_device_f32(layer.inv_mask, 'inv_mask').reshape(-1)
    return CouplingTransformFn.apply(x, z, scale, inv_mask, bool(layer.affine), bool(layer.reverse))


def bn2d(x: torch.Tensor, layer) -> Tuple[torch.Tensor, torch.Tensor]:
    """BatchNormLayer2d.apply_backward (flows/utils.py:186-208) with the statistics in the graph."""
    B, C, H, W = x.shape
    if layer.training:
        mean, var = ChannelStatsFn.apply(x, getattr(layer, 'sync_group', None))
        with torch.no_grad():
            layer.running_var.mul_(layer.momentum).add_(var.view_as(layer.running_var), alpha=1.0 - layer.momentum)
            layer.running_mean.mul_(layer.momentum).add_(mean.view_as(layer.running_mean), alpha=1.0 - layer.momentum)
    else:
        mean, var = layer.running_mean.view(-1), layer.running_var.view(-1)
    var = var + layer.eps
    weight = layer.weight.view(-1)
    a = torch.exp(weight) * torch.rsqrt(var)
    u = ChannelAffineFn.apply(x, torch.cat([a, layer.bias.view(-1) - mean * a]))
    ildj = torch.sum(weight - 0.5 * torch.log(var)) * float(H * W)
    return u, ildj.expand(B)


class SpaceToDepthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, split):
        ctx.table, ctx.split = table, split
        ctx.shape = x.shape
        if split is None:
            return ev.space_to_depth(x, table)
        a, b = ev.space_to_depth(x, table, split=split)
        return a, b

    @staticmethod
    def backward(ctx, *grads):
        B, C, H, W = ctx.shape
        if ctx.split is None:
            return ev.depth_to_space(_grad_image(grads[0]), ctx.table), None, None
        ga, gb = grads
        dev = ga.device if ga is not None else gb.device
        if ga is None:
            ga = torch.zeros((B, ctx.split, H // 2, W // 2), dtype=torch.float32, device=dev)
        if gb is None:
            gb = torch.zeros((B, 4 * C - ctx.split, H // 2, W // 2), dtype=torch.float32, device=dev)
        return ev.depth_to_space(_grad_image(ga), ctx.table, _grad_image(gb)), None, None


class DepthToSpaceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, table):
        ctx.table = table
        ctx.split = None if b is None else a.shape[1]
        return ev.depth_to_space(a, table, b)

    @staticmethod
    def backward(ctx, g):
        g = _grad_image(g)
        if ctx.split is None:
            return ev.space_to_depth(g, ctx.table), None, None
        ga, gb = ev.space_to_depth(g, ctx.table, split=ctx.split)
        return ga, gb, None


def space_to_depth(x, table, split=None):
    return SpaceToDepthFn.apply(x, table, split)


def depth_to_space(a, table, b=None):
    return DepthToSpaceFn.apply(a, b, table)
