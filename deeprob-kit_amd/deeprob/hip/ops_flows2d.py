"""RealNVP-2D operators on top of the C ABI (csrc/flows2d.hip): conditioner convolutions with the eval-mode
BatchNorm2d + ReLU folded into the operand load, coupling transformation, BatchNormLayer2d bijector and the
squeeze / multi-scale permutations.

The functions here evaluate (density and sampling directions, running statistics, no autograd graph).  When a module is
in training mode or a gradient through it is wanted they hand over to the autograd nodes of
``deeprob.hip.ops_flows2d_train`` (density direction only; the sampling direction with gradients raises).  Nothing falls
back to torch operators on image tensors.
"""
from typing import Optional, Tuple

import torch

from deeprob.hip import load_library, check, ptr, stream_ptr, require_device_f32, HipError


def graph_route(x, *modules) -> bool:
    """True when the call must build an autograd graph or use batch statistics: a module in training mode, or grad mode
    with an input / parameter that requires grad."""
    modules = [m for m in modules if m is not None]
    if any(m.training for m in modules):
        return True
    if not torch.is_grad_enabled():
        return False
    return (torch.is_tensor(x) and x.requires_grad) or any(p.requires_grad for m in modules for p in m.parameters())


def require_eval(module, what: str, *tensors):
    """The sampling direction of the 2-D flows evaluates with running statistics and has no backward."""
    if module.training:
        raise HipError("{}: the sampling direction of the HIP RealNVP-2D path is evaluation only (call .eval(); only "
                       "the density direction is built for training)".format(what))
    if torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors) or
                                    any(p.requires_grad for p in module.parameters())):
        raise HipError("{}: the sampling direction of the HIP RealNVP-2D path has no backward: wrap the call in "
                       "torch.no_grad() or freeze the parameters".format(what))


def _image(t: torch.Tensor, name: str) -> torch.Tensor:
    """A [B, C, H, W] fp32 device tensor whose samples are dense [C, H, W] blocks (a channel slice of a larger
    contiguous tensor qualifies: only the batch stride differs)."""
    if not t.is_cuda:
        raise HipError("{} lives on '{}': the deeprob HIP path only evaluates tensors on a HIP device "
                       "(there is no CPU fallback)".format(name, t.device))
    if t.dim() != 4:
        raise HipError("{} must be [B, C, H, W], got {}".format(name, tuple(t.shape)))
    if t.dtype != torch.float32:
        t = t.float()
    _, C, H, W = t.shape
    if t.stride(3) != 1 or t.stride(2) != W or t.stride(1) != H * W or t.stride(0) < C * H * W:
        t = t.contiguous()
    return t


def _key(*tensors) -> tuple:
    return tuple(None if t is None else (t.data_ptr(), t._version) for t in tensors)


def conv_tables(conv, bn=None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Packed effective weights of a weight-normalised convolution (and the folded BatchNorm2d in front of it); rebuilt
    when a parameter or running statistic changed (address / `_version`), re-packed in place on every other call."""
    p = conv.conv
    tensors = [p.weight_v, p.weight_g]
    if bn is not None:
        tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = _key(*tensors)
    hit = getattr(conv, '_hip_tables', None)
    from deeprob import hip
    if hit is not None and hit[0] == key and hip._trust_versions:
        return hit[1], hit[2]
    lib = load_library()
    v = require_device_f32(p.weight_v.detach(), 'weight_v')
    g = require_device_f32(p.weight_g.detach(), 'weight_g')
    cout, cin, ks = v.shape[0], v.shape[1], v.shape[2]
    pre, bnp = None, [None] * 4
    eps = 0.0
    if bn is not None:
        if bn.weight is None or bn.running_mean is None:
            raise HipError("conv2d: BatchNorm2d without affine parameters / running statistics is not built")
        bnp = [require_device_f32(t.detach(), 'batch norm') for t in (bn.weight, bn.bias, bn.running_mean,
                                                                       bn.running_var)]
        eps = float(bn.eps)
    if hit is not None and hit[0] == key:
        # same addresses and version counters: the pack kernels run again INTO the same tables (two small launches per
        # convolution), so that a write through `.data` is seen; hip.trust_version_counters(True) skips this
        wpack, pre = hit[1], hit[2]
    else:
        wpack = torch.empty(lib.dpk_conv2d_pack_floats(cout, cin, ks), dtype=torch.float32, device=v.device)
        if bn is not None:
            pre = torch.empty(2 * cin, dtype=torch.float32, device=v.device)
    check(lib.dpk_conv2d_prepare(ptr(v), ptr(g), cout, cin, ks, ptr(bnp[0]), ptr(bnp[1]), ptr(bnp[2]), ptr(bnp[3]),
                                 eps, ptr(wpack), ptr(pre), stream_ptr(v.device)), 'dpk_conv2d_prepare')
    conv._hip_tables = (key, wpack, pre)
    return wpack, pre


def conv2d(x: torch.Tensor, conv, bn=None, in_mask: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``conv(relu(bn(x)))`` (bn None: ``conv(x)``; in_mask: ``conv(in_mask * x)``) ``+ res``; `out` may be a channel
    slice of a larger tensor (reference: torch/utils.py:117-121 inside flows/layers/resnet.py, densenet.py)."""
    if graph_route(x, conv, bn) or (res is not None and torch.is_grad_enabled() and res.requires_grad):
        if out is not None and out is not res:
            raise HipError("conv2d: an output slice cannot be written in place while an autograd graph is built")
        from deeprob.hip import ops_flows2d_train
        return ops_flows2d_train.conv2d(x, conv, bn, in_mask, res)
    lib = load_library()
    x = _image(x, 'x')
    B, cin, H, W = x.shape
    p = conv.conv
    cout, ks = p.weight_v.shape[0], p.weight_v.shape[2]
    if p.weight_v.shape[1] != cin:
        raise HipError("conv2d: input has {} channels, the layer expects {}".format(cin, p.weight_v.shape[1]))
    wpack, pre = conv_tables(conv, bn)
    if out is None:
        out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, cout, H, W) or _image(out, 'out') is not out:
        raise HipError("conv2d: output slice has the wrong shape or layout")
    if res is not None:
        res = _image(res, 'res')
        if tuple(res.shape) != (B, cout, H, W):
            raise HipError("conv2d: residual shape {} != {}".format(tuple(res.shape), (B, cout, H, W)))
    bias = None if p.bias is None else require_device_f32(p.bias.detach(), 'bias')
    mask = None if in_mask is None else require_device_f32(in_mask, 'mask')
    if mask is not None and mask.numel() != H * W:
        raise HipError("conv2d: mask must have H*W = {} entries".format(H * W))
    check(lib.dpk_conv2d_forward(ptr(x), x.stride(0), B, cin, H, W, ptr(wpack), cout, ks, ptr(pre), ptr(mask),
                                 ptr(bias), ptr(res), 0 if res is None else res.stride(0), ptr(out), out.stride(0),
                                 stream_ptr(x.device)), 'dpk_conv2d_forward')
    return out


def coupling2d(x: torch.Tensor, z: torch.Tensor, layer, inverse: bool,
               ldj: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The transformation of CouplingLayer2d given the conditioner output z (coupling.py:181-272)."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    z = require_device_f32(z, 'z')
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    ldj_out = torch.empty(B, dtype=torch.float32, device=x.device)
    scale = require_device_f32(layer.scale_act.weight.detach(), 'scale_act.weight').view(-1) if layer.affine else None
    inv_mask = None if layer.channelwise else require_device_f32(layer.inv_mask, 'inv_mask')
    check(lib.dpk_coupling2d_transform(ptr(x), ptr(z), ptr(scale), ptr(inv_mask), B, C, H, W, int(layer.affine),
                                       int(layer.reverse), int(inverse), ptr(ldj), ptr(out), ptr(ldj_out),
                                       stream_ptr(x.device)), 'dpk_coupling2d_transform')
    return out, ldj_out


def bn2d(x: torch.Tensor, layer, inverse: bool, ldj: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """BatchNormLayer2d with running statistics (flows/utils.py:186-222)."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    ldj_out = torch.empty(B, dtype=torch.float32, device=x.device)
    w, b, m, v = (require_device_f32(t.detach(), 'batch norm').view(-1)
                  for t in (layer.weight, layer.bias, layer.running_mean, layer.running_var))
    check(lib.dpk_bn2d_bijector(ptr(x), ptr(w), ptr(b), ptr(m), ptr(v), float(layer.eps), B, C, H, W, int(inverse),
                                ptr(ldj), ptr(out), ptr(ldj_out), stream_ptr(x.device)), 'dpk_bn2d_bijector')
    return out, ldj_out


_SQUEEZE_TABLES = {}


def squeeze_table(channels: int, device) -> torch.Tensor:
    """squeeze_depth2d's channel order: output channel c*4 + dy*2 + dx (flows/utils.py:19-22)."""
    key = (channels, str(device))
    if key not in _SQUEEZE_TABLES:
        _SQUEEZE_TABLES[key] = torch.arange(4 * channels, dtype=torch.int32, device=device)
    return _SQUEEZE_TABLES[key]


def permutation_table(matrix: torch.Tensor) -> torch.Tensor:
    """The one-hot kernel [4C, C, 2, 2] of RealNVP2d's down-scaling convolution (realnvp.py:141-162) as the table
    `c*4 + dy*2 + dx` per output channel; raises if the kernel is not a permutation."""
    m = matrix.detach().reshape(matrix.shape[0], -1)
    idx = m.argmax(dim=1)
    onehot = torch.zeros_like(m).scatter_(1, idx[:, None], 1.0)
    if not torch.equal(onehot, m) or idx.unique().numel() != m.shape[0] or m.shape[0] != m.shape[1]:
        raise HipError("RealNVP2d: the down-scaling kernel is not a permutation (only the reference's one-hot order "
                       "matrices are built)")
    return idx.to(torch.int32).contiguous()


def space_to_depth(x: torch.Tensor, table: torch.Tensor, split: Optional[int] = None):
    """[B,C,H,W] -> [B,4C,H/2,W/2] in the channel order of `table`; with `split`, the two channel groups
    [0, split) and [split, 4C) as separate tensors (the torch.chunk of the multi-scale architecture)."""
    if torch.is_grad_enabled() and x.requires_grad:
        from deeprob.hip import ops_flows2d_train
        return ops_flows2d_train.space_to_depth(x, table, split)
    lib = load_library()
    x = require_device_f32(x, 'x')
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise HipError("squeeze: H and W must be even, got {}x{}".format(H, W))
    ca = 4 * C if split is None else split
    a = torch.empty((B, ca, H // 2, W // 2), dtype=torch.float32, device=x.device)
    b = None if split is None else torch.empty((B, 4 * C - ca, H // 2, W // 2), dtype=torch.float32, device=x.device)
    check(lib.dpk_space_to_depth(ptr(x), B, C, H, W, ptr(table), ptr(a), ca, ptr(b), stream_ptr(x.device)),
          'dpk_space_to_depth')
    return a if split is None else (a, b)


def depth_to_space(a: torch.Tensor, table: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse of :func:`space_to_depth` (of the concatenation [a | b] when b is given)."""
    if torch.is_grad_enabled() and (a.requires_grad or (b is not None and b.requires_grad)):
        from deeprob.hip import ops_flows2d_train
        return ops_flows2d_train.depth_to_space(a, table, b)
    lib = load_library()
    a = require_device_f32(a, 'x')
    B, ca, h, w = a.shape
    total = ca
    if b is not None:
        b = require_device_f32(b, 'x')
        total += b.shape[1]
    if total % 4:
        raise HipError("unsqueeze: the channel count {} is not a multiple of 4".format(total))
    C = total // 4
    out = torch.empty((B, C, 2 * h, 2 * w), dtype=torch.float32, device=a.device)
    check(lib.dpk_depth_to_space(ptr(a), ca, ptr(b), B, C, 2 * h, 2 * w, ptr(table), ptr(out), stream_ptr(a.device)),
          'dpk_depth_to_space')
    return out
