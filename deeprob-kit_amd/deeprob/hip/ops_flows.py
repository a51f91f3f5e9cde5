"""Flow operators on top of the C ABI: RealNVP-1D coupling, folded eval-mode batch norm, Normal base.

Forward (density / sampling) only in this round: the kernels have no backward yet, so calling them
while autograd would need a graph raises instead of silently returning a detached result.
"""
from typing import Optional, Tuple

import torch

from deeprob.hip import load_library, check, ptr, stream_ptr, require_device_f32, Workspace, HipError


def _no_graph(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise HipError(
            "the HIP flow kernels are forward-only in this round (no backward): wrap the call in "
            "torch.no_grad() / model.eval(), or freeze the parameters"
        )


def coupling1d(x: torch.Tensor, layer, inverse: bool, in_affine: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               ldj: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """CouplingLayer1d.apply_backward / apply_forward (reference: deeprob/flows/layers/coupling.py:72-104)."""
    lib = load_library()
    x = require_device_f32(x, 'x')
    lin1, lin2 = layer.network[0], layer.network[-1]
    _no_graph(x, lin1.weight, lin2.weight)
    if len(layer.network) != 3:
        raise HipError("CouplingLayer1d on the HIP path supports conditioner depth 1 (got {} hidden layers)"
                       .format((len(layer.network) - 1) // 2))
    B, D = x.shape
    units = lin1.weight.shape[0]
    n_masked, n_trans = layer._mask_counts()
    n = lib.dpk_coupling1d_workspace_bytes(D, units, n_masked, n_trans)
    if n < 0:
        check(int(n), 'dpk_coupling1d_workspace_bytes')
    ws = layer._ws.get(n, x.device)
    out = torch.empty_like(x)
    accumulate = ldj is not None
    if ldj is None:
        ldj = torch.empty(B, dtype=torch.float32, device=x.device)
    sc, sh = in_affine if in_affine is not None else (None, None)
    act = layer.scale_act.weight if layer.affine else None
    check(lib.dpk_coupling1d_forward(
        ptr(x), B, D, ptr(layer.mask), ptr(layer.inv_mask), n_masked, n_trans,
        ptr(require_device_f32(lin1.weight, 'W1')), ptr(require_device_f32(lin1.bias, 'b1')),
        ptr(require_device_f32(lin2.weight, 'W2')), ptr(require_device_f32(lin2.bias, 'b2')), units,
        ptr(act), ptr(sc), ptr(sh), int(layer.affine), int(inverse), ptr(out), ptr(ldj), int(accumulate),
        ptr(ws), ws.numel(), stream_ptr(x.device)), 'dpk_coupling1d_forward')
    return out, ldj


def bn1d_fold(bn, inverse: bool, in_affine=None, ldj_const: Optional[torch.Tensor] = None):
    """Eval-mode BatchNormLayer1d as a per-variable affine (reference: deeprob/flows/utils.py:118-153).
    Returns ((scale, shift), ldj_const[1])."""
    lib = load_library()
    D = bn.in_features
    dev = bn.weight.device
    sc = torch.empty(D, dtype=torch.float32, device=dev)
    sh = torch.empty(D, dtype=torch.float32, device=dev)
    accumulate = ldj_const is not None
    if ldj_const is None:
        ldj_const = torch.empty(1, dtype=torch.float32, device=dev)
    s_in, h_in = in_affine if in_affine is not None else (None, None)
    check(lib.dpk_bn1d_fold(ptr(require_device_f32(bn.weight, 'weight')), ptr(require_device_f32(bn.bias, 'bias')),
                            ptr(require_device_f32(bn.running_var, 'running_var')),
                            ptr(require_device_f32(bn.running_mean, 'running_mean')), float(bn.eps), D,
                            int(inverse), ptr(s_in), ptr(h_in), ptr(sc), ptr(sh), ptr(ldj_const), int(accumulate),
                            stream_ptr(dev)), 'dpk_bn1d_fold')
    return (sc, sh), ldj_const


def affine1d(x: torch.Tensor, affine) -> torch.Tensor:
    lib = load_library()
    x = require_device_f32(x, 'x')
    out = torch.empty_like(x)
    B, D = x.shape
    check(lib.dpk_affine1d_forward(ptr(x), ptr(affine[0]), ptr(affine[1]), B, D, ptr(out), stream_ptr(x.device)),
          'dpk_affine1d_forward')
    return out


def normal_base_logprob(u: torch.Tensor, affine, loc, scale, ildj, ildj_const) -> torch.Tensor:
    """sum_d log N(affine(u); loc, scale) + ildj + ildj_const (reference: flows/models/base.py:139-143)."""
    lib = load_library()
    u = require_device_f32(u, 'u')
    B, D = u.shape
    out = torch.empty(B, dtype=torch.float32, device=u.device)
    sc, sh = affine if affine is not None else (None, None)
    check(lib.dpk_normal_base_logprob(ptr(u), ptr(sc), ptr(sh), ptr(require_device_f32(loc, 'loc')),
                                      ptr(require_device_f32(scale, 'scale')), ptr(ildj), ptr(ildj_const), B, D,
                                      ptr(out), stream_ptr(u.device)), 'dpk_normal_base_logprob')
    return out
