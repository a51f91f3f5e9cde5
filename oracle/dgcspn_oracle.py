"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's DGC-SPN path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the shipped package (``deeprob-kit_amd/``) never does.

Restates ``deeprob/spn/layers/dgcspn.py`` and ``deeprob/spn/models/dgcspn.py`` as plain functions over a
``state_dict``-like mapping.  The product layer is written as an explicit sum of window taps over a
zero-padded (or cropped) map instead of ``F.conv2d`` with ones / one-hot kernels; the other layers
use the same ATen ops as the reference.

Pinned: ``tests/test_oracle_dgcspn.py`` checks every function against the golden vectors in
``tests/golden/dgcspn_*.npz`` that ``tools/gen_golden.py --only dgcspn`` produced by importing the
reference in the build container (outputs, per-layer activations, MPE completions, gradients), and
against the reference's own invariants (tests/test_dgcspn.py:46-75: all-ones input => 4.0 inside).
"""
import math
from itertools import product as iproduct
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def spatial_gaussian(x, loc, scale, drop=None):
    """SpatialGaussianLayer.forward (layers/dgcspn.py:101-120); ``drop`` (bool [B,K,C,H,W]) = the training-mode
    dropout mask of :113-114."""
    v = torch.unsqueeze(x, dim=1)                                             # [B,1,C,H,W]
    lp = -((v - loc) ** 2) / (2 * scale ** 2) - torch.log(scale) - math.log(math.sqrt(2 * math.pi))
    if drop is not None:
        lp = torch.where(drop, torch.full_like(lp, float('nan')), lp)
    lp = torch.nan_to_num(lp)
    return torch.sum(lp, dim=2)


def product_geometry(in_features: Tuple[int, int, int], padding: str, stride: int, dilation: int,
                     depthwise: bool, kernel: int = 2):
    """Pad list and output shape of a SpatialProductLayer (layers/dgcspn.py:160-184)."""
    c, h, w = in_features
    ke = (kernel - 1) * dilation + 1
    if padding == 'valid':
        pad = [0, 0, 0, 0]
    elif padding == 'full':
        pad = [ke - 1] * 4
    elif padding == 'final':
        pad = [0, (ke - 1) * 2 - w, 0, (ke - 1) * 2 - h]
    else:
        raise ValueError(padding)
    oh = int(np.ceil((pad[2] + pad[3] + h - ke + 1) / stride))
    ow = int(np.ceil((pad[0] + pad[1] + w - ke + 1) / stride))
    oc = c if depthwise else c ** (kernel * kernel)
    return pad, (oc, oh, ow)


def spatial_product(x, pad: Sequence[int], stride: int, dilation: int, depthwise: bool, kernel: int = 2):
    """SpatialProductLayer.forward (layers/dgcspn.py:224-236): F.pad, then the window sum that the
    ones / one-hot conv2d kernels (:187-195) compute."""
    xp = F.pad(x, list(pad))
    b, c, hp, wp = xp.shape
    ke = (kernel - 1) * dilation + 1
    oh = (hp - ke) // stride + 1
    ow = (wp - ke) // stride + 1
    taps = []
    for th in range(kernel):
        for tw in range(kernel):
            h0, w0 = th * dilation, tw * dilation
            taps.append(xp[:, :, h0:h0 + (oh - 1) * stride + 1:stride, w0:w0 + (ow - 1) * stride + 1:stride])
    if depthwise:
        out = taps[0]
        for t in taps[1:]:
            out = out + t
        return out
    combos = list(iproduct(range(c), repeat=kernel * kernel))               # :189 (itertools.product order)
    out = torch.empty(b, len(combos), oh, ow, dtype=x.dtype)
    for oc, ids in enumerate(combos):
        acc = taps[0][:, ids[0]]
        for t in range(1, len(taps)):
            acc = acc + taps[t][:, ids[t]]
        out[:, oc] = acc
    return out


def spatial_sum(x, weight, drop=None):
    """SpatialSumLayer.forward (layers/dgcspn.py:289-304); ``drop`` (bool, shape of x) = dropout mask of :297-298."""
    if drop is not None:
        x = x.masked_fill(drop, float('-inf'))
    w = torch.log_softmax(weight, dim=1)
    return torch.logsumexp(torch.unsqueeze(x, dim=1) + w, dim=2)


def spatial_root(x, weight):
    """SpatialRootLayer.forward (layers/dgcspn.py:343-355)."""
    v = torch.flatten(x, start_dim=1)
    w = torch.log_softmax(weight, dim=1)
    return torch.logsumexp(torch.unsqueeze(v, dim=1) + w, dim=2)


def schedule(in_features: Tuple[int, int, int], n_batch: int, sum_channels: int, depthwise, n_pooling: int):
    """Layer schedule of DgcSpn.__init__ (models/dgcspn.py:68-128): list of ('prod', pad, stride, dilation,
    depthwise) / ('sum',) entries in ``model.layers`` order."""
    depth = int(np.ceil(np.log2(in_features[1])))
    if isinstance(depthwise, bool):
        depthwise = [depthwise] * (depth + 1)
    else:
        depthwise = list(depthwise) + [depthwise[-1]] * (depth + 1 - len(depthwise))
    shape = (n_batch, in_features[1], in_features[2])
    plan = []
    for i in range(depth + 1):
        if i < n_pooling:
            padding, stride, dilation = 'valid', 2, 1
        else:
            padding, stride, dilation = ('final' if i == depth else 'full'), 1, 2 ** (i - n_pooling)
        pad, shape = product_geometry(shape, padding, stride, dilation, depthwise[i])
        plan.append(('prod', pad, stride, dilation, depthwise[i]))
        if i != depth:
            plan.append(('sum',))
            shape = (sum_channels,) + shape[1:]
    return plan


def dgcspn_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, plan, return_activations: bool = False,
                   z: Optional[torch.Tensor] = None, drops: Optional[Dict[str, torch.Tensor]] = None):
    """DgcSpn.forward (models/dgcspn.py:134-151).  ``z``: start from given leaf outputs (for mpe); ``drops``:
    training-mode dropout masks {'leaf': [B,K,C,H,W], 'layers.<i>': shape of that sum layer's input}."""
    drops = drops or {}
    acts = []
    h = spatial_gaussian(x, sd['base_layer.loc'], sd['base_layer.scale'], drops.get('leaf')) if z is None else z
    acts.append(h)
    for i, step in enumerate(plan):
        if step[0] == 'prod':
            _, pad, stride, dilation, dw = step
            h = spatial_product(h, pad, stride, dilation, dw)
        else:
            h = spatial_sum(h, sd['layers.{}.weight'.format(i)], drops.get('layers.{}'.format(i)))
        acts.append(h)
    out = spatial_root(h, sd['root_layer.weight'])
    return (out, acts) if return_activations else out


def dgcspn_mpe(sd, x, plan):
    """DgcSpn.mpe (models/dgcspn.py:153-184)."""
    with torch.enable_grad():
        z = spatial_gaussian(x, sd['base_layer.loc'], sd['base_layer.scale']).detach().requires_grad_(True)
        y = dgcspn_forward(sd, x, plan, z=z)
        z_grad, = torch.autograd.grad(y, z, grad_outputs=torch.ones_like(y))
    est = torch.sum(torch.unsqueeze(z_grad, dim=2) * sd['base_layer.loc'], dim=1)
    return torch.where(torch.isnan(x), est, x)


def dgcspn_loss(out, y=None):
    """DgcSpn.loss (models/dgcspn.py:189-195)."""
    if y is None:
        return -torch.mean(out)
    return F.nll_loss(torch.log_softmax(out, dim=1), y)
