"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's RealNVP-2D evaluation path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Op-for-op restatement (same ATen op sequence, eval mode) of
  deeprob/torch/utils.py:86-121               WeightNormConv2d (torch.nn.utils.weight_norm over dim 0)
  deeprob/flows/layers/resnet.py:9-90         ResidualBlock / ResidualNetwork
  deeprob/flows/layers/densenet.py            DenseLayer / DenseBlock / Transition / DenseNetwork
  deeprob/flows/layers/coupling.py:181-272    CouplingLayer2d.apply_backward / apply_forward
  deeprob/flows/layers/coupling.py:366-408    CouplingBlock2d.apply_backward / apply_forward
  deeprob/flows/utils.py:11-38                squeeze_depth2d / unsqueeze_depth2d
  deeprob/flows/utils.py:186-222              BatchNormLayer2d (running statistics)
  deeprob/flows/models/realnvp.py:164-220     RealNVP2d.apply_backward / apply_forward
  deeprob/flows/models/base.py:123-143        NormalizingFlow.forward
as plain functions over a state_dict (the model structure is read off the key names).  Pinned by
tests/test_oracle_flows2d.py against golden vectors that tools/gen_golden_flows2d.py produced from the imported
reference (latents, log-dets, LLs, inverse round trip; resnet / densenet, affine / NICE).

Inside ``with training():`` the batch-normalisation layers use batch statistics as the reference's modules do in
training mode (nn.BatchNorm2d training branch; flows/utils.py:190-198) -- without the running-statistics update, which
the golden fixtures pin directly -- and torch's autograd over these ops (state tensors with requires_grad) gives the
reference gradients; pinned against the reference's own training-mode LLs and gradients (``*_train.npz``).
"""
import contextlib
import math
from typing import Dict

import torch
import torch.nn.functional as F


_MODE = {'train': False, 'margins': None}


@contextlib.contextmanager
def training():
    """Batch statistics instead of running statistics (the modules' training mode)."""
    _MODE['train'] = True
    try:
        yield
    finally:
        _MODE['train'] = False


@contextlib.contextmanager
def relu_margins():
    """Collects the smallest |ReLU argument| of every BatchNorm2d + ReLU evaluated inside the block.  The gradient of
    the network is discontinuous where an argument crosses zero, and two fp32 evaluations of the same network disagree
    on the sign of arguments within rounding of zero; tests use the margin to know which comparisons can be strict."""
    _MODE['margins'] = []
    try:
        yield _MODE['margins']
    finally:
        _MODE['margins'] = None


def _wn_conv(sd, p, x):
    """WeightNormConv2d: w = g * v / ||v|| per output channel (torch _weight_norm, dim 0)."""
    v, g = sd[p + 'conv.weight_v'], sd[p + 'conv.weight_g']
    w = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1, 1))
    return F.conv2d(x, w, sd.get(p + 'conv.bias'), padding=v.shape[2] // 2)


def _bn_relu(sd, p, x):
    """nn.BatchNorm2d + nn.ReLU."""
    if _MODE['train']:
        y = F.batch_norm(x, None, None, sd[p + 'weight'], sd[p + 'bias'], True, 0.1, 1e-5)
    else:
        y = F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                         False, 0.1, 1e-5)
    if _MODE['margins'] is not None:
        _MODE['margins'].append(float(y.detach().abs().min()))
    return torch.relu(y)


def resnet(sd, p, x):
    """resnet.py:72-90."""
    x = _wn_conv(sd, p + 'in_conv.', x)
    z = _wn_conv(sd, p + 'in_skip.', x)
    i = 0
    while p + 'skips.{}.conv.weight_v'.format(i) in sd:
        b = p + 'blocks.{}.block.'.format(i)
        h = _wn_conv(sd, b + '2.', _bn_relu(sd, b + '0.', x))
        x = x + _wn_conv(sd, b + '5.', _bn_relu(sd, b + '3.', h))
        z = z + _wn_conv(sd, p + 'skips.{}.'.format(i), x)
        i += 1
    return _wn_conv(sd, p + 'out_network.2.', _bn_relu(sd, p + 'out_network.0.', z))


def densenet(sd, p, x):
    """densenet.py:176-189."""
    x = _wn_conv(sd, p + 'in_conv.', x)
    i = 0
    while p + 'blocks.{}.'.format(i) + 'layers.0.network.2.conv.weight_v' in sd:
        b = p + 'blocks.{}.'.format(i)
        outputs = [x]
        j = 0
        while b + 'layers.{}.network.2.conv.weight_v'.format(j) in sd:
            q = b + 'layers.{}.'.format(j)
            h = torch.cat(outputs, dim=1)
            h = _wn_conv(sd, q + 'bottleneck_network.2.', _bn_relu(sd, q + 'bottleneck_network.0.', h))
            outputs.append(_wn_conv(sd, q + 'network.2.', _bn_relu(sd, q + 'network.0.', h)))
            j += 1
        x = torch.cat(outputs, dim=1)
        t = p + 'blocks.{}.network.'.format(i + 1)
        x = _wn_conv(sd, t + '2.', _bn_relu(sd, t + '0.', x))
        i += 2
    return x


def _conditioner(sd, p, x):
    return (resnet if p + 'network.in_skip.conv.weight_v' in sd else densenet)(sd, p + 'network.', x)


def coupling(sd, p, x, channelwise: bool, reverse: bool, inverse: bool):
    """coupling.py:181-226 (inverse False) / :228-272 (inverse True); returns (out, log-det term)."""
    n = x.shape[0]
    act = sd.get(p + 'scale_act.weight')
    if channelwise:
        if reverse:
            mx, my = torch.chunk(x, chunks=2, dim=1)
        else:
            my, mx = torch.chunk(x, chunks=2, dim=1)
        z = _conditioner(sd, p, mx)
        if act is not None:
            t, s = torch.chunk(z, chunks=2, dim=1)
            s = act * torch.tanh(s)
            my = my * torch.exp(s) + t if inverse else (my - t) * torch.exp(-s)
            ldj = torch.sum(s.reshape(n, -1), dim=1)
            ldj = ldj if inverse else -ldj
        else:
            my = my + z if inverse else my - z
            ldj = torch.zeros(n, dtype=x.dtype)
        return (torch.cat([mx, my], dim=1) if reverse else torch.cat([my, mx], dim=1)), ldj
    mask, inv_mask = sd[p + 'mask'], sd[p + 'inv_mask']
    z = _conditioner(sd, p, mask * x)
    if act is not None:
        t, s = torch.chunk(z, chunks=2, dim=1)
        s = act * torch.tanh(s)
        t = inv_mask * t
        s = inv_mask * s
        out = x * torch.exp(s) + t if inverse else (x - t) * torch.exp(-s)
        ldj = torch.sum(s.reshape(n, -1), dim=1)
        return out, (ldj if inverse else -ldj)
    t = inv_mask * z
    return (x + t if inverse else x - t), torch.zeros(n, dtype=x.dtype)


def bn2d(sd, p, x, inverse: bool, eps: float = 1e-5):
    """flows/utils.py:186-222 with the running statistics."""
    n, grid = x.shape[0], x.shape[2] * x.shape[3]
    w, b, var, mean = sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_var'] + eps, sd[p + 'running_mean']
    if _MODE['train'] and not inverse:
        mean = torch.mean(x, dim=[0, 2, 3], keepdim=True)
        var = torch.mean((x - mean) ** 2.0, dim=[0, 2, 3], keepdim=True) + eps
    if inverse:
        u = (x - b) * torch.exp(-w)
        return u * torch.sqrt(var) + mean, (torch.sum(0.5 * torch.log(var) - w) * grid).expand(n)
    u = (x - mean) / torch.sqrt(var)
    return u * torch.exp(w) + b, (torch.sum(w - 0.5 * torch.log(var)) * grid).expand(n)


def squeeze(x):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, c * 4, h // 2, w // 2)


def unsqueeze(x):
    n, c, h, w = x.shape
    return x.reshape(n, c // 4, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c // 4, h * 2, w * 2)


def _bijectors(sd, p, group):
    """[(prefix, is_coupling, reverse)] of `in_couplings` / `out_couplings` of block p (couplings at even positions)."""
    out, i = [], 0
    while True:
        q = '{}{}.{}.'.format(p, group, i)
        if q + 'weight' not in sd and q + 'network.in_conv.conv.weight_v' not in sd:
            return out
        out.append((q, i % 2 == 0, (i // 2) % 2 == 1))
        i += 1


def block(sd, p, x, inverse: bool):
    """coupling.py:366-387 / :389-408; the block is a last block iff it has no out_couplings."""
    n = x.shape[0]
    total = torch.zeros(n, dtype=x.dtype)
    ins, outs = _bijectors(sd, p, 'in_couplings'), _bijectors(sd, p, 'out_couplings')

    def run(layers, channelwise, x, total):
        for q, is_coupling, reverse in (reversed(layers) if inverse else layers):
            x, d = coupling(sd, q, x, channelwise, reverse, inverse) if is_coupling else bn2d(sd, q, x, inverse)
            total = total + d
        return x, total

    if not inverse:
        x, total = run(ins, False, x, total)
        if outs:
            x, total = run(outs, True, squeeze(x), total)
            x = unsqueeze(x)
    else:
        if outs:
            x, total = run(outs, True, squeeze(x), total)
            x = unsqueeze(x)
        x, total = run(ins, False, x, total)
    return x, total


def n_blocks(sd) -> int:
    i = 0
    while 'layers.{}.in_couplings.0.network.in_conv.conv.weight_v'.format(i) in sd:
        i += 1
    return i


def apply_backward(sd: Dict[str, torch.Tensor], x):
    """realnvp.py:164-193."""
    nb = n_blocks(sd)
    total = torch.zeros(x.shape[0], dtype=x.dtype)
    slices = []
    for i in range(nb):
        x, d = block(sd, 'layers.{}.'.format(i), x, False)
        total = total + d
        if i != nb - 1:
            x = F.conv2d(x, sd['perm_matrices.{}'.format(i)], stride=2)
            x, z = torch.chunk(x, chunks=2, dim=1)
            slices.append(z)
    for i in range(nb - 2, -1, -1):
        x = F.conv_transpose2d(torch.cat([x, slices[i]], dim=1), sd['perm_matrices.{}'.format(i)], stride=2)
    return x, total


def apply_forward(sd: Dict[str, torch.Tensor], x):
    """realnvp.py:195-220."""
    nb = n_blocks(sd)
    total = torch.zeros(x.shape[0], dtype=x.dtype)
    slices = []
    for i in range(nb - 1):
        x = F.conv2d(x, sd['perm_matrices.{}'.format(i)], stride=2)
        x, z = torch.chunk(x, chunks=2, dim=1)
        slices.append(z)
    for i in range(nb - 1, -1, -1):
        if i != nb - 1:
            x = F.conv_transpose2d(torch.cat([x, slices[i]], dim=1), sd['perm_matrices.{}'.format(i)], stride=2)
        x, d = block(sd, 'layers.{}.'.format(i), x, True)
        total = total + d
    return x, total


def log_prob(sd: Dict[str, torch.Tensor], x, logit_alpha=None):
    """base.py:123-143 with the default Normal base (and the LogitLayer of flows/utils.py:276-284 when asked)."""
    n = x.shape[0]
    total = torch.zeros(n, dtype=x.dtype)
    if logit_alpha is not None:
        p = logit_alpha + (1.0 - 2.0 * logit_alpha) * x
        lp, lq = torch.log(p), torch.log(1.0 - p)
        total = total - (torch.sum((lp + lq).reshape(n, -1), dim=1) + sd['logit.ldj'])
        x = lp - lq
    u, d = apply_backward(sd, x)
    loc, scale = sd['in_base_loc'], sd['in_base_scale']
    base = -((u - loc) ** 2) / (2 * scale ** 2) - torch.log(scale) - math.log(math.sqrt(2 * math.pi))
    return torch.sum(base.reshape(n, -1), dim=1) + total + d
