"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's RealNVP-1D path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Op-for-op restatement (same ATen op sequence) of
  deeprob/flows/layers/coupling.py:72-104   CouplingLayer1d.apply_backward / apply_forward
  deeprob/flows/utils.py:118-153            BatchNormLayer1d (eval mode)
  deeprob/flows/utils.py:276-294            LogitLayer
  deeprob/flows/models/base.py:123-143      NormalizingFlow.forward
as plain functions over a state_dict.  Pinned by tests/test_oracle_flows.py against golden vectors that
tools/gen_golden_flows.py produced from the imported reference (per-layer outputs, log-dets, LLs, inverse
round trip) and against the reference's invertibility test (tests/test_flows.py:22-26, atol 5e-7).
"""
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


def coupling_params(sd: Dict[str, torch.Tensor], i: int):
    p = 'layers.{}.'.format(i)
    lins = []
    j = 0
    while p + 'network.{}.weight'.format(j) in sd:
        lins.append((sd[p + 'network.{}.weight'.format(j)], sd[p + 'network.{}.bias'.format(j)]))
        j += 2
    return sd[p + 'mask'], sd[p + 'inv_mask'], lins, sd.get(p + 'scale_act.weight')


def _network(h, lins):
    """nn.Sequential(Linear, ReLU, ..., Linear) (coupling.py:45-56)."""
    for w, b in lins[:-1]:
        h = torch.relu(F.linear(h, w, b))
    return F.linear(h, *lins[-1])


def coupling_backward(x, mask, inv_mask, lins, act_w):
    """coupling.py:72-87."""
    z = _network(mask * x, lins)
    if act_w is not None:
        t, s = torch.chunk(z, chunks=2, dim=1)
        s = act_w * torch.tanh(s)
        t = inv_mask * t
        s = inv_mask * s
        return (x - t) * torch.exp(-s), -torch.sum(s, dim=1)
    return x - inv_mask * z, torch.zeros(x.shape[0], dtype=x.dtype)


def coupling_forward(u, mask, inv_mask, lins, act_w):
    """coupling.py:89-104."""
    z = _network(mask * u, lins)
    if act_w is not None:
        t, s = torch.chunk(z, chunks=2, dim=1)
        s = act_w * torch.tanh(s)
        t = inv_mask * t
        s = inv_mask * s
        return u * torch.exp(s) + t, torch.sum(s, dim=1)
    return u + inv_mask * z, torch.zeros(u.shape[0], dtype=u.dtype)


def bn_backward(x, weight, bias, var, mean, eps=1e-5):
    """flows/utils.py:118-139, eval branch."""
    var = var + eps
    u = (x - mean) / torch.sqrt(var)
    u = u * torch.exp(weight) + bias
    return u, torch.sum(weight - 0.5 * torch.log(var)).expand(x.shape[0])


def bn_backward_train(x, weight, bias, running_var, running_mean, momentum=0.9, eps=1e-5):
    """flows/utils.py:118-139, training branch: batch var_mean (unbiased), running statistics updated with
    `momentum`.  Returns (u, ildj, new_running_var, new_running_mean)."""
    var, mean = torch.var_mean(x, dim=0, keepdim=True)
    new_var = running_var * momentum + var.detach() * (1.0 - momentum)
    new_mean = running_mean * momentum + mean.detach() * (1.0 - momentum)
    var = var + eps
    u = (x - mean) / torch.sqrt(var)
    u = u * torch.exp(weight) + bias
    return u, torch.sum(weight - 0.5 * torch.log(var)).expand(x.shape[0]), new_var, new_mean


class _SyncBatchNormTrain(torch.autograd.Function):
    """CPU checker of the batch-sharded training-mode BatchNormLayer1d (the product: dpk_bn1d_local_moments /
    _sync_forward / _backward_sums / _sync_backward + deeprob.parallel.bn_gather_moments / bn_reduce_sums).  Same
    protocol, torch arithmetic: every rank holds a slice of the batch, `gather` / `reduce` are the product's two
    exchange functions, and the result must equal bn_backward_train on the unsharded batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, gather, reduce):
        n, d = x.shape
        mean_r = x.mean(dim=0) if n > 0 else torch.zeros(d, dtype=x.dtype)
        m2_r = ((x - mean_r) ** 2).sum(dim=0) if n > 0 else torch.zeros(d, dtype=x.dtype)
        table = gather(torch.cat([torch.tensor([float(n)], dtype=x.dtype), mean_r, m2_r]))
        cnt, mean, m2 = 0.0, torch.zeros(d, dtype=x.dtype), torch.zeros(d, dtype=x.dtype)
        for row in table:                       # rank order, Chan et al. pairwise update
            nr = float(row[0])
            if nr > 0:
                dlt = row[1:1 + d] - mean
                tot = cnt + nr
                mean = mean + dlt * (nr / tot)
                m2 = m2 + row[1 + d:] + dlt * dlt * (cnt * nr / tot)
                cnt = tot
        var = m2 / (cnt - 1.0)
        inv = 1.0 / torch.sqrt(var + eps)
        xhat = (x - mean) * inv
        u = xhat * torch.exp(weight.reshape(-1)) + bias.reshape(-1)
        ildj = torch.sum(weight.reshape(-1) - 0.5 * torch.log(var + eps)).expand(n)
        ctx.save_for_backward(xhat, weight, inv)
        ctx.reduce, ctx.n_total = reduce, int(round(cnt))
        ctx.mark_non_differentiable(mean, var)
        return u, ildj, mean, var

    @staticmethod
    def backward(ctx, gu, gildj, _gm, _gv):
        xhat, weight, inv = ctx.saved_tensors
        n, d = xhat.shape
        ew = torch.exp(weight.reshape(-1))
        sums = torch.cat([gu.sum(dim=0), (gu * xhat).sum(dim=0), gildj.sum().reshape(1)])
        sx = ctx.reduce(sums.clone(), n, ctx.n_total)
        s1, s2, sg = sx[:d], sx[d:2 * d], sx[2 * d]
        nt = float(ctx.n_total)
        dvar = -0.5 * (ew * s2 + sg) * inv * inv
        dmean = -ew * inv * s1
        gx = gu * ew * inv + dvar * 2.0 * (xhat / inv) / (nt - 1.0) + dmean / nt
        gw = (ew * sums[d:2 * d] + sums[2 * d]).reshape(weight.shape)
        gb = sums[:d].reshape(weight.shape)
        return gx, gw, gb, None, None, None


def bn_backward_train_sync(x, weight, bias, running_var, running_mean, gather, reduce, momentum=0.9, eps=1e-5):
    """bn_backward_train for a batch sharded over ranks (statistics of the whole batch)."""
    u, ildj, mean, var = _SyncBatchNormTrain.apply(x, weight, bias, eps, gather, reduce)
    new_var = running_var * momentum + var.reshape(running_var.shape) * (1.0 - momentum)
    new_mean = running_mean * momentum + mean.reshape(running_mean.shape) * (1.0 - momentum)
    return u, ildj, new_var, new_mean


def bn_forward(u, weight, bias, var, mean, eps=1e-5):
    """flows/utils.py:141-153."""
    var = var + eps
    u = (u - bias) * torch.exp(-weight)
    return u * torch.sqrt(var) + mean, torch.sum(-weight + 0.5 * torch.log(var)).expand(u.shape[0])


def logit_forward(u, alpha, ldj_const):
    """LogitLayer.apply_forward, flows/utils.py:286-294."""
    n = u.shape[0]
    u = torch.sigmoid(u)
    x = (u - alpha) / (1.0 - 2.0 * alpha)
    lu, ru = torch.log(u), torch.log(1.0 - u)
    return x, torch.sum((lu + ru).view(n, -1), dim=1) + ldj_const


def flow_sample_from(sd, u, logit_alpha=None):
    """NormalizingFlow.sample (flows/models/base.py:145-157) from a given base draw ``u``: apply_forward, then the inverse
    preprocessing -- the deterministic part of sampling (the base draw itself is torch's generator)."""
    x, _ = flow_apply_forward(sd, u)
    if logit_alpha is not None:
        x, _ = logit_forward(x, logit_alpha, sd['logit.ldj'])
    return x


def logit_backward(x, alpha, ldj_const):
    """flows/utils.py:276-284."""
    n = x.shape[0]
    x = alpha + (1.0 - 2.0 * alpha) * x
    lx, rx = torch.log(x), torch.log(1.0 - x)
    return lx - rx, -(torch.sum((lx + rx).view(n, -1), dim=1) + ldj_const)


def flow_layers(sd) -> List[Tuple[str, int]]:
    out, i = [], 0
    while True:
        p = 'layers.{}.'.format(i)
        if p + 'mask' in sd:
            out.append(('coupling', i))
        elif p + 'running_var' in sd:
            out.append(('bn', i))
        else:
            return out
        i += 1


def flow_apply_backward(sd, x, collect=None, train=False, running=None, sync=None):
    """NormalizingFlow.apply_backward (base.py:182-193).  train=True: batch-norm layers use batch statistics;
    the updated running statistics are written into the dict `running`.  sync=(gather, reduce): x is this rank's
    slice of a sharded batch and the statistics are those of the whole batch (bn_backward_train_sync)."""
    ildj = torch.zeros(x.shape[0], dtype=x.dtype)
    for kind, i in flow_layers(sd):
        p = 'layers.{}.'.format(i)
        if kind == 'coupling':
            x, d = coupling_backward(x, *coupling_params(sd, i))
        elif train:
            if sync is not None:
                x, d, nv, nm = bn_backward_train_sync(x, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_var'],
                                                      sd[p + 'running_mean'], *sync)
            else:
                x, d, nv, nm = bn_backward_train(x, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_var'],
                                                 sd[p + 'running_mean'])
            if running is not None:
                running[p + 'running_var'], running[p + 'running_mean'] = nv, nm
        else:
            x, d = bn_backward(x, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_var'], sd[p + 'running_mean'])
        ildj = ildj + d
        if collect is not None:
            collect.append((x, d))
    return x, ildj


def flow_apply_forward(sd, u):
    """NormalizingFlow.apply_forward (base.py:195-206)."""
    ldj = torch.zeros(u.shape[0], dtype=u.dtype)
    for kind, i in reversed(flow_layers(sd)):
        p = 'layers.{}.'.format(i)
        if kind == 'coupling':
            u, d = coupling_forward(u, *coupling_params(sd, i))
        else:
            u, d = bn_forward(u, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_var'], sd[p + 'running_mean'])
        ldj = ldj + d
    return u, ldj


def flow_log_prob(sd, x, logit_alpha=None, train=False, running=None, base=None, sync=None):
    """NormalizingFlow.forward (base.py:123-143) with the default Normal base, or `base(u) -> [B, 1]` (a module
    base density such as a RAT-SPN)."""
    n = x.shape[0]
    ildj = torch.zeros(n, dtype=x.dtype)
    if logit_alpha is not None:
        x, d = logit_backward(x, logit_alpha, sd['logit.ldj'])
        ildj = ildj + d
    u, d = flow_apply_backward(sd, x, train=train, running=running, sync=sync)
    ildj = ildj + d
    if base is not None:
        return torch.sum(base(u).view(n, -1), dim=1) + ildj
    loc, scale = sd['in_base_loc'], sd['in_base_scale']
    lp = -((u - loc) ** 2) / (2 * scale ** 2) - scale.log() - math.log(math.sqrt(2 * math.pi))
    return torch.sum(lp.view(n, -1), dim=1) + ildj
