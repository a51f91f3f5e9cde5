"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's RAT-SPN path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the shipped package (``deeprob-kit_amd/``) never does.

It restates, op for op, what the reference executes on the CPU for
``deeprob/spn/layers/ratspn.py`` and ``deeprob/spn/models/ratspn.py`` (PyTorch ATen ops in the same
order, materialising the same ``[B,R,I,d]`` / ``[B,P,S,N]`` temporaries), as plain functions over a
``state_dict``-like mapping of tensors.  Because it is the same op sequence it doubles as the
"port" CPU baseline timed by ``bench.py``.

Pinned: ``tests/test_oracle_ratspn.py`` checks every function here against the golden vectors in
``tests/golden/`` that ``tools/gen_golden.py`` produced by importing the reference itself in the
build container (outputs, per-layer activations, loss and gradients), and against the reference's
own known-answer test (Bernoulli RAT-SPN normalisation, reference tests/test_ratspn.py:46-48).
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# structure (host side)
# --------------------------------------------------------------------------------------------
def region_graph_layers(n_features: int, depth: int, n_repetitions: int, seed) -> List[list]:
    """RegionGraph.random_layers / make_layers (deeprob/utils/region.py:55-99)."""
    rs = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    graph = [[tuple(range(n_features))]] + [[] for _ in range(2 * depth)]
    for _ in range(n_repetitions):
        layers = [[tuple(range(n_features))]]
        for i in range(depth):
            regions, partitions = [], []
            for r in layers[2 * i]:
                mid = len(r) // 2
                perm = rs.permutation(r).tolist()          # region.py:69
                p0, p1 = tuple(sorted(perm[:mid])), tuple(sorted(perm[mid:]))
                regions += [p0, p1]
                partitions.append((p0, p1))
            layers += [partitions, regions]
        for h in range(1, len(layers)):
            graph[h] = graph[h] + layers[h]
    return graph


def leaf_masks(regions: Sequence[tuple], in_features: int, depth: int):
    """RegionGraphLayer.__init__ buffers (deeprob/spn/layers/ratspn.py:42-66).
    Returns (mask int64 [R,d], pad_mask bool [R,1,d] or None)."""
    pad = -in_features % (2 ** depth)
    dim = (in_features + pad) // (2 ** depth)
    rows = [tuple(r) for r in regions]
    pad_mask = None
    if pad > 0:
        pad_mask = np.zeros((len(rows), 1, dim), dtype=np.bool_)
        for i, region in enumerate(rows):
            n_dummy = dim - len(region)
            if n_dummy > 0:
                pad_mask[i, :, -n_dummy:] = True
                rows[i] = region + (region[-1],) * n_dummy
        pad_mask = torch.tensor(pad_mask)
    return torch.tensor(rows), pad_mask


# --------------------------------------------------------------------------------------------
# layers (op-for-op)
# --------------------------------------------------------------------------------------------
def _normal_log_prob(value, loc, scale):
    """torch.distributions.Normal.log_prob as called at ratspn.py:96."""
    var = scale ** 2
    log_scale = scale.log()
    return -((value - loc) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))


def gaussian_leaf(x, mask, pad_mask, loc, scale, drop=None):
    """RegionGraphLayer.forward with GaussianLayer (ratspn.py:87-108).  ``drop`` (bool [B,R,I,d]) plays the
    training-mode dropout mask ``torch.lt(torch.rand_like(x), p)`` of :98-100."""
    g = torch.unsqueeze(x[:, mask], dim=2)                 # :95  [B,R,1,d]
    g = _normal_log_prob(g, loc, scale)                    # :96  [B,R,I,d]
    if drop is not None:
        g = torch.where(drop, torch.full_like(g, float('nan')), g)   # :100 (out of place: keeps autograd usable)
    if g.requires_grad:
        g = torch.nan_to_num(g)                            # :103 (out of place under autograd: same values)
        if pad_mask is not None:
            g = g.masked_fill(pad_mask, 0.0)               # :106-107
    else:
        # exactly the reference's in-place forms: no second [B,R,I,d] temporary (the oracle is also the timed
        # CPU baseline; the out-of-place form cost it 10-14 % at (rg_batch, rg_sum) = (2, 2))
        torch.nan_to_num_(g)                               # :103
        if pad_mask is not None:
            g.masked_fill_(pad_mask, 0.0)                  # :106-107
    return torch.sum(g, dim=-1)                            # :108


def bernoulli_leaf(x, mask, pad_mask, logits, drop=None):
    """Same with BernoulliLayer: Bernoulli(logits).log_prob = -BCEWithLogits (ratspn.py:243)."""
    g = torch.unsqueeze(x[:, mask], dim=2)
    lg, v = torch.broadcast_tensors(logits, g)
    g = -torch.nn.functional.binary_cross_entropy_with_logits(lg, v, reduction='none')
    if drop is not None:
        g = torch.where(drop, torch.full_like(g, float('nan')), g)
    g = torch.nan_to_num(g)
    if pad_mask is not None:
        g = g.masked_fill(pad_mask, 0.0)
    return torch.sum(g, dim=-1)


def product_layer(x):
    """ProductLayer.forward (ratspn.py:272-286)."""
    n_part, n_nodes = x.shape[1] // 2, x.shape[2]
    mask = torch.tensor([True, False] * n_part)            # buffer built at :269-270
    x1 = torch.unsqueeze(x[:, mask], dim=3)                # :280 boolean-mask select
    x2 = torch.unsqueeze(x[:, ~mask], dim=2)               # :281
    return (x1 + x2).view(-1, n_part, n_nodes * n_nodes)


def sum_layer(x, weight, drop=None):
    """SumLayer.forward (ratspn.py:363-378); ``drop`` (bool, shape of x) = the training-mode dropout mask of
    :371-372 (dropped inputs become -inf)."""
    if drop is not None:
        x = x.masked_fill(drop, float('-inf'))
    w = torch.log_softmax(weight, dim=2)
    return torch.logsumexp(torch.unsqueeze(x, dim=2) + w, dim=3)


def root_layer(x, weight):
    """RootLayer.forward (ratspn.py:446-458)."""
    x = torch.flatten(x, start_dim=1)
    w = torch.log_softmax(weight, dim=1)
    return torch.logsumexp(torch.unsqueeze(x, dim=1) + w, dim=2)


# --------------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------------
def ratspn_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, return_activations: bool = False,
                   drops: Optional[Dict[str, torch.Tensor]] = None):
    """RatSpn.forward (deeprob/spn/models/ratspn.py:105-122) from a state_dict.

    Layer kinds are read off the state_dict keys exactly as the reference builds them (:91-100):
    ``layers.{even}.mask`` -> Product, ``layers.{odd}.weight`` -> Sum.
    """
    drops = drops or {}   # training mode: {'leaf': mask [B,R,I,d], 'layers.<i>': mask of that sum layer's input}
    pad_mask = sd.get('base_layer.pad_mask')
    if 'base_layer.logits' in sd:
        h = bernoulli_leaf(x, sd['base_layer.mask'], pad_mask, sd['base_layer.logits'], drops.get('leaf'))
    else:
        h = gaussian_leaf(x, sd['base_layer.mask'], pad_mask, sd['base_layer.loc'], sd['base_layer.scale'],
                          drops.get('leaf'))
    acts = {'leaf': h}
    i = 0
    while 'layers.{}.mask'.format(i) in sd or 'layers.{}.weight'.format(i) in sd:
        if 'layers.{}.weight'.format(i) in sd:
            h = sum_layer(h, sd['layers.{}.weight'.format(i)], drops.get('layers.{}'.format(i)))
        else:
            h = product_layer(h)
        acts['layer{}'.format(i)] = h
        i += 1
    out = root_layer(h, sd['root_layer.weight'])
    return (out, acts) if return_activations else out


def ratspn_loss(out: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RatSpn.loss (models/ratspn.py:184-191)."""
    if out.shape[1] == 1:
        return -torch.mean(out)
    return torch.nn.functional.nll_loss(torch.log_softmax(out, dim=1), y)


def state_from_npz(npz, dtype=None) -> Dict[str, torch.Tensor]:
    sd = {}
    for k in npz.files:
        if k.startswith('sd.'):
            t = torch.from_numpy(np.asarray(npz[k]))
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            sd[k[3:]] = t
    return sd
