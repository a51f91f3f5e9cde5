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


# --------------------------------------------------------------------------------------------
# top-down passes
# --------------------------------------------------------------------------------------------
def _inverse_masks(sd, depth: int):
    """inv_mask / inv_pad_mask as RegionGraphLayer.__init__ builds them (ratspn.py:58-66)."""
    mask = sd['base_layer.mask']
    padded = mask.shape[1] * (2 ** depth)
    inv_mask = sd['base_layer.inv_mask'] if 'base_layer.inv_mask' in sd else torch.argsort(mask.reshape(-1, padded), dim=1)
    inv_pad = None
    if sd.get('base_layer.pad_mask') is not None:
        inv_pad = torch.gather(sd['base_layer.pad_mask'].reshape(-1, padded), dim=1, index=inv_mask)
    return inv_mask, inv_pad


def _unpad_samples(sd, x, idx_group, depth: int, in_features: int):
    """RegionGraphLayer.unpad_samples (ratspn.py:68-85).  NOTE: the reference's last statement, ``samples[inv_pad_mask[..]]``,
    selects the DUMMY positions and its ``.view`` raises for every padded region graph (no reference test reaches it); the
    restatement keeps the real variables, ``~inv_pad_mask`` -- the evident intent, and what the HIP path does."""
    n = idx_group.shape[0]
    inv_mask, inv_pad = _inverse_masks(sd, depth)
    idx_rep = torch.div(idx_group[:, 0], 2 ** depth, rounding_mode='floor')
    samples = torch.gather(x, dim=1, index=inv_mask[idx_rep])
    if inv_pad is not None:
        samples = samples[~inv_pad[idx_rep]].view(n, in_features)
    return samples


def _layer_kinds(sd):
    kinds, i = [], 0
    while 'layers.{}.mask'.format(i) in sd or 'layers.{}.weight'.format(i) in sd:
        kinds.append('sum' if 'layers.{}.weight'.format(i) in sd else 'prod')
        i += 1
    return kinds


def _product_down(idx_group, idx_offset, in_nodes: int):
    """ProductLayer.sample (= .mpe; ratspn.py:306-330)."""
    first = torch.div(idx_offset, in_nodes, rounding_mode='floor')
    second = torch.remainder(idx_offset, in_nodes)
    groups = torch.flatten(torch.stack([idx_group * 2, idx_group * 2 + 1], dim=2), start_dim=1)
    offsets = torch.flatten(torch.stack([first, second], dim=2), start_dim=1)
    return groups, offsets


def leaf_mode(sd):
    if 'base_layer.logits' in sd:
        return (torch.sigmoid(sd['base_layer.logits']) >= 0.5).float()       # Bernoulli mean >= 0.5
    return sd['base_layer.loc']                                               # Normal mean (ratspn.py:130)


def ratspn_mpe(sd: Dict[str, torch.Tensor], x: torch.Tensor, depth: int, y: Optional[torch.Tensor] = None,
               return_choice: bool = False):
    """RatSpn.mpe (deeprob/spn/models/ratspn.py:124-162) with the layers' mpe methods (layers/ratspn.py:118-136, :288-304,
    :380-399, :460-474), statement for statement."""
    n = x.shape[0]
    out, acts = ratspn_forward(sd, x, return_activations=True)
    kinds = _layer_kinds(sd)
    lls = [acts['leaf']] + [acts['layer{}'.format(i)] for i in range(len(kinds) - 1)]   # input of every layer
    top = acts['layer{}'.format(len(kinds) - 1)]                                          # input of the root
    if out.shape[1] == 1:
        y = torch.zeros(n, dtype=torch.long)
    elif y is None:
        y = torch.argmax(out, dim=1)
    in_nodes = top.shape[2]
    flat = torch.flatten(top, start_dim=1)
    w = torch.log_softmax(sd['root_layer.weight'], dim=1)
    idx = torch.argmax(flat + w[y], dim=1, keepdim=True)                                  # :470
    idx_group, idx_offset = torch.div(idx, in_nodes, rounding_mode='floor'), torch.remainder(idx, in_nodes)
    for i in reversed(range(len(kinds))):
        if kinds[i] == 'prod':
            idx_group, idx_offset = _product_down(idx_group, idx_offset, lls[i].shape[2])
        else:
            rows = torch.arange(n).unsqueeze(1)
            xs = lls[i][rows, idx_group]                                                  # :395
            ws = torch.log_softmax(sd['layers.{}.weight'.format(i)][idx_group, idx_offset], dim=2)
            idx_offset = torch.argmax(xs + ws, dim=2)                                     # :398
    mode = leaf_mode(sd)
    picked = torch.flatten(mode[idx_group, idx_offset], start_dim=1)
    picked = _unpad_samples(sd, picked, idx_group, depth, x.shape[1])
    res = torch.where(torch.isnan(x), picked, x)
    return (res, idx_group, idx_offset) if return_choice else res


def hash_uniform(seed: int, ctr: np.ndarray) -> np.ndarray:
    """The library's counter-based uniform (csrc/ratspn_topdown.hip: td_uniform; the dropout hash of csrc/common.h):
    (splitmix64(seed + ctr * golden) >> 40) / 2^24, as float32."""
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + ctr.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0))


def _categorical_from_uniform(logw: torch.Tensor, u: np.ndarray):
    """Inverse CDF over softmax weights in index order (what Categorical(logits=w).sample(), ratspn.py:416 / :489, does with
    torch's own uniform).  Returns (choice, margin): margin = distance of u * total from the nearest CDF step, for tests to
    set aside the draws that fp32 rounding may legitimately flip."""
    p = torch.exp(logw.double()).numpy()
    cdf = np.cumsum(p, axis=-1)
    target = u.astype(np.float64)[..., None] * cdf[..., -1:]
    pick = np.minimum((cdf <= target).sum(axis=-1), p.shape[-1] - 1)
    margin = np.abs(cdf - target).min(axis=-1)
    return pick, margin


def ratspn_sample_replay(sd: Dict[str, torch.Tensor], n: int, depth: int, in_features: int, seed: int,
                         y: Optional[torch.Tensor] = None):
    """RatSpn.sample (deeprob/spn/models/ratspn.py:164-182; layers :138-157, :306-330, :401-417, :476-490) with every random
    draw replaced by the counter-based uniform the HIP kernel uses (counter layout in csrc/ratspn_topdown.hip): the same
    ancestral pass, replayable.  Returns (samples [n, D], repetition [n], leaf channels [n, 2^depth], margin [n] = the
    smallest categorical margin of the sample's draws)."""
    kinds = _layer_kinds(sd)
    G0 = 2 ** depth
    K = G0 + 2 * in_features
    b = np.arange(n, dtype=np.uint64) * np.uint64(K)
    y = torch.zeros(n, dtype=torch.long) if y is None else y
    w = torch.log_softmax(sd['root_layer.weight'], dim=1)
    in_nodes = w.shape[1] // (sd['base_layer.mask'].shape[0] // G0)            # inputs per partition of the root
    idx, margin = _categorical_from_uniform(w[y], hash_uniform(seed, b + np.uint64(1)))
    idx = torch.from_numpy(idx).unsqueeze(1)
    idx_group, idx_offset = torch.div(idx, in_nodes, rounding_mode='floor'), torch.remainder(idx, in_nodes)
    rep = idx_group[:, 0].clone()
    nodes = int(round(math.sqrt(in_nodes)))
    for i in reversed(range(len(kinds))):
        if kinds[i] == 'prod':
            idx_group, idx_offset = _product_down(idx_group, idx_offset, nodes)
        else:
            G = idx_group.shape[1]
            ws = torch.log_softmax(sd['layers.{}.weight'.format(i)][idx_group, idx_offset], dim=2)
            local = (idx_group - rep.unsqueeze(1) * G).numpy().astype(np.uint64)
            u = hash_uniform(seed, b[:, None] + np.uint64(G) + local)
            pick, m = _categorical_from_uniform(ws, u)
            margin = np.minimum(margin, m.min(axis=1))
            idx_offset = torch.from_numpy(pick)
            nodes = int(round(math.sqrt(ws.shape[2])))
    # leaves: every selected distribution draws from its own two uniforms, addressed by the VARIABLE it holds
    mask = sd['base_layer.mask']                                                # [R, d]
    d = mask.shape[1]
    f = mask[idx_group].numpy().astype(np.uint64)                               # [n, G0, d]
    c = b[:, None, None] + np.uint64(G0) + np.uint64(2) * f
    u1, u2 = hash_uniform(seed, c), hash_uniform(seed, c + np.uint64(1))
    if 'base_layer.logits' in sd:
        prob = torch.sigmoid(sd['base_layer.logits'][idx_group, idx_offset]).numpy()
        draws = (u1 < prob).astype(np.float32)
    else:
        z = np.sqrt(-2.0 * np.log(1.0 - u1.astype(np.float64))) * np.cos(2.0 * np.pi * u2.astype(np.float64))
        loc = sd['base_layer.loc'][idx_group, idx_offset].double().numpy()
        scale = sd['base_layer.scale'][idx_group, idx_offset].double().numpy()
        draws = (loc + scale * z).astype(np.float32)
    flat = torch.from_numpy(draws).reshape(n, G0 * d)
    samples = _unpad_samples(sd, flat, idx_group, depth, in_features)
    return samples, rep, idx_offset, margin
