"""Golden vectors for the RealNVP-2D evaluation path (imported by tools/gen_golden.py; needs the reference on
PYTHONPATH).  The fixtures ship inputs, outputs and a per-tensor checksum of the state: every parameter / running
statistic is a pure function of (seed, tensor name) -- `randomise_flow2d`, restated in tests/util.py -- so the tests
rebuild the same state without megabytes of weights and independently of module construction order."""
import zlib

import numpy as np
import torch

from gen_golden import _np, _save


def randomise_flow2d(model, seed):
    """Default init makes s == 0 and every batch norm the identity; give every tensor signal."""
    def gen(name):
        return torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) % (2 ** 31))

    with torch.no_grad():
        for name, p in model.named_parameters():
            g = gen(name)
            if name.endswith('scale_act.weight'):
                p.copy_(0.2 + 0.3 * torch.rand(p.shape, generator=g))
            elif name.endswith('conv.weight_v'):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith('conv.weight_g'):
                p.copy_(0.15 + 0.2 * torch.rand(p.shape, generator=g))
            elif name.endswith('conv.bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif '.network.' in name and name.endswith('.weight'):      # BatchNorm2d of a conditioner
                p.copy_(0.8 + 0.4 * torch.rand(p.shape, generator=g))
            elif '.network.' in name and name.endswith('.bias'):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif 'couplings.' in name and (name.endswith('.weight') or name.endswith('.bias')):  # BatchNormLayer2d
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            g = gen(name)
            if name.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif name.endswith('running_mean'):
                b.copy_(0.3 * torch.randn(b.shape, generator=g))


def state_checksum(model):
    sd = model.state_dict()
    return np.array([float(sd[k].double().abs().sum()) for k in sorted(sd)], dtype=np.float64)


def _fixture(name, model, x):
    model.eval()
    arrays = {'x': _np(x), 'sd_check': state_checksum(model)}
    with torch.no_grad():
        arrays['ll'] = _np(model(x))
        h, _ = model.preprocess(x)
        arrays['pre'] = _np(h)
        u, ildj = model.apply_backward(h)
        arrays['u'] = _np(u)
        arrays['ildj'] = _np(ildj if torch.is_tensor(ildj) else torch.zeros(x.shape[0]))
        xr, ldj = model.apply_forward(u)
        arrays['x_rec'] = _np(xr)
        arrays['ldj'] = _np(ldj if torch.is_tensor(ldj) else torch.zeros(x.shape[0]))
        b0, d0 = model.layers[0].apply_backward(h)
        arrays['block0.u'] = _np(b0)
        arrays['block0.ildj'] = _np(d0 if torch.is_tensor(d0) else torch.zeros(x.shape[0]))
        c0, e0 = model.layers[0].in_couplings[0].apply_backward(h)
        arrays['coupling0.u'] = _np(c0)
        arrays['coupling0.ildj'] = _np(e0 if torch.is_tensor(e0) else torch.zeros(x.shape[0]))
        z0 = model.layers[0].in_couplings[0].network(model.layers[0].in_couplings[0].mask * h)
        arrays['coupling0.z'] = _np(z0)
    _save(name, **arrays)


def grad_probe(t):
    """Two checksums of a gradient tensor: sum |g| and the inner product with a fixed cosine pattern."""
    t = t.detach().double().reshape(-1)
    return [float(t.abs().sum()), float((t * torch.cos(torch.arange(t.numel(), dtype=torch.float64) * 0.37)).sum())]


def relu_margin(model, x):
    """Smallest |argument| of any nn.ReLU during one training-mode evaluation (statistics restored afterwards)."""
    state = {k: v.clone() for k, v in model.state_dict().items()}
    seen, hooks = [], []
    for m in model.modules():
        if isinstance(m, torch.nn.ReLU):
            hooks.append(m.register_forward_pre_hook(lambda mod, inp: seen.append(float(inp[0].detach().abs().min()))))
    model.train()
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    model.load_state_dict(state)
    return min(seen)


def _train_fixture(name, model, feats, seed, batch):
    """One training-mode evaluation of the reference (batch statistics, running-statistics update) and the gradients
    of loss = -mean(LL): LLs, loss, per-parameter gradient probes, the input gradient, the state checksum afterwards.
    The gradient is discontinuous where a ReLU argument crosses zero, and two fp32 evaluations of the same network
    disagree on the sign of arguments within rounding of zero: the input is the first of 120 seeded draws whose smallest
    |ReLU argument| is at least 3e-5, or the draw with the largest one (stored as `relu_margin`)."""
    best = None
    for k in range(120):
        cand = torch.randn((batch,) + feats, generator=torch.Generator().manual_seed(seed + 200 + k))
        m = relu_margin(model, cand)
        if best is None or m > best[0]:
            best = (m, cand)
        if m >= 3e-5:
            break
    margin, x = best
    model.train()
    x = x.clone().requires_grad_(True)
    ll = model(x)
    loss = model.loss(ll)
    loss.backward()
    names = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    params = dict(model.named_parameters())
    arrays = {'x': _np(x.detach()), 'll': _np(ll.detach()), 'loss': np.array(float(loss.detach())),
              'grad_names': np.array(names), 'relu_margin': np.array(margin),
              'grad_probe': np.array([grad_probe(params[n].grad if params[n].grad is not None
                                                 else torch.zeros_like(params[n])) for n in names]),
              'x_grad': _np(x.grad), 'sd_check_after': state_checksum(model)}
    model.eval()
    _save(name + '_train', **arrays)


CASES = [
    # name, in_features, kwargs, batch, seed      (the first four: the reference's own test configuration, tests/test_flows.py:86-93)
    ('realnvp2d_3x8x8_resnet', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='resnet', affine=True), 6, 21),
    ('realnvp2d_3x8x8_resnet_nice', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='resnet', affine=False), 6, 22),
    ('realnvp2d_3x8x8_densenet', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='densenet', affine=True), 5, 23),
    ('realnvp2d_3x8x8_densenet_nice', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='densenet', affine=False), 5, 24),
    ('realnvp2d_1x28x28_logit', (1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True, logit=0.05), 4, 25),
    ('realnvp2d_3x12x20_c20', (3, 12, 20), dict(n_flows=1, n_blocks=1, channels=20, network='resnet', affine=True), 3, 26),
]


TRAIN_CASES = ('realnvp2d_3x8x8_resnet', 'realnvp2d_3x8x8_resnet_nice', 'realnvp2d_3x8x8_densenet',
               'realnvp2d_3x12x20_c20')


def gen_flows2d():
    from deeprob.flows.models.realnvp import RealNVP2d
    for name, feats, kw, batch, seed in CASES:
        torch.manual_seed(seed)
        m = RealNVP2d(feats, **kw)
        randomise_flow2d(m, seed)
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.rand((batch,) + feats, generator=g) if kw.get('logit') else torch.randn((batch,) + feats, generator=g)
        _fixture(name, m, x)
        if name in TRAIN_CASES:
            _train_fixture(name, m, feats, seed, 3 if 'densenet' in name else batch)
