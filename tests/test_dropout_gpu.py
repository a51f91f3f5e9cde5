"""Training-mode probabilistic dropout on the HIP path (reference: ratspn.py:98-100, :371-372; dgcspn.py:113-114,
:297-298).  The reference draws torch.rand_like masks, so parity with it can only be statistical; the kernels'
counter-based decisions are reproducible from (seed, element index), which lets the oracle replay the very same
masks: forward values and every gradient are then held to the usual bars."""
import numpy as np
import pytest
import torch

from oracle import ratspn_oracle as orc, dgcspn_oracle as dorc
from tests.util import rel_err, grad_err, dropout_mask

pytestmark = pytest.mark.gpu


class _Seeds:
    """Replaces deeprob.hip.ops.draw_seed: hands out a fixed sequence and records it."""

    def __init__(self, start=1234567):
        self.next = start
        self.used = []

    def __call__(self):
        self.next = (self.next * 6364136223846793005 + 1442695040888963407) % (2 ** 62)
        self.used.append(self.next)
        return self.next


@pytest.fixture
def seeds(monkeypatch):
    from deeprob.hip import ops
    s = _Seeds()
    monkeypatch.setattr(ops, 'draw_seed', s)
    return s


def test_hash_matches_numpy_and_rate(seeds):
    from deeprob.hip import ops
    x = torch.randn(257, 1031, device='cuda')
    for p in (0.1, 0.5, 0.93):
        seed = seeds()
        got = ops.DropoutFillFn.apply(x, p, seed)
        mask = dropout_mask(seed, x.shape, p)
        assert torch.equal(torch.isinf(got).cpu(), mask)
        assert torch.equal(got.cpu()[~mask], x.cpu()[~mask])
        n = mask.numel()
        assert abs(mask.float().mean().item() - p) < 4 * np.sqrt(p * (1 - p) / n)


@pytest.mark.parametrize('leaf', ['gaussian', 'bernoulli'])
def test_ratspn_dropout_replayed_by_the_oracle(seeds, leaf):
    from deeprob.spn.models import GaussianRatSpn, BernoulliRatSpn
    torch.manual_seed(3)
    kw = dict(in_features=15, rg_depth=2, rg_repetitions=3, rg_batch=3, rg_sum=4, in_dropout=0.25, sum_dropout=0.2,
              random_state=7)
    model = (GaussianRatSpn(optimize_scale=True, **kw) if leaf == 'gaussian' else BernoulliRatSpn(**kw))
    B = 37
    x = torch.randn(B, 15) if leaf == 'gaussian' else (torch.rand(B, 15) < 0.4).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    xg = x.cuda().requires_grad_(leaf == 'gaussian')
    out = model(xg)
    loss = model.loss(out)
    loss.backward()

    # replay: seeds were drawn in layer order (leaf, then every SumLayer bottom-up)
    base = model.base_layer
    R, I, d = base.mask.shape[0], base.out_channels, base.mask.shape[1]
    used = list(seeds.used)
    drops = {'leaf': dropout_mask(used.pop(0), (B, R, I, d), 0.25)}
    shape = (B, R, I)
    for i, layer in enumerate(model.layers):
        if hasattr(layer, 'weight'):
            drops['layers.{}'.format(i)] = dropout_mask(used.pop(0), shape, 0.2)
            shape = (B, layer.weight.shape[0], layer.weight.shape[1])
        else:
            shape = (B, shape[1] // 2, shape[2] ** 2)
    assert not used
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    xo = x.clone().requires_grad_(leaf == 'gaussian')
    want = orc.ratspn_forward({**sd, **leaves}, xo, drops=drops)
    orc.ratspn_loss(want).backward()
    assert rel_err(out.detach().cpu().numpy(), want.detach().numpy()) <= 1e-5
    checked = 0
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert grad_err(p.grad.cpu().numpy(), leaves[k].grad.numpy()) <= 1e-4, k
            checked += 1
    assert checked >= 3
    if leaf == 'gaussian':
        assert grad_err(xg.grad.cpu().numpy(), xo.grad.numpy()) <= 1e-4
    # eval mode ignores the rates and a second training pass draws new masks
    with torch.no_grad():
        model.eval()
        ev = model(x.cuda())
        assert rel_err(ev.cpu().numpy(), orc.ratspn_forward(sd, x).detach().numpy()) <= 1e-5
        model.train()
        assert not torch.equal(model(x.cuda()), out.detach())


def test_dgcspn_dropout_replayed_by_the_oracle(seeds):
    from deeprob.spn.models import DgcSpn
    torch.manual_seed(4)
    kw = dict(in_features=(2, 8, 8), n_batch=3, sum_channels=4, depthwise=True, n_pooling=1, in_dropout=0.3,
              sum_dropout=0.15, optimize_scale=True)
    model = DgcSpn(**kw)
    B = 9
    x = torch.randn(B, 2, 8, 8)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    plan = dorc.schedule((2, 8, 8), 3, 4, True, 1)
    model = model.cuda().train()
    xg = x.cuda().requires_grad_(True)
    out = model(xg)
    model.loss(out).backward()
    used = list(seeds.used)
    drops = {'leaf': dropout_mask(used.pop(0), (B, 3, 2, 8, 8), 0.3)}
    for i, layer in enumerate(model.layers):
        if not hasattr(layer, 'pad'):
            drops['layers.{}'.format(i)] = dropout_mask(used.pop(0), (B,) + tuple(layer.in_features), 0.15)
    assert not used
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and v.dim() > 0}
    xo = x.clone().requires_grad_(True)
    want = dorc.dgcspn_forward({**sd, **leaves}, xo, plan, drops=drops)
    dorc.dgcspn_loss(want).backward()
    assert rel_err(out.detach().cpu().numpy(), want.detach().numpy()) <= 1e-5
    assert grad_err(xg.grad.cpu().numpy(), xo.grad.numpy()) <= 1e-4
    for k, p in model.named_parameters():
        if p.grad is not None and k in leaves and leaves[k].grad is not None:
            assert grad_err(p.grad.cpu().numpy(), leaves[k].grad.numpy()) <= 1e-4, k
