"""Pins oracle/flows_oracle.py against vectors produced by the reference (tools/gen_golden_flows.py)."""
import numpy as np
import pytest
import torch

from oracle import flows_oracle as forc
from tests.flow_cases import CASES, build_flow
from tests.util import rel_err


@pytest.mark.parametrize('name', sorted(CASES))
def test_flow_matches_reference(golden, name):
    g = golden(name)
    model = build_flow(name, g)   # also proves: same seeds => same parameters as the reference built
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(g['x'])
    alpha = CASES[name][0].get('logit')
    ll = forc.flow_log_prob(sd, x, logit_alpha=alpha)
    assert rel_err(ll.numpy(), g['ll']) <= 2e-6
    steps = []
    u, ildj = forc.flow_apply_backward(sd, x, collect=steps)
    assert rel_err(u.numpy(), g['u']) <= 2e-6 and rel_err(ildj.numpy(), g['ildj']) <= 2e-6
    for i, (h, d) in enumerate(steps):
        if 'layer{}.u'.format(i) in g.files:
            assert rel_err(h.numpy(), g['layer{}.u'.format(i)]) <= 2e-6
            assert rel_err(d.numpy(), g['layer{}.ildj'.format(i)]) <= 2e-6
    xr, ldj = forc.flow_apply_forward(sd, u)
    assert rel_err(xr.numpy(), g['x_rec']) <= 2e-6 and rel_err(ldj.numpy(), g['ldj']) <= 2e-6


def test_invertibility_like_reference():
    """Reference tests/test_flows.py:22-26 on the oracle: forward(backward(x)) == x, ildj == -ldj, atol 5e-7
    (default init: ScaledTanh weight 0, identity batch norm, exactly as the reference test builds it)."""
    from deeprob.flows.models import RealNVP1d
    torch.manual_seed(42)
    x = torch.rand(32, 192)
    for kw in [dict(batch_norm=True, affine=True), dict(batch_norm=False, affine=True),
               dict(batch_norm=True, affine=False)]:
        sd = {k: v.detach().clone() for k, v in RealNVP1d(192, **kw).state_dict().items()}
        u, ildj = forc.flow_apply_backward(sd, x)
        xr, ldj = forc.flow_apply_forward(sd, u)
        assert torch.allclose(ildj, -ldj, atol=5e-7) and torch.allclose(xr, x, atol=5e-7)


@pytest.mark.parametrize('name', sorted(__import__('tests.flow_cases', fromlist=['TRAIN_CASES']).TRAIN_CASES))
def test_training_route_matches_reference(golden, name):
    """Autograd through the oracle == the reference's autograd (LL, loss, d/dx, every parameter gradient) and the
    running statistics after a train-mode forward."""
    from tests.flow_cases import TRAIN_CASES
    from tests.util import grad_err
    from oracle import ratspn_oracle as orc
    g = golden(name)
    kw, train, base_kw = TRAIN_CASES[name]
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])).clone() for k in g.files if k.startswith('sd.')}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if 'grad.' + k in g.files}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    base = None
    if base_kw is not None:
        bsd = {k[len('in_base.'):]: v for k, v in sd.items() if k.startswith('in_base.')}
        base = lambda u: orc.ratspn_forward(bsd, u)   # noqa: E731
    running = {}
    ll = forc.flow_log_prob(sd, x, logit_alpha=kw.get('logit'), train=train and kw.get('batch_norm', True),
                            running=running, base=base)
    loss = -torch.mean(ll)
    loss.backward()
    assert rel_err(ll.detach().numpy(), g['ll']) <= 2e-6 and rel_err(loss.detach().numpy(), g['loss']) <= 2e-6
    assert grad_err(x.grad.numpy(), g['grad.x']) <= 1e-5
    assert leaves
    for k, v in leaves.items():
        assert grad_err(v.grad.numpy(), g['grad.' + k]) <= 1e-5, k
    for k in g.files:
        if k.startswith('after.') and train:
            assert rel_err(running[k[6:]].numpy(), g[k]) <= 2e-6, k
