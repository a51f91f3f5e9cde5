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
