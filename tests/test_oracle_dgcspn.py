"""Pins oracle/dgcspn_oracle.py against vectors produced by the reference (tools/gen_golden_dgcspn.py)."""
import numpy as np
import pytest
import torch

from oracle import dgcspn_oracle as dorc
from tests.dgc_cases import CASES, SMALL, build_dgc, plan_of
from tests.util import rel_err, grad_err


def _state(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize('name', sorted(CASES))
def test_forward_and_mpe_match_reference(golden, name):
    g = golden(name)
    model = build_dgc(name, g)
    sd, plan = _state(model), plan_of(name)
    x, xn = torch.from_numpy(g['x']), torch.from_numpy(g['x_nan'])
    with torch.no_grad():
        out, acts = dorc.dgcspn_forward(sd, x, plan, return_activations=True)
        assert rel_err(out.numpy(), g['ll']) <= 2e-6
        if 'act.leaf' in g.files:
            assert rel_err(acts[0].numpy(), g['act.leaf']) <= 2e-6
            for i, a in enumerate(acts[1:]):
                assert rel_err(a.numpy(), g['act.layer{}'.format(i)]) <= 2e-6, i
        assert rel_err(dorc.dgcspn_forward(sd, xn, plan).numpy(), g['ll_nan']) <= 2e-6
    mpe = dorc.dgcspn_mpe(sd, xn, plan)
    assert np.allclose(mpe.numpy(), g['mpe'], rtol=2e-5, atol=2e-6)
    # the architecture mirror built the same plan as the reference's layer list
    prods = [l for l in model.layers if hasattr(l, 'pad')]
    assert [list(p[1]) for p in plan if p[0] == 'prod'] == [list(l.pad) for l in prods]


@pytest.mark.parametrize('name', sorted(SMALL))
def test_gradients_match_reference(golden, name):
    g = golden(name)
    model = build_dgc(name, g)
    sd, plan = _state(model), plan_of(name)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'grad.' + k in g.files}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    loss = dorc.dgcspn_loss(dorc.dgcspn_forward(sd, x, plan), y)
    loss.backward()
    assert rel_err(loss.detach().numpy(), g['loss']) <= 2e-6
    # summation order differs from the conv2d backward of the reference (fp32): 1e-4 of the largest entry
    assert grad_err(x.grad.numpy(), g["grad.x"]) <= 1e-4
    assert leaves
    for k, v in leaves.items():
        assert grad_err(v.grad.numpy(), g["grad." + k]) <= 1e-4, k


def test_product_invariants_like_reference():
    """Reference tests/test_dgcspn.py:46-75: pads, output sizes, all-ones input => 4.0 in the interior."""
    ones = torch.ones(2, 3, 32, 32)
    pad, shape = dorc.product_geometry((3, 32, 32), 'full', 1, 4, True)
    assert pad == [4, 4, 4, 4] and shape == (3, 36, 36)
    assert torch.allclose(dorc.spatial_product(ones, pad, 1, 4, True)[:, :, 4:-4, 4:-4], torch.tensor(4.0))
    pad, shape = dorc.product_geometry((3, 32, 32), 'valid', 2, 1, True)
    assert pad == [0, 0, 0, 0] and shape == (3, 16, 16)
    assert torch.allclose(dorc.spatial_product(ones, pad, 2, 1, True), torch.tensor(4.0))
    pad, shape = dorc.product_geometry((3, 32, 32), 'full', 1, 8, False)
    assert pad == [8, 8, 8, 8] and shape == (81, 40, 40)
    out = dorc.spatial_product(ones, pad, 1, 8, False)
    assert tuple(out.shape) == (2, 81, 40, 40) and torch.allclose(out[:, :, 8:-8, 8:-8], torch.tensor(4.0))
