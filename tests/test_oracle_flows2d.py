"""Pins oracle/flows2d_oracle.py against vectors produced by the reference (tools/gen_golden_flows2d.py), and the
module structure / state_dict names of the RealNVP2d mirror against the reference's (the state checksum is taken in
sorted key order over the reference's state_dict)."""
import numpy as np
import pytest
import torch

from oracle import flows2d_oracle as orc
from tests.util import rel_err, FLOWS2D_CASES, flow2d_model, state_checksum


@pytest.mark.parametrize('case', FLOWS2D_CASES, ids=[c[0] for c in FLOWS2D_CASES])
def test_oracle_matches_reference(golden, case):
    name, feats, kw, seed = case
    g = golden(name)
    model = flow2d_model(feats, kw, seed)
    np.testing.assert_allclose(state_checksum(model), g['sd_check'], rtol=1e-6, atol=1e-6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(g['x'])
    pre = torch.from_numpy(g['pre'])
    ll = orc.log_prob(sd, x, logit_alpha=kw.get('logit'))
    assert rel_err(ll.numpy(), g['ll']) <= 2e-6
    u, ildj = orc.apply_backward(sd, pre)
    # (latents: 20+ chained couplings / batch norms; fp32 summation order of the convolutions differs between thread counts)
    assert rel_err(u.numpy(), g['u']) <= 5e-6 and rel_err(ildj.numpy(), g['ildj']) <= 2e-6
    b0, d0 = orc.block(sd, 'layers.0.', pre, False)
    assert rel_err(b0.numpy(), g['block0.u']) <= 5e-6 and rel_err(d0.numpy(), g['block0.ildj']) <= 2e-6
    c0, e0 = orc.coupling(sd, 'layers.0.in_couplings.0.', pre, False, False, False)
    assert rel_err(c0.numpy(), g['coupling0.u']) <= 2e-6 and rel_err(e0.numpy(), g['coupling0.ildj']) <= 2e-6
    xr, ldj = orc.apply_forward(sd, u)
    assert rel_err(xr.numpy(), g['x_rec']) <= 5e-6 and rel_err(ldj.numpy(), g['ldj']) <= 2e-6


def test_invertibility_like_reference():
    """Reference tests/test_flows.py:22-26, :86-93 on the restatement: forward(backward(x)) == x and ildj == -ldj."""
    from deeprob.flows.models import RealNVP2d
    torch.manual_seed(42)
    x = torch.rand(8, 3, 8, 8)
    for kw in [dict(network='resnet', affine=True), dict(network='resnet', affine=False),
               dict(network='densenet', affine=True), dict(network='densenet', affine=False)]:
        sd = {k: v.detach().clone() for k, v in RealNVP2d((3, 8, 8), n_flows=2, n_blocks=2, channels=8, **kw)
              .state_dict().items()}
        u, ildj = orc.apply_backward(sd, x)
        xr, ldj = orc.apply_forward(sd, u)
        assert torch.allclose(xr, x, atol=5e-7) and torch.allclose(ildj, -ldj, atol=5e-7)


def test_squeeze_roundtrip_like_reference():
    """Reference tests/test_flows.py:36-40."""
    x = torch.rand(4, 3, 8, 8)
    assert torch.equal(orc.unsqueeze(orc.squeeze(x)), x)


def test_constructor_errors_like_reference():
    """Reference tests/test_flows.py:101-108."""
    from deeprob.flows.models import RealNVP2d
    with pytest.raises(ValueError):
        RealNVP2d((3, 8, 8), n_flows=0)
    with pytest.raises(ValueError):
        RealNVP2d((3, 8, 8), n_blocks=0)
    with pytest.raises(ValueError):
        RealNVP2d((3, 8, 8), channels=0)
    with pytest.raises(NotImplementedError):
        RealNVP2d((3, 8, 8), network='unknown')


def test_product_path_refuses_cpu_tensors_and_training_mode():
    """No CPU / torch fallback: the 2-D flow fails loudly off the device, in training mode and when a graph is wanted."""
    from deeprob.flows.models import RealNVP2d
    from deeprob.hip import HipError
    m = RealNVP2d((3, 8, 8), n_flows=1, n_blocks=1, channels=4)
    x = torch.rand(2, 3, 8, 8)
    with pytest.raises(HipError):
        m(x)                                  # training mode
    m.eval()
    with pytest.raises(HipError):
        m(x)                                  # graph wanted (parameters require grad)
    with torch.no_grad(), pytest.raises(HipError):
        m(x)                                  # CPU tensor


def grad_probe(t):
    """tools/gen_golden_flows2d.py::grad_probe: sum |g| and the inner product with a fixed cosine pattern."""
    t = t.detach().double().reshape(-1).cpu()
    return [float(t.abs().sum()), float((t * torch.cos(torch.arange(t.numel(), dtype=torch.float64) * 0.37)).sum())]


def probe_err(got, want):
    """Both probes of every gradient tensor relative to its magnitude sum |g| (the signed probe cancels)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape and np.isfinite(got).all()
    return float(np.max(np.abs(got - want) / np.maximum(1.0, want[:, :1])))


TRAIN_CASES = [c for c in FLOWS2D_CASES if c[0] in ('realnvp2d_3x8x8_resnet', 'realnvp2d_3x8x8_resnet_nice',
                                                    'realnvp2d_3x8x8_densenet', 'realnvp2d_3x12x20_c20')]


@pytest.mark.parametrize('case', TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_oracle_training_mode_matches_reference(golden, case):
    """Batch statistics + autograd over the restatement against the reference's training-mode LLs and gradients of
    loss = -mean(LL) (per-parameter probes and the full input gradient)."""
    name, feats, kw, seed = case
    g = golden(name + '_train')
    model = flow2d_model(feats, kw, seed)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [str(n) for n in g['grad_names']]
    for n in names:
        sd[n].requires_grad_(True)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    with orc.training():
        ll = orc.log_prob(sd, x)
    loss = -torch.mean(ll)
    loss.backward()
    assert rel_err(ll.detach().numpy(), g['ll']) <= 5e-6 and abs(float(loss.detach()) - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    # (fp32 rounding through batch statistics of a few values per channel -- 12 at the 2x2 scale of the dense-net case,
    # whose gradients differ by 2e-4 between ATen thread counts)
    assert rel_err(x.grad.numpy(), g['x_grad']) <= (5e-4 if 'densenet' in name else 1e-4)
    got = np.array([grad_probe(sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in names])
    assert probe_err(got, g['grad_probe']) <= (5e-4 if 'densenet' in name else 5e-5)
