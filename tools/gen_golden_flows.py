"""Golden vectors for the RealNVP-1D path (imported by tools/gen_golden.py; needs the
reference on PYTHONPATH)."""
import numpy as np
import torch

from gen_golden import _np, _sd, _save


def _randomise_flow(model, seed):
    """Default init makes s == 0 and BN the identity; perturb every parameter / running statistic so that
    all code paths carry signal."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('scale_act.weight'):
                p.fill_(0.3 + 0.4 * torch.rand(1, generator=g).item())
            elif '.network.' in name:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
            elif name.endswith('.weight') or name.endswith('.bias'):      # batch norm log-gain / bias
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif name.endswith('running_mean'):
                b.copy_(0.5 * torch.randn(b.shape, generator=g))


def _flow_fixture(name, model, x, store_state=True, layers=None):
    """store_state=False: the test rebuilds the parameters from the seeds (same torch RNG stream and the
    same _randomise_flow, which tests/util.py restates) instead of shipping ~6 MB of weights."""
    model.eval()
    arrays = _sd(model) if store_state else {}
    arrays['x'] = _np(x)
    with torch.no_grad():
        arrays['ll'] = _np(model(x))
        u, ildj = model.apply_backward(x)
        arrays['u'] = _np(u)
        arrays['ildj'] = _np(ildj if torch.is_tensor(ildj) else torch.zeros(x.shape[0]))
        xr, ldj = model.apply_forward(u)
        arrays['x_rec'] = _np(xr)
        arrays['ldj'] = _np(ldj if torch.is_tensor(ldj) else torch.zeros(x.shape[0]))
        h = x
        for i, layer in enumerate(model.layers):
            h, d = layer.apply_backward(h)
            if layers is None or i in layers:
                arrays['layer{}.u'.format(i)] = _np(h)
                arrays['layer{}.ildj'.format(i)] = _np(d if torch.is_tensor(d) else torch.zeros(x.shape[0]))
    _save(name, **arrays)


def gen_flows():
    from deeprob.flows.models.realnvp import RealNVP1d
    x = torch.randn(70, 784, generator=torch.Generator().manual_seed(0))
    for tag, kw in [('bn_affine', dict()), ('nobn_affine', dict(batch_norm=False)),
                    ('bn_nice', dict(affine=False)), ('u64', dict(units=64, n_flows=3))]:
        torch.manual_seed(10)
        m = RealNVP1d(784, **kw)
        _randomise_flow(m, 11)
        _flow_fixture('realnvp1d_784_' + tag, m, x, store_state=False, layers=(0, 1))
    # logit preprocessing (deterministic; dequantize is stochastic and left out), data in [0, 1]
    torch.manual_seed(12)
    m = RealNVP1d(100, logit=0.05, n_flows=4, units=96)
    _randomise_flow(m, 13)
    _flow_fixture('realnvp1d_100_logit', m, torch.rand(37, 100, generator=torch.Generator().manual_seed(1)))
    # odd number of variables: mask and inv_mask have different sizes
    torch.manual_seed(14)
    m = RealNVP1d(15, n_flows=2, units=32)
    _randomise_flow(m, 15)
    _flow_fixture('realnvp1d_15', m, torch.randn(9, 15, generator=torch.Generator().manual_seed(2)))
    # deeper conditioner and a hidden width that is not a multiple of the MFMA tile
    torch.manual_seed(16)
    m = RealNVP1d(20, n_flows=3, depth=2, units=48)
    _randomise_flow(m, 17)
    _flow_fixture('realnvp1d_20_depth2_u48', m, torch.randn(33, 20, generator=torch.Generator().manual_seed(3)))



def _flow_train_fixture(name, model, x, train):
    """One optimisation step's worth of autograd through the flow: LL, loss, d/dx, every parameter gradient
    and (train mode) the running statistics after the batch-statistics forward."""
    arrays = _sd(model)
    arrays['x'] = _np(x)
    model.train(train)
    xg = x.clone().requires_grad_(True)
    with torch.enable_grad():
        ll = model(xg)
        loss = model.loss(ll)
        loss.backward()
    arrays['ll'] = _np(ll)
    arrays['loss'] = _np(loss)
    arrays['grad.x'] = _np(xg.grad)
    for k, p in model.named_parameters():
        if p.grad is not None:
            arrays['grad.' + k] = _np(p.grad)
    for k, v in model.state_dict().items():
        if 'running_' in k:
            arrays['after.' + k] = _np(v)
    _save(name, **arrays)


def gen_flows_train():
    from deeprob.flows.models.realnvp import RealNVP1d
    from deeprob.spn.models.ratspn import GaussianRatSpn
    g = torch.Generator().manual_seed(20)
    for tag, train in [('realnvp1d_train_20', True), ('realnvp1d_evalgrad_20', False)]:
        torch.manual_seed(21)
        m = RealNVP1d(20, n_flows=3, units=32)
        _randomise_flow(m, 22)
        _flow_train_fixture(tag, m, torch.randn(24, 20, generator=torch.Generator().manual_seed(23)), train)
    torch.manual_seed(24)
    m = RealNVP1d(15, n_flows=2, units=64, affine=False)
    _randomise_flow(m, 25)
    _flow_train_fixture('realnvp1d_train_nice_15', m, torch.randn(10, 15, generator=g), True)
    torch.manual_seed(26)
    m = RealNVP1d(12, n_flows=2, units=32, batch_norm=False, logit=0.1)
    _randomise_flow(m, 27)
    _flow_train_fixture('realnvp1d_train_nobn_logit_12', m, torch.rand(70, 12, generator=g), True)
    torch.manual_seed(30)
    m = RealNVP1d(14, n_flows=2, depth=3, units=40)
    _randomise_flow(m, 31)
    _flow_train_fixture('realnvp1d_train_depth3_14', m, torch.randn(18, 14, generator=g), True)
    # RAT-SPN as the base density (examples/ratspn_nvp1d_mnist.py:35-54): needs d/dx of the SPN
    torch.manual_seed(28)
    base = GaussianRatSpn(16, rg_depth=1, rg_repetitions=2, rg_batch=2, rg_sum=2, random_state=42)
    m = RealNVP1d(16, n_flows=2, units=32, in_base=base)
    _randomise_flow(m, 29)
    _flow_train_fixture('realnvp1d_train_ratspn_base_16', m, torch.randn(12, 16, generator=g), True)
