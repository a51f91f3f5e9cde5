"""Golden vectors for the DGC-SPN path (imported by tools/gen_golden.py; needs the reference on PYTHONPATH)."""
import numpy as np
import torch

from gen_golden import _np, _sd, _save


def _randomise_dgc(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if name.endswith('scale'):
                p.copy_(0.5 + 0.5 * torch.rand(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g))


def _half_nan(x, seed):
    """NaN the right half of every image, plus 20 % random pixels; one fully marginalised sample."""
    g = torch.Generator().manual_seed(seed)
    xn = x.clone()
    xn[:, :, :, x.shape[3] // 2:] = float('nan')
    xn[torch.rand(x.shape, generator=g) < 0.2] = float('nan')
    xn[1] = float('nan')
    return xn


def _dgc_fixture(name, model, x, y=None, store_state=True, layers=True, grads=True):
    model.eval()
    arrays = _sd(model) if store_state else {}
    arrays['x'] = _np(x)
    xn = _half_nan(x, 77)
    arrays['x_nan'] = _np(xn)
    with torch.no_grad():
        arrays['ll'] = _np(model(x))
        arrays['ll_nan'] = _np(model(xn))
        if layers:
            h = model.base_layer(x)
            arrays['act.leaf'] = _np(h)
            for i, layer in enumerate(model.layers):
                h = layer(h)
                arrays['act.layer{}'.format(i)] = _np(h)
    with torch.enable_grad():
        arrays['mpe'] = _np(model.mpe(xn))
    if grads:
        xg = x.clone().requires_grad_(True)
        with torch.enable_grad():
            loss = model.loss(model(xg), y)
            loss.backward()
        arrays['loss'] = _np(loss)
        arrays['grad.x'] = _np(xg.grad)
        if store_state:
            for k, p in model.named_parameters():
                if p.grad is not None:
                    arrays['grad.' + k] = _np(p.grad)
        else:
            # large models: the gradient of the leaf means and of the root weights pin the whole chain
            arrays['grad.base_layer.loc'] = _np(model.base_layer.loc.grad)
            arrays['grad.root_layer.weight'] = _np(model.root_layer.weight.grad)
        model.zero_grad()
    if y is not None:
        arrays['y'] = _np(y)
    _save(name, **arrays)


def gen_dgcspn():
    from deeprob.spn.models.dgcspn import DgcSpn
    # small models: full state, per-layer activations and every gradient
    x = torch.randn(6, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(1)
    m = DgcSpn((3, 8, 8), n_batch=4, sum_channels=4, depthwise=True, n_pooling=0)
    _dgc_fixture('dgcspn_3x8x8_dw', m, x)
    torch.manual_seed(2)
    m = DgcSpn((3, 8, 8), out_classes=3, n_batch=3, sum_channels=5, depthwise=False, n_pooling=1,
               optimize_scale=True)
    _dgc_fixture('dgcspn_3x8x8_nodw_pool1_cls', m, x, y=torch.tensor([0, 2, 1, 1, 0, 2]))
    torch.manual_seed(3)
    x12 = torch.randn(5, 1, 12, 12, generator=torch.Generator().manual_seed(4))
    m = DgcSpn((1, 12, 12), n_batch=6, sum_channels=7, depthwise=[True, False, True], n_pooling=2,
               uniform_loc=(-1.5, 1.5))
    _dgc_fixture('dgcspn_1x12x12_mixed_pool2', m, x12)
    # BASELINE config 4 architecture, B=16
    x28 = torch.randn(16, 1, 28, 28, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(5)
    m = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0)
    _randomise_dgc(m, 50)
    _dgc_fixture('dgcspn_1x28x28_dw', m, x28, store_state=False, layers=False)
    # the reference's own end-to-end cases (tests/test_dgcspn.py:89-96), seeds instead of weights
    x32 = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(6))
    for n_pooling in (0, 2):
        for dw in (False, True):
            torch.manual_seed(7)
            m = DgcSpn((3, 32, 32), n_batch=4, sum_channels=4, n_pooling=n_pooling, depthwise=dw)
            _randomise_dgc(m, 60 + n_pooling + int(dw))
            _dgc_fixture('dgcspn_3x32x32_pool{}_{}'.format(n_pooling, 'dw' if dw else 'nodw'), m, x32,
                         store_state=False, layers=False)
