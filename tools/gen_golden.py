#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE implementation.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference python3 tools/gen_golden.py [--only ratspn,flows,dgcspn,region]

Every fixture is data only: seeded inputs, the reference model's state_dict, and the outputs /
gradients the reference computes on the CPU in fp32.  Nothing of the reference's source is stored.
"""
import argparse
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _np(t):
    return t.detach().cpu().numpy().copy()


def _sd(model):
    return {'sd.' + k: _np(v) for k, v in model.state_dict().items()}


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('{:40s} {:8.1f} KiB'.format(name, os.path.getsize(path) / 1024))


def _marginalise(x, seed, p=0.3):
    g = torch.Generator().manual_seed(seed)
    x = x.clone()
    x[torch.rand(x.shape, generator=g) < p] = float('nan')
    x[1, :] = float('nan')          # fully marginalised row -> LL = 0
    x[2, 5] = float('inf')          # nan_to_num_ clamps the -inf log-density to -FLT_MAX
    return x


def gen_region():
    from deeprob.utils.region import RegionGraph
    for n, depth, reps, seed in [(784, 2, 8, 42), (15, 2, 2, 42), (15, 3, 4, 42), (100, 1, 3, 7)]:
        layers = RegionGraph(n, depth=depth, random_state=seed).make_layers(n_repetitions=reps)
        arrays = {}
        for lv, layer in enumerate(layers):
            flat, lens = [], []
            for item in layer:
                # a region is a tuple of ints, a partition a tuple of regions
                if len(item) > 0 and isinstance(item[0], tuple):
                    for sub in item:
                        flat.extend(sub)
                        lens.append(len(sub))
                else:
                    flat.extend(item)
                    lens.append(len(item))
            arrays['flat{}'.format(lv)] = np.asarray(flat, dtype=np.int64)
            arrays['lens{}'.format(lv)] = np.asarray(lens, dtype=np.int64)
        _save('region_{}_{}_{}_{}'.format(n, depth, reps, seed), n_levels=np.int64(len(layers)), **arrays)


def _ratspn_fixture(name, model, x, y=None, with_nan=True, with_grads=True, layers=True):
    model.eval()
    arrays = _sd(model)
    arrays['x'] = _np(x)
    with torch.no_grad():
        arrays['ll'] = _np(model(x))
        if layers:
            h = model.base_layer(x)
            arrays['act.leaf'] = _np(h)
            for i, layer in enumerate(model.layers):
                h = layer(h)
                arrays['act.layer{}'.format(i)] = _np(h)
        if with_nan:
            xn = _marginalise(x, seed=123)
            arrays['x_nan'] = _np(xn)
            arrays['ll_nan'] = _np(model(xn))
    if with_grads:
        xg = x.clone().requires_grad_(True)
        with torch.enable_grad():
            out = model(xg)
            loss = model.loss(out, y)
            loss.backward()
        arrays['loss'] = _np(loss)
        arrays['grad.x'] = _np(xg.grad)
        for k, p in model.named_parameters():
            if p.grad is not None:
                arrays['grad.' + k] = _np(p.grad)
        model.zero_grad()
    if y is not None:
        arrays['y'] = _np(y)
    _save(name, **arrays)


def gen_ratspn():
    from deeprob.spn.models.ratspn import GaussianRatSpn, BernoulliRatSpn
    from deeprob.spn.layers.ratspn import SumLayer, RootLayer, ProductLayer
    B = 48
    x = torch.randn(B, 784, generator=torch.Generator().manual_seed(0))
    for (i, s) in [(2, 2), (8, 8), (4, 2), (16, 16)]:
        torch.manual_seed(0)
        m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=i, rg_sum=s, random_state=42)
        _ratspn_fixture('ratspn_g784_d2_r8_i{}_s{}'.format(i, s), m, x[:32] if i == 16 else x,
                        layers=(i <= 8), with_grads=(i <= 8))
    # trainable scale, depth 1 and depth 3, several classes
    torch.manual_seed(1)
    m = GaussianRatSpn(784, rg_depth=1, rg_repetitions=4, rg_batch=8, rg_sum=8, random_state=42,
                       optimize_scale=True)
    _ratspn_fixture('ratspn_g784_d1_r4_i8_scale', m, x)
    torch.manual_seed(2)
    m = GaussianRatSpn(784, out_classes=10, rg_depth=3, rg_repetitions=5, rg_batch=4, rg_sum=4,
                       random_state=7, optimize_scale=True, uniform_loc=(-1.0, 1.0))
    y = torch.randint(10, [B], generator=torch.Generator().manual_seed(5))
    _ratspn_fixture('ratspn_g784_d3_r5_i4_s4_c10', m, x, y=y)
    # more repetitions than waves in a work-group, odd class count
    torch.manual_seed(3)
    m = GaussianRatSpn(100, out_classes=3, rg_depth=2, rg_repetitions=11, rg_batch=2, rg_sum=4, random_state=3)
    x100 = torch.randn(40, 100, generator=torch.Generator().manual_seed(1)) * 2.0 + 0.5
    y = torch.randint(3, [40], generator=torch.Generator().manual_seed(6))
    _ratspn_fixture('ratspn_g100_d2_r11_i2_s4_c3', m, x100, y=y)
    # padding path (15 variables do not split evenly) and channel counts outside the fused set
    torch.manual_seed(4)
    m = GaussianRatSpn(15, rg_depth=2, rg_repetitions=3, rg_batch=3, rg_sum=5, random_state=42,
                       optimize_scale=True)
    x15 = torch.randn(33, 15, generator=torch.Generator().manual_seed(2))
    _ratspn_fixture('ratspn_g15_d2_r3_i3_s5_pad', m, x15)
    torch.manual_seed(5)
    m = GaussianRatSpn(15, rg_depth=3, rg_repetitions=2, rg_batch=2, rg_sum=2, random_state=1)
    _ratspn_fixture('ratspn_g15_d3_r2_i2_s2_pad', m, x15)

    # Bernoulli known-answer test of the reference (tests/test_ratspn.py:46-48): all 2^15 inputs
    torch.manual_seed(42)
    np.random.seed(42)
    m = BernoulliRatSpn(15, rg_depth=3, rg_repetitions=4, rg_batch=4, rg_sum=2, random_state=42).eval()
    bits = ((np.arange(2 ** 15)[:, None] >> np.arange(14, -1, -1)[None, :]) & 1).astype(np.float32)
    xb = torch.tensor(bits)
    with torch.no_grad():
        ll = m(xb)
    arrays = _sd(m)
    arrays['ll'] = _np(ll)
    arrays['sum_exp_ll'] = np.float64(torch.sum(torch.exp(ll.double())).item())
    xs = xb[::997].clone()
    xs_nan = xs.clone()
    xs_nan[torch.rand(xs.shape, generator=torch.Generator().manual_seed(9)) < 0.4] = float('nan')
    with torch.no_grad():
        arrays['x_sub'] = _np(xs_nan)
        arrays['ll_sub_nan'] = _np(m(xs_nan))
    xg = xs.clone()
    with torch.enable_grad():
        loss = m.loss(m(xg))
        loss.backward()
    arrays['x_grad_in'] = _np(xs)
    arrays['loss'] = _np(loss)
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrays['grad.' + k] = _np(p.grad)
    _save('ratspn_bernoulli_15_d3_r4_i4_s2', **arrays)

    # layer-level edge cases: -inf inputs through Sum / Root (expect -inf, never NaN)
    torch.manual_seed(6)
    prod, sm, root = ProductLayer(8, 3), SumLayer(4, 9, 5), RootLayer(4, 5, 2)
    h = torch.randn(16, 8, 3, generator=torch.Generator().manual_seed(3)) * 30.0
    h[0] = float('-inf')
    h[1, 0] = float('-inf')
    h[2, :, 1] = float('-inf')
    with torch.no_grad():
        sm.weight[0, 1, :] = -200.0
        sm.weight[0, 1, 4] = 0.0       # one dominant weight, the others vanish (exact-path trigger)
        sm.weight[2, 0, :] *= 40.0
        p_out = prod(h)
        s_out = sm(p_out)
        r_out = root(s_out)
    _save('ratspn_layers_edge', h=_np(h), prod_out=_np(p_out), sum_weight=_np(sm.weight), sum_out=_np(s_out),
          root_weight=_np(root.weight), root_out=_np(r_out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='region,ratspn,flows,flows_train,dgcspn,mpe,mpe6,flows2d')
    args = ap.parse_args()
    import deeprob
    ref = os.path.realpath(os.path.dirname(deeprob.__file__))
    if not ref.startswith('/root/reference'):
        sys.exit('gen_golden.py must import the reference (PYTHONPATH=/root/reference), got ' + ref)
    torch.set_num_threads(8)
    todo = set(args.only.split(','))
    gens = {'region': gen_region, 'ratspn': gen_ratspn}
    try:
        from gen_golden_flows import gen_flows, gen_flows_train
        from gen_golden_dgcspn import gen_dgcspn
        from gen_golden_mpe import gen_mpe, gen_mpe_round6
        gens['mpe'] = gen_mpe
        gens['mpe6'] = gen_mpe_round6
        gens.update({'flows': gen_flows, 'flows_train': gen_flows_train, 'dgcspn': gen_dgcspn})
    except ImportError:
        pass
    try:
        from gen_golden_flows2d import gen_flows2d
        gens['flows2d'] = gen_flows2d
    except ImportError:
        pass
    for key, fn in gens.items():
        if key in todo:
            fn()


if __name__ == '__main__':
    main()
