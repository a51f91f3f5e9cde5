"""Pins the oracle (oracle/ratspn_oracle.py) against vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import ratspn_oracle as orc
from tests.util import rel_err, grad_err

FIXTURES = [
    'ratspn_g784_d2_r8_i2_s2', 'ratspn_g784_d2_r8_i8_s8', 'ratspn_g784_d2_r8_i4_s2',
    'ratspn_g784_d2_r8_i16_s16', 'ratspn_g784_d1_r4_i8_scale', 'ratspn_g784_d3_r5_i4_s4_c10',
    'ratspn_g100_d2_r11_i2_s4_c3', 'ratspn_g15_d2_r3_i3_s5_pad', 'ratspn_g15_d3_r2_i2_s2_pad',
]


@pytest.mark.parametrize('n,depth,reps,seed', [(784, 2, 8, 42), (15, 2, 2, 42), (15, 3, 4, 42), (100, 1, 3, 7)])
def test_region_graph_matches_reference(golden, n, depth, reps, seed):
    g = golden('region_{}_{}_{}_{}'.format(n, depth, reps, seed))
    layers = orc.region_graph_layers(n, depth, reps, seed)
    assert len(layers) == int(g['n_levels'])
    for lv, layer in enumerate(layers):
        flat, lens = [], []
        for item in layer:
            subs = item if (len(item) > 0 and isinstance(item[0], tuple)) else (item,)
            for sub in subs:
                flat.extend(sub)
                lens.append(len(sub))
        assert np.array_equal(np.asarray(flat), g['flat{}'.format(lv)])
        assert np.array_equal(np.asarray(lens), g['lens{}'.format(lv)])


@pytest.mark.parametrize('name', FIXTURES)
def test_forward_matches_reference(golden, name):
    g = golden(name)
    sd = orc.state_from_npz(g)
    x = torch.from_numpy(g['x'])
    out, acts = orc.ratspn_forward(sd, x, return_activations=True)
    assert rel_err(out.numpy(), g['ll']) <= 1e-6
    for k in g.files:
        if k.startswith('act.'):
            assert rel_err(acts[k[4:]].numpy(), g[k]) <= 1e-6, k
    out_nan = orc.ratspn_forward(sd, torch.from_numpy(g['x_nan']))
    assert rel_err(out_nan.numpy(), g['ll_nan']) <= 1e-6
    assert np.all(np.abs(out_nan.numpy()[1]) < 1e-5)  # fully marginalised row


@pytest.mark.parametrize('name', [f for f in FIXTURES if 'i16' not in f])
def test_gradients_match_reference(golden, name):
    g = golden(name)
    sd = orc.state_from_npz(g)
    names = [k[5:] for k in g.files if k.startswith('grad.') and k != 'grad.x']
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    loss = orc.ratspn_loss(orc.ratspn_forward(sd, x), y)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-6 * max(1.0, abs(float(g['loss'])))
    assert grad_err(x.grad.numpy(), g['grad.x']) <= 1e-5
    for k in names:
        assert grad_err(sd[k].grad.numpy(), g['grad.' + k]) <= 1e-5, k


def test_bernoulli_known_answer(golden):
    """Reference KAT tests/test_ratspn.py:46-48: the 2^15 complete assignments sum to probability 1."""
    g = golden('ratspn_bernoulli_15_d3_r4_i4_s2')
    sd = orc.state_from_npz(g)
    bits = ((np.arange(2 ** 15)[:, None] >> np.arange(14, -1, -1)[None, :]) & 1).astype(np.float32)
    ll = orc.ratspn_forward(sd, torch.from_numpy(bits))
    assert rel_err(ll.numpy(), g['ll']) <= 1e-6
    assert np.isclose(torch.sum(torch.exp(ll)).item(), 1.0)
    assert np.isclose(float(g['sum_exp_ll']), 1.0)
    ll_nan = orc.ratspn_forward(sd, torch.from_numpy(g['x_sub']))
    assert rel_err(ll_nan.numpy(), g['ll_sub_nan']) <= 1e-6


def test_layer_edge_cases(golden):
    g = golden('ratspn_layers_edge')
    h = torch.from_numpy(g['h'])
    p = orc.product_layer(h)
    s = orc.sum_layer(p, torch.from_numpy(g['sum_weight']))
    r = orc.root_layer(s, torch.from_numpy(g['root_weight']))
    assert rel_err(p.numpy(), g['prod_out']) == 0.0
    assert rel_err(s.numpy(), g['sum_out']) <= 1e-6
    assert rel_err(r.numpy(), g['root_out']) <= 1e-6
    assert not np.isnan(r.numpy()).any() and np.isneginf(r.numpy()[0]).all()


def test_fp64_agrees_with_fp32(golden):
    """The oracle in float64 bounds the fp32 reference's own rounding (<= 1.8e-7)."""
    g = golden('ratspn_g784_d2_r8_i8_s8')
    sd64 = orc.state_from_npz(g, dtype=torch.float64)
    out64 = orc.ratspn_forward(sd64, torch.from_numpy(g['x']).double())
    assert rel_err(out64.numpy(), g['ll']) <= 1e-6


MPE_DEPTH = {'ratspn_g784_d2_r8_i4_s2': 2, 'ratspn_g100_d2_r11_i2_s4_c3': 2, 'ratspn_g784_d1_r4_i8_scale': 1,
             'ratspn_g784_d3_r5_i4_s4_c10': 3}


@pytest.mark.parametrize('name', sorted(MPE_DEPTH))
def test_mpe_matches_reference(golden, name):
    """The oracle's top-down restatement (RatSpn.mpe, models/ratspn.py:124-162) against the completions the reference
    itself produced (tools/gen_golden_mpe.py): depths 1 / 2 / 3, one and several classes, given labels -- bit for bit."""
    g, m = golden(name), golden(name + '_mpe')
    sd = orc.state_from_npz(g)
    x = torch.from_numpy(m['x'])
    assert np.array_equal(orc.ratspn_mpe(sd, x, MPE_DEPTH[name]).numpy(), m['mpe'])
    if 'y' in m.files:
        assert np.array_equal(orc.ratspn_mpe(sd, x, MPE_DEPTH[name], y=torch.from_numpy(m['y'])).numpy(), m['mpe_y'])


def test_mpe_bernoulli_matches_reference(golden):
    m = golden('ratspn_bernoulli_32_d3_r3_i3_s2_c2_mpe')
    sd = orc.state_from_npz(m)
    x = torch.from_numpy(m['x'])
    assert rel_err(orc.ratspn_forward(sd, x).numpy(), m['ll']) <= 1e-6
    assert np.array_equal(orc.ratspn_mpe(sd, x, 3).numpy(), m['mpe'])
    assert np.array_equal(orc.ratspn_mpe(sd, x, 3, y=torch.from_numpy(m['y'])).numpy(), m['mpe_y'])


def test_sample_replay_is_an_ancestral_sample():
    """The replayable form of RatSpn.sample (models/ratspn.py:164-182 with the library's counter-based uniforms): the
    choices follow the model's weights (root: frequency of each repetition = its softmax mass) and the leaf draws the
    selected distributions (tight leaves around known means), on a padded region graph as well."""
    regions = orc.region_graph_layers(15, 2, 6, 3)
    leaf_regions = regions[-1]
    mask, pad = orc.leaf_masks(leaf_regions, 15, 2)
    gen = torch.Generator().manual_seed(0)
    R, I, d, S = mask.shape[0], 3, mask.shape[1], 2
    sd = {'base_layer.mask': mask, 'base_layer.loc': 5.0 * torch.randn(R, I, d, generator=gen),
          'base_layer.scale': torch.full((R, I, d), 0.01), 'layers.0.mask': torch.tensor([True, False] * (R // 2)),
          'layers.1.weight': torch.randn(R // 2, S, I * I, generator=gen), 'layers.2.mask': torch.tensor([True, False] * (R // 4)),
          'root_layer.weight': torch.randn(1, (R // 4) * S * S, generator=gen)}
    if pad is not None:
        sd['base_layer.pad_mask'] = pad
    n = 20000
    samples, rep, chan, margin = orc.ratspn_sample_replay(sd, n, 2, 15, seed=1234)
    assert tuple(samples.shape) == (n, 15) and torch.isfinite(samples).all()
    w = torch.softmax(sd['root_layer.weight'][0], 0).reshape(R // 4, S * S).sum(1).numpy()
    freq = np.bincount(rep.numpy(), minlength=R // 4) / n
    assert np.abs(freq - w).max() < 0.015
    # every variable of a sample sits within a few scales of the mean of the leaf that was selected for it
    for b in range(0, n, 997):
        r0 = int(rep[b]) * 4
        for rl in range(4):
            for j in range(d):
                if pad is not None and bool(pad[r0 + rl, 0, j]):
                    continue
                f = int(mask[r0 + rl, j])
                assert abs(float(samples[b, f]) - float(sd['base_layer.loc'][r0 + rl, int(chan[b, rl]), j])) < 0.08
