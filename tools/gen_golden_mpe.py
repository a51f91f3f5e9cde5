"""Golden MPE completions of the reference's RatSpn.mpe (deeprob/spn/models/ratspn.py:124-162) for two of the stored
RAT-SPN fixtures (imported by tools/gen_golden.py; needs the reference on PYTHONPATH).  The reference models are
rebuilt from the fixtures' own state_dicts, so the existing fixture files are not touched."""
import os

import numpy as np
import torch

from gen_golden import _np, _save, OUT


def _load_sd(name):
    g = np.load(os.path.join(OUT, name + '.npz'))
    return g, {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}


def gen_mpe_round6():
    """Round 6: the other depths (1: the root sits directly on the leaf partitions; 3: two sum levels, 10 classes) and an
    unpadded Bernoulli model (new, its state_dict stored with the fixture) -- for the one-launch top-down kernel."""
    from deeprob.spn.models.ratspn import GaussianRatSpn, BernoulliRatSpn
    cases = {
        'ratspn_g784_d1_r4_i8_scale': dict(in_features=784, rg_depth=1, rg_repetitions=4, rg_batch=8, rg_sum=8,
                                           optimize_scale=True, random_state=42),
        'ratspn_g784_d3_r5_i4_s4_c10': dict(in_features=784, out_classes=10, rg_depth=3, rg_repetitions=5, rg_batch=4,
                                            rg_sum=4, optimize_scale=True, random_state=7),
    }
    for name, kw in cases.items():
        g, sd = _load_sd(name)
        m = GaussianRatSpn(**kw)
        m.load_state_dict(sd)
        m.eval()
        x = torch.from_numpy(g['x_nan']).clone()
        x[1, :] = torch.from_numpy(g['x'])[1, :]
        x[0, :] = float('nan')
        arrays = {'x': _np(x), 'mpe': _np(m.mpe(x))}
        if kw.get('out_classes', 1) > 1:
            y = torch.arange(x.shape[0]) % kw['out_classes']
            arrays['y'] = _np(y)
            arrays['mpe_y'] = _np(m.mpe(x, y=y))
        _save(name + '_mpe', **arrays)
    torch.manual_seed(11)
    m = BernoulliRatSpn(32, out_classes=2, rg_depth=3, rg_repetitions=3, rg_batch=3, rg_sum=2, random_state=5)
    m.eval()
    gen = torch.Generator().manual_seed(12)
    x = (torch.rand(40, 32, generator=gen) < 0.5).float()
    x[torch.rand(40, 32, generator=gen) < 0.4] = float('nan')
    x[0, :] = float('nan')
    y = torch.arange(40) % 2
    arrays = {'sd.' + k: _np(v) for k, v in m.state_dict().items()}
    arrays.update(x=_np(x), mpe=_np(m.mpe(x)), y=_np(y), mpe_y=_np(m.mpe(x, y=y)), ll=_np(m(x)))
    _save('ratspn_bernoulli_32_d3_r3_i3_s2_c2_mpe', **arrays)


def gen_mpe():
    from deeprob.spn.models.ratspn import GaussianRatSpn
    # (padded region graphs are left out: the reference's own unpad_samples fails on them, layers/ratspn.py:84)
    cases = {
        'ratspn_g784_d2_r8_i4_s2': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=4, rg_sum=2,
                                        random_state=42),
        'ratspn_g100_d2_r11_i2_s4_c3': dict(in_features=100, out_classes=3, rg_depth=2, rg_repetitions=11, rg_batch=2,
                                            rg_sum=4, random_state=3),
    }
    for name, kw in cases.items():
        g, sd = _load_sd(name)
        m = GaussianRatSpn(**kw)
        m.load_state_dict(sd)
        m.eval()
        x = torch.from_numpy(g['x_nan']).clone()
        x[1, :] = torch.from_numpy(g['x'])[1, :]        # the fully marginalised row stays in as row 0's neighbour
        x[0, :] = float('nan')
        arrays = {'x': _np(x), 'mpe': _np(m.mpe(x))}
        if kw.get('out_classes', 1) > 1:
            y = torch.arange(x.shape[0]) % kw['out_classes']
            arrays['y'] = _np(y)
            arrays['mpe_y'] = _np(m.mpe(x, y=y))
        _save(name + '_mpe', **arrays)
