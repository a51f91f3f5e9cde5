"""DGC-SPN golden cases shared by the oracle (CPU) and HIP (GPU) tests."""
import numpy as np
import torch

from tests.util import randomise_dgc

# name -> (constructor kwargs, torch seed, perturbation seed or None when the state is stored)
CASES = {
    'dgcspn_3x8x8_dw': (dict(in_features=(3, 8, 8), n_batch=4, sum_channels=4, depthwise=True, n_pooling=0), 1, None),
    'dgcspn_3x8x8_nodw_pool1_cls': (dict(in_features=(3, 8, 8), out_classes=3, n_batch=3, sum_channels=5,
                                         depthwise=False, n_pooling=1, optimize_scale=True), 2, None),
    'dgcspn_1x12x12_mixed_pool2': (dict(in_features=(1, 12, 12), n_batch=6, sum_channels=7,
                                        depthwise=[True, False, True], n_pooling=2, uniform_loc=(-1.5, 1.5)), 3, None),
    'dgcspn_1x28x28_dw': (dict(in_features=(1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0),
                          5, 50),
    'dgcspn_3x32x32_pool0_nodw': (dict(in_features=(3, 32, 32), n_batch=4, sum_channels=4, n_pooling=0,
                                       depthwise=False), 7, 60),
    'dgcspn_3x32x32_pool0_dw': (dict(in_features=(3, 32, 32), n_batch=4, sum_channels=4, n_pooling=0,
                                     depthwise=True), 7, 61),
    'dgcspn_3x32x32_pool2_nodw': (dict(in_features=(3, 32, 32), n_batch=4, sum_channels=4, n_pooling=2,
                                       depthwise=False), 7, 62),
    'dgcspn_3x32x32_pool2_dw': (dict(in_features=(3, 32, 32), n_batch=4, sum_channels=4, n_pooling=2,
                                     depthwise=True), 7, 63),
}
SMALL = [k for k, v in CASES.items() if v[2] is None]


def build_dgc(name, g):
    """The mirror model with the fixture's parameters (loaded, or rebuilt from the seeds)."""
    from deeprob.spn.models import DgcSpn
    kw, seed, pseed = CASES[name]
    torch.manual_seed(seed)
    model = DgcSpn(**kw)
    if pseed is None:
        sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
        assert set(sd) == set(model.state_dict()), set(sd) ^ set(model.state_dict())
        model.load_state_dict(sd)
    else:
        randomise_dgc(model, pseed)
    return model.eval()


def plan_of(name):
    from oracle import dgcspn_oracle as dorc
    kw = CASES[name][0]
    return dorc.schedule(kw['in_features'], kw['n_batch'], kw['sum_channels'], kw['depthwise'], kw['n_pooling'])
