"""RealNVP-1D golden cases shared by the oracle (CPU) and HIP (GPU) tests."""
import numpy as np
import torch

from tests.util import randomise_flow

# name -> (constructor kwargs, torch seed, perturbation seed, state stored in the fixture?)
CASES = {
    'realnvp1d_784_bn_affine': (dict(in_features=784), 10, 11, False),
    'realnvp1d_784_nobn_affine': (dict(in_features=784, batch_norm=False), 10, 11, False),
    'realnvp1d_784_bn_nice': (dict(in_features=784, affine=False), 10, 11, False),
    'realnvp1d_784_u64': (dict(in_features=784, units=64, n_flows=3), 10, 11, False),
    'realnvp1d_100_logit': (dict(in_features=100, logit=0.05, n_flows=4, units=96), 12, 13, True),
    'realnvp1d_15': (dict(in_features=15, n_flows=2, units=32), 14, 15, True),
}


def build_flow(name, g):
    """The mirror model with the fixture's parameters (loaded, or rebuilt from the seeds)."""
    from deeprob.flows.models import RealNVP1d
    kw, seed, pseed, stored = CASES[name]
    torch.manual_seed(seed)
    model = RealNVP1d(**kw)
    if stored:
        sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
        assert set(sd) == set(model.state_dict())
        model.load_state_dict(sd)
    else:
        randomise_flow(model, pseed)
    return model.eval()
