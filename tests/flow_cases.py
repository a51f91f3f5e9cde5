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
    'realnvp1d_20_depth2_u48': (dict(in_features=20, n_flows=3, depth=2, units=48), 16, 17, True),
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


# training-route fixtures: name -> (constructor kwargs, train mode, RAT-SPN base kwargs or None)
TRAIN_CASES = {
    'realnvp1d_train_20': (dict(in_features=20, n_flows=3, units=32), True, None),
    'realnvp1d_evalgrad_20': (dict(in_features=20, n_flows=3, units=32), False, None),
    'realnvp1d_train_nice_15': (dict(in_features=15, n_flows=2, units=64, affine=False), True, None),
    'realnvp1d_train_nobn_logit_12': (dict(in_features=12, n_flows=2, units=32, batch_norm=False, logit=0.1), True,
                                      None),
    'realnvp1d_train_depth3_14': (dict(in_features=14, n_flows=2, depth=3, units=40), True, None),
    'realnvp1d_train_ratspn_base_16': (dict(in_features=16, n_flows=2, units=32), True,
                                       dict(in_features=16, rg_depth=1, rg_repetitions=2, rg_batch=2, rg_sum=2,
                                            random_state=42)),
}


def build_train_flow(name, g):
    """Mirror model of a training fixture with the stored state."""
    from deeprob.flows.models import RealNVP1d
    from deeprob.spn.models import GaussianRatSpn
    kw, train, base_kw = TRAIN_CASES[name]
    base = GaussianRatSpn(**base_kw) if base_kw is not None else None
    model = RealNVP1d(in_base=base, **kw)
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
    assert set(sd) == set(model.state_dict()), set(sd) ^ set(model.state_dict())
    model.load_state_dict(sd)
    return model.train(train)
