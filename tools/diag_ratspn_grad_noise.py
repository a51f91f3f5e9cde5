"""Where does the RAT-SPN backward's distance from the fp64 oracle come from?  (VERDICT r05 weak #2.)

For each golden model: distance of (a) the golden = the reference's own fp32 result, (b) the HIP path, (c) the HIP path
fed with correctly rounded leaf activations, from the fp64 oracle -- for the leaf activations, grad.x and the parameter
gradients.  Measurement script: the oracle is the checker column only.

    python tools/diag_ratspn_grad_noise.py [model ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import ratspn_oracle as orc  # noqa: E402
from tests.util import grad_err, state_to_model  # noqa: E402
from tests.test_ratspn_gpu import MODELS, SEEDS  # noqa: E402


def fp64_all(g):
    sd = orc.state_from_npz(g, dtype=torch.float64)
    names = [k[5:] for k in g.files if k.startswith('grad.') and k != 'grad.x']
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    x = torch.from_numpy(g['x']).double().requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    with torch.enable_grad():
        out, acts = orc.ratspn_forward(sd, x, return_activations=True)
        acts['leaf'].retain_grad()
        orc.ratspn_loss(out, y).backward()
    gr = {k: sd[k].grad.numpy() for k in names}
    gr['x'] = x.grad.numpy()
    return gr, acts, out


def main():
    from deeprob.spn.models import GaussianRatSpn
    names = sys.argv[1:] or ['ratspn_g784_d2_r8_i2_s2', 'ratspn_g784_d2_r8_i8_s8', 'ratspn_g784_d1_r4_i8_scale',
                             'ratspn_g784_d3_r5_i4_s4_c10', 'ratspn_g100_d2_r11_i2_s4_c3']
    for name in names:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        model = GaussianRatSpn(random_state=SEEDS.get(name, 42), **MODELS[name])
        state_to_model(model, g, 'cuda').eval()
        ref, acts64, out64 = fp64_all(g)
        print('== %s' % name)
        x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
        y = torch.from_numpy(g['y']).cuda() if 'y' in g.files else None
        with torch.enable_grad():
            leaf = model.base_layer(x)
        leaf64 = acts64['leaf'].detach().numpy()
        if leaf64 is not None:
            print('  leaf activations, max |err| vs fp64:  HIP %.3e   golden %s   (|leaf| up to %.1f)' % (
                np.abs(leaf.detach().cpu().numpy() - leaf64).max(),
                ('%.3e' % np.abs(g['act.leaf'] - leaf64).max()) if 'act.leaf' in g.files else 'n/a', np.abs(leaf64).max()))
        with torch.enable_grad():
            model.zero_grad()
            loss = model.loss(model(x), y)
            loss.backward()
        print('  loss: HIP %.9g  golden %.9g  fp64 %.9g' % (loss.item(), float(g['loss']), float(orc.ratspn_loss(out64.detach(), torch.from_numpy(g['y']) if 'y' in g.files else None))))
        rows = [('x', x.grad.cpu().numpy())] + [(k, p.grad.cpu().numpy()) for k, p in model.named_parameters() if 'grad.' + k in g.files]
        for k, got in rows:
            print('  grad.%-22s HIP vs fp64 %.3e   golden vs fp64 %.3e   HIP vs golden %.3e' % (
                k, grad_err(got, ref[k]), grad_err(g['grad.' + k], ref[k]), grad_err(got, g['grad.' + k])))
        # (c) the upper layers fed with the correctly rounded leaf activations
        if leaf64 is not None:
            lf = torch.from_numpy(leaf64).float().cuda().requires_grad_(True)
            with torch.enable_grad():
                h = lf
                for layer in model.layers:
                    h = layer(h)
                loss2 = model.loss(model.root_layer(h), y)
                loss2.backward()
            # d loss / d leaf against the fp64 one
            a0 = acts64['leaf']
            gl64 = a0.grad.numpy() if a0.grad is not None else None
            if gl64 is not None:
                print('  d loss / d leaf with exact leaf inputs: HIP vs fp64 %.3e' % grad_err(lf.grad.cpu().numpy(), gl64))
                # and with the HIP leaf
                lh = leaf.detach().clone().requires_grad_(True)
                with torch.enable_grad():
                    h = lh
                    for layer in model.layers:
                        h = layer(h)
                    model.loss(model.root_layer(h), y).backward()
                print('  d loss / d leaf with the HIP leaf inputs:  HIP vs fp64 %.3e' % grad_err(lh.grad.cpu().numpy(), gl64))
                if 'act.leaf' in g.files:
                    lg = torch.from_numpy(g['act.leaf']).cuda().requires_grad_(True)
                    with torch.enable_grad():
                        h = lg
                        for layer in model.layers:
                            h = layer(h)
                        model.loss(model.root_layer(h), y).backward()
                    print('  d loss / d leaf with the golden leaf inputs: HIP vs fp64 %.3e' % grad_err(lg.grad.cpu().numpy(), gl64))


if __name__ == '__main__':
    main()
