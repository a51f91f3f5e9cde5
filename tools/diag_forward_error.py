"""GPU diagnostic: error of the fused and per-layer forward against golden (fp32 reference) and the
fp64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
from tests.test_ratspn_gpu import MODELS, SEEDS
from tests.util import state_to_model
from oracle import ratspn_oracle as orc
from deeprob.spn.models import GaussianRatSpn

for name in sorted(MODELS):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    model = GaussianRatSpn(random_state=SEEDS.get(name, 42), **MODELS[name])
    state_to_model(model, g, 'cuda').eval()
    x = torch.from_numpy(g['x'])
    sd64 = orc.state_from_npz(g, dtype=torch.float64)
    ref64 = orc.ratspn_forward(sd64, x.double()).numpy()
    with torch.no_grad():
        fused = model(x.cuda()).double().cpu().numpy()
        h = model.base_layer(x.cuda())
        for layer in model.layers:
            h = layer(h)
        layered = model.root_layer(h).double().cpu().numpy()
    gold = g['ll'].astype(np.float64)
    f = lambda a: (np.max(np.abs(a - ref64)), np.max(np.abs(a - ref64) / np.abs(ref64)))
    print('{:32s} |LL|~{:8.1f}  abs/rel err vs fp64: golden {:.2e}/{:.1e}  fused {:.2e}/{:.1e}  layered {:.2e}/{:.1e}'.format(
        name, np.mean(np.abs(ref64)), *f(gold), *f(fused), *f(layered)))
