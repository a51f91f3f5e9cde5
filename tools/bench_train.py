"""Measurement: one optimisation step (forward + backward + Adam) of the three model families on the HIP path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn, DgcSpn
from deeprob.flows.models import RealNVP1d
from deeprob.torch.routines import build_optimizer

which = sys.argv[1] if len(sys.argv) > 1 else 'ratspn'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
if which == 'ratspn':
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42)
    x = torch.randn(B, 784)
elif which == 'ratspn16':
    model = GaussianRatSpn(784, rg_depth=3, rg_repetitions=8, rg_batch=16, rg_sum=16, optimize_scale=True,
                           random_state=42)
    x = torch.randn(B, 784)
elif which == 'dgcspn':
    model = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0)
    x = torch.randn(B, 1, 28, 28)
else:
    model = RealNVP1d(784)
    x = torch.randn(B, 784)
model = model.cuda().train()
x = x.cuda()
opt = (torch.optim.Adam(model.parameters(), lr=1e-3, fused=True) if os.environ.get('DPK_TORCH_ADAM') else
       build_optimizer('adam', list(model.parameters()), 1e-3, {'fused': True}))


def step():
    opt.zero_grad()
    loss = model.loss(model(x))
    loss.backward()
    opt.step()
    model.apply_constraints()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({'model': which, 'batch': B, 'ms_per_train_step': dt * 1e3, 'samples_per_s': B / dt,
                  'loss': float(loss)}))
