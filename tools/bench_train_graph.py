"""Measurement: one optimisation step captured in a HIP graph (torch.cuda.CUDAGraph) against eager launches.
usage: bench_train_graph.py <ratspn|ratspn16|dgcspn|realnvp> [B]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn, DgcSpn
from deeprob.flows.models import RealNVP1d
from deeprob.torch.routines import build_optimizer

which = sys.argv[1] if len(sys.argv) > 1 else 'realnvp'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
if which == 'ratspn':
    model, x = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42), torch.randn(B, 784)
elif which == 'ratspn16':
    model, x = GaussianRatSpn(784, rg_depth=3, rg_repetitions=8, rg_batch=16, rg_sum=16, optimize_scale=True, random_state=42), torch.randn(B, 784)
elif which == 'dgcspn':
    model, x = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0), torch.randn(B, 1, 28, 28)
else:
    model, x = RealNVP1d(784), torch.randn(B, 784)
model = model.cuda().train()
static_x = x.cuda()
opt = (torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True) if os.environ.get('DPK_TORCH_ADAM') else
       build_optimizer('adam', list(model.parameters()), 1e-3, {'fused': True, 'capturable': True}))


def step():
    opt.zero_grad(set_to_none=False)
    loss = model.loss(model(static_x))
    loss.backward()
    opt.step()
    model.apply_constraints()
    return loss


def timed(fn, K=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
eager = timed(step)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
graphed = timed(g.replay)
print(json.dumps({'model': which, 'batch': B, 'eager_ms': eager * 1e3, 'graph_ms': graphed * 1e3,
                  'loss': float(static_loss)}))
