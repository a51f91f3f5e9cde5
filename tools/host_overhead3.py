"""Measurement: per-call enqueue time of the first evaluator.step loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob.parallel import ShardedLogLikelihood
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
xs = [torch.randn(65536, 784, device='cuda') for _ in range(2)]
ev = ShardedLogLikelihood(m, static_inputs=True)
with torch.no_grad():
    for i in range(20):
        ev.step(xs[i % 2])
    ev.drain()
    torch.cuda.synchronize()
    for rep in range(2):
        ts = []
        for i in range(200):
            t0 = time.perf_counter()
            ev.step(xs[i % 2])
            ts.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
        ev.drain()
        print('loop %d: sum %.0f us; first 40: %s' % (rep, sum(ts), ' '.join('%.0f' % t for t in ts[:40])))
        print('   steps > 100us:', [(i, int(t)) for i, t in enumerate(ts) if t > 100][:30])
