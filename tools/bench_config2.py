"""Measurement: SURVEY 8d config 2 -- GaussianRatSpn(784, depth 2, reps 8) forward at B = 4096 (and 65 536) for
(rg_batch, rg_sum) in {(2,2), (8,8), (16,16)} plus the 30 %-NaN variant; model(x) under no_grad, inputs resident."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn

rows = []
for B in (4096, 65536):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, 784, generator=gen).cuda()
    xn = x.clone()
    xn[torch.rand(B, 784, generator=gen).cuda() < 0.3] = float('nan')
    for I, S in ((2, 2), (8, 8), (16, 16)):
        torch.manual_seed(0)
        m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).cuda().eval()
        for tag, inp in (('dense', x), ('30% NaN', xn)):
            with torch.no_grad():
                for _ in range(10):
                    m(inp)
                torch.cuda.synchronize()
                K = 100
                t0 = time.perf_counter()
                for _ in range(K):
                    m(inp)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / K
            rows.append({'batch': B, 'rg_batch': I, 'rg_sum': S, 'input': tag, 'ms': round(dt * 1e3, 4),
                         'M_ll_per_s': round(B / dt / 1e6, 1)})
            print(rows[-1])
print(json.dumps(rows))
