"""Measurement: fused RAT-SPN forward for model variants (general scale, more channels, classes, depth 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
B = 65536
x = torch.randn(B, 784, device='cuda')
for name, kw in [('optimize_scale (8,8)', dict(rg_batch=8, rg_sum=8, optimize_scale=True)), ('unit (2,2)', dict()), ('optimize_scale (2,2)', dict(optimize_scale=True)),
                 ('unit (4,4)', dict(rg_batch=4, rg_sum=4)), ('optimize_scale (4,4)', dict(rg_batch=4, rg_sum=4, optimize_scale=True)),
                 ('unit (8,8)', dict(rg_batch=8, rg_sum=8)), ('optimize_scale (8,8)', dict(rg_batch=8, rg_sum=8, optimize_scale=True)),
                 ('unit (2,2) 10 classes', dict(out_classes=10)), ('unit (2,2) depth 3', dict(rg_depth=3)),
                 ('unit (2,2) depth 1', dict(rg_depth=1)), ('unit (2,2) 16 reps', dict(rg_repetitions=16))]:
    torch.manual_seed(0)
    base = dict(in_features=784, rg_depth=2, rg_repetitions=8, random_state=42)
    base.update(kw)
    m = GaussianRatSpn(**base).cuda().eval()
    if kw.get('optimize_scale'):
        with torch.no_grad():
            m.base_layer.scale.mul_(1.0 + 0.1 * torch.rand_like(m.base_layer.scale))
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
    print('%-28s %8.3f ms  %7.1f M LL/s' % (name, dt * 1e3, B / dt / 1e6))
