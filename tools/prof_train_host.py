import sys, os, cProfile, pstats, io
sys.path[:0] = ['/root/repo/deeprob-kit_amd', '/root/repo']
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob.torch.routines import build_optimizer
torch.manual_seed(0)
model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42).cuda().train()
x = torch.randn(512, 784, device='cuda')
opt = build_optimizer('adam', list(model.parameters()), 1e-3, {'fused': True})
def step():
    opt.zero_grad()
    loss = model.loss(model(x))
    loss.backward()
    opt.step()
    model.apply_constraints()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
