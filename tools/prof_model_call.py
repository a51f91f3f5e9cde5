"""Host cost of model(x) under no_grad (cProfile; the GPU work is asynchronous): usage prof_model_call.py <rg_batch> [B]"""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
I = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=I, random_state=42).cuda().eval()
xs = [torch.randn(B, 784, device='cuda') for _ in range(4)]
with torch.no_grad():
    for i in range(400): m(xs[i % 4])
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(200): m(xs[i % 4])
    pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
