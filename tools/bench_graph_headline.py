"""Headline steps replayed from a HIP graph against eager launches (inter-kernel gap experiment)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn

B, D, ring = 65536, 784, 4
torch.manual_seed(0)
model = GaussianRatSpn(D, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).eval().cuda()
xs = [torch.randn(B, D, device='cuda') for _ in range(ring)]
plans = [model.fused_plan(x) for x in xs]
acc = torch.zeros(2, dtype=torch.float64, device='cuda')


def eager(n):
    for i in range(n):
        plans[i % ring].run(acc)


for _ in range(3):
    eager(400)
torch.cuda.synchronize()
t0 = time.perf_counter(); eager(400); torch.cuda.synchronize()
print('eager: %.4f ms/step' % ((time.perf_counter() - t0) / 400 * 1e3))

side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    eager(8)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        eager(40)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print('graph of 40 steps: %.4f ms/step' % ((time.perf_counter() - t0) / 400 * 1e3))
