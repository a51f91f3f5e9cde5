"""Experiment: does a HIP-graph capture of model(x) slow later eager calls of the same model? (bench.py (8,8) entry)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
import bench
from deeprob.spn.models import GaussianRatSpn
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42).cuda().eval()
xl = [torch.randn(65536, 784, device='cuda') for _ in range(4)]
xs = [torch.randn(4096, 784, device='cuda') for _ in range(8)]


def eager(xx, n=20):
    with torch.no_grad():
        for i in range(5): m(xx[i % len(xx)])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): m(xx[i % len(xx)])
        e1.record()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


print('eager 65536 before', eager(xl))
print('eager 4096 before', eager(xs, 50))
print('graph 4096', bench._time_eval_graph(m, xs))
print('eager 65536 after graph(4096)', eager(xl))
print('graph 65536', bench._time_eval_graph(m, xl))
print('eager 65536 after graph(65536)', eager(xl))
print('eager 4096 after', eager(xs, 50))
