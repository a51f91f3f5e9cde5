"""Measurement: graph-replayed `model(x)` of the headline model with 30 % marginalised (NaN) evidence against clean evidence,
small batches (the small-batch kernel's validity GEMM).  usage: python tools/bench_nan_graph.py [B ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn

torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
for B in [int(a) for a in sys.argv[1:]] or [4096]:
    for kind in ('clean', 'nan30', 'clean'):
        xs = [torch.randn(B, 784, device='cuda') for _ in range(8)]
        if kind == 'nan30':
            for x in xs:
                x[torch.rand_like(x) < 0.3] = float('nan')
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(3):
                for x in xs:
                    m(x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(4):
                    for x in xs:
                        m(x)
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        print({'B': B, 'input': kind, 'us': round((time.perf_counter() - t0) / (50 * 32) * 1e6, 2)})
