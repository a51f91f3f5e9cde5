"""Measurement: GaussianRatSpn(784, 2, 8, 16, 16) forward at B = 65536 and 4096 (run under rocprofv3 --kernel-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16, random_state=42).cuda().eval()
for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    xs = [torch.randn(B, 784, device='cuda') for _ in range(4)]
    with torch.no_grad():
        for i in range(5): m(xs[i % 4])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): m(xs[i % 4])
        torch.cuda.synchronize()
    print('B', B, 'step_us %.2f' % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
