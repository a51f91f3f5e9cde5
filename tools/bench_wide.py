"""Measurement: wide RAT-SPN settings, fused single-launch kernel vs leaf (MFMA) + folded product/sum layers."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (I, S) in ((2, 2), (4, 4), (8, 8), (16, 16)):
    torch.manual_seed(0)
    m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).cuda().eval()
    for B in (4096, 65536):
        x = torch.randn(B, 784, device='cuda')
        xn = x.clone()
        xn[torch.rand_like(xn) < 0.3] = float('nan')
        with torch.no_grad():
            row = {'I': I, 'S': S, 'B': B}
            row['model_ms'] = timeit(lambda: m(x))
            row['model_nan_ms'] = timeit(lambda: m(xn))
            if m._forward_fused(x) is not None:
                row['fused_ms'] = timeit(lambda: m._forward_fused(x))
            row['folded_ms'] = timeit(lambda: m._forward_folded(x))
            row['folded_nan_ms'] = timeit(lambda: m._forward_folded(xn))
            row['leaf_ms'] = timeit(lambda: m.base_layer(x))
            a, b = m(x), m._forward_folded(x)
            row['max_diff'] = float((a - b).abs().max())
        print(json.dumps(row))
