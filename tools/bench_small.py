"""Measurement: BASELINE config 2 region -- the fused RAT-SPN forward at small batches on both tile mappings of the
matrix-core route (small-batch kernels vs the persistent ring kernels), dense and with 30 % NaN (marginalised) inputs.
Prints step times (host clock around a loop of model(x) calls); run it under `rocprofv3 --kernel-trace` and feed the
trace to tools/trace_summary.py for the per-kernel durations by grid size."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.hip import load_library
from deeprob.spn.models import GaussianRatSpn

lib = load_library()
shapes = [(2, 2), (8, 8)] if '--wide' in sys.argv else [(2, 2)]
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 8192, 16384, 32768]
rows = []
for I, S in shapes:
    torch.manual_seed(0)
    m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).cuda().eval()
    for B in batches:
        gen = torch.Generator().manual_seed(0)
        x = torch.randn(B, 784, generator=gen).cuda()
        xn = x.clone()
        xn[torch.rand(B, 784, generator=gen).cuda() < 0.3] = float('nan')
        for mapping, thr in (('small', 1 << 40), ('ring', 0)):
            lib.dpk_ratspn_small_batch_max(thr)
            for tag, inp in (('dense', x), ('nan30', xn)):
                with torch.no_grad():
                    for _ in range(10):
                        m(inp)
                    torch.cuda.synchronize()
                    K = 50
                    t0 = time.perf_counter()
                    for _ in range(K):
                        m(inp)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / K
                rows.append({'I': I, 'S': S, 'batch': B, 'mapping': mapping, 'input': tag, 'step_us': round(dt * 1e6, 2)})
                print(rows[-1], flush=True)
lib.dpk_ratspn_small_batch_max(-1)
print(json.dumps(rows))
