"""Measurement: host-side cost of one evaluation step (enqueue only), split by call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob import hip

torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
x = torch.randn(65536, 784, device='cuda')
acc = torch.zeros(2, dtype=torch.float64, device='cuda')
lib = hip.load_library()
N = 300
with torch.no_grad():
    plan = m.fused_plan(x)
    for _ in range(20):
        plan.run(acc)
    torch.cuda.synchronize()

    def timeit(name, fn, sync_every=50):
        t = 0.0
        for i in range(N):
            t0 = time.perf_counter()
            fn()
            t += time.perf_counter() - t0
            if (i + 1) % sync_every == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        print('%-40s %8.1f us per call' % (name, t / N * 1e6))

    timeit('plan.run (prep + fused kernel launch)', lambda: plan.run(acc))
    timeit('model._forward_fused', lambda: m._forward_fused(x, acc))
    timeit('plan.valid()', plan.valid)
    timeit('torch.cuda.current_stream', lambda: torch.cuda.current_stream(x.device).cuda_stream)
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    timeit('event.record', ev.record)
    timeit('torch.empty small', lambda: torch.empty(8, device='cuda'))
    timeit('dpk_ll_accumulate (1 tiny kernel)', lambda: lib.dpk_ll_accumulate(plan.out.data_ptr(), 64, acc.data_ptr(),
                                                                             torch.cuda.current_stream().cuda_stream))
    # queue depth effect: back-to-back without syncs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        plan.run(acc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('back-to-back %d steps: enqueue %.1f us/step, total %.1f us/step' % (N, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
