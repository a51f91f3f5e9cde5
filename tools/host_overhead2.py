"""Measurement: where the host time of bench.py's event-bracketed step goes (per call, enqueue only)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob.parallel import ShardedLogLikelihood
from deeprob import hip

hiprt = ctypes.CDLL('libamdhip64.so')
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
xs = [torch.randn(65536, 784, device='cuda') for _ in range(2)]
ev = ShardedLogLikelihood(m, static_inputs=True)
lib = hip.load_library()
N = 200
handles = []
for _ in range(N):
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    hiprt.hipEventCreate(ctypes.byref(a)); hiprt.hipEventCreate(ctypes.byref(b))
    handles.append((a.value, b.value))
with torch.no_grad():
    for i in range(20):
        ev.step(xs[i % 2])
    ev.drain()
    torch.cuda.synchronize()
    def fresh(n):
        out = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            hiprt.hipEventCreate(ctypes.byref(a)); hiprt.hipEventCreate(ctypes.byref(b))
            out.append((a.value, b.value))
        return out

    for mode in sys.argv[1:] or ('evaluator.step + events', 'manual: hook + plan.run(pool slot)', 'evaluator.step + events',
                                 'evaluator.step no events', 'evaluator.step + events'):
        handles = fresh(N)
        if mode.endswith('[newpool]'):
            ev._pool = None
        if mode.endswith('[sleep]'):
            time.sleep(1.0)
        mode = mode.split(' [')[0]
        plan = ev._plans[(xs[0].data_ptr(), 65536)], ev._plans[(xs[1].data_ptr(), 65536)]
        acc = torch.zeros(2, dtype=torch.float64, device='cuda')
        pool = torch.zeros(256, 2, dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            if mode == 'evaluator.step + events':
                ev.step(xs[i % 2], kernel_events=handles[i])
            elif mode == 'evaluator.step no events':
                ev.step(xs[i % 2])
            elif mode == 'manual: hook + plan.run(pool slot)':
                lib.dpk_profile_next_kernel(*handles[i]); plan[i % 2].run(pool[i])
            else:
                lib.dpk_profile_next_kernel(*handles[i]); plan[i % 2].run(acc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ev.drain()
        print('%-40s enqueue %.1f us/step  total %.1f us/step' % (mode, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
