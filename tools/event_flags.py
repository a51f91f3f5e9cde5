"""Measurement: cost of bracketing the fused kernel with HIP events created with different flags."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob import hip

hiprt = ctypes.CDLL('libamdhip64.so')
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
xs = [torch.randn(65536, 784, device='cuda') for _ in range(2)]
acc = torch.zeros(2, dtype=torch.float64, device='cuda')
lib = hip.load_library()
N = 200
with torch.no_grad():
    plans = [m.fused_plan(x) for x in xs]
    for name, flags in [('no events', None), ('default', 0), ('ReleaseToDevice', 0x40000000),
                        ('DisableSystemFence', 0x20000000), ('torch.cuda.Event', 'torch')]:
        evs = []
        if flags == 'torch':
            for _ in range(N):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); b.record()
                evs.append((a.cuda_event, b.cuda_event, a, b))
        elif flags is not None:
            for _ in range(N):
                a, b = ctypes.c_void_p(), ctypes.c_void_p()
                assert hiprt.hipEventCreateWithFlags(ctypes.byref(a), ctypes.c_uint(flags)) == 0
                assert hiprt.hipEventCreateWithFlags(ctypes.byref(b), ctypes.c_uint(flags)) == 0
                evs.append((a.value, b.value))
        for p in plans:
            p.run(acc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            if evs:
                lib.dpk_profile_next_kernel(evs[i][0], evs[i][1])
            plans[i % 2].run(acc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        k = ''
        if evs:
            ms = ctypes.c_float()
            tot = 0.0
            for e in evs:
                rc = hiprt.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(e[0]), ctypes.c_void_p(e[1]))
                assert rc == 0, rc
                tot += ms.value
            k = 'kernel %.1f us' % (tot / N * 1e3)
        print('%-20s enqueue %.1f us/step  total %.1f us/step  %s' % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, k))
