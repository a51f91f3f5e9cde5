"""Measurement: per-wave phase timeline of the slice mapping (csrc/ratspn_gemm_slice.hip; needs libdeeprob_hip_timeline.so:
make -C deeprob-kit_amd/csrc ../lib/libdeeprob_hip_timeline.so).  usage: python tools/timeline_slice.py [B] [--frozen]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['DEEPROB_HIP_LIB'] = os.path.join(ROOT, 'deeprob-kit_amd', 'lib', 'libdeeprob_hip_timeline.so')
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
FROZEN = '--frozen' in sys.argv   # the frozen-model call (no table check in the launch)
lib = hip.load_library()
lib.dpk_ratspn_slice_batch_min(0)
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
nbuf = max(3, -(-(320 << 20) // (B * 784 * 4)))
xs = [torch.randn(B, 784, device='cuda') for _ in range(nbuf)]
with torch.no_grad():
    plans = [m.fused_plan(x, static_params=True) for x in xs] if FROZEN else None
    for i in range(3 * nbuf):
        if FROZEN:
            plans[i % nbuf].run()
        else:
            m(xs[i % nbuf])
torch.cuda.synchronize()
ptr, grid = open('/tmp/dpk_timeline_slice_ptr.txt').read().split()
grid = min(int(grid), 256)
lib.dpk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
n = grid * 8 * 16 * 8
buf = np.zeros(n, dtype=np.uint64)
assert lib.dpk_debug_read(int(ptr, 16), buf.ctypes.data, n * 8) == 0
t = buf.reshape(grid, 8, 16, 8).astype(np.int64)
nb = -(-(-(-B // 32)) // grid)
t0 = t[:, :, 15, 0].min()          # (s_memtime is per-XCD comparable only approximately; good enough for phases)
print('B = %d, %d work-groups, %d blocks each; s_memtime ticks (100 MHz -> 10 ns each? no: shader clock)' % (B, grid, nb))
c = t[:, :7]
names = ['barrier A', 'partials + B + req', 'upper layers', 'next K loop']
print('slice waves, ticks between stamps, mean over work-groups / waves, per iteration (first block K loop = entry -> loop):')
print('  row  ' + ' '.join('%20s' % n for n in names) + '      total')
for r in range(min(nb, 15)):
    d = [(c[:, :, r, i + 1] - c[:, :, r, i]).mean() for i in range(4)]
    tot = (c[:, :, r, 4] - c[:, :, r, 0]).mean()
    print('  %3d  ' % r + ' '.join('%20.0f' % v for v in d) + ' %10.0f' % tot)
start = t[:, :, 15, 0]
end = t[:, :, 15, 2]
print('kernel span per wave (entry -> loop exit), mean / max ticks: %.0f / %d' % ((end - start).mean(), (end - start).max()))
print('entry -> loop (table in registers), mean ticks: %.0f' % (t[:, :7, 15, 1] - t[:, :7, 15, 0]).mean())
for w in (0, 3):
    print('work-group 0, wave %d rows (ticks since entry):' % w)
    for r in range(min(nb, 15)):
        print('   ', ' '.join('%7d' % (v - t[0, w, 15, 0]) for v in t[0, w, r, :5]))
