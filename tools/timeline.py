"""Measurement: per-wave phase timeline of the fused kernel (needs libdeeprob_hip_timeline.so)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['DEEPROB_HIP_LIB'] = os.path.join(ROOT, 'deeprob-kit_amd', 'lib', 'libdeeprob_hip_timeline.so')
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob import hip
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
x = torch.randn(65536, 784, device='cuda')
with torch.no_grad():
    for _ in range(5):
        m(x)
torch.cuda.synchronize()
ptr, grid, NC = open('/tmp/dpk_timeline_ptr.txt').read().split()
grid, NC = int(grid), int(NC)
lib = hip.load_library()
lib.dpk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
n = grid * 8 * (NC + 2) * 6
buf = np.zeros(n, dtype=np.uint64)
assert lib.dpk_debug_read(int(ptr, 16), buf.ctypes.data, n * 8) == 0
full = buf.reshape(grid, 8, NC + 2, 6)[:, :, :NC, :].astype(np.int64)
t = full[..., :5]
rt = full[..., 5]
print('shader clock during the kernel: s_memtime ticks per s_memrealtime tick (100 MHz):',
      float((t[0, 0, -1, 4] - t[0, 0, 0, 4]) / max(1, rt[0, 0, -1] - rt[0, 0, 0])), '-> x100 MHz')
t0 = t[:, :, 0, 0].min()
print('s_memtime ticks (100 MHz => 10 ns each?) relative to the first stamp')
for b in (0, 1, 255, 256, 511):
    print('block', b)
    for w in (0, 7):
        rows = t[b, w] - t0
        print('  wave', w, ' '.join('[%d %d %d %d %d]' % tuple(r) for r in rows[:4]), '...', '[%d %d %d %d %d]' % tuple(rows[-1]))
d = t
print('mean phase durations over all waves/chunks (ticks):')
print('  wait B1     ', (d[..., 1] - d[..., 0]).mean())
print('  staging     ', (d[..., 2] - d[..., 1]).mean())
print('  wait B2     ', (d[..., 3] - d[..., 2]).mean())
print('  compute     ', (d[..., 4] - d[..., 3]).mean())
print('  chunk total ', (d[:, :, 1:, 0] - d[:, :, :-1, 0]).mean())
print('  kernel span ', d[..., 4].max() - d[..., 0].min())

ext = buf.reshape(grid, 8, NC + 2, 6)[:, :, NC, :4].astype(np.int64)   # [entry, exit, rt_entry, rt_exit]
print('per-wave (ticks): entry -> first chunk stamp', (t[:, :, 0, 0] - ext[:, :, 0]).mean(),
      ' last compute stamp -> exit', (ext[:, :, 1] - t[:, :, -1, 4]).mean(),
      ' entry -> exit', (ext[:, :, 1] - ext[:, :, 0]).mean(), 'max', (ext[:, :, 1] - ext[:, :, 0]).max())
rt0, rt1 = ext[:, :, 2], ext[:, :, 3]
print('realtime (10 ns ticks): first entry -> last exit over the grid', rt1.max() - rt0.min(),
      ' entry spread', rt0.max() - rt0.min(), ' per-block span mean', (rt1.max(axis=1) - rt0.min(axis=1)).mean())
