"""Measurement: per-wave phase timeline of the small-batch fused kernel (needs libdeeprob_hip_timeline.so:
make -C deeprob-kit_amd/csrc ../lib/libdeeprob_hip_timeline.so).  usage: python tools/timeline_small.py [B]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['DEEPROB_HIP_LIB'] = os.path.join(ROOT, 'deeprob-kit_amd', 'lib', 'libdeeprob_hip_timeline.so')
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
xs = [torch.randn(B, 784, device='cuda') for _ in range(3)]
with torch.no_grad():
    for i in range(9):
        m(xs[i % 3])
torch.cuda.synchronize()
ptr, grid = open('/tmp/dpk_timeline_small_ptr.txt').read().split()
grid = min(int(grid), 256)
lib = hip.load_library()
lib.dpk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
n = grid * 8 * 16
buf = np.zeros(n, dtype=np.uint64)
assert lib.dpk_debug_read(int(ptr, 16), buf.ctypes.data, n * 8) == 0
t = buf.reshape(grid, 8, 16).astype(np.int64)
rt = t[..., 15]
rel = t[..., :12] - t[..., :1]
names = ['entry', 'loads requested', 'first K-step operands', 'K loop done', 'partials written', 'barrier 1 passed',
         'leaf sums', 'flags + constants', 'node', 'classes', 'barrier 2 passed', 'stored']
print('B = %d, %d work-groups; s_memtime ticks since the wave\'s entry, mean / min / max over all waves' % (B, grid))
for i, nme in enumerate(names):
    print('  %-24s %8.0f %8d %8d' % (nme, rel[..., i].mean(), rel[..., i].min(), rel[..., i].max()))
print('entry spread over the grid (s_memrealtime, 10 ns units): %d' % (rt.max() - rt.min()))
print('per wave of work-group 0:')
for w in range(8):
    print('   wave', w, ' '.join('%6d' % v for v in rel[0, w]))
