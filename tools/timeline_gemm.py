"""Measurement: per-wave phase timeline of the MFMA fused kernel (needs libdeeprob_hip_timeline.so:
make -C deeprob-kit_amd/csrc ../lib/libdeeprob_hip_timeline.so)."""
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
xs = [torch.randn(65536, 784, device='cuda') for _ in range(3)]
with torch.no_grad():
    for i in range(7):
        m(xs[i % 3])
torch.cuda.synchronize()
ptr, grid, NCH = open('/tmp/dpk_timeline_ptr.txt').read().split()
grid, NCH = int(grid), int(NCH)
lib = hip.load_library()
lib.dpk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
n = grid * 4 * 64 * 8
buf = np.zeros(n, dtype=np.uint64)
assert lib.dpk_debug_read(int(ptr, 16), buf.ctypes.data, n * 8) == 0
full = buf.reshape(grid, 4, 64, 8).astype(np.int64)
ng = 2 * NCH
t = full[:, :, :ng, :]
meta = full[:, :, 63, :]
clk = (meta[..., 3] - meta[..., 0]) / np.maximum(1, meta[..., 5] - meta[..., 4])
print('shader clock: %.1f s_memtime ticks per 10 ns' % clk.mean())
print('kernel entry->exit (ticks): mean %.0f max %.0f ; realtime span over grid (10ns): %d' % (
    (meta[..., 3] - meta[..., 0]).mean(), (meta[..., 3] - meta[..., 0]).max(),
    meta[..., 5].max() - meta[..., 4].min()))
print('entry -> loop start %.0f ; loop end -> exit %.0f' % ((meta[..., 1] - meta[..., 0]).mean(), (meta[..., 3] - meta[..., 2]).mean()))
print('mean per-chunk phases over all waves (ticks):')
print('  wait vmcnt     %.0f' % (t[..., 1] - t[..., 0]).mean())
print('  barrier        %.0f' % (t[..., 2] - t[..., 1]).mean())
print('  issue DMA      %.0f' % (t[..., 3] - t[..., 2]).mean())
print('  compute        %.0f' % (t[..., 4] - t[..., 3]).mean())
per = t[:, :, 1:, 0] - t[:, :, :-1, 0]
print('  chunk period   %.0f  (within tile: %.0f)' % (per.mean(), np.delete(per, NCH - 1, axis=2).mean()))
ep = t[:, :, [NCH - 1, ng - 1], :]
print('  epilogue       %.0f  (constants + nodes %.0f, exchange + root prep %.0f, classes + store %.0f)' % ((ep[..., 6] - ep[..., 5]).mean(), (ep[..., 3] - ep[..., 5]).mean(), (ep[..., 7] - ep[..., 3]).mean(), (ep[..., 6] - ep[..., 7]).mean()))
print('     constants %.0f | nodes %.0f | fallback check + exchange %.0f | root prep %.0f' % ((ep[..., 0] - ep[..., 5]).mean(), (ep[..., 3] - ep[..., 0]).mean(), (ep[..., 1] - ep[..., 3]).mean(), (ep[..., 7] - ep[..., 1]).mean()))
for b in (0, 100, 255):
    for w in (0, 3):
        r = t[b, w] - meta[b, w, 0]
        print('block', b, 'wave', w, ' '.join('[%d %d %d %d %d]' % tuple(x[:5]) for x in r[:3]), '...', ' '.join('[%d %d %d %d %d|%d %d]' % tuple(x[:7]) for x in r[NCH - 2:NCH + 1]))
