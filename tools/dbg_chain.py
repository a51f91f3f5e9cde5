import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob.parallel import ShardedLogLikelihood, GraphedEvaluationWindow
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).eval().to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 40
nb = min(max(4, -(-(768 << 20) // (B * 784 * 4))), L)
xs = [torch.randn(B, 784, device=dev) for _ in range(nb)]
ev = ShardedLogLikelihood(model, static_inputs=True, static_params=False)
win = GraphedEvaluationWindow(ev, [xs[i % nb] for i in range(L)], chains=chains)
for r in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = win.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('replay %d: %.3f ms (%.2f us/step) mean %.4f' % (r, dt * 1e3, dt / L * 1e6, m[-1]), flush=True)
