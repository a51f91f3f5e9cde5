"""Does the eager two-stream loop (default mode: in-launch table check) ever stall?  N steps in windows of 200; prints the
slowest window."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from deeprob.spn.models import GaussianRatSpn
from deeprob.parallel import ShardedLogLikelihood, workspace_replica
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).eval().to(dev)
ring = max(4, -(-(768 << 20) // (B * 784 * 4)))
xs = [torch.randn(B, 784, device=dev) for _ in range(ring)]
models = [model] + [workspace_replica(model) for _ in range(ns - 1)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(ns)]
evs = [ShardedLogLikelihood(m, static_inputs=True, static_params=False) for m in models]
worst, tot = 0.0, 0.0
with torch.no_grad():
    for w in range(N // 200 + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            with torch.cuda.stream(lanes[i % ns]):
                evs[i % ns].step(xs[i % ring])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        for e in evs: e.drain()
        if w >= 3:
            worst = max(worst, dt); tot += dt
print('B=%d streams=%d: %d steps, mean %.4f ms/step, slowest window of 200 steps %.3f ms (%.4f ms/step)' % (B, ns, N, tot / N * 1e3, worst * 1e3, worst / 200 * 1e3))
