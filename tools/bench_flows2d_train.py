"""Secondary measurement (training direction): one forward + backward + Adam step of
RealNVP2d((1,28,28), n_flows=1, n_blocks=2, channels=32, resnet, affine) in training mode (batch statistics).
usage: bench_flows2d_train.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from tests.util import flow2d_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
flow = flow2d_model((1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True), 25).cuda().train()
opt = torch.optim.Adam(flow.parameters(), lr=1e-4)
x = torch.randn(B, 1, 28, 28, device='cuda')


def step():
    opt.zero_grad()
    loss = flow.loss(flow(x))
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(K):
        flow(x)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / K * 1e3
print('RealNVP2d training step B={}: {:.2f} ms ({:.0f} samples/s), training-mode forward alone {:.2f} ms, loss {:.3f}'.format(
    B, ms, B / ms * 1e3, fwd, float(loss.detach())))

if len(sys.argv) > 3 and sys.argv[3] == 'graph':
    # the same step replayed from a HIP graph (deeprob/hip/graphs.py)
    from deeprob.hip.graphs import GraphedTrainStep
    flow2 = flow2d_model((1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True), 25).cuda().train()
    opt2 = torch.optim.Adam(flow2.parameters(), lr=1e-4, capturable=True)
    gstep = GraphedTrainStep(flow2, opt2)
    for _ in range(6):
        l = gstep(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        l = gstep(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print('  replayed from a HIP graph: {:.2f} ms ({:.0f} samples/s), loss {:.3f}, captured {}'.format(
        ms, B / ms * 1e3, float(l.detach()), gstep.graph is not None))
