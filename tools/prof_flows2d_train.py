import os, sys
sys.path[:0] = ['/root/repo/deeprob-kit_amd', '/root/repo']
import torch
from torch.profiler import profile, ProfilerActivity
from tests.util import flow2d_model
B = 512
flow = flow2d_model((1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True), 25).cuda().train()
opt = torch.optim.Adam(flow.parameters(), lr=1e-4)
x = torch.randn(B, 1, 28, 28, device='cuda')
def step():
    opt.zero_grad(); loss = flow.loss(flow(x)); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows if e.device_type.name != 'CPU' or True)
print('kernel rows:', len(rows))
for e in rows[:22]:
    print('%-70s n=%5d  total %8.2f ms  avg %8.1f us' % (e.key[:70], e.count, e.device_time_total / 1e3 / 2, e.device_time_total / max(e.count, 1)))
