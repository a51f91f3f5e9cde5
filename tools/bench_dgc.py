"""Secondary measurement (BASELINE config 4): DgcSpn((1,28,28), n_batch 8, sum_channels 8, depthwise, n_pooling 0),
eval, B=8192.  Prints LL/s and the algorithmic HBM rate (588 KB/sample) against the 8 TB/s peak."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.spn.models import DgcSpn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(5)
model = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0)
cpu_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.cuda().eval()
xs = [torch.randn(B, 1, 28, 28, device='cuda') for _ in range(2)]
with torch.no_grad():
    for i in range(3):
        model(xs[i % 2])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 10
    ev0.record()
    for i in range(K):
        model(xs[i % 2])
    ev1.record()
    torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / K
# algorithmic bytes / sample with the product folded into the sum (each level reads its input map and writes its
# output map once): leaf 28^2*(1+8), levels (in HW + out HW)*8 channels, root 8*32^2 in
sizes = [28, 29, 31, 35, 43, 59]
alg = 4 * (28 * 28 * (1 + 8) + sum(8 * (a * a + b * b) for a, b in zip(sizes[:-1], sizes[1:])) + 8 * 59 * 59)
out = {'workload': 'DgcSpn((1,28,28), n_batch 8, sum_channels 8, depthwise, n_pooling 0) forward', 'batch': B,
       'ms_per_step': ms, 'll_per_s': B / ms * 1e3, 'alg_bytes_per_sample': alg,
       'hbm_GBps_algorithmic': alg * B / (ms * 1e-3) / 1e9, 'hbm_peak_GBps': 8000}
if len(sys.argv) > 2:
    from oracle import dgcspn_oracle as dorc
    n = int(sys.argv[2])
    plan = dorc.schedule((1, 28, 28), 8, 8, True, 0)
    x = torch.randn(n, 1, 28, 28)
    torch.set_num_threads(min(os.cpu_count(), 32))
    with torch.no_grad():
        dorc.dgcspn_forward(cpu_sd, x[:128], plan)
        t0 = time.perf_counter()
        for i in range(0, n, 256):
            dorc.dgcspn_forward(cpu_sd, x[i:i + 256], plan)
        dt = time.perf_counter() - t0
    out['cpu_oracle_ll_per_s'] = n / dt
    out['cpu_threads'] = torch.get_num_threads()
print(json.dumps(out))
