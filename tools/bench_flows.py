"""Secondary measurement (BASELINE config 5): RealNVP1d(784, 5 flows, units 128, BN, affine), eval, B=65536.
Prints LL/s and the fp32-MFMA rate of the step (mask-aware flops / time, peak 157.3 TF)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.flows.models import RealNVP1d
from tests.util import randomise_flow

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(10)
flow = RealNVP1d(784)
randomise_flow(flow, 11)
cpu_sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
flow = flow.cuda().eval()
xs = [torch.randn(B, 784, device='cuda') for _ in range(2)]
with torch.no_grad():
    for i in range(5):
        flow(xs[i % 2])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 30
    ev0.record()
    for i in range(K):
        flow(xs[i % 2])
    ev1.record()
    torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / K
flops = B * 5 * 2 * (392 * 128 + 128 * 784)       # mask-aware, SURVEY 8d
out = {'workload': 'RealNVP1d(784, 5 flows, units 128, BN, affine) forward log-likelihood', 'batch': B,
       'ms_per_step': ms, 'll_per_s': B / ms * 1e3,
       'mfma_f32_tflops_mask_aware': flops / (ms * 1e-3) / 1e12, 'mfma_f32_peak_tflops': 157.3,
       'note': 'whole step (5 couplings incl. weight re-packing, BN folds, base log-prob); the coupling '
               'kernel alone is in the rocprof stats'}
if len(sys.argv) > 2:
    from oracle import flows_oracle as forc
    n = int(sys.argv[2])
    x = torch.randn(n, 784)
    torch.set_num_threads(min(os.cpu_count(), 32))
    with torch.no_grad():
        forc.flow_log_prob(cpu_sd, x[:4096])
        t0 = time.perf_counter()
        for i in range(0, n, 4096):
            forc.flow_log_prob(cpu_sd, x[i:i + 4096])
        dt = time.perf_counter() - t0
    out['cpu_oracle_ll_per_s'] = n / dt
    out['cpu_threads'] = torch.get_num_threads()
print(json.dumps(out))
