"""Measurement: the folded route's layer kernels (leaf GEMM | product+sum | product+root) of GaussianRatSpn(784, 2, 8, 16, 16)
over a batch sweep -- run under tools/kstats.sh / rocprofv3 --kernel-trace and read tools/trace_summary.py by grid size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob import hip
from deeprob.spn.models import GaussianRatSpn
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16, random_state=42).cuda().eval()
hip.trust_version_counters(True)
for B in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536]:
    x = torch.randn(B, 784, device='cuda')
    with torch.no_grad():
        for _ in range(30):
            m(x)
    torch.cuda.synchronize()
