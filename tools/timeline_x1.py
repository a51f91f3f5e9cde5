"""Measurement: s_memtime stamps of work-group 0 of the x-once coupling kernel (needs lib/libdeeprob_hip_x3tl.so,
built with -DDPK_X3_TIMELINE; the stamps are printed to stderr by the launch wrapper)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['DEEPROB_HIP_LIB'] = os.path.join(ROOT, 'deeprob-kit_amd', 'lib', 'libdeeprob_hip_x3tl.so')
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.flows.models import RealNVP1d
from tests.util import randomise_flow
torch.manual_seed(10)
flow = RealNVP1d(784)
randomise_flow(flow, 11)
flow = flow.cuda().eval()
x = torch.randn(65536, 784, device='cuda')
with torch.no_grad():
    for i in range(3):
        flow(x)
    torch.cuda.synchronize()
    os.environ['DPK_X3_TIMELINE'] = sys.argv[1] if len(sys.argv) > 1 else '1'
    flow(x)
    torch.cuda.synchronize()
