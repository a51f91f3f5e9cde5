"""RealNVP-1D config 5 (and DGC-SPN config 4) through the graphed evaluation window with 1 / 2 / 3 parallel chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
import bench


def main():
    from deeprob.flows.models import RealNVP1d
    from deeprob.spn.models import DgcSpn
    from tests.util import randomise_flow
    dev = torch.device('cuda', 0)
    w = torch.zeros(64, device=dev)
    for _ in range(600):
        w.add_(1.0)
    torch.manual_seed(10)
    m = RealNVP1d(784)
    randomise_flow(m, 11)
    m = m.eval().to(dev)
    xs = [torch.randn(65536, 784, device=dev) for _ in range(4)]
    print('RealNVP1d B=65536: window ms/step with 1/2/3 chains:', ' '.join('%.4f' % bench._time_window(m, xs, reps=2, chains=c) for c in (1, 2, 3)), flush=True)
    del xs
    torch.manual_seed(5)
    d = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0).eval().to(dev)
    xs = [torch.randn(8192, 1, 28, 28, device=dev) for _ in range(4)]
    print('DgcSpn B=8192: window ms/step with 1/2/3 chains:', ' '.join('%.4f' % bench._time_window(d, xs, reps=2, chains=c) for c in (1, 2, 3)), flush=True)


if __name__ == '__main__':
    main()
