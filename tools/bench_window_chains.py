"""Evaluation throughput of the graphed window with 1 / 2 / 3 / 4 parallel chains at small batch sizes (BASELINE config 2:
B = 4096) for the three RAT-SPN widths.   python tools/bench_window_chains.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
import bench


def main():
    from deeprob.spn.models import GaussianRatSpn
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device('cuda', 0)
    w = torch.zeros(64, device=dev)
    for _ in range(600):
        w.add_(1.0)
    for I, S in ((2, 2), (8, 8), (16, 16)):
        torch.manual_seed(0)
        m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).eval().to(dev)
        xs = [torch.randn(B, 784, device=dev) for _ in range(12)]
        res = [bench._time_window(m, xs, reps=4, chains=c) for c in (1, 2, 3, 4)]
        print('(%d,%d) B=%d: window us/step with 1/2/3/4 chains: %s' % (I, S, B, ' '.join('%.2f' % (r * 1e3) if r else 'n/a' for r in res)), flush=True)


if __name__ == '__main__':
    main()
