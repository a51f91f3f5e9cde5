"""Build-container check (needs /root/reference, CPU only): the oracle that bench.py times as `cpu_baseline`
(oracle/ratspn_oracle.py, an op-for-op restatement) must cost what the imported reference costs on the same inputs --
SURVEY 8d: within +-10 % -- and return bit-identical log-likelihoods.  Exits non-zero otherwise.

usage: python tools/check_oracle_timing.py [--threads N] [--batch B]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = ['/root/reference', ROOT]
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--threads', type=int, default=min(8, os.cpu_count() or 1))
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--tol', type=float, default=0.10)
args = ap.parse_args()
torch.set_num_threads(args.threads)

from deeprob.spn.models.ratspn import GaussianRatSpn as RefRatSpn      # the reference (imported from /root/reference)
from oracle import ratspn_oracle as orc

rows, ok = [], True
for I, S in ((2, 2), (8, 8)):
    torch.manual_seed(0)
    ref = RefRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).eval()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    x = torch.randn(args.batch, 784, generator=torch.Generator().manual_seed(1))

    with torch.no_grad():
        f_ref, f_orc = (lambda: ref(x)), (lambda: orc.ratspn_forward(sd, x))
        y_ref, y_orc = f_ref(), f_orc()          # warm-up
        t_ref = t_orc = float('inf')
        for _ in range(9):                        # interleaved, fastest of 9 each: a shared host is noisy
            t0 = time.perf_counter(); f_ref(); t_ref = min(t_ref, time.perf_counter() - t0)
            t0 = time.perf_counter(); f_orc(); t_orc = min(t_orc, time.perf_counter() - t0)
    same = torch.equal(y_ref, y_orc)
    ratio = t_orc / t_ref
    good = same and abs(ratio - 1.0) <= args.tol
    ok = ok and good
    rows.append('({}, {}): reference {:.3f} s, oracle {:.3f} s, ratio {:.3f}, outputs bit-identical: {} -> {}'.format(
        I, S, t_ref, t_orc, ratio, same, 'ok' if good else 'FAIL'))
print('GaussianRatSpn(784, depth 2, reps 8), B = {}, {} threads, fastest of 9 interleaved runs'.format(args.batch, args.threads))
print('\n'.join(rows))
sys.exit(0 if ok else 1)
