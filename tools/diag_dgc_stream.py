"""Streaming DGC-SPN level kernels against the batch-independent ones: run once with DPK_DGC_STREAM_MIN_B=1000000000
(old route) and once with the default, each saving its log-likelihoods; `cmp` prints the largest relative difference.
usage: diag_dgc_stream.py run <out.pt> [B]   |   diag_dgc_stream.py cmp <a.pt> <b.pt>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch

if sys.argv[1] == 'run':
    from deeprob.spn.models import DgcSpn
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    out = {}
    for name, kw in [('density', dict(n_batch=8, sum_channels=8, depthwise=True, n_pooling=0)),
                     ('classes', dict(n_batch=8, sum_channels=8, depthwise=True, n_pooling=0, out_classes=10))]:
        torch.manual_seed(5)
        m = DgcSpn((1, 28, 28), **kw).cuda().eval()
        x = torch.randn(B, 1, 28, 28, generator=torch.Generator().manual_seed(1)).cuda()
        x[3] = 40.0      # far tails: the exact log-domain pass
        with torch.no_grad():
            y = m(x)
        torch.cuda.synchronize()
        out[name] = y.cpu()
        print(name, y.shape, float(y.mean()), bool(torch.isfinite(y).all()))
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = ((a[k] - b[k]).abs() / a[k].abs().clamp_min(1.0)).max()
        print(k, 'max rel diff', float(d), 'at', int(((a[k] - b[k]).abs() / a[k].abs().clamp_min(1.0)).argmax()))
