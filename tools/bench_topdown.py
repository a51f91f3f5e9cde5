"""RatSpn.mpe / .sample: the one-launch top-down kernel (csrc/ratspn_topdown.hip) against the layer-by-layer form
(torch index ops per layer, the reference's own structure), same model and evidence.

    python tools/bench_topdown.py [B]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from deeprob.spn.models import GaussianRatSpn
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    _w = torch.zeros(64, device='cuda')
    for _ in range(600):      # (the runtime's one-off per-queue pool growth, ~40 ms of host stall around the 200th launch, out of the way)
        _w.add_(1.0)
    torch.cuda.synchronize()
    for kw in (dict(rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2), dict(rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8),
               dict(rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16), dict(rg_depth=3, rg_repetitions=5, rg_batch=4, rg_sum=4, out_classes=10)):
        torch.manual_seed(0)
        model = GaussianRatSpn(784, random_state=42, **kw).cuda().eval()
        x = torch.randn(B, 784, device='cuda')
        x[torch.rand(B, 784, device='cuda') < 0.3] = float('nan')
        a, b = model.mpe(x), model._mpe_layerwise(x)
        same = (a == b).all(dim=1).float().mean().item()
        ms_new, ms_old = timed(lambda: model.mpe(x)), timed(lambda: model._mpe_layerwise(x))
        acts = model._upward_for_mpe(x)
        from deeprob.hip import ops
        leaf = model._leaf_params()
        logw, src = model._topdown_logw(), model._topdown_src()
        y = torch.zeros(B, dtype=torch.long, device='cuda') if model.out_classes > 1 else None
        ms_k = timed(lambda: ops.ratspn_topdown(0, leaf[0], B, model._fused_ctx, x, y, acts, logw, src, leaf[1], leaf[2]))
        ms_up = timed(lambda: model._upward_for_mpe(x))
        ms_s_new, ms_s_old = timed(lambda: model.sample(B)), timed(lambda: model._sample_layerwise(B))
        print('%-70s B=%d  mpe: %.3f ms (bottom-up %.3f + top-down kernel %.3f) vs layer-by-layer %.3f ms (%.1fx), rows identical %.4f;'
              '  sample: %.3f vs %.3f ms (%.1fx)' % (str(kw), B, ms_new, ms_up, ms_k, ms_old, ms_old / ms_new, same, ms_s_new, ms_s_old,
                                                    ms_s_old / ms_s_new), flush=True)


if __name__ == '__main__':
    main()
