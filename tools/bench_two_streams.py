"""Probe: do two evaluation streams overlap one headline launch's tail with the next one's prologue?

Every launch of the slice kernel holds all 256 compute units with one work-group each; on ONE stream launch k + 1 starts
when launch k has completely finished (tail: ~4 us of last-block upper layers with HBM idle, then a ~7.7 us prologue).
With the steps alternating between two streams (a model replica each: the in-launch table check's tickets assume that
the launches of a workspace follow one another), the next launch's work-groups are dispatched as compute units free up.

    python tools/bench_two_streams.py [B] [steps]
"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.parallel import ShardedLogLikelihood
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).eval().to(dev)
    ring = max(4, -(-(768 << 20) // (B * 784 * 4)))
    gen = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(B, 784, device=dev, generator=gen) for _ in range(ring)]
    for static in (False, True):
        for ns in (1, 2, 3):
            models = [model] + [copy.deepcopy(model) for _ in range(ns - 1)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            evs = [ShardedLogLikelihood(m, static_inputs=True, static_params=static) for m in models]

            def run(n):
                for i in range(n):
                    with torch.cuda.stream(streams[i % ns]):
                        evs[i % ns].step(xs[i % ring])

            with torch.no_grad():
                run(400)
                torch.cuda.synchronize()
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    run(steps)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / steps * 1e3
                    best = dt if best is None else min(best, dt)
                means = [e.drain() for e in evs]
            print('B=%d %s streams=%d: %.5f ms/step (%.3f G LL/s)  mean LL %.6f' % (
                B, 'frozen ' if static else 'default', ns, best, B / best / 1e6, means[0][-1]), flush=True)


if __name__ == '__main__':
    main()
