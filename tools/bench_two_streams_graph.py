"""Probe: a HIP graph whose evaluation steps run on TWO parallel chains (one fork at the head, one join at the tail of the
window; a model replica per chain) against the single-chain graph, at the strong-scaling shard sizes.  A shard launch at
one block per compute unit is prologue + one block + tail with HBM idle most of the time: two chains let the next
launch's work-groups start as compute units come free.

    python tools/bench_two_streams_graph.py
"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def graph_ms(models, xs, steps):
    ns = len(models)
    main = torch.cuda.Stream()
    lanes = [main] + [torch.cuda.Stream() for _ in range(ns - 1)]
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.no_grad(), torch.cuda.stream(main):
        for m in models:
            for x in xs:
                m(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            for s in lanes[1:]:
                s.wait_stream(main)                      # fork
            for i in range(steps):
                with torch.cuda.stream(lanes[i % ns]):
                    outs.append(models[i % ns](xs[i % len(xs)]))
            for s in lanes[1:]:
                main.wait_stream(s)                      # join
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = float('inf')
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (5 * steps))
    return best, outs


def main():
    from deeprob.spn.models import GaussianRatSpn
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).eval().to(dev)
    replicas = [model, copy.deepcopy(model), copy.deepcopy(model)]
    w = torch.zeros(64, device=dev)
    for _ in range(600):
        w.add_(1.0)
    for B in (8192, 16384, 32768, 65536):
        nb = max(2, -(-(320 << 20) // (B * 784 * 4)))
        xs = [torch.randn(B, 784, device=dev) for _ in range(nb)]
        steps = 2 * nb if 2 * nb >= 32 else 32
        steps += (-steps) % 6
        res = {}
        for ns in (1, 2, 3):
            ms, outs = graph_ms(replicas[:ns], xs, steps)
            res[ns] = ms
            if ns == 1:
                ref = outs[0].clone()
            else:
                assert torch.equal(outs[0], ref)
        print('B=%6d  graph, default mode: 1 chain %.5f ms/step, 2 chains %.5f (%.2fx), 3 chains %.5f (%.2fx)' % (
            B, res[1], res[2], res[1] / res[2], res[3], res[1] / res[3]), flush=True)
        del xs


if __name__ == '__main__':
    main()
