"""Measurement: BASELINE config 2 (GaussianRatSpn(784, 2, 8, I, S), B = 4096) per `model(x)` call replayed from a HIP
graph, in the default mode (cached parameter tables checked on the device in every call) and trusting the version
counters; DPK_VERIFY_INLINE=0 in the environment restores the stand-alone check launch of round 3 for A/B runs.
Also: a write through `param.data` between two replays must show up in the next replay's results (self-healing tables).

usage: python tools/bench_config2_modes.py [B] [--shapes 2,2 8,8 16,16]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob import hip
from deeprob.spn.models import GaussianRatSpn

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
shapes = [(2, 2), (8, 8), (16, 16)]
if '--shapes' in sys.argv:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[sys.argv.index('--shapes') + 1:]]


def graph_ms(model, xs, reps=4):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.stream(side):
        for x in xs:
            model(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                for x in xs:
                    out = model(x)
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
        best = min(best, e0.elapsed_time(e1) / (5 * reps * len(xs)))
    return best, g, out


rows = []
for I, S in shapes:
    torch.manual_seed(0)
    m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).cuda().eval()
    xs = [torch.randn(B, 784, device='cuda') for _ in range(8)]
    ms_default, g, out = graph_ms(m, xs)
    # self-healing: a write through .data, invisible to the host, then the SAME graph replayed twice
    with torch.no_grad():
        want0 = m(xs[-1]).clone()
    m.base_layer.loc.data.add_(0.25)
    m.root_layer.weight.data.mul_(1.5)
    g.replay(); torch.cuda.synchronize()
    got1 = out.clone()
    g.replay(); torch.cuda.synchronize()
    got2 = out.clone()
    prev = hip.trust_version_counters(True)
    m.base_layer.loc.data.add_(0.0)
    with torch.no_grad():
        m.base_layer.loc.add_(0.0)          # (version bump: the trusted path rebuilds once)
        want1 = m(xs[-1]).clone()
    ms_trust, _, _ = graph_ms(m, xs)
    hip.trust_version_counters(prev)
    rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1.0)).max())
    rows.append({'I': I, 'S': S, 'B': B, 'us_default': round(ms_default * 1e3, 2), 'us_trust': round(ms_trust * 1e3, 2),
                 'changed_by_write': rel(want1, want0), 'replay1_vs_fresh': rel(got1, want1), 'replay2_vs_fresh': rel(got2, want1)})
    print(rows[-1], flush=True)
print(json.dumps(rows))
