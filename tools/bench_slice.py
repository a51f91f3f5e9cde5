"""Measurement: the headline model (GaussianRatSpn(784, 2, 8, 2, 2)) at the shard sizes of the metric's strong-scaling
reading (65536 / N samples per rank) on the three tile mappings of the matrix-core route: small-batch kernels, 128-sample
ring kernel, persistent 32-sample blocks with the mean table in registers (csrc/ratspn_gemm_slice.hip).

Every batch size cycles through enough resident inputs to exceed the 256 MiB Infinity Cache (a single resident input
streams at 6.7-6.9 TB/s through these kernels: never benchmark this path on one buffer).  Prints host-clock step times of
the frozen-model call (ops.FusedForwardPlan, static_params) and the max LL difference against the ring mapping; run it
under `rocprofv3 --kernel-trace` and feed the trace to tools/trace_summary.py for kernel durations by grid size.
usage: python tools/bench_slice.py [batch ...] [--steps K] [--default-mode]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import torch
from deeprob.hip import load_library
from deeprob.spn.models import GaussianRatSpn

lib = load_library()
args = sys.argv[1:]
K = int(args[args.index('--steps') + 1]) if '--steps' in args else 200
STATIC = '--default-mode' not in args   # (--default-mode: the launches check their parameter tables, as model(x) does)
batches = [int(a) for a in args if a.isdigit() and (args.index(a) == 0 or args[args.index(a) - 1] != '--steps')] or \
    [4096, 8192, 16384, 32768, 65536]
torch.manual_seed(0)
m = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
rows = []
for B in batches:
    nbuf = max(2, -(-(320 << 20) // (B * 784 * 4)))
    gen = torch.Generator('cuda').manual_seed(B)
    xs = [torch.randn(B, 784, device='cuda', generator=gen) for _ in range(min(nbuf, 64))]
    ref = None
    for mapping in (('ring', 'small', 'slice') if '--only-slice' not in args else ('slice',)):
        lib.dpk_ratspn_small_batch_max(0 if mapping == 'ring' else (1 << 40 if mapping == 'small' else -1))
        lib.dpk_ratspn_slice_batch_min(0 if mapping == 'slice' else -1)
        with torch.no_grad():
            out0 = m(xs[0]).clone()
            plans = [m.fused_plan(x, static_params=STATIC) for x in xs]
            for p in plans:
                p.run()
            torch.cuda.synchronize()
            # replayed from a HIP graph (one graph = one pass over the buffers, at least 32 steps): the eager loop is
            # host-bound below ~13 us per step
            nrep = max(1, -(-32 // len(plans)))
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(nrep):
                        for p in plans:
                            p.run()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            per = nrep * len(plans)
            reps = max(2, K // per)
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (reps * per)
            del g
        if ref is None:
            ref = out0
        diff = float((out0 - ref).abs().max() / ref.abs().max())
        rows.append({'batch': B, 'mapping': mapping, 'buffers': len(xs), 'step_us': round(dt * 1e6, 2), 'rel_diff_vs_ring': diff})
        print(rows[-1], flush=True)
lib.dpk_ratspn_small_batch_max(-1)
lib.dpk_ratspn_slice_batch_min(-2)
print(json.dumps(rows))
