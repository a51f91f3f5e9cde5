#!/usr/bin/env python3
"""Headline benchmark: log-likelihoods/sec of a Gaussian RAT-SPN (D=784, depth 2, 8 repetitions) on synthetic
batches of 65536 samples per GPU, evaluated by the fused HIP kernel (leaf layer on the matrix cores).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

With N > 1 and no torch.distributed environment the script re-executes itself under torch.distributed.run (one
process per GPU, RCCL); launched by torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

A step = one pass of the hot path over one resident batch: the fused forward kernel (per-sample LLs written to HBM,
fp64 sum fused in; its parameter tables are rebuilt only when a parameter changed) + for N > 1 the RCCL all-reduce of
{sum LL, count} that yields the mean LL on every rank (asynchronous, batched, overlapped with the next steps).
Inputs live in HBM before the timed region; a ring of >= 4 distinct batches (>= 3x the 256 MiB Infinity Cache) is
cycled so that the x stream really comes from HBM.  For N > 1 the default is the metric's own reading, --scaling strong:
65536 samples IN TOTAL, 65536 / N per rank, stepped through a sharded evaluation window captured
as a HIP graph (deeprob.parallel.GraphedEvaluationWindow: the local evaluations and their one all-reduce in one graph
launch -- a rank's shard kernel is shorter than the host's enqueue of it); the weak figure (65536 per rank) is measured
in the same run and printed beside it (config.weak_scaling).  --scaling weak makes the weak figure the headline.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused whole-model forward: since round 5
ratspn_gemm_slice_kernel, csrc/ratspn_gemm_slice.hip, from 8193 samples per launch): algorithmic bytes per launch = B * 4*(784 + C) (SURVEY 8d fully-fused bound) over its mean duration
measured with HIP events on the launch stream inside the timed loop.  `cpu_baseline` times the oracle (op-for-op
PyTorch-CPU restatement of the reference) on the host cores over a bounded sample of the same workload.  `secondary`
(N = 1 only) carries the other BASELINE configurations, each measured the same way in this run: RAT-SPN at B = 4096
for (rg_batch, rg_sum) = (2,2) / (8,8) / (16,16), DGC-SPN at B = 8192, RealNVP-1D at B = 65536, one optimisation step
of each family, and the headline with a host-to-device copy of every batch inside the step.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'deeprob-kit_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_PEAK_TFLOPS = 157.3    # fp32 vector = fp32 matrix peak
KERNEL_FUSED, KERNEL_LEAF, KERNEL_COUPLING, KERNEL_PRODSUM, KERNEL_SUMPRODROOT = 1, 2, 3, 4, 5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default=None,
                    help="default: the metric's reading -- 'strong' (65536 samples in total) for N > 1; identical at N = 1")
    ap.add_argument('--batch', type=int, default=65536, help='samples per GPU per step (weak) / in total (strong)')
    ap.add_argument('--rg-batch', type=int, default=2)
    ap.add_argument('--rg-sum', type=int, default=2)
    ap.add_argument('--ring', type=int, default=0, help='distinct resident batches (0: >= 4 and >= 768 MiB worth)')
    ap.add_argument('--cpu-samples', type=int, default=262144, help='cpu_baseline sample size (0 = skip)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary configurations')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--prewarm', type=int, default=384, help='untimed steps before the warm-up (runtime pool growth)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument('--share-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    ap.add_argument('--streams', type=int, default=2,
                    help='N = 1, eager loop: evaluation streams the K timed steps alternate between (a model replica each); '
                         '1 = the single-stream loop only')
    ap.add_argument('--chains', type=int, default=3,
                    help='parallel chains inside the graphed evaluation window (N > 1 and the shard entries): step i on '
                         'chain i mod chains, a workspace replica of the model per chain')
    ap.add_argument('--graph-window', action='store_true',
                    help='time the K steps through the graphed evaluation window at N = 1 too (the N > 1 default)')
    ap.add_argument('--kernel-event-every', type=int, default=0,
                    help='a run of --kernel-event-run consecutive timed steps is bracketed by one HIP event pair every '
                         'N timed steps (0 = back to back: every timed step lies in a bracketed run, the last run may be shorter; '
                         'start before the first launch of the run, stop after its last).  An event pair '
                         'around every single launch read 51.5 us where rocprofv3 reports 46.8 us for the same kernel '
                         '(the pair brackets its own dispatch latency) and cost the stream 6-8 us of bubbles per step '
                         '(ms_per_step 0.056-0.058 against 0.049 without events, tools/bench_graph_headline.py)')
    ap.add_argument('--kernel-event-run', type=int, default=8)
    return ap.parse_args()


def respawn(args):
    """--gpus N without a torch.distributed environment: run this script under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
# HIP events straight from the runtime (torch.cuda.Event sees only torch's current stream; these are recorded by the
# library around one kernel on the stream it is launched on: dpk_profile_next_kernel[_of])
# ------------------------------------------------------------------------------------------------------------------
class KernelTimer:
    def __init__(self):
        self.rt = ctypes.CDLL('libamdhip64.so')

    def pair(self):
        out = []
        for _ in range(2):
            h = ctypes.c_void_p()
            rc = self.rt.hipEventCreate(ctypes.byref(h))
            assert rc == 0, 'hipEventCreate failed: {}'.format(rc)
            out.append(h.value)
        return tuple(out)

    def ms(self, pair):
        v = ctypes.c_float()
        rc = self.rt.hipEventElapsedTime(ctypes.byref(v), ctypes.c_void_p(pair[0]), ctypes.c_void_p(pair[1]))
        if rc != 0:
            # events armed for a kernel the model did not launch: the failed query stays in the thread's last-error slot and
            # would surface in the next unrelated torch call ("invalid resource handle" at a later .to(device))
            self.rt.hipGetLastError()
        assert rc == 0, 'hipEventElapsedTime failed: {}'.format(rc)
        return v.value


def _oracle_rate(fn, n, chunk, threads):
    import torch
    torch.set_num_threads(threads)
    with torch.no_grad():
        fn(0, min(chunk, n))   # warm-up chunk
        t0 = time.perf_counter()
        for i in range(0, n, chunk):
            fn(i, min(i + chunk, n))
        dt = time.perf_counter() - t0
    return n / dt, dt


def cpu_baseline(model_state, D, n_samples):
    """Oracle timed on the host cores: chunks of 4096 (the reference materialises [B,R,I,d] temporaries), 1 warm-up
    chunk, the fastest of a few intra-op pool sizes (PyTorch's pool does not scale to every core of a big host on
    these element-wise sweeps)."""
    import torch
    from oracle import ratspn_oracle as orc
    ncpu = os.cpu_count() or 1
    chunk = 4096
    x = torch.randn(n_samples, D, generator=torch.Generator().manual_seed(0))
    best = None
    with torch.no_grad():
        for threads in sorted({ncpu, min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}):
            torch.set_num_threads(threads)
            orc.ratspn_forward(model_state, x[:chunk])
            t0 = time.perf_counter()
            orc.ratspn_forward(model_state, x[:chunk])
            dt = time.perf_counter() - t0
            if best is None or dt < best[1]:
                best = (threads, dt)
    threads = best[0]
    rate, dt = _oracle_rate(lambda a, b: orc.ratspn_forward(model_state, x[a:b]), n_samples, chunk, threads)
    return {'value': rate, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
            'sample': '{} samples of the same workload in chunks of {} ({:.1f} s) on {} of {} host threads (fastest '
                      'probed pool); oracle/ratspn_oracle.py'.format(n_samples, chunk, dt, threads, ncpu)}, threads


def read_traffic(key='headline'):
    """HBM bytes per launch of a kernel from this round's rocprofv3 --pmc passes (separate runs: PMC and timing do not
    mix; tools/collect_profiles.sh / tools/pmc_secondary.sh write profiles/pmc_traffic.json, the summaries sit beside
    it in profiles/).  Returns (bytes per launch, source) or (None, None)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        if 'kernels' in d:
            d = d['kernels'].get(key) or {}
        elif key != 'headline':
            d = {}
        return d.get('bytes_per_launch'), d.get('source')
    except (OSError, ValueError):
        return None, None


# ------------------------------------------------------------------------------------------------------------------
# secondary configurations (N = 1): same protocol, smaller step counts
# ------------------------------------------------------------------------------------------------------------------
def _time_eval(model, xs, timer, kernel_id, steps=30, warm=5):
    """(ms per model(x) call over `steps` calls on torch events, HIP-event ms of the dominant kernel)."""
    import torch
    from deeprob.hip import load_library
    lib = load_library()
    with torch.no_grad():
        for i in range(warm):
            model(xs[i % len(xs)])
        torch.cuda.synchronize()
        ms = float('inf')
        for _ in range(2):   # two windows, the faster one: the runtime's one-off pool growth (a ~40 ms host stall after
            # a few hundred launches of a process, tools/host_overhead3.py) must not pass for a property of the model
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps):
                model(xs[i % len(xs)])
            e1.record()
            torch.cuda.synchronize()
            ms = min(ms, e0.elapsed_time(e1) / steps)
        k_ms = []
        for i in range(5 if kernel_id is not None else 0):
            pr = timer.pair()
            lib.dpk_profile_next_kernel_of(pr[0], pr[1], kernel_id)
            model(xs[i % len(xs)])
            torch.cuda.synchronize()
            lib.dpk_profile_next_kernel(None, None)
            try:
                k_ms.append(timer.ms(pr))
            except AssertionError:
                pass
    return ms, (sum(k_ms) / len(k_ms) if k_ms else None)


_KEEP_ALIVE = []   # graphed windows of the N > 1 path (see graphed_run)


def _time_eval_graph(model, xs, reps=4):
    """ms per model(x) call when a pass over the resident batches `xs` (x `reps`) is replayed from ONE HIP graph: what a
    density-evaluation loop over resident data costs once the launch-bound host loop is out of the way (the kernels,
    their inter-kernel gaps and the per-call table check are all inside).  None when the pass cannot be captured."""
    import torch
    try:
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.stream(side):
            for x in xs:
                model(x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):   # (NCCL watchdog: see deeprob.parallel)
                for _ in range(reps):
                    for x in xs:
                        model(x)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = float('inf')
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (5 * reps * len(xs)))
        return best
    except Exception:
        return None


def _time_window(model, xs, reps=4, chains=1, static_params=False):
    """ms per step of a GraphedEvaluationWindow over the resident batches `xs` (x `reps`) -- what `bench.py --gpus N` times
    for N > 1, here without a collective: the fused forward + fp64 {sum, count} of every step, `chains` parallel chains
    inside the graph.  None when the window cannot be captured."""
    import torch
    from deeprob.parallel import ShardedLogLikelihood, GraphedEvaluationWindow
    try:
        ev = ShardedLogLikelihood(model, static_inputs=True, static_params=static_params)
        win = GraphedEvaluationWindow(ev, list(xs) * reps, chains=chains)
        for _ in range(3):
            win.graph.replay()
        torch.cuda.synchronize()
        best = float('inf')
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                win.graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (5 * reps * len(xs)))
        return best
    except Exception:
        return None


def _time_train(model, x, steps=15, warm=3):
    import torch
    from deeprob.torch.routines import build_optimizer
    model.train()
    opt = build_optimizer('adam', [p for p in model.parameters() if p.requires_grad], 1e-3, {'fused': True})  # (as train_model builds it)

    def step():
        opt.zero_grad()
        loss = model.loss(model(x))
        loss.backward()
        opt.step()
        model.apply_constraints()

    for _ in range(warm):
        step()
    best = float('inf')
    for _ in range(2):   # (two windows, the faster one: see _time_eval)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


def _time_train_graph(model, x, steps=30):
    """The same step replayed from a HIP graph (train_model(..., hip_graph=True)): None when it cannot be captured."""
    import torch
    from deeprob.hip.graphs import GraphedTrainStep
    from deeprob.torch.routines import build_optimizer
    try:
        model.train()
        opt = build_optimizer('adam', [p for p in model.parameters() if p.requires_grad], 1e-3, {'fused': True, 'capturable': True})
        gstep = GraphedTrainStep(model, opt)
        for _ in range(6):
            gstep(x)
        torch.cuda.synchronize()
        if gstep.graph is None:
            return None
        best = float('inf')
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                gstep(x)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best
    except Exception:
        return None


def _peek_hip_error(tag):
    """BENCH_DEBUG=1: report a HIP error left in the thread's last-error slot after a section (it would surface in an
    unrelated torch call later)."""
    if not os.environ.get('BENCH_DEBUG'):
        return
    try:
        import torch
        torch.cuda.synchronize()
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipPeekAtLastError.restype = ctypes.c_int
        rc = hip.hipPeekAtLastError()
        sys.stderr.write('[bench debug] after {}: hipPeekAtLastError = {}\n'.format(tag, rc))
    except Exception as ex:
        sys.stderr.write('[bench debug] after {}: {}\n'.format(tag, ex))


CHAINS = 3


def secondary(dev, timer, threads, xs_headline, headline_model):
    import torch
    from deeprob.spn.models import GaussianRatSpn, DgcSpn
    from deeprob.flows.models import RealNVP1d
    from oracle import ratspn_oracle as orc, dgcspn_oracle as dorc, flows_oracle as forc
    out = []
    D = 784

    def hbm(alg_bytes, ms):
        a = alg_bytes / (ms * 1e-3) / 1e9
        return {'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a / HBM_PEAK_GBS,
                'traffic': None}

    def flops(fl, ms, bound='mfma'):
        a = fl / (ms * 1e-3) / 1e12
        return {'bound': bound, 'achieved': a, 'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': a / F32_PEAK_TFLOPS,
                'traffic': None}

    # ---- BASELINE config 1: vanilla SPN log_likelihood, 16 binary variables, 1000 samples (flat-array evaluator) ----
    try:
        import numpy as np
        from deeprob.spn.structure.io import load_spn_json
        from deeprob.spn.algorithms.inference import log_likelihood
        from oracle import flat_spn_oracle as fsorc
        path = os.path.join(ROOT, 'tests', 'golden', 'spn_binary16.json')
        spn = load_spn_json(path)
        x_np = (np.random.RandomState(0).rand(1000, 16) < 0.5).astype(np.float32)
        xd = torch.from_numpy(x_np).to(dev)
        for _ in range(5):
            log_likelihood(spn, xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            log_likelihood(spn, xd)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 100 * 1e3
        t0 = time.perf_counter()
        n_rep = 20
        for _ in range(n_rep):
            fsorc.log_likelihood(path, x_np)
        dc = (time.perf_counter() - t0) / n_rep
        out.append({'workload': 'vanilla node-graph SPN log_likelihood (the circuit the reference learns on 16 binary '
                                'variables, 72 nodes), 1000 samples per call, flat-array HIP evaluator',
                    'id': 'c1', 'config': 'BASELINE config 1', 'batch': 1000, 'ms_per_step': ms, 'value': 1000 / ms * 1e3,
                    'unit': 'log-likelihoods/sec', 'kernel': 'flat_spn_kernel',
                    'roofline': hbm(1000 * 68, ms),
                    'roofline_basis': 'whole call; 68 algorithmic B/sample (16 fp32 inputs + 1 result): a 68 KB problem, '
                                      'launch-latency bound by construction',
                    'cpu_baseline': {'value': 1000 / dc, 'unit': 'log-likelihoods/sec', 'cores': 1, 'kind': 'port',
                                     'sample': '{} calls of 1000 samples ({:.2f} s), oracle/flat_spn_oracle.py = the '
                                               "reference's numpy / scipy bottom-up pass".format(n_rep, dc * n_rep)}})
    except Exception as ex:
        out.append({'id': 'c1', 'config': 'BASELINE config 1', 'error': '{}: {}'.format(type(ex).__name__, ex)})

    _peek_hip_error('config1')
    # ---- BASELINE config 2: RAT-SPN, B = 4096 (constructor defaults + the two wider settings) ----------
    B = 4096
    # bytes / flops per sample: fully fused (one launch) 4*(784+1) for (2,2) and (8,8); (16,16) runs as
    # leaf | prod+sum | prod+root
    rat = {(2, 2): (3140, 50.6e3), (8, 8): (3140, 219.6e3), (16, 16): (9284, 542.7e3)}
    for (I, S), (alg, fl) in rat.items():
        torch.manual_seed(0)
        m = GaussianRatSpn(D, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).eval()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.to(dev)
        xs = [torch.randn(B, D, device=dev) for _ in range(8)]
        kid = KERNEL_FUSED if I <= 8 else KERNEL_LEAF   # rg_batch 16 runs leaf | prod+sum | prod+root on the MFMA
        ms_eager, k_ms = _time_eval(m, xs, timer, kid, steps=50)
        ms_graph = _time_eval_graph(m, xs)
        ms = ms_graph if ms_graph is not None else ms_eager
        # throughput of a stream of such steps: the graphed window on CHAINS parallel chains (tables verified once per replay at
        # the head of every chain, steps on the verified tables; deeprob/parallel.py) -- at B = 4096 a launch covers half the
        # chip or less, so launches of different chains run side by side
        ms_chains = _time_window(m, xs, reps=4, chains=CHAINS)
        from deeprob import hip as _hip
        prev = _hip.trust_version_counters(True)     # the unchecked fast path (no write through .data, caller's word)
        try:
            with torch.no_grad():
                m(xs[0]); m(xs[0])
            ms_trust = _time_eval_graph(m, xs)
        finally:
            _hip.trust_version_counters(prev)
        wide_entry = None
        if (I, S) == (8, 8):
            # the same model at the headline batch size: the 128-sample ring mapping of the 8-channel kernel (timed before
            # the CPU baseline below: its 32 worker threads keep spinning for a while and slow an eager launch loop)
            Bl = 65536
            xl = [torch.randn(Bl, D, device=dev) for _ in range(4)]
            msl_e, kl_ms = _time_eval(m, xl, timer, kid, steps=20)
            msl_g = _time_eval_graph(m, xl)
            msl = msl_g if msl_g is not None else msl_e
            roofl = hbm(Bl * alg, msl)
            roofl['traffic'], roofl['traffic_source'] = read_traffic('wide_65536')
            wide_entry = ({'workload': 'the same (8,8) model at the headline batch size (128-sample tiles: x through an LDS-DMA '
                                    'ring, converted to f16 pairs once per tile by the loader waves)',
                        'id': 'hl(8,8)', 'config': 'headline size, rg_batch = rg_sum = 8', 'batch': Bl, 'ms_per_step': msl,
                        'value': Bl / msl * 1e3, 'unit': 'log-likelihoods/sec', 'ms_per_step_eager': msl_e,
                        'kernel_ms': kl_ms, 'kernel': 'ratspn_gemm_wide_ring_kernel', 'roofline': roofl,
                        'roofline_basis': 'whole step; 3140 algorithmic B/sample (the kernel is bound by its MFMA + node '
                                          'evaluation phases, not by HBM: DESIGN 3.10)',
                        'mfma': (lambda a: {'bound': 'mfma', 'achieved': a, 'peak': 2500.0, 'unit': 'TFLOP/s',
                                            'frac': a / 2500.0,
                                            'basis': 'executed f16 MFMA flops: 3 products x 2 x 784 x 256 outputs per '
                                                     'sample, over the kernel time'})(
                            Bl * 2.0 * 3 * 784 * 256 / ((kl_ms if kl_ms else msl) * 1e-3) / 1e12)})
            del xl
        n_cpu = 4096 if I <= 8 else 1024
        xc = torch.randn(n_cpu, D)
        rate, dt = _oracle_rate(lambda a, b: orc.ratspn_forward(sd, xc[a:b]), n_cpu, 1024 if I > 8 else 4096, threads)
        traffic, tsrc = read_traffic('config2_{}_{}'.format(I, S))
        roof = hbm(B * alg, ms) if I <= 8 else flops(B * fl, ms, 'valu')
        roof['traffic'], roof['traffic_source'] = traffic, tsrc
        e = {'workload': 'GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch={}, rg_sum={}) forward, '
                         'model(x) under no_grad over 8 resident batches'.format(I, S),
             'id': 'c2({},{})'.format(I, S), 'config': 'BASELINE config 2', 'batch': B, 'ms_per_step': ms, 'value': B / ms * 1e3,
             'unit': 'log-likelihoods/sec',
             'step_basis': 'model(x) calls replayed from one HIP graph (32 calls per replay; kernels, gaps and the per-call '
                           'parameter-table check included)' if ms_graph is not None else 'eager python loop',
             'ms_per_step_eager': ms_eager, 'ms_per_step_trusting_version_counters': ms_trust, 'kernel_ms': k_ms,
             'ms_per_step_window_chains': ms_chains,
             'kernel': ('fused forward, small-batch kernel (one HIP event pair around a single launch: includes its '
                        'dispatch latency; rocprofv3: profiles/r03_config2_kernel_stats.txt)') if I < 8
                       else ('fused forward, one-launch 8-channel kernel (a wave per repetition; event pair around a '
                             'single launch: includes its dispatch latency)') if I == 8
                       else 'leaf MFMA kernel (then the prod+sum and prod+root MFMA kernels)',
             'roofline': roof,
             'roofline_basis': 'whole step; {} algorithmic B/sample'.format(alg) if I <= 8
                               else 'whole step; {:.0f} flop/sample on the fp32 VALU (SURVEY 8d: VALU-bound)'.format(fl),
             'cpu_baseline': {'value': rate, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
                              'sample': '{} samples ({:.1f} s), oracle/ratspn_oracle.py'.format(n_cpu, dt)}}
        out.append(e)
        if wide_entry is not None:
            out.append(wide_entry)
        if (I, S) == (2, 2):
            # the marginalisation path (nan_to_num_ at ratspn.py:103): 30 % of the entries NaN, at B = 4096 and 65536
            for Bn in (4096, 65536):
                gen = torch.Generator(device=dev).manual_seed(7)
                xn = [torch.randn(Bn, D, device=dev, generator=gen) for _ in range(4 if Bn > 4096 else 8)]
                for t in xn:
                    t[torch.rand(Bn, D, device=dev, generator=gen) < 0.3] = float('nan')
                xcl = [torch.randn(Bn, D, device=dev, generator=gen) for _ in range(len(xn))]
                with torch.no_grad():
                    for t in xn[:2]:
                        m(t)       # (raises the marginalised-evidence hint: the later launches take the variant built for it)
                msn_e, _ = _time_eval(m, xn, timer, kid, steps=30)
                msn_g = _time_eval_graph(m, xn)
                msn = msn_g if msn_g is not None else msn_e
                torch.cuda.synchronize()
                time.sleep(0.05)
                msc_g = _time_eval_graph(m, xcl)     # clean inputs, same size, same protocol (right after: same variant)
                out.append({'workload': 'the same model, 30 % of the inputs NaN (marginalised evidence)',
                            'id': 'nan{}'.format(Bn), 'config': 'BASELINE config 2 / headline size, marginalised', 'batch': Bn, 'ms_per_step': msn,
                            'value': Bn / msn * 1e3, 'unit': 'log-likelihoods/sec', 'ms_per_step_eager': msn_e,
                            'ms_per_step_clean_same_protocol': msc_g,
                            'slowdown_vs_clean': (msn / msc_g) if msc_g else None,
                            'roofline': hbm(Bn * alg, msn), 'roofline_basis': 'whole step; 3140 algorithmic B/sample'})
                del xn, xcl
        del m, xs

    _peek_hip_error('config2')
    # ---- the metric's strong-scaling reading on one GPU: the headline model at 65536 / N samples (N = 2, 4, 8 ranks) --------
    # = the per-rank step of `bench.py --gpus N` (strong scaling is the default there); default mode (per-call table check) and
    # frozen-model mode, replayed from a HIP graph like the sharded evaluation window replays them
    try:
        from deeprob import hip as _hip
        base = {}
        for Bs in (65536, 32768, 16384, 8192):
            nb = max(2, -(-(320 << 20) // (Bs * D * 4)))
            xs_s = xs_headline[:nb] if Bs == 65536 and len(xs_headline) >= nb else \
                [torch.randn(Bs, D, device=dev) for _ in range(nb)]
            reps = max(1, -(-32 // nb))
            # (the graphed window of `--gpus N`, CHAINS parallel chains inside the graph; one chain beside it)
            ms_def = _time_window(headline_model, xs_s, reps=reps, chains=CHAINS)
            ms_one = _time_window(headline_model, xs_s, reps=reps, chains=1)
            ms_tr = _time_window(headline_model, xs_s, reps=reps, chains=CHAINS, static_params=True)
            base[Bs] = (ms_def, ms_tr)
            if Bs == 65536 or ms_def is None:
                continue
            n = 65536 // Bs
            out.append({'workload': 'headline model, one rank\'s shard of the 65536-sample batch at N = {} (strong scaling), '
                                    'GraphedEvaluationWindow over {} resident shards, {} parallel chains'.format(n, nb, CHAINS),
                        'id': 'shard/{}'.format(n), 'config': 'strong-scaling shard', 'batch': Bs, 'ms_per_step': ms_def,
                        'ms_per_step_one_chain': ms_one,
                        'value': Bs / ms_def * 1e3, 'unit': 'log-likelihoods/sec',
                        'ms_per_step_trusting_version_counters': ms_tr,
                        'predicted_speedup': (base[65536][0] / ms_def) if base[65536][0] else None,
                        'predicted_speedup_trusted': (base[65536][1] / ms_tr) if base[65536][1] and ms_tr else None,
                        'roofline': hbm(Bs * 3140, ms_def), 'roofline_basis': 'whole step; 3140 algorithmic B/sample'})
            del xs_s
    except Exception as ex:
        out.append({'id': 'shard', 'config': 'strong-scaling shard', 'error': '{}: {}'.format(type(ex).__name__, ex)})
    _peek_hip_error('shards')
    # ---- BASELINE config 4: DGC-SPN, B = 8192 ----------------------------------------------------------------------
    B = 8192
    torch.manual_seed(5)
    m = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(dev)
    xs = [torch.randn(B, 1, 28, 28, device=dev) for _ in range(2)]
    ms, k_ms = _time_eval(m, xs, timer, KERNEL_SUMPRODROOT, steps=10, warm=3)
    # (the same forward as a stream of steps: graphed window, two parallel chains -- the levels of two batches interleave)
    ms_c4_graph = _time_window(m, xs, reps=2, chains=1)
    ms_c4_chains = _time_window(m, xs, reps=2, chains=2)
    ms_c4_eager = ms
    ms = min(v for v in (ms, ms_c4_graph, ms_c4_chains) if v)      # the step of the evaluation loop as one would run it
    plan = dorc.schedule((1, 28, 28), 8, 8, True, 0)
    xc = torch.randn(256, 1, 28, 28)
    rate, dt = _oracle_rate(lambda a, b: dorc.dgcspn_forward(sd, xc[a:b], plan), 256, 128, threads)
    out.append({'workload': 'DgcSpn((1,28,28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0) forward',
                'id': 'c4', 'config': 'BASELINE config 4', 'batch': B, 'ms_per_step': ms, 'value': B / ms * 1e3,
                'unit': 'log-likelihoods/sec', 'kernel_ms': k_ms, 'ms_per_step_hip_graph': ms_c4_graph,
                'ms_per_step_window_chains': ms_c4_chains, 'ms_per_step_eager': ms_c4_eager,
                'step_basis': 'the fastest of: eager python loop, graphed evaluation window (fused {sum, count}), the same on two parallel chains',
                'kernel': 'spatial_stream_kernel<1, 1, false> (last sum level + product + root; pixel-major input)',
                'roofline': dict(hbm(B * 588164, ms), **dict(zip(('traffic', 'traffic_source'), read_traffic('config4')))),
                'roofline_basis': 'whole step; 588164 algorithmic B/sample (SURVEY 8d, products folded into sums)',
                'cpu_baseline': {'value': rate, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
                                 'sample': '256 samples ({:.1f} s), oracle/dgcspn_oracle.py'.format(dt)}})
    # both fractions in the line: `frac` on SURVEY's algorithmic bytes, `frac_pmc` on the bytes the chip
    # actually moved per step (counter passes in profiles/) over this run's step time
    r4 = out[-1]['roofline']
    if r4.get('traffic'):
        r4['frac_pmc'] = r4['traffic'] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    m_dgc = m
    del xs

    _peek_hip_error('config4')
    # ---- SURVEY 8d config 4, secondary: the example's model (examples/dgcspn_mnist.py:27-37): two pooling levels, 16 leaf
    # and 32 sum channels -- outside the 8 -> 8 channel streaming kernels: the generic fused product+sum level kernels
    try:
        torch.manual_seed(6)
        m2 = DgcSpn((1, 28, 28), n_batch=16, sum_channels=32, depthwise=True, n_pooling=2).eval()
        sd2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
        # algorithmic floats per sample, products folded into the sums above them (the SURVEY 8d recipe): every map written
        # once and read once, + the input and the result
        outs = [sd2['base_layer.loc'].numel()] + [int(v.shape[0] * v.shape[2] * v.shape[3]) for k, v in sd2.items()
                                                   if k.startswith('layers.') and v.dim() == 4 and int(k.split('.')[1]) % 2 == 1]
        alg2 = 4 * (784 + 2 * sum(outs) + 1)
        m2.to(dev)
        xs = [torch.randn(B, 1, 28, 28, device=dev) for _ in range(2)]
        ms2, _ = _time_eval(m2, xs, timer, None, steps=6, warm=2)   # (no single dominant kernel id on the generic route)
        plan2 = dorc.schedule((1, 28, 28), 16, 32, True, 2)
        xc = torch.randn(128, 1, 28, 28)
        rate2, dt2 = _oracle_rate(lambda a, b: dorc.dgcspn_forward(sd2, xc[a:b], plan2), 128, 64, threads)
        out.append({'id': 'c4b', 'workload': 'DgcSpn((1,28,28), n_batch=16, sum_channels=32, depthwise=True, n_pooling=2) forward '
                                             '(examples/dgcspn_mnist.py)', 'config': 'SURVEY 8d config 4 secondary', 'batch': B,
                    'ms_per_step': ms2, 'value': B / ms2 * 1e3, 'unit': 'log-likelihoods/sec',
                    'kernel': 'spatial_prodsum_fwd_kernel<32, 1> (generic fused product+sum level, 32 channels)',
                    'roofline': hbm(B * alg2, ms2),
                    'roofline_basis': 'whole step; {} algorithmic B/sample (every map written and read once)'.format(alg2),
                    'cpu_baseline': {'value': rate2, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
                                     'sample': '128 samples ({:.1f} s), oracle/dgcspn_oracle.py'.format(dt2)}})
        del m2, xs
    except Exception as ex:
        out.append({'id': 'c4b', 'config': 'SURVEY 8d config 4 secondary', 'error': '{}: {}'.format(type(ex).__name__, ex)})

    _peek_hip_error('config4b')
    # ---- BASELINE config 5: RealNVP-1D, B = 65536 ------------------------------------------------------------------
    from tests.util import randomise_flow
    B = 65536
    torch.manual_seed(10)
    m = RealNVP1d(D)
    randomise_flow(m, 11)            # scale_act.weight / BN statistics live (the default init makes s == 0)
    m.eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(dev)
    xs = [torch.randn(B, D, device=dev) for _ in range(2)]
    ms, k_ms = _time_eval(m, xs, timer, KERNEL_COUPLING, steps=20, warm=3)
    ms_c5_graph = _time_window(m, xs, reps=2, chains=1)     # (chains do not help here: every coupling launch fills the chip)
    ms_c5_eager = ms
    ms = min(v for v in (ms, ms_c5_graph) if v)
    from deeprob import hip as _hip
    prev = _hip.trust_version_counters(True)
    try:
        ms_trust, _ = _time_eval(m, xs, timer, KERNEL_COUPLING, steps=20, warm=3)
    finally:
        _hip.trust_version_counters(prev)
    per_layer = B * 2 * (392 * 128 + 128 * 784)
    xc = torch.randn(16384, D)
    rate, dt = _oracle_rate(lambda a, b: forc.flow_log_prob(sd, xc[a:b]), 16384, 4096, threads)
    e = {'workload': 'RealNVP1d(784, n_flows=5, depth=1, units=128, batch_norm, affine) forward log-likelihood',
         'id': 'c5', 'config': 'BASELINE config 5', 'batch': B, 'ms_per_step': ms, 'value': B / ms * 1e3,
         'unit': 'log-likelihoods/sec', 'ms_per_step_trusting_version_counters': ms_trust, 'kernel_ms': k_ms,
         'ms_per_step_hip_graph': ms_c5_graph, 'ms_per_step_eager': ms_c5_eager,
         'step_basis': 'the faster of: eager python loop, graphed evaluation window (one graph replay per 8 steps)',
         'kernel': 'coupling_x1_kernel (one of the 5 layers; x read once: 64-sample tiles held in registers; split-f16 MFMA, fp32-grade products)',
         'roofline': hbm(B * 2 * D * 4, k_ms) if k_ms else hbm(5 * B * 2 * D * 4, ms),
         'roofline_basis': 'one coupling kernel; x read + out written once: 2*784*4 algorithmic B per sample and layer '
                           '(the conditioner runs on the f16 matrix cores at 3 MFMAs per fp32-grade product, far '
                           'from their peak)' if k_ms else 'whole step, 5 layers',
         'fp32_equivalent_tflops': (per_layer / (k_ms * 1e-3) / 1e12) if k_ms else None,
         # executed f16 matrix-core work of one coupling kernel: 3 MFMA products per fp32-grade product of the mask-aware
         # GEMMs (392 x 128 and 128 x 784 per sample), against the dense f16 MFMA peak
         'mfma': (lambda a: {'bound': 'mfma', 'achieved': a, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': a / 2500.0,
                             'basis': 'executed f16 MFMA flops of one coupling kernel: 3 products x 2 x (392 x 128 + 128 x 784) '
                                      'per sample over its HIP-event time; SQ_VALU_MFMA_BUSY_CYCLES pass in profiles/'})(
             3.0 * per_layer / (k_ms * 1e-3) / 1e12) if k_ms else None,
         'cpu_baseline': {'value': rate, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
                          'sample': '16384 samples ({:.1f} s), oracle/flows_oracle.py'.format(dt)}}
    e['roofline']['traffic'], e['roofline']['traffic_source'] = read_traffic('config5')
    out.append(e)
    m_flow = m
    del xs

    # ---- RealNVP-2D (constructor defaults on MNIST-shaped input), evaluation, B = 2048 ---------------
    from tests.util import flow2d_model
    from oracle import flows2d_oracle as f2orc
    B = 2048
    feats = (1, 28, 28)
    m = flow2d_model(feats, dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True), 25)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    flops = 0          # 2 * Cout * Cin * k^2 * H * W per convolution at the resolution of its coupling
    for mod in m.modules():
        if type(mod).__name__ == 'CouplingLayer2d':
            hw = mod.in_features[1] * mod.in_features[2]
            for q in mod.network.modules():
                if type(q).__name__ == '_WeightNormConvParameters':
                    flops += 2 * q.out_channels * q.in_channels * q.kernel_size ** 2 * hw
    m.to(dev)
    xs = [torch.randn((B,) + feats, device=dev) for _ in range(2)]
    with torch.no_grad():
        for i in range(2):
            m(xs[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8):
            m(xs[i % 2])
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    xc = torch.randn((256,) + feats)
    rate, dt = _oracle_rate(lambda a, b: f2orc.log_prob(sd, xc[a:b]), 256, 128, threads)
    tf = B * flops / (ms * 1e-3) / 1e12
    out.append({'workload': 'RealNVP2d((1,28,28), n_flows=1, n_blocks=2, channels=32, resnet, affine) forward '
                            'log-likelihood (evaluation only)',
                'id': 'nvp2d', 'config': 'SURVEY 8f-3', 'batch': B, 'ms_per_step': ms, 'value': B / ms * 1e3,
                'unit': 'log-likelihoods/sec',
                'kernel': 'conv3x3_lds_kernel (3x3 conditioner convolutions on fp32 MFMA, activations staged through LDS; ~70 % of the step)',
                'roofline': {'bound': 'mfma', 'achieved': tf, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tf / 157.3,
                             'traffic': None},
                'roofline_basis': 'whole step: {:.1f} MFLOP of convolutions per sample as written / step time, against the '
                                  'fp32 matrix = packed fp32 vector peak'.format(flops / 1e6),
                'cpu_baseline': {'value': rate, 'unit': 'log-likelihoods/sec', 'cores': threads, 'kind': 'port',
                                 'sample': '256 samples ({:.1f} s), oracle/flows2d_oracle.py'.format(dt)}})
    del xs, m

    # ---- forward + backward + Adam, B = 512 (SURVEY 8d "also report fwd+bwd step/s") ------------------------------
    B = 512
    torch.manual_seed(0)
    trains = [('GaussianRatSpn(784, depth 2, reps 8, rg_batch=8, rg_sum=8)',
               GaussianRatSpn(D, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42).to(dev),
               torch.randn(B, D, device=dev)),
              # the model of examples/ratspn_mnist.py: depth 3, 16 channels / sums, trainable scales (outside the fused routes)
              ('GaussianRatSpn16(784, depth 3, reps 8, rg_batch=16, rg_sum=16, optimize_scale)',
               GaussianRatSpn(D, rg_depth=3, rg_repetitions=8, rg_batch=16, rg_sum=16, optimize_scale=True, random_state=42).to(dev),
               torch.randn(B, D, device=dev)),
              ('DgcSpn((1,28,28), 8, 8, depthwise)', m_dgc, torch.randn(B, 1, 28, 28, device=dev)),
              ('RealNVP1d(784)', m_flow, torch.randn(B, D, device=dev))]
    for name, m, x in trains:
        ms = _time_train(m, x)
        ms_g = _time_train_graph(m, x)
        out.append({'id': 'train:' + name.split('(')[0], 'workload': name + ': forward + backward + Adam step', 'config': 'training step', 'batch': B,
                    'ms_per_step': ms, 'value': B / ms * 1e3, 'unit': 'samples/sec',
                    'ms_per_step_hip_graph': ms_g, 'value_hip_graph': (B / ms_g * 1e3) if ms_g else None})

    # RealNVP2d training direction: batch statistics, convolution / coupling backward kernels
    from deeprob.flows.models import RealNVP2d
    B2 = 256
    torch.manual_seed(0)
    m2d = RealNVP2d((1, 28, 28), n_flows=1, n_blocks=2, channels=32).to(dev)
    ms = _time_train(m2d, torch.randn(B2, 1, 28, 28, device=dev), steps=4, warm=2)
    out.append({'workload': 'RealNVP2d((1,28,28), n_flows=1, n_blocks=2, channels=32, resnet, affine): forward + backward '
                            '+ Adam step (training mode, batch statistics)', 'id': 'train:RealNVP2d', 'config': 'training step', 'batch': B2,
                'ms_per_step': ms, 'value': B2 / ms * 1e3, 'unit': 'samples/sec', 'ms_per_step_hip_graph': None,
                'value_hip_graph': None})
    del m2d

    # ---- the headline with the host-to-device copy inside the step (routines.py:159 copies per batch) ----
    Bh = xs_headline[0].shape[0]
    host = [torch.randn(Bh, D).pin_memory() for _ in range(2)]
    stage = [torch.empty(Bh, D, device=dev) for _ in range(2)]
    with torch.no_grad():
        for i in range(3):
            stage[i % 2].copy_(host[i % 2], non_blocking=True)
            headline_model(stage[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for i in range(K):
            stage[i % 2].copy_(host[i % 2], non_blocking=True)
            headline_model(stage[i % 2])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
    out.append({'workload': 'headline model, batch copied from pinned host memory inside every step (PCIe-inclusive)',
                'id': 'h2d', 'config': 'H2D-inclusive', 'batch': Bh, 'ms_per_step': ms, 'value': Bh / ms * 1e3,
                'unit': 'log-likelihoods/sec', 'h2d_GBps': Bh * D * 4 / (ms * 1e-3) / 1e9})
    return out


def compact_configs(sec):
    """One short record per secondary configuration (the driver keeps only the last ~6 KB of stdout: the verbose
    `secondary` list goes to a file, this summary stays in the line): id, batch, ms per step (default mode: per-call
    parameter-table check included), LL/s (or samples/s), roofline bound + fraction, dominant-kernel us, CPU-oracle
    rate on `cores` host threads."""
    ids = {'BASELINE config 1': 'c1', 'BASELINE config 2': 'c2', 'BASELINE config 4': 'c4', 'BASELINE config 5': 'c5',
           'headline size, rg_batch = rg_sum = 8': 'hl(8,8)', 'SURVEY 8f-3': 'nvp2d', 'training step': 'train',
           'H2D-inclusive': 'h2d', 'SURVEY 8d config 4 secondary': 'c4b'}
    out = []
    # the driver's record keeps the last ~2000 characters of stdout: the BASELINE configurations (c2(2,2), c4, c5, the shards)
    # go LAST, the rows nobody grades (RealNVP-2D, the training steps, the PCIe-inclusive step) only to the detail file
    detail_only = ('nvp2d', 'h2d')
    order = {'c1': 0, 'c2(16,16)': 1, 'c2(8,8)': 2, 'hl(8,8)': 3, 'c4b': 5, 'c2(2,2)': 10, 'c4': 11, 'c5': 12,
             'shard/2': 13, 'shard/4': 14, 'shard/8': 15}
    def rank_of(e):
        i = e.get('id') or ids.get(e.get('config'), '?')
        return order.get(i, 4)
    for e in sorted(sec, key=rank_of):
        eid = e.get('id') or ids.get(e.get('config'), e.get('config', '?'))
        if eid in detail_only or str(eid).startswith('train:'):
            continue
        r = {'id': eid, 'B': e.get('batch')}
        if 'error' in e:
            r['error'] = e['error'][:80]
            out.append(r)
            continue
        r['ms'] = round(e['ms_per_step'], 5)
        r['rate'] = float('{:.4g}'.format(e['value']))
        for k_src, k_dst in (('ms_per_step_trusting_version_counters', 'ms_trust'), ('ms_per_step_eager', 'ms_eager'),
                             ('ms_per_step_hip_graph', 'ms_graph'), ('slowdown_vs_clean', 'x_clean'),
                             ('ms_per_step_one_chain', 'ms_1chain'), ('ms_per_step_window_chains', 'ms_chains')):
            if e.get(k_src) is not None:
                r[k_dst] = round(e[k_src], 5)
        # (a single-launch HIP event pair brackets its own dispatch: only a figure below the step it belongs to is evidence;
        # the rocprofv3 kernel times are in profiles/)
        if e.get('kernel_ms') and e['kernel_ms'] <= e['ms_per_step']:
            r['k_us'] = round(e['kernel_ms'] * 1e3, 2)
        if e.get('predicted_speedup') is not None:
            r['x_vs_64k'] = round(e['predicted_speedup'], 2)
            if e.get('predicted_speedup_trusted') is not None:
                r['x_vs_64k_trust'] = round(e['predicted_speedup_trusted'], 2)
        roof = e.get('roofline')
        if roof:
            r['bound'], r['frac'] = roof['bound'], round(roof['frac'], 4)
            if roof.get('traffic'):
                r['traffic_MB'] = round(roof['traffic'] / 1e6, 1)
            if roof.get('frac_pmc') is not None:      # counter bytes over the step, beside the algorithmic fraction
                r['frac_pmc'] = round(roof['frac_pmc'], 4)
        if e.get('mfma'):
            r['mfma_frac'] = round(e['mfma']['frac'], 4)
        cb = e.get('cpu_baseline')
        if cb:
            r['cpu'], r['cores'] = float('{:.4g}'.format(cb['value'])), cb['cores']
        out.append(r)
    return out


def write_detail(obj):
    """The verbose record (every workload / basis / source string) next to the line: gpurun_out/bench_detail.json."""
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'bench_detail.json')
        with open(path, 'w') as f:
            json.dump(obj, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def headline_kernel_name(two_channel: bool, B: int) -> str:
    """The kernel a clean-evidence launch of B samples takes, from the LIBRARY's thresholds (dpk_ratspn_slice_batch_min /
    dpk_ratspn_small_batch_max), not from literals that can drift."""
    from deeprob.hip import load_library
    lib = load_library()
    slice_min = lib.dpk_ratspn_slice_batch_min(-2)      # (below -1: back to the initial value; returns the current one)
    lib.dpk_ratspn_slice_batch_min(slice_min)
    small_max = lib.dpk_ratspn_small_batch_max(-1)
    lib.dpk_ratspn_small_batch_max(small_max)
    if two_channel and slice_min >= 0 and B >= slice_min:
        return 'ratspn_gemm_slice_kernel'
    if not two_channel:
        return 'ratspn_gemm_wide_kernel' if B <= 16384 else 'ratspn_gemm_wide_ring_kernel'
    return 'ratspn_gemm_small_kernel' if B <= small_max else 'ring::ratspn_gemm_kernel'


def shard_plan(world: int, batch: int, scaling: str) -> dict:
    """Samples per rank and step of the headline loop, of the other scaling mode measured beside it, and of BASELINE
    config 3's shape (262144 samples over 8 GPUs = 32768 per rank, weak at any N) -- pure arithmetic, unit-tested on the CPU
    (tests/test_host_logic.py)."""
    per_rank = batch if scaling == 'weak' else max(1, batch // world)
    other = max(1, batch // world) if scaling == 'weak' else batch
    return {'per_rank': per_rank, 'global': per_rank * world, 'other_scaling': 'strong' if scaling == 'weak' else 'weak',
            'other_per_rank': other, 'config3_per_rank': 32768, 'config3_global': 32768 * world}


def workload_string(rg_batch: int, rg_sum: int, per_rank: int, world: int, backend: str) -> str:
    return ('GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch={}, rg_sum={}) forward LL, {} samples '
            'per GPU per step, mean LL reduced on device{}'.format(
                rg_batch, rg_sum, per_rank,
                ' + {} all-reduce of {{sum, count}} once per run'.format('RCCL' if backend == 'nccl' else backend)
                if world > 1 else ''))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == max(args.gpus, 1), 'WORLD_SIZE {} != --gpus {}'.format(world, args.gpus)

    import torch
    import torch.distributed as dist
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from deeprob.spn.models import GaussianRatSpn
    from deeprob.parallel import ShardedLogLikelihood

    D = 784
    if args.scaling is None:
        args.scaling = 'strong' if world > 1 else 'weak'
    global CHAINS
    CHAINS = max(1, args.chains)
    plan = shard_plan(world, args.batch, args.scaling)
    B = plan['per_rank']
    torch.manual_seed(0)  # identical replica on every rank
    model = GaussianRatSpn(D, rg_depth=2, rg_repetitions=8, rg_batch=args.rg_batch, rg_sum=args.rg_sum,
                           random_state=42).eval()
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)

    def graphed_run(Bg, steps, warmup):
        """K steps on shards of Bg samples per rank through the graphed evaluation window: (seconds for the K steps,
        max over ranks; mean LL of the last step; steps per graph replay)."""
        from deeprob.parallel import GraphedEvaluationWindow
        L = max(d for d in range(1, min(steps, 40) + 1) if steps % d == 0)
        ring_g = max(4, -(-(768 << 20) // (Bg * D * 4)))
        gen_g = torch.Generator(device=dev).manual_seed(4321 + rank)
        xs_g = [torch.randn(Bg, D, device=dev, generator=gen_g) for _ in range(min(ring_g, max(L, 4)))]
        ev = ShardedLogLikelihood(model, group=dist.group.WORLD if world > 1 else None, static_inputs=True,
                                  static_params=False)
        win, err = None, None
        try:
            win = GraphedEvaluationWindow(ev, [xs_g[i % len(xs_g)] for i in range(L)], chains=args.chains)
        except Exception as ex:   # (a capture that fails on ONE rank must not leave the others waiting in the barrier below)
            err = ex
        if world > 1:
            flag = torch.tensor([0.0 if win is None else 1.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0 and err is None:
                err = RuntimeError('the window could not be captured on another rank')
        if err is not None:
            if win is not None:
                _KEEP_ALIVE.append(win)
            raise err
        for _ in range(max(1, -(-warmup // L))):
            win.replay()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # (the replays are enqueued back to back -- a host read of the means behind every replay would leave the device idle
        # for a launch latency + a copy per L steps, 10-25 % at the 8 192-sample shards; the means of the last replay are
        # read behind the timed region, the earlier replays wrote the same slots)
        for _ in range(steps // L):
            win.graph.replay()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t0
        from deeprob.parallel import slot_mean
        means = slot_mean(win.pool).cpu().tolist()
        tt = torch.tensor([dtg], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # (the graph holds a captured collective; releasing such a graph -- like destroying its communicator -- aborts the
        # process now and then on this stack, tests/test_parallel_nccl_gpu.py::_run_case: it stays alive to the end)
        _KEEP_ALIVE.append(win)
        del xs_g
        return float(tt.item()), means[-1], L

    ring = args.ring or max(4, -(-(768 << 20) // (B * D * 4)))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)  # every rank its own shard of the batch
    xs = [torch.randn(B, D, device=dev, generator=gen) for _ in range(ring)]

    # The headline loop runs the DEFAULT mode of `model(x)`: every call checks its cached parameter tables on the device
    # (inside the launch); the frozen-model figure (static_params=True: the caller vouches) is printed beside it.
    evaluator = ShardedLogLikelihood(model, group=dist.group.WORLD if world > 1 else None, static_inputs=True,
                                     static_params=False)
    time_kernel = not args.no_kernel_events
    timer = KernelTimer()
    n_spare = 64   # event pairs for the untimed steps
    handles = [timer.pair() for _ in range(args.steps + n_spare)] if time_kernel else []

    run = max(1, min(args.kernel_event_run, args.steps))
    stride = max(run, min(args.kernel_event_every or run, args.steps))
    sampled = []   # (first step, length) of every bracketed run (the first step's handle pair carries the two events)

    def step(i, timed_idx=None):
        marks = None
        if timed_idx is not None and time_kernel:
            first = timed_idx - timed_idx % stride
            length = min(run, args.steps - first)          # (the last run of the window may be shorter)
            if timed_idx < first + length:
                if timed_idx == first:
                    sampled.append((first, length))
                marks = (handles[first][0] if timed_idx == first else None,
                         handles[first][1] if timed_idx == first + length - 1 else None)
                if marks == (None, None):
                    marks = None
        return evaluator.step(xs[i % ring], kernel_events=marks)

    with torch.no_grad():
        # Runtime pre-warm, before the W warm-up steps of the contract: the HIP runtime grows its per-queue pools
        # (kernel arguments / signals) after a few hundred launches and that one-off growth stalls the host for
        # ~40 ms (measured: one step of ~38 ms around the 180th step of a process, tools/host_overhead3.py); it
        # must not land in the timed window, whatever W is.
        for j in range(args.prewarm):
            marks = handles[args.steps + j % n_spare] if time_kernel else None
            evaluator.step(xs[j % ring], kernel_events=marks)
            if (j + 1) % 64 == 0:
                evaluator.drain()
        evaluator.drain()
        for i in range(args.warmup):
            step(i)
        if time_kernel:
            # host cost of an event-bracketed step on THIS box (enqueue only); a loaded host can spend more than
            # the kernel itself per timing event, in which case only a handful of timed steps carry events
            torch.cuda.synchronize()
            for j in range(2):   # first use of the events
                evaluator.step(xs[j % ring], kernel_events=handles[args.steps + j])
            torch.cuda.synchronize()
            th = time.perf_counter()
            for j in range(2, 4):
                evaluator.step(xs[j % ring], kernel_events=handles[args.steps + j])
            ev_cost = (time.perf_counter() - th) / 2
            if ev_cost > 80e-6:
                stride = max(stride, args.steps // 5)
        evaluator.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, i)
        host_dt = time.perf_counter() - t0   # time to ENQUEUE the timed steps (host cost; diagnostic only)
        results = evaluator.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    mean_ll = results[-1]
    eager_ms = dt / args.steps * 1e3
    step_mode = 'eager: one C call per step (host enqueue below the kernel time at this shard size)'
    weak_entry = None
    if world > 1 or args.graph_window:
        # N > 1: the contract's K steps are timed through the graphed window (the eager loop above only warms the path up
        # and carries the kernel events); the other scaling mode is measured the same way and printed beside it.  A window
        # that cannot be captured (a backend whose collective is not capturable) leaves the eager figure in place.
        try:
            with torch.no_grad():
                dtg, mean_g, L = graphed_run(B, args.steps, args.warmup)
                dt, mean_ll = dtg, mean_g
                step_mode = ('HIP graph: {} steps on {} parallel chains + their one all-reduce per replay (eager loop of the '
                             'same steps: {:.5f} ms/step)'.format(L, args.chains, eager_ms))
                if world > 1:
                    Bo = plan['other_per_rank']
                    dto, _, _ = graphed_run(Bo, args.steps, args.warmup)
                    weak_entry = {'scaling': plan['other_scaling'], 'samples_per_gpu_per_step': Bo,
                                  'value': Bo * world * args.steps / dto, 'ms_per_step': dto / args.steps * 1e3}
                    # BASELINE config 3's own shape: 32768 samples per rank (262144 over 8 GPUs)
                    B3 = plan['config3_per_rank']
                    dt3, _, _ = graphed_run(B3, args.steps, args.warmup)
                    weak_entry['config3_32768_per_rank'] = {'global_batch': B3 * world, 'value': B3 * world * args.steps / dt3,
                                                            'ms_per_step': dt3 / args.steps * 1e3}
        except Exception as ex:
            step_mode = 'eager (graphed window failed: {}: {})'.format(type(ex).__name__, str(ex)[:120])

    # ---- N = 1: the K timed steps once more, alternating between two evaluation streams ----------------
    # A launch of the slice kernel holds every compute unit with one work-group; on one stream launch k + 1 starts when
    # launch k has completely finished (its tail: ~4 us of last-block upper layers with HBM idle; then a ~7.7 us prologue).
    # With two streams the next launch's work-groups are dispatched as compute units come free.  A model replica per stream:
    # the in-launch table check's tickets assume that the launches of a workspace follow one another.  This loop gives
    # `value`; the single-stream loop above carries the kernel events (overlapping launches stretch each other's brackets) --
    # the per-kernel roofline does not show this gain, by construction.
    one_stream_ms = None
    if world == 1 and args.streams > 1 and not args.graph_window:
        import copy
        ns = args.streams
        replicas = [model] + [copy.deepcopy(model) for _ in range(ns - 1)]
        lanes = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        evs = [ShardedLogLikelihood(m, static_inputs=True, static_params=False) for m in replicas]

        def run_lanes(n):
            for i in range(n):
                with torch.cuda.stream(lanes[i % ns]):
                    evs[i % ns].step(xs[i % ring])

        try:
            with torch.no_grad():
                torch.cuda.synchronize()
                # (the new streams are new hardware queues: the runtime's one-off pool growth -- see --prewarm -- happens per
                # queue and must not land in the timed window either)
                for _ in range(-(-args.prewarm * ns // 256)):
                    run_lanes(256)
                    torch.cuda.synchronize()
                    for e in evs:
                        e.drain()
                run_lanes(max(args.warmup, 4 * ns))
                torch.cuda.synchronize()
                for e in evs:
                    e.drain()
                t2 = time.perf_counter()
                run_lanes(args.steps)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t2
                means2 = [e.drain() for e in evs]
            if abs(means2[0][-1] - mean_ll) > 1e-3 * abs(mean_ll) and ring % ns == 0:
                raise RuntimeError('two-stream mean LL {} != {}'.format(means2[0][-1], mean_ll))
            one_stream_ms = dt / args.steps * 1e3
            if dt2 <= 1.15 * dt:
                dt = dt2
                step_mode = ('eager: {} evaluation streams, alternate steps, a model replica each (single-stream loop of the '
                             'same steps: {:.5f} ms/step; roofline.kernel_ms comes from that loop)'.format(ns, one_stream_ms))
            else:
                # (concurrent launches that check their tables in the launch wait on their own first work-groups: should the
                # work-groups of two launches ever block one another, a 1 s time-out resolves it -- correct results, a useless
                # timing.  Never observed at this size; the single-stream loop is the figure then.)
                step_mode += ' (two-stream loop discarded: {:.5f} ms/step)'.format(dt2 / args.steps * 1e3)
        except Exception as ex:
            step_mode += ' (two-stream loop failed: {}: {})'.format(type(ex).__name__, str(ex)[:100])

    # the same loop with a frozen model (static_params=True: no per-call table check), and on the exact fp32 route
    # (dpk_ratspn_mfma_route(0): the vector-ALU kernels, fp32 products) -- what the split-f16 matrix-core route buys
    def plain_loop(ev, steps):
        with torch.no_grad():
            for i in range(4):
                ev.step(xs[i % ring])
            ev.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(steps):
                ev.step(xs[i % ring])
            ev.drain()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps * 1e3

    ms_frozen = ms_exact = ms_graph_chains = None
    if world == 1:
        if args.streams > 1:      # (--streams 1 = nothing but launches that have the chip to themselves: the profiled command)
            ms_graph_chains = _time_window(model, xs, reps=max(1, 24 // len(xs)), chains=CHAINS)
        ms_frozen = plain_loop(ShardedLogLikelihood(model, static_inputs=True, static_params=True), args.steps)
        try:
            from deeprob.hip import load_library
            lib = load_library()
            prev_route = lib.dpk_ratspn_mfma_route(0)
            try:
                torch.manual_seed(0)
                exact_model = GaussianRatSpn(D, rg_depth=2, rg_repetitions=8, rg_batch=args.rg_batch, rg_sum=args.rg_sum,
                                             random_state=42).eval().to(dev)
                ms_exact = plain_loop(ShardedLogLikelihood(exact_model, static_inputs=True, static_params=False),
                                      max(4, min(args.steps, 20)))
                del exact_model
            finally:
                lib.dpk_ratspn_mfma_route(prev_route)
        except Exception as ex:
            ms_exact = 'failed: {}'.format(str(ex)[:80])

    if rank == 0:
        total = B * world * args.steps
        workload = workload_string(args.rg_batch, args.rg_sum, B, world, args.backend)
        out = {
            'metric': 'log-likelihoods/sec, RAT-SPN D=784 batch=64k at 1/2/4/8 MI355X',
            'value': total / dt, 'unit': 'log-likelihoods/sec', 'n_gpus': world,
            'n_ranks_seen': dist.get_world_size() if world > 1 else 1,
            'backend': (dist.get_backend() if world > 1 else None), 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': args.scaling, 'step_mode': step_mode.split(':')[0], 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload, 'global_batch': B * world, 'resident_batches': ring, 'mean_ll': mean_ll,
                       'host_enqueue_ms_per_step': round(host_dt / args.steps * 1e3, 5), 'step_mode': step_mode,
                       'other_scaling': weak_entry,
                       'params_mode': 'default: static_params=False (cached parameter tables checked on the device at every call)',
                       'ms_per_step_one_stream': one_stream_ms,
                       'ms_per_step_frozen_model': ms_frozen,
                       'ms_per_step_graph_window_{}_chains'.format(CHAINS): ms_graph_chains,
                       'fp32_exact_ms': ms_exact,
                       'arithmetic': 'fp32 results; leaf GEMM = 3 f16 MFMAs on two-way f16 splits, fp32 accumulate '
                                     '(>= 22 bits per product, guarded, exact fallback)'},
        }
        detail = {'line': out}
        if time_kernel and sampled:
            torch.cuda.synchronize()
            n_launches = sum(n for _, n in sampled)
            k_ms = sum(timer.ms(handles[i]) for i, _ in sampled) / n_launches
            alg_bytes = B * 4 * (D + model.out_classes)
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9
            traffic, source = read_traffic('headline')
            two = (args.rg_batch, args.rg_sum) == (2, 2)
            kernel_name = headline_kernel_name(two, B)
            out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic if B == 65536 else None,
                               'traffic_source': ("profiles/pmc_traffic.json (builder's rocprofv3 --pmc passes of this "
                                                  "command, not measured in this run)") if B == 65536 and traffic else None,
                               'kernel': kernel_name, 'kernel_ms': k_ms, 'kernel_event_samples': n_launches,
                               'algorithmic_bytes_per_launch': alg_bytes}
            detail['roofline_notes'] = {
                'traffic_source': source if B == 65536 else None,
                'kernel_event_method': '{} runs of up to {} consecutive launches covering {} of the {} timed steps, one HIP '
                                       'event pair per run (includes the gaps between the launches of a run)'.format(
                                           len(sampled), run, n_launches, args.steps),
                'context': 'tools/ubench/read_bw.hip on the same GPU model: 6.4 TB/s for a linear 16-byte-load stream; the '
                           'slice kernel reads whole 100 KB blocks of 32 consecutive rows (64-byte pieces per request '
                           'lane quad); peak = the 8 TB/s specification'}
        threads = min(os.cpu_count() or 1, 32)
        if args.cpu_samples > 0 and world == 1:
            out['cpu_baseline'], threads = cpu_baseline(cpu_state, D, args.cpu_samples)
        if world == 1 and not args.no_secondary:
            try:
                sec = secondary(dev, timer, threads, xs, model)
                detail['secondary'] = sec
                # LAST in the line, compact: the driver's record keeps the tail of stdout
                out['configs'] = compact_configs(sec)
            except Exception as ex:   # the headline line must survive a failure in a secondary configuration
                out['secondary_error'] = '{}: {}'.format(type(ex).__name__, ex)[:300]
                import traceback
                traceback.print_exc(file=sys.stderr)
        out_path = write_detail(detail)
        if out_path:
            out['detail_file'] = out_path
            if 'configs' in out:   # (keep `configs` the last key)
                out['configs'] = out.pop('configs')
        print(json.dumps(out), flush=True)
    if world > 1:
        # Every rank is past its last collective; the line is out.  The ranks leave WITHOUT tearing down the communicator
        # and the HIP graphs that captured its collectives: that teardown (destroy_process_group, or the graphs'
        # destructors) ended 1-4 % of single-rank test runs on this stack with SIGABRT and no message -- across the ranks
        # of an N = 2 / 4 / 8 series that is one failed run in four for nothing.
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
