"""Batch-sharded density evaluation over the GPUs of one node (one process per GPU).

New functionality with no reference counterpart (the reference is single device): samples are
independent and the model is a few MB, so every rank holds a replica and evaluates a contiguous slice
of the batch; the only exchange is an all-reduce (sum) of ``{sum LL, count}`` in fp64 -- 16 bytes per
step over RCCL/xGMI, the pairs of up to 32 consecutive steps sharing one asynchronous collective --
after which every rank knows the mean log-likelihood of every step.  Per-sample LLs stay sharded.
Training shards the same way; the gradients meet in one flat all-reduce per step (``allreduce_gradients``).
"""
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a batch of n samples owned by ``rank`` (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# The shard of the current training batch: (samples on this rank, samples of the whole batch).  Set by the training loop
# (routines._batch) so that layers that need the size of the WHOLE batch on the host -- the synchronised 2-D batch norms:
# their kernels take the element count as an argument -- do not have to read a collective's result back.
_shard_sizes: Optional[Tuple[int, int]] = None


def set_shard_sizes(local: Optional[int], total: Optional[int] = None):
    global _shard_sizes
    _shard_sizes = None if local is None else (int(local), int(total))


def shard_sizes() -> Optional[Tuple[int, int]]:
    return _shard_sizes


def shard_batch(x: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """The slice of ``x`` this rank evaluates."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


SLOT = 17      # doubles of a {sum LL, count} slot: hip.LL_SPREAD partial sums, then the count
CAPTURE_ERROR_MODE = 'thread_local'   # torch.cuda.graph(..., capture_error_mode=): see GraphedEvaluationWindow
_CAPTURED_WORKS = []   # Work objects of collectives issued inside a HIP-graph capture (all_reduce_captured)


def quiesce_collectives(device=None, seconds: float = 0.3) -> None:
    """Call before capturing a HIP graph that contains a collective, in a process whose group runs on RCCL: drain the
    device, then give ProcessGroupNCCL's watchdog thread (poll period 100 ms) the time to retire the Work objects of the
    EAGER collectives issued so far.  A Work still on the watchdog's list is polled with hipEventQuery while the capture
    (which pulls the communicator's stream into the graph) is open; HIP answers that poll with "operation not permitted on an
    event last recorded in a capturing stream", invalidates the capture, and the watchdog aborts the process -- the rest of the
    2-4 % after all_reduce_captured's part of the fix.  No-op without a process group or on other backends."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        if 'nccl' not in str(dist.get_backend()).lower():
            return
    except Exception:
        return
    import time
    torch.cuda.synchronize(device)
    time.sleep(seconds)


def all_reduce_captured(t: torch.Tensor, group=None, keep: Optional[list] = None) -> None:
    """``dist.all_reduce(t, SUM)`` that is safe to issue while the current stream is being captured into a HIP graph: the
    collective's Work object -- and with it the HIP events ProcessGroupNCCL recorded inside the capture -- is kept alive
    (in ``keep``, by default for the life of the process) instead of being destroyed on return.  With the synchronous
    form the events are released while the capture is still open; the event cache (or, with the cache off, the HIP
    runtime's allocator) hands them to the next eager collective's Work, the NCCL watchdog thread polls that Work, HIP
    answers "operation not permitted on an event last recorded in a capturing stream", invalidates the capture in progress
    and the watchdog aborts the process (seen in 2-4 % of the constructions of a SECOND window in one process, ROCm 7.0 /
    RCCL 2.26 / PyTorch 2.10).  Outside a capture this is the plain synchronous call."""
    if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    work.wait()
    (_CAPTURED_WORKS if keep is None else keep).append(work)


def slot_mean(slots: torch.Tensor) -> torch.Tensor:
    """Mean log-likelihood of every ``[.., SLOT]`` (or plain ``[.., 2]`` = {sum, count}) slot."""
    if slots.shape[-1] == 2:
        return slots[..., 0] / slots[..., 1]
    return slots[..., :SLOT - 1].sum(dim=-1) / slots[..., SLOT - 1]


class ShardedLogLikelihood:
    """Mean log-likelihood of batches sharded over the ranks of ``group``.

    ``step(x_local)`` enqueues the local evaluation (for a RAT-SPN: the fused kernel with the fp64
    ``{sum, count}`` accumulation fused into its epilogue) and the asynchronous all-reduce;
    ``drain()`` waits for everything outstanding and returns the mean LL of every step since the last
    drain (identical on all ranks).

    :param model: a ``RatSpn``, ``DgcSpn`` or ``NormalizingFlow`` on this rank's device (ignored if
                  ``local_sum_fn`` is given).
    :param group: process group (None = single process, no collective).
    :param local_sum_fn: ``x -> float64[2] {sum LL, count}`` on x's device; lets the sharding logic be
                         exercised with any evaluator (the gloo/CPU tests plug in a CPU checker).
    """

    def __init__(self, model=None, group=None, local_sum_fn: Optional[Callable] = None,
                 static_inputs: bool = False, reduce_every: int = 32, static_params: bool = False):
        self.model = model
        self.group = group
        self.local_sum_fn = local_sum_fn
        # The {sum LL, count} pairs of up to ``reduce_every`` consecutive steps travel in ONE all-reduce (they sit
        # next to each other in the slot pool).  Fewer, larger collectives is what the xGMI mesh wants, and a
        # per-step collective kernel would also take a compute unit away from the model kernel, whose 512
        # work-groups need every one of the 2 x 256 slots to run in a single round.
        self.reduce_every = max(1, int(reduce_every))
        self._open = []        # steps whose slots are not in a collective yet: (slot tensor, pool, index)
        self._works = []
        # static_inputs: the caller steps over a fixed set of resident buffers (an evaluation ring): the fused
        # call is bound once per buffer (model.fused_plan) and each step is one C call
        self.static_inputs = static_inputs
        # static_params: the model is frozen for the evaluator's lifetime (no write to a parameter, through .data
        # included): the bound calls skip the device-side fingerprint of the cached parameter tables
        self.static_params = static_params
        self._plans = {}
        self._pending: List[Tuple[torch.Tensor, Optional[object]]] = []
        self.last_ll: Optional[torch.Tensor] = None
        self._pool: Optional[torch.Tensor] = None  # zeroed {sum, count} slots, one fill per 256 steps
        self._pool_next = 0

    def _acc_slot(self, device) -> torch.Tensor:
        # a slot = SLOT doubles: sixteen partial sums of the log-likelihoods, then the count (DPK_FLAG_LL_SUM_SPREAD: the 256
        # work-groups of a fused launch finish together, and 256 fp64 atomics on ONE address are 1.6 us in series behind a
        # 12 us shard launch); `slot_mean` adds the partial sums
        if self._pool is None or self._pool_next >= self._pool.shape[0] or self._pool.device != device:
            self._pool = torch.zeros(256, SLOT, dtype=torch.float64, device=device)
            self._pool_next = 0
        slot = self._pool[self._pool_next]
        self._slot_ref = (self._pool, self._pool_next)
        self._pool_next += 1
        return slot

    def _is_ratspn(self) -> bool:
        from deeprob.spn.models.ratspn import RatSpn
        return isinstance(self.model, RatSpn)

    def _local(self, x: torch.Tensor, kernel_events=None, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.local_sum_fn is not None:
            return self.local_sum_fn(x)
        if acc is None:
            acc = self._acc_slot(x.device)
        if kernel_events is not None:
            from deeprob.hip import load_library, check
            # (start, stop): raw hipEvent_t handles or torch.cuda.Event objects; either may be None (a run of
            # launches bracketed by the start of its first and the stop of its last)
            h0, h1 = (None if e is None else (e if isinstance(e, int) else e.cuda_event) for e in kernel_events)
            check(load_library().dpk_profile_next_kernel(h0, h1), 'dpk_profile_next_kernel')
        from deeprob.hip import ops
        fused = getattr(self.model, '_forward_fused', None)
        ll = None
        if self._is_ratspn():
            # RAT-SPN: the fused kernel adds the tile's fp64 {sum, count} itself (no second pass over the LLs)
            if self.static_inputs and not torch.is_grad_enabled():
                key = (x.data_ptr(), x.shape[0])
                plan = self._plans.get(key)
                if plan is None or (plan is not False and not plan.valid()):
                    plan = self.model.fused_plan(x, static_params=self.static_params) or False
                    self._plans[key] = plan
                if plan is not False:
                    ll = plan.run(acc)
            if ll is None:
                ll = fused(x, acc)
        if ll is None:
            # DGC-SPN, normalizing flows, RAT-SPN shapes outside the fused kernel: the model's own forward
            # (HIP kernels), then one reduction kernel over the B (x classes) log-likelihoods
            ll = self.model(x)
            ops.ll_accumulate(ll, acc)
        self.last_ll = ll
        return acc

    def step(self, x_local: torch.Tensor, kernel_events=None):
        self._slot_ref = None
        acc = self._local(x_local, kernel_events)
        self._pending.append((acc, None))
        if self.group is not None and dist.get_world_size(self.group) > 1:
            self._open.append((acc, self._slot_ref))
            if len(self._open) >= self.reduce_every:
                self._reduce_open()

    def _reduce_open(self):
        """One asynchronous all-reduce over the slots of the open steps (contiguous runs of the slot pool are
        reduced in place; anything else -- a custom local_sum_fn -- is stacked first)."""
        if not self._open:
            return
        runs, cur = [], None
        for acc, ref in self._open:
            if ref is not None and cur is not None and cur[0] is ref[0] and cur[2] == ref[1]:
                cur[2] += 1
            else:
                cur = [ref[0], ref[1], ref[1] + 1] if ref is not None else None
                runs.append(cur if cur is not None else acc)
        for r in runs:
            t = r[0][r[1]:r[2]] if isinstance(r, list) else r
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._open = []

    def drain(self) -> List[float]:
        if not self._pending:
            return []
        if self.group is not None and dist.get_world_size(self.group) > 1:
            self._reduce_open()
        for work in self._works:
            work.wait()
        self._works = []
        accs = torch.stack([a.view(-1) for a, _ in self._pending])
        self._pending = []
        return slot_mean(accs).cpu().tolist()  # one device->host copy for the whole window


def workspace_replica(model: torch.nn.Module) -> torch.nn.Module:
    """A second handle on ``model`` for a second stream: the SAME parameter and buffer tensors (a later write to a parameter
    is seen by both), its own workspaces (cached tables, tickets of the in-launch table check, NaN hint): the launches of one
    workspace must follow one another on one stream, so every concurrent chain of evaluations gets a replica."""
    import copy
    rep = copy.deepcopy(model)

    def share(dst, src):
        for name, p in src._parameters.items():
            dst._parameters[name] = p
        for name, b in src._buffers.items():
            dst._buffers[name] = b
        if hasattr(src, 'distribution'):          # (torch.distributions objects built over the parameters)
            dst.distribution = src.distribution
        for name, child in src._modules.items():
            share(dst._modules[name], child)

    share(rep, model)
    return rep


class GraphedEvaluationWindow:
    """A window of sharded evaluation steps -- the local evaluations of ``xs`` (resident inputs) AND the one all-reduce of
    their ``{sum LL, count}`` pairs -- captured once as a HIP graph and replayed: ``replay()`` returns the mean
    log-likelihood of every step of the window (identical on all ranks).  The RCCL collective is part of the graph
    (ProcessGroupNCCL records it on the capturing stream), so a replay costs one graph launch whatever the number of steps.
    Built from a ``ShardedLogLikelihood`` (its model, group and parameter mode); the evaluator itself stays usable.

    ``always_reduce``: run the collective for a world of one too (exercises the captured RCCL path on a one-GPU box).

    ``chains`` (round 6): the steps of the window run on this many parallel chains inside the graph -- one fork at its head,
    one join in front of the all-reduce, step i on chain i mod chains, a ``workspace_replica`` of the model per extra chain.
    A launch of the fused kernel holds every compute unit with one work-group; in one chain launch k + 1 starts when launch
    k has completely finished, so its prologue and the predecessor's tail (HBM idle in both) are in series -- at the
    strong-scaling shard sizes (one to four blocks per compute unit) that is most of a step.  Measured, default mode
    (tools/bench_two_streams_graph.py): 12.5 -> 11.1 -> 10.2 us per step at 8 192 samples with 1 / 2 / 3 chains, 16.7 ->
    14.3 at 16 384, 25.8 -> 22.9 at 32 768, 42.9 -> 40.8 -> 39.5 at 65 536; identical results."""

    def __init__(self, evaluator: 'ShardedLogLikelihood', xs: List[torch.Tensor], always_reduce: bool = False, chains: int = 1):
        if evaluator.local_sum_fn is not None:
            raise ValueError("GraphedEvaluationWindow captures the HIP evaluation path, not a custom local_sum_fn")
        self.evaluator, self.xs = evaluator, list(xs)
        dev = self.xs[0].device
        world = dist.get_world_size(evaluator.group) if evaluator.group is not None else 1
        reduce = evaluator.group is not None and (world > 1 or always_reduce)
        chains = max(1, min(int(chains), len(self.xs)))
        if chains == 1:
            self.lanes = [evaluator]
        else:
            # Concurrent launches must not wait on one another: the fused RAT-SPN kernels' in-launch table check has every
            # work-group wait (up to a 1 s time-out) for the verdict of the launch's first work-groups -- sound while a launch
            # has the chip to itself, a circular wait once the work-groups of three launches compete for the compute units
            # (observed: one replay in a few taking the full second, results still right through the time-out's exact
            # route).  With chains the parameters are therefore checked ONCE PER REPLAY at the head of every chain -- a
            # one-sample forward in the default, verifying mode on that chain's workspace (14 work-groups: always
            # co-resident; rebuilds the tables in place if a parameter was written) -- and the steps run on the verified
            # tables (static_params).  A write to a parameter between two replays is seen; none happens during a replay.
            models = [evaluator.model] + [workspace_replica(evaluator.model) for _ in range(chains - 1)]
            self.lanes = [ShardedLogLikelihood(m, static_inputs=True, static_params=True) for m in models]
        side = torch.cuda.Stream(device=dev)
        branches = [side] + [torch.cuda.Stream(device=dev) for _ in range(chains - 1)]
        # (inputs / parameters still being written on the caller's stream must be complete before the warm pass reads them:
        # the tables it builds and the verdicts it caches would otherwise come from incomplete data)
        side.wait_stream(torch.cuda.current_stream(dev))
        self.graph = torch.cuda.CUDAGraph()
        self._works = []     # the captured collective's Work: alive as long as the graph (all_reduce_captured)
        with torch.no_grad(), torch.cuda.stream(side):
            warm = torch.zeros(len(self.xs), SLOT, dtype=torch.float64, device=dev)
            probe = self.xs[0][:1]
            if chains > 1:
                for lane in self.lanes:
                    lane.model(probe)
            for i, x in enumerate(self.xs):          # eager pass: plans bound, tables built, RCCL communicator up
                self.lanes[i % chains]._local(x, acc=warm[i])
            if reduce:
                dist.all_reduce(warm, op=dist.ReduceOp.SUM, group=evaluator.group)
            torch.cuda.synchronize(dev)
            if reduce:
                quiesce_collectives(dev)
            # (thread-local capture mode: ProcessGroupNCCL's watchdog thread polls the events of earlier collectives with
            # hipEventQuery; under the default 'global' mode such a call from ANOTHER thread while this one captures is an
            # error that the watchdog turns into abort() -- seen as a crash of this constructor in one full test run of
            # four.  Collectives issued inside the capture are not handed to the watchdog.)
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=CAPTURE_ERROR_MODE):
                self.pool = torch.zeros(len(self.xs), SLOT, dtype=torch.float64, device=dev)
                for b in branches[1:]:
                    b.wait_stream(side)              # fork (the zeroed slots are complete on every chain)
                if chains > 1:
                    for lane, b in zip(self.lanes, branches):
                        with torch.cuda.stream(b):
                            lane.model(probe)        # the chain's table check for this replay
                for i, x in enumerate(self.xs):
                    with torch.cuda.stream(branches[i % chains]):
                        self.lanes[i % chains]._local(x, acc=self.pool[i])
                for b in branches[1:]:
                    side.wait_stream(b)              # join: the collective sees every chain's sums
                if reduce:
                    all_reduce_captured(self.pool, group=evaluator.group, keep=self._works)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

    def replay(self) -> List[float]:
        self.graph.replay()
        return slot_mean(self.pool).cpu().tolist()

    def close(self) -> None:
        """Release the HIP graph and the captured collective's Work.  A graph that captured a collective refers to its
        communicator: release such windows before ``dist.destroy_process_group()``."""
        torch.cuda.synchronize(self.pool.device)
        self.graph = None
        self._works = []


def bn_gather_moments(moments: torch.Tensor, group=None) -> torch.Tensor:
    """Sync-BN forward exchange: every rank contributes its ``{count, mean[D], M2[D]}`` vector, all ranks receive the
    ``[world, 2D+1]`` table (an all-gather, done as an all-reduce of a table that is zero outside the own row: exact,
    and available for device tensors on every backend)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    table = torch.zeros((world, moments.numel()), dtype=moments.dtype, device=moments.device)
    table[rank].copy_(moments)
    all_reduce_captured(table, group=group)
    return table


def bn_reduce_sums(sums: torch.Tensor, n_local: int, n_total: int, group=None) -> torch.Tensor:
    """Sync-BN backward exchange.  ``sums`` = this rank's column sums of the gradient of ITS loss (the mean over its
    ``n_local`` rows).  The single-process loss is the mean over all ``n_total`` rows, so the whole-batch sums are
    ``sum_r (n_r / n_total) sums_r``; they are handed back rescaled by ``n_total / n_local`` -- in the units of this
    rank's loss again, so that the later sample-weighted gradient average (``allreduce_gradients``) is the
    single-process gradient.  One all-reduce of 2D+1 floats."""
    t = sums * (float(n_local) / float(n_total))
    all_reduce_captured(t, group=group)
    if n_local > 0:
        t *= float(n_total) / float(n_local)
    return t


def synchronize_batchnorm(model: torch.nn.Module, group=None, enabled: bool = True):
    """Make every train-mode batch normalisation of ``model`` use the statistics of the whole (sharded) batch: sets
    ``layer.sync_group`` (None switches it off) on ``BatchNormLayer1d`` and, since round 4, on ``BatchNormLayer2d`` and the
    ``nn.BatchNorm2d`` inside the convolutional conditioners of a ``RealNVP2d``.  ``train_model`` does this when it shards
    batches.

    2-D layers (deeprob/hip/ops_flows2d_train.py): the per-channel fp64 sums {sum x, sum x^2} of every rank meet in one
    all-reduce per layer before the fold kernel (which then normalises with -- and updates the running statistics from --
    the statistics of the whole batch, identically on every rank), and in the backward the gradients of the local loss
    with respect to (mean, var) are all-reduced weighted by n_r / N before they flow into ``dx``.  With the
    sample-weighted gradient all-reduce of ``allreduce_gradients`` the step then equals the single-process step on the
    unsharded batch (tests/test_parallel_gpu.py)."""
    from deeprob.flows.utils import BatchNormLayer1d, BatchNormLayer2d
    for m in model.modules():
        if isinstance(m, (BatchNormLayer1d, BatchNormLayer2d, torch.nn.BatchNorm2d)):
            m.sync_group = (group if group is not None else dist.group.WORLD) if enabled else None


class local_batchnorm:
    """Context manager: with ``active`` the synchronised ``BatchNormLayer1d`` layers of ``model`` use their local
    batch only (no collective) inside the block -- for a batch that every rank evaluates in full."""

    def __init__(self, model: torch.nn.Module, active: bool = True):
        self.model, self.active, self._saved = model, active, []

    def __enter__(self):
        if self.active:
            for m in self.model.modules():
                if getattr(m, 'sync_group', None) is not None:
                    self._saved.append((m, m.sync_group))
                    m.sync_group = None
        return self

    def __exit__(self, *exc):
        for m, g in self._saved:
            m.sync_group = g
        self._saved = []
        return False


def broadcast_model(model: torch.nn.Module, group=None, src: int = 0):
    """Make every rank's replica the one of rank ``src``: parameters AND buffers (region-graph masks, BatchNorm running
    statistics) are broadcast in ONE flat collective per dtype.  Data-parallel training assumes identical replicas;
    without this, ranks that were initialised from different RNG states (``random_state=None`` region graphs,
    ``randn`` / Dirichlet initialisers) would average the gradients of different models."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    for t in list(model.parameters()) + list(model.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, tensors in by_dtype.items():
            wire = torch.uint8 if dtype == torch.bool else dtype   # (bool has no collective on every backend)
            flat = torch.cat([t.detach().reshape(-1).to(wire) for t in tensors])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape).to(dtype))
                off += n


def broadcast_seed(device, group=None, src: int = 0) -> int:
    """A random 63-bit seed drawn on rank ``src`` and agreed on by every rank (shuffles of sharded loaders)."""
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = seed.to(device)
        dist.broadcast(t, src=src, group=group)
        seed = t.cpu()
    return int(seed.item())


def allreduce_gradients(model: torch.nn.Module, group=None, weight: Optional[float] = None, force: bool = False):
    """Data-parallel gradient exchange in ONE collective: every ``.grad`` is packed into a flat fp32 bucket,
    all-reduced (sum) over RCCL/xGMI and unpacked.  The models of the path hold 50 KB .. 6 MB of parameters
    (SURVEY 8e), so a single bucket is both the latency- and the bandwidth-optimal choice on the 7-link xGMI mesh.

    ``weight`` = number of samples behind this rank's (mean-reduced) loss: the result is then the gradient of the
    mean over the GLOBAL batch, sum_r n_r g_r / sum_r n_r, exactly what a single process would compute on the
    unsharded batch (a rank with an empty shard passes 0).  ``weight=None``: plain average over the ranks.
    Parameters without a gradient on this rank contribute zeros.  No host read anywhere: the exchange can be captured in
    a HIP graph with the step around it (``deeprob.hip.graphs.GraphedTrainStep(grad_exchange=...)``)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1 and not force:        # (force: run the collective for a world of one -- one-GPU rehearsals of the path)
        return
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    w = 1.0 if weight is None else float(weight)
    parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params]
    flat = torch.cat(parts + [torch.ones(1, dtype=torch.float32, device=parts[0].device)])
    flat *= w
    all_reduce_captured(flat, group=group)      # (inside GraphedTrainStep's capture the Work must outlive the capture)
    flat /= flat[-1].clamp_min(1e-30)          # total weight (the world size when unweighted)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
