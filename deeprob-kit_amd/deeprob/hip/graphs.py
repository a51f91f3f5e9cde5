"""One optimisation step as a HIP graph.

A training step of the models on this path is a run of 40-130 short kernels (the GEMMs of a RealNVP coupling at a
batch of 512 take 10-20 us each); launched one by one the host, not the GPU, bounds the step.  ``GraphedTrainStep``
captures forward, loss, backward, optimiser update and ``apply_constraints`` of one batch shape into a
``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) after three eager warm-up steps and replays it for every further batch
of that shape; other shapes (a ragged last batch) run eagerly.  The C-ABI operators enqueue on the current stream,
allocate through torch's caching allocator and never synchronise, so they capture as they are.

Not capturable (raises): training-mode dropout, whose seeds are drawn on the host per step, and optimisers whose state
update reads host scalars (pass ``capturable=True`` to Adam-family optimisers).
"""
from typing import Callable, Optional

import torch


def _has_dropout(model: torch.nn.Module) -> bool:
    return any(getattr(m, 'dropout', None) for m in model.modules())


class GraphedTrainStep:
    """``grad_exchange(n_local)``: called between backward and the optimiser update, inside the captured region -- the
    sharded step's gradient all-reduce (``deeprob.parallel.allreduce_gradients``: device operations and one RCCL
    collective, which ProcessGroupNCCL records on the capturing stream).  Every rank captures its own shard shape; ranks
    replay / run eagerly in lockstep because both forms issue the same collectives."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, warmup: int = 3,
                 grad_exchange: Optional[Callable[[int], None]] = None):
        if _has_dropout(model):
            raise NotImplementedError("hip_graph: training-mode dropout draws its seeds on the host every step")
        for group in optimizer.param_groups:
            if 'capturable' in group and not group['capturable']:
                raise ValueError("hip_graph: build the optimizer with capturable=True")
        self.model, self.optimizer, self.warmup = model, optimizer, warmup
        self.grad_exchange = grad_exchange
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_in: Optional[torch.Tensor] = None
        self.static_loss: Optional[torch.Tensor] = None
        self.static_key = None
        self.seen = 0

    def _eager(self, inputs: torch.Tensor) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=False)
        loss = self.model.loss(self.model(inputs))
        loss.backward()
        if self.grad_exchange is not None:
            self.grad_exchange(int(inputs.shape[0]))
        self.optimizer.step()
        self.model.apply_constraints()
        return loss

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        """Train on one batch; returns the loss tensor (valid until the next call)."""
        # What a capture bakes in besides the shard's shape: the host integers of a sharded step -- the whole-batch element
        # count and the n_r / N weight of the synchronised 2-D batch norms (deeprob.parallel.shard_sizes) are kernel
        # scalars.  On a ragged last batch (511 rows on 2 ranks: 256 + 255) one rank's shard still has the captured shape:
        # replaying there while its peer runs eagerly with N = 511 would leave the replicas silently diverged.  The step is
        # therefore keyed on (local shape, shard sizes): anything else runs eagerly, on every rank alike.
        from deeprob import parallel
        key = parallel.shard_sizes()
        if self.static_in is None:
            self.static_in = torch.empty_like(inputs)
            self.static_key = key
        if inputs.shape != self.static_in.shape or key != self.static_key:
            return self._eager(inputs)                       # e.g. the last, shorter batch of an epoch
        self.static_in.copy_(inputs)
        if self.graph is not None:
            self.graph.replay()
            return self.static_loss
        if self.seen < self.warmup:                          # allocator pools and lazy state settle before capture
            self.seen += 1
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                loss = self._eager(self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            return loss
        self.graph = torch.cuda.CUDAGraph()
        if self.grad_exchange is not None:
            from deeprob.parallel import quiesce_collectives
            quiesce_collectives()     # (eager collectives of the warm-up steps: retired before the capture opens)
        # (thread-local capture mode: with a process group up, ProcessGroupNCCL's watchdog thread may call hipEventQuery
        # while this thread captures -- an error under the default 'global' mode, which the watchdog turns into abort())
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.static_loss = self._eager(self.static_in)
        # capture records the step without running it: replay once so that this batch is trained on as well
        self.graph.replay()
        return self.static_loss
