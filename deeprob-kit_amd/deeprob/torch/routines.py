"""Train / test loops behind the reference interface (deeprob/torch/routines.py:21-478), generative and
discriminative settings.

Differences that matter on an MI355X node:
* the running loss stays on the device: no ``loss.item()`` host synchronisation per batch (reference :166);
* when ``torch.distributed`` is initialised every rank trains on its own shard of each batch
  (``DistributedSampler``-style strided split of the loader's batches) and the gradients meet in ONE flat
  all-reduce per step (``deeprob.parallel.allreduce_gradients``; 50 KB .. 6 MB for the models of the path);
  validation / test log-likelihoods are reduced the same way, the early-stopping checkpoint is written by rank 0.
The history dict, the ``(mean_ll, 2 std / sqrt(n))`` test result and the ValueErrors are the reference's.
"""
import os
import time
from typing import Union, Optional, Tuple, Dict

import numpy as np
import torch
from torch import optim
from torch.utils import data
import torch.distributed as dist

from deeprob.torch.base import ProbabilisticModel
from deeprob.torch.utils import get_optimizer_class
from deeprob.torch.callbacks import EarlyStopping
from deeprob.torch.metrics import RunningAverageMetric
from deeprob.parallel import (allreduce_gradients, shard_batch, broadcast_model, broadcast_seed,
                              synchronize_batchnorm, local_batchnorm)


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _train_mode(model, train_base: bool):
    from deeprob.flows.models.base import NormalizingFlow
    if isinstance(model, NormalizingFlow):
        model.train(base_mode=train_base)
    else:
        model.train()


def train_model(
    model: ProbabilisticModel,
    data_train: Union[np.ndarray, data.Dataset],
    data_valid: Union[np.ndarray, data.Dataset],
    setting: str = 'generative',
    lr: float = 1e-3,
    batch_size: int = 100,
    epochs: int = 1000,
    optimizer: str = 'adam',
    optimizer_kwargs: Optional[dict] = None,
    patience: int = 20,
    checkpoint: Union[os.PathLike, str] = 'checkpoint.pt',
    train_base: bool = True,
    drop_last: bool = True,
    num_workers: int = 0,
    device: Optional[torch.device] = None,
    verbose: bool = True,
    hip_graph: bool = False
) -> Dict[str, list]:
    """Reference signature (:21-37) + ``hip_graph``.  ``batch_size`` is the GLOBAL batch: with N ranks each one sees
    batch_size/N samples per step.  ``hip_graph`` (generative setting, no dropout; sharded runs capture their gradient
    all-reduce and synchronised batch-norm exchanges with the step): the optimisation
    step of full batches is captured once as a HIP graph and replayed (``deeprob.hip.graphs.GraphedTrainStep``);
    Adam-family optimisers are built with ``capturable=True``.
    :raises ValueError: for an unknown setting or non-positive epochs."""
    if setting not in ('generative', 'discriminative'):
        raise ValueError("Unknown train setting called {}".format(setting))
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or device.type != 'cuda':
        raise ValueError("deeprob on MI355X trains on a HIP device (there is no CPU path)")
    # sharded training slices every batch over the ranks: all ranks must draw the SAME shuffle
    shuffle_gen = torch.Generator().manual_seed(broadcast_seed(device)) if _world()[1] > 1 else None
    train_loader = data.DataLoader(data_train, batch_size, shuffle=True, drop_last=drop_last, num_workers=num_workers,
                                   generator=shuffle_gen)
    valid_loader = data.DataLoader(data_valid, batch_size, shuffle=False, drop_last=False, num_workers=num_workers)
    model.to(device)
    optimizer_kwargs = dict(optimizer_kwargs or {})
    import inspect
    if optimizer in ('adam', 'sgd') and not (optimizer_kwargs.get('foreach') or optimizer_kwargs.get('differentiable')):
        # one launch per optimiser step instead of torch's seven multi-tensor launches (62 of the 325 us of GPU time of a
        # RAT-SPN (8,8) step at B = 512); the same update rule -- pass optimizer_kwargs={'fused': False} for torch's default.
        # Only the optimisers whose fused kernels run on a HIP device: Adagrad's constructor accepts `fused` too, but its
        # fused kernels are CPU-only (RuntimeError at the first step) and RMSprop has none; `foreach` / `differentiable`
        # exclude `fused`.
        optimizer_kwargs.setdefault('fused', True)
    if hip_graph:
        if setting != 'generative':
            raise ValueError("hip_graph covers the generative setting")
        if 'capturable' in inspect.signature(get_optimizer_class(optimizer).__init__).parameters:
            optimizer_kwargs.setdefault('capturable', True)
        elif optimizer != 'sgd':   # (plain SGD keeps no step counter on the host: it captures as it is)
            raise ValueError("hip_graph needs an optimizer that can be captured (capturable=True); "
                             "torch.optim's '{}' cannot".format(optimizer))
    opt = build_optimizer(optimizer, [p for p in model.parameters() if p.requires_grad], lr, optimizer_kwargs)
    early_stopping = EarlyStopping(model, patience=patience, filepath=checkpoint)
    if hip_graph:
        return train_generative(model, train_loader, valid_loader, opt, device, early_stopping, epochs, train_base,
                                verbose, hip_graph=True)
    fit = train_generative if setting == 'generative' else train_discriminative
    return fit(model, train_loader, valid_loader, opt, device, early_stopping, epochs, train_base, verbose)


def build_optimizer(name: str, params: list, lr: float, kwargs: Optional[dict] = None) -> optim.Optimizer:
    """The optimiser ``train_model`` steps.  'adam' with torch's fused kernels asked for (the default above) and nothing
    beyond the plain update rule (no amsgrad, no decoupled decay) becomes ``deeprob.hip.optim.FusedAdam``: the same
    update with every tensor of the model dealt out over work-groups of 2048 elements in one launch (torch's fused Adam
    deals in chunks of 65 536, i.e. 3 work-groups and 30 us for a RAT-SPN (8,8)); everything else is torch.optim's class
    of that name (reference torch/utils.py:32-49)."""
    kwargs = dict(kwargs or {})
    if name == 'adam' and kwargs.get('fused') and not kwargs.get('amsgrad') and not kwargs.get('decoupled_weight_decay'):
        from deeprob.hip.optim import FusedAdam
        plain = set(kwargs) <= {'fused', 'capturable', 'betas', 'eps', 'weight_decay', 'maximize', 'amsgrad',
                                'decoupled_weight_decay'}
        if plain and FusedAdam.supports(params):
            kwargs.pop('amsgrad', None), kwargs.pop('decoupled_weight_decay', None)
            return FusedAdam(params, lr=lr, **kwargs)
    return get_optimizer_class(name)(params, lr=lr, **kwargs)


def train_generative(
    model: ProbabilisticModel,
    train_loader: data.DataLoader,
    valid_loader: data.DataLoader,
    optimizer: optim.Optimizer,
    device: torch.device,
    early_stopping: EarlyStopping,
    epochs: int = 1000,
    train_base: bool = True,
    verbose: bool = True,
    hip_graph: bool = False
) -> Dict[str, list]:
    """Reference :98-210.  Returns ``{'train': [...], 'valid': [...]}`` (average loss per epoch)."""
    hist = _fit(model, train_loader, valid_loader, optimizer, device, early_stopping, epochs, train_base, verbose,
                supervised=False, hip_graph=hip_graph)
    return {'train': hist['train']['loss'], 'valid': hist['valid']['loss']}


def train_discriminative(
    model: ProbabilisticModel,
    train_loader: data.DataLoader,
    valid_loader: data.DataLoader,
    optimizer: optim.Optimizer,
    device: torch.device,
    early_stopping: EarlyStopping,
    epochs: int = 1000,
    train_base: bool = True,
    verbose: bool = True
) -> Dict[str, Dict[str, list]]:
    """Reference :213-346: loaders yield ``(inputs, targets)``; returns ``{'train': {'loss', 'accuracy'}, 'valid':
    {...}}`` per epoch."""
    return _fit(model, train_loader, valid_loader, optimizer, device, early_stopping, epochs, train_base, verbose,
                supervised=True)


def _batch(item, device, rank, world, supervised):
    """This rank's shard of a loader item, moved to the device."""
    from deeprob.parallel import set_shard_sizes, shard_bounds
    if world > 1:
        n = (item[0] if supervised else item).shape[0]
        lo, hi = shard_bounds(n, rank, world)
        set_shard_sizes(hi - lo, n)
    else:
        set_shard_sizes(None)      # (a replicated batch: stale sizes of the previous sharded one must not describe it)
    if supervised:
        inputs, targets = item
        return (shard_batch(inputs.to(device, non_blocking=True), rank, world),
                shard_batch(targets.to(device, non_blocking=True), rank, world))
    return shard_batch(item.to(device, non_blocking=True), rank, world), None


def _fit(model, train_loader, valid_loader, optimizer, device, early_stopping, epochs, train_base, verbose, supervised,
         hip_graph=False):
    if epochs <= 0:
        raise ValueError("epochs must be at least 1")
    rank, world = _world()
    if world > 1:
        # identical replicas on every rank (parameters and buffers of rank 0), whole-batch statistics for every
        # BatchNormLayer1d (2-D batch norms keep per-rank statistics: synchronize_batchnorm warns about them).
        # The caller's loaders must yield the same batches on every rank (train_model seeds its shuffle accordingly).
        broadcast_model(model)
        synchronize_batchnorm(model)
    graphed = None
    if hip_graph:
        if supervised:
            raise ValueError("hip_graph covers the generative setting")
        from deeprob.hip.graphs import GraphedTrainStep
        # sharded: the gradient all-reduce (and the synchronised batch norms' exchanges inside the model) are captured
        # with the step -- every rank captures its own shard shape, graph replays and eager steps issue the same collectives
        graphed = GraphedTrainStep(model, optimizer,
                                   grad_exchange=(lambda n: allreduce_gradients(model, weight=n)) if world > 1 else None)
    history = {'train': {'loss': [], 'accuracy': []}, 'valid': {'loss': [], 'accuracy': []}}
    meters = {k: RunningAverageMetric() for k in ('train_loss', 'train_hits', 'valid_loss', 'valid_hits')}
    try:
        return _fit_epochs(model, train_loader, valid_loader, optimizer, device, early_stopping, epochs, train_base, verbose,
                           supervised, graphed, history, meters, rank, world)
    finally:
        # nothing of the sharded run may describe a later call: a stale (local, total) pair would be taken for the sizes of
        # a train-mode batch that happens to have the same local size, and a sync_group left on the model would make a
        # later single-process call wait in a collective
        from deeprob.parallel import set_shard_sizes
        set_shard_sizes(None)
        if world > 1:
            synchronize_batchnorm(model, enabled=False)


def _fit_epochs(model, train_loader, valid_loader, optimizer, device, early_stopping, epochs, train_base, verbose, supervised,
                graphed, history, meters, rank, world):
    for epoch in range(1, epochs + 1):
        for m in meters.values():
            m.reset()
        t0 = time.perf_counter()
        _train_mode(model, train_base)
        for item in train_loader:
            # A batch with fewer rows than ranks would leave a rank without a shard.  With synchronised BatchNorm that
            # rank would skip the model -- and the collectives inside it -- while its peers wait in them.  Such a batch
            # (a short last batch with drop_last=False, or batch_size < world) is therefore REPLICATED: every rank
            # evaluates all of it with the statistics exchange switched off, the gradients and running statistics are
            # identical everywhere, and rank 0 alone books its loss.  No rank is ever empty.
            n_global = (item[0] if supervised else item).shape[0]
            replicated = world > 1 and n_global < world
            if replicated:
                inputs, targets = _batch(item, device, 0, 1, supervised)
            else:
                inputs, targets = _batch(item, device, rank, world, supervised)
            n_local = inputs.shape[0]
            if graphed is not None and n_local > 0 and not replicated:
                meters['train_loss'](graphed(inputs), num_samples=n_local)
                continue
            optimizer.zero_grad()
            if n_local > 0:
                with local_batchnorm(model, replicated):
                    outputs = model(inputs)
                    loss = model.loss(outputs, y=targets) if supervised else model.loss(outputs)
                    loss.backward()
            if world > 1:
                # every rank joins the collective (equal weights on a replicated batch: the average of identical values)
                allreduce_gradients(model, weight=n_local)
            optimizer.step()
            model.apply_constraints()
            if n_local > 0 and (not replicated or rank == 0):
                meters['train_loss'](loss, num_samples=n_local)           # stays on the device
                if supervised:
                    with torch.no_grad():
                        hits = torch.eq(torch.argmax(outputs, dim=1), targets).float().mean()
                    meters['train_hits'](hits, num_samples=n_local)
        model.eval()
        with torch.no_grad():
            for item in valid_loader:
                inputs, targets = _batch(item, device, rank, world, supervised)
                if inputs.shape[0] == 0:
                    continue
                outputs = model(inputs)
                meters['valid_loss'](model.loss(outputs, y=targets) if supervised else model.loss(outputs),
                                     num_samples=inputs.shape[0])
                if supervised:
                    hits = torch.eq(torch.argmax(outputs, dim=1), targets).float().mean()
                    meters['valid_hits'](hits, num_samples=inputs.shape[0])
        names = ('train_loss', 'valid_loss') + (('train_hits', 'valid_hits') if supervised else ())
        avg = dict(zip(names, _epoch_averages([meters[k] for k in names], device, world)))
        elapsed = int(time.perf_counter() - t0)
        if verbose and rank == 0:
            line = "Epoch {}/{} - train_loss: {:.4f}, valid_loss: {:.4f}".format(epoch, epochs, avg['train_loss'],
                                                                                avg['valid_loss'])
            if supervised:
                line += ", train_acc: {:.1f}%, valid_acc: {:.1f}%".format(avg['train_hits'] * 100,
                                                                         avg['valid_hits'] * 100)
            print(line + " [{}s]".format(elapsed if elapsed > 0 else '<1'))
        history['train']['loss'].append(avg['train_loss'])
        history['valid']['loss'].append(avg['valid_loss'])
        if supervised:
            history['train']['accuracy'].append(avg['train_hits'])
            history['valid']['accuracy'].append(avg['valid_hits'])
        early_stopping(avg['valid_loss'], epoch, save=(rank == 0))
        if early_stopping.should_stop:
            if verbose and rank == 0:
                print("Early Stopping... {}".format(early_stopping))
            break
    if world > 1:
        dist.barrier()   # rank 0 finished writing the checkpoint
    model.load_state_dict(early_stopping.get_best_state(map_location=device))
    return history


def _epoch_averages(metrics, device, world):
    """One device->host copy per epoch (and one all-reduce of 2 doubles per metric when sharded)."""
    out = []
    for m in metrics:
        total = m._total if torch.is_tensor(m._total) else torch.tensor(float(m._total), dtype=torch.float64, device=device)
        pair = torch.stack([total.to(torch.float64), torch.tensor(float(m._count), dtype=torch.float64, device=device)])
        if world > 1:
            dist.all_reduce(pair)
        s, n = pair.tolist()
        out.append(s / n)
    return out


def test_model(
    model: ProbabilisticModel,
    data_test: Union[np.ndarray, data.Dataset],
    setting: str = 'generative',
    batch_size: int = 100,
    num_workers: int = 0,
    device: Optional[torch.device] = None,
    verbose: bool = True
) -> Tuple[float, float]:
    """Reference :349-388 (generative): mean log-likelihood and two standard errors."""
    if setting not in ('generative', 'discriminative'):
        raise ValueError("Unknown test setting called {}".format(setting))
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or device.type != 'cuda':
        raise ValueError("deeprob on MI355X evaluates on a HIP device (there is no CPU path)")
    loader = data.DataLoader(data_test, batch_size, shuffle=False, drop_last=False, num_workers=num_workers)
    model.to(device)
    if setting == 'discriminative':
        return test_discriminative(model, loader, device, verbose)
    return test_generative(model, loader, device, verbose)


def test_generative(model: ProbabilisticModel, test_loader: data.DataLoader, device: torch.device,
                    verbose: bool = True) -> Tuple[float, float]:
    """Reference :391-426: ``(mean LL, 2 std / sqrt(n))`` with the population std, computed from fp64 sums of LL and
    LL^2 kept on the device (the reference copies every LL to the host)."""
    rank, world = _world()
    model.eval()
    acc = torch.zeros(3, dtype=torch.float64, device=device)   # n, sum, sum of squares
    with torch.no_grad():
        for inputs in test_loader:
            inputs = shard_batch(inputs.to(device, non_blocking=True), rank, world)
            if inputs.shape[0] == 0:
                continue
            ll = model(inputs).to(torch.float64).reshape(-1)
            acc += torch.stack([torch.tensor(float(ll.numel()), dtype=torch.float64, device=device), ll.sum(),
                                (ll * ll).sum()])
    if world > 1:
        dist.all_reduce(acc)
    n, s, q = acc.tolist()
    mean = s / n
    var = max(q / n - mean * mean, 0.0)
    return mean, 2.0 * float(np.sqrt(var)) / float(np.sqrt(n))


def test_discriminative(model: ProbabilisticModel, test_loader: data.DataLoader, device: torch.device,
                        verbose: bool = True) -> Tuple[float, dict]:
    """Reference :429-478: ``(negative log-likelihood, sklearn classification report dict)``.  Predictions and
    targets are gathered on the host once, after the loop (and across the ranks when sharded)."""
    from sklearn import metrics as skm
    rank, world = _world()
    model.eval()
    run = RunningAverageMetric()
    preds, trues = [], []
    with torch.no_grad():
        for item in test_loader:
            inputs, targets = _batch(item, device, rank, world, True)
            if inputs.shape[0] == 0:
                continue
            outputs = model(inputs)
            run(model.loss(outputs, y=targets), num_samples=inputs.shape[0])
            preds.append(torch.argmax(outputs, dim=1))
            trues.append(targets)
    nll, = _epoch_averages([run], device, world)
    y_pred = torch.cat(preds).cpu() if preds else torch.empty(0, dtype=torch.long)
    y_true = torch.cat(trues).cpu() if trues else torch.empty(0, dtype=torch.long)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (y_true.tolist(), y_pred.tolist()))
        y_true = [v for t, _ in gathered for v in t]
        y_pred = [v for _, q in gathered for v in q]
    else:
        y_true, y_pred = y_true.tolist(), y_pred.tolist()
    return nll, skm.classification_report(y_true, y_pred, output_dict=True, zero_division=0)
