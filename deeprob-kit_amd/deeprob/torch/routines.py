"""Train / test loops behind the reference interface (deeprob/torch/routines.py:21-210, :349-426), generative setting.

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
from deeprob.parallel import allreduce_gradients, shard_batch


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
    verbose: bool = True
) -> Dict[str, list]:
    """Reference signature (:21-37).  ``batch_size`` is the GLOBAL batch: with N ranks each one sees batch_size/N
    samples per step.  :raises ValueError: for an unknown setting or non-positive epochs."""
    if setting != 'generative':
        raise ValueError("Unknown train setting called {}".format(setting) if setting != 'discriminative' else
                         "The discriminative routines are outside the HIP density-evaluation path")
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or device.type != 'cuda':
        raise ValueError("deeprob on MI355X trains on a HIP device (there is no CPU path)")
    train_loader = data.DataLoader(data_train, batch_size, shuffle=True, drop_last=drop_last, num_workers=num_workers)
    valid_loader = data.DataLoader(data_valid, batch_size, shuffle=False, drop_last=False, num_workers=num_workers)
    model.to(device)
    opt = get_optimizer_class(optimizer)(filter(lambda p: p.requires_grad, model.parameters()), lr=lr,
                                         **(optimizer_kwargs or {}))
    early_stopping = EarlyStopping(model, patience=patience, filepath=checkpoint)
    return train_generative(model, train_loader, valid_loader, opt, device, early_stopping, epochs, train_base, verbose)


def train_generative(
    model: ProbabilisticModel,
    train_loader: data.DataLoader,
    valid_loader: data.DataLoader,
    optimizer: optim.Optimizer,
    device: torch.device,
    early_stopping: EarlyStopping,
    epochs: int = 1000,
    train_base: bool = True,
    verbose: bool = True
) -> Dict[str, list]:
    """Reference :98-210.  Returns ``{'train': [...], 'valid': [...]}`` (average loss per epoch)."""
    if epochs <= 0:
        raise ValueError("The number of epochs must be positve")
    rank, world = _world()
    history = {'train': [], 'valid': []}
    run_train, run_valid = RunningAverageMetric(), RunningAverageMetric()
    for epoch in range(1, epochs + 1):
        run_train.reset()
        run_valid.reset()
        t0 = time.perf_counter()
        _train_mode(model, train_base)
        for inputs in train_loader:
            inputs = shard_batch(inputs.to(device, non_blocking=True), rank, world)
            n_local = inputs.shape[0]
            optimizer.zero_grad()
            if n_local > 0:
                loss = model.loss(model(inputs))
                loss.backward()
            if world > 1:
                # every rank joins the collective, also one whose shard of a short last batch is empty
                allreduce_gradients(model, weight=n_local)
            optimizer.step()
            model.apply_constraints()
            if n_local > 0:
                run_train(loss, num_samples=n_local)              # stays on the device
        model.eval()
        with torch.no_grad():
            for inputs in valid_loader:
                inputs = shard_batch(inputs.to(device, non_blocking=True), rank, world)
                if inputs.shape[0] == 0:
                    continue
                run_valid(model.loss(model(inputs)), num_samples=inputs.shape[0])
        train_loss, valid_loss = _epoch_averages((run_train, run_valid), device, world)
        elapsed = int(time.perf_counter() - t0)
        if verbose and rank == 0:
            print("Epoch {}/{} - train_loss: {:.4f}, valid_loss: {:.4f} [{}s]".format(
                epoch, epochs, train_loss, valid_loss, elapsed if elapsed > 0 else '<1'))
        history['train'].append(train_loss)
        history['valid'].append(valid_loss)
        early_stopping(valid_loss, epoch, save=(rank == 0))
        if early_stopping.should_stop:
            if verbose and rank == 0:
                print("Early Stopping... {}".format(early_stopping))
            break
    if world > 1:
        dist.barrier()   # rank 0 finished writing the checkpoint
    model.load_state_dict(early_stopping.get_best_state())
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
    if setting != 'generative':
        raise ValueError("Unknown test setting called {}".format(setting))
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or device.type != 'cuda':
        raise ValueError("deeprob on MI355X evaluates on a HIP device (there is no CPU path)")
    loader = data.DataLoader(data_test, batch_size, shuffle=False, drop_last=False, num_workers=num_workers)
    model.to(device)
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
