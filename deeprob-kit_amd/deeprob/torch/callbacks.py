"""Early stopping with a best-state checkpoint (interface of deeprob/torch/callbacks.py:12-88)."""
import os
from collections import OrderedDict
from typing import Union

import torch
from torch import nn


class EarlyStopping:
    def __init__(self, model: nn.Module, patience: int = 1, filepath: Union[os.PathLike, str] = 'checkpoint.pt',
                 delta: float = 1e-3):
        """
        Stop when the monitored loss has not improved by more than ``delta`` for ``patience`` consecutive epochs;
        the best state_dict is kept in ``filepath``.

        :raises ValueError: if patience or delta are not positive.
        """
        if patience <= 0:
            raise ValueError("The patience value must be positive")
        if delta <= 0.0:
            raise ValueError("The delta value must be positive")
        self.model = model
        self.patience = patience
        self.filepath = filepath
        self.delta = delta
        self._best = float('inf')
        self._best_epoch = None
        self._stale = 0

    @property
    def should_stop(self) -> bool:
        return self._stale >= self.patience

    def get_best_state(self, map_location=None) -> OrderedDict:
        """The checkpointed state_dict; ``map_location`` as in ``torch.load`` (a rank of a sharded run passes its own
        device: the file holds rank 0's tensors)."""
        with open(self.filepath, 'rb') as f:
            return torch.load(f, map_location=map_location)

    def __call__(self, loss: float, epoch: int, save: bool = True):
        """Record the epoch's validation loss; ``save=False`` on the ranks that do not own the checkpoint file."""
        if loss < self._best - self.delta:
            self._best, self._best_epoch, self._stale = loss, epoch, 0
            if save:
                with open(self.filepath, 'wb') as f:
                    torch.save(self.model.state_dict(), f)
        else:
            self._stale += 1

    def __format__(self, format_spec) -> str:
        return "Best Loss: {:.4f} at Epoch: {}".format(self._best, self._best_epoch)
