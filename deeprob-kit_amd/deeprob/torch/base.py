"""Probabilistic model contract (interface of deeprob/torch/base.py:11-53)."""
import abc
from typing import Optional, Union

import torch
from torch import nn
from torch import distributions


class ProbabilisticModel(abc.ABC, nn.Module):
    """Base class: ``forward`` of a sub-class evaluates batched log-likelihoods."""

    has_rsample = False

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        """Batched log-likelihood; identical to calling the module."""
        return self(x)

    @abc.abstractmethod
    def sample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Draw ``n_samples`` samples (optionally class conditioned)."""

    @abc.abstractmethod
    def loss(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Training loss from the model outputs ``x`` (and labels ``y``)."""

    def apply_constraints(self):
        """Project the parameters back to their domain after an optimiser step (default: nothing)."""


#: A density estimator: a probabilistic model or a torch distribution.
DensityEstimator = Union[ProbabilisticModel, distributions.Distribution]
