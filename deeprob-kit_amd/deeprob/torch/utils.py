"""Small torch helpers used by the flows (interface of deeprob/torch/utils.py:12-70)."""
from typing import Union

import torch
from torch import nn
from torch import optim


def get_activation_class(name: str):
    """'relu' | 'leaky-relu' | 'softplus' | 'tanh' | 'sigmoid' -> nn.Module class (ValueError otherwise)."""
    table = {'relu': nn.ReLU, 'leaky-relu': nn.LeakyReLU, 'softplus': nn.Softplus, 'tanh': nn.Tanh,
             'sigmoid': nn.Sigmoid}
    try:
        return table[name]
    except KeyError as ex:
        raise ValueError from ex


def get_optimizer_class(name: str):
    """'sgd' | 'rmsprop' | 'adagrad' | 'adam' -> optimiser class (ValueError otherwise)."""
    table = {'sgd': optim.SGD, 'rmsprop': optim.RMSprop, 'adagrad': optim.Adagrad, 'adam': optim.Adam}
    try:
        return table[name]
    except KeyError as ex:
        raise ValueError from ex


class ScaledTanh(nn.Module):
    """``weight * tanh(x)`` with a learnable, zero-initialised weight (reference :52-70)."""

    def __init__(self, weight_size: Union[int, tuple, list] = 1):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(weight_size), requires_grad=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.weight * torch.tanh(x)
