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


class _WeightNormConvParameters(nn.Module):
    """Parameter holder with the names ``torch.nn.utils.weight_norm(nn.Conv2d(...))`` gives its module (``bias``,
    ``weight_g`` [Cout,1,1,1], ``weight_v`` [Cout,Cin,k,k]) so that reference checkpoints load unchanged."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, bias: bool):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        # the same initial values (and the same draws from torch's generator) as nn.Conv2d + weight_norm
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=bias)
        weight = conv.weight.detach()
        self.bias = nn.Parameter(conv.bias.detach().clone()) if bias else None
        self.weight_g = nn.Parameter(weight.reshape(out_channels, -1).norm(dim=1).reshape(out_channels, 1, 1, 1))
        self.weight_v = nn.Parameter(weight.clone())


class WeightNormConv2d(nn.Module):
    """Weight-normalised convolution (reference :86-121), evaluated by the HIP convolution kernel
    (csrc/flows2d.hip); 1x1 and 3x3 kernels with stride 1 and "same" padding -- what the flow conditioners use."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, bias: bool = True):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        same = lambda v, want: (v == want) if isinstance(v, int) else all(e == want for e in v)  # noqa: E731
        if not isinstance(kernel_size, int) and kernel_size[0] != kernel_size[1]:
            raise NotImplementedError("WeightNormConv2d: only square kernels are built")
        if ks not in (1, 3) or not same(stride, 1) or not same(padding, ks // 2):
            raise NotImplementedError("WeightNormConv2d: only 1x1 / 3x3 kernels with stride 1 and 'same' padding are "
                                      "built (kernel {}, stride {}, padding {})".format(kernel_size, stride, padding))
        self.conv = _WeightNormConvParameters(in_channels, out_channels, ks, bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from deeprob.hip import ops_flows2d
        return ops_flows2d.conv2d(x, self)
