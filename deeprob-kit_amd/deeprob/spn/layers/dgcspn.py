"""DGC-SPN layers behind the reference interface (deeprob/spn/layers/dgcspn.py).

Constructors, attributes and ``state_dict`` names follow the reference; the forward / backward
arithmetic is served by the spatial kernels in ``csrc/dgcspn.hip`` through ``deeprob.hip.ops_spatial``.
"""
from itertools import product
from typing import Optional, Union, Tuple

import numpy as np
import torch
from torch import nn

from deeprob.torch.initializers import dirichlet_
from deeprob.hip import Workspace


def _pair(v: Union[int, Tuple[int, int]]) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else tuple(v)


class _SpatialShape:
    """in/out (C, H, W) accessors shared by the spatial layers (reference properties :77-99)."""
    in_features: Tuple[int, int, int]
    out_features: Tuple[int, int, int]

    @property
    def in_channels(self) -> int:
        return self.in_features[0]

    @property
    def in_height(self) -> int:
        return self.in_features[1]

    @property
    def in_width(self) -> int:
        return self.in_features[2]

    @property
    def out_channels(self) -> int:
        return self.out_features[0]

    @property
    def out_height(self) -> int:
        return self.out_features[1]

    @property
    def out_width(self) -> int:
        return self.out_features[2]


class SpatialGaussianLayer(_SpatialShape, nn.Module):
    def __init__(
        self,
        in_features: Tuple[int, int, int],
        out_channels: int,
        optimize_scale: bool = False,
        dropout: Optional[float] = None,
        quantiles_loc: Optional[np.ndarray] = None,
        uniform_loc: Optional[Tuple[float, float]] = None
    ):
        """
        Pixel-wise Gaussian leaves, ``loc`` / ``scale`` of shape ``[K, C, H, W]`` (reference :14-75).

        :raises ValueError: if both quantiles_loc and uniform_loc are given.
        """
        if quantiles_loc is not None and uniform_loc is not None:
            raise ValueError("At most one between quantiles_loc and uniform_loc can be specified")
        super().__init__()
        self.in_features = tuple(in_features)
        self.out_features = (out_channels, self.in_features[1], self.in_features[2])
        self.dropout = dropout
        shape = (out_channels,) + self.in_features
        if quantiles_loc is not None:
            loc = torch.tensor(quantiles_loc, dtype=torch.float32)
        elif uniform_loc is not None:
            low, high = uniform_loc
            loc = torch.linspace(low, high, steps=out_channels).view(-1, 1, 1, 1).repeat(1, *self.in_features)
        else:
            loc = torch.randn(*shape)
        self.loc = nn.Parameter(loc, requires_grad=True)
        if optimize_scale:
            self.scale = nn.Parameter(0.5 + 0.1 * torch.tanh(torch.randn(*shape)), requires_grad=True)
        else:
            self.scale = nn.Parameter(torch.ones(*shape), requires_grad=False)
        self.distribution = torch.distributions.Normal(self.loc, self.scale, validate_args=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """``[B,C,H,W] -> [B,K,H,W]``: sum over input channels of the NaN-marginalised Normal log-density
        (reference :101-120)."""
        from deeprob.hip import ops_spatial
        if self.training and self.dropout is not None:
            from deeprob.hip import ops
            return ops_spatial.SpatialGaussianFn.apply(x, self.loc, self.scale, self.dropout, ops.draw_seed())
        return ops_spatial.SpatialGaussianFn.apply(x, self.loc, self.scale, 0.0, 0)


class SpatialProductLayer(_SpatialShape, nn.Module):
    def __init__(
        self,
        in_features: Tuple[int, int, int],
        kernel_size: Union[int, Tuple[int, int]],
        padding: str,
        stride: Union[int, Tuple[int, int]],
        dilation: Union[int, Tuple[int, int]],
        depthwise: bool = True
    ):
        """
        Product over a dilated ``kh x kw`` window (a sum of log-densities), depthwise or over every
        channel combination (reference :123-198).

        :param padding: 'valid' (none), 'full' (effective kernel - 1 on every side) or 'final'
                        (one-sided, right / bottom).
        :raises ValueError: for an unknown padding mode.
        """
        super().__init__()
        self.in_features = tuple(in_features)
        self.groups = self.in_channels if depthwise else 1
        kh, kw = _pair(kernel_size)
        self.stride = _pair(stride)
        self.dilation = _pair(dilation)
        eff_h = (kh - 1) * self.dilation[0] + 1
        eff_w = (kw - 1) * self.dilation[1] + 1
        if padding == 'valid':
            self.pad = [0, 0, 0, 0]
        elif padding == 'full':
            self.pad = [eff_w - 1, eff_w - 1, eff_h - 1, eff_h - 1]
        elif padding == 'final':
            self.pad = [0, (eff_w - 1) * 2 - self.in_width, 0, (eff_h - 1) * 2 - self.in_height]
        else:
            raise ValueError("Padding mode must be either 'valid', 'full' or 'final'")
        span_h = self.pad[2] + self.pad[3] + self.in_height - eff_h + 1
        span_w = self.pad[0] + self.pad[1] + self.in_width - eff_w + 1
        out_h = int(np.ceil(span_h / self.stride[0]))
        out_w = int(np.ceil(span_w / self.stride[1]))
        taps = kh * kw
        out_c = self.in_channels if depthwise else self.in_channels ** taps
        self.out_features = (out_c, out_h, out_w)
        self.depthwise = depthwise
        self.kernel_size = (kh, kw)

        if depthwise:
            weight = torch.ones(out_c, 1, kh, kw)
        else:
            # one-hot kernels enumerating every combination of input channels over the window, in
            # itertools.product order (reference :187-193)
            combos = np.array(list(product(range(self.in_channels), repeat=taps))).reshape(out_c, 1, kh, kw)
            channel = np.arange(self.in_channels).reshape(1, -1, 1, 1)
            weight = torch.tensor(np.equal(channel, combos), dtype=torch.float32)
        self.register_buffer('weight', weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Zero-pad, then sum the window taps (reference :224-236)."""
        from deeprob.hip import ops_spatial
        return ops_spatial.SpatialProductFn.apply(x, self)


class SpatialSumLayer(_SpatialShape, nn.Module):
    def __init__(self, in_features: Tuple[int, int, int], out_channels: int, dropout: Optional[float] = None):
        """Per-pixel sum nodes with position-dependent weights ``[Cout, Cin, H, W]`` (reference :239-263)."""
        super().__init__()
        self.in_features = tuple(in_features)
        self.out_features = (out_channels, self.in_features[1], self.in_features[2])
        self.dropout = dropout
        self.weight = nn.Parameter(torch.empty(out_channels, *self.in_features), requires_grad=True)
        dirichlet_(self.weight, alpha=1.0, dim=1)
        self._ws = Workspace()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """out[b,o,h,w] = logsumexp_c(x[b,c,h,w] + log_softmax(weight, 1)[o,c,h,w]) (reference :289-304)."""
        from deeprob.hip import ops_spatial
        if self.training and self.dropout is not None:
            from deeprob.hip import ops
            x = ops.DropoutFillFn.apply(x, self.dropout, ops.draw_seed())
        return ops_spatial.SpatialSumFn.apply(x, self.weight, self._ws)


class SpatialRootLayer(nn.Module):
    def __init__(self, in_features: Tuple[int, int, int], out_channels: int):
        """Root sum nodes over the flattened map, ``weight [C, Cin*H*W]`` (reference :307-327)."""
        super().__init__()
        self.in_features = tuple(in_features)
        self.out_channels = out_channels
        flat = int(np.prod(self.in_features))
        self.weight = nn.Parameter(torch.empty(self.out_channels, flat), requires_grad=True)
        dirichlet_(self.weight, alpha=1.0)
        self._ws = Workspace()
        self._ws2 = Workspace()   # fused product+root route (deeprob.hip.ops_spatial.spatial_prodroot)
        self._ws3 = Workspace()   # fused sum level + product + root route (ops_spatial.spatial_sumprodroot)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Flatten + log-sum-exp with ``log_softmax(weight, 1)`` (reference :343-355)."""
        from deeprob.hip import ops
        return ops.RootFn.apply(x, self.weight, self._ws)
