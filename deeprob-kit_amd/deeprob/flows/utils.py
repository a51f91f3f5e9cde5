"""Bijector pieces of the 1-D flows behind the reference interface (deeprob/flows/utils.py).

`Bijector`, `BatchNormLayer1d`, `DequantizeLayer`, `LogitLayer` keep the reference's constructor
signatures, attributes and `state_dict` names.  Eval-mode batch norm runs on the HIP kernels (a
per-variable affine, foldable into the next coupling); Dequantize / Logit are cheap element-wise
device ops left to PyTorch, as SURVEY 2#7 allows (Dequantize is stochastic in the reference too).
The 2-D pieces (squeeze / un-squeeze, BatchNormLayer2d) run on csrc/flows2d.hip; their training direction
(batch statistics, gradients) on csrc/flows2d_train.hip.
"""
import abc
from typing import Union, Tuple

import numpy as np
import torch
from torch import nn

from deeprob.hip import Workspace


def squeeze_depth2d(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] -> [N, 4C, H/2, W/2], output channel c*4 + dy*2 + dx (reference :11-23)."""
    from deeprob.hip import ops_flows2d
    return ops_flows2d.space_to_depth(x, ops_flows2d.squeeze_table(x.shape[1], x.device))


def unsqueeze_depth2d(x: torch.Tensor) -> torch.Tensor:
    """[N, 4C, H, W] -> [N, C, 2H, 2W], the inverse of :func:`squeeze_depth2d` (reference :26-38)."""
    from deeprob.hip import ops_flows2d
    return ops_flows2d.depth_to_space(x, ops_flows2d.squeeze_table(x.shape[1] // 4, x.device))


class Bijector(abc.ABC, nn.Module):
    """Invertible transformation with a tractable log-det-Jacobian (reference :41-88)."""

    def __init__(self, in_features: Union[int, Tuple[int, int, int]]):
        if isinstance(in_features, torch.Size):
            in_features = tuple(in_features)
        if not isinstance(in_features, int):
            if not isinstance(in_features, tuple) or len(in_features) != 3:
                raise ValueError("The number of input features must be either an int or a (C, H, W) tuple")
        super().__init__()
        self.in_features = in_features
        self.out_features = in_features

    def forward(self, x: torch.Tensor, backward: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.apply_backward(x) if backward else self.apply_forward(x)

    @abc.abstractmethod
    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """data -> latent; returns (u, inverse log-det-Jacobian)."""

    @abc.abstractmethod
    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """latent -> data; returns (x, log-det-Jacobian)."""


class _EvalOnly(nn.Module):
    """Stand-in module (eval mode, no parameters) for the graph check of the free functions above."""


_EVAL = _EvalOnly().eval()


class BatchNormLayer1d(Bijector):
    def __init__(self, in_features: int, momentum: float = 0.9, eps: float = 1e-5):
        """Batch normalisation as a bijector (reference :91-116): parameters `weight` (log-gain), `bias`,
        buffers `running_var`, `running_mean`, all of shape [1, D].

        :raises ValueError: if momentum is not in (0, 1) or eps is not positive."""
        if momentum <= 0.0 or momentum >= 1.0:
            raise ValueError("The momentum value must be in (0, 1)")
        if eps <= 0.0:
            raise ValueError("The epsilon value must be positive")
        super().__init__(in_features)
        self.momentum = momentum
        self.eps = eps
        self.weight = nn.Parameter(torch.zeros(1, self.in_features), requires_grad=True)
        self.bias = nn.Parameter(torch.zeros(1, self.in_features), requires_grad=True)
        self.register_buffer('running_var', torch.ones(1, self.in_features))
        self.register_buffer('running_mean', torch.zeros(1, self.in_features))
        self._ws = Workspace()
        # process group over which train-mode statistics are taken when batches are sharded over ranks
        # (deeprob.parallel.synchronize_batchnorm); None: the statistics of the rows this process sees
        self.sync_group = None

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """u = (x - mean)/sqrt(var + eps) * exp(weight) + bias with the batch statistics (training: running
        statistics updated in place) or the running statistics (eval); reference :118-139."""
        from deeprob.hip import ops_flows
        if self.training or ops_flows._wants_graph(x, self.weight, self.bias):
            return ops_flows.BatchNormFn.apply(x, self.weight, self.bias, self)
        affine, ldj = ops_flows.bn1d_fold(self, inverse=False)
        return ops_flows.affine1d(x, affine), ldj.expand(x.shape[0])

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Inverse of apply_backward with the running statistics (reference :141-153)."""
        from deeprob.hip import ops_flows
        if ops_flows._wants_graph(u, self.weight, self.bias):
            return ops_flows.BatchNormInverseFn.apply(u, self.weight, self.bias, self)
        affine, ldj = ops_flows.bn1d_fold(self, inverse=True)
        return ops_flows.affine1d(u, affine), ldj.expand(u.shape[0])


class BatchNormLayer2d(Bijector):
    def __init__(self, in_features: int, momentum: float = 0.9, eps: float = 1e-5):
        """Per-channel batch normalisation as a bijector (reference :165-184): parameters `weight` (log-gain), `bias`,
        buffers `running_var`, `running_mean`, all of shape [1, C, 1, 1].

        :raises ValueError: if momentum is not in (0, 1) or eps is not positive."""
        if momentum <= 0.0 or momentum >= 1.0:
            raise ValueError("The momentum value must be in (0, 1)")
        if eps <= 0.0:
            raise ValueError("The epsilon value must be positive")
        super().__init__(in_features)
        self.momentum = momentum
        self.eps = eps
        self.weight = nn.Parameter(torch.zeros(1, self.in_features, 1, 1), requires_grad=True)
        self.bias = nn.Parameter(torch.zeros(1, self.in_features, 1, 1), requires_grad=True)
        self.register_buffer('running_var', torch.ones(1, self.in_features, 1, 1))
        self.register_buffer('running_mean', torch.zeros(1, self.in_features, 1, 1))

    def transform(self, x: torch.Tensor, inverse: bool, ldj=None) -> Tuple[torch.Tensor, torch.Tensor]:
        from deeprob.hip import ops_flows2d
        if not inverse and ops_flows2d.graph_route(x, self):
            # batch statistics / gradients (reference :190-207): deeprob/hip/ops_flows2d_train.py
            from deeprob.hip import ops_flows2d_train
            u, d = ops_flows2d_train.bn2d(ops_flows2d._image(x, 'x'), self)
            return u, (d if ldj is None else ldj + d)
        ops_flows2d.require_eval(self, 'BatchNormLayer2d', x)
        return ops_flows2d.bn2d(x, self, inverse, ldj)

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """u = (x - mean)/sqrt(var + eps) * exp(weight) + bias, ildj = H W sum_c(weight - log(var + eps)/2)
        (reference :186-208; batch statistics in training mode)."""
        return self.transform(x, False)

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The inverse with the running statistics (reference :210-222)."""
        return self.transform(u, True)


class DequantizeLayer(Bijector):
    def __init__(self, in_features: Union[int, Tuple[int, int, int]], n_bits: int = 8):
        """Uniform dequantisation of `n_bits` data scaled to [0, 1] (reference :224-255).

        :raises ValueError: if n_bits is not positive."""
        if n_bits <= 0:
            raise ValueError("The number of bits must be positive")
        super().__init__(in_features)
        self.n_bits = n_bits
        self.bins = 2 ** self.n_bits
        dims = np.prod(self.in_features)
        self.register_buffer('ldj', torch.tensor(dims * np.log(self.bins), dtype=torch.float32))

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        u = (x * (self.bins - 1) + torch.rand_like(x)) / self.bins
        return u, -self.ldj.expand(x.shape[0])

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = torch.clamp(torch.floor(u * self.bins), min=0, max=self.bins - 1) / (self.bins - 1)
        return x, self.ldj.expand(u.shape[0])


class LogitLayer(Bijector):
    def __init__(self, in_features: Union[int, Tuple[int, int, int]], alpha: float = 0.05):
        """u = logit(alpha + (1 - 2 alpha) x) (reference :257-294).

        :raises ValueError: if alpha is not in (0, 1)."""
        if alpha <= 0.0 or alpha >= 1.0:
            raise ValueError("The alpha logit parameter must be in (0, 1)")
        super().__init__(in_features)
        self.alpha = alpha
        dims = np.prod(self.in_features)
        self.register_buffer('ldj', torch.tensor(-dims * np.log(1.0 - 2.0 * self.alpha), dtype=torch.float32))
        self._ldj_host = None

    def _hip(self, x: torch.Tensor, inverse: bool):
        """One HIP pass when no autograd graph through the layer is wanted (the input is data): the layer has no
        parameters, so only an input that requires grad keeps the element-wise torch route."""
        if not (x.is_cuda and x.dtype == torch.float32) or (torch.is_grad_enabled() and x.requires_grad):
            return None
        from deeprob.hip import load_library, check, ptr, stream_ptr
        xc = x.contiguous()
        n = xc.shape[0]
        out = torch.empty_like(xc)
        ldj = torch.empty(n, dtype=torch.float32, device=xc.device)
        if self._ldj_host is None:
            self._ldj_host = float(-np.prod(self.in_features) * np.log(1.0 - 2.0 * self.alpha))
        check(load_library().dpk_logit1d_forward(ptr(xc), n, xc.numel() // max(n, 1) if n else 1, float(self.alpha),
                                                 self._ldj_host, int(inverse), ptr(out), ptr(ldj),
                                                 stream_ptr(xc.device)), 'dpk_logit1d_forward')
        return out, ldj

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        r = self._hip(x, False)
        if r is not None:
            return r
        n = x.shape[0]
        p = self.alpha + (1.0 - 2.0 * self.alpha) * x
        log_p, log_q = torch.log(p), torch.log(1.0 - p)
        ldj = torch.sum((log_p + log_q).view(n, -1), dim=1) + self.ldj
        return log_p - log_q, -ldj

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        r = self._hip(u, True)
        if r is not None:
            return r
        n = u.shape[0]
        p = torch.sigmoid(u)
        x = (p - self.alpha) / (1.0 - 2.0 * self.alpha)
        ldj = torch.sum((torch.log(p) + torch.log(1.0 - p)).view(n, -1), dim=1) + self.ldj
        return x, ldj
