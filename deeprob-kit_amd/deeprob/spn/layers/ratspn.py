"""RAT-SPN layers behind the reference interface, evaluated by hand-written gfx950 kernels.

Module path, class names, constructor signatures, attribute / ``state_dict`` names and the forward
semantics follow ``deeprob/spn/layers/ratspn.py`` of the reference (cited per method); the per-batch
arithmetic lives in ``csrc/ratspn_*.hip`` and is reached through ``deeprob.hip.ops``.  There is no
PyTorch / CPU evaluation path: a CPU tensor raises ``deeprob.hip.HipError``.
"""
import abc
from typing import Optional, Tuple, List

import numpy as np
import torch
from torch import nn
from torch import distributions

from deeprob.torch.initializers import dirichlet_
from deeprob.hip import Workspace
from deeprob.hip import ops


class RegionGraphLayer(abc.ABC, nn.Module):
    def __init__(
        self,
        in_features: int,
        out_channels: int,
        regions: List[tuple],
        rg_depth: int,
        dropout: Optional[float] = None,
        **kwargs
    ):
        """
        Input-distribution layer over the leaf regions of a region graph.

        Buffers (reference: ratspn.py:42-66): ``mask [R,d]`` variable ids per region (short regions
        are right-padded by repeating their last id), ``pad_mask [R,1,d]`` (only if padding is
        needed), ``inv_mask [reps, D+pad]`` and ``inv_pad_mask`` for the top-down passes.

        :param in_features: number of input features D.
        :param out_channels: distributions per region I.
        :param regions: leaf regions (tuples of variable ids).
        :param rg_depth: depth of the region graph.
        :param dropout: leaf dropout rate (training only) or None.
        """
        super().__init__()
        self.in_features = in_features
        self.in_regions = len(regions)
        self.out_channels = out_channels
        self.rg_depth = rg_depth
        self.dropout = dropout
        self.distribution = None

        n_leaves = 2 ** self.rg_depth
        self.pad = -self.in_features % n_leaves
        padded_features = self.in_features + self.pad
        self.dimension = padded_features // n_leaves

        rows = [tuple(r) for r in regions]
        if self.pad > 0:
            dummy = np.zeros((len(rows), 1, self.dimension), dtype=np.bool_)
            for i, region in enumerate(rows):
                missing = self.dimension - len(region)
                if missing > 0:
                    dummy[i, :, self.dimension - missing:] = True
                    rows[i] = region + (region[-1],) * missing
            self.register_buffer('pad_mask', torch.tensor(dummy))
        self.register_buffer('mask', torch.tensor(rows))

        self.register_buffer('inv_mask', torch.argsort(self.mask.reshape(-1, padded_features), dim=1))
        if self.pad > 0:
            flat_pad = self.pad_mask.reshape(-1, padded_features)
            self.register_buffer('inv_pad_mask', torch.gather(flat_pad, dim=1, index=self.inv_mask))

        self._leaf_ctx = ops.LeafContext(self.in_features, self.in_regions, self.out_channels, self.dimension)

    def _pad_mask_or_none(self) -> Optional[torch.Tensor]:
        return self.pad_mask if self.pad > 0 else None

    def unpad_samples(self, x: torch.Tensor, idx_group: torch.Tensor) -> torch.Tensor:
        """Reorder region-major samples back to variable order and drop dummy variables
        (reference: ratspn.py:68-85)."""
        n_samples = idx_group.shape[0]
        idx_rep = torch.div(idx_group[:, 0], 2 ** self.rg_depth, rounding_mode='floor')
        samples = torch.gather(x, dim=1, index=self.inv_mask[idx_rep])
        if self.pad > 0:
            samples = samples[self.inv_pad_mask[idx_rep]].view(n_samples, self.in_features)
        return samples

    @abc.abstractmethod
    def _leaf_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Launch the leaf kernel for this distribution family."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """
        Log-likelihood of every leaf distribution, NaN inputs marginalised, dummy variables ignored
        (reference: ratspn.py:87-108).

        :param x: inputs ``[B, D]`` on a HIP device.
        :return: ``[B, R, I]``.
        """
        if self.training and self.dropout is not None:
            return self._leaf_forward_dropout(x, self.dropout, ops.draw_seed())
        return self._leaf_forward(x)

    @abc.abstractmethod
    def _leaf_forward_dropout(self, x: torch.Tensor, rate: float, seed: int) -> torch.Tensor:
        """Training-mode forward with input dropout (reference :98-100)."""

    @abc.abstractmethod
    def distribution_mode(self) -> torch.Tensor:
        """Mode of every leaf distribution, ``[R, I, d]``."""

    @torch.no_grad()
    def mpe(self, x: torch.Tensor, idx_group: torch.Tensor, idx_offset: torch.Tensor) -> torch.Tensor:
        """Fill NaN entries of ``x`` with the mode of the selected leaves (reference: ratspn.py:118-136)."""
        mode = self.distribution_mode()
        picked = torch.flatten(mode[idx_group, idx_offset], start_dim=1)
        picked = self.unpad_samples(picked, idx_group)
        return torch.where(torch.isnan(x), picked, x)

    @torch.no_grad()
    def sample(self, idx_group: torch.Tensor, idx_offset: torch.Tensor) -> torch.Tensor:
        """Sample the selected leaves (reference: ratspn.py:138-157)."""
        n_samples = idx_group.shape[0]
        draws = self.distribution.sample([n_samples])
        rows = torch.arange(n_samples, device=idx_group.device).unsqueeze(1)
        draws = torch.flatten(draws[rows, idx_group, idx_offset], start_dim=1)
        return self.unpad_samples(draws, idx_group)


class GaussianLayer(RegionGraphLayer):
    def __init__(
        self,
        in_features: int,
        out_channels: int,
        regions: List[tuple],
        rg_depth: int,
        dropout: Optional[float] = None,
        uniform_loc: Optional[Tuple[float, float]] = None,
        optimize_scale: bool = False
    ):
        """
        Gaussian leaves: parameters ``loc``, ``scale`` of shape ``[R, I, d]``; ``scale`` is frozen at 1
        unless ``optimize_scale`` (reference: ratspn.py:160-213).
        """
        super().__init__(in_features, out_channels, regions, rg_depth, dropout)
        shape = (self.in_regions, self.out_channels, self.dimension)
        if uniform_loc is None:
            self.loc = nn.Parameter(torch.randn(*shape), requires_grad=True)
        else:
            low, high = uniform_loc
            self.loc = nn.Parameter(low + (high - low) * torch.rand(*shape), requires_grad=True)
        if optimize_scale:
            self.scale = nn.Parameter(0.5 + 0.1 * torch.tanh(torch.randn(*shape)), requires_grad=True)
        else:
            self.scale = nn.Parameter(torch.ones(*shape), requires_grad=False)
        # same Parameter objects as the module's, as in the reference (used by sample / mode only)
        self.distribution = distributions.Normal(self.loc, self.scale, validate_args=False)

    def _leaf_forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.GaussianLeafFn.apply(x, self.loc, self.scale, self.mask, self._pad_mask_or_none(),
                                        self._leaf_ctx)

    def _leaf_forward_dropout(self, x: torch.Tensor, rate: float, seed: int) -> torch.Tensor:
        return ops.LeafDropoutFn.apply(x, self.loc, self.scale, self.mask, self._pad_mask_or_none(), self._leaf_ctx, 0,
                                       rate, seed)

    def distribution_mode(self) -> torch.Tensor:
        return self.distribution.mean


class BernoulliLayer(RegionGraphLayer):
    def __init__(
        self,
        in_features: int,
        out_channels: int,
        regions: List[tuple],
        rg_depth: int,
        dropout: Optional[float] = None
    ):
        """Bernoulli leaves with parameter ``logits [R, I, d]`` (reference: ratspn.py:216-247)."""
        super().__init__(in_features, out_channels, regions, rg_depth, dropout)
        self.logits = nn.Parameter(
            torch.randn(self.in_regions, self.out_channels, self.dimension), requires_grad=True
        )
        self.distribution = distributions.Bernoulli(logits=self.logits, validate_args=False)

    def _leaf_forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.BernoulliLeafFn.apply(x, self.logits, self.mask, self._pad_mask_or_none(), self._leaf_ctx)

    def _leaf_forward_dropout(self, x: torch.Tensor, rate: float, seed: int) -> torch.Tensor:
        return ops.LeafDropoutFn.apply(x, self.logits, None, self.mask, self._pad_mask_or_none(), self._leaf_ctx, 1,
                                       rate, seed)

    def distribution_mode(self) -> torch.Tensor:
        return (self.distribution.mean >= 0.5).float()


class ProductLayer(nn.Module):
    def __init__(self, in_regions: int, in_nodes: int):
        """
        Cross product of sibling regions in the log domain (reference: ratspn.py:250-270).

        :param in_regions: number of input regions (siblings are rows 2p and 2p+1).
        :param in_nodes: nodes per input region.
        """
        super().__init__()
        self.in_regions = in_regions
        self.in_nodes = in_nodes
        self.out_partitions = in_regions // 2
        self.out_nodes = in_nodes ** 2
        self.register_buffer('mask', torch.tensor([True, False] * self.out_partitions))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """``[B, R, N] -> [B, R/2, N*N]``, out[b,p,i*N+j] = x[b,2p,i] + x[b,2p+1,j] (reference :272-286)."""
        return ops.ProductFn.apply(x)

    @torch.no_grad()
    def mpe(self, x, idx_group, idx_offset):
        """Top-down index pass (reference :288-304)."""
        return self.sample(idx_group, idx_offset)

    @torch.no_grad()
    def sample(self, idx_group: torch.Tensor, idx_offset: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Split a product node index into its two children (reference :306-330)."""
        first = torch.div(idx_offset, self.in_nodes, rounding_mode='floor')
        second = torch.remainder(idx_offset, self.in_nodes)
        groups = torch.flatten(torch.stack([idx_group * 2, idx_group * 2 + 1], dim=2), start_dim=1)
        offsets = torch.flatten(torch.stack([first, second], dim=2), start_dim=1)
        return groups, offsets


class SumLayer(nn.Module):
    def __init__(self, in_partitions: int, in_nodes: int, out_nodes: int, dropout: Optional[float] = None):
        """
        Sum nodes over each partition; ``weight [P, S, N]`` are unnormalised log-weights initialised
        from a log-Dirichlet (reference: ratspn.py:333-361).
        """
        super().__init__()
        self.in_partitions = in_partitions
        self.in_nodes = in_nodes
        self.out_regions = in_partitions
        self.out_nodes = out_nodes
        self.dropout = dropout
        self.weight = nn.Parameter(torch.empty(self.out_regions, self.out_nodes, self.in_nodes), requires_grad=True)
        dirichlet_(self.weight, alpha=1.0)
        self._ws = Workspace()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """out[b,r,o] = logsumexp_n(x[b,r,n] + log_softmax(weight, 2)[r,o,n]) (reference :363-378)."""
        if self.training and self.dropout is not None:
            x = ops.DropoutFillFn.apply(x, self.dropout, ops.draw_seed())
        return ops.SumFn.apply(x, self.weight, self._ws)

    @torch.no_grad()
    def mpe(self, x, idx_group, idx_offset):
        """Arg-max child of the selected sum nodes (reference :380-399)."""
        rows = torch.arange(x.shape[0], device=x.device).unsqueeze(1)
        x = x[rows, idx_group]
        w = torch.log_softmax(self.weight[idx_group, idx_offset], dim=2)
        return idx_group, torch.argmax(x + w, dim=2)

    @torch.no_grad()
    def sample(self, idx_group, idx_offset):
        """Sample a child of the selected sum nodes from their weights (reference :401-417)."""
        w = torch.log_softmax(self.weight[idx_group, idx_offset], dim=2)
        return idx_group, distributions.Categorical(logits=w).sample()


class RootLayer(nn.Module):
    def __init__(self, in_partitions: int, in_nodes: int, out_classes: int):
        """Root sum nodes, ``weight [C, P*N]`` (reference: ratspn.py:420-444)."""
        super().__init__()
        self.in_partitions = in_partitions
        self.in_nodes = in_nodes
        self.out_classes = out_classes
        self.weight = nn.Parameter(
            torch.empty(self.out_classes, self.in_partitions * self.in_nodes), requires_grad=True
        )
        dirichlet_(self.weight, alpha=1.0)
        self._ws = Workspace()

    def forward(self, x):
        """out[b,c] = logsumexp_n(flatten(x)[b,n] + log_softmax(weight, 1)[c,n]) (reference :446-458)."""
        return ops.RootFn.apply(x, self.weight, self._ws)

    @torch.no_grad()
    def mpe(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Arg-max input of the root node of class ``y`` (reference :460-474)."""
        x = torch.flatten(x, start_dim=1)
        w = torch.log_softmax(self.weight, dim=1)
        idx = torch.argmax(x + w[y], dim=1, keepdim=True)
        return torch.div(idx, self.in_nodes, rounding_mode='floor'), torch.remainder(idx, self.in_nodes)

    @torch.no_grad()
    def sample(self, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Sample an input of the root node of class ``y`` (reference :476-490)."""
        w = torch.log_softmax(self.weight, dim=1)
        idx = distributions.Categorical(logits=w[y]).sample().unsqueeze(dim=1)
        return torch.div(idx, self.in_nodes, rounding_mode='floor'), torch.remainder(idx, self.in_nodes)
