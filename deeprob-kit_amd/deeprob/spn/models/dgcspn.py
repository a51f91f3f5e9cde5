"""DGC-SPN model behind the reference interface (deeprob/spn/models/dgcspn.py)."""
from typing import Optional, Union, Tuple, List

import numpy as np
import torch
from torch import autograd

from deeprob.torch.base import ProbabilisticModel
from deeprob.torch.constraints import ScaleClipper
from deeprob.spn.layers.dgcspn import SpatialGaussianLayer, SpatialProductLayer, SpatialSumLayer, SpatialRootLayer


class DgcSpn(ProbabilisticModel):
    def __init__(
        self,
        in_features: Tuple[int, int, int],
        out_classes: int = 1,
        n_batch: int = 8,
        sum_channels: int = 8,
        depthwise: Union[bool, List[bool]] = False,
        n_pooling: int = 0,
        optimize_scale: bool = False,
        in_dropout: Optional[float] = None,
        sum_dropout: Optional[float] = None,
        quantiles_loc: Optional[np.ndarray] = None,
        uniform_loc: Optional[Tuple[float, float]] = None
    ):
        """
        Deep generalized convolutional SPN (constructor contract of the reference, dgcspn.py:16-131).

        :raises ValueError: if a parameter is out of domain.
        """
        if in_features[1] != in_features[2]:
            raise ValueError("The height and width of input size must be the same")
        if out_classes <= 0:
            raise ValueError("The number of output classes must be positive")
        if n_batch <= 0:
            raise ValueError("The number of base distribution batches must be positive")
        if sum_channels <= 0:
            raise ValueError("The number of output channels of spatial sum layers must be positive")
        if in_dropout is not None and not 0.0 < in_dropout < 1.0:
            raise ValueError("The dropout rate at base distribution must be in (0, 1)")
        if sum_dropout is not None and not 0.0 < sum_dropout < 1.0:
            raise ValueError("The dropout rate at spatial sum layers must be in (0, 1)")
        if quantiles_loc is not None and uniform_loc is not None:
            raise ValueError("At least one between quantiles_loc and uniform_loc must be None")
        if quantiles_loc is not None and len(quantiles_loc.shape) != 4:
            raise ValueError("The mean quantiles must be a 4D Numpy array")
        if uniform_loc is not None and (len(uniform_loc) != 2 or uniform_loc[0] >= uniform_loc[1]):
            raise ValueError("The uniform range must be a pair (A, B) with A < B")

        depth = int(np.ceil(np.log2(in_features[1])))
        if isinstance(depthwise, bool):
            depthwise = [depthwise] * (depth + 1)
        else:
            if len(depthwise) == 0 or len(depthwise) > depth + 1:
                raise ValueError("The length of depthwise argument must be in [1, ceil(log2(D)) + 1]")
            depthwise = list(depthwise) + [depthwise[-1]] * (depth + 1 - len(depthwise))
        if n_pooling < 0 or n_pooling > depth:
            raise ValueError("The number of initial pooling spatial product layers must be in [0, ceil(log2(D))]")

        super().__init__()
        self.in_features = in_features
        self.out_classes = out_classes
        self.n_batch = n_batch
        self.sum_channels = sum_channels
        self.depthwise = depthwise
        self.n_pooling = n_pooling
        self.optimize_scale = optimize_scale
        self.in_dropout = in_dropout
        self.sum_dropout = sum_dropout
        self.layers = torch.nn.ModuleList()

        self.base_layer = SpatialGaussianLayer(
            self.in_features, self.n_batch, optimize_scale=self.optimize_scale, dropout=self.in_dropout,
            quantiles_loc=quantiles_loc, uniform_loc=uniform_loc
        )
        shape = self.base_layer.out_features

        # level i: pooling product (valid, stride 2) for i < n_pooling, else dilated product with
        # 'full' padding ('final' at the last level); every product but the last feeds a sum layer
        for i in range(depth + 1):
            if i < self.n_pooling:
                padding, stride, dilation = 'valid', (2, 2), (1, 1)
            else:
                padding = 'final' if i == depth else 'full'
                stride = (1, 1)
                dilation = (2 ** (i - self.n_pooling),) * 2
            prod = SpatialProductLayer(shape, kernel_size=(2, 2), padding=padding, stride=stride,
                                       dilation=dilation, depthwise=self.depthwise[i])
            self.layers.append(prod)
            shape = prod.out_features
            if i != depth:
                ssum = SpatialSumLayer(shape, self.sum_channels, self.sum_dropout)
                self.layers.append(ssum)
                shape = ssum.out_features
        self.root_layer = SpatialRootLayer(shape, self.out_classes)
        if self.optimize_scale:
            self.scale_clipper = ScaleClipper()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Log-likelihood ``[B, out_classes]`` of images ``x [B,C,H,W]``, NaN = marginalised
        (reference: dgcspn.py:134-151)."""
        if self._needs_graph(x):
            # training / gradients: the layer chain, with every depthwise product + sum pair of at most 8 channels as
            # one autograd node (no product map is written or kept; its backward recomputes it from the taps)
            from deeprob.hip import ops_spatial
            x = self.base_layer(x)
            i, n = 0, len(self.layers)
            while i < n:
                layer = self.layers[i]
                if (i + 1 < n and isinstance(layer, SpatialProductLayer) and isinstance(self.layers[i + 1], SpatialSumLayer)
                        and not (self.training and self.layers[i + 1].dropout is not None)):
                    y = ops_spatial.spatial_prodsum_autograd(x, layer, self.layers[i + 1].weight, self.layers[i + 1]._ws)
                    if y is not None:
                        x, i = y, i + 2
                        continue
                x = layer(x)
                i += 1
            return self.root_layer(x)
        # evaluation: every depthwise product is folded into the sum layer above it
        from deeprob.hip import ops_spatial
        # (the tables of every fused level rebuilt by one launch for the whole forward instead of one per level)
        ops_spatial.tables_prepare(self._fused_levels(), x)
        try:
            return self._forward_eval(x)
        finally:
            ops_spatial.tables_release()

    def _fused_levels(self):
        """The (product, sum) levels the evaluation loop below hands to spatial_prodsum / spatial_sumprodroot, in order."""
        levels, i, n = [], 0, len(self.layers)
        while i < n:
            layer = self.layers[i]
            nxt = self.layers[i + 1] if i + 1 < n else None
            if not (isinstance(layer, SpatialProductLayer) and isinstance(nxt, SpatialSumLayer)) or \
                    (self.training and nxt.dropout is not None):
                i += 1
                continue
            if i == n - 3 and isinstance(self.layers[i + 2], SpatialProductLayer):
                levels.append(('sumprodroot', layer, nxt, self.layers[i + 2], self.root_layer))
                break
            levels.append(('prodsum', layer, nxt))
            i += 2
        return levels

    def _forward_eval(self, x: torch.Tensor) -> torch.Tensor:
        from deeprob.hip import ops_spatial
        i, n = 0, len(self.layers)
        y = None
        if (n >= 4 and isinstance(self.base_layer, SpatialGaussianLayer) and isinstance(self.layers[0], SpatialProductLayer)
                and isinstance(self.layers[1], SpatialSumLayer) and not self.training):
            # the Gaussian leaf layer folded into the first (pooling) level: the leaf map is never written
            y = ops_spatial.spatial_leaf_prodsum(x, self.base_layer, self.layers[0], self.layers[1].weight, self.layers[1]._ws,
                                                 out_pixel_major=self._level_streams(2, x.shape[0]))
        if y is not None:
            x, i = y, 2
        else:
            x = self.base_layer(x)
        while i < n:
            layer = self.layers[i]
            if (i == n - 3 and isinstance(layer, SpatialProductLayer) and isinstance(self.layers[i + 1], SpatialSumLayer)
                    and isinstance(self.layers[i + 2], SpatialProductLayer)
                    and not (self.training and self.layers[i + 1].dropout is not None)):
                # last sum level + last product + root: the largest map stays on chip
                y = ops_spatial.spatial_sumprodroot(x, layer, self.layers[i + 1].weight, self.layers[i + 2],
                                                    self.root_layer.weight, self.root_layer._ws3)
                if y is not None:
                    return y
            if i + 1 < n and isinstance(layer, SpatialProductLayer) and isinstance(self.layers[i + 1], SpatialSumLayer):
                nxt = self.layers[i + 1]
                dropout = self.training and nxt.dropout is not None   # the sum layer raises: not on this path
                # (the map between two streaming levels stays pixel-major: csrc/dgcspn_stream.hip, round 6)
                y = None if dropout else ops_spatial.spatial_prodsum(x, layer, nxt.weight, nxt._ws,
                                                                     out_pixel_major=self._level_streams(i + 2, x.shape[0]))
                if y is not None:
                    x, i = y, i + 2
                    continue
            if i == n - 1 and isinstance(layer, SpatialProductLayer):
                # the last product feeds the root sum nodes directly
                y = ops_spatial.spatial_prodroot(x, layer, self.root_layer.weight, self.root_layer._ws2)
                if y is not None:
                    return y
            x = layer(x)
            i += 1
        return self.root_layer(x)

    def _level_streams(self, i: int, B: int) -> bool:
        """Whether the fused level that starts at layer ``i`` of the evaluation loop runs on the streaming route (the
        library's answer): its input may then arrive pixel-major."""
        from deeprob.hip import ops_spatial
        n = len(self.layers)
        if i + 1 >= n or self.training:
            return False
        layer, nxt = self.layers[i], self.layers[i + 1]
        if not (isinstance(layer, SpatialProductLayer) and isinstance(nxt, SpatialSumLayer)):
            return False
        if i == n - 3 and isinstance(self.layers[i + 2], SpatialProductLayer):
            return ops_spatial.level_streams(layer, B, nxt.weight.shape[0], self.layers[i + 2], self.root_layer.weight.shape[0])
        return ops_spatial.level_streams(layer, B, nxt.weight.shape[0])

    def _needs_graph(self, x: torch.Tensor) -> bool:
        if not torch.is_grad_enabled():
            return False
        return x.requires_grad or any(p.requires_grad for p in self.parameters())

    def mpe(self, x: torch.Tensor) -> torch.Tensor:
        """Gradient-based MPE completion of NaN pixels (reference: dgcspn.py:153-184)."""
        z = self.base_layer(x)
        if not z.requires_grad:
            z.requires_grad = True
        y = z
        for layer in self.layers:
            y = layer(y)
        y = self.root_layer(y)
        z_grad, = autograd.grad(y, z, grad_outputs=torch.ones_like(y), only_inputs=True)
        with torch.no_grad():
            estimates = torch.sum(torch.unsqueeze(z_grad, dim=2) * self.base_layer.loc, dim=1)
            return torch.where(torch.isnan(x), estimates, x)

    def sample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError("Sampling is not implemented for DGC-SPNs")

    def loss(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.out_classes == 1:
            from deeprob.hip import ops
            return ops.neg_mean(x)
        return torch.nn.functional.nll_loss(torch.log_softmax(x, dim=1), y)

    def apply_constraints(self):
        if self.optimize_scale:
            self.scale_clipper(self.base_layer)
