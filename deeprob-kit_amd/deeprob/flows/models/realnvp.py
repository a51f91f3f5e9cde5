"""RealNVP-1D and RealNVP-2D behind the reference interface (deeprob/flows/models/realnvp.py:16-72, :75-220).
RealNVP2d: density / sampling directions with running statistics on csrc/flows2d.hip; training mode and gradients of
the density direction on csrc/flows2d_train.hip (deeprob/hip/ops_flows2d_train.py)."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from deeprob.torch.base import DensityEstimator
from deeprob.flows.utils import BatchNormLayer1d
from deeprob.flows.layers.coupling import CouplingLayer1d, CouplingBlock2d
from deeprob.flows.models.base import NormalizingFlow


class RealNVP1d(NormalizingFlow):
    def __init__(
        self,
        in_features: int,
        dequantize: bool = False,
        logit: Optional[float] = None,
        in_base: Optional[DensityEstimator] = None,
        n_flows: int = 5,
        depth: int = 1,
        units: int = 128,
        batch_norm: bool = True,
        affine: bool = True
    ):
        """
        Stack of `n_flows` coupling layers with alternating masks, each optionally followed by a batch
        normalisation bijector.

        :raises ValueError: if n_flows, depth or units is not positive.
        """
        if n_flows <= 0:
            raise ValueError("The number of coupling flow layers must be positive")
        if depth <= 0:
            raise ValueError("The number of hidden layers of conditioners must be positive")
        if units <= 0:
            raise ValueError("The number of hidden units per layer must be positive")
        super().__init__(in_features, dequantize=dequantize, logit=logit, in_base=in_base)
        self.n_flows = n_flows
        self.depth = depth
        self.units = units
        self.batch_norm = batch_norm
        self.affine = affine
        for i in range(self.n_flows):
            self.layers.append(CouplingLayer1d(self.in_features, self.depth, self.units, affine=self.affine,
                                               reverse=(i % 2 == 1)))
            if self.batch_norm:
                self.layers.append(BatchNormLayer1d(self.in_features))


class RealNVP2d(NormalizingFlow):
    def __init__(
        self,
        in_features: Tuple[int, int, int],
        dequantize: bool = False,
        logit: Optional[float] = None,
        in_base: Optional[DensityEstimator] = None,
        network: str = 'resnet',
        n_flows: int = 1,
        n_blocks: int = 2,
        channels: int = 32,
        affine: bool = True
    ):
        """
        Multi-scale RealNVP on images: `n_flows` coupling blocks, each followed by a down-scaling permutation that
        factors out half of the channels, and a last coupling block (reference :75-139).

        :raises ValueError: if n_flows, n_blocks or channels is not positive.
        """
        if n_flows <= 0:
            raise ValueError("The number of coupling flow layers must be positive")
        if n_blocks <= 0:
            raise ValueError("The number of conditioners blocks must be positive")
        if channels <= 0:
            raise ValueError("The number of channels must be positive")
        super().__init__(in_features, dequantize=dequantize, logit=logit, in_base=in_base)
        self.n_flows = n_flows
        self.network = network
        self.n_blocks = n_blocks
        self.channels = channels
        self.affine = affine
        self.perm_matrices = torch.nn.ParameterList()
        channels = self.channels
        in_features = self.in_features
        for _ in range(self.n_flows):
            self.layers.append(CouplingBlock2d(in_features, self.network, self.n_blocks, channels, affine=self.affine,
                                               last_block=False))
            self.perm_matrices.append(nn.Parameter(self.build_permutation_matrix(in_features[0]), requires_grad=False))
            in_features = (in_features[0] * 2, in_features[1] // 2, in_features[2] // 2)
            channels *= 2
        self.layers.append(CouplingBlock2d(in_features, self.network, self.n_blocks, channels, affine=self.affine,
                                           last_block=True))
        self._perm_tables = {}

    @staticmethod
    def build_permutation_matrix(channels: int) -> torch.Tensor:
        """The one-hot [4C, C, 2, 2] kernel of the down-scaling convolution: per input channel the 2x2 block in the
        order (0,0), (1,1), (0,1), (1,0), then all first entries, all second entries, ... (reference :141-162)."""
        weights = np.zeros([channels * 4, channels, 2, 2], dtype=np.float32)
        for j, (dy, dx) in enumerate([(0, 0), (1, 1), (0, 1), (1, 0)]):
            for i in range(channels):
                weights[j * channels + i, i, dy, dx] = 1.0
        return torch.tensor(weights, dtype=torch.float32)

    def _perm_table(self, i: int) -> torch.Tensor:
        from deeprob.hip import ops_flows2d
        m = self.perm_matrices[i]
        key = (m.data_ptr(), m._version)
        hit = self._perm_tables.get(i)
        if hit is None or hit[0] != key:
            hit = (key, ops_flows2d.permutation_table(m))
            self._perm_tables[i] = hit
        return hit[1]

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """data -> latent (reference :164-193): after every block but the last the tensor is down-scaled by the
        permutation and its second half of channels is set aside; the pieces are put back in reverse order."""
        from deeprob.hip import ops_flows2d
        ildj, slices = None, []
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x, ildj = layer.transform(x, False, ildj)
            if i != last:
                x, z = ops_flows2d.space_to_depth(x, self._perm_table(i), split=2 * x.shape[1])
                slices.append(z)
        for i in range(last - 1, -1, -1):
            x = ops_flows2d.depth_to_space(x, self._perm_table(i), slices[i])
        return x, ildj

    # The sampling direction of the 2-D path is evaluation only (no backward kernels): rsample() says so with the
    # reference's own NotImplementedError instead of failing inside the first layer.
    has_rsample = False

    def rsample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError("RealNVP2d on the HIP path has no reparametrised sampling (the sampling direction "
                                  "of the 2-D flows is evaluation only); use sample()")

    @torch.no_grad()
    def sample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference flows/models/base.py:145-157.  Called on a model in training mode, the flow is sampled with the
        running statistics of every batch-norm layer (the module is switched to eval for the call and back): the
        reference would normalise the conditioners' activations with the statistics of the samples being drawn and
        move the running averages -- a side effect of sampling that is not reproduced."""
        was_training = self.training
        base_training = bool(getattr(self.in_base, 'training', False))
        if was_training:
            self.train(False, base_training)
        try:
            return super().sample(n_samples, y)
        finally:
            if was_training:
                self.train(True, base_training)

    def apply_forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """latent -> data (reference :195-220)."""
        from deeprob.hip import ops_flows2d
        ops_flows2d.require_eval(self, 'RealNVP2d', x)
        ldj, slices = None, []
        last = len(self.layers) - 1
        for i in range(last):
            x, z = ops_flows2d.space_to_depth(x, self._perm_table(i), split=2 * x.shape[1])
            slices.append(z)
        for i in range(last, -1, -1):
            if i != last:
                x = ops_flows2d.depth_to_space(x, self._perm_table(i), slices[i])
            x, ldj = self.layers[i].transform(x, True, ldj)
        return x, ldj
