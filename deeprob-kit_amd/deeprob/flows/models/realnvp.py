"""RealNVP-1D behind the reference interface (deeprob/flows/models/realnvp.py:16-72).  RealNVP2d is out of
scope (conv conditioners, not on the north-star path)."""
from typing import Optional

from deeprob.torch.base import DensityEstimator
from deeprob.flows.utils import BatchNormLayer1d
from deeprob.flows.layers.coupling import CouplingLayer1d
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
