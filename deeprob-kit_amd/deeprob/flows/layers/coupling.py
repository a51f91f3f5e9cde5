"""RealNVP / NICE 1-D coupling layer behind the reference interface (deeprob/flows/layers/coupling.py:15-104),
evaluated by one fused fp32-MFMA kernel per call (csrc/coupling.hip).  CouplingLayer2d / CouplingBlock2d are
out of scope (RealNVP2d only)."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from deeprob.torch.utils import ScaledTanh
from deeprob.flows.utils import Bijector
from deeprob.hip import Workspace, HipError


class CouplingLayer1d(Bijector):
    def __init__(self, in_features: int, depth: int, units: int, affine: bool = True, reverse: bool = False):
        """
        :param in_features: number of variables D.
        :param depth: hidden layers of the conditioner.
        :param units: units per hidden layer.
        :param affine: affine (RealNVP) or translation-only (NICE) transformation.
        :param reverse: swap the alternating mask and its complement.
        """
        super().__init__(in_features)
        self.affine = affine
        self.reverse = reverse
        mask, inv_mask = self.build_alternating_masks()
        if reverse:
            mask, inv_mask = inv_mask, mask
        self.register_buffer('mask', torch.tensor(mask, dtype=torch.float32))
        self.register_buffer('inv_mask', torch.tensor(inv_mask, dtype=torch.float32))

        # conditioner: Linear(D, units) -> ReLU -> ... -> Linear(units, 2D | D)   (reference :45-56)
        stack, width = [], self.in_features
        for _ in range(depth):
            stack += [nn.Linear(width, units), nn.ReLU(inplace=True)]
            width = units
        stack.append(nn.Linear(width, self.in_features * 2 if affine else self.in_features))
        self.network = nn.Sequential(*stack)
        if affine:
            self.scale_act = ScaledTanh()
        self._ws = Workspace()
        self._ws_bwd = Workspace()
        self._ws_pairs = Workspace()   # packed tables of the alternating-mask kernel (dpk_coupling1d_pairs_forward)
        self._counts = None
        self._pairs = None

    def build_alternating_masks(self) -> Tuple[np.ndarray, np.ndarray]:
        """mask = 0,1,0,1,... and its complement (reference :62-70)."""
        mask = np.arange(self.in_features) % 2
        return mask, 1.0 - mask

    def _mask_counts(self) -> Tuple[int, int]:
        """Non-zeros of mask / inv_mask (the kernel only moves those columns); binary masks only."""
        key = (self.mask._version, self.inv_mask._version, self.mask.data_ptr())
        if self._counts is None or self._counts[0] != key:
            m, im = self.mask.detach().cpu(), self.inv_mask.detach().cpu()
            if not bool(((m == 0) | (m == 1)).all() and ((im == 0) | (im == 1)).all()):
                raise HipError("CouplingLayer1d on the HIP path needs binary mask / inv_mask buffers")
            self._counts = (key, (int(m.sum().item()), int(im.sum().item())))
        return self._counts[1]

    def _pair_parity(self) -> Optional[int]:
        """Parity of the conditioning columns when the masks are the reference's alternating ones (mask =
        arange(D) % 2 or its complement, inv_mask = 1 - mask), else None.  Host check, cached per buffer version."""
        key = (self.mask._version, self.inv_mask._version, self.mask.data_ptr())
        if self._pairs is None or self._pairs[0] != key:
            m, im = self.mask.detach().cpu(), self.inv_mask.detach().cpu()
            odd = (torch.arange(m.numel()) % 2).to(m.dtype)
            par = None
            if m.dim() == 1 and m.numel() % 2 == 0 and torch.equal(im, 1 - m):
                if torch.equal(m, odd):
                    par = 1
                elif torch.equal(m, 1 - odd):
                    par = 0
            self._pairs = (key, par)
        return self._pairs[1]

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """u = (x - t) exp(-s), ildj = -sum(s) with (t, s) = conditioner(mask * x) (reference :72-87)."""
        from deeprob.hip import ops_flows
        return ops_flows.coupling1d_autograd(x, self)

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x = u exp(s) + t, ldj = sum(s) (reference :89-104)."""
        from deeprob.hip import ops_flows
        return ops_flows.coupling1d_autograd(u, self, inverse=True)
