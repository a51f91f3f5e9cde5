"""RealNVP / NICE 1-D coupling layer behind the reference interface (deeprob/flows/layers/coupling.py:15-104),
evaluated by one fused fp32-MFMA kernel per call (csrc/coupling.hip), and the 2-D coupling layers / blocks of RealNVP2d
(:107-408; csrc/flows2d.hip, training direction csrc/flows2d_train.hip)."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from deeprob.torch.utils import ScaledTanh
from deeprob.flows.utils import Bijector, BatchNormLayer2d
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


class CouplingLayer2d(Bijector):
    def __init__(self, in_features: Tuple[int, int, int], network: str, n_blocks: int, channels: int,
                 affine: bool = True, channelwise: bool = False, reverse: bool = False):
        """
        RealNVP / NICE 2-D coupling layer (reference :107-153).

        :param in_features: (C, H, W).
        :param network: conditioner, 'resnet' or 'densenet'.
        :param n_blocks: residual / dense blocks of the conditioner.
        :param channels: channels of the conditioner's convolutions.
        :param affine: affine (RealNVP) or translation-only (NICE) transformation.
        :param channelwise: channel-wise mask instead of the checkerboard mask.
        :param reverse: swap the mask and its complement.
        :raises NotImplementedError: for an unknown conditioner.
        """
        super().__init__(in_features)
        self.affine = affine
        self.channelwise = channelwise
        self.reverse = reverse
        if not channelwise:
            mask, inv_mask = self.build_checkerboard_masks()
            if reverse:
                mask, inv_mask = inv_mask, mask
            self.register_buffer('mask', torch.tensor(mask, dtype=torch.float32))
            self.register_buffer('inv_mask', torch.tensor(inv_mask, dtype=torch.float32))
        in_channels = self.in_channels // 2 if channelwise else self.in_channels
        out_channels = in_channels * 2 if affine else in_channels
        if network == 'resnet':
            from deeprob.flows.layers.resnet import ResidualNetwork
            self.network = ResidualNetwork(in_channels, channels, out_channels, n_blocks)
        elif network == 'densenet':
            from deeprob.flows.layers.densenet import DenseNetwork
            self.network = DenseNetwork(in_channels, channels, out_channels, n_blocks)
        else:
            raise NotImplementedError("Unknown network conditioner {}".format(network))
        if affine:
            self.scale_act = ScaledTanh([in_channels, 1, 1])

    @property
    def in_channels(self) -> int:
        return self.in_features[0]

    @property
    def in_height(self) -> int:
        return self.in_features[1]

    @property
    def in_width(self) -> int:
        return self.in_features[2]

    def build_checkerboard_masks(self) -> Tuple[np.ndarray, np.ndarray]:
        """mask[0, h, w] = (h + w) % 2 and its complement (reference :170-179)."""
        mask = np.sum(np.indices([1, self.in_height, self.in_width]), axis=0) % 2
        return mask, 1.0 - mask

    def transform(self, x: torch.Tensor, inverse: bool, ldj: Optional[torch.Tensor] = None):
        """Both directions; `ldj` [B] is an accumulator the layer's log-det-Jacobian is added to."""
        from deeprob.hip import ops_flows2d
        graph = ops_flows2d.graph_route(x, self)
        if graph and inverse:
            ops_flows2d.require_eval(self, 'CouplingLayer2d', x)
        x = ops_flows2d._image(x, 'x')
        if tuple(x.shape[1:]) != tuple(self.in_features):
            raise HipError("CouplingLayer2d: input {} does not match in_features {}".format(tuple(x.shape[1:]),
                                                                                        self.in_features))
        if self.channelwise:
            half = self.in_channels // 2
            mx = x[:, :half] if self.reverse else x[:, half:]
            z = self.network(mx)
        else:
            z = self.network(x, in_mask=self.mask)
        if graph:
            from deeprob.hip import ops_flows2d_train
            u, d = ops_flows2d_train.coupling2d(x, z, self)
            return u, (d if ldj is None else ldj + d)
        return ops_flows2d.coupling2d(x, z, self, inverse, ldj)

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """u = (x - t) exp(-s), ildj = -sum(s), (t, s) from the conditioner on the masked input (reference :181-226)."""
        return self.transform(x, False)

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x = u exp(s) + t, ldj = sum(s) (reference :228-272)."""
        return self.transform(u, True)


class CouplingBlock2d(Bijector):
    def __init__(self, in_features: Tuple[int, int, int], network: str, n_blocks: int, channels: int,
                 affine: bool = True, last_block: bool = False):
        """Three checkerboard couplings, then (unless `last_block`: a fourth checkerboard coupling instead) squeeze,
        three channel-wise couplings with doubled conditioner channels and un-squeeze; a BatchNormLayer2d after every
        coupling (reference :275-354)."""
        super().__init__(in_features)
        self.last_block = last_block
        couplings = []
        for i in range(4 if last_block else 3):
            couplings += [CouplingLayer2d(self.in_features, network, n_blocks, channels, affine, channelwise=False,
                                          reverse=(i % 2 == 1)), BatchNormLayer2d(self.in_channels)]
        self.in_couplings = nn.ModuleList(couplings)
        if not self.last_block:
            squeezed_channels = self.in_channels * 4
            squeezed_features = (squeezed_channels, self.in_height // 2, self.in_width // 2)
            channels *= 2
            couplings = []
            for i in range(3):
                couplings += [CouplingLayer2d(squeezed_features, network, n_blocks, channels, affine, channelwise=True,
                                              reverse=(i % 2 == 1)), BatchNormLayer2d(squeezed_channels)]
            self.out_couplings = nn.ModuleList(couplings)

    @property
    def in_channels(self) -> int:
        return self.in_features[0]

    @property
    def in_height(self) -> int:
        return self.in_features[1]

    @property
    def in_width(self) -> int:
        return self.in_features[2]

    def transform(self, x: torch.Tensor, inverse: bool, ldj: Optional[torch.Tensor] = None):
        from deeprob.hip import ops_flows2d
        if inverse:
            ops_flows2d.require_eval(self, 'CouplingBlock2d', x)
        table = None if self.last_block else ops_flows2d.squeeze_table(self.in_channels, x.device)
        if not inverse:
            for layer in self.in_couplings:
                x, ldj = layer.transform(x, False, ldj)
            if not self.last_block:
                x = ops_flows2d.space_to_depth(x, table)
                for layer in self.out_couplings:
                    x, ldj = layer.transform(x, False, ldj)
                x = ops_flows2d.depth_to_space(x, table)
        else:
            if not self.last_block:
                x = ops_flows2d.space_to_depth(x, table)
                for layer in reversed(self.out_couplings):
                    x, ldj = layer.transform(x, True, ldj)
                x = ops_flows2d.depth_to_space(x, table)
            for layer in reversed(self.in_couplings):
                x, ldj = layer.transform(x, True, ldj)
        return x, ldj

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference :366-387."""
        return self.transform(x, False)

    def apply_forward(self, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference :389-408."""
        return self.transform(u, True)
