"""Normalizing-flow base model behind the reference interface (deeprob/flows/models/base.py:13-210)."""
from typing import Optional, Tuple

import torch
from torch import nn
from torch import distributions

from deeprob.torch.base import ProbabilisticModel, DensityEstimator
from deeprob.flows.utils import DequantizeLayer, LogitLayer, BatchNormLayer1d
from deeprob.flows.layers.coupling import CouplingLayer1d


def _sum_log_dets(ds, x: torch.Tensor):
    """The layers' log-det-Jacobian terms added up.  The reference adds them one by one (flows/models/base.py:182-193);
    on the device that is one tiny element-wise launch per layer -- ten of a RealNVP-1D training step's hundred.  Per-sample
    vectors [B] and one-element constants are therefore stacked and summed in ONE reduction (same terms; the additions
    associate differently, 1 ulp); anything else keeps the loop."""
    if x.is_cuda and x.dim() >= 1 and len(ds) >= 3:
        B = x.shape[0]
        tens = [d for d in ds if torch.is_tensor(d)]
        if len(tens) == len(ds) and all(d.is_cuda and d.dtype == tens[0].dtype and (d.numel() == 1 or tuple(d.shape) == (B,))
                                        for d in tens) and any(tuple(d.shape) == (B,) for d in tens):
            return torch.stack([d.reshape(-1).expand(B) if d.numel() == 1 and tuple(d.shape) != (B,) else d for d in tens]).sum(0)
    total = 0.0
    for d in ds:
        total = total + d
    return total


class NormalizingFlow(ProbabilisticModel):
    has_rsample = True

    def __init__(self, in_features, dequantize: bool = False, logit: Optional[float] = None,
                 in_base: Optional[DensityEstimator] = None):
        """
        :param in_features: input size (int or (C, H, W)).
        :param dequantize: prepend the dequantisation transformation.
        :param logit: logit factor, None to disable the logit transformation.
        :param in_base: base density (None = standard Normal with frozen `in_base_loc` / `in_base_scale`).
        :raises ValueError: for an invalid input size or logit factor.
        """
        if isinstance(in_features, torch.Size):
            in_features = tuple(in_features)
            if len(in_features) == 1:
                in_features = in_features[0]
        if not isinstance(in_features, int):
            if not isinstance(in_features, tuple) or len(in_features) != 3:
                raise ValueError("The number of input features must be either an int or a (C, H, W) tuple")
        super().__init__()
        self.in_features = in_features
        self.dequantize = DequantizeLayer(in_features) if dequantize else None
        if logit is not None:
            if logit <= 0.0 or logit >= 1.0:
                raise ValueError("The logit factor must be in (0, 1)")
            self.logit = LogitLayer(in_features, alpha=logit)
        else:
            self.logit = None
        if in_base is None:
            self.in_base_loc = nn.Parameter(torch.zeros(in_features), requires_grad=False)
            self.in_base_scale = nn.Parameter(torch.ones(in_features), requires_grad=False)
            self.in_base = distributions.Normal(self.in_base_loc, self.in_base_scale)
        else:
            self.in_base = in_base
        self.layers = nn.ModuleList()

    def train(self, mode: bool = True, base_mode: bool = True):
        """Training mode of the flow layers and, separately, of a module base density (reference :69-81)."""
        self.training = mode
        self.layers.train(mode)
        if isinstance(self.in_base, torch.nn.Module):
            self.in_base.train(base_mode)
        return self

    def eval(self):
        return self.train(False, False)

    def preprocess(self, x: torch.Tensor):
        """Dequantize / logit in the density direction (reference :91-105)."""
        ildj = 0.0
        if self.dequantize is not None:
            x, d = self.dequantize.apply_backward(x)
            ildj = ildj + d
        if self.logit is not None:
            x, d = self.logit.apply_backward(x)
            ildj = ildj + d
        return x, ildj

    def unpreprocess(self, x: torch.Tensor):
        """Inverse preprocessing (reference :107-121)."""
        ldj = 0.0
        if self.logit is not None:
            x, d = self.logit.apply_forward(x)
            ldj = ldj + d
        if self.dequantize is not None:
            x, d = self.dequantize.apply_forward(x)
            ldj = ldj + d
        return x, ldj

    def _fusable(self) -> bool:
        """Every layer is a 1-D coupling or an eval-mode 1-D batch norm and the base is the default
        Normal: the whole density evaluation chains HIP kernels with the batch norms folded away."""
        if not isinstance(self.in_base, distributions.Normal) or not hasattr(self, 'in_base_loc'):
            return False
        if self.training or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        return all(isinstance(l, (CouplingLayer1d, BatchNormLayer1d)) for l in self.layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Log-likelihood of complete evidence, shape [B] (reference :123-143)."""
        batch_size = x.shape[0]
        x, ildj = self.preprocess(x)
        if self._fusable() and x.dim() == 2 and not (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_fused(x, ildj)
        x, d = self.apply_backward(x)
        ildj = ildj + d
        if isinstance(self.in_base, distributions.Normal) and hasattr(self, 'in_base_loc'):
            from deeprob.hip import ops_flows
            if x.dim() == 2:
                return ops_flows.NormalBaseFn.apply(x, self.in_base_loc, self.in_base_scale) + ildj
            if not ops_flows._wants_graph(x, ildj if torch.is_tensor(ildj) else None):
                # image flows (RealNVP2d): the same kernel on the flattened latent, log-det-Jacobian added in the pass
                acc = ildj.to(torch.float32).contiguous() if torch.is_tensor(ildj) else None
                ll = ops_flows.normal_base_logprob(x.reshape(batch_size, -1), None, self.in_base_loc.reshape(-1),
                                                   self.in_base_scale.reshape(-1), acc, None)
                return ll if acc is not None else ll + ildj
            return ops_flows.NormalBaseFn.apply(x.reshape(batch_size, -1).contiguous(), self.in_base_loc.reshape(-1),
                                                self.in_base_scale.reshape(-1)) + ildj
        base_lls = self.in_base.log_prob(x)
        return torch.sum(base_lls.view(batch_size, -1), dim=1) + ildj

    def _forward_fused(self, x: torch.Tensor, pre_ildj) -> torch.Tensor:
        from deeprob.hip import ops_flows
        # (one batched fold of the batch norms and one verification pass over the couplings' tables for the whole forward:
        # three launches instead of three per layer; the operators below then run their main kernels only)
        ops_flows.flow1d_prepare(self, list(self.layers), x)
        try:
            return self._forward_fused_layers(x, pre_ildj)
        finally:
            ops_flows.flow1d_release()

    def _forward_fused_layers(self, x: torch.Tensor, pre_ildj) -> torch.Tensor:
        from deeprob.hip import ops_flows
        affine, ldj_const, ildj = None, [], None
        if torch.is_tensor(pre_ildj):
            ildj = pre_ildj.to(torch.float32).contiguous().clone()
        # the last coupling, when nothing but batch-norm layers follows it, is evaluated together with their folded
        # affine and the Normal base (ops_flows.coupling1d_logprob): its output is never written, the base pass over it
        # never launched
        layers = list(self.layers)
        last = max((i for i, l in enumerate(layers) if not isinstance(l, BatchNormLayer1d)), default=-1)
        for i, layer in enumerate(layers):
            if isinstance(layer, BatchNormLayer1d):
                affine, ldj_const = ops_flows.bn1d_fold(layer, inverse=False, in_affine=affine, ldj_const=ldj_const)
            elif i == last:
                tail, tail_const = None, list(ldj_const)     # (a copy: the plain route below folds the tail itself)
                for bn in layers[i + 1:]:
                    tail, tail_const = ops_flows.bn1d_fold(bn, inverse=False, in_affine=tail, ldj_const=tail_const)
                ll = ops_flows.coupling1d_logprob(x, layer, affine, ildj, tail, self.in_base_loc, self.in_base_scale,
                                                  ops_flows.sum_constants(self, tail_const))
                if ll is not None:
                    return ll
                x, ildj = ops_flows.coupling1d(x, layer, inverse=False, in_affine=affine, ldj=ildj)
                affine = None
            else:
                x, ildj = ops_flows.coupling1d(x, layer, inverse=False, in_affine=affine, ldj=ildj)
                affine = None
        return ops_flows.normal_base_logprob(x, affine, self.in_base_loc, self.in_base_scale, ildj,
                                             ops_flows.sum_constants(self, ldj_const))

    @torch.no_grad()
    def sample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sample the base, push through the flow and undo the preprocessing (reference :145-157)."""
        shape = [n_samples] if isinstance(self.in_base, distributions.Distribution) else n_samples
        x = self.in_base.sample(shape)
        x, _ = self.apply_forward(x)
        x, _ = self.unpreprocess(x)
        return x

    def rsample(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reparametrised sampling (reference :159-180)."""
        if not self.in_base.has_rsample:
            raise NotImplementedError("Base distribution must support parametrized sampling")
        shape = [n_samples] if isinstance(self.in_base, distributions.Distribution) else n_samples
        x = self.in_base.rsample(shape)
        x, _ = self.apply_forward(x)
        x, _ = self.unpreprocess(x)
        return x

    def apply_backward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """data -> latent through every layer (reference :182-193)."""
        ds = []
        for layer in self.layers:
            x, d = layer.apply_backward(x)
            ds.append(d)
        return x, _sum_log_dets(ds, x)

    def apply_forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """latent -> data through the layers in reverse (reference :195-206)."""
        ldj = 0.0
        for layer in reversed(self.layers):
            x, d = layer.apply_forward(x)
            ldj = ldj + d
        return x, ldj

    def loss(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        from deeprob.hip import ops
        return ops.neg_mean(x)
