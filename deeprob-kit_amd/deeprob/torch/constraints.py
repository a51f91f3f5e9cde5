"""Parameter constraints (interface of deeprob/torch/constraints.py:7-27)."""
import torch
from torch import nn


class ScaleClipper(nn.Module):
    def __init__(self, eps: float = 1e-5):
        """Keep ``module.scale`` >= eps.  :raises ValueError: if eps <= 0."""
        if eps <= 0.0:
            raise ValueError("The epsilon value must be positive")
        super().__init__()
        self.register_buffer('eps', torch.tensor(eps))

    def forward(self, module: nn.Module):
        with torch.no_grad():
            module.scale.clamp_(self.eps)
