"""Weight initialisers (interface of deeprob/torch/initializers.py:7-31)."""
import torch
from torch import distributions


def dirichlet_(tensor: torch.Tensor, alpha: float = 1.0, log_space: bool = True, dim: int = -1):
    """Fill ``tensor`` in place with symmetric-Dirichlet(alpha) samples along ``dim``.

    Consumes the torch RNG exactly like the reference (one ``Dirichlet.sample`` over the remaining
    dimensions, then a transpose of ``dim`` with the last axis) so seeded models initialise
    identically.
    """
    shape = tensor.shape
    if len(shape) == 0:
        raise ValueError("Singleton tensors are not valid")
    lo, hi = -len(shape), len(shape) - 1
    if dim not in range(lo, hi):
        raise IndexError(
            "Dimension out of range (expected to be in range of [{}, {}], but got {})".format(lo, hi, dim)
        )
    axis = (len(shape) + dim) % len(shape)
    with torch.no_grad():
        prior = distributions.Dirichlet(torch.full([shape[axis]], alpha))
        draws = prior.sample([n for a, n in enumerate(shape) if a != axis])
        if log_space:
            draws = torch.log(draws)
        tensor.copy_(torch.transpose(draws, axis, -1))
