"""Weight initialisers (interface of deeprob/torch/initializers.py:7-31)."""
import torch


def dirichlet_(tensor: torch.Tensor, alpha: float = 1.0, log_space: bool = True, dim: int = -1):
    """Fill ``tensor`` in place with symmetric-Dirichlet(alpha) samples along ``dim`` (logs of them when
    ``log_space``).

    The torch RNG is consumed the way the reference consumes it -- a single Dirichlet draw batched over
    the other axes, whose simplex axis (last) is then swapped into place -- so a seeded model starts from
    the same weights.  The accepted ``dim`` range keeps the reference's quirk: the last axis can only be
    named as ``-1``, and swapping (not moving) the axes means the other axes must be symmetric.
    """
    nd = tensor.dim()
    if nd == 0:
        raise ValueError("Singleton tensors are not valid")
    if not -nd <= dim < nd - 1:
        raise IndexError(
            "Dimension out of range (expected to be in range of [{}, {}], but got {})".format(-nd, nd - 1, dim)
        )
    axis = dim % nd
    batch = [n for a, n in enumerate(tensor.shape) if a != axis]
    with torch.no_grad():
        simplex = torch.distributions.Dirichlet(tensor.new_full((tensor.shape[axis],), alpha).float().cpu())
        draws = simplex.sample(batch)
        tensor.copy_((draws.log() if log_space else draws).transpose(axis, -1))
