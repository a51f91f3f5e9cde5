"""``log_likelihood`` of a vanilla SPN (reference: deeprob/spn/algorithms/inference.py:37-58) on the HIP evaluator.

There is no CPU path: the inputs are moved to the HIP device (or already live there) and the circuit is walked
by ``dpk_flat_spn_forward``.
"""
from typing import Tuple, Union

import numpy as np
import torch

from deeprob.hip import load_library, check, ptr, stream_ptr, require_device_f32
from deeprob.spn.structure.io import FlatSpn


def log_likelihood(root: FlatSpn, x: Union[np.ndarray, torch.Tensor], return_results: bool = False, n_jobs: int = 0
                   ) -> Union[np.ndarray, torch.Tensor, Tuple]:
    """
    Compute the logarithmic likelihoods of the SPN given some inputs.

    :param root: The SPN (as loaded by ``deeprob.spn.structure.io.load_spn_json``).
    :param x: The inputs ``[B, n_features]``; they can be marginalized using NaNs.  A numpy array is evaluated on
              the current HIP device and numpy arrays come back (as the reference returns); a device tensor
              stays on its device.
    :param return_results: Whether to also return the log likelihoods of each node, ``[n_nodes, B]``.
    :param n_jobs: Accepted for compatibility (the reference's joblib thread count).
    :return: The log likelihood values ``[B]`` float32.  Additionally, the values of each node.
    :raises ValueError: If the SPN is not smooth / decomposable, or the inputs do not cover its scope.
    """
    if not isinstance(root, FlatSpn):
        raise TypeError("log_likelihood evaluates the FlatSpn returned by deeprob.spn.structure.io.load_spn_json")
    root.check()
    lib = load_library()
    as_numpy = not isinstance(x, torch.Tensor)
    if as_numpy:
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.device('cuda', torch.cuda.current_device()))
    xd = require_device_f32(x, 'x')
    if xd.dim() != 2 or xd.shape[1] < root.n_features:
        raise ValueError("expected inputs [B, >= {}], got {}".format(root.n_features, tuple(xd.shape)))
    B, D = xd.shape
    dev = xd.device
    a = root.device_arrays(dev)
    out = torch.empty(B, dtype=torch.float32, device=dev)
    table = torch.empty((root.n_nodes, B), dtype=torch.float32, device=dev) if return_results else None
    ws = None
    if table is None:
        n = lib.dpk_flat_spn_workspace_bytes(B, root.n_nodes, root.n_slots)
        if n < 0:
            check(int(n), 'dpk_flat_spn_workspace_bytes')
        if n > 0:
            ws = torch.empty(int(n), dtype=torch.uint8, device=dev)
    check(lib.dpk_flat_spn_forward(ptr(xd), B, D, root.n_nodes, root.root, ptr(a['order']), ptr(a['kind']),
                                   ptr(a['arg0']), ptr(a['arg1']), ptr(a['arg2']), ptr(a['par0']), ptr(a['par1']),
                                   ptr(a['child_index']), ptr(a['child_weight']), ptr(a['cat_value']),
                                   ptr(a['cat_logp']), root.n_slots, ptr(a['node_slot']), ptr(a['child_slot']),
                                   ptr(out), ptr(table), ptr(ws),
                                   0 if ws is None else ws.numel(), stream_ptr(dev)), 'dpk_flat_spn_forward')
    if as_numpy:
        out = out.cpu().numpy()
        table = None if table is None else table.cpu().numpy()
    return (out, table) if return_results else out
