"""ctypes binding of ``libdeeprob_hip.so`` (the C ABI declared in ``include/deeprob_hip.h``).

This module is the ONLY place the Python mirror of the DeeProb-kit interface touches native
code.  There is no CPU / PyTorch fallback: if the shared library is missing, or a tensor is not
a contiguous fp32 tensor on a HIP device, the call raises.  PyTorch is used for device memory,
streams and ``torch.distributed`` only.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(  # DEEPROB_HIP_LIB: measurement builds of the same ABI (profiles/README)
    'DEEPROB_HIP_LIB', os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libdeeprob_hip.so')))

DPK_FLAG_STRUCT_CACHED = 1
DPK_FLAG_UNIT_SCALE = 2
DPK_FLAG_PARAMS_CACHED = 4
DPK_FLAG_PARAMS_VERIFY = 8
DPK_FLAG_IN_PIXEL_MAJOR = 16
DPK_FLAG_OUT_PIXEL_MAJOR = 32
DPK_FLAG_LL_SUM_SPREAD = 64
LL_SPREAD = 16            # partial sums in front of the count of a spread {sum LL, count} slot (17 doubles)

# What the operators pass when a module's cached tables were built from parameters whose addresses, shapes and version
# counters are unchanged.  A write through ``param.data`` (hand-written optimisers, clipping, ``.data.copy_`` loaders)
# moves none of those, so by default the belief is CHECKED ON THE DEVICE (DPK_FLAG_PARAMS_VERIFY: one small fingerprint
# launch per table set and call; tables are rebuilt when the bytes differ).  ``trust_version_counters(True)`` restores
# the unchecked fast path (DPK_FLAG_PARAMS_CACHED) for callers that never write through ``.data``.
_trust_versions = False


def trust_version_counters(flag: bool = True) -> bool:
    """Skip the device-side check of cached parameter tables (see above); returns the previous setting."""
    global _trust_versions
    prev, _trust_versions = _trust_versions, bool(flag)
    return prev


def cached_tables_flag() -> int:
    return DPK_FLAG_PARAMS_CACHED if _trust_versions else DPK_FLAG_PARAMS_VERIFY

_c_void = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_u32 = ctypes.c_uint32

# name -> (restype, argtypes); mirrors include/deeprob_hip.h one to one
SIGNATURES = {
    'dpk_last_error': (ctypes.c_char_p, []),
    'dpk_abi_version': (ctypes.c_int, []),
    'dpk_workspace_forget': (ctypes.c_int, [_c_void, _i64]),
    'dpk_ratspn_workspace_bytes': (_i64, [_i32] * 8),
    'dpk_gaussian_leaf_forward_on_mfma': (ctypes.c_int, [_c_void, _c_void, _i32, _i32, _i32, _i32, _u32]),
    'dpk_gaussian_leaf_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void,
                                                 _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_bernoulli_leaf_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                                  _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_gaussian_leaf_backward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                                  _c_void, _i32, _i32, _i32, _c_void, _c_void, _c_void,
                                                  _c_void, _i64, _u32, _c_void]),
    'dpk_bernoulli_leaf_backward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                                   _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_bernoulli_leaf_backward_input': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                                         _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_product_forward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _c_void, _c_void]),
    'dpk_product_backward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _c_void, _c_void]),
    'dpk_sum_forward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _i64,
                                       _c_void]),
    'dpk_sum_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32,
                                        _c_void, _c_void, _c_void, _i64, _c_void]),
    'dpk_sum_workspace_bytes': (_i64, [_i64, _i32, _i32, _i32]),
    'dpk_root_forward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _i32, _c_void, _c_void, _i64,
                                        _c_void]),
    'dpk_root_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _c_void,
                                         _c_void, _c_void, _i64, _c_void]),
    'dpk_ratspn_forward_on_mfma': (ctypes.c_int, [_c_void, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _u32]),
    'dpk_ratspn_small_batch_max': (_i64, [_i64]),
    'dpk_ratspn_slice_batch_min': (_i64, [_i64]),
    'dpk_ratspn_mfma_route': (_i32, [_i32]),
    'dpk_spatial_leaf_fuse_min_k': (_i32, [_i32]),
    'dpk_spatial_level_streams': (_i32, [_i32, _i64, _i32, _i32, _i32, _c_void, _i32, _c_void, _i32]),
    'dpk_upper_tables_pair': (ctypes.c_int, [_c_void, _i32, _i32, _i32, _c_void, _i64, _c_void, _i32, _i32, _i32, _c_void, _i64,
                                             _c_void]),
    'dpk_prodsum_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32, _i32, _c_void,
                                            _c_void, _c_void, _i64, _c_void]),
    'dpk_ratspn_forward_train': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void, _c_void, _c_void,
                                                _i32, _i32, _i32, _i32, _i32, _c_void, _c_void, _c_void, _c_void,
                                                _c_void, _i64, _u32, _c_void]),
    'dpk_ratspn_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void, _c_void,
                                          _c_void, _c_void, _i32, _i32, _i32, _i32, _i32, _c_void, _c_void,
                                          _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_coupling1d_workspace_bytes': (_i64, [_i32] * 4),
    'dpk_coupling1d_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _i32, _i32, _c_void, _c_void,
                                              _c_void, _c_void, _i32, _c_void, _c_void, _c_void, _i32, _i32,
                                              _c_void, _c_void, _i32, _c_void, _i64, _c_void]),
    'dpk_coupling1d_pairs_workspace_bytes': (_i64, [_i32, _i32]),
    'dpk_coupling1d_pairs_logprob': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _c_void, _c_void, _c_void, _c_void, _i32,
                                                    _c_void, _c_void, _c_void, _i32, _c_void, _c_void, _c_void, _c_void,
                                                    _c_void, _c_void, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_coupling1d_pairs_forward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _c_void, _c_void, _c_void, _c_void, _i32,
                                                    _c_void, _c_void, _c_void, _i32, _i32, _c_void, _c_void, _i32,
                                                    _c_void, _i64, _u32, _c_void]),
    'dpk_bn1d_fold': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, ctypes.c_float, _i32, _i32, _c_void,
                                     _c_void, _c_void, _c_void, _c_void, _i32, _c_void]),
    'dpk_affine1d_forward': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _c_void, _c_void]),
    'dpk_logit1d_forward': (ctypes.c_int, [_c_void, _i64, _i32, ctypes.c_float, ctypes.c_float, _i32, _c_void, _c_void,
                                           _c_void]),
    'dpk_conv2d_pack_floats': (_i64, [_i32] * 3),
    'dpk_conv2d_prepare': (ctypes.c_int, [_c_void, _c_void, _i32, _i32, _i32, _c_void, _c_void, _c_void, _c_void,
                                          ctypes.c_float, _c_void, _c_void, _c_void]),
    'dpk_conv2d_forward': (ctypes.c_int, [_c_void, _i64, _i64, _i32, _i32, _i32, _c_void, _i32, _i32, _c_void, _c_void,
                                          _c_void, _c_void, _i64, _c_void, _i64, _c_void]),
    'dpk_coupling2d_transform': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32, _i32, _i32,
                                                _i32, _c_void, _c_void, _c_void, _c_void]),
    'dpk_bn2d_bijector': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _c_void, ctypes.c_float, _i64, _i32, _i32,
                                         _i32, _i32, _c_void, _c_void, _c_void, _c_void]),
    'dpk_space_to_depth': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _i32, _c_void, _c_void]),
    'dpk_depth_to_space': (ctypes.c_int, [_c_void, _i32, _c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _c_void]),
    'dpk_channel_stats': (ctypes.c_int, [_c_void, _i64, _i64, _i32, _i32, _i32, _i32, _c_void, _c_void]),
    'dpk_channel_stats_backward': (ctypes.c_int, [_c_void, _i64, _i64, _i32, _i32, _i32, _c_void, _c_void, _c_void,
                                                  _i32, _c_void, _c_void]),
    'dpk_bn2d_fold_train': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, ctypes.c_float, ctypes.c_float,
                                           _c_void, _c_void, _c_void, _c_void, _c_void]),
    'dpk_bn2d_fold_backward': (ctypes.c_int, [_c_void, _i32, _c_void, _c_void, _c_void, _c_void, _c_void, _c_void]),
    'dpk_channel_affine_forward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _c_void]),
    'dpk_channel_affine_backward': (ctypes.c_int, [_c_void, _i64, _c_void, _i64, _i32, _i32, _i32, _c_void, _i32,
                                                   _c_void, _c_void, _c_void, _c_void]),
    'dpk_conv2d_backward_weight': (ctypes.c_int, [_c_void, _i64, _c_void, _i64, _i32, _i32, _i32, _i32, _i32, _c_void,
                                                  _c_void, _c_void, _c_void]),
    'dpk_coupling2d_transform_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32,
                                                         _i32, _i32, _c_void, _c_void, _c_void, _c_void, _c_void,
                                                         _c_void]),
    'dpk_normal_base_logprob': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _c_void, _c_void, _c_void,
                                               _i64, _i32, _c_void, _c_void]),
    'dpk_spatial_gaussian_forward': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _i32, _i32, _i32,
                                                    _c_void, _c_void]),
    'dpk_spatial_gaussian_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32,
                                                     _i32, _c_void, _c_void, _c_void, _c_void]),
    'dpk_spatial_product_forward': (ctypes.c_int, [_c_void, _i64] + [_i32] * 15 + [_c_void, _c_void]),
    'dpk_spatial_product_backward': (ctypes.c_int, [_c_void, _i64] + [_i32] * 15 + [_c_void, _c_void]),
    'dpk_spatial_sum_workspace_bytes': (_i64, [_i32, _i32, _i32, _i32]),
    'dpk_spatial_sum_forward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _i32, _i32, _i32, _c_void, _c_void,
                                               _i64, _c_void]),
    'dpk_spatial_sum_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32, _i32,
                                                _c_void, _c_void, _c_void, _i64, _c_void]),
    'dpk_spatial_prodsum_forward': (ctypes.c_int, [_c_void, _i64] + [_i32] * 13 + [_c_void, _i32, _c_void, _c_void,
                                                                                  _i64, _u32, _c_void]),
    'dpk_spatial_leaf_prodsum_forward': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64] + [_i32] * 14 + [_c_void, _i32, _c_void,
                                                        _c_void, _i64, _u32, _c_void]),
    'dpk_coupling1d_backward_workspace_bytes': (_i64, [_i64, _i32, _i32, _i32]),
    'dpk_coupling1d_backward': (ctypes.c_int, [_c_void, _i64, _i32] + [_c_void] * 6 + [_i32, _c_void, _i32] +
                                [_c_void] * 9 + [_i64, _c_void]),
    'dpk_coupling1d_mlp_workspace_bytes': (_i64, [_i64, _i32, _c_void, _i32]),
    'dpk_coupling1d_mlp_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _i32, _c_void, _c_void, _c_void,
                                                  _c_void, _i32, _i32, _c_void, _c_void, _c_void, _i64, _c_void]),
    'dpk_coupling1d_mlp_backward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _i32, _c_void, _c_void,
                                                   _c_void, _c_void, _i32, _c_void, _c_void, _c_void, _c_void, _c_void,
                                                   _c_void, _i32, _c_void, _i64, _c_void]),
    'dpk_coupling1d_mlp_backward_inverse': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _i32, _c_void, _c_void,
                                                           _c_void, _c_void, _i32, _c_void, _c_void, _c_void, _c_void,
                                                           _c_void, _c_void, _i32, _c_void, _i64, _c_void]),
    'dpk_bn1d_inverse_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                                 ctypes.c_float, _c_void, _c_void, _c_void, _c_void, _i64, _c_void]),
    'dpk_bn1d_train_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void, ctypes.c_float,
                                              ctypes.c_float, _c_void, _c_void, _c_void, _c_void, _c_void, _i64,
                                              _c_void]),
    'dpk_bn1d_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void,
                                         ctypes.c_float, _i32, _c_void, _c_void, _c_void, _c_void, _i64, _c_void]),
    'dpk_bn1d_local_moments': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void]),
    'dpk_bn1d_sync_forward': (ctypes.c_int, [_c_void, _i64, _i32, _c_void, _c_void, _c_void, _i32, _c_void, _c_void,
                                             ctypes.c_float, ctypes.c_float, _c_void, _c_void, _c_void, _c_void, _c_void,
                                             _i64, _c_void]),
    'dpk_bn1d_backward_sums': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _c_void, _c_void, ctypes.c_float,
                                              _c_void, _c_void]),
    'dpk_bn1d_sync_backward': (ctypes.c_int, [_c_void, _c_void, _i64, _i64, _i32, _c_void, _c_void, _c_void,
                                              ctypes.c_float, _c_void, _c_void, _c_void, _c_void, _c_void, _c_void]),
    'dpk_normal_base_backward': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _c_void, _c_void]),
    'dpk_leaf_forward_dropout': (ctypes.c_int, [_i32, _c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void, _i32, _i32,
                                                _i32, ctypes.c_float, ctypes.c_uint64, _c_void, _c_void]),
    'dpk_leaf_backward_dropout': (ctypes.c_int, [_i32, _c_void, _c_void, _i64, _i32, _c_void, _c_void, _c_void, _c_void,
                                                 _i32, _i32, _i32, ctypes.c_float, ctypes.c_uint64, _c_void, _c_void,
                                                 _c_void, _c_void, _i64, ctypes.c_uint32, _c_void]),
    'dpk_spatial_gaussian_forward_dropout': (ctypes.c_int, [_c_void, _c_void, _c_void, _i64, _i32, _i32, _i32, _i32,
                                                            ctypes.c_float, ctypes.c_uint64, _c_void, _c_void]),
    'dpk_spatial_gaussian_backward_dropout': (ctypes.c_int, [_c_void, _c_void, _c_void, _c_void, _i64, _i32, _i32, _i32,
                                                             _i32, ctypes.c_float, ctypes.c_uint64, _c_void, _c_void,
                                                             _c_void, _c_void]),
    'dpk_dropout_fill': (ctypes.c_int, [_c_void, _i64, ctypes.c_float, ctypes.c_uint64, ctypes.c_float, _c_void,
                                        _c_void]),
    'dpk_spatial_prodroot_workspace_bytes': (_i64, [_i32, _i32, _i32, _i32]),
    'dpk_spatial_prodroot_forward': (ctypes.c_int, [_c_void, _i64] + [_i32] * 13 + [_c_void, _i32, _c_void, _c_void,
                                                                                   _i64, _c_void]),
    'dpk_prodsum_workspace_bytes': (_i64, [_i32, _i32, _i32]),
    'dpk_prodsum_forward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_prodroot_forward': (ctypes.c_int, [_c_void, _c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_ratspn_topdown': (ctypes.c_int, [_i32, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _c_void, _c_void,
                                          _c_void, _c_void, _c_void, _c_void, _c_void, ctypes.c_uint64, _c_void, _c_void, _c_void]),
    'dpk_profile_next_kernel': (ctypes.c_int, [_c_void, _c_void]),
    'dpk_profile_next_kernel_of': (ctypes.c_int, [_c_void, _c_void, _i32]),
    'dpk_ll_accumulate': (ctypes.c_int, [_c_void, _i64, _c_void, _c_void]),
    'dpk_neg_mean_forward': (ctypes.c_int, [_c_void, _i64, _c_void, _c_void]),
    'dpk_neg_mean_backward': (ctypes.c_int, [_c_void, _i64, _c_void, _c_void]),
    'dpk_adam_step': (ctypes.c_int, [_i32, _c_void, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, _i32, _c_void, _c_void, _c_void]),
    'dpk_spatial_prodsum_backward': (ctypes.c_int, [_c_void, _i64] + [_i32] * 13 + [_c_void, _i32, _c_void, _c_void,
                                                                                   _c_void, _c_void, _c_void, _c_void, _i64,
                                                                                   _u32, _c_void]),
    'dpk_spatial_sumprodroot_workspace_bytes': (_i64, [_i32] * 7),
    'dpk_spatial_sumprodroot_workspace_bytes_batch': (_i64, [_i64, _i32, _i32, _i32, _c_void, _i32, _c_void, _i32]),
    'dpk_spatial_sumprodroot_forward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _i32, _c_void, _c_void, _i32, _c_void,
                                                       _c_void, _i32, _c_void, _c_void, _i64, _u32, _c_void]),
    'dpk_spatial_tables': (ctypes.c_int, [_i32, _c_void, _c_void]),
    'dpk_bn1d_fold_many': (ctypes.c_int, [_i32, _c_void, _c_void, _c_void]),
    'dpk_coupling1d_pairs_tables': (ctypes.c_int, [_i32, _c_void, _c_void]),
    'dpk_flat_spn_workspace_bytes': (_i64, [_i64, _i32, _i32]),
    'dpk_flat_spn_forward': (ctypes.c_int, [_c_void, _i64, _i32, _i32, _i32] + [_c_void] * 11 +
                             [_i32, _c_void, _c_void, _c_void, _c_void, _c_void, _i64, _c_void]),
}

_lib = None


class HipError(RuntimeError):
    """Raised when the native library is missing or a C-ABI call reports a failure."""


def load_library() -> ctypes.CDLL:
    """Load ``libdeeprob_hip.so`` (built in-tree by ``__graft_entry__.build()``) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipError(
            "libdeeprob_hip.so not found at {} -- build it with `make -C deeprob-kit_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback".format(LIB_PATH)
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so and the header disagree
        except AttributeError:
            if 'DEEPROB_HIP_LIB' in os.environ:   # measurement builds of an older ABI (A/B runs): what exists is bound
                continue
            raise
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load_library().dpk_last_error()
        raise HipError("{} failed ({}): {}".format(what, rc, msg.decode() if msg else ''))


def is_unsupported(rc: int) -> bool:
    return rc == -4


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    """The kernels take contiguous fp32 HIP tensors.  Anything that is not on a HIP device is an error
    (no CPU path); other floating dtypes / layouts are converted on the device."""
    if not t.is_cuda:
        raise HipError(
            "{} lives on '{}': the deeprob HIP path only evaluates tensors on a HIP device "
            "(there is no CPU fallback)".format(name, t.device)
        )
    if t.dtype != torch.float32:
        if not t.is_floating_point():
            raise HipError("{} must be a floating point tensor, got {}".format(name, t.dtype))
        t = t.float()
    return t.contiguous()


class Workspace:
    """A growable device scratch buffer owned by a module (never shared between streams)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.struct_key = None  # what the cached structure tables were built from
        self.params_key = None  # parameters (addresses, versions) the MFMA route's tables were built from
        self._retired = []      # outgrown buffers: a captured HIP graph may still address them

    def __del__(self):
        # the library keeps a few words per workspace ADDRESS (fingerprint slots, the marginalised-evidence hint): hand
        # them back before the allocator reuses the memory
        try:
            if _lib is not None:
                for t in [self.buf] + list(self._retired):
                    if t is not None and t.is_cuda:
                        _lib.dpk_workspace_forget(t.data_ptr(), t.numel())
        except Exception:   # (interpreter shutdown: modules may be gone)
            pass

    def get(self, n_bytes: int, device: torch.device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < n_bytes or self.buf.device != device:
            if self.buf is not None:
                self._retired.append(self.buf)
            self.buf = torch.empty(max(int(n_bytes), 256), dtype=torch.uint8, device=device)
            self.struct_key = None
            self.params_key = None
        return self.buf
