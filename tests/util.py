import numpy as np
import torch


def rel_err(a, b):
    """max |a-b| / max(1, |b|) elementwise-relative error, inf/nan must match exactly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), 'non-finite pattern differs'
    if (~fin).any():
        assert np.array_equal(a[~fin], b[~fin], equal_nan=True)
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))))


def grad_err(a, b):
    """max |a-b| relative to the largest magnitude of the reference tensor (SURVEY 8c tolerance)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.max(np.abs(b)), 1e-12)
    return float(np.max(np.abs(a - b)) / scale)


def state_to_model(model, npz, device=None):
    sd = {k[3:]: torch.from_numpy(np.asarray(npz[k])) for k in npz.files if k.startswith('sd.')}
    model.load_state_dict(sd)
    if device is not None:
        model.to(device)
    return model
