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


def randomise_flow(model, seed):
    """Same perturbation as tools/gen_golden_flows.py::_randomise_flow (the 784-variable flow fixtures ship
    seeds instead of 6 MB of weights; the parameters are rebuilt from the same torch RNG stream)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('scale_act.weight'):
                p.fill_(0.3 + 0.4 * torch.rand(1, generator=g).item())
            elif '.network.' in name:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
            elif name.endswith('.weight') or name.endswith('.bias'):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif name.endswith('running_mean'):
                b.copy_(0.5 * torch.randn(b.shape, generator=g))


def randomise_dgc(model, seed):
    """Same as tools/gen_golden_dgcspn.py::_randomise_dgc: the (3,32,32) DGC-SPN fixtures ship seeds, not the
    tens of MB of position-dependent sum weights."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if name.endswith('scale'):
                p.copy_(0.5 + 0.5 * torch.rand(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g))


def dropout_mask(seed: int, shape, p: float):
    """The counter-based dropout decision of the HIP kernels (csrc/common.h::dropout_hit) restated in numpy:
    element idx is dropped iff splitmix64(seed + idx * golden) >> 40 < p * 2^24.  Returns a bool tensor."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = (z >> np.uint64(40)).astype(np.float32)
    return torch.from_numpy((top < np.float32(p) * np.float32(16777216.0)).reshape(shape))
