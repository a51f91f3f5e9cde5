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


def randomise_flow2d(model, seed):
    """Same as tools/gen_golden_flows2d.py::randomise_flow2d: every parameter / running statistic of a RealNVP2d is a
    pure function of (seed, tensor name), so the fixtures ship a checksum of the state instead of the weights."""
    import zlib

    def gen(name):
        return torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) % (2 ** 31))

    with torch.no_grad():
        for name, p in model.named_parameters():
            g = gen(name)
            if name.endswith('scale_act.weight'):
                p.copy_(0.2 + 0.3 * torch.rand(p.shape, generator=g))
            elif name.endswith('conv.weight_v'):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith('conv.weight_g'):
                p.copy_(0.15 + 0.2 * torch.rand(p.shape, generator=g))
            elif name.endswith('conv.bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif '.network.' in name and name.endswith('.weight'):
                p.copy_(0.8 + 0.4 * torch.rand(p.shape, generator=g))
            elif '.network.' in name and name.endswith('.bias'):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif 'couplings.' in name and (name.endswith('.weight') or name.endswith('.bias')):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            g = gen(name)
            if name.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif name.endswith('running_mean'):
                b.copy_(0.3 * torch.randn(b.shape, generator=g))


def state_checksum(model):
    import numpy as np
    sd = model.state_dict()
    return np.array([float(sd[k].double().abs().sum()) for k in sorted(sd)], dtype=np.float64)


FLOWS2D_CASES = [
    # fixture, in_features, constructor arguments, seed (tools/gen_golden_flows2d.py::CASES)
    ('realnvp2d_3x8x8_resnet', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='resnet', affine=True), 21),
    ('realnvp2d_3x8x8_resnet_nice', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='resnet', affine=False), 22),
    ('realnvp2d_3x8x8_densenet', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='densenet', affine=True), 23),
    ('realnvp2d_3x8x8_densenet_nice', (3, 8, 8), dict(n_flows=2, n_blocks=2, channels=8, network='densenet', affine=False), 24),
    ('realnvp2d_1x28x28_logit', (1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', affine=True, logit=0.05), 25),
    ('realnvp2d_3x12x20_c20', (3, 12, 20), dict(n_flows=1, n_blocks=1, channels=20, network='resnet', affine=True), 26),
]


def flow2d_model(feats, kw, seed):
    from deeprob.flows.models import RealNVP2d
    torch.manual_seed(seed)
    m = RealNVP2d(feats, **kw).eval()
    randomise_flow2d(m, seed)
    return m


def report_measured(name: str, measured: float, bound: float, note: str = ''):
    """Append `measured error vs accepted bound` of a test with a widened tolerance to gpurun_out/measured_errors.txt
    (the file of the round is copied to profiles/): a widened bound is only honest next to the number it admits."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'measured_errors.txt'), 'a') as f:
            f.write('%-70s measured %.3e  bound %.3e  %s\n' % (name, measured, bound, note))
    except OSError:
        pass
