"""Parity of the RealNVP-2D training direction (csrc/flows2d_train.hip behind deeprob/hip/ops_flows2d_train.py): batch
statistics, log-likelihoods and the gradients of loss = -mean(LL) against vectors the reference produced in training mode
(tools/gen_golden_flows2d.py::_train_fixture) and against autograd over the oracle at other shapes."""
import numpy as np
import pytest
import torch

from oracle import flows2d_oracle as orc
from tests.util import rel_err, FLOWS2D_CASES, flow2d_model, state_checksum
from tests.test_oracle_flows2d import grad_probe, probe_err, TRAIN_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-5        # forward values
TOL_G = 1e-4      # gradients: fp32 atomics / fp64 batch reductions in a different order than ATen's


@pytest.mark.parametrize('case', TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_training_step_golden(golden, case):
    name, feats, kw, seed = case
    g = golden(name + '_train')
    model = flow2d_model(feats, kw, seed).cuda().train()
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    ll = model(x)
    loss = model.loss(ll)
    loss.backward()
    assert rel_err(ll.detach().cpu().numpy(), g['ll']) <= TOL
    assert abs(float(loss.detach()) - float(g['loss'])) <= TOL * abs(float(g['loss']))
    # (the dense-net case normalises over 12 values per channel at its 2x2 scale: ATen itself moves by 2e-4 between
    # thread counts, tests/test_oracle_flows2d.py)
    tol_g = 1e-3 if 'densenet' in name else TOL_G
    assert rel_err(x.grad.cpu().numpy(), g['x_grad']) <= tol_g
    assert float(g['relu_margin']) >= 1e-5        # no ReLU argument of the reference run within rounding of zero
    params = dict(model.named_parameters())
    names = [str(n) for n in g['grad_names']]
    got = np.array([grad_probe(params[n].grad if params[n].grad is not None else torch.zeros_like(params[n]))
                    for n in names])
    assert probe_err(got, g['grad_probe']) <= tol_g, [n for n, a, b in zip(names, got, g['grad_probe'])
                                                      if probe_err(a[None], b[None]) > tol_g]
    # running statistics (BatchNorm2d momentum 0.1 with the unbiased variance, BatchNormLayer2d momentum 0.9 with the
    # biased one) and num_batches_tracked after the step
    np.testing.assert_allclose(state_checksum(model.eval()), g['sd_check_after'], rtol=2e-6, atol=1e-6)


def _oracle_grads(model, x, train):
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    for n in names:
        sd[n].requires_grad_(True)
    xo = x.detach().cpu().clone().requires_grad_(True)
    with orc.relu_margins() as margins:
        if train:
            with orc.training():
                ll = orc.log_prob(sd, xo)
        else:
            ll = orc.log_prob(sd, xo)
    (-torch.mean(ll)).backward()
    grads = {n: (sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in names}
    return ll.detach(), xo.grad, grads, min(margins)


def _margin(model, x, train):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad(), orc.relu_margins() as margins:
        if train:
            with orc.training():
                orc.log_prob(sd, x)
        else:
            orc.log_prob(sd, x)
    return min(margins)


MARGIN = 1e-5     # smallest |ReLU argument| under which two fp32 evaluations may disagree on a gate


@pytest.mark.parametrize('feats,kw,batch,train', [
    ((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=8, network='resnet'), 4, True),
    ((2, 4, 6), dict(n_flows=1, n_blocks=1, channels=5, network='densenet'), 9, True),
    ((2, 10, 10), dict(n_flows=1, n_blocks=1, channels=19, network='resnet'), 3, True),      # odd channel counts
    ((4, 8, 8), dict(n_flows=2, n_blocks=1, channels=6, network='resnet', affine=False), 5, True),
    ((4, 8, 8), dict(n_flows=2, n_blocks=1, channels=6, network='resnet'), 5, False),        # eval mode, gradients wanted
    ((1, 8, 8), dict(n_flows=1, n_blocks=1, channels=8, network='resnet'), 3, True),         # one input channel
])
def test_gradients_against_oracle_strict(feats, kw, batch, train):
    """Networks small enough that a seeded input with every ReLU argument at least MARGIN from zero exists (the first of
    up to 60 draws): every gradient to TOL_G."""
    model = flow2d_model(feats, kw, 77)
    for k in range(60):
        x = torch.randn((batch,) + feats, generator=torch.Generator().manual_seed(5 + k))
        if _margin(model, x, train) >= MARGIN:
            break
    else:
        pytest.fail('no input with a ReLU margin of {}'.format(MARGIN))
    want_ll, want_gx, want_g, margin = _oracle_grads(model, x, train)
    assert margin >= MARGIN
    model.cuda().train(train)
    xc = x.cuda().requires_grad_(True)
    ll = model(xc)
    model.loss(ll).backward()
    assert rel_err(ll.detach().cpu().numpy(), want_ll.numpy()) <= TOL
    assert rel_err(xc.grad.cpu().numpy(), want_gx.numpy()) <= TOL_G
    bad = []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = max(1.0, float(want_g[n].abs().max()))
        if float((got.cpu() - want_g[n]).abs().max()) > TOL_G * scale:
            bad.append((n, float((got.cpu() - want_g[n]).abs().max()), scale))
    assert not bad, bad


@pytest.mark.parametrize('feats,kw,batch', [
    ((1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet'), 12),      # the default MNIST flow
    ((3, 32, 32), dict(n_flows=2, n_blocks=1, channels=12, network='resnet'), 5),
    ((1, 28, 28), dict(n_flows=1, n_blocks=1, channels=24, network='densenet', affine=False), 7),
    ((2, 4, 6), dict(n_flows=1, n_blocks=1, channels=5, network='densenet'), 70),
])
def test_gradients_against_oracle_full_size(feats, kw, batch):
    """Millions of ReLU arguments: some lie within rounding of zero (margin reported by the oracle ~1e-7), where this
    path and ATen's CPU kernels may pick different gates; each such gate moves the gradients by a few 1e-3 (batch
    statistics couple every sample).  LLs to TOL; gradients as vectors to 3e-2 in the 2-norm (TOL_G when the oracle
    reports a margin of MARGIN or more)."""
    model = flow2d_model(feats, kw, 77)
    x = torch.randn((batch,) + feats, generator=torch.Generator().manual_seed(5))
    want_ll, want_gx, want_g, margin = _oracle_grads(model, x, True)
    model.cuda().train()
    xc = x.cuda().requires_grad_(True)
    ll = model(xc)
    model.loss(ll).backward()
    assert rel_err(ll.detach().cpu().numpy(), want_ll.numpy()) <= TOL
    tol = TOL_G if margin >= MARGIN else 3e-2
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))  # noqa: E731
    assert l2(xc.grad.cpu(), want_gx) <= tol
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    got = torch.cat([(model.get_parameter(n).grad if model.get_parameter(n).grad is not None
                      else torch.zeros_like(model.get_parameter(n))).cpu().reshape(-1) for n in names])
    want = torch.cat([want_g[n].reshape(-1) for n in names])
    assert l2(got, want) <= tol


def test_training_lowers_the_loss_and_eval_matches_oracle_afterwards():
    """A few Adam steps in training mode, then the evaluation path on the updated parameters / running statistics."""
    feats, kw = (1, 8, 8), dict(n_flows=1, n_blocks=1, channels=8, network='resnet')
    torch.manual_seed(3)
    from deeprob.flows.models import RealNVP2d
    model = RealNVP2d(feats, **kw).cuda().train()
    x = torch.randn((64,) + feats, generator=torch.Generator().manual_seed(1)).cuda() * 0.5 + 0.2
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = model.loss(model(x))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    model.eval()
    with torch.no_grad():
        ll = model(x)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = orc.log_prob(sd, x.cpu())
    assert rel_err(ll.cpu().numpy(), want.numpy()) <= TOL


def test_sampling_direction_with_gradients_raises():
    from deeprob.hip import HipError
    model = flow2d_model((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=8, network='resnet'), 5).cuda()
    u = torch.randn(2, 3, 8, 8).cuda()
    with pytest.raises(HipError):
        model.apply_forward(u)                      # parameters require grad: the sampling direction has no backward
    with pytest.raises(HipError):
        model.train().apply_forward(u)
    with torch.no_grad():
        model.eval().apply_forward(u)


def test_training_kernels_error_codes():
    from deeprob.hip import load_library, ptr
    lib = load_library()
    x = torch.randn(2, 3, 4, 4).cuda()
    sums = torch.zeros(6, dtype=torch.float64).cuda()
    assert lib.dpk_channel_stats(ptr(x), 10, 2, 3, 4, 4, 1, ptr(sums), None) == -1          # batch stride below C*H*W
    assert lib.dpk_channel_stats(None, 48, 0, 3, 4, 4, 1, None, None) == 0                   # empty batch
    assert lib.dpk_conv2d_backward_weight(ptr(x), 48, ptr(x), 2, 3, 3, 4, 4, 5, None, None, ptr(x), None) == -4
    assert lib.dpk_coupling2d_transform_backward(ptr(x), ptr(x), None, None, 2, 3, 4, 4, 0, 0, ptr(x), None, ptr(x),
                                                 ptr(x), None, None) == -1                   # channel-wise with odd C


def _err(a, b):
    return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))


@pytest.mark.parametrize('cin,cout,hw,ks,pre,mask,res,sliced', [
    (3, 8, (8, 8), 3, False, True, False, False),      # masked first convolution of a checkerboard coupling
    (8, 32, (8, 8), 1, True, False, False, False),     # BatchNorm2d + ReLU folded, matrix-core forward
    (32, 8, (8, 8), 3, True, False, True, True),       # residual addend, channel-slice input
    (24, 40, (4, 4), 1, False, False, False, True),
    (128, 160, (2, 2), 1, True, False, False, False),  # dense-net bottleneck at the 2x2 scale
    (32, 128, (2, 2), 3, False, False, False, False),
    (19, 19, (10, 10), 3, True, False, True, False),   # odd channel counts
    (64, 64, (14, 14), 3, True, False, True, False),   # the squeezed MNIST scale (LDS-staged forward kernel)
])
def test_convolution_node_against_fp64(cin, cout, hw, ks, pre, mask, res, sliced):
    """Conv2dFn (forward, input / weight / bias / operand-map / residual gradients) against the same expression in fp64
    torch operators on the device."""
    import torch.nn.functional as F
    from deeprob.hip import ops_flows2d_train as tr
    B, (H, W) = 5, hw
    g = torch.Generator().manual_seed(cin * 100 + cout)
    big = torch.randn(B, cin + 3, H, W, generator=g).cuda()
    x0 = (big[:, 1:1 + cin] if sliced else big[:, :cin].contiguous()).detach().requires_grad_(True)
    w0 = (0.3 * torch.randn(cout, cin, ks, ks, generator=g)).cuda().requires_grad_(True)
    b0 = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    p0 = (torch.cat([0.5 + torch.rand(cin, generator=g), 0.3 * torch.randn(cin, generator=g)]).cuda().requires_grad_(True)
          if pre else None)
    m0 = ((torch.arange(H)[:, None] + torch.arange(W)[None]) % 2).float().cuda() if mask else None
    r0 = torch.randn(B, cout, H, W, generator=g).cuda().requires_grad_(True) if res else None
    go = torch.randn(B, cout, H, W, generator=g).cuda()
    out = tr.Conv2dFn.apply(x0, w0, b0, p0, None if m0 is None else m0.reshape(-1), r0)
    ins = [t for t in (x0, w0, b0, p0, r0) if t is not None]
    got = torch.autograd.grad(out, ins, go)
    h = x0.double()
    if pre:
        h = torch.relu(p0[:cin].double().view(1, -1, 1, 1) * h + p0[cin:].double().view(1, -1, 1, 1))
    if mask:
        h = h * m0.double()
    ref = F.conv2d(h, w0.double(), b0.double(), padding=ks // 2)
    if res:
        ref = ref + r0.double()
    want = torch.autograd.grad(ref, ins, go.double())
    assert _err(out, ref) <= TOL
    for a, b in zip(got, want):
        assert _err(a, b) <= 2e-5


@pytest.mark.parametrize('C,B,H,W', [(3, 5, 8, 8), (40, 5, 4, 4), (17, 70, 6, 4), (32, 12, 28, 28)])
def test_statistics_and_affine_nodes_against_fp64(C, B, H, W):
    from deeprob.hip import ops_flows2d_train as tr
    torch.manual_seed(C)
    x = (torch.randn(B, C + 2, H, W).cuda() * 0.1 + 3.0)[:, 1:1 + C].detach().requires_grad_(True)   # |mean| >> std
    mean, var = tr.ChannelStatsFn.apply(x)
    gm, gv = torch.randn(C).cuda(), torch.randn(C).cuda()
    got = torch.autograd.grad([mean, var], [x], [gm, gv])[0]
    xd = x.double()
    m2 = xd.mean(dim=[0, 2, 3])
    v2 = ((xd - m2.view(1, -1, 1, 1)) ** 2).mean(dim=[0, 2, 3])
    want = torch.autograd.grad([m2, v2], [x], [gm.double(), gv.double()])[0]
    assert _err(mean, m2) <= 1e-6 and float(((var.double() - v2) / v2).abs().max()) <= 1e-5 and _err(got, want) <= 1e-5
    ab = torch.randn(2 * C).cuda().requires_grad_(True)
    xc = x.detach().contiguous().requires_grad_(True)
    out = tr.ChannelAffineFn.apply(xc, ab)
    go = torch.randn_like(out)
    got = torch.autograd.grad(out, [xc, ab], go)
    ref = ab[:C].view(1, -1, 1, 1).double() * xc.double() + ab[C:].view(1, -1, 1, 1).double()
    want = torch.autograd.grad(ref, [xc, ab], go.double())
    assert _err(out, ref) <= TOL and _err(got[0], want[0]) <= TOL and _err(got[1], want[1]) <= 2e-5
