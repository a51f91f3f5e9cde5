"""Parity of the HIP RealNVP-1D path (fused MFMA coupling, folded batch norm) with reference vectors."""
import numpy as np
import pytest
import torch

from oracle import flows_oracle as forc
from tests.flow_cases import CASES, build_flow
from tests.util import rel_err, grad_err, report_measured

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize('name', sorted(CASES))
def test_log_prob_golden(golden, name):
    g = golden(name)
    model = build_flow(name, g).cuda()
    with torch.no_grad():
        ll = model(torch.from_numpy(g['x']).cuda())
    assert ll.shape == g['ll'].shape
    assert rel_err(ll.cpu().numpy(), g['ll']) <= TOL


@pytest.mark.parametrize('name', sorted(CASES))
def test_layers_and_inverse_golden(golden, name):
    g = golden(name)
    model = build_flow(name, g).cuda()
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        h = x
        for i, layer in enumerate(model.layers):
            h, d = layer.apply_backward(h)
            if 'layer{}.u'.format(i) in g.files:
                assert rel_err(h.cpu().numpy(), g['layer{}.u'.format(i)]) <= TOL, i
                assert rel_err(d.cpu().numpy(), g['layer{}.ildj'.format(i)]) <= TOL, i
        u, ildj = model.apply_backward(x)
        xr, ldj = model.apply_forward(u)
    assert rel_err(u.cpu().numpy(), g['u']) <= TOL
    assert rel_err(ildj.cpu().numpy(), g['ildj']) <= TOL
    assert rel_err(xr.cpu().numpy(), g["x_rec"]) <= 1e-5   # two passes
    assert rel_err(ldj.cpu().numpy(), g["ldj"]) <= 1e-5     # (second pass too: it runs on the reconstructed inputs)


def test_invertibility_like_reference():
    """Reference tests/test_flows.py:22-26,65-75 on the HIP path."""
    from deeprob.flows.models import RealNVP1d
    torch.manual_seed(42)
    x = torch.rand(32, 192, device='cuda')
    for kw in [dict(batch_norm=True, affine=True), dict(batch_norm=False, affine=True),
               dict(batch_norm=True, affine=False)]:
        flow = RealNVP1d(192, **kw).cuda().eval()
        with torch.no_grad():
            u, ildj = flow.apply_backward(x)
            xr, ldj = flow.apply_forward(u)
        assert torch.allclose(ildj, -ldj, atol=5e-7) and torch.allclose(xr, x, atol=5e-7)
    for bad in (dict(n_flows=0), dict(depth=0), dict(units=0)):
        with pytest.raises(ValueError):
            RealNVP1d(10, **bad)


@pytest.mark.parametrize('D,pattern', [(40, 'blocks'), (40, 'random'), (40, 'alternating'), (40, 'reversed'),
                                       (41, 'alternating'), (64, 'sparse')])
def test_custom_masks_vs_oracle(D, pattern):
    """The fused kernel on masks other than the layer's own: half blocks, a random binary split, a mask pair that
    leaves some variables untouched by both (sparse), and the alternating pair at even / odd width (the column-pair
    epilogue covers exactly the even-width alternating masks; everything else takes the generic route)."""
    from deeprob.flows.layers.coupling import CouplingLayer1d
    from oracle import flows_oracle as forc
    gen = torch.Generator().manual_seed(D + len(pattern))
    layer = CouplingLayer1d(D, depth=1, units=32, affine=True).cuda().eval()
    idx = torch.arange(D)
    if pattern == 'blocks':
        mask = (idx < D // 2).float()
        inv = 1 - mask
    elif pattern == 'random':
        mask = (torch.rand(D, generator=gen) < 0.4).float()
        inv = 1 - mask
    elif pattern == 'sparse':
        mask = (idx % 3 == 0).float()
        inv = (idx % 3 == 1).float()            # variables with idx % 3 == 2 are neither input nor transformed
    else:
        mask = (idx % 2).float()
        inv = 1 - mask
        if pattern == 'reversed':
            mask, inv = inv, mask
    with torch.no_grad():
        layer.mask.copy_(mask)
        layer.inv_mask.copy_(inv)
        layer.scale_act.weight.fill_(0.7)
        for p in layer.network.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
    lins = [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in layer.network if isinstance(m, torch.nn.Linear)]
    for B in (3, 70):
        x = torch.randn(B, D, generator=gen)
        with torch.no_grad():
            u, ildj = layer.apply_backward(x.cuda())
            xr, ldj = layer.apply_forward(u)
        wu, wildj = forc.coupling_backward(x, mask, inv, lins, torch.tensor([0.7]))
        assert rel_err(u.cpu().numpy(), wu.numpy()) <= 1e-5
        assert rel_err(ildj.cpu().numpy(), wildj.numpy()) <= 1e-5
        assert torch.allclose(xr.cpu(), x, atol=5e-6) and torch.allclose(ldj, -ildj, atol=5e-6)


@pytest.mark.parametrize('D,units,affine', [(80, 32, True), (128, 64, True), (144, 96, True), (784, 128, True),
                                            (768, 128, False), (784, 32, False), (96, 32, True), (832, 64, True)])
def test_column_pair_kernels_shapes_vs_oracle(D, units, affine):
    """The alternating-mask layer on the f16 matrix cores at the edges of the x-once kernel's envelope (round 3:
    coupling_x1_kernel keeps a 64-sample tile of x in the registers of four waves; D = 64 n or 64 n + 16 with 2 chunks
    to 12 full chunks + tail): one full chunk + tail (80), full chunks only (128, 768), two full chunks + tail (144), the
    benchmark width (784), and the two-pass kernel's shapes next to it (96: a 32-column tail; 832: thirteen full
    chunks); every hidden width, both mask parities, both directions, ragged and single-row batches."""
    from deeprob.flows.layers.coupling import CouplingLayer1d
    from oracle import flows_oracle as forc
    gen = torch.Generator().manual_seed(D + units)
    layer = CouplingLayer1d(D, depth=1, units=units, affine=affine).cuda().eval()
    idx = torch.arange(D)
    for reverse in (False, True):
        mask = (idx % 2).float()
        inv = 1 - mask
        if reverse:
            mask, inv = inv, mask
        with torch.no_grad():
            layer.mask.copy_(mask)
            layer.inv_mask.copy_(inv)
            if affine:
                layer.scale_act.weight.fill_(0.8)
            for p in layer.network.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * (1.0 / D ** 0.5))
        lins = [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in layer.network if isinstance(m, torch.nn.Linear)]
        # (20001 rows at D = 784: 313 tiles on 256 work-groups -- two tiles per work-group, the ring and the holders'
        # last epilogue crossing a tile boundary, and a one-row last tile)
        for B in (1, 63, 64, 65, 200) + ((20001,) if D == 784 and units == 128 else ()):
            x = torch.randn(B, D, generator=gen)
            with torch.no_grad():
                u, ildj = layer.apply_backward(x.cuda())
                xr, ldj = layer.apply_forward(u)
            wu, wildj = forc.coupling_backward(x, mask, inv, lins, torch.tensor([0.8]) if affine else None)
            assert rel_err(u.cpu().numpy(), wu.numpy()) <= 1e-5, (B, reverse)
            assert rel_err(ildj.cpu().numpy(), wildj.numpy()) <= 1e-5, (B, reverse)
            assert torch.allclose(xr.cpu(), x, atol=2e-5) and torch.allclose(ldj, -ildj, atol=2e-5), (B, reverse)


def test_alternating_masks_on_a_view_at_an_odd_storage_offset():
    """A contiguous x whose first element is not 16-byte aligned cannot take the column-pair MFMA kernel; the generic
    kernel it falls through to must START its own log-det (ldj=None), not add into what the abandoned branch
    allocated (round-2 advisor finding)."""
    from deeprob.flows.layers.coupling import CouplingLayer1d
    gen = torch.Generator().manual_seed(7)
    D, B = 40, 33
    layer = CouplingLayer1d(D, depth=1, units=32, affine=True).cuda().eval()
    with torch.no_grad():
        layer.scale_act.weight.fill_(0.6)
        for p in layer.network.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
    lins = [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in layer.network if isinstance(m, torch.nn.Linear)]
    flat = torch.randn(1 + B * D, generator=gen)
    xc = flat[1:1 + B * D].view(B, D)
    flat_d = torch.full((1 + B * D,), float('nan'), device='cuda')
    flat_d[1:] = xc.reshape(-1).cuda()
    xd = flat_d[1:1 + B * D].view(B, D)
    assert xd.is_contiguous() and xd.data_ptr() % 16 != 0
    wu, wildj = forc.coupling_backward(xc, layer.mask.cpu(), layer.inv_mask.cpu(), lins, torch.tensor([0.6]))
    for _ in range(2):   # (a second call: whatever the caching allocator hands out now holds old values)
        with torch.no_grad():
            u, ildj = layer.apply_backward(xd)
        assert rel_err(u.cpu().numpy(), wu.numpy()) <= TOL
        assert rel_err(ildj.cpu().numpy(), wildj.numpy()) <= TOL
        junk = torch.full((B,), 1e6, device='cuda')   # poison the block the next call's ldj may reuse
        del junk


@pytest.mark.parametrize('B', [1, 63, 65, 1000])
def test_ragged_batches_vs_oracle(golden, B):
    g = golden('realnvp1d_15')
    model = build_flow('realnvp1d_15', g)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(B, 15, generator=torch.Generator().manual_seed(B))
    want = forc.flow_log_prob(sd, x).numpy()
    with torch.no_grad():
        got = model.cuda()(x.cuda()).cpu().numpy()
    assert rel_err(got, want) <= TOL


def test_full_size_round_trip():
    """BASELINE config 5 at full batch: 65536 x 784 through 5 couplings + BN; size-independent properties:
    apply_forward(apply_backward(x)) == x and ldj == -ildj, and slice independence."""
    from deeprob.flows.models import RealNVP1d
    from tests.util import randomise_flow
    torch.manual_seed(10)
    flow = RealNVP1d(784)
    randomise_flow(flow, 11)
    flow.eval()
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    flow = flow.cuda()
    x = torch.randn(65536, 784, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    # 2048 rows drawn from the WHOLE batch against the oracle: every tile position of the persistent coupling kernels
    rows = torch.randint(0, 65536, (2048,), generator=torch.Generator().manual_seed(12))
    rows[0], rows[1] = 0, 65535
    with torch.no_grad():
        ll = flow(x)
        want = forc.flow_log_prob(sd, x[rows.cuda()].cpu()).numpy()
    err = rel_err(ll[rows.cuda()].cpu().numpy(), want)
    report_measured('test_full_size_round_trip[RealNVP1d config 5, B=65536] 2048 rows of the whole batch vs oracle', err, TOL)
    assert err <= TOL
    with torch.no_grad():
        u, ildj = flow.apply_backward(x)
        xr, ldj = flow.apply_forward(u)
        part = flow(x[777:777 + 4097])
    assert torch.isfinite(ll).all()
    report_measured('test_full_size_round_trip |x_rec - x| (randomised, ill-conditioned flow)', (xr - x).abs().max().item(), 2e-4,
                    '(reference bar on its own default-initialised flows: 5e-7, held in the next test)')
    report_measured('test_full_size_round_trip |ldj + ildj|', (ldj + ildj).abs().max().item(), 1e-3)
    assert torch.allclose(xr, x, atol=2e-4, rtol=1e-4)
    assert torch.allclose(ldj, -ildj, atol=1e-3, rtol=1e-5)
    assert torch.equal(ll[777:777 + 4097], part)


@pytest.mark.parametrize('name', ['realnvp1d_784_bn_affine', 'realnvp1d_784_nobn_affine', 'realnvp1d_784_bn_nice'])
def test_full_size_round_trip_on_the_golden_flows(golden, name):
    """BASELINE config 5 at full batch (65536 x 784) on the reference-generated 784-feature flows.  The reference's bar
    for its invertibility test (tests/test_flows.py:22-26: atol 5e-7 on 192 features in [0, 1)) is a statement about ITS
    fp32 arithmetic on THAT model; on these 784-feature flows the restated reference arithmetic itself round-trips to a
    few 1e-6, so the bar here is the reference's own round-trip error on the same model and inputs (oracle, first 512
    rows) -- matched on those rows within 2x, and within 4x over the other 65024 -- with 5e-7 as the floor."""
    g = golden(name)
    model = build_flow(name, g)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    x = torch.rand(65536, 784, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    with torch.no_grad():
        u, ildj = model.apply_backward(x)
        xr, ldj = model.apply_forward(u)
        xc = x[:512].cpu()
        uo, io = forc.flow_apply_backward(sd, xc)
        xo, lo = forc.flow_apply_forward(sd, uo)
    ref_x = (xo - xc).abs().max().item()
    ref_l = ((lo + io).abs() / io.abs().clamp_min(1.0)).max().item()
    ex = (xr - x).abs()
    el = (ldj + ildj).abs() / ildj.abs().clamp_min(1.0)
    bx, bl = max(5e-7, 2 * ref_x), max(5e-7, 2 * ref_l)
    report_measured('test_full_size_round_trip_on_the_golden_flows[%s] |x_rec - x|, first 512 rows' % name, ex[:512].max().item(), bx,
                    '(restated reference, same rows: %.2e)' % ref_x)
    report_measured('test_full_size_round_trip_on_the_golden_flows[%s] |x_rec - x|, all 65536 rows' % name, ex.max().item(), 2 * bx)
    report_measured('test_full_size_round_trip_on_the_golden_flows[%s] |ldj + ildj| / max(1, |ildj|), first 512 rows' % name,
                    el[:512].max().item(), bl, '(restated reference, same rows: %.2e)' % ref_l)
    report_measured('test_full_size_round_trip_on_the_golden_flows[%s] |ldj + ildj| / max(1, |ildj|), all rows' % name,
                    el.max().item(), 2 * bl)
    assert ex[:512].max().item() <= bx and ex.max().item() <= 2 * bx
    assert el[:512].max().item() <= bl and el.max().item() <= 2 * bl


# (the six tests below were dropped by accident in an earlier commit of this round and are restored unchanged)
@pytest.mark.parametrize('keep', [True, False], ids=['kept-activations', 'recompute'])
@pytest.mark.parametrize('name', sorted(__import__('tests.flow_cases', fromlist=['TRAIN_CASES']).TRAIN_CASES))
def test_training_route_golden(golden, name, keep, monkeypatch):
    """Autograd through the HIP flow (coupling backward, train-mode batch norm, Normal / RAT-SPN base): LL, loss,
    d/dx, every parameter gradient and the running statistics after the step, against the reference's.  Both
    coupling routes: conditioner activations kept from the forward, or evaluated again in the backward (fused
    forward kernel for the two-Linear conditioner)."""
    from tests.flow_cases import TRAIN_CASES, build_train_flow
    from tests.util import grad_err
    from deeprob.hip import ops_flows
    if not keep:
        monkeypatch.setattr(ops_flows, 'KEEP_ACTIVATIONS_BYTES', 0)
    g = golden(name)
    train = TRAIN_CASES[name][1]
    model = build_train_flow(name, g).cuda()
    model.train(train)
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    ll = model(x)
    loss = model.loss(ll)
    loss.backward()
    assert rel_err(ll.detach().cpu().numpy(), g['ll']) <= 1e-5
    assert rel_err(loss.detach().cpu().numpy(), g['loss']) <= 1e-5
    assert grad_err(x.grad.cpu().numpy(), g['grad.x']) <= 1e-4
    checked = 0
    for k, p in model.named_parameters():
        if 'grad.' + k in g.files:
            assert p.grad is not None, k
            ref = g['grad.' + k]
            if np.max(np.abs(ref)) < 1e-6:
                # mathematically zero (a NICE shift in front of a train-mode batch norm): the reference holds
                # rounding noise only
                assert np.max(np.abs(p.grad.cpu().numpy())) < 1e-6, k
            else:
                assert grad_err(p.grad.cpu().numpy(), ref) <= 1e-4, k
            checked += 1
    assert checked >= 4
    sd = model.state_dict()
    for k in g.files:
        if k.startswith('after.'):
            assert rel_err(sd[k[6:]].cpu().numpy(), g[k]) <= 1e-5, k


def test_second_backward_through_kept_activations():
    """retain_graph: the first backward consumes the kept conditioner output, the second evaluates it again."""
    from deeprob.flows.models import RealNVP1d
    torch.manual_seed(0)
    flow = RealNVP1d(16, n_flows=2, units=16, batch_norm=False).cuda().train()
    x = torch.randn(64, 16).cuda()
    loss = flow.loss(flow(x))
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in flow.parameters() if p.grad is not None]
    flow.zero_grad()
    loss.backward()
    second = [p.grad for p in flow.parameters() if p.grad is not None]
    assert len(first) == len(second) > 0
    for a, b in zip(first, second):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_training_step_reduces_loss():
    """A few Adam steps on the HIP training route (the loop of deeprob/torch/routines.py:117-170 in miniature)."""
    from deeprob.flows.models import RealNVP1d
    torch.manual_seed(0)
    flow = RealNVP1d(32, n_flows=2, units=32).cuda().train()
    data = (torch.randn(256, 32) * 0.5 + 1.0).cuda()
    opt = torch.optim.Adam(flow.parameters(), lr=1e-2)
    losses = []
    for _ in range(15):
        opt.zero_grad()
        loss = flow.loss(flow(data))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 1.0 and all(np.isfinite(losses))


@pytest.mark.parametrize('kw', [dict(in_features=20, n_flows=3, units=32), dict(in_features=15, n_flows=2, units=64, affine=False),
                                dict(in_features=14, n_flows=2, depth=3, units=40, batch_norm=False)])
def test_sampling_direction_gradients_vs_oracle(kw):
    """apply_forward (the direction NormalizingFlow.rsample differentiates, reference flows/models/base.py:159-180,
    coupling.py:89-104, utils.py:141-153) is differentiable: gradients w.r.t. the latent input and every parameter
    against the oracle's autograd in fp64."""
    from deeprob.flows.models import RealNVP1d
    from tests.util import randomise_flow
    torch.manual_seed(21)
    flow = RealNVP1d(**kw)
    randomise_flow(flow, 22)
    flow.eval()
    D = kw['in_features']
    z = torch.randn(33, D, generator=torch.Generator().manual_seed(2))
    wx = torch.randn(33, D, generator=torch.Generator().manual_seed(3))
    wl = torch.randn(33, generator=torch.Generator().manual_seed(4))
    # oracle, fp64
    sd = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in flow.state_dict().items()}
    names = [n for n, _ in flow.named_parameters()]
    for n in names:
        sd[n] = sd[n].clone().requires_grad_(True)
    zo = z.double().requires_grad_(True)
    xo, lo = forc.flow_apply_forward(sd, zo)
    ((xo * wx.double()).sum() + (lo * wl.double()).sum()).backward()
    # product
    flow.cuda()
    zg = z.cuda().requires_grad_(True)
    xg, lg = flow.apply_forward(zg)
    ((xg * wx.cuda()).sum() + (lg * wl.cuda()).sum()).backward()
    assert rel_err(xg.detach().cpu().numpy(), xo.detach().numpy()) <= 2e-5
    assert grad_err(zg.grad.cpu().numpy(), zo.grad.numpy()) <= 1e-4
    for n, p in flow.named_parameters():
        if sd[n].grad is None:
            continue
        assert p.grad is not None, n
        assert grad_err(p.grad.cpu().numpy(), sd[n].grad.numpy()) <= 2e-4, n
    # rsample: gradients reach the flow's parameters
    flow.zero_grad()
    flow.rsample(16).square().mean().backward()
    assert any(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in flow.parameters())


def test_sampling_entry_points():
    """NormalizingFlow.sample / rsample (reference: flows/models/base.py:145-180): base draw pushed through
    apply_forward and the inverse preprocessing; samples must be likely under the flow itself."""
    from deeprob.flows.models import RealNVP1d
    from tests.util import randomise_flow
    torch.manual_seed(3)
    flow = RealNVP1d(24, n_flows=3, units=32, logit=0.05)
    randomise_flow(flow, 4)
    flow = flow.cuda().eval()
    s = flow.sample(500)
    assert tuple(s.shape) == (500, 24) and s.is_cuda and torch.isfinite(s).all()
    lo, hi = -0.05 / 0.9, 0.95 / 0.9                             # inverse logit range: (sigmoid(u) - a) / (1 - 2a)
    assert (s > lo).all() and (s < hi).all()
    with torch.no_grad():
        ll = flow(s)
        far = flow(torch.rand_like(s))
        r = flow.rsample(64)
    assert torch.isfinite(ll).all() and ll.mean().item() > far.mean().item()
    assert tuple(r.shape) == (64, 24)


@pytest.mark.parametrize('D,units,logit', [(24, 32, 0.05), (784, 128, None), (100, 64, 0.1)])
def test_sample_replays_against_the_oracle(D, units, logit):
    """NormalizingFlow.sample with the base draw replayed (VERDICT r05 missing #3): the same torch seed gives the same base
    draw on the device; the oracle pushes THAT draw through the reference's apply_forward + inverse preprocessing
    (flows/models/base.py:145-157) and must land on the samples the HIP path returns."""
    from deeprob.flows.models import RealNVP1d
    from tests.util import randomise_flow
    torch.manual_seed(3)
    flow = RealNVP1d(D, n_flows=3, units=units, logit=logit)
    randomise_flow(flow, 4)
    flow.eval()
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    flow = flow.cuda()
    n = 777
    torch.manual_seed(99)
    shape = [n]
    u = flow.in_base.sample(shape)                 # (the draw sample() is about to make: same generator state)
    torch.manual_seed(99)
    got = flow.sample(n)
    want = forc.flow_sample_from(sd, u.cpu(), logit_alpha=logit)
    assert tuple(got.shape) == (n, D) and torch.isfinite(got).all()
    err = ((got.cpu() - want).abs() / want.abs().clamp_min(1.0)).max().item()
    report_measured('test_sample_replays_against_the_oracle[RealNVP1d %d] max rel |sample - oracle(same base draw)|' % D, err, 1e-5)
    assert err <= 1e-5


@pytest.mark.parametrize('D,units', [(64, 128), (784, 128), (40, 32)])
def test_pairs_kernel_stress_vs_fp64_oracle(D, units):
    """The split-f16 coupling kernel away from the fixtures' comfortable ranges, against the oracle in fp64: conditioner
    weights 10x the reference initialisation (saturating tanh), small weights (subnormal low halves of the split),
    evidence up to |x| ~ 30, BatchNorm variances from 1e-4 to 1e2 folded into the first GEMM."""
    from deeprob.flows.models import RealNVP1d
    from tests.util import randomise_flow
    torch.manual_seed(21)
    model = RealNVP1d(D, n_flows=3, units=units)
    randomise_flow(model, 31)
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if '.network.' in name and name.endswith('weight'):
                p.mul_(torch.where(torch.rand(p.shape, generator=g) < 0.5, 10.0, 1e-3))
        for name, b in model.named_buffers():
            if name.endswith('running_var'):
                b.copy_(10 ** (torch.rand(b.shape, generator=g) * 6 - 4))
    model.eval()
    sd64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    x = torch.randn(257, D, generator=g) * torch.where(torch.rand(257, 1, generator=g) < 0.2, 10.0, 1.0)
    want = forc.flow_log_prob(sd64, x.double()).numpy()
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # The reference's own fp32 arithmetic on this input -- as written, and with the terms of its two GEMMs summed in four
    # other orders (the columns of each Linear permuted together with its input: the same mathematics).  On an input this
    # badly conditioned the fp32 result moves by up to 6x with the summation order (8.1e-5 .. 5.8e-4 across hosts, thread
    # counts and orders, tools/diag_coupling_noise.py: the error IS fp32 accumulation under cancellation; with the GEMMs
    # in fp64 it is 6e-6).  The yardstick is the largest of these draws, not a single one.
    noises = []
    plain_network = forc._network
    try:
        for trial in range(5):
            if trial > 0:
                gp = torch.Generator().manual_seed(100 + trial)

                def permuted(h, lins, gp=gp):
                    for w, b in lins[:-1]:
                        perm = torch.randperm(w.shape[1], generator=gp)
                        h = torch.relu(torch.nn.functional.linear(h[:, perm].contiguous(), w[:, perm].contiguous(), b))
                    w, b = lins[-1]
                    perm = torch.randperm(w.shape[1], generator=gp)
                    return torch.nn.functional.linear(h[:, perm].contiguous(), w[:, perm].contiguous(), b)
                forc._network = permuted
            want32 = forc.flow_log_prob(sd32, x).numpy()
            noises.append(float(np.max(np.abs(want32 - want) / np.maximum(np.abs(want), 1.0))))
    finally:
        forc._network = plain_network
    noise = max(noises)
    with torch.no_grad():
        got = model.cuda()(x.cuda()).cpu().numpy()
    assert np.isfinite(want).all() and np.isfinite(got).all()
    # per sample: the batch mixes log-likelihoods of very different magnitudes
    per_sample = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0))
    # Round 4: two-way f16 splits carry 22 bits per product against fp32's 24, and a layer whose conditioner gain
    # (ops_flows.PAIRS_GAIN_LIMIT) amplifies that rounding is kept on the fp32-MFMA kernel -- the flow as a whole then
    # stays within 2x of what the reference's own fp32 arithmetic loses on this input (round 3: 6.4x measured, 8x allowed).
    from deeprob.hip import ops_flows
    from deeprob.flows.layers.coupling import CouplingLayer1d
    verdicts = [getattr(l._ws_pairs, '_cond', (None, None, None)) for l in model.layers if isinstance(l, CouplingLayer1d)]
    report_measured('test_pairs_kernel_stress_vs_fp64_oracle[%d-%d]' % (D, units), per_sample, max(TOL, 2 * noise),
                    '(TOL 1e-5, or 2x the reference fp32 arithmetic\'s own distance from fp64: %s = as written + 4 summation orders); conditioner gains %s'
                    % (['%.1e' % v for v in noises], ['%.0f%s' % (v[2], '' if v[1] else ' -> fp32 kernel') for v in verdicts if v[2] is not None]))
    assert per_sample <= max(TOL, 2 * noise), (per_sample, noise)
    if D % 8 == 0 and units in (32, 64, 96, 128):
        assert any(v[1] is False for v in verdicts), verdicts        # the stress flow does trip the guard


@pytest.mark.parametrize('target', [250.0, 800.0])
def test_pairs_kernel_between_ordinary_and_the_gain_limit(target):
    """Round 5: the split-f16 column-pair kernel itself at BASELINE config 5's width (D = 784), with
    every layer's conditioner gain set between the ordinary 50-80 and ops_flows.PAIRS_GAIN_LIMIT (1024) -- all layers stay on
    ``coupling_x1_kernel`` -- held per sample to 2x the distance of the reference's as-written fp32 arithmetic from fp64."""
    from deeprob.flows.models import RealNVP1d
    from deeprob.flows.layers.coupling import CouplingLayer1d
    from deeprob.hip import ops_flows
    from tests.util import randomise_flow
    torch.manual_seed(41)
    model = RealNVP1d(784, n_flows=3, units=128)
    randomise_flow(model, 42)
    model = model.cuda().eval()
    g = torch.Generator().manual_seed(43)
    x = torch.randn(257, 784, generator=g) * torch.where(torch.rand(257, 1, generator=g) < 0.2, 3.0, 1.0)
    xd = x.cuda()
    cps = [l for l in model.layers if isinstance(l, CouplingLayer1d)]
    with torch.no_grad():
        for _ in range(2):      # the gain is bilinear in (W1, W2): one correction lands on the target
            model(xd)
            for l in cps:
                f = (target / l._ws_pairs._cond[2]) ** 0.5
                l.network[0].weight.mul_(f)
                l.network[-1].weight.mul_(f)
        got = model(xd).cpu().numpy()
    gains = [l._ws_pairs._cond for l in cps]
    assert all(c[1] and 0.8 * target <= c[2] <= 1.2 * target and c[2] < ops_flows.PAIRS_GAIN_LIMIT for c in gains), gains
    sd32 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    want = forc.flow_log_prob(sd64, x.double()).numpy()
    want32 = forc.flow_log_prob(sd32, x).numpy()
    noise = float(np.max(np.abs(want32 - want) / np.maximum(np.abs(want), 1.0)))
    assert np.isfinite(want).all() and np.isfinite(got).all()
    per_sample = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)))
    report_measured('test_pairs_kernel_between_ordinary_and_the_gain_limit[%d]' % int(target), per_sample, max(TOL, 2 * noise),
                    '(TOL 1e-5, or 2x the as-written reference fp32 distance from fp64 = %.1e); conditioner gains %s, all on the split-f16 kernel'
                    % (noise, ['%.0f' % c[2] for c in gains]))
    assert per_sample <= max(TOL, 2 * noise), (per_sample, noise, gains)


def test_accuracy_guard_leaves_ordinary_flows_on_the_split_f16_kernels():
    """The conditioner-gain guard (ops_flows.PAIRS_GAIN_LIMIT) is a property of the parameters, judged once per parameter
    version: default-initialised and fixture-style randomised flows stay on the column-pair kernels (BASELINE config 5's
    timing is theirs), a second call judges nothing, and a weight update through an optimizer-style in-place op re-judges."""
    from deeprob.flows.models import RealNVP1d
    from deeprob.flows.layers.coupling import CouplingLayer1d
    from deeprob.hip import ops_flows
    from tests.util import randomise_flow
    for randomised in (False, True):
        torch.manual_seed(10)
        flow = RealNVP1d(784)
        if randomised:
            randomise_flow(flow, 11)
        flow = flow.cuda().eval()
        x = torch.randn(256, 784, device='cuda')
        with torch.no_grad():
            flow(x)
        cps = [l for l in flow.layers if isinstance(l, CouplingLayer1d)]
        gains = [l._ws_pairs._cond for l in cps]
        assert all(g[1] and g[2] < 0.2 * ops_flows.PAIRS_GAIN_LIMIT for g in gains), gains
        with torch.no_grad():
            flow(x)
        assert all(l._ws_pairs._cond is g for l, g in zip(cps, gains))          # same verdict objects: nothing re-judged
        with torch.no_grad():
            cps[1].network[0].weight.mul_(40.0)                                     # a (visible) in-place update
            got = flow(x).cpu().numpy()
        assert cps[1]._ws_pairs._cond[1] is False and cps[0]._ws_pairs._cond is gains[0]
        sd = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
        want = forc.flow_log_prob(sd, x.cpu()).numpy()
        assert rel_err(got, want) <= 1e-4          # (an ill-conditioned flow: the reference's own fp32 noise is ~1e-5 here)
