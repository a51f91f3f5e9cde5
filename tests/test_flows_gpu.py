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
        for B in (1, 63, 64, 65, 200):
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
    flow = flow.cuda().eval()
    x = torch.randn(65536, 784, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    with torch.no_grad():
        ll = flow(x)
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
