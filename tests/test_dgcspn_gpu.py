"""Parity of the HIP DGC-SPN path (through the C ABI) with the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import dgcspn_oracle as dorc
from tests.dgc_cases import CASES, SMALL, build_dgc, plan_of
from tests.util import rel_err, grad_err, report_measured

pytestmark = pytest.mark.gpu

LL_TOL = 1e-5     # north-star: 1e-5 relative on fp32 log-likelihoods
GRAD_TOL = 1e-4   # relative to the largest magnitude of the tensor


@pytest.mark.parametrize('name', sorted(CASES))
def test_forward_golden(golden, name):
    g = golden(name)
    model = build_dgc(name, g).cuda()
    with torch.no_grad():
        ll = model(torch.from_numpy(g['x']).cuda())
        ll_nan = model(torch.from_numpy(g['x_nan']).cuda())
    assert ll.shape == g['ll'].shape and ll.dtype == torch.float32
    assert rel_err(ll.cpu().numpy(), g['ll']) <= LL_TOL
    assert rel_err(ll_nan.cpu().numpy(), g['ll_nan']) <= LL_TOL


@pytest.mark.parametrize('name', sorted(SMALL))
def test_layers_golden(golden, name):
    g = golden(name)
    model = build_dgc(name, g).cuda()
    with torch.no_grad():
        h = model.base_layer(torch.from_numpy(g['x']).cuda())
        assert rel_err(h.cpu().numpy(), g['act.leaf']) <= LL_TOL
        for i, layer in enumerate(model.layers):
            h = layer(h)
            assert tuple(h.shape[1:]) == tuple(layer.out_features)
            assert rel_err(h.cpu().numpy(), g['act.layer{}'.format(i)]) <= LL_TOL, i


def _fp64_oracle(name, g, model):
    """fp64 restatement on the CPU: measures the reference's own fp32 rounding noise in the golden gradients /
    MPE completions (they involve cancellations the forward pass does not), which sets the tolerance."""
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in
            model.state_dict().items()}
    return sd64, plan_of(name)


@pytest.mark.parametrize('name', sorted(CASES))
def test_mpe_golden(golden, name):
    g = golden(name)
    model = build_dgc(name, g).cuda()
    xn = torch.from_numpy(g['x_nan'])
    with torch.enable_grad():
        mpe = model.mpe(xn.cuda()).cpu().numpy()
    sd64, plan = _fp64_oracle(name, g, model)
    mpe64 = dorc.dgcspn_mpe(sd64, xn.double(), plan).numpy()
    noise = float(np.max(np.abs(g['mpe'] - mpe64)))
    assert np.array_equal(np.isnan(g['x_nan']) | (mpe == g['x_nan']), np.ones_like(mpe, dtype=bool))
    assert float(np.max(np.abs(mpe - mpe64))) <= max(1e-5, 4 * noise)


@pytest.mark.parametrize('name', sorted(CASES))
def test_gradients_golden(golden, name):
    g = golden(name)
    model = build_dgc(name, g).cuda()
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    loss = model.loss(model(x), y.cuda() if y is not None else None)
    loss.backward()
    assert rel_err(loss.detach().cpu().numpy(), g['loss']) <= LL_TOL
    # fp64 gradients: tolerance = max(GRAD_TOL, 4 x the reference's own fp32 error)
    sd64, plan = _fp64_oracle(name, g, model)
    leaves = {k: v.requires_grad_(True) for k, v in sd64.items() if v.is_floating_point()}
    x64 = torch.from_numpy(g['x']).double().requires_grad_(True)
    dorc.dgcspn_loss(dorc.dgcspn_forward(leaves, x64, plan), y).backward()

    def check(got, key, exact):
        noise = grad_err(g[key], exact)
        assert grad_err(got, exact) <= max(GRAD_TOL, 4 * noise), (key, noise)

    check(x.grad.cpu().numpy(), 'grad.x', x64.grad.numpy())
    checked = 0
    for k, p in model.named_parameters():
        if 'grad.' + k in g.files:
            check(p.grad.cpu().numpy(), 'grad.' + k, leaves[k].grad.numpy())
            checked += 1
    assert checked >= 2


@pytest.mark.parametrize('name', ['dgcspn_1x28x28_dw', 'dgcspn_1x12x12_mixed_pool2', 'dgcspn_3x32x32_pool2_dw'])
def test_fused_levels_match_layer_chain(golden, name):
    """DgcSpn.forward without a graph folds depthwise products into the sums; with a graph it chains the
    layer operators.  Same numbers either way (the fused kernel does the same arithmetic per pixel)."""
    g = golden(name)
    model = build_dgc(name, g).cuda()
    x = torch.from_numpy(g['x_nan']).cuda()
    with torch.no_grad():
        fused = model(x)
    chained = model(x).detach()          # parameters require grad -> layer chain
    assert rel_err(fused.cpu().numpy(), chained.cpu().numpy()) <= 1e-6
    # B not a multiple of the samples-per-thread blocking
    with torch.no_grad():
        assert torch.equal(model(x[:3]), fused[:3])


def test_product_invariants_like_reference():
    """Reference tests/test_dgcspn.py:46-75 on the HIP layers."""
    from deeprob.spn.layers.dgcspn import SpatialProductLayer
    ones = torch.ones(8, 3, 32, 32, device='cuda')
    p = SpatialProductLayer((3, 32, 32), kernel_size=2, padding='full', stride=1, dilation=4, depthwise=True)
    assert p.pad == [4, 4, 4, 4] and p.out_features == (3, 36, 36)
    assert torch.allclose(p(ones)[:, :, 4:-4, 4:-4], torch.tensor(4.0, device='cuda'))
    p = SpatialProductLayer((3, 32, 32), kernel_size=2, padding='valid', stride=2, dilation=1, depthwise=True)
    assert p.out_features == (3, 16, 16) and torch.allclose(p(ones), torch.tensor(4.0, device='cuda'))
    p = SpatialProductLayer((3, 32, 32), kernel_size=2, padding='full', stride=1, dilation=8, depthwise=False)
    assert tuple(p.weight.shape) == (81, 3, 2, 2) and p.out_features == (81, 40, 40)
    out = p(ones)
    assert tuple(out.shape) == (8, 81, 40, 40)
    assert torch.allclose(out[:, :, 8:-8, 8:-8], torch.tensor(4.0, device='cuda'))


@pytest.mark.parametrize('n_pooling,depthwise', [(0, False), (2, False), (0, True), (2, True)])
def test_mpe_log_prob_like_reference(n_pooling, depthwise):
    """Reference tests/test_dgcspn.py:89-96."""
    from deeprob.spn.models import DgcSpn
    data = torch.randn(8, 3, 32, 32)
    mar = data.clone()
    mar[torch.rand_like(mar) < 0.5] = np.nan
    model = DgcSpn((3, 32, 32), n_batch=4, sum_channels=4, n_pooling=n_pooling, depthwise=depthwise).cuda()
    lls = model.log_prob(data.cuda())
    with torch.enable_grad():
        mpe_data = model.mpe(mar.cuda())
    mpe_lls = model.log_prob(mpe_data)
    assert torch.all(mpe_lls.squeeze() > lls.squeeze())


def test_random_shapes_against_oracle():
    """Layer-level sweep over geometries the models do not reach (3x3 windows are out of the ABI's scope:
    the reference model only builds 2x2), odd sizes, stride 2, -inf inputs to the sum layer."""
    from deeprob.spn.layers.dgcspn import SpatialGaussianLayer, SpatialProductLayer, SpatialSumLayer
    gen = torch.Generator().manual_seed(5)
    for (c, h, w), padding, stride, dil, dw in [((5, 7, 9), 'full', 1, 3, True), ((2, 9, 9), 'valid', 2, 1, False),
                                                ((3, 6, 6), 'final', 1, 4, True), ((3, 11, 5), 'full', 1, 2, False)]:
        if padding == 'final' and h != w:
            continue
        x = torch.randn(5, c, h, w, generator=gen)
        layer = SpatialProductLayer((c, h, w), 2, padding, stride, dil, depthwise=dw).cuda()
        want = dorc.spatial_product(x, layer.pad, stride, dil, dw)
        got = layer(x.cuda())
        assert rel_err(got.cpu().numpy(), want.numpy()) <= 1e-6
    x = torch.randn(9, 6, 5, 5, generator=gen) * 20
    x[0] = float('-inf')
    x[1, :3] = float('-inf')
    s = SpatialSumLayer((6, 5, 5), 4).cuda()
    with torch.no_grad():
        s.weight[0, 1] = -200.0
        s.weight[0, 1, 2, 2] = 50.0
        got = s(x.cuda())
    want = dorc.spatial_sum(x, s.weight.detach().cpu())
    assert rel_err(got.cpu().numpy(), want.numpy()) <= LL_TOL
    xg = torch.randn(4, 2, 5, 5, generator=gen)
    xg[0, 0, 1, 1] = float('nan')
    xg[1, 1, 2, 2] = float('inf')
    leaf = SpatialGaussianLayer((2, 5, 5), 3, optimize_scale=True).cuda()
    with torch.no_grad():
        got = leaf(xg.cuda())
    want = dorc.spatial_gaussian(xg, leaf.loc.detach().cpu(), leaf.scale.detach().cpu())
    assert rel_err(got.cpu().numpy(), want.numpy()) <= LL_TOL


@pytest.mark.parametrize('cin,cout', [(8, 8), (6, 4), (3, 7), (12, 5)])
def test_sum_backward_extreme_weights_against_oracle(cin, cout):
    """SpatialSumLayer backward (register kernel for <= 8 channels, generic kernel above) against fp64 autograd of
    the oracle: ordinary inputs, a dominant input under a vanishing weight (the exact log-domain branch), batch
    sizes that leave ragged sample slices."""
    from deeprob.spn.layers.dgcspn import SpatialSumLayer
    gen = torch.Generator().manual_seed(11)
    for B in (1, 37):
        x = torch.randn(B, cin, 6, 6, generator=gen) * 3
        x[:, 1] += 100.0                                   # channel 1 dominates every pixel ...
        s = SpatialSumLayer((cin, 6, 6), cout).cuda()
        with torch.no_grad():
            s.weight[0, 1] = -200.0                        # ... and output 0 all but ignores it
            s.weight[1, 1, 2, :] = -95.0
        go = torch.randn(B, cout, 6, 6, generator=gen)
        xd = x.cuda().requires_grad_(True)
        s(xd).backward(go.cuda())
        x64 = x.double().requires_grad_(True)
        w64 = s.weight.detach().cpu().double().requires_grad_(True)
        dorc.spatial_sum(x64, w64).backward(go.double())
        assert grad_err(xd.grad.cpu().numpy(), x64.grad.numpy()) <= GRAD_TOL
        assert grad_err(s.weight.grad.cpu().numpy(), w64.grad.numpy()) <= GRAD_TOL


@pytest.mark.parametrize('shape,padding,stride,dil,cout,B', [((16, 14, 14), 'full', 1, 1, 32, 300), ((32, 7, 7), 'full', 1, 2, 32, 131),
                                                           ((32, 9, 9), 'valid', 2, 1, 20, 67), ((16, 6, 6), 'final', 1, 4, 32, 45),
                                                           ((16, 12, 12), 'valid', 2, 1, 32, 77), ((32, 8, 8), 'valid', 2, 1, 32, 33)])
def test_wide_fused_level_against_oracle(shape, padding, stride, dil, cout, B):
    """Round 5: the eval-mode fused level for 16 / 32 input channels (csrc/dgcspn.hip: spatial_prodsum_wide_kernel, the
    tile's weights in LDS) against the oracle's product + sum: maps that are not a multiple of the 16-pixel tile, fewer
    than 32 sum channels, a batch that is not a multiple of the sample slots, a sample of log 0, marginalised (log 1)
    inputs, and a weight row whose dominant entry sits on a vanishing input (the exact log-domain form)."""
    from deeprob.spn.layers.dgcspn import SpatialProductLayer, SpatialSumLayer
    from deeprob.hip import ops_spatial
    gen = torch.Generator().manual_seed(19)
    prod = SpatialProductLayer(shape, 2, padding, stride, dil, depthwise=True).cuda()
    ssum = SpatialSumLayer(prod.out_features, cout).cuda()
    with torch.no_grad():
        ssum.weight.copy_(torch.randn(ssum.weight.shape, generator=gen) * 2)
        ssum.weight[0, 1] = -300.0                          # softmax weight ~ e^-300 everywhere but ...
        ssum.weight[0, 1, 1, 1] = 80.0                      # ... one pixel, where channel 1 dominates
    x = torch.randn(B, *shape, generator=gen) * 3
    x[0] = float('-inf')
    x[1] = 0.0
    x[2, 1] = -400.0                                        # the dominant channel's input vanishes in the exp domain
    want = dorc.spatial_sum(dorc.spatial_product(x, prod.pad, stride, dil, True), ssum.weight.detach().cpu())
    with torch.no_grad():
        got = ops_spatial.spatial_prodsum(x.cuda(), prod, ssum.weight, ssum._ws)
        again = ops_spatial.spatial_prodsum(x.cuda()[5:40], prod, ssum.weight, ssum._ws)
    assert got is not None
    fin = torch.isfinite(want)
    assert torch.equal(fin, torch.isfinite(got.cpu()))
    assert rel_err(got.cpu()[fin].numpy(), want[fin].numpy()) <= LL_TOL
    assert torch.equal(again, got[5:40])                    # a sample's result does not depend on its place in the batch


@pytest.mark.parametrize('shape,k,cout,B,padding,stride,dil', [
    ((1, 28, 28), 16, 32, 150, 'valid', 2, 1), ((3, 12, 12), 16, 20, 67, 'valid', 2, 1), ((2, 8, 8), 32, 32, 33, 'valid', 2, 1),
    ((1, 28, 28), 8, 8, 150, 'full', 1, 1), ((3, 9, 9), 16, 12, 41, 'full', 1, 2), ((1, 11, 11), 32, 32, 37, 'valid', 2, 1),
    ((2, 7, 7), 8, 5, 300, 'final', 1, 4)])
def test_fused_leaf_and_first_level_against_oracle(shape, k, cout, B, padding, stride, dil, request):
    """Round 5: the Gaussian leaf layer folded into the first level of the eval route (the pooling form and the general one)
    (dpk_spatial_leaf_prodsum_forward: the [B, K, H, W] leaf map is never written) against the oracle's leaf + product +
    sum: several image channels, marginalised (NaN) pixels and a whole marginalised image, scales away from 1, a batch
    that is not a multiple of the sample slots; the result of a sample does not depend on its place in the batch."""
    from deeprob.spn.layers.dgcspn import SpatialGaussianLayer, SpatialProductLayer, SpatialSumLayer
    from deeprob.hip import ops_spatial
    from deeprob.hip import load_library
    prev_k = load_library().dpk_spatial_leaf_fuse_min_k(8)   # (by default 8-channel models keep leaf kernel + streaming level: faster)
    request.addfinalizer(lambda: load_library().dpk_spatial_leaf_fuse_min_k(prev_k))
    gen = torch.Generator().manual_seed(29)
    leaf = SpatialGaussianLayer(shape, k, optimize_scale=True).cuda()
    prod = SpatialProductLayer((k,) + tuple(shape[1:]), 2, padding, stride, dil, depthwise=True).cuda()
    ssum = SpatialSumLayer(prod.out_features, cout).cuda()
    with torch.no_grad():
        leaf.scale.copy_(0.3 + 2.0 * torch.rand(leaf.scale.shape, generator=gen))
        leaf.loc.copy_(torch.randn(leaf.loc.shape, generator=gen))
        ssum.weight.copy_(torch.randn(ssum.weight.shape, generator=gen) * 2)
    x = torch.randn(B, *shape, generator=gen) * 1.5
    x[0] = float('nan')
    x[1, :, ::2, 1::3] = float('nan')
    x[2] = 40.0                                             # log-densities of -800 and below
    lv = dorc.spatial_gaussian(x, leaf.loc.detach().cpu(), leaf.scale.detach().cpu())
    want = dorc.spatial_sum(dorc.spatial_product(lv, prod.pad, stride, dil, True), ssum.weight.detach().cpu())
    with torch.no_grad():
        got = ops_spatial.spatial_leaf_prodsum(x.cuda(), leaf, prod, ssum.weight, ssum._ws)
        assert got is not None                              # (2 x 2 windows, 8 / 16 / 32 leaf channels: inside the envelope)
        again = ops_spatial.spatial_leaf_prodsum(x.cuda()[3:29], leaf, prod, ssum.weight, ssum._ws)
        chained = ops_spatial.spatial_prodsum(leaf(x.cuda()), prod, ssum.weight, ssum._ws)
    fin = torch.isfinite(want)
    assert torch.equal(fin, torch.isfinite(got.cpu()))
    assert rel_err(got.cpu()[fin].numpy(), want[fin].numpy()) <= LL_TOL
    assert rel_err(got.cpu()[fin].numpy(), chained.cpu()[fin].numpy()) <= 1e-6
    assert torch.equal(again, got[3:29])


def test_empty_batch_and_errors():
    from deeprob.hip import HipError
    from deeprob.spn.models import DgcSpn
    model = DgcSpn((1, 8, 8), n_batch=2, sum_channels=2, depthwise=True).cuda()
    with torch.no_grad():
        assert tuple(model(torch.empty(0, 1, 8, 8, device='cuda')).shape) == (0, 1)
        with pytest.raises((ValueError, HipError)):
            model(torch.zeros(2, 1, 9, 8, device='cuda'))
        with pytest.raises((HipError, TypeError, ValueError)):
            model(torch.zeros(2, 1, 8, 8))       # CPU tensor: no silent fallback


def test_full_size_properties():
    """BASELINE config 4 at full batch (8192 images 28x28): size-independent properties -- any slice of the batch
    served by the same kernels gives the same LLs (bit for bit), fully marginalised images give LL = 0, marginalising pixels changes nothing
    for the other samples."""
    from deeprob.spn.models import DgcSpn
    torch.manual_seed(5)
    model = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    plan = dorc.schedule((1, 28, 28), 8, 8, True, 0)
    model.cuda()
    x = torch.randn(8192, 1, 28, 28, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    x[777] = float('nan')
    # 2048 rows drawn from the WHOLE batch (every slice of the streaming kernels' batch split) against the oracle
    rows = torch.randint(0, 8192, (2048,), generator=torch.Generator().manual_seed(9))
    rows[0], rows[1], rows[2] = 0, 8191, 777
    with torch.no_grad():
        ll = model(x)
        xr = x[rows.cuda()].cpu()
        want = np.concatenate([dorc.dgcspn_forward(sd, xr[i:i + 256], plan).numpy() for i in range(0, 2048, 256)])
    err = rel_err(ll[rows.cuda()].cpu().numpy(), want)
    report_measured('test_full_size_properties[DgcSpn config 4, B=8192] 2048 rows of the whole batch vs oracle', err, LL_TOL)
    assert err <= LL_TOL
    with torch.no_grad():
        part = model(x[1001:1001 + 2048])
        small = model(x[1001:1001 + 200])    # below the streaming kernels' batch threshold: the other route
        x2 = x.clone()
        x2[::2, :, :, 14:] = float('nan')
        ll2 = model(x2)
    assert tuple(ll.shape) == (8192, 1) and torch.isfinite(ll).all()
    assert torch.equal(ll[1001:1001 + 2048], part)
    # the two routes of the sum levels differ in summation order only (tolerance of the path: 1e-5 relative)
    assert ((ll[1001:1001 + 200] - small).abs() / ll[1001:1001 + 200].abs().clamp_min(1.0)).max().item() < 2e-6
    assert abs(ll[777].item()) < 1e-5
    assert torch.equal(ll2[1::2], ll[1::2])
    # marginalising half of an image removes non-positive-on-average terms: not an identity, but the result must
    # stay finite and differ
    assert torch.isfinite(ll2).all() and not torch.equal(ll2[0], ll[0])


def test_full_size_properties_example_model():
    """SURVEY 8d config 4's secondary model (examples/dgcspn_mnist.py:27-37: two pooling levels, 16 leaf / 32 sum channels;
    the generic fused product+sum level kernels) at the full batch: a scattered subset against the oracle, slices of the
    batch bit for bit, a fully marginalised image gives LL = 0 and marginalised pixels leave the other samples alone."""
    from deeprob.spn.models import DgcSpn
    torch.manual_seed(6)
    model = DgcSpn((1, 28, 28), n_batch=16, sum_channels=32, depthwise=True, n_pooling=2).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    plan = dorc.schedule((1, 28, 28), 16, 32, True, 2)
    model.cuda()
    x = torch.randn(8192, 1, 28, 28, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    x[555] = float('nan')
    rows = torch.randint(0, 8192, (48,), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ll = model(x)
        part = model(x[3000:3000 + 1024])
        x2 = x.clone()
        x2[::2, :, 10:, :] = float('nan')
        ll2 = model(x2)
    want = dorc.dgcspn_forward(sd, x[rows.cuda()].cpu(), plan).numpy()
    assert tuple(ll.shape) == (8192, 1) and torch.isfinite(ll).all()
    assert rel_err(ll[rows.cuda()].cpu().numpy(), want) <= LL_TOL
    assert torch.equal(ll[3000:3000 + 1024], part)
    assert abs(ll[555].item()) < 1e-5
    assert torch.equal(ll2[1::2], ll[1::2]) and not torch.equal(ll2[0], ll[0])


def test_streaming_levels_golden(golden, monkeypatch):
    """The streaming kernels of the 8 -> 8 channel levels (dgcspn_stream.hip; default route from B = 256) forced onto
    the golden batch of BASELINE config 4's model: same tolerance as the batch-independent route."""
    monkeypatch.setenv('DPK_DGC_STREAM_MIN_B', '0')
    g = golden('dgcspn_1x28x28_dw')
    model = build_dgc('dgcspn_1x28x28_dw', g).cuda()
    with torch.no_grad():
        ll = model(torch.from_numpy(g['x']).cuda())
        ll_nan = model(torch.from_numpy(g['x_nan']).cuda())
    assert rel_err(ll.cpu().numpy(), g['ll']) <= LL_TOL
    assert rel_err(ll_nan.cpu().numpy(), g['ll_nan']) <= LL_TOL


@pytest.mark.parametrize('shape,classes,B,pooling', [((1, 20, 20), 3, 37, 0), ((2, 16, 16), 1, 130, 0),
                                                     ((1, 28, 28), 10, 65, 0), ((1, 16, 16), 1, 33, 2),
                                                     ((3, 32, 32), 2, 19, 1)])
def test_streaming_levels_against_oracle(monkeypatch, shape, classes, B, pooling):
    """Streaming route against the oracle on maps and class counts the golden fixtures do not hold: ragged batch slices,
    tiles that split rows, several root classes, stride-2 (pooling) levels, far-tail inputs (exact log-domain pass),
    marginalised pixels."""
    from deeprob.spn.models import DgcSpn
    from tests.util import randomise_dgc
    monkeypatch.setenv('DPK_DGC_STREAM_MIN_B', '0')
    torch.manual_seed(11)
    model = DgcSpn(shape, out_classes=classes, n_batch=8, sum_channels=8, depthwise=True, n_pooling=pooling)
    randomise_dgc(model, 70)
    with torch.no_grad():   # vanishing and dominating sum weights: the exact log-domain pass of the streaming kernels
        from deeprob.spn.layers.dgcspn import SpatialSumLayer
        sums = [l for l in model.layers if isinstance(l, SpatialSumLayer)]
        sums[0].weight[0, 1] = -200.0
        sums[0].weight[0, 1, 2, 2] = 50.0
        sums[-1].weight[3, :, 1, :] = -150.0
        sums[-1].weight[3, 5, 1, :] = 0.0
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    plan = dorc.schedule(shape, 8, 8, True, pooling)
    x = torch.randn(B, *shape, generator=torch.Generator().manual_seed(3))
    x[1] = 35.0                                   # every Gaussian far in its tail
    x[4, :, :3] = 3.0e3                           # part of an image absurdly far out: log-densities of -1e7
    x[2, :, ::2] = float('nan')
    x[3] = float('nan')
    want = dorc.dgcspn_forward(sd, x, plan)
    model.cuda()
    with torch.no_grad():
        got = model(x.cuda())
        again = model(x.cuda()[5:])
    assert tuple(got.shape) == (B, classes)
    per_sample = ((got.cpu() - want).abs() / want.abs().clamp_min(1.0)).max().item()   # (magnitudes differ by 1e5)
    assert per_sample <= LL_TOL
    assert torch.equal(again, got[5:])
    monkeypatch.setenv('DPK_DGC_STREAM_MIN_B', '1000000000')
    with torch.no_grad():
        other = model(x.cuda())
    assert rel_err(got.cpu().numpy(), other.cpu().numpy()) <= 2e-6


@pytest.mark.parametrize('shape,padding,stride,dil,cout,B', [((8, 9, 9), 'full', 1, 2, 8, 21), ((5, 12, 12), 'valid', 2, 1, 7, 9),
                                                           ((3, 7, 7), 'full', 1, 4, 4, 300), ((8, 16, 16), 'final', 1, 4, 8, 40)])
def test_fused_level_autograd_matches_layer_chain(shape, padding, stride, dil, cout, B):
    """The training route's single node for a depthwise product + sum pair (ops_spatial.SpatialProdSumFn: fused forward,
    tap-reading backward) against the two layers chained: values, input gradient and weight gradient."""
    from deeprob.spn.layers.dgcspn import SpatialProductLayer, SpatialSumLayer
    from deeprob.hip import ops_spatial
    gen = torch.Generator().manual_seed(9)
    prod = SpatialProductLayer(shape, 2, padding, stride, dil, depthwise=True).cuda()
    ssum = SpatialSumLayer(prod.out_features, cout).cuda()
    with torch.no_grad():
        ssum.weight.copy_(torch.randn(ssum.weight.shape, generator=gen) * 2)
    x = (torch.randn(B, *shape, generator=gen) * 3).cuda()
    x[0] = float('-inf')                                   # a sample whose every input is log 0
    gout = torch.randn(B, cout, *prod.out_features[1:], generator=gen).cuda()

    xa = x.clone().requires_grad_(True)
    ya = ssum(prod(xa))
    ga_x, ga_w = torch.autograd.grad(ya, [xa, ssum.weight], gout)
    xb = x.clone().requires_grad_(True)
    yb = ops_spatial.spatial_prodsum_autograd(xb, prod, ssum.weight, ssum._ws)
    assert yb is not None
    gb_x, gb_w = torch.autograd.grad(yb, [xb, ssum.weight], gout)
    fin = torch.isfinite(ya)
    assert torch.equal(fin, torch.isfinite(yb))
    assert rel_err(yb[fin].detach().cpu().numpy(), ya[fin].detach().cpu().numpy()) <= 1e-6
    assert grad_err(gb_x[1:].cpu().numpy(), ga_x[1:].cpu().numpy()) <= 1e-5
    assert grad_err(gb_w.cpu().numpy(), ga_w.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize('shape,padding,stride,dil,B', [((8, 29, 29), 'full', 1, 2, 300), ((8, 43, 43), 'full', 1, 16, 270),
                                                        ((8, 16, 16), 'valid', 2, 1, 257), ((8, 12, 12), 'full', 1, 4, 333)])
def test_pixel_major_maps_between_streaming_levels(shape, padding, stride, dil, B):
    """Round 6: the streaming level kernels take and leave pixel-major maps (torch's channels_last; DPK_FLAG_IN_PIXEL_MAJOR /
    DPK_FLAG_OUT_PIXEL_MAJOR): all four layout combinations of one level give the SAME values bit for bit (the layout only
    moves data), far-tail / -inf inputs (exact log-domain pass) and a ragged last batch slice included -- and equal the
    oracle's product + sum."""
    from deeprob.spn.layers.dgcspn import SpatialProductLayer, SpatialSumLayer
    from deeprob.hip import ops_spatial
    torch.manual_seed(17)
    prod = SpatialProductLayer(shape, kernel_size=(2, 2), padding=padding, stride=(stride, stride), dilation=(dil, dil),
                               depthwise=True)
    sm = SpatialSumLayer(prod.out_features, 8)
    with torch.no_grad():
        sm.weight.copy_(torch.randn(sm.weight.shape) * 2)
        sm.weight[2, :, 1, :] = -150.0
        sm.weight[2, 5, 1, :] = 0.0
    gen = torch.Generator().manual_seed(18)
    x = torch.randn(B, *shape, generator=gen) * 3
    x[1] = -2.0e4
    x[2, 3] = float('-inf')
    x[3, :, ::2] = 0.0
    want = dorc.spatial_sum(dorc.spatial_product(x, prod.pad, stride, dil, True), sm.weight.detach())
    prod, sm = prod.cuda(), sm.cuda()
    xd = x.cuda()
    assert ops_spatial.level_streams(prod, B, 8)
    xpm = xd.contiguous(memory_format=torch.channels_last)
    assert ops_spatial._is_pixel_major(xpm) and not ops_spatial._is_pixel_major(xd)
    outs = {}
    with torch.no_grad():
        for in_pm in (False, True):
            for out_pm in (False, True):
                y = ops_spatial.spatial_prodsum(xpm if in_pm else xd, prod, sm.weight, sm._ws, out_pixel_major=out_pm)
                assert y is not None and tuple(y.shape) == (B,) + tuple(prod.out_features)
                assert ops_spatial._is_pixel_major(y) == out_pm
                outs[(in_pm, out_pm)] = y.contiguous()
    ref = outs[(False, False)]
    for k, y in outs.items():
        assert torch.equal(y, ref), k
    fin = torch.isfinite(want)
    assert torch.equal(fin, torch.isfinite(ref.cpu()))
    assert ((ref.cpu()[fin] - want[fin]).abs() / want[fin].abs().clamp_min(1.0)).max().item() <= LL_TOL
