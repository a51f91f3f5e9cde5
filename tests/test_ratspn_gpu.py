"""Parity of the HIP RAT-SPN path (through the C ABI) with the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import ratspn_oracle as orc
from tests.util import rel_err, grad_err, state_to_model, report_measured

pytestmark = pytest.mark.gpu

LL_TOL = 1e-5     # north-star: 1e-5 relative on fp32 log-likelihoods
GRAD_TOL = 1e-4   # relative to the largest magnitude of the tensor

# name -> constructor kwargs of GaussianRatSpn
MODELS = {
    'ratspn_g784_d2_r8_i2_s2': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2),
    'ratspn_g784_d2_r8_i8_s8': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8),
    'ratspn_g784_d2_r8_i4_s2': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=4, rg_sum=2),
    'ratspn_g784_d2_r8_i16_s16': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16),
    'ratspn_g784_d1_r4_i8_scale': dict(in_features=784, rg_depth=1, rg_repetitions=4, rg_batch=8, rg_sum=8,
                                       optimize_scale=True),
    'ratspn_g784_d3_r5_i4_s4_c10': dict(in_features=784, out_classes=10, rg_depth=3, rg_repetitions=5,
                                        rg_batch=4, rg_sum=4, optimize_scale=True),
    'ratspn_g100_d2_r11_i2_s4_c3': dict(in_features=100, out_classes=3, rg_depth=2, rg_repetitions=11,
                                        rg_batch=2, rg_sum=4),
    'ratspn_g15_d2_r3_i3_s5_pad': dict(in_features=15, rg_depth=2, rg_repetitions=3, rg_batch=3, rg_sum=5,
                                       optimize_scale=True),
    'ratspn_g15_d3_r2_i2_s2_pad': dict(in_features=15, rg_depth=3, rg_repetitions=2, rg_batch=2, rg_sum=2),
}
SEEDS = {'ratspn_g784_d3_r5_i4_s4_c10': 7, 'ratspn_g100_d2_r11_i2_s4_c3': 3, 'ratspn_g15_d3_r2_i2_s2_pad': 1}


@pytest.fixture(params=['small', 'ring', 'slice'])
def mapping(request):
    """The tile mappings of the matrix-core route: the small-batch kernels (32-sample tiles, feature axis split over
    the waves; what a batch of up to 16384 samples takes by default), the persistent 128-sample ring kernels (forced
    here by a zero threshold) and the persistent 32-sample blocks with the mean table in registers
    (csrc/ratspn_gemm_slice.hip; what larger batches of the two-channel 784-variable models take by default, forced
    here for every batch size -- shapes outside it keep the small-batch kernels).  Tests that take this fixture run on each."""
    from deeprob.hip import load_library
    lib = load_library()
    prev = lib.dpk_ratspn_small_batch_max(0 if request.param == 'ring' else -1)
    prev_slice = lib.dpk_ratspn_slice_batch_min(0 if request.param == 'slice' else -1)
    yield request.param
    lib.dpk_ratspn_small_batch_max(prev)
    lib.dpk_ratspn_slice_batch_min(prev_slice)


def build(name, golden, device='cuda'):
    from deeprob.spn.models import GaussianRatSpn
    g = golden(name)
    model = GaussianRatSpn(random_state=SEEDS.get(name, 42), **MODELS[name])
    # the region graph must come out identical to the reference's for the same seed
    assert np.array_equal(model.base_layer.mask.numpy(), g['sd.base_layer.mask'])
    state_to_model(model, g, device).eval()
    return model, g


@pytest.mark.parametrize('name', sorted(MODELS))
def test_forward_golden(golden, name, mapping):
    model, g = build(name, golden)
    with torch.no_grad():
        ll = model(torch.from_numpy(g['x']).cuda())
        ll_nan = model(torch.from_numpy(g['x_nan']).cuda())
    assert ll.shape == g['ll'].shape and ll.dtype == torch.float32 and ll.is_contiguous()
    assert rel_err(ll.cpu().numpy(), g['ll']) <= LL_TOL
    assert rel_err(ll_nan.cpu().numpy(), g['ll_nan']) <= LL_TOL
    assert np.all(np.abs(ll_nan.cpu().numpy()[1]) < 1e-5)


@pytest.mark.parametrize('name', sorted(MODELS))
def test_layers_golden(golden, name):
    """Per-layer operators chained like the reference's python loop (the non-fused route)."""
    model, g = build(name, golden)
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        h = model.base_layer(x)
        if 'act.leaf' in g.files:
            assert rel_err(h.cpu().numpy(), g['act.leaf']) <= LL_TOL
        for i, layer in enumerate(model.layers):
            h = layer(h)
            key = 'act.layer{}'.format(i)
            if key in g.files:
                assert rel_err(h.cpu().numpy(), g[key]) <= LL_TOL, key
        out = model.root_layer(h)
    assert rel_err(out.cpu().numpy(), g['ll']) <= LL_TOL


def _oracle_grads_fp64(g):
    sd = orc.state_from_npz(g, dtype=torch.float64)
    names = [k[5:] for k in g.files if k.startswith('grad.') and k != 'grad.x']
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    x = torch.from_numpy(g['x']).double().requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    with torch.enable_grad():
        orc.ratspn_loss(orc.ratspn_forward(sd, x), y).backward()
    out = {k: sd[k].grad.numpy() for k in names}
    out['x'] = x.grad.numpy()
    return out


@pytest.mark.parametrize('name', [n for n in sorted(MODELS) if 'i16' not in n])
def test_backward_golden(golden, name):
    model, g = build(name, golden)
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    y = torch.from_numpy(g['y']).cuda() if 'y' in g.files else None
    with torch.enable_grad():
        loss = model.loss(model(x), y)
        loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= LL_TOL * max(1.0, abs(float(g['loss'])))
    # Bar (in the form the coupling tests use): distance to the FP64 oracle <= GRAD_TOL, or -- where the
    # reference's own fp32 gradient (the golden) is further than half of that from fp64 -- twice the golden's own distance.
    # (The discriminative loss differentiates softmax(out) - onehot, which cancels for confident samples; the generative
    # models' responsibilities carry the fp32 rounding of log-likelihoods of magnitude 10^3.)  Never "distance to the golden
    # within a multiple of its noise": that admits a result 5x further from the truth than the reference.
    ref64 = _oracle_grads_fp64(g)

    def check(key, got):
        own = grad_err(g['grad.' + key], ref64[key])
        tol = max(GRAD_TOL, 2.0 * own)
        err = grad_err(got, ref64[key])
        report_measured('test_backward_golden[%s] grad.%s vs fp64' % (name, key), err, tol,
                        '(golden = reference fp32 vs fp64: %.2e)' % own)
        assert err <= tol, key

    check('x', x.grad.cpu().numpy())
    for k, p in model.named_parameters():
        if 'grad.' + k in g.files:
            assert p.grad is not None, k
            check(k, p.grad.cpu().numpy())


@pytest.mark.parametrize('B', [1, 63, 64, 129, 1000])
@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i2_s2', 'ratspn_g784_d2_r8_i8_s8',
                                  'ratspn_g15_d2_r3_i3_s5_pad'])
def test_forward_vs_oracle_ragged_batches(golden, name, B, mapping):
    """Seeded inputs at batch sizes around the tile size, with NaN / inf evidence, vs the oracle."""
    model, g = build(name, golden)
    D = MODELS[name]['in_features']
    gen = torch.Generator().manual_seed(B)
    x = torch.randn(B, D, generator=gen) * 1.5
    x[torch.rand(B, D, generator=gen) < 0.1] = float('nan')
    if B > 2:
        x[B // 2] = float('nan')
        x[0, 0] = float('-inf')
    sd = orc.state_from_npz(g)
    want = orc.ratspn_forward(sd, x).numpy()
    with torch.no_grad():
        got = model(x.cuda()).cpu().numpy()
    assert rel_err(got, want) <= LL_TOL


def test_empty_batch(golden):
    model, g = build('ratspn_g784_d2_r8_i2_s2', golden)
    with torch.no_grad():
        out = model(torch.empty(0, 784, device='cuda'))
    assert out.shape == (0, 1)


def test_float64_and_noncontiguous_inputs(golden):
    model, g = build('ratspn_g784_d2_r8_i2_s2', golden)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        a = model(x.double().cuda())
        b = model(x.cuda().t().contiguous().t())
    assert rel_err(a.float().cpu().numpy(), g['ll']) <= LL_TOL
    assert rel_err(b.cpu().numpy(), g['ll']) <= LL_TOL


def test_cpu_tensor_fails_loudly(golden):
    from deeprob.hip import HipError
    model, g = build('ratspn_g784_d2_r8_i2_s2', golden)
    with pytest.raises(HipError):
        model(torch.from_numpy(g['x']))


def test_bernoulli_known_answer(golden):
    """Reference KAT (tests/test_ratspn.py:46-48) on the HIP path: sum over 2^15 assignments = 1."""
    from deeprob.spn.models import BernoulliRatSpn
    g = golden('ratspn_bernoulli_15_d3_r4_i4_s2')
    model = BernoulliRatSpn(15, rg_depth=3, rg_repetitions=4, rg_batch=4, rg_sum=2, random_state=42)
    state_to_model(model, g, 'cuda').eval()
    bits = ((np.arange(2 ** 15)[:, None] >> np.arange(14, -1, -1)[None, :]) & 1).astype(np.float32)
    with torch.no_grad():
        ll = model(torch.from_numpy(bits).cuda())
        ll_nan = model(torch.from_numpy(g['x_sub']).cuda())
    assert rel_err(ll.cpu().numpy(), g['ll']) <= LL_TOL
    assert np.isclose(torch.sum(torch.exp(ll)).item(), 1.0)
    assert rel_err(ll_nan.cpu().numpy(), g['ll_sub_nan']) <= LL_TOL
    x = torch.from_numpy(g['x_grad_in']).cuda()
    with torch.enable_grad():
        loss = model.loss(model(x))
        loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= LL_TOL * abs(float(g['loss']))
    for k, p in model.named_parameters():
        assert grad_err(p.grad.cpu().numpy(), g['grad.' + k]) <= GRAD_TOL, k


def test_layer_edge_cases(golden):
    """-inf inputs through Product / Sum / Root: -inf out, never NaN; vanishing-weight exact path."""
    from deeprob.spn.layers.ratspn import ProductLayer, SumLayer, RootLayer
    g = golden('ratspn_layers_edge')
    prod, sm, root = ProductLayer(8, 3).cuda(), SumLayer(4, 9, 5).cuda(), RootLayer(4, 5, 2).cuda()
    with torch.no_grad():
        sm.weight.copy_(torch.from_numpy(g['sum_weight']))
        root.weight.copy_(torch.from_numpy(g['root_weight']))
        p = prod(torch.from_numpy(g['h']).cuda())
        s = sm(p)
        r = root(s)
    assert rel_err(p.cpu().numpy(), g['prod_out']) == 0.0
    assert rel_err(s.cpu().numpy(), g['sum_out']) <= LL_TOL
    assert rel_err(r.cpu().numpy(), g['root_out']) <= LL_TOL


def test_fused_exact_path_with_vanishing_weights(golden, mapping):
    """Trained-looking sum weights (one dominant, the rest ~e^-200) and widely spread leaf values force
    the fused kernel off the exp-domain fast path; it must still match the oracle."""
    model, g = build('ratspn_g784_d2_r8_i4_s2', golden)
    with torch.no_grad():
        for layer in model.layers:
            if hasattr(layer, 'weight'):
                layer.weight[:, :, :] = -200.0
                layer.weight[:, :, 5] = 0.0
        model.root_layer.weight[:, :] = -150.0
        model.root_layer.weight[:, 3] = 0.0
        model.base_layer.loc.mul_(3.0)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = torch.from_numpy(g['x']) * 2.0
    want = orc.ratspn_forward(sd, x).numpy()
    with torch.no_grad():
        got = model(x.cuda()).cpu().numpy()
    assert np.isfinite(want).all()
    assert rel_err(got, want) <= LL_TOL


def test_full_size_properties():
    """BASELINE config at full batch (65536): size-independent properties instead of an oracle run --
    tile independence (any slice of the batch gives the same LLs, bit for bit within one mapping), all-NaN rows give
    LL = 0, and the fused fp64 sum equals the sum of the returned LLs."""
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.hip import ops
    torch.manual_seed(0)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
    x = torch.randn(65536, 784, device='cuda', generator=torch.Generator('cuda').manual_seed(0))
    # ---- clean evidence: every launch of a given size takes the same kernel ----
    with torch.no_grad():
        ll = model(x)
        part = model(x[1000:1000 + 4097])
        acc = torch.zeros(2, dtype=torch.float64, device='cuda')
        ll2 = model._forward_fused(x, acc)
        big = model(x[20000:60000])
        odd = model(x[20007:60000])     # (not aligned to any tile of any mapping)
        sub = model(x[1000 + 33:1000 + 33 + 2000])
    # a slice of 4097 rows takes the small-batch kernels, the full batch the persistent ones: same values to fp32
    # rounding (the K-steps meet in eight / seven partial sums or one), bit-identical within one mapping
    assert rel_err(part.cpu().numpy(), ll[1000:1000 + 4097].cpu().numpy()) <= 1e-6
    assert torch.equal(ll[20000:60000], big)
    assert torch.equal(ll[20007:60000], odd)
    assert torch.equal(part[33:33 + 2000], sub)
    assert torch.equal(ll, ll2)
    assert acc[1].item() == 65536
    assert abs(acc[0].item() - ll.double().sum().item()) <= 1e-9 * abs(acc[0].item())
    # ---- one marginalised row: exactly 0 for it, the other rows unchanged to fp32 rounding (a launch that meets NaN
    # evidence raises the hint that sends the following launches of this model to the variant built for it: same values to
    # rounding, not bit for bit, while the hint lasts) ----
    xn = x.clone()
    xn[12345] = float('nan')
    with torch.no_grad():
        for _ in range(3):
            acc = torch.zeros(2, dtype=torch.float64, device='cuda')
            lln = model._forward_fused(xn, acc)
            assert abs(lln[12345].item()) < 1e-5
            keep = torch.ones(65536, dtype=torch.bool, device='cuda')
            keep[12345] = False
            assert rel_err(lln[keep].cpu().numpy(), ll[keep].cpu().numpy()) <= 1e-6
            assert acc[1].item() == 65536
            assert abs(acc[0].item() - lln.double().sum().item()) <= 1e-9 * abs(acc[0].item())


def _oracle_rows(sd, x_rows, chunk=512):
    """The oracle on a set of rows, in chunks (the (16,16) model's leaf temporaries are 1 GB per 1000 rows)."""
    return np.concatenate([orc.ratspn_forward(sd, x_rows[i:i + chunk]).numpy() for i in range(0, x_rows.shape[0], chunk)])


@pytest.mark.parametrize('B', [65536, 32768, 16384, 8192])
@pytest.mark.parametrize('chan', [(2, 2), (8, 8), (16, 16)])
def test_full_size_vs_oracle(chan, B):
    """BASELINE configs 2 / 3 at their full launch sizes through the DEFAULT routing (65536 = the headline launch, 32768 /
    16384 / 8192 = one rank's shard of config 3 and of the strong-scaling reading): 2048 rows drawn uniformly from the WHOLE
    batch -- so that the later-iteration blocks of the persistent kernels (rotated K loop, re-requested K-step slots,
    one-block-delayed commit) are hit, not only the first block of every work-group -- against the oracle, 1e-5."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(0)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=chan[0], rg_sum=chan[1], random_state=42).eval()
    with torch.no_grad():
        model.base_layer.loc.mul_(1.5)      # away from the initialiser's scale, inside the fast path's envelope
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda()
    x = torch.randn(B, 784, device='cuda', generator=torch.Generator('cuda').manual_seed(B + chan[0]))
    rows = torch.randint(0, B, (2048,), generator=torch.Generator().manual_seed(17))
    rows[0], rows[1] = 0, B - 1
    with torch.no_grad():
        acc = torch.zeros(2, dtype=torch.float64, device='cuda')
        ll = model._forward_fused(x, acc)
        fused = ll is not None
        if not fused:                       # 16 channels: the folded route (leaf | product+sum | product+root launches)
            ll = model(x)
    want = _oracle_rows(sd, x[rows.cuda()].cpu())
    err = rel_err(ll[rows.cuda()].cpu().numpy(), want)
    report_measured('test_full_size_vs_oracle[%dx%d, B=%d] 2048 rows of the whole batch' % (chan[0], chan[1], B), err, LL_TOL)
    assert err <= LL_TOL
    if fused:
        assert acc[1].item() == B and abs(acc[0].item() - ll.double().sum().item()) <= 1e-9 * abs(acc[0].item())


@pytest.mark.parametrize('kw', [dict(rg_batch=8, rg_sum=8), dict(rg_batch=8, rg_sum=4, optimize_scale=True),
                                dict(rg_batch=16, rg_sum=16), dict(rg_depth=1, rg_batch=8, rg_sum=8),
                                dict(rg_depth=3, rg_batch=8, rg_sum=8, rg_repetitions=3)])
def test_wide_channel_blocks_above_the_tile_switch(kw):
    """Launches above 32768 samples run the 8-channel blocks with two samples per lane (fused kernel up to depth 2, leaf
    kernel of the folded route): a 40000-sample batch (ragged last tile) against the oracle on a scattered subset
    (NaN / inf / far-tail rows included), and against the same rows evaluated in a small launch (one sample per lane)."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(3)
    base = dict(in_features=100, rg_depth=2, rg_repetitions=4, random_state=7)
    base.update(kw)
    model = GaussianRatSpn(**base).eval()
    if base.get('optimize_scale'):
        with torch.no_grad():
            model.base_layer.scale.mul_(1.0 + 0.2 * torch.rand_like(model.base_layer.scale))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B = 40000
    x = torch.randn(B, 100, generator=torch.Generator().manual_seed(4))
    x[5] = float('nan')
    x[64, :30] = float('nan')
    x[129, 3] = float('inf')
    x[39999, 7] = 40.0                                   # beyond the expanded-square bound, in the ragged tile
    rows = torch.cat([torch.tensor([5, 64, 129, 39999, 39936, 39937]), torch.randint(0, B, (250,),
                                                                                     generator=torch.Generator().manual_seed(5))])
    want = orc.ratspn_forward(sd, x[rows]).numpy()
    model = model.cuda()
    with torch.no_grad():
        big = model(x.cuda())
        small = model(x[rows].cuda())
    assert rel_err(big[rows.cuda()].cpu().numpy(), want) <= LL_TOL
    assert rel_err(small.cpu().numpy(), want) <= LL_TOL


@pytest.mark.parametrize('kw,D', [(dict(rg_repetitions=3, rg_sum=4, out_classes=5), 52),
                                  (dict(rg_repetitions=8, rg_sum=2), 784),
                                  (dict(rg_repetitions=5, rg_sum=8, out_classes=3, rg_depth=1), 200)])
def test_eight_channel_ring_kernel_shapes(kw, D):
    """The 128-sample mapping of the 8-channel kernel (batches above 16384; csrc/ratspn_gemm_wide.hip): fewer repetitions
    than waves, several classes, 2 / 4 / 8 sum nodes, feature counts that end inside a 64-feature chunk, a ragged last
    tile, clean rows next to rows with NaN / inf / far-tail evidence -- against the oracle on a scattered subset, and
    against the 32-sample mapping on the same rows."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(11)
    base = dict(in_features=D, rg_depth=2, rg_batch=8, random_state=5)
    base.update(kw)
    model = GaussianRatSpn(**base).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B = 16384 + 128 + 77
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, D, generator=gen)
    x[7, ::3] = float('nan')
    x[130] = float('nan')
    x[4097, 5] = float('-inf')
    x[B - 1, 11] = 35.0
    x[B - 40, :] *= 9.0                                   # sum x^2 beyond the expanded-square bound for the row
    rows = torch.cat([torch.tensor([7, 130, 4097, B - 1, B - 40, B - 77, B - 78, 128, 127]),
                      torch.randint(0, B, (200,), generator=gen)])
    want = orc.ratspn_forward(sd, x[rows]).numpy()
    model = model.cuda()
    with torch.no_grad():
        big = model(x.cuda())
        small = model(x[rows].cuda())
    assert rel_err(big[rows.cuda()].cpu().numpy(), want) <= LL_TOL
    assert rel_err(small.cpu().numpy(), want) <= LL_TOL


def test_marginalised_inputs_on_both_kernel_builds(mapping):
    """The two-channel unit-scale kernel exists in two builds (tiles with NaN / inf / out-of-bound evidence leave the
    LDS record pipeline, or stay on it in the exact per-entry form); a launch takes the second one while a recent
    launch met such a tile.  The same inputs evaluated before and after the switch: both match the oracle, an
    all-NaN row is exactly 0 on both, and clean inputs are unaffected by which build ran."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(5)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(700, 784, generator=gen)
    clean = x.clone()
    x[torch.rand(700, 784, generator=gen) < 0.3] = float('nan')
    x[3] = float('nan')
    x[10, 5] = float('inf')
    x[11, 700] = -float('inf')
    x[12, 9] = 25.0                                   # beyond the expanded-square bound
    x[200:264] = clean[200:264]                       # one clean tile in between
    want = orc.ratspn_forward(sd, x).numpy()
    want_clean = orc.ratspn_forward(sd, clean).numpy()
    model = model.cuda()
    outs = []
    with torch.no_grad():
        for _ in range(4):                            # the first launch raises the hint, the later ones see it
            outs.append(model(x.cuda()).cpu().numpy())
            torch.cuda.synchronize()
        after = model(clean.cuda()).cpu().numpy()     # clean inputs right after: still the second build
    for got in outs:
        # all-NaN row: 0 like the reference's.  Exactly 0 where the row shares its exact evaluation with the +-inf rows
        # (ring kernels: 32 rows per wave); on the exp-domain fast path log(sum softmax(w)) leaves the reference's own
        # rounding noise (~1e-8 per sum node)
        assert abs(got[3, 0]) < 1e-6
        if mapping == 'ring':
            assert got[3, 0] == 0.0
        assert rel_err(got, want) <= LL_TOL
    assert rel_err(after, want_clean) <= LL_TOL


def test_fused_plan_is_the_same_call(golden):
    """ops.FusedForwardPlan (pre-bound call for a resident buffer) == model(x) bit for bit; it reads live
    parameters and notices when a pinned address moved."""
    name = 'ratspn_g784_d2_r8_i2_s2'
    model, g = build(name, golden)
    x = torch.from_numpy(g['x_nan']).cuda()
    with torch.no_grad():
        want = model(x)
        plan = model.fused_plan(x)
        assert plan is not None and plan.valid()
        acc = torch.zeros(2, dtype=torch.float64, device='cuda')
        got = plan.run(acc)
        assert torch.equal(got, want)
        assert abs(acc[0].item() - want.double().sum().item()) < 1e-6 * abs(want.double().sum().item())
        assert acc[1].item() == want.numel()
        model.base_layer.loc.add_(0.25)            # in place: same storage, the plan must see it
        assert torch.equal(plan.run(), model(x))
        model.base_layer.loc.data = model.base_layer.loc.data.clone()   # storage moved
        assert not plan.valid()


@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i2_s2', 'ratspn_g784_d2_r8_i4_s2', 'ratspn_g784_d2_r8_i8_s8',
                                  'ratspn_g784_d2_r8_i16_s16'])
@pytest.mark.parametrize('case', ['means_beyond_bound', 'one_repetition_beyond_bound', 'outlier_evidence', 'offset_data',
                                  'at_the_bound'])
def test_expanded_square_guard(golden, case, name, mapping):
    """The unit-scale fused kernel evaluates sum (x-mu)^2 as sum x^2 - 2 sum x mu + sum mu^2 only while |x| and
    |mu| stay <= 6; outside it must take the direct / exact forms.  Every regime is held to the same
    1e-5 relative bar against the oracle, including the adversarial one (x ~ mu, both at the bound)."""
    model, g = build(name, golden)
    x = torch.from_numpy(g['x']).clone()
    with torch.no_grad():
        if case == 'means_beyond_bound':
            model.base_layer.loc.mul_(4.0)           # N(0,1) * 4: many |mu| > 6 -> direct form for the model
        elif case == 'one_repetition_beyond_bound':
            model.base_layer.loc[4:8].mul_(4.0)      # the regions of repetition 1 only: the verdict is the MODEL's (the
                                                     # root adds one common -1/2 sum x^2 term for every repetition)
        elif case == 'outlier_evidence':
            x[3, 100] = 50.0                         # one value beyond the bound: its tile leaves the fast path
            x[20] = 7.5
            x[21, ::3] = -6.5
        elif case == 'offset_data':
            x += 5.0                                 # |x| ~ 5 +- 1 with zero-centred means: large cancellation-free LL
        else:
            model.base_layer.loc.fill_(5.9)          # x ~ mu ~ 5.9: worst case for the expansion's cancellation
            model.base_layer.loc.add_(0.05 * torch.randn_like(model.base_layer.loc))
            x = 5.9 + 0.05 * x
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = orc.ratspn_forward(sd, x).numpy()
    want64 = orc.ratspn_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()},
                                x.double()).numpy()
    with torch.no_grad():
        got = model(x.cuda()).cpu().numpy()
    assert rel_err(got, want) <= LL_TOL
    assert rel_err(got, want64) <= LL_TOL


@pytest.mark.parametrize('kw', [dict(rg_depth=2, rg_batch=16, rg_sum=16), dict(rg_depth=1, rg_batch=16, rg_sum=16),
                                dict(rg_depth=3, rg_batch=16, rg_sum=12, out_classes=7, rg_repetitions=5)])
def test_folded_route_for_wide_models(kw):
    """Shapes outside the single-launch kernel (16 channels) evaluate as leaf kernel + products folded into the sum /
    root layers; same numbers as the per-layer chain and as the oracle, including NaN / -inf evidence."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(1)
    base = dict(in_features=100, rg_repetitions=8, random_state=3, optimize_scale=True)
    base.update(kw)
    model = GaussianRatSpn(**base).eval()
    with torch.no_grad():
        model.root_layer.weight[:, ::5] = -300.0        # vanishing root weights: exercises the exact fallback
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(130, 100, generator=torch.Generator().manual_seed(2)) * 2.0
    x[3] = float('nan')
    x[5, :40] = float('nan')
    x[7, 11] = float('inf')
    want = orc.ratspn_forward(sd, x).numpy()
    model = model.cuda()
    with torch.no_grad():
        assert model._forward_fused(x.cuda()) is None      # really outside the fused envelope
        folded = model(x.cuda())
    chained = model(x.cuda()).detach()                     # graph needed -> per-layer operators
    assert rel_err(folded.cpu().numpy(), want) <= LL_TOL
    assert rel_err(folded.cpu().numpy(), chained.cpu().numpy()) <= LL_TOL


def test_gradients_with_marginalised_inputs(golden):
    """with NaN evidence and autograd on, the reference's Normal.log_prob backward computes 0 * NaN
    and returns NaN in loc.grad / scale.grad / x.grad wherever a NaN input was touched (the oracle, which replays
    the same ATen ops, shows it).  The HIP path returns the mathematically masked gradient: marginalised entries
    contribute nothing.  This test documents the divergence and pins the masked gradient."""
    name = 'ratspn_g15_d2_r3_i3_s5_pad'
    model, g = build(name, golden)
    model.train()
    x = torch.from_numpy(g['x']).clone()
    nan_mask = torch.rand(x.shape, generator=torch.Generator().manual_seed(9)) < 0.3
    x[nan_mask] = float('nan')
    xg = x.cuda().requires_grad_(True)
    model.loss(model(xg)).backward()
    assert torch.isfinite(model.base_layer.loc.grad).all() and torch.isfinite(xg.grad).all()
    assert torch.equal(xg.grad.cpu()[nan_mask], torch.zeros(int(nan_mask.sum())))

    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # (1) the reference's behaviour, via the oracle: NaN gradients
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    xo = x.clone().requires_grad_(True)
    orc.ratspn_loss(orc.ratspn_forward({**sd, **leaves}, xo)).backward()
    assert torch.isnan(leaves['base_layer.loc'].grad).any() and torch.isnan(xo.grad[nan_mask]).all()
    # (2) the masked gradient: same forward value, marginalised terms cut out of the graph
    for v in leaves.values():
        v.grad = None
    mask, d = sd['base_layer.mask'], sd['base_layer.mask'].shape[1]
    drop = nan_mask[:, mask].unsqueeze(2).expand(-1, -1, sd['base_layer.loc'].shape[1], -1)     # [B,R,I,d]
    xf = torch.where(nan_mask, torch.zeros_like(x), x).requires_grad_(True)
    out = orc.ratspn_forward({**sd, **leaves}, xf, drops={'leaf': drop})
    orc.ratspn_loss(out).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert grad_err(p.grad.cpu().numpy(), leaves[k].grad.numpy()) <= GRAD_TOL, k
    assert grad_err(xg.grad.cpu().numpy(), xf.grad.numpy()) <= GRAD_TOL


@pytest.mark.parametrize('kw', [dict(rg_depth=1, rg_repetitions=8), dict(rg_depth=1, rg_repetitions=3, out_classes=4),
                                dict(rg_depth=2, rg_repetitions=19), dict(rg_depth=3, rg_repetitions=8, out_classes=2),
                                dict(rg_depth=2, rg_repetitions=8, rg_batch=1, rg_sum=2),
                                dict(rg_depth=2, rg_repetitions=5, rg_batch=2, rg_sum=8, in_features=77)])
def test_unit_scale_fused_shapes_vs_oracle(kw):
    """The expanded / x^2-factored unit-scale kernel over depths, repetition counts (more than one pass of 8 waves),
    class counts and a padded (77-variable) region graph, with clean rows, rows that turn exact mid-way (one large
    value in a late chunk), NaN rows and ragged batch sizes."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(4)
    base = dict(in_features=784, rg_batch=2, rg_sum=2, random_state=11)
    base.update(kw)
    model = GaussianRatSpn(**base).eval()
    D = base['in_features']
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(333, D, generator=torch.Generator().manual_seed(5))
    x[5, D - 3] = 9.0                 # beyond the bound in the last chunk only
    x[140, D // 2] = float('nan')
    x[141] = float('nan')
    x[300, 0] = -7.0                  # beyond the bound in the first chunk
    want = orc.ratspn_forward(sd, x).numpy()
    model = model.cuda()
    with torch.no_grad():
        got = model(x.cuda()).cpu().numpy()
    assert rel_err(got, want) <= LL_TOL


@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i4_s2', 'ratspn_g100_d2_r11_i2_s4_c3', 'ratspn_g784_d1_r4_i8_scale',
                                  'ratspn_g784_d3_r5_i4_s4_c10'])
def test_mpe_golden(golden, name):
    """RatSpn.mpe (reference: models/ratspn.py:124-162) through the one-launch top-down kernel (csrc/ratspn_topdown.hip)
    on the HIP forward activations: the reference's own completions (depths 1 / 2 / 3), observed entries untouched, also
    with given class labels; and the layer-by-layer form (the layers' mpe methods) gives the same."""
    model, _ = build(name, golden)
    g = golden(name + '_mpe')
    x = torch.from_numpy(g['x']).cuda()
    got = model.mpe(x).cpu().numpy()
    assert got.shape == g["mpe"].shape and not np.isnan(got).any()
    rows = np.ones(got.shape[0], dtype=bool)
    if model.out_classes > 1:
        # without labels the class is the arg-max of the root outputs: rows whose two best classes tie to within
        # fp32 noise (the fully marginalised row: every class scores ~0) may legitimately pick another class
        with torch.no_grad():
            top = torch.topk(model(x), 2, dim=1).values
        rows = ((top[:, 0] - top[:, 1]) > 1e-4).cpu().numpy()
        assert rows.sum() >= got.shape[0] - 2
    assert np.allclose(got[rows], g['mpe'][rows], rtol=1e-5, atol=1e-6)
    obs = ~np.isnan(g['x'])
    assert np.array_equal(got[obs], g['x'][obs])
    assert np.array_equal(model._mpe_layerwise(x).cpu().numpy()[rows], got[rows])
    if 'y' in g.files:
        got_y = model.mpe(x, y=torch.from_numpy(g['y']).cuda()).cpu().numpy()
        assert np.allclose(got_y, g['mpe_y'], rtol=1e-5, atol=1e-6)


def test_mpe_bernoulli_golden(golden):
    """Bernoulli leaves (mode = [p >= 0.5]), depth 3, two classes: the reference's completions."""
    from deeprob.spn.models import BernoulliRatSpn
    g = golden('ratspn_bernoulli_32_d3_r3_i3_s2_c2_mpe')
    model = BernoulliRatSpn(32, out_classes=2, rg_depth=3, rg_repetitions=3, rg_batch=3, rg_sum=2, random_state=5)
    state_to_model(model, g, 'cuda').eval()
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        ll = model(x)
    assert rel_err(ll.cpu().numpy(), g['ll']) <= LL_TOL
    top = torch.topk(ll, 2, dim=1).values
    rows = ((top[:, 0] - top[:, 1]) > 1e-4).cpu().numpy()
    assert np.array_equal(model.mpe(x).cpu().numpy()[rows], g['mpe'][rows])
    assert np.array_equal(model.mpe(x, y=torch.from_numpy(g['y']).cuda()).cpu().numpy(), g['mpe_y'])


@pytest.mark.parametrize('kw', [dict(in_features=15, rg_depth=2, rg_repetitions=3, rg_batch=3, rg_sum=5, optimize_scale=True),
                                dict(in_features=15, rg_depth=3, rg_repetitions=2, rg_batch=2, rg_sum=2),
                                dict(in_features=70, out_classes=4, rg_depth=4, rg_repetitions=3, rg_batch=5, rg_sum=3),
                                dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16),
                                dict(in_features=33, rg_depth=5, rg_repetitions=2, rg_batch=2, rg_sum=2)])
def test_mpe_vs_oracle_padded_and_deep(kw):
    """The top-down kernel where the reference cannot go (padded region graphs: its unpad_samples raises, see
    oracle/ratspn_oracle.py::_unpad_samples) and at depths 4 / 5, 16 nodes per region, several classes: against the oracle's
    restatement, with the chosen repetition / leaf channels compared on the rows whose decisions are not fp32 ties."""
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.hip import ops
    torch.manual_seed(5)
    model = GaussianRatSpn(random_state=9, **kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    D = kw['in_features']
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(300, D, generator=gen)
    x[torch.rand(300, D, generator=gen) < 0.5] = float('nan')
    x[0] = float('nan')
    y = (torch.arange(300) % kw['out_classes']) if kw.get('out_classes', 1) > 1 else None
    want, og, oo = orc.ratspn_mpe(sd, x, kw['rg_depth'], y=y, return_choice=True)
    model.cuda()
    got = model.mpe(x.cuda(), y=None if y is None else y.cuda()).cpu()
    same = (got == want).all(dim=1)
    # a near-tie between two children flips a whole subtree: such rows are set aside, but must be rare
    assert same.float().mean().item() >= 0.97, same.float().mean().item()
    obs = ~torch.isnan(x)
    assert torch.equal(got[obs], x[obs]) and not torch.isnan(got).any()
    # every completed value is the mean of SOME channel of the region that holds the variable in the chosen repetition
    acts = model._upward_for_mpe(x.cuda())
    leaf = model._leaf_params()
    out, choice = ops.ratspn_topdown(0, leaf[0], 300, model._fused_ctx, x.cuda(), None if y is None else y.cuda(), acts,
                                     model._topdown_logw(), model._topdown_src(), leaf[1], leaf[2], want_choice=True)
    assert torch.equal(out.cpu(), got)
    G0 = 2 ** kw['rg_depth']
    choice = choice.cpu()
    assert torch.equal(choice[same, 0].long(), torch.div(og[same, 0], G0, rounding_mode='floor'))
    assert torch.equal(choice[same, 1:].long(), oo[same])


def test_sample_shapes_and_statistics():
    """RatSpn.sample (reference :164-182): ancestral sampling is random, so the check is statistical -- the samples
    of a model whose leaves are tight around known means must reproduce the mixture's overall mean."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(0)
    model = GaussianRatSpn(16, rg_depth=2, rg_repetitions=4, rg_batch=2, rg_sum=2, optimize_scale=True,
                           random_state=5).cuda().eval()
    with torch.no_grad():
        model.base_layer.scale.fill_(0.05)
        model.base_layer.loc.copy_(3.0 + 0.1 * torch.randn_like(model.base_layer.loc))
    s = model.sample(4000)
    assert tuple(s.shape) == (4000, 16) and torch.isfinite(s).all() and s.is_cuda
    assert abs(s.mean().item() - 3.0) < 0.1 and 0.02 < s.std(dim=0).mean().item() < 0.3
    # sampled points are likely under the model: far above the density of points drawn elsewhere
    with torch.no_grad():
        assert model(s).mean().item() > model(s + 2.0).mean().item() + 100.0


@pytest.mark.parametrize('case', ['gauss_d2_pad', 'gauss_d3_classes', 'gauss_784_wide', 'bernoulli_d3'])
def test_sample_replays_against_the_oracle(case):
    """RatSpn.sample through the one-launch kernel with a fixed seed, replayed by the oracle from the same counter-based
    uniforms (oracle/ratspn_oracle.py::ratspn_sample_replay): the same repetition and leaf channels for every sample whose
    categorical draws are not within fp32 rounding of a CDF step, and the same values (Box-Muller in fp32 against fp64)."""
    from deeprob.spn.models import GaussianRatSpn, BernoulliRatSpn
    from deeprob.hip import ops
    torch.manual_seed(21)
    kw = {'gauss_d2_pad': dict(in_features=15, rg_depth=2, rg_repetitions=3, rg_batch=3, rg_sum=5, optimize_scale=True),
          'gauss_d3_classes': dict(in_features=64, out_classes=3, rg_depth=3, rg_repetitions=4, rg_batch=4, rg_sum=4),
          'gauss_784_wide': dict(in_features=784, rg_depth=2, rg_repetitions=8, rg_batch=16, rg_sum=16),
          'bernoulli_d3': dict(in_features=32, out_classes=2, rg_depth=3, rg_repetitions=3, rg_batch=3, rg_sum=2)}[case]
    model = (BernoulliRatSpn if case.startswith('bern') else GaussianRatSpn)(random_state=4, **kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    n, D, depth = 5000, kw['in_features'], kw['rg_depth']
    y = (torch.arange(n) % kw['out_classes']) if kw.get('out_classes', 1) > 1 else None
    want, rep, chan, margin = orc.ratspn_sample_replay(sd, n, depth, D, seed=987654321, y=y)
    model.cuda()
    got = model.sample(n, y=None if y is None else y.cuda(), seed=987654321)
    assert tuple(got.shape) == (n, D) and got.is_cuda and torch.isfinite(got).all()
    leaf = model._leaf_params()
    out, choice = ops.ratspn_topdown(1, leaf[0], n, model._fused_ctx, None, None if y is None else y.cuda(), None,
                                     model._topdown_logw(), model._topdown_src(), leaf[1], leaf[2], seed=987654321,
                                     want_choice=True)
    assert torch.equal(out, got)                      # (same seed: the same batch)
    choice = choice.cpu()
    clear = torch.from_numpy(margin > 3e-6)           # (fp32 CDF of up to 2048 weights: steps ~5e-4 apart, rounding ~1e-7)
    assert clear.float().mean().item() > 0.97
    assert torch.equal(choice[clear, 0].long(), rep[clear])
    assert torch.equal(choice[clear, 1:].long(), chan[clear])
    if case.startswith('bern'):
        # a Bernoulli draw is u < p: identical unless u is within rounding of p
        assert (got.cpu()[clear] != want[clear]).float().mean().item() < 1e-4
    else:
        err = (got.cpu()[clear] - want[clear]).abs().max().item()
        report_measured('test_sample_replays_against_the_oracle[%s] max |sample - replay|' % case, err, 1e-4)
        assert err <= 1e-4
    # another seed: another batch
    assert not torch.equal(model.sample(n, y=None if y is None else y.cuda(), seed=1), got)


def test_mfma_route_tables_follow_the_parameters(golden):
    """The MFMA route keeps its parameter tables between calls (DPK_FLAG_PARAMS_CACHED) keyed on the parameters'
    addresses and version counters: an in-place update (optimizer step, load_state_dict, copy_) must be seen by the
    next call, and an unchanged model must give bit-identical results call after call."""
    model, g = build('ratspn_g784_d2_r8_i2_s2', golden)
    x = torch.randn(300, 784, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        a = model(x.cuda())
        b = model(x.cuda())
        assert torch.equal(a, b)
        model.base_layer.loc.add_(0.05)
        model.root_layer.weight.mul_(0.5)
        model.layers[1].weight.add_(torch.randn_like(model.layers[1].weight))
        c = model(x.cuda())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = orc.ratspn_forward(sd, x).numpy()
    assert rel_err(c.cpu().numpy(), want) <= LL_TOL
    assert not torch.equal(a, c)
    # a plan bound to a resident buffer sees updates too
    plan = model.fused_plan(x.cuda())
    with torch.no_grad():
        p1 = plan.run().clone()
        model.base_layer.loc.sub_(0.05)
        p2 = plan.run().clone()
    assert torch.equal(p1, c)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert rel_err(p2.cpu().numpy(), orc.ratspn_forward(sd, x).numpy()) <= LL_TOL


@pytest.mark.parametrize('I', [8, 16])
def test_folded_route_tables_follow_the_parameters(I):
    """Wide models (leaf | product+sum | product+root kernels): the softmax rows and MFMA fragments cached in each
    layer's workspace follow in-place parameter updates, survive the per-layer (autograd) route writing its own
    tables into the same workspaces, and an unchanged model repeats bit for bit."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(4)
    model = GaussianRatSpn(64, rg_depth=2, rg_repetitions=4, rg_batch=I, rg_sum=I, random_state=3).cuda().eval()
    x = torch.randn(200, 64, generator=torch.Generator().manual_seed(6))

    def oracle():
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        return orc.ratspn_forward(sd, x).numpy()

    with torch.no_grad():
        a = model(x.cuda())
        b = model(x.cuda())
    assert torch.equal(a, b) and rel_err(a.cpu().numpy(), oracle()) <= LL_TOL
    # the autograd route shares the layers' workspaces (same weights, other tables)
    model.train()
    model(x.cuda().requires_grad_(True)).sum().backward()
    model.eval()
    with torch.no_grad():
        c = model(x.cuda())
        assert torch.equal(a, c)
        for layer in model.layers:
            if hasattr(layer, 'weight') and layer.weight.requires_grad:
                layer.weight.add_(torch.randn_like(layer.weight))
        model.root_layer.weight.mul_(0.3)
        d = model(x.cuda())
    assert not torch.equal(a, d) and rel_err(d.cpu().numpy(), oracle()) <= LL_TOL


@pytest.mark.parametrize('D,depth,reps', [(9, 3, 5), (100, 6, 3), (15, 2, 3)])
def test_input_gradient_with_heavily_padded_region_graphs(D, depth, reps):
    """d/dx of the leaf layer when the padding is at least a region wide (pad >= d: 2^depth regions per repetition is
    then more than ceil(D / d)) -- a RatSpn used as the base density of a flow, or gradients w.r.t. the evidence."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(3)
    model = GaussianRatSpn(D, rg_depth=depth, rg_repetitions=reps, rg_batch=3, rg_sum=2, optimize_scale=True,
                           random_state=7)
    with torch.no_grad():
        model.base_layer.scale.uniform_(0.6, 1.5)
    x = torch.randn(37, D, generator=torch.Generator().manual_seed(1))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    xo = x.clone().double().requires_grad_(True)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    orc.ratspn_forward(sd64, xo).sum().backward()
    model.cuda()
    xg = x.cuda().requires_grad_(True)
    model(xg).sum().backward()
    assert grad_err(xg.grad.cpu().numpy(), xo.grad.numpy()) <= GRAD_TOL


def test_bernoulli_input_gradient(golden):
    """d/dx through Bernoulli leaves (the reference's autograd returns it: -BCEWithLogits is linear in x, ratspn.py:243),
    against the oracle's autograd in fp64; marginalised (NaN) inputs get zero gradient."""
    from deeprob.spn.models import BernoulliRatSpn
    g = golden('ratspn_bernoulli_15_d3_r4_i4_s2')
    model = BernoulliRatSpn(15, rg_depth=3, rg_repetitions=4, rg_batch=4, rg_sum=2, random_state=42)
    state_to_model(model, g, 'cuda').eval()
    x = torch.rand(41, 15, generator=torch.Generator().manual_seed(6))
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in orc.state_from_npz(g).items()}
    xo = x.double().requires_grad_(True)
    orc.ratspn_forward(sd, xo).sum().backward()
    xg = x.cuda().requires_grad_(True)
    model(xg).sum().backward()
    assert grad_err(xg.grad.cpu().numpy(), xo.grad.numpy()) <= GRAD_TOL
    xn = x.clone()
    xn[3, 4] = float('nan')
    xg = xn.cuda().requires_grad_(True)
    model(xg).sum().backward()
    assert xg.grad[3, 4].item() == 0.0 and torch.isfinite(xg.grad).all()


def test_backward_wide_model_vs_oracle(golden):
    """(16,16) backward: the reference-generated fixture of this model ships no gradients (171 k parameters), so the
    check is against the oracle's fp64 autograd -- the oracle itself is pinned to reference gradients on the other
    models (tests/test_oracle_ratspn.py).  Tolerance as in test_backward_golden: GRAD_TOL, widened to the fp32
    restatement's own distance from fp64 (the log-softmax Jacobian of the 2048 root weights cancels)."""
    model, g = build('ratspn_g784_d2_r8_i16_s16', golden)
    x = torch.from_numpy(g['x'][:8])
    names = [k for k, p in model.named_parameters() if p.requires_grad]

    def oracle(dtype):
        sd = orc.state_from_npz(g, dtype=dtype)
        for k in names:
            sd[k] = sd[k].clone().requires_grad_(True)
        xo = x.to(dtype).clone().requires_grad_(True)
        with torch.enable_grad():
            orc.ratspn_loss(orc.ratspn_forward(sd, xo), None).backward()
        out = {k: sd[k].grad.double().numpy() for k in names}
        out['x'] = xo.grad.double().numpy()
        return out

    ref64, ref32 = oracle(torch.float64), oracle(torch.float32)
    xg = x.cuda().requires_grad_(True)
    with torch.enable_grad():
        model.loss(model(xg)).backward()
    got = {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.requires_grad}
    got['x'] = xg.grad.cpu().numpy()
    for k in got:
        own = grad_err(ref32[k], ref64[k])
        tol = max(GRAD_TOL, 2.0 * own)
        report_measured('test_backward_wide_model_vs_oracle grad.%s vs fp64' % k, grad_err(got[k], ref64[k]), tol,
                        '(reference fp32 vs fp64 = %.2e)' % own)
        assert grad_err(got[k], ref64[k]) <= tol, k


@pytest.mark.parametrize('R,N,S,B,root', [(8, 8, 8, 33, False), (4, 16, 16, 70, False), (6, 5, 3, 17, False),
                                           (8, 8, 10, 33, True), (2, 16, 1, 5, True),
                                           # the single-launch level backward (dpk_prodsum_backward) at its other shapes:
                                           # several samples per wave, classes in chunks of 8, idle lanes, long tiles
                                           (8, 2, 2, 70, False), (4, 4, 8, 33, False), (32, 8, 4, 700, False),
                                           (16, 2, 3, 40, True), (8, 4, 20, 19, True), (16, 8, 1, 5000, True),
                                           # 16 nodes per region (the example model's levels): the work-group-wide instantiation
                                           (8, 16, 8, 33, False), (6, 16, 16, 300, False)])
def test_folded_level_autograd_matches_layer_chain(R, N, S, B, root):
    """ops.ProdSumFn (product + sum / root as one autograd node: folded forward, product recomputed in the backward)
    against the per-layer operators chained: values, input gradient and weight gradient."""
    from deeprob.hip import ops, Workspace
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, R, N, generator=gen) * 4
    x[0, 0, :] = float('-inf')          # a region without support for one sample (log 0 in every node)
    x[1, :, 0] = float('-inf')          # single dead nodes
    x[2] = -1.0e4                       # far tails: every exponential underflows against the maximum's neighbours
    x = x.cuda()
    P = R // 2
    w = (torch.randn((S, P * N * N) if root else (P, S, N * N), generator=gen) * 2).cuda().requires_grad_(True)
    gout = torch.randn((B, S) if root else (B, P, S), generator=gen).cuda()
    xa = x.clone().requires_grad_(True)
    prod = ops.ProductFn.apply(xa)
    ya = ops.RootFn.apply(prod, w, Workspace()) if root else ops.SumFn.apply(prod, w, Workspace())
    ga_x, ga_w = torch.autograd.grad(ya, [xa, w], gout)
    xb = x.clone().requires_grad_(True)
    yb = ops.prodsum_autograd(xb, w, Workspace(), root=root)
    assert yb is not None
    gb_x, gb_w = torch.autograd.grad(yb, [xb, w], gout)
    fin = torch.isfinite(ya)
    assert torch.equal(fin, torch.isfinite(yb)) and torch.equal(ya[~fin], yb[~fin])
    assert rel_err(yb[fin].detach().cpu().numpy(), ya[fin].detach().cpu().numpy()) <= 2e-6
    ok = torch.isfinite(ga_x)           # (where the chain itself yields nan for a -inf input, nothing is pinned)
    assert grad_err(gb_x[ok].cpu().numpy(), ga_x[ok].cpu().numpy()) <= 1e-5
    assert grad_err(gb_w.cpu().numpy(), ga_w.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i2_s2', 'ratspn_g784_d2_r8_i8_s8', 'ratspn_g784_d2_r8_i16_s16'])
def test_mixed_magnitude_evidence_per_sample(golden, name, mapping):
    """One batch whose rows span eight orders of magnitude (1e-3 ... 1e5, straddling the expansion bound 6 and the exact-path
    bound 1e3 of the matrix-core kernels): every SAMPLE is held to the 1e-5 bar against fp64, not the batch's maximum."""
    model, g = build(name, golden)
    x = torch.randn(18 * 7, 784, generator=torch.Generator().manual_seed(77))
    scales = torch.tensor([1e-3, 0.1, 1.0, 3.0, 5.5, 6.5, 30.0, 900.0, 1100.0, 3e3, 1e5, 1.0, 2.0, 4.0, 8.0, 16.0, 64.0, 256.0])
    x = x * scales.repeat_interleave(7)[:, None]
    x[5, ::5] = float('nan')
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}
    want = orc.ratspn_forward(sd64, x.double())
    with torch.no_grad():
        got = model(x.cuda()).cpu().double()
    per_sample = ((got - want).abs() / want.abs().clamp_min(1.0)).max().item()
    assert torch.isfinite(got).all() and per_sample <= LL_TOL, per_sample


@pytest.mark.parametrize('D,one_launch', [(784, True), (100, False), (256, False)])
def test_eight_channel_route_follows_the_library_envelope(D, one_launch):
    """RatSpn._prefer_folded asks the library (dpk_ratspn_forward_on_mfma) whether the one-launch 8-channel kernel takes
    THIS call instead of re-deriving its envelope: a narrow model (D <= 256: the kernel's LDS plan declines) or a
    misaligned input keeps the folded MFMA route rather than falling to the ~3x slower VALU kernel; either way the
    numbers are the oracle's."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(2)
    model = GaussianRatSpn(D, rg_depth=2, rg_repetitions=4, rg_batch=8, rg_sum=8, random_state=9).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(300, D, generator=torch.Generator().manual_seed(3))
    want = orc.ratspn_forward(sd, x).numpy()
    model = model.cuda()
    xd = x.cuda()
    assert model._prefer_folded(xd) == (not one_launch)
    with torch.no_grad():
        assert rel_err(model(xd).cpu().numpy(), want) <= LL_TOL
    # a view whose rows start 4 bytes off a 16-byte boundary: the LDS-DMA kernels decline, the folded route serves it
    if D % 4 == 0:
        xm = torch.empty(300 * D + 1, device='cuda')[1:].view(300, D)
        xm.copy_(xd)
        assert xm.data_ptr() % 16 == 4 and model._prefer_folded(xm)
        with torch.no_grad():
            assert rel_err(model(xm).cpu().numpy(), want) <= LL_TOL


@pytest.mark.parametrize('B', [1, 37, 512])
@pytest.mark.parametrize('evidence', ['plain', 'marginalised', 'huge'])
@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i8_s8', 'ratspn_g784_d2_r8_i2_s2'])
def test_training_forward_single_launch_matches_layer_chain(golden, name, B, evidence):
    """ops.RatSpnTrainFn (dpk_ratspn_forward_train: the evaluation kernel writing the leaf / sum layer outputs relative to
    the sample's quadratic term, the layers' backward kernels on those) against the per-layer autograd chain of the same
    model: log-likelihoods and every parameter gradient; NaN evidence (validity GEMM) and evidence beyond the expansion
    bound (exact per-element leaf sums) included.  reference: models/ratspn.py:105-122 under autograd."""
    model, g = build(name, golden)
    model.train()
    gen = torch.Generator().manual_seed(5 + B)
    x = torch.randn(B, 784, generator=gen)
    if evidence == 'marginalised':
        x[torch.rand(B, 784, generator=gen) < 0.2] = float('nan')
    elif evidence == 'huge':
        x[0] *= 2000.0
        x[B // 2, 5] = 1e6
    x = x.cuda()
    params = [p for p in model.parameters() if p.requires_grad]
    calls = {'n': 0}
    from deeprob.hip import ops
    real = ops.RatSpnTrainFn.apply

    def counted(*a):
        calls['n'] += 1
        return real(*a)

    ops.RatSpnTrainFn.apply = counted
    try:
        out_f = model(x)
        gf = torch.autograd.grad(model.loss(out_f), params)
    finally:
        ops.RatSpnTrainFn.apply = real
    assert calls['n'] == 1, "the training forward did not take the single-launch route"
    model._train_fused_declined = True
    out_c = model(x)
    gc = torch.autograd.grad(model.loss(out_c), params)
    model._train_fused_declined = False
    assert rel_err(out_f.detach().cpu().numpy(), out_c.detach().cpu().numpy()) <= 2e-6
    # yardstick: the oracle's fp64 autograd on the same evidence (the two fp32 routes sit ~1e-4 of the largest gradient
    # apart at B = 1 -- responsibilities exp(in - out) of leaf sums near -1000 -- each within GRAD_TOL of fp64)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    sd = orc.state_from_npz(g, dtype=torch.float64)
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    xo = x.double().cpu()
    drops = None
    if evidence == 'marginalised':
        # (the reference's own backward returns NaN where NaN evidence was touched -- test_gradients_with_marginalised_inputs;
        # the yardstick is the masked gradient: marginalised terms cut out of the graph)
        nan_mask = torch.isnan(xo)
        mask = sd['base_layer.mask']
        drops = {'leaf': nan_mask[:, mask].unsqueeze(2).expand(-1, -1, sd['base_layer.loc'].shape[1], -1)}
        xo = torch.where(nan_mask, torch.zeros_like(xo), xo)
    with torch.enable_grad():
        orc.ratspn_loss(orc.ratspn_forward(sd, xo, drops=drops), None).backward()
    for k, p, a, b in zip(names, params, gf, gc):
        assert torch.isfinite(a).all()
        exact = sd[k].grad.numpy()
        assert grad_err(a.cpu().numpy(), exact) <= max(GRAD_TOL, 2.0 * grad_err(b.cpu().numpy(), exact)), k
        assert grad_err(a.cpu().numpy(), b.cpu().numpy()) <= 4 * GRAD_TOL, k
    with torch.no_grad():                       # and the evaluation path agrees with the training forward's values
        assert rel_err(model.eval()(x).cpu().numpy(), out_f.detach().cpu().numpy()) <= 2e-6


def test_training_forward_classes_and_stale_tables(golden):
    """10 classes; then a parameter written through .data between two training forwards (no version bump): the second
    forward's in-launch table check finds it and both values and gradients follow the new parameters.  Yardstick: the
    oracle's fp64 autograd (the cross-entropy gradient cancels: the fp32 restatement itself sits up to ~1e-3 from fp64)."""
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(0)
    model = GaussianRatSpn(784, out_classes=10, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=3).cuda().train()
    x = torch.randn(70, 784, device='cuda')
    y = torch.randint(0, 10, (70,), device='cuda')
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = [p for k, p in model.named_parameters() if p.requires_grad]

    def oracle(dtype):
        sd = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}
        for k in names:
            sd[k] = sd[k].clone().requires_grad_(True)
        with torch.enable_grad():
            out = orc.ratspn_forward(sd, x.cpu().to(dtype))
            orc.ratspn_loss(out, y.cpu()).backward()
        return out.detach().double().numpy(), [sd[k].grad.double().numpy() for k in names]

    for trial in range(2):
        out64, g64 = oracle(torch.float64)
        _, g32 = oracle(torch.float32)
        for declined in (False, True):
            model._train_fused_declined = declined
            out = model(x)
            grads = torch.autograd.grad(model.loss(out, y), params)
            assert rel_err(out.detach().cpu().numpy(), out64) <= LL_TOL
            for k, a, e64, e32 in zip(names, grads, g64, g32):
                assert grad_err(a.cpu().numpy(), e64) <= max(GRAD_TOL, 4.0 * grad_err(e32, e64)), (k, declined, trial)
        model._train_fused_declined = False
        model.base_layer.loc.data.add_(0.3)
        model.root_layer.weight.data.mul_(0.5)


@pytest.mark.parametrize('name', ['ratspn_g784_d2_r8_i8_s8', 'ratspn_g784_d2_r8_i2_s2'])
def test_training_forward_golden_gradients(golden, name):
    """The reference's own parameter gradients (fixture generated by importing the reference) through the single-launch
    training forward: test_backward_golden asks for the input gradient too, which keeps it on the layer chain."""
    model, g = build(name, golden)
    x = torch.from_numpy(g['x']).cuda()
    y = torch.from_numpy(g['y']).cuda() if 'y' in g.files else None
    assert model._forward_train_fused(x) is not None
    with torch.enable_grad():
        loss = model.loss(model(x), y)
        loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= LL_TOL * max(1.0, abs(float(g['loss'])))
    ref64 = _oracle_grads_fp64(g)
    for k, p in model.named_parameters():
        if 'grad.' + k in g.files:
            own = grad_err(g['grad.' + k], ref64[k])
            tol = max(GRAD_TOL, 2.0 * own)
            err = grad_err(p.grad.cpu().numpy(), ref64[k])
            report_measured('test_training_forward_golden_gradients[%s] grad.%s vs fp64' % (name, k), err, tol,
                            '(golden = reference fp32 vs fp64: %.2e)' % own)
            assert err <= tol, k


@pytest.mark.parametrize('B', [512, 2048, 3000])
@pytest.mark.parametrize('regime', ['mu_over_s_50', 'uint8_range', 'clusters'])
def test_leaf_parameter_gradients_ill_conditioned(regime, B):
    """the moment-GEMM leaf backward (B <= 2048) on data / parameters with |mu| / s large -- small learned
    scales around means far from 0, uint8-range evidence, and channels sitting on clusters many scales apart (what no
    per-variable pivot can centre: the kernel has to notice and evaluate the direct form).  d/dloc and d/dscale against
    the fp64 oracle at GRAD_TOL, B = 3000 (the vector-ALU kernel) as the control."""
    from deeprob.spn.models import GaussianRatSpn
    gen = torch.Generator().manual_seed(B + len(regime))
    model = GaussianRatSpn(64, rg_depth=2, rg_repetitions=4, rg_batch=8, rg_sum=4, optimize_scale=True, random_state=3).eval()
    I = model.base_layer.loc.shape[1]
    with torch.no_grad():
        if regime == 'mu_over_s_50':
            centre = 5.0 + torch.randn(64, generator=gen)
            x = centre + 0.1 * torch.randn(B, 64, generator=gen)
            spread, scale = 0.05, 0.1
        elif regime == 'uint8_range':
            centre = 40.0 + 170.0 * torch.rand(64, generator=gen)
            x = (centre + 3.0 * torch.randn(B, 64, generator=gen)).round().clamp(0, 255)
            spread, scale = 2.0, 3.0
        else:
            centre = torch.zeros(64)
            offs = torch.tensor([0.0, 60.0, 128.0, 200.0])
            x = offs[torch.randint(0, 4, (B, 1), generator=gen)] + torch.randn(B, 64, generator=gen)    # (a sample = one cluster)
            spread, scale = 0.5, 1.0
        loc = torch.empty_like(model.base_layer.loc)
        mask = model.base_layer.mask                      # [regions, positions] -> variable
        for k in range(I):
            base = centre[mask.clamp_min(0)]
            if regime == 'clusters':
                base = base + offs[k % 4]
            loc[:, k] = base + spread * torch.randn(base.shape, generator=gen)
        model.base_layer.loc.copy_(loc)
        model.base_layer.scale.fill_(scale).mul_(1.0 + 0.2 * torch.rand(model.base_layer.scale.shape, generator=gen))
    ref = {}
    for dt in (torch.float64, torch.float32):
        sd = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
        for k in ('base_layer.loc', 'base_layer.scale'):
            sd[k].requires_grad_(True)
        with torch.enable_grad():
            orc.ratspn_loss(orc.ratspn_forward(sd, x.to(dt))).backward()
        ref[dt] = {k: sd[k].grad.double().numpy() for k in ('base_layer.loc', 'base_layer.scale')}
    model.cuda()
    with torch.enable_grad():
        model.loss(model(x.cuda())).backward()
    for k, p in (('base_layer.loc', model.base_layer.loc), ('base_layer.scale', model.base_layer.scale)):
        own = grad_err(ref[torch.float32][k], ref[torch.float64][k])
        tol = max(GRAD_TOL, 2.0 * own)
        err = grad_err(p.grad.cpu().numpy(), ref[torch.float64][k])
        report_measured('test_leaf_parameter_gradients_ill_conditioned[%s, B=%d] grad.%s vs fp64' % (regime, B, k), err, tol,
                        '(reference fp32 arithmetic vs fp64: %.2e)' % own)
        assert err <= tol, (k, err)

def test_eight_channel_blocks_shared_by_two_work_groups_agree_with_whole_blocks():
    """Round 5: up to 128 blocks of 32 samples the 8-channel kernel gives a block to two work-groups (four repetitions each,
    root partials merged through the workspace on a ticket that is only counted up); larger launches keep one work-group
    per block.  The same samples through both forms, launches of different sizes interleaved (the tickets of a block are
    drawn twice per launch whatever ran before), a ragged last block, against each other and the oracle."""
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(31)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=8, rg_sum=8, random_state=42).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = torch.randn(6001, 784, generator=torch.Generator().manual_seed(32))
    xd = x.cuda()
    with torch.no_grad():
        whole = model(xd)                                   # 188 blocks: one work-group per block
        parts = []
        for lo, hi in ((0, 2000), (2000, 2977), (2977, 6001)):   # 63, 31 (ragged) and 95 blocks: shared
            parts.append(model(xd[lo:hi]))
            model(xd[:50])                                  # (a two-block launch in between)
        chunks = torch.cat(parts)
        again = model(xd[2000:2977])
    rows = torch.randint(0, 6001, (64,), generator=torch.Generator().manual_seed(33))
    want = orc.ratspn_forward(sd, x[rows]).numpy()
    assert rel_err(chunks[rows.cuda()].cpu().numpy(), want) <= 1e-5
    assert rel_err(whole[rows.cuda()].cpu().numpy(), want) <= 1e-5
    assert rel_err(chunks.cpu().numpy(), whole.cpu().numpy()) <= 2e-6
    assert torch.equal(again, parts[1])                      # the same launch twice: bit for bit


def test_topdown_edge_cases():
    """The one-launch top-down pass at its edges: an empty batch, nothing to complete (no NaN: evidence returned as it is),
    everything to complete (all NaN = the unconditional MPE state: every row the same completion), labels of another
    integer dtype, a CPU tensor (no silent fallback), more samples than one grid pass of the kernel."""
    from deeprob.hip import HipError
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(1)
    model = GaussianRatSpn(40, out_classes=3, rg_depth=2, rg_repetitions=4, rg_batch=4, rg_sum=3, random_state=2).cuda().eval()
    assert tuple(model.mpe(torch.empty(0, 40, device='cuda')).shape) == (0, 40)
    assert tuple(model.sample(0).shape) == (0, 40)
    x = torch.randn(9, 40, device='cuda')
    assert torch.equal(model.mpe(x), x)
    allnan = torch.full((5, 40), float('nan'), device='cuda')
    y = torch.tensor([0, 1, 2, 1, 1], dtype=torch.int32, device='cuda')
    full = model.mpe(allnan, y=y)
    assert not torch.isnan(full).any() and torch.equal(full[1], full[3]) and torch.equal(full[1], full[4])
    assert torch.equal(full, model._mpe_layerwise(allnan, y=y.long()))
    with pytest.raises((HipError, TypeError, ValueError, RuntimeError)):
        model.mpe(torch.randn(3, 40))
    yb = torch.arange(300000, device='cuda') % 3
    big = model.sample(300000, y=yb, seed=5)    # > 256 compute units x 16 work-groups x 4 samples: the grid-stride loop
    assert tuple(big.shape) == (300000, 40) and torch.isfinite(big).all()
    assert torch.equal(big[:1000], model.sample(1000, y=yb[:1000], seed=5))    # (a draw depends on (seed, sample index) only)
