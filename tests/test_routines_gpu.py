"""The reference's train / test entry points (deeprob/torch/routines.py) driving the HIP models."""
import numpy as np
import pytest
import torch

from oracle import ratspn_oracle as orc

pytestmark = pytest.mark.gpu


def test_train_and_test_model_ratspn(tmp_path):
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.torch.routines import train_model, test_model
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    centre = torch.randn(20, generator=gen)
    train = (centre + 0.5 * torch.randn(600, 20, generator=gen)).numpy()
    valid = (centre + 0.5 * torch.randn(150, 20, generator=gen)).numpy()
    model = GaussianRatSpn(20, rg_depth=2, rg_repetitions=4, rg_batch=4, rg_sum=4, optimize_scale=True, in_dropout=0.1,
                           random_state=42)
    hist = train_model(model, train, valid, setting='generative', lr=2e-2, batch_size=100, epochs=8, patience=3,
                       checkpoint=str(tmp_path / 'ck.pt'), verbose=False)
    assert set(hist) == {'train', 'valid'} and len(hist['train']) == len(hist['valid']) <= 8
    assert hist['valid'][-1] < hist['valid'][0] - 1.0 and np.isfinite(hist['train']).all()
    mean_ll, two_se = test_model(model, valid, setting='generative', batch_size=64, verbose=False)
    # same numbers as the reference's recipe (np.mean / 2 np.std / sqrt(n)) on the oracle's log-likelihoods
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    lls = orc.ratspn_forward(sd, torch.from_numpy(valid)).double().numpy().reshape(-1)
    assert abs(mean_ll - lls.mean()) <= 1e-5 * abs(lls.mean())
    assert abs(two_se - 2.0 * lls.std() / np.sqrt(lls.size)) <= 1e-3 * two_se
    assert float(model.base_layer.scale.min()) >= 1e-5          # ScaleClipper ran (apply_constraints)
    with pytest.raises(ValueError):
        train_model(model, train, valid, setting='unsupervised')


def test_train_flow_with_batch_norm(tmp_path):
    from deeprob.flows.models import RealNVP1d
    from deeprob.torch.routines import train_model, test_model
    torch.manual_seed(2)
    data = (torch.randn(512, 16) * torch.linspace(0.3, 2.0, 16) + 1.0).numpy()
    flow = RealNVP1d(16, n_flows=2, units=32)
    before = test_model(flow, data[:128], verbose=False)[0]
    hist = train_model(flow, data[:384], data[384:], lr=1e-2, batch_size=128, epochs=10, patience=10,
                       checkpoint=str(tmp_path / 'ck.pt'), verbose=False)
    after = test_model(flow, data[:128], verbose=False)[0]
    assert after > before + 1.0 and hist['train'][-1] < hist['train'][0]


def test_discriminative_setting(tmp_path):
    """train_model / test_model with setting='discriminative' (reference :213-346, :429-478) on a 3-class RAT-SPN."""
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.torch.routines import train_model, test_model
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(3)
    centres = 2.0 * torch.randn(3, 12, generator=gen)
    y = torch.randint(0, 3, (900,), generator=gen)
    x = centres[y] + 0.7 * torch.randn(900, 12, generator=gen)
    ds = torch.utils.data.TensorDataset(x, y)
    tr, va = torch.utils.data.random_split(ds, [700, 200], generator=gen)
    model = GaussianRatSpn(12, out_classes=3, rg_depth=1, rg_repetitions=4, rg_batch=4, rg_sum=4, random_state=1)
    hist = train_model(model, tr, va, setting='discriminative', lr=5e-2, batch_size=100, epochs=6, patience=6,
                       checkpoint=str(tmp_path / 'ck.pt'), verbose=False)
    assert set(hist['train']) == {'loss', 'accuracy'} and len(hist['valid']['accuracy']) == len(hist['train']['loss'])
    assert hist['valid']['accuracy'][-1] > 0.8 and hist['train']['loss'][-1] < hist['train']['loss'][0]
    nll, report = test_model(model, va, setting='discriminative', verbose=False)
    assert nll < 0.8 and report['accuracy'] > 0.8 and set(report) >= {'0', '1', '2', 'accuracy', 'macro avg'}


def test_train_model_with_hip_graph_matches_eager(tmp_path):
    """The optimisation step captured as a HIP graph (deeprob.hip.graphs) trains like the eager loop: same history
    shape, losses equal up to the split-K summation order, the ragged last batch handled eagerly; dropout refuses."""
    from deeprob.flows.models import RealNVP1d
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.torch.routines import train_model
    gen = torch.Generator().manual_seed(0)
    train = (torch.randn(5 * 64 + 17, 24, generator=gen) * 0.7 + 0.5).numpy()
    valid = (torch.randn(96, 24, generator=gen) * 0.7 + 0.5).numpy()

    def run(hip_graph):
        torch.manual_seed(1)
        flow = RealNVP1d(24, n_flows=2, units=32)
        hist = train_model(flow, train, valid, setting='generative', lr=5e-3, batch_size=64, epochs=3, patience=5,
                           checkpoint=str(tmp_path / ('g.pt' if hip_graph else 'e.pt')), drop_last=False, verbose=False,
                           hip_graph=hip_graph)
        return hist, flow

    torch.manual_seed(7)            # same shuffling for both runs
    eager, _ = run(False)
    torch.manual_seed(7)
    graphed, flow = run(True)
    assert len(graphed['train']) == len(eager['train']) == 3
    assert np.allclose(graphed['train'], eager['train'], rtol=2e-3)
    assert np.allclose(graphed['valid'], eager['valid'], rtol=2e-3)
    assert graphed['train'][-1] < graphed['train'][0]
    spn = GaussianRatSpn(24, rg_depth=1, rg_repetitions=2, rg_batch=2, rg_sum=2, in_dropout=0.2, random_state=1)
    with pytest.raises(NotImplementedError):
        train_model(spn, train, valid, epochs=1, batch_size=64, checkpoint=str(tmp_path / 'd.pt'), verbose=False,
                    hip_graph=True)
    with pytest.raises(ValueError):
        train_model(flow, train, valid, setting='discriminative', epochs=1, hip_graph=True)


def test_train_image_flow(tmp_path):
    """train_model / test_model on a RealNVP2d (SURVEY 8f-1 x 8f-3): training-mode batch statistics and the 2-D backward
    kernels under the reference's loop, evaluation kernels for validation and testing."""
    from deeprob.flows.models import RealNVP2d
    from deeprob.torch.routines import train_model, test_model
    torch.manual_seed(4)
    data = (0.5 * torch.randn(320, 1, 8, 8) + 0.7).numpy()
    flow = RealNVP2d((1, 8, 8), n_flows=1, n_blocks=1, channels=8)
    before = test_model(flow, data[:64], verbose=False)[0]
    hist = train_model(flow, data[:256], data[256:], lr=2e-3, batch_size=64, epochs=6, patience=6,
                       checkpoint=str(tmp_path / 'ck2d.pt'), verbose=False)
    after = test_model(flow, data[:64], verbose=False)[0]
    assert np.isfinite(hist['train']).all() and hist['train'][-1] < hist['train'][0] and after > before


@pytest.mark.parametrize('optimizer,kwargs', [('sgd', None), ('rmsprop', None), ('adagrad', None), ('adam', None),
                                              ('adam', {'foreach': True}), ('sgd', {'fused': False})])
def test_train_model_every_reference_optimizer(tmp_path, optimizer, kwargs):
    """Every optimiser name the reference accepts (torch/utils.py:32-49) takes a few steps through train_model: the
    fused-kernel default applies only where torch has HIP fused kernels (Adagrad's are CPU-only, RMSprop has none) and
    steps aside for `foreach`."""
    from deeprob.spn.models import GaussianRatSpn
    from deeprob.torch.routines import train_model
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(5)
    data = (0.5 * torch.randn(260, 12, generator=gen) + 0.3).numpy()
    model = GaussianRatSpn(12, rg_depth=1, rg_repetitions=2, rg_batch=2, rg_sum=2, random_state=3)
    hist = train_model(model, data[:200], data[200:], lr=1e-2, batch_size=50, epochs=2, patience=2, optimizer=optimizer,
                       optimizer_kwargs=kwargs, checkpoint=str(tmp_path / 'o.pt'), verbose=False)
    assert len(hist['train']) == 2 and np.isfinite(hist['train']).all() and np.isfinite(hist['valid']).all()


@pytest.mark.gpu
@pytest.mark.parametrize('kwargs', [{}, {'weight_decay': 0.01, 'betas': (0.8, 0.99)}, {'maximize': True, 'eps': 1e-6}])
def test_fused_adam_follows_torch_adam(kwargs):
    """deeprob.hip.optim.FusedAdam (one launch for all tensors, device-side step count) against torch.optim.Adam on the
    same gradients over 25 steps, ragged tensor sizes included (1 element, 2047, 2048, 2049, 70 001); fp32 round-off
    only.  reference: the optimiser of torch/routines.py:164."""
    from deeprob.hip.optim import FusedAdam
    torch.manual_seed(3)
    sizes = [(1,), (2047,), (2048,), (2049,), (7, 10001), (3, 5, 11)]
    ours = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in sizes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a, b = FusedAdam(ours, lr=3e-3, **kwargs), torch.optim.Adam(theirs, lr=3e-3, **kwargs)
    for it in range(25):
        for p, q in zip(ours, theirs):
            g = torch.randn_like(p) * (1.0 + it % 3)
            p.grad, q.grad = g.clone(), g.clone()
        a.step(), b.step()
    for p, q in zip(ours, theirs):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-6), float((p - q).abs().max())
    assert float(a.state[ours[0]]['step']) == 25.0
    # a tensor without a gradient is skipped, like torch does
    ours[1].grad = None
    before = ours[1].detach().clone()
    a.step()
    assert torch.equal(ours[1], before)


@pytest.mark.gpu
def test_build_optimizer_choices():
    from deeprob.torch.routines import build_optimizer
    from deeprob.hip.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.randn(5, device='cuda'))]
    assert isinstance(build_optimizer('adam', ps, 1e-3, {'fused': True}), FusedAdam)
    assert type(build_optimizer('adam', ps, 1e-3, {'fused': False})) is torch.optim.Adam
    assert type(build_optimizer('adam', ps, 1e-3, {'fused': True, 'amsgrad': True})) is torch.optim.Adam
    assert type(build_optimizer('adam', ps * 1 + [torch.nn.Parameter(torch.randn(2, device='cuda')) for _ in range(100)],
                                1e-3, {'fused': True})) is torch.optim.Adam
    assert type(build_optimizer('sgd', ps, 1e-3, {'fused': True})) is torch.optim.SGD


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 1), (512, 1), (4097, 1), (100000,)])
def test_neg_mean_loss_op(shape):
    """ops.neg_mean = -torch.mean (models/ratspn.py:184-191): value (fp64 accumulation) and gradient, incl. -inf entries."""
    from deeprob.hip import ops
    x = (torch.randn(shape, device='cuda') * 50 - 700).requires_grad_(True)
    got = ops.neg_mean(x)
    want = -torch.mean(x.detach().double())
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    (g,) = torch.autograd.grad(got * 3.0, x)
    assert torch.allclose(g, torch.full_like(g, -3.0 / x.numel()), rtol=1e-6, atol=0)
    y = x.detach().clone()
    y.view(-1)[0] = float('-inf')
    assert float(ops.neg_mean(y)) == float('inf')


@pytest.mark.gpu
def test_fused_adam_state_dict_round_trip():
    """The device-side step count lives under torch.optim.Adam's state name and survives state_dict() / load_state_dict():
    a resumed optimiser continues with the right bias correction (compared with torch.optim.Adam run straight through)."""
    from deeprob.hip.optim import FusedAdam
    torch.manual_seed(4)
    ours = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in [(33,), (5, 7)]]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a, b = FusedAdam(ours, lr=1e-2), torch.optim.Adam(theirs, lr=1e-2)
    grads = [[torch.randn_like(p) for p in ours] for _ in range(9)]
    for it in range(5):
        for p, q, g in zip(ours, theirs, grads[it]):
            p.grad, q.grad = g.clone(), g.clone()
        a.step(), b.step()
    sd = a.state_dict()
    resumed = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a2 = FusedAdam(resumed, lr=1e-2)
    a2.load_state_dict(sd)
    for it in range(5, 9):
        for p, q, g in zip(resumed, theirs, grads[it]):
            p.grad, q.grad = g.clone(), g.clone()
        a2.step(), b.step()
    for p, q in zip(resumed, theirs):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-6)
    assert float(a2.state[resumed[0]]['step']) == 9.0


def test_fused_adam_one_step_count_per_group_is_enforced():
    """FusedAdam keeps ONE step count per parameter group (torch.optim.Adam: one per parameter).  A parameter that joins after
    the group's first step, or a loaded state with differing per-parameter steps, would silently get other bias
    corrections than the optimiser train_model('adam') used to build: both raise instead."""
    from deeprob.hip.optim import FusedAdam
    from deeprob.hip import HipError
    a = torch.nn.Parameter(torch.randn(300, device='cuda'))
    b = torch.nn.Parameter(torch.randn(200, device='cuda'))
    opt = FusedAdam([a, b], lr=1e-2)
    a.grad = torch.randn_like(a)
    opt.step()                                   # only `a` has a gradient: the group's schedule starts with `a` alone
    a.grad = torch.randn_like(a)
    b.grad = torch.randn_like(b)
    with pytest.raises(HipError):
        opt.step()                               # `b` joins late
    ref = torch.optim.Adam([a, b], lr=1e-2, capturable=True)
    a.grad = torch.randn_like(a)
    b.grad = None
    ref.step()
    b.grad = torch.randn_like(b)
    ref.step()                                   # torch: a at step 2, b at step 1
    with pytest.raises(ValueError):
        FusedAdam([a, b], lr=1e-2).load_state_dict(ref.state_dict())


def test_graphed_train_step_is_keyed_on_the_shard_sizes():
    """A captured sharded step bakes the whole-batch element count of the synchronised batch norms into kernel scalars: a
    batch with the same LOCAL shape but another global size (511 rows on two ranks: 256 + 255) must run eagerly, or the
    replicas diverge."""
    from deeprob import parallel
    from deeprob.flows.models import RealNVP1d
    from deeprob.hip.graphs import GraphedTrainStep
    from deeprob.torch.routines import build_optimizer
    torch.manual_seed(0)
    flow = RealNVP1d(16, n_flows=2, units=32).cuda().train()
    opt = build_optimizer('adam', list(flow.parameters()), 1e-3, dict(fused=True, capturable=True))
    step = GraphedTrainStep(flow, opt, warmup=1)
    x = torch.randn(64, 16, device='cuda')
    try:
        parallel.set_shard_sizes(64, 128)
        for _ in range(3):
            step(x)
        assert step.graph is not None
        replays = []
        orig = step.graph.replay
        step.graph.replay = lambda: (replays.append(1), orig())[1]
        step(x)
        assert len(replays) == 1                  # same shape, same sizes: replayed
        parallel.set_shard_sizes(64, 127)
        step(x)
        assert len(replays) == 1                  # same local shape, other global size: eager
        parallel.set_shard_sizes(64, 128)
        step(x)
        assert len(replays) == 2
    finally:
        parallel.set_shard_sizes(None)

def test_leaf_parameter_gradients_of_a_batch_are_the_sum_over_its_halves():
    """Round 5: up to 2048 samples the leaf layer's parameter gradients are moment GEMMs on the matrix cores
    (csrc/ratspn_layers.hip: leaf_bwd_moment_kernel), larger batches keep the vector-ALU kernel with atomics.  The gradient
    of a sum over samples is additive: 3000 samples in one call (the second) against two calls of 1500 (the first), with
    marginalised entries, trainable scales, means and scales away from their initial values."""
    from deeprob.spn.models import GaussianRatSpn
    leaf = GaussianRatSpn(784, rg_depth=2, rg_repetitions=4, rg_batch=8, rg_sum=8, optimize_scale=True,
                          random_state=42).cuda().train().base_layer
    gen = torch.Generator().manual_seed(41)
    with torch.no_grad():
        leaf.loc.copy_(torch.randn(leaf.loc.shape, generator=gen))
        leaf.scale.copy_(0.5 + torch.rand(leaf.scale.shape, generator=gen))
    x = torch.randn(3000, 784, generator=gen)
    x[torch.rand(x.shape, generator=gen) < 0.05] = float('nan')
    xd = x.cuda()
    g = torch.randn(3000, leaf.loc.shape[0], 8, generator=gen).cuda()

    def grads(lo, hi):
        leaf.zero_grad(set_to_none=True)
        out = leaf(xd[lo:hi])
        out.backward(g[lo:hi])
        return leaf.loc.grad.clone(), leaf.scale.grad.clone()

    l_all, s_all = grads(0, 3000)
    l_a, s_a = grads(0, 1500)
    l_b, s_b = grads(1500, 3000)
    for whole, parts in ((l_all, l_a + l_b), (s_all, s_a + s_b)):
        scale = whole.abs().max().item()
        assert scale > 0 and (whole - parts).abs().max().item() <= 2e-5 * scale
