"""Writes through ``param.data`` move neither the address nor the version counter of a parameter, so a host-side cache
key cannot see them (round-2 verdict: the MFMA routes kept evaluating with stale tables).  Every table cache of the path
is now checked on the device (DPK_FLAG_PARAMS_VERIFY: a fingerprint of the live parameter bytes gates the table
kernels) or rebuilt per call; these tests mutate parameters and statistics through ``.data`` between two calls and
compare the second call with the oracle on the mutated state."""
import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _state(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize('kw,B', [(dict(rg_batch=2, rg_sum=2), 300), (dict(rg_batch=2, rg_sum=2), 20000),
                                  (dict(rg_batch=8, rg_sum=8), 300), (dict(rg_batch=16, rg_sum=16), 200)],
                         ids=['fused-small-batch', 'fused-ring', 'folded-8', 'folded-16'])
def test_ratspn_tables_follow_data_writes(kw, B):
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(3)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42, **kw).cuda().eval()
    x = torch.randn(B, 784, generator=torch.Generator().manual_seed(4))
    xd = x.cuda()
    with torch.no_grad():
        a = model(xd)
        versions = [p._version for p in model.parameters()]
        model.base_layer.loc.data.add_(0.05)                       # a hand-written SGD step
        model.root_layer.weight.data.mul_(0.5)
        for layer in model.layers:
            if hasattr(layer, 'weight'):
                layer.weight.data.add_(torch.randn_like(layer.weight) * 0.5)
        assert [p._version for p in model.parameters()] == versions   # the premise: no counter moved
        b = model(xd)
        c = model(xd)
        e = model(xd)
    n = min(B, 300)
    want = orc.ratspn_forward(_state(model), x[:n]).numpy()
    assert not torch.equal(a, b)
    # b: the call that FINDS the tables stale.  The 32-sample kernels check inside their own launch and evaluate that one
    # call on the table-free exact route while the tables are rebuilt (round 4); the ring kernels rebuild first.  Either
    # way it is right, and so is everything after it -- on the rebuilt tables, bit for bit the same from then on.
    assert rel_err(b[:n].cpu().numpy(), want) <= TOL
    assert rel_err(c[:n].cpu().numpy(), want) <= TOL
    assert rel_err(c.cpu().numpy(), b.cpu().numpy()) <= 2e-6
    assert torch.equal(c, e)                                       # unchanged bytes: same tables, same result
    # a bound plan follows too; with static_params the caller has waived the check (documented), so only the default
    if kw['rg_batch'] == 2:
        plan = model.fused_plan(xd)
        with torch.no_grad():
            p1 = plan.run().clone()
            model.base_layer.loc.data.sub_(0.02)
            p2 = plan.run().clone()
        assert torch.equal(p1, c)
        assert rel_err(p2[:n].cpu().numpy(), orc.ratspn_forward(_state(model), x[:n]).numpy()) <= TOL


def test_leaf_layer_tables_follow_data_writes():
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(5)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=4, rg_sum=4, random_state=42).cuda().eval()
    x = torch.randn(200, 784, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        a = model.base_layer(x.cuda())
        model.base_layer.loc.data.mul_(1.1)
        b = model.base_layer(x.cuda())
    sd = _state(model)
    want = orc.gaussian_leaf(x, sd['base_layer.mask'], sd.get('base_layer.pad_mask'), sd['base_layer.loc'],
                             sd['base_layer.scale']).numpy()
    assert not torch.equal(a, b)
    assert rel_err(b.cpu().numpy(), want) <= TOL


def test_dgcspn_tables_follow_data_writes():
    from deeprob.spn.models import DgcSpn
    from oracle import dgcspn_oracle as dorc
    torch.manual_seed(7)
    model = DgcSpn((1, 28, 28), n_batch=8, sum_channels=8, depthwise=True, n_pooling=0).cuda().eval()
    plan = dorc.schedule((1, 28, 28), 8, 8, True, 0)
    for B in (6, 300):     # batch-independent kernels / streaming kernels (from 256 samples)
        x = torch.randn(B, 1, 28, 28, generator=torch.Generator().manual_seed(B))
        with torch.no_grad():
            a = model(x.cuda())
            for name, p in model.named_parameters():
                if 'weight' in name:
                    p.data.add_(torch.randn_like(p) * 0.3)
            b = model(x.cuda())
        want = dorc.dgcspn_forward(_state(model), x[:6], plan).detach().numpy()
        assert not torch.equal(a, b)
        assert rel_err(b[:6].cpu().numpy(), want) <= TOL


def test_realnvp1d_tables_follow_data_writes():
    from deeprob.flows.models import RealNVP1d
    from oracle import flows_oracle as forc
    from tests.util import randomise_flow
    torch.manual_seed(8)
    flow = RealNVP1d(784)
    randomise_flow(flow, 9)
    flow = flow.cuda().eval()
    x = torch.randn(130, 784, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        a = flow(x.cuda())
        for name, t in list(flow.named_parameters()) + list(flow.named_buffers()):
            if name.endswith('network.0.weight') or name.endswith('network.2.bias'):
                t.data.mul_(1.05)                   # conditioner weights / biases of every coupling
            elif name.endswith('running_var'):
                t.data.mul_(1.3)                    # batch-norm statistics folded into the next coupling's tables
            elif name.endswith('running_mean'):
                t.data.add_(0.1)
        b = flow(x.cuda())
        c = flow(x.cuda())
    want = forc.flow_log_prob(_state(flow), x).numpy()
    assert not torch.equal(a, b)
    assert rel_err(b.cpu().numpy(), want) <= TOL
    assert torch.equal(b, c)


def test_realnvp2d_tables_follow_data_writes():
    from oracle import flows2d_oracle as f2orc
    from tests.util import flow2d_model
    model = flow2d_model((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=8), 11).cuda()
    x = torch.randn(5, 3, 8, 8, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        a = model(x.cuda())
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if name.endswith('weight_g') or name.endswith('running_var'):
                t.data.mul_(1.2)
        b = model(x.cuda())
    want = f2orc.log_prob(_state(model), x).numpy()
    assert not torch.equal(a, b)
    assert rel_err(b.cpu().numpy(), want) <= TOL


def test_trusting_the_version_counters_is_opt_in():
    """hip.trust_version_counters(True) restores the unchecked fast path: the stale result is then the caller's choice."""
    from deeprob import hip
    from deeprob.spn.models import GaussianRatSpn
    torch.manual_seed(13)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
    x = torch.randn(64, 784, device='cuda')
    prev = hip.trust_version_counters(True)
    try:
        with torch.no_grad():
            a = model(x)
            a = model(x)
            model.base_layer.loc.data.add_(0.5)
            stale = model(x)
        assert torch.equal(a, stale)
    finally:
        hip.trust_version_counters(prev)
    with torch.no_grad():
        fresh = model(x)
    assert not torch.equal(a, fresh)


@pytest.mark.parametrize('kw,D', [(dict(rg_batch=2, rg_sum=2, rg_repetitions=8), 784),
                                  (dict(rg_batch=4, rg_sum=4, rg_repetitions=5, out_classes=3), 200),
                                  (dict(rg_batch=8, rg_sum=8, rg_repetitions=8), 784),
                                  (dict(rg_batch=8, rg_sum=4, rg_repetitions=3, out_classes=5), 400)],
                         ids=['i2s2', 'i4s4c3', 'i8s8', 'i8s4c5'])
@pytest.mark.parametrize('what', ['loc', 'sum', 'root', 'scale'])
def test_in_launch_table_check_per_parameter(kw, D, what):
    """Round 4: the 32-sample kernels check their parameter tables inside their own launch (leading table work-groups
    publish one verdict, csrc/ratspn_gemm_prep.h).  One parameter kind at a time is written through ``.data``; the call
    that finds the tables stale (table-free exact route), the calls after it (rebuilt tables) and calls at other batch
    sizes in between (the verdict counters are left as found whatever the grid) all match the oracle."""
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(11)
    model = GaussianRatSpn(D, rg_depth=2, random_state=42, **kw).cuda().eval()
    gen = torch.Generator().manual_seed(12)
    xs = {B: torch.randn(B, D, generator=gen) for B in (77, 300, 1000)}
    xd = {B: v.cuda() for B, v in xs.items()}
    with torch.no_grad():
        for B in xd:
            model(xd[B])
        for step in range(3):
            if what == 'loc':
                model.base_layer.loc.data.add_(0.03 * (step + 1))
            elif what == 'sum':
                for layer in model.layers:
                    if hasattr(layer, 'weight'):
                        layer.weight.data.add_(torch.randn_like(layer.weight))
            elif what == 'root':
                model.root_layer.weight.data.add_(torch.randn_like(model.root_layer.weight))
            else:
                model.base_layer.scale.data.mul_(1.0 + 0.1 * torch.rand_like(model.base_layer.scale))
            sd = _state(model)
            order = (300, 77, 1000, 300) if step % 2 == 0 else (1000, 300, 77)
            for B in order:
                got = model(xd[B]).cpu().numpy()
                assert rel_err(got, orc.ratspn_forward(sd, xs[B]).numpy()) <= TOL, (step, B)


def test_in_launch_table_check_under_graph_replay():
    """A captured ``model(x)`` call carries its table check with it: a ``.data`` write between two replays is seen by the
    next replay (exact route), the one after it runs on the rebuilt tables -- no host involvement."""
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    # (9001 samples: the slice mapping, whose check is shared out over the eighth waves of its persistent work-groups)
    for I, S, B in ((2, 2, 4096), (8, 8, 4096), (2, 2, 9001)):
        torch.manual_seed(13)
        model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=I, rg_sum=S, random_state=42).cuda().eval()
        x = torch.randn(B, 784, generator=torch.Generator().manual_seed(14))
        xd = x.cuda()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.stream(side):
            model(xd)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                out = model(xd)
        torch.cuda.synchronize()
        for step in range(3):
            model.base_layer.loc.data.add_(0.02)
            model.root_layer.weight.data.mul_(1.3)
            want = orc.ratspn_forward(_state(model), x[:256]).numpy()
            g.replay()
            torch.cuda.synchronize()
            first = out.clone()
            g.replay()
            torch.cuda.synchronize()
            second = out.clone()
            g.replay()
            torch.cuda.synchronize()
            assert rel_err(first[:256].cpu().numpy(), want) <= TOL
            assert rel_err(second[:256].cpu().numpy(), want) <= TOL
            assert torch.equal(second, out)


@pytest.mark.parametrize('what', ['loc', 'sum', 'root', 'scale'])
def test_in_launch_table_check_slice_mapping_across_grids(what):
    """Round 5: the persistent 32-sample-block kernel (csrc/ratspn_gemm_slice.hip, 7681 samples and up) checks its tables in
    its own launch on counters that only grow: launches with different grids (241 .. 256 work-groups), a small-batch launch
    (the other protocol, same workspace) and a 65536-sample launch in between, one parameter kind written at a time."""
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(21)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, random_state=42).cuda().eval()
    gen = torch.Generator().manual_seed(22)
    sizes = (7700, 9001, 300, 65536, 8000)
    xs = {B: torch.randn(B, 784, generator=gen) for B in sizes}
    xd = {B: v.cuda() for B, v in xs.items()}
    rows = {B: torch.randint(0, B, (96,), generator=gen) for B in sizes}
    with torch.no_grad():
        for B in sizes:
            model(xd[B])
        for step in range(3):
            if what == 'loc':
                model.base_layer.loc.data.add_(0.03 * (step + 1))
            elif what == 'sum':
                for layer in model.layers:
                    if hasattr(layer, 'weight'):
                        layer.weight.data.add_(torch.randn_like(layer.weight))
            elif what == 'root':
                model.root_layer.weight.data.add_(torch.randn_like(model.root_layer.weight))
            else:
                model.base_layer.scale.data.mul_(1.0 + 0.1 * torch.rand_like(model.base_layer.scale))
            sd = _state(model)
            order = sizes if step % 2 == 0 else sizes[::-1]
            for B in order + order[:2]:
                got = model(xd[B])[rows[B].cuda()].cpu().numpy()
                assert rel_err(got, orc.ratspn_forward(sd, xs[B][rows[B]]).numpy()) <= TOL, (step, B)


def test_in_launch_table_check_more_work_groups_than_the_chip_holds():
    """16384 samples = 512 model work-groups behind the table work-groups, two rounds of the chip: a work-group only ever
    waits for table work-groups, which are dispatched first."""
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(15)
    model = GaussianRatSpn(784, rg_depth=2, rg_repetitions=8, rg_batch=2, rg_sum=2, random_state=42).cuda().eval()
    x = torch.randn(16384, 784, generator=torch.Generator().manual_seed(16))
    xd = x.cuda()
    rows = torch.randint(0, 16384, (200,), generator=torch.Generator().manual_seed(17))
    with torch.no_grad():
        model(xd)
        for _ in range(2):
            model.base_layer.loc.data.add_(0.01)
            want = orc.ratspn_forward(_state(model), x[rows]).numpy()
            for _ in range(3):
                assert rel_err(model(xd)[rows.cuda()].cpu().numpy(), want) <= TOL


@pytest.mark.parametrize('kw,D', [(dict(rg_batch=2, rg_sum=2, rg_repetitions=8), 784),
                                  (dict(rg_batch=4, rg_sum=4, rg_repetitions=5, out_classes=3), 200)], ids=['i2s2', 'i4s4c3'])
@pytest.mark.parametrize('self_checking', [False, True], ids=['checklaunch', 'selfcheck'])
def test_in_launch_table_check_ring_kernel(kw, D, self_checking):
    """The persistent ring kernel (batches above 16384): by default the stand-alone check launch in front of it; with
    DPK_RING_VI=1 (read once per process: the second case runs in a child process) the variant that carries the check --
    the compute waves of its first work-groups fingerprint in the prologue, a stale launch evaluates on the table-free
    route and rebuilds at its end (measured slower than kernel + check launch, hence opt-in: csrc/ratspn_gemm.hip).
    Writes to each parameter kind in turn, two batch sizes, a scattered subset against the oracle every time."""
    if self_checking:
        import os, subprocess, sys
        env = dict(os.environ, DPK_RING_VI='1')
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', __file__, '-k',
                            'test_in_launch_table_check_ring_kernel and checklaunch and ' + ('i2s2' if D == 784 else 'i4s4c3')],
                           cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    torch.manual_seed(21)
    model = GaussianRatSpn(D, rg_depth=2, random_state=42, **kw).cuda().eval()
    gen = torch.Generator().manual_seed(22)
    xs = {B: torch.randn(B, D, generator=gen) for B in (20000, 33001)}
    xd = {B: v.cuda() for B, v in xs.items()}
    rows = {B: torch.randint(0, B, (160,), generator=gen) for B in xs}
    with torch.no_grad():
        for B in xd:
            model(xd[B])
        for step, what in enumerate(('loc', 'sum', 'root', 'loc')):
            if what == 'loc':
                model.base_layer.loc.data.add_(0.02 * (step + 1))
            elif what == 'sum':
                for layer in model.layers:
                    if hasattr(layer, 'weight'):
                        layer.weight.data.add_(torch.randn_like(layer.weight))
            else:
                model.root_layer.weight.data.add_(torch.randn_like(model.root_layer.weight))
            sd = _state(model)
            for B in ((20000, 33001, 20000) if step % 2 == 0 else (33001, 20000)):
                got = model(xd[B])[rows[B].cuda()].cpu().numpy()
                assert rel_err(got, orc.ratspn_forward(sd, xs[B][rows[B]]).numpy()) <= TOL, (step, what, B)
