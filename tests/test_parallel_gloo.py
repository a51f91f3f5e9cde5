"""world_size-2 gloo test of the batch-sharded mean-LL path (runs on CPU: the per-rank evaluator is
the oracle, the sharding / all-reduce logic is the product's)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.parallel import ShardedLogLikelihood, shard_batch, shard_bounds
    from oracle import ratspn_oracle as orc
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load_golden('ratspn_g15_d2_r3_i3_s5_pad')
    sd = orc.state_from_npz(g)

    def local_sum(x):
        ll = orc.ratspn_forward(sd, x)
        return torch.stack([ll.double().sum(), torch.tensor(float(ll.numel()), dtype=torch.float64)])

    ev = ShardedLogLikelihood(group=dist.group.WORLD, local_sum_fn=local_sum)
    gen = torch.Generator().manual_seed(0)
    batches = [torch.randn(n, 15, generator=gen) for n in (33, 64, 7)]  # uneven shards, fewer than world*k
    for x in batches:
        ev.step(shard_batch(x, rank, world))
    means = ev.drain()
    want = [float(orc.ratspn_forward(sd, x).double().mean()) for x in batches]
    ok = all(abs(a - b) <= 1e-9 * abs(b) for a, b in zip(means, want))
    lo, hi = shard_bounds(33, rank, world)
    ok = ok and (hi - lo) in (16, 17) and ev.drain() == []

    # the product's slot-pool path: several steps per collective, contiguous pool runs reduced in place, a run that
    # wraps from one pool to the next
    class PoolEval(ShardedLogLikelihood):
        def _local(self, x, kernel_events=None):
            acc = self._acc_slot(x.device)     # (a slot = sixteen partial sums, then the count: {sum, count} lands in the last two)
            acc[-2:].copy_(local_sum(x))
            return acc

    ev2 = PoolEval(model=None, group=dist.group.WORLD, reduce_every=3)
    ev2._acc_slot(torch.device('cpu'))
    ev2._pool_next = 254                                   # two slots left: the first window straddles two pools
    more = [torch.randn(n, 15, generator=gen) for n in (5, 9, 21, 40, 3, 12, 8)]
    for x in more:
        ev2.step(shard_batch(x, rank, world))
    means2 = ev2.drain()
    want2 = [float(orc.ratspn_forward(sd, x).double().mean()) for x in more]
    ok = ok and len(means2) == 7 and all(abs(a - b) <= 1e-9 * abs(b) for a, b in zip(means2, want2))
    np.save(os.path.join(out_dir, 'r{}.npy'.format(rank)), np.asarray([float(ok)] + means))
    dist.destroy_process_group()


def test_sharded_mean_ll_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, start_method='spawn')
    r0, r1 = np.load(tmp_path / 'r0.npy'), np.load(tmp_path / 'r1.npy')
    assert r0[0] == 1.0 and r1[0] == 1.0
    assert np.array_equal(r0[1:], r1[1:])  # every rank ends with the same mean LL


def test_shard_bounds_cover_batch():
    from deeprob.parallel import shard_bounds
    for n in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class _ToyDensity(torch.nn.Module):
    """A CPU stand-in with the ProbabilisticModel surface the routines use (forward = LL, loss, apply_constraints):
    independent Normals.  The HIP kernels are not in play here -- the test is about the sharding logic."""

    def __init__(self, d):
        super().__init__()
        self.loc = torch.nn.Parameter(torch.zeros(d))
        self.log_scale = torch.nn.Parameter(torch.zeros(d))

    def forward(self, x):
        z = (x - self.loc) * torch.exp(-self.log_scale)
        return (-0.5 * z * z - self.log_scale - 0.9189385332046727).sum(dim=1, keepdim=True)

    def loss(self, out, y=None):
        return -out.mean()

    def apply_constraints(self):
        pass


def _train_worker(rank, world, port, out_dir):
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.torch.routines import train_generative, test_generative
    from deeprob.torch.callbacks import EarlyStopping
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(5)
    train = torch.randn(203, 6, generator=gen) * 1.7 + 0.8      # 203 = 4 batches of 48 + a ragged one of 11
    valid = torch.randn(50, 6, generator=gen) * 1.7 + 0.8
    torch.manual_seed(11)                                        # same shuffling on every rank
    model = _ToyDensity(6)
    loader = torch.utils.data.DataLoader(train, 48, shuffle=True, drop_last=False)
    vloader = torch.utils.data.DataLoader(valid, 25, shuffle=False)
    opt = torch.optim.Adam(model.parameters(), lr=5e-2)
    es = EarlyStopping(model, patience=50, filepath=os.path.join(out_dir, 'ckpt_w{}.pt'.format(world)))
    hist = train_generative(model, loader, vloader, opt, torch.device('cpu'), es, epochs=6, verbose=False)
    mean, err = test_generative(model, vloader, torch.device('cpu'), verbose=False)
    vec = torch.cat([model.loc.detach(), model.log_scale.detach(), torch.tensor(hist['train'] + hist['valid']),
                     torch.tensor([mean, err])]).double().numpy()
    np.save(os.path.join(out_dir, 'train_w{}_r{}.npy'.format(world, rank)), vec)
    if world > 1:
        dist.destroy_process_group()


def test_sharded_training_equals_single_process(tmp_path):
    """train_generative / test_generative with 2 gloo ranks (each on its shard of every batch, gradients in one
    sample-weighted all-reduce) reproduce the single-process run on the unsharded batches: parameters, the loss
    history and the (mean LL, 2 std / sqrt n) test result."""
    _train_worker(0, 1, 0, str(tmp_path))
    mp.start_processes(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    ref = np.load(tmp_path / 'train_w1_r0.npy')
    r0, r1 = np.load(tmp_path / 'train_w2_r0.npy'), np.load(tmp_path / 'train_w2_r1.npy')
    assert np.allclose(r0, r1, rtol=0, atol=1e-12)               # replicas stay in lock step
    assert np.allclose(r0, ref, rtol=1e-5, atol=1e-6)
    assert ref[12] > ref[17]                                     # the training loss went down over the epochs


def _flow_state(seed, d=12, units=16, n_flows=2, dtype=torch.float64):
    """A small RealNVP-1D state_dict (coupling, batch norm, coupling, batch norm) with every parameter live."""
    g = torch.Generator().manual_seed(seed)
    sd = {'in_base_loc': torch.zeros(d, dtype=dtype), 'in_base_scale': torch.ones(d, dtype=dtype)}
    for k in range(n_flows):
        p = 'layers.{}.'.format(2 * k)
        mask = (torch.arange(d) % 2).to(dtype)
        if k % 2:
            mask = 1 - mask
        sd[p + 'mask'], sd[p + 'inv_mask'] = mask, 1 - mask
        sd[p + 'network.0.weight'] = 0.3 * torch.randn(units, d, generator=g, dtype=dtype)
        sd[p + 'network.0.bias'] = 0.1 * torch.randn(units, generator=g, dtype=dtype)
        sd[p + 'network.2.weight'] = 0.3 * torch.randn(2 * d, units, generator=g, dtype=dtype)
        sd[p + 'network.2.bias'] = 0.1 * torch.randn(2 * d, generator=g, dtype=dtype)
        sd[p + 'scale_act.weight'] = torch.tensor([0.5], dtype=dtype)
        q = 'layers.{}.'.format(2 * k + 1)
        sd[q + 'weight'] = 0.2 * torch.randn(1, d, generator=g, dtype=dtype)
        sd[q + 'bias'] = 0.2 * torch.randn(1, d, generator=g, dtype=dtype)
        sd[q + 'running_var'] = torch.ones(1, d, dtype=dtype)
        sd[q + 'running_mean'] = torch.zeros(1, d, dtype=dtype)
    return sd


_TRAINABLE = ('network', 'scale_act', '.weight', '.bias')


def _flow_step(sd, x, sync):
    """One training-mode evaluation: loss = -mean LL of the rows in x, gradients of every trainable entry."""
    from oracle import flows_oracle as forc
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if any(t in k for t in _TRAINABLE) and 'mask' not in k and 'running' not in k}
    state = dict(sd)
    state.update(params)
    running = {}
    if x.shape[0] > 0:
        loss = -forc.flow_log_prob(state, x, train=True, running=running, sync=sync).mean()
    else:   # an empty shard still joins the collectives of every batch-norm layer
        loss = forc.flow_log_prob(state, x, train=True, running=running, sync=sync).sum()
    loss.backward()
    return params, running, float(loss.detach())


def _syncbn_worker(rank, world, port, out_dir):
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    sd = _flow_state(3)
    x = torch.randn(37, 12, generator=torch.Generator().manual_seed(8), dtype=torch.float64) * 1.3 + 0.4
    xs = parallel.shard_batch(x, rank, world)                       # 19 + 18 rows
    sync = (lambda m: parallel.bn_gather_moments(m, dist.group.WORLD),
            lambda t, n, nt: parallel.bn_reduce_sums(t, n, nt, dist.group.WORLD))
    params, running, _ = _flow_step(sd, xs, sync)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.ParameterList([torch.nn.Parameter(v.detach().clone()) for v in params.values()])

    holder = Holder()
    for hp, v in zip(holder.p, params.values()):
        hp.grad = v.grad.clone()
    parallel.allreduce_gradients(holder, weight=xs.shape[0])        # the product's sample-weighted gradient exchange
    vec = torch.cat([hp.grad.reshape(-1).double() for hp in holder.p] +
                    [running[k].reshape(-1).double() for k in sorted(running)])
    np.save(os.path.join(out_dir, 'sbn_r{}.npy'.format(rank)), vec.numpy())
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_single_process(tmp_path):
    """Train-mode BatchNormLayer1d under batch sharding (SURVEY 8e caveat): with the product's moment / gradient-sum
    exchange (deeprob.parallel.bn_gather_moments, bn_reduce_sums) and the sample-weighted gradient all-reduce, two
    ranks on 19 + 18 rows reproduce the single-process gradients and running statistics of the 37-row batch."""
    mp.start_processes(_syncbn_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    sd = _flow_state(3)
    x = torch.randn(37, 12, generator=torch.Generator().manual_seed(8), dtype=torch.float64) * 1.3 + 0.4
    params, running, _ = _flow_step(sd, x, None)
    ref = torch.cat([v.grad.reshape(-1) for v in params.values()] +
                    [running[k].reshape(-1) for k in sorted(running)]).numpy()
    r0, r1 = np.load(tmp_path / 'sbn_r0.npy'), np.load(tmp_path / 'sbn_r1.npy')
    assert np.allclose(r0, r1, rtol=0, atol=1e-13)
    # (the gradient bucket of allreduce_gradients is fp32; the running statistics, which do not pass through it, are exact)
    assert np.max(np.abs(r0 - ref)) <= 5e-7 * max(1.0, np.max(np.abs(ref)))
    n_run = 4 * 12
    assert np.max(np.abs(r0[-n_run:] - ref[-n_run:])) <= 1e-12
    assert np.max(np.abs(ref)) > 1e-3   # (the comparison is not of zeros)


def _short_batch_worker(rank, world, port, out_dir):
    """train_generative over a loader whose last batch has ONE row, model with a synchronised batch statistic."""
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.flows.utils import BatchNormLayer1d
    from deeprob.torch.routines import train_generative
    from deeprob.torch.callbacks import EarlyStopping
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)

    class CpuCentering(BatchNormLayer1d):
        """BatchNormLayer1d's role on the CPU: centres by the (detached) mean of the whole batch; with a sync_group
        the mean comes from an all-reduce that EVERY rank must join -- the collective the routine has to keep in
        step (the HIP layer exchanges its moments the same way, deeprob.parallel.bn_gather_moments)."""

        def apply_backward(self, x):
            if not self.training:      # (the real layer uses its running statistics here: nothing batch dependent)
                return x + self.bias, torch.zeros(x.shape[0])
            stat = torch.cat([x.detach().sum(0), torch.tensor([float(x.shape[0])], dtype=x.dtype)])
            if self.sync_group is not None:
                dist.all_reduce(stat, group=self.sync_group)
            return x - stat[:-1] / stat[-1] + self.bias, torch.zeros(x.shape[0])

    class Toy(_ToyDensity):
        def __init__(self, d):
            super().__init__(d)
            self.norm = CpuCentering(d)

        def forward(self, x):
            return super().forward(self.norm.apply_backward(x)[0])

    gen = torch.Generator().manual_seed(9)
    train = torch.randn(97, 5, generator=gen) * 1.3 + 0.4      # 97 = 3 batches of 32 + ONE row
    valid = torch.randn(20, 5, generator=gen)
    torch.manual_seed(3)
    model = Toy(5)
    loader = torch.utils.data.DataLoader(train, 32, shuffle=True, drop_last=False)
    vloader = torch.utils.data.DataLoader(valid, 20, shuffle=False)
    opt = torch.optim.SGD(model.parameters(), lr=5e-2)
    es = EarlyStopping(model, patience=50, filepath=os.path.join(out_dir, 'short_w{}.pt'.format(world)))
    hist = train_generative(model, loader, vloader, opt, torch.device('cpu'), es, epochs=3, verbose=False)
    vec = torch.cat([p.detach().reshape(-1) for p in model.parameters()] +
                    [torch.tensor(hist['train'] + hist['valid'])]).double().numpy()
    np.save(os.path.join(out_dir, 'short_w{}_r{}.npy'.format(world, rank)), vec)
    if world > 1:
        dist.destroy_process_group()


def test_one_row_last_batch_with_synchronised_statistics(tmp_path):
    """A last batch smaller than the world (round-2 advisor finding: the rank with the empty shard skipped the model
    and its peers hung in the batch-norm all-reduce).  Such a batch is replicated on every rank instead; the run
    finishes and equals the single-process run."""
    _short_batch_worker(0, 1, 0, str(tmp_path))
    mp.start_processes(_short_batch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    ref = np.load(tmp_path / 'short_w1_r0.npy')
    r0, r1 = np.load(tmp_path / 'short_w2_r0.npy'), np.load(tmp_path / 'short_w2_r1.npy')
    assert np.allclose(r0, r1, rtol=0, atol=1e-12)
    assert np.allclose(r0, ref, rtol=1e-5, atol=1e-6)
