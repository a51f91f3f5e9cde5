"""The batch-sharded path on the device: ``ShardedLogLikelihood(model)`` for the three model families (RAT-SPN,
DGC-SPN, RealNVP-1D) in one process and with two gloo ranks sharing ``cuda:0`` (the collective logic is the one that
runs over RCCL on a multi-GPU node; only the transport differs), against the oracles' mean log-likelihoods; and one
sharded training step of a ``RealNVP1d(batch_norm=True)`` (whole-batch BatchNorm statistics, sample-weighted gradient
all-reduce) against the single-process step on the unsharded batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

FAMILIES = ('ratspn', 'ratspn_wide', 'dgcspn', 'realnvp1d', 'realnvp2d')
BATCHES = (257, 64, 1000, 33)


def _init(rank, world, port):
    """Process group + device of a worker: two gloo ranks sharing cuda:0 by default; DPK_TEST_BACKEND=nccl (set by
    tests/test_parallel_nccl_gpu.py on a box with >= 2 devices) gives every rank its own device over RCCL."""
    backend = os.environ.get('DPK_TEST_BACKEND', 'gloo')
    dev = rank if backend == 'nccl' else 0
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _family(kind):
    """(model on the CPU in eval mode, input shape, oracle LL function)."""
    from oracle import ratspn_oracle as orc, dgcspn_oracle as dorc, flows_oracle as forc
    if kind in ('ratspn', 'ratspn_wide'):
        from tests.test_ratspn_gpu import MODELS, SEEDS
        from deeprob.spn.models import GaussianRatSpn
        from tests.util import state_to_model
        name = 'ratspn_g784_d2_r8_i2_s2' if kind == 'ratspn' else 'ratspn_g784_d2_r8_i16_s16'
        g = load_golden(name)
        model = state_to_model(GaussianRatSpn(random_state=SEEDS.get(name, 42), **MODELS[name]), g).eval()
        sd = orc.state_from_npz(g)
        return model, (784,), lambda x: orc.ratspn_forward(sd, x)
    if kind == 'dgcspn':
        from tests.dgc_cases import build_dgc, plan_of
        name = 'dgcspn_3x8x8_dw'
        model = build_dgc(name, load_golden(name))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        plan = plan_of(name)
        return model, (3, 8, 8), lambda x: dorc.dgcspn_forward(sd, x, plan).detach()
    if kind == 'realnvp2d':
        from oracle import flows2d_oracle as f2orc
        from tests.util import flow2d_model
        model = flow2d_model((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=16), 31)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return model, (3, 8, 8), lambda x: f2orc.log_prob(sd, x).detach()
    from tests.flow_cases import build_flow
    name = 'realnvp1d_15'
    model = build_flow(name, load_golden(name))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, (15,), lambda x: forc.flow_log_prob(sd, x).detach()


def _inputs(kind, shape):
    gen = torch.Generator().manual_seed(sum(map(ord, kind)) + 17)
    xs = [torch.randn(n, *shape, generator=gen) for n in BATCHES]
    if kind.startswith('ratspn'):
        xs[0][3, :40] = float('nan')          # marginalised evidence goes through the same path
    return xs


def _eval_worker(rank, world, port, out_dir):
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.parallel import ShardedLogLikelihood, shard_batch
    _init(rank, world, port)
    out = {}
    for kind in FAMILIES:
        model, shape, _ = _family(kind)
        model.cuda()
        ev = ShardedLogLikelihood(model, group=dist.group.WORLD if world > 1 else None, reduce_every=3)
        with torch.no_grad():
            for x in _inputs(kind, shape):
                ev.step(shard_batch(x, rank, world).cuda())
            out[kind] = ev.drain()
            # a resident evaluation ring (bound plans for the RAT-SPN) gives the same means
            ev2 = ShardedLogLikelihood(model, group=dist.group.WORLD if world > 1 else None, static_inputs=True)
            ring = [shard_batch(x, rank, world).cuda() for x in _inputs(kind, shape)]
            for _ in range(2):
                for x in ring:
                    ev2.step(x)
            again = ev2.drain()
        assert len(again) == 2 * len(BATCHES)
        assert again[:len(BATCHES)] == again[len(BATCHES):]
        assert np.allclose(again[:len(BATCHES)], out[kind], rtol=1e-12, atol=0)
    np.save(os.path.join(out_dir, 'ev_w{}_r{}.npy'.format(world, rank)),
            np.asarray([out[k] for k in FAMILIES], dtype=np.float64))
    if world > 1:
        dist.destroy_process_group()


def _want():
    rows = []
    for kind in FAMILIES:
        _, shape, ll = _family(kind)
        with torch.no_grad():
            rows.append([float(ll(x).double().mean()) for x in _inputs(kind, shape)])
    return np.asarray(rows)


def test_sharded_mean_ll_on_device_one_process(tmp_path):
    _eval_worker(0, 1, 0, str(tmp_path))
    got, want = np.load(tmp_path / 'ev_w1_r0.npy'), _want()
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-5


def test_sharded_mean_ll_on_device_two_ranks(tmp_path):
    mp.start_processes(_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    r0, r1 = np.load(tmp_path / 'ev_w2_r0.npy'), np.load(tmp_path / 'ev_w2_r1.npy')
    assert np.array_equal(r0, r1)                      # every rank ends with the same means
    want = _want()
    assert np.max(np.abs(r0 - want) / np.maximum(1.0, np.abs(want))) <= 1e-5


def _train_worker(rank, world, port, out_dir):
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.flows.models import RealNVP1d
    from deeprob.parallel import shard_batch, allreduce_gradients, broadcast_model, synchronize_batchnorm
    from tests.util import randomise_flow
    _init(rank, world, port)
    torch.manual_seed(100 + rank)                      # replicas start DIFFERENT: broadcast_model must fix that
    model = RealNVP1d(20, n_flows=3, units=32, batch_norm=True)
    randomise_flow(model, 5 + rank)
    model.cuda()
    if world > 1:
        broadcast_model(model)
        synchronize_batchnorm(model)
    else:
        torch.manual_seed(100)
        model = RealNVP1d(20, n_flows=3, units=32, batch_norm=True)
        randomise_flow(model, 5)
        model.cuda()
    model.train()
    x = torch.randn(101, 20, generator=torch.Generator().manual_seed(9)) * 1.2 + 0.3
    xs = shard_batch(x, rank, world).cuda()           # 51 + 50 rows
    model.zero_grad()
    loss = model.loss(model(xs))
    loss.backward()
    if world > 1:
        allreduce_gradients(model, weight=xs.shape[0])
    vec = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad] +
                    [b.reshape(-1).float() for n, b in model.named_buffers() if 'running' in n]).double().cpu().numpy()
    np.save(os.path.join(out_dir, 'tr_w{}_r{}.npy'.format(world, rank)), vec)
    if world > 1:
        dist.destroy_process_group()


def test_sharded_training_step_with_batchnorm_equals_single_process(tmp_path):
    """SURVEY 8e caveat closed: train-mode BatchNormLayer1d takes the statistics of the whole sharded batch, so the
    gradients (after the sample-weighted all-reduce) and the running statistics of a 2-rank step equal those of the
    single-process step on the 101-row batch."""
    _train_worker(0, 1, 0, str(tmp_path))
    mp.start_processes(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    ref = np.load(tmp_path / 'tr_w1_r0.npy')
    r0, r1 = np.load(tmp_path / 'tr_w2_r0.npy'), np.load(tmp_path / 'tr_w2_r1.npy')
    scale = np.max(np.abs(ref))
    assert scale > 1e-3
    assert np.max(np.abs(r0 - r1)) <= 1e-6 * scale
    assert np.max(np.abs(r0 - ref)) <= 1e-4 * scale


def _train2d_worker(rank, world, port, out_dir):
    from tests import conftest  # noqa: F401  (sys.path)
    from deeprob.flows.models import RealNVP2d
    from deeprob.parallel import (shard_batch, shard_bounds, allreduce_gradients, broadcast_model, synchronize_batchnorm,
                                  set_shard_sizes)
    _init(rank, world, port)
    torch.manual_seed(200)
    model = RealNVP2d((2, 8, 8), n_flows=1, n_blocks=1, channels=8).cuda()
    if world > 1:
        broadcast_model(model)
        synchronize_batchnorm(model)
    model.train()
    x = torch.randn(37, 2, 8, 8, generator=torch.Generator().manual_seed(10)) * 0.8 + 0.2
    lo, hi = shard_bounds(37, rank, world)
    if world > 1:
        set_shard_sizes(hi - lo, 37)                  # (what routines._batch does for every training batch)
    xs = shard_batch(x, rank, world).cuda()           # 19 + 18 images
    model.zero_grad()
    loss = model.loss(model(xs))
    loss.backward()
    if world > 1:
        allreduce_gradients(model, weight=xs.shape[0])
        set_shard_sizes(None)
    vec = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad and p.grad is not None] +
                    [b.reshape(-1).float() for n, b in model.named_buffers() if 'running' in n]).double().cpu().numpy()
    np.save(os.path.join(out_dir, 'tr2d_w{}_r{}.npy'.format(world, rank)), vec)
    if world > 1:
        dist.destroy_process_group()


def test_sharded_training_step_of_an_image_flow_equals_single_process(tmp_path):
    """Round 4: the 2-D batch norms (BatchNormLayer2d and the nn.BatchNorm2d of the convolutional conditioners) take the
    statistics of the whole sharded batch too (deeprob.parallel.synchronize_batchnorm): gradients after the sample-weighted
    all-reduce and the running statistics of a 2-rank step on 19 + 18 images equal the single-process step on the 37."""
    _train2d_worker(0, 1, 0, str(tmp_path))
    mp.start_processes(_train2d_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, start_method='spawn')
    ref = np.load(tmp_path / 'tr2d_w1_r0.npy')
    r0, r1 = np.load(tmp_path / 'tr2d_w2_r0.npy'), np.load(tmp_path / 'tr2d_w2_r1.npy')
    scale = np.max(np.abs(ref))
    assert scale > 1e-3 and ref.shape == r0.shape
    assert np.max(np.abs(r0 - r1)) <= 1e-6 * scale
    assert np.max(np.abs(r0 - ref)) <= 2e-4 * scale
