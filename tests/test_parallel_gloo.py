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
