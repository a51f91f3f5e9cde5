"""Bodies of the HIP-graph + RCCL tests of tests/test_parallel_nccl_gpu.py, run as a child process
(``python -m tests._nccl_graph_cases <case>``; why a child: test_parallel_nccl_gpu.py::_run_case).  A case prints
``CASE-OK <case>`` behind its last assertion, THEN releases its graphs and its process group."""
import os
import sys

import numpy as np
import torch

from tests import test_parallel_gpu as tp


def _init():
    import torch.distributed as dist
    from tests import conftest  # noqa: F401
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(tp._free_port())
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    return dist


def case_window():
    """A window of sharded evaluation steps AND its RCCL all-reduce captured as one HIP graph
    (deeprob.parallel.GraphedEvaluationWindow), replayed: same means as the eager evaluator, replay after replay, and after
    a parameter update through an optimizer-style in-place op (the graph reads live parameters; their tables are checked
    inside the captured launches); then the same window on three parallel chains."""
    dist = _init()
    from deeprob.parallel import ShardedLogLikelihood, GraphedEvaluationWindow
    model, shape, ll = tp._family('ratspn')
    model.cuda()
    xs = [x.cuda() for x in tp._inputs('ratspn', shape)]
    ev = ShardedLogLikelihood(model, group=dist.group.WORLD, static_inputs=True)
    win = GraphedEvaluationWindow(ev, xs, always_reduce=True)
    want = [float(ll(x.cpu()).double().mean()) for x in xs]
    for _ in range(3):
        assert np.allclose(win.replay(), want, rtol=1e-5)
    with torch.no_grad():
        model.base_layer.loc.data.add_(0.05)          # invisible to the host: the captured launches notice
    from oracle import ratspn_oracle as orc
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    want2 = [float(orc.ratspn_forward(sd, x.cpu()).double().mean()) for x in xs]
    assert not np.allclose(want2, want, rtol=1e-7)
    assert np.allclose(win.replay(), want2, rtol=1e-5) and np.allclose(win.replay(), want2, rtol=1e-5)
    # round 6: the same window on three parallel chains -- the captured all-reduce sits behind the chains' join
    win3 = GraphedEvaluationWindow(ev, xs, always_reduce=True, chains=3)
    for _ in range(3):
        assert np.allclose(win3.replay(), want2, rtol=1e-5)
    return dist, (win, win3)


def case_train_step():
    """The sharded optimisation step -- forward, backward, the RCCL gradient all-reduce, the update -- captured as one HIP
    graph (GraphedTrainStep(grad_exchange=...)): on a world of one (collective forced) the replayed steps train like the
    eager loop without the exchange."""
    dist = _init()
    from deeprob.flows.models import RealNVP1d
    from deeprob.hip.graphs import GraphedTrainStep
    from deeprob.parallel import allreduce_gradients
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.randn(64, 24, generator=gen) * 0.7 + 0.5).cuda() for _ in range(9)]
    keep = []

    def run(graphed):
        torch.manual_seed(1)
        flow = RealNVP1d(24, n_flows=2, units=32).cuda().train()
        opt = torch.optim.Adam(flow.parameters(), lr=5e-3, capturable=True, fused=True)
        step = GraphedTrainStep(flow, opt, grad_exchange=(
            lambda n: allreduce_gradients(flow, group=dist.group.WORLD, weight=n, force=True))) if graphed else None
        keep.append(step)
        losses = []
        for x in batches:
            if graphed:
                losses.append(float(step(x).detach()))
            else:
                opt.zero_grad(set_to_none=False)
                loss = flow.loss(flow(x))
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        return losses, (step.graph is not None if graphed else None)

    eager, _ = run(False)
    graphed, captured = run(True)
    assert captured
    assert np.allclose(graphed, eager, rtol=2e-3)
    return dist, keep


if __name__ == '__main__':
    name = sys.argv[1]
    dist, holders = {'window': case_window, 'train_step': case_train_step}[name]()
    torch.cuda.synchronize()
    print('CASE-OK ' + name, flush=True)
    # teardown, graphs first (GraphedEvaluationWindow.close)
    for h in holders:
        if hasattr(h, 'close'):
            h.close()
    del holders
    torch.cuda.synchronize()
    dist.destroy_process_group()
