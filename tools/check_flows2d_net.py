"""Diagnostic: gradients of one conditioner network (training mode) against autograd over oracle/flows2d_oracle.py."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'deeprob-kit_amd'))
from oracle import flows2d_oracle as orc
from deeprob.flows.layers.densenet import DenseNetwork
from deeprob.flows.layers.resnet import ResidualNetwork


def run(kind, cin, mid, cout, nb, B, H, W):
    torch.manual_seed(cin + mid)
    net = (DenseNetwork if kind == 'dense' else ResidualNetwork)(cin, mid, cout, nb)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('weight_g'):
                p.copy_(0.15 + 0.2 * torch.rand_like(p))
            elif not n.endswith('weight_v'):
                p.copy_(0.5 + torch.rand_like(p))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    names = [n for n, _ in net.named_parameters()]
    for n in names:
        sd[n].requires_grad_(True)
    x = torch.randn(B, cin, H, W)
    go = torch.randn(B, cout, H, W)
    xo = x.clone().requires_grad_(True)
    with orc.training():
        zo = (orc.densenet if kind == 'dense' else orc.resnet)(sd, '', xo)
    zo.backward(go)
    net.cuda().train()
    xc = x.cuda().requires_grad_(True)
    z = net(xc)
    z.backward(go.cuda())
    e = lambda a, b: float((a.cpu() - b).abs().max() / max(1.0, float(b.abs().max())))
    print(kind, cin, mid, cout, nb, B, H, W, 'z %.1e dx %.1e' % (e(z.detach(), zo.detach()), e(xc.grad, xo.grad)))
    for n, p in net.named_parameters():
        v = e(p.grad, sd[n].grad)
        if v > 1e-4:
            print('   ', n, '%.1e' % v, tuple(p.shape))


run('res', 3, 8, 6, 2, 5, 8, 8)
run('dense', 3, 8, 6, 1, 5, 8, 8)
run('dense', 3, 8, 6, 2, 5, 8, 8)
run('dense', 6, 16, 12, 2, 5, 4, 4)
run('dense', 2, 5, 4, 1, 70, 4, 6)
run('dense', 12, 32, 24, 2, 5, 2, 2)
run('dense', 6, 16, 12, 2, 5, 4, 4)
run('res', 12, 32, 24, 2, 5, 2, 2)
run('dense', 12, 32, 24, 1, 5, 4, 4)
run('dense', 12, 32, 24, 1, 64, 2, 2)
