"""Developer diagnostic (not collected by pytest): error table of the HIP DGC-SPN path vs the golden vectors,
next to the reference's own fp32 error vs the fp64 oracle.  Usage: python -m tests.diag_dgc"""
import sys
import numpy as np
import torch
from tests.conftest import load_golden
from tests.dgc_cases import CASES, build_dgc, plan_of
from tests.util import grad_err
from oracle import dgcspn_oracle as dorc


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(b)
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin])))) if fin.any() else 0.0


for name in sorted(CASES):
    g = load_golden(name)
    model = build_dgc(name, g)
    sd64 = {k: v.detach().double() if v.is_floating_point() else v for k, v in model.state_dict().items()}
    plan = plan_of(name)
    x, xn = torch.from_numpy(g['x']), torch.from_numpy(g['x_nan'])
    with torch.no_grad():
        ll64 = dorc.dgcspn_forward(sd64, x.double(), plan).numpy()
        lln64 = dorc.dgcspn_forward(sd64, xn.double(), plan).numpy()
    model = model.cuda()
    with torch.no_grad():
        ll = model(x.cuda()).cpu().numpy()
        lln = model(xn.cuda()).cpu().numpy()
    print('{:32s} ll: hip-vs-ref {:.2e} ref-vs-f64 {:.2e} hip-vs-f64 {:.2e} | nan: {:.2e} {:.2e} {:.2e}'.format(
        name, err(ll, g['ll']), err(g['ll'], ll64), err(ll, ll64), err(lln, g['ll_nan']), err(g['ll_nan'], lln64),
        err(lln, lln64)))
    if 'act.leaf' in g.files:
        with torch.no_grad():
            h = model.base_layer(x.cuda())
            row = ['leaf {:.1e}'.format(err(h.cpu().numpy(), g['act.leaf']))]
            for i, layer in enumerate(model.layers):
                h = layer(torch.from_numpy(g['act.layer{}'.format(i - 1)] if i else g['act.leaf']).cuda())
                row.append('{}:{:.1e}'.format(i, err(h.cpu().numpy(), g['act.layer{}'.format(i)])))
        print('    layers (each fed the golden input):', ' '.join(row))
    # gradients: hip vs ref, ref vs fp64
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.is_floating_point()}
    x64 = x.double().requires_grad_(True)
    y = torch.from_numpy(g['y']) if 'y' in g.files else None
    dorc.dgcspn_loss(dorc.dgcspn_forward(leaves, x64, plan), y).backward()
    xg = x.cuda().requires_grad_(True)
    model.loss(model(xg), y.cuda() if y is not None else None).backward()
    row = ['x: {:.1e}/{:.1e}'.format(grad_err(xg.grad.cpu().numpy(), g['grad.x']),
                                     grad_err(g['grad.x'], x64.grad.numpy()))]
    for k, p in model.named_parameters():
        if 'grad.' + k in g.files:
            row.append('{}: {:.1e}/{:.1e}'.format(k.replace('layers.', 'L').replace('.weight', ''),
                                                  grad_err(p.grad.cpu().numpy(), g['grad.' + k]),
                                                  grad_err(g['grad.' + k], leaves[k].grad.numpy())))
    print('    grads hip-vs-ref/ref-vs-f64:', ' '.join(row))
    with torch.enable_grad():
        mpe = model.mpe(xn.cuda()).cpu().numpy()
    mpe64 = dorc.dgcspn_mpe(sd64, xn.double(), plan).numpy()
    print('    mpe max abs: hip-vs-ref {:.2e} ref-vs-f64 {:.2e} hip-vs-f64 {:.2e}'.format(
        np.max(np.abs(mpe - g['mpe'])), np.max(np.abs(g['mpe'] - mpe64)), np.max(np.abs(mpe - mpe64))))
