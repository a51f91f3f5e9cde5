"""Developer diagnostic: all-marginalised input through the HIP layers vs the fp32 CPU oracle, per layer."""
import numpy as np, torch
from tests.conftest import load_golden
from tests.dgc_cases import build_dgc, plan_of
from oracle import dgcspn_oracle as dorc
for name in ['dgcspn_1x28x28_dw', 'dgcspn_3x32x32_pool0_nodw']:
    g = load_golden(name); model = build_dgc(name, g); plan = plan_of(name)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    C = model.base_layer.out_channels
    h = torch.zeros(2, C, *model.in_features[1:])
    hg = h.cuda()
    with torch.no_grad():
        for i, step in enumerate(plan):
            if step[0] == 'prod':
                h = dorc.spatial_product(h, step[1], step[2], step[3], step[4])
            else:
                h = dorc.spatial_sum(h, sd['layers.%d.weight' % i])
            hg = model.layers[i](hg)
            d = (hg.cpu() - h)
            print(name, i, step[0], 'max|d| %.2e mean d %.2e' % (d.abs().max().item(), d.mean().item()))
        r, rg = dorc.spatial_root(h, sd['root_layer.weight']), model.root_layer(hg)
        print('root', r.flatten().tolist(), rg.flatten().tolist(), 'root on oracle input:',
              model.root_layer(h.cuda()).flatten().tolist())
