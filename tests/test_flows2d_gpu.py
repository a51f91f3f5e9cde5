"""Parity of the HIP RealNVP-2D evaluation path (csrc/flows2d.hip) with reference vectors and with the oracle."""
import numpy as np
import pytest
import torch

from oracle import flows2d_oracle as orc
from tests.util import rel_err, FLOWS2D_CASES, flow2d_model, state_checksum

pytestmark = pytest.mark.gpu
TOL = 1e-5
IDS = [c[0] for c in FLOWS2D_CASES]


@pytest.mark.parametrize('case', FLOWS2D_CASES, ids=IDS)
def test_log_prob_and_latents_golden(golden, case):
    name, feats, kw, seed = case
    g = golden(name)
    model = flow2d_model(feats, kw, seed)
    np.testing.assert_allclose(state_checksum(model), g['sd_check'], rtol=1e-6, atol=1e-6)
    model.cuda()
    x, pre = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['pre']).cuda()
    with torch.no_grad():
        ll = model(x)
        u, ildj = model.apply_backward(pre)
        xr, ldj = model.apply_forward(u)
        b0, d0 = model.layers[0].apply_backward(pre)
        c0, e0 = model.layers[0].in_couplings[0].apply_backward(pre)
        z0 = model.layers[0].in_couplings[0].network(pre, in_mask=model.layers[0].in_couplings[0].mask)
    assert ll.shape == g['ll'].shape and rel_err(ll.cpu().numpy(), g['ll']) <= TOL
    assert rel_err(z0.cpu().numpy(), g['coupling0.z']) <= TOL
    assert rel_err(c0.cpu().numpy(), g['coupling0.u']) <= TOL and rel_err(e0.cpu().numpy(), g['coupling0.ildj']) <= TOL
    assert rel_err(b0.cpu().numpy(), g['block0.u']) <= TOL and rel_err(d0.cpu().numpy(), g['block0.ildj']) <= TOL
    assert rel_err(u.cpu().numpy(), g['u']) <= TOL and rel_err(ildj.cpu().numpy(), g['ildj']) <= TOL
    assert rel_err(xr.cpu().numpy(), g['x_rec']) <= TOL and rel_err(ldj.cpu().numpy(), g['ldj']) <= TOL


@pytest.mark.parametrize('feats,kw,batch', [
    ((1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet'), 37),
    ((3, 32, 32), dict(n_flows=2, n_blocks=1, channels=12, network='resnet'), 9),
    ((2, 4, 6), dict(n_flows=1, n_blocks=1, channels=5, network='densenet'), 70),
    ((1, 28, 28), dict(n_flows=1, n_blocks=1, channels=24, network='densenet', affine=False), 11),
    ((4, 16, 16), dict(n_flows=3, n_blocks=1, channels=6, network='resnet'), 1),
    ((2, 10, 10), dict(n_flows=1, n_blocks=1, channels=19, network='resnet'), 23),   # odd channel counts on the LDS / MFMA kernels, ragged last work-group
    ((3, 32, 32), dict(n_flows=1, n_blocks=1, channels=16, network='resnet'), 3),     # 1024 pixels: 8 tiles per wave; 16x16 after the squeeze
    ((1, 30, 28), dict(n_flows=1, n_blocks=1, channels=16, network='resnet'), 3),     # 840 pixels: one sample per work-group, 27 tiles
])
def test_against_oracle(feats, kw, batch):
    """Seeded inputs at sizes the restatement finishes in seconds: LL, latent, log-det and the sampling direction."""
    model = flow2d_model(feats, kw, 77)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn((batch,) + feats, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want_ll = orc.log_prob(sd, x)
        want_u, want_d = orc.apply_backward(sd, x)
        want_x, want_l = orc.apply_forward(sd, want_u)
        model.cuda()
        ll = model(x.cuda())
        u, d = model.apply_backward(x.cuda())
        xr, l = model.apply_forward(want_u.cuda())
    assert rel_err(ll.cpu().numpy(), want_ll.numpy()) <= TOL
    assert rel_err(u.cpu().numpy(), want_u.numpy()) <= TOL and rel_err(d.cpu().numpy(), want_d.numpy()) <= TOL
    assert rel_err(xr.cpu().numpy(), want_x.numpy()) <= TOL and rel_err(l.cpu().numpy(), want_l.numpy()) <= TOL


def test_invertibility_like_reference():
    """Reference tests/test_flows.py:22-26, :86-93: default initialisation, forward(backward(x)) == x, ildj == -ldj
    (atol 5e-7)."""
    from deeprob.flows.models import RealNVP2d
    torch.manual_seed(42)
    x = torch.rand(64, 3, 8, 8).cuda()
    for kw in [dict(network='resnet', affine=True), dict(network='resnet', affine=False),
               dict(network='densenet', affine=True), dict(network='densenet', affine=False)]:
        flow = RealNVP2d((3, 8, 8), n_flows=2, n_blocks=2, channels=8, **kw).eval().cuda()
        with torch.no_grad():
            u, ildj = flow.apply_backward(x)
            xr, ldj = flow.apply_forward(u)
        assert torch.allclose(xr, x, atol=5e-7) and torch.allclose(ildj, -ldj, atol=5e-7)


def test_squeeze_and_unsqueeze():
    """Reference tests/test_flows.py:36-40 plus the values against the 6-D permutation itself."""
    from deeprob.flows.utils import squeeze_depth2d, unsqueeze_depth2d
    x = torch.rand(5, 3, 8, 12)
    with torch.no_grad():
        s = squeeze_depth2d(x.cuda())
        assert torch.equal(s.cpu(), orc.squeeze(x))
        assert torch.equal(unsqueeze_depth2d(s).cpu(), x)


def test_sampling_runs_on_the_device():
    model = flow2d_model((1, 28, 28), dict(n_flows=1, n_blocks=1, channels=8, logit=0.05), 3).cuda()
    s = model.sample(6)
    assert s.shape == (6, 1, 28, 28) and s.is_cuda and torch.isfinite(s).all()
    with torch.no_grad():
        ll = model(s.clamp(0, 1))
    assert torch.isfinite(ll).all()


def test_sampling_in_training_mode_and_rsample():
    """sample() on a model left in training mode draws with the running statistics (and restores the mode); rsample()
    is declared unsupported the reference's way (NotImplementedError) instead of failing inside a layer."""
    model = flow2d_model((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=8), 4).cuda()
    torch.manual_seed(0)
    want = model.sample(5)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    torch.manual_seed(0)
    got = model.sample(5)
    assert model.training and all(m.training for m in model.layers.modules())
    assert torch.equal(got, want)
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())   # no running statistic moved
    assert model.has_rsample is False
    with pytest.raises(NotImplementedError):
        model.rsample(2)


def test_weight_tables_follow_the_parameters():
    """The packed convolution tables are cached per `_version`: an in-place update must be seen by the next call."""
    model = flow2d_model((3, 8, 8), dict(n_flows=1, n_blocks=1, channels=4), 9).cuda()
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = model(x).clone()
        conv = model.layers[0].in_couplings[0].network.in_conv.conv
        conv.weight_g.mul_(1.5)
        model.layers[0].in_couplings[0].network.out_network[0].running_var.mul_(2.0)
        b = model(x)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        want = orc.log_prob(sd, x.cpu())
    assert not torch.allclose(a, b)
    assert rel_err(b.cpu().numpy(), want.numpy()) <= TOL


def test_conv_channel_slices_and_errors():
    """The convolution entry on channel slices (batch stride above the tensor size), and its error codes."""
    from deeprob.hip import ops_flows2d, load_library, HipError, ptr
    from deeprob.torch.utils import WeightNormConv2d
    import torch.nn.functional as F
    torch.manual_seed(0)
    conv = WeightNormConv2d(5, 18, kernel_size=3, padding=1, bias=True).eval().cuda()
    big = torch.randn(3, 9, 7, 10).cuda()
    outbuf = torch.zeros(3, 40, 7, 10).cuda()
    with torch.no_grad():
        y = ops_flows2d.conv2d(big[:, 2:7], conv, out=outbuf[:, 4:22])
        p = conv.conv
        w = p.weight_v * (p.weight_g / p.weight_v.reshape(18, -1).norm(dim=1).reshape(-1, 1, 1, 1))
        want = F.conv2d(big[:, 2:7].cpu().double(), w.cpu().double(), p.bias.cpu().double(), padding=1).float()
    assert rel_err(outbuf[:, 4:22].cpu().numpy(), want.numpy()) <= TOL and y.data_ptr() == outbuf[:, 4:22].data_ptr()
    assert float(outbuf[:, :4].abs().sum()) == 0.0 and float(outbuf[:, 22:].abs().sum()) == 0.0
    lib = load_library()
    wp = torch.empty(lib.dpk_conv2d_pack_floats(18, 5, 3)).cuda()
    assert lib.dpk_conv2d_forward(ptr(big), 10, 3, 5, 7, 10, ptr(wp), 18, 3, None, None, None, None, 0, ptr(outbuf),
                                  40 * 70, None) == -1          # batch stride below the tensor size
    assert lib.dpk_conv2d_forward(ptr(big), 9 * 70, 3, 5, 7, 10, ptr(wp), 18, 5, None, None, None, None, 0, ptr(outbuf),
                                  40 * 70, None) == -4          # 5x5 kernels are not built
    assert lib.dpk_conv2d_forward(None, 9 * 70, 0, 5, 7, 10, ptr(wp), 18, 3, None, None, None, None, 0, None,
                                  40 * 70, None) == 0           # empty batch
    assert lib.dpk_space_to_depth(ptr(big), 3, 9, 7, 10, None, ptr(outbuf), 36, None, None) == -1   # odd height
    with pytest.raises(NotImplementedError):
        WeightNormConv2d(3, 3, kernel_size=5, padding=2)
    with pytest.raises(HipError), torch.no_grad():
        ops_flows2d.conv2d(big, conv)                            # 9 channels into a 5-channel layer


def test_full_size_properties():
    """MNIST-shaped flow at a serving batch: round trip through both directions, and a slice against the oracle."""
    feats, kw = (1, 28, 28), dict(n_flows=1, n_blocks=2, channels=32, network='resnet', logit=0.05)
    model = flow2d_model(feats, kw, 25)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.rand((1024,) + feats, generator=torch.Generator().manual_seed(8))
    model.cuda()
    with torch.no_grad():
        xc = x.cuda()
        ll = model(xc)
        h, _ = model.preprocess(xc)
        u, ildj = model.apply_backward(h)
        hr, ldj = model.apply_forward(u)
        want = orc.log_prob(sd, x[500:516], logit_alpha=0.05)
    assert torch.isfinite(ll).all()
    assert float((hr - h).abs().max()) <= 2e-5 * float(h.abs().max())
    assert float((ildj + ldj).abs().max()) <= 1e-5 * float(ildj.abs().max())
    assert rel_err(ll[500:516].cpu().numpy(), want.numpy()) <= TOL
    # batch independence: the same rows evaluated alone give the same bits
    with torch.no_grad():
        assert torch.equal(model(xc[500:516]), ll[500:516])
