"""Error behaviour of the C ABI called directly (include/deeprob_hip.h): negative DPK_E* codes and a message in
dpk_last_error(), never a crash, never a silent success."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DPK_EINVAL, DPK_EWORKSPACE, DPK_EUNSUPPORTED = -1, -2, -4


def _model(**kw):
    from deeprob.spn.models import GaussianRatSpn
    base = dict(in_features=64, rg_depth=2, rg_repetitions=4, rg_batch=2, rg_sum=2, random_state=1)
    base.update(kw)
    return GaussianRatSpn(**base).cuda().eval()


def _fused_args(m, x, out, ws, ws_bytes=None):
    from deeprob.hip import ptr
    base = m.base_layer
    sums = [l.weight for l in m.layers if hasattr(l, 'weight')]
    R, I, d = base.mask.shape[0], base.out_channels, base.mask.shape[1]
    return [ptr(x), x.shape[0], x.shape[1], ptr(base.mask), None, ptr(base.loc), ptr(base.scale),
            ptr(sums[0]) if sums else None, ptr(sums[1]) if len(sums) > 1 else None, ptr(m.root_layer.weight),
            m.rg_depth, R // 2 ** m.rg_depth, I, m.rg_sum, m.out_classes, ptr(out), None, None, ptr(ws),
            ws.numel() if ws_bytes is None else ws_bytes, 0, torch.cuda.current_stream().cuda_stream]


def test_fused_forward_error_codes():
    from deeprob.hip import load_library
    lib = load_library()
    m = _model()
    x = torch.randn(10, 64, device='cuda')
    out = torch.empty(10, 1, device='cuda')
    R, I, d = m.base_layer.mask.shape[0], 2, m.base_layer.mask.shape[1]
    n = lib.dpk_ratspn_workspace_bytes(64, R, d, I, 2, 4, 2, 1)
    assert n > 0
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')
    assert lib.dpk_ratspn_forward(*_fused_args(m, x, out, ws)) == 0
    # workspace too small
    assert lib.dpk_ratspn_forward(*_fused_args(m, x, out, ws, ws_bytes=1024)) == DPK_EWORKSPACE
    assert b'workspace' in lib.dpk_last_error()
    # null input
    args = _fused_args(m, x, out, ws)
    args[0] = None
    assert lib.dpk_ratspn_forward(*args) == DPK_EINVAL and lib.dpk_last_error()
    # negative batch
    args = _fused_args(m, x, out, ws)
    args[1] = -3
    assert lib.dpk_ratspn_forward(*args) == DPK_EINVAL
    # a shape outside the fused envelope is reported, not mis-evaluated
    m16 = _model(rg_batch=16, rg_sum=16)
    R16, d16 = m16.base_layer.mask.shape[0], m16.base_layer.mask.shape[1]
    n16 = lib.dpk_ratspn_workspace_bytes(64, R16, d16, 16, 2, 4, 16, 1)
    ws16 = torch.empty(max(n16, 256), dtype=torch.uint8, device='cuda')
    assert lib.dpk_ratspn_forward(*_fused_args(m16, x, out, ws16)) == DPK_EUNSUPPORTED
    assert lib.dpk_ratspn_workspace_bytes(0, R, d, I, 2, 4, 2, 1) < 0


def test_layer_entry_error_codes():
    from deeprob.hip import load_library, ptr
    lib = load_library()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(8, 4, 6, device='cuda')           # [B, R, N]
    out = torch.empty(8, 2, 36, device='cuda')
    assert lib.dpk_product_forward(ptr(x), 8, 4, 6, ptr(out), st) == 0
    assert lib.dpk_product_forward(None, 8, 4, 6, ptr(out), st) == DPK_EINVAL
    assert lib.dpk_product_forward(ptr(x), 8, 3, 6, ptr(out), st) == DPK_EINVAL      # odd number of regions
    w = torch.randn(2, 5, 36, device='cuda')
    o2 = torch.empty(8, 2, 5, device='cuda')
    n = lib.dpk_sum_workspace_bytes(8, 2, 36, 5)
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')
    assert lib.dpk_sum_forward(ptr(out), ptr(w), 8, 2, 36, 5, ptr(o2), ptr(ws), n, st) == 0
    assert lib.dpk_sum_forward(ptr(out), ptr(w), 8, 2, 36, 5, ptr(o2), ptr(ws), 16, st) == DPK_EWORKSPACE
    assert lib.dpk_sum_forward(ptr(out), None, 8, 2, 36, 5, ptr(o2), ptr(ws), n, st) == DPK_EINVAL
    # coupling: hidden width the fused kernel is not built for -> DPK_EUNSUPPORTED (the host then takes the MLP route)
    D, U = 10, 20
    xx = torch.randn(4, D, device='cuda')
    mask = (torch.arange(D) % 2).float().cuda()
    W1, b1 = torch.randn(U, D, device='cuda'), torch.randn(U, device='cuda')
    W2, b2 = torch.randn(2 * D, U, device='cuda'), torch.randn(2 * D, device='cuda')
    act = torch.ones(1, device='cuda')
    o3, ldj = torch.empty_like(xx), torch.empty(4, device='cuda')
    nws = max(lib.dpk_coupling1d_workspace_bytes(D, 32, 5, 5), 1 << 16)
    wsc = torch.empty(nws, dtype=torch.uint8, device='cuda')
    rc = lib.dpk_coupling1d_forward(ptr(xx), 4, D, ptr(mask), ptr(1 - mask), 5, 5, ptr(W1), ptr(b1), ptr(W2), ptr(b2), U,
                                    ptr(act), None, None, 1, 0, ptr(o3), ptr(ldj), 0, ptr(wsc), nws, st)
    assert rc == DPK_EUNSUPPORTED and b'units' in lib.dpk_last_error()
    torch.cuda.synchronize()


def test_round2_entry_error_codes():
    """The entry points added in round 2, called directly: envelope, workspace and null-pointer errors."""
    import ctypes
    from deeprob.hip import load_library, ptr
    lib = load_library()
    st = torch.cuda.current_stream().cuda_stream
    # --- alternating-mask coupling on the matrix cores ---------------------------------------------------------
    D, U, B = 16, 32, 5
    x = torch.randn(B, D, device='cuda')
    W1, b1 = torch.randn(U, D, device='cuda'), torch.randn(U, device='cuda')
    W2, b2 = torch.randn(2 * D, U, device='cuda'), torch.randn(2 * D, device='cuda')
    act = torch.ones(1, device='cuda')
    out, ldj = torch.empty_like(x), torch.empty(B, device='cuda')
    n = lib.dpk_coupling1d_pairs_workspace_bytes(D, U)
    assert n > 0
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')

    def pairs(d=D, u=U, parity=0, w1=W1, nbytes=n):
        return lib.dpk_coupling1d_pairs_forward(ptr(x), B, d, parity, ptr(w1), ptr(b1), ptr(W2), ptr(b2), u, ptr(act),
                                                None, None, 1, 0, ptr(out), ptr(ldj), 0, ptr(ws), nbytes, 0, st)
    assert pairs() == 0
    assert pairs(u=20) == DPK_EUNSUPPORTED                       # hidden width not a multiple of 32
    assert pairs(d=12) == DPK_EUNSUPPORTED                       # D % 8 != 0
    assert lib.dpk_coupling1d_pairs_workspace_bytes(12, U) == DPK_EUNSUPPORTED
    assert pairs(parity=2) == DPK_EINVAL
    assert pairs(w1=None) == DPK_EINVAL
    assert pairs(nbytes=64) == DPK_EWORKSPACE
    # --- the same tables through the batched entry (round 3): built by it, then used with DPK_FLAG_PARAMS_CACHED ----------
    import ctypes
    from deeprob.hip import DPK_FLAG_PARAMS_CACHED, DPK_FLAG_PARAMS_VERIFY
    from deeprob.hip.ops_flows import _PairsTablesArgs, _BnFoldArgs
    want = out.clone()
    ws2 = torch.zeros(n, dtype=torch.uint8, device='cuda')

    def tables(k=1, flags=0, d=D, nbytes=n, w1=W1):
        arr = (_PairsTablesArgs * max(k, 1))(*[_PairsTablesArgs(ptr(w1), ptr(b1), ptr(W2), ptr(b2), None, None, ptr(ws2), nbytes,
                                                                d, U, 0, 1, flags) for _ in range(max(k, 1))])
        return lib.dpk_coupling1d_pairs_tables(k, ctypes.cast(arr, ctypes.c_void_p), st)
    assert tables() == 0
    out2 = torch.empty_like(x)
    assert lib.dpk_coupling1d_pairs_forward(ptr(x), B, D, 0, ptr(W1), ptr(b1), ptr(W2), ptr(b2), U, ptr(act), None, None, 1, 0,
                                            ptr(out2), ptr(ldj), 0, ptr(ws2), n, DPK_FLAG_PARAMS_CACHED, st) == 0
    assert torch.equal(out2, want)
    assert tables(flags=DPK_FLAG_PARAMS_VERIFY) == 0              # unchanged parameters: the gate stays closed
    assert tables(k=0) == 0 and tables(k=17) == DPK_EINVAL
    assert tables(d=12) == DPK_EUNSUPPORTED and tables(w1=None) == DPK_EINVAL and tables(nbytes=64) == DPK_EWORKSPACE
    # --- batched BatchNormLayer1d fold: bit-identical to the per-layer entry, plus the sum of the constants ---------------
    Dn, nl = 50, 6
    prm = [[torch.randn(Dn, device='cuda') * 0.3, torch.randn(Dn, device='cuda'), torch.rand(Dn, device='cuda') + 0.1,
            torch.randn(Dn, device='cuda')] for _ in range(nl)]
    one = [[torch.empty(Dn, device='cuda'), torch.empty(Dn, device='cuda'), torch.empty(1, device='cuda')] for _ in range(nl)]
    many = [[torch.empty(Dn, device='cuda'), torch.empty(Dn, device='cuda'), torch.empty(1, device='cuda')] for _ in range(nl)]
    for (w, b, v, m), (sc, sh, c) in zip(prm, one):
        assert lib.dpk_bn1d_fold(ptr(w), ptr(b), ptr(v), ptr(m), 1e-5, Dn, 0, None, None, ptr(sc), ptr(sh), ptr(c), 0, st) == 0
    total = torch.empty(1, device='cuda')
    arr = (_BnFoldArgs * nl)(*[_BnFoldArgs(ptr(w), ptr(b), ptr(v), ptr(m), None, None, ptr(sc), ptr(sh), ptr(c), 1e-5, Dn, 0, 0)
                               for (w, b, v, m), (sc, sh, c) in zip(prm, many)])
    assert lib.dpk_bn1d_fold_many(nl, ctypes.cast(arr, ctypes.c_void_p), ptr(total), st) == 0
    for a3, b3 in zip(one, many):
        assert all(torch.equal(p, q) for p, q in zip(a3, b3))
    assert abs(total.item() - sum(c.item() for _, _, c in one)) <= 1e-4 * max(1.0, abs(total.item()))
    assert lib.dpk_bn1d_fold_many(17, ctypes.cast(arr, ctypes.c_void_p), None, st) == DPK_EINVAL
    assert lib.dpk_bn1d_fold_many(nl, None, None, st) == DPK_EINVAL
    # --- softmaxed-weight tables of several DGC-SPN levels in one launch: the same bytes as the per-level builds --------------
    from deeprob.hip.ops_spatial import _SpatialTablesArgs
    Ct, HWs = 8, (49, 81)
    wts = [torch.randn(8, Ct, hw, device='cuda') for hw in HWs]
    rootw = torch.randn(3, 8 * 25, device='cuda')
    segs = [((8 * Ct * hw * 4 + 255) // 256) * 256 for hw in HWs]
    wsl = [torch.zeros(2 * sg + (3 * 200 * 4 if i == 1 else 0), dtype=torch.uint8, device='cuda') for i, sg in enumerate(segs)]

    def sp_tables(k=2, nbytes=None, w0=wts[0]):
        ents = [_SpatialTablesArgs(ptr(w0 if i == 0 else wts[i]), ptr(wsl[i]), wsl[i].numel() if nbytes is None else nbytes,
                                   ptr(rootw) if i == 1 else None, Ct, 8, HWs[i], 3 if i == 1 else 0, 200 if i == 1 else 0)
                for i in range(2)]
        arr = (_SpatialTablesArgs * 2)(*ents)
        return lib.dpk_spatial_tables(k, ctypes.cast(arr, ctypes.c_void_p), st)
    assert sp_tables() == 0
    for i, hw in enumerate(HWs):
        tab = wsl[i][:8 * Ct * hw * 4].view(torch.float32).view(8, Ct, hw)
        ltab = wsl[i][segs[i]:segs[i] + 8 * Ct * hw * 4].view(torch.float32).view(8, Ct, hw)
        assert torch.allclose(tab, torch.softmax(wts[i], dim=1), atol=1e-6)
        assert torch.allclose(ltab, torch.log_softmax(wts[i], dim=1), atol=2e-6)
    lr = wsl[1][2 * segs[1]:2 * segs[1] + 3 * 200 * 4].view(torch.float32).view(3, 200)
    assert torch.allclose(lr, torch.log_softmax(rootw, dim=1), atol=2e-6)
    assert sp_tables(k=0) == 0 and sp_tables(k=9) == DPK_EINVAL and sp_tables(w0=None) == DPK_EINVAL
    assert sp_tables(nbytes=64) == DPK_EWORKSPACE
    # --- fused product + sum level of a DGC-SPN, forward and backward --------------------------------------------
    C, H, W, OH, OW = 8, 6, 6, 7, 7                               # 2x2 taps, 'full' padding 1, dilation 1
    xin = torch.randn(4, C, H, W, device='cuda')
    wsum = torch.randn(8, C, OH, OW, device='cuda')
    y = torch.empty(4, 8, OH, OW, device='cuda')
    nws = lib.dpk_spatial_sum_workspace_bytes(C, 8, OH, OW)
    wss = torch.empty(nws, dtype=torch.uint8, device='cuda')
    geo = (C, H, W, OH, OW, 2, 2, 1, 1, 1, 1, 1, 1)

    def fwd(g=geo, nbytes=nws, weight=wsum):
        return lib.dpk_spatial_prodsum_forward(ptr(xin), 4, *g, ptr(weight), 8, ptr(y), ptr(wss), nbytes, 0, st)
    assert fwd() == 0
    assert fwd(nbytes=32) == DPK_EWORKSPACE
    assert fwd(weight=None) == DPK_EINVAL
    assert fwd(g=(C, H, W, OH, OW, 3, 3, 1, 1, 1, 1, 1, 1)) == DPK_EUNSUPPORTED      # 9 taps: not fused
    g = torch.randn_like(y)
    gprod, gx, gw = torch.empty(4, C, OH, OW, device='cuda'), torch.empty_like(xin), torch.empty_like(wsum)

    def bwd(geom=geo, nbytes=nws, gp=gprod):
        return lib.dpk_spatial_prodsum_backward(ptr(xin), 4, *geom, ptr(wsum), 8, ptr(y), ptr(g), ptr(gp), ptr(gx), ptr(gw),
                                                ptr(wss), nbytes, 0, st)
    assert bwd() == 0
    assert bwd(nbytes=32) == DPK_EWORKSPACE
    assert bwd(gp=None) == DPK_EINVAL                                                # grad_in wanted, no scratch
    assert bwd(geom=(C, H, W, OH, OW, 3, 3, 1, 1, 1, 1, 1, 1)) == DPK_EUNSUPPORTED
    # --- batch-sized workspace query of the last-level kernel ----------------------------------------------------
    g5 = (ctypes.c_int32 * 10)(OH, OW, 2, 2, 1, 1, 1, 1, 1, 1)
    g6 = (ctypes.c_int32 * 10)(5, 5, 2, 2, 1, 1, 2, 2, 0, 0)
    base = lib.dpk_spatial_sumprodroot_workspace_bytes(C, 8, OH, OW, 5, 5, 1)
    assert base > 0
    assert lib.dpk_spatial_sumprodroot_workspace_bytes_batch(4096, C, H, W, g5, 8, g6, 1) >= base
    assert lib.dpk_spatial_sumprodroot_workspace_bytes_batch(4096, C, H, W, None, 8, g6, 1) == DPK_EINVAL
    assert lib.dpk_spatial_sumprodroot_workspace_bytes_batch(-1, C, H, W, g5, 8, g6, 1) == DPK_EINVAL
    torch.cuda.synchronize()


def test_workspace_forget_and_per_workspace_hint():
    """dpk_workspace_forget hands back what the library keeps per workspace address (fingerprint slots, the
    marginalised-evidence hint word); the hint is per workspace since round 4: a model fed NaN evidence and a model fed
    clean evidence in the same process both stay right, and models can come and go by the hundred."""
    import gc
    from deeprob.hip import load_library
    from deeprob.spn.models import GaussianRatSpn
    from oracle import ratspn_oracle as orc
    lib = load_library()
    assert lib.dpk_workspace_forget(None, 0) == 0
    buf = torch.empty(4096, dtype=torch.uint8, device='cuda')
    assert lib.dpk_workspace_forget(buf.data_ptr(), buf.numel()) == 0          # nothing registered: a no-op
    torch.manual_seed(0)
    a = GaussianRatSpn(64, rg_depth=2, rg_repetitions=4, random_state=1).cuda().eval()
    b = GaussianRatSpn(64, rg_depth=2, rg_repetitions=4, random_state=2).cuda().eval()
    x = torch.randn(500, 64, generator=torch.Generator().manual_seed(3))
    xn = x.clone()
    xn[torch.rand(500, 64, generator=torch.Generator().manual_seed(4)) < 0.3] = float('nan')
    sa = {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}
    sb = {k: v.detach().cpu().clone() for k, v in b.state_dict().items()}
    with torch.no_grad():
        for _ in range(4):          # a meets NaN evidence on every call, b never does
            ga, gb = a(xn.cuda()), b(x.cuda())
    err = lambda got, want: float(((got.cpu() - want).abs() / want.abs().clamp_min(1.0)).max())
    assert err(ga, orc.ratspn_forward(sa, xn)) <= 1e-5 and err(gb, orc.ratspn_forward(sb, x)) <= 1e-5
    for i in range(150):            # workspaces come and go (each model's __del__ returns its slots)
        m = GaussianRatSpn(32, rg_depth=2, rg_repetitions=2, random_state=i).cuda().eval()
        with torch.no_grad():
            m(x[:40, :32].cuda())
            m(x[:40, :32].cuda())
        del m
    gc.collect()
    with torch.no_grad():
        assert err(a(xn.cuda()), orc.ratspn_forward(sa, xn)) <= 1e-5


def test_round4_training_entries_error_codes():
    """dpk_adam_step / dpk_prodsum_backward / dpk_ratspn_forward_train / dpk_neg_mean_*: refusals are codes + messages."""
    import ctypes
    from deeprob.hip import load_library, ptr
    from deeprob.hip.optim import _AdamTensor
    lib = load_library()
    st = torch.cuda.current_stream().cuda_stream
    # Adam: more tensors than one launch carries; null state
    p = torch.zeros(4, device='cuda')
    ent = (_AdamTensor * 97)(*[_AdamTensor(ptr(p), ptr(p), ptr(p), ptr(p), 4) for _ in range(97)])
    step, tick = torch.zeros(1, device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
    assert lib.dpk_adam_step(97, ctypes.cast(ent, ctypes.c_void_p), 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, ptr(step), ptr(tick), st) == DPK_EUNSUPPORTED
    assert lib.dpk_adam_step(1, ctypes.cast(ent, ctypes.c_void_p), 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, ptr(tick), st) == DPK_EINVAL
    assert lib.dpk_adam_step(0, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, ptr(step), ptr(tick), st) == 0
    # level backward: shapes outside the single-launch kernels are refused (the caller chains the layers), small workspace
    B, R, N, S = 5, 4, 32, 16      # (32 nodes per region: beyond the 2 / 4 / 8 / 16-node instantiations)
    x = torch.randn(B, R, N, device='cuda')
    w = torch.randn(R // 2, S, N * N, device='cuda')
    out = torch.randn(B, R // 2, S, device='cuda')
    ws = torch.empty(lib.dpk_sum_workspace_bytes(B, R // 2, N * N, S), dtype=torch.uint8, device='cuda')
    assert lib.dpk_prodsum_backward(ptr(x), ptr(w), ptr(out), ptr(out), B, R, N, S, 0, ptr(x), ptr(w), ptr(ws), ws.numel(), st) == DPK_EUNSUPPORTED
    assert b'not built' in lib.dpk_last_error()
    x8, w8, o8 = torch.randn(B, R, 8, device='cuda'), torch.randn(R // 2, 8, 64, device='cuda'), torch.randn(B, R // 2, 8, device='cuda')
    assert lib.dpk_prodsum_backward(ptr(x8), ptr(w8), ptr(o8), ptr(o8), B, R, 8, 8, 0, None, None, ptr(ws), 16, st) == DPK_EWORKSPACE
    assert lib.dpk_prodsum_backward(ptr(x8), ptr(w8), ptr(o8), ptr(o8), B, 3, 8, 8, 0, None, None, ptr(ws), ws.numel(), st) == DPK_EINVAL
    # training forward: a depth-3 model is outside the single-launch route
    m = _model(rg_depth=3)
    base = m.base_layer
    Rm, I, d = base.mask.shape[0], base.out_channels, base.mask.shape[1]
    xin = torch.randn(6, 64, device='cuda')
    n = lib.dpk_ratspn_workspace_bytes(64, Rm, d, I, 3, 4, 2, 1)
    wsm = torch.empty(n, dtype=torch.uint8, device='cuda')
    t = [torch.empty(6, Rm, I, device='cuda'), torch.empty(6, Rm // 2, 2, device='cuda'), torch.empty(6, 1, device='cuda'), torch.empty(6, 1, device='cuda')]
    sums = [l.weight for l in m.layers if hasattr(l, 'weight')]
    rc = lib.dpk_ratspn_forward_train(ptr(xin), 6, 64, ptr(base.mask), None, ptr(base.loc), ptr(base.scale), ptr(sums[0]),
                                      ptr(m.root_layer.weight), 3, 4, I, 2, 1, ptr(t[3]), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(wsm),
                                      wsm.numel(), 2, st)   # (DPK_FLAG_UNIT_SCALE)
    assert rc == DPK_EUNSUPPORTED and lib.dpk_last_error()
    # loss
    assert lib.dpk_neg_mean_forward(ptr(xin), 0, ptr(step), st) == DPK_EINVAL
    assert lib.dpk_neg_mean_backward(None, 4, ptr(p), st) == DPK_EINVAL
    # one launch for the tables of a sum layer and a root layer: shapes outside the matrix-core route are refused
    w5 = torch.randn(2, 3, 25, device='cuda')
    wr = torch.randn(1, 2 * 9, device='cuda')
    wsb = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
    assert lib.dpk_upper_tables_pair(ptr(w5), 4, 5, 3, ptr(wsb), wsb.numel(), ptr(wr), 4, 3, 1, ptr(wsb), wsb.numel(), st) == DPK_EUNSUPPORTED
    assert lib.dpk_upper_tables_pair(None, 4, 8, 8, ptr(wsb), wsb.numel(), ptr(wr), 4, 8, 1, ptr(wsb), wsb.numel(), st) == DPK_EINVAL
