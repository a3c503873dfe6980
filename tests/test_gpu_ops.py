"""GPU: each HIP op through the C ABI vs a plain torch fp32/fp64 restatement of the same op."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs().max() / b.abs().max().clamp_min(1e-30)
    assert err < tol, float(err)


@pytest.mark.parametrize("M,D", [(3168, 256), (7, 1024), (130, 64)])
def test_layernorm_fwd_bwd(M, D):
    from tavsr import ops
    torch.manual_seed(0)
    xs = torch.randn(M, 2 * D, device="cuda")
    x = xs[:, D:]  # strided rows
    w, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-12)
    xr = x.double().clone().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), wr, br, 1e-12)
    _close(y, yr)
    dy = torch.randn(M, D, device="cuda")
    add = torch.randn(M, D, device="cuda")
    yr.backward(dy.double())
    dx, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, w, dx_add=add)
    _close(dx, xr.grad + add.double(), 1e-4)
    _close(dg, wr.grad, 1e-4)
    _close(db, br.grad, 1e-4)


def test_colsum_and_scale():
    from tavsr import ops
    x = torch.randn(3168, 41, device="cuda")
    _close(ops.colsum(x, scale=0.5), 0.5 * x.double().sum(0), 1e-5)
    s = torch.tensor(0.37, device="cuda")
    _close(ops.scale_dev(x, s, 2.0), x.double() * 0.74, 1e-6)
    _close(ops.axpby(x[:, :1].contiguous().view(-1)[1:2], None, 3.0, 0.0), 3.0 * x[1:2, 0], 1e-6)


@pytest.mark.parametrize("T,lens", [(23, [23, 17, 9]), (99, [99, 50, 1]), (130, [130, 64, 65])])
def test_relpos_softmax_fwd_bwd(T, lens):
    from tavsr import ops
    torch.manual_seed(1)
    H, B, W = 4, len(lens), 2 * T - 1
    ac = torch.randn(H, B, T, T, device="cuda")
    bd = torch.randn(H, B, T, W, device="cuda")
    kl = torch.tensor(lens, device="cuda")
    attn = ops.softmax_fwd(ac, bd, kl, 0.125)  # unpadded: ld_s = T, ld_w = W
    acr, bdr = ac.double().requires_grad_(True), bd.double().requires_grad_(True)
    idx = (T - 1 - torch.arange(T)[:, None] + torch.arange(T)[None, :]).cuda()
    shifted = torch.gather(bdr, 3, idx.expand(H, B, T, T))
    sc = (acr + shifted) * 0.125
    dead = (torch.arange(T).cuda()[None, :] >= kl[:, None])[None, :, None, :]
    ref = torch.softmax(sc.masked_fill(dead, torch.finfo(torch.float32).min), -1).masked_fill(dead, 0.0)
    _close(attn, ref, 1e-5)
    da = torch.randn_like(attn)
    ref.backward(da.double())
    ds, sk = ops.softmax_bwd(attn, da, 0.125, skew=True)
    _close(ds, acr.grad, 1e-4)
    _close(sk[..., :W], bdr.grad, 1e-4)


def test_plain_causal_softmax():
    from tavsr import ops
    torch.manual_seed(2)
    H, B, L, T = 4, 3, 13, 29
    sc = torch.randn(H, B, L, L, device="cuda")
    kl = torch.tensor([13, 8, 2], device="cuda")
    attn = ops.softmax_fwd(sc, None, kl, 0.5, causal=True)
    mask = (torch.arange(L).cuda()[None, None, :] < kl[:, None, None]) & torch.tril(torch.ones(L, L, device="cuda")).bool()[None]
    dead = ~mask[None]
    ref = torch.softmax((sc.double() * 0.5).masked_fill(dead, -1e300), -1).masked_fill(dead, 0.0)
    _close(attn, ref, 1e-5)
    sc2 = torch.randn(H, B, L, T, device="cuda")
    kl2 = torch.tensor([29, 11, 20], device="cuda")
    a2 = ops.softmax_fwd(sc2, None, kl2, 0.5)
    dead2 = (torch.arange(T).cuda()[None, :] >= kl2[:, None])[None, :, None, :]
    _close(a2, torch.softmax((sc2.double() * 0.5).masked_fill(dead2, -1e300), -1).masked_fill(dead2, 0.0), 1e-5)


@pytest.mark.parametrize("B,T", [(3, 23), (2, 99), (1, 300)])
def test_dwconv_gate(B, T):
    from tavsr import ops
    torch.manual_seed(3)
    Cn, K = 1024, 31
    g = torch.randn(B * T, 2 * Cn, device="cuda")
    gn = torch.randn(B * T, Cn, device="cuda")
    w, bias = torch.randn(Cn, 1, K, device="cuda") / 5, torch.randn(Cn, device="cuda")
    out, conv = ops.dwconv_gate_fwd(gn, g[:, :Cn], w.view(Cn, K), bias, B, T)
    gnr, wr, br = gn.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    rr = g[:, :Cn].double().clone().requires_grad_(True)
    cr = F.conv1d(gnr.view(B, T, Cn).transpose(1, 2), wr, br, 1, 15, 1, Cn).transpose(1, 2).reshape(B * T, Cn)
    ref = rr * cr
    _close(out, ref, 1e-5)
    _close(conv, cr, 1e-5)
    du = torch.randn_like(out)
    ref.backward(du.double())
    dg = torch.empty_like(g)
    dgn, dw, db = ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, w.view(Cn, K), dg[:, :Cn], B, T)
    _close(dg[:, :Cn], rr.grad, 1e-4)
    _close(dgn, gnr.grad, 1e-4)
    _close(dw.view(Cn, 1, K), wr.grad, 1e-4)
    _close(db, br.grad, 1e-4)
    # g = gelu(z): both halves' gradients w.r.t. z written by the two backward kernels themselves (tavsr_dwconv_gate_bwd_act,
    # tavsr_layernorm_bwd_act) == the activation-backward pass applied afterwards, bit for bit (same products, same order)
    z = torch.randn(B * T, 2 * Cn, device="cuda")
    lw = 1 + 0.1 * torch.randn(Cn, device="cuda")
    m_, r_ = ops.layernorm_fwd(g[:, Cn:], lw, lw, 1e-12)[1:]
    want = dg.clone()
    _, gw_, gb_ = ops.layernorm_bwd(dgn, g[:, Cn:], m_, r_, lw, dx=want[:, Cn:])
    ops.act_bwd_(want, z, "gelu")
    got = torch.empty_like(dg)
    dgn2, dw2, db2 = ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, w.view(Cn, K), got[:, :Cn], B, T, zr=z[:, :Cn])
    _, gw2, gb2 = ops.layernorm_bwd_act(dgn2, g[:, Cn:], m_, r_, lw, z[:, Cn:], "gelu", dx=got[:, Cn:])
    assert torch.equal(dgn2, dgn) and torch.equal(dw2, dw) and torch.equal(db2, db)
    assert torch.equal(gw2, gw_) and torch.equal(gb2, gb_)
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,T,Cn", [(2, 1, 1024), (1, 129, 1024), (3, 16, 512), (1, 257, 256)])
def test_csgu_fused_forward_edge_shapes(B, T, Cn):
    """one-row utterances (the window is almost all padding), a second time tile of one row, gate halves of 8 / 4 tiles
    (fewer partial pairs than the 16 lanes that reduce them) - with the statistics from the GEMM epilogue and from the launch"""
    from tavsr import ops
    torch.manual_seed(B * 1000 + T)
    K = 31
    x = torch.randn(B * T, 256, device="cuda")
    w1, b1 = torch.randn(2 * Cn, 256, device="cuda") / 16, 0.1 * torch.randn(2 * Cn, device="cuda")
    lw, lb = 1 + 0.1 * torch.randn(Cn, device="cuda"), 0.1 * torch.randn(Cn, device="cuda")
    w, bias = torch.randn(Cn, 1, K, device="cuda") / 5, torch.randn(Cn, device="cuda")
    rst = torch.empty(B * T, 2 * Cn // 64, 2, device="cuda")
    g = ops.linear(x, w1, b1, act="gelu", rowstat=rst)
    gd = g.double()
    gnr = F.layer_norm(gd[:, Cn:], (Cn,), lw.double(), lb.double(), 1e-12)
    cr = F.conv1d(gnr.view(B, T, Cn).transpose(1, 2), w.double(), bias.double(), 1, 15, 1, Cn).transpose(1, 2).reshape(B * T, Cn)
    for stats in (rst, None):
        u, conv, gn, mean, rstd, _ = ops.csgu_fwd(g, lw, lb, 1e-12, w.view(Cn, K), bias, B, T, save=True, rowstat=stats)
        _close(gn, gnr, 2e-5)
        _close(conv, cr, 2e-5)
        _close(u, gd[:, :Cn] * cr, 2e-5)
        _close(mean, gd[:, Cn:].mean(1), 2e-5)


@pytest.mark.parametrize("B,T,p", [(3, 23, 0.0), (32, 99, 0.0), (2, 99, 0.1), (1, 300, 0.2), (2, 128, 0.0)])
def test_csgu_fused_forward(B, T, p):
    """tavsr_csgu_fwd (LayerNorm statistics + normalise / 31-tap depthwise convolution / gate / dropout in one pass over g)
    against fp64 torch of espnet's ConvolutionalSpatialGatingUnit, and against the LayerNorm + dwconv_gate + dropout launches
    (same mask: the dropout follows tavsr_dropout's mapping)."""
    from tavsr import ops
    torch.manual_seed(31)
    Cn, K = 1024, 31
    g = torch.randn(B * T, 2 * Cn, device="cuda") * 1.3 + 0.2
    lw, lb = 1 + 0.1 * torch.randn(Cn, device="cuda"), 0.1 * torch.randn(Cn, device="cuda")
    w, bias = torch.randn(Cn, 1, K, device="cuda") / 5, torch.randn(Cn, device="cuda")
    ops.manual_seed(77)
    u, conv, gn, mean, rstd, tok = ops.csgu_fwd(g, lw, lb, 1e-12, w.view(Cn, K), bias, B, T, p=p, save=True)
    gd = g.double()
    gnr = F.layer_norm(gd[:, Cn:], (Cn,), lw.double(), lb.double(), 1e-12)
    cr = F.conv1d(gnr.view(B, T, Cn).transpose(1, 2), w.double(), bias.double(), 1, 15, 1, Cn).transpose(1, 2).reshape(B * T, Cn)
    _close(gn, gnr, 1e-5)
    _close(conv, cr, 1e-5)
    _close(mean, gd[:, Cn:].mean(1), 1e-5)
    _close(rstd, 1 / torch.sqrt(gd[:, Cn:].var(1, unbiased=False) + 1e-12), 1e-5)
    ref = gd[:, :Cn] * cr
    if p:
        keep = ops.dropout(torch.ones(B * T, Cn, device="cuda"), p, token=tok)[0] != 0
        assert abs(float(keep.float().mean()) - (1 - p)) < 1e-2
        ref = ref * keep / (1 - p)
    else:
        assert tok is None
    _close(u, ref, 1e-5)
    # eval form: nothing saved, same output
    ops.manual_seed(77)
    u2, conv2, gn2, _, _, _ = ops.csgu_fwd(g, lw, lb, 1e-12, w.view(Cn, K), bias, B, T, p=p, save=False)
    assert conv2 is None and gn2 is None and torch.equal(u2, u)
    # statistics from the producing GEMM's epilogue (tavsr_gemm_desc.rowstat) instead of the statistics launch
    x = torch.randn(B * T, 256, device="cuda")
    w1, b1 = torch.randn(2 * Cn, 256, device="cuda") / 16, 0.1 * torch.randn(2 * Cn, device="cuda")
    rst = torch.empty(B * T, 2 * Cn // 64, 2, device="cuda")
    g2 = ops.linear(x, w1, b1, act="gelu", rowstat=rst)
    g2d = g2.double().view(B * T, 2 * Cn // 64, 64)
    _close(rst[..., 0], g2d.sum(-1), 1e-5)
    _close(rst[..., 1], (g2d * g2d).sum(-1), 1e-5)
    assert torch.equal(g2, ops.linear(x, w1, b1, act="gelu"))
    ops.manual_seed(78)
    ua, ca, gna, ma, ra, _ = ops.csgu_fwd(g2, lw, lb, 1e-12, w.view(Cn, K), bias, B, T, p=p, save=True, rowstat=rst)
    ops.manual_seed(78)
    ub, cb, gnb, mb_, rb, _ = ops.csgu_fwd(g2, lw, lb, 1e-12, w.view(Cn, K), bias, B, T, p=p, save=True)
    _close(ma, mb_, 1e-5)
    _close(ra, rb, 1e-5)
    _close(gna, gnb, 1e-5)
    _close(ua, ub, 1e-5)


@pytest.mark.parametrize("T", [23, 99, 128, 150])     # <= 128: the single-read kernel; beyond: the two-pass one
def test_merge_learned_ave(T):
    from tavsr import ops
    torch.manual_seed(4)
    B, D = 3, 256
    lens = torch.tensor([T, (3 * T) // 4, max(1, T // 3)], device="cuda")
    x1, x2 = torch.randn(B, T, D, device="cuda"), torch.randn(B, T, D, device="cuda")
    prm = [torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4, torch.randn(1, device="cuda"),
           torch.randn(1, device="cuda"), torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4,
           torch.randn(1, device="cuda"), torch.randn(1, device="cuda")]
    score, pooled, w = ops.merge_pool_fwd(x1.view(-1, D), x2.view(-1, D), lens, prm, B, T)
    m = ops.merge_combine(x1.view(-1, D), x2.view(-1, D), w, B, T)
    P = [p.double().requires_grad_(True) for p in prm]
    X = [x1.double().requires_grad_(True), x2.double().requires_grad_(True)]
    mask = (torch.arange(T).cuda()[None, :] < lens[:, None])[:, None, :]
    ws = []
    for k in range(2):
        s = (X[k] @ P[k].t() + P[2 + k]).transpose(1, 2) / D ** 0.5
        s = torch.softmax(s.masked_fill(~mask, -1e300), -1).masked_fill(~mask, 0.0)
        pl = torch.matmul(s, X[k]).squeeze(1)
        ws.append(pl @ P[4 + k].t() + P[6 + k])
    mw = torch.softmax(torch.cat(ws, -1), -1)
    ref = mw[:, 0, None, None] * X[0] + mw[:, 1, None, None] * X[1]
    _close(w, mw, 1e-5)
    _close(m.view(B, T, D), ref, 1e-5)
    dm = torch.randn(B * T, D, device="cuda")
    ref.backward(dm.view(B, T, D).double())
    # pooling + combination in one call, both routes: one workgroup per utterance (one launch for T <= 128; the same four
    # results as the two calls above) and the row-parallel launches (row dots instead of the pooled vectors)
    keep = ops.MERGE_ROWS
    try:
        for rows in (False, True):
            ops.MERGE_ROWS = rows
            score2, aux, w2, m2 = ops.merge_fwd(x1.view(-1, D), x2.view(-1, D), lens, prm, B, T)
            if rows:
                assert aux.shape == (4, B * T)
                _close(aux[2].view(B, T), (X[0] @ P[4].t()).squeeze(-1), 1e-5)
                _close(score2, score, 1e-5)
                _close(w2, w, 1e-5)
            else:
                assert torch.equal(score2, score) and torch.equal(aux, pooled) and torch.equal(w2, w)
            _close(m2.view(B, T, D), ref, 1e-5)
            dx1, dx2, grads = ops.merge_bwd(dm, x1.view(-1, D), x2.view(-1, D), lens, prm, score2, aux, w2, B, T)
            _close(dx1.view(B, T, D), X[0].grad, 1e-4)
            _close(dx2.view(B, T, D), X[1].grad, 1e-4)
            for i in (0, 1, 4, 5, 6, 7):
                _close(grads[i].view(-1), P[i].grad.view(-1), 2e-4)
            for i in (2, 3):  # d/d(pooling bias) is analytically 0 (softmax shift invariance)
                assert grads[i].abs().max() < 1e-5
            # the branch outputs' dropout masks applied by the backward itself == the stand-alone dropout kernel on dx1 / dx2
            ops.manual_seed(3)
            t1, t2 = ops._new_token(0.1, B * T * D, dm.device), ops._new_token(0.25, B * T * D, dm.device)
            for d1, d2 in ((t1, t2), (t1, None), (None, t2)):
                e1, e2, g2 = ops.merge_bwd(dm, x1.view(-1, D), x2.view(-1, D), lens, prm, score2, aux, w2, B, T, drop1=d1, drop2=d2)
                assert torch.equal(e1, dx1 if d1 is None else ops.dropout(dx1, d1[0], token=d1)[0])
                assert torch.equal(e2, dx2 if d2 is None else ops.dropout(dx2, d2[0], token=d2)[0])
                assert all(torch.equal(a, b) for a, b in zip(g2, grads))
    finally:
        ops.MERGE_ROWS = keep


def test_merge_rows_second_length_vector_and_batch_32():
    """the row-parallel merge with per-stream lengths (the AV fusion's use, adaptive_audiovisual_fusion.py:146-179) at the
    encoder's batch, against the one-workgroup-per-utterance launches."""
    from tavsr import ops
    torch.manual_seed(9)
    B, T, D = 32, 99, 256
    lens = torch.randint(1, T + 1, (B,), device="cuda")
    lens2 = torch.randint(1, T + 1, (B,), device="cuda")
    x1, x2, dm = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    prm = [torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4, torch.randn(1, device="cuda"),
           torch.randn(1, device="cuda"), torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4,
           torch.randn(1, device="cuda"), torch.randn(1, device="cuda")]
    keep = ops.MERGE_ROWS
    res = []
    try:
        for rows in (False, True):
            ops.MERGE_ROWS = rows
            score, aux, w, m = ops.merge_fwd(x1, x2, lens, prm, B, T, lens2=lens2)
            dx1, dx2, grads = ops.merge_bwd(dm, x1, x2, lens, prm, score, aux, w, B, T, lens2=lens2)
            res.append((score, w, m, dx1, dx2, *[grads[i] for i in (0, 1, 4, 5, 6, 7)]))
    finally:
        ops.MERGE_ROWS = keep
    for a, b in zip(*res):
        _close(a, b, 2e-5)


@pytest.mark.parametrize("B,T,p", [(3, 99, 0.0), (32, 99, 0.1), (2, 100, 0.1), (4, 37, 0.0), (2, 300, 0.1), (1, 1, 0.0)])
def test_merge_proj_fused_tail_matches_fp64_and_the_launches_it_replaces(B, T, p):
    """tavsr_merge_proj_fwd (csrc/mergeproj.hip): learned_ave merge + merge_proj + dropout + coeff + residual in one launch
    (src/encoder/branchformer/encoder_layer.py:232-300) vs torch fp64, and - with dropout - vs the row-parallel merge followed by
    the GEMM launch under the SAME mask token; the saved tensors feed the existing backward."""
    from tavsr import ops
    torch.manual_seed(11)
    D = 256
    lens = torch.randint(1, T + 1, (B,), device="cuda")
    lens[0] = T
    x1, x2, res = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    prm = [torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4, torch.randn(1, device="cuda"),
           torch.randn(1, device="cuda"), torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4,
           torch.randn(1, device="cuda"), torch.randn(1, device="cuda")]
    W, bias = torch.randn(D, D, device="cuda") / 16, torch.randn(D, device="cuda")
    coeff = 0.7
    assert ops.merge_proj_ok(x1, x2, W, T, D, res=res)
    ops.manual_seed(77)
    ops.rng_step_begin(x1.device)
    score, dots, wts, mix, out, tok = ops.merge_proj_fwd(x1, x2, lens, prm, W, bias, res, coeff, p, B, T)
    # the launches it replaces
    score_r, dots_r, wts_r, mix_r = ops.merge_fwd(x1, x2, lens, prm, B, T)
    _close(score, score_r, 1e-5)
    _close(dots, dots_r, 1e-5)
    _close(wts, wts_r, 1e-5)
    _close(mix, mix_r, 1e-5)
    # fp64
    X = [x1.view(B, T, D).double(), x2.view(B, T, D).double()]
    P = [q.double() for q in prm]
    mask = (torch.arange(T).cuda()[None, :] < lens[:, None])[:, None, :]
    ws = []
    for k in range(2):
        s = (X[k] @ P[k].t() + P[2 + k]).transpose(1, 2) / D ** 0.5
        s = torch.softmax(s.masked_fill(~mask, -1e300), -1).masked_fill(~mask, 0.0)
        ws.append(torch.matmul(s, X[k]).squeeze(1) @ P[4 + k].t() + P[6 + k])
    mw = torch.softmax(torch.cat(ws, -1), -1)
    m64 = (mw[:, 0, None, None] * X[0] + mw[:, 1, None, None] * X[1]).view(B * T, D)
    _close(wts, mw, 1e-5)
    _close(mix, m64, 1e-5)
    y64 = m64 @ W.double().t() + bias.double()
    if p == 0.0:
        assert tok is None
        _close(out, res.double() + coeff * y64, 2e-5)
    else:
        # same mask as a dropout launch with the fused call's token on a [B*T][256] tensor
        yd, _ = ops.dropout(y64.float().contiguous(), p, token=tok)
        keep = yd != 0
        assert 0.85 < float(keep.float().mean()) < 0.95
        _close(out, res.double() + coeff * torch.where(keep, y64 / (1 - p), torch.zeros_like(y64)), 2e-5)
    # without the saved mix (a forward without a backward): same result
    ops.manual_seed(77)
    ops.rng_step_begin(x1.device)
    _, _, _, none_mix, out2, _ = ops.merge_proj_fwd(x1, x2, lens, prm, W, bias, res, coeff, p, B, T, save=False)
    assert none_mix is None and torch.equal(out2, out)
    # the row dots handed in by the launches that produce the branch outputs (tavsr_gemm_desc.rowdot_*): x_k = dropout(Linear(c_k)), the
    # merge's pooling / branch-weight projections of x_k per 64-column tile from that launch's epilogue - same tail, no re-read of the rows
    c1, c2 = torch.randn(B * T, 256, device="cuda"), torch.randn(B * T, 1024, device="cuda")
    w1, b1, w2, b2 = torch.randn(D, 256, device="cuda") / 16, torch.randn(D, device="cuda"), torch.randn(D, 1024, device="cuda") / 32, torch.randn(D, device="cuda")
    assert ops.rowdot_ok(c1, w1) and ops.rowdot_ok(c2, w2)
    ops.manual_seed(78)
    ops.rng_step_begin(x1.device)
    y1, t1, rd1 = ops.linear_drop(c1, w1, b1, p, rowdot=(prm[0], prm[4]))
    y2, t2, rd2 = ops.linear_drop(c2, w2, b2, p, rowdot=(prm[1], prm[5]))
    ops.manual_seed(78)
    ops.rng_step_begin(x1.device)
    z1, u1 = ops.linear_drop(c1, w1, b1, p)
    z2, u2 = ops.linear_drop(c2, w2, b2, p)
    assert (t1 is None) == (u1 is None)
    _close(y1, z1, 2e-6)              # (the row-dot form runs without a K split: equal up to the order of the K = 1024 sum)
    _close(y2, z2, 2e-6)
    for y, rdk, k in ((y1, rd1, 0), (y2, rd2, 1)):
        want = torch.stack([(y.double().view(-1, 4, 64) * prm[k].double().view(1, 4, 64)).sum(-1),
                            (y.double().view(-1, 4, 64) * prm[4 + k].double().view(1, 4, 64)).sum(-1)], -1)
        _close(rdk, want, 2e-5)
    ops.manual_seed(79)
    ops.rng_step_begin(x1.device)
    sc_a, dots_a, wts_a, mix_a, out_a, _ = ops.merge_proj_fwd(y1, y2, lens, prm, W, bias, res, coeff, p, B, T)
    ops.manual_seed(79)
    ops.rng_step_begin(x1.device)
    sc_b, dots_b, wts_b, mix_b, out_b, _ = ops.merge_proj_fwd(y1, y2, lens, prm, W, bias, res, coeff, p, B, T, rowdots=(rd1, rd2))
    for a_, b_ in ((sc_a, sc_b), (dots_a, dots_b), (wts_a, wts_b), (mix_a, mix_b), (out_a, out_b)):
        _close(b_, a_, 2e-5)
    # the saved tensors drive the existing backward
    dm = torch.randn(B * T, D, device="cuda")
    a = ops.merge_bwd(dm, x1, x2, lens, prm, score, dots, wts, B, T)
    b = ops.merge_bwd(dm, x1, x2, lens, prm, score_r, dots_r, wts_r, B, T)
    _close(a[0], b[0], 2e-5)
    _close(a[1], b[1], 2e-5)


def test_conv2d_subsampling_pieces():
    from tavsr import ops
    torch.manual_seed(5)
    B, T, Fq, Cn = 2, 40, 80, 256
    x = torch.randn(B, T, Fq, device="cuda")
    w1, b1 = torch.randn(Cn, 1, 3, 3, device="cuda") / 3, torch.randn(Cn, device="cuda")
    y1 = ops.conv1_fwd(x, w1.view(Cn, 9), b1)
    r1 = F.relu(F.conv2d(x.double().unsqueeze(1), w1.double(), b1.double(), 2))  # B,C,T1,F1
    _close(y1, r1.permute(0, 2, 3, 1), 1e-5)
    col, T2, F2 = ops.im2col3x3s2(y1)
    w2 = torch.randn(Cn, Cn, 3, 3, device="cuda") / 48
    w2r = ops.transpose_inner(w2, Cn, Cn, 9).view(Cn, 9 * Cn)
    y2 = ops.linear(col, w2r, None)
    r2 = F.conv2d(r1, w2.double(), None, 2)
    _close(y2.view(B, T2, F2, Cn), r2.permute(0, 2, 3, 1), 1e-5)
    # col2im(+relu') == conv2 data gradient masked by relu'
    dcol = torch.randn_like(col)
    dz = ops.col2im3x3s2_relu(dcol, y1)
    y1r = y1.double().clone().requires_grad_(True)
    colr = F.unfold(y1r.permute(0, 3, 1, 2), 3, stride=2)  # B, C*9, L  (c-major, then kh,kw)
    colr = colr.view(B, Cn, 9, -1).permute(0, 3, 2, 1).reshape(B * T2 * F2, 9 * Cn)
    (colr * dcol.double()).sum().backward()
    _close(dz, y1r.grad * (y1 > 0), 1e-5)
    # conv1 weight gradient
    dz1 = torch.randn_like(y1)
    dw, db = ops.conv1_bwd(dz1, x, Cn)
    w1r, b1r = w1.double().requires_grad_(True), b1.double().requires_grad_(True)
    (F.conv2d(x.double().unsqueeze(1), w1r, b1r, 2).permute(0, 2, 3, 1) * dz1.double()).sum().backward()
    _close(dw.view(Cn, 1, 3, 3), w1r.grad, 1e-4)
    _close(db, b1r.grad, 1e-4)


@pytest.mark.parametrize("B,T,F_", [(3, 50, 80), (32, 400, 80), (2, 3, 256), (2, 33, 4), (2, 50, 81), (3, 7, 30)])
def test_utterance_mvn(B, T, F_):
    """the 16-byte-column launch (F % 4 == 0: bench batch, one-float4 rows, 64 float4 columns, fewer rows than row groups) and
    the scalar one (other F)"""
    from tavsr import ops
    torch.manual_seed(T)
    x = torch.randn(B, T, F_, device="cuda") + 3.0
    lens = torch.tensor(([T, max(1, (2 * T) // 3), 1] + [T] * B)[:B], device="cuda")
    y = ops.utterance_mvn(x, lens)
    for b, l in enumerate(lens.tolist()):
        _close(y[b, :l], x[b, :l].double() - x[b, :l].double().mean(0, keepdim=True), 1e-5)
        assert (y[b, l:] == 0).all()
    assert torch.equal(y, ops.utterance_mvn(x, lens))


def test_ctc_loss_vs_torch():
    from tavsr import ops
    torch.manual_seed(6)
    B, T, V, L = 5, 37, 41, 12
    logits = torch.randn(B, T, V, device="cuda") * 2
    hl = torch.tensor([37, 30, 37, 5, 20], device="cuda")
    tl = torch.tensor([12, 7, 0, 9, 3], device="cuda")
    ys = torch.randint(1, V, (B, L), device="cuda")
    ys[1, 3] = ys[1, 2]
    nll, g = ops.ctc_loss(logits, hl, ys, tl)
    lr = logits.double().cpu().requires_grad_(True)
    ref = F.ctc_loss(lr.log_softmax(2).transpose(0, 1), ys.cpu(), hl.cpu(), tl.cpu(), blank=0, reduction="none",
                     zero_infinity=True)
    ref.sum().backward()
    _close(nll, ref, 1e-5)
    _close(g, lr.grad, 1e-4)
    assert float(nll[3]) == 0.0 and float(g[3].abs().max()) == 0.0  # infeasible -> zero_infinity


def test_ctc_greedy_bit_exact():
    from tavsr import ops
    from itertools import groupby
    torch.manual_seed(7)
    B, T, V = 6, 99, 41
    logits = torch.randn(B, T, V, device="cuda")
    logits[:, ::3] = logits[:, 1::3][:, : logits[:, ::3].size(1)]  # repeated frames -> repeats to collapse
    logits[0, 5, 7] = logits[0, 5, 3] = 50.0                        # exact tie -> lowest index
    hl = torch.tensor([99, 80, 1, 50, 99, 3], device="cuda")
    ids, hyp, n = ops.ctc_greedy(logits, hl, 0)
    ref = logits.cpu().argmax(-1)
    assert torch.equal(ids.cpu(), ref)
    for b in range(B):
        want = [k for k, _ in groupby(ref[b, : int(hl[b])].tolist()) if k != 0]
        assert hyp[b, : int(n[b])].tolist() == want
        assert (hyp[b, int(n[b]):] == -1).all()


def test_lsm_loss_and_embed():
    from tavsr import ops
    torch.manual_seed(8)
    N, V, D, L = 40, 41, 256, 8
    x = torch.randn(N, V, device="cuda")
    tg = torch.randint(0, V, (N,), device="cuda")
    tg[::5] = -1
    row, g, correct = ops.lsm_loss(x, tg, -1, 0.1)
    xr = x.double().requires_grad_(True)
    td = torch.full((N, V), 0.1 / (V - 1), dtype=torch.float64, device="cuda")
    ign = tg == -1
    td.scatter_(1, tg.masked_fill(ign, 0).unsqueeze(1), 0.9)
    kl = F.kl_div(torch.log_softmax(xr, 1), td, reduction="none").masked_fill(ign.unsqueeze(1), 0).sum(1)
    kl.sum().backward()
    _close(row, kl, 1e-5)
    _close(g, xr.grad, 1e-5)
    want = torch.where(ign, torch.full_like(tg, -1), (x.argmax(1) == tg).long())
    assert torch.equal(correct.long(), want)
    ids = torch.randint(0, V, (N // L, L), device="cuda")
    table, pe = torch.randn(V, D, device="cuda"), torch.randn(L, D, device="cuda")
    out = ops.embed_pe(ids, table, pe, 16.0)
    _close(out, table[ids].double() * 16 + pe.double()[None], 1e-6)
    do = torch.randn(N, D, device="cuda")
    dt = ops.embed_bwd(ids, do, 16.0, V)
    ref = torch.zeros(V, D, dtype=torch.float64, device="cuda").index_add_(0, ids.view(-1), do.double() * 16)
    _close(dt, ref, 1e-5)
    # the decoder's size (32 x 41 tokens, 5000 rows), ids that repeat a lot, N not a multiple of the block; run-to-run equal
    # (... and batches past the kernel's 15 000-token LDS list: chunks that accumulate, ADVICE round 3 - 128 x 120 tokens and more)
    for n_tok, vocab in ((1312, 5000), (1000, 7), (1, 3), (15360, 41), (40000, 41)):
        ids = torch.randint(0, min(vocab, 50), (n_tok,), device="cuda")
        do = torch.randn(n_tok, D, device="cuda")
        dt = ops.embed_bwd(ids.view(1, -1), do, 16.0, vocab)
        ref = torch.zeros(vocab, D, dtype=torch.float64, device="cuda").index_add_(0, ids, do.double() * 16)
        _close(dt, ref, 1e-5)
        assert torch.equal(dt, ops.embed_bwd(ids.view(1, -1), do, 16.0, vocab))


@pytest.mark.parametrize("normalize_length", [False, True])
def test_label_smoothing_loss_fn_matches_the_oracle_class(normalize_length):
    """functional.LabelSmoothingLossFn vs the oracle's LabelSmoothingLoss (espnet: KL sum / batch, or / number of real target
    tokens with length_normalized_loss): value and gradient on the logits."""
    from oracle.leaves import LabelSmoothingLoss
    from tavsr import functional as F_
    torch.manual_seed(3)
    B, L, V = 5, 9, 41
    logits = torch.randn(B, L, V)
    target = torch.randint(0, V, (B, L))
    for b in range(B):
        target[b, L - b:] = -1
    lo = logits.clone().requires_grad_(True)
    want = LabelSmoothingLoss(V, -1, 0.1, normalize_length)(lo, target)
    want.backward()
    lg = logits.cuda().requires_grad_(True)
    got, _ = F_.LabelSmoothingLossFn.apply(lg, target.cuda(), -1, 0.1, normalize_length)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5 * abs(float(want))
    _close(lg.grad.cpu(), lo.grad.double(), 1e-5)


def test_grouped_helpers_match_the_plain_calls():
    """LNGroup (shared reduction, and its immediate path for an odd shape), linear_group (grouped launch and its
    per-projection fallback for K % 32 != 0) and add2_colsum against the single-call forms."""
    from tavsr import ops
    torch.manual_seed(0)
    M, D = 777, 256
    xs = [torch.randn(M, D, device="cuda") for _ in range(3)]
    dys = [torch.randn(M, D, device="cuda") for _ in range(3)]
    gam = [torch.randn(D, device="cuda") for _ in range(3)]
    stats = [ops.layernorm_fwd(x, g, g, 1e-12)[1:] for x, g in zip(xs, gam)]
    ref = [ops.layernorm_bwd(dy, x, m, r, g) for dy, x, (m, r), g in zip(dys, xs, stats, gam)]
    lng = ops.LNGroup(cap=2)                       # the third one overflows the slab: reduced immediately
    got = [lng.bwd(dy, x, m, r, g) for dy, x, (m, r), g in zip(dys, xs, stats, gam)]
    x4, dy4, g4 = torch.randn(50, 1024, device="cuda"), torch.randn(50, 1024, device="cuda"), torch.randn(1024, device="cuda")
    m4, r4 = ops.layernorm_fwd(x4, g4, g4, 1e-12)[1:]
    odd = lng.bwd(dy4, x4, m4, r4, g4)             # other shape than the group's: immediate
    lng.flush()
    for a, b in zip(got, ref):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    for u, v in zip(odd, ops.layernorm_bwd(dy4, x4, m4, r4, g4)):
        assert torch.equal(u, v)
    for K in (256, 200):                           # 200: not a multiple of 32 -> one GEMM per projection
        x = torch.randn(M, K, device="cuda")
        ws = [torch.randn(D, K, device="cuda") for _ in range(3)]
        bs = [torch.randn(D, device="cuda") for _ in range(3)]
        out = torch.empty(M, 3 * D, device="cuda")
        ops.linear_group(x, [(w, b, j * D) for j, (w, b) in enumerate(zip(ws, bs))], out)
        for j, (w, b) in enumerate(zip(ws, bs)):
            assert torch.equal(out[:, j * D:(j + 1) * D], ops.linear(x, w, b))
    a, b = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    buf = torch.empty(M, 3 * D, device="cuda")
    sa, sb = ops.add2_colsum(a, b, buf[:, :D])
    assert torch.equal(buf[:, :D], a + b)
    assert torch.equal(sa, ops.colsum(a)) and torch.equal(sb, ops.colsum(b))


@pytest.mark.parametrize("M,D", [(3168, 256), (777, 256), (50, 1024), (1, 256)])
def test_layernorm_backward_masked_copy_is_the_dropout_kernels(M, D):
    """tavsr_layernorm_bwd_partial_drop: same dx / dgamma / dbeta as the plain pass, and its second output equals the
    stand-alone dropout kernel applied to dx with the same token (bitwise); both the in-launch route and the A/B-off and
    slab-overflow routes of LNGroup.bwd."""
    from tavsr import ops
    torch.manual_seed(M)
    x, dy, add = (torch.randn(M, D, device="cuda") for _ in range(3))
    gam = torch.randn(D, device="cuda")
    mean, rstd = ops.layernorm_fwd(x, gam, gam, 1e-12)[1:]
    ops.manual_seed(11)
    tok = ops._new_token(0.1, M * D, x.device)
    ref = ops.layernorm_bwd(dy, x, mean, rstd, gam, dx_add=add)
    want = ops.dropout(ref[0], 0.1, token=tok)[0]
    assert 0.05 < float((want == 0).float().mean()) < 0.15 or M * D < 1000
    keep = ops.LN_BWD_DROP
    try:
        for on, cap in ((True, 8), (False, 8), (True, 0)):
            ops.LN_BWD_DROP = on
            lng = ops.LNGroup(cap=cap)
            dx, g1, g2, dxd = lng.bwd(dy, x, mean, rstd, gam, dx_add=add, drop=tok)
            lng.flush()
            assert torch.equal(dx, ref[0]) and torch.equal(g1, ref[1]) and torch.equal(g2, ref[2])
            assert torch.equal(dxd, want)
            assert len(lng.bwd(dy, x, mean, rstd, gam)) == 3            # no token: the three-result form
            lng.flush()
    finally:
        ops.LN_BWD_DROP = keep


def test_multi_add_sums_lists_of_tensors_in_place():
    from tavsr import ops
    torch.manual_seed(4)
    sizes = [1, 3, 256, 257, 2048 * 256, 5, 64] * 5            # 35 tensors: two launches; unaligned views included
    base = torch.randn(sum(sizes) + 1, device="cuda")
    dst, src, off = [], [], 1                                   # off = 1: views that are not 16-byte aligned
    for n in sizes:
        dst.append(base[off: off + n])
        src.append(torch.randn(n, device="cuda"))
        off += n
    want = [d.clone() + s for d, s in zip(dst, src)]
    out = ops.multi_add_(dst, src)
    for o, d, w in zip(out, dst, want):
        assert o.data_ptr() == d.data_ptr() and torch.equal(d, w)


@pytest.mark.parametrize("B,H,W,C", [(4, 33, 19, 64), (2, 199, 39, 256)])
def test_unpadded_strided_3x3_implicit_conv(B, H, W, C):
    """conv_taps 90 of the implicit-GEMM loader (3x3 window, stride 2, NO padding: espnet Conv2dSubsampling's second
    convolution) vs torch conv2d in fp64: forward with bias + ReLU, weight and bias gradient."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(H)
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    w = torch.randn(C, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)
    b = torch.randn(C, device="cuda", generator=g)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    w2d = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()                 # (co, kh, kw, ci)
    y = ops.conv3x3_fwd(x.view(-1, C), w2d, H, W, stride=2, pad0=True, bias=b, act="relu")
    ref = torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=2))
    _close(y.view(B, Ho, Wo, C), ref.permute(0, 2, 3, 1), 2e-6)
    if (B * Ho * Wo) % 32 == 0:
        dz = torch.randn(B * Ho * Wo, C, device="cuda", generator=g)
        gw, gb = ops.conv3x3_dw(dz, x.view(-1, C), H, W, stride=2, pad0=True, bias_grad=True)
        wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
        torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wr, br, stride=2).backward(
            dz.double().view(B, Ho, Wo, C).permute(0, 3, 1, 2))
        _close(gw.view(C, 3, 3, C), wr.grad.permute(0, 2, 3, 1), 1e-5)
        _close(gb, br.grad, 1e-5)


@pytest.mark.gpu
def test_cut_to_longest_reads_the_maximum_without_a_stale_answer():
    """models/espnet_model.py:host_max - the collate functions' host value, else one device read per tensor object and
    version (espnet_model.py:372 cuts by ``lengths.max()`` every step)."""
    from tavsr.models import espnet_model as EM
    x = torch.arange(4 * 10, dtype=torch.float32, device="cuda").view(4, 10)
    lens = torch.tensor([3, 7, 5, 2], device="cuda")
    assert EM.cut_to_longest(x, lens).shape == (4, 7)
    assert EM._MAX_SEEN[id(lens)][2] == 7
    assert EM.cut_to_longest(x, lens).shape == (4, 7)                 # same object, same version: from the cache
    lens[1] = 4                                                        # in-place change: the version counter moves
    assert EM.cut_to_longest(x, lens).shape == (4, 5)
    key = id(lens)
    del lens
    assert key not in EM._MAX_SEEN                                     # the entry dies with the tensor
    lens2 = torch.tensor([10, 1, 1, 1], device="cuda")
    assert EM.cut_to_longest(x, lens2) is x
    hinted = torch.tensor([2, 2, 2, 2], device="cuda")
    hinted._tavsr_max = (hinted._version, 6)                           # what utils/avsr_dataloader.py attaches
    assert EM.cut_to_longest(x, hinted).shape == (4, 6)
    hinted.clamp_(max=1)                                               # an in-place edit voids the hint
    assert EM.cut_to_longest(x, hinted).shape == (4, 1)
    assert EM.cut_to_longest(x.cpu(), torch.tensor([1, 9, 2, 2])).shape == (4, 9)
