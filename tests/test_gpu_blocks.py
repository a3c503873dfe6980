"""GPU: the block-level C entry points (csrc/blocks.hip) against the launch-by-launch sequencing they replace and against torch
fp64 restatements of the espnet modules they stand for:
  tavsr_cgmlp_fwd / _bwd             ConvolutionalGatingMLP + branch dropout / residual (encoder_layer.py:213-226; tailored layer)
  tavsr_conv2d_subsample_fwd / _bwd  Conv2dSubsampling (branchformer/encoder.py:364; embedding_for_avsr/default.py:111-162)
  tavsr_workspace_bytes              one workspace query for every descriptor."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a, b = a.double().cpu(), b.double().cpu()
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err < tol, err


def _cgmlp_params(D, C2, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
    Cn = C2 // 2
    return [r(C2, D, sc=D ** -0.5), r(C2, sc=0.1), 1.0 + r(Cn, sc=0.1), r(Cn, sc=0.1), r(Cn, 1, 31, sc=0.2), 1.0 + r(Cn, sc=0.1),
            r(D, Cn, sc=Cn ** -0.5), r(D, sc=0.1)]


@pytest.mark.parametrize("B,T,D,C2", [(3, 40, 256, 2048), (2, 99, 256, 2048), (4, 17, 128, 512)])
def test_cgmlp_block_forward_backward_vs_fp64(B, T, D, C2):
    from tavsr import ops
    prm = _cgmlp_params(D, C2, 3)
    w1, b1, lw, lb, cw, cb, w2, b2 = prm
    Cn = C2 // 2
    torch.manual_seed(1)
    x, res, dy = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    out, kept, desc = ops.cgmlp_fwd(x, w1, b1, lw, lb, cw, cb, w2, b2, B, T, alpha=0.7, res=res)
    dx, grads = ops.cgmlp_bwd(desc, dy, prm)
    P = [q.double().requires_grad_(True) for q in prm]
    X = x.double().requires_grad_(True)
    g = F.gelu(X @ P[0].t() + P[1])
    xr, xg = g[:, :Cn], g[:, Cn:]
    xg = F.layer_norm(xg, (Cn,), P[2], P[3], 1e-12)
    xg = F.conv1d(xg.view(B, T, Cn).transpose(1, 2), P[4], P[5], 1, 15, groups=Cn).transpose(1, 2).reshape(B * T, Cn)
    ref = res.double() + 0.7 * ((xr * xg) @ P[6].t() + P[7])
    _close(out, ref, 2e-5)
    ref.backward(dy.double())
    _close(dx, X.grad, 1e-4)
    for a, q in zip(grads, P):
        _close(a, q.grad, 2e-4)


def test_cgmlp_block_with_dropout_equals_the_launches_it_replaces():
    """same masks (same tokens) as LayerNorm-statistics GEMM + CSGU + GEMM issued one by one; backward against the
    launch-by-launch backward of tavsr/functional_av.py on the same kept state"""
    from tavsr import ops
    B, T, D, C2 = 4, 50, 256, 2048
    Cn = C2 // 2
    prm = _cgmlp_params(D, C2, 5)
    w1, b1, lw, lb, cw, cb, w2, b2 = prm
    torch.manual_seed(2)
    x, res, dy = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    ops.manual_seed(9)
    ops.rng_step_begin(x.device)
    out, (g, z, gn, gmean, grstd, u, conv, t_u, t_out), desc = ops.cgmlp_fwd(x, w1, b1, lw, lb, cw, cb, w2, b2, B, T, p=0.1, p_out=0.1,
                                                                              alpha=1.0, res=res)
    dx, grads = ops.cgmlp_bwd(desc, dy, prm)
    # the launches
    ops.manual_seed(9)
    ops.rng_step_begin(x.device)
    rst = ops.empty(B * T, C2 // 64, 2, like=x)
    g2, z2 = ops.linear(x, w1, b1, act="gelu", save_z=True, rowstat=rst)
    u2, conv2, gn2, gm2, gr2, tu2 = ops.csgu_fwd(g2, lw, lb, 1e-12, cw.reshape(Cn, -1), cb, B, T, p=0.1, save=True, rowstat=rst)
    out2, to2 = ops.linear_drop(u2, w2, b2, 0.1, alpha=1.0, res=res)
    assert tu2[1] == t_u[1] and to2[1] == t_out[1]
    for a, b in ((g, g2), (z, z2), (u, u2), (conv, conv2), (gn, gn2), (out, out2)):
        assert torch.equal(a, b)
    dyd = ops.dropout(dy, 0.1, token=to2)[0]
    gw2, gb2 = ops.linear_dw(dyd, u2, bias_grad=True)
    du = ops.linear_dx_drop(dyd, w2, tu2)
    dg = torch.empty_like(g2)
    dgn, gcw, gcb = ops.dwconv_gate_bwd(du, gn2, g2[:, :Cn], conv2, cw.reshape(Cn, -1), dg[:, :Cn], B, T, zr=z2[:, :Cn])
    _, glw, glb = ops.layernorm_bwd_act(dgn, g2[:, Cn:], gm2, gr2, lw, z2[:, Cn:], "gelu", dx=dg[:, Cn:])
    gw1, gb1 = ops.linear_dw(dg, x, bias_grad=True)
    dx2 = ops.linear_dx(dg, w1)
    assert torch.equal(dx, dx2)
    for a, b in zip(grads, (gw1, gb1, glw, glb, gcw.view_as(cw), gcb, gw2, gb2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,T,Fq,Cn,odim", [(2, 67, 80, 64, 256), (32, 400, 80, 256, 256)])
def test_conv2d_subsample_block_vs_fp64_and_the_launches(B, T, Fq, Cn, odim):
    from tavsr import functional as FN, ops
    torch.manual_seed(4)
    x = torch.randn(B, T, Fq, device="cuda")
    T1, F1 = (T - 3) // 2 + 1, (Fq - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    if (B * T2 * F2) % 32:
        pytest.skip("shape outside the implicit route")
    w1, b1 = torch.randn(Cn, 1, 3, 3, device="cuda") / 3, torch.randn(Cn, device="cuda") / 4
    w2, b2 = torch.randn(Cn, Cn, 3, 3, device="cuda") / (3 * Cn ** 0.5), torch.randn(Cn, device="cuda") / 4
    wo, bo = torch.randn(odim, Cn * F2, device="cuda") / (Cn * F2) ** 0.5, torch.randn(odim, device="cuda") / 4
    prm = [w1, b1, w2, b2, wo, bo]
    dout = torch.randn(B, T2, odim, device="cuda")

    def run(block):
        keep = ops.BLOCKS_C
        ops.BLOCKS_C = block
        try:
            P = [q.clone().requires_grad_(True) for q in prm]
            y = FN.Conv2dSubsamplingFn.apply(x, *P, 16.0)
            (y * dout).sum().backward()
            return y.detach(), [q.grad for q in P]
        finally:
            ops.BLOCKS_C = keep

    y_c, g_c = run(True)
    y_p, g_p = run(False)
    assert torch.equal(y_c, y_p)
    for a, b in zip(g_c, g_p):
        assert torch.equal(a, b)
    P = [q.double().requires_grad_(True) for q in prm]
    h = F.relu(F.conv2d(x.double().unsqueeze(1), P[0], P[1], 2))
    h = F.relu(F.conv2d(h, P[2], P[3], 2))
    ref = 16.0 * F.linear(h.transpose(1, 2).contiguous().view(B, T2, Cn * F2), P[4], P[5])
    _close(y_c, ref, 2e-5)
    (ref * dout.double()).sum().backward()
    for a, q in zip(g_c, P):      # (the weight gradients sum 60 k - 250 k positions per tap in fp32 at the encoder's batch: the full-size bar of DESIGN.md section 2, 5e-3)
        _close(a, q.grad, 3e-4 if B * T < 2000 else 5e-3)


def test_workspace_bytes_is_the_per_entry_query_times_four():
    from tavsr._lib import BfLayerDesc, GemmDesc, SubsampleDesc, lib
    fn = lib().tavsr_workspace_bytes
    fn.restype = C.c_int64
    d = BfLayerDesc()
    d.B, d.T, d.D, d.H, d.ffn_units, d.cg_units, d.cg_kernel = 4, 99, 256, 4, 2048, 2048, 31
    per = lib().tavsr_branchformer_layer_ws
    per.restype = C.c_int64
    assert fn(2, C.byref(d)) == 4 * per(C.byref(d)) > 0
    g = GemmDesc()
    g.M, g.N, g.K, g.nb1, g.nb2 = 256, 2048, 3168, 1, 1
    g.a_kmajor = g.b_kmajor = 1
    g.lda, g.ldb, g.ldc = 256, 2048, 2048
    assert fn(0, C.byref(g)) == 4 * lib().tavsr_gemm_ws(C.byref(g))
    s = SubsampleDesc()
    s.B, s.T, s.F, s.C, s.odim = 4, 400, 80, 256, 256
    assert fn(6, C.byref(s)) >= 0
    assert fn(99, C.byref(s)) == -1 and fn(0, None) == 0
