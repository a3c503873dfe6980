"""GPU: the fused feed-forward block (csrc/ffn.hip: tavsr_ffn_fwd / tavsr_ffn_bwd_dx through the C ABI) against a plain
torch fp64 restatement of espnet's LayerNorm(eps 1e-12) -> PositionwiseFeedForward -> scaled residual
(src/encoder/branchformer/encoder_layer.py:191-194), forward values, everything it saves, and the data-gradient chain."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(name, z):
    return z * torch.sigmoid(z) if name == "swish" else torch.relu(z)


def _close(a, b, tol):
    a, b = a.double(), b.double()
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err < tol, err


def _params(D, N1, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    return (1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / D ** 0.5, 0.1 * r(N1), r(D, N1) / N1 ** 0.5, 0.1 * r(D))


@pytest.mark.parametrize("M,D,N1,act", [(3168, 256, 2048, "swish"), (1312, 256, 2048, "relu"), (100, 256, 2048, "swish"),
                                        (77, 256, 384, "swish"), (640, 512, 2048, "relu")])
def test_fused_ffn_forward_and_data_gradient(M, D, N1, act):
    from tavsr import ops
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=M)
    x = torch.randn(M, D, device="cuda")
    y, (n, mean, rstd, z, h, t_in, t_out) = ops.ffn_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, act, 0.5)
    assert t_in is None and t_out is None
    xd = x.double()
    nr = F.layer_norm(xd, (D,), ln_w.double(), ln_b.double(), 1e-12)
    zr = nr @ w1.double().t() + b1.double()
    hr = _act(act, zr)
    yr = xd + 0.5 * (hr @ w2.double().t() + b2.double())
    _close(n, nr, 2e-6)
    _close(mean, xd.mean(1), 2e-6)
    _close(rstd, 1 / torch.sqrt(xd.var(1, unbiased=False) + 1e-12), 2e-6)
    _close(z, zr, 5e-6)
    _close(h, hr, 5e-6)
    _close(y, yr, 5e-6)
    # eval form: nothing saved, same output
    y2, saved = ops.ffn_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, act, 0.5, save=False)
    assert torch.equal(y, y2) and saved[0] is None and saved[3] is None
    # data-gradient chain
    dyd = torch.randn(M, D, device="cuda")
    dz, dn = ops.ffn_bwd_dx(dyd, 0.5, w1, w2, z, act, None)
    zz = z.double().requires_grad_(True)
    _act(act, zz).backward(0.5 * dyd.double() @ w2.double())
    _close(dz, zz.grad, 5e-6)
    _close(dn, zz.grad @ w1.double(), 1e-5)


def test_fused_ffn_dropout_masks_agree_between_forward_and_backward():
    from tavsr import ops
    M, D, N1, p = 515, 256, 2048, 0.2
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=1)
    x = torch.randn(M, D, device="cuda")
    ops.manual_seed(99)
    y, (n, mean, rstd, z, h, t_in, t_out) = ops.ffn_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, p=p)
    hz = _act("swish", z.double())
    mask = h != 0                                     # swish(z) == 0 has measure zero
    assert abs(float(mask.float().mean()) - (1 - p)) < 5e-3
    _close(h, hz * mask / (1 - p), 5e-6)
    # rows share nothing: the mask must not repeat from row to row or column to column
    assert float((mask[1:] == mask[:-1]).float().mean()) < 0.72 and float((mask[:, 1:] == mask[:, :-1]).float().mean()) < 0.72
    # outer dropout: y - x = 0.5 * drop(t): recover its mask and check the rate, then the tavsr_dropout contract
    t = h.double() @ w2.double().t() + b2.double()
    out_mask = (y.double() - x.double()).abs() > 1e-12
    assert abs(float(out_mask.float().mean()) - (1 - p)) < 1e-2
    _close(y, x.double() + 0.5 * t * out_mask / (1 - p), 5e-6)
    tt = t.float().contiguous()
    dropped, _ = ops.dropout(tt, p, token=t_out)      # the standalone dropout kernel with the token reproduces the mask
    assert torch.equal(dropped != 0, out_mask)
    # backward regenerates the inner mask
    dyd = torch.randn(M, D, device="cuda")
    dz, dn = ops.ffn_bwd_dx(dyd, 0.5, w1, w2, z, "swish", t_in)
    zz = z.double().requires_grad_(True)
    (_act("swish", zz) * mask / (1 - p)).backward(0.5 * dyd.double() @ w2.double())
    _close(dz, zz.grad, 5e-6)
    _close(dn, zz.grad @ w1.double(), 1e-5)


def test_ffn_block_function_fused_equals_unfused():
    """functional._FFN on the fused path against the LayerNorm + GEMM + GEMM launches (TAVSR_FFN_FUSED=0 path)."""
    from tavsr import functional as F_
    from tavsr import ops
    M, D, N1 = 999, 256, 2048
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=5)
    x, dy = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    res = []
    for fused in (True, False):
        ops.FFN_FUSED = fused
        try:
            y, saved = F_._FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5)
            dx, grads = F_._FFN.bwd(dy, saved, ln_w, w1, w2, "swish", 0.5)
        finally:
            ops.FFN_FUSED = False
        res.append((y, dx) + tuple(grads))
    for a, b in zip(*res):
        _close(a, b, 2e-5)
