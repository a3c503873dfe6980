"""GPU: the streaming feed-forward block (csrc/ffn2.hip: tavsr_ffn2_fwd through the C ABI) against a plain torch fp64
restatement of espnet's LayerNorm(eps 1e-12) -> PositionwiseFeedForward -> scaled residual
(src/encoder/branchformer/encoder_layer.py:191-194,311-316): outputs, the LayerNorms of the output it can emit, everything
it saves for the backward pass, the dropout contract it shares with the GEMM-epilogue path, and every plan of the unit
split (workgroup counts that cut row tiles in different places, both ring depths)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(name, z):
    return z * torch.sigmoid(z) if name == "swish" else torch.relu(z)


def _close(a, b, tol):
    a, b = a.double(), b.double()
    assert a.shape == b.shape
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err < tol, err


def _params(D, N1, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    return (1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / D ** 0.5, 0.1 * r(N1), r(D, N1) / N1 ** 0.5, 0.1 * r(D))


def _ref(x, ln_w, ln_b, w1, b1, w2, b2, act, scale):
    xd = x.double()
    nr = F.layer_norm(xd, (x.shape[1],), ln_w.double(), ln_b.double(), 1e-12)
    zr = nr @ w1.double().t() + b1.double()
    hr = _act(act, zr)
    return xd, nr, zr, hr, xd + scale * (hr @ w2.double().t() + b2.double())


@pytest.fixture(autouse=True)
def _default_plan():
    os.environ.pop("TAVSR_FFN2_CFG", None)
    yield
    os.environ.pop("TAVSR_FFN2_CFG", None)


@pytest.mark.parametrize("M,N1,act,cfg", [
    (3168, 2048, "swish", None),         # the encoder's shape (25 row blocks x 64 hidden tiles, 10 workgroups per block)
    (3168, 2048, "swish", "10,3"),       # three ring stages
    (3168, 2048, "swish", "8,5"),        # five; 8 units per workgroup
    (3168, 2048, "swish", "13,4"),       # uneven parts (4 or 5 units), more workgroups than CUs
    (6400, 2048, "swish", None),         # both modality streams of a tailored AV layer in one call
    (1312, 2048, "relu", None),          # decoder block
    (100, 2048, "swish", None),          # M % 128 != 0: the last row block is padded
    (32, 1024, "relu", None),            # one row tile, fewer units than workgroups
    (300, 1056, "swish", "2,3"),         # 33 hidden tiles in parts of 16 and 17 units
])
def test_ffn2_forward_matches_fp64(M, N1, act, cfg):
    from tavsr import ops
    if cfg:
        os.environ["TAVSR_FFN2_CFG"] = cfg
    D = 256
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=M)
    g2, c2, g3, c3 = (_params(D, N1, seed=M + 7)[i] for i in (0, 1, 0, 1))
    g3 = g3 * 0.5 + 0.3
    x = torch.randn(M, D, device="cuda")
    y, (n, mean, rstd, z, h, t_in, t_out), outs, (m2, r2) = ops.ffn2_fwd(
        x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, act, 0.5, save=True, ln2=((g2, c2), (g3, c3)), ln2_stats=True)
    assert t_in is None and t_out is None
    xd, nr, zr, hr, yr = _ref(x, ln_w, ln_b, w1, b1, w2, b2, act, 0.5)
    _close(n, nr, 2e-6)
    _close(mean, xd.mean(1), 2e-6)
    _close(rstd, 1 / torch.sqrt(xd.var(1, unbiased=False) + 1e-12), 2e-6)
    _close(z, zr, 5e-6)
    _close(h, hr, 5e-6)
    _close(y, yr, 5e-6)
    _close(outs[0], F.layer_norm(yr, (D,), g2.double(), c2.double(), 1e-12), 1e-5)
    _close(outs[1], F.layer_norm(yr, (D,), g3.double(), c3.double(), 1e-12), 1e-5)
    _close(m2, yr.mean(1), 1e-5)
    _close(r2, 1 / torch.sqrt(yr.var(1, unbiased=False) + 1e-12), 1e-5)
    # eval form: nothing saved, no extra LayerNorm: bitwise the same output, and run-to-run reproducible
    y2, saved, outs2, _ = ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, act, 0.5, save=False)
    _close(y2, y, 2e-6)          # (another template instantiation: same arithmetic, not necessarily the same contractions)
    assert saved[0] is None and saved[3] is None and outs2 == []
    y3 = ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, act, 0.5, save=False)[0]
    assert torch.equal(y2, y3)


def test_ffn2_strided_input_rows():
    """x as a column window of a wider buffer (row stride != 256), as the layer's concat buffers hand it over."""
    from tavsr import ops
    M, D, N1 = 200, 256, 2048
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=3)
    big = torch.randn(M, 3 * D, device="cuda")
    x = big[:, D:2 * D]
    y = ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, save=False)[0]
    _close(y, _ref(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5)[-1], 5e-6)


@pytest.mark.parametrize("cfg", [None, "3,5"])
def test_ffn2_dropout_is_the_gemm_path_dropout(cfg):
    """Both dropout sites draw the tavsr_dropout mapping at the offsets of their tokens: the stand-alone dropout kernel
    regenerates the masks from the tokens, i.e. the GEMM-based backward pairs with this forward."""
    from tavsr import ops
    if cfg:
        os.environ["TAVSR_FFN2_CFG"] = cfg
    M, D, N1, p = 515, 256, 2048, 0.2
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=1)
    x = torch.randn(M, D, device="cuda")
    ops.manual_seed(99)
    y, (n, mean, rstd, z, h, t_in, t_out), _, _ = ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, p=p)
    xd, nr, zr, hr, _ = _ref(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5)
    _close(z, zr, 5e-6)
    ones = torch.ones(M, N1, device="cuda")
    in_mask = ops.dropout(ones, p, token=t_in)[0] != 0
    assert abs(float(in_mask.float().mean()) - (1 - p)) < 5e-3
    _close(h, hr * in_mask / (1 - p), 5e-6)
    assert torch.equal(h != 0, in_mask & (hr.float() != 0))
    t = (hr * in_mask / (1 - p)) @ w2.double().t() + b2.double()
    out_mask = ops.dropout(torch.ones(M, D, device="cuda"), p, token=t_out)[0] != 0
    _close(y, xd + 0.5 * t * out_mask / (1 - p), 5e-6)
    # eval-style call of the same pass without saving: same y (the masks do not depend on what is saved)
    ops.manual_seed(99)
    y2 = ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, p=p, save=False)[0]
    _close(y2, y, 2e-6)
    assert torch.equal(y2 == x, y == x)               # the same elements were dropped


@pytest.mark.parametrize("M,N1,act,p,cfg", [
    (3168, 2048, "swish", 0.0, None),
    (3168, 2048, "swish", 0.1, "13,4"),
    (1312, 2048, "relu", 0.1, None),
    (100, 2048, "swish", 0.0, None),       # padded last row block; z handed over with exactly M rows
    (1, 2048, "swish", 0.1, None),         # a single row
    (300, 1056, "relu", 0.2, "2,4"),
])
def test_ffn2_backward_dx_matches_fp64(M, N1, act, p, cfg):
    """tavsr_ffn2_bwd_dx: dz = ((alpha dyd) w2) * mask / keep * act'(z), dn = dz w1 against fp64 autograd of the same
    expression (the mask regenerated from the token by the stand-alone dropout kernel)."""
    from tavsr import ops
    if cfg:
        os.environ["TAVSR_FFN2_CFG"] = cfg
    D = 256
    _, _, w1, _, w2, _ = _params(D, N1, seed=M + 1)
    z = torch.randn(M, N1, device="cuda") * 1.5
    dyd = torch.randn(M, D, device="cuda")
    ops.manual_seed(5)
    tok = ops._new_token(p, M * N1, z.device) if p else None
    dz, dn = ops.ffn2_bwd_dx(dyd, 0.5, w1, w2, z, act, tok)
    zd = z.double().requires_grad_(True)
    _act(act, zd).backward(torch.ones_like(zd))
    dact = zd.grad
    mask = torch.ones(M, N1, device="cuda", dtype=torch.double)
    if p:
        mask = (ops.dropout(torch.ones(M, N1, device="cuda"), p, token=tok)[0] != 0).double() / (1 - p)
    dzr = (0.5 * dyd.double()) @ w2.double() * mask * dact
    _close(dz, dzr, 5e-6)
    _close(dn, dzr @ w1.double(), 5e-6)
    dz2, dn2 = ops.ffn2_bwd_dx(dyd, 0.5, w1, w2, z, act, tok)
    assert torch.equal(dz, dz2) and torch.equal(dn, dn2)


@pytest.mark.parametrize("M,N1,p_out", [(3168, 2048, 0.0), (3168, 2048, 0.1), (1312, 2048, 0.1), (100, 2048, 0.0), (1, 2048, 0.1), (300, 1056, 0.0)])
def test_layernorm_backward_on_the_unsummed_partials_equals_finish_then_layernorm(M, N1, p_out):
    """tavsr_layernorm_bwd_partial_slab (ops.LNGroup.bwd on ops.DnSlabs): the feed-forward block's LayerNorm backward reads dn as the
    partial slabs of tavsr_ffn2_bwd_dx(dn = NULL) and sums them in the finishing launch's order - dx, the masked copy and (dgamma, dbeta)
    bit-equal to the finishing launch + tavsr_layernorm_bwd_partial[_drop] it replaces, at every row-block count (wpb = 3 ... 23)."""
    from tavsr import ops
    D = 256
    ln_w, _, w1, _, w2, _ = _params(D, N1, seed=M + 7)
    z = torch.randn(M, N1, device="cuda") * 1.5
    dyd, x, dy = (torch.randn(M, D, device="cuda") for _ in range(3))
    mean, rstd = x.mean(1), torch.rsqrt(x.var(1, unbiased=False) + 1e-5)
    ops.manual_seed(9)
    tok = ops._new_token(p_out, M * D, z.device) if p_out else None
    res = []
    for sum_dn in (True, False):
        dz, dn = ops.ffn2_bwd_dx(dyd, 0.5, w1, w2, z, "swish", None, sum_dn=sum_dn)
        assert sum_dn or isinstance(dn, ops.DnSlabs)
        lng = ops.LNGroup()
        assert lng.takes(M, D)
        out = lng.bwd(dn, x, mean, rstd, ln_w, dx_add=dy, drop=tok)
        lng.flush()
        res.append((dz,) + tuple(t.clone() for t in out))
    assert len(res[0]) == (5 if p_out else 4)
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_ffn_block_function_streaming_equals_gemm_launches():
    """functional._FFN forward + backward with the streaming forward against the LayerNorm + GEMM + GEMM launches, with the
    recipe's dropout on: same masks, so outputs and every gradient agree."""
    from tavsr import functional as F_
    from tavsr import ops
    M, D, N1 = 999, 256, 2048
    ln_w, ln_b, w1, b1, w2, b2 = _params(D, N1, seed=5)
    x, dy = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    res = []
    keep = ops.FFN2, ops.FFN2_BWD
    for fused, fused_bwd in ((True, True), (False, False), (True, False), (False, True)):
        ops.FFN2, ops.FFN2_BWD = fused, fused_bwd
        ops.manual_seed(4242)
        try:
            y, saved = F_._FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5, p=0.1)
            dx, grads = F_._FFN.bwd(dy, saved, ln_w, w1, w2, "swish", 0.5)
        finally:
            ops.FFN2, ops.FFN2_BWD = keep
        res.append((y, dx) + tuple(grads))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            _close(a, b, 2e-5)
