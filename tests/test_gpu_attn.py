"""GPU: the fused attention core (csrc/attn_fused.hip, tavsr_attn_fwd / tavsr_attn_bwd through the C ABI) against a
plain torch fp64 restatement of espnet's RelPositionMultiHeadedAttention / MultiHeadedAttention arithmetic
((q+u)k^T + rel_shift((q+v)p^T), / sqrt(d_k), key mask with finfo.min, softmax, zero fill, dropout,
attn v) - forward, every gradient, ragged key lengths, causal and cross attention, several online-softmax blocks."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

H, DK = 4, 64
D = H * DK


def _ref(qu, qv, k, v, pos, klens, causal, drop_mask=None, keep=1.0):
    """qu / qv [B,T1,H,dk] (= q + u, q + v), k / v [B,T2,H,dk], pos [2*T1-1,H,dk] or None -> ctx [B,T1,H,dk] (fp64)."""
    B, T1 = qu.shape[:2]
    T2 = k.shape[1]
    s = torch.einsum("bihd,bjhd->bhij", qu, k)
    if pos is not None:
        raw = torch.einsum("bihd,chd->bhic", qv, pos)                       # [B,H,T1,2T1-1]
        idx = (T1 - 1 - torch.arange(T1, device=qu.device))[:, None] + torch.arange(T2, device=qu.device)[None, :]
        s = s + raw.gather(3, idx[None, None].expand(B, H, T1, T2))         # rel_shift: bd[i][j] = raw[i][T-1-i+j]
    s = s / math.sqrt(DK)
    ok = torch.arange(T2, device=qu.device)[None, :] < klens[:, None]       # [B,T2]
    ok = ok[:, None, None, :].expand(B, H, T1, T2)
    if causal:
        ok = ok & (torch.arange(T2, device=qu.device)[None, :] <= torch.arange(T1, device=qu.device)[:, None])
    s = s.masked_fill(~ok, torch.finfo(s.dtype).min)
    attn = torch.softmax(s, -1).masked_fill(~ok, 0.0)
    if drop_mask is not None:
        attn = attn * drop_mask / keep
    return torch.einsum("bhij,bjhd->bihd", attn, v), attn


def _inputs(B, T1, T2, rel, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    q, k, v = r(B * T1, D), r(B * T2, D), r(B * T2, D)
    pos = r(2 * T1 - 1, D) if rel else None
    u, vb = (r(D) * 0.5, r(D) * 0.5) if rel else (None, None)
    return q, k, v, pos, u, vb


@pytest.mark.parametrize("B,T1,T2,rel,causal,lens", [
    (3, 99, 99, True, False, [99, 70, 33]),          # the encoder shape: one key block, ragged keys
    (2, 23, 23, True, False, [23, 9]),               # a single partial tile
    (2, 150, 150, True, False, [150, 131]),          # two online-softmax blocks, partial last tile
    (2, 300, 300, True, False, [300, 257]),          # three blocks
    (3, 41, 41, False, True, [41, 30, 5]),           # decoder self-attention (causal)
    (3, 41, 99, False, False, [99, 80, 64]),         # decoder source attention (T1 != T2)
    (1, 32, 32, True, False, [0]),                   # an utterance without any valid key: zeros, no NaN
])
def test_fused_attention_matches_fp64(B, T1, T2, rel, causal, lens):
    from tavsr import functional as F_
    q, k, v, pos, u, vb = _inputs(B, T1, T2, rel, seed=T1 + T2)
    klens = torch.tensor(lens, device="cuda")
    ctx, saved = F_._AttnFused.fwd(q, 0, k, 0, v, 0, B, T1, T2, H, DK, klens, causal, pos=pos, bias_u=u, bias_v=vb)
    assert bool(torch.isfinite(ctx).all())
    # fp64 reference with separate (q + u) / (q + v) leaves, as the kernel reports the two gradients separately
    q4 = q.double().view(B, T1, H, DK)
    qu = (q4 + (u.double().view(H, DK) if rel else 0)).detach().requires_grad_(True)
    qv = (q4 + (vb.double().view(H, DK) if rel else 0)).detach().requires_grad_(True)
    k4 = k.double().view(B, T2, H, DK).detach().requires_grad_(True)
    v4 = v.double().view(B, T2, H, DK).detach().requires_grad_(True)
    p4 = pos.double().view(-1, H, DK).detach().requires_grad_(True) if rel else None
    ref, _ = _ref(qu, qv, k4, v4, p4, klens, causal)
    err = (ctx.double().view(B, T1, H, DK) - ref).abs().max() / ref.abs().max().clamp_min(1e-30)
    assert err < 2e-5, float(err)
    if lens == [0]:
        assert float(ctx.abs().max()) == 0.0
    dctx = torch.randn(B * T1, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    ref.backward(dctx.double().view(B, T1, H, DK))
    dq = torch.full((B * T1, D), float("nan"), device="cuda")
    dk_ = torch.full((B * T2, D), float("nan"), device="cuda")
    dv_ = torch.full((B * T2, D), float("nan"), device="cuda")
    dqv, dp = F_._AttnFused.bwd(dctx, ctx, saved, q, 0, k, 0, v, 0, dq, 0, dk_, 0, dv_, 0, B, T1, T2, H, DK, klens, causal,
                                pos=pos, bias_u=u, bias_v=vb)

    def close(a, b, name):
        b = b.reshape(a.shape)
        assert bool(torch.isfinite(a).all()), name
        scale = b.abs().max().clamp_min(1e-30)
        if float(scale) < 1e-20:
            assert float(a.abs().max()) < 1e-6, name
            return
        e = (a.double() - b).abs().max() / scale
        assert e < 5e-5, (name, float(e))

    close(dq, qu.grad, "dq")
    close(dk_, k4.grad, "dk")
    close(dv_, v4.grad, "dv")
    if rel:
        close(dqv, qv.grad, "dqv")
        close(dp, p4.grad, "dpos")


def test_fused_attention_on_strided_windows_of_one_buffer():
    """q / k / v as the three column windows of one [B*T, 3*D] projection output, gradients into windows likewise."""
    from tavsr import functional as F_
    B, T = 2, 57
    q, k, v, pos, u, vb = _inputs(B, T, T, True, seed=3)
    qkv = torch.cat([q, k, v], dim=1).contiguous()
    klens = torch.tensor([57, 40], device="cuda")
    c0, s0 = F_._AttnFused.fwd(q, 0, k, 0, v, 0, B, T, T, H, DK, klens, False, pos=pos, bias_u=u, bias_v=vb)
    c1, s1 = F_._AttnFused.fwd(qkv, 0, qkv, D, qkv, 2 * D, B, T, T, H, DK, klens, False, pos=pos, bias_u=u, bias_v=vb)
    assert torch.equal(c0, c1)
    dctx = torch.randn(B * T, D, device="cuda")
    dq, dk_, dv_ = (torch.empty(B * T, D, device="cuda") for _ in range(3))
    dqv0, dp0 = F_._AttnFused.bwd(dctx, c0, s0, q, 0, k, 0, v, 0, dq, 0, dk_, 0, dv_, 0, B, T, T, H, DK, klens, False, pos=pos,
                                  bias_u=u, bias_v=vb)
    dqkv = torch.empty(B * T, 3 * D, device="cuda")
    dqu = torch.empty(B * T, D, device="cuda")
    dqv1, dp1 = F_._AttnFused.bwd(dctx, c1, s1, qkv, 0, qkv, D, qkv, 2 * D, dqu, 0, dqkv, D, dqkv, 2 * D, B, T, T, H, DK, klens,
                                  False, pos=pos, bias_u=u, bias_v=vb)
    assert torch.equal(dqu, dq) and torch.equal(dqkv[:, D:2 * D], dk_) and torch.equal(dqkv[:, 2 * D:], dv_)
    assert torch.equal(dqv0, dqv1) and torch.equal(dp0, dp1)


@pytest.mark.parametrize("rel", [True, False])
def test_fused_attention_dropout_forward_and_backward_share_the_mask(rel):
    """the dropped probabilities are read out through one-hot values (ctx row = dropped attention row), the mask they
    imply is fed to the fp64 reference, and the fused backward - which regenerates the mask from the token - must agree."""
    from tavsr import functional as F_
    from tavsr import ops
    B, T, p = 2, 61, 0.25
    q, k, v, pos, u, vb = _inputs(B, T, T, rel, seed=11)
    klens = torch.tensor([61, 45], device="cuda")
    onehot = torch.zeros(B, T, H, DK, device="cuda")
    for j in range(T):
        onehot[:, j, :, j] = 1.0
    ops.manual_seed(321)
    ctx1, saved1 = F_._AttnFused.fwd(q, 0, k, 0, onehot.view(B * T, D), 0, B, T, T, H, DK, klens, False, pos=pos, bias_u=u,
                                     bias_v=vb, p_att=p)
    pd = ctx1.view(B, T, H, DK)[..., :T].permute(0, 2, 1, 3).double()              # [B,H,T1,T2] dropped probabilities
    q4 = q.double().view(B, T, H, DK)
    qu0 = q4 + (u.double().view(H, DK) if rel else 0)
    qv0 = q4 + (vb.double().view(H, DK) if rel else 0)
    p4 = pos.double().view(-1, H, DK) if rel else None
    _, attn = _ref(qu0, qv0, k.double().view(B, T, H, DK), onehot.double(), p4, klens, False)
    live = attn > 1e-12
    mask = (pd != 0) & live
    frac = float(mask.sum()) / float(live.sum())
    assert abs(frac - (1 - p)) < 0.02, frac
    assert float((pd[mask] - attn[mask] / (1 - p)).abs().max()) < 1e-5
    # same seed + same site counter -> same token: now real values, forward and backward against the reference under that mask
    ops.manual_seed(321)
    ctx2, saved2 = F_._AttnFused.fwd(q, 0, k, 0, v, 0, B, T, T, H, DK, klens, False, pos=pos, bias_u=u, bias_v=vb, p_att=p)
    qu, qv = qu0.detach().requires_grad_(True), qv0.detach().requires_grad_(True)
    k4 = k.double().view(B, T, H, DK).detach().requires_grad_(True)
    v4 = v.double().view(B, T, H, DK).detach().requires_grad_(True)
    pp = p4.detach().requires_grad_(True) if rel else None
    ref, _ = _ref(qu, qv, k4, v4, pp, klens, False, drop_mask=mask.double(), keep=1 - p)
    assert float((ctx2.double().view(B, T, H, DK) - ref).abs().max() / ref.abs().max()) < 2e-5
    dctx = torch.randn(B * T, D, device="cuda")
    ref.backward(dctx.double().view(B, T, H, DK))
    dq, dk_, dv_ = (torch.empty(B * T, D, device="cuda") for _ in range(3))
    dqv, dp = F_._AttnFused.bwd(dctx, ctx2, saved2, q, 0, k, 0, v, 0, dq, 0, dk_, 0, dv_, 0, B, T, T, H, DK, klens, False, pos=pos,
                                bias_u=u, bias_v=vb)
    for a, b in ((dq, qu.grad), (dk_, k4.grad), (dv_, v4.grad)) + (((dqv, qv.grad), (dp, pp.grad)) if rel else ()):
        b = b.reshape(a.shape)
        assert float((a.double() - b).abs().max() / b.abs().max()) < 5e-5
