"""GPU: train-mode dropout.  The reference's torch RNG stream cannot be reproduced, so the checks are (1) the kernel's
statistics and its regenerate-the-mask contract, (2) determinism in the seed, fresh masks per step, (3) at model level a
directional finite-difference test of the hand-written backward under a FROZEN mask, and that the expected train loss
matches the no-dropout loss (inverted dropout is unbiased to first order)."""
import argparse

import numpy as np
import pytest
import torch

from helpers import asr_conf

pytestmark = pytest.mark.gpu


def test_dropout_kernel_statistics_and_mask_regeneration():
    from tavsr import ops
    torch.manual_seed(0)
    ops.manual_seed(123)
    x = torch.randn(1 << 20, device="cuda") + 3.0
    for p in (0.1, 0.5):
        y, tok = ops.dropout(x, p)
        kept = (y != 0)
        assert abs(float(kept.float().mean()) - (1 - p)) < 3e-3          # 1M Bernoulli draws: sigma ~ 5e-4
        assert torch.allclose(y[kept], x[kept] / (1 - p), rtol=1e-6)
        g = torch.randn_like(x)
        gy, _ = ops.dropout(g, p, token=tok)                              # backward: the SAME mask
        assert torch.equal(gy != 0, kept)
        y2, _ = ops.dropout(x, p)                                         # next site: a different mask
        assert float(((y2 != 0) ^ kept).float().mean()) > 0.05
    # lag-1 independence of neighbouring elements
    k = (ops.dropout(x, 0.5)[0] != 0).float()
    assert abs(float((k[1:] * k[:-1]).mean()) - 0.25) < 3e-3
    # odd length / unaligned tail
    z = torch.randn(1001, device="cuda")
    yz, tz = ops.dropout(z, 0.3)
    assert yz.shape == z.shape and 0.6 < float((yz != 0).float().mean()) < 0.8


def test_dropout_seed_determinism_and_step_advance():
    from tavsr import ops
    x = torch.ones(4096, device="cuda")
    ops.manual_seed(7)
    a, _ = ops.dropout(x, 0.25)
    ops.manual_seed(7)
    b, _ = ops.dropout(x, 0.25)
    assert torch.equal(a, b)
    ops.manual_seed(7)
    ops.rng_step_begin()              # the per-step advance changes every mask, the site counter restarts
    c, _ = ops.dropout(x, 0.25)
    assert not torch.equal(a, c)
    ops.manual_seed(8)
    d, _ = ops.dropout(x, 0.25)
    assert not torch.equal(a, d)


def _model(nb=2, dropout=0.1):
    from oracle.model import fill_parameters_
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=nb, dec_blocks=1, dropout=dropout)
    model = ASRTask.build_model(argparse.Namespace(**conf))
    fill_parameters_(model, seed=17)
    return model.cuda()


def _batch():
    from oracle.model import synth
    text = synth((3, 8), seed=9, kind="int", lo=1, hi=40)
    text[1, 5:] = -1
    return (synth((3, 120, 80), seed=8).cuda(), torch.tensor([120, 100, 64]).cuda(), text.cuda(), torch.tensor([8, 5, 8]).cuda())


def test_model_backward_under_frozen_masks():
    """d loss / d theta along the gradient direction by central differences, masks frozen by re-seeding."""
    from tavsr import ops
    model = _model().train()
    batch = _batch()

    def loss_at():
        ops.manual_seed(2024)
        return model(*batch)[0]

    model.zero_grad()
    loss_at().backward()
    params = [p for p in model.parameters() if p.grad is not None]
    grads = [p.grad.detach().clone() for p in params]
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    assert np.isfinite(gnorm) and gnorm > 0
    eps = 2e-3 / gnorm * float(torch.sqrt(sum((p.double() ** 2).sum() for p in params)))   # ~0.2 % relative step
    with torch.no_grad():
        for p, g in zip(params, grads):
            p.add_(g, alpha=eps / gnorm)
        lp = float(loss_at())
        for p, g in zip(params, grads):
            p.add_(g, alpha=-2 * eps / gnorm)
        lm = float(loss_at())
        for p, g in zip(params, grads):
            p.add_(g, alpha=eps / gnorm)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gnorm) / gnorm < 3e-2, (fd, gnorm)


def test_backward_regenerates_its_own_forward_masks():
    """fwd(A), fwd(B), bwd(A): the second forward advances the device generator, the backward of the first must still
    regenerate the masks ITS forward drew (the dropout tokens own their step's seed) - gradients bit-identical to a lone
    fwd(A) -> bwd(A) step."""
    from oracle.model import synth
    from tavsr import ops
    model = _model().train()
    batch = _batch()
    other = (synth((3, 120, 80), seed=31).cuda(), batch[1], batch[2], batch[3])

    ops.manual_seed(4242)
    model.zero_grad()
    model(*batch)[0].backward()
    want = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    ops.manual_seed(4242)
    model.zero_grad()
    loss_a = model(*batch)[0]
    with torch.no_grad():
        model(*other)                  # a validation-style forward in between (CTC dropout is live in eval as well)
    loss_b = model(*other)[0]          # ... and a second micro-batch forward
    loss_a.backward()
    for n, p in model.named_parameters():
        if n in want:
            assert torch.equal(p.grad, want[n]), n
    loss_b.backward()                  # the second pass's own masks are intact as well
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_train_loss_is_stochastic_and_close_to_eval_loss():
    from tavsr import ops
    model = _model(dropout=0.1)
    batch = _batch()
    ref = float(_model(dropout=0.0).train()(*batch)[0])
    model.train()
    ops.manual_seed(1)
    losses = []
    with torch.no_grad():
        for _ in range(16):
            losses.append(float(model(*batch)[0]))      # forward() advances the generator: new masks each call
    assert len(set(round(l, 4) for l in losses)) > 8
    assert abs(np.mean(losses) - ref) / ref < 0.15, (np.mean(losses), ref)
    model.eval()
    model.ctc.dropout_rate = 0.0                         # (the reference's CTC dropout is active in eval too: Q7)
    with torch.no_grad():
        e1, e2 = float(model(*batch)[0]), float(model(*batch)[0])
    assert e1 == e2


def test_avsr_model_backward_under_frozen_masks():
    """the AV path's dropout sites (tailored streams, fusion FFN, AV embeddings): same directional check"""
    from helpers import AVSR_YAML, avsr_conf
    from oracle.model import fill_parameters_, synth
    from tavsr import ops
    from tavsr.tasks.avsr import AVSRTask
    import yaml
    conf = avsr_conf(AVSR_YAML, num_blocks=2, dec_blocks=1)
    ref = yaml.safe_load(open(AVSR_YAML))            # put the recipe's dropout rates back
    for k in ("acoustic_embed_conf", "visual_embed_conf", "audiovisual_fusion_conf", "decoder_conf", "ctc_conf"):
        for kk, v in ref[k].items():
            if kk.endswith("dropout_rate"):
                conf[k][kk] = v
    for kk in ("dropout_rate", "positional_dropout_rate", "attention_dropout_rate"):
        conf["encoder_conf"][kk] = ref["encoder_conf"][kk]
    model = AVSRTask.build_model(argparse.Namespace(**conf))
    fill_parameters_(model, seed=23)
    model = model.cuda().train()
    text = synth((2, 6), seed=9, kind="int", lo=1, hi=40)
    batch = (synth((2, 40, 80), seed=1).cuda(), torch.tensor([40, 32]).cuda(), synth((2, 9, 88, 88), seed=2).cuda(),
             torch.tensor([9, 8]).cuda(), text.cuda(), torch.tensor([6, 4]).cuda())
    bufs0 = {k: v.clone() for k, v in model.named_buffers()}

    def loss_at():
        ops.manual_seed(99)
        with torch.no_grad():                        # BatchNorm running buffers must not drift between evaluations
            for k, v in model.named_buffers():
                v.copy_(bufs0[k])
        return model(*batch)[0]

    model.zero_grad()
    loss_at().backward()
    params = [p for p in model.parameters() if p.grad is not None]
    grads = [p.grad.detach().clone() for p in params]
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    pnorm = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in params)))
    eps = 1e-3 * pnorm / gnorm
    with torch.no_grad():
        for p, g in zip(params, grads):
            p.add_(g, alpha=eps / gnorm)
        lp = float(loss_at())
        for p, g in zip(params, grads):
            p.add_(g, alpha=-2 * eps / gnorm)
        lm = float(loss_at())
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gnorm) / gnorm < 5e-2, (fd, gnorm)


@pytest.mark.gpu
def test_fused_dropout_kernels_equal_the_unfused_sequence():
    """dropout_add == dropout then axpby; dropout_act_bwd == dropout (same token) then act'(z) multiply: same masks, and
    the same roundings up to the order of the two scalings."""
    from tavsr import ops
    torch.manual_seed(0)
    a, t = torch.randn(3168, 256, device="cuda"), torch.randn(3168, 256, device="cuda")
    ops.manual_seed(5)
    y, tok = ops.dropout_add(a, t, 0.1, alpha=0.5)
    ops.manual_seed(5)
    td, tok2 = ops.dropout(t, 0.1)
    assert tok == tok2
    ref = ops.axpby(a, td, 1.0, 0.5)
    assert torch.equal(y, ref)
    for act in ("swish", "relu", "gelu"):
        dh, z = torch.randn(777, 2048, device="cuda"), torch.randn(777, 2048, device="cuda")
        ops.manual_seed(9)
        _, tk = ops.dropout(torch.zeros_like(dh), 0.1)
        dz = ops.dropout_act_bwd(dh, z, act, tk)
        ref = ops.act_bwd_(ops.dropout(dh, 0.1, token=tk)[0], z, act)
        assert torch.equal(dz == 0, ref == 0)
        assert float((dz - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("T1,T2,rel,causal", [(99, 99, True, False), (41, 41, False, True), (41, 99, False, False)])
def test_softmax_with_fused_probability_dropout_equals_the_separate_launches(T1, T2, rel, causal):
    """tavsr_softmax_dropout_fwd / _bwd == softmax_fwd -> dropout and dropout -> softmax_bwd with the same token, bit for bit
    (rel-pos self-attention of the encoder, causal decoder self-attention, source attention)."""
    from tavsr import ops
    torch.manual_seed(3)
    ops.manual_seed(77)
    H, B = 4, 3
    S = ops.pad4(T2)
    ac = torch.randn(H, B, T1, S, device="cuda")
    W = 2 * T1 - 1
    bd = torch.randn(H, B, T1, ops.pad4(W), device="cuda") if rel else None
    klens = torch.tensor([T2, T2 - 7, T2 // 2], device="cuda")
    scale = 0.125
    attn = ops.softmax_fwd(ac, bd, klens, scale, causal, T2=T2, W=W if rel else 0)
    pv, tok = ops.dropout(attn, 0.1)
    attn_f, pv_f, tok_f = ops.softmax_fwd(ac, bd, klens, scale, causal, T2=T2, W=W if rel else 0, p_drop=0.1, token=tok)
    # (the padding columns T2.. of a row are never read by the GEMMs: attn leaves them unwritten, the fused pv zeroes them)
    assert tok_f == tok and torch.equal(attn_f[..., :T2], attn[..., :T2]) and torch.equal(pv_f[..., :T2], pv[..., :T2])
    assert bool((pv_f[..., T2:] == 0).all())
    kept = float((pv[..., :T2] != 0).sum()) / float((attn[..., :T2] != 0).sum())
    assert 0.88 < kept < 0.92
    dpv = torch.randn_like(attn)
    want_ds, want_sk = ops.softmax_bwd(attn, ops.dropout(dpv, tok[0], token=tok)[0], scale, skew=rel, T2=T2)
    got_ds, got_sk = ops.softmax_bwd(attn, dpv, scale, skew=rel, T2=T2, token=tok)
    assert torch.equal(got_ds[..., :T2], want_ds[..., :T2])
    if rel:
        assert torch.equal(got_sk[..., :W], want_sk[..., :W])
    # a fresh site draws a different mask and advances the site counter by the tensor's size
    _, pv2, tok2 = ops.softmax_fwd(ac, bd, klens, scale, causal, T2=T2, W=W if rel else 0, p_drop=0.1)
    assert tok2[1] == tok[1] + attn.numel() and not torch.equal(pv2[..., :T2], pv[..., :T2])
