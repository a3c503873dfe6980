"""GPU parity tests proper: the HIP product path (tavsr modules -> C ABI -> gfx950 kernels) against
(1) the committed golden vectors generated from the reference's own modules and (2) the pinned
oracle on the same seeded inputs.  Tolerances: fp32 activations 1e-4 relative (max |err| / max |ref|,
BASELINE.json north_star); integer ids bit-exact; gradients 1e-3 relative L2 (fp32 accumulation order)."""
import argparse

import numpy as np
import pytest
import torch

from helpers import relu_gated_tol, TOKENS_EN, asr_conf, golden, grad_ok, max_rel, rel_err

pytestmark = pytest.mark.gpu

ACT_TOL = 1e-4
GRAD_TOL = 1e-3


def _fill(module, seed):
    from oracle.model import fill_parameters_
    fill_parameters_(module, seed=seed)  # runs on CPU tensors, before .cuda()
    return module.cuda()


@pytest.mark.parametrize("tag,kw", [
    ("learned", dict(merge_method="learned_ave")),
    ("fixed", dict(merge_method="fixed_ave", cgmlp_weight=0.3)),
    ("fixed_attn_only", dict(merge_method="fixed_ave", cgmlp_weight=0.0)),
    ("fixed_mlp_only", dict(merge_method="fixed_ave", cgmlp_weight=1.0)),
    ("concat", dict(merge_method="concat")),
])
def test_layer_vs_reference_golden(tag, kw):
    from oracle.model import compact, synth
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    from tavsr.layers import RelPositionalEncoding
    g = golden(f"bf_layer_{tag}")
    B, T, D = int(g["B"]), int(g["T"]), int(g["D"])
    enc = MyBranchformerEncoder(input_size=D, num_blocks=1, input_layer=None, dropout_rate=0.0,
                                positional_dropout_rate=0.0, attention_dropout_rate=0.0,
                                ffn_activation_type="swish", **kw)
    layer = enc.encoders[0]
    assert sorted(layer.state_dict().keys()) == list(g["keys"])
    layer = _fill(layer, 21).train()
    lens = torch.from_numpy(g["lens"]).cuda()
    mask = (torch.arange(T, device="cuda")[None, :] < lens[:, None])[:, None, :]
    x = synth((B, T, D), seed=22).cuda().requires_grad_(True)
    xs, pos = RelPositionalEncoding(D, 0.0)(x.detach())
    xs.requires_grad_(True)
    (y, _), _ = layer((xs, pos), mask)
    (y * synth((B, T, D), seed=23).cuda()).sum().backward()
    assert max_rel(y.cpu(), g["y"]) < ACT_TOL
    assert rel_err(xs.grad.cpu() * 16.0, g["grad_x"]) < GRAD_TOL  # golden grad is w.r.t. the unscaled input
    for n, p in layer.named_parameters():
        if "g_" + n in g.files:
            assert grad_ok(compact(p.grad.cpu()), g["g_" + n], GRAD_TOL), n
    if tag == "learned":
        assert rel_err(layer.weight_global.cpu(), g["weight_global"]) < 1e-4


@pytest.mark.parametrize("name", ["bf_encoder_6L_T49", "bf_encoder_2L_ragged", "bf_encoder_12L_T99", "bf_encoder_2L_T299",
                                  "bf_encoder_2L_T499"])
def test_encoder_vs_reference_golden(name):
    from oracle.model import synth
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    g = golden(name)
    conf = asr_conf(num_blocks=int(g["nb"]))["encoder_conf"]
    enc = MyBranchformerEncoder(input_size=80, **conf)
    assert sum(p.numel() for p in enc.parameters()) == int(g["n_params"])
    enc = _fill(enc, 31).eval()
    x = synth((int(g["B"]), int(g["Tin"]), 80), seed=32).cuda()
    with torch.no_grad():
        y, olens, _ = enc(x, torch.from_numpy(g["lens"]).cuda())
    assert np.array_equal(olens.cpu().numpy(), g["olens"])
    assert max_rel(y.cpu(), g["y"]) < ACT_TOL


def test_asr_model_vs_reference_golden():
    from oracle.model import compact, synth
    from tavsr.tasks.asr import ASRTask
    g = golden("asr_model_3L")
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=3, dec_blocks=2)))
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    model = _fill(model, 41)
    B, Tin = int(g["B"]), int(g["Tin"])
    speech = synth((B, Tin, 80), seed=42).cuda()
    slens, tlens, text = (torch.from_numpy(g[k]).cuda() for k in ("slens", "tlens", "text"))
    model.eval()
    with torch.no_grad():
        loss, stats, w = model(speech.clone(), slens, text.clone(), tlens)
        enc, olens = model.encode(speech.clone(), slens)
        ids, hyp, hl = model.ctc.greedy(enc, olens)
    assert max_rel(enc.cpu(), g["enc"]) < ACT_TOL
    assert rel_err(stats["loss_ctc"].cpu(), g["loss_ctc"]) < 1e-4
    assert rel_err(stats["loss_att"].cpu(), g["loss_att"]) < 1e-4
    assert rel_err(loss.cpu(), g["loss_eval"]) < 1e-4
    assert abs(float(stats["acc"]) - float(g["acc"][0])) < 1e-6
    assert abs(float(stats["cer_ctc"]) - float(g["cer_ctc"][0])) < 1e-6
    # CTC greedy ids: bit-exact wherever the reference's own top-2 logit gap is binding (> 1e-4)
    binding = torch.from_numpy(g["top2_gap"] > 1e-4)
    assert binding.float().mean() > 0.99
    assert torch.equal(ids.cpu()[binding], torch.from_numpy(g["ctc_ids"])[binding])
    model.train()
    loss_t, _, _ = model(speech.clone(), slens, text.clone(), tlens)
    loss_t.backward()
    assert rel_err(loss_t.detach().cpu(), g["loss_train"]) < 1e-4
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("g_"):
            assert grad_ok(compact(params[k[2:]].grad.cpu()), g[k], relu_gated_tol(k[2:], GRAD_TOL)), k
    for n, v in zip(g["gnorm_keys"], g["gnorm_vals"]):
        got = float(params[str(n)].grad.norm())
        assert abs(got - v) <= relu_gated_tol(str(n), GRAD_TOL) * max(v, 1e-6) + 1e-6, (n, got, v)


def test_full_size_vs_oracle_cfg2():
    """BASELINE configs[1] at full size (12L, B=32 x 4 s, ragged): HIP vs the pinned oracle run on the host."""
    from oracle.model import build_asr_oracle, fill_parameters_, synth
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=12, dec_blocks=6)
    oracle = build_asr_oracle(conf, TOKENS_EN)
    fill_parameters_(oracle, seed=1234)
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=12, dec_blocks=6)))
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    oracle.train()
    B = 32
    speech = synth((B, 400, 80), seed=1234)
    slens = torch.tensor([400 - 4 * (i % 20) for i in range(B)])
    text = synth((B, 40), seed=1235, kind="int", lo=1, hi=40)
    tlens = torch.tensor([40 - (i % 31) for i in range(B)])
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    torch.set_num_threads(max(1, torch.get_num_threads()))
    lo, so, _ = oracle(speech, slens, text, tlens)
    lo.backward()
    lg, sg, _ = model(speech.cuda(), slens.cuda(), text.cuda(), tlens.cuda())
    lg.backward()
    assert rel_err(lg.detach().cpu(), lo.detach()) < 1e-4
    with torch.no_grad():
        eo, _ = oracle.encode(speech, slens)
        eg, ol = model.encode(speech.cuda(), slens.cuda())
    assert max_rel(eg.cpu(), eo) < ACT_TOL
    # greedy ids vs oracle: bit-exact where the oracle's top-2 gap is binding
    logits = oracle.ctc.ctc_lo(eo)
    top2 = logits.topk(2, -1).values
    binding = (top2[..., 0] - top2[..., 1]) > 1e-4
    ids, hyp, hl = model.ctc.greedy(eg, ol)
    assert torch.equal(ids.cpu()[binding], logits.argmax(-1)[binding])
    po = dict(oracle.named_parameters())
    worst = 0.0
    for n, p in model.named_parameters():
        assert grad_ok(p.grad.cpu(), po[n].grad, 5e-3), n
        if float(po[n].grad.abs().max()) > 1e-6:
            worst = max(worst, rel_err(p.grad.cpu(), po[n].grad))
    print("worst grad rel err", worst)


def test_full_size_batch_permutation_invariance():
    """Size-independent property at BASELINE configs[1] size (12L, B=32 x 4 s, ragged): every utterance's encoder output
    and greedy hypothesis are bit-identical wherever it sits in the batch - each output row of every kernel is a
    function of its own utterance only, accumulated in a fixed order.  (Invariance to extra padding does NOT hold in the
    reference itself: the cgMLP's depthwise convolution reads the padded frames, and the subsampled lengths depend on
    the padded length; neither is asserted.)"""
    from oracle.model import fill_parameters_, synth
    from tavsr.tasks.asr import ASRTask
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=12, dec_blocks=6)))
    fill_parameters_(model, seed=77)
    model = model.cuda().eval()
    B = 32
    speech = synth((B, 400, 80), seed=5).cuda()
    slens = torch.tensor([400 - 8 * (i % 13) for i in range(B)]).cuda()
    for i in range(B):
        speech[i, int(slens[i]):] = 0
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        e0, l0 = model.encode(speech, slens)
        ids0, hyp0, hl0 = model.ctc.greedy(e0, l0)
        e1, l1 = model.encode(speech[perm].contiguous(), slens[perm].contiguous())
        ids1, hyp1, hl1 = model.ctc.greedy(e1, l1)
    assert torch.equal(l0[perm], l1)
    for j in range(B):
        i = int(perm[j])
        n = int(l0[i])
        assert torch.equal(e0[i, :n], e1[j, :n]), (i, j)
        assert torch.equal(hyp0[i, : int(hl0[i])], hyp1[j, : int(hl1[j])])


def test_full_size_training_step_is_bitwise_reproducible():
    """BASELINE configs[1] size, train mode with the recipe's dropout: two fwd+bwd runs from the same generator state give
    bit-identical loss and gradients (fixed-order two-stage reductions, no float atomics, also with the two branches
    on two streams)."""
    from oracle.model import fill_parameters_, synth
    from tavsr import ops
    from tavsr.tasks.asr import ASRTask
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=12, dec_blocks=6, dropout=0.1)))
    fill_parameters_(model, seed=3)
    model = model.cuda().train()
    B = 32
    speech = synth((B, 400, 80), seed=6).cuda()
    slens = torch.tensor([400 - 4 * (i % 20) for i in range(B)]).cuda()
    text = synth((B, 40), seed=7, kind="int", lo=1, hi=40).cuda()
    tlens = torch.tensor([40 - (i % 31) for i in range(B)]).cuda()
    for i, l in enumerate(tlens):
        text[i, int(l):] = -1
    runs = []
    for _ in range(2):
        ops.manual_seed(11)
        for p in model.parameters():
            p.grad = None
        loss = model(speech, slens, text, tlens)[0]
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0])
    for (n, _), a, b in zip(model.named_parameters(), runs[0][1], runs[1][1]):
        assert torch.equal(a, b), n


def test_ctc_direct_reference_vectors_on_hip():
    """tavsr.ctc.CTC on the GPU against the vectors the reference's own src/ctc/ctc.py produced (no stand-in involved):
    loss with a zero-infinity utterance and a repeated label, gradients through ctc_lo, argmax ids, frame posteriors."""
    from oracle.model import fill_parameters_, synth
    from tavsr.ctc.ctc import CTC
    g = golden("ctc_direct")
    B, T, D, V = int(g["B"]), int(g["T"]), int(g["D"]), int(g["V"])
    ctc = CTC(odim=V, encoder_output_size=D, dropout_rate=0.0)
    fill_parameters_(ctc, seed=11)
    ctc = ctc.cuda()
    hs = synth((B, T, D), seed=12).cuda().requires_grad_(True)
    loss = ctc(hs, torch.from_numpy(g["hlens"]).cuda(), torch.from_numpy(g["ys"]).cuda(), torch.from_numpy(g["ys_lens"]).cuda())
    loss.backward()
    assert rel_err(loss.cpu(), g["loss"]) < 1e-5
    assert rel_err(hs.grad.cpu(), g["grad_hs"]) < 1e-4
    assert rel_err(ctc.ctc_lo.weight.grad.cpu(), g["grad_w"]) < 1e-4
    assert rel_err(ctc.ctc_lo.bias.grad.cpu(), g["grad_b"]) < 1e-4
    assert float(hs.grad[3].abs().max()) == 0.0          # 2L+1 > T: infinite loss -> zero_infinity
    assert np.array_equal(ctc.argmax(hs.detach()).cpu().numpy(), g["argmax"])
    # src/ctc/ctc.py:160-178 on the reference's own logits
    logits = torch.from_numpy(g["logits"]).double()
    assert max_rel(ctc.log_softmax(hs.detach()).cpu(), logits.log_softmax(2)) < 1e-5
    assert max_rel(ctc.softmax(hs.detach()).cpu(), logits.softmax(2)) < 1e-5


def test_overpadded_batch_is_cut_like_the_reference():
    """a batch padded beyond its longest utterance (fixed-size DP batches, external collate): the reference cuts speech
    and text to lengths.max() before anything else (espnet_model.py:236,438): same loss / encoder output as the tight batch."""
    from oracle.model import synth
    from tavsr.tasks.asr import ASRTask
    g = golden("asr_model_3L")
    model = _fill(ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=3, dec_blocks=2))), 41).eval()
    B, Tin = int(g["B"]), int(g["Tin"])
    speech = synth((B, Tin, 80), seed=42)
    slens, tlens, text = (torch.from_numpy(g[k]) for k in ("slens", "tlens", "text"))
    pad_s = torch.cat([speech, torch.full((B, 37, 80), 3.0)], dim=1).cuda()         # junk past every length
    pad_t = torch.cat([text, torch.full((B, 5), -1, dtype=text.dtype)], dim=1).cuda()
    with torch.no_grad():
        loss, stats, _ = model(pad_s, slens.cuda(), pad_t, tlens.cuda())
        enc, _ = model.encode(pad_s, slens.cuda())
    assert enc.shape[1] == g["enc"].shape[1]
    assert max_rel(enc.cpu(), g["enc"]) < ACT_TOL
    assert rel_err(loss.cpu(), g["loss_eval"]) < 1e-4


def test_attention_branch_drop_matches_the_oracle():
    """attn_branch_drop_rate (src/encoder/branchformer/encoder_layer.py:233-240): with rate 1.0 every training step merges
    with the constant weights (0, 1) - output, input gradient and parameter gradients against the oracle's same branch;
    the pooling / weight projections take no part (no gradient), the dropped branch gets zero gradients."""
    from oracle.model import BranchformerEncoderOracle, fill_parameters_, synth
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    from tavsr.layers import RelPositionalEncoding
    from oracle import leaves as OL
    kw = dict(input_size=256, num_blocks=1, input_layer=None, dropout_rate=0.0, positional_dropout_rate=0.0,
              attention_dropout_rate=0.0, ffn_activation_type="swish", merge_method="learned_ave", attn_branch_drop_rate=1.0)
    ol = BranchformerEncoderOracle(**kw).encoders[0].train()
    fill_parameters_(ol, seed=21)
    pl = MyBranchformerEncoder(**kw).encoders[0]
    pl.load_state_dict(ol.state_dict())
    pl = pl.cuda().train()
    B, T, D = 3, 23, 256
    lens = torch.tensor([23, 17, 9])
    mask = (torch.arange(T)[None, :] < lens[:, None])[:, None, :]
    x = synth((B, T, D), seed=22)
    r = synth((B, T, D), seed=23)
    xo, pos = OL.RelPositionalEncoding(D, 0.0)(x)
    xo = xo.detach().requires_grad_(True)
    (yo, _), _ = ol((xo, pos), mask)
    (yo * r).sum().backward()
    xg, posg = RelPositionalEncoding(D, 0.0)(x.cuda())
    xg = xg.detach().requires_grad_(True)
    (yg, _), _ = pl((xg, posg), mask.cuda())
    (yg * r.cuda()).sum().backward()
    assert pl.weight_global == 0.0 and pl.weight_local == 1.0
    assert max_rel(yg.detach().cpu(), yo.detach()) < ACT_TOL
    assert rel_err(xg.grad.cpu(), xo.grad) < GRAD_TOL
    po = dict(ol.named_parameters())
    for n, p in pl.named_parameters():
        if po[n].grad is None:
            assert p.grad is None, n                       # pooling_proj* / weight_proj*: not part of the step
        elif n.startswith("attn.") or n.startswith("norm_mha."):
            assert float(p.grad.abs().max()) == 0.0 and float(po[n].grad.abs().max()) == 0.0, n
        else:
            assert grad_ok(p.grad.cpu(), po[n].grad, GRAD_TOL), n
