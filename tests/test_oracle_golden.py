"""CPU: the oracle restatement (oracle/) against golden vectors produced by the REFERENCE's own
modules (oracle/gen_golden.py).  This is what pins the oracle before it is used as the checker of
the HIP path."""
import numpy as np
import pytest
import torch

from helpers import TOKENS_EN, asr_conf, golden, max_rel, rel_err
from oracle import leaves as L
from oracle.model import (BranchformerEncoderOracle, CTCOracle, build_asr_oracle, compact,
                          fill_parameters_, synth)

TOL = 2e-5  # fp32 CPU vs fp32 CPU, different op order only


def test_ctc_direct_reference_vectors():
    g = golden("ctc_direct")
    B, T, D, V = int(g["B"]), int(g["T"]), int(g["D"]), int(g["V"])
    ctc = CTCOracle(odim=V, encoder_output_size=D, dropout_rate=0.0)
    fill_parameters_(ctc, seed=11)
    hs = synth((B, T, D), seed=12).requires_grad_(True)
    loss = ctc(hs, torch.from_numpy(g["hlens"]), torch.from_numpy(g["ys"]), torch.from_numpy(g["ys_lens"]))
    loss.backward()
    assert rel_err(loss, g["loss"]) < 1e-6
    assert rel_err(hs.grad, g["grad_hs"]) < 1e-5
    assert rel_err(ctc.ctc_lo.weight.grad, g["grad_w"]) < 1e-5
    assert np.array_equal(ctc.argmax(hs.detach()).numpy(), g["argmax"])


@pytest.mark.parametrize("tag,kw", [
    ("learned", dict(merge_method="learned_ave")),
    ("fixed", dict(merge_method="fixed_ave", cgmlp_weight=0.3)),
    ("fixed_attn_only", dict(merge_method="fixed_ave", cgmlp_weight=0.0)),
    ("fixed_mlp_only", dict(merge_method="fixed_ave", cgmlp_weight=1.0)),
    ("concat", dict(merge_method="concat")),
])
def test_layer_matches_reference(tag, kw):
    g = golden(f"bf_layer_{tag}")
    B, T, D = int(g["B"]), int(g["T"]), int(g["D"])
    enc = BranchformerEncoderOracle(input_size=D, num_blocks=1, input_layer=None, dropout_rate=0.0,
                                    positional_dropout_rate=0.0, attention_dropout_rate=0.0,
                                    ffn_activation_type="swish", **kw)
    layer = enc.encoders[0].train()
    assert sorted(layer.state_dict().keys()) == list(g["keys"])
    fill_parameters_(layer, seed=21)
    lens = torch.from_numpy(g["lens"])
    mask = (torch.arange(T)[None, :] < lens[:, None])[:, None, :]
    x = synth((B, T, D), seed=22).requires_grad_(True)
    xs, pos = L.RelPositionalEncoding(D, 0.0)(x)
    (y, _), _ = layer((xs, pos), mask)
    (y * synth((B, T, D), seed=23)).sum().backward()
    assert max_rel(y, g["y"]) < TOL
    assert rel_err(x.grad, g["grad_x"]) < 1e-4
    for n, p in layer.named_parameters():
        if "g_" + n in g.files:
            assert rel_err(compact(p.grad), g["g_" + n]) < 1e-4, n
    if tag == "learned":
        assert rel_err(layer.weight_global, g["weight_global"]) < 1e-5


@pytest.mark.parametrize("name", ["bf_encoder_6L_T49", "bf_encoder_2L_ragged", "bf_encoder_12L_T99", "bf_encoder_2L_T299",
                                  "bf_encoder_2L_T499"])
def test_encoder_matches_reference(name):
    g = golden(name)
    conf = asr_conf(num_blocks=int(g["nb"]))["encoder_conf"]
    enc = BranchformerEncoderOracle(input_size=80, **conf).eval()
    assert sum(p.numel() for p in enc.parameters()) == int(g["n_params"])
    fill_parameters_(enc, seed=31)
    x = synth((int(g["B"]), int(g["Tin"]), 80), seed=32)
    with torch.no_grad():
        y, olens, _ = enc(x, torch.from_numpy(g["lens"]))
    assert np.array_equal(olens.numpy(), g["olens"])
    assert max_rel(y, g["y"]) < TOL


def test_asr_model_matches_reference():
    g = golden("asr_model_3L")
    model = build_asr_oracle(asr_conf(num_blocks=3, dec_blocks=2), TOKENS_EN)
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    fill_parameters_(model, seed=41)
    B, Tin = int(g["B"]), int(g["Tin"])
    speech = synth((B, Tin, 80), seed=42)
    slens, tlens, text = (torch.from_numpy(g[k]) for k in ("slens", "tlens", "text"))
    model.eval()
    with torch.no_grad():
        loss, stats, w = model(speech.clone(), slens, text.clone(), tlens)
        enc, olens = model.encode(speech.clone(), slens)
        ids = model.ctc.argmax(enc)
    assert rel_err(loss, g["loss_eval"]) < 1e-5
    assert rel_err(stats["loss_ctc"], g["loss_ctc"]) < 1e-5
    assert rel_err(stats["loss_att"], g["loss_att"]) < 1e-5
    assert abs(float(stats["acc"]) - float(g["acc"])) < 1e-6
    assert abs(float(stats["cer_ctc"]) - float(g["cer_ctc"])) < 1e-6
    assert max_rel(enc, g["enc"]) < TOL
    assert np.array_equal(ids.numpy(), g["ctc_ids"])
    model.train()
    loss_t, _, _ = model(speech.clone(), slens, text.clone(), tlens)
    loss_t.backward()
    assert rel_err(loss_t, g["loss_train"]) < 1e-5
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("g_"):
            assert rel_err(compact(params[k[2:]].grad), g[k]) < 2e-4, k
    for n, v in zip(g["gnorm_keys"], g["gnorm_vals"]):
        assert abs(float(params[str(n)].grad.norm()) - v) <= 2e-4 * max(v, 1e-6) + 1e-7, n


def test_cfg1_wav_greedy():
    """BASELINE config 1 plumbing: synthetic 2 s WAV -> log-mel -> 6L encoder -> CTC greedy ids."""
    g = golden("cfg1_wav_greedy")
    conf = asr_conf(num_blocks=6, dec_blocks=1)
    conf["input_size"] = None
    model = build_asr_oracle(conf, TOKENS_EN).eval()
    fill_parameters_(model, seed=51)
    wav = 0.1 * synth((1, 32000), seed=52, kind="uniform")
    with torch.no_grad():
        feats, flens = model.frontend(wav, torch.tensor([32000]))
        hyp = model.ctc_greedy(wav, torch.tensor([32000]))
    assert int(flens[0]) == int(g["flens"][0]) == 201
    assert max_rel(feats, g["feats"]) < 1e-5
    assert hyp[0] == g["hyp"].tolist()
