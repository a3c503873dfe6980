"""Intermediate CTC with self-conditioning (encoder.py:378-401, tailored/encoder.py:270-318,
espnet_model.py:260-304): the oracle and the HIP product against vectors produced by the reference's own modules
(oracle/gen_golden.py:gen_interctc)."""
import argparse

import numpy as np
import pytest
import torch

from helpers import AVSR_CONV_YAML, AVSR_YAML, TOKENS_EN, relu_gated_tol, asr_conf, avsr_conf, golden, grad_ok, max_rel, rel_err
from oracle.av import build_avsr_oracle
from oracle.model import build_asr_oracle, compact, fill_parameters_, synth


def _asr_conf():
    conf = asr_conf(num_blocks=4, dec_blocks=1, interctc_layer_idx=[1, 3], interctc_use_conditioning=True)
    conf["model_conf"]["interctc_weight"] = 0.3
    return conf


def _asr_inputs(g):
    B, Tin = int(g["B"]), int(g["Tin"])
    speech = synth((B, Tin, 80), seed=302)
    slens, tlens, text = (torch.from_numpy(g[k]) for k in ("slens", "tlens", "text"))
    return speech, slens, text, tlens


def _check_asr(model, g, dev, tol_g):
    speech, slens, text, tlens = (t.to(dev) for t in _asr_inputs(g))
    model.train()
    loss_t, stats_t, _ = model(speech.clone(), slens, text.clone(), tlens)
    loss_t.backward()
    assert rel_err(loss_t.cpu(), g["loss_train"]) < 2e-5
    assert rel_err(stats_t["loss_ctc"].cpu(), g["loss_ctc_train"]) < 2e-5
    assert rel_err(stats_t["loss_interctc_layer1"].cpu(), g["loss_ic1"]) < 2e-5
    assert rel_err(stats_t["loss_interctc_layer3"].cpu(), g["loss_ic3"]) < 2e-5
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("g_"):
            assert grad_ok(compact(params[k[2:]].grad.cpu()), g[k], relu_gated_tol(k[2:], tol_g)), k
    for n, v in zip(g["gnorm_keys"], g["gnorm_vals"]):
        assert abs(float(params[str(n)].grad.norm()) - v) <= relu_gated_tol(str(n), tol_g) * max(v, 1e-6) + 1e-6, n
    model.eval()
    with torch.no_grad():
        loss_e, _, _ = model(speech.clone(), slens, text.clone(), tlens)
        enc, olens = model.encode(speech.clone(), slens)
    assert rel_err(loss_e.cpu(), g["loss_eval"]) < 2e-5
    assert max_rel(enc[0].cpu(), g["enc"]) < 1e-4
    assert max_rel(enc[1][0][1].cpu(), g["inter1"]) < 1e-4 and enc[1][0][0] == 1
    assert max_rel(enc[1][1][1].cpu(), g["inter3"]) < 1e-4 and enc[1][1][0] == 3
    assert np.array_equal(olens.cpu().numpy(), g["olens"])


def test_oracle_asr_interctc_matches_reference():
    g = golden("asr_model_interctc_4L")
    model = build_asr_oracle(_asr_conf(), TOKENS_EN)
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    fill_parameters_(model, seed=301)
    _check_asr(model, g, "cpu", 2e-4)


def _av_conf(avcond, yml=AVSR_YAML):
    conf = avsr_conf(yml, num_blocks=3, dec_blocks=1, interctc_layer_idx=[2], interctc_use_conditioning=True,
                     audiovisual_interctc_conditioning=avcond)
    conf["model_conf"]["interctc_weight"] = 0.25
    return conf


def _check_av(model, g, dev, tol_g):
    B, Ta, Tv = int(g["B"]), int(g["Ta"]), int(g["Tv"])
    audio, video = synth((B, Ta, 80), seed=312).to(dev), synth((B, Tv, 88, 88), seed=313).to(dev)
    alens, vlens, tlens, text = (torch.from_numpy(g[k]).to(dev) for k in ("alens", "vlens", "tlens", "text"))
    model.train()
    loss_t, stats_t, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
    loss_t.backward()
    assert rel_err(loss_t.cpu(), g["loss_train"]) < 2e-5
    assert rel_err(stats_t["loss_ctc"].cpu(), g["loss_ctc_train"]) < 2e-5
    assert rel_err(stats_t["loss_interctc_layer2"].cpu(), g["loss_ic2"]) < 2e-5
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("g_"):
            assert grad_ok(compact(params[k[2:]].grad.cpu()), g[k], relu_gated_tol(k[2:], tol_g)), k
    for n, v in zip(g["gnorm_keys"], g["gnorm_vals"]):
        assert abs(float(params[str(n)].grad.norm()) - v) <= relu_gated_tol(str(n), tol_g) * max(v, 1e-6) + 1e-6, n


# (conv_*: the conventional wrapper encoder's own intermediate-CTC block, conventional/encoder.py:154-199)
AV_CASES = [("av", True, AVSR_YAML), ("sep", False, AVSR_YAML), ("conv_av", True, AVSR_CONV_YAML), ("conv_sep", False, AVSR_CONV_YAML)]


@pytest.mark.parametrize("tag,avcond,yml", AV_CASES)
def test_oracle_avsr_interctc_matches_reference(tag, avcond, yml):
    g = golden(f"av_model_interctc_{tag}_3L")
    model = build_avsr_oracle(_av_conf(avcond, yml), TOKENS_EN)
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    fill_parameters_(model, seed=311)
    _check_av(model, g, "cpu", 5e-4)


@pytest.mark.gpu
def test_hip_asr_interctc_matches_reference():
    from tavsr.tasks.asr import ASRTask
    g = golden("asr_model_interctc_4L")
    conf = _asr_conf()
    conf["token_list"] = TOKENS_EN
    model = ASRTask.build_model(argparse.Namespace(**conf))
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    fill_parameters_(model, seed=301)
    _check_asr(model.cuda(), g, "cuda", 5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,avcond,yml", AV_CASES)
def test_hip_avsr_interctc_matches_reference(tag, avcond, yml):
    from tavsr.tasks.avsr import AVSRTask
    g = golden(f"av_model_interctc_{tag}_3L")
    conf = _av_conf(avcond, yml)
    conf["token_list"] = TOKENS_EN
    model = AVSRTask.build_model(argparse.Namespace(**conf))
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    fill_parameters_(model, seed=311)
    _check_av(model.cuda(), g, "cuda", 1e-3)
