"""GPU: the video-only recipes (configs/VSR/*.yaml: ``task: vsr``, ``frontend: conv3dresnet18``, Branchformer encoder with
``input_layer: linear``; the *_tailored variant adds ``merge_method: fixed_ave`` with a per-layer ``cgmlp_weight`` list)
built by ASRTask.build_model from the recipe YAML and compared with the oracle built from the same dict: state_dict keys,
encoder output, hybrid loss and every parameter gradient."""
import argparse
import copy
import os

import pytest
import torch
import yaml

from helpers import ROOT, TOKENS_EN, grad_ok, max_rel, rel_err

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, "tailored-avsr_amd", "configs")


def _conf(name, num_blocks, dec_blocks):
    conf = yaml.safe_load(open(os.path.join(CFG, name)))

    def zero(d):
        for k, v in d.items():
            if isinstance(v, dict):
                zero(v)
            elif k.endswith("dropout_rate"):
                d[k] = 0.0
    zero(conf)
    enc = conf["encoder_conf"]
    if isinstance(enc.get("cgmlp_weight"), list):            # keep the recipe's pattern: the cgMLP-only layer comes second
        enc["cgmlp_weight"] = ([0.0, 1.0, 0.0, 0.5] * num_blocks)[:num_blocks]
    enc["num_blocks"] = num_blocks
    conf["decoder_conf"]["num_blocks"] = dec_blocks
    conf["token_list"] = list(TOKENS_EN)
    return conf


@pytest.mark.parametrize("name,blocks", [
    ("vsr_conv3dresnet18_branchformer_transformer_ctc_english.yaml", 2),
    ("vsr_conv3dresnet18_branchformer_transformer_ctc_english_tailored.yaml", 4),
])
def test_vsr_recipe_matches_oracle(name, blocks):
    from oracle.model import build_asr_oracle, fill_parameters_, synth
    from tavsr.tasks.asr import ASRTask
    conf = _conf(name, blocks, 1)
    assert conf["task"] == "vsr" and conf["frontend"] == "conv3dresnet18" and conf["encoder_conf"]["input_layer"] == "linear"
    oracle = build_asr_oracle(copy.deepcopy(conf), TOKENS_EN)
    fill_parameters_(oracle, seed=77)
    model = ASRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
    assert sorted(model.state_dict().keys()) == sorted(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    model = model.cuda()
    B, T = 3, 9
    video = synth((B, T, 88, 88), seed=5)
    vlens = torch.tensor([9, 7, 4])
    text = synth((B, 6), seed=6, kind="int", lo=1, hi=40)
    tlens = torch.tensor([6, 4, 2])
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    oracle.eval()
    model.eval()
    with torch.no_grad():
        eo, lo_ = oracle.encode(video, vlens)
        eg, lg_ = model.encode(video.cuda(), vlens.cuda())
    assert torch.equal(lg_.cpu(), lo_)
    assert max_rel(eg.cpu(), eo) < 2e-4
    oracle.train()
    model.train()
    lo, so, _ = oracle(video, vlens, text, tlens)
    lo.backward()
    lg, sg, _ = model(video.cuda(), vlens.cuda(), text.cuda(), tlens.cuda())
    lg.backward()
    assert rel_err(lg.detach().cpu(), lo.detach()) < 1e-4
    assert rel_err(sg["loss_ctc"].cpu(), so["loss_ctc"]) < 1e-4 and rel_err(sg["loss_att"].cpu(), so["loss_att"]) < 1e-4
    po = dict(oracle.named_parameters())
    for n, p in model.named_parameters():
        if po[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        assert grad_ok(p.grad.cpu(), po[n].grad, 5e-3), n
