"""CPU: the audio-visual oracle restatement (oracle/av.py) against golden vectors produced by the REFERENCE's own
modules (oracle/gen_golden.py: gen_visual_frontend / gen_av_embed / gen_tailored / gen_avsr_models)."""
import numpy as np
import pytest
import torch

from helpers import AVSR_CONV_YAML, AVSR_YAML, TOKENS_EN, avsr_conf, golden, max_rel, rel_err
from oracle import leaves as L
from oracle.av import (AdaptiveFusionOracle, Conv3dResNet18Oracle, DefaultEmbeddingOracle, TailoredEncoderOracle,
                       build_avsr_oracle)
from oracle.model import compact, fill_parameters_, synth

TOL = 2e-5


def test_visual_frontend_matches_reference():
    g = golden("av_frontend")
    m = Conv3dResNet18Oracle()
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    fill_parameters_(m, seed=61)
    B, T = int(g["B"]), int(g["T"])
    x = synth((B, T, 88, 88), seed=62)
    m.train()
    y, _ = m(x, torch.tensor([5, 4]))
    (y * synth((B, T, 512), seed=63)).sum().backward()
    assert max_rel(y, g["y_train"]) < TOL
    params, bufs = dict(m.named_parameters()), dict(m.named_buffers())
    for k in g.files:
        if k.startswith("g_"):
            assert rel_err(compact(params[k[2:]].grad), g[k]) < 2e-4, k
    assert rel_err(bufs["frontend3D.1.running_mean"], g["rm_stem"]) < 1e-5
    assert rel_err(bufs["trunk.layer4.1.bn2.running_var"], g["rv_l4"]) < 1e-5
    assert int(bufs["frontend3D.1.num_batches_tracked"]) == int(g["nbt"])
    m.eval()
    with torch.no_grad():
        ye, _ = m(x, torch.tensor([5, 4]))
    assert max_rel(ye, g["y_eval"]) < TOL


@pytest.mark.parametrize("tag,kw,shape", [("audio", dict(input_size=80, input_layer="conv2d"), (3, 120, 80)),
                                          ("video", dict(input_size=512, input_layer="linear"), (3, 30, 512))])
def test_av_embedding_matches_reference(tag, kw, shape):
    g = golden(f"av_embed_{tag}")
    m = DefaultEmbeddingOracle(output_size=256, dropout_rate=0.0, positional_dropout_rate=0.0, **kw).train()
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    fill_parameters_(m, seed=91)
    x = synth(shape, seed=92).requires_grad_(True)
    y, masks = m.apply_embed_layer(x, torch.from_numpy(g["lens"]))
    ys, pos = m.apply_pos_enc(y)
    (ys * synth(tuple(ys.shape), seed=93)).sum().backward()
    assert max_rel(y, g["y"]) < TOL and max_rel(ys, g["ys"]) < TOL and max_rel(pos, g["pos"]) < 1e-6
    assert np.array_equal(masks.numpy(), g["masks"])
    assert rel_err(compact(x.grad), g["grad_x"]) < 1e-4
    for n, p in m.named_parameters():
        assert rel_err(compact(p.grad), g["g_" + n]) < 1e-4, n


def _masks(g, T):
    alens, vlens = torch.from_numpy(g["alens"]), torch.from_numpy(g["vlens"])
    am = (torch.arange(T)[None, :] < alens[:, None])[:, None, :]
    vm = (torch.arange(T)[None, :] < vlens[:, None])[:, None, :]
    return am, vm


@pytest.mark.parametrize("tag,ua,uv", [("aa", [True], [True]), ("ac", [True], [False]), ("ca", [False], [True]),
                                       ("cc", [False], [False])])
def test_tailored_layer_matches_reference(tag, ua, uv):
    g = golden(f"av_tailored_layer_{tag}")
    B, T, D = int(g["B"]), int(g["T"]), int(g["D"])
    enc = TailoredEncoderOracle("rel_pos", "latest", num_blocks=1, dropout_rate=0.0, positional_dropout_rate=0.0,
                                attention_dropout_rate=0.0, acoustic_use_attn=ua, visual_use_attn=uv).train()
    layer = enc.encoders[0]
    assert sorted(layer.state_dict().keys()) == list(g["keys"])
    fill_parameters_(layer, seed=71)
    am, vm = _masks(g, T)
    a = synth((B, T, D), seed=72).requires_grad_(True)
    v = synth((B, T, D), seed=73).requires_grad_(True)
    pe = L.RelPositionalEncoding(D, 0.0)
    xa, pos = pe(a)
    xv, _ = pe(v)
    (ya, _), _, (yv, _), _ = layer((xa, pos), am, (xv, pos), vm)
    ((ya * synth((B, T, D), seed=74)).sum() + (yv * synth((B, T, D), seed=75)).sum()).backward()
    assert max_rel(ya, g["ya"]) < TOL and max_rel(yv, g["yv"]) < TOL
    assert rel_err(a.grad, g["grad_a"]) < 1e-4 and rel_err(v.grad, g["grad_v"]) < 1e-4
    for n, p in layer.named_parameters():
        if "g_" + n in g.files:
            assert rel_err(compact(p.grad), g["g_" + n]) < 1e-4, n


@pytest.mark.parametrize("name", ["av_tailored_encoder_4L_fusion", "av_tailored_encoder_4L_fusion_T500"])
def test_tailored_encoder_and_fusion_match_reference(name):
    g = golden(name)
    B, T, D = int(g["B"]), int(g["T"]), int(g["D"])
    rows = slice(None, None, int(g["row_step"]) if "row_step" in g.files else 1)      # (the 20 s fixture keeps every 7th output row)
    enc = TailoredEncoderOracle("rel_pos", "latest", **avsr_conf(num_blocks=4)["encoder_conf"]).train()
    fusion = AdaptiveFusionOracle(input_size=256, **avsr_conf()["audiovisual_fusion_conf"]).train()
    assert sorted(enc.state_dict().keys()) == list(g["enc_keys"])
    assert sorted(fusion.state_dict().keys()) == list(g["fus_keys"])
    fill_parameters_(enc, seed=81)
    fill_parameters_(fusion, seed=82)
    am, vm = _masks(g, T)
    a = synth((B, T, D), seed=83).requires_grad_(True)
    v = synth((B, T, D), seed=84).requires_grad_(True)
    pe = L.RelPositionalEncoding(D, 0.0)
    xa, pos = pe(a)
    xv, _ = pe(v)
    ya, oam, yv, ovm, _ = enc((xa, pos), am, (xv, pos), vm)
    yf, olens = fusion(ya, oam, yv, ovm)
    (yf * synth((B, T, D), seed=85)).sum().backward()
    assert max_rel(ya[:, rows], g["ya"]) < TOL and max_rel(yv[:, rows], g["yv"]) < TOL and max_rel(yf[:, rows], g["yf"]) < TOL
    assert np.array_equal(olens.numpy(), g["olens"])
    assert rel_err(fusion.acoustic_weight, g["acoustic_weight"]) < 1e-5
    assert rel_err(a.grad[:, rows], g["grad_a"]) < 2e-4 and rel_err(v.grad[:, rows], g["grad_v"]) < 2e-4
    pe_, pf = dict(enc.named_parameters()), dict(fusion.named_parameters())
    for k in g.files:
        if k.startswith("g_enc."):
            assert rel_err(compact(pe_[k[6:]].grad), g[k]) < 2e-4, k
        if k.startswith("g_fus."):
            from helpers import grad_ok
            assert grad_ok(compact(pf[k[6:]].grad), g[k], 2e-4), k


@pytest.mark.parametrize("name,yaml_path,nb,seed", [("av_model_tailored_2L", AVSR_YAML, 2, 101),
                                                    ("av_model_conventional_1L", AVSR_CONV_YAML, 1, 111)])
def test_avsr_model_matches_reference(name, yaml_path, nb, seed):
    g = golden(name)
    model = build_avsr_oracle(avsr_conf(yaml_path, num_blocks=nb, dec_blocks=1), TOKENS_EN)
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    fill_parameters_(model, seed=seed)
    B, Ta, Tv = int(g["B"]), int(g["Ta"]), int(g["Tv"])
    audio, video = synth((B, Ta, 80), seed=seed + 1), synth((B, Tv, 88, 88), seed=seed + 2)
    alens, vlens, tlens, text = (torch.from_numpy(g[k]) for k in ("alens", "vlens", "tlens", "text"))
    model.train()
    loss_t, stats_t, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
    loss_t.backward()
    assert rel_err(loss_t, g["loss_train"]) < 1e-5
    assert rel_err(stats_t["loss_ctc"], g["loss_ctc_train"]) < 1e-5
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("g_"):
            assert rel_err(compact(params[k[2:]].grad), g[k]) < 5e-4, k
    for n, v in zip(g["gnorm_keys"], g["gnorm_vals"]):
        assert abs(float(params[str(n)].grad.norm()) - v) <= 5e-4 * max(v, 1e-6) + 1e-6, n
    model.eval()
    with torch.no_grad():
        loss, stats, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
        enc, olens = model.encode(audio.clone(), alens, video.clone(), vlens)
        ids = model.ctc.argmax(enc)
    assert rel_err(loss, g["loss_eval"]) < 1e-5
    assert abs(float(stats["acc"]) - float(g["acc"])) < 1e-6
    assert abs(float(stats["cer_ctc"]) - float(g["cer_ctc"])) < 1e-6
    assert max_rel(enc, g["enc"]) < TOL
    assert np.array_equal(olens.numpy(), g["olens"])
    assert np.array_equal(ids.numpy(), g["ctc_ids"])
