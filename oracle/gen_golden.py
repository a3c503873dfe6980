"""ORACLE tooling: generate ``tests/golden/*.npz`` from the REFERENCE's own modules.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden

The reference's Python is imported from where it lies over ``oracle._shim`` (leaf
stand-ins for the absent espnet package); ``src/ctc/ctc.py`` needs no stand-in beyond a
no-op ``typeguard`` so its vectors are *direct* reference outputs.  Weights and inputs
come from ``oracle.model.fill_parameters_`` / ``synth`` (numpy Philox keyed by parameter
name / seed) so a fixture stores only seeds, shapes and expected outputs.
"""
from __future__ import annotations

import argparse
import copy
import os

import numpy as np
import torch
import yaml

from . import _shim
from .model import compact, fill_parameters_, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TOKENS = "/root/reference/src/tokenizers/char/english.txt"
ASR_YAML = "/root/reference/configs/ASR/branchformer_transformer+ctc_english.yaml"


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def asr_conf(num_blocks=12, dropout=0.0, dec_blocks=6, **enc_over):
    conf = yaml.safe_load(open(ASR_YAML))
    conf["input_size"] = 80
    conf["specaug"] = None
    conf["encoder_conf"]["num_blocks"] = num_blocks
    conf["decoder_conf"]["num_blocks"] = dec_blocks
    for k in ("dropout_rate", "positional_dropout_rate", "attention_dropout_rate"):
        conf["encoder_conf"][k] = dropout
    for k in ("dropout_rate", "positional_dropout_rate", "self_attention_dropout_rate", "src_attention_dropout_rate"):
        conf["decoder_conf"][k] = dropout
    conf["ctc_conf"]["dropout_rate"] = dropout
    conf["encoder_conf"].update(enc_over)
    return conf


def gen_ctc_direct():
    """Direct reference vectors: src/ctc/ctc.py:133-188 (loss, grads, argmax)."""
    from src.ctc.ctc import CTC

    B, T, D, V, Lmax = 4, 25, 256, 41, 9
    ctc = CTC(odim=V, encoder_output_size=D, dropout_rate=0.0)
    fill_parameters_(ctc, seed=11)
    hs = synth((B, T, D), seed=12).requires_grad_(True)
    hlens = torch.tensor([25, 20, 25, 7])
    ys_lens = torch.tensor([9, 5, 1, 8])  # last one: 2L+1 > T on purpose -> inf -> zero_infinity
    ys = synth((B, Lmax), seed=13, kind="int", lo=1, hi=V - 1)
    ys[1, 2] = ys[1, 1]  # a repeated label (needs a blank in between)
    for i, l in enumerate(ys_lens):
        ys[i, l:] = -1
    loss = ctc(hs, hlens, ys, ys_lens)
    loss.backward()
    _save("ctc_direct", B=B, T=T, D=D, V=V, hlens=_np(hlens), ys=_np(ys), ys_lens=_np(ys_lens),
          loss=_np(loss), grad_hs=_np(hs.grad), grad_w=_np(ctc.ctc_lo.weight.grad),
          grad_b=_np(ctc.ctc_lo.bias.grad), argmax=_np(ctc.argmax(hs.detach())),
          logits=_np(ctc.ctc_lo(hs.detach())))


def _layer(merge_method, cgmlp_weight=0.5, use_attn=True, use_cgmlp=True):
    from src.encoder.branchformer.encoder import MyBranchformerEncoder

    enc = MyBranchformerEncoder(input_size=256, num_blocks=1, input_layer=None, dropout_rate=0.0,
                                positional_dropout_rate=0.0, attention_dropout_rate=0.0,
                                ffn_activation_type="swish", merge_method=merge_method,
                                cgmlp_weight=cgmlp_weight, use_attn=use_attn, use_cgmlp=use_cgmlp)
    return enc.encoders[0]


def gen_layers():
    """MyBranchformerEncoderLayer.forward (encoder_layer.py:153-321) for each merge mode."""
    from espnet.nets.pytorch_backend.transformer.embedding import RelPositionalEncoding

    B, T, D = 3, 23, 256
    lens = torch.tensor([23, 17, 9])
    mask = (torch.arange(T)[None, :] < lens[:, None])[:, None, :]
    pe = RelPositionalEncoding(D, 0.0)
    for tag, kw in {"learned": dict(merge_method="learned_ave"),
                    "fixed": dict(merge_method="fixed_ave", cgmlp_weight=0.3),
                    "fixed_attn_only": dict(merge_method="fixed_ave", cgmlp_weight=0.0),
                    "fixed_mlp_only": dict(merge_method="fixed_ave", cgmlp_weight=1.0),
                    "concat": dict(merge_method="concat")}.items():
        layer = _layer(**kw).train()
        fill_parameters_(layer, seed=21)
        x = synth((B, T, D), seed=22).requires_grad_(True)
        xs, pos = pe(x)
        (y, _), _ = layer((xs, pos), mask)
        r = synth((B, T, D), seed=23)
        (y * r).sum().backward()
        grads = {"g_" + n: compact(p.grad) for n, p in layer.named_parameters()
                 if n.endswith(("pos_bias_u", "linear_pos.weight", "conv.weight", "pooling_proj1.weight",
                                "weight_proj2.weight", "norm_mlp.weight", "merge_proj.bias",
                                "feed_forward_macaron.w_1.weight", "csgu.norm.bias", "linear_k.bias"))}
        extra = {}
        if kw["merge_method"] == "learned_ave":
            extra = dict(weight_global=_np(layer.weight_global), weight_local=_np(layer.weight_local))
        _save(f"bf_layer_{tag}", B=B, T=T, D=D, lens=_np(lens), y=_np(y), grad_x=_np(x.grad),
              keys=np.array(sorted(layer.state_dict().keys())), **grads, **extra)


def gen_encoders():
    """MyBranchformerEncoder.forward (encoder.py:324-412)."""
    from src.encoder.branchformer.encoder import MyBranchformerEncoder

    cases = {
        # config-1 shape: 2 s WAV -> 201 frames -> T=49, 6 blocks (ctor default)
        "bf_encoder_6L_T49": dict(nb=6, B=1, Tin=201, lens=[201]),
        "bf_encoder_2L_ragged": dict(nb=2, B=3, Tin=100, lens=[100, 77, 31]),
        "bf_encoder_12L_T99": dict(nb=12, B=2, Tin=400, lens=[400, 344]),
        # the reference's real length range (src/datasets/avsr_dataset.py:27: clips up to 20 s = 2000 mel frames, T = 499):
        # multi-block attention, merge tail beyond one batch, long cgMLP convolutions - wiring pinned by the reference itself
        "bf_encoder_2L_T299": dict(nb=2, B=2, Tin=1200, lens=[1200, 904]),
        "bf_encoder_2L_T499": dict(nb=2, B=1, Tin=2000, lens=[2000]),
    }
    only = os.environ.get("GEN_GOLDEN_ENCODERS")
    for name, c in cases.items():
        if only and name not in only.split(","):
            continue
        conf = asr_conf(num_blocks=c["nb"])["encoder_conf"]
        enc = MyBranchformerEncoder(input_size=80, **conf).eval()
        fill_parameters_(enc, seed=31)
        x = synth((c["B"], c["Tin"], 80), seed=32)
        lens = torch.tensor(c["lens"])
        with torch.no_grad():
            y, olens, _ = enc(x, lens)
        _save(name, nb=c["nb"], B=c["B"], Tin=c["Tin"], lens=_np(lens), y=_np(y), olens=_np(olens),
              n_params=sum(p.numel() for p in enc.parameters()))


def gen_asr_model():
    """ESPnetASRModel.forward (espnet_model.py:206-356) built by ASRTask.build_model
    (src/tasks/asr.py:482-619): loss/stats in eval and train mode (all dropout 0), grads."""
    from src.tasks.asr import ASRTask

    conf = asr_conf(num_blocks=3, dec_blocks=2)
    conf["token_list"] = TOKENS
    model = ASRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
    fill_parameters_(model, seed=41)
    B, Tin, Lmax = 3, 120, 12
    speech = synth((B, Tin, 80), seed=42)
    slens = torch.tensor([120, 96, 64])
    tlens = torch.tensor([12, 7, 10])
    text = synth((B, Lmax), seed=43, kind="int", lo=1, hi=39)
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    model.eval()
    with torch.no_grad():
        loss_e, stats_e, w = model(speech.clone(), slens, text.clone(), tlens)
        enc, olens = model.encode(speech.clone(), slens)
        ids = model.ctc.argmax(enc)
        logits = model.ctc.ctc_lo(enc)
        top2 = logits.topk(2, dim=-1).values
    model.train()
    loss_t, stats_t, _ = model(speech.clone(), slens, text.clone(), tlens)
    loss_t.backward()
    pick = ["encoder.embed.conv.0.weight", "encoder.embed.conv.2.bias", "encoder.embed.out.0.weight",
            "encoder.encoders.0.attn.pos_bias_v", "encoder.encoders.1.cgmlp.csgu.conv.bias",
            "encoder.encoders.2.feed_forward.w_2.weight", "encoder.after_norm.weight",
            "ctc.ctc_lo.weight", "decoder.embed.0.weight", "decoder.decoders.0.src_attn.linear_k.weight",
            "decoder.decoders.1.self_attn.linear_q.bias", "decoder.output_layer.bias",
            "decoder.after_norm.bias", "encoder.encoders.0.weight_proj1.weight"]
    params = dict(model.named_parameters())
    grads = {"g_" + n: compact(params[n].grad) for n in pick}
    gnorm = {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}
    _save("asr_model_3L", B=B, Tin=Tin, slens=_np(slens), tlens=_np(tlens), text=_np(text),
          loss_eval=_np(loss_e), loss_ctc=_np(stats_e["loss_ctc"]), loss_att=_np(stats_e["loss_att"]),
          acc=_np(stats_e["acc"]), cer_ctc=_np(stats_e["cer_ctc"]), cer=_np(stats_e["cer"]),
          loss_train=_np(loss_t), enc=_np(enc), olens=_np(olens), ctc_ids=_np(ids),
          top2_gap=_np(top2[..., 0] - top2[..., 1]),
          gnorm_keys=np.array(list(gnorm.keys())), gnorm_vals=np.array(list(gnorm.values()), dtype=np.float64),
          n_params=sum(p.numel() for p in model.parameters()),
          keys=np.array(sorted(model.state_dict().keys())), **grads)


def gen_cfg1_wav():
    """BASELINE config 1: 1 synthetic 2 s WAV -> DefaultFrontend -> 6L encoder -> CTC greedy
    (espnet_model.py:369-430 encode, ctc.py:180-188 argmax, maskctc_model.py:289-291 collapse)."""
    from itertools import groupby

    from src.tasks.asr import ASRTask

    conf = asr_conf(num_blocks=6, dec_blocks=1)
    conf["input_size"] = None
    conf["token_list"] = TOKENS
    model = ASRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval()
    fill_parameters_(model, seed=51)
    wav = 0.1 * synth((1, 32000), seed=52, kind="uniform")
    wlen = torch.tensor([32000])
    with torch.no_grad():
        feats, flens = model._extract_feats(wav, wlen)
        enc, olens = model.encode(wav, wlen)
        logits = model.ctc.ctc_lo(enc)
        ids = model.ctc.argmax(enc)
    hyp = [k for k, _ in groupby(ids[0].tolist()) if k != 0]
    top2 = logits.topk(2, dim=-1).values
    _save("cfg1_wav_greedy", feats=_np(feats), flens=_np(flens), enc=_np(enc), olens=_np(olens),
          ids=_np(ids), hyp=np.array(hyp, dtype=np.int64), top2_gap=_np(top2[..., 0] - top2[..., 1]))


AVSR_YAML = "/root/reference/configs/AVSR/tailored_transformer+ctc_english.yaml"
AVSR_CONV_YAML = "/root/reference/configs/AVSR/conventional_transformer+ctc_english.yaml"


def _zero_dropout(d):
    """every *dropout_rate key of a (nested) conf dict -> 0.0"""
    for k, v in d.items():
        if isinstance(v, dict):
            _zero_dropout(v)
        elif k.endswith("dropout_rate"):
            d[k] = 0.0


def avsr_conf(yaml_path=AVSR_YAML, num_blocks=12, dec_blocks=6, visual_input_size=None, **enc_over):
    conf = yaml.safe_load(open(yaml_path))
    conf["acoustic_input_size"] = 80
    conf["visual_input_size"] = visual_input_size
    conf["specaug"] = None
    _zero_dropout(conf)
    if conf["encoder"] == "tailored":
        conf["encoder_conf"]["num_blocks"] = num_blocks
        conf["encoder_conf"]["acoustic_use_attn"] = conf["encoder_conf"]["acoustic_use_attn"][:num_blocks]
        conf["encoder_conf"]["visual_use_attn"] = conf["encoder_conf"]["visual_use_attn"][:num_blocks]
    else:
        conf["encoder_conf"]["acoustic_encoder_conf"]["num_blocks"] = num_blocks
        conf["encoder_conf"]["visual_encoder_conf"]["num_blocks"] = num_blocks
    conf["decoder_conf"]["num_blocks"] = dec_blocks
    conf["encoder_conf"].update(enc_over)
    return conf


def gen_visual_frontend():
    """Conv3dResNet18.forward (conv3d_resnet18.py:77-97), train mode (batch statistics) with gradients, and eval
    mode (running statistics).  Only name-level stand-ins are involved (AbsFrontend, Swish): direct vectors."""
    from src.frontend.conv3d_resnet18.conv3d_resnet18 import Conv3dResNet18

    m = Conv3dResNet18(activation_type="swish")
    fill_parameters_(m, seed=61)
    B, T = 2, 5
    x = synth((B, T, 88, 88), seed=62)
    lens = torch.tensor([5, 4])
    m.train()
    y, _ = m(x, lens)
    r = synth((B, T, 512), seed=63)
    (y * r).sum().backward()
    params = dict(m.named_parameters())
    pick = ["frontend3D.0.weight", "frontend3D.1.weight", "frontend3D.1.bias", "trunk.layer1.0.conv1.weight",
            "trunk.layer1.1.bn2.weight", "trunk.layer2.0.downsample.0.weight", "trunk.layer2.0.downsample.1.bias",
            "trunk.layer3.1.conv2.weight", "trunk.layer4.0.conv1.weight", "trunk.layer4.1.bn2.bias"]
    grads = {"g_" + n: compact(params[n].grad) for n in pick}
    bufs = dict(m.named_buffers())
    stats = {"rm_stem": _np(bufs["frontend3D.1.running_mean"]), "rv_stem": _np(bufs["frontend3D.1.running_var"]),
             "rm_l4": _np(bufs["trunk.layer4.1.bn2.running_mean"]), "rv_l4": _np(bufs["trunk.layer4.1.bn2.running_var"]),
             "nbt": _np(bufs["frontend3D.1.num_batches_tracked"])}
    m.eval()
    with torch.no_grad():
        ye, _ = m(x, lens)
    _save("av_frontend", B=B, T=T, y_train=_np(y), y_eval=_np(ye), keys=np.array(sorted(m.state_dict().keys())),
          n_params=sum(p.numel() for p in m.parameters()), **grads, **stats)


def gen_tailored():
    """TailoredEncoderLayer / TailoredEncoder forward+backward (tailored/encoder_layer.py:118-274, encoder.py:221-332)
    and AdaptiveAudioVisualFusion (adaptive_audiovisual_fusion.py:113-211)."""
    from espnet.nets.pytorch_backend.transformer.embedding import RelPositionalEncoding
    from src.audiovisual_fusion.adaptive_audiovisual_fusion import AdaptiveAudioVisualFusion
    from src.encoder.audiovisual.tailored.encoder import TailoredEncoder

    B, T, D = 3, 21, 256
    alens, vlens = torch.tensor([21, 15, 8]), torch.tensor([21, 16, 8])
    am = (torch.arange(T)[None, :] < alens[:, None])[:, None, :]
    vm = (torch.arange(T)[None, :] < vlens[:, None])[:, None, :]
    pe = RelPositionalEncoding(D, 0.0)
    for tag, (ua, uv) in {"aa": ([True], [True]), "ac": ([True], [False]), "ca": ([False], [True]),
                          "cc": ([False], [False])}.items():
        enc = TailoredEncoder("rel_pos", "latest", num_blocks=1, dropout_rate=0.0, positional_dropout_rate=0.0,
                              attention_dropout_rate=0.0, acoustic_use_attn=ua, visual_use_attn=uv).train()
        layer = enc.encoders[0]
        fill_parameters_(layer, seed=71)
        a = synth((B, T, D), seed=72).requires_grad_(True)
        v = synth((B, T, D), seed=73).requires_grad_(True)
        (xa, pos), _ = pe(a), None
        xa, pos = pe(a)
        xv, _ = pe(v)
        (ya, _), _, (yv, _), _ = layer((xa, pos), am, (xv, pos), vm)
        ra, rv = synth((B, T, D), seed=74), synth((B, T, D), seed=75)
        ((ya * ra).sum() + (yv * rv).sum()).backward()
        params = dict(layer.named_parameters())
        grads = {"g_" + n: compact(p.grad) for n, p in params.items()
                 if n.endswith(("feed_forward_macaron.w_1.weight", "feed_forward.w_2.bias", "norm_final.weight",
                                "norm_ff_macaron.bias", "pos_bias_u", "linear_pos.weight", "csgu.conv.weight",
                                "acoustic_norm_mha.weight", "visual_norm_cgmlp.bias", "visual_norm_mha.bias",
                                "acoustic_norm_cgmlp.weight", "channel_proj2.weight", "linear_out.bias"))}
        _save(f"av_tailored_layer_{tag}", B=B, T=T, D=D, alens=_np(alens), vlens=_np(vlens), ya=_np(ya), yv=_np(yv),
              grad_a=_np(a.grad), grad_v=_np(v.grad), keys=np.array(sorted(layer.state_dict().keys())), **grads)

    _gen_tailored_encoder("av_tailored_encoder_4L_fusion", B, T, alens, vlens, 1)
    # the reference's real length range (500 lip frames = 20 s, src/datasets/avsr_dataset.py:27): every 7th output row is stored
    _gen_tailored_encoder("av_tailored_encoder_4L_fusion_T500", 2, 500, torch.tensor([500, 300]), torch.tensor([500, 304]), 7)


def _gen_tailored_encoder(name, B, T, alens, vlens, row_step):
    from espnet.nets.pytorch_backend.transformer.embedding import RelPositionalEncoding
    from src.audiovisual_fusion.adaptive_audiovisual_fusion import AdaptiveAudioVisualFusion
    from src.encoder.audiovisual.tailored.encoder import TailoredEncoder
    D = 256
    am = (torch.arange(T)[None, :] < alens[:, None])[:, None, :]
    vm = (torch.arange(T)[None, :] < vlens[:, None])[:, None, :]
    pe = RelPositionalEncoding(D, 0.0)
    conf = avsr_conf(num_blocks=4)["encoder_conf"]
    enc = TailoredEncoder("rel_pos", "latest", **conf).train()
    fusion = AdaptiveAudioVisualFusion(input_size=256, **avsr_conf()["audiovisual_fusion_conf"]).train()
    fill_parameters_(enc, seed=81)
    fill_parameters_(fusion, seed=82)
    a = synth((B, T, D), seed=83).requires_grad_(True)
    v = synth((B, T, D), seed=84).requires_grad_(True)
    xa, pos = pe(a)
    xv, _ = pe(v)
    ya, oam, yv, ovm, _ = enc((xa, pos), am, (xv, pos), vm)
    yf, olens = fusion(ya, oam, yv, ovm)
    r = synth((B, T, D), seed=85)
    (yf * r).sum().backward()
    pe_, pf = dict(enc.named_parameters()), dict(fusion.named_parameters())
    grads = {"g_enc." + n: compact(pe_[n].grad) for n in
             ["modality_encoding.weight", "encoders.0.feed_forward.w_1.weight", "encoders.3.norm_final.bias",
              "after_norm.weight", "encoders.1.acoustic_attn.linear_q.weight", "encoders.0.acoustic_cgmlp.csgu.conv.bias"]}
    grads.update({"g_fus." + n: compact(p.grad) for n, p in pf.items()})
    rows = slice(None, None, row_step)
    extra = {} if row_step == 1 else {"row_step": row_step}
    _save(name, B=B, T=T, D=D, **extra, alens=_np(alens), vlens=_np(vlens), ya=_np(ya[:, rows]), yv=_np(yv[:, rows]),
          yf=_np(yf[:, rows]), olens=_np(olens), grad_a=_np(a.grad[:, rows]), grad_v=_np(v.grad[:, rows]),
          acoustic_weight=_np(fusion.acoustic_weight), visual_weight=_np(fusion.visual_weight),
          enc_keys=np.array(sorted(enc.state_dict().keys())), fus_keys=np.array(sorted(fusion.state_dict().keys())), **grads)


def gen_av_embed():
    """DefaultEmbeddingLayerForAVSR (src/embedding_for_avsr/default.py:111-162), both input layers."""
    from src.embedding_for_avsr.default import DefaultEmbeddingLayerForAVSR

    B = 3
    for tag, kw, shape, lens in (("audio", dict(input_size=80, input_layer="conv2d"), (B, 120, 80), [120, 100, 57]),
                                 ("video", dict(input_size=512, input_layer="linear"), (B, 30, 512), [30, 25, 14])):
        m = DefaultEmbeddingLayerForAVSR(output_size=256, dropout_rate=0.0, positional_dropout_rate=0.0, **kw).train()
        fill_parameters_(m, seed=91)
        x = synth(shape, seed=92).requires_grad_(True)
        il = torch.tensor(lens)
        y, masks = m.apply_embed_layer(x, il)
        (ys, pos) = m.apply_pos_enc(y)
        r = synth(tuple(ys.shape), seed=93)
        (ys * r).sum().backward()
        grads = {"g_" + n: compact(p.grad) for n, p in m.named_parameters()}
        _save(f"av_embed_{tag}", lens=_np(il), y=_np(y), ys=_np(ys), pos=_np(pos), masks=_np(masks), grad_x=compact(x.grad),
              keys=np.array(sorted(m.state_dict().keys())), **grads)


def _gen_avsr_model(name, yaml_path, nb, seed):
    from src.tasks.avsr import AVSRTask

    conf = avsr_conf(yaml_path, num_blocks=nb, dec_blocks=1)
    conf["token_list"] = TOKENS
    model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
    fill_parameters_(model, seed=seed)
    B, Ta, Tv, Lmax = 2, 40, 9, 6
    audio = synth((B, Ta, 80), seed=seed + 1)
    video = synth((B, Tv, 88, 88), seed=seed + 2)
    alens, vlens = torch.tensor([40, 32]), torch.tensor([9, 8])
    tlens = torch.tensor([6, 4])
    text = synth((B, Lmax), seed=seed + 3, kind="int", lo=1, hi=39)
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    model.train()
    loss_t, stats_t, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
    loss_t.backward()
    params = dict(model.named_parameters())
    gnorm = {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}
    pick = [n for n in ("visual_frontend.frontend3D.0.weight", "visual_frontend.trunk.layer4.1.bn2.weight",
                        "acoustic_embed.embed.conv.0.weight", "visual_embed.embed.0.weight", "visual_embed.embed.1.weight",
                        "encoder.modality_encoding.weight", "audiovisual_fusion.audiovisual_layer.w_1.weight",
                        "audiovisual_fusion.acoustic_pooling_proj.weight", "ctc.ctc_lo.weight", "decoder.embed.0.weight")
            if n in params]
    grads = {"g_" + n: compact(params[n].grad) for n in pick}
    model.eval()
    with torch.no_grad():
        loss_e, stats_e, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
        enc, olens = model.encode(audio.clone(), alens, video.clone(), vlens)
        ids = model.ctc.argmax(enc)
        top2 = model.ctc.ctc_lo(enc).topk(2, dim=-1).values
    _save(name, B=B, Ta=Ta, Tv=Tv, alens=_np(alens), vlens=_np(vlens), tlens=_np(tlens), text=_np(text),
          loss_train=_np(loss_t), loss_ctc_train=_np(stats_t["loss_ctc"]), loss_att_train=_np(stats_t["loss_att"]),
          loss_eval=_np(loss_e), loss_ctc=_np(stats_e["loss_ctc"]), loss_att=_np(stats_e["loss_att"]),
          acc=_np(stats_e["acc"]), cer_ctc=_np(stats_e["cer_ctc"]), enc=_np(enc), olens=_np(olens), ctc_ids=_np(ids),
          top2_gap=_np(top2[..., 0] - top2[..., 1]),
          gnorm_keys=np.array(list(gnorm.keys())), gnorm_vals=np.array(list(gnorm.values()), dtype=np.float64),
          n_params=sum(p.numel() for p in model.parameters()), keys=np.array(sorted(model.state_dict().keys())), **grads)


def gen_avsr_models():
    """ESPnetAVSRModel.forward/encode (avsr_espnet_model.py:211-488) built by AVSRTask.build_model
    (src/tasks/avsr.py:506-718): tailored (2 blocks) and conventional (1 block) recipes, raw 88x88 lip frames in."""
    _gen_avsr_model("av_model_tailored_2L", AVSR_YAML, 2, 101)
    _gen_avsr_model("av_model_conventional_1L", AVSR_CONV_YAML, 1, 111)


def gen_interctc():
    """Intermediate CTC with self-conditioning: the reference's own encoder loops (encoder.py:378-401,
    tailored/encoder.py:270-318) and loss mix (espnet_model.py:260-304 == avsr_espnet_model.py:271-315)."""
    from src.tasks.asr import ASRTask
    from src.tasks.avsr import AVSRTask

    conf = asr_conf(num_blocks=4, dec_blocks=1, interctc_layer_idx=[1, 3], interctc_use_conditioning=True)
    conf["model_conf"]["interctc_weight"] = 0.3
    conf["token_list"] = TOKENS
    model = ASRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
    fill_parameters_(model, seed=301)
    B, Tin, Lmax = 3, 120, 10
    speech = synth((B, Tin, 80), seed=302)
    slens, tlens = torch.tensor([120, 96, 72]), torch.tensor([10, 6, 8])
    text = synth((B, Lmax), seed=303, kind="int", lo=1, hi=39)
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    model.train()
    loss_t, stats_t, _ = model(speech.clone(), slens, text.clone(), tlens)
    loss_t.backward()
    params = dict(model.named_parameters())
    gnorm = {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}
    pick = ["encoder.conditioning_layer.weight", "encoder.conditioning_layer.bias", "ctc.ctc_lo.weight", "ctc.ctc_lo.bias",
            "encoder.encoders.0.feed_forward.w_1.weight", "encoder.encoders.3.attn.linear_q.weight", "encoder.after_norm.weight"]
    grads = {"g_" + n: compact(params[n].grad) for n in pick}
    model.eval()
    with torch.no_grad():
        loss_e, stats_e, _ = model(speech.clone(), slens, text.clone(), tlens)
        enc, olens = model.encode(speech.clone(), slens)
    _save("asr_model_interctc_4L", B=B, Tin=Tin, slens=_np(slens), tlens=_np(tlens), text=_np(text),
          loss_train=_np(loss_t), loss_ctc_train=_np(stats_t["loss_ctc"]), loss_ic1=_np(stats_t["loss_interctc_layer1"]),
          loss_ic3=_np(stats_t["loss_interctc_layer3"]), loss_eval=_np(loss_e), enc=_np(enc[0]), inter1=_np(enc[1][0][1]),
          inter3=_np(enc[1][1][1]), olens=_np(olens),
          gnorm_keys=np.array(list(gnorm.keys())), gnorm_vals=np.array(list(gnorm.values()), dtype=np.float64),
          keys=np.array(sorted(model.state_dict().keys())), **grads)

    # (conv_*: the wrapper encoder's own intermediate-CTC block, conventional/encoder.py:154-199)
    for tag, avcond, yml in (("av", True, AVSR_YAML), ("sep", False, AVSR_YAML), ("conv_av", True, AVSR_CONV_YAML),
                             ("conv_sep", False, AVSR_CONV_YAML)):
        conf = avsr_conf(yml, num_blocks=3, dec_blocks=1, interctc_layer_idx=[2], interctc_use_conditioning=True,
                         audiovisual_interctc_conditioning=avcond)
        conf["model_conf"]["interctc_weight"] = 0.25
        conf["token_list"] = TOKENS
        model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
        fill_parameters_(model, seed=311)
        B, Ta, Tv, Lmax = 2, 40, 9, 6
        audio, video = synth((B, Ta, 80), seed=312), synth((B, Tv, 88, 88), seed=313)
        alens, vlens, tlens = torch.tensor([40, 32]), torch.tensor([9, 8]), torch.tensor([6, 4])
        text = synth((B, Lmax), seed=314, kind="int", lo=1, hi=39)
        for i, l in enumerate(tlens):
            text[i, l:] = -1
        model.train()
        loss_t, stats_t, _ = model(audio.clone(), alens, video.clone(), vlens, text.clone(), tlens)
        loss_t.backward()
        params = dict(model.named_parameters())
        gnorm = {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}
        pick = ["encoder.conditioning_layer.weight", "ctc.ctc_lo.weight",
                "encoder.modality_encoding.weight" if yml == AVSR_YAML else "encoder.visual_encoder.after_norm.weight",
                "audiovisual_fusion.audiovisual_layer.w_1.weight"]
        grads = {"g_" + n: compact(params[n].grad) for n in pick}
        _save(f"av_model_interctc_{tag}_3L", B=B, Ta=Ta, Tv=Tv, alens=_np(alens), vlens=_np(vlens), tlens=_np(tlens), text=_np(text),
              loss_train=_np(loss_t), loss_ctc_train=_np(stats_t["loss_ctc"]), loss_ic2=_np(stats_t["loss_interctc_layer2"]),
              gnorm_keys=np.array(list(gnorm.keys())), gnorm_vals=np.array(list(gnorm.values()), dtype=np.float64),
              keys=np.array(sorted(model.state_dict().keys())), **grads)


def gen_noam_adam():
    """src/schedulers/noam.py (imported as it is: pure torch): Noam rates and the parameters after 12 Adam steps on
    seeded gradients, incl. the reference loop's accumulate-then-step cadence (avsr_main.py:36-54)."""
    from src.schedulers.noam import get_noam_scheduler

    shapes = [(37, 19), (256,), (5, 3, 7), (1,)]
    params = [torch.nn.Parameter(synth(s, seed=121 + i)) for i, s in enumerate(shapes)]
    opt = get_noam_scheduler(params, 1.6, 256, 5)
    rates = []
    for step in range(12):
        opt.zero_grad()
        for micro in range(3):                       # three accumulated micro-batches per optimizer step
            for i, p in enumerate(params):
                g = synth(tuple(p.shape), seed=1000 + 100 * step + 10 * micro + i) / 3
                p.grad = g if p.grad is None else p.grad + g
        opt.step()
        rates.append(opt._rate)
    _save("noam_adam", rates=np.array(rates, dtype=np.float64),
          **{f"p{i}": _np(p) for i, p in enumerate(params)})


def main():
    import sys

    _shim.install()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--optim-only" in sys.argv:
        gen_noam_adam()
        return
    if "--cfg1-only" in sys.argv:
        gen_cfg1_wav()
        return
    if "--encoders-only" in sys.argv:
        gen_encoders()
        return
    if "--tailored-only" in sys.argv:
        gen_tailored()
        return
    if "--interctc-only" in sys.argv:
        gen_interctc()
        return
    if "--av-only" in sys.argv:
        gen_visual_frontend()
        gen_av_embed()
        gen_tailored()
        gen_avsr_models()
        return
    gen_ctc_direct()
    gen_layers()
    gen_encoders()
    gen_asr_model()
    gen_cfg1_wav()
    gen_visual_frontend()
    gen_av_embed()
    gen_tailored()
    gen_avsr_models()
    gen_interctc()
    gen_noam_adam()


if __name__ == "__main__":
    main()
