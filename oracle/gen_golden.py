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
    }
    for name, c in cases.items():
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


def main():
    _shim.install()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen_ctc_direct()
    gen_layers()
    gen_encoders()
    gen_asr_model()
    gen_cfg1_wav()


if __name__ == "__main__":
    main()
