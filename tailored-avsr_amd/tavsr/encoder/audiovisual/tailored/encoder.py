"""``TailoredEncoder`` - drop-in for src/encoder/audiovisual/tailored/encoder.py:36-332 (rel_pos "latest" subset the
shipped configs use): modality encoding -> N x TailoredEncoderLayer -> shared after_norm on both streams."""
from __future__ import annotations

from typing import List

import torch

from .... import functional as F_
from .... import functional_av as FA
from ....layers import ConvolutionalGatingMLP, LayerNorm, PositionwiseFeedForward, RelPositionMultiHeadedAttention
from .encoder_layer import TailoredEncoderLayer


class TailoredEncoder(torch.nn.Module):
    def __init__(self, embed_pos_enc_layer_type, embed_rel_pos_type, output_size=256, attention_heads=4, linear_units=2048,
                 num_blocks=12, dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                 acoustic_branch_drop_rate=0.0, attention_layer_type="rel_selfattn", positionwise_layer_type="linear",
                 ffn_activation_type="swish", cgmlp_linear_units=2048, cgmlp_conv_kernel=31, gate_activation="identity",
                 use_linear_after_conv=False, acoustic_use_attn: List[bool] = [True] * 12,
                 visual_use_attn: List[bool] = [False] * 12, macaron=True, zero_triu=False, normalize_before=True,
                 ignore_id=-1, interctc_use_conditioning: bool = False, audiovisual_interctc_conditioning: bool = False,
                 interctc_layer_idx: List[int] = [], stochastic_depth_rate=0.0, max_pos_emb_len: int = 5000):
        super().__init__()
        self.ignore_id, self._output_size = ignore_id, output_size
        if embed_rel_pos_type != "latest":
            raise ValueError("unknown embed_rel_pos_type: " + str(embed_rel_pos_type))
        if embed_pos_enc_layer_type != "rel_pos":
            raise ValueError("unknown pos_enc_layer: " + str(embed_pos_enc_layer_type))
        if attention_layer_type != "rel_selfattn":
            raise ValueError("unknown attention_layer_typer: " + attention_layer_type)
        if positionwise_layer_type != "linear":
            raise ValueError("Support only linear.")
        self.normalize_before = normalize_before
        self.modality_encoding = torch.nn.Embedding(2, output_size)
        self.modality_to_id = {"audio": 0, "video": 1}

        def per_block(v, what):
            v = [v] * num_blocks if isinstance(v, float) else list(v)
            if len(v) != num_blocks:
                raise ValueError(f"Length of {what} ({len(v)}) should be equal to num_blocks ({num_blocks})")
            return v

        sdr = per_block(stochastic_depth_rate, "stochastic_depth_rate")
        abd = per_block(acoustic_branch_drop_rate, "acoustic_branch_drop_rate")
        assert len(acoustic_use_attn) == num_blocks, f"Lenght of acoustic_use_attn ({len(acoustic_use_attn)}) should be equal to num_blocks ({num_blocks})"
        assert len(visual_use_attn) == num_blocks, f"Lenght of visual_use_attn ({len(visual_use_attn)}) should be equal to num_blocks ({num_blocks})"
        ffn = lambda: PositionwiseFeedForward(output_size, linear_units, dropout_rate, ffn_activation_type)
        att = lambda: RelPositionMultiHeadedAttention(attention_heads, output_size, attention_dropout_rate, zero_triu)
        mlp = lambda: ConvolutionalGatingMLP(output_size, cgmlp_linear_units, cgmlp_conv_kernel, dropout_rate,
                                             use_linear_after_conv, gate_activation)
        self.encoders = torch.nn.ModuleList([
            TailoredEncoderLayer(output_size, ffn() if macaron else None,
                                 att() if acoustic_use_attn[i] else None, mlp() if not acoustic_use_attn[i] else None,
                                 att() if visual_use_attn[i] else None, mlp() if not visual_use_attn[i] else None,
                                 ffn(), dropout_rate, abd[i], sdr[i])
            for i in range(num_blocks)])
        if self.normalize_before:
            self.after_norm = LayerNorm(output_size)
        self.interctc_layer_idx = list(interctc_layer_idx)
        if len(self.interctc_layer_idx) > 0:
            assert 0 < min(self.interctc_layer_idx) and max(self.interctc_layer_idx) < num_blocks
        self.interctc_use_conditioning = interctc_use_conditioning
        self.audiovisual_interctc_conditioning = audiovisual_interctc_conditioning
        assert not (self.interctc_use_conditioning is False and self.audiovisual_interctc_conditioning is True), \
            "Audio-Visual InterCTC conditioning only can be applied if interctc_use_conditioning is set to True."
        self.conditioning_layer = None

    def output_size(self) -> int:
        return self._output_size

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, prev_states=None, ctc=None, audiovisual_fusion=None):
        if not (isinstance(audio_pad, tuple) and isinstance(video_pad, tuple)):
            raise NotImplementedError("the HIP path implements the rel_pos form: inputs are (x, pos_emb) tuples")
        x, pos = audio_pad
        audio_pad = (FA.AddRowFn.apply(x, self.modality_encoding.weight[0]), pos)
        x, pos = video_pad
        video_pad = (FA.AddRowFn.apply(x, self.modality_encoding.weight[1]), pos)
        alens = audio_masks.squeeze(1).sum(-1).to(torch.int64)
        vlens = video_masks.squeeze(1).sum(-1).to(torch.int64)
        intermediate_outs = []
        for layer_idx, layer in enumerate(self.encoders):
            audio_pad, audio_masks, video_pad, video_masks = layer(audio_pad, audio_masks, video_pad, video_masks,
                                                                  alens=alens, vlens=vlens)
            if layer_idx + 1 in self.interctc_layer_idx:          # tailored/encoder.py:272-318
                a_out, v_out = audio_pad[0], video_pad[0]
                if self.normalize_before:
                    a_out, v_out = self.after_norm(a_out), self.after_norm(v_out)
                av_out, _ = audiovisual_fusion(a_out, audio_masks, v_out, video_masks)
                intermediate_outs.append((layer_idx + 1, av_out))
                if self.interctc_use_conditioning:
                    ha, hv = (av_out, av_out) if self.audiovisual_interctc_conditioning else (a_out, v_out)
                    cw, cb = self.conditioning_layer.weight, self.conditioning_layer.bias
                    audio_pad = (F_.InterCTCConditionFn.apply(audio_pad[0], ha, ctc.ctc_lo.weight, ctc.ctc_lo.bias, cw, cb),
                                 audio_pad[1])
                    video_pad = (F_.InterCTCConditionFn.apply(video_pad[0], hv, ctc.ctc_lo.weight, ctc.ctc_lo.bias, cw, cb),
                                 video_pad[1])
        audio, video = audio_pad[0], video_pad[0]
        if self.normalize_before:
            audio, video = self.after_norm(audio), self.after_norm(video)
        if len(intermediate_outs) > 0:
            return (audio, intermediate_outs), audio_masks, video, video_masks, None
        return audio, audio_masks, video, video_masks, None
