"""``AdaptiveAudioVisualFusion`` - drop-in for src/audiovisual_fusion/adaptive_audiovisual_fusion.py:29-211
(all three merge methods; ``learned_ave`` is the one the shipped configs use): same constructor, forward, state_dict keys and the
``acoustic_weight`` / ``visual_weight`` introspection attributes (read by src/scripts/study_adaptive_fusion.py:39-40)."""
from __future__ import annotations

import torch

from .. import functional_av as FA
from ..layers import LayerNorm, PositionwiseFeedForward


class AdaptiveAudioVisualFusion(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int = 256, hidden_units: int = 2048,
                 audiovisual_layer_type: str = "upsampling_positionwise", merge_method: str = "learned_ave",
                 activation_type: str = "swish", acoustic_weight: float = 0.5, dropout_rate: float = 0.1,
                 acoustic_branch_drop_rate: float = 0.0):
        super().__init__()
        self.input_size, self._output_size = input_size, output_size
        self.acoustic_weight = acoustic_weight
        self.visual_weight = None
        self.acoustic_branch_drop_rate = acoustic_branch_drop_rate
        if audiovisual_layer_type != "upsampling_positionwise":
            raise ValueError("Support only upsampling positionwise feed forward fusion.")
        self.merge_method = merge_method
        if merge_method == "learned_ave":
            self.acoustic_pooling_proj = torch.nn.Linear(input_size, 1)
            self.visual_pooling_proj = torch.nn.Linear(input_size, 1)
            self.acoustic_weight_proj = torch.nn.Linear(input_size, 1)
            self.visual_weight_proj = torch.nn.Linear(input_size, 1)
            self.audiovisual_layer = PositionwiseFeedForward(input_size, hidden_units, dropout_rate, activation_type)
        elif merge_method == "concat":            # adaptive_audiovisual_fusion.py:72-78: FFN over the concatenated streams
            self.audiovisual_layer = PositionwiseFeedForward(input_size + input_size, hidden_units, dropout_rate, activation_type)
        elif merge_method == "fixed_ave":         # :96-105
            assert 0.0 <= acoustic_weight <= 1.0, "cgmlp weight should be between 0.0 and 1.0"
            self.audiovisual_layer = PositionwiseFeedForward(input_size, hidden_units, dropout_rate, activation_type)
        else:
            raise ValueError(f"Unknow merge method: {merge_method}")
        self.norm_final = LayerNorm(output_size)
        self.dropout_rate = dropout_rate

    def output_size(self) -> int:
        return self._output_size

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, cache=None):
        if cache is not None:
            raise NotImplementedError("cache is not None, which is not tested")
        # adaptive_audiovisual_fusion.py:138-144: with probability acoustic_branch_drop_rate the step fuses with the constant
        # weights (0, 1) - the video stream alone; the pooling / weight projections are not part of that step
        dropped = (self.merge_method == "learned_ave" and self.training and self.acoustic_branch_drop_rate > 0
                   and torch.rand(1).item() < self.acoustic_branch_drop_rate)
        alens = audio_masks.squeeze(1).sum(-1).to(torch.int64)
        vlens = video_masks.squeeze(1).sum(-1).to(torch.int64)
        sd = dict(self.named_parameters())
        cfg = dict(act=self.audiovisual_layer.activation, p=self.dropout_rate if self.training else 0.0, drop_acoustic=dropped)
        if self.merge_method != "learned_ave":    # concat (:132-135) / fixed_ave (:197-200): no pooling, no learned weights
            cfg["mode"] = "concat" if self.merge_method == "concat" else "const"
            if self.merge_method == "fixed_ave":
                cfg["w"] = (float(self.acoustic_weight), 1.0 - float(self.acoustic_weight))
            out = FA.FusionFn.apply(audio_pad, video_pad, alens, vlens, cfg, *[sd[n] for n in FA.FUSION_PARAM_NAMES[8:]])
            return out, torch.maximum(alens, vlens)
        out = FA.FusionFn.apply(audio_pad, video_pad, alens, vlens, cfg, *[sd[n] for n in FA.FUSION_PARAM_NAMES])
        if dropped:
            self.acoustic_weight, self.visual_weight = 0.0, 1.0
        else:
            w = cfg["_last_w"]   # (B,2) -> the reference's (B,1,1) tensors (:186-191)
            self.acoustic_weight, self.visual_weight = w[:, 0].view(-1, 1, 1), w[:, 1].view(-1, 1, 1)
        olens = torch.maximum(alens, vlens)      # logical_or of two prefix masks, summed (:208-209)
        return out, olens
