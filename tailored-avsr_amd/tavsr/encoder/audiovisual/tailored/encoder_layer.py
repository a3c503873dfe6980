"""``TailoredEncoderLayer`` - drop-in for src/encoder/audiovisual/tailored/encoder_layer.py:49-274.

Per modality ONE of {rel-pos self-attention, cgMLP}; the two feed-forward modules and the three LayerNorms
(norm_ff_macaron, norm_ff, norm_final) are shared by the audio and the video stream.  Each stream is one autograd
node on the gfx950 kernels (``tavsr.functional_av.TailoredStreamFn``)."""
from __future__ import annotations

from typing import Optional

import torch

from .... import functional as F_
from .... import functional_av as FA
from ....layers import LayerNorm


class TailoredEncoderLayer(torch.nn.Module):
    def __init__(self, size: int, feed_forward_macaron: Optional[torch.nn.Module], acoustic_attn: Optional[torch.nn.Module],
                 acoustic_cgmlp: Optional[torch.nn.Module], visual_attn: Optional[torch.nn.Module],
                 visual_cgmlp: Optional[torch.nn.Module], feed_forward: Optional[torch.nn.Module], dropout_rate: float,
                 acoustic_branch_drop_rate: float = 0.0, stochastic_depth_rate: float = 0.0):
        super().__init__()
        if feed_forward_macaron is None or feed_forward is None:
            raise ValueError("macaron=True with both feed-forward modules is required (the reference crashes otherwise)")
        self.size, self.ff_scale = size, 0.5
        self.feed_forward_macaron = feed_forward_macaron
        self.norm_ff_macaron = LayerNorm(size)
        self.acoustic_attn = acoustic_attn
        if acoustic_attn is not None:
            self.acoustic_norm_mha = LayerNorm(size)
        self.acoustic_cgmlp = acoustic_cgmlp
        if acoustic_cgmlp is not None:
            self.acoustic_norm_cgmlp = LayerNorm(size)
        self.visual_attn = visual_attn
        if visual_attn is not None:
            self.visual_norm_mha = LayerNorm(size)
        self.visual_cgmlp = visual_cgmlp
        if visual_cgmlp is not None:
            self.visual_norm_cgmlp = LayerNorm(size)
        self.feed_forward = feed_forward
        self.norm_ff = LayerNorm(size)
        self.norm_final = LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.acoustic_branch_drop_rate = acoustic_branch_drop_rate
        self.stochastic_depth_rate = stochastic_depth_rate

    def _stream_params(self, prefix: str, use_attn: bool):
        cache = self.__dict__.setdefault("_tavsr_stream_pcache", {})      # Parameter identities never change: look up once
        if (prefix, use_attn) in cache:
            return cache[(prefix, use_attn)]
        sd = dict(self.named_parameters())
        out = cache[(prefix, use_attn)] = []
        for n in FA.tailored_stream_param_names(use_attn):
            if n.startswith(("attn.", "cgmlp.")):
                out.append(sd[prefix + "_" + n])
            elif n.startswith(("norm_mha.", "norm_cgmlp.")):
                out.append(sd[prefix + "_" + n])
            else:
                out.append(sd[n])
        return out

    def _active_dropout(self) -> bool:
        rates = [self.dropout_rate]
        for m in (self.acoustic_attn, self.visual_attn):
            if m is not None:
                rates.append(m.dropout_rate)
        return self.training and any(r > 0 for r in rates)

    def forward(self, audio_input, audio_masks, video_input, video_masks, cache=None, alens=None, vlens=None):
        if cache is not None:
            raise NotImplementedError("cache is not None, which is not tested")
        if not (isinstance(audio_input, tuple) and isinstance(video_input, tuple)):
            raise NotImplementedError("the HIP path implements the rel_pos form: inputs are (x, pos_emb) tuples")
        audio, apos = audio_input
        video, vpos = video_input
        coeff = 1.0
        if self.training and self.stochastic_depth_rate > 0:
            skip = torch.rand(1).item() < self.stochastic_depth_rate
            coeff = 1.0 / (1 - self.stochastic_depth_rate)
            if skip:
                return (audio, apos), audio_masks, (video, vpos), video_masks
        for name, a, c in (("acoustic", self.acoustic_attn, self.acoustic_cgmlp), ("visual", self.visual_attn, self.visual_cgmlp)):
            if (a is not None) and (c is not None):
                raise RuntimeError(f"Only one of the possible {name} tailored modules should be not None: {a}, {c}.")
        if alens is None:
            alens = audio_masks.squeeze(1).sum(-1).to(torch.int64)
        if vlens is None:
            vlens = video_masks.squeeze(1).sum(-1).to(torch.int64)
        act = self.feed_forward.activation
        ua, uv = self.acoustic_attn is not None, self.visual_attn is not None
        pd = self.dropout_rate if self.training else 0.0
        cfg_a = dict(use_attn=ua, heads=self.acoustic_attn.h if ua else 1, ffn_act=act, coeff=coeff, p=pd,
                     p_att=self.acoustic_attn.dropout_rate if (ua and self.training) else 0.0)
        cfg_v = dict(use_attn=uv, heads=self.visual_attn.h if uv else 1, ffn_act=act, coeff=coeff, p=pd,
                     p_att=self.visual_attn.dropout_rate if (uv and self.training) else 0.0)
        ns = len(FA.TS_SHARED)
        pa, pv = self._stream_params("acoustic", ua), self._stream_params("visual", uv)
        # one node for both streams: the video stream runs on the forked stream beside the audio stream
        audio, video = F_.grad_apply(FA.TailoredLayerFn, audio, apos, alens, cfg_a, video, vpos, vlens, cfg_v, *pa, *pv[ns:])
        return (audio, apos), audio_masks, (video, vpos), video_masks
