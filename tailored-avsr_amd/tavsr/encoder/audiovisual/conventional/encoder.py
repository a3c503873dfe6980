"""``ConventionalEncoder`` - drop-in for src/encoder/audiovisual/conventional/encoder.py:35-217: two independent
``MyBranchformerEncoder``s (embed=None) run layer by layer on the audio and the video stream."""
from __future__ import annotations

from typing import List

import torch

from .... import functional as F_
from ...branchformer.encoder import MyBranchformerEncoder


class ConventionalEncoder(torch.nn.Module):
    def __init__(self, input_size, acoustic_encoder_conf, visual_encoder_conf, output_size: int = 256,
                 embed_pos_enc_layer_type: str = "rel_pos", embed_rel_pos_type: str = "latest",
                 interctc_use_conditioning: bool = False, audiovisual_interctc_conditioning: bool = False,
                 interctc_layer_idx: List[int] = []):
        super().__init__()
        ac, vc = dict(acoustic_encoder_conf), dict(visual_encoder_conf)
        assert embed_pos_enc_layer_type == ac["pos_enc_layer_type"] == vc["pos_enc_layer_type"], (
            embed_pos_enc_layer_type, ac["pos_enc_layer_type"], vc["pos_enc_layer_type"])
        assert embed_rel_pos_type == ac["rel_pos_type"] == vc["rel_pos_type"], (
            embed_rel_pos_type, ac["rel_pos_type"], vc["rel_pos_type"])
        self.acoustic_encoder = self.get_encoder_class(ac.pop("encoder_class_type"))(input_size=input_size,
                                                                                      output_size=output_size, **ac)
        self.visual_encoder = self.get_encoder_class(vc.pop("encoder_class_type"))(input_size=input_size,
                                                                                    output_size=output_size, **vc)
        assert len(self.acoustic_encoder.encoders) == len(self.visual_encoder.encoders), \
            "Both encoders must have the same number of blocks."
        assert self.acoustic_encoder.output_size() == self.visual_encoder.output_size(), \
            "Output size should be the same in both wrapped encoders."
        assert self.acoustic_encoder.embed is None and self.visual_encoder.embed is None, \
            "The embedding layers of both encoders should be None."
        assert len(self.acoustic_encoder.interctc_layer_idx) == 0 and len(self.visual_encoder.interctc_layer_idx) == 0, \
            "InterCTC loss must be defined in the WrapperEncoder."
        assert (self.acoustic_encoder.interctc_use_conditioning is False
                and self.visual_encoder.interctc_use_conditioning is False), "InterCTC conditioning must be defined in the WrapperEncoder."
        num_blocks = len(self.acoustic_encoder.encoders)
        self.interctc_layer_idx = list(interctc_layer_idx)
        if len(self.interctc_layer_idx) > 0:
            assert 0 < min(self.interctc_layer_idx) and max(self.interctc_layer_idx) < num_blocks
        self.interctc_use_conditioning = interctc_use_conditioning
        self.audiovisual_interctc_conditioning = audiovisual_interctc_conditioning
        assert not (self.interctc_use_conditioning is False and self.audiovisual_interctc_conditioning is True), \
            "Audio-Visual InterCTC conditioning only can be applied if interctc_use_conditioning is set to True."
        self.conditioning_layer = None

    def get_encoder_class(self, encoder_class_type):
        if encoder_class_type == "branchformer":
            return MyBranchformerEncoder
        if encoder_class_type == "conformer":
            # the reference names espnet2's ConformerEncoder here (conventional/encoder.py:211-213), but as published at espnet
            # 202402 that class builds embed = Sequential(pos_enc) for input_layer=None, which this wrapper's own
            # "embed is None" assertion (:92-94) rejects: the choice cannot be constructed in the reference either
            raise ValueError("encoder_class_type 'conformer' cannot pass the wrapper's embed-is-None assertion "
                             "(conventional/encoder.py:92-94); use 'branchformer'")
        raise ValueError("unknown encoder_class_type: " + encoder_class_type)

    def output_size(self) -> int:
        return self.acoustic_encoder.output_size()

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, prev_states=None, ctc=None, audiovisual_fusion=None):
        alens = audio_masks.squeeze(1).sum(-1).to(torch.int64)
        vlens = video_masks.squeeze(1).sum(-1).to(torch.int64)
        ae, ve = self.acoustic_encoder, self.visual_encoder
        intermediate_outs = []
        for layer_idx, (la, lv) in enumerate(zip(ae.encoders, ve.encoders)):
            audio_pad, audio_masks = la(audio_pad, audio_masks, lens=alens)
            video_pad, video_masks = lv(video_pad, video_masks, lens=vlens)
            if layer_idx + 1 in self.interctc_layer_idx:          # conventional/encoder.py:154-199
                a_out, v_out = audio_pad[0], video_pad[0]
                if ae.normalize_before:
                    a_out = ae.after_norm(a_out)
                if ve.normalize_before:
                    v_out = ve.after_norm(v_out)
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
        if ae.normalize_before:
            audio = ae.after_norm(audio)
        if ve.normalize_before:
            video = ve.after_norm(video)
        if len(intermediate_outs) > 0:
            return (audio, intermediate_outs), audio_masks, video, video_masks, None
        return audio, audio_masks, video, video_masks, None
