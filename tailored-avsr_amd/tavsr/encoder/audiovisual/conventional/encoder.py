"""``ConventionalEncoder`` - drop-in for src/encoder/audiovisual/conventional/encoder.py:35-217: two independent
``MyBranchformerEncoder``s (embed=None) run layer by layer on the audio and the video stream."""
from __future__ import annotations

from typing import List

import torch

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
        self.interctc_layer_idx = list(interctc_layer_idx)
        self.interctc_use_conditioning = interctc_use_conditioning
        self.audiovisual_interctc_conditioning = audiovisual_interctc_conditioning
        self.conditioning_layer = None

    def get_encoder_class(self, encoder_class_type):
        if encoder_class_type == "branchformer":
            return MyBranchformerEncoder
        raise ValueError(f"the HIP path covers encoder_class_type='branchformer': {encoder_class_type}")

    def output_size(self) -> int:
        return self.acoustic_encoder.output_size()

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, prev_states=None, ctc=None, audiovisual_fusion=None):
        if len(self.interctc_layer_idx) > 0:
            raise NotImplementedError("intermediate CTC is not used by the shipped AVSR recipes (interctc_weight: 0.0)")
        alens = audio_masks.squeeze(1).sum(-1).to(torch.int64)
        vlens = video_masks.squeeze(1).sum(-1).to(torch.int64)
        for la, lv in zip(self.acoustic_encoder.encoders, self.visual_encoder.encoders):
            audio_pad, audio_masks = la(audio_pad, audio_masks, lens=alens)
            video_pad, video_masks = lv(video_pad, video_masks, lens=vlens)
        audio, video = audio_pad[0], video_pad[0]
        if self.acoustic_encoder.normalize_before:
            audio = self.acoustic_encoder.after_norm(audio)
        if self.visual_encoder.normalize_before:
            video = self.visual_encoder.after_norm(video)
        return audio, audio_masks, video, video_masks, None
