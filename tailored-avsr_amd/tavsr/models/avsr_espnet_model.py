"""``ESPnetAVSRModel`` - drop-in for src/models/avsr_espnet_model.py:46-685 (hybrid CTC/attention branch).

``forward(audio, audio_lengths, video, video_lengths, text, text_lengths) -> (loss, stats, weight)`` and ``encode`` as
used by inference.  Orchestration only: frontends -> normalise -> per-modality embedding -> temporal alignment
(the shorter stream's FEATURES are padded with ignore_id = -1.0, masks with False: :512-541) -> positional encoding ->
AV encoder -> adaptive fusion -> CTC / attention losses (shared with the audio-only model)."""
from __future__ import annotations

from typing import List, Tuple, Union

import torch

from .. import functional_av as FA
from .. import dp, ops
from ..ctc.ctc import CTC
from .espnet_model import ESPnetASRModel, cut_to_longest


class ESPnetAVSRModel(ESPnetASRModel):
    def __init__(self, vocab_size: int, token_list: Union[Tuple[str, ...], List[str]], specaug, normalize,
                 acoustic_frontend, visual_frontend, acoustic_preencoder, visual_preencoder, acoustic_embed, visual_embed,
                 encoder, audiovisual_fusion, postencoder, decoder, ctc: CTC, joint_network=None, aux_ctc: dict = None,
                 ctc_weight: float = 0.5, interctc_weight: float = 0.0, ignore_id: int = -1, lsm_weight: float = 0.0,
                 length_normalized_loss: bool = False, report_cer: bool = True, report_wer: bool = True,
                 sym_space: str = "<space>", sym_blank: str = "<blank>", transducer_multi_blank_durations: List = [],
                 transducer_multi_blank_sigma: float = 0.05, sym_sos: str = "<sos/eos>", sym_eos: str = "<sos/eos>",
                 extract_feats_in_collect_stats: bool = True, lang_token_id: int = -1):
        if acoustic_preencoder is not None or visual_preencoder is not None:
            raise ValueError("pre-encoders are out of the hot path (no shipped config)")
        super().__init__(vocab_size=vocab_size, token_list=token_list, frontend=None, specaug=specaug, normalize=normalize,
                         preencoder=None, encoder=encoder, postencoder=postencoder, decoder=decoder, ctc=ctc,
                         joint_network=joint_network, aux_ctc=aux_ctc, ctc_weight=ctc_weight, interctc_weight=interctc_weight,
                         ignore_id=ignore_id, lsm_weight=lsm_weight, length_normalized_loss=length_normalized_loss,
                         report_cer=report_cer, report_wer=report_wer, sym_space=sym_space, sym_blank=sym_blank,
                         sym_sos=sym_sos, sym_eos=sym_eos, lang_token_id=lang_token_id)
        del self.frontend
        self.acoustic_frontend, self.visual_frontend = acoustic_frontend, visual_frontend
        self.acoustic_embed, self.visual_embed = acoustic_embed, visual_embed
        self.audiovisual_fusion = audiovisual_fusion

    # ---------------------------------------------------------------- avsr_espnet_model.py:512-541
    def audiovisual_alignment(self, audio_feats, audio_feats_masks, video_feats, video_feats_masks):
        padding_frames = audio_feats.shape[1] - video_feats.shape[1]
        if padding_frames < 0:
            audio_feats = FA.PadTimeFn.apply(audio_feats, -padding_frames, float(self.ignore_id))
            audio_feats_masks = torch.nn.functional.pad(audio_feats_masks, (0, -padding_frames), value=False)
        elif padding_frames > 0:
            video_feats = FA.PadTimeFn.apply(video_feats, padding_frames, float(self.ignore_id))
            video_feats_masks = torch.nn.functional.pad(video_feats_masks, (0, padding_frames), value=False)
        return audio_feats, audio_feats_masks, video_feats, video_feats_masks

    def _front_pair(self, video):
        """the paired node applies when the lip front-end is the Conv3d + ResNet-18 one (one un-chunked call) and the audio
        embedding the two-convolution subsampling: returns a callable or None (then the two run one after the other)."""
        from ..embedding_for_avsr.default import Conv2dSubsamplingWOPosEnc
        from ..frontend.conv3d_resnet18 import Conv3dResNet18
        from ..layers import make_pad_mask
        vf, emb = self.visual_frontend, getattr(self.acoustic_embed, "embed", None)
        if not (FA.FRONT_PAIR and isinstance(vf, Conv3dResNet18) and isinstance(emb, Conv2dSubsamplingWOPosEnc) and video.is_cuda):
            return None
        if not vf.training and not torch.is_grad_enabled() and video.shape[0] > vf.EVAL_CHUNK:
            return None                  # chunked eval decoding of large batches stays on its own path

        def run(video, audio_feats, audio_lens):
            params = dict(vf.named_parameters())
            cfg = dict(names=vf._names, buffers=dict(vf.named_buffers()), training=vf.training)
            P = [params[n] for n in vf._names] + [emb.conv[0].weight, emb.conv[0].bias, emb.conv[2].weight, emb.conv[2].bias,
                                                   emb.out.weight, emb.out.bias]
            yv, ya = FA.FrontendPairFn.apply(video, cfg, audio_feats, *P)
            masks = (~make_pad_mask(audio_lens, audio_feats.size(1))[:, None, :]).to(audio_feats.device)
            for k, s_ in zip(emb.kernels, emb.strides):
                masks = masks[:, :, : -k + 1: s_]
            return yv, ya, masks

        return run

    # ---------------------------------------------------------------- avsr_espnet_model.py:383-488
    def encode(self, audio, audio_lengths, video, video_lengths):
        audio = cut_to_longest(audio, audio_lengths)        # _extract_feats, avsr_espnet_model.py:499
        video = cut_to_longest(video, video_lengths)
        if self.acoustic_frontend is not None:
            audio_feats, audio_feats_lengths = self.acoustic_frontend(audio, audio_lengths)
        else:
            audio_feats, audio_feats_lengths = audio, audio_lengths
        if self.specaug is not None and self.training:
            audio_feats, audio_feats_lengths = self.specaug(audio_feats, audio_feats_lengths)
        if self.normalize is not None:
            audio_feats, audio_feats_lengths = self.normalize(audio_feats, audio_feats_lengths)
        pair = self._front_pair(video)
        if pair is not None:
            # lip front-end and audio embedding side by side as one autograd node (the same arithmetic, two streams)
            video_feats, audio_feats, audio_masks = pair(video, audio_feats, audio_feats_lengths)
            video_feats_lengths = video_lengths
        else:
            if self.visual_frontend is not None:
                video_feats, video_feats_lengths = self.visual_frontend(video, video_lengths)
            else:
                video_feats, video_feats_lengths = video, video_lengths
            audio_feats, audio_masks = self.acoustic_embed.apply_embed_layer(audio_feats, audio_feats_lengths)
        # where a data-parallel step may split its backward pass (tavsr.dp.TwoPhaseBackward; identity otherwise): everything below
        # these two tensors is the front-ends' backward pass, the longest stretch of the step that completes no encoder / decoder
        # gradient bucket
        video_feats, audio_feats = dp.cut(video_feats, audio_feats)
        video_feats, video_masks = self.visual_embed.apply_embed_layer(video_feats, video_feats_lengths)
        audio_feats, audio_masks, video_feats, video_masks = self.audiovisual_alignment(audio_feats, audio_masks,
                                                                                        video_feats, video_masks)
        audio_feats = self.acoustic_embed.apply_pos_enc(audio_feats)
        video_feats = self.visual_embed.apply_pos_enc(video_feats)
        audio_out, audio_out_masks, video_out, video_out_masks, _ = self.encoder(
            audio_feats, audio_masks, video_feats, video_masks,
            ctc=self.ctc if self.encoder.interctc_use_conditioning else None,
            audiovisual_fusion=self.audiovisual_fusion if len(self.encoder.interctc_layer_idx) > 0 else None)
        intermediate_outs = None                      # avsr_espnet_model.py:460-487
        if isinstance(audio_out, tuple):
            audio_out, intermediate_outs = audio_out
        encoder_out, encoder_out_lens = self.audiovisual_fusion(audio_out, audio_out_masks, video_out, video_out_masks)
        assert encoder_out.size(0) == video.size(0), (encoder_out.size(), video.size(0))
        if intermediate_outs is not None:
            return (encoder_out, intermediate_outs), encoder_out_lens
        return encoder_out, encoder_out_lens

    # ---------------------------------------------------------------- avsr_espnet_model.py:211-367
    def forward(self, audio, audio_lengths, video, video_lengths, text, text_lengths, **kwargs):
        assert text_lengths.dim() == 1, text_lengths.shape
        assert (audio.shape[0] == audio_lengths.shape[0] == video.shape[0] == video_lengths.shape[0] == text.shape[0]
                == text_lengths.shape[0]), (audio.shape, audio_lengths.shape, video.shape, video_lengths.shape, text.shape,
                                            text_lengths.shape)
        batch_size = audio.shape[0]
        ops.rng_step_begin(audio.device)
        text = cut_to_longest(text.to(torch.int64).masked_fill(text == -1, self.ignore_id), text_lengths)
        encoder_out, encoder_out_lens = self.encode(audio, audio_lengths, video, video_lengths)
        return self._hybrid_loss(encoder_out, encoder_out_lens, text, text_lengths, batch_size)

    @torch.no_grad()
    def ctc_greedy(self, audio, audio_lengths, video, video_lengths):
        """-> (ids (B,T), hyp (B,T) padded -1, hyp_len (B)): argmax + collapse, integer exact."""
        enc, olens = self.encode(audio, audio_lengths, video, video_lengths)
        return self.ctc.greedy(enc, olens, self.blank_id)
