"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU restatement of the reference's AUDIO-VISUAL composites:
visual frontend, per-modality embedding, tailored / conventional AV encoders, adaptive fusion and the
AVSR model.  Written against ``oracle.leaves`` / ``oracle.model``; every class cites the reference
file:line it follows and keeps the reference's ``state_dict`` keys.

Pinned by ``tests/golden/av_*.npz`` (``oracle/gen_golden.py`` imports the reference's own modules; the visual
frontend needs only name-level stand-ins, so its vectors are direct reference outputs).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import leaves as L
from .model import BranchformerEncoderOracle, CTCOracle, _masked_time_softmax


# --------------------------------------------------------------------------------------------------
# visual frontend: src/frontend/conv3d_resnet18/conv3d_resnet18.py:39-97, modules/resnet.py:8-178
# --------------------------------------------------------------------------------------------------
def _conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, 3, stride, 1, bias=False)            # resnet.py:8-22


class BasicBlockOracle(nn.Module):
    """``BasicBlock`` (modules/resnet.py:44-106), activation_type="swish"."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1, self.relu2 = L.Swish(), L.Swish()
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu2(out + residual)


class ResNetOracle(nn.Module):
    """``ResNet(BasicBlock, [2,2,2,2])`` (modules/resnet.py:109-178)."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.layer1 = self._make(64, 2)
        self.layer2 = self._make(128, 2, 2)
        self.layer3 = self._make(256, 2, 2)
        self.layer4 = self._make(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)

    def _make(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlockOracle(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [BasicBlockOracle(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x).view(x.size(0), -1)


class Conv3dResNet18Oracle(nn.Module):
    """``Conv3dResNet18`` (conv3d_resnet18.py:39-97): (B,T,88,88) -> (B,T,512)."""

    def __init__(self, activation_type="swish"):
        super().__init__()
        if activation_type != "swish":
            raise ValueError("shipped configs use swish")
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), bias=False), nn.BatchNorm3d(64), L.Swish(),
            nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        self.trunk = ResNetOracle()

    def output_size(self):
        return 512

    def forward(self, speech, speech_lengths):
        x = speech.unsqueeze(1)
        B, C, T, H, W = x.size()
        x = self.frontend3D(x)
        n, c, t, h, w = x.shape
        x = x.transpose(1, 2).reshape(n * t, c, h, w)          # threeD_to_2D_tensor (:23-36)
        x = self.trunk(x)
        return x.view(B, T, x.size(-1)), speech_lengths


# --------------------------------------------------------------------------------------------------
# per-modality embedding: src/embedding_for_avsr/default.py:22-162
# --------------------------------------------------------------------------------------------------
class DefaultEmbeddingOracle(nn.Module):
    def __init__(self, input_size, output_size, pos_enc_layer_type="rel_pos", rel_pos_type="latest",
                 input_layer="conv2d", dropout_rate=0.1, positional_dropout_rate=0.1, max_pos_emb_len=5000):
        super().__init__()
        self._output_size, self._rel_pos_type, self._pos_enc_layer_type = output_size, rel_pos_type, pos_enc_layer_type
        if input_layer == "linear":
            self.embed = nn.Sequential(nn.Linear(input_size, output_size), nn.LayerNorm(output_size),
                                       nn.Dropout(dropout_rate))       # torch LayerNorm: eps 1e-5 (:57-62)
        elif input_layer == "conv2d":
            self.embed = L.Conv2dSubsamplingWOPosEnc(input_size, output_size, dropout_rate, kernels=[3, 3], strides=[2, 2])
        else:
            raise ValueError("oracle covers input_layer in {linear, conv2d}")
        if rel_pos_type != "latest" or pos_enc_layer_type != "rel_pos":
            raise ValueError("oracle covers rel_pos/latest")
        self.pos_enc = L.RelPositionalEncoding(output_size, positional_dropout_rate, max_pos_emb_len)

    def output_size(self):
        return self._output_size

    def apply_embed_layer(self, xs_pad, ilens):                       # :140-154
        masks = (~L.make_pad_mask(ilens)[:, None, :]).to(xs_pad.device)
        if isinstance(self.embed, L.Conv2dSubsamplingWOPosEnc):
            xs_pad, masks = self.embed(xs_pad, masks)
        else:
            xs_pad = self.embed(xs_pad)
        return xs_pad, masks

    def apply_pos_enc(self, xs_pad):                                  # :157-162
        return self.pos_enc(xs_pad)

    def forward(self, xs_pad, ilens):                                 # :111-138
        xs_pad, masks = self.apply_embed_layer(xs_pad, ilens)
        return self.pos_enc(xs_pad), masks


# --------------------------------------------------------------------------------------------------
# tailored AV encoder: src/encoder/audiovisual/tailored/encoder_layer.py:49-274, encoder.py:36-332
# --------------------------------------------------------------------------------------------------
class TailoredLayerOracle(nn.Module):
    def __init__(self, size, feed_forward_macaron, acoustic_attn, acoustic_cgmlp, visual_attn, visual_cgmlp,
                 feed_forward, dropout_rate, acoustic_branch_drop_rate=0.0, stochastic_depth_rate=0.0):
        super().__init__()
        self.size, self.ff_scale = size, 0.5
        self.feed_forward_macaron = feed_forward_macaron
        self.norm_ff_macaron = L.LayerNorm(size)
        self.acoustic_attn = acoustic_attn
        if acoustic_attn is not None:
            self.acoustic_norm_mha = L.LayerNorm(size)
        self.acoustic_cgmlp = acoustic_cgmlp
        if acoustic_cgmlp is not None:
            self.acoustic_norm_cgmlp = L.LayerNorm(size)
        self.visual_attn = visual_attn
        if visual_attn is not None:
            self.visual_norm_mha = L.LayerNorm(size)
        self.visual_cgmlp = visual_cgmlp
        if visual_cgmlp is not None:
            self.visual_norm_cgmlp = L.LayerNorm(size)
        self.feed_forward = feed_forward
        self.norm_ff = L.LayerNorm(size)
        self.norm_final = L.LayerNorm(size)
        self.dropout = nn.Dropout(dropout_rate)
        self.stochastic_depth_rate = stochastic_depth_rate

    def _stream(self, x, pos, mask, attn, norm_mha, cgmlp, norm_cgmlp, coeff):
        x = x + self.ff_scale * self.dropout(self.feed_forward_macaron(self.norm_ff_macaron(x)))   # :173-175
        if attn is not None and cgmlp is not None:
            raise RuntimeError("Only one of the possible tailored modules should be not None")      # :179-182
        residual = x
        if attn is not None:                                                                       # :185-196
            y = norm_mha(x)
            x = residual + coeff * self.dropout(attn(y, y, y, pos, mask))
        if cgmlp is not None:                                                                      # :199-208
            y = cgmlp((norm_cgmlp(x), pos), mask)
            y = y[0] if isinstance(y, tuple) else y
            x = residual + coeff * self.dropout(y)
        x = x + self.ff_scale * self.dropout(self.feed_forward(self.norm_ff(x)))                   # :211-213
        return self.norm_final(x)                                                                  # :216

    def forward(self, audio_input, audio_masks, video_input, video_masks, cache=None):
        if cache is not None:
            raise NotImplementedError("cache is not None, which is not tested")
        audio, apos = audio_input
        video, vpos = video_input
        coeff = 1.0
        if self.training and self.stochastic_depth_rate > 0:
            if torch.rand(1).item() < self.stochastic_depth_rate:
                return (audio, apos), audio_masks, (video, vpos), video_masks
            coeff = 1.0 / (1 - self.stochastic_depth_rate)
        audio = self._stream(audio, apos, audio_masks, self.acoustic_attn, getattr(self, "acoustic_norm_mha", None),
                             self.acoustic_cgmlp, getattr(self, "acoustic_norm_cgmlp", None), coeff)
        video = self._stream(video, vpos, video_masks, self.visual_attn, getattr(self, "visual_norm_mha", None),
                             self.visual_cgmlp, getattr(self, "visual_norm_cgmlp", None), coeff)
        return (audio, apos), audio_masks, (video, vpos), video_masks


class TailoredEncoderOracle(nn.Module):
    def __init__(self, embed_pos_enc_layer_type, embed_rel_pos_type, output_size=256, attention_heads=4,
                 linear_units=2048, num_blocks=12, dropout_rate=0.1, positional_dropout_rate=0.1,
                 attention_dropout_rate=0.1, acoustic_branch_drop_rate=0.0, attention_layer_type="rel_selfattn",
                 positionwise_layer_type="linear", ffn_activation_type="swish", cgmlp_linear_units=2048,
                 cgmlp_conv_kernel=31, gate_activation="identity", use_linear_after_conv=False,
                 acoustic_use_attn=(True,) * 12, visual_use_attn=(False,) * 12, macaron=True, zero_triu=False,
                 normalize_before=True, ignore_id=-1, interctc_use_conditioning=False,
                 audiovisual_interctc_conditioning=False, interctc_layer_idx=(), stochastic_depth_rate=0.0,
                 max_pos_emb_len=5000):
        super().__init__()
        if embed_rel_pos_type != "latest" or embed_pos_enc_layer_type != "rel_pos" or attention_layer_type != "rel_selfattn":
            raise ValueError("oracle covers rel_pos/latest/rel_selfattn")
        self._output_size, self.normalize_before = output_size, normalize_before
        self.modality_encoding = nn.Embedding(2, output_size)                                      # encoder.py:102
        assert len(acoustic_use_attn) == num_blocks and len(visual_use_attn) == num_blocks
        sd = [stochastic_depth_rate] * num_blocks if isinstance(stochastic_depth_rate, float) else list(stochastic_depth_rate)
        ffn = lambda: L.PositionwiseFeedForward(output_size, linear_units, dropout_rate, L.get_activation(ffn_activation_type))
        att = lambda: L.RelPositionMultiHeadedAttention(attention_heads, output_size, attention_dropout_rate, zero_triu)
        mlp = lambda: L.ConvolutionalGatingMLP(output_size, cgmlp_linear_units, cgmlp_conv_kernel, dropout_rate,
                                               use_linear_after_conv, gate_activation)
        self.encoders = L.repeat(num_blocks, lambda i: TailoredLayerOracle(
            output_size, ffn() if macaron else None,
            att() if acoustic_use_attn[i] else None, mlp() if not acoustic_use_attn[i] else None,
            att() if visual_use_attn[i] else None, mlp() if not visual_use_attn[i] else None,
            ffn(), dropout_rate, 0.0, sd[i]))
        if normalize_before:
            self.after_norm = L.LayerNorm(output_size)
        self.interctc_layer_idx = list(interctc_layer_idx)
        self.interctc_use_conditioning = interctc_use_conditioning
        self.audiovisual_interctc_conditioning = audiovisual_interctc_conditioning
        self.conditioning_layer = None

    def output_size(self):
        return self._output_size

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, prev_states=None, ctc=None, audiovisual_fusion=None):
        x, pos = audio_pad
        audio_pad = (x + self.modality_encoding.weight[0], pos)                                    # :251-256
        x, pos = video_pad
        video_pad = (x + self.modality_encoding.weight[1], pos)                                    # :258-263
        inter = []
        for idx, layer in enumerate(self.encoders):
            audio_pad, audio_masks, video_pad, video_masks = layer(audio_pad, audio_masks, video_pad, video_masks)
            if idx + 1 in self.interctc_layer_idx:                                                 # :274-318
                a, v = audio_pad[0], video_pad[0]
                if self.normalize_before:
                    a, v = self.after_norm(a), self.after_norm(v)
                av, _ = audiovisual_fusion(a, audio_masks, v, video_masks)
                inter.append((idx + 1, av))
                if self.interctc_use_conditioning:
                    if self.audiovisual_interctc_conditioning:
                        ca = cv = ctc.softmax(av)
                    else:
                        ca, cv = ctc.softmax(a), ctc.softmax(v)
                    audio_pad = (audio_pad[0] + self.conditioning_layer(ca), audio_pad[1])
                    video_pad = (video_pad[0] + self.conditioning_layer(cv), video_pad[1])
        a, v = audio_pad[0], video_pad[0]
        if self.normalize_before:
            a, v = self.after_norm(a), self.after_norm(v)
        if inter:
            return (a, inter), audio_masks, v, video_masks, None
        return a, audio_masks, v, video_masks, None


# --------------------------------------------------------------------------------------------------
# conventional AV encoder: src/encoder/audiovisual/conventional/encoder.py:35-217 (two branchformers, embed=None)
# --------------------------------------------------------------------------------------------------
class ConventionalEncoderOracle(nn.Module):
    def __init__(self, input_size, acoustic_encoder_conf, visual_encoder_conf, output_size=256,
                 embed_pos_enc_layer_type="rel_pos", embed_rel_pos_type="latest", interctc_use_conditioning=False,
                 audiovisual_interctc_conditioning=False, interctc_layer_idx=()):
        super().__init__()
        ac, vc = dict(acoustic_encoder_conf), dict(visual_encoder_conf)
        assert embed_pos_enc_layer_type == ac["pos_enc_layer_type"] == vc["pos_enc_layer_type"]          # :55-59
        assert embed_rel_pos_type == ac["rel_pos_type"] == vc["rel_pos_type"]                            # :61-65
        if ac.pop("encoder_class_type") != "branchformer" or vc.pop("encoder_class_type") != "branchformer":
            raise ValueError("oracle covers branchformer sub-encoders")
        self.acoustic_encoder = BranchformerEncoderOracle(input_size=input_size, output_size=output_size, **ac)
        self.visual_encoder = BranchformerEncoderOracle(input_size=input_size, output_size=output_size, **vc)
        assert self.acoustic_encoder.embed is None and self.visual_encoder.embed is None                  # :92-94
        self.interctc_layer_idx = list(interctc_layer_idx)
        self.interctc_use_conditioning = interctc_use_conditioning
        self.audiovisual_interctc_conditioning = audiovisual_interctc_conditioning
        self.conditioning_layer = None

    def output_size(self):
        return self.acoustic_encoder.output_size()

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, prev_states=None, ctc=None, audiovisual_fusion=None):
        ae, ve = self.acoustic_encoder, self.visual_encoder
        inter = []
        for idx, (la, lv) in enumerate(zip(ae.encoders, ve.encoders)):           # :150-153 lock step
            audio_pad, audio_masks = la(audio_pad, audio_masks)
            video_pad, video_masks = lv(video_pad, video_masks)
            if idx + 1 in self.interctc_layer_idx:                                   # :154-199
                a_out, v_out = audio_pad[0], video_pad[0]
                if ae.normalize_before:
                    a_out = ae.after_norm(a_out)
                if ve.normalize_before:
                    v_out = ve.after_norm(v_out)
                av_out, _ = audiovisual_fusion(a_out, audio_masks, v_out, video_masks)
                inter.append((idx + 1, av_out))
                if self.interctc_use_conditioning:
                    if self.audiovisual_interctc_conditioning:
                        ca = cv = ctc.softmax(av_out)
                    else:
                        ca, cv = ctc.softmax(a_out), ctc.softmax(v_out)
                    audio_pad = (audio_pad[0] + self.conditioning_layer(ca), audio_pad[1])
                    video_pad = (video_pad[0] + self.conditioning_layer(cv), video_pad[1])
        a, v = audio_pad[0], video_pad[0]
        if ae.normalize_before:
            a = ae.after_norm(a)
        if ve.normalize_before:
            v = ve.after_norm(v)
        if inter:
            return (a, inter), audio_masks, v, video_masks, None
        return a, audio_masks, v, video_masks, None


# --------------------------------------------------------------------------------------------------
# adaptive AV fusion: src/audiovisual_fusion/adaptive_audiovisual_fusion.py:29-211
# --------------------------------------------------------------------------------------------------
class AdaptiveFusionOracle(nn.Module):
    def __init__(self, input_size, output_size=256, hidden_units=2048, audiovisual_layer_type="upsampling_positionwise",
                 merge_method="learned_ave", activation_type="swish", acoustic_weight=0.5, dropout_rate=0.1,
                 acoustic_branch_drop_rate=0.0):
        super().__init__()
        if audiovisual_layer_type != "upsampling_positionwise":
            raise ValueError("Support only upsampling positionwise feed forward fusion.")
        self.input_size, self._output_size = input_size, output_size
        self.acoustic_weight, self.merge_method = acoustic_weight, merge_method
        self.acoustic_branch_drop_rate = acoustic_branch_drop_rate
        act = L.get_activation(activation_type)
        if merge_method == "concat":
            self.audiovisual_layer = L.PositionwiseFeedForward(2 * input_size, hidden_units, dropout_rate, act)
        elif merge_method == "learned_ave":
            self.acoustic_pooling_proj = nn.Linear(input_size, 1)
            self.visual_pooling_proj = nn.Linear(input_size, 1)
            self.acoustic_weight_proj = nn.Linear(input_size, 1)
            self.visual_weight_proj = nn.Linear(input_size, 1)
            self.audiovisual_layer = L.PositionwiseFeedForward(input_size, hidden_units, dropout_rate, act)
        elif merge_method == "fixed_ave":
            self.audiovisual_layer = L.PositionwiseFeedForward(input_size, hidden_units, dropout_rate, act)
        else:
            raise ValueError(f"Unknow merge method: {merge_method}")
        self.norm_final = L.LayerNorm(output_size)

    def output_size(self):
        return self._output_size

    def forward(self, audio_pad, audio_masks, video_pad, video_masks, cache=None):
        if self.merge_method == "concat":
            av = self.audiovisual_layer(torch.cat([audio_pad, video_pad], dim=-1))
        elif self.merge_method == "learned_ave":                                                   # :137-196
            def pooled_weight(x, m, pool, wproj):
                score = _masked_time_softmax(pool(x).transpose(1, 2) / self.input_size ** 0.5, m)
                return wproj(torch.matmul(score, x).squeeze(1))
            if (self.training and self.acoustic_branch_drop_rate > 0
                    and torch.rand(1).item() < self.acoustic_branch_drop_rate):                    # :138-144
                self.acoustic_weight, self.visual_weight = 0.0, 1.0
            else:
                wa = pooled_weight(audio_pad, audio_masks, self.acoustic_pooling_proj, self.acoustic_weight_proj)
                wv = pooled_weight(video_pad, video_masks, self.visual_pooling_proj, self.visual_weight_proj)
                mw = torch.softmax(torch.cat([wa, wv], dim=-1), dim=-1).unsqueeze(-1).unsqueeze(-1)
                self.acoustic_weight, self.visual_weight = mw[:, 0], mw[:, 1]
            av = self.audiovisual_layer(self.acoustic_weight * audio_pad + self.visual_weight * video_pad)
        else:
            av = self.audiovisual_layer(self.acoustic_weight * audio_pad + (1.0 - self.acoustic_weight) * video_pad)
        av = self.norm_final(av)
        masks = torch.logical_or(audio_masks, video_masks)
        return av, masks.squeeze(1).sum(1)


# --------------------------------------------------------------------------------------------------
# AVSR model: src/models/avsr_espnet_model.py:46-685 (attention/CTC branch)
# --------------------------------------------------------------------------------------------------
class AVSRModelOracle(nn.Module):
    def __init__(self, vocab_size, token_list, specaug, normalize, acoustic_frontend, visual_frontend,
                 acoustic_embed, visual_embed, encoder, audiovisual_fusion, decoder, ctc, ctc_weight=0.5,
                 interctc_weight=0.0, ignore_id=-1, lsm_weight=0.0, length_normalized_loss=False, report_cer=True,
                 report_wer=True, sym_space="<space>", sym_blank="<blank>", sym_sos="<sos/eos>", sym_eos="<sos/eos>",
                 **_unused):
        super().__init__()
        self.blank_id = token_list.index(sym_blank) if sym_blank in token_list else 0
        self.sos = token_list.index(sym_sos) if sym_sos in token_list else vocab_size - 1
        self.eos = token_list.index(sym_eos) if sym_eos in token_list else vocab_size - 1
        self.vocab_size, self.ignore_id = vocab_size, ignore_id
        self.ctc_weight, self.interctc_weight = ctc_weight, interctc_weight
        self.token_list = list(token_list)
        self.specaug, self.normalize = specaug, normalize
        self.acoustic_frontend, self.visual_frontend = acoustic_frontend, visual_frontend
        self.acoustic_embed, self.visual_embed = acoustic_embed, visual_embed
        self.encoder, self.audiovisual_fusion = encoder, audiovisual_fusion
        if getattr(self.encoder, "interctc_use_conditioning", False):                  # avsr_espnet_model.py:119-124
            self.encoder.conditioning_layer = nn.Linear(vocab_size, self.encoder.output_size())
        self.decoder = decoder if ctc_weight < 1.0 else None
        self.criterion_att = L.LabelSmoothingLoss(vocab_size, ignore_id, lsm_weight, length_normalized_loss)
        self.error_calculator = (L.ErrorCalculator(token_list, sym_space, sym_blank, report_cer, report_wer)
                                 if (report_cer or report_wer) else None)
        self.ctc = ctc if ctc_weight != 0.0 else None

    def audiovisual_alignment(self, a, am, v, vm):                                                 # :512-541
        pad = a.shape[1] - v.shape[1]
        if pad < 0:
            a = F.pad(a, (0, 0, 0, -pad, 0, 0), value=self.ignore_id)
            am = F.pad(am, (0, -pad), value=False)
        elif pad > 0:
            v = F.pad(v, (0, 0, 0, pad, 0, 0), value=self.ignore_id)
            vm = F.pad(vm, (0, pad), value=False)
        return a, am, v, vm

    def encode(self, audio, audio_lengths, video, video_lengths):                                  # :383-488
        audio = audio[:, : audio_lengths.max()]
        video = video[:, : video_lengths.max()]
        af, al = (self.acoustic_frontend(audio, audio_lengths) if self.acoustic_frontend is not None else (audio, audio_lengths))
        vf, vl = (self.visual_frontend(video, video_lengths) if self.visual_frontend is not None else (video, video_lengths))
        if self.normalize is not None:
            af, al = self.normalize(af, al)
        af, am = self.acoustic_embed.apply_embed_layer(af, al)
        vf, vm = self.visual_embed.apply_embed_layer(vf, vl)
        af, am, vf, vm = self.audiovisual_alignment(af, am, vf, vm)
        af = self.acoustic_embed.apply_pos_enc(af)
        vf = self.visual_embed.apply_pos_enc(vf)
        a, am, v, vm, _ = self.encoder(af, am, vf, vm, ctc=self.ctc if self.encoder.interctc_use_conditioning else None,
                                       audiovisual_fusion=self.audiovisual_fusion if len(self.encoder.interctc_layer_idx) > 0 else None)
        inter = None
        if isinstance(a, tuple):
            a, inter = a
        out, olens = self.audiovisual_fusion(a, am, v, vm)
        if inter is not None:
            return (out, inter), olens
        return out, olens

    def forward(self, audio, audio_lengths, video, video_lengths, text, text_lengths, **kwargs):   # :211-367
        b = audio.shape[0]
        text = text.clone()
        text[text == -1] = self.ignore_id
        text = text[:, : text_lengths.max()]
        enc, enc_lens = self.encode(audio, audio_lengths, video, video_lengths)
        inter = None
        if isinstance(enc, tuple):
            enc, inter = enc
        stats: Dict[str, object] = {}
        loss_ctc = loss_att = None
        if self.ctc_weight != 0.0:
            loss_ctc = self.ctc(enc, enc_lens, text, text_lengths)
            cer_ctc = None
            if not self.training and self.error_calculator is not None:
                cer_ctc = self.error_calculator(self.ctc.argmax(enc).data.cpu(), text.cpu(), is_ctc=True)
            stats["loss_ctc"], stats["cer_ctc"] = loss_ctc.detach(), cer_ctc
        if self.interctc_weight != 0.0 and inter is not None:                          # avsr_espnet_model.py:271-315
            li = 0.0
            for idx, o in inter:
                l = self.ctc(o, enc_lens, text, text_lengths)
                stats[f"loss_interctc_layer{idx}"] = l.detach()
                li = li + l
            loss_ctc = (1 - self.interctc_weight) * loss_ctc + self.interctc_weight * li / len(inter)
        acc = cer = wer = None
        if self.ctc_weight != 1.0:
            ys_in, ys_out = L.add_sos_eos(text, self.sos, self.eos, self.ignore_id)
            dec_out, _ = self.decoder(enc, enc_lens, ys_in, text_lengths + 1)
            loss_att = self.criterion_att(dec_out, ys_out)
            acc = L.th_accuracy(dec_out.view(-1, self.vocab_size), ys_out, ignore_label=self.ignore_id)
            if not self.training and self.error_calculator is not None:
                cer, wer = self.error_calculator(dec_out.argmax(dim=-1).cpu(), text.cpu())
        if self.ctc_weight == 0.0:
            loss = loss_att
        elif self.ctc_weight == 1.0:
            loss = loss_ctc
        else:
            loss = self.ctc_weight * loss_ctc + (1 - self.ctc_weight) * loss_att
        stats.update(loss_att=None if loss_att is None else loss_att.detach(), acc=acc, cer=cer, wer=wer, loss=loss.detach())
        return L.force_gatherable((loss, stats, b), loss.device)

    @torch.no_grad()
    def ctc_greedy(self, audio, audio_lengths, video, video_lengths) -> List[List[int]]:
        enc, olens = self.encode(audio, audio_lengths, video, video_lengths)
        ids = self.ctc.argmax(enc)
        return [L.ctc_greedy_collapse(ids[i], int(olens[i]), self.blank_id) for i in range(ids.size(0))]


def build_avsr_oracle(conf: dict, token_list: Sequence[str]) -> AVSRModelOracle:
    """``AVSRTask.build_model`` (src/tasks/avsr.py:506-718) for the tailored / conventional recipes."""
    vocab = len(token_list)
    if conf.get("acoustic_input_size") is None:
        afront = L.DefaultFrontend(**(conf.get("acoustic_frontend_conf") or {}))
        ain = afront.output_size()
    else:
        afront, ain = None, conf["acoustic_input_size"]
    if conf.get("visual_input_size") is None:
        assert conf["visual_frontend"] == "conv3dresnet18"
        vfront = Conv3dResNet18Oracle(**(conf.get("visual_frontend_conf") or {}))
        vin = vfront.output_size()
    else:
        vfront, vin = None, conf["visual_input_size"]
    if conf.get("specaug") is not None:
        raise ValueError("oracle parity runs use specaug: null (stochastic, train-only)")
    normalize = L.UtteranceMVN(**(conf.get("normalize_conf") or {})) if conf.get("normalize") == "utterance_mvn" else None
    d = conf["encoder_conf"]["output_size"] if "output_size" in conf["encoder_conf"] else 256
    aemb = DefaultEmbeddingOracle(input_size=ain, output_size=d, **conf["acoustic_embed_conf"])
    vemb = DefaultEmbeddingOracle(input_size=vin, output_size=d, **conf["visual_embed_conf"])
    if conf["encoder"] == "tailored":
        enc = TailoredEncoderOracle(embed_pos_enc_layer_type=aemb._pos_enc_layer_type, embed_rel_pos_type=aemb._rel_pos_type,
                                    **conf["encoder_conf"])
    elif conf["encoder"] == "conventional":
        enc = ConventionalEncoderOracle(input_size=aemb.output_size(), embed_pos_enc_layer_type=aemb._pos_enc_layer_type,
                                        embed_rel_pos_type=aemb._rel_pos_type, **conf["encoder_conf"])
    else:
        raise ValueError(conf["encoder"])
    fusion = AdaptiveFusionOracle(input_size=enc.output_size(), **conf["audiovisual_fusion_conf"])
    decoder = None
    if conf.get("decoder") is not None:
        decoder = L.TransformerDecoder(vocab_size=vocab, encoder_output_size=fusion.output_size(), **conf["decoder_conf"])
    ctc = CTCOracle(odim=vocab, encoder_output_size=fusion.output_size(), **conf["ctc_conf"])
    return AVSRModelOracle(vocab_size=vocab, token_list=list(token_list), specaug=None, normalize=normalize,
                           acoustic_frontend=afront, visual_frontend=vfront, acoustic_embed=aemb, visual_embed=vemb,
                           encoder=enc, audiovisual_fusion=fusion, decoder=decoder, ctc=ctc, **conf["model_conf"])
