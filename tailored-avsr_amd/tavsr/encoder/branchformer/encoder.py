"""``MyBranchformerEncoder`` - drop-in for src/encoder/branchformer/encoder.py:52-412.

Constructor keywords, ``forward(xs_pad, ilens, prev_states, ctc, max_layer)`` and state_dict keys
follow the reference; supported option subset = what the shipped configs use (rel_pos "latest",
``input_layer`` in {"conv2d", "linear", "conv1d" / "conv3dresnet18", None}); anything else raises ``ValueError`` like the reference does
for unknown strings.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from ... import functional as F_
from ... import functional_av as FA
from ... import dp, ops
from ...layers import (Conv2dSubsampling, ConvolutionalGatingMLP, LayerNorm, PositionwiseFeedForward,
                       RelPositionalEncoding, RelPositionMultiHeadedAttention, TooShortUttError, check_short_utt,
                       make_pad_mask)
from .encoder_layer import MyBranchformerEncoderLayer


class MyBranchformerEncoder(torch.nn.Module):
    def __init__(self, input_size=256, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 cgmlp_linear_units=2048, cgmlp_conv_kernel=31, cgmlp_weight=0.5, dropout_rate=0.1,
                 positional_dropout_rate=0.1, attention_dropout_rate=0.1, attn_branch_drop_rate=0.0,
                 input_layer="conv3dresnet18", rel_pos_type="latest", pos_enc_layer_type="rel_pos",
                 attention_layer_type="rel_selfattn", positionwise_layer_type="linear", ffn_activation_type="relu",
                 merge_method="learned_ave", gate_activation="identity", ignore_id=-1, use_attn=True, use_cgmlp=True,
                 macaron=True, zero_triu=False, normalize_before=True, use_linear_after_conv=False,
                 interctc_use_conditioning: bool = False, interctc_layer_idx: List[int] = [],
                 stochastic_depth_rate=0.0, max_pos_emb_len: int = 5000):
        super().__init__()
        self._output_size = output_size
        if rel_pos_type != "latest":
            raise ValueError("unknown rel_pos_type: " + rel_pos_type if rel_pos_type != "legacy"
                             else "legacy rel_pos is not on the HIP path")
        if pos_enc_layer_type != "rel_pos":
            raise ValueError("unknown pos_enc_layer: " + pos_enc_layer_type)
        if attention_layer_type != "rel_selfattn":
            raise ValueError("unknown encoder_attn_layer: " + attention_layer_type)
        if positionwise_layer_type != "linear":
            raise ValueError("Support only linear.")
        pos = lambda: RelPositionalEncoding(output_size, positional_dropout_rate, max_pos_emb_len)
        if input_layer == "conv2d":
            self.embed = Conv2dSubsampling(input_size, output_size, dropout_rate, pos())
        elif input_layer == "linear":             # the VSR recipes: lip features [B, T, 512] (encoder.py:123-129)
            self.embed = torch.nn.Sequential(torch.nn.Linear(input_size, output_size), torch.nn.LayerNorm(output_size),
                                             torch.nn.Dropout(dropout_rate), pos())
        elif input_layer in ("conv1d", "conv3dresnet18"):      # the constructor default (encoder.py:130-134)
            self.embed = torch.nn.Sequential(torch.nn.Linear(512, output_size),
                                             RelPositionalEncoding(output_size, positional_dropout_rate))
        elif input_layer is None:
            self.embed = None
        else:
            raise ValueError("unknown input_layer: " + str(input_layer))
        self.normalize_before = normalize_before

        def per_block(v, what):
            v = [v] * num_blocks if isinstance(v, float) else list(v)
            if len(v) != num_blocks:
                raise ValueError(f"Length of {what} ({len(v)}) should be equal to num_blocks ({num_blocks})")
            return v

        sdr = per_block(stochastic_depth_rate, "stochastic_depth_rate")
        cgw = per_block(cgmlp_weight, "cgmlp_weight")
        abd = per_block(attn_branch_drop_rate, "attn_branch_drop_rate")
        ffn = lambda: PositionwiseFeedForward(output_size, linear_units, dropout_rate, ffn_activation_type)
        self.encoders = torch.nn.ModuleList([
            MyBranchformerEncoderLayer(
                output_size,
                RelPositionMultiHeadedAttention(attention_heads, output_size, attention_dropout_rate, zero_triu)
                if use_attn else None,
                ConvolutionalGatingMLP(output_size, cgmlp_linear_units, cgmlp_conv_kernel, dropout_rate,
                                       use_linear_after_conv, gate_activation) if use_cgmlp else None,
                ffn() if macaron else None, ffn(), dropout_rate, merge_method, cgw[i], abd[i], sdr[i])
            for i in range(num_blocks)])
        if self.normalize_before:
            self.after_norm = LayerNorm(output_size)
        self.interctc_layer_idx = list(interctc_layer_idx)
        if len(self.interctc_layer_idx) > 0:
            assert 0 < min(self.interctc_layer_idx) and max(self.interctc_layer_idx) < num_blocks
        self.interctc_use_conditioning = interctc_use_conditioning
        self.conditioning_layer = None

    def output_size(self) -> int:
        return self._output_size

    def _linear_embed(self, x):
        """Sequential(Linear[, torch LayerNorm(eps 1e-5), Dropout], RelPositionalEncoding) -> (x * sqrt(d), pos_emb)."""
        lin, pe = self.embed[0], self.embed[-1]
        x = F_.LinearFn.apply(x, lin.weight, lin.bias, 1.0)
        if len(self.embed) == 4:
            ln, drop = self.embed[1], self.embed[2]
            x = F_.LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)
            if self.training and drop.p > 0:
                x = F_.DropoutFn.apply(x, drop.p)
        x, pos = FA.ScaleFn.apply(x, pe.xscale), pe.pos_emb(x.size(1), x.device)
        if self.training and pe.dropout_rate > 0:           # RelPositionalEncoding: dropout(x), dropout(pos_emb)
            x = F_.DropoutFn.apply(x, pe.dropout_rate)
            pos = ops.dropout(pos, pe.dropout_rate)[0]
        return x, pos

    def forward(self, xs_pad: torch.Tensor, ilens: torch.Tensor, prev_states: torch.Tensor = None, ctc=None,
                max_layer: int = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        masks = (~make_pad_mask(ilens, xs_pad.size(1))[:, None, :]).to(xs_pad.device)
        if isinstance(self.embed, Conv2dSubsampling):
            short_status, limit_size = check_short_utt(self.embed, xs_pad.size(1))
            if short_status:
                raise TooShortUttError(
                    f"has {xs_pad.size(1)} frames and is too short for subsampling "
                    + f"(it needs more than {limit_size} frames), return empty results", xs_pad.size(1), limit_size)
            xs_pad, masks = self.embed(xs_pad, masks)
        elif isinstance(self.embed, torch.nn.Sequential):
            xs_pad = self._linear_embed(xs_pad)
        elif not isinstance(xs_pad, tuple):
            raise ValueError("input_layer=None expects (x, pos_emb) from an external embedding (AV encoders)")
        lens = masks.squeeze(1).sum(1).to(torch.int64)
        intermediate_outs = []
        for layer_idx, encoder_layer in enumerate(self.encoders):
            xs_pad, masks = encoder_layer(xs_pad, masks, lens=lens)
            if layer_idx + 1 == len(self.encoders) // 2 and isinstance(xs_pad, tuple):
                # where a data-parallel step may split its backward pass (tavsr.dp.TwoPhaseBackward; identity otherwise)
                xs_pad = (dp.cut(xs_pad[0]),) + tuple(xs_pad[1:])
            if (len(self.interctc_layer_idx) == 0 and max_layer is not None
                    and 0 <= max_layer < len(self.encoders) and layer_idx >= max_layer):
                break
            if layer_idx + 1 in self.interctc_layer_idx:
                encoder_out = xs_pad[0]
                if self.normalize_before:
                    encoder_out = self.after_norm(encoder_out)
                intermediate_outs.append((layer_idx + 1, encoder_out))
                if self.interctc_use_conditioning:      # x + conditioning_layer(ctc.softmax(encoder_out))  (:389-401)
                    x, pos_emb = xs_pad
                    x = F_.InterCTCConditionFn.apply(x, encoder_out, ctc.ctc_lo.weight, ctc.ctc_lo.bias,
                                                     self.conditioning_layer.weight, self.conditioning_layer.bias)
                    xs_pad = (x, pos_emb)
        xs_pad = xs_pad[0]
        if self.normalize_before:
            xs_pad = self.after_norm(xs_pad)
        olens = lens
        if len(intermediate_outs) > 0:
            return (xs_pad, intermediate_outs), olens, None
        return xs_pad, olens, None
