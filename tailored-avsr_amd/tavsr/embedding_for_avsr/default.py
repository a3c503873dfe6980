"""``DefaultEmbeddingLayerForAVSR`` - drop-in for src/embedding_for_avsr/default.py:22-162: the per-modality embedding
(audio: Conv2dSubsamplingWOPosEnc k3s2 x2; video: Linear + torch LayerNorm(eps 1e-5) + Dropout) kept apart from the
positional encoding so that the two streams can be length-aligned in between."""
from __future__ import annotations

import math
from typing import Tuple, Union

import torch

from .. import functional as F_
from .. import functional_av as FA
from .. import ops
from ..layers import RelPositionalEncoding, make_pad_mask


class Conv2dSubsamplingWOPosEnc(torch.nn.Module):
    """espnet Conv2dSubsamplingWOPosEnc(idim, odim, dropout_rate, kernels, strides): keys conv.0, conv.2, out."""

    def __init__(self, idim, odim, dropout_rate, kernels, strides):
        super().__init__()
        if list(kernels) != [3, 3] or list(strides) != [2, 2]:
            raise ValueError("the HIP path covers kernels=[3,3], strides=[2,2] (default.py:63-70)")
        self.conv = torch.nn.Sequential(torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(), torch.nn.Conv2d(odim, odim, 3, 2),
                                        torch.nn.ReLU())
        olen = idim
        for k, s in zip(kernels, strides):
            olen = math.floor((olen - k) / s + 1)
        self.out = torch.nn.Linear(odim * olen, odim)
        self.kernels, self.strides = kernels, strides

    def forward(self, x, x_mask):
        y = F_.Conv2dSubsamplingFn.apply(x, self.conv[0].weight, self.conv[0].bias, self.conv[2].weight, self.conv[2].bias,
                                         self.out.weight, self.out.bias, 1.0)
        if x_mask is None:
            return y, None
        for k, s in zip(self.kernels, self.strides):
            x_mask = x_mask[:, :, : -k + 1: s]
        return y, x_mask


class DefaultEmbeddingLayerForAVSR(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, pos_enc_layer_type: str = "rel_pos", rel_pos_type: str = "latest",
                 input_layer: str = "conv2d", dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 max_pos_emb_len: int = 5000):
        super().__init__()
        self._output_size, self._rel_pos_type, self._pos_enc_layer_type = output_size, rel_pos_type, pos_enc_layer_type
        self.dropout_rate, self.positional_dropout_rate = dropout_rate, positional_dropout_rate
        if input_layer == "linear":
            self.embed = torch.nn.Sequential(torch.nn.Linear(input_size, output_size), torch.nn.LayerNorm(output_size),
                                             torch.nn.Dropout(dropout_rate))
        elif input_layer == "conv2d":
            self.embed = Conv2dSubsamplingWOPosEnc(input_size, output_size, dropout_rate, kernels=[3, 3], strides=[2, 2])
        else:
            raise ValueError("unknown input_layer: " + str(input_layer))
        if rel_pos_type != "latest":
            raise ValueError("unknown rel_pos_type: " + rel_pos_type)
        if pos_enc_layer_type != "rel_pos":
            raise ValueError("unknown pos_enc_layer: " + pos_enc_layer_type)
        self.pos_enc = RelPositionalEncoding(output_size, positional_dropout_rate, max_pos_emb_len)

    def output_size(self) -> int:
        return self._output_size

    def apply_embed_layer(self, xs_pad: torch.Tensor, ilens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        masks = (~make_pad_mask(ilens, xs_pad.size(1))[:, None, :]).to(xs_pad.device)
        if isinstance(self.embed, Conv2dSubsamplingWOPosEnc):
            xs_pad, masks = self.embed(xs_pad, masks)
        else:
            lin, ln = self.embed[0], self.embed[1]
            xs_pad = F_.LayerNormFn.apply(F_.LinearFn.apply(xs_pad, lin.weight, lin.bias, 1.0), ln.weight, ln.bias, ln.eps)
            if self.training and self.dropout_rate > 0:          # Sequential(Linear, LayerNorm, Dropout)  (default.py:57-62)
                xs_pad = F_.DropoutFn.apply(xs_pad, self.dropout_rate)
        return xs_pad, masks

    def apply_pos_enc(self, xs_pad: torch.Tensor):
        x, pos = FA.ScaleFn.apply(xs_pad, self.pos_enc.xscale), self.pos_enc.pos_emb(xs_pad.size(1), xs_pad.device)
        if self.training and self.positional_dropout_rate > 0:   # RelPositionalEncoding: dropout(x), dropout(pos_emb)
            x = F_.DropoutFn.apply(x, self.positional_dropout_rate)
            pos = ops.dropout(pos, self.positional_dropout_rate)[0]
        return x, pos

    def forward(self, xs_pad, ilens) -> Tuple[Union[Tuple, torch.Tensor], torch.Tensor]:
        xs_pad, masks = self.apply_embed_layer(xs_pad, ilens)
        return self.apply_pos_enc(xs_pad), masks
