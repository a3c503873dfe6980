"""``TransformerDecoder`` - counterpart of espnet2.asr.decoder.transformer_decoder.TransformerDecoder as
instantiated by the reference (src/tasks/asr.py:176-194, configs ``decoder: transformer``), teacher-forced
forward only (``espnet_model.py:557-560``); incremental ``batch_score`` belongs to the beam-search row."""
from __future__ import annotations

import torch

from .. import functional as F_
from ..layers import LayerNorm, MultiHeadedAttention, PositionalEncoding, PositionwiseFeedForward


class DecoderLayer(torch.nn.Module):
    def __init__(self, size, self_attn, src_attn, feed_forward, dropout_rate, normalize_before=True, concat_after=False):
        super().__init__()
        if not normalize_before or concat_after:
            raise ValueError("HIP path covers normalize_before=True, concat_after=False (espnet2 defaults)")
        self.size = size
        self.self_attn, self.src_attn, self.feed_forward = self_attn, src_attn, feed_forward
        self.norm1, self.norm2, self.norm3 = LayerNorm(size), LayerNorm(size), LayerNorm(size)
        self.dropout_rate = dropout_rate


class TransformerDecoder(torch.nn.Module):
    def __init__(self, vocab_size: int, encoder_output_size: int, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 self_attention_dropout_rate: float = 0.0, src_attention_dropout_rate: float = 0.0,
                 input_layer: str = "embed", use_output_layer: bool = True, normalize_before: bool = True,
                 concat_after: bool = False, layer_drop_rate: float = 0.0):
        super().__init__()
        if input_layer != "embed" or not use_output_layer or not normalize_before:
            raise ValueError("HIP path covers input_layer='embed', use_output_layer, normalize_before (shipped configs)")
        d = encoder_output_size
        self.embed = torch.nn.Sequential(torch.nn.Embedding(vocab_size, d), PositionalEncoding(d, positional_dropout_rate))
        self.after_norm = LayerNorm(d)
        self.output_layer = torch.nn.Linear(d, vocab_size)
        self.decoders = torch.nn.ModuleList([
            DecoderLayer(d, MultiHeadedAttention(attention_heads, d, self_attention_dropout_rate),
                         MultiHeadedAttention(attention_heads, d, src_attention_dropout_rate),
                         PositionwiseFeedForward(d, linear_units, dropout_rate, "relu"), dropout_rate,
                         normalize_before, concat_after)
            for _ in range(num_blocks)])
        self.heads, self.num_blocks = attention_heads, num_blocks
        self._rates = (dropout_rate, positional_dropout_rate, self_attention_dropout_rate, src_attention_dropout_rate)

    def _params(self):
        cached = self.__dict__.get("_tavsr_pcache")                        # Parameter identities never change: look up once
        if cached is not None:
            return cached
        P = [self.embed[0].weight]
        for layer in self.decoders:
            sd = dict(layer.named_parameters())
            P += [sd[n] for n in F_.DEC_LAYER_PARAM_NAMES]
        P = self.__dict__["_tavsr_pcache"] = P + [self.after_norm.weight, self.after_norm.bias, self.output_layer.weight,
                                                   self.output_layer.bias]
        return P

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens):
        """hs_pad (B,T,D), hlens (B), ys_in_pad (B,L) int64, ys_in_lens (B) -> (logits (B,L,V), olens)."""
        pe = self.embed[1].table(ys_in_pad.size(1), hs_pad.device)
        cfg = dict(heads=self.heads, num_blocks=self.num_blocks)
        if self.training:    # (dropout_rate, positional, self-attention, source-attention) of the espnet2 decoder
            cfg.update(p=self._rates[0], p_pos=self._rates[1], p_self=self._rates[2], p_src=self._rates[3])
        logits = F_.grad_apply(F_.TransformerDecoderFn, hs_pad, hlens.to(torch.int64), ys_in_pad.to(torch.int64),
                                               ys_in_lens.to(torch.int64), pe, cfg, *self._params())
        return logits, ys_in_lens
