"""``TransformerLM`` - parameter holder with espnet2.lm.transformer_lm.TransformerLM's constructor and state_dict keys
(configs/LM/lm-english.yaml: pos_enc null, embed 128, att 512, 8 heads, 2048 units, 16 layers), used as the ``lm``
scorer of the beam search (src/inference/avsr_inference.py:155-170).  Its one-token scoring step runs inside
``tavsr.inference.beam_search`` on the HIP kernels; ``forward`` (whole sequences, no cache) is the same arithmetic in
teacher-forced form and exists for checking."""
from __future__ import annotations


import torch

from .. import ops
from ..layers import LayerNorm, MultiHeadedAttention, PositionwiseFeedForward

EPS = 1e-12


class _EncoderLayer(torch.nn.Module):
    def __init__(self, size, heads, units, dropout_rate):
        super().__init__()
        self.self_attn = MultiHeadedAttention(heads, size, 0.0)
        self.feed_forward = PositionwiseFeedForward(size, units, dropout_rate, "relu")
        self.norm1, self.norm2 = LayerNorm(size), LayerNorm(size)


class _Encoder(torch.nn.Module):
    def __init__(self, idim, attention_dim, attention_heads, linear_units, num_blocks, dropout_rate):
        super().__init__()
        # espnet Encoder(input_layer="linear"): Linear, LayerNorm, Dropout, ReLU, pos_enc (identity for pos_enc: null)
        self.embed = torch.nn.Sequential(torch.nn.Linear(idim, attention_dim), LayerNorm(attention_dim),
                                         torch.nn.Dropout(dropout_rate), torch.nn.ReLU(), torch.nn.Sequential())
        self.encoders = torch.nn.ModuleList([_EncoderLayer(attention_dim, attention_heads, linear_units, dropout_rate)
                                             for _ in range(num_blocks)])
        self.after_norm = LayerNorm(attention_dim)


class TransformerLM(torch.nn.Module):
    def __init__(self, vocab_size: int, pos_enc: str = None, embed_unit: int = 128, att_unit: int = 256, head: int = 2,
                 unit: int = 1024, layer: int = 4, dropout_rate: float = 0.5):
        super().__init__()
        if pos_enc is not None:
            raise ValueError("the shipped LM recipe uses pos_enc: null (configs/LM/lm-english.yaml)")
        self.embed = torch.nn.Embedding(vocab_size, embed_unit)
        self.encoder = _Encoder(embed_unit, att_unit, head, unit, layer, dropout_rate)
        self.decoder = torch.nn.Linear(att_unit, vocab_size)
        self.heads, self.att_unit = head, att_unit

    @torch.no_grad()
    def forward(self, input: torch.Tensor, hidden=None):
        """input (B, L) int64 (no 0 tokens: espnet masks keys equal to 0) -> logits (B, L, V); eval only."""
        from ..functional import _SelfAttnCore
        if self.training:
            raise NotImplementedError("LM training is out of scope (lm_main.py is broken as shipped, SURVEY 2 #14)")
        B, Lq = input.shape
        D, H = self.att_unit, self.heads
        dk = D // H
        M = B * Lq
        lens = torch.full((B,), Lq, dtype=torch.int64, device=input.device)
        e = self.embed.weight[input.reshape(-1)]                     # embedding row gather (index plumbing)
        emb = self.encoder.embed
        h = ops.linear(e.contiguous(), emb[0].weight, emb[0].bias)
        h = ops.layernorm_fwd(h, emb[1].weight, emb[1].bias, EPS)[0]
        ops.act_(h, "relu")
        for layer in self.encoder.encoders:
            a = layer.self_attn
            n1 = ops.layernorm_fwd(h, layer.norm1.weight, layer.norm1.bias, EPS)[0]
            qkv = ops.empty(M, 3 * D, like=h)
            for j, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                ops.linear(n1, lin.weight, lin.bias, out=qkv, out_off=j * D, ldc=3 * D)
            cx, _, _ = _SelfAttnCore.fwd(qkv, 3 * D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, B, Lq, Lq, H, dk, lens, True)
            h = ops.linear(cx, a.linear_out.weight, a.linear_out.bias, res=h)
            n2 = ops.layernorm_fwd(h, layer.norm2.weight, layer.norm2.bias, EPS)[0]
            f = layer.feed_forward
            t = ops.linear(n2, f.w_1.weight, f.w_1.bias, act="relu")
            h = ops.linear(t, f.w_2.weight, f.w_2.bias, res=h)
        y = ops.layernorm_fwd(h, self.encoder.after_norm.weight, self.encoder.after_norm.bias, EPS)[0]
        return ops.linear(y, self.decoder.weight, self.decoder.bias).view(B, Lq, -1), None
