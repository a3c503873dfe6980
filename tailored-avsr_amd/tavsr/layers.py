"""Parameter-holding counterparts of the espnet==202402 leaves the reference composes.

Same class names, constructor arguments, attribute names and therefore ``state_dict`` keys as the
espnet classes, so reference checkpoints load unchanged.  The arithmetic
is NOT here: composite modules gather these parameters and run the hand-written HIP path
(``tavsr.functional``).  Where a leaf is useful on its own its ``forward`` runs the same kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F_
from . import ops


class LayerNorm(nn.Module):
    """espnet LayerNorm: eps = 1e-12 (torch default would be 1e-5)."""

    def __init__(self, nout: int, dim: int = -1, eps: float = 1e-12):
        super().__init__()
        if dim != -1:
            raise ValueError("only dim=-1 is used on the hot path")
        self.weight = nn.Parameter(torch.ones(nout))
        self.bias = nn.Parameter(torch.zeros(nout))
        self.eps = eps

    def forward(self, x):
        return F_.LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class Linear(nn.Linear):
    """torch.nn.Linear parameters/initialisation, forward on the MFMA GEMM."""

    def forward(self, x):
        return F_.LinearFn.apply(x, self.weight, self.bias, 1.0)


def get_activation_name(act) -> str:
    if isinstance(act, str):
        if act not in ("relu", "swish"):
            raise ValueError(f"unsupported activation on the HIP path: {act}")
        return act
    raise ValueError("activation must be given by name")


class PositionwiseFeedForward(nn.Module):
    def __init__(self, idim, hidden_units, dropout_rate, activation="relu"):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden_units)
        self.w_2 = nn.Linear(hidden_units, idim)
        self.dropout_rate = dropout_rate
        self.activation = get_activation_name(activation)


class MultiHeadedAttention(nn.Module):
    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.dropout_rate = dropout_rate


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    def __init__(self, n_head, n_feat, dropout_rate, zero_triu=False):
        super().__init__(n_head, n_feat, dropout_rate)
        if zero_triu:
            raise ValueError("zero_triu=True is not used by any shipped config")
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)


class ConvolutionalSpatialGatingUnit(nn.Module):
    def __init__(self, size, kernel_size, dropout_rate, use_linear_after_conv, gate_activation):
        super().__init__()
        if use_linear_after_conv or gate_activation != "identity":
            raise ValueError("HIP path covers use_linear_after_conv=False, gate_activation='identity' (all shipped configs)")
        n = size // 2
        self.norm = LayerNorm(n)
        self.conv = nn.Conv1d(n, n, kernel_size, 1, (kernel_size - 1) // 2, groups=n)
        self.linear = None


class ConvolutionalGatingMLP(nn.Module):
    def __init__(self, size, linear_units, kernel_size, dropout_rate, use_linear_after_conv, gate_activation):
        super().__init__()
        self.channel_proj1 = nn.Sequential(nn.Linear(size, linear_units), nn.GELU())
        self.csgu = ConvolutionalSpatialGatingUnit(linear_units, kernel_size, dropout_rate, use_linear_after_conv,
                                                   gate_activation)
        self.channel_proj2 = nn.Linear(linear_units // 2, size)
        self.dropout_rate = dropout_rate


def _rel_pe_table(length: int, d_model: int) -> torch.Tensor:
    """espnet RelPositionalEncoding.extend_pe: (1, 2*length-1, d); centre row = offset 0, rows above it
    positive offsets.  A constant, computed once on the host exactly as the reference does."""
    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pp, pn = torch.zeros(length, d_model), torch.zeros(length, d_model)
    pp[:, 0::2], pp[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    pn[:, 0::2], pn[:, 1::2] = torch.sin(-1 * pos * div), torch.cos(-1 * pos * div)
    return torch.cat([torch.flip(pp, [0]).unsqueeze(0), pn[1:].unsqueeze(0)], dim=1)


class RelPositionalEncoding(nn.Module):
    """forward(x) -> (x * sqrt(d), pe[:, c-T+1 : c+T]); the scale is normally fused into the producer GEMM."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model, self.xscale, self.dropout_rate = d_model, math.sqrt(d_model), dropout_rate
        self.max_len = max_len
        self.pe = None
        self._cache = {}

    def pos_emb(self, T: int, device) -> torch.Tensor:
        key = (T, str(device))
        if key not in self._cache:
            if self.pe is None or self.pe.size(1) < 2 * T - 1:
                self.pe = _rel_pe_table(max(T, self.max_len), self.d_model)
            c = self.pe.size(1) // 2
            self._cache[key] = self.pe[:, c - T + 1: c + T].contiguous().to(device)
        return self._cache[key]

    def forward(self, x):
        return ops.axpby(x.contiguous(), None, self.xscale, 0.0), self.pos_emb(x.size(1), x.device)


class PositionalEncoding(nn.Module):
    """Absolute sinusoid table of the decoder embed; applied inside tavsr_embed_pe."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model, self.xscale, self.dropout_rate = d_model, math.sqrt(d_model), dropout_rate
        pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self._pe_host = pe
        self._cache = {}

    def table(self, L: int, device) -> torch.Tensor:
        key = (L, str(device))
        if key not in self._cache:
            self._cache[key] = self._pe_host[:L].contiguous().to(device)
        return self._cache[key]


class TooShortUttError(Exception):
    def __init__(self, message, actual_size, limit):
        super().__init__(message)
        self.actual_size, self.limit = actual_size, limit


class Conv2dSubsampling(nn.Module):
    """espnet Conv2dSubsampling(idim, odim, dropout_rate, pos_enc): time/4, keys conv.0, conv.2, out.0."""

    def __init__(self, idim, odim, dropout_rate, pos_enc=None):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, odim, 3, 2), nn.ReLU(), nn.Conv2d(odim, odim, 3, 2), nn.ReLU())
        if pos_enc is None:
            raise ValueError("the HIP path is built with a RelPositionalEncoding (rel_pos configs)")
        self.out = nn.Sequential(nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim), pos_enc)

    def forward(self, x, x_mask):
        pe: RelPositionalEncoding = self.out[1]
        y = F_.Conv2dSubsamplingFn.apply(x, self.conv[0].weight, self.conv[0].bias, self.conv[2].weight,
                                         self.conv[2].bias, self.out[0].weight, self.out[0].bias, pe.xscale)
        pos = pe.pos_emb(y.size(1), y.device)
        if self.training and pe.dropout_rate > 0:    # RelPositionalEncoding: dropout(x), dropout(pos_emb)
            y = F_.DropoutFn.apply(y, pe.dropout_rate)
            pos = ops.dropout(pos, pe.dropout_rate)[0]
        if x_mask is None:
            return (y, pos), None
        return (y, pos), x_mask[:, :, :-2:2][:, :, :-2:2]


def check_short_utt(ins, size):
    if isinstance(ins, Conv2dSubsampling) and size < 7:
        return True, 7
    return False, -1


def make_pad_mask(lengths: torch.Tensor, maxlen=None) -> torch.Tensor:
    """bool (B, maxlen), True at padding (espnet nets_utils.make_pad_mask); tiny integer glue."""
    maxlen = int(lengths.max()) if maxlen is None else maxlen
    return torch.arange(maxlen, device=lengths.device)[None, :] >= lengths[:, None]
