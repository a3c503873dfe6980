"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU restatement, in plain eager PyTorch, of the third-party leaves the reference
composes.  The arithmetic of the hot path lives in the un-vendored dependency
``espnet==202402`` (``/root/reference/requirements.txt:1``) which is absent from
``/root/reference`` and cannot be installed here (no network).  This file restates the
published algorithm of each leaf and is anchored on the
reference's own call sites, cited per class.

PARITY STATUS: the *wiring* of the reference (its own composite modules) is pinned by
importing those modules over this file in the build container (``oracle/_shim.py`` +
``oracle/gen_golden.py`` -> ``tests/golden``).  The *leaf arithmetic* itself cannot be
diffed against the real espnet package here -> "parity unpinned" for the leaves; the
pieces whose arithmetic lives in the reference or in torch itself (``src/ctc/ctc.py``,
``src/frontend/conv3d_resnet18``) are pinned directly.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.
"""
from __future__ import annotations

import math
from itertools import groupby
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# A.1  espnet/nets/pytorch_backend/transformer/layer_norm.py
#      call sites: src/encoder/branchformer/encoder_layer.py:102-109, encoder.py:313
# --------------------------------------------------------------------------------------
class LayerNorm(nn.LayerNorm):
    """``torch.nn.LayerNorm(nout, eps=1e-12)``; ``dim=-1`` is plain ``F.layer_norm``."""

    def __init__(self, nout: int, dim: int = -1):
        super().__init__(nout, eps=1e-12)
        self.dim = dim

    def forward(self, x):
        if self.dim == -1:
            return super().forward(x)
        return super().forward(x.transpose(self.dim, -1)).transpose(self.dim, -1)


# --------------------------------------------------------------------------------------
# A.2  positionwise_feed_forward.py / nets_utils.get_activation
#      call sites: encoder.py:206-216, encoder_layer.py:193-194,313-314
# --------------------------------------------------------------------------------------
class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


def get_activation(act: str) -> nn.Module:
    table = {
        "hardtanh": nn.Hardtanh,
        "tanh": nn.Tanh,
        "relu": nn.ReLU,
        "selu": nn.SELU,
        "swish": Swish,
    }
    return table[act]()


class PositionwiseFeedForward(nn.Module):
    """``w_2(dropout(activation(w_1(x))))``."""

    def __init__(self, idim, hidden_units, dropout_rate, activation=None):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden_units)
        self.w_2 = nn.Linear(hidden_units, idim)
        self.dropout = nn.Dropout(dropout_rate)
        self.activation = activation if activation is not None else nn.ReLU()

    def forward(self, x):
        return self.w_2(self.dropout(self.activation(self.w_1(x))))


# --------------------------------------------------------------------------------------
# A.6  nets_utils.make_pad_mask / repeat.MultiSequential
# --------------------------------------------------------------------------------------
def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    """bool mask, True at padding.  (B, maxlen) or shaped like ``xs``."""
    if not isinstance(lengths, list):
        lengths = lengths.long().tolist()
    bs = len(lengths)
    if maxlen is None:
        maxlen = int(max(lengths)) if xs is None else xs.size(length_dim)
    seq = torch.arange(0, maxlen, dtype=torch.int64)
    seq = seq.unsqueeze(0).expand(bs, maxlen)
    lens = seq.new_tensor(lengths).unsqueeze(-1)
    mask = seq >= lens
    if xs is not None:
        if length_dim < 0:
            length_dim = xs.dim() + length_dim
        ind = tuple(slice(None) if i in (0, length_dim) else None for i in range(xs.dim()))
        mask = mask[ind].expand_as(xs).to(xs.device)
    return mask


class MultiSequential(nn.Sequential):
    def __init__(self, *args, layer_drop_rate=0.0):
        super().__init__(*args)
        self.layer_drop_rate = layer_drop_rate

    def forward(self, *args):
        _probs = torch.empty(len(self)).uniform_()
        for idx, m in enumerate(self):
            if not self.training or (_probs[idx] >= self.layer_drop_rate):
                args = m(*args)
        return args


def repeat(N, fn, layer_drop_rate=0.0):
    return MultiSequential(*[fn(n) for n in range(N)], layer_drop_rate=layer_drop_rate)


# --------------------------------------------------------------------------------------
# A.4  embedding.py
#      call sites: encoder.py:106-115,149-155; src/embedding_for_avsr/default.py:96-106
# --------------------------------------------------------------------------------------
class PositionalEncoding(nn.Module):
    """``x*sqrt(d) + pe[:, :T]`` then dropout (decoder embed)."""

    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.pe = None
        self.extend_pe(torch.tensor(0.0).expand(1, max_len))

    def extend_pe(self, x):
        if self.pe is not None and self.pe.size(1) >= x.size(1):
            if self.pe.dtype != x.dtype or self.pe.device != x.device:
                self.pe = self.pe.to(dtype=x.dtype, device=x.device)
            return
        pe = torch.zeros(x.size(1), self.d_model)
        position = torch.arange(0, x.size(1), dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(
            torch.arange(0, self.d_model, 2, dtype=torch.float32)
            * -(math.log(10000.0) / self.d_model)
        )
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.unsqueeze(0).to(device=x.device, dtype=x.dtype)

    def forward(self, x):
        self.extend_pe(x)
        x = x * self.xscale + self.pe[:, : x.size(1)]
        return self.dropout(x)


class RelPositionalEncoding(nn.Module):
    """"latest" relative positional encoding: returns ``(dropout(x*sqrt(d)), dropout(pe[:, c-T+1:c+T]))``."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.pe = None
        self.extend_pe(torch.tensor(0.0).expand(1, max_len))

    def extend_pe(self, x):
        if self.pe is not None and self.pe.size(1) >= x.size(1) * 2 - 1:
            if self.pe.dtype != x.dtype or self.pe.device != x.device:
                self.pe = self.pe.to(dtype=x.dtype, device=x.device)
            return
        pe_positive = torch.zeros(x.size(1), self.d_model)
        pe_negative = torch.zeros(x.size(1), self.d_model)
        position = torch.arange(0, x.size(1), dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(
            torch.arange(0, self.d_model, 2, dtype=torch.float32)
            * -(math.log(10000.0) / self.d_model)
        )
        pe_positive[:, 0::2] = torch.sin(position * div_term)
        pe_positive[:, 1::2] = torch.cos(position * div_term)
        pe_negative[:, 0::2] = torch.sin(-1 * position * div_term)
        pe_negative[:, 1::2] = torch.cos(-1 * position * div_term)
        pe_positive = torch.flip(pe_positive, [0]).unsqueeze(0)
        pe_negative = pe_negative[1:].unsqueeze(0)
        pe = torch.cat([pe_positive, pe_negative], dim=1)
        self.pe = pe.to(device=x.device, dtype=x.dtype)

    def forward(self, x):
        self.extend_pe(x)
        x = x * self.xscale
        c = self.pe.size(1) // 2
        pos_emb = self.pe[:, c - x.size(1) + 1 : c + x.size(1)]
        return self.dropout(x), self.dropout(pos_emb)


class ScaledPositionalEncoding(PositionalEncoding):
    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model=d_model, dropout_rate=dropout_rate, max_len=max_len)
        self.alpha = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        self.extend_pe(x)
        x = x + self.alpha * self.pe[:, : x.size(1)]
        return self.dropout(x)


# --------------------------------------------------------------------------------------
# A.3  attention.py
#      call sites: encoder.py:223-249; encoder_layer.py:208;
#      tailored/encoder_layer.py:192,239
# --------------------------------------------------------------------------------------
class MultiHeadedAttention(nn.Module):
    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k = n_feat // n_head
        self.h = n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout_rate)

    def forward_qkv(self, query, key, value):
        n_batch = query.size(0)
        q = self.linear_q(query).view(n_batch, -1, self.h, self.d_k)
        k = self.linear_k(key).view(n_batch, -1, self.h, self.d_k)
        v = self.linear_v(value).view(n_batch, -1, self.h, self.d_k)
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)

    def forward_attention(self, value, scores, mask):
        n_batch = value.size(0)
        if mask is not None:
            mask = mask.unsqueeze(1).eq(0)  # (batch, 1, *, time2)
            min_value = torch.finfo(scores.dtype).min
            scores = scores.masked_fill(mask, min_value)
            self.attn = torch.softmax(scores, dim=-1).masked_fill(mask, 0.0)
        else:
            self.attn = torch.softmax(scores, dim=-1)
        p_attn = self.dropout(self.attn)
        x = torch.matmul(p_attn, value)
        x = x.transpose(1, 2).contiguous().view(n_batch, -1, self.h * self.d_k)
        return self.linear_out(x)

    def forward(self, query, key, value, mask):
        q, k, v = self.forward_qkv(query, key, value)
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(self.d_k)
        return self.forward_attention(v, scores, mask)


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    """"latest" rel-pos attention (Transformer-XL style with rel_shift)."""

    def __init__(self, n_head, n_feat, dropout_rate, zero_triu=False):
        super().__init__(n_head, n_feat, dropout_rate)
        self.zero_triu = zero_triu
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def rel_shift(self, x):
        zero_pad = torch.zeros((*x.size()[:3], 1), device=x.device, dtype=x.dtype)
        x_padded = torch.cat([zero_pad, x], dim=-1)
        x_padded = x_padded.view(*x.size()[:2], x.size(3) + 1, x.size(2))
        x = x_padded[:, :, 1:].view_as(x)[:, :, :, : x.size(-1) // 2 + 1]
        if self.zero_triu:
            ones = torch.ones((x.size(2), x.size(3)), device=x.device)
            x = x * torch.tril(ones, x.size(3) - x.size(2))[None, None, :, :]
        return x

    def forward(self, query, key, value, pos_emb, mask):
        q, k, v = self.forward_qkv(query, key, value)
        q = q.transpose(1, 2)  # (batch, time1, head, d_k)
        n_batch_pos = pos_emb.size(0)
        p = self.linear_pos(pos_emb).view(n_batch_pos, -1, self.h, self.d_k)
        p = p.transpose(1, 2)  # (batch, head, 2*time1-1, d_k)
        q_with_bias_u = (q + self.pos_bias_u).transpose(1, 2)
        q_with_bias_v = (q + self.pos_bias_v).transpose(1, 2)
        matrix_ac = torch.matmul(q_with_bias_u, k.transpose(-2, -1))
        matrix_bd = torch.matmul(q_with_bias_v, p.transpose(-2, -1))
        matrix_bd = self.rel_shift(matrix_bd)
        scores = (matrix_ac + matrix_bd) / math.sqrt(self.d_k)
        return self.forward_attention(v, scores, mask)


# --------------------------------------------------------------------------------------
# A.5  espnet2/asr/layers/cgmlp.py
#      call sites: encoder.py:262-270; encoder_layer.py:220
# --------------------------------------------------------------------------------------
class ConvolutionalSpatialGatingUnit(nn.Module):
    def __init__(self, size, kernel_size, dropout_rate, use_linear_after_conv, gate_activation):
        super().__init__()
        n_channels = size // 2
        self.norm = LayerNorm(n_channels)
        self.conv = nn.Conv1d(
            n_channels, n_channels, kernel_size, 1, (kernel_size - 1) // 2, groups=n_channels
        )
        self.linear = nn.Linear(n_channels, n_channels) if use_linear_after_conv else None
        if gate_activation == "identity":
            self.act = nn.Identity()
        else:
            self.act = get_activation(gate_activation)
        self.dropout = nn.Dropout(dropout_rate)

    def espnet_initialization_fn(self):
        nn.init.normal_(self.conv.weight, std=1e-6)
        nn.init.ones_(self.conv.bias)
        if self.linear is not None:
            nn.init.normal_(self.linear.weight, std=1e-6)
            nn.init.ones_(self.linear.bias)

    def forward(self, x, gate_add=None):
        x_r, x_g = x.chunk(2, dim=-1)
        x_g = self.norm(x_g)
        x_g = self.conv(x_g.transpose(1, 2)).transpose(1, 2)
        if self.linear is not None:
            x_g = self.linear(x_g)
        if gate_add is not None:
            x_g = x_g + gate_add
        x_g = self.act(x_g)
        return self.dropout(x_r * x_g)


class ConvolutionalGatingMLP(nn.Module):
    def __init__(self, size, linear_units, kernel_size, dropout_rate, use_linear_after_conv, gate_activation):
        super().__init__()
        self.channel_proj1 = nn.Sequential(nn.Linear(size, linear_units), nn.GELU())
        self.csgu = ConvolutionalSpatialGatingUnit(
            size=linear_units,
            kernel_size=kernel_size,
            dropout_rate=dropout_rate,
            use_linear_after_conv=use_linear_after_conv,
            gate_activation=gate_activation,
        )
        self.channel_proj2 = nn.Linear(linear_units // 2, size)

    def forward(self, x, mask):
        if isinstance(x, tuple):
            xs_pad, pos_emb = x
        else:
            xs_pad, pos_emb = x, None
        xs_pad = self.channel_proj1(xs_pad)
        xs_pad = self.csgu(xs_pad)
        xs_pad = self.channel_proj2(xs_pad)
        return (xs_pad, pos_emb) if pos_emb is not None else xs_pad


class FastSelfAttention(nn.Module):
    """Name-only stand-in (``isinstance`` checks at encoder_layer.py:204); never built."""


# --------------------------------------------------------------------------------------
# A.7  subsampling.py / subsampling_without_posenc.py
#      call sites: encoder.py:149-155,347-364; src/embedding_for_avsr/default.py:63-70
# --------------------------------------------------------------------------------------
class TooShortUttError(Exception):
    def __init__(self, message, actual_size, limit):
        super().__init__(message)
        self.actual_size = actual_size
        self.limit = limit


class Conv2dSubsampling(nn.Module):
    def __init__(self, idim, odim, dropout_rate, pos_enc=None):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(1, odim, 3, 2), nn.ReLU(), nn.Conv2d(odim, odim, 3, 2), nn.ReLU()
        )
        self.out = nn.Sequential(
            nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim),
            pos_enc if pos_enc is not None else PositionalEncoding(odim, dropout_rate),
        )

    def forward(self, x, x_mask):
        x = x.unsqueeze(1)  # (b, c, t, f)
        x = self.conv(x)
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).contiguous().view(b, t, c * f))
        if x_mask is None:
            return x, None
        return x, x_mask[:, :, :-2:2][:, :, :-2:2]


class _NotBuilt(nn.Module):
    """Other subsampling variants: only their names are needed for ``isinstance``."""


Conv1dSubsampling2 = type("Conv1dSubsampling2", (_NotBuilt,), {})
Conv1dSubsampling3 = type("Conv1dSubsampling3", (_NotBuilt,), {})
Conv2dSubsampling1 = type("Conv2dSubsampling1", (_NotBuilt,), {})
Conv2dSubsampling2 = type("Conv2dSubsampling2", (_NotBuilt,), {})
Conv2dSubsampling6 = type("Conv2dSubsampling6", (_NotBuilt,), {})
Conv2dSubsampling8 = type("Conv2dSubsampling8", (_NotBuilt,), {})


def check_short_utt(ins, size):
    if isinstance(ins, Conv2dSubsampling) and size < 7:
        return True, 7
    return False, -1


class Conv2dSubsamplingWOPosEnc(nn.Module):
    def __init__(self, idim, odim, dropout_rate, kernels, strides):
        super().__init__()
        assert len(kernels) == len(strides)
        conv = []
        olen = idim
        for i, (k, s) in enumerate(zip(kernels, strides)):
            conv += [nn.Conv2d(1 if i == 0 else odim, odim, k, s), nn.ReLU()]
            olen = math.floor((olen - k) / s + 1)
        self.conv = nn.Sequential(*conv)
        self.out = nn.Linear(odim * olen, odim)
        self.strides = strides
        self.kernels = kernels

    def forward(self, x, x_mask):
        x = x.unsqueeze(1)
        x = self.conv(x)
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).contiguous().view(b, t, c * f))
        if x_mask is None:
            return x, None
        for k, s in zip(self.kernels, self.strides):
            x_mask = x_mask[:, :, : -k + 1 : s]
        return x, x_mask


# --------------------------------------------------------------------------------------
# A.8  espnet2 DefaultFrontend (log-mel) / UtteranceMVN
#      configured at configs/ASR/branchformer_transformer+ctc_english.yaml:9-20
# --------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = torch.as_tensor(f, dtype=torch.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return torch.where(f >= min_log_hz, min_log_mel + torch.log(f.clamp(min=1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = torch.as_tensor(m, dtype=torch.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(fs=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=None):
    """librosa.filters.mel(htk=False, norm='slaney') restated. Returns (n_mels, n_fft//2+1) fp32."""
    fmax = fs / 2 if fmax is None else fmax
    n_freq = n_fft // 2 + 1
    fftfreqs = torch.linspace(0, fs / 2, n_freq, dtype=torch.float64)
    mel_pts = torch.linspace(float(_hz_to_mel_slaney(fmin)), float(_hz_to_mel_slaney(fmax)), n_mels + 2, dtype=torch.float64)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f.unsqueeze(1) - fftfreqs.unsqueeze(0)
    lower = -ramps[:-2] / fdiff[:-1].unsqueeze(1)
    upper = ramps[2:] / fdiff[1:].unsqueeze(1)
    weights = torch.clamp(torch.minimum(lower, upper), min=0)
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights = weights * enorm.unsqueeze(1)
    return weights.to(torch.float32)


class DefaultFrontend(nn.Module):
    """STFT(n_fft, win_length, hop, hann, center, reflect) -> power -> Slaney mel -> log(clamp 1e-10)."""

    def __init__(self, fs=16000, n_fft=512, win_length=None, hop_length=128, window="hann",
                 center=True, normalized=False, onesided=True, n_mels=80, fmin=None, fmax=None,
                 htk=False, frontend_conf=None, apply_stft=True):
        super().__init__()
        assert window == "hann" and not htk
        self.n_fft, self.hop_length = n_fft, hop_length
        self.win_length = n_fft if win_length is None else win_length
        self.center, self.n_mels = center, n_mels
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        self.register_buffer("melmat", slaney_mel_filterbank(fs, n_fft, n_mels, fmin, fmax).T.contiguous(), persistent=False)

    def output_size(self):
        return self.n_mels

    def forward(self, input, input_lengths):
        window = torch.hann_window(self.win_length, dtype=input.dtype, device=input.device)
        spec = torch.stft(input, self.n_fft, self.hop_length, self.win_length, window,
                          center=self.center, pad_mode="reflect", normalized=False,
                          onesided=True, return_complex=True)
        spec = spec.transpose(1, 2)  # (B, frames, freq)
        if self.center:
            pad = self.n_fft // 2
            input_lengths = input_lengths + 2 * pad
        olens = torch.div(input_lengths - self.n_fft, self.hop_length, rounding_mode="trunc") + 1
        power = spec.real ** 2 + spec.imag ** 2
        pad_mask = make_pad_mask(olens, power, 1)
        power = power.masked_fill(pad_mask, 0.0)
        mel = torch.matmul(power, self.melmat)
        mel = torch.clamp(mel, min=1e-10).log()
        mel = mel.masked_fill(make_pad_mask(olens, mel, 1), 0.0)
        return mel, olens


# --------------------------------------------------------------------------------------
# A.8b espnet2 SpecAug (espnet2/asr/specaug/specaug.py, layers/time_warp.py, layers/mask_along_axis.py), configured at
#      configs/ASR/branchformer_transformer+ctc_english.yaml:21-37, applied train-only at espnet_model.py:383-385.
#      espnet2 is absent here: restated from its published source (parity unpinned for this leaf).  Draws use the
#      torch HOST generator (the reference draws on the feature tensor's device).
# --------------------------------------------------------------------------------------
def time_warp(x, window=80, mode="bicubic"):
    org_size = x.size()
    if x.dim() == 3:
        x = x[:, None]
    t = x.shape[2]
    if t - window <= window:
        return x.view(*org_size)
    center = torch.randint(window, t - window, (1,))[0]
    warped = torch.randint(center - window, center + window, (1,))[0] + 1
    left = F.interpolate(x[:, :, :center], (warped, x.shape[3]), mode=mode, align_corners=False)
    right = F.interpolate(x[:, :, center:], (t - warped, x.shape[3]), mode=mode, align_corners=False)
    x = torch.cat([left, right], dim=-2)
    return x.view(*org_size)


def mask_along_axis(spec, spec_lengths, mask_width_range=(0, 30), dim=1, num_mask=2, replace_with_zero=True):
    org_size = spec.size()
    if spec.dim() == 4:
        spec = spec.view(-1, spec.size(2), spec.size(3))
    B, D = spec.shape[0], spec.shape[dim]
    mask_length = torch.randint(mask_width_range[0], mask_width_range[1], (B, num_mask)).unsqueeze(2)
    mask_pos = torch.randint(0, max(1, D - int(mask_length.max())), (B, num_mask)).unsqueeze(2)
    aran = torch.arange(D)[None, None, :]
    mask = ((mask_pos <= aran) * (aran < (mask_pos + mask_length))).any(dim=1)
    mask = mask.unsqueeze(2) if dim == 1 else mask.unsqueeze(1)
    value = 0.0 if replace_with_zero else spec.mean()
    spec = spec.masked_fill(mask, value)
    return spec.view(*org_size), spec_lengths


class SpecAug(nn.Module):
    def __init__(self, apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True,
                 freq_mask_width_range=(0, 20), num_freq_mask=2, apply_time_mask=True, time_mask_width_range=None,
                 time_mask_width_ratio_range=None, num_time_mask=2):
        super().__init__()
        assert apply_time_warp or apply_freq_mask or apply_time_mask
        assert not (apply_time_mask and time_mask_width_range is not None and time_mask_width_ratio_range is not None)
        self.apply_time_warp, self.window, self.mode = apply_time_warp, time_warp_window, time_warp_mode
        self.apply_freq_mask, self.num_freq_mask = apply_freq_mask, num_freq_mask
        self.freq_range = (0, freq_mask_width_range) if isinstance(freq_mask_width_range, int) else tuple(freq_mask_width_range)
        self.apply_time_mask, self.num_time_mask = apply_time_mask, num_time_mask
        self.time_range = time_mask_width_range
        if isinstance(self.time_range, int):
            self.time_range = (0, self.time_range)
        self.time_ratio = time_mask_width_ratio_range
        if isinstance(self.time_ratio, float):
            self.time_ratio = (0.0, self.time_ratio)

    def forward(self, x, x_lengths=None):
        if self.apply_time_warp:                                      # TimeWarp.forward
            if x_lengths is None or all(le == x_lengths[0] for le in x_lengths):
                x = time_warp(x, window=self.window, mode=self.mode)
            else:
                ys = [time_warp(x[i][None, : x_lengths[i]], window=self.window, mode=self.mode)[0] for i in range(x.size(0))]
                x = torch.nn.utils.rnn.pad_sequence(ys, batch_first=True, padding_value=0.0)
        if self.apply_freq_mask:                                      # MaskAlongAxis(dim="freq")
            x, x_lengths = mask_along_axis(x, x_lengths, self.freq_range, dim=2, num_mask=self.num_freq_mask)
        if self.apply_time_mask:
            if self.time_range is not None:                           # MaskAlongAxis(dim="time")
                x, x_lengths = mask_along_axis(x, x_lengths, tuple(self.time_range), dim=1, num_mask=self.num_time_mask)
            else:                                                     # MaskAlongAxisVariableMaxWidth
                max_width = x.shape[1]
                lo = max(0, math.floor(max_width * self.time_ratio[0]))
                hi = min(max_width, math.floor(max_width * self.time_ratio[1]))
                if hi > lo:
                    x, x_lengths = mask_along_axis(x, x_lengths, (lo, hi), dim=1, num_mask=self.num_time_mask)
        return x, x_lengths


class UtteranceMVN(nn.Module):
    def __init__(self, norm_means=True, norm_vars=False, eps=1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps

    def forward(self, x, ilens=None):
        if ilens is None:
            ilens = x.new_full([x.size(0)], x.size(1))
        ilens_ = ilens.to(x.device, x.dtype).view(-1, *[1 for _ in range(x.dim() - 1)])
        pad = make_pad_mask(ilens, x, 1)
        x = x.masked_fill(pad, 0.0)
        mean = x.sum(dim=1, keepdim=True) / ilens_
        if self.norm_means:
            x = x - mean
            x = x.masked_fill(pad, 0.0)
            if self.norm_vars:
                var = x.pow(2).sum(dim=1, keepdim=True) / ilens_
                std = torch.clamp(var.sqrt(), min=self.eps)
                x = x / std
        elif self.norm_vars:
            y = (x - mean).masked_fill(pad, 0.0)
            var = y.pow(2).sum(dim=1, keepdim=True) / ilens_
            std = torch.clamp(var.sqrt(), min=self.eps)
            x = x / std
        return x, ilens


# --------------------------------------------------------------------------------------
# A.9  espnet2/asr/decoder/transformer_decoder.py + transformer/decoder_layer.py
#      call sites: src/tasks/asr.py:176-194, espnet_model.py:557-560
# --------------------------------------------------------------------------------------
def subsequent_mask(size, device="cpu", dtype=torch.bool):
    ret = torch.ones(size, size, device=device, dtype=dtype)
    return torch.tril(ret, out=ret)


class DecoderLayer(nn.Module):
    def __init__(self, size, self_attn, src_attn, feed_forward, dropout_rate,
                 normalize_before=True, concat_after=False):
        super().__init__()
        self.size = size
        self.self_attn, self.src_attn, self.feed_forward = self_attn, src_attn, feed_forward
        self.norm1, self.norm2, self.norm3 = LayerNorm(size), LayerNorm(size), LayerNorm(size)
        self.dropout = nn.Dropout(dropout_rate)
        self.normalize_before = normalize_before
        self.concat_after = concat_after
        assert not concat_after

    def forward(self, tgt, tgt_mask, memory, memory_mask, cache=None):
        residual = tgt
        if self.normalize_before:
            tgt = self.norm1(tgt)
        if cache is None:
            tgt_q, tgt_q_mask = tgt, tgt_mask
        else:
            tgt_q = tgt[:, -1:, :]
            residual = residual[:, -1:, :]
            tgt_q_mask = None if tgt_mask is None else tgt_mask[:, -1:, :]
        x = residual + self.dropout(self.self_attn(tgt_q, tgt, tgt, tgt_q_mask))
        if not self.normalize_before:
            x = self.norm1(x)
        residual = x
        if self.normalize_before:
            x = self.norm2(x)
        x = residual + self.dropout(self.src_attn(x, memory, memory, memory_mask))
        if not self.normalize_before:
            x = self.norm2(x)
        residual = x
        if self.normalize_before:
            x = self.norm3(x)
        x = residual + self.dropout(self.feed_forward(x))
        if not self.normalize_before:
            x = self.norm3(x)
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x, tgt_mask, memory, memory_mask


class TransformerDecoder(nn.Module):
    def __init__(self, vocab_size, encoder_output_size, attention_heads=4, linear_units=2048,
                 num_blocks=6, dropout_rate=0.1, positional_dropout_rate=0.1,
                 self_attention_dropout_rate=0.0, src_attention_dropout_rate=0.0,
                 input_layer="embed", use_output_layer=True, pos_enc_class=PositionalEncoding,
                 normalize_before=True, concat_after=False, layer_drop_rate=0.0):
        super().__init__()
        attention_dim = encoder_output_size
        assert input_layer == "embed"
        self.embed = nn.Sequential(
            nn.Embedding(vocab_size, attention_dim),
            pos_enc_class(attention_dim, positional_dropout_rate),
        )
        self.normalize_before = normalize_before
        if normalize_before:
            self.after_norm = LayerNorm(attention_dim)
        self.output_layer = nn.Linear(attention_dim, vocab_size) if use_output_layer else None
        self.decoders = repeat(
            num_blocks,
            lambda lnum: DecoderLayer(
                attention_dim,
                MultiHeadedAttention(attention_heads, attention_dim, self_attention_dropout_rate),
                MultiHeadedAttention(attention_heads, attention_dim, src_attention_dropout_rate),
                PositionwiseFeedForward(attention_dim, linear_units, dropout_rate),
                dropout_rate,
                normalize_before,
                concat_after,
            ),
            layer_drop_rate,
        )

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens):
        tgt = ys_in_pad
        tgt_mask = (~make_pad_mask(ys_in_lens)[:, None, :]).to(tgt.device)
        m = subsequent_mask(tgt_mask.size(-1), device=tgt_mask.device).unsqueeze(0)
        tgt_mask = tgt_mask & m
        memory = hs_pad
        memory_mask = (~make_pad_mask(hlens, maxlen=memory.size(1)))[:, None, :].to(memory.device)
        x = self.embed(tgt)
        x, tgt_mask, memory, memory_mask = self.decoders(x, tgt_mask, memory, memory_mask)
        if self.normalize_before:
            x = self.after_norm(x)
        if self.output_layer is not None:
            x = self.output_layer(x)
        olens = tgt_mask.sum(1)
        return x, olens


# --------------------------------------------------------------------------------------
# A.10  losses / utilities
#       call sites: espnet_model.py:175-186,553-577
# --------------------------------------------------------------------------------------
class LabelSmoothingLoss(nn.Module):
    def __init__(self, size, padding_idx, smoothing, normalize_length=False, criterion=None):
        super().__init__()
        self.criterion = nn.KLDivLoss(reduction="none") if criterion is None else criterion
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.normalize_length = normalize_length

    def forward(self, x, target):
        assert x.size(2) == self.size
        batch_size = x.size(0)
        x = x.view(-1, self.size)
        target = target.view(-1)
        with torch.no_grad():
            true_dist = x.clone()
            true_dist.fill_(self.smoothing / (self.size - 1))
            ignore = target == self.padding_idx
            total = len(target) - ignore.sum().item()
            target = target.masked_fill(ignore, 0)
            true_dist.scatter_(1, target.unsqueeze(1), self.confidence)
        kl = self.criterion(torch.log_softmax(x, dim=1), true_dist)
        denom = total if self.normalize_length else batch_size
        return kl.masked_fill(ignore.unsqueeze(1), 0).sum() / denom


def pad_list(xs, pad_value):
    n_batch = len(xs)
    max_len = max(x.size(0) for x in xs)
    pad = xs[0].new(n_batch, max_len, *xs[0].size()[1:]).fill_(pad_value)
    for i in range(n_batch):
        pad[i, : xs[i].size(0)] = xs[i]
    return pad


def add_sos_eos(ys_pad, sos, eos, ignore_id):
    _sos = ys_pad.new([sos])
    _eos = ys_pad.new([eos])
    ys = [y[y != ignore_id] for y in ys_pad]
    ys_in = [torch.cat([_sos, y], dim=0) for y in ys]
    ys_out = [torch.cat([y, _eos], dim=0) for y in ys]
    return pad_list(ys_in, eos), pad_list(ys_out, ignore_id)


def th_accuracy(pad_outputs, pad_targets, ignore_label):
    pad_pred = pad_outputs.view(pad_targets.size(0), pad_targets.size(1), pad_outputs.size(1)).argmax(2)
    mask = pad_targets != ignore_label
    numerator = torch.sum(pad_pred.masked_select(mask) == pad_targets.masked_select(mask))
    denominator = torch.sum(mask)
    return float(numerator) / float(denominator)


def force_gatherable(data, device):
    if isinstance(data, dict):
        return {k: force_gatherable(v, device) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return type(data)(force_gatherable(v, device) for v in data)
    if isinstance(data, float):
        return torch.tensor([data], dtype=torch.float, device=device)
    if isinstance(data, int):
        return torch.tensor([data], dtype=torch.long, device=device)
    if isinstance(data, torch.Tensor):
        if data.dim() == 0:
            data = data[None]
        return data.to(device)
    return data


def _levenshtein(a: List, b: List) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


class ErrorCalculator:
    """espnet.nets.e2e_asr_common.ErrorCalculator (CER/WER on id sequences)."""

    def __init__(self, char_list, sym_space, sym_blank, report_cer=False, report_wer=False):
        self.report_cer, self.report_wer = report_cer, report_wer
        self.char_list = char_list
        self.space, self.blank = sym_space, sym_blank
        self.idx_blank = self.char_list.index(self.blank)
        self.idx_space = self.char_list.index(self.space) if self.space in self.char_list else None

    def __call__(self, ys_hat, ys_pad, is_ctc=False):
        if is_ctc:
            return self.calculate_cer_ctc(ys_hat, ys_pad)
        if not self.report_cer and not self.report_wer:
            return None, None
        seqs_hat, seqs_true = self.convert_to_char(ys_hat, ys_pad)
        cer = self.calculate_cer(seqs_hat, seqs_true) if self.report_cer else None
        wer = self.calculate_wer(seqs_hat, seqs_true) if self.report_wer else None
        return cer, wer

    def calculate_cer_ctc(self, ys_hat, ys_pad):
        cers, char_ref_lens = [], []
        for i, y in enumerate(ys_hat):
            y_hat = [x[0] for x in groupby(y)]
            y_true = ys_pad[i]
            seq_hat, seq_true = [], []
            for idx in y_hat:
                idx = int(idx)
                if idx != -1 and idx != self.idx_blank and idx != self.idx_space:
                    seq_hat.append(self.char_list[int(idx)])
            for idx in y_true:
                idx = int(idx)
                if idx != -1 and idx != self.idx_blank and idx != self.idx_space:
                    seq_true.append(self.char_list[int(idx)])
            hyp_chars = "".join(seq_hat)
            ref_chars = "".join(seq_true)
            if len(ref_chars) > 0:
                cers.append(_levenshtein(list(hyp_chars), list(ref_chars)))
                char_ref_lens.append(len(ref_chars))
        return float(sum(cers)) / sum(char_ref_lens) if cers else None

    def convert_to_char(self, ys_hat, ys_pad):
        seqs_hat, seqs_true = [], []
        for i, y_hat in enumerate(ys_hat):
            y_true = ys_pad[i]
            eos_true = (y_true == -1).nonzero()
            ymax = int(eos_true[0]) if len(eos_true) > 0 else len(y_true)
            seq_hat = [self.char_list[int(idx)] for idx in y_hat[:ymax]]
            seq_true = [self.char_list[int(idx)] for idx in y_true if int(idx) != -1]
            seq_hat_text = "".join(seq_hat).replace(self.space, " ").replace(self.blank, "")
            seq_true_text = "".join(seq_true).replace(self.space, " ")
            seqs_hat.append(seq_hat_text)
            seqs_true.append(seq_true_text)
        return seqs_hat, seqs_true

    def calculate_cer(self, seqs_hat, seqs_true):
        dists, lens = [], []
        for h, r in zip(seqs_hat, seqs_true):
            h, r = h.replace(" ", ""), r.replace(" ", "")
            dists.append(_levenshtein(list(h), list(r)))
            lens.append(len(r))
        return float(sum(dists)) / sum(lens)

    def calculate_wer(self, seqs_hat, seqs_true):
        dists, lens = [], []
        for h, r in zip(seqs_hat, seqs_true):
            dists.append(_levenshtein(h.split(), r.split()))
            lens.append(len(r.split()))
        return float(sum(dists)) / sum(lens)


# --------------------------------------------------------------------------------------
# CTC greedy collapse: reference semantics at src/models/maskctc_model.py:289-291
# --------------------------------------------------------------------------------------
def ctc_greedy_collapse(ids: torch.Tensor, length: Optional[int] = None, blank: int = 0) -> List[int]:
    seq = ids.tolist() if length is None else ids[:length].tolist()
    return [k for k, _ in groupby(seq) if k != blank]
