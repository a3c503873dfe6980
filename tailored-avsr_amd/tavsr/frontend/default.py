"""``DefaultFrontend`` - drop-in for espnet2.asr.frontend.default.DefaultFrontend as the reference configures it
(configs/ASR/branchformer_transformer+ctc_english.yaml:9-20, built at src/tasks/asr.py, called at
src/models/espnet_model.py:378-388): waveform -> STFT (n_fft, win_length, hop, hann, centre + reflect) -> power ->
Slaney mel filterbank -> log(clamp 1e-10), padded frames zeroed, ``olens = 1 + len // hop``.

The windowed DFT and the mel projection are two fp32 MFMA GEMMs (``tavsr_gemm``) against constant matrices built
once on the host in float64; framing, power and log+mask are ``csrc/frontend.hip``.  No parameters, no backward (the
waveform never requires a gradient on this path)."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from .. import ops


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm="slaney") -> [n_mels, n_fft//2 + 1] float64."""
    n_freq = n_fft // 2 + 1
    fftfreqs = np.linspace(0, fs / 2, n_freq)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower, upper = -ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


class DefaultFrontend(torch.nn.Module):
    def __init__(self, fs: int = 16000, n_fft: int = 512, win_length: Optional[int] = None, hop_length: int = 128,
                 window: Optional[str] = "hann", center: bool = True, normalized: bool = False, onesided: bool = True,
                 n_mels: int = 80, fmin: Optional[int] = None, fmax: Optional[int] = None, htk: bool = False,
                 frontend_conf: Optional[dict] = None, apply_stft: bool = True):
        super().__init__()
        if isinstance(fs, str):
            raise ValueError("fs must be an integer sample rate")
        if window != "hann" or htk or normalized or not onesided or not apply_stft or frontend_conf:
            raise ValueError("only the shipped recipe's frontend is implemented: hann window, Slaney mel, "
                             "normalized=False, onesided=True, no WPE/beamformer frontend_conf")
        self.fs, self.n_fft, self.hop_length = fs, n_fft, hop_length
        self.win_length = n_fft if win_length is None else win_length
        self.center, self.n_mels = center, n_mels
        self.fmin = 0 if fmin is None else fmin
        self.fmax = fs / 2 if fmax is None else fmax
        self.nfreq = n_fft // 2 + 1
        self.kpad = (self.nfreq + 31) // 32 * 32           # mel GEMM K, whole 32-wide K-steps
        self._const = {}

    def output_size(self) -> int:
        return self.n_mels

    def _constants(self, device):
        c = self._const.get(device)
        if c is None:
            n = np.arange(self.n_fft, dtype=np.float64)
            k = np.arange(self.nfreq, dtype=np.float64)
            ang = 2.0 * np.pi * np.outer(k, n) / self.n_fft
            dft = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0)              # [2*nfreq, n_fft]: re | im rows
            win = np.zeros(self.n_fft)
            left = (self.n_fft - self.win_length) // 2
            m = np.arange(self.win_length, dtype=np.float64)
            win[left:left + self.win_length] = 0.5 - 0.5 * np.cos(2.0 * np.pi * m / self.win_length)   # periodic hann
            mel = np.zeros((self.n_mels, self.kpad))
            mel[:, :self.nfreq] = slaney_mel_filterbank(self.fs, self.n_fft, self.n_mels, self.fmin, self.fmax)
            c = tuple(torch.from_numpy(a.astype(np.float32)).to(device) for a in (dft, win, mel))
            self._const[device] = c
        return c

    def forward(self, input: torch.Tensor, input_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if input.dim() != 2:
            raise ValueError("the HIP frontend takes single-channel waveforms (B, Nsamples)")
        B, N = input.shape
        dft, win, mel = self._constants(input.device)
        pad = self.n_fft // 2 if self.center else 0
        T = (N + 2 * pad - self.n_fft) // self.hop_length + 1
        olens = torch.div(input_lengths.to(torch.int64) + 2 * pad - self.n_fft, self.hop_length, rounding_mode="trunc") + 1
        olens = olens.to(input.device)
        frames = ops.stft_frames(input.contiguous().float(), win, T, self.n_fft, self.hop_length, self.center)
        spec = ops.linear(frames, dft)                                             # [B*T, 2*nfreq]
        power = ops.power_spec(spec, self.nfreq, self.kpad, B, T, olens)
        feats = ops.log_mask(ops.linear(power, mel).view(B, T, self.n_mels), B, T, olens)
        return feats, olens
