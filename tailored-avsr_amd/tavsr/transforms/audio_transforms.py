"""Audio augmentation on the MI355X: AddNoise of src/transforms/audio_transforms.py:74-139 (the evaluation
pipeline of avsr_main.py:160-162), same constructor arguments and python ``random`` draws (offset, then SNR level).
The noise recording is given as a tensor or as a PCM WAV file read with the standard library (the reference goes through
sox; a file at another sample rate needs resampling first and is refused).  SpeedRate (src/transforms/audio_transforms.py:141-178:
sox "speed" + "rate") is a windowed-sinc resampling launch - sox itself is absent here, so this one is a parity-unpinned restatement
of its documented effect (band-limited rate conversion), with the reference's random draw."""
from __future__ import annotations

import os
import random
import wave

import numpy as np
import torch

from .. import ops

SNR_LEVELS = [-5, 0, 5, 10, 15, 20, 9999]


def _read_wav_mono(path, sample_rate):
    with wave.open(path, "rb") as w:
        if w.getframerate() != sample_rate:
            raise NotImplementedError(f"{path}: {w.getframerate()} Hz noise needs resampling to {sample_rate} Hz first")
        if w.getsampwidth() != 2:
            raise NotImplementedError(f"{path}: only 16-bit PCM noise files are read here")
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
    return torch.from_numpy(data[:, 0].astype(np.float32)).unsqueeze(0)        # "remix 1": first channel, unnormalised


class AddNoise:
    def __init__(self, noise_path=None, sample_rate: float = 16000, snr_target: int = None, noise: torch.Tensor = None,
                 device="cuda"):
        if noise is None:
            if noise_path is None or not os.path.exists(noise_path):
                raise IOError(f"Noise path `{noise_path}` does not exist")
            noise = _read_wav_mono(noise_path, int(sample_rate))
        self.entire_noise = noise.float().reshape(1, -1).to(device).contiguous()
        self.entire_noise_length = self.entire_noise.shape[-1]
        self.sample_rate = sample_rate
        self.snr_target = snr_target

    def __call__(self, audio_data: torch.Tensor) -> torch.Tensor:
        """audio_data [1, T] on the device -> noisy waveform [1, T]."""
        ops.require_cuda(audio_data)
        audio_length = audio_data.shape[-1]
        if self.entire_noise_length > audio_length:
            offset = random.randint(0, self.entire_noise_length - audio_length)
            noise = self.entire_noise[..., offset:offset + audio_length]
        elif self.entire_noise_length < audio_length:
            noise = torch.cat([self.entire_noise, torch.zeros((1, audio_length - self.entire_noise_length),
                                                              device=self.entire_noise.device)], dim=-1)
        else:
            # the reference leaves `noise` unbound when the two lengths are equal (UnboundLocalError); use the recording
            noise = self.entire_noise
        snr_db = random.choice(SNR_LEVELS) if not self.snr_target else self.snr_target
        if snr_db == 9999:
            return audio_data
        snr = (10 ** (snr_db / 10.0)) ** 0.5
        return ops.add_noise(audio_data.float().contiguous(), noise.contiguous(), 1 / snr)


class SpeedRate:
    """Subsample / upsample the waveform: speed factor drawn from (0.9, 1.0, 1.1) as the reference does, then the clip is played
    that many times faster and resampled back to ``sample_rate`` (sox effects "speed f", "rate sample_rate").  Same constructor,
    same single ``random.choice`` draw, [1, T] in -> [1, round(T / f)] out on the device."""

    FACTORS = [0.9, 1.0, 1.1]

    def __init__(self, sample_rate: float = 16000):
        self.sample_rate = sample_rate

    def __call__(self, audio_data: torch.Tensor) -> torch.Tensor:
        speed_factor = random.choice(self.FACTORS)
        if speed_factor == 1.0:
            return audio_data
        ops.require_cuda(audio_data)
        return ops.resample(audio_data, speed_factor)
