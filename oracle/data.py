"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's batch assembly.

* video transforms: src/transforms/video_transforms.py (Compose :11-35, Normalise :59-80, TimeMasking :82-121 - in place,
  mean frame taken once before the masks -, CenterCrop :123-145, VideoSpeedRate :147-185) and the two torchvision
  transforms avsr_main.py:172-173 adds: RandomCrop (get_params: ``torch.randint`` for the top row, then for the left
  column; identity when the sizes match) and RandomHorizontalFlip (``torch.rand(1) < p`` -> flip of the last axis).
  torchvision is absent from this image: **those two are parity-unpinned** restatements of its documented behaviour.
* AddNoise.__call__: src/transforms/audio_transforms.py:102-139.
* collate: src/utils/avsr_dataloader.py:40-142 (pad_sequence with ignore_id, audio cropped to a multiple of 640).
Pinned by tests/golden/data_pipeline.npz, which oracle/gen_golden_data.py writes by running the reference's own classes
and collate function (imported from /root/reference) on seeded inputs."""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn


class Compose:
    def __init__(self, preprocess):
        self.preprocess = preprocess

    def __call__(self, x):
        for p in self.preprocess:
            if p is not None:
                x = p(x)
        return x


class Normalise:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, v):
        return (v - self.mean) / self.std


class TimeMasking:
    def __init__(self, fps=25.0, max_frames=None, max_seconds=None):
        assert max_frames or max_seconds
        self.fps, self.max_frames, self.max_seconds = fps, max_frames, max_seconds

    def __call__(self, v):
        max_frames = int(self.fps * self.max_seconds) if self.max_frames is None else self.max_frames
        n = v.shape[0]
        mean_frame = v.mean(axis=0)
        for _ in range(int(n / self.fps)):
            ln = random.randint(0, max_frames)
            if ln > 0:
                off = random.randint(0, n - ln)
                v[off:off + ln] = mean_frame
        return v


class CenterCrop:
    def __init__(self, crop_size):
        self.crop_size = crop_size

    def __call__(self, v):
        _, h, w = v.shape
        th, tw = self.crop_size
        dw = int(round((w - tw)) / 2.)
        dh = int(round((h - th)) / 2.)
        return v[:, dh:dh + th, dw:dw + tw]


class RandomCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, v):
        h, w = v.shape[-2:]
        th, tw = self.size
        if w == tw and h == th:
            return v
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return v[..., i:i + th, j:j + tw]


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, v):
        return v.flip(-1) if torch.rand(1) < self.p else v


class VideoSpeedRate:
    def __init__(self, speed_rate=1.0):
        self._speed_rate = speed_rate

    def __call__(self, x):
        if self._speed_rate == 1.:
            return x
        n = x.shape[0]
        idx = list(map(int, np.linspace(start=0, stop=n, num=int(n / self._speed_rate), endpoint=False)))
        return x[idx]


class AddNoise:
    def __init__(self, noise, snr_target=None):
        self.entire_noise = noise
        self.entire_noise_length = noise.shape[-1]
        self.snr_target = snr_target

    def __call__(self, audio):
        n = audio.shape[-1]
        if self.entire_noise_length > n:
            off = random.randint(0, self.entire_noise_length - n)
            noise = self.entire_noise[..., off:off + n]
        elif self.entire_noise_length < n:
            noise = torch.cat([self.entire_noise, torch.zeros((self.entire_noise.shape[0], n - self.entire_noise_length))], dim=-1)
        else:
            noise = self.entire_noise
        snr_db = random.choice([-5, 0, 5, 10, 15, 20, 9999]) if not self.snr_target else self.snr_target
        if snr_db == 9999:
            return audio
        snr = (10 ** (snr_db / 10.0)) ** 0.5
        pa = (audio ** 2).sum() / (n * 1.0)
        pn = (noise ** 2).sum() / (noise.shape[-1] * 1.0)
        return audio + 1 / snr * noise * torch.sqrt(pa / pn)


def avsr_data_processing(data, audio_transforms, video_transforms, tokenizer, converter, ignore_id=-1):
    b = {"audio": [], "audio_lengths": [], "video": [], "video_lengths": [], "text": [], "text_lengths": []}
    for s in data:
        a = audio_transforms(s["audio"]) if audio_transforms else s["audio"]
        a = a.transpose(1, 0)
        a = a[: a.shape[0] // 640 * 640, :]
        b["audio"].append(a)
        b["audio_lengths"].append(a.shape[0])
        v = video_transforms(s["video"]) if video_transforms else s["video"]
        b["video"].append(v)
        b["video_lengths"].append(v.shape[0])
        t = torch.Tensor(converter.tokens2ids(tokenizer.text2tokens(s["transcription"])))
        b["text"].append(t)
        b["text_lengths"].append(t.shape[0])
    for k, dt in (("audio", torch.float32), ("video", torch.float32), ("text", torch.int64)):
        b[k] = nn.utils.rnn.pad_sequence(b[k], padding_value=ignore_id, batch_first=True).type(dt)
        b[k + "_lengths"] = torch.Tensor(b[k + "_lengths"]).type(torch.int64)
    return b


# ---- seeded inputs shared by the generator and the tests --------------------------------------------------------
class CharTokenizer:
    def text2tokens(self, s):
        return list(s)


class CharConverter:
    def tokens2ids(self, toks):
        return [1 + (ord(c) % 37) for c in toks]


def make_samples(seed, n=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        T = int(torch.randint(30, 80, (1,), generator=g))
        L = 640 * int(torch.randint(20, 60, (1,), generator=g)) + int(torch.randint(0, 639, (1,), generator=g))
        out.append({"sample_id": f"s{seed}_{i}", "audio": 0.1 * torch.randn(1, L, generator=g),
                    "video": torch.randint(0, 256, (T, 96, 96), generator=g, dtype=torch.uint8),
                    "transcription": "HOLA QUE TAL"[: 4 + 3 * i]})
    return out


def make_noise(seed, n=90000):
    return 300.0 * torch.randn(1, n, generator=torch.Generator().manual_seed(seed))


MEAN, STD, FPS = 0.421, 0.165, 25.0          # avsr_main.py:164-167 (lip-ROI statistics, frame rate)


def seed_all(seed):
    random.seed(seed)
    torch.manual_seed(seed)


def resample_sinc(x, factor, rolloff=0.9475937167399596, zeros=64, beta=14.769656459379492):
    """CPU restatement (numpy, fp64) of the band-limited resampling the SpeedRate augmentation stands for
    (src/transforms/audio_transforms.py:141-178 calls sox "speed f" + "rate"; sox is absent from this image, so this follows the
    textbook windowed-sinc interpolation with the Kaiser "best" constants - parity UNPINNED against sox itself):
    y[n] = sum_k x[k] c sinc(c (n f - k)) w((n f - k) / W), c = rolloff min(1, 1 / f), W = zeros / c, n < round(len / f)."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    n_out = int(round(len(x) / factor))
    c = rolloff * min(1.0, 1.0 / factor)
    W = zeros / c
    y = np.zeros(n_out)
    i0b = np.i0(beta)
    for n in range(n_out):
        t = n * factor
        k0, k1 = max(0, int(np.ceil(t - W))), min(len(x) - 1, int(np.floor(t + W)))
        k = np.arange(k0, k1 + 1)
        u = t - k
        w = np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - (u / W) ** 2))) / i0b
        y[n] = np.sum(x[k] * c * np.sinc(c * u) * w)
    return y
