"""Video augmentation on the MI355X: the transforms of src/transforms/video_transforms.py and the two
torchvision ones avsr_main.py:168-179 composes with them, with the same class names, constructor arguments and RANDOM
DRAWS (python ``random`` for TimeMasking, torch's host generator for RandomCrop / RandomHorizontalFlip, in the
reference's order - so a seeded run picks the same masks, windows and flips).

A transform does not touch pixels: it records what it does on a ``VideoClip`` (source frames on the device + frame index
list + the chain of Normalise steps + crop window / mirror + masked frames).  ``VideoClip.render`` - called by the
collate function for every sample, straight into that sample's row of the padded batch - produces the result with ONE
pass over the pixels (``tavsr_video_prep``).  Order rules: Normalise steps and the crop / mirror commute with everything
(the mean frame of TimeMasking is per pixel); a Normalise placed after TimeMasking is applied before it (affine maps
commute with the mean up to fp32 rounding)."""
from __future__ import annotations

import random
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import ops


class VideoClip:
    def __init__(self, frames: torch.Tensor):
        if frames.dim() != 3:
            raise ValueError(f"video clip must be [T, H, W], got {tuple(frames.shape)}")
        if frames.dtype not in (torch.uint8, torch.float32):
            frames = frames.float()
        ops.require_cuda(frames)
        self.src = frames.contiguous()
        self.index: Optional[List[int]] = None                 # source frame of each output frame (VideoSpeedRate)
        self.affine: List[Tuple[float, float]] = []
        self.h, self.w = int(frames.shape[1]), int(frames.shape[2])
        self.y0 = self.x0 = 0
        self.flipped = False
        self.masked: Optional[np.ndarray] = None               # uint8 [T]

    # the reference's transforms read video_data.shape: (frames, height, width) of the CURRENT view
    @property
    def shape(self):
        return (len(self.index) if self.index is not None else int(self.src.shape[0]), self.h, self.w)

    def crop(self, dy, dx, th, tw):
        if dy < 0 or dx < 0 or dy + th > self.h or dx + tw > self.w:
            raise ValueError(f"crop {th}x{tw} at ({dy},{dx}) outside the {self.h}x{self.w} view")
        self.y0 += dy
        self.x0 += (self.w - tw - dx) if self.flipped else dx    # columns of a mirrored view run right to left
        self.h, self.w = th, tw

    def render(self, out: Optional[torch.Tensor] = None, pad_value: float = 0.0) -> torch.Tensor:
        """-> float32 [T, h, w]; with ``out`` [Tpad, h, w] (a row of the padded batch) frames T.. are set to pad_value."""
        T = self.shape[0]
        if out is None:
            out = torch.empty((T, self.h, self.w), dtype=torch.float32, device=self.src.device)
        assert out.is_contiguous() and out.shape[1:] == (self.h, self.w) and out.shape[0] >= T
        ops.video_prep(self.src, self.index, T, self.y0, self.x0, self.h, self.w, self.flipped, self.affine, self.masked, out,
                       pad_value)
        return out


def _clip(v) -> VideoClip:
    return v if isinstance(v, VideoClip) else VideoClip(v)


class Compose:
    def __init__(self, preprocess):
        self.preprocess = preprocess

    def __call__(self, video_data):
        video_data = _clip(video_data)
        for p in self.preprocess:
            if p is not None:
                video_data = p(video_data)
        return video_data


class Normalise:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, video_data):
        clip = _clip(video_data)
        if len(clip.affine) >= 4:
            raise NotImplementedError("at most 4 Normalise steps per clip")
        clip.affine.append((float(self.mean), float(self.std)))
        return clip

    def __repr__(self):
        return self.__class__.__name__ + "(mean={0}, std={1})".format(self.mean, self.std)


class TimeMasking:
    """video_transforms.py:82-121: one mask per second of video, each of 0..max_frames frames, filled with the mean frame."""

    def __init__(self, fps=25.0, max_frames: Optional[int] = None, max_seconds: Optional[float] = None):
        assert max_frames or max_seconds
        self.fps, self.max_frames, self.max_seconds = fps, max_frames, max_seconds

    def __call__(self, video_data):
        clip = _clip(video_data)
        max_frames = int(self.fps * self.max_seconds) if self.max_frames is None else self.max_frames
        video_length = clip.shape[0]
        num_seconds = int(video_length / self.fps)
        masked = np.zeros(video_length, dtype=np.uint8) if clip.masked is None else clip.masked
        for _ in range(num_seconds):
            mask_length = random.randint(0, max_frames)
            if mask_length > 0:
                offset = random.randint(0, video_length - mask_length)
                masked[offset:offset + mask_length] = 1
        clip.masked = masked
        return clip


class CenterCrop:
    def __init__(self, crop_size):
        self.crop_size = crop_size

    def __call__(self, video_data):
        clip = _clip(video_data)
        _, h, w = clip.shape
        th, tw = self.crop_size
        delta_w = int(round((w - tw)) / 2.)
        delta_h = int(round((h - th)) / 2.)
        clip.crop(delta_h, delta_w, th, tw)
        return clip

    def __repr__(self):
        return self.__class__.__name__ + "(size={0})".format(self.crop_size)


class RandomCrop:
    """torchvision.transforms.RandomCrop(size) on a [T, H, W] tensor (no padding): one window for the whole clip, drawn as
    torchvision's get_params does (two ``torch.randint`` calls on the host generator: top, then left)."""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, video_data):
        clip = _clip(video_data)
        _, h, w = clip.shape
        th, tw = self.size
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        if w == tw and h == th:
            return clip
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        clip.crop(i, j, th, tw)
        return clip


class RandomHorizontalFlip:
    """torchvision.transforms.RandomHorizontalFlip(p): ``torch.rand(1) < p`` mirrors the width axis of every frame."""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, video_data):
        clip = _clip(video_data)
        if torch.rand(1) < self.p:
            clip.flipped = not clip.flipped
        return clip


class VideoSpeedRate:
    """video_transforms.py:147-185: keeps frames int(linspace(0, T, T / rate, endpoint=False))."""

    def __init__(self, speed_rate=1.0):
        self._speed_rate = speed_rate

    def __call__(self, x):
        clip = _clip(x)
        if self._speed_rate <= 0:
            raise ValueError("speed_rate should be greater than zero.")
        if self._speed_rate == 1.:
            return clip
        if clip.masked is not None:
            raise NotImplementedError("VideoSpeedRate after TimeMasking")
        old_length = clip.shape[0]
        new_length = int(old_length / self._speed_rate)
        new_indices = list(map(int, np.linspace(start=0, stop=old_length, num=new_length, endpoint=False)))
        base = clip.index if clip.index is not None else list(range(old_length))
        clip.index = [base[i] for i in new_indices]
        return clip
