"""``SpecAug`` - drop-in for espnet2.asr.specaug.specaug.SpecAug with the reference's recipe
(configs/ASR/branchformer_transformer+ctc_english.yaml:21-37; applied train-only at src/models/espnet_model.py:383-385):
bicubic time warp (one warp per batch), ``num_freq_mask`` frequency bands and ``num_time_mask`` time bands per
utterance, zero fill.  The random draws follow espnet2's order and distributions (torch.randint) on the HOST
generator - so that the oracle can replay them - and only the band tables travel to the device; warp and masks are
``csrc/frontend.hip`` kernels."""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch

from .. import ops


class SpecAug(torch.nn.Module):
    def __init__(self, apply_time_warp: bool = True, time_warp_window: int = 5, time_warp_mode: str = "bicubic",
                 apply_freq_mask: bool = True, freq_mask_width_range: Union[int, Sequence[int]] = (0, 20),
                 num_freq_mask: int = 2, apply_time_mask: bool = True,
                 time_mask_width_range: Optional[Union[int, Sequence[int]]] = None,
                 time_mask_width_ratio_range: Optional[Union[float, Sequence[float]]] = None, num_time_mask: int = 2):
        if not apply_time_warp and (not apply_time_mask) and (not apply_freq_mask):
            raise ValueError("Either one of time_warp, time_mask, or freq_mask should be applied")
        if apply_time_mask and (time_mask_width_range is not None) and (time_mask_width_ratio_range is not None):
            raise ValueError('Either one of "time_mask_width_range" or "time_mask_width_ratio_range" can be used')
        if apply_time_warp and time_warp_mode != "bicubic":
            raise ValueError("only the bicubic time warp of the shipped recipe is implemented")
        super().__init__()
        self.apply_time_warp, self.apply_freq_mask, self.apply_time_mask = apply_time_warp, apply_freq_mask, apply_time_mask
        self.time_warp_window = time_warp_window
        self.freq_mask_width_range = self._range(freq_mask_width_range)
        self.num_freq_mask, self.num_time_mask = num_freq_mask, num_time_mask
        self.time_mask_width_range = None if time_mask_width_range is None else self._range(time_mask_width_range)
        self.time_mask_width_ratio_range = None
        if time_mask_width_ratio_range is not None:
            r = time_mask_width_ratio_range
            self.time_mask_width_ratio_range = (0.0, float(r)) if isinstance(r, float) else tuple(float(v) for v in r)
        if apply_time_mask and self.time_mask_width_range is None and self.time_mask_width_ratio_range is None:
            raise ValueError('Either one of "time_mask_width_range" or "time_mask_width_ratio_range" should be used.')

    @staticmethod
    def _range(r):
        r = (0, r) if isinstance(r, int) else tuple(r)
        if len(r) != 2 or r[0] > r[1]:
            raise TypeError(f"mask_width_range must be (min, max) with min <= max: {r}")
        return r

    @staticmethod
    def _bands(B, D, width_range, num_mask):
        """espnet2 mask_along_axis draws: lengths in [lo, hi), positions in [0, max(1, D - max length))."""
        mask_length = torch.randint(width_range[0], width_range[1], (B, num_mask))
        mask_pos = torch.randint(0, max(1, D - int(mask_length.max())), (B, num_mask))
        return mask_pos, mask_length

    def _draw_warp(self, t):
        """espnet2 time_warp's draws for a t-frame utterance: (center, warped), or (0, 0) when it is too short."""
        w = self.time_warp_window
        if t - w <= w:
            return 0, 0
        center = int(torch.randint(w, t - w, (1,))[0])
        warped = int(torch.randint(center - w, center + w, (1,))[0]) + 1
        return center, warped

    def _time_warp(self, x, x_lengths):
        """espnet2 TimeWarp.forward: one warp for the batch when all lengths agree, else one per utterance on its own
        frames (results zero-padded to the longest)."""
        B, T, F = x.shape
        lens = None if x_lengths is None else [int(v) for v in x_lengths]
        if lens is None or all(le == lens[0] for le in lens):
            c, wp = self._draw_warp(T)
            cs, ws, ls = [c] * B, [wp] * B, [T] * B
        else:
            cs, ws, ls = [], [], lens
            for le in lens:
                c, wp = self._draw_warp(le)
                cs.append(c)
                ws.append(wp)
            if max(lens) < T:
                x = x[:, :max(lens)].contiguous()
        dev = x.device
        return ops.time_warp(x, *(torch.tensor(v, dtype=torch.int64).to(dev) for v in (cs, ws, ls)))

    def forward(self, x: torch.Tensor, x_lengths: torch.Tensor = None):
        assert x.dim() == 3, "features are (Batch, Length, Freq)"
        B, T, F = x.shape
        x = x.contiguous()
        if self.apply_time_warp:
            x = self._time_warp(x, x_lengths)
            T = x.shape[1]
        else:
            x = x.clone()
        fpos = flen = tpos = tlen = None
        if self.apply_freq_mask:
            fpos, flen = self._bands(B, F, self.freq_mask_width_range, self.num_freq_mask)
        if self.apply_time_mask:
            if self.time_mask_width_range is not None:
                rng = self.time_mask_width_range
            else:      # MaskAlongAxisVariableMaxWidth: widths up to a ratio of this batch's length
                lo = int(T * self.time_mask_width_ratio_range[0])
                hi = min(T, int(T * self.time_mask_width_ratio_range[1]))
                rng = (lo, hi)
            if rng[1] > rng[0]:
                tpos, tlen = self._bands(B, T, rng, self.num_time_mask)
        dev = x.device
        to = lambda t: None if t is None else t.to(dev)
        ops.specaug_mask_(x, to(fpos), to(flen), to(tpos), to(tlen))
        return x, x_lengths
