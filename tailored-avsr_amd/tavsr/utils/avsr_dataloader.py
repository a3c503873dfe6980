"""Batch assembly on the MI355X: the collate functions of src/utils/avsr_dataloader.py:40-142 with the same
names, arguments and batch dictionaries, producing the padded batch directly in HBM.

* audio: ``audio_transforms`` (tavsr.transforms.AddNoise) on the device, [1, T] -> [T, 1], cropped to a multiple of 640
  samples (:50, :114), padded with ``config.model_conf['ignore_id']`` (:66, :131).
* video: ``video_transforms`` record what they do (tavsr.transforms.video_transforms); every clip is rendered ONCE,
  straight into its row of the padded [B, Tmax, h, w] tensor (crop, mirror, normalisation, mean-frame masks and the
  padding frames in one pass: ``tavsr_video_prep``).
* text: token ids through the caller's tokenizer / converter (host strings), padded with ignore_id, int64.
Samples may hold CPU tensors (they are moved once, uint8 lip frames as uint8) or device tensors."""
from __future__ import annotations

import torch

from ..transforms.video_transforms import VideoClip

DEVICE = "cuda"


def _pad_stack(seqs, pad_value, dtype, device):
    """nn.utils.rnn.pad_sequence(batch_first=True) of already-resident sequences (index plumbing: B narrow copies)."""
    tmax = max(int(s.shape[0]) for s in seqs)
    out = torch.full((len(seqs), tmax) + tuple(seqs[0].shape[1:]), pad_value, dtype=dtype, device=device)
    for b, s in enumerate(seqs):
        out[b, : s.shape[0]] = s
    return out


def _lengths(values):
    """int64 lengths in HBM with their maximum attached as a host value: the model's "cut to the longest" (espnet_model.py:372)
    then needs no device read (tavsr.models.espnet_model.host_max)."""
    values = [int(v) for v in values]
    t = torch.tensor(values, dtype=torch.int64).to(DEVICE)
    t._tavsr_max = (t._version, max(values) if values else 0)      # espnet_model.host_max checks the version counter
    return t


def _audio(sample, audio_transforms):
    audio = sample["audio"].to(DEVICE)
    audio = audio_transforms(audio) if audio_transforms else audio
    audio = audio.transpose(1, 0)
    return audio[: audio.shape[0] // 640 * 640, :]


def _videos(data, video_transforms, pad_value):
    clips = []
    for sample in data:
        v = sample["video"]
        v = v if isinstance(v, VideoClip) else VideoClip(v.to(DEVICE))
        clips.append(video_transforms(v) if video_transforms else v)
    tmax = max(c.shape[0] for c in clips)
    h, w = clips[0].shape[1:]
    if any(c.shape[1:] != (h, w) for c in clips):
        raise ValueError("clips of one batch must have one frame size after the transforms")
    out = torch.empty((len(clips), tmax, h, w), dtype=torch.float32, device=DEVICE)
    for b, c in enumerate(clips):
        c.render(out[b], pad_value)
    return out, _lengths(c.shape[0] for c in clips)


def _texts(data, tokenizer, converter, pad_value):
    ids = [list(converter.tokens2ids(tokenizer.text2tokens(s["transcription"]))) for s in data]
    tmax = max(len(t) for t in ids)
    text = torch.tensor([t + [pad_value] * (tmax - len(t)) for t in ids], dtype=torch.int64).to(DEVICE)
    return text, _lengths(len(t) for t in ids)


def asr_data_processing(data, audio_transforms, video_transforms, tokenizer, converter, config):
    pad = config.model_conf["ignore_id"]
    speech = [_audio(s, audio_transforms) for s in data]
    text, text_lengths = _texts(data, tokenizer, converter, pad)
    return {"sample_id": [s["sample_id"] for s in data],
            "speech": _pad_stack(speech, pad, torch.float32, DEVICE),
            "speech_lengths": _lengths(a.shape[0] for a in speech),
            "text": text, "text_lengths": text_lengths, "refs": [s["transcription"] for s in data]}


def vsr_data_processing(data, audio_transforms, video_transforms, tokenizer, converter, config):
    pad = config.model_conf["ignore_id"]
    speech, speech_lengths = _videos(data, video_transforms, float(pad))
    text, text_lengths = _texts(data, tokenizer, converter, pad)
    return {"sample_id": [s["sample_id"] for s in data], "speech": speech, "speech_lengths": speech_lengths,
            "text": text, "text_lengths": text_lengths, "refs": [s["transcription"] for s in data]}


def avsr_data_processing(data, audio_transforms, video_transforms, tokenizer, converter, config):
    pad = config.model_conf["ignore_id"]
    # per sample: audio first, then video (the order in which the reference's transforms draw their random numbers)
    audio, clips = [], []
    for sample in data:
        audio.append(_audio(sample, audio_transforms))
        v = sample["video"]
        v = v if isinstance(v, VideoClip) else VideoClip(v.to(DEVICE))
        clips.append(video_transforms(v) if video_transforms else v)
    video, video_lengths = _videos([{"video": c} for c in clips], None, float(pad))
    text, text_lengths = _texts(data, tokenizer, converter, pad)
    return {"sample_id": [s["sample_id"] for s in data],
            "audio": _pad_stack(audio, pad, torch.float32, DEVICE),
            "audio_lengths": _lengths(a.shape[0] for a in audio),
            "video": video, "video_lengths": video_lengths,
            "text": text, "text_lengths": text_lengths, "refs": [s["transcription"] for s in data]}
