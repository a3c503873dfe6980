"""Batch assembly: video / audio transforms and collate.  The oracle (oracle/data.py) against the fixture
the reference's own classes produced (oracle/gen_golden_data.py -> tests/golden/data_pipeline.npz); the HIP path
(tavsr.transforms, tavsr.utils.avsr_dataloader, through tavsr_video_prep / tavsr_add_noise) against both."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import ROOT

from oracle import data as D

G = np.load(os.path.join(ROOT, "tests", "golden", "data_pipeline.npz"))
CONFIG = types.SimpleNamespace(model_conf={"ignore_id": -1})


def digest(x, per_frame):
    x = x.cpu()
    flat = x.reshape(-1)
    sums = x.double().reshape(x.shape[0], x.shape[1], -1).sum(-1) if per_frame else x.double().reshape(x.shape[0], -1).sum(-1)
    return flat[:: max(1, flat.numel() // 2048)][:2048].numpy(), sums.numpy()


def pipelines(mod_v, add_noise):
    return {
        "eval": (mod_v.Compose([mod_v.Normalise(0.0, 250.0), mod_v.Normalise(D.MEAN, D.STD), mod_v.CenterCrop((88, 88))]), add_noise),
        "train": (mod_v.Compose([mod_v.Normalise(0.0, 250.0), mod_v.Normalise(D.MEAN, D.STD),
                                 mod_v.TimeMasking(fps=D.FPS, max_seconds=0.4), mod_v.RandomCrop((88, 88)),
                                 mod_v.RandomHorizontalFlip(p=0.5)]), None),
        "speed": (mod_v.Compose([mod_v.VideoSpeedRate(1.25), mod_v.Normalise(0.0, 250.0), mod_v.TimeMasking(fps=D.FPS, max_frames=6),
                                 mod_v.CenterCrop((80, 72))]), add_noise),
    }


def check_against_golden(b, key, tol):
    vs, vf = digest(b["video"], True)
    au, asum = digest(b["audio"], False)
    assert list(b["video"].shape) == list(G[key + "video_shape"]) and list(b["audio"].shape) == list(G[key + "audio_shape"])
    assert np.abs(vs - G[key + "video_samples"]).max() < tol
    assert np.abs(vf - G[key + "video_frame_sums"]).max() < tol * 88 * 88
    assert np.abs(au - G[key + "audio_samples"]).max() < tol
    assert np.abs(asum - G[key + "audio_sums"]).max() < 1e-4 * (1 + np.abs(G[key + "audio_sums"]).max())
    for f in ("audio_lengths", "video_lengths", "text", "text_lengths"):
        assert np.array_equal(b[f].cpu().numpy(), G[key + f]), f


@pytest.mark.parametrize("name", ["eval", "train", "speed"])
@pytest.mark.parametrize("seed", [11, 12])
def test_oracle_batches_equal_the_reference_generated_fixture(name, seed):
    vtr, atr = pipelines(D, D.AddNoise(D.make_noise(5), snr_target=5))[name]
    D.seed_all(seed)
    b = D.avsr_data_processing(D.make_samples(seed), atr, vtr, D.CharTokenizer(), D.CharConverter())
    check_against_golden(b, f"{name}_{seed}_", 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["eval", "train", "speed"])
@pytest.mark.parametrize("seed", [11, 12])
def test_hip_batches_equal_oracle_and_fixture(name, seed):
    from tavsr.transforms import audio_transforms as PA
    from tavsr.transforms import video_transforms as PV
    from tavsr.utils.avsr_dataloader import avsr_data_processing
    vtr, atr = pipelines(PV, PA.AddNoise(noise=D.make_noise(5), snr_target=5))[name]
    D.seed_all(seed)
    b = avsr_data_processing(D.make_samples(seed), atr, vtr, D.CharTokenizer(), D.CharConverter(), CONFIG)
    assert b["video"].is_cuda and b["audio"].is_cuda and b["text"].is_cuda
    check_against_golden(b, f"{name}_{seed}_", 2e-6)
    # and element for element against the oracle run with the same seeds (same masks, windows, flips, noise offsets)
    ovtr, oatr = pipelines(D, D.AddNoise(D.make_noise(5), snr_target=5))[name]
    D.seed_all(seed)
    want = D.avsr_data_processing(D.make_samples(seed), oatr, ovtr, D.CharTokenizer(), D.CharConverter())
    assert torch.equal(b["video_lengths"].cpu(), want["video_lengths"]) and torch.equal(b["text"].cpu(), want["text"])
    assert (b["video"].cpu() - want["video"]).abs().max() < 2e-6
    assert (b["audio"].cpu() - want["audio"]).abs().max() < 2e-6
    assert [s for s in b["refs"]] == [s["transcription"] for s in D.make_samples(seed)]


@pytest.mark.gpu
def test_video_clip_geometry_composes_in_any_order():
    """crop after mirror, mirror after crop, two crops: the recorded window equals slicing / flipping real tensors"""
    from tavsr.transforms import video_transforms as PV
    g = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, (9, 40, 50), generator=g, dtype=torch.uint8)
    clip = PV.VideoClip(raw.cuda())
    clip.flipped = True
    clip.crop(3, 5, 30, 40)
    clip.flipped = not clip.flipped
    clip.crop(2, 7, 20, 21)
    want = raw.float().flip(-1)[:, 3:33, 5:45].flip(-1)[:, 2:22, 7:28]
    assert torch.equal(clip.render().cpu(), want)
    out = torch.empty(12, 20, 21, device="cuda")
    clip.render(out, -1.0)
    assert torch.equal(out[:9].cpu(), want) and bool((out[9:] == -1.0).all())
    with pytest.raises(ValueError):
        clip.crop(0, 0, 30, 30)


@pytest.mark.gpu
def test_asr_and_vsr_collate_and_float_clips():
    from tavsr.transforms import video_transforms as PV
    from tavsr.utils.avsr_dataloader import asr_data_processing, vsr_data_processing
    samples = D.make_samples(21, n=2)
    b = asr_data_processing(samples, None, None, D.CharTokenizer(), D.CharConverter(), CONFIG)
    for i, s in enumerate(samples):
        L = s["audio"].shape[1] // 640 * 640
        assert int(b["speech_lengths"][i]) == L
        assert torch.equal(b["speech"][i, :L, 0].cpu(), s["audio"][0, :L]) and bool((b["speech"][i, L:] == -1).all())
    fl = [dict(s, video=s["video"].float()) for s in samples]                    # float32 frames take the same path
    tr = PV.Compose([PV.Normalise(0.0, 250.0), PV.CenterCrop((88, 88))])
    v8 = vsr_data_processing(samples, None, tr, D.CharTokenizer(), D.CharConverter(), CONFIG)
    vf = vsr_data_processing(fl, None, tr, D.CharTokenizer(), D.CharConverter(), CONFIG)
    assert torch.equal(v8["speech"], vf["speech"]) and v8["speech"].shape[2:] == (88, 88)


def test_oracle_resampling_of_a_sine_keeps_amplitude_and_scales_frequency():
    """SpeedRate's resampling (src/transforms/audio_transforms.py:141-178, sox "speed f" + "rate"): a 440 Hz tone played 1.1 times faster
    is a 484 Hz tone of the same amplitude and round(T / 1.1) samples (the oracle's windowed-sinc restatement; sox itself is absent)."""
    import numpy as np
    from oracle.data import resample_sinc
    fs, T = 16000, 4000
    x = np.sin(2 * np.pi * 440.0 * np.arange(T) / fs)
    for f in (0.9, 1.1):
        y = resample_sinc(x, f)
        assert len(y) == int(round(T / f))
        n = np.arange(len(y))
        want = np.sin(2 * np.pi * 440.0 * f * n / fs)
        mid = slice(200, len(y) - 200)                    # away from the clip edges (the filter sees zeros beyond them)
        assert np.abs(y[mid] - want[mid]).max() < 2e-3, (f, np.abs(y[mid] - want[mid]).max())


@pytest.mark.gpu
def test_speed_rate_transform_equals_the_oracle_resampling():
    import random

    import numpy as np
    from oracle.data import resample_sinc
    from tavsr.transforms.audio_transforms import SpeedRate
    torch.manual_seed(3)
    x = torch.randn(1, 6000, device="cuda")
    tr = SpeedRate(16000)
    seen = set()
    for seed in range(12):
        random.seed(seed)
        f = random.choice([0.9, 1.0, 1.1])               # the draw the transform will make
        random.seed(seed)
        y = tr(x)
        seen.add(f)
        if f == 1.0:
            assert y is x
            continue
        ref = resample_sinc(x.cpu().numpy(), f)
        assert y.shape == (1, len(ref))
        assert np.abs(y.cpu().numpy().reshape(-1) - ref).max() < 2e-5 * np.abs(ref).max() + 1e-6
    assert seen == {0.9, 1.0, 1.1}
