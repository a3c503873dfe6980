"""Log-mel frontend + SpecAug on the HIP path against the CPU oracle (torch.stft / F.interpolate
restatements of espnet2 DefaultFrontend / SpecAug) and the committed BASELINE-config-1 golden."""
import argparse

import numpy as np
import pytest
import torch

from helpers import TOKENS_EN, asr_conf, golden, max_rel
from oracle import leaves as L
from oracle.model import fill_parameters_, synth

pytestmark = pytest.mark.gpu


def test_logmel_frontend_matches_oracle_ragged():
    from tavsr.frontend.default import DefaultFrontend
    kw = dict(n_fft=512, win_length=400, hop_length=160)
    ref, hip = L.DefaultFrontend(**kw), DefaultFrontend(**kw)
    wav = 0.1 * synth((3, 16000 + 123), seed=5, kind="uniform")
    lens = torch.tensor([16123, 12000, 8001])
    for i, le in enumerate(lens):
        wav[i, le:] = 0.0
    fr, lr = ref(wav, lens)
    fh, lh = hip(wav.cuda(), lens.cuda())
    assert lh.cpu().tolist() == lr.tolist()
    assert fh.shape == fr.shape
    # log-mel values are O(1..10): absolute agreement of the fp32 DFT-by-GEMM with torch's FFT
    assert float((fh.cpu() - fr).abs().max()) < 2e-4
    assert max_rel(fh.cpu(), fr) < 1e-4
    for i, le in enumerate(lr.tolist()):
        assert float(fh[i, le:].abs().max()) == 0.0 if le < fh.shape[1] else True


def test_cfg1_wav_to_greedy_ids_matches_reference_golden():
    """BASELINE configs[0]: 2 s synthetic WAV -> log-mel -> UtteranceMVN -> 6L encoder -> CTC greedy, ids bit-identical."""
    from tavsr.tasks.asr import ASRTask
    g = golden("cfg1_wav_greedy")
    conf = asr_conf(num_blocks=6, dec_blocks=1)
    conf["input_size"] = None
    conf["token_list"] = TOKENS_EN
    model = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(model, seed=51)
    model = model.cuda()
    wav = (0.1 * synth((1, 32000), seed=52, kind="uniform")).cuda()
    with torch.no_grad():
        feats, flens = model.frontend(wav, torch.tensor([32000]).cuda())
        ids, hyp, hyp_len = model.ctc_greedy(wav, torch.tensor([32000]).cuda())
    assert int(flens[0]) == 201
    assert max_rel(feats.cpu(), g["feats"]) < 1e-4
    assert hyp[0, : int(hyp_len[0])].cpu().tolist() == g["hyp"].tolist()


@pytest.mark.parametrize("ragged", [False, True])
def test_specaug_matches_oracle_with_same_draws(ragged):
    from tavsr.specaug.specaug import SpecAug
    kw = dict(apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True,
              freq_mask_width_range=[0, 27], num_freq_mask=2, apply_time_mask=True,
              time_mask_width_ratio_range=[0.0, 0.05], num_time_mask=5)
    ref, hip = L.SpecAug(**kw), SpecAug(**kw)
    x = synth((4, 400, 80), seed=11)
    lens = torch.tensor([400, 333, 250, 400]) if ragged else torch.tensor([400] * 4)
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        yr, _ = ref(x.clone(), lens)
        torch.manual_seed(seed)
        yh, _ = hip(x.clone().cuda(), lens.cuda())
        assert yh.shape == yr.shape
        assert torch.equal(yh.cpu() == 0, yr == 0)                  # the same bands are zeroed
        assert float((yh.cpu() - yr).abs().max()) < 1e-5            # bicubic weights in fp32 on both sides


def test_specaug_short_utterance_is_not_warped_and_eval_skips_it():
    from tavsr.specaug.specaug import SpecAug
    sa = SpecAug(apply_time_warp=True, time_warp_window=5, apply_freq_mask=False, apply_time_mask=False)
    x = synth((2, 10, 8), seed=3).cuda()          # t - window <= window: espnet2 returns the input
    y, _ = sa(x.clone(), torch.tensor([10, 10]))
    assert torch.equal(x, y)


def test_train_step_from_waveforms_with_specaug():
    """the full recipe surface: waveform batch -> frontend -> SpecAug -> MVN -> model loss/backward (finite, stochastic)"""
    from tavsr.tasks.asr import ASRTask
    import yaml
    from helpers import ASR_YAML
    conf = asr_conf(num_blocks=2, dec_blocks=1)
    conf["input_size"] = None
    ref = yaml.safe_load(open(ASR_YAML))
    conf["specaug"], conf["specaug_conf"] = ref["specaug"], ref["specaug_conf"]
    conf["token_list"] = TOKENS_EN
    model = ASRTask.build_model(argparse.Namespace(**conf))
    fill_parameters_(model, seed=7)
    model = model.cuda().train()
    wav = (0.1 * synth((2, 16000), seed=8, kind="uniform")).cuda()
    lens = torch.tensor([16000, 12800]).cuda()
    wav[1, 12800:] = 0
    text = synth((2, 6), seed=9, kind="int", lo=1, hi=39).cuda()
    tl = torch.tensor([6, 4]).cuda()
    text[1, 4:] = -1
    torch.manual_seed(0)
    l1 = model(wav, lens, text, tl)[0]
    l1.backward()
    torch.manual_seed(1)
    l2 = model(wav, lens, text, tl)[0]
    assert np.isfinite(float(l1)) and np.isfinite(float(l2)) and float(l1) != float(l2)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
