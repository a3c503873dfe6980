"""Optimizer / training-loop row.  CPU: the oracle restatement vs the golden generated from the reference's
own src/schedulers/noam.py, and the product harness's step cadence vs the oracle loop.  GPU: the fused HIP Adam step."""
import numpy as np
import pytest
import torch

from helpers import golden, rel_err
from oracle.model import synth
from oracle.optim import get_noam_oracle, training_oracle

SHAPES = [(37, 19), (256,), (5, 3, 7), (1,)]


def _run(opt, params, device="cpu"):
    rates = []
    for step in range(12):
        opt.zero_grad()
        for micro in range(3):
            for i, p in enumerate(params):
                g = (synth(tuple(p.shape), seed=1000 + 100 * step + 10 * micro + i) / 3).to(device)
                p.grad = g if p.grad is None else p.grad + g
        opt.step()
        rates.append(opt._rate)
    return rates


def test_noam_adam_oracle_matches_reference():
    g = golden("noam_adam")
    params = [torch.nn.Parameter(synth(s, seed=121 + i)) for i, s in enumerate(SHAPES)]
    rates = _run(get_noam_oracle(params, 1.6, 256, 5), params)
    assert np.allclose(rates, g["rates"], rtol=1e-12)
    for i, p in enumerate(params):
        assert rel_err(p, g[f"p{i}"]) < 1e-6


class _Toy(torch.nn.Module):
    """stands in for an E2E model: forward(**batch) -> (loss, stats, weight)"""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 3)

    def forward(self, x, y):
        loss = (self.lin(x) - y).square().mean()
        return loss, {"cer_ctc": torch.tensor(0.25)}, torch.tensor(x.shape[0])


@pytest.mark.parametrize("nbatch,accum", [(7, 3), (6, 2), (5, 8)])
def test_training_loop_cadence_matches_reference_loop(nbatch, accum):
    """same optimizer on both sides (torch Adam under the Noam wrapper): the product harness must step on the same
    micro-batches and return the same epoch loss as the restated reference loop"""
    import sys, os
    from helpers import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr.train import NoamScheduler, training, validation
    loader = [dict(x=synth((4, 6), seed=200 + i), y=synth((4, 3), seed=300 + i)) for i in range(nbatch)]
    torch.manual_seed(0)
    a, b = _Toy(), _Toy()
    b.load_state_dict(a.state_dict())
    oa = get_noam_oracle(a.parameters(), 1.6, 256, 4)
    ob = NoamScheduler(256, 1.6, 4, torch.optim.Adam(b.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
    la = training_oracle(a, loader, oa, None, accum)
    lb = training(b, loader, ob, None, accum, device="cpu")
    assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    assert oa._step == ob._step == -(-nbatch // accum)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
    vl, vc = validation(b, loader, device="cpu")
    assert vc == 25.0 and vl > 0


@pytest.mark.gpu
def test_fused_adam_matches_reference_golden():
    from tavsr.train import get_noam_scheduler
    g = golden("noam_adam")
    params = [torch.nn.Parameter(synth(s, seed=121 + i).cuda()) for i, s in enumerate(SHAPES)]
    opt = get_noam_scheduler(params, 1.6, 256, 5)
    rates = _run(opt, params, device="cuda")
    assert np.allclose(rates, g["rates"], rtol=1e-12)
    for i, p in enumerate(params):
        assert rel_err(p.detach().cpu(), g[f"p{i}"]) < 2e-6
        assert p.data_ptr() >= opt.optimizer.flat.data_ptr()      # parameters live in the flat buffer


@pytest.mark.gpu
def test_training_epoch_on_gpu_model():
    """two optimizer steps of the real ASR model through the harness: loss finite and decreasing on a repeated batch"""
    import argparse
    from helpers import asr_conf
    from oracle.model import fill_parameters_
    from tavsr.tasks.asr import ASRTask
    from tavsr.train import get_noam_scheduler, training
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=2, dec_blocks=1)))
    fill_parameters_(model, seed=7)
    model = model.cuda()
    text = synth((2, 6), seed=9, kind="int", lo=1, hi=40)
    batch = dict(speech=synth((2, 100, 80), seed=8), speech_lengths=torch.tensor([100, 76]), text=text,
                 text_lengths=torch.tensor([6, 4]))
    opt = get_noam_scheduler(model.parameters(), 0.05, 256, 10)   # rates 1e-4 .. 3e-4: a sane Adam step on random weights
    l1 = training(model, [batch, batch], opt, None, 2)
    l2 = training(model, [batch, batch], opt, None, 2)
    l3 = training(model, [batch, batch], opt, None, 2)
    assert np.isfinite([l1, l2, l3]).all() and l3 < l1
