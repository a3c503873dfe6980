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
def test_fused_adam_skips_missing_gradients_like_torch_adam():
    """a parameter whose grad is None is skipped by torch.optim.Adam (no moment decay, no update, its own step count);
    the fused optimizer must do the same: parameters 1 and 3 miss steps 2-4, parameter 0 misses step 6."""
    from tavsr.train import FusedAdam
    ref = [torch.nn.Parameter(synth(s, seed=121 + i).cuda()) for i, s in enumerate(SHAPES)]
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    oa = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.98), eps=1e-9)
    ob = FusedAdam(got, lr=3e-3, betas=(0.9, 0.98), eps=1e-9)
    for step in range(8):
        missing = {1, 3} if 2 <= step <= 4 else ({0} if step == 6 else set())
        for i, (pa, pb) in enumerate(zip(ref, got)):
            g = None if i in missing else synth(tuple(pa.shape), seed=500 + 10 * step + i).cuda()
            pa.grad = None if g is None else g.clone()
            pb.grad = g
        oa.step()
        ob.step()
    for i, (pa, pb) in enumerate(zip(ref, got)):
        assert rel_err(pb.detach().cpu(), pa.detach().cpu()) < 2e-6, i


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


def test_checkpoint_save_load_average_roundtrip(tmp_path):
    """src/utils/model_checkpoint.py semantics: plain state_dict files, module-wise load, entry-wise mean."""
    import argparse

    import torch
    from helpers import TOKENS_EN, asr_conf
    from oracle.model import fill_parameters_
    from tavsr.tasks.asr import ASRTask
    from tavsr.utils import model_checkpoint as MC

    def build(seed):
        conf = asr_conf(num_blocks=1, dec_blocks=1)
        conf["token_list"] = TOKENS_EN
        m = ASRTask.build_model(argparse.Namespace(**conf))
        fill_parameters_(m, seed=seed)
        return m

    a, b = build(1), build(2)
    pa, pb = MC.save_model(str(tmp_path), a, "epoch001"), MC.save_model(str(tmp_path), b, "epoch002")
    assert pa.endswith("models/model_epoch001.pth")
    sd = torch.load(pa)
    assert list(sd.keys()) == list(a.state_dict().keys())
    c = build(3)
    MC.average_model(c, [pa, pb])
    for k, v in c.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.allclose(v, (a.state_dict()[k] + b.state_dict()[k]) / 2, atol=1e-7), k
    d = build(4)
    MC.load_e2e(d, ["encoder", "ctc"], pa, ctc_weight=0.1)
    assert all(torch.equal(v, a.encoder.state_dict()[k]) for k, v in d.encoder.state_dict().items())
    assert all(torch.equal(v, a.ctc.state_dict()[k]) for k, v in d.ctc.state_dict().items())
    assert not torch.equal(d.decoder.output_layer.weight, a.decoder.output_layer.weight)
    MC.freeze_e2e(d, ["encoder", "ctc"], 0.1)
    assert not any(p.requires_grad for p in d.encoder.parameters())
    assert all(p.requires_grad for p in d.ctc.parameters())          # the reference's typo: the CTC head stays trainable
    MC.save_val_stats(str(tmp_path), [(pa, 12.5), (pb, 11.0)])
    assert open(tmp_path / "val_stats.csv").read().splitlines()[0] == ",model_check_path,cer"


@pytest.mark.parametrize("anneal,total", [("linear", 40), ("cos", 37), ("linear", 7)])
def test_one_cycle_schedule_equals_torch(anneal, total):
    """tavsr.train.OneCycleLR (for optimizers that only expose param_groups) vs torch.optim.lr_scheduler.OneCycleLR on a
    torch Adam: learning rate and cycled beta1 at every step, and the error after the last one."""
    from tavsr.train import OneCycleLR

    class Bare:           # what the flat-buffer optimizer exposes
        def __init__(self):
            self.param_groups = [dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)]

    p = torch.nn.Parameter(torch.zeros(3))
    to = torch.optim.AdamW([p], 5e-4)
    ts = torch.optim.lr_scheduler.OneCycleLR(to, max_lr=5e-4, total_steps=total, anneal_strategy=anneal)
    bo = Bare()
    bs = OneCycleLR(bo, max_lr=5e-4, total_steps=total, anneal_strategy=anneal)
    for step in range(total):
        assert bo.param_groups[0]["lr"] == pytest.approx(to.param_groups[0]["lr"], rel=1e-12), step
        assert bo.param_groups[0]["betas"][0] == pytest.approx(to.param_groups[0]["betas"][0], rel=1e-12), step
        assert bs.get_last_lr() == pytest.approx(ts.get_last_lr(), rel=1e-12)
        if step < total - 1:
            to.step()
            ts.step()
            bs.step()
    ts.step()
    bs.step()                      # step number total_steps is still allowed, the next one is not
    with pytest.raises(ValueError):
        bs.step()


def test_set_optimizer_follows_the_reference_factory():
    """src/utils/scheduler.py:6-45 on a bare namespace: which optimizer / scheduler objects come back for the recipes'
    (optimizer, scheduler) pairs, steps per epoch with gradient accumulation, and the error for an unknown scheduler."""
    import argparse
    from tavsr import train as T

    made = {}

    class FakeAdam:
        def __init__(self, params, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0):
            made.update(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
            self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]

    real = T.FusedAdam
    T.FusedAdam = FakeAdam
    try:
        e2e = torch.nn.Linear(4, 4)
        loader = list(range(50))
        ts = dict(optimizer="adamw", scheduler="onecycle", learning_rate=5e-4, accum_grad=16, epochs=10)
        opt, sch = T.set_optimizer(argparse.Namespace(training_settings=ts, encoder_conf=dict(output_size=256)), e2e, loader)
        assert made == dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
        assert isinstance(sch, T.OneCycleLR) and sch.total_steps == 10 * 4 and sch.anneal_strategy == "linear"
        ts = dict(optimizer="adam", scheduler="onecycle", learning_rate=1e-3, accum_grad=0, epochs=2)
        opt, sch = T.set_optimizer(argparse.Namespace(training_settings=ts, encoder_conf=dict(output_size=256)), e2e, loader)
        assert made == dict(lr=1e-3, betas=(0.9, 0.98), eps=10e-09, weight_decay=0.0) and sch.total_steps == 100
        ts = dict(optimizer="adam", scheduler="noam", noam_factor=1.6, warmup_steps=10000, accum_grad=4, learning_rate=1e-3)
        opt, sch = T.set_optimizer(argparse.Namespace(training_settings=ts, encoder_conf=dict(output_size=256)), e2e, loader)
        assert isinstance(opt, T.NoamScheduler) and sch is None and made["lr"] == 0 and made["eps"] == 1e-9
        ts = dict(optimizer="adam", scheduler="cosine", learning_rate=1e-3, accum_grad=1, epochs=1)
        with pytest.raises(RuntimeError):
            T.set_optimizer(argparse.Namespace(training_settings=ts, encoder_conf=dict(output_size=256)), e2e, loader)
    finally:
        T.FusedAdam = real


@pytest.mark.gpu
def test_fused_adamw_under_one_cycle_matches_torch():
    """the ``optimizer: adamw`` + ``scheduler: onecycle`` recipes: decoupled weight decay, lr and beta1 moved every step."""
    from tavsr.train import FusedAdam, OneCycleLR
    ref = [torch.nn.Parameter(synth(s, seed=121 + i).cuda()) for i, s in enumerate(SHAPES)]
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    oa = torch.optim.AdamW(ref, 5e-3)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=5e-3, total_steps=12, anneal_strategy="linear")
    ob = FusedAdam(got, 5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    sb = OneCycleLR(ob, max_lr=5e-3, total_steps=12, anneal_strategy="linear")
    for step in range(11):
        for i, (pa, pb) in enumerate(zip(ref, got)):
            g = synth(tuple(pa.shape), seed=900 + 10 * step + i).cuda()
            pa.grad, pb.grad = g.clone(), g
        oa.step()
        sa.step()
        ob.step()
        sb.step()
    for i, (pa, pb) in enumerate(zip(ref, got)):
        assert rel_err(pb.detach().cpu(), pa.detach().cpu()) < 2e-6, i
