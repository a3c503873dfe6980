"""Training harness: counterpart of the reference's loop (avsr_main.py:27-82), optimizer factory
(src/utils/scheduler.py:6-45) and Noam wrapper (src/schedulers/noam.py:11-81) - SURVEY section 8 row a17.

* ``NoamScheduler`` / ``get_noam_scheduler`` keep the reference's names, arguments and ``state_dict`` layout.
* ``FusedAdam`` has torch.optim.Adam's arithmetic (betas (0.9, 0.98), eps 1e-9 as the reference constructs it) but one
  HIP kernel over flat fp32 buffers instead of ~700 small tensor updates: parameters are re-homed as views of one flat
  buffer, gradients are packed next to them and ``tavsr_adam_step`` updates everything in one pass.
* ``training`` / ``validation`` reproduce the reference's semantics: loss / accum_grad, optimizer step every
  ``accum_grad`` micro-batches and at the end of the loader, no gradient clipping, returned epoch loss
  ``sum(loss_i) / (len(loader) / accum_grad)`` (Q10).  Losses are summed on the device: one host sync per epoch
  instead of one ``.item()`` per micro-batch.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch

from . import _lib as L
from . import dp


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps) semantics on flat buffers (no weight decay / amsgrad: the reference
    uses neither).  ``param_groups`` is kept (one group) so that a learning-rate wrapper can set ``lr``."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.98), eps: float = 1e-9):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        L.require_cuda(*self.params)
        # every parameter starts on a 256-byte boundary of the flat buffer (the kernels want 16-byte aligned
        # operands; the padding elements have zero gradient and stay zero)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += -(-p.numel() // 64) * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):   # re-home every parameter as a view (values preserved)
                k = p.numel()
                self.flat[off: off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off: off + k].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.grad = torch.zeros_like(self.flat)
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps)]
        self._step = 0
        self._steps = [0] * len(self.params)      # per-parameter step counts (torch.optim.Adam keeps one per parameter)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """torch.optim.Adam semantics incl. its treatment of missing gradients: a parameter whose ``grad`` is None is
        skipped (no moment decay, no update) and keeps its own step count for the bias correction.  Gradients are packed
        by multi-tensor launches (24 tensors each); the update is one launch per maximal run of neighbouring parameters
        that have a gradient and the same step count - ONE launch in the usual case where every parameter has one."""
        from . import ops
        g = self.param_groups[0]
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        dst = [self.grad[self.offsets[i]: self.offsets[i] + self.params[i].numel()] for i in live]
        src = [self.params[i].grad.contiguous().reshape(-1) for i in live]
        for j in range(0, len(live), 24):
            ops.multi_copy_(dst[j: j + 24], src[j: j + 24])
        for i in live:
            self._steps[i] += 1
        self._step += 1
        runs, k = [], 0
        while k < len(live):                 # maximal runs of adjacent parameters with equal step counts
            a = b = k
            while b + 1 < len(live) and live[b + 1] == live[b] + 1 and self._steps[live[b + 1]] == self._steps[live[a]]:
                b += 1
            runs.append((live[a], live[b]))
            k = b + 1
        for ia, ib in runs:
            lo = self.offsets[ia]
            hi = self.offsets[ib + 1] if ib + 1 < len(self.params) else self.flat.numel()
            L.check(L.lib().tavsr_adam_step(C.c_void_p(self.flat.data_ptr() + 4 * lo), C.c_void_p(self.grad.data_ptr() + 4 * lo),
                                            C.c_void_p(self.exp_avg.data_ptr() + 4 * lo),
                                            C.c_void_p(self.exp_avg_sq.data_ptr() + 4 * lo), C.c_int64(hi - lo),
                                            C.c_float(g["lr"]), C.c_float(g["betas"][0]), C.c_float(g["betas"][1]),
                                            C.c_float(g["eps"]), C.c_int64(self._steps[ia]), C.c_float(grad_scale), L.stream()),
                    "tavsr_adam_step")

    def state_dict(self):
        return dict(step=self._step, steps=list(self._steps), exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self._step = sd["step"]
        self._steps = list(sd.get("steps", [sd["step"]] * len(self.params)))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])


class NoamScheduler(object):
    """Optim wrapper that implements rate (src/schedulers/noam.py:11-69)."""

    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer = optimizer
        self._step = 0
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self._rate = 0

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def step(self):
        self._step += 1
        rate = self.rate()
        for p in self.optimizer.param_groups:
            p["lr"] = rate
        self._rate = rate
        self.optimizer.step()

    def rate(self, step=None):
        if step is None:
            step = self._step
        return self.factor * self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5))

    def zero_grad(self):
        self.optimizer.zero_grad()

    def state_dict(self):
        return {"_step": self._step, "warmup": self.warmup, "factor": self.factor, "model_size": self.model_size,
                "_rate": self._rate, "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        for key, value in state_dict.items():
            if key == "optimizer":
                self.optimizer.load_state_dict(state_dict["optimizer"])
            else:
                setattr(self, key, value)


def get_noam_scheduler(model_params, factor, d_model, warmup):
    """src/schedulers/noam.py:72-81: Adam(lr=0, betas=(0.9, 0.98), eps=1e-9) under the Noam rate."""
    return NoamScheduler(d_model, factor, warmup, FusedAdam(model_params, lr=0, betas=(0.9, 0.98), eps=1e-9))


def set_optimizer(config, e2e, train_loader):
    """src/utils/scheduler.py:6-45 for the ``scheduler: noam`` recipes (every shipped config)."""
    ts = config.training_settings
    if ts["scheduler"] != "noam":
        raise RuntimeError("The scheduler should be specified as 'noam' (onecycle is not used by any shipped config)")
    return get_noam_scheduler(e2e.parameters(), ts["noam_factor"], config.encoder_conf["output_size"], ts["warmup_steps"]), None


def _to_device(batch, device):
    return {k: v.to(device=device, non_blocking=True) if hasattr(v, "to") else v for k, v in batch.items()}


def training(e2e, train_loader, optimizer, scheduler, accum_grad, device="cuda", buckets: Optional[dp.GradBuckets] = None):
    """One epoch (avsr_main.py:27-58).  ``buckets`` (tavsr.dp.GradBuckets) adds the data-parallel gradient average
    right before each optimizer step - the only collective on the path."""
    e2e.train()
    total = None
    optimizer.zero_grad()
    n = len(train_loader)
    if buckets is not None:
        buckets.attach_overlap_hooks()      # a bucket's all-reduce leaves while the rest of the backward pass still runs
    for batch_idx, batch in enumerate(train_loader):
        batch = _to_device(batch, device)
        loss = e2e(**batch)[0] / accum_grad
        stepping = ((batch_idx + 1) % accum_grad == 0) or (batch_idx + 1 == n)
        if buckets is not None:             # gradients are exchanged once per optimizer step: the hooks launch buckets
            buckets.overlap = stepping      # only during the window's LAST micro-batch (every hook fires once in it)
        loss.backward()
        if stepping:
            if buckets is not None:
                buckets.allreduce_mean()
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        d = loss.detach().reshape(())
        total = d if total is None else total + d
    return float(total) / (n / accum_grad)


def validation(e2e, data_loader, device="cuda"):
    """avsr_main.py:60-82: mean loss and mean CTC character error rate (%) over the loader, rounded to 3 decimals."""
    e2e.eval()
    data_loss, data_cer = 0.0, 0.0
    with torch.no_grad():
        for batch in data_loader:
            batch = _to_device(batch, device)
            loss, stats, weight = e2e(**batch)
            data_loss += float(loss)
            data_cer += float(stats["cer_ctc"]) * 100.0
    return round(data_loss / len(data_loader), 3), round(data_cer / len(data_loader), 3)
