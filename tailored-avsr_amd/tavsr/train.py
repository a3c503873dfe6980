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
    """torch.optim.Adam(params, lr, betas, eps) semantics on flat buffers; ``weight_decay`` > 0 gives torch.optim.AdamW's
    decoupled decay (the ``optimizer: adamw`` recipes).  No amsgrad (the reference does not use it).  ``param_groups`` is
    kept (one group) so that a learning-rate wrapper or scheduler can set ``lr`` / ``betas``."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.98), eps: float = 1e-9,
                 weight_decay: float = 0.0):
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
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
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
        that have a gradient and the same step count - ONE launch in the usual case where every parameter has one.

        Under data parallelism ``GradBuckets.allreduce_mean`` replaces a missing gradient by zeros before the exchange (a
        layer skipped by stochastic depth on this rank may have run on another), so there every parameter is stepped every
        time - moment decay, decoupled weight decay and step count included - exactly as torch DDP + torch.optim.Adam
        behave; a single-process run skips such a parameter.  The shipped recipes set the skip rates to 0."""
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
            L.check(L.lib().tavsr_adamw_step(C.c_void_p(L.addr(self.flat, lo)), C.c_void_p(L.addr(self.grad, lo)),
                                             C.c_void_p(L.addr(self.exp_avg, lo)),
                                             C.c_void_p(L.addr(self.exp_avg_sq, lo)), C.c_int64(hi - lo),
                                             C.c_float(g["lr"]), C.c_float(g["betas"][0]), C.c_float(g["betas"][1]),
                                             C.c_float(g["eps"]), C.c_float(g.get("weight_decay", 0.0)),
                                             C.c_int64(self._steps[ia]), C.c_float(grad_scale), L.stream()),
                    "tavsr_adamw_step")

    def state_dict(self):
        return dict(step=self._step, steps=list(self._steps), exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self._step = sd["step"]
        steps = list(sd.get("steps", [sd["step"]] * len(self.params)))
        if len(steps) != len(self.params):
            raise ValueError(f"optimizer state holds {len(steps)} per-parameter step counts, the optimizer has "
                             f"{len(self.params)} parameters")
        if sd["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError("optimizer state was saved for a different parameter layout")
        self._steps = steps
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


class OneCycleLR:
    """torch.optim.lr_scheduler.OneCycleLR as src/utils/scheduler.py:36-41 builds it (two phases, ``anneal_strategy``
    "linear" or "cos", pct_start 0.3, div_factor 25, final_div_factor 1e4, momentum cycled through the optimizer's beta1
    between 0.85 and 0.95) for an optimizer that exposes ``param_groups`` - torch's class insists on a
    ``torch.optim.Optimizer`` instance, which the flat-buffer optimizer is not.  Same values step for step
    (tests/test_train_harness.py compares with torch's class)."""

    def __init__(self, optimizer, max_lr, total_steps=None, epochs=None, steps_per_epoch=None, pct_start=0.3,
                 anneal_strategy="cos", cycle_momentum=True, base_momentum=0.85, max_momentum=0.95, div_factor=25.0,
                 final_div_factor=1e4):
        if total_steps is None:
            if epochs is None or steps_per_epoch is None or epochs <= 0 or steps_per_epoch <= 0:
                raise ValueError("You must define either total_steps OR (epochs AND steps_per_epoch)")
            total_steps = epochs * steps_per_epoch
        if total_steps <= 0:
            raise ValueError(f"Expected positive integer total_steps, but got {total_steps}")
        if anneal_strategy not in ("cos", "linear"):
            raise ValueError(f"anneal_strategy must be one of 'cos' or 'linear', instead got {anneal_strategy}")
        self.optimizer, self.total_steps, self.anneal_strategy = optimizer, total_steps, anneal_strategy
        self.cycle_momentum = cycle_momentum
        self._phases = [dict(end_step=float(pct_start * total_steps) - 1, start_lr="initial_lr", end_lr="max_lr",
                             start_momentum="max_momentum", end_momentum="base_momentum"),
                        dict(end_step=total_steps - 1, start_lr="max_lr", end_lr="min_lr", start_momentum="base_momentum",
                             end_momentum="max_momentum")]
        for g in optimizer.param_groups:
            g["initial_lr"] = max_lr / div_factor
            g["max_lr"] = max_lr
            g["min_lr"] = g["initial_lr"] / final_div_factor
            if cycle_momentum:
                g["max_momentum"], g["base_momentum"] = max_momentum, base_momentum
                g["betas"] = (max_momentum, g["betas"][1])
        self.last_epoch = -1
        self._last_lr = []
        self.step()                                    # LRScheduler._initial_step

    def _anneal(self, start, end, pct):
        if self.anneal_strategy == "linear":
            return (end - start) * pct + start
        import math
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def step(self):
        self.last_epoch += 1
        n = self.last_epoch
        if n > self.total_steps:
            raise ValueError(f"Tried to step {n} times. The specified number of total steps is {self.total_steps}")
        self._last_lr = []
        for g in self.optimizer.param_groups:
            start = 0.0
            for i, ph in enumerate(self._phases):
                end = ph["end_step"]
                if n <= end or i == len(self._phases) - 1:
                    pct = (n - start) / (end - start)
                    lr = self._anneal(g[ph["start_lr"]], g[ph["end_lr"]], pct)
                    mom = self._anneal(g[ph["start_momentum"]], g[ph["end_momentum"]], pct) if self.cycle_momentum else None
                    break
                start = end
            g["lr"] = lr
            if self.cycle_momentum:
                g["betas"] = (mom, g["betas"][1])
            self._last_lr.append(lr)

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def set_optimizer(config, e2e, train_loader):
    """src/utils/scheduler.py:6-45: ``scheduler: noam`` -> Adam(0.9, 0.98, 1e-9) under the Noam rate, no scheduler object;
    ``scheduler: onecycle`` -> ``optimizer: adamw`` (torch defaults: betas (0.9, 0.999), eps 1e-8, weight decay 0.01) or
    ``adam`` (betas (0.9, 0.98), eps 10e-09) at ``learning_rate`` under a linear one-cycle schedule over
    ``epochs x ceil(len(loader) / accum_grad)`` steps."""
    import math
    ts = config.training_settings
    steps_per_epoch = math.ceil(len(train_loader) / ts["accum_grad"]) if ts["accum_grad"] != 0 else len(train_loader)
    optimizer = scheduler = None
    if ts["scheduler"] != "noam":
        params = [p for p in e2e.parameters() if p.requires_grad]
        if ts["optimizer"] == "adamw":
            optimizer = FusedAdam(params, ts["learning_rate"], betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
        elif ts["optimizer"] == "adam":
            optimizer = FusedAdam(params, ts["learning_rate"], betas=(0.9, 0.98), eps=10e-09)
    if ts["scheduler"] == "noam":
        optimizer = get_noam_scheduler(e2e.parameters(), ts["noam_factor"], config.encoder_conf["output_size"], ts["warmup_steps"])
    elif ts["scheduler"] == "onecycle":
        scheduler = OneCycleLR(optimizer, max_lr=ts["learning_rate"], steps_per_epoch=steps_per_epoch, epochs=ts["epochs"],
                               anneal_strategy="linear")
    else:
        raise RuntimeError("The scheduler should be specified as 'noam' or 'onecycle'")
    return optimizer, scheduler


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
            if stepping:
                buckets.begin_step()        # nothing of an earlier, unfinished window survives into this one
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
