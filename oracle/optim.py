"""ORACLE (test infrastructure, never shipped): CPU restatement of the reference's optimizer wrapper and training loop
(src/schedulers/noam.py:11-81, avsr_main.py:27-58) on torch.optim.Adam.  Pinned by ``tests/golden/noam_adam.npz``,
generated from the reference's own ``src/schedulers/noam.py`` (pure torch: imported as it is)."""
from __future__ import annotations

import torch


class NoamOracle:
    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer, self._step, self.warmup, self.factor, self.model_size, self._rate = optimizer, 0, warmup, factor, model_size, 0

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def rate(self, step=None):                                         # noam.py:38-46
        step = self._step if step is None else step
        return self.factor * self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5))

    def step(self):                                                    # noam.py:29-36
        self._step += 1
        rate = self.rate()
        for p in self.optimizer.param_groups:
            p["lr"] = rate
        self._rate = rate
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()


def get_noam_oracle(params, factor, d_model, warmup):                  # noam.py:72-81
    return NoamOracle(d_model, factor, warmup, torch.optim.Adam(params, lr=0, betas=(0.9, 0.98), eps=1e-9))


def training_oracle(e2e, loader, optimizer, scheduler, accum_grad):    # avsr_main.py:27-58
    e2e.train()
    train_loss = 0.0
    optimizer.zero_grad()
    for batch_idx, batch in enumerate(loader):
        loss = e2e(**batch)[0] / accum_grad
        loss.backward()
        if ((batch_idx + 1) % accum_grad == 0) or (batch_idx + 1 == len(loader)):
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        train_loss += loss.item()
    return train_loss / (len(loader) / accum_grad)
