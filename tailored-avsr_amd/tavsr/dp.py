"""Utterance-level data parallelism: one process per GPU, gradients summed once per step with RCCL
(``torch.distributed`` backend "nccl" on ROCm) over xGMI.  The reference is single-process
(SURVEY 2.1); this is the only collective on the path - no data-path exchange.

Gradients are packed into a few large flat fp32 buckets (xGMI is point-to-point: few large messages beat
many small ones), all-reduced asynchronously in reverse-parameter order and unpacked.  The 1/world
average is folded into the unpack.  Works unchanged on CPU tensors with the gloo backend (tests)."""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment; initialises the group."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradBuckets:
    """Static bucket plan over a model's parameters (built once; parameters never change identity)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(self.params):  # backward produces the last layers' gradients first
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def broadcast_parameters(self, src: int = 0) -> None:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        for bucket in self.buckets:
            flat = torch.cat([p.data.reshape(-1) for p in bucket])
            dist.broadcast(flat, src)
            off = 0
            for p in bucket:
                p.data.copy_(flat[off: off + p.numel()].view_as(p))
                off += p.numel()

    def allreduce_mean(self) -> None:
        """Sum gradients over ranks and divide by the world size (in place on ``p.grad``)."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        works = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._flat[i] = flat
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, bucket in enumerate(self.buckets):
            works[i].wait()
            flat = self._flat[i]
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                g = flat[off: off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
            self._flat[i] = None
