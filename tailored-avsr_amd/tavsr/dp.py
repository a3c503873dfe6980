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


def init_from_env(backend: str | None = None, seed: int | None = 0) -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment; initialises the group.

    ``seed`` (None: leave the generators alone): every rank seeds its generators with ``seed + rank`` - the device
    generator of the dropout masks (``ops.manual_seed``) and torch's host generator (SpecAug draws, stochastic depth) -
    so that data-parallel ranks do not draw identical masks.  Parameters are made equal separately
    (``GradBuckets.broadcast_parameters``)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # TAVSR_DP_BACKEND=gloo: several ranks on ONE GPU (RCCL needs a device per rank) - test rigs only
            backend = os.environ.get("TAVSR_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if seed is not None:
        torch.manual_seed(int(seed) + rank)
        if torch.cuda.is_available():
            from . import ops
            ops.manual_seed(0x5EED5EED + int(seed) + rank, torch.device("cuda", local))
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
        self._tab, self._flatbuf = {}, {}

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
        """Sum gradients over ranks and divide by the world size (in place on ``p.grad``).

        On the GPU a bucket is packed and unpacked by ONE launch each (``tavsr_bucket_copy`` over a pointer table; the
        1/world average rides on the unpack) and the flat buffers persist across steps; all buckets are packed and their
        all-reduces issued before the first wait.  CPU tensors (gloo tests) take the torch path."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self.params and self.params[0].is_cuda:
            return self._allreduce_mean_hip(world)
        works = []
        for i, bucket in enumerate(self.buckets):
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            self._flat[i] = flat
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, bucket in enumerate(self.buckets):
            works[i].wait()
            flat = self._flat[i]
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                p.grad.copy_(flat[off: off + p.numel()].view_as(p))
                off += p.numel()
            self._flat[i] = None

    def _tables(self, i, bucket):
        """device pointer / offset / size tables of bucket i; rebuilt only when a gradient tensor moved (under hipGraph
        replay the gradients are graph-owned buffers: the tables are built once)."""
        ptrs = [p.grad.data_ptr() for p in bucket]
        cached = self._tab.get(i)
        if cached is not None and cached[0] == ptrs:
            return cached[1:]
        dev = bucket[0].device
        sizes = [p.numel() for p in bucket]
        offs = [0]
        for n in sizes[:-1]:
            offs.append(offs[-1] + (n + 3) // 4 * 4)                # 16-byte aligned slots
        total = offs[-1] + sizes[-1]
        assert all(p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in bucket)
        t = tuple(torch.tensor(v, dtype=torch.int64).to(dev) for v in (ptrs, offs, sizes))
        flat = self._flatbuf.get(i)
        if flat is None or flat.numel() != total:
            flat = self._flatbuf[i] = torch.zeros(total, dtype=torch.float32, device=dev)
        self._tab[i] = (ptrs,) + t + (flat, max(sizes))
        return t + (flat, max(sizes))

    def _allreduce_mean_hip(self, world):
        from . import ops
        works = []
        for i, bucket in enumerate(self.buckets):
            ptrs, offs, sizes, flat, mx = self._tables(i, bucket)
            ops.bucket_copy(ptrs, offs, sizes, len(bucket), flat, 1.0, True, mx)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, bucket in enumerate(self.buckets):
            works[i].wait()
            ptrs, offs, sizes, flat, mx = self._tables(i, bucket)
            ops.bucket_copy(ptrs, offs, sizes, len(bucket), flat, 1.0 / world, False, mx)
