"""Utterance-level data parallelism: one process per GPU, gradients summed once per step with RCCL over xGMI.
The reference is single-process; this is the only collective on the path - no data-path exchange.

Gradients are packed into a few large flat fp32 buckets (xGMI is point-to-point: few large messages beat many small
ones) in reverse-parameter order, all-reduced and unpacked; the 1/world average is folded into the unpack.  On GPUs the
all-reduce is the C ABI's own RCCL call (``tavsr_dp_allreduce``, csrc/dp.cpp) on a communication stream;
``torch.distributed`` provides the rendezvous (the 128-byte communicator id travels through it) and the barriers.  With
``overlap`` (eager training loops) a bucket's all-reduce is enqueued from a gradient hook the moment its last gradient
has been produced, so the exchange runs under the rest of the backward pass.  CPU tensors (gloo tests) and several ranks
on ONE GPU (``TAVSR_DP_BACKEND=gloo`` test rigs; RCCL needs a device per rank) go through ``torch.distributed``."""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Iterable, List

import torch
import torch.distributed as dist

from ._lib import addr as L_addr, ptr as L_ptr

RCCL_ABI = False          # set by init_from_env once tavsr_dp_init has succeeded on this rank
FORCE_WORLD1 = False      # rehearsal: a ONE-rank RCCL communicator driven through the whole exchange path (init_from_env(force_rccl=True))


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def _exchange_on() -> bool:
    """is there a gradient exchange to run?  (N > 1 ranks, or the one-rank RCCL rehearsal)"""
    return FORCE_WORLD1 or (dist.is_initialized() and dist.get_world_size() > 1)


def _rccl_init(rank: int, world: int, device: torch.device) -> bool:
    """communicator of the C ABI: rank 0 draws the id, torch.distributed ships it, every rank joins."""
    from ._lib import check, lib
    ident = torch.zeros(128, dtype=torch.uint8, device=device)
    err = None
    if rank == 0:
        # rank 0 ALWAYS takes part in the broadcast below: an all-zero id is the failure marker (RCCL ids are never all zero),
        # so a rank 0 that cannot draw an id does not leave the others blocked in the collective
        try:
            buf = (C.c_char * 128)()
            check(lib().tavsr_dp_unique_id(buf), "tavsr_dp_unique_id")
            ident.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
        except Exception as e:          # librccl absent, symbol missing, stale libtavsr_hip.so
            err = e
            ident.zero_()
    dist.broadcast(ident, 0)
    if err is not None:
        raise err
    raw = bytes(ident.cpu().tolist())
    if not any(raw):
        raise RuntimeError("rank 0 could not create the RCCL communicator id")
    check(lib().tavsr_dp_init(rank, world, C.c_char_p(raw)), "tavsr_dp_init")
    return True


import contextlib


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL prints a version banner to the C library's stdout when a communicator is created; programs whose stdout is a
    protocol (bench.py: ONE JSON line) must not carry it.  Inside this scope file descriptor 1 is descriptor 2, and the C
    library's buffer is flushed before the descriptor comes back."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _rccl_init_world1(device: torch.device) -> None:
    """a communicator with ONE rank through the C ABI (no torch.distributed involved): every call of the exchange path then
    runs for real - id, init, all-reduce kernels on the communication stream - without a second GPU."""
    global RCCL_ABI, FORCE_WORLD1
    from ._lib import check, lib
    torch.cuda.set_device(device)
    buf = (C.c_char * 128)()
    check(lib().tavsr_dp_unique_id(buf), "tavsr_dp_unique_id")
    check(lib().tavsr_dp_init(0, 1, C.c_char_p(buf.raw)), "tavsr_dp_init")
    RCCL_ABI = FORCE_WORLD1 = True


def init_from_env(backend: str | None = None, seed: int | None = 0, force_rccl: bool = False) -> tuple[int, int, int]:
    """(see _init_from_env; communicator creation runs with the C library's stdout pointed at stderr)"""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or force_rccl:
        with _c_stdout_to_stderr():
            out = _init_from_env(backend, seed, force_rccl)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            return out
    return _init_from_env(backend, seed, force_rccl)


def _init_from_env(backend: str | None = None, seed: int | None = 0, force_rccl: bool = False) -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment; initialises the group.

    ``seed`` (None: leave the generators alone): every rank seeds its generators with ``seed + rank`` - the device
    generator of the dropout masks (``ops.manual_seed``) and torch's host generator (SpecAug draws, stochastic depth) -
    so that data-parallel ranks do not draw identical masks.  Parameters are made equal separately
    (``GradBuckets.broadcast_parameters``)."""
    global RCCL_ABI
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
        if backend == "nccl" and os.environ.get("TAVSR_DP_RCCL", "1") == "1":
            try:
                RCCL_ABI = _rccl_init(rank, world, torch.device("cuda", local))
            except Exception as e:          # both are GPU paths; say which one runs
                warnings.warn(f"tavsr_dp_init failed ({e}); gradients go through torch.distributed's RCCL instead")
                RCCL_ABI = False
            mine = RCCL_ABI
            flag = torch.tensor([int(RCCL_ABI)], device=f"cuda:{local}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # all ranks or none
            RCCL_ABI = bool(int(flag))
            if mine and not RCCL_ABI:                         # joined here but not everywhere: give the communicator back
                from ._lib import lib
                lib().tavsr_dp_destroy()
    if world == 1 and force_rccl and torch.cuda.is_available() and not RCCL_ABI:
        _rccl_init_world1(torch.device("cuda", local))
    if seed is not None:
        torch.manual_seed(int(seed) + rank)
        if torch.cuda.is_available():
            from . import ops
            ops.manual_seed(0x5EED5EED + int(seed) + rank, torch.device("cuda", local))
    return rank, local, world


def world_report(peak_bytes: int = 0) -> dict:
    """First-contact record of an N > 1 run for the bench line: how many ranks the C ABI's RCCL communicator really holds
    (``tavsr_dp_world()``: 0 = no communicator, gradients went through torch.distributed or nowhere), how many ranks
    ``torch.distributed`` sees, and every rank's peak device memory.  A collective on the host side (``all_gather_object``):
    call it on every rank, outside the timed region."""
    from ._lib import lib
    world = dist.get_world_size() if dist.is_initialized() else 1
    peaks = [None] * world
    if world > 1:
        dist.all_gather_object(peaks, float(peak_bytes))
    else:
        peaks = [float(peak_bytes)]
    return {"rccl_world": int(lib().tavsr_dp_world()), "dist_world": world,
            "dist_backend": dist.get_backend() if dist.is_initialized() else None,
            "hbm_peak_gb_per_rank": [round(p / 2**30, 2) for p in peaks]}


def shutdown():
    """frees the C ABI's communicator and the process group."""
    global RCCL_ABI, FORCE_WORLD1
    if RCCL_ABI:
        from ._lib import check, lib
        check(lib().tavsr_dp_destroy(), "tavsr_dp_destroy")
        RCCL_ABI = FORCE_WORLD1 = False
    if dist.is_initialized():
        dist.destroy_process_group()


GLOO_HOST_STAGED = os.environ.get("TAVSR_DP_GLOO_HOST_STAGED", "1") == "1"


class GradBuckets:
    """Static bucket plan over a model's parameters (built once; parameters never change identity)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self._hooked, self.overlap = False, True
        self._plan(list(reversed(self.params)), [])  # backward produces the last layers' gradients first

    def replan(self, late_params) -> int:
        """Rebuild the plan for a step whose backward pass runs in two phases (TwoPhaseBackward): the parameters of the second
        phase go to the end of the issue order, in buckets of their own, so that every bucket before them is complete when the
        first phase has run.  Every rank must call it with the same partition (same model: same partition), outside a window.
        Returns the number of buckets that hold no late parameter."""
        assert not self._armed and self._next == 0, "replan() inside an exchange window"
        late = {id(p) for p in late_params}
        order = list(reversed(self.params))
        n_early = self._plan([p for p in order if id(p) not in late], [p for p in order if id(p) in late])
        if self._hooked:
            self._index.clear()
            self._index.update({id(p): i for i, b in enumerate(self.buckets) for p in b})
        return n_early

    def _plan(self, first, second) -> int:
        self.buckets: List[List[torch.nn.Parameter]] = []
        n_first = 0
        for group in (first, second):
            cur, size = [], 0
            for p in group:
                cur.append(p)
                size += p.numel() * 4
                if size >= self.bucket_bytes:
                    self.buckets.append(cur)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
            if group is first:
                n_first = len(self.buckets)
        self._reset_plan_state()
        return n_first

    def _reset_plan_state(self):
        self._flat = [None] * len(self.buckets)
        self._tab, self._flatbuf = {}, {}
        self._hostbuf = {}                      # (pinned staging of the gloo rehearsal rig: sized per bucket of the OLD plan)
        self._hook_streams = {}
        self._comm = getattr(self, "_comm", None)      # the communication stream outlives a re-plan
        self._works = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0            # buckets 0 .. _next-1 have been enqueued in this window (ALWAYS in index order)
        self._armed = False       # a window is open: begin_step() has been called and allreduce_mean() has not yet
        # flat sizes do not depend on the gradients: [HIP layout (16-byte aligned slots), torch.cat layout]
        self._numel_hip = [sum((p.numel() + 3) // 4 * 4 for p in b[:-1]) + b[-1].numel() for b in self.buckets]
        self._numel_cat = [sum(p.numel() for p in b) for b in self.buckets]

    def begin_step(self) -> None:
        """Open a backward window: from here to ``allreduce_mean()`` the gradient hooks may enqueue buckets (overlap with
        the backward pass).  Every rank must call it at the same point of the program.  A window issues EXACTLY one
        collective per bucket, in index order, on every rank - that is the invariant that keeps ranks paired.  If the
        previous window was never closed (an exception, a skipped step, a stray backward), it is completed here: the
        buckets it did not issue are issued now with whatever the flat buffers hold, all results are dropped.  Without
        ``begin_step`` the hooks stay idle and ``allreduce_mean`` issues everything itself (no overlap, same result)."""
        if self._armed:
            cuda = bool(self.params) and self.params[0].is_cuda
            for i in range(self._next, len(self.buckets)):
                if cuda:
                    self._issue_flat(i, self._flatbuf_for(i))
                else:
                    self._flat[i] = torch.zeros(self._numel_cat[i])
                    self._works[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, async_op=True)
            for i, w in enumerate(self._works):
                if w is not None and w != "rccl":
                    w.wait()
                self._works[i] = None
                self._flat[i] = None
            if self._comm is not None and RCCL_ABI:
                torch.cuda.current_stream().wait_stream(self._comm)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._hook_streams = {}
        self._armed = _exchange_on()

    def broadcast_parameters(self, src: int = 0) -> None:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        for bucket in self.buckets:
            flat = torch.cat([p.data.reshape(-1) for p in bucket])
            if RCCL_ABI and flat.is_cuda:
                from ._lib import check, lib
                check(lib().tavsr_dp_broadcast(L_ptr(flat), C.c_int64(flat.numel()), src,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tavsr_dp_broadcast")
            else:
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
        if not _exchange_on():
            return
        world = _world()
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self.params and self.params[0].is_cuda:
            return self._allreduce_mean_hip(world)
        for i in range(self._next, len(self.buckets)):      # what the gradient hooks did not enqueue, in index order
            self._launch_bucket_cpu(i)
        self._next = len(self.buckets)
        for i, bucket in enumerate(self.buckets):
            self._works[i].wait()
            flat = self._flat[i]
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                p.grad.copy_(flat[off: off + p.numel()].view_as(p))
                off += p.numel()
            self._flat[i] = None
            self._works[i] = None
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._armed = False

    def _launch_bucket_cpu(self, i):
        """torch path (gloo, CPU tensors): the same fixed issue order as the GPU path."""
        flat = torch.cat([p.grad.reshape(-1) for p in self.buckets[i]])
        self._flat[i] = flat
        self._works[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _tables(self, i, bucket):
        """device pointer / offset / size tables of bucket i; rebuilt only when a gradient tensor moved (under hipGraph
        replay the gradients are graph-owned buffers: the tables are built once)."""
        ptrs = [L_addr(p.grad) for p in bucket]
        cached = self._tab.get(i)
        if cached is not None and cached[0] == ptrs:
            return cached[1:]
        dev = bucket[0].device
        sizes = [p.numel() for p in bucket]
        offs = [0]
        for n in sizes[:-1]:
            offs.append(offs[-1] + (n + 3) // 4 * 4)                # 16-byte aligned slots
        assert offs[-1] + sizes[-1] == self._numel_hip[i]
        assert all(p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in bucket)
        t = tuple(torch.tensor(v, dtype=torch.int64).to(dev) for v in (ptrs, offs, sizes))
        flat = self._flatbuf_for(i)
        self._tab[i] = (ptrs,) + t + (flat, max(sizes))
        return t + (flat, max(sizes))

    def _flatbuf_for(self, i):
        flat = self._flatbuf.get(i)
        if flat is None:
            flat = self._flatbuf[i] = torch.zeros(self._numel_hip[i], dtype=torch.float32, device=self.buckets[i][0].device)
        return flat

    # ---- GPU path: pack (one launch per bucket) -> all-reduce on the communication stream -> unpack * 1/world
    def _comm_stream(self):
        from . import _lib
        if _lib.SINGLE_STREAM:                  # one queue: the exchange follows the backward pass on the calling stream
            return torch.cuda.current_stream()
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        return self._comm

    def _launch_bucket(self, i):
        """pack bucket i on the current stream and enqueue its all-reduce behind the pack on the communication stream."""
        from . import ops
        from ._lib import check, lib
        bucket = self.buckets[i]
        ptrs, offs, sizes, flat, mx = self._tables(i, bucket)
        # gradients are accumulated on the stream of their node (autograd replays a node on its forward stream: the CTC branch's
        # ctc_lo lives on a forked one), and the hook that completes a bucket may itself run there: the pack is enqueued on the
        # current stream behind EVERY stream a hook of this window has fired on
        cur = torch.cuda.current_stream()
        for h, st in self._hook_streams.items():
            if h != cur.cuda_stream:
                cur.wait_stream(st)
        ops.wgrad_fence()          # ... and behind the weight-gradient launches a layer left open on its side queue
        ops.bucket_copy(ptrs, offs, sizes, len(bucket), flat, 1.0, True, mx)
        self._issue_flat(i, flat)

    def _issue_flat(self, i, flat):
        from ._lib import check, lib
        if RCCL_ABI:
            comm = self._comm_stream()
            comm.wait_stream(torch.cuda.current_stream())
            check(lib().tavsr_dp_allreduce(L_ptr(flat), C.c_int64(flat.numel()), C.c_void_p(comm.cuda_stream)),
                  "tavsr_dp_allreduce")
            self._works[i] = "rccl"
        elif dist.get_backend() == "gloo" and GLOO_HOST_STAGED:
            self._works[i] = self._issue_host_staged(i, flat)
        else:
            self._works[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _issue_host_staged(self, i, flat):
        """gloo with GPU gradients (the one-GPU rehearsal rig): the flat buffer goes to a PERSISTENT pinned host buffer on the
        communication stream, one helper thread all-reduces it over gloo's host path and sends it back on the same stream.
        (gloo's own device path allocates its pinned staging per call; next to a replaying hipGraph those calls took seconds.)
        The helper issues the collectives in submission order - the same on every rank."""
        import concurrent.futures
        if getattr(self, "_pool", None) is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        host = self._hostbuf.get(i)
        if host is None:
            host = self._hostbuf[i] = torch.empty(flat.numel(), dtype=flat.dtype, pin_memory=True)
        comm = self._comm_stream()
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            host.copy_(flat, non_blocking=True)
            done = torch.cuda.Event()
            done.record(comm)
        dev = flat.device

        def job():
            torch.cuda.set_device(dev)
            done.synchronize()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            with torch.cuda.stream(comm):
                flat.copy_(host, non_blocking=True)

        fut = self._pool.submit(job)

        class _Work:
            def wait(self_inner):
                fut.result()
                torch.cuda.current_stream().wait_stream(comm)

        return _Work()

    def _allreduce_mean_hip(self, world):
        from . import ops
        for i in range(self._next, len(self.buckets)):      # what the gradient hooks did not enqueue, in index order
            self._launch_bucket(i)
        self._next = len(self.buckets)
        if RCCL_ABI:
            torch.cuda.current_stream().wait_stream(self._comm_stream())
        for i, bucket in enumerate(self.buckets):
            if self._works[i] != "rccl":
                self._works[i].wait()
            ptrs, offs, sizes, flat, mx = self._tables(i, bucket)
            ops.bucket_copy(ptrs, offs, sizes, len(bucket), flat, 1.0 / world, False, mx)
            self._works[i] = None
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._armed = False

    # ---- overlap for captured steps: the backward pass as two hipGraphs around a cut (TwoPhaseBackward below)
    def ready_prefix(self, late_params) -> int:
        """number of leading buckets (issue order) that hold none of ``late_params``: complete once the first phase has run"""
        late = {id(p) for p in late_params}
        n = 0
        while n < len(self.buckets) and not any(id(p) in late for p in self.buckets[n]):
            n += 1
        return n

    def launch_prefix(self, n: int) -> None:
        """pack buckets ``_next .. n-1`` and enqueue their all-reduces now (GPU tensors; their gradients must exist): what follows
        on the compute stream runs beside them, ``allreduce_mean`` issues the rest and completes all.  Every rank must call it
        with the same ``n`` - the issue order stays 0, 1, 2, ..."""
        if not _exchange_on():
            return
        for b in self.buckets[self._next: n]:                  # (as allreduce_mean: a rank that skipped a layer sends zeros)
            for p in b:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        cuda = bool(self.params) and self.params[0].is_cuda
        for i in range(self._next, min(n, len(self.buckets))):
            (self._launch_bucket if cuda else self._launch_bucket_cpu)(i)
        self._next = max(self._next, min(n, len(self.buckets)))

    # ---- overlap with the backward pass (eager loops): a bucket leaves as soon as its last gradient exists
    def attach_overlap_hooks(self) -> None:
        """The hooks work inside a window opened by ``begin_step()`` (``tavsr.train.training`` opens one before the last
        micro-batch's backward pass).  ``p.grad`` must be None at the start of that backward pass (``optimizer.zero_grad()``
        does that) or hold the earlier micro-batches' sums: each parameter's hook then fires once in it.  Not for captured
        (hipGraph) steps - no window is opened around a capture; ``allreduce_mean`` enqueues whatever the hooks did not.

        Issue order is FIXED: collectives are paired across ranks by the order in which they are enqueued on the communicator,
        and ranks do not complete their buckets in the same order (every rank draws its own stochastic-depth / branch-drop
        coins, so a rank may never produce some layer's gradients at all).  A hook therefore only marks its bucket complete;
        bucket i is enqueued when buckets 0 .. i-1 have been (as torch DDP does), and ``allreduce_mean`` enqueues the rest
        in index order - every rank issues 0, 1, 2, ... whatever its gradients looked like."""
        if self._hooked or not _exchange_on():
            return
        self._hooked = True
        index = self._index = {id(p): i for i, b in enumerate(self.buckets) for p in b}

        def hook(p):
            if not (self.overlap and self._armed):
                return
            i = index[id(p)]
            if p.is_cuda:
                st = torch.cuda.current_stream()
                self._hook_streams.setdefault(st.cuda_stream, st)
            self._pending[i] -= 1               # (a second backward pass inside one window drives this below 0: no launch)
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                (self._launch_bucket if p.is_cuda else self._launch_bucket_cpu)(self._next)
                self._next += 1

        for b in self.buckets:
            for p in b:
                if getattr(p, "_post_accumulate_grad_hooks", None):
                    foreign = True       # somebody else's hook reads the gradient where it is accumulated: no deferred weight gradients
                else:
                    foreign = False
                p.register_post_accumulate_grad_hook(hook)
                p._tavsr_hooks_fence = not foreign      # (ops.wgrad_may_go_beside: this package's hooks fence in _launch_bucket)



def _params_below(tensors) -> set:
    """ids of the leaf tensors whose AccumulateGrad nodes the autograd graph under ``tensors`` reaches"""
    seen, out = set(), set()
    stack = [t.grad_fn for t in tensors if t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        if hasattr(fn, "variable"):
            out.add(id(fn.variable))
        stack.extend(n for n, _ in fn.next_functions if n is not None)
    return out


_ACTIVE_CUT = [None]


def cut(*tensors):
    """A model marks where its backward pass may be split: ``a, v = dp.cut(a, v)``.  Outside ``TwoPhaseBackward.forward()`` this
    returns its arguments unchanged.  Inside, each tensor that requires a gradient is replaced by a detached leaf for the rest of
    the forward pass and the pair is recorded: ``phase_a`` then ends at the leaves, ``phase_b`` continues from the originals with
    the leaves' gradients.  Only the first ``cut`` call of a forward pass is honoured (one cut = two phases)."""
    plan = _ACTIVE_CUT[0]
    if plan is None or plan.pairs or not torch.is_grad_enabled():
        return tensors if len(tensors) != 1 else tensors[0]
    out = []
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad and t.grad_fn is not None:
            leaf = t.detach().requires_grad_(True)
            plan.pairs.append((t, leaf))
            out.append(leaf)
        else:
            out.append(t)
    return tuple(out) if len(out) != 1 else out[0]


class TwoPhaseBackward:
    """The backward pass of one step as two calls around a cut of the autograd graph, so that a captured step is TWO hipGraphs
    and the gradient buckets that the first one completes are exchanged while the second one replays:

        with two.forward(): loss = model(batch)     the model's ``dp.cut(...)`` call detaches the graph there
        two.phase_a(loss)                           loss.backward(): everything above the cut (ends at the detached leaves)
        two.phase_b()                               continues below the cut from the leaves' gradients

    Same gradients as one ``loss.backward()`` PROVIDED the cut tensors are the only connection between the two halves: every node
    then runs once, in one of the two calls (``phase_a`` refuses a loss that reaches the lower half some other way, e.g. through an
    intermediate-CTC tap below the cut); a parameter used on both sides of the cut accumulates both contributions in ``.grad``
    and counts as late (its hook fires once per phase: ``late_params`` puts it in a late bucket, which no hook completes early).  ``late_params(params)`` after a forward pass:
    the parameters the graph below the cut reaches - the buckets that hold none of them are complete after ``phase_a``."""

    def __init__(self):
        self.pairs = []

    def forward(self):
        two = self

        class _Scope:
            def __enter__(self):
                two.pairs = []
                _ACTIVE_CUT[0] = two
                return two

            def __exit__(self, *exc):
                _ACTIVE_CUT[0] = None
                return False

        return _Scope()

    @property
    def split(self) -> bool:
        return bool(self.pairs)

    def late_params(self, params):
        below = _params_below([t for t, _ in self.pairs])
        return [p for p in params if id(p) in below]

    def bypassed(self, loss) -> bool:
        """does the graph under ``loss`` reach a node below the cut WITHOUT passing a detached leaf (an intermediate-CTC tap of
        a layer below the cut, a parameter-sharing path)?  Then ``phase_a`` would already walk - and free - part of the lower
        graph and ``phase_b`` would run those nodes a second time: such a step must not be split."""
        # every node of the lower graph, not only the cut tensors' own: a tap may sit any number of layers below the cut (the
        # cut follows the middle encoder layer, an intermediate-CTC tap may follow layer 3).  AccumulateGrad nodes are left out:
        # a parameter used on both sides of the cut is legal (it counts as late).
        below, stack = set(), [t.grad_fn for t, _ in self.pairs if t.grad_fn is not None]
        while stack:
            fn = stack.pop()
            if fn in below or hasattr(fn, "variable"):
                continue
            below.add(fn)
            stack.extend(n for n, _ in fn.next_functions if n is not None)
        seen, stack = set(), [loss.grad_fn] if loss.grad_fn is not None else []
        while stack:
            fn = stack.pop()
            if fn in seen:
                continue
            if fn in below:
                return True
            seen.add(fn)
            stack.extend(n for n, _ in fn.next_functions if n is not None)
        return False

    def phase_a(self, loss) -> None:
        if self.pairs and self.bypassed(loss):
            raise RuntimeError("TwoPhaseBackward: the loss reaches the graph below dp.cut() without passing the cut (an intermediate "
                               "CTC layer below it?): run this step as one backward pass (bench.py --no-split-backward)")
        loss.backward()

    def phase_b(self) -> None:
        pairs = [(t, leaf) for t, leaf in self.pairs if leaf.grad is not None]
        if pairs:
            torch.autograd.backward([t for t, _ in pairs], [leaf.grad for _, leaf in pairs])
        self.pairs = []
