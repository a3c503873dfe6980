"""Tensor-level wrappers over the C ABI (include/tavsr.h).  No autograd, no torch arithmetic:
torch is used only to allocate buffers on the caching allocator and to name the current stream.
Every function raises ``TavsrError`` on a CPU tensor or a failed call - there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import ACT, AttnDesc, GemmDesc, check, lib, ptr, require_cuda, stream

f32 = torch.float32


def empty(*shape, like: Optional[torch.Tensor] = None, dtype=f32, device=None):
    dev = like.device if like is not None else (device if device is not None else "cuda")
    return torch.empty(shape, dtype=dtype, device=dev)


_addr = L.addr      # tensor (+ element offset) -> integer address; the stream-safety hook lives there (_lib.py)


class GemmProfile:
    """Optional per-launch timing of tavsr_gemm with HIP events on the launch stream (bench.py's roofline
    leg).  Off on the product path (``PROFILE is None``): zero overhead."""

    def __init__(self, by_shape: bool = False):
        self.records = []  # (key, flops, operand bytes, start_event, end_event)
        self.by_shape = by_shape

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += nbytes
        return {k: dict(calls=v[0], flops=v[1], seconds=v[2], bytes=v[3]) for k, v in agg.items()}


PROFILE: Optional[GemmProfile] = None


def _gemm_kernel_key(d) -> str:
    """Names the kernel family a launch runs (layout); the tile is the planner's choice (csrc/gemm.hip)."""
    return f"gemm_kernel<{'T' if d.a_kmajor else 'N'}{'N' if d.b_kmajor else 'T'}>"


_ZERO = {}


def _zero_page(device) -> torch.Tensor:
    z = _ZERO.get(device)
    if z is None:
        z = _ZERO[device] = torch.zeros(64, dtype=f32, device=device)
    return z


# ---------------------------------------------------------------------------------------------- GEMM
def gemm(M, N, K, A, lda, B, ldb, Cc, ldc, *, a_off=0, b_off=0, c_off=0, a_kmajor=False, b_kmajor=False,
         nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, act=None, alpha=1.0, Z=None,
         R=None, r_off=0, ldr=0, sR=(0, 0), DZ=None, dact=None, a_rowsum=None, conv=None, force=None, ws_cap=None,
         drop=None, rowstat=None, ln=None, rowdot=None):
    """Raw descriptor call; offsets are in elements into the given tensors.  ``force=(cfg, nsplit)`` bypasses
    the planner (tuning / tests); ``ws_cap`` caps the split-K workspace handed to the library (tests of its fallback).
    ``ln`` = (gamma, beta, eps, out [M, N]): also the LayerNorm of the result rows (tavsr_gemm_ln)."""
    require_cuda(A, B, Cc, bias, Z, R, DZ, a_rowsum)
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.A, d.lda = _addr(A, a_off), lda
    d.B, d.ldb = _addr(B, b_off), ldb
    d.C, d.ldc = _addr(Cc, c_off), ldc
    d.nb1, d.nb2 = nb1, nb2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.bias = _addr(bias)
    d.act, d.alpha = ACT[act], alpha
    d.Z = None if Z is None else _addr(Z, c_off)
    if R is not None:
        d.R, d.ldr = _addr(R, r_off), ldr
        d.sR1, d.sR2 = sR
    if DZ is not None:
        d.DZ, d.dact = _addr(DZ, c_off), ACT[dact]
    if a_rowsum is not None:
        d.a_rowsum = _addr(a_rowsum)
    if drop is not None:        # dropout token (p, offset, seed tensor): the mask rides in the epilogue
        d.drop_p, d.drop_offset, d.drop_seed = drop[0], drop[1], _addr(drop[2])
    if rowstat is not None:     # [M, ceil(N / 64), 2]: per-row (sum, sum of squares) of every 64-column tile of the result
        d.rowstat = _addr(rowstat)
        if rowdot is not None:  # ... or the two weighted sums (<rowdot[0], row>, <rowdot[1], row>) per tile (tavsr_gemm_desc.rowdot_*)
            require_cuda(rowdot[0], rowdot[1])
            d.rowdot_a, d.rowdot_b = _addr(rowdot[0]), _addr(rowdot[1])
    if conv is not None:       # (mode, H, W, C[, stride, taps]): implicit convolution operand (include/tavsr.h)
        d.conv_mode, d.conv_H, d.conv_W, d.conv_C = conv[:4]
        if len(conv) > 4:
            d.conv_stride, d.conv_taps = conv[4], conv[5]
        d.conv_zero = _addr(_zero_page(Cc.device))
    if force is not None and force[1] > 1:
        need = force[1] * max(1, nb1) * max(1, nb2) * M * N + force[1] * M
    else:
        need = lib().tavsr_gemm_ws(C.byref(d))
    if ws_cap is not None:
        need = min(need, int(ws_cap))
    if need > 0:
        ws = torch.empty(need, dtype=f32, device=Cc.device)
        d.ws, d.ws_floats = _addr(ws), need

    def call():
        if ln is not None:
            assert force is None
            require_cuda(ln[0], ln[1], ln[3])
            check(lib().tavsr_gemm_ln(C.byref(d), ptr(ln[0]), ptr(ln[1]), C.c_float(ln[2]), ptr(ln[3]), C.c_int64(ln[3].stride(0)),
                                      stream()), "tavsr_gemm_ln")
        elif force is None:
            check(lib().tavsr_gemm(C.byref(d), stream()), "tavsr_gemm")
        else:
            check(lib().tavsr_gemm_tune(C.byref(d), int(force[0]), int(force[1]), stream()), "tavsr_gemm_tune")

    if PROFILE is None:
        call()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    key = _gemm_kernel_key(d)
    if PROFILE.by_shape:
        key += f" M={M} N={N} K={K} nb={max(1, nb1) * max(1, nb2)}"
    nb = max(1, nb1) * max(1, nb2)
    a_el, b_el = M * K, K * N                 # each operand once; an implicit-conv operand is the image, not its 9 taps
    if conv is not None and conv[0] >= 4:      # Conv3d stem: the clip tensor is the gathered operand
        a_el, b_el = (A.numel(), b_el) if conv[0] in (4, 6) else (a_el, B.numel())
    elif conv is not None:
        a_el, b_el = (a_el // 9, b_el) if conv[0] == 1 else (a_el, b_el // 9)
    c_el = M * N * (1 + (R is not None) + (Z is not None) + (DZ is not None))
    fl = 2.0 * M * N * K * nb
    if conv is not None and conv[0] in (6, 7):    # padded-clip stem: 245 taps are algorithmic, the 288-wide layout is not
        fl = fl * 245.0 / 288.0
    elif conv is not None and conv[0] in (4, 5):  # 4-byte-gather stem: 245 taps padded to 256
        fl = fl * 245.0 / 256.0
    PROFILE.records.append((key, fl, 4.0 * nb * (a_el + b_el + c_el), e0, e1))


def linear(x, w, b=None, *, act=None, alpha=1.0, res=None, save_z=False, out=None, out_off=0, ldc=None, force=None,
           rowstat=None, ln=None):
    """y = res + alpha*act(x @ w.T + b); x [M,K] (row stride x.stride(0)), w [N,K] torch layout.  ``rowstat`` (tensor
    [M, ceil(N / 64), 2]): receives the per-row (sum, sum of squares) of every 64-column tile of y (tavsr_gemm_desc.rowstat).
    ``ln`` = (gamma, beta, eps): also returns LayerNorm(y) - (y, normed) - from the launch that finishes the rows (tavsr_gemm_ln)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = empty(M, N, like=x)
        ldc = N
    z = empty(M, N, like=x) if save_z else None
    assert not save_z or (out_off == 0 and ldc == N)
    normed = empty(M, N, like=x) if ln is not None else None
    assert ln is None or (not save_z and rowstat is None and out_off == 0)
    gemm(M, N, K, x, x.stride(0), w, w.stride(0), out, ldc, c_off=out_off, bias=b, act=act, alpha=alpha, Z=z,
         R=res, ldr=0 if res is None else res.stride(0), force=force, rowstat=rowstat,
         ln=None if ln is None else (ln[0], ln[1], ln[2], normed))
    if ln is not None:
        return out, normed
    return (out, z) if save_z else out


def linear_dx(dy, w, *, alpha=1.0, DZ=None, dact=None, res=None, out=None, force=None):
    """dx = res + alpha * (dy @ w) * act'(DZ);  dy [M,N], w [N,K] -> [M,K]."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = empty(M, K, like=dy)
    gemm(M, K, N, dy, dy.stride(0), w, w.stride(0), out, out.stride(0), b_kmajor=True, alpha=alpha, DZ=DZ, dact=dact,
         R=res, ldr=0 if res is None else res.stride(0), force=force)
    return out


# Dropout fused into GEMM epilogues (tavsr_gemm_desc.drop_*): same mask as the stand-alone kernels draw for the contiguous
# [M, N] result, so a fused forward pairs with a stand-alone backward and vice versa.  TAVSR_GEMM_DROP=0: separate launches.


def _drop_fusable(x, N, K) -> bool:
    return N % 4 == 0 and K % 32 == 0 and K >= 32 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


def rowdot_ok(x, w) -> bool:
    """can the Linear x @ w.T leave weighted row sums of its result (tavsr_gemm_desc.rowstat + rowdot_*: the fast kernel, no K split)?"""
    K, N = x.shape[1], w.shape[0]
    return (N % 64 == 0 and K % 32 == 0 and K <= 1024 and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0)


def linear_drop(x, w, b, p, *, act=None, alpha=1.0, res=None, save_z=False, rowdot=None):
    """out = res + alpha * dropout(act(x @ w.T + b), p) -> (out[, z], token); one launch when the GEMM's 16-byte epilogue
    can carry the mask (else GEMM + dropout / dropout_add launches with the same mask and token).  ``rowdot`` = (a [N], b [N]) with
    ``rowdot_ok(x, w)``: also returns rd [M, N / 64, 2] = per row and 64-column tile (<a, out row>, <b, out row>) as the LAST element."""
    M, K = x.shape
    N = w.shape[0]
    if rowdot is not None:
        assert not save_z and res is None and act is None and rowdot_ok(x, w)
        rd = empty(M, N // 64, 2, like=x)
        out = empty(M, N, like=x)
        tok = _new_token(p, M * N, x.device) if p and p > 0.0 else None
        gemm(M, N, K, x, x.stride(0), w, w.stride(0), out, N, bias=b, alpha=alpha, drop=tok, rowstat=rd, rowdot=rowdot)
        return out, tok, rd
    if not p or p <= 0.0:
        r = linear(x, w, b, act=act, alpha=alpha, res=res, save_z=save_z)
        return (r[0], r[1], None) if save_z else (r, None)
    if _drop_fusable(x, N, K) and w.stride(0) % 4 == 0 and (res is None or res.stride(0) % 4 == 0):
        tok = _new_token(p, M * N, x.device)
        out = empty(M, N, like=x)
        z = empty(M, N, like=x) if save_z else None
        gemm(M, N, K, x, x.stride(0), w, w.stride(0), out, N, bias=b, act=act, alpha=alpha, Z=z, R=res,
             ldr=0 if res is None else res.stride(0), drop=tok)
        return (out, z, tok) if save_z else (out, tok)
    r = linear(x, w, b, act=act, save_z=save_z)
    t, z = (r if save_z else (r, None))
    if res is None:
        out, tok = dropout(t, p, out=t)
        if alpha != 1.0:
            out = axpby(out, None, alpha, 0.0, out=out)
    else:
        out, tok = dropout_add(res, t, p, alpha=alpha)
    return (out, z, tok) if save_z else (out, tok)


def linear_dx_drop(dy, w, tok, *, alpha=1.0, DZ=None, dact=None):
    """dx = mask(tok) / keep * alpha * (dy @ w) * act'(DZ): the backward of y = dropout(act(z)) in the data-gradient GEMM's
    epilogue (tok None: no mask)."""
    M, N = dy.shape
    K = w.shape[1]
    if tok is None:
        return linear_dx(dy, w, alpha=alpha, DZ=DZ, dact=dact)
    if _drop_fusable(dy, K, N) and w.stride(0) % 4 == 0:
        out = empty(M, K, like=dy)
        gemm(M, K, N, dy, dy.stride(0), w, w.stride(0), out, K, b_kmajor=True, alpha=alpha, DZ=DZ, dact=dact, drop=tok)
        return out
    dh = linear_dx(dy, w, alpha=alpha)
    if DZ is not None:
        return dropout_act_bwd(dh, DZ, dact, tok, out=dh)
    return dropout(dh, tok[0], out=dh, token=tok)[0]


def linear_dx_cat(dy_cat, ws, *, res=None, out=None):
    """dx = res + sum_j dy_j @ w_j for projections of ONE input whose output gradients sit side by side in ``dy_cat``
    [M, sum_j N_j] (query / key / value): the weights are stacked into one [sum_j N_j, K] operand (one multi-tensor copy)
    and the sum over j becomes the K loop of a single GEMM instead of a chain of accumulating launches."""
    K = ws[0].shape[1]
    rows = [w.shape[0] for w in ws]
    assert dy_cat.shape[1] == sum(rows) and all(w.shape[1] == K and w.is_contiguous() for w in ws)
    wcat = empty(sum(rows), K, like=dy_cat)
    offs = [sum(rows[:j]) for j in range(len(ws))]
    for i in range(0, len(ws), 24):
        multi_copy_([wcat[o: o + r] for o, r in zip(offs[i: i + 24], rows[i: i + 24])], list(ws[i: i + 24]))
    return linear_dx(dy_cat, wcat, res=res, out=out)


# The two Branchformer branches (attention | cgMLP) are independent between the fork after the macaron FFN and the
# merge: the attention branch is a chain of small latency-bound launches that fits beside the cgMLP GEMMs.


_BRANCH = {}


def forks_enabled() -> bool:
    """TAVSR_SINGLE_STREAM=1 (or ``_lib.SINGLE_STREAM = True`` at run time) puts every launch of the package on the calling
    stream - the reference's queueing; results must not depend on it (tests/test_gpu_streams.py)."""
    return not L.SINGLE_STREAM


def branch_stream(main: torch.cuda.Stream, slot: int = 0) -> torch.cuda.Stream:
    """the side stream that belongs to ``main`` (one per forking stream and ``slot``, so nested forks never share a stream with
    their siblings; ``slot`` > 0: a further side stream of the same owner, for a second section that runs beside the first)."""
    key = (main.device.index, main.cuda_stream, slot)
    s = _BRANCH.get(key)
    if s is None:
        s = _BRANCH[key] = own_stream(main.device)
        L.register_fork(s, main)
    return s


def own_stream(device) -> torch.cuda.Stream:
    """a queue that is nobody else's (tavsr_stream_create): ``torch.cuda.Stream()`` comes out of a round-robin pool of 32 per device, and a
    side stream that later re-appears as some capture's stream would make the fork registry (keyed by raw handles) order that capture
    behind a stream outside it."""
    h = C.c_void_p(0)
    with torch.cuda.device(device):
        check(lib().tavsr_stream_create(C.byref(h)), "tavsr_stream_create")
    return torch.cuda.ExternalStream(h.value, device=device)


# Race amplifier (tests only): TAVSR_RACE_PROBE=<microseconds> enqueues a spin kernel at the head of every forked body
# ("body": the forked stream falls behind its owner - a main-pool block freed too early is overwritten under its readers),
# right after every join on the owning stream ("join": the owner falls behind - a forked-pool block handed out again too early
# is overwritten under the owner's readers), or alternately per scope ("alt", the default).  TAVSR_RACE_PROBE_MODE picks.
RACE_PROBE_US = float(os.environ.get("TAVSR_RACE_PROBE", "0") or 0)
RACE_PROBE_MODE = os.environ.get("TAVSR_RACE_PROBE_MODE", "alt")
_PROBE_TICK = [0]


def spin(us: float) -> None:
    """occupies the current stream for ``us`` microseconds (one wave polling the wall clock)"""
    check(lib().tavsr_spin(C.c_float(us), stream()), "tavsr_spin")


def arm_race_probe(us: float, mode: str = "alt") -> None:
    """(tests) set the amplifier at run time, for the Python-side scopes and the C-side sequencers alike"""
    global RACE_PROBE_US, RACE_PROBE_MODE
    RACE_PROBE_US, RACE_PROBE_MODE = float(us), mode
    check(lib().tavsr_race_probe(C.c_float(us), {"body": 0, "join": 1, "alt": 2}[mode]), "tavsr_race_probe")


def _probe(where: str, tick: int) -> None:
    if RACE_PROBE_US <= 0:
        return
    mode = RACE_PROBE_MODE
    if mode == "alt":
        mode = "body" if tick % 2 == 0 else "join"
    if mode == where:
        spin(RACE_PROBE_US)


class BranchScope:
    """``with BranchScope() as br: <launches>`` enqueues the body on the side stream (ordered after everything
    already on the main stream); ``br.join()`` orders the main stream after the body.
    Allocator safety (the two rules of _lib.py): tensors of the main stream's pool that the body hands to a launch are
    ``record_stream``-ed on the side stream as their pointers are taken (``_lib.ptr`` / ``_lib.addr``) - the main stream can
    free them whenever it likes; tensors allocated in the body live in the side stream's pool and are only handed out again
    to a later body, which starts with a wait on the main stream, so main-stream readers enqueued before that are always
    finished.  Capturable (fork/join inside one hipGraph capture)."""

    def __init__(self, enabled=True, slot=0):
        self.enabled = enabled and forks_enabled()
        self.slot = slot              # which of the owner's side streams (two scopes open at once take different slots)
        self.on = False               # (join() of a scope that was never entered is a no-op)

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        self.on = self.enabled
        if self.on:
            self.side = branch_stream(self.main, self.slot)
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
            self._prev = L.push_fork(self.side)
            self._tick = _PROBE_TICK[0]
            _PROBE_TICK[0] += 1
            _probe("body", self._tick)
        return self

    def __exit__(self, *exc):
        if self.on:
            L.pop_fork(self._prev)
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)
            _probe("join", self._tick)


# Weight gradients beside the backward chain.  A layer's weight gradients (one grouped launch of ~180 us at 12.4 rows per compute unit, the
# (dgamma, dbeta) partial sums behind it) have no consumer inside the backward pass; on the chain's own queue they sit between two layers
# whose first 150 us are one streaming FFN launch and eight latency-bound ones.  ``wgrad_beside(fn)`` enqueues them on the owner's side queue
# instead and does NOT join: the side queue is in-order, so the next layer's forked branch (and its join) comes behind them, and the end of the
# autograd pass joins what is still open (``Variable._execution_engine.queue_callback``).  Only taken when nobody reads the gradients before
# that: every parameter's ``.grad`` is None (AccumulateGrad then keeps the tensor, it does not add into one) and carries no hook other than
# this package's own bucket hooks, which fence themselves (``wgrad_fence``).
WGRAD_BESIDE = os.environ.get("TAVSR_WGRAD_BESIDE", "1") != "0"
WGRAD_SLOT = int(os.environ.get("TAVSR_WGRAD_SLOT", "0"))
_WGRAD_OPEN = {}        # raw handle of the owning stream -> (owner, side) with weight-gradient launches nobody has joined yet


def wgrad_fence() -> None:
    """orders the CURRENT stream behind every weight-gradient launch that ``wgrad_beside`` left open (a no-op when there is none)"""
    if not _WGRAD_OPEN:
        return
    cur = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()
    for owner, side in list(_WGRAD_OPEN.values()):
        if capturing:
            with torch.cuda.stream(side):
                if not torch.cuda.is_current_stream_capturing():
                    continue       # left open by a pass that died before this capture began: not this graph's work (a capture cannot wait for it)
        cur.wait_stream(side)      # (only the calling stream: an owner that is itself a forked stream has been joined already - a wait enqueued
    _WGRAD_OPEN.clear()            # on it now would be work no capture ever joins)


def _wgrad_end_of_pass() -> None:
    wgrad_fence()


def _wgrad_callback() -> bool:
    """queues the end-of-pass join with the autograd pass that is running.  One per caller, not one per pass behind a flag: a pass that dies
    half-way (an out-of-memory error the training loop survives) never runs its callbacks, and a flag it had set would leave every later pass
    un-joined; a second ... twelfth callback of one pass finds nothing open and returns."""
    from torch.autograd import Variable
    try:
        Variable._execution_engine.queue_callback(_wgrad_end_of_pass)
    except RuntimeError:              # not inside an autograd pass (a test driving a backward by hand): nothing would join it
        return False
    return True


def wgrad_may_go_beside(params) -> bool:
    # (not in an instrumented pass: ``PROFILE`` brackets every GEMM launch with events to time the KERNEL - beside the chain a launch shares the
    # chip with whatever runs there, and bench.py's roofline would read 0.47 of peak for a kernel that does 0.63 alone)
    if not (WGRAD_BESIDE and forks_enabled()) or PROFILE is not None:
        return False
    for p_ in params:
        if p_ is None:
            continue
        if p_.grad is not None or p_._backward_hooks:
            return False
        if getattr(p_, "_post_accumulate_grad_hooks", None) and not getattr(p_, "_tavsr_hooks_fence", False):
            return False
    return True


def wgrad_open(main, side) -> bool:
    """(the C-side sequencer enqueues its weight gradients on ``side`` itself) notes them as open; False: not inside an autograd pass"""
    if not _wgrad_callback():
        return False
    _WGRAD_OPEN[main.cuda_stream] = (main, side)
    return True


def wgrad_beside(fn) -> None:
    """``fn()`` (launches only) on the side queue of the current stream, left open until the next ``wgrad_fence`` / the end of the pass"""
    if not _wgrad_callback():
        return fn()
    sc = BranchScope(True, slot=WGRAD_SLOT)
    with sc:
        fn()
    if sc.on:
        _WGRAD_OPEN[sc.main.cuda_stream] = (sc.main, sc.side)


def linear_dw(dy, x, *, alpha=1.0, out=None, bias_grad=False, force=None):
    """dW = alpha * dy.T @ x;  dy [M,N], x [M,K] -> [N,K] (torch weight layout).  With ``bias_grad`` also returns
    db = alpha * dy.sum(0), computed by the same launch from the A fragments (tavsr_gemm a_rowsum)."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = empty(N, K, like=dy)
    gb = empty(N, like=dy) if bias_grad else None
    gemm(N, K, M, dy, dy.stride(0), x, x.stride(0), out, out.stride(0), a_kmajor=True, b_kmajor=True, alpha=alpha,
         a_rowsum=gb, force=force)
    return (out, gb) if bias_grad else out


class WgradGroup:
    """Deferred weight gradients of one backward node: ``add`` allocates the outputs and records the problem,
    ``flush`` computes all of them with ONE grouped launch (tavsr_gemm_grouped; chunks of 12).  The operands must stay
    unchanged between add and flush (the group keeps them alive).  Problems the fast kernel cannot take (unaligned,
    K % 32 != 0) fall back to individual tavsr_gemm calls - same results either way."""

    MAX = 12

    def __init__(self):
        self.items = []

    def add(self, dy, x, *, alpha=1.0, bias_grad=False):
        M, N = dy.shape
        K = x.shape[1]
        require_cuda(dy, x)
        out = empty(N, K, like=dy)
        gb = empty(N, like=dy) if bias_grad else None
        self.items.append((dy, x, float(alpha), out, gb))
        return (out, gb) if bias_grad else out

    @staticmethod
    def _desc(d, dy, x, alpha, out, gb):
        M, N = dy.shape
        K = x.shape[1]
        d.M, d.N, d.K = N, K, M
        d.a_kmajor, d.b_kmajor = 1, 1
        d.A, d.lda = _addr(dy), dy.stride(0)
        d.B, d.ldb = _addr(x), x.stride(0)
        d.C, d.ldc = _addr(out), out.stride(0)
        d.nb1 = d.nb2 = 1
        d.alpha = alpha
        d.a_rowsum = _addr(gb)

    @staticmethod
    def _tiles(it):
        dy, x = it[0], it[1]
        return -(-dy.shape[1] // 64) * -(-x.shape[1] // 64)

    @classmethod
    def _chunks(cls, items):
        """the grouped launches of ``items``: up to MAX problems each; more than MAX problems (many layers at once: the decoder): problems of
        one reduction length together (a launch lasts as long as its longest tile), at most 5 x 256 tiles per launch (what is resident at once)"""
        if len(items) <= cls.MAX:
            return [items] if items else []
        chunks, cur, cur_t = [], [], 0
        for it in sorted(items, key=lambda it: (-it[0].shape[0], -cls._tiles(it))):
            if cur and (len(cur) == cls.MAX or cur_t + cls._tiles(it) > 1280 or it[0].shape[0] != cur[0][0].shape[0]):
                chunks.append(cur)
                cur, cur_t = [], 0
            cur.append(it)
            cur_t += cls._tiles(it)
        chunks.append(cur)
        return chunks

    def flush(self):
        items, self.items = self.items, []
        # All tiles of a group are resident at once (5 block slots x 256 CUs) and equally long (same K), so the launch
        # lasts as long as the fullest CU: with 784 tiles 16 CUs hold 4 blocks and the other 240 idle a quarter of the
        # time.  Shedding the smallest problems down to a multiple of 256 tiles (they go through the planner's own
        # K split) evens that out; only worth it when one or two small problems do it.
        total = sum(self._tiles(it) for it in items)
        if len(items) <= self.MAX and total > 256 and total % 256:
            target, shed, rest = total // 256 * 256, [], []
            for it in sorted(items, key=self._tiles, reverse=True):      # fewest problems that add up to the surplus
                if self._tiles(it) <= total - target:
                    total -= self._tiles(it)
                    shed.append(it)
                else:
                    rest.append(it)
            if total == target and len(shed) <= 2:       # every shed problem costs a launch of its own
                for dy, x, alpha, out, gb in shed:
                    gemm(dy.shape[1], x.shape[1], dy.shape[0], dy, dy.stride(0), x, x.stride(0), out, out.stride(0),
                         a_kmajor=True, b_kmajor=True, alpha=alpha, a_rowsum=gb)
                keep = set(id(it) for it in rest)
                items = [it for it in items if id(it) in keep]
        for chunk in self._chunks(items):
            arr = (GemmDesc * len(chunk))()
            for d, it in zip(arr, chunk):
                self._desc(d, *it)
            if PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = lib().tavsr_gemm_grouped(arr, len(chunk), stream()) if len(chunk) > 1 else -3
            if rc == -3:      # TAVSR_EUNSUPPORTED (or a single problem): one by one through the planner
                for dy, x, alpha, out, gb in chunk:
                    gemm(dy.shape[1], x.shape[1], dy.shape[0], dy, dy.stride(0), x, x.stride(0), out, out.stride(0),
                         a_kmajor=True, b_kmajor=True, alpha=alpha, a_rowsum=gb)
                continue
            check(rc, "tavsr_gemm_grouped")
            if PROFILE is not None:
                e1.record()
                fl = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _, _ in chunk)
                nby = sum(4.0 * (dy.numel() + x.numel() + dy.shape[1] * x.shape[1]) for dy, x, _, _, _ in chunk)
                key = "gemm_kernel<TN>" + (f" grouped x{len(chunk)}" if PROFILE.by_shape else "")
                PROFILE.records.append((key, fl, nby, e0, e1))


def linear_group(x, wbs, out, ldc=None):
    """several y_j = x @ w_j.T + b_j of ONE input in one grouped launch (query/key/value projections): ``wbs`` is a
    list of (w_j, b_j, column offset of y_j in ``out``); out rows have stride ``ldc``.  Falls back to one GEMM per
    projection when the grouped kernel cannot take the problems (alignment / K % 32) - same results."""
    M, K = x.shape
    ldc = out.stride(0) if ldc is None else ldc
    require_cuda(x, out)
    arr = (GemmDesc * len(wbs))()
    for d, (w, b, off) in zip(arr, wbs):
        d.M, d.N, d.K = M, w.shape[0], K
        d.a_kmajor, d.b_kmajor = 0, 0
        d.A, d.lda = _addr(x), x.stride(0)
        d.B, d.ldb = _addr(w), w.stride(0)
        d.C, d.ldc = _addr(out, off), ldc
        d.nb1 = d.nb2 = 1
        d.bias = _addr(b)
        d.alpha = 1.0
    rc = lib().tavsr_gemm_grouped(arr, len(wbs), stream()) if (len(wbs) > 1 and PROFILE is None) else -3
    if rc == -3:
        for w, b, off in wbs:
            gemm(M, w.shape[0], K, x, x.stride(0), w, w.stride(0), out, ldc, c_off=off, bias=b)
        return out
    check(rc, "tavsr_gemm_grouped")
    return out


def colsum(x, *, scale=1.0, out=None, accumulate=False):
    M, N = x.shape
    require_cuda(x)
    if out is None:
        out = empty(N, like=x)
    ws = empty(lib_i64("tavsr_colsum_ws", M, N), like=x)
    check(lib().tavsr_colsum(ptr(x), C.c_int64(x.stride(0)), M, N, C.c_float(scale), ptr(out), int(accumulate), ptr(ws),
                             stream()), "tavsr_colsum")
    return out


def add2_colsum(x, y, out, lazy_sums=False):
    """out = x + y (row-strided 2-D views); returns (column sums of x, column sums of y).  ``lazy_sums``: a third result, a function that
    reduces the launch's partial rows into the two sums - they are bias gradients; the caller runs it with its other weight gradients."""
    M, N = x.shape
    require_cuda(x, y, out)
    sx, sy = empty(N, like=x), empty(N, like=x)
    nws = lib_i64("tavsr_colsum_ws", M, N)
    ws = empty(2 * nws, like=x)
    check(lib().tavsr_add2_colsum(ptr(x), C.c_int64(x.stride(0)), ptr(y), C.c_int64(y.stride(0)), ptr(out),
                                  C.c_int64(out.stride(0)), M, N, ptr(None if lazy_sums else sx), ptr(None if lazy_sums else sy), ptr(ws),
                                  stream()), "tavsr_add2_colsum")
    if not lazy_sums:
        return sx, sy

    def reduce():
        check(lib().tavsr_sum_partials2(ptr(ws), nws // N, C.c_int64(2 * N), ptr(sx), N, ptr(sy), N, 0, stream()), "tavsr_sum_partials2")
    return sx, sy, reduce


def lib_i64(name, *args) -> int:
    fn = getattr(lib(), name)
    fn.restype = C.c_int64
    return int(fn(*args))


# ---------------------------------------------------------------------------------------------- norms
def layernorm_fwd(x, gamma, beta, eps, *, save=True):
    M, D = x.shape
    require_cuda(x, gamma, beta)
    y = empty(M, D, like=x)
    mean = empty(M, like=x) if save else None
    rstd = empty(M, like=x) if save else None
    check(lib().tavsr_layernorm_fwd(ptr(x), C.c_int64(x.stride(0)), ptr(gamma), ptr(beta), C.c_float(eps), ptr(y),
                                    C.c_int64(D), ptr(mean), ptr(rstd), M, D, stream()), "tavsr_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dx_add=None, dx=None):
    """returns dx (= dx_add + LN'(dy)), dgamma, dbeta."""
    M, D = x.shape
    require_cuda(dy, x, mean, rstd, gamma, dx_add)
    if dx is None:
        dx = empty(M, D, like=x)
    dg, db = empty(D, like=x), empty(D, like=x)
    ws = empty(lib_i64("tavsr_layernorm_bwd_ws", M, D), like=x)
    check(lib().tavsr_layernorm_bwd(ptr(dy), C.c_int64(dy.stride(0)), ptr(x), C.c_int64(x.stride(0)), ptr(mean),
                                    ptr(rstd), ptr(gamma), ptr(dx_add),
                                    C.c_int64(0 if dx_add is None else dx_add.stride(0)), ptr(dx),
                                    C.c_int64(dx.stride(0)), ptr(dg), ptr(db), 0, ptr(ws), M, D, stream()),
          "tavsr_layernorm_bwd")
    return dx, dg, db


def layernorm_bwd_act(dy, x, mean, rstd, gamma, z, act, *, dx=None):
    """layernorm_bwd for x = act(z): dx is the gradient w.r.t. z; returns dx, dgamma, dbeta."""
    M, D = x.shape
    require_cuda(dy, x, mean, rstd, gamma, z)
    if dx is None:
        dx = empty(M, D, like=x)
    dg, db = empty(D, like=x), empty(D, like=x)
    ws = empty(lib_i64("tavsr_layernorm_bwd_ws", M, D), like=x)
    check(lib().tavsr_layernorm_bwd_act(ptr(dy), C.c_int64(dy.stride(0)), ptr(x), C.c_int64(x.stride(0)), ptr(mean), ptr(rstd),
                                        ptr(gamma), ptr(dx), C.c_int64(dx.stride(0)), ptr(dg), ptr(db), 0, ptr(ws), M, D, ptr(z),
                                        C.c_int64(z.stride(0)), ACT[act], stream()), "tavsr_layernorm_bwd_act")
    return dx, dg, db


LN_BWD_DROP = True      # masked gradient copy from the LayerNorm backward (tests flip it in-process to compare with the dropout launch)


class LNGroup:
    """LayerNorm backward passes of one backward node whose (dgamma, dbeta) partials share ONE reduction launch:
    ``bwd`` runs the main pass (dx is ready when it returns) and parks the per-block partials in a common slab,
    ``flush`` sums the slab once; the gradient tensors handed out by ``bwd`` are views that are valid after ``flush``.
    LayerNorms of another shape than the group's first one are reduced immediately (same results either way)."""

    def __init__(self, cap: int = 8):
        self.CAP = cap
        self.key, self.slab, self.out, self.k = None, None, None, 0

    def takes(self, M, D) -> bool:
        """the next ``bwd`` of this shape goes through the group's own launch (D = 256: also in the slab form)"""
        return D == 256 and (self.key is None or ((M, D) == self.key and self.k < self.CAP))

    def bwd(self, dy, x, mean, rstd, gamma, *, dx_add=None, dx=None, drop=None):
        """``drop`` (a dropout token): also returns dx * mask / keep as a fourth result (the masked gradient the next residual
        block's branch starts from) - from the same launch when the group takes the call, else from a dropout launch."""
        M, D = x.shape
        if isinstance(dy, DnSlabs):       # (callers ask ``takes`` first: the group's own launch is the only one that reads slabs)
            assert self.takes(M, D) and (drop is None or LN_BWD_DROP)
        if drop is not None and not LN_BWD_DROP:
            r = self.bwd(dy, x, mean, rstd, gamma, dx_add=dx_add, dx=dx)
            return r + (dropout(r[0], drop[0], token=drop)[0],)
        if self.key is None:
            self.key = (M, D)
            self.nb = lib_i64("tavsr_layernorm_bwd_ws", M, D) // (2 * D)
            self.slab = empty(self.nb, self.CAP * 2 * D, like=x)
            self.out = empty(self.CAP * 2 * D, like=x)
        if (M, D) != self.key or self.k >= self.CAP:
            r = layernorm_bwd(dy, x, mean, rstd, gamma, dx_add=dx_add, dx=dx)
            return r if drop is None else r + (dropout(r[0], drop[0], token=drop)[0],)
        slabs = dy if isinstance(dy, DnSlabs) else None
        require_cuda(slabs.ws if slabs else dy, x, mean, rstd, gamma, dx_add)
        if dx is None:
            dx = empty(M, D, like=x)
        off = self.k * 2 * D
        if slabs is not None:
            dxd = empty(M, D, like=x) if drop is not None else None
            assert dx.is_contiguous()
            check(lib().tavsr_layernorm_bwd_partial_slab(ptr(slabs.ws), slabs.wpb, slabs.rb, ptr(x), C.c_int64(x.stride(0)), ptr(mean),
                                                         ptr(rstd), ptr(gamma), ptr(dx_add),
                                                         C.c_int64(0 if dx_add is None else dx_add.stride(0)), ptr(dx),
                                                         C.c_int64(dx.stride(0)), C.c_void_p(_addr(self.slab) + 4 * off),
                                                         C.c_int64(self.slab.stride(0)), M, D, ptr(dxd),
                                                         C.c_float(drop[0] if drop is not None else 0.0),
                                                         ptr(drop[2] if drop is not None else None),
                                                         C.c_uint64(drop[1] if drop is not None else 0), stream()),
                  "tavsr_layernorm_bwd_partial_slab")
            self.k += 1
            r = (dx, self.out[off: off + D], self.out[off + D: off + 2 * D])
            return r if drop is None else r + (dxd,)
        if drop is not None:
            assert dx.is_contiguous()
            dxd = empty(M, D, like=x)
            check(lib().tavsr_layernorm_bwd_partial_drop(ptr(dy), C.c_int64(dy.stride(0)), ptr(x), C.c_int64(x.stride(0)), ptr(mean),
                                                         ptr(rstd), ptr(gamma), ptr(dx_add),
                                                         C.c_int64(0 if dx_add is None else dx_add.stride(0)), ptr(dx),
                                                         C.c_int64(dx.stride(0)), C.c_void_p(_addr(self.slab) + 4 * off),
                                                         C.c_int64(self.slab.stride(0)), M, D, ptr(dxd), C.c_float(drop[0]),
                                                         ptr(drop[2]), C.c_uint64(drop[1]), stream()),
                  "tavsr_layernorm_bwd_partial_drop")
            self.k += 1
            return dx, self.out[off: off + D], self.out[off + D: off + 2 * D], dxd
        check(lib().tavsr_layernorm_bwd_partial(ptr(dy), C.c_int64(dy.stride(0)), ptr(x), C.c_int64(x.stride(0)), ptr(mean),
                                                ptr(rstd), ptr(gamma), ptr(dx_add),
                                                C.c_int64(0 if dx_add is None else dx_add.stride(0)), ptr(dx),
                                                C.c_int64(dx.stride(0)), C.c_void_p(_addr(self.slab) + 4 * off),
                                                C.c_int64(self.slab.stride(0)), M, D, stream()),
              "tavsr_layernorm_bwd_partial")
        self.k += 1
        return dx, self.out[off: off + D], self.out[off + D: off + 2 * D]

    def flush(self):
        if self.k:
            n = self.k * 2 * self.key[1]
            check(lib().tavsr_sum_partials(ptr(self.slab), self.nb, C.c_int64(self.slab.stride(0)), ptr(self.out), n, 0,
                                           stream()), "tavsr_sum_partials")
            self.k = 0


# ---------------------------------------------------------------------------------------------- attention glue
def add_head_bias(q, u, v):
    M, D = q.shape
    qu, qv = empty(M, D, like=q), empty(M, D, like=q)
    require_cuda(q, u, v)
    check(lib().tavsr_add_head_bias(ptr(q), C.c_int64(q.stride(0)), ptr(u), ptr(v), ptr(qu), ptr(qv), C.c_int64(M), D,
                                    stream()), "tavsr_add_head_bias")
    return qu, qv


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def softmax_fwd(ac, bd, klens, scale, causal=False, T2=None, W=0, p_drop=0.0, token=None):
    """ac [H,B,T1,ld_s] (ld_s >= T2, padded rows), bd [H,B,T1,ld_w] or None -> attn like ac.
    ``p_drop`` > 0: also returns (pv, token) = dropout(attn, p_drop) computed by the same launch, with the mask and token
    ``dropout(attn, p_drop)`` would give (``token`` from a previous call reproduces its mask)."""
    H, B, T1, ld_s = ac.shape
    T2 = ld_s if T2 is None else T2
    attn = torch.empty_like(ac)
    ld_w = 0
    if bd is not None:
        ld_w = bd.shape[-1]
        W = W or ld_w
    require_cuda(ac, bd, klens)
    if p_drop and p_drop > 0.0:
        assert ac.is_contiguous() and ld_s % 4 == 0
        pv = torch.empty_like(ac)
        if token is None:
            token = _new_token(p_drop, ac.numel(), ac.device)
        check(lib().tavsr_softmax_dropout_fwd(ptr(ac), ptr(bd), ptr(klens), ptr(attn), ptr(pv), H, B, T1, T2, W, C.c_int64(ld_s),
                                              C.c_int64(ld_w), C.c_float(scale), int(causal), C.c_float(token[0]),
                                              ptr(token[2]), C.c_uint64(token[1]), stream()),
              "tavsr_softmax_dropout_fwd")
        return attn, pv, token
    check(lib().tavsr_softmax_fwd(ptr(ac), ptr(bd), ptr(klens), ptr(attn), H, B, T1, T2, W, C.c_int64(ld_s),
                                  C.c_int64(ld_w), C.c_float(scale), int(causal), stream()), "tavsr_softmax_fwd")
    return attn


def softmax_bwd(attn, dattn, scale, skew=False, T2=None, token=None):
    """``token`` (p, offset): dattn is the gradient of the DROPPED probabilities of softmax_fwd(p_drop=...): the mask is
    regenerated inside the launch."""
    H, B, T1, ld_s = attn.shape
    T2 = ld_s if T2 is None else T2
    ds = torch.empty_like(attn)
    W = 2 * T1 - 1 if skew else 0
    ld_w = pad4(W)
    sk = empty(H, B, T1, ld_w, like=attn) if skew else None
    if token is not None:
        check(lib().tavsr_softmax_dropout_bwd(ptr(attn), ptr(dattn), ptr(ds), ptr(sk), H, B, T1, T2, W, C.c_int64(ld_s),
                                              C.c_int64(ld_w), C.c_float(scale), C.c_float(token[0]), ptr(token[2]),
                                              C.c_uint64(token[1]), stream()), "tavsr_softmax_dropout_bwd")
        return ds, sk
    check(lib().tavsr_softmax_bwd(ptr(attn), ptr(dattn), ptr(ds), ptr(sk), H, B, T1, T2, W, C.c_int64(ld_s),
                                  C.c_int64(ld_w), C.c_float(scale), stream()), "tavsr_softmax_bwd")
    return ds, sk


# Fused attention core (csrc/attn_fused.hip): scores, rel_shift, mask, softmax, dropout and the context product in one
# launch per direction.  TAVSR_ATTN_FUSED=0 keeps the GEMM + softmax chain (A/B switch; also the path for head sizes != 64).
ATTN_FUSED = True


def _attn_desc(q, q_off, k, k_off, v, v_off, B, T1, T2, H, dk, klens, causal, pos, bias_u, bias_v, token):
    require_cuda(q, k, v, klens, pos, bias_u, bias_v)
    d = AttnDesc()
    d.q, d.k, d.v = _addr(q, q_off), _addr(k, k_off), _addr(v, v_off)
    d.ldq, d.ldk, d.ldv = q.stride(0), k.stride(0), v.stride(0)
    if pos is not None:
        d.pos, d.ldp = _addr(pos), pos.stride(0)
    d.bias_u = _addr(bias_u)
    d.bias_v = _addr(bias_v)
    d.klens = _addr(klens)
    d.B, d.H, d.T1, d.T2, d.dk = B, H, T1, T2, dk
    d.scale, d.causal = 1.0 / (dk ** 0.5), int(bool(causal))
    if token is not None:
        d.p_drop, d.drop_offset, d.seed_dev = token[0], token[1], _addr(token[2])
    return d


def attn_fwd(q, q_off, k, k_off, v, v_off, B, T1, T2, H, dk, klens=None, causal=False, pos=None, bias_u=None, bias_v=None,
             p_drop=0.0):
    """q / k / v: 2-D row buffers (row b*T + t; head h at columns off + h*dk ..); pos [2*T1-1, H*dk] projected positions
    (rel-pos form) with the flat pos_bias_u / pos_bias_v.  -> (ctx [B*T1, H*dk], lse [B*H, T1], dropout token or None)."""
    tok = _new_token(p_drop, B * H * T1 * pad4(T2), q.device) if p_drop and p_drop > 0.0 else None
    d = _attn_desc(q, q_off, k, k_off, v, v_off, B, T1, T2, H, dk, klens, causal, pos, bias_u, bias_v, tok)
    ctx = empty(B * T1, H * dk, like=q)
    lse = empty(B * H, T1, like=q)
    check(lib().tavsr_attn_fwd(C.byref(d), ptr(ctx), C.c_int64(ctx.stride(0)), ptr(lse), stream()), "tavsr_attn_fwd")
    return ctx, lse, tok


def attn_bwd(dctx, ctx, lse, tok, q, q_off, k, k_off, v, v_off, B, T1, T2, H, dk, dq, dq_off, dk_buf, dk_off, dv_buf, dv_off,
             klens=None, causal=False, pos=None, bias_u=None, bias_v=None):
    """writes d/d(q + u) rows into dq, dK / dV into dk_buf / dv_buf (head-strided windows at the given offsets); rel-pos:
    returns (dqv [B*T1, H*dk], ds_skew [H, B, T1, pad4(2*T1-1)]) else (None, None)."""
    d = _attn_desc(q, q_off, k, k_off, v, v_off, B, T1, T2, H, dk, klens, causal, pos, bias_u, bias_v, tok)
    require_cuda(dctx, ctx, lse, dq, dk_buf, dv_buf)
    assert dctx.stride(0) == ctx.stride(0)
    dqv = sk = None
    ldw = 0
    if pos is not None:
        dqv = empty(B * T1, H * dk, like=q)
        ldw = pad4(2 * T1 - 1)
        sk = torch.zeros(H, B, T1, ldw, dtype=f32, device=q.device)      # the kernel writes the band of every row
        assert dqv.stride(0) == dq.stride(0)
    check(lib().tavsr_attn_bwd(C.byref(d), ptr(dctx), ptr(ctx), C.c_int64(ctx.stride(0)), ptr(lse),
                               C.c_void_p(_addr(dq, dq_off)), ptr(dqv), C.c_int64(dq.stride(0)),
                               C.c_void_p(_addr(dk_buf, dk_off)), C.c_int64(dk_buf.stride(0)),
                               C.c_void_p(_addr(dv_buf, dv_off)), C.c_int64(dv_buf.stride(0)), ptr(sk), C.c_int64(ldw),
                               stream()), "tavsr_attn_bwd")
    return dqv, sk


# Streaming feed-forward block (csrc/ffn2.hip, round 3): LayerNorm prologue + both GEMMs in one launch whose weights stream
# through per-wave LDS-DMA rings, + a finishing launch that can also emit the LayerNorms the consumers of y start with.
# Shapes it does not take keep the LayerNorm + GEMM + GEMM launches; tests flip these two in-process to compare the routes.
FFN2 = True
FFN2_BWD = True      # the dgrad pair of the block as the streaming kernel too


def ffn2_usable(x, w1, act) -> bool:
    return FFN2 and ffn2_shape_ok(x, w1, act)


def ffn2_shape_ok(x, w1, act) -> bool:
    return (x.dim() == 2 and x.shape[1] == 256 and w1.shape[0] >= 1024 and w1.shape[0] % 32 == 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w1.is_contiguous()
            and act in ("relu", "swish"))


def ffn2_fwd(x, ln_w, ln_b, eps, w1, b1, w2, b2, act, scale, p=0.0, save=True, ln2=(), ln2_eps=1e-12, ln2_stats=False):
    """y = x + scale * dropout(w2 dropout(act(w1 LN(x) + b1)) + b2) and, for every (gamma, beta) in ``ln2`` (at most two),
    LayerNorm(y) * gamma + beta.  Returns (y, saved, ln2_outs, (ln2_mean, ln2_rstd)) with saved = (n, mean, rstd, z, h,
    tok_in, tok_out) as the GEMM path keeps it (None entries when ``save`` is false): the dropout masks are the ones the GEMM
    epilogues would draw, so the GEMM-based backward pairs with this forward."""
    from ._lib import FfnDesc
    M, D = x.shape
    N1 = w1.shape[0]
    require_cuda(x, ln_w, ln_b, w1, b1, w2, b2)
    assert w1.is_contiguous() and w2.is_contiguous() and w1.shape == (N1, D) and w2.shape == (D, N1) and len(ln2) <= 2
    d = FfnDesc()
    d.M, d.D, d.N1, d.act, d.scale, d.eps = M, D, N1, ACT[act], scale, eps
    d.x, d.ldx = _addr(x), x.stride(0)
    d.ln_w, d.ln_b, d.w1, d.b1, d.w2, d.b2 = (_addr(t) for t in (ln_w, ln_b, w1, b1, w2, b2))
    y = empty(M, D, like=x)
    d.y = _addr(y)
    n = mean = rstd = z = h = None
    if save:
        n, mean, rstd = empty(M, D, like=x), empty(M, like=x), empty(M, like=x)
        Mp = (M + 127) // 128 * 128       # whole 128-row blocks are stored
        z, h = empty(Mp, N1, like=x)[:M], empty(Mp, N1, like=x)[:M]
        d.n_out, d.mean, d.rstd, d.z, d.h = (_addr(t) for t in (n, mean, rstd, z, h))
    tok_in = tok_out = None
    if p and p > 0.0:
        tok_in = _new_token(p, M * N1, x.device)
        tok_out = _new_token(p, M * D, x.device)
        d.p_drop, d.seed, d.offset_in, d.offset_out = p, _addr(tok_in[2]), tok_in[1], tok_out[1]
    outs = []
    for k, (g, b) in enumerate(ln2):
        o = empty(M, D, like=x)
        outs.append(o)
        d.ln2_w[k], d.ln2_b[k], d.ln2_out[k] = _addr(g), _addr(b), _addr(o)
    m2 = r2 = None
    if ln2 and ln2_stats:
        m2, r2 = empty(M, like=x), empty(M, like=x)
        d.ln2_mean, d.ln2_rstd = _addr(m2), _addr(r2)
    d.ln2_eps = ln2_eps
    nws = lib_i64("tavsr_ffn2_ws", M, D, N1)
    ws = empty(nws, like=x)
    d.ws, d.ws_floats = _addr(ws), nws
    check(lib().tavsr_ffn2_fwd(C.byref(d), stream()), "tavsr_ffn2_fwd")
    return y, (n, mean, rstd, z, h, tok_in, tok_out), outs, (m2, r2)


FFN2_BWD_LN = os.environ.get("TAVSR_FFN2_BWD_LN", "1") != "0"      # the block's LayerNorm backward sums dn's partials itself (no finishing launch); flipped in tests/test_gpu_switches.py


class DnSlabs:
    """dn of ``ffn2_bwd_dx(..., sum_dn=False)``: the unsummed partials in the launch's workspace (tavsr_ffn2_slab_layout)"""
    __slots__ = ("ws", "wpb", "rb", "shape")

    def __init__(self, ws, wpb, rb, shape):
        self.ws, self.wpb, self.rb, self.shape = ws, wpb, rb, shape


def ffn2_bwd_dx(dyd, alpha, w1, w2, z, act, tok_in, sum_dn=True):
    """(dz [M, N1], dn [M, 256]) of the block: dz = ((alpha * dyd) w2) * mask / keep * act'(z), dn = dz w1 - the streaming
    counterpart of linear_dx_drop + linear_dx (weights untransposed).  ``sum_dn`` False: dn comes back as ``DnSlabs`` for
    ``LNGroup.bwd``, which sums the partials where it reads the rows."""
    M, D = dyd.shape
    N1 = w1.shape[0]
    require_cuda(dyd, w1, w2, z)
    assert w1.is_contiguous() and w2.is_contiguous() and z.is_contiguous() and z.shape == (M, N1)
    dz = empty((M + 127) // 128 * 128, N1, like=dyd)[:M]
    dn = empty(M, D, like=dyd) if sum_dn else None
    nws = lib_i64("tavsr_ffn2_ws", M, D, N1)
    ws = empty(nws, like=dyd)
    check(lib().tavsr_ffn2_bwd_dx(ptr(dyd), C.c_int64(dyd.stride(0)), C.c_float(alpha), ptr(w1), ptr(w2), ptr(z), ACT[act], M, D,
                                  N1, C.c_float(tok_in[0] if tok_in else 0.0), ptr(tok_in[2] if tok_in else None),
                                  C.c_uint64(tok_in[1] if tok_in else 0), ptr(dz), ptr(dn), ptr(ws), C.c_int64(nws), stream()),
          "tavsr_ffn2_bwd_dx")
    if sum_dn:
        return dz, dn
    lay = _SLAB_LAYOUT.get((M, N1))
    if lay is None:
        wpb, rb = C.c_int32(0), C.c_int32(0)
        check(lib().tavsr_ffn2_slab_layout(M, N1, C.byref(wpb), C.byref(rb)), "tavsr_ffn2_slab_layout")
        lay = _SLAB_LAYOUT[(M, N1)] = (wpb.value, rb.value)
    return dz, DnSlabs(ws, lay[0], lay[1], (M, D))


_SLAB_LAYOUT = {}


# One Branchformer layer forward as ONE C call (csrc/layer.hip): the same launches, sequenced in C.  For un-captured loops
# (the host is what limits an eager step); a captured step replays the same kernels either way.  TAVSR_LAYER_C=0: Python sequencing.
BLOCKS_C = os.environ.get("TAVSR_BLOCKS_C", "1") != "0"      # block-level C entry points (tavsr_conv2d_subsample_*, tavsr_cgmlp_fwd) instead of launch-by-launch sequencing
LAYER_C = os.environ.get("TAVSR_LAYER_C", "1") != "0"
LAYER_C_EAGER_ONLY = os.environ.get("TAVSR_LAYER_C", "1") != "capture"     # TAVSR_LAYER_C=capture: also while a hipGraph is being captured
_BR_EVENTS = {}


def branch_events(main: torch.cuda.Stream):
    """(fork, join) events of the side stream that belongs to ``main`` (created once; recorded once so that the handles exist)."""
    key = (main.device.index, main.cuda_stream)
    ev = _BR_EVENTS.get(key)
    if ev is None:
        ev = (torch.cuda.Event(), torch.cuda.Event())
        for e in ev:
            e.record(main)
        _BR_EVENTS[key] = ev
    return ev


def axpby(x, y=None, a=1.0, b=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    assert x.is_contiguous() and (y is None or y.is_contiguous()) and out.is_contiguous()
    require_cuda(x, y, out)
    check(lib().tavsr_axpby(ptr(x), ptr(y), C.c_float(a), C.c_float(b), ptr(out), C.c_int64(x.numel()), stream()),
          "tavsr_axpby")
    return out


def scale_dev(x, s, c=1.0):
    """x * (c * s) with ``s`` a 0-dim / 1-element device tensor."""
    require_cuda(x, s)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    check(lib().tavsr_scale_dev(ptr(x), ptr(s), C.c_float(c), ptr(out), C.c_int64(x.numel()), stream()), "tavsr_scale_dev")
    return out


def axpby2d(x, y, a, b, out):
    M, N = x.shape
    require_cuda(x, y, out)
    check(lib().tavsr_axpby2d(ptr(x), C.c_int64(x.stride(0)), ptr(y), C.c_int64(0 if y is None else y.stride(0)),
                              C.c_float(a), C.c_float(b), ptr(out), C.c_int64(out.stride(0)), C.c_int64(M), N, stream()),
          "tavsr_axpby2d")
    return out


def act_bwd_(dh, z, act):
    """in place: dh *= act'(z)."""
    assert dh.is_contiguous() and z.is_contiguous()
    check(lib().tavsr_act_bwd(ptr(dh), ptr(z), ptr(dh), C.c_int64(dh.numel()), ACT[act], stream()), "tavsr_act_bwd")
    return dh


# ---------------------------------------------------------------------------------------------- cgMLP / merge
def dwconv_gate_fwd(gn, r, w, bias, B, T):
    M, Cn = gn.shape
    K = w.shape[-1]
    out, conv = empty(M, Cn, like=gn), empty(M, Cn, like=gn)
    require_cuda(gn, r, w, bias)
    check(lib().tavsr_dwconv_gate_fwd(ptr(gn), ptr(r), C.c_int64(r.stride(0)), ptr(w), ptr(bias), ptr(out), ptr(conv),
                                      B, T, Cn, K, stream()), "tavsr_dwconv_gate_fwd")
    return out, conv


CSGU_FUSED = True
CSGU_STATS_IN_GEMM = True


def csgu_usable(g, w) -> bool:
    return (CSGU_FUSED and w.shape[-1] == 31 and g.shape[1] % 128 == 0 and g.is_contiguous() and g.data_ptr() % 16 == 0)


def csgu_rowstat_ok(x, w1) -> bool:
    """can channel_proj1's GEMM leave the CSGU's LayerNorm statistics (tavsr_gemm_desc.rowstat)?  (16-byte path, gate half of
    at most 1024 channels in whole 64-column tiles)"""
    return (CSGU_STATS_IN_GEMM and PROFILE is None 
            and w1.is_contiguous() and w1.shape[0] % 128 == 0 and w1.shape[0] <= 2048
            and x.shape[1] % 32 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def csgu_fwd(g, ln_w, ln_b, eps, w, bias, B, T, p=0.0, save=True, rowstat=None):
    """dropout(g[:, :C] * dwconv(LayerNorm(g[:, C:]))) in one pass over g (+ the statistics launch, unless ``rowstat`` - the
    row statistics the GEMM that produced g left, ops.linear(..., rowstat=) - is given).  Returns
    (u, conv, gn, mean, rstd, token); conv / gn only when ``save``."""
    M, C2 = g.shape
    Cn = C2 // 2
    require_cuda(g, ln_w, ln_b, w, bias)
    out = empty(M, Cn, like=g)
    conv, gn = (empty(M, Cn, like=g), empty(M, Cn, like=g)) if save else (None, None)
    mean, rstd = empty(M, like=g), empty(M, like=g)
    tok = _new_token(p, M * Cn, g.device) if p and p > 0.0 else None
    check(lib().tavsr_csgu_fwd(ptr(g), C.c_int64(g.stride(0)), ptr(ln_w), ptr(ln_b), C.c_float(eps), ptr(w), ptr(bias), ptr(out),
                               ptr(gn), ptr(conv), ptr(mean), ptr(rstd), C.c_float(tok[0] if tok else 0.0),
                               ptr(tok[2] if tok else None), C.c_uint64(tok[1] if tok else 0), B, T, Cn, w.shape[-1], ptr(rowstat),
                               stream()),
          "tavsr_csgu_fwd")
    return out, conv, gn, mean, rstd, tok


CGMLP_ACT_BWD_FUSED = True   # gelu' in the CSGU's two backward kernels


def cgmlp_block_ok(x, w1, cw) -> bool:
    """the shapes tavsr_cgmlp_fwd sequences (kernel 31, C <= 1024, one-pass CSGU with the statistics from channel_proj1's epilogue)"""
    C2 = w1.shape[0]
    return (PROFILE is None and CSGU_FUSED and CSGU_STATS_IN_GEMM and cw.shape[-1] == 31 and C2 % 128 == 0 and C2 // 2 <= 1024
            and x.is_contiguous() and x.shape[1] % 32 == 0 and w1.is_contiguous() and cw.is_contiguous())


def cgmlp_fwd(x, w1, b1, ln_w, ln_b, cw, cb, w2, b2, B, T, *, p=0.0, p_out=0.0, alpha=1.0, res=None, save=True):
    """espnet ConvolutionalGatingMLP with the branch's dropout / residual around it as ONE C call (tavsr_cgmlp_fwd, csrc/blocks.hip):
    -> (out, (g, z, gn, gmean, grstd, u, conv, t_u, t_out), desc); ``desc`` is what ``cgmlp_bwd`` wants back."""
    M, D = x.shape
    C2 = w1.shape[0]
    Cn = C2 // 2
    require_cuda(x, res, w1, b1, ln_w, ln_b, cw, cb, w2, b2)
    d = L.CgmlpDesc()
    d.B, d.T, d.D, d.units, d.kernel, d.save = B, T, D, C2, cw.shape[-1], int(save)
    t_u = _new_token(p, M * Cn, x.device) if p and p > 0.0 else None
    t_out = _new_token(p_out, M * D, x.device) if p_out and p_out > 0.0 else None
    d.p_drop, d.p_out, d.alpha = (p if t_u else 0.0), (p_out if t_out else 0.0), alpha
    seed = (t_u or t_out or (0.0, 0, None))[2]
    d.seed, d.off_u, d.off_out = _addr(seed), (t_u[1] if t_u else 0), (t_out[1] if t_out else 0)
    g, u, out = empty(M, C2, like=x), empty(M, Cn, like=x), empty(M, D, like=x)
    gmean, grstd = empty(M, like=x), empty(M, like=x)
    z = gn = conv = None
    if save:
        z, gn, conv = empty(M, C2, like=x), empty(M, Cn, like=x), empty(M, Cn, like=x)
    for f, t in (("x", x), ("res", res), ("w1", w1), ("b1", b1), ("ln_w", ln_w), ("ln_b", ln_b), ("cw", cw), ("cb", cb), ("w2", w2),
                 ("b2", b2), ("g", g), ("g_z", z), ("gn", gn), ("g_mean", gmean), ("g_rstd", grstd), ("u", u), ("conv", conv), ("out", out)):
        setattr(d, f, _addr(t))
    nws = lib_i64("tavsr_cgmlp_ws", C.byref(d))
    ws = empty(max(nws, 4), like=x)
    d.ws, d.ws_floats = _addr(ws), nws
    check(lib().tavsr_cgmlp_fwd(C.byref(d), stream()), "tavsr_cgmlp_fwd")
    return out, (g, z, gn, gmean, grstd, u, conv, t_u, t_out), d


def cgmlp_bwd(desc, dy, params):
    """backward of ``cgmlp_fwd`` (its descriptor; the caller keeps the forward's tensors alive): -> (dx, (g_w1, g_b1, g_ln_w, g_ln_b,
    g_cw, g_cb, g_w2, g_b2)); ``params`` = the eight parameters in that order (shapes of the gradients)."""
    M, D = dy.shape
    require_cuda(dy)
    b = L.CgmlpBwdDesc()
    b.fwd = C.pointer(desc)
    dx = empty(M, D, like=dy)
    grads = [torch.empty_like(q, memory_format=torch.contiguous_format) for q in params]
    b.dy, b.dx = _addr(dy), _addr(dx)
    for f, t in zip(("g_w1", "g_b1", "g_ln_w", "g_ln_b", "g_cw", "g_cb", "g_w2", "g_b2"), grads):
        setattr(b, f, _addr(t))
    nws = lib_i64("tavsr_cgmlp_bwd_ws", C.byref(b))
    ws = empty(max(nws, 4), like=dy)
    b.ws, b.ws_floats = _addr(ws), nws
    check(lib().tavsr_cgmlp_bwd(C.byref(b), stream()), "tavsr_cgmlp_bwd")
    return dx, grads


def conv2d_subsample_ok(x, w1, wo) -> bool:
    B, T, F = x.shape
    Cn = w1.shape[0]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    return PROFILE is None and T >= 7 and F >= 7 and Cn % 64 == 0 and (B * T2 * F2) % 32 == 0 and wo.shape[0] % 4 == 0


def conv2d_subsample_fwd(x, w1, b1, w2, b2, wo, bo, xscale):
    """espnet Conv2dSubsampling as ONE C call (tavsr_conv2d_subsample_fwd): -> (out [B*T2, odim], T2, F2, kept tensors, desc)"""
    B, T, F = x.shape
    Cn, odim = w1.shape[0], wo.shape[0]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    x = x.contiguous()
    require_cuda(x, w1, b1, w2, b2, wo, bo)
    assert all(t.is_contiguous() for t in (w1, w2, wo))
    d = L.SubsampleDesc()
    d.B, d.T, d.F, d.C, d.odim, d.xscale = B, T, F, Cn, odim, xscale
    y1, y2 = empty(B, T1, F1, Cn, like=x), empty(B * T2 * F2, Cn, like=x)
    w2r, wor, out = empty(Cn, 9 * Cn, like=x), empty(odim, F2 * Cn, like=x), empty(B * T2, odim, like=x)
    for f, t in (("x", x), ("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("wo", wo), ("bo", bo), ("zero_page", _zero_page(x.device)),
                 ("y1", y1), ("y2", y2), ("w2r", w2r), ("wor", wor), ("out", out)):
        setattr(d, f, _addr(t))
    nws = lib_i64("tavsr_conv2d_subsample_ws", C.byref(d))
    ws = empty(max(nws, 4), like=x)
    d.ws, d.ws_floats = _addr(ws), nws
    check(lib().tavsr_conv2d_subsample_fwd(C.byref(d), stream()), "tavsr_conv2d_subsample_fwd")
    return out, T2, F2, (x, y1, y2, w2r, wor), d


def conv2d_subsample_bwd(desc, dout, shapes, params=None, kept=()):
    """-> (g_w1, g_b1, g_w2, g_b2, g_wo, g_bo) in the torch layouts ``shapes`` = (w1, w2, wo shapes).  ``params`` (the module's six
    parameters) + ``kept`` (the forward's buffers): the weight gradients may run on the side queue, un-joined (``wgrad_may_go_beside``)."""
    require_cuda(dout)
    w1s, w2s, wos = shapes
    b = L.SubsampleBwdDesc()
    b.fwd = C.pointer(desc)
    grads = [empty(*w1s, like=dout), empty(w1s[0], like=dout), empty(*w2s, like=dout), empty(w2s[0], like=dout),
             empty(*wos, like=dout), empty(wos[0], like=dout)]
    b.dout = _addr(dout)
    for f, t in zip(("g_w1", "g_b1", "g_w2", "g_b2", "g_wo", "g_bo"), grads):
        setattr(b, f, _addr(t))
    nws = lib_i64("tavsr_conv2d_subsample_bwd_ws", C.byref(b))
    ws = empty(max(nws, 4), like=dout)
    b.ws, b.ws_floats = _addr(ws), nws
    main = torch.cuda.current_stream()
    side = branch_stream(main) if (params is not None and forks_enabled() and WGRAD_SLOT == 0) else None
    beside = side is not None and wgrad_may_go_beside(params) and wgrad_open(main, side)
    if beside:
        b.wgrad_beside, b.stream2, b.ev_fork = 1, side.cuda_stream, branch_events(main)[0].cuda_event
    check(lib().tavsr_conv2d_subsample_bwd(C.byref(b), stream()), "tavsr_conv2d_subsample_bwd")
    if beside:      # what the side queue's launches read and write is freed on THIS stream (rule 1 of _lib.py, by hand)
        for t in (ws, dout, *grads, *kept):
            if torch.is_tensor(t) and not isinstance(t, torch.nn.Parameter):
                t.record_stream(side)
    return grads


def dwconv_gate_bwd(du, gn, r, conv, w, dr, B, T, zr=None, act="gelu"):
    """``zr`` (kernel size 31): r = act(zr); dr comes back as the gradient w.r.t. zr."""
    M, Cn = gn.shape
    K = w.shape[-1]
    dgn = empty(M, Cn, like=gn)
    dw, db = torch.empty_like(w), empty(Cn, like=gn)
    ws = empty(lib_i64("tavsr_dwconv_gate_bwd_ws", B, T, Cn, K), like=gn)
    if zr is not None:
        require_cuda(du, gn, r, conv, w, dr, zr)
        check(lib().tavsr_dwconv_gate_bwd_act(ptr(du), ptr(gn), ptr(r), C.c_int64(r.stride(0)), ptr(conv), ptr(w), ptr(dr),
                                              C.c_int64(dr.stride(0)), ptr(dgn), ptr(dw), ptr(db), 0, ptr(ws), B, T, Cn, K,
                                              ptr(zr), C.c_int64(zr.stride(0)), ACT[act], stream()), "tavsr_dwconv_gate_bwd_act")
        return dgn, dw, db
    check(lib().tavsr_dwconv_gate_bwd(ptr(du), ptr(gn), ptr(r), C.c_int64(r.stride(0)), ptr(conv), ptr(w), ptr(dr),
                                      C.c_int64(dr.stride(0)), ptr(dgn), ptr(dw), ptr(db), 0, ptr(ws), B, T, Cn, K,
                                      stream()), "tavsr_dwconv_gate_bwd")
    return dgn, dw, db


def _ptr_array(ts: Sequence[torch.Tensor]):
    require_cuda(*ts)
    return (C.c_void_p * len(ts))(*[_addr(t) for t in ts])


def merge_pool_fwd(x1, x2, lens, params, B, T, lens2=None):
    D = x1.shape[-1]
    score, pooled, w = empty(2, B, T, like=x1), empty(2, B, D, like=x1), empty(B, 2, like=x1)
    require_cuda(x1, x2, lens, lens2)
    check(lib().tavsr_merge_pool_fwd(ptr(x1), ptr(x2), ptr(lens), ptr(lens2), _ptr_array(params), ptr(score), ptr(pooled), ptr(w),
                                     B, T, D, stream()), "tavsr_merge_pool_fwd")
    return score, pooled, w


MERGE_ROWS = True      # the row-parallel learned_ave merge launches (D = 256; tests flip it in-process to compare the routes)


def merge_rows_ok(T, D) -> bool:
    return MERGE_ROWS and bool(lib().tavsr_merge_rows_ok(T, D))


def merge_fwd(x1, x2, lens, params, B, T, lens2=None):
    """merge_pool_fwd + merge_combine: (score, aux, w, w[:, 0] * x1 + w[:, 1] * x2); ``aux`` is what ``merge_bwd`` wants back
    with the other two: the pooled vectors [2, B, D] of the one-workgroup-per-utterance launches, or the row dot products
    [4, B*T] of the row-parallel ones (D = 256)."""
    D = x1.shape[-1]
    score, w = empty(2, B, T, like=x1), empty(B, 2, like=x1)
    out = torch.empty_like(x1)
    require_cuda(x1, x2, lens, lens2)
    assert x1.is_contiguous() and x2.is_contiguous()
    if merge_rows_ok(T, D):
        dots = empty(4, B * T, like=x1)
        check(lib().tavsr_merge_rows_fwd(ptr(x1), ptr(x2), ptr(lens), ptr(lens2), _ptr_array(params), ptr(dots), ptr(score), ptr(w),
                                         ptr(out), B, T, D, stream()), "tavsr_merge_rows_fwd")
        return score, dots, w, out
    score, pooled, w = merge_pool_fwd(x1, x2, lens, params, B, T, lens2=lens2)     # other widths: one workgroup per utterance
    return score, pooled, w, merge_combine(x1, x2, w, B, T)


MERGE_PROJ = os.environ.get("TAVSR_MERGE_PROJ", "1") == "1"      # A/B switch: merge + merge_proj + residual as one launch
MERGE_ROWDOT = True      # the fused tail reads the merge's row dots from the producers' GEMM epilogues (flipped in tests/test_gpu_switches.py)


def merge_proj_ok(x1, x2, w, T, D, res=None) -> bool:
    return (MERGE_PROJ and MERGE_ROWS and w.shape == (D, D) and w.is_contiguous() and x1.is_contiguous() and x2.is_contiguous()
            and (res is None or res.is_contiguous()) and bool(lib().tavsr_merge_proj_ok(T, D)))


def merge_proj_fwd(x1, x2, lens, params, w, b, res, alpha, p, B, T, lens2=None, save=True, rowdots=None):
    """the layer's tail behind the branch join in one launch (csrc/mergeproj.hip): learned_ave merge of x1 / x2 and
    ``res + alpha * dropout(merge_proj(mix), p)`` -> (score, dots, wts, mix | None, out, token).  ``dots`` / ``score`` / ``wts`` are
    what ``merge_bwd`` wants back (the row-parallel route's saved tensors), ``mix`` merge_proj's saved input (``save``)."""
    D = x1.shape[-1]
    M = B * T
    require_cuda(x1, x2, lens, lens2, w, b, res)
    score, wts, dots = empty(2, B, T, like=x1), empty(B, 2, like=x1), empty(4, M, like=x1)
    mix = torch.empty_like(x1) if save else None
    out = torch.empty_like(x1)
    tok = _new_token(p, M * D, x1.device) if p and p > 0.0 else None
    rd1, rd2 = rowdots if rowdots is not None else (None, None)      # the producers' GEMM epilogues left the row dots (linear_drop(rowdot=))
    assert rd1 is None or (rd1.shape == (M, 4, 2) and rd2.shape == (M, 4, 2) and rd1.is_contiguous() and rd2.is_contiguous())
    check(lib().tavsr_merge_proj_fwd_dots(ptr(x1), ptr(x2), ptr(lens), ptr(lens2), _ptr_array(params), ptr(w), ptr(b), ptr(res),
                                          C.c_float(alpha), C.c_float(p if tok is not None else 0.0), ptr(None if tok is None else tok[2]),
                                          C.c_uint64(0 if tok is None else tok[1]), ptr(rd1), ptr(rd2), ptr(dots), ptr(score), ptr(wts),
                                          ptr(mix), ptr(out), B, T, D, stream()), "tavsr_merge_proj_fwd_dots")
    return score, dots, wts, mix, out, tok


def merge_combine(x1, x2, w, B, T):
    D = x1.shape[-1]
    out = torch.empty_like(x1)
    check(lib().tavsr_merge_combine(ptr(x1), ptr(x2), ptr(w), ptr(out), B, T, D, stream()), "tavsr_merge_combine")
    return out


def merge_bwd(dm, x1, x2, lens, params, score, pooled, w, B, T, lens2=None, drop1=None, drop2=None):
    """``pooled``: the ``aux`` of merge_fwd.  ``drop1`` / ``drop2`` (dropout tokens of the two branch outputs): dx1 / dx2 come
    back under those masks - from the same launch on the row-parallel route, by a dropout launch each otherwise."""
    D = x1.shape[-1]
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    # gradient order: weight{pool1,pool2,w1,w2} then bias{pool1,pool2,w1,w2}
    order = [0, 1, 4, 5, 2, 3, 6, 7]
    dparams = [torch.empty_like(params[i]) for i in order]
    if pooled.dim() == 2:          # row dots: the row-parallel launches
        require_cuda(dm, x1, x2, lens, lens2, score, pooled, w)
        assert dm.is_contiguous() and x1.is_contiguous() and x2.is_contiguous()
        ws = empty(lib_i64("tavsr_merge_rows_bwd_ws", B, T, D), like=x1)
        seed = (drop1 or drop2 or (0.0, 0, None))[2]
        assert drop1 is None or drop2 is None or drop1[2] is drop2[2] or drop1[2].data_ptr() == drop2[2].data_ptr()
        p1, o1 = (drop1[0], drop1[1]) if drop1 is not None else (0.0, 0)
        p2, o2 = (drop2[0], drop2[1]) if drop2 is not None else (0.0, 0)
        check(lib().tavsr_merge_rows_bwd(ptr(dm), ptr(x1), ptr(x2), ptr(lens), ptr(lens2), _ptr_array(params), ptr(score), ptr(w),
                                         ptr(pooled), ptr(dx1), ptr(dx2), _ptr_array(dparams), 0, ptr(ws), C.c_float(p1),
                                         C.c_uint64(o1), C.c_float(p2), C.c_uint64(o2), ptr(seed), B, T, D, stream()),
              "tavsr_merge_rows_bwd")
        grads = [None] * 8
        for g, i in zip(dparams, order):
            grads[i] = g
        return dx1, dx2, grads
    ws = empty(lib_i64("tavsr_merge_bwd_ws", B, D), like=x1)
    check(lib().tavsr_merge_bwd(ptr(dm), ptr(x1), ptr(x2), ptr(lens), ptr(lens2), _ptr_array(params), ptr(score), ptr(pooled),
                                ptr(w), ptr(dx1), ptr(dx2), _ptr_array(dparams), 0, ptr(ws), B, T, D, stream()),
          "tavsr_merge_bwd")
    grads = [None] * 8
    for g, i in zip(dparams, order):
        grads[i] = g
    if drop1 is not None:
        dropout(dx1, drop1[0], out=dx1, token=drop1)
    if drop2 is not None:
        dropout(dx2, drop2[0], out=dx2, token=drop2)
    return dx1, dx2, grads


# ---------------------------------------------------------------------------------------------- subsampling
def conv1_fwd(x, w, bias):
    B, T, F = x.shape
    Cn = w.shape[0]
    To, Fo = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    y = empty(B, To, Fo, Cn, like=x)
    require_cuda(x, w, bias)
    check(lib().tavsr_conv1_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, T, F, Cn, stream()), "tavsr_conv1_fwd")
    return y


def conv1_bwd(dz, x, Cn):
    B, T, F = x.shape
    dw, db = empty(Cn, 9, like=x), empty(Cn, like=x)
    ws = empty(lib_i64("tavsr_conv1_bwd_ws", B, T, F, Cn), like=x)
    check(lib().tavsr_conv1_bwd(ptr(dz), ptr(x), ptr(dw), ptr(db), 0, ptr(ws), B, T, F, Cn, stream()), "tavsr_conv1_bwd")
    return dw, db


def im2col3x3s2(y):
    B, Ti, Fi, Cn = y.shape
    To, Fo = (Ti - 3) // 2 + 1, (Fi - 3) // 2 + 1
    col = empty(B * To * Fo, 9 * Cn, like=y)
    check(lib().tavsr_im2col3x3s2(ptr(y), ptr(col), B, Ti, Fi, Cn, stream()), "tavsr_im2col3x3s2")
    return col, To, Fo


def col2im3x3s2_relu(dcol, yrelu):
    B, Ti, Fi, Cn = yrelu.shape
    dz = torch.empty_like(yrelu)
    check(lib().tavsr_col2im3x3s2_relu(ptr(dcol), ptr(yrelu), ptr(dz), B, Ti, Fi, Cn, stream()), "tavsr_col2im3x3s2_relu")
    return dz


def transpose_inner(x, nb, R, Cc, out=None):
    """out[n][c][r] = x[n][r][c] for a contiguous [nb, R, Cc] view of x."""
    require_cuda(x)
    if out is None:
        out = torch.empty_like(x)
    check(lib().tavsr_transpose_inner(ptr(x), ptr(out), C.c_int64(nb), R, Cc, 0, stream()), "tavsr_transpose_inner")
    return out


def utterance_mvn(x, lens):
    B, T, F = x.shape
    require_cuda(x, lens)
    y = torch.empty_like(x)
    check(lib().tavsr_utterance_mvn(ptr(x), ptr(lens), ptr(y), B, T, F, stream()), "tavsr_utterance_mvn")
    return y


# ---------------------------------------------------------------------------------------------- CTC / losses
def ctc_loss(logits, hlens, targets, tlens, blank=0, zero_infinity=True):
    """logits [B,T,V] contiguous -> (nll [B], dnll/dlogits [B,T,V])."""
    B, T, V = logits.shape
    Lmax = targets.shape[1]
    require_cuda(logits, hlens, targets, tlens)
    loss, grad = empty(B, like=logits), torch.empty_like(logits)
    ws = empty(lib_i64("tavsr_ctc_loss_ws", B, T, Lmax), like=logits)
    check(lib().tavsr_ctc_loss(ptr(logits), C.c_int64(V), C.c_int64(T * V), ptr(hlens), ptr(targets),
                               C.c_int64(targets.stride(0)), ptr(tlens), blank, int(zero_infinity), ptr(loss), ptr(grad),
                               ptr(ws), B, T, V, Lmax, stream()), "tavsr_ctc_loss")
    return loss, grad


def ctc_greedy(logits, hlens=None, blank=0, collapse=True):
    B, T, V = logits.shape
    require_cuda(logits, hlens)
    ids = torch.empty((B, T), dtype=torch.int64, device=logits.device)
    hyp = torch.empty((B, T), dtype=torch.int64, device=logits.device) if collapse else None
    hl = torch.empty((B,), dtype=torch.int64, device=logits.device) if collapse else None
    check(lib().tavsr_ctc_greedy(ptr(logits), C.c_int64(V), C.c_int64(T * V), ptr(hlens), blank, ptr(ids), ptr(hyp),
                                 ptr(hl), B, T, V, stream()), "tavsr_ctc_greedy")
    return ids, hyp, hl


def lsm_loss(logits2d, target, ignore, smoothing):
    rows, V = logits2d.shape
    require_cuda(logits2d, target)
    row_loss, grad = empty(rows, like=logits2d), torch.empty_like(logits2d)
    correct = torch.empty((rows,), dtype=torch.int32, device=logits2d.device)
    check(lib().tavsr_lsm_loss(ptr(logits2d), C.c_int64(V), ptr(target), ignore, C.c_float(smoothing), ptr(row_loss),
                               ptr(grad), ptr(correct), C.c_int64(rows), V, stream()), "tavsr_lsm_loss")
    return row_loss, grad, correct


def embed_pe(ids, table, pe, scale, step_dev=None):
    """table[ids] * scale + pe; ``step_dev`` (int32 device scalar): ids is [N, 1] and every row takes pe[step] (tavsr_embed_pe_step)"""
    B, Lq = ids.shape
    D = table.shape[1]
    require_cuda(ids, table, pe)
    out = empty(B, Lq, D, like=table)
    if step_dev is not None:
        assert Lq == 1 and step_dev.dtype == torch.int32 and pe.is_contiguous() and pe.shape[1] == D
        check(lib().tavsr_embed_pe_step(ptr(ids), ptr(table), ptr(pe), C.c_float(scale), ptr(out), C.c_int64(B), pe.shape[0], D,
                                        ptr(step_dev), stream()), "tavsr_embed_pe_step")
        return out
    check(lib().tavsr_embed_pe(ptr(ids), ptr(table), ptr(pe), C.c_float(scale), ptr(out), C.c_int64(B * Lq), Lq, D,
                               stream()), "tavsr_embed_pe")
    return out


def embed_bwd(ids, dout, scale, V):
    D = dout.shape[-1]
    dt = empty(V, D, like=dout)
    check(lib().tavsr_embed_bwd(ptr(ids), ptr(dout), C.c_float(scale), ptr(dt), C.c_int64(ids.numel()), V, D, 0,
                                stream()), "tavsr_embed_bwd")
    return dt


# ---------------------------------------------------------------------------------------------- visual frontend
def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def im2col2d(x, N, H, W, Cn, KH, KW, stride, pad):
    """x [N*H*W, C] channels-last -> col [N*Ho*Wo, KH*KW*C]."""
    require_cuda(x)
    Ho, Wo = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
    col = empty(N * Ho * Wo, KH * KW * Cn, like=x)
    check(lib().tavsr_im2col2d(ptr(x), ptr(col), C.c_int64(N), H, W, Cn, KH, KW, stride, pad, stream()), "tavsr_im2col2d")
    return col, Ho, Wo


def col2im2d(dcol, N, H, W, Cn, KH, KW, stride, pad, extra=None):
    """``extra`` [N*Ho*Wo, C]: data-gradient rows of a parallel 1x1 convolution with the same stride (downsample path),
    added at the pixels it touches."""
    require_cuda(dcol, extra)
    assert extra is None or (extra.is_contiguous() and extra.shape == (dcol.shape[0], Cn))
    dx = empty(N * H * W, Cn, like=dcol)
    check(lib().tavsr_col2im2d(ptr(dcol), ptr(dx), C.c_int64(N), H, W, Cn, KH, KW, stride, pad, ptr(extra), stream()),
          "tavsr_col2im2d")
    return dx


# Conv3d stem as an implicit GEMM (tavsr_gemm_desc.conv_mode 4 / 5): no 6 GB patch matrix.  TAVSR_STEM_IMPLICIT=0 returns to
# im2col_stem + plain GEMMs (A/B switch; also the route for shapes the gather loader does not take).
STEM_IMPLICIT = True


def stem_implicit_ok(x) -> bool:
    B, T, H, W = x.shape
    Ho, Wo = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    return (STEM_IMPLICIT and H % 2 == 0 and W % 2 == 0 and W >= 8 and (B * T * Ho * Wo) % 32 == 0 and T < 1024 and H < 1000 and W < 1000
            and B * T * H * W < 2 ** 31 and x.data_ptr() % 16 == 0)


# Default stem route: zero-padded clips + taps laid out 35 x 8, fetched with the GEMM's ordinary 16-byte LDS-DMA (conv_mode 6 / 7).
# TAVSR_STEM_PAD16=0: the 4-byte gather route (conv_mode 4 / 5, no padded copy of the clips).
STEM_PAD16 = True


def stem_pad16_ok(x) -> bool:
    B, T, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    return (STEM_IMPLICIT and STEM_PAD16 and H % 2 == 0 and W % 4 == 0 and (B * T * Ho * Wo) % 32 == 0
            and B * (T + 5) * (H + 6) * (W + 8) < 2 ** 31)


def stem_pad(x):
    """x [B,T,H,W] -> zero-padded clips [B, T+5, H+6, W+8] (2/3 frames, 3/3 rows, 3/5 columns): every tap of the (5,7,7)
    stride (1,2,2) stem is inside the buffer, also the zero-weight tap columns."""
    return torch.nn.functional.pad(x, (3, 5, 3, 3, 2, 3)).contiguous()


def stem_weight_288(w):
    """Conv3d weight [Cout,1,5,7,7] -> [Cout, 288]: column ((kt*7 + kh)*8 + kw), zeros at kw = 7 and in the last 8 columns."""
    co = w.shape[0]
    return torch.nn.functional.pad(torch.nn.functional.pad(w.reshape(co, 35, 7), (0, 1)).reshape(co, 280), (0, 8)).contiguous()


def stem_weight_grad_from_288(g, shape):
    co = g.shape[0]
    return g[:, :280].reshape(co, 35, 8)[:, :, :7].reshape(shape).contiguous()


def stem_conv_fwd_pad16(xp, w288, B, T, H, W):
    """padded clips xp (stem_pad), w288 [Cout, 288] -> z [B*T*Ho*Wo, Cout]."""
    Ho, Wo = H // 2, W // 2
    M = B * T * Ho * Wo
    z = empty(M, w288.shape[0], like=xp)
    gemm(M, w288.shape[0], 288, xp, 4, w288, 288, z, w288.shape[0], conv=(6, H + 6, W + 8, T + 5))
    return z, Ho, Wo


def stem_conv_dw_pad16(dz, xp, T, H, W):
    """dW [Cout, 288] = dz^T patches(xp) in the 35 x 8 tap layout (the zero-weight columns receive values nobody reads)."""
    M, cout = dz.shape
    dw = empty(cout, 288, like=dz)
    gemm(cout, 288, M, dz, cout, xp, 4, dw, 288, a_kmajor=True, b_kmajor=True, conv=(7, H + 6, W + 8, T + 5))
    return dw


def stem_conv_fwd(x, w0p):
    """x [B,T,H,W] clips, w0p [64, 256] (245 taps of the (5,7,7) kernel + zero columns) -> z [B*T*Ho*Wo, 64]."""
    B, T, H, W = x.shape
    Ho, Wo = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    M = B * T * Ho * Wo
    z = empty(M, w0p.shape[0], like=x)
    gemm(M, w0p.shape[0], 256, x, 4, w0p, 256, z, w0p.shape[0], conv=(4, H, W, T))
    return z, Ho, Wo


def stem_conv_dw(dz, x):
    """dW0 [64, 256] = dz^T patches(x) (columns >= 245 are zero); dz [B*T*Ho*Wo, 64]."""
    B, T, H, W = x.shape
    M, cout = dz.shape
    dw = empty(cout, 256, like=dz)
    gemm(cout, 256, M, dz, cout, x, 4, dw, 256, a_kmajor=True, b_kmajor=True, conv=(5, H, W, T))
    return dw


def im2col_stem(x):
    """x [B,T,H,W] -> col [B*T*Ho*Wo, 256] (245 taps of the (5,7,7) kernel + zero padding)."""
    B, T, H, W = x.shape
    require_cuda(x)
    Ho, Wo = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    col = empty(B * T * Ho * Wo, 256, like=x)
    check(lib().tavsr_im2col_stem(ptr(x), ptr(col), B, T, H, W, stream()), "tavsr_im2col_stem")
    return col, Ho, Wo


def bn_stats(x, eps, momentum, running_mean=None, running_var=None, nbt=None):
    M, Cn = x.shape
    require_cuda(x, running_mean, running_var, nbt)
    mean, var, rstd = empty(Cn, like=x), empty(Cn, like=x), empty(Cn, like=x)
    ws = empty(lib_i64("tavsr_bn_ws", C.c_int64(M), Cn), like=x)
    check(lib().tavsr_bn_stats(ptr(x), C.c_int64(M), Cn, C.c_float(eps), C.c_float(momentum), ptr(mean), ptr(var), ptr(rstd),
                               ptr(running_mean), ptr(running_var), ptr(nbt), ptr(ws), stream()), "tavsr_bn_stats")
    return mean, rstd


def rsqrt_eps(v, eps):
    out = torch.empty_like(v)
    check(lib().tavsr_rsqrt_eps(ptr(v), C.c_float(eps), ptr(out), C.c_int64(v.numel()), stream()), "tavsr_rsqrt_eps")
    return out


def bn_apply_fwd(x, mean, rstd, gamma, beta, res=None, act=None):
    M, Cn = x.shape
    require_cuda(x, mean, rstd, gamma, beta, res)
    y = torch.empty_like(x)
    check(lib().tavsr_bn_apply_fwd(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(res), ptr(y), C.c_int64(M), Cn,
                                   ACT[act], stream()), "tavsr_bn_apply_fwd")
    return y


def bn_bwd(dy, x, mean, rstd, gamma, beta, res=None, act=None, need_dz=True):
    """returns dz (gradient w.r.t. the pre-activation, = gradient of ``res``), dx, dgamma, dbeta.  ``need_dz=False`` (only
    without ``res``): dz is not materialised (returned as None), which saves one [M, C] write."""
    M, Cn = x.shape
    require_cuda(dy, x, mean, rstd, gamma, beta, res)
    assert need_dz or res is None
    dz, dx = (torch.empty_like(x) if need_dz else None), torch.empty_like(x)
    dg, db = empty(Cn, like=x), empty(Cn, like=x)
    ws = empty(lib_i64("tavsr_bn_ws", C.c_int64(M), Cn), like=x)
    check(lib().tavsr_bn_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(res), ptr(dz), ptr(dx), ptr(dg),
                             ptr(db), C.c_int64(M), Cn, ACT[act], ptr(ws), stream()), "tavsr_bn_bwd")
    return dz, dx, dg, db


def bn_bwd_pooled(dpool, idx, x, mean, rstd, gamma, beta, N, H, W, act=None):
    """bn_bwd (no residual) whose incoming gradient is the pooled one of maxpool3x3s2_fwd(act(bn(x))): -> dx, dgamma, dbeta."""
    M, Cn = x.shape
    require_cuda(dpool, idx, x, mean, rstd, gamma, beta)
    assert M == N * H * W and dpool.is_contiguous() and idx.dtype == torch.uint8 and idx.is_contiguous()
    dx = torch.empty_like(x)
    dg, db = empty(Cn, like=x), empty(Cn, like=x)
    ws = empty(lib_i64("tavsr_bn_ws", C.c_int64(M), Cn), like=x)
    check(lib().tavsr_bn_bwd_pooled(ptr(dpool), ptr(idx), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dg),
                                    ptr(db), C.c_int64(N), H, W, Cn, ACT[act], ptr(ws), stream()), "tavsr_bn_bwd_pooled")
    return dx, dg, db


def maxpool3x3s2_fwd(x, N, H, W, Cn):
    Ho, Wo = conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1)
    y = empty(N * Ho * Wo, Cn, like=x)
    idx = torch.empty((N * Ho * Wo, Cn), dtype=torch.uint8, device=x.device)
    check(lib().tavsr_maxpool3x3s2_fwd(ptr(x), ptr(y), ptr(idx), C.c_int64(N), H, W, Cn, stream()), "tavsr_maxpool3x3s2_fwd")
    return y, idx, Ho, Wo


def bn_act_maxpool3x3s2_fwd(z, mean, rstd, gamma, beta, act, N, H, W, Cn):
    """maxpool(act(BatchNorm(z))) without writing the activation map: -> (pooled, idx, Ho, Wo)."""
    Ho, Wo = conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1)
    require_cuda(z, mean, rstd, gamma, beta)
    y = empty(N * Ho * Wo, Cn, like=z)
    idx = torch.empty((N * Ho * Wo, Cn), dtype=torch.uint8, device=z.device)
    check(lib().tavsr_bn_act_maxpool3x3s2_fwd(ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(y), ptr(idx),
                                              C.c_int64(N), H, W, Cn, stream()), "tavsr_bn_act_maxpool3x3s2_fwd")
    return y, idx, Ho, Wo


def maxpool3x3s2_bwd(dy, idx, N, H, W, Cn):
    dx = empty(N * H * W, Cn, like=dy)
    check(lib().tavsr_maxpool3x3s2_bwd(ptr(dy), ptr(idx), ptr(dx), C.c_int64(N), H, W, Cn, stream()), "tavsr_maxpool3x3s2_bwd")
    return dx


def avgpool_fwd(x, N, P, Cn):
    y = empty(N, Cn, like=x)
    check(lib().tavsr_avgpool_fwd(ptr(x), ptr(y), C.c_int64(N), P, Cn, stream()), "tavsr_avgpool_fwd")
    return y


def avgpool_bwd(dy, N, P, Cn):
    dx = empty(N * P, Cn, like=dy)
    check(lib().tavsr_avgpool_bwd(ptr(dy), ptr(dx), C.c_int64(N), P, Cn, stream()), "tavsr_avgpool_bwd")
    return dx


def conv_wflip(w2d, cout, cin):
    require_cuda(w2d)
    out = empty(cin, 9 * cout, like=w2d)
    check(lib().tavsr_conv_wflip(ptr(w2d), ptr(out), cout, cin, stream()), "tavsr_conv_wflip")
    return out


def conv3x3_fwd(x, w2d, H, W, stride=1, taps=9, pad0=False, bias=None, act=None):
    """implicit 3x3/p1 (taps 9) or 1x1/p0 (taps 1) convolution with stride: x [images*H*W, Cin] channels-last image rows,
    w2d [Cout, taps*Cin] -> [images*Ho*Wo, Cout].  ``pad0``: the 3x3 window without padding (Conv2dSubsampling)."""
    M, cin = x.shape
    cout = w2d.shape[0]
    if pad0:
        assert taps == 9
        Ho, Wo = (H - 3) // stride + 1, (W - 3) // stride + 1
    else:
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    Mo = M // (H * W) * Ho * Wo
    z = empty(Mo, cout, like=x)
    gemm(Mo, cout, taps * cin, x, cin, w2d, taps * cin, z, cout, conv=(1, H, W, cin, stride, 90 if pad0 else taps), bias=bias, act=act)
    return z


def conv3x3_dx(dz, wflip, H, W, res=None):
    """data gradient: dz [pixels, Cout], wflip [Cin, 9*Cout] (conv_wflip) -> [pixels, Cin] (+ res: the skip path's
    gradient, added in the GEMM epilogue)."""
    M, cout = dz.shape
    cin = wflip.shape[0]
    dx = empty(M, cin, like=dz)
    gemm(M, cin, 9 * cout, dz, cout, wflip, 9 * cout, dx, cin, conv=(1, H, W, cout), R=res, ldr=0 if res is None else cin)
    return dx


def conv3x3_dw(dz, x, H, W, stride=1, taps=9, pad0=False, bias_grad=False):
    """weight gradient: dz [output pixels, Cout], x [images*H*W, Cin] -> [Cout, taps*Cin]; needs output pixels % 32 == 0.
    ``bias_grad``: also the column sums of dz (the bias gradient) from the same launch."""
    M, cout = dz.shape
    cin = x.shape[1]
    dw = empty(cout, taps * cin, like=dz)
    gb = empty(cout, like=dz) if bias_grad else None
    gemm(cout, taps * cin, M, dz, cout, x, cin, dw, taps * cin, a_kmajor=True, b_kmajor=True,
         conv=(2, H, W, cin, stride, 90 if pad0 else taps), a_rowsum=gb)
    return (dw, gb) if bias_grad else dw


def fill_(t, value):
    require_cuda(t)
    assert t.is_contiguous()
    check(lib().tavsr_fill(ptr(t), C.c_float(value), C.c_int64(t.numel()), stream()), "tavsr_fill")
    return t


def copy2d(src, dst):
    """dst[m, :N] = src[m, :N] for 2-D (row-strided) views of equal shape; no alignment requirement."""
    require_cuda(src, dst)
    assert src.shape == dst.shape and src.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    check(lib().tavsr_copy2d(ptr(src), C.c_int64(src.stride(0)), ptr(dst), C.c_int64(dst.stride(0)), C.c_int64(src.shape[0]),
                             C.c_int64(src.shape[1]), stream()), "tavsr_copy2d")
    return dst


# ---------------------------------------------------------------------------------------------- log-mel frontend, SpecAug
def stft_frames(wav, window, T, n_fft, hop, center=True):
    """wav [B, N] -> windowed frames [B*T, n_fft] (reflect padding of n_fft/2 when center)."""
    B, N = wav.shape
    require_cuda(wav, window)
    frames = empty(B * T, n_fft, like=wav)
    check(lib().tavsr_stft_frames(ptr(wav), ptr(window), ptr(frames), B, C.c_int64(N), T, n_fft, hop, int(center), stream()),
          "tavsr_stft_frames")
    return frames


def power_spec(spec, nfreq, ldp, B, T, olens):
    """spec [B*T, >= 2*nfreq] (re | im) -> power [B*T, ldp] (zero K padding, zero past olens)."""
    require_cuda(spec, olens)
    P = empty(B * T, ldp, like=spec)
    check(lib().tavsr_power_spec(ptr(spec), C.c_int64(spec.stride(0)), ptr(P), ldp, nfreq, B, T, ptr(olens), stream()),
          "tavsr_power_spec")
    return P


def log_mask(mel, B, T, olens, floor=1e-10):
    require_cuda(mel, olens)
    out = torch.empty_like(mel)
    check(lib().tavsr_log_mask(ptr(mel), ptr(out), B, T, mel.shape[-1], ptr(olens), C.c_float(floor), stream()), "tavsr_log_mask")
    return out


def time_warp(x, center, warped, lens):
    """center / warped / lens: int64 [B] on the device (center 0 = leave that utterance unwarped)."""
    B, T, F = x.shape
    require_cuda(x, center, warped, lens)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    check(lib().tavsr_time_warp(ptr(x), ptr(y), B, T, F, ptr(center), ptr(warped), ptr(lens), stream()), "tavsr_time_warp")
    return y


def specaug_mask_(x, fpos=None, flen=None, tpos=None, tlen=None):
    """in place; band arrays [B, n] int64 on the device (None: that axis is not masked)."""
    B, T, F = x.shape
    require_cuda(x, fpos, flen, tpos, tlen)
    assert x.is_contiguous()
    nf = 0 if fpos is None else fpos.shape[1]
    nt = 0 if tpos is None else tpos.shape[1]
    check(lib().tavsr_specaug_mask(ptr(x), B, T, F, ptr(fpos), ptr(flen), nf, ptr(tpos), ptr(tlen), nt, stream()),
          "tavsr_specaug_mask")
    return x


def bucket_copy(ptrs, offs, sizes, n, flat, scale, to_flat, max_n):
    """pack (to_flat) or unpack-and-scale the tensors listed in the device tables into / from ``flat`` (tavsr/dp.py)."""
    require_cuda(ptrs, offs, sizes, flat)
    check(lib().tavsr_bucket_copy(ptr(ptrs), ptr(offs), ptr(sizes), int(n), ptr(flat), C.c_float(scale), int(bool(to_flat)),
                                  C.c_int64(max_n), stream()), "tavsr_bucket_copy")


def multi_add_(dst, src):
    """dst[t] += src[t] for lists of contiguous fp32 tensors, one launch per 24 tensors."""
    assert len(dst) == len(src)
    if not dst:
        return dst
    require_cuda(*dst, *src)
    n = len(dst)
    for a, b in zip(dst, src):
        assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() and a.dtype == f32 and b.dtype == f32
    dp = (C.c_void_p * n)(*[_addr(t) for t in dst])
    sp = (C.c_void_p * n)(*[_addr(t) for t in src])
    cnt = (C.c_int64 * n)(*[t.numel() for t in dst])
    check(lib().tavsr_multi_add(dp, sp, cnt, n, stream()), "tavsr_multi_add")
    return dst


def multi_copy_(dst, src, inc=None):
    """dst[t].copy_(src[t]) for lists of contiguous same-shape/dtype tensors in ONE launch (<= 24 tensors); ``inc`` (int64 tensor):
    every element incremented by one in the same launch."""
    assert len(dst) == len(src)
    require_cuda(*dst, *src)
    n = len(dst)
    for a, b in zip(dst, src):
        assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.numel() == b.numel()
    dp = (C.c_void_p * n)(*[_addr(t) for t in dst])
    sp = (C.c_void_p * n)(*[_addr(t) for t in src])
    cnt = (C.c_int64 * n)(*[t.numel() * t.element_size() for t in dst])
    if inc is not None:
        require_cuda(inc)
        assert inc.dtype == torch.int64 and inc.is_contiguous()
        check(lib().tavsr_multi_copy_inc(dp, sp, cnt, n, ptr(inc), inc.numel(), stream()), "tavsr_multi_copy_inc")
        return dst
    check(lib().tavsr_multi_copy(dp, sp, cnt, n, stream()), "tavsr_multi_copy")
    return dst


# ---------------------------------------------------------------------------------------------- batch assembly
def video_prep(src, index, T, y0, x0, th, tw, flip, affine, masked, out, pad_value):
    """one clip -> its row of the padded batch (tavsr_video_prep); ``index`` / ``masked`` are host lists / arrays or None."""
    require_cuda(src, out)
    dev = src.device
    Ts, H, W = src.shape
    frames = None if index is None else torch.tensor(index, dtype=torch.int32).to(dev)
    if frames is not None and T > 0 and (min(index) < 0 or max(index) >= Ts):
        raise ValueError("frame index outside the clip")
    m = None if masked is None or not masked.any() else torch.from_numpy(masked.astype("uint8")).to(dev)
    ws = empty(th * tw, like=out) if m is not None else None
    n = len(affine)
    mean = (C.c_float * 4)(*[a[0] for a in affine], *([0.0] * (4 - n)))
    std = (C.c_float * 4)(*[a[1] for a in affine], *([1.0] * (4 - n)))
    check(lib().tavsr_video_prep(ptr(src), int(src.dtype == torch.uint8), Ts, H, W, ptr(frames), int(T), int(y0), int(x0), int(th),
                                 int(tw), int(bool(flip)), mean, std, n, ptr(m), ptr(ws), ptr(out), int(out.shape[0]),
                                 C.c_float(pad_value), stream()), "tavsr_video_prep")
    return out


def add_noise(audio, noise, inv_snr):
    require_cuda(audio, noise)
    assert audio.numel() == noise.numel() and audio.is_contiguous() and noise.is_contiguous()
    out = torch.empty_like(audio)
    check(lib().tavsr_add_noise(ptr(audio), ptr(noise), ptr(out), C.c_int64(audio.numel()), C.c_float(inv_snr), stream()),
          "tavsr_add_noise")
    return out


# ---------------------------------------------------------------------------------------------- error rates
RESAMPLE_ROLLOFF, RESAMPLE_ZEROS, RESAMPLE_BETA = 0.9475937167399596, 64, 14.769656459379492      # "kaiser_best" filter constants


def resample(audio, factor: float):
    """band-limited resampling of a mono waveform [1, T] or [T] by ``factor`` (output length round(T / factor)): the clip played
    ``factor`` times faster at the same sample rate (tavsr_resample_sinc)"""
    require_cuda(audio)
    x = audio.float().contiguous().view(-1)
    fn = lib().tavsr_resample_len
    fn.restype = C.c_int64
    n_out = int(fn(C.c_int64(x.numel()), C.c_double(factor)))
    y = empty(n_out, like=x)
    check(lib().tavsr_resample_sinc(ptr(x), C.c_int64(x.numel()), ptr(y), C.c_int64(n_out), C.c_double(factor), C.c_double(RESAMPLE_ROLLOFF),
                                    RESAMPLE_ZEROS, C.c_double(RESAMPLE_BETA), stream()), "tavsr_resample_sinc")
    return y.view(*audio.shape[:-1], n_out)


def edit_distance(ref, ref_off, hyp, hyp_off, n_pairs, max_len):
    """Levenshtein distance of n_pairs (reference, hypothesis) id sequences packed as (int32 ids, int64 offsets)."""
    require_cuda(ref, ref_off, hyp, hyp_off)
    assert ref.dtype == torch.int32 and hyp.dtype == torch.int32 and ref_off.dtype == torch.int64 and hyp_off.dtype == torch.int64
    assert ref_off.numel() == n_pairs + 1 and hyp_off.numel() == n_pairs + 1
    dist = torch.empty(n_pairs, dtype=torch.int32, device=ref.device)
    check(lib().tavsr_edit_distance(ptr(ref), ptr(ref_off), ptr(hyp), ptr(hyp_off), int(n_pairs), int(max_len), ptr(dist),
                                    stream()), "tavsr_edit_distance")
    return dist


def bootstrap_rates(dist, reflen, iters, seed):
    """100 * sum(dist[idx]) / sum(reflen[idx]) for ``iters`` resamples idx of the n sentences -> float64 [iters]."""
    require_cuda(dist, reflen)
    assert dist.dtype == torch.int32 and reflen.dtype == torch.int32 and dist.numel() == reflen.numel()
    rates = torch.empty(iters, dtype=torch.float64, device=dist.device)
    check(lib().tavsr_bootstrap_rates(ptr(dist), ptr(reflen), dist.numel(), int(iters), C.c_uint64(seed), ptr(rates), stream()),
          "tavsr_bootstrap_rates")
    return rates


# ---------------------------------------------------------------------------------------------- decode steps
ROWLIN = True


def rowlin_ok(x, w, n_rows=None, ln=False, ksplit=1) -> bool:
    """does the one-launch small-step Linear (tavsr_rowlin / tavsr_rowlin_parts) take this call?  Up to 32 rows, K a power of two
    in 64 .. 2048; which (rows, K, LayerNorm prologue, K slices) combinations have a spill-free plan is the library's answer
    (tavsr_rowlin_ok: e.g. K = 2048 with more than 16 rows only in slices, K = 1024 with LayerNorm only up to 16 rows)."""
    if isinstance(x, RowParts):
        x = x.t[0]
    K = x.shape[1]
    N = x.shape[0] if n_rows is None else n_rows
    return bool(ROWLIN and N <= 32 and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0
                and x.stride(1) == 1 and w.stride(1) == 1 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
                and lib().tavsr_rowlin_ok(int(N), int(K), int(bool(ln)), int(ksplit)))


class RowParts:
    """rows held as a SUM of tensors: ``t`` [P, N, D] contiguous - what a K-split ``rowlin`` leaves and the next launches of the
    chain add while they load (tavsr_rowlin_parts)"""
    __slots__ = ("t",)

    def __init__(self, t):
        assert t.dim() == 3 and t.is_contiguous()
        self.t = t

    @property
    def shape(self):
        return self.t.shape[1:]


ROWLIN_KSPLIT = 4      # K slices of the 2048 -> d projection of a one-token feed-forward block; 1: one block per tile (<= 16 rows only)


def rowlin(x, w, b=None, *, ln=None, act=None, res=None, out=None, gather=None, ksplit=1, res_gather=None):
    """out = res + act(LN(x[gather]) @ w.T + b) in one launch (csrc/decode.hip:rowlin_kernel) - the Linear layers of a
    one-token scorer step.  ln = (gamma, beta, eps) or None; gather: int64 row indices into x (embedding lookup).
    ``x`` / ``res`` may be ``RowParts``; ``ksplit`` > 1 deals K to that many blocks per column tile and returns ``RowParts``."""
    if isinstance(x, RowParts) or isinstance(res, RowParts) or ksplit > 1:
        assert gather is None and res_gather is None
        xt, xp = (x.t[0], x.t.shape[0]) if isinstance(x, RowParts) else (x, 1)
        rt, rp = (res.t[0], res.t.shape[0]) if isinstance(res, RowParts) else (res, 1)
        N, K, Nout = xt.shape[0], xt.shape[1], w.shape[0]
        require_cuda(xt, w)
        assert w.shape[1] == K and out is None
        g, be, eps = ln if ln is not None else (None, None, 0.0)
        o = empty(ksplit, N, Nout, like=xt) if ksplit > 1 else empty(N, Nout, like=xt)
        check(lib().tavsr_rowlin_parts(ptr(xt), C.c_int64(xt.stride(0)), xp, C.c_int64(x.t.stride(0) if xp > 1 else 0), ptr(g),
                                       ptr(be), C.c_float(eps), ptr(w), C.c_int64(w.stride(0)), ptr(b), ACT[act], ptr(rt),
                                       C.c_int64(0 if rt is None else rt.stride(0)), rp, C.c_int64(res.t.stride(0) if rp > 1 else 0),
                                       ptr(o), C.c_int64(Nout), ksplit, C.c_int64(N * Nout if ksplit > 1 else 0), N, K, Nout,
                                       stream()), "tavsr_rowlin_parts")
        return RowParts(o) if ksplit > 1 else o
    N = x.shape[0] if gather is None else gather.numel()
    K, Nout = x.shape[1], w.shape[0]
    require_cuda(x, w)
    assert w.shape[1] == K and (res_gather is None or (res is not None and res_gather.numel() == N and res_gather.dtype == torch.int64))
    if out is None:
        out = empty(N, Nout, like=x)
    assert out.data_ptr() != x.data_ptr()
    g, be, eps = ln if ln is not None else (None, None, 0.0)
    check(lib().tavsr_rowlin(ptr(x), C.c_int64(x.stride(0)), ptr(gather), ptr(g), ptr(be), C.c_float(eps), ptr(w),
                             C.c_int64(w.stride(0)), ptr(b), ACT[act], ptr(res), C.c_int64(0 if res is None else res.stride(0)),
                             ptr(res_gather), ptr(out), C.c_int64(out.stride(0)), N, K, Nout, stream()), "tavsr_rowlin")
    return out


def tree_attn_step(q, kpool, vpool, anc, nkeys, H, dk, out=None, step_dev=None, k_new=None, v_new=None, group=1):
    """q [N, H*dk] (row stride q.stride(0)); kpool/vpool [nodes, H*dk]; anc int32 [N, >= nkeys] -> [N, H*dk].
    ``step_dev`` (int32 device scalar): use min(step + 1, nkeys) keys (graph replays).  ``k_new`` / ``v_new`` (row stride of
    q): this step's keys / values - the last key of every hypothesis - appended to the pools by the same launch."""
    N = q.shape[0]
    require_cuda(q, kpool, vpool, anc)
    assert anc.dtype == torch.int32 and kpool.stride(0) == vpool.stride(0)
    assert k_new is None or (k_new.stride(0) == q.stride(0) and v_new.stride(0) == q.stride(0))
    if out is None:
        out = empty(N, H * dk, like=q)
    check(lib().tavsr_tree_attn_step(ptr(q), C.c_int64(q.stride(0)), ptr(kpool), ptr(vpool), C.c_int64(kpool.stride(0)),
                                     ptr(anc), C.c_int64(anc.stride(0)), int(nkeys), ptr(out), C.c_int64(out.stride(0)), N, H, dk,
                                     C.c_float(1.0 / (dk ** 0.5)), ptr(step_dev), ptr(k_new), ptr(v_new), int(group), stream()),
          "tavsr_tree_attn_step")
    return out


def kv_append(k, v, kpool, vpool, N, max_steps, step_dev):
    """kpool/vpool[step * N + n] = k[n] / v[n] with the step index read from device memory."""
    require_cuda(k, v, kpool, vpool, step_dev)
    assert step_dev.dtype == torch.int32 and k.stride(0) == v.stride(0) and kpool.stride(0) == vpool.stride(0)
    assert kpool.shape[0] >= max_steps * N
    check(lib().tavsr_kv_append(ptr(k), ptr(v), C.c_int64(k.stride(0)), ptr(kpool), ptr(vpool), C.c_int64(kpool.stride(0)),
                                N, k.shape[1], int(max_steps), ptr(step_dev), stream()), "tavsr_kv_append")


def ctc_prefix_step(logp, lens, r_prev, s_prev, last_tok, cand, K, out_len, blank=0, step_dev=None):
    """-> (r_new [N,T,2,C], psi [N,C], psi_abs [N,C], eos [N], eos_abs [N]); see include/tavsr.h.
    ``step_dev`` (int32 device scalar) replaces ``out_len`` (graph replays)."""
    U, T, V = logp.shape
    N, Cn = cand.shape
    require_cuda(logp, lens, r_prev, s_prev, last_tok, cand)
    r_new = empty(N, T, 2, Cn, like=logp)
    psi, psi_abs = empty(N, Cn, like=logp), empty(N, Cn, like=logp)
    eos, eos_abs = empty(N, like=logp), empty(N, like=logp)
    check(lib().tavsr_ctc_prefix_step(ptr(logp), ptr(lens), ptr(r_prev), ptr(s_prev), ptr(last_tok), ptr(cand), ptr(r_new),
                                      ptr(psi), ptr(psi_abs), ptr(eos), ptr(eos_abs), N, K, T, V, Cn,
                                      0 if step_dev is not None else int(out_len), blank, ptr(step_dev), stream()),
          "tavsr_ctc_prefix_step")
    return r_new, psi, psi_abs, eos, eos_abs


def ctc_prefix_step_topk(logp, lens, r_prev, s_prev, last_tok, full, Cn, K, out_len, blank=0, step_dev=None):
    """pre-beam + prefix scores in one launch: -> (cand [N,C] int64, r_new, psi, psi_abs, eos, eos_abs); see include/tavsr.h."""
    U, T, V = logp.shape
    N = full.shape[0]
    require_cuda(logp, lens, r_prev, s_prev, last_tok, full)
    assert full.is_contiguous() and full.shape[1] == V
    cand = torch.empty(N, Cn, dtype=torch.int64, device=logp.device)
    r_new = empty(N, T, 2, Cn, like=logp)
    psi, psi_abs = empty(N, Cn, like=logp), empty(N, Cn, like=logp)
    eos, eos_abs = empty(N, like=logp), empty(N, like=logp)
    check(lib().tavsr_ctc_prefix_step_topk(ptr(logp), ptr(lens), ptr(r_prev), ptr(s_prev), ptr(last_tok), ptr(full), ptr(cand),
                                           ptr(r_new), ptr(psi), ptr(psi_abs), ptr(eos), ptr(eos_abs), N, K, T, V, Cn,
                                           0 if step_dev is not None else int(out_len), blank, ptr(step_dev), stream()),
          "tavsr_ctc_prefix_step_topk")
    return cand, r_new, psi, psi_abs, eos, eos_abs


def act_(x, act):
    """x = act(x) in place."""
    require_cuda(x)
    assert x.is_contiguous()
    check(lib().tavsr_act_fwd(ptr(x), ptr(x), C.c_int64(x.numel()), ACT[act], stream()), "tavsr_act_fwd")
    return x


def log_softmax_rows(x, V=None, out=None, alpha=1.0, add=0.0, accumulate=False):
    """out = (out if accumulate else 0) + alpha * log_softmax(x[:, :V]) + add"""
    M = x.shape[0]
    V = x.shape[1] if V is None else V
    require_cuda(x)
    if out is None:
        out = empty(M, V, like=x)
    assert not accumulate or out is not None
    check(lib().tavsr_log_softmax_rows(ptr(x), C.c_int64(x.stride(0)), ptr(out), C.c_int64(out.stride(0)), M, V, C.c_float(alpha),
                                       C.c_float(add), int(bool(accumulate)), stream()), "tavsr_log_softmax_rows")
    return out


def beam_combine(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score, eos, w_ctc):
    """weighted = full + w_ctc * (partial CTC scorer row) + score; fixes psi_abs of <eos> candidates in place (tavsr.h)."""
    N, V = full.shape
    Cn = cand.shape[1]
    require_cuda(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score)
    weighted = empty(N, V, like=full)
    check(lib().tavsr_beam_combine(ptr(full), ptr(cand), ptr(psi), ptr(psi_abs), ptr(eos_s), ptr(eos_abs), ptr(s_prev), ptr(score),
                                   ptr(weighted), N, V, Cn, int(eos), C.c_float(w_ctc), stream()), "tavsr_beam_combine")
    return weighted


def beam_combine_topk_ok(K, V) -> bool:
    return K * V <= 8192


def beam_combine_topk(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score, eos, w_ctc, K, keep_weighted=False):
    """``beam_combine`` and ``torch.topk(weighted.view(U, K * V), K)`` in one launch -> (top_s, top_i[, weighted])."""
    N, V = full.shape
    Cn = cand.shape[1]
    require_cuda(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score)
    top_s = empty(N // K, K, like=full)
    top_i = torch.empty(N // K, K, dtype=torch.int64, device=full.device)
    weighted = empty(N, V, like=full) if keep_weighted else None
    check(lib().tavsr_beam_combine_topk(ptr(full), ptr(cand), ptr(psi), ptr(psi_abs), ptr(eos_s), ptr(eos_abs), ptr(s_prev), ptr(score),
                                        ptr(weighted), ptr(top_s), ptr(top_i), N, K, V, Cn, int(eos), C.c_float(w_ctc), stream()),
          "tavsr_beam_combine_topk")
    return (top_s, top_i, weighted) if keep_weighted else (top_s, top_i)


def beam_select_topk_ok(K, V) -> bool:
    return V <= 64 and K <= 16 and K <= V


def beam_select_topk(dec, z_lm, w_lm, add, psi_all, psi_abs_all, eos_s, eos_abs, s_prev, score, eos, w_ctc, K, Cn, keep=False):
    """the beam update behind the scorers in one launch (tavsr_beam_select_topk; V <= 64): -> (top_s, top_i) or, with ``keep``,
    (top_s, top_i, full, weighted, cand) - the intermediate values the separate launches would have produced"""
    N, V = dec.shape
    require_cuda(dec, z_lm, psi_all, psi_abs_all, eos_s, eos_abs, s_prev, score)
    assert dec.is_contiguous() and psi_all.is_contiguous() and psi_abs_all.is_contiguous() and (z_lm is None or z_lm.is_contiguous())
    top_s = empty(N // K, K, like=dec)
    top_i = torch.empty(N // K, K, dtype=torch.int64, device=dec.device)
    full = empty(N, V, like=dec) if keep else None
    weighted = empty(N, V, like=dec) if keep else None
    cand = torch.empty(N, Cn, dtype=torch.int64, device=dec.device) if keep else None
    check(lib().tavsr_beam_select_topk(ptr(dec), ptr(z_lm), C.c_float(w_lm), C.c_float(add), ptr(psi_all), ptr(psi_abs_all), ptr(eos_s),
                                       ptr(eos_abs), ptr(s_prev), ptr(score), ptr(full), ptr(weighted), ptr(cand), ptr(top_s), ptr(top_i),
                                       N, K, V, int(Cn), int(eos), C.c_float(w_ctc), stream()), "tavsr_beam_select_topk")
    return (top_s, top_i, full, weighted, cand) if keep else (top_s, top_i)


def beam_step_begin(score, tok, anc, maxlen, K, eos, step_dev):
    """head of a captured search step (tavsr.h): ended hypotheses leave the beam, anc[:, step] names this step's rows."""
    N = score.numel()
    require_cuda(score, tok, anc, maxlen, step_dev)
    assert anc.dtype == torch.int32 and maxlen.dtype == torch.int32 and tok.dtype == torch.int64 and step_dev.dtype == torch.int32
    check(lib().tavsr_beam_step_begin(ptr(score), ptr(tok), ptr(anc), anc.stride(0), ptr(maxlen), N, K, int(eos), ptr(step_dev),
                                      stream()), "tavsr_beam_step_begin")


def beam_reorder(top_i, top_s, cand, r_new, psi_abs, yseq, anc, outs, K, V, step_dev, hist=None, maxlen=None, eos=0):
    """gathers the state of the extended slots into ``outs`` = (r, s, yseq, anc, tok, score) buffers (tavsr.h).  ``maxlen`` (int32
    [N / K]): the head of the next step rides along (tavsr_beam_reorder_begin)."""
    N, Cn = cand.shape
    T = r_new.shape[1]
    r_out, s_out, y_out, a_out, t_out, sc_out = outs
    require_cuda(top_i, top_s, cand, r_new, psi_abs, yseq, anc, *outs, step_dev, hist)
    assert hist is None or (hist.dtype == torch.int32 and hist.is_contiguous() and hist.shape[1:] == (3, N))
    assert top_i.is_contiguous() and top_s.is_contiguous() and top_i.numel() == N and yseq.dtype == torch.int64
    assert anc.dtype == torch.int32 and y_out.shape == yseq.shape and a_out.shape == anc.shape and r_out.shape == (N, T, 2)
    if maxlen is not None:
        require_cuda(maxlen)
        assert maxlen.dtype == torch.int32 and maxlen.numel() == N // K
        check(lib().tavsr_beam_reorder_begin(ptr(top_i), ptr(top_s), ptr(cand), ptr(r_new), ptr(psi_abs), ptr(yseq), ptr(anc), ptr(r_out),
                                             ptr(s_out), ptr(y_out), ptr(a_out), ptr(t_out), ptr(sc_out), N, K, V, Cn, T, yseq.stride(0),
                                             anc.stride(0), ptr(step_dev), ptr(hist), 0 if hist is None else hist.shape[0],
                                             ptr(maxlen), int(eos), stream()), "tavsr_beam_reorder_begin")
        return
    check(lib().tavsr_beam_reorder(ptr(top_i), ptr(top_s), ptr(cand), ptr(r_new), ptr(psi_abs), ptr(yseq), ptr(anc), ptr(r_out),
                                   ptr(s_out), ptr(y_out), ptr(a_out), ptr(t_out), ptr(sc_out), N, K, V, Cn, T, yseq.stride(0),
                                   anc.stride(0), ptr(step_dev), ptr(hist), 0 if hist is None else hist.shape[0], stream()),
          "tavsr_beam_reorder")


# ---------------------------------------------------------------------------------------------- dropout
# The generator state is ONE uint64 per device, resident in HBM.  ``rng_step_begin`` advances it with a kernel (so a
# captured step graph draws new masks at every replay), writes the advanced value into a fresh one-element tensor that
# belongs to THIS forward pass, and rewinds the per-step site counter, which hands every dropout call of the pass its own,
# reproducible counter range.  A dropout call returns the token (p, offset, step seed tensor) that regenerates its mask
# in the backward pass: the token owns the seed it was drawn with, so a second forward pass (micro-batches, a validation
# pass, another model) before the first backward cannot change the masks that backward regenerates.
_RNG = {}
_STEP_SEED = {}
_SITE = [0]


def _dev_index(device=None) -> int:
    return torch.cuda.current_device() if device is None or device.index is None else device.index


def rng_state(device=None) -> torch.Tensor:
    dev = _dev_index(device)
    t = _RNG.get(dev)
    if t is None:
        t = _RNG[dev] = torch.full((1,), 0x5EED5EED, dtype=torch.int64, device=f"cuda:{dev}")
    return t


def manual_seed(seed: int, device=None):
    """the seeding hook of the device generator (dropout masks); data-parallel ranks seed with base + rank (dp.init_from_env)."""
    rng_state(device).fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)
    _STEP_SEED.pop(_dev_index(device), None)
    _SITE[0] = 0


def rng_step_begin(device=None):
    """call once per forward pass (before its first dropout site)."""
    dev = _dev_index(device)
    step = torch.empty(1, dtype=torch.int64, device=f"cuda:{dev}")     # inside a graph capture: lives in the graph's pool
    check(lib().tavsr_rng_step(ptr(rng_state(device)), ptr(step), stream()), "tavsr_rng_step")
    _STEP_SEED[dev] = step
    _SITE[0] = 0


def step_seed(device=None) -> torch.Tensor:
    """the seed tensor new dropout sites draw from: the current pass's copy (the live state before any rng_step_begin)."""
    t = _STEP_SEED.get(_dev_index(device))
    return rng_state(device) if t is None else t


def _new_token(p, n, device):
    tok = (float(p), _SITE[0], step_seed(device))
    _SITE[0] += (n + 3) // 4 * 4
    return tok


def dropout(x, p: float, out=None, token=None):
    """y = dropout(x, p); returns (y, token).  ``token`` from a previous call reproduces that call's mask."""
    require_cuda(x)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    n = x.numel()
    if token is None:
        token = _new_token(p, n, x.device)
    check(lib().tavsr_dropout(ptr(x), ptr(out), C.c_int64(n), C.c_float(token[0]), ptr(token[2]),
                              C.c_uint64(token[1]), stream()), "tavsr_dropout")
    return out, token


def dropout_add(a, t, p: float, alpha: float = 1.0, out=None):
    """y = a + alpha * dropout(t, p) in one pass; returns (y, token) - the token is the one ``dropout(t)`` would give."""
    require_cuda(a, t)
    assert a.is_contiguous() and t.is_contiguous() and a.numel() == t.numel()
    if out is None:
        out = torch.empty_like(a)
    n = t.numel()
    token = _new_token(p, n, a.device)
    check(lib().tavsr_dropout_add(ptr(a), ptr(t), ptr(out), C.c_int64(n), C.c_float(p), C.c_float(alpha),
                                  ptr(token[2]), C.c_uint64(token[1]), stream()), "tavsr_dropout_add")
    return out, token


def dropout_act_bwd(dh, z, act, token, out=None):
    """dz = dropout_mask(dh) * act'(z) with the mask of ``token`` (the site that dropped act(z) in the forward pass)."""
    require_cuda(dh, z)
    assert dh.is_contiguous() and z.is_contiguous() and dh.numel() == z.numel()
    if out is None:
        out = torch.empty_like(dh)
    check(lib().tavsr_dropout_act_bwd(ptr(dh), ptr(z), ptr(out), C.c_int64(dh.numel()), C.c_float(token[0]), ACT[act],
                                      ptr(token[2]), C.c_uint64(token[1]), stream()), "tavsr_dropout_act_bwd")
    return out
