"""ctypes binding of libtavsr_hip.so (C ABI declared in include/tavsr.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
``TavsrError`` is raised.  Device pointers are passed as raw addresses (``tensor.data_ptr()``) and
every call is enqueued on torch's current HIP stream, so the calls compose with torch's caching
allocator, stream semantics and HIP-graph capture.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TAVSR_LIB selects another build of the same library (profiles/gemm_trace.py's instrumented one); never a fallback.
LIB_PATH = os.environ.get("TAVSR_LIB") or os.path.join(_HERE, "lib", "libtavsr_hip.so")

ACT = {None: 0, "none": 0, "relu": 1, "swish": 2, "gelu": 3}


class TavsrError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("nb1", C.c_int32), ("nb2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64),
        ("bias", C.c_void_p),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("Z", C.c_void_p),
        ("R", C.c_void_p), ("ldr", C.c_int64), ("sR1", C.c_int64), ("sR2", C.c_int64),
        ("DZ", C.c_void_p), ("dact", C.c_int32),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("a_rowsum", C.c_void_p),
        ("conv_mode", C.c_int32), ("conv_H", C.c_int32), ("conv_W", C.c_int32), ("conv_C", C.c_int32),
        ("conv_zero", C.c_void_p), ("conv_stride", C.c_int32), ("conv_taps", C.c_int32),
        ("drop_p", C.c_float), ("drop_seed", C.c_void_p), ("drop_offset", C.c_uint64),
        ("rowstat", C.c_void_p), ("rowdot_a", C.c_void_p), ("rowdot_b", C.c_void_p),
    ]


class AttnDesc(C.Structure):
    """tavsr_attn_desc (include/tavsr.h)"""
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64),
        ("pos", C.c_void_p), ("ldp", C.c_int64),
        ("bias_u", C.c_void_p), ("bias_v", C.c_void_p),
        ("klens", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("T1", C.c_int32), ("T2", C.c_int32), ("dk", C.c_int32),
        ("scale", C.c_float), ("causal", C.c_int32), ("p_drop", C.c_float),
        ("seed_dev", C.c_void_p), ("drop_offset", C.c_uint64),
    ]


class BfLayerDesc(C.Structure):
    """tavsr_bf_layer_desc (include/tavsr.h): field order is the header's"""
    _P = C.c_void_p
    _fields_ = (
        [(n, C.c_int32) for n in ("B", "T", "D", "H", "ffn_units", "cg_units", "cg_kernel", "ffn_act", "save")]
        + [(n, C.c_float) for n in ("p_drop", "p_att", "coeff")]
        + [(n, C.c_void_p) for n in (
            "x", "pos_emb", "lens",
            "ffm_ln_w", "ffm_ln_b", "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2",
            "mha_ln_w", "mha_ln_b", "wq", "bq", "wk", "bk", "wv", "bv", "wpos", "pos_u", "pos_v", "wo", "bo",
            "mlp_ln_w", "mlp_ln_b", "cg_w1", "cg_b1", "csgu_ln_w", "csgu_ln_b", "csgu_cw", "csgu_cb", "cg_w2", "cg_b2")]
        + [("merge_p", C.c_void_p * 8)]
        + [(n, C.c_void_p) for n in (
            "merge_w", "merge_b", "ff_ln_w", "ff_ln_b", "ff_w1", "ff_b1", "ff_w2", "ff_b2", "final_ln_w", "final_ln_b", "seed")]
        + [("drop_off", C.c_uint64 * 9)]
        + [(n, C.c_void_p) for n in (
            "x1", "ffm_n", "ffm_mean", "ffm_rstd", "ffm_z", "ffm_h",
            "n_mha", "n_mlp", "br_mean", "br_rstd",
            "qkv", "pp", "cx", "lse", "xa",
            "g", "g_z", "gn", "g_mean", "g_rstd", "u", "conv", "xm",
            "score", "pooled", "wts", "m",
            "x2", "ff_n", "ff_mean", "ff_rstd", "ff_z", "ff_h", "x3", "y", "fin_mean", "fin_rstd",
            "stream2", "ev_fork", "ev_join", "ws")]
        + [("ws_floats", C.c_int64)]
    )


class BfLayerBwdDesc(C.Structure):
    """tavsr_bf_layer_bwd_desc (include/tavsr.h)"""
    _fields_ = (
        [("fwd", C.POINTER(BfLayerDesc)), ("dy", C.c_void_p), ("dx", C.c_void_p)]
        + [(n, C.c_void_p) for n in (
            "g_ffm_w1", "g_ffm_b1", "g_ffm_w2", "g_ffm_b2",
            "g_wq", "g_bq", "g_wk", "g_bk", "g_wv", "g_bv", "g_wo", "g_bo", "g_wpos", "g_pos_u", "g_pos_v",
            "g_cg_w1", "g_cg_b1", "g_csgu_ln_w", "g_csgu_ln_b", "g_csgu_cw", "g_csgu_cb", "g_cg_w2", "g_cg_b2")]
        + [("g_merge_p", C.c_void_p * 8)]
        + [(n, C.c_void_p) for n in ("g_merge_w", "g_merge_b", "g_ff_w1", "g_ff_b1", "g_ff_w2", "g_ff_b2", "g_ln", "ws")]
        + [("ws_floats", C.c_int64), ("wgrad_beside", C.c_int32)]
    )


class TailoredStreamDesc(C.Structure):
    """tavsr_tailored_stream_desc (include/tavsr.h): field order is the header's"""
    _fields_ = (
        [(n, C.c_int32) for n in ("B", "T", "D", "H", "ffn_units", "cg_units", "cg_kernel", "ffn_act", "save", "use_attn")]
        + [(n, C.c_float) for n in ("p_drop", "p_att", "coeff")]
        + [(n, C.c_void_p) for n in (
            "x", "pos_emb", "lens",
            "ffm_ln_w", "ffm_ln_b", "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2", "br_ln_w", "br_ln_b",
            "wq", "bq", "wk", "bk", "wv", "bv", "wpos", "pos_u", "pos_v", "wo", "bo",
            "cg_w1", "cg_b1", "csgu_ln_w", "csgu_ln_b", "csgu_cw", "csgu_cb", "cg_w2", "cg_b2",
            "ff_ln_w", "ff_ln_b", "ff_w1", "ff_b1", "ff_w2", "ff_b2", "final_ln_w", "final_ln_b", "seed")]
        + [("drop_off", C.c_uint64 * 6)]
        + [(n, C.c_void_p) for n in (
            "x1", "ffm_n", "ffm_mean", "ffm_rstd", "ffm_z", "ffm_h", "n_br", "br_mean", "br_rstd",
            "qkv", "pp", "cx", "lse", "g", "g_z", "gn", "g_mean", "g_rstd", "u", "conv",
            "x2", "ff_n", "ff_mean", "ff_rstd", "ff_z", "ff_h", "x3", "y", "fin_mean", "fin_rstd", "ws")]
        + [("ws_floats", C.c_int64)]
    )


class TailoredLayerDesc(C.Structure):
    """tavsr_tailored_layer_desc (include/tavsr.h)"""
    _fields_ = [("audio", C.POINTER(TailoredStreamDesc)), ("video", C.POINTER(TailoredStreamDesc)), ("stream2", C.c_void_p),
                ("ev_fork", C.c_void_p), ("ev_join", C.c_void_p)]


class CgmlpDesc(C.Structure):
    """tavsr_cgmlp_desc (include/tavsr.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("B", "T", "D", "units", "kernel", "save")]
                + [(n, C.c_float) for n in ("p_drop", "p_out", "alpha")]
                + [("seed", C.c_void_p), ("off_u", C.c_uint64), ("off_out", C.c_uint64)]
                + [(n, C.c_void_p) for n in ("x", "res", "w1", "b1", "ln_w", "ln_b", "cw", "cb", "w2", "b2",
                                             "g", "g_z", "gn", "g_mean", "g_rstd", "u", "conv", "out", "ws")]
                + [("ws_floats", C.c_int64)])


class CgmlpBwdDesc(C.Structure):
    """tavsr_cgmlp_bwd_desc (include/tavsr.h)"""
    _fields_ = ([("fwd", C.POINTER(CgmlpDesc)), ("dy", C.c_void_p), ("dx", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("g_w1", "g_b1", "g_ln_w", "g_ln_b", "g_cw", "g_cb", "g_w2", "g_b2", "ws")]
                + [("ws_floats", C.c_int64)])


class SubsampleDesc(C.Structure):
    """tavsr_subsample_desc (include/tavsr.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("B", "T", "F", "C", "odim")] + [("xscale", C.c_float)]
                + [(n, C.c_void_p) for n in ("x", "w1", "b1", "w2", "b2", "wo", "bo", "zero_page", "y1", "y2", "w2r", "wor", "out", "ws")]
                + [("ws_floats", C.c_int64)])


class SubsampleBwdDesc(C.Structure):
    """tavsr_subsample_bwd_desc (include/tavsr.h)"""
    _fields_ = ([("fwd", C.POINTER(SubsampleDesc)), ("dout", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("g_w1", "g_b1", "g_w2", "g_b2", "g_wo", "g_bo", "ws")]
                + [("ws_floats", C.c_int64), ("wgrad_beside", C.c_int32), ("stream2", C.c_void_p), ("ev_fork", C.c_void_p)])


class FfnDesc(C.Structure):
    """tavsr_ffn_desc (include/tavsr.h)"""
    _fields_ = [
        ("M", C.c_int32), ("D", C.c_int32), ("N1", C.c_int32), ("act", C.c_int32),
        ("scale", C.c_float), ("eps", C.c_float),
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("res", C.c_void_p), ("ldr", C.c_int64),
        ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
        ("b2", C.c_void_p),
        ("y", C.c_void_p),
        ("p_drop", C.c_float),
        ("seed", C.c_void_p),
        ("offset_in", C.c_uint64), ("offset_out", C.c_uint64),
        ("n_out", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("z", C.c_void_p), ("h", C.c_void_p),
        ("ln2_w", C.c_void_p * 2), ("ln2_b", C.c_void_p * 2), ("ln2_out", C.c_void_p * 2),
        ("ln2_mean", C.c_void_p), ("ln2_rstd", C.c_void_p),
        ("ln2_eps", C.c_float),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TavsrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.tavsr_last_error_string.restype = C.c_char_p
        _lib.tavsr_version.restype = C.c_int
        _lib.tavsr_gemm_ws.restype = C.c_int64
        us = float(os.environ.get("TAVSR_RACE_PROBE", "0") or 0)
        if us > 0:      # race amplifier of the C-side sequencers (tests; ops.py arms the Python-side scopes from the same variables)
            _lib.tavsr_race_probe(C.c_float(us), {"body": 0, "join": 1, "alt": 2}[os.environ.get("TAVSR_RACE_PROBE_MODE", "alt")])
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().tavsr_last_error_string().decode()
        raise TavsrError(f"{what} failed (rc={rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> C.c_void_p:
    """torch's current HIP stream of the current device as a raw handle (the fast private accessor when torch has it:
    torch.cuda.current_stream() builds a Stream object per call, ~9 us, and a step makes ~1100 calls)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------- stream safety
# The package runs independent sections on forked HIP streams (ops.BranchScope; autograd replays a node on the stream its
# forward ran on).  Two rules make that safe against torch's per-stream caching allocator, and both are enforced HERE, at the
# one place where a tensor becomes a raw pointer, instead of by hand at the fork sites:
#   1. a tensor handed to a launch on a forked stream is ``record_stream``-ed on it (a no-op for tensors of that stream's own
#      pool): the allocator will not hand its block to anybody before the forked stream has passed the point of the free;
#   2. a forked stream never starts work without first waiting for the stream that owns it (``BranchScope.__enter__``,
#      ``enter_node`` at the head of every autograd backward): blocks of the forked stream's pool are only re-used behind
#      every reader the owning stream had enqueued when they were freed.
# ``tests/test_host_api.py`` checks that no module takes ``.data_ptr()`` behind this file's back.
_FORKED = {}          # raw handle -> (torch.cuda.Stream forked, torch.cuda.Stream owner)
_CUR_FORK = None      # the forked stream launches currently go to (None: an owning / main stream), kept by the scopes below
_CUR_SEEN = None      # addresses already recorded in the innermost scope (a scope re-reads its operands many times)
SINGLE_STREAM = os.environ.get("TAVSR_SINGLE_STREAM", "0") == "1"      # every fork disabled: one queue, as the reference
RULES_OFF = False     # (tests only: switch both rules off to show that the race amplifier then catches the missing dependencies)


def register_fork(forked: "torch.cuda.Stream", owner: "torch.cuda.Stream") -> None:
    _FORKED[forked.cuda_stream] = (forked, owner)


def _note(t) -> None:
    """rule 1 for one tensor (parameters and other step-persistent tensors are never freed inside a step: skipped)"""
    if isinstance(t, torch.nn.Parameter) or RULES_OFF:
        return
    a = t.data_ptr()
    if _CUR_SEEN is not None:
        if a in _CUR_SEEN:
            return
        _CUR_SEEN.add(a)
    t.record_stream(_CUR_FORK)


def push_fork(forked) -> tuple:
    """launches go to ``forked`` from here (ops.BranchScope); returns the state ``pop_fork`` restores"""
    global _CUR_FORK, _CUR_SEEN
    prev = (_CUR_FORK, _CUR_SEEN)
    _CUR_FORK, _CUR_SEEN = forked, set()
    return prev


def pop_fork(prev: tuple) -> None:
    global _CUR_FORK, _CUR_SEEN
    _CUR_FORK, _CUR_SEEN = prev


def enter_node() -> tuple:
    """head of an autograd node's backward: autograd runs the node on the stream its forward ran on.  If that is a forked
    stream and no scope put us there, apply rule 2 (wait for the owner) and switch rule 1 on for the node's launches."""
    global _CUR_FORK, _CUR_SEEN
    prev = (_CUR_FORK, _CUR_SEEN)
    if _FORKED:
        ent = _FORKED.get(_raw_stream(torch.cuda.current_device()) if _raw_stream is not None
                          else torch.cuda.current_stream().cuda_stream)
        if ent is None:
            _CUR_FORK = _CUR_SEEN = None
        elif _CUR_FORK is not ent[0]:
            if not RULES_OFF:
                ent[0].wait_stream(ent[1])
            _CUR_FORK, _CUR_SEEN = ent[0], set()
    return prev


def guarded(fn):
    """decorator of every ``torch.autograd.Function.backward`` of the package (see enter_node)"""
    import functools

    @functools.wraps(fn)
    def wrapper(*args):
        prev = enter_node()
        try:
            return fn(*args)
        finally:
            pop_fork(prev)
    return wrapper


def ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if _CUR_FORK is not None:
        _note(t)
    return C.c_void_p(t.data_ptr())


def addr(t, off: int = 0):
    """``ptr`` as a plain integer (descriptor fields), ``off`` in 4-byte elements; None stays None"""
    if t is None:
        return None
    if _CUR_FORK is not None:
        _note(t)
    return t.data_ptr() + 4 * off


def param_ptrs(P):
    """storage addresses of a parameter list (None -> 0), as one tuple: the signature a cached descriptor template is valid for
    (a ``.to()`` or an optimizer that swaps ``.data`` changes it).  Parameters are not noted on a forked stream (``_note``)."""
    return tuple(0 if p is None else p.data_ptr() for p in P)


def cached_params(module, names):
    """[parameter or None for n in names] of ``module``, looked up once: the Parameter objects of a module never change
    identity (``.to()`` / ``load_state_dict`` / the fused optimizer swap ``.data``), while ``dict(named_parameters())`` per
    forward call costs ~1.5 ms of host time per step over the model's layers."""
    cache = module.__dict__.get("_tavsr_pcache")
    if cache is None or cache[0] is not names:
        sd = dict(module.named_parameters())
        cache = module.__dict__["_tavsr_pcache"] = (names, [sd.get(n) for n in names])
    return cache[1]


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TavsrError("tavsr ops run on the MI355X only: got a CPU tensor (no CPU fallback exists)")
