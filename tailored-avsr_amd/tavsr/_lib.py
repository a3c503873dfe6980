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
        ("rowstat", C.c_void_p),
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


class Lin2Seg(C.Structure):
    """tavsr_lin2_seg (include/tavsr.h)"""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64), ("z", C.c_void_p),
                ("ldz", C.c_int64), ("n", C.c_int32)]


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


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


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
