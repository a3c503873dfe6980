"""Hand-written forward/backward of the hot-path blocks as ``torch.autograd.Function``s.

Each Function is one node of the autograd graph (a whole Branchformer layer, the whole
Conv2dSubsampling, a decoder layer, ...) whose forward and backward are sequences of C-ABI calls
(``tavsr.ops``).  Gradient accumulation at fan-out points, residual adds, activations and their
derivatives are all fused into those kernels (GEMM epilogues, ``dx_add`` of the LayerNorm
backward), so no torch arithmetic kernel runs on the hot path.  The Python here only orders calls
and owns buffers; it is written so that each Function body can move behind a single C entry point
(`tavsr_bf_layer_fwd/bwd`) without changing callers.

Reference semantics followed are cited per Function.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import ops
from ._lib import guarded

EPS_ESPNET = 1e-12  # espnet LayerNorm eps

# Does the node being run have a backward pass?  Set by every Function.forward from ctx.needs_input_grad: a forward under
# no_grad / in eval mode does not write the [M, 2048] pre-activations of its feed-forward and cgMLP blocks (26 MB each).
_NEED_BWD = [True]
_GRAD_MODE = [True]


def grad_apply(fn, *args):
    """``fn.apply(*args)`` that also tells the node whether autograd is recording: ``ctx.needs_input_grad`` only mirrors the
    inputs' ``requires_grad`` and ignores ``torch.no_grad()``, and inside ``Function.forward`` grad mode reads disabled."""
    _GRAD_MODE[0] = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _GRAD_MODE[0] = True


def _note_ctx(ctx):
    """does this node get a backward pass?  (returned, and passed on explicitly by the callers)"""
    nig = getattr(ctx, "needs_input_grad", None)
    need = (True if nig is None else any(nig)) and _GRAD_MODE[0]
    _NEED_BWD[0] = need
    return need


# ------------------------------------------------------------------------------------------------
# building blocks shared by the Functions (plain python, explicit saved state)
# ------------------------------------------------------------------------------------------------
def _drop_(x, p):
    """in-place train-mode dropout; returns the token that regenerates the mask (None when p == 0)."""
    if not p or p <= 0.0:
        return None
    return ops.dropout(x, p, out=x)[1]


def _drop_bwd_(dy, tok):
    """in-place backward of _drop_ (same mask, same scale); no-op without a token."""
    if tok is not None:
        ops.dropout(dy, tok[0], out=dy, token=tok)
    return dy


def _drop_bwd(dy, tok):
    """out-of-place variant for gradients that are still needed unmasked (residual paths)."""
    return dy if tok is None else ops.dropout(dy, tok[0], token=tok)[0]


class _FFN:
    """y = x + scale * drop(W2 drop(act(W1 LN(x) + b1)) + b2)   (encoder_layer.py:192-194,312-314; decoder FFN;
    the inner dropout is PositionwiseFeedForward's, the outer one the layer's: both rate ``p`` in the reference)."""

    @staticmethod
    def fwd(x, ln_w, ln_b, w1, b1, w2, b2, act, scale, eps=EPS_ESPNET, p=0.0, save=True):
        """``save``: keep what the backward needs (the [M, hidden] pre-activations); False for passes without one."""
        if ops.ffn2_usable(x, w1, act):      # streaming chain kernel (csrc/ffn2.hip) + finishing launch; GEMM-path masks
            y, (n, mean, rstd, z, h, t_in, t_out), _, _ = ops.ffn2_fwd(x, ln_w, ln_b, eps, w1, b1, w2, b2, act, scale, p=p,
                                                                       save=save)
            return y, (x, mean, rstd, n, z, h, t_in, t_out)
        n, mean, rstd = ops.layernorm_fwd(x, ln_w, ln_b, eps)
        if save:
            h, z, t_in = ops.linear_drop(n, w1, b1, p, act=act, save_z=True)     # both dropouts ride in the GEMM epilogues
        else:
            (h, t_in), z = ops.linear_drop(n, w1, b1, p, act=act), None
        y, t_out = ops.linear_drop(h, w2, b2, p, alpha=scale, res=x)             # x + scale * dropout(.)
        return y, (x, mean, rstd, n, z, h, t_in, t_out)

    @staticmethod
    def fwd_ln(x, ln_w, ln_b, w1, b1, w2, b2, act, scale, norms, eps=EPS_ESPNET, p=0.0, save=True):
        """``fwd`` plus the LayerNorms (espnet eps) the consumers of y start with: ``norms`` = [(gamma, beta), ...] (at most
        two) -> (y, saved, [n_k], mean, rstd).  On the streaming path they ride in the finishing launch (one statistics
        pass for all of them); otherwise they are the usual LayerNorm launches."""
        if ops.ffn2_usable(x, w1, act):
            y, (n, mean, rstd, z, h, t_in, t_out), outs, (m2, r2) = ops.ffn2_fwd(
                x, ln_w, ln_b, eps, w1, b1, w2, b2, act, scale, p=p, save=save, ln2=norms, ln2_eps=EPS_ESPNET, ln2_stats=save)
            return y, (x, mean, rstd, n, z, h, t_in, t_out), outs, m2, r2
        y, saved = _FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, act, scale, eps=eps, p=p, save=save)
        outs, m2, r2 = [], None, None
        for g, b in norms:
            o, m2, r2 = ops.layernorm_fwd(y, g, b, EPS_ESPNET)
            outs.append(o)
        return y, saved, outs, m2, r2

    @staticmethod
    def bwd(dy, saved, ln_w, w1, w2, act, scale, grp=None, lng=None, chain=True, dyd=None, out_drop=None):
        """returns dx (includes the residual path) and grads (ln_w, ln_b, w1, b1, w2, b2).  ``grp`` (ops.WgradGroup)
        defers the two weight gradients to the caller's grouped launch.  ``chain``: the two activation gradients as one
        streaming launch (ops.ffn2_bwd_dx) instead of two dgrad GEMMs - callers that run two of these blocks side by side
        on two launch queues pass False (a chain kernel owns every CU; two of them serialise, two GEMM sequences overlap)."""
        x, mean, rstd, n, z, h, t_in, t_out = saved
        wgrad = ops.linear_dw if grp is None else grp.add
        if dyd is None:              # (callers whose producer of dy is a LayerNorm backward get the masked copy from that launch)
            dyd = _drop_bwd(dy, t_out)
        gw2, gb2 = wgrad(dyd, h, alpha=scale, bias_grad=True)
        stream2 = chain and ops.FFN2_BWD and ops.ffn2_shape_ok(dyd, w1, act) and z.is_contiguous()
        if stream2:
            # the block's own LayerNorm backward is dn's only reader: it sums the launch's partials itself (no finishing launch)
            slab_ok = (ops.FFN2_BWD_LN and lng is not None and lng.takes(*x.shape) and (out_drop is None or ops.LN_BWD_DROP)
                       and x.is_contiguous())
            dz, dn = ops.ffn2_bwd_dx(dyd, scale, w1, w2, z, act, t_in, sum_dn=not slab_ok)
        else:
            dz = ops.linear_dx_drop(dyd, w2, t_in, alpha=scale, DZ=z, dact=act)    # inner mask and act'(z) in the epilogue
        gw1, gb1 = wgrad(dz, n, bias_grad=True)
        if not stream2:
            dn = ops.linear_dx(dz, w1)
        if out_drop is not None and lng is not None:       # + dx under the NEXT block's outer mask, from the same launch
            dx, gln_w, gln_b, dxd = lng.bwd(dn, x, mean, rstd, ln_w, dx_add=dy, drop=out_drop)
            return dx, (gln_w, gln_b, gw1, gb1, gw2, gb2), dxd
        ln_bwd = ops.layernorm_bwd if lng is None else lng.bwd      # lng: the node's shared (dgamma, dbeta) reduction
        dx, gln_w, gln_b = ln_bwd(dn, x, mean, rstd, ln_w, dx_add=dy)
        return dx, (gln_w, gln_b, gw1, gb1, gw2, gb2)


_POS_DW_BESIDE = os.environ.get("TAVSR_POS_DW_BESIDE", "1") != "0"


class _AttnFused:
    """Attention core on the fused kernels (ops.attn_fwd / attn_bwd): scores, rel_shift, mask, softmax, dropout and the
    context product in one launch; only the per-row log-sum-exp is kept for the backward, which recomputes the
    probabilities.  q / k / v are 2-D row buffers with element offsets of their column windows (as _SelfAttnCore)."""

    @staticmethod
    def fwd(q, q_off, kbuf, k_off, vbuf, v_off, B, T1, T2, H, dk, klens, causal, pos=None, bias_u=None, bias_v=None,
            p_att=0.0):
        ctx, lse, tok = ops.attn_fwd(q, q_off, kbuf, k_off, vbuf, v_off, B, T1, T2, H, dk, klens=klens, causal=causal,
                                     pos=pos, bias_u=bias_u, bias_v=bias_v, p_drop=p_att)
        return ctx, (lse, tok)

    @staticmethod
    def bwd(dctx, ctx, saved, q, q_off, kbuf, k_off, vbuf, v_off, dq, dq_off, dk_buf, dk_off, dv_buf, dv_off, B, T1, T2, H,
            dk, klens, causal, pos=None, bias_u=None, bias_v=None, lazy_dp=False):
        """writes d/d(q+u) into dq, dK, dV into their windows; rel-pos: returns (dqv, dp) with dp the gradient of the
        projected positional rows [2*T1-1, H*dk].  ``lazy_dp``: dp comes back as a function that computes it (two launches whose only
        reader is linear_pos's weight gradient: the caller may run them with its other weight gradients, off the backward chain)."""
        lse, tok = saved
        dqv, sk = ops.attn_bwd(dctx, ctx, lse, tok, q, q_off, kbuf, k_off, vbuf, v_off, B, T1, T2, H, dk, dq, dq_off,
                               dk_buf, dk_off, dv_buf, dv_off, klens=klens, causal=causal, pos=pos, bias_u=bias_u,
                               bias_v=bias_v)
        if pos is None:
            return None, None
        D = H * dk
        W, Wp = 2 * T1 - 1, sk.shape[-1]

        def make_dp():
            # dP[:,h] = sum_b ds_skew[h,b]^T (q + v)[b,:,h]  == one K = B*T1 GEMM per head
            _, qv = ops.add_head_bias(q[:, q_off: q_off + D], bias_u, bias_v)
            dp = ops.empty(W, D, like=dctx)
            ops.gemm(W, dk, B * T1, sk, Wp, qv, D, dp, D, a_kmajor=True, b_kmajor=True, nb1=H, sA=(B * T1 * Wp, 0),
                     sB=(dk, 0), sC=(dk, 0))
            return dp
        return dqv, (make_dp if lazy_dp else make_dp())


class _SelfAttnCore:
    """Scores/softmax/context of one attention call on head-strided buffers.

    q rows live in ``qbuf`` (row stride ldq, element offset q_off), k/v likewise; outputs go to
    ``ctx`` [B*T1, D].  rel-pos (espnet RelPositionMultiHeadedAttention) when ``p`` is given."""

    @staticmethod
    def fwd(qu, ldq, q_off, kbuf, ldk, k_off, vbuf, ldv, v_off, B, T1, T2, H, dk, klens, causal, qv=None, p=None,
            p_att=0.0):
        D = H * dk
        dev = qu
        S = ops.pad4(T2)   # padded score-row stride: 16-byte loads in the GEMMs that read the scores
        ac = ops.empty(H, B, T1, S, like=dev)
        # ac[h,b] = Qu[b,:,h] K[b,:,h]^T
        ops.gemm(T1, T2, dk, qu, ldq, kbuf, ldk, ac, S, a_off=q_off, b_off=k_off, nb1=B, nb2=H,
                 sA=(T1 * ldq, dk), sB=(T2 * ldk, dk), sC=(T1 * S, B * T1 * S))
        bd = None
        W = 0
        if p is not None:
            W = 2 * T1 - 1
            Wp = ops.pad4(W)
            bd = ops.empty(H, B, T1, Wp, like=dev)
            ops.gemm(T1, W, dk, qv, D, p, D, bd, Wp, nb1=B, nb2=H, sA=(T1 * D, dk), sB=(0, dk),
                     sC=(T1 * Wp, B * T1 * Wp))
        if p_att and p_att > 0.0:      # dropout on the probabilities (espnet forward_attention) by the softmax launch itself;
            attn, pv, tok = ops.softmax_fwd(ac, bd, klens, 1.0 / math.sqrt(dk), causal, T2=T2, W=W, p_drop=p_att)
            tok = (tok, pv)            # attn is kept; the dropped probabilities stay resident for dV (5 MB per layer)
        else:
            attn = ops.softmax_fwd(ac, bd, klens, 1.0 / math.sqrt(dk), causal, T2=T2, W=W)
            pv, tok = attn, None
        ctx = ops.empty(B * T1, D, like=dev)
        # ctx[b,:,h] = drop(attn)[h,b] V[b,:,h]
        ops.gemm(T1, dk, T2, pv, S, vbuf, ldv, ctx, D, b_off=v_off, b_kmajor=True, nb1=B, nb2=H,
                 sA=(T1 * S, B * T1 * S), sB=(T2 * ldv, dk), sC=(T1 * D, dk))
        return ctx, attn, tok

    @staticmethod
    def bwd(dctx, attn, qu, ldq, q_off, kbuf, ldk, k_off, vbuf, ldv, v_off, dq, lddq, dq_off, dk_buf, lddk, dk_off,
            dv_buf, lddv, dv_off, B, T1, T2, H, dk, qv=None, p=None, tok=None):
        """Writes dQ(u) into dq, dK into dk_buf, dV into dv_buf (head-strided); returns (dqv, dp) for rel-pos."""
        D = H * dk
        S = attn.shape[-1]
        sS = (T1 * S, B * T1 * S)
        dattn = torch.empty_like(attn)
        # dattn[h,b] = dctx[b,:,h] V[b,:,h]^T
        ops.gemm(T1, T2, dk, dctx, D, vbuf, ldv, dattn, S, b_off=v_off, nb1=B, nb2=H, sA=(T1 * D, dk),
                 sB=(T2 * ldv, dk), sC=sS)
        # dV[b,:,h] = drop(attn)[h,b]^T dctx[b,:,h]
        pv = attn if tok is None else tok[1]
        ops.gemm(T2, dk, T1, pv, S, dctx, D, dv_buf, lddv, c_off=dv_off, a_kmajor=True, b_kmajor=True, nb1=B, nb2=H,
                 sA=sS, sB=(T1 * D, dk), sC=(T2 * lddv, dk))
        del pv
        # dattn is the gradient of the dropped probabilities: the softmax backward regenerates the mask itself
        ds, sk = ops.softmax_bwd(attn, dattn, 1.0 / math.sqrt(dk), skew=p is not None, T2=T2,
                                 token=None if tok is None else tok[0])
        # dQu[b,:,h] = ds[h,b] K[b,:,h]
        ops.gemm(T1, dk, T2, ds, S, kbuf, ldk, dq, lddq, b_off=k_off, c_off=dq_off, b_kmajor=True, nb1=B, nb2=H,
                 sA=sS, sB=(T2 * ldk, dk), sC=(T1 * lddq, dk))
        # dK[b,:,h] = ds[h,b]^T Qu[b,:,h]
        ops.gemm(T2, dk, T1, ds, S, qu, ldq, dk_buf, lddk, b_off=q_off, c_off=dk_off, a_kmajor=True, b_kmajor=True,
                 nb1=B, nb2=H, sA=sS, sB=(T1 * ldq, dk), sC=(T2 * lddk, dk))
        if p is None:
            return None, None
        W = 2 * T1 - 1
        Wp = sk.shape[-1]
        dqv = ops.empty(B * T1, D, like=dctx)
        # dQv[b,:,h] = ds_skew[h,b] P[:,h]
        ops.gemm(T1, dk, W, sk, Wp, p, D, dqv, D, b_kmajor=True, nb1=B, nb2=H, sA=(T1 * Wp, B * T1 * Wp), sB=(0, dk),
                 sC=(T1 * D, dk))
        # dP[:,h] = sum_b ds_skew[h,b]^T Qv[b,:,h]  == one K = B*T1 GEMM per head
        dp = ops.empty(W, D, like=dctx)
        ops.gemm(W, dk, B * T1, sk, Wp, qv, D, dp, D, a_kmajor=True, b_kmajor=True, nb1=H, sA=(B * T1 * Wp, 0),
                 sB=(dk, 0), sC=(dk, 0))
        return dqv, dp


# ------------------------------------------------------------------------------------------------
# Branchformer encoder layer
# ------------------------------------------------------------------------------------------------
# parameter order of BranchformerLayerFn (None entries allowed for absent branches)
BF_PARAM_NAMES = (
    "norm_ff_macaron.weight", "norm_ff_macaron.bias",
    "feed_forward_macaron.w_1.weight", "feed_forward_macaron.w_1.bias",
    "feed_forward_macaron.w_2.weight", "feed_forward_macaron.w_2.bias",
    "norm_mha.weight", "norm_mha.bias",
    "attn.linear_q.weight", "attn.linear_q.bias", "attn.linear_k.weight", "attn.linear_k.bias",
    "attn.linear_v.weight", "attn.linear_v.bias", "attn.linear_out.weight", "attn.linear_out.bias",
    "attn.linear_pos.weight", "attn.pos_bias_u", "attn.pos_bias_v",
    "norm_mlp.weight", "norm_mlp.bias",
    "cgmlp.channel_proj1.0.weight", "cgmlp.channel_proj1.0.bias",
    "cgmlp.csgu.norm.weight", "cgmlp.csgu.norm.bias", "cgmlp.csgu.conv.weight", "cgmlp.csgu.conv.bias",
    "cgmlp.channel_proj2.weight", "cgmlp.channel_proj2.bias",
    "pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias", "pooling_proj2.bias",
    "weight_proj1.weight", "weight_proj2.weight", "weight_proj1.bias", "weight_proj2.bias",
    "merge_proj.weight", "merge_proj.bias",
    "norm_ff.weight", "norm_ff.bias",
    "feed_forward.w_1.weight", "feed_forward.w_1.bias", "feed_forward.w_2.weight", "feed_forward.w_2.bias",
    "norm_final.weight", "norm_final.bias",
)
_I = {n: i for i, n in enumerate(BF_PARAM_NAMES)}


_LAYER_WS = {}


def _layer_c_ok(x, cfg, P, pd, pos_emb=None) -> bool:
    """the recipe form csrc/layer.hip sequences (both branches, learned-average merge + merge_proj); the shapes are the
    library's to judge (tavsr_branchformer_layer_ok).  Un-captured loops only unless TAVSR_LAYER_C=capture: a captured step
    replays the same kernels either way, and the capture of the Python sequencing measured 1 % faster (allocation order)."""
    if not ops.LAYER_C or ops.PROFILE is not None or (ops.LAYER_C_EAGER_ONLY and torch.cuda.is_current_stream_capturing()):
        return False
    if not (cfg["has_attn"] and cfg["has_mlp"] and cfg["merge"] == "learned_ave" and not cfg["merge_identity"]):
        return False
    B, T, D = x.shape
    cw, w1, w1m = P[_I["cgmlp.csgu.conv.weight"]], P[_I["feed_forward.w_1.weight"]], P[_I["feed_forward_macaron.w_1.weight"]]
    return (x.is_contiguous() and (pos_emb is None or pos_emb.is_contiguous()) and w1.shape == w1m.shape
            and bool(ops.lib().tavsr_branchformer_layer_ok(B, T, D, cfg["heads"], w1.shape[0], 2 * cw.shape[0], cw.shape[-1])))


_DESC_PARAMS = (("ffm_ln_w", "norm_ff_macaron.weight"), ("ffm_ln_b", "norm_ff_macaron.bias"),
                ("ffm_w1", "feed_forward_macaron.w_1.weight"), ("ffm_b1", "feed_forward_macaron.w_1.bias"),
                ("ffm_w2", "feed_forward_macaron.w_2.weight"), ("ffm_b2", "feed_forward_macaron.w_2.bias"),
                ("mha_ln_w", "norm_mha.weight"), ("mha_ln_b", "norm_mha.bias"),
                ("wq", "attn.linear_q.weight"), ("bq", "attn.linear_q.bias"), ("wk", "attn.linear_k.weight"),
                ("bk", "attn.linear_k.bias"), ("wv", "attn.linear_v.weight"), ("bv", "attn.linear_v.bias"),
                ("wpos", "attn.linear_pos.weight"), ("pos_u", "attn.pos_bias_u"), ("pos_v", "attn.pos_bias_v"),
                ("wo", "attn.linear_out.weight"), ("bo", "attn.linear_out.bias"),
                ("mlp_ln_w", "norm_mlp.weight"), ("mlp_ln_b", "norm_mlp.bias"),
                ("cg_w1", "cgmlp.channel_proj1.0.weight"), ("cg_b1", "cgmlp.channel_proj1.0.bias"),
                ("csgu_ln_w", "cgmlp.csgu.norm.weight"), ("csgu_ln_b", "cgmlp.csgu.norm.bias"),
                ("csgu_cw", "cgmlp.csgu.conv.weight"), ("csgu_cb", "cgmlp.csgu.conv.bias"),
                ("cg_w2", "cgmlp.channel_proj2.weight"), ("cg_b2", "cgmlp.channel_proj2.bias"),
                ("merge_w", "merge_proj.weight"), ("merge_b", "merge_proj.bias"),
                ("ff_ln_w", "norm_ff.weight"), ("ff_ln_b", "norm_ff.bias"),
                ("ff_w1", "feed_forward.w_1.weight"), ("ff_b1", "feed_forward.w_1.bias"),
                ("ff_w2", "feed_forward.w_2.weight"), ("ff_b2", "feed_forward.w_2.bias"),
                ("final_ln_w", "norm_final.weight"), ("final_ln_b", "norm_final.bias"))
_DESC_MERGE = ("pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias", "pooling_proj2.bias",
               "weight_proj1.weight", "weight_proj2.weight", "weight_proj1.bias", "weight_proj2.bias")
_DESC_TMPL = {}       # id(parameter list of a layer) -> (storage addresses, descriptor bytes with the parameter fields filled)
_LAYER_LAYOUT = {}    # shape key -> ({buffer: (offset, shape)}, floats): the kept state of one layer as ONE allocation


def _layer_layout(B, T, D, H, N1, C2, need):
    """Every buffer a C-side layer forward writes besides its output, as offsets (256-byte aligned) into one allocation:
    fifty allocator calls and tensor objects per layer were a third of an un-captured step's host time."""
    key = (B, T, D, H, N1, C2, need)
    lay = _LAYER_LAYOUT.get(key)
    if lay is None:
        M, W, Cn = B * T, 2 * T - 1, C2 // 2
        Mp = (M + 127) // 128 * 128
        items = [(k, (M, D)) for k in ("x1", "n_mha", "n_mlp", "cx", "xa", "xm", "m", "x2", "x3")]
        items += [("qkv", (M, 3 * D)), ("pp", (W, D)), ("lse", (B * H, T)), ("g", (M, C2)), ("u", (M, Cn)), ("g_mean", (M,)),
                  ("g_rstd", (M,)), ("score", (2, B, T)), ("pooled", (4, M))]
        rows = {}
        if need:
            items += [("ffm_n", (M, D)), ("ff_n", (M, D))]
            items += [(k, (Mp, N1)) for k in ("ffm_z", "ffm_h", "ff_z", "ff_h")]      # (the streaming kernel writes whole row blocks)
            rows = {k: M for k in ("ffm_z", "ffm_h", "ff_z", "ff_h")}
            items += [("g_z", (M, C2)), ("gn", (M, Cn)), ("conv", (M, Cn))]
            items += [(k, (M,)) for k in ("ffm_mean", "ffm_rstd", "br_mean", "br_rstd", "ff_mean", "ff_rstd", "fin_mean", "fin_rstd")]
        off, o = {}, 0
        for k, shp in items:
            n = 1
            for v in shp:
                n *= v
            off[k] = (o, (rows[k],) + tuple(shp[1:]) if k in rows else tuple(shp))
            o += (n + 63) // 64 * 64
        lay = _LAYER_LAYOUT[key] = (off, o)
    return lay


def _layer_sv(flat, off, x2d, toks, wts):
    """the kept state as the tensors the Python sequencing keeps (views of the one allocation)"""
    def g(k):
        e = off.get(k)
        if e is None:
            return None
        o, shp = e
        n = 1
        for v in shp:
            n *= v
        return flat[o: o + n].view(shp)
    return {"ffm": (x2d, g("ffm_mean"), g("ffm_rstd"), g("ffm_n"), g("ffm_z"), g("ffm_h"), toks[0], toks[1]),
            "attn": (g("br_mean"), g("br_rstd"), g("n_mha"), g("qkv"), g("pp"), None, None, g("cx"), (g("lse"), toks[2]), None, toks[3]),
            "mlp": (g("br_mean"), g("br_rstd"), g("n_mlp"), g("g"), g("g_z"), g("gn"), g("g_mean"), g("g_rstd"), g("u"), g("conv"),
                    toks[4], toks[5]),
            "merge": (g("score"), g("pooled"), wts, g("m")),
            "drop": (None, toks[6]),
            "ff": (g("x2"), g("ff_mean"), g("ff_rstd"), g("ff_n"), g("ff_z"), g("ff_h"), toks[7], toks[8]),
            "final": (g("x3"), g("fin_mean"), g("fin_rstd")),
            "x1": g("x1"), "xa": g("xa"), "xm": g("xm")}


class _LazySV:
    """``ctx.sv`` of a C-side forward: the tensors of the Python backward are only built if that path runs"""
    __slots__ = ("flat", "off", "x2d", "toks", "wts", "_sv")

    def __init__(self, flat, off, x2d, toks, wts):
        self.flat, self.off, self.x2d, self.toks, self.wts, self._sv = flat, off, x2d, toks, wts, None

    def __getitem__(self, k):
        if self._sv is None:
            self._sv = _layer_sv(self.flat, self.off, self.x2d, self.toks, self.wts)
        return self._sv[k]


def _layer_c_forward(ctx, x, pos_emb, lens, cfg, P, need):
    """BranchformerLayerFn.forward as one C call (tavsr_branchformer_layer_fwd).  Host side of an un-captured step: the
    parameter fields of the descriptor come from a per-layer template (rebuilt when a parameter's storage moves), everything the
    call writes besides its output is ONE allocation (``_layer_layout``), and the tensors the Python backward would read are
    views built only if that path runs (``_LazySV``)."""
    from ._lib import BfLayerDesc, check, lib, param_ptrs
    import ctypes as C
    B, T, D = x.shape
    M, H = B * T, cfg["heads"]
    N1 = P[_I["feed_forward.w_1.weight"]].shape[0]
    C2 = P[_I["cgmlp.channel_proj1.0.weight"]].shape[0]
    Cn = C2 // 2
    pd, pa = cfg.get("p", 0.0), cfg.get("p_att", 0.0)
    x2d = x.view(M, D)
    sig = param_ptrs(P)
    tm = _DESC_TMPL.get(id(P))
    if tm is None or tm[0] != sig:
        t = BfLayerDesc()
        for f, n in _DESC_PARAMS:
            setattr(t, f, sig[_I[n]])
        for j, n in enumerate(_DESC_MERGE):
            t.merge_p[j] = sig[_I[n]]
        tm = _DESC_TMPL[id(P)] = (sig, bytes(t))
    d = BfLayerDesc.from_buffer_copy(tm[1])
    d.B, d.T, d.D, d.H, d.ffn_units, d.cg_units, d.cg_kernel = B, T, D, H, N1, C2, 31
    d.ffn_act, d.save = ops.ACT[cfg["ffn_act"]], int(need)
    d.p_drop, d.p_att, d.coeff = pd, pa, cfg.get("coeff", 1.0)
    d.x, d.pos_emb, d.lens = ops._addr(x2d), ops._addr(pos_emb), ops._addr(lens)
    # dropout tokens in the order the Python sequencing draws them (same masks either way)
    toks = [None] * 9
    if pd > 0.0 or pa > 0.0:
        T4 = ops.pad4(T)
        sizes = (M * N1, M * D, B * H * T * T4, M * D, M * Cn, M * D, M * D, M * N1, M * D)
        rates = (pd, pd, pa, pd, pd, pd, pd, pd, pd)
        toks = [ops._new_token(r, n, x.device) if r and r > 0.0 else None for r, n in zip(rates, sizes)]
        for j, t in enumerate(toks):
            if t is not None:
                d.drop_off[j] = t[1]
                d.seed = ops._addr(t[2])
    off, nfl = _layer_layout(B, T, D, H, N1, C2, bool(need))
    flat = ops.empty(nfl, like=x)
    y = ops.empty(M, D, like=x)
    wts = ops.empty(B, 2, like=x)      # (its own block: the module keeps it as weight_global / weight_local beyond the step)
    base = ops._addr(flat)
    for k, (o, _) in off.items():
        setattr(d, k, base + 4 * o)
    d.y, d.wts = ops._addr(y), ops._addr(wts)
    main = torch.cuda.current_stream()
    side = ops.branch_stream(main) if ops.forks_enabled() else main      # (one queue: the fork / join events order nothing new)
    ev = ops.branch_events(main)
    d.stream2, d.ev_fork, d.ev_join = side.cuda_stream, ev[0].cuda_event, ev[1].cuda_event
    key = (B, T, D, H, N1, C2, int(need), pd > 0.0, pa > 0.0)
    nws = _LAYER_WS.get(key)
    if nws is None:
        fn = lib().tavsr_branchformer_layer_ws
        fn.restype = C.c_int64
        nws = _LAYER_WS[key] = int(fn(C.byref(d)))
    ws = ops.empty(max(nws, 4), like=x)
    d.ws, d.ws_floats = ops._addr(ws), nws
    check(lib().tavsr_branchformer_layer_fwd(C.byref(d), C.c_void_p(main.cuda_stream)), "tavsr_branchformer_layer_fwd")
    ctx.sv, ctx.cfg, ctx.P, ctx.lens, ctx.pos_emb = _LazySV(flat, off, x2d, toks, wts), cfg, P, lens, pos_emb
    ctx.shape = (B, T, D)
    if need:
        ctx.cdesc = d      # the descriptor (raw addresses of the parameters and of every kept buffer; ctx.sv / ctx.P hold the tensors)
    # (ws goes back to the allocator here: every launch that reads it is enqueued, the side queue has been joined into the calling
    # one inside the call, and the block can only be handed to later work of the calling queue)
    cfg["_last_w"] = wts
    return y.view(B, T, D)


# gradient slot of tavsr_bf_layer_bwd_desc -> parameter (BF_PARAM_NAMES); the five d_model LayerNorms come back in one buffer
_BWD_FIELDS = (
    ("g_ffm_w1", "feed_forward_macaron.w_1.weight"), ("g_ffm_b1", "feed_forward_macaron.w_1.bias"),
    ("g_ffm_w2", "feed_forward_macaron.w_2.weight"), ("g_ffm_b2", "feed_forward_macaron.w_2.bias"),
    ("g_wq", "attn.linear_q.weight"), ("g_bq", "attn.linear_q.bias"), ("g_wk", "attn.linear_k.weight"), ("g_bk", "attn.linear_k.bias"),
    ("g_wv", "attn.linear_v.weight"), ("g_bv", "attn.linear_v.bias"), ("g_wo", "attn.linear_out.weight"), ("g_bo", "attn.linear_out.bias"),
    ("g_wpos", "attn.linear_pos.weight"), ("g_pos_u", "attn.pos_bias_u"), ("g_pos_v", "attn.pos_bias_v"),
    ("g_cg_w1", "cgmlp.channel_proj1.0.weight"), ("g_cg_b1", "cgmlp.channel_proj1.0.bias"),
    ("g_csgu_ln_w", "cgmlp.csgu.norm.weight"), ("g_csgu_ln_b", "cgmlp.csgu.norm.bias"),
    ("g_csgu_cw", "cgmlp.csgu.conv.weight"), ("g_csgu_cb", "cgmlp.csgu.conv.bias"),
    ("g_cg_w2", "cgmlp.channel_proj2.weight"), ("g_cg_b2", "cgmlp.channel_proj2.bias"),
    ("g_merge_w", "merge_proj.weight"), ("g_merge_b", "merge_proj.bias"),
    ("g_ff_w1", "feed_forward.w_1.weight"), ("g_ff_b1", "feed_forward.w_1.bias"),
    ("g_ff_w2", "feed_forward.w_2.weight"), ("g_ff_b2", "feed_forward.w_2.bias"))
_BWD_MERGE = ("pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias", "pooling_proj2.bias",
              "weight_proj1.weight", "weight_proj2.weight", "weight_proj1.bias", "weight_proj2.bias")
_BWD_NORMS = ("norm_final", "norm_ff", "norm_mlp", "norm_mha", "norm_ff_macaron")


_GRAD_LAYOUT = {}     # id(parameter list) -> (shapes, [(slot, field or merge index, offset, numel, shape)], floats)


def _grad_layout(P, D):
    """the layer's parameter gradients as views of ONE allocation (256-byte aligned offsets); the five d_model LayerNorms'
    (dgamma, dbeta) pairs are one contiguous run, as tavsr_branchformer_layer_bwd writes them"""
    shapes = tuple(None if p is None else tuple(p.shape) for p in P)
    lay = _GRAD_LAYOUT.get(id(P))
    if lay is None or lay[0] != shapes:
        ent, o = [], 0
        for f, n in _BWD_FIELDS:
            shp = shapes[_I[n]]
            k = 1
            for v in shp:
                k *= v
            ent.append((_I[n], f, o, k, shp))
            o += (k + 63) // 64 * 64
        for j, n in enumerate(_BWD_MERGE):
            shp = shapes[_I[n]]
            k = 1
            for v in shp:
                k *= v
            ent.append((_I[n], j, o, k, shp))
            o += (k + 63) // 64 * 64
        ln0 = o
        for j, n in enumerate(_BWD_NORMS):
            ent.append((_I[n + ".weight"], None, o, D, (D,)))
            ent.append((_I[n + ".bias"], None, o + D, D, (D,)))
            o += 2 * D
        lay = _GRAD_LAYOUT[id(P)] = (shapes, ent, ln0, o)
    return lay


def _layer_c_backward(ctx, dy):
    """BranchformerLayerFn.backward as one C call (tavsr_branchformer_layer_bwd) on the state a C forward call left: allocates
    the gradients (one block, handed out as views), fills the descriptor; bit-identical to the Python sequencing below."""
    from ._lib import BfLayerBwdDesc, check, lib
    import ctypes as C
    d, P = ctx.cdesc, ctx.P
    B, T, D = ctx.shape
    M = B * T
    dy2 = dy.contiguous().view(M, D)
    main = torch.cuda.current_stream()
    side = ops.branch_stream(main) if ops.forks_enabled() else main
    ev = ops.branch_events(main)
    d.stream2, d.ev_fork, d.ev_join = side.cuda_stream, ev[0].cuda_event, ev[1].cuda_event
    b = BfLayerBwdDesc()
    b.fwd = C.pointer(d)
    _, ent, ln0, nfl = _grad_layout(P, D)
    gflat = ops.empty(nfl, like=dy2)
    base = ops._addr(gflat)
    G: List[Optional[torch.Tensor]] = [None] * len(BF_PARAM_NAMES)
    for slot, f, o, k, shp in ent:
        G[slot] = gflat[o: o + k].view(shp)
        if f is None:
            continue
        if isinstance(f, int):
            b.g_merge_p[f] = base + 4 * o
        else:
            setattr(b, f, base + 4 * o)
    dx = ops.empty(M, D, like=dy2)
    b.dy, b.dx, b.g_ln = ops._addr(dy2), ops._addr(dx), base + 4 * ln0
    key = ("bwd", B, T, D, d.H, d.ffn_units, d.cg_units, d.p_drop > 0.0)
    nws = _LAYER_WS.get(key)
    if nws is None:
        fn = lib().tavsr_branchformer_layer_bwd_ws
        fn.restype = C.c_int64
        nws = _LAYER_WS[key] = int(fn(C.byref(b)))
    ws = ops.empty(max(nws, 4), like=dy2)
    b.ws, b.ws_floats = ops._addr(ws), nws
    beside = side is not main and ops.WGRAD_SLOT == 0 and ops.wgrad_may_go_beside(P) and ops.wgrad_open(main, side)
    b.wgrad_beside = 1 if beside else 0
    check(lib().tavsr_branchformer_layer_bwd(C.byref(b), C.c_void_p(main.cuda_stream)), "tavsr_branchformer_layer_bwd")
    if beside:      # what those launches read and write is freed on THIS stream (rule 1 of _lib.py, by hand: the addresses were taken here)
        sv = ctx.sv
        for t in (ws, gflat, dy2, sv.flat, sv.x2d, sv.wts, ctx.pos_emb):
            if torch.is_tensor(t) and not isinstance(t, torch.nn.Parameter):
                t.record_stream(side)
    ctx.sv = ctx.cdesc = None
    return (dx.view(B, T, D), None, None, None, *G)


class BranchformerLayerFn(torch.autograd.Function):
    """``MyBranchformerEncoderLayer.forward`` (src/encoder/branchformer/encoder_layer.py:153-321) with
    dropout / stochastic depth disabled (rate 0 or eval); ``coeff`` is the stochastic-depth scale."""

    @staticmethod
    def forward(ctx, x, pos_emb, lens, cfg, *P):
        need = _note_ctx(ctx)
        B, T, D = x.shape
        M = B * T
        H = cfg["heads"]
        dk = D // H
        act = cfg["ffn_act"]
        merge = cfg["merge"]  # learned_ave | fixed_ave | concat | attn_only | mlp_only (+ identity flag)
        has_attn, has_mlp = cfg["has_attn"], cfg["has_mlp"]
        coeff = cfg.get("coeff", 1.0)
        pd, pa = cfg.get("p", 0.0), cfg.get("p_att", 0.0)     # dropout rates (0 in eval)
        p = lambda n: P[_I[n]]
        if _layer_c_ok(x, cfg, P, pd, pos_emb):
            return _layer_c_forward(ctx, x, pos_emb, lens, cfg, P, need)
        x2d = x.reshape(M, D)
        sv = {}

        # the macaron block's finishing launch also normalises its output for the two branches (one statistics pass)
        br_norms = ([(p("norm_mha.weight"), p("norm_mha.bias"))] if has_attn else []) + \
                   ([(p("norm_mlp.weight"), p("norm_mlp.bias"))] if has_mlp else [])
        x1, sv["ffm"], nbr, bmean, brstd = _FFN.fwd_ln(x2d, p("norm_ff_macaron.weight"), p("norm_ff_macaron.bias"),
                                                       p("feed_forward_macaron.w_1.weight"), p("feed_forward_macaron.w_1.bias"),
                                                       p("feed_forward_macaron.w_2.weight"), p("feed_forward_macaron.w_2.bias"),
                                                       act, 0.5, br_norms, p=pd, save=need)
        two = has_attn and has_mlp
        cat = ops.empty(M, 2 * D, like=x) if merge == "concat" else None
        xa = xm = None
        mp = rd = None
        use_rd = False
        if two and merge == "learned_ave":
            mp = [p(k) for k in ("pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias",
                                 "pooling_proj2.bias", "weight_proj1.weight", "weight_proj2.weight",
                                 "weight_proj1.bias", "weight_proj2.bias")]
            rd = [None, None]
            # the fused tail takes the merge's row dots from the epilogues of the launches that store the branch outputs
            use_rd = (ops.MERGE_ROWDOT and not cfg["merge_identity"] and ops.MERGE_PROJ and ops.MERGE_ROWS
                      and bool(ops.lib().tavsr_merge_proj_ok(T, D)))
        br = ops.BranchScope(two)     # attention branch beside the cgMLP branch (joined before the merge)
        with br:
            if has_attn:
                n, mean, rstd = nbr[0], bmean, brstd
                qkv = ops.empty(M, 3 * D, like=x)
                ops.linear_group(n, [(p(f"attn.linear_{c}.weight"), p(f"attn.linear_{c}.bias"), j * D)
                                     for j, c in enumerate("qkv")], qkv)
                pe2d = pos_emb.reshape(-1, D)
                pp = ops.linear(pe2d, p("attn.linear_pos.weight"))
                if ops.ATTN_FUSED and dk == 64:
                    qu = qv = t_att = None
                    cx, attn = _AttnFused.fwd(qkv, 0, qkv, D, qkv, 2 * D, B, T, T, H, dk, lens, False, pos=pp,
                                              bias_u=p("attn.pos_bias_u").reshape(-1), bias_v=p("attn.pos_bias_v").reshape(-1),
                                              p_att=pa)
                else:
                    qu, qv = ops.add_head_bias(qkv[:, :D], p("attn.pos_bias_u").reshape(-1), p("attn.pos_bias_v").reshape(-1))
                    cx, attn, t_att = _SelfAttnCore.fwd(qu, D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, B, T, T, H, dk, lens, False,
                                                        qv=qv, p=pp, p_att=pa)
                t_xa = None
                if merge == "concat":
                    ops.linear(cx, p("attn.linear_out.weight"), p("attn.linear_out.bias"), out=cat, out_off=0, ldc=2 * D)
                    xa = cat[:, :D]
                else:
                    wo_ = p("attn.linear_out.weight")
                    if use_rd and ops.rowdot_ok(cx, wo_):     # ... and the merge's row dots of x1 from the same launch's epilogue
                        xa, t_xa, rd[0] = ops.linear_drop(cx, wo_, p("attn.linear_out.bias"), pd, rowdot=(mp[0], mp[4]))
                    else:
                        xa, t_xa = ops.linear_drop(cx, wo_, p("attn.linear_out.bias"), pd)   # x1 = dropout(x_att)  (encoder_layer.py:212)
                sv["attn"] = (mean, rstd, n, qkv, pp, qu, qv, cx, attn, t_att, t_xa)
        if has_mlp:
            n, mean, rstd = nbr[-1], bmean, brstd
            w1c = p("cgmlp.channel_proj1.0.weight")
            cw = p("cgmlp.csgu.conv.weight")
            # channel_proj1's epilogue leaves the CSGU's LayerNorm statistics as per-tile row sums (no statistics launch)
            rst = (ops.empty(M, w1c.shape[0] // 64, 2, like=x)
                   if (ops.CSGU_FUSED and cw.shape[-1] == 31 and ops.csgu_rowstat_ok(n, w1c)) else None)
            if need:
                g, z = ops.linear(n, w1c, p("cgmlp.channel_proj1.0.bias"), act="gelu", save_z=True, rowstat=rst)
            else:
                g, z = ops.linear(n, w1c, p("cgmlp.channel_proj1.0.bias"), act="gelu", rowstat=rst), None
            Cn = g.shape[1] // 2
            if ops.csgu_usable(g, cw):       # LayerNorm + depthwise convolution + gate + dropout: one pass over g
                u, conv, gn, gmean, grstd, t_u = ops.csgu_fwd(g, p("cgmlp.csgu.norm.weight"), p("cgmlp.csgu.norm.bias"), EPS_ESPNET,
                                                              cw.reshape(Cn, -1), p("cgmlp.csgu.conv.bias"), B, T, p=pd, save=need,
                                                              rowstat=rst)
            else:
                gn, gmean, grstd = ops.layernorm_fwd(g[:, Cn:], p("cgmlp.csgu.norm.weight"), p("cgmlp.csgu.norm.bias"),
                                                     EPS_ESPNET)
                u, conv = ops.dwconv_gate_fwd(gn, g[:, :Cn], cw.reshape(Cn, -1), p("cgmlp.csgu.conv.bias"), B, T)
                t_u = _drop_(u, pd)                             # csgu: dropout(x_r * x_g)
            t_xm = None
            if merge == "concat":
                ops.linear(u, p("cgmlp.channel_proj2.weight"), p("cgmlp.channel_proj2.bias"), out=cat, out_off=D,
                           ldc=2 * D)
                xm = cat[:, D:]
            else:
                w2_ = p("cgmlp.channel_proj2.weight")
                if use_rd and ops.rowdot_ok(u, w2_):
                    xm, t_xm, rd[1] = ops.linear_drop(u, w2_, p("cgmlp.channel_proj2.bias"), pd, rowdot=(mp[1], mp[5]))
                else:
                    xm, t_xm = ops.linear_drop(u, w2_, p("cgmlp.channel_proj2.bias"), pd)   # x2 = dropout(x2)  (encoder_layer.py:224)
            sv["mlp"] = (mean, rstd, n, g, z, gn, gmean, grstd, u, conv, t_u, t_xm)
        br.join()
        t_cat = _drop_(cat, pd) if (merge == "concat" and cat is not None) else None   # both halves in one call (iid)
        wts = None
        x2 = None
        if two and merge == "learned_ave":
            if not cfg["merge_identity"] and ops.merge_proj_ok(xa, xm, p("merge_proj.weight"), T, D, res=x1):
                # merge + merge_proj + dropout + residual: the whole tail behind the join as ONE launch
                score, pooled, wts, m, x2, t_m = ops.merge_proj_fwd(xa, xm, lens, mp, p("merge_proj.weight"), p("merge_proj.bias"),
                                                                    x1, coeff, pd, B, T, save=need,
                                                                    rowdots=(rd[0], rd[1]) if rd[0] is not None and rd[1] is not None else None)
            else:
                score, pooled, wts, m = ops.merge_fwd(xa, xm, lens, mp, B, T)       # pooling + weighted sum: one launch for T <= 128
            sv["merge"] = (score, pooled, wts, m)
        elif two and merge == "fixed_ave":
            cw_ = cfg["cgmlp_weight"]
            m = ops.axpby(xa, xm, 1.0 - cw_, cw_)
            sv["merge"] = (m,)
        elif two and merge == "concat":
            m = cat
            sv["merge"] = (m,)
        else:
            m = xa if has_attn else xm
            sv["merge"] = (m,)
        if x2 is not None:
            pass                                            # (the fused tail above)
        elif cfg["merge_identity"]:
            t_m = None
            md = m
            if pd > 0.0:                                    # x + coeff * dropout(x1 | x2)  (encoder_layer.py:302-309)
                md, t_m = ops.dropout(m.contiguous(), pd)
            x2 = ops.axpby(x1, md, 1.0, coeff)
        else:                                               # x + coeff * dropout(merge_proj(.))  (:232-300)
            x2, t_m = ops.linear_drop(m, p("merge_proj.weight"), p("merge_proj.bias"), pd, alpha=coeff, res=x1)
        sv["drop"] = (t_cat, t_m)
        x3, sv["ff"], (y,), fmean, frstd = _FFN.fwd_ln(x2, p("norm_ff.weight"), p("norm_ff.bias"), p("feed_forward.w_1.weight"),
                                                       p("feed_forward.w_1.bias"), p("feed_forward.w_2.weight"),
                                                       p("feed_forward.w_2.bias"), act, 0.5,
                                                       [(p("norm_final.weight"), p("norm_final.bias"))], p=pd, save=need)
        sv["final"] = (x3, fmean, frstd)
        sv["x1"], sv["xa"], sv["xm"] = x1, xa, xm
        ctx.sv, ctx.cfg, ctx.P, ctx.lens, ctx.pos_emb = sv, cfg, P, lens, pos_emb
        ctx.shape = (B, T, D)
        cfg["_last_w"] = wts   # (weight_global, weight_local) for the introspection attributes
        return y.view(B, T, D)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        if (getattr(ctx, "cdesc", None) is not None and ops.LAYER_C and ops.PROFILE is None and all(p is not None for p in ctx.P)
                and not (ops.LAYER_C_EAGER_ONLY and torch.cuda.is_current_stream_capturing())):
            return _layer_c_backward(ctx, dy)
        sv, cfg, P = ctx.sv, ctx.cfg, ctx.P
        B, T, D = ctx.shape
        M = B * T
        H = cfg["heads"]
        dk = D // H
        act, merge = cfg["ffn_act"], cfg["merge"]
        has_attn, has_mlp = cfg["has_attn"], cfg["has_mlp"]
        coeff = cfg.get("coeff", 1.0)
        two = has_attn and has_mlp
        p = lambda n: P[_I[n]]
        G: List[Optional[torch.Tensor]] = [None] * len(BF_PARAM_NAMES)

        def put(name, g, like=None):
            G[_I[name]] = g if like is None else g.view_as(like)

        beside = ops.wgrad_may_go_beside(P)
        pos_dw = pos_sums = None
        grp = ops.WgradGroup()     # every weight gradient of the layer in one grouped launch (flushed at the end)
        lng = ops.LNGroup()        # ... and the five d=256 LayerNorms' (dgamma, dbeta) partials in one reduction
        dy2 = dy.contiguous().view(M, D)
        x3, fmean, frstd = sv["final"]
        # each LayerNorm backward whose dx the next block's dropout mask is applied to also writes the masked copy
        t_m = sv["drop"][1]
        t_ff, t_ffm = sv["ff"][-1], sv["ffm"][-1]
        dx3, g1, g2, *dyd = lng.bwd(dy2, x3, fmean, frstd, p("norm_final.weight"), drop=t_ff)
        put("norm_final.weight", g1); put("norm_final.bias", g2)
        dx2, gs, *dxd = _FFN.bwd(dx3, sv["ff"], p("norm_ff.weight"), p("feed_forward.w_1.weight"),
                                 p("feed_forward.w_2.weight"), act, 0.5, grp=grp, lng=lng, dyd=dyd[0] if dyd else None,
                                 out_drop=None if cfg["merge_identity"] else t_m)
        for n_, g in zip(("norm_ff.weight", "norm_ff.bias", "feed_forward.w_1.weight", "feed_forward.w_1.bias",
                          "feed_forward.w_2.weight", "feed_forward.w_2.bias"), gs):
            put(n_, g)
        # merge projection: x2 = x1 + coeff * (m Wm^T + bm)
        m = sv["merge"][-1]
        t_cat = sv["drop"][0]
        if cfg["merge_identity"]:
            dm = ops.axpby(dx2, None, coeff, 0.0) if (coeff != 1.0 or t_m is not None) else dx2
            _drop_bwd_(dm, t_m) if t_m is not None else None
        else:
            dxd = dxd[0] if dxd else _drop_bwd(dx2, t_m)
            gw_, gb_ = grp.add(dxd, m, alpha=coeff, bias_grad=True)
            put("merge_proj.weight", gw_); put("merge_proj.bias", gb_)
            dm = ops.linear_dx(dxd, p("merge_proj.weight"), alpha=coeff)
        xa, xm = sv["xa"], sv["xm"]
        masked = False
        if two and merge == "learned_ave":
            score, pooled, wts, _ = sv["merge"]
            mp = [p(k) for k in ("pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias",
                                 "pooling_proj2.bias", "weight_proj1.weight", "weight_proj2.weight",
                                 "weight_proj1.bias", "weight_proj2.bias")]
            # (dxa / dxm come back under the branch outputs' dropout masks: no mask launches at the head of the branches)
            dxa, dxm, mg = ops.merge_bwd(dm, xa, xm, ctx.lens, mp, score, pooled, wts, B, T, drop1=sv["attn"][-1],
                                         drop2=sv["mlp"][-1])
            masked = True
            for k, g in zip(("pooling_proj1.weight", "pooling_proj2.weight", "pooling_proj1.bias", "pooling_proj2.bias",
                             "weight_proj1.weight", "weight_proj2.weight", "weight_proj1.bias", "weight_proj2.bias"), mg):
                put(k, g, like=p(k))
        elif two and merge == "fixed_ave":
            cw_ = cfg["cgmlp_weight"]
            dxa = ops.axpby(dm, None, 1.0 - cw_, 0.0)
            dxm = ops.axpby(dm, None, cw_, 0.0)
        elif two and merge == "concat":
            _drop_bwd_(dm, t_cat)
            dxa, dxm = dm[:, :D], dm[:, D:]
        else:
            dxa = dm if has_attn else None
            dxm = dm if has_mlp else None

        x1 = sv["x1"]
        dx1 = dx2  # residual path; branch gradients are folded in through dx_add
        br = ops.BranchScope(two)     # attention-branch backward beside the cgMLP-branch backward
        dn_a = None
        with br:
            if has_attn:
                a_mean, a_rstd, n, qkv, pp, qu, qv, cx, attn, t_att, t_xa = sv["attn"]
                if t_xa is not None and not masked:
                    dxa = _drop_bwd(dxa.contiguous(), t_xa)
                gw_, gb_ = grp.add(dxa, cx, bias_grad=True)
                put("attn.linear_out.weight", gw_); put("attn.linear_out.bias", gb_)
                dcx = ops.linear_dx(dxa, p("attn.linear_out.weight"))
                dqkv = torch.empty_like(qkv)
                dqu = ops.empty(M, D, like=dy2)
                if qu is None:       # fused attention core
                    dqv, dp = _AttnFused.bwd(dcx, cx, attn, qkv, 0, qkv, D, qkv, 2 * D, dqu, 0, dqkv, D, dqkv, 2 * D, B, T, T,
                                             H, dk, ctx.lens, False, pos=pp, bias_u=p("attn.pos_bias_u").reshape(-1),
                                             bias_v=p("attn.pos_bias_v").reshape(-1), lazy_dp=beside and _POS_DW_BESIDE)
                else:
                    dqv, dp = _SelfAttnCore.bwd(dcx, attn, qu, D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, dqu, D, 0, dqkv, 3 * D, D,
                                                dqkv, 3 * D, 2 * D, B, T, T, H, dk, qv=qv, p=pp, tok=t_att)
                gu_, gv_, *late = ops.add2_colsum(dqu, dqv, dqkv[:, :D], lazy_sums=beside and _POS_DW_BESIDE)      # dQ = dQu + dQv and both bias gradients, one pass
                pos_sums = late[0] if late else None
                put("attn.pos_bias_u", gu_, like=p("attn.pos_bias_u"))
                put("attn.pos_bias_v", gv_, like=p("attn.pos_bias_v"))
                pe2d = ctx.pos_emb.reshape(-1, D)
                if callable(dp):      # the positional rows' gradient and linear_pos's weight gradient (4 launches, ~55 us of the attention
                    pos_dw = lambda dp=dp: put("attn.linear_pos.weight", ops.linear_dw(dp(), pe2d))      # branch): with the others, beside
                else:
                    put("attn.linear_pos.weight", ops.linear_dw(dp, pe2d))   # K = 2T-1: not a multiple of 32, stays alone
                for j, nm in enumerate(("q", "k", "v")):   # three problems with their own outputs (no sliced gradients)
                    gw_, gb_ = grp.add(dqkv[:, j * D:(j + 1) * D], n, bias_grad=True)
                    put(f"attn.linear_{nm}.weight", gw_); put(f"attn.linear_{nm}.bias", gb_)
                dn_a = ops.linear_dx_cat(dqkv, [p(f"attn.linear_{c}.weight") for c in "qkv"])    # one K = 3D GEMM
        if has_mlp:
            mean, rstd, n, g, z, gn, gmean, grstd, u, conv, t_u, t_xm = sv["mlp"]
            Cn = g.shape[1] // 2
            if t_xm is not None and not masked:
                dxm = _drop_bwd(dxm.contiguous(), t_xm)
            gw_, gb_ = grp.add(dxm, u, bias_grad=True)
            put("cgmlp.channel_proj2.weight", gw_); put("cgmlp.channel_proj2.bias", gb_)
            du = ops.linear_dx_drop(dxm, p("cgmlp.channel_proj2.weight"), t_u)
            dg = torch.empty_like(g)
            cw = p("cgmlp.csgu.conv.weight")
            fused = ops.CGMLP_ACT_BWD_FUSED and cw.shape[-1] == 31      # gelu'(z) applied by the two kernels that write dg's halves
            dgn, gcw, gcb = ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, cw.reshape(Cn, -1), dg[:, :Cn], B, T,
                                                zr=z[:, :Cn] if fused else None)
            put("cgmlp.csgu.conv.weight", gcw, like=cw); put("cgmlp.csgu.conv.bias", gcb)
            if fused:
                _, g1, g2 = ops.layernorm_bwd_act(dgn, g[:, Cn:], gmean, grstd, p("cgmlp.csgu.norm.weight"), z[:, Cn:], "gelu",
                                                  dx=dg[:, Cn:])
            else:
                _, g1, g2 = ops.layernorm_bwd(dgn, g[:, Cn:], gmean, grstd, p("cgmlp.csgu.norm.weight"), dx=dg[:, Cn:])
            put("cgmlp.csgu.norm.weight", g1); put("cgmlp.csgu.norm.bias", g2)
            if not fused:
                ops.act_bwd_(dg, z, "gelu")
            gw_, gb_ = grp.add(dg, n, bias_grad=True)
            put("cgmlp.channel_proj1.0.weight", gw_); put("cgmlp.channel_proj1.0.bias", gb_)
            dn = ops.linear_dx(dg, p("cgmlp.channel_proj1.0.weight"))
            dx1, g1, g2, *dyd = lng.bwd(dn, x1, mean, rstd, p("norm_mlp.weight"), dx_add=dx1, drop=None if has_attn else t_ffm)
            put("norm_mlp.weight", g1); put("norm_mlp.bias", g2)
        br.join()
        if has_attn:     # same accumulation order into dx1 as a single stream: cgMLP branch first, then attention
            dx1, g1, g2, *dyd = lng.bwd(dn_a, x1, a_mean, a_rstd, p("norm_mha.weight"), dx_add=dx1, drop=t_ffm)
            put("norm_mha.weight", g1); put("norm_mha.bias", g2)
        dx, gs = _FFN.bwd(dx1, sv["ffm"], p("norm_ff_macaron.weight"), p("feed_forward_macaron.w_1.weight"),
                          p("feed_forward_macaron.w_2.weight"), act, 0.5, grp=grp, lng=lng, dyd=dyd[0] if dyd else None)
        for n_, g in zip(("norm_ff_macaron.weight", "norm_ff_macaron.bias", "feed_forward_macaron.w_1.weight",
                          "feed_forward_macaron.w_1.bias", "feed_forward_macaron.w_2.weight",
                          "feed_forward_macaron.w_2.bias"), gs):
            put(n_, g)
        for i, prm in enumerate(P):
            if prm is None:
                G[i] = None
        if beside:       # no reader before the end of the pass: beside the next layer's chain
            ops.wgrad_beside(lambda: (pos_sums() if pos_sums else None, pos_dw() if pos_dw else None, grp.flush(), lng.flush()))
        else:
            grp.flush()
            lng.flush()
        ctx.sv = None
        return (dx.view(B, T, D), None, None, None, *G)


# ------------------------------------------------------------------------------------------------
# LayerNorm / Linear as stand-alone nodes (after_norm, ctc_lo, decoder output layer)
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mean, rstd, w)
        return y.view(shp)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        dx, gw, gb = ops.layernorm_bwd(dy.contiguous().view(x2.shape), x2, mean, rstd, w)
        return dx.view(dy.shape), gw, gb, None


class InterCTCConditionFn(torch.autograd.Function):
    """x + conditioning_layer(softmax(ctc_lo(h)))  - self-conditioned intermediate CTC
    (src/encoder/branchformer/encoder.py:389-401, tailored/encoder.py:296-318; ``CTC.softmax`` src/ctc/ctc.py:160-168).
    ``h`` is the (normalised) intermediate output the posteriors are taken from, ``x`` the stream they are added to."""

    @staticmethod
    def forward(ctx, x, h, ctc_w, ctc_b, cond_w, cond_b):
        shp = x.shape
        D = shp[-1]
        x2, h2 = x.reshape(-1, D), h.reshape(-1, D)
        M, V = x2.shape[0], ctc_w.shape[0]
        S = ops.pad4(V)
        logits = ops.empty(1, 1, M, S, like=x)
        ops.linear(h2, ctc_w, ctc_b, out=logits, ldc=S)
        vlen = torch.full((1,), V, dtype=torch.int64, device=x.device)
        prob = ops.softmax_fwd(logits, None, vlen, 1.0, T2=V)
        y = ops.empty(M, D, like=x)
        ops.gemm(M, D, V, prob, S, cond_w, cond_w.stride(0), y, D, bias=cond_b, R=x2, ldr=x2.stride(0))
        ctx.save_for_backward(h2, prob, ctc_w, cond_w)
        ctx.dims = (shp, M, V, S, D)
        return y.view(shp)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        h2, prob, ctc_w, cond_w = ctx.saved_tensors
        shp, M, V, S, D = ctx.dims
        dy2 = dy.contiguous().view(M, D)
        p2 = prob.view(M, S)
        # conditioning layer: y = x + prob W_c^T + b_c
        gcw = ops.empty(D, V, like=dy2)
        ops.gemm(D, V, M, dy2, D, p2, S, gcw, V, a_kmajor=True, b_kmajor=True)
        gcb = ops.colsum(dy2)
        dprob = ops.empty(1, 1, M, S, like=dy2)
        ops.gemm(M, V, D, dy2, D, cond_w, cond_w.stride(0), dprob, S, b_kmajor=True)
        dlog, _ = ops.softmax_bwd(prob, dprob, 1.0, T2=V)
        dl2 = dlog.view(M, S)
        # ctc_lo: logits = h W^T + b
        gw = ops.empty(V, D, like=dy2)
        ops.gemm(V, D, M, dl2, S, h2, h2.stride(0), gw, D, a_kmajor=True, b_kmajor=True)
        gb = ops.empty(V, like=dy2)
        ops.colsum(dl2[:, :V], out=gb)
        dh = ops.empty(M, D, like=dy2)
        ops.gemm(M, D, V, dl2, S, ctc_w, ctc_w.stride(0), dh, D, b_kmajor=True)
        return dy, dh.view(shp), gw, gb, gcw, gcb


class DropoutFn(torch.autograd.Function):
    """stand-alone dropout node (positional-encoding dropouts, src/ctc/ctc.py:143)."""

    @staticmethod
    def forward(ctx, x, p):
        y, ctx.tok = ops.dropout(x.contiguous(), p)
        return y

    @staticmethod
    @guarded
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.tok[0], token=ctx.tok)[0], None


class LinearFn(torch.autograd.Function):
    """y = alpha * (x W^T + b)."""

    @staticmethod
    def forward(ctx, x, w, b, alpha):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y = ops.linear(x2, w, b, alpha=alpha)
        ctx.save_for_backward(x2, w)
        ctx.alpha, ctx.has_b = alpha, b is not None
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    @guarded
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.contiguous().view(x2.shape[0], w.shape[0])
        dx = ops.linear_dx(dy2, w, alpha=ctx.alpha) if ctx.needs_input_grad[0] else None
        if ctx.has_b:
            gw, gb = ops.linear_dw(dy2, x2, alpha=ctx.alpha, bias_grad=True)
        else:
            gw, gb = ops.linear_dw(dy2, x2, alpha=ctx.alpha), None
        return (None if dx is None else dx.view(*dy.shape[:-1], w.shape[1])), gw, gb, None


# ------------------------------------------------------------------------------------------------
# Conv2dSubsampling (espnet subsampling.py; encoder.py:149-155,364): conv-relu-conv-relu-linear, x sqrt(d)
# ------------------------------------------------------------------------------------------------
CONV2_IMPLICIT = True      # Conv2dSubsampling's second convolution without im2col (shapes it does not take: im2col + GEMM)


class Conv2dSubsamplingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wo, bo, xscale):
        B, T, F = x.shape
        Cn = w1.shape[0]
        if CONV2_IMPLICIT and ops.BLOCKS_C and ops.conv2d_subsample_ok(x, w1, wo):      # the whole module as one C call
            out, T2, F2, kept, desc = ops.conv2d_subsample_fwd(x, w1.contiguous(), b1, w2.contiguous(), b2, wo.contiguous(), bo, xscale)
            ctx.cdesc, ctx.ckept, ctx.cparams = desc, kept, (w1, b1, w2, b2, wo, bo)
            ctx.save_for_backward()
            ctx.dims = (B, T, F, Cn, T2, F2, xscale, w1.shape, w2.shape, wo.shape)
            return out.view(B, T2, -1)
        ctx.cdesc, ctx.cparams = None, (w1, b1, w2, b2, wo, bo)
        y1 = ops.conv1_fwd(x.contiguous(), w1.reshape(Cn, 9), b1)              # [B,T1,F1,C] NHWC, relu
        # torch (co, ci, kh, kw) -> (co, kh, kw, ci) to match the channels-last patch order
        w2r = ops.transpose_inner(w2, Cn, Cn, 9).view(Cn, 9 * Cn)
        T1, F1 = y1.shape[1], y1.shape[2]
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        if CONV2_IMPLICIT and Cn % 64 == 0 and (B * T2 * F2) % 32 == 0:        # image rows as the GEMM operand: no patch matrix
            col = None
            y2 = ops.conv3x3_fwd(y1.view(B * T1 * F1, Cn), w2r, T1, F1, stride=2, pad0=True, bias=b2, act="relu")
        else:
            col, T2, F2 = ops.im2col3x3s2(y1)                                  # [B*T2*F2, 9C]
            y2 = ops.linear(col, w2r, b2, act="relu")                          # [B*T2*F2, C] == (b,t,f,c)
        # out Linear consumes (c*F2 + f); re-index its weight to (f*C + c) instead of transposing activations
        wor = ops.transpose_inner(wo, wo.shape[0], Cn, F2).view(wo.shape[0], F2 * Cn)
        out = ops.linear(y2.view(B * T2, F2 * Cn), wor, bo, alpha=xscale)
        ctx.save_for_backward(x, y1, col, y2, w2r, wor)      # col is None on the implicit route
        ctx.dims = (B, T, F, Cn, T2, F2, xscale, w1.shape, w2.shape, wo.shape)
        return out.view(B, T2, -1)

    @staticmethod
    @guarded
    def backward(ctx, dout):
        B, T, F, Cn, T2, F2, xscale, w1s, w2s, wos = ctx.dims
        do = dout.contiguous().view(B * T2, -1)
        if ctx.cdesc is not None:
            gw1, gb1, gw2, gb2, gwo, gbo = ops.conv2d_subsample_bwd(ctx.cdesc, do, (w1s, w2s, wos), params=ctx.cparams, kept=ctx.ckept)
            ctx.cdesc = ctx.ckept = ctx.cparams = None
            return None, gw1, gb1, gw2, gb2, gwo, gbo, None
        x, y1, col, y2, w2r, wor = ctx.saved_tensors
        y2f = y2.view(B * T2, F2 * Cn)
        # dz2 = (do @ wor) * xscale * relu'(y2)
        dz2 = ops.linear_dx(do, wor, alpha=xscale, DZ=y2f, dact="relu").view(B * T2 * F2, Cn)
        res = {}

        def wgrads():
            gwor, res["gbo"] = ops.linear_dw(do, y2f, alpha=xscale, bias_grad=True)        # [odim, F2*C], [odim]
            if col is None:
                gw2r, res["gb2"] = ops.conv3x3_dw(dz2, y1.view(-1, Cn), y1.shape[1], y1.shape[2], stride=2, pad0=True, bias_grad=True)
            else:
                gw2r, res["gb2"] = ops.linear_dw(dz2, col, bias_grad=True)                 # [C, 9C], [C]
            res["gwo"] = ops.transpose_inner(gwor, wos[0], F2, Cn).view(wos)
            res["gw2"] = ops.transpose_inner(gw2r, Cn, 9, Cn).view(w2s)

        ops.wgrad_beside(wgrads) if ops.wgrad_may_go_beside(ctx.cparams) else wgrads()      # (as the C-side block: beside the dgrad chain)
        dcol = ops.linear_dx(dz2, w2r)                                          # [B*T2*F2, 9C]
        dz1 = ops.col2im3x3s2_relu(dcol, y1)
        gw1, gb1 = ops.conv1_bwd(dz1, x.contiguous(), Cn)
        return None, gw1.view(w1s), gb1, res["gw2"], res["gb2"], res["gwo"], res["gbo"], None


# ------------------------------------------------------------------------------------------------
# CTC loss (src/ctc/ctc.py:133-158): Linear -> log_softmax -> CTCLoss(none, zero_infinity) -> sum/B
# ------------------------------------------------------------------------------------------------
class CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hs, w, b, hlens, ys, ylens, reduce, zero_infinity):
        B, T, D = hs.shape
        x2 = hs.reshape(B * T, D)
        logits = ops.linear(x2, w, b).view(B, T, -1)
        nll, g = ops.ctc_loss(logits, hlens, ys, ylens, 0, zero_infinity)
        ctx.save_for_backward(x2, w, g)
        ctx.reduce, ctx.B = reduce, B
        if reduce:
            return ops.colsum(nll.view(B, 1), scale=1.0 / B).view(())   # loss.sum() / B  (ctc.py:64-66)
        return ops.axpby(nll, None, 1.0 / B, 0.0)

    @staticmethod
    @guarded
    def backward(ctx, dl):
        x2, w, g = ctx.saved_tensors
        B = ctx.B
        V = w.shape[0]
        g2 = g.view(-1, V)
        if not ctx.reduce:
            raise NotImplementedError("reduce=False backward is not on the shipped path")
        gs = ops.scale_dev(g2, dl.contiguous(), 1.0 / B)   # dlogits = g * dl / B, dl stays on the device
        dx = ops.linear_dx(gs, w)
        gw, gb = ops.linear_dw(gs, x2, bias_grad=True)
        return dx.view(B, -1, x2.shape[1]), gw, gb, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# Transformer decoder, teacher forced (espnet2 TransformerDecoder.forward + DecoderLayer.forward,
# called at src/models/espnet_model.py:557-560).  The whole stack is ONE autograd node so the six
# per-layer gradients w.r.t. the encoder memory are accumulated in GEMM epilogues.
# ------------------------------------------------------------------------------------------------
DEC_LAYER_PARAM_NAMES = (
    "norm1.weight", "norm1.bias",
    "self_attn.linear_q.weight", "self_attn.linear_q.bias", "self_attn.linear_k.weight", "self_attn.linear_k.bias",
    "self_attn.linear_v.weight", "self_attn.linear_v.bias", "self_attn.linear_out.weight", "self_attn.linear_out.bias",
    "norm2.weight", "norm2.bias",
    "src_attn.linear_q.weight", "src_attn.linear_q.bias", "src_attn.linear_k.weight", "src_attn.linear_k.bias",
    "src_attn.linear_v.weight", "src_attn.linear_v.bias", "src_attn.linear_out.weight", "src_attn.linear_out.bias",
    "norm3.weight", "norm3.bias",
    "feed_forward.w_1.weight", "feed_forward.w_1.bias", "feed_forward.w_2.weight", "feed_forward.w_2.bias",
)
_NL = len(DEC_LAYER_PARAM_NAMES)
_DI = {n: i for i, n in enumerate(DEC_LAYER_PARAM_NAMES)}


class TransformerDecoderFn(torch.autograd.Function):
    """P = [embed.0.weight, (26 per layer) x num_blocks, after_norm.weight, after_norm.bias,
    output_layer.weight, output_layer.bias];  returns logits [B, L, V]."""

    @staticmethod
    def forward(ctx, memory, hlens, ys_in, ys_lens, pe, cfg, *P):
        need = _note_ctx(ctx)
        B, T, D = memory.shape
        L = ys_in.shape[1]
        H = cfg["heads"]
        dk = D // H
        nb = cfg["num_blocks"]
        M = B * L
        mem2 = memory.reshape(B * T, D)
        emb_w = P[0]
        pd, ppos, pself, psrc = (cfg.get(k, 0.0) for k in ("p", "p_pos", "p_self", "p_src"))
        x = ops.embed_pe(ys_in.contiguous(), emb_w, pe, math.sqrt(D)).view(M, D)
        t_pos = _drop_(x, ppos)                              # PositionalEncoding dropout
        saved = []
        # the key / value projections of the encoder memory do not depend on the decoder's state: all layers' in grouped launches up front
        # (2400 tiles of M = B T rows) instead of one 200-tile launch inside every layer's chain; layer li reads its window of kv_all
        ldkv = nb * 2 * D
        kv_all = ops.empty(B * T, ldkv, like=memory)
        kvp = [(P[1 + li * _NL + _DI[f"src_attn.linear_{c}.weight"]], P[1 + li * _NL + _DI[f"src_attn.linear_{c}.bias"]], (2 * li + j) * D)
               for li in range(nb) for j, c in enumerate("kv")]
        for i in range(0, len(kvp), 12):
            ops.linear_group(mem2, kvp[i: i + 12], kv_all)
        for li in range(nb):
            p = lambda n, li=li: P[1 + li * _NL + _DI[n]]
            s = {}
            # --- masked self attention
            n1, m1, r1 = ops.layernorm_fwd(x, p("norm1.weight"), p("norm1.bias"), EPS_ESPNET)
            qkv = ops.empty(M, 3 * D, like=x)
            ops.linear_group(n1, [(p(f"self_attn.linear_{c}.weight"), p(f"self_attn.linear_{c}.bias"), j * D)
                                  for j, c in enumerate("qkv")], qkv)
            fused = ops.ATTN_FUSED and dk == 64
            if fused:
                tk_a = "fused"
                cx, attn = _AttnFused.fwd(qkv, 0, qkv, D, qkv, 2 * D, B, L, L, H, dk, ys_lens, True, p_att=pself)
            else:
                cx, attn, tk_a = _SelfAttnCore.fwd(qkv, 3 * D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, B, L, L, H, dk, ys_lens, True,
                                                   p_att=pself)
            x1, tk_r = ops.linear_drop(cx, p("self_attn.linear_out.weight"), p("self_attn.linear_out.bias"), pd, res=x)   # x + dropout(self_attn(...))
            s["self"] = (x, m1, r1, n1, qkv, cx, attn, tk_a, tk_r)
            # --- source attention over the encoder memory
            n2, m2, r2 = ops.layernorm_fwd(x1, p("norm2.weight"), p("norm2.bias"), EPS_ESPNET)
            q2 = ops.linear(n2, p("src_attn.linear_q.weight"), p("src_attn.linear_q.bias"))
            ko = 2 * li * D
            if fused:
                tk_a2 = "fused"
                cx2, attn2 = _AttnFused.fwd(q2, 0, kv_all, ko, kv_all, ko + D, B, L, T, H, dk, hlens, False, p_att=psrc)
            else:
                cx2, attn2, tk_a2 = _SelfAttnCore.fwd(q2, D, 0, kv_all, ldkv, ko, kv_all, ldkv, ko + D, B, L, T, H, dk, hlens, False,
                                                      p_att=psrc)
            x2, tk_r2 = ops.linear_drop(cx2, p("src_attn.linear_out.weight"), p("src_attn.linear_out.bias"), pd, res=x1)   # x + dropout(src_attn(...))
            s["src"] = (x1, m2, r2, n2, q2, ko, cx2, attn2, tk_a2, tk_r2)
            # --- position-wise FFN (ReLU, scale 1)
            x, s["ff"] = _FFN.fwd(x2, p("norm3.weight"), p("norm3.bias"), p("feed_forward.w_1.weight"),
                                  p("feed_forward.w_1.bias"), p("feed_forward.w_2.weight"), p("feed_forward.w_2.bias"),
                                  "relu", 1.0, p=pd, save=need)
            saved.append(s)
        an_w, an_b, out_w, out_b = P[1 + nb * _NL: 1 + nb * _NL + 4]
        xn, mf, rf = ops.layernorm_fwd(x, an_w, an_b, EPS_ESPNET)
        logits = ops.linear(xn, out_w, out_b)
        ctx.saved, ctx.final, ctx.t_pos = saved, (x, mf, rf, xn), t_pos
        ctx.P, ctx.cfg, ctx.dims = P, cfg, (B, T, L, D, H, dk, nb)
        ctx.mem2, ctx.ys_in, ctx.hlens, ctx.ys_lens, ctx.kv_all = mem2, ys_in, hlens, ys_lens, kv_all
        return logits.view(B, L, -1)

    @staticmethod
    @guarded
    def backward(ctx, dlogits):
        P = ctx.P
        B, T, L, D, H, dk, nb = ctx.dims
        M = B * L
        G: List[Optional[torch.Tensor]] = [None] * len(P)
        an_i = 1 + nb * _NL
        an_w, an_b, out_w, out_b = P[an_i: an_i + 4]
        x, mf, rf, xn = ctx.final
        dl = dlogits.contiguous().view(M, -1)
        G[an_i + 2], G[an_i + 3] = ops.linear_dw(dl, xn, bias_grad=True)
        dxn = ops.linear_dx(dl, out_w)
        lng = ops.LNGroup(cap=3 * nb + 1)     # all LayerNorms of the decoder: one (dgamma, dbeta) reduction at the end
        # (every LayerNorm backward also writes its dx under the mask of the residual block below it: no dropout launches)
        dx, G[an_i], G[an_i + 1], *dyd = lng.bwd(dxn, x, mf, rf, an_w, drop=ctx.saved[nb - 1]["ff"][-1])
        mem2, kv_all = ctx.mem2, ctx.kv_all
        ldkv = nb * 2 * D
        dkv_all = torch.empty_like(kv_all)      # every layer's (dK | dV) side by side: the memory's gradient is ONE K = 2 D nb GEMM at the end
        # the weight gradients of ALL layers in a few grouped launches at the end: a layer's own group is 384 tiles of K = 1312 (1.5 per compute
        # unit, 42 TFLOP/s, 95 us of the 280 a layer's backward takes); 2304 tiles together run at the rate of the encoder's groups
        grp = ops.WgradGroup()
        beside = ops.wgrad_may_go_beside(P)
        for li in reversed(range(nb)):
            base = 1 + li * _NL
            p = lambda n, base=base: P[base + _DI[n]]

            def put(n, g, base=base):
                G[base + _DI[n]] = g

            s = ctx.saved[li]
            dx2, gs, *dt2 = _FFN.bwd(dx, s["ff"], p("norm3.weight"), p("feed_forward.w_1.weight"), p("feed_forward.w_2.weight"),
                                     "relu", 1.0, grp=grp, lng=lng, dyd=dyd[0] if dyd else None, out_drop=s["src"][-1])
            for n_, g in zip(("norm3.weight", "norm3.bias", "feed_forward.w_1.weight", "feed_forward.w_1.bias",
                              "feed_forward.w_2.weight", "feed_forward.w_2.bias"), gs):
                put(n_, g)
            # --- source attention
            x1, m2, r2, n2, q2, ko, cx2, attn2, tk_a2, tk_r2 = s["src"]
            dt2 = dt2[0] if dt2 else _drop_bwd(dx2, tk_r2)
            gw_, gb_ = grp.add(dt2, cx2, bias_grad=True)
            put("src_attn.linear_out.weight", gw_); put("src_attn.linear_out.bias", gb_)
            dcx2 = ops.linear_dx(dt2, p("src_attn.linear_out.weight"))
            dq2 = ops.empty(M, D, like=dl)
            if tk_a2 == "fused":
                _AttnFused.bwd(dcx2, cx2, attn2, q2, 0, kv_all, ko, kv_all, ko + D, dq2, 0, dkv_all, ko, dkv_all, ko + D, B, L, T, H, dk,
                               ctx.hlens, False)
            else:
                _SelfAttnCore.bwd(dcx2, attn2, q2, D, 0, kv_all, ldkv, ko, kv_all, ldkv, ko + D, dq2, D, 0, dkv_all, ldkv, ko,
                                  dkv_all, ldkv, ko + D, B, L, T, H, dk, tok=tk_a2)
            gw_, gb_ = grp.add(dq2, n2, bias_grad=True)
            put("src_attn.linear_q.weight", gw_); put("src_attn.linear_q.bias", gb_)
            for j, nm in enumerate(("k", "v")):
                gw_, gb_ = grp.add(dkv_all[:, ko + j * D: ko + (j + 1) * D], mem2, bias_grad=True)
                put(f"src_attn.linear_{nm}.weight", gw_); put(f"src_attn.linear_{nm}.bias", gb_)
            dn2 = ops.linear_dx(dq2, p("src_attn.linear_q.weight"))
            dx1, g1, g2, *dt1 = lng.bwd(dn2, x1, m2, r2, p("norm2.weight"), dx_add=dx2, drop=s["self"][-1])
            put("norm2.weight", g1); put("norm2.bias", g2)
            # --- self attention
            x0, m1, r1, n1, qkv, cx, attn, tk_a, tk_r = s["self"]
            dt1 = dt1[0] if dt1 else _drop_bwd(dx1, tk_r)
            gw_, gb_ = grp.add(dt1, cx, bias_grad=True)
            put("self_attn.linear_out.weight", gw_); put("self_attn.linear_out.bias", gb_)
            dcx = ops.linear_dx(dt1, p("self_attn.linear_out.weight"))
            dqkv = torch.empty_like(qkv)
            if tk_a == "fused":
                _AttnFused.bwd(dcx, cx, attn, qkv, 0, qkv, D, qkv, 2 * D, dqkv, 0, dqkv, D, dqkv, 2 * D, B, L, L, H, dk,
                               ctx.ys_lens, True)
            else:
                _SelfAttnCore.bwd(dcx, attn, qkv, 3 * D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, dqkv, 3 * D, 0, dqkv, 3 * D, D,
                                  dqkv, 3 * D, 2 * D, B, L, L, H, dk, tok=tk_a)
            for j, nm in enumerate(("q", "k", "v")):
                gw_, gb_ = grp.add(dqkv[:, j * D:(j + 1) * D], n1, bias_grad=True)
                put(f"self_attn.linear_{nm}.weight", gw_); put(f"self_attn.linear_{nm}.bias", gb_)
            dn1 = ops.linear_dx_cat(dqkv, [p(f"self_attn.linear_{c}.weight") for c in "qkv"])
            dx, g1, g2, *dyd = lng.bwd(dn1, x0, m1, r1, p("norm1.weight"), dx_add=dx1,
                                       drop=ctx.saved[li - 1]["ff"][-1] if li else None)
            put("norm1.weight", g1); put("norm1.bias", g2)
            if beside and li and (nb - li) % _DEC_WGRAD == 0:
                ops.wgrad_beside(grp.flush)
        # (in the layers' chain it was an accumulating K = 2 D launch over B T rows per layer - the only launches of the chain over the memory's
        # rows rather than the 1312 token rows)
        dmem = ops.linear_dx_cat(dkv_all, [P[1 + li * _NL + _DI[f"src_attn.linear_{c}.weight"]] for li in range(nb) for c in "kv"])
        grp.flush()
        lng.flush()
        _drop_bwd_(dx, ctx.t_pos)
        G[0] = ops.embed_bwd(ctx.ys_in.contiguous(), dx, math.sqrt(D), P[0].shape[0])
        ctx.saved = ctx.kv_all = None
        return (dmem.view(B, T, D), None, None, None, None, None, *G)


_DEC_WGRAD = int(os.environ.get("TAVSR_DEC_WGRAD", "5"))     # layers between two flushes of the decoder's weight gradients (6: one flush at the end)


class LabelSmoothingLossFn(torch.autograd.Function):
    """espnet LabelSmoothingLoss (KL, sum / batch) on decoder logits; also yields the th_accuracy counters."""

    @staticmethod
    def forward(ctx, logits, target, ignore, smoothing, normalize_length):
        B, L, V = logits.shape
        row, g, correct = ops.lsm_loss(logits.reshape(B * L, V), target.reshape(-1).contiguous(), ignore, smoothing)
        denom = B
        if normalize_length:      # espnet LabelSmoothingLoss: sum / number of real target tokens (a host count, as espnet's .item())
            if logits.is_cuda and torch.cuda.is_current_stream_capturing():
                raise NotImplementedError("length_normalized_loss=true reads the token count on the host: not capturable")
            denom = max(1, int((target != ignore).sum()))
        ctx.save_for_backward(g)
        ctx.B, ctx.denom = B, denom
        ctx.mark_non_differentiable(correct)
        return ops.colsum(row.view(-1, 1), scale=1.0 / denom).view(()), correct

    @staticmethod
    @guarded
    def backward(ctx, dl, _dc):
        (g,) = ctx.saved_tensors
        return ops.scale_dev(g, dl.contiguous(), 1.0 / ctx.denom).view(ctx.B, -1, g.shape[-1]), None, None, None, None


class WeightedSumFn(torch.autograd.Function):
    """loss = a*l1 + b*l2 on 0-dim device tensors (espnet_model.py:330)."""

    @staticmethod
    def forward(ctx, l1, l2, a, b):
        ctx.ab = (a, b)
        return ops.axpby(l1.reshape(1).contiguous(), l2.reshape(1).contiguous(), a, b).view(())

    @staticmethod
    @guarded
    def backward(ctx, dl):
        a, b = ctx.ab
        d = dl.reshape(1).contiguous()
        return ops.axpby(d, None, a, 0.0).view(()), ops.axpby(d, None, b, 0.0).view(()), None, None
