"""Hand-written forward/backward of the audio-visual blocks as
``torch.autograd.Function``s over the C ABI, in the same style as ``tavsr.functional``.

Reference semantics followed are cited per Function.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import ops
from ._lib import guarded
from .functional import EPS_ESPNET, _FFN, _AttnFused, _SelfAttnCore, _drop_, _drop_bwd, _drop_bwd_, _note_ctx

BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # torch.nn.BatchNorm defaults (conv3d_resnet18.py:57, resnet.py:39,68,84)


# ------------------------------------------------------------------------------------------------
# visual frontend: Conv3d stem + ResNet-18 trunk  (src/frontend/conv3d_resnet18/conv3d_resnet18.py:77-97,
# modules/resnet.py:89-106,167-178).  Activations are [N*H*W, C] matrices (N = B*T frames), convolutions are
# im2col + tavsr_gemm, BatchNorm uses batch statistics in training (running buffers updated in place) and the
# running statistics in eval.  3x3/stride-1 convolutions are implicit GEMMs (no im2col matrix); the stem patch matrix and the
# BN/Swish outputs feeding the second convolutions stay resident for the backward pass (HBM is plentiful).
# ------------------------------------------------------------------------------------------------
def frontend_param_names() -> List[str]:
    """differentiable parameters of Conv3dResNet18 in the order VisualFrontendFn takes them."""
    names = ["frontend3D.0.weight", "frontend3D.1.weight", "frontend3D.1.bias"]
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"trunk.layer{li}.{bi}."
            names += [p + "conv1.weight", p + "bn1.weight", p + "bn1.bias", p + "conv2.weight", p + "bn2.weight", p + "bn2.bias"]
            if bi == 0 and (li > 1 or inpl != planes):
                names += [p + "downsample.0.weight", p + "downsample.1.weight", p + "downsample.1.bias"]
        inpl = planes
    return names


def _w2d(w):
    """torch conv weight (co, ci, kh, kw) -> GEMM weight [co, (kh*kw)*ci] matching the channels-last im2col."""
    co, ci, kh, kw = w.shape
    return ops.transpose_inner(w.contiguous(), co, ci, kh * kw).view(co, kh * kw * ci)


def _w2d_grad(g, shape):
    co, ci, kh, kw = shape
    return ops.transpose_inner(g, co, kh * kw, ci).view(shape)


def _conv3x3_dw(dz, x, N, H, W, cin, stride=1, taps=9):
    """weight gradient of a 3x3/p1 (taps 9) or 1x1/p0 (taps 1) convolution [Cout, taps*Cin]: implicit GEMM when the output
    pixel count gives whole 32-row K-steps (every BASELINE shape), else through the im2col matrix."""
    if dz.shape[0] % 32 == 0 and cin % 64 == 0:
        return ops.conv3x3_dw(dz, x, H, W, stride, taps)
    k = 3 if taps == 9 else 1
    col, _, _ = ops.im2col2d(x, N, H, W, cin, k, k, stride, k // 2)
    return ops.linear_dw(dz, col)


def _plan_env(name, default):
    v = os.environ.get(name)
    if v is None:
        return default
    return None if v in ("", "0") else tuple(int(t) for t in v.split(","))


# tile variant and K split (tavsr_gemm_tune) of the stem's weight gradient [64 x 256] = dz0^T col0 over 6.2 M patch rows:
# 64x128 tiles read dz0 twice instead of four times (the launch is HBM-bound: 12.7 GB -> 9.5 GB); +0.35 % on the AV step
# against the planner's 64x64 / 250 slices (in-call A/B).  TAVSR_STEM_DW_PLAN=0 returns to the planner.
STEM_DW_PLAN = _plan_env("TAVSR_STEM_DW_PLAN", (3, 512))
STEM_POOL_FUSED = True


class _BN:
    """(mean, rstd) for a [M,C] matrix: batch statistics (+ running update) in training, running statistics in eval."""

    @staticmethod
    def stats(z, prefix, bufs, training):
        rm, rv, nbt = bufs[prefix + "running_mean"], bufs[prefix + "running_var"], bufs[prefix + "num_batches_tracked"]
        if training:
            return ops.bn_stats(z, BN_EPS, BN_MOMENTUM, rm, rv, nbt)
        return rm, ops.rsqrt_eps(rv, BN_EPS)


def _bn_eval_bwd(dy, z, mean, rstd, gamma, beta, res, act):
    """eval-mode BatchNorm backward (constant statistics): dz as in training, dx = dz * gamma * rstd."""
    # not used by training; kept minimal: training-mode formula with the batch terms dropped is not exposed by the
    # C ABI, and eval-mode backward is not on any reference path (validation runs under no_grad).
    raise NotImplementedError("backward through eval-mode BatchNorm is not on the reference's path (validation is no_grad)")


_FRONT_WGRAD_BESIDE = os.environ.get("TAVSR_FRONT_WGRAD_BESIDE", "1") != "0"


class VisualFrontendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, *P):
        names = cfg["names"]
        p = dict(zip(names, P))
        bufs, training = cfg["buffers"], cfg["training"]
        B, T, H, W = x.shape
        N = B * T
        x = x.contiguous()
        saved = {}
        # ---- stem: Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) -> BN3d -> Swish -> MaxPool(1,3,3)/(1,2,2)
        xp = col0 = w0 = None
        if ops.stem_pad16_ok(x):               # zero-padded clips, taps 35 x 8: the GEMM's ordinary 16-byte loader, no patch matrix
            xp = ops.stem_pad(x)
            w0 = ops.stem_weight_288(p["frontend3D.0.weight"])
            z0, H0, W0 = ops.stem_conv_fwd_pad16(xp, w0, B, T, H, W)
        else:
            w0 = ops.fill_(ops.empty(64, 256, like=x), 0.0)
            ops.copy2d(p["frontend3D.0.weight"].reshape(64, 245), w0[:, :245])
            if ops.stem_implicit_ok(x):        # every patch element is its own 4-byte gather: no padded copy either
                z0, H0, W0 = ops.stem_conv_fwd(x, w0)
            else:
                col0, H0, W0 = ops.im2col_stem(x)
                z0 = ops.linear(col0, w0)
        m0, r0 = _BN.stats(z0, "frontend3D.1.", bufs, training)
        if STEM_POOL_FUSED:      # BatchNorm + Swish inside the pool's window loads: the 1.6 GB activation map is never written
            cur, idx0, Hc, Wc = ops.bn_act_maxpool3x3s2_fwd(z0, m0, r0, p["frontend3D.1.weight"], p["frontend3D.1.bias"], "swish",
                                                            N, H0, W0, 64)
        else:
            y0 = ops.bn_apply_fwd(z0, m0, r0, p["frontend3D.1.weight"], p["frontend3D.1.bias"], None, "swish")
            cur, idx0, Hc, Wc = ops.maxpool3x3s2_fwd(y0, N, H0, W0, 64)
            del y0
        saved["stem"] = (x if xp is None else xp, z0, m0, r0, idx0, H0, W0, w0, col0, xp is not None)
        # ---- trunk
        blocks = []
        cin = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):
            for bi in range(2):
                pre = f"trunk.layer{li}.{bi}."
                stride = 2 if (li > 1 and bi == 0) else 1
                has_ds = (pre + "downsample.0.weight") in p
                Xin, Hin, Win = cur, Hc, Wc
                w1 = _w2d(p[pre + "conv1.weight"])
                # implicit GEMM: the image itself is the A operand (no im2col matrix), also for the stride-2 blocks
                Ho, Wo = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
                z1 = ops.conv3x3_fwd(Xin, w1, Hin, Win, stride)
                m1, r1 = _BN.stats(z1, pre + "bn1.", bufs, training)
                y1 = ops.bn_apply_fwd(z1, m1, r1, p[pre + "bn1.weight"], p[pre + "bn1.bias"], None, "swish")
                w2 = _w2d(p[pre + "conv2.weight"])
                z2 = ops.conv3x3_fwd(y1, w2, Ho, Wo)
                m2, r2 = _BN.stats(z2, pre + "bn2.", bufs, training)
                ds = None
                if has_ds:
                    wd = _w2d(p[pre + "downsample.0.weight"])
                    zd = ops.conv3x3_fwd(Xin, wd, Hin, Win, stride, taps=1)
                    md, rd = _BN.stats(zd, pre + "downsample.1.", bufs, training)
                    res = ops.bn_apply_fwd(zd, md, rd, p[pre + "downsample.1.weight"], p[pre + "downsample.1.bias"], None, None)
                    ds = (zd, md, rd, wd)
                else:
                    res = Xin
                cur = ops.bn_apply_fwd(z2, m2, r2, p[pre + "bn2.weight"], p[pre + "bn2.bias"], res, "swish")
                blocks.append((pre, stride, cin, planes, Hin, Win, Ho, Wo, Xin, z1, m1, r1, w1, z2, m2, r2, w2, res, ds, y1))
                cin, Hc, Wc = planes, Ho, Wo
        feat = ops.avgpool_fwd(cur, N, Hc * Wc, cin)
        ctx.saved, ctx.blocks, ctx.p, ctx.names = saved, blocks, p, names
        ctx.dims = (B, T, N, Hc, Wc, cin)
        ctx.training = training
        return feat.view(B, T, cin)

    @staticmethod
    @guarded
    def backward(ctx, dfeat):
        if not ctx.training:
            _bn_eval_bwd(None, None, None, None, None, None, None, None)
        p = ctx.p
        B, T, N, Hc, Wc, cl = ctx.dims
        G = {}
        d = ops.avgpool_bwd(dfeat.contiguous().view(N, cl), N, Hc * Wc, cl)
        # The chain of this backward is BatchNorm backward (two passes over the maps: HBM-bound) -> dgrad convolution (MFMA-bound) -> BatchNorm
        # backward -> ...; the weight-gradient convolutions (12.7 ms of the step at batch 32) hang off it with no reader inside the pass.  On the
        # side queue, un-joined (ops.wgrad_beside), they run under the BatchNorm passes instead of between them.
        beside = _FRONT_WGRAD_BESIDE and ops.wgrad_may_go_beside(list(p.values()))

        def wgrad(name, fn):
            if beside:
                ops.wgrad_beside(lambda: G.__setitem__(name, fn()))
            else:
                G[name] = fn()

        def self_ds_bwd(ds, dres, pre, Xin, N, Hin, Win, cin, stride):
            """backward of the 1x1 downsample convolution + its BatchNorm: parameter gradients into G, returns the data
            gradient rows [N*Ho*Wo, cin] (still to be scattered to the stride-s pixels)"""
            zd, md, rd, wd = ds
            _, dzd, G[pre + "downsample.1.weight"], G[pre + "downsample.1.bias"] = ops.bn_bwd(
                dres, zd, md, rd, p[pre + "downsample.1.weight"], p[pre + "downsample.1.bias"], None, None, need_dz=False)
            wgrad(pre + "downsample.0.weight", lambda: _w2d_grad(_conv3x3_dw(dzd, Xin, N, Hin, Win, cin, stride, taps=1),
                                                                 p[pre + "downsample.0.weight"].shape))
            return ops.linear_dx(dzd, wd)

        for (pre, stride, cin, planes, Hin, Win, Ho, Wo, Xin, z1, m1, r1, w1, z2, m2, r2, w2, res, ds, y1) in reversed(ctx.blocks):
            # out = swish(bn2(z2) + res)
            dres, dz2, G[pre + "bn2.weight"], G[pre + "bn2.bias"] = ops.bn_bwd(
                d, z2, m2, r2, p[pre + "bn2.weight"], p[pre + "bn2.bias"], res, "swish")
            wgrad(pre + "conv2.weight", lambda: _w2d_grad(_conv3x3_dw(dz2, y1, N, Ho, Wo, planes), p[pre + "conv2.weight"].shape))
            dy1 = ops.conv3x3_dx(dz2, ops.conv_wflip(w2, planes, planes), Ho, Wo)
            _, dz1, G[pre + "bn1.weight"], G[pre + "bn1.bias"] = ops.bn_bwd(
                dy1, z1, m1, r1, p[pre + "bn1.weight"], p[pre + "bn1.bias"], None, "swish", need_dz=False)
            fused_ds = False
            if stride == 1:
                wgrad(pre + "conv1.weight", lambda: _w2d_grad(_conv3x3_dw(dz1, Xin, N, Hin, Win, cin), p[pre + "conv1.weight"].shape))
                # identity skip: its gradient joins in the GEMM epilogue (no separate add over the 0.4 GB maps)
                dX = ops.conv3x3_dx(dz1, ops.conv_wflip(w1, planes, cin), Hin, Win, res=None if ds is not None else dres)
            else:
                wgrad(pre + "conv1.weight", lambda: _w2d_grad(_conv3x3_dw(dz1, Xin, N, Hin, Win, cin, stride), p[pre + "conv1.weight"].shape))
                dcol1 = ops.linear_dx(dz1, w1)
                dcold = None
                if ds is not None:   # the downsample path's data gradient joins inside the col2im pass (same stride, same input)
                    dcold = self_ds_bwd(ds, dres, pre, Xin, N, Hin, Win, cin, stride)
                dX = ops.col2im2d(dcol1, N, Hin, Win, cin, 3, 3, stride, 1, extra=dcold)
                del dcol1
                fused_ds = ds is not None
            if ds is not None and not fused_ds:
                dcold = self_ds_bwd(ds, dres, pre, Xin, N, Hin, Win, cin, stride)
                dXd = ops.col2im2d(dcold, N, Hin, Win, cin, 1, 1, stride, 0)
                d = ops.axpby(dX, dXd, 1.0, 1.0)
            elif stride == 1 or fused_ds:     # the skip / downsample gradient is already inside dX
                d = dX
            else:
                d = ops.axpby(dX, dres, 1.0, 1.0)
        # ---- stem
        x, z0, m0, r0, idx0, H0, W0, w0, col0, padded = ctx.saved["stem"]
        # max-pool backward inside the BatchNorm backward passes: the 1.6 GB gradient of the pool's input is never written
        dz0, G["frontend3D.1.weight"], G["frontend3D.1.bias"] = ops.bn_bwd_pooled(
            d.contiguous(), idx0, z0, m0, r0, p["frontend3D.1.weight"], p["frontend3D.1.bias"], N, H0, W0, "swish")
        if padded:
            T_ = ctx.dims[1]
            gw0 = ops.stem_conv_dw_pad16(dz0, x, T_, 2 * H0, 2 * W0)      # [64, 288] in the 35 x 8 tap layout
            G["frontend3D.0.weight"] = ops.stem_weight_grad_from_288(gw0, p["frontend3D.0.weight"].shape)
        else:
            if col0 is None:
                gw0 = ops.stem_conv_dw(dz0, x)                            # [64, 256], columns >= 245 are padding
            else:
                gw0 = ops.linear_dw(dz0, col0, force=STEM_DW_PLAN)
            g0 = ops.empty(64, 245, like=gw0)
            ops.copy2d(gw0[:, :245], g0)
            G["frontend3D.0.weight"] = g0.view(p["frontend3D.0.weight"].shape)
        del col0
        ctx.saved = ctx.blocks = None
        return (None, None, *[G[n] for n in ctx.names])


# ------------------------------------------------------------------------------------------------
# small differentiable glue used by the AV embedding / alignment / modality encoding
# ------------------------------------------------------------------------------------------------
class ScaleFn(torch.autograd.Function):
    """y = s * x  (RelPositionalEncoding's xscale, applied after the AV alignment: default.py:157-162)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return ops.axpby(x.contiguous(), None, s, 0.0)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        return ops.axpby(dy.contiguous(), None, ctx.s, 0.0), None


class PadTimeFn(torch.autograd.Function):
    """x [B,T,D] -> [B,T+p,D] with the p new frames filled with ``value`` (audiovisual_alignment pads the shorter
    stream's FEATURES with ignore_id = -1.0: avsr_espnet_model.py:531-538)."""

    @staticmethod
    def forward(ctx, x, p, value):
        B, T, D = x.shape
        x = x.contiguous()
        out = ops.fill_(ops.empty(B, T + p, D, like=x), float(value))
        ops.copy2d(x.view(B, T * D), out.view(B, (T + p) * D)[:, : T * D])
        ctx.dims = (B, T, D)
        return out

    @staticmethod
    @guarded
    def backward(ctx, dy):
        B, T, D = ctx.dims
        dy = dy.contiguous()
        dx = ops.empty(B, T * D, like=dy)
        ops.copy2d(dy.view(B, -1)[:, : T * D], dx)
        return dx.view(B, T, D), None, None


class AddRowFn(torch.autograd.Function):
    """x [B,T,D] + row [D]  (modality encoding, tailored/encoder.py:251-263); d row = column sum of dy."""

    @staticmethod
    def forward(ctx, x, row):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        y, _ = ops.add_head_bias(x2, row.contiguous(), row.contiguous())
        return y.view(shp)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        d2 = dy.contiguous().view(-1, dy.shape[-1])
        return dy, ops.colsum(d2)


# ------------------------------------------------------------------------------------------------
# one modality stream of a TailoredEncoderLayer (src/encoder/audiovisual/tailored/encoder_layer.py:171-216 audio,
# :218-264 video): macaron FFN -> (rel-pos MHSA | cgMLP) with its own LayerNorm and residual -> FFN -> norm_final.
# The FFNs and the three shared norms get gradients from both streams' nodes; autograd sums them.
# ------------------------------------------------------------------------------------------------
TS_SHARED = ("norm_ff_macaron.weight", "norm_ff_macaron.bias",
             "feed_forward_macaron.w_1.weight", "feed_forward_macaron.w_1.bias",
             "feed_forward_macaron.w_2.weight", "feed_forward_macaron.w_2.bias",
             "norm_ff.weight", "norm_ff.bias", "feed_forward.w_1.weight", "feed_forward.w_1.bias",
             "feed_forward.w_2.weight", "feed_forward.w_2.bias", "norm_final.weight", "norm_final.bias")
TS_ATTN = ("norm_mha.weight", "norm_mha.bias", "attn.linear_q.weight", "attn.linear_q.bias", "attn.linear_k.weight",
           "attn.linear_k.bias", "attn.linear_v.weight", "attn.linear_v.bias", "attn.linear_out.weight",
           "attn.linear_out.bias", "attn.linear_pos.weight", "attn.pos_bias_u", "attn.pos_bias_v")
TS_MLP = ("norm_cgmlp.weight", "norm_cgmlp.bias", "cgmlp.channel_proj1.0.weight", "cgmlp.channel_proj1.0.bias",
          "cgmlp.csgu.norm.weight", "cgmlp.csgu.norm.bias", "cgmlp.csgu.conv.weight", "cgmlp.csgu.conv.bias",
          "cgmlp.channel_proj2.weight", "cgmlp.channel_proj2.bias")


def tailored_stream_param_names(use_attn: bool):
    return TS_SHARED + (TS_ATTN if use_attn else TS_MLP)


def _ts_branch_fwd(p, cfg, x1, n, mean, rstd, pos_emb, lens, B, T, need):
    """the stream's attention OR cgMLP branch with its residual (src/encoder/audiovisual/tailored/encoder_layer.py:185-208,
    232-256) on the already normalised rows ``n``: returns (x2, saved)."""
    M, D = x1.shape
    H = cfg["heads"]
    dk = D // H
    coeff = cfg.get("coeff", 1.0)
    pd, pa = cfg.get("p", 0.0), cfg.get("p_att", 0.0)      # dropout rates (0 in eval)
    if cfg["use_attn"]:
        qkv = ops.empty(M, 3 * D, like=x1)
        ops.linear_group(n, [(p[f"attn.linear_{c}.weight"], p[f"attn.linear_{c}.bias"], j * D) for j, c in enumerate("qkv")],
                         qkv)
        pp = ops.linear(pos_emb.reshape(-1, D), p["attn.linear_pos.weight"])
        if ops.ATTN_FUSED and dk == 64:
            qu = qv = t_att = None
            cx, attn = _AttnFused.fwd(qkv, 0, qkv, D, qkv, 2 * D, B, T, T, H, dk, lens, False, pos=pp,
                                      bias_u=p["attn.pos_bias_u"].reshape(-1), bias_v=p["attn.pos_bias_v"].reshape(-1),
                                      p_att=pa)
        else:
            qu, qv = ops.add_head_bias(qkv[:, :D], p["attn.pos_bias_u"].reshape(-1), p["attn.pos_bias_v"].reshape(-1))
            cx, attn, t_att = _SelfAttnCore.fwd(qu, D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, B, T, T, H, dk, lens, False,
                                                qv=qv, p=pp, p_att=pa)
        # residual + coeff * dropout(att)  (encoder_layer.py:196,243): the dropout rides in the GEMM epilogue
        x2, t_br = ops.linear_drop(cx, p["attn.linear_out.weight"], p["attn.linear_out.bias"], pd, alpha=coeff, res=x1)
        return x2, (mean, rstd, n, qkv, pp, qu, qv, cx, attn, t_att, t_br)
    w1c = p["cgmlp.channel_proj1.0.weight"]
    cw = p["cgmlp.csgu.conv.weight"]
    if ops.BLOCKS_C and ops.cgmlp_block_ok(n, w1c, cw) and x1.is_contiguous():      # the whole branch as one C call
        x2, (g, z, gn, gmean, grstd, u, conv, t_u, t_br), _ = ops.cgmlp_fwd(
            n, w1c, p["cgmlp.channel_proj1.0.bias"], p["cgmlp.csgu.norm.weight"], p["cgmlp.csgu.norm.bias"], cw,
            p["cgmlp.csgu.conv.bias"], p["cgmlp.channel_proj2.weight"], p["cgmlp.channel_proj2.bias"], B, T, p=pd, p_out=pd, alpha=coeff,
            res=x1, save=need)
        return x2, (mean, rstd, n, g, z, gn, gmean, grstd, u, conv, t_u, t_br)
    # channel_proj1's epilogue leaves the CSGU's LayerNorm statistics as per-tile row sums (no statistics launch)
    rst = (ops.empty(n.shape[0], w1c.shape[0] // 64, 2, like=n)
           if (ops.CSGU_FUSED and cw.shape[-1] == 31 and ops.csgu_rowstat_ok(n, w1c)) else None)
    if need:
        g, z = ops.linear(n, w1c, p["cgmlp.channel_proj1.0.bias"], act="gelu", save_z=True, rowstat=rst)
    else:
        g, z = ops.linear(n, w1c, p["cgmlp.channel_proj1.0.bias"], act="gelu", rowstat=rst), None
    Cn = g.shape[1] // 2
    if ops.csgu_usable(g, cw):         # LayerNorm + depthwise convolution + gate + dropout: one pass over g
        u, conv, gn, gmean, grstd, t_u = ops.csgu_fwd(g, p["cgmlp.csgu.norm.weight"], p["cgmlp.csgu.norm.bias"], EPS_ESPNET,
                                                      cw.reshape(Cn, -1), p["cgmlp.csgu.conv.bias"], B, T, p=pd, save=need,
                                                      rowstat=rst)
    else:
        gn, gmean, grstd = ops.layernorm_fwd(g[:, Cn:], p["cgmlp.csgu.norm.weight"], p["cgmlp.csgu.norm.bias"], EPS_ESPNET)
        u, conv = ops.dwconv_gate_fwd(gn, g[:, :Cn], cw.reshape(Cn, -1), p["cgmlp.csgu.conv.bias"], B, T)
        t_u = _drop_(u, pd)            # csgu: dropout(x_r * x_g)
    # residual + coeff * dropout(cgmlp)  (encoder_layer.py:208,256)
    x2, t_br = ops.linear_drop(u, p["cgmlp.channel_proj2.weight"], p["cgmlp.channel_proj2.bias"], pd, alpha=coeff, res=x1)
    return x2, (mean, rstd, n, g, z, gn, gmean, grstd, u, conv, t_u, t_br)


def _ts_branch_bwd(p, cfg, saved, dx2, x1, pos_emb, lens, B, T, grp, lng, G, dbr=None, out_drop=None):
    """backward of _ts_branch_fwd: returns dx1 (the residual path included) and fills ``G`` with the branch's gradients.
    ``dbr``: dx2 under the branch's outer mask when the caller's LayerNorm backward already produced it; ``out_drop``: the
    token of the block below - the branch's own LayerNorm backward then also returns dx1 under that mask (dx1, dx1_masked)."""
    M, D = x1.shape
    H = cfg["heads"]
    dk = D // H
    coeff = cfg.get("coeff", 1.0)
    if cfg["use_attn"]:
        mean, rstd, n, qkv, pp, qu, qv, cx, attn, t_att, t_br = saved
        dbr = _drop_bwd(dx2, t_br) if dbr is None else dbr
        G["attn.linear_out.weight"], G["attn.linear_out.bias"] = grp.add(dbr, cx, alpha=coeff, bias_grad=True)
        dcx = ops.linear_dx(dbr, p["attn.linear_out.weight"], alpha=coeff)
        dqkv = torch.empty_like(qkv)
        dqu = ops.empty(M, D, like=dx2)
        if qu is None:           # fused attention core
            dqv, dp = _AttnFused.bwd(dcx, cx, attn, qkv, 0, qkv, D, qkv, 2 * D, dqu, 0, dqkv, D, dqkv, 2 * D, B, T, T, H, dk,
                                     lens, False, pos=pp, bias_u=p["attn.pos_bias_u"].reshape(-1),
                                     bias_v=p["attn.pos_bias_v"].reshape(-1))
        else:
            dqv, dp = _SelfAttnCore.bwd(dcx, attn, qu, D, 0, qkv, 3 * D, D, qkv, 3 * D, 2 * D, dqu, D, 0, dqkv, 3 * D, D,
                                        dqkv, 3 * D, 2 * D, B, T, T, H, dk, qv=qv, p=pp, tok=t_att)
        gu_, gv_ = ops.add2_colsum(dqu, dqv, dqkv[:, :D])
        G["attn.pos_bias_u"], G["attn.pos_bias_v"] = gu_.view_as(p["attn.pos_bias_u"]), gv_.view_as(p["attn.pos_bias_v"])
        G["attn.linear_pos.weight"] = ops.linear_dw(dp, pos_emb.reshape(-1, D))
        for j, nm in enumerate(("q", "k", "v")):
            G[f"attn.linear_{nm}.weight"], G[f"attn.linear_{nm}.bias"] = grp.add(dqkv[:, j * D:(j + 1) * D], n, bias_grad=True)
        dn = ops.linear_dx_cat(dqkv, [p[f"attn.linear_{c}.weight"] for c in "qkv"])      # one K = 3D GEMM
        dx1, G["norm_mha.weight"], G["norm_mha.bias"], *dxd = lng.bwd(dn, x1, mean, rstd, p["norm_mha.weight"], dx_add=dx2,
                                                                      drop=out_drop)
        return (dx1, dxd[0]) if dxd else dx1
    mean, rstd, n, g, z, gn, gmean, grstd, u, conv, t_u, t_br = saved
    Cn = g.shape[1] // 2
    dbr = _drop_bwd(dx2, t_br) if dbr is None else dbr
    G["cgmlp.channel_proj2.weight"], G["cgmlp.channel_proj2.bias"] = grp.add(dbr, u, alpha=coeff, bias_grad=True)
    du = ops.linear_dx_drop(dbr, p["cgmlp.channel_proj2.weight"], t_u, alpha=coeff)
    dg = torch.empty_like(g)
    cw = p["cgmlp.csgu.conv.weight"]
    fused = ops.CGMLP_ACT_BWD_FUSED and cw.shape[-1] == 31      # gelu'(z) applied by the two kernels that write dg's halves
    dgn, gcw, gcb = ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, cw.reshape(Cn, -1), dg[:, :Cn], B, T,
                                        zr=z[:, :Cn] if fused else None)
    G["cgmlp.csgu.conv.weight"], G["cgmlp.csgu.conv.bias"] = gcw.view_as(cw), gcb
    if fused:
        _, G["cgmlp.csgu.norm.weight"], G["cgmlp.csgu.norm.bias"] = ops.layernorm_bwd_act(
            dgn, g[:, Cn:], gmean, grstd, p["cgmlp.csgu.norm.weight"], z[:, Cn:], "gelu", dx=dg[:, Cn:])
    else:
        _, G["cgmlp.csgu.norm.weight"], G["cgmlp.csgu.norm.bias"] = ops.layernorm_bwd(
            dgn, g[:, Cn:], gmean, grstd, p["cgmlp.csgu.norm.weight"], dx=dg[:, Cn:])
        ops.act_bwd_(dg, z, "gelu")
    G["cgmlp.channel_proj1.0.weight"], G["cgmlp.channel_proj1.0.bias"] = grp.add(dg, n, bias_grad=True)
    dn = ops.linear_dx(dg, p["cgmlp.channel_proj1.0.weight"])
    dx1, G["norm_cgmlp.weight"], G["norm_cgmlp.bias"], *dxd = lng.bwd(dn, x1, mean, rstd, p["norm_cgmlp.weight"], dx_add=dx2,
                                                                      drop=out_drop)
    return (dx1, dxd[0]) if dxd else dx1


_FFM = ("norm_ff_macaron.weight", "norm_ff_macaron.bias", "feed_forward_macaron.w_1.weight", "feed_forward_macaron.w_1.bias",
        "feed_forward_macaron.w_2.weight", "feed_forward_macaron.w_2.bias")
_FF = ("norm_ff.weight", "norm_ff.bias", "feed_forward.w_1.weight", "feed_forward.w_1.bias", "feed_forward.w_2.weight",
       "feed_forward.w_2.bias")


class TailoredStreamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos_emb, lens, cfg, *P):
        names = tailored_stream_param_names(cfg["use_attn"])
        p = dict(zip(names, P))
        B, T, D = x.shape
        M = B * T
        act = cfg["ffn_act"]
        pd = cfg.get("p", 0.0)
        need = cfg.get("need_bwd", True)                       # set by TailoredLayerFn: does the node get a backward pass?
        x2d = x.contiguous().view(M, D)
        sv = {}
        bn = "norm_mha" if cfg["use_attn"] else "norm_cgmlp"     # the branch's LayerNorm rides in the macaron block's finishing launch
        x1, sv["ffm"], (n,), mean, rstd = _FFN.fwd_ln(x2d, *[p[k] for k in _FFM], act, 0.5, [(p[bn + ".weight"], p[bn + ".bias"])],
                                                      p=pd, save=need)
        x2, sv["br"] = _ts_branch_fwd(p, cfg, x1, n, mean, rstd, pos_emb, lens, B, T, need)
        x3, sv["ff"], (y,), fmean, frstd = _FFN.fwd_ln(x2, *[p[k] for k in _FF], act, 0.5,
                                                       [(p["norm_final.weight"], p["norm_final.bias"])], p=pd, save=need)
        sv["final"] = (x3, fmean, frstd)
        sv["x1"] = x1
        ctx.sv, ctx.cfg, ctx.p, ctx.names, ctx.pos_emb, ctx.shape, ctx.lens = sv, cfg, p, names, pos_emb, (B, T, D), lens
        return y.view(B, T, D)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        sv, cfg, p = ctx.sv, ctx.cfg, ctx.p
        B, T, D = ctx.shape
        M = B * T
        act = cfg["ffn_act"]
        G = {}
        grp = ops.WgradGroup()
        lng = ops.LNGroup()        # the stream's four d-wide LayerNorms: one (dgamma, dbeta) reduction
        x3, fmean, frstd = sv["final"]
        # (each LayerNorm backward also writes its dx under the outer mask of the block below: no stand-alone mask launches)
        t_ff, t_br, t_ffm = sv["ff"][-1], sv["br"][-1], sv["ffm"][-1]
        dx3, G["norm_final.weight"], G["norm_final.bias"], *dyd = lng.bwd(dy.contiguous().view(M, D), x3, fmean, frstd,
                                                                         p["norm_final.weight"], drop=t_ff)
        # chain=False: the two modality streams run side by side (measured on the AV step: 367.9 utt/s with the dgrad GEMMs,
        # 365.3 with the streaming launch, which cannot share the chip with the other stream's kernels)
        dx2, gs, *dbr = _FFN.bwd(dx3, sv["ff"], p["norm_ff.weight"], p["feed_forward.w_1.weight"], p["feed_forward.w_2.weight"], act,
                                 0.5, grp=grp, lng=lng, chain=False, dyd=dyd[0] if dyd else None, out_drop=t_br)
        G.update(zip(_FF, gs))
        dx1 = _ts_branch_bwd(p, cfg, sv["br"], dx2, sv["x1"], ctx.pos_emb, ctx.lens, B, T, grp, lng, G,
                             dbr=dbr[0] if dbr else None, out_drop=t_ffm)
        dx1, dyd = dx1 if isinstance(dx1, tuple) else (dx1, None)
        dx, gs = _FFN.bwd(dx1, sv["ffm"], p["norm_ff_macaron.weight"], p["feed_forward_macaron.w_1.weight"],
                          p["feed_forward_macaron.w_2.weight"], act, 0.5, grp=grp, lng=lng, chain=False, dyd=dyd)
        G.update(zip(_FFM, gs))
        grp.flush()        # (on the side queue, un-joined, as the Branchformer layer does: 380 -> 375 utt/s on the AV step - the two modality
        lng.flush()        # streams already share the chip, and the second stream's side queue is a third queue)
        ctx.sv = None
        return (dx.view(B, T, D), None, None, None, *[G[n] for n in ctx.names])


# (One feed-forward call for BOTH modality streams - the shared FFNs as one [Ma + Mv, 256] problem, SURVEY a9 - was built and
# measured in round 3: 363.5-364.3 utt/s against 367.2-367.5 per stream on the batch-32 AV step, profiles/r03_notes.md section 3;
# the per-stream form keeps two launch queues whose kernels fill each other's prologue / finishing phases.  Removed in round 4.)


_TS_WS = {}


def _ts_c_ok(x, cfg, p) -> bool:
    """the shapes csrc/layer.hip's tailored-stream sequencer takes (un-captured loops only unless TAVSR_LAYER_C=capture)"""
    if not ops.LAYER_C or ops.PROFILE is not None or (ops.LAYER_C_EAGER_ONLY and torch.cuda.is_current_stream_capturing()):
        return False
    B, T, D = x.shape
    w1, w1m = p["feed_forward.w_1.weight"], p["feed_forward_macaron.w_1.weight"]
    if not (x.is_cuda and D == 256 and D // cfg["heads"] == 64 and w1.shape == w1m.shape and w1.shape[0] >= 1024 and w1.shape[0] % 32 == 0):
        return False
    if cfg["use_attn"]:
        return True
    cw, c1 = p["cgmlp.csgu.conv.weight"], p["cgmlp.channel_proj1.0.weight"]
    return cw.shape[-1] == 31 and c1.shape[0] % 128 == 0 and c1.shape[0] // 2 <= 1024


def _ts_c_desc(ns, x, pos_emb, lens, cfg, p, names, need):
    """descriptor + buffers of one stream for tavsr_tailored_layer_fwd; leaves ``ns`` exactly as TailoredStreamFn.forward does
    (the Python backward runs on it).  Draws the stream's dropout tokens in the order of the Python sequencing."""
    from ._lib import TailoredStreamDesc, lib
    import ctypes as C
    B, T, D = x.shape
    M, H = B * T, cfg["heads"]
    N1 = p["feed_forward.w_1.weight"].shape[0]
    ua = bool(cfg["use_attn"])
    pd, pa = cfg.get("p", 0.0), cfg.get("p_att", 0.0)
    x2d = x.contiguous().view(M, D)
    E = lambda *s: ops.empty(*s, like=x)
    Mp = (M + 127) // 128 * 128
    d = TailoredStreamDesc()
    d.B, d.T, d.D, d.H, d.ffn_units, d.ffn_act, d.save, d.use_attn = B, T, D, H, N1, ops.ACT[cfg["ffn_act"]], int(need), int(ua)
    d.p_drop, d.p_att, d.coeff = pd, pa, cfg.get("coeff", 1.0)
    bn = "norm_mha" if ua else "norm_cgmlp"
    fields = [("ffm_ln_w", "norm_ff_macaron.weight"), ("ffm_ln_b", "norm_ff_macaron.bias"), ("ffm_w1", "feed_forward_macaron.w_1.weight"),
              ("ffm_b1", "feed_forward_macaron.w_1.bias"), ("ffm_w2", "feed_forward_macaron.w_2.weight"),
              ("ffm_b2", "feed_forward_macaron.w_2.bias"), ("br_ln_w", bn + ".weight"), ("br_ln_b", bn + ".bias"),
              ("ff_ln_w", "norm_ff.weight"), ("ff_ln_b", "norm_ff.bias"), ("ff_w1", "feed_forward.w_1.weight"), ("ff_b1", "feed_forward.w_1.bias"),
              ("ff_w2", "feed_forward.w_2.weight"), ("ff_b2", "feed_forward.w_2.bias"), ("final_ln_w", "norm_final.weight"),
              ("final_ln_b", "norm_final.bias")]
    if ua:
        fields += [("wq", "attn.linear_q.weight"), ("bq", "attn.linear_q.bias"), ("wk", "attn.linear_k.weight"), ("bk", "attn.linear_k.bias"),
                   ("wv", "attn.linear_v.weight"), ("bv", "attn.linear_v.bias"), ("wpos", "attn.linear_pos.weight"), ("pos_u", "attn.pos_bias_u"),
                   ("pos_v", "attn.pos_bias_v"), ("wo", "attn.linear_out.weight"), ("bo", "attn.linear_out.bias")]
        sizes = (M * N1, M * D, B * H * T * ops.pad4(T), M * D, M * N1, M * D)
        rates = (pd, pd, pa, pd, pd, pd)
    else:
        C2 = p["cgmlp.channel_proj1.0.weight"].shape[0]
        Cn = C2 // 2
        d.cg_units, d.cg_kernel = C2, p["cgmlp.csgu.conv.weight"].shape[-1]
        fields += [("cg_w1", "cgmlp.channel_proj1.0.weight"), ("cg_b1", "cgmlp.channel_proj1.0.bias"), ("csgu_ln_w", "cgmlp.csgu.norm.weight"),
                   ("csgu_ln_b", "cgmlp.csgu.norm.bias"), ("csgu_cw", "cgmlp.csgu.conv.weight"), ("csgu_cb", "cgmlp.csgu.conv.bias"),
                   ("cg_w2", "cgmlp.channel_proj2.weight"), ("cg_b2", "cgmlp.channel_proj2.bias")]
        sizes = (M * N1, M * D, M * Cn, M * D, M * N1, M * D)
        rates = (pd,) * 6
    for f, n in fields:
        setattr(d, f, ops._addr(p[n]))
    d.x, d.pos_emb, d.lens = ops._addr(x2d), ops._addr(pos_emb), ops._addr(lens)
    toks = [ops._new_token(r, n, x.device) if r and r > 0.0 else None for r, n in zip(rates, sizes)]
    for j, t in enumerate(toks):
        if t is not None:
            d.drop_off[j] = t[1]
            d.seed = ops._addr(t[2])
    b = {k: E(M, D) for k in ("x1", "n_br", "x2", "x3", "y")}
    if ua:
        b.update(qkv=E(M, 3 * D), pp=E(2 * T - 1, D), cx=E(M, D), lse=E(B * H, T))
    else:
        b.update(g=E(M, C2), u=E(M, Cn), g_mean=E(M), g_rstd=E(M))
        if need:
            b.update(g_z=E(M, C2), gn=E(M, Cn), conv=E(M, Cn))
    if need:
        b.update(ffm_n=E(M, D), ff_n=E(M, D), ffm_z=E(Mp, N1)[:M], ffm_h=E(Mp, N1)[:M], ff_z=E(Mp, N1)[:M], ff_h=E(Mp, N1)[:M])
        b.update({k: E(M) for k in ("ffm_mean", "ffm_rstd", "br_mean", "br_rstd", "ff_mean", "ff_rstd", "fin_mean", "fin_rstd")})
    for k, t in b.items():
        setattr(d, k, ops._addr(t))
    key = (B, T, D, H, N1, int(ua), d.cg_units, int(need))
    nws = _TS_WS.get(key)
    if nws is None:
        fn = lib().tavsr_tailored_stream_ws
        fn.restype = C.c_int64
        nws = _TS_WS[key] = int(fn(C.byref(d)))
    ws = ops.empty(max(nws, 4), like=x)
    d.ws, d.ws_floats = ops._addr(ws), nws
    g = b.get
    if ua:
        br = (g("br_mean"), g("br_rstd"), b["n_br"], b["qkv"], b["pp"], None, None, b["cx"], (b["lse"], toks[2]), None, toks[3])
    else:
        br = (g("br_mean"), g("br_rstd"), b["n_br"], b["g"], g("g_z"), g("gn"), b["g_mean"], b["g_rstd"], b["u"], g("conv"), toks[2], toks[3])
    sv = {"ffm": (x2d, g("ffm_mean"), g("ffm_rstd"), g("ffm_n"), g("ffm_z"), g("ffm_h"), toks[0], toks[1]), "br": br,
          "ff": (b["x2"], g("ff_mean"), g("ff_rstd"), g("ff_n"), g("ff_z"), g("ff_h"), toks[4], toks[5]),
          "final": (b["x3"], g("fin_mean"), g("fin_rstd")), "x1": b["x1"]}
    ns.sv, ns.cfg, ns.p, ns.names, ns.pos_emb, ns.shape, ns.lens = sv, cfg, p, names, pos_emb, (B, T, D), lens
    return d, ws, b["y"].view(B, T, D)


def _tailored_c_forward(ca, cv, audio, apos, alens, cfg_a, video, vpos, vlens, cfg_v, pa, pv, names_a, names_v, need):
    """TailoredLayerFn.forward as one C call (tavsr_tailored_layer_fwd): the video stream on the forked queue beside the audio
    stream; every buffer comes from the calling stream's pool (the call joins the queues before it returns)."""
    from ._lib import TailoredLayerDesc, check, lib
    import ctypes as C
    dv, wsv, yv = _ts_c_desc(cv, video, vpos, vlens, cfg_v, pv, names_v, need)      # (the Python sequencing draws the video stream's tokens first)
    da, wsa, ya = _ts_c_desc(ca, audio, apos, alens, cfg_a, pa, names_a, need)
    main = torch.cuda.current_stream()
    side = ops.branch_stream(main) if ops.forks_enabled() else main
    ev = ops.branch_events(main)
    L = TailoredLayerDesc()
    L.audio, L.video = C.pointer(da), C.pointer(dv)
    L.stream2, L.ev_fork, L.ev_join = side.cuda_stream, ev[0].cuda_event, ev[1].cuda_event
    check(lib().tavsr_tailored_layer_fwd(C.byref(L), C.c_void_p(main.cuda_stream)), "tavsr_tailored_layer_fwd")
    return ya, yv


class TailoredLayerFn(torch.autograd.Function):
    """Both modality streams of one TailoredEncoderLayer as ONE autograd node, so that the video stream can run on the
    forked stream beside the audio stream, forward and backward (two separate nodes would leave the stream hand-over
    to autograd).  P = shared (14) + audio-only + video-only parameters; the shared FFNs / norms receive the sum of
    the two streams' gradients."""

    @staticmethod
    def forward(ctx, audio, apos, alens, cfg_a, video, vpos, vlens, cfg_v, *P):
        import types
        need = _note_ctx(ctx)
        cfg_a, cfg_v = dict(cfg_a, need_bwd=need), dict(cfg_v, need_bwd=need)
        names_a, names_v = tailored_stream_param_names(cfg_a["use_attn"]), tailored_stream_param_names(cfg_v["use_attn"])
        na, nv = len(names_a), len(names_v)
        ns = len(TS_SHARED)
        Pa, Pv = P[:na], P[:ns] + P[na: na + nv - ns]
        ca, cv = types.SimpleNamespace(), types.SimpleNamespace()
        pa_, pv_ = dict(zip(names_a, Pa)), dict(zip(names_v, Pv))
        if _ts_c_ok(audio, cfg_a, pa_) and _ts_c_ok(video, cfg_v, pv_):
            ya, yv = _tailored_c_forward(ca, cv, audio, apos, alens, cfg_a, video, vpos, vlens, cfg_v, pa_, pv_, names_a, names_v, need)
            ctx.ca, ctx.cv, ctx.n = ca, cv, (ns, na, nv)
            return ya, yv
        br = ops.BranchScope(audio.is_cuda)
        with br:
            yv = TailoredStreamFn.forward(cv, video, vpos, vlens, cfg_v, *Pv)
        ya = TailoredStreamFn.forward(ca, audio, apos, alens, cfg_a, *Pa)
        br.join()
        ctx.ca, ctx.cv, ctx.n = ca, cv, (ns, na, nv)
        return ya, yv

    @staticmethod
    @guarded
    def backward(ctx, dya, dyv):
        ns, na, nv = ctx.n
        br = ops.BranchScope(dya.is_cuda)
        dyv = dyv.contiguous()
        # (the video stream's saved state holds tensors of the MAIN stream's allocator pool - layer 0: the stream's input from
        # the embedding - and its backward drops that state while its launches are still queued on the forked stream: every
        # such tensor was record_stream-ed when a launch of the body took its address, _lib.py rule 1)
        with br:
            gv = TailoredStreamFn.backward(ctx.cv, dyv)
        ga = TailoredStreamFn.backward(ctx.ca, dya)
        br.join()
        dxa, dxv = ga[0], gv[0]
        Ga, Gv = ga[4:], gv[4:]
        shared = ops.multi_add_([a.contiguous() for a in Ga[:ns]], [b.contiguous() for b in Gv[:ns]])   # one launch
        ctx.ca = ctx.cv = None
        return (dxa, None, None, None, dxv, None, None, None, *shared, *Ga[ns:], *Gv[ns:])


class _Ctx:
    """stand-in for an autograd ctx when a Function's forward / backward bodies are reused inside another node"""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


# Run two independent nodes side by side?  TAVSR_FRONT_PAIR=0 keeps them as two nodes on one stream (A/B switch).
import os as _os
FRONT_PAIR = True


class FrontendPairFn(torch.autograd.Function):
    """The lip front-end (Conv3dResNet18) and the audio embedding (Conv2dSubsamplingWOPosEnc) of the AV model as ONE
    autograd node: the two are independent until the streams are aligned (src/models/avsr_espnet_model.py:400-434), so the
    audio embedding's launches go to the forked stream and run beside the front-end's HBM-bound BatchNorm phases, forward
    and backward (two separate nodes would be serialised by autograd on one stream).
    P = front-end parameters (cfg["names"] order) + (conv.0.weight, conv.0.bias, conv.2.weight, conv.2.bias, out.weight,
    out.bias) of the audio embedding."""

    @staticmethod
    def forward(ctx, video, cfg, audio, *P):
        from .functional import Conv2dSubsamplingFn
        npv = len(cfg["names"])
        cv, ca = _Ctx(), _Ctx()
        br = ops.BranchScope(video.is_cuda)
        with br:
            ya = Conv2dSubsamplingFn.forward(ca, audio, *P[npv:], 1.0)
        yv = VisualFrontendFn.forward(cv, video, cfg, *P[:npv])
        br.join()
        ctx.cv, ctx.ca = cv, ca
        return yv, ya

    @staticmethod
    @guarded
    def backward(ctx, dyv, dya):
        from .functional import Conv2dSubsamplingFn
        br = ops.BranchScope(dyv.is_cuda)
        dya = dya.contiguous()
        with br:
            ga = Conv2dSubsamplingFn.backward(ctx.ca, dya)
        gv = VisualFrontendFn.backward(ctx.cv, dyv)
        br.join()
        ctx.cv = ctx.ca = None
        return (None, None, None, *gv[2:], *ga[1:7])


# ------------------------------------------------------------------------------------------------
# AdaptiveAudioVisualFusion, merge_method="learned_ave" (src/audiovisual_fusion/adaptive_audiovisual_fusion.py:137-205):
# attention pooling of each stream under its own mask -> softmax over {audio, video} -> weighted sum ->
# PositionwiseFeedForward (no residual) -> LayerNorm
# ------------------------------------------------------------------------------------------------
FUSION_PARAM_NAMES = ("acoustic_pooling_proj.weight", "visual_pooling_proj.weight", "acoustic_pooling_proj.bias",
                      "visual_pooling_proj.bias", "acoustic_weight_proj.weight", "visual_weight_proj.weight",
                      "acoustic_weight_proj.bias", "visual_weight_proj.bias",
                      "audiovisual_layer.w_1.weight", "audiovisual_layer.w_1.bias", "audiovisual_layer.w_2.weight",
                      "audiovisual_layer.w_2.bias", "norm_final.weight", "norm_final.bias")


class FusionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, audio, video, alens, vlens, cfg, *P):
        """cfg["mode"]: "learned" (P = 8 pooling / weight projections + FFN + norm), "const" (cfg["w"] = (w_audio, w_video):
        merge_method fixed_ave, or the dropped acoustic branch of learned_ave) or "concat" (P = FFN over 2 D + norm)."""
        B, T, D = audio.shape
        a2, v2 = audio.contiguous().view(B * T, D), video.contiguous().view(B * T, D)
        mode = cfg.get("mode", "learned")
        if cfg.get("drop_acoustic"):      # constant weights (0, 1): the fused stream is the video stream
            mode, cfg = "const", dict(cfg, w=(0.0, 1.0))
        nm = len(P) - 6                   # merge projections in front of (w1, b1, w2, b2, norm weight, norm bias)
        mp = list(P[:nm])
        w1, b1, w2, b2, lw, lb = P[nm:nm + 6]
        score = pooled = wts = None
        if mode == "const":
            wa, wv = cfg["w"]
            m = v2 if (wa == 0.0 and wv == 1.0) else ops.axpby(a2, v2, float(wa), float(wv))
        elif mode == "concat":
            m = ops.empty(B * T, 2 * D, like=a2)
            ops.copy2d(a2, m[:, :D])
            ops.copy2d(v2, m[:, D:])
        else:
            score, pooled, wts, m = ops.merge_fwd(a2, v2, alens, mp[:8], B, T, lens2=vlens)
        h, z = ops.linear(m, w1, b1, act=cfg["act"], save_z=True)
        t_in = _drop_(h, cfg.get("p", 0.0))       # PositionwiseFeedForward's inner dropout (the fusion has no outer one)
        y2 = ops.linear(h, w2, b2)
        out, mean, rstd = ops.layernorm_fwd(y2, lw, lb, EPS_ESPNET)
        ctx.sv = (a2, v2, score, pooled, wts, m, h, z, y2, mean, rstd, t_in)
        ctx.P, ctx.cfg, ctx.lens, ctx.dims, ctx.mode, ctx.nm = P, cfg, (alens, vlens), (B, T, D), mode, nm
        cfg["_last_w"] = wts
        return out.view(B, T, -1)

    @staticmethod
    @guarded
    def backward(ctx, dy):
        a2, v2, score, pooled, wts, m, h, z, y2, mean, rstd, t_in = ctx.sv
        P, cfg, mode, nm = ctx.P, ctx.cfg, ctx.mode, ctx.nm
        B, T, D = ctx.dims
        alens, vlens = ctx.lens
        mp = list(P[:nm])
        w1, b1, w2, b2, lw, lb = P[nm:nm + 6]
        dy2, glw, glb = ops.layernorm_bwd(dy.contiguous().view(B * T, -1), y2, mean, rstd, lw)
        gw2, gb2 = ops.linear_dw(dy2, h, bias_grad=True)
        if t_in is None:
            dz = ops.linear_dx(dy2, w2, DZ=z, dact=cfg["act"])
        else:
            dh = ops.linear_dx(dy2, w2)
            dz = ops.dropout_act_bwd(dh, z, cfg["act"], t_in, out=dh)
        gw1, gb1 = ops.linear_dw(dz, m, bias_grad=True)
        dm = ops.linear_dx(dz, w1)
        mg = [None] * nm
        if mode == "const":               # constant weights: the merge projections (if any) took no part
            wa, wv = cfg["w"]
            da = torch.zeros_like(dm) if wa == 0.0 else ops.axpby(dm, None, float(wa), 0.0)
            dv = dm if wv == 1.0 else ops.axpby(dm, None, float(wv), 0.0)
        elif mode == "concat":
            da, dv = ops.empty(B * T, D, like=dm), ops.empty(B * T, D, like=dm)
            ops.copy2d(dm[:, :D], da)
            ops.copy2d(dm[:, D:], dv)
        else:
            da, dv, mg = ops.merge_bwd(dm, a2, v2, alens, mp, score, pooled, wts, B, T, lens2=vlens)
            mg = [g.view_as(q) for g, q in zip(mg, mp)]
        ctx.sv = None
        return (da.view(B, T, D), dv.view(B, T, D), None, None, None, *mg, gw1, gb1, gw2, gb2, glw, glb)
