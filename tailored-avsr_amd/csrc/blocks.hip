// Block-level entry points of the C ABI (include/tavsr.h): whole modules of the reference model as ONE call each, sequenced in C
// over the primitive entry points - what tavsr/functional*.py otherwise enqueues call by call.  Host code only; same launches,
// same order, same results as the caller-side sequencing.
//   tavsr_cgmlp_fwd / _bwd              espnet ConvolutionalGatingMLP (+ the branch's dropout / residual around it) as called at
//                                       src/encoder/branchformer/encoder_layer.py:213-226 and
//                                       src/encoder/audiovisual/tailored/encoder_layer.py:198-208,246-256
//   tavsr_conv2d_subsample_fwd / _bwd   espnet Conv2dSubsampling / Conv2dSubsamplingWOPosEnc (conv 3x3/2 + ReLU, conv 3x3/2 + ReLU,
//                                       Linear) as called at src/encoder/branchformer/encoder.py:364 and
//                                       src/embedding_for_avsr/default.py:111-162
//   tavsr_workspace_bytes               one workspace query for every descriptor-driven entry point
#include "seq.h"

using namespace tavsr;
using namespace tavsr::seq;

namespace {

// ------------------------------------------------------------------------------------------------ cgMLP
int cgmlp_ok(const tavsr_cgmlp_desc* d, const char* who) {
  TAVSR_REQUIRE(d, TAVSR_EINVAL, "%s: null descriptor", who);
  TAVSR_REQUIRE(d->B > 0 && d->T > 0 && d->D > 0 && d->D % 32 == 0 && d->units > 0 && d->units % 128 == 0 && d->units / 2 <= 1024 &&
                    d->kernel == 31,
                TAVSR_EUNSUPPORTED, "%s: d_model %% 32 == 0, 2C %% 128 == 0, C <= 1024 and kernel 31 only", who);
  return TAVSR_OK;
}

int cgmlp_fwd_seq(const tavsr_cgmlp_desc* d, hipStream_t s, Bump& ws) {
  const int M = d->B * d->T, D = d->D, C2 = d->units, Cn = C2 / 2;
  const bool dry = ws.dry;
  int rc;
  tavsr_gemm_desc g1 = lin(M, C2, D, d->x, D, d->w1, d->b1, d->g, C2);
  g1.act = TAVSR_ACT_GELU;
  if (d->save) g1.Z = d->g_z;
  // the GEMM's epilogue leaves per-row partial sums of its 64-column tiles: the CSGU's LayerNorm statistics without a launch
  float* rowstat = ws.take((int64_t)M * (C2 / 64) * 2);
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "cgmlp_fwd: workspace too small");
  g1.rowstat = rowstat;
  if ((rc = run_gemm(g1, ws, s))) return rc;
  if (!dry && (rc = tavsr_csgu_fwd(d->g, C2, d->ln_w, d->ln_b, 1e-12f, d->cw, d->cb, d->u, d->save ? d->gn : nullptr, d->save ? d->conv : nullptr,
                                   d->g_mean, d->g_rstd, d->p_drop, d->seed, d->off_u, d->B, d->T, Cn, d->kernel, rowstat, (tavsr_stream_t)s)))
    return rc;
  tavsr_gemm_desc g2 = lin(M, D, Cn, d->u, Cn, d->w2, d->b2, d->out, D);
  g2.alpha = d->alpha;
  if (d->res) { g2.R = d->res; g2.ldr = D; }
  if (d->p_out > 0.f) { g2.drop_p = d->p_out; g2.drop_seed = d->seed; g2.drop_offset = d->off_out; }
  return run_gemm(g2, ws, s);
}

int wgrad(const float* dy, int64_t lddy, const float* x, int64_t ldx, int rows, int N, int K, float alpha, float* out, float* gb, Bump& ws,
          hipStream_t s) {
  tavsr_gemm_desc g;
  memset(&g, 0, sizeof g);
  g.M = N; g.N = K; g.K = rows;
  g.a_kmajor = g.b_kmajor = 1;
  g.A = dy; g.lda = lddy; g.B = x; g.ldb = ldx; g.C = out; g.ldc = K;
  g.nb1 = g.nb2 = 1;
  g.alpha = alpha;
  g.a_rowsum = gb;
  return run_gemm(g, ws, s);
}

int cgmlp_bwd_seq(const tavsr_cgmlp_bwd_desc* b, hipStream_t s, Bump& ws) {
  const tavsr_cgmlp_desc* d = b->fwd;
  const int M = d->B * d->T, D = d->D, C2 = d->units, Cn = C2 / 2;
  const bool dry = ws.dry;
  int rc;
  // gradient of the dropped, scaled branch output: alpha * mask(dy) / keep
  const float* dyd = b->dy;
  if (d->p_out > 0.f || d->alpha != 1.f) {
    float* t = ws.take((int64_t)M * D);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "cgmlp_bwd: workspace too small");
    if (!dry) {
      if (d->p_out > 0.f) {
        if ((rc = tavsr_dropout(b->dy, t, (int64_t)M * D, d->p_out, d->seed, d->off_out, (tavsr_stream_t)s))) return rc;
        if (d->alpha != 1.f && (rc = tavsr_axpby(t, nullptr, d->alpha, 0.f, t, (int64_t)M * D, (tavsr_stream_t)s))) return rc;
      } else if ((rc = tavsr_axpby(b->dy, nullptr, d->alpha, 0.f, t, (int64_t)M * D, (tavsr_stream_t)s))) {
        return rc;
      }
    }
    dyd = t;
  }
  if ((rc = wgrad(dyd, D, d->u, Cn, M, D, Cn, 1.f, b->g_w2, b->g_b2, ws, s))) return rc;
  float* du = ws.take((int64_t)M * Cn);
  tavsr_gemm_desc g = lin(M, Cn, D, dyd, D, d->w2, nullptr, du, Cn);
  g.b_kmajor = 1; g.ldb = Cn;
  if (d->p_drop > 0.f) { g.drop_p = d->p_drop; g.drop_seed = d->seed; g.drop_offset = d->off_u; }
  if ((rc = run_gemm(g, ws, s))) return rc;
  float* dg = ws.take((int64_t)M * C2);
  float* dgn = ws.take((int64_t)M * Cn);
  float* cws = ws.take(tavsr_dwconv_gate_bwd_ws(d->B, d->T, Cn, d->kernel));
  float* lws = ws.take(tavsr_layernorm_bwd_ws(M, Cn));
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "cgmlp_bwd: workspace too small");
  if (!dry) {
    if ((rc = tavsr_dwconv_gate_bwd_act(du, d->gn, d->g, C2, d->conv, d->cw, dg, C2, dgn, b->g_cw, b->g_cb, 0, cws, d->B, d->T, Cn, d->kernel,
                                        d->g_z, C2, TAVSR_ACT_GELU, (tavsr_stream_t)s)))
      return rc;
    if ((rc = tavsr_layernorm_bwd_act(dgn, Cn, d->g + Cn, C2, d->g_mean, d->g_rstd, d->ln_w, dg + Cn, C2, b->g_ln_w, b->g_ln_b, 0, lws, M, Cn,
                                      d->g_z + Cn, C2, TAVSR_ACT_GELU, (tavsr_stream_t)s)))
      return rc;
  }
  if ((rc = wgrad(dg, C2, d->x, D, M, C2, D, 1.f, b->g_w1, b->g_b1, ws, s))) return rc;
  tavsr_gemm_desc g1 = lin(M, D, C2, dg, C2, d->w1, nullptr, b->dx, D);
  g1.b_kmajor = 1; g1.ldb = D;
  return run_gemm(g1, ws, s);
}

// ------------------------------------------------------------------------------------------------ Conv2dSubsampling
struct SubDims { int T1, F1, T2, F2; };
SubDims sub_dims(const tavsr_subsample_desc* d) {
  SubDims q;
  q.T1 = (d->T - 3) / 2 + 1; q.F1 = (d->F - 3) / 2 + 1;
  q.T2 = (q.T1 - 3) / 2 + 1; q.F2 = (q.F1 - 3) / 2 + 1;
  return q;
}

int sub_ok(const tavsr_subsample_desc* d, const char* who) {
  TAVSR_REQUIRE(d, TAVSR_EINVAL, "%s: null descriptor", who);
  TAVSR_REQUIRE(d->B > 0 && d->T >= 7 && d->F >= 7 && d->C > 0 && d->odim > 0, TAVSR_EINVAL, "%s: B > 0, T >= 7, F >= 7 (espnet check_short_utt)", who);
  const SubDims q = sub_dims(d);
  TAVSR_REQUIRE(d->C % 64 == 0 && ((int64_t)d->B * q.T2 * q.F2) % 32 == 0 && d->odim % 4 == 0, TAVSR_EUNSUPPORTED,
                "%s: channels %% 64 == 0 and B*T2*F2 %% 32 == 0 required (implicit second convolution); callers keep im2col + GEMM", who);
  return TAVSR_OK;
}

int sub_fwd_seq(const tavsr_subsample_desc* d, hipStream_t s, Bump& ws) {
  const SubDims q = sub_dims(d);
  const int B = d->B, C = d->C;
  const bool dry = ws.dry;
  int rc;
  if (!dry) {
    if ((rc = tavsr_conv1_fwd(d->x, d->w1, d->b1, d->y1, B, d->T, d->F, C, (tavsr_stream_t)s))) return rc;
    // torch (co, ci, kh, kw) -> (co, kh, kw, ci): the channels-last patch order of the implicit GEMM
    if ((rc = tavsr_transpose_inner(d->w2, d->w2r, C, C, 9, 0, (tavsr_stream_t)s))) return rc;
  }
  {
    const int64_t Mo = (int64_t)B * q.T2 * q.F2;
    tavsr_gemm_desc g = lin((int)Mo, C, 9 * C, d->y1, C, d->w2r, d->b2, d->y2, C);
    g.act = TAVSR_ACT_RELU;
    g.conv_mode = 1; g.conv_H = q.T1; g.conv_W = q.F1; g.conv_C = C; g.conv_stride = 2; g.conv_taps = 90;
    g.conv_zero = d->zero_page;
    if ((rc = run_gemm(g, ws, s))) return rc;
  }
  // the out Linear consumes (c * F2 + f); its weight is re-indexed to (f * C + c) instead of transposing activations
  if (!dry && (rc = tavsr_transpose_inner(d->wo, d->wor, d->odim, C, q.F2, 0, (tavsr_stream_t)s))) return rc;
  tavsr_gemm_desc go = lin(B * q.T2, d->odim, q.F2 * C, d->y2, (int64_t)q.F2 * C, d->wor, d->bo, d->out, d->odim);
  go.alpha = d->xscale;
  return run_gemm(go, ws, s);
}

int sub_bwd_seq(const tavsr_subsample_bwd_desc* b, hipStream_t s, Bump& ws) {
  const tavsr_subsample_desc* d = b->fwd;
  const SubDims q = sub_dims(d);
  const int B = d->B, C = d->C, odim = d->odim;
  const int64_t M2 = (int64_t)B * q.T2, K2 = (int64_t)q.F2 * C, Mo = M2 * q.F2;
  const bool dry = ws.dry;
  int rc;
  // wgrad_beside: the two weight-gradient GEMMs (0.09 + 0.67 ms at batch 32) and their re-layouts have no reader inside the backward pass;
  // on stream2 they run beside the dgrad GEMM, col2im and the first convolution's backward instead of in front of them.  NOT joined here.
  hipStream_t sw = s;
  if (!dry && b->wgrad_beside && b->stream2 && (hipStream_t)b->stream2 != s) sw = (hipStream_t)b->stream2;
  float* gwor = ws.take((int64_t)odim * K2);
  // dz2 = (dout wor) * xscale * relu'(y2)
  float* dz2 = ws.take(Mo * C);
  {
    tavsr_gemm_desc g = lin((int)M2, (int)K2, odim, b->dout, odim, d->wor, nullptr, dz2, K2);
    g.b_kmajor = 1; g.ldb = K2; g.alpha = d->xscale; g.DZ = d->y2; g.dact = TAVSR_ACT_RELU;
    if ((rc = run_gemm(g, ws, s))) return rc;
  }
  if (sw != s) {
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)b->ev_fork, s));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(sw, (hipEvent_t)b->ev_fork, 0));
  }
  if ((rc = wgrad(b->dout, odim, d->y2, K2, (int)M2, odim, (int)K2, d->xscale, gwor, b->g_bo, ws, sw))) return rc;
  float* gw2r = ws.take((int64_t)C * 9 * C);
  {
    tavsr_gemm_desc g;
    memset(&g, 0, sizeof g);
    g.M = C; g.N = 9 * C; g.K = (int)Mo;
    g.a_kmajor = g.b_kmajor = 1;
    g.A = dz2; g.lda = C; g.B = d->y1; g.ldb = C; g.C = gw2r; g.ldc = 9 * C;
    g.nb1 = g.nb2 = 1;
    g.alpha = 1.f;
    g.a_rowsum = b->g_b2;
    g.conv_mode = 2; g.conv_H = q.T1; g.conv_W = q.F1; g.conv_C = C; g.conv_stride = 2; g.conv_taps = 90;
    g.conv_zero = d->zero_page;
    if ((rc = run_gemm(g, ws, sw))) return rc;
  }
  if (!dry) {
    if ((rc = tavsr_transpose_inner(gwor, b->g_wo, odim, q.F2, C, 0, (tavsr_stream_t)sw))) return rc;
    if ((rc = tavsr_transpose_inner(gw2r, b->g_w2, C, 9, C, 0, (tavsr_stream_t)sw))) return rc;
  }
  float* dcol = ws.take(Mo * 9 * C);
  {
    tavsr_gemm_desc g = lin((int)Mo, 9 * C, C, dz2, C, d->w2r, nullptr, dcol, 9 * C);
    g.b_kmajor = 1; g.ldb = 9 * C;
    if ((rc = run_gemm(g, ws, s))) return rc;
  }
  float* dz1 = ws.take((int64_t)B * q.T1 * q.F1 * C);
  float* cws = ws.take(tavsr_conv1_bwd_ws(B, d->T, d->F, C));
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "conv2d_subsample_bwd: workspace too small");
  if (!dry) {
    if ((rc = tavsr_col2im3x3s2_relu(dcol, d->y1, dz1, B, q.T1, q.F1, C, (tavsr_stream_t)s))) return rc;
    if ((rc = tavsr_conv1_bwd(dz1, d->x, b->g_w1, b->g_b1, 0, cws, B, d->T, d->F, C, (tavsr_stream_t)s))) return rc;
  }
  return TAVSR_OK;
}

template <class F>
int64_t dry_floats(F&& f) {
  Bump ws{nullptr, 0, 0, true, false};
  if (f(ws)) return 0;
  return ws.used;
}

}  // namespace

extern "C" int64_t tavsr_cgmlp_ws(const tavsr_cgmlp_desc* d) {
  if (cgmlp_ok(d, "cgmlp_ws")) return 0;
  return dry_floats([&](Bump& ws) { return cgmlp_fwd_seq(d, nullptr, ws); });
}

extern "C" int tavsr_cgmlp_fwd(const tavsr_cgmlp_desc* d, tavsr_stream_t stream) {
  int rc = cgmlp_ok(d, "cgmlp_fwd");
  if (rc) return rc;
  TAVSR_REQUIRE(d->x && d->w1 && d->b1 && d->ln_w && d->ln_b && d->cw && d->cb && d->w2 && d->b2 && d->g && d->u && d->out && d->g_mean &&
                    d->g_rstd && d->ws,
                TAVSR_EINVAL, "cgmlp_fwd: null buffer");
  TAVSR_REQUIRE(!d->save || (d->g_z && d->gn && d->conv), TAVSR_EINVAL, "cgmlp_fwd: save = 1 needs g_z, gn and conv");
  TAVSR_REQUIRE((d->p_drop == 0.f && d->p_out == 0.f) || d->seed, TAVSR_EINVAL, "cgmlp_fwd: dropout needs a device seed");
  TAVSR_REQUIRE(dry_floats([&](Bump& w) { return cgmlp_fwd_seq(d, nullptr, w); }) <= d->ws_floats, TAVSR_EINVAL,
                "cgmlp_fwd: workspace too small (tavsr_cgmlp_ws)");
  Bump ws{d->ws, d->ws_floats, 0, false, false};
  return cgmlp_fwd_seq(d, (hipStream_t)stream, ws);
}

extern "C" int64_t tavsr_cgmlp_bwd_ws(const tavsr_cgmlp_bwd_desc* b) {
  if (!b || cgmlp_ok(b->fwd, "cgmlp_bwd_ws")) return 0;
  return dry_floats([&](Bump& ws) { return cgmlp_bwd_seq(b, nullptr, ws); });
}

extern "C" int tavsr_cgmlp_bwd(const tavsr_cgmlp_bwd_desc* b, tavsr_stream_t stream) {
  TAVSR_REQUIRE(b && b->fwd, TAVSR_EINVAL, "cgmlp_bwd: null descriptor");
  int rc = cgmlp_ok(b->fwd, "cgmlp_bwd");
  if (rc) return rc;
  TAVSR_REQUIRE(b->fwd->save, TAVSR_EINVAL, "cgmlp_bwd: the forward call must have kept its state (save = 1)");
  TAVSR_REQUIRE(b->dy && b->dx && b->g_w1 && b->g_b1 && b->g_ln_w && b->g_ln_b && b->g_cw && b->g_cb && b->g_w2 && b->g_b2 && b->ws, TAVSR_EINVAL,
                "cgmlp_bwd: null buffer");
  TAVSR_REQUIRE(dry_floats([&](Bump& w) { return cgmlp_bwd_seq(b, nullptr, w); }) <= b->ws_floats, TAVSR_EINVAL,
                "cgmlp_bwd: workspace too small (tavsr_cgmlp_bwd_ws)");
  Bump ws{b->ws, b->ws_floats, 0, false, false};
  return cgmlp_bwd_seq(b, (hipStream_t)stream, ws);
}

extern "C" int64_t tavsr_conv2d_subsample_ws(const tavsr_subsample_desc* d) {
  if (sub_ok(d, "conv2d_subsample_ws")) return 0;
  return dry_floats([&](Bump& ws) { return sub_fwd_seq(d, nullptr, ws); });
}

extern "C" int tavsr_conv2d_subsample_fwd(const tavsr_subsample_desc* d, tavsr_stream_t stream) {
  int rc = sub_ok(d, "conv2d_subsample_fwd");
  if (rc) return rc;
  TAVSR_REQUIRE(d->x && d->w1 && d->b1 && d->w2 && d->b2 && d->wo && d->bo && d->y1 && d->y2 && d->w2r && d->wor && d->out && d->zero_page &&
                    (d->ws || d->ws_floats == 0),
                TAVSR_EINVAL, "conv2d_subsample_fwd: null buffer");
  TAVSR_REQUIRE(dry_floats([&](Bump& w) { return sub_fwd_seq(d, nullptr, w); }) <= d->ws_floats, TAVSR_EINVAL,
                "conv2d_subsample_fwd: workspace too small (tavsr_conv2d_subsample_ws)");
  Bump ws{d->ws, d->ws_floats, 0, false, false};
  return sub_fwd_seq(d, (hipStream_t)stream, ws);
}

extern "C" int64_t tavsr_conv2d_subsample_bwd_ws(const tavsr_subsample_bwd_desc* b) {
  if (!b || sub_ok(b->fwd, "conv2d_subsample_bwd_ws")) return 0;
  return dry_floats([&](Bump& ws) { return sub_bwd_seq(b, nullptr, ws); });
}

extern "C" int tavsr_conv2d_subsample_bwd(const tavsr_subsample_bwd_desc* b, tavsr_stream_t stream) {
  TAVSR_REQUIRE(b && b->fwd, TAVSR_EINVAL, "conv2d_subsample_bwd: null descriptor");
  int rc = sub_ok(b->fwd, "conv2d_subsample_bwd");
  if (rc) return rc;
  TAVSR_REQUIRE(b->dout && b->g_w1 && b->g_b1 && b->g_w2 && b->g_b2 && b->g_wo && b->g_bo && b->ws, TAVSR_EINVAL, "conv2d_subsample_bwd: null buffer");
  TAVSR_REQUIRE(dry_floats([&](Bump& w) { return sub_bwd_seq(b, nullptr, w); }) <= b->ws_floats, TAVSR_EINVAL,
                "conv2d_subsample_bwd: workspace too small (tavsr_conv2d_subsample_bwd_ws)");
  Bump ws{b->ws, b->ws_floats, 0, false, false};
  return sub_bwd_seq(b, (hipStream_t)stream, ws);
}

// One workspace query for every descriptor-driven entry point: bytes (the per-entry queries count floats).
extern "C" int64_t tavsr_workspace_bytes(int32_t kind, const void* desc) {
  if (!desc) return 0;
  int64_t f = 0;
  switch (kind) {
    case TAVSR_WS_GEMM: f = tavsr_gemm_ws((const tavsr_gemm_desc*)desc); break;
    case TAVSR_WS_FFN2: {
      const tavsr_ffn_desc* d = (const tavsr_ffn_desc*)desc;
      f = tavsr_ffn2_ws(d->M, d->D, d->N1);
      break;
    }
    case TAVSR_WS_BF_LAYER_FWD: f = tavsr_branchformer_layer_ws((const tavsr_bf_layer_desc*)desc); break;
    case TAVSR_WS_BF_LAYER_BWD: f = tavsr_branchformer_layer_bwd_ws((const tavsr_bf_layer_bwd_desc*)desc); break;
    case TAVSR_WS_CGMLP_FWD: f = tavsr_cgmlp_ws((const tavsr_cgmlp_desc*)desc); break;
    case TAVSR_WS_CGMLP_BWD: f = tavsr_cgmlp_bwd_ws((const tavsr_cgmlp_bwd_desc*)desc); break;
    case TAVSR_WS_SUBSAMPLE_FWD: f = tavsr_conv2d_subsample_ws((const tavsr_subsample_desc*)desc); break;
    case TAVSR_WS_SUBSAMPLE_BWD: f = tavsr_conv2d_subsample_bwd_ws((const tavsr_subsample_bwd_desc*)desc); break;
    default: return -1;
  }
  return f * (int64_t)sizeof(float);
}
