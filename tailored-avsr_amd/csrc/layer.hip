// One Branchformer encoder layer forward sequenced in C (tavsr_branchformer_layer_fwd, include/tavsr.h): the launches that
// tavsr/functional.py:BranchformerLayerFn.forward enqueues through ~45 Python-level calls, as one C call over the same entry
// points - for un-captured (ragged) training loops, whose step is host-bound on that sequencing.  Host code only.
// Reference: src/encoder/branchformer/encoder_layer.py:153-321 (MyBranchformerEncoderLayer.forward).
#include <functional>

#include "seq.h"

using namespace tavsr;

namespace {

using namespace tavsr::seq;

tavsr_ffn_desc ffn(const tavsr_bf_layer_desc* d, const float* x, const float* ln_w, const float* ln_b, const float* w1,
                   const float* b1, const float* w2, const float* b2, float* y, float* n, float* mean, float* rstd, float* z,
                   float* h, uint64_t off_in, uint64_t off_out) {
  tavsr_ffn_desc f;
  memset(&f, 0, sizeof f);
  f.M = d->B * d->T; f.D = d->D; f.N1 = d->ffn_units; f.act = d->ffn_act;
  f.scale = 0.5f; f.eps = 1e-12f;
  f.x = x; f.ldx = d->D;
  f.ln_w = ln_w; f.ln_b = ln_b; f.w1 = w1; f.b1 = b1; f.w2 = w2; f.b2 = b2;
  f.y = y;
  if (d->save) { f.n_out = n; f.mean = mean; f.rstd = rstd; f.z = z; f.h = h; }
  f.p_drop = d->p_drop; f.seed = d->seed; f.offset_in = off_in; f.offset_out = off_out;
  f.ln2_eps = 1e-12f;
  return f;
}

int sequence(const tavsr_bf_layer_desc* d, hipStream_t s, Bump& ws) {
  const int M = d->B * d->T, D = d->D, C2 = d->cg_units, Cn = C2 / 2, W = 2 * d->T - 1;
  const bool dry = ws.dry;
  hipStream_t s2 = (hipStream_t)d->stream2;
  int rc;
  // ---- x1 = x + 0.5 dropout(ffn_macaron(norm_ff_macaron(x))); norm_mha(x1), norm_mlp(x1) from the same finishing launch
  {
    tavsr_ffn_desc f = ffn(d, d->x, d->ffm_ln_w, d->ffm_ln_b, d->ffm_w1, d->ffm_b1, d->ffm_w2, d->ffm_b2, d->x1, d->ffm_n, d->ffm_mean,
                           d->ffm_rstd, d->ffm_z, d->ffm_h, d->drop_off[0], d->drop_off[1]);
    f.ln2_w[0] = d->mha_ln_w; f.ln2_b[0] = d->mha_ln_b; f.ln2_out[0] = d->n_mha;
    f.ln2_w[1] = d->mlp_ln_w; f.ln2_b[1] = d->mlp_ln_b; f.ln2_out[1] = d->n_mlp;
    if (d->save) { f.ln2_mean = d->br_mean; f.ln2_rstd = d->br_rstd; }
    f.ws_floats = tavsr_ffn2_ws(M, D, d->ffn_units);
    f.ws = ws.take(f.ws_floats);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small");
    if (!dry && (rc = tavsr_ffn2_fwd(&f, (tavsr_stream_t)s))) return rc;
  }
  // ---- attention branch on the second queue (stream2 == stream: one queue, the events order nothing new)
  if (!dry) {
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_fork, s));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(s2, (hipEvent_t)d->ev_fork, 0));
    if ((rc = probe_fork(s2, s, false))) return rc;
    tavsr_gemm_desc q3[3] = {lin(M, D, D, d->n_mha, D, d->wq, d->bq, d->qkv, 3 * D),
                             lin(M, D, D, d->n_mha, D, d->wk, d->bk, d->qkv + D, 3 * D),
                             lin(M, D, D, d->n_mha, D, d->wv, d->bv, d->qkv + 2 * D, 3 * D)};
    if ((rc = tavsr_gemm_grouped(q3, 3, (tavsr_stream_t)s2))) return rc;
  }
  {
    tavsr_gemm_desc gp = lin(W, D, D, d->pos_emb, D, d->wpos, nullptr, d->pp, D);
    if ((rc = run_gemm(gp, ws, s2))) return rc;
  }
  if (!dry) {
    tavsr_attn_desc a;
    memset(&a, 0, sizeof a);
    a.q = d->qkv; a.k = d->qkv + D; a.v = d->qkv + 2 * D;
    a.ldq = a.ldk = a.ldv = 3 * D;
    a.pos = d->pp; a.ldp = D; a.bias_u = d->pos_u; a.bias_v = d->pos_v; a.klens = d->lens;
    a.B = d->B; a.H = d->H; a.T1 = a.T2 = d->T; a.dk = D / d->H;
    a.scale = 1.f / sqrtf((float)a.dk);
    a.p_drop = d->p_att; a.seed_dev = d->seed; a.drop_offset = d->drop_off[2];
    if ((rc = tavsr_attn_fwd(&a, d->cx, D, d->lse, (tavsr_stream_t)s2))) return rc;
  }
  // the merge's four row dots (pooling / branch-weight projections of both branch outputs) come out of the epilogues of the GEMMs that
  // store those outputs (tavsr_gemm_desc.rowdot_*): [M][4 tiles][2] per branch, read by the tail launch instead of the rows themselves
  float* rd1 = ws.take((int64_t)M * 8);
  float* rd2 = ws.take((int64_t)M * 8);
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small");
  {
    tavsr_gemm_desc go = lin(M, D, D, d->cx, D, d->wo, d->bo, d->xa, D);       // x_att = dropout(linear_out(ctx))
    go.drop_p = d->p_drop; go.drop_seed = d->seed; go.drop_offset = d->drop_off[3];
    go.rowstat = rd1; go.rowdot_a = d->merge_p[0]; go.rowdot_b = d->merge_p[4];
    if ((rc = run_gemm(go, ws, s2))) return rc;
  }
  // ---- cgMLP branch on the calling queue
  {
    tavsr_gemm_desc g1 = lin(M, C2, D, d->n_mlp, D, d->cg_w1, d->cg_b1, d->g, C2);
    g1.act = TAVSR_ACT_GELU;
    if (d->save) g1.Z = d->g_z;
    // the GEMM's epilogue leaves per-row partial sums of its 64-column tiles: the CSGU's LayerNorm statistics without a launch
    float* rowstat = Cn <= 1024 ? ws.take((int64_t)M * (C2 / 64) * 2) : nullptr;
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small");
    g1.rowstat = rowstat;
    if ((rc = run_gemm(g1, ws, s))) return rc;
    if (!dry && (rc = tavsr_csgu_fwd(d->g, C2, d->csgu_ln_w, d->csgu_ln_b, 1e-12f, d->csgu_cw, d->csgu_cb, d->u, d->save ? d->gn : nullptr,
                                     d->save ? d->conv : nullptr, d->g_mean, d->g_rstd, d->p_drop, d->seed, d->drop_off[4], d->B,
                                     d->T, Cn, d->cg_kernel, rowstat, (tavsr_stream_t)s)))
      return rc;
    tavsr_gemm_desc g2 = lin(M, D, Cn, d->u, Cn, d->cg_w2, d->cg_b2, d->xm, D);
    g2.drop_p = d->p_drop; g2.drop_seed = d->seed; g2.drop_offset = d->drop_off[5];
    g2.rowstat = rd2; g2.rowdot_a = d->merge_p[1]; g2.rowdot_b = d->merge_p[5];
    if ((rc = run_gemm(g2, ws, s))) return rc;
  }
  if (!dry) {
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_join, s2));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)d->ev_join, 0));
    if ((rc = probe_fork(s2, s, true))) return rc;
    // ---- the tail behind the join as one launch: learned-average merge (d->pooled receives the row dots [4][B*T] its backward
    //      reads), x2 = x1 + coeff dropout(merge_proj(m))
    if ((rc = tavsr_merge_proj_fwd_dots(d->xa, d->xm, d->lens, nullptr, d->merge_p, d->merge_w, d->merge_b, d->x1, d->coeff, d->p_drop,
                                        d->seed, d->drop_off[6], rd1, rd2, d->pooled, d->score, d->wts, d->save ? d->m : nullptr, d->x2,
                                        d->B, d->T, D, (tavsr_stream_t)s)))
      return rc;
  }
  // ---- x3 = x2 + 0.5 dropout(ffn(norm_ff(x2))); y = norm_final(x3)
  {
    tavsr_ffn_desc f = ffn(d, d->x2, d->ff_ln_w, d->ff_ln_b, d->ff_w1, d->ff_b1, d->ff_w2, d->ff_b2, d->x3, d->ff_n, d->ff_mean, d->ff_rstd,
                           d->ff_z, d->ff_h, d->drop_off[7], d->drop_off[8]);
    f.ln2_w[0] = d->final_ln_w; f.ln2_b[0] = d->final_ln_b; f.ln2_out[0] = d->y;
    if (d->save) { f.ln2_mean = d->fin_mean; f.ln2_rstd = d->fin_rstd; }
    f.ws_floats = tavsr_ffn2_ws(M, D, d->ffn_units);
    f.ws = ws.take(f.ws_floats);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small");
    if (!dry && (rc = tavsr_ffn2_fwd(&f, (tavsr_stream_t)s))) return rc;
  }
  return TAVSR_OK;
}

int supported(const tavsr_bf_layer_desc* d, const char* who) {
  TAVSR_REQUIRE(d, TAVSR_EINVAL, "%s: null descriptor", who);
  TAVSR_REQUIRE(tavsr_branchformer_layer_ok(d->B, d->T, d->D, d->H, d->ffn_units, d->cg_units, d->cg_kernel), TAVSR_EUNSUPPORTED,
                "%s: d_model 256, 64-wide heads, hidden >= 1024, cgMLP kernel 31, T <= 2048 only", who);
  return TAVSR_OK;
}

}  // namespace

extern "C" int tavsr_branchformer_layer_ok(int32_t B, int32_t T, int32_t D, int32_t H, int32_t ffn_units, int32_t cg_units,
                                          int32_t cg_kernel) {
  return B > 0 && T > 0 && D == 256 && H > 0 && D / H == 64 && D % H == 0 && ffn_units >= 1024 && ffn_units % 32 == 0 &&
         cg_units > 0 && cg_units % 128 == 0 && cg_kernel == 31 && tavsr_merge_proj_ok(T, D);
}

extern "C" int64_t tavsr_branchformer_layer_ws(const tavsr_bf_layer_desc* d) {
  if (supported(d, "branchformer_layer_ws")) return 0;
  Bump ws{nullptr, 0, 0, true, false};
  if (sequence(d, nullptr, ws)) return 0;
  return ws.used;
}

extern "C" int tavsr_branchformer_layer_fwd(const tavsr_bf_layer_desc* d, tavsr_stream_t stream) {
  int rc = supported(d, "branchformer_layer_fwd");
  if (rc) return rc;
  TAVSR_REQUIRE(d->x && d->pos_emb && d->x1 && d->n_mha && d->n_mlp && d->qkv && d->pp && d->cx && d->lse && d->xa && d->g && d->u &&
                    d->xm && d->g_mean && d->g_rstd && d->score && d->pooled && d->wts && d->m && d->x2 && d->x3 && d->y && d->ws &&
                    d->ev_fork && d->ev_join,      // (stream2 may be the null stream: a caller that runs one queue)
                TAVSR_EINVAL, "branchformer_layer_fwd: null buffer");
  TAVSR_REQUIRE(!d->save || (d->ffm_n && d->ffm_mean && d->ffm_rstd && d->ffm_z && d->ffm_h && d->br_mean && d->br_rstd && d->g_z && d->gn &&
                             d->conv && d->ff_n && d->ff_mean && d->ff_rstd && d->ff_z && d->ff_h && d->fin_mean && d->fin_rstd),
                TAVSR_EINVAL, "branchformer_layer_fwd: save = 1 needs every saved buffer");
  TAVSR_REQUIRE((d->p_drop == 0.f && d->p_att == 0.f) || d->seed, TAVSR_EINVAL, "branchformer_layer_fwd: dropout needs a device seed");
  Bump ws{d->ws, d->ws_floats, 0, false, false};
  {
    Bump dryrun{nullptr, 0, 0, true, false};
    if ((rc = sequence(d, nullptr, dryrun))) return rc;
    TAVSR_REQUIRE(dryrun.used <= d->ws_floats, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small (tavsr_branchformer_layer_ws)");
  }
  return sequence(d, (hipStream_t)stream, ws);
}

// ---------------------------------------------------------------------------------------------- tailored AV layer
// One modality stream of a TailoredEncoderLayer (src/encoder/audiovisual/tailored/encoder_layer.py:118-274): macaron FFN, the
// stream's ONE branch (rel-pos attention or cgMLP, chosen per layer and modality) with its residual, FFN, norm_final - the launches
// of tavsr/functional_av.py:TailoredStreamFn.forward; tavsr_tailored_layer_fwd runs the video stream on the second queue beside the
// audio stream (the FFNs and three of the norms are shared modules: both descriptors point at the same parameters).
namespace {

int ts_ok(const tavsr_tailored_stream_desc* d, const char* who) {
  TAVSR_REQUIRE(d, TAVSR_EINVAL, "%s: null descriptor", who);
  TAVSR_REQUIRE(d->B > 0 && d->T > 0 && d->D == 256 && d->H > 0 && d->D % d->H == 0 && d->D / d->H == 64 && d->ffn_units >= 1024 &&
                    d->ffn_units % 32 == 0 && (d->use_attn || (d->cg_units > 0 && d->cg_units % 128 == 0 && d->cg_units / 2 <= 1024 && d->cg_kernel == 31)),
                TAVSR_EUNSUPPORTED, "%s: d_model 256, 64-wide heads, hidden >= 1024, cgMLP kernel 31 only", who);
  return TAVSR_OK;
}

tavsr_ffn_desc ts_ffn(const tavsr_tailored_stream_desc* d, const float* x, const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                      const float* w2, const float* b2, float* y, float* n, float* mean, float* rstd, float* z, float* h, uint64_t off_in,
                      uint64_t off_out) {
  tavsr_ffn_desc f;
  memset(&f, 0, sizeof f);
  f.M = d->B * d->T; f.D = d->D; f.N1 = d->ffn_units; f.act = d->ffn_act;
  f.scale = 0.5f; f.eps = 1e-12f;
  f.x = x; f.ldx = d->D;
  f.ln_w = ln_w; f.ln_b = ln_b; f.w1 = w1; f.b1 = b1; f.w2 = w2; f.b2 = b2;
  f.y = y;
  if (d->save) { f.n_out = n; f.mean = mean; f.rstd = rstd; f.z = z; f.h = h; }
  f.p_drop = d->p_drop; f.seed = d->seed; f.offset_in = off_in; f.offset_out = off_out;
  f.ln2_eps = 1e-12f;
  return f;
}

// dropout offsets: 0 macaron inner, 1 macaron outer, then attention: 2 probabilities, 3 branch output | cgMLP: 2 gate product, 3 branch
// output; 4 FFN inner, 5 FFN outer  (the order tavsr/functional_av.py draws its tokens in)
int ts_sequence(const tavsr_tailored_stream_desc* d, hipStream_t s, Bump& ws) {
  const int M = d->B * d->T, D = d->D, W = 2 * d->T - 1;
  const bool dry = ws.dry;
  int rc;
  {
    tavsr_ffn_desc f = ts_ffn(d, d->x, d->ffm_ln_w, d->ffm_ln_b, d->ffm_w1, d->ffm_b1, d->ffm_w2, d->ffm_b2, d->x1, d->ffm_n, d->ffm_mean, d->ffm_rstd,
                              d->ffm_z, d->ffm_h, d->drop_off[0], d->drop_off[1]);
    f.ln2_w[0] = d->br_ln_w; f.ln2_b[0] = d->br_ln_b; f.ln2_out[0] = d->n_br;
    if (d->save) { f.ln2_mean = d->br_mean; f.ln2_rstd = d->br_rstd; }
    f.ws_floats = tavsr_ffn2_ws(M, D, d->ffn_units);
    f.ws = ws.take(f.ws_floats);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "tailored_stream_fwd: workspace too small");
    if (!dry && (rc = tavsr_ffn2_fwd(&f, (tavsr_stream_t)s))) return rc;
  }
  if (d->use_attn) {
    if (!dry) {
      tavsr_gemm_desc q3[3] = {lin(M, D, D, d->n_br, D, d->wq, d->bq, d->qkv, 3 * D), lin(M, D, D, d->n_br, D, d->wk, d->bk, d->qkv + D, 3 * D),
                               lin(M, D, D, d->n_br, D, d->wv, d->bv, d->qkv + 2 * D, 3 * D)};
      if ((rc = tavsr_gemm_grouped(q3, 3, (tavsr_stream_t)s))) return rc;
    }
    tavsr_gemm_desc gp = lin(W, D, D, d->pos_emb, D, d->wpos, nullptr, d->pp, D);
    if ((rc = run_gemm(gp, ws, s))) return rc;
    if (!dry) {
      tavsr_attn_desc a;
      memset(&a, 0, sizeof a);
      a.q = d->qkv; a.k = d->qkv + D; a.v = d->qkv + 2 * D;
      a.ldq = a.ldk = a.ldv = 3 * D;
      a.pos = d->pp; a.ldp = D; a.bias_u = d->pos_u; a.bias_v = d->pos_v; a.klens = d->lens;
      a.B = d->B; a.H = d->H; a.T1 = a.T2 = d->T; a.dk = D / d->H;
      a.scale = 1.f / sqrtf((float)a.dk);
      a.p_drop = d->p_att; a.seed_dev = d->seed; a.drop_offset = d->drop_off[2];
      if ((rc = tavsr_attn_fwd(&a, d->cx, D, d->lse, (tavsr_stream_t)s))) return rc;
    }
    tavsr_gemm_desc go = lin(M, D, D, d->cx, D, d->wo, d->bo, d->x2, D);       // x2 = x1 + coeff dropout(linear_out(ctx))
    go.alpha = d->coeff; go.R = d->x1; go.ldr = D;
    go.drop_p = d->p_drop; go.drop_seed = d->seed; go.drop_offset = d->drop_off[3];
    if ((rc = run_gemm(go, ws, s))) return rc;
  } else {
    const int C2 = d->cg_units, Cn = C2 / 2;
    tavsr_gemm_desc g1 = lin(M, C2, D, d->n_br, D, d->cg_w1, d->cg_b1, d->g, C2);
    g1.act = TAVSR_ACT_GELU;
    if (d->save) g1.Z = d->g_z;
    float* rowstat = ws.take((int64_t)M * (C2 / 64) * 2);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "tailored_stream_fwd: workspace too small");
    g1.rowstat = rowstat;
    if ((rc = run_gemm(g1, ws, s))) return rc;
    if (!dry && (rc = tavsr_csgu_fwd(d->g, C2, d->csgu_ln_w, d->csgu_ln_b, 1e-12f, d->csgu_cw, d->csgu_cb, d->u, d->save ? d->gn : nullptr,
                                     d->save ? d->conv : nullptr, d->g_mean, d->g_rstd, d->p_drop, d->seed, d->drop_off[2], d->B, d->T, Cn,
                                     d->cg_kernel, rowstat, (tavsr_stream_t)s)))
      return rc;
    tavsr_gemm_desc g2 = lin(M, D, Cn, d->u, Cn, d->cg_w2, d->cg_b2, d->x2, D);  // x2 = x1 + coeff dropout(cgmlp(.))
    g2.alpha = d->coeff; g2.R = d->x1; g2.ldr = D;
    g2.drop_p = d->p_drop; g2.drop_seed = d->seed; g2.drop_offset = d->drop_off[3];
    if ((rc = run_gemm(g2, ws, s))) return rc;
  }
  {
    tavsr_ffn_desc f = ts_ffn(d, d->x2, d->ff_ln_w, d->ff_ln_b, d->ff_w1, d->ff_b1, d->ff_w2, d->ff_b2, d->x3, d->ff_n, d->ff_mean, d->ff_rstd,
                              d->ff_z, d->ff_h, d->drop_off[4], d->drop_off[5]);
    f.ln2_w[0] = d->final_ln_w; f.ln2_b[0] = d->final_ln_b; f.ln2_out[0] = d->y;
    if (d->save) { f.ln2_mean = d->fin_mean; f.ln2_rstd = d->fin_rstd; }
    f.ws_floats = tavsr_ffn2_ws(M, D, d->ffn_units);
    f.ws = ws.take(f.ws_floats);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "tailored_stream_fwd: workspace too small");
    if (!dry && (rc = tavsr_ffn2_fwd(&f, (tavsr_stream_t)s))) return rc;
  }
  return TAVSR_OK;
}

int ts_check(const tavsr_tailored_stream_desc* d, const char* who) {
  int rc = ts_ok(d, who);
  if (rc) return rc;
  TAVSR_REQUIRE(d->x && d->x1 && d->n_br && d->x2 && d->x3 && d->y && d->ws && d->ffm_ln_w && d->ffm_w1 && d->ffm_w2 && d->ff_ln_w && d->ff_w1 &&
                    d->ff_w2 && d->br_ln_w && d->br_ln_b && d->final_ln_w && d->final_ln_b,
                TAVSR_EINVAL, "%s: null buffer", who);
  if (d->use_attn)
    TAVSR_REQUIRE(d->pos_emb && d->wq && d->wk && d->wv && d->wo && d->wpos && d->pos_u && d->pos_v && d->qkv && d->pp && d->cx && d->lse, TAVSR_EINVAL,
                  "%s: null attention buffer", who);
  else
    TAVSR_REQUIRE(d->cg_w1 && d->cg_b1 && d->csgu_ln_w && d->csgu_ln_b && d->csgu_cw && d->csgu_cb && d->cg_w2 && d->cg_b2 && d->g && d->u &&
                      d->g_mean && d->g_rstd,
                  TAVSR_EINVAL, "%s: null cgMLP buffer", who);
  TAVSR_REQUIRE(!d->save || (d->ffm_n && d->ffm_mean && d->ffm_rstd && d->ffm_z && d->ffm_h && d->br_mean && d->br_rstd && d->ff_n && d->ff_mean &&
                             d->ff_rstd && d->ff_z && d->ff_h && d->fin_mean && d->fin_rstd && (d->use_attn || (d->g_z && d->gn && d->conv))),
                TAVSR_EINVAL, "%s: save = 1 needs every saved buffer", who);
  TAVSR_REQUIRE((d->p_drop == 0.f && d->p_att == 0.f) || d->seed, TAVSR_EINVAL, "%s: dropout needs a device seed", who);
  Bump dryrun{nullptr, 0, 0, true, false};
  if ((rc = ts_sequence(d, nullptr, dryrun))) return rc;
  TAVSR_REQUIRE(dryrun.used <= d->ws_floats, TAVSR_EINVAL, "%s: workspace too small (tavsr_tailored_stream_ws)", who);
  return TAVSR_OK;
}

}  // namespace

extern "C" int64_t tavsr_tailored_stream_ws(const tavsr_tailored_stream_desc* d) {
  if (ts_ok(d, "tailored_stream_ws")) return 0;
  Bump ws{nullptr, 0, 0, true, false};
  if (ts_sequence(d, nullptr, ws)) return 0;
  return ws.used;
}

extern "C" int tavsr_tailored_stream_fwd(const tavsr_tailored_stream_desc* d, tavsr_stream_t stream) {
  int rc = ts_check(d, "tailored_stream_fwd");
  if (rc) return rc;
  Bump ws{d->ws, d->ws_floats, 0, false, false};
  return ts_sequence(d, (hipStream_t)stream, ws);
}

extern "C" int tavsr_tailored_layer_fwd(const tavsr_tailored_layer_desc* d, tavsr_stream_t stream) {
  TAVSR_REQUIRE(d && d->audio && d->video && d->ev_fork && d->ev_join, TAVSR_EINVAL, "tailored_layer_fwd: null descriptor / event");
  int rc;
  if ((rc = ts_check(d->audio, "tailored_layer_fwd (audio)")) || (rc = ts_check(d->video, "tailored_layer_fwd (video)"))) return rc;
  hipStream_t s = (hipStream_t)stream, s2 = (hipStream_t)d->stream2;
  TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_fork, s));
  TAVSR_HIP_CHECK(hipStreamWaitEvent(s2, (hipEvent_t)d->ev_fork, 0));
  if ((rc = probe_fork(s2, s, false))) return rc;
  Bump wv{d->video->ws, d->video->ws_floats, 0, false, false};
  if ((rc = ts_sequence(d->video, s2, wv))) return rc;
  Bump wa{d->audio->ws, d->audio->ws_floats, 0, false, false};
  if ((rc = ts_sequence(d->audio, s, wa))) return rc;
  TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_join, s2));
  TAVSR_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)d->ev_join, 0));
  return probe_fork(s2, s, true);
}

// ---------------------------------------------------------------------------------------------- backward
// tavsr_branchformer_layer_bwd: what tavsr/functional.py:BranchformerLayerFn.backward enqueues (autograd of
// MyBranchformerEncoderLayer.forward, src/encoder/branchformer/encoder_layer.py:153-321) as one C call over the same entry
// points, in the same order, with the same grouping of the weight gradients (one grouped launch at the end) and of the five
// d_model LayerNorms' (dgamma, dbeta) reductions - results are bit-identical to the Python sequencing.
namespace {

struct WItem { const float* dy; int64_t lddy; const float* x; int64_t ldx; int rows, N, K; float alpha; float *out, *gb; };

struct WGroup {        // tavsr/ops.py:WgradGroup
  WItem it[16];
  int n = 0;
  void add(const float* dy, int64_t lddy, const float* x, int64_t ldx, int rows, int N, int K, float alpha, float* out, float* gb) {
    it[n++] = WItem{dy, lddy, x, ldx, rows, N, K, alpha, out, gb};
  }
  static tavsr_gemm_desc desc(const WItem& w) {
    tavsr_gemm_desc g;
    memset(&g, 0, sizeof g);
    g.M = w.N; g.N = w.K; g.K = w.rows;
    g.a_kmajor = g.b_kmajor = 1;
    g.A = w.dy; g.lda = w.lddy; g.B = w.x; g.ldb = w.ldx; g.C = w.out; g.ldc = w.K;
    g.nb1 = g.nb2 = 1;
    g.alpha = w.alpha;
    g.a_rowsum = w.gb;
    return g;
  }
  static long tiles(const WItem& w) { return (long)cdiv(w.N, 64) * cdiv(w.K, 64); }
  int flush(Bump& ws, hipStream_t s) {
    bool keep[16];
    for (int i = 0; i < n; ++i) keep[i] = true;
    long total = 0;
    for (int i = 0; i < n; ++i) total += tiles(it[i]);
    int rc;
    if (n <= 12 && total > 256 && total % 256) {      // shed the fewest small problems down to a multiple of 256 tiles
      const long target = total / 256 * 256;
      int order[16];
      for (int i = 0; i < n; ++i) order[i] = i;
      for (int i = 1; i < n; ++i)                       // stable, descending by tiles
        for (int j = i; j > 0 && tiles(it[order[j]]) > tiles(it[order[j - 1]]); --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
      bool shed[16];
      for (int i = 0; i < n; ++i) shed[i] = false;
      long tot = total;
      int nshed = 0;
      for (int o = 0; o < n; ++o) {
        const int i = order[o];
        if (tiles(it[i]) <= tot - target) { tot -= tiles(it[i]); shed[i] = true; ++nshed; }
      }
      if (tot == target && nshed <= 2)
        for (int i = 0; i < n; ++i)
          if (shed[i]) {
            keep[i] = false;
            tavsr_gemm_desc g = desc(it[i]);
            if ((rc = run_gemm(g, ws, s))) return rc;
          }
    }
    tavsr_gemm_desc arr[12];
    int m = 0;
    for (int i = 0; i <= n; ++i) {
      if (i < n && keep[i]) arr[m++] = desc(it[i]);
      if (m == 12 || (i == n && m > 0)) {
        if (!ws.dry) {
          rc = m > 1 ? tavsr_gemm_grouped(arr, m, (tavsr_stream_t)s) : TAVSR_EUNSUPPORTED;
          if (rc == TAVSR_EUNSUPPORTED) {
            for (int j = 0; j < m; ++j)
              if ((rc = run_gemm(arr[j], ws, s))) return rc;
          } else if (rc) {
            return rc;
          }
        } else {
          for (int j = 0; j < m; ++j) { const int64_t need = tavsr_gemm_ws(&arr[j]); if (need > 0) ws.take(need); }
        }
        m = 0;
      }
    }
    n = 0;
    return TAVSR_OK;
  }
};

struct LnGroup {       // tavsr/ops.py:LNGroup for the five d_model LayerNorms of a layer
  float* slab = nullptr;
  int64_t slab_ld = 0;
  int nb = 0, k = 0, M = 0, D = 0;
  float* out = nullptr;      // [5][2][D]
  int bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_add, float* dx,
          float* dx_drop, float p, const uint64_t* seed, uint64_t off, bool dry, hipStream_t s) {
    float* part = dry ? nullptr : slab + (int64_t)k * 2 * D;
    ++k;
    if (dry) return TAVSR_OK;
    if (dx_drop)
      return tavsr_layernorm_bwd_partial_drop(dy, D, x, D, mean, rstd, gamma, dx_add, dx_add ? D : 0, dx, D, part, slab_ld, M, D, dx_drop, p,
                                              seed, off, (tavsr_stream_t)s);
    return tavsr_layernorm_bwd_partial(dy, D, x, D, mean, rstd, gamma, dx_add, dx_add ? D : 0, dx, D, part, slab_ld, M, D, (tavsr_stream_t)s);
  }
  // dy as the unsummed partials of tavsr_ffn2_bwd_dx(dn = NULL) (tavsr/ops.py: LNGroup.bwd on ops.DnSlabs)
  int bwd_slab(const float* slabs, int wpb, int rbr, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_add,
               float* dx, float* dx_drop, float p, const uint64_t* seed, uint64_t off, bool dry, hipStream_t s) {
    float* part = dry ? nullptr : slab + (int64_t)k * 2 * D;
    ++k;
    if (dry) return TAVSR_OK;
    return tavsr_layernorm_bwd_partial_slab(slabs, wpb, rbr, x, D, mean, rstd, gamma, dx_add, dx_add ? D : 0, dx, D, part, slab_ld, M, D, dx_drop, p,
                                            seed, off, (tavsr_stream_t)s);
  }
  int flush(bool dry, hipStream_t s) {
    if (dry || !k) return TAVSR_OK;
    return tavsr_sum_partials(slab, nb, slab_ld, out, k * 2 * D, 0, (tavsr_stream_t)s);
  }
};

// backward of one feed-forward residual block (tavsr/functional.py:_FFN.bwd, streaming dgrad pair): dyd = dy under the block's outer
// mask (== dy without dropout); returns dx (+ dx under `out_drop` when the next block wants it) through the LayerNorm group
int ffn_bwd(const tavsr_bf_layer_desc* f, const float* dy, const float* dyd, const float* x, const float* mean, const float* rstd,
            const float* n, const float* z, const float* h, const float* ln_w, const float* w1, const float* w2, uint64_t off_in,
            float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* dx, float* dx_drop, uint64_t off_next, WGroup& grp, LnGroup& lng,
            Bump& ws, hipStream_t s) {
  const int M = f->B * f->T, D = f->D, N1 = f->ffn_units, Mp = (M + 127) / 128 * 128;
  int rc;
  grp.add(dyd, D, h, N1, M, D, N1, 0.5f, g_w2, g_b2);
  float* dz = ws.take((int64_t)Mp * N1);
  const int64_t nws = tavsr_ffn2_ws(M, D, N1);
  float* fws = ws.take(nws);
  TAVSR_REQUIRE(ws.dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
  // dn stays in the launch's partial slabs: the block's LayerNorm backward, its only reader, sums them (no finishing launch)
  if (!ws.dry && (rc = tavsr_ffn2_bwd_dx(dyd, D, 0.5f, w1, w2, z, f->ffn_act, M, D, N1, f->p_drop, f->p_drop > 0.f ? f->seed : nullptr, off_in, dz, nullptr,
                                         fws, nws, (tavsr_stream_t)s)))
    return rc;
  grp.add(dz, N1, n, D, M, N1, D, 1.f, g_w1, g_b1);
  int wpb = 0, rbr = 0;
  if ((rc = tavsr_ffn2_slab_layout(M, N1, &wpb, &rbr))) return rc;
  return lng.bwd_slab(fws, wpb, rbr, x, mean, rstd, ln_w, dy, dx, dx_drop, f->p_drop, f->seed, off_next, ws.dry, s);
}

int sequence_bwd(const tavsr_bf_layer_bwd_desc* b, hipStream_t s, Bump& ws) {
  const tavsr_bf_layer_desc* d = b->fwd;
  const int B = d->B, T = d->T, M = B * T, D = d->D, H = d->H, dk = D / H, C2 = d->cg_units, Cn = C2 / 2, W = 2 * T - 1, Wp = (W + 3) / 4 * 4;
  const bool dry = ws.dry, drop = d->p_drop > 0.f;
  hipStream_t s2 = (hipStream_t)d->stream2;
  const uint64_t* seed = drop || d->p_att > 0.f ? d->seed : nullptr;
  int rc;
  auto mat = [&](int64_t n) { return ws.take(n); };
  WGroup grp;
  LnGroup lng;
  std::function<int(hipStream_t)> pos_chain, pos_sums;        // (set by the attention branch)
  const bool pos_late = b->wgrad_beside != 0;      // (not a function of `dry`: the workspace is taken in the same order either way)
  lng.M = M; lng.D = D;
  lng.nb = (int)(tavsr_layernorm_bwd_ws(M, D) / (2 * D));
  lng.slab_ld = 8 * 2 * D;
  lng.slab = mat((int64_t)lng.nb * lng.slab_ld);
  lng.out = b->g_ln;
  float* g_ln = b->g_ln;
  // ---- norm_final; the feed-forward block (its outer mask rides in norm_final's backward, merge_proj's in its own LayerNorm's)
  float* dx3 = mat((int64_t)M * D);
  float* dyd = drop ? mat((int64_t)M * D) : nullptr;
  float* dx2 = mat((int64_t)M * D);
  float* dxd = drop ? mat((int64_t)M * D) : nullptr;
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
  if ((rc = lng.bwd(b->dy, d->x3, d->fin_mean, d->fin_rstd, d->final_ln_w, nullptr, dx3, dyd, d->p_drop, seed, d->drop_off[8], dry, s))) return rc;
  if ((rc = ffn_bwd(d, dx3, drop ? dyd : dx3, d->x2, d->ff_mean, d->ff_rstd, d->ff_n, d->ff_z, d->ff_h, d->ff_ln_w, d->ff_w1, d->ff_w2,
                    d->drop_off[7], b->g_ff_w1, b->g_ff_b1, b->g_ff_w2, b->g_ff_b2, dx2, dxd, d->drop_off[6], grp, lng, ws, s)))
    return rc;
  // ---- merge_proj, learned-average merge (dxa / dxm come back under the branch outputs' masks)
  const float* dmp = drop ? dxd : dx2;
  grp.add(dmp, D, d->m, D, M, D, D, d->coeff, b->g_merge_w, b->g_merge_b);
  float* dm = mat((int64_t)M * D);
  float* dxa = mat((int64_t)M * D);
  float* dxm = mat((int64_t)M * D);
  {
    tavsr_gemm_desc g = lin(M, D, D, dmp, D, d->merge_w, nullptr, dm, D);
    g.b_kmajor = 1; g.ldb = D; g.alpha = d->coeff;
    if ((rc = run_gemm(g, ws, s))) return rc;
    const int64_t nws = tavsr_merge_rows_bwd_ws(B, T, D);
    float* mws = mat(nws);
    float* dparams[8] = {b->g_merge_p[0], b->g_merge_p[1], b->g_merge_p[4], b->g_merge_p[5],
                         b->g_merge_p[2], b->g_merge_p[3], b->g_merge_p[6], b->g_merge_p[7]};
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
    if (!dry && (rc = tavsr_merge_rows_bwd(dm, d->xa, d->xm, d->lens, nullptr, d->merge_p, d->score, d->wts, d->pooled, dxa, dxm, dparams, 0, mws,
                                           d->p_drop, d->drop_off[3], d->p_drop, d->drop_off[5], seed, B, T, D, (tavsr_stream_t)s)))
      return rc;
  }
  // ---- attention branch on the second queue
  float* dn_a = mat((int64_t)M * D);
  {
    if (!dry) {
      TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_fork, s));
      TAVSR_HIP_CHECK(hipStreamWaitEvent(s2, (hipEvent_t)d->ev_fork, 0));
      if ((rc = probe_fork(s2, s, false))) return rc;
    }
    grp.add(dxa, D, d->cx, D, M, D, D, 1.f, b->g_wo, b->g_bo);
    float* dcx = mat((int64_t)M * D);
    tavsr_gemm_desc g = lin(M, D, D, dxa, D, d->wo, nullptr, dcx, D);
    g.b_kmajor = 1; g.ldb = D;
    if ((rc = run_gemm(g, ws, s2))) return rc;
    float* dqkv = mat((int64_t)M * 3 * D);
    float* dqu = mat((int64_t)M * D);
    float* dqv = mat((int64_t)M * D);
    float* sk = mat((int64_t)H * B * T * Wp);
    float* qu = mat((int64_t)M * D);
    float* qv = mat((int64_t)M * D);
    float* dp = mat((int64_t)W * D);
    float* csws = mat(2 * tavsr_colsum_ws(M, D));
    float* wcat = mat((int64_t)3 * D * D);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
    if (!dry) {
      if ((rc = tavsr_fill(sk, 0.f, (int64_t)H * B * T * Wp, (tavsr_stream_t)s2))) return rc;      // the kernel writes the band of every row
      tavsr_attn_desc a;
      memset(&a, 0, sizeof a);
      a.q = d->qkv; a.k = d->qkv + D; a.v = d->qkv + 2 * D;
      a.ldq = a.ldk = a.ldv = 3 * D;
      a.pos = d->pp; a.ldp = D; a.bias_u = d->pos_u; a.bias_v = d->pos_v; a.klens = d->lens;
      a.B = B; a.H = H; a.T1 = a.T2 = T; a.dk = dk;
      a.scale = 1.f / sqrtf((float)dk);
      if (d->p_att > 0.f) { a.p_drop = d->p_att; a.seed_dev = d->seed; a.drop_offset = d->drop_off[2]; }
      if ((rc = tavsr_attn_bwd(&a, dcx, d->cx, D, d->lse, dqu, dqv, D, dqkv + D, 3 * D, dqkv + 2 * D, 3 * D, sk, Wp, (tavsr_stream_t)s2))) return rc;
    }
    // The gradient of the projected positional rows and linear_pos's weight gradient (add_head_bias + two GEMMs, ~55 us at batch 32) have ONE
    // reader: the parameter's gradient.  With wgrad_beside they leave the attention branch - the longer one - and run with the other weight
    // gradients at the end of the call (tavsr/functional.py: pos_dw; audio-only step 1902 -> 1954 utt/s).
    pos_chain = [=, &ws](hipStream_t st) -> int {
      int rc2;
      // dP[:, h] = sum_b ds_skew[h, b]^T (q + v)[b, :, h]: one K = B T contraction per head
      if (!ws.dry && (rc2 = tavsr_add_head_bias(d->qkv, 3 * D, d->pos_u, d->pos_v, qu, qv, M, D, (tavsr_stream_t)st))) return rc2;
      tavsr_gemm_desc gp;
      memset(&gp, 0, sizeof gp);
      gp.M = W; gp.N = dk; gp.K = M;
      gp.a_kmajor = gp.b_kmajor = 1;
      gp.A = sk; gp.lda = Wp; gp.B = qv; gp.ldb = D; gp.C = dp; gp.ldc = D;
      gp.nb1 = H; gp.nb2 = 1;
      gp.sA1 = (int64_t)M * Wp; gp.sB1 = dk; gp.sC1 = dk;
      gp.alpha = 1.f;
      if ((rc2 = run_gemm(gp, ws, st))) return rc2;
      tavsr_gemm_desc gw;       // linear_pos.weight: K = 2T - 1 is not a multiple of 32, it stays alone
      memset(&gw, 0, sizeof gw);
      gw.M = D; gw.N = D; gw.K = W;
      gw.a_kmajor = gw.b_kmajor = 1;
      gw.A = dp; gw.lda = D; gw.B = d->pos_emb; gw.ldb = D; gw.C = b->g_wpos; gw.ldc = D;
      gw.nb1 = gw.nb2 = 1;
      gw.alpha = 1.f;
      return run_gemm(gw, ws, st);
    };
    if (!pos_late && (rc = pos_chain(s2))) return rc;
    if (!dry && (rc = tavsr_add2_colsum(dqu, D, dqv, D, dqkv, 3 * D, M, D, pos_late ? nullptr : b->g_pos_u, pos_late ? nullptr : b->g_pos_v, csws,
                                        (tavsr_stream_t)s2)))
      return rc;
    if (pos_late) pos_sums = [=](hipStream_t st) -> int {      // the two bias gradients: reduced with the other weight gradients at the end
      return tavsr_sum_partials2(csws, (int32_t)(tavsr_colsum_ws(M, D) / D), 2 * (int64_t)D, b->g_pos_u, D, b->g_pos_v, D, 0, (tavsr_stream_t)st);
    };
    grp.add(dqkv, 3 * D, d->n_mha, D, M, D, D, 1.f, b->g_wq, b->g_bq);
    grp.add(dqkv + D, 3 * D, d->n_mha, D, M, D, D, 1.f, b->g_wk, b->g_bk);
    grp.add(dqkv + 2 * D, 3 * D, d->n_mha, D, M, D, D, 1.f, b->g_wv, b->g_bv);
    if (!dry) {
      void* dst[3] = {wcat, wcat + (int64_t)D * D, wcat + (int64_t)2 * D * D};
      const void* src[3] = {d->wq, d->wk, d->wv};
      const int64_t nbytes[3] = {(int64_t)D * D * 4, (int64_t)D * D * 4, (int64_t)D * D * 4};
      if ((rc = tavsr_multi_copy(dst, src, nbytes, 3, (tavsr_stream_t)s2))) return rc;
    }
    tavsr_gemm_desc gc = lin(M, D, 3 * D, dqkv, 3 * D, wcat, nullptr, dn_a, D);     // dn = dqkv [wq; wk; wv]: one K = 3 D GEMM
    gc.b_kmajor = 1; gc.ldb = D;
    if ((rc = run_gemm(gc, ws, s2))) return rc;
  }
  // ---- cgMLP branch on the calling queue
  float* dx1 = mat((int64_t)M * D);
  {
    grp.add(dxm, D, d->u, Cn, M, D, Cn, 1.f, b->g_cg_w2, b->g_cg_b2);
    float* du = mat((int64_t)M * Cn);
    tavsr_gemm_desc g = lin(M, Cn, D, dxm, D, d->cg_w2, nullptr, du, Cn);
    g.b_kmajor = 1; g.ldb = Cn;
    if (drop) { g.drop_p = d->p_drop; g.drop_seed = d->seed; g.drop_offset = d->drop_off[4]; }
    if ((rc = run_gemm(g, ws, s))) return rc;
    float* dg = mat((int64_t)M * C2);
    float* dgn = mat((int64_t)M * Cn);
    float* cws = mat(tavsr_dwconv_gate_bwd_ws(B, T, Cn, d->cg_kernel));
    float* lws = mat(tavsr_layernorm_bwd_ws(M, Cn));
    float* dn = mat((int64_t)M * D);
    TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
    if (!dry) {
      if ((rc = tavsr_dwconv_gate_bwd_act(du, d->gn, d->g, C2, d->conv, d->csgu_cw, dg, C2, dgn, b->g_csgu_cw, b->g_csgu_cb, 0, cws, B, T, Cn,
                                          d->cg_kernel, d->g_z, C2, TAVSR_ACT_GELU, (tavsr_stream_t)s)))
        return rc;
      if ((rc = tavsr_layernorm_bwd_act(dgn, Cn, d->g + Cn, C2, d->g_mean, d->g_rstd, d->csgu_ln_w, dg + Cn, C2, b->g_csgu_ln_w, b->g_csgu_ln_b, 0,
                                        lws, M, Cn, d->g_z + Cn, C2, TAVSR_ACT_GELU, (tavsr_stream_t)s)))
        return rc;
    }
    grp.add(dg, C2, d->n_mlp, D, M, C2, D, 1.f, b->g_cg_w1, b->g_cg_b1);
    tavsr_gemm_desc g1 = lin(M, D, C2, dg, C2, d->cg_w1, nullptr, dn, D);
    g1.b_kmajor = 1; g1.ldb = D;
    if ((rc = run_gemm(g1, ws, s))) return rc;
    if ((rc = lng.bwd(dn, d->x1, d->br_mean, d->br_rstd, d->mlp_ln_w, dx2, dx1, nullptr, 0.f, nullptr, 0, dry, s))) return rc;
  }
  if (!dry) {
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_join, s2));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)d->ev_join, 0));
    if ((rc = probe_fork(s2, s, true))) return rc;
  }
  // ---- norm_mha (same accumulation order into dx1 as a single queue: cgMLP branch first, then attention), macaron block
  float* dx1b = mat((int64_t)M * D);
  float* dyd2 = drop ? mat((int64_t)M * D) : nullptr;
  TAVSR_REQUIRE(dry || !ws.overflow, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small");
  if ((rc = lng.bwd(dn_a, d->x1, d->br_mean, d->br_rstd, d->mha_ln_w, dx1, dx1b, dyd2, d->p_drop, seed, d->drop_off[1], dry, s))) return rc;
  if ((rc = ffn_bwd(d, dx1b, drop ? dyd2 : dx1b, d->x, d->ffm_mean, d->ffm_rstd, d->ffm_n, d->ffm_z, d->ffm_h, d->ffm_ln_w, d->ffm_w1, d->ffm_w2,
                    d->drop_off[0], b->g_ffm_w1, b->g_ffm_b1, b->g_ffm_w2, b->g_ffm_b2, b->dx, nullptr, 0, grp, lng, ws, s)))
    return rc;
  hipStream_t sw = s;
  if (!dry && b->wgrad_beside && s2 != s) {     // nobody inside the backward pass reads a weight gradient: beside the next layer's chain
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_fork, s));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(s2, (hipEvent_t)d->ev_fork, 0));
    sw = s2;
  }
  if (!dry && pos_sums && (rc = pos_sums(sw))) return rc;
  if (pos_late && pos_chain && (rc = pos_chain(sw))) return rc;
  if ((rc = grp.flush(ws, sw))) return rc;
  (void)g_ln;
  return lng.flush(dry, sw);
}

}  // namespace

extern "C" int64_t tavsr_branchformer_layer_bwd_ws(const tavsr_bf_layer_bwd_desc* b) {
  if (!b || !b->fwd || supported(b->fwd, "branchformer_layer_bwd_ws")) return 0;
  Bump ws{nullptr, 0, 0, true, false};
  if (sequence_bwd(b, nullptr, ws)) return 0;
  return ws.used;
}

extern "C" int tavsr_branchformer_layer_bwd(const tavsr_bf_layer_bwd_desc* b, tavsr_stream_t stream) {
  TAVSR_REQUIRE(b && b->fwd, TAVSR_EINVAL, "branchformer_layer_bwd: null descriptor");
  const tavsr_bf_layer_desc* d = b->fwd;
  int rc = supported(d, "branchformer_layer_bwd");
  if (rc) return rc;
  TAVSR_REQUIRE(d->save, TAVSR_EINVAL, "branchformer_layer_bwd: the forward call must have kept its state (save = 1)");
  TAVSR_REQUIRE(b->dy && b->dx && b->g_ln && b->ws && d->ev_fork && d->ev_join, TAVSR_EINVAL, "branchformer_layer_bwd: null buffer");
  const float* const need[] = {b->g_ffm_w1, b->g_ffm_b1, b->g_ffm_w2, b->g_ffm_b2, b->g_wq, b->g_bq, b->g_wk, b->g_bk, b->g_wv, b->g_bv, b->g_wo, b->g_bo,
                               b->g_wpos, b->g_pos_u, b->g_pos_v, b->g_cg_w1, b->g_cg_b1, b->g_csgu_ln_w, b->g_csgu_ln_b, b->g_csgu_cw, b->g_csgu_cb,
                               b->g_cg_w2, b->g_cg_b2, b->g_merge_w, b->g_merge_b, b->g_ff_w1, b->g_ff_b1, b->g_ff_w2, b->g_ff_b2};
  for (const float* g : need) TAVSR_REQUIRE(g, TAVSR_EINVAL, "branchformer_layer_bwd: null gradient buffer");
  for (int i = 0; i < 8; ++i) TAVSR_REQUIRE(b->g_merge_p[i], TAVSR_EINVAL, "branchformer_layer_bwd: null merge gradient %d", i);
  {
    Bump dryrun{nullptr, 0, 0, true, false};
    if ((rc = sequence_bwd(b, nullptr, dryrun))) return rc;
    TAVSR_REQUIRE(dryrun.used <= b->ws_floats, TAVSR_EINVAL, "branchformer_layer_bwd: workspace too small (tavsr_branchformer_layer_bwd_ws)");
  }
  Bump ws{b->ws, b->ws_floats, 0, false, false};
  return sequence_bwd(b, (hipStream_t)stream, ws);
}

