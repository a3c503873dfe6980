// One Branchformer encoder layer forward sequenced in C (tavsr_branchformer_layer_fwd, include/tavsr.h): the launches that
// tavsr/functional.py:BranchformerLayerFn.forward enqueues through ~45 Python-level calls, as one C call over the same entry
// points - for un-captured (ragged) training loops, whose step is host-bound on that sequencing.  Host code only.
// Reference: src/encoder/branchformer/encoder_layer.py:153-321 (MyBranchformerEncoderLayer.forward).
#include <cmath>
#include <cstring>

#include "common.h"

using namespace tavsr;

#define TAVSR_HIP_CHECK(call)                                                                    \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      ::tavsr::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));     \
      return (int)e__;                                                                            \
    }                                                                                             \
  } while (0)

namespace {

struct Bump {           // workspace carving (two queues run side by side: every launch gets its own region)
  float* base;
  int64_t cap, used;
  bool dry;
  bool overflow = false;
  // dry run: sizes only, but a non-null sentinel so that descriptors which switch on "is this pointer given" (rowstat)
  // plan the same launches as the real pass; real pass: never past the caller's capacity
  float* take(int64_t n) {
    n = (n + 63) / 64 * 64;
    float* p = dry ? reinterpret_cast<float*>(uintptr_t(64)) : base + used;
    if (!dry && used + n > cap) { overflow = true; p = nullptr; }
    used += n;
    return p;
  }
};

tavsr_gemm_desc lin(int M, int N, int K, const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t ldo) {
  tavsr_gemm_desc g;
  memset(&g, 0, sizeof g);
  g.M = M; g.N = N; g.K = K;
  g.A = x; g.lda = ldx; g.B = w; g.ldb = K; g.C = out; g.ldc = ldo;
  g.nb1 = g.nb2 = 1;
  g.bias = b;
  g.alpha = 1.f;
  return g;
}

int run_gemm(tavsr_gemm_desc& g, Bump& ws, hipStream_t s) {
  const int64_t need = tavsr_gemm_ws(&g);
  if (need > 0) { g.ws = ws.take(need); g.ws_floats = need; }
  if (ws.dry) return TAVSR_OK;
  TAVSR_REQUIRE(!ws.overflow, TAVSR_EINVAL, "branchformer_layer_fwd: workspace too small for a GEMM's split-K slabs");
  return tavsr_gemm(&g, (tavsr_stream_t)s);
}

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
  {
    tavsr_gemm_desc go = lin(M, D, D, d->cx, D, d->wo, d->bo, d->xa, D);       // x_att = dropout(linear_out(ctx))
    go.drop_p = d->p_drop; go.drop_seed = d->seed; go.drop_offset = d->drop_off[3];
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
    if ((rc = run_gemm(g2, ws, s))) return rc;
  }
  if (!dry) {
    TAVSR_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_join, s2));
    TAVSR_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)d->ev_join, 0));
    if ((rc = probe_fork(s2, s, true))) return rc;
    // ---- the tail behind the join as one launch: learned-average merge (d->pooled receives the row dots [4][B*T] its backward
    //      reads), x2 = x1 + coeff dropout(merge_proj(m))
    if ((rc = tavsr_merge_proj_fwd(d->xa, d->xm, d->lens, nullptr, d->merge_p, d->merge_w, d->merge_b, d->x1, d->coeff, d->p_drop,
                                   d->seed, d->drop_off[6], d->pooled, d->score, d->wts, d->save ? d->m : nullptr, d->x2, d->B, d->T,
                                   D, (tavsr_stream_t)s)))
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
