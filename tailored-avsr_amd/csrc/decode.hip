// One-token decode steps of the hybrid CTC/attention beam search with LM scoring (the reference builds it
// at src/inference/avsr_inference.py:141-304 from espnet's BatchBeamSearch, TransformerDecoder.batch_score,
// TransformerLM.batch_score, CTCPrefixScorer / CTCPrefixScoreTH and runs it at :449-518, on the CPU, one utterance at
// a time).  Here all hypotheses of all utterances of a batch advance together; these are the kernels that have no
// counterpart on the training path:
//   * tree_attn_step  - self-attention of ONE new query per hypothesis over its own history.  Keys/values live in a
//     node pool (one row per (step, slot)); a hypothesis is the list of its ancestors' rows, so re-ordering the beam
//     copies int32 ancestor lists instead of key/value caches.  HBM/L2-bound gather: 2 * L * dk * 4 B per (hyp, head).
//   * ctc_prefix_step - CTCPrefixScoreTH.__call__ for the pre-beam candidates of every hypothesis: forward variables
//     r^n_t, r^b_t in log space, sequential in t, one thread per (hypothesis, candidate).
//   * log_softmax_rows - scorer outputs.
#include "common.h"

namespace tavsr {

constexpr int kTreeMaxKeys = 1024;

// one wave per (hypothesis n, head h).  Latency-bound at small N (a chain of dependent gathers): the ancestor list is
// read once into LDS, the scores use one lane per key (16 float4 loads in flight per lane), the weighted value sum uses
// 64 / DL key groups x DL lanes of float4 so that a wave has 8 independent row gathers in flight per lane.
// Large steps (a batch of utterances x beam: 5120 items at 64 x 10 x 8 heads) are bound by the rate at which the L2s serve the row
// gathers - 262 MB of requests per launch at 100 keys, 28 us, whatever the beams share (measured 1.2 - 1.4 distinct rows per step and
// utterance).  Three forms that tried to use the sharing all lost (profiles/r05_tree_attn_large_step_variants.txt, r05_notes.md): a
// whole beam per workgroup (same time: the vector cache does not merge the requests), one workgroup per (utterance, head) copying
// the distinct rows into LDS once (1.5x slower: its phases are serial behind workgroup barriers, one workgroup per compute unit),
// rows read 256 bytes at a time by 16 lanes with the scores summed on the lane network (1.3x slower).
template <int DL>      // lanes along the head dimension (float4 each): 16 (dk <= 64) or 32 (dk <= 128)
__global__ __launch_bounds__(256) void tree_attn_step_kernel(const float* __restrict__ q, int64_t ldq,
                                                             const float* __restrict__ kpool, const float* __restrict__ vpool,
                                                             int64_t ldkv, const int32_t* __restrict__ anc, int64_t ld_anc,
                                                             int nkeys, float* __restrict__ out, int64_t ldo, int N, int H,
                                                             int dk, float scale, const int32_t* __restrict__ step_dev,
                                                             const float* __restrict__ k_new, const float* __restrict__ v_new,
                                                             float* __restrict__ kpool_w, float* __restrict__ vpool_w, int group) {
  constexpr int KG = 64 / DL;
  constexpr int VP = 16;                                    // value rows per lane fetched ahead of the softmax: VP * KG keys
  __shared__ float s_p[4][kTreeMaxKeys];
  __shared__ int32_t s_a[4][kTreeMaxKeys];
  if (step_dev) nkeys = min(*step_dev + 1, nkeys);        // replayed graphs: the step counter lives in device memory
  // k_new / v_new (row stride ldq): this step's own key / value rows.  They are the LAST key of every hypothesis (pool row
  // (nkeys-1) * N + n): read from here instead of the pool, and appended to the pool by the same wave (no append launch).
  __shared__ float s_q[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;                       // waves (items) per workgroup: 4, or 1 for small steps (see the launch)
  const bool live = blockIdx.x * wpb + wave < N * H;     // surplus waves recompute the last item and store nothing
  const int item = min(blockIdx.x * wpb + wave, N * H - 1);
  // item order (utterance, head, beam slot): the waves of a block are hypotheses of ONE utterance on ONE head - their
  // ancestor lists share most rows (a beam is a tree), so the block's gathers hit the same lines in the CU's cache
  const int n = (item / (H * group)) * group + item % group, h = (item / group) % H;
  const float* qv = q + (int64_t)n * ldq + h * dk;
  const int32_t* a = anc + (int64_t)n * ld_anc;
  for (int d = lane; d < dk; d += 64) s_q[wave][d] = qv[d] * scale;
  for (int j = lane; j < nkeys; j += 64) s_a[wave][j] = a[j];
  // every wave works on its own slices of the LDS arrays: the order of one wave's LDS operations is all that is needed
  // (no workgroup barrier: a wave does not wait for its three neighbours' loads)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int dl = lane & (DL - 1), kg = lane / DL;
  const bool dok = dl * 4 < dk;
  const int doff = h * dk + (dok ? dl * 4 : 0);
  const float* vb = vpool + doff;
  const float* vlast = v_new ? v_new + (int64_t)n * ldq + doff : nullptr;       // key nkeys - 1 of this hypothesis
  auto vrow = [&](int j) { return (vlast && j == nkeys - 1) ? vlast : vb + (int64_t)s_a[wave][j] * ldkv; };
  // the value rows depend on the ancestor list only, not on the scores: the first VP * KG keys' rows are requested now, with
  // the key rows - one memory round trip for both instead of one after the other (a search step is a chain of such trips)
  float4 vpre[VP];
#pragma unroll
  for (int u = 0; u < VP; ++u) {
    const int j = kg + u * KG;
    vpre[u] = j < nkeys ? *reinterpret_cast<const float4*>(vrow(j)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mx = -INFINITY;
  for (int j = lane; j < nkeys; j += 64) {
    const float* kr = (k_new && j == nkeys - 1) ? k_new + (int64_t)n * ldq + h * dk : kpool + (int64_t)s_a[wave][j] * ldkv + h * dk;
    float dot = 0.f;
    for (int d = 0; d < dk; d += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + d);
      dot += s_q[wave][d] * kv.x + s_q[wave][d + 1] * kv.y + s_q[wave][d + 2] * kv.z + s_q[wave][d + 3] * kv.w;
    }
    s_p[wave][j] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < nkeys; j += 64) {
    const float e = expf(s_p[wave][j] - mx);
    s_p[wave][j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float inv = 1.f / sum;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < VP; ++u) {
    const int j = kg + u * KG;
    if (j < nkeys) {
      const float pj = s_p[wave][j];
      acc.x += pj * vpre[u].x; acc.y += pj * vpre[u].y; acc.z += pj * vpre[u].z; acc.w += pj * vpre[u].w;
    }
  }
  int j = kg + VP * KG;
  for (; j + 7 * KG < nkeys; j += 8 * KG) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(vrow(j + u * KG));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pj = s_p[wave][j + u * KG];
      acc.x += pj * v[u].x; acc.y += pj * v[u].y; acc.z += pj * v[u].z; acc.w += pj * v[u].w;
    }
  }
  for (; j < nkeys; j += KG) {
    const float4 v = *reinterpret_cast<const float4*>(vrow(j));
    const float pj = s_p[wave][j];
    acc.x += pj * v.x; acc.y += pj * v.y; acc.z += pj * v.z; acc.w += pj * v.w;
  }
#pragma unroll
  for (int o = DL; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (kg == 0 && dok && live) {
    if (v_new) {
      const int64_t rowp = (int64_t)(nkeys - 1) * N + n;
      *reinterpret_cast<float4*>(vpool_w + rowp * ldkv + doff) = *reinterpret_cast<const float4*>(vlast);
      *reinterpret_cast<float4*>(kpool_w + rowp * ldkv + doff) = *reinterpret_cast<const float4*>(k_new + (int64_t)n * ldq + doff);
    }
    *reinterpret_cast<float4*>(out + (int64_t)n * ldo + doff) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// The same step for SMALL steps (N H <= 256 items: batch 1 - 3 x beam 10): one workgroup of four waves per (hypothesis, head), the keys
// dealt to the waves in four contiguous runs.  A one-wave item walks its ~100 keys in two passes of dependent gathers (ancestor list ->
// key rows -> value rows beyond the prefetched ones): 8 us of a 30 us LM layer.  Here a wave has at most 64 keys per round - ONE per
// lane for up to 256 keys - so a round is two round trips to memory: the ancestor indices (with q), then every key AND value row of
// the run at once (the value rows in the (float4 column, key group) lane layout, their row indices passed over the lane network).
// Each wave keeps an online softmax (running max, sum, un-normalised context); the four partial results meet in LDS behind the one
// workgroup barrier and wave 0 merges them.  Item -> (n, h) mapping, k_new / v_new append and step_dev as in the kernel above.
template <int DL>
__global__ __launch_bounds__(256) void tree_attn_split_kernel(const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ kpool, const float* __restrict__ vpool,
                                                              int64_t ldkv, const int32_t* __restrict__ anc, int64_t ld_anc,
                                                              int nkeys, float* __restrict__ out, int64_t ldo, int N, int H,
                                                              int dk, float scale, const int32_t* __restrict__ step_dev,
                                                              const float* __restrict__ k_new, const float* __restrict__ v_new,
                                                              float* __restrict__ kpool_w, float* __restrict__ vpool_w, int group) {
  constexpr int KG = 64 / DL;             // key groups of a wave in the value phase
  constexpr int NW = 4;                   // waves per item
  constexpr int NQ = DL;                  // float4s of a head row (dk <= 4 DL)
  __shared__ float s_m[NW], s_s[NW];
  __shared__ float4 s_acc[NW][DL];
  if (step_dev) nkeys = min(*step_dev + 1, nkeys);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x;
  const int n = (item / (H * group)) * group + item % group, h = (item / group) % H;
  const int nq = dk >> 2;                 // float4s actually present
  const float* qv = q + (int64_t)n * ldq + h * dk;
  const int32_t* a = anc + (int64_t)n * ld_anc;
  const int per = (nkeys + NW - 1) / NW;
  const int j0 = wave * per, j1 = min(nkeys, j0 + per);
  const int dl = lane & (DL - 1), kg = lane / DL;
  const bool dok = dl < nq;
  const int doff = h * dk + (dok ? dl * 4 : 0);
  const float* klast = k_new ? k_new + (int64_t)n * ldq + h * dk : nullptr;
  const float* vlast = v_new ? v_new + (int64_t)n * ldq + doff : nullptr;
  // round trip 1: the query (the same 16 / 32 float4s for every lane) and the first round's ancestor index of this lane's key
  float4 qr[NQ];
#pragma unroll
  for (int d = 0; d < NQ; ++d) qr[d] = d < nq ? *reinterpret_cast<const float4*>(qv + 4 * d) : make_float4(0.f, 0.f, 0.f, 0.f);
  float m = -INFINITY, ssum = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = j0; base < j1; base += 64) {
    const int j = base + lane;
    const bool kvalid = j < j1;
    const int aj = kvalid ? a[j] : 0;
    // round trip 2: this lane's key row, and the value rows of keys base + kg + u KG in the (dl, kg) layout
    const float* kr = (klast && j == nkeys - 1) ? klast : kpool + (int64_t)aj * ldkv + h * dk;
    float4 kv[NQ];
#pragma unroll
    for (int d = 0; d < NQ; ++d) kv[d] = (kvalid && d < nq) ? *reinterpret_cast<const float4*>(kr + 4 * d) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 vv[DL];
#pragma unroll
    for (int u = 0; u < DL; ++u) {
      const int src = kg + u * KG;                       // lane that holds this key's ancestor index
      const int jj = base + src;
      const int av = __shfl(aj, src, 64);
      const float* vr = (vlast && jj == nkeys - 1) ? vlast : vpool + (int64_t)av * ldkv + doff;
      vv[u] = (jj < j1 && dok) ? *reinterpret_cast<const float4*>(vr) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < NQ; ++d) dot += (qr[d].x * kv[d].x + qr[d].y * kv[d].y) + (qr[d].z * kv[d].z + qr[d].w * kv[d].w);
    dot = kvalid ? dot * scale : -INFINITY;
    const float mr = wave_max(dot);
    const float mn = fmaxf(m, mr);                        // (a round of a non-empty run always holds a valid key: mn is finite)
    const float p = kvalid ? expf(dot - mn) : 0.f;
    const float corr = m == -INFINITY ? 0.f : expf(m - mn);
    ssum = ssum * corr + wave_sum(p);
    acc.x *= corr; acc.y *= corr; acc.z *= corr; acc.w *= corr;
#pragma unroll
    for (int u = 0; u < DL; ++u) {
      const float pj = __shfl(p, kg + u * KG, 64);
      acc.x += pj * vv[u].x; acc.y += pj * vv[u].y; acc.z += pj * vv[u].z; acc.w += pj * vv[u].w;
    }
    m = mn;
  }
#pragma unroll
  for (int o = DL; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (lane == 0) { s_m[wave] = m; s_s[wave] = ssum; }
  if (kg == 0) s_acc[wave][dl] = acc;
  __syncthreads();
  if (wave != 0 || kg != 0 || !dok) return;
  float M = s_m[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w]);
  float S = 0.f;
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float f = s_m[w] == -INFINITY ? 0.f : expf(s_m[w] - M);
    const float4 t = s_acc[w][dl];
    S += s_s[w] * f;
    o4.x += t.x * f; o4.y += t.y * f; o4.z += t.z * f; o4.w += t.w * f;
  }
  const float inv = 1.f / S;
  if (v_new) {
    const int64_t rowp = (int64_t)(nkeys - 1) * N + n;
    *reinterpret_cast<float4*>(vpool_w + rowp * ldkv + doff) = *reinterpret_cast<const float4*>(vlast);
    *reinterpret_cast<float4*>(kpool_w + rowp * ldkv + doff) = *reinterpret_cast<const float4*>(k_new + (int64_t)n * ldq + doff);
  }
  *reinterpret_cast<float4*>(out + (int64_t)n * ldo + doff) = make_float4(o4.x * inv, o4.y * inv, o4.z * inv, o4.w * inv);
}

// kpool/vpool row (step * N + n) = this step's key / value of hypothesis n; step read from device memory so that one
// captured graph serves every step of the search (steps past the pool are dropped)
__global__ __launch_bounds__(256) void kv_append_kernel(const float* __restrict__ k, const float* __restrict__ v, int64_t ld_src,
                                                        float* __restrict__ kpool, float* __restrict__ vpool, int64_t ldkv, int N,
                                                        int D4, int max_steps, const int32_t* __restrict__ step_dev) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int step = *step_dev;
  if (i >= (int64_t)N * D4 || step < 0 || step >= max_steps) return;
  const int n = (int)(i / D4), c = (int)(i % D4) * 4;
  const int64_t dst = ((int64_t)step * N + n) * ldkv + c;
  *reinterpret_cast<float4*>(kpool + dst) = *reinterpret_cast<const float4*>(k + (int64_t)n * ld_src + c);
  *reinterpret_cast<float4*>(vpool + dst) = *reinterpret_cast<const float4*>(v + (int64_t)n * ld_src + c);
}

// log(e^a + e^b) = max + log(1 + e^-|a - b|) on v_exp_f32 / v_log_f32 (1 ulp each): ~8 instructions instead of two expf and a
// logf (~70) - the prefix recursion runs three of these per frame, one frame after the other
__device__ __forceinline__ float logaddexp2(float a, float b) {
  const float m = fmaxf(a, b), d = -fabsf(a - b);
  return m + 0.69314718055994530942f * __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(1.4426950408889634f * d));
}

// One (hypothesis n, candidate column c) of CTCPrefixScoreTH.__call__.  logp [U][T][V] (log-softmax of the CTC head),
// lens [U].  r_prev [N][T][2] (nb, b) and s_prev [N] of the hypothesis (first == 1: the <sos> state is built here instead).
// Writes r_new [N][T][2][C], psi [N][C] = log_psi(candidate) - s_prev  and  psi_abs [N][C] = log_psi(candidate);
// column c == 0 also writes eos [N] = r_sum[len-1] - s_prev and eos_abs.
__device__ __forceinline__ void ctc_prefix_one(const float* __restrict__ logp, const int64_t* __restrict__ lens,
                                               const float* __restrict__ r_prev, const float* __restrict__ s_prev,
                                               const int64_t* __restrict__ last_tok, int tok, float* __restrict__ r_new,
                                               float* __restrict__ psi, float* __restrict__ psi_abs, float* __restrict__ eos,
                                               float* __restrict__ eos_abs, int n, int c, int K, int T, int V, int C, int out_len,
                                               int blank, int first) {
  const float logzero = -10000000000.0f;
  const int u = n / K;
  const int L = (int)lens[u];
  const float* lp = logp + (int64_t)u * T * V;
  const float* rp = r_prev + (int64_t)n * T * 2;
  const float sp = first ? 0.f : s_prev[n];
  const bool same = !first && tok == (int)last_tok[n];
  float* rn = r_new + (int64_t)n * T * 2 * C + c;
  // previous forward variables: (nb, b) per frame; <sos>: nb = logzero, b = cumulative blank
  float cum = 0.f;
  auto prev = [&](int t, float& pn, float& pb) {
    if (first) { pn = logzero; pb = cum; }     // cum must already hold sum_{t' <= t} logp[t'][blank]
    else { pn = rp[t * 2]; pb = rp[t * 2 + 1]; }
  };
  const int start = max(out_len, 1);
  for (int t = 0; t < start && t < T; ++t) { rn[(int64_t)(t * 2) * C] = logzero; rn[(int64_t)(t * 2 + 1) * C] = logzero; }
  float rn_n = logzero, rn_b = logzero;        // r[start - 1]
  if (out_len == 0) {
    rn_n = lp[tok];                            // r[0, nb] = x[0][tok]
    rn[0] = rn_n;
  }
  if (first) { for (int t = 0; t < start - 1; ++t) cum += lp[(int64_t)t * V + blank]; }
  // log_psi accumulates logsumexp over t in [start, L) of (phi[t-1] + x[t][tok]) and r[start-1, nb]
  float acc = rn_n;
  if (first) {
    for (int t = start; t < L; ++t) {
      float pn, pb;
      cum += lp[(int64_t)(t - 1) * V + blank];
      prev(t - 1, pn, pb);
      const float phi = same ? pb : logaddexp2(pn, pb);
      const float xt = lp[(int64_t)t * V + tok], xb = lp[(int64_t)t * V + blank];
      const float nn = logaddexp2(rn_n, phi) + xt;
      const float nb = logaddexp2(rn_n, rn_b) + xb;
      acc = logaddexp2(acc, phi + xt);
      rn_n = nn; rn_b = nb;
      rn[(int64_t)(t * 2) * C] = nn;
      rn[(int64_t)(t * 2 + 1) * C] = nb;
    }
  } else {
    // The recursion is sequential in t, its INPUTS are not: the four values a frame needs (two posteriors, the previous
    // prefix's two forward variables) are fetched 16 frames at a time, all loads of a chunk in flight together - one memory round
    // trip per chunk instead of one per frame (100 frames: 51 us of a 130 us tail behind the scorers were these round trips).
    constexpr int CH = 16;
    for (int t0 = start; t0 < L; t0 += CH) {
      float xt[CH], xb[CH], pn[CH], pb[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int t = min(t0 + j, L - 1);
        xt[j] = lp[(int64_t)t * V + tok];
        xb[j] = lp[(int64_t)t * V + blank];
        pn[j] = rp[(t - 1) * 2];
        pb[j] = rp[(t - 1) * 2 + 1];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int t = t0 + j;
        if (t < L) {
          const float phi = same ? pb[j] : logaddexp2(pn[j], pb[j]);
          const float nn = logaddexp2(rn_n, phi) + xt[j];
          const float nb = logaddexp2(rn_n, rn_b) + xb[j];
          acc = logaddexp2(acc, phi + xt[j]);
          rn_n = nn; rn_b = nb;
          rn[(int64_t)(t * 2) * C] = nn;
          rn[(int64_t)(t * 2 + 1) * C] = nb;
        }
      }
    }
  }
  for (int t = max(L, start); t < T; ++t) { rn[(int64_t)(t * 2) * C] = logzero; rn[(int64_t)(t * 2 + 1) * C] = logzero; }
  const float out_psi = tok == blank ? logzero : acc;
  psi_abs[(int64_t)n * C + c] = out_psi;
  psi[(int64_t)n * C + c] = out_psi - sp;
  if (c == 0) {
    float pn, pb;
    if (first) { cum = 0.f; for (int t = 0; t < L; ++t) cum += lp[(int64_t)t * V + blank]; }
    prev(L - 1, pn, pb);
    const float e = logaddexp2(pn, pb);
    eos_abs[n] = e;
    eos[n] = e - sp;
  }
}

// thread = (hypothesis n, candidate c), candidates given
__global__ __launch_bounds__(256) void ctc_prefix_step_kernel(const float* __restrict__ logp, const int64_t* __restrict__ lens,
                                                              const float* __restrict__ r_prev, const float* __restrict__ s_prev,
                                                              const int64_t* __restrict__ last_tok,
                                                              const int64_t* __restrict__ cand, float* __restrict__ r_new,
                                                              float* __restrict__ psi, float* __restrict__ psi_abs,
                                                              float* __restrict__ eos, float* __restrict__ eos_abs, int N,
                                                              int K, int T, int V, int C, int out_len, int blank, int first,
                                                              const int32_t* __restrict__ step_dev) {
  if (step_dev) { out_len = *step_dev; first = out_len == 0; }     // replayed graphs: the step counter lives in device memory
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  ctc_prefix_one(logp, lens, r_prev, s_prev, last_tok, (int)cand[(int64_t)n * C + c], r_new, psi, psi_abs, eos, eos_abs, n, c, K, T, V,
                 C, out_len, blank, first);
}

// wave = hypothesis n.  The pre-beam rides in front of the scorer: the C <= 64 best tokens of full[n][0..V) (espnet's
// pre_beam on the weighted full scores; descending, the lower index first among equal scores, V <= 4096) are selected by the
// wave, written to cand [N][C], and lane c runs the prefix recursion of candidate c - two torch.topk launches less per token.
__global__ __launch_bounds__(256) void ctc_prefix_topk_kernel(const float* __restrict__ logp, const int64_t* __restrict__ lens,
                                                              const float* __restrict__ r_prev, const float* __restrict__ s_prev,
                                                              const int64_t* __restrict__ last_tok, const float* __restrict__ full,
                                                              int64_t* __restrict__ cand, float* __restrict__ r_new,
                                                              float* __restrict__ psi, float* __restrict__ psi_abs,
                                                              float* __restrict__ eos, float* __restrict__ eos_abs, int N,
                                                              int K, int T, int V, int C, int out_len, int blank, int first,
                                                              const int32_t* __restrict__ step_dev) {
  if (step_dev) { out_len = *step_dev; first = out_len == 0; }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const float* f = full + (int64_t)n * V;
  uint64_t taken = 0;                 // bit j: element lane + 64 j of this row is already a candidate
  int mytok = 0;
  constexpr int VR = 8;               // rows of up to 512 tokens live in registers: a selection round is shuffles only
  float xv[VR];
  const bool inreg = V <= 64 * VR;
#pragma unroll
  for (int j = 0; j < VR; ++j) xv[j] = (inreg && lane + 64 * j < V) ? f[lane + 64 * j] : -INFINITY;
  for (int c = 0; c < C; ++c) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (inreg) {
#pragma unroll
      for (int j = 0; j < VR; ++j) {
        const int v = lane + 64 * j;
        if (v < V && !((taken >> j) & 1) && (xv[j] > best || (xv[j] == best && v < bi))) { best = xv[j]; bi = v; }
      }
    } else {
      for (int j = 0, v = lane; v < V; ++j, v += 64) {
        const float x = f[v];
        if (!((taken >> j) & 1) && (x > best || (x == best && v < bi))) { best = x; bi = v; }
      }
    }
    {     // the wave's best: highest score, the lower index among equal ones (lane network reductions, common.h)
      const float m = wave_max_dpp(best);
      bi = wave_min_dpp((bi != 0x7fffffff && best == m) ? bi : 0x7fffffff);
    }
    if (bi == 0x7fffffff) bi = 0;     // fewer than C finite-or-not elements cannot happen for C <= V; keeps indices valid
    if ((bi & 63) == lane) taken |= 1ull << (bi >> 6);
    if (lane == c) mytok = bi;
  }
  if (lane < C) {
    cand[(int64_t)n * C + lane] = mytok;
    ctc_prefix_one(logp, lens, r_prev, s_prev, last_tok, mytok, r_new, psi, psi_abs, eos, eos_abs, n, lane, K, T, V, C, out_len, blank,
                   first);
  }
}

// y = [y +] alpha * log_softmax(x) [+ add]: the weighted sum of scorer outputs is built by the scorers' own launches
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                               int64_t ldy, int M, int V, float alpha, float add, int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < V; j += 64) s += expf(xr[j] - mx);
  s = wave_sum(s);
  const float lse = mx + logf(s);
  for (int j = lane; j < V; j += 64) {
    float v = xr[j] - lse;
    if (alpha != 1.f) v *= alpha;
    if (accumulate) v = y[(int64_t)row * ldy + j] + v;
    if (add != 0.f) v += add;
    y[(int64_t)row * ldy + j] = v;
  }
}

// ---- beam update (espnet BatchBeamSearch.search / batch_beam: the index plumbing around the top-k) ------------------
// weighted[n][v] = full[n][v] + w_ctc * ctc_full[n][v] + score[n]  with the partial CTC scorer's row
//   ctc_full[n][v] = -1e10 - s_prev[n], [eos] = eos_s[n], [cand[n][c]] = psi[n][c] (eos_s[n] when the candidate IS <eos>);
// psi_abs[n][c] of an <eos> candidate becomes eos_abs[n] (the state the search stores for it).  One wave per hypothesis.
__global__ __launch_bounds__(256) void beam_combine_kernel(const float* __restrict__ full, const int64_t* __restrict__ cand,
                                                           const float* __restrict__ psi, float* __restrict__ psi_abs,
                                                           const float* __restrict__ eos_s, const float* __restrict__ eos_abs,
                                                           const float* __restrict__ s_prev, const float* __restrict__ score,
                                                           float* __restrict__ weighted, int N, int V, int C, int eos,
                                                           float w_ctc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const float base = -10000000000.0f - s_prev[n], es = eos_s[n], sc = score[n];
  for (int v = lane; v < V; v += 64) {
    float cf = v == eos ? es : base;
    for (int c = 0; c < C; ++c)
      if ((int)cand[(int64_t)n * C + c] == v) cf = v == eos ? es : psi[(int64_t)n * C + c];
    {
      // separately rounded multiply and adds (no fused multiply-add): the scores equal the elementwise torch expression bit for bit
#pragma clang fp contract(off)
      const float prod = w_ctc * cf;
      const float sum = full[(int64_t)n * V + v] + prod;
      weighted[(int64_t)n * V + v] = sum + sc;
    }
  }
  for (int c = lane; c < C; c += 64)
    if ((int)cand[(int64_t)n * C + c] == eos) psi_abs[(int64_t)n * C + c] = eos_abs[n];
}

// K rounds of a wave arg-max over the K V weighted scores of one utterance (in LDS): the K best (slot * V + token) in descending order,
// among equal scores the lower index first.  One wave's work: up to 16 scores per lane in registers (K V <= 1024; the rest stay in
// LDS), a round is an arg-max over the lane's own values and the lane-network reductions - no workgroup barrier inside the rounds.
__device__ __forceinline__ void beam_topk_rounds(float* s_w, int KV, int K, int u, float* __restrict__ top_s,
                                                 int64_t* __restrict__ top_i, int lane) {
  constexpr int RG = 16;
  float xv[RG];
#pragma unroll
  for (int j = 0; j < RG; ++j) xv[j] = lane + 64 * j < KV ? s_w[lane + 64 * j] : __builtin_nanf("");
  for (int r = 0; r < K; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const float x = xv[j];
      const int i = lane + 64 * j;
      if (x == x && (bi == 0x7fffffff || x > bv || (x == bv && i < bi))) { bv = x; bi = i; }     // (NaN: taken / past the end)
    }
    for (int i = lane + 64 * RG; i < KV; i += 64) {
      const float x = s_w[i];
      if (x == x && (bi == 0x7fffffff || x > bv || (x == bv && i < bi))) { bv = x; bi = i; }
    }
    {     // the wave's best: highest score, the lower index among equal ones (lane network reductions, common.h)
      const float lv = bv;
      const int li = bi;
      bv = wave_max_dpp(li == 0x7fffffff ? -INFINITY : lv);
      bi = wave_min_dpp((li != 0x7fffffff && lv == bv) ? li : 0x7fffffff);
    }
    // fewer than K comparable scores (a diverged model: NaN rows): the round has nothing left to pick.  The indices feed
    // beam_reorder's slot arithmetic, so they stay in range - slot 0, token 0 at -inf, a hypothesis that is dead on arrival
    const bool exhausted = bi == 0x7fffffff;
    if (exhausted) { bi = 0; bv = -INFINITY; }
    if (lane == 0) {
      top_s[(int64_t)u * K + r] = bv;
      top_i[(int64_t)u * K + r] = bi;
    }
    if (!exhausted && (bi & 63) == lane) {            // the owner retires the element
      if (bi < 64 * RG) {
#pragma unroll
        for (int j = 0; j < RG; ++j)
          if (j == (bi >> 6)) xv[j] = __builtin_nanf("");
      } else {
        s_w[bi] = __builtin_nanf("");
      }
    }
  }
}

// beam_combine and the top-k behind it in one launch, one workgroup per utterance: the K x V weighted scores of its beam slots
// go to LDS (and to `weighted` when given), then K rounds of a wave arg-max pick the K best (slot * V + token) in descending
// order - among equal scores the lower index first.  Replaces torch.topk's two launches (gather + sort, 25 us of a 130 us tail).
constexpr int kTopkMax = 8192;
__global__ __launch_bounds__(1024) void beam_combine_topk_kernel(const float* __restrict__ full, const int64_t* __restrict__ cand,
                                                                const float* __restrict__ psi, float* __restrict__ psi_abs,
                                                                const float* __restrict__ eos_s, const float* __restrict__ eos_abs,
                                                                const float* __restrict__ s_prev, const float* __restrict__ score,
                                                                float* __restrict__ weighted, float* __restrict__ top_s,
                                                                int64_t* __restrict__ top_i, int K, int V, int C, int eos,
                                                                float w_ctc) {
  __shared__ float s_w[kTopkMax];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x;
  for (int k = wave; k < K; k += 16) {      // (16 waves: every beam slot of the usual widths has a wave of its own)
    const int n = u * K + k;
    const float base = -10000000000.0f - s_prev[n], es = eos_s[n], sc = score[n];
    for (int v = lane; v < V; v += 64) {
      float cf = v == eos ? es : base;
      for (int c = 0; c < C; ++c)
        if ((int)cand[(int64_t)n * C + c] == v) cf = v == eos ? es : psi[(int64_t)n * C + c];
      {
#pragma clang fp contract(off)
        const float prod = w_ctc * cf;
        const float sum = full[(int64_t)n * V + v] + prod;
        const float wv = sum + sc;
        s_w[k * V + v] = wv;
        if (weighted) weighted[(int64_t)n * V + v] = wv;
      }
    }
    for (int c = lane; c < C; c += 64)
      if ((int)cand[(int64_t)n * C + c] == eos) psi_abs[(int64_t)n * C + c] = eos_abs[n];
  }
  __syncthreads();
  if (wave != 0) return;
  beam_topk_rounds(s_w, K * V, K, u, top_s, top_i, lane);
}

// The whole beam update behind the scorers in ONE launch, for vocabularies of up to 64 tokens (character models) whose CTC prefix
// scores were computed for EVERY token beside the scorers (tavsr_ctc_prefix_step with the identity candidate list, on its own
// queue: the recursion over the frames - 20 us - leaves the step's critical path).  One workgroup per utterance, wave k = beam slot k,
// lane v = token v:
//   full = dec + w_lm log_softmax(z_lm) + add                                  (tavsr_log_softmax_rows' accumulate form, same arithmetic)
//   pre-beam: the C best tokens of full (descending, lower index first)        (espnet pre_beam on the weighted full scores)
//   weighted = full + w_ctc * (v == eos ? eos_s : in pre-beam ? psi_all[v] : -1e10 - s_prev) + score      (tavsr_beam_combine)
//   top-K over the K x V weighted scores                                        (tavsr_beam_combine_topk)
// Same values, bit for bit, as those launches in sequence; psi_abs_all[n][eos] = eos_abs[n] (the state stored for an <eos> extension).
__global__ __launch_bounds__(1024) void beam_select_topk_kernel(const float* __restrict__ dec, const float* __restrict__ z_lm,
                                                               float w_lm, float add, const float* __restrict__ psi_all,
                                                               float* __restrict__ psi_abs_all, const float* __restrict__ eos_s,
                                                               const float* __restrict__ eos_abs, const float* __restrict__ s_prev,
                                                               const float* __restrict__ score, float* __restrict__ full_out,
                                                               float* __restrict__ weighted, int64_t* __restrict__ cand_out,
                                                               float* __restrict__ top_s, int64_t* __restrict__ top_i, int K, int V,
                                                               int C, int eos, float w_ctc) {
  __shared__ float s_w[16 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x;
  if (wave < K) {
    const int n = u * K + wave;
    const bool in = lane < V;
    float f = in ? dec[(int64_t)n * V + lane] : -INFINITY;
    if (z_lm) {                    // log_softmax_rows_kernel with alpha = w_lm, accumulate, add - one element per lane
      const float x = in ? z_lm[(int64_t)n * V + lane] : -INFINITY;
      const float mx = wave_max(x);
      const float sm = wave_sum(in ? expf(x - mx) : 0.f);
      const float lse = mx + logf(sm);
      if (in) {
        float v = x - lse;
        if (w_lm != 1.f) v *= w_lm;
        v = f + v;
        if (add != 0.f) v += add;
        f = v;
      }
    }
    if (full_out && in) full_out[(int64_t)n * V + lane] = f;
    // pre-beam: C rounds of the wave's arg-max (ctc_prefix_topk_kernel's selection with the row in one register per lane)
    bool taken = false;
    for (int c = 0; c < C; ++c) {
      const float mine = (in && !taken) ? f : -INFINITY;
      const float m = wave_max_dpp(mine);
      int bi = wave_min_dpp((in && !taken && mine == m) ? lane : 0x7fffffff);
      if (bi == 0x7fffffff) bi = 0;
      if (bi == lane) taken = true;
      if (cand_out && lane == 0) cand_out[(int64_t)n * C + c] = bi;
    }
    if (in) {
      const float base = -10000000000.0f - s_prev[n], es = eos_s[n], sc = score[n];
      const float cf = lane == eos ? es : (taken ? psi_all[(int64_t)n * V + lane] : base);
      {
#pragma clang fp contract(off)
        const float prod = w_ctc * cf;
        const float sum = f + prod;
        const float wv = sum + sc;
        s_w[wave * V + lane] = wv;
        if (weighted) weighted[(int64_t)n * V + lane] = wv;
      }
      if (lane == eos) psi_abs_all[(int64_t)n * V + lane] = eos_abs[n];
    }
  }
  __syncthreads();
  if (wave != 0) return;
  beam_topk_rounds(s_w, K * V, K, u, top_s, top_i, lane);
}

// After the top-k over (beam slot, token) of every utterance (top_i [U][K] = slot * V + token): hypothesis n extends slot
// prev = top_i / V + u * K with token top_i % V.  Gathers the running state of `prev` into the *_out buffers (the state
// arrays are re-ordered, so they cannot be updated in place): CTC forward variables r_new[prev][:, :, cidx] (cidx = the
// candidate column holding the token), log_psi, token history (+ the new token at column *step + 1), ancestor lists.
__global__ __launch_bounds__(256) void beam_reorder_kernel(const int64_t* __restrict__ top_i, const float* __restrict__ top_s,
                                                           const int64_t* __restrict__ cand, const float* __restrict__ r_new,
                                                           const float* __restrict__ psi_abs, const int64_t* __restrict__ yseq,
                                                           const int32_t* __restrict__ anc, float* __restrict__ r_out,
                                                           float* __restrict__ s_out, int64_t* __restrict__ yseq_out,
                                                           int32_t* __restrict__ anc_out, int64_t* __restrict__ tok_out,
                                                           float* __restrict__ score_out, int N, int K, int V, int C, int T,
                                                           int ld_y, int ld_a, const int32_t* __restrict__ step_dev,
                                                           int32_t* __restrict__ hist, int hist_steps,
                                                           const int32_t* __restrict__ maxlen, int eos) {
  const int n = blockIdx.x;
  const int u = n / K;
  const int64_t ti = top_i[n];
  const int prev = (int)(ti / V) + u * K, tk = (int)(ti % V);
  int cidx = 0;                                                   // first candidate column equal to the token (argmax of ==)
  for (int c = C - 1; c >= 0; --c)
    if ((int)cand[(int64_t)prev * C + c] == tk) cidx = c;
  const int step = *step_dev;
  for (int e = threadIdx.x; e < T * 2; e += 256) r_out[(int64_t)n * T * 2 + e] = r_new[((int64_t)prev * T * 2 + e) * C + cidx];
  for (int e = threadIdx.x; e < ld_y; e += 256) yseq_out[(int64_t)n * ld_y + e] = e == step + 1 ? (int64_t)tk : yseq[(int64_t)prev * ld_y + e];
  // maxlen given: the head of the NEXT step rides along (tavsr_beam_step_begin for step + 1: a hypothesis that just took <eos>, or whose
  // utterance has used up its token budget, leaves the beam; column step + 1 of the ancestor list names the next step's own row)
  for (int e = threadIdx.x; e < ld_a; e += 256)
    anc_out[(int64_t)n * ld_a + e] = (maxlen && e == step + 1) ? n + (step + 1) * N : anc[(int64_t)prev * ld_a + e];
  if (threadIdx.x == 0) {
    s_out[n] = psi_abs[(int64_t)prev * C + cidx];
    tok_out[n] = tk;
    score_out[n] = (maxlen && (tk == eos || step + 1 >= maxlen[u])) ? -INFINITY : top_s[n];
    if (hist && step < hist_steps) {          // back-pointer record of this token: (token, extended slot, score bits)
      int32_t* h = hist + (int64_t)step * 3 * N;
      h[n] = tk;
      h[N + n] = prev;
      h[2 * N + n] = __float_as_int(top_s[n]);
    }
  }
}

// Head of a captured search step: the hypotheses that ended with the previous token - their last token is <eos>, or the
// previous iteration was the last one of their utterance (step >= maxlen[u]) - leave the beam (score -inf), and column
// `step` of every ancestor list names this step's own key / value row.  Replaces the host's per-token kill mask upload.
__global__ __launch_bounds__(256) void beam_step_begin_kernel(float* __restrict__ score, const int64_t* __restrict__ tok,
                                                              int32_t* __restrict__ anc, int ld_a, const int32_t* __restrict__ maxlen,
                                                              int N, int K, int eos, const int32_t* __restrict__ step_dev) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int step = *step_dev;
  if (step > 0 && ((int)tok[n] == eos || step >= maxlen[n / K])) score[n] = -INFINITY;
  if (step < ld_a) anc[(int64_t)n * ld_a + step] = n + step * N;
}

// ---- one-token linear layers -----------------------------------------------------------------------------------------
// y[n][c] = res[n][c] + act(LN(x)[n] . W[c] + bias[c]) for the N = utterances x beam current-token rows of a scorer step
// (espnet TransformerDecoder.forward_one_step / TransformerLM.batch_score: every Linear sees ONE row per hypothesis).
// With few rows the step is a chain of ~150 dependent launches, each bound by the latency of its global loads, not by
// arithmetic.  This kernel is built for that regime: LayerNorm, Linear, bias, activation and residual in one launch, ONE
// round trip to memory per launch:
//   block = R rows (R = 16 or 32 >= N) x R output columns, WPB waves; wave w owns k in [w KW, (w + 1) KW), K = WPB KW;
//   lane (r, g) issues all its float4 loads of x[r][.], W[col0 + r][.] (and gamma / beta) up front - they are the operands
//   of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (exact fp32);
//   the LayerNorm statistics come from the same registers (the block's waves hold whole rows between them): sum, then
//   squared deviations, each reduced over the lane groups by shuffles and over the waves through LDS;
//   the WPB partial tiles meet in LDS and the first R * R threads finish the outputs.
struct RowLinArgs {
  const float* x; int64_t ldx; const int64_t* gather;
  const float *gamma, *beta; float eps;
  const float* W; int64_t ldw; const float* bias;
  const float* res; int64_t ldr; float* out; int64_t ldo;
  int N, K, Nout, act;
  const int64_t* rgather;       // residual rows by index (the LM's input-table rows as the residual of layer 0's output projection)
  // rows held as a SUM of tensors (tavsr_rowlin_parts): x = sum_p x[p * xps ..], res likewise; gridDim.y > 1: block y owns a K
  // slice and writes partial tensor y of the output (stride ops) - the next launch of the chain adds them while it loads
  int xparts; int64_t xps; int rparts; int64_t rps; int64_t ops;
};

template <int R, int WPB, int KW, bool LN>      // LN: LayerNorm prologue compiled in (its gamma / beta operands are 2 KW / G more registers)
__global__ __launch_bounds__(WPB * 64) void rowlin_kernel(RowLinArgs a) {
  constexpr int G = 64 / R;                 // lane groups along k
  constexpr int NJ = KW / (4 * G);          // float4 loads per operand per lane
  constexpr int NACC = R * R / 64;          // accumulator registers (4 / 16)
  typedef float accv __attribute__((ext_vector_type(NACC)));
  __shared__ float s_acc[WPB][NACC][64];
  __shared__ float s_red[2][WPB][R];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & (R - 1), g = lane / R;
  const int col0 = blockIdx.x * R;
  const int row = min(r, a.N - 1), col = min(col0 + r, a.Nout - 1);     // surplus lanes repeat the last row / column
  // (K dealt to gridDim.y blocks: block y owns k in [y WPB KW, (y + 1) WPB KW); no LayerNorm then - it needs whole rows)
  const int kb = blockIdx.y * (WPB * KW) + wave * KW + 4 * g;
  const float* xr = a.x + (a.gather ? a.gather[row] : (int64_t)row) * a.ldx + kb;
  const float* wr = a.W + (int64_t)col * a.ldw + kb;
  float4 av[NJ], bv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(wr + 4 * G * j);
#pragma unroll
  for (int j = 0; j < NJ; ++j) av[j] = *reinterpret_cast<const float4*>(xr + 4 * G * j);
  // the epilogue's operands (bias, up to four residual parts) of this thread's FIRST output element are requested now, with the
  // tiles: they used to be a second, dependent round trip to memory behind the last barrier (~1 us of a ~5 us launch)
  constexpr int PRE = 4;
  float pre_b = 0.f, pre_r[PRE] = {0.f, 0.f, 0.f, 0.f};
  bool pre_ok = false;
  if (threadIdx.x < R * R && blockIdx.y == 0) {
    const int q0 = threadIdx.x >> 6, l0 = threadIdx.x & 63;
    const int rw0 = R == 32 ? (q0 & 3) + 8 * (q0 >> 2) + 4 * (l0 >> 5) : 4 * (l0 >> 4) + q0;
    const int c0 = col0 + (l0 & (R - 1));
    if (rw0 < a.N && c0 < a.Nout) {
      pre_ok = true;
      if (a.bias) pre_b = a.bias[c0];
      if (a.res) {
#pragma unroll
        for (int p = 0; p < PRE; ++p)
          if (p < a.rparts) pre_r[p] = a.res[p * a.rps + (a.rgather ? a.rgather[rw0] : (int64_t)rw0) * a.ldr + c0];
      }
    }
  }
  for (int p = 1; p < a.xparts; ++p) {      // the rows arrive as a sum of partial tensors (same order every time)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(xr + p * a.xps + 4 * G * j);
      av[j].x += t.x; av[j].y += t.y; av[j].z += t.z; av[j].w += t.w;
    }
  }
  if constexpr (LN) {
    float4 gv[NJ], ev[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      gv[j] = *reinterpret_cast<const float4*>(a.gamma + kb + 4 * G * j);
      ev[j] = *reinterpret_cast<const float4*>(a.beta + kb + 4 * G * j);
    }
    // Row statistics with ONE workgroup barrier: every wave reduces its own K slice to (mean, sum of squared deviations from
    // that mean) on the lane network, the WPB pairs meet in LDS and are merged by the equal-count form of Chan's update,
    //   mean = avg_w mean_w,   K var = sum_w M2_w + KW sum_w (mean_w - mean)^2
    // - the deviations are taken from a mean, never from raw second moments (no cancellation), and the second barrier + LDS round
    // of a global two-pass form is gone (1.3 us of a 4.9 us launch went into the statistics).
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) sm += (av[j].x + av[j].y) + (av[j].z + av[j].w);
#pragma unroll
    for (int o = R; o < 64; o <<= 1) sm += __shfl_xor(sm, o, 64);
    const float mw = sm * (1.f / (float)KW);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float dx = av[j].x - mw, dy = av[j].y - mw, dz = av[j].z - mw, dw = av[j].w - mw;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
#pragma unroll
    for (int o = R; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
    if (g == 0) { s_red[0][wave][r] = mw; s_red[1][wave][r] = sq; }
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) mean += s_red[0][w][r];
    mean *= 1.f / (float)WPB;
    float var = 0.f, dm2 = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) {
      const float d = s_red[0][w][r] - mean;
      var += s_red[1][w][r];
      dm2 += d * d;
    }
    var += (float)KW * dm2;
    const float rstd = rsqrtf(var / (float)a.K + a.eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { av[j].x -= mean; av[j].y -= mean; av[j].z -= mean; av[j].w -= mean; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      av[j].x = av[j].x * rstd * gv[j].x + ev[j].x;
      av[j].y = av[j].y * rstd * gv[j].y + ev[j].y;
      av[j].z = av[j].z * rstd * gv[j].z + ev[j].z;
      av[j].w = av[j].w * rstd * gv[j].w + ev[j].w;
    }
  }
  accv acc;
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if constexpr (R == 32) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].x, bv[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].y, bv[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].z, bv[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].w, bv[j].w, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < NACC; ++q) s_acc[wave][q][lane] = acc[q];
  __syncthreads();
  // accumulator register q of lane l: 32x32 tile - row (q & 3) + 8 (q >> 2) + 4 (l >> 5), column l & 31;
  //                                   16x16 tile - row 4 (l >> 4) + q, column l & 15
  for (int o = threadIdx.x; o < R * R; o += WPB * 64) {
    const int q = o >> 6, l = o & 63;
    const int rw = R == 32 ? (q & 3) + 8 * (q >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + q;
    const int c = col0 + (l & (R - 1));
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) v += s_acc[w][q][l];
    if (rw < a.N && c < a.Nout) {
      if (blockIdx.y == 0) {          // (a K slice other than the first carries neither bias nor residual; no activation when split)
        const bool pre = pre_ok && o == (int)threadIdx.x;      // the element whose operands came with the tiles
        if (a.bias) v += pre ? pre_b : a.bias[c];
        v = act_fwd(a.act, v);
        if (a.res) {
          int p0 = 0;
          if (pre) {                  // same order of additions as the loop below (parts 0, 1, 2, ...)
#pragma unroll
            for (int p = 0; p < PRE; ++p)
              if (p < a.rparts) v += pre_r[p];
            p0 = PRE;
          }
          for (int p = p0; p < a.rparts; ++p) v += a.res[p * a.rps + (a.rgather ? a.rgather[rw] : (int64_t)rw) * a.ldr + c];
        }
      }
      a.out[blockIdx.y * a.ops + (int64_t)rw * a.ldo + c] = v;
    }
  }
}

// Launch plan.  Every variant fits its registers (no scratch: a spilled operand is a second trip to memory in a kernel whose whole
// point is ONE): blocks of at most 8 waves (two per SIMD: 256 VGPRs each), K = waves x KW.
//   R = 16 rows: K 64 .. 512 as K / 64 waves of 64; 1024 as 8 x 128; 2048 as 8 x 256 (no LayerNorm there: its gamma / beta would
//                be 128 registers more - K = 2048 is the projection that closes a feed-forward block);
//   R = 32 rows: K 64 .. 512 as K / 64 waves of 64; 1024 as 8 x 128 without LayerNorm; 2048 only in K slices (tavsr_rowlin_parts);
//   K slices of 256 / 512 / 1024: the plans above (a slice carries no LayerNorm).
#define TAVSR_ROWLIN_GO(RR, W, KWW, LNN) \
  do { hipLaunchKernelGGL((rowlin_kernel<RR, W, KWW, LNN>), grid, dim3(W * 64), 0, st, a); return true; } while (0)
template <int R>
static bool rowlin_launch(const RowLinArgs& a, hipStream_t st, int ksplit = 1) {
  const dim3 grid((unsigned)((a.Nout + R - 1) / R), (unsigned)ksplit);
  const int Ks = a.K / ksplit;      // K of one block
  if (a.gamma) {
    switch (Ks) {
      case 64: TAVSR_ROWLIN_GO(R, 1, 64, true);
      case 128: TAVSR_ROWLIN_GO(R, 2, 64, true);
      case 256: TAVSR_ROWLIN_GO(R, 4, 64, true);
      case 512: TAVSR_ROWLIN_GO(R, 8, 64, true);
      case 1024: if constexpr (R == 16) TAVSR_ROWLIN_GO(16, 8, 128, true); else return false;
      default: return false;
    }
  }
  switch (Ks) {
    case 64: TAVSR_ROWLIN_GO(R, 1, 64, false);
    case 128: TAVSR_ROWLIN_GO(R, 2, 64, false);
    case 256: TAVSR_ROWLIN_GO(R, 4, 64, false);
    case 512: TAVSR_ROWLIN_GO(R, 8, 64, false);
    case 1024: TAVSR_ROWLIN_GO(R, 8, 128, false);
    case 2048: if constexpr (R == 16) TAVSR_ROWLIN_GO(16, 8, 256, false); else return false;
    default: return false;
  }
}
#undef TAVSR_ROWLIN_GO

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = act_fwd(act, x[i]);
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_act_fwd(const float* x, float* y, int64_t n, int32_t act, tavsr_stream_t stream) {
  TAVSR_REQUIRE((x && y) || n <= 0, TAVSR_EINVAL, "act_fwd: null pointer");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}


// which (rows, K, LayerNorm, K slices) combinations have a one-launch plan (rowlin_launch): the host asks before it routes a step here
extern "C" int tavsr_rowlin_ok(int32_t N, int32_t K, int32_t with_ln, int32_t ksplit) {
  if (N < 0 || N > 32 || ksplit < 1 || ksplit > 16 || K <= 0 || K % ksplit) return 0;
  const int Ks = K / ksplit;
  if (ksplit > 1) return !with_ln && (Ks == 256 || Ks == 512 || Ks == 1024);
  if (with_ln) return Ks == 64 || Ks == 128 || Ks == 256 || Ks == 512 || (Ks == 1024 && N <= 16);
  return Ks == 64 || Ks == 128 || Ks == 256 || Ks == 512 || Ks == 1024 || (Ks == 2048 && N <= 16);
}

extern "C" int tavsr_rowlin(const float* x, int64_t ldx, const int64_t* gather, const float* gamma, const float* beta, float eps,
                            const float* W, int64_t ldw, const float* bias, int32_t act, const float* res, int64_t ldr,
                            const int64_t* res_gather, float* out, int64_t ldo, int32_t N, int32_t K, int32_t Nout,
                            tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && W && out, TAVSR_EINVAL, "rowlin: null pointer");
  TAVSR_REQUIRE((gamma == nullptr) == (beta == nullptr), TAVSR_EINVAL, "rowlin: gamma and beta go together");
  TAVSR_REQUIRE(N <= 32, TAVSR_EUNSUPPORTED, "rowlin: at most 32 rows (got %d): larger steps go through tavsr_gemm", N);
  TAVSR_REQUIRE(K == 64 || K == 128 || K == 256 || K == 512 || K == 1024 || K == 2048, TAVSR_EUNSUPPORTED,
                "rowlin: K in {64, 128, 256, 512, 1024, 2048} (got %d)", K);
  TAVSR_REQUIRE(tavsr_rowlin_ok(N, K, gamma != nullptr, 1), TAVSR_EUNSUPPORTED,
                "rowlin: no one-launch plan for %d rows, K = %d%s (tavsr_rowlin_ok; K = 2048 with more than 16 rows goes through "
                "tavsr_rowlin_parts in K slices)", N, K, gamma ? " with LayerNorm" : "");
  TAVSR_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)x | (uintptr_t)W) & 15) == 0 &&
                    (!gamma || (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0),
                TAVSR_EALIGN, "rowlin: rows of x / W (and gamma / beta) must be 16-byte aligned");
  TAVSR_REQUIRE(out != x, TAVSR_EINVAL, "rowlin: out must not alias x");
  if (N <= 0 || Nout <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(!res_gather || res != out, TAVSR_EINVAL, "rowlin: a gathered residual must not alias out");
  RowLinArgs a{x, ldx, gather, gamma, beta, eps, W, ldw, bias, res, ldr, out, ldo, N, K, Nout, act, res_gather, 1, 0, 1, 0, 0};
  const bool ok = N <= 16 ? rowlin_launch<16>(a, (hipStream_t)stream) : rowlin_launch<32>(a, (hipStream_t)stream);
  TAVSR_REQUIRE(ok, TAVSR_EUNSUPPORTED, "rowlin: no kernel for K = %d, %d rows", K, N);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_rowlin_parts(const float* x, int64_t ldx, int32_t x_parts, int64_t x_pstride, const float* gamma,
                                  const float* beta, float eps, const float* W, int64_t ldw, const float* bias, int32_t act,
                                  const float* res, int64_t ldr, int32_t res_parts, int64_t res_pstride, float* out, int64_t ldo,
                                  int32_t ksplit, int64_t out_pstride, int32_t N, int32_t K, int32_t Nout, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && W && out, TAVSR_EINVAL, "rowlin_parts: null pointer");
  TAVSR_REQUIRE((gamma == nullptr) == (beta == nullptr), TAVSR_EINVAL, "rowlin_parts: gamma and beta go together");
  TAVSR_REQUIRE(N >= 0 && N <= 32, TAVSR_EUNSUPPORTED, "rowlin_parts: up to 32 rows (got %d)", N);
  TAVSR_REQUIRE(x_parts >= 1 && x_parts <= 16 && res_parts >= 1 && res_parts <= 16 && ksplit >= 1 && ksplit <= 16, TAVSR_EINVAL,
                "rowlin_parts: 1..16 partial tensors / K slices");
  TAVSR_REQUIRE(x_pstride % 4 == 0 && (x_parts == 1 || x_pstride >= (int64_t)N * ldx) && (res_parts == 1 || !res || res_pstride > 0),
                TAVSR_EINVAL, "rowlin_parts: partial tensors must be whole [N][ld] slabs, 16-byte aligned");
  if (ksplit > 1) {
    TAVSR_REQUIRE(!gamma && act == TAVSR_ACT_NONE, TAVSR_EUNSUPPORTED,
                  "rowlin_parts: a K split carries neither LayerNorm (whole rows) nor an activation (it is not additive)");
    TAVSR_REQUIRE(K % ksplit == 0 && (K / ksplit == 256 || K / ksplit == 512 || K / ksplit == 1024), TAVSR_EUNSUPPORTED,
                  "rowlin_parts: slices of 256, 512 or 1024 (got K = %d, %d slices)", K, ksplit);
    TAVSR_REQUIRE(out_pstride >= (int64_t)N * ldo, TAVSR_EINVAL, "rowlin_parts: output partial tensors overlap");
  } else {
    TAVSR_REQUIRE(K == 64 || K == 128 || K == 256 || K == 512 || K == 1024 || K == 2048, TAVSR_EUNSUPPORTED,
                  "rowlin_parts: K in {64, 128, 256, 512, 1024, 2048} (got %d)", K);
  }
  TAVSR_REQUIRE(tavsr_rowlin_ok(N, K, gamma != nullptr, ksplit), TAVSR_EUNSUPPORTED,
                "rowlin_parts: no one-launch plan for %d rows, K = %d in %d slice(s)%s (tavsr_rowlin_ok)", N, K, ksplit,
                gamma ? " with LayerNorm" : "");
  TAVSR_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && (((uintptr_t)x | (uintptr_t)W) & 15) == 0 &&
                    (!gamma || (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0),
                TAVSR_EALIGN, "rowlin_parts: rows of x / W (and gamma / beta) must be 16-byte aligned");
  TAVSR_REQUIRE(out != x, TAVSR_EINVAL, "rowlin_parts: out must not alias x");
  if (N <= 0 || Nout <= 0) return TAVSR_OK;
  RowLinArgs a{x, ldx, nullptr, gamma, beta, eps, W, ldw, bias, res, ldr, out, ldo, N, K, Nout, act, nullptr,
               x_parts, x_pstride, res_parts, res_pstride, out_pstride};
  const bool ok = N <= 16 ? rowlin_launch<16>(a, (hipStream_t)stream, ksplit) : rowlin_launch<32>(a, (hipStream_t)stream, ksplit);
  TAVSR_REQUIRE(ok, TAVSR_EUNSUPPORTED, "rowlin_parts: no kernel for K = %d in %d slices", K, ksplit);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// tuning / test aid: how a SMALL step (N H <= 256 items) is launched - 3: the plan (by history length: 0 for long ones, else 1);
// 0: four waves per item, keys dealt to the waves; 1: one wave per item, one item per workgroup; 2: one wave per item, four items per
// workgroup (large steps' plan)
static int g_tree_mode = 3;
extern "C" int tavsr_tree_attn_tune(int32_t mode) {
  TAVSR_REQUIRE(mode >= 0 && mode <= 3, TAVSR_EINVAL, "tree_attn_tune: 0 .. 3");
  g_tree_mode = mode;
  return TAVSR_OK;
}


extern "C" int tavsr_tree_attn_step(const float* q, int64_t ldq, const float* kpool, const float* vpool, int64_t ldkv,
                                    const int32_t* anc, int64_t ld_anc, int32_t nkeys, float* out, int64_t ldo, int32_t N,
                                    int32_t H, int32_t dk, float scale, const int32_t* step_dev, const float* k_new,
                                    const float* v_new, int32_t group, tavsr_stream_t stream) {
  TAVSR_REQUIRE(q && kpool && vpool && anc && out, TAVSR_EINVAL, "tree_attn_step: null pointer");
  TAVSR_REQUIRE((k_new == nullptr) == (v_new == nullptr), TAVSR_EINVAL, "tree_attn_step: k_new and v_new go together");
  TAVSR_REQUIRE(nkeys > 0 && nkeys <= kTreeMaxKeys, TAVSR_EUNSUPPORTED, "tree_attn_step: 1..%d keys supported (got %d)",
                kTreeMaxKeys, nkeys);
  TAVSR_REQUIRE(dk % 4 == 0 && dk <= 128 && ldkv % 4 == 0 && ((uintptr_t)kpool & 15) == 0, TAVSR_EALIGN,
                "tree_attn_step: dk %% 4, dk <= 128 and 16-byte aligned key rows are required");
  if (N <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)q | (uintptr_t)out | (uintptr_t)vpool | (uintptr_t)k_new | (uintptr_t)v_new) & 15) == 0,
                TAVSR_EALIGN, "tree_attn_step: q / out / k_new / v_new rows must be 16-byte aligned");
  if (group <= 0 || N % group != 0) group = 1;
  // small steps with LONG histories: four waves per item, keys dealt to the waves (tree_attn_split_kernel).  Its time is flat in
  // the keys (6.0 us at 1 key, 6.5 at 100, 10.1 at 300) where one wave per item grows (3.9 / 7.1 / 13.8): it pays from ~65 keys.  A
  // captured step is launched with its pool capacity as `nkeys` and sees every length up to it, so the choice goes by half of that.
  if (N * H <= 256 && (g_tree_mode == 0 || (g_tree_mode == 3 && (step_dev ? nkeys / 2 : nkeys) > 80))) {
    const dim3 grid((unsigned)(N * H)), block(256);
    if (dk <= 64)
      hipLaunchKernelGGL(tree_attn_split_kernel<16>, grid, block, 0, (hipStream_t)stream, q, ldq, kpool, vpool, ldkv, anc, ld_anc,
                         nkeys, out, ldo, N, H, dk, scale, step_dev, k_new, v_new, const_cast<float*>(kpool), const_cast<float*>(vpool), group);
    else
      hipLaunchKernelGGL(tree_attn_split_kernel<32>, grid, block, 0, (hipStream_t)stream, q, ldq, kpool, vpool, ldkv, anc, ld_anc,
                         nkeys, out, ldo, N, H, dk, scale, step_dev, k_new, v_new, const_cast<float*>(kpool), const_cast<float*>(vpool), group);
    TAVSR_LAUNCH_CHECK();
    return TAVSR_OK;
  }
  // few items (a batch-1 step: 10 hypotheses x 8 heads): one wave per workgroup - 80 CUs bring the rows in instead of 20
  const int wpb = (N * H <= 256 && g_tree_mode != 2) ? 1 : 4;      // (reached with mode 1, 2, or 3 and a short history)
  const dim3 grid((unsigned)((N * H + wpb - 1) / wpb)), block((unsigned)(64 * wpb));
  if (dk <= 64)
    hipLaunchKernelGGL(tree_attn_step_kernel<16>, grid, block, 0, (hipStream_t)stream, q, ldq, kpool, vpool, ldkv, anc, ld_anc,
                       nkeys, out, ldo, N, H, dk, scale, step_dev, k_new, v_new, const_cast<float*>(kpool), const_cast<float*>(vpool), group);
  else
    hipLaunchKernelGGL(tree_attn_step_kernel<32>, grid, block, 0, (hipStream_t)stream, q, ldq, kpool, vpool, ldkv, anc, ld_anc,
                       nkeys, out, ldo, N, H, dk, scale, step_dev, k_new, v_new, const_cast<float*>(kpool), const_cast<float*>(vpool), group);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_kv_append(const float* k, const float* v, int64_t ld_src, float* kpool, float* vpool, int64_t ldkv,
                               int32_t N, int32_t D, int32_t max_steps, const int32_t* step_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(k && v && kpool && vpool && step_dev, TAVSR_EINVAL, "kv_append: null pointer");
  TAVSR_REQUIRE(D % 4 == 0 && ld_src % 4 == 0 && ldkv % 4 == 0 && (((uintptr_t)k | (uintptr_t)v | (uintptr_t)kpool | (uintptr_t)vpool) & 15) == 0,
                TAVSR_EALIGN, "kv_append: rows must be float4-aligned");
  if (N <= 0) return TAVSR_OK;
  const int64_t n4 = (int64_t)N * (D / 4);
  hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k, v, ld_src, kpool,
                     vpool, ldkv, N, D / 4, max_steps, step_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_ctc_prefix_step(const float* logp, const int64_t* lens, const float* r_prev, const float* s_prev,
                                     const int64_t* last_tok, const int64_t* cand, float* r_new, float* psi, float* psi_abs,
                                     float* eos, float* eos_abs, int32_t N, int32_t K, int32_t T, int32_t V, int32_t C,
                                     int32_t out_len, int32_t blank, const int32_t* step_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(logp && lens && cand && r_new && psi && psi_abs && eos && eos_abs, TAVSR_EINVAL, "ctc_prefix_step: null pointer");
  TAVSR_REQUIRE((out_len == 0 && !step_dev) || (r_prev && s_prev && last_tok), TAVSR_EINVAL, "ctc_prefix_step: state needed after <sos>");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0 && C > 0, TAVSR_EINVAL, "ctc_prefix_step: bad sizes");
  hipLaunchKernelGGL(ctc_prefix_step_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logp,
                     lens, r_prev, s_prev, last_tok, cand, r_new, psi, psi_abs, eos, eos_abs, N, K, T, V, C, out_len, blank,
                     out_len == 0 ? 1 : 0, step_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_ctc_prefix_step_topk(const float* logp, const int64_t* lens, const float* r_prev, const float* s_prev,
                                          const int64_t* last_tok, const float* full, int64_t* cand, float* r_new, float* psi,
                                          float* psi_abs, float* eos, float* eos_abs, int32_t N, int32_t K, int32_t T, int32_t V,
                                          int32_t C, int32_t out_len, int32_t blank, const int32_t* step_dev,
                                          tavsr_stream_t stream) {
  TAVSR_REQUIRE(logp && lens && full && cand && r_new && psi && psi_abs && eos && eos_abs, TAVSR_EINVAL,
                "ctc_prefix_step_topk: null pointer");
  TAVSR_REQUIRE((out_len == 0 && !step_dev) || (r_prev && s_prev && last_tok), TAVSR_EINVAL,
                "ctc_prefix_step_topk: state needed after <sos>");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0 && C > 0, TAVSR_EINVAL, "ctc_prefix_step_topk: bad sizes");
  TAVSR_REQUIRE(C <= 64 && C <= V && V <= 4096, TAVSR_EUNSUPPORTED,
                "ctc_prefix_step_topk: up to 64 candidates of up to 4096 tokens (got C = %d, V = %d): select them first and call "
                "tavsr_ctc_prefix_step", C, V);
  hipLaunchKernelGGL(ctc_prefix_topk_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logp, lens, r_prev,
                     s_prev, last_tok, full, cand, r_new, psi, psi_abs, eos, eos_abs, N, K, T, V, C, out_len, blank,
                     out_len == 0 ? 1 : 0, step_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_log_softmax_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t M, int32_t V, float alpha,
                                      float add, int32_t accumulate, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y, TAVSR_EINVAL, "log_softmax_rows: null pointer");
  if (M <= 0 || V <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M,
                     V, alpha, add, accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_beam_combine(const float* full, const int64_t* cand, const float* psi, float* psi_abs, const float* eos_s,
                                  const float* eos_abs, const float* s_prev, const float* score, float* weighted, int32_t N,
                                  int32_t V, int32_t C, int32_t eos, float w_ctc, tavsr_stream_t stream) {
  TAVSR_REQUIRE(full && cand && psi && psi_abs && eos_s && eos_abs && s_prev && score && weighted, TAVSR_EINVAL,
                "beam_combine: null pointer");
  if (N <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(beam_combine_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, full, cand, psi, psi_abs,
                     eos_s, eos_abs, s_prev, score, weighted, N, V, C, eos, w_ctc);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_beam_combine_topk(const float* full, const int64_t* cand, const float* psi, float* psi_abs, const float* eos_s,
                                       const float* eos_abs, const float* s_prev, const float* score, float* weighted, float* top_s,
                                       int64_t* top_i, int32_t N, int32_t K, int32_t V, int32_t C, int32_t eos, float w_ctc,
                                       tavsr_stream_t stream) {
  TAVSR_REQUIRE(full && cand && psi && psi_abs && eos_s && eos_abs && s_prev && score && top_s && top_i, TAVSR_EINVAL,
                "beam_combine_topk: null pointer");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0 && V > 0 && K <= V, TAVSR_EINVAL, "beam_combine_topk: bad sizes");
  TAVSR_REQUIRE((int64_t)K * V <= kTopkMax, TAVSR_EUNSUPPORTED,
                "beam_combine_topk: beam x vocabulary up to %d (got %d x %d): use tavsr_beam_combine and a top-k of your own", kTopkMax,
                K, V);
  hipLaunchKernelGGL(beam_combine_topk_kernel, dim3((unsigned)(N / K)), dim3(1024), 0, (hipStream_t)stream, full, cand, psi, psi_abs,
                     eos_s, eos_abs, s_prev, score, weighted, top_s, top_i, K, V, C, eos, w_ctc);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_beam_select_topk(const float* dec, const float* z_lm, float w_lm, float add, const float* psi_all, float* psi_abs_all,
                                      const float* eos_s, const float* eos_abs, const float* s_prev, const float* score, float* full_out,
                                      float* weighted, int64_t* cand_out, float* top_s, int64_t* top_i, int32_t N, int32_t K, int32_t V,
                                      int32_t C, int32_t eos, float w_ctc, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dec && psi_all && psi_abs_all && eos_s && eos_abs && s_prev && score && top_s && top_i, TAVSR_EINVAL,
                "beam_select_topk: null pointer");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0 && V > 0 && K <= V && C > 0 && C <= V && eos >= 0 && eos < V, TAVSR_EINVAL,
                "beam_select_topk: bad sizes");
  TAVSR_REQUIRE(V <= 64 && K <= 16, TAVSR_EUNSUPPORTED,
                "beam_select_topk: vocabularies of up to 64 tokens and beams of up to 16 (got %d, %d): tavsr_ctc_prefix_step_topk + "
                "tavsr_beam_combine_topk take the rest", V, K);
  hipLaunchKernelGGL(beam_select_topk_kernel, dim3((unsigned)(N / K)), dim3(1024), 0, (hipStream_t)stream, dec, z_lm, w_lm, add, psi_all,
                     psi_abs_all, eos_s, eos_abs, s_prev, score, full_out, weighted, cand_out, top_s, top_i, K, V, C, eos, w_ctc);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_beam_reorder(const int64_t* top_i, const float* top_s, const int64_t* cand, const float* r_new,
                                  const float* psi_abs, const int64_t* yseq, const int32_t* anc, float* r_out, float* s_out,
                                  int64_t* yseq_out, int32_t* anc_out, int64_t* tok_out, float* score_out, int32_t N, int32_t K,
                                  int32_t V, int32_t C, int32_t T, int32_t ld_y, int32_t ld_a, const int32_t* step_dev,
                                  int32_t* hist, int32_t hist_steps, tavsr_stream_t stream) {
  return tavsr_beam_reorder_begin(top_i, top_s, cand, r_new, psi_abs, yseq, anc, r_out, s_out, yseq_out, anc_out, tok_out, score_out, N,
                                  K, V, C, T, ld_y, ld_a, step_dev, hist, hist_steps, nullptr, 0, stream);
}

extern "C" int tavsr_beam_reorder_begin(const int64_t* top_i, const float* top_s, const int64_t* cand, const float* r_new,
                                        const float* psi_abs, const int64_t* yseq, const int32_t* anc, float* r_out, float* s_out,
                                        int64_t* yseq_out, int32_t* anc_out, int64_t* tok_out, float* score_out, int32_t N, int32_t K,
                                        int32_t V, int32_t C, int32_t T, int32_t ld_y, int32_t ld_a, const int32_t* step_dev,
                                        int32_t* hist, int32_t hist_steps, const int32_t* maxlen, int32_t eos, tavsr_stream_t stream) {
  TAVSR_REQUIRE(top_i && top_s && cand && r_new && psi_abs && yseq && anc && r_out && s_out && yseq_out && anc_out && tok_out &&
                    score_out && step_dev, TAVSR_EINVAL, "beam_reorder: null pointer");
  TAVSR_REQUIRE(r_out != r_new && (const int64_t*)yseq_out != yseq && (const int32_t*)anc_out != anc, TAVSR_EINVAL,
                "beam_reorder: the state is re-ordered, outputs must not alias the inputs");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0, TAVSR_EINVAL, "beam_reorder: bad sizes");
  hipLaunchKernelGGL(beam_reorder_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, top_i, top_s, cand, r_new, psi_abs, yseq,
                     anc, r_out, s_out, yseq_out, anc_out, tok_out, score_out, N, K, V, C, T, ld_y, ld_a, step_dev, hist, hist_steps, maxlen,
                     eos);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_beam_step_begin(float* score, const int64_t* tok, int32_t* anc, int32_t ld_a, const int32_t* maxlen, int32_t N,
                                     int32_t K, int32_t eos, const int32_t* step_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(score && tok && anc && maxlen && step_dev, TAVSR_EINVAL, "beam_step_begin: null pointer");
  TAVSR_REQUIRE(N > 0 && K > 0 && N % K == 0, TAVSR_EINVAL, "beam_step_begin: bad sizes");
  hipLaunchKernelGGL(beam_step_begin_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, score, tok, anc, ld_a,
                     maxlen, N, K, eos, step_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
