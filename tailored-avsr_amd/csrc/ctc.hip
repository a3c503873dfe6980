// CTC head: loss + gradient (torch.nn.CTCLoss(reduction="none", zero_infinity) as called by
// src/ctc/ctc.py:58-69,143-156) and greedy decode (argmax at ctc.py:180-188 + the collapse at
// src/models/maskctc_model.py:289-291).  One workgroup per utterance; the alpha lattice lives in a
// caller-provided workspace, the beta recursion is fused with the gradient.  Log-space fp32.
#include <float.h>
#include <math.h>

#include "common.h"

namespace tavsr {

__device__ __forceinline__ float log_add(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}
__device__ __forceinline__ float log_add3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// label of lattice state s (2L+1 states): blank at even s, target[(s-1)/2] at odd s
__device__ __forceinline__ int lat_label(const int64_t* tgt, int s, int blank) { return (s & 1) ? (int)tgt[s >> 1] : blank; }

__global__ __launch_bounds__(256) void ctc_loss_kernel(const float* __restrict__ logits, int64_t ld_t, int64_t ld_b,
                                                       const int64_t* __restrict__ hlens,
                                                       const int64_t* __restrict__ targets, int64_t ld_tgt,
                                                       const int64_t* __restrict__ tlens, int blank, int zero_infinity,
                                                       float* __restrict__ loss, float* __restrict__ grad,
                                                       float* __restrict__ ws, int T, int V, int Smax) {
  extern __shared__ float sm[];
  float* s_prev = sm;                 // [Smax]
  float* s_cur = sm + Smax;           // [Smax]
  float* s_ab = sm + 2 * Smax;        // [Smax]   alpha+beta at the current frame
  float* s_lz = sm + 3 * Smax;        // [T]      log-partition of every frame
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Tb = (int)min((int64_t)T, hlens[b]);
  const int L = (int)tlens[b];
  const int S = 2 * L + 1;
  const int64_t* tgt = targets + (int64_t)b * ld_tgt;
  const float* lg = logits + (int64_t)b * ld_b;
  float* gr = grad + (int64_t)b * ld_b;
  float* alpha = ws + (int64_t)b * T * Smax;

  // log-softmax partition per frame
  for (int t = wv; t < Tb; t += 4) {
    const float* row = lg + (int64_t)t * ld_t;
    float mx = -FLT_MAX;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int v = lane; v < V; v += 64) se += expf(row[v] - mx);
    se = wave_sum(se);
    if (lane == 0) s_lz[t] = mx + logf(se);
  }
  __syncthreads();

  // ---- alpha
  for (int s = tid; s < S; s += 256) {
    float a = -INFINITY;
    if (Tb > 0 && s < 2) a = lg[lat_label(tgt, s, blank)] - s_lz[0];
    s_prev[s] = a;
    if (Tb > 0) alpha[s] = a;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* row = lg + (int64_t)t * ld_t;
    for (int s = tid; s < S; s += 256) {
      int lab = lat_label(tgt, s, blank);
      float a0 = s_prev[s];
      float a1 = s >= 1 ? s_prev[s - 1] : -INFINITY;
      float a2 = (s >= 2 && (s & 1) && lab != lat_label(tgt, s - 2, blank)) ? s_prev[s - 2] : -INFINITY;
      float a = log_add3(a0, a1, a2);
      if (a != -INFINITY) a += row[lab] - s_lz[t];
      s_cur[s] = a;
      alpha[(int64_t)t * Smax + s] = a;
    }
    __syncthreads();
    float* tmp = s_prev; s_prev = s_cur; s_cur = tmp;
  }
  float ll = -INFINITY;
  if (Tb > 0) ll = log_add(s_prev[S - 1], S >= 2 ? s_prev[S - 2] : -INFINITY);
  const float nll = -ll;
  const bool inf = !(nll < INFINITY);
  if (tid == 0) loss[b] = (inf && zero_infinity) ? 0.f : nll;
  __syncthreads();

  // ---- beta fused with the gradient.  grad[t][v] = softmax[t][v] - exp(lse_{s:lab(s)=v}(alpha+beta) + nll - lp[t][v])
  for (int s = tid; s < S; s += 256) {
    float bt = -INFINITY;
    if (Tb > 0 && s >= S - 2) bt = lg[(int64_t)(Tb - 1) * ld_t + lat_label(tgt, s, blank)] - s_lz[Tb - 1];
    s_prev[s] = bt;
  }
  __syncthreads();
  for (int t = Tb - 1; t >= 0; --t) {
    const float* row = lg + (int64_t)t * ld_t;
    float* grow = gr + (int64_t)t * ld_t;
    if (t < Tb - 1) {
      for (int s = tid; s < S; s += 256) {
        int lab = lat_label(tgt, s, blank);
        float b0 = s_prev[s];
        float b1 = s + 1 < S ? s_prev[s + 1] : -INFINITY;
        float b2 = (s + 2 < S && (s & 1) && lab != lat_label(tgt, s + 2, blank)) ? s_prev[s + 2] : -INFINITY;
        float bt = log_add3(b0, b1, b2);
        if (bt != -INFINITY) bt += row[lab] - s_lz[t];
        s_cur[s] = bt;
      }
      __syncthreads();
      float* tmp = s_prev; s_prev = s_cur; s_cur = tmp;
    }
    for (int s = tid; s < S; s += 256) s_ab[s] = alpha[(int64_t)t * Smax + s] + s_prev[s];
    __syncthreads();
    for (int v = tid; v < V; v += 256) {
      float lp = row[v] - s_lz[t];
      float lse = -INFINITY;
      if (v == blank) {
        for (int s = 0; s < S; s += 2) lse = log_add(lse, s_ab[s]);
      } else {
        for (int s = 1; s < S; s += 2)
          if ((int)tgt[s >> 1] == v) lse = log_add(lse, s_ab[s]);
      }
      float g = expf(lp);
      if (lse != -INFINITY) g -= expf(lse + nll - lp);
      grow[v] = (inf && zero_infinity) ? 0.f : g;
    }
    __syncthreads();
  }
  // frames beyond the utterance get no gradient
  for (int64_t i = (int64_t)Tb * ld_t + tid; i < (int64_t)T * ld_t; i += 256) {
    if ((int)(i % ld_t) < V) gr[i] = 0.f;
  }
}

// Fast path of the same computation when the whole lattice fits in LDS (T*(V + 2*Smax) floats <= ~150 KB; the
// shipped configs: T = 99, V = 41, Smax = 81 -> 80 KB): log-probabilities, alpha and beta live in LDS, the alpha
// step of frame t and the beta step of frame Tb-1-t share one barrier (the two recursions are independent), the
// state posteriors gamma[t][s] = alpha*beta / (P * y) are formed state-parallel, and the gradient of every (t, v)
// is a deterministic in-order scan over the states of its parity - no atomics, no global round trips per frame.
__global__ __launch_bounds__(256) void ctc_loss_lds_kernel(const float* __restrict__ logits, int64_t ld_t, int64_t ld_b,
                                                           const int64_t* __restrict__ hlens,
                                                           const int64_t* __restrict__ targets, int64_t ld_tgt,
                                                           const int64_t* __restrict__ tlens, int blank,
                                                           int zero_infinity, float* __restrict__ loss,
                                                           float* __restrict__ grad, int T, int V, int Smax) {
  extern __shared__ float sm[];
  float* s_lp = sm;                         // [T][V] log-softmax
  float* s_al = s_lp + (size_t)T * V;       // [T][Smax] alpha, later gamma
  float* s_be = s_al + (size_t)T * Smax;    // [T][Smax] beta
  int* s_lab = reinterpret_cast<int*>(s_be + (size_t)T * Smax);   // [Smax]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Tb = (int)min((int64_t)T, hlens[b]);
  const int L = (int)tlens[b];
  const int S = 2 * L + 1;
  const int64_t* tgt = targets + (int64_t)b * ld_tgt;
  const float* lg = logits + (int64_t)b * ld_b;
  float* gr = grad + (int64_t)b * ld_b;

  for (int s = tid; s < S; s += 256) s_lab[s] = lat_label(tgt, s, blank);
  for (int t = wv; t < Tb; t += 4) {
    const float* row = lg + (int64_t)t * ld_t;
    float mx = -FLT_MAX;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int v = lane; v < V; v += 64) se += expf(row[v] - mx);
    se = wave_sum(se);
    const float lz = mx + logf(se);
    for (int v = lane; v < V; v += 64) s_lp[t * V + v] = row[v] - lz;
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    float a = -INFINITY, bt = -INFINITY;
    if (Tb > 0) {
      if (s < 2) a = s_lp[s_lab[s]];
      if (s >= S - 2) bt = s_lp[(Tb - 1) * V + s_lab[s]];
      s_al[s] = a;
      s_be[(Tb - 1) * Smax + s] = bt;
    }
  }
  __syncthreads();
  for (int i = 1; i < Tb; ++i) {
    const int ta = i, tb = Tb - 1 - i;
    for (int s = tid; s < S; s += 256) {
      const int lab = s_lab[s];
      const float* ap = s_al + (ta - 1) * Smax;
      float a = log_add3(ap[s], s >= 1 ? ap[s - 1] : -INFINITY,
                         (s >= 2 && (s & 1) && lab != s_lab[s - 2]) ? ap[s - 2] : -INFINITY);
      if (a != -INFINITY) a += s_lp[ta * V + lab];
      s_al[ta * Smax + s] = a;
      const float* bp = s_be + (tb + 1) * Smax;
      float bt = log_add3(bp[s], s + 1 < S ? bp[s + 1] : -INFINITY,
                          (s + 2 < S && (s & 1) && lab != s_lab[s + 2]) ? bp[s + 2] : -INFINITY);
      if (bt != -INFINITY) bt += s_lp[tb * V + lab];
      s_be[tb * Smax + s] = bt;
    }
    __syncthreads();
  }
  float ll = -INFINITY;
  if (Tb > 0) ll = log_add(s_al[(Tb - 1) * Smax + S - 1], S >= 2 ? s_al[(Tb - 1) * Smax + S - 2] : -INFINITY);
  const float nll = -ll;
  const bool inf = !(nll < INFINITY);
  if (tid == 0) loss[b] = (inf && zero_infinity) ? 0.f : nll;
  // gamma[t][s] = exp(alpha + beta + nll - lp[t][lab(s)])  (a posterior: in [0, 1])
  for (int i = tid; i < Tb * S; i += 256) {
    const int t = i / S, s = i % S;
    const float ab = s_al[t * Smax + s] + s_be[t * Smax + s];
    s_al[t * Smax + s] = (ab == -INFINITY || inf) ? 0.f : expf(ab + nll - s_lp[t * V + s_lab[s]]);
  }
  __syncthreads();
  for (int i = tid; i < Tb * V; i += 256) {
    const int t = i / V, v = i % V;
    const float* gm = s_al + t * Smax;
    float occ = 0.f;
    if (v == blank) {
      for (int s = 0; s < S; s += 2) occ += gm[s];
    } else {
      for (int s = 1; s < S; s += 2) occ += (s_lab[s] == v) ? gm[s] : 0.f;
    }
    gr[(int64_t)t * ld_t + v] = (inf && zero_infinity) ? 0.f : expf(s_lp[i]) - occ;
  }
  // frames beyond the utterance get no gradient
  for (int64_t i = (int64_t)Tb * ld_t + tid; i < (int64_t)T * ld_t; i += 256) {
    if ((int)(i % ld_t) < V) gr[i] = 0.f;
  }
}

// ids[b,t] = argmax_v logits[b,t,v] (lowest index wins ties, like torch.argmax); hyp[b,:n] = collapse
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, int64_t ld_t, int64_t ld_b,
                                                         const int64_t* __restrict__ hlens, int blank,
                                                         int64_t* __restrict__ ids, int64_t* __restrict__ hyp,
                                                         int64_t* __restrict__ hyp_len, int T, int V) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* lg = logits + (int64_t)b * ld_b;
  for (int t = wv; t < T; t += 4) {
    const float* row = lg + (int64_t)t * ld_t;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      float x = row[v];
      // NaN compares as the maximum in torch.argmax
      bool better = (x > best) || (x != x && best == best) || (x == best && v < bi);
      if (bi == 0x7fffffff || better) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ob = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      bool take = (oi != 0x7fffffff) && ((bi == 0x7fffffff) || (ob > best) || (ob != ob && best == best) ||
                                         (ob == best && oi < bi));
      if (take) { best = ob; bi = oi; }
    }
    if (lane == 0) ids[(int64_t)b * T + t] = bi;
  }
  __syncthreads();
  if (hyp && tid == 0) {
    const int Tb = hlens ? (int)min((int64_t)T, hlens[b]) : T;
    int n = 0;
    int64_t prev = -1;
    for (int t = 0; t < Tb; ++t) {
      int64_t c = ids[(int64_t)b * T + t];
      if (c != prev && c != blank) hyp[(int64_t)b * T + n++] = c;
      prev = c;
    }
    for (int t = n; t < T; ++t) hyp[(int64_t)b * T + t] = -1;
    hyp_len[b] = n;
  }
}

// Label-smoothing KL loss (espnet LabelSmoothingLoss, espnet_model.py:175-180,563) fused with its
// gradient and the th_accuracy counters.  One wave per (b, l) row.
__global__ __launch_bounds__(256) void lsm_loss_kernel(const float* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ target, int ignore, float smoothing,
                                                       float* __restrict__ row_loss, float* __restrict__ grad,
                                                       int32_t* __restrict__ correct, int64_t rows, int V) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = logits + r * ld;
  float* g = grad + r * ld;
  const int64_t tg = target[r];
  const bool ign = tg == ignore;
  float mx = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = lane; v < V; v += 64) {
    float xv = x[v];
    if (bi == 0x7fffffff || xv > mx) { mx = xv; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(mx, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || om > mx || (om == mx && oi < bi))) { mx = om; bi = oi; }
  }
  float se = 0.f;
  for (int v = lane; v < V; v += 64) se += expf(x[v] - mx);
  se = wave_sum(se);
  const float lz = mx + logf(se);
  const float conf = 1.f - smoothing, low = smoothing / (float)(V - 1);
  float l = 0.f;
  for (int v = lane; v < V; v += 64) {
    float lp = x[v] - lz;
    float td = (v == (int)tg) ? conf : low;
    if (!ign && td > 0.f) l += td * (logf(td) - lp);
    g[v] = ign ? 0.f : (expf(lp) - td);
  }
  l = wave_sum(l);
  if (lane == 0) {
    row_loss[r] = l;
    if (correct) correct[r] = ign ? -1 : (bi == (int)tg ? 1 : 0);
  }
}

// out[n,:] = table[ids[n],:] * scale + pe[n % L, :]      (decoder embed: Embedding + PositionalEncoding)
// (step_dev: every row takes positional row min(*step_dev, L - 1) - the one-token step of a replayed search graph, whose step
// counter lives in device memory: no index_select launch in front of the scorers)
__global__ void embed_pe_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                const float* __restrict__ pe, float scale, float* __restrict__ out, int64_t total4,
                                int D4, int L, const int32_t* __restrict__ step_dev) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int64_t n = i / D4;
  int c4 = (int)(i % D4);
  const int64_t prow = step_dev ? (int64_t)min(max(*step_dev, 0), L - 1) : n % L;
  float4 e = reinterpret_cast<const float4*>(table)[ids[n] * D4 + c4];
  float4 p = reinterpret_cast<const float4*>(pe)[prow * D4 + c4];
  reinterpret_cast<float4*>(out)[i] = make_float4(e.x * scale + p.x, e.y * scale + p.y, e.z * scale + p.z, e.w * scale + p.w);
}

// dtable[v,:] = scale * sum_{n: ids[n]==v} dout[n,:]   (one block per vocabulary row: deterministic, n ascending)
// The block finds its row's occurrences together: thread t scans the ids of its contiguous chunk, an exclusive scan of the
// counts gives every thread its place in the list (so the list is in ascending n), then the columns sum over the list.
// (Every thread walking all N ids by itself took 110 us at N = 1312, V = 5000.)  Dynamic LDS: N ints.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float scale, float* __restrict__ dtable, int64_t N, int D,
                                                        int accumulate) {
  extern __shared__ int s_list[];
  __shared__ int s_cnt[257];
  const int v = blockIdx.x, t = threadIdx.x;
  const int per = (int)((N + 255) / 256);
  const int64_t n0 = (int64_t)t * per, n1 = min(N, n0 + per);
  int cnt = 0;
  for (int64_t n = n0; n < n1; ++n) cnt += ids[n] == v;
  s_cnt[t + 1] = cnt;
  if (t == 0) s_cnt[0] = 0;
  __syncthreads();
  if (t < 64) {      // inclusive scan of 256 counts by one wave: four per lane, then a wave scan of the lane sums
    int a0 = s_cnt[4 * t + 1], a1 = a0 + s_cnt[4 * t + 2], a2 = a1 + s_cnt[4 * t + 3], a3 = a2 + s_cnt[4 * t + 4];
    int run = a3;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(run, o, 64);
      if (t >= o) run += up;
    }
    const int base = run - a3;
    s_cnt[4 * t + 1] = base + a0; s_cnt[4 * t + 2] = base + a1; s_cnt[4 * t + 3] = base + a2; s_cnt[4 * t + 4] = base + a3;
  }
  __syncthreads();
  int pos = s_cnt[t];
  for (int64_t n = n0; n < n1; ++n)
    if (ids[n] == v) s_list[pos++] = (int)n;
  __syncthreads();
  const int total = s_cnt[256];
  for (int c = t; c < D; c += 256) {
    float acc = 0.f;
    for (int i = 0; i < total; ++i) acc += dout[(int64_t)s_list[i] * D + c];
    acc *= scale;
    dtable[(int64_t)v * D + c] = accumulate ? dtable[(int64_t)v * D + c] + acc : acc;
  }
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int64_t tavsr_ctc_loss_ws(int32_t B, int32_t T, int32_t Lmax) { return (int64_t)B * T * (2 * Lmax + 1); }

extern "C" int tavsr_ctc_loss(const float* logits, int64_t ld_t, int64_t ld_b, const int64_t* hlens,
                              const int64_t* targets, int64_t ld_tgt, const int64_t* tlens, int32_t blank,
                              int32_t zero_infinity, float* loss, float* grad, float* ws, int32_t B, int32_t T,
                              int32_t V, int32_t Lmax, tavsr_stream_t stream) {
  TAVSR_REQUIRE(logits && hlens && targets && tlens && loss && grad && ws, TAVSR_EINVAL, "ctc_loss: null pointer");
  if (B <= 0) return TAVSR_OK;
  const int Smax = 2 * Lmax + 1;
  const size_t lds_fast = ((size_t)T * (V + 2 * (size_t)Smax) + Smax) * sizeof(float);
  if (lds_fast <= 150 * 1024) {
    static bool attr_set = false;   // > 64 KB of dynamic LDS needs the opt-in once per process
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_loss_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      TAVSR_REQUIRE(e == hipSuccess, (int)e, "ctc_loss: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      attr_set = true;
    }
    hipLaunchKernelGGL(ctc_loss_lds_kernel, dim3(B), dim3(256), lds_fast, (hipStream_t)stream, logits, ld_t, ld_b, hlens,
                       targets, ld_tgt, tlens, blank, zero_infinity, loss, grad, T, V, Smax);
    TAVSR_LAUNCH_CHECK();
    return TAVSR_OK;
  }
  size_t lds = (3 * (size_t)Smax + T) * sizeof(float);
  TAVSR_REQUIRE(lds <= 64000, TAVSR_EUNSUPPORTED, "ctc_loss: T=%d / Lmax=%d exceed the LDS lattice budget", T, Lmax);
  hipLaunchKernelGGL(ctc_loss_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logits, ld_t, ld_b, hlens, targets,
                     ld_tgt, tlens, blank, zero_infinity, loss, grad, ws, T, V, Smax);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_ctc_greedy(const float* logits, int64_t ld_t, int64_t ld_b, const int64_t* hlens, int32_t blank,
                                int64_t* ids, int64_t* hyp, int64_t* hyp_len, int32_t B, int32_t T, int32_t V,
                                tavsr_stream_t stream) {
  TAVSR_REQUIRE(logits && ids, TAVSR_EINVAL, "ctc_greedy: null pointer");
  TAVSR_REQUIRE(!hyp || hyp_len, TAVSR_EINVAL, "ctc_greedy: hyp needs hyp_len");
  if (B <= 0 || T <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld_t, ld_b, hlens, blank, ids,
                     hyp, hyp_len, T, V);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_lsm_loss(const float* logits, int64_t ld, const int64_t* target, int32_t ignore, float smoothing,
                              float* row_loss, float* grad, int32_t* correct, int64_t rows, int32_t V,
                              tavsr_stream_t stream) {
  TAVSR_REQUIRE(logits && target && row_loss && grad, TAVSR_EINVAL, "lsm_loss: null pointer");
  if (rows <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(lsm_loss_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, ignore,
                     smoothing, row_loss, grad, correct, rows, V);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_embed_pe(const int64_t* ids, const float* table, const float* pe, float scale, float* out,
                              int64_t N, int32_t L, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ids && table && pe && out, TAVSR_EINVAL, "embed_pe: null pointer");
  TAVSR_REQUIRE(D % 4 == 0 && L > 0, TAVSR_EINVAL, "embed_pe: D %% 4 == 0 required");
  int64_t total4 = N * (D / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(embed_pe_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, ids, table, pe, scale,
                     out, total4, D / 4, L, (const int32_t*)nullptr);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_embed_pe_step(const int64_t* ids, const float* table, const float* pe, float scale, float* out,
                                   int64_t N, int32_t L, int32_t D, const int32_t* step_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ids && table && pe && out && step_dev, TAVSR_EINVAL, "embed_pe_step: null pointer");
  TAVSR_REQUIRE(D % 4 == 0 && L > 0, TAVSR_EINVAL, "embed_pe_step: D %% 4 == 0 required");
  int64_t total4 = N * (D / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(embed_pe_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, ids, table, pe, scale,
                     out, total4, D / 4, L, step_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_embed_bwd(const int64_t* ids, const float* dout, float scale, float* dtable, int64_t N, int32_t V,
                               int32_t D, int32_t accumulate, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ids && dout && dtable, TAVSR_EINVAL, "embed_bwd: null pointer");
  if (V <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(N >= 0, TAVSR_EINVAL, "embed_bwd: N < 0");
  // the kernel keeps a launch's token ids in LDS: longer batches go through in chunks of 15000 tokens that accumulate
  // into dtable (chunk by chunk in token order: deterministic; one rounding per chunk instead of one per call)
  constexpr int64_t kChunk = 15000;
  int64_t done = 0;
  do {
    const int64_t n = N - done < kChunk ? N - done : kChunk;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V), dim3(256), (size_t)(n > 0 ? n : 1) * sizeof(int), (hipStream_t)stream, ids + done,
                       dout + done * D, scale, dtable, n, D, (done > 0 || accumulate) ? 1 : 0);
    TAVSR_LAUNCH_CHECK();
    done += n;
  } while (done < N);
  return TAVSR_OK;
}
