// Fused multi-head attention core for gfx950 (espnet RelPositionMultiHeadedAttention.forward after the projections,
// MultiHeadedAttention.forward_attention; called at src/encoder/branchformer/encoder_layer.py:196-210 and
// src/encoder/audiovisual/tailored/encoder_layer.py:185-196,232-243; decoder self / source attention of espnet's
// DecoderLayer):
//     scores[i][j] = ((q_i + u) . k_j + (q_i + v) . p[T-1-i+j]) * scale        (second term only with rel-pos)
//     attn         = softmax_j(scores masked to keys j < klens[b] (and j <= i if causal)), 0 on masked keys
//     ctx_i        = sum_j dropout(attn)[i][j] v_j
// in ONE launch: the [H,B,T,T] score / [H,B,T,2T-1] positional matrices never exist in HBM, rel_shift is an index
// offset, the key mask, softmax, dropout and both contractions run on registers.  The backward kernels recompute
// the probabilities from the saved per-row log-sum-exp.
//
// Work split: one wave per (batch b, head h, tile of 32 queries); 4 waves per workgroup (4 query tiles).  Every product
// is computed TRANSPOSED so that the query index sits on the MFMA lane and the contracted / softmax index in the
// accumulator registers (v_mfma_f32_32x32x2_f32, exact fp32):
//     S^T[j][i]   = sum_k K[j][k] qu[i][k]            A = K rows (16-byte loads along k), B = q rows held in registers
//     raw^T[c][i] = sum_k P[c][k] qv[i][k]            c = cb0 + 32*ct + row: a window of T+31 positional rows per query tile
//     bd^T[j][i]  = raw^T[T-1-i+j][i]                 a per-LANE shift along the register axis: one LDS round trip of a
//                                                     32x32 tile ([c][i] image, row stride 32: conflict free both ways)
//     O^T[d][i]  += sum_j V[j][d] P^T[j][i]           the probabilities are used as the B operand straight from the
//                                                     accumulator registers (register r of lane half h is key
//                                                     (r&3) + 8(r>>2) + 4h: MFMA step r contracts keys {rho(r), rho(r)+4})
// so the softmax statistics are lane-local (+ one exchange between the two lane halves), and nothing but the skew goes
// through LDS.  Operands are read straight from global memory (L2): an fp32 MFMA takes 64 cycles and consumes one
// register per operand, one 16-byte load feeds four of them.  Keys are processed in blocks of 128 with the running
// max / sum of an online softmax, so any sequence length works; T <= 128 is a single block.
#include <math.h>
#include <algorithm>

#include "common.h"

namespace tavsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
  const float *q, *k, *v;          // row buffers, head h at columns h*64 .. h*64+63 of a row
  int64_t ldq, ldk, ldv;
  const float* pos;                // [2*T1-1][ldp] projected positional rows (linear_pos(pos_emb)); null: no rel-pos term
  int64_t ldp;
  const float *bias_u, *bias_v;    // [H*64] pos_bias_u / pos_bias_v (null: none)
  const int64_t* klens;            // [B] valid keys (null: T2)
  float* out;                      // [B*T1][ldo] context
  int64_t ldo;
  float* lse;                      // [B*H][T1] log-sum-exp of the scaled, masked scores (+inf for rows without keys)
  int B, H, T1, T2;
  float scale;
  int causal;
  uint32_t thr;                    // dropout threshold (0: off), keep iff philox word >= thr
  float inv_keep;
  const uint64_t* seed;
  uint64_t offset4;                // counter of element 0 (/4)
  // backward only
  const float *dout, *ctx;         // [B*T1][ldo] gradient of / saved context
  float *dq, *dqv;                 // [B*T1][lddq] gradients w.r.t. (q + u) rows and (q + v) rows
  int64_t lddq;
  float *dk, *dv;                  // head-strided like k / v
  int64_t lddk, lddv;
  float* ds_skew;                  // [H][B][T1][ldw] un-shifted score gradients (rel-pos only; zero-initialised by caller)
  int64_t ldw;
};

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }   // accumulator register -> tile row (lane half 0)

// 32 fp32 operands of one 32-row tile for this lane: row `row` of a k-contiguous operand, k = 8g + 4*h2 + jj
__device__ __forceinline__ void load_row32(const float* __restrict__ p, float (&f)[32]) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float4 x = *reinterpret_cast<const float4*>(p + 8 * g);
    f[4 * g + 0] = x.x; f[4 * g + 1] = x.y; f[4 * g + 2] = x.z; f[4 * g + 3] = x.w;
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  // orders this wave's LDS writes before its following LDS reads (other lanes' data); no other wave touches the slot
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int KB = 4;     // key tiles per online-softmax block

template <bool POS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ float scr_all[4][2][32 * 32];       // per wave: ring of two raw^T tiles
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  const int qt = blockIdx.x * 4 + wave;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int i0 = qt * 32;
  if (i0 >= a.T1) return;
  float* scr = &scr_all[wave][0][0];
  const int i = i0 + li, ic = min(i, a.T1 - 1);
  const int nk = a.klens ? (int)min((int64_t)a.T2, a.klens[b]) : a.T2;

  // ---- this wave's queries as the B operand of every score product: qu = q + u, qv = q + v
  float qu[32], qv[32];
  load_row32(a.q + (int64_t)(b * a.T1 + ic) * a.ldq + h * 64 + 4 * h2, qu);
#pragma unroll
  for (int s = 0; s < 32; ++s) qv[s] = qu[s];
  if (a.bias_u) {
    float t[32];
    load_row32(a.bias_u + h * 64 + 4 * h2, t);
#pragma unroll
    for (int s = 0; s < 32; ++s) qu[s] += t[s];
  }
  if (POS && a.bias_v) {
    float t[32];
    load_row32(a.bias_v + h * 64 + 4 * h2, t);
#pragma unroll
    for (int s = 0; s < 32; ++s) qv[s] += t[s];
  }

  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // key tiles that can hold a valid key for some query of this tile
  int nkt = (min(nk, a.T2) + 31) >> 5;
  if (a.causal) nkt = min(nkt, (min(i0 + 31, a.T1 - 1) >> 5) + 1);
  const int W = 2 * a.T1 - 1;
  const int cb0 = a.T1 - 1 - i0 - 31;      // positional row of (tile ct = 0, row 0)
  const float* kbase = a.k + (int64_t)b * a.T2 * a.ldk + h * 64 + 4 * h2;
  const float* vbase = a.v + (int64_t)b * a.T2 * a.ldv + h * 64 + li;
  const float* pbase = POS ? a.pos + h * 64 + 4 * h2 : nullptr;

  // All global loads of a key tile (K rows, the next positional tile, the V rows of the context product) are issued
  // together before its first MFMA: one exposed L2 latency per tile instead of one per product.
  auto raw_load = [&](int ct, float (&pf)[32]) {
    const int c = min(max(cb0 + 32 * ct + li, 0), W - 1);
    load_row32(pbase + (int64_t)c * a.ldp, pf);
  };
  auto raw_mma = [&](int ct, const float (&pf)[32]) {      // raw^T tile ct -> ring slot ct & 1
    f32x16 rt;
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) rt = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[s], qv[s], rt, 0, 0, 0);
    float* dst = scr + (ct & 1) * 1024 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(rho(r) + 4 * h2) * 32] = rt[r];
  };

  for (int kb0 = 0; kb0 < nkt; kb0 += KB) {
    f32x16 st[KB];
    float vv[KB][32];                         // V^T operands of the context product (two 32-wide d tiles per key row)
    if (POS && kb0 == 0) {                    // later blocks: tile kb0 was computed as tile jt + 1 of the previous block
      float pf0[32];
      raw_load(0, pf0);
      raw_mma(0, pf0);
    }
#pragma unroll
    for (int t = 0; t < KB; ++t) {
      const int jt = kb0 + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
      if (jt < nkt) {
        float kf[32], pf[32];
        const int jr = min(jt * 32 + li, a.T2 - 1);
        load_row32(kbase + (int64_t)jr * a.ldk, kf);
        if (POS) raw_load(jt + 1, pf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* vr = vbase + (int64_t)min(jt * 32 + rho(r) + 4 * h2, a.T2 - 1) * a.ldv;
          vv[t][2 * r] = vr[0];
          vv[t][2 * r + 1] = vr[32];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qu[s], st[t], 0, 0, 0);
        if (POS) {
          raw_mma(jt + 1, pf);
          wave_lds_sync();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int x = 31 - li + rho(r) + 4 * h2;             // c - cb0 - 32*jt of (key row, this query)
            st[t][r] += scr[((jt + (x >> 5)) & 1) * 1024 + (x & 31) * 32 + li];
          }
          wave_lds_sync();                                       // the slot of tile jt is overwritten by tile jt + 2
        }
      }
    }
    // ---- scale, mask, block max
    float mb = -INFINITY;
#pragma unroll
    for (int t = 0; t < KB; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (kb0 + t) * 32 + rho(r) + 4 * h2;
        const bool ok = j < nk && (!a.causal || j <= i);
        const float s = ok ? st[t][r] * a.scale : -INFINITY;
        st[t][r] = s;
        mb = fmaxf(mb, s);
      }
    mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
    const float m_new = fmaxf(m_run, mb);
    float alpha = 1.f, lb = 0.f;
    if (m_new == -INFINITY) {            // no valid key so far for this query
#pragma unroll
      for (int t = 0; t < KB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
    } else {
      alpha = __expf(m_run - m_new);     // m_run = -inf -> 0
#pragma unroll
      for (int t = 0; t < KB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __expf(st[t][r] - m_new);      // masked: exp(-inf) = 0
          st[t][r] = p;
          lb += p;
        }
    }
    lb += __shfl_xor(lb, 32, 64);
    l_run = l_run * alpha + lb;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
    // ---- dropout of the probabilities (espnet forward_attention: self.dropout(attn)); the normaliser keeps all of them
    if (a.thr) {
      const uint64_t sd = a.seed[0];
      const int T2p = (a.T2 + 3) & ~3;
      const uint64_t rowc = a.offset4 + ((uint64_t)((int64_t)bh * a.T1 + ic) * (uint64_t)T2p >> 2);
#pragma unroll
      for (int t = 0; t < KB; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int j = (kb0 + t) * 32 + 8 * q4 + 4 * h2;       // 4 consecutive keys: one Philox call
          if (j < nk) {
            const uint64_t ctr = rowc + (uint64_t)(j >> 2);
            uint32_t w[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[t][4 * q4 + e] = w[e] >= a.thr ? st[t][4 * q4 + e] * a.inv_keep : 0.f;
          }
        }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < KB; ++t) {
      if (kb0 + t < nkt) {                  // the tiles whose V rows were loaded (a skipped tile's registers are garbage)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t][2 * r], st[t][r], ot[0], 0, 0, 0);
          ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t][2 * r + 1], st[t][r], ot[1], 0, 0, 0);
        }
      }
    }
  }

  if (i < a.T1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* o = a.out + (int64_t)(b * a.T1 + i) * a.ldo + h * 64 + 4 * h2;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<float4*>(o + d * 32 + 8 * q4) =
            make_float4(ot[d][4 * q4] * inv, ot[d][4 * q4 + 1] * inv, ot[d][4 * q4 + 2] * inv, ot[d][4 * q4 + 3] * inv);
    if (h2 == 0 && a.lse) a.lse[(int64_t)bh * a.T1 + i] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------ backward
// The probabilities are recomputed from the saved log-sum-exp: p = exp(score * scale - lse_i).  With the dropout mask m
// (regenerated from the counter), Pd = p * m / keep:
//     dPd = dO V^T,  D_i = dO_i . ctx_i (= sum_j Pd_ij dPd_ij),  dS = p * (dPd * m / keep - D_i) * scale
//     dQu = dS K,  dQv = skew(dS) P,  dK = dS^T Qu,  dV = Pd^T dO,  ds_skew[i][T-1-i+j] = dS[i][j]
// Two passes, each a wave per 32-row tile, run side by side in one launch (blockIdx.z): contractions over the KEYS want
// the query on the lane (pass 0: dQu, dQv, same orientation as the forward), contractions over the QUERIES want the
// key on the lane (pass 1: dK, dV, and the rows of ds_skew, which feed the positional-projection gradient GEMM).  Each
// pass recomputes the scores in its own orientation instead of transposing tiles through LDS.

// pass 0: wave = query tile; lane = query, registers = keys
template <bool POS>
__device__ __forceinline__ void attn_bwd_dq(const AttnArgs& a, float* scr, int qt, int bh, int lane) {
  const int li = lane & 31, h2 = lane >> 5;
  const int b = bh / a.H, h = bh % a.H;
  const int i0 = qt * 32;
  const int i = i0 + li, ic = min(i, a.T1 - 1);
  const int nk = a.klens ? (int)min((int64_t)a.T2, a.klens[b]) : a.T2;
  float qu[32], qv[32], dof[32];
  load_row32(a.q + (int64_t)(b * a.T1 + ic) * a.ldq + h * 64 + 4 * h2, qu);
#pragma unroll
  for (int s = 0; s < 32; ++s) qv[s] = qu[s];
  if (a.bias_u) {
    float t[32];
    load_row32(a.bias_u + h * 64 + 4 * h2, t);
#pragma unroll
    for (int s = 0; s < 32; ++s) qu[s] += t[s];
  }
  if (POS && a.bias_v) {
    float t[32];
    load_row32(a.bias_v + h * 64 + 4 * h2, t);
#pragma unroll
    for (int s = 0; s < 32; ++s) qv[s] += t[s];
  }
  load_row32(a.dout + (int64_t)(b * a.T1 + ic) * a.ldo + h * 64 + 4 * h2, dof);
  float Di = 0.f;
  {
    float cf[32];
    load_row32(a.ctx + (int64_t)(b * a.T1 + ic) * a.ldo + h * 64 + 4 * h2, cf);
#pragma unroll
    for (int s = 0; s < 32; ++s) Di += dof[s] * cf[s];
    Di += __shfl_xor(Di, 32, 64);
  }
  const float lse = i < a.T1 ? a.lse[(int64_t)bh * a.T1 + i] : INFINITY;     // +inf: p = 0 (padding rows, rows without keys)

  f32x16 gu[2], gv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) gu[d][r] = gv[d][r] = 0.f;

  int nkt = (min(nk, a.T2) + 31) >> 5;
  if (a.causal) nkt = min(nkt, (min(i0 + 31, a.T1 - 1) >> 5) + 1);
  const int W = 2 * a.T1 - 1;
  const int cb0 = a.T1 - 1 - i0 - 31;
  const float* kbase = a.k + (int64_t)b * a.T2 * a.ldk + h * 64;
  const float* vbase = a.v + (int64_t)b * a.T2 * a.ldv + h * 64 + 4 * h2;
  const float* pbase = POS ? a.pos + h * 64 : nullptr;
  float* raw_scr = scr;              // ring of two raw^T tiles
  float* ds_scr = scr + 2048;        // ring of two dS^T tiles

  auto raw_load = [&](int ct, float (&pf)[32]) {
    const int c = min(max(cb0 + 32 * ct + li, 0), W - 1);
    load_row32(pbase + 4 * h2 + (int64_t)c * a.ldp, pf);
  };
  auto raw_mma = [&](int ct, const float (&pf)[32]) {
    f32x16 rt;
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) rt = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[s], qv[s], rt, 0, 0, 0);
    float* dst = raw_scr + (ct & 1) * 1024 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(rho(r) + 4 * h2) * 32] = rt[r];
  };
  auto pt_load = [&](int ct, float (&pt)[32]) {          // P^T operands of dQv for positional tile ct
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = min(max(cb0 + 32 * ct + rho(r) + 4 * h2, 0), W - 1);
      const float* pr = pbase + (int64_t)c * a.ldp + li;
      pt[2 * r] = pr[0];
      pt[2 * r + 1] = pr[32];
    }
  };
  // dQv^T += P^T(tile ct) draw^T(tile ct), draw^T[c][i] = dS^T[c - (T-1-i)][i] gathered from the dS^T tiles ct and ct - 1
  auto dqv_tile = [&](int ct, bool have_cur, bool have_prev, const float (&pt)[32]) {
    f32x16 dr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rho(r) + 4 * h2;
      float v = 0.f;
      if (row >= 31 - li) {
        if (have_cur) v = ds_scr[(ct & 1) * 1024 + (row - 31 + li) * 32 + li];
      } else {
        if (have_prev) v = ds_scr[((ct - 1) & 1) * 1024 + (row + li + 1) * 32 + li];
      }
      dr[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[2 * r], dr[r], gv[0], 0, 0, 0);
      gv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[2 * r + 1], dr[r], gv[1], 0, 0, 0);
    }
  };

  if (POS && nkt > 0) {
    float pf0[32];
    raw_load(0, pf0);
    raw_mma(0, pf0);
  }
  for (int jt = 0; jt < nkt; ++jt) {
    // ---- every global load of this key tile up front: one exposed L2 latency per tile instead of one per product
    float kf[32], vf[32], pf[32], kt[32], pt[32];
    const int jr = min(jt * 32 + li, a.T2 - 1);
    load_row32(kbase + 4 * h2 + (int64_t)jr * a.ldk, kf);
    load_row32(vbase + (int64_t)jr * a.ldv, vf);
    if (POS) raw_load(jt + 1, pf);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* kr = kbase + (int64_t)min(jt * 32 + rho(r) + 4 * h2, a.T2 - 1) * a.ldk + li;
      kt[2 * r] = kr[0];
      kt[2 * r + 1] = kr[32];
    }
    if (POS) pt_load(jt, pt);
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qu[s], st, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 32; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[s], dof[s], dp, 0, 0, 0);
    if (POS) {
      raw_mma(jt + 1, pf);
      wave_lds_sync();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = 31 - li + rho(r) + 4 * h2;
        st[r] += raw_scr[((jt + (x >> 5)) & 1) * 1024 + (x & 31) * 32 + li];
      }
    }
    // dropout mask of this tile's keys (same counters as the forward)
    if (a.thr) {
      const uint64_t sd = a.seed[0];
      const int T2p = (a.T2 + 3) & ~3;
      const uint64_t rowc = a.offset4 + ((uint64_t)((int64_t)bh * a.T1 + ic) * (uint64_t)T2p >> 2);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int j = jt * 32 + 8 * q4 + 4 * h2;
        if (j < nk) {
          const uint64_t ctr = rowc + (uint64_t)(j >> 2);
          uint32_t w[4];
          philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
#pragma unroll
          for (int e = 0; e < 4; ++e) dp[4 * q4 + e] = w[e] >= a.thr ? dp[4 * q4 + e] * a.inv_keep : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jt * 32 + rho(r) + 4 * h2;
      const bool ok = j < nk && (!a.causal || j <= i);
      const float p = ok ? __expf(st[r] * a.scale - lse) : 0.f;
      st[r] = p * (dp[r] - Di) * a.scale;            // dS^T
    }
    // dQu^T += K^T dS^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gu[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[2 * r], st[r], gu[0], 0, 0, 0);
      gu[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[2 * r + 1], st[r], gu[1], 0, 0, 0);
    }
    if (POS) {
      wave_lds_sync();                                // raw reads done; dS^T slot (jt & 1) free (tile jt - 2 consumed)
      float* dst = ds_scr + (jt & 1) * 1024 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(rho(r) + 4 * h2) * 32] = st[r];
      wave_lds_sync();
      dqv_tile(jt, true, jt > 0, pt);
    }
  }
  if (POS && nkt > 0) {
    float pt[32];
    pt_load(nkt, pt);
    wave_lds_sync();
    dqv_tile(nkt, false, true, pt);                   // the positional rows only the last key tile reaches
  }
  if (i < a.T1) {
    float* o = a.dq + (int64_t)(b * a.T1 + i) * a.lddq + h * 64 + 4 * h2;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<float4*>(o + d * 32 + 8 * q4) = make_float4(gu[d][4 * q4], gu[d][4 * q4 + 1], gu[d][4 * q4 + 2], gu[d][4 * q4 + 3]);
    if (POS) {
      float* o2 = a.dqv + (int64_t)(b * a.T1 + i) * a.lddq + h * 64 + 4 * h2;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<float4*>(o2 + d * 32 + 8 * q4) = make_float4(gv[d][4 * q4], gv[d][4 * q4 + 1], gv[d][4 * q4 + 2], gv[d][4 * q4 + 3]);
    }
  }
}

// pass 1: wave = key tile; lane = key, registers = queries
template <bool POS>
__device__ __forceinline__ void attn_bwd_dkv(const AttnArgs& a, float* scr, int jt, int bh, int lane) {
  const int li = lane & 31, h2 = lane >> 5;
  const int b = bh / a.H, h = bh % a.H;
  const int j0 = jt * 32;
  const int j = j0 + li, jc = min(j, a.T2 - 1);
  const int nk = a.klens ? (int)min((int64_t)a.T2, a.klens[b]) : a.T2;
  float kf[32], vf[32];
  load_row32(a.k + (int64_t)(b * a.T2 + jc) * a.ldk + h * 64 + 4 * h2, kf);
  load_row32(a.v + (int64_t)(b * a.T2 + jc) * a.ldv + h * 64 + 4 * h2, vf);
  f32x16 gk[2], gvv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) gk[d][r] = gvv[d][r] = 0.f;
  float* raw_scr = scr;            // two raw tiles [i][c]
  float* row_scr = scr + 2048;     // [0..31] lse, [32..63] D of the current query tile
  const int W = 2 * a.T1 - 1;
  const int nqt = (a.T1 + 31) >> 5;
  const bool live = j0 < nk;       // a key tile without valid keys has zero gradients (still stored)
  const int it0 = a.causal ? (j0 >> 5) : 0;       // causal: queries i >= j
  const float* ubias = a.bias_u ? a.bias_u + h * 64 : nullptr;
  const float* vbias = (POS && a.bias_v) ? a.bias_v + h * 64 : nullptr;
  const float ub0 = ubias ? ubias[li] : 0.f, ub1 = ubias ? ubias[32 + li] : 0.f;
  for (int it = it0; it < nqt && live; ++it) {
    const int i0 = it * 32;
    const int irow = min(i0 + li, a.T1 - 1);
    // ---- every global load of this query tile up front (one exposed L2 latency per tile instead of one per product):
    //      rows i of dO, ctx, q (k-contiguous A operands), the bias rows, the two positional tiles, and the row-contiguous
    //      dO / q operands of the dV / dK products
    float dof[32], cf[32], qf[32], qa[32], pf0[32], pf1[32], dq_[32], qq_[32];
    load_row32(a.dout + (int64_t)(b * a.T1 + irow) * a.ldo + h * 64 + 4 * h2, dof);
    load_row32(a.ctx + (int64_t)(b * a.T1 + irow) * a.ldo + h * 64 + 4 * h2, cf);
    load_row32(a.q + (int64_t)(b * a.T1 + irow) * a.ldq + h * 64 + 4 * h2, qf);
    const int cb = a.T1 - 1 - i0 - 31 + j0;          // raw[i][c]: c = cb + 32*ct' + lane for the two positional tiles touched
    if (POS) {
      load_row32(a.pos + h * 64 + 4 * h2 + (int64_t)min(max(cb + li, 0), W - 1) * a.ldp, pf0);
      load_row32(a.pos + h * 64 + 4 * h2 + (int64_t)min(max(cb + 32 + li, 0), W - 1) * a.ldp, pf1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = min(i0 + rho(r) + 4 * h2, a.T1 - 1);
      const float* dr = a.dout + (int64_t)(b * a.T1 + i) * a.ldo + h * 64 + li;
      const float* qr = a.q + (int64_t)(b * a.T1 + i) * a.ldq + h * 64 + li;
      dq_[2 * r] = dr[0]; dq_[2 * r + 1] = dr[32];
      qq_[2 * r] = qr[0] + ub0; qq_[2 * r + 1] = qr[32] + ub1;
    }
    // per-row statistics of this query tile: lane li (both halves) handles row i0 + li
    {
      float Di = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) Di += dof[s] * cf[s];
      Di += __shfl_xor(Di, 32, 64);
      wave_lds_sync();                                // previous tile's readers are done
      if (h2 == 0) {
        row_scr[li] = i0 + li < a.T1 ? a.lse[(int64_t)bh * a.T1 + i0 + li] : INFINITY;
        row_scr[32 + li] = Di;
      }
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) qa[s] = qf[s];
    if (ubias) {
      float t[32];
      load_row32(ubias + 4 * h2, t);
#pragma unroll
      for (int s = 0; s < 32; ++s) qa[s] += t[s];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kf[s], st, 0, 0, 0);
    if (POS) {
      if (vbias) {
        float t[32];
        load_row32(vbias + 4 * h2, t);
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] += t[s];
      }
      f32x16 rt;
#pragma unroll
      for (int r = 0; r < 16; ++r) rt[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) rt = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s], pf0[s], rt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) raw_scr[(rho(r) + 4 * h2) * 32 + li] = rt[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) rt[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) rt = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s], pf1[s], rt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) raw_scr[1024 + (rho(r) + 4 * h2) * 32 + li] = rt[r];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[s], vf[s], dp, 0, 0, 0);
    wave_lds_sync();
    const uint64_t sd = a.thr ? a.seed[0] : 0;
    const int T2p = (a.T2 + 3) & ~3;
    f32x16 pd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ir = rho(r) + 4 * h2, i = i0 + ir;
      float s = st[r];
      if (POS) {
        const int x = 31 - ir + li;
        s += raw_scr[(x >> 5) * 1024 + ir * 32 + (x & 31)];
      }
      const bool ok = j < nk && i < a.T1 && (!a.causal || j <= i);
      const float p = ok ? __expf(s * a.scale - row_scr[ir]) : 0.f;
      float keep = 1.f;
      if (a.thr) {
        const uint64_t ctr = a.offset4 + ((uint64_t)((int64_t)bh * a.T1 + min(i, a.T1 - 1)) * (uint64_t)T2p >> 2) + (uint64_t)(jc >> 2);
        uint32_t w[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
        const uint32_t ww = (jc & 3) == 0 ? w[0] : (jc & 3) == 1 ? w[1] : (jc & 3) == 2 ? w[2] : w[3];
        keep = ww >= a.thr ? a.inv_keep : 0.f;
      }
      pd[r] = p * keep;
      st[r] = p * (dp[r] * keep - row_scr[32 + ir]) * a.scale;       // dS
      if (POS && a.ds_skew && ok)
        a.ds_skew[(((int64_t)h * a.B + b) * a.T1 + i) * a.ldw + (a.T1 - 1 - i + j)] = st[r];
    }
    // dV^T += dO^T Pd ;  dK^T += Qu^T dS   (A operands: dword loads along d of rows i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gvv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq_[2 * r], pd[r], gvv[0], 0, 0, 0);
      gvv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq_[2 * r + 1], pd[r], gvv[1], 0, 0, 0);
      gk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qq_[2 * r], st[r], gk[0], 0, 0, 0);
      gk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qq_[2 * r + 1], st[r], gk[1], 0, 0, 0);
    }
  }
  if (j < a.T2) {
    float* ok_ = a.dk + (int64_t)(b * a.T2 + j) * a.lddk + h * 64 + 4 * h2;
    float* ov = a.dv + (int64_t)(b * a.T2 + j) * a.lddv + h * 64 + 4 * h2;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        *reinterpret_cast<float4*>(ok_ + d * 32 + 8 * q4) = make_float4(gk[d][4 * q4], gk[d][4 * q4 + 1], gk[d][4 * q4 + 2], gk[d][4 * q4 + 3]);
        *reinterpret_cast<float4*>(ov + d * 32 + 8 * q4) = make_float4(gvv[d][4 * q4], gvv[d][4 * q4 + 1], gvv[d][4 * q4 + 2], gvv[d][4 * q4 + 3]);
      }
  }
}

template <bool POS>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  __shared__ float scr_all[4][4096];       // per wave: 16 KB of skew / row-statistics scratch
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wave;
  float* scr = &scr_all[wave][0];
  if (blockIdx.z == 0) {
    if (tile * 32 < a.T1) attn_bwd_dq<POS>(a, scr, tile, blockIdx.y, lane);
  } else {
    if (tile * 32 < a.T2) attn_bwd_dkv<POS>(a, scr, tile, blockIdx.y, lane);
  }
}

}  // namespace tavsr

using namespace tavsr;

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Common argument checks of the fused attention entry points; fills the shared part of AttnArgs.
static int attn_args(AttnArgs& a, const tavsr_attn_desc* d, const char* who) {
  TAVSR_REQUIRE(d != nullptr, TAVSR_EINVAL, "%s: null descriptor", who);
  TAVSR_REQUIRE(d->q && d->k && d->v, TAVSR_EINVAL, "%s: null q / k / v", who);
  TAVSR_REQUIRE(d->B > 0 && d->H > 0 && d->T1 > 0 && d->T2 > 0, TAVSR_EINVAL, "%s: bad sizes", who);
  TAVSR_REQUIRE(d->dk == 64, TAVSR_EUNSUPPORTED, "%s: head size 64 only (got %d)", who, d->dk);
  TAVSR_REQUIRE((int64_t)d->B * d->H <= 65535, TAVSR_EUNSUPPORTED, "%s: B*H too large", who);
  TAVSR_REQUIRE(d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && al16(d->q) && al16(d->k) && al16(d->v),
                TAVSR_EALIGN, "%s: q / k / v rows must be 16-byte aligned", who);
  TAVSR_REQUIRE(!d->pos || (d->T1 == d->T2 && d->ldp % 4 == 0 && al16(d->pos)), TAVSR_EINVAL,
                "%s: the rel-pos term needs T1 == T2 and 16-byte aligned positional rows", who);
  TAVSR_REQUIRE((!d->bias_u || al16(d->bias_u)) && (!d->bias_v || al16(d->bias_v)), TAVSR_EALIGN, "%s: unaligned bias", who);
  TAVSR_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f && d->drop_offset % 4 == 0 && (d->p_drop == 0.f || d->seed_dev),
                TAVSR_EINVAL, "%s: dropout needs p in [0, 1), offset %% 4 == 0 and a device seed", who);
  a.q = d->q; a.k = d->k; a.v = d->v;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
  a.pos = d->pos; a.ldp = d->ldp;
  a.bias_u = d->bias_u; a.bias_v = d->bias_v;
  a.klens = d->klens;
  a.B = d->B; a.H = d->H; a.T1 = d->T1; a.T2 = d->T2;
  a.scale = d->scale; a.causal = d->causal;
  a.thr = d->p_drop > 0.f ? (uint32_t)((double)d->p_drop * 4294967296.0) : 0u;
  a.inv_keep = d->p_drop > 0.f ? 1.f / (1.f - d->p_drop) : 1.f;
  a.seed = d->seed_dev; a.offset4 = d->drop_offset / 4;
  return TAVSR_OK;
}

extern "C" int tavsr_attn_fwd(const tavsr_attn_desc* d, float* out, int64_t ldo, float* lse, tavsr_stream_t stream) {
  AttnArgs a{};
  int rc = attn_args(a, d, "attn_fwd");
  if (rc) return rc;
  TAVSR_REQUIRE(out && ldo % 4 == 0 && al16(out), TAVSR_EALIGN, "attn_fwd: context rows must be 16-byte aligned");
  a.out = out; a.ldo = ldo; a.lse = lse;
  dim3 grid(cdiv(cdiv(a.T1, 32), 4), a.B * a.H);
  if (a.pos)
    hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// dq <- d loss / d (q + u) rows, dqv <- d loss / d (q + v) rows (rel-pos only), dk / dv head-strided like k / v, and
// (rel-pos) ds_skew[h][b][i][T-1-i+j] = dS[i][j]: the caller provides it ZERO-FILLED (the kernel writes the band only).
extern "C" int tavsr_attn_bwd(const tavsr_attn_desc* d, const float* dout, const float* ctx, int64_t ldo, const float* lse,
                              float* dq, float* dqv, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                              float* ds_skew, int64_t ldw, tavsr_stream_t stream) {
  AttnArgs a{};
  int rc = attn_args(a, d, "attn_bwd");
  if (rc) return rc;
  TAVSR_REQUIRE(dout && ctx && lse && dq && dk && dv, TAVSR_EINVAL, "attn_bwd: null pointer");
  TAVSR_REQUIRE(!a.pos || (dqv && ds_skew && ldw >= 2 * a.T1 - 1), TAVSR_EINVAL,
                "attn_bwd: the rel-pos form needs dqv and a [H,B,T,>= 2T-1] ds_skew buffer");
  TAVSR_REQUIRE(ldo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && al16(dout) && al16(ctx) && al16(dq) &&
                al16(dk) && al16(dv) && (!dqv || al16(dqv)), TAVSR_EALIGN, "attn_bwd: 16-byte aligned rows required");
  a.dout = dout; a.ctx = ctx; a.ldo = ldo;
  a.lse = const_cast<float*>(lse);
  a.dq = dq; a.dqv = dqv; a.lddq = lddq;
  a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
  a.ds_skew = ds_skew; a.ldw = ldw;
  const int tiles = std::max(cdiv(a.T1, 32), cdiv(a.T2, 32));
  dim3 grid(cdiv(tiles, 4), a.B * a.H, 2);
  if (a.pos)
    hipLaunchKernelGGL(attn_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
