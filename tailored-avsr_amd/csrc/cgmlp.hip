// cgMLP spatial gating unit and branch-merge kernels.
//   ConvolutionalSpatialGatingUnit (espnet2/asr/layers/cgmlp.py, called at
//   src/encoder/branchformer/encoder_layer.py:220): x_r * (depthwise_conv_k(LN(x_g)) + bias)
//   learned_ave merge (encoder_layer.py:232-293): attention pooling + softmax over the two branches.
// HBM-bound, time-major tiles staged through LDS with the conv halo; parameter gradients go through
// per-block partial slabs + tavsr_sum_partials (deterministic, no atomics).
#include <float.h>

#include "common.h"

namespace tavsr {

constexpr int CG_CH = 64;    // channels per block (lanes -> consecutive channels: coalesced, LDS conflict free)
constexpr int CG_TT = 128;   // time steps per block
constexpr int CG_KMAX = 63;  // max depthwise kernel size

// out[b,t,c] = r[b,t,c] * (bias[c] + sum_k w[c,k] * gn[b,t+k-pad,c]) ; conv (pre-gate) optionally saved
__global__ __launch_bounds__(256) void dwconv_gate_fwd_kernel(const float* __restrict__ gn, const float* __restrict__ r,
                                                              int64_t ldr, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              float* __restrict__ conv, int B, int T, int C, int K) {
  extern __shared__ float sm[];
  const int pad = (K - 1) / 2;
  float* s_x = sm;                             // [(CG_TT + K - 1)][CG_CH]
  float* s_w = sm + (CG_TT + K - 1) * CG_CH;   // [K][CG_CH]
  const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.x * CG_CH, t0 = blockIdx.y * CG_TT, b = blockIdx.z;
  const int c = c0 + cx;
  const int rows = CG_TT + K - 1;
  for (int ib = ty * 4; ib < rows; ib += 16) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ib + q, t = t0 + i - pad;
      v[q] = (i < rows && t >= 0 && t < T && c < C) ? gn[((int64_t)b * T + t) * C + c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ib + q < rows) s_x[(ib + q) * CG_CH + cx] = v[q];
  }
  for (int k = ty; k < K; k += 4) s_w[k * CG_CH + cx] = c < C ? w[(int64_t)c * K + k] : 0.f;
  __syncthreads();
  if (c >= C) return;
  const float bv = bias[c];
  for (int tt = ty; tt < CG_TT; tt += 4) {
    int t = t0 + tt;
    if (t >= T) break;
    int64_t m = (int64_t)b * T + t;
    const float rv = r[m * ldr + c];      // issued before the tap loop: its latency hides under the FMAs
    float a0 = bv, a1 = 0.f;
    int k = 0;
    for (; k + 1 < K; k += 2) {
      a0 += s_w[k * CG_CH + cx] * s_x[(tt + k) * CG_CH + cx];
      a1 += s_w[(k + 1) * CG_CH + cx] * s_x[(tt + k + 1) * CG_CH + cx];
    }
    if (k < K) a0 += s_w[k * CG_CH + cx] * s_x[(tt + k) * CG_CH + cx];
    const float acc = a0 + a1;
    if (conv) conv[m * C + c] = acc;
    out[m * C + c] = rv * acc;
  }
}

// Same, with the kernel size a compile-time constant (the recipe's 31): the taps live in registers and four consecutive
// outputs share one sliding window of LDS reads (K + 3 reads for 4 outputs instead of 8 K).
template <int K>
__global__ __launch_bounds__(256) void dwconv_gate_fwd_kfix_kernel(const float* __restrict__ gn, const float* __restrict__ r,
                                                                   int64_t ldr, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                                   float* __restrict__ conv, int B, int T, int C) {
  extern __shared__ float sm[];
  constexpr int pad = (K - 1) / 2, rows = CG_TT + K - 1;
  float* s_x = sm;                             // [rows][CG_CH]
  const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.x * CG_CH, t0 = blockIdx.y * CG_TT, b = blockIdx.z;
  const int c = c0 + cx;
  for (int ib = ty * 4; ib < rows; ib += 16) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ib + q, t = t0 + i - pad;
      v[q] = (i < rows && t >= 0 && t < T && c < C) ? gn[((int64_t)b * T + t) * C + c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ib + q < rows) s_x[(ib + q) * CG_CH + cx] = v[q];
  }
  float wr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wr[k] = c < C ? w[(int64_t)c * K + k] : 0.f;
  __syncthreads();
  if (c >= C) return;
  const float bv = bias[c];
  constexpr int RPW = CG_TT / 4;               // rows per wave, contiguous
  for (int g = 0; g < RPW / 4; ++g) {
    const int tb = ty * RPW + g * 4;
    if (t0 + tb >= T) break;
    float rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rv[q] = (t0 + tb + q < T) ? r[((int64_t)b * T + t0 + tb + q) * ldr + c] : 0.f;
    float o[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int u = 0; u < K + 3; ++u) {
      const float v = s_x[(tb + u) * CG_CH + cx];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (u - q >= 0 && u - q < K) o[q] += wr[u - q] * v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + tb + q;
      if (t < T) {
        const int64_t m = (int64_t)b * T + t;
        if (conv) conv[m * C + c] = o[q];
        out[m * C + c] = rv[q] * o[q];
      }
    }
  }
}

// The whole CSGU between the two channel projections in one launch (espnet ConvolutionalSpatialGatingUnit.forward: x_r, x_g =
// chunk(2); x_g = norm(x_g); x_g = depthwise conv over time; out = dropout(x_r * x_g)), given the LayerNorm statistics of the
// gate rows: block (64 channels, utterance) normalises its [T x 64] gate tile on the way into LDS (16-byte loads, all in flight
// together), convolves along time with the taps in registers and multiplies by x_r - the normalised gate never makes a round
// trip through HBM in eval, and is written once (with the convolution output) when the backward pass needs it.
template <int K>
__global__ __launch_bounds__(256, 2) void csgu_fwd_kernel(const float* __restrict__ g, int64_t ldg, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, float* __restrict__ conv, float* __restrict__ gn_out,
                                                       int B, int T, uint32_t thr, float inv_keep,
                                                       const uint64_t* __restrict__ seed, uint64_t offset4,
                                                       const float* __restrict__ rowstat, int stat_ld, float eps,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  // rowstat != null: the statistics come as per-row partial (sum, sum of squares) pairs of the 64-column tiles of the GEMM
  // that produced g ([rows][stat_ld][2]; the gate half is tiles C / 64 ..): the 16 lanes that load a row fetch one pair each
  // and reduce; block 0 writes mean / rstd out for the backward pass.
  constexpr int pad = (K - 1) / 2, rows = CG_TT + K - 1;
  __shared__ __attribute__((aligned(16))) float s_x[rows * CG_CH];
  __shared__ float s_w[CG_CH * K];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * CG_CH, b = blockIdx.y;
  for (int i = tid; i < CG_CH * K; i += 256) s_w[i] = w[(int64_t)c0 * K + i];      // the 64 x K tap block is contiguous
  // a thread = 4 consecutive channels (16-byte accesses everywhere) x, per pass, one of 16 rows
  const float4 gam = *reinterpret_cast<const float4*>(ln_w + c0 + 4 * (tid & 15));
  const float4 bet = *reinterpret_cast<const float4*>(ln_b + c0 + 4 * (tid & 15));
  const uint64_t sd = thr ? seed[0] : 0;
  // compute phase: a thread = 2 consecutive channels (taps in registers: 2 K of them) x CG_TT / 8 consecutive rows
  constexpr int RP = CG_TT / 8;
  const float2 bv = *reinterpret_cast<const float2*>(bias + c0 + 2 * (tid & 31));
  const int tid_ = tid;
  for (int t0 = 0; t0 < T; t0 += CG_TT) {
    if (t0) __syncthreads();                           // the previous tile has been read
    // lane coordinates the compiler cannot see through: otherwise every tile-invariant address of the body (~150 of them) is
    // hoisted out of this loop and spilled
    int tq_ = tid_;
    asm volatile("" : "+v"(tq_));
    const int l4 = tq_ & 15, rr = tq_ >> 4, cc = c0 + 4 * l4, l2 = tq_ & 31, rg = tq_ >> 5, c2 = c0 + 2 * l2;
    constexpr int NP = (rows + 15) / 16;
    float4 xv[NP];
    float mu[NP], rs[NP];
    // unconditional loads at clamped rows (straight-line code: all of them in flight together); uniform bases, 32-bit offsets
    const float* gb = g + (int64_t)b * T * ldg;
    const float* mb = mean + (int64_t)b * T;
    const float* sb = rstd + (int64_t)b * T;
    const int ldg32 = (int)ldg;
    float2 sp[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int tc = min(max(t0 + rr + 16 * q - pad, 0), T - 1);
      xv[q] = *reinterpret_cast<const float4*>(gb + (tc * ldg32 + C + cc));
      if (rowstat) {
        const int nt = C >> 6;                           // gate tiles (<= 16: one per lane of the row's 16)
        sp[q] = l4 < nt ? *reinterpret_cast<const float2*>(rowstat + ((int64_t)(b * T + tc) * stat_ld + nt + l4) * 2)
                        : make_float2(0.f, 0.f);
      } else {
        mu[q] = mb[tc];
        rs[q] = sb[tc];
      }
    }
    if (rowstat) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        float s1 = sp[q].x, s2 = sp[q].y;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float m_ = s1 / (float)C;
        mu[q] = m_;
        rs[q] = rsqrtf(fmaxf(s2 / (float)C - m_ * m_, 0.f) + eps);
        const int i = rr + 16 * q, t = t0 + i - pad;
        if (mean_out && blockIdx.x == 0 && l4 == 0 && i >= pad && i < pad + CG_TT && t < T) {
          mean_out[(int64_t)b * T + t] = mu[q];
          rstd_out[(int64_t)b * T + t] = rs[q];
        }
      }
    }
    // x_r of the thread's output rows (rows tb .. tb + RP - 1 of the tile), fetched together with the gate tile
    const int tb = rg * RP;
    float2 rv[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) rv[q] = *reinterpret_cast<const float2*>(gb + (min(t0 + tb + q, T - 1) * ldg32 + c2));
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int i = rr + 16 * q, t = t0 + i - pad;
      if (i < rows) {
        const bool ok = t >= 0 && t < T;                // outside the utterance the convolution sees zeros, not beta
        const float4 v = ok ? make_float4((xv[q].x - mu[q]) * rs[q] * gam.x + bet.x, (xv[q].y - mu[q]) * rs[q] * gam.y + bet.y,
                                          (xv[q].z - mu[q]) * rs[q] * gam.z + bet.z, (xv[q].w - mu[q]) * rs[q] * gam.w + bet.w)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&s_x[i * CG_CH + 4 * l4]) = v;
        if (gn_out && ok && i >= pad && i < pad + CG_TT)
          *reinterpret_cast<float4*>(gn_out + ((int64_t)b * T + t) * C + cc) = v;
      }
    }
    __syncthreads();
    float2 wv[K];                                       // (per tile: not live across the load phase)
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = make_float2(s_w[(2 * l2) * K + k], s_w[(2 * l2 + 1) * K + k]);
#pragma unroll
    for (int gi = 0; gi < RP / 4; ++gi) {               // 4 outputs share one sliding window of K + 3 LDS rows
      const int tq = tb + 4 * gi;
      if (t0 + tq >= T) break;
      float2 o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = bv;
#pragma unroll
      for (int u = 0; u < K + 3; ++u) {
        const float2 v = *reinterpret_cast<const float2*>(&s_x[(tq + u) * CG_CH + 2 * l2]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (u - q >= 0 && u - q < K) { o[q].x += wv[u - q].x * v.x; o[q].y += wv[u - q].y * v.y; }
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // a few window rows in flight at a time, not all of them
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = t0 + tq + q;
        if (t < T) {
          const int64_t e = ((int64_t)b * T + t) * C + c2;
          if (conv) *reinterpret_cast<float2*>(conv + e) = o[q];
          float2 y = make_float2(rv[4 * gi + q].x * o[q].x, rv[4 * gi + q].y * o[q].y);
          if (thr) {                                    // tavsr_dropout's mapping: element e = word e & 3 of counter offset4 + e / 4
            const uint64_t ctr = offset4 + ((uint64_t)e >> 2);
            uint32_t r4[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), r4);
            const uint32_t ra = (e & 2) ? r4[2] : r4[0], rb = (e & 2) ? r4[3] : r4[1];
            y.x = ra >= thr ? y.x * inv_keep : 0.f;
            y.y = rb >= thr ? y.y * inv_keep : 0.f;
          }
          *reinterpret_cast<float2*>(out + e) = y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// du -> dr = du*conv ; dconv = du*r ; dgn[t] = sum_k w[c,k]*dconv[t-k+pad] ;
// dw[c,k] = sum_{b,t} dconv[t]*gn[t+k-pad], dbias[c] = sum_{b,t} dconv[t].
// One block per (64 channels, utterance): it walks the utterance in time tiles of CG_TB steps (dconv and gn tiles
// with their conv halo in LDS, lanes = consecutive channels: coalesced and bank-conflict free) and keeps the
// weight-gradient sums in registers - thread (channel c, wave y) owns taps k = y, y+4, ... (tap K is the bias) - so
// there is no cross-thread reduction and one partial slab [C][K+1] per utterance, summed by sum_partials_kernel.
constexpr int CG_TB = 64;                      // time steps per tile in the backward kernel
constexpr int CG_TAPS = (CG_KMAX + 1 + 3) / 4;  // taps per thread
__global__ __launch_bounds__(256) void dwconv_gate_bwd_kernel(const float* __restrict__ du, const float* __restrict__ gn,
                                                              const float* __restrict__ r, int64_t ldr,
                                                              const float* __restrict__ conv,
                                                              const float* __restrict__ w, float* __restrict__ dr,
                                                              int64_t lddr, float* __restrict__ dgn,
                                                              float* __restrict__ part, int B, int T, int C, int K) {
  extern __shared__ float sm[];
  const int pad = (K - 1) / 2;
  const int rows = CG_TB + K - 1;
  float* s_d = sm;                         // dconv with halo [rows][CG_CH]
  float* s_g = s_d + rows * CG_CH;         // gn with halo    [rows][CG_CH]
  float* s_w = s_g + rows * CG_CH;         // [K][CG_CH]
  const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.x * CG_CH, b = blockIdx.y;
  const int c = c0 + cx;
  const bool cok = c < C;
  for (int k = ty; k < K; k += 4) s_w[k * CG_CH + cx] = cok ? w[(int64_t)c * K + k] : 0.f;
  float acc[CG_TAPS];
#pragma unroll
  for (int j = 0; j < CG_TAPS; ++j) acc[j] = 0.f;
  const int ntap = (K + 1 - ty + 3) / 4;   // taps ty, ty+4, ... <= K
  for (int t0 = 0; t0 < T; t0 += CG_TB) {
    __syncthreads();   // previous tile fully consumed (and s_w visible)
    // tile load, four rows per wave in flight (du, r, gn and - for the tile's own rows - conv: dr = du*conv is
    // written here so that the compute loops below touch LDS only)
    for (int ib = ty * 4; ib < rows; ib += 16) {
      float duv[4], rv[4], gv[4], cv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + q, t = t0 + i - pad;
        duv[q] = rv[q] = gv[q] = cv[q] = 0.f;
        if (i < rows && t >= 0 && t < T && cok) {
          const int64_t m = (int64_t)b * T + t;
          duv[q] = du[m * C + c];
          rv[q] = r[m * ldr + c];
          gv[q] = gn[m * C + c];
          if (i >= pad && i < pad + CG_TB) cv[q] = conv[m * C + c];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + q, t = t0 + i - pad;
        if (i < rows) {
          s_d[i * CG_CH + cx] = duv[q] * rv[q];
          s_g[i * CG_CH + cx] = gv[q];
          if (i >= pad && i < pad + CG_TB && t < T && cok) dr[((int64_t)b * T + t) * lddr + c] = duv[q] * cv[q];
        }
      }
    }
    __syncthreads();
    const int nt = min(CG_TB, T - t0);
    if (cok) {
      // data gradients: rows tt = ty, ty+4, ...
      for (int tt = ty; tt < nt; tt += 4) {
        const int64_t m = (int64_t)b * T + t0 + tt;
        float a0 = 0.f, a1 = 0.f;
        int k = 0;
        // dgn[t] = sum_k w[k] * dconv[t - k + pad]  -> LDS row (tt + pad) - k + pad
        for (; k + 1 < K; k += 2) {
          a0 += s_w[k * CG_CH + cx] * s_d[(tt + 2 * pad - k) * CG_CH + cx];
          a1 += s_w[(k + 1) * CG_CH + cx] * s_d[(tt + 2 * pad - k - 1) * CG_CH + cx];
        }
        if (k < K) a0 += s_w[k * CG_CH + cx] * s_d[(tt + 2 * pad - k) * CG_CH + cx];
        dgn[m * C + c] = a0 + a1;
      }
      // weight gradient: acc[j] += dconv[t] * gn[t + k - pad], k = ty + 4j (k == K: bias)
      for (int tt = 0; tt < nt; ++tt) {
        const float dv = s_d[(tt + pad) * CG_CH + cx];
        const float* g = s_g + (tt + ty) * CG_CH + cx;
#pragma unroll
        for (int j = 0; j < CG_TAPS; ++j) {
          if (j < ntap) {
            const int k = ty + 4 * j;
            acc[j] += dv * (k < K ? g[4 * j * CG_CH] : 1.f);
          }
        }
      }
    }
  }
  if (cok) {
    float* pw = part + ((int64_t)b * C + c) * (K + 1);
#pragma unroll
    for (int j = 0; j < CG_TAPS; ++j)
      if (j < ntap) pw[ty + 4 * j] = acc[j];   // [C][K+1]: k == K holds the bias partial
  }
}

// Backward with a compile-time kernel size: taps in registers, four consecutive data gradients per sliding window, and
// the weight gradient in groups of four rows 4 apart (rows r, r+4, r+8, r+12 need gn rows r+ty+4s, s = 0..10: the taps
// this thread owns, k = ty + 4j, line up across them) - 4x fewer LDS reads than the generic kernel.
template <int K>
__global__ __launch_bounds__(256) void dwconv_gate_bwd_kfix_kernel(const float* __restrict__ du, const float* __restrict__ gn,
                                                                   const float* __restrict__ r, int64_t ldr,
                                                                   const float* __restrict__ conv,
                                                                   const float* __restrict__ w, float* __restrict__ dr,
                                                                   int64_t lddr, float* __restrict__ dgn,
                                                                   float* __restrict__ part, int B, int T, int C,
                                                                   const float* __restrict__ zr, int64_t ldz, int act) {
  // zr != null: r = act(zr) is the left half of cgMLP's activated projection; dr is written as the gradient w.r.t. zr
  // (dr * act'(zr)): no activation-backward pass over that half afterwards
  extern __shared__ float sm[];
  constexpr int pad = (K - 1) / 2, rows = CG_TB + K - 1;
  constexpr int NTAP = (K + 1 + 3) / 4;        // taps per thread (tap K is the bias)
  static_assert(CG_TB == 64 && NTAP == 8, "row grouping below assumes 64-step tiles and 8 taps per thread");
  float* s_d = sm;                         // dconv with halo [rows][CG_CH]
  float* s_g = s_d + rows * CG_CH;         // gn with halo    [rows][CG_CH]
  const int cx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.x * CG_CH, b = blockIdx.y;
  const int c = c0 + cx;
  const bool cok = c < C;
  float wr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wr[k] = cok ? w[(int64_t)c * K + k] : 0.f;
  float acc[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j) acc[j] = 0.f;
  for (int t0 = 0; t0 < T; t0 += CG_TB) {
    __syncthreads();
    for (int ib = ty * 4; ib < rows; ib += 16) {
      float duv[4], rv[4], gv[4], cv[4], zv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + q, t = t0 + i - pad;
        duv[q] = rv[q] = gv[q] = cv[q] = zv[q] = 0.f;
        if (i < rows && t >= 0 && t < T && cok) {
          const int64_t m = (int64_t)b * T + t;
          duv[q] = du[m * C + c];
          rv[q] = r[m * ldr + c];
          gv[q] = gn[m * C + c];
          if (i >= pad && i < pad + CG_TB) {
            cv[q] = conv[m * C + c];
            if (zr) zv[q] = zr[m * ldz + c];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + q, t = t0 + i - pad;
        if (i < rows) {
          s_d[i * CG_CH + cx] = duv[q] * rv[q];
          s_g[i * CG_CH + cx] = gv[q];
          if (i >= pad && i < pad + CG_TB && t < T && cok) {
            float v = duv[q] * cv[q];
            if (zr) v *= act_bwd(act, zv[q]);
            dr[((int64_t)b * T + t) * lddr + c] = v;
          }
        }
      }
    }
    __syncthreads();
    const int nt = min(CG_TB, T - t0);
    if (cok) {
      // data gradients: wave ty owns rows [16 ty, 16 ty + 16), four at a time; dgn[tt] = sum_k w[k] * s_d[tt + 2 pad - k]
      for (int g = 0; g < 4; ++g) {
        const int tb = ty * 16 + g * 4;
        if (tb >= nt) break;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = -3; u < K; ++u) {          // LDS row tb + 2 pad - u feeds output q with tap u + q
          const float v = s_d[(tb + 2 * pad - u) * CG_CH + cx];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (u + q >= 0 && u + q < K) o[q] += wr[u + q] * v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (tb + q < nt) dgn[((int64_t)b * T + t0 + tb + q) * C + c] = o[q];
      }
      // weight gradient: acc[j] += dconv[tt] * gn row (tt + ty + 4 j); the bias tap (k == K: ty == 3, j == 7) adds dconv
      for (int m4 = 0; m4 < 4; ++m4) {
#pragma unroll
        for (int r0 = 0; r0 < 4; ++r0) {
          const int rb = m4 * 16 + r0;          // rows rb, rb + 4, rb + 8, rb + 12
          if (rb >= nt) continue;
          float dv[4], G[11];
#pragma unroll
          for (int a = 0; a < 4; ++a) dv[a] = (rb + 4 * a < nt) ? s_d[(rb + 4 * a + pad) * CG_CH + cx] : 0.f;
#pragma unroll
          for (int sidx = 0; sidx < 11; ++sidx) {
            const int row = rb + ty + 4 * sidx;
            G[sidx] = row < rows ? s_g[row * CG_CH + cx] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < NTAP; ++j) {
            const bool is_bias = (ty + 4 * j == K);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[j] += dv[a] * (is_bias ? 1.f : G[a + j]);
          }
        }
      }
    }
  }
  if (cok) {
    float* pw = part + ((int64_t)b * C + c) * (K + 1);
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
      if (ty + 4 * j <= K) pw[ty + 4 * j] = acc[j];
  }
}

// split [C][K+1] summed slab into dw[C][K] and dbias[C]
__global__ void dwconv_split_kernel(const float* __restrict__ sum, float* __restrict__ dw, float* __restrict__ db, int C,
                                    int K, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * (K + 1)) return;
  int c = i / (K + 1), k = i % (K + 1);
  float v = sum[i];
  if (k < K) dw[c * K + k] = accumulate ? dw[c * K + k] + v : v;
  else db[c] = accumulate ? db[c] + v : v;
}

// ------------------------------- learned_ave merge ---------------------------------------------
// One 512-thread block per utterance b: waves 0-3 work on branch 1, waves 4-7 on branch 2, in lock step.
// For branch k in {1,2}:
//   score_k[t] = softmax_t( (x_k[t,:] . wp_k + bp_k) / sqrt(D) ) over t < len, 0 beyond
//   pooled_k   = sum_t score_k[t] x_k[t,:] ;  weight_k = pooled_k . ww_k + bw_k
//   (w1, w2) = softmax(weight_1, weight_2)
struct MergeParams {
  const float* wp[2]; const float* bp[2]; const float* ww[2]; const float* bw[2];
};

// sum over the 256 threads of one half of the block (s_red: 4 floats of that half)
__device__ __forceinline__ float half_sum(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[(threadIdx.x >> 6) & 3] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ float half_max(float v, float* s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[(threadIdx.x >> 6) & 3] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// dot[t] = <x[t,:], v> for the rows of one utterance (x rows are D floats, D % 4 == 0): one wave per row,
// eight rows (eight 16-byte loads per lane) in flight per wave
__device__ __forceinline__ void row_dots(const float* __restrict__ x, const float* __restrict__ v, int len, int T, int D,
                                         int wv, int lane, float bias, float scale, float* __restrict__ out) {
  const int D4 = D >> 2;
  for (int tb = wv * 8; tb < T; tb += 32) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int c4 = lane; c4 < D4; c4 += 64) {
      const float4 wv4 = reinterpret_cast<const float4*>(v)[c4];
      float4 xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = tb + i;
        xv[i] = t < len ? reinterpret_cast<const float4*>(x + (int64_t)t * D)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += (xv[i].x * wv4.x + xv[i].y * wv4.y) + (xv[i].z * wv4.z + xv[i].w * wv4.w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float r = wave_sum(acc[i]);
      if (lane == 0 && tb + i < T) out[tb + i] = (r + bias) * scale;
    }
  }
}

__global__ __launch_bounds__(512) void merge_pool_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                             const int64_t* __restrict__ lens,
                                                             const int64_t* __restrict__ lens2, MergeParams p,
                                                             float* __restrict__ score, float* __restrict__ pooled,
                                                             float* __restrict__ wout, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int k = threadIdx.x >> 8;            // branch of this half
  const int ht = threadIdx.x & 255;          // thread within the half
  const int T2 = (2 * T + 3) & ~3;           // regions start on 16-byte boundaries (float4 LDS accesses below)
  float* s_sc = sm + k * T;                  // [2][T]
  float* s_red = sm + T2 + k * 4;            // [2][4]
  float* s_w = sm + T2 + 8;                  // [2] branch logits
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = ht >> 6;
  const int64_t* lk = (k == 1 && lens2) ? lens2 : lens;     // per-branch valid length (AV fusion: audio / video masks)
  const int len = lk ? (int)min((int64_t)T, lk[b]) : T;
  const float inv_sqrt_d = 1.f / sqrtf((float)D);
  const float* x = (k == 0 ? x1 : x2) + (int64_t)b * T * D;
  row_dots(x, p.wp[k], len, T, D, wv, lane, p.bp[k][0], inv_sqrt_d, s_sc);
  __syncthreads();
  float mx = -FLT_MAX;
  for (int t = ht; t < len; t += 256) mx = fmaxf(mx, s_sc[t]);
  mx = half_max(mx, s_red);
  float sum = 0.f;
  for (int t = ht; t < len; t += 256) sum += expf(s_sc[t] - mx);
  sum = half_sum(sum, s_red);
  const float inv = len > 0 ? 1.f / sum : 0.f;
  __syncthreads();
  for (int t = ht; t < T; t += 256) {
    float v = t < len ? expf(s_sc[t] - mx) * inv : 0.f;
    s_sc[t] = v;
    score[((int64_t)k * B + b) * T + t] = v;
  }
  __syncthreads();
  // pooled_k = sum_t score[t] x[t,:]: the four waves of the half take rows t = wv, wv + 4, ... with 16-byte lanes along the
  // channels (a quarter of the serial chain of a thread-per-channel loop), partial sums meet in LDS
  float* s_part = sm + T2 + 16;                    // [2][4][D]
  {
    const int D4 = D >> 2;
    for (int c4 = lane; c4 < D4; c4 += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = wv; t < len; t += 4) {
        const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)t * D)[c4];
        const float sv = s_sc[t];
        acc.x += sv * xv.x; acc.y += sv * xv.y; acc.z += sv * xv.z; acc.w += sv * xv.w;
      }
      *reinterpret_cast<float4*>(s_part + (k * 4 + wv) * D + 4 * c4) = acc;
    }
  }
  __syncthreads();
  float wacc = 0.f;
  for (int c = ht; c < D; c += 256) {
    const float* pp = s_part + k * 4 * D + c;
    const float acc = (pp[0] + pp[D]) + (pp[2 * D] + pp[3 * D]);
    pooled[((int64_t)k * B + b) * D + c] = acc;
    wacc += acc * p.ww[k][c];
  }
  wacc = half_sum(wacc, s_red);
  if (ht == 0) s_w[k] = wacc + p.bw[k][0];
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = fmaxf(s_w[0], s_w[1]);
    float e0 = expf(s_w[0] - m), e1 = expf(s_w[1] - m);
    wout[b * 2 + 0] = e0 / (e0 + e1);
    wout[b * 2 + 1] = e1 / (e0 + e1);
  }
}


// out[b,t,:] = w[b,0]*x1[b,t,:] + w[b,1]*x2[b,t,:]
__global__ void merge_combine_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                     const float* __restrict__ w, float* __restrict__ out, int64_t total4, int TD4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int b = (int)(i / TD4);
  float w1 = w[b * 2], w2 = w[b * 2 + 1];
  float4 a = reinterpret_cast<const float4*>(x1)[i], c = reinterpret_cast<const float4*>(x2)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(w1 * a.x + w2 * c.x, w1 * a.y + w2 * c.y, w1 * a.z + w2 * c.z,
                                                  w1 * a.w + w2 * c.w);
}

// Backward of pool + combine: one 512-thread block per utterance, one branch per half (as the forward).
// part[b] = { dwp1[D], dwp2[D], dww1[D], dww2[D], dbp1, dbp2, dbw1, dbw2 }  (4*D + 4 floats)
__global__ __launch_bounds__(512) void merge_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ x1,
                                                        const float* __restrict__ x2, const int64_t* __restrict__ lens,
                                                        const int64_t* __restrict__ lens2, MergeParams p,
                                                        const float* __restrict__ score,
                                                        const float* __restrict__ pooled, const float* __restrict__ w,
                                                        float* __restrict__ dx1, float* __restrict__ dx2,
                                                        float* __restrict__ part, int B, int T, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int k = threadIdx.x >> 8, ht = threadIdx.x & 255;
  const int T2 = (2 * T + 3) & ~3;          // regions start on 16-byte boundaries (float4 LDS accesses below)
  float* s_ds = sm + k * T;                 // [2][T]  d(score) then ds_pre
  float* s_dp = sm + T2 + k * D;            // [2][D]  dpooled
  float* s_red = sm + T2 + 2 * D + k * 4;   // [2][4]
  float* s_a = sm + T2 + 2 * D + 8;         // [2][8] per-wave partials of <dm,x1>, <dm,x2>
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = ht >> 6, wv8 = threadIdx.x >> 6;
  const int64_t* lk = (k == 1 && lens2) ? lens2 : lens;
  const int len = lk ? (int)min((int64_t)T, lk[b]) : T;
  const int64_t base = (int64_t)b * T * D;
  const float inv_sqrt_d = 1.f / sqrtf((float)D);
  const float w1 = w[b * 2], w2 = w[b * 2 + 1];
  // dL/dw_k = <dm, x_k>: all 512 threads stream the utterance once (16-byte loads)
  float a1 = 0.f, a2 = 0.f;
  {
    const float4* g4 = reinterpret_cast<const float4*>(dm + base);
    const float4* p1 = reinterpret_cast<const float4*>(x1 + base);
    const float4* p2 = reinterpret_cast<const float4*>(x2 + base);
    const int64_t n4 = (int64_t)T * D / 4;
    for (int64_t i = threadIdx.x; i < n4; i += 512) {
      const float4 g = g4[i], u = p1[i], v = p2[i];
      a1 += (g.x * u.x + g.y * u.y) + (g.z * u.z + g.w * u.w);
      a2 += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
    }
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) { s_a[wv8] = a1; s_a[8 + wv8] = a2; }
  __syncthreads();
  a1 = a2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a1 += s_a[i]; a2 += s_a[8 + i]; }
  const float dotw = w1 * a1 + w2 * a2;
  const float dweight = k == 0 ? w1 * (a1 - dotw) : w2 * (a2 - dotw);
  float* pb = part + (int64_t)b * (4 * D + 4);
  const float* x = (k == 0 ? x1 : x2) + base;
  float* dx = (k == 0 ? dx1 : dx2) + base;
  const float* sc = score + ((int64_t)k * B + b) * T;
  const float* po = pooled + ((int64_t)k * B + b) * D;
  const float wk = k == 0 ? w1 : w2;
  for (int c = ht; c < D; c += 256) {
    s_dp[c] = dweight * p.ww[k][c];
    pb[(2 + k) * D + c] = dweight * po[c];  // dww_k
  }
  __syncthreads();
  // dscore[t] = <dpooled, x[t,:]>
  row_dots(x, s_dp, len, T, D, wv, lane, 0.f, 1.f, s_ds);
  __syncthreads();
  float dot = 0.f;
  for (int t = ht; t < len; t += 256) dot += sc[t] * s_ds[t];
  dot = half_sum(dot, s_red);
  float sb = 0.f;
  __syncthreads();
  for (int t = ht; t < T; t += 256) {
    float v = t < len ? sc[t] * (s_ds[t] - dot) * inv_sqrt_d : 0.f;
    s_ds[t] = v;  // ds_pre
    sb += v;
  }
  sb = half_sum(sb, s_red);  // also orders the s_ds writes before the reads below
  if (ht == 0) {
    pb[4 * D + k] = sb;            // dbp_k
    pb[4 * D + 2 + k] = dweight;   // dbw_k
  }
  // dx_k and dwp_k: the four waves of the half take rows t = wv, wv + 4, ... with 16-byte lanes along the channels;
  // the dwp partials of the waves meet in LDS
  float* s_part = sm + T2 + 2 * D + 32;            // [2][4][D]
  {
    const int D4 = D >> 2;
    for (int c4 = lane; c4 < D4; c4 += 64) {
      const float4 wpc = reinterpret_cast<const float4*>(p.wp[k])[c4];
      const float4 dpc = *reinterpret_cast<const float4*>(s_dp + 4 * c4);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = wv; t < T; t += 4) {
        const int64_t o = (int64_t)t * D + 4 * c4;
        const float4 xv = *reinterpret_cast<const float4*>(x + o);
        const float4 gv = *reinterpret_cast<const float4*>(dm + base + o);
        const float sv = t < len ? sc[t] : 0.f, dsv = s_ds[t];
        *reinterpret_cast<float4*>(dx + o) = make_float4(wk * gv.x + sv * dpc.x + dsv * wpc.x, wk * gv.y + sv * dpc.y + dsv * wpc.y,
                                                          wk * gv.z + sv * dpc.z + dsv * wpc.z, wk * gv.w + sv * dpc.w + dsv * wpc.w);
        acc.x += dsv * xv.x; acc.y += dsv * xv.y; acc.z += dsv * xv.z; acc.w += dsv * xv.w;
      }
      *reinterpret_cast<float4*>(s_part + (k * 4 + wv) * D + 4 * c4) = acc;
    }
  }
  __syncthreads();
  for (int c = ht; c < D; c += 256) {
    const float* pp = s_part + k * 4 * D + c;
    pb[k * D + c] = (pp[0] + pp[D]) + (pp[2 * D] + pp[3 * D]);  // dwp_k
  }
}

// scatter the summed merge-gradient slab {dwp1, dwp2, dww1, dww2 [D each], dbp1, dbp2, dbw1, dbw2} to the eight
// parameter gradients (one launch instead of eight)
struct MergeGradPtrs { float* g[8]; };
__global__ void merge_scatter_kernel(const float* __restrict__ sum, MergeGradPtrs o, int D, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * D + 4) return;
  float* dst = i < 4 * D ? o.g[i / D] + (i % D) : o.g[4 + (i - 4 * D)];
  *dst = accumulate ? *dst + sum[i] : sum[i];
}


// ------------------------------- learned_ave merge, row-parallel form (D = 256) -----------------
// The one-block-per-utterance kernels above run on B (= 32) of the 256 CUs and are bound by what ONE CU can stream
// (~12 GB/s each: 16 us forward, 32 us backward at B = 32, T = 99).  Everything the merge needs from a row is four dot
// products, because  weight_k = <ww_k, sum_t p_k[t] x_k[t,:]> + bw_k = sum_t p_k[t] <ww_k, x_k[t,:]> + bw_k :
//   pass 1 (all rows in parallel)  dots[0..3][row] = <wp_1,x_1>, <wp_2,x_2>, <ww_1,x_1>, <ww_2,x_2>
//   pass 2 (blocks of MR_RPB rows) every block redoes the utterance's two softmaxes over time and the softmax over the
//                                  branches from the dots (4 T floats), then combines its own rows.
// The backward pass has the same shape: pass 1 the row dots <dm, x_k>, pass 2 the per-utterance scalars from the dots, then
// dx_k for the block's rows (optionally already under the branch's dropout mask) and the block's partial parameter gradients.
constexpr int MR_RPB = 16;      // rows per block of the second passes (8 waves x 2 rows)

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// dots[j][row], j < 4 (dm == null) or adots[k][row] = <dm[row], x_k[row]>, k < 2 (dm != null); one wave per two rows
__global__ __launch_bounds__(256) void merge_rowdots_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                            const float* __restrict__ dm, MergeParams p,
                                                            float* __restrict__ dots, int M) {
  constexpr int D = 256;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 4 + wv) * 2;
  float4 a[2], c[2], g[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t o = (int64_t)min(r0 + i, M - 1) * D + 4 * lane;
    a[i] = *reinterpret_cast<const float4*>(x1 + o);
    c[i] = *reinterpret_cast<const float4*>(x2 + o);
    if (dm) g[i] = *reinterpret_cast<const float4*>(dm + o);
  }
  if (dm) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a1 = wave_sum(dot4(g[i], a[i])), a2 = wave_sum(dot4(g[i], c[i]));
      if (lane == 0 && r0 + i < M) { dots[r0 + i] = a1; dots[(int64_t)M + r0 + i] = a2; }
    }
    return;
  }
  const float4 wp1 = *reinterpret_cast<const float4*>(p.wp[0] + 4 * lane), wp2 = *reinterpret_cast<const float4*>(p.wp[1] + 4 * lane);
  const float4 ww1 = *reinterpret_cast<const float4*>(p.ww[0] + 4 * lane), ww2 = *reinterpret_cast<const float4*>(p.ww[1] + 4 * lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float s1 = wave_sum(dot4(a[i], wp1)), s2 = wave_sum(dot4(c[i], wp2));
    const float q1 = wave_sum(dot4(a[i], ww1)), q2 = wave_sum(dot4(c[i], ww2));
    if (lane == 0 && r0 + i < M) {
      const int64_t r = r0 + i;
      dots[r] = s1; dots[M + r] = s2; dots[2 * (int64_t)M + r] = q1; dots[3 * (int64_t)M + r] = q2;
    }
  }
}

// grid (cdiv(T, MR_RPB), B), 512 threads, dynamic LDS 4 T floats
__global__ __launch_bounds__(512) void merge_rows_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                             const int64_t* __restrict__ lens, const int64_t* __restrict__ lens2,
                                                             MergeParams p, const float* __restrict__ dots,
                                                             float* __restrict__ score, float* __restrict__ wout,
                                                             float* __restrict__ mix, int B, int T) {
  constexpr int D = 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_s = sm;            // [2][T] scores, then the softmax over time
  float* s_q = sm + 2 * T;    // [2][T] <ww_k, x_k[t]>
  __shared__ float s_w[2];
  const int b = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t M = (int64_t)B * T, r0 = (int64_t)b * T;
  const float inv_sqrt_d = 1.f / 16.f;
  for (int i = threadIdx.x; i < 2 * T; i += 512) {
    const int k = i >= T, t = i - k * T;
    s_s[i] = (dots[k * M + r0 + t] + p.bp[k][0]) * inv_sqrt_d;
    s_q[i] = dots[(2 + k) * M + r0 + t];
  }
  __syncthreads();
  if (wv < 2) {
    const int k = wv;
    const int64_t* lk = (k == 1 && lens2) ? lens2 : lens;
    const int len = lk ? (int)min((int64_t)T, lk[b]) : T;
    float* sc = s_s + k * T;
    float mx = -FLT_MAX;
    for (int t = lane; t < len; t += 64) mx = fmaxf(mx, sc[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < len; t += 64) sum += expf(sc[t] - mx);
    sum = wave_sum(sum);
    const float inv = len > 0 ? 1.f / sum : 0.f;
    float lg = 0.f;
    for (int t = lane; t < T; t += 64) {      // (each lane rewrites only the entries it read)
      const float pv = t < len ? expf(sc[t] - mx) * inv : 0.f;
      sc[t] = pv;
      lg += pv * s_q[k * T + t];
    }
    lg = wave_sum(lg);
    if (lane == 0) s_w[k] = lg + p.bw[k][0];
  }
  __syncthreads();
  const float m = fmaxf(s_w[0], s_w[1]);
  const float e0 = expf(s_w[0] - m), e1 = expf(s_w[1] - m);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  if (ch == 0 && threadIdx.x == 0) { wout[b * 2 + 0] = w0; wout[b * 2 + 1] = w1; }
  if (threadIdx.x < 2 * MR_RPB) {
    const int k = threadIdx.x / MR_RPB, t = ch * MR_RPB + threadIdx.x % MR_RPB;
    if (t < T) score[((int64_t)k * B + b) * T + t] = s_s[k * T + t];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = ch * MR_RPB + wv * 2 + i;
    if (t < T) {
      const int64_t o = (r0 + t) * D + 4 * lane;
      const float4 a = *reinterpret_cast<const float4*>(x1 + o), c = *reinterpret_cast<const float4*>(x2 + o);
      *reinterpret_cast<float4*>(mix + o) = make_float4(w0 * a.x + w1 * c.x, w0 * a.y + w1 * c.y, w0 * a.z + w1 * c.z, w0 * a.w + w1 * c.w);
    }
  }
}

struct MergeDrop { uint32_t thr[2]; float inv_keep[2]; uint64_t offset4[2]; const uint64_t* seed; };

// grid (cdiv(T, MR_RPB), B), 512 threads, dynamic LDS 6 T floats; part[(b * chunks + ch)] = { dwp1[D], dwp2[D], dww1[D],
// dww2[D], dbp1, dbp2, dbw1, dbw2 } (the four scalars from chunk 0 only)
__global__ __launch_bounds__(512) void merge_rows_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ x1,
                                                             const float* __restrict__ x2, const int64_t* __restrict__ lens,
                                                             const int64_t* __restrict__ lens2, MergeParams p,
                                                             const float* __restrict__ score, const float* __restrict__ w,
                                                             const float* __restrict__ dots, const float* __restrict__ adots,
                                                             float* __restrict__ dx1, float* __restrict__ dx2,
                                                             float* __restrict__ part, MergeDrop dr, int B, int T) {
  constexpr int D = 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_a = sm;             // [2][T] <dm, x_k[t]>, then ds_pre_k[t]
  float* s_q = sm + 2 * T;     // [2][T] <ww_k, x_k[t]>
  float* s_c = sm + 4 * T;     // [2][T] softmax over time
  __shared__ float s_red[4];
  __shared__ __attribute__((aligned(16))) float s_part[8][4][D];
  const int b = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t M = (int64_t)B * T, r0 = (int64_t)b * T;
  for (int i = threadIdx.x; i < 2 * T; i += 512) {
    const int k = i >= T, t = i - k * T;
    s_a[i] = adots[k * M + r0 + t];
    s_q[i] = dots[(2 + k) * M + r0 + t];
    s_c[i] = score[((int64_t)k * B + b) * T + t];
  }
  __syncthreads();
  if (wv < 2) {
    float a = 0.f;
    for (int t = lane; t < T; t += 64) a += s_a[wv * T + t];
    a = wave_sum(a);
    if (lane == 0) s_red[wv] = a;
  }
  __syncthreads();
  const float w1 = w[b * 2], w2 = w[b * 2 + 1];
  const float dotw = w1 * s_red[0] + w2 * s_red[1];
  const float dwt[2] = {w1 * (s_red[0] - dotw), w2 * (s_red[1] - dotw)};       // d(weight_k), the branch softmax backward
  int len[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t* lk = (k == 1 && lens2) ? lens2 : lens;
    len[k] = lk ? (int)min((int64_t)T, lk[b]) : T;
  }
  __syncthreads();      // (s_red[0..1] are read by every thread above; the ds_pre writes below reuse s_a)
  if (wv < 2) {
    const int k = wv;
    const float dw = k == 0 ? dwt[0] : dwt[1];
    const int ln = k == 0 ? len[0] : len[1];
    float dot = 0.f;
    for (int t = lane; t < ln; t += 64) dot += s_c[k * T + t] * (dw * s_q[k * T + t]);
    dot = wave_sum(dot);
    float sb = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float v = t < ln ? s_c[k * T + t] * (dw * s_q[k * T + t] - dot) * (1.f / 16.f) : 0.f;
      s_a[k * T + t] = v;      // ds_pre_k[t]
      sb += v;
    }
    sb = wave_sum(sb);
    if (lane == 0) s_red[2 + k] = sb;
  }
  __syncthreads();
  float4 acc[4];      // dwp1, dwp2, pooled1, pooled2 partials over this wave's rows
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 wp1 = *reinterpret_cast<const float4*>(p.wp[0] + 4 * lane), wp2 = *reinterpret_cast<const float4*>(p.wp[1] + 4 * lane);
  const float4 ww1 = *reinterpret_cast<const float4*>(p.ww[0] + 4 * lane), ww2 = *reinterpret_cast<const float4*>(p.ww[1] + 4 * lane);
  const uint64_t sd = dr.seed ? dr.seed[0] : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = ch * MR_RPB + wv * 2 + i;
    if (t < T) {
      const int64_t row = r0 + t, o = row * D + 4 * lane;
      const float4 g = *reinterpret_cast<const float4*>(dm + o);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 xv = *reinterpret_cast<const float4*>((k == 0 ? x1 : x2) + o);
        const float4 wp = k == 0 ? wp1 : wp2, ww = k == 0 ? ww1 : ww2;
        const float wk = k == 0 ? w1 : w2;
        const float sv = s_c[k * T + t], dsv = s_a[k * T + t], sd_ = sv * dwt[k];
        float4 v = make_float4(wk * g.x + sd_ * ww.x + dsv * wp.x, wk * g.y + sd_ * ww.y + dsv * wp.y,
                               wk * g.z + sd_ * ww.z + dsv * wp.z, wk * g.w + sd_ * ww.w + dsv * wp.w);
        if (dr.seed && dr.thr[k]) {       // the branch's outer dropout (x_k = dropout(branch output)): tavsr_dropout's mapping
          const uint64_t ctr = dr.offset4[k] + (uint64_t)row * (D / 4) + (uint64_t)lane;
          uint32_t r4[4];
          philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), r4);
          const float ik = dr.inv_keep[k];
          v = make_float4(r4[0] >= dr.thr[k] ? v.x * ik : 0.f, r4[1] >= dr.thr[k] ? v.y * ik : 0.f,
                          r4[2] >= dr.thr[k] ? v.z * ik : 0.f, r4[3] >= dr.thr[k] ? v.w * ik : 0.f);
        }
        *reinterpret_cast<float4*>((k == 0 ? dx1 : dx2) + o) = v;
        acc[k].x += dsv * xv.x; acc[k].y += dsv * xv.y; acc[k].z += dsv * xv.z; acc[k].w += dsv * xv.w;
        acc[2 + k].x += sv * xv.x; acc[2 + k].y += sv * xv.y; acc[2 + k].z += sv * xv.z; acc[2 + k].w += sv * xv.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&s_part[wv][j][4 * lane]) = acc[j];
  __syncthreads();
  float* pb = part + ((int64_t)b * gridDim.x + ch) * (4 * D + 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = threadIdx.x + h * 512, j = idx >> 8, c = idx & 255;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += s_part[u][j][c];
    pb[j * D + c] = j < 2 ? a : a * dwt[j - 2];      // dwp_k ; dww_k = d(weight_k) * pooled_k
  }
  if (threadIdx.x < 4) {
    const int k = threadIdx.x & 1;
    pb[4 * D + threadIdx.x] = ch != 0 ? 0.f : (threadIdx.x < 2 ? s_red[2 + k] : dwt[k]);     // dbp_k, dbw_k
  }
}

// sum[i] = sum_r part[r][i] over nparts rows in a fixed order, written (or added) straight to the eight parameter gradients;
// one block per 64 slab columns, the four waves take rows wv, wv + 4, ...
__global__ __launch_bounds__(256) void merge_reduce_scatter_kernel(const float* __restrict__ part, int nparts, MergeGradPtrs o,
                                                                   int D, int accumulate) {
  __shared__ float s[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = 4 * D + 4, i = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < n) {
    int r = wv;
    for (; r + 12 < nparts; r += 16) {
      a0 += part[(int64_t)r * n + i];
      a1 += part[(int64_t)(r + 4) * n + i];
      a2 += part[(int64_t)(r + 8) * n + i];
      a3 += part[(int64_t)(r + 12) * n + i];
    }
    for (; r < nparts; r += 4) a0 += part[(int64_t)r * n + i];
  }
  s[wv][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wv == 0 && i < n) {
    const float v = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
    float* dst = i < 4 * D ? o.g[i / D] + (i % D) : o.g[4 + (i - 4 * D)];
    *dst = accumulate ? *dst + v : v;
  }
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_dwconv_gate_fwd(const float* gn, const float* r, int64_t ldr, const float* w, const float* bias,
                                     float* out, float* conv, int32_t B, int32_t T, int32_t C, int32_t K,
                                     tavsr_stream_t stream) {
  TAVSR_REQUIRE(gn && r && w && bias && out, TAVSR_EINVAL, "dwconv_gate_fwd: null pointer");
  TAVSR_REQUIRE(K >= 1 && K <= CG_KMAX && (K & 1), TAVSR_EUNSUPPORTED, "dwconv_gate_fwd: odd K <= %d required", CG_KMAX);
  if (B <= 0 || T <= 0 || C <= 0) return TAVSR_OK;
  size_t lds = ((CG_TT + K - 1) * CG_CH + K * CG_CH) * sizeof(float);
  if (K == 31)
    hipLaunchKernelGGL(dwconv_gate_fwd_kfix_kernel<31>, dim3(cdiv(C, CG_CH), cdiv(T, CG_TT), B), dim3(256),
                       (CG_TT + 30) * CG_CH * sizeof(float), (hipStream_t)stream, gn, r, ldr, w, bias, out, conv, B, T, C);
  else
    hipLaunchKernelGGL(dwconv_gate_fwd_kernel, dim3(cdiv(C, CG_CH), cdiv(T, CG_TT), B), dim3(256), lds,
                       (hipStream_t)stream, gn, r, ldr, w, bias, out, conv, B, T, C, K);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y,
                                   int64_t ldy, float* mean, float* rstd, int32_t M, int32_t D, tavsr_stream_t stream);

extern "C" int tavsr_csgu_fwd(const float* g, int64_t ldg, const float* ln_w, const float* ln_b, float eps, const float* conv_w,
                              const float* conv_b, float* out, float* gn, float* conv, float* mean, float* rstd, float p_drop,
                              const uint64_t* seed_dev, uint64_t offset, int32_t B, int32_t T, int32_t C, int32_t K,
                              const float* rowstat, tavsr_stream_t stream) {
  TAVSR_REQUIRE(g && ln_w && ln_b && conv_w && conv_b && out && mean && rstd, TAVSR_EINVAL, "csgu_fwd: null pointer");
  TAVSR_REQUIRE(K == 31 && C > 0 && C % CG_CH == 0, TAVSR_EUNSUPPORTED, "csgu_fwd: kernel size 31 and C %% 64 == 0 (got %d, %d)", K, C);
  TAVSR_REQUIRE(ldg % 4 == 0 && ldg >= 2 * (int64_t)C && ((uintptr_t)g % 16 == 0) && ((uintptr_t)ln_w % 16 == 0) &&
                    ((uintptr_t)ln_b % 16 == 0) && (!gn || (uintptr_t)gn % 16 == 0), TAVSR_EALIGN, "csgu_fwd: rows must be 16-byte aligned");
  TAVSR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_dev) && offset % 4 == 0, TAVSR_EINVAL,
                "csgu_fwd: dropout needs p in [0, 1), a device seed and an offset %% 4 == 0");
  if (B <= 0 || T <= 0) return TAVSR_OK;
  // statistics of the gate rows (columns C .. 2C-1 of g) - a launch of their own, or the producing GEMM's row statistics -
  // then the fused pass
  if (rowstat) {
    TAVSR_REQUIRE(C <= 1024 && ldg % 64 == 0 && ((uintptr_t)rowstat % 8 == 0), TAVSR_EUNSUPPORTED,
                  "csgu_fwd: GEMM row statistics need C <= 1024 and a row stride %% 64 == 0");
  } else {
    int rc = tavsr_layernorm_fwd(g + C, ldg, nullptr, nullptr, eps, nullptr, 0, mean, rstd, B * T, C, stream);
    if (rc) return rc;
  }
  const uint32_t thr = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  hipLaunchKernelGGL(csgu_fwd_kernel<31>, dim3(C / CG_CH, B), dim3(256), 0, (hipStream_t)stream, g, ldg, C, mean, rstd, ln_w, ln_b,
                     conv_w, conv_b, out, conv, gn, B, T, thr, p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed_dev, offset / 4,
                     rowstat, (int)(ldg / 64), eps, mean, rstd);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_dwconv_gate_bwd_ws(int32_t B, int32_t T, int32_t C, int32_t K) {
  (void)T;
  return ((int64_t)B + 1) * C * (K + 1);
}

static int dwconv_gate_bwd_impl(const float* du, const float* gn, const float* r, int64_t ldr, const float* conv, const float* w,
                               float* dr, int64_t lddr, float* dgn, float* dw, float* dbias, int32_t accumulate, float* ws, int32_t B,
                               int32_t T, int32_t C, int32_t K, const float* zr, int64_t ldz, int32_t act, tavsr_stream_t stream) {
  TAVSR_REQUIRE(du && gn && r && conv && w && dr && dgn && dw && dbias && ws, TAVSR_EINVAL,
                "dwconv_gate_bwd: null pointer");
  TAVSR_REQUIRE(K >= 1 && K <= CG_KMAX && (K & 1), TAVSR_EUNSUPPORTED, "dwconv_gate_bwd: odd K <= %d required", CG_KMAX);
  if (B <= 0 || T <= 0 || C <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  size_t lds = (2 * (CG_TB + K - 1) * CG_CH + K * CG_CH) * sizeof(float);
  if (K == 31)
    hipLaunchKernelGGL(dwconv_gate_bwd_kfix_kernel<31>, dim3(cdiv(C, CG_CH), B), dim3(256),
                       2 * (CG_TB + 30) * CG_CH * sizeof(float), s, du, gn, r, ldr, conv, w, dr, lddr, dgn, ws, B, T, C, zr, ldz, (int)act);
  else
    hipLaunchKernelGGL(dwconv_gate_bwd_kernel, dim3(cdiv(C, CG_CH), B), dim3(256), lds, s, du, gn, r, ldr, conv, w, dr,
                       lddr, dgn, ws, B, T, C, K);
  TAVSR_LAUNCH_CHECK();
  const int nblk = B, n = C * (K + 1);
  float* sum = ws + (int64_t)nblk * n;
  int rc = tavsr_sum_partials(ws, nblk, n, sum, n, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(dwconv_split_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, sum, dw, dbias, C, K, accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_dwconv_gate_bwd(const float* du, const float* gn, const float* r, int64_t ldr, const float* conv,
                                     const float* w, float* dr, int64_t lddr, float* dgn, float* dw, float* dbias,
                                     int32_t accumulate, float* ws, int32_t B, int32_t T, int32_t C, int32_t K,
                                     tavsr_stream_t stream) {
  return dwconv_gate_bwd_impl(du, gn, r, ldr, conv, w, dr, lddr, dgn, dw, dbias, accumulate, ws, B, T, C, K, nullptr, 0, 0, stream);
}

// ... with r = act(zr): dr comes back as the gradient w.r.t. the pre-activation zr (kernel size 31 only)
extern "C" int tavsr_dwconv_gate_bwd_act(const float* du, const float* gn, const float* r, int64_t ldr, const float* conv,
                                         const float* w, float* dr, int64_t lddr, float* dgn, float* dw, float* dbias,
                                         int32_t accumulate, float* ws, int32_t B, int32_t T, int32_t C, int32_t K,
                                         const float* zr, int64_t ldz, int32_t act, tavsr_stream_t stream) {
  TAVSR_REQUIRE(zr && K == 31, TAVSR_EUNSUPPORTED, "dwconv_gate_bwd_act: needs zr and kernel size 31 (K=%d)", K);
  return dwconv_gate_bwd_impl(du, gn, r, ldr, conv, w, dr, lddr, dgn, dw, dbias, accumulate, ws, B, T, C, K, zr, ldz, act, stream);
}


static MergeParams mk(const float* const* prm) {
  MergeParams p;
  for (int k = 0; k < 2; ++k) {
    p.wp[k] = prm[k]; p.bp[k] = prm[2 + k]; p.ww[k] = prm[4 + k]; p.bw[k] = prm[6 + k];
  }
  return p;
}

// params = { pooling_proj1.weight, pooling_proj2.weight, pooling_proj1.bias, pooling_proj2.bias,
//            weight_proj1.weight, weight_proj2.weight, weight_proj1.bias, weight_proj2.bias } (device pointers,
//            the array itself is HOST memory)
extern "C" int tavsr_merge_pool_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                                    const float* const* params,
                                    float* score, float* pooled, float* w, int32_t B, int32_t T, int32_t D,
                                    tavsr_stream_t stream) {
  TAVSR_REQUIRE(x1 && x2 && params && score && pooled && w, TAVSR_EINVAL, "merge_pool_fwd: null pointer");
  for (int i = 0; i < 8; ++i) TAVSR_REQUIRE(params[i], TAVSR_EINVAL, "merge_pool_fwd: null parameter %d", i);
  if (B <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(D % 4 == 0 && ((uintptr_t)x1 % 16 == 0) && ((uintptr_t)x2 % 16 == 0), TAVSR_EALIGN,
                "merge_pool_fwd: D %% 4 == 0 and 16-byte aligned rows required");
  size_t lds = (2 * T + 4 + 16 + 8 * D) * sizeof(float);
  TAVSR_REQUIRE(lds <= 60000, TAVSR_EUNSUPPORTED, "merge_pool_fwd: T=%d too long", T);
  hipLaunchKernelGGL(merge_pool_fwd_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, x1, x2, lens, lens2, mk(params),
                     score, pooled, w, B, T, D);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_merge_combine(const float* x1, const float* x2, const float* w, float* out, int32_t B, int32_t T,
                                   int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x1 && x2 && w && out, TAVSR_EINVAL, "merge_combine: null pointer");
  TAVSR_REQUIRE(((int64_t)T * D) % 4 == 0, TAVSR_EALIGN, "merge_combine: T*D must be a multiple of 4");
  int64_t total4 = (int64_t)B * T * D / 4;
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(merge_combine_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x1, x2, w, out,
                     total4, (int)((int64_t)T * D / 4));
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_merge_bwd_ws(int32_t B, int32_t D) { return (int64_t)(B + 1) * (4 * D + 4); }

// dparams = { d pooling_proj1.weight[D], d pooling_proj2.weight[D], d weight_proj1.weight[D], d weight_proj2.weight[D],
//             d pooling_proj1.bias[1], d pooling_proj2.bias[1], d weight_proj1.bias[1], d weight_proj2.bias[1] }
extern "C" int tavsr_merge_bwd(const float* dm, const float* x1, const float* x2, const int64_t* lens,
                               const int64_t* lens2, const float* const* params, const float* score, const float* pooled, const float* w,
                               float* dx1, float* dx2, float* const* dparams, int32_t accumulate, float* ws, int32_t B,
                               int32_t T, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dm && x1 && x2 && params && score && pooled && w && dx1 && dx2 && dparams && ws, TAVSR_EINVAL,
                "merge_bwd: null pointer");
  if (B <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(D % 4 == 0 && ((uintptr_t)x1 % 16 == 0) && ((uintptr_t)x2 % 16 == 0) && ((uintptr_t)dm % 16 == 0),
                TAVSR_EALIGN, "merge_bwd: D %% 4 == 0 and 16-byte aligned rows required");
  size_t lds = (2 * T + 4 + 2 * D + 32 + 8 * D) * sizeof(float);
  TAVSR_REQUIRE(lds <= 60000, TAVSR_EUNSUPPORTED, "merge_bwd: T=%d too long", T);
  hipLaunchKernelGGL(merge_bwd_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, dm, x1, x2, lens, lens2, mk(params), score,
                     pooled, w, dx1, dx2, ws, B, T, D);
  TAVSR_LAUNCH_CHECK();
  const int n = 4 * D + 4;
  float* sum = ws + (int64_t)B * n;
  int rc = tavsr_sum_partials(ws, B, n, sum, n, 0, stream);
  if (rc) return rc;
  MergeGradPtrs o;
  for (int i = 0; i < 8; ++i) {
    TAVSR_REQUIRE(dparams[i], TAVSR_EINVAL, "merge_bwd: null gradient pointer %d", i);
    o.g[i] = dparams[i];
  }
  hipLaunchKernelGGL(merge_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, sum, o, D, accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// ---- the row-parallel form (D = 256, T <= MR_TMAX): see the kernels' header comment
static constexpr int MR_TMAX = 2048;

extern "C" int tavsr_merge_rows_ok(int32_t T, int32_t D) { return D == 256 && T >= 1 && T <= MR_TMAX; }

extern "C" int tavsr_merge_rows_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                                    const float* const* params, float* dots, float* score, float* w, float* out, int32_t B,
                                    int32_t T, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x1 && x2 && params && dots && score && w && out, TAVSR_EINVAL, "merge_rows_fwd: null pointer");
  for (int i = 0; i < 8; ++i) TAVSR_REQUIRE(params[i], TAVSR_EINVAL, "merge_rows_fwd: null parameter %d", i);
  TAVSR_REQUIRE(tavsr_merge_rows_ok(T, D), TAVSR_EUNSUPPORTED, "merge_rows_fwd: D == 256 and T <= %d required (T=%d D=%d)", MR_TMAX, T, D);
  TAVSR_REQUIRE(((uintptr_t)x1 % 16 == 0) && ((uintptr_t)x2 % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                    ((uintptr_t)params[0] % 16 == 0) && ((uintptr_t)params[1] % 16 == 0) && ((uintptr_t)params[4] % 16 == 0) &&
                    ((uintptr_t)params[5] % 16 == 0),
                TAVSR_EALIGN, "merge_rows_fwd: 16-byte aligned rows and weight vectors required");
  if (B <= 0) return TAVSR_OK;
  const int M = B * T;
  hipLaunchKernelGGL(merge_rowdots_kernel, dim3(cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, x1, x2, (const float*)nullptr,
                     mk(params), dots, M);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge_rows_fwd_kernel, dim3(cdiv(T, MR_RPB), B), dim3(512), 4 * T * sizeof(float), (hipStream_t)stream, x1, x2,
                     lens, lens2, mk(params), dots, score, w, out, B, T);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_merge_rows_bwd_ws(int32_t B, int32_t T, int32_t D) {
  return 2 * (int64_t)B * T + (int64_t)B * cdiv(T, MR_RPB) * (4 * D + 4);
}

// p_drop1 / p_drop2 > 0: dx1 / dx2 are written under the tavsr_dropout mask of a contiguous [B*T][D] tensor at offset1 /
// offset2 (the branch outputs' outer dropouts, encoder_layer.py:212,224): dx_k = merge'(dm) * mask_k / keep_k
extern "C" int tavsr_merge_rows_bwd(const float* dm, const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                                    const float* const* params, const float* score, const float* w, const float* dots, float* dx1,
                                    float* dx2, float* const* dparams, int32_t accumulate, float* ws, float p_drop1,
                                    uint64_t offset1, float p_drop2, uint64_t offset2, const uint64_t* seed_dev, int32_t B,
                                    int32_t T, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dm && x1 && x2 && params && score && w && dots && dx1 && dx2 && dparams && ws, TAVSR_EINVAL,
                "merge_rows_bwd: null pointer");
  TAVSR_REQUIRE(tavsr_merge_rows_ok(T, D), TAVSR_EUNSUPPORTED, "merge_rows_bwd: D == 256 and T <= %d required (T=%d D=%d)", MR_TMAX, T, D);
  TAVSR_REQUIRE(((uintptr_t)x1 % 16 == 0) && ((uintptr_t)x2 % 16 == 0) && ((uintptr_t)dm % 16 == 0) && ((uintptr_t)dx1 % 16 == 0) &&
                    ((uintptr_t)dx2 % 16 == 0),
                TAVSR_EALIGN, "merge_rows_bwd: 16-byte aligned rows required");
  TAVSR_REQUIRE(p_drop1 >= 0.f && p_drop1 < 1.f && p_drop2 >= 0.f && p_drop2 < 1.f && offset1 % 4 == 0 && offset2 % 4 == 0 &&
                    (seed_dev || (p_drop1 == 0.f && p_drop2 == 0.f)),
                TAVSR_EINVAL, "merge_rows_bwd: dropout rates in [0, 1), offsets %% 4 == 0, a seed when a rate is set");
  MergeGradPtrs o;
  for (int i = 0; i < 8; ++i) {
    TAVSR_REQUIRE(dparams[i] && params[i], TAVSR_EINVAL, "merge_rows_bwd: null parameter or gradient pointer %d", i);
    o.g[i] = dparams[i];
  }
  if (B <= 0) return TAVSR_OK;
  const int M = B * T, chunks = cdiv(T, MR_RPB), n = 4 * D + 4;
  float* adots = ws;
  float* part = ws + 2 * (int64_t)M;
  MergeDrop dr;
  const float pd[2] = {p_drop1, p_drop2};
  const uint64_t off[2] = {offset1, offset2};
  for (int k = 0; k < 2; ++k) {
    dr.thr[k] = (uint32_t)((double)pd[k] * 4294967296.0);
    dr.inv_keep[k] = 1.f / (1.f - pd[k]);
    dr.offset4[k] = off[k] / 4;
  }
  dr.seed = (p_drop1 > 0.f || p_drop2 > 0.f) ? seed_dev : nullptr;
  hipLaunchKernelGGL(merge_rowdots_kernel, dim3(cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, x1, x2, dm, mk(params), adots, M);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge_rows_bwd_kernel, dim3(chunks, B), dim3(512), 6 * T * sizeof(float), (hipStream_t)stream, dm, x1, x2, lens,
                     lens2, mk(params), score, w, dots, adots, dx1, dx2, part, dr, B, T);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge_reduce_scatter_kernel, dim3(cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, part, B * chunks, o, D,
                     accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

