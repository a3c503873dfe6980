// Visual frontend pieces: Conv3d(1->64, k 5x7x7, s 1x2x2) + BatchNorm3d + Swish + MaxPool(1,3,3) and the
// ResNet-18 trunk (BasicBlock: conv3x3-BN-Swish-conv3x3-BN-(+1x1 conv/BN shortcut)-add-Swish, global average pool)
// of src/frontend/conv3d_resnet18/conv3d_resnet18.py:42-97 and modules/resnet.py:25-178.
//
// Layout: every activation is channels-last, one row per output pixel: [N*H*W, C] with N = B*T frames, so each
// convolution is an im2col gather (here) + one tavsr_gemm whose output IS the next layer's activation matrix, and
// BatchNorm statistics are column reductions of that matrix.  All kernels here are HBM-bound streaming kernels
// (16-byte accesses along C); reductions are two-stage with a fixed order (deterministic, no atomics).
#include <float.h>

#include <algorithm>

#include "common.h"

namespace tavsr {

// col[(n,ho,wo)][(kh*KW+kw)*C + c] = x[n, ho*s-p+kh, wo*s-p+kw, c]  (0 outside)      C % 4 == 0
__global__ void im2col2d_kernel(const float* __restrict__ x, float* __restrict__ col, int H, int W, int Ho, int Wo,
                                int C4, int KH, int KW, int stride, int pad, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  int64_t r = i / C4;
  const int k = (int)(r % (KH * KW));
  const int64_t m = r / (KH * KW);
  const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
  const int64_t n = m / ((int64_t)Wo * Ho);
  const int h = ho * stride - pad + k / KW, w = wo * stride - pad + k % KW;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (h >= 0 && h < H && w >= 0 && w < W) v = reinterpret_cast<const float4*>(x)[((n * H + h) * W + w) * C4 + c4];
  reinterpret_cast<float4*>(col)[i] = v;
}

// dx[n,h,w,c] = sum over the (kh,kw) whose output position (h+p-kh)/s, (w+p-kw)/s exists of dcol[...]   (gather form)
// extra (nullable, [n][Ho][Wo][C]): the data gradient of a parallel 1x1 / same-stride / pad-0 convolution of the same input
// (the downsample path of a ResNet block): added at the pixels (stride*ho, stride*wo) it touches, so that neither its
// mostly-zero scatter nor the sum of the two gradients needs a pass of its own
__global__ void col2im2d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int H, int W, int Ho, int Wo,
                                int C4, int KH, int KW, int stride, int pad, int64_t total4, const float* __restrict__ extra) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  int64_t r = i / C4;
  const int w = (int)(r % W), h = (int)((r / W) % H);
  const int64_t n = r / ((int64_t)W * H);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kh = 0; kh < KH; ++kh) {
    const int hn = h + pad - kh;
    if (hn < 0 || hn % stride) continue;
    const int ho = hn / stride;
    if (ho >= Ho) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int wn = w + pad - kw;
      if (wn < 0 || wn % stride) continue;
      const int wo = wn / stride;
      if (wo >= Wo) continue;
      const float4 v = reinterpret_cast<const float4*>(dcol)[(((n * Ho + ho) * Wo + wo) * (KH * KW) + kh * KW + kw) * C4 + c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (extra && h % stride == 0 && w % stride == 0 && h / stride < Ho && w / stride < Wo) {
    const float4 v = reinterpret_cast<const float4*>(extra)[((n * Ho + h / stride) * Wo + w / stride) * C4 + c4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  reinterpret_cast<float4*>(dx)[i] = acc;
}

// Stem im2col: x [B,T,H,W] (one channel) -> col [(b,t,ho,wo)][KP] with k = (kt*7 + kh)*7 + kw < 245 and zeros up to
// KP = 256; kernel (5,7,7), stride (1,2,2), padding (2,3,3) (conv3d_resnet18.py:48-56).
// One block per output row (b, t, ho): the 5 x 7 input rows it touches are staged in LDS once (coalesced 16-byte reads,
// zero columns / rows for the padding) and every patch row leaves as 64 consecutive float4 - a gather straight from
// global memory costs ~40 cache lines per load instruction (measured 2.3 TB/s of patch-matrix writes).
constexpr int kStemLd = 104;   // LDS row: 3 zero columns, W <= 96 pixels, zeros up to 2*(Wo-1)+7
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, float* __restrict__ col, int T, int H, int W,
                                                          int Ho, int Wo) {
  __shared__ float rows[35][kStemLd];
  const int ho = blockIdx.x % Ho, bt = blockIdx.x / Ho;
  const int t = bt % T;
  const float* xb = x + (int64_t)(bt - t) * H * W;            // frame 0 of utterance b
  for (int e = threadIdx.x; e < 35 * kStemLd; e += 256) {
    const int r = e / kStemLd, cix = e % kStemLd;
    const int kt = r / 7, kh = r % 7;
    const int tt = t - 2 + kt, h = ho * 2 - 3 + kh, w = cix - 3;
    float v = 0.f;
    if ((unsigned)tt < (unsigned)T && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = xb[((int64_t)tt * H + h) * W + w];
    rows[r][cix] = v;
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(col) + (int64_t)blockIdx.x * Wo * 64;
  for (int e = threadIdx.x; e < Wo * 64; e += 256) {
    const int wo = e >> 6, k4 = e & 63;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k4 * 4 + j;
      v[j] = k < 245 ? rows[k / 7][2 * wo + k % 7] : 0.f;      // k / 7 = kt*7 + kh
    }
    out[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- BatchNorm (training mode: batch statistics over the M rows of [M, C]) ----------------------------------
// Single pass: per-block partial sums of (x - s) and (x - s)^2 with the shift s[c] = x[0][c] (any sample is within a few
// sigma of the mean, so var = E[(x-s)^2] - E[x-s]^2 loses no digits to cancellation): part[blk][2][C].  One read of x
// instead of the two of the mean-then-variance form.
__global__ __launch_bounds__(256) void bn_partial2_kernel(const float* __restrict__ x, int64_t M, int C,
                                                          int64_t rows_per_block, float* __restrict__ part) {
  __shared__ float4 red[16][2][16];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (c < C) {
    const float4 sh = *reinterpret_cast<const float4*>(x + c);
    for (int64_t r = r0 + ry; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
      const float a = v.x - sh.x, b = v.y - sh.y, cc = v.z - sh.z, d = v.w - sh.w;
      s1.x += a; s1.y += b; s1.z += cc; s1.w += d;
      s2.x += a * a; s2.y += b * b; s2.z += cc * cc; s2.w += d * d;
    }
  }
  red[ry][0][cq] = s1;
  red[ry][1][cq] = s2;
  __syncthreads();
  if (ry < 2 && c < C) {
    float4 t = red[0][ry][cq];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = red[k][ry][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2 + ry) * C + c) = t;
  }
}

// mean, biased variance, rstd and the running statistics from the single-pass partials (combined in double)
__global__ __launch_bounds__(1024) void bn_finalize2_kernel(const float* __restrict__ part, int nparts, int C, int64_t M,
                                                            const float* __restrict__ x, float eps, float momentum,
                                                            float* __restrict__ mean, float* __restrict__ var,
                                                            float* __restrict__ rstd, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, int64_t* __restrict__ nbt) {
  __shared__ double red[16][2][64];
  const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int p = g;
    for (; p + 48 < nparts; p += 64) {      // four loads of each stream in flight; the sums stay in partial order
      const float a0 = part[((int64_t)p * 2 + 0) * C + c], a1 = part[((int64_t)(p + 16) * 2 + 0) * C + c];
      const float a2 = part[((int64_t)(p + 32) * 2 + 0) * C + c], a3 = part[((int64_t)(p + 48) * 2 + 0) * C + c];
      const float b0 = part[((int64_t)p * 2 + 1) * C + c], b1 = part[((int64_t)(p + 16) * 2 + 1) * C + c];
      const float b2 = part[((int64_t)(p + 32) * 2 + 1) * C + c], b3 = part[((int64_t)(p + 48) * 2 + 1) * C + c];
      a += (double)a0; a += (double)a1; a += (double)a2; a += (double)a3;
      b += (double)b0; b += (double)b1; b += (double)b2; b += (double)b3;
    }
    for (; p < nparts; p += 16) {
      a += (double)part[((int64_t)p * 2 + 0) * C + c];
      b += (double)part[((int64_t)p * 2 + 1) * C + c];
    }
  }
  red[g][0][cx] = a;
  red[g][1][cx] = b;
  __syncthreads();
  if (g != 0 || c >= C) return;
  a = b = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { a += red[q][0][cx]; b += red[q][1][cx]; }
  const double d = a / (double)M;                          // E[x - s]
  const double v = fmax(b / (double)M - d * d, 0.0);       // biased variance
  const float mu = (float)((double)x[c] + d);
  mean[c] = mu;
  var[c] = (float)v;
  rstd[c] = 1.f / sqrtf((float)v + eps);
  if (running_mean) {
    const float unb = M > 1 ? (float)(v * (double)M / (double)(M - 1)) : (float)v;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
  if (nbt && c == 0) nbt[0] += 1;
}

// y = act( (x - mean) * rstd * gamma + beta (+ res) ),  act = swish or identity;  float4 over C
__global__ void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ res,
                                    float* __restrict__ y, int C4, int act, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  const float4 mu = reinterpret_cast<const float4*>(mean)[c4], rs = reinterpret_cast<const float4*>(rstd)[c4];
  const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
  float4 z = make_float4((xv.x - mu.x) * rs.x * g.x + b.x, (xv.y - mu.y) * rs.y * g.y + b.y,
                         (xv.z - mu.z) * rs.z * g.z + b.z, (xv.w - mu.w) * rs.w * g.w + b.w);
  if (res) {
    const float4 r = reinterpret_cast<const float4*>(res)[i];
    z.x += r.x; z.y += r.y; z.z += r.z; z.w += r.w;
  }
  reinterpret_cast<float4*>(y)[i] = make_float4(act_fwd(act, z.x), act_fwd(act, z.y), act_fwd(act, z.z), act_fwd(act, z.w));
}

// backward, pass 1: dz = dy * act'(z) with z recomputed from x (and res); writes dz (also the gradient of `res`) and
// per-block partial column sums of dz and dz * xhat: part[blk][2][C]
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ res, float* __restrict__ dz,
                                                            int64_t M, int C, int64_t rows_per_block, int act,
                                                            float* __restrict__ part) {
  __shared__ float4 red[16][2][16];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;      // 4 channels x one of 16 row lanes (16-byte accesses)
  const int c = blockIdx.x * 64 + cq * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    auto one = [&](float xv, float rv, float dv, float m_, float r_, float g_, float b_, float& osb, float& osg) {
      const float xh = (xv - m_) * r_;
      const float d = dv * act_bwd(act, xh * g_ + b_ + rv);
      osb += d;
      osg += d * xh;
      return d;
    };
    for (int64_t r = r0 + ry; r < r1; r += 16) {
      const int64_t o = r * C + c;
      const float4 xv = *reinterpret_cast<const float4*>(x + o), dv = *reinterpret_cast<const float4*>(dy + o);
      const float4 rv = res ? *reinterpret_cast<const float4*>(res + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 d;
      d.x = one(xv.x, rv.x, dv.x, mu.x, rs.x, g.x, b.x, sb.x, sg.x);
      d.y = one(xv.y, rv.y, dv.y, mu.y, rs.y, g.y, b.y, sb.y, sg.y);
      d.z = one(xv.z, rv.z, dv.z, mu.z, rs.z, g.z, b.z, sb.z, sg.z);
      d.w = one(xv.w, rv.w, dv.w, mu.w, rs.w, g.w, b.w, sb.w, sg.w);
      if (dz) *reinterpret_cast<float4*>(dz + o) = d;      // dz == nullptr: recomputed by the apply pass (no skip path wants it)
    }
  }
  red[ry][0][cq] = sb;
  red[ry][1][cq] = sg;
  __syncthreads();
  if (ry < 2 && c < C) {        // row lane 0 sums dbeta, row lane 1 dgamma (fixed order)
    float4 t = red[0][ry][cq];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = red[k][ry][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2 + ry) * C + c) = t;
  }
}

// backward, pass 2: dx = gamma * rstd * (dz - dbeta/M - xhat * dgamma/M)
// dy != nullptr: dz was not stored by pass 1 (no residual input): it is recomputed here as dy * act'(z), z from x
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, float* __restrict__ dx, int C4, float invM,
                                    int64_t total4, const float* __restrict__ dy, const float* __restrict__ beta, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  const float4 mu = reinterpret_cast<const float4*>(mean)[c4], rs = reinterpret_cast<const float4*>(rstd)[c4];
  const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
  const float4 dg = reinterpret_cast<const float4*>(dgamma)[c4], db = reinterpret_cast<const float4*>(dbeta)[c4];
  float4 d;
  if (dy) {
    const float4 dv = reinterpret_cast<const float4*>(dy)[i], b = reinterpret_cast<const float4*>(beta)[c4];
    d.x = dv.x * act_bwd(act, (xv.x - mu.x) * rs.x * g.x + b.x);
    d.y = dv.y * act_bwd(act, (xv.y - mu.y) * rs.y * g.y + b.y);
    d.z = dv.z * act_bwd(act, (xv.z - mu.z) * rs.z * g.z + b.z);
    d.w = dv.w * act_bwd(act, (xv.w - mu.w) * rs.w * g.w + b.w);
  } else {
    d = reinterpret_cast<const float4*>(dz)[i];
  }
  float4 o;
  o.x = g.x * rs.x * (d.x - db.x * invM - (xv.x - mu.x) * rs.x * dg.x * invM);
  o.y = g.y * rs.y * (d.y - db.y * invM - (xv.y - mu.y) * rs.y * dg.y * invM);
  o.z = g.z * rs.z * (d.z - db.z * invM - (xv.z - mu.z) * rs.z * dg.z * invM);
  o.w = g.w * rs.w * (d.w - db.w * invM - (xv.w - mu.w) * rs.w * dg.w * invM);
  reinterpret_cast<float4*>(dx)[i] = o;
}

// ---- pooling ------------------------------------------------------------------------------------------------
// MaxPool 3x3 stride 2 pad 1 per frame, NHWC (the (1,3,3)/(1,2,2) MaxPool3d of the stem); idx = winning tap 0..8
// thread = 4 channels of one output pixel (16-byte loads; C % 4 == 0)
// mean != nullptr: the pooled map is act(BatchNorm(x)) - the normalisation and activation of bn_apply_fwd_kernel applied
// to every window element on the fly (same arithmetic), so the [N*H*W, C] activation map is never written (stem: 1.6 GB).
__global__ void maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                        int H, int W, int Ho, int Wo, int C, int64_t total4, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2;
  const int c4 = (int)(i % C4);
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu, g = mu, b = mu;
  if (mean) {
    mu = reinterpret_cast<const float4*>(mean)[c4]; rs = reinterpret_cast<const float4*>(rstd)[c4];
    g = reinterpret_cast<const float4*>(gamma)[c4]; b = reinterpret_cast<const float4*>(beta)[c4];
  }
  int64_t r = i / C4;
  const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
  const int64_t n = r / ((int64_t)Wo * Ho);
  float best[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  uint32_t bi[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int h = ho * 2 - 1 + k / 3, w = wo * 2 - 1 + k % 3;
    if (h < 0 || h >= H || w < 0 || w >= W) continue;
    float4 v4 = *reinterpret_cast<const float4*>(x + ((n * H + h) * W + w) * C + c4 * 4);
    if (mean)
      v4 = make_float4(act_fwd(act, (v4.x - mu.x) * rs.x * g.x + b.x), act_fwd(act, (v4.y - mu.y) * rs.y * g.y + b.y),
                       act_fwd(act, (v4.z - mu.z) * rs.z * g.z + b.z), act_fwd(act, (v4.w - mu.w) * rs.w * g.w + b.w));
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = k; }   // first maximum wins (torch scan order); NaN propagates
  }
  reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
  reinterpret_cast<uint32_t*>(idx)[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

// Gradient of MaxPool 3x3/s2/p1 for the 2x2 block of input pixels (2a..2a+1, 2b..2b+1), channels 4*c4..: the block lies
// under the 4 windows (a..a+1, b..b+1), each read once (idx + dy) for its 9 window references - 2 loads per output
// instead of 4.5; deterministic (gather form).
__device__ __forceinline__ void maxpool_block_grad(const float* __restrict__ dy, const uint8_t* __restrict__ idx, int64_t n, int a,
                                                   int b, int c4, int Ho, int Wo, int C4, float (&acc)[2][2][4]) {
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ph][pw][j] = 0.f;
#pragma unroll
  for (int dh = 0; dh < 2; ++dh) {
    const int ho = a + dh;
    if (ho >= Ho) continue;
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      const int wo = b + dw;
      if (wo >= Wo) continue;
      const int64_t o4 = ((n * Ho + ho) * Wo + wo) * C4 + c4;
      const uint32_t sel = reinterpret_cast<const uint32_t*>(idx)[o4];
      const float4 d4 = reinterpret_cast<const float4*>(dy)[o4];
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
      // window (ho, wo) covers rows 2ho-1..2ho+1: of this block's rows 2a, 2a+1 that is kh = 1, 2 (dh = 0) or kh = 0 at row 2a+1 (dh = 1)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int kh = 2 * a + ph - 2 * ho + 1;
        if (kh < 0 || kh > 2) continue;
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          const int kw = 2 * b + pw - 2 * wo + 1;
          if (kw < 0 || kw > 2) continue;
          const uint32_t k = kh * 3 + kw;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (((sel >> (8 * j)) & 0xFFu) == k) acc[ph][pw][j] += d[j];
        }
      }
    }
  }
}

// dx[n,h,w,c] = sum of dy over the (at most 4) windows that selected this pixel; thread = 4 channels of a 2x2 pixel block
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                        float* __restrict__ dx, int H, int W, int Ho, int Wo, int C, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2, H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const int c4 = (int)(i % C4);
  int64_t r = i / C4;
  const int b = (int)(r % W2), a = (int)((r / W2) % H2);
  const int64_t n = r / ((int64_t)W2 * H2);
  float acc[2][2][4];
  maxpool_block_grad(dy, idx, n, a, b, c4, Ho, Wo, C4, acc);
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw) {
      const int h = 2 * a + ph, w = 2 * b + pw;
      if (h < H && w < W)
        reinterpret_cast<float4*>(dx)[((n * H + h) * W + w) * C4 + c4] =
            make_float4(acc[ph][pw][0], acc[ph][pw][1], acc[ph][pw][2], acc[ph][pw][3]);
    }
}

// ---- BatchNorm backward whose incoming gradient is the POOLED one (the stem: conv -> BN -> act -> max-pool): the
// [N*H*W, C] gradient of the pool's input is never materialised - both passes rebuild it per 2x2 pixel block from the pooled
// gradient and the winning taps (0.5 GB instead of three 1.6 GB round trips at the benchmark size).
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                 const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int64_t nblocks, int H, int W,
                                                                 int Ho, int Wo, int C, int64_t blocks_per_wg, int act,
                                                                 float* __restrict__ part) {
  __shared__ float4 red[16][2][16];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const int C4 = C >> 2, H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const int64_t r0 = (int64_t)blockIdx.y * blocks_per_wg, r1 = min(nblocks, r0 + blocks_per_wg);
  float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + c), rs4 = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
    for (int64_t r = r0 + ry; r < r1; r += 16) {
      const int b = (int)(r % W2), a = (int)((r / W2) % H2);
      const int64_t n = r / ((int64_t)W2 * H2);
      float acc[2][2][4];
      maxpool_block_grad(dpool, idx, n, a, b, c >> 2, Ho, Wo, C4, acc);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          const int h = 2 * a + ph, w = 2 * b + pw;
          if (h >= H || w >= W) continue;
          const float4 xv4 = *reinterpret_cast<const float4*>(x + ((n * H + h) * W + w) * C + c);
          const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float d = acc[ph][pw][j] * act_bwd(act, xh * g[j] + bt[j]);
            sb[j] += d;
            sg[j] += d * xh;
          }
        }
    }
  }
  red[ry][0][cq] = make_float4(sb[0], sb[1], sb[2], sb[3]);
  red[ry][1][cq] = make_float4(sg[0], sg[1], sg[2], sg[3]);
  __syncthreads();
  if (ry < 2 && c < C) {        // row lane 0 sums dbeta, row lane 1 dgamma (fixed order)
    float4 t = red[0][ry][cq];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = red[k][ry][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2 + ry) * C + c) = t;
  }
}

__global__ void bn_pool_bwd_apply_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                         const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dx,
                                         int H, int W, int Ho, int Wo, int C, int act, float invM, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2, H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const int c4 = (int)(i % C4);
  int64_t r = i / C4;
  const int b = (int)(r % W2), a = (int)((r / W2) % H2);
  const int64_t n = r / ((int64_t)W2 * H2);
  float acc[2][2][4];
  maxpool_block_grad(dpool, idx, n, a, b, c4, Ho, Wo, C4, acc);
  const float4 mu4 = reinterpret_cast<const float4*>(mean)[c4], rs4 = reinterpret_cast<const float4*>(rstd)[c4];
  const float4 g4 = reinterpret_cast<const float4*>(gamma)[c4], b4 = reinterpret_cast<const float4*>(beta)[c4];
  const float4 dg4 = reinterpret_cast<const float4*>(dgamma)[c4], db4 = reinterpret_cast<const float4*>(dbeta)[c4];
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
  const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
  const float dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w}, db[4] = {db4.x, db4.y, db4.z, db4.w};
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw) {
      const int h = 2 * a + ph, w = 2 * b + pw;
      if (h >= H || w >= W) continue;
      const int64_t o = ((n * H + h) * W + w) * C4 + c4;
      const float4 xv4 = reinterpret_cast<const float4*>(x)[o];
      const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = acc[ph][pw][j] * act_bwd(act, (xv[j] - mu[j]) * rs[j] * g[j] + bt[j]);
        out[j] = g[j] * rs[j] * (d - db[j] * invM - (xv[j] - mu[j]) * rs[j] * dg[j] * invM);
      }
      reinterpret_cast<float4*>(dx)[o] = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// global average pool over the P pixels of each frame: y[n,c] = mean_p x[n,p,c]; bwd: dx[n,p,c] = dy[n,c] / P
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t n = i / C;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += x[(n * P + p) * C + c];
  y[i] = s / (float)P;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int P, int C, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t n = i / ((int64_t)C * P);
  dx[i] = dy[n * C + c] / (float)P;
}
// C % 4 == 0: one thread per (frame, 4 channels) writes its P rows (16-byte stores, no 64-bit divisions per element)
__global__ void avgpool_bwd_vec_kernel(const float* __restrict__ dy, float* __restrict__ dx, int P, int C4, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  const int64_t n = i / C4;
  float4 v = reinterpret_cast<const float4*>(dy)[i];
  v = make_float4(v.x / (float)P, v.y / (float)P, v.z / (float)P, v.w / (float)P);      // the division of the scalar kernel
  float4* o = reinterpret_cast<float4*>(dx) + n * P * C4 + c4;
  for (int p = 0; p < P; ++p) o[(int64_t)p * C4] = v;
}

__global__ void rsqrt_eps_kernel(const float* __restrict__ v, float eps, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.f / sqrtf(v[i] + eps);
}

// dst[m*ldd + n] = src[m*lds + n]: strided 2-D copy without alignment requirements (stem weight 245 <-> 256 columns,
// alignment padding of [B, T*D] rows)
__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int64_t N,
                              int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t m = i / N, n = i % N;
  dst[m * ldd + n] = src[m * lds + n];
}

// weights of the data-gradient convolution: wflip[ci][tap'*Cout + co] = w2d[co][(8 - tap')*Cin + ci]
// (w2d = [Cout][9*Cin] tap-major GEMM weight of a 3x3 convolution; the transposed convolution visits the taps mirrored)
__global__ void conv_wflip_kernel(const float* __restrict__ w2d, float* __restrict__ wflip, int Cout, int Cin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cin * 9;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  const int tap = (int)((i / Cout) % 9);
  const int ci = (int)(i / ((int64_t)Cout * 9));
  wflip[i] = w2d[(int64_t)co * 9 * Cin + (int64_t)(8 - tap) * Cin + ci];
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

static inline int bn_chunks(int64_t M) { return (int)std::min<int64_t>(std::max<int64_t>(1, M / 256), 1024); }

}  // namespace tavsr

using namespace tavsr;

static inline dim3 grid1d(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

extern "C" int tavsr_im2col2d(const float* x, float* col, int64_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                              int32_t stride, int32_t pad, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && col, TAVSR_EINVAL, "im2col2d: null pointer");
  TAVSR_REQUIRE(C % 4 == 0 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, TAVSR_EUNSUPPORTED, "im2col2d: C %% 4 == 0 required");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const int64_t total4 = N * Ho * Wo * KH * KW * (C / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(im2col2d_kernel, grid1d(total4), dim3(256), 0, (hipStream_t)stream, x, col, H, W, Ho, Wo, C / 4, KH, KW,
                     stride, pad, total4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_col2im2d(const float* dcol, float* dx, int64_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                              int32_t stride, int32_t pad, const float* extra, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dcol && dx, TAVSR_EINVAL, "col2im2d: null pointer");
  TAVSR_REQUIRE(C % 4 == 0 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, TAVSR_EUNSUPPORTED, "col2im2d: C %% 4 == 0 required");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const int64_t total4 = N * H * W * (C / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(col2im2d_kernel, grid1d(total4), dim3(256), 0, (hipStream_t)stream, dcol, dx, H, W, Ho, Wo, C / 4, KH, KW,
                     stride, pad, total4, extra);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_im2col_stem(const float* x, float* col, int32_t B, int32_t T, int32_t H, int32_t W, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && col, TAVSR_EINVAL, "im2col_stem: null pointer");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  if ((int64_t)B * T * Ho * Wo <= 0) return TAVSR_OK;
  TAVSR_REQUIRE((int64_t)B * T * Ho < (1ll << 31) && W + 3 <= kStemLd && 2 * (Wo - 1) + 7 <= kStemLd, TAVSR_EUNSUPPORTED,
                "im2col_stem: frames up to %d pixels wide, fewer than 2^31 output rows (decode/eval goes through in slices)",
                kStemLd - 3);
  hipLaunchKernelGGL(im2col_stem_kernel, dim3((unsigned)((int64_t)B * T * Ho)), dim3(256), 0, (hipStream_t)stream, x, col, T, H, W,
                     Ho, Wo);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_bn_ws(int64_t M, int32_t C) { return (int64_t)bn_chunks(M) * 2 * C; }

extern "C" int tavsr_bn_stats(const float* x, int64_t M, int32_t C, float eps, float momentum, float* mean, float* var,
                              float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ws,
                              tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && mean && var && rstd && ws, TAVSR_EINVAL, "bn_stats: null pointer");
  TAVSR_REQUIRE(M > 0 && C > 0, TAVSR_EINVAL, "bn_stats: empty input");
  TAVSR_REQUIRE(C % 4 == 0 && ((uintptr_t)x & 15) == 0, TAVSR_EUNSUPPORTED, "bn_stats: C %% 4 == 0 and 16-byte aligned rows required");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = bn_chunks(M);
  const int64_t rpb = (M + chunks - 1) / chunks;
  const dim3 g(cdiv(C, 64), chunks);
  hipLaunchKernelGGL(bn_partial2_kernel, g, dim3(256), 0, s, x, M, C, rpb, ws);
  hipLaunchKernelGGL(bn_finalize2_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, ws, chunks, C, M, x, eps, momentum, mean, var, rstd,
                     running_mean, running_var, num_batches_tracked);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_bn_apply_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* res, float* y, int64_t M, int32_t C, int32_t act, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && mean && rstd && gamma && beta && y, TAVSR_EINVAL, "bn_apply_fwd: null pointer");
  TAVSR_REQUIRE(C % 4 == 0, TAVSR_EUNSUPPORTED, "bn_apply_fwd: C %% 4 == 0 required");
  const int64_t total4 = M * (C / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(bn_apply_fwd_kernel, grid1d(total4), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, res, y,
                     C / 4, act, total4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// dz (the gradient w.r.t. the pre-activation = the gradient of `res`), dx, dgamma, dbeta from dy; ws >= tavsr_bn_ws floats.
// dz == nullptr (allowed without res): the [M, C] intermediate is never written - pass 2 recomputes it from dy and x.
extern "C" int tavsr_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const float* res, float* dz, float* dx, float* dgamma, float* dbeta, int64_t M,
                            int32_t C, int32_t act, float* ws, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dy && x && mean && rstd && gamma && beta && dx && dgamma && dbeta && ws, TAVSR_EINVAL, "bn_bwd: null pointer");
  TAVSR_REQUIRE(dz || !res, TAVSR_EINVAL, "bn_bwd: dz (the gradient of res) is needed when there is a residual input");
  TAVSR_REQUIRE(C % 4 == 0, TAVSR_EUNSUPPORTED, "bn_bwd: C %% 4 == 0 required");
  if (M <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = bn_chunks(M);
  const int64_t rpb = (M + chunks - 1) / chunks;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C, 64), chunks), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta, res, dz, M,
                     C, rpb, act, ws);
  TAVSR_LAUNCH_CHECK();
  int rc = tavsr_sum_partials2(ws, chunks, (int64_t)2 * C, dbeta, C, dgamma, C, 0, stream);      // (dbeta | dgamma) slab, one launch
  if (rc) return rc;
  const int64_t total4 = M * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid1d(total4), dim3(256), 0, s, dz, x, mean, rstd, gamma, dgamma, dbeta, dx, C / 4,
                     1.f / (float)M, total4, dz ? nullptr : dy, beta, act);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// tavsr_bn_bwd for a BatchNorm whose activated output went through MaxPool 3x3/s2/p1: dpool [N,Ho,Wo,C] + idx are the pool's
// gradient and winning taps; dx, dgamma, dbeta as tavsr_bn_bwd (no residual input)
extern "C" int tavsr_bn_bwd_pooled(const float* dpool, const uint8_t* idx, const float* x, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, int64_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t act, float* ws, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dpool && idx && x && mean && rstd && gamma && beta && dx && dgamma && dbeta && ws, TAVSR_EINVAL,
                "bn_bwd_pooled: null pointer");
  TAVSR_REQUIRE(C % 4 == 0 && ((uintptr_t)idx & 3) == 0 && (((uintptr_t)dpool | (uintptr_t)x | (uintptr_t)dx) & 15) == 0,
                TAVSR_EUNSUPPORTED, "bn_bwd_pooled: C %% 4 == 0 and aligned tensors required");
  if (N <= 0 || H <= 0 || W <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t M = N * H * W, nblocks = N * ((H + 1) / 2) * ((W + 1) / 2);
  const int chunks = bn_chunks(M);                               // ws sized by tavsr_bn_ws(M, C)
  const int64_t bpw = (nblocks + chunks - 1) / chunks;
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(cdiv(C, 64), chunks), dim3(256), 0, s, dpool, idx, x, mean, rstd, gamma, beta,
                     nblocks, H, W, Ho, Wo, C, bpw, act, ws);
  TAVSR_LAUNCH_CHECK();
  int rc = tavsr_sum_partials2(ws, chunks, (int64_t)2 * C, dbeta, C, dgamma, C, 0, stream);      // (dbeta | dgamma) slab, one launch
  if (rc) return rc;
  const int64_t total4 = nblocks * (C / 4);
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, grid1d(total4), dim3(256), 0, s, dpool, idx, x, mean, rstd, gamma, beta, dgamma, dbeta,
                     dx, H, W, Ho, Wo, C, act, 1.f / (float)M, total4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int32_t H, int32_t W, int32_t C,
                                      tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y && idx, TAVSR_EINVAL, "maxpool_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  TAVSR_REQUIRE(C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && ((uintptr_t)idx & 3) == 0, TAVSR_EUNSUPPORTED,
                "maxpool_fwd: C %% 4 == 0 and aligned tensors required");
  const int64_t total = N * Ho * Wo * (C / 4);
  if (total <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, idx, H, W, Ho, Wo, C, total,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_bn_act_maxpool3x3s2_fwd(const float* x, const float* mean, const float* rstd, const float* gamma,
                                             const float* beta, int32_t act, float* y, uint8_t* idx, int64_t N, int32_t H,
                                             int32_t W, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && mean && rstd && gamma && beta && y && idx, TAVSR_EINVAL, "bn_act_maxpool_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  TAVSR_REQUIRE(C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 &&
                    ((uintptr_t)idx & 3) == 0, TAVSR_EUNSUPPORTED, "bn_act_maxpool_fwd: C %% 4 == 0 and aligned tensors required");
  const int64_t total = N * Ho * Wo * (C / 4);
  if (total <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, idx, H, W, Ho, Wo, C, total,
                     mean, rstd, gamma, beta, act);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                                      tavsr_stream_t stream) {
  TAVSR_REQUIRE(dy && idx && dx, TAVSR_EINVAL, "maxpool_bwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  TAVSR_REQUIRE(C % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && ((uintptr_t)idx & 3) == 0, TAVSR_EUNSUPPORTED,
                "maxpool_bwd: C %% 4 == 0 and aligned tensors required");
  const int64_t total = N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  if (total <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, H, W, Ho, Wo, C, total);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_avgpool_fwd(const float* x, float* y, int64_t N, int32_t P, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y, TAVSR_EINVAL, "avgpool_fwd: null pointer");
  const int64_t total = N * C;
  if (total <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(avgpool_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, P, C, total);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_avgpool_bwd(const float* dy, float* dx, int64_t N, int32_t P, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dy && dx, TAVSR_EINVAL, "avgpool_bwd: null pointer");
  const int64_t total = N * P * C;
  if (total <= 0) return TAVSR_OK;
  if (C % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0)
    hipLaunchKernelGGL(avgpool_bwd_vec_kernel, grid1d(N * (C / 4)), dim3(256), 0, (hipStream_t)stream, dy, dx, P, C / 4, N * (C / 4));
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, dy, dx, P, C, total);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_rsqrt_eps(const float* v, float eps, float* out, int64_t n, tavsr_stream_t stream) {
  TAVSR_REQUIRE((v && out) || n <= 0, TAVSR_EINVAL, "rsqrt_eps: null pointer");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(rsqrt_eps_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, v, eps, out, n);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int64_t N, tavsr_stream_t stream) {
  TAVSR_REQUIRE((src && dst) || M * N <= 0, TAVSR_EINVAL, "copy2d: null pointer");
  if (M * N <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(copy2d_kernel, grid1d(M * N), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, N, M * N);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_conv_wflip(const float* w2d, float* wflip, int32_t Cout, int32_t Cin, tavsr_stream_t stream) {
  TAVSR_REQUIRE(w2d && wflip && Cout > 0 && Cin > 0, TAVSR_EINVAL, "conv_wflip: bad arguments");
  hipLaunchKernelGGL(conv_wflip_kernel, grid1d((int64_t)Cout * Cin * 9), dim3(256), 0, (hipStream_t)stream, w2d, wflip, Cout, Cin);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_fill(float* p, float value, int64_t n, tavsr_stream_t stream) {
  TAVSR_REQUIRE(p || n <= 0, TAVSR_EINVAL, "fill: null pointer");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(fill_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, p, value, n);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
