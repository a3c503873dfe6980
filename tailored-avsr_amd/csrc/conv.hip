// Conv2dSubsampling front-end pieces (espnet subsampling.py, built at
// src/encoder/branchformer/encoder.py:149-155, called at :364) and UtteranceMVN
// (espnet2 utterance_mvn.py, called at src/models/espnet_model.py:388).
// Activations are kept channels-last (NHWC) so that conv2 becomes an im2col GEMM whose output is
// already the (b, t, f, c) row the out-Linear consumes (its weight is re-indexed once per step by
// tavsr_transpose_inner).  All kernels here are HBM-bound streaming kernels with 16-B accesses.
#include "common.h"

namespace tavsr {

// y[b,to,fo,c] = relu(bias[c] + sum_{kh,kw} w[c,kh,kw] * x[b, 2to+kh, 2fo+kw]),  x: [B,T,F] (1 channel)
constexpr int kConv1Pos = 64;     // output positions per block

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int T, int F, int To, int Fo, int C) {
  // thread -> 4 channels (its 36 weights stay in registers) x one of 4 position lanes; a block walks 64 positions:
  // the kernel is bound by the 1 KB it writes per position, not by re-reading weights
  const int cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t npos = (int64_t)B * To * Fo;
  const int64_t p0 = (int64_t)blockIdx.x * kConv1Pos;
  for (int c = cq * 4; c < C; c += 256) {
    float wr[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 9; ++k) wr[j][k] = w[(int64_t)(c + j) * 9 + k];
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    for (int q = pl; q < kConv1Pos; q += 4) {
      const int64_t pos = p0 + q;
      if (pos >= npos) break;
      const int fo = (int)(pos % Fo);
      const int to = (int)((pos / Fo) % To);
      const int b = (int)(pos / ((int64_t)Fo * To));
      const float* xp = x + ((int64_t)b * T + 2 * to) * F + 2 * fo;
      float xv[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = xp[kh * F + kw];
      float o[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) o[j] += wr[j][k] * xv[k];
        o[j] = fmaxf(o[j], 0.f);
      }
      *reinterpret_cast<float4*>(y + pos * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// partial dW1[c][k], db1[c] over a chunk of positions: dz = dy (already masked by relu')
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                        float* __restrict__ part, int B, int T, int F, int To, int Fo,
                                                        int C, int pos_per_block) {
  __shared__ float s_x[64][9];
  const int64_t npos = (int64_t)B * To * Fo;
  const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
  const int64_t p1 = min(npos, p0 + pos_per_block);
  float acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.f;
  const int c = threadIdx.x;  // C <= 256 handled by one pass; larger C loops below
  for (int64_t pb = p0; pb < p1; pb += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 9; i += 256) {
      int pl = i / 9, k = i % 9;
      int64_t pos = pb + pl;
      float v = 0.f;
      if (pos < p1) {
        int fo = (int)(pos % Fo), to = (int)((pos / Fo) % To), b = (int)(pos / ((int64_t)Fo * To));
        v = x[((int64_t)b * T + 2 * to + k / 3) * F + 2 * fo + k % 3];
      }
      s_x[pl][k] = v;
    }
    __syncthreads();
    const int n = (int)min((int64_t)64, p1 - pb);
    if (c < C)
      for (int pl = 0; pl < n; ++pl) {
        float g = dz[(pb + pl) * C + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += g * s_x[pl][k];
        acc[9] += g;
      }
  }
  if (c < C) {
    float* pp = part + (int64_t)blockIdx.x * C * 10 + (int64_t)c * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) pp[k] = acc[k];
  }
}

__global__ void conv1_split_kernel(const float* __restrict__ sum, float* __restrict__ dw, float* __restrict__ db, int C,
                                   int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 10) return;
  int c = i / 10, k = i % 10;
  float v = sum[i];
  if (k < 9) dw[c * 9 + k] = accumulate ? dw[c * 9 + k] + v : v;
  else db[c] = accumulate ? db[c] + v : v;
}

// col[(b,to,fo)][(kh*3+kw)*C + c] = y[b, 2to+kh, 2fo+kw, c]    (3x3, stride 2, NHWC, C % 4 == 0)
__global__ void im2col3x3s2_kernel(const float* __restrict__ y, float* __restrict__ col, int B, int Ti, int Fi, int To,
                                   int Fo, int C4, int64_t total4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int c4 = (int)(i % C4);
  int64_t r = i / C4;
  int k = (int)(r % 9);
  int64_t m = r / 9;
  int fo = (int)(m % Fo), to = (int)((m / Fo) % To), b = (int)(m / ((int64_t)Fo * To));
  int kh = k / 3, kw = k % 3;
  const float4* src = reinterpret_cast<const float4*>(y) + (((int64_t)b * Ti + 2 * to + kh) * Fi + 2 * fo + kw) * C4 + c4;
  reinterpret_cast<float4*>(col)[i] = *src;
}

// dz[b,ti,fi,c] = (yrelu[b,ti,fi,c] > 0) * sum_{kh,kw valid} dcol[(b,(ti-kh)/2,(fi-kw)/2)][(kh*3+kw)*C + c]
__global__ void col2im3x3s2_relu_kernel(const float* __restrict__ dcol, const float* __restrict__ yrelu,
                                        float* __restrict__ dz, int B, int Ti, int Fi, int To, int Fo, int C4,
                                        int64_t total4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int c4 = (int)(i % C4);
  int64_t r = i / C4;
  int fi = (int)(r % Fi), ti = (int)((r / Fi) % Ti), b = (int)(r / ((int64_t)Fi * Ti));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    int tn = ti - kh;
    if (tn < 0 || (tn & 1) || (tn >> 1) >= To) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      int fn = fi - kw;
      if (fn < 0 || (fn & 1) || (fn >> 1) >= Fo) continue;
      int64_t m = ((int64_t)b * To + (tn >> 1)) * Fo + (fn >> 1);
      float4 v = reinterpret_cast<const float4*>(dcol)[(m * 9 + kh * 3 + kw) * C4 + c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  float4 yv = reinterpret_cast<const float4*>(yrelu)[i];
  acc.x = yv.x > 0.f ? acc.x : 0.f; acc.y = yv.y > 0.f ? acc.y : 0.f;
  acc.z = yv.z > 0.f ? acc.z : 0.f; acc.w = yv.w > 0.f ? acc.w : 0.f;
  reinterpret_cast<float4*>(dz)[i] = acc;
}

// out[n][c][r] = in[n][r][c]  (batched inner transpose; used to re-index conv / out-Linear weights
// between torch's (co, ci, kh, kw) / (c*F+f) orders and the channels-last orders, and back for grads)
__global__ void transpose_inner_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t nb, int R, int Cc,
                                       int accumulate) {
  __shared__ float tile[32][33];
  int64_t n = blockIdx.z;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const float* ip = in + n * (int64_t)R * Cc;
  float* op = out + n * (int64_t)R * Cc;
  for (int j = threadIdx.y; j < 32; j += 8) {
    int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < R && c < Cc) ? ip[(int64_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    int c = c0 + j, r = r0 + threadIdx.x;
    if (r < R && c < Cc) {
      float v = tile[threadIdx.x][j];
      int64_t o = (int64_t)c * R + r;
      op[o] = accumulate ? op[o] + v : v;
    }
  }
}

// y[b,t,:] = (t < len[b]) ? x[b,t,:] - mean_b : 0 ; mean over valid frames (norm_means only)
__global__ __launch_bounds__(256) void utt_mvn_kernel(const float* __restrict__ x, const int64_t* __restrict__ lens,
                                                      float* __restrict__ y, int T, int F) {
  __shared__ float s_mean[256];
  __shared__ float s_part[4][64];
  const int b = blockIdx.x;
  const int len = (int)min((int64_t)T, lens[b]);
  const float* xb = x + (int64_t)b * T * F;
  float* yb = y + (int64_t)b * T * F;
  const int fx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int f0 = 0; f0 < F; f0 += 64) {
    int f = f0 + fx;
    float s = 0.f;
    if (f < F)
      for (int t = ty; t < len; t += 4) s += xb[(int64_t)t * F + f];
    s_part[ty][fx] = s;
    __syncthreads();
    if (ty == 0 && f < F) s_mean[f] = ((s_part[0][fx] + s_part[1][fx]) + (s_part[2][fx] + s_part[3][fx])) / (float)len;
    __syncthreads();
  }
  for (int64_t i = threadIdx.x; i < (int64_t)T * F; i += 256) {
    int t = (int)(i / F), f = (int)(i % F);
    yb[i] = t < len ? xb[i] - s_mean[f] : 0.f;
  }
}

// The same for F % 4 == 0, 16-byte aligned rows: grid (UM_SLICES, B).  Every block of an utterance computes the column means
// (threads = row groups x 16-byte columns, a fixed-order LDS reduction over the row groups) and normalises its own slice of the
// rows.  (One block per utterance walking its columns 64 at a time took 80 us at B = 32, T = 400, F = 80 - the first kernel of
// every audio step.)
constexpr int UM_SLICES = 4;
__global__ __launch_bounds__(256) void utt_mvn_vec_kernel(const float* __restrict__ x, const int64_t* __restrict__ lens,
                                                          float* __restrict__ y, int T, int F) {
  __shared__ float4 s_part[256];
  __shared__ float4 s_mean[64];
  const int b = blockIdx.y, F4 = F >> 2;
  const int len = (int)min((int64_t)T, lens[b]);
  const float4* xb = reinterpret_cast<const float4*>(x + (int64_t)b * T * F);
  float4* yb = reinterpret_cast<float4*>(y + (int64_t)b * T * F);
  const int groups = 256 / F4, c4 = threadIdx.x % F4, rg = threadIdx.x / F4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rg < groups) {
    int t = rg;
    for (; t + 3 * groups < len; t += 4 * groups) {
      const float4 a = xb[(int64_t)t * F4 + c4], c = xb[(int64_t)(t + groups) * F4 + c4];
      const float4 d = xb[(int64_t)(t + 2 * groups) * F4 + c4], e = xb[(int64_t)(t + 3 * groups) * F4 + c4];
      acc.x += (a.x + c.x) + (d.x + e.x); acc.y += (a.y + c.y) + (d.y + e.y);
      acc.z += (a.z + c.z) + (d.z + e.z); acc.w += (a.w + c.w) + (d.w + e.w);
    }
    for (; t < len; t += groups) {
      const float4 a = xb[(int64_t)t * F4 + c4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    s_part[rg * F4 + c4] = acc;
  }
  __syncthreads();
  if (threadIdx.x < F4) {
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < groups; ++g) {
      const float4 a = s_part[g * F4 + threadIdx.x];
      m.x += a.x; m.y += a.y; m.z += a.z; m.w += a.w;
    }
    const float inv = 1.f / (float)len;
    s_mean[threadIdx.x] = make_float4(m.x * inv, m.y * inv, m.z * inv, m.w * inv);
  }
  __syncthreads();
  const int rows = (T + UM_SLICES - 1) / UM_SLICES, t0 = blockIdx.x * rows, t1 = min(T, t0 + rows);
  for (int i = t0 * F4 + threadIdx.x; i < t1 * F4; i += 256) {
    const int t = i / F4;
    const float4 m = s_mean[i - t * F4], v = xb[i];
    yb[i] = t < len ? make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_conv1_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t T,
                               int32_t F, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && w && bias && y, TAVSR_EINVAL, "conv1_fwd: null pointer");
  TAVSR_REQUIRE(T >= 3 && F >= 3 && C % 4 == 0, TAVSR_EINVAL, "conv1_fwd: need T,F >= 3 and C %% 4 == 0");
  const int To = (T - 3) / 2 + 1, Fo = (F - 3) / 2 + 1;
  int64_t npos = (int64_t)B * To * Fo;
  if (npos <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3(cdiv(npos, kConv1Pos)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, T, F,
                     To, Fo, C);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

static inline int conv1_blocks(int64_t npos) { return (int)min((int64_t)2048, (npos + 127) / 128); }

extern "C" int64_t tavsr_conv1_bwd_ws(int32_t B, int32_t T, int32_t F, int32_t C) {
  const int To = (T - 3) / 2 + 1, Fo = (F - 3) / 2 + 1;
  return ((int64_t)conv1_blocks((int64_t)B * To * Fo) + 1) * C * 10;
}

extern "C" int tavsr_conv1_bwd(const float* dz, const float* x, float* dw, float* db, int32_t accumulate, float* ws,
                               int32_t B, int32_t T, int32_t F, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dz && x && dw && db && ws, TAVSR_EINVAL, "conv1_bwd: null pointer");
  TAVSR_REQUIRE(C <= 256, TAVSR_EUNSUPPORTED, "conv1_bwd: C <= 256 supported");
  const int To = (T - 3) / 2 + 1, Fo = (F - 3) / 2 + 1;
  int64_t npos = (int64_t)B * To * Fo;
  if (npos <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nb = conv1_blocks(npos);
  const int ppb = (int)((npos + nb - 1) / nb);
  hipLaunchKernelGGL(conv1_bwd_kernel, dim3(nb), dim3(256), 0, s, dz, x, ws, B, T, F, To, Fo, C, ppb);
  TAVSR_LAUNCH_CHECK();
  float* sum = ws + (int64_t)nb * C * 10;
  int rc = tavsr_sum_partials(ws, nb, (int64_t)C * 10, sum, C * 10, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(conv1_split_kernel, dim3(cdiv(C * 10, 256)), dim3(256), 0, s, sum, dw, db, C, accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_im2col3x3s2(const float* y, float* col, int32_t B, int32_t Ti, int32_t Fi, int32_t C,
                                 tavsr_stream_t stream) {
  TAVSR_REQUIRE(y && col, TAVSR_EINVAL, "im2col: null pointer");
  TAVSR_REQUIRE(C % 4 == 0 && Ti >= 3 && Fi >= 3, TAVSR_EINVAL, "im2col: C %% 4 == 0 and Ti,Fi >= 3 required");
  const int To = (Ti - 3) / 2 + 1, Fo = (Fi - 3) / 2 + 1;
  int64_t total4 = (int64_t)B * To * Fo * 9 * (C / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(im2col3x3s2_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, y, col, B, Ti, Fi, To,
                     Fo, C / 4, total4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_col2im3x3s2_relu(const float* dcol, const float* yrelu, float* dz, int32_t B, int32_t Ti,
                                      int32_t Fi, int32_t C, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dcol && yrelu && dz, TAVSR_EINVAL, "col2im: null pointer");
  TAVSR_REQUIRE(C % 4 == 0 && Ti >= 3 && Fi >= 3, TAVSR_EINVAL, "col2im: C %% 4 == 0 and Ti,Fi >= 3 required");
  const int To = (Ti - 3) / 2 + 1, Fo = (Fi - 3) / 2 + 1;
  int64_t total4 = (int64_t)B * Ti * Fi * (C / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(col2im3x3s2_relu_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, dcol, yrelu, dz,
                     B, Ti, Fi, To, Fo, C / 4, total4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_transpose_inner(const float* in, float* out, int64_t nb, int32_t R, int32_t Cc, int32_t accumulate,
                                     tavsr_stream_t stream) {
  TAVSR_REQUIRE(in && out, TAVSR_EINVAL, "transpose_inner: null pointer");
  TAVSR_REQUIRE(nb <= 65535, TAVSR_EINVAL, "transpose_inner: batch too large");
  if (nb <= 0 || R <= 0 || Cc <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(transpose_inner_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), (unsigned)nb), dim3(32, 8), 0,
                     (hipStream_t)stream, in, out, nb, R, Cc, accumulate);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_utterance_mvn(const float* x, const int64_t* lens, float* y, int32_t B, int32_t T, int32_t F,
                                   tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && lens && y, TAVSR_EINVAL, "utterance_mvn: null pointer");
  TAVSR_REQUIRE(F <= 256, TAVSR_EUNSUPPORTED, "utterance_mvn: F <= 256 supported");
  if (B <= 0 || T <= 0) return TAVSR_OK;
  if (F % 4 == 0 && F >= 4 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL(utt_mvn_vec_kernel, dim3(UM_SLICES, B), dim3(256), 0, (hipStream_t)stream, x, lens, y, T, F);
  else
    hipLaunchKernelGGL(utt_mvn_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, lens, y, T, F);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
