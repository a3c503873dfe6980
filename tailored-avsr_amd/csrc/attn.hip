// Attention glue kernels around the batched MFMA GEMMs (scores and context are tavsr_gemm calls):
//   * q + pos_bias_u / q + pos_bias_v              (espnet RelPositionMultiHeadedAttention.forward)
//   * rel_shift + scale + key mask + softmax + zero-fill, forward and backward
// Score buffers are laid out [H, B, T1, T2] so that the positional-projection gradient becomes one
// K = B*T1 GEMM per head.  One wave per score row; rows are re-read from L2 instead of held in
// registers so any T2 works.
#include <float.h>
#include <algorithm>

#include "common.h"

namespace tavsr {

// qu[m][c] = q[m*ldq + c] + u[c];  qv[m][c] = q[m*ldq + c] + v[c]     (c < D, float4 lanes)
__global__ void add_head_bias_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ u,
                                     const float* __restrict__ v, float* __restrict__ qu, float* __restrict__ qv,
                                     int64_t total4, int D4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int64_t m = i / D4;
  int c = (int)(i % D4) * 4;
  float4 x = *reinterpret_cast<const float4*>(q + m * ldq + c);
  float4 a = *reinterpret_cast<const float4*>(u + c);
  float4 b = *reinterpret_cast<const float4*>(v + c);
  *reinterpret_cast<float4*>(qu + m * D4 * 4 + c) = make_float4(x.x + a.x, x.y + a.y, x.z + a.z, x.w + a.w);
  *reinterpret_cast<float4*>(qv + m * D4 * 4 + c) = make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w);
}

// attn[h,b,i,j] = softmax_j( (ac[h,b,i,j] + bd[h,b,i, T-1-i+j]) * scale ) over valid keys, 0 elsewhere.
// valid key: j < klens[b] (if klens) and j <= i (if causal).  Equivalent to the reference's
// masked_fill(min) -> softmax -> masked_fill(0) because exp(min - max) == 0 in fp32.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ ac, const float* __restrict__ bd,
                                                          const int64_t* __restrict__ klens, float* __restrict__ attn,
                                                          int H, int B, int T1, int T2, int W, int64_t ld_s,
                                                          int64_t ld_w, float scale, int causal, float* __restrict__ pv,
                                                          uint32_t thr, float inv_keep, const uint64_t* __restrict__ seed,
                                                          uint64_t offset4) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)H * B * T1) return;
  const int i = (int)(row % T1);
  const int b = (int)((row / T1) % B);
  int nk = klens ? (int)min((int64_t)T2, klens[b]) : T2;
  if (causal) nk = min(nk, i + 1);
  const float* a = ac + row * ld_s;
  const float* p = bd ? bd + row * ld_w + (T1 - 1 - i) : nullptr;
  float* o = attn + row * ld_s;
  float mx = -FLT_MAX;
  for (int j = lane; j < nk; j += 64) {
    float s = (a[j] + (p ? p[j] : 0.f)) * scale;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < nk; j += 64) {
    float s = (a[j] + (p ? p[j] : 0.f)) * scale;
    sum += expf(s - mx);
  }
  sum = wave_sum(sum);
  const float inv = nk > 0 ? 1.f / sum : 0.f;
  for (int j = lane; j < T2; j += 64) {
    float r = 0.f;
    if (j < nk) {
      float s = (a[j] + (p ? p[j] : 0.f)) * scale;
      r = expf(s - mx) * inv;
    }
    o[j] = r;
    if (pv) {      // dropout of the probabilities in the same pass: element e = row*ld_s + j of the flat tensor, as tavsr_dropout
      const uint64_t ctr = offset4 + (uint64_t)(row * (ld_s >> 2)) + (uint64_t)(j >> 2), sd = seed[0];
      uint32_t w[4];
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
      pv[row * ld_s + j] = w[j & 3] >= thr ? r * inv_keep : 0.f;
    }
  }
  if (pv)
    for (int j = T2 + lane; j < ld_s; j += 64) pv[row * ld_s + j] = 0.f;      // row padding (dropout of the zeros there)
}

// ds = attn * (dattn - sum_j attn*dattn) * scale ; optionally also the skewed copy
// ds_skew[h,b,i, T-1-i+j] = ds[h,b,i,j] (zeros elsewhere) that feeds the positional GEMMs.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ attn,
                                                          const float* __restrict__ dattn, float* __restrict__ ds,
                                                          float* __restrict__ ds_skew, int64_t rows, int T1, int T2,
                                                          int W, int64_t ld_s, int64_t ld_w, float scale, int drop, uint32_t thr,
                                                          float inv_keep, const uint64_t* __restrict__ seed, uint64_t offset4) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int i = (int)(row % T1);
  const float* a = attn + row * ld_s;
  const float* g0 = dattn + row * ld_s;
  // drop: dattn is the gradient w.r.t. the DROPPED probabilities; the mask of the forward site is regenerated here
  auto grad = [&](int j) {
    float v = g0[j];
    if (drop) {
      const uint64_t ctr = offset4 + (uint64_t)(row * (ld_s >> 2)) + (uint64_t)(j >> 2), sd = seed[0];
      uint32_t w[4];
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
      v = w[j & 3] >= thr ? v * inv_keep : 0.f;
    }
    return v;
  };
  float dot = 0.f;
  for (int j = lane; j < T2; j += 64) dot += a[j] * grad(j);
  dot = wave_sum(dot);
  float* o = ds + row * ld_s;
  float* sk = ds_skew ? ds_skew + row * ld_w : nullptr;
  const int off = T1 - 1 - i;
  if (sk)
    for (int c = lane; c < W; c += 64)
      if (c < off || c >= off + T2) sk[c] = 0.f;
  for (int j = lane; j < T2; j += 64) {
    float v = a[j] * (grad(j) - dot) * scale;
    o[j] = v;
    if (sk) sk[off + j] = v;
  }
}

// out = a*x + b*y  (y may be NULL), float4 lanes with a scalar tail
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b,
                             float* __restrict__ out, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 xv = *reinterpret_cast<const float4*>(x + i);
    float4 r = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
    if (y) {
      float4 yv = *reinterpret_cast<const float4*>(y + i);
      r.x += b * yv.x; r.y += b * yv.y; r.z += b * yv.z; r.w += b * yv.w;
    }
    *reinterpret_cast<float4*>(out + i) = r;
  } else {
    for (; i < n; ++i) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
  }
}

__global__ void axpby_scalar_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b,
                                    float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// out = x * (c * s[0]) with s a device scalar (upstream loss gradient): no host round trip
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float c,
                                 float* __restrict__ out, int64_t n) {
  const float f = c * s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] * f;
}

// strided 2-D variant: out[m*ldo + c] = a*x[m*ldx + c] + b*y[m*ldy + c], c < N (N % 4 == 0)
__global__ void axpby2d_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t ldy,
                               float a, float b, float* __restrict__ out, int64_t ldo, int64_t total4, int N4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int64_t m = i / N4;
  int c = (int)(i % N4) * 4;
  float4 xv = *reinterpret_cast<const float4*>(x + m * ldx + c);
  float4 r = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
  if (y) {
    float4 yv = *reinterpret_cast<const float4*>(y + m * ldy + c);
    r.x += b * yv.x; r.y += b * yv.y; r.z += b * yv.z; r.w += b * yv.w;
  }
  *reinterpret_cast<float4*>(out + m * ldo + c) = r;
}

// dz[m*ld + c] = dh[m*ld + c] * act'(z[m*ld + c])   (in place allowed)
__global__ void act_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z, float* __restrict__ dz,
                               int64_t n, int act) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 d = *reinterpret_cast<const float4*>(dh + i);
    float4 zz = *reinterpret_cast<const float4*>(z + i);
    d.x *= act_bwd(act, zz.x); d.y *= act_bwd(act, zz.y); d.z *= act_bwd(act, zz.z); d.w *= act_bwd(act, zz.w);
    *reinterpret_cast<float4*>(dz + i) = d;
  } else {
    for (; i < n; ++i) dz[i] = dh[i] * act_bwd(act, z[i]);
  }
}

}  // namespace tavsr

using namespace tavsr;

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int tavsr_add_head_bias(const float* q, int64_t ldq, const float* u, const float* v, float* qu, float* qv,
                                   int64_t M, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(q && u && v && qu && qv, TAVSR_EINVAL, "add_head_bias: null pointer");
  TAVSR_REQUIRE(D % 4 == 0 && ldq % 4 == 0 && al16(q) && al16(u) && al16(v) && al16(qu) && al16(qv), TAVSR_EALIGN,
                "add_head_bias: 16-byte alignment required");
  int64_t total4 = M * (D / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(add_head_bias_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, q, ldq, u, v,
                     qu, qv, total4, D / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_softmax_fwd(const float* ac, const float* bd, const int64_t* klens, float* attn, int32_t H,
                                 int32_t B, int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w,
                                 float scale, int32_t causal, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ac && attn, TAVSR_EINVAL, "softmax_fwd: null pointer");
  TAVSR_REQUIRE(ld_s >= T2 && (!bd || ld_w >= W), TAVSR_EINVAL, "softmax_fwd: leading dimension too small");
  TAVSR_REQUIRE(!bd || (T1 == T2 && W == 2 * T1 - 1), TAVSR_EINVAL,
                "softmax_fwd: rel-pos term needs T1 == T2 and W == 2*T1-1 (got %d, %d, %d)", T1, T2, W);
  int64_t rows = (int64_t)H * B * T1;
  if (rows <= 0 || T2 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, ac, bd, klens, attn, H,
                     B, T1, T2, W, ld_s, ld_w, scale, causal, (float*)nullptr, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// softmax_fwd that also writes pv = dropout(attn, p) with the mask tavsr_dropout(attn, ..., offset) would draw (the
// probabilities' dropout of espnet's forward_attention) - one launch and one read of the probabilities less
extern "C" int tavsr_softmax_dropout_fwd(const float* ac, const float* bd, const int64_t* klens, float* attn, float* pv, int32_t H,
                                         int32_t B, int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale,
                                         int32_t causal, float p, const uint64_t* seed_dev, uint64_t offset,
                                         tavsr_stream_t stream) {
  TAVSR_REQUIRE(ac && attn && pv && seed_dev, TAVSR_EINVAL, "softmax_dropout_fwd: null pointer");
  TAVSR_REQUIRE(ld_s >= T2 && ld_s % 4 == 0 && (!bd || ld_w >= W), TAVSR_EINVAL, "softmax_dropout_fwd: rows must be padded to 4");
  TAVSR_REQUIRE(!bd || (T1 == T2 && W == 2 * T1 - 1), TAVSR_EINVAL, "softmax_dropout_fwd: bad rel-pos geometry");
  TAVSR_REQUIRE(p >= 0.f && p < 1.f && offset % 4 == 0, TAVSR_EINVAL, "softmax_dropout_fwd: p in [0, 1), offset %% 4 == 0");
  int64_t rows = (int64_t)H * B * T1;
  if (rows <= 0 || T2 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, ac, bd, klens, attn, H,
                     B, T1, T2, W, ld_s, ld_w, scale, causal, pv, (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), seed_dev,
                     offset / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_softmax_bwd(const float* attn, const float* dattn, float* ds, float* ds_skew, int32_t H,
                                 int32_t B, int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w,
                                 float scale, tavsr_stream_t stream) {
  TAVSR_REQUIRE(attn && dattn && ds, TAVSR_EINVAL, "softmax_bwd: null pointer");
  TAVSR_REQUIRE(ld_s >= T2 && (!ds_skew || ld_w >= W), TAVSR_EINVAL, "softmax_bwd: leading dimension too small");
  TAVSR_REQUIRE(!ds_skew || (T1 == T2 && W == 2 * T1 - 1), TAVSR_EINVAL, "softmax_bwd: bad skew geometry");
  int64_t rows = (int64_t)H * B * T1;
  if (rows <= 0 || T2 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, attn, dattn, ds,
                     ds_skew, rows, T1, T2, W, ld_s, ld_w, scale, 0, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// softmax_bwd whose dattn is the gradient of the dropped probabilities of tavsr_softmax_dropout_fwd (same p, offset)
extern "C" int tavsr_softmax_dropout_bwd(const float* attn, const float* dpv, float* ds, float* ds_skew, int32_t H, int32_t B,
                                         int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale, float p,
                                         const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream) {
  TAVSR_REQUIRE(attn && dpv && ds && seed_dev, TAVSR_EINVAL, "softmax_dropout_bwd: null pointer");
  TAVSR_REQUIRE(ld_s >= T2 && ld_s % 4 == 0 && (!ds_skew || ld_w >= W), TAVSR_EINVAL, "softmax_dropout_bwd: rows must be padded to 4");
  TAVSR_REQUIRE(!ds_skew || (T1 == T2 && W == 2 * T1 - 1), TAVSR_EINVAL, "softmax_dropout_bwd: bad skew geometry");
  TAVSR_REQUIRE(p >= 0.f && p < 1.f && offset % 4 == 0, TAVSR_EINVAL, "softmax_dropout_bwd: p in [0, 1), offset %% 4 == 0");
  int64_t rows = (int64_t)H * B * T1;
  if (rows <= 0 || T2 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, attn, dpv, ds, ds_skew, rows, T1,
                     T2, W, ld_s, ld_w, scale, 1, (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), seed_dev, offset / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_axpby(const float* x, const float* y, float a, float b, float* out, int64_t n,
                           tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && out, TAVSR_EINVAL, "axpby: null pointer");
  if (n <= 0) return TAVSR_OK;
  if (al16(x) && al16(y) && al16(out))
    hipLaunchKernelGGL(axpby_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, out, n);
  else
    hipLaunchKernelGGL(axpby_scalar_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, out, n);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_axpby2d(const float* x, int64_t ldx, const float* y, int64_t ldy, float a, float b, float* out,
                             int64_t ldo, int64_t M, int32_t N, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && out, TAVSR_EINVAL, "axpby2d: null pointer");
  TAVSR_REQUIRE(N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && al16(x) && al16(y) && al16(out),
                TAVSR_EALIGN, "axpby2d: 16-byte alignment required");
  int64_t total4 = M * (N / 4);
  if (total4 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(axpby2d_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, a, b,
                     out, ldo, total4, N / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_act_bwd(const float* dh, const float* z, float* dz, int64_t n, int32_t act,
                             tavsr_stream_t stream) {
  TAVSR_REQUIRE(dh && z && dz, TAVSR_EINVAL, "act_bwd: null pointer");
  TAVSR_REQUIRE(al16(dh) && al16(z) && al16(dz), TAVSR_EALIGN, "act_bwd: 16-byte alignment required");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, dh, z, dz, n, act);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_scale_dev(const float* x, const float* s, float c, float* out, int64_t n, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && s && out, TAVSR_EINVAL, "scale_dev: null pointer");
  if (n <= 0) return TAVSR_OK;
  int blocks = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  hipLaunchKernelGGL(scale_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, s, c, out, n);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
