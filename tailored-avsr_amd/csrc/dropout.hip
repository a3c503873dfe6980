// Train-mode dropout (torch.nn.Dropout / F.dropout sites of the reference: encoder_layer.py:194,212,224,232-309,314,
// espnet attention.forward_attention, cgmlp csgu, positional encodings, decoder layers, src/ctc/ctc.py:143).
// y = x * keep / (1 - p) with keep ~ Bernoulli(1 - p) from a counter-based generator (Philox4x32-10): the mask of an
// element is a pure function of (seed, offset + element index), so the backward pass regenerates it instead of
// storing it, and one captured hipGraph draws fresh masks every replay because the seed lives in DEVICE memory and is
// advanced by a kernel inside the graph.  HBM-bound: 8 bytes per element.
#include "common.h"

namespace tavsr {

// element e of a call uses word (e & 3) of philox(counter = offset/4 + e/4); offset % 4 == 0
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                      uint32_t thr, float inv_keep, const uint64_t* __restrict__ seed,
                                                      uint64_t offset4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // group of 4 elements
  const int64_t e = i << 2;
  if (e >= n) return;
  const uint64_t s = seed[0], ctr = offset4 + (uint64_t)i;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)s, (uint32_t)(s >> 32), r);
  if (e + 3 < n && (((uintptr_t)(x + e) | (uintptr_t)(y + e)) & 15) == 0) {
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    float4 o;
    o.x = r[0] >= thr ? v.x * inv_keep : 0.f;
    o.y = r[1] >= thr ? v.y * inv_keep : 0.f;
    o.z = r[2] >= thr ? v.z * inv_keep : 0.f;
    o.w = r[3] >= thr ? v.w * inv_keep : 0.f;
    *reinterpret_cast<float4*>(y + e) = o;
  } else {
    for (int j = 0; j < 4 && e + j < n; ++j) y[e + j] = r[j] >= thr ? x[e + j] * inv_keep : 0.f;
  }
}

// y = a + alpha * dropout(t): the residual add of a dropped branch in one pass (encoder_layer.py:194,309,314:
// x + ff_scale * dropout(f(x))); t also receives dropout(t) when t_out != nullptr (kept for nothing: callers pass null)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                          float* __restrict__ y, int64_t n, uint32_t thr, float inv_keep,
                                                          float alpha, const uint64_t* __restrict__ seed, uint64_t offset4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e = i << 2;
  if (e >= n) return;
  const uint64_t s = seed[0], ctr = offset4 + (uint64_t)i;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)s, (uint32_t)(s >> 32), r);
  if (e + 3 < n && (((uintptr_t)(a + e) | (uintptr_t)(t + e) | (uintptr_t)(y + e)) & 15) == 0) {
    const float4 u = *reinterpret_cast<const float4*>(a + e), v = *reinterpret_cast<const float4*>(t + e);
    float4 o;
    o.x = u.x + alpha * (r[0] >= thr ? v.x * inv_keep : 0.f);
    o.y = u.y + alpha * (r[1] >= thr ? v.y * inv_keep : 0.f);
    o.z = u.z + alpha * (r[2] >= thr ? v.z * inv_keep : 0.f);
    o.w = u.w + alpha * (r[3] >= thr ? v.w * inv_keep : 0.f);
    *reinterpret_cast<float4*>(y + e) = o;
  } else {
    for (int j = 0; j < 4 && e + j < n; ++j) y[e + j] = a[e + j] + alpha * (r[j] >= thr ? t[e + j] * inv_keep : 0.f);
  }
}

// dz = dropout_mask(dh) * act'(z): backward of h = dropout(act(z)) in one pass (PositionwiseFeedForward's inner dropout)
__global__ __launch_bounds__(256) void dropout_act_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z,
                                                              float* __restrict__ dz, int64_t n, uint32_t thr, float inv_keep,
                                                              int act, const uint64_t* __restrict__ seed, uint64_t offset4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e = i << 2;
  if (e >= n) return;
  const uint64_t s = seed[0], ctr = offset4 + (uint64_t)i;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)s, (uint32_t)(s >> 32), r);
  if (e + 3 < n && (((uintptr_t)(dh + e) | (uintptr_t)(z + e) | (uintptr_t)(dz + e)) & 15) == 0) {
    const float4 d = *reinterpret_cast<const float4*>(dh + e), zz = *reinterpret_cast<const float4*>(z + e);
    float4 o;
    o.x = (r[0] >= thr ? d.x * inv_keep : 0.f) * act_bwd(act, zz.x);
    o.y = (r[1] >= thr ? d.y * inv_keep : 0.f) * act_bwd(act, zz.y);
    o.z = (r[2] >= thr ? d.z * inv_keep : 0.f) * act_bwd(act, zz.z);
    o.w = (r[3] >= thr ? d.w * inv_keep : 0.f) * act_bwd(act, zz.w);
    *reinterpret_cast<float4*>(dz + e) = o;
  } else {
    for (int j = 0; j < 4 && e + j < n; ++j) dz[e + j] = (r[j] >= thr ? dh[e + j] * inv_keep : 0.f) * act_bwd(act, z[e + j]);
  }
}

// step_out (optional): this step's own copy of the advanced seed - the dropout sites of ONE forward pass and of ITS backward
// pass read that copy, so a second forward before the first backward (micro-batches, a validation pass, a second model)
// cannot change the masks the backward regenerates
__global__ void rng_advance_kernel(uint64_t* __restrict__ seed, uint64_t* __restrict__ step_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint64_t s = seed[0] * 6364136223846793005ull + 1442695040888963407ull;
    seed[0] = s;
    if (step_out) step_out[0] = s;
  }
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_dropout(const float* x, float* y, int64_t n, float p, const uint64_t* seed_dev, uint64_t offset,
                             tavsr_stream_t stream) {
  TAVSR_REQUIRE((x && y && seed_dev) || n <= 0, TAVSR_EINVAL, "dropout: null pointer");
  TAVSR_REQUIRE(p >= 0.f && p < 1.f, TAVSR_EINVAL, "dropout: p must be in [0, 1)");
  TAVSR_REQUIRE(offset % 4 == 0, TAVSR_EALIGN, "dropout: offset must be a multiple of 4");
  if (n <= 0) return TAVSR_OK;
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, thr,
                     1.f / (1.f - p), seed_dev, offset / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_rng_advance(uint64_t* seed_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(seed_dev, TAVSR_EINVAL, "rng_advance: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed_dev, (uint64_t*)nullptr);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_rng_step(uint64_t* seed_dev, uint64_t* step_seed_dev, tavsr_stream_t stream) {
  TAVSR_REQUIRE(seed_dev && step_seed_dev, TAVSR_EINVAL, "rng_step: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed_dev, step_seed_dev);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_dropout_add(const float* a, const float* t, float* y, int64_t n, float p, float alpha,
                                 const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream) {
  TAVSR_REQUIRE((a && t && y && seed_dev) || n <= 0, TAVSR_EINVAL, "dropout_add: null pointer");
  TAVSR_REQUIRE(p >= 0.f && p < 1.f, TAVSR_EINVAL, "dropout_add: p must be in [0, 1)");
  TAVSR_REQUIRE(offset % 4 == 0, TAVSR_EALIGN, "dropout_add: offset must be a multiple of 4");
  if (n <= 0) return TAVSR_OK;
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, t, y, n,
                     (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), alpha, seed_dev, offset / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_dropout_act_bwd(const float* dh, const float* z, float* dz, int64_t n, float p, int32_t act,
                                     const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream) {
  TAVSR_REQUIRE((dh && z && dz && seed_dev) || n <= 0, TAVSR_EINVAL, "dropout_act_bwd: null pointer");
  TAVSR_REQUIRE(p >= 0.f && p < 1.f, TAVSR_EINVAL, "dropout_act_bwd: p must be in [0, 1)");
  TAVSR_REQUIRE(offset % 4 == 0, TAVSR_EALIGN, "dropout_act_bwd: offset must be a multiple of 4");
  if (n <= 0) return TAVSR_OK;
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(dropout_act_bwd_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, z, dz,
                     n, (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p), act, seed_dev, offset / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
