// Optimizer step of the reference's training harness (avsr_main.py:50-54): torch.optim.Adam(betas=(0.9, 0.98), eps=1e-9)
// driven by the Noam learning-rate wrapper (src/schedulers/noam.py:29-46,72-81; src/utils/scheduler.py:27-34), fused
// into ONE pass over flat fp32 buffers (parameters, gradient, exp_avg, exp_avg_sq): 7 x 4 bytes of HBM traffic per
// parameter, HBM-bound.  Arithmetic follows torch's single-tensor Adam operation by operation so that parameters
// track the reference's to fp32 rounding.
#include <math.h>

#include <algorithm>

#include "common.h"

namespace tavsr {

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                        int64_t n, float beta1, float beta2, float eps, float step_size,
                                                        float bc2_sqrt, float grad_scale, float decay) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 P = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    const float4 G4 = reinterpret_cast<const float4*>(g)[i];
    float* pp = &P.x; float* mm = &M.x; float* vv = &V.x;
    const float* gg = &G4.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * grad_scale;
      mm[j] = mm[j] + (gr - mm[j]) * (1.f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
      vv[j] = vv[j] * beta2 + (1.f - beta2) * gr * gr;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] = pp[j] * decay - step_size * (mm[j] / denom);          // [param.mul_(1 - lr * weight_decay);] param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
  // tail (n % 4 elements) by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    const float gr = g[k] * grad_scale;
    m[k] = m[k] + (gr - m[k]) * (1.f - beta1);
    v[k] = v[k] * beta2 + (1.f - beta2) * gr * gr;
    p[k] = p[k] * decay - step_size * (m[k] / (sqrtf(v[k]) / bc2_sqrt + eps));
  }
}

// Gradient bucket pack / unpack for the data-parallel exchange (tavsr/dp.py): tensor t (n[t] floats at ptrs[t]) <->
// flat[off[t] .. off[t] + n[t]).  One launch for a whole bucket of hundreds of parameters: grid (chunks, tensors).
__global__ __launch_bounds__(256) void bucket_copy_kernel(float* const* __restrict__ ptrs, const int64_t* __restrict__ off,
                                                          const int64_t* __restrict__ n, float* __restrict__ flat, float scale,
                                                          int to_flat) {
  const int t = blockIdx.y;
  float* p = ptrs[t];
  float* f = flat + off[t];
  const int64_t cnt = n[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
    if (to_flat) f[i] = p[i];
    else p[i] = f[i] * scale;
  }
}

// dst[t][i] += src[t][i] for up to kMultiAddMax tensors in one launch (pointers travel in the kernel arguments: nothing to
// stage in device memory, capturable).  Sums the shared parameters' gradients of the two modality streams of a tailored
// AV layer (14 tensors per layer, most of them a few hundred floats).
constexpr int kMultiAddMax = 24;
struct MultiAddArgs {
  float* dst[kMultiAddMax];
  const float* src[kMultiAddMax];
  int64_t n[kMultiAddMax];
};
__global__ __launch_bounds__(256) void multi_add_kernel(const MultiAddArgs a) {
  const int t = blockIdx.y;
  float* d = a.dst[t];
  const float* s = a.src[t];
  const int64_t cnt = a.n[t];
  const bool vec = (((uintptr_t)d | (uintptr_t)s) & 15) == 0;
  const int64_t n4 = vec ? cnt >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 u = reinterpret_cast<float4*>(d)[i];
    const float4 v = reinterpret_cast<const float4*>(s)[i];
    u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    reinterpret_cast<float4*>(d)[i] = u;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) d[i] += s[i];
}

// dst[t] <- src[t] (nbytes[t] bytes, any dtype) for up to kMultiAddMax buffers in one launch: the commit of the beam
// search's re-ordered state (six arrays of three dtypes) back into the buffers the captured step reads
struct MultiCopyArgs {
  void* dst[kMultiAddMax];
  const void* src[kMultiAddMax];
  int64_t n[kMultiAddMax];
  int64_t* counters;      // (nullable) ncounters int64 words incremented by one by this launch: the step counter of a captured search step
  int ncounters;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(const MultiCopyArgs a) {
  const int t = blockIdx.y;
  char* d = static_cast<char*>(a.dst[t]);
  const char* s = static_cast<const char*>(a.src[t]);
  const int64_t cnt = a.n[t];
  const bool vec = (((uintptr_t)d | (uintptr_t)s) & 15) == 0;
  const int64_t n16 = vec ? cnt >> 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
  for (int64_t i = n16 * 16 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) d[i] = s[i];
  if (a.counters && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < a.ncounters) a.counters[threadIdx.x] += 1;
}

}  // namespace tavsr

using namespace tavsr;

static int multi_copy_impl(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers, int64_t* counters,
                           int32_t ncounters, tavsr_stream_t stream);

extern "C" int tavsr_multi_copy(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers,
                                tavsr_stream_t stream) {
  return multi_copy_impl(dst, src, nbytes, nbuffers, nullptr, 0, stream);
}

extern "C" int tavsr_multi_copy_inc(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers,
                                    int64_t* counters, int32_t ncounters, tavsr_stream_t stream) {
  TAVSR_REQUIRE(counters && ncounters > 0 && ncounters <= 64 && nbuffers > 0, TAVSR_EINVAL,
                "multi_copy_inc: 1..64 counters and at least one buffer");
  return multi_copy_impl(dst, src, nbytes, nbuffers, counters, ncounters, stream);
}

static int multi_copy_impl(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers, int64_t* counters,
                           int32_t ncounters, tavsr_stream_t stream) {
  TAVSR_REQUIRE(nbuffers <= 0 || (dst && src && nbytes), TAVSR_EINVAL, "multi_copy: null table");
  TAVSR_REQUIRE(nbuffers <= kMultiAddMax, TAVSR_EUNSUPPORTED, "multi_copy: at most %d buffers per call", kMultiAddMax);
  if (nbuffers <= 0) return TAVSR_OK;
  MultiCopyArgs a{};
  int64_t mx = 0;
  for (int t = 0; t < nbuffers; ++t) {
    TAVSR_REQUIRE(nbytes[t] == 0 || (dst[t] && src[t]), TAVSR_EINVAL, "multi_copy: null buffer %d", t);
    a.dst[t] = dst[t]; a.src[t] = src[t]; a.n[t] = nbytes[t];
    mx = std::max(mx, nbytes[t]);
  }
  a.counters = counters;
  a.ncounters = ncounters;
  if (mx <= 0 && !counters) return TAVSR_OK;
  const int64_t chunks = std::min<int64_t>(std::max<int64_t>(1, (mx / 16 + 255) / 256), 512);
  hipLaunchKernelGGL(tavsr::multi_copy_kernel, dim3((unsigned)chunks, (unsigned)nbuffers), dim3(256), 0, (hipStream_t)stream, a);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_multi_add(float* const* dst, const float* const* src, const int64_t* n, int32_t ntensors,
                               tavsr_stream_t stream) {
  TAVSR_REQUIRE(ntensors <= 0 || (dst && src && n), TAVSR_EINVAL, "multi_add: null table");
  for (int t0 = 0; t0 < ntensors; t0 += kMultiAddMax) {
    MultiAddArgs a{};
    const int cnt = std::min(kMultiAddMax, ntensors - t0);
    int64_t mx = 0;
    for (int t = 0; t < cnt; ++t) {
      TAVSR_REQUIRE(n[t0 + t] == 0 || (dst[t0 + t] && src[t0 + t]), TAVSR_EINVAL, "multi_add: null tensor %d", t0 + t);
      a.dst[t] = dst[t0 + t];
      a.src[t] = src[t0 + t];
      a.n[t] = n[t0 + t];
      mx = std::max(mx, n[t0 + t]);
    }
    if (mx <= 0) continue;
    const int64_t chunks = std::min<int64_t>(std::max<int64_t>(1, (mx / 4 + 255) / 256), 512);
    hipLaunchKernelGGL(tavsr::multi_add_kernel, dim3((unsigned)chunks, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a);
    TAVSR_LAUNCH_CHECK();
  }
  return TAVSR_OK;
}

extern "C" int tavsr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int64_t step, float grad_scale, tavsr_stream_t stream) {
  TAVSR_REQUIRE((p && g && m && v) || n <= 0, TAVSR_EINVAL, "adam_step: null pointer");
  TAVSR_REQUIRE(step >= 1, TAVSR_EINVAL, "adam_step: step counts from 1");
  TAVSR_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                TAVSR_EALIGN, "adam_step: 16-byte aligned flat buffers required");
  if (n <= 0) return TAVSR_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const int64_t n4 = n >> 2;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)std::max<int64_t>(1, (n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     p, g, m, v, n4, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, decay);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, int64_t step, float grad_scale, tavsr_stream_t stream) {
  return tavsr_adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, 0.f, step, grad_scale, stream);
}

extern "C" int tavsr_bucket_copy(float* const* ptrs_dev, const int64_t* off_dev, const int64_t* n_dev, int32_t ntensors,
                                 float* flat, float scale, int32_t to_flat, int64_t max_n, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ptrs_dev && off_dev && n_dev && flat, TAVSR_EINVAL, "bucket_copy: null pointer");
  if (ntensors <= 0) return TAVSR_OK;
  TAVSR_REQUIRE(ntensors <= 65535, TAVSR_EINVAL, "bucket_copy: at most 65535 tensors per launch");
  const int64_t chunks = std::min<int64_t>(std::max<int64_t>(1, (max_n + 255) / 256), 64);
  hipLaunchKernelGGL(tavsr::bucket_copy_kernel, dim3((unsigned)chunks, (unsigned)ntensors), dim3(256), 0, (hipStream_t)stream,
                     ptrs_dev, off_dev, n_dev, flat, scale, to_flat);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
