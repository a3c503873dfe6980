// Internal helpers shared by the gfx950 kernels of libtavsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tavsr.h"

namespace tavsr {

void set_error(const char* fmt, ...);

#define TAVSR_REQUIRE(cond, code, ...)   \
  do {                                   \
    if (!(cond)) {                       \
      ::tavsr::set_error(__VA_ARGS__);   \
      return (code);                     \
    }                                    \
  } while (0)

// Launch-status check: never synchronises (graph-capture safe).
#define TAVSR_LAUNCH_CHECK()                                                   \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      ::tavsr::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,        \
                         hipGetErrorString(e__));                              \
      return (int)e__;                                                         \
    }                                                                          \
  } while (0)

constexpr int kWave = 64;

// The lane-network reductions below use the row_bcast:15 / row_bcast:31 DPP controls, which exist on GFX9 / CDNA only, and this
// library is written for one ISA.  A device pass for anything else stops here instead of assembling garbage.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libtavsr_hip kernels are written for gfx950 (CDNA4): build with --offload-arch=gfx950"
#endif
// Debug builds (-DTAVSR_DEBUG_WAVE) trap when a whole-wave reduction is entered with lanes switched off: the result is read from
// lane 63 and the row broadcasts take their sources by lane number, so a partial wave would return a stale value silently.
#ifdef TAVSR_DEBUG_WAVE
#define TAVSR_ASSERT_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define TAVSR_ASSERT_FULL_WAVE() do { } while (0)
#endif

// Whole-wave reductions on the data-parallel-primitive lane network (quad permutes, row mirrors, row broadcasts; the result is read
// from lane 63 and is uniform): six steps of ~8 cycles against six dependent ds_bpermute round trips (~120 cycles each) of a
// __shfl_xor butterfly.  LayerNorm rows, softmax rows and the selection loops of the search step are chains of such reductions.
// All 64 lanes must be active (every caller reduces under wave-uniform control flow).  `IDENT`: what a lane that a step does
// not write contributes (0 for sums; the lane's own value for max / min).
#define TAVSR_DPP_F(V, CTRL, RMASK, OLD) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(OLD), __float_as_int(V), CTRL, RMASK, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
  TAVSR_ASSERT_FULL_WAVE();
  v += TAVSR_DPP_F(v, 0xB1, 0xf, 0.f);       // quad_perm [1, 0, 3, 2]
  v += TAVSR_DPP_F(v, 0x4E, 0xf, 0.f);       // quad_perm [2, 3, 0, 1]
  v += TAVSR_DPP_F(v, 0x141, 0xf, 0.f);      // row_half_mirror
  v += TAVSR_DPP_F(v, 0x140, 0xf, 0.f);      // row_mirror: every lane of a row of 16 holds the row's sum
  v += TAVSR_DPP_F(v, 0x142, 0xa, 0.f);      // row_bcast:15 into rows 1 and 3
  v += TAVSR_DPP_F(v, 0x143, 0xc, 0.f);      // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  TAVSR_ASSERT_FULL_WAVE();
  v = fmaxf(v, TAVSR_DPP_F(v, 0xB1, 0xf, v));
  v = fmaxf(v, TAVSR_DPP_F(v, 0x4E, 0xf, v));
  v = fmaxf(v, TAVSR_DPP_F(v, 0x141, 0xf, v));
  v = fmaxf(v, TAVSR_DPP_F(v, 0x140, 0xf, v));
  v = fmaxf(v, TAVSR_DPP_F(v, 0x142, 0xa, v));
  v = fmaxf(v, TAVSR_DPP_F(v, 0x143, 0xc, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_max(v); }
__device__ __forceinline__ int wave_min_dpp(int v) {
  TAVSR_ASSERT_FULL_WAVE();
#define TAVSR_DPP_I(CTRL, RMASK) v = min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, RMASK, 0xf, false))
  TAVSR_DPP_I(0xB1, 0xf);
  TAVSR_DPP_I(0x4E, 0xf);
  TAVSR_DPP_I(0x141, 0xf);
  TAVSR_DPP_I(0x140, 0xf);
  TAVSR_DPP_I(0x142, 0xa);
  TAVSR_DPP_I(0x143, 0xc);
#undef TAVSR_DPP_I
  return __builtin_amdgcn_readlane(v, 63);
}

// Activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp each) instead of libm's expf / erff and an IEEE
// division: Swish 6 instructions instead of ~35, GELU ~16 instead of ~50 (erff is two branches of ~45, both taken by a wave) -
// in a K = 256 GEMM epilogue the libm GELU cost 40 % of the tile's MFMA time, in the LayerNorm backward of cgMLP's gate half it
// was most of the launch.  erf: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute; everything here is ~1e-7 relative on the
// activation's value, three orders inside the 1e-4 parity bar (espnet: torch.nn.GELU() = erf form, Swish = x * sigmoid(x)).
__device__ __forceinline__ float sigmoid_fast(float z) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
// (E = exp(-x^2), shared by erf's tail and the normal density of the GELU derivative)
__device__ __forceinline__ float erf_from_exp(float x, float E) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return copysignf(fmaf(-p * t, E, 1.f), x);
}
__device__ __forceinline__ float act_fwd(int act, float z) {
  switch (act) {
    case TAVSR_ACT_RELU: return z > 0.f ? z : 0.f;
    case TAVSR_ACT_SWISH: return z * sigmoid_fast(z);
    case TAVSR_ACT_GELU: {
      const float E = __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);      // exp(-z^2 / 2)
      return 0.5f * z * (1.f + erf_from_exp(z * 0.70710678118654752440f, E));
    }
    default: return z;
  }
}
// derivative of the activation at pre-activation z
__device__ __forceinline__ float act_bwd(int act, float z) {
  switch (act) {
    case TAVSR_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case TAVSR_ACT_SWISH: {
      const float s = sigmoid_fast(z);
      return s * (1.f + z * (1.f - s));
    }
    case TAVSR_ACT_GELU: {
      const float E = __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);      // exp(-z^2 / 2)
      const float cdf = 0.5f * (1.f + erf_from_exp(z * 0.70710678118654752440f, E));
      return cdf + z * (0.39894228040143267794f * E);
    }
    default: return 1.f;
  }
}

// Philox4x32-10 counter-based generator (dropout masks, bootstrap resampling)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int probe_fork(hipStream_t forked, hipStream_t caller, bool at_join);   // probe.hip: race amplifier at C-side forks / joins

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace tavsr
