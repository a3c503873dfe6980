// Test instrumentation of the stream plumbing (no product path calls these with a non-zero time):
//   tavsr_spin(us, stream)      one wave that polls the 100 MHz wall clock for `us` microseconds - a launch that does nothing but
//                               occupy its queue, so that a missing dependency between two queues becomes deterministic;
//   tavsr_race_probe(us, mode)  arms the same delay inside the C-side sequencers that fork a second queue themselves
//                               (tavsr_branchformer_layer_fwd): mode 0 = head of the forked section, 1 = behind the join on the
//                               calling queue, 2 = alternately per call.  tavsr/ops.py (TAVSR_RACE_PROBE) drives both.
// The reference is single-queue (src/models/espnet_model.py:258-356 runs on torch's current stream): results must not depend on
// how the launches of one step are spread over queues; tests/test_gpu_streams.py holds that.
#include "common.h"

namespace tavsr {

__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static float g_probe_us = 0.f;
static int g_probe_mode = 2;
static unsigned g_probe_tick = 0;

int probe_fork(hipStream_t forked, hipStream_t caller, bool at_join) {
  if (g_probe_us <= 0.f) return TAVSR_OK;
  if (!at_join) ++g_probe_tick;
  const int mode = g_probe_mode == 2 ? (int)(g_probe_tick & 1u) : g_probe_mode;
  if (mode == (at_join ? 1 : 0)) return tavsr_spin(g_probe_us, (tavsr_stream_t)(at_join ? caller : forked));
  return TAVSR_OK;
}

}  // namespace tavsr

extern "C" int tavsr_spin(float us, tavsr_stream_t stream) {
  TAVSR_REQUIRE(us >= 0.f && us <= 1e6f, TAVSR_EINVAL, "spin: 0 .. 1e6 microseconds");
  if (us == 0.f) return TAVSR_OK;
  hipLaunchKernelGGL(tavsr::spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0));
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// What does a dependent launch cost before it does anything?  kind 0: every thread returns; 1: every thread reads one float4 of
// `buf` (element (block * threads + thread) % n4) and writes it back + 1 - one memory round trip and one store, the skeleton of
// a one-token Linear; 2: the same with a workgroup barrier and a second dependent read (an epilogue operand).  Timed in a
// captured chain by scripts/launch_floor.py at the grids of the search step.
namespace tavsr {
__global__ void probe_launch_kernel(int kind, float* buf, long long n4) {
  if (kind == 0) return;
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) % n4;
  float4 v = reinterpret_cast<const float4*>(buf)[i];
  if (kind == 2) {
    __syncthreads();
    const long long j = (i + (long long)(v.x != 12345.f) * 977) % n4;
    const float4 w = reinterpret_cast<const float4*>(buf)[j];
    v.y += w.y;
  }
  v.x += 1.f;
  reinterpret_cast<float4*>(buf)[i] = v;
}
}  // namespace tavsr

extern "C" int tavsr_probe_launch(int32_t kind, int32_t grid, int32_t block, float* buf, int64_t n, tavsr_stream_t stream) {
  TAVSR_REQUIRE(kind >= 0 && kind <= 2 && grid > 0 && block > 0 && block <= 1024, TAVSR_EINVAL, "probe_launch: kind 0..2, block <= 1024");
  TAVSR_REQUIRE(kind == 0 || (buf && n >= 4), TAVSR_EINVAL, "probe_launch: a buffer of at least 4 floats");
  hipLaunchKernelGGL(tavsr::probe_launch_kernel, dim3((unsigned)grid), dim3((unsigned)block), 0, (hipStream_t)stream, kind, buf,
                     (long long)(n / 4));
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_race_probe(float us, int mode) {
  TAVSR_REQUIRE(us >= 0.f && us <= 1e6f && mode >= 0 && mode <= 2, TAVSR_EINVAL, "race_probe: us in 0 .. 1e6, mode 0 / 1 / 2");
  tavsr::g_probe_us = us;
  tavsr::g_probe_mode = mode;
  return TAVSR_OK;
}

// Box calibration (bench.py `box` object): what THIS device delivers on the one fp32 matrix instruction every GEMM of the
// path issues, with nothing else in the way - 4 independent accumulator tiles per wave, 8 waves per CU, no memory traffic.
// FLOPs = blocks x 4 waves x iters x 4 MFMAs x 4096.
namespace tavsr {
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = 1.0f + 1e-9f * threadIdx.x, y = 1.0f - 1e-9f * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  const float s = a0[0] + a1[1] + a2[2] + a3[3];
  if (s == 12345.678f) sink[0] = s;      // (never true: keeps the loop alive)
}

// The same loop fed with DATA: 8 operand pairs per lane from memory (the caller's random numbers), a different pair for every
// consecutive instruction - the constant operands above barely toggle the multiplier inputs, real activations do, and the
// device's power management answers to that.
__global__ __launch_bounds__(256) void mfma_peak_data_kernel(int iters, const float* __restrict__ ops, float* sink) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    x[j] = ops[(j * 2) * 256 + threadIdx.x];
    y[j] = ops[(j * 2 + 1) * 256 + threadIdx.x];
  }
  for (int i = 0; i < iters; i += 2) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[0], y[0], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[1], y[1], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[2], y[2], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[3], y[3], a3, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[4], y[4], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[5], y[5], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[6], y[6], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[7], y[7], a3, 0, 0, 0);
  }
  const float s = a0[0] + a1[1] + a2[2] + a3[3];
  if (s == 12345.678f) sink[0] = s;
}
}  // namespace tavsr

extern "C" int tavsr_mfma_peak_f32_data(int32_t iters, int32_t blocks, const float* operands, float* sink, tavsr_stream_t stream) {
  TAVSR_REQUIRE(iters > 0 && iters % 2 == 0 && blocks > 0 && blocks <= 65536 && operands && sink, TAVSR_EINVAL,
                "mfma_peak_f32_data: an even iters > 0, blocks > 0, 4096 operand words and a sink word");
  hipLaunchKernelGGL(tavsr::mfma_peak_data_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, operands, sink);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_mfma_peak_f32(int32_t iters, int32_t blocks, float* sink, tavsr_stream_t stream) {
  TAVSR_REQUIRE(iters > 0 && blocks > 0 && blocks <= 65536 && sink, TAVSR_EINVAL, "mfma_peak_f32: iters, blocks > 0 and a sink word");
  hipLaunchKernelGGL(tavsr::mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
