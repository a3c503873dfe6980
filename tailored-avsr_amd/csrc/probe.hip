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

// What does a phase boundary INSIDE a launch cost against a launch boundary?  ("a persistent FFN -> tail -> FFN phase
// pair with barrier-xcd-style seams: measure the seam against the finish + launch it replaces before building more".)  Two phases of the
// same shape - every workgroup writes `per_wg` floats of its own slab, then reads the slab of workgroup (g + 97) % G written in the phase
// before (the cross-workgroup dependency of a layer seam: partial outputs of other compute units) - either as two launches, or as one
// launch with a grid-wide barrier between them: per-XCD arrival counters (workgroup b runs on XCD b % 8), the last arrival of an XCD
// arrives at a top counter, the last XCD releases every XCD's generation word; lane 0 of each workgroup publishes with an agent-scope
// release fence before it arrives and takes an agent-scope acquire after it saw its generation.  Grid <= one workgroup per compute unit.
namespace tavsr {
struct SeamCtl { unsigned xcd[8][32]; unsigned top[32]; unsigned gen[8][32]; unsigned base[32]; };      // every word on a 128-byte line of its own

__device__ __forceinline__ void seam_phase(float* buf, long long per_wg, int phase, int G, int shift) {
  if (shift & 0x10000) return;                                   // barriers only
  const bool old = (shift & 0x20000) != 0;                       // read what was written a whole chain ago instead of a phase ago
  shift &= 0xffff;
  float* mine = buf + ((long long)phase * G + blockIdx.x) * per_wg;
  const float4 seed = make_float4((float)phase, 1.f, 2.f, 3.f);
  if (phase == 0) {
    for (long long i = threadIdx.x * 4ll; i < per_wg; i += blockDim.x * 4ll) *reinterpret_cast<float4*>(mine + i) = seed;
  } else {
    const float* src = buf + ((long long)(old ? (phase + 1) % 9 : phase - 1) * G + (blockIdx.x + shift) % G) * per_wg;
    for (long long i = threadIdx.x * 4ll; i < per_wg; i += blockDim.x * 4ll) {
      float4 v = *reinterpret_cast<const float4*>(src + i);
      v.x += 1.f;
      *reinterpret_cast<float4*>(mine + i) = v;
    }
  }
}

__global__ __launch_bounds__(256) void seam_phase_kernel(float* buf, long long per_wg, int phase, int shift) { seam_phase(buf, per_wg, phase, gridDim.x, shift); }

__global__ __launch_bounds__(256) void seam_fused_kernel(float* buf, long long per_wg, int nphase, SeamCtl* ctl, unsigned epoch0, int shift) {
  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const unsigned per_xcd = (unsigned)((G - xcd + 7) / 8);
  // barriers taken by earlier launches: device data, so that the launch can be captured and replayed (read before this launch's first
  // barrier by every workgroup; workgroup 0 moves it on behind the last one, when every workgroup has read it)
  __shared__ unsigned s_base;
  if (threadIdx.x == 0) s_base = __hip_atomic_load(&ctl->base[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  epoch0 += s_base;
  for (int ph = 0; ph < nphase; ++ph) {
    seam_phase(buf, per_wg, ph, G, shift);
    if (ph + 1 == nphase) break;
    const unsigned epoch = epoch0 + (unsigned)ph + 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned a = __hip_atomic_fetch_add(&ctl->xcd[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (a == epoch * per_xcd) {
        const unsigned t = __hip_atomic_fetch_add(&ctl->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (t == epoch * 8u)
          for (int x = 0; x < 8; ++x) __hip_atomic_store(&ctl->gen[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long long spins = 0;
      while (__hip_atomic_load(&ctl->gen[xcd][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1ll << 24))
        __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (ph + 2 == nphase && blockIdx.x == 0)
        __hip_atomic_store(&ctl->base[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}
}  // namespace tavsr

// kind 0: `nphase` launches; kind 1: one launch with nphase - 1 grid barriers.  ctl: sizeof(SeamCtl) = 2304 zeroed bytes that persist
// across calls (the barrier epochs only grow; the count of barriers taken so far lives in it, so the launch can be replayed from a graph;
// epoch0: 0).  buf: nphase * grid * per_wg floats.
extern "C" int tavsr_probe_seam(int32_t kind, int32_t grid, int64_t per_wg, int32_t nphase, float* buf, void* ctl, uint32_t epoch0,
                                tavsr_stream_t stream) {
  int shift = kind >> 8;          // (kind = form + 256 * (s + 1) + 0x1000000 * flags; shift 0: the default 97; flags 1: barriers only, 2: read old data)
  const int flags = shift >> 16;
  shift &= 0xffff;
  kind &= 255;
  TAVSR_REQUIRE(buf && (kind == 0 || ctl) && grid >= 8 && grid <= 256 && nphase >= 1 && per_wg >= 4 && per_wg % 4 == 0, TAVSR_EINVAL,
                "probe_seam: 8 <= grid <= 256 (one workgroup per compute unit), per_wg %% 4 == 0");
  if (kind == 0) {
    for (int ph = 0; ph < nphase; ++ph)
      hipLaunchKernelGGL(tavsr::seam_phase_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, buf, (long long)per_wg, ph,
                         (shift ? shift - 1 : 97) | (flags << 16));
  } else {
    hipLaunchKernelGGL(tavsr::seam_fused_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, buf, (long long)per_wg, nphase,
                       (tavsr::SeamCtl*)ctl, epoch0, (shift ? shift - 1 : 97) | (flags << 16));
  }
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_stream_create(tavsr_stream_t* out) {
  TAVSR_REQUIRE(out, TAVSR_EINVAL, "stream_create: null pointer");
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  TAVSR_REQUIRE(e == hipSuccess, TAVSR_EINVAL, "stream_create: %s", hipGetErrorString(e));
  *out = (tavsr_stream_t)s;
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
