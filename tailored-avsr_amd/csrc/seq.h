// Host-side helpers shared by the C-side sequencers (layer.hip, blocks.hip): entry points that enqueue the launches of a whole
// block / layer of the reference model over the primitive entry points of this library.  Internal.
#pragma once
#include <cmath>
#include <cstring>

#include "common.h"

#define TAVSR_HIP_CHECK(call)                                                                    \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      ::tavsr::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));     \
      return (int)e__;                                                                            \
    }                                                                                             \
  } while (0)

namespace tavsr {
namespace seq {

struct Bump {           // workspace carving (two queues run side by side: every launch gets its own region)
  float* base;
  int64_t cap, used;
  bool dry;
  bool overflow = false;
  // dry run: sizes only, but a non-null sentinel so that descriptors which switch on "is this pointer given" (rowstat)
  // plan the same launches as the real pass; real pass: never past the caller's capacity
  float* take(int64_t n) {
    n = (n + 63) / 64 * 64;
    float* p = dry ? reinterpret_cast<float*>(uintptr_t(64)) : base + used;
    if (!dry && used + n > cap) { overflow = true; p = nullptr; }
    used += n;
    return p;
  }
};

inline tavsr_gemm_desc lin(int M, int N, int K, const float* x, int64_t ldx, const float* w, const float* b, float* out, int64_t ldo) {
  tavsr_gemm_desc g;
  memset(&g, 0, sizeof g);
  g.M = M; g.N = N; g.K = K;
  g.A = x; g.lda = ldx; g.B = w; g.ldb = K; g.C = out; g.ldc = ldo;
  g.nb1 = g.nb2 = 1;
  g.bias = b;
  g.alpha = 1.f;
  return g;
}

inline int run_gemm(tavsr_gemm_desc& g, Bump& ws, hipStream_t s) {
  const int64_t need = tavsr_gemm_ws(&g);
  if (need > 0) { g.ws = ws.take(need); g.ws_floats = need; }
  if (ws.dry) return TAVSR_OK;
  TAVSR_REQUIRE(!ws.overflow, TAVSR_EINVAL, "workspace too small for a GEMM's split-K slabs");
  return tavsr_gemm(&g, (tavsr_stream_t)s);
}

}  // namespace seq
}  // namespace tavsr
