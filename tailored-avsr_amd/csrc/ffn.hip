// Fused position-wise feed-forward chain for gfx950 (espnet PositionwiseFeedForward inside its residual block:
// src/encoder/branchformer/encoder_layer.py:191-194,311-314, the tailored AV layer's shared FFNs, the decoder FFN):
//     forward : y = x + scale * dropout(W2 dropout(act(W1 LN(x) + b1)) + b2)
//     backward: dn = ((alpha * dyd) W2 * mask / keep * act'(z)) W1            (the data-gradient chain; dz is kept for dW1)
// One kernel runs BOTH GEMMs of a direction: a workgroup owns 32 rows and a slice of the hidden units; per sub-chunk of
// 128 hidden units each of its 4 waves computes one 32x32 tile of the first product (K = d_model), the epilogue (bias,
// pre-activation store, activation, dropout / mask * act') runs on the accumulator registers, the 32x128 result goes
// through LDS once (accumulator layout -> A-operand layout) and is contracted straight away against the matching slice
// of the second weight into accumulators that persist over the workgroup's hidden slice.  The hidden activations make
// no HBM round trip between the GEMMs (they are written once when the backward needs them), the LayerNorm is the
// prologue, and one launch carries 2 x 3.3 GFLOP at M = 3168 instead of two launches whose 64x64 tiles live for 8 K-steps.
// The hidden dimension is split over S workgroups per row tile (so that ~2 workgroups per CU exist); each writes its
// partial [32 x d_model] output to a slab and the finishing kernel sums the S slabs in fixed order (deterministic).
//
// v_mfma_f32_32x32x2_f32 (exact fp32).  A operands come from LDS (k-contiguous rows, 16-byte chunk c of row r stored at
// chunk c ^ (r & 15) inside its group of 16: conflict-free ds_read_b128); B operands (weights) are read from global
// memory / L2 straight into registers - every weight element is used by exactly one wave of a workgroup, LDS staging
// would not save a byte.  Both weights are read k-contiguous (16-byte loads feed four MFMAs): the backward chain takes the
// TRANSPOSED weights (w2^T as first, w1^T as second operand; two 2 MB transposes per block and step).  The steady-state
// loop is branch free (template flags for saving / dropout, row-padded [M] -> [roundup32(M)] buffers instead of row
// predicates): a branch inside it makes hipcc drain the load queue at every join.
#include <algorithm>

#include "common.h"

namespace tavsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef FFN_WPS
#define FFN_WPS 2        // workgroups per CU of the d_model = 256 kernels (waves per SIMD); 3 (48 KB LDS, <= 168 registers) measured slower
#endif

struct FfnArgs {
  int M, N1, S, nsub;              // rows, hidden units, hidden splits, sub-chunks (128 hidden units) per workgroup
  float* slab;                     // [S][M][D]
  const float *W1, *W2;            // first-product weight [N1][D], second-product weight [D][N1], both k-contiguous
  float *Z, *Hs;                   // [roundup32(M)][N1]: forward stores z, h (SAVE); backward reads Z
  uint32_t thr;                    // inner dropout (0: off): element (m, c) = word m & 3 of counter offset4 + (m >> 2) * N1 + c
  float inv_keep;
  const uint64_t* seed;
  uint64_t offset4;
  int act;
  // forward
  const float* x;                  // [M][ldx]
  int64_t ldx;
  const float *ln_w, *ln_b, *b1;
  float eps;
  float *n_out, *mean, *rstd;      // saved LayerNorm output / statistics (null: not kept)
  // backward
  const float* dy;                 // [M][lddy]
  int64_t lddy;
  float alpha;
  float* DZ;                       // [roundup32(M)][N1]
};

__device__ __forceinline__ int rho16(int r) { return (r & 3) + 8 * (r >> 2); }
// physical 16-byte chunk of logical chunk c in row r (XOR inside each group of 16 chunks = one 256-byte bank row)
__device__ __forceinline__ int swz(int c, int r) { return (c & ~15) | ((c ^ r) & 15); }

template <int D, bool BWD, bool SAVE, bool DROP, int ACT>
__global__ __launch_bounds__(256, D == 256 ? FFN_WPS : 1) void ffn_chain_kernel(const FfnArgs a) {
  constexpr int NT2 = D / 128;                     // 32-column tiles of the second product per wave
  constexpr int HB = (D == 256 && FFN_WPS == 3) ? 1 : 2;      // LDS: 32*D*4 + HB*16 KB per workgroup (48 KB: three per CU)
  __shared__ __attribute__((aligned(16))) float As[32 * D];
  __shared__ __attribute__((aligned(16))) float Hsm[HB][32 * 128];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  const int s = blockIdx.x % a.S, rt = blockIdx.x / a.S;       // neighbouring blocks (same XCD = id % 8) share a hidden slice
  const int m0 = rt * 32;

  // ---- prologue: the A tile of the first product -> LDS (LayerNorm(x) forward, alpha * dy backward)
  {
    constexpr int NV = D / 256;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = w * 8 + q, m = min(m0 + row, a.M - 1);
      float4 v[NV];
      if (!BWD) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[i] = *reinterpret_cast<const float4*>(a.x + (int64_t)m * a.ldx + (i * 64 + lane) * 4);
          sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = wave_sum(sum) / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float c0 = v[i].x - mu, c1 = v[i].y - mu, c2 = v[i].z - mu, c3 = v[i].w - mu;
          qq += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
        }
        const float rs = rsqrtf(wave_sum(qq) / (float)D + a.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c = (i * 64 + lane) * 4;
          const float4 g = *reinterpret_cast<const float4*>(a.ln_w + c), bb = *reinterpret_cast<const float4*>(a.ln_b + c);
          v[i] = make_float4((v[i].x - mu) * rs * g.x + bb.x, (v[i].y - mu) * rs * g.y + bb.y, (v[i].z - mu) * rs * g.z + bb.z,
                             (v[i].w - mu) * rs * g.w + bb.w);
          if (s == 0 && a.n_out && m0 + row < a.M) *reinterpret_cast<float4*>(a.n_out + (int64_t)m * D + c) = v[i];
        }
        if (s == 0 && lane == 0 && a.mean && m0 + row < a.M) {
          a.mean[m] = mu;
          a.rstd[m] = rs;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[i] = *reinterpret_cast<const float4*>(a.dy + (int64_t)m * a.lddy + (i * 64 + lane) * 4);
          v[i] = make_float4(v[i].x * a.alpha, v[i].y * a.alpha, v[i].z * a.alpha, v[i].w * a.alpha);
        }
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(As + row * D + swz(i * 64 + lane, row) * 4) = v[i];
    }
  }
  __syncthreads();

  f32x16 acc2[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

  // B fragments of one 32-deep k group for this lane (4 x 16-byte loads feed 16 MFMAs).  Every address = a wave-uniform
  // pointer (scalar registers) + ONE per-lane 32-bit offset that is the same for all loads of a product.
  const int lofs1 = li * D + 4 * h2;
  const int lofs2 = li * a.N1 + 4 * h2;
  auto load_b1 = [&](int c0, int kg, float (&bf)[16]) {        // first product: column (hidden unit) c0 + w*32 + li
    const float* ub = a.W1 + (int64_t)(c0 + w * 32) * D + kg * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(ub + 8 * g + lofs1);
      bf[4 * g] = t.x; bf[4 * g + 1] = t.y; bf[4 * g + 2] = t.z; bf[4 * g + 3] = t.w;
    }
  };
  auto load_b2 = [&](int c0, int t, int kg, float (&bf)[16]) { // second product: output column n = (w*NT2 + t)*32 + li
    const float* ub = a.W2 + (int64_t)((w * NT2 + t) * 32) * a.N1 + c0 + kg * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 tt = *reinterpret_cast<const float4*>(ub + 8 * g + lofs2);
      bf[4 * g] = tt.x; bf[4 * g + 1] = tt.y; bf[4 * g + 2] = tt.z; bf[4 * g + 3] = tt.w;
    }
  };
  auto mma16 = [&](const float* __restrict__ tile, int ld, int kg, const float (&bf)[16], f32x16& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 af = *reinterpret_cast<const float4*>(tile + li * ld + swz(kg * 8 + 2 * g + h2, li) * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[4 * g], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[4 * g + 3], acc, 0, 0, 0);
    }
  };

  const uint64_t sd = DROP ? a.seed[0] : 0;
  constexpr int KG1 = D / 32;          // 32-deep groups of the first product (even)
  float bfa[16], bfb[16];
  load_b1((s * a.nsub) * 128, 0, bfa);
  for (int sc = 0; sc < a.nsub; ++sc) {
    const int c0 = (s * a.nsub + sc) * 128;
    float* Ht = Hsm[HB == 2 ? (sc & 1) : 0];
    // ---- first product: 32 rows x 32 hidden units per wave, K = D; group kg + 1 is in flight while group kg multiplies
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG1; ++kg) {
      if (kg & 1) {
        if (kg + 1 < KG1) load_b1(c0, kg + 1, bfa); else load_b2(c0, 0, 0, bfa);
        mma16(As, D, kg, bfb, acc1);
      } else {
        load_b1(c0, kg + 1, bfb);
        mma16(As, D, kg, bfa, acc1);
      }
    }
    // (KG1 is even: the first B group of the second product is now in bfa)
    // ---- epilogue on the accumulators: lane = hidden unit c, registers = rows (all rows stored: the buffers are padded)
    {
      const int c = c0 + w * 32 + li;
      const float bias = BWD ? 0.f : a.b1[c];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        uint32_t wv[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (DROP) {
          const uint64_t ctr = a.offset4 + (uint64_t)((m0 >> 2) + 2 * q4 + h2) * (uint64_t)a.N1 + (uint64_t)c;
          philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), wv);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e, row = rho16(r) + 4 * h2;
          const int64_t o = (int64_t)(m0 + row) * a.N1 + c;
          const float keep = (DROP && wv[e] < a.thr) ? 0.f : a.inv_keep;
          float v;
          if (!BWD) {
            const float z = acc1[r] + bias;
            v = act_fwd(ACT, z) * keep;
            if (SAVE) {
              a.Z[o] = z;
              a.Hs[o] = v;
            }
          } else {
            v = acc1[r] * keep * act_bwd(ACT, a.Z[o]);
            a.DZ[o] = v;
          }
          Ht[row * 128 + swz(w * 8 + (li >> 2), row) * 4 + (li & 3)] = v;
        }
      }
    }
    __syncthreads();     // the 32 x 128 tile is complete (the other buffer's readers passed the previous barrier)
    // ---- second product: acc2 += tile (32 x 128) * second weight slice; 64 output columns per wave, 4 groups per tile
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        // group (t, kg) is in bfa (kg even) / bfb (kg odd); fetch the next one: (t, kg + 1), (t + 1, 0) or - behind the last
        // group - the first group of the next sub-chunk's first product (clamped in bounds: no branch)
        if (kg & 1) {
          if (kg + 1 < 4) load_b2(c0, t, kg + 1, bfa);
          else if (t + 1 < NT2) load_b2(c0, t + 1, 0, bfa);
          else load_b1(min(c0 + 128, a.N1 - 128), 0, bfa);
          mma16(Ht, 128, kg, bfb, acc2[t]);
        } else {
          load_b2(c0, t, kg + 1, bfb);
          mma16(Ht, 128, kg, bfa, acc2[t]);
        }
      }
    }
    if (HB == 1) __syncthreads();     // single tile buffer: everyone has read it before the next epilogue overwrites it
  }
  // ---- partial output of this hidden slice -> slab s
  float* out = a.slab + (int64_t)s * a.M * D;
#pragma unroll
  for (int t = 0; t < NT2; ++t) {
    const int n = (w * NT2 + t) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + rho16(r) + 4 * h2;
      if (m < a.M) out[(int64_t)m * D + n] = acc2[t][r];
    }
  }
}

// y = res + scale * dropout(sum_s slab_s + bias)   (bias / res / dropout optional); float4 lanes
__global__ __launch_bounds__(256) void ffn_finish_kernel(const float* __restrict__ slab, int S, int64_t slab_stride,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         int64_t ldr, float* __restrict__ y, int64_t M, int D, float scale,
                                                         uint32_t thr, float inv_keep, const uint64_t* __restrict__ seed,
                                                         uint64_t offset4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // group of 4 elements
  const int D4 = D >> 2;
  if (i >= M * D4) return;
  const int64_t m = i / D4;
  const int c = (int)(i % D4) * 4;
  float4 v = *reinterpret_cast<const float4*>(slab + m * D + c);
  for (int s = 1; s < S; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(slab + s * slab_stride + m * D + c);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (thr) {
    const uint64_t sd = seed[0], ctr = offset4 + (uint64_t)i;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), r);
    v.x = r[0] >= thr ? v.x * inv_keep : 0.f;
    v.y = r[1] >= thr ? v.y * inv_keep : 0.f;
    v.z = r[2] >= thr ? v.z * inv_keep : 0.f;
    v.w = r[3] >= thr ? v.w * inv_keep : 0.f;
  }
  v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
  if (res) {
    const float4 x = *reinterpret_cast<const float4*>(res + m * ldr + c);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  *reinterpret_cast<float4*>(y + m * D + c) = v;
}

}  // namespace tavsr

using namespace tavsr;

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// hidden splits: about two workgroups per CU (512 in flight), a power of two that divides the 128-wide sub-chunks
static int ffn_splits(int M, int N1) {
  const int nrt = cdiv(M, 32), nsub = N1 / 128;
  int S = 1;
  while (S * 2 <= nsub && nsub % (S * 2) == 0 && nrt * S * 2 <= 256 * FFN_WPS + 64) S *= 2;
  return S;
}

extern "C" int64_t tavsr_ffn_ws(int32_t M, int32_t D, int32_t N1) {
  if (M <= 0 || N1 <= 0 || N1 % 128 != 0) return 0;
  return (int64_t)ffn_splits(M, N1) * M * D;
}

static int ffn_check(const char* who, int M, int D, int N1, int act, const void* w1, const void* w2, const float* ws) {
  TAVSR_REQUIRE(M > 0 && (D == 256 || D == 512) && N1 > 0 && N1 % 128 == 0, TAVSR_EUNSUPPORTED,
                "%s: d_model 256 or 512 and a hidden size that is a multiple of 128 (got %d, %d)", who, D, N1);
  TAVSR_REQUIRE(act == TAVSR_ACT_RELU || act == TAVSR_ACT_SWISH, TAVSR_EUNSUPPORTED, "%s: ReLU or Swish only", who);
  TAVSR_REQUIRE(w1 && w2 && ws && al16(w1) && al16(w2) && al16(ws), TAVSR_EINVAL, "%s: null / unaligned weights or workspace", who);
  return TAVSR_OK;
}

template <int D, bool BWD, int ACT>
static void ffn_launch_act(const FfnArgs& a, bool save, hipStream_t s) {
  dim3 grid(cdiv(a.M, 32) * a.S);
  const bool drop = a.thr != 0;
  if (save && drop) hipLaunchKernelGGL((ffn_chain_kernel<D, BWD, true, true, ACT>), grid, dim3(256), 0, s, a);
  else if (save) hipLaunchKernelGGL((ffn_chain_kernel<D, BWD, true, false, ACT>), grid, dim3(256), 0, s, a);
  else if (drop) hipLaunchKernelGGL((ffn_chain_kernel<D, BWD, false, true, ACT>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ffn_chain_kernel<D, BWD, false, false, ACT>), grid, dim3(256), 0, s, a);
}
template <int D, bool BWD>
static void ffn_launch(const FfnArgs& a, bool save, hipStream_t s) {
  if (a.act == TAVSR_ACT_RELU) ffn_launch_act<D, BWD, TAVSR_ACT_RELU>(a, save, s);
  else ffn_launch_act<D, BWD, TAVSR_ACT_SWISH>(a, save, s);
}

// z / h (forward) and dz (backward) are [roundup32(M)][N1] buffers: the kernel stores whole 32-row tiles.
extern "C" int tavsr_ffn_fwd(const float* x, int64_t ldx, const float* ln_w, const float* ln_b, float eps, const float* w1,
                             const float* b1, const float* w2, const float* b2, int32_t act, float scale, int32_t M, int32_t D,
                             int32_t N1, float p_drop, const uint64_t* seed_dev, uint64_t offset_in, uint64_t offset_out,
                             float* n_out, float* mean, float* rstd, float* z, float* h, float* y, float* ws,
                             tavsr_stream_t stream) {
  int rc = ffn_check("ffn_fwd", M, D, N1, act, w1, w2, ws);
  if (rc) return rc;
  TAVSR_REQUIRE(x && ln_w && ln_b && b1 && b2 && y && ldx % 4 == 0 && al16(x) && al16(y), TAVSR_EINVAL, "ffn_fwd: bad operands");
  TAVSR_REQUIRE((z == nullptr) == (h == nullptr) && (mean == nullptr) == (rstd == nullptr), TAVSR_EINVAL,
                "ffn_fwd: z / h and mean / rstd are saved together");
  TAVSR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_dev) && offset_in % 4 == 0 && offset_out % 4 == 0,
                TAVSR_EINVAL, "ffn_fwd: dropout needs p in [0, 1), a device seed and offsets %% 4 == 0");
  FfnArgs a{};
  a.M = M; a.N1 = N1; a.S = ffn_splits(M, N1); a.nsub = N1 / 128 / a.S;
  a.slab = ws; a.W1 = w1; a.W2 = w2; a.Z = z; a.Hs = h;
  a.thr = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  a.seed = seed_dev; a.offset4 = offset_in / 4; a.act = act;
  a.x = x; a.ldx = ldx; a.ln_w = ln_w; a.ln_b = ln_b; a.b1 = b1; a.eps = eps;
  a.n_out = n_out; a.mean = mean; a.rstd = rstd;
  hipStream_t s = (hipStream_t)stream;
  if (D == 256) ffn_launch<256, false>(a, z != nullptr, s);
  else ffn_launch<512, false>(a, z != nullptr, s);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ffn_finish_kernel, dim3(cdiv((int64_t)M * (D / 4), 256)), dim3(256), 0, s, ws, a.S, (int64_t)M * D, b2, x, ldx,
                     y, (int64_t)M, D, scale, a.thr, a.inv_keep, seed_dev, offset_out / 4);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// dz <- ((alpha * dy) W2) * mask / keep * act'(z)  (kept for the W1 weight gradient) and dn <- dz W1, with the weights given
// TRANSPOSED: w2t = W2^T [N1][D], w1t = W1^T [D][N1]
extern "C" int tavsr_ffn_bwd_dx(const float* dy, int64_t lddy, float alpha, const float* w1t, const float* w2t, const float* z,
                                int32_t act, int32_t M, int32_t D, int32_t N1, float p_drop, const uint64_t* seed_dev,
                                uint64_t offset_in, float* dz, float* dn, float* ws, tavsr_stream_t stream) {
  int rc = ffn_check("ffn_bwd_dx", M, D, N1, act, w1t, w2t, ws);
  if (rc) return rc;
  TAVSR_REQUIRE(dy && z && dz && dn && lddy % 4 == 0 && al16(dy) && al16(dn), TAVSR_EINVAL, "ffn_bwd_dx: bad operands");
  TAVSR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_dev) && offset_in % 4 == 0, TAVSR_EINVAL,
                "ffn_bwd_dx: dropout needs p in [0, 1), a device seed and an offset %% 4 == 0");
  FfnArgs a{};
  a.M = M; a.N1 = N1; a.S = ffn_splits(M, N1); a.nsub = N1 / 128 / a.S;
  a.slab = ws; a.W1 = w2t; a.W2 = w1t; a.Z = const_cast<float*>(z);
  a.thr = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  a.seed = seed_dev; a.offset4 = offset_in / 4; a.act = act;
  a.dy = dy; a.lddy = lddy; a.alpha = alpha; a.DZ = dz;
  hipStream_t s = (hipStream_t)stream;
  if (D == 256) ffn_launch<256, true>(a, true, s);
  else ffn_launch<512, true>(a, true, s);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ffn_finish_kernel, dim3(cdiv((int64_t)M * (D / 4), 256)), dim3(256), 0, s, ws, a.S, (int64_t)M * D,
                     (const float*)nullptr, (const float*)nullptr, (int64_t)0, dn, (int64_t)M, D, 1.f, 0u, 1.f,
                     (const uint64_t*)nullptr, (uint64_t)0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
