// The tail of a Branchformer layer behind the branch join as ONE launch (tavsr_merge_proj_fwd, include/tavsr.h):
//     learned_ave merge of the two branch outputs  (src/encoder/branchformer/encoder_layer.py:232-290)
//     x = x + coeff * dropout(merge_proj(x_merged)) (encoder_layer.py:291-300)
// Before: row dots (5.6 us) -> softmaxes + weighted sum per 16-row block (6.1 us) -> a 3168 x 256 x 256 GEMM launch at 0.15 of
// the fp32 MFMA peak (17-29 us), three dependent launches at the one point of the layer where nothing else can run.
//
// One workgroup = 16 consecutive frames of one utterance (8 ceil(T / 16) ceil(B / 8) workgroups, 224 at B = 32, T = 99;
// the workgroups of an utterance share an XCD):
//   phase 0  the four dot products per frame of the WHOLE utterance (<wp_k, x_k[t]>, <ww_k, x_k[t]>; 16 lanes per row, four
//            rows per wave instruction, every load of a batch in flight before the first reduction) - every workgroup of an
//            utterance recomputes them (the rows come from L2: 200 KB per utterance) instead of waiting for a launch that does;
//   phase 1  the two softmaxes over time and the softmax over the branches from the dots (as merge_rows_fwd_kernel);
//   phase 2  the block's 16 mixed rows w_1 x_1 + w_2 x_2 -> LDS (and HBM when the backward pass wants them);
//   phase 3  [16 x 256] x merge_proj.weight^T on v_mfma_f32_16x16x4_f32: wave w owns output columns 64 w .. 64 w + 63, the mixed
//            rows are its A operand (ds_read_b128 along k), the weight rows its B operand straight from global memory / L2
//            (16-byte loads along k; k is permuted identically on both sides: MFMA (g, j) contracts k = 16 g + 4 q + j, q < 4);
//   phase 4  accumulators -> LDS image -> bias, dropout mask of the GEMM epilogue (same Philox counters: the backward pass
//            regenerates it from the token), coeff, residual, 16-byte stores.
#include <float.h>

#include "common.h"

namespace tavsr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MP_R = 16;       // frames per workgroup
constexpr int MP_LD = 260;     // LDS row stride (floats) of the [16][256] images: 16-byte groups of a b128 read hit distinct banks
constexpr int MP_D = 256;

struct MpArgs {
  const float *x1, *x2;
  const int64_t *lens, *lens2;
  const float *wp[2], *bp[2], *ww[2], *bw[2];
  const float *w, *bias, *res;
  float alpha;
  uint32_t thr;
  float inv_keep;
  const uint64_t* seed;
  uint64_t offset4;
  float *dots, *score, *wout, *mix, *out;
  int B, T;
  const float *rd1, *rd2;      // [B T][4][2]: per row and 64-column tile (<wp_k, x_k>, <ww_k, x_k>) from the producers' GEMM epilogues (or null)
};

__device__ __forceinline__ float dot4f(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float sum16(float v) {      // over the 16 lanes that share a row
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void merge_proj_fwd_kernel(const MpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T;
  float* s_s = sm;                         // [2][T] pooling scores, then the softmax over time
  float* s_q = sm + 2 * T;                 // [2][T] <ww_k, x_k[t]>
  float* s_m = sm + ((4 * T + 3) & ~3);    // [16][MP_LD] mixed rows (A operand)
  float* s_o = s_m + MP_R * MP_LD;         // [16][MP_LD] product image
  __shared__ float s_w[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  // consecutive workgroup ids go to consecutive XCDs: the ceil(T / 16) workgroups of an utterance - which all read its rows in
  // phase 0 - are dealt to ONE XCD (utterance b lives on XCD b % 8), so the rows cross the fabric once and come from that L2 after
  const int nch = (T + MP_R - 1) / MP_R;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = xcd + 8 * (slot / nch), ch = slot % nch;
  if (b >= a.B) return;
  const int64_t M = (int64_t)a.B * T, r0 = (int64_t)b * T;
  const int row = tid >> 4, t_row = ch * MP_R + row;
  const bool row_ok = t_row < T;
  const int64_t orow = (r0 + min(t_row, T - 1)) * MP_D + 4 * l16;

  // ---- everything that depends on nothing is in flight before the first wait: the first quarter of this wave's weight rows
  //      (B operand of phase 3), the thread's share of the block's own rows (phase 2) and of the residual rows (phase 4)
  const float* wbase = a.w + (int64_t)(64 * wv + l16) * MP_D + 4 * q;
  float4 bfr[2][4][4];                     // [buffer][n tile][g within the quarter]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) bfr[0][nt][g] = *reinterpret_cast<const float4*>(wbase + (int64_t)nt * 16 * MP_D + 16 * g);
  float4 own1[4], own2[4], resv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    own1[j] = *reinterpret_cast<const float4*>(a.x1 + orow + 64 * j);
    own2[j] = *reinterpret_cast<const float4*>(a.x2 + orow + 64 * j);
    resv[j] = a.res ? *reinterpret_cast<const float4*>(a.res + orow + 64 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- phase 0: dots of every frame of the utterance
  if (a.rd1) {
    // the branches' last Linear launches left them as four partial sums per row (tavsr_gemm_desc.rowdot_*): 64 bytes per row and
    // branch instead of the 2 KB of the rows themselves (every workgroup of an utterance used to re-read all of them from L2)
    for (int t = tid; t < T; t += 256) {
      const float4* p1 = reinterpret_cast<const float4*>(a.rd1 + (r0 + t) * 8);
      const float4* p2 = reinterpret_cast<const float4*>(a.rd2 + (r0 + t) * 8);
      const float4 u0 = p1[0], u1 = p1[1], v0 = p2[0], v1 = p2[1];         // (tile 0: pool, weight | tile 1: pool, weight), (tiles 2, 3)
      s_s[t] = (u0.x + u0.z) + (u1.x + u1.z);
      s_q[t] = (u0.y + u0.w) + (u1.y + u1.w);
      s_s[T + t] = (v0.x + v0.z) + (v1.x + v1.z);
      s_q[T + t] = (v0.y + v0.w) + (v1.y + v1.w);
    }
  } else {

    float4 wp1[4], wp2[4], ww1[4], ww2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * l16 + 64 * j;
      wp1[j] = *reinterpret_cast<const float4*>(a.wp[0] + c); wp2[j] = *reinterpret_cast<const float4*>(a.wp[1] + c);
      ww1[j] = *reinterpret_cast<const float4*>(a.ww[0] + c); ww2[j] = *reinterpret_cast<const float4*>(a.ww[1] + c);
    }
    const int nq = (T + 3) >> 2;                       // row quads; wave wv takes quads wv, wv + 4, ...
    constexpr int NB = 7;                              // quads per batch (8 NB 16-byte loads in flight per lane: T <= 112 is one batch)
    for (int q0 = wv; q0 < nq; q0 += 4 * NB) {
      float4 xa[NB][4], xc[NB][4];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int t = min(4 * (q0 + 4 * u) + q, T - 1);
        const float* p1 = a.x1 + (r0 + t) * MP_D + 4 * l16;
        const float* p2 = a.x2 + (r0 + t) * MP_D + 4 * l16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xa[u][j] = *reinterpret_cast<const float4*>(p1 + 64 * j);
          xc[u][j] = *reinterpret_cast<const float4*>(p2 + 64 * j);
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int t = 4 * (q0 + 4 * u) + q;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d0 += dot4f(xa[u][j], wp1[j]); d1 += dot4f(xc[u][j], wp2[j]);
          d2 += dot4f(xa[u][j], ww1[j]); d3 += dot4f(xc[u][j], ww2[j]);
        }
        d0 = sum16(d0); d1 = sum16(d1); d2 = sum16(d2); d3 = sum16(d3);
        if (l16 == 0 && t < T && q0 + 4 * u < nq) {
          s_s[t] = d0; s_s[T + t] = d1; s_q[t] = d2; s_q[T + t] = d3;
        }
      }
    }
  }
  __syncthreads();
  if (ch == 0)                       // what the backward pass reads back: dots[4][M]
    for (int i = tid; i < 4 * T; i += 256) {
      const int j = i / T, t = i - j * T;
      a.dots[(int64_t)j * M + r0 + t] = j < 2 ? s_s[j * T + t] : s_q[(j - 2) * T + t];
    }
  __syncthreads();
  // ---- phase 1: softmax over time per branch, softmax over the branches (arithmetic of merge_rows_fwd_kernel)
  if (wv < 2) {
    const int k = wv;
    const int64_t* lk = (k == 1 && a.lens2) ? a.lens2 : a.lens;
    const int len = lk ? (int)min((int64_t)T, lk[b]) : T;
    float* sc = s_s + k * T;
    const float bpk = a.bp[k][0], inv_sqrt_d = 1.f / 16.f;
    float mx = -FLT_MAX;
    for (int t = lane; t < len; t += 64) mx = fmaxf(mx, (sc[t] + bpk) * inv_sqrt_d);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < len; t += 64) sum += expf((sc[t] + bpk) * inv_sqrt_d - mx);
    sum = wave_sum(sum);
    const float inv = len > 0 ? 1.f / sum : 0.f;
    float lg = 0.f;
    for (int t = lane; t < T; t += 64) {      // (each lane rewrites only the entries it read)
      const float pv = t < len ? expf((sc[t] + bpk) * inv_sqrt_d - mx) * inv : 0.f;
      sc[t] = pv;
      lg += pv * s_q[k * T + t];
    }
    lg = wave_sum(lg);
    if (lane == 0) s_w[k] = lg + a.bw[k][0];
  }
  __syncthreads();
  const float mw = fmaxf(s_w[0], s_w[1]);
  const float e0 = expf(s_w[0] - mw), e1 = expf(s_w[1] - mw);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  if (ch == 0 && tid == 0) { a.wout[b * 2 + 0] = w0; a.wout[b * 2 + 1] = w1; }
  if (tid < 2 * MP_R) {
    const int k = tid / MP_R, t = ch * MP_R + tid % MP_R;
    if (t < T) a.score[((int64_t)k * a.B + b) * T + t] = s_s[k * T + t];
  }
  // ---- phase 2: the block's mixed rows (from the registers loaded at the top)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 m = make_float4(w0 * own1[j].x + w1 * own2[j].x, w0 * own1[j].y + w1 * own2[j].y, w0 * own1[j].z + w1 * own2[j].z,
                           w0 * own1[j].w + w1 * own2[j].w);
    if (!row_ok) m = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(s_m + row * MP_LD + 4 * l16 + 64 * j) = m;
    if (row_ok && a.mix) *reinterpret_cast<float4*>(a.mix + orow + 64 * j) = m;
  }
  __syncthreads();
  // ---- phase 3: [16 x 256] x W^T, wave wv -> columns 64 wv ..; the weight rows arrive a quarter of K ahead of their MFMAs
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int cur = qt & 1;
      if (qt + 1 < 4) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            bfr[cur ^ 1][nt][g] = *reinterpret_cast<const float4*>(wbase + (int64_t)nt * 16 * MP_D + 16 * (4 * (qt + 1) + g));
      }
      float4 af[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const float4*>(s_m + l16 * MP_LD + 16 * (4 * qt + g) + 4 * q);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float av[4] = {af[g].x, af[g].y, af[g].z, af[g].w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float4 bw = bfr[cur][nt][g];
            const float bv = jj == 0 ? bw.x : jj == 1 ? bw.y : jj == 2 ? bw.z : bw.w;
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bv, acc[nt], 0, 0, 0);
          }
      }
    }
  }
  // ---- phase 4: image, epilogue
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[(4 * q + r) * MP_LD + 64 * wv + 16 * nt + l16] = acc[nt][r];
  __syncthreads();
  if (row_ok) {
    const int64_t mrow = r0 + t_row;
    const uint64_t sd = a.thr ? a.seed[0] : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = 4 * l16 + 64 * j;
      float4 v = *reinterpret_cast<const float4*>(s_o + row * MP_LD + n);
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + n);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      if (a.thr) {
        const uint64_t ctr = a.offset4 + (uint64_t)((mrow * MP_D + n) >> 2);
        uint32_t wd[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), wd);
        v.x = wd[0] >= a.thr ? v.x * a.inv_keep : 0.f; v.y = wd[1] >= a.thr ? v.y * a.inv_keep : 0.f;
        v.z = wd[2] >= a.thr ? v.z * a.inv_keep : 0.f; v.w = wd[3] >= a.thr ? v.w * a.inv_keep : 0.f;
      }
      v.x = v.x * a.alpha + resv[j].x; v.y = v.y * a.alpha + resv[j].y; v.z = v.z * a.alpha + resv[j].z; v.w = v.w * a.alpha + resv[j].w;
      *reinterpret_cast<float4*>(a.out + mrow * MP_D + n) = v;
    }
  }
}

}  // namespace
}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_merge_proj_ok(int32_t T, int32_t D) { return D == 256 && T >= 1 && T <= 2048; }

extern "C" int tavsr_merge_proj_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                                    const float* const* params, const float* w, const float* bias, const float* res, float alpha,
                                    float p_drop, const uint64_t* seed, uint64_t drop_offset, float* dots, float* score, float* wout,
                                    float* mix, float* out, int32_t B, int32_t T, int32_t D, tavsr_stream_t stream) {
  return tavsr_merge_proj_fwd_dots(x1, x2, lens, lens2, params, w, bias, res, alpha, p_drop, seed, drop_offset, nullptr, nullptr, dots,
                                   score, wout, mix, out, B, T, D, stream);
}

extern "C" int tavsr_merge_proj_fwd_dots(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                                         const float* const* params, const float* w, const float* bias, const float* res, float alpha,
                                         float p_drop, const uint64_t* seed, uint64_t drop_offset, const float* rowdots1,
                                         const float* rowdots2, float* dots, float* score, float* wout, float* mix, float* out,
                                         int32_t B, int32_t T, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE((rowdots1 == nullptr) == (rowdots2 == nullptr), TAVSR_EINVAL, "merge_proj_fwd: the row dots of both branches or of neither");
  TAVSR_REQUIRE(((reinterpret_cast<uintptr_t>(rowdots1) | reinterpret_cast<uintptr_t>(rowdots2)) & 15) == 0, TAVSR_EALIGN,
                "merge_proj_fwd: row dots must be 16-byte aligned");
  TAVSR_REQUIRE(x1 && x2 && params && w && bias && dots && score && wout && out, TAVSR_EINVAL, "merge_proj_fwd: null pointer");
  for (int i = 0; i < 8; ++i) TAVSR_REQUIRE(params[i], TAVSR_EINVAL, "merge_proj_fwd: null parameter %d", i);
  TAVSR_REQUIRE(tavsr_merge_proj_ok(T, D), TAVSR_EUNSUPPORTED, "merge_proj_fwd: D == 256 and T <= 2048 required (T=%d D=%d)", T, D);
  TAVSR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), TAVSR_EINVAL, "merge_proj_fwd: p_drop in [0, 1), seed with dropout");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TAVSR_REQUIRE(al(x1) && al(x2) && al(w) && al(bias) && al(out) && (!res || al(res)) && (!mix || al(mix)) && al(params[0]) && al(params[1]) &&
                    al(params[4]) && al(params[5]) && (drop_offset & 3) == 0,
                TAVSR_EALIGN, "merge_proj_fwd: 16-byte aligned rows / weight vectors and a dropout offset %% 4 == 0 required");
  if (B <= 0) return TAVSR_OK;
  MpArgs a;
  a.x1 = x1; a.x2 = x2; a.lens = lens; a.lens2 = lens2;
  for (int k = 0; k < 2; ++k) { a.wp[k] = params[k]; a.bp[k] = params[2 + k]; a.ww[k] = params[4 + k]; a.bw[k] = params[6 + k]; }
  a.w = w; a.bias = bias; a.res = res; a.alpha = alpha;
  a.thr = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.inv_keep = 1.f / (1.f - p_drop);
  a.seed = seed; a.offset4 = drop_offset >> 2;
  a.dots = dots; a.score = score; a.wout = wout; a.mix = mix; a.out = out;
  a.B = B; a.T = T;
  a.rd1 = rowdots1; a.rd2 = rowdots2;
  const size_t lds = (size_t)(((4 * T + 3) & ~3) + 2 * MP_R * MP_LD) * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(merge_proj_fwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);     // T = 2048: 66 KB
  TAVSR_REQUIRE(attr == hipSuccess || lds <= 64 * 1024, TAVSR_EUNSUPPORTED, "merge_proj_fwd: %zu bytes of LDS not available", lds);
  hipLaunchKernelGGL(merge_proj_fwd_kernel, dim3(8 * cdiv(T, MP_R) * cdiv(B, 8)), dim3(256), lds, (hipStream_t)stream, a);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
