// LayerNorm forward/backward and column reductions (HBM-bound: one wave per row, 16-B loads,
// shuffle reductions, deterministic two-stage parameter-gradient sums - no atomics).
#include "common.h"

namespace tavsr {

constexpr int LN_MAXV = 4;  // float4 chunks per lane: D <= 4*64*4 = 1024

// ---- forward: y = (x-mean)*rstd*gamma + beta ; one wave per row -------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) v[i] = *reinterpret_cast<const float4*>(xr + c);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < D) {
      float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + dd * dd);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  if (!y) return;                   // statistics only
  float* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < D) {
      float4 g = *reinterpret_cast<const float4*>(gamma + c);
      float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      *reinterpret_cast<float4*>(yr + c) = o;
    }
  }
}

// ---- backward ---------------------------------------------------------------------------------
// dx = dx_add + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  per-block partial sums of
// dgamma = sum dy*xhat and dbeta = sum dy go to part[blk][2][D], summed by sum_partials_kernel.
// One wave per row, LN_RPW rows per wave with the loads of both rows issued together (the kernel is pure
// latency otherwise: 3168 rows x 1 KB); NV = D / 256 float4 chunks per lane.
constexpr int LN_RPW = 2;
// SLAB (NV == 1, D == 256): dy is not a tensor but the UNSUMMED partial slabs a streaming feed-forward backward left in its workspace
// (tavsr_ffn2_bwd_dx with dn == NULL): row m = the sum over j < lddy of dy[((m / ldaz) * lddy + j) * ldaz + m % ldaz][256], added in
// that order - exactly the sum the block's finishing launch forms, which this mode replaces (lddy = partials per row block, ldaz = rows
// per row block; az itself is not used).
template <int NV, bool SLAB = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ dx_add, int64_t ldadd,
                                                            float* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ part, int64_t ld_part, int M, int D,
                                                            float* __restrict__ dx_drop, uint32_t thr, float inv_keep,
                                                            const uint64_t* __restrict__ seed, uint64_t offset4,
                                                            const float* __restrict__ az, int64_t ldaz, int act) {
  // az != null: the normalised tensor was x = act(az) - dx is written as the gradient w.r.t. az (dx * act'(az)); cgMLP's gate
  // half: LayerNorm(gelu(z)[:, C:]), no activation-backward pass over that half afterwards
  // dx_drop != null: a second output [M][D] = dropout_backward(dx) under the tavsr_dropout mask at offset4 - the consumer of dx
  // is a residual block whose branch starts with that mask (x + s * dropout(f(x))): saves its stand-alone mask launch
  __shared__ float red[4][2][NV * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 ag[NV], ab[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (i * 64 + lane) * 4;
    g[i] = c < D ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row0 = (blockIdx.x * 4 + w) * LN_RPW; row0 < M; row0 += gridDim.x * 4 * LN_RPW) {
    float4 xv[LN_RPW][NV], dv[LN_RPW][NV], av[LN_RPW][NV];      // av: the dx_add rows, or (az given: never both) the pre-activations
    float mu[LN_RPW], rs[LN_RPW];
#pragma unroll
    for (int q = 0; q < LN_RPW; ++q) {
      const int row = min(row0 + q, M - 1);
      mu[q] = mean[row];
      rs[q] = rstd[row];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        xv[q][i] = dv[q][i] = av[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < D) {
          xv[q][i] = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + c);
          if constexpr (!SLAB) dv[q][i] = *reinterpret_cast<const float4*>(dy + (int64_t)row * lddy + c);
          if (dx_add) av[q][i] = *reinterpret_cast<const float4*>(dx_add + (int64_t)row * ldadd + c);
          else if (!SLAB && az) av[q][i] = *reinterpret_cast<const float4*>(az + (int64_t)row * ldaz + c);      // with the other loads, not behind the reductions
        }
      }
    }
    if constexpr (SLAB) {
      static_assert(!SLAB || NV == 1, "slab mode: D == 256");
      const int wpb = (int)lddy, rbr = (int)ldaz;
      const float* p[LN_RPW];
#pragma unroll
      for (int q = 0; q < LN_RPW; ++q) {
        const int row = min(row0 + q, M - 1);
        p[q] = dy + (((int64_t)(row / rbr) * wpb) * rbr + row % rbr) * 256 + lane * 4;
      }
      constexpr int NP = 6;      // partials of both rows in flight per round (the finishing launch's order: j ascending)
      for (int j0 = 0; j0 < wpb; j0 += NP) {
        float4 t[LN_RPW][NP];
#pragma unroll
        for (int q = 0; q < LN_RPW; ++q)
#pragma unroll
          for (int j = 0; j < NP; ++j) t[q][j] = *reinterpret_cast<const float4*>(p[q] + (int64_t)min(j0 + j, wpb - 1) * rbr * 256);
#pragma unroll
        for (int q = 0; q < LN_RPW; ++q)
#pragma unroll
          for (int j = 0; j < NP; ++j)
            if (j0 + j < wpb) { dv[q][0].x += t[q][j].x; dv[q][0].y += t[q][j].y; dv[q][0].z += t[q][j].z; dv[q][0].w += t[q][j].w; }
      }
    }
#pragma unroll
    for (int q = 0; q < LN_RPW; ++q) {
#pragma clang fp contract(off)      // the SLAB instantiation must round like the plain one: no per-instantiation choice of fused multiply-adds
      if (row0 + q >= M) break;
      float4 xh[NV], gd[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 X = xv[q][i], Dv = dv[q][i];
        xh[i] = make_float4((X.x - mu[q]) * rs[q], (X.y - mu[q]) * rs[q], (X.z - mu[q]) * rs[q], (X.w - mu[q]) * rs[q]);
        const int c = (i * 64 + lane) * 4;
        if (c >= D) xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i].x += Dv.x * xh[i].x; ag[i].y += Dv.y * xh[i].y; ag[i].z += Dv.z * xh[i].z; ag[i].w += Dv.w * xh[i].w;
        ab[i].x += Dv.x; ab[i].y += Dv.y; ab[i].z += Dv.z; ab[i].w += Dv.w;
        gd[i] = make_float4(Dv.x * g[i].x, Dv.y * g[i].y, Dv.z * g[i].z, Dv.w * g[i].w);
        s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
      }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
      float* or_ = dx + (int64_t)(row0 + q) * lddx;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
          float4 o;
          o.x = rs[q] * (gd[i].x - s1 - xh[i].x * s2);
          o.y = rs[q] * (gd[i].y - s1 - xh[i].y * s2);
          o.z = rs[q] * (gd[i].z - s1 - xh[i].z * s2);
          o.w = rs[q] * (gd[i].w - s1 - xh[i].w * s2);
          if (dx_add) {
            o.x += av[q][i].x; o.y += av[q][i].y; o.z += av[q][i].z; o.w += av[q][i].w;
          } else if (!SLAB && az) {
            const float4 zz = av[q][i];
            o.x *= act_bwd(act, zz.x); o.y *= act_bwd(act, zz.y); o.z *= act_bwd(act, zz.z); o.w *= act_bwd(act, zz.w);
          }
          *reinterpret_cast<float4*>(or_ + c) = o;
          if (dx_drop) {
            const uint64_t sd = seed[0], ctr = offset4 + (((uint64_t)(row0 + q) * (uint64_t)D + (uint64_t)c) >> 2);
            uint32_t w4[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w4);
            *reinterpret_cast<float4*>(dx_drop + (int64_t)(row0 + q) * D + c) =
                make_float4(w4[0] >= thr ? o.x * inv_keep : 0.f, w4[1] >= thr ? o.y * inv_keep : 0.f,
                            w4[2] >= thr ? o.z * inv_keep : 0.f, w4[3] >= thr ? o.w * inv_keep : 0.f);
          }
        }
      }
    }
  }
  // cross-wave reduce of the column partials, fixed order (deterministic)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = (i * 64 + lane) * 4;
    *reinterpret_cast<float4*>(&red[w][0][c]) = ag[i];
    *reinterpret_cast<float4*>(&red[w][1][c]) = ab[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    int which = c / D, col = c % D;
    float s = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
    part[(int64_t)blockIdx.x * ld_part + (int64_t)which * D + col] = s;
  }
}

// out[c] (+)= scale * sum_p part[p*stride + c].  One block per 64 columns; the 4 waves take the partials
// p = w, w+4, ... (four independent load streams per lane), then a fixed-order cross-wave sum: deterministic.
// Columns c >= n1 go to out2[c - n1] (LayerNorm: dgamma and dbeta from one [blk][2][D] slab in one launch).
constexpr int kSumWaves = 16;      // waves per block: the partial rows are dealt to them (a 1024-row slab is 64 rows per wave)
__global__ __launch_bounds__(kSumWaves * 64) void sum_partials_kernel(const float* __restrict__ part, int nparts, int64_t stride,
                                                                      float* __restrict__ out, float* __restrict__ out2, int n1,
                                                                      int n, int accumulate, float scale) {
  __shared__ float red[kSumWaves][64];
  const int cx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    const float* p = part + c;
    int q = w;
    for (; q + 3 * kSumWaves < nparts; q += 4 * kSumWaves) {
      s0 += p[(int64_t)q * stride];
      s1 += p[(int64_t)(q + kSumWaves) * stride];
      s2 += p[(int64_t)(q + 2 * kSumWaves) * stride];
      s3 += p[(int64_t)(q + 3 * kSumWaves) * stride];
    }
    for (; q < nparts; q += kSumWaves) s0 += p[(int64_t)q * stride];
  }
  red[w][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kSumWaves; ++j) s += red[j][cx];      // fixed order: deterministic
    s *= scale;
    float* o = c < n1 ? out + c : out2 + (c - n1);
    *o = accumulate ? *o + s : s;
  }
}

// ---- column sums of a [M,N] matrix (bias gradients, pos_bias_u/v gradients) -------------------
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, int64_t ldx, int M, int N,
                                                          int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(M, r0 + rows_per_chunk);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + ry; r < r1; r += 4) s += x[(int64_t)r * ldx + col];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// out = x + y with the column sums of x and of y on the way (dQ = dQu + dQv and the pos_bias_u / pos_bias_v gradients of
// the rel-pos attention in one pass): part[chunk][0][N] = partial column sums of x, part[chunk][1][N] of y
__global__ __launch_bounds__(256) void add2_colsum_part_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ y, int64_t ldy,
                                                               float* __restrict__ out, int64_t ldo, int M, int N,
                                                               int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][2][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(M, r0 + rows_per_chunk);
  float sx = 0.f, sy = 0.f;
  if (col < N)
    for (int r = r0 + ry; r < r1; r += 4) {
      const float a = x[(int64_t)r * ldx + col], b = y[(int64_t)r * ldy + col];
      out[(int64_t)r * ldo + col] = a + b;
      sx += a;
      sy += b;
    }
  red[ry][0][cx] = sx;
  red[ry][1][cx] = sy;
  __syncthreads();
  if (ry < 2 && col < N)
    part[((int64_t)blockIdx.y * 2 + ry) * N + col] = (red[0][ry][cx] + red[1][ry][cx]) + (red[2][ry][cx] + red[3][ry][cx]);
}

static inline int ln_blocks(int M) { return min(cdiv(M, 4 * LN_RPW), 1024); }   // one row pair per wave up to 8192 rows: every row of the
                                                                                  // encoder's shapes in flight at once (two passes measured 8 / 31 us at D = 256 / 1024)
static inline int colsum_chunks(int M) { return min(cdiv(M, 64), 64); }

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                   float* y, int64_t ldy, float* mean, float* rstd, int32_t M, int32_t D,
                                   tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && ((gamma && beta && y) || (!y && mean && rstd)), TAVSR_EINVAL,
                "layernorm_fwd: null pointer (y may be null only for a statistics-only call with mean and rstd)");
  TAVSR_REQUIRE(D > 0 && D % 4 == 0 && D <= LN_MAXV * 256, TAVSR_EUNSUPPORTED,
                "layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, LN_MAXV * 256);
  TAVSR_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                    ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0),
                TAVSR_EALIGN, "layernorm_fwd: rows must be 16-byte aligned");
  if (M <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                     eps, y, ldy, mean, rstd, M, D);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_layernorm_bwd_ws(int32_t M, int32_t D) { return (int64_t)ln_blocks(M) * 2 * D; }

// main pass only: dx and the per-block partials of (dgamma | dbeta) at ws[blk * ws_ld + (0..2D)); the caller reduces them
// (tavsr_sum_partials over `tavsr_layernorm_bwd_ws(M, D) / (2 D)` blocks) - lets several LayerNorms of one backward node
// share ONE reduction launch (ws_ld = total columns of the shared slab).
static int layernorm_bwd_partial_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd, float* dx,
                                      int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D, const float* az, int64_t ldaz,
                                      int32_t act, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dy && x && mean && rstd && gamma && dx && ws, TAVSR_EINVAL, "layernorm_bwd: null pointer");
  TAVSR_REQUIRE(!az || (ldaz % 4 == 0 && (uintptr_t)az % 16 == 0), TAVSR_EALIGN, "layernorm_bwd: pre-activation rows must be 16-byte aligned");
  TAVSR_REQUIRE(D > 0 && D % 4 == 0 && D <= LN_MAXV * 256, TAVSR_EUNSUPPORTED, "layernorm_bwd: unsupported D=%d", D);
  TAVSR_REQUIRE(ws_ld >= 2 * (int64_t)D, TAVSR_EINVAL, "layernorm_bwd: partial slab rows too short");
  TAVSR_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldadd % 4 == 0 && ((uintptr_t)x % 16 == 0) &&
                    ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0) && ((uintptr_t)dx_add % 16 == 0),
                TAVSR_EALIGN, "layernorm_bwd: rows must be 16-byte aligned");
  if (M <= 0) return TAVSR_OK;
  const int nb = ln_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  if (D <= 256)
    hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add,
                       ldadd, dx, lddx, ws, ws_ld, M, D, (float*)nullptr, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0, az, ldaz, (int)act);
  else if (D <= 512)
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add,
                       ldadd, dx, lddx, ws, ws_ld, M, D, (float*)nullptr, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0, az, ldaz, (int)act);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add,
                       ldadd, dx, lddx, ws, ws_ld, M, D, (float*)nullptr, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0, az, ldaz, (int)act);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_layernorm_bwd_partial(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                           const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd,
                                           float* dx, int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D,
                                           tavsr_stream_t stream) {
  return layernorm_bwd_partial_impl(dy, lddy, x, ldx, mean, rstd, gamma, dx_add, ldadd, dx, lddx, ws, ws_ld, M, D, nullptr, 0, 0, stream);
}

// tavsr_layernorm_bwd_partial that also writes dx_drop [M][D] = dx * mask / keep (the tavsr_dropout mask of a contiguous [M][D]
// tensor at `offset`): the masked gradient the next residual block's branch starts from
extern "C" int tavsr_layernorm_bwd_partial_drop(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                                const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd,
                                                float* dx, int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D,
                                                float* dx_drop, float p_drop, const uint64_t* seed_dev, uint64_t offset,
                                                tavsr_stream_t stream) {
  TAVSR_REQUIRE(dy && x && mean && rstd && gamma && dx && ws && dx_drop && seed_dev, TAVSR_EINVAL, "layernorm_bwd_drop: null pointer");
  TAVSR_REQUIRE(D > 0 && D % 4 == 0 && D <= LN_MAXV * 256, TAVSR_EUNSUPPORTED, "layernorm_bwd_drop: unsupported D=%d", D);
  TAVSR_REQUIRE(ws_ld >= 2 * (int64_t)D, TAVSR_EINVAL, "layernorm_bwd_drop: partial slab rows too short");
  TAVSR_REQUIRE(p_drop > 0.f && p_drop < 1.f && offset % 4 == 0, TAVSR_EINVAL, "layernorm_bwd_drop: p in (0, 1), offset %% 4 == 0");
  TAVSR_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldadd % 4 == 0 && ((uintptr_t)x % 16 == 0) &&
                    ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0) && ((uintptr_t)dx_add % 16 == 0) &&
                    ((uintptr_t)dx_drop % 16 == 0),
                TAVSR_EALIGN, "layernorm_bwd_drop: rows must be 16-byte aligned");
  if (M <= 0) return TAVSR_OK;
  const int nb = ln_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t thr = (uint32_t)((double)p_drop * 4294967296.0);
  const float ik = 1.f / (1.f - p_drop);
  if (D <= 256)
    hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add, ldadd, dx,
                       lddx, ws, ws_ld, M, D, dx_drop, thr, ik, seed_dev, offset / 4, (const float*)nullptr, (int64_t)0, 0);
  else if (D <= 512)
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add, ldadd, dx,
                       lddx, ws, ws_ld, M, D, dx_drop, thr, ik, seed_dev, offset / 4, (const float*)nullptr, (int64_t)0, 0);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nb), dim3(256), 0, s, dy, lddy, x, ldx, mean, rstd, gamma, dx_add, ldadd, dx,
                       lddx, ws, ws_ld, M, D, dx_drop, thr, ik, seed_dev, offset / 4, (const float*)nullptr, (int64_t)0, 0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// tavsr_layernorm_bwd_partial[_drop] on the unsummed partial slabs of tavsr_ffn2_bwd_dx(dn = NULL): the sum over the partials is formed
// where the row is read (same order as the finishing launch: same bits) - one launch less per feed-forward block of a backward pass
extern "C" int tavsr_layernorm_bwd_partial_slab(const float* slab, int32_t wpb, int32_t rb_rows, const float* x, int64_t ldx, const float* mean,
                                                const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd, float* dx,
                                                int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D, float* dx_drop, float p_drop,
                                                const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream) {
  TAVSR_REQUIRE(slab && x && mean && rstd && gamma && dx && ws, TAVSR_EINVAL, "layernorm_bwd_partial_slab: null pointer");
  TAVSR_REQUIRE(D == 256 && wpb >= 1 && rb_rows >= 1, TAVSR_EUNSUPPORTED, "layernorm_bwd_partial_slab: D = 256 only (got %d), wpb / rows per block >= 1", D);
  TAVSR_REQUIRE(ws_ld >= 2 * (int64_t)D, TAVSR_EINVAL, "layernorm_bwd_partial_slab: partial slab rows too short");
  TAVSR_REQUIRE(!dx_drop || (p_drop > 0.f && p_drop < 1.f && seed_dev && offset % 4 == 0 && lddx == D), TAVSR_EINVAL,
                "layernorm_bwd_partial_slab: the masked copy needs p in (0, 1), a device seed, an offset %% 4 == 0 and contiguous dx");
  TAVSR_REQUIRE(ldx % 4 == 0 && lddx % 4 == 0 && ldadd % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)slab % 16 == 0) &&
                    ((uintptr_t)dx % 16 == 0) && ((uintptr_t)dx_add % 16 == 0) && ((uintptr_t)dx_drop % 16 == 0),
                TAVSR_EALIGN, "layernorm_bwd_partial_slab: rows must be 16-byte aligned");
  if (M <= 0) return TAVSR_OK;
  const uint32_t thr = dx_drop ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  const float ik = dx_drop ? 1.f / (1.f - p_drop) : 1.f;
  hipLaunchKernelGGL((layernorm_bwd_kernel<1, true>), dim3(ln_blocks(M)), dim3(256), 0, (hipStream_t)stream, slab, (int64_t)wpb, x, ldx, mean, rstd,
                     gamma, dx_add, ldadd, dx, lddx, ws, ws_ld, M, D, dx_drop, thr, ik, seed_dev, offset / 4, (const float*)nullptr,
                     (int64_t)rb_rows, 0);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                   const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd,
                                   float* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                                   float* ws, int32_t M, int32_t D, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dgamma && dbeta, TAVSR_EINVAL, "layernorm_bwd: null pointer");
  int rc = tavsr_layernorm_bwd_partial(dy, lddy, x, ldx, mean, rstd, gamma, dx_add, ldadd, dx, lddx, ws, 2 * (int64_t)D, M, D,
                                       stream);
  if (rc || M <= 0) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(2 * D, 64)), dim3(kSumWaves * 64), 0, (hipStream_t)stream, ws, ln_blocks(M),
                     (int64_t)2 * D, dgamma, dbeta, D, 2 * D, accumulate, 1.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// tavsr_layernorm_bwd for x = act(z): dx is the gradient w.r.t. z (dx_ln * act'(z)) - the LayerNorm of cgMLP's gate half
// (espnet ConvolutionalSpatialGatingUnit.norm on gelu(channel_proj1(x))[..., C:]) without an activation-backward pass afterwards
extern "C" int tavsr_layernorm_bwd_act(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                       const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                                       int32_t accumulate, float* ws, int32_t M, int32_t D, const float* z, int64_t ldz, int32_t act,
                                       tavsr_stream_t stream) {
  TAVSR_REQUIRE(dgamma && dbeta && z, TAVSR_EINVAL, "layernorm_bwd_act: null pointer");
  int rc = layernorm_bwd_partial_impl(dy, lddy, x, ldx, mean, rstd, gamma, nullptr, 0, dx, lddx, ws, 2 * (int64_t)D, M, D, z, ldz, act,
                                      stream);
  if (rc || M <= 0) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(2 * D, 64)), dim3(kSumWaves * 64), 0, (hipStream_t)stream, ws, ln_blocks(M),
                     (int64_t)2 * D, dgamma, dbeta, D, 2 * D, accumulate, 1.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int64_t tavsr_colsum_ws(int32_t M, int32_t N) { return (int64_t)colsum_chunks(M) * N; }

extern "C" int tavsr_colsum(const float* x, int64_t ldx, int32_t M, int32_t N, float scale, float* out,
                            int32_t accumulate, float* ws, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && out && ws, TAVSR_EINVAL, "colsum: null pointer");
  if (N <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = M > 0 ? colsum_chunks(M) : 1;
  const int rpc = M > 0 ? cdiv(M, chunks) : 1;
  hipLaunchKernelGGL(colsum_part_kernel, dim3(cdiv(N, 64), chunks), dim3(256), 0, s, x, ldx, M, N, rpc, ws);
  TAVSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(N, 64)), dim3(kSumWaves * 64), 0, s, ws, chunks, (int64_t)N, out, out, N, N,
                     accumulate, scale);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_add2_colsum(const float* x, int64_t ldx, const float* y, int64_t ldy, float* out, int64_t ldo, int32_t M,
                                 int32_t N, float* sum_x, float* sum_y, float* ws, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y && out && ws && ((sum_x != nullptr) == (sum_y != nullptr)), TAVSR_EINVAL, "add2_colsum: null pointer");
  if (N <= 0 || M <= 0) return TAVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = colsum_chunks(M);
  const int rpc = cdiv(M, chunks);
  hipLaunchKernelGGL(add2_colsum_part_kernel, dim3(cdiv(N, 64), chunks), dim3(256), 0, s, x, ldx, y, ldy, out, ldo, M, N, rpc, ws);
  TAVSR_LAUNCH_CHECK();
  // sum_x == sum_y == NULL: the column sums stay as tavsr_colsum_ws(M, N) / N partial rows of 2 N in ws; the caller reduces them when it
  // likes (tavsr_sum_partials2(ws, chunks, 2 N, sum_x, N, sum_y, N, 0, stream)) - the sums are bias gradients nobody reads inside a backward pass
  if (!sum_x) return TAVSR_OK;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(2 * N, 64)), dim3(kSumWaves * 64), 0, s, ws, chunks, (int64_t)2 * N, sum_x, sum_y, N,
                     2 * N, 0, 1.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// ... a slab of n1 + n2 columns summed into two outputs by one launch (BatchNorm backward: dbeta | dgamma)
extern "C" int tavsr_sum_partials2(const float* part, int32_t nparts, int64_t stride, float* out1, int32_t n1, float* out2,
                                   int32_t n2, int32_t accumulate, tavsr_stream_t stream) {
  TAVSR_REQUIRE(part && out1 && out2 && n1 >= 0 && n2 >= 0, TAVSR_EINVAL, "sum_partials2: null pointer");
  if (n1 + n2 <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(n1 + n2, 64)), dim3(kSumWaves * 64), 0, (hipStream_t)stream, part, nparts, stride,
                     out1, out2, n1, n1 + n2, accumulate, 1.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// Generic "sum nparts slabs of n floats" (used by the host for per-block parameter-gradient partials).
extern "C" int tavsr_sum_partials(const float* part, int32_t nparts, int64_t stride, float* out, int32_t n,
                                  int32_t accumulate, tavsr_stream_t stream) {
  TAVSR_REQUIRE(part && out, TAVSR_EINVAL, "sum_partials: null pointer");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(n, 64)), dim3(kSumWaves * 64), 0, (hipStream_t)stream, part, nparts, stride,
                     out, out, n, n, accumulate, 1.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
