// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain per output).
//
// One kernel family serves every contraction on the hot path:
//   NT  C[M,N] = A[M,K] * W[N,K]^T          torch Linear forward (espnet Linear leaves)
//   NN  C[M,N] = A[M,K] * B[K,N]            data gradients, attention P*V
//   TN  C[M,N] = A[K,M]^T * B[K,N]          weight gradients, dK/dV
// with two batch dimensions given by element strides (attention heads are addressed in place inside
// the [B*T, 3*256] QKV buffer: no transposes are ever materialised) and a fused epilogue
// (bias, ReLU/Swish/GELU, pre-activation store, residual + alpha, multiply by act'(z) for backward,
// row sums of op(A) = bias gradients of the weight-gradient GEMMs).
//
// Tiling: BM x BN x BK block tile, WM x WN wavefronts (64 lanes), each wave owns TM x TN MFMA tiles
// of 32x32.  LDS images (double buffered):
//   k-contiguous operand  : [row][BK+4]   filled by ds_write_b128, fragments read by ds_read_b128:
//                           lane (r = lane&31, h = lane>>5) gets k = 8g+4h .. 8g+4h+3 of row r, i.e. the
//                           operands of FOUR consecutive MFMAs in one LDS instruction (conflict free:
//                           144-byte row stride puts the 16 lanes of a b128 group on 16 distinct slots)
//   row-contiguous operand: [k][rows+4]   filled by ds_write_b128, fragments read by ds_read_b32 at the
//                           same k = 8g+4h+j (the MFMA sums over k in any order as long as A and B agree)
// Pipeline per K-step (register staging, one barrier): tile t+1, fetched from global memory during the
// previous step, is written to the other LDS buffer at the TOP of step t, the global loads of tile t+2
// are issued right behind it, and only then the MFMAs of tile t run - so the loads have a whole step to
// land and the LDS writes overlap the first MFMAs.
//
// Few-tile / long-K problems are split over K (gridDim.z slices): every slice stores its raw fp32
// accumulators to a slab and a second, fully parallel kernel sums the slabs in slice order (deterministic)
// and runs the epilogue.  (An in-kernel "last arriver reduces" variant was measured and lost: the reduction
// of a 64x64 tile over 24 slices by ONE workgroup serialises ~400 KB of reads - profiles/r01_gemm_sweep_v2.txt.)
#include <algorithm>
#include <type_traits>

#include <cstdlib>
#include "common.h"

namespace tavsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  tavsr_gemm_desc d;
  int kchunk;     // K elements per split (multiple of BK)
  int nsplit;
  int tiles_m, tiles_n;
  int vec_epi;    // LDS-DMA kernels: epilogue through LDS with 16-byte row accesses (finish_tile_vec)
  int zmap;       // split-K convolution weight gradients: all tiles of a K slice on one XCD (see gemm_glds_kernel)
};

// Fixed-order sum of the split-K slabs + the fused epilogue (same math as the in-kernel one); one thread per
// 4 consecutive columns when N % 4 == 0 (16-byte slab reads), else per element.
template <int V>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs args) {
  const tavsr_gemm_desc& d = args.d;
  const int64_t mn = (int64_t)d.M * d.N;
  const int nbatch = d.nb1 * d.nb2;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  const int z = blockIdx.y, z1 = z / d.nb2, z2 = z % d.nb2;
  if (d.a_rowsum && z == 0 && i < d.M) {   // bias gradient: sum of the slices' row sums
    const float* rs = d.ws + (int64_t)args.nsplit * nbatch * mn;
    for (int q = 0; q < V && i + q < d.M; ++q) {
      float v = 0.f;
      for (int s = 0; s < args.nsplit; ++s) v += rs[(int64_t)s * d.M + i + q];
      d.a_rowsum[i + q] = d.alpha * v;
    }
  }
  if (i >= mn) return;
  const int m = (int)(i / d.N), n = (int)(i % d.N);
  float v[V];
#pragma unroll
  for (int q = 0; q < V; ++q) v[q] = 0.f;
  const float* p = d.ws + (int64_t)z * mn + i;
#pragma unroll 4
  for (int s = 0; s < args.nsplit; ++s) {
    if (V == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)s * nbatch * mn);
      v[0] += t.x; v[1 % V] += t.y; v[2 % V] += t.z; v[3 % V] += t.w;
    } else {
      v[0] += p[(int64_t)s * nbatch * mn];
    }
  }
  const int64_t o = z1 * d.sC1 + z2 * d.sC2 + (int64_t)m * d.ldc + n;
  const int64_t ro = z1 * d.sR1 + z2 * d.sR2 + (int64_t)m * d.ldr + n;
  uint32_t keep[4] = {1u, 1u, 1u, 1u};
  float inv_keep = 1.f;
  if (V == 4 && d.drop_p > 0.f) {        // the four columns of this thread are one Philox call (host: unbatched, N % 4 == 0)
    const uint64_t sd = d.drop_seed[0], ctr = (d.drop_offset >> 2) + (uint64_t)(i >> 2);
    const uint32_t thr = (uint32_t)((double)d.drop_p * 4294967296.0);
    uint32_t w[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
#pragma unroll
    for (int q = 0; q < 4; ++q) keep[q] = w[q] >= thr;
    inv_keep = 1.f / (1.f - d.drop_p);
  }
#pragma unroll
  for (int q = 0; q < V; ++q) {
    float x = v[q];
    if (d.bias) x += d.bias[n + q];
    if (d.Z) d.Z[o + q] = x;
    x = act_fwd(d.act, x);
    if (d.DZ) x *= act_bwd(d.dact, d.DZ[o + q]);
    x = keep[q % 4] ? x * inv_keep : 0.f;
    x *= d.alpha;
    if (d.R) x += d.R[ro + q];
    d.C[o + q] = x;
  }
}

template <int ROWS, int BK, bool KMAJOR>
struct Tile {
  static constexpr int LD = KMAJOR ? (ROWS + 4) : (BK + 4);
  static constexpr int SIZE = KMAJOR ? BK * LD : ROWS * LD;
};

// Global -> register -> LDS staging of one ROWS x BK operand tile (NV float4 per thread).
template <int ROWS, int BK, bool KMAJOR, bool VEC, int NT>
struct Loader {
  static constexpr int NV = ROWS * BK / 4 / NT;
  static_assert(ROWS * BK % (4 * NT) == 0, "tile must divide over the block");
  static constexpr int VPL = KMAJOR ? ROWS / 4 : BK / 4;   // float4 per contiguous line
  using T = Tile<ROWS, BK, KMAJOR>;

  // vector v covers 4 consecutive elements along the contiguous direction
  __device__ static __forceinline__ void coords(int v, int& row, int& k) {
    if (KMAJOR) {
      k = v / VPL;
      row = (v % VPL) * 4;
    } else {
      row = v / VPL;
      k = (v % VPL) * 4;
    }
  }
  // element offsets of this thread's vectors relative to (row0, k = 0); rows are clamped for the
  // k-contiguous layout (the epilogue never stores rows >= nrows, so what they hold is irrelevant)
  __device__ static __forceinline__ void offsets(int64_t ld, int row0, int nrows, int tid, int64_t (&off)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, k;
      coords(tid + i * NT, row, k);
      if (KMAJOR)
        off[i] = (int64_t)k * ld + row0 + row;
      else
        off[i] = (int64_t)min(row0 + row, nrows - 1) * ld + k;
    }
  }
  __device__ static __forceinline__ void load_fast(const float* __restrict__ g, const int64_t (&off)[NV],
                                                   float4 (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const float4*>(g + off[i]);
  }
  // fully predicated (edge tiles, K tails, unaligned operands): zero fill outside [nrows) x [K)
  __device__ static __forceinline__ void load_safe(const float* __restrict__ g, int64_t ld, int row0, int k0,
                                                   int nrows, int K, int tid, float4 (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, k;
      coords(tid + i * NT, row, k);
      int gr = row0 + row, gk = k0 + k;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KMAJOR) {
        if (gk < K) {
          const float* p = g + (int64_t)gk * ld + gr;
          if (VEC && gr + 3 < nrows) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            if (gr + 0 < nrows) val.x = p[0];
            if (gr + 1 < nrows) val.y = p[1];
            if (gr + 2 < nrows) val.z = p[2];
            if (gr + 3 < nrows) val.w = p[3];
          }
        }
      } else {
        if (gr < nrows) {
          const float* p = g + (int64_t)gr * ld + gk;
          if (VEC && gk + 3 < K) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk + 0 < K) val.x = p[0];
            if (gk + 1 < K) val.y = p[1];
            if (gk + 2 < K) val.z = p[2];
            if (gk + 3 < K) val.w = p[3];
          }
        }
      }
      r[i] = val;
    }
  }
  __device__ static __forceinline__ void store(float* __restrict__ s, int tid, const float4 (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, k;
      coords(tid + i * NT, row, k);
      *reinterpret_cast<float4*>(s + (KMAJOR ? k * T::LD + row : row * T::LD + k)) = r[i];
    }
  }
};

// Fragments of one 32-row sub-tile for k-group g (8 k values): f[j] is the operand of MFMA j, k = 8g+4h+j.
template <int ROWS, int BK, bool KMAJOR>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int row, int g, int lk, float (&f)[4]) {
  using T = Tile<ROWS, BK, KMAJOR>;
  if (KMAJOR) {
    const float* p = s + (g * 8 + 4 * lk) * T::LD + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = p[j * T::LD];
  } else {
    const float4 v = *reinterpret_cast<const float4*>(s + row * T::LD + g * 8 + 4 * lk);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
}

// Common tail of both kernels: lane pairs complete the row sums, then either the split-K slab store or the fused
// epilogue.  Accumulator layout: lane owns column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN>
__device__ __forceinline__ void finish_tile(const tavsr_gemm_desc& d, int nsplit, f32x16 (&acc)[TM][TN],
                                            float (&asum)[TM], bool want_rowsum, int m0, int n0, int wm, int wn, int lr,
                                            int lk, int z1, int z2, int64_t coff, int zidx, const float* bias_pre = nullptr) {
  if (want_rowsum) {
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] += __shfl_xor(asum[i], 32, 64);
  }

  // ---- split-K: every slice stores its raw accumulators (and row sums) to its slab; splitk_epilogue_kernel
  //      sums the slabs in slice order (deterministic) and applies the epilogue
  if (nsplit > 1) {
    const int64_t mn = (int64_t)d.M * d.N;
    const int nbatch = gridDim.y;
    float* slab = d.ws + ((int64_t)zidx * nbatch + blockIdx.y) * mn;
    float* rsum0 = d.ws + (int64_t)nsplit * nbatch * mn;       // [nsplit][M] (only when nbatch == 1)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + lr;
      if (n >= d.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < d.M) slab[(int64_t)m * d.N + n] = acc[i][j][r];
        }
      }
    }
    if (want_rowsum && lk == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 32 + i * 32 + lr;
        if (m < d.M) rsum0[(int64_t)zidx * d.M + m] = asum[i];
      }
    }
    return;
  }

  // ---- epilogue: lane owns column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (want_rowsum && lk == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * TM * 32 + i * 32 + lr;
      if (m < d.M) d.a_rowsum[m] = d.alpha * asum[i];
    }
  }
  float* C = d.C + coff;
  float* Z = d.Z ? d.Z + coff : nullptr;
  const float* R = d.R ? d.R + z1 * d.sR1 + z2 * d.sR2 : nullptr;
  const float* DZ = d.DZ ? d.DZ + coff : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + lr;
    if (n >= d.N) continue;
    const float bv = bias_pre ? bias_pre[j] : (d.bias ? d.bias[n] : 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lk;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m >= d.M) continue;
        float v = acc[i][j][r] + bv;
        const int64_t o = (int64_t)m * d.ldc + n;
        if (Z) Z[o] = v;
        v = act_fwd(d.act, v);
        if (DZ) v *= act_bwd(d.dact, DZ[o]);
        v *= d.alpha;
        if (R) v += R[(int64_t)m * d.ldr + n];
        C[o] = v;
      }
    }
  }
}

// MINW: waves per SIMD the register allocation must leave room for (blocks per CU x waves per block / 4):
// 4 blocks/CU for the 64x64 tile (its K-step is short: latency is hidden by co-resident blocks), 2 for the
// larger tiles (LDS admits two of them per CU).
template <int BM, int BN, int BK, int WM, int WN, int PF, int MINW, bool AK, bool BKM, bool VEC>
__global__ __launch_bounds__(WM* WN * 64, MINW)
void gemm_kernel(const GemmArgs args) {
  const tavsr_gemm_desc& d = args.d;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NG = BK / 8;
  using TA = Tile<BM, BK, AK>;
  using TB = Tile<BN, BK, BKM>;
  using LA = Loader<BM, BK, AK, VEC, NT>;
  using LB = Loader<BN, BK, BKM, VEC, NT>;
  constexpr int STAGE = TA::SIZE + TB::SIZE;  // stage s: A at smem + s*STAGE, B right after it
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lk = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (its L2), give them neighbouring tiles.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // n fastest: neighbouring tiles share the A panel
  const int m0 = (bid / args.tiles_n) * BM;
  const int n0 = (bid % args.tiles_n) * BN;
  const int z1 = blockIdx.y / d.nb2, z2 = blockIdx.y % d.nb2;

  const float* A = d.A + z1 * d.sA1 + z2 * d.sA2;
  const float* B = d.B + z1 * d.sB1 + z2 * d.sB2;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) asum[i] = 0.f;
  const bool want_rowsum = d.a_rowsum != nullptr && n0 == 0 && wn == 0;

  // split-K: slice z covers k in [kbeg, kend)
  const int kbeg = blockIdx.z * args.kchunk;
  const int kend = min(d.K, kbeg + args.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // a tile may use the unpredicated loads when it lies inside the operand (k-contiguous rows are clamped)
  const bool fastA = VEC && (AK ? (m0 + BM <= d.M) : true);
  const bool fastB = VEC && (BKM ? (n0 + BN <= d.N) : true);
  const int64_t kstepA = AK ? (int64_t)BK * d.lda : BK;
  const int64_t kstepB = BKM ? (int64_t)BK * d.ldb : BK;
  int64_t offA[LA::NV], offB[LB::NV];
  LA::offsets(d.lda, m0, d.M, tid, offA);
  LB::offsets(d.ldb, n0, d.N, tid, offB);
  const float* Ak = A + (AK ? (int64_t)kbeg * d.lda : kbeg);
  const float* Bk = B + (BKM ? (int64_t)kbeg * d.ldb : kbeg);

  // register ring: tile j is staged in set j % PF, PF K-steps before its MFMAs.
  float4 ra[PF][LA::NV], rb[PF][LB::NV];
  // A block whose tiles lie inside both operands and whose K range is whole K-steps runs a branch-free
  // steady-state loop (FAST): any branch inside the K loop splits it into basic blocks, and hipcc then
  // drains vmcnt/lgkmcnt at every join - the loads turn synchronous (measured: 2x on the whole kernel).
  const bool fast = fastA && fastB && (kend - kbeg) % BK == 0;
  auto load_tile = [&](int kt, float4 (&qa)[LA::NV], float4 (&qb)[LB::NV], auto fast_tag) {
    if constexpr (decltype(fast_tag)::value) {
      LA::load_fast(Ak + kt * kstepA, offA, qa);
      LB::load_fast(Bk + kt * kstepB, offB, qb);
    } else {
      const int k0 = kbeg + kt * BK;
      LA::load_safe(A, d.lda, m0, k0, d.M, kend, tid, qa);
      LB::load_safe(B, d.ldb, n0, k0, d.N, kend, tid, qb);
    }
  };
  const int arow = wm * TM * 32 + lr, brow = wn * TN * 32 + lr;
  // one K-step on LDS buffer (kt & 1); STORE: write ring set `rs` (tile kt+1) to the other buffer first,
  // LOAD: refill that set with tile kt+1+PF
  auto kstep = [&](int kt, float4 (&qa)[LA::NV], float4 (&qb)[LB::NV], bool do_store, bool do_load, auto fast_tag) {
    const int cur = kt & 1;
    const float* a_s = smem + cur * STAGE;
    const float* b_s = a_s + TA::SIZE;
    float af[2][TM][4], bf[2][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) read_frag<BM, BK, AK>(a_s, arow + i * 32, 0, lk, af[0][i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_frag<BN, BK, BKM>(b_s, brow + j * 32, 0, lk, bf[0][j]);
    if (do_store) {   // tile kt+1 -> the other buffer (its readers finished at the previous barrier)
      LA::store(smem + (cur ^ 1) * STAGE, tid, qa);
      LB::store(smem + (cur ^ 1) * STAGE + TA::SIZE, tid, qb);
      if (do_load) load_tile(kt + 1 + PF, qa, qb, fast_tag);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = g & 1;
      if (g + 1 < NG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag<BM, BK, AK>(a_s, arow + i * 32, g + 1, lk, af[c ^ 1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag<BN, BK, BKM>(b_s, brow + j * 32, g + 1, lk, bf[c ^ 1][j]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i][kk], bf[c][j][kk], acc[i][j], 0, 0, 0);
      // row sums of op(A) ride along unconditionally (TM adds per 16*TM*TN MFMA cycles; no branch in the loop)
#pragma unroll
      for (int i = 0; i < TM; ++i) asum[i] += (af[c][i][0] + af[c][i][1]) + (af[c][i][2] + af[c][i][3]);
    }
    __syncthreads();
  };
  auto run_loop = [&](auto fast_tag) {
    if (nk > 0) {
      load_tile(0, ra[0], rb[0], fast_tag);
      LA::store(smem, tid, ra[0]);
      LB::store(smem + TA::SIZE, tid, rb[0]);
#pragma unroll
      for (int j = 1; j <= PF; ++j)
        if (j < nk) load_tile(j, ra[j % PF], rb[j % PF], fast_tag);
    }
    __syncthreads();
    int kt0 = 0;
    // steady state: every step stores and refills, no conditions
    for (; kt0 + 2 * PF < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) kstep(kt0 + u, ra[(u + 1) % PF], rb[(u + 1) % PF], true, true, fast_tag);
    }
    // drain: the last (at most 2*PF) steps
    for (; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kt = kt0 + u;
        if (kt < nk) kstep(kt, ra[(u + 1) % PF], rb[(u + 1) % PF], kt + 1 < nk, kt + 1 + PF < nk, fast_tag);
      }
    }
  };
  if (fast) run_loop(std::true_type{});
  else run_loop(std::false_type{});

  finish_tile<TM, TN>(d, args.nsplit, acc, asum, want_rowsum, m0, n0, wm, wn, lr, lk, z1, z2, coff, (int)blockIdx.z);
}

// Epilogue through LDS (the staging ring is free once the K loop is over): the accumulators (lane = column, registers =
// rows) are written to a [BM][BN] image and read back row-wise, so that bias / pre-activation store / activation /
// act' / alpha / residual and the C (or split-K slab) store all move 16 bytes per lane along rows - a wave instruction
// covers whole 256-byte row segments instead of 2 x 128 bytes, and a 64x64 tile with a pre-activation output issues 8
// store instructions per lane instead of 32.  At K = 256 the old per-register epilogue was a third of a block's life
// (profiles/r01_gemm_trace.txt).  Needs N % 4 == 0 and 16-byte aligned rows of every output / epilogue operand.
template <int BM, int BN, int NT, int TM, int TN>
__device__ __forceinline__ void finish_tile_vec(const tavsr_gemm_desc& d, int nsplit, f32x16 (&acc)[TM][TN], float* __restrict__ img,
                                                int m0, int n0, int wm, int wn, int lr, int lk, int z1, int z2, int64_t coff,
                                                int tid, int zidx) {
  __syncthreads();                                   // every wave has finished reading the staging ring
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        img[(wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * BN + wn * TN * 32 + j * 32 + lr] = acc[i][j][r];
  __syncthreads();
  constexpr int V4R = BN / 4, PER = BM * BN / 4 / NT;
  static_assert(BM * BN % (4 * NT) == 0, "tile must divide over the block");
  const bool split = nsplit > 1;
  float* base;
  int64_t ld;
  if (split) {
    base = d.ws + ((int64_t)zidx * gridDim.y + blockIdx.y) * ((int64_t)d.M * d.N);
    ld = d.N;
  } else {
    base = d.C + coff;
    ld = d.ldc;
  }
  const float* R = (!split && d.R) ? d.R + z1 * d.sR1 + z2 * d.sR2 : nullptr;
  float* Z = (!split && d.Z) ? d.Z + coff : nullptr;
  const float* DZ = (!split && d.DZ) ? d.DZ + coff : nullptr;
  const bool rowstat = BN == 64 && !split && d.rowstat != nullptr;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int idx = q * NT + tid, row = idx / V4R, c4 = idx % V4R;
    const int m = m0 + row, n = n0 + 4 * c4;
    const bool ok = m < d.M && n < d.N;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
    v = *reinterpret_cast<const float4*>(img + row * BN + 4 * c4);
    const int64_t o = (int64_t)m * ld + n;
    if (!split) {
      if (d.bias) {
        const float4 b = *reinterpret_cast<const float4*>(d.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (Z) *reinterpret_cast<float4*>(Z + o) = v;
      v.x = act_fwd(d.act, v.x); v.y = act_fwd(d.act, v.y); v.z = act_fwd(d.act, v.z); v.w = act_fwd(d.act, v.w);
      if (DZ) {
        const float4 z = *reinterpret_cast<const float4*>(DZ + o);
        v.x *= act_bwd(d.dact, z.x); v.y *= act_bwd(d.dact, z.y); v.z *= act_bwd(d.dact, z.z); v.w *= act_bwd(d.dact, z.w);
      }
      if (d.drop_p > 0.f) {              // one Philox call per 16-byte group (the mask tavsr_dropout draws for [M][N])
        const uint64_t sd = d.drop_seed[0], ctr = (d.drop_offset >> 2) + (uint64_t)(((int64_t)m * d.N + n) >> 2);
        const uint32_t thr = (uint32_t)((double)d.drop_p * 4294967296.0);
        const float ik = 1.f / (1.f - d.drop_p);
        uint32_t w[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), w);
        v.x = w[0] >= thr ? v.x * ik : 0.f; v.y = w[1] >= thr ? v.y * ik : 0.f;
        v.z = w[2] >= thr ? v.z * ik : 0.f; v.w = w[3] >= thr ? v.w * ik : 0.f;
      }
      v.x *= d.alpha; v.y *= d.alpha; v.z *= d.alpha; v.w *= d.alpha;
      if (R) {
        const float4 rr = *reinterpret_cast<const float4*>(R + (int64_t)m * d.ldr + n);
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
    }
    *reinterpret_cast<float4*>(base + o) = v;
    }
    if (rowstat) {       // the 16 lanes that share a row of a 64-wide tile: sum and sum of squares of what was stored ...
      float s1 = ok ? (v.x + v.y) + (v.z + v.w) : 0.f, s2 = ok ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
      if (d.rowdot_a) {  // ... or its two weighted sums (tavsr_gemm_desc.rowdot_a / _b: the merge's pooling and branch-weight projections)
        const float4 wa = ok ? *reinterpret_cast<const float4*>(d.rowdot_a + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 wb = ok ? *reinterpret_cast<const float4*>(d.rowdot_b + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        s1 = (v.x * wa.x + v.y * wa.y) + (v.z * wa.z + v.w * wa.w);
        s2 = (v.x * wb.x + v.y * wb.y) + (v.z * wb.z + v.w * wb.w);
      }
#pragma unroll
      for (int o2 = 8; o2 > 0; o2 >>= 1) { s1 += __shfl_xor(s1, o2, 64); s2 += __shfl_xor(s2, o2, 64); }
      if (c4 == 0 && m < d.M)
        *reinterpret_cast<float2*>(d.rowstat + ((int64_t)m * ((d.N + 63) / 64) + n0 / 64) * 2) = make_float2(s1, s2);
    }
  }
}

// host-side test of the vectorised epilogue's requirements
static bool vec_epi_ok(const tavsr_gemm_desc& d) {
  constexpr int on = 1;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return on && d.N % 4 == 0 && d.ldc % 4 == 0 && d.sC1 % 4 == 0 && d.sC2 % 4 == 0 && al(d.C) && (!d.bias || al(d.bias)) &&
         (!d.Z || al(d.Z)) && (!d.DZ || al(d.DZ)) && (!d.R || (al(d.R) && d.ldr % 4 == 0 && d.sR1 % 4 == 0 && d.sR2 % 4 == 0)) &&
         (!d.ws || al(d.ws));
}

// ---------------------------------------------------------------------------------------------- LDS-DMA kernel
// Fast path for tiles whose operands can be fetched with unpredicated 16-byte loads (aligned, K a multiple of 32,
// row-contiguous operands with rows % 4 == 0).  Operand tiles go global -> LDS directly (global_load_lds_dwordx4:
// no staging registers, so the compiler cannot turn the prefetch into a synchronous load by copying its result
// registers - which is what it did to the register ring of gemm_kernel, profiles/r01_gemm_notes.md) through a ring
// of S LDS stages; a counted s_waitcnt vmcnt leaves S-2 tiles in flight across the ONE barrier per K-step.
// LDS images (a wave-instruction writes 1 KB linearly: no padding possible, conflicts are avoided by swizzling):
//   k-contiguous operand  : [row][32 floats]; 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7)
//                           (the SOURCE address is permuted, the LDS write stays linear; ds_read_b128 applies the
//                           same XOR: the 16 lanes of a b128 group hit 16 distinct 16-byte slots)
//   row-contiguous operand: [k][ROWS floats], read by ds_read_b32 over 32 consecutive rows
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;

template <int ROWS, bool KMAJOR, int NT>
struct GLoader {
  static constexpr int NR = ROWS * 8 / NT;   // LDS-DMA instructions per thread per tile
  static_assert(ROWS * 8 % NT == 0, "tile must divide over the block");
  __device__ static __forceinline__ void offsets(int64_t ld, int row0, int nrows, int tid, int64_t (&off)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int q = i * NT + tid;
      if (KMAJOR) {
        const int k = q / (ROWS / 4);
        int r = row0 + (q % (ROWS / 4)) * 4;
        if (r + 3 >= nrows) r = row0;            // rows past the edge are never stored: any valid address will do
        off[i] = (int64_t)k * ld + r;
      } else {
        const int row = q >> 3, cp = q & 7;
        const int cl = cp ^ ((row >> 1) & 7);
        off[i] = (int64_t)min(row0 + row, nrows - 1) * ld + cl * 4;
      }
    }
  }
  __device__ static __forceinline__ void issue(const float* __restrict__ g, const int64_t (&off)[NR],
                                               float* __restrict__ stage, int wave) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      __builtin_amdgcn_global_load_lds((glb_float*)(g + off[i]), (lds_float*)(stage + (i * NT + wave * 64) * 4), 16, 0, 0);
  }
};

#ifndef TAVSR_GEMM_SB
#define TAVSR_GEMM_SB 1
#endif
#if TAVSR_GEMM_SB
#define GEMM_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define GEMM_SB()
#endif

template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void read_frag_g(const float* __restrict__ s, int row, int g, int lk, float (&f)[4]) {
  if (KMAJOR) {
    const float* p = s + (g * 8 + 4 * lk) * ROWS + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = p[j * ROWS];
  } else {
    const float4 v = *reinterpret_cast<const float4*>(s + row * 32 + (((2 * g + lk) ^ ((row >> 1) & 7)) << 2));
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
}

// 16 readable zero floats for LDS-DMA loads that must deliver zeros (K tails)
__device__ float g_zero_page[16];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One output tile of one problem: `bid` is the (already XCD-remapped) linear tile index inside the problem.
#ifdef TAVSR_GEMM_TRACE
// Debug build only (scripts/gpu_trace.sh): per-workgroup timestamps of the LDS-DMA kernel's phases.
constexpr int kTraceMax = 1 << 15;
__device__ unsigned long long g_trace[kTraceMax][6];
__device__ unsigned int g_trace_n;
#define TAVSR_TRACE_DECL unsigned long long tr_t[4], tr_c[4]; tr_t[0] = wall_clock64(); tr_t[1] = tr_t[0]; tr_c[1] = tr_c[2] = 0;
#define TAVSR_TRACE_AT(i) { tr_t[i] = wall_clock64(); tr_c[i] = __builtin_readcyclecounter(); }
#else
#define TAVSR_TRACE_DECL
#define TAVSR_TRACE_AT(i)
#endif

// KW > 1: the k-groups of every K-step are dealt to KW wave sets (intra-block K split, summed through LDS at the end):
// a lone 64x64 tile on a CU then runs 2 waves per SIMD with half the dependent-MFMA chain per K-step each.
// CONV (implicit 3x3 / stride 1 / pad 1 convolution over a channels-last image, no im2col matrix in HBM):
//   1: the A operand is the image X [n*H*W][C]; A(m, k = tap*C + c) = X[m + (tap/3-1)*W + (tap%3-1)][c] inside the
//      image, 0 outside (forward and data gradient).  Per K-step the tap is uniform, so the row offsets of the plain
//      loader move by one scalar and out-of-image rows are pointed at a zero page.
//   2: the k-major B operand is the image (weight gradient dW = dY^T patches): B(k = m, n = tap*C + c); a 64-wide n
//      tile lies in one tap, validity is per k row.
// Conv3d stem (1 -> 64 channels, kernel (5,7,7), stride (1,2,2), padding (2,3,3); conv3d_resnet18.py:48-57) over clips
// x [clips][T = conv_C][H][W], single input channel, 245 taps padded to K / N = 256 - every operand element is its own
// 4-byte LDS-DMA gather (a patch row is 35 runs of 7 floats), so no patch matrix is ever written:
//   4: A(m = output pixel (clip, t, ho, wo), k = (kt*7 + kh)*7 + kw) = x[clip][t + kt - 2][2 ho - 3 + kh][2 wo - 3 + kw];
//   5: the k-major B operand (weight gradient): B(k = pixel, n = tap), as 4 with the roles of rows and columns swapped.
// The same stem over ZERO-PADDED clips xp [clips][T + 5][H + 6][W + 8] (2 / 3 frames, 3 / 3 rows, 3 / 5 columns of zeros
// around every clip, tavsr_stem_pad) with the taps laid out k = ((kt*7 + kh) * 8 + kw), K / N = 288 (kw = 7 and the last
// 8 columns carry zero weights): every tap is inside the buffer and four consecutive k are four consecutive floats, so the
// operand is fetched with the GEMM's ordinary 16-byte LDS-DMA (two per thread and K-step instead of eight 4-byte gathers; the
// source is only 8-byte aligned, which gfx950's global_load_lds takes) and no validity test is left:
//   6: A(m, k) = xp[clip][t + kt][2 ho + kh][2 wo + kw];   7: B(k = pixel, n = tap) likewise (weight gradient).
template <int BM, int BN, int WM, int WN, int S, bool AK, bool BKM, int KW = 1, int CONV = 0>
__device__ __forceinline__ void glds_tile(const tavsr_gemm_desc& d, int kchunk, int nsplit, int tiles_n, int bid, bool vec_epi,
                                          int zidx) {
  constexpr int BK = 32, NG = BK / 8;
  static_assert(NG % KW == 0, "k-groups must divide over the wave sets");
  constexpr int NT = WM * WN * KW * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using LA = GLoader<BM, AK, NT>;
  using LB = GLoader<BN, BKM, NT>;
  constexpr int ASZ = BM * BK, STAGE = (BM + BN) * BK;
  constexpr int NR4A = BM * BK / NT, NR4B = BN * BK / NT;      // 4-byte gathers per thread per tile (CONV 4 / 5)
  constexpr int G = (CONV == 4 ? NR4A : LA::NR) + (CONV == 5 ? NR4B : LB::NR);          // LDS-DMA instructions per wave per tile
  static_assert((S - 2) * G <= 63, "vmcnt field");
  __shared__ __attribute__((aligned(1024))) float smem[S * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kw = wave / (WM * WN), w2 = wave % (WM * WN);
  const int wm = w2 / WN, wn = w2 % WN;
  const int lr = lane & 31, lk = lane >> 5;

  TAVSR_TRACE_DECL
  const int m0 = (bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;
  const int z1 = blockIdx.y / d.nb2, z2 = blockIdx.y % d.nb2;
  const float* A = d.A + z1 * d.sA1 + z2 * d.sA2;
  const float* B = d.B + z1 * d.sB1 + z2 * d.sB2;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) asum[i] = 0.f;
  const bool want_rowsum = d.a_rowsum != nullptr && n0 == 0 && wn == 0;
  float bpre[TN];      // the epilogue's bias values, fetched under the K loop instead of in front of the stores
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + lr;
    bpre[j] = (d.bias && nsplit == 1 && n < d.N) ? d.bias[n] : 0.f;
  }

  const int kbeg = zidx * kchunk;
  const int kend = min(d.K, kbeg + kchunk);
  const int nk = CONV == 3 ? (kend - kbeg + BK - 1) / BK : (kend - kbeg) / BK;     // whole K-steps (host guarantees it); CONV 3: K tail
  const int64_t kstepA = AK ? (int64_t)BK * d.lda : BK;
  const int64_t kstepB = BKM ? (int64_t)BK * d.ldb : BK;
  int64_t offA[LA::NR], offB[LB::NR];
  LA::offsets(d.lda, m0, d.M, tid, offA);
  LB::offsets(d.ldb, n0, d.N, tid, offB);
  const float* Ak = A + (AK ? (int64_t)kbeg * d.lda : kbeg);
  const float* Bk = B + (BKM ? (int64_t)kbeg * d.ldb : kbeg);
  uint32_t cmask[LA::NR];               // CONV 1: bit tap = the tap's neighbour of this thread's row is inside the image
  const int cs = d.conv_stride > 1 ? d.conv_stride : 1;         // CONV: stride; 9 taps (3x3, pad 1) or 1 (1x1, pad 0)
  const bool c9 = d.conv_taps != 1;
  // conv_taps 90: the 3x3 window without padding (espnet Conv2dSubsampling's second convolution): output pixel (ho, wo) is
  // centred on input pixel (cs*ho + 1, cs*wo + 1) and every tap is inside the image
  const int cp0 = (CONV == 1 || CONV == 2) && d.conv_taps == 90 ? 1 : 0;
  const int cHo = CONV ? (d.conv_H - 1 - 2 * cp0) / cs + 1 : 1, cWo = CONV ? (d.conv_W - 1 - 2 * cp0) / cs + 1 : 1;
  if (CONV == 1) {
#pragma unroll
    for (int i = 0; i < LA::NR; ++i) {
      const int m = min(m0 + ((i * NT + tid) >> 3), d.M - 1);
      // row m = output pixel (n, ho, wo), centred on input pixel (cs*ho, cs*wo)
      const int x = (m % cWo) * cs + cp0, y = ((m / cWo) % cHo) * cs + cp0;
      if (cs > 1 || cp0) offA[i] += ((int64_t)((m / (cWo * cHo)) * d.conv_H + y) * d.conv_W + x - m) * d.conv_C;
      uint32_t mk = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        mk |= (uint32_t)((unsigned)(y + tap / 3 - 1) < (unsigned)d.conv_H && (unsigned)(x + tap % 3 - 1) < (unsigned)d.conv_W) << tap;
      cmask[i] = c9 ? mk : 1u;
    }
  }
  // CONV 4: gather i of this thread fills LDS float (i NT + tid) of the k-contiguous image: row 8 i + (tid >> 5), physical
  // column p = tid & 31, i.e. (XOR swizzle) k = 32 kt + kk with kk = (((p >> 2) ^ ((4 i + wave) & 7)) << 2) + (p & 3) - two
  // values per thread (i even / odd), so a K-step decodes two taps, not eight.  Per gather stay: the pixel's offset in x and
  // one validity mask (bit a: frame t + a - 2 exists; bit 8 + b: row 2 ho - 3 + b; bit 16 + c: column 2 wo - 3 + c).
  static_assert(CONV != 4 || NT == 256, "the two-taps-per-thread decoding assumes 8 rows per gather instruction");
  int s4_base[CONV == 4 ? NR4A : 1], s4_mask[CONV == 4 ? NR4A : 1];
  const int sT = d.conv_C, sH = d.conv_H, sW = d.conv_W, sHo = (sH - 1) / 2 + 1, sWo = (sW - 1) / 2 + 1;
  const int s4_p = tid & 31;
  const int s4_kk0 = ((((s4_p >> 2) ^ (wave & 7))) << 2) + (s4_p & 3), s4_kk1 = ((((s4_p >> 2) ^ ((wave + 4) & 7))) << 2) + (s4_p & 3);
  if (CONV == 4) {
#pragma unroll
    for (int i = 0; i < NR4A; ++i) {
      const int row = i * 8 + (tid >> 5);
      const int m = min(m0 + row, d.M - 1);
      const int wo = m % sWo, ho = (m / sWo) % sHo, ft = m / (sWo * sHo);          // ft = clip * T + t
      const int t = ft % sT, hy = 2 * ho - 3, wx = 2 * wo - 3;
      s4_base[i] = (ft * sH + hy) * sW + wx;
      int mk = 0;
#pragma unroll
      for (int q = 0; q < 5; ++q) mk |= (int)((unsigned)(t + q - 2) < (unsigned)sT) << q;
#pragma unroll
      for (int q = 0; q < 7; ++q)
        mk |= ((int)((unsigned)(hy + q) < (unsigned)sH) << (8 + q)) | ((int)((unsigned)(wx + q) < (unsigned)sW) << (16 + q));
      s4_mask[i] = mk;
    }
  }
  // CONV 5: gather (i, wave) of a tile is B(k = 32 kt + 4 i + wave, n = n0 + lane): the tap is fixed per lane, the pixel is
  // the same for the whole wave and walks on by 4 per gather (tiles are issued in k order): its coordinates and its offset
  // in x are carried (wave-uniform), not divided out.  Needs even H and W (host).
  static_assert(CONV != 5 || (BN == 64 && NT == 256), "one k row per wave instruction");
  int s5_wo = 0, s5_ho = 0, s5_t = 0, s5_off = 0, s5_tapoff = 0, s5_c3 = 0;
  bool s5_th = false;           // frame t + a - 2 and row 2 ho - 3 + b exist (changes only when the pixel changes row)
  int s5_a = 0, s5_b = 0;
  bool s5_nok = false;
  if (CONV == 5) {
    const int nn = n0 + lane;
    s5_a = nn / 49; s5_b = (nn % 49) / 7;
    const int c = nn % 7;
    s5_c3 = c - 3;
    s5_nok = nn < 245;
    s5_tapoff = ((s5_a - 2) * sH + s5_b - 3) * sW + c - 3;
    const int m = kbeg + wave;
    s5_wo = m % sWo; s5_ho = (m / sWo) % sHo;
    const int ft = m / (sWo * sHo);
    s5_t = ft % sT;
    s5_off = (ft * sH + 2 * s5_ho) * sW + 2 * s5_wo;
    s5_th = s5_nok && (unsigned)(s5_t + s5_a - 2) < (unsigned)sT && (unsigned)(2 * s5_ho - 3 + s5_b) < (unsigned)sH;
  }
  // CONV 6 / 7: padded clips, 16-byte chunks.  Hp x Wp padded frame, Tp padded frames per clip (conv_H, conv_W, conv_C).
  const int pHp = d.conv_H, pWp = d.conv_W, pTp = d.conv_C, pHo = (pHp - 6) / 2, pWo = (pWp - 8) / 2, pT = pTp - 5;
  int s6_base[CONV == 6 ? LA::NR : 1], s6_cl[CONV == 6 ? LA::NR : 1];
  if (CONV == 6) {
#pragma unroll
    for (int i = 0; i < LA::NR; ++i) {
      const int q = i * NT + tid, row = q >> 3;
      s6_cl[i] = (q & 7) ^ ((row >> 1) & 7);                 // logical 16-byte chunk of the K-step this DMA fetches
      const int m = min(m0 + row, d.M - 1);
      const int wo = m % pWo, ho = (m / pWo) % pHo, ft = m / (pWo * pHo);
      s6_base[i] = (((ft / pT) * pTp + ft % pT) * pHp + 2 * ho) * pWp + 2 * wo;
    }
  }
  // CONV 7: the chunk's tap is fixed per thread, its pixel walks on by NT / (BN / 4) per gather (tiles are issued in k order)
  int s7_wo[CONV == 7 ? LB::NR : 1], s7_ho[CONV == 7 ? LB::NR : 1], s7_t[CONV == 7 ? LB::NR : 1], s7_off[CONV == 7 ? LB::NR : 1];
  int s7_tapoff = 0;
  if (CONV == 7) {
    static_assert(CONV != 7 || BKM, "weight gradient: k-major patch operand");
    constexpr int CPR = BN / 4;                               // chunks per k row
    int c = (n0 >> 2) + (tid % CPR);
    if (c >= 72) c = 0;                                       // columns >= 288 are never stored: any valid address
    const int r = c >> 1, a = r / 7, b = r - 7 * a;
    s7_tapoff = (a * pHp + b) * pWp + 4 * (c & 1);
#pragma unroll
    for (int i = 0; i < LB::NR; ++i) {
      const int m = kbeg + (i * NT + tid) / CPR;
      const int wo = m % pWo, ho = (m / pWo) % pHo, ft = m / (pWo * pHo);
      s7_wo[i] = wo; s7_ho[i] = ho; s7_t[i] = ft % pT;
      s7_off[i] = (((ft / pT) * pTp + ft % pT) * pHp + 2 * ho) * pWp + 2 * wo;
    }
  }
  // CONV 2: the output pixel (image, oy, ox) of every chunk of this thread is CARRIED from K-step to K-step (tiles are issued in
  // k order; a step moves on by BK pixels = (sa_hi * cHo + sa_lo) rows + sb pixels, at most one wrap each): the three divisions
  // per chunk that recomputed it were ~170 integer instructions per K-step beside 32 MFMAs per wave
  int c2_ox[CONV == 2 ? LB::NR : 1], c2_oy[CONV == 2 ? LB::NR : 1], c2_img[CONV == 2 ? LB::NR : 1];
  int c2_sb = 0, c2_salo = 0, c2_sahi = 0, c2_tapoff = 0, c2_cb = 0, c2_dy = 0, c2_dx = 0;
  if (CONV == 2) {
    const int sa = BK / cWo;
    c2_sb = BK - sa * cWo;
    c2_sahi = sa / cHo;
    c2_salo = sa - c2_sahi * cHo;
    const int tap = n0 / d.conv_C;
    c2_cb = n0 - tap * d.conv_C;
    c2_dy = c9 ? tap / 3 - 1 : 0;
    c2_dx = c9 ? tap % 3 - 1 : 0;
    c2_tapoff = c2_dy * d.conv_W + c2_dx;
#pragma unroll
    for (int i = 0; i < LB::NR; ++i) {
      const int m = kbeg + (i * NT + tid) / (BN / 4);
      const int t = m / cWo;
      c2_ox[i] = m - t * cWo;
      c2_img[i] = t / cHo;
      c2_oy[i] = t - c2_img[i] * cHo;
    }
  }
  // CONV 3 (K tail: K % 32 != 0): chunks whose first k lies at or past K are fetched from a zero page; a k-contiguous chunk
  // that straddles K (K % 4 != 0) is fetched whole and its k >= K elements are zeroed in LDS before the last K-step
  int kofsA[LA::NR], kofsB[LB::NR];
  if (CONV == 3) {
#pragma unroll
    for (int i = 0; i < LA::NR; ++i) {
      const int q = i * NT + tid;
      kofsA[i] = AK ? q / (BM / 4) : ((q & 7) ^ (((q >> 3) >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int i = 0; i < LB::NR; ++i) {
      const int q = i * NT + tid;
      kofsB[i] = BKM ? q / (BN / 4) : ((q & 7) ^ (((q >> 3) >> 1) & 7)) * 4;
    }
  }
  auto issue = [&](int kt, int st) {
    if (CONV == 3) {
      const int kleft = kend - (kbeg + kt * BK);                 // k values of this step that exist
#pragma unroll
      for (int i = 0; i < LA::NR; ++i) {
        const float* src = kofsA[i] < kleft ? Ak + kt * kstepA + offA[i] : g_zero_page;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + (i * NT + wave * 64) * 4), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < LB::NR; ++i) {
        const float* src = kofsB[i] < kleft ? Bk + kt * kstepB + offB[i] : g_zero_page;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + ASZ + (i * NT + wave * 64) * 4), 16, 0, 0);
      }
      return;
    }
    if (CONV == 6) {
#pragma unroll
      for (int i = 0; i < LA::NR; ++i) {
        const int c = (kbeg >> 2) + kt * 8 + s6_cl[i], r = c >> 1, a = r / 7, b = r - 7 * a;
        const float* src = A + (s6_base[i] + (a * pHp + b) * pWp + 4 * (c & 1));
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + (i * NT + wave * 64) * 4), 16, 0, 0);
      }
    } else if (CONV == 4) {
      int tapoff[2], sh[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = kbeg + kt * BK + (q ? s4_kk1 : s4_kk0);
        const int a = k / 49, r = k - 49 * a, b = r / 7, c = r - 7 * b;
        tapoff[q] = ((a - 2) * sH + b) * sW + c;
        sh[q] = k < 245 ? (a | ((8 + b) << 8) | ((16 + c) << 16)) : -1;
      }
#pragma unroll
      for (int i = 0; i < NR4A; ++i) {
        const int q = i & 1, mk = s4_mask[i];
        const bool ok = sh[q] >= 0 && (((mk >> (sh[q] & 31)) & (mk >> ((sh[q] >> 8) & 31)) & (mk >> ((sh[q] >> 16) & 31))) & 1);
        const float* src = ok ? A + (s4_base[i] + tapoff[q]) : d.conv_zero;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + i * NT + wave * 64), 4, 0, 0);
      }
    } else if (CONV == 1) {
      const int kk = kbeg + kt * BK, tap = kk / d.conv_C;
      const int toff = c9 ? (tap / 3 - 1) * d.conv_W + (tap % 3 - 1) : 0;
      const int64_t delta = (int64_t)toff * d.conv_C + (kk - tap * d.conv_C);
#pragma unroll
      for (int i = 0; i < LA::NR; ++i) {
        const float* src = ((cmask[i] >> tap) & 1u) ? A + offA[i] + delta : d.conv_zero;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + (i * NT + wave * 64) * 4), 16, 0, 0);
      }
    } else {
      LA::issue(Ak + kt * kstepA, offA, smem + st * STAGE, wave);
    }
    if (CONV == 7) {
      constexpr int STEP = 32;                                // a thread's gather i of the next tile is 32 pixels further on
#pragma unroll
      for (int i = 0; i < LB::NR; ++i) {
        __builtin_amdgcn_global_load_lds((glb_float*)(B + (s7_off[i] + s7_tapoff)),
                                         (lds_float*)(smem + st * STAGE + ASZ + (i * NT + wave * 64) * 4), 16, 0, 0);
        s7_wo[i] += STEP;
        s7_off[i] += 2 * STEP;
        while (s7_wo[i] >= pWo) {                             // next output row: 2 rows of the padded frame further down
          s7_wo[i] -= pWo;
          s7_off[i] += 2 * pWp - 2 * pWo;
          if (++s7_ho[i] == pHo) {                            // next frame, at the end of a clip over its padding frames
            s7_ho[i] = 0;
            s7_off[i] += (pHp - 2 * pHo) * pWp;
            if (++s7_t[i] == pT) { s7_t[i] = 0; s7_off[i] += (pTp - pT) * pHp * pWp; }
          }
        }
      }
    } else if (CONV == 5) {
#pragma unroll
      for (int i = 0; i < NR4B; ++i) {
        const bool ok = s5_th && (unsigned)(2 * s5_wo + s5_c3) < (unsigned)sW;
        const float* src = ok ? B + (s5_off + s5_tapoff) : d.conv_zero;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + ASZ + i * NT + wave * 64), 4, 0, 0);
        s5_wo += 4;
        s5_off += 8;
        if (s5_wo >= sWo) {                          // wave-uniform: next output row (W = 2 Wo, H = 2 Ho: the offset in x
          s5_wo -= sWo;                              // moves on by one input row, also across frames and clips)
          s5_off += sW;
          if (++s5_ho == sHo) {
            s5_ho = 0;
            if (++s5_t == sT) s5_t = 0;
          }
          s5_th = s5_nok && (unsigned)(s5_t + s5_a - 2) < (unsigned)sT && (unsigned)(2 * s5_ho - 3 + s5_b) < (unsigned)sH;
        }
      }
    } else if (CONV == 2) {
#pragma unroll
      for (int i = 0; i < LB::NR; ++i) {
        const int r = (((i * NT + tid) % (BN / 4)) * 4);
        const int x = c2_ox[i] * cs + cp0, y = c2_oy[i] * cs + cp0;
        const bool ok = (unsigned)(y + c2_dy) < (unsigned)d.conv_H && (unsigned)(x + c2_dx) < (unsigned)d.conv_W;
        const int64_t pix = (int64_t)(c2_img[i] * d.conv_H + y) * d.conv_W + x;
        const float* src = ok ? B + (pix + c2_tapoff) * d.conv_C + c2_cb + r : d.conv_zero;
        __builtin_amdgcn_global_load_lds((glb_float*)src, (lds_float*)(smem + st * STAGE + ASZ + (i * NT + wave * 64) * 4), 16, 0, 0);
        // the next K-step's pixel
        int ox = c2_ox[i] + c2_sb;
        const int w1 = ox >= cWo ? 1 : 0;
        ox -= w1 ? cWo : 0;
        int oy = c2_oy[i] + c2_salo + w1;
        const int w2 = oy >= cHo ? 1 : 0;
        oy -= w2 ? cHo : 0;
        c2_ox[i] = ox; c2_oy[i] = oy; c2_img[i] += c2_sahi + w2;
      }
    } else {
      LB::issue(Bk + kt * kstepB, offB, smem + st * STAGE + ASZ, wave);
    }
  };
  const int arow = wm * TM * 32 + lr, brow = wn * TN * 32 + lr;
  auto compute = [&](int st) {
    const float* a_s = smem + st * STAGE;
    const float* b_s = a_s + ASZ;
    float af[2][TM][4], bf[2][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) read_frag_g<BM, AK>(a_s, arow + i * 32, kw, lk, af[0][i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_frag_g<BN, BKM>(b_s, brow + j * 32, kw, lk, bf[0][j]);
#pragma unroll
    for (int q = 0; q < NG / KW; ++q) {      // this wave set's k-groups: kw, kw + KW, ...
      const int c = q & 1;
      if (q + 1 < NG / KW) {
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag_g<BM, AK>(a_s, arow + i * 32, kw + (q + 1) * KW, lk, af[c ^ 1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag_g<BN, BKM>(b_s, brow + j * 32, kw + (q + 1) * KW, lk, bf[c ^ 1][j]);
      }
      // keep the order "next group's fragment reads, then this group's MFMAs": left alone, hipcc sinks the reads to just in
      // front of their first use and waits lgkmcnt(0) there - one exposed LDS latency per k-group (TAVSR_GEMM_SB=0 at build
      // time restores that)
      GEMM_SB();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i][kk], bf[c][j][kk], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) asum[i] += (af[c][i][0] + af[c][i][1]) + (af[c][i][2] + af[c][i][3]);
      GEMM_SB();
    }
  };

  // prologue: tiles 0 .. S-2 in flight
#pragma unroll
  for (int j = 0; j < S - 1; ++j)
    if (j < nk) issue(j, j);
  int st = 0;             // stage of tile kt
  int kt = 0;
  // steady state: tile kt landed when at most (S-2) tiles issued after it are still in flight
  for (; kt + S - 1 < nk; ++kt) {
    wait_vmcnt<(S - 2) * G>();
    __builtin_amdgcn_s_barrier();      // every wave's part of tile kt is in LDS; everyone left stage (kt-1) % S
#ifdef TAVSR_GEMM_TRACE
    if (kt == 0) TAVSR_TRACE_AT(1)
#endif
    const int sn = st == 0 ? S - 1 : st - 1;
    issue(kt + S - 1, sn);
    compute(st);
    st = st + 1 == S ? 0 : st + 1;
  }
  // drain: nothing left to issue
  for (; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (CONV == 3 && kt == nk - 1 && ((kend - kbeg) & 3) != 0) {
      // k-contiguous operands: the chunk that straddles K brought 1..3 elements of k >= K along: zero them (row r,
      // element kk of a stage lives at r*32 + (((kk >> 2) ^ ((r >> 1) & 7)) << 2) + (kk & 3))
      const int kt_len = (kend - kbeg) - kt * BK, k4 = (kt_len + 3) & ~3;
      float* a_s = smem + st * STAGE;
      float* b_s = a_s + ASZ;
      if (!AK)
        for (int r = tid; r < BM; r += NT)
          for (int kk = kt_len; kk < k4; ++kk) a_s[r * 32 + ((((kk >> 2) ^ ((r >> 1) & 7))) << 2) + (kk & 3)] = 0.f;
      if (!BKM)
        for (int r = tid; r < BN; r += NT)
          for (int kk = kt_len; kk < k4; ++kk) b_s[r * 32 + ((((kk >> 2) ^ ((r >> 1) & 7))) << 2) + (kk & 3)] = 0.f;
      __syncthreads();
    }
    compute(st);
    st = st + 1 == S ? 0 : st + 1;
  }
  TAVSR_TRACE_AT(2)
  if (KW > 1) {     // sum the wave sets' accumulators (and row sums) through LDS; set 0 runs the epilogue
    constexpr int PER = TM * TN * 16;
    float* red = smem;                                         // [KW-1][WM*WN][PER][64]
    float* rsum = smem + (KW - 1) * WM * WN * PER * 64;        // [KW-1][WM*WN][TM][64]
    static_assert(((KW - 1) * WM * WN * (PER + TM) * 64) <= S * STAGE, "reduction must fit in the staging ring");
    __syncthreads();                                           // all LDS reads of the K loop are done
    if (kw > 0) {
      float* r0 = red + ((kw - 1) * WM * WN + w2) * PER * 64 + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) r0[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
#pragma unroll
      for (int i = 0; i < TM; ++i) rsum[(((kw - 1) * WM * WN + w2) * TM + i) * 64 + lane] = asum[i];
    }
    __syncthreads();
    if (kw > 0) return;
#pragma unroll
    for (int s2 = 0; s2 < KW - 1; ++s2) {
      const float* r0 = red + (s2 * WM * WN + w2) * PER * 64 + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += r0[((i * TN + j) * 16 + r) * 64];
#pragma unroll
      for (int i = 0; i < TM; ++i) asum[i] += rsum[((s2 * WM * WN + w2) * TM + i) * 64 + lane];
    }
  }
  if (KW == 1 && vec_epi) {
    static_assert(KW > 1 || BM * BN <= S * STAGE, "the output tile image must fit in the staging ring");
    if (want_rowsum) {              // bias gradients (row sums of op(A)): as finish_tile
#pragma unroll
      for (int i = 0; i < TM; ++i) asum[i] += __shfl_xor(asum[i], 32, 64);
      if (lk == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + wm * TM * 32 + i * 32 + lr;
          if (m < d.M) {
            if (nsplit > 1) (d.ws + (int64_t)nsplit * gridDim.y * d.M * d.N)[(int64_t)zidx * d.M + m] = asum[i];
            else d.a_rowsum[m] = d.alpha * asum[i];
          }
        }
      }
    }
    finish_tile_vec<BM, BN, NT, TM, TN>(d, nsplit, acc, smem, m0, n0, wm, wn, lr, lk, z1, z2, coff, tid, zidx);
  } else {
    finish_tile<TM, TN>(d, nsplit, acc, asum, want_rowsum, m0, n0, wm, wn, lr, lk, z1, z2, coff, zidx, bpre);
  }
#ifdef TAVSR_GEMM_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  TAVSR_TRACE_AT(3)
  if (tid == 0) {
    unsigned int slot = atomicAdd(&g_trace_n, 1u);
    if (slot < (unsigned)kTraceMax) {
      for (int i = 0; i < 4; ++i) g_trace[slot][i] = tr_t[i];
      g_trace[slot][4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
      g_trace[slot][5] = tr_c[2] - tr_c[1];      // shader-clock cycles of the K loop
    }
  }
#endif
}

// XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (its L2): give them neighbouring tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int BM, int BN, int WM, int WN, int S, int MINW, bool AK, bool BKM, int KW = 1, int CONV = 0>
__global__ __launch_bounds__(WM* WN * KW * 64, MINW)
void gemm_glds_kernel(const GemmArgs args) {
  int bid = xcd_remap(blockIdx.x, gridDim.x), zidx = blockIdx.z;
  if (args.zmap) {
    // Workgroups go to the XCDs round-robin in launch order (x fastest, then z).  The tiles of one K slice of a convolution
    // weight gradient read the same dY rows and overlapping image rows (one tile per tap / channel block): give ALL tiles
    // of a slice to one XCD, back to back, so that the slice is fetched from HBM once and served from that XCD's L2 to the
    // others.  Launch l = x + tiles * z runs on XCD l % 8 as that XCD's (l / 8)-th block: slice (l % 8) + 8 * ((l / 8) /
    // tiles), tile (l / 8) % tiles - a bijection when the number of slices is a multiple of 8 (host).
    const int tiles = gridDim.x, l = blockIdx.x + tiles * blockIdx.z, j = l >> 3;
    zidx = (l & 7) + 8 * (j / tiles);
    bid = j % tiles;
  }
  glds_tile<BM, BN, WM, WN, S, AK, BKM, KW, CONV>(args.d, args.kchunk, args.nsplit, args.tiles_n, bid, args.vec_epi != 0, zidx);
}

// Grouped launch: up to kMaxGroup independent problems of one layout share ONE grid (tile ranges by prefix sums).
// The weight gradients of a layer have 16-128 tiles each: alone they need a K split (slabs + a second launch);
// together they fill the chip without one.
constexpr int kMaxGroup = 12;
struct GroupArgs {
  tavsr_gemm_desc d[kMaxGroup];
  int tile_start[kMaxGroup + 1];
  int tiles_n[kMaxGroup];
  int n;
  int vec_epi;    // every problem of the group qualifies for finish_tile_vec
};

template <int BM, int BN, int WM, int WN, int S, int MINW, bool AK, bool BKM>
__global__ __launch_bounds__(WM* WN * 64, MINW)
void gemm_glds_grouped_kernel(const GroupArgs g) {
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < g.n && bid >= g.tile_start[i]) pi = i;
  glds_tile<BM, BN, WM, WN, S, AK, BKM>(g.d[pi], g.d[pi].K, 1, g.tiles_n[pi], bid - g.tile_start[pi], g.vec_epi != 0, 0);
}

// ---------------------------------------------------------------------------------------------- host side
struct Cfg {
  int bm, bn, wm, wn, stages;
};
// LDS-DMA tile configurations (index = cfg id of tavsr_gemm_tune); id kFallbackCfg = the register-staged,
// fully predicated 64x64 kernel that serves unaligned operands, K tails and ragged row-contiguous operands.
static const Cfg kCfgs[] = {
    {128, 128, 2, 2, 3},   // 0: wave tile 64x64, 96 KB LDS, one block per CU
    {128, 128, 2, 4, 3},   // 1: 8 waves, wave tile 64x32 (two waves per SIMD inside the block)
    {128, 64, 2, 2, 3},    // 2: wave tile 64x32, 72 KB, two blocks per CU
    {64, 128, 2, 2, 3},    // 3: wave tile 32x64
    {64, 64, 2, 2, 3},     // 4: wave tile 32x32, 48 KB, three blocks per CU
    {64, 64, 2, 2, 4},     // 5: wave tile 32x32, four stages (three tiles in flight), two blocks per CU
    {128, 64, 2, 2, 4},    // 6: wave tile 64x32, four stages, one block per CU
    {64, 64, 2, 2, 3},     // 7: as 4 with the K-step split over 2 wave sets (8 waves)
    {64, 64, 2, 2, 2},     // 8: two stages (32 KB): five blocks per CU
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kFallbackCfg = 9;

template <typename F>
static int launch_layout(const tavsr_gemm_desc& d, F&& f) {
  if (!d.a_kmajor && !d.b_kmajor) return f(std::false_type{}, std::false_type{});
  if (!d.a_kmajor && d.b_kmajor) return f(std::false_type{}, std::true_type{});
  if (d.a_kmajor && d.b_kmajor) return f(std::true_type{}, std::true_type{});
  return f(std::true_type{}, std::false_type{});
}

// tavsr_gemm_ln: the LayerNorm that follows a Linear, taken where the result row is finished.  A one-token step of a batched search
// (640 hypothesis rows) runs its N = 256 / 512 projections with K split over workgroups: the slabs are summed by an epilogue launch
// and the next block's LayerNorm is another launch over the same rows - 38 + 38 launches of ~5 us per token at batch 64.  Here one
// wave per row sums the slabs (fixed order), applies the same epilogue arithmetic as splitk_epilogue_kernel, stores C and, with the
// row still in registers, its LayerNorm (two-pass statistics, as layernorm_fwd_kernel).  nsplit == 1: C is final; the wave reads it.
struct LnTail {
  const float* gamma; const float* beta; float eps; float* out; int64_t ld;
};
static const LnTail* g_ln_tail = nullptr;      // set by tavsr_gemm_ln around its run(): the library is not re-entrant

constexpr int kLnTailMaxN = 2048;
__global__ __launch_bounds__(256) void epilogue_ln_kernel(const GemmArgs args, const LnTail ln) {
  const tavsr_gemm_desc& d = args.d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= d.M) return;
  constexpr int Q = kLnTailMaxN / 256;             // float4s per lane
  const int n4 = d.N >> 2;
  const int64_t mn = (int64_t)d.M * d.N;
  float4 x[Q];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c4 = lane + 64 * q;
    x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < n4) {
      const int n = 4 * c4;
      const int64_t o = (int64_t)m * d.ldc + n;
      float4 v;
      if (args.nsplit > 1) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = d.ws + (int64_t)m * d.N + n;
        for (int sidx = 0; sidx < args.nsplit; ++sidx) {
          const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)sidx * mn);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y = e[k];
          if (d.bias) y += d.bias[n + k];
          y = act_fwd(d.act, y);
          y *= d.alpha;
          if (d.R) y += d.R[(int64_t)m * d.ldr + n + k];
          e[k] = y;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<float4*>(d.C + o) = v;
      } else {
        v = *reinterpret_cast<const float4*>(d.C + o);
      }
      x[q] = v;
      sum += (v.x + v.y) + (v.z + v.w);
    }
  }
  const float mean = wave_sum(sum) / (float)d.N;
  float sq = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (lane + 64 * q < n4) {
      const float a = x[q].x - mean, b = x[q].y - mean, c = x[q].z - mean, e = x[q].w - mean;
      sq += (a * a + b * b) + (c * c + e * e);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)d.N + ln.eps);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c4 = lane + 64 * q;
    if (c4 < n4) {
      const float4 g = *reinterpret_cast<const float4*>(ln.gamma + 4 * c4), b = *reinterpret_cast<const float4*>(ln.beta + 4 * c4);
      *reinterpret_cast<float4*>(ln.out + (int64_t)m * ln.ld + 4 * c4) =
          make_float4((x[q].x - mean) * rstd * g.x + b.x, (x[q].y - mean) * rstd * g.y + b.y, (x[q].z - mean) * rstd * g.z + b.z,
                      (x[q].w - mean) * rstd * g.w + b.w);
    }
  }
}

static int launch_epilogue(const GemmArgs& a, hipStream_t s) {
  const tavsr_gemm_desc& d = a.d;
  if (g_ln_tail) {                 // tavsr_gemm_ln: slab sum + epilogue + the next LayerNorm, one wave per row
    hipLaunchKernelGGL(epilogue_ln_kernel, dim3(cdiv(d.M, 4)), dim3(256), 0, s, a, *g_ln_tail);
    TAVSR_LAUNCH_CHECK();
    return TAVSR_OK;
  }
  if (a.nsplit > 1) {
    const int64_t mn = (int64_t)d.M * d.N;
    if (d.N % 4 == 0)
      hipLaunchKernelGGL(splitk_epilogue_kernel<4>, dim3(cdiv(mn / 4, 256), d.nb1 * d.nb2), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(splitk_epilogue_kernel<1>, dim3(cdiv(mn, 256), d.nb1 * d.nb2), dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
  }
  return TAVSR_OK;
}

template <int BM, int BN, int WM, int WN, int S, int MINW, int KW = 1>
static int launch_glds(const tavsr_gemm_desc& d, int nsplit, int kchunk, hipStream_t s) {
  GemmArgs a{d, kchunk, nsplit, cdiv(d.M, BM), cdiv(d.N, BN), (int)vec_epi_ok(d)};
  dim3 grid(a.tiles_m * a.tiles_n, d.nb1 * d.nb2, nsplit);
  int rc = launch_layout(d, [&](auto ak, auto bk) {
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, S, MINW, decltype(ak)::value, decltype(bk)::value, KW>), grid,
                       dim3(WM * WN * KW * 64), 0, s, a);
    TAVSR_LAUNCH_CHECK();
    return (int)TAVSR_OK;
  });
  return rc ? rc : launch_epilogue(a, s);
}

// implicit-convolution launches (two-stage 64x64 variant): mode 1 = A patches (NT / NN), mode 2 = B patches (TN)
static int launch_conv(const tavsr_gemm_desc& d, int nsplit, int kchunk, hipStream_t s) {
  const int ve = (int)vec_epi_ok(d);
  // Wider tiles where the shape allows - the gathered patch operand is the expensive one to load:
  //   forward / data gradient: a 64x128 tile reads the image rows once for two column tiles of weights (Cout % 128 == 0):
  //     +0.6 % on the AV step (128x64 and 128x128 tiles: nothing / worse);
  //   weight gradient: a 128x64 tile (Cout % 128 == 0) shares one patch tile between 128 output channels: +1.6 %, and
  //     another +0.7 % with the K split re-fitted to its three block slots per CU (2304 blocks).
  // (in-call A/B of rounds 1-2; the run-time switches are gone, the constants below record the winners)
  constexpr int wide = 1;
  constexpr int dw_wide = 1;
  constexpr int zmap_on = 1;
  const int zmap = zmap_on && (d.conv_mode == 2 || d.conv_mode == 5 || d.conv_mode == 7) && nsplit >= 8 && nsplit % 8 == 0;
  if (d.conv_mode == 6 || d.conv_mode == 7) {      // Conv3d stem over padded clips: ordinary 16-byte chunks
    GemmArgs a6{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 64), ve, zmap};
    const dim3 grid6(a6.tiles_m * a6.tiles_n, 1, nsplit);
    constexpr int st3 = 2;   // tuning aid
    constexpr int tile128 = 0;   // tuning aid
    if (d.conv_mode == 6 && tile128) {
      GemmArgs a7{d, kchunk, nsplit, cdiv(d.M, 128), cdiv(d.N, 64), ve, 0};
      hipLaunchKernelGGL((gemm_glds_kernel<128, 64, 2, 2, 2, 3, false, false, 1, 6>), dim3(a7.tiles_m * a7.tiles_n, 1, 1), dim3(256), 0, s, a7);
      TAVSR_LAUNCH_CHECK();
      return TAVSR_OK;
    }
    if (d.conv_mode == 6 && st3 == 3)
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 3, 3, false, false, 1, 6>), grid6, dim3(256), 0, s, a6);
    else if (d.conv_mode == 6)
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, false, false, 1, 6>), grid6, dim3(256), 0, s, a6);
    else if (st3 == 3)
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 3, 3, true, true, 1, 7>), grid6, dim3(256), 0, s, a6);
    else
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, true, true, 1, 7>), grid6, dim3(256), 0, s, a6);
    TAVSR_LAUNCH_CHECK();
    return launch_epilogue(a6, s);
  }
  if (d.conv_mode == 4 || d.conv_mode == 5) {      // Conv3d stem: 4-byte gathers
    GemmArgs a4{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 64), ve, zmap};
    const dim3 grid4(a4.tiles_m * a4.tiles_n, 1, nsplit);
    if (d.conv_mode == 4)
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, false, false, 1, 4>), grid4, dim3(256), 0, s, a4);
    else
      hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, true, true, 1, 5>), grid4, dim3(256), 0, s, a4);
    TAVSR_LAUNCH_CHECK();
    return launch_epilogue(a4, s);
  }
  if (d.conv_mode == 1 && !d.b_kmajor && nsplit == 1 && wide && d.N % 128 == 0) {
    GemmArgs a2{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 128), (int)vec_epi_ok(d)};
    hipLaunchKernelGGL((gemm_glds_kernel<64, 128, 2, 2, 2, 3, false, false, 1, 1>), dim3(a2.tiles_m * a2.tiles_n, 1, 1), dim3(256), 0, s, a2);
    TAVSR_LAUNCH_CHECK();
    return TAVSR_OK;
  }
  if (d.conv_mode == 2 && dw_wide && d.M % 128 == 0) {     // weight gradient: 128 output channels share one patch tile
    GemmArgs a2{d, kchunk, nsplit, cdiv(d.M, 128), cdiv(d.N, 64), (int)vec_epi_ok(d), zmap};
    hipLaunchKernelGGL((gemm_glds_kernel<128, 64, 2, 2, 2, 3, true, true, 1, 2>), dim3(a2.tiles_m * a2.tiles_n, 1, nsplit), dim3(256), 0, s, a2);
    TAVSR_LAUNCH_CHECK();
    return launch_epilogue(a2, s);
  }
  GemmArgs a{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 64), ve, zmap};
  dim3 grid(a.tiles_m * a.tiles_n, 1, nsplit);
  if (d.conv_mode == 1 && !d.b_kmajor)
    hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, false, false, 1, 1>), grid, dim3(256), 0, s, a);
  else if (d.conv_mode == 1)
    hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, false, true, 1, 1>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, true, true, 1, 2>), grid, dim3(256), 0, s, a);
  TAVSR_LAUNCH_CHECK();
  return launch_epilogue(a, s);
}

static int launch_fallback(const tavsr_gemm_desc& d, bool vec, int nsplit, int kchunk, hipStream_t s) {
  GemmArgs a{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 64), 0};
  dim3 grid(a.tiles_m * a.tiles_n, d.nb1 * d.nb2, nsplit);
  int rc = launch_layout(d, [&](auto ak, auto bk) {
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<64, 64, 32, 2, 2, 2, 4, decltype(ak)::value, decltype(bk)::value, true>), grid,
                         dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_kernel<64, 64, 32, 2, 2, 2, 4, decltype(ak)::value, decltype(bk)::value, false>), grid,
                         dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
    return (int)TAVSR_OK;
  });
  return rc ? rc : launch_epilogue(a, s);
}

static int launch_tail(const tavsr_gemm_desc& d, int nsplit, int kchunk, hipStream_t s) {
  GemmArgs a{d, kchunk, nsplit, cdiv(d.M, 64), cdiv(d.N, 64), (int)vec_epi_ok(d)};
  dim3 grid(a.tiles_m * a.tiles_n, d.nb1 * d.nb2, nsplit);
  int rc = launch_layout(d, [&](auto ak, auto bk) {
    hipLaunchKernelGGL((gemm_glds_kernel<64, 64, 2, 2, 2, 5, decltype(ak)::value, decltype(bk)::value, 1, 3>), grid, dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
    return (int)TAVSR_OK;
  });
  return rc ? rc : launch_epilogue(a, s);
}

static int launch(int cfg, const tavsr_gemm_desc& d, bool vec, int nsplit, int kchunk, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_glds<128, 128, 2, 2, 3, 1>(d, nsplit, kchunk, s);
    case 1: return launch_glds<128, 128, 2, 4, 3, 2>(d, nsplit, kchunk, s);
    case 2: return launch_glds<128, 64, 2, 2, 3, 2>(d, nsplit, kchunk, s);
    case 3: return launch_glds<64, 128, 2, 2, 3, 2>(d, nsplit, kchunk, s);
    case 4: return launch_glds<64, 64, 2, 2, 3, 3>(d, nsplit, kchunk, s);
    case 5: return launch_glds<64, 64, 2, 2, 4, 2>(d, nsplit, kchunk, s);
    case 6: return launch_glds<128, 64, 2, 2, 4, 1>(d, nsplit, kchunk, s);
    case 7: return launch_glds<64, 64, 2, 2, 3, 4, 2>(d, nsplit, kchunk, s);
    case 8: return launch_glds<64, 64, 2, 2, 2, 5>(d, nsplit, kchunk, s);
    default: return launch_fallback(d, vec, nsplit, kchunk, s);
  }
}

struct Plan {
  int cfg, nsplit, kchunk;
};

// Can the LDS-DMA kernel take this problem?  (unpredicated 16-byte loads: aligned operands, whole K-steps per
// K slice, row-contiguous operands with a row count that is a multiple of 4)
static bool glds_ok(const tavsr_gemm_desc& d, bool vec) {
  return vec && d.K % 32 == 0 && d.K >= 32 && (!d.a_kmajor || d.M % 4 == 0) && (!d.b_kmajor || d.N % 4 == 0);
}

// K % 32 != 0 on the LDS-DMA kernel (tail variant): 16-byte chunks past K come from a zero page; a k-contiguous operand
// must hold the (up to 3) elements between K and the next multiple of 4 inside its rows (ld >= roundup4(K))
static bool tail_ok(const tavsr_gemm_desc& d, bool vec) {
  constexpr int on = 1;
  const int64_t k4 = (d.K + 3) / 4 * 4;
  return on && vec && d.K % 32 != 0 && d.K >= 32 && (!d.a_kmajor || d.M % 4 == 0) && (!d.b_kmajor || d.N % 4 == 0) &&
         (d.a_kmajor || d.lda >= k4) && (d.b_kmajor || d.ldb >= k4) && d.conv_mode == 0;
}

// Planner (fitted to profiles/r01_gemm_sweep_v3/v4.txt and end-to-end A/B runs, MI355X).  The 64x64 tile wins or ties every
// hot-path shape (larger tiles lose more to tile quantisation at M = 3168 than they gain).  Few-tile, long-K problems
// (weight gradients: K = B*T; the N = 256 projections with K >= 2048) are split over K until about 1000 blocks exist
// (four of the five block slots of every CU: 5 slices for the 200-tile shapes); more slices than that cost more in
// slab traffic than they gain in balance.
static Plan plan(const tavsr_gemm_desc& d, bool allow_split, bool fast) {
  const long nbatch = (long)d.nb1 * d.nb2;
  Plan p{kFallbackCfg, 1, d.K};
  const long tiles = (long)cdiv(d.M, 64) * cdiv(d.N, 64) * nbatch;
  // tile variant: two LDS stages (32 KB, five blocks per CU cover each other's epilogues).  The K-step-split variant
  // (cfg 7) wins isolated one-block-per-CU launches by 5-8 % but loses inside the two-stream step (end-to-end A/B).
  // (tavsr_gemm_tune forces a variant for tests / sweeps)
  constexpr int forced = -1;
  auto variant = [&](long blocks) { (void)blocks; return !fast ? kFallbackCfg : forced >= 0 ? forced : 8; };
  p.cfg = variant(tiles);
  if (!allow_split || tiles >= 384 || d.K < 512) return p;
  if (d.K <= 1024 && tiles >= 150) return p;
  constexpr long target = 1000L;   // tuning aid
  long want = std::min<long>((target + tiles / 2) / tiles, d.K / 256);
  if (want < 2) return p;
  p.kchunk = cdiv(cdiv(d.K, want), 32) * 32;
  p.nsplit = cdiv(d.K, p.kchunk);
  if (p.nsplit < 2) p = Plan{p.cfg, 1, d.K};
  p.cfg = variant(tiles * p.nsplit);
  return p;
}

// plan of an implicit-convolution launch: the weight gradient (mode 2) has an enormous K = frames*H*W and few tiles, so K
// is split until all five block slots of every CU are filled (the slabs stay tiny)
static Plan plan_conv(const tavsr_gemm_desc& d, bool can_split) {
  Plan pc = plan(d, can_split, true);
  if ((d.conv_mode == 2 || d.conv_mode == 5 || d.conv_mode == 7) && can_split) {
    constexpr int dw_wide = 1;
    const bool wide = d.conv_mode == 2 && dw_wide && d.M % 128 == 0;     // 128x64 tiles (launch_conv): three block slots per CU
    const long tiles = (long)cdiv(d.M, wide ? 128 : 64) * cdiv(d.N, 64);
    constexpr long target64 = 2560L;
    constexpr long target128 = 2304L;
    const long target = wide ? target128 : target64;
    const long want = std::min<long>(std::max<long>(1, target / tiles), d.K / 512);
    if (want > pc.nsplit) {
      pc.kchunk = cdiv(cdiv(d.K, want), 32) * 32;
      pc.nsplit = cdiv(d.K, pc.kchunk);
    }
    if (pc.nsplit >= 16 && pc.nsplit % 8 != 0) {        // a multiple of 8 slices lets launch_conv keep each slice on one XCD
      for (long w8 = pc.nsplit / 8 * 8; w8 >= 8; w8 -= 8) {
        const int kc = cdiv(cdiv(d.K, w8), 32) * 32;
        if (cdiv(d.K, kc) % 8 == 0) { pc.kchunk = kc; pc.nsplit = cdiv(d.K, kc); break; }
        if (w8 < pc.nsplit / 2) break;
      }
    }
  }
  return pc;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int64_t ws_floats_for(const tavsr_gemm_desc& d, int nsplit) {
  if (nsplit <= 1) return 0;
  return (int64_t)nsplit * d.nb1 * d.nb2 * d.M * d.N + (d.a_rowsum ? (int64_t)nsplit * d.M : 0);
}

static int run(const tavsr_gemm_desc* dp, int force_cfg, int force_split, hipStream_t s) {
  TAVSR_REQUIRE(dp != nullptr, TAVSR_EINVAL, "tavsr_gemm: null descriptor");
  tavsr_gemm_desc d = *dp;
  TAVSR_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, TAVSR_EINVAL, "tavsr_gemm: negative dims");
  if (d.nb1 <= 0) d.nb1 = 1;
  if (d.nb2 <= 0) d.nb2 = 1;
  if (d.M == 0 || d.N == 0) return TAVSR_OK;
  TAVSR_REQUIRE(d.A && d.B && d.C, TAVSR_EINVAL, "tavsr_gemm: null operand");
  TAVSR_REQUIRE((long)d.nb1 * d.nb2 <= 65535, TAVSR_EINVAL, "tavsr_gemm: batch too large");
  TAVSR_REQUIRE(d.a_rowsum == nullptr || d.nb1 * d.nb2 == 1, TAVSR_EUNSUPPORTED,
                "tavsr_gemm: a_rowsum needs an unbatched problem");
  if (d.R == nullptr) { d.ldr = 0; d.sR1 = d.sR2 = 0; }
  // vector (16-B) operand loads need aligned bases, leading dims and batch strides
  const bool vec = aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.sA1 % 4 == 0 &&
                   d.sA2 % 4 == 0 && d.sB1 % 4 == 0 && d.sB2 % 4 == 0;
  const bool can_split = d.ws != nullptr;
  const bool fast = glds_ok(d, vec);
  if (d.conv_mode != 0) {       // implicit 3x3/s1/p1 convolution: only the LDS-DMA kernel reads images as patch operands
    TAVSR_REQUIRE(d.conv_mode == 1 || d.conv_mode == 2 || (d.conv_mode >= 4 && d.conv_mode <= 7), TAVSR_EINVAL,
                  "tavsr_gemm: conv_mode must be 0, 1, 2 or 4..7");
    TAVSR_REQUIRE(d.drop_p == 0.f, TAVSR_EUNSUPPORTED, "tavsr_gemm: no epilogue dropout on convolution operands");
    TAVSR_REQUIRE(d.conv_zero && aligned16(d.conv_zero) && d.conv_H > 0 && d.conv_W > 0 && d.conv_C > 0, TAVSR_EINVAL,
                  "tavsr_gemm: conv needs H, W, C and a 16-byte aligned zero page");
    TAVSR_REQUIRE(d.nb1 * d.nb2 == 1 && fast && force_cfg < 0, TAVSR_EUNSUPPORTED,
                  "tavsr_gemm: conv operands need an unbatched, aligned problem with K %% 32 == 0");
    if (d.conv_mode >= 6) {       // Conv3d stem over padded clips [clips][conv_C = T + 5][conv_H = H + 6][conv_W = W + 8], 288 tap columns
      TAVSR_REQUIRE(d.conv_C > 5 && d.conv_H > 6 && d.conv_W > 8 && (d.conv_H - 6) % 2 == 0 && (d.conv_W - 8) % 2 == 0 &&
                        d.conv_W % 4 == 0, TAVSR_EINVAL, "tavsr_gemm: padded stem clips are [T + 5][H + 6][W + 8], H and W even");
      const int64_t per_clip = (int64_t)(d.conv_C - 5) * ((d.conv_H - 6) / 2) * ((d.conv_W - 8) / 2);
      const int64_t pixels = d.conv_mode == 6 ? d.M : d.K;
      TAVSR_REQUIRE(pixels % per_clip == 0 && pixels / per_clip * d.conv_C * d.conv_H * d.conv_W < (1ll << 31), TAVSR_EINVAL,
                    "tavsr_gemm: stem rows must be whole clips (fewer than 2^31 padded input pixels)");
      if (d.conv_mode == 6)
        TAVSR_REQUIRE(!d.a_kmajor && !d.b_kmajor && d.K == 288 && d.ldb >= 288, TAVSR_EUNSUPPORTED,
                      "tavsr_gemm: conv mode 6 needs the NT layout with K = 288 (35 x 8 tap columns + padding)");
      else
        TAVSR_REQUIRE(d.a_kmajor && d.b_kmajor && d.N == 288 && d.M % 4 == 0, TAVSR_EUNSUPPORTED,
                      "tavsr_gemm: conv mode 7 needs the TN layout with N = 288 (35 x 8 tap columns + padding)");
      Plan p6 = plan_conv(d, can_split);
      if (p6.nsplit > 1 && d.ws_floats < ws_floats_for(d, p6.nsplit)) p6 = plan(d, false, true);
      return launch_conv(d, p6.nsplit, p6.kchunk, s);
    }
    if (d.conv_mode >= 4) {       // Conv3d stem: conv_H x conv_W input frames, conv_C frames per clip, 245 taps padded to 256
      const int64_t per_frame = (int64_t)((d.conv_H - 1) / 2 + 1) * ((d.conv_W - 1) / 2 + 1);
      const int64_t pixels = d.conv_mode == 4 ? d.M : d.K;
      TAVSR_REQUIRE(pixels % (per_frame * d.conv_C) == 0 && d.conv_C < 1024 && d.conv_H < 1000 && d.conv_W < 1000 &&
                        pixels / per_frame * d.conv_H * d.conv_W < (1ll << 31),
                    TAVSR_EINVAL, "tavsr_gemm: stem rows must be whole clips of conv_C frames (fewer than 2^31 input pixels)");
      if (d.conv_mode == 4)
        TAVSR_REQUIRE(!d.a_kmajor && !d.b_kmajor && d.K == 256 && d.ldb >= 256, TAVSR_EUNSUPPORTED,
                      "tavsr_gemm: conv mode 4 needs the NT layout with K = 256 (245 taps + padding)");
      else
        TAVSR_REQUIRE(d.a_kmajor && d.b_kmajor && d.N == 256 && d.M % 4 == 0, TAVSR_EUNSUPPORTED,
                      "tavsr_gemm: conv mode 5 needs the TN layout with N = 256 (245 taps + padding)");
      Plan p4 = plan_conv(d, can_split);
      if (p4.nsplit > 1 && d.ws_floats < ws_floats_for(d, p4.nsplit)) p4 = plan(d, false, true);
      return launch_conv(d, p4.nsplit, p4.kchunk, s);
    }
    const int cs = d.conv_stride > 1 ? d.conv_stride : 1, taps = d.conv_taps == 1 ? 1 : 9;
    TAVSR_REQUIRE(d.conv_taps == 0 || d.conv_taps == 1 || d.conv_taps == 9 || d.conv_taps == 90, TAVSR_EINVAL,
                  "tavsr_gemm: conv_taps must be 1, 9 or 90 (3x3 without padding)");
    const int p0 = d.conv_taps == 90 ? 1 : 0;
    TAVSR_REQUIRE(!p0 || (d.conv_H >= 3 && d.conv_W >= 3), TAVSR_EINVAL, "tavsr_gemm: an unpadded 3x3 window needs a 3x3 image");
    const int64_t pixels = d.conv_mode == 1 ? d.M : d.K;       // output pixels
    const int64_t per_image = (int64_t)((d.conv_H - 1 - 2 * p0) / cs + 1) * ((d.conv_W - 1 - 2 * p0) / cs + 1);
    TAVSR_REQUIRE(pixels % per_image == 0, TAVSR_EINVAL, "tavsr_gemm: conv rows are not whole images");
    if (d.conv_mode == 1)
      TAVSR_REQUIRE(!d.a_kmajor && d.K == taps * d.conv_C && d.conv_C % 32 == 0 && d.lda == d.conv_C, TAVSR_EUNSUPPORTED,
                    "tavsr_gemm: conv mode 1 needs a row-major image operand A, K = taps * C, C %% 32 == 0");
    else
      TAVSR_REQUIRE(d.a_kmajor && d.b_kmajor && d.N == taps * d.conv_C && d.conv_C % 64 == 0 && d.ldb == d.conv_C,
                    TAVSR_EUNSUPPORTED, "tavsr_gemm: conv mode 2 needs the TN layout, N = taps * C, C %% 64 == 0");
    Plan pc = plan_conv(d, can_split);
    if (pc.nsplit > 1 && d.ws_floats < ws_floats_for(d, pc.nsplit)) pc = plan(d, false, true);
    return launch_conv(d, pc.nsplit, pc.kchunk, s);
  }
  const bool tail = !fast && force_cfg < 0 && tail_ok(d, vec);
  if (d.drop_p > 0.f) {
    TAVSR_REQUIRE(d.drop_p < 1.f && d.drop_seed && d.drop_offset % 4 == 0, TAVSR_EINVAL,
                  "tavsr_gemm: dropout needs p in (0, 1), a device seed and an offset %% 4 == 0");
    TAVSR_REQUIRE((fast || tail) && force_cfg < 0 && d.nb1 * d.nb2 == 1 && vec_epi_ok(d), TAVSR_EUNSUPPORTED,
                  "tavsr_gemm: epilogue dropout needs an unbatched problem on the 16-byte path");
  }
  TAVSR_REQUIRE((d.rowdot_a == nullptr) == (d.rowdot_b == nullptr) && (!d.rowdot_a || d.rowstat), TAVSR_EINVAL,
                "tavsr_gemm: rowdot_a and rowdot_b go together, with rowstat as their output");
  if (d.rowstat) {
    TAVSR_REQUIRE(fast && force_cfg < 0 && d.nb1 * d.nb2 == 1 && vec_epi_ok(d) && aligned16(d.rowstat) && aligned16(d.rowdot_a) &&
                      aligned16(d.rowdot_b), TAVSR_EUNSUPPORTED,
                  "tavsr_gemm: row statistics / row dots need an unbatched problem on the 16-byte path");
    return launch(8, d, vec, 1, d.K, s);             // 64-wide tiles, no K split: the statistics are taken where the tile is stored
  }
  Plan p = plan(d, can_split, fast || tail);
  if (tail) {
    if (p.nsplit > 1 && d.ws_floats < ws_floats_for(d, p.nsplit)) p = plan(d, false, true);
    return launch_tail(d, p.nsplit, p.kchunk, s);
  }
  if (force_cfg >= 0) {
    TAVSR_REQUIRE(force_cfg < kNumCfgs || force_cfg == kFallbackCfg, TAVSR_EINVAL, "tavsr_gemm_tune: cfg %d out of range",
                  force_cfg);
    p.cfg = fast ? force_cfg : kFallbackCfg;
    const int bk = 32;
    int ns = std::max(1, force_split);
    p.kchunk = cdiv(cdiv(d.K, ns), bk) * bk;
    p.nsplit = std::max(1, cdiv(d.K, p.kchunk));
  }
  if (p.nsplit > 1) {
    if (!can_split || d.ws_floats < ws_floats_for(d, p.nsplit)) {
      TAVSR_REQUIRE(force_cfg < 0, TAVSR_EINVAL, "tavsr_gemm_tune: workspace too small for the forced split");
      p = plan(d, false, fast);
    }
  }
  return launch(p.cfg, d, vec, p.nsplit, p.kchunk, s);
}

}  // namespace tavsr

extern "C" int tavsr_gemm(const tavsr_gemm_desc* dp, tavsr_stream_t stream) {
  return tavsr::run(dp, -1, 0, static_cast<hipStream_t>(stream));
}

extern "C" int tavsr_gemm_ln(const tavsr_gemm_desc* dp, const float* gamma, const float* beta, float eps, float* ln_out, int64_t ld_ln,
                             tavsr_stream_t stream) {
  using namespace tavsr;
  TAVSR_REQUIRE(dp && gamma && beta && ln_out, TAVSR_EINVAL, "tavsr_gemm_ln: null pointer");
  const tavsr_gemm_desc& d = *dp;
  TAVSR_REQUIRE(d.nb1 * d.nb2 <= 1 && d.conv_mode == 0 && d.drop_p == 0.f && !d.Z && !d.DZ && !d.a_rowsum && !d.rowstat,
                TAVSR_EUNSUPPORTED, "tavsr_gemm_ln: a plain unbatched Linear (bias / activation / alpha / residual) only");
  TAVSR_REQUIRE(d.N % 4 == 0 && d.N <= kLnTailMaxN && d.ldc % 4 == 0 && ld_ln % 4 == 0 && (!d.R || d.ldr % 4 == 0) &&
                    aligned16(d.C) && aligned16(ln_out) && aligned16(gamma) && aligned16(beta) && ln_out != d.C,
                TAVSR_EALIGN, "tavsr_gemm_ln: N %% 4 == 0, N <= %d, 16-byte aligned rows of C / ln_out / gamma / beta", kLnTailMaxN);
  const LnTail ln{gamma, beta, eps, ln_out, ld_ln};
  g_ln_tail = &ln;
  const int rc = run(dp, -1, 0, static_cast<hipStream_t>(stream));
  g_ln_tail = nullptr;
  return rc;
}

extern "C" int tavsr_gemm_grouped(const tavsr_gemm_desc* descs, int32_t n, tavsr_stream_t stream) {
  using namespace tavsr;
  TAVSR_REQUIRE(descs != nullptr && n >= 1 && n <= kMaxGroup, TAVSR_EINVAL, "tavsr_gemm_grouped: 1..%d problems", kMaxGroup);
  GroupArgs g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    tavsr_gemm_desc d = descs[i];
    if (d.nb1 <= 0) d.nb1 = 1;
    if (d.nb2 <= 0) d.nb2 = 1;
    TAVSR_REQUIRE(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0, TAVSR_EINVAL, "tavsr_gemm_grouped: bad problem %d", i);
    TAVSR_REQUIRE(d.a_kmajor == descs[0].a_kmajor && d.b_kmajor == descs[0].b_kmajor, TAVSR_EUNSUPPORTED,
                  "tavsr_gemm_grouped: all problems must share one layout");
    TAVSR_REQUIRE(d.nb1 * d.nb2 == 1 && d.drop_p == 0.f, TAVSR_EUNSUPPORTED, "tavsr_gemm_grouped: unbatched problems without epilogue dropout only");
    const bool vec = aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0;
    TAVSR_REQUIRE(glds_ok(d, vec), TAVSR_EUNSUPPORTED,
                  "tavsr_gemm_grouped: problem %d needs the predicated kernel (alignment / K %% 32 / rows %% 4)", i);
    if (d.R == nullptr) { d.ldr = 0; d.sR1 = d.sR2 = 0; }
    g.d[i] = d;
    g.tile_start[i] = total;
    g.tiles_n[i] = cdiv(d.N, 64);
    total += cdiv(d.M, 64) * g.tiles_n[i];
  }
  for (int i = n; i <= kMaxGroup; ++i) g.tile_start[i] = total;
  g.vec_epi = 1;
  for (int i = 0; i < n; ++i) g.vec_epi &= (int)vec_epi_ok(g.d[i]);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = launch_layout(descs[0], [&](auto ak, auto bk) {
    hipLaunchKernelGGL((gemm_glds_grouped_kernel<64, 64, 2, 2, 2, 5, decltype(ak)::value, decltype(bk)::value>), dim3(total),
                       dim3(256), 0, s, g);
    TAVSR_LAUNCH_CHECK();
    return (int)TAVSR_OK;
  });
  return rc;
}

extern "C" int tavsr_gemm_tune(const tavsr_gemm_desc* dp, int32_t cfg, int32_t nsplit, tavsr_stream_t stream) {
  return tavsr::run(dp, cfg, nsplit, static_cast<hipStream_t>(stream));
}

// Workspace the planner would like for this problem: floats of split-K slabs (0: no split) and, through
// *sync_ints, the number of zero-initialised int32 tile counters.
extern "C" int64_t tavsr_gemm_ws(const tavsr_gemm_desc* dp) {
  using namespace tavsr;
  if (!dp) return 0;
  tavsr_gemm_desc d = *dp;
  if (d.nb1 <= 0) d.nb1 = 1;
  if (d.nb2 <= 0) d.nb2 = 1;
  if (d.M <= 0 || d.N <= 0) return 0;
  const bool vec = aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.sA1 % 4 == 0 &&
                   d.sA2 % 4 == 0 && d.sB1 % 4 == 0 && d.sB2 % 4 == 0;
  Plan p = d.conv_mode != 0 ? plan_conv(d, true) : plan(d, true, glds_ok(d, vec) || tail_ok(d, vec));
  return ws_floats_for(d, p.nsplit);
}

#ifdef TAVSR_GEMM_TRACE
// Debug build only: copy out (and reset) the per-workgroup phase timestamps. out: [max_rows][6] uint64. Returns rows.
extern "C" int tavsr_gemm_trace_read(unsigned long long* out, int max_rows) {
  unsigned int n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(tavsr::g_trace_n), sizeof(n)) != hipSuccess) return -1;
  int rows = (int)(n < (unsigned)tavsr::kTraceMax ? n : tavsr::kTraceMax);
  if (rows > max_rows) rows = max_rows;
  if (rows > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(tavsr::g_trace), (size_t)rows * 6 * sizeof(unsigned long long)) != hipSuccess) return -1;
  n = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(tavsr::g_trace_n), &n, sizeof(n)) != hipSuccess) return -1;
  return rows;
}
#endif
