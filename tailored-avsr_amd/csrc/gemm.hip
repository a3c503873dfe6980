// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32, exact f32 == k-ordered fmaf chain).
//
// One kernel family serves every contraction on the hot path:
//   NT  C[M,N] = A[M,K] * W[N,K]^T          torch Linear forward (espnet Linear leaves)
//   NN  C[M,N] = A[M,K] * B[K,N]            data gradients, attention P*V
//   TN  C[M,N] = A[K,M]^T * B[K,N]          weight gradients, dK/dV
// with two batch dimensions given by element strides (attention heads are addressed in place inside
// the [B*T, 3*256] QKV buffer: no transposes are ever materialised) and a fused epilogue
// (bias, ReLU/Swish/GELU, pre-activation store, residual + alpha, multiply by act'(z) for backward).
//
// Tiling: BM x BN block, WM x WN wavefronts (64 lanes), each wave owns TM x TN MFMA tiles of 32x32.
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K-step); the next
// K-step's global loads are issued before the MFMAs of the current one.  LDS images:
//   k-contiguous operand  : [row][BK+1]  (odd stride: the per-lane column read is conflict free)
//   row-contiguous operand: [k][rows+4]  (16-B aligned rows: ds_write_b128, row read conflict free)
#include <algorithm>

#include "common.h"

namespace tavsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  tavsr_gemm_desc d;
  float* ws;      // split-K slabs [nsplit][nb1*nb2][M][N]
  int kchunk;     // K elements per split (multiple of BK)
  int nsplit;
};

// Fixed-order sum of the split-K slabs + the fused epilogue (same math as the in-kernel one).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs args) {
  const tavsr_gemm_desc& d = args.d;
  const int64_t mn = (int64_t)d.M * d.N;
  const int nbatch = d.nb1 * d.nb2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  const int z = blockIdx.y, z1 = z / d.nb2, z2 = z % d.nb2;
  const int m = (int)(i / d.N), n = (int)(i % d.N);
  float v = 0.f;
  for (int s = 0; s < args.nsplit; ++s) v += args.ws[((int64_t)s * nbatch + z) * mn + i];
  if (d.bias) v += d.bias[n];
  const int64_t o = z1 * d.sC1 + z2 * d.sC2 + (int64_t)m * d.ldc + n;
  if (d.Z) d.Z[o] = v;
  v = act_fwd(d.act, v);
  if (d.DZ) v *= act_bwd(d.dact, d.DZ[o]);
  v *= d.alpha;
  if (d.R) v += d.R[z1 * d.sR1 + z2 * d.sR2 + (int64_t)m * d.ldr + n];
  d.C[o] = v;
}

template <int ROWS, int BK, bool KMAJOR>
struct Tile {
  static constexpr int LD = KMAJOR ? (ROWS + 4) : (BK + 1);
  static constexpr int SIZE = KMAJOR ? BK * (ROWS + 4) : ROWS * (BK + 1);
  __device__ static __forceinline__ int idx(int row, int k) { return KMAJOR ? k * LD + row : row * LD + k; }
};

// Load one ROWS x BK operand tile into registers (NV float4 per thread).
template <int ROWS, int BK, bool KMAJOR, bool VEC, int NT>
struct Loader {
  static constexpr int NV = ROWS * BK / 4 / NT;
  static_assert(ROWS * BK % (4 * NT) == 0, "tile must divide over the block");
  // vector v covers 4 consecutive elements along the contiguous direction
  __device__ static __forceinline__ void coords(int v, int& row, int& k) {
    if (KMAJOR) {
      k = v / (ROWS / 4);
      row = (v % (ROWS / 4)) * 4;
    } else {
      row = v / (BK / 4);
      k = (v % (BK / 4)) * 4;
    }
  }
  __device__ static __forceinline__ void load(const float* __restrict__ g, int64_t ld, int row0, int k0,
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
    using T = Tile<ROWS, BK, KMAJOR>;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, k;
      coords(tid + i * NT, row, k);
      if (KMAJOR) {
        *reinterpret_cast<float4*>(s + T::idx(row, k)) = r[i];
      } else {
        float* p = s + T::idx(row, k);
        p[0] = r[i].x;
        p[1] = r[i].y;
        p[2] = r[i].z;
        p[3] = r[i].w;
      }
    }
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool AK, bool BKM, bool VEC, bool SPLITK>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmArgs args) {
  const tavsr_gemm_desc& d = args.d;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using TA = Tile<BM, BK, AK>;
  using TB = Tile<BN, BK, BKM>;
  using LA = Loader<BM, BK, AK, VEC, NT>;
  using LB = Loader<BN, BK, BKM, VEC, NT>;
  __shared__ __attribute__((aligned(16))) float smem[2 * (TA::SIZE + TB::SIZE)];
  constexpr int STAGE = TA::SIZE + TB::SIZE;  // stage s: A at smem + s*STAGE, B right after it

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // tile index: x-fastest over N so consecutive blocks share the A panel
  const int tiles_n = (d.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM;
  const int n0 = (blockIdx.x % tiles_n) * BN;
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

  float4 ra[LA::NV], rb[LB::NV];
  // split-K: slice z covers k in [kbeg, kend); each slice writes a raw fp32 slab to the workspace
  const int kbeg = SPLITK ? blockIdx.z * args.kchunk : 0;
  const int kend = SPLITK ? min(d.K, kbeg + args.kchunk) : d.K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  LA::load(A, d.lda, m0, kbeg, d.M, kend, tid, ra);
  LB::load(B, d.ldb, n0, kbeg, d.N, kend, tid, rb);
  LA::store(smem, tid, ra);
  LB::store(smem + TA::SIZE, tid, rb);
  __syncthreads();

  const int lr = lane & 31, lk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      LA::load(A, d.lda, m0, kbeg + (kt + 1) * BK, d.M, kend, tid, ra);
      LB::load(B, d.ldb, n0, kbeg + (kt + 1) * BK, d.N, kend, tid, rb);
    }
    const float* a_s = smem + cur * STAGE;
    const float* b_s = a_s + TA::SIZE;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a_s[TA::idx(wm * TM * 32 + i * 32 + lr, kk + lk)];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b_s[TB::idx(wn * TN * 32 + j * 32 + lr, kk + lk)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      LA::store(smem + (cur ^ 1) * STAGE, tid, ra);
      LB::store(smem + (cur ^ 1) * STAGE + TA::SIZE, tid, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (SPLITK) {
    float* slab = args.ws + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * (int64_t)d.M * d.N;
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
    return;
  }
  float* C = d.C + coff;
  float* Z = d.Z ? d.Z + coff : nullptr;
  const float* R = d.R ? d.R + z1 * d.sR1 + z2 * d.sR2 : nullptr;
  const float* DZ = d.DZ ? d.DZ + coff : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + lr;
    if (n >= d.N) continue;
    const float bv = d.bias ? d.bias[n] : 0.f;
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

template <int BM, int BN, int BK, int WM, int WN>
static int launch_cfg(const tavsr_gemm_desc& d, bool vec, int nsplit, int kchunk, hipStream_t s) {
  GemmArgs a{d, d.ws, kchunk, nsplit};
  dim3 grid(cdiv(d.M, BM) * cdiv(d.N, BN), d.nb1 * d.nb2, nsplit);
  dim3 block(WM * WN * 64);
#define TAVSR_GEMM_LAUNCH(AK, BKM)                                                                        \
  do {                                                                                                    \
    if (nsplit > 1) {                                                                                     \
      if (vec)                                                                                            \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, AK, BKM, true, true>), grid, block, 0, s, a);  \
      else                                                                                                \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, AK, BKM, false, true>), grid, block, 0, s, a); \
    } else {                                                                                              \
      if (vec)                                                                                            \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, AK, BKM, true, false>), grid, block, 0, s, a); \
      else                                                                                                \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, AK, BKM, false, false>), grid, block, 0, s, a);\
    }                                                                                                     \
  } while (0)
  if (!d.a_kmajor && !d.b_kmajor) TAVSR_GEMM_LAUNCH(false, false);
  else if (!d.a_kmajor && d.b_kmajor) TAVSR_GEMM_LAUNCH(false, true);
  else if (d.a_kmajor && d.b_kmajor) TAVSR_GEMM_LAUNCH(true, true);
  else TAVSR_GEMM_LAUNCH(true, false);
#undef TAVSR_GEMM_LAUNCH
  TAVSR_LAUNCH_CHECK();
  if (nsplit > 1) {
    dim3 g2(cdiv((int64_t)d.M * d.N, 256), d.nb1 * d.nb2, 1);
    hipLaunchKernelGGL(splitk_epilogue_kernel, g2, dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
  }
  return TAVSR_OK;
}

// Split-K plan: fill the 256 CUs (>= 2 waves per SIMD) when the output has too few tiles but K is long
// (weight gradients: K = B*T rows).  Returns nsplit (1 = none) and the K chunk per slice.
static void plan_splitk(const tavsr_gemm_desc& d, int BM, int BN, int BK, int* nsplit, int* kchunk) {
  *nsplit = 1;
  *kchunk = d.K;
  const long tiles = (long)cdiv(d.M, BM) * cdiv(d.N, BN) * d.nb1 * d.nb2;
  if (tiles >= 384 || d.K < 8 * BK) return;
  long want = std::min<long>({(long)cdiv(768, tiles), (long)d.K / (4 * BK), 64L});
  if (want < 2) return;
  int kc = cdiv(cdiv(d.K, want), BK) * BK;
  *kchunk = kc;
  *nsplit = cdiv(d.K, kc);
}

static bool use_big_tile(const tavsr_gemm_desc& d) {
  return (long)cdiv(d.M, 128) * cdiv(d.N, 128) * d.nb1 * d.nb2 >= 768;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace tavsr

extern "C" int tavsr_gemm(const tavsr_gemm_desc* dp, tavsr_stream_t stream) {
  using namespace tavsr;
  TAVSR_REQUIRE(dp != nullptr, TAVSR_EINVAL, "tavsr_gemm: null descriptor");
  tavsr_gemm_desc d = *dp;
  TAVSR_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, TAVSR_EINVAL, "tavsr_gemm: negative dims");
  if (d.nb1 <= 0) d.nb1 = 1;
  if (d.nb2 <= 0) d.nb2 = 1;
  if (d.M == 0 || d.N == 0) return TAVSR_OK;
  TAVSR_REQUIRE(d.A && d.B && d.C, TAVSR_EINVAL, "tavsr_gemm: null operand");
  TAVSR_REQUIRE((long)d.nb1 * d.nb2 <= 65535, TAVSR_EINVAL, "tavsr_gemm: batch too large");
  if (d.R == nullptr) { d.ldr = 0; d.sR1 = d.sR2 = 0; }
  // vector (16-B) operand loads need aligned bases, leading dims and batch strides
  bool vec = aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.sA1 % 4 == 0 &&
             d.sA2 % 4 == 0 && d.sB1 % 4 == 0 && d.sB2 % 4 == 0;
  // (a float4 that straddles the end of the contiguous direction falls back to predicated scalar loads)
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (use_big_tile(d)) return launch_cfg<128, 128, 16, 2, 2>(d, vec, 1, d.K, s);
  int nsplit, kchunk;
  plan_splitk(d, 64, 64, 32, &nsplit, &kchunk);
  if (nsplit > 1) {
    const int64_t need = (int64_t)nsplit * d.nb1 * d.nb2 * d.M * d.N;
    if (d.ws == nullptr || d.ws_floats < need) nsplit = 1, kchunk = d.K;  // caller gave no workspace: plain path
  }
  return launch_cfg<64, 64, 32, 2, 2>(d, vec, nsplit, kchunk, s);
}

extern "C" int64_t tavsr_gemm_ws(const tavsr_gemm_desc* dp) {
  using namespace tavsr;
  if (!dp) return 0;
  tavsr_gemm_desc d = *dp;
  if (d.nb1 <= 0) d.nb1 = 1;
  if (d.nb2 <= 0) d.nb2 = 1;
  if (d.M <= 0 || d.N <= 0 || use_big_tile(d)) return 0;
  int nsplit, kchunk;
  plan_splitk(d, 64, 64, 32, &nsplit, &kchunk);
  return nsplit > 1 ? (int64_t)nsplit * d.nb1 * d.nb2 * d.M * d.N : 0;
}
