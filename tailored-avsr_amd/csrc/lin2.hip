// Linear layers with a d_model = 256 input as ONE streaming launch for gfx950: y_s = act(x W_s^T + b_s) for up to four weight
// matrices of one input (espnet's query / key / value projections of RelPositionMultiHeadedAttention.forward_qkv and cgMLP's
// channel_proj1 + GELU, called at src/encoder/branchformer/encoder_layer.py:196-222 and
// src/encoder/audiovisual/tailored/encoder_layer.py:185-208).
//
// Why not the tiled GEMM: a 64x64 tile needs 16 bytes of operands per clock and CU at the full fp32 MFMA rate and a CU's
// vector-memory path delivers about 7.5 while its MFMA pipe is busy (profiles/r03_notes.md): with K = 256 these launches sit
// at 30-66 TFLOP/s.  This is the first product of the streaming feed-forward kernel (ffn2.hip) on its own: a workgroup owns
// 128 rows, wave w keeps its 32 rows x 256 in registers as the B operand for the whole launch, and the four waves share every
// weight tile (32 output columns x 256) through an LDS ring of 16 KB stages fed by LDS-DMA - 4 bytes per clock and CU.
// The accumulator tile (output column on the registers, row on the lane) gets bias, activation and its 16-byte stores in
// the shadow of the NEXT unit's MFMAs.  v_mfma_f32_32x32x2_f32: exact fp32.  512 registers per lane (one wave per SIMD).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace tavsr {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_f;
typedef const __attribute__((address_space(1))) float glb_f;

constexpr int kMaxU = 32;          // units per workgroup (bias slices staged in LDS)
constexpr int kRB = 128;           // rows per row block (4 waves x 32)
constexpr int kMaxSeg = 4;

struct Lin2Args {
  int M, G, UPR, wpb, nseg;
  const float* x;                  // [M][ldx]
  long ldx;
  const float* W[kMaxSeg];         // [n_s][256]
  const float* b[kMaxSeg];         // [n_s] or null
  float* out[kMaxSeg];             // [M][ldo_s]
  long ldo[kMaxSeg];
  float* z[kMaxSeg];               // pre-activations (null: not kept)
  long ldz[kMaxSeg];
  int u0[kMaxSeg + 1];             // first unit of segment s (prefix sums of n_s / 32)
};

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }
template <int N>
__device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() {      // LDS-only barrier: LDS-DMA stays in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#define SB() __builtin_amdgcn_sched_barrier(0)
// (LDS reads through a __restrict__ helper: see ffn2.hip - keeps hipcc from draining the LDS-DMA stream in front of them)
__device__ __forceinline__ float4 lds4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// SAVEZ: the pre-activations are written too (4 more stores per unit).
template <bool SAVEZ, int ACT>
__global__ __launch_bounds__(256, 1) void lin2_fwd_kernel(const Lin2Args a) {
  constexpr int NS = 4, STG = 4096;               // ring: four stages of 16 KB; a unit (32 output columns) = 2 stages
  constexpr int ST = SAVEZ ? 8 : 4;               // stores of a unit's deferred epilogue (per lane)
  constexpr int XSF = 2 * 4 * 4096;               // x staging [2 halves][4 waves][32 rows][128 floats], overlaid by ring slots 1 .. 3
  __shared__ __attribute__((aligned(1024))) float smem[STG + XSF + kMaxU * 32];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  float* const ring = smem;
  float* const xs = smem + STG + w * 4096;        // + half * 4 * 4096
  float* const bs = smem + STG + XSF;

  const int UPR = a.UPR;
  const int rb = blockIdx.x / a.wpb, part = blockIdx.x - rb * a.wpb;
  const int ht0 = part * UPR / a.wpb;
  const int nu = (part + 1) * UPR / a.wpb - ht0;
  if (nu <= 0) return;
  const int m0 = rb * kRB + 32 * w;               // this wave's row tile

  auto seg_of = [&](int u) {
    int s = 0;
#pragma unroll
    for (int j = 1; j < kMaxSeg; ++j) s += (j < a.nseg && u >= a.u0[j]) ? 1 : 0;
    return s;
  };
  // bias slices of this workgroup's units -> LDS (ordinary loads first: nothing waits behind the LDS-DMA queue for them)
  float bpre[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + 256 * it, u = ht0 + (e >> 5);
    float v = 0.f;
    if (e < nu * 32) {
      const int s = seg_of(u);
      if (a.b[s]) v = a.b[s][32 * (u - a.u0[s]) + (e & 31)];
    }
    bpre[it] = v;
  }
  auto issue_x = [&](int half) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = 2 * j + h2, pc = li;
      const float* src = a.x + (long)min(m0 + row, a.M - 1) * a.ldx + 128 * half + (((pc & ~15) | ((pc ^ row) & 15)) << 2);
      __builtin_amdgcn_global_load_lds((glb_f*)src, (lds_f*)(xs + half * (4 * 4096) + j * 256), 16, 0, 0);
    }
  };
  issue_x(0);
  issue_x(1);

  // weight stage s of a unit = W[32 u .. +32][128 s .. +128] as a [32 rows][128 floats] image (16-byte chunk c of row r at
  // chunk (c & ~15) | ((c ^ r) & 15)); a wave brings a quarter: 4 LDS-DMA instructions of 1 KB
  int offW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (4 * w + j) * 64 + lane, row = p >> 5, pc = p & 31;
    offW[j] = row * 256 + (((pc & ~15) | ((pc ^ row) & 15)) << 2);
  }
  // weight rows of the units iu, iu + 1, iu + 2 (clamped to the range's last unit)
  const float* wu[3];
  auto set_wu = [&](int iu_) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int u = ht0 + min(iu_ + d, nu - 1), s = seg_of(u);
      wu[d] = a.W[s] + (long)(32 * (u - a.u0[s])) * 256;
    }
  };
  auto issue_one = [&](int du, int sub, int slot, int j) {
    __builtin_amdgcn_global_load_lds((glb_f*)(wu[du] + 128 * sub + offW[j]), (lds_f*)(ring + slot * STG + w * 1024 + j * 256), 16, 0, 0);
  };
  set_wu(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) issue_one(0, 0, 0, j);            // stage 0 behind the rows; stages 1 .. 3 once the staging area is free
#pragma unroll
  for (int it = 0; it < 4; ++it) bs[tid + 256 * it] = bpre[it];

  // ---- the wave's 32 rows into the B-operand registers: lane (li, h2) holds row li, xr[4 g + j] = x[8 g + 4 h2 + j]
#define LIN2_A1(ST_, G) ((ST_) + li * 128 + ((((2 * (G) + h2) & ~15) | (((2 * (G) + h2) ^ li) & 15)) << 2))
  float xr[128];
  vmwait<0>();
#pragma unroll
  for (int g = 0; g < 32; ++g) {
    const float4 v = lds4(LIN2_A1(xs + (g >> 4) * (4 * 4096), g & 15));
    xr[4 * g] = v.x; xr[4 * g + 1] = v.y; xr[4 * g + 2] = v.z; xr[4 * g + 3] = v.w;
    if ((g & 3) == 3) SB();
  }
#undef LIN2_A1
  wg_barrier();             // rows in registers everywhere, own quarter of stage 0 landed (barrier 0 of the ring), bias slices visible
#pragma unroll
  for (int s_ = 1; s_ < NS; ++s_)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_one(s_ / 2, s_ % 2, s_, j);

  // Ring protocol as in ffn2.hip: stage k in slot k % 4; barrier k + 1 is passed two fragment groups before the end of stage k.
  // vmcnt (per wave, program order of unit i): [ST stores of unit i - 1 inside stage 0] | tail 0: D(2i+4) | tail 1: D(2i+5);
  // prologue: D(1), D(2), D(3).  Tail p waits for D(2i+p+1): the operations younger than it are 8 + ST (8 in the first unit).
  int cs = 0;
  float4 fa, fb, ga, gb;
#define LIN2_A1L(ST_, G) ((ST_) + lq * 128 + ((((2 * (G) + hq) & ~15) | (((2 * (G) + hq) ^ lq) & 15)) << 2))
  {
    int lq = li, hq = h2;
    ga = lds4(LIN2_A1L(ring, 0));
    gb = lds4(LIN2_A1L(ring, 1));
  }
  // A wave whose rows all lie beyond M issues no stores at all: its vmcnt bookkeeping is the store-free one (it still brings
  // its quarter of every stage, which the other waves read).
  const bool stores_on = m0 < a.M;
  f32x16 prev;                                    // finished tile of the previous unit, emitted in this unit's shadow
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
  int pu = ht0;                                   // its unit (valid when iu > 0)
  auto emit = [&](int q, int e, bool live) {      // element 4 q + e of `prev`: activation in place, 16-byte stores at e == 3
    const float zv = prev[4 * q + e];
    if (ACT != TAVSR_ACT_NONE) prev[4 * q + e] = act_fwd(ACT, zv);
    if (e == 3 && live && stores_on && m0 + li < a.M) {
      const int s = seg_of(pu);
      const long col = 32 * (pu - a.u0[s]) + 8 * q + 4 * h2;
      *reinterpret_cast<float4*>(a.out[s] + (long)(m0 + li) * a.ldo[s] + col) =
          make_float4(prev[4 * q], prev[4 * q + 1], prev[4 * q + 2], prev[4 * q + 3]);
    }
  };
  f32x16 prevz;                                   // SAVEZ: the same tile before the activation
  auto emit_z = [&](int q, bool live) {
    if (SAVEZ && live && stores_on && m0 + li < a.M) {
      const int s = seg_of(pu);
      *reinterpret_cast<float4*>(a.z[s] + (long)(m0 + li) * a.ldz[s] + 32 * (pu - a.u0[s]) + 8 * q + 4 * h2) =
            make_float4(prevz[4 * q], prevz[4 * q + 1], prevz[4 * q + 2], prevz[4 * q + 3]);
    }
  };

#define LIN2_TAIL(POS, MF7A, MF7B)                                                                                     \
        fa = ga; fb = gb;                                                                                               \
        if (iu == 0 || !stores_on) vmwait<8>(); else vmwait<8 + ST>();                                                                \
        wg_barrier();                                                                                                   \
        {                                                                                                               \
          const int cn = cs + 1 == NS ? 0 : cs + 1;                                                                     \
          const float* stn = ring + cn * STG;                                                                           \
          SB();                                                                                                         \
          MF7A                                                                                                          \
          ga = lds4(LIN2_A1L(stn, 0));                                                                                  \
          issue_one((POS + NS) / 2, (POS + NS) % 2, cs, 0);                                                             \
          issue_one((POS + NS) / 2, (POS + NS) % 2, cs, 1);                                                             \
          SB();                                                                                                         \
          MF7B                                                                                                          \
          gb = lds4(LIN2_A1L(stn, 1));                                                                                  \
          issue_one((POS + NS) / 2, (POS + NS) % 2, cs, 2);                                                             \
          issue_one((POS + NS) / 2, (POS + NS) % 2, cs, 3);                                                             \
          SB();                                                                                                         \
          cs = cn;                                                                                                      \
        }
#define LIN2_MF1(F, XO, G)                                                                                              \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.x, xr[XO + 4 * (G)], acc1, 0, 0, 0);                            \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.y, xr[XO + 4 * (G) + 1], acc1b, 0, 0, 0);                      \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.z, xr[XO + 4 * (G) + 2], acc1, 0, 0, 0);                        \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.w, xr[XO + 4 * (G) + 3], acc1b, 0, 0, 0);
#define LIN2_STAGE(POS, XO, EMIT)                                                                                       \
      {                                                                                                                 \
        const float* st = ring + cs * STG;                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                 \
          fa = ga; fb = gb;                                                                                             \
          ga = lds4(LIN2_A1L(st, 2 * p + 2));                                                                           \
          gb = lds4(LIN2_A1L(st, 2 * p + 3));                                                                           \
          LIN2_MF1(fa, XO, 2 * p)                                                                                       \
          LIN2_MF1(fb, XO, 2 * p + 1)                                                                                   \
          if (EMIT) {                                                                                                   \
            emit(p >> 1, 2 * (p & 1), live); emit(p >> 1, 2 * (p & 1) + 1, live);                                       \
            if (SAVEZ && (p & 1)) emit_z(p >> 1, live);                                                                 \
          }                                                                                                             \
          SB();                                                                                                         \
        }                                                                                                               \
        if (EMIT) { emit(3, 2, live); emit(3, 3, live); emit_z(3, live); }                                              \
        LIN2_TAIL(POS, LIN2_MF1(fa, XO, 14), LIN2_MF1(fb, XO, 15))                                                      \
      }

  int iu = 0;
  for (; iu < nu; ++iu) {
    set_wu(iu);
    int lq = li, hq = h2;                         // opaque lane coordinates: see ffn2.hip
    asm volatile("" : "+v"(lq), "+v"(hq));
    const bool live = iu > 0;
    f32x16 acc1, acc1b;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1b[r] = 0.f;
    {
      const float* bp = bs + iu * 32 + 4 * h2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = lds4(bp + 8 * q);
        acc1[4 * q] = b.x; acc1[4 * q + 1] = b.y; acc1[4 * q + 2] = b.z; acc1[4 * q + 3] = b.w;
      }
    }
    LIN2_STAGE(0, 0, true)
    LIN2_STAGE(1, 64, false)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      prev[r] = acc1[r] + acc1b[r];
      if (SAVEZ) prevz[r] = prev[r];
    }
    pu = ht0 + iu;
  }
#undef LIN2_STAGE
#undef LIN2_MF1
#undef LIN2_TAIL
#undef LIN2_A1L
  // the last unit's tile
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) emit(q, e, true);
    emit_z(q, true);
  }
  vmwait<0>();          // the stream ran ahead of the last unit: nothing may land in LDS after this workgroup has left
}

struct Plan {
  int G, wpb, UPR;
};

// Row blocks of 128 rows; a block's units are dealt to wpb workgroups: the smallest wpb that reaches the shortest longest range
// with about one workgroup per CU.  TAVSR_LIN2_WPB overrides (tuning runs).
Plan lin2_plan(int M, int units) {
  Plan p;
  p.UPR = units;
  const int nrb = cdiv(M, kRB);
  const int wmax = std::max(1, std::min(256 / nrb, units));
  const int best = cdiv(units, wmax);
  int wpb = wmax;
  while (wpb > 1 && cdiv(units, wpb - 1) == best) --wpb;
  if (const char* e = getenv("TAVSR_LIN2_WPB")) {
    const int g = atoi(e);
    if (g > 0) wpb = std::min(g, units);
  }
  wpb = std::max(wpb, cdiv(units, kMaxU - 1));
  p.wpb = wpb;
  p.G = nrb * wpb;
  return p;
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int ACT>
void launch(const Lin2Args& a, bool savez, hipStream_t s) {
  if (savez) hipLaunchKernelGGL((lin2_fwd_kernel<true, ACT>), dim3(a.G), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lin2_fwd_kernel<false, ACT>), dim3(a.G), dim3(256), 0, s, a);
}

}  // namespace
}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_lin2_fwd(const float* x, int64_t ldx, int32_t M, int32_t K, const tavsr_lin2_seg* segs, int32_t nseg, int32_t act,
                              tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && segs && nseg >= 1 && nseg <= kMaxSeg, TAVSR_EINVAL, "lin2_fwd: 1 .. %d weight matrices", kMaxSeg);
  TAVSR_REQUIRE(M > 0 && K == 256, TAVSR_EUNSUPPORTED, "lin2_fwd: input width 256 only (got %d)", K);
  TAVSR_REQUIRE(act == TAVSR_ACT_NONE || act == TAVSR_ACT_RELU || act == TAVSR_ACT_SWISH || act == TAVSR_ACT_GELU, TAVSR_EUNSUPPORTED,
                "lin2_fwd: unknown activation");
  TAVSR_REQUIRE(al16(x) && ldx % 4 == 0, TAVSR_EALIGN, "lin2_fwd: x rows must be 16-byte aligned");
  Lin2Args a{};
  a.M = M; a.x = x; a.ldx = ldx; a.nseg = nseg;
  int units = 0;
  bool savez = false;
  for (int s = 0; s < nseg; ++s) {
    const tavsr_lin2_seg& g = segs[s];
    TAVSR_REQUIRE(g.w && g.out && g.n > 0 && g.n % 32 == 0, TAVSR_EUNSUPPORTED, "lin2_fwd: matrix %d: rows must be a multiple of 32", s);
    TAVSR_REQUIRE(al16(g.w) && al16(g.out) && g.ldo % 4 == 0 && (!g.z || (al16(g.z) && g.ldz % 4 == 0)) && (!g.b || al16(g.b)), TAVSR_EALIGN,
                  "lin2_fwd: matrix %d: operands must be 16-byte aligned", s);
    a.W[s] = g.w; a.b[s] = g.b; a.out[s] = g.out; a.ldo[s] = g.ldo; a.z[s] = g.z; a.ldz[s] = g.ldz;
    a.u0[s] = units;
    units += g.n / 32;
    savez = savez || g.z != nullptr;
  }
  for (int s = 0; s < nseg; ++s)
    TAVSR_REQUIRE(!savez || segs[s].z, TAVSR_EINVAL, "lin2_fwd: pre-activations are kept for all matrices of a call or for none");
  a.u0[nseg] = units;
  const Plan p = lin2_plan(M, units);
  a.G = p.G; a.UPR = p.UPR; a.wpb = p.wpb;
  hipStream_t s = (hipStream_t)stream;
  switch (act) {
    case TAVSR_ACT_RELU: launch<TAVSR_ACT_RELU>(a, savez, s); break;
    case TAVSR_ACT_SWISH: launch<TAVSR_ACT_SWISH>(a, savez, s); break;
    case TAVSR_ACT_GELU: launch<TAVSR_ACT_GELU>(a, savez, s); break;
    default: launch<TAVSR_ACT_NONE>(a, savez, s); break;
  }
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
