// Position-wise feed-forward block as ONE streaming kernel for gfx950, second design (espnet PositionwiseFeedForward inside
// its residual block: src/encoder/branchformer/encoder_layer.py:191-194,311-314; the tailored AV layer's shared FFNs,
// src/encoder/audiovisual/tailored/encoder_layer.py:173-175,211-213; d_model 256):
//     y = x + scale * dropout(W2 dropout(act(W1 LN(x) + b1)) + b2)       [+ LayerNorms of y for the consumers of y]
//
// What bounds this block on MI355X (measured, profiles/r03_ffn_notes.md): not the fp32 MFMA pipe but the CU's vector-memory
// path.  Any form that streams 16 KB of weights per 1 MFLOP - the 64x64-tile GEMM launches, the first chain kernel
// (ffn.hip), a K-split chain with per-wave LDS-DMA rings - ran the 6.6 GFLOP in 104-108 us (63 TFLOP/s, MFMA pipe 48 % busy)
// whatever the ring depth, the occupancy or the cache level the weights came from: a CU takes in 7-8 bytes per clock.
// So this design buys REUSE: a workgroup owns a block of 128 rows, wave w its 32-row tile w, and the four waves share every
// weight tile through one LDS ring - 4 bytes of weights per clock and CU at the full MFMA rate instead of 16.
//   * unit of work = (128 rows) x (32 hidden units) = 256 MFMAs per wave:
//       phase 1: z^T tile (32 hidden x 32 rows) = W1[unit] (A operand, from LDS) x LN(x)^T (B operand: the wave's row tile
//                lives in 128 registers for the whole row block), bias as the initial accumulator;
//       the accumulator tile - hidden unit on the registers, row on the lane - IS the A operand of phase 2 after the
//       activation (the MFMA sums over k in any order as long as A and B agree, and W2's fragment is read with the
//       accumulator's k permutation): the hidden activations never leave the registers, no exchange, no second barrier;
//       phase 2: out tile (32 rows x 256) += act(z) x W2[:, unit]^T into 8 accumulator tiles that live as long as the row block.
//   * the flat unit list (row blocks x hidden tiles) is cut into equal contiguous ranges, one workgroup per CU; a range that
//     crosses a row-block boundary writes two partial outputs and the finishing kernel adds a row block's partials in
//     workgroup order (deterministic, no atomics);
//   * weights stream through a ring of NS stages of 16 KB (W1: 32 hidden x 128 k; W2: 128 out x 32 hidden) by 16-byte
//     LDS-DMA, each wave bringing a quarter of every stage; one raw barrier per stage (64 MFMAs) publishes it, a counted
//     s_waitcnt vmcnt keeps the younger stages in flight across it.
// v_mfma_f32_32x32x2_f32: exact fp32 (the 1e-4 parity bar).  512 registers per lane (one wave per SIMD).
//
// The finishing kernel (one wave per row) adds the partials, bias, dropout, scale and residual, and can emit up to two
// LayerNorms of the result (norm_mha / norm_mlp after the macaron block, norm_final after the second block): the layer's
// stand-alone LayerNorm launches disappear.
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "common.h"

namespace tavsr {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_f;
typedef const __attribute__((address_space(1))) float glb_f;

constexpr int kMaxU = 32;          // units per workgroup (bias slices staged in LDS)
constexpr int kRB = 128;           // rows per row block (4 waves x 32)

struct Ffn2Args {
  int dbg;                         // tuning runs only (TAVSR_FFN2_DBG): bit 0 = every unit streams hidden tile 0's weights
  int M, N1, G, UPR, wpb, act;     // rows, hidden units, workgroups, units per row block (N1 / 32), workgroups per row block
  const float* x;                  // [M][ldx]
  long ldx;
  const float *ln_w, *ln_b, *W1, *b1, *W2;
  float eps, alpha;              // alpha: backward only (scale of dyd)
  float* slab;                     // [G][128][256]
  float *n_out, *mean, *rstd;      // saved LayerNorm output / statistics (null: not kept)
  float *Z, *H;                    // [roundup128(M)][N1] (SAVE)
  uint32_t thr;                    // inner dropout: element (m, c) = word c & 3 of counter offset4 + (m * N1 + c) / 4
  float inv_keep;
  const uint64_t* seed;
  uint64_t offset4;
};

// Tuning runs only (TAVSR_FFN2_DBG bit 1): per-workgroup time stamps (100 MHz wall clock) of the kernel's phases.
constexpr int kTraceWG = 1024, kTraceN = 16;     // [0..7] 100 MHz wall clock, [8..15] shader clock (s_memtime)
__device__ unsigned long long g_ffn2_trace[kTraceWG * kTraceN];
__device__ __forceinline__ void stamp(const Ffn2Args& a, int i) {
  if ((a.dbg & 2) && threadIdx.x == 0 && blockIdx.x < kTraceWG) {
    g_ffn2_trace[blockIdx.x * kTraceN + i] = wall_clock64();
    g_ffn2_trace[blockIdx.x * kTraceN + 8 + i] = __builtin_readcyclecounter();
  }
}

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }
// Activation inside the chain: hardware exp2 / reciprocal (1 ulp each) instead of expf and an IEEE division - 6 instructions
// per element instead of ~35, which matters for a wave that is alone on its SIMD: what it issues between two MFMAs beyond
// ~64 cycles delays the second one.  |error| ~ 2e-7 relative, far inside the 1e-4 parity bar.
template <int ACT>
__device__ __forceinline__ float act_fast(float z) {
  if (ACT == TAVSR_ACT_SWISH) return z * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
  return z > 0.f ? z : 0.f;
}
template <int N>
__device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() {      // LDS-only barrier: LDS-DMA stays in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#define SB() __builtin_amdgcn_sched_barrier(0)

// Every LDS read of the steady-state loop goes through a function with a __restrict__ pointer: the alias scope this gives
// the load is what keeps hipcc from waiting vmcnt(0) - draining the whole LDS-DMA weight stream - in front of it (without
// scope information its waitcnt pass assumes that any LDS read may alias any LDS-DMA in flight).
__device__ __forceinline__ float4 lds4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// NS: stages (16 KB) of the weight ring.  SAVE: z / h (and the LayerNorm output) are written for the backward pass.
template <int NS, bool SAVE, bool DROP, int ACT>
__global__ __launch_bounds__(256, 1) void ffn2_fwd_kernel(const Ffn2Args a) {
  constexpr int STG = 4096;                       // floats per stage
  constexpr int NVM = (NS - 2) * 4;               // a wave's LDS-DMA instructions that may stay in flight behind the awaited stage
  static_assert(NS >= 3 && NS <= 5, "one stage being read, one about to be, at least one in flight; NS * 16 KB + 84 KB of LDS");
  // LDS: ring slot 0 | x staging [2 halves][4 waves][32 rows][128 floats] = 128 KB, which ring slots 1 .. NS-1 overlay once the
  // rows are in registers | bias slices | LayerNorm weights
  constexpr int XSF = 2 * 4 * 4096;
  static_assert((NS - 1) * STG <= XSF, "ring slots 1.. live in the x staging area");
  __shared__ __attribute__((aligned(1024))) float smem[STG + XSF + kMaxU * 32 + 512];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  float* const ring = smem;
  float* const xs = smem + STG + w * 4096;           // + half * 4 * 4096
  float* const b1s = smem + STG + XSF;
  float* const gbs = b1s + kMaxU * 32;            // LayerNorm gamma | beta

  // Workgroup g = row block g / wpb, part g % wpb of that block's hidden tiles: ranges never cross a row block, so a
  // workgroup pays ONE LayerNorm prologue and writes ONE partial output.
  const int UPR = a.UPR;
  const int rb = blockIdx.x / a.wpb, part = blockIdx.x - rb * a.wpb;
  const int ht0 = part * UPR / a.wpb;
  const int nu = (part + 1) * UPR / a.wpb - ht0;
  if (nu <= 0) return;
  stamp(a, 0);
  const int m0 = rb * kRB + 32 * w;               // this wave's row tile

  // ---- x, first half of the wave's rows (k 0 .. 127): LDS-DMA of whole 512-byte half rows into the staging image
  // [32 rows][128 floats] (16-byte chunk c of row r at chunk (c & ~15) | ((c ^ r) & 15)), ahead of everything else
  // (ordinary loads first, so that nothing waits behind the LDS-DMA queue for them: bias slices of this workgroup's units
  // and the LayerNorm weights go to LDS - an ordinary load inside the loop would drain the weight stream)
  float bpre[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) bpre[it] = tid + 256 * it < nu * 32 ? a.b1[32 * ht0 + tid + 256 * it] : 0.f;
  const float gpre = a.ln_w[tid], cpre = a.ln_b[tid];
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

  // ---- weight stream.  Stages of a unit: 0, 1 = W1[32 ih .. +32][128 s .. +128] as a [32 rows][128 floats] image (16-byte
  // chunk c of row r at chunk (c & ~15) | ((c ^ r) & 15)); 2, 3 = W2[128 (s-2) .. +128][32 ih .. +32] as four [32 rows][32 floats]
  // images (chunk c of row r at c ^ ((r >> 1) & 7)).  A wave brings a quarter of every stage: 4 LDS-DMA instructions of 1 KB.
  int offW1[4], offW2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (4 * w + j) * 64 + lane, row = p >> 5, pc = p & 31;
    offW1[j] = row * 256 + (((pc & ~15) | ((pc ^ row) & 15)) << 2);
    const int q = j * 64 + lane, r2 = q >> 3;
    offW2[j] = (32 * w + r2) * a.N1 + (((q & 7) ^ ((r2 >> 1) & 7)) << 2);
  }
  // hidden tiles of the units iu, iu + 1, iu + 2 (clamped to the range's last unit: past the end the stream fetches valid addresses
  // whose data is never used)
  int hta[3];
  auto set_hta = [&](int iu_) {
#pragma unroll
    for (int d = 0; d < 3; ++d) hta[d] = (a.dbg & 1) ? 0 : ht0 + min(iu_ + d, nu - 1);
  };
  // LDS-DMA instruction j of this wave's quarter of stage `sub` of unit iu + du -> ring slot (sub, du: compile-time at every
  // call site)
  auto issue_one = [&](int du, int sub, int slot, int j) {
    const int ih = hta[du];
    const float* src = sub < 2 ? a.W1 + ((long)(32 * ih) * 256 + 128 * sub) + offW1[j]
                               : a.W2 + ((long)(128 * (sub - 2)) * a.N1 + 32 * ih) + offW2[j];
    __builtin_amdgcn_global_load_lds((glb_f*)src, (lds_f*)(ring + slot * STG + w * 1024 + j * 256), 16, 0, 0);
  };

  set_hta(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) issue_one(0, 0, 0, j);            // stage 0 behind the rows; stages 1 .. NS-1 once the staging area is free
#pragma unroll
  for (int it = 0; it < 4; ++it) b1s[tid + 256 * it] = bpre[it];       // kMaxU * 32 = 1024 floats
  gbs[tid] = gpre;
  gbs[256 + tid] = cpre;

  const uint64_t sd = DROP ? a.seed[0] : 0;
  // ---- LayerNorm of the wave's 32 rows into the B-operand registers: lane (li, h2) holds row li,
  // xr[4 g + j] = LN(x)[8 g + 4 h2 + j] (the k order of the weight fragments); statistics over the lane pair (li, 0), (li, 1)
#define FFN2_A1(ST, G) ((ST) + li * 128 + ((((2 * (G) + h2) & ~15) | (((2 * (G) + h2) ^ li) & 15)) << 2))
  float xr[128];
  {
    // three passes over the staged rows (sum; squared deviations; normalise): the values are read from LDS each time instead of
    // being kept - 128 live VALU registers in the middle of the prologue push the loop's invariants out to scratch
    stamp(a, 1);           // everything issued
    vmwait<0>();                                  // both halves of the rows (and this wave's quarter of stage 0) have landed
    stamp(a, 5);           // rows landed
    float s1 = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float4 v = lds4(FFN2_A1(xs, g)), u = lds4(FFN2_A1(xs + 4 * 4096, g));
      s1 += ((v.x + v.y) + (v.z + v.w)) + ((u.x + u.y) + (u.z + u.w));
    }
    const float mu = (s1 + __shfl_xor(s1, 32, 64)) * (1.f / 256.f);
    float s2 = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float4 v = lds4(FFN2_A1(xs, g)), u = lds4(FFN2_A1(xs + 4 * 4096, g));
      const float a0 = v.x - mu, a1 = v.y - mu, a2 = v.z - mu, a3 = v.w - mu, c0 = u.x - mu, c1 = u.y - mu, c2 = u.z - mu, c3 = u.w - mu;
      s2 += ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3));
    }
    const float rs = rsqrtf((s2 + __shfl_xor(s2, 32, 64)) * (1.f / 256.f) + a.eps);
    const int m = min(m0 + li, a.M - 1);
    // the workgroup that holds the row block's first unit keeps LN(x) (a run-time test in every variant: with the store
    // compiled out hipcc allocates the whole kernel differently and spills 40 loop invariants)
    const bool keepn = a.n_out != nullptr && part == 0 && m0 + li < a.M;
    // (gamma / beta were written to LDS by all threads before this point only in program order: make them visible)
    wg_barrier();
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const float4 v = lds4(FFN2_A1(xs + (g >> 4) * (4 * 4096), g & 15));
      const float4 gg = lds4(gbs + 8 * g + 4 * h2), bb = lds4(gbs + 256 + 8 * g + 4 * h2);
      xr[4 * g] = (v.x - mu) * rs * gg.x + bb.x;
      xr[4 * g + 1] = (v.y - mu) * rs * gg.y + bb.y;
      xr[4 * g + 2] = (v.z - mu) * rs * gg.z + bb.z;
      xr[4 * g + 3] = (v.w - mu) * rs * gg.w + bb.w;
      if (keepn)
        *reinterpret_cast<float4*>(a.n_out + (long)m * 256 + 8 * g + 4 * h2) = make_float4(xr[4 * g], xr[4 * g + 1], xr[4 * g + 2], xr[4 * g + 3]);
      if ((g & 3) == 3) SB();       // a few loads in flight at a time: hoisting all 96 of them costs 384 registers
    }
    // everybody holds its rows in registers and has waited for its quarter of stage 0 (barrier 0 of the ring); the staging
    // area is free: stages 1 .. NS-1 go out over it
    wg_barrier();
#pragma unroll
    for (int s_ = 1; s_ < NS; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_one(s_ / 4, s_ % 4, s_, j);
    if (keepn && a.mean && h2 == 0) { a.mean[m] = mu; a.rstd[m] = rs; }
  }
  // Ring protocol.  Stage k lives in slot k % NS.  "Barrier k" = every wave has waited for its own quarter of stage k
  // (counted vmcnt) and holds in registers everything it will still read of stage k - 1; behind it stage k may be read and
  // slot (k - 1) % NS is refilled with stage k - 1 + NS.  Barrier k + 1 is passed two fragment groups BEFORE the end of stage
  // k: the first fragment of stage k + 1 and the four LDS-DMA instructions are then issued in the shadow of the last eight
  // MFMAs of stage k, and no MFMA waits at a stage boundary.  (Barrier 0 is the one above.)
  int cs = 0;                                     // ring slot of the stage whose fragments are read
  float4 fa, fb, ga, gb;                          // fragments of the pair of groups being multiplied / of the next pair
  ga = lds4(FFN2_A1(ring, 0));
  gb = lds4(FFN2_A1(ring, 1));
  int iu = 0, ht = ht0;
  stamp(a, 2);
  {
    f32x16 acc2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    // fragment addresses: W1 stage (group g: k = 8 g + 4 h2 ..), W2 stage (group i: q = i >> 2, output tile i & 3)
#define FFN2_A2(ST, I) ((ST) + ((I) & 3) * 1024 + lq * 32 + (((2 * ((I) >> 2) + hq) ^ ((lq >> 1) & 7)) << 2))
#define FFN2_A1L(ST, G) ((ST) + lq * 128 + ((((2 * (G) + hq) & ~15) | (((2 * (G) + hq) ^ lq) & 15)) << 2))
    // A stage = 8 pairs of fragment groups; the fragments of pair p + 1 are read while pair p multiplies (8 MFMAs).
    // Consecutive MFMAs never share an accumulator (two chains side by side): a dependent f32 MFMA does not issue back to back.
    // Tail of a stage (pair 7): barrier POS + 1, then - in the shadow of the pair's MFMAs - the first two fragments of the
    // next stage and the refill of this stage's slot.  NEXTW1: the next stage is a W1 stage.
#define FFN2_TAIL(POS, NWAIT, NEXTW1, MF7A, MF7B)                                                                       \
        fa = ga; fb = gb;                                                                                               \
        vmwait<NWAIT>();                                                                                                \
        wg_barrier();                                                                                                   \
        {                                                                                                               \
          const int cn = cs + 1 == NS ? 0 : cs + 1;                                                                     \
          const float* stn = ring + cn * STG;                                                                           \
          SB();                                                                                                         \
          MF7A                                                                                                          \
          ga = NEXTW1 ? lds4(FFN2_A1L(stn, 0)) : lds4(FFN2_A2(stn, 0));                                                  \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 0);                                                             \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 1);                                                             \
          SB();                                                                                                         \
          MF7B                                                                                                          \
          gb = NEXTW1 ? lds4(FFN2_A1L(stn, 1)) : lds4(FFN2_A2(stn, 1));                                                  \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 2);                                                             \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 3);                                                             \
          SB();                                                                                                         \
          cs = cn;                                                                                                      \
        }

    for (; iu < nu; ++iu, ++ht) {
      set_hta(iu);
      // lane coordinates the compiler cannot see through: the ~40 fragment addresses of a unit are then recomputed in the
      // MFMAs' shadow instead of being hoisted out of the loop, where they cost 40 registers and - the LayerNorm prologue
      // peaks at 128 + registers - a spill round trip each (6 us of serial scratch reloads per workgroup)
      int lq = li, hq = h2;
      asm volatile("" : "+v"(lq), "+v"(hq));
      // ---- phase 1: z^T tile = W1[unit] x LN(x)^T, hidden unit on the registers (j = rho(r) + 4 h2), row on the lane
      f32x16 acc1, acc1b;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1b[r] = 0.f;
      {
        const float* bp = b1s + iu * 32 + 4 * h2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = lds4(bp + 8 * q);
          acc1[4 * q] = b.x; acc1[4 * q + 1] = b.y; acc1[4 * q + 2] = b.z; acc1[4 * q + 3] = b.w;
        }
      }
#define FFN2_MF1(F, XO, G)                                                                                              \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.x, xr[XO + 4 * (G)], acc1, 0, 0, 0);                            \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.y, xr[XO + 4 * (G) + 1], acc1b, 0, 0, 0);                      \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.z, xr[XO + 4 * (G) + 2], acc1, 0, 0, 0);                        \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.w, xr[XO + 4 * (G) + 3], acc1b, 0, 0, 0);
#define FFN2_P1_STAGE(POS, XO, NEXTW1)                                                                                  \
      {                                                                                                                 \
        const float* st = ring + cs * STG;                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                 \
          fa = ga; fb = gb;                                                                                             \
          ga = lds4(FFN2_A1L(st, 2 * p + 2));                                                                            \
          gb = lds4(FFN2_A1L(st, 2 * p + 3));                                                                            \
          FFN2_MF1(fa, XO, 2 * p)                                                                                       \
          FFN2_MF1(fb, XO, 2 * p + 1)                                                                                   \
          SB();                                                                                                         \
        }                                                                                                               \
        FFN2_TAIL(POS, NVM, NEXTW1, FFN2_MF1(fa, XO, 14), FFN2_MF1(fb, XO, 15))                                         \
      }
      FFN2_P1_STAGE(0, 0, true)
      FFN2_P1_STAGE(1, 64, false)
#undef FFN2_P1_STAGE
#undef FFN2_MF1
      // ---- activation (+ inner dropout) in place: av[4 q + e] = h[row li][hidden 8 q + 4 h2 + e] - the A operand of phase 2.
      // Chunk 0 here; the elements of chunks 1 .. 3 one per fragment group, in the shadow of phase 2's MFMAs (q-major order).
      float av[16];
      uint32_t wv[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      auto activate = [&](int q, int e) {
        const long e0 = (long)(m0 + li) * a.N1 + 32 * ht + 8 * q + 4 * h2;       // 4 consecutive hidden units of one row
        if (DROP && e == 0) {
          const uint64_t ctr = a.offset4 + ((uint64_t)e0 >> 2);
          philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), wv);
        }
        av[4 * q + e] = act_fast<ACT>(acc1[4 * q + e]) * ((!DROP || wv[e] >= a.thr) ? a.inv_keep : 0.f);
        if (SAVE && e == 3) {     // rows >= M exist in the buffers (roundup128(M) rows): unconditional 16-byte stores
          *reinterpret_cast<float4*>(a.Z + e0) = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
          *reinterpret_cast<float4*>(a.H + e0) = make_float4(av[4 * q], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]);
        }
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] += acc1b[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) activate(0, e);
      // ---- phase 2: out tile += h x W2[:, unit]^T, 4 + 4 output tiles of 32 columns; q-major so that a chunk of h is used
      // for 16 MFMAs in a row.  (SAVE: the 8 stores of a unit are all issued inside stage 2, before its tail barrier.)
      // pair P = groups 2 P, 2 P + 1: the same chunk q = P >> 1 of h against output tiles nt, nt + 1 (nt = 2 (P & 1))
#define FFN2_M2(AVI, X, Y, T0)                                                                                          \
          acc2[T0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[AVI], X, acc2[T0], 0, 0, 0);                               \
          acc2[(T0) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[AVI], Y, acc2[(T0) + 1], 0, 0, 0);
#define FFN2_MF2A(FA, FB, NT0, P)                                                                                       \
          FFN2_M2(4 * ((P) >> 1), FA.x, FB.x, NT0 + 2 * ((P) & 1))                                                      \
          FFN2_M2(4 * ((P) >> 1) + 1, FA.y, FB.y, NT0 + 2 * ((P) & 1))
#define FFN2_MF2B(FA, FB, NT0, P)                                                                                       \
          FFN2_M2(4 * ((P) >> 1) + 2, FA.z, FB.z, NT0 + 2 * ((P) & 1))                                                  \
          FFN2_M2(4 * ((P) >> 1) + 3, FA.w, FB.w, NT0 + 2 * ((P) & 1))
#define FFN2_P2_STAGE(POS, NT0, ACTIVATE, NWAIT, NEXTW1)                                                                \
      {                                                                                                                 \
        const float* st = ring + cs * STG;                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                 \
          fa = ga; fb = gb;                                                                                             \
          ga = lds4(FFN2_A2(st, 2 * p + 2));                                                                            \
          gb = lds4(FFN2_A2(st, 2 * p + 3));                                                                            \
          FFN2_MF2A(fa, fb, NT0, p)                                                                                     \
          FFN2_MF2B(fa, fb, NT0, p)                                                                                     \
          if (ACTIVATE && p < 6) { activate((p >> 1) + 1, 2 * (p & 1)); activate((p >> 1) + 1, 2 * (p & 1) + 1); }      \
          SB();                                                                                                         \
        }                                                                                                               \
        FFN2_TAIL(POS, NWAIT, NEXTW1, FFN2_MF2A(fa, fb, NT0, 7), FFN2_MF2B(fa, fb, NT0, 7))                             \
      }
      FFN2_P2_STAGE(2, 0, true, NVM + (SAVE ? 8 : 0), false)
      FFN2_P2_STAGE(3, 4, false, NVM + (SAVE ? 8 : 0), true)
#undef FFN2_P2_STAGE
#undef FFN2_MF2A
#undef FFN2_MF2B
#undef FFN2_M2
    }
#undef FFN2_TAIL
#undef FFN2_A1
#undef FFN2_A2
#undef FFN2_A1L
    stamp(a, 3);
    if (threadIdx.x == 0 && (a.dbg & 2) && blockIdx.x < kTraceWG) g_ffn2_trace[blockIdx.x * kTraceN + 7] = nu;
    // ---- partial output of this (row block, unit range) -> slab slot
    {
      float* out = a.slab + ((long)blockIdx.x * kRB + 32 * w) * 256;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(rho(r) + 4 * h2) * 256 + 32 * t + li] = acc2[t][r];
    }
    stamp(a, 4);      // epilogue's stores issued
  }
  vmwait<0>();
  stamp(a, 6);          // the stream ran ahead of the last unit: nothing may land in LDS after this workgroup has left
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the block w.r.t. its activations, same structure (espnet's autograd of the block above):
//     dz = ((alpha * dyd) W2) * mask / keep * act'(z)   [M][N1]   (operand of W1's weight gradient)
//     dn = dz W1                                        [M][256]  (gradient w.r.t. LN(x))
// A workgroup holds alpha * dyd of its 128 rows in registers; unit = 32 hidden units:
//   phase 1: dh^T tile (32 hidden x 32 rows) = W2[:, unit]^T x dyd^T.  W2 is [256][N1]: a stage is the k-major image
//            [128 k][32 hidden], a fragment value is one float per lane (lanes = hidden units, consecutive banks);
//   the tile times act'(z) and the inner mask (z tile: 4 LDS-DMA instructions per wave and unit into a private double
//   buffer, one unit ahead) is dz - written out once - and the A operand of
//   phase 2: dn tile (32 rows x 256) += dz x W1[unit, :].  W1 is [N1][256]: a stage is [32 hidden][128 columns], again one
//            float per lane (lanes = output columns).
// No transposed weight copies (the first chain kernel, ffn.hip, needs two per call).  Four ring stages.
// vmcnt bookkeeping (per wave, program order of one unit i; D(k) = this wave's 4 LDS-DMA instructions of stage k):
//   tail 0: D(4i+4) | tail 1: D(4i+5), z(i+1) | 4 dz stores (during stage 2) | tail 2: D(4i+6) | tail 3: D(4i+7);
//   prologue: z(0), D(1), D(2), D(3).  Tail p waits for D(4i+p+1) before its own issues; the operations younger than it are at
//   least 8, 8, 16, 16 (first unit included) - the counts below.  z(i) is older than everything tail 1 of unit i leaves in flight.
template <int S>
__device__ __forceinline__ float4 lds4s(const float* __restrict__ p) { return make_float4(p[0], p[S], p[2 * S], p[3 * S]); }
template <int ACT>
__device__ __forceinline__ float dact_fast(float z) {
  if (ACT == TAVSR_ACT_SWISH) {
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    return s * (1.f + z * (1.f - s));
  }
  return z > 0.f ? 1.f : 0.f;
}

template <bool DROP, int ACT>
__global__ __launch_bounds__(256, 1) void ffn2_bwd_kernel(const Ffn2Args a) {
  constexpr int NS = 4, STG = 4096, XSF = 2 * 4 * 4096;
  // LDS: ring slot 0 | dyd staging (128 KB), overlaid after the prologue by ring slots 1 .. 3 and the waves' z double buffers
  __shared__ __attribute__((aligned(1024))) float smem[STG + XSF];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  float* const ring = smem;
  float* const xs = smem + STG + w * 4096;           // + half * 4 * 4096
  float* const zb = smem + STG + (NS - 1) * STG + w * 2048;       // [2][32 rows][32 floats]
  static_assert((NS - 1) * STG + 4 * 2048 <= XSF, "ring slots 1.. and the z buffers live in the staging area");

  const int UPR = a.UPR;
  const int rb = blockIdx.x / a.wpb, part = blockIdx.x - rb * a.wpb;
  const int ht0 = part * UPR / a.wpb;
  const int nu = (part + 1) * UPR / a.wpb - ht0;
  if (nu <= 0) return;
  const int m0 = rb * kRB + 32 * w;               // this wave's row tile

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

  // stages of a unit: 0, 1 = W2[128 s .. +128][32 ih .. +32] as [128 rows][32 floats]; 2, 3 = W1[32 ih .. +32][128 (s-2) .. +128]
  // as [32 rows][128 floats]; unswizzled (every fragment read is 32 consecutive floats per half wave).  z tile of the wave:
  // [32 rows][32 floats], 16-byte chunk c of row r at c ^ ((r >> 1) & 7).
  int offA[4], offB[4], offZ[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    offA[j] = (32 * w + 8 * j + (lane >> 3)) * a.N1 + ((lane & 7) << 2);
    offB[j] = (8 * w + 2 * j + (lane >> 5)) * 256 + ((lane & 31) << 2);
    const int r = 8 * j + (lane >> 3);
    offZ[j] = min(m0 + r, a.M - 1) * a.N1 + (((lane & 7) ^ ((r >> 1) & 7)) << 2);
  }
  int hta[3];
  auto set_hta = [&](int iu_) {
#pragma unroll
    for (int d = 0; d < 3; ++d) hta[d] = (a.dbg & 1) ? 0 : ht0 + min(iu_ + d, nu - 1);
  };
  auto issue_one = [&](int du, int sub, int slot, int j) {
    const int ih = hta[du];
    const float* src = sub < 2 ? a.W2 + ((long)(128 * sub) * a.N1 + 32 * ih) + offA[j]
                               : a.W1 + ((long)(32 * ih) * 256 + 128 * (sub - 2)) + offB[j];
    __builtin_amdgcn_global_load_lds((glb_f*)src, (lds_f*)(ring + slot * STG + w * 1024 + j * 256), 16, 0, 0);
  };
  auto issue_z = [&](int ih, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((glb_f*)(a.Z + 32 * ih + offZ[j]), (lds_f*)(zb + buf * 1024 + j * 256), 16, 0, 0);
  };

  set_hta(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) issue_one(0, 0, 0, j);
  const uint64_t sd = DROP ? a.seed[0] : 0;
  // ---- alpha * dyd of the wave's 32 rows into the B-operand registers: xr[4 g + j] = row li, column 8 g + 4 h2 + j
#define FFN2_A1(ST, G) ((ST) + li * 128 + ((((2 * (G) + h2) & ~15) | (((2 * (G) + h2) ^ li) & 15)) << 2))
  float xr[128];
  {
    vmwait<0>();
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const float4 v = lds4(FFN2_A1(xs + (g >> 4) * (4 * 4096), g & 15));
      xr[4 * g] = v.x * a.alpha; xr[4 * g + 1] = v.y * a.alpha; xr[4 * g + 2] = v.z * a.alpha; xr[4 * g + 3] = v.w * a.alpha;
      if ((g & 3) == 3) SB();
    }
    wg_barrier();           // rows in registers everywhere, own quarter of stage 0 landed: the staging area is free
    issue_z(hta[0], 0);
#pragma unroll
    for (int s_ = 1; s_ < NS; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_one(0, s_, s_, j);
  }
#undef FFN2_A1
  int cs = 0;
  float4 fa, fb, ga, gb;
#define FFN2_B1(ST, G) ((ST) + (8 * (G) + 4 * hq) * 32 + lq)
#define FFN2_B2(ST, I) ((ST) + (8 * ((I) >> 2) + 4 * hq) * 128 + 32 * ((I) & 3) + lq)
  {
    int lq = li, hq = h2;
    ga = lds4s<32>(FFN2_B1(ring, 0));
    gb = lds4s<32>(FFN2_B1(ring, 1));
  }
  int iu = 0, ht = ht0;
  {
    f32x16 acc2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

#define FFN2_BTAIL(POS, NWAIT, NEXTP1, MF7A, MF7B)                                                                      \
        fa = ga; fb = gb;                                                                                               \
        vmwait<NWAIT>();                                                                                                \
        wg_barrier();                                                                                                   \
        {                                                                                                               \
          const int cn = cs + 1 == NS ? 0 : cs + 1;                                                                     \
          const float* stn = ring + cn * STG;                                                                           \
          SB();                                                                                                         \
          MF7A                                                                                                          \
          ga = NEXTP1 ? lds4s<32>(FFN2_B1(stn, 0)) : lds4s<128>(FFN2_B2(stn, 0));                                       \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 0);                                                             \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 1);                                                             \
          SB();                                                                                                         \
          MF7B                                                                                                          \
          gb = NEXTP1 ? lds4s<32>(FFN2_B1(stn, 1)) : lds4s<128>(FFN2_B2(stn, 1));                                       \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 2);                                                             \
          issue_one((POS + NS) / 4, (POS + NS) % 4, cs, 3);                                                             \
          SB();                                                                                                         \
          cs = cn;                                                                                                      \
        }

    for (; iu < nu; ++iu, ++ht) {
      set_hta(iu);
      int lq = li, hq = h2;
      asm volatile("" : "+v"(lq), "+v"(hq));
      // ---- phase 1: dh^T tile, hidden unit on the registers (j = rho(r) + 4 h2), row on the lane
      f32x16 acc1, acc1b;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc1b[r] = 0.f; }
#define FFN2_MF1(F, XO, G)                                                                                              \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.x, xr[XO + 4 * (G)], acc1, 0, 0, 0);                            \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.y, xr[XO + 4 * (G) + 1], acc1b, 0, 0, 0);                      \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F.z, xr[XO + 4 * (G) + 2], acc1, 0, 0, 0);                        \
          acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(F.w, xr[XO + 4 * (G) + 3], acc1b, 0, 0, 0);
#define FFN2_P1_STAGE(POS, XO, NEXTP1)                                                                                  \
      {                                                                                                                 \
        const float* st = ring + cs * STG;                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                 \
          fa = ga; fb = gb;                                                                                             \
          ga = lds4s<32>(FFN2_B1(st, 2 * p + 2));                                                                       \
          gb = lds4s<32>(FFN2_B1(st, 2 * p + 3));                                                                       \
          FFN2_MF1(fa, XO, 2 * p)                                                                                       \
          FFN2_MF1(fb, XO, 2 * p + 1)                                                                                   \
          SB();                                                                                                         \
        }                                                                                                               \
        FFN2_BTAIL(POS, 8, NEXTP1, FFN2_MF1(fa, XO, 14), FFN2_MF1(fb, XO, 15))                                          \
      }
      FFN2_P1_STAGE(0, 0, true)
      FFN2_P1_STAGE(1, 64, false)
#undef FFN2_P1_STAGE
#undef FFN2_MF1
      issue_z(hta[1], (iu + 1) & 1);              // next unit's z tile, right behind D(4 iu + 5)
      // ---- dz = dh * act'(z) * mask / keep in place: av[4 q + e] = dz[row li][hidden 8 q + 4 h2 + e] - the A operand of phase 2
      float av[16];
      uint32_t wv[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      float4 zq;
      const float* zt = zb + (iu & 1) * 1024 + li * 32;
      auto activate = [&](int q, int e) {
        const long e0 = (long)(m0 + li) * a.N1 + 32 * ht + 8 * q + 4 * h2;
        if (e == 0) {
          zq = lds4(zt + (((2 * q + h2) ^ ((li >> 1) & 7)) << 2));
          if (DROP) {
            const uint64_t ctr = a.offset4 + ((uint64_t)e0 >> 2);
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), wv);
          }
        }
        const float zv = e == 0 ? zq.x : e == 1 ? zq.y : e == 2 ? zq.z : zq.w;
        av[4 * q + e] = acc1[4 * q + e] * dact_fast<ACT>(zv) * ((!DROP || wv[e] >= a.thr) ? a.inv_keep : 0.f);
        if (e == 3)       // rows >= M exist in the buffer (roundup128(M) rows): unconditional 16-byte stores
          *reinterpret_cast<float4*>(a.H + e0) = make_float4(av[4 * q], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]);
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] += acc1b[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) activate(0, e);
      // ---- phase 2: dn tile += dz x W1[unit, :], 4 + 4 output tiles of 32 columns
#define FFN2_M2(AVI, X, Y, T0)                                                                                          \
          acc2[T0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[AVI], X, acc2[T0], 0, 0, 0);                               \
          acc2[(T0) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[AVI], Y, acc2[(T0) + 1], 0, 0, 0);
#define FFN2_MF2A(FA, FB, NT0, P)                                                                                       \
          FFN2_M2(4 * ((P) >> 1), FA.x, FB.x, NT0 + 2 * ((P) & 1))                                                      \
          FFN2_M2(4 * ((P) >> 1) + 1, FA.y, FB.y, NT0 + 2 * ((P) & 1))
#define FFN2_MF2B(FA, FB, NT0, P)                                                                                       \
          FFN2_M2(4 * ((P) >> 1) + 2, FA.z, FB.z, NT0 + 2 * ((P) & 1))                                                  \
          FFN2_M2(4 * ((P) >> 1) + 3, FA.w, FB.w, NT0 + 2 * ((P) & 1))
#define FFN2_P2_STAGE(POS, NT0, ACTIVATE, NEXTP1)                                                                       \
      {                                                                                                                 \
        const float* st = ring + cs * STG;                                                                              \
        _Pragma("unroll") for (int p = 0; p < 7; ++p) {                                                                 \
          fa = ga; fb = gb;                                                                                             \
          ga = lds4s<128>(FFN2_B2(st, 2 * p + 2));                                                                      \
          gb = lds4s<128>(FFN2_B2(st, 2 * p + 3));                                                                      \
          FFN2_MF2A(fa, fb, NT0, p)                                                                                     \
          FFN2_MF2B(fa, fb, NT0, p)                                                                                     \
          if (ACTIVATE && p < 6) { activate((p >> 1) + 1, 2 * (p & 1)); activate((p >> 1) + 1, 2 * (p & 1) + 1); }      \
          SB();                                                                                                         \
        }                                                                                                               \
        FFN2_BTAIL(POS, 16, NEXTP1, FFN2_MF2A(fa, fb, NT0, 7), FFN2_MF2B(fa, fb, NT0, 7))                               \
      }
      FFN2_P2_STAGE(2, 0, true, false)
      FFN2_P2_STAGE(3, 4, false, true)
#undef FFN2_P2_STAGE
#undef FFN2_MF2A
#undef FFN2_MF2B
#undef FFN2_M2
    }
#undef FFN2_BTAIL
#undef FFN2_B1
#undef FFN2_B2
    {
      float* out = a.slab + ((long)blockIdx.x * kRB + 32 * w) * 256;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(rho(r) + 4 * h2) * 256 + 32 * t + li] = acc2[t][r];
    }
  }
  vmwait<0>();
}

// y = res + scale * dropout(sum of the row block's partials + bias); optional LayerNorms of y.  One wave per row.
__global__ __launch_bounds__(256) void ffn2_finish_kernel(const float* __restrict__ slab, int wpb,
                                                          const float* __restrict__ bias, const float* __restrict__ res, long ldr,
                                                          float* __restrict__ y, int M, float scale, uint32_t thr, float inv_keep,
                                                          const uint64_t* __restrict__ seed, uint64_t offset4,
                                                          const float* __restrict__ g0, const float* __restrict__ b0, float* __restrict__ o0,
                                                          const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ o1,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float eps2) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int rb = m / kRB, rr = m % kRB;       // row block, row inside it
  const float* p = slab + (((long)rb * wpb) * kRB + rr) * 256 + lane * 4;     // partial j of the block: + j * kRB * 256
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  // every partial of the row (and the residual row) in flight before the first add: ONE exposed memory latency per row instead
  // of one per group of four; added in workgroup order (the same sum as before)
  float4 xres = make_float4(0.f, 0.f, 0.f, 0.f);
  if (res) xres = *reinterpret_cast<const float4*>(res + (long)m * ldr + lane * 4);
  auto sum_parts = [&](auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    for (int j0 = 0; j0 < wpb; j0 += NP) {
      float4 t[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) t[j] = *reinterpret_cast<const float4*>(p + (long)min(j0 + j, wpb - 1) * (kRB * 256));
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (j0 + j < wpb) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
    }
  };
  if (wpb <= 6) sum_parts(std::integral_constant<int, 6>{});      // (loads past wpb repeat the last partial: keep them few)
  else sum_parts(std::integral_constant<int, 12>{});
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + lane * 4);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (thr) {
    const uint64_t sd = seed[0], ctr = offset4 + (uint64_t)m * 64u + (uint64_t)lane;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), r);
    v.x = r[0] >= thr ? v.x * inv_keep : 0.f;
    v.y = r[1] >= thr ? v.y * inv_keep : 0.f;
    v.z = r[2] >= thr ? v.z * inv_keep : 0.f;
    v.w = r[3] >= thr ? v.w * inv_keep : 0.f;
  }
  v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
  if (res) { v.x += xres.x; v.y += xres.y; v.z += xres.z; v.w += xres.w; }
  *reinterpret_cast<float4*>(y + (long)m * 256 + lane * 4) = v;
  if (o0) {
    const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
    const float c0 = v.x - mu, c1 = v.y - mu, c2 = v.z - mu, c3 = v.w - mu;
    const float rs = rsqrtf(wave_sum((c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3)) * (1.f / 256.f) + eps2);
    if (mean && lane == 0) { mean[m] = mu; rstd[m] = rs; }
    const float4 gg = *reinterpret_cast<const float4*>(g0 + lane * 4), bb = *reinterpret_cast<const float4*>(b0 + lane * 4);
    *reinterpret_cast<float4*>(o0 + (long)m * 256 + lane * 4) =
        make_float4(c0 * rs * gg.x + bb.x, c1 * rs * gg.y + bb.y, c2 * rs * gg.z + bb.z, c3 * rs * gg.w + bb.w);
    if (o1) {
      const float4 g2 = *reinterpret_cast<const float4*>(g1 + lane * 4), b2 = *reinterpret_cast<const float4*>(b1 + lane * 4);
      *reinterpret_cast<float4*>(o1 + (long)m * 256 + lane * 4) =
          make_float4(c0 * rs * g2.x + b2.x, c1 * rs * g2.y + b2.y, c2 * rs * g2.z + b2.z, c3 * rs * g2.w + b2.w);
    }
  }
}

struct Plan {
  int G, NS, wpb, UPR;
};

// Row blocks of 128 rows; every block's hidden tiles are dealt to wpb workgroups (about one workgroup per CU in all).
// TAVSR_FFN2_CFG="wpb,NS" overrides the plan (tuning runs).
Plan ffn2_plan(int M, int N1) {
  Plan p;
  p.UPR = N1 / 32;
  const int nrb = cdiv(M, kRB);
  p.NS = 4;
  p.wpb = std::max(1, 256 / nrb);
  if (const char* e = getenv("TAVSR_FFN2_CFG")) {
    int g = 0, ns = 0;
    if (sscanf(e, "%d,%d", &g, &ns) == 2 && g > 0 && ns >= 3 && ns <= 5) { p.wpb = g; p.NS = ns; }
  }
  p.wpb = std::min(p.wpb, p.UPR);                              // at least one unit per workgroup
  p.wpb = std::max(p.wpb, cdiv(p.UPR, kMaxU - 1));             // bias slices of at most kMaxU - 1 units fit in LDS
  p.G = nrb * p.wpb;
  return p;
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int NS, int ACT>
void launch_fwd(const Ffn2Args& a, bool save, bool drop, hipStream_t s) {
  dim3 grid(a.G);
  if (save && drop) hipLaunchKernelGGL((ffn2_fwd_kernel<NS, true, true, ACT>), grid, dim3(256), 0, s, a);
  else if (save) hipLaunchKernelGGL((ffn2_fwd_kernel<NS, true, false, ACT>), grid, dim3(256), 0, s, a);
  else if (drop) hipLaunchKernelGGL((ffn2_fwd_kernel<NS, false, true, ACT>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ffn2_fwd_kernel<NS, false, false, ACT>), grid, dim3(256), 0, s, a);
}

}  // namespace
}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_ffn2_trace_read(unsigned long long* dst_host, int32_t n) {      // tuning aid, not part of the product ABI
  if (n > kTraceWG * kTraceN) n = kTraceWG * kTraceN;
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_ffn2_trace), sizeof(unsigned long long) * n);
}

extern "C" int64_t tavsr_ffn2_ws(int32_t M, int32_t D, int32_t N1) {
  if (M <= 0 || D != 256 || N1 < 1024 || N1 % 32 != 0) return 0;
  const Plan p = ffn2_plan(M, N1);
  return (int64_t)p.G * kRB * 256;
}

extern "C" int tavsr_ffn2_fwd(const tavsr_ffn_desc* d, tavsr_stream_t stream) {
  TAVSR_REQUIRE(d, TAVSR_EINVAL, "ffn2_fwd: null descriptor");
  TAVSR_REQUIRE(d->M > 0 && d->D == 256 && d->N1 >= 1024 && d->N1 % 32 == 0, TAVSR_EUNSUPPORTED,
                "ffn2_fwd: d_model 256 and a hidden size >= 1024 that is a multiple of 32 (got %d, %d)", d->D, d->N1);
  TAVSR_REQUIRE(d->act == TAVSR_ACT_RELU || d->act == TAVSR_ACT_SWISH, TAVSR_EUNSUPPORTED, "ffn2_fwd: ReLU or Swish only");
  TAVSR_REQUIRE(d->x && d->ln_w && d->ln_b && d->w1 && d->b1 && d->w2 && d->b2 && d->y && d->ws, TAVSR_EINVAL, "ffn2_fwd: null operand");
  TAVSR_REQUIRE(al16(d->x) && al16(d->w1) && al16(d->w2) && al16(d->y) && al16(d->ws) && al16(d->ln_w) && al16(d->ln_b) &&
                    al16(d->b2) && d->ldx % 4 == 0 && (!d->res || (al16(d->res) && d->ldr % 4 == 0)),
                TAVSR_EINVAL, "ffn2_fwd: operands must be 16-byte aligned");
  TAVSR_REQUIRE((d->z == nullptr) == (d->h == nullptr) && (d->mean == nullptr) == (d->rstd == nullptr), TAVSR_EINVAL,
                "ffn2_fwd: z / h and mean / rstd are saved together");
  TAVSR_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f && (d->p_drop == 0.f || d->seed) && d->offset_in % 4 == 0 &&
                    d->offset_out % 4 == 0, TAVSR_EINVAL, "ffn2_fwd: dropout needs p in [0, 1), a device seed and offsets %% 4 == 0");
  TAVSR_REQUIRE(!d->ln2_out[1] || d->ln2_out[0], TAVSR_EINVAL, "ffn2_fwd: the second LayerNorm output needs the first");
  for (int k = 0; k < 2; ++k)
    TAVSR_REQUIRE(!d->ln2_out[k] || (d->ln2_w[k] && d->ln2_b[k] && al16(d->ln2_w[k]) && al16(d->ln2_b[k]) && al16(d->ln2_out[k])),
                  TAVSR_EINVAL, "ffn2_fwd: LayerNorm %d of the output lacks weights", k);
  const Plan p = ffn2_plan(d->M, d->N1);
  TAVSR_REQUIRE(d->ws_floats >= (int64_t)p.G * kRB * 256, TAVSR_EINVAL, "ffn2_fwd: workspace too small (tavsr_ffn2_ws)");
  Ffn2Args a{};
  a.M = d->M; a.N1 = d->N1; a.G = p.G; a.UPR = p.UPR; a.wpb = p.wpb; a.act = d->act;
  a.x = d->x; a.ldx = d->ldx; a.ln_w = d->ln_w; a.ln_b = d->ln_b; a.W1 = d->w1; a.b1 = d->b1; a.W2 = d->w2; a.eps = d->eps;
  a.slab = d->ws; a.n_out = d->n_out; a.mean = d->mean; a.rstd = d->rstd; a.Z = d->z; a.H = d->h;
  a.thr = d->p_drop > 0.f ? (uint32_t)((double)d->p_drop * 4294967296.0) : 0u;
  a.inv_keep = d->p_drop > 0.f ? 1.f / (1.f - d->p_drop) : 1.f;
  a.seed = d->seed; a.offset4 = d->offset_in / 4;
  if (const char* e = getenv("TAVSR_FFN2_DBG")) a.dbg = atoi(e);
  hipStream_t s = (hipStream_t)stream;
  const bool save = d->z != nullptr, drop = a.thr != 0;
  if (p.NS == 5) {
    if (d->act == TAVSR_ACT_RELU) launch_fwd<5, TAVSR_ACT_RELU>(a, save, drop, s);
    else launch_fwd<5, TAVSR_ACT_SWISH>(a, save, drop, s);
  } else if (p.NS == 4) {
    if (d->act == TAVSR_ACT_RELU) launch_fwd<4, TAVSR_ACT_RELU>(a, save, drop, s);
    else launch_fwd<4, TAVSR_ACT_SWISH>(a, save, drop, s);
  } else {
    if (d->act == TAVSR_ACT_RELU) launch_fwd<3, TAVSR_ACT_RELU>(a, save, drop, s);
    else launch_fwd<3, TAVSR_ACT_SWISH>(a, save, drop, s);
  }
  TAVSR_LAUNCH_CHECK();
  const float* res = d->res ? d->res : d->x;
  const int64_t ldr = d->res ? d->ldr : d->ldx;
  hipLaunchKernelGGL(ffn2_finish_kernel, dim3(cdiv(d->M, 4)), dim3(256), 0, s, d->ws, p.wpb, d->b2, res, (long)ldr,
                     d->y, d->M, d->scale, a.thr, a.inv_keep, d->seed, d->offset_out / 4, d->ln2_w[0], d->ln2_b[0], d->ln2_out[0],
                     d->ln2_w[1], d->ln2_b[1], d->ln2_out[1], d->ln2_mean, d->ln2_rstd, d->ln2_eps);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_ffn2_bwd_dx(const float* dy, int64_t lddy, float alpha, const float* w1, const float* w2, const float* z,
                                 int32_t act, int32_t M, int32_t D, int32_t N1, float p_drop, const uint64_t* seed_dev,
                                 uint64_t offset_in, float* dz, float* dn, float* ws, int64_t ws_floats, tavsr_stream_t stream) {
  TAVSR_REQUIRE(M > 0 && D == 256 && N1 >= 1024 && N1 % 32 == 0, TAVSR_EUNSUPPORTED,
                "ffn2_bwd_dx: d_model 256 and a hidden size >= 1024 that is a multiple of 32 (got %d, %d)", D, N1);
  TAVSR_REQUIRE(act == TAVSR_ACT_RELU || act == TAVSR_ACT_SWISH, TAVSR_EUNSUPPORTED, "ffn2_bwd_dx: ReLU or Swish only");
  TAVSR_REQUIRE(dy && w1 && w2 && z && dz && ws, TAVSR_EINVAL, "ffn2_bwd_dx: null operand");
  TAVSR_REQUIRE(al16(dy) && al16(w1) && al16(w2) && al16(z) && al16(dz) && al16(dn) && al16(ws) && lddy % 4 == 0, TAVSR_EINVAL,
                "ffn2_bwd_dx: operands must be 16-byte aligned");
  TAVSR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_dev) && offset_in % 4 == 0, TAVSR_EINVAL,
                "ffn2_bwd_dx: dropout needs p in [0, 1), a device seed and an offset %% 4 == 0");
  Plan p = ffn2_plan(M, N1);
  TAVSR_REQUIRE(ws_floats >= (int64_t)p.G * kRB * 256, TAVSR_EINVAL, "ffn2_bwd_dx: workspace too small (tavsr_ffn2_ws)");
  Ffn2Args a{};
  a.M = M; a.N1 = N1; a.G = p.G; a.UPR = p.UPR; a.wpb = p.wpb; a.act = act;
  a.x = dy; a.ldx = lddy; a.alpha = alpha; a.W1 = w1; a.W2 = w2; a.slab = ws; a.Z = const_cast<float*>(z); a.H = dz;
  a.thr = p_drop > 0.f ? (uint32_t)((double)p_drop * 4294967296.0) : 0u;
  a.inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  a.seed = seed_dev; a.offset4 = offset_in / 4;
  if (const char* e = getenv("TAVSR_FFN2_DBG")) a.dbg = atoi(e) & 1;
  hipStream_t s = (hipStream_t)stream;
  const bool drop = a.thr != 0;
  dim3 grid(a.G);
  if (act == TAVSR_ACT_RELU) {
    if (drop) hipLaunchKernelGGL((ffn2_bwd_kernel<true, TAVSR_ACT_RELU>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ffn2_bwd_kernel<false, TAVSR_ACT_RELU>), grid, dim3(256), 0, s, a);
  } else {
    if (drop) hipLaunchKernelGGL((ffn2_bwd_kernel<true, TAVSR_ACT_SWISH>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ffn2_bwd_kernel<false, TAVSR_ACT_SWISH>), grid, dim3(256), 0, s, a);
  }
  TAVSR_LAUNCH_CHECK();
  if (!dn) return TAVSR_OK;      // the caller's LayerNorm backward sums the partials itself (tavsr_layernorm_bwd_partial_slab)
  hipLaunchKernelGGL(ffn2_finish_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, ws, p.wpb, (const float*)nullptr, (const float*)nullptr,
                     0L, dn, M, 1.f, 0u, 1.f, (const uint64_t*)nullptr, (uint64_t)0, (const float*)nullptr, (const float*)nullptr,
                     (float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

// where tavsr_ffn2_bwd_dx(dn = NULL) leaves the partials of dn: row m = sum over j < *wpb of ws[((m / *rb_rows) * *wpb + j) * *rb_rows + m % *rb_rows][256]
extern "C" int tavsr_ffn2_slab_layout(int32_t M, int32_t N1, int32_t* wpb, int32_t* rb_rows) {
  TAVSR_REQUIRE(M > 0 && N1 >= 1024 && N1 % 32 == 0 && wpb && rb_rows, TAVSR_EINVAL, "ffn2_slab_layout: bad argument");
  const Plan p = ffn2_plan(M, N1);
  *wpb = p.wpb;
  *rb_rows = kRB;
  return TAVSR_OK;
}
