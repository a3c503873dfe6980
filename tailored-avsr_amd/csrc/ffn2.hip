// Position-wise feed-forward block as ONE streaming kernel for gfx950, second design (espnet PositionwiseFeedForward inside
// its residual block: src/encoder/branchformer/encoder_layer.py:191-194,311-314; the tailored AV layer's shared FFNs,
// src/encoder/audiovisual/tailored/encoder_layer.py:173-175,211-213; d_model 256):
//     y = x + scale * dropout(W2 dropout(act(W1 LN(x) + b1)) + b2)       [+ LayerNorms of y for the consumers of y]
//
// Why a second design: the first chain kernel (ffn.hip) gathered its weight fragments row-per-lane from L2 and its
// (row tile x hidden slice) workgroups filled 77 % of the chip.  Here
//   * the unit of work is (32 rows) x (32 hidden units): phase 1 computes the 32x32 tile of LN(x) W1^T with the K = 256
//     contraction SPLIT over the workgroup's four waves (each wave keeps its 64-wide slice of the LN'd rows in 32
//     registers for the whole row tile), the four partial tiles meet in LDS (one barrier per unit, double buffered),
//     every wave rebuilds the activated tile as its A operand and phase 2 adds tile x W2[:, unit]^T into the wave's 64
//     output columns (2 accumulator tiles that live as long as the row tile);
//   * the flat unit list (row tiles x hidden tiles, 6336 units at M = 3168, hidden 2048) is cut into EQUAL contiguous
//     ranges, one per workgroup (two workgroups per CU): 12 or 13 units each, 95+ % balance instead of 77 %; a range that
//     crosses a row-tile boundary writes two partial outputs, and the finishing kernel adds the (at most four) partials
//     of a row tile in workgroup order - deterministic, no atomics;
//   * weights never touch registers on their way in: every wave streams ITS slices (W1: 32 hidden x 64 k, W2: 64 out x
//     32 hidden per unit, 16 KB) through a private LDS ring with 16-byte LDS-DMA (global_load_lds, full 128-byte lines,
//     XOR swizzle on the source address), NS stages of 4 KB ahead, ordered by counted s_waitcnt vmcnt only - no barrier
//     guards the weight stream;
//   * the hidden activations make no HBM round trip (written once, coalesced, when a backward pass will need them).
// v_mfma_f32_32x32x2_f32: exact fp32 (the 1e-4 parity bar).  One unit = 64 MFMAs per wave.
//
// The finishing kernel (one wave per row) adds the partials, bias, dropout, scale and residual, and can emit up to two
// LayerNorms of the result (norm_mha / norm_mlp after the macaron block, norm_final after the second block): the layer's
// stand-alone LayerNorm launches disappear.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace tavsr {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_f;
typedef const __attribute__((address_space(1))) float glb_f;

constexpr int kMaxU = 32;          // units per workgroup (bias slices staged in LDS)

struct Ffn2Args {
  int dbg;                         // tuning runs only (TAVSR_FFN2_DBG): bit 0 = every unit streams hidden tile 0's weights
  int M, N1, G, UPR, maxseg, act;  // rows, hidden units, workgroups, units per row tile (N1 / 32), slab slots per workgroup
  long U;                          // units in all: row tiles * UPR
  const float* x;                  // [M][ldx]
  long ldx;
  const float *ln_w, *ln_b, *W1, *b1, *W2;
  float eps;
  float* slab;                     // [G][maxseg][32][256]
  float *n_out, *mean, *rstd;      // saved LayerNorm output / statistics (null: not kept)
  float *Z, *H;                    // [roundup32(M)][N1] (SAVE)
  uint32_t thr;                    // inner dropout: element (m, c) = word c & 3 of counter offset4 + (m * N1 + c) / 4
  float inv_keep;
  const uint64_t* seed;
  uint64_t offset4;
};

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }
template <int N>
__device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lgkwait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
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
__device__ __forceinline__ float lds1(const float* __restrict__ p) { return *p; }
__device__ __forceinline__ uint32_t ldsu(const uint32_t* __restrict__ p) { return *p; }
__device__ __forceinline__ uint32_t ldsb(const unsigned char* __restrict__ p) { return *p; }

// NS: stages (4 KB) of a wave's weight ring.  SAVE: z / h (and the LayerNorm output) are written for the backward pass.
template <int NS, bool SAVE, bool DROP, int ACT>
__global__ __launch_bounds__(256, 2) void ffn2_fwd_kernel(const Ffn2Args a) {
  constexpr int HPF = 2 * 4 * 1024;               // partial tiles [2 buffers][4 waves][32 x 32]; aliased by the LN'd row tile
  constexpr int RINGF = 4 * NS * 1024;
  constexpr int NVM = (NS - 1) * 4;               // LDS-DMA instructions that may stay in flight behind the awaited stage
  __shared__ __attribute__((aligned(1024))) float smem[HPF + RINGF + kMaxU * 32 + 128];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h2 = lane >> 5;
  float* const Hp = smem;
  float* const ring = smem + HPF + w * (NS * 1024);
  float* const b1s = smem + HPF + RINGF;
  unsigned char* const Mk = reinterpret_cast<unsigned char*>(b1s + kMaxU * 32);   // keep bits [2][32 rows][8 chunks]

  const int UPR = a.UPR;
  const long ub = (long)blockIdx.x * a.U / a.G, ue = (long)(blockIdx.x + 1) * a.U / a.G;
  const int nu = (int)(ue - ub);
  if (nu <= 0) return;
  int rt = (int)(ub / UPR);
  int ht = (int)(ub - (long)rt * UPR);            // hidden tile of the unit at hand
  const int ht0 = ht;

  // ---- weight stream: stage k of unit i is W1[32 ih .. +32][64 w + 32 k .. +32] (k = 0, 1) or W2[64 w + 32 (k-2) .. +32][32 ih .. +32]
  // as a k-contiguous [32 rows][32 floats] image; 16-byte chunk c of row r lands at chunk c ^ ((r >> 1) & 7)
  int offW1[4], offW2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = j * 64 + lane, row = q >> 3, cl = (q & 7) ^ ((row >> 1) & 7);
    offW1[j] = row * 256 + cl * 4;
    offW2[j] = row * a.N1 + cl * 4;
  }
  // hidden tiles of the units iu, iu + 1, iu + 2 (clamped to the range's last unit: past the end the stream fetches valid
  // addresses whose data is never used).  UPR >= 32 > units per workgroup: one conditional subtraction wraps.
  int hta[3];
  auto set_hta = [&](int iu_) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int v = ht0 + min(iu_ + d, nu - 1);
      hta[d] = (a.dbg & 1) ? 0 : (v >= UPR ? v - UPR : v);
    }
  };
  // LDS-DMA instruction j of stage `sub` of unit iu + du -> ring slot (sub, du: compile-time at every call site)
  auto issue_one = [&](int du, int sub, int slot, int j) {
    const int ih = hta[du];
    const float* src = sub < 2 ? a.W1 + ((long)(32 * ih) * 256 + 64 * w + 32 * sub) + offW1[j]
                               : a.W2 + ((long)(64 * w + 32 * (sub - 2)) * a.N1 + 32 * ih) + offW2[j];
    __builtin_amdgcn_global_load_lds((glb_f*)src, (lds_f*)(ring + slot * 1024 + j * 256), 16, 0, 0);
  };
  auto read16 = [&](int slot, float (&f)[16]) {
    const float* s = ring + slot * 1024 + li * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = lds4(s + (((2 * g + h2) ^ ((li >> 1) & 7)) << 2));
      f[4 * g] = v.x; f[4 * g + 1] = v.y; f[4 * g + 2] = v.z; f[4 * g + 3] = v.w;
    }
  };

  set_hta(0);
#pragma unroll
  for (int s = 0; s < NS; ++s)                    // stages 0 .. NS-1 (all of unit 0: NS <= 3) on their way before anything else
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_one(0, s, s, j);
  // bias slices of this workgroup's units (LDS: an ordinary load inside the loop would drain the weight stream)
  for (int i = tid; i < nu * 32; i += 256) {
    int ih = ht0 + (i >> 5);
    while (ih >= UPR) ih -= UPR;
    b1s[i] = a.b1[32 * ih + (i & 31)];
  }

  const uint64_t sd = DROP ? a.seed[0] : 0;
  float fbC[16], fbN[16];
  // fragments of the very first stage; its slot takes stage NS
  vmwait<NVM>();
  read16(0, fbC);
  lgkwait0();
  SB();
#pragma unroll
  for (int j = 0; j < 4; ++j) issue_one(NS / 4, NS % 4, 0, j);
  int cs = 1 % NS;                                // ring slot of the stage whose fragments are read next
  int iu = 0, seg = 0;
  while (iu < nu) {
    const int m0 = rt * 32;
    const int nsu = min(UPR - ht, nu - iu);       // units of this row tile
    // ---- LayerNorm(x) of the row tile -> LDS (16-byte chunk c of row r at chunk (c & ~15) | ((c ^ r) & 15))
    wg_barrier();                                 // everyone is done with the previous row tile's partials
    {
      const bool owner = SAVE && ht == 0;         // the workgroup that holds the row tile's first unit keeps the LN output
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = w * 8 + q, m = min(m0 + row, a.M - 1);
        float4 v = *reinterpret_cast<const float4*>(a.x + (long)m * a.ldx + lane * 4);
        const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
        const float c0 = v.x - mu, c1 = v.y - mu, c2 = v.z - mu, c3 = v.w - mu;
        const float rs = rsqrtf(wave_sum((c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3)) * (1.f / 256.f) + a.eps);
        const float4 gg = *reinterpret_cast<const float4*>(a.ln_w + lane * 4), bb = *reinterpret_cast<const float4*>(a.ln_b + lane * 4);
        v = make_float4(c0 * rs * gg.x + bb.x, c1 * rs * gg.y + bb.y, c2 * rs * gg.z + bb.z, c3 * rs * gg.w + bb.w);
        if (owner && m0 + row < a.M) {
          if (a.n_out) *reinterpret_cast<float4*>(a.n_out + (long)m * 256 + lane * 4) = v;
          if (a.mean && lane == 0) { a.mean[m] = mu; a.rstd[m] = rs; }
        }
        *reinterpret_cast<float4*>(Hp + row * 256 + (((lane & ~15) | ((lane ^ row) & 15)) << 2)) = v;
      }
    }
    wg_barrier();
    float xq[32];                                 // this wave's K slice of the row tile: k = 64 w + 8 g + 4 h2 + j
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int c = 16 * w + 2 * g + h2;
      const float4 v = lds4(Hp + li * 256 + (((c & ~15) | ((c ^ li) & 15)) << 2));
      xq[4 * g] = v.x; xq[4 * g + 1] = v.y; xq[4 * g + 2] = v.z; xq[4 * g + 3] = v.w;
    }
    wg_barrier();                                 // the row tile image may now be overwritten by partial tiles
    f32x16 acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    // One stage = 16 MFMAs with the fragments FC; the NEXT stage's fragments are read into FN after the 4th MFMA (the
    // LDS-DMA that brings them was issued NS stages ago), and once those reads are back the freed ring slot is refilled,
    // one LDS-DMA instruction per MFMA shadow.
#define FFN2_STAGE(POS, AV, AO, FC, FN, ACC, NWAIT)                                                              \
    {                                                                                                       \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[AO + s_], FC[s_], ACC, 0, 0, 0);                      \
      SB();                                                                                                 \
      vmwait<NWAIT>();                                                                                      \
      read16(cs, FN);                                                                                       \
      SB();                                                                                                 \
      _Pragma("unroll") for (int s_ = 4; s_ < 10; ++s_)                                                     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[AO + s_], FC[s_], ACC, 0, 0, 0);                      \
      SB();                                                                                                 \
      lgkwait0();                                                                                           \
      SB();                                                                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                    \
        issue_one((POS + 1 + NS) / 4, (POS + 1 + NS) % 4, cs, j_);                                          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[AO + 10 + j_], FC[10 + j_], ACC, 0, 0, 0);            \
        SB();                                                                                               \
      }                                                                                                     \
      ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[AO + 14], FC[14], ACC, 0, 0, 0);                        \
      ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[AO + 15], FC[15], ACC, 0, 0, 0);                        \
      SB();                                                                                                 \
      cs = cs + 1 == NS ? 0 : cs + 1;                                                                       \
    }

    for (int t = 0; t < nsu; ++t, ++iu, ++ht) {
      const int buf = iu & 1;
      float* const P = Hp + buf * 4096;
      set_hta(iu);
      // ---- phase 1: partial tile of this wave's K slice (rows on the registers, hidden unit on the lane); bias rides on wave 0
      const float bv = lds1(b1s + iu * 32 + li);
      f32x16 acc1;
      {
        const float b0 = w == 0 ? bv : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = b0;
      }
      FFN2_STAGE(0, xq, 0, fbC, fbN, acc1, NVM)
      FFN2_STAGE(1, xq, 16, fbN, fbC, acc1, NVM)
      // ---- the four partial tiles meet in LDS: [wave][m][32 floats], 16-byte chunk c of row m at chunk c ^ ((m >> 1) & 7)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rho(r) + 4 * h2;
        P[w * 1024 + m * 32 + ((((li >> 2) ^ ((m >> 1) & 7))) << 2) + (li & 3)] = acc1[r];
      }
      if (DROP) {       // keep bits of chunk 2 w + h2 of row li (one Philox call covers 4 consecutive hidden units)
        const uint64_t e = (uint64_t)(m0 + li) * (uint64_t)a.N1 + (uint64_t)(32 * ht + 8 * w + 4 * h2);
        const uint64_t ctr = a.offset4 + (e >> 2);
        uint32_t wv[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), wv);
        Mk[buf * 256 + li * 8 + h2 * 4 + w] =
            (unsigned char)((wv[0] >= a.thr) | ((wv[1] >= a.thr) << 1) | ((wv[2] >= a.thr) << 2) | ((wv[3] >= a.thr) << 3));
      }
      wg_barrier();
      // ---- every wave rebuilds the activated tile as its A operand: lane = row, 4 consecutive hidden units per read
      float av[16];
      {
        uint32_t kb = 0xffffffffu;
        if (DROP) kb = ldsu(reinterpret_cast<const uint32_t*>(Mk + buf * 256 + li * 8 + h2 * 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = li * 32 + (((2 * q + h2) ^ ((li >> 1) & 7)) << 2);
          const float4 v0 = lds4(P + off), v1 = lds4(P + 1024 + off), v2 = lds4(P + 2048 + off), v3 = lds4(P + 3072 + off);
          const float z0 = (v0.x + v1.x) + (v2.x + v3.x), z1 = (v0.y + v1.y) + (v2.y + v3.y);
          const float z2 = (v0.z + v1.z) + (v2.z + v3.z), z3 = (v0.w + v1.w) + (v2.w + v3.w);
          const uint32_t b = kb >> (8 * q);
          av[4 * q] = act_fwd(ACT, z0) * ((b & 1u) ? a.inv_keep : 0.f);
          av[4 * q + 1] = act_fwd(ACT, z1) * ((b & 2u) ? a.inv_keep : 0.f);
          av[4 * q + 2] = act_fwd(ACT, z2) * ((b & 4u) ? a.inv_keep : 0.f);
          av[4 * q + 3] = act_fwd(ACT, z3) * ((b & 8u) ? a.inv_keep : 0.f);
          if (q & 1) SB();          // two chunks' loads (32 registers) in flight at a time, not all sixteen
        }
      }
      if (SAVE) {       // rows 8 w .. 8 w + 7 of z and h, hidden unit on the lane: 128-byte row segments (8 stores per wave)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = 8 * w + e + 4 * h2;
          const int off = m * 32 + ((((li >> 2) ^ ((m >> 1) & 7))) << 2) + (li & 3);
          const float z = (lds1(P + off) + lds1(P + 1024 + off)) + (lds1(P + 2048 + off) + lds1(P + 3072 + off));
          float keep = a.inv_keep;
          if (DROP) keep = ((ldsb(Mk + buf * 256 + m * 8 + ((li >> 2) & 1) * 4 + (li >> 3)) >> (li & 3)) & 1) ? a.inv_keep : 0.f;
          const long o = (long)(m0 + m) * a.N1 + 32 * ht + li;
          a.Z[o] = z;
          a.H[o] = act_fwd(ACT, z) * keep;
        }
        SB();
      }
      // ---- phase 2: tile x W2[:, unit]^T into this wave's 64 output columns
      FFN2_STAGE(2, av, 0, fbC, fbN, acc2[0], NVM + (SAVE ? 8 : 0))
      FFN2_STAGE(3, av, 0, fbN, fbC, acc2[1], NVM + (SAVE ? 8 : 0))
    }
#undef FFN2_STAGE
    // ---- partial output of this (row tile, unit range) -> slab slot
    {
      float* out = a.slab + ((long)blockIdx.x * a.maxseg + seg) * (32 * 256);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(rho(r) + 4 * h2) * 256 + 64 * w + 32 * t + li] = acc2[t][r];
    }
    ++seg;
    ++rt;
    ht = 0;
  }
  vmwait<0>();          // the stream ran ahead of the last unit: nothing may land in LDS after this workgroup has left
}

// Where the flat unit list is cut: workgroup g owns units [g U / G, (g + 1) U / G).
__device__ __forceinline__ int wg_of_unit(long u, long U, int G) { return (int)(((u + 1) * G + U - 1) / U) - 1; }

// y = res + scale * dropout(sum of the row tile's partials + bias); optional LayerNorms of y.  One wave per row.
__global__ __launch_bounds__(256) void ffn2_finish_kernel(const float* __restrict__ slab, int G, int maxseg, int UPR, long U,
                                                          const float* __restrict__ bias, const float* __restrict__ res, long ldr,
                                                          float* __restrict__ y, int M, float scale, uint32_t thr, float inv_keep,
                                                          const uint64_t* __restrict__ seed, uint64_t offset4,
                                                          const float* __restrict__ g0, const float* __restrict__ b0, float* __restrict__ o0,
                                                          const float* __restrict__ g1, const float* __restrict__ b1, float* __restrict__ o1,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float eps2) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int rt = m >> 5, rr = m & 31;
  const long u0 = (long)rt * UPR;
  const int ga = wg_of_unit(u0, U, G), gb = wg_of_unit(u0 + UPR - 1, U, G);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = ga; g <= gb; ++g) {
    const int sg = rt - (int)(((long)g * U / G) / UPR);       // which of the workgroup's row tiles this one is
    const float4 t = *reinterpret_cast<const float4*>(slab + (((long)g * maxseg + sg) * 32 + rr) * 256 + lane * 4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
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
  if (res) {
    const float4 x = *reinterpret_cast<const float4*>(res + (long)m * ldr + lane * 4);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
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
  int G, NS, maxseg, UPR;
  long U;
};

// TAVSR_FFN2_CFG="G,NS" overrides the plan (tuning runs).
Plan ffn2_plan(int M, int N1) {
  Plan p;
  p.UPR = N1 / 32;
  p.U = (long)cdiv(M, 32) * p.UPR;
  p.NS = 2;
  p.G = 512;
  if (const char* e = getenv("TAVSR_FFN2_CFG")) {
    int g = 0, ns = 0;
    if (sscanf(e, "%d,%d", &g, &ns) == 2 && g > 0 && (ns == 2 || ns == 3)) { p.G = g; p.NS = ns; }
  }
  if (p.G > p.U) p.G = (int)p.U;
  while (cdiv(p.U, p.G) > kMaxU - 1) p.G *= 2;        // bias slices of at most kMaxU units fit in LDS
  p.maxseg = 2 + (cdiv(p.U, p.G) + 1) / p.UPR;
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

extern "C" int64_t tavsr_ffn2_ws(int32_t M, int32_t D, int32_t N1) {
  if (M <= 0 || D != 256 || N1 < 1024 || N1 % 32 != 0) return 0;
  const Plan p = ffn2_plan(M, N1);
  return (int64_t)p.G * p.maxseg * 32 * 256;
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
  TAVSR_REQUIRE(d->ws_floats >= (int64_t)p.G * p.maxseg * 32 * 256, TAVSR_EINVAL, "ffn2_fwd: workspace too small (tavsr_ffn2_ws)");
  Ffn2Args a{};
  a.M = d->M; a.N1 = d->N1; a.G = p.G; a.UPR = p.UPR; a.maxseg = p.maxseg; a.act = d->act; a.U = p.U;
  a.x = d->x; a.ldx = d->ldx; a.ln_w = d->ln_w; a.ln_b = d->ln_b; a.W1 = d->w1; a.b1 = d->b1; a.W2 = d->w2; a.eps = d->eps;
  a.slab = d->ws; a.n_out = d->n_out; a.mean = d->mean; a.rstd = d->rstd; a.Z = d->z; a.H = d->h;
  a.thr = d->p_drop > 0.f ? (uint32_t)((double)d->p_drop * 4294967296.0) : 0u;
  a.inv_keep = d->p_drop > 0.f ? 1.f / (1.f - d->p_drop) : 1.f;
  a.seed = d->seed; a.offset4 = d->offset_in / 4;
  if (const char* e = getenv("TAVSR_FFN2_DBG")) a.dbg = atoi(e);
  hipStream_t s = (hipStream_t)stream;
  const bool save = d->z != nullptr, drop = a.thr != 0;
  if (p.NS == 3) {
    if (d->act == TAVSR_ACT_RELU) launch_fwd<3, TAVSR_ACT_RELU>(a, save, drop, s);
    else launch_fwd<3, TAVSR_ACT_SWISH>(a, save, drop, s);
  } else {
    if (d->act == TAVSR_ACT_RELU) launch_fwd<2, TAVSR_ACT_RELU>(a, save, drop, s);
    else launch_fwd<2, TAVSR_ACT_SWISH>(a, save, drop, s);
  }
  TAVSR_LAUNCH_CHECK();
  const float* res = d->res ? d->res : d->x;
  const int64_t ldr = d->res ? d->ldr : d->ldx;
  hipLaunchKernelGGL(ffn2_finish_kernel, dim3(cdiv(d->M, 4)), dim3(256), 0, s, d->ws, p.G, p.maxseg, p.UPR, p.U, d->b2, res, (long)ldr,
                     d->y, d->M, d->scale, a.thr, a.inv_keep, d->seed, d->offset_out / 4, d->ln2_w[0], d->ln2_b[0], d->ln2_out[0],
                     d->ln2_w[1], d->ln2_b[1], d->ln2_out[1], d->ln2_mean, d->ln2_rstd, d->ln2_eps);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
