// Word / character error rates with bootstrap confidence intervals: the evaluator the reference shells
// out to (src/evaluation/bootstrap_wer.py:3-16 -> src/evaluation/tasas/tasas.c, tasasIntervalo.c, option -ie, p = 1):
// per sentence pair the unit-cost edit distance of the symbol sequences (words split on blanks, or single bytes),
// rate = 100 * sum(substitutions + insertions + deletions) / sum(reference lengths); the interval is 1.64 standard
// deviations of that rate over bootstrap resamples of the sentence set (1000 by default).
//   * edit distance: one workgroup per sentence pair, anti-diagonal wavefront over the DP lattice, three diagonals in LDS;
//     integer arithmetic, results are exact.  (The reference's backtrace splits the distance into substitution /
//     insertion / deletion counts by a tie-breaking order; the -ie rate only uses their sum and the reference length.)
//   * bootstrap: one workgroup per resample draws n sentence indices (Philox counter = (resample, position)) and sums
//     their distances and reference lengths; the reference's rand() stream (seeded with time(0)) is not reproducible,
//     so parity here is statistical.
#include "common.h"

namespace tavsr {

constexpr int kEditMaxLen = 4095;     // symbols per sequence (tasas reads lines of at most 2047 bytes)

__global__ __launch_bounds__(256) void edit_distance_kernel(const int32_t* __restrict__ ref, const int64_t* __restrict__ ref_off,
                                                            const int32_t* __restrict__ hyp, const int64_t* __restrict__ hyp_off,
                                                            int32_t* __restrict__ dist) {
  __shared__ int32_t diag[3][kEditMaxLen + 1];
  const int p = blockIdx.x;
  const int32_t* r = ref + ref_off[p];
  const int32_t* h = hyp + hyp_off[p];
  const int Lr = (int)(ref_off[p + 1] - ref_off[p]), Lh = (int)(hyp_off[p + 1] - hyp_off[p]);
  // D[i][j], i symbols of the reference against j of the hypothesis; diagonal k holds D[i][k - i] at index i
  int a = 0, b = 1, c = 2;             // diagonals k - 2, k - 1, k
  if (threadIdx.x == 0) diag[b][0] = 0;            // k = 0: D[0][0]
  __syncthreads();
  for (int k = 1; k <= Lr + Lh; ++k) {
    const int lo = max(0, k - Lh), hi = min(Lr, k);
    for (int i = lo + threadIdx.x; i <= hi; i += 256) {
      const int j = k - i;
      int v;
      if (i == 0) v = j;
      else if (j == 0) v = i;
      else {
        const int sub = diag[a][i - 1] + (r[i - 1] != h[j - 1] ? 1 : 0);
        v = min(sub, min(diag[b][i - 1], diag[b][i]) + 1);
      }
      diag[c][i] = v;
    }
    __syncthreads();
    const int t = a; a = b; b = c; c = t;
  }
  if (threadIdx.x == 0) dist[p] = diag[b][Lr];
}

// rates[it] = 100 * sum_x dist[idx(it, x)] / sum_x reflen[idx(it, x)],  idx uniform on [0, n)
__global__ __launch_bounds__(256) void bootstrap_rates_kernel(const int32_t* __restrict__ dist, const int32_t* __restrict__ reflen,
                                                              int n, uint64_t seed, double* __restrict__ rates) {
  __shared__ long long red[2][256];
  const uint32_t it = blockIdx.x;
  long long sd = 0, sl = 0;
  for (int x4 = threadIdx.x; x4 * 4 < n; x4 += 256) {
    uint32_t w[4];
    philox4x32_10((uint32_t)x4, it, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (x4 * 4 + q >= n) break;
      const int idx = (int)(((uint64_t)w[q] * (uint64_t)n) >> 32);      // unbiased to 2^-32 (rand() % n in the reference)
      sd += dist[idx];
      sl += reflen[idx];
    }
  }
  red[0][threadIdx.x] = sd;
  red[1][threadIdx.x] = sl;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) rates[it] = 100.0 * (double)red[0][0] / (double)red[1][0];
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_edit_distance(const int32_t* ref, const int64_t* ref_off, const int32_t* hyp, const int64_t* hyp_off,
                                   int32_t n_pairs, int32_t max_len, int32_t* dist, tavsr_stream_t stream) {
  TAVSR_REQUIRE(ref_off && hyp_off && dist, TAVSR_EINVAL, "edit_distance: null pointer");
  TAVSR_REQUIRE(max_len >= 0 && max_len <= kEditMaxLen, TAVSR_EUNSUPPORTED, "edit_distance: sequences of at most %d symbols (got %d)",
                kEditMaxLen, max_len);
  if (n_pairs <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(edit_distance_kernel, dim3((unsigned)n_pairs), dim3(256), 0, (hipStream_t)stream, ref, ref_off, hyp, hyp_off,
                     dist);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_bootstrap_rates(const int32_t* dist, const int32_t* reflen, int32_t n, int32_t iters, uint64_t seed,
                                     double* rates, tavsr_stream_t stream) {
  TAVSR_REQUIRE(dist && reflen && rates, TAVSR_EINVAL, "bootstrap_rates: null pointer");
  TAVSR_REQUIRE(n > 0 && iters > 0, TAVSR_EINVAL, "bootstrap_rates: need sentences and resamples");
  hipLaunchKernelGGL(bootstrap_rates_kernel, dim3((unsigned)iters), dim3(256), 0, (hipStream_t)stream, dist, reflen, n, seed, rates);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
