// Log-mel feature frontend and SpecAug: espnet2 DefaultFrontend as configured by
// configs/ASR/branchformer_transformer+ctc_english.yaml:9-37 and called at src/models/espnet_model.py:378-388
// (waveform -> STFT 512/400/160 hann, centre + reflect -> power -> 80 Slaney mel -> log(clamp 1e-10)), and espnet2
// SpecAug (bicubic time warp, frequency / time masks).  The two contractions (windowed DFT: K = n_fft; mel: K = 257
// padded to 288) run on the fp32 MFMA GEMM (tavsr_gemm); the kernels here are the HBM-bound steps around them:
// framing (8 B / element), power (12 B / bin), log + length mask (8 B / value), warp and masks (8 B / value).
#include "common.h"

namespace tavsr {

// frames[b*T + t][n] = win[n] * x_reflect[b][t*hop + n - pad],  pad = center ? n_fft/2 : 0.  torch.stft pads the
// batch tensor (length N), not each utterance: samples past an utterance's own length are whatever the batch holds.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                          float* __restrict__ frames, int B, int64_t N, int T, int n_fft,
                                                          int hop, int pad) {
  const int64_t row = blockIdx.x;        // b*T + t
  const int b = (int)(row / T), t = (int)(row % T);
  const float* xb = x + (int64_t)b * N;
  for (int n = threadIdx.x; n < n_fft; n += 256) {
    int64_t i = (int64_t)t * hop + n - pad;
    if (i < 0) i = -i;                   // reflect (no edge repeat), as torch.nn.functional.pad(mode="reflect")
    if (i >= N) i = 2 * (N - 1) - i;
    frames[row * n_fft + n] = win[n] * xb[i];
  }
}

// P[row][k] = re^2 + im^2 for k < nfreq (spec row = [re_0..re_{nfreq-1} | im_0..im_{nfreq-1}]), 0 in the K padding and in
// rows past the utterance's frame count (espnet Stft / LogMel pad masks)
__global__ __launch_bounds__(256) void power_spec_kernel(const float* __restrict__ spec, int64_t ld_spec,
                                                         float* __restrict__ P, int ldp, int nfreq, int T,
                                                         const int64_t* __restrict__ olens) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / T), t = (int)(row % T);
  const bool live = t < olens[b];
  const float* s = spec + row * ld_spec;
  for (int k = threadIdx.x; k < ldp; k += 256) {
    float v = 0.f;
    if (live && k < nfreq) {
      const float re = s[k], im = s[nfreq + k];
      v = re * re + im * im;
    }
    P[row * ldp + k] = v;
  }
}

__global__ __launch_bounds__(256) void log_mask_kernel(const float* __restrict__ mel, float* __restrict__ out, int64_t n,
                                                       int n_mels, int T, const int64_t* __restrict__ olens, float floor_) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / n_mels;
  const int b = (int)(row / T), t = (int)(row % T);
  out[i] = t < olens[b] ? logf(fmaxf(mel[i], floor_)) : 0.f;
}

// ---- SpecAug ------------------------------------------------------------------------------------------------------
// torch upsample_bicubic2d coefficients (A = -0.75), align_corners = False
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// espnet2 time_warp on utterance b's first len[b] frames: rows [0, center) are resized to [0, warped), rows
// [center, len) to [warped, len); the frequency axis keeps its size (weights (0,1,0,0) there).  center[b] == 0: the
// utterance is copied unwarped (too short for the window).  Frames t >= len[b] become 0 (espnet2 TimeWarp pads its
// per-utterance results with 0.0; with equal lengths len = T and nothing is cut).
__global__ __launch_bounds__(256) void time_warp_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T,
                                                        int F, const int64_t* __restrict__ centers,
                                                        const int64_t* __restrict__ warpeds, const int64_t* __restrict__ lens) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T * F) return;
  const int f = (int)(i % F);
  const int t = (int)((i / F) % T);
  const int b = (int)(i / ((int64_t)F * T));
  const int center = (int)centers[b], warped = (int)warpeds[b], L = (int)lens[b];
  if (t >= L) { y[i] = 0.f; return; }
  if (center == 0) { y[i] = x[i]; return; }
  int src0, in_len, out_len, tt;
  if (t < warped) { src0 = 0; in_len = center; out_len = warped; tt = t; }
  else { src0 = center; in_len = L - center; out_len = L - warped; tt = t - warped; }
  const float scale = (float)in_len / (float)out_len;
  const float real = scale * (tt + 0.5f) - 0.5f;
  const float fl = floorf(real);
  float w[4];
  cubic_coeffs(real - fl, w);
  const int i0 = (int)fl;
  const float* xb = x + ((int64_t)b * T + src0) * F + f;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = min(max(i0 - 1 + k, 0), in_len - 1);
    acc += w[k] * xb[(int64_t)idx * F];
  }
  y[i] = acc;
}

// espnet2 mask_along_axis for both axes in one pass: element (b, t, f) becomes 0 if f lies in one of the nf frequency
// bands [fpos, fpos + flen) of utterance b or t in one of its nt time bands
__global__ __launch_bounds__(256) void specaug_mask_kernel(float* __restrict__ x, int B, int T, int F,
                                                           const int64_t* __restrict__ fpos, const int64_t* __restrict__ flen,
                                                           int nf, const int64_t* __restrict__ tpos,
                                                           const int64_t* __restrict__ tlen, int nt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T * F) return;
  const int f = (int)(i % F);
  const int t = (int)((i / F) % T);
  const int b = (int)(i / ((int64_t)F * T));
  bool hit = false;
  for (int k = 0; k < nf; ++k) hit |= (fpos[b * nf + k] <= f) && (f < fpos[b * nf + k] + flen[b * nf + k]);
  for (int k = 0; k < nt; ++k) hit |= (tpos[b * nt + k] <= t) && (t < tpos[b * nt + k] + tlen[b * nt + k]);
  if (hit) x[i] = 0.f;
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_stft_frames(const float* wav, const float* window, float* frames, int32_t B, int64_t N, int32_t T,
                                 int32_t n_fft, int32_t hop, int32_t center, tavsr_stream_t stream) {
  TAVSR_REQUIRE(wav && window && frames, TAVSR_EINVAL, "stft_frames: null pointer");
  TAVSR_REQUIRE(B > 0 && T > 0 && n_fft > 0 && hop > 0, TAVSR_EINVAL, "stft_frames: bad sizes");
  const int pad = center ? n_fft / 2 : 0;
  TAVSR_REQUIRE(N > pad, TAVSR_EINVAL, "stft_frames: reflect padding needs more than n_fft/2 samples (got %ld)", (long)N);
  TAVSR_REQUIRE((int64_t)(T - 1) * hop + n_fft <= N + 2 * pad, TAVSR_EINVAL, "stft_frames: T frames do not fit the signal");
  hipLaunchKernelGGL(stft_frames_kernel, dim3((unsigned)((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream, wav, window,
                     frames, B, N, T, n_fft, hop, pad);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_power_spec(const float* spec, int64_t ld_spec, float* P, int32_t ldp, int32_t nfreq, int32_t B,
                                int32_t T, const int64_t* olens, tavsr_stream_t stream) {
  TAVSR_REQUIRE(spec && P && olens, TAVSR_EINVAL, "power_spec: null pointer");
  TAVSR_REQUIRE(ldp >= nfreq && ld_spec >= 2 * (int64_t)nfreq, TAVSR_EINVAL, "power_spec: leading dimensions too small");
  hipLaunchKernelGGL(power_spec_kernel, dim3((unsigned)((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream, spec, ld_spec, P,
                     ldp, nfreq, T, olens);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_log_mask(const float* mel, float* out, int32_t B, int32_t T, int32_t n_mels, const int64_t* olens,
                              float floor_, tavsr_stream_t stream) {
  TAVSR_REQUIRE(mel && out && olens, TAVSR_EINVAL, "log_mask: null pointer");
  const int64_t n = (int64_t)B * T * n_mels;
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(log_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mel, out, n, n_mels,
                     T, olens, floor_);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_time_warp(const float* x, float* y, int32_t B, int32_t T, int32_t F, const int64_t* center,
                               const int64_t* warped, const int64_t* lens, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y && x != y && center && warped && lens, TAVSR_EINVAL, "time_warp: null or aliased pointers");
  const int64_t n = (int64_t)B * T * F;
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(time_warp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, T, F,
                     center, warped, lens);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_specaug_mask(float* x, int32_t B, int32_t T, int32_t F, const int64_t* fpos, const int64_t* flen,
                                  int32_t nf, const int64_t* tpos, const int64_t* tlen, int32_t nt, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x, TAVSR_EINVAL, "specaug_mask: null pointer");
  TAVSR_REQUIRE((nf == 0 || (fpos && flen)) && (nt == 0 || (tpos && tlen)), TAVSR_EINVAL, "specaug_mask: null band arrays");
  const int64_t n = (int64_t)B * T * F;
  if (n <= 0 || (nf == 0 && nt == 0)) return TAVSR_OK;
  hipLaunchKernelGGL(specaug_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, T, F, fpos,
                     flen, nf, tpos, tlen, nt);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
