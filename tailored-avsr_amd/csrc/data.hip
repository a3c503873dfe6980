// Batch assembly on the device: the per-sample video pipeline of avsr_main.py:168-179
// (Normalise, Normalise, TimeMasking, RandomCrop / CenterCrop, RandomHorizontalFlip; src/transforms/video_transforms.py:
// 59-147) and the padding of src/utils/avsr_dataloader.py:102-142 as ONE pass per clip: every output frame of the padded
// batch row is produced from the raw uint8 / float lip frames exactly once (crop window, mirrored columns, frame
// re-indexing of VideoSpeedRate, the chain of (x - mean) / std steps, masked frames replaced by the clip's mean frame,
// frames past the clip's length set to the padding value).  HBM-bound: 1 byte read + 4 bytes written per output pixel.
// AddNoise (src/transforms/audio_transforms.py:74-139): noise mixed at a signal-to-noise ratio, one workgroup per clip.
#include "common.h"

namespace tavsr {

struct VideoPrepArgs {
  const void* src;        // [Ts][H][W] uint8 (is_u8) or float
  const int32_t* frames;  // nullable: source frame of output frame t (VideoSpeedRate)
  const uint8_t* masked;  // nullable: 1 = frame t is replaced by the mean frame (TimeMasking)
  float* mean_frame;      // [th*tw] workspace (only read when masked != nullptr)
  float* dst;             // [Tpad][th][tw] row of the padded batch
  float mean[4], std[4];  // the Normalise steps, applied in order
  int n_affine, is_u8;
  int T, Tpad, H, W, y0, x0, th, tw, flip;
  float pad;
};

__device__ __forceinline__ float video_value(const VideoPrepArgs& a, int t, int y, int x) {
  const int ts = a.frames ? a.frames[t] : t;
  const int xs = a.x0 + (a.flip ? a.tw - 1 - x : x);
  const int64_t o = ((int64_t)ts * a.H + (a.y0 + y)) * a.W + xs;
  float v = a.is_u8 ? (float)reinterpret_cast<const uint8_t*>(a.src)[o] : reinterpret_cast<const float*>(a.src)[o];
  for (int k = 0; k < a.n_affine; ++k) v = (v - a.mean[k]) / a.std[k];
  return v;
}

// mean over the clip's T frames of the normalised, cropped, mirrored pixels (TimeMasking's video_data.mean(axis=0))
__global__ __launch_bounds__(256) void video_mean_frame_kernel(const VideoPrepArgs a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.th * a.tw) return;
  const int y = p / a.tw, x = p % a.tw;
  float s = 0.f;
  for (int t = 0; t < a.T; ++t) s += video_value(a, t, y, x);
  a.mean_frame[p] = s / (float)a.T;
}

__global__ __launch_bounds__(256) void video_prep_kernel(const VideoPrepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int P = a.th * a.tw;
  if (i >= (int64_t)a.Tpad * P) return;
  const int t = (int)(i / P), p = (int)(i % P);
  float v;
  if (t >= a.T) v = a.pad;
  else if (a.masked && a.masked[t]) v = a.mean_frame[p];
  else v = video_value(a, t, p / a.tw, p % a.tw);
  a.dst[i] = v;
}

// out = audio + (inv_snr * noise) * sqrt(P_audio / P_noise),  P = mean of squares (audio_transforms.py:126-133)
__global__ __launch_bounds__(1024) void add_noise_kernel(const float* __restrict__ audio, const float* __restrict__ noise,
                                                         float* __restrict__ out, int64_t n, float inv_snr) {
  __shared__ double red[2][1024];
  double sa = 0.0, sn = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float a = audio[i], b = noise[i];
    sa += (double)a * a;
    sn += (double)b * b;
  }
  red[0][threadIdx.x] = sa;
  red[1][threadIdx.x] = sn;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  const float pa = (float)(red[0][0] / (double)n), pn = (float)(red[1][0] / (double)n);
  const float scale = sqrtf(pa / pn);
  for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = audio[i] + (inv_snr * noise[i]) * scale;
}

}  // namespace tavsr

using namespace tavsr;

extern "C" int tavsr_video_prep(const void* src, int32_t is_u8, int32_t Ts, int32_t H, int32_t W, const int32_t* frames,
                                int32_t T, int32_t y0, int32_t x0, int32_t th, int32_t tw, int32_t flip, const float* mean,
                                const float* std, int32_t n_affine, const uint8_t* masked, float* mean_frame_ws, float* dst,
                                int32_t Tpad, float pad, tavsr_stream_t stream) {
  TAVSR_REQUIRE(src && dst, TAVSR_EINVAL, "video_prep: null pointer");
  TAVSR_REQUIRE(T >= 0 && Tpad >= T && Ts > 0 && (frames || T <= Ts), TAVSR_EINVAL, "video_prep: bad frame counts");
  TAVSR_REQUIRE(th > 0 && tw > 0 && y0 >= 0 && x0 >= 0 && y0 + th <= H && x0 + tw <= W, TAVSR_EINVAL,
                "video_prep: crop window [%d+%d, %d+%d] outside the %d x %d frame", y0, th, x0, tw, H, W);
  TAVSR_REQUIRE(n_affine >= 0 && n_affine <= 4 && (n_affine == 0 || (mean && std)), TAVSR_EINVAL, "video_prep: at most 4 Normalise steps");
  TAVSR_REQUIRE(!masked || mean_frame_ws, TAVSR_EINVAL, "video_prep: time masking needs the mean-frame workspace");
  VideoPrepArgs a{};
  a.src = src; a.frames = frames; a.masked = masked; a.mean_frame = mean_frame_ws; a.dst = dst;
  for (int k = 0; k < n_affine; ++k) { a.mean[k] = mean[k]; a.std[k] = std[k]; }
  a.n_affine = n_affine; a.is_u8 = is_u8;
  a.T = T; a.Tpad = Tpad; a.H = H; a.W = W; a.y0 = y0; a.x0 = x0; a.th = th; a.tw = tw; a.flip = flip; a.pad = pad;
  const int P = th * tw;
  hipStream_t s = (hipStream_t)stream;
  if (masked && T > 0) {
    hipLaunchKernelGGL(video_mean_frame_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
  }
  const int64_t total = (int64_t)Tpad * P;
  if (total > 0) {
    hipLaunchKernelGGL(video_prep_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, a);
    TAVSR_LAUNCH_CHECK();
  }
  return TAVSR_OK;
}

// Band-limited resampling of a mono waveform by a rational-free factor (the audio SpeedRate augmentation,
// src/transforms/audio_transforms.py:141-178: sox "speed f" + "rate 16000" = play the clip f times faster and resample back):
//   y[n] = sum_k x[k] c sinc(c (n f - k)) w((n f - k) / W),  c = rolloff min(1, 1 / f),  W = zeros / c,  w = Kaiser(beta)
// One thread per output sample, taps k in [n f - W, n f + W] clipped to the clip; the window's I0 by its power series.
__device__ __forceinline__ double bessel_i0(double x) {
  double s = 1.0, t = 1.0;
  const double q = 0.25 * x * x;
  for (int k = 1; k < 64; ++k) {
    t *= q / ((double)k * (double)k);
    s += t;
    if (t < 1e-17 * s) break;
  }
  return s;
}

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, int64_t n_in, float* __restrict__ y, int64_t n_out,
                                                            double step, double cutoff, double halfw, double beta, double inv_i0b) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const double t = (double)n * step;
  int64_t k0 = (int64_t)ceil(t - halfw), k1 = (int64_t)floor(t + halfw);
  k0 = k0 < 0 ? 0 : k0;
  k1 = k1 > n_in - 1 ? n_in - 1 : k1;
  double acc = 0.0;
  for (int64_t k = k0; k <= k1; ++k) {
    const double u = t - (double)k, r = u / halfw;
    const double a = cutoff * u;
    const double sn = a == 0.0 ? 1.0 : sinpi(a) / (3.14159265358979323846 * a);
    const double w = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) * inv_i0b;
    acc += (double)x[k] * cutoff * sn * w;
  }
  y[n] = (float)acc;
}

static double host_i0(double x) {
  double s = 1.0, t = 1.0;
  const double q = 0.25 * x * x;
  for (int k = 1; k < 64; ++k) {
    t *= q / ((double)k * (double)k);
    s += t;
    if (t < 1e-17 * s) break;
  }
  return s;
}

extern "C" int64_t tavsr_resample_len(int64_t n_in, double factor) { return factor > 0.0 ? (int64_t)llround((double)n_in / factor) : 0; }

extern "C" int tavsr_resample_sinc(const float* x, int64_t n_in, float* y, int64_t n_out, double factor, double rolloff, int32_t zeros,
                                   double beta, tavsr_stream_t stream) {
  TAVSR_REQUIRE(x && y, TAVSR_EINVAL, "resample_sinc: null pointer");
  TAVSR_REQUIRE(factor > 0.0 && rolloff > 0.0 && rolloff <= 1.0 && zeros > 0 && zeros <= 256 && beta >= 0.0, TAVSR_EINVAL,
                "resample_sinc: factor > 0, rolloff in (0, 1], 1..256 zero crossings, beta >= 0");
  if (n_in <= 0 || n_out <= 0) return TAVSR_OK;
  const double cutoff = rolloff * (factor > 1.0 ? 1.0 / factor : 1.0);
  hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n_in, y, n_out, factor,
                     cutoff, (double)zeros / cutoff, beta, 1.0 / host_i0(beta));
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}

extern "C" int tavsr_add_noise(const float* audio, const float* noise, float* out, int64_t n, float inv_snr, tavsr_stream_t stream) {
  TAVSR_REQUIRE(audio && noise && out, TAVSR_EINVAL, "add_noise: null pointer");
  if (n <= 0) return TAVSR_OK;
  hipLaunchKernelGGL(add_noise_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, audio, noise, out, n, inv_snr);
  TAVSR_LAUNCH_CHECK();
  return TAVSR_OK;
}
