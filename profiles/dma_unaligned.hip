// does global_load_lds ... 16 bytes accept a source address that is only 4- (or 8-) byte aligned on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;
__global__ void k(const float* src, float* out, int shift) {
  __shared__ __attribute__((aligned(1024))) float sm[64 * 4];
  const int lane = threadIdx.x;
  const float* p = src + lane * 6 + shift;          // 24-byte stride: alignment = 4 * (shift % 4) mod 16 varies per lane
  __builtin_amdgcn_global_load_lds((glb_float*)p, (lds_float*)sm, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = sm[lane * 4 + j];
}
int main() {
  const int n = 64 * 6 + 16;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(o, 0, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (r[l * 4 + j] != (float)(l * 6 + shift + j)) ++bad;
    printf("shift %d: err=%d bad=%d  sample lane1: %g %g %g %g (want %d..)\n", shift, (int)e, bad, r[4], r[5], r[6], r[7], 6 + shift);
  }
  return 0;
}
