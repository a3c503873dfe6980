/* libtavsr_hip.so - C ABI of the MI355X (gfx950) Branchformer AVSR hot path.
 *
 * The reference (david-gimeno/tailored-avsr) has no FFI: its boundary for this path is the set of
 * torch.nn.Module.forward signatures listed in SURVEY.md section 8(a) and the ATen kernels they
 * dispatch to.  Each entry point below replaces the ATen/espnet leaf (or fused group of leaves)
 * named in its comment; tailored-avsr_amd/tavsr mirrors the reference's module classes on top
 * of these calls (same names, ctor kwargs, forward signatures, state_dict keys) - INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated;
 *   - the caller allocates every buffer including workspaces; the library never allocates or frees
 *     device memory, never synchronises the device, and enqueues work only on `stream`
 *     (a hipStream_t passed as void*), so calls may be captured into a hipGraph;
 *   - row-major fp32 unless stated; lengths/ids are int32 or int64 as stated;
 *   - returns 0 on success, a negative TAVSR_E* for a rejected descriptor (nothing was launched),
 *     a positive hipError_t if a launch failed; tavsr_last_error_string() describes the last error
 *     of the calling thread.
 */
#ifndef TAVSR_H_
#define TAVSR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tavsr_stream_t; /* hipStream_t */

enum { TAVSR_OK = 0, TAVSR_EINVAL = -1, TAVSR_EALIGN = -2, TAVSR_EUNSUPPORTED = -3 };
enum { TAVSR_ACT_NONE = 0, TAVSR_ACT_RELU = 1, TAVSR_ACT_SWISH = 2, TAVSR_ACT_GELU = 3 };

int tavsr_version(void);                     /* ABI version, bumped on any signature change */
const char* tavsr_last_error_string(void);   /* host string, thread-local */

/* ---------------------------------------------------------------------------------------------
 * GEMM (fp32 MFMA, v_mfma_f32_32x32x2_f32).  Replaces torch.nn.Linear / torch.matmul wherever the
 * reference's leaves call them: PositionwiseFeedForward.w_1/w_2 (encoder_layer.py:193-194,313-314),
 * attention linear_q/k/v/out/pos and the score/context matmuls (espnet attention.py, called at
 * encoder_layer.py:208), cgMLP channel_proj1/2 (encoder_layer.py:220), merge_proj (:291-293),
 * Conv2dSubsampling conv (as im2col GEMM) and out Linear (encoder.py:149-155,364), ctc_lo
 * (src/ctc/ctc.py:143), decoder projections (espnet_model.py:557-560) - and their backward.
 *
 *   C[z][m][n] = R[z][m][n] + alpha * act( sum_k A(z,m,k) * B(z,k,n) + bias[n] ) * act'(DZ[z][m][n])
 *
 *   A(z,m,k) = A[z*sA + (a_kmajor ? k*lda + m : m*lda + k)]
 *   B(z,k,n) = B[z*sB + (b_kmajor ? k*ldb + n : n*ldb + k)]     (b_kmajor=0 is torch's W[N,K])
 *   z = (z1, z2), z1 < nb1, z2 < nb2, offsets z1*s?1 + z2*s?2 (elements).
 *   bias, R, Z, DZ may be NULL.  Z receives the pre-activation (acc + bias) with C's layout.
 *   DZ uses C's layout/strides; dact selects which activation's derivative is applied.
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_gemm_desc {
  int32_t M, N, K;
  int32_t a_kmajor, b_kmajor;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int32_t nb1, nb2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  const float* bias;
  int32_t act;
  float alpha;
  float* Z;
  const float* R; int64_t ldr; int64_t sR1, sR2;
  const float* DZ; int32_t dact;
} tavsr_gemm_desc;

int tavsr_gemm(const tavsr_gemm_desc* desc, tavsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAVSR_H_ */
