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

/* Stream-plumbing test hooks (csrc/probe.hip).  The reference runs its whole step on ONE queue
 * (src/models/espnet_model.py:258-356 on torch's current stream); this library's callers spread a step over forked queues,
 * and results must not depend on that.  tavsr_spin occupies `stream` for `us` microseconds (one idle wave), tavsr_race_probe
 * arms the same delay inside the entry points that fork a second queue themselves (tavsr_branchformer_layer_fwd):
 * mode 0 = head of the forked section, 1 = behind the join on the calling queue, 2 = alternately per call; us = 0 disarms. */
int tavsr_spin(float us, tavsr_stream_t stream);
int tavsr_race_probe(float us, int mode);
/* a queue of the host side's own (hipStreamCreateWithFlags, non-blocking) for the side of a fork.  torch hands its streams out of a pool
 * of 32 per device, round-robin: the 33rd `torch.cuda.Stream()` of a process IS the first one again, so a side stream taken from the pool can
 * turn up later as somebody's capturing stream - and the host's fork registry, keyed by the raw handle, would then order a capture behind a
 * stream outside it.  The handle lives as long as the process. */
int tavsr_stream_create(tavsr_stream_t* out);
/* tuning aid (scripts/launch_floor.py): a launch of `grid` x `block` threads that does nothing (kind 0), one 16-byte read + write per
 * thread (1), or that plus a barrier and a dependent second read (2) - the floor under a one-token step's dependent launches */
int tavsr_probe_launch(int32_t kind, int32_t grid, int32_t block, float* buf, int64_t n, tavsr_stream_t stream);
/* tuning aid (scripts/seam_bench.py): `nphase` dependent phases (every workgroup writes per_wg floats, then reads what another workgroup
 * wrote in the phase before) as nphase launches (kind 0) or as ONE launch with XCD-hierarchical grid barriers between the phases (kind 1;
 * kind + 256 * (s + 1): a phase reads the slab of workgroup (g + s) % grid instead of (g + 97) % grid; ctl: 2304 zeroed bytes kept across calls, epoch0 = 0; grid <= 256: one workgroup per compute unit) - what a phase
 * boundary inside a persistent layer kernel costs against the launch boundary it would replace */
int tavsr_probe_seam(int32_t kind, int32_t grid, int64_t per_wg, int32_t nphase, float* buf, void* ctl, uint32_t epoch0,
                     tavsr_stream_t stream);
/* Box calibration for bench.py's `box` object: `blocks` workgroups of 4 waves, each wave issuing 4 * iters independent
 * v_mfma_f32_32x32x2_f32 (4096 FLOP each) and nothing else - the fp32 matrix rate this device holds, against which a 5 %
 * difference between two boxes of a pool can be told from a regression.  `sink`: one device word (never written). */
int tavsr_mfma_peak_f32(int32_t iters, int32_t blocks, float* sink, tavsr_stream_t stream);
/* ... with operands from memory: `operands` = 16 x 256 floats (8 A / B pairs per lane, [2 j + {0, 1}][lane of the workgroup]), a
 * different pair for every consecutive instruction; same FLOP count.  Random operands toggle the multiplier inputs the way
 * activations do - the rate the device's power management leaves for real data (bench.py `box.fp32_mfma_tflops_data`). */
int tavsr_mfma_peak_f32_data(int32_t iters, int32_t blocks, const float* operands, float* sink, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM (fp32 MFMA, v_mfma_f32_32x32x2_f32).  Replaces torch.nn.Linear / torch.matmul wherever the
 * reference's leaves call them: PositionwiseFeedForward.w_1/w_2 (encoder_layer.py:193-194,313-314),
 * attention linear_q/k/v/out/pos and the score/context matmuls (espnet attention.py, called at
 * encoder_layer.py:208), cgMLP channel_proj1/2 (encoder_layer.py:220), merge_proj (:291-293),
 * Conv2dSubsampling conv (as im2col GEMM) and out Linear (encoder.py:149-155,364), ctc_lo
 * (src/ctc/ctc.py:143), decoder projections (espnet_model.py:557-560) - and their backward.
 *
 *   C[z][m][n] = R[z][m][n] + alpha * dropout( act( sum_k A(z,m,k) * B(z,k,n) + bias[n] ) * act'(DZ[z][m][n]) )
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
  float* ws; int64_t ws_floats;   /* optional split-K workspace (>= tavsr_gemm_ws(desc) floats), may be NULL */
  float* a_rowsum;                /* optional [M]: alpha * sum_k A(m,k) (nb1*nb2 == 1).  With a_kmajor (weight
                                     gradient dW = dY^T X) this is the bias gradient sum_rows dY - fused, so no
                                     separate column-sum pass over dY is needed */
  /* implicit 3x3 / stride 1 / pad 1 convolution over a channels-last image X [images*H*W][C] (the ResNet trunk of
     src/frontend/conv3d_resnet18/modules/resnet.py:89-106), no im2col matrix:
       conv_mode 1: A is X (row-major, lda = C), K = 9*C: A(m, tap*C + c) = X[m + (tap/3-1)*W + (tap%3-1)][c], 0 outside
                    the image - forward with B = weights [Cout][9*C], data gradient with B = flipped weights;
       conv_mode 2: B is X (k-major, ldb = C), N = 9*C, a_kmajor: weight gradient dW[co][tap*C + c] = sum_m dY[m][co] *
                    patch(m, tap, c).
     conv_zero: >= 16 readable zero bytes (16-byte aligned) that out-of-image loads are pointed at.  0: plain GEMM.
     conv_stride s (0 = 1) and conv_taps (0 = 9, or 1): the strided 3x3 / pad 1 and 1x1 / pad 0 convolutions of the blocks
     that halve the maps (resnet.py:68-87 downsample, :95-97 conv1).  conv_H x conv_W is always the INPUT image; the rows
     of the patch operand are the OUTPUT pixels (n, ho, wo), Ho = (H-1)/s + 1, centred on input pixel (s*ho, s*wo);
     K (mode 1) / N (mode 2) = taps * C.  conv_taps = 90: the 3x3 window WITHOUT padding (espnet Conv2dSubsampling's second
     convolution, subsampling.py: Conv2d(odim, odim, 3, 2)): Ho = (H-3)/s + 1, output pixel centred on (s*ho + 1, s*wo + 1).
     Conv3d stem of the lip front-end (src/frontend/conv3d_resnet18/conv3d_resnet18.py:48-57: 1 -> Cout channels, kernel
     (5,7,7), stride (1,2,2), padding (2,3,3), no bias) over clips X [clips][conv_C frames][conv_H][conv_W] (even H, W), the
     245 taps padded to 256, every element gathered by the loader - no patch matrix:
       conv_mode 4: A is X, M = output pixels (clip, t, ho, wo), K = 256: A(m, (kt*7 + kh)*7 + kw) = X[clip][t + kt - 2]
                    [2 ho - 3 + kh][2 wo - 3 + kw] (0 outside the clip / frame, 0 for k >= 245); B = weights [Cout][256];
       conv_mode 5: B is X (TN layout, a_kmajor and b_kmajor), N = 256, K = output pixels (a multiple of 32): weight gradient
                    dW[co][tap] = sum_m dY[m][co] * patch(m, tap); columns >= 245 come out 0.
     lda (mode 4) / ldb (mode 5) are ignored (pass a multiple of 4). */
  int32_t conv_mode, conv_H, conv_W, conv_C;
  const float* conv_zero;
  int32_t conv_stride, conv_taps;
  /* train-mode dropout in the epilogue (drop_p > 0): element (m, n) keeps iff word (n & 3) of the Philox counter
     drop_offset/4 + (m*N + n)/4 of *drop_seed is >= drop_p * 2^32 - exactly the mask tavsr_dropout draws for the
     contiguous [M][N] result at that offset, so either side of a forward / backward pair may be the fused or the
     stand-alone form (x + s * dropout(f(x)) of encoder_layer.py:194,309,314; PositionwiseFeedForward's inner dropout and
     its backward mask * act'(z)).  Unbatched plain GEMMs on the 16-byte path only (N % 4 == 0, aligned operands,
     K % 32 == 0 or the K-tail variant): otherwise TAVSR_EUNSUPPORTED and nothing is launched. */
  float drop_p;
  const uint64_t* drop_seed;
  uint64_t drop_offset;
  /* optional [M][ceil(N / 64)][2]: per row and 64-column tile, sum and sum of squares of the stored C values (after bias /
     activation / alpha / residual) - LayerNorm statistics of the result without another pass over it (cgMLP: the gate half of
     channel_proj1's output, consumed by tavsr_csgu_fwd).  Unbatched problems on the 16-byte path, no K split; otherwise
     TAVSR_EUNSUPPORTED and nothing is launched. */
  float* rowstat;
  /* with rowstat: per row and 64-column tile the two weighted sums (sum_n C[m][n] * rowdot_a[n], sum_n C[m][n] * rowdot_b[n]) instead of
     (sum, sum of squares) - the learned-average merge's pooling / branch-weight projections of a branch output (encoder_layer.py:
     246-280: pooling_proj_k, weight_proj_k are Linear(d, 1)) taken where the branch's last Linear stores its rows, so that
     tavsr_merge_proj_fwd does not re-read both branch outputs of the whole utterance per workgroup.  [N] each, 16-byte aligned; both or
     neither. */
  const float* rowdot_a;
  const float* rowdot_b;
} tavsr_gemm_desc;

int tavsr_gemm(const tavsr_gemm_desc* desc, tavsr_stream_t stream);
/* floats of workspace with which tavsr_gemm will split K over workgroups (0: not needed).  Few-tile, long-K
 * problems (weight gradients, K = B*T) are split so the whole chip works; each slice stores an fp32 slab and a
 * second kernel sums the slabs in slice order (deterministic) and applies the epilogue. */
int64_t tavsr_gemm_ws(const tavsr_gemm_desc* desc);
/* tavsr_gemm of a plain unbatched Linear (bias / activation / alpha / residual; no dropout, Z, DZ, a_rowsum, rowstat, convolution)
 * and the LayerNorm that follows it: ln_out[m][:N] = LN(C[m][:N]) * gamma + beta (row stride ld_ln; eps as torch).  Where the GEMM
 * splits K, the slab sum, the epilogue and the LayerNorm are ONE launch (one wave per row) instead of an epilogue launch plus a
 * LayerNorm launch - the x + f(x) -> norm(x) seams of a batched one-token decoder / LM step (espnet decoder_layer.py,
 * encoder_layer.py: every sub-block starts with a LayerNorm of the residual stream).  N % 4 == 0, N <= 2048, 16-byte aligned rows. */
int tavsr_gemm_ln(const tavsr_gemm_desc* desc, const float* gamma, const float* beta, float eps, float* ln_out, int64_t ld_ln,
                  tavsr_stream_t stream);
/* up to 12 independent, unbatched problems of ONE layout (a_kmajor, b_kmajor) in one launch, no K split: the weight
 * gradients of a layer (16-128 output tiles each) fill the chip together.  Every problem must satisfy the fast
 * kernel's conditions (16-byte aligned operands and leading dimensions, K %% 32 == 0, row-contiguous operands with a
 * row count %% 4 == 0); otherwise TAVSR_EUNSUPPORTED is returned, nothing is launched and the caller issues the
 * problems one by one with tavsr_gemm. */
int tavsr_gemm_grouped(const tavsr_gemm_desc* descs, int32_t n, tavsr_stream_t stream);
/* tuning/bench entry: force tile configuration cfg (see kCfgs in csrc/gemm.hip; BK = 32) and a K split (<= 1: none; needs ws/sync large enough) instead of the planner's choice. */
int tavsr_gemm_tune(const tavsr_gemm_desc* desc, int32_t cfg, int32_t nsplit, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (espnet LayerNorm = torch.nn.LayerNorm(eps=1e-12); the five norms of
 * encoder_layer.py:102-109, csgu.norm, encoder.py:313 after_norm, decoder norms).  D % 4 == 0,
 * D <= 1024, rows 16-byte aligned.  mean/rstd [M] are saved for backward (may be NULL in fwd).
 * bwd: dx = dx_add + LN'(dy) (dx_add may be NULL; lets the caller fold a residual-branch gradient
 * in), dgamma/dbeta are overwritten or accumulated; ws >= tavsr_layernorm_bwd_ws(M, D) floats.
 * ------------------------------------------------------------------------------------------- */
int tavsr_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                        float* y, int64_t ldy, float* mean, float* rstd, int32_t M, int32_t D,
                        tavsr_stream_t stream);
int64_t tavsr_layernorm_bwd_ws(int32_t M, int32_t D);
int tavsr_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                        const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd,
                        float* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                        float* ws, int32_t M, int32_t D, tavsr_stream_t stream);

/* Column sums out[n] (+)= scale * sum_m x[m*ldx + n]: bias gradients of every Linear, pos_bias_u/v
 * gradients.  ws >= tavsr_colsum_ws(M, N) floats.  Deterministic (two-stage, no atomics). */
/* main pass only: dx and per-block partials (dgamma | dbeta) at ws[blk*ws_ld + 0..2D), blk < tavsr_layernorm_bwd_ws(M,D)/(2D);
 * several LayerNorms of one backward node write into one slab and share one tavsr_sum_partials launch */
int tavsr_layernorm_bwd_partial(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd, float* dx,
                                int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D, tavsr_stream_t stream);
/* ... and with a second output dx_drop [M][D] = dx * mask / keep under the tavsr_dropout mask of a contiguous [M][D] tensor at
 * `offset`: the masked gradient that the backward of the next residual block's branch starts from
 * (x + s * dropout(f(x)), encoder_layer.py:194,309,314) - saves that block's stand-alone mask launch. */
int tavsr_layernorm_bwd_partial_drop(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                     const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd, float* dx,
                                     int64_t lddx, float* ws, int64_t ws_ld, int32_t M, int32_t D, float* dx_drop, float p_drop,
                                     const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream);
/* tavsr_layernorm_bwd for x = act(z): dx is written as the gradient w.r.t. z (the LayerNorm's dx times act'(z)): cgMLP's gate half
 * LayerNorm(gelu(channel_proj1(x))[..., C:]) (espnet cgmlp.py ConvolutionalSpatialGatingUnit.norm; encoder_layer.py:220) */
int tavsr_layernorm_bwd_act(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd,
                            const float* gamma, float* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                            int32_t M, int32_t D, const float* z, int64_t ldz, int32_t act, tavsr_stream_t stream);
int64_t tavsr_colsum_ws(int32_t M, int32_t N);
/* out = x + y, sum_x[n] = sum_m x[m][n], sum_y[n] = sum_m y[m][n] in two launches (dQ = dQu + dQv with the pos_bias_u/v
 * gradients of RelPositionMultiHeadedAttention); ws >= 2 * tavsr_colsum_ws(M, N) floats */
int tavsr_add2_colsum(const float* x, int64_t ldx, const float* y, int64_t ldy, float* out, int64_t ldo, int32_t M, int32_t N,
                      float* sum_x, float* sum_y, float* ws, tavsr_stream_t stream);
/* (sum_x == sum_y == NULL: the reduction launch is left out; the column sums stay as tavsr_colsum_ws(M, N) / N partial rows of 2 N floats in ws
 * for a later tavsr_sum_partials2(ws, rows, 2 N, sum_x, N, sum_y, N, 0, stream) - bias gradients have no reader inside a backward pass) */
int tavsr_colsum(const float* x, int64_t ldx, int32_t M, int32_t N, float scale, float* out,
                 int32_t accumulate, float* ws, tavsr_stream_t stream);
/* out[i] (+)= sum_{p < nparts} part[p*stride + i], i < n */
int tavsr_sum_partials(const float* part, int32_t nparts, int64_t stride, float* out, int32_t n,
                       int32_t accumulate, tavsr_stream_t stream);
/* columns [0, n1) of the slab to out1, [n1, n1 + n2) to out2, one launch */
int tavsr_sum_partials2(const float* part, int32_t nparts, int64_t stride, float* out1, int32_t n1, float* out2, int32_t n2,
                        int32_t accumulate, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention glue (espnet attention.py: RelPositionMultiHeadedAttention.forward / rel_shift /
 * MultiHeadedAttention.forward_attention; called at encoder_layer.py:208 and by the decoder).
 * Score tensors are [H, B, T1, ld_s] fp32 with ld_s >= T2 (rows padded to a multiple of 4 floats so the
 * GEMMs around them use 16-byte loads; likewise ld_w >= W); the QK^T, (q+v)P^T, PV products are tavsr_gemm calls.
 *   add_head_bias: qu = q + pos_bias_u, qv = q + pos_bias_v   (q rows strided by ldq, D = H*d_k)
 *   softmax_fwd  : attn = softmax_j((ac + rel_shift(bd)) * scale) over keys j < klens[b]
 *                  (and j <= i when causal), exactly 0 on masked keys; bd may be NULL
 *                  (plain attention); bd is the UNSHIFTED [H,B,T,2T-1] product, W = 2T-1:
 *                  rel_shift(bd)[i,j] = bd[i, T-1-i+j]
 *   softmax_bwd  : ds = attn*(dattn - <attn,dattn>)*scale and (optional) its un-shifted copy
 *                  ds_skew[i, T-1-i+j] = ds[i,j], 0 elsewhere, that feeds the positional GEMMs
 * ------------------------------------------------------------------------------------------- */
int tavsr_add_head_bias(const float* q, int64_t ldq, const float* u, const float* v, float* qu, float* qv,
                        int64_t M, int32_t D, tavsr_stream_t stream);
int tavsr_softmax_fwd(const float* ac, const float* bd, const int64_t* klens, float* attn, int32_t H,
                      int32_t B, int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale,
                      int32_t causal, tavsr_stream_t stream);
int tavsr_softmax_bwd(const float* attn, const float* dattn, float* ds, float* ds_skew, int32_t H,
                      int32_t B, int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale,
                      tavsr_stream_t stream);
/* tavsr_softmax_fwd / _bwd with the dropout of the attention probabilities (espnet forward_attention: self.dropout(attn))
 * folded in: pv = dropout(attn, p) with exactly the mask tavsr_dropout(attn, ..., offset) draws (rows padded to ld_s % 4 == 0);
 * the backward takes the gradient w.r.t. pv and regenerates the mask. */
int tavsr_softmax_dropout_fwd(const float* ac, const float* bd, const int64_t* klens, float* attn, float* pv, int32_t H, int32_t B,
                              int32_t T1, int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale, int32_t causal, float p,
                              const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream);
int tavsr_softmax_dropout_bwd(const float* attn, const float* dpv, float* ds, float* ds_skew, int32_t H, int32_t B, int32_t T1,
                              int32_t T2, int32_t W, int64_t ld_s, int64_t ld_w, float scale, float p, const uint64_t* seed_dev,
                              uint64_t offset, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused attention core (csrc/attn_fused.hip) - replaces, in ONE launch per direction, the add_head_bias / QK^T /
 * (q+v)P^T / rel_shift / mask / softmax / dropout / PV chain of espnet RelPositionMultiHeadedAttention.forward
 * (called at src/encoder/branchformer/encoder_layer.py:196-210, src/encoder/audiovisual/tailored/encoder_layer.py:
 * 185-196,232-243) and of MultiHeadedAttention.forward_attention (decoder self / source attention): no [H,B,T,T] score
 * tensor ever exists in HBM.  Head size 64.  q / k / v are row buffers whose row (b*T + t) holds head h at columns
 * h*64 .. h*64+63 (q, k, v may be three column windows of one [B*T, 3*256] projection output: pass the window base
 * pointers and the shared row stride).
 *   scores[i][j] = ((q_i + bias_u) . k_j + (q_i + bias_v) . pos[T1-1-i+j]) * scale     (second term iff pos != NULL)
 *   attn = softmax over keys j < klens[b] (and j <= i when causal), exactly 0 on masked keys; ctx = dropout(attn) v
 * Dropout: element (b, h, i, j) uses Philox counter drop_offset/4 + ((b*H + h)*T1 + i) * roundup4(T2)/4 + j/4, word j%4,
 * of the device seed - the backward regenerates it; a call consumes B*H*T1*roundup4(T2) counter elements.
 * tavsr_attn_fwd writes ctx [B*T1][ldo] and lse [B*H][T1] (log-sum-exp of the scaled masked scores, +inf for a row
 * without keys), all the backward needs besides its inputs.
 * tavsr_attn_bwd: dq = d/d(q+bias_u) rows, dqv = d/d(q+bias_v) rows (rel-pos only; d/dq = dq + dqv, the bias gradients
 * are their column sums), dk / dv laid out like k / v (every row written), and for rel-pos ds_skew [H][B][T1][ldw]:
 * ds_skew[i][T1-1-i+j] = dscores[i][j] - the caller passes it ZERO-FILLED, the kernel writes the band; the positional
 * projection gradient is then one tavsr_gemm per head over K = B*T1 rows.
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_attn_desc {
  const float *q, *k, *v;
  int64_t ldq, ldk, ldv;
  const float* pos;          /* [2*T1-1][ldp] = linear_pos(pos_emb), or NULL */
  int64_t ldp;
  const float *bias_u, *bias_v; /* [H*64] or NULL */
  const int64_t* klens;      /* [B] or NULL */
  int32_t B, H, T1, T2, dk;
  float scale;
  int32_t causal;
  float p_drop;              /* 0: no dropout */
  const uint64_t* seed_dev;
  uint64_t drop_offset;
} tavsr_attn_desc;
int tavsr_attn_fwd(const tavsr_attn_desc* d, float* ctx, int64_t ldo, float* lse, tavsr_stream_t stream);
int tavsr_attn_bwd(const tavsr_attn_desc* d, const float* dctx, const float* ctx, int64_t ldo, const float* lse, float* dq,
                   float* dqv, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* ds_skew, int64_t ldw,
                   tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Feed-forward block, streaming form (csrc/ffn2.hip) - the same espnet PositionwiseFeedForward residual block
 * (src/encoder/branchformer/encoder_layer.py:191-194,311-314; src/encoder/audiovisual/tailored/encoder_layer.py:173-175,
 * 211-213) for d_model 256, hidden N1 % 32 == 0, weights in torch Linear layout (w1 [N1][256], w2 [256][N1]):
 *     y = res + scale * dropout(w2 dropout(act(w1 LN(x) + b1)) + b2)          (res NULL: res = x)
 *     ln2_out[k] = LayerNorm(y) * ln2_w[k] + ln2_b[k], k = 0, 1 (optional)    - the norms the consumers of y start with
 *                                                                               (encoder_layer.py:197,215 after the macaron
 *                                                                               block, :316 norm_final after the second)
 * Two launches: the chain kernel (a workgroup owns a block of 128 rows, wave w its 32-row tile, LayerNorm straight into the
 * B-operand registers; unit of work 128 rows x 32 hidden units: z^T tile = w1[unit] x LN(x)^T, whose accumulators - hidden
 * unit on the registers, row on the lane - are, after the activation, the A operand of the second product, so the hidden
 * activations never leave the registers; weight tiles shared by the four waves through one LDS ring filled by LDS-DMA; the
 * flat unit list cut into equal contiguous ranges, one per workgroup) and the finishing kernel (fixed-order sum of a row
 * block's partials, bias, dropout, scale, residual, the optional LayerNorms).
 * Saved for a backward pass (all optional): n_out = LN(x) [M][256], mean / rstd [M], z / h = pre-activations and dropped
 * activations, buffers of roundup128(M) rows x N1 (whole 128-row blocks are stored); ln2_mean / ln2_rstd [M].
 * Dropout: both sites use the tavsr_dropout mapping (element e of the contiguous [M][N1] / [M][256] result = word e & 3 of
 * Philox counter offset/4 + e/4), i.e. the masks the GEMM-epilogue path draws from the same offsets.
 * ws: tavsr_ffn2_ws(M, 256, N1) floats.  Unsupported shapes: TAVSR_EUNSUPPORTED (callers keep the GEMM launches).
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_ffn_desc {
  int32_t M, D, N1, act;
  float scale, eps;
  const float* x;            /* [M][ldx] */
  int64_t ldx;
  const float* res;          /* residual rows [M][ldr]; NULL: x */
  int64_t ldr;
  const float *ln_w, *ln_b, *w1, *b1, *w2, *b2;
  float* y;                  /* [M][256] */
  float p_drop;
  const uint64_t* seed;      /* device */
  uint64_t offset_in, offset_out;
  float *n_out, *mean, *rstd, *z, *h;
  const float* ln2_w[2];
  const float* ln2_b[2];
  float* ln2_out[2];
  float *ln2_mean, *ln2_rstd;
  float ln2_eps;
  float* ws;
  int64_t ws_floats;
} tavsr_ffn_desc;
int64_t tavsr_ffn2_ws(int32_t M, int32_t D, int32_t N1);
int tavsr_ffn2_fwd(const tavsr_ffn_desc* d, tavsr_stream_t stream);
/* Backward of the block w.r.t. its activations, same streaming structure: dz = ((alpha * dy) w2) * mask / keep * act'(z)
 * [roundup128(M)][N1] (operand of w1's weight gradient; rows >= M are scratch) and dn = dz w1 [M][256] (gradient w.r.t.
 * LN(x)).  Weights in torch Linear layout, untransposed (w1 [N1][256], w2 [256][N1]); z [>= M rows][N1] as saved by the
 * forward; the inner mask is regenerated from (p_drop, seed_dev, offset_in).  dy is the block's output gradient with the
 * outer mask already applied.  ws: tavsr_ffn2_ws(M, 256, N1) floats.  Replaces the two dgrad GEMMs of espnet's autograd
 * for PositionwiseFeedForward (src/encoder/branchformer/encoder_layer.py:191-194). */
int tavsr_ffn2_bwd_dx(const float* dy, int64_t lddy, float alpha, const float* w1, const float* w2, const float* z, int32_t act,
                      int32_t M, int32_t D, int32_t N1, float p_drop, const uint64_t* seed_dev, uint64_t offset_in, float* dz,
                      float* dn, float* ws, int64_t ws_floats, tavsr_stream_t stream);
/* dn == NULL: the finishing launch that sums dn's partials is left out; they stay in ws in the layout tavsr_ffn2_slab_layout reports
 * (row m = sum over j < wpb of ws[((m / rb_rows) * wpb + j) * rb_rows + m % rb_rows][256], j ascending) for the LayerNorm backward that
 * consumes dn to sum them where it reads the row: tavsr_layernorm_bwd_partial_slab = tavsr_layernorm_bwd_partial (dx_drop NULL) /
 * tavsr_layernorm_bwd_partial_drop on those slabs - same bits, one launch less per feed-forward block
 * (src/encoder/branchformer/encoder_layer.py:191-194, 310-314: the block's LayerNorm is the first thing its input gradient meets). */
int tavsr_ffn2_slab_layout(int32_t M, int32_t N1, int32_t* wpb, int32_t* rb_rows);
int tavsr_layernorm_bwd_partial_slab(const float* slab, int32_t wpb, int32_t rb_rows, const float* x, int64_t ldx, const float* mean,
                                     const float* rstd, const float* gamma, const float* dx_add, int64_t ldadd, float* dx, int64_t lddx,
                                     float* ws, int64_t ws_ld, int32_t M, int32_t D, float* dx_drop, float p_drop,
                                     const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One Branchformer encoder layer forward as ONE call (csrc/layer.hip): MyBranchformerEncoderLayer.forward
 * (src/encoder/branchformer/encoder_layer.py:153-321) in its recipe form - macaron FFN, rel-pos attention branch beside the
 * cgMLP branch, learned-average merge + merge_proj, FFN, norm_final - sequenced in C over the entry points above (14 launches;
 * the attention branch is enqueued on `stream2` between two events, as the Python sequencing does with a side stream).
 * All buffers are the caller's: the outputs / saved tensors below are exactly what the backward pass reads; `save` = 0 (eval)
 * leaves z / h / gn / conv untouched (may be NULL).  Dropout: rate p_drop (p_att inside the attention core) with the counter
 * offsets drop_off[] in the order {ffm inner, ffm outer, attention, attention out-proj, csgu, channel_proj2, merge_proj,
 * ff inner, ff outer}; p_drop = 0 in eval.  ws: tavsr_branchformer_layer_ws(desc) floats.
 * Shapes outside the streaming kernels' range (D != 256, d_k != 64, conv kernel != 31, ...): TAVSR_EUNSUPPORTED, nothing is
 * launched, the caller keeps its own sequencing.
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_bf_layer_desc {
  int32_t B, T, D, H, ffn_units, cg_units /* 2 C */, cg_kernel, ffn_act, save;
  float p_drop, p_att, coeff;
  const float* x;                  /* [B*T][D] */
  const float* pos_emb;            /* [2T-1][D] */
  const int64_t* lens;             /* [B] */
  /* parameters (torch layouts) */
  const float *ffm_ln_w, *ffm_ln_b, *ffm_w1, *ffm_b1, *ffm_w2, *ffm_b2;
  const float *mha_ln_w, *mha_ln_b, *wq, *bq, *wk, *bk, *wv, *bv, *wpos, *pos_u, *pos_v, *wo, *bo;
  const float *mlp_ln_w, *mlp_ln_b, *cg_w1, *cg_b1, *csgu_ln_w, *csgu_ln_b, *csgu_cw, *csgu_cb, *cg_w2, *cg_b2;
  const float* merge_p[8];         /* pooling_proj{1,2}.weight, pooling_proj{1,2}.bias, weight_proj{1,2}.weight, weight_proj{1,2}.bias */
  const float *merge_w, *merge_b;
  const float *ff_ln_w, *ff_ln_b, *ff_w1, *ff_b1, *ff_w2, *ff_b2, *final_ln_w, *final_ln_b;
  const uint64_t* seed;            /* device; NULL when p_drop == p_att == 0 */
  uint64_t drop_off[9];
  /* results and what the backward pass keeps */
  float *x1, *ffm_n, *ffm_mean, *ffm_rstd, *ffm_z, *ffm_h;      /* ffm_z / ffm_h: [roundup128(B*T)][ffn_units] */
  float *n_mha, *n_mlp, *br_mean, *br_rstd;
  float *qkv, *pp, *cx, *lse, *xa;
  float *g, *g_z, *gn, *g_mean, *g_rstd, *u, *conv, *xm;
  float *score, *pooled /* the merge's row dots [4][B*T] (tavsr_merge_rows_fwd) */, *wts, *m;
  float *x2, *ff_n, *ff_mean, *ff_rstd, *ff_z, *ff_h, *x3, *y, *fin_mean, *fin_rstd;
  tavsr_stream_t stream2;          /* the attention branch's queue */
  void *ev_fork, *ev_join;         /* hipEvent_t */
  float* ws;
  int64_t ws_floats;
} tavsr_bf_layer_desc;
/* the shapes the sequencer takes (1) or not (0: the caller keeps its own sequencing of the primitive entry points) */
int tavsr_branchformer_layer_ok(int32_t B, int32_t T, int32_t D, int32_t H, int32_t ffn_units, int32_t cg_units, int32_t cg_kernel);
int64_t tavsr_branchformer_layer_ws(const tavsr_bf_layer_desc* d);
int tavsr_branchformer_layer_fwd(const tavsr_bf_layer_desc* d, tavsr_stream_t stream);

/* The backward pass of the same layer as ONE call (autograd of MyBranchformerEncoderLayer.forward,
 * src/encoder/branchformer/encoder_layer.py:153-321): `fwd` is the descriptor of a tavsr_branchformer_layer_fwd call made with
 * save = 1 (its parameters, dropout offsets and kept buffers), dy the gradient of its output y.  Writes dx and every parameter
 * gradient (torch layouts, overwritten, not accumulated); the five d_model LayerNorms' gradients come as g_ln [5][2][D] =
 * (dgamma | dbeta) of norm_final, norm_ff, norm_mlp, norm_mha, norm_ff_macaron.  Same launches, same order and the same grouping
 * (all eleven weight gradients in one grouped launch, one reduction for the five LayerNorms) as the caller-side sequencing of
 * the primitive entry points: bit-identical results.  ws >= tavsr_branchformer_layer_bwd_ws(b) floats; fwd->stream2 /
 * ev_fork / ev_join are used as in the forward call (attention-branch backward beside the cgMLP-branch backward). */
typedef struct tavsr_bf_layer_bwd_desc {
  const tavsr_bf_layer_desc* fwd;
  const float* dy;                 /* [B*T][D] */
  float* dx;                       /* [B*T][D] */
  float *g_ffm_w1, *g_ffm_b1, *g_ffm_w2, *g_ffm_b2;
  float *g_wq, *g_bq, *g_wk, *g_bk, *g_wv, *g_bv, *g_wo, *g_bo, *g_wpos, *g_pos_u, *g_pos_v;
  float *g_cg_w1, *g_cg_b1, *g_csgu_ln_w, *g_csgu_ln_b, *g_csgu_cw, *g_csgu_cb, *g_cg_w2, *g_cg_b2;
  float* g_merge_p[8];             /* in the order of fwd->merge_p */
  float *g_merge_w, *g_merge_b;
  float *g_ff_w1, *g_ff_b1, *g_ff_w2, *g_ff_b2;
  float* g_ln;                     /* [5][2][D] */
  float* ws;
  int64_t ws_floats;
  int32_t wgrad_beside;            /* 1: the weight-gradient launches at the end of the call (one grouped launch, the problems it sheds, the
                                      (dgamma, dbeta) reduction) are enqueued on fwd->stream2 behind what `stream` holds at that point and are
                                      NOT joined: the caller orders every reader of the g_* buffers, and the next user of ws / dy / the kept
                                      forward state, behind stream2.  Same results.  0 (or stream2 == stream): on `stream`, as before. */
} tavsr_bf_layer_bwd_desc;
int64_t tavsr_branchformer_layer_bwd_ws(const tavsr_bf_layer_bwd_desc* b);
int tavsr_branchformer_layer_bwd(const tavsr_bf_layer_bwd_desc* b, tavsr_stream_t stream);

/* Elementwise helpers: out = a*x + b*y (y may be NULL); strided 2-D form; dz = dh * act'(z). */
int tavsr_axpby(const float* x, const float* y, float a, float b, float* out, int64_t n, tavsr_stream_t stream);
int tavsr_axpby2d(const float* x, int64_t ldx, const float* y, int64_t ldy, float a, float b, float* out,
                  int64_t ldo, int64_t M, int32_t N, tavsr_stream_t stream);
int tavsr_act_bwd(const float* dh, const float* z, float* dz, int64_t n, int32_t act, tavsr_stream_t stream);
/* out = x * (c * s[0]), s a DEVICE scalar (an upstream loss gradient): keeps backward free of host syncs */
int tavsr_scale_dev(const float* x, const float* s, float c, float* out, int64_t n, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * cgMLP spatial gating unit (espnet2 cgmlp.py ConvolutionalSpatialGatingUnit, reached from
 * encoder_layer.py:220): out = r * (bias + depthwise_conv1d_K(gn)) along time, "same" padding,
 * per utterance; gn = LayerNorm(gate half) is a tavsr_layernorm_fwd call on the strided half.
 *   gn, out, conv: [B*T, C];  r: gate-free half, row stride ldr;  w: [C, K] (torch [C,1,K]).
 * bwd: dr (row stride lddr), dgn, dw, dbias; ws >= tavsr_dwconv_gate_bwd_ws floats.
 * ------------------------------------------------------------------------------------------- */
int tavsr_dwconv_gate_fwd(const float* gn, const float* r, int64_t ldr, const float* w, const float* bias,
                          float* out, float* conv, int32_t B, int32_t T, int32_t C, int32_t K,
                          tavsr_stream_t stream);
/* The whole ConvolutionalSpatialGatingUnit between the channel projections (espnet cgmlp.py, called at
 * src/encoder/branchformer/encoder_layer.py:213-222): out = dropout(g[:, :C] * dwconv_K(LayerNorm(g[:, C:2C]))) over B utterances
 * of T rows, g [B*T][ldg] (ldg >= 2C).  Two launches: LayerNorm statistics of the gate rows (mean / rstd [B*T], always written),
 * then normalise + convolve + gate (+ dropout: the tavsr_dropout mapping at `offset` over the [B*T][C] result).  gn / conv
 * (optional, [B*T][C]): the normalised gate and the convolution output the backward pass needs.  K = 31, C % 64 == 0; other
 * shapes: TAVSR_EUNSUPPORTED (callers keep tavsr_layernorm_fwd + tavsr_dwconv_gate_fwd). */
int tavsr_csgu_fwd(const float* g, int64_t ldg, const float* ln_w, const float* ln_b, float eps, const float* conv_w,
                   const float* conv_b, float* out, float* gn, float* conv, float* mean, float* rstd, float p_drop,
                   const uint64_t* seed_dev, uint64_t offset, int32_t B, int32_t T, int32_t C, int32_t K,
                   const float* rowstat, tavsr_stream_t stream);
/* rowstat (optional): the tavsr_gemm_desc.rowstat output of the GEMM that produced g ([B*T][ldg / 64][2]): the statistics launch is
 * skipped, the gate rows' mean / rstd come from the C / 64 gate tiles' partial sums (C <= 1024) and are still written out. */
int64_t tavsr_dwconv_gate_bwd_ws(int32_t B, int32_t T, int32_t C, int32_t K);
int tavsr_dwconv_gate_bwd(const float* du, const float* gn, const float* r, int64_t ldr, const float* conv,
                          const float* w, float* dr, int64_t lddr, float* dgn, float* dw, float* dbias,
                          int32_t accumulate, float* ws, int32_t B, int32_t T, int32_t C, int32_t K,
                          tavsr_stream_t stream);
/* ... for r = act(zr) (cgMLP: the left half of gelu(channel_proj1(x)), espnet cgmlp.py ConvolutionalGatingMLP.forward as called at
 * src/encoder/branchformer/encoder_layer.py:220): dr is written as the gradient w.r.t. zr, dr * act'(zr);
 * kernel size 31 only.  With tavsr_layernorm_bwd_act on the gate half no activation-backward pass over the projection remains. */
int tavsr_dwconv_gate_bwd_act(const float* du, const float* gn, const float* r, int64_t ldr, const float* conv, const float* w,
                              float* dr, int64_t lddr, float* dgn, float* dw, float* dbias, int32_t accumulate, float* ws,
                              int32_t B, int32_t T, int32_t C, int32_t K, const float* zr, int64_t ldz, int32_t act,
                              tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The tailored audio-visual encoder layer (csrc/layer.hip): TailoredEncoderLayer.forward
 * (src/encoder/audiovisual/tailored/encoder_layer.py:118-274).  Each modality stream runs  macaron FFN -> ONE branch (rel-pos
 * attention or cgMLP, per layer and modality: encoder.py's acoustic_use_attn / visual_use_attn lists) with its residual -> FFN ->
 * norm_final; the two FFNs and norm_ff_macaron / norm_ff / norm_final are shared modules, so both stream descriptors carry the same
 * parameter pointers there.  tavsr_tailored_stream_fwd sequences ONE stream (9-10 launches); tavsr_tailored_layer_fwd runs the video
 * stream on `stream2` beside the audio stream on the calling queue and joins them.  Fields as in tavsr_bf_layer_desc; br_ln_* is the
 * branch's LayerNorm (norm_mha / norm_cgmlp), n_br its output; drop_off: 0 macaron inner, 1 macaron outer, 2 attention
 * probabilities | cgMLP gate product, 3 branch output, 4 FFN inner, 5 FFN outer.  ws >= tavsr_tailored_stream_ws(d) floats each.
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_tailored_stream_desc {
  int32_t B, T, D, H, ffn_units, cg_units /* 2 C */, cg_kernel, ffn_act, save, use_attn;
  float p_drop, p_att, coeff;
  const float* x;                  /* [B*T][D] */
  const float* pos_emb;            /* [2T-1][D] (attention streams) */
  const int64_t* lens;             /* [B] */
  const float *ffm_ln_w, *ffm_ln_b, *ffm_w1, *ffm_b1, *ffm_w2, *ffm_b2;
  const float *br_ln_w, *br_ln_b;
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wpos, *pos_u, *pos_v, *wo, *bo;
  const float *cg_w1, *cg_b1, *csgu_ln_w, *csgu_ln_b, *csgu_cw, *csgu_cb, *cg_w2, *cg_b2;
  const float *ff_ln_w, *ff_ln_b, *ff_w1, *ff_b1, *ff_w2, *ff_b2, *final_ln_w, *final_ln_b;
  const uint64_t* seed;
  uint64_t drop_off[6];
  float *x1, *ffm_n, *ffm_mean, *ffm_rstd, *ffm_z, *ffm_h;
  float *n_br, *br_mean, *br_rstd;
  float *qkv, *pp, *cx, *lse;
  float *g, *g_z, *gn, *g_mean, *g_rstd, *u, *conv;
  float *x2, *ff_n, *ff_mean, *ff_rstd, *ff_z, *ff_h, *x3, *y, *fin_mean, *fin_rstd;
  float* ws;
  int64_t ws_floats;
} tavsr_tailored_stream_desc;
int64_t tavsr_tailored_stream_ws(const tavsr_tailored_stream_desc* d);
int tavsr_tailored_stream_fwd(const tavsr_tailored_stream_desc* d, tavsr_stream_t stream);
typedef struct tavsr_tailored_layer_desc {
  const tavsr_tailored_stream_desc *audio, *video;
  tavsr_stream_t stream2;          /* the video stream's queue */
  void *ev_fork, *ev_join;         /* hipEvent_t */
} tavsr_tailored_layer_desc;
int tavsr_tailored_layer_fwd(const tavsr_tailored_layer_desc* d, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Block-level entry points (csrc/blocks.hip): a whole module of the reference model as ONE call, sequenced in C over the
 * primitive entry points above - same launches, order and results as a caller that issues them one by one.
 *
 * tavsr_cgmlp_fwd: espnet ConvolutionalGatingMLP with the dropout / residual the layers put around it
 *   (src/encoder/branchformer/encoder_layer.py:213-226; src/encoder/audiovisual/tailored/encoder_layer.py:198-208,246-256):
 *       g = gelu(x w1^T + b1);  u = dropout_p(g[:, :C] * dwconv_31(LayerNorm(g[:, C:])));  out = res + alpha * dropout_p_out(u w2^T + b2)
 *   x: the block's (already normalised) input rows [B*T][D]; res may be NULL; units = 2C (C <= 1024, 2C % 128 == 0), kernel 31.
 *   Kept for the backward pass (save = 1): g, g_z (pre-activations), gn, conv, g_mean / g_rstd, u.
 * tavsr_cgmlp_bwd: dy = gradient w.r.t. out (the residual path's share is the caller's); writes dx (w.r.t. x) and the eight
 *   parameter gradients (overwritten).
 * ------------------------------------------------------------------------------------------- */
typedef struct tavsr_cgmlp_desc {
  int32_t B, T, D, units, kernel, save;
  float p_drop, p_out, alpha;
  const uint64_t* seed;            /* device; NULL without dropout */
  uint64_t off_u, off_out;         /* dropout offsets of u [B*T][C] and of the output [B*T][D] (tavsr_dropout mapping) */
  const float *x, *res;
  const float *w1, *b1, *ln_w, *ln_b, *cw /* [C][31] */, *cb, *w2, *b2;
  float *g /* [B*T][2C] */, *g_z, *gn, *g_mean, *g_rstd, *u, *conv, *out;
  float* ws;
  int64_t ws_floats;
} tavsr_cgmlp_desc;
int64_t tavsr_cgmlp_ws(const tavsr_cgmlp_desc* d);
int tavsr_cgmlp_fwd(const tavsr_cgmlp_desc* d, tavsr_stream_t stream);
typedef struct tavsr_cgmlp_bwd_desc {
  const tavsr_cgmlp_desc* fwd;
  const float* dy;
  float* dx;
  float *g_w1, *g_b1, *g_ln_w, *g_ln_b, *g_cw, *g_cb, *g_w2, *g_b2;
  float* ws;
  int64_t ws_floats;
} tavsr_cgmlp_bwd_desc;
int64_t tavsr_cgmlp_bwd_ws(const tavsr_cgmlp_bwd_desc* b);
int tavsr_cgmlp_bwd(const tavsr_cgmlp_bwd_desc* b, tavsr_stream_t stream);

/* tavsr_conv2d_subsample_fwd: espnet Conv2dSubsampling / Conv2dSubsamplingWOPosEnc (src/encoder/branchformer/encoder.py:364,
 *   src/embedding_for_avsr/default.py:111-162): out = xscale * (Linear(flatten_{c,f}(relu(conv2(relu(conv1(x)))))) ), both
 *   convolutions 3x3 / stride 2 / no padding, x [B][T][F] (one input channel), parameters in torch layouts (w1 [C][1][3][3],
 *   w2 [C][C][3][3], wo [odim][C * F2]).  Channels-last inside: y1 [B][T1][F1][C], y2 [B*T2*F2][C] (kept for the backward pass
 *   together with the re-indexed weights w2r [C][9C] and wor [odim][F2*C], which the call writes).  The second convolution runs as
 *   an implicit GEMM (C % 64 == 0, B*T2*F2 % 32 == 0, else TAVSR_EUNSUPPORTED).  zero_page: 64 zero floats (device).
 * tavsr_conv2d_subsample_bwd: dout [B*T2][odim] -> the six parameter gradients in torch layouts (the input gets none). */
typedef struct tavsr_subsample_desc {
  int32_t B, T, F, C, odim;
  float xscale;
  const float* x;
  const float *w1, *b1, *w2, *b2, *wo, *bo;
  const float* zero_page;
  float *y1, *y2, *w2r, *wor, *out;
  float* ws;
  int64_t ws_floats;
} tavsr_subsample_desc;
int64_t tavsr_conv2d_subsample_ws(const tavsr_subsample_desc* d);
int tavsr_conv2d_subsample_fwd(const tavsr_subsample_desc* d, tavsr_stream_t stream);
typedef struct tavsr_subsample_bwd_desc {
  const tavsr_subsample_desc* fwd;
  const float* dout;
  float *g_w1, *g_b1, *g_w2, *g_b2, *g_wo, *g_bo;
  float* ws;
  int64_t ws_floats;
  int32_t wgrad_beside;            /* 1 (with stream2 and ev_fork): the weight gradients of the second convolution and of the output Linear, and their
                                      re-layouts, are enqueued on stream2 behind the launch that produces dz2 and are NOT joined (as
                                      tavsr_bf_layer_bwd_desc.wgrad_beside): the caller orders the readers of g_w2 / g_b2 / g_wo / g_bo and the next
                                      user of ws / dout behind stream2.  Same results. */
  tavsr_stream_t stream2;
  void* ev_fork;                   /* a hipEvent_t of the caller's */
} tavsr_subsample_bwd_desc;
int64_t tavsr_conv2d_subsample_bwd_ws(const tavsr_subsample_bwd_desc* b);
int tavsr_conv2d_subsample_bwd(const tavsr_subsample_bwd_desc* b, tavsr_stream_t stream);

/* Workspace of any descriptor-driven entry point in BYTES (the per-entry *_ws queries count floats): kind names the descriptor
 * type behind `desc`.  -1: unknown kind; 0: nothing needed, or a descriptor the entry point would refuse. */
enum { TAVSR_WS_GEMM = 0, TAVSR_WS_FFN2 = 1, TAVSR_WS_BF_LAYER_FWD = 2, TAVSR_WS_BF_LAYER_BWD = 3, TAVSR_WS_CGMLP_FWD = 4,
       TAVSR_WS_CGMLP_BWD = 5, TAVSR_WS_SUBSAMPLE_FWD = 6, TAVSR_WS_SUBSAMPLE_BWD = 7 };
int64_t tavsr_workspace_bytes(int32_t kind, const void* desc);

/* ---------------------------------------------------------------------------------------------
 * learned_ave branch merge (encoder_layer.py:232-293) and the same pooling used by
 * AdaptiveAudioVisualFusion (adaptive_audiovisual_fusion.py:137-191).
 *   params (HOST array of 8 device pointers): pooling_proj{1,2}.weight[D], pooling_proj{1,2}.bias[1],
 *   weight_proj{1,2}.weight[D], weight_proj{1,2}.bias[1].
 *   pool_fwd: score [2,B,T], pooled [2,B,D], w [B,2] = (weight_global, weight_local)
 *   combine : out = w[b,0]*x1 + w[b,1]*x2
 *   lens2   : optional valid lengths of the second stream (NULL: lens serves both) - the AV fusion pools the
 *             audio and the video stream under their own masks (adaptive_audiovisual_fusion.py:146-179)
 *   bwd     : dx1, dx2 and the 8 parameter gradients (dparams: HOST array of device pointers,
 *             order weight{pool1,pool2,w1,w2} then bias{pool1,pool2,w1,w2});
 *             ws >= tavsr_merge_bwd_ws(B, D) floats.
 * ------------------------------------------------------------------------------------------- */
int tavsr_merge_pool_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                         const float* const* params,
                         float* score, float* pooled, float* w, int32_t B, int32_t T, int32_t D,
                         tavsr_stream_t stream);
int tavsr_merge_combine(const float* x1, const float* x2, const float* w, float* out, int32_t B, int32_t T,
                        int32_t D, tavsr_stream_t stream);
/* The row-parallel form of the same merge (D = 256, T <= 2048: tavsr_merge_rows_ok): the launches above run one workgroup
 * per utterance (B of the 256 CUs); these split every utterance into 16-row blocks.  Pass 1 writes four dot products per row
 * (dots [4][B*T] = <wp_1,x_1>, <wp_2,x_2>, <ww_1,x_1>, <ww_2,x_2>; weight_k = sum_t score_k[t] <ww_k, x_k[t]> + bw_k, so
 * no pooled vector is needed), pass 2 redoes the utterance's softmaxes from them in every block and combines the block's
 * rows.  The backward pass reads `dots` back (it replaces `pooled` as the saved tensor), writes dx1 / dx2 - optionally
 * already under the dropout masks of the two branch outputs (encoder_layer.py:212,224) - and the eight parameter
 * gradients; ws >= tavsr_merge_rows_bwd_ws(B, T, D) floats. */
int tavsr_merge_rows_ok(int32_t T, int32_t D);
int tavsr_merge_rows_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2, const float* const* params,
                         float* dots, float* score, float* w, float* out, int32_t B, int32_t T, int32_t D, tavsr_stream_t stream);
int64_t tavsr_merge_rows_bwd_ws(int32_t B, int32_t T, int32_t D);
int tavsr_merge_rows_bwd(const float* dm, const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                         const float* const* params, const float* score, const float* w, const float* dots, float* dx1,
                         float* dx2, float* const* dparams, int32_t accumulate, float* ws, float p_drop1, uint64_t offset1,
                         float p_drop2, uint64_t offset2, const uint64_t* seed_dev, int32_t B, int32_t T, int32_t D,
                         tavsr_stream_t stream);
/* The whole tail of a Branchformer layer behind the branch join in ONE launch (csrc/mergeproj.hip, D = 256, T <= 2048:
 * tavsr_merge_proj_ok): the learned_ave merge above AND  out = res + alpha * dropout(mix W^T + bias)  with mix the merged
 * rows (encoder_layer.py:232-300: pooling, weight softmax, weighted sum, merge_proj, dropout, residual).  One workgroup per 16
 * frames of an utterance: it recomputes the utterance's row dots itself (no dots launch to wait for), forms its mixed rows and
 * multiplies them by W [256][256] (torch layout) on v_mfma_f32_16x16x4_f32.  dots [4][B*T], score [2][B][T] and w [B][2] are what
 * tavsr_merge_rows_bwd reads back; mix [B*T][256] (may be NULL: passes without a backward) is merge_proj's saved input.  The
 * dropout mask is the one tavsr_gemm draws for a [B*T][256] result at drop_offset (drop_offset % 4 == 0), so the
 * backward pass regenerates it from the same token. */
int tavsr_merge_proj_ok(int32_t T, int32_t D);
int tavsr_merge_proj_fwd(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2, const float* const* params,
                         const float* w, const float* bias, const float* res, float alpha, float p_drop, const uint64_t* seed_dev,
                         uint64_t drop_offset, float* dots, float* score, float* wout, float* mix, float* out, int32_t B, int32_t T,
                         int32_t D, tavsr_stream_t stream);
/* ... with the row dots handed in: rowdots_k [B*T][4][2] = per row and 64-column tile of branch output x_k the two sums (<pooling_proj_k,
 * x_k>, <weight_proj_k, x_k>) that the GEMM producing x_k left (tavsr_gemm_desc.rowstat with rowdot_a = pooling_proj_k.weight,
 * rowdot_b = weight_proj_k.weight): the launch then reads 64 bytes per row and branch instead of both branch outputs of the whole
 * utterance per workgroup.  Both NULL: tavsr_merge_proj_fwd. */
int tavsr_merge_proj_fwd_dots(const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2, const float* const* params,
                              const float* w, const float* bias, const float* res, float alpha, float p_drop, const uint64_t* seed_dev,
                              uint64_t drop_offset, const float* rowdots1, const float* rowdots2, float* dots, float* score, float* wout,
                              float* mix, float* out, int32_t B, int32_t T, int32_t D, tavsr_stream_t stream);
int64_t tavsr_merge_bwd_ws(int32_t B, int32_t D);
int tavsr_merge_bwd(const float* dm, const float* x1, const float* x2, const int64_t* lens, const int64_t* lens2,
                    const float* const* params, const float* score, const float* pooled, const float* w,
                    float* dx1, float* dx2, float* const* dparams, int32_t accumulate, float* ws, int32_t B,
                    int32_t T, int32_t D, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Conv2dSubsampling (espnet subsampling.py; encoder.py:149-155,364) in channels-last form, and
 * UtteranceMVN (espnet_model.py:388).
 *   conv1_fwd : y[B,To,Fo,C] = relu(conv3x3 stride 2 (x[B,T,F], w[C,1,3,3]) + bias)
 *   conv1_bwd : dw[C,9], db[C] from dz (relu' already applied); ws >= tavsr_conv1_bwd_ws floats
 *   im2col    : col[(b,to,fo)][(kh*3+kw)*C + c] = y[b,2to+kh,2fo+kw,c]   -> conv2 is a tavsr_gemm
 *   col2im    : dz = relu'(y) * scatter-free gather of dcol (gradient w.r.t. conv1's output)
 *   transpose_inner: out[n][c][r] (+)= in[n][r][c]  (weight re-indexing torch order <-> NHWC order)
 *   utterance_mvn  : per-utterance mean removal over valid frames, padded frames -> 0
 * ------------------------------------------------------------------------------------------- */
int tavsr_conv1_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t T,
                    int32_t F, int32_t C, tavsr_stream_t stream);
int64_t tavsr_conv1_bwd_ws(int32_t B, int32_t T, int32_t F, int32_t C);
int tavsr_conv1_bwd(const float* dz, const float* x, float* dw, float* db, int32_t accumulate, float* ws,
                    int32_t B, int32_t T, int32_t F, int32_t C, tavsr_stream_t stream);
int tavsr_im2col3x3s2(const float* y, float* col, int32_t B, int32_t Ti, int32_t Fi, int32_t C,
                      tavsr_stream_t stream);
int tavsr_col2im3x3s2_relu(const float* dcol, const float* yrelu, float* dz, int32_t B, int32_t Ti,
                           int32_t Fi, int32_t C, tavsr_stream_t stream);
int tavsr_transpose_inner(const float* in, float* out, int64_t nb, int32_t R, int32_t Cc, int32_t accumulate,
                          tavsr_stream_t stream);
int tavsr_utterance_mvn(const float* x, const int64_t* lens, float* y, int32_t B, int32_t T, int32_t F,
                        tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * CTC (src/ctc/ctc.py).  logits[b][t][v] at b*ld_b + t*ld_t + v.
 *   ctc_loss  : per-utterance negative log-likelihood of torch.nn.CTCLoss(reduction="none",
 *               zero_infinity) on log_softmax(logits) (ctc.py:58-69) AND d loss[b] / d logits
 *               (the caller scales by its upstream gradient).  targets int64 [B, ld_tgt] padded,
 *               ws >= tavsr_ctc_loss_ws(B, T, Lmax) floats.
 *   ctc_greedy: ids = argmax_v (ctc.py:180-188, lowest index on ties); hyp/hyp_len (optional) =
 *               collapse-repeats-then-drop-blank of ids[:hlens[b]] (maskctc_model.py:289-291),
 *               hyp padded with -1.  Integer outputs: bit-exact contract.
 * ------------------------------------------------------------------------------------------- */
int64_t tavsr_ctc_loss_ws(int32_t B, int32_t T, int32_t Lmax);
int tavsr_ctc_loss(const float* logits, int64_t ld_t, int64_t ld_b, const int64_t* hlens,
                   const int64_t* targets, int64_t ld_tgt, const int64_t* tlens, int32_t blank,
                   int32_t zero_infinity, float* loss, float* grad, float* ws, int32_t B, int32_t T,
                   int32_t V, int32_t Lmax, tavsr_stream_t stream);
int tavsr_ctc_greedy(const float* logits, int64_t ld_t, int64_t ld_b, const int64_t* hlens, int32_t blank,
                     int64_t* ids, int64_t* hyp, int64_t* hyp_len, int32_t B, int32_t T, int32_t V,
                     tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention-decoder loss side (espnet LabelSmoothingLoss + th_accuracy, espnet_model.py:553-569)
 * and decoder input embedding (Embedding + PositionalEncoding, espnet transformer_decoder.py).
 *   lsm_loss: row_loss[r] = KL(smoothed one-hot || softmax(logits[r])) (0 for ignored rows),
 *             grad = d row_loss / d logits, correct[r] in {1, 0, -1 = ignored}.
 * ------------------------------------------------------------------------------------------- */
int tavsr_lsm_loss(const float* logits, int64_t ld, const int64_t* target, int32_t ignore, float smoothing,
                   float* row_loss, float* grad, int32_t* correct, int64_t rows, int32_t V,
                   tavsr_stream_t stream);
int tavsr_embed_pe(const int64_t* ids, const float* table, const float* pe, float scale, float* out, int64_t N,
                   int32_t L, int32_t D, tavsr_stream_t stream);
/* the one-token form of a replayed search step (espnet TransformerDecoder.forward_one_step embeds ONE position): every row gets
 * positional row min(*step_dev, L - 1) of pe [L][D]; the step counter is device data, so the launch is captured once */
int tavsr_embed_pe_step(const int64_t* ids, const float* table, const float* pe, float scale, float* out, int64_t N,
                        int32_t L, int32_t D, const int32_t* step_dev, tavsr_stream_t stream);
int tavsr_embed_bwd(const int64_t* ids, const float* dout, float scale, float* dtable, int64_t N, int32_t V,
                    int32_t D, int32_t accumulate, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Visual frontend (src/frontend/conv3d_resnet18/conv3d_resnet18.py:42-97, modules/resnet.py:25-178), channels-last:
 * every activation is a [N*H*W, C] matrix (N = B*T frames), each convolution = im2col + tavsr_gemm.
 *   im2col2d / col2im2d : KHxKW patches, stride, zero padding (conv3x3 of BasicBlock resnet.py:8-22, 1x1 shortcut
 *                         conv :25-41); col rows (n,ho,wo), columns (kh*KW+kw)*C + c; col2im is the gather-form adjoint
 *   im2col_stem         : Conv3d(1->64, k (5,7,7), s (1,2,2), p (2,3,3)) patches of x[B,T,H,W], 245 taps padded to 256
 *   bn_stats            : training-mode BatchNorm statistics over the M rows (two-pass, deterministic): mean, biased
 *                         var, rstd; running_mean/var/num_batches_tracked updated like torch.nn.BatchNorm (may be NULL)
 *   bn_apply_fwd        : y = act((x - mean) * rstd * gamma + beta (+ res)), act in {none, swish}
 *   bn_bwd              : dz = dy * act'(z) (z recomputed; dz is also d/d res), dgamma, dbeta, dx; ws >= tavsr_bn_ws floats
 *   rsqrt_eps           : out = 1 / sqrt(v + eps)  (eval-mode BatchNorm: rstd from running_var)
 *   maxpool3x3s2        : per-frame 3x3 / stride 2 / pad 1 max pooling (the stem's MaxPool3d (1,3,3)); idx = winning tap
 *   avgpool             : mean over the P pixels of a frame (AdaptiveAvgPool2d(1), resnet.py:176)
 *   fill                : p[i] = value (alignment padding rows, avsr_espnet_model.py:531-538)
 * ------------------------------------------------------------------------------------------- */
int tavsr_im2col2d(const float* x, float* col, int64_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                   int32_t stride, int32_t pad, tavsr_stream_t stream);
/* extra (nullable, [N][Ho][Wo][C]): gradient rows of a parallel 1x1 / same stride / pad 0 convolution of the same input (the
 * downsample path, resnet.py:68-87), added at pixels (stride*ho, stride*wo): dx = col2im(dcol) + scatter(extra) in one pass */
int tavsr_col2im2d(const float* dcol, float* dx, int64_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                   int32_t stride, int32_t pad, const float* extra, tavsr_stream_t stream);
int tavsr_im2col_stem(const float* x, float* col, int32_t B, int32_t T, int32_t H, int32_t W, tavsr_stream_t stream);
int64_t tavsr_bn_ws(int64_t M, int32_t C);
int tavsr_bn_stats(const float* x, int64_t M, int32_t C, float eps, float momentum, float* mean, float* var, float* rstd,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ws, tavsr_stream_t stream);
int tavsr_bn_apply_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* res, float* y, int64_t M, int32_t C, int32_t act, tavsr_stream_t stream);
/* dz may be NULL when res is NULL (nobody reads the pre-activation gradient): one [M, C] write less. */
int tavsr_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 const float* res, float* dz, float* dx, float* dgamma, float* dbeta, int64_t M, int32_t C, int32_t act,
                 float* ws, tavsr_stream_t stream);
/* tavsr_bn_bwd for the stem (conv -> BatchNorm -> activation -> MaxPool(1,3,3)/(1,2,2), conv3d_resnet18.py:57-59): the incoming
 * gradient is the POOLED one (dpool [N,Ho,Wo,C] + the winning taps idx of tavsr_maxpool3x3s2_fwd); the gradient of the pool's
 * [N*H*W, C] input is rebuilt per 2x2 pixel block inside both passes and never written.  ws >= tavsr_bn_ws(N*H*W, C). */
int tavsr_bn_bwd_pooled(const float* dpool, const uint8_t* idx, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, int64_t N, int32_t H,
                        int32_t W, int32_t C, int32_t act, float* ws, tavsr_stream_t stream);
int tavsr_rsqrt_eps(const float* v, float eps, float* out, int64_t n, tavsr_stream_t stream);
int tavsr_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int32_t H, int32_t W, int32_t C,
                           tavsr_stream_t stream);
/* y = maxpool3x3/s2/p1(act(BatchNorm(x))) per frame with the statistics given (mean, rstd): tavsr_bn_apply_fwd fused into the
 * pool's window loads - the stem's [frames*H*W, C] activation map (conv3d_resnet18.py:57-63) is never written */
int tavsr_bn_act_maxpool3x3s2_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  int32_t act, float* y, uint8_t* idx, int64_t N, int32_t H, int32_t W, int32_t C,
                                  tavsr_stream_t stream);
int tavsr_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int32_t H, int32_t W, int32_t C,
                           tavsr_stream_t stream);
int tavsr_avgpool_fwd(const float* x, float* y, int64_t N, int32_t P, int32_t C, tavsr_stream_t stream);
int tavsr_avgpool_bwd(const float* dy, float* dx, int64_t N, int32_t P, int32_t C, tavsr_stream_t stream);
int tavsr_fill(float* p, float value, int64_t n, tavsr_stream_t stream);
/* dst[m*ldd + n] = src[m*lds + n], m < M, n < N (no alignment requirement) */
/* data-gradient weights of a 3x3 convolution for the implicit GEMM (tavsr_gemm_desc.conv_mode 1 on dY):
 * wflip[ci][tap*Cout + co] = w2d[co][(8 - tap)*Cin + ci],  w2d = [Cout][9*Cin] */
int tavsr_conv_wflip(const float* w2d, float* wflip, int32_t Cout, int32_t Cin, tavsr_stream_t stream);
int tavsr_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int64_t N, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Log-mel frontend and SpecAug (espnet2 DefaultFrontend / SpecAug as configured by
 * configs/ASR/branchformer_transformer+ctc_english.yaml:9-37, called at src/models/espnet_model.py:378-388).
 * The DFT and the mel projection are tavsr_gemm calls; these are the steps around them.
 *   tavsr_stft_frames : frames[b*T+t][n] = window[n] * reflect_pad(wav[b])[t*hop + n]   (window: hann(win_length)
 *                       zero-padded to n_fft, centred, as torch.stft does; centre padding n_fft/2 when center != 0)
 *   tavsr_power_spec  : P[row][k] = re_k^2 + im_k^2 from spec rows [re_0..re_{nfreq-1} | im_0..im_{nfreq-1}];
 *                       0 in the padding columns k >= nfreq and in frames t >= olens[b]
 *   tavsr_log_mask    : out = log(max(mel, floor)), 0 in frames t >= olens[b]
 *   tavsr_time_warp   : espnet2 TimeWarp with torch's bicubic (A = -0.75, align_corners = False) on the first lens[b]
 *                       frames of utterance b: [0, center[b]) resized to [0, warped[b]), [center[b], lens[b]) to
 *                       [warped[b], lens[b]); center[b] == 0 copies; frames >= lens[b] become 0; device int64 [B]; y != x
 *   tavsr_specaug_mask: espnet2 mask_along_axis on both axes, in place: zero where f is in one of utterance b's nf bands
 *                       [fpos, fpos+flen) or t in one of its nt bands (arrays [B][nf] / [B][nt], device int64)
 * ------------------------------------------------------------------------------------------- */
int tavsr_stft_frames(const float* wav, const float* window, float* frames, int32_t B, int64_t N, int32_t T, int32_t n_fft,
                      int32_t hop, int32_t center, tavsr_stream_t stream);
int tavsr_power_spec(const float* spec, int64_t ld_spec, float* P, int32_t ldp, int32_t nfreq, int32_t B, int32_t T,
                     const int64_t* olens, tavsr_stream_t stream);
int tavsr_log_mask(const float* mel, float* out, int32_t B, int32_t T, int32_t n_mels, const int64_t* olens, float floor_,
                   tavsr_stream_t stream);
int tavsr_time_warp(const float* x, float* y, int32_t B, int32_t T, int32_t F, const int64_t* center, const int64_t* warped,
                    const int64_t* lens, tavsr_stream_t stream);
int tavsr_specaug_mask(float* x, int32_t B, int32_t T, int32_t F, const int64_t* fpos, const int64_t* flen, int32_t nf,
                       const int64_t* tpos, const int64_t* tlen, int32_t nt, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Decode steps of the hybrid CTC/attention beam search with LM scoring (src/inference/avsr_inference.py:141-304,
 * 449-518: espnet BatchBeamSearch + TransformerDecoder.batch_score + TransformerLM.batch_score + CTCPrefixScorer).
 *   tavsr_tree_attn_step : out[n][h] = softmax(scale * q[n][h] . K[anc[n][j]][h], j < nkeys) . V[anc[n][j]][h].
 *       Keys/values of all hypotheses live in a pool (row = one token of one hypothesis, H*dk floats, row stride ldkv);
 *       anc [N][ld_anc] int32 lists each hypothesis' own rows, so a beam re-order copies these lists only.
 *       step_dev (nullable): device int32 holding the step index; the kernel then uses min(*step_dev + 1, nkeys) keys,
 *       so one captured hipGraph of the scorer step can be replayed for every step of the search.
 *       k_new / v_new (nullable, row stride ldq): this step's keys / values [N][H*dk].  They are key number nkeys-1 of every
 *       hypothesis (pool row (nkeys-1)*N + n, which anc must name): read from here and written to the pool by the same launch.
 *       group (0 / 1: none): hypotheses n in [u*group, (u+1)*group) belong to one utterance (the beam): they are scheduled
 *       next to each other per head, so that the rows their ancestor lists share are fetched once per CU.
 *   tavsr_kv_append      : kpool/vpool row (*step_dev * N + n) = k[n] / v[n] (the rows this step's anc column names).
 *   tavsr_ctc_prefix_step: espnet CTCPrefixScoreTH.__call__ (no attention window) for C candidate tokens per
 *       hypothesis.  logp [U][T][V] log-softmax of the CTC head, lens [U] frames, hypotheses n belong to utterance n / K.
 *       r_prev [N][T][2] / s_prev [N] / last_tok [N]: forward variables (non-blank, blank), log_psi and last token of
 *       each hypothesis (ignored when out_len == 0: the <sos> state is used).  Writes r_new [N][T][2][C],
 *       psi [N][C] = log_psi(cand) - s_prev, psi_abs [N][C] = log_psi(cand), eos [N] = log p(prefix ends) - s_prev,
 *       eos_abs [N].  A candidate equal to <eos> must take the eos value (caller); blank scores -1e10.
 *       step_dev (nullable): device int32 that replaces out_len (graph replays, as in tavsr_tree_attn_step).
 *   tavsr_log_softmax_rows: y[m][:] = (accumulate ? y[m][:] : 0) + alpha * log_softmax(x[m][:V]) + add  (the weighted sum of
 *       the scorers: decoder 1 - ctc_weight, lm lm_weight, length bonus as `add`).
 *   tavsr_rowlin         : out[n][c] = res[n][c] + act(LN(x[g(n)])[:K] . W[c][:K] + bias[c]), n < N, c < Nout - one Linear of
 *       forward_one_step / batch_score (espnet transformer decoder_layer.py / encoder_layer.py with cache) with the
 *       LayerNorm in front of it (gamma/beta nullable: none), bias, activation (TAVSR_ACT_*) and residual (nullable; may
 *       alias out) in ONE launch.  gather (nullable): g(n) = gather[n] (embedding rows), else g(n) = n; res_gather (nullable): the
 *       residual row of n is res[res_gather[n]] (res must not alias out then).  K % 32 == 0, 16-byte aligned rows; out must not alias x.
 * ------------------------------------------------------------------------------------------- */
int tavsr_rowlin(const float* x, int64_t ldx, const int64_t* gather, const float* gamma, const float* beta, float eps,
                 const float* W, int64_t ldw, const float* bias, int32_t act, const float* res, int64_t ldr,
                 const int64_t* res_gather, float* out, int64_t ldo, int32_t N, int32_t K, int32_t Nout, tavsr_stream_t stream);
/* The same launch with rows held as a SUM of tensors: x = sum_{p < x_parts} x[p * x_pstride + ...] (added while the operand is
 * loaded, in that order; the LayerNorm sees the sum), res likewise; ksplit > 1: K is dealt to ksplit blocks per column tile and block y
 * writes partial tensor y of the output at out + y * out_pstride (bias and residual ride in partial 0; no LayerNorm, no activation) -
 * the next launch of the chain reads the partials as its x_parts / res_parts.  The 2048 -> d projection that closes a feed-forward
 * block is otherwise 32 blocks of 16 waves, each alone with 128 KB of weights and all of x (8.3 us against 4.9 for its K = 512
 * neighbours); no reduction launch, no atomics, no fences: the kernel boundary publishes the partials. */
int tavsr_rowlin_parts(const float* x, int64_t ldx, int32_t x_parts, int64_t x_pstride, const float* gamma, const float* beta,
                       float eps, const float* W, int64_t ldw, const float* bias, int32_t act, const float* res, int64_t ldr,
                       int32_t res_parts, int64_t res_pstride, float* out, int64_t ldo, int32_t ksplit, int64_t out_pstride,
                       int32_t N, int32_t K, int32_t Nout, tavsr_stream_t stream);
/* tuning / test aid (scripts/tree_attn_bench.py, tests/test_gpu_rowlin.py): how a small step (N * H <= 256 items) is launched -
 * 3: the plan (0 for histories beyond ~80 keys - a captured step: half its pool capacity -, else 1); 0: a workgroup of four waves per
 * item, the keys dealt to the waves; 1: one wave per item and workgroup; 2: one wave per item, four items per workgroup (what larger
 * steps always run) */
int tavsr_tree_attn_tune(int32_t mode);
/* 1 when tavsr_rowlin (ksplit 1) / tavsr_rowlin_parts has a one-launch plan for N rows, this K, with / without the LayerNorm
 * prologue, in `ksplit` K slices: N <= 32; K in {64 .. 512}; 1024 (with LayerNorm: up to 16 rows); 2048 without LayerNorm and up to 16
 * rows (more rows: in slices); slices of 256 / 512 / 1024 without LayerNorm.  Every plan runs without register spills. */
int tavsr_rowlin_ok(int32_t N, int32_t K, int32_t with_ln, int32_t ksplit);
int tavsr_tree_attn_step(const float* q, int64_t ldq, const float* kpool, const float* vpool, int64_t ldkv,
                         const int32_t* anc, int64_t ld_anc, int32_t nkeys, float* out, int64_t ldo, int32_t N, int32_t H,
                         int32_t dk, float scale, const int32_t* step_dev, const float* k_new, const float* v_new,
                         int32_t group, tavsr_stream_t stream);
int tavsr_kv_append(const float* k, const float* v, int64_t ld_src, float* kpool, float* vpool, int64_t ldkv, int32_t N,
                    int32_t D, int32_t max_steps, const int32_t* step_dev, tavsr_stream_t stream);
int tavsr_ctc_prefix_step(const float* logp, const int64_t* lens, const float* r_prev, const float* s_prev,
                          const int64_t* last_tok, const int64_t* cand, float* r_new, float* psi, float* psi_abs, float* eos,
                          float* eos_abs, int32_t N, int32_t K, int32_t T, int32_t V, int32_t C, int32_t out_len,
                          int32_t blank, const int32_t* step_dev, tavsr_stream_t stream);
/* as tavsr_ctc_prefix_step with the pre-beam in front: cand [N][C] is an OUTPUT - the C <= 64 best tokens of full[n][0..V)
 * (descending, lower index first among equal scores; espnet's pre_beam on the weighted full scores), V <= 4096. */
int tavsr_ctc_prefix_step_topk(const float* logp, const int64_t* lens, const float* r_prev, const float* s_prev,
                               const int64_t* last_tok, const float* full, int64_t* cand, float* r_new, float* psi, float* psi_abs,
                               float* eos, float* eos_abs, int32_t N, int32_t K, int32_t T, int32_t V, int32_t C, int32_t out_len,
                               int32_t blank, const int32_t* step_dev, tavsr_stream_t stream);
int tavsr_log_softmax_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t M, int32_t V, float alpha, float add,
                           int32_t accumulate, tavsr_stream_t stream);
/* Beam update around the top-k (espnet BatchBeamSearch.search / batch_beam, avsr_inference.py:449-518):
 *   tavsr_beam_combine: weighted [N][V] = full + w_ctc * ctc_full + score[n], where the partial CTC scorer's row ctc_full is
 *       -1e10 - s_prev[n] everywhere except [eos] = eos_s[n] and the C pre-beam candidates [cand[n][c]] = psi[n][c]; an <eos>
 *       candidate takes eos_s[n], and its psi_abs entry is overwritten with eos_abs[n].
 *   tavsr_beam_reorder: top_i [U][K] (= slot * V + token, from the top-k over weighted viewed as [U][K*V]) and top_s select,
 *       for every new hypothesis n, the slot `prev` it extends: *_out[n] = state[prev] for the CTC forward variables
 *       (r_new [N][T][2][C] at the token's candidate column -> r_out [N][T][2]), log_psi (s_out), the token history
 *       (yseq_out, row stride ld_y, new token written at column *step_dev + 1), the ancestor lists (anc_out, stride ld_a),
 *       tok_out and score_out.  Outputs must not alias the inputs. */
int tavsr_beam_combine(const float* full, const int64_t* cand, const float* psi, float* psi_abs, const float* eos_s,
                       const float* eos_abs, const float* s_prev, const float* score, float* weighted, int32_t N, int32_t V,
                       int32_t C, int32_t eos, float w_ctc, tavsr_stream_t stream);
/* tavsr_beam_combine and the top-k over (beam slot, token) of every utterance in ONE launch (one workgroup per utterance): top_s /
 * top_i [N / K][K], descending, top_i = slot * V + token (int64, what tavsr_beam_reorder takes); among equal scores the lower index
 * first.  weighted (nullable): also stores the [N][V] weighted scores.  K * V <= 8192, else TAVSR_EUNSUPPORTED. */
int tavsr_beam_combine_topk(const float* full, const int64_t* cand, const float* psi, float* psi_abs, const float* eos_s,
                            const float* eos_abs, const float* s_prev, const float* score, float* weighted, float* top_s,
                            int64_t* top_i, int32_t N, int32_t K, int32_t V, int32_t C, int32_t eos, float w_ctc,
                            tavsr_stream_t stream);
/* The beam update behind the scorers in one launch, for vocabularies of up to 64 tokens and beams of up to 16 (espnet
 * BatchBeamSearch.search with the partial CTC scorer, avsr_inference.py:277-304): psi_all / psi_abs_all [N][V] are the CTC prefix scores
 * of EVERY token (tavsr_ctc_prefix_step with the identity candidate list - it can run beside the scorers, off the step's critical
 * path).  full = dec + w_lm * log_softmax(z_lm) + add (z_lm nullable: full = dec); pre-beam = the C best tokens of full; weighted as
 * tavsr_beam_combine with those candidates; top_s / top_i as tavsr_beam_combine_topk.  psi_abs_all[n][eos] becomes eos_abs[n].
 * full_out / weighted [N][V], cand_out [N][C] (int64, the pre-beam in descending order): nullable records of the intermediate values.
 * Bit for bit what tavsr_log_softmax_rows (accumulate) + tavsr_ctc_prefix_step_topk + tavsr_beam_combine_topk give. */
int tavsr_beam_select_topk(const float* dec, const float* z_lm, float w_lm, float add, const float* psi_all, float* psi_abs_all,
                           const float* eos_s, const float* eos_abs, const float* s_prev, const float* score, float* full_out,
                           float* weighted, int64_t* cand_out, float* top_s, int64_t* top_i, int32_t N, int32_t K, int32_t V,
                           int32_t C, int32_t eos, float w_ctc, tavsr_stream_t stream);
int tavsr_beam_reorder(const int64_t* top_i, const float* top_s, const int64_t* cand, const float* r_new, const float* psi_abs,
                       const int64_t* yseq, const int32_t* anc, float* r_out, float* s_out, int64_t* yseq_out, int32_t* anc_out,
                       int64_t* tok_out, float* score_out, int32_t N, int32_t K, int32_t V, int32_t C, int32_t T, int32_t ld_y,
                       int32_t ld_a, const int32_t* step_dev, int32_t* hist, int32_t hist_steps, tavsr_stream_t stream);
/* ... with the head of the NEXT step in the same launch (maxlen [N / K] int32 given): score_out = -inf for a hypothesis that took
 * <eos> or whose utterance has used its token budget (step + 1 >= maxlen[u]), anc_out[n][step + 1] = n + (step + 1) * N - what
 * tavsr_beam_step_begin would do at the top of step + 1 (the records in `hist` keep the unmasked score). */
int tavsr_beam_reorder_begin(const int64_t* top_i, const float* top_s, const int64_t* cand, const float* r_new, const float* psi_abs,
                             const int64_t* yseq, const int32_t* anc, float* r_out, float* s_out, int64_t* yseq_out, int32_t* anc_out,
                             int64_t* tok_out, float* score_out, int32_t N, int32_t K, int32_t V, int32_t C, int32_t T, int32_t ld_y,
                             int32_t ld_a, const int32_t* step_dev, int32_t* hist, int32_t hist_steps, const int32_t* maxlen,
                             int32_t eos, tavsr_stream_t stream);
/* head of a captured search step: score[n] = -inf for the hypotheses that ended with the previous token (tok[n] == eos, or
 * *step_dev >= maxlen[n / K]: the previous iteration was their utterance's last), anc[n][*step_dev] = n + *step_dev * N.
 * tavsr_beam_reorder's hist (nullable) [hist_steps][3][N] int32 receives the token's record (token, extended slot, score
 * bits) at row *step_dev: the host rebuilds ended hypotheses from these back-pointers without reading the state back. */
int tavsr_beam_step_begin(float* score, const int64_t* tok, int32_t* anc, int32_t ld_a, const int32_t* maxlen, int32_t N, int32_t K,
                          int32_t eos, const int32_t* step_dev, tavsr_stream_t stream);
/* y = act(x) elementwise (the LM's Linear -> LayerNorm -> ReLU input layer); in place allowed */
int tavsr_act_fwd(const float* x, float* y, int64_t n, int32_t act, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step of the reference's harness (avsr_main.py:50-54): torch.optim.Adam(betas (0.9, 0.98), eps 1e-9) under
 * the Noam rate (src/schedulers/noam.py:29-46,72-81), fused over flat fp32 buffers.  `step` counts from 1 (bias
 * correction), `lr` is the Noam rate of that step, gradients are multiplied by grad_scale first (1/world_size under DP).
 * ------------------------------------------------------------------------------------------- */
int tavsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int64_t step, float grad_scale, tavsr_stream_t stream);
/* torch.optim.AdamW (the `optimizer: adamw` recipes, src/utils/scheduler.py:22-23): p *= 1 - lr * weight_decay before the Adam
 * update; weight_decay = 0 is tavsr_adam_step. */
int tavsr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int64_t step, float grad_scale, tavsr_stream_t stream);

/* Gradient buckets of the data-parallel exchange (tavsr/dp.py): tensor t = n[t] floats at ptrs[t], its slot in
 * the flat bucket starts at off[t].  to_flat != 0: flat <- tensors (pack before the all-reduce); else tensors <-
 * scale * flat (unpack, scale = 1 / world_size).  ptrs / off / n are DEVICE arrays; max_n = max n[t] sizes the grid. */
int tavsr_bucket_copy(float* const* ptrs_dev, const int64_t* off_dev, const int64_t* n_dev, int32_t ntensors, float* flat,
                      float scale, int32_t to_flat, int64_t max_n, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (csrc/dp.cpp): one process per GPU, RCCL over xGMI, the ONLY collective on
 * the path (the reference is single-process: avsr_main.py:27-58 steps one device).  RCCL is resolved at run time
 * (dlopen; a copy the process already holds - torch's - is shared), so the library loads on hosts without it.
 *   tavsr_dp_unique_id : rank 0 obtains the 128-byte communicator id; the caller ships it to every rank.
 *   tavsr_dp_init      : collective; every rank calls it on its own device with the same id.
 *   tavsr_dp_allreduce : in-place fp32 sum over all ranks of flat[0..n), enqueued on `stream` (no host sync): the flat
 *                        gradient buckets packed by tavsr_bucket_copy; 1/world rides on the unpack.
 *   tavsr_dp_broadcast : flat[0..n) of `root` to every rank (identical initial parameters).
 *   tavsr_dp_world     : ranks of the live communicator (0: none).   tavsr_dp_destroy: frees it.
 * Errors: 1000 + ncclResult_t for RCCL failures.
 * ------------------------------------------------------------------------------------------- */
int tavsr_dp_unique_id(void* id128);
int tavsr_dp_init(int32_t rank, int32_t nranks, const void* id128);
int32_t tavsr_dp_world(void);
int tavsr_dp_allreduce(float* flat, int64_t n, tavsr_stream_t stream);
int tavsr_dp_broadcast(float* flat, int64_t n, int32_t root, tavsr_stream_t stream);
int tavsr_dp_destroy(void);

/* dst[t][i] += src[t][i], t < ntensors: the gradients of parameters shared by the two modality streams of a tailored AV
 * layer (src/encoder/audiovisual/tailored/encoder_layer.py:118-274 applies the same FFN / norm modules to both streams;
 * autograd sums their gradients).  dst / src / n are HOST arrays (the pointers ride in the kernel arguments). */
int tavsr_multi_add(float* const* dst, const float* const* src, const int64_t* n, int32_t ntensors, tavsr_stream_t stream);
/* dst[t] <- src[t], nbytes[t] bytes each (any dtype), nbuffers <= 24, one launch; HOST tables as in tavsr_multi_add. */
int tavsr_multi_copy(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers, tavsr_stream_t stream);
/* ... and `ncounters` (<= 64) int64 words at `counters` incremented by one in the same launch (the step counters of a captured search
 * step: nothing in this launch reads them). */
int tavsr_multi_copy_inc(void* const* dst, const void* const* src, const int64_t* nbytes, int32_t nbuffers, int64_t* counters,
                         int32_t ncounters, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Train-mode dropout (every torch Dropout / F.dropout site of the path).  y[i] = keep_i ? x[i] / (1 - p) : 0 where
 * keep_i is a pure function of (seed_dev[0], offset + i) (Philox4x32-10): the SAME call on the upstream gradient is the
 * backward pass, no mask is stored.  seed_dev is a device-resident uint64; tavsr_rng_advance steps it (captured inside
 * a hipGraph it gives every replay fresh masks).  offset % 4 == 0; in place (y == x) allowed.
 * ------------------------------------------------------------------------------------------- */
int tavsr_dropout(const float* x, float* y, int64_t n, float p, const uint64_t* seed_dev, uint64_t offset,
                  tavsr_stream_t stream);
int tavsr_rng_advance(uint64_t* seed_dev, tavsr_stream_t stream);
/* tavsr_rng_advance that also writes the advanced value to step_seed_dev: the per-step seed copy the dropout sites of ONE
 * forward pass and of its backward pass read (a later forward advances seed_dev without touching earlier steps' masks). */
int tavsr_rng_step(uint64_t* seed_dev, uint64_t* step_seed_dev, tavsr_stream_t stream);
/* fused forms with the same mask as tavsr_dropout(x = t / dh, same p, seed, offset):
 *   tavsr_dropout_add    : y = a + alpha * dropout(t)        (x + ff_scale * dropout(f(x)), encoder_layer.py:194,309,314)
 *   tavsr_dropout_act_bwd: dz = dropout(dh) * act'(z)        (backward of dropout(act(z)), PositionwiseFeedForward) */
int tavsr_dropout_add(const float* a, const float* t, float* y, int64_t n, float p, float alpha, const uint64_t* seed_dev,
                      uint64_t offset, tavsr_stream_t stream);
int tavsr_dropout_act_bwd(const float* dh, const float* z, float* dz, int64_t n, float p, int32_t act,
                          const uint64_t* seed_dev, uint64_t offset, tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Error rates of the decode output: what src/evaluation/bootstrap_wer.py:3-16 obtains from the
 * tasas / tasasIntervalo programs (src/evaluation/tasas/tasas.c: gp() unit-cost alignment, tasa_ie; tasasIntervalo.c:
 * 1000 bootstrap resamples, 1.64 sigma).
 *   tavsr_edit_distance  : dist[p] = Levenshtein distance of ref[ref_off[p] .. ref_off[p+1]) and hyp[hyp_off[p] .. ) (symbol
 *       ids; offsets have n_pairs + 1 entries; max_len >= every sequence length, <= 4095).
 *   tavsr_bootstrap_rates: rates[it] = 100 * sum dist[idx] / sum reflen[idx] over n indices drawn uniformly with
 *       replacement (Philox, counter = (position, it), key = seed), it < iters.
 * ------------------------------------------------------------------------------------------- */
int tavsr_edit_distance(const int32_t* ref, const int64_t* ref_off, const int32_t* hyp, const int64_t* hyp_off, int32_t n_pairs,
                        int32_t max_len, int32_t* dist, tavsr_stream_t stream);
int tavsr_bootstrap_rates(const int32_t* dist, const int32_t* reflen, int32_t n, int32_t iters, uint64_t seed, double* rates,
                          tavsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Batch assembly: the video pipeline of avsr_main.py:168-179 (src/transforms/video_transforms.py:59-147 +
 * torchvision RandomCrop / RandomHorizontalFlip) fused with the padding of src/utils/avsr_dataloader.py:102-142.
 *   tavsr_video_prep: dst[t][y][x], t < Tpad, of one clip: for t < T the source pixel
 *       src[frames ? frames[t] : t][y0 + y][x0 + (flip ? tw-1-x : x)] (uint8 or float) through n_affine steps
 *       v = (v - mean[k]) / std[k]; frames with masked[t] != 0 take the clip's mean frame (mean over t < T of those values,
 *       computed first into mean_frame_ws [th*tw]); frames t >= T take `pad`.  mean / std are HOST arrays (they ride in the
 *       kernel arguments); src, frames, masked, mean_frame_ws, dst are device pointers.
 *   tavsr_add_noise : out = audio + (inv_snr * noise) * sqrt(mean(audio^2) / mean(noise^2))  (audio_transforms.py:126-133).
 * ------------------------------------------------------------------------------------------- */
int tavsr_video_prep(const void* src, int32_t is_u8, int32_t Ts, int32_t H, int32_t W, const int32_t* frames, int32_t T,
                     int32_t y0, int32_t x0, int32_t th, int32_t tw, int32_t flip, const float* mean, const float* std,
                     int32_t n_affine, const uint8_t* masked, float* mean_frame_ws, float* dst, int32_t Tpad, float pad,
                     tavsr_stream_t stream);
int tavsr_add_noise(const float* audio, const float* noise, float* out, int64_t n, float inv_snr, tavsr_stream_t stream);
/* Audio SpeedRate augmentation (src/transforms/audio_transforms.py:141-178: sox effects "speed f", "rate 16000" - the clip played
 * f times faster, resampled back to the sample rate): y[n] = sum_k x[k] c sinc(c (n f - k)) Kaiser_beta((n f - k) / W) with
 * c = rolloff * min(1, 1 / f) and W = zeros / c input samples; n_out = tavsr_resample_len(n_in, f) = round(n_in / f).  A windowed-sinc
 * restatement of sox's band-limited rate conversion (sox itself is not reproduced bit for bit: its polyphase filter design is its own). */
int64_t tavsr_resample_len(int64_t n_in, double factor);
int tavsr_resample_sinc(const float* x, int64_t n_in, float* y, int64_t n_out, double factor, double rolloff, int32_t zeros, double beta,
                        tavsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAVSR_H_ */
