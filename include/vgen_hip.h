/*
 * vgen_hip.h — C ABI of libvgen_hip.so: the MI355X (gfx950) hot path of the VGen
 * video-diffusion sampling loop (spatio-temporal UNet + AutoencoderKL + DDIM update).
 *
 * Boundary rules:
 *   - plain C: raw device pointers, explicit sizes/strides, dtype enums, a hipStream_t
 *     passed as void*.  No torch / C++ types in any signature.
 *   - every function only ENQUEUES work on `stream`; it never allocates, frees or
 *     synchronises.  The caller (PyTorch on the host side) owns every buffer.
 *   - return value: 0 = ok, >0 = hipError_t of the failed launch, <0 = VGEN_E_* below.
 *     vgen_last_error() returns a static human-readable string for the last failure.
 *
 * Canonical activation layout ("rows x channels", channels-last):
 *   a video feature map [B, C, F, H, W] of the reference is held as a row-major matrix
 *   [B*F*H*W, C]; row index m = ((b*F + f)*H + y)*W + x.  Spatial ops see B*F images,
 *   temporal ops see F frames of S=H*W rows each — both are views of the same buffer,
 *   so none of the reference's rearrange()/contiguous() copies exist here.
 *   Residual streams are fp32; GEMM/conv operands are 16-bit (bf16 default, fp16 optional).
 *
 * Each entry point cites the reference call site it replaces (paths relative to the
 * reference repo root).
 */
#ifndef VGEN_HIP_H
#define VGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGEN_ABI_VERSION 5

enum { VGEN_BF16 = 0, VGEN_F16 = 1, VGEN_F32 = 2 };

enum {
  VGEN_E_BADARG = -1,   /* shape / alignment / dtype constraint violated */
  VGEN_E_UNSUPPORTED = -2,
  VGEN_E_WORKSPACE = -3 /* caller-provided workspace too small */
};

int vgen_version(void);
const char* vgen_last_error(void);

/* ------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), fp32 in -> 16-bit out.
 * Replaces nn.GroupNorm(32, C)[+nn.SiLU] at tools/modules/unet/util.py:846,870 (ResBlock),
 * :1663-1681 (TemporalConvBlock_v2, 5-D: statistics across all frames), :329 / :1211
 * (Spatial/TemporalTransformer.norm, eps 1e-6, no SiLU), unet_t2v.py:205 (head) and
 * Normalize()+nonlinearity() at tools/modules/autoencoder.py:11-16.
 *
 * Input rows are the virtual channel-concat of two sources (x2 may be NULL, C2 = 0):
 *   row r = [ x1[r, 0:C1] | x2[r, 0:C2] ]  — this is how the decoder's
 *   torch.cat([x, xs.pop()], 1) (unet_t2v.py:269) is consumed without materialising it.
 * nb independent normalisation batches of S rows each (4-D GN: nb = B*F, S = H*W;
 * 5-D GN: nb = B, S = F*H*W).  C = C1 + C2, C % (4*groups)... only C % 4 == 0 and
 * C % groups == 0 are required; C <= 3072.
 * y   : [nb*S, C] 16-bit, = act(gamma * (x - mean) * rstd + beta), act = SiLU if silu != 0
 * raw : optional [nb*S, C] 16-bit plain cast of the concatenated input (feeds the 1x1
 *       skip_connection conv, util.py:885); NULL to skip.  raw_split != 0: two-term rows
 *       [nb*S, 2C] = [hi | lo], lo = round16(x - hi) (the models' two-term-activation modes;
 *       the skip conv's weight is then packed twice, see vgen_cast_split).
 * ws  : fp32 scratch, at least vgen_groupnorm_ws_bytes(nb, S) bytes.
 */
size_t vgen_groupnorm_ws_bytes(int64_t nb, int64_t S);
int vgen_groupnorm(const float* x1, int32_t C1, const float* x2, int32_t C2,
                   int64_t nb, int64_t S, int32_t groups, float eps,
                   const float* gamma, const float* beta, int32_t silu,
                   void* y, void* raw, int32_t raw_split, int32_t dtype,
                   float* ws, size_t ws_bytes, void* stream);

/* Same operator when the producer of x1 / x2 already left per-column statistics behind
 * (vgen_tapgemm_args.colstats): the statistics pass over the fp32 input — 40 % of the streaming
 * GroupNorm's HBM traffic — is replaced by a reduction of the column partials.
 *   cs1 : [nb*S / 64][2][C1] fp32 — for every 64-row slab of x1: plane 0 = column sums, plane 1 =
 *         column sums of squares (exactly what vgen_tapgemm writes); cs2 likewise for x2 (C2 > 0).
 * Requires S % 64 == 0 (slabs must not straddle two normalisation batches). */
int vgen_groupnorm_cs(const float* x1, int32_t C1, const float* cs1,
                      const float* x2, int32_t C2, const float* cs2,
                      int64_t nb, int64_t S, int32_t groups, float eps,
                      const float* gamma, const float* beta, int32_t silu,
                      void* y, void* raw, int32_t raw_split, int32_t dtype,
                      float* ws, size_t ws_bytes, void* stream);

/* Token + positional embedding of the CLIP text tower (tools/modules/clip_embedder.py:155-156):
 * out[r, :] = table[tokens[r], :] + pos[r % L, :], fp32; tokens int64 [rows], table [vocab, d], pos [L, d]. */
int vgen_embed_tokens(const int64_t* tokens, int64_t rows, int32_t L, int32_t d, int32_t vocab,
                      const float* table, const float* pos, float* out, void* stream);

/* LayerNorm over the last dim, fp32 in -> 16-bit out, or fp32 out with dtype VGEN_F32 (eps 1e-5 in the reference).
 * Replaces nn.LayerNorm norm1/2/3 of BasicTransformerBlock (util.py:692-694,700-704).
 * x [M, d] fp32 (row stride d), d % 4 == 0, d <= 2048. */
int vgen_layernorm(const float* x, int64_t M, int32_t d, float eps,
                   const float* gamma, const float* beta,
                   void* y, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Tap-GEMM: the one MFMA contraction kernel behind every Linear / Conv1d(k=1) / Conv2d 1x1,
 * Conv2d 3x3 (stride 1|2, optional folded nearest-2x upsample) and Conv3d (3,1,1).
 *
 *   out[m, n] = epi( sum_{tap} sum_{c<C1} A[src(m,tap), c] * W[n, tap*C1 + c]
 *                  + sum_{c<C2} A2[m, c] * W[n, taps*C1 + c] )
 *
 * mode VGEN_TAP_LINEAR : taps = 1, src(m,0) = m.
 *      nn.Linear at util.py:224-228 (to_q/k/v/out), :710 (GEGLU proj), :737 (FF out),
 *      :337,351 (SpatialTransformer proj_in/out), unet_t2v.py:93-96 (time_embed),
 *      util.py:862-868 (emb_layers); nn.Conv1d k=1 at util.py:1213,1229; 1x1 Conv2d at
 *      util.py:885 and autoencoder.py:309,397-416.
 * mode VGEN_TAP_CONV3X3: taps = 9 (ky*3+kx), rows m = (img*Ho + oy)*Wo + ox,
 *      iy = oy*stride + ky - pad_t, ix = ox*stride + kx - pad_l over a (Hi<<ups)x(Wi<<ups)
 *      virtual input whose pixel (iy,ix) is source pixel (iy>>ups, ix>>ups); out-of-range
 *      taps contribute zero.  nn.Conv2d 3x3 at util.py:848,874 (ResBlock), :946 (Downsample,
 *      stride 2 pad 1), :759+768 (Upsample: F.interpolate nearest x2 then conv),
 *      unet_t2v.py:112,207; autoencoder.py:286,296 (ResnetBlock), :449-459 (Upsample),
 *      :468-478 (Downsample: pad (0,1,0,1), stride 2 => pad_t = pad_l = 0).
 * mode VGEN_TAP_TEMPORAL3: taps = 3 (kt), rows m = (b*F + f)*S + p, source frame f+kt-1
 *      (zero outside [0,F)).  nn.Conv3d(C, C, (3,1,1), padding=(1,0,0)) at
 *      util.py:1665,1670,1675,1680 (TemporalConvBlock_v2).
 * The optional second K segment (A2, C2) with identity row mapping fuses the ResBlock's 1x1
 * skip_connection (util.py:885,920) into the out-conv.
 *
 * Constraints: C1 % 64 == 0, C2 % 64 == 0, lda/lda2 % 8 == 0 and in [0, 2^30), source rows < 2^31, 16-byte aligned pointers,
 * ldw % 8 == 0, K = taps*C1 + C2 <= 131008.  A, A2, W are `dtype` (VGEN_BF16 | VGEN_F16);
 * accumulation is fp32 on the MFMA units.
 *
 * Epilogue, applied in fp32 in this order:
 *   v = acc + bias[n] + rowbias[(m / rows_per_rb) * rowbias_ld + n] + residual[m*ldr + n]
 *   epilogue == VGEN_EPI_GEGLU: W rows are interleaved in blocks of 16 as
 *       [16 value rows | 16 gate rows] (packed index pn; value j <-> rows 32*(j/16)+j%16,
 *       gate j <-> +16), bias likewise; out[m, j] = v_value * gelu_erf(v_gate), j < N/2.
 *       (GEGLU.forward, util.py:712-714.)
 *   out is fp32 or 16-bit (out_dtype), row stride ldo.
 */
enum { VGEN_TAP_LINEAR = 0, VGEN_TAP_CONV3X3 = 1, VGEN_TAP_TEMPORAL3 = 2 };
enum { VGEN_EPI_NONE = 0, VGEN_EPI_GEGLU = 1 };

typedef struct vgen_tapgemm_args {
  int64_t M;
  int32_t N;
  int32_t dtype;
  const void* A;
  int64_t lda;
  int32_t C1;
  int32_t taps;
  int32_t mode;
  int32_t Hi, Wi, Ho, Wo, stride, pad_t, pad_l, ups; /* CONV3X3 */
  int32_t F;                                          /* TEMPORAL3 */
  int64_t S;                                          /* TEMPORAL3: rows per frame */
  const void* A2;
  int64_t lda2;
  int32_t C2;
  const void* W;
  int64_t ldw; /* W row stride in elements; 0 = dense (taps*C1 + C2) */
  const float* bias;
  const float* rowbias;
  int64_t rowbias_ld;
  int64_t rows_per_rb;
  const float* residual;
  int64_t ldr;
  void* out;
  int64_t ldo;
  int32_t out_dtype;
  int32_t epilogue;
  void* ws;        /* optional split-K workspace (fp32), see vgen_tapgemm_ws_bytes */
  size_t ws_bytes;
  int32_t crop_t;  /* CONV3X3 with ups: rows cropped from the top AND bottom of the upsampled image
                      before the conv (UpsampleSR600: x[..., 1:-1, :], util.py:801) */
  float* colstats; /* optional [ceil(M/64)][2][N] fp32: per 64-row slab of the FINAL fp32 output, plane 0
                      = column sums, plane 1 = column sums of squares (rows >= M excluded).  Lets the
                      GroupNorm that consumes `out` (util.py:846,870,1663-1681) skip its statistics pass
                      (vgen_groupnorm_cs).  Needs out_dtype = F32, no GEGLU, N % 4 == 0, ldo/ldr/
                      rowbias_ld % 4 == 0; the launch is then never split along K. */
  int32_t dualw;   /* 1: two-term weights.  W holds, for every 64-element K-tile of the contraction (in the K order
                      above), the 64 columns of W_hi followed by the 64 columns of W_lo = round16(W - W_hi): row
                      stride ldw >= 2 K (0 = dense 2 K).  The launch computes epi(A . (W_hi + W_lo)^T) with every A
                      K-tile staged once — the models' precision="high" mode (packed 16-bit weights are the largest
                      rounding in the UNet; with them as hi + lo pairs its output is within 1e-3 of the reference's
                      fp32 forward, tools/modules/unet/unet_t2v.py:210-277).  K <= 65504.  0: W is [N, K]. */
  int32_t split_out; /* 1: 16-bit output as TWO-TERM rows: out[m, n] = hi = round16(v), out[m, N + n] = round16(v - hi)
                      (row stride ldo >= 2 N) — what vgen_cast_split would make of the fp32 value, without the fp32
                      round trip.  The FF output + token stream that SpatialTransformer / TemporalTransformer.proj_out
                      consume (util.py:351,1229,737) in the models' two-term-activation modes.  Needs out_dtype 16-bit,
                      no GEGLU / colstats, N % 32 == 0, ldo % 8 == 0; never split along K. */
} vgen_tapgemm_args;

/* Launches whose tile count cannot fill the 256 CUs (small M: the 4x7 / 8x14 UNet levels) are
 * split along K; the partial fp32 tiles need `vgen_tapgemm_ws_bytes(args)` bytes of caller
 * scratch (0 = no split for this shape).  Without a large enough `ws` the call still succeeds
 * unsplit.  The reduction order is fixed, so results are deterministic. */
size_t vgen_tapgemm_ws_bytes(const vgen_tapgemm_args* args);
int vgen_tapgemm(const vgen_tapgemm_args* args, void* stream);

/* Tuning introspection (tools/autotune_gemm.py; not needed by callers).  query_plan: the launch plan vgen_tapgemm
 * would use for `args` -> out3 = {block shape (0 pp / 1 dual / 2 pp128 / 3 panel: csrc/panelgemm.hip / 4 pp256 / 5 q128: r06, reached
 * through the plan table only), BN, split-K}.  set_plans: replace the
 * measured-plan table consulted before the cost model; rows of 12 int64 {mode, M, N, C1, C2, taps, epilogue,
 * out_dtype, flags (residual | rowbias<<1 | colstats<<2), shape, bn, splitk}; n < 0 restores the compiled-in table. */
int vgen_tapgemm_query_plan(const vgen_tapgemm_args* args, int32_t* out3);
int vgen_tapgemm_set_plans(const int64_t* rows, int32_t n);

/* ------------------------------------------------------------------------------------
 * Fused attention, head_dim = 64, softmax(Q K^T * scale) V, no mask / dropout.
 * Replaces xformers.ops.memory_efficient_attention at util.py:254-259 (the batch
 * chunking at :248-257 is a workaround the fused kernel does not need).
 *
 * A "sequence" is identified by (batch index bi in [0,nbatch), head h in [0,heads)).
 * Element (row r, head h, lane e) of tensor X lives at
 *   X + (bi / inner) * X_bo + (bi % inner) * X_bi + r * X_rs + h*64 + e
 * which expresses, without copies:
 *   spatial self-attn   : bi = image, rows = pixels            (inner = 1)
 *   spatial cross-attn  : q as above; k/v = per-prompt context (inner = F, k_bi = 0)
 *   temporal self-attn  : bi = (b, pixel), rows = frames, row stride = H*W*ld (inner = H*W)
 * nq <= 16 && nk <= 16 selects the one-wave-per-sequence temporal kernel.
 */
typedef struct vgen_attn_args {
  const void* q;
  const void* k;
  const void* v;
  void* out;
  int32_t dtype;
  int32_t heads;
  int32_t nq, nk;
  int64_t nbatch;
  int64_t inner;
  int64_t q_rs, q_bo, q_bi;
  int64_t k_rs, k_bo, k_bi;
  int64_t v_rs, v_bo, v_bi;
  int64_t o_rs, o_bo, o_bi;
  float scale;
  int32_t causal; /* 1: key j contributes to query i only if j <= i (nn.MultiheadAttention with open_clip's
                     build_attention_mask, tools/modules/clip_embedder.py:157 `attn_mask=self.model.attn_mask`) */
} vgen_attn_args;

int vgen_attention(const vgen_attn_args* args, void* stream);

/* Row softmax: P[r, :] = softmax(S[r, :] * scale), fp32 in -> 16-bit out.
 * Single-head 512-channel VAE attention, autoencoder.py:430-437 (bmm, scale, softmax). */
int vgen_softmax_rows(const float* S, int64_t rows, int32_t cols, int64_t lds, float scale,
                      void* P, int64_t ldp, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Small elementwise / layout kernels.
 */

/* y = act(x) cast to 16-bit; act: 0 = identity, 1 = SiLU, 2 = exact (erf) GELU.  (nn.SiLU in time_embed /
 * emb_layers, unet_t2v.py:94, util.py:863; nn.GELU of the OpenCLIP text tower's mlp.)  GELU is a separate pass on
 * purpose: inside the tap-GEMM epilogue its polynomial cost the hot UNet instantiations 55-75 spilled VGPRs. */
int vgen_act_cast(const float* x, void* y, int64_t n, int32_t act, int32_t dtype, void* stream);

/* fp32 rows -> TWO-TERM 16-bit rows: out[m, c] = hi = round16(x[m, c]), out[m, lo_off + c] = round16(x[m, c] - hi),
 * c < C; x row stride ldx, out row stride ldo (elements).  The A-side counterpart of vgen_tapgemm_args.dualw: a GEMM
 * over [A_hi | A_lo] against a weight repeated [W | W] computes (A_hi + A_lo) W^T — an fp32-accurate operand where the
 * models' precision modes want one (the token stream entering SpatialTransformer / TemporalTransformer.proj_out,
 * util.py:351,1229; the raw input of the ResBlock's 1x1 skip_connection, util.py:885,920). */
int vgen_cast_split(const float* x, int64_t M, int32_t C, int64_t ldx, void* out, int64_t ldo, int32_t lo_off,
                    int32_t dtype, void* stream);

/* sinusoidal_embedding (util.py:178-190): out[b, :] = [cos(t_b * w_i) | sin(t_b * w_i)],
 * w_i = 10000^(-i/half); out is `dtype` (VGEN_BF16 | VGEN_F16 | VGEN_F32).  t is fp32 [B]. */
int vgen_timestep_embedding(const float* t, int32_t B, int32_t dim, void* out, int32_t dtype,
                            void* stream);

/* Small-batch fp32 linear, out[r, j] = bias[j] + sum_k act(x[r, k]) * W[j, k] (+ add[r, j]); act_in: 0 =
 * identity, 1 = SiLU applied to x.  x [n, K], W [N, K], out / add [n, N], all fp32 contiguous, K % 4 == 0.
 * The time / fps embedding MLPs (unet_t2v.py:93-101, 244-245) and the ResBlocks' emb_layers (util.py:862-868,
 * all 22 as one [sum(Cout), 1280] matrix): B rows per step, or every timestep once for a session's table.  The
 * summation order of an output element is independent of n. */
int vgen_linear_f32(const float* x, int32_t n, int32_t K, const float* W, const float* bias, int32_t N,
                    int32_t act_in, const float* add, float* out, void* stream);

/* im2col of a small-channel 3x3/pad-1/stride-1 conv into a [M, Kpad] 16-bit matrix
 * (Kpad % 64 == 0, columns (ky*3+kx)*Cin + c, zero padded), so that the 4->320 input conv
 * (unet_t2v.py:112) and the VAE conv_in (autoencoder.py:605) run on the tap-GEMM.
 * Source element (img = bo*Fi + fi, c, y, x) at src + bo*s_bo + fi*s_fi + c*s_c + y*s_y + x*s_x
 * (fp32) — covers both [B,C,F,H,W] latents and [N,H,W,C] maps.
 * split = 1: three 9*Cin-column segments [hi | lo | hi] (hi = the 16-bit rounding of the value, lo = the 16-bit rounding
 * of what it lost; Kpad >= 27*Cin) for weights packed [W_hi | W_hi | W_lo]: the stem conv then sees fp32 inputs and
 * weights to ~2^-22 at 3x a negligible K. */
int vgen_im2col3x3_small(const float* src, int64_t nimg, int32_t Fi, int32_t Cin, int32_t H,
                         int32_t W, int64_t s_bo, int64_t s_fi, int64_t s_c, int64_t s_y,
                         int64_t s_x, void* out, int32_t Kpad, int32_t dtype, int32_t split, void* stream);

/* out[p, co] = sum_ci Wm[co, ci] * in[p, ci] + b[co], tiny channel counts (<= 16), fp32.
 * Source/destination addressed with the same (bo, fi, c, y, x) strides as above so the op
 * doubles as a layout change (post_quant_conv / quant_conv 1x1, autoencoder.py:50-51,
 * NCHW<->rows).  */
int vgen_pointwise_small(const float* src, int64_t nimg, int32_t Fi, int32_t Cin, int32_t H,
                         int32_t W, int64_t s_bo, int64_t s_fi, int64_t s_c, int64_t s_y,
                         int64_t s_x, const float* Wm, const float* b, int32_t Cout, float* dst,
                         int64_t d_bo, int64_t d_fi, int64_t d_c, int64_t d_y, int64_t d_x,
                         void* stream);

/* Fused classifier-free-guidance combine + DDIM update on the fp32 latent, replicating the
 * arithmetic ORDER of DiffusionDDIM.p_mean_variance / ddim_sample
 * (tools/modules/diffusions/diffusion_ddim.py:157-162, 194-197, 230-240) op for op in
 * fp32 without FMA contraction, for mean_type in {eps, v, x0}:
 *   out  = u + g*(y - u)                     (guide != 0; else out = y)
 *   x0   = c[0]*xt - c[1]*out  (v)  |  c[0]*xt - c[1]*out (eps, other coefs) | out (x0)
 *   eps  = (c[2]*xt - x0) / c[3]
 *   xt_1 = sqrt(c[4])*x0 + sqrt(1 - c[4] - c[5]^2)*eps + c[6]*c[5]*noise
 * coef = 7 fp32 per batch element b: [a0, a1, sqrt_recip, sqrt_recipm1, alpha_prev, sigma, mask].
 * All tensors contiguous fp32 with `per_b` elements per batch element; noise may be NULL
 * when every sigma is 0.  mean_type: 0 = eps, 1 = v, 2 = x0. */
int vgen_cfg_ddim_step(const float* xt, const float* y, const float* u, const float* noise,
                       const float* coef, float guide, int32_t use_guide, int32_t mean_type,
                       int64_t B, int64_t per_b, float* xt_1, float* x0_out, void* stream);

/* The same update addressed for a sampling session (vgen_amd/session.py) whose denoise step is ONE hipGraph:
 *   xt      : batch element b at xt + b*xt_bstride (per_b contiguous floats) — e.g. the latent channels of
 *             unit slot b inside the UNet's stacked input [units, C_stem, F, H, W];
 *   t_idx   : optional int64[B]; coefficient row of b is coef[t_idx[b]*7 ..] (a table over all timesteps, so
 *             the captured graph needs no per-step host work) instead of coef[b*7 ..];
 *   xt_1 / x0_out : optional contiguous [B, per_b] outputs;
 *   rep     : optional; x_{t-1} of b is also written at rep + g*rep_gstride + b*rep_bstride for g < nrep —
 *             the cond / uncond unit slots of the NEXT step's UNet batch (may alias xt).
 * Same arithmetic, same order. */
int vgen_cfg_ddim_step_units(const float* xt, int64_t xt_bstride, const float* y, const float* u,
                             const float* noise, const float* coef, const int64_t* t_idx, float guide,
                             int32_t use_guide, int32_t mean_type, int64_t B, int64_t per_b, float* xt_1,
                             float* x0_out, float* rep, int32_t nrep, int64_t rep_gstride,
                             int64_t rep_bstride, void* stream);

/* DiagonalGaussianDistribution.sample() * scale (autoencoder.py:212-225, 19-27):
 * moments rows [P, 2*zc] fp32 (mean | logvar), logvar clamped to [-30, 20],
 * z = (mean + exp(0.5*logvar) * noise) * scale written as [nimg, zc, H, W] fp32;
 * noise has the output layout. */
int vgen_gaussian_sample(const float* moments, const float* noise, int64_t nimg, int32_t zc,
                         int64_t HW, float scale, float* z, void* stream);

/* ------------------------------------------------------------------------------------
 * GaussianDiffusion (sigma-parametrised VP diffusion) pieces, tools/modules/diffusions/
 * diffusion_gauss.py:163-247 (denoise) and :85-142 (sample_dpmpp_2m_sde).
 *
 * vgen_cfg_stats: out = use_guide ? u + guide*(y - u) : y, plus per-sample partial sums of
 *   y, y^2, out, out^2 in `ws` (fp64; vgen_cfg_stats_ws_bytes(B) bytes) for guide_rescale.
 * vgen_gauss_x0 : optional guide_rescale (rescale < 0 disables; arXiv:2305.08891, :212-218)
 *   out *= rescale * std(y)/(std(out)+1e-12) + (1-rescale), then
 *   x0 = (xt - sigma*out)/alpha (pred_type 0 'eps') | alpha*xt - sigma*out (1 'v') | out (2 'x0'),
 *   eps = (xt - alpha*x0)/sigma (optional).  coef = [B][2] fp32 (alpha, sigma).
 * vgen_lincomb4 : out = ca*a + cb*b + cc*c + cd*d (NULL operands skipped, fp32, no contraction):
 *   scalings around the model call of DPM-Solver++(2M).
 * vgen_dpmpp2m_sde_step: ONE solver update (diffusion_gauss.py:122-139), every intermediate rounded where the
 *   reference's three tensor statements round (ABI 5: r03's form followed three lincomb4 calls instead, 1-2 ulp off):
 *   r = ca*x + cb*denoised;  old_denoised != NULL (2M correction, :126-134): r = r + cc*(denoised - old_denoised), the
 *   difference rounded first;  noise != NULL (SDE term, :136-139): r = r + ((noise*cn1)*cn2)*cn3.  ca = sigma_next /
 *   sigma * exp(-eta h), cb = -expm1(-h - eta h), cc = midpoint / Heun coefficient * (1 / r), cn1 = sigma_next, cn2 =
 *   sqrt(-expm1(-2 eta h)), cn3 = s_noise — host scalars in the reference's own expressions.
 */
/* FreeU-style skip filter of UNetSD_SR600 (unet_sr600.py:30-49, 276-287), on rows [nimg*H*W, C] fp32:
 * Fourier_filter(x, threshold=1, scale) multiplies the 2x2 block of centred-spectrum bins
 * {-1,0} x {-1,0} by `scale` and keeps the real part of the inverse FFT; being linear in 4 bins it is
 * computed exactly as  y = x + (scale-1)/(H*W) * Re[sum_{u,v in {0,-1}} X[u,v] e^{2 pi i (u h/H + v w/W)}]
 * from 7 weighted sums per (image, channel) — no FFT.  ws: 7*nimg*C floats. */
int vgen_lowfreq_filter(const float* x, int64_t nimg, int32_t H, int32_t W, int32_t C, float scale,
                        float* y, float* ws, size_t ws_bytes, void* stream);
/* x[:, c0:c1] *= s in place on rows [M, C] fp32 (unet_sr600.py:278,284: backbone half-channel boost). */
int vgen_scale_channels(float* x, int64_t M, int32_t C, int32_t c0, int32_t c1, float s, void* stream);

/* ------------------------------------------------------------------------------------
 * Condition stems ahead of the trunk (prompt constants: evaluated once per sampling session).  fp32, NCHW frames.
 *
 * vgen_conv3x3_small: y = act(conv2d(x, w, b, stride, padding = 1)), x [n, Cin, H, W], w [Cout, Cin, 3, 3],
 *   act 0 = none, 1 = SiLU — the nn.Conv2d(+nn.SiLU) of local_image_concat / local_image_embedding
 *   (unet_i2vgen.py:116-132) and of the depth / motion / canny / mask / sketch / local_image stems
 *   (unet_videolcm.py:294-372).
 * vgen_adaptive_avgpool2d: nn.AdaptiveAvgPool2d((Ho, Wo)) over `planes` = n*C planes of H x W (same stems).
 * vgen_frame_transformer: ONE layer of TransformerV2 over the frame axis of every pixel (util.py:1396-1453,
 *   used at unet_i2vgen.py:289-293 and unet_videolcm.py:606-699): x tokens [B, F, d, HW] (frames of d channels),
 *   x <- to_out(MHA(LayerNorm(x))) + x ; x <- W2 gelu(W1 x + b1) + b2 + x.  wqkv [3*heads*dim_head, d] (no bias),
 *   wout [d, heads*dim_head] / bout [d] (both NULL when heads == 1 && dim_head == d: identity), w1 [hidden, d],
 *   w2 [d, hidden].  last == 0: y in the input layout; last == 1: y is the trunk's stem-channel layout
 *   [B, d, F, HW], value * out_scale, added to y when `accumulate` (the composer sums its stems,
 *   unet_videolcm.py:612-699; UNetSD_I2VGen adds its map twice, unet_i2vgen.py:294-295).
 *   F <= 32, d <= 16, heads*dim_head <= 32, hidden <= 64. */
int vgen_conv3x3_small(const float* x, int64_t n, int32_t Cin, int32_t H, int32_t W, const float* w,
                       const float* b, int32_t Cout, int32_t stride, int32_t act, float* y, void* stream);
int vgen_adaptive_avgpool2d(const float* x, int64_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                            float* y, void* stream);
int vgen_frame_transformer(const float* x, int64_t B, int32_t F, int32_t d, int64_t HW, int32_t heads,
                           int32_t dim_head, int32_t hidden, const float* ln_w, const float* ln_b,
                           const float* wqkv, const float* wout, const float* bout, const float* w1,
                           const float* b1, const float* w2, const float* b2, float* y, int32_t last,
                           float out_scale, int32_t accumulate, void* stream);

/* Decoded frames to displayable bytes — the step right after AutoencoderKL.decode in every engine:
 * utils/video_op.py:181-188 (`gen_video.mul_(std).add_(mean)`, `clamp_(0, 1)`, `* 255.0`, rearrange
 * 'b c f h w -> b f h w c', `.astype('uint8')`).  x: decoder output rows [rows = n*H*W, C] fp32 (row stride
 * ldx) — rows are already in (frame, y, x, channel) order, so out [rows*C] uint8 IS the [n, H, W, C] byte
 * image; mean / stdv: C device floats.  Bit-exact with the reference arithmetic (fp32, truncation). */
int vgen_frames_u8(const float* x, int64_t rows, int32_t C, int64_t ldx, const float* mean,
                   const float* stdv, void* out, void* stream);

/* Glue of a sampling session's step graph (vgen_amd/session.py), so that a captured step holds only this library's
 * launches.  vgen_repeat_rows: dst[g*bytes .. (g+1)*bytes) = src[0 .. bytes) for g < G — the rows of the context-free
 * prefix shared by a classifier-free-guidance pair (diffusion_ddim.py:157-158 evaluates it per branch) fanned out to
 * the G units.  vgen_gather_rows_f32: out[r, :] = table[idx[r], :] — the time-embedding row biases of timestep t
 * (unet_t2v.py:93-96,244-245 folded over all integer timesteps once per weight load); idx clamped to the table. */
int vgen_repeat_rows(const void* src, int64_t bytes, int32_t G, void* dst, void* stream);
int vgen_gather_rows_f32(const float* table, int64_t nrows_table, int64_t cols, const int64_t* idx, int64_t rows,
                         float* out, void* stream);

size_t vgen_cfg_stats_ws_bytes(int64_t B);
int vgen_cfg_stats(const float* y, const float* u, float guide, int32_t use_guide, int64_t B,
                   int64_t per_b, float* out, void* ws, size_t ws_bytes, void* stream);
int vgen_gauss_x0(const float* xt, const float* out, const void* ws, float rescale, const float* coef,
                  int32_t pred_type, int64_t B, int64_t per_b, float* x0, float* eps, void* stream);
int vgen_lincomb4(const float* a, const float* b, const float* c, const float* d, float ca, float cb,
                  float cc, float cd, float* out, int64_t n, void* stream);
int vgen_dpmpp2m_sde_step(const float* x, const float* denoised, const float* old_denoised, const float* noise,
                          float ca, float cb, float cc, float cn1, float cn2, float cn3, float* out, int64_t n,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VGEN_HIP_H */
