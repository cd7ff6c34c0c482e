/*
 * neurosis_hip.h -- C-ABI of libneurosis_hip.so: the MI355X (gfx950) kernels behind the SDXL
 * training-step hot path of neggles/neurosis.
 *
 * Conventions
 *   - every entry point returns 0 on success, non-zero on error; nk_last_error() gives the message.
 *     Nothing falls back to a CPU path: a bad argument or a failed launch is an error.
 *   - all pointers are DEVICE pointers unless the name says host; `stream` is a hipStream_t passed as void*.
 *   - activations are bf16 (raw uint16 bits), physically channels-last: an image batch is [N][H][W][C],
 *     a token matrix is [rows][C].  Parameters are fp32 masters with bf16 shadows; conv weights are
 *     [Cout][KH][KW][Cin] (= an OIHW tensor in torch.channels_last memory format), Linear weights [out][in].
 *   - gradients of parameters are fp32 and are written ("accumulate=0") or added ("accumulate=1") in place.
 *   - kernels are asynchronous on `stream`; no entry point synchronises, allocates or frees device memory
 *     (so every one of them can be captured into a hipGraph).
 *
 * Each declaration cites the reference call site it replaces (paths relative to
 * /root/reference/src/neurosis/).
 */
#ifndef NEUROSIS_HIP_H
#define NEUROSIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

const char* nk_last_error(void);
int nk_abi_version(void);     /* 5 (round 6: + the four *_geglu*_s entry points; bumped whenever entry points are added or change) */

/* ------------------------------------------------------------------------------------------------
 * nn.Linear  (modules/attention.py:53,65,70,204-209,283-290,618,639; modules/diffusion/openaimodel.py:273-279,
 * 586-590,612-618).  1x1 Conv2d on channels-last data is the same contraction (openaimodel.py:301 skip_connection,
 * modules/diffusion/model.py:150-153 q/k/v/proj_out, :108 nin_shortcut).
 * ---------------------------------------------------------------------------------------------- */
/* y[M,N] = alpha * x[M,K] @ w[N,K]^T + bias[N] + residual[M,N]      (bias, residual optional = NULL) */
int nk_linear_fwd(const void* x, const void* w, const float* bias, const void* residual, void* y,
                  int M, int N, int K, long ldx, long ldw, long ldr, long ldy, float alpha, void* stream);
/* dx[M,K] = dy[M,N] @ w[N,K] + dx_add[M,K]                          (dx_add optional) */
int nk_linear_dgrad(const void* dy, const void* w, const void* dx_add, void* dx, int M, int N, int K,
                    long lddy, long ldw, long ldadd, long lddx, void* stream);
/* dw[N,K] (+)= dy[M,N]^T @ x[M,K]   fp32.  accumulate: 0 overwrite, 1 add, 2 destination is known to be all zero */
/* nk_linear_dgrad of FeedForward.net[2] with the GEGLU backward in its epilogue (modules/attention.py:50-74): du[M][2I] =
 * [d * gelu(g) | d * a * gelu'(g)], d = dy[M][N] @ w[N][I], u = [a | g] [M][2I] saved by the forward */
/* FeedForward.net[0] = GEGLU (modules/attention.py:50-57) in one launch: u[M, 2I] = x @ w^T + bias (kept for the backward) and
 * h[M, I] = u[:, :I] * gelu_erf(u[:, I:]).  Only for shapes nk_linear_fwd_geglu_ok() accepts (whole 256-column tiles on the 256 x 256
 * two-group kernel); otherwise NK_ERR_ARG: call nk_linear_fwd then nk_geglu_fwd. */
long nk_linear_fwd_geglu_ok(int M, int I, int K);
int nk_linear_fwd_geglu(const void* x, const void* w, const float* bias, void* u, void* h, int M, int I, int K, long ldx, long ldw, long ldu,
                        long ldh, void* stream);
int nk_linear_dgrad_geglu(const void* dy, const void* w, const void* u, void* du, int M, int N, int I, long lddy, long ldw, long ldu,
                          long lddu, void* stream);
/* The same pair on the SAVED-DERIVATIVE form of the projection output (round 6).  torch autograd, which is what backs FeedForward in the
 * reference (modules/attention.py:50-74; no backward source), saves a and g and evaluates gelu'(g) in the backward; here the forward -- which
 * has the normal cdf and its exponential in registers already -- writes s[M, 2I] = [gelu(g) | a * gelu'(g)] INSTEAD of u = [a | g] (u itself is
 * never written), and the input gradient of net[2] becomes du[M, 2I] = [d * s[:, :I] | d * s[:, I:]]: two products per element where the
 * erf-GELU derivative took ~22 vector instructions beside an idle matrix pipe.  s is rounded to bf16 like every stored activation.
 * nk_linear_fwd_geglu_s takes the shapes nk_linear_fwd_geglu_ok() accepts; elsewhere: nk_linear_fwd then nk_geglu_fwd_s. */
int nk_linear_fwd_geglu_s(const void* x, const void* w, const float* bias, void* s, void* h, int M, int I, int K, long ldx, long ldw, long lds,
                          long ldh, void* stream);
int nk_linear_dgrad_geglu_s(const void* dy, const void* w, const void* s, void* du, int M, int N, int I, long lddy, long ldw, long lds,
                            long lddu, void* stream);
int nk_linear_wgrad(const void* dy, const void* x, float* dw, int M, int N, int K, long lddy, long ldx,
                    long lddw, int accumulate, void* stream);
/* ... with the bias gradient dbias[N] (+)= column sums of dy from the same launch: the weight-gradient kernel already stages the dy panel;
 * the first column tile of each row block sums it with one extra MFMA per k sub-step against ones (no separate reduction launches) */
int nk_linear_wgrad_bias(const void* dy, const void* x, float* dw, float* dbias, int M, int N, int K, long lddy, long ldx, long lddw,
                         int accumulate, void* stream);

/* Health of the persistent stream-K tile kernel (gemm.hip): a workgroup that finishes a K-split tile waits -- bounded --
 * for the partial tiles of lower-indexed workgroups.  Returns 0 if no launch ever gave up that wait, 1 otherwise
 * (that launch's output is wrong), -1 on a device error.  Synchronises the device; meant for tests and end-of-run checks. */
int nk_gemm_sk_status(void);

/* Backward-health word: fail-closed handling of such a give-up on the TRAINING path (no reference counterpart; the
 * reference's cuBLAS / cuDNN calls cannot produce a partial tile).  The workgroup that gives up poisons its output tile
 * with NaN and raises one device word; nk_adafactor_chunk / nk_adamw_flat gate every kernel of the update on that word
 * (masters, optimizer state and bf16 shadows stay untouched) and return NK_ERR_HEALTH (3) from the NEXT call on, without
 * synchronising (a stream-ordered snapshot of the word to pinned memory).
 *   nk_health_status(): 0 healthy / 1 raised / -1 device error; synchronises.
 *   nk_health_clear():  synchronises, lowers the word and every stream-K flag, so a caller that has decided to go on can.
 *   nk_debug_raise_health(stream): test hook, raises the word the way a kernel would. */
int nk_health_status(void);
int nk_health_clear(void);
/* Data parallelism: nk_health_export writes this rank's word (0 / 1) to a device int32 the caller owns; the caller reduces it over the ranks
 * (MAX) and hands the result to nk_health_import, which raises the local word if any rank's was raised.  Both stream-ordered. */
int nk_health_export(int* dst, void* stream);
int nk_health_import(const int* src, void* stream);
int nk_debug_raise_health(void* stream);
/* Diagnostic: writes the device's 100 MHz constant clock (s_memrealtime) to *dst, stream-ordered and capturable into a hipGraph
 * (tools/step_timeline.py: where the replayed backward segments of the two streams lie in time, untraced). */
int nk_debug_stamp(unsigned long long* dst, void* stream);

/* `count` (<= 8) weight gradients of identical shape in ONE launch: the three 1280x1280 projections of a transformer
 * block are 100 tiles each, far below one workgroup per CU on their own.  dy / x / dw are HOST arrays of device pointers. */
/* `count` (<= 8) bias-free nn.Linear forwards of identical shape in ONE launch: y[i] = x[i] @ w[i]^T.  Serves the key / value
 * projections of cross-attention (reference modules/attention.py:383-385, `k = self.to_k(context); v = self.to_v(context)`): the
 * context is the same for all 70 transformer blocks of a step and does not depend on the UNet's activations, so UNetModel.fwd
 * projects it for every block up front, eight blocks per launch (308 x 2560 x 2048 each: 60 tiles alone, 480 together). */
/* column sums of `nbatch` consecutive row blocks of M rows each: out[b][N] (+)= sum_rows dy[b*M + r][:].  The per-image gradient of
 * the ResBlock embedding projection (h += emb_out[:, :, None, None], reference openaimodel.py:331-333 -> d emb_out[n] = sum over
 * image n's pixels) in one pair of launches instead of one per image.  ws: nbatch * nk_colsum_ws_floats(M, N) floats. */
int nk_colsum_batched(const void* dy, float* out, float* ws, long M, int N, long ld, int nbatch, int accumulate, void* stream);

int nk_linear_fwd_batched(const void* const* x, const void* const* w, void* const* y, int count, int M, int N, int K,
                          long ldx, long ldw, long ldy, void* stream);

/* dbias: NULL, or `count` pointers (NULL entries allowed) to the layers' bias gradients [N], summed in the same launch */
int nk_linear_wgrad_batched(const void* const* dy, const void* const* x, float* const* dw, float* const* dbias, int count, int M, int N,
                            int K, long lddy, long ldx, long lddw, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * nn.Conv2d as implicit GEMM  (openaimodel.py:124 Upsample.conv, :183-190 Downsample.op, :247-301 ResBlock convs,
 * :622-624 input conv, :797-801 out conv; model.py:71-79 asymmetric-pad stride-2 conv, :98-102, :519, :540 VAE convs).
 * upsample=1 fuses F.interpolate(scale_factor=2, mode="nearest") (openaimodel.py:140) into the gather.
 * ---------------------------------------------------------------------------------------------- */
typedef struct NkConvDesc {
  int N, H, W, Cin;   /* input  x [N][H][W][Cin] (before the optional virtual 2x upsample) */
  int Cout, KH, KW;   /* weight w [Cout][KH][KW][Cin] */
  int stride;         /* 1 or 2 */
  int pad_t, pad_l;   /* zero padding on top / left; bottom / right are implied by Ho, Wo */
  int Ho, Wo;         /* output y [N][Ho][Wo][Cout] */
  int upsample;       /* 1: x is read as its 2x nearest-neighbour upsampling */
} NkConvDesc;
/* y = conv(x, w) + bias[Cout] + rowvec[n][Cout] + residual[N][Ho][Wo][Cout]   (all three optional) */
int nk_conv2d_fwd(const NkConvDesc* d, const void* x, const void* w, const float* bias, const void* rowvec,
                  const void* residual, void* y, void* stream);
/* nk_conv2d_fwd that also emits, from its epilogue, the GroupNorm sums of its OUTPUT as per-tile partials [N][tiles][2 * stats_groups]
 * (entry 2g = sum, 2g+1 = sum of squares of the bf16-rounded values; tiles = nk_conv2d_stats_tiles): the GroupNorm that reads y then needs no
 * statistics pass (openaimodel.py:247-283 in_layers conv -> out_layers GroupNorm; model.py:116-124 conv1 -> norm2).  3 x 3 / stride 1 /
 * padding 1 convolutions the halo-tile kernel takes (conv_halo.h); nk_conv2d_stats_tiles returns 0 where it is not available (run
 * nk_conv2d_fwd and nk_groupnorm_fwd then); with stats_groups = 0 it tells whether the halo-tile kernel takes the shape at all. */
long nk_conv2d_stats_tiles(const NkConvDesc* d, int stats_groups);
int nk_conv2d_fwd_stats(const NkConvDesc* d, const void* x, const void* w, const float* bias, const void* rowvec, const void* residual,
                        void* y, float* stats_part, int stats_groups, void* stream);
/* 3 x 3 / stride 1 / padding 1 forward for images with 3 or 4 real channels stored padded to 8 (x [N][H][W][8], w [Cout][3][3][8]): the
 * first convolutions of the VAE encoder (model.py:519) and of the UNet (openaimodel.py:622-624) as a register-resident FMA kernel --
 * K = 27 / 36 gives the MFMA engine nothing to do, and its gather spent milliseconds on a 1 GB output. */
int nk_conv3x3_few_channels_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cout, int cin_real,
                                void* stream);
/* wt[Cin][KH*KW][Cout], taps mirrored (tap t of w lands at tap KH*KW-1-t): the weights with which a stride-1 "same" convolution's
 * INPUT GRADIENT is itself such a convolution, dx = nk_conv2d_fwd(dy, wt) with the channel roles swapped -- how the 3 x 3 input
 * gradients of the ResBlocks (autograd of openaimodel.py:247-301) reach the halo-tile forward kernel. */
int nk_conv_weight_flip(const void* w, void* wt, int Cout, int Cin, int taps, void* stream);
/* nk_conv2d_dgrad with those weights, on the halo-tile forward kernel; _ok = 1 where it applies (3 x 3, stride 1, padding 1, whole
 * 64-channel slabs, images the 32-pixel-wide tiles cover), else use nk_conv2d_dgrad */
long nk_conv2d_dgrad_flipped_ok(const NkConvDesc* d);
int nk_conv2d_dgrad_flipped(const NkConvDesc* d, const void* dy, const void* wt, void* dx, void* stream);
/* dx over the conv input grid ([N][2H][2W][Cin] when upsample=1: follow with nk_upsample2x_bwd) */
int nk_conv2d_dgrad(const NkConvDesc* d, const void* dy, const void* w, void* dx, void* stream);
/* dw[Cout][KH][KW][Cin] (+)= ...  fp32 */
int nk_conv2d_wgrad(const NkConvDesc* d, const void* dy, const void* x, float* dw, int accumulate, void* stream);
/* ... with the bias gradient dbias[Cout] (+)= sum over pixels of dy from the same launch */
int nk_conv2d_wgrad_bias(const NkConvDesc* d, const void* dy, const void* x, float* dw, float* dbias, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention  softmax(q k^T * scale) v, no dropout; no mask
 * (modules/attention.py:410-412 TorchSDPCrossAttention, :337-352 MemoryEfficientCrossAttention) or, forward only, the causal
 * mask of the CLIP text transformers (models/text_encoder/clip.py:311-343 attn_mask; HF CLIPTextModel's causal mask).
 * q/k/v/o are token-major [B][L][H*D] views with explicit row and batch strides (elements), so they can be
 * column slices of a fused projection buffer.  lse is [B][H][Lq] fp32 (saved for the backward).
 * ---------------------------------------------------------------------------------------------- */
typedef struct NkAttnDesc {
  int B, H, Lq, Lk, D;          /* D % 8 == 0, D <= 160; or D == 512 (the VAE mid block's single head, model.py:224-243: forward attn512.h -- lse may be NULL when no backward follows --, backward attn512_bwd.h) */
  long sq, sk, sv, so;          /* row strides of q, k, v, o */
  long bq, bk, bv, bo;          /* batch strides */
  long sdq, sdk, sdv, sdo;      /* backward only: row strides of dq, dk, dv, do */
  long bdq, bdk, bdv, bdo;      /* backward only: batch strides */
  float scale;                  /* D^-0.5 */
  int causal;                   /* forward only: key j contributes to query i iff j <= i (Lq == Lk) */
} NkAttnDesc;
int nk_attention_fwd(const NkAttnDesc* d, const void* q, const void* k, const void* v, void* o, float* lse,
                     void* stream);
/* delta_ws: uninitialised fp32 workspace of nk_attention_bwd_ws_floats(d) elements (row dots + cross-attention partials) */
long nk_attention_bwd_ws_floats(const NkAttnDesc* d);
int nk_attention_bwd(const NkAttnDesc* d, const void* q, const void* k, const void* v, const void* o,
                     const float* lse, const void* d_o, void* dq, void* dk, void* dv, float* delta_ws, void* stream);
/* in-place row softmax on bf16 [M][L]: unfused single-head attention = nk_linear_fwd (q k^T) -> nk_softmax_rows -> nk_linear_dgrad (p v).
 * Serves head dims the flash kernels do not take, and the chunked recomputing backward of the VAE mid block (d = 512) beyond 2 048 tokens per
 * sample (ops.attention512_fwd; up to there nk_attention_bwd's flash kernels run); d = 512 forward and backward are nk_attention_fwd / _bwd. */
int nk_softmax_rows(void* s, long M, int L, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32, C) (+ fused SiLU) on channels-last x [N][HW][C]
 * (openaimodel.py:247-250,281-283,797-799; attention.py:612 with eps 1e-6, no SiLU; layers.py:5-7 Normalize).
 * mean/rstd: [N][G] fp32 (saved for backward).  ws: uninitialised fp32 workspace of nk_groupnorm_ws_floats elements
 * (statistics are reduced through per-block partials, not atomics: deterministic, no memset).
 * Backward adds into dgamma/dbeta (fp32) and optionally adds dx_add into dx.
 * ---------------------------------------------------------------------------------------------- */
long nk_groupnorm_ws_floats(int N, int HW, int C, int G); /* fp32 elements of `ws` (per-block partial sums) */
int nk_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     float* ws, int N, int HW, int C, int G, float eps, int silu, void* stream);
/* The forward in separable passes (what nk_groupnorm_fwd runs back to back): sums [N][2G] (entry 2g = sum, 2g+1 = sum of squares over
 * HW * C/G elements) from x, or from a convolution's per-tile partials; and the normalisation given the sums. */
int nk_groupnorm_sums(const void* x, float* sums, float* ws, int N, int HW, int C, int G, void* stream);
long nk_groupnorm_sums_ws_floats(int N, int nparts, int G);
int nk_groupnorm_sums_from_parts(const float* part, float* sums, float* ws, int N, int nparts, int G, void* stream);
int nk_groupnorm_apply(const void* x, const float* sums, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int N, int HW, int C, int G, float eps, int silu, void* stream);
int nk_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta, float* ws,
                     int N, int HW, int C, int G, int silu, int accumulate, void* stream);

/* nn.LayerNorm(C) over rows of [M][C] (attention.py:468-470).  mean/rstd: [M] fp32. */
int nk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int M, int C, float eps, void* stream);
long nk_layernorm_ws_floats(int M, int C);
int nk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     const void* dx_add, void* dx, float* dgamma, float* dbeta, float* ws, int M, int C, int accumulate, void* stream);
/* The one-pass form in its two halves (round 5: what the transformer blocks use): _rows writes dx and nk_layernorm_part_rows(M) partial rows
 * [rows][2][C] (sum dy * xhat | sum dy over the rows each workgroup walked) into `part`; nk_colpart_reduce_batch sums the partial rows of up to
 * NK_COLPART_MAX such launches into their dgamma / dbeta in ONE launch, on any stream, any time later (the three LayerNorms of a
 * BasicTransformerBlock, attention.py:487-511, are reduced behind the block's batched weight gradients). */
#define NK_COLPART_MAX 32
typedef struct NkColpartBatch {
  const float* part[NK_COLPART_MAX];
  float* dgamma[NK_COLPART_MAX];
  float* dbeta[NK_COLPART_MAX];
  int nrows[NK_COLPART_MAX];
  int C[NK_COLPART_MAX];
  int accumulate[NK_COLPART_MAX];     /* 0: overwrite dgamma / dbeta, 1: add to them */
  int n;
} NkColpartBatch;
long nk_layernorm_part_rows(int M);
int nk_layernorm_bwd_rows(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                          const void* dx_add, void* dx, float* part, int M, int C, void* stream);
int nk_colpart_reduce_batch(const NkColpartBatch* b, void* stream);
/* The same in two parts, so the caller can run the parameter gradients (off the critical path of backward) on another
 * stream: _dx needs no workspace; _params reads dy, x and the saved statistics only. */
int nk_layernorm_bwd_dx(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* dx_add, void* dx, int M, int C, void* stream);
int nk_layernorm_bwd_params(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma,
                            float* dbeta, float* ws, int M, int C, int accumulate, void* stream);

/* GEGLU (attention.py:55-57): y[M][I] = u[:, :I] * gelu_erf(u[:, I:]) */
int nk_geglu_fwd(const void* u, void* y, long M, int I, void* stream);
int nk_geglu_bwd(const void* dy, const void* u, void* du, long M, int I, void* stream);
/* ... and in the saved-derivative form: y = a * gelu(g), s[M][2I] = [gelu(g) | a * gelu'(g)] from u = [a | g] (s may alias u); du = [dy * s1 | dy * s2] */
int nk_geglu_fwd_s(const void* u, void* y, void* s, long M, int I, void* stream);
int nk_geglu_bwd_s(const void* dy, const void* s, void* du, long M, int I, void* stream);

/* nn.SiLU on a flat bf16 array (openaimodel.py:274,588,615) */
int nk_silu_fwd(const void* x, void* y, long n, void* stream);
int nk_silu_bwd(const void* dy, const void* x, void* dx, long n, void* stream);

/* out = a + b on flat bf16 arrays: gradient join where one tensor feeds two consumers (skip connections,
 * openaimodel.py:832-836) */
int nk_add(const void* a, const void* b, void* out, long n, void* stream);

/* torch.cat([h, skip], dim=1) on channels-last rows and its backward (openaimodel.py:836) */
int nk_cat_channels(const void* a, const void* b, void* out, long rows, int Ca, int Cb, void* stream);
int nk_split_channels(const void* src, void* a, void* b, long rows, int Ca, int Cb, void* stream);

/* backward of F.interpolate(scale_factor=2, mode="nearest") (openaimodel.py:140): 2x2 sum-pool */
int nk_upsample2x_bwd(const void* dup, void* dx, int N, int H, int W, int C, void* stream);

/* boundary layout/dtype conversion: NCHW (fp32 or bf16) <-> channels-last bf16 with channels padded to Cpad */
int nk_nchw_to_nhwc(const void* src, int src_is_f32, void* dst, int N, int C, int HW, int Cpad, float scale,
                    void* stream);
int nk_nhwc_to_nchw(const void* src, void* dst, int dst_is_f32, int N, int C, int HW, int Cpad, void* stream);
int nk_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream);
int nk_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream);

/* bias gradient: out[N] (+)= sum over rows of dy[M][N] (row stride ld) */
long nk_colsum_ws_floats(long M, int N);
int nk_colsum(const void* dy, float* out, float* ws, long M, int N, long ld, int accumulate, void* stream);

/* backward of nk_softmax_rows, in place on dp [M][L] bf16 given the probabilities p: ds = p * (dp - sum_j dp*p) * scale
 * (AttnBlock of the VAE, modules/diffusion/model.py:144-222, when the autoencoder itself is trained: SURVEY 8(f) N2) */
int nk_softmax_rows_bwd(const void* p, void* dp, long M, int L, float scale, void* stream);

/* PatchGAN discriminator pieces (modules/losses/patchgan/model.py:21-95).  Tokens x[M][C] bf16, M = N*H*W.
 * LeakyReLU: y = x >= 0 ? x : slope*x; the backward takes the OUTPUT y (same sign as x).
 * BatchNorm2d in training mode with the following LeakyReLU fused (slope = 1: none): batch mean / biased variance per
 * channel over all M rows, y = act((x - mean) * rstd * gamma + beta); running_mean / running_var (optional) updated with
 * `momentum` and the unbiased variance, as nn.BatchNorm2d does.  backward: dgamma, dbeta (+= when accumulate) and dx.
 * ws: nk_batchnorm_ws_floats(M, C) fp32 elements of scratch. */
int nk_leaky_relu_fwd(const void* x, void* y, long n, float slope, void* stream);
int nk_leaky_relu_bwd(const void* dy, const void* y, void* dx, long n, float slope, void* stream);
long nk_batchnorm_ws_floats(long M, int C);
int nk_batchnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean,
                     float* running_var, float* ws, long M, int C, float eps, float momentum, float slope, void* stream);
int nk_batchnorm_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* dgamma, float* dbeta, float* ws, long M, int C, float slope, int accumulate, void* stream);
/* BatchNorm2d in evaluation mode (+ LeakyReLU): y = act((x - running_mean) * rsqrt(running_var + eps) * gamma + beta); the discriminator
 * as log_images / validation see it (nn.BatchNorm2d.eval() inside NLayerDiscriminator, patchgan/model.py:53-83).  ws: C fp32 elements. */
int nk_batchnorm_eval(const void* x, const float* gamma, const float* beta, const float* running_mean, const float* running_var, void* y,
                      float* ws, long M, int C, float eps, float slope, void* stream);

/* LPIPS pieces (modules/losses/perceptual.py:64-228 over the AlexNet or VGG16 trunk of extractors.py:11-30).
 * ReLU = nk_leaky_relu_* with slope 0.
 * maxpool2x2: tokens [N][H][W][C] -> [N][H/2][W/2][C]; the backward routes each window's gradient to its first maximum.
 * maxpool: k x k windows at stride s, no padding, floor mode (nn.MaxPool2d(k, s); AlexNet's 3 x 3 / 2): [N][H][W][C] ->
 * [N][(H-k)/s+1][(W-k)/s+1][C]; overlapping windows' gradients add up in the backward.
 * lpips_layer: out[n] (+)= mean_p sum_c w[c] (f0/(|f0|+eps) - f1/(|f1|+eps))^2 over a layer's features [N][HW][C] (normalize_tensor,
 * NetLinLayer, spatial_average: perceptual.py:215-224,198-212); the backward returns upstream[n] * d out[n] / d f1. */
int nk_maxpool2x2_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream);
int nk_maxpool2x2_bwd(const void* dy, const void* x, void* dx, int N, int H, int W, int C, void* stream);
int nk_maxpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int s, void* stream);
int nk_maxpool_bwd(const void* dy, const void* x, void* dx, int N, int H, int W, int C, int k, int s, void* stream);
long nk_lpips_layer_ws_floats(int N, int HW);
int nk_lpips_layer_fwd(const void* f0, const void* f1, const float* w, float* out, float* ws, int N, int HW, int C, float eps,
                       int accumulate, void* stream);
int nk_lpips_layer_bwd(const void* f0, const void* f1, const float* w, const float* upstream, void* df1, int N, int HW, int C,
                       float eps, void* stream);

/* y = gelu(x) elementwise over n bf16 values (n % 8 == 0).  mode 0: exact, 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU in open_clip's
 * text tower, models/text_encoder/clip.py:333-343); mode 1: "quick_gelu" x * sigmoid(1.702 x) (HF CLIPTextModel of
 * openai/clip-vit-large-patch14, models/text_encoder/clip.py:49-56). */
int nk_gelu_fwd(const void* x, void* y, long n, int mode, void* stream);

/* timestep_embedding (modules/diffusion/util.py:152-177): out[B][dim] bf16 = [cos | sin](t * freq) */
int nk_timestep_embedding(const float* t, void* out, int B, int dim, float max_period, void* stream);

/* StandardDiffusionLoss "edm" branch + Denoiser scaling (modules/diffusion/loss.py:117-157,
 * modules/diffusion/denoiser.py:41-53, modules/losses/functions.py:91-94).
 * prepare: z_t = x + sigma*eps (fp32 NCHW); net_in = bf16 channels-last z_t*c_in, channels padded to Cpad.
 * loss:    D = net_out*c_out + z_t*c_skip; loss[b] = w[b]*mean((D-target)^2);
 *          dnet (optional) = upstream * dloss[b]/dnet_out, bf16 channels-last padded. */
int nk_edm_prepare(const float* x, const float* eps, const float* sigma, const float* c_in, float* zt, void* net_in,
                   int B, int C, int HW, int Cpad, void* stream);
int nk_edm_loss(const void* net_out, const float* zt, const float* target, const float* c_out, const float* c_skip,
                const float* w, float* loss, void* dnet, int B, int C, int HW, int Cpad, float upstream, void* stream);

/* Sampler-side latent kernels.  x / denoised / x_next: fp32 NCHW [B][C][HW]; net_in / net_out: bf16
 * channels-last tokens [rep*B][HW][Cpad], rep = 2 for classifier-free guidance laid out [uncond | cond]
 * (VanillaCFG.prepare_inputs, modules/guidance.py:26-37).
 * prepare:    net_in[r*B+b] = bf16(c_in[b] * x[b]) for r < rep        (Denoiser.forward input scaling, denoiser.py:41-49,
 *             + the guider's torch.cat([x] * 2))
 * denoise:    D = c_skip[b]*x + c_out[b]*(F_u + scale*(F_c - F_u))    (denoiser.py:49-53 + VanillaCFG.__call__ guidance.py:21-24;
 *             rep == 1: F = net_out, scale ignored)
 * euler_step: d = (x - D)/sigma_hat[b]; x_next = x + (sigma_next[b] - sigma_hat[b])*d   (EDMSampler.sampler_step with the
 *             Euler correction, sampling/sampling.py:166-181,313-316, to_d sampling/utils.py:49-51).  x_next may alias x;
 *             `denoised` is optional (NULL to skip). */
int nk_sample_prepare(const float* x, const float* c_in, void* net_in, int B, int C, int HW, int Cpad, int rep, void* stream);
int nk_sample_denoise(const void* net_out, const float* x, const float* c_skip, const float* c_out, float scale,
                      float* denoised, int B, int C, int HW, int Cpad, int rep, void* stream);
int nk_sample_euler_step(const void* net_out, const float* x, const float* c_skip, const float* c_out, const float* sigma_hat,
                         const float* sigma_next, float scale, float* x_next, float* denoised, int B, int C, int HW, int Cpad,
                         int rep, void* stream);

/* Fused AdamW over the flat fp32 parameter buffer; also rewrites the bf16 shadow the kernels read.
 * (The optimizer itself is outside SURVEY section 8(a); bench.py needs a real parameter update in the timed step.) */
int nk_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor Adafactor on the flat buffers.  Replaces Adafactor.step,
 * reference optimizers/adafactor.py:162-255 (_get_lr :133-147, _rms :150-151, _approx_sq_grad :154-159): the per-tensor
 * Python loop becomes three launches per chunk of consecutive tensors (statistics, update RMS of the matrices, apply; the per-strip and
 * per-tensor finalisations ride in the last block of each pass to finish).  `tensors` / `items` are device tables built by the
 * host (layout: neurosis_amd/csrc/optim.hip NkAfTensor / NkAfItem; nk_adafactor_tensor_bytes() guards the mirror).
 * nk_adafactor_init fills the per-item partial sums of p^2 once; nk_adafactor_chunk performs one step for tensors
 * [tensor_lo, tensor_hi) = items [item_lo, item_hi).  beta2t = 1 - step^decay_rate and rel_step are host scalars. */
typedef struct NkAdafactorArgs {
  float* master; const float* grad; void* shadow; float* state; float* ws;
  const void* tensors; const void* items;
  float* u2_part; float* p2_part; float* mean_row; float* scale; float* lr_t;
  int item_lo, item_hi, tensor_lo, tensor_hi;
  float beta2t, eps1, eps2, clip_threshold, rel_step, weight_decay, grad_scale;
  int scale_parameter;
  unsigned* counters;   /* "blocks done" counters, one range per tensor (NkAfTensor.cnt0); all zero before a step, all zero after it */
  int has_matrix;       /* the chunk holds at least one 2-D tensor (the update-RMS pass of the matrices is launched) */
  int reserved;
} NkAdafactorArgs;
long nk_adafactor_tensor_bytes(void);
int nk_adafactor_init(const NkAdafactorArgs* args, void* stream);
int nk_adafactor_chunk(const NkAdafactorArgs* args, void* stream);

/* LitEma.forward (reference modules/ema.py:40-59) as one pass over the flat fp32 buffers (n % 4 == 0):
 * ema[i] -= one_minus_decay * (ema[i] - p[i]).  The decay schedule min(decay, (1+n)/(10+n)) is the host's. */
int nk_ema_flat(float* ema, const float* p, long n, float one_minus_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
