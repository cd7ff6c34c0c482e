// C-ABI launchers for the contraction family: nn.Linear and nn.Conv2d forward / dgrad / wgrad.
// Each maps its problem onto the MFMA tile engine (gemm.hip) by choosing operand modes.
#include "../../include/neurosis_hip.h"
#include "nk_gemm.h"
#include <string.h>

static NkGemmParams zero_params() {
  NkGemmParams p;
  memset(&p, 0, sizeof(p));
  p.alpha = 1.0f;
  p.fRowsPerBatch = make_fastdiv(1);
  p.ga.fWo = p.ga.fHoWo = p.ga.fC = p.ga.fKW = make_fastdiv(1);
  p.gb.fWo = p.gb.fHoWo = p.gb.fC = p.gb.fKW = make_fastdiv(1);
  p.tw.fCout = make_fastdiv(1);
  return p;
}

extern "C" int nk_linear_fwd(const void* x, const void* w, const float* bias, const void* residual, void* y,
                             int M, int N, int K, long ldx, long ldw, long ldr, long ldy, float alpha,
                             void* stream) {
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)x; p.lda = ldx;
  p.B = (const bf16_t*)w; p.ldb = ldw;
  p.M = M; p.N = N; p.K = K;
  p.C = y; p.ldc = ldy;
  p.bias = bias;
  p.residual = (const bf16_t*)residual; p.ldr = ldr;
  p.alpha = alpha;
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

extern "C" int nk_linear_fwd_batched(const void* const* x, const void* const* w, void* const* y, int count, int M, int N, int K,
                                     long ldx, long ldw, long ldy, void* stream) {
  // `count` (<= 8) bias-free projections of IDENTICAL shape in one launch (blockIdx.z): y[i][M,N] = x[i][M,K] @ w[i][N,K]^T.
  // Host pointer arrays; the device pointers are copied into the kernel arguments.
  NK_CHECK_ARG(x && w && y && count >= 1 && count <= NK_MAX_BATCH);
  NkGemmParams p = zero_params();
  p.lda = ldx; p.ldb = ldw;
  p.M = M; p.N = N; p.K = K;
  p.ldc = ldy;
  p.nbatch = count;
  for (int i = 0; i < count; ++i) {
    p.Ab[i] = (const bf16_t*)x[i];
    p.Bb[i] = (const bf16_t*)w[i];
    p.Cb[i] = y[i];
  }
  p.A = p.Ab[0]; p.B = p.Bb[0]; p.C = p.Cb[0];
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

extern "C" int nk_linear_dgrad(const void* dy, const void* w, const void* dx_add, void* dx, int M, int N, int K,
                               long lddy, long ldw, long ldadd, long lddx, void* stream) {
  // dx[M,K] = dy[M,N] @ w[N,K]  : reduction over N; w is r-contiguous (r = K index)
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)dy; p.lda = lddy;
  p.B = (const bf16_t*)w; p.ldb = ldw;
  p.M = M; p.N = K; p.K = N;
  p.C = dx; p.ldc = lddx;
  p.residual = (const bf16_t*)dx_add; p.ldr = ldadd;
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_MC, 0, 0, (hipStream_t)stream);
}

static NkGemmParams geglu_fwd_params(const void* x, const void* w, const float* bias, void* u, void* h, int M, int I, int K, long ldx, long ldw,
                                      long ldu, long ldh) {
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)x; p.lda = ldx;
  p.B = (const bf16_t*)w; p.ldb = ldw;
  p.M = M; p.N = 2 * I; p.K = K;
  p.C = u; p.ldc = ldu;
  p.bias = bias;
  p.geglu_h = (bf16_t*)h; p.ld_h = ldh;
  return p;
}
extern "C" long nk_linear_fwd_geglu_ok(int M, int I, int K) {
  // 1 when nk_linear_fwd_geglu takes this shape (else: nk_linear_fwd followed by nk_geglu_fwd)
  if (M <= 0 || I <= 0 || K <= 0 || (I & 127) || (K & 7)) return 0;
  NkGemmParams p = geglu_fwd_params(nullptr, nullptr, nullptr, nullptr, (void*)16, M, I, K, K, K, 2l * I, I);
  return nk_geglu_fwd_fusable(p);
}
extern "C" int nk_linear_fwd_geglu(const void* x, const void* w, const float* bias, void* u, void* h, int M, int I, int K, long ldx, long ldw,
                                   long ldu, long ldh, void* stream) {
  // FeedForward.net[0] (GEGLU, modules/attention.py:50-57) in one launch: u[M, 2I] = x[M, K] @ w[2I, K]^T + bias (kept for the backward) and
  // h[M, I] = u[:, :I] * gelu(u[:, I:]) from the same accumulators -- the stand-alone GEGLU forward kernel and its read of u are gone
  NK_CHECK_ARG(x && w && u && h && M > 0 && I > 0 && K > 0);
  NkGemmParams p = geglu_fwd_params(x, w, bias, u, h, M, I, K, ldx, ldw, ldu, ldh);
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

extern "C" int nk_linear_fwd_geglu_s(const void* x, const void* w, const float* bias, void* s, void* h, int M, int I, int K, long ldx, long ldw,
                                     long lds, long ldh, void* stream) {
  // nk_linear_fwd_geglu with the SAVED-DERIVATIVE form of the projection output: with u = x w^T + bias = [a | g] (never written),
  // s[M, 2I] = [gelu(g) | a gelu'(g)] and h = a gelu(g) -- what nk_linear_dgrad_geglu_s needs is then two products per element
  NK_CHECK_ARG(x && w && s && h && M > 0 && I > 0 && K > 0);
  NkGemmParams p = geglu_fwd_params(x, w, bias, s, h, M, I, K, ldx, ldw, lds, ldh);
  p.geglu_save = 1;
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

static int dgrad_geglu(const void* dy, const void* w, const void* u, void* du, int M, int N, int I, long lddy, long ldw, long ldu, long lddu, int save,
                       void* stream) {
  NK_CHECK_ARG(dy && w && u && du && I > 0 && (I & 7) == 0);
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)dy; p.lda = lddy;
  p.B = (const bf16_t*)w; p.ldb = ldw;
  p.M = M; p.N = I; p.K = N;
  p.C = du; p.ldc = lddu;
  p.geglu_u = (const bf16_t*)u; p.ld_u = ldu;
  p.geglu_save = save;
  return nk_gemm_dispatch(p, NK_OP_KC, NK_OP_MC, 0, 0, (hipStream_t)stream);
}
extern "C" int nk_linear_dgrad_geglu_s(const void* dy, const void* w, const void* s, void* du, int M, int N, int I, long lddy, long ldw,
                                       long lds, long lddu, void* stream) {
  // nk_linear_dgrad_geglu on the saved-derivative tensor of nk_linear_fwd_geglu_s / nk_geglu_fwd_s: d = dy[M,N] @ w[N,I],
  // du[M, 2I] = [d * s[:, :I] | d * s[:, I:]]
  return dgrad_geglu(dy, w, s, du, M, N, I, lddy, ldw, lds, lddu, 1, stream);
}

extern "C" int nk_linear_dgrad_geglu(const void* dy, const void* w, const void* u, void* du, int M, int N, int I, long lddy, long ldw,
                                     long ldu, long lddu, void* stream) {
  // FeedForward backward through net[2] and the GEGLU in one launch (modules/attention.py:60-74): d = dy[M,N] @ w[N,I] is the gradient of
  // a * gelu(g); the epilogue turns it into du[M, 2I] = [d * gelu(g) | d * a * gelu'(g)] with u = [a | g] [M, 2I] from the forward --
  // the stand-alone GEGLU backward kernel and its read of d are gone
  return dgrad_geglu(dy, w, u, du, M, N, I, lddy, ldw, ldu, lddu, 0, stream);
}

static int linear_wgrad(const void* dy, const void* x, float* dw, float* dbias, int M, int N, int K, long lddy, long ldx, long lddw,
                        int accumulate, void* stream) {
  // dw[N,K] (+)= dy[M,N]^T @ x[M,K] : reduction over M; both operands r-contiguous.  dbias[N] (+)= column sums of dy, in the same launch
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)dy; p.lda = lddy;
  p.B = (const bf16_t*)x; p.ldb = ldx;
  p.M = N; p.N = K; p.K = M;
  p.C = dw; p.ldc = lddw;
  p.accumulate = accumulate;
  p.dbias = dbias;
  return nk_gemm_dispatch(p, NK_OP_MC, NK_OP_MC, 1, lddw == K, (hipStream_t)stream);
}
extern "C" int nk_linear_wgrad(const void* dy, const void* x, float* dw, int M, int N, int K, long lddy,
                               long ldx, long lddw, int accumulate, void* stream) {
  return linear_wgrad(dy, x, dw, nullptr, M, N, K, lddy, ldx, lddw, accumulate, stream);
}
extern "C" int nk_linear_wgrad_bias(const void* dy, const void* x, float* dw, float* dbias, int M, int N, int K, long lddy,
                                    long ldx, long lddw, int accumulate, void* stream) {
  NK_CHECK_ARG(dbias != nullptr);
  return linear_wgrad(dy, x, dw, dbias, M, N, K, lddy, ldx, lddw, accumulate, stream);
}

extern "C" int nk_linear_wgrad_batched(const void* const* dy, const void* const* x, float* const* dw, float* const* dbias, int count, int M,
                                       int N, int K, long lddy, long ldx, long lddw, int accumulate, void* stream) {
  // `count` weight gradients of IDENTICAL shape in one launch (blockIdx.z): dw[i][N,K] (+)= dy[i][M,N]^T @ x[i][M,K].
  // Host pointer arrays; the device pointers are copied into the kernel arguments.  dbias: NULL, or one pointer per problem (NULL entries
  // allowed: layers without a bias) for the bias gradients dbias[i][N] (+)= column sums of dy[i].
  NK_CHECK_ARG(dy && x && dw && count >= 1 && count <= NK_MAX_BATCH);
  NkGemmParams p = zero_params();
  p.lda = lddy; p.ldb = ldx;
  p.M = N; p.N = K; p.K = M;
  p.ldc = lddw;
  p.accumulate = accumulate;
  p.nbatch = count;
  for (int i = 0; i < count; ++i) {
    p.Ab[i] = (const bf16_t*)dy[i];
    p.Bb[i] = (const bf16_t*)x[i];
    p.Cb[i] = dw[i];
    p.dbias_b[i] = dbias ? dbias[i] : nullptr;
  }
  p.A = p.Ab[0]; p.B = p.Bb[0]; p.C = p.Cb[0];
  return nk_gemm_dispatch(p, NK_OP_MC, NK_OP_MC, 1, lddw == K, (hipStream_t)stream);
}

static int check_conv(const NkConvDesc* d) {
  NK_CHECK_ARG(d != nullptr);
  NK_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0);
  NK_CHECK_ARG(d->KH > 0 && d->KW > 0 && (d->stride == 1 || d->stride == 2 || d->stride == 4));
  NK_CHECK_ARG((d->Cin & 7) == 0 && (d->Cout & 7) == 0);
  NK_CHECK_ARG(!(d->upsample && d->stride != 1));
  const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
  NK_CHECK_ARG(d->Ho > 0 && d->Wo > 0);
  // the last output pixel's first tap must start inside the (left/top padded) input
  NK_CHECK_ARG((d->Ho - 1) * d->stride - d->pad_t < Hin && (d->Wo - 1) * d->stride - d->pad_l < Win);
  NK_CHECK_ARG((long)d->N * Hin * Win * (long)d->Cin < (1l << 31) && (long)d->N * d->Ho * d->Wo * (long)d->Cout < (1l << 31));
  return NK_OK;
}

// gather of the conv INPUT x[N,H,W,Cin] indexed by output pixels (forward and wgrad)
static NkGather fwd_gather(const NkConvDesc* d) {
  NkGather g;
  memset(&g, 0, sizeof(g));
  g.H = d->H; g.W = d->W; g.C = d->Cin;
  g.Ho = d->Ho; g.Wo = d->Wo; g.KW = d->KW;
  g.rs = d->stride; g.ks = 1;
  g.off_h = -d->pad_t; g.off_w = -d->pad_l;
  g.div = d->upsample ? 2 : 1;
  g.need_even = 0;
  // upsample: hnum runs over the virtual 2x grid; hnum>>1 < H is the same bound as hnum < 2H
  g.fWo = make_fastdiv(d->Wo); g.fHoWo = make_fastdiv(d->Ho * d->Wo);
  g.fC = make_fastdiv(d->Cin); g.fKW = make_fastdiv(d->KW);
  return g;
}

static NkGemmParams conv_fwd_params(const NkConvDesc* d, const void* x, const void* w, const float* bias, const void* rowvec,
                                    const void* residual, void* y) {
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)x;
  p.ga = fwd_gather(d);
  p.B = (const bf16_t*)w; p.ldb = (long)d->KH * d->KW * d->Cin;
  p.M = d->N * d->Ho * d->Wo; p.N = d->Cout; p.K = d->KH * d->KW * d->Cin;
  p.C = y; p.ldc = d->Cout;
  p.bias = bias;
  p.rowvec = (const bf16_t*)rowvec; p.ld_rowvec = d->Cout;
  p.fRowsPerBatch = make_fastdiv(d->Ho * d->Wo);
  p.residual = (const bf16_t*)residual; p.ldr = d->Cout;
  p.halo_nb = (d->KH == 3 && d->KW == 3) ? d->N : 0;       // 3 x 3 / stride 1 / padding 1 shapes may take the halo-tile kernel
  return p;
}

extern "C" int nk_conv2d_fwd(const NkConvDesc* d, const void* x, const void* w, const float* bias,
                             const void* rowvec, const void* residual, void* y, void* stream) {
  if (int e = check_conv(d)) return e;
  NkGemmParams p = conv_fwd_params(d, x, w, bias, rowvec, residual, y);
  return nk_gemm_dispatch(p, NK_OP_KCG, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

// Can this convolution emit the GroupNorm sums of its OUTPUT over `stats_groups` groups from its epilogue (the halo-tile kernel takes the
// shape, and a column tile holds whole groups)?  Returns the number of pixel tiles per image of that launch -- the partials have that many
// rows per image -- or 0: run nk_conv2d_fwd and a statistics pass.  stats_groups = 0 asks whether the halo-tile kernel takes the shape.
extern "C" long nk_conv2d_stats_tiles(const NkConvDesc* d, int stats_groups) {
  if (check_conv(d)) return 0;
  if (stats_groups < 0 || stats_groups > 32) return 0;
  if (stats_groups) {
    if (d->Cout % stats_groups) return 0;
    const int bn = nk_halo_bn(d->Cout), cpg = d->Cout / stats_groups;
    if (bn % cpg) return 0;               // a column tile must hold whole groups
  }
  NkGemmParams p = conv_fwd_params(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return nk_halo_tiles_per_image(p);
}

extern "C" int nk_conv2d_fwd_stats(const NkConvDesc* d, const void* x, const void* w, const float* bias, const void* rowvec,
                                   const void* residual, void* y, float* stats_part, int stats_groups, void* stream) {
  // nk_conv2d_fwd that also writes, per pixel tile, the sums and sums of squares of its (bf16-rounded) output per GroupNorm group:
  // stats_part [N][tiles][2 * stats_groups], tiles = nk_conv2d_stats_tiles(d, stats_groups); nk_groupnorm_sums_from_parts adds them up
  if (int e = check_conv(d)) return e;
  NK_CHECK_ARG(stats_part && stats_groups > 0);
  NK_CHECK_ARG(nk_conv2d_stats_tiles(d, stats_groups) > 0);
  NkGemmParams p = conv_fwd_params(d, x, w, bias, rowvec, residual, y);
  p.stats_part = stats_part; p.stats_groups = stats_groups;
  return nk_gemm_dispatch(p, NK_OP_KCG, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

extern "C" int nk_conv2d_dgrad(const NkConvDesc* d, const void* dy, const void* w, void* dx, void* stream) {
  // dx[n,hi,wi,ci] = sum_{kh,kw,co} dy[n,(hi+pad-kh)/s,(wi+pad-kw)/s,co] * w[co,kh,kw,ci]
  // rows = pixels of the conv input grid (the virtual 2x grid when upsample=1; reduce with
  // nk_upsample2x_bwd afterwards), k = (tap, co)
  if (int e = check_conv(d)) return e;
  const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
  NkGemmParams p = zero_params();
  NkGather g;
  memset(&g, 0, sizeof(g));
  g.H = d->Ho; g.W = d->Wo; g.C = d->Cout;
  g.Ho = Hin; g.Wo = Win; g.KW = d->KW;
  g.rs = 1; g.ks = -1;
  g.off_h = d->pad_t; g.off_w = d->pad_l;
  g.div = d->stride; g.need_even = d->stride - 1;
  g.fWo = make_fastdiv(Win); g.fHoWo = make_fastdiv(Hin * Win);
  g.fC = make_fastdiv(d->Cout); g.fKW = make_fastdiv(d->KW);
  p.A = (const bf16_t*)dy; p.ga = g;
  p.B = (const bf16_t*)w;
  p.tw.co_stride = (long)d->KH * d->KW * d->Cin;
  p.tw.tap_stride = d->Cin;
  p.tw.fCout = make_fastdiv(d->Cout);
  p.M = d->N * Hin * Win; p.N = d->Cin; p.K = d->KH * d->KW * d->Cout;
  p.C = dx; p.ldc = d->Cin;
  return nk_gemm_dispatch(p, NK_OP_KCG, NK_OP_MCT, 0, 0, (hipStream_t)stream);
}

extern "C" long nk_conv2d_dgrad_flipped_ok(const NkConvDesc* d) {
  // 1 when nk_conv2d_dgrad_flipped takes this convolution: 3 x 3 / stride 1 / padding 1 (also behind a fused 2x upsample: the gradient
  // is then taken over the virtual 2x grid) whose transposed problem the halo-tile forward kernel runs
  if (check_conv(d) || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1) return 0;
  const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
  if (d->Ho != Hin || d->Wo != Win) return 0;
  NkConvDesc t = *d;
  t.H = Hin; t.W = Win; t.Cin = d->Cout; t.Cout = d->Cin; t.upsample = 0;
  NkGemmParams p = conv_fwd_params(&t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return nk_halo_tiles_per_image(p) > 0;
}

extern "C" int nk_conv2d_dgrad_flipped(const NkConvDesc* d, const void* dy, const void* wt, void* dx, void* stream) {
  // nk_conv2d_dgrad for stride-1 3 x 3 "same" convolutions as a FORWARD convolution of dy with wt = nk_conv_weight_flip(w)
  // ([Cin][9 taps mirrored][Cout]): dx[n,y,x,ci] = sum_{tap,co} dy[n, y+dy'-1, x+dx'-1, co] * wt[ci][tap'][co] -- the halo-tile kernel.
  NK_CHECK_ARG(nk_conv2d_dgrad_flipped_ok(d) == 1);
  const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
  NkConvDesc t = *d;
  t.H = Hin; t.W = Win; t.Cin = d->Cout; t.Cout = d->Cin; t.upsample = 0;
  NkGemmParams p = conv_fwd_params(&t, dy, wt, nullptr, nullptr, nullptr, dx);
  return nk_gemm_dispatch(p, NK_OP_KCG, NK_OP_KC, 0, 0, (hipStream_t)stream);
}

static int conv_wgrad(const NkConvDesc* d, const void* dy, const void* x, float* dw, float* dbias, int accumulate, void* stream) {
  // dw[co, (tap,ci)] (+)= sum_pix dy[pix, co] * x[gather(pix, tap), ci];  dbias[co] (+)= sum_pix dy[pix, co] in the same launch
  if (int e = check_conv(d)) return e;
  NkGemmParams p = zero_params();
  p.A = (const bf16_t*)dy; p.lda = d->Cout;
  p.B = (const bf16_t*)x; p.gb = fwd_gather(d);
  p.M = d->Cout; p.N = d->KH * d->KW * d->Cin; p.K = d->N * d->Ho * d->Wo;
  p.C = dw; p.ldc = p.N;
  p.accumulate = accumulate;
  p.dbias = dbias;
  p.halo_nb = (d->KH == 3 && d->KW == 3) ? d->N : 0;       // 3 x 3 / stride 1 / padding 1 shapes may take the halo-tile weight-gradient kernel
  return nk_gemm_dispatch(p, NK_OP_MC, NK_OP_MCG, 1, 1, (hipStream_t)stream);
}
extern "C" int nk_conv2d_wgrad(const NkConvDesc* d, const void* dy, const void* x, float* dw, int accumulate,
                               void* stream) {
  return conv_wgrad(d, dy, x, dw, nullptr, accumulate, stream);
}
extern "C" int nk_conv2d_wgrad_bias(const NkConvDesc* d, const void* dy, const void* x, float* dw, float* dbias, int accumulate,
                                    void* stream) {
  NK_CHECK_ARG(dbias != nullptr);
  return conv_wgrad(d, dy, x, dw, dbias, accumulate, stream);
}
