// Internal parameter block of the MFMA tile engine (gemm.hip).  Not part of the C-ABI.
#pragma once
#include "nk_common.h"

// Geometry of an implicit-GEMM gather over an NHWC tensor [NB][H][W][C].
// A "pixel" index decomposes over the grid [NB][Ho][Wo]; a "tap" over [KH][KW].
//   hnum = ph*rs + kh*ks + off_h ;  h = hnum / div ; valid iff 0 <= hnum, (need_even -> hnum even), h < H
struct NkGather {
  int H, W, C;
  int Ho, Wo;
  int KW;
  int rs, ks;
  int off_h, off_w;
  int div;
  int need_even;
  FastDiv fWo, fHoWo, fC, fKW;
};

// conv-dgrad weight operand: k = tap*Cout + co  ->  W[co*co_stride + tap*tap_stride + ci]
struct NkTapW {
  long co_stride, tap_stride;
  FastDiv fCout;
};

struct NkGemmParams {
  const bf16_t* A;
  const bf16_t* B;
  long lda, ldb;
  int M, N, K;
  NkGather ga, gb;
  NkTapW tw;
  void* C;
  long ldc;
  float alpha;
  const float* bias;        // [N] fp32, optional
  const bf16_t* residual;   // [M][ldr] bf16, optional
  long ldr;
  const bf16_t* rowvec;     // [batch][ld_rowvec] bf16 broadcast over the rows of one batch item, optional
  long ld_rowvec;
  FastDiv fRowsPerBatch;
  int ksplit_len;
  int group_m;              // rows of the XCD-local tile patch (nk_gemm_dma_kernel / nk_gemm_ring_kernel), set by the launcher
  int accumulate;           // fp32 output: 0 = store, 1 = atomic add
  int lean_src;             // producer-wave two-group kernel: dense operands through LeanSrcG2 (gemm_g2.h), set by the launcher
  int k_rotate;             // two-group kernels: XCD x starts its k loop x / 8 of the way through K and wraps (OpG2::rotate), set by the launcher
  // batched launch: blockIdx.z selects one of nbatch (<= NK_MAX_BATCH) problems of identical shape
  int nbatch;
  const bf16_t* Ab[8];
  const bf16_t* Bb[8];
  void* Cb[8];
  // stream-K (nk_gemm_sk_kernel): per-stream workspace owned by the dispatcher
  unsigned* sk_counter;     // (unused)
  unsigned sk_base;         // (unused)
  unsigned* sk_flags;       // [grid] flag[ticket] = 1 while that workgroup's partial tile waits in sk_ws; its one reader lowers it
  unsigned sk_epoch;        // (unused: no per-launch state, so a launch replayed from a hipGraph is a fresh one)
  float* sk_ws;             // [grid][128*128] fp32 partial tiles in accumulator-register order
  int sk_chunked;           // 1: each XCD owns a contiguous eighth of the tile list
  int sk_debug;             // NK_SK_DEBUG=2 (fault injection): every fix-up wait gives up at once -- tests the fail-closed path
  unsigned* sk_health;      // backward-health word (errors.hip): raised when a fix-up wait gives up
  // weight-gradient launches (A = dy, r-contiguous): the bias gradient dbias[m] (+)= sum_k A(m, k), accumulated by the first column tile of
  // every row block with one extra MFMA per row block and k sub-step against a fragment of ones (the dy panel is already in registers)
  float* dbias;
  float* dbias_b[8];        // batched launch: one per problem
  // GEGLU backward fused into the epilogue of the feed-forward-out input gradient (the LDS-staged epilogue of the 128 x 128 kernels):
  // the GEMM's result is d = dL/d(a * gelu(g)) [M][N]; written instead: C[m][n] = d * gelu(g), C[m][N + n] = d * a * gelu'(g), with
  // a = geglu_u[m][n], g = geglu_u[m][N + n] (ld_u elements per row) -- i.e. C is dL/du [M][2N]
  const bf16_t* geglu_u;
  long ld_u;
  // GEGLU FORWARD fused into the FeedForward projection (the 256 x 256 two-group kernel only): C = u [M][N = 2 I] as always, and
  // geglu_h[m][j] = a * gelu(g), a = u[m][j], g = u[m][I + j] (ld_h elements per row).  The kernel pairs the two halves by reading the
  // weight rows of a column tile as 16 a-rows, 16 gate-rows, 16 a-rows, ...: a lane then holds a and g of the same j.
  bf16_t* geglu_h;
  long ld_h;
  // round 6: 1 = the saved tensor is s = [gelu(g) | a * gelu'(g)] instead of u = [a | g] -- written by the fused forward INSTEAD of u, read by
  // the fused input gradient as du = [d * s1 | d * s2]: the erf-GELU derivative is evaluated in the forward's epilogue, which has the cdf and
  // the exponential already
  int geglu_save;
  // halo-tile 3 x 3 convolution (conv_halo.h)
  int halo_nb;              // images in the batch (0: not a convolution the halo kernel may take)
  float* stats_part;        // statistics epilogue: [halo_nb][pixel tiles per image][2 * stats_groups] partial sums of the OUTPUT
  int stats_groups;
};
#define NK_MAX_BATCH 8

enum { NK_OP_KC = 0, NK_OP_KCG = 1, NK_OP_MC = 2, NK_OP_MCT = 3, NK_OP_MCG = 4 };

int nk_gemm_dispatch(NkGemmParams& p, int amode, int bmode, int out_f32, int allow_splitk, hipStream_t stream);
int nk_halo_tiles_per_image(const NkGemmParams& p);
int nk_halo_bn(int N);        // column-tile width of the halo-tile launch (conv_halo.h: halo_bn)
int nk_geglu_fwd_fusable(const NkGemmParams& p);
