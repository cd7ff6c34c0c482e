// MFMA tile engine for every contraction on the SDXL training-step path (gfx950 / CDNA4).
//
// One templated kernel computes C[M,N] = sum_k A(m,k) * B(n,k) with bf16 operands and fp32
// accumulation on v_mfma_f32_16x16x32_bf16.  What differs between Linear forward / dgrad / wgrad and
// implicit-GEMM Conv2d forward / dgrad / wgrad is only HOW an operand element (r, k) is found in HBM,
// so each operand is described by a "mode":
//
//   OP_KC   k-contiguous dense          elem(r,k) = P[r*ld + k]                (Linear x, W; conv W)
//   OP_KCG  k-contiguous conv gather    r -> pixel (n,ph,pw), k -> (tap,c)      (conv fwd x, conv dgrad dy)
//   OP_MC   r-contiguous dense          elem(r,k) = P[k*ld + r]                (dgrad W, wgrad dy / x)
//   OP_MCT  r-contiguous, k=(tap,co)    elem(ci,k) = W[co*cs + tap*C + ci]     (conv dgrad W)
//   OP_MCG  r-contiguous conv gather    r -> (tap,c), k -> pixel               (conv wgrad x)
//
// k-contiguous operands are staged into an XOR-swizzled [rows][64] LDS image and read with
// ds_read_b128; r-contiguous operands are staged as they lie in memory ([64 k][rows]) and read with
// gfx950's transposing ds_read_b64_tr_b16, so no operand is ever transposed in HBM.
//
// The kernels in this file, all over the same operand loaders, chosen per launch by shape (nk_gemm_dispatch):
//   nk_gemm_dma_kernel    128x128x64, 8 waves, LDS-DMA (global_load_lds) double buffer, two workgroups per CU: the general kernel
//   nk_gemm_ring_kernel   the same tile with a 4-stage ring and counted vmcnt for grids of <= one workgroup per CU
//   nk_gemm_sk_kernel     persistent stream-K over that tile for under-filled bf16-output grids (fix-up through a workspace)
//   nk_gemm_xl_kernel     256x256x64, 16 waves, one workgroup per CU: large conv-forward grids (gathered A)
//   nk_gemm_xl2g_kernel   256x256x64, 8 waves in two groups staggered by a barrier, four phases per k-slab: large Linear forward grids
//   nk_gemm_g2_kernel     (gemm_g2.h) 128x160 / 128x128 two-group staggered ring at one workgroup per CU: the K = 640 / 1280 Linear shapes
//   nk_conv3x3_halo_kernel (conv_halo.h) 3 x 3 / stride-1 convolutions from an LDS halo tile: each input byte staged once per 9 taps
//   (variants that lost their A/B -- the register-staged first version, 256x128 "big", the software-pipelined 256x256, other wave shapes --
//   were removed in round 3; HISTORY.md keeps their measurements)
// All use an XCD-aware block->tile mapping with grouped tile order; bf16 outputs leave through a register-direct
// permlane16_swap epilogue or an LDS-staged one (16 B per lane, row-contiguous), fp32 weight gradients through vector stores/atomics.
//
// Reference call sites this engine serves: K1 conv2d, K3 nn.Linear, and the
// unfused attention products of the VAE mid block (K10).
#include "nk_common.h"
#include "nk_gemm.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256
#define GROUP_M 8
#define KC_IMAGE_BYTES (128 * 128)          // [128 rows][64 k] bf16
#define MC_ROW_BYTES (128 * 2 + 16)         // [64 k][128 r] bf16, rows padded by 16 B
#define MC_IMAGE_BYTES (BK * MC_ROW_BYTES)  // 17408
#define OPND_BYTES MC_IMAGE_BYTES           // per-operand slot (max of the two images)
#define STAGE_BYTES (2 * OPND_BYTES)
#define CS_LD 132                            // fp32 epilogue staging row stride (floats)
#define SMEM_BYTES (2 * STAGE_BYTES)        // 69632 >= 128*132*4 = 67584

static_assert(SMEM_BYTES >= BM * CS_LD * 4, "epilogue staging must fit");

enum { OP_KC = 0, OP_KCG = 1, OP_MC = 2, OP_MCT = 3, OP_MCG = 4 };

__device__ __forceinline__ bool is_kc(int mode) { return mode == OP_KC || mode == OP_KCG; }

// ---------------------------------------------------------------------------------------------
// conv gather geometry: maps (pixel, tap) -> source element offset or "padding"
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long gather_offset(const NkGather& g, int n, int bh, int bw, int kh, int kw,
                                              int c, bool& valid) {
  int hn = bh + kh * g.ks;
  int wn = bw + kw * g.ks;
  bool ok = (hn >= 0) & (wn >= 0);
  ok &= (((hn | wn) & g.need_even) == 0);     // need_even: div - 1 (a transposed gather lands on every div-th position only)
  const int sh = g.div >> 1;                   // div is 1, 2 or 4
  int h = hn >> sh, w = wn >> sh;
  ok &= (h < g.H) & (w < g.W);
  valid = valid && ok;
  return (((long)n * g.H + h) * g.W + w) * g.C + c;
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int OUT_F32, int TBM = BM, int TNT = NTHREADS, int NJ = 4>
__device__ __forceinline__ void nk_gemm_epilogue(const NkGemmParams& p, char* smem, float4_t (&acc)[4][NJ], int m0, int n0, int tid,
                                                 int lane, int wm, int wn) {
  // ---- epilogue: accumulators -> LDS (fp32, [128][CS_LD]) -> coalesced global stores ----
  float* cs = (float*)smem;
  // GEGLU backward (below): the saved u = [a | g] of this thread's chunks is fetched BEFORE the accumulators are staged -- issued inside the
  // store loop, two iterations at a time, the loads' latency (two dependent HBM round trips per tile, nothing else in flight at two
  // workgroups per CU) was a third of the launch
  constexpr int GIT = OUT_F32 ? 1 : (TBM * BN / 8) / TNT;
  uint4_t gu_a[GIT], gu_g[GIT];
  const bool geglu = !OUT_F32 && p.geglu_u != nullptr && (p.N & 7) == 0;      // wave-uniform
  if constexpr (!OUT_F32) {
    if (geglu) {
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int idx = tid + it * TNT;
        const int m = m0 + (idx >> 4), n = n0 + (idx & 15) * 8;
        const bool ok = m < p.M && n < p.N;
        const bf16_t* up = p.geglu_u + (long)(ok ? m : 0) * p.ld_u + (ok ? n : 0);
        gu_a[it] = *(const uint4_t*)up;
        gu_g[it] = *(const uint4_t*)(up + p.N);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        int col = wn * (NJ * 16) + j * 16 + (lane & 15);
        cs[row * CS_LD + col] = acc[i][j][r];
      }
  __syncthreads();

  if constexpr (!OUT_F32) {
    if (geglu) {
      // GEGLU backward (modules/attention.py:55-57, y = a * gelu(g)): this tile holds d = dL/dy; u = [a | g] was saved by the forward
      bf16_t* C = (bf16_t*)p.C;
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int idx = tid + it * TNT;
        const int row = idx >> 4, cc = idx & 15;
        const int m = m0 + row, n = n0 + cc * 8;
        if (m >= p.M || n >= p.N) continue;
        const float4_t c0 = *(const float4_t*)(cs + row * CS_LD + cc * 8);
        const float4_t c1 = *(const float4_t*)(cs + row * CS_LD + cc * 8 + 4);
        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        if (p.alpha != 1.0f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        }
        float a[8], g[8], da[8], dg[8];
        unpack8(gu_a[it], a);
        unpack8(gu_g[it], g);
        if (p.geglu_save) {      // the forward saved s = [gelu(g) | a gelu'(g)]: two products (wave-uniform branch)
#pragma unroll
          for (int e = 0; e < 8; ++e) { da[e] = v[e] * a[e]; dg[e] = v[e] * g[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float cdf, pdf;
            normal_cdf_pdf(g[e], cdf, pdf);
            da[e] = v[e] * (g[e] * cdf);
            dg[e] = v[e] * a[e] * (cdf + g[e] * pdf);
          }
        }
        *(uint4_t*)(C + (long)m * p.ldc + n) = pack8(da);
        *(uint4_t*)(C + (long)m * p.ldc + p.N + n) = pack8(dg);
      }
      return;
    }
  }

  if constexpr (OUT_F32) {
    float* C = (float*)(p.nbatch ? p.Cb[blockIdx.z] : p.C);
    if (!p.accumulate && (p.N & 3) == 0 && (p.ldc & 3) == 0) {
      // plain stores: 16 B per lane, a wave covers two 256-byte row segments per instruction
      for (int it = 0; it < (TBM * BN / 4) / TNT; ++it) {
        int idx = tid + it * TNT;
        int row = idx >> 5, c4 = (idx & 31) * 4;
        int m = m0 + row, n = n0 + c4;
        if (m < p.M && n < p.N) {
          float4_t v = *(const float4_t*)(cs + row * CS_LD + c4);
          v *= p.alpha;
          *(float4_t*)(C + (long)m * p.ldc + n) = v;
        }
      }
      return;
    }
    for (int it = 0; it < (TBM * BN) / TNT; ++it) {
      int idx = tid + it * TNT;
      int row = idx >> 7, col = idx & 127;
      int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) {
        float v = cs[row * CS_LD + col] * p.alpha;
        float* dst = C + (long)m * p.ldc + n;
        if (p.accumulate) unsafeAtomicAdd(dst, v);
        else *dst = v;
      }
    }
  } else {
    bf16_t* C = (bf16_t*)(p.nbatch ? p.Cb[blockIdx.z] : p.C);
    const bool vec = (p.N & 7) == 0;
#pragma unroll 2
    for (int it = 0; it < (TBM * BN / 8) / TNT; ++it) {
      int idx = tid + it * TNT;
      int row = idx >> 4, cc = idx & 15;
      int m = m0 + row, n = n0 + cc * 8;
      if (m >= p.M || n >= p.N) continue;
      float v[8];
      const float4_t c0 = *(const float4_t*)(cs + row * CS_LD + cc * 8);
      const float4_t c1 = *(const float4_t*)(cs + row * CS_LD + cc * 8 + 4);
      v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3];
      v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
      if (p.alpha != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      }
      if (vec) {
        if (p.bias) {
          const float4_t b0 = *(const float4_t*)(p.bias + n);
          const float4_t b1 = *(const float4_t*)(p.bias + n + 4);
          v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
          v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
        }
        if (p.rowvec) {
          unsigned b = fdiv((unsigned)m, p.fRowsPerBatch);
          float t[8];
          unpack8(*(const uint4_t*)(p.rowvec + (long)b * p.ld_rowvec + n), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (p.residual) {
          float t[8];
          unpack8(*(const uint4_t*)(p.residual + (long)m * p.ldr + n), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        *(uint4_t*)(C + (long)m * p.ldc + n) = pack8(v);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) {
          float x = v[e];
          if (p.bias) x += p.bias[n + e];
          if (p.rowvec) x += bf2f(p.rowvec[(long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n + e]);
          if (p.residual) x += bf2f(p.residual[(long)m * p.ldr + n + e]);
          C[(long)m * p.ldc + n + e] = f2bf(x);
        }
      }
    }
  }
}

// =============================================================================================
// v2 main loop: LDS-DMA staging (global_load_lds_dwordx4).  Register staging costs a ds_write_b128 per 16 bytes,
// and ds_write_b128 moves only ~79 B/clk/CU: at 128x128x64 tiles the LDS write path alone (830 clk per pair of
// k-steps) plus fragment reads (512 clk) exceeds the MFMA time (1024 clk) -- measured: removing the MFMAs from the
// register-staged kernel saved only 30 %.  LDS-DMA writes the tile straight from the memory pipe (no VGPRs, no
// ds_write): a wave instruction deposits 64 lanes x 16 B = 1 KiB LINEARLY at a wave-uniform LDS address, while
// the SOURCE address is per lane, so gathers, transposed operands and the bank swizzle are all expressed on the
// source side; out-of-range / padding chunks are fetched from a zero page.
//   KC image  [128 rows][8 slots of 16 B]: slot (row, c) holds source chunk c ^ (row & 7)          (ds_read_b128)
//   MC image  [64 k][16 slots of 16 B]   : slot (k, c) holds source chunk c ^ swz(k),
//             swz(k) = 2 * ((k & 3) | (((k >> 3) & 1) << 2))   -> the 8 k-rows one half-wave touches in a
//             ds_read_b64_tr_b16 land on 8 disjoint 32-byte bank windows (conflict-free)
// =============================================================================================
__device__ __attribute__((aligned(64))) unsigned int nk_zero_page[16];

typedef __attribute__((address_space(1))) const void* nk_gptr;
typedef __attribute__((address_space(3))) void* nk_lptr;

#define V2_OPND_BYTES 16384
#define V2_STAGE_BYTES (2 * V2_OPND_BYTES)
#define V2_SMEM_BYTES (BM * CS_LD * 4)   // 67584: epilogue staging is the larger need (2 stages = 65536)

__device__ __forceinline__ int mc_swz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

template <int MODE, int NP = 4>
struct OperandDMA {
  const bf16_t* P;
  long ld;
  int R, r0, wave, lane;
  int pn[NP], pbh[NP], pbw[NP];   // KCG: pixel decode of this thread's rows.  For plain (non-transposed) windows pn holds, instead of
                                  // the image index, the ELEMENT offset of (n, bh, bw, this lane's 8-channel chunk): a tap then adds
                                  // one wave-uniform offset to it (see sources(); tensors are < 2^31 elements: check_conv)
  int tkh0, tkh1, tkw0, tkw1, tc0, tc1;   // MCG: tap / channel of this thread's r-chunk (two swizzle variants; scalars, not
                                          // arrays: hipcc put the arrays in scratch and re-read them every k-step)
  bool rvalid[NP < 2 ? 2 : NP];  // KC*: row valid ; MC*: [0],[1] r-chunk variant valid

  __device__ __forceinline__ int kc_row(int i) const { return wave * (NP * 8) + i * 8 + (lane >> 3); }
  __device__ __forceinline__ int kc_chunk() const { return (lane & 7) ^ (lane >> 3); }
  __device__ __forceinline__ int mc_k(int i) const { return (wave * NP + i) * 4 + (lane >> 4); }
  __device__ __forceinline__ int mc_var(int i) const { return (((wave * NP + i) * 4) >> 3) & 1; }   // (k >> 3) & 1 of that piece
  // two-way selects instead of runtime-indexed arrays (hipcc puts those in scratch)
  __device__ __forceinline__ bool mc_valid(int v) const { return v ? rvalid[1] : rvalid[0]; }
  __device__ __forceinline__ int sel(int a0, int a1, int v) const { return v ? a1 : a0; }
  __device__ __forceinline__ int mc_chunk(int v) const { return (lane & 15) ^ (((lane >> 4) | (v << 2)) << 1); }

  __device__ __forceinline__ void init(const bf16_t* p, long ld_, int R_, int r0_, int tid, const NkGather& g) {
    P = p; ld = ld_; R = R_; r0 = r0_; wave = tid >> 6; lane = tid & 63;
    if constexpr (MODE == OP_KC || MODE == OP_KCG) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int r = r0 + kc_row(i);
        rvalid[i] = r < R;
        if constexpr (MODE == OP_KCG) {
          unsigned p_ = rvalid[i] ? (unsigned)r : 0u;
          unsigned n = fdiv(p_, g.fHoWo);
          unsigned rem = p_ - n * g.fHoWo.d;
          unsigned ph = fdiv(rem, g.fWo);
          unsigned pw = rem - ph * g.fWo.d;
          pn[i] = (int)n;
          pbh[i] = (int)ph * g.rs + g.off_h;
          pbw[i] = (int)pw * g.rs + g.off_w;
          if (g.div == 1 && (g.C & 63) == 0) pn[i] = ((pn[i] * g.H + pbh[i]) * g.W + pbw[i]) * g.C + kc_chunk() * 8;
        }
      }
    } else {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        int r = r0 + mc_chunk(v) * 8;
        rvalid[v] = r < R;
        if constexpr (MODE == OP_MCG) {
          unsigned rr = rvalid[v] ? (unsigned)r : 0u;
          unsigned tap = fdiv(rr, g.fC);
          const int c_ = (int)(rr - tap * g.fC.d);
          unsigned kh = fdiv(tap, g.fKW);
          const int kw_ = (int)(tap - kh * g.fKW.d);
          if (v == 0) { tc0 = c_; tkh0 = (int)kh; tkw0 = kw_; }
          else { tc1 = c_; tkh1 = (int)kh; tkw1 = kw_; }
        }
      }
    }
  }

  // issue this thread's NP LDS-DMA loads of k-tile [k0, k0+BK) into the operand image at `img`
  __device__ __forceinline__ void issue(int k0, int kend, char* img, const NkGather& g, const NkTapW& tw) const {
    const bf16_t* src[NP];
    sources(k0, kend, src, g, tw);
#pragma unroll
    for (int i = 0; i < NP; ++i) fire(i, src[i], img);
  }
  // piece i of this wave: 64 lanes x 16 B from per-lane sources, 1 KiB linear in the operand image
  __device__ __forceinline__ void fire(int i, const bf16_t* s, char* img) const {
    __builtin_amdgcn_global_load_lds((nk_gptr)s, (nk_lptr)(img + (wave * NP + i) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void sources(int k0, int kend, const bf16_t* (&src)[NP], const NkGather& g, const NkTapW& tw) const {
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
    if constexpr (MODE == OP_KC) {
      int k = k0 + kc_chunk() * 8;
      bool kv = k < kend;
#pragma unroll
      for (int i = 0; i < NP; ++i) src[i] = (kv && rvalid[i]) ? P + (long)(r0 + kc_row(i)) * ld + k : zp;
    } else if constexpr (MODE == OP_KCG) {
      int k = k0 + kc_chunk() * 8;
      bool kv = k < kend;
      unsigned tap, kh;
      int c, kw;
      if (((g.C | k0) & 63) == 0) {
        // channel counts in whole 64-deep slabs (every 3 x 3 convolution of the UNet and the VAE except the image / latent
        // ends): the slab [k0, k0 + 64) lies inside ONE tap, so its decode is wave-uniform scalar work
        const unsigned k0u = (unsigned)k0;
        tap = fdiv(k0u, g.fC);
        const int c0 = (int)(k0u - tap * g.fC.d);
        c = c0 + kc_chunk() * 8;
        kh = fdiv(tap, g.fKW);
        kw = (int)(tap - kh * g.fKW.d);
        if (g.div == 1) {
          // ... and for a plain window the tap is ONE scalar offset on top of the pixel's own (pbase): a piece costs two
          // adds and two compares for the bounds and one 64-bit add for the address (the general form below: ~25 vector
          // instructions per piece, six of them integer multiplies)
          const int dh = (int)kh * g.ks, dw = kw * g.ks;
          const int tapoff = (dh * g.W + dw) * g.C + c0;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            const bool ok = kv && rvalid[i] && (unsigned)(pbh[i] + dh) < (unsigned)g.H && (unsigned)(pbw[i] + dw) < (unsigned)g.W;
            src[i] = ok ? P + (long)(pn[i] + tapoff) : zp;
          }
          return;
        }
      } else {
        unsigned kk = kv ? (unsigned)k : 0u;
        tap = fdiv(kk, g.fC);
        c = (int)(kk - tap * g.fC.d);
        kh = fdiv(tap, g.fKW);
        kw = (int)(tap - kh * g.fKW.d);
        if (g.div == 1 && (g.C & 63) == 0) {   // (pn holds the plain-window base: a slab that does not start on a 64-channel boundary decodes per lane)
          const int dh = (int)kh * g.ks, dw = kw * g.ks;
          const int tapoff = (dh * g.W + dw) * g.C + c - kc_chunk() * 8;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            const bool ok = kv && rvalid[i] && (unsigned)(pbh[i] + dh) < (unsigned)g.H && (unsigned)(pbw[i] + dw) < (unsigned)g.W;
            src[i] = ok ? P + (long)(pn[i] + tapoff) : zp;
          }
          return;
        }
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        bool ok = kv && rvalid[i];
        long off = gather_offset(g, pn[i], pbh[i], pbw[i], (int)kh, kw, c, ok);
        src[i] = ok ? P + off : zp;
      }
    } else if constexpr (MODE == OP_MC) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int k = k0 + mc_k(i);
        const int v = mc_var(i);
        src[i] = (k < kend && mc_valid(v)) ? P + (long)k * ld + r0 + mc_chunk(v) * 8 : zp;
      }
    } else if constexpr (MODE == OP_MCT) {
      if (((tw.fCout.d | (unsigned)k0) & 63) == 0) {       // a slab is one tap (see OP_KCG): scalar decode, one address per piece
        const unsigned k0u = (unsigned)k0;
        const unsigned tap = fdiv(k0u, tw.fCout);
        const unsigned co0 = k0u - tap * tw.fCout.d;
        const bf16_t* base = P + (long)co0 * tw.co_stride + (long)tap * tw.tap_stride + r0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int v = mc_var(i);
          const bool ok = k0 + mc_k(i) < kend && mc_valid(v);
          src[i] = ok ? base + (long)mc_k(i) * tw.co_stride + mc_chunk(v) * 8 : zp;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          int k = k0 + mc_k(i);
          const int v = mc_var(i);
          bool ok = k < kend && mc_valid(v);
          unsigned kk = ok ? (unsigned)k : 0u;
          unsigned tap = fdiv(kk, tw.fCout);
          unsigned co = kk - tap * tw.fCout.d;
          src[i] = ok ? P + (long)co * tw.co_stride + (long)tap * tw.tap_stride + r0 + mc_chunk(v) * 8 : zp;
        }
      }
    } else {  // OP_MCG
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int k = k0 + mc_k(i);
        const int v = mc_var(i);
        bool ok = (k < kend) && mc_valid(v);
        unsigned p_ = ok ? (unsigned)k : 0u;
        unsigned n = fdiv(p_, g.fHoWo);
        unsigned rem = p_ - n * g.fHoWo.d;
        unsigned ph = fdiv(rem, g.fWo);
        unsigned pw = rem - ph * g.fWo.d;
        long off = gather_offset(g, (int)n, (int)ph * g.rs + g.off_h, (int)pw * g.rs + g.off_w, sel(tkh0, tkh1, v), sel(tkw0, tkw1, v), sel(tc0, tc1, v), ok);
        src[i] = ok ? P + off : zp;
      }
    }
  }

  // ---- running form (stream-K kernel): the k position and, for the plain strided modes, this thread's source pointers
  // are carried from slab to slab, so a k-step costs one 64-bit add and one select per piece (operands are re-initialised
  // inside the k loop there, which keeps hipcc from hoisting the row * ld products the way it does for issue()) ----
  int kcur;
  const bf16_t* rp[NP];
  // MCG running state (plain windows): the piece's pixel as input row / column WITH this lane's tap applied, and the element
  // offset of (n, row, column, channel chunk).  A slab moves every pixel 64 places along the (n, ph, pw) raster: two carries.
  int mh[NP], mw[NP], moff[NP];
  bool mplain;
  __device__ __forceinline__ void start(int k_begin, const NkGather& g) {
    start(k_begin);
    if constexpr (MODE == OP_MCG) {
      const int q = (int)fdiv(64u, g.fWo);
      mplain = g.div == 1 && q + 1 < g.Ho;
      if (mplain) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int v = mc_var(i);
          const unsigned p_ = (unsigned)(k_begin + mc_k(i));
          const unsigned n = fdiv(p_, g.fHoWo);
          const unsigned rem = p_ - n * g.fHoWo.d;
          const unsigned ph = fdiv(rem, g.fWo);
          const unsigned pw = rem - ph * g.fWo.d;
          mh[i] = (int)ph * g.rs + g.off_h + sel(tkh0, tkh1, v) * g.ks;
          mw[i] = (int)pw * g.rs + g.off_w + sel(tkw0, tkw1, v) * g.ks;
          moff[i] = (((int)n * g.H + mh[i]) * g.W + mw[i]) * g.C + sel(tc0, tc1, v);
        }
      }
    }
  }
  __device__ __forceinline__ void start(int k_begin) {
    kcur = k_begin;
    if constexpr (MODE == OP_KC) {
#pragma unroll
      for (int i = 0; i < NP; ++i) rp[i] = P + (long)(r0 + kc_row(i)) * ld + (k_begin + kc_chunk() * 8);
    } else if constexpr (MODE == OP_MC) {
#pragma unroll
      for (int i = 0; i < NP; ++i) rp[i] = P + (long)(k_begin + mc_k(i)) * ld + r0 + mc_chunk(mc_var(i)) * 8;
    }
  }
  __device__ __forceinline__ void next_sources(int kend, const bf16_t* (&src)[NP], const NkGather& g, const NkTapW& tw) {
    if constexpr (MODE == OP_KC || MODE == OP_MC) {
      const bf16_t* zp = (const bf16_t*)nk_zero_page;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        bool ok;
        if constexpr (MODE == OP_KC) ok = (kcur + kc_chunk() * 8 < kend) && rvalid[i];
        else ok = (kcur + mc_k(i) < kend) && mc_valid(mc_var(i));
        src[i] = ok ? rp[i] : zp;
        rp[i] += MODE == OP_KC ? (long)BK : (long)BK * ld;   // (ld is wave-uniform: a scalar multiply)
      }
    } else if constexpr (MODE == OP_MCG) {
      if (mplain) {
        // (the general form below costs ~45 vector instructions per piece and slab -- two magic-number divisions and a five-term
        // 64-bit offset; with a free decode the conv weight gradients ran 20-32 % faster: an ablation build of round 2)
        const bf16_t* zp = (const bf16_t*)nk_zero_page;
        const int q = (int)fdiv(64u, g.fWo);
        const int rm = 64 - q * g.Wo;
        const int wspan = g.Wo * g.rs, hspan = g.Ho * g.rs;
        const int A = (q * g.rs * g.W + rm * g.rs) * g.C, Bc = (g.rs * g.W - wspan) * g.C, Ci = (g.H - hspan) * g.W * g.C;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int v = mc_var(i);
          const bool ok = (kcur + mc_k(i) < kend) && mc_valid(v) && (unsigned)mh[i] < (unsigned)g.H && (unsigned)mw[i] < (unsigned)g.W;
          src[i] = ok ? P + (long)moff[i] : zp;
          // 64 pixels on: the tapped column / row keep their offset from the untapped ones, so the carries are tested against
          // limits shifted by this lane's tap
          const int dw = sel(tkw0, tkw1, v) * g.ks + g.off_w, dh = sel(tkh0, tkh1, v) * g.ks + g.off_h;
          int w = mw[i] + rm * g.rs;
          const bool c1 = w - dw >= wspan;
          w -= c1 ? wspan : 0;
          int h = mh[i] + (q + (c1 ? 1 : 0)) * g.rs;
          const bool c2 = h - dh >= hspan;
          h -= c2 ? hspan : 0;
          mw[i] = w; mh[i] = h;
          moff[i] += A + (c1 ? Bc : 0) + (c2 ? Ci : 0);
        }
      } else {
        sources(kcur, kend, src, g, tw);
      }
    } else {
      sources(kcur, kend, src, g, tw);
    }
    kcur += BK;
  }
  __device__ __forceinline__ void issue_next(int kend, char* img, const NkGather& g, const NkTapW& tw) {
    const bf16_t* src[NP];
    next_sources(kend, src, g, tw);
#pragma unroll
    for (int i = 0; i < NP; ++i) fire(i, src[i], img);
  }

  static __device__ __forceinline__ bf16x8_t frag(const char* img, int sub, int ks, int lane) {
    if constexpr (MODE == OP_KC || MODE == OP_KCG) {
      int row = sub + (lane & 15);
      int chunk = ks * 4 + (lane >> 4);
      int byte = row * 128 + ((chunk ^ (row & 7)) << 4);
      return *(const bf16x8_t*)(img + byte);
    } else {
      int g = lane >> 4, i = lane & 15;
      int q = i >> 2, p = i & 3;
      int k = ks * 32 + 8 * g + q;
      int col = sub + 4 * p;                       // element column; sub is a multiple of 16
      int byte_lo = k * 256 + (((col >> 3) ^ mc_swz(k)) << 4) + (col & 7) * 2;
      int k2 = k + 4;
      int byte_hi = k2 * 256 + (((col >> 3) ^ mc_swz(k2)) << 4) + (col & 7) * 2;
      typedef __attribute__((address_space(3))) short4_t* lds_p;
      short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + byte_lo));
      short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + byte_hi));
      short8_t r;
      r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
      r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
      return __builtin_bit_cast(bf16x8_t, r);
    }
  }
};

// ---- bias gradient inside a weight-gradient kernel ----
// The A operand of a weight-gradient GEMM is dy (rows = output features, k = tokens / pixels): its row sums ARE the bias gradient.  The
// first column tile of every row block accumulates them with an MFMA against ones -- acc[row][any column] += sum_k A(row, k) -- one per
// 16-row block and k sub-step, shared out over the waves that hold the same rows.  Deterministic unless K is split (then fp32 atomics,
// as for the weight gradient itself).
__device__ __forceinline__ bf16x8_t nk_ones_frag() {
  const short8_t o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};     // bf16 1.0
  return __builtin_bit_cast(bf16x8_t, o);
}
// standard operand order (acc row = (lane >> 4) * 4 + r, column = lane & 15): lanes of column 0 hold the 16 row sums of the block
__device__ __forceinline__ void nk_store_bias_rows(float* db, const float4_t& a, int row0, int M, float alpha, int mode, int lane) {
  if (lane & 15) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = row0 + (lane >> 4) * 4 + r;
    if (m < M) {
      const float v = a[r] * alpha;
      if (mode == 2) unsafeAtomicAdd(db + m, v);        // K split over workgroups (destination zeroed or accumulating)
      else db[m] = mode == 1 ? db[m] + v : v;
    }
  }
}

// NW = 4: waves 2x2, 64x64 per wave, 2 waves per SIMD at two workgroups per CU.
// NW = 8: waves 2x4, 64x32 per wave, 4 waves per SIMD: same tile, same LDS, twice the waves to cover each other's
//         DMA waits and fragment-read latency (under-filled grids run one workgroup per CU, i.e. 1 vs 2 waves per SIMD).
template <int AMODE, int BMODE, int OUT_F32, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void nk_gemm_dma_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = NW == 4 ? 4 : 2;      // 16-column MFMA tiles per wave
  constexpr int NP = 16 / NW;              // DMA pieces per thread per operand tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = NW == 4 ? (wave >> 1) : (wave >> 2), wn = NW == 4 ? (wave & 1) : (wave & 3);

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int GMv = p.group_m;      // rows of a tile patch: ~sqrt(tiles per XCD), so that an XCD's tiles share as few operand panels as possible
  const int per_group = GMv * ntn;
  const int group = wg / per_group;
  const int first_m = group * GMv;
  const int gm = min(GMv, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int mt = first_m + (in_group - nt * gm);
  const int m0 = mt * BM, n0 = nt * BN;

  const int kbeg = blockIdx.y * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int nk = (kend - kbeg + BK - 1) / BK;

  OperandDMA<AMODE, NP> opa;
  OperandDMA<BMODE, NP> opb;
  opa.init(p.nbatch ? p.Ab[blockIdx.z] : p.A, p.lda, p.M, m0, tid, p.ga);
  opb.init(p.nbatch ? p.Bb[blockIdx.z] : p.B, p.ldb, p.N, n0, tid, p.gb);

  float4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient (weight-gradient instantiations, first column tile): this wave's share of the row blocks it holds fragments of
  constexpr bool CAN_BIAS = AMODE == OP_MC && OUT_F32 == 1;
  constexpr int NWN = NW == 4 ? 2 : 4, BPW = 4 / NWN;         // waves sharing the same rows; row blocks per wave
  float* const dbias = CAN_BIAS ? (p.nbatch ? p.dbias_b[blockIdx.z] : p.dbias) : nullptr;
  const bool do_bias = CAN_BIAS && dbias != nullptr && nt == 0;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  float4_t accb[BPW];
#pragma unroll
  for (int u = 0; u < BPW; ++u) accb[u] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // rotated k order (OpG2::rotate in gemm_g2.h has the why): XCD x walks slabs s0, ..., nk - 1, 0, ..., s0 - 1 of its k range, s0 = x nk / 8
  // (not the gathered weight-gradient instantiations: the restart costs them registers they do not have)
  constexpr bool CAN_ROT = AMODE == OP_KC || AMODE == OP_KCG || (AMODE == OP_MC && BMODE == OP_MC);
  const int s0 = CAN_ROT && p.k_rotate ? (xcd * nk) >> 3 : 0;
  opa.start(kbeg + s0 * BK, p.ga);      // running source pointers, as in the ring and stream-K kernels
  opb.start(kbeg + s0 * BK, p.gb);
  if (nk > 0) {
    opa.issue_next(kend, smem, p.ga, p.tw);
    opb.issue_next(kend, smem + V2_OPND_BYTES, p.gb, p.tw);
  }
  for (int kt = 0; kt < nk; ++kt) {
    // vmcnt(0) + barrier: tile kt has landed for every wave, and every wave is done reading the other buffer.  The wait is
    // spelled out: left to __syncthreads(), hipcc has been seen placing it AFTER the barrier (it orders the DMA against this
    // wave's own LDS reads only), which lets a wave read rows another wave's DMA has not delivered yet.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* cur = smem + (kt & 1) * V2_STAGE_BYTES;
    // fragment reads of BOTH k sub-steps first, THEN the next tile's DMA, then the MFMAs.  hipcc (ROCm 7.2) puts an
    // s_waitcnt vmcnt(0) in front of ds_read_b64_tr_b16 whenever an LDS-DMA is outstanding (it cannot prove the
    // transposing read does not alias the DMA's LDS destination), which serialised the whole pipeline for the
    // r-contiguous operand modes (-22 %); with the DMA issued after the reads there is nothing outstanding to wait for.
    bf16x8_t af[2][4], bfr[2][NJ];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = OperandDMA<AMODE>::frag(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[ks][j] = OperandDMA<BMODE>::frag(cur + V2_OPND_BYTES, wn * (NJ * 16) + j * 16, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * V2_STAGE_BYTES;
      if constexpr (CAN_ROT) {
        if (kt + 1 == nk - s0) {            // the rotated order wraps to the start of the k range
          opa.start(kbeg, p.ga);
          opb.start(kbeg, p.gb);
        }
      }
      opa.issue_next(kend, nxt, p.ga, p.tw);
      opb.issue_next(kend, nxt + V2_OPND_BYTES, p.gb, p.tw);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    if constexpr (CAN_BIAS) {
      if (do_bias) {
        // (the row block is chosen by a wave-uniform switch over COMPILE-TIME indices: af[ks][wn] with a run-time wn would put the
        // fragment array in scratch memory -- measured: weight gradients at 90 TFLOP/s)
        const bf16x8_t ones = nk_ones_frag();
#pragma unroll
        for (int u = 0; u < BPW; ++u) {
          switch (wn_s * BPW + u) {
            case 0: accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][0], ones, accb[u], 0, 0, 0); accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][0], ones, accb[u], 0, 0, 0); break;
            case 1: accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][1], ones, accb[u], 0, 0, 0); accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][1], ones, accb[u], 0, 0, 0); break;
            case 2: accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][2], ones, accb[u], 0, 0, 0); accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][2], ones, accb[u], 0, 0, 0); break;
            default: accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][3], ones, accb[u], 0, 0, 0); accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][3], ones, accb[u], 0, 0, 0); break;
          }
        }
      }
    }
  }
  if constexpr (CAN_BIAS) {
    if (do_bias) {
      const int mode = gridDim.y > 1 ? 2 : (p.accumulate ? 1 : 0);
#pragma unroll
      for (int u = 0; u < BPW; ++u) nk_store_bias_rows(dbias, accb[u], m0 + wm * 64 + (wn * BPW + u) * 16, p.M, p.alpha, mode, lane);
    }
  }
  __syncthreads();
  nk_gemm_epilogue<OUT_F32, BM, NW * 64, NJ>(p, smem, acc, m0, n0, tid, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------
// Deep-ring variant for UNDER-FILLED grids (<= one workgroup per CU, e.g. the 100-tile 1280x1280 weight gradients): a lone
// 8-wave workgroup cannot hide a loaded global-memory latency behind one k-step of its own MFMAs, and the LDS the second
// workgroup would have used is idle -- so the ring gets FOUR stages (three slabs in flight), with counted s_waitcnt vmcnt
// and raw barriers.  r-contiguous operands are read with ds_read_b64_tr_b16 through inline asm: hipcc would drain vmcnt(0)
// in front of the builtin whenever an LDS-DMA is outstanding, which here is always.
// ---------------------------------------------------------------------------------------------
#define RING_NS 4
#define RING_SMEM_BYTES (RING_NS * V2_STAGE_BYTES)    // 131072 >= 67584 (epilogue staging)

template <int MODE>
__device__ __forceinline__ bf16x8_t ring_frag(const char* img, int sub, int ks, int lane) {
  if constexpr (MODE == OP_KC || MODE == OP_KCG) {
    return OperandDMA<MODE>::frag(img, sub, ks, lane);
  } else {
    int g = lane >> 4, i = lane & 15;
    int q = i >> 2, p = i & 3;
    int k = ks * 32 + 8 * g + q;
    int col = sub + 4 * p;
    int byte_lo = k * 256 + (((col >> 3) ^ mc_swz(k)) << 4) + (col & 7) * 2;
    int k2 = k + 4;
    int byte_hi = k2 * 256 + (((col >> 3) ^ mc_swz(k2)) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) const char* lds_c;
    const unsigned a_lo = (unsigned)(size_t)(lds_c)(img + byte_lo), a_hi = (unsigned)(size_t)(lds_c)(img + byte_hi);
    short4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_lo));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a_hi));
    short8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

template <int AMODE, int BMODE, int OUT_F32>
__global__ __launch_bounds__(512, 2) void nk_gemm_ring_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = 2, NP = 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int GMv = p.group_m;      // rows of a tile patch: ~sqrt(tiles per XCD), so that an XCD's tiles share as few operand panels as possible
  const int per_group = GMv * ntn;
  const int group = wg / per_group;
  const int first_m = group * GMv;
  const int gm = min(GMv, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int mt = first_m + (in_group - nt * gm);
  const int m0 = mt * BM, n0 = nt * BN;

  const int kbeg = blockIdx.y * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int nk = (kend - kbeg + BK - 1) / BK;

  OperandDMA<AMODE, NP> opa;
  OperandDMA<BMODE, NP> opb;
  opa.init(p.A, p.lda, p.M, m0, tid, p.ga);
  opb.init(p.B, p.ldb, p.N, n0, tid, p.gb);

  float4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  constexpr bool CAN_BIAS = AMODE == OP_MC && OUT_F32 == 1;      // bias gradient: see nk_gemm_dma_kernel
  const bool do_bias = CAN_BIAS && p.dbias != nullptr && nt == 0;
  float4_t accb = (float4_t){0.f, 0.f, 0.f, 0.f};

  // running source pointers (slabs are issued in k order): a k-step costs one 64-bit add and one select per piece instead of
  // the k * ld products -- this loop had 65 vector instructions per 16 MFMAs, 12 of them integer multiplies
  opa.start(kbeg, p.ga);
  opb.start(kbeg, p.gb);
#pragma unroll
  for (int t = 0; t < RING_NS - 1; ++t)
    if (t < nk) {
      opa.issue_next(kend, smem + t * V2_STAGE_BYTES, p.ga, p.tw);
      opb.issue_next(kend, smem + t * V2_STAGE_BYTES + V2_OPND_BYTES, p.gb, p.tw);
    }
  int cur_stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // every wave issued 4 pieces per slab, in order: slab kt has landed once at most the pieces of the later slabs remain
    const int later = min(nk - 1 - kt, RING_NS - 2);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // slab kt is in LDS for every wave; every wave is done reading stage (kt - 1) % NS
    const char* cur = smem + cur_stage * V2_STAGE_BYTES;
    bf16x8_t af[2][4], bfr[2][NJ];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = ring_frag<AMODE>(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[ks][j] = ring_frag<BMODE>(cur + V2_OPND_BYTES, wn * (NJ * 16) + j * 16, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + RING_NS - 1 < nk) {
      int ns = cur_stage + RING_NS - 1; if (ns >= RING_NS) ns -= RING_NS;
      opa.issue_next(kend, smem + ns * V2_STAGE_BYTES, p.ga, p.tw);
      opb.issue_next(kend, smem + ns * V2_STAGE_BYTES + V2_OPND_BYTES, p.gb, p.tw);
    }
    // the asm reads are invisible to hipcc's wait insertion: wait here, with the fragments as operands so the MFMAs below
    // cannot be scheduled above it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]),
                   "+v"(af[1][3]), "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    if constexpr (CAN_BIAS) {
      if (do_bias) {     // (compile-time fragment indices behind a wave-uniform switch: see nk_gemm_dma_kernel)
        const bf16x8_t ones = nk_ones_frag();
        switch (__builtin_amdgcn_readfirstlane(wn)) {
          case 0: accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][0], ones, accb, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][0], ones, accb, 0, 0, 0); break;
          case 1: accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][1], ones, accb, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][1], ones, accb, 0, 0, 0); break;
          case 2: accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][2], ones, accb, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][2], ones, accb, 0, 0, 0); break;
          default: accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][3], ones, accb, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][3], ones, accb, 0, 0, 0); break;
        }
      }
    }
    if (++cur_stage == RING_NS) cur_stage = 0;
  }
  if constexpr (CAN_BIAS) {
    if (do_bias) nk_store_bias_rows(p.dbias, accb, m0 + wm * 64 + wn * 16, p.M, p.alpha, gridDim.y > 1 ? 2 : (p.accumulate ? 1 : 0), lane);
  }
  __syncthreads();
  nk_gemm_epilogue<OUT_F32, BM, 512, NJ>(p, smem, acc, m0, n0, tid, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------
// Few-row Linear forward (M <= 512: the frozen text towers' 308 rows, the UNet's time / label embeddings): 64 x 64 tiles, EIGHT-stage ring.
// Such a launch is a weight-streaming problem run by a few dozen workgroups, and a workgroup's fetch rate is latency x bytes in flight: a
// 128 x 128 tile moves (128 + 128) x K x 2 bytes through one CU with three 32-KiB slabs in flight (~45 GB/s: 15 us at K = 1280, whatever N is).
// A 64 x 64 tile halves the bytes per workgroup, quadruples the workgroups (308 x 1280: 100 instead of 30) and, at 16 KiB per slab, keeps SEVEN
// slabs in flight in the same LDS.  k-contiguous dense operands, bf16 output with the fused bias / row vector / residual; 8 waves as 2 x 4, 32 x 16
// per wave.  Column tiles are the slow index of the tile order, so the row tiles that share a weight panel are neighbours on one XCD.
// ---------------------------------------------------------------------------------------------
#define R64_NS 8
#define R64_OPND 8192
#define R64_STAGE (2 * R64_OPND)
#define R64_SMEM (R64_NS * R64_STAGE)      // 131072
#define R64_CS_LD 68
__global__ __launch_bounds__(512, 2) void nk_gemm_ring64_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntm = (p.M + 63) >> 6;
  const int nt = wg / ntm, mt = wg - nt * ntm;
  const int m0 = mt * 64, n0 = nt * 64;
  const int kend = p.K, nk = (p.K + BK - 1) / BK;

  OperandDMA<OP_KC, 1> opa, opb;        // 8 waves x 1 piece x 8 rows = 64 rows each
  opa.init(p.A, p.lda, p.M, m0, tid, p.ga);
  opb.init(p.B, p.ldb, p.N, n0, tid, p.gb);
  float4_t acc[2] = {(float4_t){0.f, 0.f, 0.f, 0.f}, (float4_t){0.f, 0.f, 0.f, 0.f}};
  opa.start(0);
  opb.start(0);
#pragma unroll
  for (int t = 0; t < R64_NS - 1; ++t)
    if (t < nk) {
      opa.issue_next(kend, smem + t * R64_STAGE, p.ga, p.tw);
      opb.issue_next(kend, smem + t * R64_STAGE + R64_OPND, p.gb, p.tw);
    }
  int cur_stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // every wave issued 2 pieces per slab, in order: slab kt has landed once at most the pieces of the later slabs remain
    switch (min(nk - 1 - kt, R64_NS - 2)) {
      case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();   // slab kt is in LDS for every wave; every wave is done reading stage (kt - 1) % NS
    const char* cur = smem + cur_stage * R64_STAGE;
    bf16x8_t af[2][2], bfr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[ks][i] = ring_frag<OP_KC>(cur, wm * 32 + i * 16, ks, lane);
      bfr[ks] = ring_frag<OP_KC>(cur + R64_OPND, wn * 16, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + R64_NS - 1 < nk) {
      int ns = cur_stage + R64_NS - 1; if (ns >= R64_NS) ns -= R64_NS;
      opa.issue_next(kend, smem + ns * R64_STAGE, p.ga, p.tw);
      opb.issue_next(kend, smem + ns * R64_STAGE + R64_OPND, p.gb, p.tw);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bfr[0]), "+v"(bfr[1]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks], acc[i], 0, 0, 0);
    if (++cur_stage == R64_NS) cur_stage = 0;
  }
  __syncthreads();
  // ---- epilogue: accumulators -> LDS (fp32, [64][R64_CS_LD]) -> one 16-byte store per thread ----
  float* cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[(wm * 32 + i * 16 + (lane >> 4) * 4 + r) * R64_CS_LD + wn * 16 + (lane & 15)] = acc[i][r];
  __syncthreads();
  const int row = tid >> 3, cc = tid & 7;
  const int m = m0 + row, n = n0 + cc * 8;
  if (m >= p.M || n >= p.N) return;
  bf16_t* C = (bf16_t*)p.C;
  const float4_t c0 = *(const float4_t*)(cs + row * R64_CS_LD + cc * 8), c1 = *(const float4_t*)(cs + row * R64_CS_LD + cc * 8 + 4);
  float v[8] = {c0[0] * p.alpha, c0[1] * p.alpha, c0[2] * p.alpha, c0[3] * p.alpha, c1[0] * p.alpha, c1[1] * p.alpha, c1[2] * p.alpha, c1[3] * p.alpha};
  if ((p.N & 7) == 0) {
    if (p.bias) {
      const float4_t b0 = *(const float4_t*)(p.bias + n), b1 = *(const float4_t*)(p.bias + n + 4);
      v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
      v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
    }
    if (p.rowvec) {
      float t[8];
      unpack8(*(const uint4_t*)(p.rowvec + (long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n), t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += t[e];
    }
    if (p.residual) {
      float t[8];
      unpack8(*(const uint4_t*)(p.residual + (long)m * p.ldr + n), t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += t[e];
    }
    *(uint4_t*)(C + (long)m * p.ldc + n) = pack8(v);
  } else {
    for (int e = 0; e < 8 && n + e < p.N; ++e) {
      float x = v[e];
      if (p.bias) x += p.bias[n + e];
      if (p.rowvec) x += bf2f(p.rowvec[(long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n + e]);
      if (p.residual) x += bf2f(p.residual[(long)m * p.ldr + n + e]);
      C[(long)m * p.ldc + n + e] = f2bf(x);
    }
  }
}
// NK_GEMM_R64: 0 = never (A/B runs).  Dense k-contiguous bf16-output launches of at most 512 rows whose 128 x 128 grid would leave most CUs idle.
static bool use_ring64(const NkGemmParams& p, int amode, int bmode, int out_f32) {
  if (amode != OP_KC || bmode != OP_KC || out_f32 || p.nbatch || p.geglu_u || p.geglu_h || p.stats_part) return false;
  if (const char* e = getenv("NK_GEMM_R64")) if (e[0] == '0') return false;
  if (p.M > 512 || p.K < 4 * BK) return false;
  // one round at one workgroup per CU (the ring takes 128 KiB of LDS): 308 x 1280 -> 100 tiles, x 3072 -> 240; at 308 x 3840 / 5120 (300 / 400
  // tiles: two rounds) the 128 x 128 kernels are faster again (16.8 / 17.1 against 19.2 / 19.9 us, tools/bench_skinny.py)
  const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  return t64 <= 256;
}
static int launch_ring64(const NkGemmParams& p, hipStream_t stream) {
  nk_optin_lds((const void*)nk_gemm_ring64_kernel, R64_SMEM);
  const unsigned tiles = (unsigned)(((p.M + 63) >> 6) * ((p.N + 63) >> 6));
  hipLaunchKernelGGL(nk_gemm_ring64_kernel, dim3(tiles), dim3(512), R64_SMEM, stream, p);
  return nk_check_launch("nk_gemm_ring64_kernel");
}

// =============================================================================================
// stream-K main kernel (default).  Measured on the 128x128 data-parallel kernel above: a k-step costs 0.94 us per
// round of 512 tiles, but every round also pays ~4.5 us of fixed cost (first loads from a cold pipeline, LDS-staged
// epilogue, all 512 workgroups loading and then storing in lock-step), and the SDXL shapes quantise badly (320 tiles on
// 512 slots; 100-tile weight gradients): 166 ms of tile-engine time per step against 79 ms at the marginal rate.
//   * PERSISTENT workgroups (<= 2 per CU) each own an equal contiguous share of the (tile, k-step) iteration space,
//     so the chip is full whatever the tile count; no split-K atomics, no memset, deterministic.
//   * the LDS-DMA producer runs one slab ahead of the MFMA consumer ACROSS tile boundaries: the next tile's first
//     slab is in flight while this tile's epilogue runs.
//   * the epilogue goes straight from the accumulator registers to global memory: MFMA operands are swapped (D = B.A^T,
//     so a lane holds 4 consecutive columns of one row) and one v_permlane16_swap per register pair widens that to 8
//     consecutive columns (16 B of bf16 / 32 B of fp32 per lane) -- no LDS staging, no barrier, the ring stays free.
//   * a tile shared by several workgroups: every workgroup but the last writes its fp32 partial (register order, fully
//     coalesced) to the workspace and raises a flag; the LAST one (highest ticket) adds them in fixed order and runs
//     the fused epilogue.  Workgroups walk their share from its END, so a partial is written at the start of a run and
//     consumed at the end of the neighbour's.  A waiter only waits for LOWER-indexed workgroups of its own XCD, which the
//     dispatcher has already handed out; the wait is bounded anyway (a failed launch is flagged, never a hang).
// =============================================================================================
#define SK_NT 512
#define SK_SMEM_BYTES (2 * V2_STAGE_BYTES)
#define SK_MAX_GRID 512
#define SK_TILE_FLOATS (BM * BN)

__device__ __forceinline__ void sk_decode(const NkGemmParams& p, int t, int ntm, int ntn, int& m0, int& n0, int& z) {
  const int per = ntm * ntn;
  z = p.nbatch ? t / per : 0;
  const int wg = t - z * per;
  const int per_group = GROUP_M * ntn;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(GROUP_M, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  m0 = (first_m + (in_group - nt * gm)) * BM;
  n0 = nt * BN;
}

// The residual's chunks a lane will add in reg_epilogue_64x32 (same row / column arithmetic), fetched BEFORE the k loop by kernels whose waves
// have the registers: the epilogue of a 20-k-step launch otherwise starts with a round trip to memory (bf16 output, N % 8 == 0 only).
template <int MI>
__device__ __forceinline__ void residual_prefetch_64x32(const NkGemmParams& p, uint4_t (&r)[MI], int mbase, int nbase, int lane) {
  const int g = lane >> 4;
  const int n = nbase + (g & 1) * 16 + (g >> 1) * 8;
  const int mrow = mbase + (lane & 15);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + i * 16;
    r[i] = (n < p.N && m < p.M) ? *(const uint4_t*)(p.residual + (long)m * p.ldr + n) : (uint4_t){0u, 0u, 0u, 0u};
  }
}

// accumulators -> global, fused bias / rowvec / residual.  acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane&15)]
//                                                                           [n0 + wn*32 + j*16 + (lane>>4)*4 + r]
template <int OUT_F32, int MI = 4>     // MI 16-row blocks x one pair of 16-column blocks
__device__ __forceinline__ void reg_epilogue_64x32(const NkGemmParams& p, void* Cv, float4_t (&acc)[MI][2], int mbase, int nbase, int lane,
                                                   int mlimit,          // mlimit >= 0: rows at or past it are not stored (instead of p.M)
                                                   const uint4_t (&pre_res)[MI], bool use_pre) {   // use_pre (wave-uniform): the residual's chunks were fetched by the caller (residual_prefetch_64x32)
  const int Mrows = mlimit >= 0 ? mlimit : p.M;
  const int g = lane >> 4;
  // after the row swap: lanes g=0 hold columns 0-7 of the wave's 32, g=1 16-23, g=2 8-15, g=3 24-31
  const int n = nbase + (g & 1) * 16 + (g >> 1) * 8;
  const int mrow = mbase + (lane & 15);
  const bool n_ok = n < p.N;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = 0.f;
  if constexpr (!OUT_F32) {
    if (p.bias && n_ok) {
      if ((p.N & 7) == 0) {
        const float4_t b0 = *(const float4_t*)(p.bias + n), b1 = *(const float4_t*)(p.bias + n + 4);
        bias[0] = b0[0]; bias[1] = b0[1]; bias[2] = b0[2]; bias[3] = b0[3];
        bias[4] = b1[0]; bias[5] = b1[1]; bias[6] = b1[2]; bias[7] = b1[3];
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) bias[e] = p.bias[n + e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // every lane takes part in the swap; the bounds predicates come after it
      // (plain float temporaries: __builtin_bit_cast applied directly to an ext-vector element read element 0 for every r)
      const float a_own = acc[i][0][r], b_own = acc[i][1][r];
      auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a_own), __float_as_uint(b_own), false, false);
      v[r] = __uint_as_float(sw[0]) * p.alpha;
      v[4 + r] = __uint_as_float(sw[1]) * p.alpha;
    }
    const int m = mrow + i * 16;
    if (!(n_ok && m < Mrows)) continue;
    if constexpr (OUT_F32) {
      float* dst = (float*)Cv + (long)m * p.ldc + n;
      if ((p.N & 3) == 0 && (p.ldc & 3) == 0 && n + 8 <= p.N) {
        float4_t lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        if (p.accumulate) { lo += *(const float4_t*)dst; hi += *(const float4_t*)(dst + 4); }
        *(float4_t*)dst = lo;
        *(float4_t*)(dst + 4) = hi;
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
      }
    } else {
      bf16_t* C = (bf16_t*)Cv;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[e];
      if ((p.N & 7) == 0) {
        if (p.rowvec) {
          unsigned b = fdiv((unsigned)m, p.fRowsPerBatch);
          float t[8];
          unpack8(*(const uint4_t*)(p.rowvec + (long)b * p.ld_rowvec + n), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (p.residual) {
          float t[8];
          unpack8(use_pre ? pre_res[i] : *(const uint4_t*)(p.residual + (long)m * p.ldr + n), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        *(uint4_t*)(C + (long)m * p.ldc + n) = pack8(v);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) {
          float x = v[e];
          if (p.rowvec) x += bf2f(p.rowvec[(long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n + e]);
          if (p.residual) x += bf2f(p.residual[(long)m * p.ldr + n + e]);
          C[(long)m * p.ldc + n + e] = f2bf(x);
        }
      }
    }
  }
}
template <int OUT_F32, int MI = 4>
__device__ __forceinline__ void reg_epilogue_64x32(const NkGemmParams& p, void* Cv, float4_t (&acc)[MI][2], int mbase, int nbase, int lane,
                                                   int mlimit = -1) {
  const uint4_t none[MI] = {};
  reg_epilogue_64x32<OUT_F32, MI>(p, Cv, acc, mbase, nbase, lane, mlimit, none, false);
}
template <int OUT_F32>
__device__ __forceinline__ void sk_epilogue(const NkGemmParams& p, void* Cv, float4_t (&acc)[4][2], int m0, int n0, int lane,
                                            int wm, int wn) {
  reg_epilogue_64x32<OUT_F32>(p, Cv, acc, m0 + wm * 64, n0 + wn * 32, lane);
}

// Partial tiles travel between workgroups (possibly on different XCDs, i.e. different L2s) with agent-scope (sc1)
// write-through stores and sc1 loads: a release FENCE would write back the whole L2 (measured: +150 us per launch).
// Layout: [wave][q = i*2+j][lane] float4 -- every instruction moves one contiguous KiB; the eight q-planes of a wave
// are reached with immediate offsets from the middle of its 8 KiB region.
__device__ __forceinline__ void sk_store_partial(float* ws_tile, const float4_t (&acc)[4][2], int wave, int lane) {
  const char* ptr = (const char*)ws_tile + wave * 8192 + 4096 + lane * 16;
  asm volatile(
      "global_store_dwordx4 %0, %1, off offset:-4096 sc1\n"
      "global_store_dwordx4 %0, %2, off offset:-3072 sc1\n"
      "global_store_dwordx4 %0, %3, off offset:-2048 sc1\n"
      "global_store_dwordx4 %0, %4, off offset:-1024 sc1\n"
      "global_store_dwordx4 %0, %5, off sc1\n"
      "global_store_dwordx4 %0, %6, off offset:1024 sc1\n"
      "global_store_dwordx4 %0, %7, off offset:2048 sc1\n"
      "global_store_dwordx4 %0, %8, off offset:3072 sc1\n"
      "s_waitcnt vmcnt(0)"
      :
      : "v"(ptr), "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[2][0]), "v"(acc[2][1]), "v"(acc[3][0]), "v"(acc[3][1])
      : "memory");
}

__device__ __forceinline__ void sk_add_partial(const float* ws_tile, float4_t (&acc)[4][2], int wave, int lane) {
  const char* ptr = (const char*)ws_tile + wave * 8192 + 4096 + lane * 16;
  float4_t r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "global_load_dwordx4 %0, %8, off offset:-4096 sc1\n"
      "global_load_dwordx4 %1, %8, off offset:-3072 sc1\n"
      "global_load_dwordx4 %2, %8, off offset:-2048 sc1\n"
      "global_load_dwordx4 %3, %8, off offset:-1024 sc1\n"
      "global_load_dwordx4 %4, %8, off sc1\n"
      "global_load_dwordx4 %5, %8, off offset:1024 sc1\n"
      "global_load_dwordx4 %6, %8, off offset:2048 sc1\n"
      "global_load_dwordx4 %7, %8, off offset:3072 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(ptr)
      : "memory");
  acc[0][0] += r0; acc[0][1] += r1; acc[1][0] += r2; acc[1][1] += r3;
  acc[2][0] += r4; acc[2][1] += r5; acc[3][0] += r6; acc[3][1] += r7;
}

// One workgroup's list of (tile, k-step range) segments: first its share of the stream-K region (walked from the end),
// then its data-parallel tiles (round r: tile dp_tile0 + r*nb + l, so the workgroups of an XCD sit on ADJACENT tiles at
// any moment and share operand panels in L2 -- contiguous per-workgroup tile ranges measured 1.9x slower per k-step).
struct SkCursor {
  int sk_s, sk_end, r;
  __device__ __forceinline__ bool next(int nk, int sk_tile0, int dp_tile0, int R, int nb, int l, int& t, int& k0, int& k1, int& ts) {
    if (sk_end > sk_s) {
      const int tl = (sk_end - 1) / nk;
      ts = tl * nk;
      const int seg_s = max(sk_s, ts);
      k0 = seg_s - ts; k1 = sk_end - ts; sk_end = seg_s;
      t = sk_tile0 + tl;
      return true;
    }
    if (r < R) { t = dp_tile0 + r * nb + l; k0 = 0; k1 = nk; ts = 0; ++r; return true; }
    return false;
  }
};

template <int AMODE, int BMODE, int OUT_F32>
__global__ __launch_bounds__(SK_NT, 4) void nk_gemm_sk_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  // Work is assigned by blockIdx: workgroup b runs on XCD b % 8 (round-robin dispatch), so (b & 7, b >> 3) keeps an XCD's
  // workgroups on one contiguous eighth of the tile list whatever order they start in.
  const int w = blockIdx.x;
  const int G = gridDim.x;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int nk = (p.K + BK - 1) / BK;
  const int T = ntm * ntn * (p.nbatch ? p.nbatch : 1);

  // logical position: (xcd, l) of nb workgroups on the tile range [t0, t1)
  int xcd = 0, l = w, nb = G, t0 = 0, t1 = T;
  if (p.sk_chunked) {
    xcd = w & 7; l = w >> 3; nb = G >> 3;
    t0 = (int)(((long)T * xcd) >> 3);
    t1 = (int)(((long)T * (xcd + 1)) >> 3);
  }
  const int R = (t1 - t0) / nb;                  // full data-parallel rounds
  const int rem = (t1 - t0) - R * nb;            // tiles left for the stream-K region (< nb)
  const int sk_tile0 = t0 + R * nb;
  const int Wsk = rem * nk;
  const int max_split = max(1, min(16, nk >> 3));
  const int P = min(nb, rem * max_split);        // workgroups taking part in the stream-K region
  auto start_of = [&](int li) -> int { return (int)(((unsigned)Wsk * (unsigned)li) / (unsigned)P); };
  SkCursor cons;
  cons.r = 0;
  cons.sk_s = l < P ? start_of(l) : 0;
  cons.sk_end = l < P ? start_of(l + 1) : 0;
  SkCursor prod = cons;

  OperandDMA<AMODE, 2> opa;
  OperandDMA<BMODE, 2> opb;
  // ---- producer: one slab ahead of the consumer, across segment boundaries ----
  int p_k = 0, p_k1 = 0;
  bool p_live = false;
  auto p_next = [&]() {
    int t, ts_unused;
    p_live = prod.next(nk, sk_tile0, t0, R, nb, l, t, p_k, p_k1, ts_unused);
    if (!p_live) return;
    int m0, n0, z;
    sk_decode(p, t, ntm, ntn, m0, n0, z);
    opa.init(p.nbatch ? p.Ab[z] : p.A, p.lda, p.M, m0, tid, p.ga);
    opb.init(p.nbatch ? p.Bb[z] : p.B, p.ldb, p.N, n0, tid, p.gb);
    opa.start(p_k * BK, p.ga);
    opb.start(p_k * BK, p.gb);
  };
  auto p_issue = [&](char* stage) {
    opa.issue_next(p.K, stage, p.ga, p.tw);
    opb.issue_next(p.K, stage + V2_OPND_BYTES, p.gb, p.tw);
    if (++p_k == p_k1) p_next();
  };

  p_next();
  if (p_live) p_issue(smem);
  int slab = 0;
  int t, k0, k1, ts;
  while (cons.next(nk, sk_tile0, t0, R, nb, l, t, k0, k1, ts)) {
    float4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    for (int kt = k0; kt < k1; ++kt, ++slab) {
      // vmcnt(0) + barrier: this slab has landed for every wave, and every wave is done reading the other stage (the wait is
      // explicit for the reason given in nk_gemm_dma_kernel)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const char* cur = smem + (slab & 1) * V2_STAGE_BYTES;
      bf16x8_t af[2][4], bfr[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = OperandDMA<AMODE>::frag(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[ks][j] = OperandDMA<BMODE>::frag(cur + V2_OPND_BYTES, wn * 32 + j * 16, ks, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (p_live) p_issue(smem + ((slab + 1) & 1) * V2_STAGE_BYTES);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)   // operands swapped: D = B.A^T, a lane holds 4 consecutive COLUMNS of one row
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    }

    int m0, n0, z;
    sk_decode(p, t, ntm, ntn, m0, n0, z);
    void* C = p.nbatch ? p.Cb[z] : p.C;
    if (k1 < nk) {
      // not the end of the tile: publish the partial and raise this ticket's flag
      sk_store_partial(p.sk_ws + (size_t)w * SK_TILE_FLOATS, acc, wave, lane);
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (k0 > 0) {
        // last part of a shared tile: add the partials of the workgroups before us (lower tickets: they started earlier),
        // nearest first, down to the one that holds the tile's first k-step
        for (int li = l - 1; li >= 0; --li) {
          const int ticket = p.sk_chunked ? xcd + 8 * li : li;
          const int s_li = start_of(li);
          if (s_li < start_of(li + 1)) {   // (an empty share has nothing to add)
            // The workgroup waited for has a LOWER index on the same XCD: the dispatcher hands workgroups out in index
            // order, so it is running or done.  The wait is bounded all the same (~0.2 s): on expiry the launch is marked
            // failed (checked by nk_gemm_sk_status) instead of hanging the device.
            // FAIL CLOSED: a tile whose partials never arrived is poisoned (NaN reaches the loss / the gradient norm) and the
            // process-wide health word is raised, which makes the fused optimizer kernels skip the update of this step and
            // the next optimizer call return an error -- the launch never continues with whatever the workspace held.
            int spins = 0;
            bool gave_up = (p.sk_debug & 2) != 0;
            while (!gave_up && __hip_atomic_load(p.sk_flags + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > (1 << 21)) gave_up = true;
            }
            if (gave_up) {
              if (tid == 0) {
                __hip_atomic_store(p.sk_flags + SK_MAX_GRID + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.sk_health, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              const float poison = __builtin_nanf("");
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){poison, poison, poison, poison};
              break;      // (a publisher that does show up later finds its flag lowered by nobody: nk_health_clear() re-zeroes
                          // every workspace's flags before training may continue)
            }
            sk_add_partial(p.sk_ws + (size_t)ticket * SK_TILE_FLOATS, acc, wave, lane);
            // Every published partial has exactly one reader (the workgroup finishing that tile), which lowers the flag again
            // once all its waves are past the wait: the flags are all 0 when the launch ends, so there is no per-launch
            // state in the kernel arguments and a launch replayed from a captured hipGraph behaves like a fresh one.
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.sk_flags + ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (s_li <= ts) break;
        }
      }
      sk_epilogue<OUT_F32>(p, C, acc, m0, n0, lane, wm, wn);
    }
  }
}

// =============================================================================================
// "XL" variant (default for large k-contiguous GEMMs; NK_GEMM_XL=0 turns it off): 256 x 256 x 64 tiles, 16 waves as 4 x 4 with 64 x 64 per wave, one workgroup per CU,
// two 64 KiB LDS-DMA stages, register-direct epilogue.  The point is bytes per FLOP through the texture -> LDS path, which is
// what bounds the 128 x 128 kernel: half of its 32 B per 1 Ki MAC.  k-contiguous operands, bf16 output.
// Fragments are read one k sub-step at a time (64 accumulator + 32 fragment registers at 4 waves per SIMD).
// =============================================================================================
#define XL_BM 256
#define XL_BN 256
#define XL_STAGE_BYTES 65536
#define XL_SMEM_BYTES (2 * XL_STAGE_BYTES)
// WM x WN is the per-wave tile: 64 x 64 (16 waves, 4 per SIMD), 128 x 64 (8 waves, 2 per SIMD) or 128 x 128 (4 waves, 1 per SIMD).
// Larger wave tiles read fewer fragment bytes from LDS per MFMA (0.5 / 0.375 / 0.25 KiB) at the price of fewer waves to hide them.
template <int AMODE, int WM, int WN>
__global__ __launch_bounds__((XL_BM / WM) * (XL_BN / WN) * 64, (XL_BM / WM) * (XL_BN / WN) / 4) void nk_gemm_xl_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WCOLS = XL_BN / WN, NWAVES = (XL_BM / WM) * WCOLS, NP = 32 / NWAVES, MI = WM / 16, NJ = WN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + XL_BN - 1) / XL_BN, ntm = (p.M + XL_BM - 1) / XL_BM;
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int m0 = (first_m + (in_group - nt * gm)) * XL_BM, n0 = nt * XL_BN;
  const int kend = p.K, nk = (p.K + BK - 1) / BK;

  OperandDMA<AMODE, NP> opa;   // NWAVES x NP pieces x 8 rows = 256 rows
  OperandDMA<OP_KC, NP> opb;
  opa.init(p.A, p.lda, p.M, m0, tid, p.ga);
  opb.init(p.B, p.ldb, p.N, n0, tid, p.gb);
  float4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  opa.start(0);          // running source pointers: a k-step costs one 64-bit add per piece instead of the row * ld products
  opb.start(0);
  if (nk > 0) {
    opa.issue_next(kend, smem, p.ga, p.tw);
    opb.issue_next(kend, smem + 32768, p.gb, p.tw);
  }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slab kt has landed (this wave's share) ...
    __syncthreads();                                    // ... for every wave, and everyone is done with the other stage
    const char* cur = smem + (kt & 1) * XL_STAGE_BYTES;
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * XL_STAGE_BYTES;
      opa.issue_next(kend, nxt, p.ga, p.tw);
      opb.issue_next(kend, nxt + 32768, p.gb, p.tw);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[MI], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = OperandDMA<OP_KC>::frag(cur, wm * WM + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = OperandDMA<OP_KC>::frag(cur + 32768, wn * WN + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)   // operands swapped (D = B.A^T): a lane holds 4 consecutive COLUMNS of one row
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ib = 0; ib < MI / 4; ++ib)
#pragma unroll
    for (int half = 0; half < NJ / 2; ++half) {
      float4_t pair[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pair[i][0] = acc[ib * 4 + i][2 * half]; pair[i][1] = acc[ib * 4 + i][2 * half + 1]; }
      reg_epilogue_64x32<0>(p, p.C, pair, m0 + wm * WM + ib * 64, n0 + wn * WN + half * 32, lane);
    }
}

// GEGLU = 1: the FeedForward projection with its GEGLU (modules/attention.py:50-57) in the epilogue -- see NkGemmParams::geglu_h.  Column tile nt
// covers a-columns [128 nt, 128 nt + 128) and the gate columns I + the same: tile row blocks (16 rows of B) alternate a / gate.
template <int AMODE, int GEGLU = 0>
__global__ __launch_bounds__(512, 2) void nk_gemm_xl2g_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + XL_BN - 1) / XL_BN, ntm = (p.M + XL_BM - 1) / XL_BM;
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int m0 = (first_m + (in_group - nt * gm)) * XL_BM, n0 = nt * XL_BN;
  const int kend = p.K, nk = (p.K + BK - 1) / BK;

  OperandDMA<AMODE, 2> a0, a1;     // 8 waves x 2 pieces x 8 rows = 128 rows each
  OperandDMA<OP_KC, 2> b0, b1;
  a0.init(p.A, p.lda, p.M, m0, tid, p.ga);
  a1.init(p.A, p.lda, p.M, m0 + 128, tid, p.ga);
  if constexpr (GEGLU) {
    // a wave stages one 16-row block of each B half (kc_row = wave * 16 + ...): block gb = wave (b0) or 8 + wave (b1) of the tile holds
    // weight rows (gb odd ? I : 0) + n0 / 2 + (gb >> 1) * 16 + (0..15); the loader is given r0 such that r0 + kc_row lands there
    const int I = p.N >> 1, wv = tid >> 6;
    const int gb0 = wv, gb1 = 8 + wv;
    b0.init(p.B, p.ldb, p.N, ((gb0 & 1) ? I : 0) + (n0 >> 1) + (gb0 >> 1) * 16 - wv * 16, tid, p.gb);
    b1.init(p.B, p.ldb, p.N, ((gb1 & 1) ? I : 0) + (n0 >> 1) + (gb1 >> 1) * 16 - wv * 16, tid, p.gb);
  } else {
    b0.init(p.B, p.ldb, p.N, n0, tid, p.gb);
    b1.init(p.B, p.ldb, p.N, n0 + 128, tid, p.gb);
  }
  // rotated k order (OpG2::rotate in gemm_g2.h has the why): XCD x walks slabs s0, ..., nk - 1, 0, ..., s0 - 1, s0 = x nk / 8.  Every operand
  // counts the slabs it has handed out: after nk - s0 of them it restarts at k = 0 with s0 slabs to go, after nk it yields the zero page.
  const int s0 = p.k_rotate ? (xcd * nk) >> 3 : 0;
  const int wrap_at = nk - s0;
  int na0 = 0, na1 = 0, nb0 = 0, nb1 = 0, ka0 = kend, ka1 = kend, kb0 = kend, kb1 = kend;
  a0.start(s0 * BK); a1.start(s0 * BK); b0.start(s0 * BK); b1.start(s0 * BK);
#define X2_NEXT(op, src, g, cnt, ke)                        \
  do {                                                      \
    op.next_sources(ke, src, g, p.tw);                      \
    if (++cnt == wrap_at) { op.start(0); ke = s0 * BK; }    \
  } while (0)
#define X2_ISSUE(op, img, g, cnt, ke)                       \
  do {                                                      \
    const bf16_t* s_[2];                                    \
    X2_NEXT(op, s_, g, cnt, ke);                            \
    op.fire(0, s_[0], img); op.fire(1, s_[1], img);         \
  } while (0)
  float4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  bf16x8_t ar[8], bc0[4], bc1[4];      // A: [k sub-step][4 row blocks of the current row half]; B: [k sub-step][2 column blocks]
  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  const unsigned x0 = (unsigned)(((lane >> 4) ^ (lane & 7)) << 4);              // 16-byte slot of k sub-step 0; sub-step 1 is x0 ^ 64
  const unsigned arow = lds0 + (unsigned)(wm * 128 + (lane & 15)) * 128u;
  const unsigned brow = lds0 + 32768u + (unsigned)(wn * 64 + (lane & 15)) * 128u;
  const unsigned a_k0 = arow + x0, a_k1 = arow + (x0 ^ 64u), b_k0 = brow + x0, b_k1 = brow + (x0 ^ 64u);
#define X2_RD1(f, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"((OFF)))
#define X2_RDA(half)                                                                                             \
  X2_RD1(ar[0], ca0, (half) * 8192); X2_RD1(ar[1], ca0, (half) * 8192 + 2048); X2_RD1(ar[2], ca0, (half) * 8192 + 4096);    \
  X2_RD1(ar[3], ca0, (half) * 8192 + 6144); X2_RD1(ar[4], ca1, (half) * 8192); X2_RD1(ar[5], ca1, (half) * 8192 + 2048);    \
  X2_RD1(ar[6], ca1, (half) * 8192 + 4096); X2_RD1(ar[7], ca1, (half) * 8192 + 6144)
#define X2_RDB(f, half)                                                                                          \
  X2_RD1(f[0], cb0, (half) * 4096); X2_RD1(f[1], cb0, (half) * 4096 + 2048); X2_RD1(f[2], cb1, (half) * 4096);              \
  X2_RD1(f[3], cb1, (half) * 4096 + 2048)
#define X2_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)
#define X2_MM(rb, cb, bf, PREP)                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                             \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  PREP;              /* source addresses of the NEXT phase's DMA: VALU work that issues in the MFMAs' shadow */ \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
        acc[(rb) + i][(cb) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks * 2 + j], ar[ks * 4 + i], acc[(rb) + i][(cb) + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0)

  const bf16_t* sa0[2];
  const bf16_t* sa1[2];
  const bf16_t* sb0[2];
  const bf16_t* sb1[2];
  if (nk > 0) {
    X2_ISSUE(a0, smem, p.ga, na0, ka0);
    X2_ISSUE(a1, smem + 16384, p.ga, na1, ka1);
    X2_ISSUE(b0, smem + 32768, p.gb, nb0, kb0);
    X2_ISSUE(b1, smem + 49152, p.gb, nb1, kb1);
    X2_ISSUE(b0, smem + XL_STAGE_BYTES + 32768, p.gb, nb0, kb0);        // B of slab 1 (past the end: the zero page, never read)
    X2_ISSUE(b1, smem + XL_STAGE_BYTES + 49152, p.gb, nb1, kb1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  X2_NEXT(a1, sa1, p.ga, na1, ka1);                                     // A1 of slab 1, fired in phase 0 of slab 0
  X2_BAR();
  if (wm == 1) { X2_BAR(); }               // the second group runs one barrier behind
  for (int t = 0; t < nk; ++t) {
    const unsigned so = (unsigned)(t & 1) * XL_STAGE_BYTES;
    char* cur = smem + so;
    char* oth = smem + (so ^ XL_STAGE_BYTES);
    const unsigned ca0 = a_k0 + so, ca1 = a_k1 + so, cb0 = b_k0 + so, cb1 = b_k1 + so;
    // phase 0: rows 0-63 x columns 0-31 of the wave's tile
    __builtin_amdgcn_sched_barrier(0);
    X2_RDB(bc0, 0);
    X2_RDA(0);
    a1.fire(0, sa1[0], oth + 16384);       // A1(t+1)
    a1.fire(1, sa1[1], oth + 16384);
    X2_BAR();
    X2_MM(0, 0, bc0, X2_NEXT(a0, sa0, p.ga, na0, ka0));
    X2_BAR();
    // phase 1: rows 0-63 x columns 32-63
    X2_RDB(bc1, 1);
    a0.fire(0, sa0[0], oth);               // A0(t+1)
    a0.fire(1, sa0[1], oth);
    X2_BAR();
    X2_MM(0, 2, bc1, (void)0);
    X2_BAR();
    // phase 2: rows 64-127 x columns 32-63
    X2_RDA(1);
    X2_BAR();
    X2_MM(4, 2, bc1, X2_NEXT(b0, sb0, p.gb, nb0, kb0); X2_NEXT(b1, sb1, p.gb, nb1, kb1));
    X2_BAR();
    // phase 3: rows 64-127 x columns 0-31; B of slab t + 2 goes into THIS slab's stage (its B reads retired two phases ago)
    b0.fire(0, sb0[0], cur + 32768);
    b0.fire(1, sb0[1], cur + 32768);
    b1.fire(0, sb1[0], cur + 49152);
    b1.fire(1, sb1[1], cur + 49152);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // all but those four pieces: slab t + 1 is complete (this wave's share)
    X2_BAR();
    X2_MM(4, 0, bc0, X2_NEXT(a1, sa1, p.ga, na1, ka1));
    X2_BAR();
  }
  if (wm == 0) { X2_BAR(); }               // ... and the first group waits for it here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the past-the-end zero-page pieces
#undef X2_NEXT
#undef X2_ISSUE
#undef X2_RD1
#undef X2_RDA
#undef X2_RDB
#undef X2_BAR
#undef X2_MM
  if constexpr (GEGLU) {
    // the wave's four column blocks are (a, gate, a, gate) of a-columns abase .. abase + 31: the lane holds a and gate of the same four j
    const int I = p.N >> 1;
    const int abase = (n0 >> 1) + wn * 32;
    const int cl = (lane >> 4) * 4;
    float4_t ba[2], bg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ba[q] = p.bias ? *(const float4_t*)(p.bias + abase + q * 16 + cl) : (float4_t){0.f, 0.f, 0.f, 0.f};
      bg[q] = p.bias ? *(const float4_t*)(p.bias + I + abase + q * 16 + cl) : (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    NkGemmParams ph = p;               // the GEGLU output: [M][I], no bias (it is inside a and g already)
    ph.N = I; ph.ldc = p.ld_h; ph.bias = nullptr;
    NkGemmParams ps = p;               // the saved-derivative form of u: same place and shape, no bias
    ps.bias = nullptr;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      float4_t pa[4][2], pg[4][2], phh[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          pa[i][q] = acc[ib * 4 + i][2 * q];
          pg[i][q] = acc[ib * 4 + i][2 * q + 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // what the consumer of u would read: the bf16-rounded a and g
            const float av = bf2f(f2bf(pa[i][q][r] + ba[q][r])), gv = bf2f(f2bf(pg[i][q][r] + bg[q][r]));
            float cdf, pdf;
            normal_cdf_pdf(gv, cdf, pdf);
            const float ge = gv * cdf;                               // gelu_erf(gv)
            phh[i][q][r] = av * ge;
            if (p.geglu_save) {                                      // (wave-uniform) s = [gelu(g) | a gelu'(g)] leaves instead of u = [a | g]
              pa[i][q][r] = ge;
              pg[i][q][r] = av * (cdf + gv * pdf);
            }
          }
        }
      const int mb = m0 + wm * 128 + ib * 64;
      if (p.geglu_save) {
        reg_epilogue_64x32<0>(ps, p.C, pa, mb, abase, lane);        // s[:, a-columns] = gelu(g)
        reg_epilogue_64x32<0>(ps, p.C, pg, mb, I + abase, lane);    // s[:, I + a-columns] = a gelu'(g)
      } else {
        reg_epilogue_64x32<0>(p, p.C, pa, mb, abase, lane);         // u[:, a-columns]  (+ bias[abase ..])
        reg_epilogue_64x32<0>(p, p.C, pg, mb, I + abase, lane);     // u[:, I + a-columns]  (+ bias[I + abase ..])
      }
      reg_epilogue_64x32<0>(ph, p.geglu_h, phh, mb, abase, lane);
    }
    return;
  }
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4_t pair[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pair[i][0] = acc[ib * 4 + i][2 * half]; pair[i][1] = acc[ib * 4 + i][2 * half + 1]; }
      reg_epilogue_64x32<0>(p, p.C, pair, m0 + wm * 128 + ib * 64, n0 + wn * 64 + half * 32, lane);
    }
}

static bool use_xl(const NkGemmParams& p, int amode, int bmode, int out_f32, int splitk);
__global__ void nk_zero_f32_kernel(float* __restrict__ dst, size_t n);
#include "gemm_g2.h"
#include "gemm_w160.h"
#include "conv_halo.h"
#include "conv_wgrad_halo.h"

static bool use_xl(const NkGemmParams& p, int amode, int bmode, int out_f32, int splitk) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("NK_GEMM_XL"); on = (e && e[0] == '0') ? 0 : 1; }
  if (!on || p.nbatch || out_f32 || splitk != 1 || bmode != OP_KC || !(amode == OP_KC || amode == OP_KCG)) return false;
  const long ntn = (p.N + XL_BN - 1) / XL_BN;
  const long tiles = (long)((p.M + XL_BM - 1) / XL_BM) * ntn;
  // at least ~one workgroup per CU, and no more than 12 % of the last column tile wasted (N = 320 / 640 would idle 37 % / 17 %)
  // ... and rounds of 256 workgroups that are at least 80 % full (320 tiles would run as two rounds at 62 %: measured 775 vs 847 TFLOP/s
  // against the 128 x 128 kernel's finer rounds)
  const long rounds = (tiles + 255) / 256;
  return tiles >= 224 && tiles * 10 >= rounds * 256 * 8 && ntn * XL_BN * 100 <= (long)p.N * 112 && p.K >= 4 * BK;
}
template <int AMODE, int WM, int WN>
static int launch_xl_as(const NkGemmParams& p, hipStream_t stream) {
  auto kern = nk_gemm_xl_kernel<AMODE, WM, WN>;
  nk_optin_lds((const void*)kern, XL_SMEM_BYTES);
  dim3 grid(((p.M + XL_BM - 1) / XL_BM) * ((p.N + XL_BN - 1) / XL_BN), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3((XL_BM / WM) * (XL_BN / WN) * 64), XL_SMEM_BYTES, stream, p);
  return nk_check_launch("nk_gemm_xl_kernel");
}
template <int AMODE>
static int launch_xl(const NkGemmParams& p_in, hipStream_t stream) {
  // by operand mode: dense k-contiguous A -> the two-group phased kernel (+3..6 % on the Linear shapes); gathered A -> the 16-wave kernel
  // (the gather's address arithmetic would sit in the phased kernel's read phases, where only one wave per SIMD is there to absorb it:
  // conv forward 781-785 vs 835-840 TFLOP/s)
  if (AMODE == OP_KC) {
    auto kern = nk_gemm_xl2g_kernel<OP_KC, 0>;
    auto kerng = nk_gemm_xl2g_kernel<OP_KC, 1>;
    nk_optin_lds((const void*)kern, XL_SMEM_BYTES);
    nk_optin_lds((const void*)kerng, XL_SMEM_BYTES);
    NkGemmParams p = p_in;
    p.k_rotate = k_rotate_on(p.K) ? 1 : 0;
    dim3 grid(((p.M + XL_BM - 1) / XL_BM) * ((p.N + XL_BN - 1) / XL_BN), 1, 1);
    if (p.geglu_h) hipLaunchKernelGGL(kerng, grid, dim3(512), XL_SMEM_BYTES, stream, p);
    else hipLaunchKernelGGL(kern, grid, dim3(512), XL_SMEM_BYTES, stream, p);
    return nk_check_launch("nk_gemm_xl2g_kernel");
  }
  return launch_xl_as<AMODE, 64, 64>(p_in, stream);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int AMODE, int BMODE, int OUT_F32>
static int launch(const NkGemmParams& p_in, int splitk, hipStream_t stream) {
  NkGemmParams p2 = p_in;
  const NkGemmParams& p = p2;
  auto kern8 = nk_gemm_dma_kernel<AMODE, BMODE, OUT_F32, 8>;      // 8 waves per 128 x 128 tile: +4..14 % over 4 waves on every SDXL shape
  auto kernr = nk_gemm_ring_kernel<AMODE, BMODE, OUT_F32>;
  nk_optin_lds((const void*)kern8, V2_SMEM_BYTES);
  nk_optin_lds((const void*)kernr, RING_SMEM_BYTES);
  int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, splitk, p.nbatch ? p.nbatch : 1);
  {  // patch height: with T tiles over 8 XCDs an XCD runs ~T/8 tiles at a time; a gm x (T/8/gm) patch touches gm + T/8/gm operand
     // panels, least at gm = sqrt(T/8).  (A fixed 8 gave a 100-tile weight gradient 8 x 1.5 patches: 10 panels per XCD where 7 do.)
    int per_xcd = (ntm * ntn + 7) / 8, g = 1;
    while ((g + 1) * (g + 1) <= per_xcd) ++g;
    p2.group_m = g > GROUP_M ? GROUP_M : g;
  }
  p2.k_rotate = k_rotate_on(splitk > 1 ? p.ksplit_len : p.K) ? 1 : 0;      // (nk_gemm_dma_kernel; the ring kernel walks k in order)
  // under-filled grids (at most one workgroup per CU): the four-stage ring.  (The ring on LARGE grids was measured too: 723 vs 830 TFLOP/s
  // at 65536 x 1280 x 1280 -- three slabs in flight do not make up for two waves per SIMD meeting at a barrier every k-step.)
  if (!p.nbatch && (long)ntm * ntn * splitk <= 256) {
    hipLaunchKernelGGL(kernr, grid, dim3(512), RING_SMEM_BYTES, stream, p);
    return nk_check_launch("nk_gemm_ring_kernel");
  }
  hipLaunchKernelGGL(kern8, grid, dim3(512), V2_SMEM_BYTES, stream, p);
  return nk_check_launch("nk_gemm_dma_kernel");
}


// ---- stream-K workspace: one per stream (launches on one stream are ordered; two streams run concurrently) ----
#include <mutex>
#include <unordered_map>
struct SkWorkspace {
  unsigned* counter = nullptr;
  unsigned* flags = nullptr;
  float* ws = nullptr;
};
static std::mutex sk_mutex;
static std::unordered_map<void*, SkWorkspace> sk_spaces;

// NK_GEMM_SK: 0 = never, 1 = always, 2 = fp32 outputs (weight gradients) only, 3 = by shape, 4 (default) = by shape and
// bf16 outputs only (weight gradients run on the side stream, where non-persistent grids back-fill the main stream's
// kernels: measured 203.7 ms/step vs 206.3 without stream-K, 211 with it on every kernel).  By shape:
// stream-K where the data-parallel grid fills the chip badly or would need split-K, the plain kernel for big grids
// (measured 5-19 % faster there: its workgroups drift out of phase, the persistent ones load and store in lock-step).
static bool use_sk(const NkGemmParams& p, int out_f32) {
  int mode = 4;
  if (const char* e = getenv("NK_GEMM_SK")) mode = atoi(e);   // read per call: tools/ab_step.py flips it in-process
  if (mode == 0) return false;
  if (mode == 1) return true;
  if (mode == 2) return out_f32 != 0;
  if (mode == 4 && out_f32) return false;     // by shape, bf16 outputs (main-stream forward / dgrad) only
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.nbatch ? p.nbatch : 1);
  const long nk = (p.K + BK - 1) / BK;
  const long rounds = (tiles + 511) / 512;
  const double fill = (double)tiles / (double)(rounds * 512);
  if (tiles >= 1024 && fill >= 0.8) return false;
  if (tiles >= 512 && fill >= 0.95) return false;
  return nk >= 24;    // a fix-up costs about as much as 6-8 k-steps
}

static int sk_prepare(NkGemmParams& p, int grid, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(sk_mutex);
  SkWorkspace& w = sk_spaces[(void*)stream];
  if (!w.ws) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
      nk_set_error(__FILE__, __LINE__, "stream-K workspace requested while the stream is capturing: run the launch once on this stream before capturing it");
      return NK_ERR_LAUNCH;
    }
    char* raw = nullptr;
    const size_t ws_bytes = (size_t)SK_MAX_GRID * SK_TILE_FLOATS * sizeof(float);
    if (hipMalloc((void**)&raw, ws_bytes + (SK_MAX_GRID + 64) * sizeof(unsigned)) != hipSuccess) {
      nk_set_error(__FILE__, __LINE__, "hipMalloc of the stream-K workspace failed");
      return NK_ERR_LAUNCH;
    }
    w.ws = (float*)raw;
    w.flags = (unsigned*)(raw + ws_bytes);
    w.counter = w.flags + SK_MAX_GRID;         // flags[SK_MAX_GRID + 1] is the timeout mark
    // zeroed ON THE LAUNCHING STREAM (a null-stream memset is not ordered against a non-blocking stream's kernels)
    if (hipMemsetAsync(w.flags, 0, (SK_MAX_GRID + 64) * sizeof(unsigned), stream) != hipSuccess) return NK_ERR_LAUNCH;
  }
  p.sk_health = nk_health_word();
  if (!p.sk_health) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  p.sk_counter = w.counter;
  p.sk_flags = w.flags;
  p.sk_ws = w.ws;
  p.sk_base = 0;
  p.sk_epoch = 0;
  return NK_OK;
}

// called by nk_health_clear() (device already synchronised): lower every flag a failed launch may have left raised
int nk_gemm_sk_reset(void) {
  std::lock_guard<std::mutex> lock(sk_mutex);
  for (auto& kv : sk_spaces)
    if (kv.second.flags && hipMemset(kv.second.flags, 0, (SK_MAX_GRID + 64) * sizeof(unsigned)) != hipSuccess) return NK_ERR_LAUNCH;
  return NK_OK;
}

// 0 = every stream-K launch so far completed its fix-ups; 1 = some launch gave up waiting for a partial tile (its output
// is wrong).  Synchronises the device.
extern "C" int nk_gemm_sk_status(void) {
  std::lock_guard<std::mutex> lock(sk_mutex);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int bad = 0;
  for (auto& kv : sk_spaces) {
    unsigned v = 0;
    if (kv.second.flags && hipMemcpy(&v, kv.second.flags + SK_MAX_GRID + 1, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess && v) bad = 1;
  }
  return bad;
}

template <int AMODE, int BMODE, int OUT_F32>
static int launch_sk(NkGemmParams& p, hipStream_t stream) {
  auto kern = nk_gemm_sk_kernel<AMODE, BMODE, OUT_F32>;
  nk_optin_lds((const void*)kern, SK_SMEM_BYTES);
  const long ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nk = (p.K + BK - 1) / BK;
  const long T = ntm * ntn * (p.nbatch ? p.nbatch : 1), W = T * nk;
  // persistent grid: two workgroups per CU, at least ~4 k-steps each
  const int max_grid = SK_MAX_GRID & ~7, min_iters = 4;
  long grid = (W / min_iters) & ~7l;
  if (grid > max_grid) grid = max_grid;
  if (grid < 8) grid = 8;
  p.sk_chunked = T >= 64;
  { const char* d = getenv("NK_SK_DEBUG"); p.sk_debug = d ? atoi(d) & 2 : 0; }      // fault injection for tests/test_health_gpu.py
  if (int e = sk_prepare(p, (int)grid, stream)) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SK_NT), SK_SMEM_BYTES, stream, p);
  return nk_check_launch("nk_gemm_sk_kernel");
}

// zero-fill of a split-K destination (grid-stride, 16 B per lane; n need not be a multiple of 4)
__global__ __launch_bounds__(256) void nk_zero_f32_kernel(float* __restrict__ dst, size_t n) {
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  float4_t* d4 = (float4_t*)dst;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d4[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = 0.f;
}

static int pick_splitk(int M, int N, int K, int max_split) {
  // Split K only for grids far below one workgroup per CU.  Each extra split costs M*N*4 bytes of fp32 atomics at the
  // chip-wide ~1.3 TB/s atomic rate (MI355X_MICROARCH.md), which is 923/K_red of the GEMM's own time per split -- 22 %
  // per split at a 4096-row reduction -- while under-filled grids are back-filled by the kernels running concurrently
  // on the other stream (dgrad chain vs weight-gradient stream).
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int nk = (K + BK - 1) / BK;
  const int lo = 96, hi = 192;
  // ... except long reductions into grids that leave half the 512 slots empty (the 128^2- and 64^2-level convolution weight gradients:
  // 65 536 / 16 384 pixels = 1 024 / 256 k-steps into 135-225 tiles; the 64^2-level Linear ones): the atomics of one extra split are a few
  // per cent of such a launch and two splits fill the slots -- 65536 x 320 x 5760: 938 -> 587 us, x 8640: 972 -> 617; step -1 ms (round 3)
  if (tiles >= lo && tiles * 2 <= 512 && nk >= 256 && max_split >= 2) return 2;
  if (tiles >= lo) return 1;
  int s = 1;
  while (s < max_split && tiles * s < hi && nk / (s * 2) >= 8) s *= 2;
  return s;
}

static void set_split(NkGemmParams& p, int splitk) {
  int nk = (p.K + BK - 1) / BK;
  int per = (nk + splitk - 1) / splitk;
  p.ksplit_len = per * BK;
}

// split-K partials are summed with fp32 atomics, which need a zeroed destination (weight gradient and fused bias gradient)
// (a kernel, not hipMemsetAsync: as a node of a captured hipGraph the memset of a multi-MB buffer was not ordered before the
// kernel behind it on this ROCm -- replayed weight gradients of the 320/640-channel layers came out as garbage)
static int zero_split_outputs(const NkGemmParams& p, hipStream_t stream) {
  for (int z = 0; z < (p.nbatch ? p.nbatch : 1); ++z) {
    const size_t n = (size_t)p.M * p.N;
    float* dst = (float*)(p.nbatch ? p.Cb[z] : p.C);
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256);
    hipLaunchKernelGGL(nk_zero_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, dst, n);
    float* db = p.nbatch ? p.dbias_b[z] : p.dbias;        // the fused bias gradient is summed by the same atomics
    if (db) hipLaunchKernelGGL(nk_zero_f32_kernel, dim3(1), dim3(256), 0, stream, db, (size_t)p.M);
  }
  return hipGetLastError() != hipSuccess ? NK_ERR_LAUNCH : NK_OK;
}

int nk_gemm_dispatch(NkGemmParams& p, int amode, int bmode, int out_f32, int allow_splitk,
                     hipStream_t stream) {
  NK_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0);
  if (!p.nbatch) NK_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0);
  for (int z = 0; z < p.nbatch; ++z) NK_CHECK_ARG(((uintptr_t)p.Ab[z] & 15) == 0 && ((uintptr_t)p.Bb[z] & 15) == 0 && ((uintptr_t)p.Cb[z] & 15) == 0);
  // 16-byte chunk rules: k-contiguous operands need K % 8 == 0 and ld % 8 == 0; r-contiguous ones R % 8 == 0
  if (amode == OP_KC) NK_CHECK_ARG((p.K & 7) == 0 && (p.lda & 7) == 0);
  if (bmode == OP_KC) NK_CHECK_ARG((p.K & 7) == 0 && (p.ldb & 7) == 0);
  if (amode == OP_KCG) NK_CHECK_ARG((p.ga.C & 7) == 0);
  if (amode == OP_MC) NK_CHECK_ARG((p.M & 7) == 0 && (p.lda & 7) == 0);
  if (bmode == OP_MC) NK_CHECK_ARG((p.N & 7) == 0 && (p.ldb & 7) == 0);
  if (bmode == OP_MCT) NK_CHECK_ARG((p.N & 7) == 0 && (p.tw.co_stride & 7) == 0 && (p.tw.tap_stride & 7) == 0);
  if (bmode == OP_MCG) NK_CHECK_ARG((p.gb.C & 7) == 0);
  if (p.residual) NK_CHECK_ARG(((uintptr_t)p.residual & 15) == 0 && ((p.ldr & 7) == 0 || (p.N & 7) != 0));
  if (!out_f32) NK_CHECK_ARG((p.ldc & 7) == 0 || (p.N & 7) != 0);
  if (p.fRowsPerBatch.d == 0) p.fRowsPerBatch = make_fastdiv(1);

  if (p.dbias || (p.nbatch && p.dbias_b[0])) NK_CHECK_ARG(amode == OP_MC && out_f32);      // weight-gradient launches only
  if (p.geglu_h) {   // the fused GEGLU forward lives in the 256 x 256 two-group kernel only: the caller asks nk_linear_fwd_geglu_ok first
    NK_CHECK_ARG(amode == OP_KC && bmode == OP_KC && !out_f32 && !p.nbatch && (p.N & 255) == 0 && (p.ld_h & 7) == 0 && !p.rowvec && !p.residual && p.alpha == 1.0f);
    NK_CHECK_ARG(((uintptr_t)p.geglu_h & 15) == 0 && (!p.bias || ((uintptr_t)p.bias & 15) == 0));
    if (!use_xl(p, amode, bmode, out_f32, 1)) {
      nk_set_error(__FILE__, __LINE__, "fused GEGLU forward on a shape the 256 x 256 kernel does not take (ask nk_linear_fwd_geglu_ok first)");
      return NK_ERR_ARG;
    }
    set_split(p, 1);
    return launch_xl<OP_KC>(p, stream);
  }
  if (p.geglu_u) {   // the fused GEGLU backward lives in the LDS-staged epilogue of the 128 x 128 data-parallel / ring kernels only
    NK_CHECK_ARG(amode == OP_KC && bmode == OP_MC && !out_f32 && !p.nbatch && (p.N & 7) == 0 && (p.ld_u & 7) == 0 && !p.bias && !p.rowvec && !p.residual);
    set_split(p, 1);
    return launch<OP_KC, OP_MC, 0>(p, 1, stream);
  }
  // 3 x 3 / stride 1 / padding 1 convolutions over whole 64-channel slabs: the halo-tile kernel (conv_halo.h)
  if (use_halo(p, amode, bmode, out_f32)) return launch_halo(p, stream);
  // ... and their weight gradients: nine taps from one staged halo per pixel tile (conv_wgrad_halo.h)
  if (use_wgrad_halo(p, amode, bmode, out_f32)) return launch_wgrad_halo(p, stream);
  if (p.stats_part) {      // the GroupNorm statistics epilogue exists in the halo-tile kernel only
    nk_set_error(__FILE__, __LINE__, "statistics epilogue on a convolution the halo-tile kernel does not take (ask nk_conv2d_stats_tiles first)");
    return NK_ERR_ARG;
  }

  // Linear weight gradients whose 160-row tiles come out in whole rounds of 256 workgroups (gemm_w160.h), the token range split where the
  // weight alone is too small a grid (fp32 atomics: zeroed destination, as below)
  {
    const W160Plan wp = w160_plan(p, amode, bmode, out_f32, allow_splitk);
    if (wp.bn) {
      set_split(p, wp.splitk);
      if (wp.splitk > 1 && p.accumulate == 0) {
        NK_CHECK_ARG(p.ldc == p.N);
        if (int rc = zero_split_outputs(p, stream)) return rc;
        p.accumulate = 1;
      } else if (p.accumulate == 2) {
        p.accumulate = wp.splitk > 1 ? 1 : 0;
      }
      p.k_rotate = k_rotate_on(wp.splitk > 1 ? p.ksplit_len : p.K) ? 1 : 0;
      return wp.bn == 160 ? launch_w160_as<160>(p, wp.splitk, stream) : launch_w160_as<128>(p, wp.splitk, stream);
    }
  }
  // two-group staggered ring at one workgroup per CU (gemm_g2.h): Linear forward / dgrad / wgrad shapes whose 128 x 160 (or
  // 128 x 128) tiles come out in whole rounds of 256
  if (use_ring64(p, amode, bmode, out_f32)) return launch_ring64(p, stream);
  if ((!p.nbatch || p.nbatch <= NK_MAX_BATCH) && use_g2(p, amode, bmode, out_f32, 1) && (g2_mode() == 2 || !use_xl(p, amode, bmode, out_f32, 1))) {
    if (p.accumulate == 2) p.accumulate = 0;       // no K split here: "destination known zero" means plain stores
    if (amode == OP_KC && bmode == OP_KC) return out_f32 ? launch_g2<OP_KC, OP_KC, 1>(p, stream) : launch_g2<OP_KC, OP_KC, 0>(p, stream);
    if (amode == OP_KC && bmode == OP_MC) return out_f32 ? launch_g2<OP_KC, OP_MC, 1>(p, stream) : launch_g2<OP_KC, OP_MC, 0>(p, stream);
    if (amode == OP_KCG && bmode == OP_KC) return out_f32 ? launch_g2<OP_KCG, OP_KC, 1>(p, stream) : launch_g2<OP_KCG, OP_KC, 0>(p, stream);
    if (amode == OP_KCG && bmode == OP_MCT) return out_f32 ? launch_g2<OP_KCG, OP_MCT, 1>(p, stream) : launch_g2<OP_KCG, OP_MCT, 0>(p, stream);
    return out_f32 ? launch_g2<OP_MC, OP_MC, 1>(p, stream) : launch_g2<OP_MC, OP_MC, 0>(p, stream);
  }

  // (a launch that carries a fused bias gradient never goes to stream-K, whatever NK_GEMM_SK says: that kernel has no ones-MFMA row sum,
  // and the gradient would silently stay unwritten)
  bool has_dbias = p.dbias != nullptr;
  for (int z = 0; z < p.nbatch && z < NK_MAX_BATCH; ++z) has_dbias = has_dbias || p.dbias_b[z] != nullptr;
  if (!has_dbias && use_sk(p, out_f32)) {
    const long ntm_ = (p.M + BM - 1) / BM, ntn_ = (p.N + BN - 1) / BN, nk_ = (p.K + BK - 1) / BK;
    if (ntm_ * ntn_ * (p.nbatch ? p.nbatch : 1) * nk_ < (1l << 22)) {   // share arithmetic is 32-bit: W * grid < 2^31
      if (p.accumulate == 2) p.accumulate = 0;     // "destination known zero" only matters to the atomic split-K path
#define NK_SK_CASE(A_, B_)                                                        \
      if (amode == A_ && bmode == B_) return out_f32 ? launch_sk<A_, B_, 1>(p, stream) : launch_sk<A_, B_, 0>(p, stream);
      NK_SK_CASE(OP_KC, OP_KC)
      NK_SK_CASE(OP_KC, OP_MC)
      NK_SK_CASE(OP_MC, OP_MC)
      NK_SK_CASE(OP_KCG, OP_KC)
      NK_SK_CASE(OP_KCG, OP_MCT)
      NK_SK_CASE(OP_MC, OP_MCG)
#undef NK_SK_CASE
    }
  }
  int splitk = 1;
  if (out_f32 && allow_splitk) splitk = pick_splitk(p.M * (p.nbatch ? p.nbatch : 1), p.N, p.K, 32);
  if (p.nbatch) NK_CHECK_ARG(p.nbatch <= NK_MAX_BATCH);
  set_split(p, splitk);
  // accumulate: 0 = overwrite, 1 = add, 2 = the destination is known to be zero (flat gradient buffer right after
  // zero_grad): plain stores when there is a single K split, atomics otherwise -- and no memset either way
  if (out_f32 && splitk > 1 && p.accumulate == 0) {
    // split-K partials are summed with fp32 atomics, which need a zeroed destination
    NK_CHECK_ARG(p.ldc == p.N);
    if (int rc = zero_split_outputs(p, stream)) return rc;
    p.accumulate = 1;
  } else if (p.accumulate == 2) {
    p.accumulate = splitk > 1 ? 1 : 0;
  }

  if (use_xl(p, amode, bmode, out_f32, splitk))
    return amode == OP_KC ? launch_xl<OP_KC>(p, stream) : launch_xl<OP_KCG>(p, stream);

#define NK_CASE(A_, B_)                                                          \
  if (amode == A_ && bmode == B_) {                                              \
    return out_f32 ? launch<A_, B_, 1>(p, splitk, stream) : launch<A_, B_, 0>(p, splitk, stream); \
  }
  NK_CASE(OP_KC, OP_KC)
  NK_CASE(OP_KC, OP_MC)
  NK_CASE(OP_MC, OP_MC)
  NK_CASE(OP_KCG, OP_KC)
  NK_CASE(OP_KCG, OP_MCT)
  NK_CASE(OP_MC, OP_MCG)
#undef NK_CASE
  nk_set_error(__FILE__, __LINE__, "unsupported operand mode combination");
  return NK_ERR_ARG;
}

// pixel tiles per image of the halo-tile launch this convolution would get (= rows per image of the statistics epilogue's partials);
// 0 when it is not eligible (shape, NK_CONV_HALO=0)
int nk_halo_bn(int N) { return halo_bn(N); }
int nk_halo_tiles_per_image(const NkGemmParams& p) {
  if (!use_halo(p, OP_KCG, OP_KC, 0)) return 0;
  return halo_tiles_per_image(p);
}


// 1 when nk_linear_fwd_geglu takes this FeedForward projection (the 256 x 256 two-group kernel's shape rule, whole 256-column tiles)
int nk_geglu_fwd_fusable(const NkGemmParams& p) {
  return (p.N & 255) == 0 && use_xl(p, OP_KC, OP_KC, 0, 1) ? 1 : 0;
}
