// HBM-bound elementwise / data-movement kernels of the SDXL training step (gfx950).
// All bf16 traffic is 16 B per lane; fp32 traffic 16 B per lane where the layout allows.
#include "../../include/neurosis_hip.h"
#include "nk_common.h"

#define EW_THREADS 256
static inline int ew_blocks(long work_items) {
  long b = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (b > 8192) b = 8192;  // grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

// ---- GEGLU (modules/attention.py:55-57): y = u[:, :I] * gelu(u[:, I:]) ------------------------
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long M, int I) {
  const int cpr = I >> 3;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / cpr;
    int ch = (int)(i - row * cpr);
    float a[8], g[8];
    unpack8(*(const uint4_t*)(u + row * 2 * I + ch * 8), a);
    unpack8(*(const uint4_t*)(u + row * 2 * I + I + ch * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= gelu_erf(g[e]);
    *(uint4_t*)(y + row * I + ch * 8) = pack8(a);
  }
}
__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ du,
                                 long M, int I) {
  const int cpr = I >> 3;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / cpr;
    int ch = (int)(i - row * cpr);
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8(*(const uint4_t*)(u + row * 2 * I + ch * 8), a);
    unpack8(*(const uint4_t*)(u + row * 2 * I + I + ch * 8), g);
    unpack8(*(const uint4_t*)(dy + row * I + ch * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float cdf, pdf;
      normal_cdf_pdf(g[e], cdf, pdf);
      da[e] = d[e] * (g[e] * cdf);
      dg[e] = d[e] * a[e] * (cdf + g[e] * pdf);
    }
    *(uint4_t*)(du + row * 2 * I + ch * 8) = pack8(da);
    *(uint4_t*)(du + row * 2 * I + I + ch * 8) = pack8(dg);
  }
}
// saved-derivative form (round 6): y = a gelu(g) and s = [gelu(g) | a gelu'(g)] from u = [a | g] (s may BE u: every thread rewrites the two
// chunks it read); the backward is then two products per element
__global__ void geglu_fwd_s_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, bf16_t* s, long M, int I) {
  const int cpr = I >> 3;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / cpr;
    int ch = (int)(i - row * cpr);
    float a[8], g[8], h[8], s1[8], s2[8];
    unpack8(*(const uint4_t*)(u + row * 2 * I + ch * 8), a);
    unpack8(*(const uint4_t*)(u + row * 2 * I + I + ch * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float cdf, pdf;
      normal_cdf_pdf(g[e], cdf, pdf);
      s1[e] = g[e] * cdf;
      h[e] = a[e] * s1[e];
      s2[e] = a[e] * (cdf + g[e] * pdf);
    }
    *(uint4_t*)(y + row * I + ch * 8) = pack8(h);
    *(uint4_t*)(s + row * 2 * I + ch * 8) = pack8(s1);
    *(uint4_t*)(s + row * 2 * I + I + ch * 8) = pack8(s2);
  }
}
__global__ void geglu_bwd_s_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s, bf16_t* __restrict__ du, long M, int I) {
  const int cpr = I >> 3;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / cpr;
    int ch = (int)(i - row * cpr);
    float s1[8], s2[8], d[8];
    unpack8(*(const uint4_t*)(s + row * 2 * I + ch * 8), s1);
    unpack8(*(const uint4_t*)(s + row * 2 * I + I + ch * 8), s2);
    unpack8(*(const uint4_t*)(dy + row * I + ch * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] *= d[e]; s2[e] *= d[e]; }
    *(uint4_t*)(du + row * 2 * I + ch * 8) = pack8(s1);
    *(uint4_t*)(du + row * 2 * I + I + ch * 8) = pack8(s2);
  }
}
extern "C" int nk_geglu_fwd_s(const void* u, void* y, void* s, long M, int I, void* stream) {
  NK_CHECK_ARG(u && y && s && M > 0 && I > 0 && (I & 7) == 0);
  hipLaunchKernelGGL(geglu_fwd_s_kernel, dim3(ew_blocks(M * (I >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)u, (bf16_t*)y,
                     (bf16_t*)s, M, I);
  return nk_check_launch("geglu_fwd_s");
}
extern "C" int nk_geglu_bwd_s(const void* dy, const void* s, void* du, long M, int I, void* stream) {
  NK_CHECK_ARG(dy && s && du && M > 0 && I > 0 && (I & 7) == 0);
  hipLaunchKernelGGL(geglu_bwd_s_kernel, dim3(ew_blocks(M * (I >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)s, (bf16_t*)du, M, I);
  return nk_check_launch("geglu_bwd_s");
}
extern "C" int nk_geglu_fwd(const void* u, void* y, long M, int I, void* stream) {
  NK_CHECK_ARG(u && y && M > 0 && I > 0 && (I & 7) == 0);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_blocks(M * (I >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)u, (bf16_t*)y, M, I);
  return nk_check_launch("geglu_fwd");
}
extern "C" int nk_geglu_bwd(const void* dy, const void* u, void* du, long M, int I, void* stream) {
  NK_CHECK_ARG(dy && u && du && M > 0 && I > 0 && (I & 7) == 0);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_blocks(M * (I >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du, M, I);
  return nk_check_launch("geglu_bwd");
}

// ---- SiLU on a flat bf16 array (openaimodel.py:274, 588, 615) ----------------------------------
__global__ void silu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*(const uint4_t*)(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
    *(uint4_t*)(y + i * 8) = pack8(f);
  }
}
__global__ void silu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx,
                                long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], d[8];
    unpack8(*(const uint4_t*)(x + i * 8), f);
    unpack8(*(const uint4_t*)(dy + i * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= dsilu_f(f[e]);
    *(uint4_t*)(dx + i * 8) = pack8(d);
  }
}
extern "C" int nk_silu_fwd(const void* x, void* y, long n, void* stream) {
  NK_CHECK_ARG(x && y && n > 0 && (n & 7) == 0);
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(ew_blocks(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, n >> 3);
  return nk_check_launch("silu_fwd");
}
extern "C" int nk_silu_bwd(const void* dy, const void* x, void* dx, long n, void* stream) {
  NK_CHECK_ARG(dy && x && dx && n > 0 && (n & 7) == 0);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_blocks(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, n >> 3);
  return nk_check_launch("silu_bwd");
}

// ---- GELU of the CLIP text transformers (forward only: the encoders are frozen) -------------------
// mode 0: exact erf form (open_clip's nn.GELU); mode 1: quick_gelu x * sigmoid(1.702 x) (openai/clip-vit-large-patch14)
template <int MODE>
__global__ void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*(const uint4_t*)(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      f[e] = MODE == 0 ? 0.5f * f[e] * (1.0f + erff(f[e] * 0.70710678118654752f)) : f[e] / (1.0f + __expf(-1.702f * f[e]));
    *(uint4_t*)(y + i * 8) = pack8(f);
  }
}
extern "C" int nk_gelu_fwd(const void* x, void* y, long n, int mode, void* stream) {
  NK_CHECK_ARG(x && y && n > 0 && (n & 7) == 0 && (mode == 0 || mode == 1));
  if (mode == 0)
    hipLaunchKernelGGL(gelu_fwd_kernel<0>, dim3(ew_blocks(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n >> 3);
  else
    hipLaunchKernelGGL(gelu_fwd_kernel<1>, dim3(ew_blocks(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n >> 3);
  return nk_check_launch("gelu_fwd");
}

// ---- out = a + b (gradient join where a tensor feeds two consumers) ----------------------------
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(*(const uint4_t*)(a + i * 8), f);
    unpack8(*(const uint4_t*)(b + i * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
    *(uint4_t*)(o + i * 8) = pack8(f);
  }
}
extern "C" int nk_add(const void* a, const void* b, void* out, long n, void* stream) {
  NK_CHECK_ARG(a && b && out && n > 0 && (n & 7) == 0);
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)out, n >> 3);
  return nk_check_launch("add");
}

// ---- channel concat / split on channels-last rows (torch.cat(dim=1), openaimodel.py:836) -------
// dir 0: out[row] = [a[row] | b[row]] ; dir 1: a[row], b[row] = split(out[row]) (either may be NULL)
__global__ void cat_kernel(bf16_t* __restrict__ a, bf16_t* __restrict__ b, bf16_t* __restrict__ o, long rows, int Ca,
                           int Cb, int dir) {
  const int cpr = (Ca + Cb) >> 3, ca8 = Ca >> 3;
  const long total = rows * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / cpr;
    int ch = (int)(i - row * cpr);
    bf16_t* side = ch < ca8 ? (a ? a + row * Ca + ch * 8 : nullptr) : (b ? b + row * Cb + (ch - ca8) * 8 : nullptr);
    if (!side) continue;
    uint4_t* op = (uint4_t*)(o + i * 8);
    if (dir == 0) *op = *(const uint4_t*)side;
    else *(uint4_t*)side = *op;
  }
}
extern "C" int nk_cat_channels(const void* a, const void* b, void* out, long rows, int Ca, int Cb, void* stream) {
  NK_CHECK_ARG(a && b && out && rows > 0 && Ca > 0 && Cb > 0 && (Ca & 7) == 0 && (Cb & 7) == 0);
  hipLaunchKernelGGL(cat_kernel, dim3(ew_blocks(rows * ((Ca + Cb) >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (bf16_t*)a, (bf16_t*)b, (bf16_t*)out, rows, Ca, Cb, 0);
  return nk_check_launch("cat_channels");
}
extern "C" int nk_split_channels(const void* src, void* a, void* b, long rows, int Ca, int Cb, void* stream) {
  NK_CHECK_ARG(src && (a || b) && rows > 0 && Ca > 0 && Cb > 0 && (Ca & 7) == 0 && (Cb & 7) == 0);
  hipLaunchKernelGGL(cat_kernel, dim3(ew_blocks(rows * ((Ca + Cb) >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (bf16_t*)a, (bf16_t*)b, (bf16_t*)src, rows, Ca, Cb, 1);
  return nk_check_launch("split_channels");
}

// ---- backward of nearest 2x upsampling: dx[n,h,w,:] = sum of the 2x2 block of dup --------------
__global__ void up2_bwd_kernel(const bf16_t* __restrict__ dup, bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  const int cpr = C >> 3;
  const long total = (long)N * H * W * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ch = (int)(i % cpr);
    long pix = i / cpr;
    int w = (int)(pix % W);
    long t = pix / W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    const bf16_t* s = dup + (((long)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + ch * 8;
    float acc[8], f[8];
    unpack8(*(const uint4_t*)s, acc);
    unpack8(*(const uint4_t*)(s + C), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
    unpack8(*(const uint4_t*)(s + (long)2 * W * C), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
    unpack8(*(const uint4_t*)(s + (long)2 * W * C + C), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
    *(uint4_t*)(dx + i * 8) = pack8(acc);
  }
}
extern "C" int nk_upsample2x_bwd(const void* dup, void* dx, int N, int H, int W, int C, void* stream) {
  NK_CHECK_ARG(dup && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0);
  hipLaunchKernelGGL(up2_bwd_kernel, dim3(ew_blocks((long)N * H * W * (C >> 3))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, (const bf16_t*)dup, (bf16_t*)dx, N, H, W, C);
  return nk_check_launch("upsample2x_bwd");
}

// ---- layout / dtype boundary: NCHW (fp32 or bf16) <-> channels-last bf16 with channel padding ---
// tile of 32 pixels x 32 channels through LDS
template <typename SrcT>
__global__ void nchw_to_nhwc_kernel(const SrcT* __restrict__ src, bf16_t* __restrict__ dst, int C, int HW, int Cpad,
                                    float scale) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    float v = 0.f;
    if (c < C && p < HW) {
      if constexpr (sizeof(SrcT) == 4) v = src[((long)n * C + c) * HW + p];
      else v = bf2f(src[((long)n * C + c) * HW + p]);
    }
    tile[j][tx] = v * scale;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    if (p < HW && c < Cpad) dst[((long)n * HW + p) * Cpad + c] = f2bf(tile[tx][j]);
  }
}
template <typename DstT>
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ src, DstT* __restrict__ dst, int C, int HW, int Cpad) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (p < HW && c < C) ? bf2f(src[((long)n * HW + p) * Cpad + c]) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) {
      float v = tile[tx][j];
      if constexpr (sizeof(DstT) == 4) dst[((long)n * C + c) * HW + p] = v;
      else dst[((long)n * C + c) * HW + p] = f2bf(v);
    }
  }
}
// ------------------------------------------------------------------------------------------------
// conv weight [Cout][KH*KW][Cin] -> [Cin][KH*KW flipped][Cout]: the weights of the convolution that IS the input gradient of a
// stride-1 "same" convolution (dx = conv(dy, wt), taps mirrored, channel roles swapped).  With them the 3 x 3 input gradients run on
// the forward halo-tile kernel (conv_halo.h) instead of the transposed-operand gather.  One 64 x 64 (co, ci) tile of one tap per
// block, transposed through LDS; 16-byte loads along ci, 16-byte stores along co.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_weight_flip_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int Cout, int Cin, int taps) {
  __shared__ bf16_t tile[64][64 + 8];          // [ci][co], rows padded by 16 B
  const int tap = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + 256 * it;            // 64 rows (co) x 8 chunks (ci)
    const int r = idx >> 3, ch = idx & 7;
    uint4_t v = {0u, 0u, 0u, 0u};
    if (co0 + r < Cout && ci0 + ch * 8 < Cin) v = *(const uint4_t*)(w + ((long)(co0 + r) * taps + tap) * Cin + ci0 + ch * 8);
    const bf16_t* e = (const bf16_t*)&v;
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[ch * 8 + k][r] = e[k];
  }
  __syncthreads();
  const int tflip = taps - 1 - tap;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + 256 * it;            // 64 rows (ci) x 8 chunks (co)
    const int r = idx >> 3, ch = idx & 7;
    if (ci0 + r < Cin && co0 + ch * 8 < Cout)
      *(uint4_t*)(wt + ((long)(ci0 + r) * taps + tflip) * Cout + co0 + ch * 8) = *(const uint4_t*)&tile[r][ch * 8];
  }
}
extern "C" int nk_conv_weight_flip(const void* w, void* wt, int Cout, int Cin, int taps, void* stream) {
  NK_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && taps > 0 && taps <= 65535);
  NK_CHECK_ARG((Cout & 7) == 0 && (Cin & 7) == 0);
  hipLaunchKernelGGL(conv_weight_flip_kernel, dim3((Cin + 63) / 64, (Cout + 63) / 64, taps), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)wt,
                     Cout, Cin, taps);
  return nk_check_launch("conv_weight_flip_kernel");
}

// ------------------------------------------------------------------------------------------------
// 3 x 3 / stride 1 / padding 1 convolution of an image with 3 or 4 REAL channels (stored padded to 8): the VAE's conv_in
// (modules/diffusion/model.py:519, 3 -> 128 at 1024^2) and the UNet's (openaimodel.py:622-624, 4 -> 320).  As an implicit GEMM with K = 72
// the gather kernel spent 1.5 ms decoding taps for a 1 GB output, and a plain FMA kernel (a thread = 4 output channels, 108 FMAs and 9 unpacked
// 16-byte loads per pixel) took 3.65 ms: VALU-bound.  Here K = (tap, 4 channels) = 36, padded to 64 = two v_mfma_f32_16x16x32_bf16 steps:
// a wave owns 128 output channels (their 16 weight fragments live in registers for the whole launch) and walks groups of 16 pixels; a lane
// fetches the two or three 8-byte (4-channel) taps its k-group covers straight from HBM/L2 into the A fragment -- no LDS, no unpacking --
// and the 16 x 128 outputs leave through the tile engine's permlane16_swap epilogue as 16-byte stores.  HBM-bound (the output).
// (Channel 3 of a 3-channel image is zero padding in x and in the channel-padded weights, so one kernel serves both.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_few_channels_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                                   bf16_t* __restrict__ y, int N, int H, int W, int Cout, int groups_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int co_base = blockIdx.y * 128;
  // weight fragments: block b = output channels co_base + 16 b + r; k-step 0 holds taps 2g, 2g + 1; k-step 1 holds tap 8 (k-group 0 only)
  bf16x8_t wf[8][2];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int co = co_base + b * 16 + r;
    uint2_t t0 = {0u, 0u}, t1 = {0u, 0u}, t8 = {0u, 0u};
    if (co < Cout) {
      t0 = *(const uint2_t*)(w + ((long)co * 9 + 2 * g) * 8);
      t1 = *(const uint2_t*)(w + ((long)co * 9 + 2 * g + 1) * 8);
      if (g == 0) t8 = *(const uint2_t*)(w + ((long)co * 9 + 8) * 8);
    }
    wf[b][0] = __builtin_bit_cast(bf16x8_t, (uint4_t){t0.x, t0.y, t1.x, t1.y});
    wf[b][1] = __builtin_bit_cast(bf16x8_t, (uint4_t){t8.x, t8.y, 0u, 0u});
  }
  // after the epilogue's row swap lane group g holds channels (g & 1) * 16 + (g >> 1) * 8 .. + 7 of each 32-channel pair
  const int nloc = (g & 1) * 16 + (g >> 1) * 8;
  float bv[4][8];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = co_base + h * 32 + nloc + e;
      bv[h][e] = (bias && co < Cout) ? bias[co] : 0.f;
    }
  const long total = (long)N * H * W;
  const long gw = (long)blockIdx.x * 4 + wave;                 // this wave's first group of 16 pixels
  const int HW = H * W;
  // tap offsets of this lane's k-group
  const int ta = 2 * g, tb = 2 * g + 1;
  const int dya = ta / 3 - 1, dxa = ta % 3 - 1, dyb = tb / 3 - 1, dxb = tb % 3 - 1;

  auto fetch = [&](long grp, uint2_t& a, uint2_t& b, uint2_t& c) {
    a = (uint2_t){0u, 0u}; b = a; c = a;
    const long pix = grp * 16 + r;
    if (pix >= total) return;
    const int n = (int)(pix / HW);
    const int rem = (int)(pix - (long)n * HW);
    const int py = rem / W, px = rem - py * W;
    const bf16_t* img = x + (long)n * HW * 8;
    int yy = py + dya, xx = px + dxa;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) a = *(const uint2_t*)(img + ((long)yy * W + xx) * 8);
    yy = py + dyb; xx = px + dxb;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) b = *(const uint2_t*)(img + ((long)yy * W + xx) * 8);
    if (g == 0) {
      yy = py + 1; xx = px + 1;
      if (yy < H && xx < W) c = *(const uint2_t*)(img + ((long)yy * W + xx) * 8);
    }
  };

  const long gstride = (long)gridDim.x * 4;
  uint2_t na, nb, nc;
  long grp = gw;
  fetch(grp, na, nb, nc);
  for (int it = 0; it < groups_per_wave; ++it, grp += gstride) {
    if (grp * 16 >= total) break;
    const bf16x8_t x0 = __builtin_bit_cast(bf16x8_t, (uint4_t){na.x, na.y, nb.x, nb.y});
    const bf16x8_t x1 = __builtin_bit_cast(bf16x8_t, (uint4_t){nc.x, nc.y, 0u, 0u});
    if (it + 1 < groups_per_wave) fetch(grp + gstride, na, nb, nc);       // the next group's taps fly during this group's MFMAs and stores
    float4_t acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][0], x0, (float4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][1], x1, acc[b], 0, 0, 0);
    }
    // acc[b][e] = y[pixel r][channel co_base + 16 b + 4 g + e]
    const long pix = grp * 16 + r;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a_own = acc[2 * h][e], b_own = acc[2 * h + 1][e];
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a_own), __float_as_uint(b_own), false, false);
        v[e] = __uint_as_float(sw[0]) + bv[h][e];
        v[4 + e] = __uint_as_float(sw[1]) + bv[h][4 + e];
      }
      const int co = co_base + h * 32 + nloc;
      if (pix < total && co < Cout) *(uint4_t*)(y + pix * Cout + co) = pack8(v);
    }
  }
}
extern "C" int nk_conv3x3_few_channels_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cout, int cin_real,
                                           void* stream) {
  // x [N][H][W][8] bf16 (channels >= cin_real are zero padding), w [Cout][3][3][8] bf16 (likewise), y [N][H][W][Cout] bf16, bias [Cout] fp32 or NULL
  NK_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0 && Cout > 0 && (Cout & 7) == 0);
  NK_CHECK_ARG(cin_real == 3 || cin_real == 4);
  NK_CHECK_ARG((long)N * H * W < (1l << 31));
  const long groups = ((long)N * H * W + 15) / 16;
  // a wave keeps 16 weight fragments in registers: at least 32 pixel groups per wave to amortise loading them, ~8 workgroups per CU
  long blocks = (groups + 4 * 32 - 1) / (4 * 32);
  if (blocks > 2048) blocks = 2048;
  const int per_wave = (int)((groups + blocks * 4 - 1) / (blocks * 4));
  dim3 grid((unsigned)blocks, (unsigned)((Cout + 127) / 128));
  hipLaunchKernelGGL(conv3x3_few_channels_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, N, H, W, Cout,
                     per_wave);
  return nk_check_launch("conv3x3_few_channels_kernel");
}

extern "C" int nk_nchw_to_nhwc(const void* src, int src_is_f32, void* dst, int N, int C, int HW, int Cpad, float scale,
                               void* stream) {
  NK_CHECK_ARG(src && dst && N > 0 && C > 0 && HW > 0 && Cpad >= C);
  dim3 grid((HW + 31) / 32, (Cpad + 31) / 32, N);
  if (src_is_f32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src,
                       (bf16_t*)dst, C, HW, Cpad, scale);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (bf16_t*)dst, C, HW, Cpad, scale);
  return nk_check_launch("nchw_to_nhwc");
}
extern "C" int nk_nhwc_to_nchw(const void* src, void* dst, int dst_is_f32, int N, int C, int HW, int Cpad,
                               void* stream) {
  NK_CHECK_ARG(src && dst && N > 0 && C > 0 && HW > 0 && Cpad >= C);
  dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
  if (dst_is_f32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (float*)dst, C, HW, Cpad);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (bf16_t*)dst, C, HW, Cpad);
  return nk_check_launch("nhwc_to_nchw");
}

// ---- casts ----------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  const long n8 = n >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4_t a = *(const float4_t*)(s + i * 8), b = *(const float4_t*)(s + i * 8 + 4);
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(uint4_t*)(d + i * 8) = pack8(f);
  }
  if (blockIdx.x == 0) for (long i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) d[i] = f2bf(s[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = bf2f(s[i]);
}
extern "C" int nk_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  NK_CHECK_ARG(src && dst && n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_blocks((n >> 3) + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, n);
  return nk_check_launch("cast_f32_to_bf16");
}
extern "C" int nk_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream) {
  NK_CHECK_ARG(src && dst && n > 0);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)src, dst, n);
  return nk_check_launch("cast_bf16_to_f32");
}

// ---- bias gradient: out[n] (+)= sum_m dy[m][n] ---------------------------------------------------
// stage 1: per-block partial column sums part[split][N] (no atomics); stage 2: single-writer reduce over splits.
// A block is 16 row lanes x 16 chunk lanes (a chunk = 8 channels = 16 B): every row lane reads 256 contiguous bytes, a
// thread keeps four rows' loads in flight, and with 64 rows per split even M = 4096 gives hundreds of blocks (the
// earlier one-row-at-a-time layout ran at 1-2 TB/s: 12 ms of GPU time per step for bias gradients alone).
#define CS_ROWS 64
// rows per workgroup: 64 up to 65 536 rows (the UNet's token counts), then as many as keeps the second stage at <= 1 024 partial rows
// (the autoencoder's 2 M-row feature maps gave it 32 768 rows to walk: 80 us per bias gradient)
static int colsum_rows(long M) {
  long rows = CS_ROWS;
  while ((M + rows - 1) / rows > 1024) rows *= 2;
  return (int)rows;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ dy, float* __restrict__ part, long M, int N,
                                                             long ld, int rows_per_block) {
  __shared__ float ps[16][16 * 8 + 4];  // [row lane][chunk lane * 8 + e], combined in a fixed order (bitwise reproducible)
  const int tid = threadIdx.x;
  const int cx = tid & 15, ry = tid >> 4;
  const int chunk = blockIdx.y * 16 + cx;
  const bool cok = chunk < (N >> 3);
  dy += (long)blockIdx.z * M * ld;                     // batched form: blockIdx.z = one of gridDim.z row blocks of M rows each
  part += (long)blockIdx.z * gridDim.x * N;
  const long row_lo = (long)blockIdx.x * rows_per_block;
  const long row_hi = row_lo + rows_per_block < M ? row_lo + rows_per_block : M;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (cok) {
    for (long base = row_lo + ry; base < row_hi; base += 64) {      // four rows' loads in flight per pass
      uint4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long r = base + 16 * i;
        v[i] = r < row_hi ? *(const uint4_t*)(dy + r * ld + chunk * 8) : (uint4_t){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8(v[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) ps[ry][cx * 8 + e] = s[e];
  __syncthreads();
  if (tid < 128) {
    const int c = blockIdx.y * 128 + tid;
    if (c < N) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += ps[r][tid];
      part[(long)blockIdx.x * N + c] = a;
    }
  }
}
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             int nsplit, int N, int accumulate) {
  __shared__ float sa[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  part += (long)blockIdx.y * nsplit * N;               // batched form: blockIdx.y = row block
  out += (long)blockIdx.y * N;
  float a = 0.f;
  if (c < N)
#pragma unroll 4
    for (int r = ty; r < nsplit; r += 16) a += part[(long)r * N + c];
  sa[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c < N) {
#pragma unroll
    for (int j = 1; j < 16; ++j) a += sa[j][tx];
    out[c] = accumulate ? out[c] + a : a;
  }
}
static int colsum_split(long M) { const int rows = colsum_rows(M); return (int)((M + rows - 1) / rows); }
extern "C" long nk_colsum_ws_floats(long M, int N) { return (long)colsum_split(M) * N + 64; }
extern "C" int nk_colsum_batched(const void* dy, float* out, float* ws, long M, int N, long ld, int nbatch, int accumulate, void* stream_) {
  // out[b][N] (+)= column sums of rows [b*M, (b+1)*M) of dy, b < nbatch, in ONE pair of launches (ws: nbatch x nk_colsum_ws_floats(M, N))
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(dy && out && ws && M > 0 && N > 0 && (N & 7) == 0 && (ld & 7) == 0 && nbatch >= 1 && nbatch <= 65535);
  const int nsplit = colsum_split(M);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nsplit, (N + 127) / 128, nbatch), dim3(256), 0, stream, (const bf16_t*)dy, ws, M, N, ld, colsum_rows(M));
  if (int e = nk_check_launch("colsum_partial")) return e;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64, nbatch), dim3(1024), 0, stream, ws, out, nsplit, N, accumulate);
  return nk_check_launch("colsum_reduce");
}
extern "C" int nk_colsum(const void* dy, float* out, float* ws, long M, int N, long ld, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(dy && out && ws && M > 0 && N > 0 && (N & 7) == 0 && (ld & 7) == 0);
  const int nsplit = colsum_split(M);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nsplit, (N + 127) / 128), dim3(256), 0, stream, (const bf16_t*)dy, ws, M, N, ld, colsum_rows(M));
  if (int e = nk_check_launch("colsum_partial")) return e;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, stream, ws, out, nsplit, N, accumulate);
  return nk_check_launch("colsum_reduce");
}

// ---- sinusoidal timestep embedding (modules/diffusion/util.py:152-177): [cos | sin] -------------
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int B, int dim,
                                          float max_period) {
  const int half = dim >> 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    int b = i / dim, j = i - b * dim;
    float v = 0.f;
    if (j < 2 * half) {
      int k = j < half ? j : j - half;
      float freq = expf(-logf(max_period) * (float)k / (float)half);
      float arg = t[b] * freq;
      v = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = f2bf(v);
  }
}
extern "C" int nk_timestep_embedding(const float* t, void* out, int B, int dim, float max_period, void* stream) {
  NK_CHECK_ARG(t && out && B > 0 && dim > 0);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                     (bf16_t*)out, B, dim, max_period);
  return nk_check_launch("timestep_embedding");
}

// ---- row softmax in place on bf16 [M][L] (unfused d=512 VAE mid attention, model.py:224-243) ----
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, long M, int L) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long row = blockIdx.x; row < M; row += gridDim.x) {
    bf16_t* p = s + row * L;
    float mx = -INFINITY;
    for (int i = tid * 8; i < L; i += 256 * 8) {
      float f[8];
      unpack8(*(const uint4_t*)(p + i), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid * 8; i < L; i += 256 * 8) {
      float f[8];
      unpack8(*(const uint4_t*)(p + i), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += __expf(f[e] - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid * 8; i < L; i += 256 * 8) {
      float f[8];
      unpack8(*(const uint4_t*)(p + i), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __expf(f[e] - mx) * inv;
      *(uint4_t*)(p + i) = pack8(f);
    }
    __syncthreads();
  }
}
// The same for rows of up to 16 384 elements (the SD VAE's 128 x 128 mid block at 1024^2, and everything smaller): a row is held in the
// block's registers between its one read and its one write -- 2 bytes in, 2 out per element where the general kernel reads the row three
// times (1.59 GB fetched per 0.54 GB score matrix: profiles/r03_pmc_summary.csv).
template <int NPT>     // 16-byte pieces per thread: L <= NPT * 2048
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(bf16_t* __restrict__ s, long M, int L) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long row = blockIdx.x; row < M; row += gridDim.x) {
    bf16_t* p = s + row * L;
    uint4_t v[NPT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = (j * 256 + tid) * 8;
      v[j] = (uint4_t){0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};     // -inf: exp -> 0 past the end of the row
      if (i < L) v[j] = *(const uint4_t*)(p + i);
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      float f[8];
      unpack8(v[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      float f[8];
      unpack8(v[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += __expf(f[e] - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = (j * 256 + tid) * 8;
      float f[8];
      unpack8(v[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __expf(f[e] - mx) * inv;
      if (i < L) *(uint4_t*)(p + i) = pack8(f);
    }
    __syncthreads();
  }
}
extern "C" int nk_softmax_rows(void* s, long M, int L, void* stream) {
  NK_CHECK_ARG(s && M > 0 && L > 0 && (L & 7) == 0);
  const dim3 grid((int)(M < 8192 ? M : 8192));
  if (L <= 4096) hipLaunchKernelGGL(softmax_rows_reg_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, M, L);
  else if (L <= 8192) hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, M, L);
  else if (L <= 16384) hipLaunchKernelGGL(softmax_rows_reg_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, M, L);
  else hipLaunchKernelGGL(softmax_rows_kernel, dim3((int)(M < 4096 ? M : 4096)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, M, L);
  return nk_check_launch("softmax_rows");
}

// ---- backward of the row softmax, in place on dp: ds = p * (dp - sum_j dp*p) * scale ---------------
// (the VAE mid-block attention when the VAE itself is trained: its probabilities are kept from the forward pass)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const bf16_t* __restrict__ p, bf16_t* __restrict__ dp, long M, int L,
                                                               float scale) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long row = blockIdx.x; row < M; row += gridDim.x) {
    const bf16_t* pr = p + row * L;
    bf16_t* dr = dp + row * L;
    float dot = 0.f;
    for (int i = tid * 8; i < L; i += 256 * 8) {
      float a[8], b[8];
      unpack8(*(const uint4_t*)(pr + i), a);
      unpack8(*(const uint4_t*)(dr + i), b);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += a[e] * b[e];
    }
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    for (int i = tid * 8; i < L; i += 256 * 8) {
      float a[8], b[8];
      unpack8(*(const uint4_t*)(pr + i), a);
      unpack8(*(const uint4_t*)(dr + i), b);
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = a[e] * (b[e] - dot) * scale;
      *(uint4_t*)(dr + i) = pack8(b);
    }
    __syncthreads();
  }
}
extern "C" int nk_softmax_rows_bwd(const void* p, void* dp, long M, int L, float scale, void* stream) {
  NK_CHECK_ARG(p && dp && M > 0 && L > 0 && (L & 7) == 0);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((int)(M < 4096 ? M : 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p,
                     (bf16_t*)dp, M, L, scale);
  return nk_check_launch("softmax_rows_bwd");
}

// ---- EDM noising + preconditioning (loss.py:117-140, denoiser.py:41-49) -------------------------
// z_t = x + sigma*eps (fp32, NCHW) ; net_in = bf16 channels-last (z_t * c_in), channels padded to Cpad
__global__ void edm_prepare_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                   const float* __restrict__ sigma, const float* __restrict__ c_in,
                                   float* __restrict__ zt, bf16_t* __restrict__ net_in, int B, int C, int HW, int Cpad) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / HW);
    int p = (int)(i - (long)b * HW);
    const float sg = sigma[b], ci = c_in[b];
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        long o = ((long)b * C + c) * HW + p;
        float z = x[o] + sg * eps[o];
        zt[o] = z;
        v = z * ci;
      }
      net_in[i * Cpad + c] = f2bf(v);
    }
  }
}
extern "C" int nk_edm_prepare(const float* x, const float* eps, const float* sigma, const float* c_in, float* zt,
                              void* net_in, int B, int C, int HW, int Cpad, void* stream) {
  NK_CHECK_ARG(x && eps && sigma && c_in && zt && net_in && B > 0 && C > 0 && HW > 0 && Cpad >= C);
  hipLaunchKernelGGL(edm_prepare_kernel, dim3(ew_blocks((long)B * HW)), dim3(EW_THREADS), 0, (hipStream_t)stream, x,
                     eps, sigma, c_in, zt, (bf16_t*)net_in, B, C, HW, Cpad);
  return nk_check_launch("edm_prepare");
}

// ---- EDM loss forward + its gradient w.r.t. the network output (loss.py:142-157, functions.py:91-94)
// D = net_out*c_out + z_t*c_skip ; loss[b] = w[b] * mean_chw((D - target)^2)
// dnet = upstream * w[b] * 2/(C*HW) * (D - target) * c_out       (bf16 channels-last, padded channels = 0)
// one workgroup per sample (the latents are MB-sized): a fixed-order reduction keeps the loss bit-reproducible
__global__ __launch_bounds__(1024) void edm_loss_kernel(const bf16_t* __restrict__ net_out, const float* __restrict__ zt,
                                                        const float* __restrict__ target, const float* __restrict__ c_out,
                                                        const float* __restrict__ c_skip, const float* __restrict__ w,
                                                        float* __restrict__ loss, bf16_t* __restrict__ dnet, int B, int C,
                                                        int HW, int Cpad, float upstream) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float co = c_out[b], cs = c_skip[b], wb = w[b];
  const float inv_cnt = 1.0f / ((float)C * (float)HW);
  const float gscale = upstream * wb * 2.0f * inv_cnt * co;
  float acc = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    for (int c = 0; c < Cpad; ++c) {
      float g = 0.f;
      if (c < C) {
        long o = ((long)b * C + c) * HW + p;
        float D = bf2f(net_out[((long)b * HW + p) * Cpad + c]) * co + zt[o] * cs;
        float diff = D - target[o];
        acc += diff * diff;
        g = diff * gscale;
      }
      if (dnet) dnet[((long)b * HW + p) * Cpad + c] = f2bf(g);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    loss[b] = t * inv_cnt * wb;
  }
}
extern "C" int nk_edm_loss(const void* net_out, const float* zt, const float* target, const float* c_out,
                           const float* c_skip, const float* w, float* loss, void* dnet, int B, int C, int HW, int Cpad,
                           float upstream, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(net_out && zt && target && c_out && c_skip && w && loss && B > 0 && C > 0 && HW > 0 && Cpad >= C);
  hipLaunchKernelGGL(edm_loss_kernel, dim3(B), dim3(1024), 0, stream, (const bf16_t*)net_out, zt, target, c_out,
                     c_skip, w, loss, (bf16_t*)dnet, B, C, HW, Cpad, upstream);
  return nk_check_launch("edm_loss");
}

// ---- flat fused AdamW over the whole parameter buffer; also refreshes the bf16 shadow ----------
__global__ void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, bf16_t* __restrict__ shadow, long n, float lr, float b1,
                                  float b2, float eps, float wd, float bc1, float bc2, float gscale, const unsigned* health) {
  if (*(const volatile unsigned*)health) return;     // a flagged backward: leave masters, moments and shadows untouched
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4_t pp = *(float4_t*)(p + i * 4);
    const float4_t gg = *(const float4_t*)(g + i * 4);
    float4_t mm = *(float4_t*)(m + i * 4), vv = *(float4_t*)(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gg[e] * gscale;
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      float upd = (mm[e] / bc1) / (sqrtf(vv[e] / bc2) + eps);
      pp[e] = pp[e] * (1.f - lr * wd) - lr * upd;
    }
    *(float4_t*)(p + i * 4) = pp;
    *(float4_t*)(m + i * 4) = mm;
    *(float4_t*)(v + i * 4) = vv;
    uint2_t s;
    s.x = pack2bf(pp[0], pp[1]);
    s.y = pack2bf(pp[2], pp[3]);
    *(uint2_t*)(shadow + i * 4) = s;
  }
}
extern "C" int nk_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, long n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  NK_CHECK_ARG(p && g && m && v && shadow && n > 0 && (n & 3) == 0 && step >= 1);
  if (int e = nk_health_poll()) return e;
  const unsigned* health = nk_health_word();
  if (!health) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(ew_blocks(n >> 2)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)shadow, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, health);
  if (int e = nk_check_launch("adamw_flat")) return e;
  nk_health_snapshot((hipStream_t)stream);
  return NK_OK;
}
