// PatchGAN discriminator pieces: BatchNorm2d in training mode with the LeakyReLU that always follows it
// fused in, and a stand-alone LeakyReLU for the first layer (modules/losses/patchgan/model.py:21-95).
//
// Layout: channels-last bf16 tokens x[M][C], M = N*H*W.  Batch statistics are per CHANNEL over all M rows -- a column
// reduction, done like the bias-gradient column sums: 16 row lanes x 16 chunk lanes per workgroup (a chunk = 8 channels =
// 16 B), fp32 partials per 64-row slab, single-writer finalisation in a fixed order (bitwise reproducible).  All three passes
// are HBM-bound: forward = 2 reads + 1 write of x, backward = 3 reads + 1 write.
#include "nk_common.h"
#include "../../include/neurosis_hip.h"

#define BN_ROWS 64
static int bn_slabs(long M) { return (int)((M + BN_ROWS - 1) / BN_ROWS); }

__device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }

// ---- LeakyReLU --------------------------------------------------------------------------------------
__global__ void leaky_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8, float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*(const uint4_t*)(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = leaky(f[e], slope);
    *(uint4_t*)(y + i * 8) = pack8(f);
  }
}
// dx = dy where the OUTPUT is > 0, else dy * slope (the output has the input's sign for slope > 0, and is 0 exactly where a
// ReLU -- slope 0 -- blocked it; torch takes the `slope` branch at 0 as well)
__global__ void leaky_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ dx, long n8,
                                 float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(*(const uint4_t*)(y + i * 8), f);
    unpack8(*(const uint4_t*)(dy + i * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = f[e] > 0.f ? g[e] : g[e] * slope;
    *(uint4_t*)(dx + i * 8) = pack8(g);
  }
}
static int ew_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}
extern "C" int nk_leaky_relu_fwd(const void* x, void* y, long n, float slope, void* stream) {
  NK_CHECK_ARG(x && y && n > 0 && (n & 7) == 0 && slope >= 0.f);
  hipLaunchKernelGGL(leaky_fwd_kernel, dim3(ew_grid(n >> 3)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n >> 3, slope);
  return nk_check_launch("leaky_relu_fwd");
}
extern "C" int nk_leaky_relu_bwd(const void* dy, const void* y, void* dx, long n, float slope, void* stream) {
  NK_CHECK_ARG(dy && y && dx && n > 0 && (n & 7) == 0 && slope >= 0.f);
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(ew_grid(n >> 3)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)y,
                     (bf16_t*)dx, n >> 3, slope);
  return nk_check_launch("leaky_relu_bwd");
}

// ---- column partials: part[slab][2][C] = { sum_m a, sum_m a*b } over the slab's rows -----------------
// MODE 0 (forward statistics):  a = x,               b = x
// MODE 1 (backward sums):       a = g = dy * act'(y), b = xhat = (x - mean) * rstd     ->  { sum g, sum g*xhat }
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                         const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ part, long M, int C,
                                                         float slope) {
  __shared__ float ps[2][16][16 * 8 + 4];
  const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;
  const int chunk = blockIdx.y * 16 + cx;
  const bool cok = chunk < (C >> 3);
  const long row_lo = (long)blockIdx.x * BN_ROWS;
  float s0[8], s1[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f, mu[e] = 0.f, rs[e] = 1.f;
  if (cok) {
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) mu[e] = mean[chunk * 8 + e], rs[e] = rstd[chunk * 8 + e];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = row_lo + ry + 16 * i;
      if (r >= M) continue;
      float a[8], b[8];
      unpack8(*(const uint4_t*)(x + r * C + chunk * 8), b);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] += b[e], s1[e] += b[e] * b[e];
      } else {
        float o[8];
        unpack8(*(const uint4_t*)(dy + r * C + chunk * 8), a);
        if (slope != 1.f) {
          unpack8(*(const uint4_t*)(y + r * C + chunk * 8), o);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = o[e] > 0.f ? a[e] : a[e] * slope;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] += a[e], s1[e] += a[e] * (b[e] - mu[e]) * rs[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) ps[0][ry][cx * 8 + e] = s0[e], ps[1][ry][cx * 8 + e] = s1[e];
  __syncthreads();
  const int which = tid >> 7, col = tid & 127;
  const int c = blockIdx.y * 128 + col;
  if (c < C) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += ps[which][r][col];
    part[((long)blockIdx.x * 2 + which) * C + c] = a;
  }
}

// Sum the slab partials of 64 channels: 16 slab lanes x 64 channel lanes per workgroup, then a fixed-order LDS reduction (a single
// thread per channel walking 8 192 slabs took 240 us at the autoencoder's 256^2 / batch 32 sizes).  Returns the two sums of
// channel blockIdx.x * 64 + (tid & 63) in every thread of slab lane 0.
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ part, int slabs, int C, float& s, float& q) {
  __shared__ float sh[2][16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f, b = 0.f;
  if (c < C)
#pragma unroll 4
    for (int i = ty; i < slabs; i += 16) a += part[((long)i * 2) * C + c], b += part[((long)i * 2 + 1) * C + c];
  sh[0][ty][tx] = a;
  sh[1][ty][tx] = b;
  __syncthreads();
  s = q = 0.f;
  if (ty == 0)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += sh[0][j][tx], q += sh[1][j][tx];
}
// forward finalisation: mean, rstd (biased variance), running statistics (unbiased variance, momentum)
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ part, int slabs, int C, long M, float eps,
                                                                 float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var) {
  float s, q;
  bn_sum_partials(part, slabs, C, s, q);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  const float m = s / (float)M;
  const float var = fmaxf(q / (float)M - m * m, 0.f);
  mean[c] = m;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}
// backward finalisation: dgamma = sum g*xhat, dbeta = sum g (kept in sums[2][C] for the dx pass, written / added to the grads)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int slabs, int C, float* __restrict__ sums,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  float s, q;
  bn_sum_partials(part, slabs, C, s, q);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  sums[c] = s;
  sums[C + c] = q;
  dbeta[c] = accumulate ? dbeta[c] + s : s;
  dgamma[c] = accumulate ? dgamma[c] + q : q;
}

// y = act((x - mean) * rstd * gamma + beta)
__global__ void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, long M, int C,
                                float slope) {
  const int chunks = C >> 3;
  const long total = M * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    float f[8];
    unpack8(*(const uint4_t*)(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = leaky((f[e] - mean[ch + e]) * rstd[ch + e] * gamma[ch + e] + beta[ch + e], slope);
    *(uint4_t*)(y + i * 8) = pack8(f);
  }
}
// dx = gamma * rstd * (g - sum_g / M - xhat * sum_gx / M),  g = dy * act'(y)
__global__ void bn_bwd_dx_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                 const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                 const float* __restrict__ sums, bf16_t* __restrict__ dx, long M, int C, float slope) {
  const int chunks = C >> 3;
  const long total = M * chunks;
  const float inv_m = 1.f / (float)M;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    float g[8], v[8], o[8];
    unpack8(*(const uint4_t*)(dy + i * 8), g);
    unpack8(*(const uint4_t*)(x + i * 8), v);
    if (slope != 1.f) {
      unpack8(*(const uint4_t*)(y + i * 8), o);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = o[e] > 0.f ? g[e] : g[e] * slope;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xhat = (v[e] - mean[ch + e]) * rstd[ch + e];
      g[e] = gamma[ch + e] * rstd[ch + e] * (g[e] - sums[ch + e] * inv_m - xhat * sums[C + ch + e] * inv_m);
    }
    *(uint4_t*)(dx + i * 8) = pack8(g);
  }
}

extern "C" long nk_batchnorm_ws_floats(long M, int C) { return (long)bn_slabs(M) * 2 * C + 2 * (long)C + 64; }

extern "C" int nk_batchnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                float* running_mean, float* running_var, float* ws, long M, int C, float eps, float momentum,
                                float slope, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(x && gamma && beta && y && mean && rstd && ws && M > 0 && C > 0 && (C & 7) == 0 && slope > 0.f);
  NK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  const int slabs = bn_slabs(M);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(slabs, (C + 127) / 128), dim3(256), 0, stream, (const bf16_t*)x, nullptr, nullptr, nullptr,
                     nullptr, ws, M, C, 1.f);
  if (int e = nk_check_launch("bn_partial<0>")) return e;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, ws, slabs, C, M, eps, momentum, mean, rstd,
                     running_mean, running_var);
  if (int e = nk_check_launch("bn_stats_finalize")) return e;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)x, mean, rstd, gamma, beta,
                     (bf16_t*)y, M, C, slope);
  return nk_check_launch("bn_apply");
}

// nn.BatchNorm2d in evaluation mode (+ the LeakyReLU behind it): the running statistics as a per-channel affine map.
// ws: C fp32 elements (rstd of the running variance).
__global__ void bn_eval_rstd_kernel(const float* __restrict__ running_var, float eps, float* __restrict__ rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) rstd[c] = rsqrtf(running_var[c] + eps);
}
extern "C" int nk_batchnorm_eval(const void* x, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, void* y, float* ws, long M, int C, float eps, float slope, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && ws && M > 0 && C > 0 && (C & 7) == 0 && slope > 0.f);
  hipLaunchKernelGGL(bn_eval_rstd_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_var, eps, ws, C);
  if (int e = nk_check_launch("bn_eval_rstd")) return e;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)x, running_mean, ws, gamma, beta,
                     (bf16_t*)y, M, C, slope);
  return nk_check_launch("bn_apply (eval)");
}

extern "C" int nk_batchnorm_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                                const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, long M, int C, float slope,
                                int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(dy && x && y && gamma && mean && rstd && dx && dgamma && dbeta && ws && M > 0 && C > 0 && (C & 7) == 0 && slope > 0.f);
  const int slabs = bn_slabs(M);
  float* sums = ws + (long)slabs * 2 * C;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(slabs, (C + 127) / 128), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy,
                     (const bf16_t*)y, mean, rstd, ws, M, C, slope);
  if (int e = nk_check_launch("bn_partial<1>")) return e;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, ws, slabs, C, sums, dgamma, dbeta, accumulate);
  if (int e = nk_check_launch("bn_bwd_finalize")) return e;
  hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(ew_grid(M * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)y, mean, rstd, gamma, sums, (bf16_t*)dx, M, C, slope);
  return nk_check_launch("bn_bwd_dx");
}


// =====================================================================================================
// LPIPS pieces (modules/losses/perceptual.py:64-228 over a VGG16 trunk): 2x2 max-pool and the per-layer distance.
// =====================================================================================================
// y[n][ho][wo][c] = max over the 2x2 window (H, W even).  One thread per 8 channels of an output pixel.
__global__ void maxpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, chunks = C >> 3;
  const long total = (long)N * Ho * Wo * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    long pix = i / chunks;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
    const bf16_t* base = x + (((long)n * H + 2 * ho) * W + 2 * wo) * C + ch;
    float a[8], b[8], c[8], d[8];
    unpack8(*(const uint4_t*)base, a);
    unpack8(*(const uint4_t*)(base + C), b);
    unpack8(*(const uint4_t*)(base + (long)W * C), c);
    unpack8(*(const uint4_t*)(base + (long)W * C + C), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    *(uint4_t*)(y + i * 8) = pack8(a);
  }
}
// dx: the window's gradient goes to its FIRST maximal element in scan order (as torch's max_pool2d); the rest get 0
__global__ void maxpool2_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, int N, int H,
                                    int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, chunks = C >> 3;
  const long total = (long)N * Ho * Wo * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    long pix = i / chunks;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
    const long o00 = (((long)n * H + 2 * ho) * W + 2 * wo) * C + ch;
    const long offs[4] = {o00, o00 + C, o00 + (long)W * C, o00 + (long)W * C + C};
    float v[4][8], g[8], out[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) unpack8(*(const uint4_t*)(x + offs[k]), v[k]);
    unpack8(*(const uint4_t*)(dy + i * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int best = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) best = v[k][e] > v[best][e] ? k : best;
#pragma unroll
      for (int k = 0; k < 4; ++k) out[k][e] = k == best ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *(uint4_t*)(dx + offs[k]) = pack8(out[k]);
  }
}
extern "C" int nk_maxpool2x2_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  NK_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (H & 1) == 0 && (W & 1) == 0);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid((long)N * (H / 2) * (W / 2) * (C >> 3))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, N, H, W, C);
  return nk_check_launch("maxpool2x2_fwd");
}
extern "C" int nk_maxpool2x2_bwd(const void* dy, const void* x, void* dx, int N, int H, int W, int C, void* stream) {
  NK_CHECK_ARG(dy && x && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (H & 1) == 0 && (W & 1) == 0);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid((long)N * (H / 2) * (W / 2) * (C >> 3))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, N, H, W, C);
  return nk_check_launch("maxpool2x2_bwd");
}

// General k x k / stride s max pooling without padding (floor mode), for the AlexNet trunk's overlapping 3x3 / 2 pools.
// Forward: one thread per 8 channels of an output pixel.  Backward GATHERS: one thread per 8 channels of an INPUT pixel walks the
// (at most ceil(k/s)^2) windows that contain it, re-derives each window's first maximal element in scan order (torch's max_pool2d
// tie rule) and adds that window's gradient when the element is this pixel -- overlapping windows need no atomics this way.
__global__ void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, int k, int s, int Ho,
                                   int Wo) {
  const int chunks = C >> 3;
  const long total = (long)N * Ho * Wo * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    long pix = i / chunks;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        float v[8];
        unpack8(*(const uint4_t*)(x + (((long)n * H + ho * s + kh) * W + wo * s + kw) * C + ch), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    *(uint4_t*)(y + i * 8) = pack8(m);
  }
}
__global__ void maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, int N, int H,
                                   int W, int C, int k, int s, int Ho, int Wo) {
  const int chunks = C >> 3;
  const long total = (long)N * H * W * chunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks) * 8;
    long pix = i / chunks;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H), n = (int)(pix / H);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    const int ho_lo = h - k + 1 > 0 ? (h - k + s) / s : 0, ho_hi = min(h / s, Ho - 1);
    const int wo_lo = w - k + 1 > 0 ? (w - k + s) / s : 0, wo_hi = min(w / s, Wo - 1);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        float best[8];
        int arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = -1; }
        for (int kh = 0; kh < k; ++kh)
          for (int kw = 0; kw < k; ++kw) {
            float v[8];
            unpack8(*(const uint4_t*)(x + (((long)n * H + ho * s + kh) * W + wo * s + kw) * C + ch), v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (v[e] > best[e] || arg[e] < 0) { best[e] = v[e]; arg[e] = kh * k + kw; }
          }
        const int mine = (h - ho * s) * k + (w - wo * s);
        float d[8];
        unpack8(*(const uint4_t*)(dy + (((long)n * Ho + ho) * Wo + wo) * C + ch), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += arg[e] == mine ? d[e] : 0.f;
      }
    *(uint4_t*)(dx + i * 8) = pack8(g);
  }
}
extern "C" int nk_maxpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int s, void* stream) {
  NK_CHECK_ARG(x && y && N > 0 && C > 0 && (C & 7) == 0 && k > 0 && s > 0 && H >= k && W >= k);
  const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid((long)N * Ho * Wo * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, N, H, W, C, k, s, Ho, Wo);
  return nk_check_launch("maxpool_fwd");
}
extern "C" int nk_maxpool_bwd(const void* dy, const void* x, void* dx, int N, int H, int W, int C, int k, int s, void* stream) {
  NK_CHECK_ARG(dy && x && dx && N > 0 && C > 0 && (C & 7) == 0 && k > 0 && s > 0 && H >= k && W >= k);
  const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid((long)N * H * W * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)x, (bf16_t*)dx, N, H, W, C, k, s, Ho, Wo);
  return nk_check_launch("maxpool_bwd");
}

// One LPIPS layer: per pixel, unit-normalise both feature vectors over the channels (x / (||x|| + eps)), square the difference,
// weight the channels with the layer's 1x1 "lin" convolution and average over the pixels of each image:
//   out[n] (+)= mean_p sum_c w[c] (a_c - u_c)^2,   a = f0 / (||f0|| + eps),  u = f1 / (||f1|| + eps)
// One wavefront per pixel (lanes stride the channels), 4 pixels per workgroup; per-workgroup partials summed in a fixed order.
#define LP_PIX 4
__global__ __launch_bounds__(256) void lpips_layer_fwd_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ f1,
                                                              const float* __restrict__ w, float* __restrict__ part, int HW, int C,
                                                              float eps) {
  __shared__ float red[LP_PIX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int p = blockIdx.x * LP_PIX + wave;
  float d = 0.f;
  if (p < HW) {
    const bf16_t* a = f0 + ((long)n * HW + p) * C;
    const bf16_t* b = f1 + ((long)n * HW + p) * C;
    float na = 0.f, nb = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float x = bf2f(a[c]), y = bf2f(b[c]);
      na += x * x;
      nb += y * y;
    }
    na = 1.f / (sqrtf(wave_sum(na)) + eps);
    nb = 1.f / (sqrtf(wave_sum(nb)) + eps);
    for (int c = lane; c < C; c += 64) {
      const float t = bf2f(a[c]) * na - bf2f(b[c]) * nb;
      d += w[c] * t * t;
    }
    d = wave_sum(d);
  }
  if (lane == 0) red[wave] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)n * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void lpips_layer_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int nblk,
                                                                 int HW, int accumulate) {
  __shared__ float red[4];
  const int n = blockIdx.x;           // one workgroup per image: 16 384 partials at 256^2, summed in a fixed order
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[(long)n * nblk + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0] + red[1] + red[2] + red[3]) / (float)HW;
    out[n] = accumulate ? out[n] + s : s;
  }
}
// d out[n] / d f1, times upstream[n]:  g_c = -2 w_c (a_c - u_c) / HW ;  df1_k = g_k * r - (sum_c g_c f1_c) * f1_k * r^2 / ||f1||,
// r = 1 / (||f1|| + eps)
__global__ __launch_bounds__(256) void lpips_layer_bwd_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ f1,
                                                              const float* __restrict__ w, const float* __restrict__ upstream,
                                                              bf16_t* __restrict__ df1, int HW, int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int p = blockIdx.x * LP_PIX + wave;
  if (p >= HW) return;
  const bf16_t* a = f0 + ((long)n * HW + p) * C;
  const bf16_t* b = f1 + ((long)n * HW + p) * C;
  bf16_t* o = df1 + ((long)n * HW + p) * C;
  float na = 0.f, nb = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float x = bf2f(a[c]), y = bf2f(b[c]);
    na += x * x;
    nb += y * y;
  }
  const float norm_b = sqrtf(wave_sum(nb));
  const float ra = 1.f / (sqrtf(wave_sum(na)) + eps), rb = 1.f / (norm_b + eps);
  const float scale = -2.f * upstream[n] / (float)HW;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float y = bf2f(b[c]);
    dot += scale * w[c] * (bf2f(a[c]) * ra - y * rb) * y;
  }
  dot = wave_sum(dot);
  const float back = norm_b > 0.f ? dot * rb * rb / norm_b : 0.f;
  for (int c = lane; c < C; c += 64) {
    const float y = bf2f(b[c]);
    o[c] = f2bf(scale * w[c] * (bf2f(a[c]) * ra - y * rb) * rb - back * y);
  }
}
extern "C" long nk_lpips_layer_ws_floats(int N, int HW) { return (long)N * ((HW + LP_PIX - 1) / LP_PIX) + 64; }
extern "C" int nk_lpips_layer_fwd(const void* f0, const void* f1, const float* w, float* out, float* ws, int N, int HW, int C, float eps,
                                  int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(f0 && f1 && w && out && ws && N > 0 && N <= 65535 && HW > 0 && C > 0);
  const int nblk = (HW + LP_PIX - 1) / LP_PIX;
  hipLaunchKernelGGL(lpips_layer_fwd_kernel, dim3(nblk, N), dim3(256), 0, stream, (const bf16_t*)f0, (const bf16_t*)f1, w, ws, HW, C, eps);
  if (int e = nk_check_launch("lpips_layer_fwd")) return e;
  hipLaunchKernelGGL(lpips_layer_reduce_kernel, dim3(N), dim3(256), 0, stream, ws, out, N, nblk, HW, accumulate);
  return nk_check_launch("lpips_layer_reduce");
}
extern "C" int nk_lpips_layer_bwd(const void* f0, const void* f1, const float* w, const float* upstream, void* df1, int N, int HW, int C,
                                  float eps, void* stream) {
  NK_CHECK_ARG(f0 && f1 && w && upstream && df1 && N > 0 && N <= 65535 && HW > 0 && C > 0);
  hipLaunchKernelGGL(lpips_layer_bwd_kernel, dim3((HW + LP_PIX - 1) / LP_PIX, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)f0,
                     (const bf16_t*)f1, w, upstream, (bf16_t*)df1, HW, C, eps);
  return nk_check_launch("lpips_layer_bwd");
}
