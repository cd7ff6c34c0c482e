// Sampler-side latent kernels: everything the sampling loop does to the latents between two UNet
// forwards, in two launches per step instead of the reference's dozen elementwise ops.
//
// Layouts: latents x / denoised / x_next are fp32 NCHW [B][C][HW] (what the sampler API hands around, as in the
// reference); the network side is bf16 channels-last tokens [rep*B][HW][Cpad] (Cpad = C rounded up to 8, what the
// UNet's first conv gathers from and its last conv writes).  rep = 2 is classifier-free guidance with the batch laid out
// [unconditional | conditional] as VanillaCFG.prepare_inputs does (modules/guidance.py:26-37).
//
// All three kernels are latency-sized (an SDXL 1024^2 latent is 256 KB per sample): one thread per pixel, the token side
// moved as one 16-byte vector per pixel when Cpad == 8, the NCHW side coalesced across pixels.
#include "nk_common.h"
#include "../../include/neurosis_hip.h"

#define SMP_THREADS 256
static inline int smp_blocks(long n) {
  long b = (n + SMP_THREADS - 1) / SMP_THREADS;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// net_in[r*B + b][p][c] = bf16(c_in[b] * x[b][c][p]) for every replica r (padding channels = 0).
// Denoiser.forward's `inputs * c_in` (modules/diffusion/denoiser.py:41-49) fused with the guider's torch.cat([x] * 2).
__global__ __launch_bounds__(SMP_THREADS) void sample_prepare_kernel(const float* __restrict__ x, const float* __restrict__ c_in,
                                                                     bf16_t* __restrict__ net_in, int B, int C, int HW, int Cpad,
                                                                     int rep) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (long)b * HW);
    const float ci = c_in[b];
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        f[e] = c < C ? x[((long)b * C + c) * HW + p] * ci : 0.f;
      }
      const uint4_t v = pack8(f);
      for (int r = 0; r < rep; ++r) *(uint4_t*)(net_in + (((long)r * B + b) * HW + p) * Cpad + c0) = v;
    }
  }
}

// The guided denoiser output for pixel (b, p), channels c0..c0+7:
//   F  = F_u + scale * (F_c - F_u)          (rep == 2; rep == 1: F = net_out)
//   D  = c_skip[b] * x + c_out[b] * F
// (reference: D_u/D_c = F * c_out + x * c_skip per half, then D_u + scale (D_c - D_u): the same value, the x term
// factored out since both halves share x and sigma.)
__device__ __forceinline__ void guided_denoise8(const bf16_t* __restrict__ net_out, const float* x, int B, int C,
                                                int HW, int Cpad, int rep, int b, int p, int c0, float cs, float co, float scale,
                                                float* xv, float* D) {
  float fu[8], fc[8];
  unpack8(*(const uint4_t*)(net_out + (((long)b) * HW + p) * Cpad + c0), fu);
  if (rep == 2) {
    unpack8(*(const uint4_t*)(net_out + (((long)B + b) * HW + p) * Cpad + c0), fc);
#pragma unroll
    for (int e = 0; e < 8; ++e) fu[e] = fu[e] + scale * (fc[e] - fu[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    xv[e] = c < C ? x[((long)b * C + c) * HW + p] : 0.f;
    D[e] = cs * xv[e] + co * fu[e];
  }
}

__global__ __launch_bounds__(SMP_THREADS) void sample_denoise_kernel(const bf16_t* __restrict__ net_out, const float* __restrict__ x,
                                                                     const float* __restrict__ c_skip, const float* __restrict__ c_out,
                                                                     float scale, float* __restrict__ denoised, int B, int C, int HW,
                                                                     int Cpad, int rep) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (long)b * HW);
    const float cs = c_skip[b], co = c_out[b];
    for (int c0 = 0; c0 < C; c0 += 8) {
      float xv[8], D[8];
      guided_denoise8(net_out, x, B, C, HW, Cpad, rep, b, p, c0, cs, co, scale, xv, D);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < C) denoised[((long)b * C + c0 + e) * HW + p] = D[e];
    }
  }
}

// Euler step of the Karras ODE (sampling.py:166-181 with possible_correction_step = identity, utils.py:49-51):
//   d = (x - D) / sigma_hat[b] ; x_next = x + (sigma_next[b] - sigma_hat[b]) * d
__global__ __launch_bounds__(SMP_THREADS) void sample_euler_kernel(const bf16_t* __restrict__ net_out, const float* x,
                                                                   const float* __restrict__ c_skip, const float* __restrict__ c_out,
                                                                   const float* __restrict__ sigma_hat, const float* __restrict__ sigma_next,
                                                                   float scale, float* x_next, float* __restrict__ denoised,
                                                                   int B, int C, int HW, int Cpad, int rep) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (long)b * HW);
    const float cs = c_skip[b], co = c_out[b], sh = sigma_hat[b];
    const float dt = sigma_next[b] - sh;
    for (int c0 = 0; c0 < C; c0 += 8) {
      float xv[8], D[8];
      guided_denoise8(net_out, x, B, C, HW, Cpad, rep, b, p, c0, cs, co, scale, xv, D);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (c0 + e < C) {
          const long o = ((long)b * C + c0 + e) * HW + p;
          const float d = (xv[e] - D[e]) / sh;
          x_next[o] = xv[e] + dt * d;
          if (denoised) denoised[o] = D[e];
        }
      }
    }
  }
}

static bool smp_shape_ok(int B, int C, int HW, int Cpad, int rep) {
  return B > 0 && C > 0 && HW > 0 && Cpad >= C && (Cpad & 7) == 0 && Cpad - C < 8 && (rep == 1 || rep == 2);
}

extern "C" int nk_sample_prepare(const float* x, const float* c_in, void* net_in, int B, int C, int HW, int Cpad, int rep,
                                 void* stream) {
  NK_CHECK_ARG(x && c_in && net_in && smp_shape_ok(B, C, HW, Cpad, rep));
  hipLaunchKernelGGL(sample_prepare_kernel, dim3(smp_blocks((long)B * HW)), dim3(SMP_THREADS), 0, (hipStream_t)stream, x, c_in,
                     (bf16_t*)net_in, B, C, HW, Cpad, rep);
  return nk_check_launch("sample_prepare");
}

extern "C" int nk_sample_denoise(const void* net_out, const float* x, const float* c_skip, const float* c_out, float scale,
                                 float* denoised, int B, int C, int HW, int Cpad, int rep, void* stream) {
  NK_CHECK_ARG(net_out && x && c_skip && c_out && denoised && smp_shape_ok(B, C, HW, Cpad, rep));
  hipLaunchKernelGGL(sample_denoise_kernel, dim3(smp_blocks((long)B * HW)), dim3(SMP_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)net_out, x, c_skip, c_out, scale, denoised, B, C, HW, Cpad, rep);
  return nk_check_launch("sample_denoise");
}

extern "C" int nk_sample_euler_step(const void* net_out, const float* x, const float* c_skip, const float* c_out,
                                    const float* sigma_hat, const float* sigma_next, float scale, float* x_next, float* denoised,
                                    int B, int C, int HW, int Cpad, int rep, void* stream) {
  NK_CHECK_ARG(net_out && x && c_skip && c_out && sigma_hat && sigma_next && x_next && smp_shape_ok(B, C, HW, Cpad, rep));
  // x_next may alias x: every element is read before it is written, by the same thread (no __restrict__ on the pair)
  hipLaunchKernelGGL(sample_euler_kernel, dim3(smp_blocks((long)B * HW)), dim3(SMP_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)net_out, x, c_skip, c_out, sigma_hat, sigma_next, scale, x_next, denoised, B, C, HW, Cpad, rep);
  return nk_check_launch("sample_euler_step");
}
