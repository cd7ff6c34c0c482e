// Common device helpers for the neurosis_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int uint2_t;

#define NK_OK 0
#define NK_ERR_ARG 1
#define NK_ERR_LAUNCH 2
#define NK_ERR_HEALTH 3   // a kernel of an earlier step reported an unusable result (see nk_health_* in errors.hip)

#define NK_CHECK_ARG(cond)                                                        \
  do {                                                                            \
    if (!(cond)) {                                                                \
      nk_set_error(__FILE__, __LINE__, #cond);                                    \
      return NK_ERR_ARG;                                                          \
    }                                                                             \
  } while (0)

void nk_set_error(const char* file, int line, const char* what);
int nk_check_launch(const char* what);
// Opt a kernel in to `bytes` of dynamic LDS ON THE CURRENT DEVICE, once per (kernel, device): the attribute is per device, so a process-wide
// "done" flag would leave a second GPU of the same process launching 100+ KiB kernels without it (round-3 advisor finding).  Thread-safe.
void nk_optin_lds(const void* kernel, int bytes);
// Backward-health word (errors.hip): ONE device word per process.  A kernel that cannot deliver a correct result (a
// stream-K fix-up that gave up waiting) raises it and poisons its output; the fused optimizer kernels read it first and
// leave every buffer untouched while it is set; the optimizer entry points report it on the host without synchronising.
unsigned* nk_health_word(void);                 // device pointer (allocated and zeroed on first use; nullptr on failure)
int nk_health_poll(void);                        // NK_OK, or NK_ERR_HEALTH once a finished snapshot has seen the word set
void nk_health_snapshot(hipStream_t stream);     // async copy of the word to pinned host memory behind `stream`'s work

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return __builtin_bit_cast(unsigned short, b);
}
typedef __attribute__((ext_vector_type(2))) float float2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  float2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));  // one v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void unpack8(const uint4_t& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4_t pack8(const float* f) {
  uint4_t v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}
__device__ __forceinline__ void unpack4(const uint2_t& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

// 64-lane wavefront reductions (xor butterfly; every lane ends with the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Division by a runtime-constant divisor without the ~20-instruction udiv sequence.
// Valid for n < 2^31.
struct FastDiv {
  unsigned d, m, s;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.m = 0; f.s = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s = l;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return (__umulhi(n, f.m) + n) >> f.s;      // d == 1 is m = 0, s = 0: the same expression, no branch in the callers' hot loops
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

// exact (erf) GELU and its derivative: GEGLU (modules/attention.py:50-57) in elementwise.hip and in the GEMM epilogue that fuses its backward
// The normal cdf / pdf at x from ONE exponential: erf by Abramowitz & Stegun 7.1.26, erf(y) = 1 - (a1 t + ... + a5 t^5) exp(-y^2), t = 1 / (1 + p y),
// |error| <= 1.5e-7 -- and with y = |x| / sqrt(2) its exp(-y^2) is the pdf's exp(-x^2 / 2).  ~16 vector instructions where the library erff
// (range-split polynomials) plus a separate exponential took ~60: the GEGLU backward fused into the FeedForward dgrad epilogue spent a third of
// its tile time on them.  (Outputs are rounded to bf16, 8 significant bits: the 1e-7 is invisible.)
__device__ __forceinline__ void normal_cdf_pdf(float x, float& cdf, float& pdf) {
  const float e = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * fabsf(x));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, pdf;
  normal_cdf_pdf(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float dgelu_erf(float x) {
  float cdf, pdf;
  normal_cdf_pdf(x, cdf, pdf);
  return cdf + x * pdf;
}
