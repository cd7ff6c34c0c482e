// Issue cost of v_exp_f32 (and of the other vector instructions of an attention loop) beside v_mfma_f32_32x32x16_bf16 on gfx950.
// Settles DESIGN.md section 3.4's premise: is an exponential 8 or 16 issue cycles next to the matrix pipe?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro_exp_mfma.hip -o tools/bin/micro_exp_mfma && tools/bin/micro_exp_mfma
//
// Each variant runs a loop of GAPS MFMA gaps; a gap is one 32x32x16 MFMA (four independent accumulators, so no dependency stall) followed by
// a fixed filler (NEXP x v_exp_f32, NADD x v_add_f32, NCVT x v_cvt_pk_bf16_f32), pinned in that order by sched_barriers.  One workgroup per
// CU (LDS-limited), 1 / 2 / 3 waves per SIMD; lane 0 of every wave stamps s_memtime around the loop; the table prints the median cycles per
// gap of a wave and, x waves per SIMD, the SIMD's cycles per MFMA.  MFMA floor: 32 cycles per gap and SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

template <int NEXP, int NADD, int NCVT, int MFMA>
__global__ __launch_bounds__(768) void probe(unsigned long long* out, float* sink, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  float16_t acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  float x[8], s[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned pk[4] = {0u, 0u, 0u, 0u};
  for (int e = 0; e < 8; ++e) x[e] = -0.001f * (lane + e + 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (MFMA) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NEXP; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(2 * g + k) & 7]));
#pragma unroll
      for (int k = 0; k < NADD; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[k & 3]) : "v"(x[(g + k) & 7]));
#pragma unroll
      for (int k = 0; k < NCVT; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[k & 3]) : "v"(x[k & 7]), "v"(x[(k + 1) & 7]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float keep = s[0] + s[1] + s[2] + s[3] + __uint_as_float(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]);
  for (int e = 0; e < 8; ++e) keep += x[e];
  for (int i = 0; i < 4; ++i) keep += acc[i][lane & 15];
  if (keep == 123.456f) sink[threadIdx.x] = keep;
  if (lane == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NEXP, int NADD, int NCVT, int MFMA>
static void run(const char* label, unsigned long long* dout, float* sink) {
  const int iters = 2000, gaps = 16 * iters;
  for (int waves = 1; waves <= 3; ++waves) {
    hipFuncSetAttribute((const void*)probe<NEXP, NADD, NCVT, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((probe<NEXP, NADD, NCVT, MFMA>), dim3(256), dim3(256 * waves), 100 * 1024, 0, dout, sink, iters);
    hipLaunchKernelGGL((probe<NEXP, NADD, NCVT, MFMA>), dim3(256), dim3(256 * waves), 100 * 1024, 0, dout, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 12);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 4 * waves; ++w) v.push_back((double)h[b * 12 + w] / gaps);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    printf("%-44s waves/SIMD %d: %7.1f cycles per gap and wave = %7.1f per MFMA and SIMD\n", label, waves, med, med / waves);
  }
}

int main() {
  unsigned long long* dout;
  float* sink;
  hipMalloc(&dout, 256 * 12 * 8);
  hipMalloc(&sink, 4096);
  run<0, 0, 0, 1>("MFMA only", dout, sink);
  run<1, 0, 0, 1>("MFMA + 1 exp", dout, sink);
  run<2, 0, 0, 1>("MFMA + 2 exp", dout, sink);
  run<3, 0, 0, 1>("MFMA + 3 exp", dout, sink);
  run<4, 0, 0, 1>("MFMA + 4 exp", dout, sink);
  run<0, 4, 0, 1>("MFMA + 4 add", dout, sink);
  run<0, 6, 0, 1>("MFMA + 6 add", dout, sink);
  run<0, 0, 4, 1>("MFMA + 4 cvt_pk", dout, sink);
  run<2, 2, 1, 1>("MFMA + 2 exp + 2 add + 1 cvt (attn64 loop)", dout, sink);
  run<1, 1, 1, 1>("MFMA + 1 exp + 1 add + 1 cvt", dout, sink);
  run<4, 0, 0, 0>("4 exp, no MFMA", dout, sink);
  run<0, 4, 0, 0>("4 add, no MFMA", dout, sink);
  run<2, 2, 1, 0>("2 exp + 2 add + 1 cvt, no MFMA", dout, sink);
  return 0;
}
