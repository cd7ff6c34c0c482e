// one-wave probe: D layout of v_mfma_f32_16x16x32_bf16 and the result order of v_permlane16_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
__global__ void probe(float* out, unsigned* sw) {
  int l = threadIdx.x;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(l % 16); b[e] = (__bf16)((l / 16 == 0 && e == 0) ? 1.0f : 0.0f); }
  float4_t c = {0, 0, 0, 0};
  float4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);   // D[i][j] = i
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
  float4_t d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);  // A = e0 of k0 -> D[i][j] = [i row has 1 at k0] * B[0][j] = j for every i
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = d2[r];
  auto s = __builtin_amdgcn_permlane16_swap((unsigned)l, 100u + l, false, false);
  sw[l] = s[0]; sw[64 + l] = s[1];
}
int main() {
  float* out; unsigned* sw;
  hipMalloc(&out, 512 * 4); hipMalloc(&sw, 128 * 4);
  probe<<<1, 64>>>(out, sw);
  float h[512]; unsigned hs[128];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, sw, sizeof(hs), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5) printf("lane %2d: D(a=i)= %g %g %g %g   D2= %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
  printf("swap[0]:"); for (int l = 0; l < 64; ++l) printf(" %u", hs[l]); printf("\nswap[1]:"); for (int l = 0; l < 64; ++l) printf(" %u", hs[64 + l]); printf("\n");
  return 0;
}
