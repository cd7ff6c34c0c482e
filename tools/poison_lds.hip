// Fills every CU's LDS (and the vector registers of the waves that do it) with a NaN pattern (0x7FC07FC0: NaN as fp32, two NaNs as bf16), so that a
// kernel launched next that reads LDS it has not written yet -- or registers it has not loaded yet -- produces NaN instead of whatever the previous
// launch of the same test left there (tools/race_stress.py: a stale read of the SAME data is invisible).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/poison_lds.hip -o tools/bin/libpoison_lds.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void poison_kernel(unsigned* sink) {
  extern __shared__ unsigned lds[];
  const unsigned pat = 0x7FC07FC0u;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = pat;
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x * 64) acc ^= lds[i];
  if (acc == 12345u) sink[threadIdx.x] = acc;
}

extern "C" int poison_lds(void* sink, void* stream) {
  static bool opted = false;
  if (!opted) {
    if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    opted = true;
  }
  hipLaunchKernelGGL(poison_kernel, dim3(256), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned*)sink);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
