// How many LDS-array cycles does a global->LDS DMA piece cost?  One 512-thread workgroup per CU; every wave issues 1-KiB pieces
// (global_load_lds_dwordx4) from a 64 KiB L2-resident source into a 128 KiB LDS ring, and/or streams ds_read_b128 over the ring.
// Prints bytes per clock per CU for: DMA only, reads only, both at once (are they additive or do they share the array?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int DMA_WAVES, int READ_WAVES, int RD_PER_PIECE>
__global__ __launch_bounds__(512) void k(const char* src, float* sink, long long* cycles, int iters) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* s = src + lane * 16;
  f4 acc = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
  if (wave < DMA_WAVES) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((gptr)(s + ((wave * 8 + i) & 63) * 1024), (lptr)(smem + ((wave * 8 + i) & 127) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave >= 8 - READ_WAVES) {
    typedef __attribute__((address_space(3))) const f4* l4;
    for (int it = 0; it < iters * RD_PER_PIECE; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f4 v = *(l4)(lptr)(smem + (((wave * 8 + i + it) & 127) * 1024) + lane * 16);
        acc += v;
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc.x == 12345.f) sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int D, int R, int RP>
void run(const char* name, const char* src, float* sink, long long* cyc, int iters) {
  hipFuncSetAttribute((const void*)k<D, R, RP>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<D, R, RP><<<256, 512, 128 * 1024>>>(src, sink, cyc, iters);
  hipEventRecord(a);
  k<D, R, RP><<<256, 512, 128 * 1024>>>(src, sink, cyc, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double cy = 0; for (int i = 0; i < 256; ++i) cy += h[i]; cy /= 256;
  double dma = (double)D * iters * 8 * 1024, rd = (double)R * iters * RP * 8 * 1024;
  // s_memtime ticks at 100 MHz; derive B/clk from wall time at an assumed shader clock instead, and print both raw numbers
  printf("%-34s  %8.1f us   DMA %7.1f GB/s/CU  reads %7.1f GB/s/CU   s_memtime %.0f ticks = %.0f MHz: DMA %6.1f B/tick  reads %6.1f B/tick\n", name, ms * 1e3,
         dma / (ms * 1e-3) / 1e9, rd / (ms * 1e-3) / 1e9, cy, cy / (ms * 1e3), dma / cy, rd / cy);
}

int main() {
  char* src; float* sink; long long* cyc;
  hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20); hipMalloc(&sink, 4096); hipMalloc(&cyc, 256 * 8);
  const int iters = 4000;
  run<1, 0, 1>("DMA 1 wave", src, sink, cyc, iters);
  run<2, 0, 1>("DMA 2 waves", src, sink, cyc, iters);
  run<4, 0, 1>("DMA 4 waves", src, sink, cyc, iters);
  run<8, 0, 1>("DMA 8 waves", src, sink, cyc, iters);
  run<0, 4, 4>("reads 4 waves", src, sink, cyc, iters);
  run<0, 8, 4>("reads 8 waves", src, sink, cyc, iters);
  run<4, 4, 1>("DMA 4 + reads 4 (1:1 bytes)", src, sink, cyc, iters);
  run<4, 4, 3>("DMA 4 + reads 4 (1:3 bytes)", src, sink, cyc, iters);
  run<2, 4, 3>("DMA 2 + reads 4 (1:6 bytes)", src, sink, cyc, iters);
  return 0;
}
