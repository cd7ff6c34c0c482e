// Ablation harness for the MFMA tile engine: times one linear-forward shape with parts of the main loop disabled.
#include "../../neurosis_amd/csrc/gemm.hip"
#include "../../neurosis_amd/csrc/errors.hip"
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 640;
  int am = argc > 4 ? atoi(argv[4]) : 0, bm = argc > 5 ? atoi(argv[5]) : 0, f32 = argc > 6 ? atoi(argv[6]) : 0;
  size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  if (am == 2) { /* A is [K][M] */ }
  std::vector<unsigned short> ha(na), hb(nb);
  for (size_t i = 0; i < na; ++i) ha[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (size_t i = 0; i < nb; ++i) hb[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  void *da, *db, *dc;
  hipMalloc(&da, na * 2); hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 4);
  hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice);
  NkGemmParams p; memset((void*)&p, 0, sizeof(p));
  p.alpha = 1.f; p.fRowsPerBatch = make_fastdiv(1);
  p.A = (bf16_t*)da; p.B = (bf16_t*)db; p.C = dc;
  p.M = M; p.N = N; p.K = K;
  p.lda = am == 2 ? M : K; p.ldb = bm == 2 ? N : K; p.ldc = N;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int it = 0; it < 3; ++it) { NkGemmParams q = p; nk_gemm_dispatch(q, am, bm, f32, 0, 0); }
  hipEventRecord(s);
  const int iters = 20;
  for (int it = 0; it < iters; ++it) { NkGemmParams q = p; nk_gemm_dispatch(q, am, bm, f32, 0, 0); }
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= iters;
  printf("%-10s M=%d N=%d K=%d modes=%d,%d: %.1f us  %.1f TF/s\n", ABL_NAME, M, N, K, am, bm, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  return 0;
}
