// Ablation harness for the MFMA tile engine: times one linear-forward shape with parts of the main loop disabled.
#include "../../neurosis_amd/csrc/gemm.hip"
#include "../../neurosis_amd/csrc/errors.hip"
#include <vector>
#include <algorithm>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 640;
  int am = argc > 4 ? atoi(argv[4]) : 0, bm = argc > 5 ? atoi(argv[5]) : 0, f32 = argc > 6 ? atoi(argv[6]) : 0;
  size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  if (am == 2) { /* A is [K][M] */ }
  std::vector<unsigned short> ha(na), hb(nb);
  const int zero = argc > 7 ? atoi(argv[7]) : 0;
  for (size_t i = 0; i < na; ++i) ha[i] = zero ? 0 : 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (size_t i = 0; i < nb; ++i) hb[i] = zero ? 0 : 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
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
  printf("%-10s M=%d N=%d K=%d modes=%d,%d zero=%d: %.1f us  %.1f TF/s", ABL_NAME, M, N, K, am, bm, zero, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
#ifdef NK_CLOCK_STAMPS
  {
    static unsigned long long st[8192];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(nk_clock_stamps), sizeof(st));
    std::vector<double> clk;
    int nb_ = ((M + 127) / 128) * ((N + 127) / 128); if (nb_ > 4096) nb_ = 4096;
    for (int i = 0; i < nb_; ++i) if (st[2 * i + 1]) clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    if (!clk.empty()) printf("   in-loop shader clock: median %.2f GHz (min %.2f max %.2f), loop %.1f us", clk[clk.size() / 2], clk.front(), clk.back(), (double)st[1] * 0.01);
  }
#endif
  printf("\n");
  return 0;
}
