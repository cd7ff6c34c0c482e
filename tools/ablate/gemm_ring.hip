// Experiment: KC x KC tile engine with an NS-stage LDS-DMA ring, counted vmcnt and raw barriers (1 workgroup per CU).
#include "../../neurosis_amd/csrc/gemm.hip"
#include "../../neurosis_amd/csrc/errors.hip"
#include <vector>
#include <string.h>
#include <stdio.h>

template <int NS, int NT>
__global__ __launch_bounds__(NT, 1) void ring_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int per_group = GROUP_M * ntn, group = wg / per_group, first_m = group * GROUP_M;
  const int gm = min(GROUP_M, ntm - first_m), in_group = wg - group * per_group;
  const int nt = in_group / gm, mt = first_m + (in_group - nt * gm);
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = (p.K + BK - 1) / BK, kend = p.K;
  OperandDMA<OP_KC> opa, opb;
  opa.init(p.A, p.lda, p.M, m0, tid, p.ga);
  opb.init(p.B, p.ldb, p.N, n0, tid, p.gb);
  float4_t acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int D = NS - 1;
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) { opa.issue(t * BK, kend, smem + t * V2_STAGE_BYTES, p.ga, p.tw); opb.issue(t * BK, kend, smem + t * V2_STAGE_BYTES + V2_OPND_BYTES, p.gb, p.tw); }
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(kt + D - 1, nk - 1) - kt;   // tiles issued after tile kt
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + D < nk) {
      char* nx = smem + ((kt + D) % NS) * V2_STAGE_BYTES;
      opa.issue((kt + D) * BK, kend, nx, p.ga, p.tw);
      opb.issue((kt + D) * BK, kend, nx + V2_OPND_BYTES, p.gb, p.tw);
    }
    const char* cur = smem + (kt % NS) * V2_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = OperandDMA<OP_KC>::frag(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = OperandDMA<OP_KC>::frag(cur + V2_OPND_BYTES, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();
  nk_gemm_epilogue<0>(p, smem, acc, m0, n0, tid, lane, wm, wn);
}

template <int NS>
float run(const NkGemmParams& p, int iters) {
  int smem = NS * V2_STAGE_BYTES; if (smem < V2_SMEM_BYTES) smem = V2_SMEM_BYTES;
  hipFuncSetAttribute((const void*)ring_kernel<NS, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN));
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring_kernel<NS, 256>), grid, dim3(256), smem, 0, p);
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((ring_kernel<NS, 256>), grid, dim3(256), smem, 0, p);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms / iters;
}

int main(int argc, char** argv) {
  int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  std::vector<unsigned short> ha(na), hb(nb);
  for (size_t i = 0; i < na; ++i) ha[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (size_t i = 0; i < nb; ++i) hb[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  void *da, *db, *dc, *dc2;
  hipMalloc(&da, na * 2); hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 2); hipMalloc(&dc2, nc * 2);
  hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice);
  NkGemmParams p; memset((void*)&p, 0, sizeof(p));
  p.alpha = 1.f; p.fRowsPerBatch = make_fastdiv(1);
  p.A = (bf16_t*)da; p.B = (bf16_t*)db; p.C = dc; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  { NkGemmParams q = p; q.C = dc2; nk_gemm_dispatch(q, 0, 0, 0, 0, 0); }
  double fl = 2.0 * M * N * K;
  float t2 = run<2>(p, 20), t3 = run<3>(p, 20), t4 = run<4>(p, 20);
  // check vs the product kernel
  std::vector<unsigned short> h1(nc), h2(nc);
  hipMemcpy(h1.data(), dc, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), dc2, nc * 2, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < nc; ++i) bad += h1[i] != h2[i];
  printf("M=%d N=%d K=%d ring2 %.1f us %.0f TF/s | ring3 %.1f us %.0f TF/s | ring4 %.1f us %.0f TF/s | mismatches %zu\n", M, N, K, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9, t4 * 1e3, fl / t4 / 1e9, bad);
  return 0;
}
