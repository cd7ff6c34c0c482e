// HBM-bound normalisation kernels on channels-last bf16 data: GroupNorm(+SiLU) and LayerNorm,
// forward and backward.  Statistics are fp32; reductions are staged per thread -> LDS -> one fp32
// atomic per (block, statistic); loads and stores are 16 B per lane and row-contiguous.
//
// Reference: nn.GroupNorm(32, C)+nn.SiLU (modules/diffusion/openaimodel.py:247-250,281-283,797-799),
// SpatialTransformer.norm (modules/attention.py:612, eps 1e-6), Normalize (modules/layers.py:5-7),
// nn.LayerNorm (modules/attention.py:468-470).
#include "../../include/neurosis_hip.h"
#include "nk_common.h"

#define GN_THREADS 256
#define GN_MAXC 4096

// ------------------------------------------------------------------------------------------------
// GroupNorm forward, pass 1: per-(n, group) sum and sum of squares
// grid (nsplit, N); block handles rows [split*rows_per, ...) of image n
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ stats,
                                                              int HW, int C, int G, int rows_per) {
  // thread (rsub, chunk) accumulates 8 channels over its rows; partials are combined in a FIXED order
  // (LDS staging, no atomics) so the statistics -- and therefore the forward -- are bitwise reproducible
  __shared__ float ps[GN_THREADS * 8], pss[GN_THREADS * 8];
  const int n = blockIdx.y;
  const int cpr = (C >> 3) / gridDim.z;        // 16-byte chunks per row in this block's channel slab
  const int ch0 = blockIdx.z * cpr;            // first chunk of the slab
  const int rows_par = GN_THREADS / cpr;       // rows processed in parallel by the block
  const int tid = threadIdx.x;
  const int cpg = C / G;
  const int row_lo = blockIdx.x * rows_per;
  const int row_hi = min(HW, row_lo + rows_per);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  const int chunk = tid % cpr, rsub = tid / cpr;
  if (rsub < rows_par) {
    const bf16_t* base = x + ((long)n * HW) * C + (ch0 + chunk) * 8;
    int r = row_lo + rsub;
    for (; r + 3 * rows_par < row_hi; r += 4 * rows_par) {   // four rows' loads in flight per thread
      uint4_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const uint4_t*)(base + (long)(r + u * rows_par) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
      }
    }
    for (; r < row_hi; r += rows_par) {
      float f[8];
      unpack8(*(const uint4_t*)(base + (long)r * C), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ps[(rsub * cpr + chunk) * 8 + e] = s[e];
      pss[(rsub * cpr + chunk) * 8 + e] = ss[e];
    }
  }
  __syncthreads();
  // per-block partial: part[n][blockIdx.x * gridDim.z + blockIdx.z][2G]; groups outside this slab get 0
  float* part = stats + (((long)n * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * 2 * G;
  for (int i = tid; i < 2 * G; i += GN_THREADS) {
    const int g = i >> 1;
    const float* src = (i & 1) ? pss : ps;
    const int c_lo = max(g * cpg, ch0 * 8), c_hi = min((g + 1) * cpg, (ch0 + cpr) * 8);
    float a = 0.f;
    for (int c = c_lo; c < c_hi; ++c)
      for (int r = 0; r < rows_par; ++r) a += src[r * cpr * 8 + (c - ch0 * 8)];
    part[i] = a;
  }
}

// sums the per-block partials of one image in a fixed order: out[n][2G] = sum_p part[n][p][2G]
// block = G2 (<= 128) columns x (1024 / 128) row groups; each row group sums a strided subset, then a serial combine
__global__ __launch_bounds__(1024) void gn_reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                 int nparts, int G2) {
  __shared__ float sm[8][128];
  const int n = blockIdx.x;
  const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;   // 128 x 8
  float a = 0.f;
  if (tx < G2)
    for (int p = ty; p < nparts; p += 8) a += part[((long)n * nparts + p) * G2 + tx];
  sm[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && tx < G2) {
#pragma unroll
    for (int j = 1; j < 8; ++j) a += sm[j][tx];
    out[(long)n * G2 + tx] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm forward, pass 2: y = silu?((x - mean) * rstd * gamma + beta); also emits mean / rstd
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                              float* __restrict__ rstd_out, int HW, int C, int G,
                                                              float eps, int silu, int rows_per) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;   // [C]
  float* sh = sc + C;             // [C]
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  for (int c = tid; c < C; c += GN_THREADS) {
    int g = c / cpg;
    float m = stats[(long)n * 2 * G + 2 * g] * inv_cnt;
    float var = fmaxf(stats[(long)n * 2 * G + 2 * g + 1] * inv_cnt - m * m, 0.f);
    float rs = rsqrtf(var + eps);
    float a = rs * gamma[c];
    sc[c] = a;
    sh[c] = beta[c] - m * a;
    if (blockIdx.x == 0 && (c % cpg) == 0) { mean_out[n * G + g] = m; rstd_out[n * G + g] = rs; }
  }
  __syncthreads();
  const int cpr = C >> 3;
  const int row_lo = blockIdx.x * rows_per;
  const int row_hi = min(HW, row_lo + rows_per);
  const unsigned total = (unsigned)(row_hi - row_lo) * (unsigned)cpr;
  const bf16_t* xb = x + ((long)n * HW + row_lo) * C;
  bf16_t* yb = y + ((long)n * HW + row_lo) * C;
  auto one = [&](unsigned i, const uint4_t& raw) {
    const unsigned chunk = i % (unsigned)cpr;
    float f[8];
    unpack8(raw, f);
    const float4_t a0 = *(const float4_t*)(sc + chunk * 8), a1 = *(const float4_t*)(sc + chunk * 8 + 4);
    const float4_t b0 = *(const float4_t*)(sh + chunk * 8), b1 = *(const float4_t*)(sh + chunk * 8 + 4);
    f[0] = f[0] * a0[0] + b0[0]; f[1] = f[1] * a0[1] + b0[1]; f[2] = f[2] * a0[2] + b0[2]; f[3] = f[3] * a0[3] + b0[3];
    f[4] = f[4] * a1[0] + b1[0]; f[5] = f[5] * a1[1] + b1[1]; f[6] = f[6] * a1[2] + b1[2]; f[7] = f[7] * a1[3] + b1[3];
    if (silu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
    }
    *(uint4_t*)(yb + (size_t)i * 8) = pack8(f);
  };
  unsigned i = tid;
  for (; i + 3 * GN_THREADS < total; i += 4 * GN_THREADS) {   // four 16-byte loads in flight per thread
    uint4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const uint4_t*)(xb + (size_t)(i + u * GN_THREADS) * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(i + u * GN_THREADS, v[u]);
  }
  for (; i < total; i += GN_THREADS) one(i, *(const uint4_t*)(xb + (size_t)i * 8));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm backward, pass 1.
//   dz = dy * silu'(z) (z = xhat*gamma+beta) or dy
//   per channel: a_c = sum dz, b_c = sum dz*xhat           -> dbeta += a, dgamma += b
//   per (n,g):   s1 = sum_c gamma_c a_c, s2 = sum_c gamma_c b_c -> gsum[n][g][2]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GN_THREADS, 4) void gn_bwd_stats_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ gsum, float* __restrict__ chan_part,
                                                                  int HW, int C, int G, int silu, int rows_per) {
  __shared__ float pa[GN_THREADS * 8], pb[GN_THREADS * 8];   // per-thread partials, combined in a fixed order
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ca = (float*)smem_raw;   // [slab channels] sum dz
  float* cb = ca + C;             // [slab channels] sum dz*xhat
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int cpg = C / G;
  const int cpr = (C >> 3) / gridDim.z;
  const int ch0 = blockIdx.z * cpr;
  const int rows_par = GN_THREADS / cpr;
  const int row_lo = blockIdx.x * rows_per;
  const int row_hi = min(HW, row_lo + rows_per);
  const int chunk = tid % cpr, rsub = tid / cpr;
  if (rsub < rows_par) {
    float a[8], b[8], mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = (ch0 + chunk) * 8 + e;
      int g = c / cpg;
      a[e] = 0.f; b[e] = 0.f;
      mu[e] = mean[n * G + g]; rs[e] = rstd[n * G + g];
      ga[e] = gamma[c]; be[e] = beta[c];
    }
    const long base = ((long)n * HW) * C + (ch0 + chunk) * 8;
    auto one = [&](const uint4_t& rx, const uint4_t& rd) {
      float fx[8], fd[8];
      unpack8(rx, fx);
      unpack8(rd, fd);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = (fx[e] - mu[e]) * rs[e];
        float dz = fd[e];
        if (silu) dz *= dsilu_f(xh * ga[e] + be[e]);
        a[e] += dz;
        b[e] += dz * xh;
      }
    };
    int r = row_lo + rsub;
    for (; r + 3 * rows_par < row_hi; r += 4 * rows_par) {   // eight 16-byte loads in flight per thread
      uint4_t vx[4], vd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vx[u] = *(const uint4_t*)(x + base + (long)(r + u * rows_par) * C);
        vd[u] = *(const uint4_t*)(dy + base + (long)(r + u * rows_par) * C);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(vx[u], vd[u]);
    }
    for (; r < row_hi; r += rows_par) one(*(const uint4_t*)(x + base + (long)r * C), *(const uint4_t*)(dy + base + (long)r * C));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pa[(rsub * cpr + chunk) * 8 + e] = a[e];
      pb[(rsub * cpr + chunk) * 8 + e] = b[e];
    }
  }
  __syncthreads();
  // per channel of the slab: combine the rows_par partials; write channel partials for dgamma / dbeta
  float* cp = chan_part + ((long)n * gridDim.x + blockIdx.x) * 2 * C;
  for (int cl = tid; cl < cpr * 8; cl += GN_THREADS) {
    float av = 0.f, bv = 0.f;
    for (int r = 0; r < rows_par; ++r) { av += pa[r * cpr * 8 + cl]; bv += pb[r * cpr * 8 + cl]; }
    const int c = ch0 * 8 + cl;
    ca[cl] = av;
    cb[cl] = bv;
    cp[c] = bv;        // dgamma partial
    cp[C + c] = av;    // dbeta partial
  }
  __syncthreads();
  float* part = gsum + (((long)n * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * 2 * G;
  for (int i = tid; i < 2 * G; i += GN_THREADS) {
    const int g = i >> 1;
    const float* src = (i & 1) ? cb : ca;
    const int c_lo = max(g * cpg, ch0 * 8), c_hi = min((g + 1) * cpg, (ch0 + cpr) * 8);
    float acc = 0.f;
    for (int c = c_lo; c < c_hi; ++c) acc += gamma[c] * src[c - ch0 * 8];
    part[i] = acc;
  }
}

// dgamma[c] += sum_rows part[row][0][c] ; dbeta[c] += sum_rows part[row][1][c]
// block = 64 channels x 16 row groups (1024 threads); single writer per channel, so no atomics
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int nrows, int C, int accumulate) {
  __shared__ float sa[16][64], sb[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = ty; r < nrows; r += 16) {
      a += part[(long)r * 2 * C + c];
      b += part[(long)r * 2 * C + C + c];
    }
  }
  sa[ty][tx] = a;
  sb[ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int j = 1; j < 16; ++j) { a += sa[j][tx]; b += sb[j][tx]; }
    dgamma[c] = accumulate ? dgamma[c] + a : a;
    dbeta[c] = accumulate ? dbeta[c] + b : b;
  }
}

// GroupNorm backward, pass 2: dx = rstd * (dz*gamma - (s1 + xhat*s2)/cnt) (+ dx_add)
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gsum, const bf16_t* __restrict__ dx_add,
                                                                  bf16_t* __restrict__ dx, int HW, int C, int G, int silu,
                                                                  int rows_per) {
  // With z = x*A + B the pre-activation (A = rstd*gamma, B = beta - mean*A) and dz = dy * silu'(z):
  //   dx = rstd*(dz*gamma - (s1 + xhat*s2)/cnt) = A*dz + x*Cc + Dd,  Cc = -rstd^2*s2/cnt,  Dd = -rstd*s1/cnt + mean*rstd^2*s2/cnt
  // four per-channel coefficients in LDS, read as float4 pairs (the first version kept six and read them one float at a time: 48 LDS reads
  // per 16-byte chunk -- the kernel ran at 1.7 TB/s where the forward apply pass runs at 4.8)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* p_a = (float*)smem_raw;
  float* p_b = p_a + C;
  float* p_c = p_b + C;
  float* p_d = p_c + C;
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  for (int c = tid; c < C; c += GN_THREADS) {
    const int g = c / cpg;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const float s1 = gsum[(long)n * 2 * G + 2 * g] * inv_cnt, s2 = gsum[(long)n * 2 * G + 2 * g + 1] * inv_cnt;
    const float a = rs * gamma[c];
    p_a[c] = a;
    p_b[c] = beta[c] - mu * a;
    p_c[c] = -rs * rs * s2;
    p_d[c] = -rs * s1 + mu * rs * rs * s2;
  }
  __syncthreads();
  const int cpr = C >> 3;
  const int row_lo = blockIdx.x * rows_per;
  const int row_hi = min(HW, row_lo + rows_per);
  const unsigned total = (unsigned)(row_hi - row_lo) * (unsigned)cpr;
  const long off0 = ((long)n * HW + row_lo) * C;
  auto one = [&](unsigned i, const uint4_t& rx, const uint4_t& rd, const uint4_t& ra) {
    const int c0 = (int)(i % (unsigned)cpr) * 8;
    float fx[8], fd[8], fa[8], ca[8], cb[8], cc[8], cd[8];
    unpack8(rx, fx);
    unpack8(rd, fd);
    unpack8(ra, fa);
    *(float4_t*)ca = *(const float4_t*)(p_a + c0); *(float4_t*)(ca + 4) = *(const float4_t*)(p_a + c0 + 4);
    *(float4_t*)cc = *(const float4_t*)(p_c + c0); *(float4_t*)(cc + 4) = *(const float4_t*)(p_c + c0 + 4);
    *(float4_t*)cd = *(const float4_t*)(p_d + c0); *(float4_t*)(cd + 4) = *(const float4_t*)(p_d + c0 + 4);
    if (silu) {
      *(float4_t*)cb = *(const float4_t*)(p_b + c0); *(float4_t*)(cb + 4) = *(const float4_t*)(p_b + c0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) fd[e] *= dsilu_f(fx[e] * ca[e] + cb[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) fx[e] = ca[e] * fd[e] + (fx[e] * cc[e] + cd[e]) + fa[e];
    *(uint4_t*)(dx + off0 + (size_t)i * 8) = pack8(fx);
  };
  const uint4_t zero4 = {0u, 0u, 0u, 0u};
  unsigned i = tid;
  for (; i + GN_THREADS < total; i += 2 * GN_THREADS) {   // two chunks (4-6 loads of 16 bytes) in flight per thread
    uint4_t vx[2], vd[2], va[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t o = off0 + (size_t)(i + u * GN_THREADS) * 8;
      vx[u] = *(const uint4_t*)(x + o);
      vd[u] = *(const uint4_t*)(dy + o);
      va[u] = dx_add ? *(const uint4_t*)(dx_add + o) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) one(i + u * GN_THREADS, vx[u], vd[u], va[u]);
  }
  for (; i < total; i += GN_THREADS) {
    const size_t o = off0 + (size_t)i * 8;
    one(i, *(const uint4_t*)(x + o), *(const uint4_t*)(dy + o), dx_add ? *(const uint4_t*)(dx_add + o) : zero4);
  }
}

// channel slabs so that one thread owns one 16-byte chunk column: (C/8)/nz <= 256
static int gn_nz(int C) {
  int cpr = C >> 3, nz = (cpr + GN_THREADS - 1) / GN_THREADS;
  while (cpr % nz) ++nz;
  return nz;
}

static int gn_rows_per(int N, int HW, int* nsplit) {
  // aim for ~2048 blocks in total
  int want = (1024 + N - 1) / N;
  int rows_per = (HW + want - 1) / want;
  if (rows_per < 32) rows_per = 32;
  *nsplit = (HW + rows_per - 1) / rows_per;
  return rows_per;
}

extern "C" long nk_groupnorm_ws_floats(int N, int HW, int C, int G) {
  // fp32 elements of workspace nk_groupnorm_fwd / nk_groupnorm_bwd need (per-block partial sums)
  int nsplit;
  gn_rows_per(N, HW, &nsplit);
  long parts = (long)N * nsplit * gn_nz(C) * 2 * G;   // group partials
  long chan = (long)N * nsplit * 2 * C;                // channel partials (backward)
  return parts + chan + (long)N * 2 * G + 64;
}

extern "C" int nk_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, float* ws, int N, int HW, int C, int G, float eps, int silu,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(N > 0 && HW > 0 && C > 0 && G > 0 && G <= 64);
  NK_CHECK_ARG((C & 7) == 0 && C % G == 0 && C <= GN_MAXC);
  NK_CHECK_ARG(x && gamma && beta && y && mean && rstd && ws);
  int nsplit;
  int rows_per = gn_rows_per(N, HW, &nsplit);
  const int nz = gn_nz(C);
  float* part = ws;
  float* stats = ws + (long)N * nsplit * nz * 2 * G;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, N, nz), dim3(GN_THREADS), 0, stream, (const bf16_t*)x, part, HW, C,
                     G, rows_per);
  if (int e = nk_check_launch("gn_stats_kernel")) return e;
  hipLaunchKernelGGL(gn_reduce_partials_kernel, dim3(N), dim3(1024), 0, stream, part, stats, nsplit * nz, 2 * G);
  if (int e = nk_check_launch("gn_reduce_partials_kernel")) return e;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nsplit, N), dim3(GN_THREADS), 2 * C * sizeof(float), stream,
                     (const bf16_t*)x, stats, gamma, beta, (bf16_t*)y, mean, rstd, HW, C, G, eps, silu, rows_per);
  return nk_check_launch("gn_apply_kernel");
}

// ---- GroupNorm in separable passes: the sums, and the normalisation given the sums.  A convolution's statistics epilogue
// (conv_halo.h) produces the sums' per-tile partials itself, and its GroupNorm prologue consumes the sums: the passes below are
// what remains for tensors that come from elsewhere.  sums[n][2g] = sum, sums[n][2g+1] = sum of squares over H*W*(C/G) elements. ----
__global__ __launch_bounds__(256) void gn_reduce_partials_l1_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int G2, int nchunks) {
  // level 1 of a two-level fixed-order sum for many partial rows: block (chunk, n) sums rows chunk, chunk + nchunks, ... of image n
  __shared__ float sm[4][64];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float a = 0.f;
  if (tx < G2)
    for (int p = chunk + ty * nchunks; p < nparts; p += 4 * nchunks) a += part[((long)n * nparts + p) * G2 + tx];
  sm[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && tx < G2) out[((long)n * nchunks + chunk) * G2 + tx] = (a + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

extern "C" long nk_groupnorm_sums_ws_floats(int N, int nparts, int G) {
  (void)nparts;
  return (long)N * 64 * 2 * G + 64;
}
extern "C" int nk_groupnorm_sums_from_parts(const float* part, float* sums, float* ws, int N, int nparts, int G, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(part && sums && N > 0 && nparts > 0 && G > 0 && G <= 32);
  if (nparts > 128) {
    NK_CHECK_ARG(ws != nullptr);
    const int nchunks = 64;
    hipLaunchKernelGGL(gn_reduce_partials_l1_kernel, dim3(nchunks, N), dim3(256), 0, stream, part, ws, nparts, 2 * G, nchunks);
    if (int e = nk_check_launch("gn_reduce_partials_l1_kernel")) return e;
    part = ws;
    nparts = nchunks;
  }
  hipLaunchKernelGGL(gn_reduce_partials_kernel, dim3(N), dim3(1024), 0, stream, part, sums, nparts, 2 * G);
  return nk_check_launch("gn_reduce_partials_kernel");
}

extern "C" int nk_groupnorm_sums(const void* x, float* sums, float* ws, int N, int HW, int C, int G, void* stream_) {
  // the statistics pass of nk_groupnorm_fwd alone; ws: nk_groupnorm_ws_floats(N, HW, C, G) floats
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(N > 0 && HW > 0 && C > 0 && G > 0 && G <= 64);
  NK_CHECK_ARG((C & 7) == 0 && C % G == 0 && C <= GN_MAXC);
  NK_CHECK_ARG(x && sums && ws);
  int nsplit;
  int rows_per = gn_rows_per(N, HW, &nsplit);
  const int nz = gn_nz(C);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, N, nz), dim3(GN_THREADS), 0, stream, (const bf16_t*)x, ws, HW, C, G, rows_per);
  if (int e = nk_check_launch("gn_stats_kernel")) return e;
  hipLaunchKernelGGL(gn_reduce_partials_kernel, dim3(N), dim3(1024), 0, stream, ws, sums, nsplit * nz, 2 * G);
  return nk_check_launch("gn_reduce_partials_kernel");
}

extern "C" int nk_groupnorm_apply(const void* x, const float* sums, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int N, int HW, int C, int G, float eps, int silu, void* stream_) {
  // the normalisation pass of nk_groupnorm_fwd given the sums (from nk_groupnorm_sums or a convolution's statistics epilogue)
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(N > 0 && HW > 0 && C > 0 && G > 0 && G <= 64);
  NK_CHECK_ARG((C & 7) == 0 && C % G == 0 && C <= GN_MAXC);
  NK_CHECK_ARG(x && sums && gamma && beta && y && mean && rstd);
  int nsplit;
  int rows_per = gn_rows_per(N, HW, &nsplit);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nsplit, N), dim3(GN_THREADS), 2 * C * sizeof(float), stream, (const bf16_t*)x, sums, gamma, beta,
                     (bf16_t*)y, mean, rstd, HW, C, G, eps, silu, rows_per);
  return nk_check_launch("gn_apply_kernel");
}

extern "C" int nk_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, const void* dx_add, void* dx, float* dgamma,
                                float* dbeta, float* ws, int N, int HW, int C, int G, int silu, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(N > 0 && HW > 0 && C > 0 && G > 0 && G <= 64);
  NK_CHECK_ARG((C & 7) == 0 && C % G == 0 && C <= GN_MAXC);
  NK_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && ws);
  int nsplit;
  int rows_per = gn_rows_per(N, HW, &nsplit);
  const int nz = gn_nz(C);
  float* part = ws;
  float* gsum = part + (long)N * nsplit * nz * 2 * G;
  float* chan = gsum + (long)N * 2 * G;
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nsplit, N, nz), dim3(GN_THREADS), 2 * C * sizeof(float), stream,
                     (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, part, chan, HW, C, G, silu, rows_per);
  if (int e = nk_check_launch("gn_bwd_stats_kernel")) return e;
  hipLaunchKernelGGL(gn_reduce_partials_kernel, dim3(N), dim3(1024), 0, stream, part, gsum, nsplit * nz, 2 * G);
  if (int e = nk_check_launch("gn_reduce_partials_kernel")) return e;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, chan, dgamma, dbeta, N * nsplit, C, accumulate);
  if (int e = nk_check_launch("colpart_reduce_kernel")) return e;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nsplit, N), dim3(GN_THREADS), 4 * C * sizeof(float), stream,
                     (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, gsum, (const bf16_t*)dx_add,
                     (bf16_t*)dx, HW, C, G, silu, rows_per);
  return nk_check_launch("gn_bwd_apply_kernel");
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim of [M][C]; one wavefront per row, rows strided over the grid.
// ------------------------------------------------------------------------------------------------
#define LN_MAXCH 4   // 16-byte chunks per lane: C <= 64*8*4 = 2048

// NCH = 16-byte chunks per lane actually needed (2: C <= 1024, 3: <= 1536, 4: <= 2048): the per-lane arrays are sized by
// it, which is what decides the register count and with it how many rows a CU keeps in flight.
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int C,
                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int cpr = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row = wave; row < M; row += nwaves) {
    float f[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      if (ch < cpr) {
        unpack8(*(const uint4_t*)(x + (long)row * C + ch * 8), f[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[j][e];
      }
    }
    const float mu = wave_sum(s) * invC;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      if (ch < cpr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = f[j][e] - mu; v += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(v) * invC + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      if (ch < cpr) {
        const float4_t g0 = *(const float4_t*)(gamma + ch * 8), g1 = *(const float4_t*)(gamma + ch * 8 + 4);
        const float4_t b0 = *(const float4_t*)(beta + ch * 8), b1 = *(const float4_t*)(beta + ch * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (f[j][e] - mu) * rs * g0[e] + b0[e];
          o[4 + e] = (f[j][4 + e] - mu) * rs * g1[e] + b1[e];
        }
        *(uint4_t*)(y + (long)row * C + ch * 8) = pack8(o);
      }
    }
  }
}

// LayerNorm backward, input gradient:
// dx = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)) (+ dx_add).  One wavefront per row; x and dy stay
// in registers as packed bf16 between the statistics pass and the output pass.
template <int NCH>
__global__ __launch_bounds__(256, 4) void ln_bwd_dx_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const bf16_t* __restrict__ dx_add,
                                                        bf16_t* __restrict__ dx, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int cpr = C >> 3;
  const float invC = 1.0f / (float)C;
  float gm[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int ch = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) gm[j][e] = 0.f;
    if (ch < cpr) {
      const float4_t g0 = *(const float4_t*)(gamma + ch * 8), g1 = *(const float4_t*)(gamma + ch * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[j][e] = g0[e]; gm[j][4 + e] = g1[e]; }
    }
  }
  for (int row = wave; row < M; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    uint4_t rx[NCH], rd[NCH], ra[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      rx[j] = rd[j] = ra[j] = (uint4_t){0u, 0u, 0u, 0u};
      if (ch < cpr) {
        rx[j] = *(const uint4_t*)(x + (long)row * C + ch * 8);
        rd[j] = *(const uint4_t*)(dy + (long)row * C + ch * 8);
        if (dx_add) ra[j] = *(const uint4_t*)(dx_add + (long)row * C + ch * 8);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float fx[8], fd[8];
      unpack8(rx[j], fx);
      unpack8(rd[j], fd);
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // lanes past the row hold zeros (dy = 0), so they add nothing
        float d = fd[e] * gm[j][e];
        s1 += d;
        s2 += d * ((fx[e] - mu) * rs);
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      if (ch < cpr) {
        float fx[8], fd[8], o[8];
        unpack8(rx[j], fx);
        unpack8(rd[j], fd);
        unpack8(ra[j], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rs * (fd[e] * gm[j][e] - s1 - (fx[e] - mu) * rs * s2);
        *(uint4_t*)(dx + (long)row * C + ch * 8) = pack8(o);
      }
    }
  }
}

// LayerNorm backward, parameter gradients: part[split][0][c] = sum_rows dy*xhat, part[split][1][c] = sum_rows dy over the
// split's 64 rows.  Block = 16 row lanes x 16 chunk lanes (256 contiguous bytes per row lane, four rows in flight per
// thread); splits are combined by colpart_reduce_kernel (single writer per channel, deterministic).
#define LNP_ROWS 64
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ part, int M, int C) {
  __shared__ float pg[16][16 * 8 + 4], pb[16][16 * 8 + 4];
  const int tid = threadIdx.x;
  const int cx = tid & 15, ry = tid >> 4;
  const int chunk = blockIdx.y * 16 + cx;
  const bool cok = chunk < (C >> 3);
  const int row_lo = blockIdx.x * LNP_ROWS;
  float sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
  if (cok) {
    uint4_t vx[4], vd[4];
    float mu[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row_lo + ry + 16 * i;
      const bool ok = r < M;
      vx[i] = ok ? *(const uint4_t*)(x + (long)r * C + chunk * 8) : (uint4_t){0u, 0u, 0u, 0u};
      vd[i] = ok ? *(const uint4_t*)(dy + (long)r * C + chunk * 8) : (uint4_t){0u, 0u, 0u, 0u};
      mu[i] = ok ? mean[r] : 0.f;
      rs[i] = ok ? rstd[r] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float fx[8], fd[8];
      unpack8(vx[i], fx);
      unpack8(vd[i], fd);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sb[e] += fd[e];
        sg[e] += fd[e] * ((fx[e] - mu[i]) * rs[i]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { pg[ry][cx * 8 + e] = sg[e]; pb[ry][cx * 8 + e] = sb[e]; }
  __syncthreads();
  if (tid < 128) {
    const int c = blockIdx.y * 128 + tid;
    if (c < C) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { a += pg[r][tid]; b += pb[r][tid]; }
      float* out = part + (long)blockIdx.x * 2 * C;
      out[c] = a;
      out[C + c] = b;
    }
  }
}

static int ln_param_split(int M) { return (M + LNP_ROWS - 1) / LNP_ROWS; }

// LayerNorm backward in ONE pass over x and dy (round 5; replaces the round-2 one-row-per-wave fused kernel, whose per-wave LDS fold cost
// more than the row it had processed).  A wave walks SEVERAL rows (row = wave, wave + nwaves, ...: 4 at M = 4096, 8 at M = 16384), the next
// row's x / dy / residual-gradient loads are issued before the current row's reductions (two rows in flight per wave: these 10-40 MB
// tensors are latency-bound), dx is produced as in ln_bwd_dx_kernel, and -- from the same registers -- the wave keeps column partials of
// dgamma (sum dy * xhat) and dbeta (sum dy) over ITS rows.  At the end the block's four waves fold through two LDS slots in a fixed order
// ((w0 + w2) + (w1 + w3): bit-reproducible) into ONE partial row per block: <= 512 rows, 1.3-2.6 MB, against the 21 MB the separate
// parameter kernel re-read.  colpart_reduce_kernel / colpart_reduce_batch_kernel sum the partial rows (single writer per channel).
// Algorithmic traffic 6 B/elem (+2 with dx_add); the three-kernel form moved 10 (+2) and took three launches.
#define LNR_WAVES 4      // (8 waves x 2 rows: 10.2 vs 11.1 us alone at 4096 x 1280, 20.5 vs 18.0 at 16384 x 640, and the smaller A/B gain in the step)
#define LNR_MAX_BLOCKS 512
template <int NCH>
__global__ __launch_bounds__(LNR_WAVES * 64) void ln_bwd_rows_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, const bf16_t* __restrict__ dx_add,
                                                                       bf16_t* __restrict__ dx, float* __restrict__ part, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) char ln_smem[];
  float* red = (float*)ln_smem;                     // [2 slots][2][C]: 20 KB at C = 1280
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * LNR_WAVES + wv;
  const int nwaves = gridDim.x * LNR_WAVES;
  const int cpr = C >> 3;
  const float invC = 1.0f / (float)C;
  float gm[NCH][8], sg[NCH][8], sb[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int ch = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[j][e] = 0.f; sg[j][e] = 0.f; sb[j][e] = 0.f; }
    if (ch < cpr) {
      const float4_t g0 = *(const float4_t*)(gamma + ch * 8), g1 = *(const float4_t*)(gamma + ch * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[j][e] = g0[e]; gm[j][4 + e] = g1[e]; }
    }
  }
  uint4_t rx[NCH], rd[NCH], ra[NCH];
  float mu = 0.f, rs = 0.f;
  auto fetch = [&](int row, uint4_t (&fx)[NCH], uint4_t (&fd)[NCH], uint4_t (&fa)[NCH], float& fmu, float& frs) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = lane + 64 * j;
      fx[j] = fd[j] = fa[j] = (uint4_t){0u, 0u, 0u, 0u};
      if (row < M && ch < cpr) {
        fx[j] = *(const uint4_t*)(x + (long)row * C + ch * 8);
        fd[j] = *(const uint4_t*)(dy + (long)row * C + ch * 8);
        if (dx_add) fa[j] = *(const uint4_t*)(dx_add + (long)row * C + ch * 8);
      }
    }
    fmu = row < M ? mean[row] : 0.f;
    frs = row < M ? rstd[row] : 0.f;
  };
  constexpr bool AHEAD = NCH < 4;      // (C > 1536 -- no SDXL / SD1.5 LayerNorm -- would spill with two rows in registers: one row at a time)
  if (AHEAD) fetch(wave, rx, rd, ra, mu, rs);
  for (int row = wave; row < M; row += nwaves) {
    uint4_t nx[NCH], nd[NCH], na[NCH];
    float nmu = 0.f, nrs = 0.f;
    if (AHEAD) fetch(row + nwaves, nx, nd, na, nmu, nrs);        // the next row's loads are in flight under this row's arithmetic
    else fetch(row, rx, rd, ra, mu, rs);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float fx[8], fd[8];
      unpack8(rx[j], fx);
      unpack8(rd[j], fd);
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // lanes past the row hold zeros (dy = 0), so they add nothing
        const float xh = (fx[e] - mu) * rs;
        const float d = fd[e] * gm[j][e];
        s1 += d;
        s2 += d * xh;
        sb[j][e] += fd[e];
        sg[j][e] += fd[e] * xh;
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int ch = lane + 64 * j;
      if (ch < cpr) {
        float fx[8], fd[8], o[8];
        unpack8(rx[j], fx);
        unpack8(rd[j], fd);
        unpack8(ra[j], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rs * (fd[e] * gm[j][e] - s1 - (fx[e] - mu) * rs * s2);
        *(uint4_t*)(dx + (long)row * C + ch * 8) = pack8(o);
      }
    }
    if (AHEAD) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) { rx[j] = nx[j]; rd[j] = nd[j]; ra[j] = na[j]; }
      mu = nmu;
      rs = nrs;
    }
  }
  // this wave's column partials -> LDS -> one row per block.  Waves 0, 1 write slots 0, 1; waves 2, 3 add into them; then slot 0 + slot 1.
#pragma unroll
  for (int round = 0; round < LNR_WAVES / 2; ++round) {
    if ((wv >> 1) == round) {
      float* gbase = red + ((long)(wv & 1) * 2) * C;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        int ch = lane + 64 * j;
        if (ch < cpr) {
          float* g = gbase + ch * 8;
          float4_t g0 = {sg[j][0], sg[j][1], sg[j][2], sg[j][3]}, g1 = {sg[j][4], sg[j][5], sg[j][6], sg[j][7]};
          float4_t b0 = {sb[j][0], sb[j][1], sb[j][2], sb[j][3]}, b1 = {sb[j][4], sb[j][5], sb[j][6], sb[j][7]};
          if (round) { g0 += *(const float4_t*)g; g1 += *(const float4_t*)(g + 4); b0 += *(const float4_t*)(g + C); b1 += *(const float4_t*)(g + C + 4); }
          *(float4_t*)g = g0;
          *(float4_t*)(g + 4) = g1;
          *(float4_t*)(g + C) = b0;
          *(float4_t*)(g + C + 4) = b1;
        }
      }
    }
    __syncthreads();
  }
  float* out = part + (long)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += LNR_WAVES * 64) out[c] = red[c] + red[2 * C + c];
}
// one block per 16 rows (four per wave), at most 512 blocks (two per CU; 8 rows per wave at M = 16384)
static int ln_rows_blocks(int M) {
  int b = (M + 4 * LNR_WAVES - 1) / (4 * LNR_WAVES);
  return b < 1 ? 1 : (b > LNR_MAX_BLOCKS ? LNR_MAX_BLOCKS : b);
}

// Several column-partial reductions in ONE launch (blockIdx.y = entry): the three LayerNorms of a transformer block hand their partial
// rows to the weight-gradient queue, which reduces them behind the block's batched weight gradients (ops.WgradQueue).
__global__ __launch_bounds__(1024) void colpart_reduce_batch_kernel(const NkColpartBatch b) {
  // block = 32 channels x 32 row groups: 120 workgroups for three 1280-wide LayerNorms, eight rows per thread at 256 partial rows, all of them
  // in flight at once (the 64 x 16 layout of colpart_reduce_kernel measured 10.4 us per launch in the step: 60 workgroups, 16 dependent rounds)
  __shared__ float sa[32][33], sb[32][33];
  const int z = blockIdx.y;
  const int C = b.C[z], nrows = b.nrows[z];
  if ((int)blockIdx.x * 32 >= C) return;
  const float* __restrict__ part = b.part[z];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float a = 0.f, v = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int r = ty; r < nrows; r += 32) {
      a += part[(long)r * 2 * C + c];
      v += part[(long)r * 2 * C + C + c];
    }
  }
  sa[ty][tx] = a;
  sb[ty][tx] = v;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int j = 1; j < 32; ++j) { a += sa[j][tx]; v += sb[j][tx]; }
    float* dgamma = b.dgamma[z];
    float* dbeta = b.dbeta[z];
    dgamma[c] = b.accumulate[z] ? dgamma[c] + a : a;
    dbeta[c] = b.accumulate[z] ? dbeta[c] + v : v;
  }
}

extern "C" int nk_colpart_reduce_batch(const NkColpartBatch* b, void* stream_) {
  NK_CHECK_ARG(b && b->n > 0 && b->n <= NK_COLPART_MAX);
  int maxC = 0;
  for (int z = 0; z < b->n; ++z) {
    NK_CHECK_ARG(b->part[z] && b->dgamma[z] && b->dbeta[z] && b->nrows[z] > 0 && b->C[z] > 0);
    maxC = b->C[z] > maxC ? b->C[z] : maxC;
  }
  hipLaunchKernelGGL(colpart_reduce_batch_kernel, dim3((maxC + 31) / 32, b->n), dim3(1024), 0, (hipStream_t)stream_, *b);
  return nk_check_launch("colpart_reduce_batch_kernel");
}

extern "C" int nk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int M, int C, float eps, void* stream) {
  NK_CHECK_ARG(M > 0 && C > 0 && (C & 7) == 0 && (C >> 3) <= 64 * LN_MAXCH);
  NK_CHECK_ARG(x && gamma && beta && y && mean && rstd);
  int blocks = min((M + 3) / 4, 4096);
  const int nch = ((C >> 3) + 63) / 64;
#define NK_LN_FWD(NCH_) hipLaunchKernelGGL(ln_fwd_kernel<NCH_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps)
  if (nch <= 2) NK_LN_FWD(2); else if (nch == 3) NK_LN_FWD(3); else NK_LN_FWD(4);
#undef NK_LN_FWD
  return nk_check_launch("ln_fwd_kernel");
}

extern "C" long nk_layernorm_ws_floats(int M, int C) {
  const long a = (long)ln_param_split(M), b = LNR_MAX_BLOCKS;
  return (a > b ? a : b) * 2 * C + 64;
}

extern "C" long nk_layernorm_part_rows(int M) { return ln_rows_blocks(M); }

extern "C" int nk_layernorm_bwd_dx(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                   const void* dx_add, void* dx, int M, int C, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(M > 0 && C > 0 && (C & 7) == 0 && (C >> 3) <= 64 * LN_MAXCH);
  NK_CHECK_ARG(dy && x && gamma && mean && rstd && dx);
  // input gradient: one wavefront per row.  (Fewer, longer-lived workgroups -- a grid capped at 256 / 512 / 1024 blocks with the rows
  // strided over them -- make no difference beside the weight-gradient stream: 180.4 / 180.3 / 180.2 vs 180.1 ms per step.)
  const int blocks = min((M + 3) / 4, 4096);
  const int nch = ((C >> 3) + 63) / 64;
#define NK_LN_DX(NCH_) hipLaunchKernelGGL(ln_bwd_dx_kernel<NCH_>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dx_add, (bf16_t*)dx, M, C)
  if (nch <= 2) NK_LN_DX(2); else if (nch == 3) NK_LN_DX(3); else NK_LN_DX(4);
#undef NK_LN_DX
  return nk_check_launch("ln_bwd_dx_kernel");
}

extern "C" int nk_layernorm_bwd_params(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma,
                                       float* dbeta, float* ws, int M, int C, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(M > 0 && C > 0 && (C & 7) == 0);
  NK_CHECK_ARG(dy && x && mean && rstd && dgamma && dbeta && ws);
  // parameter gradients: column-parallel partial sums over row splits, then a single-writer reduce
  const int nsplit = ln_param_split(M);
  hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(nsplit, (C + 127) / 128), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                     mean, rstd, ws, M, C);
  if (int e = nk_check_launch("ln_bwd_param_kernel")) return e;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, ws, dgamma, dbeta, nsplit, C, accumulate);
  return nk_check_launch("colpart_reduce_kernel");
}

static int launch_ln_rows(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dx_add,
                          void* dx, float* part, int M, int C, hipStream_t stream) {
  const int blocks = ln_rows_blocks(M);
  const int nch = ((C >> 3) + 63) / 64;
  const int smem = 2 * 2 * C * (int)sizeof(float);
#define NK_LN_ROWS(NCH_)                                                                                                       \
  do {                                                                                                                          \
    nk_optin_lds((const void*)ln_bwd_rows_kernel<NCH_>, 2 * 2 * 2048 * 4);                                                      \
    hipLaunchKernelGGL(ln_bwd_rows_kernel<NCH_>, dim3(blocks), dim3(LNR_WAVES * 64), smem, stream, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,   \
                       (const bf16_t*)dx_add, (bf16_t*)dx, part, M, C);                                                        \
  } while (0)
  if (nch <= 2) NK_LN_ROWS(2); else if (nch == 3) NK_LN_ROWS(3); else NK_LN_ROWS(4);
#undef NK_LN_ROWS
  return nk_check_launch("ln_bwd_rows_kernel");
}

/* dx and nk_layernorm_part_rows(M) partial rows [rows][2][C] (dgamma partials, dbeta partials) in one pass; the caller reduces the rows with
 * nk_colpart_reduce_batch (any stream, any later time: `part` is the only state) */
extern "C" int nk_layernorm_bwd_rows(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                     const void* dx_add, void* dx, float* part, int M, int C, void* stream_) {
  NK_CHECK_ARG(M > 0 && C > 0 && (C & 7) == 0 && (C >> 3) <= 64 * LN_MAXCH);
  NK_CHECK_ARG(dy && x && gamma && mean && rstd && dx && part);
  return launch_ln_rows(dy, x, gamma, mean, rstd, dx_add, dx, part, M, C, (hipStream_t)stream_);
}

extern "C" int nk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta, float* ws,
                                int M, int C, int accumulate, void* stream_) {
  // one pass over x and dy for the input gradient AND the per-block parameter partials, then the single-writer column reduce
  hipStream_t stream = (hipStream_t)stream_;
  NK_CHECK_ARG(M > 0 && C > 0 && (C & 7) == 0 && (C >> 3) <= 64 * LN_MAXCH);
  NK_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ws);
  if (int e = launch_ln_rows(dy, x, gamma, mean, rstd, dx_add, dx, ws, M, C, stream)) return e;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, ws, dgamma, dbeta, ln_rows_blocks(M), C, accumulate);
  return nk_check_launch("colpart_reduce_kernel");
}
