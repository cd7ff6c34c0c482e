// Flash BACKWARD for head dim 512 (included by attention.hip; not a stand-alone translation unit) -- round 5.
//
// Reference: AttnBlock / TorchSDPAttnBlock of the VAE, modules/diffusion/model.py:155-166, 224-243, trained by models/autoencoder.py:280-293
// (BASELINE config 5).  The forward is attn512.h (flash, log-sum-exp out); rounds 2-4 trained through a two-GEMM forward with saved
// probabilities, round 5's first form recomputed them chunk by chunk through HBM (ops.attention512_fwd).  Here nothing of size L x L exists
// anywhere: scores are recomputed tile by tile in registers from Q, K and the forward's log-sum-exp.
//
// ONE kernel template serves both halves of the backward, because they are the same program with the roles of queries and keys exchanged:
//   a workgroup OWNS 32 rows (queries for dQ; keys for dK / dV) and walks the OTHER side in tiles of 32 rows;
//   stage 1:  X = A1own T1^T,  Y = A2own T2^T   [32 own x 32 other], contraction over the 512 head dims
//                 dQ     : own = (Q, dO),  tiles = (K, V):   X = S,    Y = dP
//                 dK/dV  : own = (K, V),   tiles = (Q, dO):  X = S^T,  Y = dP^T
//             P = exp2(X c - lse2[query]),  dS = P (Y - delta[query]) scale      (the query is the ROW for dQ, the COLUMN for dK / dV)
//   stage 2:  acc[32 own x 512] += W [32 own x 32 other] Tile[32 other x 512]
//                 dQ     : dQ += dS T1 (K)
//                 dK/dV  : dV += P^T T2 (dO),  dK += dS^T T1 (Q)
// with lse2 = LSE log2 e, c = scale log2 e, delta = rowsum(dO o O) (attn512_delta_kernel).  dQ = scale dS' K, dK = scale dS'^T Q, dV = P^T dO.
//
// Layout.  v_mfma_f32_16x16x32_bf16 throughout.  4 waves; in stage 1 wave w computes the 16 x 16 block (w >> 1, w & 1) of X and of Y
// (16 + 16 MFMAs over the 16 k-chunks of 32 head dims; the workgroup's own rows sit in LDS in the tiles' layout), writes W as bf16 into a [32][32] LDS image (two of them for dK / dV); in stage 2 wave w owns head dims [128 w, 128 w + 128) of the
// accumulators (2 row blocks x 8 column blocks per accumulator).  The OTHER side's tiles live in LDS as [32 rows][512] bf16 with the 16-byte
// chunks of a row XOR-swizzled by (row & 15): ONE image is read by rows (ds_read_b128: stage 1's B operand, k = head dim) and by columns
// (ds_read_b64_tr_b16: stage 2's B operand, k = the tile's row), as the head-dim-64 kernels do.  128 KiB of tiles + 5 KiB of W images per
// workgroup.  Three barriers per tile, tiles staged through registers one tile ahead: this is a CORRECT flash backward sized for config 5 (L = 1024: 5.4
// GFLOP per sample), not a tuned one -- attn512.h's physically addressed accumulator file is what a tuned one would need.
#pragma once

#define A5B_ROWS 32
#define A5B_TILE (A5B_ROWS * 1024)          // bytes of one [32][512] bf16 tile
#define A5B_WROW 80                         // bytes per row of a W image [32][32] bf16 (64 + 16: 16-byte aligned rows, spread over banks)
#define A5B_SMEM (4 * A5B_TILE + 2 * A5B_ROWS * A5B_WROW)      // two tiles of the other side, the two own blocks, two W images: 133 KiB

// delta[b][h][q] = sum_d dO[q][d] O[q][d]  (fp32), one wave per row
__global__ __launch_bounds__(256) void attn512_delta_kernel(const AttnParams p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nrows = (long)p.B * p.H * p.Lq;
  if (row >= nrows) return;
  const int q = (int)(row % p.Lq);
  const long bh = row / p.Lq;
  const int hd = (int)(bh % p.H), b = (int)(bh / p.H);
  const bf16_t* o = p.Oc + (long)b * p.bo + (long)q * p.so + (long)hd * 512 + lane * 8;
  const bf16_t* g = p.dO + (long)b * p.bdo + (long)q * p.sdo + (long)hd * 512 + lane * 8;
  float fo[8], fg[8];
  unpack8(*(const uint4_t*)o, fo);
  unpack8(*(const uint4_t*)g, fg);
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc += fo[e] * fg[e];
  acc = wave_sum(acc);
  if (lane == 0) p.delta[row] = acc;
}

__device__ __forceinline__ unsigned a5b_tile_addr(int row, int chunk) { return (unsigned)(row * 1024 + ((chunk ^ (row & 15)) << 4)); }

// the rows [row0, row0 + 32) of `src` (row stride `stride` elements, rows >= nrows read as zero): 8 x 16 bytes per thread into registers ...
__device__ __forceinline__ void a5b_fetch(uint4_t (&v)[8], const bf16_t* src, long stride, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, ch = idx & 63;
    v[i] = (uint4_t){0u, 0u, 0u, 0u};
    if (row0 + row < nrows) v[i] = *(const uint4_t*)(src + (long)(row0 + row) * stride + ch * 8);
  }
}
// ... and from there into the swizzled tile image
__device__ __forceinline__ void a5b_put(char* tile, const uint4_t (&v)[8], int tid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, ch = idx & 63;
    *(uint4_t*)(tile + a5b_tile_addr(row, ch)) = v[i];
  }
}
__device__ __forceinline__ void a5b_stage(char* tile, const bf16_t* src, long stride, int row0, int nrows, int tid) {
  uint4_t v[8];
  a5b_fetch(v, src, stride, row0, nrows, tid);
  a5b_put(tile, v, tid);
}

// B operand of stage 1: rows 16 cb + (lane & 15) of the tile, head dims 32 kc + 8 (lane >> 4) .. + 7
__device__ __forceinline__ bf16x8_t a5b_rowfrag(const char* tile, int cb, int kc, int lane) {
  const int row = 16 * cb + (lane & 15);
  return *(const bf16x8_t*)(tile + a5b_tile_addr(row, 4 * kc + (lane >> 4)));
}
// B operand of stage 2: n = head dim d0 + (lane & 15), k = tile row 8 (lane >> 4) + j -- the transposing read of gemm.hip's r-contiguous
// operands (lane i of a 16-lane group supplies row 8 g + (i >> 2), columns d0 + 4 (i & 3) .. + 3; it receives column d0 + i of rows 8 g .. 8 g + 3)
__device__ __forceinline__ bf16x8_t a5b_colfrag(const char* tile, int d0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = d0 + 4 * pp;
  const int r_lo = 8 * g + q, r_hi = r_lo + 4;
  typedef __attribute__((address_space(3))) short4_t* lds_p;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + a5b_tile_addr(r_lo, col >> 3) + (col & 7) * 2));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + a5b_tile_addr(r_hi, col >> 3) + (col & 7) * 2));
  short8_t r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// DKDV = 0: dQ of 32 queries per workgroup (tiles = keys).  DKDV = 1: dK and dV of 32 keys per workgroup (tiles = queries).
template <int DKDV>
__global__ __launch_bounds__(256, 1) void attn512_bwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const t1 = smem;                               // K (dQ) | Q (dK/dV)
  char* const t2 = smem + A5B_TILE;                    // V (dQ) | dO (dK/dV)
  char* const o1 = smem + 2 * A5B_TILE;                // the own block of Q (dQ) | K (dK/dV)
  char* const o2 = smem + 3 * A5B_TILE;                //                  dO (dQ) | V (dK/dV)
  char* const w1 = smem + 4 * A5B_TILE;                // dS (dQ) | dS^T (dK/dV)
  char* const w2 = w1 + A5B_ROWS * A5B_WROW;           //         | P^T  (dK/dV)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, cb = wave & 1;             // stage 1: this wave's 16 x 16 block of X and Y
  const int bx = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int own0 = bx * A5B_ROWS;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * 512;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 512;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 512;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * 512;
  const float* lse = p.LSE + ((long)b * p.H + hd) * p.Lq;
  const float* dlt = p.delta + ((long)b * p.H + hd) * p.Lq;

  const bf16_t* A1 = DKDV ? Kb : Qb;   const long s1 = DKDV ? p.sk : p.sq;     // own operands
  const bf16_t* A2 = DKDV ? Vb : dOb;  const long s2 = DKDV ? p.sv : p.sdo;
  const bf16_t* T1 = DKDV ? Qb : Kb;   const long st1 = DKDV ? p.sq : p.sk;    // tiles
  const bf16_t* T2 = DKDV ? dOb : Vb;  const long st2 = DKDV ? p.sdo : p.sv;
  const int n_own = DKDV ? p.Lk : p.Lq, n_other = DKDV ? p.Lq : p.Lk;

  // the workgroup's own 32 rows of both operands, staged once (the same swizzled image as the tiles; stage 1 reads its A fragments from them:
  // held in registers instead -- 2 x 16 x 16 bytes per lane -- they pushed the dK / dV instance to 155 spilled registers)
  a5b_stage(o1, A1, s1, own0, n_own, tid);
  a5b_stage(o2, A2, s2, own0, n_own, tid);
  // dQ: the statistics belong to the OWN rows (queries): this lane's four accumulator rows (lane >> 4) * 4 + r of row block rb
  float lse_own[4], dlt_own[4];
  if (!DKDV) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = own0 + rb * 16 + (lane >> 4) * 4 + r;
      lse_own[r] = q < p.Lq ? -lse[q] * LOG2E : 0.f;
      dlt_own[r] = q < p.Lq ? dlt[q] : 0.f;
    }
  }

  float4_t acc1[2][8], acc2[2][8];                     // stage-2 accumulators: [row block][16-wide column block of this wave's 128 head dims]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc1[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f}; }

  const int ntile = (n_other + A5B_ROWS - 1) / A5B_ROWS;
  // the tiles are fetched ONE AHEAD into registers (2 x 8 x 16 bytes per thread): a tile's global-memory latency runs under the previous
  // tile's two stages (fetched where they are needed the loop took 8.6 us per tile at L = 4096 -- two dependent memory round trips)
  uint4_t pf1[8], pf2[8];
  a5b_fetch(pf1, T1, st1, 0, n_other, tid);
  a5b_fetch(pf2, T2, st2, 0, n_other, tid);
  for (int t = 0; t < ntile; ++t) {
    const int o0 = t * A5B_ROWS;
    __syncthreads();                                   // the previous tile's stage-2 reads of t1 / t2 / w1 / w2 are done
    a5b_put(t1, pf1, tid);
    a5b_put(t2, pf2, tid);
    if (t + 1 < ntile) {
      a5b_fetch(pf1, T1, st1, o0 + A5B_ROWS, n_other, tid);
      a5b_fetch(pf2, T2, st2, o0 + A5B_ROWS, n_other, tid);
    }
    __syncthreads();
    // ---- stage 1: this wave's block of X and Y ----
    float4_t x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a5b_rowfrag(o1, rb, kc, lane), a5b_rowfrag(t1, cb, kc, lane), x, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a5b_rowfrag(o2, rb, kc, lane), a5b_rowfrag(t2, cb, kc, lane), y, 0, 0, 0);
    }
    // accumulator element r of this lane: own row rb * 16 + (lane >> 4) * 4 + r, other row (column) cb * 16 + (lane & 15)
    const int ocol = o0 + cb * 16 + (lane & 15);
    const bool col_ok = ocol < n_other;
    float lcol = 0.f, dcol = 0.f;
    if (DKDV && col_ok) { lcol = -lse[ocol] * LOG2E; dcol = dlt[ocol]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l2 = DKDV ? lcol : lse_own[r];
      const float dl = DKDV ? dcol : dlt_own[r];
      float pr = col_ok ? EXP2(x[r] * c + l2) : 0.f;
      const float ds = pr * (y[r] - dl) * p.scale;
      const int wrow = rb * 16 + (lane >> 4) * 4 + r, wcol = cb * 16 + (lane & 15);
      *(bf16_t*)(w1 + wrow * A5B_WROW + wcol * 2) = f2bf(ds);
      if (DKDV) *(bf16_t*)(w2 + wrow * A5B_WROW + wcol * 2) = f2bf(pr);
    }
    __syncthreads();
    // ---- stage 2: acc[own x 128 head dims of this wave] += W [own x 32 other] Tile[32 other x head dims] ----
    bf16x8_t wa1[2], wa2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 16 + (lane & 15);
      wa1[i] = *(const bf16x8_t*)(w1 + row * A5B_WROW + (lane >> 4) * 16);
      if (DKDV) wa2[i] = *(const bf16x8_t*)(w2 + row * A5B_WROW + (lane >> 4) * 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d0 = wave * 128 + j * 16;
      const bf16x8_t bt1 = a5b_colfrag(t1, d0, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[i], bt1, acc1[i][j], 0, 0, 0);
      if (DKDV) {
        const bf16x8_t bt2 = a5b_colfrag(t2, d0, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa2[i], bt2, acc2[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: accumulator element r = row i * 16 + (lane >> 4) * 4 + r, head dim wave * 128 + j * 16 + (lane & 15) ----
  bf16_t* out1 = DKDV ? p.dK + (long)b * p.bdk + (long)hd * 512 : p.dQ + (long)b * p.bdq + (long)hd * 512;
  const long so1 = DKDV ? p.sdk : p.sdq;
  bf16_t* out2 = p.dV + (long)b * p.bdv + (long)hd * 512;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = own0 + i * 16 + (lane >> 4) * 4 + r;
      if (row >= n_own) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = wave * 128 + j * 16 + (lane & 15);
        out1[(long)row * so1 + d] = f2bf(acc1[i][j][r]);
        if (DKDV) out2[(long)row * p.sdv + d] = f2bf(acc2[i][j][r]);
      }
    }
}
