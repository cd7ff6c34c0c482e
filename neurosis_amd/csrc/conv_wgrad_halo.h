// Weight gradient of a 3 x 3 / stride-1 / padding-1 convolution from LDS halo tiles (included by gemm.hip; not a stand-alone translation unit).
//
//   dW[co][tap][ci] (+)= sum over pixels  dy[n, h, w, co] * x[n, h + kh - 1, w + kw - 1, ci]        (reference: autograd of F.conv2d,
//   openaimodel.py:247-283 / model.py:116-134 -- every ResBlock / ResnetBlock convolution of the UNet and the VAE)
//
// Why it exists.  As an implicit GEMM (nk_gemm_dma_kernel<OP_MC, OP_MCG>) the reduction runs over pixels and the B operand is the im2col
// gather of x: every input byte is staged NINE times (once per tap) and dy once per 128-column tile of the (tap, ci) axis, all through the
// 64 B/clk/CU global -> LDS path that bounds the 128 x 128 tile anyway: 400-760 TFLOP/s in the step.  Here a workgroup owns a
// block of 128 output channels x 64 input channels x ALL NINE taps and walks pixel tiles: per tile of 4 x 32 pixels it stages the dy tile
// (128 px x 128 co) and the x halo (6 x 34 px x 64 ci) ONCE, and the nine taps are nine shifted windows of the same halo image -- 68 KB staged
// per 9.4 M MACs = 15 B/clk/CU at the MFMA rate.
//
//   * 8 waves = 4 (32 co) x 2 (32 ci); a wave keeps nine 32 x 32 fp32 accumulators (144 registers), one per tap, for its whole pixel range.
//   * The contraction index is the PIXEL: both operands lie pixel-major in memory (dy [px][co], x [px][ci]) and are staged as they lie by
//     LDS-DMA; fragments come out through the transposing ds_read_b64_tr_b16.  A k-group is 16 consecutive pixels of one image row.
//   * x halo image: [6 rows][48 slots][128 B].  The row pitch of 48 slots (34 used) makes a k-group's first slot = kw (mod 16) for every row
//     and column half, so a lane needs three address sets (kw = 0, 1, 2) and every other displacement is an instruction immediate.  16-byte
//     chunks are XOR-swizzled on the SOURCE side of the DMA with the attention kernels' f(slot) = ((slot>>1)&1)<<2 | (slot>>2)&3: the
//     transposed reads are conflict-free at all three alignments (checked by enumeration).  dy image: [128 px][256 B], chunk ^ ((px&3)<<2 | (px>>2)&3).
//   * double-buffered stages (2 x 68 KB), one barrier per pixel tile (72 MFMAs of 32x32x16 per wave between barriers).
//   * the pixel range is split over S workgroups per block when the (co, ci) blocks alone do not fill the chip; partial sums meet in dW through
//     fp32 atomics (memory-side, ~1.3 TB/s: the host picks S from a cost model that prices them).
//   * the bias gradient rides along: dy fragments are summed on the VALU beside the MFMAs; the waves of the first ci block that hold ci
//     sub-block 0 write theirs.
#pragma once

#define WH_TH 4
#define WH_TW 32
#define WH_HS 48                                   // halo slots per row (34 used)
#define WH_X_BYTES ((WH_TH + 2) * WH_HS * 128)     // 36864
#define WH_DY_BYTES (WH_TH * WH_TW * 256)          // 32768
#define WH_STAGE (WH_X_BYTES + WH_DY_BYTES)        // 69632
#define WH_SMEM (2 * WH_STAGE)                     // 139264
#define WH_BCO 128
#define WH_BCI 64

typedef __attribute__((address_space(3))) char* wh_lds;
typedef __attribute__((address_space(3))) short4_t* wh_lds4;

__device__ __forceinline__ int wh_swz_x(int slot) { return (((slot >> 1) & 1) << 2) | ((slot >> 2) & 3); }
__device__ __forceinline__ int wh_swz_dy(int px) { return ((px & 3) << 2) | ((px >> 2) & 3); }

__device__ __forceinline__ bf16x8_t wh_read(wh_lds lo, wh_lds hi) {
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wh_lds4)lo);
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wh_lds4)hi);
  short8_t r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// The 62 one-KiB pieces of a stage (30 of the x halo: row pc / 5, slots 8 (pc % 5) ..; 32 of the dy tile: pixels 4 q ..) over 8 waves: piece
// pc = wave + 8 i.  A lane keeps four constants; the piece's own terms (halo row, slot group, pixel group) are wave-uniform scalars, so a
// piece costs ~8 vector instructions per tile and no registers of its own (eight precomputed offsets per lane spilled, and a scratch reload
// in the issue phase waits for the DMAs just issued: vmcnt counts both).
//   x:  slot = 8 pg + l8, chunk = (lane & 7) ^ f(slot) = xd ^ ((pg & 1) << 1)         with xd = (lane & 7) ^ (((l8 >> 1) & 1) << 2 | l8 >> 2)
//   dy: px = 4 q + l16,   chunk = (lane & 15) ^ ((px & 3) << 2 | (px >> 2) & 3) = dd ^ (q & 3)   with dd = (lane & 15) ^ (l16 << 2)
struct WhStager {
  int xd, xe, dd, de;      // xe = l8 * Cin, de = l16 * Cout: the lane's pixel within the piece, in elements
  int n, ty, tx;           // the tile about to be issued

  __device__ __forceinline__ void init(const NkGemmParams& p, int t, int txn, int tyn, int lane) {
    const int per_img = txn * tyn;
    n = t / per_img;
    const int rem = t - n * per_img;
    ty = rem / txn;
    tx = rem - ty * txn;
    const int l8 = lane >> 3, l16 = lane >> 4;
    xd = (lane & 7) ^ ((((l8 >> 1) & 1) << 2) | (l8 >> 2));
    xe = l8 * p.gb.C;
    dd = (lane & 15) ^ (l16 << 2);
    de = l16 * (int)p.lda;
  }
  __device__ __forceinline__ void issue(const NkGemmParams& p, char* stage, int txn, int tyn, int cb, int ib, int wave, int lane) {
    const NkGather& g = p.gb;
    const int h0 = ty * WH_TH, w0 = tx * WH_TW;
    const bf16_t* xo = p.B + (((long)n * g.H + h0 - 1) * g.W + (w0 - 1)) * g.C + ib * WH_BCI;
    const bf16_t* dyo = p.A + (((long)n * g.H + h0) * g.W + w0) * p.lda + cb * WH_BCO;
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
    int l8 = lane >> 3, l16 = lane >> 4, xd_ = xd, xe_ = xe, dd_ = dd, de_ = de;
    // (opaque to the optimiser: everything below is loop-invariant per piece, and hoisted out of the tile loop it costs 16+ registers)
    asm volatile("" : "+v"(l8), "+v"(l16), "+v"(xd_), "+v"(xe_), "+v"(dd_), "+v"(de_));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pc = wave + 8 * i;                          // wave-uniform
      if (pc < 30) {
        const int hr = pc / 5, pg = pc - hr * 5;
        const bool row_ok = (unsigned)(h0 - 1 + hr) < (unsigned)g.H;
        const int slot = pg * 8 + l8;
        const bool v = row_ok && slot < WH_TW + 2 && (unsigned)(w0 - 1 + slot) < (unsigned)g.W;
        const int off = (hr * g.W + pg * 8) * g.C + xe_ + ((xd_ ^ ((pg & 1) << 1)) << 3);
        __builtin_amdgcn_global_load_lds((nk_gptr)(v ? xo + off : zp), (nk_lptr)(stage + (hr * WH_HS + pg * 8) * 128), 16, 0, 0);
      } else if (pc < 62) {
        const int q = pc - 30;
        const bool row_ok = h0 + (q >> 3) < g.H;
        const int chunk = dd_ ^ (q & 3);
        const bool v = row_ok && w0 + (q & 7) * 4 + l16 < g.W && cb * WH_BCO + chunk * 8 < p.M;
        const int off = ((q >> 3) * g.W + (q & 7) * 4) * (int)p.lda + de_ + (chunk << 3);
        __builtin_amdgcn_global_load_lds((nk_gptr)(v ? dyo + off : zp), (nk_lptr)(stage + WH_X_BYTES + q * 1024), 16, 0, 0);
      }
    }
    if (++tx == txn) { tx = 0; if (++ty == tyn) { ty = 0; ++n; } }       // the next tile (scalar: no division in the loop)
  }
};

template <int BIAS>
__global__ __launch_bounds__(512, 2) void nk_conv3x3_wgrad_halo_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cs = wave & 3, is = wave >> 2;        // 32-channel sub-blocks of co / ci
  const NkGather& g = p.gb;
  const int Cin = g.C, Cout = p.M;
  const int nci = Cin / WH_BCI;
  const int nblk = ((Cout + WH_BCO - 1) / WH_BCO) * nci;
  const int blk = blockIdx.x % nblk, split = blockIdx.x / nblk;
  const int cb = blk / nci, ib = blk - cb * nci;
  const int txn = (g.W + WH_TW - 1) / WH_TW, tyn = (g.H + WH_TH - 1) / WH_TH;
  const int T = p.halo_nb * txn * tyn;
  const int t0 = split * p.ksplit_len, t1 = min(T, t0 + p.ksplit_len);

  float16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;       // BIAS: every wave sums its dy fragments (16 VALU instructions beside 9 MFMAs; a branch here would fence the
                          // scheduler between k-groups); the waves that own the bias gradient store theirs at the end

  // per-lane fragment addresses (stage-relative): x at the three k-group alignments, dy
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
  unsigned xo[3][2], dyo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int row = a + 4 * h + q4 + 8 * j;
      xo[a][j] = (unsigned)(row * 128 + (((4 * is + 2 * g1 + (pp >> 1)) ^ wh_swz_x(row)) << 4) + 8 * (pp & 1));
    }
    const int row = 4 * h + q4 + 8 * j;
    dyo[j] = (unsigned)(WH_X_BYTES + row * 256 + (((4 * cs + 2 * g1 + (pp >> 1)) ^ wh_swz_dy(row)) << 4) + 8 * (pp & 1));
  }
  const wh_lds sm = (wh_lds)smem;

  WhStager stg;
  stg.init(p, t0 < t1 ? t0 : 0, txn, tyn, lane);
  if (t0 < t1) stg.issue(p, smem, txn, tyn, cb, ib, wave, lane);
  for (int t = t0; t < t1; ++t) {
    const int st = (t - t0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile t have landed ...
    __syncthreads();                                      // ... everyone's have, and everyone is done reading the other stage
    if (t + 1 < t1) stg.issue(p, smem + (st ^ 1) * WH_STAGE, txn, tyn, cb, ib, wave, lane);
    const wh_lds base = sm + st * WH_STAGE;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {                      // k-group: 16 pixels of tile row kg >> 1, columns 16 (kg & 1) ..
      const int r = kg >> 1, c = (kg & 1) * 16;
      const bf16x8_t af = wh_read(base + dyo[0] + kg * 4096, base + dyo[1] + kg * 4096);
      if constexpr (BIAS) {
        const short8_t raw = __builtin_bit_cast(short8_t, af);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum += bf2f((bf16_t)raw[e]);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int off = ((r + kh) * WH_HS + c) * 128;
          const bf16x8_t xf = wh_read(base + xo[kw][0] + off, base + xo[kw][1] + off);
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf, acc[kh * 3 + kw], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: acc[tap][r] = dW[co0 + 32 cs + (r&3) + 8 (r>>2) + 4 h][tap][ci0 + 32 is + (lane & 31)]: 128 contiguous bytes per half-wave ----
  float* dw = (float*)p.C;
  const int ci = ib * WH_BCI + is * 32 + (lane & 31);
  const int co_base = cb * WH_BCO + cs * 32 + 4 * h;
  const bool atomic = gridDim.x > (unsigned)nblk || p.accumulate == 1;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2);
      if (co < Cout) {
        float* dst = dw + (long)co * p.ldc + t * Cin + ci;
        const float v = acc[t][r] * p.alpha;
        if (atomic) unsafeAtomicAdd(dst, v);
        else *dst = v;
      }
    }
  if constexpr (BIAS) {
    if (ib == 0 && is == 0) {                     // wave-uniform
      bsum += __shfl_xor(bsum, 32);
      const int co = cb * WH_BCO + cs * 32 + (lane & 31);
      if (lane < 32 && co < Cout) {
        if (atomic) unsafeAtomicAdd(p.dbias + co, bsum * p.alpha);
        else p.dbias[co] = bsum * p.alpha;
      }
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------
static bool wgrad_halo_shape_ok(const NkGemmParams& p) {
  const NkGather& g = p.gb;
  if (!p.halo_nb || g.KW != 3 || p.N != 9 * g.C || g.rs != 1 || g.ks != 1 || g.div != 1 || g.off_h != -1 || g.off_w != -1) return false;
  if (g.Ho != g.H || g.Wo != g.W || (g.C % WH_BCI) || (p.M & 7) || p.nbatch || p.ldc != p.N || p.lda != p.M) return false;
  if (p.K != (long)p.halo_nb * g.H * g.W) return false;
  // ragged small images (the 26-wide level of a 1216 x 832 bucket) would spend the tile on padding: they keep the gather kernel
  const long cover = (long)((g.W + WH_TW - 1) / WH_TW) * WH_TW * ((g.H + WH_TH - 1) / WH_TH) * WH_TH;
  return cover * 100 <= (long)g.H * g.W * 125;
}
// pixel-range splits per (co, ci) block: rounds of 256 workgroups at ~3 us per pixel tile against the atomics the splits cost (1.3 TB/s)
static int wgrad_halo_splits(const NkGemmParams& p, int& per) {
  const NkGather& g = p.gb;
  const long T = (long)p.halo_nb * ((g.W + WH_TW - 1) / WH_TW) * ((g.H + WH_TH - 1) / WH_TH);
  const long nblk = (long)((p.M + WH_BCO - 1) / WH_BCO) * (g.C / WH_BCI);
  const double dw_bytes = (double)p.M * p.N * 4.0;
  double best = 1e30;
  int best_s = 1;
  for (int s = 1; s <= 64 && s <= T; ++s) {
    const long tiles = (T + s - 1) / s;
    const long rounds = (nblk * s + 255) / 256;
    const double cost = (double)rounds * (tiles * 3.0e-6 + 4.0e-6) + (s > 1 ? s * dw_bytes / 1.3e12 : dw_bytes / 4.0e12);
    if (cost < best * 0.97) { best = cost; best_s = s; }     // (a larger S must win by 3 %: fewer atomics at a tie)
  }
  per = (int)((T + best_s - 1) / best_s);
  return (int)((T + per - 1) / per);
}
// NK_CONV_WGRAD_HALO: 0 = never (A/B runs, tests), 2 = every eligible shape, unset / 1 = by shape.  By shape (tools/bench_conv_wgrad.py,
// profiles/r04_conv_wgrad.txt, one box, alternating): the kernel wins where the reduction is long -- the 64^2 and 128^2 levels, x1.0-1.8 -- and
// where the (co, ci) blocks fill the chip without splitting the pixel range (1280 -> 1280 at 32^2: 200 blocks, x1.12); few pixels into a
// half-empty grid (640 -> 1280 at 32^2: 100 blocks, two splits, as many atomics as products: x0.69) stay with the gather kernel.
static bool use_wgrad_halo(const NkGemmParams& p, int amode, int bmode, int out_f32) {
  if (amode != OP_MC || bmode != OP_MCG || !out_f32) return false;
  int mode = 1;
  if (const char* e = getenv("NK_CONV_WGRAD_HALO")) mode = atoi(e);
  if (!mode || !wgrad_halo_shape_ok(p)) return false;
  if (mode == 2) return true;
  if (p.M < 64) return false;
  const long nblk = (long)((p.M + WH_BCO - 1) / WH_BCO) * (p.gb.C / WH_BCI);
  return p.K >= 16384 || nblk >= 180;
}
static int launch_wgrad_halo(NkGemmParams& p, hipStream_t stream) {
  int per = 0;
  const int S = wgrad_halo_splits(p, per);
  const long nblk = (long)((p.M + WH_BCO - 1) / WH_BCO) * (p.gb.C / WH_BCI);
  // accumulate: 0 = overwrite, 1 = add, 2 = destination known to be zero.  Split pixel ranges meet through atomics and need a zeroed
  // destination; one range per block stores (or atomically adds to what is there: nobody else touches those elements)
  if (S > 1 && p.accumulate == 0) {
    const size_t n = (size_t)p.M * p.N;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256);
    hipLaunchKernelGGL(nk_zero_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, (float*)p.C, n);
    if (p.dbias) hipLaunchKernelGGL(nk_zero_f32_kernel, dim3(1), dim3(256), 0, stream, p.dbias, (size_t)p.M);
    if (hipGetLastError() != hipSuccess) return NK_ERR_LAUNCH;
  }
  p.accumulate = p.accumulate == 1 ? 1 : 0;      // (kernel: 1 = add to what is there; with S > 1 it adds anyway)
  p.ksplit_len = per;
  auto kern = p.dbias ? nk_conv3x3_wgrad_halo_kernel<1> : nk_conv3x3_wgrad_halo_kernel<0>;
  nk_optin_lds((const void*)kern, WH_SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * S)), dim3(512), WH_SMEM, stream, p);
  return nk_check_launch("nk_conv3x3_wgrad_halo_kernel");
}
