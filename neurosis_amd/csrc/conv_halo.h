// 3 x 3 / stride 1 / padding 1 convolution with an LDS halo tile (included by gemm.hip; not a stand-alone translation unit).
//
// Why it exists.  The implicit-GEMM gather (OP_KCG) stages one [128 pixels][64 channels] A image PER TAP, so every activation
// passes through the 64 B/clk/CU global -> LDS path nine times; at 128-wide CU tiles that path is the bound.  Here
// a workgroup owns a 4 x 32 patch of output pixels of one image and keeps the (4+2) x (32+2) x 64-channel HALO of its inputs in
// LDS: the nine taps of a 64-channel slab read their A fragments from that one image at per-lane addresses
// (pixel + tap offset), and only the weights stream through the ring -- 26 KiB of activations per nine k-steps instead of 144.
//
// Structure: the two-group staggered loop of nk_gemm_g2_kernel (gemm_g2.h) with the A operand replaced.
//   * tile = 128 output pixels (4 rows x 32 columns of one image) x BN output channels (160 or 128), 8 waves = 2 column groups x 4
//     pixel rows; a wave's 32 pixels are one row segment, i.e. 32 CONSECUTIVE rows of the [N*H*W][Cout] output matrix;
//   * k order is (64-channel slab, tap): the halo of slab s+1 is fetched while slab s is multiplied (two 26 KiB buffers);
//   * halo image: pixel hp = py * 34 + px at byte hp * 128, its eight 16-byte channel chunks XOR-swizzled by hp & 7 -- the sixteen
//     consecutive pixels of a fragment read (one pixel row, any tap shift) then hit every bank once (ds_read_b128);
//   * roles: waves 0-5 stage the weights (LDS-DMA, four-stage ring, counted exactly as in the g2 kernel), waves 6-7 stage the halo.
//     vmcnt is per wave and in order, so a wave that staged both would have to land its HBM-latency halo pieces every k-step
//     before its next weight slab; the halo waves instead keep up to seven k-steps of pieces in flight and drain once per slab;
//   * optional GroupNorm(+SiLU) PROLOGUE (the VAE's frozen encoder): the raw input x lands in the halo buffer, and during the last
//     taps of the previous slab all eight waves rewrite it in place as silu(x * a_c + b_c) (padding pixels stay zero) from the
//     per-(image, group) sums of the producer -- the normalised tensor is never written to HBM;
//   * optional GroupNorm STATISTICS epilogue: per (tile, group) sum and sum of squares of the bf16-rounded outputs, combined in a
//     fixed order (no atomics), for the GroupNorm that consumes this convolution's output;
//   * swapped-operand MFMAs and the register-direct epilogue of the g2 kernel (bias / per-image row vector / residual fused).
// Reference call sites: ResBlock in_layers / out_layers convolutions (modules/diffusion/openaimodel.py:247-301), the VAE's
// ResnetBlock conv1 / conv2 (modules/diffusion/model.py:85-134) and the GroupNorm + SiLU in front of them (:116-124).
#pragma once

#define CH_TW 32
#define CH_TH 4
#define CH_HW (CH_TW + 2)
#define CH_HPX ((CH_TH + 2) * CH_HW)            // 204 halo pixels
#define CH_HPIECES 26                           // 1 KiB pieces (8 pixels each) per halo buffer
#define CH_HBUF (CH_HPIECES * 1024)             // 26624
#define CH_BSTAGE 20480                         // weights of one k-step: up to 160 rows x 128 B
#define CH_NS 4
#define CH_SMEM_BYTES (2 * CH_HBUF + CH_NS * CH_BSTAGE)   // 135168
#define CH_NBW 6                                // weight-staging waves (0..5); waves 6, 7 stage the halo
#define CH_HPW (CH_HPIECES / 2)                 // halo pieces per halo wave and slab: 13

// weights of one k-step: rows = output channels, 64 k of (tap, slab); the KC image of the g2 kernel, staged by waves 0-5
template <int BN_>
struct HaloWeights {
  static constexpr int NPC = BN_ / 8;                               // 1 KiB pieces per k-step: 20 or 16
  static constexpr int NPW = (NPC + CH_NBW - 1) / CH_NBW;           // 4 or 3
  const bf16_t* rp[NPW];
  bool ok[NPW];
  __device__ __forceinline__ static int pieces(int wave) { return wave >= CH_NBW ? 0 : (wave + CH_NBW * (NPW - 1) < NPC ? NPW : NPW - 1); }
  __device__ __forceinline__ void init(const bf16_t* Wt, long ld, int Cout, int n0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pc = wave + CH_NBW * i;
      const int row = pc * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (lane >> 3);                   // slot (lane & 7) of row (lane >> 3) holds source chunk slot ^ (row & 7)
      ok[i] = wave < CH_NBW && pc < NPC && n0 + row < Cout;
      rp[i] = Wt + (long)(n0 + (ok[i] ? row : 0)) * ld + chunk * 8;
    }
  }
  // sources of the next k-step; `adv` = element advance to the one after it (next tap: Cin; next slab: 64 - 8 * Cin)
  __device__ __forceinline__ void next_sources(bool live, long adv, const bf16_t* (&src)[NPW]) {
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      src[i] = (live && ok[i]) ? rp[i] : zp;
      rp[i] += adv;
    }
  }
  __device__ __forceinline__ void fire(const bf16_t* const (&src)[NPW], char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (wave + CH_NBW * i < NPC)      // (wave-uniform)
        __builtin_amdgcn_global_load_lds((nk_gptr)src[i], (nk_lptr)(img + (wave + CH_NBW * i) * 1024), 16, 0, 0);
  }
};

// PRO: 0 = the input is read as it is; 1 = GroupNorm(+SiLU) of the input applied in LDS (p.gn_*)
// STATS: 1 = per-tile GroupNorm partial sums of the output written to p.stats_part
template <int BN_, int PRO, int STATS>
__global__ __launch_bounds__(512, 2) void nk_conv3x3_halo_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HN = BN_ / 2, NJ = HN / 16;          // columns per group; 16-column blocks per wave: 5 or 4
  constexpr int NPWB = HaloWeights<BN_>::NPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const bool hwave = wave >= CH_NBW;

  const int H = p.ga.H, W = p.ga.W, Cin = p.ga.C, Cout = p.N;
  const int txn = (W + CH_TW - 1) / CH_TW, tyn = (H + CH_TH - 1) / CH_TH;
  const int per_img = txn * tyn;
  const int ntm = p.halo_nb * per_img, ntn = (Cout + BN_ - 1) / BN_;
  // XCD-aware bijective remap; groups of 16 pixel tiles x all column tiles, column-major inside a group: the 32 workgroups an XCD
  // runs at a time are 16 pixel tiles x 2 weight panels
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GM = 16;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int mt = first_m + (in_group - nt * gm);
  const int n0 = nt * BN_;
  const int img = mt / per_img;
  const int rem = mt - img * per_img;
  const int tyi = rem / txn;
  const int y0 = tyi * CH_TH, x0 = (rem - tyi * txn) * CH_TW;
  const int nslab = Cin >> 6, nk = nslab * 9;

  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  char* const ring = smem + 2 * CH_HBUF;

  // ---- weights ----
  HaloWeights<BN_> wb;
  wb.init(p.B, p.ldb, Cout, n0, wave, lane);
  const long adv_tap = Cin, adv_slab = 64 - 8l * Cin;
  FragG2<OP_KC, BN_, NJ> fb;
  fb.init(lds0 + 2 * CH_HBUF, grp * HN, lane);

  // ---- halo: this lane's A-fragment addresses, one per tap (block 1 = +16 pixels = +2048 B; k sub-step 1 = slot ^ 4 = ^64 B) ----
  unsigned atap[9];
  {
    const int c = lane >> 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hp = (wq + t / 3) * CH_HW + (lane & 15) + (t % 3);
      atap[t] = (unsigned)(hp * 128 + ((c ^ (hp & 7)) << 4));
    }
  }
  // ---- halo staging (waves 6, 7): element offsets of this lane's 16-byte chunk of each piece, relative to channel slab 0; < 0 = padding ----
  int hofs[CH_HPW];
#pragma unroll
  for (int i = 0; i < CH_HPW; ++i) hofs[i] = -1;
  if (hwave) {
#pragma unroll
    for (int i = 0; i < CH_HPW; ++i) {
      const int pc = (wave - CH_NBW) * CH_HPW + i;
      const int hp = pc * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (hp & 7);
      const int py = hp / CH_HW, px = hp - py * CH_HW;
      const int y = y0 - 1 + py, x = x0 - 1 + px;
      const bool v = hp < CH_HPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      hofs[i] = v ? (((img * H + y) * W + x) * Cin + chunk * 8) : -1;
    }
  }
  auto fire_halo = [&](int i, int slab, char* hbuf) {     // piece i of this halo wave, channel slab `slab`
    const bf16_t* src = hofs[i] >= 0 ? p.A + (long)hofs[i] + slab * 64 : (const bf16_t*)nk_zero_page;
    __builtin_amdgcn_global_load_lds((nk_gptr)src, (nk_lptr)(hbuf + ((wave - CH_NBW) * CH_HPW + i) * 1024), 16, 0, 0);
  };

  float4_t acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4], bfr[2 * NJ];

#define CH_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)
  const int nbw = HaloWeights<BN_>::pieces(wave);         // weight pieces this wave stages per k-step (0 for the halo waves)

  // ---- prologue: halo of slab 0, weights of k-steps 0 and 1 in flight; halo 0 and weights 0 landed for everyone ----
  const bf16_t* sb[NPWB];
  if (hwave) {
#pragma unroll
    for (int i = 0; i < CH_HPW; ++i) fire_halo(i, 0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    wb.next_sources(true, adv_tap, sb);
    wb.fire(sb, ring, wave);
    wb.next_sources(nk > 1, adv_tap, sb);
    wb.fire(sb, ring + CH_BSTAGE, wave);
    wb.next_sources(nk > 2, nk > 2 ? adv_tap : 0, sb);    // sources of k-step 2, fired in the first M phase
    // leave k-step 1 in flight
    if (nbw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (nbw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  CH_BAR();
  if (grp == 1) { CH_BAR(); }                              // the second group runs one barrier behind

  unsigned so = 0, sn = 2 * CH_BSTAGE;                     // ring stage of k-step t / of k-step t + 2
  int t = 0;
  for (int s = 0; s < nslab; ++s) {
    const unsigned hcur = lds0 + (unsigned)(s & 1) * CH_HBUF;
    char* const hnext = smem + ((s + 1) & 1) * CH_HBUF;
    const bool more = s + 1 < nslab;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++t) {
      // ---- R: fragment reads of k-step t; halo pieces of slab s + 1; wait for the weights of k-step t + 1 ----
      __builtin_amdgcn_sched_barrier(0);
      g2_read<OP_KC, BN_, NJ>(bfr, fb, so);
      {
        const unsigned a0 = hcur + atap[tap], a1 = hcur + (atap[tap] ^ 64u);
        G2_RD128(af[0], a0, 0);
        G2_RD128(af[1], a0, 2048);
        G2_RD128(af[2], a1, 0);
        G2_RD128(af[3], a1, 2048);
      }
      if (hwave) {
        if (more) {
          if (tap < 6) { fire_halo(2 * tap, s + 1, hnext); fire_halo(2 * tap + 1, s + 1, hnext); }
          else if (tap == 6) fire_halo(12, s + 1, hnext);
        }
        if (tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next slab's halo has landed (this wave's share)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the weights of k-step t + 1 have landed (this wave's share)
      }
      CH_BAR();
      // ---- M: the wave's MFMAs; weights of k-step t + 2 staged and the sources after them computed in their shadow ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (!hwave) {
        wb.fire(sb, ring + sn, wave);
        // k-step t + 3: the one after t + 2 is the next tap unless t + 2 is a slab's last tap
        // sources of k-step t + 3 (fired in the next M phase); the pointer then moves on to k-step t + 4: the next tap, unless
        // t + 3 is a slab's last tap
        const int tap3 = (tap + 3) % 9;
        wb.next_sources(t + 3 < nk, tap3 == 8 ? adv_slab : adv_tap, sb);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)     // operands swapped (D = B.A^T): a lane holds 4 consecutive COLUMNS of one row
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks * NJ + j], af[ks * 2 + i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      CH_BAR();
      so += CH_BSTAGE; if (so == CH_NS * CH_BSTAGE) so = 0;
      sn += CH_BSTAGE; if (sn == CH_NS * CH_BSTAGE) sn = 0;
    }
  }
  if (grp == 0) { CH_BAR(); }                              // ... and the first group waits for it here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the past-the-end zero-page pieces must land before the LDS is given up
#undef CH_BAR

  // ---- epilogue: the wave's 32 pixels are 32 consecutive rows of the output matrix ----
  const int y = y0 + wq;
  const int mb = (img * H + y) * W + x0;
  const int mlimit = y < H ? mb + min(CH_TW, W - x0) : 0;
  const int nb = n0 + grp * HN;
#pragma unroll
  for (int half = 0; half < NJ / 2; ++half) {
    float4_t pair[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { pair[i][0] = acc[i][2 * half]; pair[i][1] = acc[i][2 * half + 1]; }
    reg_epilogue_64x32<0, 2>(p, p.C, pair, mb, nb + half * 32, lane, mlimit);
  }
  if constexpr (NJ & 1) {
    float4_t last[2] = {acc[0][NJ - 1], acc[1][NJ - 1]};
    reg_epilogue_col16<0, 2>(p, p.C, last, mb, nb + (NJ - 1) * 16, lane, mlimit);
  }
}

// NK_CONV_HALO=0 keeps every convolution on the gather kernels (A/B runs)
static bool use_halo(const NkGemmParams& p, int amode, int bmode, int out_f32) {
  if (amode != OP_KCG || bmode != OP_KC || out_f32 || p.nbatch) return false;
  if (const char* e = getenv("NK_CONV_HALO")) if (e[0] == '0') return false;
  const NkGather& g = p.ga;
  if (!p.halo_nb || g.KW != 3 || p.K != 9 * g.C || g.rs != 1 || g.ks != 1 || g.div != 1 || g.off_h != -1 || g.off_w != -1) return false;
  if (g.Ho != g.H || g.Wo != g.W || (g.C & 63) || p.alpha != 1.0f) return false;
  if (p.N % 160 && p.N % 128) return false;
  if ((p.N & 7) || (p.ldc & 7) || (p.residual && (p.ldr & 7))) return false;
  // whole tiles only where it pays: patches of 4 x 32 pixels must cover the image with little waste
  const long cover = (long)((g.W + CH_TW - 1) / CH_TW) * CH_TW * ((g.H + CH_TH - 1) / CH_TH) * CH_TH;
  return cover * 100 <= (long)g.H * g.W * 115;
}

template <int BN_>
static int launch_halo_as(const NkGemmParams& p, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = nk_conv3x3_halo_kernel<BN_, 0, 0>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM_BYTES);
    attr_set = true;
  }
  const NkGather& g = p.ga;
  const long tiles = (long)p.halo_nb * ((g.W + CH_TW - 1) / CH_TW) * ((g.H + CH_TH - 1) / CH_TH) * (p.N / BN_);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), CH_SMEM_BYTES, stream, p);
  return nk_check_launch("nk_conv3x3_halo_kernel");
}
static int launch_halo(const NkGemmParams& p, hipStream_t stream) {
  return p.N % 160 == 0 ? launch_halo_as<160>(p, stream) : launch_halo_as<128>(p, stream);
}
