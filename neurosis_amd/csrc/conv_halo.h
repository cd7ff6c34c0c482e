// 3 x 3 / stride 1 / padding 1 convolution with an LDS halo tile (included by gemm.hip; not a stand-alone translation unit).
//
// Why it exists.  The implicit-GEMM gather (OP_KCG) stages one [128 pixels][64 channels] A image PER TAP, so every activation
// passes through the 64 B/clk/CU global -> LDS path nine times; at 128-wide CU tiles that path is the bound.  Here
// a workgroup owns a TH x 32 patch of output pixels of one image (TH = 4 or 8) and keeps the (TH+2) x 34 x 64-channel HALO of its
// inputs in LDS: the nine taps of a 64-channel slab read their A fragments from that one image at per-lane addresses
// (pixel + tap offset), and only the weights stream through the ring.
//
// Structure: the two-group staggered loop of nk_gemm_g2_kernel (gemm_g2.h) with the A operand replaced.
//   * tile = 32*TH output pixels x BN output channels (160 or 128); 8 waves = 2 column groups x 4 pixel-row groups; a wave owns
//     TH/4 tile rows of 32 pixels = MI 16-pixel row blocks, each tile row being 32 CONSECUTIVE rows of the [N*H*W][Cout] output;
//   * k order is (64-channel slab, tap): the halo of slab s+1 is fetched while slab s is multiplied (two buffers);
//   * halo image: pixel hp = py * 34 + px at byte hp * 128, its eight 16-byte channel chunks XOR-swizzled by hp & 7 -- the sixteen
//     consecutive pixels of a fragment read (one pixel row, any tap shift) then hit every bank once (ds_read_b128);
//   * roles: waves 0-5 stage the weights (LDS-DMA into a THREE-stage ring: the weights of k-step t+2 are issued in the MFMA phase
//     of k-step t, after the barrier behind which the last reads of stage (t-1) % 3 have retired), waves 6-7 stage the halo.
//     vmcnt is per wave and in order, so a wave that staged both would have to land its HBM-latency halo pieces every k-step
//     before its next weight slab; the halo waves instead keep several k-steps of pieces in flight and drain once per slab;
//   * optional GroupNorm STATISTICS epilogue: per (tile, group) sum and sum of squares of the bf16-rounded outputs, combined in a
//     fixed order (no atomics), in the layout gn_reduce_partials_kernel (norm.hip) sums: the GroupNorm that consumes this
//     convolution's output needs no statistics pass of its own;
//   * swapped-operand MFMAs; register-direct epilogue (permlane16_swap pairs columns so a lane stores 16 B) with bias / per-image
//     row vector / residual, all fetched in ONE batch of loads, so the
//     epilogue exposes one memory latency instead of one per row block (measured on the first version of this kernel: ~18 us of
//     fixed cost per 128-pixel tile, most of it five dependent load -> store round trips).
// Reference call sites: ResBlock in_layers / out_layers convolutions (modules/diffusion/openaimodel.py:247-301), the VAE's
// ResnetBlock conv1 / conv2 (modules/diffusion/model.py:85-134) and the GroupNorm + SiLU in front of them (:116-124).
#pragma once
#include <type_traits>

// Diagnostic build only (make EXTRA=-DNK_HALO_STAMPS; tools/halo_stamps.py): wave 0 of every workgroup (first 8192) stamps s_memtime at entry,
// behind the prologue's barrier, behind the k loop and behind the drained epilogue, s_memrealtime around the loop; wave 6 (a halo wave) adds
// the cycles it spent issuing halo pieces.  None of it exists in the shipped library.
#ifdef NK_HALO_STAMPS
__device__ unsigned long long nk_halo_stamp_buf[8 * 8192];
__device__ unsigned long long nk_halo_kstep_buf[3 * 16 * 20 * 5];      // [wave 0 | 4 | 6][workgroup 2048 + i, i < 16][k-step < 20][phase stamp]
// phase stamps of a k-step: 0 = behind the R barrier (M begins), 1 = weight DMA issued, 2 = MFMAs issued, 3 = behind the M barrier (R begins),
// 4 = fragment reads + halo pieces issued and the vmcnt wait done (in front of the R barrier of the NEXT k-step)
#define CH_KSTAMP(t, ph) do { const int w_ = tid >> 6; if ((tid & 63) == 0 && (w_ == 0 || w_ == 4 || w_ == 6) && blockIdx.x >= 2048 && blockIdx.x < 2048 + 16 && (t) < 20) \
    nk_halo_kstep_buf[((((w_ == 0 ? 0 : (w_ == 4 ? 1 : 2)) * 16 + (blockIdx.x - 2048)) * 20 + (t)) * 5) + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int nk_debug_halo_ksteps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nk_halo_kstep_buf), sizeof(nk_halo_kstep_buf)) == hipSuccess ? 0 : 1;
}
#define CH_STAMP(slot) do { if (tid == 0 && blockIdx.x < 8192) nk_halo_stamp_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CH_STAMP_RT(slot) do { if (tid == 0 && blockIdx.x < 8192) nk_halo_stamp_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int nk_debug_halo_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nk_halo_stamp_buf), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#else
#define CH_STAMP(slot)
#define CH_STAMP_RT(slot)
#define CH_KSTAMP(t, ph)
#endif
#define CH_TW 32
#define CH_HW (CH_TW + 2)
#define CH_BSTAGE 20480                         // weights of one k-step: up to 160 rows x 128 B
#define CH_NS 3
// Round 6: the 128-column tile stages its weights in the R phase, into a ring of FOUR stages.  Phase stamps (tools/halo_stamps.py, the VAE's
// 128 -> 128 convolution at 1024^2: a k-step of 1 830 cycles against a 1 024-cycle MFMA floor, every k-step alike) put ~450 cycles of stalled
// weight-DMA issue IN FRONT of the 32 MFMAs of every M phase -- a wave issues in order -- while the R phase ends in 600-900 cycles of waiting
// at its barrier: the two groups' M phases alternate, so the stall is paid twice per k-step.  Issued behind the fragment reads of the R
// phase instead, half a k-step earlier, the DMA of k-step t + 2 would race the other group's last reads of k-step t - 1 in a three-stage
// ring (issued before the barrier, awaited behind it); with four stages it overwrites k-step t - 2's.  (Three k-steps ahead from the M
// phase was measured first and changed nothing: the weights were never late.)  The 160-column tile has no room for a fourth stage.
#define CH_BSTAGE_DEEP 16384
#define CH_NS_DEEP 4
template <int BN_> struct HaloRing {
#ifdef NK_HALO_FIRE_IN_M      // diagnostic build: the round-5 arrangement for every tile (A/B runs)
  static constexpr bool DEEP = false;
#else
  static constexpr bool DEEP = BN_ == 128;
#endif
  static constexpr int BST = DEEP ? CH_BSTAGE_DEEP : CH_BSTAGE;
  static constexpr int NS = DEEP ? CH_NS_DEEP : CH_NS;
  static constexpr int AHEAD = 2;                                     // k-steps between a stage's DMA and its first read
  // Round 6 (late): the 128-column tile hands ALL of its DMA to four PRODUCER WAVES (waves 8-11, one per SIMD beside one wave of each compute
  // group; the two-group GEMM's arrangement, gemm_g2.h nk_gemm_g2p_kernel).  A group's R phase is bound below by its LDS reads (4 waves x
  // 16 KiB = 512 cycles at 128 B/clk, as long as the other group's 512-cycle M phase): whatever the same waves spend issuing DMA -- an LDS-DMA
  // instruction holds the wave's issue until the fill path takes it -- lands on top, in BOTH groups' R phases, twice per k-step.  Waves 8-10
  // stage the weights (pieces pw + 3 i: 6 / 5 / 5 per k-step), wave 11 the halo (44 / 26 pieces per slab over taps 0-6).  Needs three waves
  // per SIMD, i.e. <= 168 VGPRs: the 160-column tile (233) keeps the eight-wave form.  -DNK_HALO_NO_PROD: the eight-wave form everywhere (A/B).
#ifdef NK_HALO_NO_PROD
  static constexpr bool PROD = false;
#else
  static constexpr bool PROD = DEEP;
#endif
  static constexpr bool FIRE_IN_R = DEEP && !PROD;                    // weights of k-step t + 2 issued in the R phase of k-step t (else: its M phase)
  static constexpr int THREADS = PROD ? 768 : 512;
  static constexpr int WAVES_PER_SIMD = PROD ? 3 : 2;
};
#define CH_NPROD_W 3                            // producer waves that stage weights (waves 8, 9, 10); wave 11 stages the halo
#define CH_NBW 6                                // weight-staging waves (0..5); waves 6, 7 stage the halo

template <int MI>
struct HaloGeom {
  static constexpr int TH = 2 * MI;                                   // tile rows: 4 or 8
  static constexpr int RPW = MI / 2;                                  // tile rows per wave
  static constexpr int HPX = (TH + 2) * CH_HW;                        // halo pixels: 204 / 340
  static constexpr int HPIECES = ((HPX + 7) / 8 + 1) & ~1;            // 1 KiB pieces (8 pixels each), even: 26 / 44
  static constexpr int HBUF = HPIECES * 1024;
  static constexpr int HPW = HPIECES / 2;                             // pieces per halo wave and slab: 13 / 22
  static constexpr int SMEM = 2 * HBUF + CH_NS * CH_BSTAGE;           // 114688 / 151552 (the 160-column tile)
  static constexpr int SMEM_DEEP = 2 * HBUF + CH_NS_DEEP * CH_BSTAGE_DEEP;   // 118784 / 155648 (the 128-column tile's four-stage ring)
  template <int BN_> static constexpr int smem() { return HaloRing<BN_>::DEEP ? SMEM_DEEP : SMEM; }
};
// halo pieces a halo wave issues at tap t when its NPH pieces are spread over taps 0..LAST
constexpr int ch_count(int NPH, int LAST, int t) { return t > LAST ? 0 : (NPH + LAST - t) / (LAST + 1); }
constexpr int ch_start(int NPH, int LAST, int t) { int s = 0; for (int u = 0; u < t; ++u) s += ch_count(NPH, LAST, u); return s; }

// weights of one k-step: rows = output channels, 64 k of (tap, slab); the KC image of the g2 kernel, staged by waves 0-5
template <int BN_, int NBW = CH_NBW>
struct HaloWeights {
  static constexpr int NPC = BN_ / 8;                               // 1 KiB pieces per k-step: 20 or 16
  static constexpr int NPW = (NPC + NBW - 1) / NBW;                 // 4 or 3 (six staging waves); 6 (three producer waves)
  const bf16_t* rp[NPW];
  bool ok[NPW];
  // `wave`: index among the staging waves (0 .. NBW-1; anything else stages nothing)
  __device__ __forceinline__ static int pieces(int wave) { return wave >= NBW ? 0 : (wave + NBW * (NPW - 1) < NPC ? NPW : NPW - 1); }
  __device__ __forceinline__ void init(const bf16_t* Wt, long ld, int Cout, int n0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pc = wave + NBW * i;
      const int row = pc * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (lane >> 3);                   // slot (lane & 7) of row (lane >> 3) holds source chunk slot ^ (row & 7)
      ok[i] = wave < NBW && pc < NPC && n0 + row < Cout;
      rp[i] = Wt + (long)(n0 + (ok[i] ? row : 0)) * ld + chunk * 8;
    }
  }
  // stage the current k-step's pieces into `img` (dead k-steps and rows past Cout: the zero page), then move on by `adv` elements
  // (next tap: Cin; next slab: 64 - 8 * Cin)
  __device__ __forceinline__ void fire_next(bool live, long adv, char* img, int wave) {
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (wave + NBW * i < NPC)      // (wave-uniform)
        __builtin_amdgcn_global_load_lds((nk_gptr)((live && ok[i]) ? rp[i] : zp), (nk_lptr)(img + (wave + NBW * i) * 1024), 16, 0, 0);
      rp[i] += adv;
    }
  }
};

// MI: 16-pixel row blocks per wave (2: 4 x 32 tiles, 4: 8 x 32 tiles)
// STATS: 1 = per-tile GroupNorm partial sums of the output written to p.stats_part
template <int BN_, int MI, int STATS>
__global__ __launch_bounds__(HaloRing<BN_>::THREADS, HaloRing<BN_>::WAVES_PER_SIMD) void nk_conv3x3_halo_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = HaloGeom<MI>;
  using R = HaloRing<BN_>;
  constexpr int HN = BN_ / 2, NJ = HN / 16;          // columns per group; 16-column blocks per wave: 5 or 4
  constexpr int RPW = G::RPW, HPW = G::HPW, HBUF = G::HBUF;
  constexpr int HLAST = 6;                           // last tap at which halo pieces of the next slab are issued
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  constexpr bool PROD = HaloRing<BN_>::PROD;           // waves 8-11 own the DMA; the eight compute waves stage nothing
  const bool hwave = !PROD && wave >= CH_NBW;
  CH_STAMP(0);

  const int H = p.ga.H, W = p.ga.W, Cin = p.ga.C, Cout = p.N;
  const int txn = (W + CH_TW - 1) / CH_TW, tyn = (H + G::TH - 1) / G::TH;
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
  const int trem = mt - img * per_img;               // tile index inside the image
  const int tyi = trem / txn;
  const int y0 = tyi * G::TH, x0 = (trem - tyi * txn) * CH_TW;
  const int nslab = Cin >> 6, nk = nslab * 9;

  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  char* const ring = smem + 2 * HBUF;

  // ---- halo staging (waves 6, 7; the producer form: wave 11 alone): piece i of a halo wave covers halo pixels hp0 + 8 i (one per 8 lanes),
  // slot lane & 7 of each; the source of a lane's 16 bytes is recomputed per piece (a dozen scalar-ish VALU instructions) instead of kept in
  // 13 / 22 registers ----
  constexpr int HPP = PROD ? G::HPIECES : HPW;            // pieces per halo-staging wave and slab
  const int hw = PROD ? 0 : wave - CH_NBW;                // index among the halo-staging waves
  const int hp0 = hw * HPP * 8 + (lane >> 3);
  auto fire_halo = [&](int i, int slab, char* hbuf) {     // piece i of this halo wave, channel slab `slab`
    const int hp = hp0 + 8 * i;
    const int py = hp / CH_HW, px = hp - py * CH_HW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool v = hp < G::HPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;      // else padding: the zero page
    const int chunk = (lane & 7) ^ (hp & 7);
    const bf16_t* src = v ? p.A + ((long)((img * H + y) * W + x) * Cin + chunk * 8 + slab * 64) : (const bf16_t*)nk_zero_page;
    __builtin_amdgcn_global_load_lds((nk_gptr)src, (nk_lptr)(hbuf + (hw * HPP + i) * 1024), 16, 0, 0);
  };

  // ---- weights ----
  using WB = HaloWeights<BN_, PROD ? CH_NPROD_W : CH_NBW>;
  const int sw = PROD ? wave - 8 : wave;                  // index among the weight-staging waves (the compute waves of the producer form: none)
  WB wb;
  if (!PROD || wave >= 8) wb.init(p.B, p.ldb, Cout, n0, sw, lane);
  const long adv_tap = Cin, adv_slab = 64 - 8l * Cin;
  // rotated slab order (OpG2::rotate in gemm_g2.h has the why: every XCD reads the whole weight tensor -- 29.5 MB for 1280 -> 1280): XCD x walks
  // the channel slabs s0, ..., nslab - 1, 0, ..., s0 - 1, s0 = x nslab / 8; the taps of a slab keep their order
  const int s0 = p.k_rotate ? (xcd * nslab) >> 3 : 0;
  const long adv_wrap = 64 - 9l * Cin;               // from (last slab, tap 8) back to (slab 0, tap 0)
#define CH_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)
  if constexpr (PROD) {
    if (wave >= 8) {
      // ================= producers =================
      // Barrier b (b = 0: the prologue's) is the boundary the compute groups see: group 0 reads k-step t between barriers 2t and 2t + 1 and
      // multiplies between 2t + 1 and 2t + 2, group 1 one barrier later; a read issued in an R phase has returned behind the `lgkmcnt(0)` that
      // opens the M phase, i.e. before the barrier that closes it.  So stage (t + 2) % 4 -- last read as k-step t - 2, by group 1, done
      // before barrier 2t - 1 -- may be written behind barrier 2t, and must have landed before barrier 2t + 4 (group 0's R of t + 2): the
      // weight producers fire k-step t + 2 between barriers 2t and 2t + 1 and wait for k-step t + 1 (counted vmcnt: one k-step stays in
      // flight) between 2t + 1 and 2t + 2.  The halo of slab s + 1 goes into the buffer slab s - 1 was read from, last by group 1 in its R of
      // k-step 9s - 1, done before barrier 18s + 1: the halo producer fires in the SECOND half of k-steps 9s .. 9s + 6 (behind barrier
      // 18s + 1 at the earliest) and drains in the second half of k-step 9s + 8, in front of barrier 18s + 18 = group 0's first read of it.
      const int pw = wave - 8;
      const bool halo_wave = pw == CH_NPROD_W;
      const int npw = WB::pieces(pw);                      // 6 / 5 / 5; 0 for the halo producer
      if (halo_wave) {
#pragma unroll
        for (int i = 0; i < HPP; ++i) fire_halo(i, s0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int i = 0; i < WB::NPW; ++i) wb.rp[i] += s0 * 64;
        wb.fire_next(true, adv_tap, ring, pw);
        wb.fire_next(nk > 1, adv_tap, ring + R::BST, pw);
        if (npw == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // k-step 0 landed (this wave's pieces)
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      }
      CH_BAR();                                            // barrier 0
      unsigned sn = R::AHEAD * R::BST;
      int t = 0, sl = s0;
      for (int s = 0; s < nslab; ++s) {
        char* const hnext = smem + ((s + 1) & 1) * HBUF;
        const bool more = s + 1 < nslab;
        const int sln = sl + 1 == nslab ? 0 : sl + 1;
        const long adv_end = sl + 1 == nslab ? adv_wrap : adv_slab;
        auto pstep = [&](auto tapc) {
          constexpr int tap = decltype(tapc)::value;
          constexpr int tap2 = (tap + 2) % 9;
          if (!halo_wave) wb.fire_next(t + 2 < nk, tap2 == 8 ? adv_end : adv_tap, ring + sn, pw);
          CH_BAR();                                        // barrier 2t + 1
          if (halo_wave) {
            if (more) {
              constexpr int cnt = ch_count(HPP, HLAST, tap), st = ch_start(HPP, HLAST, tap);
#pragma unroll
              for (int i = 0; i < cnt; ++i) fire_halo(st + i, sln, hnext);
            }
            if (tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            if (npw == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k-step t + 1 landed; t + 2, just issued, stays in flight
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          }
          CH_BAR();                                        // barrier 2t + 2
          sn += R::BST; if (sn == R::NS * R::BST) sn = 0;
          ++t;
        };
        pstep(std::integral_constant<int, 0>{}); pstep(std::integral_constant<int, 1>{}); pstep(std::integral_constant<int, 2>{});
        pstep(std::integral_constant<int, 3>{}); pstep(std::integral_constant<int, 4>{}); pstep(std::integral_constant<int, 5>{});
        pstep(std::integral_constant<int, 6>{}); pstep(std::integral_constant<int, 7>{}); pstep(std::integral_constant<int, 8>{});
        sl = sln;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the past-the-end zero-page pieces must land before the LDS is given up
      CH_BAR();                                            // barrier 2 nk + 1 (group 1's last)
      return;
    }
  } else {
#pragma unroll
    for (int i = 0; i < WB::NPW; ++i) wb.rp[i] += s0 * 64;
    // ---- prologue loads go out first: halo of slab 0, weights of k-steps 0 and 1 ----
    if (hwave) {
#pragma unroll
      for (int i = 0; i < HPW; ++i) fire_halo(i, s0, smem);
    } else {
      wb.fire_next(true, adv_tap, ring, wave);
      wb.fire_next(nk > 1, adv_tap, ring + R::BST, wave);
    }
  }

  FragG2<OP_KC, BN_, NJ> fb;
  fb.init(lds0 + 2 * HBUF, grp * HN, lane);
  // A-fragment address of this lane in a halo buffer: row block (r, c) of tap (dy, dx) reads pixel hp = hpl + (r + dy) * 34 + dx at
  // hp * 128 + ((lane >> 4) ^ (hp & 7)) * 16 (+ c * 2048; k sub-step 1 = slot ^ 4 = ^64 B) -- five VALU instructions per tile row and tap
  const int hpl = wq * RPW * CH_HW + (lane & 15);
  const int cl4 = lane >> 4;
  const int g4 = lane >> 4;
  const int nb = n0 + grp * HN;
  float4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[2 * MI], bfr[2 * NJ];

  const int nbw = PROD ? 0 : WB::pieces(wave);            // weight pieces this wave stages per k-step (0 for the halo waves)

  // ---- prologue waits: halo 0 and weights 0 landed for everyone (weights 1 stay in flight) ----
  if constexpr (!PROD) {
    if (hwave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (nbw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (nbw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  }
  CH_BAR();
  CH_STAMP(1); CH_STAMP_RT(4);
  if (grp == 1) { CH_BAR(); }                              // the second group runs one barrier behind

  unsigned so = 0, sn = R::AHEAD * R::BST;                 // ring stage of k-step t / of k-step t + AHEAD
  int t = 0;
  int sl = s0;                                             // the channel slab of loop iteration s
  for (int s = 0; s < nslab; ++s) {
    const unsigned hcur = lds0 + (unsigned)(s & 1) * HBUF;
    char* const hnext = smem + ((s + 1) & 1) * HBUF;
    const bool more = s + 1 < nslab;
    const int sln = sl + 1 == nslab ? 0 : sl + 1;
    const long adv_end = sl + 1 == nslab ? adv_wrap : adv_slab;
    auto step = [&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      constexpr int dy = tap / 3, dx = tap % 3;
      // ---- R: fragment reads of k-step t; halo pieces of slab s + 1; wait for the weights of k-step t + 1 ----
      __builtin_amdgcn_sched_barrier(0);
      int hpv = hpl;
      asm volatile("" : "+v"(hpv));          // (opaque: keeps the 9 x RPW tap addresses from being hoisted out of the slab loop into 18-36 registers)
      g2_read<OP_KC, BN_, NJ>(bfr, fb, so);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int hp = hpv + (r + dy) * CH_HW + dx;
        const unsigned ar = (unsigned)(hp * 128 + ((cl4 ^ (hp & 7)) << 4));
        const unsigned a0 = hcur + ar, a1 = hcur + (ar ^ 64u);
        G2_RD128(af[2 * r], a0, 0);
        G2_RD128(af[2 * r + 1], a0, 2048);
        G2_RD128(af[MI + 2 * r], a1, 0);
        G2_RD128(af[MI + 2 * r + 1], a1, 2048);
      }
      if constexpr (PROD) {
        // (nothing to stage, nothing to wait for: the producers publish k-step t + 1 and the next halo behind the barriers)
      } else if (hwave) {
        if (more) {
          constexpr int cnt = ch_count(HPW, HLAST, tap), st = ch_start(HPW, HLAST, tap);
#pragma unroll
          for (int i = 0; i < cnt; ++i) fire_halo(st + i, sln, hnext);
        }
        // the next slab's halo has landed (this wave's share): before the barrier that precedes its first use
        if (tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if constexpr (R::FIRE_IN_R) {
        constexpr int tap2 = (tap + 2) % 9;
        wb.fire_next(t + 2 < nk, tap2 == 8 ? adv_end : adv_tap, ring + sn, wave);
        // the weights of k-step t + 1 have landed (this wave's share); those of t + 2, just issued, stay in flight
        if (nbw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the weights of k-step t + 1 have landed (this wave's share)
      }
      CH_KSTAMP(t, 4);
      CH_BAR();
      CH_KSTAMP(t, 0);
      // ---- M: the wave's MFMAs; weights of k-step t + 2 staged and the sources after them computed in their shadow ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!R::FIRE_IN_R && !PROD) {
        if (!hwave) {
          // weights of k-step t + 2; the pointer then moves on to k-step t + 3: the next tap, unless t + 2 is a slab's last tap
          constexpr int tap2 = (tap + 2) % 9;
          wb.fire_next(t + 2 < nk, tap2 == 8 ? adv_end : adv_tap, ring + sn, wave);
        }
      }
      CH_KSTAMP(t, 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)     // operands swapped (D = B.A^T): a lane holds 4 consecutive COLUMNS of one row
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks * NJ + j], af[ks * MI + i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      CH_KSTAMP(t, 2);
      CH_BAR();
      CH_KSTAMP(t, 3);
      so += R::BST; if (so == R::NS * R::BST) so = 0;
      sn += R::BST; if (sn == R::NS * R::BST) sn = 0;
      ++t;
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    sl = sln;
  }
  if (grp == 0) { CH_BAR(); }                              // ... and the first group waits for it here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the past-the-end zero-page pieces must land before the LDS is given up
  CH_STAMP(2); CH_STAMP_RT(5);

  // ---- epilogue ----
  // row block i = (tile row wq * RPW + (i >> 1), pixel columns (i & 1) * 16 + (lane & 15)); acc[i][j][r] = its column grp*HN + j*16 + g4*4 + r
  int mrow[MI];
  bool mok[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int y = y0 + wq * RPW + (i >> 1), x = x0 + (i & 1) * 16 + (lane & 15);
    mok[i] = y < H && x < W;
    mrow[i] = (img * H + y) * W + x;
  }
  bf16_t* const C = (bf16_t*)p.C;
  // the residual of every row block in one batch of loads
  uint4_t res8[MI][NJ / 2 ? NJ / 2 : 1];
  uint2_t res4[MI];
  if (p.residual) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int h = 0; h < NJ / 2; ++h) {
        const int n = nb + h * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;
        res8[i][h] = (mok[i] && n < Cout) ? *(const uint4_t*)(p.residual + (long)mrow[i] * p.ldr + n) : (uint4_t){0u, 0u, 0u, 0u};
      }
      if constexpr (NJ & 1) {
        const int n = nb + (NJ - 1) * 16 + g4 * 4;
        res4[i] = (mok[i] && n < Cout) ? *(const uint2_t*)(p.residual + (long)mrow[i] * p.ldr + n) : (uint2_t){0u, 0u};
      }
    }
  }
  // addends: bias + the image's row vector for this lane's columns (8 per column pair, 4 for a lone block), in the same batch of loads
  // after the permlane swap a lane holds columns n8(h) .. n8(h)+7 of column pair h, and g*4 .. +3 of the lone fifth block
  float cadd[NJ / 2][8], cadd4[4];
#pragma unroll
  for (int h = 0; h < NJ / 2; ++h) {
    const int n = nb + h * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) cadd[h][e] = 0.f;
    if (n < Cout) {
      if (p.bias) {
        const float4_t b0 = *(const float4_t*)(p.bias + n), b1 = *(const float4_t*)(p.bias + n + 4);
        cadd[h][0] = b0[0]; cadd[h][1] = b0[1]; cadd[h][2] = b0[2]; cadd[h][3] = b0[3];
        cadd[h][4] = b1[0]; cadd[h][5] = b1[1]; cadd[h][6] = b1[2]; cadd[h][7] = b1[3];
      }
      if (p.rowvec) {
        float t[8];
        unpack8(*(const uint4_t*)(p.rowvec + (long)img * p.ld_rowvec + n), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) cadd[h][e] += t[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) cadd4[e] = 0.f;
  if constexpr (NJ & 1) {
    const int n = nb + (NJ - 1) * 16 + g4 * 4;
    if (n < Cout) {
      if (p.bias) { const float4_t b = *(const float4_t*)(p.bias + n); cadd4[0] = b[0]; cadd4[1] = b[1]; cadd4[2] = b[2]; cadd4[3] = b[3]; }
      if (p.rowvec) {
        float t[4];
        unpack4(*(const uint2_t*)(p.rowvec + (long)img * p.ld_rowvec + n), t);
        cadd4[0] += t[0]; cadd4[1] += t[1]; cadd4[2] += t[2]; cadd4[3] += t[3];
      }
    }
  }

  // per-lane column sums for the statistics epilogue: [column slot][sum, sum of squares]
  float st8[NJ / 2 ? NJ / 2 : 1][8][2], st4[4][2];
  if constexpr (STATS) {
#pragma unroll
    for (int h = 0; h < NJ / 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) { st8[h][e][0] = 0.f; st8[h][e][1] = 0.f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { st4[e][0] = 0.f; st4[e][1] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int h = 0; h < NJ / 2; ++h) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // every lane takes part in the swap; the bounds predicates come after it
        const float a_own = acc[i][2 * h][r], b_own = acc[i][2 * h + 1][r];
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a_own), __float_as_uint(b_own), false, false);
        v[r] = __uint_as_float(sw[0]);
        v[4 + r] = __uint_as_float(sw[1]);
      }
      const int n = nb + h * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += cadd[h][e];
      if (p.residual) {
        float tr[8];
        unpack8(res8[i][h], tr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += tr[e];
      }
      const uint4_t packed = pack8(v);
      const bool ok = mok[i] && n < Cout;
      if (ok) *(uint4_t*)(C + (long)mrow[i] * p.ldc + n) = packed;
      if constexpr (STATS) {
        float q[8];
        unpack8(packed, q);                   // statistics of what the consumer will read: the bf16-rounded values
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = ok ? q[e] : 0.f; st8[h][e][0] += x; st8[h][e][1] += x * x; }
      }
    }
    if constexpr (NJ & 1) {
      const int n = nb + (NJ - 1) * 16 + g4 * 4;
      float v[4] = {acc[i][NJ - 1][0] + cadd4[0], acc[i][NJ - 1][1] + cadd4[1], acc[i][NJ - 1][2] + cadd4[2], acc[i][NJ - 1][3] + cadd4[3]};
      if (p.residual) {
        float tr[4];
        unpack4(res4[i], tr);
        v[0] += tr[0]; v[1] += tr[1]; v[2] += tr[2]; v[3] += tr[3];
      }
      uint2_t o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      const bool ok = mok[i] && n < Cout;
      if (ok) *(uint2_t*)(C + (long)mrow[i] * p.ldc + n) = o;
      if constexpr (STATS) {
        float q[4];
        unpack4(o, q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float x = ok ? q[e] : 0.f; st4[e][0] += x; st4[e][1] += x * x; }
      }
    }
  }
  if constexpr (STATS) {
    // column sums over the tile, in a fixed order: 16 lanes of a column slot (xor butterfly) -> the 4 pixel-row waves of a column group
    // (LDS) -> the columns of a GroupNorm group -> part[img][tile][2 * group + {0, 1}].  The ring is free: every wave is past the loop.
    __syncthreads();
    float* const cs = (float*)ring;            // [8 waves][HN columns][2]
    // sum over the 16 lanes of a DPP row (= the 16 pixels of a row block), in every lane: four VALU instructions with DPP operands.
    // (`__shfl_xor` compiled to ds_bpermute_b32 for all four steps: 128 LDS round trips per wave, 313 us of a 1.8 ms launch at the VAE's
    // 1024^2 level.)  Same pairing as the xor butterfly -- quad, quad pair, row half, row -- so the sums are bit-identical to it.
    auto lane16_sum = [](float x) {
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
      return x;
    };
#pragma unroll
    for (int h = 0; h < NJ / 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float tot = lane16_sum(st8[h][e][k]);
          const int col = h * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8 + e;          // column inside the group's HN
          if ((lane & 15) == 0) cs[(wave * HN + col) * 2 + k] = tot;
        }
    if constexpr (NJ & 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float tot = lane16_sum(st4[e][k]);
          const int col = (NJ - 1) * 16 + g4 * 4 + e;
          if ((lane & 15) == 0) cs[(wave * HN + col) * 2 + k] = tot;
        }
    }
    __syncthreads();
    const int cpg = Cout / p.stats_groups;      // channels per GroupNorm group; BN_ % cpg == 0 (use_halo)
    const int ngt = BN_ / cpg;                  // groups this column tile covers
    if (tid < 2 * ngt) {
      const int gi = tid >> 1, k = tid & 1;
      float a = 0.f;
      for (int cc = 0; cc < cpg; ++cc) {
        const int col = gi * cpg + cc;          // column of the tile: group half col / HN, its 4 pixel-row waves
        const int gh = col / HN, cl = col - gh * HN;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) a += cs[((gh * 4 + w4) * HN + cl) * 2 + k];
      }
      p.stats_part[((long)img * per_img + trem) * 2 * p.stats_groups + 2 * (n0 / cpg + gi) + k] = a;
    }
  }
#ifdef NK_HALO_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CH_STAMP(3);
#endif
#undef CH_BAR
}

// NK_CONV_HALO=0 keeps every convolution on the gather kernels (A/B runs)
static bool halo_shape_ok(const NkGemmParams& p) {
  const NkGather& g = p.ga;
  if (!p.halo_nb || g.KW != 3 || p.K != 9 * g.C || g.rs != 1 || g.ks != 1 || g.div != 1 || g.off_h != -1 || g.off_w != -1) return false;
  if (g.Ho != g.H || g.Wo != g.W || (g.C & 63) || p.alpha != 1.0f || p.nbatch) return false;
  if (p.N % 160 && p.N % 128) return false;
  if ((p.N & 7) || (p.ldc & 7) || (p.residual && (p.ldr & 7))) return false;
  return true;
}
// column-tile width of the halo-tile launch for N output channels: THE rule (tile planner, launcher and the statistics-epilogue query
// nk_conv2d_stats_tiles all ask here: the epilogue's partial layout depends on it)
static int halo_bn(int N) { return N % 160 == 0 ? 160 : 128; }
// tile height for this problem: 8-row tiles where they still give about one workgroup per CU, else 4-row tiles; 0 = the patches
// would cover the image with too much waste (ragged small images keep the gather kernels)
static int halo_tile_rows(const NkGemmParams& p) {
  const NkGather& g = p.ga;
  const int bn = halo_bn(p.N);
  const long txn = (g.W + CH_TW - 1) / CH_TW;
  for (int th = 8; th >= 4; th -= 4) {
    const long tyn = (g.H + th - 1) / th;
    const long cover = txn * CH_TW * tyn * th;
    if (cover * 100 > (long)g.H * g.W * 115) continue;
    const long tiles = (long)p.halo_nb * txn * tyn * (p.N / bn);
    if (th == 8 && tiles < 224) continue;
    return th;
  }
  return 0;
}
static bool use_halo(const NkGemmParams& p, int amode, int bmode, int out_f32) {
  if (amode != OP_KCG || bmode != OP_KC || out_f32) return false;
  if (const char* e = getenv("NK_CONV_HALO")) if (e[0] == '0') return false;
  return halo_shape_ok(p) && halo_tile_rows(p) != 0;
}

template <int BN_, int MI, int STATS>
static int launch_halo_as(const NkGemmParams& p_in, hipStream_t stream) {
  NkGemmParams p = p_in;
  p.k_rotate = k_rotate_on(p.ga.C) ? 1 : 0;          // (channel slabs, not k-steps: 512 input channels and up)
  auto kern = nk_conv3x3_halo_kernel<BN_, MI, STATS>;
  nk_optin_lds((const void*)kern, HaloGeom<MI>::template smem<BN_>());
  const NkGather& g = p.ga;
  const long tiles = (long)p.halo_nb * ((g.W + CH_TW - 1) / CH_TW) * ((g.H + HaloGeom<MI>::TH - 1) / HaloGeom<MI>::TH) * (p.N / BN_);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(HaloRing<BN_>::THREADS), HaloGeom<MI>::template smem<BN_>(), stream, p);
  return nk_check_launch("nk_conv3x3_halo_kernel");
}
template <int STATS>
static int launch_halo_s(const NkGemmParams& p, hipStream_t stream) {
  const int th = halo_tile_rows(p);
  const bool wide = halo_bn(p.N) == 160;
  if (th == 8) return wide ? launch_halo_as<160, 4, STATS>(p, stream) : launch_halo_as<128, 4, STATS>(p, stream);
  return wide ? launch_halo_as<160, 2, STATS>(p, stream) : launch_halo_as<128, 2, STATS>(p, stream);
}
static int launch_halo(const NkGemmParams& p, hipStream_t stream) {
  return p.stats_part ? launch_halo_s<1>(p, stream) : launch_halo_s<0>(p, stream);
}
// pixel tiles per image of the launch `launch_halo` would make (the statistics epilogue writes one partial row per tile)
static int halo_tiles_per_image(const NkGemmParams& p) {
  const int th = halo_tile_rows(p);
  if (!th) return 0;
  return ((p.ga.W + CH_TW - 1) / CH_TW) * ((p.ga.H + th - 1) / th);
}
