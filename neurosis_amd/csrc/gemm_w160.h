// Exact-round fp32 weight-gradient kernel of the MFMA tile engine (included by gemm.hip behind gemm_g2.h; not a stand-alone translation unit).
//
// Why it exists (round 6).  A Linear weight gradient  dW[out][in] = sum_tokens dy[token][out] * x[token][in]  is a K = 4096-token reduction into
// an output the SHAPE of the weight, and with 128 x 128 tiles at two workgroups per CU the SDXL weights quantise badly against 512 slots:
// 10240 x 1280 -> 800 tiles = 1.56 rounds, 1280 x 5120 -> 400 = 0.78, 3840 x 1280 -> 300 = 0.59: the per-shape rates of round 5 (769 / 671 /
// 576 TFLOP/s serialized, profiles/r05_gemm_shapes.txt) are ~980 x those fill factors.  K cannot be split to fill the rounds (every split
// costs out x in x 4 B of fp32 atomics at ~1.3 TB/s), but every 1280-level weight is a multiple of 160 in both directions, and 160-row tiles
// at ONE workgroup per CU come out in whole rounds of 256: 10240 x 1280 -> 64 x 8 = 512 (160 x 160), 1280 x 5120 -> 8 x 32 = 256,
// 3840 x 1280 -> 24 x 10 = 240 (160 x 128), three batched 1280 x 1280 -> 3 x 8 x 10 = 240.
//
// Structure: 512 threads = 4 compute waves + 4 producer waves, one of each per SIMD (the cyclic wave -> SIMD placement).
//   * compute wave (wr, wc) owns rows [80 wr, 80 wr + 80) x columns [BN/2 wc, BN/2 wc + BN/2) of the tile: 5 x 5 (BN 160) or 5 x 4 (BN 128)
//     blocks of v_mfma_f32_16x16x32_bf16 = 100 / 80 accumulator registers; with a SIMD to itself it has the registers to software-pipeline
//     its own loop: the fragments of k sub-step s + 1 (32 tokens: 5 + 5 fragments = 20 ds_read_b64_tr_b16) are requested one per MFMA gap
//     under the 25 MFMAs of sub-step s and awaited row block by row block with counted lgkmcnt -- there is no read phase and no partner
//     group, the matrix pipe is the only thing a sub-step waits for (MFMA floor 400 cycles per sub-step; LDS reads 80 KB per 64-token slab
//     and CU = 320 of its 800 cycles; the 40 KB slab = 640 cycles of the 64 B / clk / CU fill path).
//   * producer waves 4-7 own the tile DMA (OpG2<OP_MC, ., 4>: both operands are staged as they lie in memory, [64 tokens][rows], and read
//     through the transposing LDS read): ring of THREE stages, two slabs in flight, counted vmcnt.
//   * ONE barrier per slab, numbered by the slab it makes readable: producers wait (vmcnt) for their pieces of slab b in front of barrier b;
//     compute waves pass barrier b in the middle of slab b - 1's second sub-step (after its first row of MFMAs, so the pipe has work while
//     the barrier settles), by which time their reads of slab b - 1 have all RETURNED (they are being consumed) -- so behind barrier b the
//     producers may overwrite the stage of slab b - 1 with slab b + 2.
//   * bias gradient (row sums of dy) by one extra MFMA against a fragment of ones per sub-step, but SPREAD over the column tiles: tile (., nt)
//     sums row blocks nt, nt + ntn, ... of its ten -- at one workgroup per CU the launch lasts as long as its slowest tile, and ten extra
//     MFMAs per slab in the first column tile alone (the 128 x 128 kernels' arrangement) would cost every launch 10 %.  No atomics: every
//     row block of every row tile has exactly one owner.
//   * token range split over blockIdx.y (the 64^2 level: 16384 tokens into weights of 48-160 tiles): fp32 atomics into a zeroed / accumulating
//     destination, as the 128 x 128 kernels do.
//   * rotated k order per XCD (OpG2::rotate), XCD-aware tile order, fp32 epilogue through permlane16_swap (32-byte runs per lane).
#pragma once

#define W160_BM 160
#define W160_NS 3
// Diagnostic build only (make EXTRA=-DNK_W160_STAMPS; tools/w160_stamps.py).  Per workgroup, compute wave 0: s_memtime at entry, behind barrier 0,
// behind the k loop, behind the epilogue (+ s_memrealtime around the loop: the clock held); producer wave 4: cycles spent issuing DMA, waiting on
// vmcnt and waiting at the slab barrier, summed over the loop.  None of it exists in the shipped library.
#ifdef NK_W160_STAMPS
__device__ unsigned long long nk_w160_stamp_buf[12 * 4096];
#define W160_STAMP(slot) do { if (tid == 0 && blockIdx.x < 4096 && blockIdx.z == 0 && blockIdx.y == 0) nk_w160_stamp_buf[blockIdx.x * 12 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W160_STAMP_RT(slot) do { if (tid == 0 && blockIdx.x < 4096 && blockIdx.z == 0 && blockIdx.y == 0) nk_w160_stamp_buf[blockIdx.x * 12 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define W160_PSTAMP_DECL unsigned long long ps_fire = 0, ps_wait = 0, ps_bar = 0, ps_t = 0
#define W160_PSTAMP_T() (ps_t = __builtin_amdgcn_s_memtime())
#define W160_PSTAMP_ADD(acc) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - ps_t; ps_t = n_; } while (0)
#define W160_PSTAMP_OUT() do { if (tid == 256 && blockIdx.x < 4096 && blockIdx.z == 0 && blockIdx.y == 0) { nk_w160_stamp_buf[blockIdx.x * 12 + 6] = ps_fire; nk_w160_stamp_buf[blockIdx.x * 12 + 7] = ps_wait; nk_w160_stamp_buf[blockIdx.x * 12 + 8] = ps_bar; } } while (0)
#else
#define W160_STAMP(slot)
#define W160_STAMP_RT(slot)
#define W160_PSTAMP_DECL
#define W160_PSTAMP_T()
#define W160_PSTAMP_ADD(acc)
#define W160_PSTAMP_OUT()
#endif

template <int BN_>
struct W160Cfg {
  static constexpr int NJ = BN_ / 32;                         // 16-column blocks per compute wave: 5 or 4
  static constexpr int A_BYTES = W160_BM * 128;               // [64 tokens][160 rows] bf16
  static constexpr int B_BYTES = BN_ * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;             // 40960 / 36864
  static constexpr int SMEM = W160_NS * STAGE;                // 122880 / 110592
  static constexpr int NRD = 2 * (NJ + 5);                    // transposing reads per sub-step and wave: 20 / 18
};

#define W160_RDTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define W160_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)

// one fragment half: read r (0 .. NRD-1) of a sub-step, in the order the MFMAs need them: B blocks 0 .. NJ-1 (lo, hi), then A blocks 0 .. 4
template <int BN_, int KS, int R>
__device__ __forceinline__ void w160_issue_read(short4_t (&alo)[5], short4_t (&ahi)[5], short4_t (&blo)[W160Cfg<BN_>::NJ], short4_t (&bhi)[W160Cfg<BN_>::NJ],
                                                const unsigned (&aa)[5], const unsigned (&ba)[W160Cfg<BN_>::NJ]) {
  constexpr int NJ = W160Cfg<BN_>::NJ;
  if constexpr (R < 2 * NJ) {
    constexpr int j = R >> 1, RS = BN_ * 2;
    if constexpr ((R & 1) == 0) W160_RDTR(blo[j], ba[j], KS * 32 * RS);
    else W160_RDTR(bhi[j], ba[j], KS * 32 * RS + 4 * RS);
  } else if constexpr (R < 2 * NJ + 10) {
    constexpr int i = (R - 2 * NJ) >> 1, RS = W160_BM * 2;
    if constexpr (((R - 2 * NJ) & 1) == 0) W160_RDTR(alo[i], aa[i], KS * 32 * RS);
    else W160_RDTR(ahi[i], aa[i], KS * 32 * RS + 4 * RS);
  }
}

// The DMA sources of one r-contiguous operand ([tokens][rows] in memory, staged as it lies) for ONE of the four producer waves: piece
// pc = pw + 4 i is the 1 KiB run [64 pc, 64 pc + 64) of the image's 16-byte chunks (k-row major, ROWS / 8 chunks per k-row, the column-block
// swizzle of OpG2 / FragG2 applied on the source side).
template <int ROWS>
struct W160Src {
  static constexpr int NPW = ROWS / 8 / 4, CH = ROWS / 8;
  const bf16_t* rp[NPW];      // this lane's 16 bytes of the piece in the NEXT slab (the zero page for rows past the operand's end)
  long st[NPW];               // elements per slab (0 for zero-page pieces)
  int kk[NPW];                // k-row of the piece inside a slab
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int R, int r0, int pw, int lane) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int S = 64 * (pw + 4 * i) + lane;
      const int k = S / CH, c = S - k * CH;
      const int src = CH == 16 ? (c ^ mc_swz(k)) : (c ^ (((k >> 3) & 1) << 1));
      const bool ok = r0 + src * 8 < R;
      kk[i] = k;
      rp[i] = ok ? P + (long)k * ld + r0 + src * 8 : (const bf16_t*)nk_zero_page;
      st[i] = ok ? (long)BK * ld : 0;
    }
  }
  __device__ __forceinline__ void skip(int slabs) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) rp[i] += (long)slabs * st[i];
  }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int i = 0; i < NPW; ++i) rp[i] += st[i];
  }
  __device__ __forceinline__ void rewind(int slabs) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) rp[i] -= (long)slabs * st[i];
  }
  __device__ __forceinline__ void fire_full(char* img, int pw) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i) __builtin_amdgcn_global_load_lds((nk_gptr)rp[i], (nk_lptr)(img + (pw + 4 * i) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void fire_tail(char* img, int pw, int kbase, int klen) const {      // the ragged last slab: k-rows past klen are zeros
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      __builtin_amdgcn_global_load_lds((nk_gptr)(kbase + kk[i] < klen ? rp[i] : (const bf16_t*)nk_zero_page), (nk_lptr)(img + (pw + 4 * i) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void fire_zero(char* img, int pw) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i) __builtin_amdgcn_global_load_lds((nk_gptr)nk_zero_page, (nk_lptr)(img + (pw + 4 * i) * 1024), 16, 0, 0);
  }
};

template <int BN_>
__global__ __launch_bounds__(512, 1) void nk_gemm_w160_kernel(const NkGemmParams p) {
  using Cfg = W160Cfg<BN_>;
  constexpr int NJ = Cfg::NJ, NRD = Cfg::NRD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  W160_STAMP(0);

  // XCD-aware bijective remap + grouped tile order (nk_gemm_g2_kernel's): an XCD's consecutive tiles share GM dy panels and walk the x panels
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN_ - 1) / BN_, ntm = (p.M + W160_BM - 1) / W160_BM;
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int m0 = (first_m + (in_group - nt * gm)) * W160_BM, n0 = nt * BN_;
  // this workgroup's token range (blockIdx.y: K split)
  const int kbeg = blockIdx.y * p.ksplit_len;
  const int klen = min(p.K, kbeg + p.ksplit_len) - kbeg;
  const int nk = (klen + BK - 1) / BK;
  const bf16_t* Ap = (p.nbatch ? p.Ab[blockIdx.z] : p.A) + (long)kbeg * p.lda;
  const bf16_t* Bp = (p.nbatch ? p.Bb[blockIdx.z] : p.B) + (long)kbeg * p.ldb;

  if (wave >= 4) {
    // ================= producer: the tile DMA of all 160 + BN_ rows, pieces pw + 4 i =================
    // (Lean on purpose: the producer shares its SIMD's issue slots with a compute wave that runs the matrix pipe flat out, and OpG2's per-slab
    // source selection -- ~10 vector instructions per piece -- would need ~400 issue cycles per slab out of what the MFMAs leave over.  Here
    // a piece whose rows are out of range points at the zero page with a step of zero, so a full slab costs one 64-bit add per piece; the
    // ragged last slab and the past-the-end slabs take wave-uniform side branches.)
    const int pw = wave - 4;
    W160Src<W160_BM> oa;
    W160Src<BN_> ob;
    oa.init(Ap, p.lda, p.M, m0, pw, lane);
    ob.init(Bp, p.ldb, p.N, n0, pw, lane);
    constexpr int PPS = W160Src<W160_BM>::NPW + W160Src<BN_>::NPW;      // pieces per slab and producer: 10 or 9
    int kslab = p.k_rotate ? (xcd * nk) >> 3 : 0;                       // rotated k order per XCD (OpG2::rotate has the why)
    oa.skip(kslab); ob.skip(kslab);
    int handed = 0;
    auto fire_next = [&](char* stage) {
      if (handed >= nk) {
        oa.fire_zero(stage, pw); ob.fire_zero(stage + Cfg::A_BYTES, pw);
      } else if (kslab * BK + BK <= klen) {
        oa.fire_full(stage, pw); ob.fire_full(stage + Cfg::A_BYTES, pw);
      } else {
        oa.fire_tail(stage, pw, kslab * BK, klen); ob.fire_tail(stage + Cfg::A_BYTES, pw, kslab * BK, klen);
      }
      ++handed; ++kslab;
      oa.advance(); ob.advance();
      if (kslab == nk) { kslab = 0; oa.rewind(nk); ob.rewind(nk); }
    };
    fire_next(smem);
    fire_next(smem + Cfg::STAGE);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPS) : "memory");        // slab 0 landed (this wave's pieces)
    W160_BAR();                                                       // barrier 0
    unsigned sn = 2 * Cfg::STAGE;                                     // stage of slab t + 2
    W160_PSTAMP_DECL;
    W160_PSTAMP_T();
    for (int t = 0; t < nk; ++t) {
      fire_next(smem + sn);
      W160_PSTAMP_ADD(ps_fire);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPS) : "memory");      // slab t + 1 landed
      W160_PSTAMP_ADD(ps_wait);
      W160_BAR();                                                     // barrier t + 1
      W160_PSTAMP_ADD(ps_bar);
      sn += Cfg::STAGE; if (sn == W160_NS * Cfg::STAGE) sn = 0;
    }
    W160_PSTAMP_OUT();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the past-the-end zero-page pieces land before the LDS is given up
    if (gridDim.y > 1) { W160_BAR(); }                                // ... and before the compute waves stage their atomics through it
    return;
  }

  // ================= compute =================
  __builtin_amdgcn_s_setprio(1);
  const int wr = wave >> 1, wc = wave & 1;
  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  FragG2<OP_MC, W160_BM, 5> fa;
  FragG2<OP_MC, BN_, NJ> fb;
  fa.init(lds0, wr * 80, lane);
  fb.init(lds0 + Cfg::A_BYTES, wc * (BN_ / 2), lane);

  float4_t acc[5][NJ];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient: the row blocks of this tile that THIS column tile sums (bit rb of the tile's ten), and this wave's five of them
  float* const dbias = p.nbatch ? p.dbias_b[blockIdx.z] : p.dbias;
  unsigned bmask = 0;
  if (dbias != nullptr && wc == 0) {
    for (int rb = nt; rb < 10; rb += ntn) bmask |= 1u << rb;
    bmask = (bmask >> (5 * wr)) & 31u;
  }
  bmask = __builtin_amdgcn_readfirstlane(bmask);
  float4_t accb[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) accb[i] = (float4_t){0.f, 0.f, 0.f, 0.f};

  short4_t alo[2][5], ahi[2][5], blo[2][NJ], bhi[2][NJ];      // [sub-step parity]: the fragments in use and the ones on their way
  unsigned aa[5], ba[NJ];                                       // per-lane read addresses of the stage being read
  auto set_stage = [&](unsigned so) {
#pragma unroll
    for (int i = 0; i < 5; ++i) aa[i] = fa.a[i] + so;
#pragma unroll
    for (int j = 0; j < NJ; ++j) ba[j] = fb.a[j] + so;
  };
  auto frag = [](const short4_t& lo, const short4_t& hi) {
    short8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, r);
  };

  // One k sub-step: the MFMAs of parity CUR's fragments, row block by row block, each row behind a counted wait for its A fragment; the
  // fragments of the NEXT sub-step (parity 1 - CUR: k sub-step NKS of the stage `aa / ba` point at) are requested one per MFMA gap from
  // row 1 on.  BARRIER: pass the slab barrier behind row 0 and move the read addresses to stage `so_next` (second sub-step of a slab).
  // Outstanding LDS reads at entry: exactly the NRD of parity CUR, issued in the order B 0 .. NJ-1, A 0 .. 4.
  // BIAS (compile time): this wave sums some of its row blocks for the bias gradient -- the k loop exists twice, so that the waves
  // that do not (all but one or two per tile) run it without the five wave-uniform tests and branches per sub-step.
  auto substep = [&](auto cur_tag, auto nks_tag, auto barrier_tag, auto bias_tag, unsigned so_next) {
    constexpr int CUR = decltype(cur_tag)::value, NKS = decltype(nks_tag)::value;
    constexpr bool BARRIER = decltype(barrier_tag)::value, BIAS = decltype(bias_tag)::value;
    constexpr int NXT = 1 - CUR;
    // B fragments and A 0: all but the last 8 reads (A 1 .. 4) have returned
    if constexpr (NJ == 5)
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(blo[CUR][0]), "+v"(bhi[CUR][0]), "+v"(blo[CUR][1]), "+v"(bhi[CUR][1]), "+v"(blo[CUR][2]), "+v"(bhi[CUR][2]),
                   "+v"(blo[CUR][3]), "+v"(bhi[CUR][3]), "+v"(blo[CUR][4]), "+v"(bhi[CUR][4]), "+v"(alo[CUR][0]), "+v"(ahi[CUR][0]));
    else
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(blo[CUR][0]), "+v"(bhi[CUR][0]), "+v"(blo[CUR][1]), "+v"(bhi[CUR][1]), "+v"(blo[CUR][2]), "+v"(bhi[CUR][2]),
                   "+v"(blo[CUR][3]), "+v"(bhi[CUR][3]), "+v"(alo[CUR][0]), "+v"(ahi[CUR][0]));
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t bf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[j] = frag(blo[CUR][j], bhi[CUR][j]);
    {
      const bf16x8_t a0 = frag(alo[CUR][0], ahi[CUR][0]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], a0, acc[0][j], 0, 0, 0);
      if constexpr (BIAS) {
        if (bmask & 1u) accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), a0, accb[0], 0, 0, 0);
      }
    }
    if constexpr (BARRIER) {
      W160_BAR();
      set_stage(so_next);
    }
    __builtin_amdgcn_sched_barrier(0);
    // rows 1 .. 4: reads of the next sub-step ride in the MFMA gaps (RPG per gap: 20 reads over 20 gaps, or 18 over 16)
    constexpr int GAPS = 4 * NJ;
    auto row = [&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value;
      // A i has returned: of parity CUR only A i+1 .. 4 may be outstanding, plus what this sub-step has requested so far
      constexpr int issued = ((i - 1) * NJ * NRD + GAPS - 1) / GAPS;        // reads issued in front of row i (ceil split, below)
      constexpr int allow = 2 * (4 - i) + issued;
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(alo[CUR][i]), "+v"(ahi[CUR][i]) : "n"(allow > 15 ? 15 : allow));
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8_t ai = frag(alo[CUR][i], ahi[CUR][i]);
      auto gap = [&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int g = (i - 1) * NJ + j;                                   // gap index 0 .. GAPS-1
        constexpr int r0 = (g * NRD + GAPS - 1) / GAPS, r1 = ((g + 1) * NRD + GAPS - 1) / GAPS;      // reads [r0, r1) in this gap
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], ai, acc[i][j], 0, 0, 0);
        if constexpr (r1 > r0) w160_issue_read<BN_, NKS, r0>(alo[NXT], ahi[NXT], blo[NXT], bhi[NXT], aa, ba);
        if constexpr (r1 > r0 + 1) w160_issue_read<BN_, NKS, r0 + 1>(alo[NXT], ahi[NXT], blo[NXT], bhi[NXT], aa, ba);
        __builtin_amdgcn_sched_barrier(0);
      };
      gap(std::integral_constant<int, 0>{}); gap(std::integral_constant<int, 1>{}); gap(std::integral_constant<int, 2>{}); gap(std::integral_constant<int, 3>{});
      if constexpr (NJ == 5) gap(std::integral_constant<int, 4>{});
      if constexpr (BIAS) {
        if (bmask & (1u << i)) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), ai, accb[i], 0, 0, 0);
      }
    };
    row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{}); row(std::integral_constant<int, 4>{});
  };

  // prologue: slab 0 readable behind barrier 0; its first sub-step's fragments requested in order
  W160_BAR();
  W160_STAMP(1); W160_STAMP_RT(4);
  set_stage(0);
  {
    auto rd = [&](auto r_tag) { w160_issue_read<BN_, 0, decltype(r_tag)::value>(alo[0], ahi[0], blo[0], bhi[0], aa, ba); };
    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
    rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{});
    rd(std::integral_constant<int, 8>{}); rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
    rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{}); rd(std::integral_constant<int, 14>{}); rd(std::integral_constant<int, 15>{});
    rd(std::integral_constant<int, 16>{}); rd(std::integral_constant<int, 17>{});
    if constexpr (NRD == 20) { rd(std::integral_constant<int, 18>{}); rd(std::integral_constant<int, 19>{}); }
  }
  // ALL of them back before either k loop is entered.  To the compiler an asm read defines its outputs where it stands; the edge into a loop
  // that keeps fragments in registers across iterations carries the allocator's phi copies of them, and it placed those (for the bias
  // variant of the loop) straight behind the reads, in front of the loop's first counted wait: the last-requested fragment (A 4) was copied
  // before its data had returned, and the first sub-step's row 4 ran on whatever the registers held before -- one 16 x 64 block of one
  // weight gradient wrong about once per hundred steps, seen only when the leftovers were huge (tools/check_async_reads.py
  // screens the assembly for this).  Naming every fragment "+v" here pins the copies behind the wait; inside the loop the counted waits name theirs.
  if constexpr (NJ == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[0][0]), "+v"(bhi[0][0]), "+v"(blo[0][1]), "+v"(bhi[0][1]), "+v"(blo[0][2]), "+v"(bhi[0][2]), "+v"(blo[0][3]),
                 "+v"(bhi[0][3]), "+v"(blo[0][4]), "+v"(bhi[0][4]), "+v"(alo[0][0]), "+v"(ahi[0][0]), "+v"(alo[0][1]), "+v"(ahi[0][1]), "+v"(alo[0][2]),
                 "+v"(ahi[0][2]), "+v"(alo[0][3]), "+v"(ahi[0][3]), "+v"(alo[0][4]), "+v"(ahi[0][4]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[0][0]), "+v"(bhi[0][0]), "+v"(blo[0][1]), "+v"(bhi[0][1]), "+v"(blo[0][2]), "+v"(bhi[0][2]), "+v"(blo[0][3]),
                 "+v"(bhi[0][3]), "+v"(alo[0][0]), "+v"(ahi[0][0]), "+v"(alo[0][1]), "+v"(ahi[0][1]), "+v"(alo[0][2]), "+v"(ahi[0][2]), "+v"(alo[0][3]),
                 "+v"(ahi[0][3]), "+v"(alo[0][4]), "+v"(ahi[0][4]));
  __builtin_amdgcn_sched_barrier(0);
  auto k_loop = [&](auto bias_tag) {
    unsigned so = 0;
    for (int t = 0; t < nk; ++t) {
      unsigned son = so + Cfg::STAGE; if (son == W160_NS * Cfg::STAGE) son = 0;
      // first sub-step: MFMAs of (t, 0), reads of (t, 1) from the same stage
      substep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::false_type{}, bias_tag, 0u);
      // second sub-step: MFMAs of (t, 1); behind its first row barrier t + 1, then the reads of (t + 1, 0) from the next stage
      substep(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::true_type{}, bias_tag, son);
      so = son;
    }
  };
  if (bmask) k_loop(std::true_type{}); else k_loop(std::false_type{});
  // the reads requested in the last sub-step (slab nk: the zero page's stage) must return before the registers and the LDS are given up
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_setprio(0);
  W160_STAMP(2); W160_STAMP_RT(5);

  // ---- epilogue: fp32, register-direct ----
  const int mb = m0 + wr * 80, nb = n0 + wc * (BN_ / 2);
  const int mode = gridDim.y > 1 ? 2 : (p.accumulate ? 1 : 0);
  if (dbias != nullptr && bmask) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if ((bmask & (1u << i)) && lane < 16) {           // swapped operands: every register of the result holds the sum of row lane & 15
        const int m = mb + i * 16 + lane;
        if (m < p.M) {
          const float v = accb[i][0] * p.alpha;
          if (mode == 2) unsafeAtomicAdd(dbias + m, v);
          else dbias[m] = mode == 1 ? dbias[m] + v : v;
        }
      }
    }
  }
  float* const Cp = (float*)(p.nbatch ? p.Cb[blockIdx.z] : p.C);
  if (mode == 2) {
    // K split over workgroups: fp32 atomics, ROW-CONTIGUOUS.  Straight from the accumulator layout (acc[i][j][r] = C[mb + 16 i + (lane & 15)]
    // [nb + 16 j + 4 (lane >> 4) + r]) one atomic instruction scatters 64 dwords over 16 rows and ran the 64^2-level weights at a quarter
    // of the ~1.3 TB/s the memory-side atomic units take whole lines at (first version: 5120 x 640 in 179 us against 155 for the 128 x 128
    // kernel).  So the wave's 80 x BW block goes through a wave-private LDS image [80][BW + 4] and leaves as runs of 64 consecutive floats.
    // (The producers' last past-the-end pieces must have landed before the ring is written over: one more barrier, taken by both roles.)
    W160_BAR();
    constexpr int BW = BN_ / 2, LDW = BW + 4;
    static_assert(4 * 80 * LDW * 4 <= Cfg::SMEM, "the four waves' staging images fit the ring");
    float* img = (float*)smem + wave * (80 * LDW);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float4_t v = acc[i][j];
        v *= p.alpha;
        *(float4_t*)(img + (i * 16 + (lane & 15)) * LDW + j * 16 + (lane >> 4) * 4) = v;
      }
    // (no wait: the image is wave-private and a wave's LDS operations complete in order)
    for (int e = lane; e < 80 * BW; e += 64) {
      const int m = e / BW, n = e - m * BW;
      if (mb + m < p.M && nb + n < p.N) unsafeAtomicAdd(Cp + (long)(mb + m) * p.ldc + nb + n, img[m * LDW + n]);
    }
    return;
  }
#pragma unroll
  for (int half = 0; half < NJ / 2; ++half) {
    float4_t pair[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i) { pair[i][0] = acc[i][2 * half]; pair[i][1] = acc[i][2 * half + 1]; }
    reg_epilogue_64x32<1, 5>(p, Cp, pair, mb, nb + half * 32, lane);
  }
  if constexpr (NJ & 1) {
    float4_t last[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) last[i] = acc[i][NJ - 1];
    reg_epilogue_col16<1, 5>(p, Cp, last, mb, nb + (NJ - 1) * 16, lane);
  }
#ifdef NK_W160_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W160_STAMP(3);
#endif
}
#ifdef NK_W160_STAMPS
extern "C" int nk_debug_w160_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nk_w160_stamp_buf), (size_t)nwg * 12 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// NK_GEMM_W160: 0 = never; 1 (default) = by shape; 2 = every eligible launch (tests: ragged shapes, K splits)
static int w160_mode() {
  int mode = 1;
  if (const char* e = getenv("NK_GEMM_W160")) mode = atoi(e);     // read per call: tools and tests flip it in-process
  return mode;
}
struct W160Plan { int bn, splitk; };
// tile width and token split for this weight gradient, {0, 0} when the kernel does not take it
static W160Plan w160_plan(const NkGemmParams& p, int amode, int bmode, int out_f32, int allow_splitk) {
  const int mode = w160_mode();
  if (!mode || amode != OP_MC || bmode != OP_MC || !out_f32) return {0, 0};
  if (p.nbatch > NK_MAX_BATCH) return {0, 0};
  const int nb = p.nbatch ? p.nbatch : 1;
  const long nk = (p.K + BK - 1) / BK;
  if (mode == 2) {      // tests: everything, split when asked to by a second variable
    int sk = 1;
    if (const char* e = getenv("NK_GEMM_W160_SPLIT")) sk = atoi(e);
    if (sk < 1 || !allow_splitk || nk < 2 * sk) sk = 1;
    return {p.N % 160 == 0 || p.N % 128 != 0 ? 160 : 128, sk};
  }
  if (p.M % W160_BM) return {0, 0};                        // rows of the weight in whole 160-row tiles (every 640 / 1280-level Linear)
  if (nk < 32) return {0, 0};                              // (the 308-token context projections stay where they are)
  W160Plan best = {0, 0};
  double best_us = 1e30;
  for (int bn = 160; bn >= 128; bn -= 32) {
    const long ntn = (p.N + bn - 1) / bn;
    if (ntn * bn * 100 > (long)p.N * 104) continue;        // at most 4 % of a column tile wasted
    const long tiles = (long)(p.M / W160_BM) * ntn * nb;
    for (int sk = 1; sk <= 8; ++sk) {
      if (sk > 1 && (!allow_splitk || nk / sk < 32)) break;
      const long wgs = tiles * sk, rounds = (wgs + 255) / 256;
      if ((double)wgs < 0.85 * (double)(rounds * 256)) continue;
      // calibrated on tools/bench_w160.py (round 6): a 160 x 160 tile takes ~0.66 us per 64-token slab at the clock the chip holds under this
      // load; the atomics of a split launch all arrive at its end (one round: nothing left to hide them behind) at ~0.6 TB/s
      const double us = (double)rounds * (double)((nk + sk - 1) / sk) * 0.66 * bn / 160.0 + (sk > 1 ? sk * (double)p.M * p.N * nb * 4.0 / 0.6e6 : 0.0);
      if (us < best_us) { best_us = us; best = {bn, sk}; }
    }
  }
  return best;
}
template <int BN_>
static int launch_w160_as(NkGemmParams& p, int splitk, hipStream_t stream) {
  auto kern = nk_gemm_w160_kernel<BN_>;
  nk_optin_lds((const void*)kern, W160Cfg<BN_>::SMEM);
  dim3 grid(((p.M + W160_BM - 1) / W160_BM) * ((p.N + BN_ - 1) / BN_), splitk, p.nbatch ? p.nbatch : 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), W160Cfg<BN_>::SMEM, stream, p);
  return nk_check_launch("nk_gemm_w160_kernel");
}
