// Two-group staggered ring kernel of the MFMA tile engine (included by gemm.hip; not a stand-alone translation unit).
//
// Why it exists.  SDXL at batch 4 / 1024^2 puts 4096 tokens in its 1280-channel levels (60 of the 70 transformer blocks)
// and 16384 in the 640-channel ones, so the GEMMs are 4096 x {1280, 3840, 5120, 10240} and 16384 x {640, 1920, 5120}.
// With 128 x 128 tiles these give 320 / 960 / 1280 / 2560 and 640 / 1920 / 5120 tiles against 512 slots (2 workgroups per
// CU): 62 % / 94 % / 83 % / 100 % and 62 % / 94 % / 100 % full rounds -- the 1280- and 640-wide outputs (three to four GEMMs
// per transformer block, forward and backward) ran at 410-480 TFLOP/s for that reason alone.  All those widths are
// multiples of 160, and 128 x 160 tiles at ONE workgroup per CU quantise exactly: 256, 768, 1024, 2048 and 512, 1536, 4096
// tiles = whole rounds of 256.  One workgroup per CU needs a loop that keeps the MFMA pipes busy on its own, which the
// lock-step ring kernel (nk_gemm_ring_kernel: every wave reads, then every wave multiplies) does not.
//
// Structure (the two-group idea of nk_gemm_xl2g_kernel at k-step granularity):
//   * 8 waves = 2 groups x 4.  Group g owns columns [g*BN/2, (g+1)*BN/2) of the tile, its 4 waves 32 rows each: wave tile
//     32 x 64 (BN = 128) or 32 x 80 (BN = 160): 2 A + 4|5 B fragments and 8|10 MFMAs (16x16x32) per k sub-step.
//   * a k-step of a wave is  R { fragment reads of slab t; LDS-DMA of slab t+2; s_waitcnt vmcnt(pieces per slab) }  barrier
//     M { lgkmcnt(0); 16|20 MFMAs }  barrier, and group 1 runs ONE BARRIER BEHIND group 0: while one group multiplies the other
//     reads LDS and stages -- the MFMA pipes and the LDS / texture path alternate owners instead of idling together.
//   * ring of FOUR 36 KiB stages, prefetch distance two slabs, never drained (counted vmcnt; past-the-end slabs come from the
//     zero page so the count is the same in every iteration).  With barriers numbered so that group 0 passes 2t between R and
//     M of slab t and 2t+1 after M (group 1: 2t+1 and 2t+2):
//       RAW  slab t+1 is first read after barrier 2t+1; every wave waited for its own pieces of it before its barrier 2t / 2t+1.
//       WAR  slab t+2 overwrites the stage of slab t-2, whose last reads (group 1) retired right after barrier 2t-3; the
//            first DMA into it is issued after barrier 2t-1.
//   * operands: k-contiguous (KC: [rows][64 k] image, XOR-swizzled 16-byte slots, ds_read_b128) or r-contiguous (MC: staged
//     as it lies in memory, [64 k][rows], read with the transposing ds_read_b64_tr_b16).  The 160-wide MC image has 320-byte
//     k-rows; its 8 k-rows that one half-wave touches fall on 8 distinct 32-byte bank windows once the rows with (k>>3)&1 set
//     swap their 32-byte column-block pairs (source chunk c ^ 2): conflict-free without any other swizzle.
//   * swapped-operand MFMAs and the register-direct permlane16_swap epilogue (no LDS staging): bf16 with fused bias / row
//     vector / residual, or fp32 (weight gradients; overwrite or read-add-write: no K split here, so no atomics).
// Linear forward (KC x KC), dgrad (KC x MC) and wgrad (MC x MC, also the batched form); convolutions keep the gather kernels.
#pragma once

#define G2_BM 128
// Diagnostic build only (make EXTRA=-DNK_G2_STAMPS; tools/g2_stamps.py): wave 0 of every workgroup stamps s_memtime at entry, after the
// prologue, after the k loop and after the epilogue, plus s_memrealtime around the loop (the clock the chip holds: cdna guide section 7).
// In the shipped library none of this exists.
#ifdef NK_G2_STAMPS
__device__ unsigned long long nk_g2_stamp_buf[8 * 4096];
#define G2_STAMP(slot) do { if (tid == 0 && blockIdx.x < 4096 && blockIdx.z == 0) nk_g2_stamp_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define G2_STAMP_RT(slot) do { if (tid == 0 && blockIdx.x < 4096 && blockIdx.z == 0) nk_g2_stamp_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G2_STAMP(slot)
#define G2_STAMP_RT(slot)
#endif
#define G2_STAGE_BYTES 36864                     // A image 16 KiB + B image up to 20 KiB
#define G2_NS 4
#define G2_SMEM_BYTES (G2_NS * G2_STAGE_BYTES)   // 147456

// one operand of the tile: ROWS rows (128 for A; 128 or 160 for B), staged by NSW waves (all 8 of nk_gemm_g2_kernel, the 4 producer waves of
// nk_gemm_g2p_kernel): piece pc = wave + NSW i
template <int MODE, int ROWS, int NSW = 8>
struct OpG2 {
  static constexpr int NPC = ROWS / 8;                 // 1 KiB pieces per slab
  static constexpr int NPW = (NPC + NSW - 1) / NSW;    // pieces per wave (waves past NPC - NSW*(NPW-1) issue one fewer)
  static constexpr int CH = ROWS / 8;                  // MC: 16-byte chunks per k-row
  const bf16_t* rp[NPW];                               // running source pointer of each piece of this lane (KCG / MCT: the operand's base)
  int kk[NPW];                                         // KC: k offset of the lane's chunk (same for all pieces); MC: k row of the piece
  bool ok[NPW];                                        // row / r-chunk in range
  long step;                                           // pointer advance per slab
  int kcur;
  int left = 0x7fffffff, kwrap = 0x7fffffff;           // slabs still to hand out; k at which the rotated order wraps to 0 (rotate())
  long wrap = 0;                                       // pointer rewind at the wrap
  int pn[NPW], pbh[NPW], pbw[NPW];                     // KCG: pixel of the piece's row (image, top-left input row / column of its window);
                                                       // for plain windows pn is the element offset of (n, bh, bw, the lane's chunk) instead
  long cofs[NPW];                                      // MCT: r-chunk offset of the piece
  const NkGather* g;                                   // KCG
  const NkTapW* tw;                                    // MCT

  __device__ __forceinline__ int pieces(int wave) const { return wave + NSW * (NPW - 1) < NPC ? NPW : NPW - 1; }

  __device__ __forceinline__ void init(const bf16_t* P, long ld, int R, int r0, int wave, int lane, const NkGather* g_ = nullptr,
                                       const NkTapW* tw_ = nullptr) {
    kcur = 0;
    g = g_; tw = tw_;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pc = wave + NSW * i;
      if constexpr (MODE == OP_KCG) {                               // conv activations: the KC image, rows = output pixels
        const int row = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        ok[i] = (pc < NPC) && (r0 + row < R);
        kk[i] = chunk * 8;
        rp[i] = P;
        const unsigned p_ = ok[i] ? (unsigned)(r0 + row) : 0u;
        const unsigned n = fdiv(p_, g->fHoWo);
        const unsigned rem = p_ - n * g->fHoWo.d;
        const unsigned ph = fdiv(rem, g->fWo);
        const unsigned pw = rem - ph * g->fWo.d;
        pn[i] = (int)n;
        pbh[i] = (int)ph * g->rs + g->off_h;
        pbw[i] = (int)pw * g->rs + g->off_w;
        if (g->div == 1) pn[i] = ((pn[i] * g->H + pbh[i]) * g->W + pbw[i]) * g->C + chunk * 8;    // plain window: element offset instead
        step = 0;
      } else if constexpr (MODE == OP_MCT) {                        // conv-dgrad weights: the MC image, k = (tap, co), r = ci
        const int S = 64 * pc + lane;
        const int k = S / CH, c = S - k * CH;
        const int src = CH == 16 ? (c ^ mc_swz(k)) : (c ^ (((k >> 3) & 1) << 1));
        ok[i] = (pc < NPC) && (r0 + src * 8 < R);
        kk[i] = k;
        rp[i] = P;
        cofs[i] = (long)k * tw->co_stride + r0 + src * 8;
        step = 0;
      } else if constexpr (MODE == OP_KC) {
        const int row = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);                 // slot (lane&7) of row (lane>>3) holds source chunk slot ^ (row & 7)
        ok[i] = (pc < NPC) && (r0 + row < R);
        kk[i] = chunk * 8;
        rp[i] = P + (long)(r0 + row) * ld + chunk * 8;
        step = BK;
      } else {                                                      // OP_MC
        const int S = 64 * pc + lane;
        const int k = S / CH, c = S - k * CH;
        const int src = CH == 16 ? (c ^ mc_swz(k)) : (c ^ (((k >> 3) & 1) << 1));
        ok[i] = (pc < NPC) && (r0 + src * 8 < R);
        kk[i] = k;
        rp[i] = P + (long)k * ld + r0 + src * 8;
        step = (long)BK * ld;
      }
    }
  }
  // Rotated k order: hand out slabs s0, s0 + 1, ..., nk - 1, 0, ..., s0 - 1 (then the zero page).  Every XCD of a launch reads the WHOLE of the
  // operand its tiles share (the weights of a forward / input-gradient GEMM), and with all eight walking k in step each line of it is missed by
  // eight L2s at the same moment: cold weights cost a 20-k-step GEMM 10-20 % (tools/bench_cold.py).  Started an eighth of K apart, one XCD's miss
  // has filled the Infinity Cache by the time the next XCD asks.  (Sums the k-slabs in a different order per XCD: deterministic, not bit-equal to the
  // unrotated order.)
  __device__ __forceinline__ void rotate(int nk, int s0) {
    left = nk;
    kwrap = nk * BK;
    wrap = (long)nk * step;
    kcur = s0 * BK;
    if constexpr (MODE == OP_KC || MODE == OP_MC) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) rp[i] += (long)s0 * step;
    }
  }
  // sources of the next slab (advances the running state); past K or out of range: the zero page
  __device__ __forceinline__ void next_sources(int K_, const bf16_t* (&src)[NPW]) {
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
    const int K = left > 0 ? K_ : 0;                  // (every slab handed out: the zero page from here on)
    if constexpr (MODE == OP_KCG) {
      // (channel counts here are multiples of 64 -- use_g2() asks for it -- so a 64-deep slab lies inside ONE tap: the tap
      // decode is wave-uniform scalar work, and only the window's bounds test and the address are per lane)
      const bool kv = kcur < K;
      const unsigned ku = kv ? (unsigned)kcur : 0u;
      const unsigned tap = fdiv(ku, g->fC);
      const int c0 = (int)(ku - tap * g->fC.d);
      const unsigned kh = fdiv(tap, g->fKW);
      const int kw = (int)(tap - kh * g->fKW.d);
      if (g->div == 1) {        // plain window: the tap is one scalar offset on top of the pixel's own (OperandDMA::sources)
        const int dh = (int)kh * g->ks, dw = kw * g->ks;
        const int tapoff = (dh * g->W + dw) * g->C + c0;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const bool v = kv && ok[i] && (unsigned)(pbh[i] + dh) < (unsigned)g->H && (unsigned)(pbw[i] + dw) < (unsigned)g->W;
          src[i] = v ? rp[i] + (long)(pn[i] + tapoff) : zp;
        }
      } else {
        const int c = c0 + kk[0];
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          bool v = kv && ok[i];
          const long off = gather_offset(*g, pn[i], pbh[i], pbw[i], (int)kh, kw, c, v);
          src[i] = v ? rp[i] + off : zp;
        }
      }
    } else if constexpr (MODE == OP_MCT) {
      const bool kv = kcur < K;                                     // (K and Cout are multiples of 64: a slab is one tap, all of it in range)
      const unsigned ku = kv ? (unsigned)kcur : 0u;
      const unsigned tap = fdiv(ku, tw->fCout);
      const unsigned co0 = ku - tap * tw->fCout.d;
      const bf16_t* base = rp[0] + (long)co0 * tw->co_stride + (long)tap * tw->tap_stride;
#pragma unroll
      for (int i = 0; i < NPW; ++i) src[i] = (kv && ok[i]) ? base + cofs[i] : zp;
    } else {
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        src[i] = (ok[i] && kcur + kk[i] < K) ? rp[i] : zp;
        rp[i] += step;
      }
    }
    kcur += BK;
    --left;
    if (kcur >= kwrap) {
      kcur = 0;
      if constexpr (MODE == OP_KC || MODE == OP_MC) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) rp[i] -= wrap;
      }
    }
  }
  __device__ __forceinline__ void fire(const bf16_t* const (&src)[NPW], char* img, int wave) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (i < NPW - 1 || wave + NSW * i < NPC)      // (wave-uniform)
        __builtin_amdgcn_global_load_lds((nk_gptr)src[i], (nk_lptr)(img + (wave + NSW * i) * 1024), 16, 0, 0);
  }
};

// Per-lane LDS read addresses of the fragments of a wave: NF 16-row blocks starting at row `first` of the operand image.
// KC: one ds_read_b128 per (block, k sub-step); MC: two ds_read_b64_tr_b16 (k and k+4).
template <int MODE, int ROWS, int NF>
struct FragG2 {
  unsigned a[MODE == OP_KC ? 2 : NF];     // KC: base of k sub-step 0 / 1 (blocks by immediate offset); MC: base per block
  __device__ __forceinline__ void init(unsigned img0, int first, int lane) {
    if constexpr (MODE == OP_KC) {
      const unsigned x0 = (unsigned)(((lane >> 4) ^ (lane & 7)) << 4);          // first must be a multiple of 8: row & 7 == lane & 7
      const unsigned row = img0 + (unsigned)(first + (lane & 15)) * 128u;
      a[0] = row + x0;
      a[1] = row + (x0 ^ 64u);
    } else {
      const int g4 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
      const int k = 8 * g4 + q;
      if constexpr (ROWS == 128) {
        const unsigned base = (unsigned)(k * 256 + (mc_swz(k) << 4) + (p >> 1) * 16 + (p & 1) * 8);
#pragma unroll
        for (int j = 0; j < NF; ++j) a[j] = img0 + (base ^ (unsigned)((first + 16 * j) * 2));
      } else {                                                                   // 160-wide: 320-byte k-rows
        const unsigned base = (unsigned)(k * 320 + (p >> 1) * 16 + (p & 1) * 8);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int sub = first + 16 * j;
          a[j] = img0 + base + 2u * (unsigned)((g4 & 1) ? (sub ^ 16) : sub);
        }
      }
    }
  }
};

#define G2_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define G2_RDTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

// fragment f[ks*NF + j] <- block j, k sub-step ks of the image whose per-lane addresses are `fa` (+ so: stage offset)
template <int MODE, int ROWS, int NF>
__device__ __forceinline__ void g2_read(bf16x8_t (&f)[2 * NF], const FragG2<MODE, ROWS, NF>& fa, unsigned so) {
  if constexpr (MODE == OP_KC) {
    const unsigned a0 = fa.a[0] + so, a1 = fa.a[1] + so;
    G2_RD128(f[0], a0, 0);
    G2_RD128(f[1], a0, 2048);
    if constexpr (NF > 2) { G2_RD128(f[2], a0, 4096); G2_RD128(f[3], a0, 6144); }
    if constexpr (NF > 4) { G2_RD128(f[4], a0, 8192); }
    G2_RD128(f[NF + 0], a1, 0);
    G2_RD128(f[NF + 1], a1, 2048);
    if constexpr (NF > 2) { G2_RD128(f[NF + 2], a1, 4096); G2_RD128(f[NF + 3], a1, 6144); }
    if constexpr (NF > 4) { G2_RD128(f[NF + 4], a1, 8192); }
  } else {
    constexpr int RS = ROWS * 2;          // bytes per k-row
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const unsigned aj = fa.a[j] + so;
      short4_t lo0, hi0, lo1, hi1;
      G2_RDTR(lo0, aj, 0);
      G2_RDTR(hi0, aj, 4 * RS);
      G2_RDTR(lo1, aj, 32 * RS);
      G2_RDTR(hi1, aj, 36 * RS);
      short8_t r0, r1;
      r0[0] = lo0[0]; r0[1] = lo0[1]; r0[2] = lo0[2]; r0[3] = lo0[3]; r0[4] = hi0[0]; r0[5] = hi0[1]; r0[6] = hi0[2]; r0[7] = hi0[3];
      r1[0] = lo1[0]; r1[1] = lo1[1]; r1[2] = lo1[2]; r1[3] = lo1[3]; r1[4] = hi1[0]; r1[5] = hi1[1]; r1[6] = hi1[2]; r1[7] = hi1[3];
      f[j] = __builtin_bit_cast(bf16x8_t, r0);
      f[NF + j] = __builtin_bit_cast(bf16x8_t, r1);
    }
  }
}

template <int MI>
__device__ __forceinline__ void residual_prefetch_col16(const NkGemmParams& p, uint2_t (&r)[MI], int mbase, int nbase, int lane) {
  const int n = nbase + (lane >> 4) * 4;
  const int mrow = mbase + (lane & 15);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + i * 16;
    r[i] = (n < p.N && m < p.M) ? *(const uint2_t*)(p.residual + (long)m * p.ldr + n) : (uint2_t){0u, 0u};
  }
}
// a lone 16-column block of the wave tile (the fifth of the 80-column half): the lane holds 4 consecutive columns of one row
template <int OUT_F32, int MI>
__device__ __forceinline__ void reg_epilogue_col16(const NkGemmParams& p, void* Cv, float4_t (&acc)[MI], int mbase, int nbase, int lane,
                                                   int mlimit, const uint2_t (&pre_res)[MI], bool use_pre) {      // pre_res: residual_prefetch_col16
  const int Mrows = mlimit >= 0 ? mlimit : p.M;
  const int n = nbase + (lane >> 4) * 4;
  const int mrow = mbase + (lane & 15);
  if (n >= p.N) return;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + i * 16;
    if (m >= Mrows) continue;
    float v[4] = {acc[i][0] * p.alpha, acc[i][1] * p.alpha, acc[i][2] * p.alpha, acc[i][3] * p.alpha};
    if constexpr (OUT_F32) {
      float* dst = (float*)Cv + (long)m * p.ldc + n;
      if ((p.N & 3) == 0 && (p.ldc & 3) == 0) {
        float4_t o = {v[0], v[1], v[2], v[3]};
        if (p.accumulate) o += *(const float4_t*)dst;
        *(float4_t*)dst = o;
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
      }
    } else {
      bf16_t* C = (bf16_t*)Cv;
      if ((p.N & 3) == 0 && (p.ldc & 3) == 0) {
        if (p.bias) { const float4_t b = *(const float4_t*)(p.bias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
        if (p.rowvec) {
          float t[4];
          unpack4(*(const uint2_t*)(p.rowvec + (long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n), t);
          v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        if (p.residual) {
          float t[4];
          unpack4(use_pre ? pre_res[i] : *(const uint2_t*)(p.residual + (long)m * p.ldr + n), t);
          v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        uint2_t o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2_t*)(C + (long)m * p.ldc + n) = o;
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
          float x = v[e];
          if (p.bias) x += p.bias[n + e];
          if (p.rowvec) x += bf2f(p.rowvec[(long)fdiv((unsigned)m, p.fRowsPerBatch) * p.ld_rowvec + n + e]);
          if (p.residual) x += bf2f(p.residual[(long)m * p.ldr + n + e]);
          C[(long)m * p.ldc + n + e] = f2bf(x);
        }
      }
    }
  }
}
template <int OUT_F32, int MI>
__device__ __forceinline__ void reg_epilogue_col16(const NkGemmParams& p, void* Cv, float4_t (&acc)[MI], int mbase, int nbase, int lane, int mlimit = -1) {
  const uint2_t none[MI] = {};
  reg_epilogue_col16<OUT_F32, MI>(p, Cv, acc, mbase, nbase, lane, mlimit, none, false);
}

template <int AMODE, int BMODE, int OUT_F32, int BN_>
__global__ __launch_bounds__(512, 2) void nk_gemm_g2_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HN = BN_ / 2, NJ = HN / 16;          // columns per group; 16-column blocks per wave: 4 or 5
  const int tid = threadIdx.x, lane = tid & 63;
  G2_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;

  // XCD-aware bijective remap + grouped tile order (4 x ntn patches: with 8 column tiles of 160 an XCD's 32 workgroups
  // share 4 A panels and the whole of B)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN_ - 1) / BN_, ntm = (p.M + G2_BM - 1) / G2_BM;
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int m0 = (first_m + (in_group - nt * gm)) * G2_BM, n0 = nt * BN_;
  const int nk = (p.K + BK - 1) / BK;
  const bf16_t* Ap = p.nbatch ? p.Ab[blockIdx.z] : p.A;
  const bf16_t* Bp = p.nbatch ? p.Bb[blockIdx.z] : p.B;
  void* Cp = p.nbatch ? p.Cb[blockIdx.z] : p.C;

  constexpr int AF = (AMODE == OP_KC || AMODE == OP_KCG) ? OP_KC : OP_MC;               // the LDS image: the gathers stage the dense layouts
  constexpr int BF = (BMODE == OP_KC || BMODE == OP_KCG) ? OP_KC : OP_MC;
  OpG2<AMODE, 128> oa;
  OpG2<BMODE, BN_> ob;
  oa.init(Ap, p.lda, p.M, m0, wave, lane, &p.ga, &p.tw);
  ob.init(Bp, p.ldb, p.N, n0, wave, lane, &p.gb, &p.tw);
  if (p.k_rotate) { oa.rotate(nk, (xcd * nk) >> 3); ob.rotate(nk, (xcd * nk) >> 3); }
  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  FragG2<AF, 128, 2> fa;
  FragG2<BF, BN_, NJ> fb;
  fa.init(lds0, wq * 32, lane);
  fb.init(lds0 + 16384u, grp * HN, lane);

  float4_t acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4], bfr[2 * NJ];
  // bias gradient of a weight-gradient launch (see nk_gemm_dma_kernel): the two column groups hold the same 32 rows per row quarter;
  // group g sums the 16-row block g.  Swapped operands: the row is on lane & 15, every register of the result holds its sum.
  constexpr bool CAN_BIAS = AMODE == OP_MC && OUT_F32 == 1;
  float* const dbias = CAN_BIAS ? (p.nbatch ? p.dbias_b[blockIdx.z] : p.dbias) : nullptr;
  const bool do_bias = CAN_BIAS && dbias != nullptr && nt == 0;
  float4_t accb = (float4_t){0.f, 0.f, 0.f, 0.f};

#define G2_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)
  // this wave's pieces per slab: 4 (BN 128) or 5 / 4 (BN 160, waves 0-3 / 4-7) -- the counted wait leaves exactly one slab in flight
  const bool five = ob.pieces(wave) + oa.pieces(wave) == 5;
#define G2_WAIT_ONE_SLAB()                                            \
  do {                                                                \
    if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");        \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             \
  } while (0)

  const bf16_t* sa[OpG2<AMODE, 128>::NPW];
  const bf16_t* sb[OpG2<BMODE, BN_>::NPW];
  // prologue: slabs 0 and 1 in flight, slab 0 landed for everyone
  oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);
  oa.fire(sa, smem, wave); ob.fire(sb, smem + 16384, wave);
  oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);
  oa.fire(sa, smem + G2_STAGE_BYTES, wave); ob.fire(sb, smem + G2_STAGE_BYTES + 16384, wave);
  oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);             // sources of slab 2, fired in the first R phase
  G2_WAIT_ONE_SLAB();
  G2_BAR();
  if (grp == 1) { G2_BAR(); }                                       // the second group runs one barrier behind

  G2_STAMP(1); G2_STAMP_RT(4);
  unsigned so = 0, sn = 2 * G2_STAGE_BYTES;                         // stage of slab t / of slab t + 2
  for (int t = 0; t < nk; ++t) {
    // ---- R: fragment reads of slab t, DMA of slab t + 2, wait for slab t + 1 ----
    __builtin_amdgcn_sched_barrier(0);
    g2_read<BF, BN_, NJ>(bfr, fb, so);
    g2_read<AF, 128, 2>(af, fa, so);
    oa.fire(sa, smem + sn, wave);                        // the two A pieces here, the two or three B pieces behind the barrier, in the
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // all but the two A pieces just issued: slab t + 1 is complete (this wave's share)
    G2_BAR();
    // ---- M: the wave's MFMAs; the next slab's source addresses are computed in their shadow ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    ob.fire(sb, smem + sn + 16384, wave);      // (all five pieces in the R phase instead: 1237 vs 1117 cycles per k-step, tools/g2_stamps.py, round 4)
    oa.next_sources(p.K, sa);
    ob.next_sources(p.K, sb);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)     // operands swapped (D = B.A^T): a lane holds 4 consecutive COLUMNS of one row
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks * NJ + j], af[ks * 2 + i], acc[i][j], 0, 0, 0);
    if constexpr (CAN_BIAS) {
      if (do_bias) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)     // (grp is wave-uniform: a select between two compile-time fragments, not an indexed array)
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), grp ? af[ks * 2 + 1] : af[ks * 2], accb, 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    G2_BAR();
    so += G2_STAGE_BYTES; if (so == G2_NS * G2_STAGE_BYTES) so = 0;
    sn += G2_STAGE_BYTES; if (sn == G2_NS * G2_STAGE_BYTES) sn = 0;
  }
  if (grp == 0) { G2_BAR(); }                                       // ... and the first group waits for it here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the past-the-end zero-page pieces must land before the LDS is given up
  G2_STAMP(2); G2_STAMP_RT(5);
#undef G2_BAR
#undef G2_WAIT_ONE_SLAB

  const int mb = m0 + wq * 32, nb = n0 + grp * HN;
  if constexpr (CAN_BIAS) {
    if (do_bias && lane < 16) {
      const int m = mb + grp * 16 + lane;
      if (m < p.M) dbias[m] = p.accumulate ? dbias[m] + accb[0] * p.alpha : accb[0] * p.alpha;       // (no K split in this kernel)
    }
  }
#pragma unroll
  for (int half = 0; half < NJ / 2; ++half) {
    float4_t pair[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { pair[i][0] = acc[i][2 * half]; pair[i][1] = acc[i][2 * half + 1]; }
    reg_epilogue_64x32<OUT_F32, 2>(p, Cp, pair, mb, nb + half * 32, lane);
  }
  if constexpr (NJ & 1) {
    float4_t last[2] = {acc[0][NJ - 1], acc[1][NJ - 1]};
    reg_epilogue_col16<OUT_F32, 2>(p, Cp, last, mb, nb + (NJ - 1) * 16, lane);
  }
#ifdef NK_G2_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G2_STAMP(3);
#endif
}
#ifdef NK_G2_STAMPS
extern "C" int nk_debug_g2_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nk_g2_stamp_buf), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif


// Lean DMA sources of a DENSE operand (OP_KC / OP_MC) for a producer wave (round 6): OpG2::next_sources spends ~10 vector instructions per
// piece and slab on bounds selects that are loop-invariant for all but the ragged last slab -- ~100 instructions per slab in a wave that shares
// its SIMD's issue port with two compute waves whose MFMAs hold it half the time.  Here a piece whose rows are out of range points at the zero
// page with a step of zero: a whole slab costs one 64-bit add per piece; the ragged last slab and the past-the-end slabs take wave-uniform
// side branches.  Same pieces, same LDS images, same rotated k order as OpG2.
template <int MODE, int ROWS, int NSW>
struct LeanSrcG2 {
  static_assert(MODE == OP_KC || MODE == OP_MC, "dense operands only");
  static constexpr int NPC = ROWS / 8, NPW = NPC / NSW, CH = ROWS / 8;
  static_assert(NPW * NSW == NPC, "every staging wave issues the same number of pieces");
  const bf16_t* rp[NPW];      // this lane's 16 bytes of the piece in the NEXT slab (the zero page for rows past the operand's end)
  long st[NPW];               // elements per slab (0 for zero-page pieces)
  int kk[NPW];                // KC: k offset of the lane's chunk; MC: k-row of the piece inside a slab
  int nk, klen, kslab, handed;
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int R, int r0, int pw, int lane, int K, int first_slab) {
    klen = K; nk = (K + BK - 1) / BK; kslab = first_slab; handed = 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pc = pw + NSW * i;
      bool ok;
      const bf16_t* src;
      if constexpr (MODE == OP_KC) {
        const int row = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        ok = r0 + row < R;
        kk[i] = chunk * 8;
        src = P + (long)(r0 + row) * ld + chunk * 8;
        st[i] = ok ? (long)BK : 0;
      } else {
        const int S = 64 * pc + lane;
        const int k = S / CH, c = S - k * CH;
        const int sc = CH == 16 ? (c ^ mc_swz(k)) : (c ^ (((k >> 3) & 1) << 1));
        ok = r0 + sc * 8 < R;
        kk[i] = k;
        src = P + (long)k * ld + r0 + sc * 8;
        st[i] = ok ? (long)BK * ld : 0;
      }
      rp[i] = ok ? src + (long)first_slab * st[i] : (const bf16_t*)nk_zero_page;
    }
  }
  // fire the next slab of the rotated order into `img`, then advance
  __device__ __forceinline__ void fire_next(char* img, int pw) {
    const bf16_t* zp = (const bf16_t*)nk_zero_page;
    if (handed >= nk) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) __builtin_amdgcn_global_load_lds((nk_gptr)zp, (nk_lptr)(img + (pw + NSW * i) * 1024), 16, 0, 0);
    } else if (kslab * BK + BK <= klen) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) __builtin_amdgcn_global_load_lds((nk_gptr)rp[i], (nk_lptr)(img + (pw + NSW * i) * 1024), 16, 0, 0);
    } else {
      const int kbase = kslab * BK;
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        __builtin_amdgcn_global_load_lds((nk_gptr)(kbase + kk[i] < klen ? rp[i] : zp), (nk_lptr)(img + (pw + NSW * i) * 1024), 16, 0, 0);
    }
    ++handed; ++kslab;
#pragma unroll
    for (int i = 0; i < NPW; ++i) rp[i] += st[i];
    if (kslab == nk) {
      kslab = 0;
#pragma unroll
      for (int i = 0; i < NPW; ++i) rp[i] -= (long)nk * st[i];
    }
  }
};

// ---- producer-wave variant (round 4, this session) ---------------------------------------------------------------------------------
// What the loop above waits for (tools/g2_stamps.py: 1 117 cycles per k-step at 4096 x 1280 x 1280 against an MFMA floor of 640): a wave
// issues IN ORDER, and an LDS-DMA instruction does not issue until the 64 B / clk / CU fill path takes it.  A slab is 36 KiB = 562 cycles of
// that path, so the 4-5 pieces of a wave cost it ~280 cycles of stalled issue per k-step on top of its fragment reads (~220): an R phase of
// ~560 cycles against the 320 of the other group's M phase, twice per k-step = the 1 117 measured (attn512.h met the same wall: ~960 of the
// ~4 450 cycles of an iteration).  Here the tile DMA belongs to FOUR PRODUCER WAVES that do nothing else (waves 8-11: with the cyclic wave ->
// SIMD placement one per SIMD, beside one wave of each compute group; 125-136 VGPRs allow three waves per SIMD): a compute wave's R phase is
// its fragment reads only, and the producers sit stalled at issue for as long as the fill path needs.  Same tile, same LDS images, same ring
// of four stages and the same barrier numbering (producers keep group 0's cadence: one barrier per half k-step):
//   slab s may be written after barrier 2s - 5 (its stage's last readers, group 1's R of slab s - 4, retired before barrier 2s - 7 ... 2s - 6)
//   and is first read after barrier 2s - 1: the producers fire slab t + 2 between barriers 2t - 1 and 2t, and wait for slab t + 1 (counted
//   vmcnt: one slab stays in flight) between barriers 2t and 2t + 1.
template <int AMODE, int BMODE, int OUT_F32, int BN_>
__global__ __launch_bounds__(768, 1) void nk_gemm_g2p_kernel(const NkGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HN = BN_ / 2, NJ = HN / 16;          // columns per group; 16-column blocks per wave: 4 or 5
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntn = (p.N + BN_ - 1) / BN_, ntm = (p.M + G2_BM - 1) / G2_BM;
  constexpr int GM = 4;
  const int per_group = GM * ntn;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gm = min(GM, ntm - first_m);
  const int in_group = wg - group * per_group;
  const int nt = in_group / gm;
  const int m0 = (first_m + (in_group - nt * gm)) * G2_BM, n0 = nt * BN_;
  const int nk = (p.K + BK - 1) / BK;
  const bf16_t* Ap = p.nbatch ? p.Ab[blockIdx.z] : p.A;
  const bf16_t* Bp = p.nbatch ? p.Bb[blockIdx.z] : p.B;
  void* Cp = p.nbatch ? p.Cb[blockIdx.z] : p.C;
#define G2_BAR() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0)

  if (wave >= 8) {
    // ================= producer: the tile DMA of all 128 + BN_ rows, pieces pw + 4 i =================
    const int pw = wave - 8;
    if constexpr ((AMODE == OP_KC || AMODE == OP_MC) && (BMODE == OP_KC || BMODE == OP_MC)) {
      if (p.lean_src) {     // NK_GEMM_LEAN (default 1): dense operands through LeanSrcG2
        LeanSrcG2<AMODE, 128, 4> la;
        LeanSrcG2<BMODE, BN_, 4> lb;
        const int s0 = p.k_rotate ? (xcd * nk) >> 3 : 0;
        la.init(Ap, p.lda, p.M, m0, pw, lane, p.K, s0);
        lb.init(Bp, p.ldb, p.N, n0, pw, lane, p.K, s0);
        constexpr int PPS = LeanSrcG2<AMODE, 128, 4>::NPW + LeanSrcG2<BMODE, BN_, 4>::NPW;
        la.fire_next(smem, pw); lb.fire_next(smem + 16384, pw);
        la.fire_next(smem + G2_STAGE_BYTES, pw); lb.fire_next(smem + G2_STAGE_BYTES + 16384, pw);
        if constexpr (PPS == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slab 0 landed
        G2_BAR();
        unsigned sn = 2 * G2_STAGE_BYTES;                                // stage of slab t + 2
        for (int t = 0; t < nk; ++t) {
          la.fire_next(smem + sn, pw);
          G2_BAR();                                                      // 2t
          lb.fire_next(smem + sn + 16384, pw);
          if constexpr (PPS == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // slab t + 1 landed
          G2_BAR();                                                      // 2t + 1
          sn += G2_STAGE_BYTES; if (sn == G2_NS * G2_STAGE_BYTES) sn = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G2_BAR();
        return;
      }
    }
    OpG2<AMODE, 128, 4> oa;
    OpG2<BMODE, BN_, 4> ob;
    oa.init(Ap, p.lda, p.M, m0, pw, lane, &p.ga, &p.tw);
    ob.init(Bp, p.ldb, p.N, n0, pw, lane, &p.gb, &p.tw);
    if (p.k_rotate) { oa.rotate(nk, (xcd * nk) >> 3); ob.rotate(nk, (xcd * nk) >> 3); }
    static_assert(OpG2<AMODE, 128, 4>::NPC % 4 == 0 && OpG2<BMODE, BN_, 4>::NPC % 4 == 0, "every producer issues the same number of pieces");
    constexpr int PPS = OpG2<AMODE, 128, 4>::NPW + OpG2<BMODE, BN_, 4>::NPW;      // pieces per slab and producer: 8 or 9
    const bf16_t* sa[OpG2<AMODE, 128, 4>::NPW];
    const bf16_t* sb[OpG2<BMODE, BN_, 4>::NPW];
    oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);
    oa.fire(sa, smem, pw); ob.fire(sb, smem + 16384, pw);
    oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);
    oa.fire(sa, smem + G2_STAGE_BYTES, pw); ob.fire(sb, smem + G2_STAGE_BYTES + 16384, pw);
    oa.next_sources(p.K, sa); ob.next_sources(p.K, sb);             // sources of slab 2
    if constexpr (PPS == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slab 0 landed
    G2_BAR();
    unsigned sn = 2 * G2_STAGE_BYTES;                                // stage of slab t + 2
    for (int t = 0; t < nk; ++t) {
      // (the A pieces in front of barrier 2t, the B pieces behind it: 36 KiB are ~560 cycles of the fill path, and all of them in one half k-step
      // made that half as long -- the compute waves wait at the barrier for the producers)
      oa.fire(sa, smem + sn, pw);
      oa.next_sources(p.K, sa);
      G2_BAR();                                                      // 2t
      ob.fire(sb, smem + sn + 16384, pw);
      ob.next_sources(p.K, sb);
      if constexpr (PPS == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // slab t + 1 landed
      G2_BAR();                                                      // 2t + 1
      sn += G2_STAGE_BYTES; if (sn == G2_NS * G2_STAGE_BYTES) sn = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the past-the-end zero-page pieces land before this wave gives up
    G2_BAR();
    return;
  }

  // ================= compute: nk_gemm_g2_kernel's two groups without any staging =================
  const int grp = wave >> 2, wq = wave & 3;
  constexpr int AF = (AMODE == OP_KC || AMODE == OP_KCG) ? OP_KC : OP_MC;
  constexpr int BF = (BMODE == OP_KC || BMODE == OP_KCG) ? OP_KC : OP_MC;
  typedef __attribute__((address_space(3))) const char* lds_c;
  const unsigned lds0 = (unsigned)(size_t)(lds_c)smem;
  FragG2<AF, 128, 2> fa;
  FragG2<BF, BN_, NJ> fb;
  fa.init(lds0, wq * 32, lane);
  fb.init(lds0 + 16384u, grp * HN, lane);

  float4_t acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[4], bfr[2 * NJ];
  constexpr bool CAN_BIAS = AMODE == OP_MC && OUT_F32 == 1;
  float* const dbias = CAN_BIAS ? (p.nbatch ? p.dbias_b[blockIdx.z] : p.dbias) : nullptr;
  const bool do_bias = CAN_BIAS && dbias != nullptr && nt == 0;
  float4_t accb = (float4_t){0.f, 0.f, 0.f, 0.f};

  // the residual of the epilogue (attention / FeedForward output projections, gradient joins) is fetched here, under the whole k loop: the compute
  // waves issue no other global loads, and the epilogue of a 20-k-step launch would otherwise begin with a round trip to memory
  const int mb = m0 + wq * 32, nb = n0 + grp * HN;
  uint4_t pre_r[NJ / 2][2];
  uint2_t pre_r16[2];
  const bool pre = !OUT_F32 && p.residual != nullptr && (p.N & 7) == 0 && (p.ldr & 7) == 0 && !p.nbatch;
  if constexpr (!OUT_F32) {
    if (pre) {
#pragma unroll
      for (int half = 0; half < NJ / 2; ++half) residual_prefetch_64x32<2>(p, pre_r[half], mb, nb + half * 32, lane);
      if constexpr (NJ & 1) residual_prefetch_col16<2>(p, pre_r16, mb, nb + (NJ - 1) * 16, lane);
    }
  }

  G2_BAR();
  if (grp == 1) { G2_BAR(); }                                       // the second group runs one barrier behind
  unsigned so = 0;
  for (int t = 0; t < nk; ++t) {
    // ---- R: fragment reads of slab t ----
    __builtin_amdgcn_sched_barrier(0);
    g2_read<BF, BN_, NJ>(bfr, fb, so);
    g2_read<AF, 128, 2>(af, fa, so);
    G2_BAR();
    // ---- M: the wave's MFMAs ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)     // operands swapped (D = B.A^T): a lane holds 4 consecutive COLUMNS of one row
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks * NJ + j], af[ks * 2 + i], acc[i][j], 0, 0, 0);
    if constexpr (CAN_BIAS) {
      if (do_bias) {      // two wave-uniform branches, not `grp ? af[1] : af[0]`: hipcc turned that select into an indexed copy of the fragments in
        if (grp) {        // scratch, stored right behind the asm reads and BEFORE their s_waitcnt -- garbage bias gradients (caught by the batched test)
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), af[1], accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), af[3], accb, 0, 0, 0);
        } else {
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), af[0], accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nk_ones_frag(), af[2], accb, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    G2_BAR();
    so += G2_STAGE_BYTES; if (so == G2_NS * G2_STAGE_BYTES) so = 0;
  }
  if (grp == 0) { G2_BAR(); }
#undef G2_BAR

  if constexpr (CAN_BIAS) {
    if (do_bias && lane < 16) {
      const int m = mb + grp * 16 + lane;
      if (m < p.M) dbias[m] = p.accumulate ? dbias[m] + accb[0] * p.alpha : accb[0] * p.alpha;
    }
  }
#pragma unroll
  for (int half = 0; half < NJ / 2; ++half) {
    float4_t pair[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { pair[i][0] = acc[i][2 * half]; pair[i][1] = acc[i][2 * half + 1]; }
    reg_epilogue_64x32<OUT_F32, 2>(p, Cp, pair, mb, nb + half * 32, lane, -1, pre_r[half], pre);
  }
  if constexpr (NJ & 1) {
    float4_t last[2] = {acc[0][NJ - 1], acc[1][NJ - 1]};
    reg_epilogue_col16<OUT_F32, 2>(p, Cp, last, mb, nb + (NJ - 1) * 16, lane, -1, pre_r16, pre);
  }
}

// NK_GEMM_G2P: 1 (default) = the producer-wave variant wherever the two-group kernel is selected; 0 = nk_gemm_g2_kernel (A/B runs)
static bool g2p_enabled() {
  const char* e = getenv("NK_GEMM_G2P");
  return !e || atoi(e) != 0;
}

// NK_GEMM_G2: 0 = never; 1 (default) = by shape; 2 = every eligible launch (A/B runs)
static int g2_mode() {
  int mode = 1;
  if (const char* e = getenv("NK_GEMM_G2")) mode = atoi(e);     // read per call: tools flip it in-process
  return mode;
}
// tile width for this N: 160 when it divides N (1280, 640, 1920, 3840, 5120, 10240 ...), else 128 when that wastes little
static int g2_bn(int N) {
  if (N % 160 == 0) return 160;
  const int ntn = (N + 127) / 128;
  return (long)ntn * 128 * 100 <= (long)N * 112 ? 128 : 0;
}
static bool use_g2(const NkGemmParams& p, int amode, int bmode, int out_f32, int splitk) {
  const int mode = g2_mode();
  if (!mode || splitk != 1) return false;
  const bool conv = amode == OP_KCG;
  if (conv) {   // the gather modes decode the tap once per slab: channels in whole 64-deep slabs, full slabs only
    if (p.ga.C % 64 || p.K % 64) return false;
    if (bmode == OP_MCT && p.tw.fCout.d % 64) return false;
  }
  if (!((amode == OP_KC && bmode == OP_KC) || (amode == OP_KC && bmode == OP_MC) || (amode == OP_MC && bmode == OP_MC) ||
        (conv && (bmode == OP_KC || bmode == OP_MCT))))
    return false;
  if (p.K < 2 * BK) return false;
  // not the SINGLE weight gradients: in the two-stream step they do better with the co-resident 128 x 128 kernels (187.6 vs 189.0 ms);
  // the batched ones (three 1280 x 1280 per launch = 240 tiles, one round) do better here: 49.7 vs 65.4 us alone, 177.2 vs 177.6 ms/step
  if (amode == OP_MC && bmode == OP_MC && !p.nbatch) return false;      // (round 4, with the producer-wave kernel: 158.0 / 157.9 vs 158.0 / 158.9 ms per step -- still nothing)
  const int bn = g2_bn(p.N);
  if (!bn) return false;
  if (mode == 2) return true;
  // By shape (tools/bench_g2.py, interleaved A/B on the SDXL Linear shapes).  The kernel wins where its tiles come out in ONE or
  // TWO whole rounds of 256 (one workgroup per CU: 4096 x 1280 -> 256 tiles, 16384 x 640 -> 512, 3840 x 1280 -> 240, three batched
  // 1280 x 1280 weight gradients -> 240) -- forward +12..29 %, dgrad +5..27 %, wgrad +32..35 % -- and where K is long enough (>= 40
  // slabs) to amortise a tile's prologue and epilogue over up to four rounds.  It loses where many short rounds follow each other
  // (at one workgroup per CU nothing overlaps a tile's epilogue: 16384 x 1280 x 640 forward 0.87x, 65536 x 1280 x 1280 0.67x) and
  // against the 256 x 256 two-group kernel on the shapes that one takes (4096 x 3840 / 10240 x 1280 forward 0.76-0.80x).
  const long tiles = (long)((p.M + G2_BM - 1) / G2_BM) * ((p.N + bn - 1) / bn) * (p.nbatch ? p.nbatch : 1);
  const long rounds = (tiles + 255) / 256;
  const double fill = (double)tiles / (double)(rounds * 256);
  const long nk = (p.K + BK - 1) / BK;
  if (fill < 0.85) return false;
  return rounds <= 2 || (rounds <= 4 && nk >= 40);
}

// NK_GEMM_KROT: 1 (default) = rotated k order per XCD (OpG2::rotate; the two-group, 128 x 128 double-buffer and 256 x 256 two-group kernels) in launches of at least eight slabs; 0 = every XCD starts at k = 0 (A/B runs)
static bool k_rotate_on(int k_len) {
  const char* e = getenv("NK_GEMM_KROT");
  return (!e || atoi(e) != 0) && (k_len + BK - 1) / BK >= 8;
}

template <int AMODE, int BMODE, int OUT_F32, int BN_>
static int launch_g2_as(const NkGemmParams& p_in, hipStream_t stream) {
  NkGemmParams p = p_in;
  p.k_rotate = k_rotate_on(p.K) ? 1 : 0;
  dim3 grid(((p.M + G2_BM - 1) / G2_BM) * ((p.N + BN_ - 1) / BN_), 1, p.nbatch ? p.nbatch : 1);
  if (g2p_enabled()) {
    { const char* e = getenv("NK_GEMM_LEAN"); p.lean_src = (!e || atoi(e) != 0) ? 1 : 0; }
    auto kp = nk_gemm_g2p_kernel<AMODE, BMODE, OUT_F32, BN_>;
    nk_optin_lds((const void*)kp, G2_SMEM_BYTES);
    hipLaunchKernelGGL(kp, grid, dim3(768), G2_SMEM_BYTES, stream, p);
    return nk_check_launch("nk_gemm_g2p_kernel");
  }
  auto kern = nk_gemm_g2_kernel<AMODE, BMODE, OUT_F32, BN_>;
  nk_optin_lds((const void*)kern, G2_SMEM_BYTES);
  hipLaunchKernelGGL(kern, grid, dim3(512), G2_SMEM_BYTES, stream, p);
  return nk_check_launch("nk_gemm_g2_kernel");
}
template <int AMODE, int BMODE, int OUT_F32>
static int launch_g2(const NkGemmParams& p, hipStream_t stream) {
  return g2_bn(p.N) == 160 ? launch_g2_as<AMODE, BMODE, OUT_F32, 160>(p, stream) : launch_g2_as<AMODE, BMODE, OUT_F32, 128>(p, stream);
}
