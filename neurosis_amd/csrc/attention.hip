// Flash-style multi-head attention for the SpatialTransformer blocks (gfx950 / CDNA4), forward and
// backward, softmax(Q K^T * scale) V with no mask and no dropout.
// Reference: TorchSDPCrossAttention.forward (modules/attention.py:369-417) /
// MemoryEfficientCrossAttention.forward (modules/attention.py:293-366).
//
// Layout: tokens-major "[B][L][H*D]" views with arbitrary row stride, so q/k/v can be column slices of
// one fused projection buffer and the output feeds to_out directly -- no head transposes in HBM.
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16): the score tile is computed TRANSPOSED, S^T[key][q] =
// K Q^T, so a query row lives on one lane (col = lane&31): the online-softmax state (m, l) is a per-lane
// scalar, the P^T accumulator is directly the B operand of the next product (O^T += V^T P^T) with no
// LDS round trip, and V^T / K^T / Q^T / dO^T operands come from row-major LDS tiles through the
// transposing ds_read_b64_tr_b16.
//
// forward : 1 kernel, 128 queries per workgroup (4 waves x 32), KV tiles of 64 keys double-buffered in LDS
// backward: dQ kernel (queries on lanes, loops over key tiles, no atomics; also emits delta = rowsum(dO*O));
//           dK/dV kernel (keys on lanes, loops over query tiles, no atomics).  P is recomputed from LSE.
#include "../../include/neurosis_hip.h"
#include "nk_common.h"
#include <stdlib.h>
#include <type_traits>

struct AttnParams {
  const bf16_t *Q, *K, *V, *dO;
  const bf16_t* Oc;     // forward output (backward: for delta)
  bf16_t *O, *dQ, *dK, *dV;
  float* LSE;           // [B][H][Lq] natural-log sum-exp of scaled scores
  float* delta;         // [B][H][Lq]
  int B, H, Lq, Lk, D;
  long sq, sk, sv, so;      // row (token) strides in elements
  long bq, bk, bv, bo;      // batch strides in elements
  long sdq, sdk, sdv, sdo;  // strides of gradients
  long bdq, bdk, bdv, bdo;
  float scale;
  int causal;           // forward: key j is visible to query i iff j <= i
  int qsplit;           // dK/dV kernel: workgroups per key block along the query range (partials in dkv_part)
  float* dkv_part;      // [qsplit][2][B][Lk][H*D] fp32 partial dK / dV when qsplit > 1
  int gx;               // > 0: the launch is a 1-D grid of gx * H * B workgroups and attn_wg() maps it XCD-aware; 0: the 3-D grid (x, head, batch)
};

// Which (x block, head, batch item) this workgroup is.  Workgroups go to the eight XCDs round-robin in launch order.  With the plain 3-D grid
// (x fastest) the query blocks of ONE head -- which all stream that head's K and V (the key blocks of the dK / dV kernel: its Q' and dO) --
// land on eight different XCDs: eight L2s each fetch their own copy over the fabric, at the same moment, from the same memory channels
// (profiles/r04_pmc_summary.csv: the forward fetched 89 MB per launch for 31 MB of operands, the dQ kernel 249 MB).  The 1-D launch gives every
// XCD a CONTIGUOUS range of the (batch, head, x) order instead, so a head's blocks share one L2.
__device__ __forceinline__ void attn_wg(const AttnParams& p, int& bx, int& hd, int& b) {
  if (p.gx == 0) { bx = blockIdx.x; hd = blockIdx.y; b = blockIdx.z; return; }
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int t = w / p.gx;
  bx = w - t * p.gx;
  b = t / p.H;
  hd = t - b * p.H;
}
// the grid to launch for the 3-D extent `g` (sets p.gx); NK_ATTN_XCD=0 keeps the 3-D grid (A/B runs)
static dim3 attn_grid(AttnParams& p, dim3 g) {
  const char* e = getenv("NK_ATTN_XCD");
  if (e && e[0] == '0') { p.gx = 0; return g; }
  p.gx = (int)g.x;
  return dim3(g.x * g.y * g.z, 1, 1);
}

#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1.0e30f)
#define EXP2(x) __builtin_amdgcn_exp2f(x)   // raw v_exp_f32: arguments here are <= 0 or bounded, no range fix-up needed

typedef __attribute__((address_space(3))) short4_t* lds_s4p;

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int rs_bytes, int row0, int col0, int lane) {
  // A-operand fragment of a 32x32x16 MFMA whose rows run along the tile's COLUMNS:
  //   lane (r = lane&31, h = lane>>5), element j <- tile[row0 + 4h + (j&3) + 8*(j>>2)][col0 + r]
  const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3, h = g >> 1;
  const int byte = (row0 + 4 * h + q4) * rs_bytes + (col0 + 16 * (g & 1) + 4 * p) * 2;
  short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + byte));
  short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + byte + 8 * rs_bytes));
  short8_t r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int rs_bytes, int row0, int ks, int lane) {
  // A-operand fragment whose rows are the tile's rows: elem j <- tile[row0 + (lane&31)][16ks + 8(lane>>5) + j]
  const int byte = (row0 + (lane & 31)) * rs_bytes + (16 * ks + 8 * (lane >> 5)) * 2;
  return *(const bf16x8_t*)(tile + byte);
}
__device__ __forceinline__ bf16x8_t pack_frag(const float16_t& x, int s) {
  // registers 8s..8s+7 of an accumulator tile as the bf16 B operand of k-step s
  uint4_t v;
  v.x = pack2bf(x[8 * s + 0], x[8 * s + 1]); v.y = pack2bf(x[8 * s + 2], x[8 * s + 3]);
  v.z = pack2bf(x[8 * s + 4], x[8 * s + 5]); v.w = pack2bf(x[8 * s + 6], x[8 * s + 7]);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// cooperative load of a [ROWS][DP] tile (rows >= nrows or cols >= D read as zero) into registers / LDS
template <int ROWS, int DP, int NT = 256>
struct TileLoader {
  static constexpr int CPR = DP / 8;
  static constexpr int PER = (ROWS * CPR + NT - 1) / NT;
  static constexpr int RS = DP * 2 + 16;
  uint4_t v[PER];
  // Whole tiles (every row inside the tensor, head dim == DP) are fetched from `tile = base + row0 * stride`, a wave-uniform pointer, plus
  // ONE per-thread byte offset computed once (piece i lies NT / CPR rows further: a wave-uniform step): one load instruction per piece.
  // (The general form below recomputes row / chunk / 64-bit address / two bounds predicates per piece per tile: ~100 of the ~330 vector
  // instructions of a forward iteration.)
  static __device__ __forceinline__ int piece_offset(long stride, int tid) {
    return (int)(((tid / CPR) * stride + (tid % CPR) * 8) * 2);
  }
  __device__ __forceinline__ void load_full(const bf16_t* tile, long stride, int off0) {
    static_assert((ROWS * CPR) % NT == 0 && NT % CPR == 0, "whole pieces of whole rows only");
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = *(const uint4_t*)((const char*)(tile + (long)i * (NT / CPR) * stride) + off0);
  }
  __device__ __forceinline__ void load(const bf16_t* base, long stride, int row0, int nrows, int D, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int c = tid + NT * i;
      int row = c / CPR, ch = c - row * CPR;
      uint4_t z = {0u, 0u, 0u, 0u};
      if (c < ROWS * CPR && row0 + row < nrows && ch * 8 < D)
        z = *(const uint4_t*)(base + (long)(row0 + row) * stride + ch * 8);
      v[i] = z;
    }
  }
  __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int c = tid + NT * i;
      int row = c / CPR, ch = c - row * CPR;
      if (c < ROWS * CPR) *(uint4_t*)(tile + row * RS + ch * 16) = v[i];
    }
  }
};

// ================================================================================================
// forward
// ================================================================================================
// Occupancy target 3 waves per SIMD (<= 168 VGPRs): left alone hipcc takes 192 registers (2 waves per SIMD) and these
// VALU-heavy loops stall half their cycles with nobody to cover -- measured forward 314 -> 247 us at L = 4096 (696 TFLOP/s),
// 66 -> 48 us at L = 1024; 4 waves per SIMD spills and is slower again (308 us).
template <int DP, int NW = 4>   // NW waves x 32 rows per workgroup (2: twice the workgroups for short sequences)
__global__ __launch_bounds__(NW * 64, 3) void attn_fwd_kernel(const AttnParams p) {
  constexpr int RS = DP * 2 + 16, KS = DP / 16, DT = DP / 32, TILE = 64 * RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K,V][64][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, ql = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int q0 = bx * (NW * 32) + wave * 32;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * p.D;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * p.D;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * p.D;

  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4_t z = {0u, 0u, 0u, 0u};
    int d0 = 16 * ks + 8 * h5;
    if (q0 + ql < p.Lq && d0 < p.D) z = *(const uint4_t*)(Qb + (long)(q0 + ql) * p.sq + d0);
    qf[ks] = __builtin_bit_cast(bf16x8_t, z);
  }
  float16_t oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m = NEG_BIG, l = 0.f;

  const int nt = (p.Lk + 63) / 64;
  // K/V tiles travel global -> registers -> LDS (double-buffered LDS), with the global loads TWO tiles ahead: tile t + 2 is
  // requested at the top of iteration t and written to LDS at the bottom of iteration t + 1.  One tile ahead (the first version)
  // left a single iteration of compute (~1 us) to cover a loaded-chip memory latency of 1-2 us: every shape ran ~2-2.9 us per
  // iteration, whatever its arithmetic (MFMA busy 0.18).  Two register sets, named statically (the loop is unrolled by two).
  // (head dims > 64 keep one tile ahead: their tiles are larger and a second register set would spill)
  constexpr int PFD = 1;   // (two ahead, a second register set, measured +1..2 % only: the loop was not waiting on global memory)
  TileLoader<64, DP, NW * 64> lkA, lvA;
  const int koff = TileLoader<64, DP, NW * 64>::piece_offset(p.sk, tid);      // (K and V share it: the whole-tile path asks for sk == sv)
  lkA.load(Kb, p.sk, 0, p.Lk, p.D, tid);
  lvA.load(Vb, p.sv, 0, p.Lk, p.D, tid);
  lkA.store(smem, tid);
  lvA.store(smem + TILE, tid);
  __syncthreads();

  // MASKED: tile t reaches past Lk, or the causal variant -- scores are masked.  PF_WHOLE: the tile prefetched in this iteration lies wholly
  // inside K / V and the head dim is DP (TileLoader::load_full).  Compile-time flags on purpose: as run-time, wave-uniform branches hipcc
  // flattened them into the common path -- 33 adds + 32 compares + 33 selects per iteration for a mask only the last tile needs, and the
  // general loader's address / bounds arithmetic, together ~180 of ~330 vector instructions per iteration of a VALU-bound loop.
  auto iteration = [&](int t, TileLoader<64, DP, NW * 64>& useK, TileLoader<64, DP, NW * 64>& useV, TileLoader<64, DP, NW * 64>& pfK,
                       TileLoader<64, DP, NW * 64>& pfV, auto masked_tag, auto pf_whole_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value, PF_WHOLE = decltype(pf_whole_tag)::value;
    const char* kt = smem + (t & 1) * 2 * TILE;
    const char* vt = kt + TILE;
    if constexpr (PF_WHOLE) {
      pfK.load_full(Kb + (long)(t + PFD) * 64 * p.sk, p.sk, koff);
      pfV.load_full(Vb + (long)(t + PFD) * 64 * p.sv, p.sv, koff);
    } else if (t + PFD < nt) {
      pfK.load(Kb, p.sk, (t + PFD) * 64, p.Lk, p.D, tid);
      pfV.load(Vb, p.sv, (t + PFD) * 64, p.Lk, p.D, tid);
    }
    float16_t s[2];
    if constexpr (DP == 64) {
      // All eight K fragments of the tile in ONE batch, one wait, then the eight MFMAs.  Left to hipcc (168-register cap) the
      // loop was  ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma  eight times over through one fragment register: a full LDS
      // latency in front of every MFMA.  Inline asm so that the reads stay where they are put (section 5.7 of the cdna guide).
      typedef __attribute__((address_space(3))) const char* lds_c;
      const unsigned ka = (unsigned)(size_t)(lds_c)kt + (unsigned)((lane & 31) * RS + 16 * (lane >> 5));
      bf16x8_t kf[2][4];
#define ATT_RD128(dst, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ka), "n"(OFF))
      ATT_RD128(kf[0][0], 0); ATT_RD128(kf[0][1], 32); ATT_RD128(kf[0][2], 64); ATT_RD128(kf[0][3], 96);
      ATT_RD128(kf[1][0], 32 * RS); ATT_RD128(kf[1][1], 32 * RS + 32); ATT_RD128(kf[1][2], 32 * RS + 64); ATT_RD128(kf[1][3], 32 * RS + 96);
#undef ATT_RD128
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]), "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[hf][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[hf][ks], qf[ks], s[hf], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[hf][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          s[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, RS, hf * 32, ks, lane), qf[ks], s[hf], 0, 0, 0);
      }
    }
    // the V^T fragments of the P.V product are requested NOW (DP == 64): they do not depend on P, and the softmax arithmetic below
    // (~800 issue cycles) covers their LDS latency
    short4_t vlo[2][2][2], vhi[2][2][2];
    if constexpr (DP == 64) {
      typedef __attribute__((address_space(3))) const char* lds_c;
      const int g4 = lane >> 4, i16 = lane & 15;
      const unsigned va = (unsigned)(size_t)(lds_c)vt + (unsigned)((4 * (g4 >> 1) + (i16 >> 2)) * RS + (16 * (g4 & 1) + 4 * (i16 & 3)) * 2);
#define ATT_RDTR(dst, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(OFF))
#define ATT_RDV(hf, s2, dt)                                                                                              \
      ATT_RDTR(vlo[hf][s2][dt], ((hf) * 32 + 16 * (s2)) * RS + (dt) * 64);                                               \
      ATT_RDTR(vhi[hf][s2][dt], ((hf) * 32 + 16 * (s2) + 8) * RS + (dt) * 64)
      ATT_RDV(0, 0, 0); ATT_RDV(0, 0, 1); ATT_RDV(0, 1, 0); ATT_RDV(0, 1, 1); ATT_RDV(1, 0, 0); ATT_RDV(1, 0, 1); ATT_RDV(1, 1, 0); ATT_RDV(1, 1, 1);
#undef ATT_RDV
#undef ATT_RDTR
    }
    // keys beyond Lk exist only in the last tile (wave-uniform branch); the causal variant (text transformers, L = 77)
    // masks every tile it visits
    const int kbase = t * 64;
    if constexpr (MASKED) {
      const int last = p.causal ? min(p.Lk - 1, q0 + ql) : p.Lk - 1;   // highest visible key of this lane's query
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = kbase + hf * 32 + acc_row(r, h5);
          s[hf][r] = key <= last ? s[hf][r] : NEG_BIG;
        }
    }
    float mloc = s[0][0];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[hf][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m, mloc);
    const float mc = mnew * c;
    float lsum = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = EXP2(s[hf][r] * c - mc);
        s[hf][r] = pv;
        lsum += pv;
      }
    if (__any(mnew > m)) {  // some query row's running max moved: rescale (alpha = 1 where it did not)
      const float alpha = EXP2((m - mnew) * c);
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
    }
    m = mnew;
    l += lsum;
    if constexpr (DP == 64) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0][0][0]), "+v"(vlo[0][0][1]), "+v"(vlo[0][1][0]), "+v"(vlo[0][1][1]), "+v"(vlo[1][0][0]), "+v"(vlo[1][0][1]), "+v"(vlo[1][1][0]),
                     "+v"(vlo[1][1][1]), "+v"(vhi[0][0][0]), "+v"(vhi[0][0][1]), "+v"(vhi[0][1][0]), "+v"(vhi[0][1][1]), "+v"(vhi[1][0][0]), "+v"(vhi[1][0][1]),
                     "+v"(vhi[1][1][0]), "+v"(vhi[1][1][1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8_t pf = pack_frag(s[hf], s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            short8_t r;
            r[0] = vlo[hf][s2][dt][0]; r[1] = vlo[hf][s2][dt][1]; r[2] = vlo[hf][s2][dt][2]; r[3] = vlo[hf][s2][dt][3];
            r[4] = vhi[hf][s2][dt][0]; r[5] = vhi[hf][s2][dt][1]; r[6] = vhi[hf][s2][dt][2]; r[7] = vhi[hf][s2][dt][3];
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, r), pf, oacc[dt], 0, 0, 0);
          }
        }
    } else {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8_t pf = pack_frag(s[hf], s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vt, RS, hf * 32 + 16 * s2, dt * 32, lane), pf,
                                                               oacc[dt], 0, 0, 0);
        }
    }
    if (t + 1 < nt) {     // tile t + 1 was requested an iteration ago: it has had this whole iteration to arrive
      char* nk_ = smem + ((t + 1) & 1) * 2 * TILE;
      useK.store(nk_, tid);
      useV.store(nk_ + TILE, tid);
    }
    __syncthreads();
  };
  {
    static_assert(PFD == 1, "one tile ahead");
    using T = std::true_type;
    using F = std::false_type;
    // tiles [0, nwhole) lie wholly inside K / V and need no mask; the prefetch of tile t + 1 may use the whole-tile loader while t + 1 < nwhole
    const int nwhole = (p.causal || p.D != DP || DP != 64 || p.sk != p.sv) ? 0 : p.Lk / 64;
    int t = 0;
    if constexpr (DP == 64) {
      for (; t + 1 < nwhole; ++t) iteration(t, lkA, lvA, lkA, lvA, F{}, T{});
    }
    for (; t < nwhole; ++t) iteration(t, lkA, lvA, lkA, lvA, F{}, F{});
    for (; t < nt; ++t) iteration(t, lkA, lvA, lkA, lvA, T{}, F{});
  }

  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int q = q0 + ql;
  if (q < p.Lq) {
    bf16_t* Ob = p.O + (long)b * p.bo + (long)q * p.so + (long)hd * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        int d = dt * 32 + 8 * r4 + 4 * h5;
        if (d < p.D) {
          uint2_t o;
          o.x = pack2bf(oacc[dt][4 * r4 + 0] * inv, oacc[dt][4 * r4 + 1] * inv);
          o.y = pack2bf(oacc[dt][4 * r4 + 2] * inv, oacc[dt][4 * r4 + 3] * inv);
          *(uint2_t*)(Ob + d) = o;
        }
      }
    if (h5 == 0) p.LSE[((long)b * p.H + hd) * p.Lq + q] = m * p.scale + logf(l);
  }
}

// ================================================================================================
// head dim 64 (round 4): the SDXL / text-tower shape, rebuilt around VALU issue and LDS-DMA staging.
// ================================================================================================
// One LDS image for a [rows][64] bf16 tile (128-byte rows), filled by LDS-DMA and read BOTH by rows (ds_read_b128: the A operand of
// S^T = K Q^T, dP^T = V dO^T, S = Q K^T, dP = dO V^T) and by columns (ds_read_b64_tr_b16: V^T, K^T, Q^T, dO^T operands): the 16-byte
// chunk c of row r lives at  r * 128 + ((c ^ a64_swz(r)) << 4),  a64_swz(r) = bit 1 of r -> bit 2, bits 2-3 of r -> bits 0-1.
//  * row read (lane&31 = row, 16 lanes per LDS pass = rows {0-3,12-15,20-27} / {4-11,16-19,28-31}, one chunk): the 8 even (odd) rows of
//    a pass carry 8 different swizzles -> 16 different 16-byte bank windows: conflict-free;
//  * transposed read (a 32-lane half = 4 consecutive rows x 64 bytes): bit 2 of the swizzle differs between rows r and r + 2, so the
//    four rows land in the four 64-byte windows of the 256-byte bank line: conflict-free.
// A DMA piece (one wave instruction) deposits 1 KiB = 8 rows linearly; the swizzle is applied on the SOURCE side (lane -> row l >> 3,
// slot l & 7 <- chunk (l & 7) ^ swz(row)).  Rows past the end of the tensor are clamped to its last row (their scores are masked; a
// zero page is not needed and stale LDS can never reach an MFMA as NaN).
typedef __attribute__((address_space(1))) const void* att_gptr;
typedef __attribute__((address_space(3))) void* att_lptr;
typedef __attribute__((address_space(3))) const char* lds_c;
__device__ __forceinline__ int a64_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// LDS-DMA of [ROWS][64] tiles by a workgroup of NW waves: ROWS / 8 pieces, PPW per wave
template <int ROWS, int NW>
struct Dma64 {
  static constexpr int PPW = ROWS / 8 / NW;
  static_assert(PPW * 8 * NW == ROWS, "whole pieces per wave");
  int off[PPW];      // byte offset of this lane's 16 bytes of piece j from the tile's first row
  __device__ __forceinline__ void init(long stride, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int row = (wave * PPW + j) * 8 + (lane >> 3);
      off[j] = (int)((row * stride + ((lane & 7) ^ a64_swz(row)) * 8) * 2);
    }
  }
  // tile whose rows all exist: `tile` = address of its first row (wave-uniform)
  __device__ __forceinline__ void issue_whole(const bf16_t* tile, char* img, int wave) const {
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      __builtin_amdgcn_global_load_lds((att_gptr)((const char*)tile + off[j]), (att_lptr)(img + (wave * PPW + j) * 1024), 16, 0, 0);
  }
  // general (first / last tiles only: recomputes what issue_whole keeps in registers): rows row0 + r >= nrows are fetched from row nrows - 1
  __device__ __forceinline__ void issue(const bf16_t* base, long stride, int row0, int nrows, char* img, int wave, int lane) const {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int row = (wave * PPW + j) * 8 + (lane >> 3);
      const int r = min(row0 + row, nrows - 1);
      __builtin_amdgcn_global_load_lds((att_gptr)(base + (long)r * stride + ((lane & 7) ^ a64_swz(row)) * 8), (att_lptr)(img + (wave * PPW + j) * 1024), 16, 0, 0);
    }
  }
};
// per-lane byte offsets (tile-relative) of the row-read fragments: k-step ks of rows [r0, r0 + 32), r0 a multiple of 32 (immediate)
__device__ __forceinline__ void a64_row_bases(unsigned (&a)[4], int lane) {
  const int r = lane & 31, f = a64_swz(r);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a[ks] = (unsigned)(r * 128 + (((2 * ks + (lane >> 5)) ^ f) << 4));
}
// ... and of the transposed-read fragments: [j = low / high 8 rows of a 16-row block][dt = columns 0-31 / 32-63]; the block's first
// row (a multiple of 16) goes into the immediate
__device__ __forceinline__ void a64_tr_bases(unsigned (&a)[2][2], int lane) {
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = ((q4 >> 1) << 2) | ((h + 2 * j) & 3);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      a[j][dt] = (unsigned)((4 * h + q4 + 8 * j) * 128 + (((4 * dt + 2 * g1 + (pp >> 1)) ^ f) << 4) + 8 * (pp & 1));
  }
}
#define A64_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
// an empty asm that "rewrites" x: pure instructions (MFMAs) float freely past asm volatile statements and sched_barriers when the block is
// linearised, so a chain that must precede the next batch of LDS reads (register budget) is pinned by making those reads' predecessor
// in the asm chain consume its result
#define A64_PIN(x) asm volatile("" : "+v"(x))
#define A64_RDTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
__device__ __forceinline__ bf16x8_t a64_join(const short4_t& lo, const short4_t& hi) {
  short8_t r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// Row-per-lane epilogue of a [64 columns][32 rows] accumulator pair (acc[dt][4 r4 + e] = column 32 dt + 8 r4 + 4 h5 + e of this lane's row):
// a row is split across the half-waves (lane: columns 8 k .. 8 k + 3, lane + 32: 8 k + 4 .. 8 k + 7, k = 4 dt + r4).  One
// v_permlane32_swap per register pairs the column groups k and k + 1, so that every lane stores 16 contiguous bytes: four stores per lane
// instead of eight (cdna guide T21: these tails are store-issue-bound; forward 39.1 -> 36.6 us at B = 4, H = 20, L = 1024, cross-attention
// 18.1 -> 15.6 us).  EVERY lane of the wave must call it (the swaps); `live` only gates the stores.  `row` = the lane's row, 16-byte aligned.
__device__ __forceinline__ void a64_store_row(bf16_t* row, bool live, const float16_t& t0, const float16_t& t1, float scale, int h5) {
  uint2_t o2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float16_t& t = k < 4 ? t0 : t1;
    o2[k].x = pack2bf(t[4 * (k & 3) + 0] * scale, t[4 * (k & 3) + 1] * scale);
    o2[k].y = pack2bf(t[4 * (k & 3) + 2] * scale, t[4 * (k & 3) + 3] * scale);
  }
  bf16_t* dst = row + 8 * h5;
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const auto sx = __builtin_amdgcn_permlane32_swap(o2[k].x, o2[k + 1].x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(o2[k].y, o2[k + 1].y, false, false);
    const uint4_t w = {sx[0], sy[0], sx[1], sy[1]};
    if (live) *(uint4_t*)(dst + 8 * k) = w;
  }
}

#define A64_WR128(addr, val) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(val) : "memory")
#define A64_WR64(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(OFF) : "memory")
// ---- forward ----
// The generic kernel above spends, per 64-key tile and wave, 16 MFMAs (512 matrix-pipe cycles) against ~200 vector issue slots (~930
// cycles: 32 fmax, 32 fma, 32 v_exp at two slots each, 32 adds, 16 conversions, the alpha path, 4 ds_write_b128 of the staged tile):
// the loop is VALU-issue-bound (profiles/r02_pmc_attention.txt).  Here the per-element work is  v_exp_f32 + v_add_f32  and half a
// conversion:
//  * Q is multiplied by scale * log2(e) ONCE, in its registers (bf16, as the reference's own math path rounds q * scale): scores arrive
//    in log2 units;
//  * the running maximum m is SUBTRACTED BY THE MFMA: the first MFMA of a score chain takes a 16-register block holding -m as its C
//    operand (D != C is free), so  S' = K Q'^T - m  needs no vector instruction;
//  * no per-tile maximum: p = exp2(S') is computed optimistically and the tile's row sum -- needed anyway -- is the overflow detector.
//    While every lane's sum stays <= RESCALE_SUM (2^13: any p <= 2^13 is as exact in bf16 as a p <= 1 and far from fp32 overflow) m is
//    left alone (cdna guide T13, "defer-max", here without even computing the max).  Otherwise (wave-uniform, rare after the first
//    tile, which always takes it) the scores are recomputed from the K tile still in LDS, the maximum taken, O / l / m rescaled ONCE
//    and the tile exponentiated against the new m -- the textbook order, so nothing is ever scaled twice or not at all;
//  * K / V tiles arrive by LDS-DMA two tiles ahead into a ring of three stages (no staging registers, no ds_write, counted vmcnt).
#define RESCALE_SUM 8192.0f
// Diagnostic build only (make EXTRA=-DNK_ATTN_STAMPS; tools/attn_stamps.py): wave 0 of every workgroup stamps s_memtime at entry, after the
// prologue, after the key loop and after the epilogue, plus s_memrealtime around the loop.  None of it exists in the shipped library.
#ifdef NK_ATTN_STAMPS
__device__ unsigned long long nk_attn_stamp_buf[8 * 8192];
#define ATT_STAMP(slot) do { if (threadIdx.x == 0) { const unsigned w_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (w_ < 8192) nk_attn_stamp_buf[w_ * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define ATT_STAMP_RT(slot) do { if (threadIdx.x == 0) { const unsigned w_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (w_ < 8192) nk_attn_stamp_buf[w_ * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int nk_debug_attn_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nk_attn_stamp_buf), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#else
#define ATT_STAMP(slot)
#define ATT_STAMP_RT(slot)
#endif
template <int NW = 4>
__global__ __launch_bounds__(NW * 64, 3) void attn64_fwd_kernel(const AttnParams p) {
  ATT_STAMP(0);
  constexpr int TILE = 64 * 128, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [3 stages][K, V][64][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, ql = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int q0 = bx * (NW * 32) + wave * 32;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * 64;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 64;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 64;
  const int nt = (p.Lk + 63) / 64;

  Dma64<64, NW> dk, dv;
  dk.init(p.sk, wave, lane);
  dv.init(p.sv, wave, lane);
  dk.issue(Kb, p.sk, 0, p.Lk, smem, wave, lane);
  dv.issue(Vb, p.sv, 0, p.Lk, smem + TILE, wave, lane);
  if (nt > 1) {
    dk.issue(Kb, p.sk, 64, p.Lk, smem + STAGE, wave, lane);
    dv.issue(Vb, p.sv, 64, p.Lk, smem + STAGE + TILE, wave, lane);
  }

  // Q' = bf16(Q * scale * log2 e), straight into the lane's fragments.  (Whole rows through an LDS image instead -- coalesced, one more
  // barrier -- measured SLOWER: 13.3 vs 11.5 us on a cross-attention launch; the prologue is bound by the burst of every workgroup's first
  // tiles, ~11 B / clk / CU, not by the shape of these loads: tools/attn_stamps.py.)
  const unsigned smem_a = (unsigned)(size_t)(lds_c)smem;
  unsigned kab[4], vab[2][2];
  a64_row_bases(kab, lane);
  a64_tr_bases(vab, lane);
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4_t z = {0u, 0u, 0u, 0u};
    if (q0 + ql < p.Lq) z = *(const uint4_t*)(Qb + (long)(q0 + ql) * p.sq + 16 * ks + 8 * h5);
    float f[8];
    unpack8(z, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= c;
    qf[ks] = __builtin_bit_cast(bf16x8_t, pack8(f));
  }
  float16_t oacc[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; negm[r] = 0.f; }
  float m = 0.f, l = 0.f;      // m: the reference point of this lane's query row, log2 units; l: this lane's half of the row sum


  // tile 0 has landed for everyone (the Q loads and tile 0 are older than tile 1's four pieces)
  if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  ATT_STAMP(1); ATT_STAMP_RT(4);
  unsigned so = 0, sn = 2 * STAGE;      // stage of tile t / of tile t + 2
  // FIRST: tile 0 (defines m).  MASKED: tile t reaches past Lk, or the causal variant.  ISSUE: 2 = tile t + 2 lies wholly inside K / V,
  // 1 = it may not (or may not exist).  Compile-time flags: as run-time wave-uniform branches hipcc flattens them into the common path.
  auto iteration = [&](int t, auto first_tag, auto masked_tag, auto issue_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, MASKED = decltype(masked_tag)::value;
    constexpr int ISSUE = decltype(issue_tag)::value;
    const bool more = ISSUE == 2 || t + 2 < nt;
    const unsigned kt = smem_a + so, vt = kt + TILE;
    float16_t s[2];
    short4_t vlo[2][2][2], vhi[2][2][2];
    const unsigned v00 = vt + vab[0][0], v01 = vt + vab[0][1], v10 = vt + vab[1][0], v11 = vt + vab[1][1];
#define ATT_RDV(hf, s2)                                                                                                  \
    A64_RDTR(vlo[hf][s2][0], v00, ((hf) * 32 + 16 * (s2)) * 128); A64_RDTR(vlo[hf][s2][1], v01, ((hf) * 32 + 16 * (s2)) * 128); \
    A64_RDTR(vhi[hf][s2][0], v10, ((hf) * 32 + 16 * (s2)) * 128); A64_RDTR(vhi[hf][s2][1], v11, ((hf) * 32 + 16 * (s2)) * 128)
#define ATT_WAITV(hf)                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                    \
                 : "+v"(vlo[hf][0][0]), "+v"(vlo[hf][0][1]), "+v"(vlo[hf][1][0]), "+v"(vlo[hf][1][1]), "+v"(vhi[hf][0][0]), "+v"(vhi[hf][0][1]), \
                   "+v"(vhi[hf][1][0]), "+v"(vhi[hf][1][1]))
    // S'^T[key][q] = K Q'^T - m: the chains start from the -m block.  Round 6: a wave's tile time was its own chain of exposed latencies (every
    // LDS batch awaited right behind its issue, the tile DMA issued first of all, where all twelve waves of the CU ask the fill path at once),
    // ~2 500 cycles for 584 cycles of issue (tools/attn_stamps.py).  FIRST PASS of a tile: the eight K fragments AND the V^T fragments of the
    // first 32 keys are requested together (they do not depend on P; 80 transient registers at this point, within the 168), the DMA of tile
    // t + 2 is issued UNDER that latency (its issue stalls on the fill path, not on the LDS), and the two chains start behind counted waits
    // for their own four fragments.  The re-computation pass (maximum moved) reads K alone.
    auto scores = [&](auto first_pass_tag) {
      constexpr bool FP = decltype(first_pass_tag)::value;
      bf16x8_t kf[2][4];
      const unsigned k0 = kt + kab[0], k1 = kt + kab[1], k2 = kt + kab[2], k3 = kt + kab[3];
      A64_RD128(kf[0][0], k0, 0); A64_RD128(kf[0][1], k1, 0); A64_RD128(kf[0][2], k2, 0); A64_RD128(kf[0][3], k3, 0);
      A64_RD128(kf[1][0], k0, 4096); A64_RD128(kf[1][1], k1, 4096); A64_RD128(kf[1][2], k2, 4096); A64_RD128(kf[1][3], k3, 4096);
      if constexpr (FP) {
        ATT_RDV(0, 0); ATT_RDV(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE == 2) {
          dk.issue_whole(Kb + (long)(t + 2) * 64 * p.sk, smem + sn, wave);
          dv.issue_whole(Vb + (long)(t + 2) * 64 * p.sv, smem + sn + TILE, wave);
        } else if (more) {
          dk.issue(Kb, p.sk, (t + 2) * 64, p.Lk, smem + sn, wave, lane);
          dv.issue(Vb, p.sv, (t + 2) * 64, p.Lk, smem + sn + TILE, wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]));
      }
      __builtin_amdgcn_sched_barrier(0);
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][0], qf[0], negm, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][ks], qf[ks], s[0], 0, 0, 0);
      if constexpr (FP) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3]));
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][0], qf[0], negm, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][ks], qf[ks], s[1], 0, 0, 0);
      if constexpr (MASKED) {   // keys beyond Lk exist only in the last tile; the causal variant (text towers, L = 77) masks every tile
        const int last = p.causal ? min(p.Lk - 1, q0 + ql) : p.Lk - 1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[hf][r] = (t * 64 + hf * 32 + acc_row(r, h5)) <= last ? s[hf][r] : NEG_BIG;
      }
    };
    scores(std::true_type{});
    float lsum = 0.f;
    bool redo = FIRST;
    if constexpr (!FIRST) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[hf][r] = EXP2(s[hf][r]);
          lsum += s[hf][r];
        }
      redo = __any(!(lsum <= RESCALE_SUM));
      if (redo) scores(std::false_type{});      // (the V fragments requested above simply arrive earlier: this pass's waits cover them too)
    }
    if (redo) {
      float mloc = s[0][0];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[hf][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // the first tile defines m (its highest score, whatever its sign); later m only rises
      const float shift = FIRST ? mloc : fmaxf(mloc, 0.f);
      if constexpr (!FIRST) {
        const float alpha = EXP2(-shift);
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      }
      m += shift;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m;
      lsum = 0.f;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[hf][r] = EXP2(s[hf][r] - shift);
          lsum += s[hf][r];
        }
    }
    l += lsum;
    auto pv_half = [&](auto hf_tag) {
      constexpr int hf = decltype(hf_tag)::value;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t pf = pack_frag(s[hf], s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(vlo[hf][s2][dt], vhi[hf][s2][dt]), pf, oacc[dt], 0, 0, 0);
      }
    };
    ATT_WAITV(0);
    ATT_RDV(1, 0); ATT_RDV(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    pv_half(std::integral_constant<int, 0>{});
    ATT_WAITV(1);
    __builtin_amdgcn_sched_barrier(0);
    pv_half(std::integral_constant<int, 1>{});
#undef ATT_WAITV
#undef ATT_RDV
    // tile t + 1 (requested an iteration ago) has landed -- this wave's pieces by the counted wait, everyone's by the barrier, which also
    // releases tile t's stage to the DMA of tile t + 3
    __builtin_amdgcn_sched_barrier(0);
    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    so += STAGE; if (so == 3 * STAGE) so = 0;
    sn += STAGE; if (sn == 3 * STAGE) sn = 0;
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // tiles [0, nwhole) lie wholly inside K / V and need no mask
    const int nwhole = p.causal ? 0 : p.Lk / 64;
    if (nwhole > 0) iteration(0, T{}, F{}, I1{}); else iteration(0, T{}, T{}, I1{});
    int t = 1;
    for (; t + 2 < nwhole; ++t) iteration(t, F{}, F{}, I2{});
    for (; t < nwhole; ++t) iteration(t, F{}, F{}, I1{});
    for (; t < nt; ++t) iteration(t, F{}, T{}, I1{});
  }

  ATT_STAMP(2); ATT_STAMP_RT(5);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int q = q0 + ql;
  if (q < p.Lq) {
    if (h5 == 0) p.LSE[((long)b * p.H + hd) * p.Lq + q] = (m + __log2f(l)) * 0.6931471805599453f;
  }
  {
    const bool live = q < p.Lq;
    a64_store_row(p.O + (long)b * p.bo + (long)(live ? q : 0) * p.so + (long)hd * 64, live, oacc[0], oacc[1], inv, h5);
  }
#ifdef NK_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_STAMP(3);
#endif
}

// ---- backward, dQ (head dim 64) ----
// Queries on lanes, K / V tiles of 64 keys by LDS-DMA (the forward's ring).  The recomputation costs no vector instruction beyond the
// exponential and one subtraction: Q' = bf16(Q * scale * log2 e) is rebuilt in registers exactly as the forward built it and -LSE * log2(e)
// rides in as the C operand of the score chain (a 16-register block, constant for the lane's query), so p = exp2(S') and
// dS = p * (dP - delta)  (a second block for -delta does not fit the 168 registers of three waves per SIMD).  Also written out for the dK / dV kernel, which runs behind this one: -delta, -LSE * log2(e) (what ITS chains start
// from, fetched by DMA) and Q' itself as [B][H][Lq][64] (its A operand: both kernels then see bit-identical scores, and so does the
// forward's LSE).
struct Attn64Ws {       // layout of the backward workspace (floats) for head dim 64
  long ndelta, nlse2, qs, part;
  __host__ __device__ static Attn64Ws make(long B, long H, long Lq) {
    Attn64Ws w;
    const long n = (B * H * Lq + 63) & ~63l;
    w.ndelta = 0; w.nlse2 = n; w.qs = 2 * n; w.part = 2 * n + B * H * Lq * 32;
    return w;
  }
};
template <int NW = 4>
__global__ __launch_bounds__(NW * 64, 3) void attn64_bwd_dq_kernel(const AttnParams p) {
  constexpr int TILE = 64 * 128, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, ql = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int q0 = bx * (NW * 32) + wave * 32;
  const int q = q0 + ql;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * 64;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 64;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 64;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * 64;
  const bf16_t* Ocb = p.Oc + (long)b * p.bo + (long)hd * 64;
  const int nt = (p.Lk + 63) / 64;

  Dma64<64, NW> dk_, dv_;
  dk_.init(p.sk, wave, lane);
  dv_.init(p.sv, wave, lane);
  dk_.issue(Kb, p.sk, 0, p.Lk, smem, wave, lane);
  dv_.issue(Vb, p.sv, 0, p.Lk, smem + TILE, wave, lane);
  if (nt > 1) {
    dk_.issue(Kb, p.sk, 64, p.Lk, smem + STAGE, wave, lane);
    dv_.issue(Vb, p.sv, 64, p.Lk, smem + STAGE + TILE, wave, lane);
  }

  const Attn64Ws ws = Attn64Ws::make(p.B, p.H, p.Lq);
  const long row = ((long)b * p.H + hd) * p.Lq + q;
  bf16x8_t qf[4], dof[4];
  float dacc = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4_t zq = {0u, 0u, 0u, 0u}, zd = {0u, 0u, 0u, 0u}, zo = {0u, 0u, 0u, 0u};
    const int d0 = 16 * ks + 8 * h5;
    if (q < p.Lq) {
      zq = *(const uint4_t*)(Qb + (long)q * p.sq + d0);
      zd = *(const uint4_t*)(dOb + (long)q * p.sdo + d0);
      zo = *(const uint4_t*)(Ocb + (long)q * p.so + d0);
    }
    float fq[8], fo[8], fd[8];
    unpack8(zq, fq);
    unpack8(zo, fo);
    unpack8(zd, fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) { fq[e] *= c; dacc += fo[e] * fd[e]; }
    const uint4_t qs = pack8(fq);
    qf[ks] = __builtin_bit_cast(bf16x8_t, qs);
    dof[ks] = __builtin_bit_cast(bf16x8_t, zd);
    if (q < p.Lq) *(uint4_t*)((bf16_t*)(p.delta + ws.qs) + row * 64 + d0) = qs;
  }
  const float dlt = dacc + __shfl_xor(dacc, 32, 64);
  const float lse2 = q < p.Lq ? p.LSE[row] * LOG2E : 1.0e30f;       // rows past Lq: P = 0
  if (q < p.Lq && h5 == 0) {
    p.delta[ws.ndelta + row] = -dlt;
    p.delta[ws.nlse2 + row] = -lse2;
  }
  float16_t dq[2], negl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; negl[r] = -lse2; }

  unsigned kab[4], ktb[2][2];
  a64_row_bases(kab, lane);
  a64_tr_bases(ktb, lane);
  const unsigned smem_a = (unsigned)(size_t)(lds_c)smem;
  if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  unsigned so = 0, sn = 2 * STAGE;
  auto iteration = [&](int t, auto masked_tag, auto issue_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int ISSUE = decltype(issue_tag)::value;
    const bool more = ISSUE == 2 || t + 2 < nt;
    const unsigned kt = smem_a + so, vt = kt + TILE;
    const unsigned k0 = kt + kab[0], k1 = kt + kab[1], k2 = kt + kab[2], k3 = kt + kab[3];
    const unsigned v0 = vt + kab[0], v1 = vt + kab[1], v2 = vt + kab[2], v3 = vt + kab[3];
    const unsigned t00 = kt + ktb[0][0], t01 = kt + ktb[0][1], t10 = kt + ktb[1][0], t11 = kt + ktb[1][1];
    auto half = [&](auto hf_tag) {
      constexpr int hf = decltype(hf_tag)::value;
      // register budget (168 at three waves per SIMD): at most 48 transient registers beside Q', dO, -lse2 and dQ.  Round 6: the K rows AND
      // the V rows of the half are requested together (K, V, S = 48; dP starts once the K rows have died into S), the DMA of tile t + 2 is
      // issued under that latency (first half), and each chain waits for its own four fragments only -- a wave's tile used to be a chain of
      // six exposed LDS round trips behind a DMA issue that met all twelve waves of the CU at the fill path.
      bf16x8_t kf[4], vf[4];
      A64_RD128(kf[0], k0, hf * 4096); A64_RD128(kf[1], k1, hf * 4096); A64_RD128(kf[2], k2, hf * 4096); A64_RD128(kf[3], k3, hf * 4096);
      A64_RD128(vf[0], v0, hf * 4096); A64_RD128(vf[1], v1, hf * 4096); A64_RD128(vf[2], v2, hf * 4096); A64_RD128(vf[3], v3, hf * 4096);
      if constexpr (hf == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE == 2) {
          dk_.issue_whole(Kb + (long)(t + 2) * 64 * p.sk, smem + sn, wave);
          dv_.issue_whole(Vb + (long)(t + 2) * 64 * p.sv, smem + sn + TILE, wave);
        } else if (more) {
          dk_.issue(Kb, p.sk, (t + 2) * 64, p.Lk, smem + sn, wave, lane);
          dv_.issue(Vb, p.sv, (t + 2) * 64, p.Lk, smem + sn + TILE, wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
      __builtin_amdgcn_sched_barrier(0);
      float16_t s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negl, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
      __builtin_amdgcn_sched_barrier(0);
      float16_t dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], dof[ks], dp, 0, 0, 0);
      A64_PIN(dp);
      __builtin_amdgcn_sched_barrier(0);
      // the K^T fragments of this half's dQ product: requested now, covered by the exponentials
      short4_t klo[2][2], khi[2][2];
      A64_RDTR(klo[0][0], t00, (hf * 32) * 128); A64_RDTR(klo[0][1], t01, (hf * 32) * 128);
      A64_RDTR(khi[0][0], t10, (hf * 32) * 128); A64_RDTR(khi[0][1], t11, (hf * 32) * 128);
      A64_RDTR(klo[1][0], t00, (hf * 32 + 16) * 128); A64_RDTR(klo[1][1], t01, (hf * 32 + 16) * 128);
      A64_RDTR(khi[1][0], t10, (hf * 32 + 16) * 128); A64_RDTR(khi[1][1], t11, (hf * 32 + 16) * 128);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = EXP2(s[r]) * (dp[r] - dlt);             // (the softmax scale multiplies dQ once, after the loop)
      if constexpr (MASKED) {   // keys beyond Lk exist only in the last tile
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = (t * 64 + hf * 32 + acc_row(r, h5)) < p.Lk ? dp[r] : 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(klo[0][0]), "+v"(klo[0][1]), "+v"(khi[0][0]), "+v"(khi[0][1]), "+v"(klo[1][0]), "+v"(klo[1][1]), "+v"(khi[1][0]), "+v"(khi[1][1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t dsf = pack_frag(dp, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(klo[s2][dt], khi[s2][dt]), dsf, dq[dt], 0, 0, 0);
      }
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    so += STAGE; if (so == 3 * STAGE) so = 0;
    sn += STAGE; if (sn == 3 * STAGE) sn = 0;
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    const int nwhole = p.Lk / 64;
    int t = 0;
    for (; t + 2 < nwhole; ++t) iteration(t, F{}, I2{});
    for (; t < nwhole; ++t) iteration(t, F{}, I1{});
    for (; t < nt; ++t) iteration(t, T{}, I1{});
  }

  {
    const bool live = q < p.Lq;
    a64_store_row(p.dQ + (long)b * p.bdq + (long)(live ? q : 0) * p.sdq + (long)hd * 64, live, dq[0], dq[1], p.scale, h5);
  }
}

// ---- backward, dK / dV (head dim 64) ----
// Keys on lanes (K, V fragments of the wave's 32 keys live in registers), 32-query tiles of Q' (the dQ kernel's copy: [B][H][Lq][64], bf16
// (Q * scale * log2 e)), of dO and of the two row constants by LDS-DMA into a ring of three stages.  S[q][key] and dP[q][key] have the
// query on the accumulator ROW: -LSE * log2(e) and -delta are read from the stage straight into the accumulator registers the chains
// then start from (C = D), so again p = exp2(S'), dS = p * dP' and nothing else.  dK = ln 2 * dS^T Q' (= scale * dS^T Q).
template <int NW = 4>
__global__ __launch_bounds__(NW * 64, 3) void attn64_bwd_dkdv_kernel(const AttnParams p) {
  constexpr int TILE = 32 * 128, STAGE = 2 * TILE + 256;   // Q' tile, dO tile, [32] -lse2, [32] -delta
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, kl = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int kb = bx / p.qsplit, qs = bx - kb * p.qsplit;
  const int k0 = kb * (NW * 32) + wave * 32;

  const Attn64Ws ws = Attn64Ws::make(p.B, p.H, p.Lq);
  const long head_row = ((long)b * p.H + hd) * p.Lq;
  const bf16_t* Qs = (const bf16_t*)(p.delta + ws.qs) + head_row * 64;
  const float* nl2 = p.delta + ws.nlse2 + head_row;
  const float* ndl = p.delta + ws.ndelta + head_row;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 64;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 64;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * 64;
  // this workgroup's slice of the query tiles
  const int nt_all = (p.Lq + 31) / 32;
  const int per = (nt_all + p.qsplit - 1) / p.qsplit;
  const int t_lo = qs * per, t_hi = min(nt_all, t_lo + per);
  const int nt = max(t_hi - t_lo, 0);

  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4_t zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
    if (k0 + kl < p.Lk) {
      zk = *(const uint4_t*)(Kb + (long)(k0 + kl) * p.sk + 16 * ks + 8 * h5);
      zv = *(const uint4_t*)(Vb + (long)(k0 + kl) * p.sv + 16 * ks + 8 * h5);
    }
    kf[ks] = __builtin_bit_cast(bf16x8_t, zk);
    vf[ks] = __builtin_bit_cast(bf16x8_t, zv);
  }
  // DMA of query tile `qt` (first row q0) into stage `img`: one piece of Q', one of dO per wave; wave 0 adds the 256 bytes of row constants
  Dma64<32, NW> dq_, do_;
  dq_.init(64, wave, lane);
  do_.init(p.sdo, wave, lane);
  auto issue_tile = [&](int q0, char* img, bool whole) {
    if (whole) {
      dq_.issue_whole(Qs + (long)q0 * 64, img, wave);
      do_.issue_whole(dOb + (long)q0 * p.sdo, img + TILE, wave);
    } else {
      dq_.issue(Qs, 64, q0, p.Lq, img, wave, lane);
      do_.issue(dOb, p.sdo, q0, p.Lq, img + TILE, wave, lane);
    }
    if (wave == 0) {
      const int qi = min(q0 + (lane & 31), p.Lq - 1);          // (rows past Lq are masked in the loop)
      const float* src = (lane < 32 ? nl2 : ndl) + qi;
      __builtin_amdgcn_global_load_lds((att_gptr)src, (att_lptr)(img + 2 * TILE), 4, 0, 0);
    }
  };
  if (nt > 0) issue_tile(t_lo * 32, smem, false);
  if (nt > 1) issue_tile((t_lo + 1) * 32, smem + STAGE, false);

  float16_t dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

  unsigned rab[4], tab[2][2];
  a64_row_bases(rab, lane);
  a64_tr_bases(tab, lane);
  const unsigned smem_a = (unsigned)(size_t)(lds_c)smem;
  const unsigned stat_a = (unsigned)(2 * TILE + 16 * h5);     // this lane's first row constant: rows 8j + 4 h5 + (0..3), j = 0..3
  // counted waits: a wave has 2 (wave 0: 3) DMA instructions per tile in flight
#define DKV_WAIT_ONE_TILE() do { if (wave == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } while (0)
  if (nt > 1) DKV_WAIT_ONE_TILE(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  unsigned so = 0, sn = 2 * STAGE;
  // MASKED: the tile reaches past Lq (the last one).  ISSUE: 2 = tile t + 2 lies wholly inside Q / dO, 1 = it may not (or may not exist)
  auto iteration = [&](int t, auto masked_tag, auto issue_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int ISSUE = decltype(issue_tag)::value;
    const bool more = ISSUE == 2 || t + 2 < nt;
    const unsigned qt = smem_a + so, dot = qt + TILE, st = qt + stat_a;
    // register budget (168 at three waves per SIMD): K, V, dK, dV take 96; at most 48 transient registers at any point.  Round 6: the row
    // constants of BOTH chains and the Q' rows are requested together (48), the DMA of tile t + 2 is issued under the S chain's four MFMAs,
    // the dO rows are requested behind it, and every chain waits for its own operands only.
    float16_t s, dp;
    {
      float4_t c0, c1, c2, c3, e0, e1, e2, e3;
      bf16x8_t qr[4], dr[4];
      A64_RD128(c0, st, 0); A64_RD128(c1, st, 32); A64_RD128(c2, st, 64); A64_RD128(c3, st, 96);
      A64_RD128(qr[0], qt + rab[0], 0); A64_RD128(qr[1], qt + rab[1], 0); A64_RD128(qr[2], qt + rab[2], 0); A64_RD128(qr[3], qt + rab[3], 0);
      A64_RD128(e0, st, 128); A64_RD128(e1, st, 160); A64_RD128(e2, st, 192); A64_RD128(e3, st, 224);
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(qr[0]), "+v"(qr[1]), "+v"(qr[2]), "+v"(qr[3]));
      __builtin_amdgcn_sched_barrier(0);
      s = __builtin_shufflevector(__builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(c2, c3, 0, 1, 2, 3, 4, 5, 6, 7),
                                  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[ks], kf[ks], s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // (the DMA here, where the Q' rows are dead: issued in front of the chain, beside 48 live transient registers, its address arithmetic
      // pushed the V fragments into scratch)
      if constexpr (ISSUE == 2) issue_tile((t_lo + t + 2) * 32, smem + sn, true);
      else if (more) issue_tile((t_lo + t + 2) * 32, smem + sn, false);
      __builtin_amdgcn_sched_barrier(0);
      A64_RD128(dr[0], dot + rab[0], 0); A64_RD128(dr[1], dot + rab[1], 0); A64_RD128(dr[2], dot + rab[2], 0); A64_RD128(dr[3], dot + rab[3], 0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(dr[0]), "+v"(dr[1]), "+v"(dr[2]), "+v"(dr[3]));
      __builtin_amdgcn_sched_barrier(0);
      dp = __builtin_shufflevector(__builtin_shufflevector(e0, e1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(e2, e3, 0, 1, 2, 3, 4, 5, 6, 7),
                                   0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dr[ks], vf[ks], dp, 0, 0, 0);
      A64_PIN(s);
      A64_PIN(dp);
      __builtin_amdgcn_sched_barrier(0);
    }
    // dO^T fragments of the dV product: requested now, covered by the exponentials
    short4_t olo[2][2], ohi[2][2];
    {
      const unsigned o00 = dot + tab[0][0], o01 = dot + tab[0][1], o10 = dot + tab[1][0], o11 = dot + tab[1][1];
      A64_RDTR(olo[0][0], o00, 0); A64_RDTR(olo[0][1], o01, 0); A64_RDTR(ohi[0][0], o10, 0); A64_RDTR(ohi[0][1], o11, 0);
      A64_RDTR(olo[1][0], o00, 2048); A64_RDTR(olo[1][1], o01, 2048); A64_RDTR(ohi[1][0], o10, 2048); A64_RDTR(ohi[1][1], o11, 2048);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = EXP2(s[r]);
      dp[r] *= s[r];                                  // (the softmax scale multiplies dK once, after the loop)
    }
    if constexpr (MASKED) {   // query rows past Lq (clamped copies of the last row) contribute nothing
      const int q0 = (t_lo + t) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = q0 + acc_row(r, h5) < p.Lq;
        s[r] = ok ? s[r] : 0.f;
        dp[r] = ok ? dp[r] : 0.f;
      }
    }
    bf16x8_t pf[2], dsf[2];
    pf[0] = pack_frag(s, 0); pf[1] = pack_frag(s, 1);
    dsf[0] = pack_frag(dp, 0); dsf[1] = pack_frag(dp, 1);
    A64_PIN(pf[0]); A64_PIN(pf[1]); A64_PIN(dsf[0]); A64_PIN(dsf[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // Q'^T fragments of the dK product: requested in front of the dV MFMAs, which cover them
    short4_t qlo[2][2], qhi[2][2];
    {
      const unsigned q00 = qt + tab[0][0], q01 = qt + tab[0][1], q10 = qt + tab[1][0], q11 = qt + tab[1][1];
      A64_RDTR(qlo[0][0], q00, 0); A64_RDTR(qlo[0][1], q01, 0); A64_RDTR(qhi[0][0], q10, 0); A64_RDTR(qhi[0][1], q11, 0);
      A64_RDTR(qlo[1][0], q00, 2048); A64_RDTR(qlo[1][1], q01, 2048); A64_RDTR(qhi[1][0], q10, 2048); A64_RDTR(qhi[1][1], q11, 2048);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(olo[s2][dt], ohi[s2][dt]), pf[s2], dv[dt], 0, 0, 0);
    A64_PIN(dv[0]); A64_PIN(dv[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(qlo[s2][dt], qhi[s2][dt]), dsf[s2], dk[dt], 0, 0, 0);
    A64_PIN(dk[0]); A64_PIN(dk[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (more) DKV_WAIT_ONE_TILE(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    so += STAGE; if (so == 3 * STAGE) so = 0;
    sn += STAGE; if (sn == 3 * STAGE) sn = 0;
  };
#undef DKV_WAIT_ONE_TILE
  {
    using T = std::true_type;
    using F = std::false_type;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // tiles (t_lo + t) with (t_lo + t + 1) * 32 <= Lq are whole; the prefetched tile t + 2 is whole while (t_lo + t + 3) * 32 <= Lq
    const int nwhole = min(nt, p.Lq / 32 - t_lo);
    int t = 0;
    for (; t + 2 < nwhole; ++t) iteration(t, F{}, I2{});
    for (; t < nwhole; ++t) iteration(t, F{}, I1{});
    for (; t < nt; ++t) iteration(t, T{}, I1{});
  }

  const float ln2 = 0.6931471805599453f;
  const int key = k0 + kl;
  const bool live = key < p.Lk;
  if (live) {
    if (p.qsplit > 1) {
      // fp32 partials [qs][0=dK,1=dV][b][key][H*D]; summed in a fixed order by attn_dkv_reduce_kernel
      const long hd_all = (long)p.H * 64;
      const long plane = (long)p.B * p.Lk * hd_all;
      float* pk = p.dkv_part + ((long)qs * 2) * plane + ((long)b * p.Lk + key) * hd_all + (long)hd * 64;
      float* pv = pk + plane;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int d = dt * 32 + 8 * r4 + 4 * h5;
          *(float4_t*)(pk + d) = (float4_t){dk[dt][4 * r4 + 0] * ln2, dk[dt][4 * r4 + 1] * ln2, dk[dt][4 * r4 + 2] * ln2, dk[dt][4 * r4 + 3] * ln2};
          *(float4_t*)(pv + d) = (float4_t){dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1], dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]};
        }
    }
  }
  if (p.qsplit <= 1) {      // (wave-uniform; every lane takes part in the half-wave swaps of the 16-byte stores)
    const long krow = live ? key : 0;
    a64_store_row(p.dK + (long)b * p.bdk + krow * p.sdk + (long)hd * 64, live, dk[0], dk[1], ln2, h5);
    a64_store_row(p.dV + (long)b * p.bdv + krow * p.sdv + (long)hd * 64, live, dv[0], dv[1], 1.0f, h5);
  }
}

// ---- backward in ONE kernel for at most 96 keys (head dim 64): the UNet's cross-attention (77 text tokens) ----
// The two-kernel backward above costs a cross-attention layer three launches (dQ, dK / dV over query splits, their reduction), each of
// them mostly fixed cost: 23 + 17 + 6 us for 3 GFLOP (profiles/r04_attention_kernels.txt).  With every key inside one workgroup nothing
// has to be recomputed: waves 0-2 own 32 keys each (K, V fragments in registers, keys on lanes, exactly the dK / dV kernel's tile loop),
// write dS^T to LDS, and wave 3 -- which has no keys -- turns the PREVIOUS tile's dS^T into dQ^T = K^T dS^T (complete: no sum across
// workgroups) while the others are on the next tile.  Q' = bf16(Q scale log2 e), -lse2 and -delta = -rowsum(dO o O) are made on the fly
// per 32-query tile, from rows fetched two tiles ahead by LDS-DMA.
// dK / dV leave as fp32 partials per query split (summed by attn_dkv_reduce_kernel) or, with one split, directly.
#define SMALL_DSROW 72                               // bytes per row of the dS^T image [96 keys][32 queries] (64 + 8: conflict-free 8-byte writes)
__global__ __launch_bounds__(256, 2) void attn64_bwd_small_kernel(const AttnParams p) {
  constexpr int TILE = 32 * 128, STAGE = 3 * TILE + 256 + 1024;   // Q' image, dO image, [32] -lse2, [32] -delta | O rows | 4 x 256 B raw log-sum-exp
  constexpr int KIMG = 96 * 128, DSBUF = 96 * SMALL_DSROW;
  constexpr int NST = 3;                                          // stages: the tile in work and the two ahead of it
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [K image][3 stages][2 dS^T buffers]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, kl = lane & 31;
  int qs, hd, b;
  attn_wg(p, qs, hd, b);
  const float c = p.scale * LOG2E;
  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * 64;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 64;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 64;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * 64;
  const bf16_t* Ocb = p.Oc + (long)b * p.bo + (long)hd * 64;
  const float* lse = p.LSE + ((long)b * p.H + hd) * p.Lq;
  const int nt_all = (p.Lq + 31) / 32;
  const int per = (nt_all + p.qsplit - 1) / p.qsplit;
  const int t_lo = qs * per, t_hi = min(nt_all, t_lo + per);
  const int nt = max(t_hi - t_lo, 0);
  const unsigned smem_a = (unsigned)(size_t)(lds_c)smem;
  const unsigned stage_a = smem_a + KIMG, ds_a = stage_a + NST * STAGE;
  const int k0 = wave * 32;                    // waves 0-2: first key of the wave's block

  // K, V fragments of the wave's keys (waves 0-2); the K image (rows past Lk zero) for wave 3's K^T reads
  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4_t zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
    if (wave < 3 && k0 + kl < p.Lk) {
      zk = *(const uint4_t*)(Kb + (long)(k0 + kl) * p.sk + 16 * ks + 8 * h5);
      zv = *(const uint4_t*)(Vb + (long)(k0 + kl) * p.sv + 16 * ks + 8 * h5);
    }
    kf[ks] = __builtin_bit_cast(bf16x8_t, zk);
    vf[ks] = __builtin_bit_cast(bf16x8_t, zv);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = i * 32 + (tid >> 3);
    uint4_t z = {0u, 0u, 0u, 0u};
    if (row < p.Lk) z = *(const uint4_t*)(Kb + (long)row * p.sk + (tid & 7) * 8);
    A64_WR128(smem_a + (unsigned)(row * 128 + (((tid & 7) ^ a64_swz(row)) << 4)), z);
  }

  // Tiles are fetched TWO ahead (a tile's compute, ~1 us, does not cover a loaded chip's memory latency), entirely by LDS-DMA -- no staging
  // registers: wave w fetches rows 8 w .. 8 w + 7 of Q, dO and O (one 1 KiB piece each: its lane l holds chunk (l & 7) ^ swz(row) of row l >> 3
  // in all three) and those rows' log-sum-exps (a 4-byte piece).  FOUR vector-memory operations per wave per fetch, so "the older tile has
  // landed" is the counted wait vmcnt(4).  The wave then publishes ITS rows in place: Q -> Q' = bf16(Q scale log2 e), -lse2, and
  // -delta = -rowsum(dO o O) from the matching chunks (a row's 8 chunks sit on 8 adjacent lanes).
  Dma64<32, 4> dq_, do_, dc_;
  dq_.init(p.sq, wave, lane);
  do_.init(p.sdo, wave, lane);
  dc_.init(p.so, wave, lane);
  const int prow = wave * 8 + (lane >> 3);                         // the row of the tile this lane stages
  const unsigned poff = (unsigned)(wave * 1024 + lane * 16);       // its slot in each image
  auto fetch = [&](int q0, unsigned stage_off) {
    char* st = smem + KIMG + stage_off;
    dq_.issue(Qb, p.sq, q0, p.Lq, st, wave, lane);
    do_.issue(dOb, p.sdo, q0, p.Lq, st + TILE, wave, lane);
    dc_.issue(Ocb, p.so, q0, p.Lq, st + 2 * TILE + 256, wave, lane);
    __builtin_amdgcn_global_load_lds((att_gptr)(lse + min(q0 + prow, p.Lq - 1)), (att_lptr)(st + 3 * TILE + 256 + wave * 256), 4, 0, 0);
  };
  auto publish = [&](int q0, unsigned stage_off, bool younger) {
    if (younger) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned st = stage_a + stage_off;
    uint4_t zq, zd, zo;
    float lv;
    A64_RD128(zq, st + poff, 0);
    A64_RD128(zd, st + TILE + poff, 0);
    A64_RD128(zo, st + 2 * TILE + 256 + poff, 0);
    asm volatile("ds_read_b32 %0, %1" : "=v"(lv) : "v"(st + 3 * TILE + 256 + (unsigned)(wave * 256 + lane * 4)));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zq), "+v"(zd), "+v"(zo), "+v"(lv));
    const bool ok = q0 + prow < p.Lq;                               // (rows past Lq are copies of the last row: Q' = 0, P = 0)
    float fq[8], fo[8], fd[8];
    unpack8(zq, fq);
    unpack8(zo, fo);
    unpack8(zd, fd);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fq[e] = ok ? fq[e] * c : 0.f; acc += fo[e] * fd[e]; }
    const uint4_t zs = pack8(fq);
    A64_WR128(st + poff, zs);
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if ((lane & 7) == 0) {
      const float nd = ok ? -acc : 0.f, nl = ok ? -lv * LOG2E : -1.0e30f;
      asm volatile("ds_write_b32 %0, %1" : : "v"(st + 2 * TILE + (unsigned)(128 + 4 * prow)), "v"(nd) : "memory");
      asm volatile("ds_write_b32 %0, %1" : : "v"(st + 2 * TILE + (unsigned)(4 * prow)), "v"(nl) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  float16_t dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
  unsigned rab[4], tab[2][2];
  a64_row_bases(rab, lane);
  a64_tr_bases(tab, lane);
  const unsigned stat_a = (unsigned)(2 * TILE + 16 * h5);
  const bool tail_keys = wave < 3 && k0 + 32 > p.Lk;        // the wave's block reaches past Lk: those lanes' P must be zero
  const bool key_ok = k0 + kl < p.Lk;

  if (nt > 0) fetch(t_lo * 32, 0);
  if (nt > 1) fetch((t_lo + 1) * 32, STAGE);
  if (nt > 0) {
    publish(t_lo * 32, 0, nt > 1);
    __builtin_amdgcn_s_barrier();
  }
  const int nks = (p.Lk + 15) / 16;                          // 16-key steps of wave 3's product
  // wave 3: dQ of tile `tq` (first query q0) from the dS^T buffer `buf`
  auto dq_tile = [&](int q0, unsigned buf) {
    float16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    for (int kk = 0; kk < nks; ++kk) {
      const bf16x8_t dsf = tr_frag((const char*)smem + KIMG + NST * STAGE + buf, SMALL_DSROW, 16 * kk, 0, lane);
      short4_t lo0, hi0, lo1, hi1;
      const unsigned ko = smem_a + (unsigned)(16 * kk * 128);
      A64_RDTR(lo0, ko + tab[0][0], 0); A64_RDTR(hi0, ko + tab[1][0], 0); A64_RDTR(lo1, ko + tab[0][1], 0); A64_RDTR(hi1, ko + tab[1][1], 0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
      dq[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(lo0, hi0), dsf, dq[0], 0, 0, 0);
      dq[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(lo1, hi1), dsf, dq[1], 0, 0, 0);
    }
    const int q = q0 + kl;
    const bool live = q < p.Lq;
    a64_store_row(p.dQ + (long)b * p.bdq + (long)(live ? q : 0) * p.sdq + (long)hd * 64, live, dq[0], dq[1], p.scale, h5);
  };

  for (int t = 0; t < nt; ++t) {
    const unsigned so = (unsigned)((t % NST) * STAGE), sn = (unsigned)(((t + 1) % NST) * STAGE), s2 = (unsigned)(((t + 2) % NST) * STAGE);
    const unsigned dbuf = (unsigned)((t & 1) * DSBUF);
    const bool more = t + 1 < nt, more2 = t + 2 < nt;
    // (wave 3 fetches BEHIND its dQ stores, so that for every wave the youngest vector-memory operations at the end of the tile are the four of
    // this fetch; the stage of tile t + 2 is the one tile t - 1 used: free since the barrier that ended that tile)
    if (more2 && wave < 3) fetch((t_lo + t + 2) * 32, s2);
    if (wave < 3) {
      const unsigned qt = stage_a + so, dot = qt + TILE, st = qt + stat_a;
      float16_t s, dp;
      {
        float4_t c0, c1, c2, c3;
        bf16x8_t qr[4];
        A64_RD128(c0, st, 0); A64_RD128(c1, st, 32); A64_RD128(c2, st, 64); A64_RD128(c3, st, 96);
        A64_RD128(qr[0], qt + rab[0], 0); A64_RD128(qr[1], qt + rab[1], 0); A64_RD128(qr[2], qt + rab[2], 0); A64_RD128(qr[3], qt + rab[3], 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        s = __builtin_shufflevector(__builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(c2, c3, 0, 1, 2, 3, 4, 5, 6, 7),
                                    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[ks], kf[ks], s, 0, 0, 0);
        A64_PIN(s);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        float4_t e0, e1, e2, e3;
        bf16x8_t dr[4];
        A64_RD128(e0, st, 128); A64_RD128(e1, st, 160); A64_RD128(e2, st, 192); A64_RD128(e3, st, 224);
        A64_RD128(dr[0], dot + rab[0], 0); A64_RD128(dr[1], dot + rab[1], 0); A64_RD128(dr[2], dot + rab[2], 0); A64_RD128(dr[3], dot + rab[3], 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dp = __builtin_shufflevector(__builtin_shufflevector(e0, e1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(e2, e3, 0, 1, 2, 3, 4, 5, 6, 7),
                                     0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dr[ks], vf[ks], dp, 0, 0, 0);
        A64_PIN(dp);
        __builtin_amdgcn_sched_barrier(0);
      }
      short4_t olo[2][2], ohi[2][2];
      {
        const unsigned o00 = dot + tab[0][0], o01 = dot + tab[0][1], o10 = dot + tab[1][0], o11 = dot + tab[1][1];
        A64_RDTR(olo[0][0], o00, 0); A64_RDTR(olo[0][1], o01, 0); A64_RDTR(ohi[0][0], o10, 0); A64_RDTR(ohi[0][1], o11, 0);
        A64_RDTR(olo[1][0], o00, 2048); A64_RDTR(olo[1][1], o01, 2048); A64_RDTR(ohi[1][0], o10, 2048); A64_RDTR(ohi[1][1], o11, 2048);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = EXP2(s[r]);
        dp[r] *= s[r];
      }
      if (tail_keys) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = key_ok ? s[r] : 0.f;
          dp[r] = key_ok ? dp[r] : 0.f;
        }
      }
      bf16x8_t pf[2], dsf[2];
      pf[0] = pack_frag(s, 0); pf[1] = pack_frag(s, 1);
      dsf[0] = pack_frag(dp, 0); dsf[1] = pack_frag(dp, 1);
      A64_PIN(pf[0]); A64_PIN(pf[1]); A64_PIN(dsf[0]); A64_PIN(dsf[1]);
      // dS^T [key][32 queries] for wave 3: register pair j of the lane = queries 8 j + 4 h5 .. + 3 of its key
      {
        const unsigned da = ds_a + dbuf + (unsigned)((k0 + kl) * SMALL_DSROW + 8 * h5);
        const uint4_t d0 = __builtin_bit_cast(uint4_t, dsf[0]), d1 = __builtin_bit_cast(uint4_t, dsf[1]);
        const uint2_t w0 = {d0.x, d0.y}, w1 = {d0.z, d0.w}, w2 = {d1.x, d1.y}, w3 = {d1.z, d1.w};
        A64_WR64(da, w0, 0); A64_WR64(da, w1, 16); A64_WR64(da, w2, 32); A64_WR64(da, w3, 48);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      short4_t qlo[2][2], qhi[2][2];
      {
        const unsigned q00 = qt + tab[0][0], q01 = qt + tab[0][1], q10 = qt + tab[1][0], q11 = qt + tab[1][1];
        A64_RDTR(qlo[0][0], q00, 0); A64_RDTR(qlo[0][1], q01, 0); A64_RDTR(qhi[0][0], q10, 0); A64_RDTR(qhi[0][1], q11, 0);
        A64_RDTR(qlo[1][0], q00, 2048); A64_RDTR(qlo[1][1], q01, 2048); A64_RDTR(qhi[1][0], q10, 2048); A64_RDTR(qhi[1][1], q11, 2048);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(olo[s2][dt], ohi[s2][dt]), pf[s2], dv[dt], 0, 0, 0);
      A64_PIN(dv[0]); A64_PIN(dv[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_join(qlo[s2][dt], qhi[s2][dt]), dsf[s2], dk[dt], 0, 0, 0);
      A64_PIN(dk[0]); A64_PIN(dk[1]);
    } else {
      if (t > 0) dq_tile((t_lo + t - 1) * 32, (unsigned)(((t - 1) & 1) * DSBUF));
      if (more2) fetch((t_lo + t + 2) * 32, s2);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) publish((t_lo + t + 1) * 32, sn, more2);
    __builtin_amdgcn_s_barrier();           // dS^T of tile t written; this wave's rows of tile t + 1 published: Q' / -lse2 / -delta / dO in place
  }
  if (wave == 3 && nt > 0) dq_tile((t_lo + nt - 1) * 32, (unsigned)(((nt - 1) & 1) * DSBUF));

  const float ln2 = 0.6931471805599453f;
  const int key = k0 + kl;
  const bool live = wave < 3 && key < p.Lk;
  if (live) {
    if (p.qsplit > 1) {
      const long hd_all = (long)p.H * 64;
      const long plane = (long)p.B * p.Lk * hd_all;
      float* pk = p.dkv_part + ((long)qs * 2) * plane + ((long)b * p.Lk + key) * hd_all + (long)hd * 64;
      float* pv = pk + plane;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int d = dt * 32 + 8 * r4 + 4 * h5;
          *(float4_t*)(pk + d) = (float4_t){dk[dt][4 * r4 + 0] * ln2, dk[dt][4 * r4 + 1] * ln2, dk[dt][4 * r4 + 2] * ln2, dk[dt][4 * r4 + 3] * ln2};
          *(float4_t*)(pv + d) = (float4_t){dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1], dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]};
        }
    }
  }
  if (p.qsplit <= 1) {      // (wave-uniform; every lane takes part in the half-wave swaps of the 16-byte stores)
    const long krow = live ? key : 0;
    a64_store_row(p.dK + (long)b * p.bdk + krow * p.sdk + (long)hd * 64, live, dk[0], dk[1], ln2, h5);
    a64_store_row(p.dV + (long)b * p.bdv + krow * p.sdv + (long)hd * 64, live, dv[0], dv[1], 1.0f, h5);
  }
}

// ================================================================================================
// backward: dK, dV.  Workgroup = 128 keys (4 waves x 32 keys on lanes), loops over 32-query tiles.
// ================================================================================================
template <int DP, int NW = 4>   // NW waves x 32 rows per workgroup (2: twice the workgroups for short sequences)
__global__ __launch_bounds__(NW * 64, 3) void attn_bwd_dkdv_kernel(const AttnParams p) {
  constexpr int RS = DP * 2 + 16, KS = DP / 16, DT = DP / 32, TILE = 32 * RS;
  constexpr int STAGE = 2 * TILE + 256;  // Q tile, dO tile, lse2[32], delta[32]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, kl = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int kb = bx / p.qsplit, qs = bx - kb * p.qsplit;
  const int k0 = kb * (NW * 32) + wave * 32;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * p.D;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * p.D;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * p.D;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * p.D;
  const float* lse = p.LSE + ((long)b * p.H + hd) * p.Lq;
  const float* dl = p.delta + ((long)b * p.H + hd) * p.Lq;
  // this workgroup's slice of the query tiles
  const int nt_all = (p.Lq + 31) / 32;
  const int per = (nt_all + p.qsplit - 1) / p.qsplit;
  const int t_lo = qs * per, t_hi = min(nt_all, t_lo + per);

  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4_t zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
    int d0 = 16 * ks + 8 * h5;
    if (k0 + kl < p.Lk && d0 < p.D) {
      zk = *(const uint4_t*)(Kb + (long)(k0 + kl) * p.sk + d0);
      zv = *(const uint4_t*)(Vb + (long)(k0 + kl) * p.sv + d0);
    }
    kf[ks] = __builtin_bit_cast(bf16x8_t, zk);
    vf[ks] = __builtin_bit_cast(bf16x8_t, zv);
  }
  float16_t dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  const int nt = max(t_hi - t_lo, 0);
  TileLoader<32, DP, NW * 64> lq, ld_;
  const int qoff = TileLoader<32, DP, NW * 64>::piece_offset(p.sq, tid), dooff = TileLoader<32, DP, NW * 64>::piece_offset(p.sdo, tid);
  float st_lse = 0.f, st_dl = 0.f;
  auto load_stats = [&](int q0) {
    if (tid < 32) {
      int q = q0 + tid;
      st_lse = q < p.Lq ? lse[q] * LOG2E : 1.0e30f;  // invalid query rows -> P = 0
      st_dl = q < p.Lq ? dl[q] : 0.f;
    }
  };
  auto store_stats = [&](char* stage) {
    if (tid < 32) {
      ((float*)(stage + 2 * TILE))[tid] = st_lse;
      ((float*)(stage + 2 * TILE))[32 + tid] = st_dl;
    }
  };
  lq.load(Qb, p.sq, t_lo * 32, p.Lq, p.D, tid);
  ld_.load(dOb, p.sdo, t_lo * 32, p.Lq, p.D, tid);
  load_stats(t_lo * 32);
  lq.store(smem, tid);
  ld_.store(smem + TILE, tid);
  store_stats(smem);
  __syncthreads();

  // (PF_WHOLE: the query tile prefetched in this iteration lies wholly inside Q / dO -- compile-time, as in the forward kernel)
  auto iteration = [&](int t, auto pf_whole_tag) {
    constexpr bool PF_WHOLE = decltype(pf_whole_tag)::value;
    const char* st = smem + (t & 1) * STAGE;
    const char* qt = st;
    const char* dot = st + TILE;
    const float* s_lse = (const float*)(st + 2 * TILE);
    const float* s_dl = s_lse + 32;
    const bool more = PF_WHOLE || t + 1 < nt;
    if constexpr (PF_WHOLE) {
      const int qn = (t_lo + t + 1) * 32;
      lq.load_full(Qb + (long)qn * p.sq, p.sq, qoff);
      ld_.load_full(dOb + (long)qn * p.sdo, p.sdo, dooff);
      if (tid < 32) { st_lse = lse[qn + tid] * LOG2E; st_dl = dl[qn + tid]; }
    } else if (more) {
      lq.load(Qb, p.sq, (t_lo + t + 1) * 32, p.Lq, p.D, tid);
      ld_.load(dOb, p.sdo, (t_lo + t + 1) * 32, p.Lq, p.D, tid);
      load_stats((t_lo + t + 1) * 32);
    }
    float16_t s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qt, RS, 0, ks, lane), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dot, RS, 0, ks, lane), vf[ks], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int qr = acc_row(r, h5);
      float pv = EXP2(s[r] * c - s_lse[qr]);
      s[r] = pv;
      dp[r] = pv * (dp[r] - s_dl[qr]);          // (the softmax scale multiplies dK once, after the loop)
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8_t pf = pack_frag(s, s2);
      const bf16x8_t dsf = pack_frag(dp, s2);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dot, RS, 16 * s2, dt * 32, lane), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qt, RS, 16 * s2, dt * 32, lane), dsf, dk[dt], 0, 0, 0);
      }
    }
    if (more) {
      char* nx = smem + ((t + 1) & 1) * STAGE;
      lq.store(nx, tid);
      ld_.store(nx + TILE, tid);
      store_stats(nx);
    }
    __syncthreads();
  };
  {
    // iterations whose PREFETCHED tile (t_lo + t + 1) lies wholly inside Q / dO: (t_lo + t + 2) * 32 <= Lq
    int t = 0;
    if constexpr (DP == 64) {
      const int nw = p.D == DP ? min(nt - 1, p.Lq / 32 - t_lo - 1) : 0;
      for (; t < nw; ++t) iteration(t, std::true_type{});
    }
    for (; t < nt; ++t) iteration(t, std::false_type{});
  }

#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] *= p.scale;
  const int key = k0 + kl;
  if (key < p.Lk) {
    if (p.qsplit > 1) {
      // fp32 partials [qs][0=dK,1=dV][b][key][H*D]; summed in a fixed order by attn_dkv_reduce_kernel
      const long hd_all = (long)p.H * p.D;
      const long plane = (long)p.B * p.Lk * hd_all;
      float* pk = p.dkv_part + ((long)qs * 2) * plane + ((long)b * p.Lk + key) * hd_all + (long)hd * p.D;
      float* pv = pk + plane;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          int d = dt * 32 + 8 * r4 + 4 * h5;
          if (d < p.D) {
            *(float4_t*)(pk + d) = (float4_t){dk[dt][4 * r4 + 0], dk[dt][4 * r4 + 1], dk[dt][4 * r4 + 2], dk[dt][4 * r4 + 3]};
            *(float4_t*)(pv + d) = (float4_t){dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1], dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]};
          }
        }
      return;
    }
    bf16_t* dKb = p.dK + (long)b * p.bdk + (long)key * p.sdk + (long)hd * p.D;
    bf16_t* dVb = p.dV + (long)b * p.bdv + (long)key * p.sdv + (long)hd * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        int d = dt * 32 + 8 * r4 + 4 * h5;
        if (d < p.D) {
          uint2_t a, v;
          a.x = pack2bf(dk[dt][4 * r4 + 0], dk[dt][4 * r4 + 1]);
          a.y = pack2bf(dk[dt][4 * r4 + 2], dk[dt][4 * r4 + 3]);
          v.x = pack2bf(dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1]);
          v.y = pack2bf(dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]);
          *(uint2_t*)(dKb + d) = a;
          *(uint2_t*)(dVb + d) = v;
        }
      }
  }
}

// dK / dV = sum over the query splits of the fp32 partials (fixed order), written bf16 with the caller's strides
__global__ void attn_dkv_reduce_kernel(const AttnParams p) {
  const long hd_all = (long)p.H * p.D;
  const long rows = (long)p.B * p.Lk;
  const long plane = rows * hd_all;
  const long total = rows * (hd_all >> 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / (hd_all >> 2);
    int c4 = (int)(i - row * (hd_all >> 2)) * 4;
    int b = (int)(row / p.Lk), key = (int)(row - (long)b * p.Lk);
    float4_t ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.qsplit; ++s) {
      const float* base = p.dkv_part + ((long)s * 2) * plane + row * hd_all + c4;
      const float4_t x = *(const float4_t*)base, y = *(const float4_t*)(base + plane);
      ak += x;
      av += y;
    }
    uint2_t ok, ov;
    ok.x = pack2bf(ak[0], ak[1]); ok.y = pack2bf(ak[2], ak[3]);
    ov.x = pack2bf(av[0], av[1]); ov.y = pack2bf(av[2], av[3]);
    *(uint2_t*)(p.dK + (long)b * p.bdk + (long)key * p.sdk + c4) = ok;
    *(uint2_t*)(p.dV + (long)b * p.bdv + (long)key * p.sdv + c4) = ov;
  }
}

// ================================================================================================
// backward: dQ.  Workgroup = 128 queries (on lanes), loops over 64-key tiles.
// ================================================================================================
template <int DP, int NW = 4>   // NW waves x 32 rows per workgroup (2: twice the workgroups for short sequences)
__global__ __launch_bounds__(NW * 64, 3) void attn_bwd_dq_kernel(const AttnParams p) {
  constexpr int RS = DP * 2 + 16, KS = DP / 16, DT = DP / 32, TILE = 64 * RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, ql = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int q0 = bx * (NW * 32) + wave * 32;
  const int q = q0 + ql;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * p.D;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * p.D;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * p.D;
  const bf16_t* dOb = p.dO + (long)b * p.bdo + (long)hd * p.D;

  // delta[q] = sum_d dO[q][d] * O[q][d] is computed HERE, from the dO fragments this lane holds anyway and the matching 16 bytes of
  // O (the lane and its partner lane ^ 32 cover the row between them), and written out for the dK / dV kernel, which is launched
  // after this one: the separate delta kernel (140 launches and 1.4 ms of the step's main stream) is gone.
  const bf16_t* Ocb = p.Oc + (long)b * p.bo + (long)hd * p.D;
  bf16x8_t qf[KS], dof[KS];
  float dacc = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4_t zq = {0u, 0u, 0u, 0u}, zd = {0u, 0u, 0u, 0u}, zo = {0u, 0u, 0u, 0u};
    int d0 = 16 * ks + 8 * h5;
    if (q < p.Lq && d0 < p.D) {
      zq = *(const uint4_t*)(Qb + (long)q * p.sq + d0);
      zd = *(const uint4_t*)(dOb + (long)q * p.sdo + d0);
      zo = *(const uint4_t*)(Ocb + (long)q * p.so + d0);
    }
    qf[ks] = __builtin_bit_cast(bf16x8_t, zq);
    dof[ks] = __builtin_bit_cast(bf16x8_t, zd);
    float fo[8], fd[8];
    unpack8(zo, fo);
    unpack8(zd, fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) dacc += fo[e] * fd[e];
  }
  const float dlt = dacc + __shfl_xor(dacc, 32, 64);
  if (q < p.Lq && h5 == 0) p.delta[((long)b * p.H + hd) * p.Lq + q] = dlt;
  const float lse2 = q < p.Lq ? p.LSE[((long)b * p.H + hd) * p.Lq + q] * LOG2E : 1.0e30f;

  float16_t dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  const int nt = (p.Lk + 63) / 64;
  TileLoader<64, DP, NW * 64> lk, lv;
  const int koff = TileLoader<64, DP, NW * 64>::piece_offset(p.sk, tid);
  lk.load(Kb, p.sk, 0, p.Lk, p.D, tid);
  lv.load(Vb, p.sv, 0, p.Lk, p.D, tid);
  lk.store(smem, tid);
  lv.store(smem + TILE, tid);
  __syncthreads();

  // (MASKED / PF_WHOLE: compile-time flags as in the forward kernel -- the key mask of the last tile and the general loader's arithmetic
  // stay out of the common path)
  auto iteration = [&](int t, auto masked_tag, auto pf_whole_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value, PF_WHOLE = decltype(pf_whole_tag)::value;
    const char* kt = smem + (t & 1) * 2 * TILE;
    const char* vt = kt + TILE;
    const bool more = PF_WHOLE || t + 1 < nt;
    if constexpr (PF_WHOLE) {
      lk.load_full(Kb + (long)(t + 1) * 64 * p.sk, p.sk, koff);
      lv.load_full(Vb + (long)(t + 1) * 64 * p.sv, p.sv, koff);
    } else if (more) {
      lk.load(Kb, p.sk, (t + 1) * 64, p.Lk, p.D, tid);
      lv.load(Vb, p.sv, (t + 1) * 64, p.Lk, p.D, tid);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, RS, hf * 32, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vt, RS, hf * 32, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = EXP2(s[r] * c - lse2);
        dp[r] = pv * (dp[r] - dlt);             // (the softmax scale multiplies dQ once, after the loop)
      }
      if constexpr (MASKED) {   // keys beyond Lk exist only in the last tile
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = (t * 64 + hf * 32 + acc_row(r, h5)) < p.Lk ? dp[r] : 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t dsf = pack_frag(dp, s2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(kt, RS, hf * 32 + 16 * s2, dt * 32, lane), dsf,
                                                           dq[dt], 0, 0, 0);
      }
    }
    if (more) {
      char* nx = smem + ((t + 1) & 1) * 2 * TILE;
      lk.store(nx, tid);
      lv.store(nx + TILE, tid);
    }
    __syncthreads();
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const int nwhole = (p.D != DP || DP != 64 || p.sk != p.sv) ? 0 : p.Lk / 64;
    int t = 0;
    if constexpr (DP == 64) {
      for (; t + 1 < nwhole; ++t) iteration(t, F{}, T{});
    }
    for (; t < nwhole; ++t) iteration(t, F{}, F{});
    for (; t < nt; ++t) iteration(t, T{}, F{});
  }

#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] *= p.scale;
  if (q < p.Lq) {
    bf16_t* dQb = p.dQ + (long)b * p.bdq + (long)q * p.sdq + (long)hd * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        int d = dt * 32 + 8 * r4 + 4 * h5;
        if (d < p.D) {
          uint2_t a;
          a.x = pack2bf(dq[dt][4 * r4 + 0], dq[dt][4 * r4 + 1]);
          a.y = pack2bf(dq[dt][4 * r4 + 2], dq[dt][4 * r4 + 3]);
          *(uint2_t*)(dQb + d) = a;
        }
      }
  }
}

// ================================================================================================
// host
// ================================================================================================
#include "attn512.h"
#include "attn512_bwd.h"

static int attn_check(const NkAttnDesc* d) {
  NK_CHECK_ARG(d != nullptr);
  NK_CHECK_ARG(d->B > 0 && d->H > 0 && d->Lq > 0 && d->Lk > 0 && d->D > 0);
  NK_CHECK_ARG((d->D & 7) == 0 && (d->D <= 160 || d->D == 512));     // 512: attn512.h (forward), attn512_bwd.h (backward)
  NK_CHECK_ARG((d->sq & 7) == 0 && (d->sk & 7) == 0 && (d->sv & 7) == 0 && (d->so & 7) == 0);
  NK_CHECK_ARG((d->bq & 7) == 0 && (d->bk & 7) == 0 && (d->bv & 7) == 0 && (d->bo & 7) == 0);
  NK_CHECK_ARG(d->B <= 65535 && d->H <= 65535);
  return NK_OK;
}
static int attn_dp(int D) { return D <= 64 ? 64 : (D <= 96 ? 96 : 160); }
// waves per workgroup: 4 x 32 rows.  (A 2-wave variant -- twice the workgroups for SDXL's L = 1024 layers, which give only 640 -- was
// measured SLOWER: forward 72 vs 65 us, backward 200 vs 181 us, twice the K/V tile loads per query row and half the waves sharing a tile.)
static constexpr int ATTN_NW = 4;

// NK_ATTN64=0: the generic kernels for head dim 64 as well (A/B switch of round 4; read per call)
static bool attn64_enabled() {
  const char* e = getenv("NK_ATTN64");
  return !e || atoi(e) != 0;
}
template <typename K>
static void set_smem(K kern, int bytes) {
  nk_optin_lds((const void*)kern, bytes);
}

extern "C" int nk_attention_fwd(const NkAttnDesc* d, const void* q, const void* k, const void* v, void* o, float* lse,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = attn_check(d)) return e;
  NK_CHECK_ARG(q && k && v && o && (lse || d->D == 512));
  AttnParams p = {};
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o; p.LSE = lse;
  p.B = d->B; p.H = d->H; p.Lq = d->Lq; p.Lk = d->Lk; p.D = d->D;
  p.sq = d->sq; p.sk = d->sk; p.sv = d->sv; p.so = d->so;
  p.bq = d->bq; p.bk = d->bk; p.bv = d->bv; p.bo = d->bo;
  p.scale = d->scale;
  p.causal = d->causal != 0;
  NK_CHECK_ARG(!d->causal || d->Lq == d->Lk);
  constexpr int nw = ATTN_NW;
  dim3 grid((d->Lq + nw * 32 - 1) / (nw * 32), d->H, d->B);
  const dim3 lgrid = attn_grid(p, grid);       // 1-D, XCD-aware (attn_wg)
  if (d->D == 512 || (d->D == 64 && attn64_enabled())) NK_CHECK_ARG(((uintptr_t)o & 15) == 0);      // 16-byte output stores
  if (d->D == 512) {
    // the VAE mid block's single head: one workgroup per CU, 512 registers per lane, all 160 KiB of LDS (attn512.h); lse may be null
    NK_CHECK_ARG(!d->causal);
    const int smem512 = 5 * 32 * 1024;
    set_smem(attn512_fwd_kernel, smem512);
    hipLaunchKernelGGL(attn512_fwd_kernel, lgrid, dim3(256), smem512, stream, p);
    return nk_check_launch("attn512_fwd_kernel");
  }
  const int dp = attn_dp(d->D);
  const int smem = 2 * 2 * 64 * (dp * 2 + 16);
#define FWD_CASE(DP_)                                                                          \
  if (dp == DP_) {                                                                             \
    set_smem(attn_fwd_kernel<DP_, ATTN_NW>, smem);                                                \
    hipLaunchKernelGGL((attn_fwd_kernel<DP_, ATTN_NW>), lgrid, dim3(ATTN_NW * 64), smem, stream, p); \
  }
  if (d->D == 64 && attn64_enabled()) {
    const int smem64 = 3 * 2 * 64 * 128;
    set_smem(attn64_fwd_kernel<ATTN_NW>, smem64);
    hipLaunchKernelGGL((attn64_fwd_kernel<ATTN_NW>), lgrid, dim3(ATTN_NW * 64), smem64, stream, p);
    return nk_check_launch("attn64_fwd_kernel");
  }
  FWD_CASE(64) FWD_CASE(96) FWD_CASE(160)
#undef FWD_CASE
  return nk_check_launch("attn_fwd_kernel");
}

// NK_ATTN64_SMALL=0: cross-attention backward through the two-kernel path (A/B switch of round 4; read per call)
static bool attn64_small_enabled() {
  const char* e = getenv("NK_ATTN64_SMALL");
  return !e || atoi(e) != 0;
}
// query splits of the one-kernel backward: enough workgroups for the chip (two per CU fit), at least two 32-query tiles each
static int attn_small_qsplit(const NkAttnDesc* d) {
  const int base = d->B * d->H;
  int s = 1;
  while (s < 64 && base * s * 2 <= 512 && d->Lq / (s * 2) >= 64) s *= 2;      // (at most one round of two workgroups per CU)
  return s;
}
static int attn_qsplit(const NkAttnDesc* d) {
  // a single key block (cross-attention, Lk = 77) gives only B*H workgroups that each walk the whole query range:
  // split the query range so the grid has >= ~512 workgroups
  if (d->Lk > 128 || d->Lq < 512) return 1;
  int base = d->B * d->H;
  int s = 1;
  while (s < 16 && base * s < 512 && d->Lq / (s * 2) >= 128) s *= 2;
  return s;
}
extern "C" long nk_attention_bwd_ws_floats(const NkAttnDesc* d) {
  int s = attn_qsplit(d);
  if (d->D == 64 && d->Lk <= 96 && attn_small_qsplit(d) > s) s = attn_small_qsplit(d);
  long part = s > 1 ? (long)s * 2 * d->B * d->Lk * d->H * d->D : 0;
  if (d->D == 64) return Attn64Ws::make(d->B, d->H, d->Lq).part + part + 64;     // -delta, -lse2, Q' (whichever kernels run)
  long delta = (long)d->B * d->H * d->Lq;
  return delta + part + 64;
}

extern "C" int nk_attention_bwd(const NkAttnDesc* d, const void* q, const void* k, const void* v, const void* o,
                                const float* lse, const void* d_o, void* dq, void* dk, void* dv, float* delta_ws,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = attn_check(d)) return e;
  NK_CHECK_ARG(q && k && v && o && lse && d_o && dq && dk && dv && delta_ws);
  NK_CHECK_ARG(!d->causal);   // the causal variant serves the frozen text encoders: forward only
  NK_CHECK_ARG((d->sdq & 7) == 0 && (d->sdk & 7) == 0 && (d->sdv & 7) == 0 && (d->sdo & 7) == 0);
  NK_CHECK_ARG((d->bdq & 7) == 0 && (d->bdk & 7) == 0 && (d->bdv & 7) == 0 && (d->bdo & 7) == 0);
  AttnParams p = {};
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.Oc = (const bf16_t*)o;
  p.dO = (const bf16_t*)d_o; p.dQ = (bf16_t*)dq; p.dK = (bf16_t*)dk; p.dV = (bf16_t*)dv;
  p.LSE = (float*)lse; p.delta = delta_ws;
  p.B = d->B; p.H = d->H; p.Lq = d->Lq; p.Lk = d->Lk; p.D = d->D;
  p.sq = d->sq; p.sk = d->sk; p.sv = d->sv; p.so = d->so;
  p.bq = d->bq; p.bk = d->bk; p.bv = d->bv; p.bo = d->bo;
  p.sdq = d->sdq; p.sdk = d->sdk; p.sdv = d->sdv; p.sdo = d->sdo;
  p.bdq = d->bdq; p.bdk = d->bdk; p.bdv = d->bdv; p.bdo = d->bdo;
  p.scale = d->scale;
  if (d->D == 512) {
    // head dim 512 (the VAE mid block under autoencoder training): delta = rowsum(dO o O), then the same kernel template twice --
    // dQ per 32-query block, dK / dV per 32-key block -- recomputing the scores tile by tile from the forward's log-sum-exp (attn512_bwd.h)
    NK_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0);
    {
      const long rows = (long)d->B * d->H * d->Lq;
      hipLaunchKernelGGL(attn512_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
      if (int e = nk_check_launch("attn512_delta_kernel")) return e;
    }
    set_smem(attn512_bwd_kernel<0>, A5B_SMEM);
    set_smem(attn512_bwd_kernel<1>, A5B_SMEM);
    hipLaunchKernelGGL(attn512_bwd_kernel<0>, dim3((d->Lq + A5B_ROWS - 1) / A5B_ROWS, d->H, d->B), dim3(256), A5B_SMEM, stream, p);
    if (int e = nk_check_launch("attn512_bwd_kernel<0>")) return e;
    hipLaunchKernelGGL(attn512_bwd_kernel<1>, dim3((d->Lk + A5B_ROWS - 1) / A5B_ROWS, d->H, d->B), dim3(256), A5B_SMEM, stream, p);
    return nk_check_launch("attn512_bwd_kernel<1>");
  }
  NK_CHECK_ARG(d->D <= 160);
  if (d->D == 64 && attn64_enabled())      // 16-byte gradient stores
    NK_CHECK_ARG(((uintptr_t)dq & 15) == 0 && ((uintptr_t)dk & 15) == 0 && ((uintptr_t)dv & 15) == 0);
  if (d->D == 64 && d->Lk <= 96 && attn64_enabled() && attn64_small_enabled()) {
    // head dim 64, at most 96 keys (cross-attention): everything in one kernel (+ the sum of the query splits' dK / dV partials)
    NK_CHECK_ARG(((uintptr_t)delta_ws & 15) == 0);
    const Attn64Ws w = Attn64Ws::make(d->B, d->H, d->Lq);
    p.qsplit = attn_small_qsplit(d);
    p.dkv_part = p.qsplit > 1 ? delta_ws + w.part : nullptr;
    const int smem = 96 * 128 + 3 * (3 * 32 * 128 + 256 + 1024) + 2 * 96 * SMALL_DSROW;
    set_smem(attn64_bwd_small_kernel, smem);
    const dim3 lgrid = attn_grid(p, dim3(p.qsplit, d->H, d->B));
    hipLaunchKernelGGL(attn64_bwd_small_kernel, lgrid, dim3(256), smem, stream, p);
    if (int e = nk_check_launch("attn64_bwd_small_kernel")) return e;
    if (p.qsplit > 1) {
      long total = (long)d->B * d->Lk * ((long)d->H * d->D / 4);
      long blocks = (total + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
      if (int e = nk_check_launch("attn_dkv_reduce_kernel")) return e;
    }
    return NK_OK;
  }
  if (d->D == 64 && attn64_enabled()) {
    // head dim 64: the dQ kernel (which also writes -delta, -lse2 and Q' into the workspace), then dK / dV
    NK_CHECK_ARG(((uintptr_t)delta_ws & 15) == 0);
    const Attn64Ws w = Attn64Ws::make(d->B, d->H, d->Lq);
    constexpr int nw = ATTN_NW;
    {
      dim3 grid((d->Lq + nw * 32 - 1) / (nw * 32), d->H, d->B);
      const dim3 lgrid = attn_grid(p, grid);       // 1-D, XCD-aware (attn_wg)
      const int smem = 3 * 2 * 64 * 128;
      set_smem(attn64_bwd_dq_kernel<ATTN_NW>, smem);
      hipLaunchKernelGGL((attn64_bwd_dq_kernel<ATTN_NW>), lgrid, dim3(nw * 64), smem, stream, p);
      if (int e = nk_check_launch("attn64_bwd_dq_kernel")) return e;
    }
    p.qsplit = attn_qsplit(d);
    p.dkv_part = p.qsplit > 1 ? delta_ws + w.part : nullptr;
    {
      dim3 grid(((d->Lk + nw * 32 - 1) / (nw * 32)) * p.qsplit, d->H, d->B);
      const dim3 lgrid = attn_grid(p, grid);       // 1-D, XCD-aware (attn_wg)
      const int smem = 3 * (2 * 32 * 128 + 256);
      set_smem(attn64_bwd_dkdv_kernel<ATTN_NW>, smem);
      hipLaunchKernelGGL((attn64_bwd_dkdv_kernel<ATTN_NW>), lgrid, dim3(nw * 64), smem, stream, p);
      if (int e = nk_check_launch("attn64_bwd_dkdv_kernel")) return e;
    }
    if (p.qsplit > 1) {
      long total = (long)d->B * d->Lk * ((long)d->H * d->D / 4);
      long blocks = (total + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
      if (int e = nk_check_launch("attn_dkv_reduce_kernel")) return e;
    }
    return NK_OK;
  }
  // order: dQ kernel first (it also produces delta = rowsum(dO * O) for the dK / dV kernel), then dK / dV
  const int dp = attn_dp(d->D);
  {
    constexpr int nw = ATTN_NW;
    dim3 grid((d->Lq + nw * 32 - 1) / (nw * 32), d->H, d->B);
    const dim3 lgrid = attn_grid(p, grid);       // 1-D, XCD-aware (attn_wg)
    const int smem = 2 * 2 * 64 * (dp * 2 + 16);
#define Q_CASE(DP_)                                                                          \
  if (dp == DP_) {                                                                             \
    set_smem(attn_bwd_dq_kernel<DP_, ATTN_NW>, smem);                                                \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DP_, ATTN_NW>), lgrid, dim3(ATTN_NW * 64), smem, stream, p); \
  }
    Q_CASE(64) Q_CASE(96) Q_CASE(160)
#undef Q_CASE
  }
  if (int e = nk_check_launch("attn_bwd_dq_kernel")) return e;
  p.qsplit = attn_qsplit(d);
  p.dkv_part = p.qsplit > 1 ? delta_ws + (((long)d->B * d->H * d->Lq + 3) & ~3l) : nullptr;
  {
    constexpr int nw = ATTN_NW;
    dim3 grid(((d->Lk + nw * 32 - 1) / (nw * 32)) * p.qsplit, d->H, d->B);
    const dim3 lgrid = attn_grid(p, grid);       // 1-D, XCD-aware (attn_wg)
    const int smem = 2 * (2 * 32 * (dp * 2 + 16) + 256);
#define KV_CASE(DP_)                                                                          \
  if (dp == DP_) {                                                                             \
    set_smem(attn_bwd_dkdv_kernel<DP_, ATTN_NW>, smem);                                                \
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<DP_, ATTN_NW>), lgrid, dim3(ATTN_NW * 64), smem, stream, p); \
  }
    KV_CASE(64) KV_CASE(96) KV_CASE(160)
#undef KV_CASE
    if (int e = nk_check_launch("attn_bwd_dkdv_kernel")) return e;
    if (p.qsplit > 1) {
      long total = (long)d->B * d->Lk * ((long)d->H * d->D / 4);
      long blocks = (total + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
      if (int e = nk_check_launch("attn_dkv_reduce_kernel")) return e;
    }
  }
  return NK_OK;
}
