// Head dim 512: the single-head attention of the VAE mid block (modules/diffusion/model.py:224-243, AttnBlock / TorchSDPAttnBlock:
// softmax(q k^T / sqrt(512)) v over H*W tokens), as ONE flash kernel -- no [L][L] score matrix in HBM.  Included by attention.hip.
//
// Why this shape.  At d = 512 the operands are what is large: a query row is 1 KiB, the O accumulator of 32 queries is 64 KiB of fp32.
// The only place both fit is the register file of a wave that owns its SIMD: 4 waves per workgroup, ONE workgroup per CU,
// launch_bounds(256, 1) = 512 registers per lane:
//   Q'^T fragments (32 queries x 512, bf16, pre-multiplied by scale * log2 e)   128 VGPRs
//   O^T accumulator (512 x 32 queries, fp32)                                    256 AGPRs
//   a score block, two batches of K / V fragments, addresses                    ~100 VGPRs
// K and V tiles of 32 keys (32 KiB each) arrive by LDS-DMA: K two tiles ahead into three stages, V one ahead into two (all 160 KiB).
// Per 32-key tile and wave: 32 + 32 v_mfma_f32_32x32x16_bf16 (2 048 matrix-pipe cycles) against 16 exponentials per lane, so unlike
// head dim 64 the loop is matrix-bound; what has to be hidden is LDS latency with one wave per SIMD, hence the explicit batches
// (eight fragments requested while the previous eight are multiplied).  LDS traffic: every wave reads the whole K and V tile,
// 256 KiB per tile and CU = 1 024 of the 2 048 cycles at 256 B / clk; the fill path carries 64 KiB per tile = 1 024 cycles at 64 B / clk.
//
// LDS image of a [32 keys][512] bf16 tile (1 KiB rows): the 16-byte chunk c of row r lives in slot  (c & 48) | ((c & 15) ^ g(r)),
// g(r) = (r & 3) << 2 | (r >> 2) & 3:
//  * row reads (ds_read_b128, lane & 31 = row, one chunk; a pass = 16 lanes = rows {0-3,12-15,20-27} or {4-11,16-19,28-31}): the 16 rows
//    of a pass have 16 different g -> 16 different 16-byte windows of the 256-byte bank line: conflict-free;
//  * transposed reads (ds_read_b64_tr_b16; a 32-lane half = 4 consecutive rows x 64 bytes): bits 2-3 of g are the row's bits 0-1, so the
//    four rows take the four 64-byte windows: conflict-free.
// One DMA wave instruction deposits 1 KiB = one row linearly; the swizzle is applied on the SOURCE side (lane l fetches chunk
// (l & 48) | ((l & 15) ^ g(r))).  Rows past Lk are fetched from row Lk - 1 and their scores masked.
#pragma once

__device__ __forceinline__ int a512_g(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

#define A512_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define A512_RDTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define A512_WAIT4(x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
#define A512_WAIT4_KEEP4(x) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))

struct A512V {          // one batch of V^T fragments: four d-tiles x (low, high) 8 key rows
  short4_t lo[4], hi[4];
};
#define A512_WAITV(x)                                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.lo[0]), "+v"(x.lo[1]), "+v"(x.lo[2]), "+v"(x.lo[3]), "+v"(x.hi[0]), "+v"(x.hi[1]), \
               "+v"(x.hi[2]), "+v"(x.hi[3]))
#define A512_WAITV_KEEP8(x)                                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x.lo[0]), "+v"(x.lo[1]), "+v"(x.lo[2]), "+v"(x.lo[3]), "+v"(x.hi[0]), "+v"(x.hi[1]), \
               "+v"(x.hi[2]), "+v"(x.hi[3]))

// ---- the O^T accumulator: the whole AGPR file, a[16 n : 16 n + 15] = d-tile n, addressed PHYSICALLY from inline asm.
// Left to the register allocator (MFMA builtins, or inline-asm MFMAs on "+a" operands) a kernel with 256 accumulator + 128 operand registers
// copied accumulator blocks between the two register files at the loop head and spilled 340-370 registers per lane.  Here the compiler never
// sees the accumulator: it allocates no AGPR of its own (checked in the ISA: tools/check_resources.sh lists 0 bytes of scratch and the only
// v_accvgpr_* / a[...] instructions are the ones below), the "a255" clobber makes it reserve the file.  What the hazard recogniser therefore
// cannot see is covered by construction: an accumulator block is read as SrcC >= 16 MFMAs after it was written, v_accvgpr_read / _write run
// only after a score chain's result has been consumed by vector code (all earlier MFMAs have retired) or behind explicit s_nops.
// every AGPR, as a clobber list: the compiler may keep nothing of its own in the accumulator file across the statements that carry it
#define A512_ALL_AGPRS \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define A512_R16(OP_, IDX_)                                                                                                      \
  OP_(IDX_, 0) OP_(IDX_, 1) OP_(IDX_, 2) OP_(IDX_, 3) OP_(IDX_, 4) OP_(IDX_, 5) OP_(IDX_, 6) OP_(IDX_, 7)                        \
  OP_(IDX_, 8) OP_(IDX_, 9) OP_(IDX_, 10) OP_(IDX_, 11) OP_(IDX_, 12) OP_(IDX_, 13) OP_(IDX_, 14) OP_(IDX_, 15)
template <int N>
__device__ __forceinline__ void a512_acc_zero() {
#define A512_Z(IDX_, I_) "v_accvgpr_write_b32 a[%0*16+" #I_ "], 0\n\t"
  asm volatile(A512_R16(A512_Z, 0) : : "n"(N) : A512_ALL_AGPRS);
#undef A512_Z
}
template <int N>
__device__ __forceinline__ void a512_acc_scale(float alpha) {
  float t0, t1;
#define A512_S(IDX_, I_) "v_accvgpr_read_b32 %0, a[%3*16+" #I_ "]\n\tv_mul_f32 %0, %2, %0\n\tv_accvgpr_write_b32 a[%3*16+" #I_ "], %0\n\t"
  asm volatile(A512_R16(A512_S, 0) : "=&v"(t0), "=&v"(t1) : "v"(alpha), "n"(N));
#undef A512_S
}
template <int N>
__device__ __forceinline__ float16_t a512_acc_read() {
  float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
#define A512_G(IDX_, I_) "v_accvgpr_read_b32 %" #I_ ", a[%16*16+" #I_ "]\n\t"
  asm volatile(A512_R16(A512_G, 0)
               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7), "=v"(r8), "=v"(r9), "=v"(r10), "=v"(r11),
                 "=v"(r12), "=v"(r13), "=v"(r14), "=v"(r15)
               : "n"(N));
#undef A512_G
  float16_t r = {r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
  return r;
}
template <int N>
__device__ __forceinline__ void a512_mfma(const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%2*16:%2*16+15], %0, %1, a[%2*16:%2*16+15]" : : "v"(a), "v"(b), "n"(N));
}
template <int N, typename F>
__device__ __forceinline__ void a512_for_blocks(F&& f) {
  if constexpr (N < 16) {
    f(std::integral_constant<int, N>{});
    a512_for_blocks<N + 1>(f);
  }
}

__global__ __launch_bounds__(256, 1) void attn512_fwd_kernel(const AttnParams p) {
  ATT_STAMP(0);
  constexpr int KT = 32, ROWB = 1024, TILE = KT * ROWB;     // 32 KiB
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // K stages 0-2, V stages 0-1: 160 KiB, all of the CU's LDS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, ql = lane & 31;
  int bx, hd, b;
  attn_wg(p, bx, hd, b);
  const int q0 = bx * 128 + wave * 32;
  const float c = p.scale * LOG2E;

  const bf16_t* Qb = p.Q + (long)b * p.bq + (long)hd * 512;
  const bf16_t* Kb = p.K + (long)b * p.bk + (long)hd * 512;
  const bf16_t* Vb = p.V + (long)b * p.bv + (long)hd * 512;
  const int nt = (p.Lk + KT - 1) / KT;

  // ---- LDS-DMA: a tile is 32 rows = 32 wave instructions, 8 per wave; lane l of row r fetches chunk (l & 48) | ((l & 15) ^ g(r)).
  // Whole tiles: wave-uniform tile pointer + one per-lane byte offset per piece, kept in registers (K and V share them: whole-tile path needs sk == sv).
  unsigned doff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = wave * 8 + j;
    const int ch = (lane & 48) | ((lane & 15) ^ a512_g(row));
    doff[j] = (unsigned)((row * p.sk + ch * 8) * 2);
  }
  auto issue_whole = [&](const bf16_t* tile, char* img) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((att_gptr)((const char*)tile + doff[j]), (att_lptr)(img + (wave * 8 + j) * ROWB), 16, 0, 0);
  };
  // the last tile when Lk % 32 != 0: rows past the end come from row Lk - 1 (masked below)
  auto issue_tail = [&](const bf16_t* base, long stride, int row0, char* img) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wave * 8 + j;
      const int r = min(row0 + row, p.Lk - 1);
      const int ch = (lane & 48) | ((lane & 15) ^ a512_g(row));
      __builtin_amdgcn_global_load_lds((att_gptr)(base + (long)r * stride + ch * 8), (att_lptr)(img + row * ROWB), 16, 0, 0);
    }
  };
  auto issue_k = [&](int t, char* img) {
    if ((t + 1) * KT <= p.Lk && p.sk == p.sv) issue_whole(Kb + (long)t * KT * p.sk, img); else issue_tail(Kb, p.sk, t * KT, img);
  };
  auto issue_v = [&](int t, char* img) {
    if ((t + 1) * KT <= p.Lk && p.sk == p.sv) issue_whole(Vb + (long)t * KT * p.sv, img); else issue_tail(Vb, p.sv, t * KT, img);
  };
  issue_k(0, smem);                       // K stages 0-2, V stages 3-4
  issue_v(0, smem + 3 * TILE);
  if (nt > 1) issue_k(1, smem + TILE);

  // ---- per-lane fragment addresses: ONE tile-relative base per operand; every other fragment address is that base XOR a constant
  // (eight + eight precomputed addresses, and the per-stage sums the compiler kept of them, were what pushed this kernel into scratch --
  // and a scratch reload inside the key loop waits for vmcnt(0), i.e. for the tile DMA).  The XORs run in the shadow of the MFMAs.
  // K row reads: fragment ks = chunk 2 ks + h5 of row ql: slot (2 ks) ^ h5 ^ g(ql) -> kab0 ^ ((ks & 7) << 5), + (ks >> 3) * 256
  const unsigned kab0 = (unsigned)(ql * ROWB + ((h5 ^ a512_g(ql)) << 4));
  // V^T transposed reads (see tr_frag): lane -> row 4 h + q4 (+ 8 for the high half, + 16 s2), 8 bytes at columns 32 dt + 16 (g4 & 1) + 4 pp:
  // chunk 4 dt + c0, c0 = 2 (g4 & 1) + (pp >> 1); slot = (dt >> 2) * 16 + (((dt & 3) ^ q4) << 2 | (c0 ^ ((h + 2 j) & 3)))
  //   -> vab0 ^ ((dt & 3) << 6) ^ (j ? 8192 | 32 : 0)      (h <= 1: (h + 2) & 3 == h ^ 2; row + 8 sets bit 13)
  unsigned vab0;
  {
    const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h = g4 >> 1;
    const int c0 = 2 * (g4 & 1) + (pp >> 1);
    vab0 = (unsigned)((4 * h + q4) * ROWB + (((q4 << 2) | (c0 ^ h)) << 4) + 8 * (pp & 1));
  }
  if ((unsigned)(size_t)(lds_c)smem & (TILE - 1)) __builtin_trap();      // the XOR addressing needs stage bases that are multiples of 32 KiB

  // ---- Q'^T fragments: bf16(Q * scale * log2 e), as the head-dim-64 kernel (and the reference's math path) rounds them
  bf16x8_t qf[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    uint4_t z = {0u, 0u, 0u, 0u};
    if (q0 + ql < p.Lq) z = *(const uint4_t*)(Qb + (long)(q0 + ql) * p.sq + 16 * ks + 8 * h5);
    float f[8];
    unpack8(z, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= c;
    qf[ks] = __builtin_bit_cast(bf16x8_t, pack8(f));
    if ((ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // eight rows' worth of loads in flight at a time (register pressure)
  }
  a512_for_blocks<0>([&](auto n) { a512_acc_zero<decltype(n)::value>(); });
  float m = 0.f, l = 0.f;      // m: reference point of this lane's query row (log2 units); l: this lane's half of the row sum

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const unsigned smem_a = (unsigned)(size_t)(lds_c)smem;
  ATT_STAMP(1); ATT_STAMP_RT(4);
#ifdef NK_ATTN_STAMPS      // (diagnostic build, tools/attn512_stamps.py: where an iteration's cycles go; stamps only where no LDS read is outstanding)
  unsigned long long st_a = 0, st_b = 0, st_c = 0, acc_s = 0, acc_pv = 0, acc_sync = 0;
#define A512_STAMP(x) x = __builtin_amdgcn_s_memtime()
#else
#define A512_STAMP(x)
#endif
  // batch B_ (0..7) of a score chain = fragments ks = 4 B_ .. 4 B_ + 3: addresses (stage + kab0) ^ ((4 (B_ & 1) + j) << 5), + (B_ >> 1) * 256
#define A512_RDK(dst, KT_, B_)                                                                                                  \
  A512_RD128(dst[0], KT_ ^ ((4 * ((B_) & 1) + 0) << 5), ((B_) >> 1) * 256); A512_RD128(dst[1], KT_ ^ ((4 * ((B_) & 1) + 1) << 5), ((B_) >> 1) * 256); \
  A512_RD128(dst[2], KT_ ^ ((4 * ((B_) & 1) + 2) << 5), ((B_) >> 1) * 256); A512_RD128(dst[3], KT_ ^ ((4 * ((B_) & 1) + 3) << 5), ((B_) >> 1) * 256)
  // (the score chain in VGPRs by inline asm as well: as a builtin chain the compiler placed it in a[0:15], on top of d-tile 0.  Back-to-back
  // MFMAs accumulating into the same block need no wait states; the vector reads behind a chain get theirs from an s_nop.)
#define A512_MMK0(S_, src)                                                                                                      \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(S_) : "v"(src[0]), "v"(qf[0]));                                 \
  _Pragma("unroll") for (int j = 1; j < 4; ++j)                                                                                 \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S_) : "v"(src[j]), "v"(qf[j]))
#define A512_MMK(S_, src, B_)                                                                                                   \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                                 \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S_) : "v"(src[j]), "v"(qf[4 * (B_) + j]))
#define A512_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- S'^T(0) = K(0) Q'^T: 32 chained MFMAs, K fragments in eight batches of four, each requested one batch ahead
  float16_t s_cur, s_nxt;
  {
    bf16x8_t ka[4], kb[4];
    unsigned kt = smem_a + kab0;
    A64_PIN(kt);
    A512_RDK(ka, kt, 0); A512_RDK(kb, kt, 1);
    A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK0(s_cur, ka);
    A512_RDK(ka, kt, 2); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_cur, kb, 1);
    A512_RDK(kb, kt, 3); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_cur, ka, 2);
    A512_RDK(ka, kt, 4); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_cur, kb, 3);
    A512_RDK(kb, kt, 5); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_cur, ka, 4);
    A512_RDK(ka, kt, 6); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_cur, kb, 5);
    A512_RDK(kb, kt, 7); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_cur, ka, 6);
    A512_WAIT4(kb); A512_SB(); A512_MMK(s_cur, kb, 7);
    asm volatile("s_nop 15" : "+v"(s_cur));
  }

  // ---- the key loop, software-pipelined: iteration t runs the score chain of tile t + 1 (matrix pipe) UNDER the softmax of tile t
  // (vector pipe; with one wave per SIMD nothing else would cover it: ~500 of ~2 800 cycles per tile in the unpipelined version), then
  // O^T += V(t)^T P(t)^T.  K tiles therefore arrive two tiles ahead (three stages), V tiles one ahead (two stages).
  // Where an iteration's ~4 450 cycles go (diagnostic build, tools/attn512_stamps.py; MFMA floor 2 x 1 024): score phase 2 590, second product
  // 1 250, DMA wait + barrier 325.  The score phase carries the ISSUE of the sixteen 1-KiB DMA instructions: a wave issues in order and the
  // fill path takes 64 B / clk / CU, ~960 cycles with all four waves issuing (timing-only build without the DMA: 1 630).  Measured and
  // dropped, all slower at 4 x 16384 tokens (2 233 us): K three tiles ahead / V awaited in front of the second product with counted vmcnt
  // (2 301); the pieces spread one or two per MFMA batch (2 654 / 2 935) or staggered over the waves (2 454) -- more than sixteen pieces in
  // flight per wave stall at issue, and every variant the compiler answered with scratch reloads that wait for vmcnt(0); two score chains
  // instead of one (2 257: the chain's dependency is not what the phase waits for); the chain's accumulator in AGPRs (no change).
  unsigned kst_n = TILE, kst_i = 2 * TILE;     // K stage of tile t + 1 (read here) / of tile t + 2 (filled here); tile t's was read an iteration ago
  for (int t = 0; t < nt; ++t) {
    asm volatile("" ::: A512_ALL_AGPRS);      // (nothing of the compiler's lives in the AGPRs across an iteration)
    A512_STAMP(st_a);
    const int vs = t & 1;
    if (t + 2 < nt) issue_k(t + 2, smem + kst_i);
    if (t + 1 < nt) issue_v(t + 1, smem + (3 + (vs ^ 1)) * TILE);
    unsigned kt = smem_a + kst_n + kab0, vt = smem_a + (3 + vs) * TILE + vab0;
    A64_PIN(kt); A64_PIN(vt);       // (opaque: keeps the XORed addresses from being hoisted out of the loop into registers)
    // (past the last tile the chain below runs on whatever its stage holds and its result is dropped: cheaper than a second copy of the loop body)

    bf16x8_t ka[4], kb[4];
    A512_RDK(ka, kt, 0); A512_RDK(kb, kt, 1);
    A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK0(s_nxt, ka);
    // -- softmax(t), part 1: mask, row maximum (both halves of the key tile through one v_permlane32_swap: no LDS traffic among the counted reads)
    A64_PIN(s_cur);
    if (t == nt - 1 && (p.Lk & (KT - 1))) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_cur[r] = (t * KT + acc_row(r, h5)) < p.Lk ? s_cur[r] : NEG_BIG;
    }
    float mloc = fmaxf(fmaxf(s_cur[0], s_cur[1]), s_cur[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mloc = fmaxf(fmaxf(mloc, s_cur[r]), s_cur[r + 1]);
    mloc = fmaxf(mloc, s_cur[15]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
      mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    A64_PIN(mloc); A512_SB();
    A512_RDK(ka, kt, 2); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_nxt, kb, 1);
    // -- part 2: the reference point m moves only when a score exceeds it by more than 2^8 ("defer-max", cdna guide T13); rare after tile 0
    A64_PIN(mloc);
    if (t == 0 || __any(mloc > m + 8.0f)) {
      const float mnew = t == 0 ? mloc : fmaxf(m, mloc);      // the first tile defines m (whatever its sign); later m only rises
      if (t != 0) {
        const float alpha = EXP2(m - mnew);
        l *= alpha;
        a512_for_blocks<0>([&](auto n) { a512_acc_scale<decltype(n)::value>(alpha); });
      }
      m = mnew;
    }
    A64_PIN(m); A512_SB();
    A512_RDK(kb, kt, 3); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_nxt, ka, 2);
    // -- parts 3-6: the exponentials, four registers per gap
#define A512_EXP4(R0_)                                                                                                          \
    A64_PIN(s_cur);                                                                                                             \
    _Pragma("unroll") for (int r = (R0_); r < (R0_) + 4; ++r) s_cur[r] = EXP2(s_cur[r] - m);                                    \
    A64_PIN(s_cur); A512_SB()
    A512_EXP4(0);
    A512_RDK(ka, kt, 4); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_nxt, kb, 3);
    A512_EXP4(4);
    A512_RDK(kb, kt, 5); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_nxt, ka, 4);
    A512_EXP4(8);
    A512_RDK(ka, kt, 6); A512_WAIT4_KEEP4(kb); A512_SB(); A512_MMK(s_nxt, kb, 5);
    A512_EXP4(12);
#undef A512_EXP4
    A512_RDK(kb, kt, 7); A512_WAIT4_KEEP4(ka); A512_SB(); A512_MMK(s_nxt, ka, 6);
    // -- part 7: row sum, P^T as the bf16 B operands of the second product
    A64_PIN(s_cur);
    float lsum = (s_cur[0] + s_cur[1]) + (s_cur[2] + s_cur[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 4) lsum += (s_cur[r] + s_cur[r + 1]) + (s_cur[r + 2] + s_cur[r + 3]);
    l += lsum;
    bf16x8_t pf0 = pack_frag(s_cur, 0), pf1 = pack_frag(s_cur, 1);
    A64_PIN(pf0); A64_PIN(pf1); A64_PIN(l); A512_SB();
    A512_WAIT4(kb); A512_SB();
    A512_STAMP(st_b);
    // ---- the first two batches of V^T fragments are requested right behind the chain's last four MFMAs, which cover most of their latency
    // batch (s2, q) = k-step s2 (keys 16 s2 ..), d-tiles 4 q .. 4 q + 3
    A512V va, vb;
#define A512_RDV(dst, S2_, Q_)                                                                                                  \
    A512_RDTR(dst.lo[0], vt, (S2_) * 16384 + (Q_) * 256);              A512_RDTR(dst.hi[0], vt ^ 8224u, (S2_) * 16384 + (Q_) * 256);              \
    A512_RDTR(dst.lo[1], vt ^ (1u << 6), (S2_) * 16384 + (Q_) * 256);  A512_RDTR(dst.hi[1], vt ^ (8224u | (1u << 6)), (S2_) * 16384 + (Q_) * 256);  \
    A512_RDTR(dst.lo[2], vt ^ (2u << 6), (S2_) * 16384 + (Q_) * 256);  A512_RDTR(dst.hi[2], vt ^ (8224u | (2u << 6)), (S2_) * 16384 + (Q_) * 256);  \
    A512_RDTR(dst.lo[3], vt ^ (3u << 6), (S2_) * 16384 + (Q_) * 256);  A512_RDTR(dst.hi[3], vt ^ (8224u | (3u << 6)), (S2_) * 16384 + (Q_) * 256)
    A512_MMK(s_nxt, kb, 7);
    A512_RDV(va, 0, 0);
    A512_RDV(vb, 0, 1);

    // ---- O^T += V^T P^T: 2 k-steps x 16 d-tiles, V^T fragments in eight batches of four, each requested two batches ahead
#define A512_PV(src, PF_, Q_)                                                                                                   \
    a512_mfma<4 * (Q_) + 0>(a64_join(src.lo[0], src.hi[0]), PF_); a512_mfma<4 * (Q_) + 1>(a64_join(src.lo[1], src.hi[1]), PF_);  \
    a512_mfma<4 * (Q_) + 2>(a64_join(src.lo[2], src.hi[2]), PF_); a512_mfma<4 * (Q_) + 3>(a64_join(src.lo[3], src.hi[3]), PF_)
    A512_WAITV_KEEP8(va); A512_SB(); A512_PV(va, pf0, 0);
    A512_RDV(va, 0, 2); A512_WAITV_KEEP8(vb); A512_SB(); A512_PV(vb, pf0, 1);
    A512_RDV(vb, 0, 3); A512_WAITV_KEEP8(va); A512_SB(); A512_PV(va, pf0, 2);
    A512_RDV(va, 1, 0); A512_WAITV_KEEP8(vb); A512_SB(); A512_PV(vb, pf0, 3);
    A512_RDV(vb, 1, 1); A512_WAITV_KEEP8(va); A512_SB(); A512_PV(va, pf1, 0);
    A512_RDV(va, 1, 2); A512_WAITV_KEEP8(vb); A512_SB(); A512_PV(vb, pf1, 1);
    A512_RDV(vb, 1, 3); A512_WAITV_KEEP8(va); A512_SB(); A512_PV(va, pf1, 2);
    A512_WAITV(vb); A512_SB();
    A512_STAMP(st_c);
    A512_PV(vb, pf1, 3);
#undef A512_PV
#undef A512_RDV

    // K(t + 2) and V(t + 1) have landed (this wave's rows by the wait, everyone's by the barrier, which also releases K(t + 1)'s and V(t)'s stages);
    // the s_nop gives the score chain of tile t + 1 its wait states before vector code reads it
    A512_SB();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" : "+v"(s_nxt) : : "memory");
    __builtin_amdgcn_s_barrier();
    A512_SB();
#ifdef NK_ATTN_STAMPS
    { const unsigned long long st_d = __builtin_amdgcn_s_memtime(); acc_s += st_b - st_a; acc_pv += st_c - st_b; acc_sync += st_d - st_c; }
#endif
    s_cur = s_nxt;
    kst_n += TILE; if (kst_n == 3 * TILE) kst_n = 0;
    kst_i += TILE; if (kst_i == 3 * TILE) kst_i = 0;
  }
  ATT_STAMP(2); ATT_STAMP_RT(5);
#ifdef NK_ATTN_STAMPS
  if (threadIdx.x == 0) {
    const unsigned w_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (w_ < 8192) { nk_attn_stamp_buf[w_ * 8 + 3] = acc_s; nk_attn_stamp_buf[w_ * 8 + 6] = acc_pv; nk_attn_stamp_buf[w_ * 8 + 7] = acc_sync; }
  }
#endif
#undef A512_STAMP
#undef A512_RDK
#undef A512_MMK0
#undef A512_MMK
#undef A512_SB

  // (the last MFMAs are invisible to the compiler's hazard recogniser: their wait states before the AGPRs are read)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory", A512_ALL_AGPRS);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int q = q0 + ql;
  {
    // 16-byte stores: column groups k, k + 1 of a row paired across the half-waves by v_permlane32_swap (see attn64_fwd_kernel's epilogue)
    const bool live = q < p.Lq;
    bf16_t* Ob = p.O + (long)b * p.bo + (long)(live ? q : 0) * p.so + (long)hd * 512 + 8 * h5;
    a512_for_blocks<0>([&](auto n) {
      constexpr int dt = decltype(n)::value;
      const float16_t oa = a512_acc_read<dt>();
      uint2_t o2[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        o2[r4].x = pack2bf(oa[4 * r4 + 0] * inv, oa[4 * r4 + 1] * inv);
        o2[r4].y = pack2bf(oa[4 * r4 + 2] * inv, oa[4 * r4 + 3] * inv);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; r4 += 2) {
        const auto sx = __builtin_amdgcn_permlane32_swap(o2[r4].x, o2[r4 + 1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(o2[r4].y, o2[r4 + 1].y, false, false);
        uint4_t w = {sx[0], sy[0], sx[1], sy[1]};
        if (live) *(uint4_t*)(Ob + dt * 32 + 8 * r4) = w;
      }
    });
  }
  if (q < p.Lq) {
    if (h5 == 0 && p.LSE) p.LSE[((long)b * p.H + hd) * p.Lq + q] = (m + __log2f(l)) * 0.6931471805599453f;
  }
}
