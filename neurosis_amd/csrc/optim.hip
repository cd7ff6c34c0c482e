// Fused multi-tensor Adafactor over the flat parameter / gradient buffers.
// Replaces the reference's per-tensor Python loop (`optimizers/adafactor.py:162-255`: ~1 700 tensors x ~15 small kernels,
// `_approx_sq_grad` :154-159, `_rms` :150-151, `_get_lr` :133-147) with five launches per CHUNK of tensors:
//
//   A  statistics   matrices: row / column partial sums of g^2+eps per 256x64 tile
//                   conv (factored over the 3x3 taps of every (o,i) pair, as the reference's rule "last two dims" implies
//                   for OIHW weights) and vectors: state update and u = g / sqrt(v) in one go, partial sum of u^2
//   F1 per matrix   row / column EMAs from the partials, mean of the row EMA
//   B  matrices     partial sum of u^2,  u = g * rsqrt(row_i / mean_row) * rsqrt(col_j)
//   F2 per tensor   RMS(u) -> clip denominator, RMS(p) -> relative step size, scale = lr / denominator
//   C  apply        p -= wd*lr*p + scale*u ; bf16 shadow rewritten ; partial sum of p^2 (next step's RMS(p))
//
// A chunk is a run of consecutive tensors (host-chosen size).  Per parameter the passes move g three times, p once in and once
// out and the bf16 shadow out: 22 B against fused AdamW's 30, and no m / v buffers (-20 GB for the SDXL UNet).  Chunks small
// enough for the gradients to be re-read from the 256 MB Infinity Cache were measured SLOWER (5 dependent launches per chunk
// cost more than the HBM re-reads), so the default chunk is 2 GB.  Every reduction goes through per-tile partials combined
// in a fixed order: no atomics, bitwise reproducible.
#include "../../include/neurosis_hip.h"
#include "nk_common.h"

struct NkAfTensor {      // mirrored by neurosis_amd/optim.py (AF_TENSOR_DTYPE); 72 bytes
  long off;              // element offset of the tensor in master / grad / shadow
  long row_off;          // matrix: row EMA [d0]; conv: [O][KH][I]; vector: full second moment [numel]   (offset in `state`)
  long col_off;          // matrix: col EMA [d1]; conv: [O][KW][I]
  long ws_row;           // matrix: row partials [ntc][d0], offset in the chunk workspace
  long ws_col;           // matrix: col partials [ntr][d1]
  int kind;              // 0 vector, 1 matrix [d0][d1], 2 conv stored [O=d0][KH][KW][I=d1]
  int d0, d1;
  int kh, kw;
  int item0, nitems;     // this tensor's range in the item table
  int mr0;               // matrix: first slot of its row-EMA partial sums (one per 256-row strip) in `mean_row`
  int cnt0;              // first of this tensor's "blocks done" counters: [0] the tensor's, matrices: [1 + tr] row strips, [1 + ntr + tc] column strips
  int pad;
};
static_assert(sizeof(NkAfTensor) == 80, "mirrored by neurosis_amd/optim.py");

struct NkAfItem { int tensor, tr, tc, pad; };

struct NkAfArgs {
  float* master; const float* grad; bf16_t* shadow; float* state; float* ws;
  const NkAfTensor* tensors; const NkAfItem* items;
  float* u2_part;      // [nitems] partial sums of u^2
  float* p2_part;      // [nitems] partial sums of p^2 after the update (read by the NEXT step's F2)
  float* mean_row;     // per matrix: one partial sum of the row EMA per 256-row strip (slots mr0...)
  unsigned* counters;  // "blocks done" counters (zero before the step; every one is back at zero when its last block has passed)
  float* scale;        // [ntensors] lr / max(1, rms(u)/clip)
  float* lr_t;         // [ntensors]
  int item_lo, item_hi, tensor_lo, tensor_hi;
  float beta2t, eps1, eps2, clip, rel_step, weight_decay, grad_scale;
  int scale_parameter;
  const unsigned* health;   // backward-health word (errors.hip): non-zero -> every kernel of the step returns at once
};
// (wave-uniform scalar load; `volatile` so it is not hoisted or cached across the check)
#define AF_HEALTH_GATE(a) do { if (*(const volatile unsigned*)(a).health) return; } while (0)

#define AF_TR 256   // matrix tile rows
#define AF_TC 64    // matrix tile cols
#define AF_CONV_PAIRS 1024
#define AF_VEC 1024

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// u of one conv (o,i) pair from the updated EMAs; g[kh][kw] are that pair's gradients (KH, KW <= 3)
template <int KH_, int KW_>
__device__ __forceinline__ void conv_u(const float (&row)[3], const float (&col)[3], const float (&g)[3][3], int KH, int KW,
                                       float (&u)[3][3]) {
  float mr = 0.f;
#pragma unroll
  for (int x = 0; x < 3; ++x)
    if (x < KH) mr += row[x];
  mr /= (float)KH;
#pragma unroll
  for (int x = 0; x < 3; ++x)
    if (x < KH) {
      const float rf = rsqrtf(row[x] / mr);
#pragma unroll
      for (int y = 0; y < 3; ++y)
        if (y < KW) u[x][y] = rf * rsqrtf(col[y]) * g[x][y];
    }
}

// conv weights: one thread per (o, i) pair, everything about that pair's KH x KW taps stays in registers.
// KH_/KW_ = 0: runtime sizes (<= 3), otherwise compile-time (no scratch-resident arrays).
template <int KH_, int KW_>
__device__ __forceinline__ float af_conv_stats(const NkAfArgs& a, const NkAfTensor& t, int tr, int tid, float gs) {
  const int KH = KH_ ? KH_ : t.kh, KW = KW_ ? KW_ : t.kw, I = t.d1;
  const long npairs = (long)t.d0 * I;
  float acc = 0.f;
  for (int q = 0; q < AF_CONV_PAIRS / 256; ++q) {
    const long pr = (long)tr * AF_CONV_PAIRS + q * 256 + tid;
    if (pr < npairs) {
      const int o = (int)(pr / I), i = (int)(pr - (long)o * I);
      float g[3][3], upd[3][3], row[3], col[3], u[3][3];
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
          if (x < KH && y < KW) {
            g[x][y] = a.grad[t.off + (((long)o * KH + x) * KW + y) * I + i] * gs;
            upd[x][y] = g[x][y] * g[x][y] + a.eps1;
          }
#pragma unroll
      for (int x = 0; x < 3; ++x)
        if (x < KH) {
          float m = 0.f;
#pragma unroll
          for (int y = 0; y < 3; ++y)
            if (y < KW) m += upd[x][y];
          float* st = a.state + t.row_off + ((long)o * KH + x) * I + i;
          row[x] = *st * a.beta2t + (m / (float)KW) * (1.0f - a.beta2t);
          *st = row[x];
        }
#pragma unroll
      for (int y = 0; y < 3; ++y)
        if (y < KW) {
          float m = 0.f;
#pragma unroll
          for (int x = 0; x < 3; ++x)
            if (x < KH) m += upd[x][y];
          float* st = a.state + t.col_off + ((long)o * KW + y) * I + i;
          col[y] = *st * a.beta2t + (m / (float)KH) * (1.0f - a.beta2t);
          *st = col[y];
        }
      conv_u<KH_, KW_>(row, col, g, KH, KW, u);
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
          if (x < KH && y < KW) acc += u[x][y] * u[x][y];
    }
  }
  return acc;
}

template <int KH_, int KW_>
__device__ __forceinline__ float af_conv_apply(const NkAfArgs& a, const NkAfTensor& t, int tr, int tid, float gs, float sc, float decay) {
  const int KH = KH_ ? KH_ : t.kh, KW = KW_ ? KW_ : t.kw, I = t.d1;
  const long npairs = (long)t.d0 * I;
  float acc = 0.f;
  for (int q = 0; q < AF_CONV_PAIRS / 256; ++q) {
    const long pr = (long)tr * AF_CONV_PAIRS + q * 256 + tid;
    if (pr < npairs) {
      const int o = (int)(pr / I), i = (int)(pr - (long)o * I);
      float g[3][3], row[3], col[3], u[3][3];
#pragma unroll
      for (int x = 0; x < 3; ++x)
        if (x < KH) row[x] = a.state[t.row_off + ((long)o * KH + x) * I + i];
#pragma unroll
      for (int y = 0; y < 3; ++y)
        if (y < KW) col[y] = a.state[t.col_off + ((long)o * KW + y) * I + i];
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
          if (x < KH && y < KW) g[x][y] = a.grad[t.off + (((long)o * KH + x) * KW + y) * I + i] * gs;
      conv_u<KH_, KW_>(row, col, g, KH, KW, u);
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
          if (x < KH && y < KW) {
            const long e = t.off + (((long)o * KH + x) * KW + y) * I + i;
            const float pn = a.master[e] * decay - sc * u[x][y];
            a.master[e] = pn;
            a.shadow[e] = f2bf(pn);
            acc += pn * pn;
          }
    }
  }
  return acc;
}

// "last block done": count this block in; true (for every thread of the block) in the block that completes `total`.
// No fences: an agent-scope release / acquire writes back and invalidates the XCD's WHOLE L2 -- with every block of a 1 000-block launch
// doing that beside the next step's VAE encoder the step went from 165 to 238 ms (round 4, first version).  Instead every handed-off
// value is STORED with an agent-scope atomic store (sc1: written through to the coherence point) and drained (vmcnt(0)) before the
// block is counted, and the finishing block LOADS them with agent-scope atomic loads (sc1: past its own non-coherent L2 lines) --
// the second valid form of MI355X_MICROARCH.md "Correctness boundaries".
#define AF_PUBLISH(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AF_FETCH(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ bool af_last_block(unsigned* counter, unsigned total, int tid, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's published values have been acknowledged
  __syncthreads();
  if (tid == 0) {
    const unsigned old = atomicAdd(counter, 1u);
    *flag = old + 1u == total;
    if (old + 1u == total) AF_PUBLISH(counter, 0u);   // back to zero for the next step (nobody else touches it any more)
  }
  __syncthreads();
  return *flag != 0u;
}

// one tensor's step size from its items' partial sums (RMS(u) -> clip denominator, RMS(p) -> relative step size)
__device__ __forceinline__ void af_tensor_scale(const NkAfArgs& a, const NkAfTensor& t, int ti, int tid, float* red) {
  float su = 0.f, sp = 0.f;
  for (int i = tid; i < t.nitems; i += 256) {
    su += AF_FETCH(a.u2_part + t.item0 + i);       // (this launch's blocks)
    sp += a.p2_part[t.item0 + i];                  // (an earlier launch: the previous step's apply pass)
  }
  su = block_sum_256(su, red);
  sp = block_sum_256(sp, red);
  if (tid == 0) {
    const float numel = t.kind == 0 ? (float)t.d0 : (float)t.d0 * (float)t.d1 * (float)(t.kind == 2 ? t.kh * t.kw : 1);
    const float rms_u = sqrtf(su / numel);
    const float denom = fmaxf(1.0f, rms_u / a.clip);
    const float rms_p = sqrtf(sp) / sqrtf(numel);
    const float lr = (a.scale_parameter ? fmaxf(a.eps2, rms_p) : 1.0f) * a.rel_step;
    a.lr_t[ti] = lr;
    a.scale[ti] = lr / denom;
  }
}

__global__ __launch_bounds__(256) void af_stats_kernel(const NkAfArgs a) {
  AF_HEALTH_GATE(a);
  __shared__ float red[4];
  __shared__ float colsh[16][AF_TC + 4];
  __shared__ float rowsh[AF_TR];
  __shared__ unsigned flag[2];
  const NkAfItem it = a.items[a.item_lo + blockIdx.x];
  const NkAfTensor t = a.tensors[it.tensor];
  const int tid = threadIdx.x;
  const float gs = a.grad_scale;
  if (t.kind == 1) {
    const int ry = tid >> 4, cx = tid & 15;
    const int col = it.tc * AF_TC + cx * 4;
    const bool cok = col < t.d1;
    const int ntc = (t.d1 + AF_TC - 1) / AF_TC, ntr = (t.d0 + AF_TR - 1) / AF_TR;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    float* rowpart = a.ws + t.ws_row + (long)it.tc * t.d0;
#pragma unroll 1
    for (int i0 = 0; i0 < AF_TR / 16; i0 += 4) {
      float4_t g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = it.tr * AF_TR + ry + 16 * (i0 + u);
        g[u] = (cok && r < t.d0) ? *(const float4_t*)(a.grad + t.off + (long)r * t.d1 + col) : (float4_t){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = it.tr * AF_TR + ry + 16 * (i0 + u);
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = g[u][e] * gs;
          const float v = (cok && r < t.d0) ? gg * gg + a.eps1 : 0.f;
          cs[e] += v;
          rs += v;
        }
        // sum over the 16 column lanes of this row (lanes cx = 0..15 are adjacent within the wave)
        rs += __shfl_xor(rs, 8); rs += __shfl_xor(rs, 4); rs += __shfl_xor(rs, 2); rs += __shfl_xor(rs, 1);
        // (through LDS, published in one piece below: one coalesced 1 KiB store per block instead of 64 masked 16-byte write-through stores
        // on the same vmcnt as the next batch of loads; 51.9 -> 50.3 us per 128 MB chunk.  What holds this pass at 2.4 TB/s against the
        // 4.5 of af_u2_kernel's identical reads is its tail -- two counter round trips and the finishing blocks' reductions, with one
        // generation of blocks per chunk and nothing to overlap them; bigger chunks run the update faster alone (13.6 -> 11.1 ms) and the
        // step slower beside the frozen VAE encoder (146.2 -> 147.2 ms): profiles/r06_adafactor_chunks.txt)
        if (cx == 0) rowsh[ry + 16 * (i0 + u)] = rs;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) colsh[ry][cx * 4 + e] = cs[e];
    __syncthreads();
    if (it.tr * AF_TR + tid < t.d0) AF_PUBLISH(rowpart + it.tr * AF_TR + tid, rowsh[tid]);      // 256 consecutive floats
    if (tid < AF_TC) {
      const int c = it.tc * AF_TC + tid;
      if (c < t.d1) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += colsh[r][tid];
        AF_PUBLISH(a.ws + t.ws_col + (long)it.tr * t.d1 + c, s);
      }
    }
    // the last tile of this 256-row strip: row EMAs of the strip and the sum of them (mean of the row EMA = sum of the slots / d0)
    if (af_last_block(a.counters + t.cnt0 + 1 + it.tr, (unsigned)ntc, tid, &flag[0])) {
      const int r = it.tr * AF_TR + tid;
      float v = 0.f;
      if (r < t.d0) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < ntc; ++c) s += AF_FETCH(a.ws + t.ws_row + (long)c * t.d0 + r);   // (independent loads: keep 8 in flight)
        float* st = a.state + t.row_off + r;
        v = *st * a.beta2t + (s / (float)t.d1) * (1.0f - a.beta2t);
        *st = v;
      }
      v = block_sum_256(v, red);
      if (tid == 0) a.mean_row[t.mr0 + it.tr] = v;
    }
    // ... and of this 64-column strip: its column EMAs
    if (af_last_block(a.counters + t.cnt0 + 1 + ntr + it.tc, (unsigned)ntr, tid, &flag[1])) {
      const int c = it.tc * AF_TC + tid;
      if (tid < AF_TC && c < t.d1) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < ntr; ++r) s += AF_FETCH(a.ws + t.ws_col + (long)r * t.d1 + c);
        float* st = a.state + t.col_off + c;
        *st = *st * a.beta2t + (s / (float)t.d0) * (1.0f - a.beta2t);
      }
    }
    return;
  }
  float acc;
  if (t.kind == 2) {
    if (t.kh == 3 && t.kw == 3) acc = af_conv_stats<3, 3>(a, t, it.tr, tid, gs);
    else if (t.kh == 1 && t.kw == 1) acc = af_conv_stats<1, 1>(a, t, it.tr, tid, gs);
    else acc = af_conv_stats<0, 0>(a, t, it.tr, tid, gs);
  } else {
    const long n = (long)t.d0;
    acc = 0.f;
    for (int q = 0; q < AF_VEC / 256; ++q) {
      const long e = (long)it.tr * AF_VEC + q * 256 + tid;
      if (e < n) {
        const float g = a.grad[t.off + e] * gs;
        float* st = a.state + t.row_off + e;
        const float v = *st * a.beta2t + (g * g + a.eps1) * (1.0f - a.beta2t);
        *st = v;
        const float u = g * rsqrtf(v);
        acc += u * u;
      }
    }
  }
  acc = block_sum_256(acc, red);
  if (tid == 0) AF_PUBLISH(a.u2_part + a.item_lo + blockIdx.x, acc);
  // convolutions and vectors have their update's RMS now: the last block of the tensor turns it into the step size
  if (af_last_block(a.counters + t.cnt0, (unsigned)t.nitems, tid, &flag[0])) af_tensor_scale(a, t, it.tensor, tid, red);
}

__device__ __forceinline__ float af_mean_row(const NkAfArgs& a, const NkAfTensor& t) {
  float s = 0.f;
  const int n = (t.d0 + AF_TR - 1) / AF_TR;
  for (int i = 0; i < n; ++i) s += a.mean_row[t.mr0 + i];
  return s / (float)t.d0;
}

// matrices: partial sum of u^2 per tile
__global__ __launch_bounds__(256) void af_u2_kernel(const NkAfArgs a) {
  AF_HEALTH_GATE(a);
  __shared__ float red[4];
  __shared__ unsigned flag;
  const NkAfItem it = a.items[a.item_lo + blockIdx.x];
  const NkAfTensor t = a.tensors[it.tensor];
  if (t.kind != 1) return;
  const int tid = threadIdx.x;
  const int ry = tid >> 4, cx = tid & 15;
  const int col = it.tc * AF_TC + cx * 4;
  const bool cok = col < t.d1;
  const float mr = af_mean_row(a, t);
  const float gs = a.grad_scale;
  float ci[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const float4_t c4 = *(const float4_t*)(a.state + t.col_off + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) ci[e] = rsqrtf(c4[e]);
  }
  float acc = 0.f;
#pragma unroll 1
  for (int i0 = 0; i0 < AF_TR / 16; i0 += 4) {
    float4_t g[4];
    float rf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = it.tr * AF_TR + ry + 16 * (i0 + u);
      const bool ok = cok && r < t.d0;
      g[u] = ok ? *(const float4_t*)(a.grad + t.off + (long)r * t.d1 + col) : (float4_t){0.f, 0.f, 0.f, 0.f};
      rf[u] = ok ? rsqrtf(a.state[t.row_off + r] / mr) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float uu = rf[u] * ci[e] * (g[u][e] * gs);
        acc += uu * uu;
      }
  }
  acc = block_sum_256(acc, red);
  if (tid == 0) AF_PUBLISH(a.u2_part + a.item_lo + blockIdx.x, acc);
  if (af_last_block(a.counters + t.cnt0, (unsigned)t.nitems, tid, &flag)) af_tensor_scale(a, t, it.tensor, tid, red);
}

__global__ __launch_bounds__(256) void af_apply_kernel(const NkAfArgs a) {
  AF_HEALTH_GATE(a);
  __shared__ float red[4];
  const NkAfItem it = a.items[a.item_lo + blockIdx.x];
  const NkAfTensor t = a.tensors[it.tensor];
  const int tid = threadIdx.x;
  const float gs = a.grad_scale;
  const float sc = a.scale[it.tensor];
  const float decay = 1.0f - a.weight_decay * a.lr_t[it.tensor];
  float acc = 0.f;
  if (t.kind == 1) {
    const int ry = tid >> 4, cx = tid & 15;
    const int col = it.tc * AF_TC + cx * 4;
    const bool cok = col < t.d1;
    const float mr = af_mean_row(a, t);
    float ci[4] = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
      const float4_t c4 = *(const float4_t*)(a.state + t.col_off + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) ci[e] = rsqrtf(c4[e]);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < AF_TR / 16; i0 += 4) {
      float4_t g[4], p[4];
      float rf[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = it.tr * AF_TR + ry + 16 * (i0 + u);
        const bool ok = cok && r < t.d0;
        const long o = t.off + (long)r * t.d1 + col;
        g[u] = ok ? *(const float4_t*)(a.grad + o) : (float4_t){0.f, 0.f, 0.f, 0.f};
        p[u] = ok ? *(const float4_t*)(a.master + o) : (float4_t){0.f, 0.f, 0.f, 0.f};
        rf[u] = ok ? rsqrtf(a.state[t.row_off + r] / mr) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = it.tr * AF_TR + ry + 16 * (i0 + u);
        if (!(cok && r < t.d0)) continue;
        const long o = t.off + (long)r * t.d1 + col;
        float4_t pn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pn[e] = p[u][e] * decay - sc * (rf[u] * ci[e] * (g[u][e] * gs));
          acc += pn[e] * pn[e];
        }
        *(float4_t*)(a.master + o) = pn;
        uint2_t sh;
        sh.x = pack2bf(pn[0], pn[1]);
        sh.y = pack2bf(pn[2], pn[3]);
        *(uint2_t*)(a.shadow + o) = sh;
      }
    }
  } else if (t.kind == 2) {
    if (t.kh == 3 && t.kw == 3) acc = af_conv_apply<3, 3>(a, t, it.tr, tid, gs, sc, decay);
    else if (t.kh == 1 && t.kw == 1) acc = af_conv_apply<1, 1>(a, t, it.tr, tid, gs, sc, decay);
    else acc = af_conv_apply<0, 0>(a, t, it.tr, tid, gs, sc, decay);
  } else {
    const long n = (long)t.d0;
    for (int q = 0; q < AF_VEC / 256; ++q) {
      const long e = (long)it.tr * AF_VEC + q * 256 + tid;
      if (e < n) {
        const float g = a.grad[t.off + e] * gs;
        const float u = g * rsqrtf(a.state[t.row_off + e]);
        const float pn = a.master[t.off + e] * decay - sc * u;
        a.master[t.off + e] = pn;
        a.shadow[t.off + e] = f2bf(pn);
        acc += pn * pn;
      }
    }
  }
  acc = block_sum_256(acc, red);
  if (tid == 0) a.p2_part[a.item_lo + blockIdx.x] = acc;
}

// partial sums of p^2 for the FIRST step (afterwards pass C leaves them behind)
__global__ __launch_bounds__(256) void af_init_p2_kernel(const NkAfArgs a) {
  __shared__ float red[4];
  const NkAfItem it = a.items[a.item_lo + blockIdx.x];
  const NkAfTensor t = a.tensors[it.tensor];
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (t.kind == 1) {
    const int ry = tid >> 4, cx = tid & 15;
    const int col = it.tc * AF_TC + cx * 4;
    for (int i = 0; i < AF_TR / 16; ++i) {
      const int r = it.tr * AF_TR + ry + 16 * i;
      if (col < t.d1 && r < t.d0) {
        const float4_t p = *(const float4_t*)(a.master + t.off + (long)r * t.d1 + col);
        acc += p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
      }
    }
  } else {
    const long per = t.kind == 2 ? (long)AF_CONV_PAIRS * t.kh * t.kw : AF_VEC;
    const long n = t.kind == 2 ? (long)t.d0 * t.d1 * t.kh * t.kw : (long)t.d0;
    // conv items cover AF_CONV_PAIRS (o,i) pairs = a strided set of elements; summing p^2 needs no pairing, but the item
    // partition must match pass C's, so walk the same pairs
    if (t.kind == 2) {
      const int KH = t.kh, KW = t.kw, I = t.d1;
      const long npairs = (long)t.d0 * I;
      for (int q = 0; q < AF_CONV_PAIRS / 256; ++q) {
        const long pr = (long)it.tr * AF_CONV_PAIRS + q * 256 + tid;
        if (pr < npairs) {
          const int o = (int)(pr / I), i = (int)(pr - (long)o * I);
          for (int x = 0; x < KH * KW; ++x) { const float p = a.master[t.off + ((long)o * KH * KW + x) * I + i]; acc += p * p; }
        }
      }
    } else {
      for (int q = 0; q < AF_VEC / 256; ++q) {
        const long e = (long)it.tr * per + q * 256 + tid;
        if (e < n) { const float p = a.master[t.off + e]; acc += p * p; }
      }
    }
  }
  acc = block_sum_256(acc, red);
  if (tid == 0) a.p2_part[a.item_lo + blockIdx.x] = acc;
}

static int af_check(const NkAdafactorArgs* h) {
  NK_CHECK_ARG(h && h->master && h->grad && h->shadow && h->state && h->ws && h->tensors && h->items);
  NK_CHECK_ARG(h->u2_part && h->p2_part && h->mean_row && h->scale && h->lr_t && h->counters);
  NK_CHECK_ARG(h->item_hi > h->item_lo && h->tensor_hi > h->tensor_lo);
  return NK_OK;
}
static NkAfArgs af_args(const NkAdafactorArgs* h) {
  NkAfArgs a;
  a.master = h->master; a.grad = h->grad; a.shadow = (bf16_t*)h->shadow; a.state = h->state; a.ws = h->ws;
  a.tensors = (const NkAfTensor*)h->tensors; a.items = (const NkAfItem*)h->items;
  a.u2_part = h->u2_part; a.p2_part = h->p2_part; a.mean_row = h->mean_row; a.scale = h->scale; a.lr_t = h->lr_t;
  a.counters = h->counters;
  a.item_lo = h->item_lo; a.item_hi = h->item_hi; a.tensor_lo = h->tensor_lo; a.tensor_hi = h->tensor_hi;
  a.beta2t = h->beta2t; a.eps1 = h->eps1; a.eps2 = h->eps2; a.clip = h->clip_threshold; a.rel_step = h->rel_step;
  a.weight_decay = h->weight_decay; a.grad_scale = h->grad_scale; a.scale_parameter = h->scale_parameter;
  a.health = nk_health_word();
  return a;
}

extern "C" long nk_adafactor_tensor_bytes(void) { return (long)sizeof(NkAfTensor); }

extern "C" int nk_adafactor_init(const NkAdafactorArgs* h, void* stream) {
  if (int e = af_check(h)) return e;
  NkAfArgs a = af_args(h);
  hipLaunchKernelGGL(af_init_p2_kernel, dim3(a.item_hi - a.item_lo), dim3(256), 0, (hipStream_t)stream, a);
  return nk_check_launch("af_init_p2_kernel");
}

extern "C" int nk_adafactor_chunk(const NkAdafactorArgs* h, void* stream_) {
  if (int e = af_check(h)) return e;
  if (h->tensor_lo == 0)                        // first chunk of a step: an EARLIER step's backward was flagged -> refuse to go on silently
    if (int e = nk_health_poll()) return e;     // (later chunks of the same step must not trip over this step's own snapshot)
  hipStream_t stream = (hipStream_t)stream_;
  NkAfArgs a = af_args(h);
  if (!a.health) { nk_set_error(__FILE__, __LINE__, "health word allocation failed"); return NK_ERR_LAUNCH; }
  const int nitems = a.item_hi - a.item_lo, ntens = a.tensor_hi - a.tensor_lo;
  (void)ntens;
  hipLaunchKernelGGL(af_stats_kernel, dim3(nitems), dim3(256), 0, stream, a);
  if (int e = nk_check_launch("af_stats_kernel")) return e;
  if (h->has_matrix) {
    hipLaunchKernelGGL(af_u2_kernel, dim3(nitems), dim3(256), 0, stream, a);
    if (int e = nk_check_launch("af_u2_kernel")) return e;
  }
  hipLaunchKernelGGL(af_apply_kernel, dim3(nitems), dim3(256), 0, stream, a);
  if (int e = nk_check_launch("af_apply_kernel")) return e;
  nk_health_snapshot(stream);                    // what the backward in front of this update left in the word
  return NK_OK;
}

// ---- LitEma.forward (modules/ema.py:40-59) over the flat buffer: shadow -= (1 - decay) * (shadow - p) -------------------
__global__ __launch_bounds__(256) void ema_flat_kernel(float* __restrict__ ema, const float* __restrict__ p, long n4, float omd) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4_t e = ((const float4_t*)ema)[i];
    const float4_t q = ((const float4_t*)p)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] -= omd * (e[k] - q[k]);
    ((float4_t*)ema)[i] = e;
  }
}
extern "C" int nk_ema_flat(float* ema, const float* p, long n, float one_minus_decay, void* stream) {
  NK_CHECK_ARG(ema && p && n > 0 && (n & 3) == 0);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(ema_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ema, p, n / 4, one_minus_decay);
  return nk_check_launch("ema_flat_kernel");
}
