// K5 backward: gradient of Residual(PreNorm(LinearAttention)) (reference forward: dquartic/model/unet1d.py:446-496; the
// reference's backward is autograd over those ops).  Same wave-per-row scheme and the same re-associated forward as
// k_linattn.hip (per head: M[d][c] = sum_n K[d][n] xh[c][n], P[c][n] = sum_d M[d][c] Q[d][n], ypre = sum_h W2_h P_h + b with
// W2_h = Wo_h Wv_h), differentiated in that form -- no 32x32 context / value tiles exist in either direction.
//
// One launch does the whole block (rows of up to 64 positions; longer rows: k_la_long.hip + two k_block_bwd).  A workgroup is FOUR
// waves = the FOUR heads of the same units (a unit = one row of 32 / 64 positions, or 32 / n rows of n < 32 positions): every wave
//   (1) does the post-norm backward from the saved pre-norm output (per position, over channels: in-lane + one swap with lane^32;
//       the four waves load the same x / ypre / dy at the same time: HBM sees them once)      -> dYpre ; d g_out, d b_out (head 0's wave)
//   (2) for ITS head, recomputing the forward in registers:
//         dP[c][n] = sum_c' W2[c'][c] dYpre[c'][n]                  VALU, C x C per position
//         dW2[c'][c] += sum_n dYpre[c'][n] P[c][n]                  4x4x1 MFMA, both operands staged [c][n] in LDS
//         dQ[d][n] = sum_c M[d][c] dP[c][n]   ; dK^T[n][d] = sum_c xh[c][n] dM[d][c]      32x32x2 MFMA with K = C
//         dM^T[c][d] = sum_n dP[c][n] Q[d][n]                       4x4x1 MFMA (B = Q^T tile)
//         softmax backward of q (over d, in-lane) and k (over n, in-lane in the K^T orientation)
//         dXh_h = Wq^T dq_raw + Wk^T dk_raw + sum_d K[d][n] dM[d][c]                       4x4x1 MFMA
//         dWq += xh dq_raw^T, dWk += xh dk_raw^T                     4x4x1 MFMA
//       Short rows (n < 32, 32/n rows per wave): the masked quadratic form S^T[n'][n] = sum_d K[d][n'] Q[d][n],
//       R[c][n] = sum_n' xh[c][n'] S^T[n'][n] in place of M / P.
//   (3) publishes dXh_h to an LDS exchange buffer; behind ONE workgroup barrier per unit the last head's wave sums the four heads (in
//       head order) and does the residual + pre-norm backward -> dx (+)= dy + d/dx ; d g_pre.  dXh never goes through HBM.
// (Until round 2 every wave walked its units once per head: x / ypre / dy were loaded and normalised four times by different waves at
// different times, dXh was read-modified-written through HBM per head, and a wave flushed its weight-gradient registers four times --
// 25-40 % of a wave's life at 12 / 16 channels.  Heads across the waves of a block: one flush per wave, a slot per BLOCK instead of per
// wave (4x less slot traffic for the reduce), 10-25 % per launch, tools/probe/la_bwd_time.hip.)
// dWv and dWo follow from the summed dW2 once per layer (k_linattn_dwvo): dWo_h = dW2_h Wv_h^T, dWv_h = Wo_h^T dW2_h.
// All parameter gradients go to the block's partial slot (plain stores, each head's wave its own sections) and are summed by
// k_linattn_dw_reduce_multi in a fixed order: no atomics, bitwise repeatable.  Orientation changes (q -> q^T etc.) go through a
// wave-private 32x33 LDS tile.  Rows of ONE position have a closed form: k_linattn_bwd1.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include <algorithm>
#include <cstdlib>

namespace dq {

__device__ __forceinline__ f32x16 mfma32b(float a, float b, f32x16 c) { return mfma_f32(a, b, c); }
__device__ __forceinline__ float swp32(float v) { return swap_half(v); }
__device__ __forceinline__ void wfence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 32x32 transpose of an accumulator tile through a wave-private LDS tile [32][TRP]: written as 16 dwords per lane (a row's 32 lanes are
// consecutive: conflict-free), read back as FOUR 16-byte reads -- registers 4 q .. 4 q + 3 are rows 8 q + 4 half .. + 3 of the transposed
// tile, consecutive in memory; the row pitch of 36 floats keeps them 16-byte aligned and spreads a 16-lane phase over all 64 banks
// (36 col mod 64 takes every multiple of 4 once).  With pitch 33 the reads were eight ds_read2_b32 per transpose.
constexpr int TRP = 36;
__device__ __forceinline__ f32x16 tr32(f32x16 a, float* tile, int col, int half) {
  wfence();
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[rmap(r, half) * TRP + col] = a[r];
  wfence();
  f32x16 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(tile + col * TRP + 8 * q + 4 * half);
    o[4 * q + 0] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
  }
  return o;
}

struct LinAttnBwdK {
  const float* x; const float* ypre; const float* dy;  // (rows, C, n): block input, saved pre-norm output, d loss / d y
  float* dx;                                            // += d loss / d x (incl. the residual)
  const float* w_qkv; const float* w_out; const float* g_pre; const float* g_out;
  float* part;  // partial slots (per block; per wave for rows of one position): [slot][la_slot(C)] = dWq | dWk (256 C) | dW2 per head (4 C C) | d g_out | d b_out | d g_pre
  int rows; int units_per_wave;  // units per BLOCK in k_linattn_bwd (its four waves share them), per wave in k_linattn_bwd1
  const float* prep;  // nullable: this layer's LA_PREP_FLOATS prepared by launch_linattn_prepare (W2 = 4 C C floats, the bounded-logit flag)
  int dx_store;       // dx is written, not accumulated into (its old contents are not read)
#ifdef DQ_LA_PROBE
  unsigned long long* probe;  // tools/probe/la_bwd_time.hip: [wave][16] shader-clock stamps
#endif
};
#ifdef DQ_LA_PROBE
#define DQ_STAMP(i) do { if (a.probe && (threadIdx.x & 63) == 0) a.probe[(int64_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = clock64(); } while (0)
#else
#define DQ_STAMP(i) do {} while (0)
#endif
constexpr int la_slot(int C) { return 256 * C + 4 * C * C + 3 * C; }  // dWq | dWk (256 C) | dW2 of the four heads (4 C C) | d g_out | d b_out | d g_pre

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products.  Block = lane >> 2; a lane supplies A_blk[i = lane & 3] and
// B_blk[j = lane & 3]; register i of lane (blk, j) receives A_blk[i] * B_blk[j] (mapping measured: tools/probe/mfma4x4.hip).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// sum over the 32 lanes that share (lane >> 5)
// (four DPP adds give every lane its 16-lane row sum, the two rows of a half meet through v_readlane: no LDS round trips)
__device__ __forceinline__ float half_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));  // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));  // row_mirror
  x = __float_as_int(v);
  const float lo = __int_as_float(__builtin_amdgcn_readlane(x, 0)) + __int_as_float(__builtin_amdgcn_readlane(x, 16));
  const float hi = __int_as_float(__builtin_amdgcn_readlane(x, 32)) + __int_as_float(__builtin_amdgcn_readlane(x, 48));
  return (threadIdx.x & 32) ? hi : lo;
}

// Two waves per SIMD (a 256-register budget, enforced through the launch bounds) where the spills that costs stay small: C = 4, and
// C = 8 with rows of 32 / 16 / 8 positions (measured with tools/probe/la_bwd_time.hip, 12,800 rows: <8,16> 183 -> 161 us with 44 B of
// scratch per lane; C = 12 / 16 need 386 / 466 registers and keep one wave per SIMD).  The partner wave covers the latencies the
// one-unit-ahead prefetch was there for, so those variants run without it.
#ifndef DQ_LA_PF4
#define DQ_LA_PF4 0
#endif
// DQ_LA_12TW=1 (build-time experiment): the 12-channel short-row variants at two waves per SIMD -- 256 registers by launch bounds, 79 KB of
// LDS by dropping the second exchange buffer.  Measured: the allocator spills ~130 registers (500 B of scratch per lane) and the launches take
// 1.4-1.5x as long (<12,8> 153 -> 225 us, <12,4> 95 -> 142, <12,2> 69 -> 94 stand-alone).  Two waves per SIMD at 12 / 16 channels needs the
// register diet done by hand (per-head phases that keep fewer tiles live), not a launch bound.
#ifndef DQ_LA_12TW
#define DQ_LA_12TW 0
#endif
constexpr bool la_two_waves(int C, int N) { return C == 4 || (C == 8 && (N == 32 || N == 16 || N == 8)) || (DQ_LA_12TW && C == 12 && N <= 8); }
#ifndef DQ_LA_4_3W
#define DQ_LA_4_3W 0  // build-time experiment: <4,64> at three waves per SIMD (168 registers by launch bound).  Round 3: 63 spilled registers, 281 -> 300 us.  Round 4 (no tile carried across the phases, K^T recomputed in the k phase, x / dy re-read: 24 spilled registers): 278 -> 288 us stand-alone, the step unchanged -- the LDS pipe, not the wave count, is what the transposes and operand reads saturate
#endif
template <int C, int N>
__global__ void __launch_bounds__(256, (DQ_LA_4_3W && C == 4 && N == 64) ? 3 : (la_two_waves(C, N) ? 2 : 1)) k_linattn_bwd(LinAttnBwdK a) {
  static_assert(N >= 2, "rows of one position: k_linattn_bwd1");
  constexpr int NB = N >= 32 ? N / 32 : 1;
  constexpr int RW = N >= 32 ? 1 : 32 / N;
  constexpr int NJ = la_nj(C);  // x registers per lane; register j holds channel la_chan(C, j, half)
  constexpr int SEG = N >= 32 ? 16 : (N >= 8 ? N / 2 : N);
  // rows of 16 positions in the M / P form per ROW (a row = register segment s of k^T), as in the forward.  (Rows of 8: measured slower
  // than the masked quadratic form -- four rows per unit mean 2 x 4 C cross-half sums and four short chains per product: <8,8> 90 -> 98 us,
  // <12,8> 132 -> 200 us per 12,800 rows; <8,16> 160 -> 141 us.)
  constexpr bool SEGM = N == 16;
  constexpr int MS_ROW = C * 32 + 8;        // one row's M as [c][d]; + 8 floats: the rows of a unit start in different LDS banks
  constexpr int MS_FLOATS = SEGM ? (RW * MS_ROW > 2 * C * 32 ? RW * MS_ROW : 2 * C * 32) : 2 * C * 32;  // M | dM ; SEGM: M_s, then dM_s over them
  constexpr bool PARTNER = N >= 8;
  // three waves per SIMD (168 registers): K^T is not carried across the q phase (recomputed in the k phase: 2 NJ MFMAs + 32 exps per unit
  // and head), and the raw x / dy the end of the unit needs are read again instead of held
  constexpr bool LEAN = DQ_LA_4_3W && C == 4 && N == 64;
  constexpr bool SERIAL = la_two_waves(C, N);  // quadratic form: one tile chain at a time (fewer live tiles) instead of interleaved chains
  // C = 4 runs two waves per SIMD instead (the partner wave hides the latency); C = 16 and the 64-position C = 12 variant have
  // no registers to spare
  constexpr bool PREFETCH = DQ_LA_PF4 ? (C <= 8 || (C == 12 && N < 64)) : ((C == 8 && !la_two_waves(C, N)) || (C == 12 && N < 64));
  constexpr int CG = C / 4;      // channel groups of 4 (one 4x4x1 MFMA chain each)
  constexpr int NP = NB * 32;    // positions (lanes x blocks) of one unit
  static_assert(NB <= 2, "rows longer than 64 are not built");
  static_assert(C % 4 == 0, "channel count must be a multiple of 4");

  __shared__ __attribute__((aligned(16))) float wp_lds[2 * 4 * 2 * C * 16];  // [q|k][head][half][c][r] = Wqkv[m*128 + head*32 + rmap(r,half)][c]
  __shared__ __attribute__((aligned(16))) float w2_lds[4 * C * C];           // [head][c'][c] = sum_e Wo[c'][head*32+e] Wv[head*32+e][c]
  __shared__ __attribute__((aligned(16))) float tiles[4][32 * TRP];
  // per wave: xh | dYpre | P (normalised) | dP as [c][n] ; M | dM as [c][d] ; dW2 of the head being flushed [c'][c] and [c][c']
  __shared__ __attribute__((aligned(16))) float stage[4][4 * C * NP + MS_FLOATS + 2 * C * C];
  // d xh of the four heads of one unit, [parity of the unit][head][c][n]: double-buffered so that ONE barrier per unit is enough (a
  // wave that runs ahead writes the other parity; it cannot reach this parity again before the barrier of the unit in between).
  // The 64-position C >= 12 variants have no LDS left for the second buffer and pay a second barrier instead.
  constexpr bool EX2 = !(C >= 12 && N == 64) && !(DQ_LA_12TW && C == 12 && N <= 8);
  __shared__ float exch[(EX2 ? 2 : 1) * 4 * C * NP];
  DQ_STAMP(0);
  {  // (every load of a thread requested before its first store: a loop striding by blockDim.x cannot be unrolled -- 4..16 round trips in a row)
    constexpr int NW = 2 * 4 * 2 * C * 16 / 256;
    float v[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = (i / (32 * C)) & 3, m = i / (128 * C);
      v[u] = a.w_qkv[(m * 128 + hd * 32 + rmap(r, hh)) * C + c];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) wp_lds[u * 256 + (int)threadIdx.x] = v[u];
  }
  if (a.prep) {
    for (int i = threadIdx.x; i < 4 * C * C; i += 256) w2_lds[i] = a.prep[i];
  } else {
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) {
      const int c = i % C, cp = (i / C) % C, hd = i / (C * C);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) s = fmaf(a.w_out[cp * 128 + hd * 32 + e], a.w_qkv[(256 + hd * 32 + e) * C + c], s);
      w2_lds[i] = s;
    }
  }
  __syncthreads();
  DQ_STAMP(1);

  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  float* xs = stage[wv];
  float* dys = xs + C * NP;
  float* ps = dys + C * NP;
  float* dps = ps + C * NP;
  float* ms = dps + C * NP;
  float* dms = SEGM ? ms : ms + C * 32;
  float* w2g = ms + MS_FLOATS;
  const int n_units = (a.rows + RW - 1) / RW;
  const int u0 = blockIdx.x * a.units_per_wave;  // (units per BLOCK here: its four waves walk the same units, one head each)
  if (u0 >= n_units) return;                     // (the whole block)
  const int u1 = min(n_units, u0 + a.units_per_wave);
  const float sqC = sqrtf((float)C);
  const float scale = 0.17677669529663687f;
  const float LOG2E = 1.4426950408889634f;
  const bool bounded = a.prep && a.prep[LA_PREP_BOUNDED] != 0.f;  // (wave-uniform)
  const int rl = N >= 32 ? 0 : col / N;
  // norm gains of this lane's channels, once per wave (a load inside the row loop cannot be hoisted past the loop's stores
  // by the compiler and would sit on the critical path of every row)
  float gpre[NJ], gout[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = la_chan(C, j, half);
    gpre[j] = c < C ? a.g_pre[c] : 0.f;
    gout[j] = c < C ? a.g_out[c] : 0.f;
  }

  // sum_r mfma4(A = src[(4*g + (lane&3)) * pitch + off + rmap(r, half)], B = t[r]) on two interleaved accumulator chains (a
  // dependent 4x4x1 MFMA issues every ~14.5 cycles, independent ones every ~8.5)
  auto chain4 = [&](const float* src, int pitch, int off, int g, const f32x16& t) {
    const float* ar = src + (g * 4 + (lane & 3)) * pitch + off + 4 * half;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
      t0 = mfma4(a4.x, t[q4 * 4 + 0], t0); t1 = mfma4(a4.y, t[q4 * 4 + 1], t1);
      t0 = mfma4(a4.z, t[q4 * 4 + 2], t0); t1 = mfma4(a4.w, t[q4 * 4 + 3], t1);
    }
    return t0 + t1;
  };
  // static-weight variant: A = wp_lds[m][hd][half][c = 4*g + (lane&3)][r] (16 consecutive r)
  auto chainw = [&](int m, int hd, int g, const f32x16& t) {
    const float* wr = wp_lds + (((m * 4 + hd) * 2 + half) * C + g * 4 + (lane & 3)) * 16;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wr + r4 * 4);
      t0 = mfma4(w4.x, t[r4 * 4 + 0], t0); t1 = mfma4(w4.y, t[r4 * 4 + 1], t1);
      t0 = mfma4(w4.z, t[r4 * 4 + 2], t0); t1 = mfma4(w4.w, t[r4 * 4 + 3], t1);
    }
    return t0 + t1;
  };

  {
    const int hd = wv;  // this wave's head
    const bool first = hd == 0, last = hd == 3;
    float wq[NJ], wk[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      const bool ok = c < C;
      wq[j] = ok ? a.w_qkv[(hd * 32 + col) * C + c] * LOG2E : 0.f;   // log2(e) folded in: the softmaxes use exp2 like the forward
      wk[j] = ok ? a.w_qkv[(128 + hd * 32 + col) * C + c] * LOG2E : 0.f;
    }
    // weight-gradient accumulators of this head, 4x4x1 form: register i of group g = channel 4*g + i, lane = (half, d);
    // each lane-half sums its own 16 positions of every 32-block, the halves are added at the flush
    f32x4 gq[CG], gk[CG];
    f32x4 gw2[CG][CG];  // dW2[c' = 4*g1 + i][c = 4*g2 + (lane&3)], partial over the positions of this 4-lane block
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      gq[g] = gk[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g2 = 0; g2 < CG; ++g2) gw2[g][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float nacc0[NJ], nacc1[NJ];  // norm-gain / bias gradient partials: head 0's wave: (d g_out, d b_out); head 3's: (d g_pre, -)
#pragma unroll
    for (int j = 0; j < NJ; ++j) nacc0[j] = nacc1[j] = 0.f;

    // raw operands of one unit (row block): loaded one unit AHEAD when the registers allow it (C <= 8), so that the global
    // latency hides behind the previous unit's MFMAs -- with one wave per SIMD nothing else would cover it
    float px[NB][NJ], pu[NB][NJ], pd[NB][NJ];
    auto load_unit = [&](int u) {
      const int row = u * RW + rl;
      const bool row_ok = row < a.rows;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pos = N >= 32 ? b * 32 + col : col % N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = la_chan(C, j, half);
          const bool ok = row_ok && c < C;
          // (predicated on purpose: the unpredicated form -- clamped row, select afterwards -- was measured: <8,32> 207 -> 290 us, the other
          // instantiations unchanged)
          const int64_t off = ((int64_t)row * C + c) * N + pos;
          px[b][j] = ok ? a.x[off] : 0.f;
          pu[b][j] = ok ? a.ypre[off] : 0.f;
          pd[b][j] = ok ? a.dy[off] : 0.f;
        }
      }
    };
    if (PREFETCH) load_unit(u0);

#pragma unroll 1
    for (int u = u0; u < u1; ++u) {
      const int row = u * RW + rl;
      const bool row_ok = row < a.rows;
      // ---- x, ypre, dy; pre-norm recompute; post-norm backward (same arithmetic as k_block_bwd) -> DY = dYpre
      float Xh[NB][NJ], DY[NB][NJ];
      float cx[NB][NJ], cu[NB][NJ], cd[NB][NJ];
      if (!PREFETCH) load_unit(u);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { cx[b][j] = px[b][j]; cu[b][j] = pu[b][j]; cd[b][j] = pd[b][j]; }
      if (PREFETCH && u + 1 < u1) load_unit(u + 1);
      // what the tail of this iteration reads back -- dx, in the last head's wave when it accumulates -- is requested now, so that
      // its latency hides behind the MFMA work instead of stalling the read-modify-write at the end
      float pdx[NB][NJ];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = la_chan(C, j, half);
          const int64_t off = ((int64_t)row * C + c) * N + (N >= 32 ? b * 32 + col : col % N);
          pdx[b][j] = (PREFETCH && row_ok && c < C && last && !a.dx_store) ? a.dx[off] : 0.f;
        }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float xv[NJ], uv[NJ], dv_[NJ];
        float ssq = 0.f, usq = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          xv[j] = cx[b][j]; uv[j] = cu[b][j]; dv_[j] = cd[b][j];
          ssq = fmaf(xv[j], xv[j], ssq);
          usq = fmaf(uv[j], uv[j], usq);
        }
        ssq += swp32(ssq);
        usq += swp32(usq);
        // (v_sqrt / v_rcp, 1 ulp each, as the forward kernel: a correctly rounded division or square root is ~10 instructions, and this
        // kernel issued 13 of them per unit and head)
        const float inv = rms_inv(ssq, sqC);
        const float unrm = fast_sqrt(usq);
        const float uinv = fast_rcp(fmaxf(unrm, RMS_EPS));
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          Xh[b][j] = xv[j] * inv * gpre[j];
          const float uh = uv[j] * uinv;
          if (first) nacc0[j] = fmaf(dv_[j], uh * sqC, nacc0[j]);  // d g_out
          const float gd = dv_[j] * gout[j] * sqC;
          uv[j] = uh;
          dv_[j] = gd;
          dot = fmaf(gd, uh, dot);
        }
        dot += swp32(dot);
        const bool clamped = unrm < RMS_EPS;  // F.normalize clamps the norm: below eps the map is linear
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          DY[b][j] = clamped ? dv_[j] * uinv : uinv * (dv_[j] - uv[j] * dot);
          if (first) nacc1[j] += DY[b][j];  // d b_out (bias of to_out)
        }
      }
      // stage xh and dYpre as [c][n]: 4x4x1 A operands, and every lane needs ALL channels of dYpre at its position
      wfence();
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = la_chan(C, j, half);
          if (c < C) {
            xs[c * NP + b * 32 + col] = Xh[b][j];
            dys[c * NP + b * 32 + col] = DY[b][j];
          }
        }
      wfence();

      // dP[c][n] = sum_c' W2[hd][c'][c] dYpre[c'][n] at this lane's position of block b (identical in both lane halves)
      auto make_dp = [&](int b, float (&dp)[C]) {
        float dya[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { dya[c] = dys[c * NP + b * 32 + col]; dp[c] = 0.f; }
#pragma unroll
        for (int cp = 0; cp < C; ++cp) {
          const float* w = w2_lds + (hd * C + cp) * C;
#pragma unroll
          for (int c4 = 0; c4 < CG; ++c4) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * c4);
            dp[4 * c4 + 0] = fmaf(w4.x, dya[cp], dp[4 * c4 + 0]); dp[4 * c4 + 1] = fmaf(w4.y, dya[cp], dp[4 * c4 + 1]);
            dp[4 * c4 + 2] = fmaf(w4.z, dya[cp], dp[4 * c4 + 2]); dp[4 * c4 + 3] = fmaf(w4.w, dya[cp], dp[4 * c4 + 3]);
          }
        }
      };
      // this lane's operand slice of a per-position / per-d channel vector: v[c = la_chan(C, j, half)] (zero beyond C)
      auto own = [&](const float (&v)[C], int j) {
        const int c0 = la_chan(C, j, 0), c1 = la_chan(C, j, 1);
        const float lo = c0 < C ? v[c0 < C ? c0 : 0] : 0.f, hi = c1 < C ? v[c1 < C ? c1 : 0] : 0.f;
        return half ? hi : lo;
      };

      f32x4 part[NB][CG];  // d xh partial sums of this lane-half (this head): register i = channel 4*g + i, lane = position
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < CG; ++g) part[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      // part[c][n] += sum_r W[m][hd][rmap(r,half)][c] * t[r][n]: B = the tile register (rows d, col n), A = the weight column
      auto add_dxh = [&](int b, int m, const f32x16& t) {
#pragma unroll
        for (int g = 0; g < CG; ++g) part[b][g] += chainw(m, hd, g, t);
      };
      // dW[c][d] += sum_n xh[c][n] * tt[n][d] over the positions of 32-block b (B = register r of the (rows n, col d) tile)
      auto add_dw = [&](f32x4 (&acc)[CG], int b, const f32x16& tt) {
#pragma unroll
        for (int g = 0; g < CG; ++g) acc[g] += chain4(xs, NP, b * 32, g, tt);
      };
      // dW2[c'][c] += sum_n dYpre[c'][n] p[c][n] over all staged positions: position n = 16 * s + (lane >> 2)
      auto add_dw2 = [&]() {
#pragma unroll
        for (int s = 0; s < NP / 16; ++s) {
          const int n = 16 * s + (lane >> 2);
#pragma unroll
          for (int g1 = 0; g1 < CG; ++g1) {
            const float av = dys[(4 * g1 + (lane & 3)) * NP + n];
#pragma unroll
            for (int g2 = 0; g2 < CG; ++g2) gw2[g1][g2] = mfma4(av, ps[(4 * g2 + (lane & 3)) * NP + n], gw2[g1][g2]);
          }
        }
      };

      // ---- K^T (rows n, col d): exps, then normalised in place (softmax over the positions of each row)
      f32x16 kT[NB];
      float k_m = 0.f, k_rs = 1.f;  // LEAN: shift and 1 / sum of this lane's k-softmax row (d = col), for the recompute in the k phase
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 ak = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) ak = mfma32b(Xh[b][j], wk[j], ak);
        kT[b] = ak;
      }
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += SEG) {
        float ssum = 0.f;
        if (bounded) {  // |logit| <= 64 for every input (LA_PREP_BOUNDED, k_linattn_prepare): no shift by the row maximum, as in the forward
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) {
              const float e = __builtin_amdgcn_exp2f(kT[b][r]);
              kT[b][r] = e;
              ssum += e;
            }
        } else {
          float m = -INFINITY;
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) m = fmaxf(m, kT[b][r]);
          if (PARTNER) m = fmaxf(m, swp32(m));
          if (LEAN) k_m = m;
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) {
              const float e = __builtin_amdgcn_exp2f(kT[b][r] - m);
              kT[b][r] = e;
              ssum += e;
            }
        }
        if (PARTNER) ssum += swp32(ssum);
        const float rs = fast_rcp(ssum);
        if (LEAN) k_rs = rs;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) kT[b][r] *= rs;
      }
      // q (rows d, col n) with its softmax (normalised, incl. 32^-0.5)
      auto make_q = [&](int b) {
        f32x16 q = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) q = mfma32b(wq[j], Xh[b][j], q);
        float ssum = 0.f;
        if (bounded) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            q[r] = __builtin_amdgcn_exp2f(q[r]);
            ssum += q[r];
          }
        } else {
          float m = q[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
          m = fmaxf(m, swp32(m));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            q[r] = __builtin_amdgcn_exp2f(q[r] - m);
            ssum += q[r];
          }
        }
        ssum += swp32(ssum);
        const float qs = scale * fast_rcp(ssum);
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] *= qs;
        return q;
      };
      // the same with the sum over d taken from the C-row side: sum_d q dq = sum_d q sum_c M[d][c] dP[c] = sum_c dP[c] P[c] (P = M^T q, both in every
      // lane of the position already): C FMAs in place of 16 + a cross-half swap
      auto q_softmax_bwd_pp = [&](const f32x16& q, const f32x16& dq, const float (&dP)[C], const float (&P)[C]) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) t = fmaf(dP[c], P[c], t);
        t *= (1.0f / scale);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = q[r] * (dq[r] - t);
        return o;
      };
      auto q_softmax_bwd = [&](const f32x16& q, const f32x16& dq) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t = fmaf(q[r], dq[r], t);
        t = (t + swp32(t)) * (1.0f / scale);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = q[r] * (dq[r] - t);
        return o;
      };

      if (N >= 32) {
        // ================= one row per wave: M / P form =================
        // M[d = col][c] (both lane halves hold the total) and its [c][d] image for the P chains
        float Mr[C];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          f32x4 mt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int b = 0; b < NB; ++b) mt += chain4(xs, NP, b * 32, g, kT[b]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            Mr[g * 4 + i] = mt[i] + swp32(mt[i]);
            if (half == 0) ms[(g * 4 + i) * 32 + col] = Mr[g * 4 + i];
          }
        }
        wfence();
        // (Round 4: no tile is carried from one phase to the next any more.  Q^T of a block was kept for dM = sum_b dP_b Q_b^T and dK^T of
        // both blocks for the k-softmax backward's sum over the row -- 64 registers at two blocks; now dM accumulates inside the block loop
        // (same order of additions: bit-identical) and dK^T is formed twice (the sum, then the gradient: 2 NJ more K = C MFMAs per unit).)
        f32x4 mtd[CG];  // dM^T[c][d] partial sums of this lane-half: sum over the blocks of dP Q^T
#pragma unroll
        for (int g = 0; g < CG; ++g) mtd[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x16 q = make_q(b);
          float dP[C], P[C];
          make_dp(b, dP);
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            const f32x4 pp = chain4(ms, 32, 0, g, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swp32(pp[i]);
          }
          if (half == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) { ps[c * NP + b * 32 + col] = P[c]; dps[c * NP + b * 32 + col] = dP[c]; }
          }
          // dQ[d][n] = sum_c M[d][c] dP[c][n] (K = C on the 32x32x2 pipe), softmax backward, Wq paths
          f32x16 dq = {0};
#pragma unroll
          for (int j = 0; j < NJ; ++j) dq = mfma32b(own(Mr, j), own(dP, j), dq);
          const f32x16 dq_raw = q_softmax_bwd_pp(q, dq, dP, P);
          add_dxh(b, 0, dq_raw);
          add_dw(gq, b, tr32(dq_raw, tile, col, half));
          const f32x16 qT = tr32(q, tile, col, half);  // Q^T (rows n, col d); (the fences inside tr32 also complete this block's ps / dps stores)
#pragma unroll
          for (int g = 0; g < CG; ++g) mtd[g] += chain4(dps, NP, b * 32, g, qT);  // dM^T[c][d] += sum_(n in block b) dP[c][n] Q[d][n]
          if (LEAN) __builtin_amdgcn_sched_barrier(0);  // (the scheduler interleaves the two blocks' streams otherwise: twice the live tiles)
        }
        wfence();  // ps / dps complete
        add_dw2();
        // dM: both halves ; its [c][d] image for the dXh chain
        float dMr[C];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dMr[g * 4 + i] = mtd[g][i] + swp32(mtd[g][i]);
            if (half == 0) dms[(g * 4 + i) * 32 + col] = dMr[g * 4 + i];
          }
        }
        wfence();
        // dK^T[n][d] = sum_c xh[c][n] dM[d][c] ; softmax backward over the positions of the row.  Its sum over the row needs no tile:
        // sum_n K[d][n] dK[d][n] = sum_c dM[d][c] sum_n K[d][n] xh[c][n] = sum_c dM[d][c] M[d][c]  (this lane's d; both halves hold M, dM)
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) dl = fmaf(dMr[c], Mr[c], dl);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f32x16 kTb;
          if (LEAN) {  // K^T of this block again (it was not kept across the q phase: 16 registers per block)
            f32x16 ak = {0};
#pragma unroll
            for (int j = 0; j < NJ; ++j) ak = mfma32b(Xh[b][j], wk[j], ak);
#pragma unroll
            for (int r = 0; r < 16; ++r) ak[r] = __builtin_amdgcn_exp2f(ak[r] - k_m) * k_rs;
            kTb = ak;
          } else {
            kTb = kT[b];
          }
          f32x16 dk_rawT = {0};
#pragma unroll
          for (int j = 0; j < NJ; ++j) dk_rawT = mfma32b(Xh[b][j], own(dMr, j), dk_rawT);
#pragma unroll
          for (int r = 0; r < 16; ++r) dk_rawT[r] = kTb[r] * (dk_rawT[r] - dl);
          add_dw(gk, b, dk_rawT);
          add_dxh(b, 1, tr32(dk_rawT, tile, col, half));
          const f32x16 Kd = tr32(kTb, tile, col, half);  // rows d, col n
#pragma unroll
          for (int g = 0; g < CG; ++g) part[b][g] += chain4(dms, 32, 0, g, Kd);  // dXh[c][n] += sum_d K[d][n] dM[d][c]
          if (LEAN) __builtin_amdgcn_sched_barrier(0);
        }
      } else if (SEGM) {
        // ================= rows of 16 positions (two rows per unit): M / P form per ROW =================
        // Row s of the unit is register segment s of k^T in both lane halves (the forward's mapping), so M_s / dM_s are the 4x4x1 chains
        // restricted to that segment, and a position contracts with the M of its own row (a 4-lane block never straddles rows).  The
        // products with K = C that mix rows -- dQ = M_row(n) dP, dK^T = xh dM_row(n) -- run once per row with the other rows' columns
        // zeroed: 2 NJ + 2 RW NJ 32x32x2 MFMAs per unit instead of the 64 + 4 NJ of the masked S tiles, no S / dS tiles at all.
        auto seg_chain = [&](const float* src, int g, int s, const f32x16& t) {  // sum over the registers of segment s
          const float* ar = src + (g * 4 + (lane & 3)) * NP + 4 * half;
          f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q4 = s * SEG / 4; q4 < (s + 1) * SEG / 4; ++q4) {
            const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
            t0 = mfma4(a4.x, t[q4 * 4 + 0], t0); t1 = mfma4(a4.y, t[q4 * 4 + 1], t1);
            t0 = mfma4(a4.z, t[q4 * 4 + 2], t0); t1 = mfma4(a4.w, t[q4 * 4 + 3], t1);
          }
          return t0 + t1;
        };
        // M_s[d = col][c] = sum_(n in row s) k[d][n] xh[c][n], staged [s][c][d]
#pragma unroll
        for (int s = 0; s < RW; ++s)
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            const f32x4 mt = seg_chain(xs, g, s, kT[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = mt[i] + swp32(mt[i]);
              if (half == 0) ms[s * MS_ROW + (g * 4 + i) * 32 + col] = v;
            }
          }
        wfence();
        const f32x16 q = make_q(0);
        float dP[C], P[C];
        make_dp(0, dP);
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const f32x4 pp = chain4(ms + rl * MS_ROW, 32, 0, g, q);  // P[c][n] = sum_d M_row(n)[d][c] q[d][n]
#pragma unroll
          for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swp32(pp[i]);
        }
        if (half == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) { ps[c * NP + col] = P[c]; dps[c * NP + col] = dP[c]; }
        }
        // dQ[d][n] = sum_c M_row(n)[d][c] dP[c][n]: row by row, the columns of the other rows zeroed
        f32x16 dq = {0};
#pragma unroll
        for (int s = 0; s < RW; ++s)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = la_chan(C, j, half);
            const float mv = c < C ? ms[s * MS_ROW + (c < C ? c : 0) * 32 + col] : 0.f;
            dq = mfma32b(mv, rl == s ? own(dP, j) : 0.f, dq);
          }
        const f32x16 dq_raw = q_softmax_bwd_pp(q, dq, dP, P);
        add_dxh(0, 0, dq_raw);
        add_dw(gq, 0, tr32(dq_raw, tile, col, half));
        const f32x16 qT = tr32(q, tile, col, half);  // rows n, col d
        wfence();  // ps / dps complete; every read of M_s is done (dM_s goes over it)
        add_dw2();
        // dM_s^T[c][d] = sum_(n in row s) dP[c][n] Q[d][n], staged [s][c][d] over M_s
#pragma unroll
        for (int s = 0; s < RW; ++s)
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            const f32x4 mt = seg_chain(dps, g, s, qT);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = mt[i] + swp32(mt[i]);
              if (half == 0) dms[s * MS_ROW + (g * 4 + i) * 32 + col] = v;
            }
          }
        wfence();
        // dK^T[n][d] = sum_c xh[c][n] dM_row(n)[d][c]: row by row, the rows (positions) of the other rows zeroed
        f32x16 dkT = {0};
#pragma unroll
        for (int s = 0; s < RW; ++s)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = la_chan(C, j, half);
            const float dmv = c < C ? dms[s * MS_ROW + (c < C ? c : 0) * 32 + col] : 0.f;
            dkT = mfma32b(rl == s ? Xh[0][j] : 0.f, dmv, dkT);
          }
        f32x16 dk_rawT;  // softmax over the positions of each row: a lane's own register segment (+ lane^32)
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += SEG) {
          float dl = 0.f;
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) dl = fmaf(dkT[r], kT[0][r], dl);
          dl += swp32(dl);
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) dk_rawT[r] = kT[0][r] * (dkT[r] - dl);
        }
        add_dw(gk, 0, dk_rawT);
        add_dxh(0, 1, tr32(dk_rawT, tile, col, half));
        const f32x16 Kd = tr32(kT[0], tile, col, half);  // rows d, col n
#pragma unroll
        for (int g = 0; g < CG; ++g) part[0][g] += chain4(dms + rl * MS_ROW, 32, 0, g, Kd);  // dXh[c][n] += sum_d K[d][n] dM_row(n)[d][c]
      } else {
        // ================= 32/N rows per wave, one block: masked quadratic form =================
        // Two waves per SIMD (SERIAL): one 32x32 tile chain at a time (round 4): S^T -> R, then S -> its dXh chain, then dS^T -> dQ and the whole q side, then dS -> dK^T.
        // Each 16-deep chain runs back to back on one accumulator (this MFMA needs no interleaving with a second chain), the same sums in
        // the same order as before -- but at most four tiles are live (q, K, K^T and the chain's accumulator) instead of nine.
        // One wave per SIMD (12 / 16 channels) keeps the interleaved order: nothing else overlaps a chain's latencies there, and the serial
        // order measured 2-6 % slower (<12,8> 151 -> 159 us, <12,2> 68 -> 73 us stand-alone).
        if constexpr (SERIAL) {
          const f32x16 q = make_q(0);
          const f32x16 Kd = tr32(kT[0], tile, col, half);
          float dR[C];
          {
            f32x16 st = {0};
  #pragma unroll
            for (int r = 0; r < 16; ++r) st = mfma32b(Kd[r], q[r], st);  // S^T : rows n', col n
            st = mask_same_row<N>(st, col, half);
            float R[C];
  #pragma unroll
            for (int g = 0; g < CG; ++g) {
              const f32x4 rr = chain4(xs, NP, 0, g, st);  // R[c][n] = sum_n' xh[c][n'] S^T[n'][n]
  #pragma unroll
              for (int i = 0; i < 4; ++i) R[g * 4 + i] = rr[i] + swp32(rr[i]);
            }
            make_dp(0, dR);
            if (half == 0) {
  #pragma unroll
              for (int c = 0; c < C; ++c) { ps[c * NP + col] = R[c]; dps[c * NP + col] = dR[c]; }
            }
          }
          wfence();
          add_dw2();
          __builtin_amdgcn_sched_barrier(0);
          {
            f32x16 sm = {0};
  #pragma unroll
            for (int r = 0; r < 16; ++r) sm = mfma32b(q[r], Kd[r], sm);  // S   : rows n,  col n'
            sm = mask_same_row<N>(sm, col, half);
            // dXh[c][n'] += sum_n S[n][n'] dR[c][n]
  #pragma unroll
            for (int g = 0; g < CG; ++g) part[0][g] += chain4(dps, NP, 0, g, sm);
          }
          __builtin_amdgcn_sched_barrier(0);
          // dS^T[n'][n] = sum_c xh[c][n'] dR[c][n] (K = C product), masked to pairs of the same row ; dQ ; the q side
          f32x16 qT;
          {
            f32x16 dst = {0};
  #pragma unroll
            for (int j = 0; j < NJ; ++j) dst = mfma32b(Xh[0][j], own(dR, j), dst);
            dst = mask_same_row<N>(dst, col, half);
            f32x16 dq = {0};
  #pragma unroll
            for (int r = 0; r < 16; ++r) dq = mfma32b(kT[0][r], dst[r], dq);    // rows d,  col n
            const f32x16 dq_raw = q_softmax_bwd(q, dq);
            add_dxh(0, 0, dq_raw);
            add_dw(gq, 0, tr32(dq_raw, tile, col, half));
            qT = tr32(q, tile, col, half);  // (only now: one tile fewer is live across the S / dS products)
          }
          __builtin_amdgcn_sched_barrier(0);
          // dS = the transpose of dS^T (the K = C product with the operands swapped) ; dK^T ; the k side
          f32x16 dkT = {0};
          {
            f32x16 dsm = {0};
  #pragma unroll
            for (int j = 0; j < NJ; ++j) dsm = mfma32b(own(dR, j), Xh[0][j], dsm);
            dsm = mask_same_row<N>(dsm, col, half);
  #pragma unroll
            for (int r = 0; r < 16; ++r) dkT = mfma32b(dsm[r], qT[r], dkT);     // rows n', col d
          }
          f32x16 dk_rawT;  // softmax over the positions of each row: a lane's own register segment (+ lane^32)
  #pragma unroll
          for (int s0 = 0; s0 < 16; s0 += SEG) {
            float dl = 0.f;
  #pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) dl = fmaf(dkT[r], kT[0][r], dl);
            if (PARTNER) dl += swp32(dl);
  #pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) dk_rawT[r] = kT[0][r] * (dkT[r] - dl);
          }
          add_dw(gk, 0, dk_rawT);
          add_dxh(0, 1, tr32(dk_rawT, tile, col, half));
      
        } else {
          const f32x16 q = make_q(0);
          const f32x16 qT = tr32(q, tile, col, half);
          const f32x16 Kd = tr32(kT[0], tile, col, half);
          f32x16 st = {0}, sm = {0};
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            st = mfma32b(Kd[r], q[r], st);  // S^T : rows n', col n
            sm = mfma32b(q[r], Kd[r], sm);  // S   : rows n,  col n'
          }
          st = mask_same_row<N>(st, col, half);
          sm = mask_same_row<N>(sm, col, half);
          float dR[C], R[C];
          make_dp(0, dR);
  #pragma unroll
          for (int g = 0; g < CG; ++g) {
            const f32x4 rr = chain4(xs, NP, 0, g, st);  // R[c][n] = sum_n' xh[c][n'] S^T[n'][n]
  #pragma unroll
            for (int i = 0; i < 4; ++i) R[g * 4 + i] = rr[i] + swp32(rr[i]);
          }
          if (half == 0) {
  #pragma unroll
            for (int c = 0; c < C; ++c) { ps[c * NP + col] = R[c]; dps[c * NP + col] = dR[c]; }
          }
          wfence();
          add_dw2();
          // dS^T[n'][n] = sum_c xh[c][n'] dR[c][n] and dS = its transpose (K = C products), masked to pairs of the same row
          f32x16 dst = {0}, dsm = {0};
  #pragma unroll
          for (int j = 0; j < NJ; ++j) {
            dst = mfma32b(Xh[0][j], own(dR, j), dst);
            dsm = mfma32b(own(dR, j), Xh[0][j], dsm);
          }
          dst = mask_same_row<N>(dst, col, half);
          dsm = mask_same_row<N>(dsm, col, half);
          // dXh[c][n'] += sum_n S[n][n'] dR[c][n]
  #pragma unroll
          for (int g = 0; g < CG; ++g) part[0][g] += chain4(dps, NP, 0, g, sm);
          f32x16 dq = {0}, dkT = {0};
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            dq = mfma32b(kT[0][r], dst[r], dq);    // rows d,  col n
            dkT = mfma32b(dsm[r], qT[r], dkT);     // rows n', col d
          }
          const f32x16 dq_raw = q_softmax_bwd(q, dq);
          add_dxh(0, 0, dq_raw);
          add_dw(gq, 0, tr32(dq_raw, tile, col, half));
          f32x16 dk_rawT;  // softmax over the positions of each row: a lane's own register segment (+ lane^32)
  #pragma unroll
          for (int s0 = 0; s0 < 16; s0 += SEG) {
            float dl = 0.f;
  #pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) dl = fmaf(dkT[r], kT[0][r], dl);
            if (PARTNER) dl += swp32(dl);
  #pragma unroll
            for (int r = s0; r < s0 + SEG; ++r) dk_rawT[r] = kT[0][r] * (dkT[r] - dl);
          }
          add_dw(gk, 0, dk_rawT);
          add_dxh(0, 1, tr32(dk_rawT, tile, col, half));
      
        }
      }

      // ---- d xh of this head: both halves' partial sums; each lane publishes its own channels to the block's exchange buffer, and
      // the last head's wave sums the four heads (in head order) behind the barrier.  Nothing of d xh goes through HBM.
      float* ex = exch + ((EX2 ? ((u - u0) & 1) : 0) * 4 + hd) * (C * NP);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float full[C];
#pragma unroll
        for (int c = 0; c < C; ++c) full[c] = part[b][c >> 2][c & 3] + swp32(part[b][c >> 2][c & 3]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = la_chan(C, j, half);
          if (c < C) ex[c * NP + b * 32 + col] = own(full, j);
          // (no registers to hold dx across the MFMA work in these variants: requested here, in front of the barrier)
          if (!PREFETCH && row_ok && c < C && last && !a.dx_store)
            pdx[b][j] = a.dx[((int64_t)row * C + c) * N + (N >= 32 ? b * 32 + col : col % N)];
          if (LEAN && last) {  // the raw x / dy of this unit again (L2-hot: the four waves read them at the top of the unit)
            const int64_t off = ((int64_t)row * C + c) * N + (N >= 32 ? b * 32 + col : col % N);
            cx[b][j] = (row_ok && c < C) ? a.x[off] : 0.f;
            cd[b][j] = (row_ok && c < C) ? a.dy[off] : 0.f;
          }
        }
      }
      lds_barrier();  // (LDS-only: __syncthreads() would also wait for the dx values just requested and for a prefetched next unit)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pos = N >= 32 ? b * 32 + col : col % N;
        float tot[NJ];
        if (last) {
          const float* e0 = exch + (EX2 ? ((u - u0) & 1) : 0) * 4 * (C * NP);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = la_chan(C, j, half);
            const int e = (c < C ? c : 0) * NP + b * 32 + col;
            const float v = ((e0[e] + e0[C * NP + e]) + e0[2 * C * NP + e]) + e0[3 * C * NP + e];
            tot[j] = (row_ok && c < C) ? v : 0.f;
          }
          // ---- residual + pre-norm backward on the completed dXh (own channels c = la_chan(C, j, half)); dx += dy + d/dx
          float xv[NJ];
          float ssq = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            xv[j] = cx[b][j];  // raw x of this unit (zero for masked rows / channels), still in registers
            ssq = fmaf(xv[j], xv[j], ssq);
          }
          ssq += swp32(ssq);
          const float nrm = fast_sqrt(ssq);
          const float inv = fast_rcp(fmaxf(nrm, RMS_EPS));
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float uh = xv[j] * inv;
            nacc0[j] = fmaf(tot[j], uh * sqC, nacc0[j]);  // d g_pre (this wave's nacc0: d g_out lives in head 0's wave)
            const float gd = tot[j] * gpre[j] * sqC;
            xv[j] = uh;
            tot[j] = gd;
            dot = fmaf(gd, uh, dot);
          }
          dot += swp32(dot);
          const bool clamped = nrm < RMS_EPS;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = la_chan(C, j, half);
            if (row_ok && c < C) {
              const int64_t off = ((int64_t)row * C + c) * N + pos;
              const float du = clamped ? tot[j] * inv : inv * (tot[j] - xv[j] * dot);
              a.dx[off] = (pdx[b][j] + cd[b][j]) + du;  // cd: raw dy of this unit, still in registers
            }
          }
        }
      }
      if (!EX2) lds_barrier();  // (single exchange buffer: nobody overwrites it before the last head's wave has read it)
    }

    DQ_STAMP(2);
    // ---- flush this head's gradients to its sections of the BLOCK's partial slot (plain stores; the ordered reduce kernel sums the
    // slots): once per wave -- with the heads walked one after the other by every wave this was four flushes per wave
    float* slot = a.part + (int64_t)blockIdx.x * la_slot(C);
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * g + i;
        // (dq_raw / dk_raw are gradients w.r.t. the natural-log logits: the log2(e) folded into the projection operands only
        // changes how the softmax is evaluated, not the function)
        const float vq = gq[g][i] + swp32(gq[g][i]), vk = gk[g][i] + swp32(gk[g][i]);
        if (half == 0) {  // lane = d
          slot[(hd * 32 + col) * C + c] = vq;
          slot[(128 + hd * 32 + col) * C + c] = vk;
        }
      }
    DQ_STAMP(10);
    // dW2 of this head: sum the 16 position blocks (lanes with equal lane & 3) and publish [c'][c] to the slot.  dWv = Wo^T dW2 and
    // dWo = dW2 Wv^T are linear in dW2, so they are formed ONCE per layer from the slot SUM (k_linattn_dwvo, after the ordered
    // reduce) instead of once per (wave, head) here: that was 2 C loads + C^2 FMAs + 2 C stores per lane and head (3-5 k of a
    // 7-12 k cycle flush at 12 / 16 channels, tools/probe/la_bwd_time.hip) and 256 C of the 515 C floats of a slot.
    // Inside a row of 16 lanes: two DPP rotations (by 4 and by 8 lanes) leave every lane with the sum of its (lane & 3) class;
    // across the four rows: through the wave-private tile (4 x 16 C^2 / 4 floats <= 1024).  The four-step ds_bpermute butterfly
    // this replaces was 4 C^2 dependent LDS round trips per head: 8 us of a 25 us head at 16 channels (tools/probe/la_bwd_time.hip).
    constexpr int NV4 = CG * CG * 16;  // floats one row contributes: [value vi = (g1 * CG + g2) * 4 + i][j = lane & 3]
    wfence();
#pragma unroll
    for (int g1 = 0; g1 < CG; ++g1)
#pragma unroll
      for (int g2 = 0; g2 < CG; ++g2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = gw2[g1][g2][i];
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));  // row_ror:4
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8
          if ((lane & 15) < 4) tile[(lane >> 4) * NV4 + ((g1 * CG + g2) * 4 + i) * 4 + (lane & 3)] = v;
        }
    wfence();
#pragma unroll
    for (int t = 0; t < (NV4 + 63) / 64; ++t) {
      const int e = lane + 64 * t;
      if (e < NV4) {
        const float v = (tile[e] + tile[NV4 + e]) + (tile[2 * NV4 + e] + tile[3 * NV4 + e]);
        const int vi = e >> 2, j = e & 3, i = vi & 3, g2 = (vi >> 2) % CG, g1 = (vi >> 2) / CG;
        w2g[(4 * g1 + i) * C + 4 * g2 + j] = v;
      }
    }
    wfence();
    DQ_STAMP(11);
    for (int i = lane; i < C * C; i += 64) slot[256 * C + hd * C * C + i] = w2g[i];
    constexpr int GB = 256 * C + 4 * C * C;  // [d g_out | d b_out | d g_pre]
    if (first || last) {  // norm gains / bias: sum over the 32 positions-lanes of this half, one lane stores
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = la_chan(C, j, half);
        const float s0 = half_sum(nacc0[j]), s1 = half_sum(nacc1[j]);
        if (col == 0 && c < C) {
          if (first) { slot[GB + c] = s0; slot[GB + C + c] = s1; }  // d g_out, d b_out
          else slot[GB + 2 * C + c] = s0;                          // d g_pre
        }
      }
    }
    DQ_STAMP(3);
  }
}

// Rows of ONE position (the deepest level of the default U-Net): a closed form.  The softmax over a single position is 1 and the q
// softmax sums to 1, so S = 32^-0.5 whatever Wq and Wk are: R = 32^-0.5 xh, dWq = dWk = 0, dXh = 32^-0.5 (sum_h W2_h)^T dYpre and
// dW2_h = dYpre (32^-0.5 xh)^T for every head.  One wave per 32 rows (lane = (row, channel half)), per-WAVE slots in the layout of
// k_linattn_bwd (the q / k sections are written as zeros, the dW2 section four times).
template <int C>
__global__ void __launch_bounds__(256) k_linattn_bwd1(LinAttnBwdK a) {
  constexpr int NJ = la_nj(C), CG = C / 4, NP = 32;
  __shared__ __attribute__((aligned(16))) float w2_lds[4 * C * C];
  __shared__ __attribute__((aligned(16))) float tiles[4][32 * TRP];
  __shared__ __attribute__((aligned(16))) float stage[4][3 * C * NP + C * C];  // per wave: xh | dYpre | 32^-0.5 xh as [c][row] ; dW2
  if (a.prep) {
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) w2_lds[i] = a.prep[i];
  } else {
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) {
      const int c = i % C, cp = (i / C) % C, hd = i / (C * C);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) s = fmaf(a.w_out[cp * 128 + hd * 32 + e], a.w_qkv[(256 + hd * 32 + e) * C + c], s);
      w2_lds[i] = s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += blockDim.x)  // sum_h W2_h into slot 0
    w2_lds[i] = ((w2_lds[i] + w2_lds[C * C + i]) + w2_lds[2 * C * C + i]) + w2_lds[3 * C * C + i];
  __syncthreads();

  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  float* xs = stage[wv];
  float* dys = xs + C * NP;
  float* ps = dys + C * NP;
  float* w2g = ps + C * NP;
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + wv;
  const int n_units = (a.rows + 31) / 32;
  const int u0 = wave_id * a.units_per_wave;
  if (u0 >= n_units) return;
  const int u1 = min(n_units, u0 + a.units_per_wave);
  const float sqC = sqrtf((float)C);
  const float scale = 0.17677669529663687f;
  float gpre[NJ], gout[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = la_chan(C, j, half);
    gpre[j] = c < C ? a.g_pre[c] : 0.f;
    gout[j] = c < C ? a.g_out[c] : 0.f;
  }
  f32x4 gw2[CG][CG];  // dW2[c' = 4*g1 + i][c = 4*g2 + (lane&3)], partial over the rows of this 4-lane block
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int g2 = 0; g2 < CG; ++g2) gw2[g][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
  float nacc0[NJ], nacc1[NJ], nacc2[NJ];  // d g_out, d b_out, d g_pre
#pragma unroll
  for (int j = 0; j < NJ; ++j) nacc0[j] = nacc1[j] = nacc2[j] = 0.f;

#pragma unroll 1
  for (int u = u0; u < u1; ++u) {
    const int row = u * 32 + col;
    const bool row_ok = row < a.rows;
    float xv[NJ], uv[NJ], dv_[NJ], Xh[NJ], DY[NJ], pdx[NJ];
    float ssq = 0.f, usq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      const bool ok = row_ok && c < C;
      const int64_t off = (int64_t)row * C + c;
      xv[j] = ok ? a.x[off] : 0.f;
      uv[j] = ok ? a.ypre[off] : 0.f;
      dv_[j] = ok ? a.dy[off] : 0.f;
      pdx[j] = (ok && !a.dx_store) ? a.dx[off] : 0.f;
      ssq = fmaf(xv[j], xv[j], ssq);
      usq = fmaf(uv[j], uv[j], usq);
    }
    // ---- pre-norm recompute; post-norm backward (same arithmetic as k_block_bwd) -> DY = dYpre
    ssq += swp32(ssq);
    usq += swp32(usq);
    const float nrm = fast_sqrt(ssq);
    const float inv = sqC * fast_rcp(fmaxf(nrm, RMS_EPS));
    const float unrm = fast_sqrt(usq);
    const float uinv = fast_rcp(fmaxf(unrm, RMS_EPS));
    float dot = 0.f;
    float gdv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      Xh[j] = xv[j] * inv * gpre[j];
      const float uh = uv[j] * uinv;
      nacc0[j] = fmaf(dv_[j], uh * sqC, nacc0[j]);  // d g_out
      gdv[j] = dv_[j] * gout[j] * sqC;
      uv[j] = uh;
      dot = fmaf(gdv[j], uh, dot);
    }
    dot += swp32(dot);
    const bool uclamped = unrm < RMS_EPS;  // F.normalize clamps the norm: below eps the map is linear
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      DY[j] = uclamped ? gdv[j] * uinv : uinv * (gdv[j] - uv[j] * dot);
      nacc1[j] += DY[j];  // d b_out (bias of to_out)
    }
    // stage xh, 32^-0.5 xh and dYpre as [c][row]: 4x4x1 operands, and every lane needs ALL channels of dYpre of its row
    wfence();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      if (c < C) {
        xs[c * NP + col] = Xh[j];
        ps[c * NP + col] = scale * Xh[j];
        dys[c * NP + col] = DY[j];
      }
    }
    wfence();
    // dR[c] = sum_c' (sum_h W2_h)[c'][c] dYpre[c'] ; dXh = 32^-0.5 dR
    float dR[C];
    {
      float dya[C];
#pragma unroll
      for (int c = 0; c < C; ++c) { dya[c] = dys[c * NP + col]; dR[c] = 0.f; }
#pragma unroll
      for (int cp = 0; cp < C; ++cp)
#pragma unroll
        for (int c = 0; c < C; ++c) dR[c] = fmaf(w2_lds[cp * C + c], dya[cp], dR[c]);
    }
    // dW2[c'][c] += sum_rows dYpre[c'][row] (32^-0.5 xh)[c][row]: row = 16 * s + (lane >> 2)
#pragma unroll
    for (int s = 0; s < NP / 16; ++s) {
      const int n = 16 * s + (lane >> 2);
#pragma unroll
      for (int g1 = 0; g1 < CG; ++g1) {
        const float av = dys[(4 * g1 + (lane & 3)) * NP + n];
#pragma unroll
        for (int g2 = 0; g2 < CG; ++g2) gw2[g1][g2] = mfma4(av, ps[(4 * g2 + (lane & 3)) * NP + n], gw2[g1][g2]);
      }
    }
    // ---- residual + pre-norm backward on dXh (own channels); dx += dy + d/dx
    const float pinv = fast_rcp(fmaxf(nrm, RMS_EPS));
    float tot[NJ], uh2[NJ];
    float dot2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c0 = la_chan(C, j, 0), c1 = la_chan(C, j, 1);
      const float lo = c0 < C ? dR[c0 < C ? c0 : 0] : 0.f, hi = c1 < C ? dR[c1 < C ? c1 : 0] : 0.f;
      const float dxh = row_ok ? scale * (half ? hi : lo) : 0.f;
      uh2[j] = xv[j] * pinv;
      nacc2[j] = fmaf(dxh, uh2[j] * sqC, nacc2[j]);  // d g_pre
      tot[j] = dxh * gpre[j] * sqC;
      dot2 = fmaf(tot[j], uh2[j], dot2);
    }
    dot2 += swp32(dot2);
    const bool clamped = nrm < RMS_EPS;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      if (row_ok && c < C) {
        const float du = clamped ? tot[j] * pinv : pinv * (tot[j] - uh2[j] * dot2);
        a.dx[(int64_t)row * C + c] = (pdx[j] + dv_[j]) + du;
      }
    }
  }

  // ---- flush: zeros for dWq | dWk, the block sum of dW2 for each of the four heads, the gains
  float* slot = a.part + (int64_t)wave_id * la_slot(C);
  for (int i = lane; i < 256 * C; i += 64) slot[i] = 0.f;
  constexpr int NV4 = CG * CG * 16;
  wfence();
#pragma unroll
  for (int g1 = 0; g1 < CG; ++g1)
#pragma unroll
    for (int g2 = 0; g2 < CG; ++g2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = gw2[g1][g2][i];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));  // row_ror:4
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8
        if ((lane & 15) < 4) tile[(lane >> 4) * NV4 + ((g1 * CG + g2) * 4 + i) * 4 + (lane & 3)] = v;
      }
  wfence();
#pragma unroll
  for (int t = 0; t < (NV4 + 63) / 64; ++t) {
    const int e = lane + 64 * t;
    if (e < NV4) {
      const float v = (tile[e] + tile[NV4 + e]) + (tile[2 * NV4 + e] + tile[3 * NV4 + e]);
      const int vi = e >> 2, j = e & 3, i = vi & 3, g2 = (vi >> 2) % CG, g1 = (vi >> 2) / CG;
      w2g[(4 * g1 + i) * C + 4 * g2 + j] = v;
    }
  }
  wfence();
  for (int i = lane; i < C * C; i += 64) {
    const float v = w2g[i];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) slot[256 * C + hd * C * C + i] = v;
  }
  constexpr int GB = 256 * C + 4 * C * C;  // [d g_out | d b_out | d g_pre]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = la_chan(C, j, half);
    const float s0 = half_sum(nacc0[j]), s1 = half_sum(nacc1[j]), s2 = half_sum(nacc2[j]);
    if (col == 0 && c < C) { slot[GB + c] = s0; slot[GB + C + c] = s1; slot[GB + 2 * C + c] = s2; }
  }
}

#ifndef DQ_LA_BIG_TU
// grad[e] += sum over the wave slots in a fixed order (deterministic).  A block owns 16 consecutive elements x 16 slot
// groups: thread (e, g) sums slots g, g+16, g+32, ... (64-byte segments per group: every fetched sector is fully used, unlike
// a lane-per-slot gather), the 16 group sums meet in LDS.
// Slot layout: dWqkv (384C) | dWo (128C) [| d g_out (C) | d b_out (C) | d g_pre (C) when nelem == 515C]
__global__ void __launch_bounds__(256) k_linattn_dw_reduce(const float* __restrict__ part, int nslots, int C, int nelem,
                                                           float* __restrict__ dw_qkv, float* __restrict__ dw_out,
                                                           float* __restrict__ dg_out, float* __restrict__ db_out,
                                                           float* __restrict__ dg_pre) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem) {
    int b = g;
    // (eight loads in flight, added in the order of the two-at-a-time loop: same sums bit for bit, a quarter of the memory round trips)
    for (; b + 112 < nslots; b += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 16 * u) * nelem + e];
#pragma unroll
      for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
    }
    for (; b + 16 < nslots; b += 32) {
      s0 += part[(int64_t)b * nelem + e];
      s1 += part[(int64_t)(b + 16) * nelem + e];
    }
    if (b < nslots) s0 += part[(int64_t)b * nelem + e];
  }
  red[g][el] = s0 + s1;
  __syncthreads();
  if (g == 0 && e < nelem) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    if (e < 384 * C) dw_qkv[e] += s;
    else if (e < 512 * C) dw_out[e - 384 * C] += s;
    else if (e < 513 * C) dg_out[e - 512 * C] += s;
    else if (e < 514 * C) db_out[e - 513 * C] += s;
    else dg_pre[e - 514 * C] += s;
  }
}
// The slot reduction of the register-resident backward (slot layout la_slot(C)), for several LinearAttention layers at once: block ->
// (item, 16-element group) through a prefix table.  dWq | dWk and the gains go to the gradient buffers (+=); the summed dW2 of the four
// heads goes to the item's w2sum scratch, from which k_linattn_dwvo forms dWv and dWo.
struct LaReduceMulti { LaReduceItem it[LA_REDUCE_MAX]; int first_block[LA_REDUCE_MAX + 1]; int count; };
__global__ void __launch_bounds__(256) k_linattn_dw_reduce_multi(LaReduceMulti m) {
  int i = 0;
  while (i + 1 < m.count && (int)blockIdx.x >= m.first_block[i + 1]) ++i;
  const LaReduceItem& it = m.it[i];
  const int C = it.C, nelem = la_slot(C), nslots = it.nslots;
  const float* __restrict__ part = it.part;
  // 32 consecutive elements x 8 slot groups per block: a half-wave reads one whole 128-byte line of a slot.  (With 16 elements x 16 groups
  // every access was half a line -- the launch reads ~45 MB of slots per train step and ran at the rate of twice that, 28 us.)
  __shared__ float red[8][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = ((int)blockIdx.x - m.first_block[i]) * 32 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem) {
    int b = g;
    for (; b + 56 < nslots; b += 64) {  // eight loads in flight, added in a fixed order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 8 * u) * nelem + e];
#pragma unroll
      for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
    }
    for (; b + 8 < nslots; b += 16) {
      s0 += part[(int64_t)b * nelem + e];
      s1 += part[(int64_t)(b + 8) * nelem + e];
    }
    if (b < nslots) s0 += part[(int64_t)b * nelem + e];
  }
  red[g][el] = s0 + s1;
  __syncthreads();
  if (g == 0 && e < nelem) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][el];
    const int w2b = 256 * C, gb = w2b + 4 * C * C;
    if (e < w2b) it.dw_qkv[e] += s;                 // rows 0..255 of to_qkv: q | k
    else if (e < gb) it.w2sum[e - w2b] = s;         // dW2[head][c'][c]
    else if (e < gb + C) it.dg_out[e - gb] += s;
    else if (e < gb + 2 * C) it.db_out[e - gb - C] += s;
    else it.dg_pre[e - gb - 2 * C] += s;
  }
}
// dWv[e][c] += sum_c' Wo[c'][e] dW2[h(e)][c'][c] ; dWo[c'][e] += sum_c dW2[h(e)][c'][c] Wv[e][c]   (e = head * 32 + d: the 128 value /
// output channels; W2_h = Wo_h Wv_h).  One block per (layer, head); the head's dW2 and weight slices in LDS (as
// one block per layer with every operand read from memory in a runtime-length loop this was 42 us at the end of the backward).
__global__ void __launch_bounds__(256) k_linattn_dwvo(LaReduceMulti m) {
  const LaReduceItem& it = m.it[blockIdx.x];
  const int C = it.C, hd = blockIdx.y;
  __shared__ float w2[16 * 16], wo[16 * 32], wv[32 * 16];  // dW2_h [c'][c] ; Wo[c'][e = hd*32 + d] as [c'][d] ; Wv[e][c] as [d][c]
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) w2[i] = it.w2sum[hd * C * C + i];
  for (int i = threadIdx.x; i < 32 * C; i += blockDim.x) {
    wo[i] = it.w_out[(i >> 5) * 128 + hd * 32 + (i & 31)];
    wv[i] = it.w_qkv[(256 + hd * 32) * C + i];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * C; idx += blockDim.x) {
    const int d = idx / C, c = idx - d * C, e = hd * 32 + d;
    float sv = 0.f, so = 0.f;
    for (int k = 0; k < C; ++k) {
      sv = fmaf(wo[k * 32 + d], w2[k * C + c], sv);   // k = c'
      so = fmaf(w2[c * C + k], wv[d * C + k], so);    // row c' = c of dW2, k = c
    }
    it.dw_qkv[(256 + e) * C + c] += sv;
    it.dw_out[c * 128 + e] += so;
  }
}
#endif  // !DQ_LA_BIG_TU

// The two largest instantiations (C >= 12 with 64-position rows; no BASELINE config uses them) crash this compiler's
// "AMDGPU Rewrite AGPR-Copy-MFMA" pass under -amdgpu-mfma-vgpr-form=1: they live in a second translation unit of this same
// file (-DDQ_LA_BIG_TU, built without that option; see the Makefile).
void launch_linattn_bwd_big(const LinAttnBwdK& kk, int C, int waves, hipStream_t s);

#ifdef DQ_LA_BIG_TU
void launch_linattn_bwd_big(const LinAttnBwdK& kk, int C, int waves, hipStream_t s) {
  if (C == 12) hipLaunchKernelGGL((k_linattn_bwd<12, 64>), dim3(cdiv(waves, 4)), dim3(256), 0, s, kk);
  else hipLaunchKernelGGL((k_linattn_bwd<16, 64>), dim3(cdiv(waves, 4)), dim3(256), 0, s, kk);
}
#else
template <int C, int NN>
static void launch_one(const LinAttnBwdK& kk, int waves, hipStream_t s) {
  if constexpr (C >= 12 && NN == 64) launch_linattn_bwd_big(kk, C, waves, s);
  else hipLaunchKernelGGL((k_linattn_bwd<C, NN>), dim3(cdiv(waves, 4)), dim3(256), 0, s, kk);
}

// workgroups of k_linattn_bwd<C, NN> one resident round holds on this device
template <int C, int NN>
static int la_resident_blocks() {
  if constexpr (C >= 12 && NN == 64) return 256;  // (the second translation unit's kernels: one wave per SIMD)
  else {
    static const int v = [] {
      int occ = 1, dev = 0, cus = 256;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_linattn_bwd<C, NN>, 256, 0) != hipSuccess) occ = 1;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      return std::max(1, occ) * std::max(1, cus);
    }();
    return v;
  }
}

template <int C>
static int linattn_bwd_n(const LinAttnBwdK& k, int n, const LinAttnBwd& g, hipStream_t s) {
  // one resident round of blocks of four waves (= the four heads): 1-3 per CU, as the kernel's registers / LDS allow; a partial slot per
  // block.  Rows of one position: a wave per 32 rows, a slot per wave.
  const int64_t k_part_floats = g.part_floats;
  const int slots_max = (int)((k_part_floats - 4 * C * C) / la_slot(C));
  LinAttnBwdK kk = k;
  int slots = 0;
#define DQ_LB(NN)                                                                                  \
  case NN: {                                                                                       \
    constexpr int RW = NN >= 32 ? 1 : 32 / NN;                                                     \
    const int units = cdiv(k.rows, RW);                                                            \
    const int max_blocks = std::min(la_resident_blocks<C, NN>(), slots_max);                       \
    kk.units_per_wave = std::max(1, cdiv(units, max_blocks));                                      \
    slots = cdiv(units, kk.units_per_wave);                                                        \
    launch_one<C, NN>(kk, 4 * slots, s);                                                           \
    break;                                                                                         \
  }
  switch (n) {
    case 1: {
      const int units = cdiv(k.rows, 32);
      kk.units_per_wave = std::max(1, cdiv(units, std::min(1024, slots_max)));
      slots = cdiv(units, kk.units_per_wave);
      hipLaunchKernelGGL((k_linattn_bwd1<C>), dim3(cdiv(slots, 4)), dim3(256), 0, s, kk);
      break;
    }
    DQ_LB(2) DQ_LB(4) DQ_LB(8) DQ_LB(16) DQ_LB(32) DQ_LB(64)
    default:
      set_error("linattn_bwd: m/z length " + std::to_string(n) + " is not built (powers of two up to 64)");
      return 2;
  }
#undef DQ_LB
  DQ_LAUNCH_CHECK();
  /* (4 C C floats behind the slots hold the summed dW2 between the reduce and k_linattn_dwvo) */
  if (g.defer_reduce) {
    *g.waves_out = slots;
    if (g.w2sum_out) *g.w2sum_out = kk.part + (int64_t)slots * la_slot(C);
    return 0;
  }
  const LaReduceItem it{kk.part, slots, C, g.dw_qkv, g.dw_out, g.dg_out, g.db_out, g.dg_pre, kk.part + (int64_t)slots * la_slot(C),
                        k.w_qkv, k.w_out};
  return launch_linattn_dw_reduce_multi(&it, 1, s);
}

// one slot per wave of a resident round + the summed dW2
int64_t la_part_reserve(int C) { return ((int64_t)(C <= 8 ? 2048 : 1024) * la_slot(C) + 4 * C * C + 63) / 64 * 64; }
int launch_linattn_dw_reduce_multi(const LaReduceItem* items, int count, hipStream_t s) {
  if (count == 0) return 0;
  DQ_REQUIRE(count <= LA_REDUCE_MAX, "linattn dw reduce: too many deferred layers");
  LaReduceMulti m;
  m.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    m.it[i] = items[i];
    m.first_block[i] = blocks;
    blocks += cdiv(la_slot(items[i].C), 32);
  }
  m.first_block[count] = blocks;
  hipLaunchKernelGGL(k_linattn_dw_reduce_multi, dim3(blocks), dim3(256), 0, s, m);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_linattn_dwvo, dim3(count, 4), dim3(256), 0, s, m);
  DQ_LAUNCH_CHECK();
  return 0;
}

// a.f.y is unused; needs: a.ypre (saved pre-norm output), the (rows, C, n) scratch dxh (and dyp for rows longer than 64) and
// the partial-slot scratch
int launch_linattn_bwd(const LinAttnBwd& a, hipStream_t s) {
  DQ_REQUIRE(a.f.x && a.dy && a.dx && a.ypre && a.dyp && a.dxh && a.dw_qkv && a.dw_out && a.db_out && a.dg_pre && a.dg_out,
             "linattn_bwd: missing operand");
  if (a.f.rows == 0) return 0;
  const int C = a.f.C, rows = a.f.rows, n = a.f.n;
  if (a.waves_out) *a.waves_out = 0;
  const bool short_rows = n <= 64 && (n & (n - 1)) == 0;  // (a slot per block; the sweep kernel of longer rows: 512 C floats per wave)
  DQ_REQUIRE(a.part && a.part_floats >= (short_rows ? la_part_reserve(C) : (int64_t)LA_MAX_WAVES * 512 * C),
             "linattn_bwd: partial-sum scratch missing or too small");
  static_assert((int64_t)LA_MAX_WAVES * 512 * 4 >= 2048 * (int64_t)la_slot(4) + 64, "slot scratch: a resident round of slots must fit");
  if (n > 64 || (n & (n - 1)) != 0) {
    // rows of 128 / 256 positions: the sweep kernel between two pointwise norm-backward launches
    // (these launches accumulate into dx: a caller that asked for a plain store gets a cleared dx first)
    DQ_REQUIRE(a.part_floats >= (int64_t)LA_MAX_WAVES * 512 * C, "linattn_bwd: partial-sum scratch too small for the sweep kernel");
    if (a.dx_store)
      if (int rz = launch_zero(a.dx, (int64_t)rows * C * n, s)) return rz;
    BlockBwd b2;  // (1) post-norm backward: dyp = d loss / d ypre, d g_out, d b_out
    b2.u = a.ypre; b2.dy = a.dy; b2.du = a.dyp; b2.C = C; b2.rows = rows; b2.n = n; b2.rows_per_sample = rows;
    b2.g = a.f.g_out; b2.dg = a.dg_out; b2.dbias = a.db_out;
    b2.part = a.part; b2.part_floats = a.part_floats;  // per-block partial sums: the slot scratch is free until the sweep kernel runs
    if (int rc = launch_block_bwd(b2, s)) return rc;
    int waves = 0;
    if (int rc = launch_linattn_bwd_long(a.f.x, a.dyp, a.dxh, a.f.w_qkv, a.f.w_out, a.f.g_pre, a.part, C, rows, n, &waves, s)) return rc;
    hipLaunchKernelGGL(k_linattn_dw_reduce, dim3(cdiv(512 * C, 16)), dim3(256), 0, s, a.part, waves, C, 512 * C, a.dw_qkv, a.dw_out,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DQ_LAUNCH_CHECK();
    // (3) residual + pre-norm backward, accumulated into dx
    if (int r2 = launch_axpy(a.dx, a.dy, (int64_t)rows * C * n, s)) return r2;
    BlockBwd b1;
    b1.u = a.f.x; b1.dy = a.dxh; b1.du = a.dx; b1.C = C; b1.rows = rows; b1.n = n; b1.rows_per_sample = rows;
    b1.g = a.f.g_pre; b1.dg = a.dg_pre; b1.accumulate = 1;
    b1.part = a.part; b1.part_floats = a.part_floats;  // (free again: the slot reduce above has consumed it, in stream order)
    return launch_block_bwd(b1, s);
  }
  // rows of 2 / 4 positions at 8 / 12 / 16 channels: one m/z row per lane column (k_la_rows_bwd.hip), when the layer's prepared weights are at hand
  if (a.f.prep && la_rows_bwd_usable(C, n) && rows >= la_rows_bwd_min_rows() &&
      (((uintptr_t)a.f.x | (uintptr_t)a.ypre | (uintptr_t)a.dy | (uintptr_t)a.dx) & 15) == 0) {
    const int slots_max = (int)std::min<int64_t>((a.part_floats - 4 * C * C) / la_slot(C), C <= 8 ? 2048 : 1024);
    int slots = 0;
    if (int rc = launch_la_rows_bwd(a, slots_max, &slots, s)) return rc;
    if (a.defer_reduce) {
      *a.waves_out = slots;
      if (a.w2sum_out) *a.w2sum_out = a.part + (int64_t)slots * la_slot(C);
      return 0;
    }
    const LaReduceItem it{a.part, slots, C, a.dw_qkv, a.dw_out, a.dg_out, a.db_out, a.dg_pre, a.part + (int64_t)slots * la_slot(C), a.f.w_qkv, a.f.w_out};
    return launch_linattn_dw_reduce_multi(&it, 1, s);
  }
  LinAttnBwdK k;
  k.x = a.f.x; k.ypre = a.ypre; k.dy = a.dy; k.dx = a.dx; k.w_qkv = a.f.w_qkv; k.w_out = a.f.w_out;
  k.g_pre = a.f.g_pre; k.g_out = a.f.g_out; k.part = a.part; k.rows = rows; k.units_per_wave = 1;
  k.prep = a.f.prep; k.dx_store = a.dx_store;
  switch (C) {
    case 4: return linattn_bwd_n<4>(k, n, a, s);
    case 8: return linattn_bwd_n<8>(k, n, a, s);
    case 12: return linattn_bwd_n<12>(k, n, a, s);
    case 16: return linattn_bwd_n<16>(k, n, a, s);
    default: set_error("linattn_bwd: unsupported channel count " + std::to_string(C)); return 2;
  }
}
#endif  // DQ_LA_BIG_TU

}  // namespace dq
