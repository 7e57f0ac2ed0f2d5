// K5: Residual(PreNorm(LinearAttention)) over m/z, forward; 89 % of the reference network's FLOPs.
// Reference arithmetic: dquartic/model/unet1d.py:446-496 (LinearAttention), :143-176 (PreNorm), :64-79 (Residual),
// :113-140 (RMSNorm):
//   xh = rmsnorm(x)*g_pre ; q,k,v = Wqkv xh  (4 heads x 32) ; q = softmax_d(q) * 32^-0.5 ; k = softmax_n(k)
//   ctx[d][e] = sum_n k[d][n] v[e][n] ; out[e][n] = sum_d ctx[d][e] q[d][n] ; y = rmsnorm(Wo out + b)*g_out + x
//
// The value path is LINEAR in xh, and xh has only C <= 16 channels.  Re-associating the three contractions
//   M[d][c]  = sum_n k[d][n] xh[c][n]                      (32 x C per head, instead of ctx: 32 x 32)
//   P[c][n]  = sum_d M[d][c] q[d][n]                       (C x n,           instead of out: 32 x n)
//   y[c'][n] = sum_heads sum_c W2_h[c'][c] P_h[c][n] + b ,  W2_h = Wo_h Wv_h  (C x C per head, formed once per block)
// gives the same function with 32*C instead of 32*32 multiply-adds per (position, d) in the two big products: 4.75x fewer
// FLOPs at C = 4, 2.75x at C = 8 (exact algebra; only the fp32 summation order differs from the reference).
//
// gfx950 design: ONE WAVE owns one m/z row (n >= 32) or 32/n rows (n < 32) and keeps the block in registers.
//   kT[n][d] = mfma32(A = xh, B = Wk)  -> column d on the lane, positions n in the registers  (softmax over n in-lane)
//   q[d][n]  = mfma32(A = Wq, B = xh)  -> position on the lane, d in the registers             (softmax over d in-lane)
//   M^T[c][d] and P[c][n] have C rows: they run on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, no padding of
//   C to 32): B operand = a register of the 32x32 tile, A operand = xh[c][n] resp. M[d][c] staged as [c][..] in LDS.
//   Short rows (n < 32): masked S^T[n'][n] = sum_d k[d][n'] q[d][n] as before, then R[c][n] = sum_n' xh[c][n'] S^T[n'][n]
//   (again C rows, 4x4x1).  Both softmax normalisations are applied to the C-row results (M, P / R), not to the 32-row tiles.
//   The W2 contraction and the post-norm run on the VALU for this lane's own channels; HBM traffic: x in, y out (+ ypre).
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include "dq_probe.h"
#include <algorithm>
#include <cstdlib>
#include <string>

namespace dq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// block = lane >> 2; a lane supplies A_blk[i = lane & 3] and B_blk[j = lane & 3]; register i of lane (blk, j) += A_blk[i] * B_blk[j]
__device__ __forceinline__ f32x4 mfma4f(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ constexpr int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// (channel map of a lane's x registers: la_nj / la_chan, dq_mfma.h)
__device__ __forceinline__ float swap32(float v) { return swap_half(v); }  // v_permlane32_swap (dq_mfma.h)
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- split-bf16 form of the K = C projections (kT = xh^T Wk^T, q = Wq xh): an fp32 value is EXACTLY the sum of three bf16 values
// (H = its top 16 bits, M = the top 16 bits of v - H, L = the rest: 8 + 8 + 8 significant bits), so x w = sum of nine part products; the
// six of them at or above 2^-16 of the largest -- HH, HM, MH, MM, HL, LH -- sit in six K slots of v_mfma_f32_32x32x16_bf16 (every bf16
// product is exact in fp32, the accumulation is fp32), the three dropped ones (M L, L M, L L) are < 2^-21 of |x||w| together in the worst
// case and ~2^-23 typically: a few fp32 roundings (every forward fixture and trajectory tolerance holds unchanged, tests/test_hip_forward.py).  A K = 4 projection is two 32-cycle bf16 MFMAs instead of two 64-cycle fp32 ones, a K = 8 projection three instead of four.
// Slot order inside a lane half: channel j of the half (la_chan) owns slots 6j .. 6j+5 = dwords 3j .. 3j+2:
//   x parts (H, H | M, M | H, L)   weight parts (H, M | H, M | L, H)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ constexpr int la_nu(int C) { return (3 * (C / 2) + 3) / 4; }  // bf16 MFMAs per projection
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { union { u32x4 u; bf16x8 b; } c; c.u = v; return c.b; }
// the three packed dwords of one x value
__device__ __forceinline__ void la_split_x(float v, unsigned& d0, unsigned& d1, unsigned& d2) {
  const unsigned vb = __float_as_uint(v);
  const float r = v - __uint_as_float(vb & 0xffff0000u);          // exact
  const unsigned rb = __float_as_uint(r);
  const float l = r - __uint_as_float(rb & 0xffff0000u);          // exact, <= 8 significant bits
  d0 = __builtin_amdgcn_perm(vb, vb, 0x07060706u);                 // (H, H): bytes 2, 3 of v twice
  d1 = __builtin_amdgcn_perm(rb, rb, 0x07060706u);                 // (M, M)
  d2 = __builtin_amdgcn_perm(__float_as_uint(l), vb, 0x07060302u); // (H, L): low half = top of v, high half = top of l
}
// dword e (0..3) of bf16 MFMA u for lane (col, half) of the operand image of row `o` of Wq | Wk (pre-scaled by log2(e) like the fp32 image)
__device__ __forceinline__ unsigned la_bf16_image_dword(const float* __restrict__ w_qkv, int C, int o, int half, int u, int e) {
  const int dw = 4 * u + e, j = dw / 3, t = dw % 3;
  if (j >= C / 2) return 0u;
  const float w = w_qkv[o * C + la_chan(C, j, half)] * 1.4426950408889634f;
  const unsigned hb = __float_as_uint(w) & 0xffff0000u;
  const float r = w - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r) & 0xffff0000u;
  const unsigned lb = __float_as_uint(r - __uint_as_float(mb));
  return t < 2 ? ((hb >> 16) | mb) : ((lb >> 16) | hb);              // (H, M) twice, then (L, H)
}

template <int C, int N, bool BF>
__global__ void __launch_bounds__(256) k_linattn_fwd(LinAttn a) {
  static_assert(!BF || (C <= 8 && N > 1), "the bf16 operand image of 12 / 16 channels does not fit the static LDS next to the other images");
  constexpr int NU = la_nu(C);                // bf16 MFMAs per projection (BF)
  constexpr int NB = N >= 32 ? N / 32 : 1;    // 32-position blocks per row
  constexpr int RW = N >= 32 ? 1 : 32 / N;    // rows per wave
  constexpr int NJ = la_nj(C);                // x registers per lane (channel = la_chan(C, j, half))
  constexpr int SEG = N >= 32 ? 16 : (N >= 8 ? N / 2 : N);  // registers of one row inside a lane
  constexpr bool PARTNER = N >= 8;            // does lane^32 hold positions of the same row?
  constexpr int CG = C / 4;                   // channel groups of 4 (one 4x4x1 MFMA chain each)
  constexpr int NP = NB * 32;                 // staged positions per wave
  constexpr bool SEGM = N == 8 || N == 16;    // short rows whose M is accumulated per row (register segment) instead of S^T
  static_assert(NB <= 2, "rows longer than 64 take the two-pass path (k_la_long.hip)");
  static_assert(C % 4 == 0, "channel count must be a multiple of 4");

  __shared__ __attribute__((aligned(16))) float w2_lds[4 * C * C];  // [head][c'][c] = sum_e Wo[c'][head*32+e] Wv[head*32+e][c]
  __shared__ __attribute__((aligned(16))) float xs_lds[4][C * NP];  // per wave: xh as [c][n]
  constexpr int MS_ROW = C * 32 + 8;  // row stride of M: + 8 floats so that the rows of a unit start in different LDS banks
  __shared__ __attribute__((aligned(16))) float ms_lds[4][(SEGM ? RW : 1) * MS_ROW];  // per wave: M of the current head as [row][c][d]
  constexpr bool QUAD = N > 1 && N < 32 && !SEGM;  // masked quadratic form (rows of 2 / 4 positions)
  __shared__ float tiles[QUAD ? 4 : 1][QUAD ? 32 * 33 : 1];  // wave-private transpose tiles (quadratic form only)
  float* tile = tiles[QUAD ? (threadIdx.x >> 6) : 0];
  DQ_PSTAMP(300000 + C * 100 + N, 0);
#ifdef DQ_KPROBE  // entry clocks of waves 1..3 (stamps 6..8): how far apart do the waves of a workgroup start?
  if ((threadIdx.x & 63) == 0 && threadIdx.x > 0 && dq_kprobe_want == 300000 + C * 100 + N)
    dq_kprobe_buf[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 16 + 5 + (threadIdx.x >> 6)] = clock64();
#endif
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  const int rl = N >= 32 ? 0 : col / N;
  // ONE unit per wave by default.  A wave can also walk the units unit0, unit0 + ustride, ... of one resident round of workgroups with the
  // NEXT unit's x in flight while the current one is worked on (DQ_LA_FWD_OCC=<workgroups per CU>): in the network at batch 512 a wave spends
  // 13,000 of its unit's 40,000 clocks at the top (the x loads take ~9,000 clocks to come back under that load), yet the walking form is
  // SLOWER -- sampling 1,209 against 1,261 windows/s, as without the prefetch in round 2 (842 against 879): the SIMDs are ~80 % busy issuing
  // (four waves each), the waiting wave costs them nothing, and the hardware's workgroup dispatch balances better than a fixed stride.
  const int unit0 = blockIdx.x * 4 + wv, ustride = gridDim.x * 4;
  // The wave's x and the workgroup's weight images are requested TOGETHER, before anything waits: as "stage the weights in a loop, barrier,
  // then load x" a workgroup spent 9,500 + 5,900 of its 40,400 clocks (<4,64>, batch 512) on three to five memory round trips in a row
  // (the staging loops strode by blockDim.x, so the compiler could not unroll them: load -> store -> load ...).
  float Xn[NB][NJ], gpre_r[NJ], gout_r[NJ], bout_r[NJ];
  // (requested BEHIND the prepared images' loads: the memory counter retires in order, so the wait for the images in front of their LDS
  // stores would otherwise be a wait for x as well)
  auto request_unit = [&](int unit) __attribute__((always_inline)) {  // x of `unit` -> Xn (raw: masked at its first use)
    const int rq = unit * RW + rl;
    const int rc = rq < a.rows ? rq : a.rows - 1;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int pos = N >= 32 ? blk * 32 + col : col % N;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = la_chan(C, j, half);
        // (not predicated: rows beyond the last and the padding channels of the C = 12 register map read a valid element and are zeroed)
        Xn[blk][j] = a.x[((int64_t)rc * C + (c < C ? c : 0)) * N + pos];
      }
    }
  };
  auto request_x = [&]() __attribute__((always_inline)) {
    request_unit(unit0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {  // (gains and bias of this lane's channels: once per wave, requested with the first unit's x)
      const int c = la_chan(C, j, half);
      gpre_r[j] = a.g_pre[c < C ? c : 0];
      gout_r[j] = a.g_out[c < C ? c : 0];
      bout_r[j] = a.b_out[c < C ? c : 0];
    }
  };
  float bflag = 0.f;
  constexpr int WQ = BF ? 2 * 4 * NU * 64 * 4 : 2 * 4 * NJ * 2 * 32;  // dwords of the Wq | Wk operand image (bf16: [q|k][head][u][lane][4])
  __shared__ __attribute__((aligned(16))) float wqk_lds[N > 1 ? (WQ + 1023) / 1024 * 1024 : 4];  // (WQ floats used; whole 16-byte copies per thread)
  if (a.prep) {
    // prepared images (k_linattn_prepare): linear 16-byte copies, one (w2) + one or two (q | k operands) per thread, NO guard -- a guarded
    // copy is a branch, the compiler sinks the load into it behind the x loads, and its wait becomes a wait for x.  (The slot holds 1024 +
    // 4096 floats whatever C is; what lies beyond a layer's own images is copied along and never read.)
    constexpr int NQ = N > 1 ? (WQ / 4 + 255) / 256 : 0;
    const float4 v2 = reinterpret_cast<const float4*>(a.prep)[threadIdx.x];
    float4 vq[NQ > 0 ? NQ : 1];
#pragma unroll
    for (int u = 0; u < NQ; ++u) vq[u] = reinterpret_cast<const float4*>(a.prep + (BF ? LA_PREP_BF16 : 1024))[u * 256 + (int)threadIdx.x];
    bflag = a.prep[LA_PREP_BOUNDED];  // (also in front of x: read behind the barrier it was a full wait again)
    request_x();
    DQ_PSTAMP(300000 + C * 100 + N, 9);
    // (the LOAD above is unguarded: the slot holds 1024 floats whatever C is.  The STORE is unguarded too -- threads beyond the 4 C C floats of W2
    // write into the per-wave M staging, which nothing reads before the barrier below: with `if (tid < C * C) w2_lds[tid] = v2` the compiler sank
    // the load under the branch, BEHIND the x requests, and its wait became s_waitcnt vmcnt(0): a full memory round trip, with the operand-image
    // loads issued only after it -- two round trips in series at the head of every launch (ISA of every instantiation, round 4))
    static_assert(sizeof(ms_lds) >= 64 * sizeof(float4), "the dump area of the W2 copy aliases ms_lds");
    float4* w2dst = (int)threadIdx.x < C * C ? reinterpret_cast<float4*>(w2_lds) + threadIdx.x
                                              : reinterpret_cast<float4*>(&ms_lds[0][0]) + (threadIdx.x & 63);
    *w2dst = v2;
#pragma unroll
    for (int u = 0; u < NQ; ++u) reinterpret_cast<float4*>(wqk_lds)[u * 256 + (int)threadIdx.x] = vq[u];
    DQ_PSTAMP(300000 + C * 100 + N, 10);
  } else {
    request_x();
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) {
      const int c = i % C, cp = (i / C) % C, hd = i / (C * C);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) s = fmaf(a.w_out[cp * 128 + hd * 32 + e], a.w_qkv[(256 + hd * 32 + e) * C + c], s);
      w2_lds[i] = s;
    }
  }
  // MFMA weight operands, laid out [q|k][head][j][half][col] so that a wave reads 2 x 32 consecutive floats; channels beyond C
  // are zero.  The rows carry log2(e): both softmaxes then use exp2 (v_exp_f32) directly, which changes nothing
  // mathematically (softmax(x) = 2^(x*log2e - max) / sum).
  if (N > 1) {
    if (!a.prep && BF) {
      for (int i = threadIdx.x; i < WQ; i += blockDim.x) {
        const int e = i & 3, ln = (i >> 2) & 63, u = (i >> 8) % NU, mh = i / (256 * NU);  // mh = (q | k) * 4 + head
        wqk_lds[i] = __uint_as_float(la_bf16_image_dword(a.w_qkv, C, mh * 32 + (ln & 31), ln >> 5, u, e));
      }
    } else if (!a.prep) {
      for (int i = threadIdx.x; i < WQ; i += blockDim.x) {
        const int cc = i & 31, hh = (i >> 5) & 1, j = (i >> 6) % NJ, hd = (i / (64 * NJ)) & 3, m = i / (256 * NJ);
        const int c = la_chan(C, j, hh);
        wqk_lds[i] = c < C ? a.w_qkv[(m * 128 + hd * 32 + cc) * C + c] * 1.4426950408889634f : 0.f;
      }
    }
  }
  // Without prepared weights (the stand-alone entry point) the block also decides by itself whether the logits are bounded
  // (k_linattn_prepare's criterion, LA_PREP_BOUNDED): thread t = row t of Wq | Wk, the 256 row norms meet in LDS
  float* bnd_lds = &ms_lds[0][0];  // (256 floats of the per-wave M staging, which nothing uses before the barrier below)
  static_assert(sizeof(ms_lds) >= 256 * sizeof(float), "bnd_lds aliases ms_lds");
  float bnd0 = 0.f;
  if (N > 1 && !a.prep) {
    float n2 = 0.f;
    for (int c = 0; c < C; ++c) { const float w = a.w_qkv[threadIdx.x * C + c]; n2 = fmaf(w, w, n2); }
    bnd_lds[threadIdx.x] = n2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) bnd_lds[threadIdx.x] = fmaxf(bnd_lds[threadIdx.x], bnd_lds[threadIdx.x + st]);
      __syncthreads();
    }
    bnd0 = bnd_lds[0];  // (read in front of the barrier below: the staging area is the waves' M buffer afterwards)
  }
  // Workgroup barrier for the LDS images only.  __syncthreads() is a full fence: it waits for vmcnt(0), i.e. for the x loads requested above --
  // the barrier then cost a whole memory round trip (9,500 clocks at batch 512) whether or not anything was staged.
  if (a.prep) lds_barrier();
  else __syncthreads();
  DQ_PSTAMP(300000 + C * 100 + N, 1);

  float* xs = xs_lds[wv];
  float* ms = ms_lds[wv];
  const float sqC = sqrtf((float)C);
  const float scale = 0.17677669529663687f;  // 32^-0.5
  bool bounded;  // (wave-uniform)
  if (a.prep) {
    bounded = __builtin_amdgcn_readfirstlane(__float_as_int(bflag)) != 0;  // (+-0.0f / 1.0f)
  } else if (N > 1) {
    float gm = 0.f;
    for (int c = 0; c < C; ++c) gm = fmaxf(gm, fabsf(a.g_pre[c]));
    bounded = 1.4426950408889634f * sqrtf(bnd0) * sqC * gm <= 64.f;
  } else {
    bounded = false;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const bool ok = la_chan(C, j, half) < C;
    gpre_r[j] = ok ? gpre_r[j] : 0.f;
    gout_r[j] = ok ? gout_r[j] : 0.f;
    bout_r[j] = ok ? bout_r[j] : 0.f;
  }
#pragma unroll 1
  for (int unit = unit0; unit * RW < a.rows; unit += ustride) {
  const int row = unit * RW + rl;
  const bool row_ok = row < a.rows;
  // ---- this unit's x (requested one unit ago); the next unit's is requested now; pre-norm; stage xh as [c][n] for the 4x4x1 A operands
  float X[NB][NJ], Xh[NB][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const bool ok = la_chan(C, j, half) < C;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) X[blk][j] = (row_ok && ok) ? Xn[blk][j] : 0.f;
  }
  if ((unit + ustride) * RW < a.rows) request_unit(unit + ustride);  // (wave-uniform; never taken with one unit per wave)
  wave_fence();                  // the previous unit's readers of xs / ms (this wave's own LDS reads) are done
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) ssq = fmaf(X[blk][j], X[blk][j], ssq);
    ssq += swap32(ssq);
    const float inv = rms_inv(ssq, sqC);  // (v_sqrt + v_rcp, 1 ulp each: the correctly rounded division / square root are ~20 instructions per use)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      Xh[blk][j] = X[blk][j] * inv * gpre_r[j];
      if (c < C) xs[c * NP + blk * 32 + col] = Xh[blk][j];
    }
  }
  wave_fence();
  // split-bf16 operand of this lane's xh values (head-independent: once per unit); it is the A operand of kT and the B operand of q
  u32x4 xp[NB][BF ? NU : 1];
  if (BF) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      unsigned dw[4 * NU];
#pragma unroll
      for (int i = 0; i < 4 * NU; ++i) dw[i] = 0u;
#pragma unroll
      for (int j = 0; j < NJ; ++j) la_split_x(Xh[blk][j], dw[3 * j], dw[3 * j + 1], dw[3 * j + 2]);
#pragma unroll
      for (int u = 0; u < NU; ++u) xp[blk][u] = u32x4{dw[4 * u], dw[4 * u + 1], dw[4 * u + 2], dw[4 * u + 3]};
    }
  }

  DQ_PSTAMP(300000 + C * 100 + N, 2);
  float yown[NB][NJ];  // (Wo out) of this lane's channels c' = la_chan(C, j, half)
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int j = 0; j < NJ; ++j) yown[blk][j] = 0.f;

  // yown[blk][j] += f * sum_c W2[hd][c' = la_chan(C, j, half)][c] * p[c]
  auto add_w2 = [&](int hd, int blk, const float (&p)[C], float f) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cp = la_chan(C, j, half);
      if (cp < C) {
        const float* w = w2_lds + (hd * C + cp) * C;
        float s = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CG; ++c4) {
          const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * c4);
          s = fmaf(w4.x, p[4 * c4 + 0], fmaf(w4.y, p[4 * c4 + 1], fmaf(w4.z, p[4 * c4 + 2], fmaf(w4.w, p[4 * c4 + 3], s))));
        }
        yown[blk][j] = fmaf(f, s, yown[blk][j]);
      }
    }
  };
  // sum_r mfma4(A = src[(4*g + (lane&3)) * pitch + off + rowmap(r, half)], B = t[r]): two interleaved accumulator chains
  auto chain4 = [&](const float* src, int pitch, int off, int g, const f32x16& t) {
    const float* ar = src + (g * 4 + (lane & 3)) * pitch + off + 4 * half;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
      t0 = mfma4f(a4.x, t[q4 * 4 + 0], t0); t1 = mfma4f(a4.y, t[q4 * 4 + 1], t1);
      t0 = mfma4f(a4.z, t[q4 * 4 + 2], t0); t1 = mfma4f(a4.w, t[q4 * 4 + 3], t1);
    }
    return t0 + t1;
  };

#pragma unroll 1
  for (int hd = 0; hd < 4; ++hd) {
    if (N == 1) {
      // softmax over a single position is 1 and the q softmax sums to 1: out = 32^-0.5 * v, i.e. y += 32^-0.5 * W2 xh
      float xf[C];
#pragma unroll
      for (int c = 0; c < C; ++c) xf[c] = xs[c * NP + col];
      add_w2(hd, 0, xf, scale);
      continue;
    }
    // weight operands of this head: lane (col, half) supplies W[o_base + col][la_chan(C, j, half)]
    float wq[BF ? 1 : NJ], wk[BF ? 1 : NJ];
    u32x4 wkb[BF ? NU : 1];
    if (BF) {
#pragma unroll
      for (int u = 0; u < NU; ++u)  // one ds_read_b128 per MFMA and operand, consecutive lanes 16 bytes apart (the q operand: make_q)
        wkb[u] = reinterpret_cast<const u32x4*>(wqk_lds)[((1 * 4 + hd) * NU + u) * 64 + lane];
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        wq[j] = wqk_lds[(((0 * 4 + hd) * NJ + j) * 2 + half) * 32 + col];
        wk[j] = wqk_lds[(((1 * 4 + hd) * NJ + j) * 2 + half) * 32 + col];
      }
    }
    // ---------------- K^T (rows n, col d), softmax over the positions of each row
    f32x16 kT[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      f32x16 ak = {0};
      if (BF) {
#pragma unroll
        for (int u = 0; u < NU; ++u) ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(xp[blk][u]), as_bf16x8(wkb[u]), ak, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ak = mfma32(Xh[blk][j], wk[j], ak);
      }
      kT[blk] = ak;
    }
    float krs = 1.f;  // n >= 32: 1 / sum_n exp(k[d][n]) of this lane's d, applied to M instead of to the 32 x n tile
    float krs_seg[16 / SEG];  // n = 8, 16: the same per row (register segment) of the unit
#pragma unroll
    for (int s = 0; s < 16 / SEG; ++s) krs_seg[s] = 1.f;
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += SEG) {
      float ssum = 0.f;
      if (bounded) {  // |logit| <= 64 whatever the input (see LA_PREP_BOUNDED): 2^logit needs no shift by the row maximum
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) {
            const float e = __builtin_amdgcn_exp2f(kT[blk][r]);  // k rows are pre-scaled by log2(e)
            kT[blk][r] = e;
            ssum += e;
          }
      } else {
        float m = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) m = fmaxf(m, kT[blk][r]);
        if (PARTNER) m = fmaxf(m, swap32(m));
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) {
            const float e = __builtin_amdgcn_exp2f(kT[blk][r] - m);
            kT[blk][r] = e;
            ssum += e;
          }
      }
      if (PARTNER) ssum += swap32(ssum);
      const float rs = fast_rcp(ssum);
      if (N >= 32) krs = rs;
      else if (SEGM) krs_seg[s0 / SEG] = rs;
      else {
#pragma unroll
        for (int r = s0; r < s0 + SEG; ++r) kT[0][r] *= rs;
      }
    }
    // q (rows d, col n): un-normalised exp; its 32^-0.5 / sum factor goes onto the C-row result
    auto make_q = [&](int blk, float& qs) {
      f32x16 q = {0};
      if (BF) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const u32x4 wqb = reinterpret_cast<const u32x4*>(wqk_lds)[((0 * 4 + hd) * NU + u) * 64 + lane];
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wqb), as_bf16x8(xp[blk][u]), q, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) q = mfma32(wq[j], Xh[blk][j], q);
      }
      float ssum = 0.f;
      if (bounded) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          q[r] = __builtin_amdgcn_exp2f(q[r]);  // q rows are pre-scaled by log2(e)
          ssum += q[r];
        }
      } else {
        float m = q[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
        m = fmaxf(m, swap32(m));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          q[r] = __builtin_amdgcn_exp2f(q[r] - m);
          ssum += q[r];
        }
      }
      ssum += swap32(ssum);
      qs = scale * fast_rcp(ssum);
      return q;
    };

    if (N >= 32) {
      // ---------------- M^T[c][d] = sum_n xh[c][n] e_k[n][d] (this lane: d = col, its half's positions), then both halves, / sum
      wave_fence();  // the previous head's reads of ms are done
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        f32x4 mt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) mt += chain4(xs, NP, blk * 32, g, kT[blk]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = (mt[i] + swap32(mt[i])) * krs;
          if (half == 0) ms[(g * 4 + i) * 32 + col] = v;
        }
      }
      wave_fence();
      // ---------------- per block: q ; P[c][n] = sum_d M[d][c] e_q[d][n] ; y += W2 P * (32^-0.5 / sum_d e_q)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        float qs;
        const f32x16 q = make_q(blk, qs);
        float P[C];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const f32x4 pp = chain4(ms, 32, 0, g, q);
#pragma unroll
          for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swap32(pp[i]);
        }
        add_w2(hd, blk, P, qs);
      }
    } else if (SEGM) {
      // ---------------- rows of 8 / 16 positions (4 / 2 rows share this block): M per ROW.  Row s of the unit is register
      // segment s of k^T in both lane halves, so M_s^T[c][d] is the 4x4x1 chain restricted to that segment; a position then
      // contracts with the M of its own row (a 4-lane block never straddles rows).  No S^T tile, no transpose.
      wave_fence();  // the previous head's reads of ms are done
#pragma unroll
      for (int s = 0; s < 16 / SEG; ++s)
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const float* ar = xs + (g * 4 + (lane & 3)) * NP + 4 * half;
          f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q4 = s * SEG / 4; q4 < (s + 1) * SEG / 4; ++q4) {
            const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
            t0 = mfma4f(a4.x, kT[0][q4 * 4 + 0], t0); t1 = mfma4f(a4.y, kT[0][q4 * 4 + 1], t1);
            t0 = mfma4f(a4.z, kT[0][q4 * 4 + 2], t0); t1 = mfma4f(a4.w, kT[0][q4 * 4 + 3], t1);
          }
          const f32x4 mt = t0 + t1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = (mt[i] + swap32(mt[i])) * krs_seg[s];
            if (half == 0) ms[s * MS_ROW + (g * 4 + i) * 32 + col] = v;
          }
        }
      wave_fence();
      float qs;
      const f32x16 q = make_q(0, qs);
      float P[C];
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const f32x4 pp = chain4(ms + rl * MS_ROW, 32, 0, g, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swap32(pp[i]);
      }
      add_w2(hd, 0, P, qs);
    } else {
      // ---------------- rows of 2 / 4 positions (16 / 8 rows share this block): masked S^T[n'][n] = sum_d k[d][n'] e_q[d][n] for
      // pairs of the same m/z row, then R[c][n] = sum_n' xh[c][n'] S^T[n'][n]
      float qs;
      const f32x16 q = make_q(0, qs);
      const f32x16 Kd = transpose_tile(kT[0], tile, col, half);  // rows d, col n'
      f32x16 st = {0};
      st = xty(Kd, q, st);
      st = mask_same_row<N>(st, col, half);
      float R[C];
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const f32x4 rr = chain4(xs, NP, 0, g, st);
#pragma unroll
        for (int i = 0; i < 4; ++i) R[g * 4 + i] = rr[i] + swap32(rr[i]);
      }
      add_w2(hd, 0, R, qs);
    }
  }
  DQ_PSTAMP(300000 + C * 100 + N, 3);

  // ---- bias, post-norm, residual, store (this lane's channels c = la_chan(C, j, half)).  The gains and biases are read before the
  // first store: a load issued after a store waits for it, and the loop below would be 2 NJ serial round trips per block

#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int pos = N >= 32 ? blk * 32 + col : col % N;
    float yv[NJ];
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      yv[j] = c < C ? yown[blk][j] + bout_r[j] : 0.f;
      ssq = fmaf(yv[j], yv[j], ssq);
    }
    ssq += swap32(ssq);
    const float inv = rms_inv(ssq, sqC);  // (v_sqrt + v_rcp, 1 ulp each: the correctly rounded division / square root are ~20 instructions per use)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      if (row_ok && c < C) {
        const int64_t off = ((int64_t)row * C + c) * N + pos;
        if (a.ypre) a.ypre[off] = yv[j];
        a.y[off] = fmaf(yv[j] * gout_r[j], inv, X[blk][j]);
      }
    }
  }
  }  // unit
  DQ_PSTAMP(300000 + C * 100 + N, 4);
}

static int la_num_cus() {
  static const int v = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
  return v;
}
#ifdef DQ_DEV_SWITCHES
#define DQ_LA_FP32_TOO 1
#else
#define DQ_LA_FP32_TOO 0
#endif
// split-bf16 K = C projections for 4 / 8 channels; DQ_LA_PROJ=fp32 in the dev build keeps the fp32 matrix-pipe form of rounds 1-3
bool la_proj_bf16() { return !DQ_DEV_FLAG("DQ_LA_PROJ", 'f'); }
template <int C>
static int linattn_fwd_n(const LinAttn& a, hipStream_t s) {
#define DQ_LA(NN)                                                                      \
  case NN: {                                                                           \
    constexpr int RW = NN >= 32 ? 1 : 32 / NN;                                         \
    /* 8 channels, rows of <= 8 positions: the 24 KB bf16 image costs the fourth workgroup per CU (278 -> 296 us at batch 512): fp32 form */ \
    constexpr bool CAN_BF = C <= 8 && NN > 1 && !(C == 8 && NN <= 8);                  \
    const int units = cdiv(a.rows, RW);                                                \
    const int grid = cdiv(units, 4);  /* one unit per wave: a wave walking several units with the next unit's x in flight was measured slower */ \
    /* (the fp32-projection instantiation of a shape whose default is split-bf16 exists in the dev build only: DQ_LA_PROJ=fp32) */ \
    if (DQ_LA_FP32_TOO && !(CAN_BF && la_proj_bf16()))                                 \
      hipLaunchKernelGGL((k_linattn_fwd<C, NN, CAN_BF && !DQ_LA_FP32_TOO>), dim3(grid), dim3(256), 0, s, a); \
    else                                                                               \
      hipLaunchKernelGGL((k_linattn_fwd<C, NN, CAN_BF>), dim3(grid), dim3(256), 0, s, a); \
    break;                                                                             \
  }
  switch (a.n) {
    DQ_LA(1) DQ_LA(2) DQ_LA(4) DQ_LA(8) DQ_LA(16) DQ_LA(32) DQ_LA(64)
    default:
      set_error("linattn_fwd: m/z length " + std::to_string(a.n) + " is not built (powers of two up to 64)");
      return 2;
  }
#undef DQ_LA
  DQ_LAUNCH_CHECK();
  return 0;
}

// The weights every forward block needs, derived once per launch sequence instead of once per block: block = layer; same
// expressions (and fmaf order) as the in-kernel prologue, so the results are bit-identical.
struct LaPrepMulti { LaPrepItem it[LA_PREP_MAX]; PrepCopy cp[PREP_COPY_MAX]; int count; };
__global__ void __launch_bounds__(256) k_linattn_prepare(LaPrepMulti m) {
  if ((int)blockIdx.x >= m.count) {  // the trailing blocks are plain copies
    const PrepCopy& c = m.cp[blockIdx.x - m.count];
    // (eight loads in flight per thread: as `dst[i] = src[i]` over pointers the compiler cannot tell apart every load waited for the store
    // in front of it -- this launch is the first of every forward and took 16 us)
    for (int base = 0; base < c.n; base += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = base + u * 256 + (int)threadIdx.x; v[u] = c.src[i < c.n ? i : 0]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = base + u * 256 + (int)threadIdx.x; if (i < c.n) c.dst[i] = v[u]; }
    }
    return;
  }
  const LaPrepItem& it = m.it[blockIdx.x];
  const int C = it.C, NJ = la_nj(C);
  {  // W2 (4 C C <= 1024 values: at most four per thread), all computed before the first store
    float sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int ic = i < 4 * C * C ? i : 0;
      const int c = ic % C, cp = (ic / C) % C, hd = ic / (C * C);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) s = fmaf(it.w_out[cp * 128 + hd * 32 + e], it.w_qkv[(256 + hd * 32 + e) * C + c], s);
      sv[u] = s;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = u * 256 + (int)threadIdx.x; if (i < 4 * C * C) it.prep[i] = sv[u]; }
  }
  const int WQ = 2 * 4 * NJ * 2 * 32;  // 512 NJ <= 4096: two to sixteen per thread, requested before the first store
  {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int cc = i & 31, hh = (i >> 5) & 1, j = (i >> 6) % NJ, hd = (i / (64 * NJ)) & 3, mm = (i / (256 * NJ)) & 1;
      const int c = la_chan(C, j, hh);
      v[u] = 0.f;
      if (u * 256 < WQ) v[u] = it.w_qkv[(mm * 128 + hd * 32 + cc) * C + (c < C ? c : 0)];  // (wave-uniform guard)
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int hh = (i >> 5) & 1, j = (i >> 6) % NJ;
      const int c = la_chan(C, j, hh);
      if (i < WQ) it.prep[1024 + i] = c < C ? v[u] * 1.4426950408889634f : 0.f;
    }
  }
  if (C <= 8) {  // split-bf16 operand image of Wq | Wk (k_linattn_fwd<C, N, true>): 2048 la_nu(C) dwords, [q|k][head][u][lane][4]
    const int NU = la_nu(C);
    // thread t owns dword e = t & 3 of lane t >> 2 of every (q|k, head, u) group of 256 dwords: its channel and part are the same in all
    // groups of one u, so the weight column and the split are decoded once per u; eight loads in flight per pass
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
      const int e = (int)threadIdx.x & 3, ln = (int)threadIdx.x >> 2, dw = 4 * u + e, j = dw / 3, t = dw % 3;
      const bool live = j < C / 2;
      const int c = live ? la_chan(C, j, ln >> 5) : 0;
      float w[8];
#pragma unroll
      for (int mh = 0; mh < 8; ++mh) w[mh] = it.w_qkv[(mh * 32 + (ln & 31)) * C + c];
#pragma unroll
      for (int mh = 0; mh < 8; ++mh) {
        const float ws = w[mh] * 1.4426950408889634f;
        const unsigned hb = __float_as_uint(ws) & 0xffff0000u;
        const float r = ws - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r) & 0xffff0000u;
        const unsigned lb = __float_as_uint(r - __uint_as_float(mb));
        const unsigned v = !live ? 0u : (t < 2 ? ((hb >> 16) | mb) : ((lb >> 16) | hb));
        it.prep[LA_PREP_BF16 + (mh * NU + u) * 256 + (int)threadIdx.x] = __uint_as_float(v);
      }
    }
  }
  // LA_PREP_BOUNDED: are the softmax logits bounded for EVERY input?  xh = x / max(|x|, eps) * sqrt(C) * g_pre has |xh| <= sqrt(C)
  // max|g_pre|, so a logit (row r of Wq | Wk, in the log2 domain) is at most log2(e) |W_r| sqrt(C) max|g_pre| in magnitude.  Below 64
  // the kernels evaluate softmax as 2^x / sum 2^x without the shift by the row maximum: the same function (fp32 keeps its relative
  // precision over that range, the sums of <= 64 terms stay below 2^70), ~1/3 fewer VALU instructions per softmax.  Thread t = row t.
  __shared__ float red[256];
  {
    float n2 = 0.f;
    for (int c = 0; c < C; ++c) { const float w = it.w_qkv[threadIdx.x * C + c]; n2 = fmaf(w, w, n2); }
    red[threadIdx.x] = n2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      float gm = 0.f;
      for (int c = 0; c < C; ++c) gm = fmaxf(gm, fabsf(it.g_pre[c]));
      const float bound = 1.4426950408889634f * sqrtf(red[0]) * sqrtf((float)C) * gm;
      it.prep[LA_PREP_BOUNDED] = (bound <= 64.f) ? 1.f : 0.f;  // (a NaN anywhere compares false: the shifted form)
    }
  }
  // operand image of k_la_small (k_la_small.hip): [q | k][head][step][lane] = W[(q|k) head row lane & 31][sm_chan(step, lane >> 5)] log2(e),
  // then [head][step][lane] = W2_head[out channel lane & 31][sm_chan(step, lane >> 5)] (zero above C).  W2 was written to prep[] above by
  // this workgroup (the barriers of the reduction stand between).
  if (C >= 8) {
    const int S = C / 2;
    for (int i = threadIdx.x; i < 12 * S * 64; i += 256) {
      const int l = i & 63, st = (i >> 6) % S, g = i / (64 * S);  // g: 0..3 q heads, 4..7 k heads, 8..11 W2 heads
      const int mrow = l & 31, c = sm_chan(C, st, l >> 5);
      float v = 0.f;
      if (g < 8) v = c < C ? it.w_qkv[((g >> 2) * 128 + (g & 3) * 32 + mrow) * C + c] * 1.4426950408889634f : 0.f;
      else v = (mrow < C && c < C) ? it.prep[((g - 8) * C + mrow) * C + c] : 0.f;
      it.prep[LA_PREP_SMALL + i] = v;
    }
    // operand image of k_la_rows_bwd (k_la_rows_bwd.hip): A operands of v_mfma_f32_16x16x4_f32 -- lane (g, i) supplies A[i][k = g] -- per head and
    // lane contiguous.  CPL = C / 4; channel of (lane group g, register r): CPL g + r; output row i = 4 g'' + r'' of an M = C product is channel
    // CPL g'' + r'' (r'' < CPL), the other rows are padding.
    const int CPL = C / 4;
    constexpr int NR = 4 * 64 * LA_ROWS_LANE_FLOATS / 256;  // 36 values per thread: every load requested before the first store
    float rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int k = i % LA_ROWS_LANE_FLOATS, l = (i / LA_ROWS_LANE_FLOATS) & 63, hd = i / (64 * LA_ROWS_LANE_FLOATS);
      const int g = l >> 4, ii = l & 15;
      const int ci = (ii & 3) < CPL ? CPL * (ii >> 2) + (ii & 3) : -1;  // channel of output row ii of an M = C product
      // groups at fixed 16-byte aligned offsets: [q: 8 | k: 8 | W2^T: 4 | Wq^T: 8 | Wk^T: 8]; one unconditional load per value (a valid
      // address is formed for the padding entries too), the select afterwards
      const float* src = it.w_qkv;
      int off = 0;
      bool ok = false;
      float mul = 1.f;
      if (k < 16) {  // q | k projections: M = head channel 16 t + ii, K-step s = channel CPL g + s ; entry t CPL + s of the group
        const int m = k >> 3, e = k & 7, t = e / CPL, s = e % CPL;
        ok = e < 2 * CPL;
        off = (m * 128 + hd * 32 + 16 * (ok ? t : 0) + ii) * C + CPL * g + s;
        mul = 1.4426950408889634f;
      } else if (k < 20) {  // dZ = W2^T DY: M = channel c(ii) of Z, K-step s = channel c' = CPL g + s of DY
        const int s = k - 16;
        ok = s < CPL && ci >= 0;
        src = it.prep;
        off = ok ? (hd * C + CPL * g + s) * C + ci : 0;
      } else {  // d xh += Wq^T dql / Wk^T dkl: M = channel c(ii), K-step (t, r) = head channel 16 t + 4 g + r (natural-log weights)
        const int e = k - 20, m = e >> 3, t = (e >> 2) & 1, r = e & 3;
        ok = ci >= 0;
        off = (m * 128 + hd * 32 + 16 * t + 4 * g + r) * C + (ok ? ci : 0);
      }
      const float v = src[off];
      rv[u] = ok ? v * mul : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) it.prep[LA_PREP_ROWS + u * 256 + (int)threadIdx.x] = rv[u];
  }
}
int launch_linattn_prepare(const LaPrepItem* items, int count, hipStream_t s, const PrepCopy* copies, int n_copies) {
  if (count + n_copies == 0) return 0;
  DQ_REQUIRE(count <= LA_PREP_MAX && n_copies <= PREP_COPY_MAX, "linattn prepare: too many layers / copies");
  LaPrepMulti m;
  m.count = count;
  for (int i = 0; i < n_copies; ++i) m.cp[i] = copies[i];
  for (int i = 0; i < count; ++i) {
    DQ_REQUIRE(items[i].C % 4 == 0 && items[i].C <= 16, "linattn prepare: unsupported channel count");
    m.it[i] = items[i];
  }
  hipLaunchKernelGGL(k_linattn_prepare, dim3(count + n_copies), dim3(256), 0, s, m);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_linattn_fwd(const LinAttn& a, hipStream_t s) {
  DQ_REQUIRE(a.x && a.y && a.w_qkv && a.w_out && a.b_out && a.g_pre && a.g_out, "linattn_fwd: missing operand");
  if (a.rows == 0) return 0;
  if (a.n > 64 || (a.n & (a.n - 1)) != 0) return launch_linattn_fwd_long(a, s);  // long rows, and lengths that are not a power of two
  if (a.prep && la_small_usable(a.C, a.n) && a.rows >= la_small_min_rows() &&
      (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.ypre) & 15) == 0)  // (its tiles move as 16-byte runs)
    return launch_la_small_fwd(a, s);
  // rows of 2 / 4 positions below that threshold (training batches): one m/z row per lane column (k_la_rows_fwd.hip), wherever the rows
  // backward is in use (the same option switches both: a run with the register-resident forms forced keeps the register forward)
  if (a.prep && la_rows_fwd_usable(a.C, a.n) && a.rows >= la_rows_bwd_min_rows() && (((uintptr_t)a.x | (uintptr_t)a.prep) & 15) == 0)
    return launch_la_rows_fwd(a, s);
  switch (a.C) {
    case 4: return linattn_fwd_n<4>(a, s);
    case 8: return linattn_fwd_n<8>(a, s);
    case 12: return linattn_fwd_n<12>(a, s);
    case 16: return linattn_fwd_n<16>(a, s);
    default: set_error("linattn_fwd: unsupported channel count " + std::to_string(a.C)); return 2;
  }
}

}  // namespace dq
