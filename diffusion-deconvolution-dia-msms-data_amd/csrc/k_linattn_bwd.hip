// K5 backward: gradient of Residual(PreNorm(LinearAttention)) (reference forward: dquartic/model/unet1d.py:446-496; the
// reference's backward is autograd over those ops).  Same wave-per-row, all-in-registers MFMA scheme as the forward
// (k_linattn.hip); derivation and lane-level check: oracle/wave_emu.py::la_bwd_unit.
//
// One launch does the whole block (rows of up to 64 positions; longer rows: k_linattn_long.hip + two k_block_bwd):
//   (1) post-norm backward, recomputed per head from the saved pre-norm output (per position, over channels: in-lane
//       + one swap with lane^32)                                                   -> dYpre ; d g_out, d b_out (head 0 only)
//   (2) everything between xh = rmsnorm(x)*g_pre and Ypre                         -> dXh (accumulated over heads), dWqkv, dWo
//   (3) in the last head's pass: residual + pre-norm backward on the completed dXh -> dx += dy + d/dx ; d g_pre
// All parameter gradients go to this wave's partial slot (plain stores) and are summed by k_linattn_dw_reduce in a fixed
// order: no atomics, bitwise repeatable.
//
// (2): the HEAD loop is the outer loop of a wave and its rows the inner one, so the four per-head weight-gradient
// tiles (dWq, dWk, dWv, dWo as 32x32 f32 MFMA accumulators, rows = channel) stay in registers across the wave's rows and
// are flushed with one atomic per element per (wave, head).  Per head and 32-position block (recomputing the forward):
//   kT, vT, q, v, do = Wo^T dYpre, doT           (projections, K = C)
//   ctx = kT^T vT ; dctx = qT^T doT ; outT = q^T ctx ; dq = ctxT^T do ; dkT = v^T dctxT ; dv = dctx^T K
//   softmax backward of q (over d, in-lane) and of k (over n, in-lane in the kT orientation), with
//   sum_n dK K = rowsum(dctx o ctx) so that no cross-lane reduction over positions is ever needed
//   dW* += XhT^T (.)T ; dXh via VALU from the (rows d/e, col n) tiles with Wqkv pre-permuted in LDS.
// Orientation changes (q -> qT etc.) go through a wave-private 32x33 LDS tile (16 ds_write + 16 ds_read, conflict-free).
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include <cstdlib>

namespace dq {

__device__ __forceinline__ f32x16 mfma32b(float a, float b, f32x16 c) { return mfma_f32(a, b, c); }
__device__ __forceinline__ float swp32(float v) { return swap_half(v); }

// 32x32 transpose of an accumulator tile through a wave-private LDS tile [32][33]
__device__ __forceinline__ f32x16 tr32(f32x16 a, float* tile, int col, int half) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[rmap(r, half) * 33 + col] = a[r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = tile[col * 33 + rmap(r, half)];
  return o;
}

struct LinAttnBwdK {
  const float* x; const float* ypre; const float* dy;  // (rows, C, n): block input, saved pre-norm output, d loss / d y
  float* dx;                                            // += d loss / d x (incl. the residual)
  float* dxh;                                           // scratch (rows, C, n): dXh accumulated over heads 0..2
  const float* w_qkv; const float* w_out; const float* g_pre; const float* g_out;
  float* part;  // per-wave partial slots: [wave][LA_SLOT(C)] = dWqkv (384C) | dWo (128C) | d g_out | d b_out | d g_pre
  int rows; int units_per_wave;
};
constexpr int la_slot(int C) { return 515 * C; }

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products.  Block = lane >> 2; a lane supplies A_blk[i = lane & 3] and
// B_blk[j = lane & 3]; register i of lane (blk, j) receives A_blk[i] * B_blk[j] (mapping measured: tools/probe/mfma4x4.hip).
// With C <= 16 channels the weight-gradient (rows = channel) and dXh (rows = channel) products are exactly this shape: on the
// 32x32x2 form their 4..16 rows were padded to 32.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int C, int N>
__global__ void __launch_bounds__(256) k_linattn_bwd(LinAttnBwdK a) {
  constexpr int NB = N >= 32 ? N / 32 : 1;
  constexpr int RW = N >= 32 ? 1 : 32 / N;
  constexpr int NJ = C <= 8 ? 4 : 8;
  constexpr int SEG = N >= 32 ? 16 : (N >= 8 ? N / 2 : N);
  constexpr bool PARTNER = N >= 8;
  constexpr bool PREFETCH = C <= 8;
  constexpr int CG = C / 4;      // channel groups of 4 (one 4x4x1 MFMA each)
  constexpr int NP = NB * 32;    // positions (lanes x blocks) of one unit
  static_assert(NB <= 2, "rows longer than 64 are not built");
  static_assert(C % 4 == 0, "channel count must be a multiple of 4");

  __shared__ float wp_lds[3 * 4 * 2 * C * 16];  // [q|k|v][head][half][c][r] = Wqkv[m*128 + head*32 + rmap(r,half)][c]
  __shared__ float tiles[4][32 * 33];
  __shared__ __attribute__((aligned(16))) float stage[4][2 * C * NP];  // per wave: xh[c][n] | dYpre[c][n] of the current unit
  for (int i = threadIdx.x; i < 3 * 4 * 2 * C * 16; i += blockDim.x) {
    const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = (i / (32 * C)) & 3, m = i / (128 * C);
    wp_lds[i] = a.w_qkv[(m * 128 + hd * 32 + rmap(r, hh)) * C + c];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  float* xs = stage[wv];
  float* dys = xs + C * NP;
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + wv;
  const int n_units = (a.rows + RW - 1) / RW;
  const int u0 = wave_id * a.units_per_wave;
  if (u0 >= n_units) return;
  const int u1 = min(n_units, u0 + a.units_per_wave);
  const float sqC = sqrtf((float)C);
  const float scale = 0.17677669529663687f;
  const int rl = N >= 32 ? 0 : col / N;
  // norm gains of this lane's channels, once per wave (a load inside the row loop cannot be hoisted past the loop's stores
  // by the compiler and would sit on the critical path of every row)
  // (the two largest variants have no registers to spare for it -- and crash this compiler's AGPR-copy rewrite when pushed)
  constexpr bool HOIST_G = !(C >= 12 && N == 64);
  float gpre[NJ], gout[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = rmap(j, half);
    gpre[j] = (HOIST_G && c < C) ? a.g_pre[c] : 0.f;
    gout[j] = (HOIST_G && c < C) ? a.g_out[c] : 0.f;
  }
  auto g_pre_of = [&](int j) { return HOIST_G ? gpre[j] : (rmap(j, half) < C ? a.g_pre[rmap(j, half)] : 0.f); };
  auto g_out_of = [&](int j) { return HOIST_G ? gout[j] : (rmap(j, half) < C ? a.g_out[rmap(j, half)] : 0.f); };

#pragma unroll 1
  for (int hd = 0; hd < 4; ++hd) {
    float wq[NJ], wk[NJ], wvv[NJ], wo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = rmap(j, half);
      const bool ok = c < C;
      wq[j] = ok ? a.w_qkv[(hd * 32 + col) * C + c] : 0.f;
      wk[j] = ok ? a.w_qkv[(128 + hd * 32 + col) * C + c] : 0.f;
      wvv[j] = ok ? a.w_qkv[(256 + hd * 32 + col) * C + c] : 0.f;
      wo[j] = ok ? a.w_out[c * 128 + hd * 32 + col] : 0.f;
    }
    // weight-gradient accumulators of this head, 4x4x1 form: register i of group cg = channel 4*cg + i, lane = (half, d / e);
    // each lane-half sums its own 16 positions of every 32-block, the halves are added at the flush
    f32x4 gq[CG], gk[CG], gv[CG], go[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) gq[g] = gk[g] = gv[g] = go[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float nacc0[NJ], nacc1[NJ];  // norm-gain / bias gradient partials: head 0: (d g_out, d b_out); head 3: (d g_pre, -)
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
          const int c = rmap(j, half);
          const bool ok = row_ok && c < C;
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
      // what the tail of this iteration reads back -- dXh of the earlier heads, and dx in the last head's pass -- is requested
      // now, so that its latency hides behind the MFMA work instead of stalling the read-modify-write at the end
      float pdxh[NB][NJ], pdx[NB][NJ];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = rmap(j, half);
          const int64_t off = ((int64_t)row * C + c) * N + (N >= 32 ? b * 32 + col : col % N);
          const bool ok = row_ok && c < C;
          pdxh[b][j] = (PREFETCH && ok && hd > 0) ? a.dxh[off] : 0.f;
          pdx[b][j] = (PREFETCH && ok && hd == 3) ? a.dx[off] : 0.f;
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
        const float inv = sqC / fmaxf(sqrtf(ssq), RMS_EPS);
        const float unrm = sqrtf(usq);
        const float uinv = 1.0f / fmaxf(unrm, RMS_EPS);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          Xh[b][j] = xv[j] * inv * g_pre_of(j);
          const float uh = uv[j] * uinv;
          if (hd == 0) nacc0[j] = fmaf(dv_[j], uh * sqC, nacc0[j]);  // d g_out
          const float gd = dv_[j] * g_out_of(j) * sqC;
          uv[j] = uh;
          dv_[j] = gd;
          dot = fmaf(gd, uh, dot);
        }
        dot += swp32(dot);
        const bool clamped = unrm < RMS_EPS;  // F.normalize clamps the norm: below eps the map is linear
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          DY[b][j] = clamped ? dv_[j] * uinv : uinv * (dv_[j] - uv[j] * dot);
          if (hd == 0) nacc1[j] += DY[b][j];  // d b_out (bias of to_out)
        }
      }
      // stage xh and dYpre as [c][n] for the 4x4x1 A operands (every lane needs 4 channels of OTHER lanes' positions)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = rmap(j, half);
          if (c < C) {
            xs[c * NP + b * 32 + col] = Xh[b][j];
            dys[c * NP + b * 32 + col] = DY[b][j];
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      f32x4 part[NB][CG];  // d xh partial sums of this lane-half (this head): register i = channel 4*cg + i, lane = position
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < CG; ++g) part[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};

      // part[c][n] += sum_r W[m][hd][rmap(r,half)][c] * t[r][n]: B = the tile register (rows d, col n), A = the weight column
      // W[..][c = 4*cg + (lane & 3)] read from LDS (16 consecutive r = 4 ds_read_b128)
      auto add_dxh = [&](int b, int m, const f32x16& t) {
        const float* wl = wp_lds + ((m * 4 + hd) * 2 + half) * C * 16;
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const float* wr = wl + (g * 4 + (lane & 3)) * 16;
          // two interleaved chains: a dependent 4x4x1 MFMA issues every ~14.5 cycles, independent ones every ~8.5
          f32x4 acc = part[b][g], acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + r4 * 4);
            acc = mfma4(w4.x, t[r4 * 4 + 0], acc); acc2 = mfma4(w4.y, t[r4 * 4 + 1], acc2);
            acc = mfma4(w4.z, t[r4 * 4 + 2], acc); acc2 = mfma4(w4.w, t[r4 * 4 + 3], acc2);
          }
          part[b][g] = acc + acc2;
        }
      };
      // dW[c][d] += sum_n src[c][n] * tt[n][d] over the positions of 32-block b: B = register r of the (rows n, col d) tile,
      // A = src[c = 4*cg + (lane & 3)][b*32 + rmap(r, half)] from the staged copy (4 ds_read_b128 per group)
      auto add_dw = [&](f32x4 (&acc)[CG], const float* src, int b, const f32x16& tt) {
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const float* ar = src + (g * 4 + (lane & 3)) * NP + b * 32 + 4 * half;
          f32x4 t = acc[g], t2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
            t = mfma4(a4.x, tt[q4 * 4 + 0], t); t2 = mfma4(a4.y, tt[q4 * 4 + 1], t2);
            t = mfma4(a4.z, tt[q4 * 4 + 2], t); t2 = mfma4(a4.w, tt[q4 * 4 + 3], t2);
          }
          acc[g] = t + t2;
        }
      };

      // ---- K^T (normalised over the positions of each row) and V^T
      f32x16 kT[NB], vT[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 ak = {0}, av = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          ak = mfma32b(Xh[b][j], wk[j], ak);
          av = mfma32b(Xh[b][j], wvv[j], av);
        }
        kT[b] = ak;
        vT[b] = av;
      }
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += SEG) {
        float m = -INFINITY;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) m = fmaxf(m, kT[b][r]);
        if (PARTNER) m = fmaxf(m, swp32(m));
        float ssum = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) {
            const float e = __expf(kT[b][r] - m);
            kT[b][r] = e;
            ssum += e;
          }
        if (PARTNER) ssum += swp32(ssum);
        const float rs = 1.0f / ssum;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) kT[b][r] *= rs;
      }

      // q (rows d, col n) with its softmax; returns the tile
      auto make_q = [&](int b) {
        f32x16 q = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) q = mfma32b(wq[j], Xh[b][j], q);
        float m = q[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
        m = fmaxf(m, swp32(m));
        float ssum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          q[r] = __expf(q[r] - m);
          ssum += q[r];
        }
        ssum += swp32(ssum);
        const float qs = scale / ssum;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] *= qs;
        return q;
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
        // ================= one row per wave, NB blocks: phased to keep few tiles live =================
        f32x16 ctx = {0};
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) ctx = mfma32b(kT[b][r], vT[b][r], ctx);
        const f32x16 ctxT = tr32(ctx, tile, col, half);
        f32x16 dctx = {0};
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x16 q = make_q(b);
          f32x16 dO = {0}, dOT = {0};
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            dO = mfma32b(wo[j], DY[b][j], dO);
            dOT = mfma32b(DY[b][j], wo[j], dOT);
          }
          const f32x16 qT = tr32(q, tile, col, half);
#pragma unroll
          for (int r = 0; r < 16; ++r) dctx = mfma32b(qT[r], dOT[r], dctx);
          f32x16 outT = {0}, dq = {0};
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            outT = mfma32b(q[r], ctx[r], outT);
            dq = mfma32b(ctxT[r], dO[r], dq);
          }
          add_dw(go, dys, b, outT);
          const f32x16 dq_raw = q_softmax_bwd(q, dq);
          add_dxh(b, 0, dq_raw);
          add_dw(gq, xs, b, tr32(dq_raw, tile, col, half));
        }
        const f32x16 dctxT = tr32(dctx, tile, col, half);
        float delta = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) delta = fmaf(dctxT[r], ctxT[r], delta);
        delta += swp32(delta);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f32x16 v = {0};
#pragma unroll
          for (int j = 0; j < NJ; ++j) v = mfma32b(wvv[j], Xh[b][j], v);
          f32x16 dkT = {0};
#pragma unroll
          for (int r = 0; r < 16; ++r) dkT = mfma32b(v[r], dctxT[r], dkT);
          f32x16 dk_rawT;
#pragma unroll
          for (int r = 0; r < 16; ++r) dk_rawT[r] = kT[b][r] * (dkT[r] - delta);
          add_dw(gk, xs, b, dk_rawT);
          add_dxh(b, 1, tr32(dk_rawT, tile, col, half));
          const f32x16 Kd = tr32(kT[b], tile, col, half);
          f32x16 dv = {0};
#pragma unroll
          for (int r = 0; r < 16; ++r) dv = mfma32b(dctx[r], Kd[r], dv);
          add_dxh(b, 2, dv);
          add_dw(gv, xs, b, tr32(dv, tile, col, half));
        }
      } else {
        // ================= 32/N rows per wave, one block: per-row ctx / dctx =================
        const f32x16 q = make_q(0);
        const f32x16 qT = tr32(q, tile, col, half);
        f32x16 dO = {0}, dOT = {0}, v = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          dO = mfma32b(wo[j], DY[0][j], dO);
          dOT = mfma32b(DY[0][j], wo[j], dOT);
          v = mfma32b(wvv[j], Xh[0][j], v);
        }
        const f32x16 Kd = tr32(kT[0], tile, col, half);
        // "quadratic" form (oracle/wave_emu.py::la_bwd_unit_quad): S[n][n'] = sum_d q[d][n] K[d][n'] restricted to pairs of
        // the same m/z row replaces the per-row ctx tiles -- no loop over the 32/N rows, no 1/RW-utilised MFMAs.
        f32x16 st = {0}, sm = {0}, dst = {0}, dsm = {0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st = mfma32b(Kd[r], q[r], st);     // S^T : rows n', col n
          sm = mfma32b(q[r], Kd[r], sm);     // S   : rows n,  col n'
          dst = mfma32b(v[r], dO[r], dst);   // dS^T: rows n', col n
          dsm = mfma32b(dO[r], v[r], dsm);   // dS  : rows n,  col n'
        }
        st = mask_same_row<N>(st, col, half);
        sm = mask_same_row<N>(sm, col, half);
        dst = mask_same_row<N>(dst, col, half);
        dsm = mask_same_row<N>(dsm, col, half);
        f32x16 outT = {0}, dvT = {0}, dv = {0}, dq = {0}, dkT = {0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          outT = mfma32b(st[r], vT[0][r], outT);  // rows n,  col e
          dvT = mfma32b(sm[r], dOT[r], dvT);      // rows n', col e
          dv = mfma32b(dOT[r], sm[r], dv);        // rows e,  col n'
          dq = mfma32b(kT[0][r], dst[r], dq);     // rows d,  col n
          dkT = mfma32b(dsm[r], qT[r], dkT);      // rows n', col d
        }
        add_dw(go, dys, 0, outT);
        const f32x16 dq_raw = q_softmax_bwd(q, dq);
        add_dxh(0, 0, dq_raw);
        const f32x16 dq_rawT = tr32(dq_raw, tile, col, half);
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
        add_dw(gq, xs, 0, dq_rawT);
        add_dw(gk, xs, 0, dk_rawT);
        add_dw(gv, xs, 0, dvT);
        add_dxh(0, 1, tr32(dk_rawT, tile, col, half));
        add_dxh(0, 2, dv);
      }

      // ---- d xh of this head: both halves' partial sums, then each lane keeps its own channels
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pos = N >= 32 ? b * 32 + col : col % N;
        float full[C], tot[NJ];
#pragma unroll
        for (int c = 0; c < C; ++c) full[c] = part[b][c >> 2][c & 3] + swp32(part[b][c >> 2][c & 3]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c0 = rmap(j, 0), c1 = c0 + 4;
          float lo = 0.f, hi = 0.f;
          if (c0 < C) lo = full[c0 < C ? c0 : 0];
          if (c1 < C) hi = full[c1 < C ? c1 : 0];
          const int c = c0 + 4 * half;
          const float val = half ? hi : lo;
          lo = 0.f;  // from here on: this lane's total dXh of channel c (heads 0..3), only formed in the last head's pass
          if (row_ok && c < C) {
            float* dst = a.dxh + ((int64_t)row * C + c) * N + pos;
            // head 0 initialises, heads 1, 2 accumulate (same lane, same address); C > 8 has no registers for the prefetch
            const float prev = PREFETCH ? pdxh[b][j] : (hd > 0 ? *dst : 0.f);
            if (hd < 3) *dst = prev + val;
            else lo = prev + val;
          }
          tot[j] = lo;
        }
        if (hd == 3) {
          // ---- residual + pre-norm backward on the completed dXh (own channels c = rmap(j, half)); dx += dy + d/dx
          float xv[NJ];
          float ssq = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = rmap(j, half);
            xv[j] = (c < C) ? cx[b][j] : 0.f;  // raw x of this unit (zero for masked rows / channels), still in registers
            ssq = fmaf(xv[j], xv[j], ssq);
          }
          ssq += swp32(ssq);
          const float nrm = sqrtf(ssq);
          const float inv = 1.0f / fmaxf(nrm, RMS_EPS);
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float uh = xv[j] * inv;
            nacc0[j] = fmaf(tot[j], uh * sqC, nacc0[j]);  // d g_pre
            const float gd = tot[j] * g_pre_of(j) * sqC;
            xv[j] = uh;
            tot[j] = gd;
            dot = fmaf(gd, uh, dot);
          }
          dot += swp32(dot);
          const bool clamped = nrm < RMS_EPS;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = rmap(j, half);
            if (row_ok && c < C) {
              const int64_t off = ((int64_t)row * C + c) * N + pos;
              const float du = clamped ? tot[j] * inv : inv * (tot[j] - xv[j] * dot);
              a.dx[off] = ((PREFETCH ? pdx[b][j] : a.dx[off]) + cd[b][j]) + du;  // cd: raw dy of this unit, still in registers
            }
          }
        }
      }
    }

    // ---- flush this head's weight gradients to this wave's partial slot (plain stores; float atomics at this access
    // shape -- one dword per lane, lanes C floats apart -- run ~17x below the store rate and made the flush the
    // kernel's critical path).  Register r holds channel rmap(r, half), lane column = d / e.
    float* slot = a.part + (int64_t)wave_id * la_slot(C);
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * g + i;
        const float vq = gq[g][i] + swp32(gq[g][i]), vk = gk[g][i] + swp32(gk[g][i]);
        const float vv = gv[g][i] + swp32(gv[g][i]), vo = go[g][i] + swp32(go[g][i]);
        if (half == 0) {  // lane = d / e
          slot[(hd * 32 + col) * C + c] = vq;
          slot[(128 + hd * 32 + col) * C + c] = vk;
          slot[(256 + hd * 32 + col) * C + c] = vv;
          slot[384 * C + c * 128 + hd * 32 + col] = vo;
        }
      }
    if (hd == 0 || hd == 3) {  // norm gains / bias: sum over the 32 positions-lanes of this half, one lane stores
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = rmap(j, half);
        const float s0 = half_sum(nacc0[j]), s1 = half_sum(nacc1[j]);
        if (col == 0 && c < C) {
          if (hd == 0) { slot[512 * C + c] = s0; slot[513 * C + c] = s1; }
          else slot[514 * C + c] = s0;
        }
      }
    }
  }
}

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

template <int C>
static int linattn_bwd_n(const LinAttnBwdK& k, int n, const LinAttnBwd& g, hipStream_t s) {
#define DQ_LB(NN)                                                                                  \
  case NN: {                                                                                       \
    constexpr int RW = NN >= 32 ? 1 : 32 / NN;                                                     \
    const int units = cdiv(k.rows, RW);                                                            \
    LinAttnBwdK kk = k;                                                                            \
    /* 466 registers => one wave per SIMD, 1024 resident waves: size the grid to ONE resident round */ \
    kk.units_per_wave = std::max(1, cdiv(units, 1024));                                            \
    const int waves = cdiv(units, kk.units_per_wave);                                              \
    hipLaunchKernelGGL((k_linattn_bwd<C, NN>), dim3(cdiv(waves, 4)), dim3(256), 0, s, kk);         \
    hipLaunchKernelGGL(k_linattn_dw_reduce, dim3(cdiv(la_slot(C), 16)), dim3(256), 0, s, kk.part, waves, C, la_slot(C), g.dw_qkv, \
                       g.dw_out, g.dg_out, g.db_out, g.dg_pre);                                    \
    break;                                                                                         \
  }
  switch (n) {
    DQ_LB(1) DQ_LB(2) DQ_LB(4) DQ_LB(8) DQ_LB(16) DQ_LB(32) DQ_LB(64)
    default:
      set_error("linattn_bwd: m/z length " + std::to_string(n) + " is not built (powers of two up to 64)");
      return 2;
  }
#undef DQ_LB
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
  DQ_REQUIRE(a.part && a.part_floats >= (int64_t)LA_MAX_WAVES * 512 * C, "linattn_bwd: partial-sum scratch missing or too small");
  static_assert((int64_t)LA_MAX_WAVES * 512 >= 1024 * (int64_t)la_slot(1), "slot scratch: 1024 waves x 515*C floats must fit");
  static const bool long_all = [] { const char* e = std::getenv("DQ_LA_BWD_LONG"); return e && e[0] == '1'; }();
  if (n > 64 || (long_all && n >= 32 && C <= 8)) {
    // rows of 128 / 256 positions: the sweep kernel between two pointwise norm-backward launches
    BlockBwd b2;  // (1) post-norm backward: dyp = d loss / d ypre, d g_out, d b_out
    b2.u = a.ypre; b2.dy = a.dy; b2.du = a.dyp; b2.C = C; b2.rows = rows; b2.n = n; b2.rows_per_sample = rows;
    b2.g = a.f.g_out; b2.dg = a.dg_out; b2.dbias = a.db_out;
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
    return launch_block_bwd(b1, s);
  }
  LinAttnBwdK k;
  k.x = a.f.x; k.ypre = a.ypre; k.dy = a.dy; k.dx = a.dx; k.dxh = a.dxh; k.w_qkv = a.f.w_qkv; k.w_out = a.f.w_out;
  k.g_pre = a.f.g_pre; k.g_out = a.f.g_out; k.part = a.part; k.rows = rows; k.units_per_wave = 1;
  switch (C) {
    case 4: return linattn_bwd_n<4>(k, n, a, s);
    case 8: return linattn_bwd_n<8>(k, n, a, s);
    case 12: return linattn_bwd_n<12>(k, n, a, s);
    case 16: return linattn_bwd_n<16>(k, n, a, s);
    default: set_error("linattn_bwd: unsupported channel count " + std::to_string(C)); return 2;
  }
}

}  // namespace dq
