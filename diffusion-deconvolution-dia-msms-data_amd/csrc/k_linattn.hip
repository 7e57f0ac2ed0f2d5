// K5: Residual(PreNorm(LinearAttention)) over m/z, forward (and backward, below), 89 % of the network's FLOPs.
// Reference arithmetic: dquartic/model/unet1d.py:446-496 (LinearAttention), :143-176 (PreNorm), :64-79 (Residual),
// :113-140 (RMSNorm):
//   xh = rmsnorm(x)*g_pre ; q,k,v = Wqkv xh  (4 heads x 32) ; q = softmax_d(q) * 32^-0.5 ; k = softmax_n(k)
//   ctx[d][e] = sum_n k[d][n] v[e][n] ; out[e][n] = sum_d ctx[d][e] q[d][n] ; y = rmsnorm(Wo out + b)*g_out + x
//
// gfx950 design: ONE WAVE owns one m/z row (n >= 32) or 32/n rows (n < 32) and keeps the whole block in registers;
// all four contractions run on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32) with operand orientations
// chosen so that every accumulator is directly the next MFMA's operand (an accumulator X feeds, register by
// register, a product that sums over X's ROW index) and both softmaxes are reductions over a lane's own
// registers plus one swap with lane^32:
//   kT[n][d] = mfma(A = xh, B = Wk)   -> column d on the lane, positions n in the registers  (softmax over n in-lane)
//   vT[n][e] = mfma(A = xh, B = Wv)
//   ctx[d][e] = sum_r mfma(A = kT.r, B = vT.r)
//   q[d][n]  = mfma(A = Wq, B = xh)   -> position on the lane, d in the registers             (softmax over d in-lane)
//   out[e][n] = sum_r mfma(A = ctx.r, B = q.r)
// to_out (K = 128, M = C <= 16) is done on the VALU from out's registers with Wo pre-permuted in LDS; no other LDS,
// no barriers after the weight staging, no HBM traffic besides x in / y out (8*C*n bytes per row).
// Checked lane-for-lane on the CPU by oracle/wave_emu.py.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"

namespace dq {


__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ constexpr int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ float swap32(float v) { return __shfl_xor(v, 32, 64); }

template <int C, int N>
__global__ void __launch_bounds__(256) k_linattn_fwd(LinAttn a) {
  constexpr int NB = N >= 32 ? N / 32 : 1;    // 32-position blocks per row
  constexpr int RW = N >= 32 ? 1 : 32 / N;    // rows per wave
  constexpr int NJ = C <= 8 ? 4 : 8;          // x registers per lane (channel = rowmap(j, half))
  constexpr int SEG = N >= 32 ? 16 : (N >= 8 ? N / 2 : N);  // registers of one row inside a lane
  constexpr bool PARTNER = N >= 8;            // does lane^32 hold positions of the same row?
  static_assert(NB <= 2, "rows longer than 64 take the two-pass path (not built here)");

  __shared__ float wo_lds[4 * 2 * C * 16];  // [head][half][c][r] = Wo[c][head*32 + rowmap(r, half)]
  __shared__ float tiles[(N > 1 && N < 32) ? 4 : 1][(N > 1 && N < 32) ? 32 * 33 : 1];  // wave-private transpose tiles (short rows only)
  float* tile = tiles[(N > 1 && N < 32) ? (threadIdx.x >> 6) : 0];
  for (int i = threadIdx.x; i < 4 * 2 * C * 16; i += blockDim.x) {
    const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = i / (32 * C);
    wo_lds[i] = a.w_out[c * 128 + hd * 32 + rowmap(r, hh)];
  }
  // MFMA weight operands, laid out [q|k|v][head][j][half][col] so that a wave reads 2 x 32 consecutive floats; channels beyond C
  // are zero.  The q and k rows carry log2(e): both softmaxes then use exp2 (v_exp_f32) directly, which changes nothing
  // mathematically (softmax(x) = 2^(x*log2e - max) / sum).
  constexpr int WQ = 3 * 4 * NJ * 2 * 32;
  __shared__ float wqkv_lds[N > 1 ? WQ : 1];
  if (N > 1) {
    for (int i = threadIdx.x; i < WQ; i += blockDim.x) {
      const int cc = i & 31, hh = (i >> 5) & 1, j = (i >> 6) % NJ, hd = (i / (64 * NJ)) & 3, m = i / (256 * NJ);
      const int c = rowmap(j, hh);
      const float w = c < C ? a.w_qkv[(m * 128 + hd * 32 + cc) * C + c] : 0.f;
      wqkv_lds[i] = m < 2 ? w * 1.4426950408889634f : w;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int unit = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int row0 = unit * RW;
  if (row0 >= a.rows) return;
  const int rl = N >= 32 ? 0 : col / N;
  const int row = row0 + rl;
  const bool row_ok = row < a.rows;
  const float sqC = sqrtf((float)C);
  const float scale = 0.17677669529663687f;  // 32^-0.5

  // ---- load x, pre-norm
  float X[NB][NJ], Xh[NB][NJ];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int pos = N >= 32 ? blk * 32 + col : col % N;
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = rowmap(j, half);
      X[blk][j] = (row_ok && c < C) ? a.x[((int64_t)row * C + c) * N + pos] : 0.f;
      ssq = fmaf(X[blk][j], X[blk][j], ssq);
    }
    ssq += swap32(ssq);
    const float inv = sqC / fmaxf(sqrtf(ssq), RMS_EPS);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = rowmap(j, half);
      Xh[blk][j] = X[blk][j] * inv * (c < C ? a.g_pre[c] : 0.f);
    }
  }

  float ypart[NB][C];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int c = 0; c < C; ++c) ypart[blk][c] = 0.f;

#pragma unroll 1
  for (int hd = 0; hd < 4; ++hd) {
    // weight operands of this head: lane (col, half) supplies W[o_base + col][rowmap(j, half)]
    float wq[NJ], wk[NJ], wv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (N > 1) {
        wq[j] = wqkv_lds[(((0 * 4 + hd) * NJ + j) * 2 + half) * 32 + col];
        wk[j] = wqkv_lds[(((1 * 4 + hd) * NJ + j) * 2 + half) * 32 + col];
        wv[j] = wqkv_lds[(((2 * 4 + hd) * NJ + j) * 2 + half) * 32 + col];
      } else {
        const int c = rowmap(j, half);
        wq[j] = wk[j] = 0.f;
        wv[j] = c < C ? a.w_qkv[(256 + hd * 32 + col) * C + c] : 0.f;
      }
    }

    f32x16 out[NB];
    if (N == 1) {
      // softmax over a single position is 1 and the q softmax sums to 1: out = 32^-0.5 * v
      f32x16 v = {0};
#pragma unroll
      for (int j = 0; j < NJ; ++j) v = mfma32(wv[j], Xh[0][j], v);  // v[e][row]
      out[0] = v * scale;
    } else {
      // ---------------- K^T and V^T, softmax over the positions of each row
      f32x16 kT[NB], vT[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        f32x16 ak = {0}, av = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          ak = mfma32(Xh[blk][j], wk[j], ak);
          av = mfma32(Xh[blk][j], wv[j], av);
        }
        kT[blk] = ak;
        vT[blk] = av;
      }
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += SEG) {
        float m = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) m = fmaxf(m, kT[blk][r]);
        if (PARTNER) m = fmaxf(m, swap32(m));
        float ssum = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) {
            const float e = __builtin_amdgcn_exp2f(kT[blk][r] - m);  // k rows are pre-scaled by log2(e)
            kT[blk][r] = e;
            ssum += e;
          }
        if (PARTNER) ssum += swap32(ssum);
        const float rs = 1.0f / ssum;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
          for (int r = s0; r < s0 + SEG; ++r) kT[blk][r] *= rs;
      }

      // ---------------- per block: q, softmax over d ; per row: ctx, out
      f32x16 ctx_row = {0};  // n >= 32: the wave's single row
      if (N >= 32) {
#pragma unroll
        for (int b2 = 0; b2 < NB; ++b2)
#pragma unroll
          for (int r = 0; r < 16; ++r) ctx_row = mfma32(kT[b2][r], vT[b2][r], ctx_row);
      }
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        f32x16 q = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) q = mfma32(wq[j], Xh[blk][j], q);
        float m = q[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
        m = fmaxf(m, swap32(m));
        float ssum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          q[r] = __builtin_amdgcn_exp2f(q[r] - m);  // q rows are pre-scaled by log2(e)
          ssum += q[r];
        }
        ssum += swap32(ssum);
        const float qs = scale / ssum;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] *= qs;

        f32x16 o = {0};
        if (N >= 32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o = mfma32(ctx_row[r], q[r], o);
        } else {
          // short rows (32/N rows share this block): "quadratic" form, no per-row loop and no wasted MFMAs:
          //   S^T[n'][n] = sum_d K[d][n'] q[d][n], kept only for pairs of the same m/z row ; out[e][n] = sum_n' v[e][n'] S^T[n'][n]
          const f32x16 Kd = transpose_tile(kT[0], tile, col, half);  // rows d, col n'
          f32x16 st = {0};
          st = xty(Kd, q, st);
          st = mask_same_row<N>(st, col, half);
          o = xty(vT[0], st, o);
        }
        out[blk] = o;
      }
    }

    // ---------------- to_out on the VALU: lane holds out[e = rowmap(r, half)][position]
    const float* wl = wo_lds + (hd * 2 + half) * C * 16;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float w16[16];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 t = *reinterpret_cast<const float4*>(wl + c * 16 + r4 * 4);
        w16[r4 * 4 + 0] = t.x; w16[r4 * 4 + 1] = t.y; w16[r4 * 4 + 2] = t.z; w16[r4 * 4 + 3] = t.w;
      }
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) ypart[blk][c] = fmaf(w16[r], out[blk][r], ypart[blk][c]);
    }
  }

  // ---- bias, post-norm, residual, store
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int pos = N >= 32 ? blk * 32 + col : col % N;
    float yv[C];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      yv[c] = ypart[blk][c] + swap32(ypart[blk][c]) + a.b_out[c];
      ssq = fmaf(yv[c], yv[c], ssq);
    }
    const float inv = sqC / fmaxf(sqrtf(ssq), RMS_EPS);
    if (a.ypre) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = rowmap(j, 0), c1 = c0 + 4;
        float lo = 0.f, hi = 0.f;
        if (c0 < C) lo = yv[c0 < C ? c0 : 0];
        if (c1 < C) hi = yv[c1 < C ? c1 : 0];
        const int c = c0 + 4 * half;
        if (row_ok && c < C) a.ypre[((int64_t)row * C + c) * N + pos] = half ? hi : lo;
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c0 = rowmap(j, 0), c1 = c0 + 4;  // this lane's channel is c0 + 4*half
      float lo = 0.f, hi = 0.f;
      if (c0 < C) lo = yv[c0 < C ? c0 : 0] * a.g_out[c0 < C ? c0 : 0];
      if (c1 < C) hi = yv[c1 < C ? c1 : 0] * a.g_out[c1 < C ? c1 : 0];
      const int c = c0 + 4 * half;
      if (row_ok && c < C) a.y[((int64_t)row * C + c) * N + pos] = fmaf(half ? hi : lo, inv, X[blk][j]);
    }
  }
}

template <int C>
static int linattn_fwd_n(const LinAttn& a, hipStream_t s) {
#define DQ_LA(NN)                                                                      \
  case NN: {                                                                           \
    constexpr int RW = NN >= 32 ? 1 : 32 / NN;                                         \
    const int units = cdiv(a.rows, RW);                                                \
    hipLaunchKernelGGL((k_linattn_fwd<C, NN>), dim3(cdiv(units, 4)), dim3(256), 0, s, a); \
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

int launch_linattn_fwd(const LinAttn& a, hipStream_t s) {
  DQ_REQUIRE(a.x && a.y && a.w_qkv && a.w_out && a.b_out && a.g_pre && a.g_out, "linattn_fwd: missing operand");
  if (a.rows == 0) return 0;
  if (a.n > 64) return launch_linattn_fwd_long(a, s);
  switch (a.C) {
    case 4: return linattn_fwd_n<4>(a, s);
    case 8: return linattn_fwd_n<8>(a, s);
    case 12: return linattn_fwd_n<12>(a, s);
    case 16: return linattn_fwd_n<16>(a, s);
    default: set_error("linattn_fwd: unsupported channel count " + std::to_string(a.C)); return 2;
  }
}

}  // namespace dq
