// K5 for m/z rows longer than 64 positions (n = 128, 256: the 256 x 2000 configuration, BASELINE configs[4]).  Same
// re-associated algorithm as k_linattn.hip / k_la_bwd.hip (per head M[d][c] = sum_n K[d][n] xh[c][n], P[c][n] = sum_d M[d][c]
// Q[d][n], ypre = sum_h W2_h P_h + b), but a row no longer fits in registers: the row is swept in 32-position blocks and x is
// re-read per sweep (L2 hits).  The k-softmax over the row is ONLINE: M is only 32 x C per head, so rescaling the running
// accumulator by exp(m_old - m_new) costs C multiplies per block -- one sweep instead of a max pass plus a sum pass.
//   forward : sweep 1 builds M of all four heads (state per head: running max, running sum, C accumulators);
//             sweep 2 finishes each block completely (q, P, W2, bias, post-norm, residual, store).
//   backward: head loop outermost (gradient accumulators stay in registers across a wave's rows); per (head, row):
//             sweep 1 = M ; sweep 2 = q side (dP, dW2, dQ, dq_raw, dWq, Wq^T path of dXh, dM accumulated) ;
//             sweep 3 = k side (dK^T = xh^T dM, softmax backward with sum_n dK K = sum_c dM M in-lane, dWk, Wk^T and K^T paths
//             of dXh).  dXh is accumulated in global memory per block (head 0's first touch initialises).
// The post-norm / pre-norm backward and the residual stay in k_block_bwd for these rows (launch_linattn_bwd).
// Row length: a compile-time N (128 / 256: the BASELINE configurations) or N = 0 = "read it from the launch" -- any length >= 1,
// e.g. the 40000 -> 625 positions of the reference's shipped configuration (dquartic_train_config.json:35, unet1d.py:1027) or the
// 5 .. 320 of an odd test shape.  With a run-time length the last 32-block may be ragged: its missing positions load as zeros
// (so xh, dYpre and with them every gradient contribution vanish there), their k-softmax logits are -inf, and nothing is stored.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"

namespace dq {

namespace {

typedef float lf32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_4x4x1_16b_f32: block = lane >> 2; register i of lane (blk, j) += A_blk[i] * B_blk[j]
__device__ __forceinline__ lf32x4 lmfma4(float a, float b, lf32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lfence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
constexpr float L_LOG2E = 1.4426950408889634f;
constexpr float L_SCALE = 0.17677669529663687f;  // 32^-0.5

template <int C, int NJ>
__device__ __forceinline__ void load_block(const float* __restrict__ src, int64_t row, int N, int pos, int half, float* out) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = la_chan(C, j, half);
    out[j] = (c < C && pos < N) ? src[(row * C + c) * N + pos] : 0.f;  // (pos >= N: the ragged tail of a run-time row length)
  }
}

// k-softmax logits (rows n = rmap(r, half) of the block starting at p0, col d): positions beyond the row are -inf
__device__ __forceinline__ f32x16 mask_tail(f32x16 kT, int p0, int N, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (p0 + rmap(r, half) >= N) kT[r] = -INFINITY;
  return kT;
}

template <int C, int NJ>
__device__ __forceinline__ void prenorm(const float* x, const float* gpre, float* xh) {
  float ssq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) ssq = fmaf(x[j], x[j], ssq);
  ssq += swap_half(ssq);
  const float inv = rms_inv(ssq, sqrtf((float)C));
#pragma unroll
  for (int j = 0; j < NJ; ++j) xh[j] = x[j] * inv * gpre[j];
}

// stage this block's xh (and optionally a second per-position channel vector) as [c][32]
template <int C, int NJ>
__device__ __forceinline__ void stage_cn(float* dst, const float* v, int col, int half) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = la_chan(C, j, half);
    if (c < C) dst[c * 32 + col] = v[j];
  }
}

template <int NJ>
__device__ __forceinline__ f32x16 proj_a(const float* xh, const float* w) {  // A = xh, B = w  -> (rows n, col o)
  f32x16 t = {0};
#pragma unroll
  for (int j = 0; j < NJ; ++j) t = mfma_f32(xh[j], w[j], t);
  return t;
}
template <int NJ>
__device__ __forceinline__ f32x16 proj_b(const float* w, const float* xh) {  // A = w, B = xh  -> (rows o, col n)
  f32x16 t = {0};
#pragma unroll
  for (int j = 0; j < NJ; ++j) t = mfma_f32(w[j], xh[j], t);
  return t;
}

// sum_r mfma4(A = src[(4*g + (lane&3)) * 32 + rmap(r, half)], B = t[r]) on two interleaved chains
__device__ __forceinline__ lf32x4 chain32(const float* src, int g, const f32x16& t, int lane, int half) {
  const float* ar = src + (g * 4 + (lane & 3)) * 32 + 4 * half;
  lf32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 a4 = *reinterpret_cast<const float4*>(ar + 8 * q4);
    t0 = lmfma4(a4.x, t[q4 * 4 + 0], t0); t1 = lmfma4(a4.y, t[q4 * 4 + 1], t1);
    t0 = lmfma4(a4.z, t[q4 * 4 + 2], t0); t1 = lmfma4(a4.w, t[q4 * 4 + 3], t1);
  }
  return t0 + t1;
}

// online softmax-weighted accumulation of one block into (m, ssum, mt): kT holds log2-domain logits (rows n, col d)
template <int CG>
__device__ __forceinline__ void online_m(f32x16 kT, const float* xs, float& m, float& ssum, lf32x4 (&mt)[CG], int lane, int half) {
  float bm = kT[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) bm = fmaxf(bm, kT[r]);
  bm = fmaxf(bm, swap_half(bm));
  const float mn = fmaxf(m, bm);
  const float al = __builtin_amdgcn_exp2f(m - mn);  // first block: exp2(-inf) = 0
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    kT[r] = __builtin_amdgcn_exp2f(kT[r] - mn);
    s += kT[r];
  }
  ssum = fmaf(ssum, al, s);
  m = mn;
#pragma unroll
  for (int g = 0; g < CG; ++g) mt[g] = mt[g] * al + chain32(xs, g, kT, lane, half);
}

__device__ __forceinline__ f32x16 q_exp(f32x16 q, float& qs) {  // un-normalised exps; qs = 32^-0.5 / sum
  float m = q[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
  m = fmaxf(m, swap_half(m));
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    q[r] = __builtin_amdgcn_exp2f(q[r] - m);
    s += q[r];
  }
  s += swap_half(s);
  qs = L_SCALE * fast_rcp(s);
  return q;
}

template <int C>
__device__ __forceinline__ float own_of(const float (&v)[C], int j, int half) {
  const int c0 = la_chan(C, j, 0), c1 = la_chan(C, j, 1);
  const float lo = c0 < C ? v[c0 < C ? c0 : 0] : 0.f, hi = c1 < C ? v[c1 < C ? c1 : 0] : 0.f;
  return half ? hi : lo;
}

template <int C>
__device__ __forceinline__ void stage_w2(float* w2_lds, const float* __restrict__ w_out, const float* __restrict__ w_qkv) {
  for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) {
    const int c = i % C, cp = (i / C) % C, hd = i / (C * C);
    float s = 0.f;
#pragma unroll 8
    for (int e = 0; e < 32; ++e) s = fmaf(w_out[cp * 128 + hd * 32 + e], w_qkv[(256 + hd * 32 + e) * C + c], s);
    w2_lds[i] = s;
  }
}

}  // namespace

struct LinAttnBwdLongK {
  const float* x; const float* dyp; float* dxh;
  const float* w_qkv; const float* w_out; const float* g_pre;
  float* part;
  int rows; int units_per_wave;
  int n;  // row length when the kernel is instantiated with N = 0
};

// ---------------------------------------------------------------------------------------------------------------
template <int C, int NT>
__global__ void __launch_bounds__(256) k_linattn_fwd_long(LinAttn a) {
  const int N = NT ? NT : a.n;
  const int NB = (N + 31) / 32;
  constexpr bool RAGGED = NT == 0;
  constexpr int NJ = la_nj(C);
  constexpr int CG = C / 4;
  __shared__ __attribute__((aligned(16))) float w2_lds[4 * C * C];
  __shared__ __attribute__((aligned(16))) float xs_lds[4][C * 32];
  __shared__ __attribute__((aligned(16))) float ms_lds[4][4 * C * 32];  // per wave: M of the four heads as [head][c][d]
  stage_w2<C>(w2_lds, a.w_out, a.w_qkv);
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* xs = xs_lds[wv];
  float* ms = ms_lds[wv];
  const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + wv;
  if (row >= a.rows) return;
  float gpre[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) gpre[j] = la_chan(C, j, half) < C ? a.g_pre[la_chan(C, j, half)] : 0.f;

  // ---- sweep 1: M of the four heads, online softmax over the blocks
  {
    float wk[4][NJ], m[4], ssum[4];
    lf32x4 mt[4][CG];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      m[hd] = -INFINITY; ssum[hd] = 0.f;
#pragma unroll
      for (int g = 0; g < CG; ++g) mt[hd][g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = la_chan(C, j, half);
        wk[hd][j] = c < C ? a.w_qkv[(128 + hd * 32 + col) * C + c] * L_LOG2E : 0.f;
      }
    }
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      float x[NJ], xh[NJ];
      load_block<C, NJ>(a.x, row, N, b * 32 + col, half, x);
      prenorm<C, NJ>(x, gpre, xh);
      lfence();
      stage_cn<C, NJ>(xs, xh, col, half);
      lfence();
#pragma unroll
      for (int hd = 0; hd < 4; ++hd) {
        f32x16 kT = proj_a<NJ>(xh, wk[hd]);
        if (RAGGED) kT = mask_tail(kT, b * 32, N, half);
        online_m<CG>(kT, xs, m[hd], ssum[hd], mt[hd], lane, half);
      }
    }
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      const float rs = fast_rcp(ssum[hd] + swap_half(ssum[hd]));
#pragma unroll
      for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = (mt[hd][g][i] + swap_half(mt[hd][g][i])) * rs;
          if (half == 0) ms[(hd * C + g * 4 + i) * 32 + col] = v;
        }
    }
    lfence();
  }

  // ---- sweep 2: finish every block
  float wq[4][NJ];
#pragma unroll
  for (int hd = 0; hd < 4; ++hd)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      wq[hd][j] = c < C ? a.w_qkv[(hd * 32 + col) * C + c] * L_LOG2E : 0.f;
    }
#pragma unroll 1
  for (int b = 0; b < NB; ++b) {
    const int pos = b * 32 + col;
    float x[NJ], xh[NJ], yown[NJ];
    load_block<C, NJ>(a.x, row, N, pos, half, x);
    prenorm<C, NJ>(x, gpre, xh);
#pragma unroll
    for (int j = 0; j < NJ; ++j) yown[j] = 0.f;
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      float qs;
      const f32x16 q = q_exp(proj_b<NJ>(wq[hd], xh), qs);
      float P[C];
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const lf32x4 pp = chain32(ms + hd * C * 32, g, q, lane, half);
#pragma unroll
        for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swap_half(pp[i]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cp = la_chan(C, j, half);
        if (cp < C) {
          const float* w = w2_lds + (hd * C + cp) * C;
          float s = 0.f;
#pragma unroll
          for (int c4 = 0; c4 < CG; ++c4) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * c4);
            s = fmaf(w4.x, P[4 * c4 + 0], fmaf(w4.y, P[4 * c4 + 1], fmaf(w4.z, P[4 * c4 + 2], fmaf(w4.w, P[4 * c4 + 3], s))));
          }
          yown[j] = fmaf(qs, s, yown[j]);
        }
      }
    }
    float yv[NJ];
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      yv[j] = c < C ? yown[j] + a.b_out[c] : 0.f;
      ssq = fmaf(yv[j], yv[j], ssq);
    }
    ssq += swap_half(ssq);
    const float inv = rms_inv(ssq, sqrtf((float)C));
    float go[NJ];  // gains read before the first store of this position (a load behind a store waits for it)
#pragma unroll
    for (int j = 0; j < NJ; ++j) go[j] = la_chan(C, j, half) < C ? a.g_out[la_chan(C, j, half)] : 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      if (c < C && pos < N) {
        const int64_t off = (row * C + c) * N + pos;
        if (a.ypre) a.ypre[off] = yv[j];
        a.y[off] = fmaf(yv[j] * go[j], inv, x[j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int C, int NT>
__global__ void __launch_bounds__(256, 2) k_linattn_bwd_long(LinAttnBwdLongK a) {  // <= 256 registers: two waves per SIMD
  const int N = NT ? NT : a.n;
  const int NB = (N + 31) / 32;
  constexpr bool RAGGED = NT == 0;
  constexpr int NJ = la_nj(C);
  constexpr int CG = C / 4;
  __shared__ __attribute__((aligned(16))) float wp_lds[2 * 4 * 2 * C * 16];  // [q|k][head][half][c][r]
  __shared__ __attribute__((aligned(16))) float w2_lds[4 * C * C];
  __shared__ float tiles[4][32 * 33];
  // per wave: xh | dYpre | P | dP as [c][32] ; M | dM as [c][d] ; dW2 of the head being flushed
  __shared__ __attribute__((aligned(16))) float stage[4][6 * C * 32 + C * C];
  for (int i = threadIdx.x; i < 2 * 4 * 2 * C * 16; i += blockDim.x) {
    const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = (i / (32 * C)) & 3, m = i / (128 * C);
    wp_lds[i] = a.w_qkv[(m * 128 + hd * 32 + rmap(r, hh)) * C + c];
  }
  stage_w2<C>(w2_lds, a.w_out, a.w_qkv);
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  float* xs = stage[wv];
  float* dys = xs + C * 32;
  float* ps = dys + C * 32;
  float* dps = ps + C * 32;
  float* ms = dps + C * 32;
  float* dms = ms + C * 32;
  float* w2g = dms + C * 32;
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + wv;
  const int u0 = wave_id * a.units_per_wave;
  if (u0 >= a.rows) return;
  const int u1 = min(a.rows, u0 + a.units_per_wave);
  float gpre[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) gpre[j] = la_chan(C, j, half) < C ? a.g_pre[la_chan(C, j, half)] : 0.f;

  auto chainw = [&](int m, int hd, int g, const f32x16& t) {  // A = wp_lds[m][hd][half][c = 4*g + (lane&3)][r]
    const float* wr = wp_lds + (((m * 4 + hd) * 2 + half) * C + g * 4 + (lane & 3)) * 16;
    lf32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wr + r4 * 4);
      t0 = lmfma4(w4.x, t[r4 * 4 + 0], t0); t1 = lmfma4(w4.y, t[r4 * 4 + 1], t1);
      t0 = lmfma4(w4.z, t[r4 * 4 + 2], t0); t1 = lmfma4(w4.w, t[r4 * 4 + 3], t1);
    }
    return t0 + t1;
  };
  // dxh[c][pos] (=|+=) both halves' sums of part; each lane stores its own channels
  auto dxh_store = [&](int64_t row, int pos, const lf32x4 (&part)[CG], bool first) {
    float full[C];
#pragma unroll
    for (int c = 0; c < C; ++c) full[c] = part[c >> 2][c & 3] + swap_half(part[c >> 2][c & 3]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      if (c < C && pos < N) {
        float* dst = a.dxh + (row * C + c) * N + pos;
        const float val = own_of<C>(full, j, half);
        *dst = first ? val : *dst + val;
      }
    }
  };

#pragma unroll 1
  for (int hd = 0; hd < 4; ++hd) {
    float wq[NJ], wk[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = la_chan(C, j, half);
      wq[j] = c < C ? a.w_qkv[(hd * 32 + col) * C + c] * L_LOG2E : 0.f;
      wk[j] = c < C ? a.w_qkv[(128 + hd * 32 + col) * C + c] * L_LOG2E : 0.f;
    }
    lf32x4 gq[CG], gk[CG], gw2[CG][CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      gq[g] = gk[g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g2 = 0; g2 < CG; ++g2) gw2[g][g2] = lf32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll 1
    for (int u = u0; u < u1; ++u) {
      const int64_t row = u;
      // ---- sweep 1: M (online softmax over the blocks)
      float m = -INFINITY, ssum = 0.f;
      float Mr[C];
      {
        lf32x4 mt[CG];
#pragma unroll
        for (int g = 0; g < CG; ++g) mt[g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int b = 0; b < NB; ++b) {
          float x[NJ], xh[NJ];
          load_block<C, NJ>(a.x, row, N, b * 32 + col, half, x);
          prenorm<C, NJ>(x, gpre, xh);
          lfence();
          stage_cn<C, NJ>(xs, xh, col, half);
          lfence();
          f32x16 kT = proj_a<NJ>(xh, wk);
          if (RAGGED) kT = mask_tail(kT, b * 32, N, half);
          online_m<CG>(kT, xs, m, ssum, mt, lane, half);
        }
        ssum += swap_half(ssum);
        const float rs = fast_rcp(ssum);
        lfence();
#pragma unroll
        for (int g = 0; g < CG; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            Mr[g * 4 + i] = (mt[g][i] + swap_half(mt[g][i])) * rs;
            if (half == 0) ms[(g * 4 + i) * 32 + col] = Mr[g * 4 + i];
          }
        lfence();
      }
      // ---- sweep 2: q side
      lf32x4 dmt[CG];
#pragma unroll
      for (int g = 0; g < CG; ++g) dmt[g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        const int pos = b * 32 + col;
        float x[NJ], xh[NJ], dy[NJ];
        load_block<C, NJ>(a.x, row, N, pos, half, x);
        load_block<C, NJ>(a.dyp, row, N, pos, half, dy);
        prenorm<C, NJ>(x, gpre, xh);
        lfence();
        stage_cn<C, NJ>(xs, xh, col, half);
        stage_cn<C, NJ>(dys, dy, col, half);
        lfence();
        float qs;
        f32x16 q = q_exp(proj_b<NJ>(wq, xh), qs);
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] *= qs;  // normalised Q (incl. 32^-0.5)
        // dP[c] = sum_c' W2[c'][c] dYpre[c'][n] ; P[c] = sum_d M[d][c] Q[d][n]
        float dP[C], P[C];
        {
          float dya[C];
#pragma unroll
          for (int c = 0; c < C; ++c) { dya[c] = dys[c * 32 + col]; dP[c] = 0.f; }
#pragma unroll
          for (int cp = 0; cp < C; ++cp) {
            const float* w = w2_lds + (hd * C + cp) * C;
#pragma unroll
            for (int c4 = 0; c4 < CG; ++c4) {
              const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * c4);
              dP[4 * c4 + 0] = fmaf(w4.x, dya[cp], dP[4 * c4 + 0]); dP[4 * c4 + 1] = fmaf(w4.y, dya[cp], dP[4 * c4 + 1]);
              dP[4 * c4 + 2] = fmaf(w4.z, dya[cp], dP[4 * c4 + 2]); dP[4 * c4 + 3] = fmaf(w4.w, dya[cp], dP[4 * c4 + 3]);
            }
          }
        }
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const lf32x4 pp = chain32(ms, g, q, lane, half);
#pragma unroll
          for (int i = 0; i < 4; ++i) P[g * 4 + i] = pp[i] + swap_half(pp[i]);
        }
        if (half == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) { ps[c * 32 + col] = P[c]; dps[c * 32 + col] = dP[c]; }
        }
        // dQ = M dP (K = C), softmax backward, Wq paths
        f32x16 dq = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) dq = mfma_f32(own_of<C>(Mr, j, half), own_of<C>(dP, j, half), dq);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t = fmaf(q[r], dq[r], t);
        t = (t + swap_half(t)) * (1.0f / L_SCALE);
        f32x16 dq_raw;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_raw[r] = q[r] * (dq[r] - t);
        lf32x4 part[CG];
#pragma unroll
        for (int g = 0; g < CG; ++g) part[g] = chainw(0, hd, g, dq_raw);
        dxh_store(row, pos, part, hd == 0);
        const f32x16 dq_rawT = transpose_tile(dq_raw, tile, col, half);
        const f32x16 qT = transpose_tile(q, tile, col, half);
        lfence();  // ps / dps of this block complete
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          gq[g] += chain32(xs, g, dq_rawT, lane, half);
          dmt[g] += chain32(dps, g, qT, lane, half);
        }
        // dW2[c'][c] += sum_n dYpre[c'][n] P[c][n] over this block: position n = 16 * s + (lane >> 2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int n = 16 * s + (lane >> 2);
#pragma unroll
          for (int g1 = 0; g1 < CG; ++g1) {
            const float av = dys[(4 * g1 + (lane & 3)) * 32 + n];
#pragma unroll
            for (int g2 = 0; g2 < CG; ++g2) gw2[g1][g2] = lmfma4(av, ps[(4 * g2 + (lane & 3)) * 32 + n], gw2[g1][g2]);
          }
        }
      }
      // dM (both halves), its [c][d] image, and sum_n dK K = sum_c dM M
      float dMr[C];
      float dl = 0.f;
      lfence();
#pragma unroll
      for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dMr[g * 4 + i] = dmt[g][i] + swap_half(dmt[g][i]);
          dl = fmaf(dMr[g * 4 + i], Mr[g * 4 + i], dl);
          if (half == 0) dms[(g * 4 + i) * 32 + col] = dMr[g * 4 + i];
        }
      lfence();
      const float rs = fast_rcp(ssum);
      // ---- sweep 3: k side
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        const int pos = b * 32 + col;
        float x[NJ], xh[NJ];
        load_block<C, NJ>(a.x, row, N, pos, half, x);
        prenorm<C, NJ>(x, gpre, xh);
        lfence();
        stage_cn<C, NJ>(xs, xh, col, half);
        lfence();
        f32x16 kT = proj_a<NJ>(xh, wk);
        if (RAGGED) kT = mask_tail(kT, b * 32, N, half);  // exp2(-inf) = 0: no key beyond the row
#pragma unroll
        for (int r = 0; r < 16; ++r) kT[r] = __builtin_amdgcn_exp2f(kT[r] - m) * rs;  // normalised K^T
        f32x16 dkT = {0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) dkT = mfma_f32(xh[j], own_of<C>(dMr, j, half), dkT);
        f32x16 dk_rawT;
#pragma unroll
        for (int r = 0; r < 16; ++r) dk_rawT[r] = kT[r] * (dkT[r] - dl);
#pragma unroll
        for (int g = 0; g < CG; ++g) gk[g] += chain32(xs, g, dk_rawT, lane, half);
        const f32x16 dk_raw = transpose_tile(dk_rawT, tile, col, half);
        const f32x16 Kd = transpose_tile(kT, tile, col, half);
        lf32x4 part[CG];
#pragma unroll
        for (int g = 0; g < CG; ++g) part[g] = chainw(1, hd, g, dk_raw) + chain32(dms, g, Kd, lane, half);
        dxh_store(row, pos, part, false);
      }
    }

    // ---- flush: dWq, dWk; dW2 -> dWv, dWo (slot layout of the 512*C form: dWqkv | dWo)
    float* slot = a.part + (int64_t)wave_id * (512 * C);
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * g + i;
        const float vq = gq[g][i] + swap_half(gq[g][i]), vk = gk[g][i] + swap_half(gk[g][i]);
        if (half == 0) {
          slot[(hd * 32 + col) * C + c] = vq;
          slot[(128 + hd * 32 + col) * C + c] = vk;
        }
      }
    lfence();
#pragma unroll
    for (int g1 = 0; g1 < CG; ++g1)
#pragma unroll
      for (int g2 = 0; g2 < CG; ++g2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = gw2[g1][g2][i];
          v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          if (lane < 4) w2g[(4 * g1 + i) * C + 4 * g2 + lane] = v;
        }
    lfence();
    {
      float wvr[C], wor[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        wvr[c] = a.w_qkv[(256 + hd * 32 + col) * C + c];
        wor[c] = a.w_out[c * 128 + hd * 32 + col];
      }
      if (half == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float sv = 0.f, so = 0.f;
#pragma unroll
          for (int k = 0; k < C; ++k) {
            sv = fmaf(wor[k], w2g[k * C + c], sv);
            so = fmaf(w2g[c * C + k], wvr[k], so);
          }
          slot[(256 + hd * 32 + col) * C + c] = sv;
          slot[384 * C + c * 128 + hd * 32 + col] = so;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
int launch_linattn_fwd_long(const LinAttn& a, hipStream_t s) {
  dim3 grid(cdiv(a.rows, 4)), block(256);
#define DQ_LF(CC, NN)                                                              \
  if (a.C == CC && a.n == NN) {                                                    \
    hipLaunchKernelGGL((k_linattn_fwd_long<CC, NN>), grid, block, 0, s, a);        \
    DQ_LAUNCH_CHECK();                                                             \
    return 0;                                                                      \
  }
  DQ_LF(4, 128) DQ_LF(4, 256) DQ_LF(8, 128) DQ_LF(8, 256) DQ_LF(12, 128) DQ_LF(16, 128)
#undef DQ_LF
  // any other row length: the same kernel with the length read from the launch (ragged last block masked)
  switch (a.C) {
    case 4: hipLaunchKernelGGL((k_linattn_fwd_long<4, 0>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((k_linattn_fwd_long<8, 0>), grid, block, 0, s, a); break;
    case 12: hipLaunchKernelGGL((k_linattn_fwd_long<12, 0>), grid, block, 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_linattn_fwd_long<16, 0>), grid, block, 0, s, a); break;
    default: set_error("linattn_fwd: channel count " + std::to_string(a.C) + " is not built (4, 8, 12, 16)"); return 2;
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_linattn_bwd_long(const float* x, const float* dyp, float* dxh, const float* w_qkv, const float* w_out, const float* g_pre,
                            float* part, int C, int rows, int n, int* waves_out, hipStream_t s) {
  // two waves per SIMD are resident (<= 256 registers): 2048 waves = one resident round, and <= LA_MAX_WAVES partial slots
  LinAttnBwdLongK k{x, dyp, dxh, w_qkv, w_out, g_pre, part, rows, std::max(1, cdiv(rows, 2048)), n};
  const int waves = cdiv(rows, k.units_per_wave);
  *waves_out = waves;
  dim3 grid(cdiv(waves, 4)), block(256);
#define DQ_LBL(CC, NN)                                                             \
  if (C == CC && n == NN) {                                                        \
    hipLaunchKernelGGL((k_linattn_bwd_long<CC, NN>), grid, block, 0, s, k);        \
    DQ_LAUNCH_CHECK();                                                             \
    return 0;                                                                      \
  }
  DQ_LBL(4, 128) DQ_LBL(4, 256) DQ_LBL(8, 128) DQ_LBL(8, 256) DQ_LBL(12, 128) DQ_LBL(16, 128)
  DQ_LBL(4, 64) DQ_LBL(4, 32) DQ_LBL(8, 32) DQ_LBL(8, 64)
#undef DQ_LBL
  switch (C) {  // any other row length (see launch_linattn_fwd_long)
    case 4: hipLaunchKernelGGL((k_linattn_bwd_long<4, 0>), grid, block, 0, s, k); break;
    case 8: hipLaunchKernelGGL((k_linattn_bwd_long<8, 0>), grid, block, 0, s, k); break;
    case 12: hipLaunchKernelGGL((k_linattn_bwd_long<12, 0>), grid, block, 0, s, k); break;
    case 16: hipLaunchKernelGGL((k_linattn_bwd_long<16, 0>), grid, block, 0, s, k); break;
    default: set_error("linattn_bwd: channel count " + std::to_string(C) + " is not built (4, 8, 12, 16)"); return 2;
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
