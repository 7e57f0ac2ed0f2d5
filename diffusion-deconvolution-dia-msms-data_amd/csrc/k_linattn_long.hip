// K5 for m/z rows longer than 64 positions (n = 128, 256: the 256 x 2000 configuration): same algorithm and MFMA
// orientations as k_linattn.hip / k_linattn_bwd.hip, but a row no longer fits in registers, so the k-softmax over the row
// is done in two cheap passes (pass 1: column maxima; pass 2: exponentials, sums and the unnormalised ctx; the 1/sum per
// channel d is applied to ctx's rows through a 32-float LDS broadcast) and x is re-read per pass (L2 hits).
// Forward: the four heads' ctx tiles are built first, then one sweep over the blocks finishes each block completely
// (q, out, to_out over the heads, bias, post-norm, residual, store).  Backward: head loop outermost (weight-gradient tiles
// stay in accumulators), per row three sweeps over the blocks; dXh is accumulated in global memory per block.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"

namespace dq {

struct LinAttnBwdLongK {
  const float* x; const float* dyp; float* dxh;
  const float* w_qkv; const float* w_out; const float* g_pre;
  float* part;
  int rows; int units_per_wave;
};

namespace {

template <int C, int NJ>
__device__ __forceinline__ void load_block(const float* __restrict__ src, int64_t row, int N, int pos, int half, bool row_ok, float* out) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = rmap(j, half);
    out[j] = (row_ok && c < C) ? src[(row * C + c) * N + pos] : 0.f;
  }
}

template <int C, int NJ>
__device__ __forceinline__ void prenorm(const float* x, const float* __restrict__ g, int half, float* xh) {
  float ssq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) ssq = fmaf(x[j], x[j], ssq);
  ssq += swap_half(ssq);
  const float inv = sqrtf((float)C) / fmaxf(sqrtf(ssq), RMS_EPS);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = rmap(j, half);
    xh[j] = x[j] * inv * (c < C ? g[c] : 0.f);
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

__device__ __forceinline__ f32x16 q_softmax(f32x16 q, float scale) {
  float m = q[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
  m = fmaxf(m, swap_half(m));
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    q[r] = __expf(q[r] - m);
    s += q[r];
  }
  s += swap_half(s);
  const float qs = scale / s;
#pragma unroll
  for (int r = 0; r < 16; ++r) q[r] *= qs;
  return q;
}

// scale the rows (registers) of t by a lane-indexed vector v (v of lane col = value for row index col)
__device__ __forceinline__ f32x16 scale_rows(f32x16 t, float v, float* tile, int col, int half) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (half == 0) tile[col] = v;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] *= tile[rmap(r, half)];
  return t;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
template <int C, int N>
__global__ void __launch_bounds__(256) k_linattn_fwd_long(LinAttn a) {
  constexpr int NB = N / 32;
  constexpr int NJ = C <= 8 ? 4 : 8;
  __shared__ float wo_lds[4 * 2 * C * 16];
  __shared__ float tiles[4][64];
  for (int i = threadIdx.x; i < 4 * 2 * C * 16; i += blockDim.x) {
    const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = i / (32 * C);
    wo_lds[i] = a.w_out[c * 128 + hd * 32 + rmap(r, hh)];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + wv;
  if (row >= a.rows) return;
  const float scale = 0.17677669529663687f;

  f32x16 ctx[4];
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) {
    float wk[NJ], wv_[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = rmap(j, half);
      wk[j] = c < C ? a.w_qkv[(128 + hd * 32 + col) * C + c] : 0.f;
      wv_[j] = c < C ? a.w_qkv[(256 + hd * 32 + col) * C + c] : 0.f;
    }
    float m = -INFINITY;
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      float x[NJ], xh[NJ];
      load_block<C, NJ>(a.x, row, N, b * 32 + col, half, true, x);
      prenorm<C, NJ>(x, a.g_pre, half, xh);
      const f32x16 kT = proj_a<NJ>(xh, wk);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, kT[r]);
    }
    m = fmaxf(m, swap_half(m));
    float ssum = 0.f;
    f32x16 cx = {0};
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      float x[NJ], xh[NJ];
      load_block<C, NJ>(a.x, row, N, b * 32 + col, half, true, x);
      prenorm<C, NJ>(x, a.g_pre, half, xh);
      f32x16 kT = proj_a<NJ>(xh, wk);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        kT[r] = __expf(kT[r] - m);
        ssum += kT[r];
      }
      const f32x16 vT = proj_a<NJ>(xh, wv_);
      cx = xty(kT, vT, cx);
    }
    ssum += swap_half(ssum);
    ctx[hd] = scale_rows(cx, 1.0f / ssum, tile, col, half);
  }

#pragma unroll 1
  for (int b = 0; b < NB; ++b) {
    const int pos = b * 32 + col;
    float x[NJ], xh[NJ];
    load_block<C, NJ>(a.x, row, N, pos, half, true, x);
    prenorm<C, NJ>(x, a.g_pre, half, xh);
    float ypart[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ypart[c] = 0.f;
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      float wq[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = rmap(j, half);
        wq[j] = c < C ? a.w_qkv[(hd * 32 + col) * C + c] : 0.f;
      }
      const f32x16 q = q_softmax(proj_b<NJ>(wq, xh), scale);
      f32x16 o = {0};
      o = xty(ctx[hd], q, o);
      const float* wl = wo_lds + (hd * 2 + half) * C * 16;
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 t = *reinterpret_cast<const float4*>(wl + c * 16 + r4 * 4);
          ypart[c] = fmaf(t.x, o[r4 * 4 + 0], ypart[c]); ypart[c] = fmaf(t.y, o[r4 * 4 + 1], ypart[c]);
          ypart[c] = fmaf(t.z, o[r4 * 4 + 2], ypart[c]); ypart[c] = fmaf(t.w, o[r4 * 4 + 3], ypart[c]);
        }
      }
    }
    float yv[C];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      yv[c] = ypart[c] + swap_half(ypart[c]) + a.b_out[c];
      ssq = fmaf(yv[c], yv[c], ssq);
    }
    const float inv = sqrtf((float)C) / fmaxf(sqrtf(ssq), RMS_EPS);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c0 = rmap(j, 0), c1 = c0 + 4;
      float lo = 0.f, hi = 0.f, plo = 0.f, phi = 0.f;
      if (c0 < C) { plo = yv[c0 < C ? c0 : 0]; lo = plo * a.g_out[c0 < C ? c0 : 0]; }
      if (c1 < C) { phi = yv[c1 < C ? c1 : 0]; hi = phi * a.g_out[c1 < C ? c1 : 0]; }
      const int c = c0 + 4 * half;
      if (c < C) {
        if (a.ypre) a.ypre[(row * C + c) * N + pos] = half ? phi : plo;
        a.y[(row * C + c) * N + pos] = fmaf(half ? hi : lo, inv, x[j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int C, int N>
__global__ void __launch_bounds__(256, 2) k_linattn_bwd_long(LinAttnBwdLongK a) {  // <= 256 registers: two waves per SIMD
  constexpr int NB = N / 32;
  constexpr int NJ = C <= 8 ? 4 : 8;
  __shared__ float wp_lds[3 * 4 * 2 * C * 16];
  __shared__ float tiles[4][32 * 33];
  for (int i = threadIdx.x; i < 3 * 4 * 2 * C * 16; i += blockDim.x) {
    const int r = i & 15, c = (i >> 4) % C, hh = (i / (16 * C)) & 1, hd = (i / (32 * C)) & 3, m = i / (128 * C);
    wp_lds[i] = a.w_qkv[(m * 128 + hd * 32 + rmap(r, hh)) * C + c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
  float* tile = tiles[wv];
  const int wave_id = blockIdx.x * (blockDim.x >> 6) + wv;
  const int u0 = wave_id * a.units_per_wave;
  if (u0 >= a.rows) return;
  const int u1 = min(a.rows, u0 + a.units_per_wave);
  const float scale = 0.17677669529663687f;

  auto as_acc = [&](const float* xr) {
    f32x16 t = {0};
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = xr[j];
    return t;
  };

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
    f32x16 gq = {0}, gk = {0}, gv = {0}, go = {0};

    // part[c] += sum_r W[m][hd][half][c][r] * t[r], then add this lane-half pair's sum into dxh (first: '=' for head 0)
    auto dxh_add = [&](int64_t row, int pos, int m0, const f32x16& t0, int m1, const f32x16* t1, bool first) {
      float part[C];
#pragma unroll
      for (int c = 0; c < C; ++c) part[c] = 0.f;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        if (which == 1 && t1 == nullptr) break;
        const int m = which ? m1 : m0;
        const f32x16& t = which ? *t1 : t0;
        const float* wl = wp_lds + ((m * 4 + hd) * 2 + half) * C * 16;
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + c * 16 + r4 * 4);
            part[c] = fmaf(w4.x, t[r4 * 4 + 0], part[c]); part[c] = fmaf(w4.y, t[r4 * 4 + 1], part[c]);
            part[c] = fmaf(w4.z, t[r4 * 4 + 2], part[c]); part[c] = fmaf(w4.w, t[r4 * 4 + 3], part[c]);
          }
        }
      }
      float full[C];
#pragma unroll
      for (int c = 0; c < C; ++c) full[c] = part[c] + swap_half(part[c]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = rmap(j, 0), c1 = c0 + 4;
        float lo = 0.f, hi = 0.f;
        if (c0 < C) lo = full[c0 < C ? c0 : 0];
        if (c1 < C) hi = full[c1 < C ? c1 : 0];
        const int c = c0 + 4 * half;
        if (c < C) {
          float* dst = a.dxh + (row * C + c) * N + pos;
          const float val = half ? hi : lo;
          *dst = first ? val : *dst + val;
        }
      }
    };

#pragma unroll 1
    for (int u = u0; u < u1; ++u) {
      const int64_t row = u;
      // ---- sweep 0: column maxima of k
      float m = -INFINITY;
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        float x[NJ], xh[NJ];
        load_block<C, NJ>(a.x, row, N, b * 32 + col, half, true, x);
        prenorm<C, NJ>(x, a.g_pre, half, xh);
        const f32x16 kT = proj_a<NJ>(xh, wk);
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, kT[r]);
      }
      m = fmaxf(m, swap_half(m));
      // ---- sweep 1: sums, ctx
      float ssum = 0.f;
      f32x16 ctx = {0};
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        float x[NJ], xh[NJ];
        load_block<C, NJ>(a.x, row, N, b * 32 + col, half, true, x);
        prenorm<C, NJ>(x, a.g_pre, half, xh);
        f32x16 kT = proj_a<NJ>(xh, wk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          kT[r] = __expf(kT[r] - m);
          ssum += kT[r];
        }
        ctx = xty(kT, proj_a<NJ>(xh, wvv), ctx);
      }
      ssum += swap_half(ssum);
      const float inv_sum = 1.0f / ssum;
      ctx = scale_rows(ctx, inv_sum, tile, col, half);
      const f32x16 ctxT = transpose_tile(ctx, tile, col, half);
      // ---- sweep A: q side, dctx
      f32x16 dctx = {0};
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        const int pos = b * 32 + col;
        float x[NJ], xh[NJ], dy[NJ];
        load_block<C, NJ>(a.x, row, N, pos, half, true, x);
        load_block<C, NJ>(a.dyp, row, N, pos, half, true, dy);
        prenorm<C, NJ>(x, a.g_pre, half, xh);
        const f32x16 q = q_softmax(proj_b<NJ>(wq, xh), scale);
        const f32x16 dO = proj_b<NJ>(wo, dy), dOT = proj_a<NJ>(dy, wo);
        const f32x16 qT = transpose_tile(q, tile, col, half);
        dctx = xty(qT, dOT, dctx);
        f32x16 outT = {0}, dq = {0};
        outT = xty(q, ctx, outT);
        dq = xty(ctxT, dO, dq);
        go = xty(transpose_tile(as_acc(dy), tile, col, half), outT, go);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t = fmaf(q[r], dq[r], t);
        t = (t + swap_half(t)) * (1.0f / scale);
        f32x16 dq_raw;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_raw[r] = q[r] * (dq[r] - t);
        dxh_add(row, pos, 0, dq_raw, 0, nullptr, hd == 0);
        const f32x16 xhT = transpose_tile(as_acc(xh), tile, col, half);
        gq = xty(xhT, transpose_tile(dq_raw, tile, col, half), gq);
      }
      const f32x16 dctxT = transpose_tile(dctx, tile, col, half);
      float delta = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) delta = fmaf(dctxT[r], ctxT[r], delta);
      delta += swap_half(delta);
      // ---- sweep B: k and v sides
#pragma unroll 1
      for (int b = 0; b < NB; ++b) {
        const int pos = b * 32 + col;
        float x[NJ], xh[NJ];
        load_block<C, NJ>(a.x, row, N, pos, half, true, x);
        prenorm<C, NJ>(x, a.g_pre, half, xh);
        f32x16 kT = proj_a<NJ>(xh, wk);
#pragma unroll
        for (int r = 0; r < 16; ++r) kT[r] = __expf(kT[r] - m) * inv_sum;
        const f32x16 v = proj_b<NJ>(wvv, xh);
        f32x16 dkT = {0};
        dkT = xty(v, dctxT, dkT);
        f32x16 dk_rawT;
#pragma unroll
        for (int r = 0; r < 16; ++r) dk_rawT[r] = kT[r] * (dkT[r] - delta);
        const f32x16 xhT = transpose_tile(as_acc(xh), tile, col, half);
        gk = xty(xhT, dk_rawT, gk);
        const f32x16 dk_raw = transpose_tile(dk_rawT, tile, col, half);
        const f32x16 Kd = transpose_tile(kT, tile, col, half);
        f32x16 dv = {0};
        dv = xty(dctx, Kd, dv);
        dxh_add(row, pos, 1, dk_raw, 2, &dv, false);
        gv = xty(xhT, transpose_tile(dv, tile, col, half), gv);
      }
    }
    float* slot = a.part + (int64_t)wave_id * (512 * C);
#pragma unroll
    for (int r = 0; r < NJ; ++r) {
      const int c = rmap(r, half);
      if (c < C) {
        slot[(hd * 32 + col) * C + c] = gq[r];
        slot[(128 + hd * 32 + col) * C + c] = gk[r];
        slot[(256 + hd * 32 + col) * C + c] = gv[r];
        slot[384 * C + c * 128 + hd * 32 + col] = go[r];
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
  set_error("linattn_fwd: (C, n) = (" + std::to_string(a.C) + ", " + std::to_string(a.n) + ") is not built");
  return 2;
}

int launch_linattn_bwd_long(const float* x, const float* dyp, float* dxh, const float* w_qkv, const float* w_out, const float* g_pre,
                            float* part, int C, int rows, int n, int* waves_out, hipStream_t s) {
  // two waves per SIMD are resident (<= 256 registers): 2048 waves = one resident round, and <= LA_MAX_WAVES partial slots
  LinAttnBwdLongK k{x, dyp, dxh, w_qkv, w_out, g_pre, part, rows, std::max(1, cdiv(rows, 2048))};
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
  set_error("linattn_bwd: (C, n) = (" + std::to_string(C) + ", " + std::to_string(n) + ") is not built");
  return 2;
}

}  // namespace dq
