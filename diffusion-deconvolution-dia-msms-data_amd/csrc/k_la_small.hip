// Residual(PreNorm(LinearAttention)) (reference dquartic/model/unet1d.py:446-496, 64-79, 143-176) for rows of 2 / 4 positions at 8 / 12 /
// 16 channels -- the deep levels, where the register-resident kernel of k_linattn.hip spends a 32-position unit (block-diagonal 32 x 32
// tiles, C-row chains on the 4x4x1 pipe) on 8..16 rows.  Here every product is a v_mfma_f32_32x32x2_f32 without padding in the row
// dimension:
//   wave = 32 rows; lane = (half, row); ONE GROUP of registers per position m < N
//   projection operand of position m, K-step i: half h supplies xhat[sm_chan(i, h)][m] of the lane's row          (C / 2 steps)
//   q_m, k_m (32 head channels x 32 rows per head) = sum_i mfma(W image, xhat_m[i])  -- accumulator layout: 16 of a head's 32 channels per lane
//   k.softmax(dim = -1) runs over the N groups IN the lane; q.softmax(dim = -2) over 16 registers + the other half (one ds_bpermute)
//   S[n][m] = sum_d q[d][n] k[d][m]  (N^2 scalars per row and head: 16 FMAs + one exchange each)
//   the reference's context / out / to_out chain re-associates exactly to  y_pre[:, n] = sum_head W2_head (sum_m S[n][m] xhat[:, m]) + b,
//   W2_head = Wo_head Wv_head (C x C, prepared):  the inner sum is per-lane arithmetic on the projection operands, the outer one C / 2
//   MFMAs per head and position.
// 12 N C / 2 MFMAs per 32 rows (N = 4, C = 12: 288 for 128 positions), against ~1,000 4x4x1 + 32x32x2 MFMAs per 32-position unit before.
// Softmax in the log2 domain (the image holds W log2 e) and without the shift by the maximum when the layer's logits are bounded
// (LA_PREP_BOUNDED), as in k_linattn.hip.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include "dq_options.h"
#include <climits>
#include <algorithm>
#include <cstdlib>

namespace dq {

namespace {

struct LaSmallK {
  const float* x; float* y; float* ypre; const float* prep; const float* b_out; const float* g_pre; const float* g_out;
  int rows, ntiles;
};

template <int C, int N>
__global__ void __launch_bounds__(256, N <= 4 ? 2 : 1) k_la_small(LaSmallK a) {  // (N = 8: 128 registers of k tiles alone -- one wave per SIMD)
  constexpr int S = C / 2;
  constexpr int NR = C == 8 ? 4 : 8;  // output / residual registers per lane: channel rmap(j, half), valid below C
  __shared__ __attribute__((aligned(16))) float img[12 * S * 64];
  __shared__ float prm[3 * 16];
  // wave-private x tile (32 rows x (C N + 1) floats): the rows are read as contiguous 16-byte runs and stay here for the residual; y (and
  // y_pre) leave through it the same way.  As per-lane 4-byte accesses at a row pitch of C N floats every load / store instruction touched
  // 32..64 cache lines: ~90 such instructions per tile were a third of the launch at the sampling batch.
  constexpr int E = C * N, EP = E + 1;
  extern __shared__ __attribute__((aligned(16))) float xt_all[];  // 4 x 32 x EP floats (dynamic: with the image it passes 64 KB at N = 8)
  float* xt = xt_all + (threadIdx.x >> 6) * 32 * EP;
  {
    constexpr int T4 = 12 * S * 16, NLD = (T4 + 255) / 256;
    const float4* src = reinterpret_cast<const float4*>(a.prep + LA_PREP_SMALL);
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) { const int i = u * 256 + (int)threadIdx.x; v[u] = src[i < T4 ? i : 0]; }
    float pv = 0.f;
    if (threadIdx.x < 48) {
      const int w = threadIdx.x >> 4, c = threadIdx.x & 15;
      const float* q = w == 0 ? a.g_pre : (w == 1 ? a.b_out : a.g_out);
      pv = c < C ? q[c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) { const int i = u * 256 + (int)threadIdx.x; if (i < T4) reinterpret_cast<float4*>(img)[i] = v[u]; }
    if (threadIdx.x < 48) prm[threadIdx.x] = pv;
  }
  const bool bounded = a.prep[LA_PREP_BOUNDED] != 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const float sqC = sqrtf((float)C);
  const float* il = img + lane;
  auto chain = [&](int g, const float* b, f32x16 acc) __attribute__((always_inline)) -> f32x16 {
#pragma unroll
    for (int i = 0; i < S; ++i) acc = mfma_f32(il[(g * S + i) * 64], b[i], acc);
    return acc;
  };
  float gp[S];
#pragma unroll
  for (int i = 0; i < S; ++i) gp[i] = prm[sm_chan(C, i, half)];

#pragma unroll 1
  for (int tile = wid; tile < a.ntiles; tile += nwaves) {
    const int row = tile * 32 + col;
    const bool live = row < a.rows;
    const int valid = a.rows - tile * 32 < 32 ? a.rows - tile * 32 : 32;
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    {
      const float4* g4 = reinterpret_cast<const float4*>(a.x + (int64_t)tile * 32 * E);  // (32 E floats from a 128-byte aligned tile start)
      const int lim4 = valid * E / 4;  // E is a multiple of 4
      float4 raw[E / 8];
#pragma unroll
      for (int k = 0; k < E / 8; ++k) { const int i4 = k * 64 + lane; raw[k] = g4[i4 < lim4 ? i4 : lim4 - 1]; }
      __builtin_amdgcn_sched_barrier(0);
      wsync();  // (the previous tile's readers of xt are done)
#pragma unroll
      for (int k = 0; k < E / 8; ++k) {
        const int lin = (k * 64 + lane) * 4, r = lin / E, e = lin - r * E;  // four consecutive floats of one row (E % 4 == 0)
        float* d = xt + r * EP + e;
        d[0] = raw[k].x; d[1] = raw[k].y; d[2] = raw[k].z; d[3] = raw[k].w;
      }
      wsync();
    }
    const float* xp = xt + col * EP;  // the lane's row: element (c, m) at c * N + m
    float xh[N][S];
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int i = 0; i < S; ++i) xh[m][i] = xp[sm_chan(C, i, half) * N + m];
    // PreNorm (unet1d.py:140, 171)
#pragma unroll
    for (int m = 0; m < N; ++m) {
      float ssq = 0.f;
#pragma unroll
      for (int i = 0; i < S; ++i) ssq = fmaf(xh[m][i], xh[m][i], ssq);
      ssq += swap_half(ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int i = 0; i < S; ++i) xh[m][i] = xh[m][i] * inv * gp[i];
    }
    float yp[N][NR];  // (a product tile's rows 16..31 are padding: only registers 0..7 are kept between heads)
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int j = 0; j < NR; ++j) yp[n][j] = 0.f;
#pragma unroll 1
    for (int hd = 0; hd < 4; ++hd) {
      // k of every position, softmax over the positions (unet1d.py:479)
      f32x16 ks[N];
#pragma unroll
      for (int m = 0; m < N; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ks[m][r] = 0.f;
        ks[m] = chain(4 + hd, xh[m], ks[m]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float mx = 0.f;
        if (!bounded) {
          mx = ks[0][r];
#pragma unroll
          for (int m = 1; m < N; ++m) mx = fmaxf(mx, ks[m][r]);
        }
        float z = 0.f;
#pragma unroll
        for (int m = 0; m < N; ++m) { ks[m][r] = __builtin_amdgcn_exp2f(ks[m][r] - mx); z += ks[m][r]; }
        const float rz = fast_rcp(z);
#pragma unroll
        for (int m = 0; m < N; ++m) ks[m][r] *= rz;
      }
      // q of one position at a time: softmax over the head's 32 channels (unet1d.py:478, 481), then its row of S
      float Sm[N][N];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        f32x16 q;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = 0.f;
        q = chain(hd, xh[n], q);
        float mx = 0.f;
        if (!bounded) {
          mx = q[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
          mx = fmaxf(mx, swap_half(mx));
        }
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { q[r] = __builtin_amdgcn_exp2f(q[r] - mx); sum += q[r]; }
        sum += swap_half(sum);
        const float sc = 0.17677669529663687f * fast_rcp(sum);  // softmax, then * dim_head^-0.5
#pragma unroll
        for (int m = 0; m < N; ++m) {
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) t = fmaf(q[r], ks[m][r], t);
          t *= sc;
          Sm[n][m] = t + swap_half(t);
        }
      }
      // y_pre[:, n] += W2_head (sum_m S[n][m] xhat[:, m])
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float zg[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
          float t = 0.f;
#pragma unroll
          for (int m = 0; m < N; ++m) t = fmaf(Sm[n][m], xh[m][i], t);
          zg[i] = t;
        }
        f32x16 t16;
#pragma unroll
        for (int r = 0; r < 16; ++r) t16[r] = 0.f;
        t16 = chain(8 + hd, zg, t16);
#pragma unroll
        for (int j = 0; j < NR; ++j) yp[n][j] += t16[j];
      }
    }
    // to_out bias, RMSNorm, residual (unet1d.py:470-473, 495, 79): y replaces x in the staging tile element by element (every (row, c, n) is
    // read and written by exactly one lane), then the tile leaves as contiguous 16-byte runs; y_pre (training) takes the same way first
    float yv[N][NR];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float ssq = 0.f;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int co = rmap(j, half);
        const bool ok = co < C;
        yp[n][j] = ok ? yp[n][j] + prm[16 + (ok ? co : 0)] : 0.f;
        ssq = fmaf(yp[n][j], yp[n][j], ssq);
      }
      ssq += swap_half(ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int co = rmap(j, half);
        yv[n][j] = co < C ? fmaf(yp[n][j] * inv, prm[32 + co], xp[co * N + n]) : 0.f;
      }
    }
    auto flush = [&](float* dst, float (&val)[N][NR]) __attribute__((always_inline)) {
      wsync();
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int co = rmap(j, half);
          if (co < C) xt[col * EP + co * N + n] = val[n][j];
        }
      wsync();
      float4* g4 = reinterpret_cast<float4*>(dst + (int64_t)tile * 32 * E);
      const int lim4 = valid * E / 4;
#pragma unroll
      for (int k = 0; k < E / 8; ++k) {
        const int i4 = k * 64 + lane, lin = i4 * 4, r = lin / E, e = lin - r * E;
        const float* sp = xt + r * EP + e;
        const float4 o = make_float4(sp[0], sp[1], sp[2], sp[3]);
        if (i4 < lim4) g4[i4] = o;
      }
    };
    if (a.ypre) flush(a.ypre, yp);
    flush(a.y, yv);
    (void)live;
  }
}


}  // namespace

bool la_small_usable(int C, int n) {
  return ((n == 2 || n == 4) && (C == 12 || C == 16)) || (n == 8 && C == 12);  // (8 channels at 8 positions measured slower than the register-resident kernel: 297 vs 276 us at batch 512; 12 channels: 371 vs 429)
}
// Below one 32-row tile per SIMD the launch is a latency chain (a training batch of 32 windows: 400 tiles for 1,024 SIMDs) and the
// register-resident kernel's shorter prologue wins (measured at 12,800 rows: 4 launches +35 us per step); above it the matrix pipe is the limit
// and this form's ~3x fewer MFMA cycles per position pay.  The rule comes from the device (4 SIMDs per compute unit x 32 rows);
// dq_set_option("la_small_min_rows", rows) overrides it (tests run both forms at every size).
int la_small_min_rows() {
  const int64_t o = option(OPT_LA_SMALL_MIN_ROWS);
  if (o >= 0) return (int)std::min<int64_t>(o, INT32_MAX);
  static const int rule = [] { int d = 0; hipDeviceProp_t pr; return 32 * 4 * ((hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256); }();
  return rule;
}

int launch_la_small_fwd(const LinAttn& a, hipStream_t s) {
  DQ_REQUIRE(a.x && a.y && a.prep && a.b_out && a.g_pre && a.g_out && la_small_usable(a.C, a.n), "la_small: missing operand / unsupported shape");
  DQ_REQUIRE((((uintptr_t)a.prep | (uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.ypre) & 15) == 0, "la_small: misaligned tensor / prepared-weights buffer");
  if (a.rows == 0) return 0;
  LaSmallK k{a.x, a.y, a.ypre, a.prep, a.b_out, a.g_pre, a.g_out, a.rows, cdiv(a.rows, 32)};
  static const int cus = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
#define DQ_LAS(CC, NN)                                                                                        \
  if (a.C == CC && a.n == NN) {                                                                               \
    const size_t lds = (size_t)4 * 32 * (CC * NN + 1) * 4;                                                   \
    const int nb = occ_blocks_per_cu((const void*)k_la_small<CC, NN>, 256, lds);                              \
    if (nb < 0) return 1;                                                                                     \
    const int grid = std::max(1, std::min(nb * cus, (k.ntiles + 3) / 4));                                     \
    hipLaunchKernelGGL((k_la_small<CC, NN>), dim3(grid), dim3(256), lds, s, k);                               \
    DQ_LAUNCH_CHECK();                                                                                        \
    return 0;                                                                                                 \
  }
  DQ_LAS(12, 2) DQ_LAS(16, 2) DQ_LAS(12, 4) DQ_LAS(16, 4) DQ_LAS(12, 8)
#undef DQ_LAS
  set_error("la_small: unsupported (C, n)");
  return 2;
}

}  // namespace dq
