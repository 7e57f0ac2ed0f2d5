// Residual(PreNorm(LinearAttention)) FORWARD over m/z rows of 2 / 4 positions at 8 / 12 / 16 channels (reference
// dquartic/model/unet1d.py:446-496), one m/z row per lane column on v_mfma_f32_16x16x4_f32 -- the forward half of k_la_rows_bwd.hip, same
// layout, same re-associated algebra (W2 = Wo_h Wv_h; S[n][m] = sum_d q[d][n] k[d][m]; Z[:, n] = sum_m S[n][m] xh[:, m]; y_pre = sum_h W2_h Z_h + b):
//   lane = (g = lane / 16, row = lane % 16), a wave = 16 rows, register r of a lane = channel CPL g + r (CPL = C / 4) of each of the row's N
//   positions; a workgroup = four waves = the four heads of the same 16-row tile.  q / k projections and W2 Z are MFMAs whose B operand is
//   the register holding the previous result; both softmaxes, the N x N scalars of a row and the two RMSNorms are per-lane arithmetic + a sum
//   over the four lane groups.  The heads' y_pre contributions meet in LDS behind one barrier; wave h finishes positions n = h (mod 4): bias,
//   post-norm, residual, stores.
// The register-resident k_linattn_fwd spends a 32-position unit on 8 - 16 such rows (block-diagonal 32 x 32 tiles); k_la_small (lane =
// (channel half, row), 32x32x2 MFMAs) wins from one tile per SIMD on (la_small_min_rows).  This kernel takes the training batch sizes in
// between: launch_linattn_fwd, whenever the rows backward is in use (option la_rows_bwd_min_rows) below that threshold.
// Operands: the first 16 floats per (head, lane) of the rows-backward image (k_linattn_prepare: [q: 8 | k: 8], log2(e) folded in) and the
// plain W2 of the prepared weights.
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
#include "dq_options.h"
#include <algorithm>
#include <cstdint>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float gsum4(float t) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = __int_as_float(a[0]) + __int_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}
__device__ __forceinline__ float gmax4(float t) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = fmaxf(__int_as_float(a[0]), __int_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return fmaxf(__int_as_float(b[0]), __int_as_float(b[1]));
}

struct LaRowsFwdK {
  const float* x; float* y; float* ypre;  // ypre: nullable (saved for the backward)
  const float* prep; const float* g_pre; const float* g_out; const float* b_out;
  int rows, ntiles;
};

template <int C, int N>
__global__ void __launch_bounds__(256) k_la_rows_fwd(LaRowsFwdK a) {
  static_assert(C == 8 || C == 12 || C == 16, "channel widths of the deep levels");
  static_assert(N == 2 || N == 4, "rows of 2 / 4 positions");
  constexpr int CPL = C / 4, RUN = CPL * N;
  constexpr int VW = RUN % 4 == 0 ? 4 : 2, NV = RUN / VW;
  constexpr int LS = LA_ROWS_LANE_FLOATS;
  constexpr float scale = 0.17677669529663687f;  // dim_head^-0.5 (unet1d.py:481)
  typedef float vecf __attribute__((ext_vector_type(VW)));
  // the heads' y_pre contributions of a tile: [head][position][lane][4]
  __shared__ __attribute__((aligned(16))) float ex[4 * N * 64 * 4];
  const int lane = threadIdx.x & 63, g = lane >> 4, row = lane & 15, hd = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this head's A operands of this lane: q | k projections from the image; W2_head[out channel c(row)][in channel CPL g + s]
  float aq[8], ak[8], aw2[4];
  {
    const float4* wp = reinterpret_cast<const float4*>(a.prep + LA_PREP_ROWS + (hd * 64 + lane) * LS);
    const float4 v0 = wp[0], v1 = wp[1], v2 = wp[2], v3 = wp[3];
    aq[0] = v0.x; aq[1] = v0.y; aq[2] = v0.z; aq[3] = v0.w; aq[4] = v1.x; aq[5] = v1.y; aq[6] = v1.z; aq[7] = v1.w;
    ak[0] = v2.x; ak[1] = v2.y; ak[2] = v2.z; ak[3] = v2.w; ak[4] = v3.x; ak[5] = v3.y; ak[6] = v3.z; ak[7] = v3.w;
    const int co = (row & 3) < CPL ? CPL * (row >> 2) + (row & 3) : -1;  // the channel of output row `row` of an M = C product
#pragma unroll
    for (int s = 0; s < 4; ++s) aw2[s] = (co >= 0 && s < CPL) ? a.prep[(hd * C + co) * C + CPL * g + (s < CPL ? s : 0)] : 0.f;
  }
  const bool bounded = a.prep[LA_PREP_BOUNDED] != 0.f;
  float gpre[CPL], gout[CPL], bo[CPL];
#pragma unroll
  for (int r = 0; r < CPL; ++r) { gpre[r] = a.g_pre[CPL * g + r]; gout[r] = a.g_out[CPL * g + r]; bo[r] = a.b_out[CPL * g + r]; }
  const float sqC = sqrtf((float)C);

#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int grow = tile * 16 + row;
    const bool live = grow < a.rows;
    const int64_t off = ((int64_t)(live ? grow : a.rows - 1) * C + CPL * g) * N;  // the lane's run: channels CPL g .., all positions
    float xr[RUN];
    {
      vecf vx[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) vx[k] = *reinterpret_cast<const vecf*>(a.x + off + k * VW);
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) xr[k * VW + e] = live ? vx[k][e] : 0.f;
    }
    // ---- PreNorm (unet1d.py:140, 171)
    float xh[N][CPL];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) ssq = fmaf(xr[r * N + n], xr[r * N + n], ssq);
      const float inv = rms_inv(gsum4(ssq), sqC);
#pragma unroll
      for (int r = 0; r < CPL; ++r) xh[n][r] = xr[r * N + n] * inv * gpre[r];
    }
    // ---- k of every position; softmax over the positions, in the lane (unet1d.py:479)
    f32x4 kk[N][2];
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) acc = mfma16(ak[t * CPL + s], xh[m][s], acc);
        kk[m][t] = acc;
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mx = 0.f;
        if (!bounded) {
          mx = kk[0][t][r];
#pragma unroll
          for (int m = 1; m < N; ++m) mx = fmaxf(mx, kk[m][t][r]);
        }
        float z = 0.f;
#pragma unroll
        for (int m = 0; m < N; ++m) { kk[m][t][r] = __builtin_amdgcn_exp2f(kk[m][t][r] - mx); z += kk[m][t][r]; }
        const float rz = fast_rcp(z);
#pragma unroll
        for (int m = 0; m < N; ++m) kk[m][t][r] *= rz;
      }
    // ---- per position n: q (softmax over the head's 32 channels: 8 in the lane, the rest in the other lane groups; * 32^-0.5), the row of S,
    //      Z[:, n] and this head's y_pre contribution W2 Z[:, n]
#pragma unroll
    for (int n = 0; n < N; ++n) {
      f32x4 qs[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) acc = mfma16(aq[t * CPL + s], xh[n][s], acc);
        qs[t] = acc;
      }
      float mx = 0.f;
      if (!bounded) {
        mx = qs[0][0];
#pragma unroll
        for (int e = 1; e < 8; ++e) mx = fmaxf(mx, qs[e >> 2][e & 3]);
        mx = gmax4(mx);
      }
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { qs[e >> 2][e & 3] = __builtin_amdgcn_exp2f(qs[e >> 2][e & 3] - mx); sum += qs[e >> 2][e & 3]; }
      const float sc = scale * fast_rcp(gsum4(sum));
      float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < N; ++m) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(qs[e >> 2][e & 3], kk[m][e >> 2][e & 3], t);
        const float S = gsum4(t) * sc;  // (the q normalisation applied to the scalar, not to the eight registers)
#pragma unroll
        for (int r = 0; r < CPL; ++r) z[r] = fmaf(S, xh[m][r], z[r]);
      }
      f32x4 yh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < CPL; ++s) yh = mfma16(aw2[s], z[s], yh);
      *reinterpret_cast<float4*>(ex + ((hd * N + n) * 64 + lane) * 4) = make_float4(yh[0], yh[1], yh[2], yh[3]);
    }
    lds_barrier();
    // ---- wave h finishes positions n = h (mod 4): the four heads in head order, bias, post-norm (unet1d.py:470-473), residual
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if ((n & 3) == hd) {  // (wave-uniform)
        float yp[CPL];
        {
          const float4 v0 = *reinterpret_cast<const float4*>(ex + ((0 * N + n) * 64 + lane) * 4), v1 = *reinterpret_cast<const float4*>(ex + ((1 * N + n) * 64 + lane) * 4);
          const float4 v2 = *reinterpret_cast<const float4*>(ex + ((2 * N + n) * 64 + lane) * 4), v3 = *reinterpret_cast<const float4*>(ex + ((3 * N + n) * 64 + lane) * 4);
          const float t[4] = {((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y, ((v0.z + v1.z) + v2.z) + v3.z, ((v0.w + v1.w) + v2.w) + v3.w};
#pragma unroll
          for (int r = 0; r < CPL; ++r) yp[r] = t[r] + bo[r];
        }
        float usq = 0.f;
#pragma unroll
        for (int r = 0; r < CPL; ++r) usq = fmaf(yp[r], yp[r], usq);
        const float uinv = rms_inv(gsum4(usq), sqC);
        if (live) {
#pragma unroll
          for (int r = 0; r < CPL; ++r) {
            a.y[off + r * N + n] = xr[r * N + n] + yp[r] * uinv * gout[r];
            if (a.ypre) a.ypre[off + r * N + n] = yp[r];
          }
        }
      }
    }
    lds_barrier();  // (the next tile's contributions go to the same buffer)
  }
}

template <int C, int N>
int la_rows_fwd_blocks() {
  static const int v = [] {
    int occ = 1, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_la_rows_fwd<C, N>, 256, 0) != hipSuccess) occ = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return std::max(1, occ) * std::max(1, cus);
  }();
  return v;
}

}  // namespace

bool la_rows_fwd_usable(int C, int n) {
  if (DQ_DEV_FLAG("DQ_NO_LA_ROWS_FWD", '1')) return false;  // (dev switch)
  return (C == 8 || C == 12 || C == 16) && (n == 2 || n == 4);
}

int launch_la_rows_fwd(const LinAttn& a, hipStream_t s) {
  DQ_REQUIRE(a.prep && la_rows_fwd_usable(a.C, a.n) && a.x && a.y && a.g_pre && a.g_out && a.b_out, "la_rows_fwd: unsupported shape or missing operand");
  DQ_REQUIRE((((uintptr_t)a.x | (uintptr_t)a.prep) & 15) == 0, "la_rows_fwd: 16-byte aligned x and prepared weights");
  if (a.rows == 0) return 0;
  LaRowsFwdK k{a.x, a.y, a.ypre, a.prep, a.g_pre, a.g_out, a.b_out, a.rows, (int)cdiv(a.rows, 16)};
#define DQ_LRF(CC, NN)                                                                              \
  if (a.C == CC && a.n == NN) {                                                                     \
    const int grid = std::min(k.ntiles, la_rows_fwd_blocks<CC, NN>());                              \
    hipLaunchKernelGGL((k_la_rows_fwd<CC, NN>), dim3(grid), dim3(256), 0, s, k);                    \
    DQ_LAUNCH_CHECK();                                                                              \
    return 0;                                                                                       \
  }
  DQ_LRF(8, 2) DQ_LRF(8, 4) DQ_LRF(12, 2) DQ_LRF(12, 4) DQ_LRF(16, 2) DQ_LRF(16, 4)
#undef DQ_LRF
  set_error("la_rows_fwd: unsupported (C, n)");
  return 2;
}

}  // namespace dq
