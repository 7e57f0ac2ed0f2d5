// The bottleneck of UNet1d when it is WIDE: mid_dim * downsampled_n channels over the RT axis (reference unet1d.py:1027-1058,
// 1144-1148).  At the BASELINE shapes that is 16 / 64 channels and the per-thread "all channels in registers" kernels of
// k_conv.hip apply; at the reference's shipped configuration (downsample_dim 40000 -> 625 positions x 16 channels) it is 10,000
// channels, and 80 at an odd test shape (MZ = 320).  For those the two ResnetBlocks and the attention projections run as
//   im2col (k = 3, zero padded)  ->  fp32 matrix-core GEMM (k_gemm.hip)  ->  RMSNorm over the channel axis + scale/shift + SiLU
// with the pieces below.  Tensors are (B, C, P): P = the RT axis padded to a multiple of 4 floats (the GEMM reads 16-byte
// vectors along rows), t < RT valid.  Every kernel here writes ZERO into the pad columns of its outputs.
// All reductions are fixed-order (no atomics).
#include "dq_common.h"
#include "dq_kernels.h"

namespace dq {

#define DQ_TRY_RC(expr)         \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

// xcol[b][c*3 + k][t] = x[b][c][t + k - 1] (zero outside [0, RT)) -- Conv1d(k3, p1) as a product with W viewed as (cout, 3 cin)
__global__ void __launch_bounds__(256) k_im2col3(const float* __restrict__ x, float* __restrict__ xcol, int C, int RT, int P, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % P);
  const int64_t rc = i / P;  // b * 3C + c*3 + k
  const int k = (int)(rc % 3);
  const int64_t bc = rc / 3;  // b * C + c
  const int q = t + k - 1;
  xcol[i] = (t < RT && q >= 0 && q < RT) ? x[bc * P + q] : 0.f;
}
int launch_im2col3(const float* x, float* xcol, int B, int C, int RT, int P, hipStream_t s) {
  const int64_t total = (int64_t)B * 3 * C * P;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_im2col3, dim3(cdiv(total, 256)), dim3(256), 0, s, x, xcol, C, RT, P, total);
  DQ_LAUNCH_CHECK();
  return 0;
}

// dx[b][c][t] (+)= sum_k dxcol[b][c*3 + k][t - k + 1] : the transpose of the above
__global__ void __launch_bounds__(256) k_col2im3(const float* __restrict__ dxcol, float* __restrict__ dx, int RT, int P, int accumulate,
                                                 int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % P);
  const int64_t bc = i / P;
  float v = 0.f;
  if (t < RT) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int p = t - k + 1;  // output position whose tap k reads input t
      if (p >= 0 && p < RT) v += dxcol[(bc * 3 + k) * P + p];
    }
    if (accumulate) v += dx[i];
  }
  dx[i] = v;
}
int launch_col2im3(const float* dxcol, float* dx, int B, int C, int RT, int P, int accumulate, hipStream_t s) {
  const int64_t total = (int64_t)B * C * P;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_col2im3, dim3(cdiv(total, 256)), dim3(256), 0, s, dxcol, dx, RT, P, accumulate, total);
  DQ_LAUNCH_CHECK();
  return 0;
}

// dst (rows, dpitch) <- src (rows, spitch), n valid columns, the rest of a dst row zero
__global__ void __launch_bounds__(256) k_repitch(float* __restrict__ dst, int dpitch, const float* __restrict__ src, int spitch, int n,
                                                 int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % dpitch);
  const int64_t r = i / dpitch;
  dst[i] = t < n ? src[r * spitch + t] : 0.f;
}
int launch_repitch(float* dst, int dpitch, const float* src, int spitch, int64_t rows, int n, hipStream_t s) {
  const int64_t total = rows * dpitch;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_repitch, dim3(cdiv(total, 256)), dim3(256), 0, s, dst, dpitch, src, spitch, n, total);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// RMSNorm over the channel axis of (B, C, P) [-> (scale + 1, shift) of sample b -> SiLU] [+ residual]
//   y[b][c][t] = act( u / max(||u[b][:][t]||, eps) * g[c] * sqrt(C) * (ss[b][c] + 1) + ss[b][C + c] ) + res[b][c][t]
// Block = (b, 32 positions): 8 channel slices x 32 position lanes; the slices meet in LDS in slice order.
// ---------------------------------------------------------------------------------------------------------------
struct WideNorm {
  const float* u; const float* g; const float* ss; int ss_stride; int act; const float* res; float* y;
  int B, C, RT, P;
};
__global__ void __launch_bounds__(256) k_wnorm_fwd(WideNorm a) {
  __shared__ float red[8][33];
  const int b = blockIdx.y, tl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + tl;
  const bool ok = t < a.RT;
  const int64_t base = (int64_t)b * a.C * a.P + t;
  float ssq = 0.f;
  if (ok)
    for (int c = sl; c < a.C; c += 8) { const float v = a.u[base + (int64_t)c * a.P]; ssq = fmaf(v, v, ssq); }
  red[sl][tl] = ssq;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += red[k][tl];
  const float inv = sqrtf((float)a.C) / fmaxf(sqrtf(tot), RMS_EPS);
  if (t >= a.P) return;
  for (int c = sl; c < a.C; c += 8) {
    const int64_t o = base + (int64_t)c * a.P;
    float v = 0.f;
    if (ok) {
      v = a.u[o] * inv * a.g[c];
      if (a.ss) v = fmaf(v, a.ss[(int64_t)b * a.ss_stride + c] + 1.0f, a.ss[(int64_t)b * a.ss_stride + a.C + c]);
      if (a.act == ACT_SILU) v = silu_f(v);
      if (a.res) v += a.res[o];
    }
    a.y[o] = v;
  }
}
int launch_wnorm_fwd(const float* u, const float* g, const float* ss, int ss_stride, int act, const float* res, float* y, int B, int C,
                     int RT, int P, hipStream_t s) {
  if (B == 0 || C == 0) return 0;
  WideNorm a{u, g, ss, ss_stride, act, res, y, B, C, RT, P};
  hipLaunchKernelGGL(k_wnorm_fwd, dim3(cdiv(P, 32), B), dim3(256), 0, s, a);
  DQ_LAUNCH_CHECK();
  return 0;
}

// Backward, same arithmetic as k_block_bwd (k_conv.hip) but with the channel axis too long for registers: two kernels.
//   (1) per (b, t): inv = 1 / max(||u||, eps) and dot = sum_c gd[c] uh[c]  (gd = dy act'(w) (scale+1) g sqrt(C), uh = u inv)
//   (2) per (b, c): du[b][c][t] for every t, and this (b, c)'s sums over t of d scale, d shift, d g and du (a conv bias gradient)
struct WideNormBwd {
  const float* u; const float* dy; const float* g; const float* ss; int ss_stride; int act;
  float* stats;  // (B, 2, P): [inv | dot]
  float* du;     // (B, C, P), pad columns zeroed
  float* dss;    // += (B, ss_stride): [d scale (C) | d shift (C)] of this block's slice (nullable when ss is null)
  float* dgp;    // (B, C) partial d g (summed over b by launch_sum_b)
  float* dbp;    // (B, C) partial sum of du over t (nullable)
  int B, C, RT, P;
};
__device__ __forceinline__ float wide_dw(const WideNormBwd& a, int b, int c, float uh, float dyv, float& z, float& sc) {
  z = uh * a.g[c] * sqrtf((float)a.C);
  sc = a.ss ? a.ss[(int64_t)b * a.ss_stride + c] + 1.0f : 1.0f;
  const float sh = a.ss ? a.ss[(int64_t)b * a.ss_stride + a.C + c] : 0.f;
  const float w = fmaf(z, sc, sh);
  return a.act == ACT_SILU ? dyv * silu_grad_f(w) : dyv;
}
__global__ void __launch_bounds__(256) k_wnorm_bwd_stats(WideNormBwd a) {
  __shared__ float red[8][33];
  const int b = blockIdx.y, tl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + tl;
  const bool ok = t < a.RT;
  const int64_t base = (int64_t)b * a.C * a.P + t;
  float ssq = 0.f;
  if (ok)
    for (int c = sl; c < a.C; c += 8) { const float v = a.u[base + (int64_t)c * a.P]; ssq = fmaf(v, v, ssq); }
  red[sl][tl] = ssq;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += red[k][tl];
  __syncthreads();
  const float inv = 1.0f / fmaxf(sqrtf(tot), RMS_EPS);
  const float sqC = sqrtf((float)a.C);
  float dot = 0.f;
  if (ok)
    for (int c = sl; c < a.C; c += 8) {
      const int64_t o = base + (int64_t)c * a.P;
      const float uh = a.u[o] * inv;
      float z, sc;
      const float dw = wide_dw(a, b, c, uh, a.dy[o], z, sc);
      dot = fmaf(dw * sc * a.g[c] * sqC, uh, dot);
    }
  red[sl][tl] = dot;
  __syncthreads();
  if (sl == 0 && t < a.P) {
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += red[k][tl];
    a.stats[((int64_t)b * 2 + 0) * a.P + t] = ok ? inv : 0.f;
    a.stats[((int64_t)b * 2 + 1) * a.P + t] = ok ? d : 0.f;
  }
}
// block = (b, 4 channels): one wave per channel, lanes over t; the wave sum (fixed DPP order) closes the per-(b, c) sums
__global__ void __launch_bounds__(256) k_wnorm_bwd_apply(WideNormBwd a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.C) return;
  const float sqC = sqrtf((float)a.C);
  const float* inv = a.stats + ((int64_t)b * 2 + 0) * a.P;
  const float* dot = a.stats + ((int64_t)b * 2 + 1) * a.P;
  const int64_t base = ((int64_t)b * a.C + c) * a.P;
  float dsc = 0.f, dsh = 0.f, dg = 0.f, dbs = 0.f;
  for (int t = lane; t < a.P; t += 64) {
    float out = 0.f;
    if (t < a.RT) {
      const float iv = inv[t];
      const float uh = a.u[base + t] * iv;
      float z, sc;
      const float dw = wide_dw(a, b, c, uh, a.dy[base + t], z, sc);
      dsh += dw;
      dsc = fmaf(dw, z, dsc);
      const float dz = dw * sc;
      dg = fmaf(dz, uh * sqC, dg);
      const float gd = dz * a.g[c] * sqC;
      const bool clamped = iv >= 1.0f / RMS_EPS;  // the norm was below eps: F.normalize is linear there
      out = clamped ? gd * iv : iv * (gd - uh * dot[t]);
      dbs += out;
    }
    a.du[base + t] = out;
  }
  dsc = wave_sum(dsc); dsh = wave_sum(dsh); dg = wave_sum(dg); dbs = wave_sum(dbs);
  if (lane == 0) {
    if (a.dss) { a.dss[(int64_t)b * a.ss_stride + c] += dsc; a.dss[(int64_t)b * a.ss_stride + a.C + c] += dsh; }
    a.dgp[(int64_t)b * a.C + c] = dg;
    if (a.dbp) a.dbp[(int64_t)b * a.C + c] = dbs;
  }
}
// dst[c] += sum_b part[b][c], b in order
__global__ void __launch_bounds__(256) k_sum_b(const float* __restrict__ part, int B, int C, float* __restrict__ dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[(int64_t)b * C + c];
  dst[c] += s;
}
int launch_sum_b(const float* part, int B, int C, float* dst, hipStream_t s) {
  if (B == 0 || C == 0) return 0;
  hipLaunchKernelGGL(k_sum_b, dim3(cdiv(C, 256)), dim3(256), 0, s, part, B, C, dst);
  DQ_LAUNCH_CHECK();
  return 0;
}
// du, d g (+=), per-sample d(scale, shift) (+=) and optionally a bias gradient (+= sum of du over (b, t)).
// scratch: B * (2 P + 2 C) floats
int launch_wnorm_bwd(const float* u, const float* dy, const float* g, const float* ss, int ss_stride, int act, float* du, float* dg,
                     float* dss, float* dbias, float* scratch, int B, int C, int RT, int P, hipStream_t s) {
  if (B == 0 || C == 0) return 0;
  DQ_REQUIRE(u && dy && g && du && dg && scratch, "wnorm_bwd: missing operand");
  DQ_REQUIRE(!ss || dss, "wnorm_bwd: scale/shift needs a gradient buffer");
  WideNormBwd a{u, dy, g, ss, ss_stride, act, scratch, du, ss ? dss : nullptr, scratch + (int64_t)B * 2 * P,
                dbias ? scratch + (int64_t)B * 2 * P + (int64_t)B * C : nullptr, B, C, RT, P};
  hipLaunchKernelGGL(k_wnorm_bwd_stats, dim3(cdiv(P, 32), B), dim3(256), 0, s, a);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_wnorm_bwd_apply, dim3(cdiv(C, 4), B), dim3(256), 0, s, a);
  DQ_LAUNCH_CHECK();
  DQ_TRY_RC(launch_sum_b(a.dgp, B, C, dg, s));
  if (dbias) DQ_TRY_RC(launch_sum_b(a.dbp, B, C, dbias, s));
  return 0;
}

// sum over t < RT of every row of (rows, P): the bias gradient of a conv whose dU is at hand (rows = B * C, reduced over b by
// launch_sum_b).  One wave per row.
__global__ void __launch_bounds__(256) k_rowsum(const float* __restrict__ x, int64_t rows, int RT, int P, float* __restrict__ out) {
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int t = threadIdx.x & 63; t < RT; t += 64) s += x[r * P + t];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) out[r] = s;
}
int launch_rowsum(const float* x, int64_t rows, int RT, int P, float* out, hipStream_t s) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_rowsum, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, RT, P, out);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
