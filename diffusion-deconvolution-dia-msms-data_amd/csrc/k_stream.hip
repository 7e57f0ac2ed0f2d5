// Streaming (HBM-bound) kernels of the DDIM process: q_sample (K0), DDIM update (K9), MSE loss + its
// gradient (K10), global-norm clip + AdamW on the flat parameter buffer (K11).
// Reference arithmetic: dquartic/model/model.py:239-242 (q_sample), :265-289 (p_sample update),
// :319-322 (sample epilogue), :361 (mse_loss); dquartic/model/model_interface.py:1121-1122
// (clip_grad_norm_(10.0) + AdamW with torch defaults).
// All kernels: 16 B per lane, grid-stride, no LDS.  Algorithmic bytes per element are stated per kernel.
#include "dq_common.h"
#include "dq_kernels.h"
#include <algorithm>
#include <cmath>

namespace dq {

// ---- K0: x_t = sqrt(ab[t_b]) * (2*x0-1) + sqrt(1-ab[t_b]) * noise : 12 B / element (2 reads + 1 write)
__global__ void __launch_bounds__(256) k_q_sample(const float* __restrict__ alpha_bars, const float* __restrict__ x0,
                                                  const int64_t* __restrict__ t, const float* __restrict__ noise,
                                                  float* __restrict__ x_t, int B, int64_t per4, int normalize) {
  const int64_t total = (int64_t)B * per4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per4);
    const float ab = alpha_bars[t[b]];
    const float sa = sqrtf(ab), sb = sqrtf(1.0f - ab);
    float4 x = reinterpret_cast<const float4*>(x0)[i];
    const float4 nz = reinterpret_cast<const float4*>(noise)[i];
    if (normalize) { x.x = x.x * 2.f - 1.f; x.y = x.y * 2.f - 1.f; x.z = x.z * 2.f - 1.f; x.w = x.w * 2.f - 1.f; }
    float4 o;
    o.x = sa * x.x + sb * nz.x; o.y = sa * x.y + sb * nz.y; o.z = sa * x.z + sb * nz.z; o.w = sa * x.w + sb * nz.w;
    reinterpret_cast<float4*>(x_t)[i] = o;
  }
}

int launch_q_sample(const float* alpha_bars, const float* x0, const int64_t* t, const float* noise, float* x_t, int B,
                    int64_t per_sample, int normalize, hipStream_t s) {
  DQ_REQUIRE(per_sample % 4 == 0, "q_sample: RT*MZ must be a multiple of 4");
  const int64_t per4 = per_sample / 4, total = B * per4;
  if (total == 0) return 0;
  const int grid = (int)std::min<int64_t>(cdiv(total, 256), 2048);
  hipLaunchKernelGGL(k_q_sample, dim3(grid), dim3(256), 0, s, alpha_bars, x0, t, noise, x_t, B, per4, normalize);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- K9: x0 = (x_t - sb*eps)/sa ; x_prev = sap*x0 + sbp*eps (t>0) else x0 : 12 B / element
// ``coef`` is a device table of 4 floats per step [sa, sb, sap, sbp]; sap < 0 marks the t == 0 step.
// PRED_X0 (model.py:274-278): ``eps`` holds the network's x0 prediction, eps = (x_t - sa*x0)/sb is derived (and written to
// eps_out when given: p_sample returns it).
template <bool PRED_X0>
__global__ void __launch_bounds__(256) k_ddim_step(const float* __restrict__ x_t, const float* __restrict__ eps,
                                                   float* __restrict__ x_prev, float* __restrict__ eps_out,
                                                   const float* __restrict__ coef, int64_t n4, const int* __restrict__ step_ptr) {
  if (step_ptr) coef += 4 * step_ptr[0];  // graph replay: this step's row of the coefficient table
  const float sa = coef[0], sb = coef[1], sap = coef[2], sbp = coef[3];
  const bool last = sap < 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(x_t)[i];
    const float4 e = reinterpret_cast<const float4*>(eps)[i];
    const float xv[4] = {x.x, x.y, x.z, x.w}, ev[4] = {e.x, e.y, e.z, e.w};
    float ov[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0, ep;
      if (PRED_X0) { x0 = ev[j]; ep = (xv[j] - sa * x0) / sb; }
      else         { ep = ev[j]; x0 = (xv[j] - sb * ep) / sa; }
      ov[j] = last ? x0 : sap * x0 + sbp * ep;
      dv[j] = ep;
    }
    reinterpret_cast<float4*>(x_prev)[i] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (PRED_X0 && eps_out) reinterpret_cast<float4*>(eps_out)[i] = make_float4(dv[0], dv[1], dv[2], dv[3]);
  }
}

__global__ void k_inc_step(int* p) { p[0] += 1; }

int launch_inc_step(int* p, hipStream_t s) {
  hipLaunchKernelGGL(k_inc_step, dim3(1), dim3(1), 0, s, p);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_ddim_step(const float* x_t, const float* eps, float* x_prev, const float* coef_dev, int64_t n, const int* step_ptr,
                     hipStream_t s, int pred_x0, float* eps_out) {
  DQ_REQUIRE(n % 4 == 0, "ddim_step: element count must be a multiple of 4");
  if (n == 0) return 0;
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), 2048);
  if (pred_x0) hipLaunchKernelGGL(k_ddim_step<true>, dim3(grid), dim3(256), 0, s, x_t, eps, x_prev, eps_out, coef_dev, n / 4, step_ptr);
  else hipLaunchKernelGGL(k_ddim_step<false>, dim3(grid), dim3(256), 0, s, x_t, eps, x_prev, eps_out, coef_dev, n / 4, step_ptr);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- sample() epilogue (model.py:319-322): x = (x+1)/2 ; pred = ((2c-1)+1)/2 - x : 16 B / element
__global__ void __launch_bounds__(256) k_sample_finish(const float* __restrict__ x, const float* __restrict__ ms2_cond,
                                                       float* __restrict__ out_x, float* __restrict__ out_noise, int64_t n4,
                                                       int normalize) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 c = reinterpret_cast<const float4*>(ms2_cond)[i];
    float4 o, p;
    if (!normalize) {  // identity normalisation: x stays, pred_noise = ms2_cond - x
      p.x = c.x - v.x; p.y = c.y - v.y; p.z = c.z - v.z; p.w = c.w - v.w;
      reinterpret_cast<float4*>(out_x)[i] = v;
      reinterpret_cast<float4*>(out_noise)[i] = p;
      continue;
    }
    o.x = (v.x + 1.f) * 0.5f; o.y = (v.y + 1.f) * 0.5f; o.z = (v.z + 1.f) * 0.5f; o.w = (v.w + 1.f) * 0.5f;
    p.x = ((c.x * 2.f - 1.f) + 1.f) * 0.5f - o.x; p.y = ((c.y * 2.f - 1.f) + 1.f) * 0.5f - o.y;
    p.z = ((c.z * 2.f - 1.f) + 1.f) * 0.5f - o.z; p.w = ((c.w * 2.f - 1.f) + 1.f) * 0.5f - o.w;
    reinterpret_cast<float4*>(out_x)[i] = o;
    reinterpret_cast<float4*>(out_noise)[i] = p;
  }
}

int launch_sample_finish(const float* x, const float* ms2_cond, float* out_x, float* out_noise, int64_t n, int normalize,
                         hipStream_t s) {
  DQ_REQUIRE(n % 4 == 0, "sample_finish: element count must be a multiple of 4");
  if (n == 0) return 0;
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), 2048);
  hipLaunchKernelGGL(k_sample_finish, dim3(grid), dim3(256), 0, s, x, ms2_cond, out_x, out_noise, n / 4, normalize);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- K10: loss = mean((eps-noise)^2) ; grad = 2*(eps-noise)*gscale : 12 B / element
// partial sums go to ``partials`` (one per block, fixed order => deterministic); k_loss_final sums them.
// Weighted form (pred_type "x0", model.py:372-376, 404): target' = target*tm + ta (the normalised x0), every sample's
// squared error and gradient are multiplied by lw[t[b]] (the SNR table); lw == nullptr => weight 1.
__global__ void __launch_bounds__(256) k_mse_fwd_bwd(const float* __restrict__ eps, const float* __restrict__ noise,
                                                     float* __restrict__ grad, float* __restrict__ partials, int64_t n4,
                                                     float gscale, const float* __restrict__ lw, const int64_t* __restrict__ t,
                                                     int64_t per4, float tm, float ta) {
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 e = reinterpret_cast<const float4*>(eps)[i];
    float4 z = reinterpret_cast<const float4*>(noise)[i];
    float w = 1.f;
    if (lw) {
      w = lw[t[i / per4]];
      z.x = z.x * tm + ta; z.y = z.y * tm + ta; z.z = z.z * tm + ta; z.w = z.w * tm + ta;
    }
    float4 d;
    d.x = e.x - z.x; d.y = e.y - z.y; d.z = e.z - z.z; d.w = e.w - z.w;
    acc += w * (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
    if (grad) {
      const float gs = gscale * w;
      float4 g;
      g.x = d.x * gs; g.y = d.y * gs; g.z = d.z * gs; g.w = d.w * gs;
      reinterpret_cast<float4*>(grad)[i] = g;
    }
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(64) k_sum_partials(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partials[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc * scale;
}

__global__ void __launch_bounds__(256) k_axpy(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

__global__ void __launch_bounds__(256) k_copy(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_zero(float4* __restrict__ dst, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = float4{0.f, 0.f, 0.f, 0.f};
}
// dst[0..n) = 0 as a kernel (n a multiple of 4, dst 16-byte aligned): no memset node in the middle of the step
int launch_zero(float* dst, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  DQ_REQUIRE(n % 4 == 0 && ((uintptr_t)dst & 15) == 0, "zero: needs a 16-byte aligned buffer of a multiple of 4 floats");
  hipLaunchKernelGGL(k_zero, dim3((int)std::min<int64_t>(cdiv(n / 4, 256), 8192)), dim3(256), 0, s, reinterpret_cast<float4*>(dst), n / 4);
  DQ_LAUNCH_CHECK();
  return 0;
}

// plain kernel copy: a hipMemcpyAsync in the middle of the step costs far more than its 3 us blit (queue barriers around it)
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_copy, dim3((int)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0, s, dst, src, n);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_axpy(float* dst, const float* src, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpy, dim3((int)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0, s, dst, src, n);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_sum_partials(const float* partials, int count, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, s, partials, count, scale, out);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_mse_fwd_bwd(const float* eps, const float* noise, float* loss_out, float* grad_out, float* partials, int64_t n,
                       hipStream_t s, const float* lw, const int64_t* t, int64_t per_sample, float tm, float ta, int* defer_sum) {
  DQ_REQUIRE(n % 4 == 0 && n > 0, "mse: element count must be a positive multiple of 4");
  DQ_REQUIRE(!lw || (t && per_sample > 0 && per_sample % 4 == 0 && n % per_sample == 0),
             "mse: the weighted form needs t and a per-sample element count that is a multiple of 4");
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), MSE_MAX_BLOCKS);
  hipLaunchKernelGGL(k_mse_fwd_bwd, dim3(grid), dim3(256), 0, s, eps, noise, grad_out, partials, n / 4, 2.0f / (float)n, lw, t,
                     lw ? per_sample / 4 : (int64_t)1, tm, ta);
  DQ_LAUNCH_CHECK();
  if (defer_sum) { *defer_sum = grid; return 0; }  // the caller sums the partials later (launch_sum_partials(partials, grid, 1 / n, loss_out))
  return launch_sum_partials(partials, grid, 1.0f / (float)n, loss_out, s);
}

// ---- the MS1 term of train_step (reference model.py:364-371, 379-386, 398-402; semantics chosen in DESIGN.md section 12 because
// the reference's branch raises): per sample b, with D = x_t - eps_pred ('eps') or x0_pred ('x0'),
//   additional_b = sum over f in {sum, mean, max over m/z} of mean_rt (s_f[rt] / max_rt s_f - ms1n[rt] / max_rt ms1n)^2
//   loss_b = (1 - w) * MSE_b + w * additional_b ; loss = mean_b loss_weight[t_b] * loss_b
// (1) k_ms1_rows: one wave per (b, rt) row -> sum, max and arg-max over m/z;
// (2) k_ms1_sample: one block per sample -> the three normalisers, additional_b and d additional_b / d s_f[rt];
// (3) k_ms1_apply: grad = (1 - w) * grad_MSE + sign * w * loss_weight_b / B * (d/d s_sum + d/d s_mean / MZ + [mz == argmax] d/d s_max)
//     and the loss scalar.  Fixed-order reductions throughout.
__global__ void __launch_bounds__(256) k_ms1_rows(const float* __restrict__ out, const float* __restrict__ x_t, int64_t rows, int MZ,
                                                  float* __restrict__ rsum, float* __restrict__ rmax, int* __restrict__ ramax) {
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float sm = 0.f, mx = -INFINITY;
  int am = 0;
  for (int mz = lane; mz < MZ; mz += 64) {
    const float o = out[row * MZ + mz];
    const float d = x_t ? x_t[row * MZ + mz] - o : o;
    sm += d;
    if (d > mx) { mx = d; am = mz; }
  }
  sm = wave_sum(sm);
  // arg-max over the lanes: larger value wins, the smaller index on ties (torch.max returns the first maximum)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omx = __shfl_xor(mx, o, 64);
    const int oam = __shfl_xor(am, o, 64);
    if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
  }
  if (lane == 0) { rsum[row] = sm; rmax[row] = mx; ramax[row] = am; }
}

__device__ __forceinline__ void block_argmax(float& v, int& idx, float* sv, int* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = idx; }
  __syncthreads();
  v = sv[0]; idx = si[0];
  for (int k = 1; k < 4; ++k)
    if (sv[k] > v || (sv[k] == v && si[k] < idx)) { v = sv[k]; idx = si[k]; }
}
__device__ __forceinline__ float block_sum4(float v, float* sv) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sv[0] + sv[1]) + (sv[2] + sv[3]);
}

__global__ void __launch_bounds__(256) k_ms1_sample(const float* __restrict__ rsum, const float* __restrict__ rmax,
                                                    const float* __restrict__ ms1, float cm, float ca, int RT, int MZ,
                                                    const float* __restrict__ lw, const int64_t* __restrict__ t,
                                                    float* __restrict__ dsum, float* __restrict__ dmax, float* __restrict__ addl) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x;
  const float* s_sum = rsum + (int64_t)b * RT;
  const float* s_max = rmax + (int64_t)b * RT;
  const float* m1 = ms1 + (int64_t)b * RT;
  const float invMZ = 1.0f / (float)MZ;
  // the four row maxima (and where they sit): sum, mean (= sum / MZ: same place), max, and the MS1 chromatogram
  float a0 = -INFINITY, a2 = -INFINITY, a3 = -INFINITY;
  int i0 = 0, i2 = 0, i3 = 0;
  for (int rt = threadIdx.x; rt < RT; rt += 256) {
    const float v0 = s_sum[rt], v2 = s_max[rt], v3 = fmaf(m1[rt], cm, ca);
    if (v0 > a0) { a0 = v0; i0 = rt; }
    if (v2 > a2) { a2 = v2; i2 = rt; }
    if (v3 > a3) { a3 = v3; i3 = rt; }
  }
  block_argmax(a0, i0, sv, si);
  block_argmax(a2, i2, sv, si);
  block_argmax(a3, i3, sv, si);
  const float m_sum = a0, m_mean = a0 * invMZ, m_max = a2, m_ms1 = a3;
  // A_f = mean_rt (u_f - v)^2 ; c_f = sum_rt g_f s_f / m_f^2 with g_f = 2 (u_f - v) / RT
  float A0 = 0.f, A1 = 0.f, A2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  const float gs = 2.0f / (float)RT;
  for (int rt = threadIdx.x; rt < RT; rt += 256) {
    const float v = fmaf(m1[rt], cm, ca) / m_ms1;
    const float s0 = s_sum[rt], s1 = s0 * invMZ, s2 = s_max[rt];
    const float e0 = s0 / m_sum - v, e1 = s1 / m_mean - v, e2 = s2 / m_max - v;
    A0 = fmaf(e0, e0, A0); A1 = fmaf(e1, e1, A1); A2 = fmaf(e2, e2, A2);
    c0 = fmaf(gs * e0, s0, c0); c1 = fmaf(gs * e1, s1, c1); c2 = fmaf(gs * e2, s2, c2);
  }
  A0 = block_sum4(A0, sv); A1 = block_sum4(A1, sv); A2 = block_sum4(A2, sv);
  c0 = block_sum4(c0, sv) / (m_sum * m_sum); c1 = block_sum4(c1, sv) / (m_mean * m_mean); c2 = block_sum4(c2, sv) / (m_max * m_max);
  for (int rt = threadIdx.x; rt < RT; rt += 256) {
    const float v = fmaf(m1[rt], cm, ca) / m_ms1;
    const float s0 = s_sum[rt], s1 = s0 * invMZ, s2 = s_max[rt];
    const float d0 = gs * (s0 / m_sum - v) / m_sum - (rt == i0 ? c0 : 0.f);
    const float d1 = gs * (s1 / m_mean - v) / m_mean - (rt == i0 ? c1 : 0.f);
    const float d2 = gs * (s2 / m_max - v) / m_max - (rt == i2 ? c2 : 0.f);
    dsum[(int64_t)b * RT + rt] = fmaf(d1, invMZ, d0);
    dmax[(int64_t)b * RT + rt] = d2;
  }
  if (threadIdx.x == 0) addl[b] = (lw ? lw[t[b]] : 1.0f) * (((A0 + A1) + A2) / (float)RT);
}

__global__ void __launch_bounds__(256) k_ms1_apply(float* __restrict__ grad, const float* __restrict__ dsum, const float* __restrict__ dmax,
                                                   const int* __restrict__ ramax, const float* __restrict__ lw,
                                                   const int64_t* __restrict__ t, const float* __restrict__ addl, float w, float sign,
                                                   int B, int RT, int MZ, float* __restrict__ loss) {
  const int64_t n = (int64_t)B * RT * MZ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / MZ;
    const int mz = (int)(i - row * MZ);
    const int b = (int)(row / RT);
    const float k = sign * w * (lw ? lw[t[b]] : 1.0f) / (float)B;
    const float extra = dsum[row] + (mz == ramax[row] ? dmax[row] : 0.f);
    if (grad) grad[i] = fmaf(1.0f - w, grad[i], k * extra);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += addl[b];
    loss[0] = (1.0f - w) * loss[0] + w * acc / (float)B;
  }
}

// scratch: 5 * B * RT + B floats.  loss_out / grad hold the MSE part on entry (launch_mse_fwd_bwd) and the combined loss on exit.
int launch_ms1_loss(const float* out, const float* x_t, const float* ms1, float cm, float ca, const float* lw, const int64_t* t, float w,
                    int B, int RT, int MZ, float* grad, float* loss_out, float* scratch, hipStream_t s) {
  DQ_REQUIRE(out && ms1 && loss_out && scratch && B > 0 && RT > 0 && MZ > 0, "ms1 loss: missing operand");
  DQ_REQUIRE(!lw || t, "ms1 loss: per-timestep weights need t");
  const int64_t rows = (int64_t)B * RT;
  float* rsum = scratch; float* rmax = rsum + rows; int* ramax = reinterpret_cast<int*>(rmax + rows);
  float* dsum = rmax + 2 * rows; float* dmax = dsum + rows; float* addl = dmax + rows;
  hipLaunchKernelGGL(k_ms1_rows, dim3(cdiv(rows, 4)), dim3(256), 0, s, out, x_t, rows, MZ, rsum, rmax, ramax);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ms1_sample, dim3(B), dim3(256), 0, s, rsum, rmax, ms1, cm, ca, RT, MZ, lw, t, dsum, dmax, addl);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ms1_apply, dim3((int)std::min<int64_t>(cdiv(rows * MZ, 256), 4096)), dim3(256), 0, s, grad, dsum, dmax, ramax, lw, t,
                     addl, w, x_t ? -1.0f : 1.0f, B, RT, MZ, loss_out);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- K11: global L2 norm (two-stage, deterministic) + clip + AdamW, flat buffers.
// norm pass: 4 B / param ; update pass: 28 B / param (read p,g,m,v ; write p,m,v).
__global__ void __launch_bounds__(256) k_sumsq(const float* __restrict__ g, int64_t n, float gscale, float* __restrict__ partials) {
  // 16-byte loads, four independent accumulators, two loads in flight: the 764 MB gradient of the transformer (191 M parameters)
  // is a bandwidth job; the scalar one-accumulator loop ran it at 2.3 TB/s
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t n4 = (((uintptr_t)g & 15) == 0) ? n / 4 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 u = g4[i], w = g4[i + stride];
    a0 = fmaf(u.x * gscale, u.x * gscale, a0); a1 = fmaf(u.y * gscale, u.y * gscale, a1);
    a2 = fmaf(u.z * gscale, u.z * gscale, a2); a3 = fmaf(u.w * gscale, u.w * gscale, a3);
    a0 = fmaf(w.x * gscale, w.x * gscale, a0); a1 = fmaf(w.y * gscale, w.y * gscale, a1);
    a2 = fmaf(w.z * gscale, w.z * gscale, a2); a3 = fmaf(w.w * gscale, w.w * gscale, a3);
  }
  for (; i < n4; i += stride) {
    const float4 u = g4[i];
    a0 = fmaf(u.x * gscale, u.x * gscale, a0); a1 = fmaf(u.y * gscale, u.y * gscale, a1);
    a2 = fmaf(u.z * gscale, u.z * gscale, a2); a3 = fmaf(u.w * gscale, u.w * gscale, a3);
  }
  for (int64_t j = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += stride) {
    const float v = g[j] * gscale;
    a0 = fmaf(v, v, a0);
  }
  float acc = (a0 + a1) + (a2 + a3);
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) k_adamw_clip(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ partials,
                                                    int n_partials, float gscale, float max_norm, float decay, float step_size,
                                                    float one_m_b1, float b2, float one_m_b2, float eps, float bc2_sqrt,
                                                    float* __restrict__ gnorm_out) {
  // every block re-derives the same clip coefficient from the same partials in the same order
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 64) acc += partials[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
      const float nrm = sqrtf(acc);
      const float c = max_norm / (nrm + 1e-6f);  // torch clip_grad_norm_: coef clamped to 1
      s_coef = (max_norm > 0.f) ? fminf(c, 1.0f) : 1.0f;
      if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = nrm;
    }
  }
  __syncthreads();
  const float coef = s_coef * gscale;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * decay;
    const float mo = m[i];
    const float mi = mo + one_m_b1 * (gi - mo);          // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(one_m_b2 * gi, gi, b2 * v[i]);  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---- the same step with the step count and the learning rate in DEVICE memory: nothing about the call changes from one step to the next, so
// a captured graph of it can be replayed (the host-side variant bakes lr and the bias corrections of step t into its launch arguments).
// hyp (8 floats behind the partial sums): decay, step_size, bc2_sqrt -- formed in double from the device step counter exactly as
// launch_adamw_clip forms them on the host; the counter is incremented first (step t = 1 for the first call).
__global__ void k_adamw_hyper(int* __restrict__ step, const float* __restrict__ lr_dev, double b1, double b2, double wd, float* __restrict__ hyp) {
  const int t = step[0] + 1;
  step[0] = t;
  const double lr = (double)lr_dev[0] + (double)lr_dev[1];  // (hi, lo) float pair: ~48 bits of the host's double lr
  const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
  hyp[0] = (float)(1.0 - lr * wd);
  hyp[1] = (float)(lr / bc1);
  hyp[2] = (float)sqrt(bc2);
}
__global__ void __launch_bounds__(256) k_adamw_clip_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, const float* __restrict__ partials, int n_partials,
                                                        float gscale, float max_norm, const float* __restrict__ hyp, float one_m_b1, float b2,
                                                        float one_m_b2, float eps, float* __restrict__ gnorm_out) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 64) acc += partials[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
      const float nrm = sqrtf(acc);
      const float c = max_norm / (nrm + 1e-6f);
      s_coef = (max_norm > 0.f) ? fminf(c, 1.0f) : 1.0f;
      if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = nrm;
    }
  }
  __syncthreads();
  const float coef = s_coef * gscale;
  const float decay = hyp[0], step_size = hyp[1], bc2_sqrt = hyp[2];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * decay;
    const float mo = m[i];
    const float mi = mo + one_m_b1 * (gi - mo);
    const float vi = fmaf(one_m_b2 * gi, gi, b2 * v[i]);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
int launch_adamw_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, float* partials, float gscale, float max_norm,
                          const float* lr_dev, double b1, double b2, double eps, double wd, int* step_dev, float* gnorm_out, hipStream_t s) {
  DQ_REQUIRE(n > 0 && lr_dev && step_dev, "adamw: need n > 0, the device learning rate and the device step counter");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), MSE_MAX_BLOCKS - 8);
  float* hyp = partials + (MSE_MAX_BLOCKS - 8);  // (the partial-sum scratch has MSE_MAX_BLOCKS floats)
  hipLaunchKernelGGL(k_adamw_hyper, dim3(1), dim3(1), 0, s, step_dev, lr_dev, b1, b2, wd, hyp);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sumsq, dim3(grid), dim3(256), 0, s, g, n, gscale, partials);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_adamw_clip_dev, dim3(grid), dim3(256), 0, s, p, g, m, v, n, partials, grid, gscale, max_norm, hyp, (float)(1.0 - b1), (float)b2,
                     (float)(1.0 - b2), (float)eps, gnorm_out);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, float* partials, float gscale, float max_norm,
                      double lr, double b1, double b2, double eps, double wd, int step, float* gnorm_out, hipStream_t s) {
  DQ_REQUIRE(n > 0 && step >= 1, "adamw: need n > 0 and step >= 1");
  // (the same grid as the device-state variant above, which keeps 8 floats of the scratch for its scalars: the norm's partial sums are then
  // taken in the same order by both, for any n)
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), MSE_MAX_BLOCKS - 8);
  hipLaunchKernelGGL(k_sumsq, dim3(grid), dim3(256), 0, s, g, n, gscale, partials);
  DQ_LAUNCH_CHECK();
  const double bc1 = 1.0 - std::pow(b1, step), bc2 = 1.0 - std::pow(b2, step);
  // scalar factors are formed in double like torch's Python-side arithmetic, then applied in fp32
  hipLaunchKernelGGL(k_adamw_clip, dim3(grid), dim3(256), 0, s, p, g, m, v, n, partials, grid, gscale, max_norm,
                     (float)(1.0 - lr * wd), (float)(lr / bc1), (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps,
                     (float)std::sqrt(bc2), gnorm_out);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
