// Pointwise and row-wise kernels of the CustomTransformer path (reference: dquartic/model/building_blocks.py): RoPE over the
// hidden axis (:6-66), the Linear(1, hidden) conditional projection (:213, 241-242), sinusoidal time features (:104-108), exact
// GELU, post-norm LayerNorm with its residual (:168, 172), row softmax of the attention scores and the column sums that make the
// bias / gain gradients.  All HBM-bound and small next to the GEMMs (k_gemm.hip); parameter-gradient sums go through per-block
// partials and an ordered second pass (deterministic, no atomics).
#include "dq_common.h"
#include "dq_tfm.h"
#include <algorithm>

namespace dq {

namespace {
__global__ void __launch_bounds__(256) k_rope_add(float* __restrict__ x, const float* __restrict__ sin_t, const float* __restrict__ cos_t,
                                                  const float* __restrict__ temb, int B, int S, int H, int inverse) {
  const int half = H >> 1;
  const int64_t total = (int64_t)B * S * half;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / half;
    const int j = (int)(e - row * half), sidx = (int)(row % S), b = (int)(row / S);
    float2* p = reinterpret_cast<float2*>(x + row * H) + j;
    const float2 v = *p;
    const float sn = sin_t[sidx * half + j], cs = cos_t[sidx * half + j];
    float2 o;
    if (!inverse) {
      o.x = v.x * cs - v.y * sn;  // building_blocks.py:59-60
      o.y = v.x * sn + v.y * cs;
      if (temb) { o.x += temb[(int64_t)b * H + 2 * j]; o.y += temb[(int64_t)b * H + 2 * j + 1]; }
    } else {  // transpose of the rotation: gradient w.r.t. the un-rotated pair
      o.x = v.x * cs + v.y * sn;
      o.y = v.y * cs - v.x * sn;
    }
    *p = o;
  }
}

__global__ void __launch_bounds__(256) k_cond_embed(const float* __restrict__ xc, const float* __restrict__ w, const float* __restrict__ bias,
                                                    const float* __restrict__ sin_t, const float* __restrict__ cos_t, float* __restrict__ c, int B,
                                                    int S, int H) {
  const int half = H >> 1;
  const int64_t total = (int64_t)B * S * half;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / half;
    const int j = (int)(e - row * half), sidx = (int)(row % S);
    const float xv = xc[row];
    const float v1 = xv * w[2 * j] + bias[2 * j], v2 = xv * w[2 * j + 1] + bias[2 * j + 1];
    const float sn = sin_t[sidx * half + j], cs = cos_t[sidx * half + j];
    float2 o;
    o.x = v1 * cs - v2 * sn;
    o.y = v1 * sn + v2 * cs;
    reinterpret_cast<float2*>(c + row * H)[j] = o;
  }
}

// column part of the conditional-projection backward: block y sums its rows for the pairs of block x
__global__ void __launch_bounds__(256) k_cond_bwd_cols(const float* __restrict__ dc, const float* __restrict__ xc, const float* __restrict__ sin_t,
                                                       const float* __restrict__ cos_t, float* __restrict__ part, int rows, int S, int H) {
  const int half = H >> 1;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half) return;
  float w1 = 0.f, w2 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const float2 g = reinterpret_cast<const float2*>(dc + (int64_t)row * H)[j];
    const int sidx = row % S;
    const float sn = sin_t[sidx * half + j], cs = cos_t[sidx * half + j];
    const float d1 = g.x * cs + g.y * sn, d2 = g.y * cs - g.x * sn;
    const float xv = xc[row];
    w1 = fmaf(d1, xv, w1); w2 = fmaf(d2, xv, w2);
    b1 += d1; b2 += d2;
  }
  float* p = part + (int64_t)blockIdx.y * 2 * H;
  p[2 * j] = w1; p[2 * j + 1] = w2;
  p[H + 2 * j] = b1; p[H + 2 * j + 1] = b2;
}
// out[i] += sum over nb partial vectors (stride `stride`), fixed order
__global__ void __launch_bounds__(256) k_partial_reduce(const float* __restrict__ part, int nb, int64_t stride, int n, float* __restrict__ out,
                                                        int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= nb; b += 8) {  // eight loads in flight, summed in index order
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)(b + k) * stride + i];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; b < nb; ++b) s += part[(int64_t)b * stride + i];
  out[i] = accumulate ? out[i] + s : s;
}
// row part: dx_cond[row] = sum_h dv[row][h] * w[h]   (one wave per row)
__global__ void __launch_bounds__(256) k_cond_bwd_rows(const float* __restrict__ dc, const float* __restrict__ w, const float* __restrict__ sin_t,
                                                       const float* __restrict__ cos_t, float* __restrict__ dxc, int rows, int S, int H) {
  const int half = H >> 1;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;  // whole waves leave together
  const int sidx = row % S;
  float acc = 0.f;
  for (int j = lane; j < half; j += 64) {
    const float2 g = reinterpret_cast<const float2*>(dc + (int64_t)row * H)[j];
    const float sn = sin_t[sidx * half + j], cs = cos_t[sidx * half + j];
    acc = fmaf(g.x * cs + g.y * sn, w[2 * j], acc);
    acc = fmaf(g.y * cs - g.x * sn, w[2 * j + 1], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) dxc[row] = acc;
}

__global__ void __launch_bounds__(256) k_time_features(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ e, int B,
                                                       int H) {
  const int half = H >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float a = (float)t[b] * freqs[j];  // building_blocks.py:107
  e[(int64_t)b * H + j] = sinf(a);
  e[(int64_t)b * H + half + j] = cosf(a);
}

__global__ void __launch_bounds__(256) k_gelu(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] = gelu_f(x[e]);
}
__global__ void __launch_bounds__(256) k_gelu_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) dx[e] = dy[e] * gelu_grad_f(x[e]);
}

// one wave per row: y = x + r ; out = (y - mean) * rstd * g + b ; biased variance, eps inside the root (nn.LayerNorm)
__global__ void __launch_bounds__(256) k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ g,
                                                       const float* __restrict__ b, float* __restrict__ y, float* __restrict__ out,
                                                       float* __restrict__ stats, int rows, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * H;
  const float* rr = r ? r + (int64_t)row * H : nullptr;
  float* yr = y + (int64_t)row * H;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) {
    const float v = rr ? xr[h] + rr[h] : xr[h];
    yr[h] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
  for (int h = lane; h < H; h += 64) {
    const float d = yr[h] - mean;  // this lane's own stores
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
  for (int h = lane; h < H; h += 64) out[(int64_t)row * H + h] = (yr[h] - mean) * rstd * g[h] + b[h];
  if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// H <= 64 * VM: the row lives in registers (every load is issued up front; with a few dozen rows per launch the kernel is one
// dependent chain, and passes through memory would triple it)
template <int VM>
__global__ void __launch_bounds__(256) k_layernorm_fwd_reg(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ g,
                                                           const float* __restrict__ b, float* __restrict__ y, float* __restrict__ out,
                                                           float* __restrict__ stats, int rows, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int64_t base = (int64_t)row * H;
  float v[VM], gg[VM], bb[VM];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VM; ++i) {
    const int h = lane + 64 * i;
    const bool ok = h < H;
    v[i] = ok ? (r ? x[base + h] + r[base + h] : x[base + h]) : 0.f;
    gg[i] = ok ? g[h] : 0.f;
    bb[i] = ok ? b[h] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VM; ++i) {
    const float d = (lane + 64 * i < H) ? v[i] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
#pragma unroll
  for (int i = 0; i < VM; ++i) {
    const int h = lane + 64 * i;
    if (h < H) {
      y[base + h] = v[i];
      out[base + h] = (v[i] - mean) * rstd * gg[i] + bb[i];
    }
  }
  if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
template <int VM>
__global__ void __launch_bounds__(256) k_layernorm_bwd_rows_reg(const float* __restrict__ y, const float* __restrict__ stats,
                                                                const float* __restrict__ g, const float* __restrict__ dout, float* __restrict__ dy,
                                                                int rows, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const int64_t base = (int64_t)row * H;
  float dxh[VM], xh[VM];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VM; ++i) {
    const int h = lane + 64 * i;
    const bool ok = h < H;
    dxh[i] = ok ? dout[base + h] * g[h] : 0.f;
    xh[i] = ok ? (y[base + h] - mean) * rstd : 0.f;
    s1 += dxh[i];
    s2 = fmaf(dxh[i], xh[i], s2);
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int i = 0; i < VM; ++i) {
    const int h = lane + 64 * i;
    if (h < H) dy[base + h] = rstd * (dxh[i] - s1 - xh[i] * s2);
  }
}

// H <= 1024, H % 4 == 0: one block per row, one float4 per thread -- a single round of loads and two LDS reductions (with a few
// dozen rows per launch the kernel is one dependent chain; its length, not its bandwidth, is what the batch-1 step pays 16 times)
__device__ __forceinline__ float block4_sum(float v, float* red, int slot) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[slot * 4 + (threadIdx.x >> 6)] = v;
  __syncthreads();
  return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
}
__global__ void __launch_bounds__(256) k_layernorm_fwd_blk(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ g,
                                                           const float* __restrict__ b, float* __restrict__ y, float* __restrict__ out,
                                                           float* __restrict__ stats, int H) {
  __shared__ float red[8];
  const int row = blockIdx.x, h = threadIdx.x * 4;
  const bool ok = h < H;
  const int64_t at = (int64_t)row * H + h;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), gg = v, bb = v;
  if (ok) {
    v = *reinterpret_cast<const float4*>(x + at);
    gg = *reinterpret_cast<const float4*>(g + h);
    bb = *reinterpret_cast<const float4*>(b + h);
    if (r) {
      const float4 rv = *reinterpret_cast<const float4*>(r + at);
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    *reinterpret_cast<float4*>(y + at) = v;
  }
  const float mean = block4_sum((v.x + v.y) + (v.z + v.w), red, 0) / (float)H;
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  const float rstd = 1.0f / sqrtf(block4_sum(fmaf(d.x, d.x, d.y * d.y) + fmaf(d.z, d.z, d.w * d.w), red, 1) / (float)H + 1e-5f);
  if (ok) *reinterpret_cast<float4*>(out + at) = make_float4(d.x * rstd * gg.x + bb.x, d.y * rstd * gg.y + bb.y, d.z * rstd * gg.z + bb.z,
                                                             d.w * rstd * gg.w + bb.w);
  if (threadIdx.x == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
__global__ void __launch_bounds__(256) k_layernorm_bwd_rows_blk(const float* __restrict__ y, const float* __restrict__ stats,
                                                                const float* __restrict__ g, const float* __restrict__ dout, float* __restrict__ dy,
                                                                int H) {
  __shared__ float red[8];
  const int row = blockIdx.x, h = threadIdx.x * 4;
  const bool ok = h < H;
  const int64_t at = (int64_t)row * H + h;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float4 dxh = make_float4(0.f, 0.f, 0.f, 0.f), xh = dxh;
  if (ok) {
    const float4 dv = *reinterpret_cast<const float4*>(dout + at), gg = *reinterpret_cast<const float4*>(g + h);
    const float4 yv = *reinterpret_cast<const float4*>(y + at);
    dxh = make_float4(dv.x * gg.x, dv.y * gg.y, dv.z * gg.z, dv.w * gg.w);
    xh = make_float4((yv.x - mean) * rstd, (yv.y - mean) * rstd, (yv.z - mean) * rstd, (yv.w - mean) * rstd);
  }
  float s1 = (dxh.x + dxh.y) + (dxh.z + dxh.w);
  float s2 = fmaf(dxh.x, xh.x, dxh.y * xh.y) + fmaf(dxh.z, xh.z, dxh.w * xh.w);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  s1 = ((red[0] + red[1]) + (red[2] + red[3])) / (float)H;
  s2 = ((red[4] + red[5]) + (red[6] + red[7])) / (float)H;
  if (ok) *reinterpret_cast<float4*>(dy + at) = make_float4(rstd * (dxh.x - s1 - xh.x * s2), rstd * (dxh.y - s1 - xh.y * s2),
                                                            rstd * (dxh.z - s1 - xh.z * s2), rstd * (dxh.w - s1 - xh.w * s2));
}

// dy = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = dout * g, xh = (y - mean) * rstd
__global__ void __launch_bounds__(256) k_layernorm_bwd_rows(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ g,
                                                            const float* __restrict__ dout, float* __restrict__ dy, int rows, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const float* yr = y + (int64_t)row * H;
  const float* dr = dout + (int64_t)row * H;
  float s1 = 0.f, s2 = 0.f;
  for (int h = lane; h < H; h += 64) {
    const float dxh = dr[h] * g[h], xh = (yr[h] - mean) * rstd;
    s1 += dxh;
    s2 = fmaf(dxh, xh, s2);
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
  for (int h = lane; h < H; h += 64) {
    const float dxh = dr[h] * g[h], xh = (yr[h] - mean) * rstd;
    dy[(int64_t)row * H + h] = rstd * (dxh - s1 - xh * s2);
  }
}
__global__ void __launch_bounds__(256) k_layernorm_bwd_cols(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ dout,
                                                            float* __restrict__ part, int rows, int H) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  float dg = 0.f, db = 0.f;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const float d = dout[(int64_t)row * H + h];
    dg = fmaf(d, (y[(int64_t)row * H + h] - stats[2 * row]) * stats[2 * row + 1], dg);
    db += d;
  }
  part[(int64_t)blockIdx.y * 2 * H + h] = dg;
  part[(int64_t)blockIdx.y * 2 * H + H + h] = db;
}

__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ p, int64_t rows, int n, int ld, float scale) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* pr = p + row * ld;
  float m = -INFINITY;
  for (int i = lane; i < n; i += 64) m = fmaxf(m, pr[i] * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float e = expf(pr[i] * scale - m);
    pr[i] = e;
    s += e;
  }
  const float inv = 1.0f / wave_sum(s);
  for (int i = lane; i < n; i += 64) pr[i] *= inv;
}
__global__ void __launch_bounds__(256) k_softmax_rows_bwd(const float* __restrict__ p, float* __restrict__ dp, int64_t rows, int n, int ld, float scale) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = p + row * ld;
  float* dr = dp + row * ld;
  float t = 0.f;
  for (int i = lane; i < n; i += 64) t = fmaf(pr[i], dr[i], t);
  t = wave_sum(t);
  for (int i = lane; i < n; i += 64) dr[i] = pr[i] * (dr[i] - t) * scale;
}

__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ x, int M, int N, int64_t ld, float* __restrict__ part) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = blockIdx.y; m < M; m += gridDim.y) s += x[(int64_t)m * ld + n];
  part[(int64_t)blockIdx.y * N + n] = s;
}
__global__ void __launch_bounds__(256) k_seqsum(const float* __restrict__ x, int B, int S, int N, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N, n = i - b * N;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += x[((int64_t)b * S + q) * N + n];
  out[i] = s;
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>(cdiv(n, 256), 8192); }
}  // namespace

int launch_rope_add(float* x, const float* sin_t, const float* cos_t, const float* temb, int B, int S, int H, int inverse, hipStream_t s) {
  DQ_REQUIRE(H % 2 == 0, "rope: hidden_dim must be even");
  const int64_t total = (int64_t)B * S * (H / 2);
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_rope_add, dim3(grid_for(total)), dim3(256), 0, s, x, sin_t, cos_t, temb, B, S, H, inverse);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_cond_embed(const float* x_cond, const float* w, const float* bias, const float* sin_t, const float* cos_t, float* c, int B, int S,
                      int H, hipStream_t s) {
  const int64_t total = (int64_t)B * S * (H / 2);
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_cond_embed, dim3(grid_for(total)), dim3(256), 0, s, x_cond, w, bias, sin_t, cos_t, c, B, S, H);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_cond_embed_bwd(const float* dc, const float* x_cond, const float* w, const float* sin_t, const float* cos_t, float* dw, float* db,
                          float* dx_cond, float* scratch, int B, int S, int H, hipStream_t s, int accumulate) {
  const int rows = B * S;
  if (rows == 0) return 0;
  const int nb = std::min(rows, 64);
  hipLaunchKernelGGL(k_cond_bwd_cols, dim3(cdiv(H / 2, 256), nb), dim3(256), 0, s, dc, x_cond, sin_t, cos_t, scratch, rows, S, H);
  hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(H, 256)), dim3(256), 0, s, scratch, nb, (int64_t)2 * H, H, dw, accumulate);
  hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(H, 256)), dim3(256), 0, s, scratch + H, nb, (int64_t)2 * H, H, db, accumulate);
  if (dx_cond) hipLaunchKernelGGL(k_cond_bwd_rows, dim3(cdiv(rows, 4)), dim3(256), 0, s, dc, w, sin_t, cos_t, dx_cond, rows, S, H);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_time_features(const int64_t* t, const float* freqs, float* e, int B, int H, hipStream_t s) {
  hipLaunchKernelGGL(k_time_features, dim3(cdiv(B * (H / 2), 256)), dim3(256), 0, s, t, freqs, e, B, H);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_gelu(const float* x, float* y, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gelu, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gelu_bwd, dim3(grid_for(n)), dim3(256), 0, s, x, dy, dx, n);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_layernorm_fwd(const float* x, const float* r, const float* g, const float* b, float* y, float* out, float* stats, int rows, int H,
                         hipStream_t s) {
  if (rows == 0) return 0;
  const bool vec = H % 4 == 0 && ((((uintptr_t)x | (uintptr_t)r | (uintptr_t)g | (uintptr_t)b | (uintptr_t)y | (uintptr_t)out) & 15) == 0);
  if (vec && H > 256 && H <= 1024) hipLaunchKernelGGL(k_layernorm_fwd_blk, dim3(rows), dim3(256), 0, s, x, r, g, b, y, out, stats, H);
  else if (H <= 256) hipLaunchKernelGGL(k_layernorm_fwd_reg<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, r, g, b, y, out, stats, rows, H);
  else if (H <= 1024) hipLaunchKernelGGL(k_layernorm_fwd_reg<16>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, r, g, b, y, out, stats, rows, H);
  else hipLaunchKernelGGL(k_layernorm_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, r, g, b, y, out, stats, rows, H);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_layernorm_bwd(const float* y, const float* stats, const float* g, const float* dout, float* dy, float* dg, float* db, float* scratch,
                         int rows, int H, hipStream_t s, int accumulate) {
  if (rows == 0) return 0;
  const bool vec = H % 4 == 0 && ((((uintptr_t)y | (uintptr_t)g | (uintptr_t)dout | (uintptr_t)dy) & 15) == 0);
  if (vec && H > 256 && H <= 1024) hipLaunchKernelGGL(k_layernorm_bwd_rows_blk, dim3(rows), dim3(256), 0, s, y, stats, g, dout, dy, H);
  else if (H <= 256) hipLaunchKernelGGL(k_layernorm_bwd_rows_reg<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, y, stats, g, dout, dy, rows, H);
  else if (H <= 1024) hipLaunchKernelGGL(k_layernorm_bwd_rows_reg<16>, dim3(cdiv(rows, 4)), dim3(256), 0, s, y, stats, g, dout, dy, rows, H);
  else hipLaunchKernelGGL(k_layernorm_bwd_rows, dim3(cdiv(rows, 4)), dim3(256), 0, s, y, stats, g, dout, dy, rows, H);
  const int nb = std::min(rows, LN_BWD_BLOCKS);
  hipLaunchKernelGGL(k_layernorm_bwd_cols, dim3(cdiv(H, 256), nb), dim3(256), 0, s, y, stats, dout, scratch, rows, H);
  if (db == dg + H) {  // gain and bias adjacent (the flat parameter layout): the [dg | db] partial rows reduce in one launch
    hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(2 * H, 256)), dim3(256), 0, s, scratch, nb, (int64_t)2 * H, 2 * H, dg, accumulate);
  } else {
    hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(H, 256)), dim3(256), 0, s, scratch, nb, (int64_t)2 * H, H, dg, accumulate);
    hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(H, 256)), dim3(256), 0, s, scratch + H, nb, (int64_t)2 * H, H, db, accumulate);
  }
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_softmax_rows(float* p, int64_t rows, int n, int ld, float scale, hipStream_t s) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, p, rows, n, ld, scale);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_softmax_rows_bwd(const float* p, float* dp, int64_t rows, int n, int ld, float scale, hipStream_t s) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_softmax_rows_bwd, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, p, dp, rows, n, ld, scale);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_colsum(const float* x, int M, int N, int64_t ld, float* out, float* scratch, hipStream_t s, int accumulate) {
  if (M == 0 || N == 0) return 0;
  if (M <= COLSUM_BLOCKS) {  // a few dozen rows: they are the partial vectors already
    hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(N, 256)), dim3(256), 0, s, x, M, ld, N, out, accumulate);
    DQ_LAUNCH_CHECK();
    return 0;
  }
  const int nb = COLSUM_BLOCKS;
  hipLaunchKernelGGL(k_colsum, dim3(cdiv(N, 256), nb), dim3(256), 0, s, x, M, N, ld, scratch);
  hipLaunchKernelGGL(k_partial_reduce, dim3(cdiv(N, 256)), dim3(256), 0, s, scratch, nb, (int64_t)N, N, out, accumulate);
  DQ_LAUNCH_CHECK();
  return 0;
}
int launch_seqsum(const float* x, int B, int S, int N, float* out, hipStream_t s) {
  if (B * N == 0) return 0;
  hipLaunchKernelGGL(k_seqsum, dim3(cdiv((int64_t)B * N, 256)), dim3(256), 0, s, x, B, S, N, out);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
