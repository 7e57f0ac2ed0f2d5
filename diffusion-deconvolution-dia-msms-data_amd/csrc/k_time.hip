// K1: timestep embedding MLP and every per-block scale/shift projection that hangs off it.
// Reference arithmetic: dquartic/model/unet1d.py:196-218 (SinusoidalPosEmb), :956-960 (time_mlp: Linear ->
// exact GELU -> Linear), :292-296 and :315-318 (ResnetBlock.mlp = SiLU -> Linear(time_dim, 2*C_out)), :662-678
// (ConditionalScaleShift of the mixture input).  All the SiLU->Linear heads read the same silu(temb), so they are
// evaluated as ONE (ss_total x 16) mat-vec per sample; row r of that virtual matrix is described by the device
// tables ss_w_off[r] / ss_b_off[r] (offsets into the flat parameter buffer).
// Launch-bound work (B x ~10 kFLOP): one wave per sample, everything in registers/LDS.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_plan.h"
#include "dq_unet.h"

namespace dq {

// tbuf per sample: [0,4) sinu | [4,20) h_pre | [20,36) h_act | [36,52) temb | [52,68) silu(temb) | [68,84) dtemb | [84,100) dh_pre
// one SiLU -> Linear(16, .) output: bias + sum_i w[i] * silu(temb)[i], in this order everywhere it is evaluated
__device__ __forceinline__ float ss_head_row(const float* __restrict__ w, float bias, const float* st) {
  float v = bias;
#pragma unroll
  for (int i = 0; i < 16; ++i) v = fmaf(w[i], st[i], v);
  return v;
}

constexpr int TIME_FWD_RPT = 3;  // head rows per thread and pass of k_time_fwd (256 threads: 768 rows per pass)
__global__ void __launch_bounds__(256) k_time_fwd(const float* __restrict__ P, const int64_t* __restrict__ t, int t_scalar,
                                                 float* __restrict__ tbuf, float* __restrict__ ss, int ss_total,
                                                 const int64_t* __restrict__ ss_w_off, const int64_t* __restrict__ ss_b_off,
                                                 const float* __restrict__ t1w, const float* __restrict__ t1b,
                                                 const float* __restrict__ t2w, const float* __restrict__ t2b, int dim, float theta,
                                                 const int* __restrict__ step_tab, const int* __restrict__ step_ptr) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float* tb = tbuf + (int64_t)b * TBUF_FLOATS;
  __shared__ float sinu[4], hact[16], st[16];
  // The first pass of head rows (offset table -> 16 weights + bias per row) is requested HERE, in front of the time MLP it does not depend on:
  // the MLP's three dependent steps run under those two memory round trips.  (64 threads walking ss_total / 64 rows one after the other behind
  // the MLP were ~20 round trips in series: 11 us at the head of every forward, 15 us per sampling step.)
  float hw[TIME_FWD_RPT][16], hb[TIME_FWD_RPT];
  if (ss) {
    int64_t wo[TIME_FWD_RPT], bo[TIME_FWD_RPT];
#pragma unroll
    for (int u = 0; u < TIME_FWD_RPT; ++u) {
      const int r = u * 256 + tid, rc = r < ss_total ? r : ss_total - 1;
      wo[u] = ss_w_off[rc]; bo[u] = ss_b_off[rc];
    }
#pragma unroll
    for (int u = 0; u < TIME_FWD_RPT; ++u) {
      hb[u] = P[bo[u]];
#pragma unroll
      for (int i = 0; i < 16; ++i) hw[u][i] = P[wo[u] + i];
    }
  }
  // timestep: per-sample tensor, or a scalar, or (graph replay) table[*step] read on the device
  const float tv = (float)(t ? t[b] : (int64_t)(step_tab ? step_tab[*step_ptr] : t_scalar));  // int64 * fp32 -> fp32 (unet1d.py:215)
  const int half = dim / 2;
  if (tid < dim) {
    const int j = tid % half;
    const float kf = (float)(-log((double)theta) / (double)(half - 1));
    const float f = expf((float)j * kf);
    const float e = tv * f;
    const float v = tid < half ? sinf(e) : cosf(e);
    sinu[tid] = v;
    tb[tid] = v;
  }
  __syncthreads();
  if (tid < 16) {
    float h = t1b[tid];
    for (int i = 0; i < dim; ++i) h = fmaf(t1w[tid * dim + i], sinu[i], h);
    tb[4 + tid] = h;
    const float a = gelu_f(h);
    tb[20 + tid] = a;
    hact[tid] = a;
  }
  __syncthreads();
  if (tid < 16) {
    float e = t2b[tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) e = fmaf(t2w[tid * 16 + i], hact[i], e);
    tb[36 + tid] = e;
    const float sv = silu_f(e);
    tb[52 + tid] = sv;
    st[tid] = sv;
  }
  __syncthreads();
  if (!ss) return;
#pragma unroll
  for (int u = 0; u < TIME_FWD_RPT; ++u) {
    const int r = u * 256 + tid;
    if (r < ss_total) ss[(int64_t)b * ss_total + r] = ss_head_row(hw[u], hb[u], st);
  }
  for (int r = TIME_FWD_RPT * 256 + tid; r < ss_total; r += 256) ss[(int64_t)b * ss_total + r] = ss_head_row(P + ss_w_off[r], P[ss_b_off[r]], st);  // (networks with more than 768 head rows)
}

// stand-alone heads (dq_scale_shift_fwd): ss[b][r] = Linear(SiLU(temb_b))[r] for a caller-supplied temb
__global__ void __launch_bounds__(64) k_ss_heads(const float* __restrict__ temb, const float* __restrict__ w, const float* __restrict__ bias,
                                                 float* __restrict__ ss, int m, int stride) {
  __shared__ float st[16];
  const int b = blockIdx.x;
  if (threadIdx.x < 16) st[threadIdx.x] = silu_f(temb[(int64_t)b * 16 + threadIdx.x]);
  __syncthreads();
  for (int r = threadIdx.x; r < m; r += 64) ss[(int64_t)b * stride + r] = ss_head_row(w + (int64_t)r * 16, bias[r], st);
}

int launch_ss_heads_strided(const float* temb, const float* w, const float* bias, float* ss, int stride, int B, int m, hipStream_t s) {
  if (B == 0 || m == 0) return 0;
  hipLaunchKernelGGL(k_ss_heads, dim3(B), dim3(64), 0, s, temb, w, bias, ss, m, stride);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_ss_heads(const float* temb, const float* w, const float* bias, float* ss, int B, int m, hipStream_t s) {
  if (B == 0 || m == 0) return 0;
  hipLaunchKernelGGL(k_ss_heads, dim3(B), dim3(64), 0, s, temb, w, bias, ss, m, m);
  DQ_LAUNCH_CHECK();
  return 0;
}

// stand-alone time MLP (dq_time_mlp_fwd): fills tbuf only (no heads)
int launch_time_mlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, const int64_t* t, float* tbuf, int B,
                        hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_time_fwd, dim3(B), dim3(256), 0, s, (const float*)nullptr, t, 0, tbuf, (float*)nullptr, 0, (const int64_t*)nullptr,
                     (const int64_t*)nullptr, w1, b1, w2, b2, 4, 10000.0f, (const int*)nullptr, (const int*)nullptr);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_time_embed_fwd(const Plan& p, const DevTables& dt, const float* params, const int64_t* t, int t_scalar, float* tbuf,
                          float* ss, int B, const int* step_tab, const int* step_ptr, hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_time_fwd, dim3(B), dim3(256), 0, s, params, t, t_scalar, tbuf, ss, p.ss_total, dt.ss_w_off, dt.ss_b_off,
                     params + p.t1_w, params + p.t1_b, params + p.t2_w, params + p.t2_b, p.dim, 10000.0f, step_tab, step_ptr);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- backward ----
constexpr int TIME_MAX_B = 128;  // samples whose silu(temb) k_time_bwd_rows stages in LDS (larger batches read them from memory)
constexpr int TIME_RPT = 3;      // rows per thread and pass of k_time_bwd_sample
// (1) per ss row r: dW[r][:] += sum_b dss[b][r] * silu(temb_b) ; db[r] += sum_b dss[b][r]
// (launched together with (3): the last block of the launch is the time_mlp block below -- (1) does not depend on (2), (3) does, so the
// order is (2), then (1) + (3) in one launch)
__device__ __forceinline__ void time_bwd_mlp_block(float* __restrict__ G, const float* __restrict__ tbuf, int B, int64_t t1w, int64_t t1b, int64_t t2w,
                                                   int64_t t2b, int dim);
__global__ void __launch_bounds__(256) k_time_bwd_rows(float* __restrict__ G, const float* __restrict__ tbuf,
                                                       const float* __restrict__ dss, int ss_total, int B,
                                                       const int64_t* __restrict__ ss_w_off, const int64_t* __restrict__ ss_b_off, int64_t t1w,
                                                       int64_t t1b, int64_t t2w, int64_t t2b, int dim) {
  if (blockIdx.x == gridDim.x - 1) {  // (block-uniform)
    time_bwd_mlp_block(G, tbuf, B, t1w, t1b, t2w, t2b, dim);
    return;
  }
  // silu(temb) of every sample in LDS (B x 16 floats), then the batch loop with EIGHT samples' dss loads in flight: as `for (b) { d = dss[b][r];
  // st = tbuf + ...; fma x 16 }` over a run-time B the loop was 32 dependent memory round trips per thread -- 13 us at the very end of the step
  // (nothing overlaps the time-embedding backward: it waits for every d(scale, shift) sum and the optimiser waits for it).
  __shared__ float st_lds[TIME_MAX_B * 16];
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int rc = r < ss_total ? r : ss_total - 1;
  const bool staged = B <= TIME_MAX_B;
  if (staged) {
    for (int i = threadIdx.x; i < B * 16; i += 256) st_lds[i] = tbuf[(int64_t)(i >> 4) * TBUF_FLOATS + 52 + (i & 15)];
    __syncthreads();
  }
  const int64_t wo = ss_w_off[rc], bo = ss_b_off[rc];
  float dw[16], db = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dw[i] = 0.f;
  for (int b0 = 0; b0 < B; b0 += 8) {
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = b0 + u < B ? dss[(int64_t)(b0 + u) * ss_total + rc] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b0 + u < B) {  // (wave-uniform)
        const float* st = staged ? st_lds + (b0 + u) * 16 : tbuf + (int64_t)(b0 + u) * TBUF_FLOATS + 52;
        db += d[u];
#pragma unroll
        for (int i = 0; i < 16; ++i) dw[i] = fmaf(d[u], st[i], dw[i]);
      }
    }
  }
  if (r >= ss_total) return;
  float* gw = G + wo;
#pragma unroll
  for (int i = 0; i < 16; ++i) gw[i] += dw[i];
  G[bo] += db;
}

// (2) per sample: d silu(temb) = sum_r dss[b][r] W[r][:] -> dtemb -> dh_act -> dh_pre (stored in tbuf)
__global__ void __launch_bounds__(256) k_time_bwd_sample(const float* __restrict__ P, float* __restrict__ tbuf,
                                                         const float* __restrict__ dss, int ss_total,
                                                         const int64_t* __restrict__ ss_w_off, int64_t t2w) {
  // 256 threads per sample, up to TIME_RPT rows each with every load in flight (offset table + dss, then the 16 weights of each row): two memory
  // round trips.  (64 threads walking ss_total / 64 rows one after the other, offset -> weights -> next row, were ~20 dependent round trips: 14 us.)
  const int b = blockIdx.x, tid = threadIdx.x;
  float* tb = tbuf + (int64_t)b * TBUF_FLOATS;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int r0 = 0; r0 < ss_total; r0 += 256 * TIME_RPT) {
    float d[TIME_RPT];
    int64_t wo[TIME_RPT];
#pragma unroll
    for (int u = 0; u < TIME_RPT; ++u) {
      const int r = r0 + u * 256 + tid;
      const int rc = r < ss_total ? r : ss_total - 1;
      d[u] = r < ss_total ? dss[(int64_t)b * ss_total + rc] : 0.f;
      wo[u] = ss_w_off[rc];
    }
    float4 w[TIME_RPT][4];
#pragma unroll
    for (int u = 0; u < TIME_RPT; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // (the rows of the flat parameter buffer are not 16-byte aligned in general: four scalar loads each)
        const float* wp = P + wo[u] + 4 * q;
        w[u][q] = make_float4(wp[0], wp[1], wp[2], wp[3]);
      }
#pragma unroll
    for (int u = 0; u < TIME_RPT; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[4 * q + 0] = fmaf(d[u], w[u][q].x, acc[4 * q + 0]); acc[4 * q + 1] = fmaf(d[u], w[u][q].y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(d[u], w[u][q].z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(d[u], w[u][q].w, acc[4 * q + 3]);
      }
  }
  __shared__ float part[4][16];
  __shared__ float dtemb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = wave_sum(acc[i]);
    if ((tid & 63) == 0) part[tid >> 6][i] = v;
  }
  __syncthreads();
  if (tid < 16) {
    const float v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);  // fixed order
    const float dt = v * silu_grad_f(tb[36 + tid]);
    dtemb[tid] = dt;
    tb[68 + tid] = dt;
  }
  __syncthreads();
  if (tid < 16) {
    float dh = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) dh = fmaf(dtemb[o], P[t2w + o * 16 + tid], dh);
    tb[84 + tid] = dh * gelu_grad_f(tb[4 + tid]);
  }
}

// (3) weights of the two time_mlp Linears: thread (o, i) loops over the batch
__device__ __forceinline__ void time_bwd_mlp_block(float* __restrict__ G, const float* __restrict__ tbuf, int B, int64_t t1w, int64_t t1b, int64_t t2w,
                                                   int64_t t2b, int dim) {
  const int tid = threadIdx.x, o = tid >> 4, i = tid & 15;
  float dw2 = 0.f, db2 = 0.f, dw1 = 0.f, db1 = 0.f;
  // the four 16-vectors of every sample this block reads (x | h_act | d temb | d h_pre) staged in LDS with eight loads in flight per thread;
  // as a loop over the samples with four dependent loads each it was the longest block of the launch (~10 us)
  __shared__ float sv[TIME_MAX_B * 64];
  const bool staged = B <= TIME_MAX_B;
  if (staged) {
    for (int base = 0; base < B * 64; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * 256 + tid, ec = e < B * 64 ? e : 0;
        const int bb = ec >> 6, q = (ec >> 4) & 3, j = ec & 15;
        v[u] = tbuf[(int64_t)bb * TBUF_FLOATS + (q == 0 ? 0 : q == 1 ? 20 : q == 2 ? 68 : 84) + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = base + u * 256 + tid; if (e < B * 64) sv[e] = v[u]; }
    }
    __syncthreads();
  }
  for (int b = 0; b < B; ++b) {
    const float* tb = tbuf + (int64_t)b * TBUF_FLOATS;
    const float dt = staged ? sv[b * 64 + 32 + o] : tb[68 + o], dh = staged ? sv[b * 64 + 48 + o] : tb[84 + o];
    dw2 = fmaf(dt, staged ? sv[b * 64 + 16 + i] : tb[20 + i], dw2);
    db2 += dt;
    if (i < dim) dw1 = fmaf(dh, staged ? sv[b * 64 + i] : tb[i], dw1);
    db1 += dh;
  }
  G[t2w + o * 16 + i] += dw2;
  if (i < dim) G[t1w + o * dim + i] += dw1;
  if (i == 0) { G[t2b + o] += db2; G[t1b + o] += db1; }
}

int launch_time_embed_bwd(const Plan& p, const DevTables& dt, const float* params, float* grads, float* tbuf, const float* dss,
                          int B, hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_time_bwd_sample, dim3(B), dim3(256), 0, s, params, tbuf, dss, p.ss_total, dt.ss_w_off, p.t2_w);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_time_bwd_rows, dim3(cdiv(p.ss_total, 256) + 1), dim3(256), 0, s, grads, tbuf, dss, p.ss_total, B, dt.ss_w_off, dt.ss_b_off,
                     p.t1_w, p.t1_b, p.t2_w, p.t2_b, p.dim);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
