// Fused ResnetBlock for the DEEP m/z levels (rows of 1..8 positions, 12 or 16 channels; reference
// dquartic/model/unet1d.py:271-323), "channel-parallel" mapping.
//
// At these levels a (row, position) thread of k_res.hip runs a serial chain of ~1,500 dependent FMAs while the whole level
// only has 12,800 x n of them: 256 waves on 1,024 SIMDs, pure latency.  Here 16 lanes own one ROW and lane = channel: a lane
// computes ITS output channel of both convs for the row's n positions, the channel reductions of RMSNorm are 4 DPP-style
// exchanges inside the 16-lane group, and the operand a lane needs from all channels (the row's input, then the block-1
// activation) is staged in LDS as [position + halo][channel] and read as broadcast 16-byte loads.  Weights are staged once
// per block as [ci/4][tap][co][4 ci] so a lane's 4 weights for 4 input channels are one ds_read_b128.  16x more waves, each
// 16x shorter.  A block = 16 rows of ONE sample (per-sample scale/shift and their gradients need no search).
// Same arithmetic order over (ci, tap) as k_res.hip up to the 4-channel grouping of the FMA chain; results agree to fp32
// rounding (parity tests: tests/test_hip_forward.py, tests/test_hip_backward.py run every level through these kernels).
#include "dq_common.h"
#include "dq_kernels.h"
#include <algorithm>

namespace dq {

namespace {
constexpr int GR = 16;   // rows (16-lane groups) per 256-thread block
constexpr int CI = 32;   // staged input channels (cat(A, B): <= 16 + 16)

// sum over the 16 lanes of a group (all lanes receive it)
__device__ __forceinline__ float gsum16(float v) {  // four DPP adds inside the 16-lane row, no LDS
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));  // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

// stage W (cout x cin x K, row-major [co][ci][k]) as dst[((ci4 * K + k) * 16 + co) * 4 + (ci & 3)], zero padded to 16 x cin_pad
// (256 threads.  Every element's load is requested before the first store, no load is predicated: as a loop striding by blockDim.x -- not
// unrollable -- of `decode, branch, load, store` the three weight tensors of a block cost 14 memory round trips in a row per workgroup:
// the 14 channel-parallel backward launches of a train step 276 -> 251 us.)
template <int K>
__device__ __forceinline__ void stage_w(float* dst, const float* __restrict__ w, int cout, int cin, int cin_pad) {
  constexpr int NIT = 32 * K * 16 / 256;  // cin_pad <= 32
  float v[NIT];
  const int total = cin_pad * K * 16;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + (int)threadIdx.x;
    const int q = i & 3, co = (i >> 2) & 15, k = (i >> 6) % K, ci = ((i >> 6) / K) * 4 + q;
    const bool ok = i < total && co < cout && ci < cin;
    v[it] = w[ok ? (co * cin + ci) * K + k : 0];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + (int)threadIdx.x;
    const int q = i & 3, co = (i >> 2) & 15, ci = ((i >> 6) / K) * 4 + q;
    if (i < total) dst[i] = (co < cout && ci < cin) ? v[it] : 0.f;
  }
}
// transposed roles for the backward data path: dst[((co4 * K + k) * 32 + ci) * 4 + (co & 3)] = W[co][ci][k]
template <int K>
__device__ __forceinline__ void stage_wt(float* dst, const float* __restrict__ w, int cout, int cin) {
  constexpr int NIT = 16 * K * 32 / 256;
  float v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + (int)threadIdx.x;
    const int q = i & 3, ci = (i >> 2) & 31, k = (i >> 7) % K, co = ((i >> 7) / K) * 4 + q;
    v[it] = w[(co < cout && ci < cin) ? (co * cin + ci) * K + k : 0];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + (int)threadIdx.x;
    const int q = i & 3, ci = (i >> 2) & 31, co = ((i >> 7) / K) * 4 + q;
    dst[i] = (co < cout && ci < cin) ? v[it] : 0.f;
  }
}
}  // namespace

template <int C, int N>
__global__ void __launch_bounds__(256) k_res_fwd_cp(ResFwd a, int iters) {  // iters: 16-row groups per block (weights staged once)
  __shared__ __attribute__((aligned(16))) float w1s[CI * 3 * 16];
  __shared__ __attribute__((aligned(16))) float w2s[16 * 3 * 16];
  __shared__ __attribute__((aligned(16))) float wrs[CI * 16];
  __shared__ __attribute__((aligned(16))) float xs[GR][N + 2][CI];
  __shared__ __attribute__((aligned(16))) float ys[GR][N + 2][16];
  const int cin = a.cinA + a.cinB, cin4 = (cin + 3) >> 2;
  const int g = threadIdx.x >> 4, co = threadIdx.x & 15;
  const int b = blockIdx.y;
  const bool act = co < C;

  stage_w<3>(w1s, a.w1, C, cin, cin4 * 4);
  stage_w<3>(w2s, a.w2, C, C, 16);
  if (a.wr) stage_w<1>(wrs, a.wr, C, cin, cin4 * 4);
  for (int i = threadIdx.x; i < GR * (N + 2) * CI; i += blockDim.x) (&xs[0][0][0])[i] = 0.f;
  for (int i = threadIdx.x; i < GR * (N + 2) * 16; i += blockDim.x) (&ys[0][0][0])[i] = 0.f;
  // Large batches: a block walks `iters` groups of 16 rows with the weights staged once.  A group's xs / ys slice is only ever
  // touched by its own 16 lanes; the halo slots and the unused channels are never written again and stay zero.
#pragma unroll 1
  for (int itr = 0; itr < iters; ++itr) {
  const int rs = (blockIdx.x * iters + itr) * GR + g;
  const bool live = rs < a.rows_per_sample;
  const int row = b * a.rows_per_sample + (live ? rs : 0);
  __syncthreads();
  if (live) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      if (co < a.cinA) xs[g][p + 1][co] = a.inA[((int64_t)row * a.cinA + co) * N + p];
      if (co < a.cinB) xs[g][p + 1][a.cinA + co] = a.inB[((int64_t)row * a.cinB + co) * N + p];
    }
  }
  __syncthreads();

  const float sqC = sqrtf((float)C);
  float acc[N];
  // ---- conv1 (k3, zero padding) over cat(A, B): this lane's output channel
  {
    const float bias = act ? a.b1[co] : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) acc[p] = bias;
    for (int c4 = 0; c4 < cin4; ++c4) {
      float4 x4[N + 2];
#pragma unroll
      for (int q = 0; q < N + 2; ++q) x4[q] = *reinterpret_cast<const float4*>(&xs[g][q][c4 * 4]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 w4 = *reinterpret_cast<const float4*>(&w1s[((c4 * 3 + k) * 16 + co) * 4]);
#pragma unroll
        for (int p = 0; p < N; ++p)
          acc[p] = fmaf(w4.x, x4[p + k].x, fmaf(w4.y, x4[p + k].y, fmaf(w4.z, x4[p + k].z, fmaf(w4.w, x4[p + k].w, acc[p]))));
      }
    }
  }
  const int64_t obase = ((int64_t)row * C + co) * N;
  if (live && act && a.u1) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.u1[obase + p] = acc[p];
  }
  {
    const float g1 = act ? a.g1[co] : 0.f;
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
    const float sc = act ? ss[co] + 1.0f : 0.f, sh = act ? ss[C + co] : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const float ssq = gsum16(acc[p] * acc[p]);
      const float inv = rms_inv(ssq, sqC);
      acc[p] = act ? silu_f(fmaf(acc[p] * inv * g1, sc, sh)) : 0.f;
    }
  }
  if (live && act && a.a1) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.a1[obase + p] = acc[p];
  }
#pragma unroll
  for (int p = 0; p < N; ++p) ys[g][p + 1][co] = acc[p];
  __syncthreads();
  // ---- conv2 (k3) over the block-1 activation
  float o[N];
  {
    const float bias = act ? a.b2[co] : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) o[p] = bias;
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
      float4 y4[N + 2];
#pragma unroll
      for (int q = 0; q < N + 2; ++q) y4[q] = *reinterpret_cast<const float4*>(&ys[g][q][c4 * 4]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 w4 = *reinterpret_cast<const float4*>(&w2s[((c4 * 3 + k) * 16 + co) * 4]);
#pragma unroll
        for (int p = 0; p < N; ++p)
          o[p] = fmaf(w4.x, y4[p + k].x, fmaf(w4.y, y4[p + k].y, fmaf(w4.z, y4[p + k].z, fmaf(w4.w, y4[p + k].w, o[p]))));
      }
    }
  }
  if (live && act && a.u2) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.u2[obase + p] = o[p];
  }
  {
    const float g2 = act ? a.g2[co] : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const float ssq = gsum16(o[p] * o[p]);
      const float inv = rms_inv(ssq, sqC);
      o[p] = silu_f(o[p] * inv * g2);
    }
  }
  // ---- residual: 1x1 conv over cat(A, B) or identity
  if (a.wr) {
    const float br = act ? a.br[co] : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) o[p] += br;
    for (int c4 = 0; c4 < cin4; ++c4) {
      const float4 w4 = *reinterpret_cast<const float4*>(&wrs[(c4 * 16 + co) * 4]);
#pragma unroll
      for (int p = 0; p < N; ++p) {
        const float4 x4 = *reinterpret_cast<const float4*>(&xs[g][p + 1][c4 * 4]);
        o[p] = fmaf(w4.x, x4.x, fmaf(w4.y, x4.y, fmaf(w4.z, x4.z, fmaf(w4.w, x4.w, o[p]))));
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < N; ++p) o[p] += xs[g][p + 1][co];
  }
  if (live && act) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.out[obase + p] = o[p];
  }
  }  // itr
}

// -----------------------------------------------------------------------------------------------------------------
// backward data path (same outputs as k_res_bwd: dU2, dU1 for the weight-gradient kernels, dA/dB +=, per-block partial sums of dg1/dg2/dss)
// -----------------------------------------------------------------------------------------------------------------
namespace {
// pointwise backward of RMSNorm -> (scale+1, shift) -> SiLU for this lane's channel at one position; the two channel
// reductions run over the 16-lane group.  Returns dU; accumulates dg / dsc / dsh.
template <int C, bool SS>
__device__ __forceinline__ float norm_act_bwd_cp(float u, float d, float g, float sc, float sh, bool act, float& dg, float& dsc,
                                                 float& dsh) {
  const float sqC = sqrtf((float)C);
  const float ssq = gsum16(u * u);
  const float nrm = fast_sqrt(ssq), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
  const float uh = u * inv;
  const float z = uh * g * sqC;
  const float w = SS ? fmaf(z, sc, sh) : z;
  const float dw = act ? d * silu_grad_f(w) : 0.f;
  if (SS) { dsh += dw; dsc = fmaf(dw, z, dsc); }
  const float dz = SS ? dw * sc : dw;
  dg = fmaf(dz, uh * sqC, dg);
  const float gd = dz * g * sqC;
  const float dot = gsum16(gd * uh);
  return nrm < RMS_EPS ? gd * inv : inv * (gd - uh * dot);
}
}  // namespace

template <int C, int N>
__global__ void __launch_bounds__(256) k_res_bwd_cp(ResBwd a) {
  __shared__ __attribute__((aligned(16))) float w2t[4 * 3 * 32 * 4];  // [co4][k][ci][4 co]
  __shared__ __attribute__((aligned(16))) float w1t[4 * 3 * 32 * 4];
  __shared__ __attribute__((aligned(16))) float wrt[4 * 1 * 32 * 4];
  __shared__ __attribute__((aligned(16))) float d2s[GR][N + 2][16];  // dU2 with halo
  __shared__ __attribute__((aligned(16))) float d1s[GR][N + 2][16];  // dU1 with halo
  __shared__ __attribute__((aligned(16))) float dos[GR][N][16];      // d out
  __shared__ float red[GR][4 * 16];
  const int cin = a.cinA + a.cinB;
  const int g = threadIdx.x >> 4, ch = threadIdx.x & 15;
  const int b = blockIdx.y, rs = blockIdx.x * GR + g;
  const bool live = rs < a.rows_per_sample;
  const int row = b * a.rows_per_sample + (live ? rs : 0);
  const bool act = ch < C && live;
  const int64_t obase = ((int64_t)row * C + ch) * N;
  const int64_t oload = ((int64_t)row * C + (ch < C ? ch : 0)) * N;  // (loads of the padding lanes)

  // Every per-lane input of the launch is requested HERE, in front of the weight staging: d out, u2, u1, the gains and this sample's scale /
  // shift.  Read where they are used (behind the first and the second barrier) they were two more memory round trips in series per launch --
  // and a launch is one short dependent chain per workgroup (10 of them per train step at the deep levels).
  float dvp[N], u2p[N], u1p[N];
#pragma unroll
  for (int p = 0; p < N; ++p) { dvp[p] = a.dout[oload + p]; u2p[p] = a.u2[oload + p]; u1p[p] = a.u1[oload + p]; }
  const int chc = ch < C ? ch : 0;
  const float g2p = a.g2[chc], g1p = a.g1[chc];
  const float scp = (a.ss + (int64_t)b * a.ss_stride)[chc], shp = (a.ss + (int64_t)b * a.ss_stride)[C + chc];
  __builtin_amdgcn_sched_barrier(0);
  stage_wt<3>(w2t, a.w2, C, C);
  stage_wt<3>(w1t, a.w1, C, cin);
  if (a.wr) stage_wt<1>(wrt, a.wr, C, cin);
  for (int i = threadIdx.x; i < GR * (N + 2) * 16; i += blockDim.x) { (&d2s[0][0][0])[i] = 0.f; (&d1s[0][0][0])[i] = 0.f; }
  __syncthreads();

  float dg2 = 0.f, dg1 = 0.f, dsc = 0.f, dsh = 0.f;
  // ---- block2: dU2 = norm/act backward of d out (this lane's channel)
  float dout[N], d2[N];
  {
    const float g2 = ch < C ? g2p : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const float dv = dvp[p], uv = u2p[p];  // (not predicated: padding lanes re-read channel 0 / row 0 and are zeroed)
      dout[p] = act ? dv : 0.f;
      const float u = act ? uv : 0.f;
      float z0 = 0.f, z1 = 0.f;
      d2[p] = norm_act_bwd_cp<C, false>(u, dout[p], g2, 1.f, 0.f, act, dg2, z0, z1);
      if (!act) d2[p] = 0.f;
    }
  }
  if (act) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.du2[obase + p] = d2[p];
  }
#pragma unroll
  for (int p = 0; p < N; ++p) { d2s[g][p + 1][ch] = d2[p]; dos[g][p][ch] = dout[p]; }
  __syncthreads();
  // ---- d a1[ci = lane][p] = sum_co sum_k W2[co][ci][k] dU2[co][p + 1 - k]
  float da1[N];
#pragma unroll
  for (int p = 0; p < N; ++p) da1[p] = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < C / 4; ++c4) {
    float4 t4[N + 2];
#pragma unroll
    for (int q = 0; q < N + 2; ++q) t4[q] = *reinterpret_cast<const float4*>(&d2s[g][q][c4 * 4]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(&w2t[((c4 * 3 + k) * 32 + ch) * 4]);
#pragma unroll
      for (int p = 0; p < N; ++p) {  // position p + 1 - k -> halo index p + 2 - k
        const float4 t = t4[p + 2 - k];
        da1[p] = fmaf(w4.x, t.x, fmaf(w4.y, t.y, fmaf(w4.z, t.z, fmaf(w4.w, t.w, da1[p]))));
      }
    }
  }
  // ---- block1: dU1
  {
    const float g1 = ch < C ? g1p : 0.f;
    const float sc = ch < C ? scp + 1.0f : 0.f, sh = ch < C ? shp : 0.f;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const float uv = u1p[p];
      const float u = act ? uv : 0.f;
      da1[p] = norm_act_bwd_cp<C, true>(u, da1[p], g1, sc, sh, act, dg1, dsc, dsh);
      if (!act) da1[p] = 0.f;
    }
  }
  if (act) {
#pragma unroll
    for (int p = 0; p < N; ++p) a.du1[obase + p] = da1[p];
  }
#pragma unroll
  for (int p = 0; p < N; ++p) d1s[g][p + 1][ch] = da1[p];
  __syncthreads();
  // ---- d x[ci][p] = sum_co sum_k W1[co][ci][k] dU1[co][p + 1 - k] (+ residual branch), accumulated into dA / dB.
  // Lane = input channel: pass 0 covers A's channels, pass 1 B's.
  if (a.dA || a.dB) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float* dst = pass == 0 ? a.dA : a.dB;
      const int cpart = pass == 0 ? a.cinA : a.cinB;
      if (!dst || cpart == 0) continue;  // uniform
      const int ci = (pass == 0 ? 0 : a.cinA) + ch;
      float dx[N];
#pragma unroll
      for (int p = 0; p < N; ++p) dx[p] = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < C / 4; ++c4) {
        float4 t4[N + 2];
#pragma unroll
        for (int q = 0; q < N + 2; ++q) t4[q] = *reinterpret_cast<const float4*>(&d1s[g][q][c4 * 4]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 w4 = *reinterpret_cast<const float4*>(&w1t[((c4 * 3 + k) * 32 + (ci & 31)) * 4]);
#pragma unroll
          for (int p = 0; p < N; ++p) {
            const float4 t = t4[p + 2 - k];
            dx[p] = fmaf(w4.x, t.x, fmaf(w4.y, t.y, fmaf(w4.z, t.z, fmaf(w4.w, t.w, dx[p]))));
          }
        }
        if (a.wr) {
          const float4 w4 = *reinterpret_cast<const float4*>(&wrt[(c4 * 32 + (ci & 31)) * 4]);
#pragma unroll
          for (int p = 0; p < N; ++p) {
            const float4 t = *reinterpret_cast<const float4*>(&dos[g][p][c4 * 4]);
            dx[p] = fmaf(w4.x, t.x, fmaf(w4.y, t.y, fmaf(w4.z, t.z, fmaf(w4.w, t.w, dx[p]))));
          }
        }
      }
      if (!a.wr) {  // identity residual: cin == C, single input
#pragma unroll
        for (int p = 0; p < N; ++p) dx[p] += dout[p];
      }
      if (live && ch < cpart) {
        float* o = dst + ((int64_t)row * cpart + ch) * N;
        float oldv[N];
#pragma unroll
        for (int p = 0; p < N; ++p) oldv[p] = (pass == 0 ? a.dA_store : a.dB_store) ? 0.f : o[p];  // all reads before the first store
#pragma unroll
        for (int p = 0; p < N; ++p) o[p] = oldv[p] + dx[p];
      }
    }
  }
  // ---- reductions over the block's 16 rows: dg2, dg1, this sample's d(scale), d(shift)
  red[g][ch] = dg2; red[g][16 + ch] = dg1; red[g][32 + ch] = dsc; red[g][48 + ch] = dsh;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x & 15, what = threadIdx.x >> 4;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < GR; ++r) v += red[r][what * 16 + c];
    // [dg2 | dg1 | dscale | dshift] of this block into its own slot (summed in block order by launch_part_reduce)
    if (c < C) a.gpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * C) + what * C + c] = v;
  }
}

bool res_cp_usable(int n, int C, int cinA, int cinB) {
  return (C == 12 || C == 16) && (n == 1 || n == 2 || n == 4 || n == 8) && cinA <= 16 && cinB <= 16;
}

int launch_res_fwd_cp(const ResFwd& a, hipStream_t s) {
  const int B = a.rows / a.rows_per_sample;
  // training batches: one 16-row group per block (parallelism); sampling batches: up to 8 groups per block while >= ~2048
  // blocks remain, so that the per-block weight staging amortises
  const int groups = cdiv(a.rows_per_sample, GR);
  const int iters = std::max(1, std::min({8, groups, (int)((int64_t)groups * B / 2048)}));
  dim3 grid(cdiv(groups, iters), B), block(256);
#define DQ_CP(CC, NN) \
  if (a.C == CC && a.n == NN) { hipLaunchKernelGGL((k_res_fwd_cp<CC, NN>), grid, block, 0, s, a, iters); DQ_LAUNCH_CHECK(); return 0; }
  DQ_CP(12, 1) DQ_CP(12, 2) DQ_CP(12, 4) DQ_CP(12, 8) DQ_CP(16, 1) DQ_CP(16, 2) DQ_CP(16, 4) DQ_CP(16, 8)
#undef DQ_CP
  set_error("res_fwd_cp: unsupported (C, n)");
  return 2;
}

int launch_res_bwd_cp(const ResBwd& a, hipStream_t s) {
  const int B = a.rows / a.rows_per_sample;
  dim3 grid(cdiv(a.rows_per_sample, GR), B), block(256);
  ResBwd k = a;  // partial sums of the norm gains and scale / shift (see ResBwd::gpart)
  DQ_REQUIRE(k.gpart && k.gblocks && k.gpart_floats >= (int64_t)grid.x * grid.y * 4 * a.C, "res_bwd_cp: partial-sum slot missing or too small");
  *k.gblocks = (int)grid.x;  // blocks per sample
#define DQ_CP(CC, NN) \
  if (a.C == CC && a.n == NN) { hipLaunchKernelGGL((k_res_bwd_cp<CC, NN>), grid, block, 0, s, k); DQ_LAUNCH_CHECK(); return 0; }
  DQ_CP(12, 1) DQ_CP(12, 2) DQ_CP(12, 4) DQ_CP(12, 8) DQ_CP(16, 1) DQ_CP(16, 2) DQ_CP(16, 4) DQ_CP(16, 8)
#undef DQ_CP
  set_error("res_bwd_cp: unsupported (C, n)");
  return 2;
}

}  // namespace dq
