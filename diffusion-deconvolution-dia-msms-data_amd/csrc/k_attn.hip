// K7: the bottleneck's RT x RT softmax attention (queries/values from the folded U-Net state, keys from the MS1
// chromatogram features), with RoPE on q and k.  Forward and backward.
// Reference arithmetic: dquartic/model/unet1d.py:552-567 (Attention.forward, x-attn branch), :428-443 (Attend:
// softmax(q k^T * 32^-0.5) v).  RoPE comes from the third-party rotary_embedding_torch (unet1d.py:529, 560-561):
// default 'lang' mode, adjacent-pair rotation of the first 16 of the 32 head channels -- restated, parity unpinned.
//
// q, k, v, o live in conv layout (B, heads*32, RT), RT contiguous.  One thread owns one query row (forward, dQ) or one
// key row (dK, dV): its 32-wide vectors and accumulators stay in registers, the other side is streamed through LDS
// 64 rows at a time and read with wave-uniform (broadcast) ds_read_b128, 4 rows per read.  Flash-style online softmax;
// the score matrix never exists in memory.  3.4 % of the network FLOPs at 64x400 -- kept simple on purpose.
#include "dq_common.h"
#include "dq_kernels.h"

namespace dq {

constexpr float ATT_SCALE = 0.17677669529663687f;  // 32^-0.5
constexpr int TJ = 64;                              // rows per LDS tile

// ---- RoPE, in place.  sign = +1 forward, -1 backward (transpose of the rotation).
__global__ void __launch_bounds__(256) k_rope(float* __restrict__ t, const float* __restrict__ freqs, int64_t batch_stride, int RT,
                                              float sign, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pos = (int)(i % RT);
  const int pr = (int)((i / RT) % 8);
  const int64_t bh = i / ((int64_t)RT * 8);  // b * 4 + head
  const int64_t b = bh >> 2, h = bh & 3;
  const float ang = (float)pos * freqs[pr];
  float sn, cs;
  sincosf(ang, &sn, &cs);
  sn *= sign;
  float* pa = t + b * batch_stride + (h * 32 + 2 * pr) * RT + pos;
  const float xa = pa[0], xb = pa[RT];
  pa[0] = xa * cs - xb * sn;
  pa[RT] = xb * cs + xa * sn;
}

int launch_rope(float* qk, const float* freqs, int B, int64_t batch_stride, int RT, float sign, hipStream_t s) {
  // rotates the 4 heads x 32 channels that start at each sample's base (B, >=128, RT); batch_stride in floats
  const int64_t total = (int64_t)B * 4 * 8 * RT;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_rope, dim3(cdiv(total, 256)), dim3(256), 0, s, qk, freqs, batch_stride, RT, sign, total);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- forward: thread = query row
__global__ void __launch_bounds__(64) k_attn_fwd(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k, int64_t k_bs,
                                                 const float* __restrict__ v, int64_t v_bs, float* __restrict__ o,
                                                 float* __restrict__ lse, int RT) {
  __shared__ __attribute__((aligned(16))) float k_lds[32][TJ];
  __shared__ __attribute__((aligned(16))) float v_lds[32][TJ];
  const int lane = threadIdx.x, bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int i = blockIdx.x * 64 + lane;
  const bool valid = i < RT;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  float qr[32], oa[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    qr[c] = valid ? qb[(int64_t)c * RT + i] * ATT_SCALE : 0.f;
    oa[c] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < RT; j0 += TJ) {
    __syncthreads();
    {
      const int j = j0 + lane;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        k_lds[c][lane] = j < RT ? kb[(int64_t)c * RT + j] : 0.f;
        v_lds[c][lane] = j < RT ? vb[(int64_t)c * RT + j] : 0.f;
      }
    }
    __syncthreads();
    const int jmax = min(TJ, RT - j0);
    for (int jj = 0; jj < jmax; jj += 4) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 kk = *reinterpret_cast<const float4*>(&k_lds[c][jj]);
        s0 = fmaf(qr[c], kk.x, s0); s1 = fmaf(qr[c], kk.y, s1); s2 = fmaf(qr[c], kk.z, s2); s3 = fmaf(qr[c], kk.w, s3);
      }
      if (jj + 1 >= jmax) s1 = -INFINITY;
      if (jj + 2 >= jmax) s2 = -INFINITY;
      if (jj + 3 >= jmax) s3 = -INFINITY;
      const float mn = fmaxf(fmaxf(m, s0), fmaxf(fmaxf(s1, s2), s3));
      const float al = expf(m - mn);
      const float p0 = expf(s0 - mn), p1 = expf(s1 - mn), p2 = expf(s2 - mn), p3 = expf(s3 - mn);
      l = fmaf(l, al, (p0 + p1) + (p2 + p3));
      m = mn;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 vv = *reinterpret_cast<const float4*>(&v_lds[c][jj]);
        oa[c] = fmaf(oa[c], al, fmaf(p0, vv.x, fmaf(p1, vv.y, fmaf(p2, vv.z, p3 * vv.w))));
      }
    }
  }
  if (valid) {
    const float rl = 1.0f / l;
    float* ob = o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
#pragma unroll
    for (int c = 0; c < 32; ++c) ob[(int64_t)c * RT + i] = oa[c] * rl;
    if (lse) lse[(int64_t)bh * RT + i] = m + logf(l);
  }
}

int launch_attn_fwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, float* o, float* lse,
                    int B, int RT, hipStream_t s) {
  if (B == 0 || RT == 0) return 0;
  hipLaunchKernelGGL(k_attn_fwd, dim3(cdiv(RT, 64), B * 4), dim3(64), 0, s, q, q_bs, k, k_bs, v, v_bs, o, lse, RT);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- backward, query side: delta_i = dO_i . O_i ; dQ_i = 32^-0.5 * sum_j P_ij (dP_ij - delta_i) K_j
__global__ void __launch_bounds__(64) k_attn_bwd_q(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k,
                                                   int64_t k_bs, const float* __restrict__ v, int64_t v_bs,
                                                   const float* __restrict__ o, const float* __restrict__ d_o,
                                                   const float* __restrict__ lse, float* __restrict__ delta,
                                                   float* __restrict__ dq, int64_t dq_bs, int RT) {
  __shared__ __attribute__((aligned(16))) float k_lds[32][TJ];
  __shared__ __attribute__((aligned(16))) float v_lds[32][TJ];
  const int lane = threadIdx.x, bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int i = blockIdx.x * 64 + lane;
  const bool valid = i < RT;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  const float* ob = o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  const float* dob = d_o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  float qr[32], dor[32], dqa[32];
  float dl = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    qr[c] = valid ? qb[(int64_t)c * RT + i] * ATT_SCALE : 0.f;
    dor[c] = valid ? dob[(int64_t)c * RT + i] : 0.f;
    dl = fmaf(dor[c], valid ? ob[(int64_t)c * RT + i] : 0.f, dl);
    dqa[c] = 0.f;
  }
  const float ls = valid ? lse[(int64_t)bh * RT + i] : INFINITY;
  for (int j0 = 0; j0 < RT; j0 += TJ) {
    __syncthreads();
    {
      const int j = j0 + lane;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        k_lds[c][lane] = j < RT ? kb[(int64_t)c * RT + j] : 0.f;
        v_lds[c][lane] = j < RT ? vb[(int64_t)c * RT + j] : 0.f;
      }
    }
    __syncthreads();
    const int jmax = min(TJ, RT - j0);
    for (int jj = 0; jj < jmax; jj += 4) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 kk = *reinterpret_cast<const float4*>(&k_lds[c][jj]);
        const float4 vv = *reinterpret_cast<const float4*>(&v_lds[c][jj]);
        s[0] = fmaf(qr[c], kk.x, s[0]); s[1] = fmaf(qr[c], kk.y, s[1]); s[2] = fmaf(qr[c], kk.z, s[2]); s[3] = fmaf(qr[c], kk.w, s[3]);
        dp[0] = fmaf(dor[c], vv.x, dp[0]); dp[1] = fmaf(dor[c], vv.y, dp[1]); dp[2] = fmaf(dor[c], vv.z, dp[2]); dp[3] = fmaf(dor[c], vv.w, dp[3]);
      }
      float ds[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ds[u] = (jj + u < jmax) ? expf(s[u] - ls) * (dp[u] - dl) : 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 kk = *reinterpret_cast<const float4*>(&k_lds[c][jj]);
        dqa[c] = fmaf(ds[0], kk.x, fmaf(ds[1], kk.y, fmaf(ds[2], kk.z, fmaf(ds[3], kk.w, dqa[c]))));
      }
    }
  }
  if (valid) {
    float* dqb = dq + b * dq_bs + (int64_t)h * 32 * RT;
#pragma unroll
    for (int c = 0; c < 32; ++c) dqb[(int64_t)c * RT + i] = dqa[c] * ATT_SCALE;
    delta[(int64_t)bh * RT + i] = dl;
  }
}

// ---- backward, key side: dV_j = sum_i P_ij dO_i ; dK_j = 32^-0.5 * sum_i P_ij (dP_ij - delta_i) Q_i
__global__ void __launch_bounds__(64) k_attn_bwd_kv(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k,
                                                    int64_t k_bs, const float* __restrict__ v, int64_t v_bs,
                                                    const float* __restrict__ d_o, const float* __restrict__ lse,
                                                    const float* __restrict__ delta, float* __restrict__ dk, int64_t dk_bs,
                                                    float* __restrict__ dv, int64_t dv_bs, int RT) {
  __shared__ __attribute__((aligned(16))) float q_lds[32][TJ];
  __shared__ __attribute__((aligned(16))) float do_lds[32][TJ];
  __shared__ __attribute__((aligned(16))) float ls_lds[TJ];
  __shared__ __attribute__((aligned(16))) float dl_lds[TJ];
  const int lane = threadIdx.x, bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int j = blockIdx.x * 64 + lane;
  const bool valid = j < RT;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  const float* dob = d_o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  float kr[32], vr[32], dka[32], dva[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    kr[c] = valid ? kb[(int64_t)c * RT + j] : 0.f;
    vr[c] = valid ? vb[(int64_t)c * RT + j] : 0.f;
    dka[c] = 0.f;
    dva[c] = 0.f;
  }
  for (int i0 = 0; i0 < RT; i0 += TJ) {
    __syncthreads();
    {
      const int i = i0 + lane;
      const bool ok = i < RT;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        q_lds[c][lane] = ok ? qb[(int64_t)c * RT + i] * ATT_SCALE : 0.f;
        do_lds[c][lane] = ok ? dob[(int64_t)c * RT + i] : 0.f;
      }
      ls_lds[lane] = ok ? lse[(int64_t)bh * RT + i] : INFINITY;  // exp(s - inf) = 0 masks the tail
      dl_lds[lane] = ok ? delta[(int64_t)bh * RT + i] : 0.f;
    }
    __syncthreads();
    for (int ii = 0; ii < TJ; ii += 4) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 qq = *reinterpret_cast<const float4*>(&q_lds[c][ii]);
        const float4 dd = *reinterpret_cast<const float4*>(&do_lds[c][ii]);
        s[0] = fmaf(kr[c], qq.x, s[0]); s[1] = fmaf(kr[c], qq.y, s[1]); s[2] = fmaf(kr[c], qq.z, s[2]); s[3] = fmaf(kr[c], qq.w, s[3]);
        dp[0] = fmaf(vr[c], dd.x, dp[0]); dp[1] = fmaf(vr[c], dd.y, dp[1]); dp[2] = fmaf(vr[c], dd.z, dp[2]); dp[3] = fmaf(vr[c], dd.w, dp[3]);
      }
      const float4 l4 = *reinterpret_cast<const float4*>(&ls_lds[ii]);
      const float4 d4 = *reinterpret_cast<const float4*>(&dl_lds[ii]);
      float p[4], ds[4];
      p[0] = expf(s[0] - l4.x); p[1] = expf(s[1] - l4.y); p[2] = expf(s[2] - l4.z); p[3] = expf(s[3] - l4.w);
      ds[0] = p[0] * (dp[0] - d4.x); ds[1] = p[1] * (dp[1] - d4.y); ds[2] = p[2] * (dp[2] - d4.z); ds[3] = p[3] * (dp[3] - d4.w);
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 qq = *reinterpret_cast<const float4*>(&q_lds[c][ii]);
        const float4 dd = *reinterpret_cast<const float4*>(&do_lds[c][ii]);
        dva[c] = fmaf(p[0], dd.x, fmaf(p[1], dd.y, fmaf(p[2], dd.z, fmaf(p[3], dd.w, dva[c]))));
        dka[c] = fmaf(ds[0], qq.x, fmaf(ds[1], qq.y, fmaf(ds[2], qq.z, fmaf(ds[3], qq.w, dka[c]))));
      }
    }
  }
  if (valid) {
    float* dkb = dk + b * dk_bs + (int64_t)h * 32 * RT;
    float* dvb = dv + b * dv_bs + (int64_t)h * 32 * RT;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      dkb[(int64_t)c * RT + j] = dka[c];  // q_lds already carries the 32^-0.5
      dvb[(int64_t)c * RT + j] = dva[c];
    }
  }
}

int launch_attn_bwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, const float* o,
                    const float* d_o, const float* lse, float* delta, float* dq, int64_t dq_bs, float* dk, int64_t dk_bs, float* dv,
                    int64_t dv_bs, int B, int RT, hipStream_t s) {
  if (B == 0 || RT == 0) return 0;
  dim3 grid(cdiv(RT, 64), B * 4), block(64);
  hipLaunchKernelGGL(k_attn_bwd_q, grid, block, 0, s, q, q_bs, k, k_bs, v, v_bs, o, d_o, lse, delta, dq, dq_bs, RT);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attn_bwd_kv, grid, block, 0, s, q, q_bs, k, k_bs, v, v_bs, d_o, lse, delta, dk, dk_bs, dv, dv_bs, RT);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
