// Fused ResnetBlock FORWARD for the wide m/z levels (4 or 8 channels, rows of 8..256 positions; reference
// dquartic/model/unet1d.py:271-323): a thread owns 4 CONSECUTIVE positions of one row.
//
// Same fusion as k_res_fwd (conv3 -> RMSNorm -> (scale+1, shift) -> SiLU -> conv3 -> RMSNorm -> SiLU -> + res_conv(x) | x, one
// launch), but every global access is a 16-byte load / store (a quarter of the memory instructions of the one-position
// mapping, which left these HBM-streaming kernels at ~1.2 TB/s at sampling batch sizes), the +-1 neighbours of the conv
// inputs come from the thread's own registers except at the two edges of its group (two scalar loads for x; the block-1
// activation's edges go through LDS, ordered by a wave-level fence: a row's n/4 <= 64 threads always sit in one wave).
#include "dq_common.h"
#include "dq_kernels.h"

namespace dq {

// Packed fp32 FMAs (v_pk_fma_f32: two FMAs per lane and instruction -- the vector fp32 peak of the machine assumes them): the
// four positions of a thread as two pairs.  Element by element the nesting is the scalar one, fmaf(w0, x[q], fmaf(w1, x[q+1],
// fmaf(w2, x[q+2], acc))), so the results are bit-identical.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v fma2(float w, f2v x, f2v a) { return __builtin_elementwise_fma(f2v{w, w}, x, a); }
__device__ __forceinline__ void conv3_pk(float (&acc)[4], const float (&x)[6], float w0, float w1, float w2) {
  const f2v p0 = {x[0], x[1]}, p1 = {x[1], x[2]}, p2 = {x[2], x[3]}, p3 = {x[3], x[4]}, p4 = {x[4], x[5]};
  f2v a0 = {acc[0], acc[1]}, a1 = {acc[2], acc[3]};
  a0 = fma2(w0, p0, fma2(w1, p1, fma2(w2, p2, a0)));
  a1 = fma2(w0, p2, fma2(w1, p3, fma2(w2, p4, a1)));
  acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
}
__device__ __forceinline__ void axpy4_pk(float (&acc)[4], const float (&x)[6], float w) {  // acc[q] += w * x[q + 1]
  f2v a0 = {acc[0], acc[1]}, a1 = {acc[2], acc[3]};
  a0 = fma2(w, f2v{x[1], x[2]}, a0);
  a1 = fma2(w, f2v{x[3], x[4]}, a1);
  acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
}

// HASB: the input is cat(A, B) (up path); without it (down path) the second input's 6 C window registers do not exist and the
// kernel fits one more wave per SIMD
template <int C, bool HASB>
__global__ void __launch_bounds__(256) k_res_fwd_v4(ResFwd a) {
  __shared__ float eL[C][256], eR[C][256];
  // The block's weights in LDS, one 16-byte entry per (input channel, output channel): {w[k=0], w[k=1], w[k=2], res_conv weight} for
  // conv1 over cat(A, B) (input channels 0..2C-1, absent ones zero) and {w0, w1, w2, -} for conv2.  As wave-uniform scalar loads
  // straight from memory (fully unrolled: 600-800 of them at C = 8, in batches the 104 SGPRs can hold) their latency was exposed
  // batch after batch with two waves per SIMD: 1.4 TB/s and 15 TFLOP/s at sampling batch sizes, bound by neither.
  __shared__ __attribute__((aligned(16))) float w1q[2 * C][C][4], w2q[C][C][4];
  {
    const int cin_ = a.cinA + a.cinB;
    for (int i = threadIdx.x; i < 2 * C * C; i += blockDim.x) {
      const int co = i % C, ci = i / C;
      const bool ok = ci < cin_;
      const float* w = a.w1 + ((int64_t)co * cin_ + (ok ? ci : 0)) * 3;
      w1q[ci][co][0] = ok ? w[0] : 0.f; w1q[ci][co][1] = ok ? w[1] : 0.f; w1q[ci][co][2] = ok ? w[2] : 0.f;
      w1q[ci][co][3] = (ok && a.wr) ? a.wr[(int64_t)co * cin_ + ci] : 0.f;
    }
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
      const int co = i % C, ci = i / C;
      const float* w = a.w2 + ((int64_t)co * C + ci) * 3;
      w2q[ci][co][0] = w[0]; w2q[ci][co][1] = w[1]; w2q[ci][co][2] = w[2]; w2q[ci][co][3] = 0.f;
    }
    __syncthreads();
  }
  const int b = blockIdx.y;
  const int n = a.n, n4 = n >> 2;
  const int per_sample4 = a.rows_per_sample * n4;
  const int it = blockIdx.x * 256 + threadIdx.x;
  const bool live = it < per_sample4;
  const int row = b * a.rows_per_sample + (live ? it / n4 : 0), q4 = live ? it % n4 : 0, p0 = q4 * 4;
  const float sqC = sqrtf((float)C);
  const bool hasL = live && q4 > 0, hasR = live && q4 + 1 < n4;

  // ---- every global read of the thread, up front: window [p0-1 .. p0+4] of each input channel
  float xa[C][6], xb[C][6];
#pragma unroll
  for (int ci = 0; ci < C; ++ci) {
    const float* src = a.inA + ((int64_t)row * C + ci) * n + p0;
    const float4 v = live ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    xa[ci][0] = hasL ? src[-1] : 0.f;
    xa[ci][1] = v.x; xa[ci][2] = v.y; xa[ci][3] = v.z; xa[ci][4] = v.w;
    xa[ci][5] = hasR ? src[4] : 0.f;
  }
#pragma unroll
  for (int ci = 0; ci < (HASB ? C : 0); ++ci) {
    const bool ok = ci < a.cinB;
    const float* src = a.inB + ((int64_t)row * a.cinB + ci) * n + p0;
    const float4 v = (ok && live) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    xb[ci][0] = (ok && hasL) ? src[-1] : 0.f;
    xb[ci][1] = v.x; xb[ci][2] = v.y; xb[ci][3] = v.z; xb[ci][4] = v.w;
    xb[ci][5] = (ok && hasR) ? src[4] : 0.f;
  }

  // ---- conv1 (k3, zero padding) over cat(A, B)
  float acc[C][4];
#pragma unroll
  for (int co = 0; co < C; ++co) {
    const float bias = a.b1[co];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[co][q] = bias;
  }
#pragma unroll
  for (int ci = 0; ci < C; ++ci)
#pragma unroll
    for (int co = 0; co < C; ++co) {
      const float4 w = *reinterpret_cast<const float4*>(w1q[ci][co]);
      conv3_pk(acc[co], xa[ci], w.x, w.y, w.z);
    }
  if (HASB) {
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
      if (ci < a.cinB) {
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float4 w = *reinterpret_cast<const float4*>(w1q[C + ci][co]);
          conv3_pk(acc[co], xb[ci], w.x, w.y, w.z);
        }
      }
    }
  }
  const int64_t obase = ((int64_t)row * C) * n + p0;
  if (live && a.u1) {
#pragma unroll
    for (int co = 0; co < C; ++co)
      *reinterpret_cast<float4*>(a.u1 + obase + (int64_t)co * n) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
  }
  {
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float ssq = 0.f;
#pragma unroll
      for (int co = 0; co < C; ++co) ssq = fmaf(acc[co][q], acc[co][q], ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int co = 0; co < C; ++co) acc[co][q] = live ? silu_f(fmaf(acc[co][q] * inv * a.g1[co], ss[co] + 1.0f, ss[C + co])) : 0.f;
    }
  }
  if (live && a.a1) {
#pragma unroll
    for (int co = 0; co < C; ++co)
      *reinterpret_cast<float4*>(a.a1 + obase + (int64_t)co * n) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
  }
  // ---- the two neighbours of this group's block-1 activation, through LDS (zero outside the row)
#pragma unroll
  for (int co = 0; co < C; ++co) { eL[co][threadIdx.x] = acc[co][0]; eR[co][threadIdx.x] = acc[co][3]; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- conv2 (k3) over the block-1 activation
  float o[C][4];
#pragma unroll
  for (int co = 0; co < C; ++co) {
    const float bias = a.b2[co];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[co][q] = bias;
  }
#pragma unroll
  for (int ci = 0; ci < C; ++ci) {
    float win[6];
    win[0] = hasL ? eR[ci][threadIdx.x - 1] : 0.f;
    win[1] = acc[ci][0]; win[2] = acc[ci][1]; win[3] = acc[ci][2]; win[4] = acc[ci][3];
    win[5] = hasR ? eL[ci][threadIdx.x + 1] : 0.f;
#pragma unroll
    for (int co = 0; co < C; ++co) {
      const float4 w = *reinterpret_cast<const float4*>(w2q[ci][co]);
      conv3_pk(o[co], win, w.x, w.y, w.z);
    }
  }
  if (!live) return;
  if (a.u2) {
#pragma unroll
    for (int co = 0; co < C; ++co)
      *reinterpret_cast<float4*>(a.u2 + obase + (int64_t)co * n) = make_float4(o[co][0], o[co][1], o[co][2], o[co][3]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float ssq = 0.f;
#pragma unroll
    for (int co = 0; co < C; ++co) ssq = fmaf(o[co][q], o[co][q], ssq);
    const float inv = rms_inv(ssq, sqC);
#pragma unroll
    for (int co = 0; co < C; ++co) o[co][q] = silu_f(o[co][q] * inv * a.g2[co]);
  }
  // ---- residual: 1x1 conv over cat(A, B) or identity
  if (a.wr) {
#pragma unroll
    for (int co = 0; co < C; ++co) {
      const float br = a.br[co];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[co][q] += br;
    }
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
      for (int co = 0; co < C; ++co) axpy4_pk(o[co], xa[ci], w1q[ci][co][3]);
#pragma unroll
    for (int ci = 0; ci < (HASB ? C : 0); ++ci) {
      if (ci < a.cinB) {
#pragma unroll
        for (int co = 0; co < C; ++co) axpy4_pk(o[co], xb[ci], w1q[C + ci][co][3]);
      }
    }
  } else {
#pragma unroll
    for (int co = 0; co < C; ++co)
#pragma unroll
      for (int q = 0; q < 4; ++q) o[co][q] += xa[co][q + 1];
  }
#pragma unroll
  for (int co = 0; co < C; ++co)
    *reinterpret_cast<float4*>(a.out + obase + (int64_t)co * n) = make_float4(o[co][0], o[co][1], o[co][2], o[co][3]);
}

bool res_v4_usable(int n, int C, int cinA, int cinB) {
  return (C == 4 || C == 8) && n >= 8 && n <= 256 && (n & (n - 1)) == 0 && cinA == C && cinB <= C;
}

int launch_res_fwd_v4(const ResFwd& a, hipStream_t s) {
  const int B = a.rows / a.rows_per_sample;
  dim3 grid(cdiv((int64_t)a.rows_per_sample * (a.n / 4), 256), B), block(256);
  if (a.C == 4) {
    if (a.cinB) hipLaunchKernelGGL((k_res_fwd_v4<4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_res_fwd_v4<4, false>), grid, block, 0, s, a);
  } else {
    if (a.cinB) hipLaunchKernelGGL((k_res_fwd_v4<8, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_res_fwd_v4<8, false>), grid, block, 0, s, a);
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
