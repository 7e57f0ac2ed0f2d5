// ResnetBlock backward data path for the DEEP levels (12 / 16 channels, m/z rows of 2 / 4 / 8 positions; reference
// dquartic/model/unet1d.py:271-323, autograd of it) with the m/z ROW as the lane column of v_mfma_f32_16x16x4_f32 -- the layout of
// k_la_rows_bwd.hip and k_wgrad_rows:
//   lane = (g = lane / 16, row = lane % 16), a wave = 16 rows of one sample, a row's N positions live in the lane's registers;
//   a channel set of 4 CPL channels is held as "register r of lane (g, row) = channel CPL g + r" (r < CPL), which is the accumulator
//   layout of a product whose A rows are arranged as i = 4 g' + r' -> channel CPL g' + r' (rows with r' >= CPL are zero) AND the B operand of
//   the next product (K-step (r, tap): lane group g supplies channel CPL g + r).
// A k = 3 convolution along the row is then, per position p, CPL x 3 MFMAs whose B operands are the lane's OWN registers of positions
// p - 1, p, p + 1 (taps outside the row are skipped at compile time): nothing is exchanged between lanes except RMSNorm's channel sums
// (4 registers in the lane + a sum over the four lane groups, v_permlane16_swap / v_permlane32_swap).  The weights (W2^T, W1^T for both
// halves of cat(A, B), Wr^T) are 12 + 24 + 8 registers per lane, gathered once per wave.
// k_res_cp.hip (lane = channel, 16 lanes = a row, operands and weights as LDS broadcasts, three workgroup barriers) takes 14 - 26 us per
// launch at 12,800 rows where its floor is 6 - 11 (batch 1); this form: see DESIGN.md section 17.
// Same contract as k_res_bwd / k_res_bwd_cp (ResBwd): dU1 / dU2 written for the weight-gradient kernels, d(input) stored or accumulated
// into dA / dB, every workgroup's [d g2 | d g1 | d scale | d shift] sums in its own gpart slot (launch_part_reduce: ordered, repeatable).
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
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
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

// N contiguous floats (N = 2 / 4 / 8; the launcher checks 16-byte alignment of the tensors)
template <int N>
__device__ __forceinline__ void ld_n(const float* p, float (&v)[N]) {
  if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}
template <int N>
__device__ __forceinline__ void st_n(float* p, const float (&v)[N]) {
  if constexpr (N == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}

template <int C, int N, bool WR>
__global__ void __launch_bounds__(256) k_res_rows_bwd(ResBwd a) {
  constexpr int CPL = C / 4;
  constexpr float SQC = C == 16 ? 4.0f : 3.4641016151377544f;  // sqrt(C)
  __shared__ float red[4][4 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  const int b = blockIdx.y, rps = a.rows_per_sample;
  const int cinA = a.cinA, cinB = a.cinB, cin = cinA + cinB, cplA = cinA >> 2, cplB = cinB >> 2;
  const int tile0 = ((int)blockIdx.x * 4 + wv) * 16;  // first row (inside the sample) of this wave's tile
  float dg2[CPL], dg1[CPL], dsc[CPL], dsh[CPL];
#pragma unroll
  for (int r = 0; r < CPL; ++r) dg2[r] = dg1[r] = dsc[r] = dsh[r] = 0.f;

  if (tile0 < rps) {  // (wave-uniform; a wave without rows only takes part in the sums below)
    const int rs = tile0 + i;
    const bool live = rs < rps;
    const int64_t row = (int64_t)b * rps + (live ? rs : 0);
    // ---- A operands.  Output row i of a product = channel cpl * (i / 4) + (i % 4) of the product's output set (none if i % 4 >= cpl);
    //      K-slot g of step (r, k) = d U channel CPL g + r, tap k of the TRANSPOSED convolution (reads position p + 1 - k)
    const int ig = i >> 2, ir = i & 3;
    const bool v2 = ir < CPL, vA = ir < cplA, vB = ir < cplB;
    const int c2 = CPL * ig + ir, cA = cplA * ig + ir, cB = cinA + cplB * ig + ir;
    float w2t[CPL][3], w1a[CPL][3], w1b[CPL][3], wra[CPL], wrb[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
      const int co = CPL * g + r;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        w2t[r][k] = v2 ? a.w2[(co * C + c2) * 3 + k] : 0.f;
        w1a[r][k] = vA ? a.w1[(co * cin + cA) * 3 + k] : 0.f;
        w1b[r][k] = vB ? a.w1[(co * cin + cB) * 3 + k] : 0.f;
      }
      wra[r] = (WR && vA) ? a.wr[co * cin + cA] : 0.f;
      wrb[r] = (WR && vB) ? a.wr[co * cin + cB] : 0.f;
    }
    float g2[CPL], g1[CPL], sc[CPL], sh[CPL];
    {
      const float* ss = a.ss + (int64_t)b * a.ss_stride;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        const int c = CPL * g + r;
        g2[r] = a.g2[c]; g1[r] = a.g1[c]; sc[r] = ss[c] + 1.0f; sh[r] = ss[C + c];
      }
    }
    // ---- this lane's operands: d out, u2, u1 of its CPL channels at the row's N positions
    float dout[CPL][N], d[CPL][N], u1[CPL][N];
    const int64_t base = (row * C + CPL * g) * N;
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
      ld_n<N>(a.dout + base + r * N, dout[r]);
      ld_n<N>(a.u2 + base + r * N, d[r]);  // (u2 for now)
      ld_n<N>(a.u1 + base + r * N, u1[r]);
    }
    // ---- block2: dU2 = (RMSNorm g2 -> SiLU)^T d out, per position (k_res_common.h: norm_act_bwd, the channel sums over the lane groups)
#pragma unroll
    for (int p = 0; p < N; ++p) {
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        dout[r][p] = live ? dout[r][p] : 0.f;
        d[r][p] = live ? d[r][p] : 1.f;
        ssq = fmaf(d[r][p], d[r][p], ssq);
      }
      const float nrm = fast_sqrt(gsum4(ssq)), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
      float uh[CPL];
      float dot = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        uh[r] = d[r][p] * inv;
        const float z = uh[r] * g2[r] * SQC;
        const float dz = dout[r][p] * silu_grad_f(z);
        dg2[r] = fmaf(dz, uh[r] * SQC, dg2[r]);
        d[r][p] = dz * g2[r] * SQC;
        dot = fmaf(d[r][p], uh[r], dot);
      }
      dot = gsum4(dot);
      const bool clamped = nrm < RMS_EPS;
#pragma unroll
      for (int r = 0; r < CPL; ++r) d[r][p] = clamped ? d[r][p] * inv : inv * (d[r][p] - uh[r] * dot);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < CPL; ++r) st_n<N>(a.du2 + base + r * N, d[r]);
    }
    // ---- d a1 = W2^T * dU2 ; block1: dU1 = (RMSNorm g1 -> (scale + 1, shift) -> SiLU)^T d a1
    // (the positions' accumulation chains are interleaved: consecutive MFMAs are independent -- a dependent 16x16x4 waits ~4x its issue time)
    f32x4 da[N];
#pragma unroll
    for (int p = 0; p < N; ++p) da[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < CPL; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int p = 0; p < N; ++p) {
          const int q = p + 1 - k;
          if (q >= 0 && q < N) da[p] = mfma16(w2t[r][k], d[r][q], da[p]);
        }
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const f32x4 acc = da[p];  // (register r = channel CPL g + r, r < CPL)
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        u1[r][p] = live ? u1[r][p] : 1.f;
        ssq = fmaf(u1[r][p], u1[r][p], ssq);
      }
      const float nrm = fast_sqrt(gsum4(ssq)), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
      float uh[CPL];
      float dot = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        uh[r] = u1[r][p] * inv;
        const float z = uh[r] * g1[r] * SQC;
        const float w = fmaf(z, sc[r], sh[r]);
        const float dw = (live ? acc[r] : 0.f) * silu_grad_f(w);
        dsh[r] += dw;
        dsc[r] = fmaf(dw, z, dsc[r]);
        const float dz = dw * sc[r];
        dg1[r] = fmaf(dz, uh[r] * SQC, dg1[r]);
        u1[r][p] = dz * g1[r] * SQC;  // (u1 becomes dU1)
        dot = fmaf(u1[r][p], uh[r], dot);
      }
      dot = gsum4(dot);
      const bool clamped = nrm < RMS_EPS;
#pragma unroll
      for (int r = 0; r < CPL; ++r) u1[r][p] = clamped ? u1[r][p] * inv : inv * (u1[r][p] - uh[r] * dot);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < CPL; ++r) st_n<N>(a.du1 + base + r * N, u1[r]);
    }
    // ---- d x = W1^T * dU1 + (Wr^T d out | d out), for cat(A, B): A's channels, then B's
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float* dst = pass == 0 ? a.dA : a.dB;
      const int cpart = pass == 0 ? cinA : cinB, cpl = pass == 0 ? cplA : cplB;
      if (!dst || cpart == 0) continue;  // (uniform)
      const bool store = pass == 0 ? a.dA_store != 0 : a.dB_store != 0;
      float dx[4][N];
      {
        f32x4 acc[N];
#pragma unroll
        for (int p = 0; p < N; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int p = 0; p < N; ++p) {
              const int q = p + 1 - k;
              if (q >= 0 && q < N) acc[p] = mfma16(pass == 0 ? w1a[r][k] : w1b[r][k], u1[r][q], acc[p]);
            }
          if constexpr (WR) {
#pragma unroll
            for (int p = 0; p < N; ++p) acc[p] = mfma16(pass == 0 ? wra[r] : wrb[r], dout[r][p], acc[p]);
          }
        }
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
          for (int r = 0; r < 4; ++r) dx[r][p] = acc[p][r];
      }
      if constexpr (!WR) {  // identity residual: the input IS the C-channel set of this layout (cinA == C, no B)
#pragma unroll
        for (int r = 0; r < CPL; ++r)
#pragma unroll
          for (int p = 0; p < N; ++p) dx[r][p] += dout[r][p];
      }
      if (live) {
        float* o = dst + (row * cpart + cpl * g) * N;
        float oldv[4][N];
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // (all reads before the first store)
          if (r < cpl && !store) ld_n<N>(o + r * N, oldv[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r < cpl) {
            if (!store) {
#pragma unroll
              for (int p = 0; p < N; ++p) dx[r][p] += oldv[r][p];
            }
            st_n<N>(o + r * N, dx[r]);
          }
        }
      }
    }
  }
  // ---- this workgroup's [d g2 | d g1 | d scale | d shift]: over a tile's 16 rows (DPP), then over the four waves (fixed order)
#pragma unroll
  for (int r = 0; r < CPL; ++r) {
    const float s0 = row_sum16(dg2[r]), s1 = row_sum16(dg1[r]), s2 = row_sum16(dsc[r]), s3 = row_sum16(dsh[r]);
    if (i == 0) {
      const int c = CPL * g + r;
      red[wv][c] = s0; red[wv][16 + c] = s1; red[wv][32 + c] = s2; red[wv][48 + c] = s3;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x & 15, what = threadIdx.x >> 4;
    if (c < C)
      a.gpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * C) + what * C + c] =
          (red[0][what * 16 + c] + red[1][what * 16 + c]) + (red[2][what * 16 + c] + red[3][what * 16 + c]);
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int device_cus() {
  static const int v = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus;
  }();
  return v;
}

}  // namespace

bool res_rows_bwd_usable(const ResBwd& a) {
  if (DQ_DEV_FLAG("DQ_NO_RES_ROWS", '1')) return false;  // (dev switch)
  if (!(a.C == 12 || a.C == 16) || !(a.n == 2 || a.n == 4 || a.n == 8) || a.rows_per_sample < 2) return false;
  // One wave does a 16-row tile here where k_res_cp.hip spreads it over four (lane = channel): 4x fewer wave-instructions, but a 1.2 - 1.5x
  // longer chain per tile -- the better form from about one tile per CU on (measured at batch 1 / 4 / 32: 75 -> 89, -> +14, 149 -> 106 us per step)
  if (a.rows < 16 * device_cus()) return false;
  if (a.cinA < 4 || a.cinA > 16 || (a.cinA & 3) || a.cinB < 0 || a.cinB > 16 || (a.cinB & 3)) return false;
  if (!a.wr && !(a.cinA == a.C && a.cinB == 0)) return false;
  // 16-byte accesses of a row's positions (rows of 2 positions: 8-byte)
  return aligned16(a.dout) && aligned16(a.u1) && aligned16(a.u2) && aligned16(a.du1) && aligned16(a.du2) && (!a.dA || aligned16(a.dA)) &&
         (!a.dB || aligned16(a.dB));
}

int launch_res_rows_bwd(const ResBwd& a, hipStream_t s) {
  DQ_REQUIRE(res_rows_bwd_usable(a) && a.dout && a.u1 && a.u2 && a.du1 && a.du2 && a.ss && a.w1 && a.w2 && a.g1 && a.g2, "res_rows_bwd: unsupported shape or missing operand");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0, "res_rows_bwd: rows must be a multiple of rows_per_sample");
  const int B = a.rows / a.rows_per_sample;
  const dim3 grid(cdiv(a.rows_per_sample, 64), B);
  DQ_REQUIRE(a.gpart && a.gblocks && a.gpart_floats >= (int64_t)grid.x * grid.y * 4 * a.C, "res_rows_bwd: partial-sum slot missing or too small");
  *a.gblocks = (int)grid.x;  // workgroups per sample
#define DQ_RR(CC, NN)                                                                                        \
  if (a.C == CC && a.n == NN) {                                                                              \
    if (a.wr) hipLaunchKernelGGL((k_res_rows_bwd<CC, NN, true>), grid, dim3(256), 0, s, a);                  \
    else hipLaunchKernelGGL((k_res_rows_bwd<CC, NN, false>), grid, dim3(256), 0, s, a);                      \
    DQ_LAUNCH_CHECK();                                                                                       \
    return 0;                                                                                                \
  }
  DQ_RR(12, 2) DQ_RR(12, 4) DQ_RR(12, 8) DQ_RR(16, 2) DQ_RR(16, 4) DQ_RR(16, 8)
#undef DQ_RR
  set_error("res_rows_bwd: unsupported (C, n)");
  return 2;
}

}  // namespace dq
