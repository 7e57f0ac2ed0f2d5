// Fused ResnetBlock FORWARD with both convolutions (and the 1x1 residual conv) on the matrix pipe (reference
// dquartic/model/unet1d.py:271-323): conv3 -> RMSNorm -> (scale+1, shift) -> SiLU -> conv3 -> RMSNorm -> SiLU, + res_conv(x) | x.
//
// Why: with a thread per position and the channels in registers, every FMA of a conv needs its weight as a wave-uniform operand.
// The 200..1,700 weights of a block fit neither the scalar registers nor (unrolled) the vector registers, so k_res_fwd / _v4 / _cp
// read them from LDS as broadcasts -- about one LDS instruction per FMA instruction -- and run at 0.2-0.3 of the HBM rate they are
// meant to stream at.  v_mfma_f32_4x4x1 turns the roles around: of its 16 independent (4 x 1)(1 x 4) outer products, block blk =
// lane / 4 takes  A[i] = W[4 g + i][c][k]  from lane (blk, i) and  B[j] = x[c][position of lane (blk, j) + k - 1]  from lane
// (blk, j) and adds  A[i] B[j]  to register i of lane (blk, j).  With lane = position that is: the B operand IS the register that
// holds input channel c (shifted by one lane for the outer taps: two DPP wave shifts per channel), the result lands as "output
// channel 4 g + i of this lane's position in register i" -- the layout the norm / activation code wants -- and the weight
// operand is one register per (g, c, k) job whose value depends only on lane & 3: an LDS image [job / 4][lane & 3][job % 4], read
// 16 bytes (four jobs) at a time.  One LDS instruction per four MFMAs (1,024 multiply-adds each) instead of one per 64-lane FMA;
// the VALU is left with the norms and activations.  Same MAC rate as unpacked VALU FMAs, on the otherwise idle pipe, bit-for-bit
// an fp32 FMA chain per output (taps accumulate in three chains per channel quad and are added at the end).
//
// A wave owns 64 consecutive positions of one sample = 64 / n whole rows (n = 1 .. 64, a power of two); nothing crosses a wave, so
// there is no barrier after the weight image is staged; a workgroup's waves walk `tpw` tiles each with the image staged once.
#include "dq_common.h"
#include "dq_kernels.h"
#include <algorithm>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// value of lane - 1 / lane + 1 (0 beyond the wave's ends)
__device__ __forceinline__ float lane_m1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_p1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true)); }

constexpr int pad4(int x) { return (x + 3) / 4 * 4; }

}  // namespace

// CBQ: channel quads of the second input (cat(A, B), up path; then the residual is a 1x1 conv); 0: single input, identity residual
template <int C, int CBQ>
__global__ void __launch_bounds__(256) k_res_fwd_mm(ResFwd a, int tiles_ps, int total_tiles) {
  constexpr int G = C / 4, CB = 4 * CBQ, CIN = C + CB;
  constexpr bool WR = CBQ > 0;
  constexpr int J1 = G * CIN * 3, J2 = G * C * 3, JR = WR ? G * CIN : 0;   // jobs (one weight operand each)
  constexpr int O2 = pad4(J1), OR = O2 + pad4(J2), JT = OR + pad4(JR);
  __shared__ __attribute__((aligned(16))) float wl[JT * 4];  // [job / 4][lane & 3][job % 4]
  for (int idx = threadIdx.x; idx < JT * 4; idx += 256) {
    const int j = (idx >> 4) * 4 + (idx & 3), li = (idx >> 2) & 3;
    float v = 0.f;
    if (j < J1) {                       // conv1: job = (c * G + g) * 3 + k
      const int k = j % 3, g = (j / 3) % G, c = j / (3 * G);
      v = a.w1[((4 * g + li) * CIN + c) * 3 + k];
    } else if (j >= O2 && j < O2 + J2) {  // conv2
      const int jj = j - O2, k = jj % 3, g = (jj / 3) % G, c = jj / (3 * G);
      v = a.w2[((4 * g + li) * C + c) * 3 + k];
    } else if (WR && j >= OR && j < OR + JR) {  // res_conv: job = c * G + g
      const int jj = j - OR, g = jj % G, c = jj / G;
      v = a.wr[(4 * g + li) * CIN + c];
    }
    wl[idx] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, li = lane & 3;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int n = a.n;
  const int per_sample = a.rows_per_sample * n;
  const float sqC = sqrtf((float)C);
  const float* wlane = wl + li * 4;
  auto wop = [&](int j) -> float { return wlane[(j >> 2) * 16 + (j & 3)]; };  // (the compiler merges the four jobs of a group into one 16-byte read)

  // the grid is ONE resident round; wave w takes the 64-position tiles w, w + nwaves, ... of the (sample, tile) list
#pragma unroll 1
  for (int gt = wid; gt < total_tiles; gt += nwaves) {
    const int b = gt / tiles_ps, tile = gt - b * tiles_ps;  // wave-uniform
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
    const int it = tile * 64 + lane;
    const bool live = it < per_sample;
    const int rr = live ? it / n : 0, p = live ? it - rr * n : 0;
    const int row = b * a.rows_per_sample + rr;
    const bool hasL = p > 0, hasR = p + 1 < n;
    const int64_t obase = ((int64_t)row * C) * n + p;
    float x[CIN];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = live ? a.inA[obase + (int64_t)c * n] : 0.f;
    if constexpr (WR) {
#pragma unroll
      for (int c = 0; c < CB; ++c) x[C + c] = live ? a.inB[((int64_t)row * CB + c) * n + p] : 0.f;
    }
    // ---- conv1 (k3, zero padding) over cat(A, B)
    f32x4 acc[G][3];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float tm = lane_m1(x[c]), tp = lane_p1(x[c]);  // (every lane takes part in the shifts: no control flow around them)
      const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int j = (c * G + g) * 3;
        acc[g][0] = mfma4(wop(j + 0), xm, acc[g][0]);
        acc[g][1] = mfma4(wop(j + 1), x[c], acc[g][1]);
        acc[g][2] = mfma4(wop(j + 2), xp, acc[g][2]);
      }
    }
    float u[C];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) u[4 * g + i] = (acc[g][0][i] + acc[g][1][i]) + (acc[g][2][i] + a.b1[4 * g + i]);
    if (a.u1 && live) {
#pragma unroll
      for (int c = 0; c < C; ++c) a.u1[obase + (int64_t)c * n] = u[c];
    }
    {
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) ssq = fmaf(u[c], u[c], ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int c = 0; c < C; ++c) u[c] = live ? silu_f(fmaf(u[c] * inv * a.g1[c], ss[c] + 1.0f, ss[C + c])) : 0.f;
    }
    if (a.a1 && live) {
#pragma unroll
      for (int c = 0; c < C; ++c) a.a1[obase + (int64_t)c * n] = u[c];
    }
    // ---- conv2 (k3) over the block-1 activation
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float tm = lane_m1(u[c]), tp = lane_p1(u[c]);
      const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int j = O2 + (c * G + g) * 3;
        acc[g][0] = mfma4(wop(j + 0), xm, acc[g][0]);
        acc[g][1] = mfma4(wop(j + 1), u[c], acc[g][1]);
        acc[g][2] = mfma4(wop(j + 2), xp, acc[g][2]);
      }
    }
    float o[C];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[4 * g + i] = (acc[g][0][i] + acc[g][1][i]) + (acc[g][2][i] + a.b2[4 * g + i]);
    if (a.u2 && live) {
#pragma unroll
      for (int c = 0; c < C; ++c) a.u2[obase + (int64_t)c * n] = o[c];
    }
    {
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) ssq = fmaf(o[c], o[c], ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = silu_f(o[c] * inv * a.g2[c]);
    }
    // ---- residual: 1x1 conv over cat(A, B) (two accumulation chains per channel quad) or identity
    if constexpr (WR) {
      f32x4 ar[G][2];
#pragma unroll
      for (int g = 0; g < G; ++g) { ar[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ar[g][1] = ar[g][0]; }
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int g = 0; g < G; ++g) ar[g][c & 1] = mfma4(wop(OR + c * G + g), x[c], ar[g][c & 1]);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * g + i] += (ar[g][0][i] + ar[g][1][i]) + a.br[4 * g + i];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] += x[c];
    }
    if (live) {
#pragma unroll
      for (int c = 0; c < C; ++c) a.out[obase + (int64_t)c * n] = o[c];
    }
  }
}

bool res_mm_usable(int n, int C, int cinA, int cinB, int rows_per_sample, bool has_wr) {
  return (C == 4 || C == 8 || C == 12 || C == 16) && n >= 1 && n <= 64 && (n & (n - 1)) == 0 && rows_per_sample > 1 && cinA == C &&
         cinB % 4 == 0 && cinB <= C && has_wr == (cinB > 0);
}

int launch_res_fwd_mm(const ResFwd& a, hipStream_t s) {
  DQ_REQUIRE(res_mm_usable(a.n, a.C, a.cinA, a.cinB, a.rows_per_sample, a.wr != nullptr), "res_fwd_mm: unsupported shape");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0, "res_fwd_mm: rows must be a multiple of rows_per_sample");
  const int B = a.rows / a.rows_per_sample;
  const int tiles_ps = cdiv((int64_t)a.rows_per_sample * a.n, 64);
  const int64_t total = (int64_t)tiles_ps * B;
  DQ_REQUIRE(total < (1ll << 31), "res_fwd_mm: too many tiles");
  int dev = 0;
  hipDeviceProp_t pr;
  static const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
  // one resident round (blocks per CU from the occupancy query, at most 7: the scalar-register cap of MI355X_MICROARCH.md, Residency)
#define DQ_MM(CC, QQ)                                                                                          \
  if (a.C == CC && a.cinB == 4 * QQ) {                                                                         \
    static int occ = 0;                                                                                        \
    if (!occ) {                                                                                                \
      int nb = 0;                                                                                              \
      DQ_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_res_fwd_mm<CC, QQ>, 256, 0));              \
      occ = std::max(1, std::min(nb, 7));                                                                      \
    }                                                                                                          \
    const int nblk = (int)std::min<int64_t>((int64_t)occ * cus, (total + 3) / 4);                              \
    hipLaunchKernelGGL((k_res_fwd_mm<CC, QQ>), dim3(nblk), dim3(256), 0, s, a, tiles_ps, (int)total);          \
    DQ_LAUNCH_CHECK();                                                                                         \
    return 0;                                                                                                  \
  }
  DQ_MM(4, 0) DQ_MM(4, 1) DQ_MM(8, 0) DQ_MM(8, 1) DQ_MM(8, 2) DQ_MM(12, 0) DQ_MM(12, 1) DQ_MM(12, 2) DQ_MM(12, 3)
  DQ_MM(16, 0) DQ_MM(16, 1) DQ_MM(16, 2) DQ_MM(16, 3) DQ_MM(16, 4)
#undef DQ_MM
  set_error("res_fwd_mm: unsupported (C, cinB)");
  return 2;
}

}  // namespace dq
