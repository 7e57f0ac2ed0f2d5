// Fused ResnetBlock kernels for the m/z levels (reference dquartic/model/unet1d.py:271-323):
//   forward : conv3 -> RMSNorm -> (scale+1, shift) -> SiLU -> conv3 -> RMSNorm -> SiLU, + res_conv(x) | x      (one launch)
//   backward: the whole data path  d out -> dU2 -> d a1 -> dU1 -> d x  incl. the residual branch, d g1/g2 and the per-sample
//             d(scale)/d(shift)                                                                                 (one launch)
// instead of 2 and 5-6 launches of the generic kernels in k_conv.hip.  Thread = (row, position), all channels in
// registers; the +-1 neighbours a k=3 conv needs of an intermediate (a1 forward; dU2, dU1 backward) are exchanged through
// LDS inside the block.  A block covers 256 consecutive positions of ONE sample (grid = (blocks per sample, B)); rows never
// straddle blocks because the row length divides 256 -- or, at the bottleneck, because a sample IS one row (its RT axis, up to 512
// positions) and gets one block of 256 / 512 threads; the launcher refuses anything else (longer RT axes keep the unfused path).
// Weight gradients stay in k_conv_wgrad (they read dU1 / dU2 written here).
#include "dq_common.h"
#include "dq_kernels.h"
#include "k_res_common.h"
#include <cstdlib>

namespace dq {

namespace {
__device__ __forceinline__ float ldg(const float* p) { return *p; }
// Ordering point of the LDS neighbour exchange.  When the row length divides 64, a wave's 64 consecutive items are whole
// rows: the +-1 neighbours a lane reads were written by its own wave (reads across a row boundary are masked), so a
// wave-level fence is enough and the four waves of the block never wait for each other.
__device__ __forceinline__ void exch_sync(bool wave_local) {
  if (wave_local) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}
}

// The block's three weight tensors -> LDS (plain copies) with EVERY load of a thread requested before its first store: as three loops
// striding by blockDim.x (not unrollable) this was 4-11 memory round trips in a row per workgroup.
template <int C, int BS>
__device__ __forceinline__ void stage_res_weights(float* w2s, float* w1s, float* wrs, const float* __restrict__ w2, const float* __restrict__ w1,
                                                  const float* __restrict__ wr, int cin) {
  constexpr int N2 = (C * C * 3 + BS - 1) / BS, N1 = (C * 2 * C * 3 + BS - 1) / BS, NR = (C * 2 * C + BS - 1) / BS;
  float v2[N2], v1[N1], vr[NR];
  const int n1 = C * cin * 3, nr = wr ? C * cin : 0;
  const float* wrp = wr ? wr : w2;  // (a valid address for the unused loads)
#pragma unroll
  for (int u = 0; u < N2; ++u) { const int i = u * BS + (int)threadIdx.x; v2[u] = w2[i < C * C * 3 ? i : 0]; }
#pragma unroll
  for (int u = 0; u < N1; ++u) { const int i = u * BS + (int)threadIdx.x; v1[u] = w1[i < n1 ? i : 0]; }
#pragma unroll
  for (int u = 0; u < NR; ++u) { const int i = u * BS + (int)threadIdx.x; vr[u] = wrp[i < nr ? i : 0]; }
#pragma unroll
  for (int u = 0; u < N2; ++u) { const int i = u * BS + (int)threadIdx.x; if (i < C * C * 3) w2s[i] = v2[u]; }
#pragma unroll
  for (int u = 0; u < N1; ++u) { const int i = u * BS + (int)threadIdx.x; if (i < n1) w1s[i] = v1[u]; }
#pragma unroll
  for (int u = 0; u < NR; ++u) { const int i = u * BS + (int)threadIdx.x; if (i < nr) wrs[i] = vr[u]; }
}

// HB: the block has skip channels (cat(A, B)).  Without them (the bottleneck's blocks, the down path) the B-side registers do not exist:
// <16, 512> held 255 registers + 60 spilled ones with them (two waves per SIMD at 512 threads).
template <int C, int BS, bool HB>
__global__ void __launch_bounds__(BS) k_res_fwd(ResFwd a) {
  __shared__ float sh[C][BS + 2];
  // the block's weights in LDS (broadcast reads): as scalar loads from memory inside the channel loops nothing overlapped their latency
  // -- 33 us for the bottleneck's 16-channel block, one workgroup per sample
  __shared__ float w2s[C * C * 3], w1s[C * 2 * C * 3], wrs[C * 2 * C];
  const int cin = a.cinA + a.cinB;  // <= 2 C (checked by the launcher)
  stage_res_weights<C, BS>(w2s, w1s, wrs, a.w2, a.w1, a.wr, cin);
  __syncthreads();
  const int b = blockIdx.y;
  const int per_sample = a.rows_per_sample * a.n;
  const bool wave_local = a.n <= 64 && (64 % a.n) == 0;
  const int it = blockIdx.x * BS + threadIdx.x;
  const bool live = it < per_sample;
  const int row = b * a.rows_per_sample + (live ? it / a.n : 0), p = live ? it % a.n : 0;
  const float sqC = sqrtf((float)C);
  float acc[C];
  // ---- conv1 (k3, zero padding) over cat(A, B)
#pragma unroll
  for (int co = 0; co < C; ++co) acc[co] = a.b1[co];
  // cat(A, B) has cinA == C channels from A (launcher-checked) and cinB <= C from B: every load of the thread is issued
  // up front from compile-time-unrolled loops -- a runtime ci loop serialised one exposed global latency per channel
  float xa[C][3], xb[HB ? C : 1][3];
#pragma unroll
  for (int ci = 0; ci < C; ++ci) {
    const float* src = a.inA + ((int64_t)row * C + ci) * a.n;
    xa[ci][0] = (live && p > 0) ? src[p - 1] : 0.f;
    xa[ci][1] = live ? src[p] : 0.f;
    xa[ci][2] = (live && p + 1 < a.n) ? src[p + 1] : 0.f;
  }
#pragma unroll
  for (int ci = 0; ci < (HB ? C : 0); ++ci) {
    const bool ok = live && ci < a.cinB;
    const float* src = a.inB + ((int64_t)row * a.cinB + ci) * a.n;
    xb[ci][0] = (ok && p > 0) ? src[p - 1] : 0.f;
    xb[ci][1] = ok ? src[p] : 0.f;
    xb[ci][2] = (ok && p + 1 < a.n) ? src[p + 1] : 0.f;
  }
#pragma unroll
  for (int ci = 0; ci < C; ++ci)
#pragma unroll
    for (int co = 0; co < C; ++co) {
      const float* w = w1s + (co * cin + ci) * 3;
      acc[co] = fmaf(w[0], xa[ci][0], fmaf(w[1], xa[ci][1], fmaf(w[2], xa[ci][2], acc[co])));
    }
  if (HB && a.cinB) {
#pragma unroll
    for (int ci = 0; ci < (HB ? C : 0); ++ci) {
      if (ci < a.cinB) {
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float* w = w1s + (co * cin + C + ci) * 3;
          acc[co] = fmaf(w[0], xb[ci][0], fmaf(w[1], xb[ci][1], fmaf(w[2], xb[ci][2], acc[co])));
        }
      }
    }
  }
  const int64_t obase = ((int64_t)row * C) * a.n + p;
  if (live && a.u1) {
#pragma unroll
    for (int co = 0; co < C; ++co) a.u1[obase + (int64_t)co * a.n] = acc[co];
  }
  {
    float ssq = 0.f;
#pragma unroll
    for (int co = 0; co < C; ++co) ssq = fmaf(acc[co], acc[co], ssq);
    const float inv = rms_inv(ssq, sqC);
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
#pragma unroll
    for (int co = 0; co < C; ++co) acc[co] = silu_f(fmaf(acc[co] * inv * a.g1[co], ss[co] + 1.0f, ss[C + co]));
  }
  if (live && a.a1) {
#pragma unroll
    for (int co = 0; co < C; ++co) a.a1[obase + (int64_t)co * a.n] = acc[co];
  }
  // ---- neighbours of a1 through LDS (zero outside the row)
#pragma unroll
  for (int co = 0; co < C; ++co) sh[co][threadIdx.x + 1] = live ? acc[co] : 0.f;
  exch_sync(wave_local);
  float o[C];
#pragma unroll
  for (int co = 0; co < C; ++co) o[co] = a.b2[co];
  const bool hasL = p > 0, hasR = p + 1 < a.n;
#pragma unroll 1
  for (int ci = 0; ci < C; ++ci) {  // not unrolled: C*C*3 weights would not fit in registers
    const float x0 = hasL ? sh[ci][threadIdx.x] : 0.f, x1 = sh[ci][threadIdx.x + 1], x2 = hasR ? sh[ci][threadIdx.x + 2] : 0.f;
    const float* w = w2s + ci * 3;
#pragma unroll
    for (int co = 0; co < C; ++co) o[co] = fmaf(w[co * C * 3 + 0], x0, fmaf(w[co * C * 3 + 1], x1, fmaf(w[co * C * 3 + 2], x2, o[co])));
  }
  if (!live) return;
  if (a.u2) {
#pragma unroll
    for (int co = 0; co < C; ++co) a.u2[obase + (int64_t)co * a.n] = o[co];
  }
  {
    float ssq = 0.f;
#pragma unroll
    for (int co = 0; co < C; ++co) ssq = fmaf(o[co], o[co], ssq);
    const float inv = rms_inv(ssq, sqC);
#pragma unroll
    for (int co = 0; co < C; ++co) o[co] = silu_f(o[co] * inv * a.g2[co]);
  }
  // ---- residual: 1x1 conv over cat(A, B) or identity
  if (a.wr) {
#pragma unroll
    for (int co = 0; co < C; ++co) o[co] += a.br[co];
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
      for (int co = 0; co < C; ++co) o[co] = fmaf(wrs[co * cin + ci], xa[ci][1], o[co]);
#pragma unroll
    for (int ci = 0; ci < (HB ? C : 0); ++ci) {
      if (ci < a.cinB) {
#pragma unroll
        for (int co = 0; co < C; ++co) o[co] = fmaf(wrs[co * cin + C + ci], xb[ci][1], o[co]);
      }
    }
  } else {
#pragma unroll
    for (int co = 0; co < C; ++co) o[co] += xa[co][1];
  }
#pragma unroll
  for (int co = 0; co < C; ++co) a.out[obase + (int64_t)co * a.n] = o[co];
}

// m/z levels: many rows per sample, the row length divides the block's 256 positions.  Bottleneck: ONE row per sample (its RT axis) of
// up to 512 positions = one block per sample, whatever the length (no row can straddle a block then).
bool res_fusable(int n, int C, int rows_per_sample) {
  if (!(C == 4 || C == 8 || C == 12 || C == 16) || n < 1) return false;
  return rows_per_sample == 1 ? (n <= 512 || C == 16) : (n <= 256 && (256 % n) == 0);  // (C == 16: k_res_rt.hip takes any RT length)
}
namespace {
int res_block_size(int n, int rows_per_sample) { return rows_per_sample == 1 && n > 256 ? 512 : 256; }
}

int launch_res_fwd(const ResFwd& a, hipStream_t s) {
  DQ_REQUIRE(res_fusable(a.n, a.C, a.rows_per_sample), "res_fwd: row length must divide 256 (or one row of <= 512 per sample) and C be 4/8/12/16");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0, "res_fwd: rows must be a multiple of rows_per_sample");
  DQ_REQUIRE(a.wr || (a.cinA == a.C && a.cinB == 0), "res_fwd: identity residual needs C input channels");
  if (res_rt_usable(a.C, a.cinA, a.cinB, a.wr != nullptr, a.rows_per_sample)) return launch_res_rt_fwd(a, s);  // the bottleneck's blocks
  // m/z rows of up to 64 positions: the convolutions on the matrix pipe (k_res_mm.hip)
  if (res_mm_usable(a.n, a.C, a.cinA, a.cinB, a.rows_per_sample, a.wr != nullptr)) return launch_res_fwd_mm(a, s);
  if (a.rows_per_sample > 1 && res_cp_usable(a.n, a.C, a.cinA, a.cinB)) return launch_res_fwd_cp(a, s);
  if (a.rows_per_sample > 1 && res_v4_usable(a.n, a.C, a.cinA, a.cinB)) return launch_res_fwd_v4(a, s);
  DQ_REQUIRE(a.cinA == a.C && a.cinB <= a.C && (a.cinB == 0 || a.inB), "res_fwd: input must be C channels (+ at most C skip channels)");
  DQ_REQUIRE(a.rows_per_sample > 1 || a.n <= 512, "res_fwd: one RT row per sample longer than 512 positions needs the 16-channel identity block");
  const int B = a.rows / a.rows_per_sample;
  const int BS = res_block_size(a.n, a.rows_per_sample);
  dim3 grid(cdiv((int64_t)a.rows_per_sample * a.n, BS), B), block(BS);
#define DQ_RF(CC)                                                                \
  case CC:                                                                       \
    if (BS == 512 && a.cinB) hipLaunchKernelGGL((k_res_fwd<CC, 512, true>), grid, block, 0, s, a);  \
    else if (BS == 512) hipLaunchKernelGGL((k_res_fwd<CC, 512, false>), grid, block, 0, s, a);      \
    else if (a.cinB) hipLaunchKernelGGL((k_res_fwd<CC, 256, true>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((k_res_fwd<CC, 256, false>), grid, block, 0, s, a);                     \
    break;
  switch (a.C) { DQ_RF(4) DQ_RF(8) DQ_RF(12) DQ_RF(16) }
#undef DQ_RF
  DQ_LAUNCH_CHECK();
  return 0;
}

// -----------------------------------------------------------------------------------------------------------------
// backward data path
// -----------------------------------------------------------------------------------------------------------------
template <int C, int BS>
__global__ void __launch_bounds__(BS) k_res_bwd(ResBwd a) {
  __shared__ float sh[C][BS + 2];
  __shared__ float red[BS / 64][4 * C];
  // the block's weights in LDS: read from memory inside the channel loops they are scalar loads that nothing overlaps
  __shared__ float w2s[C * C * 3], w1s[C * 2 * C * 3], wrs[C * 2 * C];
  {
    const int cin_ = a.cinA + a.cinB;  // <= 2 C (checked by the launcher)
    stage_res_weights<C, BS>(w2s, w1s, wrs, a.w2, a.w1, a.wr, cin_);
    __syncthreads();
  }
  const int b = blockIdx.y;
  const int per_sample = a.rows_per_sample * a.n;
  const bool wave_local = a.n <= 64 && (64 % a.n) == 0;
  const int it = blockIdx.x * BS + threadIdx.x;
  const bool live = it < per_sample;
  const int row = b * a.rows_per_sample + (live ? it / a.n : 0), p = live ? it % a.n : 0;
  const int cin = a.cinA + a.cinB;
  const int64_t obase = ((int64_t)row * C) * a.n + p;
  const bool hasL = live && p > 0, hasR = live && p + 1 < a.n;  // dead threads must not read the (unwritten) halo slots
  float dg2[C], dg1[C], dsc[C], dsh[C];
#pragma unroll
  for (int c = 0; c < C; ++c) dg2[c] = dg1[c] = dsc[c] = dsh[c] = 0.f;

  // ---- block2: dU2 = norm/act backward of d out.  Everything this thread will read from global memory (d out, u2, u1 and,
  // for the identity residual, the old dA) is requested here, so one latency is exposed instead of three
  float dout[C], d[C], u[C], u1v[C], dold[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    dout[c] = live ? a.dout[obase + (int64_t)c * a.n] : 0.f;
    u[c] = live ? a.u2[obase + (int64_t)c * a.n] : 1.f;
    u1v[c] = live ? a.u1[obase + (int64_t)c * a.n] : 1.f;
    dold[c] = (live && !a.wr && a.dA && !a.dA_store) ? a.dA[obase + (int64_t)c * a.n] : 0.f;
    d[c] = dout[c];
  }
  norm_act_bwd<C, false>(u, d, a.g2, nullptr, dg2, nullptr, nullptr);
  if (live) {
#pragma unroll
    for (int c = 0; c < C; ++c) a.du2[obase + (int64_t)c * a.n] = d[c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) sh[c][threadIdx.x + 1] = live ? d[c] : 0.f;
  exch_sync(wave_local);
  // ---- d a1[ci][p] = sum_co sum_k W2[co][ci][k] dU2[co][p + 1 - k]
  float da1[C];
#pragma unroll
  for (int ci = 0; ci < C; ++ci) da1[ci] = 0.f;
#pragma unroll 1
  for (int co = 0; co < C; ++co) {
    const float dr = hasR ? sh[co][threadIdx.x + 2] : 0.f, dc = sh[co][threadIdx.x + 1], dl = hasL ? sh[co][threadIdx.x] : 0.f;
    const float* w = w2s + co * C * 3;
#pragma unroll
    for (int ci = 0; ci < C; ++ci) da1[ci] = fmaf(w[ci * 3 + 0], dr, fmaf(w[ci * 3 + 1], dc, fmaf(w[ci * 3 + 2], dl, da1[ci])));
  }
  exch_sync(wave_local);
  // ---- block1: dU1
#pragma unroll
  for (int c = 0; c < C; ++c) u[c] = u1v[c];
  norm_act_bwd<C, true>(u, da1, a.g1, a.ss + (int64_t)b * a.ss_stride, dg1, dsc, dsh);
  if (live) {
#pragma unroll
    for (int c = 0; c < C; ++c) a.du1[obase + (int64_t)c * a.n] = da1[c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) sh[c][threadIdx.x + 1] = live ? da1[c] : 0.f;
  exch_sync(wave_local);
  // ---- d x[ci][p] = sum_co sum_k W1[co][ci][k] dU1[co][p + 1 - k]  (+ residual branch), accumulated into dA / dB
  if (live && (a.dA || a.dB)) {
    if (a.wr) {
      // All input channels of this position at once (cin <= 2 C: a block input is at most the concatenation of two C-channel
      // tensors): the taps of dU1 are read from LDS once per output channel instead of once per (ci, co), d out comes from the
      // registers, and the old values of dA / dB are requested together before the first store -- the former per-channel
      // `*dst += v` was cin serial round trips (a load behind a store the compiler cannot tell apart waits for it).
      // Same per-channel summation order as before.
      constexpr int CM = 2 * C;
      float v[CM];
#pragma unroll
      for (int ci = 0; ci < CM; ++ci) v[ci] = 0.f;
#pragma unroll
      for (int co = 0; co < C; ++co) {
        const float dr = hasR ? sh[co][threadIdx.x + 2] : 0.f, dc = sh[co][threadIdx.x + 1], dl = hasL ? sh[co][threadIdx.x] : 0.f;
        const float dco = dout[co];
#pragma unroll
        for (int ci = 0; ci < CM; ++ci) {
          if (ci < cin) {
            const float* w = w1s + (co * cin + ci) * 3;
            v[ci] = fmaf(w[0], dr, fmaf(w[1], dc, fmaf(w[2], dl, v[ci])));
            v[ci] = fmaf(wrs[co * cin + ci], dco, v[ci]);
          }
        }
      }
      float oldv[CM];
#pragma unroll
      for (int ci = 0; ci < CM; ++ci) {
        float* dst = nullptr;
        if (ci < a.cinA) { if (a.dA) dst = a.dA + ((int64_t)row * a.cinA + ci) * a.n + p; }
        else if (ci < cin) { if (a.dB) dst = a.dB + ((int64_t)row * a.cinB + (ci - a.cinA)) * a.n + p; }
        oldv[ci] = (dst && !(ci < a.cinA ? a.dA_store : a.dB_store)) ? *dst : 0.f;
      }
#pragma unroll
      for (int ci = 0; ci < CM; ++ci) {
        float* dst = nullptr;
        if (ci < a.cinA) { if (a.dA) dst = a.dA + ((int64_t)row * a.cinA + ci) * a.n + p; }
        else if (ci < cin) { if (a.dB) dst = a.dB + ((int64_t)row * a.cinB + (ci - a.cinA)) * a.n + p; }
        if (dst) *dst = oldv[ci] + v[ci];
      }
    } else if (a.dA) {  // identity residual: cin == C, single input
      float dx[C];
#pragma unroll
      for (int ci = 0; ci < C; ++ci) dx[ci] = dout[ci];
#pragma unroll 1
      for (int co = 0; co < C; ++co) {
        const float dr = hasR ? sh[co][threadIdx.x + 2] : 0.f, dc = sh[co][threadIdx.x + 1], dl = hasL ? sh[co][threadIdx.x] : 0.f;
        const float* w = w1s + co * C * 3;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) dx[ci] = fmaf(w[ci * 3 + 0], dr, fmaf(w[ci * 3 + 1], dc, fmaf(w[ci * 3 + 2], dl, dx[ci])));
      }
#pragma unroll
      for (int ci = 0; ci < C; ++ci) a.dA[obase + (int64_t)ci * a.n] = dold[ci] + dx[ci];
    }
  }
  // ---- reductions: dg2, dg1, this sample's d(scale), d(shift)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float s0 = wave_sum(dg2[c]), s1 = wave_sum(dg1[c]), s2 = wave_sum(dsc[c]), s3 = wave_sum(dsh[c]);
    if (lane == 0) { red[wv][c] = s0; red[wv][C + c] = s1; red[wv][2 * C + c] = s2; red[wv][3 * C + c] = s3; }
  }
  __syncthreads();
  // every block leaves its four sums [dg2 | dg1 | dscale | dshift] in its own slot of gpart; launch_part_reduce adds them up in
  // block order (repeatable to the bit: no float atomics)
  for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) {
    float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (BS == 512) v += (red[4][i] + red[5][i]) + (red[6][i] + red[7][i]);
    a.gpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * C) + i] = v;
  }
}

int launch_res_bwd(const ResBwd& a, hipStream_t s) {
  DQ_REQUIRE(res_fusable(a.n, a.C, a.rows_per_sample), "res_bwd: row length must divide 256 (or one row of <= 512 per sample) and C be 4/8/12/16");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0, "res_bwd: rows must be a multiple of rows_per_sample");
  DQ_REQUIRE(a.wr || (a.cinA == a.C && a.cinB == 0), "res_bwd: identity residual needs C input channels");
  DQ_REQUIRE(a.cinA + a.cinB <= 2 * a.C, "res_bwd: a block input wider than two C-channel tensors is not built");
  if (a.gblocks) *a.gblocks = 0;
  if (res_rt_usable(a.C, a.cinA, a.cinB, a.wr != nullptr, a.rows_per_sample)) return launch_res_rt_bwd(a, s);  // the bottleneck's blocks
  DQ_REQUIRE(a.rows_per_sample > 1 || a.n <= 512, "res_bwd: one RT row per sample longer than 512 positions needs the 16-channel identity block");
  if (res_rows_bwd_usable(a)) return launch_res_rows_bwd(a, s);  // the deep levels: m/z row = lane column
  if (a.rows_per_sample > 1 && res_cp_usable(a.n, a.C, a.cinA, a.cinB)) return launch_res_bwd_cp(a, s);
  const int B = a.rows / a.rows_per_sample;
  const int BS = res_block_size(a.n, a.rows_per_sample);
  dim3 grid(cdiv((int64_t)a.rows_per_sample * a.n, BS), B), block(BS);
  ResBwd k = a;
  DQ_REQUIRE(k.gpart && k.gblocks && k.gpart_floats >= (int64_t)grid.x * grid.y * 4 * a.C, "res_bwd: partial-sum slot missing or too small");
  *k.gblocks = (int)grid.x;  // blocks per sample
#define DQ_RB(CC)                                                                \
  case CC:                                                                       \
    if (BS == 512) hipLaunchKernelGGL((k_res_bwd<CC, 512>), grid, block, 0, s, k); \
    else hipLaunchKernelGGL((k_res_bwd<CC, 256>), grid, block, 0, s, k);          \
    break;
  switch (a.C) { DQ_RB(4) DQ_RB(8) DQ_RB(12) DQ_RB(16) }
#undef DQ_RB
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
