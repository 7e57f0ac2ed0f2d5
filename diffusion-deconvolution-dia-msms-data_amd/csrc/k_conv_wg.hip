// Backward of a level's resample conv -- Downsample (k4 s2 p1), Upsample (nearest x2 + k3 p1) or the k3 conv of the last levels
// (reference dquartic/model/unet1d.py:82-110) -- data gradient AND weight / bias gradient in ONE launch, both on the 4x4x1 matrix
// pipe.  Replaces k_conv_bwd_data (main stream) + k_conv_wgrad + k_wgrad_reduce (side stream), which read dY and the conv input
// twice.
//
// Thread = output position q of the conv (a wave = 64 consecutive positions = whole rows, n <= 64):
//   data gradient  : the transposed conv in the lane = position form of k_res_mm.hip (B operand = the register holding dY[co], one DPP
//                    wave shift for the neighbours; A operand = W[co][4 gi + (lane & 3)][k] from an LDS operand image):
//                      DOWN: d in[2 q] = sum_co W[.][.][1] dY[q] + W[.][.][3] dY[q - 1],  d in[2 q + 1] = sum_co W[.][.][2] dY[q] + W[.][.][0] dY[q + 1]
//                            (stored as one 8-byte pair per channel)
//                      UP:   d up[q] = sum_co sum_k W[.][.][k] dY[q + 1 - k],  d in[q / 2] = d up[q] + d up[q ^ 1]  (the even lane stores)
//                      S1:   d in[q] = sum_co sum_k W[.][.][k] dY[q + 1 - k]
//   weight gradient: dW[co][ci][k] = sum_q dY[co][q] tap_k[ci][q] with the taps the forward conv multiplies (DOWN: in[2 q - 1 .. 2 q + 2];
//                    UP / S1: the (upsampled) input at q - 1, q, q + 1) -- dY and the K tap tensors are staged in LDS as [position][channel]
//                    images, and the (co quad, ci quad, tap) jobs are split over the workgroup's four waves, exactly as in k_res_wg.hip.
// A workgroup walks several 256-position tiles of one sample and leaves [dW | dbias] in its slot; k_res_wg_reduce sums the slots in
// block order into the flat gradient buffer (a conv's weight and bias are adjacent there).  No atomics: bitwise repeatable.
#include "dq_common.h"
#include "dq_kernels.h"
#include "k_res_common.h"
#include "dq_probe.h"
#include <algorithm>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float lane_m1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_p1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_x1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); }  // lane ^ 1
constexpr int pad4(int x) { return (x + 3) / 4 * 4; }
constexpr int TILE = 256, RUN = 16;
// [position + 1][channel] image (k_res_wg.hip): floats per position an odd number of 16-byte slots, 4 floats of padding per 16 positions
constexpr int img_st(int c) { return c == 4 ? 4 : (c == 8 ? 12 : 20); }
constexpr int img_floats(int c) { return (TILE + 2) * img_st(c) + ((TILE + 2) / RUN + 1) * 4; }
__device__ __forceinline__ int img_at(int q, int st) { return q * st + (q >> 4) * 4; }


}  // namespace

template <int C, int PRE, int CP>
__global__ void __launch_bounds__(256) k_conv_bwd_wg(ConvBwdWg a) {
  constexpr int K = PRE == LEVEL_PRE_DOWN ? 4 : 3;
  constexpr int G = C / 4, H = CP / 4;
  constexpr int STY = img_st(C), STX = img_st(CP);
  constexpr int JW = (G * H * K + 3) / 4;  // weight-gradient jobs per wave
  // operand image of the transposed conv: job (co, gi, k) = W[co][4 gi + li][k]
  constexpr int JT = pad4(C * H * K);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                          // JT * 4
  float* i_dy = wl + JT * 4;                // img_floats(C)
  float* i_tap = i_dy + img_floats(C);      // K x img_floats(CP)
  DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 0);
  for (int idx = threadIdx.x; idx < JT * 4; idx += 256) {
    const int j = (idx >> 4) * 4 + (idx & 3), l4 = (idx >> 2) & 3;
    float v = 0.f;
    if (j < C * H * K) {
      const int k = j % K, gi = (j / K) % H, co = j / (K * H);
      v = a.w[(co * CP + 4 * gi + l4) * K + k];
    }
    wl[idx] = v;
  }
  DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 1);
  const int b = blockIdx.y, n = a.n;
  const int per_sample = a.rows_per_sample * n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q1 = tid + 1;
  const int blk = lane >> 2, li = lane & 3;
  const int lby = img_at(blk * RUN + 1, STY) + li, lbx = img_at(blk * RUN + 1, STX) + li;
  const float* wlane = wl + li * 4;
  // the transposed conv takes its weight operands as a stream of jobs in image order through a ring of RD 16-byte reads in flight
  // (k_level.hip: left to the compiler one read is in flight, issued ~34 cycles before its use against an LDS latency of 64+)
  constexpr int RD = 3;
  struct WRing { float4 r[RD]; float4 cur; };
  auto ldw = [&](int q) -> float4 { return *reinterpret_cast<const float4*>(wlane + q * 16); };
  auto ring_start = [&](WRing& R) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < RD; ++d) R.r[d] = ldw(d);  // (reads past the last job land in the position images behind the weight image)
  };
  auto wgroup = [&](WRing& R, int q) __attribute__((always_inline)) {  // makes group q (jobs 4 q .. 4 q + 3) current; groups are taken in order
    R.cur = R.r[q % RD];
    R.r[q % RD] = ldw(q + RD);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wcur = [&](const WRing& R, int j) -> float { return (j & 3) == 0 ? R.cur.x : (j & 3) == 1 ? R.cur.y : (j & 3) == 2 ? R.cur.z : R.cur.w; };
  // this wave's weight-gradient jobs j = wv * JW + jj = (g * H + h) * K + k, and the bias sum of channel quad g == wv
  f32x4 aw[JW], ab = {0.f, 0.f, 0.f, 0.f};
  const float* pa[JW]; const float* pb[JW];
  bool jv[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    aw[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int j = wv * JW + jj;
    jv[jj] = j < G * H * K;
    const int jc = jv[jj] ? j : 0;
    const int k = jc % K, h = (jc / K) % H, g = jc / (K * H);
    pa[jj] = i_dy + lby + 4 * g;
    pb[jj] = i_tap + k * img_floats(CP) + lbx + 4 * h;
  }
  const bool do_bias = wv < G;
  const int n_in = PRE == LEVEL_PRE_DOWN ? 2 * n : (PRE == LEVEL_PRE_UP ? n / 2 : n);

  const int tile_end = min(a.tiles_ps, ((int)blockIdx.x + 1) * a.tpb);
#pragma unroll 1
  for (int tile = blockIdx.x * a.tpb; tile < tile_end; ++tile) {
    const int it = tile * TILE + tid;
    const bool live = it < per_sample;
    const int itc = live ? it : per_sample - 1;   // (no load is predicated; the threads beyond the sample contribute dY = 0)
    const int rr = itc / n, p = itc - rr * n;
    const int row = b * a.rows_per_sample + rr;
    const bool hasL = live && p > 0, hasR = live && p + 1 < n;
    const int p_in = PRE == LEVEL_PRE_DOWN ? 2 * p : (PRE == LEVEL_PRE_UP ? p >> 1 : p);
    float dy[C], v0[CP], v1[PRE == LEVEL_PRE_DOWN ? CP : 1];
    // wave-uniform (tensor + channel) bases + one 32-bit byte offset per lane and tensor (the launcher checks the tensors stay below 4 GB)
    const unsigned boY = (((unsigned)row * C) * (unsigned)n + (unsigned)p) * 4u;
    const unsigned boX = (((unsigned)row * CP) * (unsigned)n_in + (unsigned)p_in) * 4u;
#pragma unroll
    for (int c = 0; c < C; ++c) dy[c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dy + (size_t)c * n) + boY);
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const char* s = reinterpret_cast<const char*>(a.in + (size_t)c * n_in) + boX;
      if constexpr (PRE == LEVEL_PRE_DOWN) {
        const float2 v = *reinterpret_cast<const float2*>(s);
        v0[c] = v.x; v1[c] = v.y;
      } else {
        v0[c] = *reinterpret_cast<const float*>(s);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dy[c] = live ? dy[c] : 0.f;
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 2);
    __syncthreads();  // the previous tile's readers of the images are done (first tile: the operand image is visible)
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 3);
    {
      const int oy = img_at(q1, STY), ox = img_at(q1, STX);
#pragma unroll
      for (int g = 0; g < G; ++g)
        *reinterpret_cast<float4*>(i_dy + oy + 4 * g) = make_float4(dy[4 * g], dy[4 * g + 1], dy[4 * g + 2], dy[4 * g + 3]);
      // the K taps the forward conv multiplied at this output position
      float t[K][CP];
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        if constexpr (PRE == LEVEL_PRE_DOWN) {
          const float tm = lane_m1(v1[c]), tp = lane_p1(v0[c]);
          t[0][c] = hasL ? tm : 0.f; t[1][c] = v0[c]; t[2][c] = v1[c]; t[3][c] = hasR ? tp : 0.f;
        } else {
          const float tm = lane_m1(v0[c]), tp = lane_p1(v0[c]);
          t[0][c] = hasL ? tm : 0.f; t[1][c] = v0[c]; t[2][c] = hasR ? tp : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h)
          *reinterpret_cast<float4*>(i_tap + k * img_floats(CP) + ox + 4 * h) = make_float4(t[k][4 * h], t[k][4 * h + 1], t[k][4 * h + 2], t[k][4 * h + 3]);
    }
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 4);
    __syncthreads();
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 5);
    // ---- weight / bias gradient on the matrix pipe: lane block blk walks positions 16 blk .. 16 blk + 15 of the tile
    // (no wave-uniform `if` around a job: a branch per MFMA ends the basic block, every LDS read is then waited for right where it is
    // issued, and the phase ran at one LDS latency per MFMA -- 22,000 clocks for 192 MFMAs at 16 channels.  A wave's surplus jobs
    // recompute job 0 and are not flushed.)  The operands of step s + 1 are read while step s multiplies; the scheduling barrier keeps
    // the scheduler from hoisting all 16 steps' reads (390 registers in some instantiations) or sinking them next to their use.
    {
      float ca[JW], cb[JW], cbias;
      const int bq = lby + 4 * (do_bias ? wv : 0);
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) { ca[jj] = pa[jj][0]; cb[jj] = pb[jj][0]; }
      cbias = i_dy[bq];
#pragma unroll
      for (int s = 0; s < RUN; ++s) {
        float na[JW], nb[JW], nbias = 0.f;
        if (s + 1 < RUN) {
          const int oy = (s + 1) * STY + (s + 1 == RUN - 1 ? 4 : 0), ox = (s + 1) * STX + (s + 1 == RUN - 1 ? 4 : 0);
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) { na[jj] = pa[jj][oy]; nb[jj] = pb[jj][ox]; }
          nbias = i_dy[bq + oy];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) aw[jj] = mfma4(ca[jj], cb[jj], aw[jj]);
        ab = mfma4(cbias, 1.f, ab);
        if (s + 1 < RUN) {
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) { ca[jj] = na[jj]; cb[jj] = nb[jj]; }
          cbias = nbias;
        }
      }
    }
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 6);
    // ---- data gradient (transposed conv, registers + DPP shifts)
    if (a.din) {
      if constexpr (PRE == LEVEL_PRE_DOWN) {
        f32x4 e0[H], e1[H], o0[H], o1[H];
#pragma unroll
        for (int h = 0; h < H; ++h) e0[h] = e1[h] = o0[h] = o1[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        WRing R;
        ring_start(R);
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float tm = lane_m1(dy[co]), tp = lane_p1(dy[co]);
          const float dl = hasL ? tm : 0.f, dr = hasR ? tp : 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const int j = (co * H + h) * 4;
            wgroup(R, j >> 2);
            e0[h] = mfma4(wcur(R, 1), dy[co], e0[h]);
            e1[h] = mfma4(wcur(R, 3), dl, e1[h]);
            o0[h] = mfma4(wcur(R, 2), dy[co], o0[h]);
            o1[h] = mfma4(wcur(R, 0), dr, o1[h]);
          }
        }
        if (live) {
          float2 old[CP];
#pragma unroll
          for (int c = 0; c < CP; ++c) old[c] = make_float2(0.f, 0.f);
          if (a.accumulate) {
#pragma unroll
            for (int c = 0; c < CP; ++c) old[c] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.din + (size_t)c * n_in) + boX);
          }
#pragma unroll
          for (int c = 0; c < CP; ++c) {
            const int h = c >> 2, i = c & 3;
            *reinterpret_cast<float2*>(reinterpret_cast<char*>(a.din + (size_t)c * n_in) + boX) =
                make_float2(old[c].x + (e0[h][i] + e1[h][i]), old[c].y + (o0[h][i] + o1[h][i]));
          }
        }
      } else {
        f32x4 acc[H][3];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[h][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        WRing R;
        ring_start(R);
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float tm = lane_m1(dy[co]), tp = lane_p1(dy[co]);
          const float dl = hasL ? tm : 0.f, dr = hasR ? tp : 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const int j = (co * H + h) * 3;
            if (((j + 0) & 3) == 0) wgroup(R, (j + 0) >> 2);
            acc[h][0] = mfma4(wcur(R, j + 0), dr, acc[h][0]);     // tap k reads dY[q + 1 - k]
            if (((j + 1) & 3) == 0) wgroup(R, (j + 1) >> 2);
            acc[h][1] = mfma4(wcur(R, j + 1), dy[co], acc[h][1]);
            if (((j + 2) & 3) == 0) wgroup(R, (j + 2) >> 2);
            acc[h][2] = mfma4(wcur(R, j + 2), dl, acc[h][2]);
          }
        }
        float v[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) v[c] = (acc[c >> 2][0][c & 3] + acc[c >> 2][1][c & 3]) + acc[c >> 2][2][c & 3];
        if constexpr (PRE == LEVEL_PRE_UP) {  // d in[q / 2] = d up[q] + d up[q ^ 1]
#pragma unroll
          for (int c = 0; c < CP; ++c) v[c] += lane_x1(v[c]);
        }
        if (live && (PRE != LEVEL_PRE_UP || (lane & 1) == 0)) {
          float old[CP];
#pragma unroll
          for (int c = 0; c < CP; ++c) old[c] = 0.f;
          if (a.accumulate) {
#pragma unroll
            for (int c = 0; c < CP; ++c) old[c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.din + (size_t)c * n_in) + boX);
          }
#pragma unroll
          for (int c = 0; c < CP; ++c) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.din + (size_t)c * n_in) + boX) = old[c] + v[c];
        }
      }
    }
    DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 7);
  }  // tile

  // ---- the block's slot [dW (C x CP x K) | dbias (C)]
  float* part = a.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * a.nv;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    if (jv[jj]) {  // wave-uniform
      const int j = wv * JW + jj;
      const int k = j % K, h = (j / K) % H, g = j / (K * H);
      const f32x4 t = blocks_sum(aw[jj]);
      if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part[((4 * g + i) * CP + 4 * h + lane) * K + k] = t[i];
      }
    }
  }
  if (do_bias) {
    const f32x4 t = blocks_sum(ab);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[C * CP * K + 4 * wv + i] = t[i];
    }
  }
  DQ_PSTAMP(200000 + C * 1000 + PRE * 100 + CP, 8);
}

static bool conv_wg_built(int C, int pre, int cp) {
  if (pre == LEVEL_PRE_DOWN) return cp == C || (cp == C - 4 && cp >= 4);
  if (pre == LEVEL_PRE_UP || pre == LEVEL_PRE_S1) return cp == C || (cp == C + 4 && cp <= 16);
  return false;
}
bool conv_wg_usable(int C, int pre, int cp, int n, int rows_per_sample) {
  return (C == 4 || C == 8 || C == 12 || C == 16) && conv_wg_built(C, pre, cp) && n >= 1 && n <= 64 && (n & (n - 1)) == 0 && rows_per_sample > 1 &&
         !(pre == LEVEL_PRE_UP && n < 2);
}
static void conv_wg_grid(int B, int rows_per_sample, int n, int* tiles_ps, int* tpb, int* gx) {
  *tiles_ps = cdiv((int64_t)rows_per_sample * n, TILE);
  const int64_t total = (int64_t)*tiles_ps * B;
  *tpb = (int)std::max<int64_t>(1, (total + 511) / 512);  // (two workgroups per CU at most: the LDS images)
  *gx = cdiv(*tiles_ps, *tpb);
}
int64_t conv_wg_part_floats(int C, int pre, int cp, int B, int rows_per_sample, int n) {
  int tiles_ps, tpb, gx;
  conv_wg_grid(B, rows_per_sample, n, &tiles_ps, &tpb, &gx);
  return (int64_t)gx * B * (C * cp * (pre == LEVEL_PRE_DOWN ? 4 : 3) + C);
}

int launch_conv_bwd_wg(const ConvBwdWg& a_in, hipStream_t s, ResWgReduce* red_out) {
  ConvBwdWg a = a_in;
  DQ_REQUIRE(conv_wg_usable(a.C, a.pre, a.cp, a.n, a.rows_per_sample), "conv_bwd_wg: unsupported shape");
  DQ_REQUIRE(a.dy && a.in && a.w && a.part && a.dparams && a.rows % a.rows_per_sample == 0, "conv_bwd_wg: missing operand");
  DQ_REQUIRE((int64_t)a.rows * std::max(a.C, a.cp) * std::max(a.n, 2) * 2 * 4 < (1ll << 32), "conv_bwd_wg: tensors of 4 GB or more are not built (32-bit byte offsets)");
  const int B = a.rows / a.rows_per_sample;
  const int K = a.pre == LEVEL_PRE_DOWN ? 4 : 3;
  int gx;
  conv_wg_grid(B, a.rows_per_sample, a.n, &a.tiles_ps, &a.tpb, &gx);
  a.nv = a.C * a.cp * K + a.C;
  DQ_REQUIRE(a.part_floats >= (int64_t)gx * B * a.nv, "conv_bwd_wg: slot scratch too small");
  const size_t lds = sizeof(float) * ((size_t)pad4(a.C * (a.cp / 4) * K) * 4 + img_floats(a.C) + (size_t)K * img_floats(a.cp));
  DQ_REQUIRE(lds <= 160 * 1024, "conv_bwd_wg: LDS images too large");
  dim3 grid(gx, B), block(256);
#define DQ_CW(CC, PP, PC)                                                                                              \
  if (a.C == CC && a.pre == PP && a.cp == PC) {                                                                        \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      DQ_HIP_OK(hipFuncSetAttribute((const void*)k_conv_bwd_wg<CC, PP, PC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_conv_bwd_wg<CC, PP, PC>), grid, block, lds, s, a);                                           \
    DQ_LAUNCH_CHECK();                                                                                                 \
  } else
  DQ_CW(4, LEVEL_PRE_DOWN, 4) DQ_CW(8, LEVEL_PRE_DOWN, 4) DQ_CW(8, LEVEL_PRE_DOWN, 8) DQ_CW(12, LEVEL_PRE_DOWN, 8) DQ_CW(12, LEVEL_PRE_DOWN, 12)
  DQ_CW(16, LEVEL_PRE_DOWN, 12) DQ_CW(16, LEVEL_PRE_DOWN, 16)
  DQ_CW(4, LEVEL_PRE_UP, 4) DQ_CW(4, LEVEL_PRE_UP, 8) DQ_CW(8, LEVEL_PRE_UP, 8) DQ_CW(8, LEVEL_PRE_UP, 12) DQ_CW(12, LEVEL_PRE_UP, 12) DQ_CW(12, LEVEL_PRE_UP, 16)
  DQ_CW(16, LEVEL_PRE_UP, 16)
  DQ_CW(4, LEVEL_PRE_S1, 4) DQ_CW(4, LEVEL_PRE_S1, 8) DQ_CW(8, LEVEL_PRE_S1, 8) DQ_CW(8, LEVEL_PRE_S1, 12) DQ_CW(12, LEVEL_PRE_S1, 12) DQ_CW(12, LEVEL_PRE_S1, 16)
  DQ_CW(16, LEVEL_PRE_S1, 16)
  { set_error("conv_bwd_wg: unsupported (C, stage, input width)"); return 2; }
#undef DQ_CW
  if (red_out) {
    ResWgReduce r;
    r.part = a.part; r.B = B; r.gx = gx; r.nv = a.nv; r.nglob = a.nv; r.C = 0;
    r.dst = a.dparams; r.dss = a.dparams; r.ss_stride = 0;  // (no per-sample part)
    *red_out = r;
  }
  return 0;
}

}  // namespace dq
