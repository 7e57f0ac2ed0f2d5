// ResnetBlock backward for the wide m/z levels (4 / 8 channels, rows of 8..256 positions) WITH the block's weight gradients
// formed in the same launch (reference dquartic/model/unet1d.py:271-323; autograd of it is what model_interface.py:1120 runs).
//
// k_res_bwd (k_res.hip) leaves dU2 / dU1 in memory and three weight-gradient launches (+ reduces) on the side stream re-read
// them together with a1, x (twice) and d out: six tensor reads and two tensor writes per block, at a time when the main chain's
// kernels want the same HBM.  Here the data path is the same (thread = (row, position), all channels in registers), but
//   * every operand the weight gradients need is staged ONCE in LDS as a [position][channel] image (16-byte stores; the k = 3
//     neighbour exchange of dU2 / dU1 reads the same images with 16-byte loads),
//   * dW2 = dU2 (x) a1, dW1 = dU1 (x) x, dWr = d out (x) x and the three bias sums run on the matrix pipe, which this kernel
//     otherwise leaves idle: v_mfma_f32_4x4x1 takes 16 positions per instruction as 16 independent (4 co) x (4 ci) outer
//     products, lane (blk, i) supplying channel 4g + i of position 16 blk + s at step s -- a block of lanes walks a RUN of 16
//     consecutive positions, so the operand of tap k - 1 / k + 1 is the centre operand of the previous / next step (one LDS read
//     per step and input-channel quad serves three MFMAs); the (co quad, ci quad, tap) jobs are split over the four waves,
//   * the data path's own (transposed) convolutions  d a1 = W2^T * dU2,  d x = W1^T * dU1 + Wr^T d out  run on the matrix pipe too, in the
//     lane = position form of k_res_mm.hip: B operand = the register holding dU[co] (one DPP wave shift for the outer taps), A
//     operand = W[co][4 g + (lane & 3)][k] from an LDS operand image, result = "channel 4 g + i of this lane's position in register i".
//     (As VALU FMAs with the weights read from LDS as broadcasts -- one LDS instruction per FMA -- this kernel took 3-4x longer than
//     the data path + the separate weight-gradient launches it replaces.)
//   * a1 = SiLU(norm(u1) (scale + 1) + shift) is recomputed from u1 (which the data path loads anyway): the forward stores one
//     tensor less per block in training,
//   * a workgroup walks several 256-position tiles of ONE sample with its accumulators (MFMA results, norm-gain and scale /
//     shift sums) in registers and leaves them in its own slot [c1.w | c1.b | g1 | c2.w | c2.b | g2 | res.w | res.b | dscale |
//     dshift] -- the order of the flat parameter buffer -- for k_res_wg_reduce to add up in block order: no atomics, bitwise
//     repeatable.
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
#include "k_res_common.h"
#include "dq_probe.h"
#include <algorithm>
#include <cstdlib>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// value of lane - 1 / lane + 1 (0 beyond the wave's ends); every lane takes part: never under control flow
__device__ __forceinline__ float lane_m1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_p1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true)); }
constexpr int pad4(int x) { return (x + 3) / 4 * 4; }

constexpr int TILE = 256;
constexpr int RUN = 16;  // consecutive positions a 4-lane block of the MFMA walks

// LDS image of one C-channel operand over a tile: [position + 1 (one halo slot on either side)][channel], C = 8 padded to 12
// floats per position (16-byte accesses of consecutive positions then fall into different 16-byte bank slots), and 4 floats of
// padding after every 16 positions: the 8 lane blocks of a half-wave read positions 16 apart, whose distance in floats must be
// an odd multiple of 4 modulo the 32 banks of a 4-byte LDS read.
template <int C>
struct Img {
  static constexpr int ST = C == 4 ? 4 : (C == 8 ? 12 : 20);  // (floats per position: an odd number of 16-byte slots)
  static constexpr int FLOATS = (TILE + 2) * ST + ((TILE + 2) / RUN + 1) * 4;
  __device__ static __forceinline__ int at(int q) { return q * ST + (q >> 4) * 4; }  // q = position in the tile + 1
};
// offset of step s of a run from the run's first position (the run's last position lies behind a padding slot)
template <int ST>
__device__ __forceinline__ constexpr int run_off(int s) { return s * ST + (s == RUN - 1 ? 4 : 0); }


}  // namespace

template <int C, bool WR>
__global__ void __launch_bounds__(256, C >= 12 ? 1 : 2) k_res_bwd_wg(ResBwdWg a) {
  using I = Img<C>;
  constexpr int ST = I::ST, CQ = C / 4;
  // (dynamic LDS: 12 / 16 channels need 84 - 140 KB -- one workgroup per CU, whose 512 registers per lane the 47 accumulator quads of a
  // 16-channel block with a residual conv need anyway)
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* i_du2 = lds_all;
  float* i_du1 = i_du2 + I::FLOATS;
  float* i_a1 = i_du1 + I::FLOATS;
  float* i_xa = i_a1 + I::FLOATS;
  float* i_xb = i_xa + I::FLOATS;
  float* i_do = i_xb + (WR ? I::FLOATS : 4);
  float (*red)[4 * C] = reinterpret_cast<float (*)[4 * C]>(i_do + (WR ? I::FLOATS : 4));
  // operand image of the TRANSPOSED convolutions of the data path, [job / 4][lane & 3][job % 4]:
  //   T2: job (co, g, k)  = W2[co][4 g + li][k]      (d a1: C output channels = input channels of conv2)
  //   T1: job (co, gi, k) = W1[co][4 gi + li][k]     (d x: cin channels, gi < GI quads of cat(A, B))
  //   TR: job (co, gi)    = Wr[co][4 gi + li]
  constexpr int GI = WR ? 2 * CQ : CQ;
  constexpr int JT2 = C * CQ * 3, JT1 = C * GI * 3, JTR = WR ? C * GI : 0;
  constexpr int OT1 = pad4(JT2), OTR = OT1 + pad4(JT1), JTT = OTR + pad4(JTR);
  float* wl = reinterpret_cast<float*>(red) + 4 * 4 * C;   // JTT * 4 floats
  float* prm = wl + JTT * 4;  // 4 C floats: g2 | g1 | scale | shift (this sample's): no global reads of them inside the tile loop
  const int cin = a.cinA + a.cinB;  // C (identity residual) or C + cinB, cinB in {4, .., C} (checked by the launcher)
  DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 0);
  for (int idx = threadIdx.x; idx < JTT * 4; idx += 256) {
    const int j = (idx >> 4) * 4 + (idx & 3), l4 = (idx >> 2) & 3;
    float v = 0.f;
    if (j < JT2) {
      const int k = j % 3, gg = (j / 3) % CQ, co = j / (3 * CQ);
      v = a.w2[(co * C + 4 * gg + l4) * 3 + k];
    } else if (j >= OT1 && j < OT1 + JT1) {
      const int jj = j - OT1, k = jj % 3, gi = (jj / 3) % GI, co = jj / (3 * GI), ci = 4 * gi + l4;
      if (ci < cin) v = a.w1[(co * cin + ci) * 3 + k];
    } else if (WR && j >= OTR && j < OTR + JTR) {
      const int jj = j - OTR, gi = jj % GI, co = jj / GI, ci = 4 * gi + l4;
      if (ci < cin) v = a.wr[co * cin + ci];
    }
    wl[idx] = v;
  }
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    const int what = i / C, c = i % C;
    const float* ssp = a.ss + (int64_t)blockIdx.y * a.ss_stride;
    prm[i] = what == 0 ? a.g2[c] : what == 1 ? a.g1[c] : what == 2 ? ssp[c] : ssp[C + c];
  }
  __syncthreads();  // (the first tile reads prm before its first barrier)
  DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 1);
  const int b = blockIdx.y, n = a.n;
  const int per_sample = a.rows_per_sample * n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = tid + 1;                      // this thread's slot in the images
  const int blk = lane >> 2, li = lane & 3;   // MFMA: 4-lane block and the lane's channel inside a quad
  const int lb = I::at(blk * RUN + 1) + li;   // image offset of (first position of the block's run, channel li)
  // rows never straddle a run of 16 unless they are 8 long (then every run is two rows); n is a power of two in 8..64
  const bool n8 = n == 8;
  const bool zl = ((blk * RUN) % n) == 0;            // the run starts a row: its tap k = 0 operand at step 0 is the zero padding
  const bool zr = ((blk * RUN + RUN) % n) == 0;      // the run ends a row
  // ---- which weight-gradient jobs this wave runs (wave-uniform)
  //   C = 8: wave = (co quad g, half hs): dW2 (g, ci quad hs), dWr / dW1 (g, ci quads 2 hs, 2 hs + 1 of cat(A, B)) or dW1 (g, hs)
  //          for the identity residual; the bias sums on the waves with hs = 0
  //   C = 4: wave 0: dW2 + db2; wave 1: dWr + dbr; wave 2: dW1 (ci quad 0) + db1; wave 3: dW1 (ci quad 1)
  const int cinq = cin >> 2;
  const int g = C == 8 ? (wv & 1) : 0, hs = C == 8 ? (wv >> 1) : wv;
  const bool doW2 = C == 8 ? true : wv == 0;
  const int hW2 = C == 8 ? hs : 0;
  const bool doB2 = C == 8 ? hs == 0 : wv == 0;
  const bool doWr = WR && (C == 8 ? true : wv == 1);
  const int hr0 = C == 8 ? 2 * hs : 0, hr1 = C == 8 ? 2 * hs + 1 : 1;
  const bool doBr = WR && (C == 8 ? hs == 0 : wv == 1);
  const bool doW1 = C == 8 ? true : (wv == 2 || (WR && wv == 3));
  const int h10 = C == 8 ? (WR ? 2 * hs : hs) : (wv == 3 ? 1 : 0);
  const int h11 = (C == 8 && WR) ? 2 * hs + 1 : -1;
  const bool doB1 = C == 8 ? hs == 0 : wv == 2;
  const bool v_r0 = doWr && hr0 < cinq, v_r1 = doWr && hr1 < cinq;
  const bool v_10 = doW1 && h10 < cinq, v_11 = doW1 && h11 >= 0 && h11 < cinq;
  auto ximg = [&](int h) -> const float* { return (h < CQ || !WR ? i_xa + 4 * (h < CQ ? h : 0) : i_xb + 4 * (h - CQ)) + lb; };

  // 12 / 16 channels (WIDE): wave = output-channel quad (12 channels: the fourth wave repeats quad 0 and is not flushed); it runs every
  // input-channel quad of dW2 (CQ), dWr and dW1 (GI of cat(A, B)).  Rows of any power-of-two length <= 64: the taps' zero padding per step
  // and lane from the position's place in its row.
  constexpr bool WIDE = C >= 12;
  constexpr int NQ2 = WIDE ? CQ : 1, NQ1 = WIDE ? GI : 2;
  const int gw = WIDE ? (wv < CQ ? wv : 0) : g;   // this wave's output-channel quad
  const bool w_on = !WIDE || wv < CQ;             // its accumulators are flushed
  f32x4 aW2[NQ2][3], aWr[NQ1], aW1[NQ1][3], aB2 = {0.f, 0.f, 0.f, 0.f}, aBr = aB2, aB1 = aB2;
#pragma unroll
  for (int h = 0; h < NQ2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) aW2[h][k] = aB2;
#pragma unroll
  for (int h = 0; h < NQ1; ++h) {
#pragma unroll
    for (int k = 0; k < 3; ++k) aW1[h][k] = aB2;
    aWr[h] = aB2;
  }
  float dg2[C], dg1[C], dsc[C], dsh[C];
#pragma unroll
  for (int c = 0; c < C; ++c) dg2[c] = dg1[c] = dsc[c] = dsh[c] = 0.f;
  const float sqC = sqrtf((float)C);
  const float* wlane = wl + li * 4;
  auto wop = [&](int j) -> float { return wlane[(j >> 2) * 16 + (j & 3)]; };  // (four jobs of a group: one 16-byte read)

  const int tile_end = min(a.tiles_ps, ((int)blockIdx.x + 1) * a.tpb);
#pragma unroll 1
  for (int tile = blockIdx.x * a.tpb; tile < tile_end; ++tile) {
    const int it = tile * TILE + tid;
    const bool live = it < per_sample;
    // (no load is predicated: the threads beyond the sample's last position read its last position and contribute zeros -- d out := 0)
    const int itc = live ? it : per_sample - 1;
    const int row = b * a.rows_per_sample + itc / n, p = itc % n;
    // addresses: wave-uniform (tensor + channel) base in scalar registers + ONE 32-bit byte offset per lane and tensor shape (the launcher
    // checks every tensor stays below 4 GB); as 64-bit element indices per channel the address arithmetic of the 16 stores at the end
    // of a tile alone took ~9,000 clocks
    const unsigned boA = (((unsigned)row * C) * (unsigned)n + (unsigned)p) * 4u;  // (row, channel 0, p) of a (rows, C, n) tensor
    const unsigned boB = (((unsigned)row * (unsigned)a.cinB) * (unsigned)n + (unsigned)p) * 4u;
    auto ld = [&](const float* base, int c, unsigned boff) -> float {
      return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (size_t)c * n) + boff);
    };
    auto st = [&](float* base, int c, unsigned boff, float v) {
      *reinterpret_cast<float*>(reinterpret_cast<char*>(base + (size_t)c * n) + boff) = v;
    };
    const bool hasL = live && p > 0, hasR = live && p + 1 < n;
    // ---- every global read of the tile up front
    float dout[C], d[C], u[C], u1v[C], xa[C], xb[WR ? C : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      dout[c] = ld(a.dout, c, boA);
      u[c] = ld(a.u2, c, boA);
      u1v[c] = ld(a.u1, c, boA);
      xa[c] = ld(a.inA, c, boA);  // cinA == C
    }
    if constexpr (WR) {
#pragma unroll
      for (int c = 0; c < C; ++c) xb[c] = c < a.cinB ? ld(a.inB, c, boB) : 0.f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) { dout[c] = live ? dout[c] : 0.f; d[c] = dout[c]; }
    // ---- block2: dU2; block1's activation a1 recomputed from u1 with the forward's expression
    norm_act_bwd<C, false>(u, d, prm, nullptr, dg2, nullptr, nullptr);
    float a1v[C];
    {
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) ssq = fmaf(u1v[c], u1v[c], ssq);
      const float inv = rms_inv(ssq, sqC);
#pragma unroll
      for (int c = 0; c < C; ++c) a1v[c] = silu_f(fmaf(u1v[c] * inv * prm[C + c], prm[2 * C + c] + 1.0f, prm[3 * C + c]));
    }
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 2);
    __syncthreads();  // the previous tile's readers of the images are done (first tile: the staged weights are visible)
    {
      const int o = I::at(q);
#pragma unroll
      for (int c4 = 0; c4 < CQ; ++c4) {
        const int c = 4 * c4;
        *reinterpret_cast<float4*>(i_du2 + o + c) = live ? make_float4(d[c], d[c + 1], d[c + 2], d[c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(i_a1 + o + c) = make_float4(a1v[c], a1v[c + 1], a1v[c + 2], a1v[c + 3]);
        *reinterpret_cast<float4*>(i_xa + o + c) = make_float4(xa[c], xa[c + 1], xa[c + 2], xa[c + 3]);
        if constexpr (WR) {
          *reinterpret_cast<float4*>(i_xb + o + c) = make_float4(xb[c], xb[c + 1], xb[c + 2], xb[c + 3]);
          *reinterpret_cast<float4*>(i_do + o + c) = make_float4(dout[c], dout[c + 1], dout[c + 2], dout[c + 3]);
        }
      }
    }
    __syncthreads();
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 3);
    // ---- matrix pipe: dW2 (+ db2), dWr (+ dbr)
    if constexpr (WIDE) {
      // zero padding of the outer taps: position 16 blk + s is the first / last of its row (per lane; n is a power of two)
      auto first_in_row = [&](int s) { return ((blk * RUN + s) & (n - 1)) == 0; };
      auto last_in_row = [&](int s) { return ((blk * RUN + s) & (n - 1)) == n - 1; };
      {
        const float* A = i_du2 + lb + 4 * gw;
        float bprev[CQ], bcur[CQ];
#pragma unroll
        for (int h = 0; h < CQ; ++h) { bprev[h] = (i_a1 + lb + 4 * h)[-ST]; bcur[h] = (i_a1 + lb + 4 * h)[0]; }
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
          const float av = A[run_off<ST>(s)];
          const bool fl = first_in_row(s), ll = last_in_row(s);
#pragma unroll
          for (int h = 0; h < CQ; ++h) {
            const float* Bp = i_a1 + lb + 4 * h;
            const float bnext = s + 1 < RUN ? Bp[run_off<ST>(s + 1 < RUN ? s + 1 : s)] : Bp[RUN * ST + 4];
            aW2[h][0] = mfma4(av, fl ? 0.f : bprev[h], aW2[h][0]);
            aW2[h][1] = mfma4(av, bcur[h], aW2[h][1]);
            aW2[h][2] = mfma4(av, ll ? 0.f : bnext, aW2[h][2]);
            bprev[h] = bcur[h]; bcur[h] = bnext;
          }
          aB2 = mfma4(av, 1.f, aB2);
        }
      }
      if constexpr (WR) {
        const float* A = i_do + lb + 4 * gw;
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
          const float av = A[run_off<ST>(s)];
#pragma unroll
          for (int h = 0; h < GI; ++h) aWr[h] = mfma4(av, (h < CQ ? i_xa + 4 * h : i_xb + 4 * (h - CQ))[lb + run_off<ST>(s)], aWr[h]);
          aBr = mfma4(av, 1.f, aBr);
        }
      }
    } else {
    if (doW2) {
      const float* A = i_du2 + lb + 4 * g;
      const float* Bp = i_a1 + lb + 4 * hW2;
      float bprev = zl ? 0.f : Bp[-ST], bcur = Bp[0];
#pragma unroll
      for (int s = 0; s < RUN; ++s) {
        const float av = A[run_off<ST>(s)];
        float bnext = s + 1 < RUN ? Bp[run_off<ST>(s + 1 < RUN ? s + 1 : s)] : (zr ? 0.f : Bp[RUN * ST + 4]);
        float b0 = bprev, b2 = bnext;
        if (s == 8) b0 = n8 ? 0.f : b0;
        if (s == 7) b2 = n8 ? 0.f : b2;
        aW2[0][0] = mfma4(av, b0, aW2[0][0]);
        aW2[0][1] = mfma4(av, bcur, aW2[0][1]);
        aW2[0][2] = mfma4(av, b2, aW2[0][2]);
        aB2 = mfma4(av, 1.f, aB2);  // (every wave that runs this phase: flushed only where doB2 -- no branch per MFMA, k_conv_wg.hip)
        bprev = bcur; bcur = bnext;
      }
    }
    if constexpr (WR) {
     if (doWr) {
      const float* A = i_do + lb + 4 * g;
      const float* X0 = ximg(hr0);
      const float* X1 = ximg(v_r1 ? hr1 : hr0);
#pragma unroll
      for (int s = 0; s < RUN; ++s) {
        const float av = A[run_off<ST>(s)];
        aWr[0] = mfma4(av, X0[run_off<ST>(s)], aWr[0]);  // (jobs that are not this wave's read a valid image and are not flushed)
        aWr[1] = mfma4(av, X1[run_off<ST>(s)], aWr[1]);
        aBr = mfma4(av, 1.f, aBr);
      }
     }
    }
    }  // narrow
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 4);
    // ---- d a1[ci][p] = sum_co sum_k W2[co][ci][k] dU2[co][p + 1 - k]  (matrix pipe; tap k reads position p + 1 - k)
    float da1[C];
    {
      f32x4 acc[CQ][3];
#pragma unroll
      for (int gg = 0; gg < CQ; ++gg)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[gg][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int co = 0; co < C; ++co) {
        const float dc = live ? d[co] : 0.f;
        const float tp = lane_p1(dc), tm = lane_m1(dc);
        const float dr = hasR ? tp : 0.f, dl = hasL ? tm : 0.f;
#pragma unroll
        for (int gg = 0; gg < CQ; ++gg) {
          const int j = (co * CQ + gg) * 3;
          acc[gg][0] = mfma4(wop(j + 0), dr, acc[gg][0]);
          acc[gg][1] = mfma4(wop(j + 1), dc, acc[gg][1]);
          acc[gg][2] = mfma4(wop(j + 2), dl, acc[gg][2]);
        }
      }
#pragma unroll
      for (int gg = 0; gg < CQ; ++gg)
#pragma unroll
        for (int i = 0; i < 4; ++i) da1[4 * gg + i] = (acc[gg][0][i] + acc[gg][1][i]) + acc[gg][2][i];
    }
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 5);
    // ---- block1: dU1
#pragma unroll
    for (int c = 0; c < C; ++c) u[c] = u1v[c];
    norm_act_bwd<C, true>(u, da1, prm + C, prm + 2 * C, dg1, dsc, dsh);
    {
      const int o = I::at(q);
#pragma unroll
      for (int c4 = 0; c4 < CQ; ++c4) {
        const int c = 4 * c4;
        *reinterpret_cast<float4*>(i_du1 + o + c) = live ? make_float4(da1[c], da1[c + 1], da1[c + 2], da1[c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 6);
    __syncthreads();
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 7);
    // ---- matrix pipe: dW1 (+ db1)
    if constexpr (WIDE) {
      auto first_in_row = [&](int s) { return ((blk * RUN + s) & (n - 1)) == 0; };
      auto last_in_row = [&](int s) { return ((blk * RUN + s) & (n - 1)) == n - 1; };
      const float* A = i_du1 + lb + 4 * gw;
      float xprev[GI], xcur[GI];
#pragma unroll
      for (int h = 0; h < GI; ++h) {
        const float* X = (h < CQ ? i_xa + 4 * h : i_xb + 4 * (h - CQ)) + lb;
        xprev[h] = X[-ST]; xcur[h] = X[0];
      }
#pragma unroll
      for (int s = 0; s < RUN; ++s) {
        const float av = A[run_off<ST>(s)];
        const bool fl = first_in_row(s), ll = last_in_row(s);
#pragma unroll
        for (int h = 0; h < GI; ++h) {
          const float* X = (h < CQ ? i_xa + 4 * h : i_xb + 4 * (h - CQ)) + lb;
          const float xnext = s + 1 < RUN ? X[run_off<ST>(s + 1 < RUN ? s + 1 : s)] : X[RUN * ST + 4];
          aW1[h][0] = mfma4(av, fl ? 0.f : xprev[h], aW1[h][0]);
          aW1[h][1] = mfma4(av, xcur[h], aW1[h][1]);
          aW1[h][2] = mfma4(av, ll ? 0.f : xnext, aW1[h][2]);
          xprev[h] = xcur[h]; xcur[h] = xnext;
        }
        aB1 = mfma4(av, 1.f, aB1);
      }
    } else {
    if (doW1) {
      const float* A = i_du1 + lb + 4 * g;
      const float* X0 = ximg(v_10 ? h10 : 0);
      const float* X1 = ximg(v_11 ? h11 : 0);
      float p0 = zl ? 0.f : X0[-ST], c0 = X0[0], p1 = zl ? 0.f : X1[-ST], c1 = X1[0];
#pragma unroll
      for (int s = 0; s < RUN; ++s) {
        const float av = A[run_off<ST>(s)];
        float n0 = s + 1 < RUN ? X0[run_off<ST>(s + 1 < RUN ? s + 1 : s)] : (zr ? 0.f : X0[RUN * ST + 4]);
        float n1 = s + 1 < RUN ? X1[run_off<ST>(s + 1 < RUN ? s + 1 : s)] : (zr ? 0.f : X1[RUN * ST + 4]);
        float l0 = p0, r0 = n0, l1 = p1, r1 = n1;
        if (s == 8) { l0 = n8 ? 0.f : l0; l1 = n8 ? 0.f : l1; }
        if (s == 7) { r0 = n8 ? 0.f : r0; r1 = n8 ? 0.f : r1; }
        aW1[0][0] = mfma4(av, l0, aW1[0][0]);
        aW1[0][1] = mfma4(av, c0, aW1[0][1]);
        aW1[0][2] = mfma4(av, r0, aW1[0][2]);
        if constexpr (C == 8 && WR) {  // (the only shape with a second input-channel quad per wave)
          aW1[1][0] = mfma4(av, l1, aW1[1][0]);
          aW1[1][1] = mfma4(av, c1, aW1[1][1]);
          aW1[1][2] = mfma4(av, r1, aW1[1][2]);
        }
        aB1 = mfma4(av, 1.f, aB1);
        p0 = c0; c0 = n0; p1 = c1; c1 = n1;
      }
    }
    }  // narrow
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 8);
    // ---- d x[ci][p] = sum_co sum_k W1[co][ci][k] dU1[co][p + 1 - k]  (+ residual branch) into dA / dB  (matrix pipe)
    // The input-channel quads of cat(A, B) in two passes (A's, then B's): half the accumulators live at a time -- with all 2 C / 4 quads
    // of a residual-conv block in flight the 8-channel instantiation needed 285 registers against the 256 of two waves per SIMD, and
    // its spill reloads waited behind the tile's stores.
    if (a.dA || a.dB) {
#pragma unroll
      for (int pass = 0; pass < (WR ? 2 : 1); ++pass) {
        f32x4 acc[CQ][3], ar[CQ];
#pragma unroll
        for (int q4 = 0; q4 < CQ; ++q4) {
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[q4][k] = f32x4{0.f, 0.f, 0.f, 0.f};
          ar[q4] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float dc = live ? da1[co] : 0.f;  // dU1
          const float tp = lane_p1(dc), tm = lane_m1(dc);
          const float dr = hasR ? tp : 0.f, dl = hasL ? tm : 0.f;
#pragma unroll
          for (int q4 = 0; q4 < CQ; ++q4) {
            // (every quad, also those beyond cin: their operands are the image's zero padding and their results are not stored -- a
            // wave-uniform branch per quad made every group of four MFMAs its own basic block with its LDS reads waited for in place)
            const int gi = pass * CQ + q4;
            const int j = OT1 + (co * GI + gi) * 3;
            acc[q4][0] = mfma4(wop(j + 0), dr, acc[q4][0]);
            acc[q4][1] = mfma4(wop(j + 1), dc, acc[q4][1]);
            acc[q4][2] = mfma4(wop(j + 2), dl, acc[q4][2]);
            if constexpr (WR) ar[q4] = mfma4(wop(OTR + co * GI + gi), dout[co], ar[q4]);
          }
        }
        if (pass == 0) DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 11);
        float v[C];
#pragma unroll
        for (int q4 = 0; q4 < CQ; ++q4)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * q4 + i] = (acc[q4][0][i] + acc[q4][1][i]) + (acc[q4][2][i] + (WR ? ar[q4][i] : dout[4 * q4 + i]));
        // The old values of a gradient tensor this launch is not the first writer of are requested together, before the first store.
        if (pass == 0) {  // channels 0 .. C - 1 of cat(A, B) are A's (cinA == C)
          if (live && a.dA) {
            float oldv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) oldv[c] = 0.f;
            if (!a.dA_store) {
#pragma unroll
              for (int c = 0; c < C; ++c) oldv[c] = ld(a.dA, c, boA);
            }
            DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 12);
#pragma unroll
            for (int c = 0; c < C; ++c) st(a.dA, c, boA, oldv[c] + v[c]);
          }
        } else {
          if (live && a.dB) {
            float oldv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) oldv[c] = 0.f;
            if (!a.dB_store) {
#pragma unroll
              for (int c = 0; c < C; ++c) oldv[c] = c < a.cinB ? ld(a.dB, c, boB) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
              if (c < a.cinB) st(a.dB, c, boB, oldv[c] + v[c]);
          }
        }
      }
    }
    DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 9);
  }  // tile

  // ---- the block's slot: MFMA results (each job belongs to exactly one wave), then the VALU sums
  float* part = a.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * a.nv;
  const int oC1W = 0, oC1B = C * cin * 3, oG1 = oC1B + C, oC2W = oG1 + C, oC2B = oC2W + C * C * 3, oG2 = oC2B + C, oRW = oG2 + C,
            oRB = oRW + C * cin, nglob = WR ? oRB + C : oRW;
  auto put_w = [&](f32x4 acc, int base, int h, int kdim, int k, int ld) {  // D[i][j] -> W[4 g + i][4 h + j][k]
    const f32x4 t = blocks_sum(acc);
    if (lane < 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[base + ((4 * gw + i) * ld + 4 * h + lane) * kdim + k] = t[i];
    }
  };
  auto put_b = [&](f32x4 acc, int base) {
    const f32x4 t = blocks_sum(acc);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[base + 4 * gw + i] = t[i];
    }
  };
  if constexpr (WIDE) {
    if (w_on) {  // wave-uniform
#pragma unroll
      for (int h = 0; h < CQ; ++h)
#pragma unroll
        for (int k = 0; k < 3; ++k) put_w(aW2[h][k], oC2W, h, 3, k, C);
      put_b(aB2, oC2B);
#pragma unroll
      for (int h = 0; h < GI; ++h) {
        if (h < cinq) {
#pragma unroll
          for (int k = 0; k < 3; ++k) put_w(aW1[h][k], oC1W, h, 3, k, cin);
          if constexpr (WR) put_w(aWr[h], oRW, h, 1, 0, cin);
        }
      }
      put_b(aB1, oC1B);
      if constexpr (WR) put_b(aBr, oRB);
    }
  } else {
  if (doW2) {
#pragma unroll
    for (int k = 0; k < 3; ++k) put_w(aW2[0][k], oC2W, hW2, 3, k, C);
  }
  if (doB2) put_b(aB2, oC2B);
  if (v_r0) put_w(aWr[0], oRW, hr0, 1, 0, cin);
  if (v_r1) put_w(aWr[1], oRW, hr1, 1, 0, cin);
  if (doBr) put_b(aBr, oRB);
  if (v_10) {
#pragma unroll
    for (int k = 0; k < 3; ++k) put_w(aW1[0][k], oC1W, h10, 3, k, cin);
  }
  if (v_11) {
#pragma unroll
    for (int k = 0; k < 3; ++k) put_w(aW1[1][k], oC1W, h11, 3, k, cin);
  }
  if (doB1) put_b(aB1, oC1B);
  }  // narrow
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float s0 = wave_sum(dg2[c]), s1 = wave_sum(dg1[c]), s2 = wave_sum(dsc[c]), s3 = wave_sum(dsh[c]);
    if (lane == 0) { red[wv][c] = s0; red[wv][C + c] = s1; red[wv][2 * C + c] = s2; red[wv][3 * C + c] = s3; }
  }
  __syncthreads();
  for (int i = tid; i < 4 * C; i += 256) {
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    const int what = i / C, c = i % C;
    part[(what == 0 ? oG2 : what == 1 ? oG1 : what == 2 ? nglob : nglob + C) + c] = v;
  }
  DQ_PSTAMP(100000 + C * 10 + (WR ? 1 : 0), 10);
}

// -----------------------------------------------------------------------------------------------------------------
// ordered sums of the slots: dst[e] += sum over all blocks (e < nglob: the block's parameters are contiguous in the flat gradient
// buffer, in slot order), dss[b][j] += sum over the sample's blocks (j < 2 C).  16 elements x 16 partial groups per workgroup.
// -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_res_wg_reduce(ResWgReduceMulti m) {
  const ResWgReduce& r = m.it[blockIdx.y];
  const int twoC = 2 * r.C;
  const int nelem = r.nglob + r.B * twoC;
  if ((int)blockIdx.x * 16 >= nelem) return;  // uniform per block
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem) {
    const bool glob = e < r.nglob;
    const int bs = glob ? 0 : (e - r.nglob) / twoC;
    const float* src = r.part + (glob ? e : (int64_t)bs * r.gx * r.nv + r.nglob + (e - r.nglob) % twoC);
    const int cnt = glob ? r.gx * r.B : r.gx;
    int k = gq;
    // (eight loads in flight, added in the order of the two-at-a-time loop: same sums bit for bit, a quarter of the memory round trips)
    for (; k + 112 < cnt; k += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 16 * u) * r.nv];
#pragma unroll
      for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
    }
    for (; k + 16 < cnt; k += 32) {
      s0 += src[(int64_t)k * r.nv];
      s1 += src[(int64_t)(k + 16) * r.nv];
    }
    if (k < cnt) s0 += src[(int64_t)k * r.nv];
  }
  red[gq][el] = s0 + s1;
  __syncthreads();
  if (gq == 0 && e < nelem) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    if (e < r.nglob) r.dst[e] += s;
    else r.dss[(int64_t)((e - r.nglob) / twoC) * r.ss_stride + (e - r.nglob) % twoC] += s;
  }
}

bool res_wg_usable(int n, int C, int cinA, int cinB, int rows_per_sample) {
  // (rows of up to 64 positions: a row lives inside one wave, whose DPP shifts are the conv's neighbours; 4 / 8 channels: rows of 8 or
  // more, the run-of-16 walk of their weight-gradient phases special-cases only rows of 8)
  if (!(C == 4 || C == 8 || C == 12 || C == 16)) return false;
  // 12 / 16 channels (rows of 1..8 positions at the default widths): built and parity-tested, NOT the default -- measured inside the train
  // step (B = 32) the 14 launches take 496 us against 279 us of k_res_bwd_cp + their weight-gradient launches on the side stream: one
  // 256-position tile per workgroup leaves staging (15,000 clocks), the channel-strided loads of rows of 2 positions (28,000) and the flush of
  // 47 accumulator quads (24,000) with nothing to overlap them.  DQ_WG_WIDE=1 selects it (tests/test_blocks_gpu.py runs both).
  if (C >= 12 && !DQ_DEV_FLAG("DQ_WG_WIDE", '1')) return false;  // (dev switch; the 12 / 16-channel instantiations exist in the dev build only)
  const int nmin = C >= 12 ? 1 : 8;
  return n >= nmin && n <= 64 && (n & (n - 1)) == 0 && rows_per_sample > 1 && cinA == C && (cinB == 0 || (cinB % 4 == 0 && cinB <= C));
}

template <int C, bool WR>
static size_t res_wg_lds_bytes() {
  constexpr int CQ = C / 4, GI = WR ? 2 * CQ : CQ;
  constexpr int JT2 = C * CQ * 3, JT1 = C * GI * 3, JTR = WR ? C * GI : 0;
  constexpr int JTT = pad4(JT2) + pad4(JT1) + pad4(JTR);
  return sizeof(float) * ((size_t)(WR ? 6 : 4) * Img<C>::FLOATS + (WR ? 0 : 8) + 16 * C + (size_t)JTT * 4 + 4 * C);
}

// grid of a launch: a workgroup takes `tpb` consecutive tiles of one sample; about 1024 workgroups in all (one resident round at
// 2-4 per CU).  A pure function of the shape, so that the arena (sized without a device) and the launcher agree.
void res_wg_grid(int B, int rows_per_sample, int n, int* tiles_ps, int* tpb, int* gx) {
  *tiles_ps = cdiv((int64_t)rows_per_sample * n, TILE);
  const int64_t total = (int64_t)*tiles_ps * B;
  *tpb = (int)std::max<int64_t>(1, (total + 1023) / 1024);
  *gx = cdiv(*tiles_ps, *tpb);
}
int res_wg_nv(int C, int cin, bool wr) { return C * cin * 3 + C + C + C * C * 3 + C + C + (wr ? C * cin + C : 0) + 2 * C; }
int64_t res_wg_part_floats(int C, int cin, bool wr, int B, int rows_per_sample, int n) {
  int tiles_ps, tpb, gx;
  res_wg_grid(B, rows_per_sample, n, &tiles_ps, &tpb, &gx);
  return (int64_t)gx * B * res_wg_nv(C, cin, wr);
}

int launch_res_bwd_wg(const ResBwdWg& a_in, hipStream_t s, ResWgReduce* red_out) {
  ResBwdWg a = a_in;
  DQ_REQUIRE(res_wg_usable(a.n, a.C, a.cinA, a.cinB, a.rows_per_sample), "res_bwd_wg: unsupported shape");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0, "res_bwd_wg: rows must be a multiple of rows_per_sample");
  DQ_REQUIRE((int64_t)a.rows * a.C * a.n * 4 < (1ll << 32), "res_bwd_wg: tensors of 4 GB or more are not built (32-bit byte offsets)");
  DQ_REQUIRE(a.wr || a.cinB == 0, "res_bwd_wg: identity residual needs C input channels");
  DQ_REQUIRE(a.dout && a.u1 && a.u2 && a.inA && (a.cinB == 0 || a.inB) && a.w1 && a.w2 && a.g1 && a.g2 && a.ss && a.part,
             "res_bwd_wg: missing operand");
  const int B = a.rows / a.rows_per_sample;
  int gx;
  res_wg_grid(B, a.rows_per_sample, a.n, &a.tiles_ps, &a.tpb, &gx);
  const bool wr = a.wr != nullptr;
  // ONE resident round: with more workgroups than the CUs hold (1-4 each: the LDS images), the few left over ran as a second round
  // behind the first -- up to twice the time.  More tiles per workgroup instead (never fewer than the arena's slot count assumes).
  const void* fn = nullptr;
  size_t lds = 0;
#define DQ_WGK(CC, WW) if (a.C == CC && wr == WW) { fn = (const void*)k_res_bwd_wg<CC, WW>; lds = res_wg_lds_bytes<CC, WW>(); }
  DQ_WGK(4, true) DQ_WGK(4, false) DQ_WGK(8, true) DQ_WGK(8, false)
#ifdef DQ_DEV_SWITCHES  // (12 / 16 channels: built, parity-tested under DQ_WG_WIDE=1, slower than k_res_bwd_cp at those row lengths -- dev build only)
  DQ_WGK(12, true) DQ_WGK(12, false) DQ_WGK(16, true) DQ_WGK(16, false)
#endif
#undef DQ_WGK
  DQ_REQUIRE(fn && lds <= 160 * 1024, "res_bwd_wg: no kernel for this shape");
  {
    static int occ[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    int& o = occ[a.C / 4 - 1][wr];
    if (!o) {
      if (lds > 48 * 1024) DQ_HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int nb = 0;
      DQ_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds));
      o = std::max(1, std::min(nb, 6));
    }
    int dev = 0;
    hipDeviceProp_t pr;
    static const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    const int64_t resident = (int64_t)o * cus, total = (int64_t)a.tiles_ps * B;
    a.tpb = std::max(a.tpb, (int)((total + resident - 1) / resident));
    while (a.tpb < a.tiles_ps && (int64_t)cdiv(a.tiles_ps, a.tpb) * B > resident) ++a.tpb;  // (the per-sample rounding can still overshoot)
    gx = cdiv(a.tiles_ps, a.tpb);
  }
  a.nv = res_wg_nv(a.C, a.cinA + a.cinB, wr);
  DQ_REQUIRE(a.part_floats >= (int64_t)gx * B * a.nv, "res_bwd_wg: slot scratch too small");
  dim3 grid(gx, B), block(256);
#define DQ_WGL(CC, WW) if (a.C == CC && wr == WW) hipLaunchKernelGGL((k_res_bwd_wg<CC, WW>), grid, block, lds, s, a);
  DQ_WGL(4, true) DQ_WGL(4, false) DQ_WGL(8, true) DQ_WGL(8, false)
#ifdef DQ_DEV_SWITCHES
  DQ_WGL(12, true) DQ_WGL(12, false) DQ_WGL(16, true) DQ_WGL(16, false)
#endif
#undef DQ_WGL
  DQ_LAUNCH_CHECK();
  if (red_out) {
    ResWgReduce r;
    r.part = a.part; r.B = B; r.gx = gx; r.nv = a.nv; r.nglob = a.nv - 2 * a.C; r.C = a.C;
    r.dst = a.dparams; r.dss = a.dss; r.ss_stride = a.ss_stride;
    *red_out = r;
  }
  return 0;
}

int launch_res_wg_reduce(const ResWgReduce* items, int count, hipStream_t s) {
  if (count == 0) return 0;
  DQ_REQUIRE(count <= RES_WG_REDUCE_MAX, "res_wg_reduce: too many items");
  ResWgReduceMulti m;
  int nmax = 1;
  for (int i = 0; i < count; ++i) {
    m.it[i] = items[i];
    DQ_REQUIRE(items[i].part && items[i].dst && items[i].dss, "res_wg_reduce: missing operand");
    nmax = std::max(nmax, items[i].nglob + items[i].B * 2 * items[i].C);
  }
  hipLaunchKernelGGL(k_res_wg_reduce, dim3(cdiv(nmax, 16), count), dim3(256), 0, s, m);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
