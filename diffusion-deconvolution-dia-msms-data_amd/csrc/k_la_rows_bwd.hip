// Backward of Residual(PreNorm(LinearAttention)) (reference forward dquartic/model/unet1d.py:446-496, 64-79, 143-176; the reference's
// backward is autograd over those ops) for the DEEP levels -- m/z rows of 2 / 4 positions at 8 / 12 / 16 channels -- one m/z row per
// LANE COLUMN, every product on v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles).  k_la_bwd.hip spends a 32-position unit of block-diagonal
// 32 x 32 tiles on 8 .. 16 such rows: of a lane's 16 tile registers 1 - 2 mean anything, and its ~2,400 VALU instructions per unit and
// head are tile-wide work on padding.
//
//   wave = 16 rows; lane = (g = lane / 16, row = lane % 16); CPL = C / 4 channels per lane
//   c-layout : a C-channel tensor of position n lives in CPL registers, register r of lane (g, row) = channel CPL g + r of that row
//   d-layout : a head's 32 channels live in 2 x 4 registers, register (t, r) of lane (g, row) = channel 16 t + 4 g + r
// Both are at the same time the ACCUMULATOR layout of a 16 x 16 output tile whose rows are (a permutation of) the channels and a valid
// B operand of a product that sums over the channels (K-step s takes register s of every lane: lane group g supplies k = g), so every
// product chains register by register with no data movement, and everything per position -- the two RMSNorms, both softmaxes, the
// N x N scalars S[n][m] = sum_d q[d][n] k[d][m] of a row -- is per-lane arithmetic plus a sum over the four lane groups (two
// v_permlane swaps).  The algebra (per row and head; xh = PreNorm(x), W2 = Wo_h Wv_h, DY = d loss / d y_pre):
//   forward (recomputed)   k = softmax_n(Wk xh), q = 32^-0.5 softmax_d(Wq xh), S[n][m] = sum_d q[d][n] k[d][m],
//                          Z[:, n] = sum_m S[n][m] xh[:, m],  y_pre[:, n] += W2 Z[:, n]                       (k_la_small.hip)
//   dZ[:, n] = W2^T DY[:, n]                          K = C   (A = W2^T image)
//   dS[n][m] = sum_c dZ[c, n] xh[c, m]                per lane + group sum;      d xh[:, m] += sum_n S[n][m] dZ[:, n]
//   dq[d][n] = sum_m dS[n][m] k[d][m] ; dk[d][m] = sum_n dS[n][m] q[d][n]        per lane
//   softmax backward:  dql = q (dq - sum_d q dq / 32^-0.5) ;  dkl[d][m] = k[d][m] (dk[d][m] - T[d]),  T[d] = sum_m k dk = sum_n q[d][n] dq[d][n]
//   d xh[:, n] += Wq^T dql[:, n] + Wk^T dkl[:, n]     K = 32  (A = transposed weight images, B = the d-layout registers)
//   dWq += dql xh^T, dWk += dkl xh^T, dW2 += DY Z^T   K = rows = lanes: both operands through an LDS transpose
// dWv = Wo^T dW2 and dWo = dW2 Wv^T follow once per layer from the slot sum (k_linattn_dwvo).  A workgroup leaves ONE slot in the layout of
// k_la_bwd.hip (la_slot(C)): the ordered slot reduce is unchanged and the step stays bitwise repeatable.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_options.h"
#include "dq_probe.h"
#include <algorithm>
#include <climits>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// Sum over the four lanes {row, row + 16, row + 32, row + 48} (every lane receives it) ON THE MATRIX PIPE: a 16x16x4 product sums over
// k = lane / 16, so with A = 1 every output row is sum_g B[g][row].  One MFMA (the pipe is two-thirds idle in this kernel) instead of two
// v_permlane swaps, their register copies and two adds on the VALU, which is what bounds the kernel.
// Measured (12,800 rows, kernel trace): the matrix-pipe form and the v_permlane form run within 3 % of each other at every shape (the
// kernel is bound by the sum of VALU and MFMA issue at two waves per SIMD: 27,000 clocks per pair of head-tiles against 18,500 for one
// alone); the VALU form keeps 20 registers more free at 4 positions (each MFMA result is a 4-register tuple) and is used there.
template <bool ON_MFMA>
__device__ __forceinline__ float gsum_t(float t) {
  if (ON_MFMA) return mfma16(1.0f, t, f32x4{0.f, 0.f, 0.f, 0.f})[0];
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = __int_as_float(a[0]) + __int_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}
// the maximum stays on the VALU: v_permlane16_swap (rows 0 <-> 1, 2 <-> 3), v_permlane32_swap (halves) (tools/probe/blocks_sum.hip pins the semantics)
__device__ __forceinline__ float gmax(float t) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = fmaxf(__int_as_float(a[0]), __int_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return fmaxf(__int_as_float(b[0]), __int_as_float(b[1]));
}
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// sum over the 16 lanes of a row (every lane receives it): four DPP adds
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

struct LaRowsBwdK {
  const float* x; const float* ypre; const float* dy; float* dx;
  const float* prep; const float* g_pre; const float* g_out;
  float* part;  // one slot per workgroup, layout la_slot(C) of k_la_bwd.hip
  int rows, ntiles, dx_store;
};

constexpr int la_slot_floats(int C) { return 256 * C + 4 * C * C + 3 * C; }

// A workgroup = four waves = the FOUR HEADS of the same 16-row tiles (as in k_la_bwd.hip): a wave keeps its head's 36 operand values per
// lane and its head's weight-gradient accumulators in registers for the whole launch; the four d xh contributions meet in LDS behind one
// barrier and the last head's wave finishes the tile.  ~50 KB of LDS and <= 256 registers: 2 - 3 workgroups per CU, so that at a training
// batch (800 tiles x 4 heads for 1,024 SIMDs) several waves share a SIMD and cover each other's MFMA -> VALU hand-offs.  (The first
// version ran the four heads one after the other in ONE wave per tile: 46 us per launch at (12 channels, 4 positions, 12,800 rows) against
// 67 us for k_la_bwd.hip -- a chain of four heads on a single wave per SIMD.)
template <int C, int N>
__global__ void __launch_bounds__(256, 2) k_la_rows_bwd(LaRowsBwdK a) {
  static_assert(C == 8 || C == 12 || C == 16, "channel widths of the deep levels");
  // (Rows of 8 positions: the straightforward instantiation holds k, q, S and dS of eight positions -- 256 + 256 registers and scratch, one workgroup
  // per CU, no faster than k_la_bwd.hip's 121 us at 12,800 rows by the clocks of the 4-position kernel; this compiler's MFMA register-form
  // pass also crashes on it.  Not built.)
  static_assert(N == 2 || N == 4, "rows of 2 / 4 positions");
  constexpr int CPL = C / 4;            // channels per lane
  constexpr int RUN = CPL * N;          // a lane's contiguous run of a (row, C, N) tensor: channels CPL g .. CPL g + CPL - 1, all positions
  constexpr int VW = RUN % 4 == 0 ? 4 : 2, NV = RUN / VW;
  constexpr int LS = LA_ROWS_LANE_FLOATS;  // image floats per (head, lane)
  constexpr int UP = 17;                // pitch of the [channel][row] transposes (conflict-free both ways)
  constexpr int TP = 48;                // pitch of the [row][d] transposes: 16-byte stores, the two lane groups of a read phase 16 banks apart
  constexpr float scale = 0.17677669529663687f;  // dim_head^-0.5 (unet1d.py:481)
  auto gsum = [](float t) __attribute__((always_inline)) { return gsum_t<N == 2>(t); };
  // LDS: the operand image of the four heads ([head][lane][LS]: a wave reads only its head's quarter, 16 bytes at a time, at the point of use --
  // held in registers the 36 values per lane cost the second wave per SIMD at 4 positions; read from global memory per tile they were five
  // exposed L2 round trips per tile); shared by the four waves: xh and DY of the tile as [n][c][row] (B / A operands of the K = rows products);
  // per wave: two [row][d] tiles, which also carry the wave's d xh to the exchange at the end of a tile; the last wave's stash of x and dy.
  __shared__ __attribute__((aligned(16))) float img[4 * 64 * LS];
  __shared__ __attribute__((aligned(16))) float ush[2 * N * 16 * UP];
  __shared__ __attribute__((aligned(16))) float tls[4][2 * 16 * TP];
  __shared__ __attribute__((aligned(16))) float stash[64 * 2 * RUN];
  __shared__ float gains[2 * 16];  // g_pre | g_out (read per tile from here: as per-lane global addresses they were 64-bit values held -- and spilled -- across the tile loop)
  static_assert(64 * RUN <= 2 * 16 * TP, "a wave's d xh fits its tile region");
  DQ_PSTAMP((500000 + C * 100 + N), 0);
  {
    constexpr int T4 = 4 * 64 * LS / 4, NLD = T4 / 256;
    static_assert(T4 % 256 == 0, "whole rounds of 256 x 16 bytes");
    const float4* src = reinterpret_cast<const float4*>(a.prep + LA_PREP_ROWS);
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = src[u * 256 + (int)threadIdx.x];
#pragma unroll
    for (int u = 0; u < NLD; ++u) reinterpret_cast<float4*>(img)[u * 256 + (int)threadIdx.x] = v[u];
  }
  const int lane = threadIdx.x & 63, g = lane >> 4, row = lane & 15, hd = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ux = ush;
  float* ud = ush + N * 16 * UP;
  float* t0 = tls[hd];
  float* t1 = t0 + 16 * TP;
  // this head's A operands of this lane, contiguous in the image: [q: 8 | k: 8 | W2^T: 4 | Wq^T: 8 | Wk^T: 8] (the first three hold
  // 2 CPL / 2 CPL / CPL values)
  const float4* wp = reinterpret_cast<const float4*>(img + (hd * 64 + lane) * LS);
  auto ld8 = [&](int q4, float (&d)[8]) __attribute__((always_inline)) {
    const float4 v0 = wp[q4], v1 = wp[q4 + 1];
    d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
  };
  const bool bounded = a.prep[LA_PREP_BOUNDED] != 0.f;
  // channel rows of the shared transposes no lane writes (c >= C) are read as operand padding: zero them once
  if (C < 16) {
    for (int i = threadIdx.x; i < 2 * N * 16 * UP; i += 256) ush[i] = 0.f;
  }
  if (threadIdx.x < 32) gains[threadIdx.x] = (threadIdx.x & 15) < C ? (threadIdx.x < 16 ? a.g_pre : a.g_out)[threadIdx.x & 15] : 0.f;
  __syncthreads();
  DQ_PSTAMP((500000 + C * 100 + N), 1);
  const float sqC = sqrtf((float)C);

  // weight gradients of this head: dWq / dWk [d = 16 t + 4 g + r][c = row] ; dW2 [c' = 4 g + r][channel slot = row]
  f32x4 gq0 = {0.f, 0.f, 0.f, 0.f}, gq1 = gq0, gk0 = gq0, gk1 = gq0, gw = gq0;
  float na[CPL], nb[CPL];  // head 0's wave: d g_out, d b_out ; head 3's wave: d g_pre (in na)
#pragma unroll
  for (int r = 0; r < CPL; ++r) na[r] = nb[r] = 0.f;

#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int grow = tile * 16 + row;
    const bool live = grow < a.rows;
    // 32-bit BYTE offset of the lane's run against the wave-uniform tensor bases (the launcher checks 4 GB): no 64-bit per-lane address lives across the tile
    const unsigned boff = (unsigned)(((live ? grow : a.rows - 1) * C + CPL * g) * N) * 4u;
    auto at = [&](const float* t, int k) __attribute__((always_inline)) { return reinterpret_cast<const char*>(t) + boff + k * VW * 4; };
    typedef float vecf __attribute__((ext_vector_type(VW)));
    float xh[N][CPL], DY[N][CPL];
    {
    float xr[RUN], ur[RUN], dr[RUN];
    float gpre[CPL], gout[CPL];  // (per tile, L1-hot: not held across the head's work)
#pragma unroll
    for (int r = 0; r < CPL; ++r) { gpre[r] = gains[CPL * g + r]; gout[r] = gains[16 + CPL * g + r]; }
    {
      vecf vx[NV], vu[NV], vd[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        vx[k] = *reinterpret_cast<const vecf*>(at(a.x, k));
        vu[k] = *reinterpret_cast<const vecf*>(at(a.ypre, k));
        vd[k] = *reinterpret_cast<const vecf*>(at(a.dy, k));
      }
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          xr[k * VW + e] = live ? vx[k][e] : 0.f;  // (a row beyond the tensor contributes nothing: xh = DY = 0)
          ur[k * VW + e] = live ? vu[k][e] : 0.f;
          dr[k * VW + e] = live ? vd[k][e] : 0.f;
        }
      if (hd == 3) {  // the wave that finishes the tile keeps x and dy in LDS (its own lanes read them back: no fence beyond program order)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          vecf sx, sd;
#pragma unroll
          for (int e = 0; e < VW; ++e) { sx[e] = xr[k * VW + e]; sd[e] = dr[k * VW + e]; }
          *reinterpret_cast<vecf*>(stash + lane * 2 * RUN + k * VW) = sx;
          *reinterpret_cast<vecf*>(stash + lane * 2 * RUN + RUN + k * VW) = sd;
        }
      }
    }
    // ---- PreNorm recompute (unet1d.py:140, 171) and the post-norm backward (RMSNorm behind to_out, :470-473): DY = d loss / d y_pre
    // (every wave for itself: ~100 instructions against a second barrier and an LDS round trip)
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float ssq = 0.f, usq = 0.f;
#pragma unroll
      for (int r = 0; r < CPL; ++r) { ssq = fmaf(xr[r * N + n], xr[r * N + n], ssq); usq = fmaf(ur[r * N + n], ur[r * N + n], usq); }
      ssq = gsum(ssq);
      usq = gsum(usq);
      const float inv = rms_inv(ssq, sqC);
      const float unrm = fast_sqrt(usq), uinv = fast_rcp(fmaxf(unrm, RMS_EPS));
      float dot = 0.f, gdv[CPL], uh[CPL];
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        xh[n][r] = xr[r * N + n] * inv * gpre[r];
        uh[r] = ur[r * N + n] * uinv;
        if (hd == 0) na[r] = fmaf(dr[r * N + n], uh[r] * sqC, na[r]);  // d g_out
        gdv[r] = dr[r * N + n] * gout[r] * sqC;
        dot = fmaf(gdv[r], uh[r], dot);
      }
      dot = gsum(dot);
      const bool uclamped = unrm < RMS_EPS;  // F.normalize clamps the norm: below eps the map is linear
#pragma unroll
      for (int r = 0; r < CPL; ++r) {
        DY[n][r] = uclamped ? gdv[r] * uinv : uinv * (gdv[r] - uh[r] * dot);
        if (hd == 0) nb[r] += DY[n][r];  // d b_out
      }
    }
    }
    // xh (head 0's wave) and DY (head 1's) as [n][c][row]: lane (g', j) then reads element (c = j, row = 4 s + g') for K-step s of a product
    // over the rows.  (The previous tile's readers are behind its second barrier.)
    DQ_PSTAMP((500000 + C * 100 + N), 2);
    if (hd == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int r = 0; r < CPL; ++r) ux[(n * 16 + CPL * g + r) * UP + row] = xh[n][r];
    } else if (hd == 1) {
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int r = 0; r < CPL; ++r) ud[(n * 16 + CPL * g + r) * UP + row] = DY[n][r];
    }
    lds_barrier();
    DQ_PSTAMP((500000 + C * 100 + N), 3);
    float dxh[N][CPL];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < CPL; ++r) dxh[n][r] = 0.f;

    {
      // ---- k of every position; softmax over the positions, in the lane (unet1d.py:479)
      f32x4 kk[N][2];
      float ak[8], aq[8];
      ld8(2, ak);
      ld8(0, aq);
#pragma unroll
      for (int m = 0; m < N; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < CPL; ++s) acc = mfma16(ak[t * CPL + s], xh[m][s], acc);
          kk[m][t] = acc;
        }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float mx = 0.f;
          if (!bounded) {
            mx = kk[0][t][r];
#pragma unroll
            for (int m = 1; m < N; ++m) mx = fmaxf(mx, kk[m][t][r]);
          }
          float z = 0.f;
#pragma unroll
          for (int m = 0; m < N; ++m) { kk[m][t][r] = __builtin_amdgcn_exp2f(kk[m][t][r] - mx); z += kk[m][t][r]; }
          const float rz = fast_rcp(z);
#pragma unroll
          for (int m = 0; m < N; ++m) kk[m][t][r] *= rz;
        }
      // ---- q of every position: softmax over the head's 32 channels (8 in the lane, the rest in the other lane groups), * 32^-0.5
      f32x4 qs[N][2];
#pragma unroll
      for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < CPL; ++s) acc = mfma16(aq[t * CPL + s], xh[n][s], acc);
          qs[n][t] = acc;
        }
        float mx = 0.f;
        if (!bounded) {
          mx = qs[n][0][0];
#pragma unroll
          for (int e = 1; e < 8; ++e) mx = fmaxf(mx, qs[n][e >> 2][e & 3]);
          mx = gmax(mx);
        }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { qs[n][e >> 2][e & 3] = __builtin_amdgcn_exp2f(qs[n][e >> 2][e & 3] - mx); sum += qs[n][e >> 2][e & 3]; }
        sum = gsum(sum);
        const float sc = scale * fast_rcp(sum);
#pragma unroll
        for (int e = 0; e < 8; ++e) qs[n][e >> 2][e & 3] *= sc;
      }
      // ---- per position n, one row of S at a time (the whole N x N of S and dZ of all positions live at once cost the second wave per SIMD):
      //   S[n][m] = sum_d q[d][n] k[d][m] ;  Z[:, n] = sum_m S[n][m] xh[:, m] -> [row][channel slot] tile: dW2 += DY[:, n] Z[:, n]^T (K = rows) ;
      //   dZ[:, n] = W2^T DY[:, n] ;  dS[n][m] = sum_c dZ[c, n] xh[c, m] ;  d xh[:, m] += S[n][m] dZ[:, n]
      DQ_PSTAMP((500000 + C * 100 + N), 4);
      float dS[N][N];
      float aw2[4];
      { const float4 v = wp[4]; aw2[0] = v.x; aw2[1] = v.y; aw2[2] = v.z; aw2[3] = v.w; }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float Sn[N];
#pragma unroll
        for (int m = 0; m < N; ++m) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) t = fmaf(qs[n][e >> 2][e & 3], kk[m][e >> 2][e & 3], t);
          Sn[m] = gsum(t);
        }
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CPL; ++r)
#pragma unroll
          for (int m = 0; m < N; ++m) z[r] = fmaf(Sn[m], xh[m][r], z[r]);
        float* tz = (n & 1) ? t1 : t0;
        wsync();
        *reinterpret_cast<float4*>(tz + row * TP + 4 * g) = make_float4(z[0], z[1], z[2], z[3]);
        f32x4 dZ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) dZ = mfma16(aw2[s], DY[n][s], dZ);
        wsync();
#pragma unroll
        for (int s = 0; s < 4; ++s) gw = mfma16(ud[(n * 16 + row) * UP + 4 * s + g], tz[(4 * s + g) * TP + row], gw);
#pragma unroll
        for (int m = 0; m < N; ++m) {
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < CPL; ++r) { t = fmaf(dZ[r], xh[m][r], t); dxh[m][r] = fmaf(Sn[m], dZ[r], dxh[m][r]); }
          dS[n][m] = gsum(t);
        }
        if (N == 4) __builtin_amdgcn_sched_barrier(0);  // (interleaving the positions' chains costs the registers of the second wave per SIMD)
      }
      // ---- q side: dq, the q-softmax backward, d xh += Wq^T dql, dWq += dql xh^T ; T[d] = sum_n q dq (= sum_m k dk) for the k softmax
      DQ_PSTAMP((500000 + C * 100 + N), 5);
      f32x4 T[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      float aqt[8];
      ld8(5, aqt);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        f32x4 dq[2];
        float D = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int m = 0; m < N; ++m) v = fmaf(dS[n][m], kk[m][t][r], v);
            dq[t][r] = v;
            const float qd = qs[n][t][r] * v;
            D += qd;
            T[t][r] += qd;
          }
        D = gsum(D) * (1.0f / scale);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CPL; ++r) acc[r] = dxh[n][r];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dq[t][r] = qs[n][t][r] * (dq[t][r] - D);  // d loss / d (natural-log q logit)
            acc = mfma16(aqt[t * 4 + r], dq[t][r], acc);
          }
#pragma unroll
        for (int r = 0; r < CPL; ++r) dxh[n][r] = acc[r];
        // dql as [row][d] -> A operand of dWq (M = d, K = rows)
        float* tq = (n & 1) ? t1 : t0;
        wsync();
        *reinterpret_cast<float4*>(tq + row * TP + 4 * g) = make_float4(dq[0][0], dq[0][1], dq[0][2], dq[0][3]);
        *reinterpret_cast<float4*>(tq + row * TP + 16 + 4 * g) = make_float4(dq[1][0], dq[1][1], dq[1][2], dq[1][3]);
        wsync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float xt = ux[(n * 16 + row) * UP + 4 * s + g];  // xh[c = row][batch row 4 s + g] (read here, not held: registers)
          gq0 = mfma16(tq[(4 * s + g) * TP + row], xt, gq0);
          gq1 = mfma16(tq[(4 * s + g) * TP + 16 + row], xt, gq1);
        }
        if (N == 4) __builtin_amdgcn_sched_barrier(0);
      }
      // ---- k side: dk, the k-softmax backward (over the positions), d xh += Wk^T dkl, dWk += dkl xh^T
      DQ_PSTAMP((500000 + C * 100 + N), 6);
      float akt[8];
      ld8(7, akt);
#pragma unroll
      for (int m = 0; m < N; ++m) {
        f32x4 dk[2];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CPL; ++r) acc[r] = dxh[m][r];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) v = fmaf(dS[n][m], qs[n][t][r], v);
            dk[t][r] = kk[m][t][r] * (v - T[t][r]);
            acc = mfma16(akt[t * 4 + r], dk[t][r], acc);
          }
#pragma unroll
        for (int r = 0; r < CPL; ++r) dxh[m][r] = acc[r];
        float* tk = (m & 1) ? t1 : t0;
        wsync();
        *reinterpret_cast<float4*>(tk + row * TP + 4 * g) = make_float4(dk[0][0], dk[0][1], dk[0][2], dk[0][3]);
        *reinterpret_cast<float4*>(tk + row * TP + 16 + 4 * g) = make_float4(dk[1][0], dk[1][1], dk[1][2], dk[1][3]);
        wsync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float xt = ux[(m * 16 + row) * UP + 4 * s + g];
          gk0 = mfma16(tk[(4 * s + g) * TP + row], xt, gk0);
          gk1 = mfma16(tk[(4 * s + g) * TP + 16 + row], xt, gk1);
        }
        if (N == 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
    DQ_PSTAMP((500000 + C * 100 + N), 7);
    // ---- the four heads' d xh meet: [head][lane][RUN]
    {
      wsync();  // (this wave's last readers of its tiles are done)
      float* ex = t0 + lane * RUN;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        vecf o;
#pragma unroll
        for (int e = 0; e < VW; ++e) { const int i = k * VW + e; o[e] = dxh[i % N][i / N]; }  // (run order: element r N + n)
        *reinterpret_cast<vecf*>(ex + k * VW) = o;
      }
    }
    lds_barrier();
    DQ_PSTAMP((500000 + C * 100 + N), 8);
    if (hd == 3) {
      // ---- residual + PreNorm backward on the completed d xh (heads added in head order) ; dx (+)= dy + d/dx
      float tot_[RUN];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const vecf v0 = *reinterpret_cast<const vecf*>(tls[0] + lane * RUN + k * VW), v1 = *reinterpret_cast<const vecf*>(tls[1] + lane * RUN + k * VW);
        const vecf v2 = *reinterpret_cast<const vecf*>(tls[2] + lane * RUN + k * VW), v3 = *reinterpret_cast<const vecf*>(tls[3] + lane * RUN + k * VW);
#pragma unroll
        for (int e = 0; e < VW; ++e) tot_[k * VW + e] = ((v0[e] + v1[e]) + v2[e]) + v3[e];
      }
      // (x and dy come back from this wave's LDS stash instead of being held across the head's work: 2 RUN registers of every wave)
      float xr[RUN], dr[RUN];
      float gpre[CPL];
#pragma unroll
      for (int r = 0; r < CPL; ++r) gpre[r] = gains[CPL * g + r];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const vecf sx = *reinterpret_cast<const vecf*>(stash + lane * 2 * RUN + k * VW), sd = *reinterpret_cast<const vecf*>(stash + lane * 2 * RUN + RUN + k * VW);
#pragma unroll
        for (int e = 0; e < VW; ++e) { xr[k * VW + e] = sx[e]; dr[k * VW + e] = sd[e]; }
      }
      float out[RUN];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float ssq = 0.f;
#pragma unroll
        for (int r = 0; r < CPL; ++r) ssq = fmaf(xr[r * N + n], xr[r * N + n], ssq);
        ssq = gsum(ssq);
        const float nrm = fast_sqrt(ssq), pinv = fast_rcp(fmaxf(nrm, RMS_EPS));
        float dot = 0.f, tot[CPL], uh[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
          uh[r] = xr[r * N + n] * pinv;
          na[r] = fmaf(tot_[r * N + n], uh[r] * sqC, na[r]);  // d g_pre
          tot[r] = tot_[r * N + n] * gpre[r] * sqC;
          dot = fmaf(tot[r], uh[r], dot);
        }
        dot = gsum(dot);
        const bool clamped = nrm < RMS_EPS;
#pragma unroll
        for (int r = 0; r < CPL; ++r) out[r * N + n] = dr[r * N + n] + (clamped ? tot[r] * pinv : pinv * (tot[r] - uh[r] * dot));
      }
      if (live) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          vecf o;
          if (a.dx_store) {
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = out[k * VW + e];
          } else {
            const vecf p = *reinterpret_cast<const vecf*>(at(a.dx, k));
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = p[e] + out[k * VW + e];
          }
          *reinterpret_cast<vecf*>(const_cast<char*>(at(a.dx, k))) = o;
        }
      }
    }
  }

  DQ_PSTAMP((500000 + C * 100 + N), 9);
  // ---- flush: one slot per workgroup, layout la_slot(C) = dWq | dWk (256 C) | dW2 of the four heads (4 C C) | d g_out | d b_out | d g_pre;
  // each wave its head's sections
  float* slot = a.part + (int64_t)blockIdx.x * la_slot_floats(C);
  if (row < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d0 = 4 * g + r;
      slot[(hd * 32 + d0) * C + row] = gq0[r];
      slot[(hd * 32 + 16 + d0) * C + row] = gq1[r];
      slot[(128 + hd * 32 + d0) * C + row] = gk0[r];
      slot[(128 + hd * 32 + 16 + d0) * C + row] = gk1[r];
    }
  }
  {
    const int c = (row & 3) < CPL ? CPL * (row >> 2) + (row & 3) : -1;  // the channel of slot `row` of a [row][4 g + r] tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cp = 4 * g + r;
      if (cp < C && c >= 0) slot[256 * C + hd * C * C + cp * C + c] = gw[r];
    }
  }
  constexpr int GB = 256 * C + 4 * C * C;
  if (hd == 0 || hd == 3) {
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
      const float s0 = row16_sum(na[r]), s1 = row16_sum(nb[r]);
      if (row == 0) {
        const int c = CPL * g + r;
        if (hd == 0) { slot[GB + c] = s0; slot[GB + C + c] = s1; }
        else slot[GB + 2 * C + c] = s0;
      }
    }
  }
  DQ_PSTAMP((500000 + C * 100 + N), 10);
}

}  // namespace

bool la_rows_bwd_usable(int C, int n) { return (n == 2 || n == 4) && (C == 8 || C == 12 || C == 16); }
int la_rows_bwd_min_rows() {
  const int64_t o = option(OPT_LA_ROWS_BWD_MIN_ROWS);
  return o < 0 ? 0 : (int)std::min<int64_t>(o, INT32_MAX);
}

// Launches the backward and reports the number of slots written (one per workgroup; at most max_slots).
int launch_la_rows_bwd(const LinAttnBwd& a, int max_slots, int* slots_out, hipStream_t s) {
  const int C = a.f.C, n = a.f.n, rows = a.f.rows;
  DQ_REQUIRE(a.f.x && a.ypre && a.dy && a.dx && a.f.prep && a.f.g_pre && a.f.g_out && a.part && la_rows_bwd_usable(C, n), "la_rows_bwd: missing operand / unsupported shape");
  DQ_REQUIRE((((uintptr_t)a.f.prep | (uintptr_t)a.f.x | (uintptr_t)a.ypre | (uintptr_t)a.dy | (uintptr_t)a.dx) & 15) == 0, "la_rows_bwd: misaligned tensor / prepared-weights buffer");
  DQ_REQUIRE(max_slots >= 1, "la_rows_bwd: slot scratch too small");
  DQ_REQUIRE((int64_t)rows * C * n * 4 < (1ll << 32), "la_rows_bwd: tensors of 4 GB or more are not built (32-bit byte offsets)");
  const int ntiles = cdiv(rows, 16);
  static const int cus = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
  LaRowsBwdK k{a.f.x, a.ypre, a.dy, a.dx, a.f.prep, a.f.g_pre, a.f.g_out, a.part, rows, ntiles, a.dx_store};
#define DQ_LRB(CC, NN)                                                                                   \
  if (C == CC && n == NN) {                                                                              \
    const int nb = occ_blocks_per_cu((const void*)k_la_rows_bwd<CC, NN>, 256, 0);                        \
    if (nb < 0) return 1;                                                                                \
    const int grid = std::max(1, std::min(ntiles, std::min(max_slots, nb * cus)));  /* one resident round, a slot per workgroup */ \
    hipLaunchKernelGGL((k_la_rows_bwd<CC, NN>), dim3(grid), dim3(256), 0, s, k);                         \
    DQ_LAUNCH_CHECK();                                                                                   \
    *slots_out = grid;                                                                                   \
    return 0;                                                                                            \
  }
  DQ_LRB(8, 2) DQ_LRB(8, 4) DQ_LRB(12, 2) DQ_LRB(12, 4) DQ_LRB(16, 2) DQ_LRB(16, 4)
#undef DQ_LRB
  set_error("la_rows_bwd: unsupported (C, n)");
  return 2;
}

}  // namespace dq
