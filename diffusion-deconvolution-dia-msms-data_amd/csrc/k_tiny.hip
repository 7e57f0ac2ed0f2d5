// The levels whose m/z rows are 1 or 2 positions long (reference dquartic/model/unet1d.py:1134-1142, 1150-1158 at the bottom of the
// pyramid): a row's whole state is n * C <= 32 floats, so every convolution over it is a DENSE layer on that vector -- a k3 conv over two
// positions sees both of them from either one, a k3 conv over one position is its centre tap, Downsample / Upsample are (2 C_in x C)
// matrices.  The level then is a chain of small matrix products over rows, and runs on v_mfma_f32_32x32x2_f32 with
//   lane = (half, slot):  n == 2: half = position, slot = row of the tile (32 rows per wave)
//                         n == 1: half * 32 + slot = row of the tile (64 rows per wave)
//   register c = channel c of the lane's (row, position)                                  (16 registers, zero above C)
// This IS the MFMA accumulator layout (dq_mfma.h: register r of half h holds output row (r & 3) + 8 (r >> 2) + 4 h) when output row m
// is read as (position (m >> 2) & 1, channel (m & 3) + 4 (m >> 3)); and register c of the two halves is one K = 2 slice of the B operand.
// A layer's output tile therefore feeds the next layer register by register -- no data movement between layers -- and everything that
// acts per position (RMSNorm over the channels, scale / shift, SiLU, the residual) is plain per-lane arithmetic over 16 registers.
// The A operands (weights placed by the position structure of each layer; zero where a tap does not exist) are built once per
// parameter state by k_tiny_images and copied to LDS by every workgroup.
//
// One launch = [resample conv producing the level's input] -> ResnetBlock -> ResnetBlock
//              (-> Residual(PreNorm(LinearAttention)) at n == 1, where it is linear: k.softmax over one position is 1, q.softmax
//                  sums to 1, so out = scale * Wo Wv xhat + b, unet1d.py:466-496)
//              (-> the last down level's k3 conv, written straight in the bottleneck's (B, C, RT) layout, unet1d.py:1144-1148)
// replacing k_level_fwd + k_linattn_fwd (+ k_conv_fwd + k_fold) at those levels: at batch 32 they were latency floors of 20-34 us each
// (one 64-position wave per tile walking ~1,000 dependent 4x4x1 MFMAs at one wave per SIMD).
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include "dq_plan.h"
#include "dq_probe.h"
#include "k_res_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace dq {

namespace {

enum { TL_CONV3_N2 = 0, TL_DIAG = 1, TL_DOWN_N2 = 2, TL_DOWN_N1 = 3, TL_UP_N2 = 4, TL_LA1 = 5,
       TL_DIAG_T = 6, TL_LA1_T = 7, TL_DOWN_N1_T = 8, TL_UP_N2_T = 9 };  // transposed forms of the backward data path (n == 1): step = output channel of the conv
// One dense layer of the chain = `steps` MFMAs.  Element (step s, lane) of its operand image:
//   m = lane & 31 -> output (position pm = (m >> 2) & 1, channel co = (m & 3) + 4 (m >> 3)); k = lane >> 5 = the half that supplies B
struct TinyLayer {
  int kind;
  int w;      // weight offset in the flat parameter buffer (TL_LA1: to_qkv)
  int w2;     // TL_LA1: to_out.0.weight
  int cin;    // input channels of the conv (row pitch of the weight tensor)
  int K;      // kernel width of the weight tensor
  int c0;     // first input channel this layer's steps walk (the skip half of cat(x, skip) starts at C)
  int steps, cout;
};
constexpr int TINY_MAX_LAYERS = 14;
struct TinyImgItem { TinyLayer L[TINY_MAX_LAYERS]; int nl, total_steps; float* dst; };
struct TinyImgMulti { TinyImgItem it[TINY_IMG_MAX]; };

__device__ __forceinline__ float tiny_img_value(const TinyLayer& L, int s, int lane, const float* __restrict__ P) {
  const int m = lane & 31, k = lane >> 5;
  const int pm = (m >> 2) & 1, co = (m & 3) + 4 * (m >> 3);
  if (co >= L.cout) return 0.f;
  switch (L.kind) {
    case TL_CONV3_N2: {  // k3 over two positions: input position k reaches output position pm through tap k - pm + 1
      const int ci = L.c0 + s, t = k - pm + 1;
      return P[L.w + (co * L.cin + ci) * 3 + t];
    }
    case TL_DIAG: {  // per (row, position): the centre tap of a k3 conv over one position, or a 1x1 conv
      const int ci = L.c0 + s;
      return pm == k ? P[L.w + (co * L.cin + ci) * L.K + L.K / 2] : 0.f;
    }
    case TL_DOWN_N2: {  // k4 s2 p1, 4 -> 2 positions: half k holds in[2 k], in[2 k + 1]; step = (ci, j)
      const int ci = s >> 1, j = s & 1, t = 2 * k + j - 2 * pm + 1;
      return (t >= 0 && t < 4) ? P[L.w + (co * L.cin + ci) * 4 + t] : 0.f;
    }
    case TL_DOWN_N1: {  // k4 s2 p1, 2 -> 1 positions: out = w[1] in[0] + w[2] in[1]
      const int ci = s >> 1, j = s & 1;
      return pm == k ? P[L.w + (co * L.cin + ci) * 4 + j + 1] : 0.f;
    }
    case TL_UP_N2: {  // nearest x2 of one position, then k3: out[0] = (w1 + w2) in, out[1] = (w0 + w1) in, as two groups of steps
      const int g = s / L.cin, ci = s - g * L.cin;
      const int t = g == 0 ? 1 : (pm == 0 ? 2 : 0);
      return pm == k ? P[L.w + (co * L.cin + ci) * 3 + t] : 0.f;
    }
    case TL_DIAG_T: {  // d in[ci = idx] = sum_co W[co = s][c0 + ci][centre] d out[co]; `cout` = number of valid ci
      return pm == k ? P[L.w + (s * L.cin + L.c0 + co) * L.K + L.K / 2] : 0.f;
    }
    case TL_DOWN_N1_T: {  // d in[ci][j] = sum_co W[co = s][ci][1 + j] d out[co]; j = L.w2
      return pm == k ? P[L.w + (s * L.cin + co) * 4 + 1 + L.w2] : 0.f;
    }
    case TL_UP_N2_T: {  // nearest x2 + k3, transposed: d in[ci] = sum_co W_p[co = s][ci] d out[co][p], W_0 = w1 + w2, W_1 = w0 + w1 (TL_UP_N2); p = L.w2
      const float* q = P + L.w + (s * L.cin + co) * 3;
      return pm == k ? q[1] + (L.w2 == 0 ? q[2] : q[0]) : 0.f;
    }
    case TL_LA1_T: {  // d xhat[ci = idx] = scale * sum_c' (Wo Wv)[c' = s][ci] d ypre[c']
      if (pm != k) return 0.f;
      float acc = 0.f;
      for (int j = 0; j < HID; ++j) acc = fmaf(P[L.w2 + s * HID + j], P[L.w + (2 * HID + j) * L.cin + co], acc);
      return acc * 0.17677669529663687f;
    }
    default: {  // TL_LA1: scale * (Wo Wv)[co][ci]  (to_qkv (3 HID, C, 1): the v rows start at 2 HID; to_out.0 (C, HID, 1))
      if (pm != k) return 0.f;
      const int ci = s;
      float acc = 0.f;
      for (int j = 0; j < HID; ++j) acc = fmaf(P[L.w2 + co * HID + j], P[L.w + (2 * HID + j) * L.cin + ci], acc);
      return acc * 0.17677669529663687f;  // 32^-0.5
    }
  }
}

__global__ void __launch_bounds__(256) k_tiny_images(TinyImgMulti mm, const float* __restrict__ P) {
  const TinyImgItem& it = mm.it[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= it.total_steps * 64) return;
  int s = idx >> 6, l = 0;
  while (l + 1 < it.nl && s >= it.L[l].steps) { s -= it.L[l].steps; ++l; }
  it.dst[idx] = tiny_img_value(it.L[l], s, idx & 63, P);
}

struct TinyBlkK {
  int b1, g1, b2, g2, br;  // offsets in P (br < 0: identity residual)
  int ss_off;
  const float* inB; float* u1; float* a1; float* u2; float* out;
};
struct TinyFwdK {
  const float* in; float* pre_out; float* in_copy;
  int pb, rows_per_sample, ss_stride, in_folded, total_steps;
  TinyBlkK blk[2];
  int la, la_gpre, la_bo, la_go; float* la_y; float* la_ypre;
  int post, post_b; float* post_out;
};

__host__ __device__ constexpr int tiny_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
__host__ __device__ constexpr int tiny_stg_floats(int N, int e_in, int C, int CS) { return N == 1 ? 0 : (64 / N) * (tiny_max3(e_in, C * N, CS * N) + 1); }
__host__ __device__ constexpr int tiny_rounds(int steps) { return (steps * 16 + 255) / 256; }
constexpr int NPRM = 20;  // 16-float parameter vectors in LDS (the table is padded to 32 vectors: two unguarded stores per thread): stage bias | per block: b1 g1 b2 g2 br scale+1 shift | g_pre b_out g_out | post bias

// C: the level's channels; N: row length; PRE: input stage; CP: its input channels; CS: skip channels of cat(x, skip) (0: none, identity residual)
template <int C, int N, int PRE, int CP, int CS>
__global__ void __launch_bounds__(256) k_tiny_fwd(TinyFwdK a, const float* __restrict__ P, const float* __restrict__ ssb, const float* __restrict__ img,
                                                 int tiles_ps) {
  constexpr int RPT = 64 / N;
  constexpr bool WR = CS > 0;
  constexpr int S_PRE = (PRE == LEVEL_PRE_DOWN || PRE == LEVEL_PRE_UP) ? 2 * CP : 0;
  constexpr int S_C1 = C + CS, S_C2 = C, S_RS = WR ? C + CS : 0;
  constexpr int BLK = S_C1 + S_C2 + S_RS;
  constexpr int B_LA = S_PRE + 2 * BLK, B_POST = B_LA + C;
  constexpr int PID = 300000 + C * 1000 + N * 100 + PRE * 10 + (CS ? 1 : 0);  // tools/probe_step.py id
  DQ_PSTAMP(PID, 0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NLD = tiny_rounds(B_POST + C);       // 16-byte loads per thread that cover the largest image of this instantiation
  float* wl = lds;                                  // [total_steps][64], padded to NLD * 1024 floats
  float* prm = lds + NLD * 1024;                    // [NPRM][16]
  // wave-private staging tile (RPT rows x (E + 1) floats): global tensors are read and written as contiguous runs of whole rows (4 bytes per
  // lane, 256 bytes per instruction) and transposed to / from the lane = (position, row) layout here.  As direct accesses every instruction
  // touched 32 cache lines with 4..8 bytes each: the 2-position levels spent 170 clocks per store instruction (probe), 64 of them per block.
  constexpr int E_IN = PRE == LEVEL_PRE_DOWN ? CP * 2 * N : (PRE == LEVEL_PRE_UP ? CP : 0);
  constexpr int STG = tiny_stg_floats(N, E_IN, C, CS);
  float* stg = prm + 32 * 16 + (threadIdx.x >> 6) * STG;
  // Preamble in ONE memory round trip: every 16-byte load of the image and the thread's (<= 2) parameter values are requested before the
  // first LDS store (as a load -> store loop plus a branchy parameter gather it was three to four dependent round trips, ~5 us of a
  // ~16 us launch at the training batch).
  const int b = blockIdx.y;
  {
    // (no guards: the image slot and the LDS region are padded to whole rounds of 256 x 16 bytes -- a guarded store made the compiler sink
    // each load under its store's branch, i.e. one memory round trip per round: 14,000 clocks of a 60,000-clock launch)
    // the thread's two parameter values (vectors 0..15 and 16..31 of the padded table): one unconditional load from P and one from the
    // sample's scale / shift vector each, selected after the barrier below
    int offs[2], soffs[2], kk[2];
    bool hasS[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int what = (i >> 4) < NPRM ? (i >> 4) : 19, c = i & 15;
      const int bi = what >= 8 ? 1 : 0, k = (what >= 1 && what <= 14) ? (what - 1) % 7 : 7;
      const int b1 = bi ? a.blk[1].b1 : a.blk[0].b1, g1 = bi ? a.blk[1].g1 : a.blk[0].g1, b2 = bi ? a.blk[1].b2 : a.blk[0].b2;
      const int g2 = bi ? a.blk[1].g2 : a.blk[0].g2, br = bi ? a.blk[1].br : a.blk[0].br, sso = bi ? a.blk[1].ss_off : a.blk[0].ss_off;
      int off = k == 0 ? b1 : k == 1 ? g1 : k == 2 ? b2 : k == 3 ? g2 : (k == 4 && WR) ? br : -1;
      off = what == 0 ? (S_PRE ? a.pb : -1) : off;
      off = what == 15 ? (a.la ? a.la_gpre : -1) : what == 16 ? (a.la ? a.la_bo : -1) : what == 17 ? (a.la ? a.la_go : -1) : off;
      off = what == 18 ? (a.post ? a.post_b : -1) : off;
      const bool inC = c < C;
      kk[u] = k;
      offs[u] = (off >= 0 && inC) ? off + c : -1;
      hasS[u] = (k == 5 || k == 6) && inC;  // (an offset into the scale / shift vector is relative to block 0's and may be negative)
      soffs[u] = hasS[u] ? (k == 5 ? sso : sso + C) + c : 0;
    }
    float pv[2], sv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pv[u] = P[offs[u] >= 0 ? offs[u] : 0];
      sv[u] = ssb[(int64_t)b * a.ss_stride + soffs[u]];
    }
    const float4* src = reinterpret_cast<const float4*>(img);
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = src[u * 256 + (int)threadIdx.x];
    __builtin_amdgcn_sched_barrier(0);  // every load of the preamble is in flight here
#pragma unroll
    for (int u = 0; u < NLD; ++u) reinterpret_cast<float4*>(wl)[u * 256 + (int)threadIdx.x] = v[u];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      prm[u * 256 + (int)threadIdx.x] = hasS[u] ? (kk[u] == 5 ? sv[u] + 1.0f : sv[u]) : (offs[u] >= 0 ? pv[u] : 0.f);
  }
  __syncthreads();
  DQ_PSTAMP(PID, 1);
  const int lane = threadIdx.x & 63, half = lane >> 5, slot = lane & 31;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int RT = a.rows_per_sample;
  const float sqC = sqrtf((float)C);
  const float* wlane = wl + lane;
  auto prmv = [&](int what, int c) -> float { return prm[what * 16 + c]; };
  // one dense layer: STEPS MFMAs, B operand = breg[0 .. STEPS) in image order
  auto dense = [&](int base, const float* breg, auto steps_c, f32x16 acc) __attribute__((always_inline)) -> f32x16 {
    constexpr int STEPS = decltype(steps_c)::value;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc = mfma_f32(wlane[(base + s) * 64], breg[s], acc);
    return acc;
  };
  const int rowl = N == 2 ? slot : lane, p = N == 2 ? half : 0;

#pragma unroll 1
  for (int tile = wid; tile < tiles_ps; tile += nwaves) {
    const int r_in_s = tile * RPT + rowl;
    const bool live = r_in_s < RT;
    const int rs_c = live ? r_in_s : RT - 1;
    const int64_t row = (int64_t)b * RT + rs_c;
    const int valid = RT - tile * RPT < RPT ? RT - tile * RPT : RPT;          // rows of this tile that exist
    const int64_t trow = (int64_t)b * RT + (int64_t)tile * RPT;                // the tile's first row
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // (rows, CH, N) tensors, E = CH * N floats per row.  lin_load: the tile's E * RPT floats as E * RPT / 64 contiguous 256-byte reads
    // (rows beyond the sample re-read its last element); to_lanes: through the staging tile into dst[c] = element (row, c, p) of the lane
    auto lin_load = [&](const float* base, auto e_c, float* raw) __attribute__((always_inline)) {
      constexpr int E = decltype(e_c)::value;
      const float* g = base + trow * E;
      const int lim = valid * E;
#pragma unroll
      for (int k = 0; k < E * RPT / 64; ++k) { const int lin = k * 64 + lane; raw[k] = g[lin < lim ? lin : lim - 1]; }
    };
    auto lin_to_stg = [&](auto e_c, const float* raw) __attribute__((always_inline)) {
      constexpr int E = decltype(e_c)::value;
      wsync();
#pragma unroll
      for (int k = 0; k < E * RPT / 64; ++k) { const int lin = k * 64 + lane; stg[lin + lin / E] = raw[k]; }
      wsync();
    };
    auto to_lanes = [&](auto ch_c, const float* raw, float* dst) __attribute__((always_inline)) {
      constexpr int CH = decltype(ch_c)::value, E = CH * N;
      if constexpr (N == 1) {  // (lin_load_rows read the lane's own row: nothing to transpose)
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c] = c < CH ? raw[c] : 0.f;
        return;
      }
      lin_to_stg(std::integral_constant<int, E>{}, raw);
#pragma unroll
      for (int c = 0; c < 16; ++c) dst[c] = c < CH ? stg[rowl * (E + 1) + c * N + p] : 0.f;
    };
    // direct form (the LEVEL_PRE_NONE input when it is not the bottleneck's layout)
    auto ldrow = [&](const float* base, int CH, float* dst, int cnt) __attribute__((always_inline)) {
      const float* q = base + (row * CH) * N + p;
#pragma unroll
      for (int c = 0; c < 16; ++c) dst[c] = c < cnt ? q[c * N] : 0.f;
    };
    auto strow = [&](float* base, const float* v) __attribute__((always_inline)) {
      if constexpr (N == 1) {  // lane = row: neighbouring lanes share cache lines already, and the staged form measured slower (5,200 -> 10,100 clocks per block)
        if (base && live) {
          float* q = base + row * C;
#pragma unroll
          for (int c = 0; c < C; ++c) q[c] = v[c];
        }
      } else if (base) {  // (wave-uniform)
        constexpr int E = C * N;
        wsync();
#pragma unroll
        for (int c = 0; c < C; ++c) stg[rowl * (E + 1) + c * N + p] = v[c];
        wsync();
        float* g = base + trow * E;
        const int lim = valid * E;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int lin = k * 64 + lane;
          const float val = stg[lin + lin / E];
          if (lin < lim) g[lin] = val;
        }
      }
    };
    float x[16], xb0[16], xb1[16];
    // ---------------------------------------------------------------- input stage (every global read of the tile is requested up front)
    if constexpr (PRE == LEVEL_PRE_DOWN) {
      // input (rows, CP, 2 N): the lane holds in[ci][2 p], in[ci][2 p + 1]
      static_assert(PRE != LEVEL_PRE_DOWN || CS == 0, "a Downsample stage in front of blocks with skip channels is not built");
      float v[2 * CP], raw[2 * CP];
      if constexpr (N == 1) {
        const float* q = a.in + row * (CP * 2);
#pragma unroll
        for (int ci = 0; ci < CP; ++ci) { const float2 t = *reinterpret_cast<const float2*>(q + ci * 2); v[2 * ci] = t.x; v[2 * ci + 1] = t.y; }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        lin_load(a.in, std::integral_constant<int, E_IN>{}, raw);
        __builtin_amdgcn_sched_barrier(0);
        lin_to_stg(std::integral_constant<int, E_IN>{}, raw);
#pragma unroll
        for (int ci = 0; ci < CP; ++ci) {
          v[2 * ci] = stg[rowl * (E_IN + 1) + ci * 2 * N + 2 * p];
          v[2 * ci + 1] = stg[rowl * (E_IN + 1) + ci * 2 * N + 2 * p + 1];
        }
      }
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = dense(0, v, std::integral_constant<int, 2 * CP>{}, acc);
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = c < C ? acc[c] + prmv(0, c) : 0.f;
      strow(a.pre_out, x);
    } else if constexpr (PRE == LEVEL_PRE_UP) {
      // input (rows, CP, 1): both positions of the row read the row's one input position
      float v[2 * CP], raw[CP * RPT / 64], rb0[CS > 0 ? CS : 1], rb1[CS > 0 ? CS : 1];
      lin_load(a.in, std::integral_constant<int, E_IN>{}, raw);
      if constexpr (CS > 0) { lin_load(a.blk[0].inB, std::integral_constant<int, CS * N>{}, rb0); lin_load(a.blk[1].inB, std::integral_constant<int, CS * N>{}, rb1); }
      __builtin_amdgcn_sched_barrier(0);  // (every global read of the tile is in flight before the first product)
      lin_to_stg(std::integral_constant<int, E_IN>{}, raw);
#pragma unroll
      for (int ci = 0; ci < CP; ++ci) { v[ci] = stg[rowl * (E_IN + 1) + ci]; v[CP + ci] = v[ci]; }
      if constexpr (CS > 0) { to_lanes(std::integral_constant<int, CS>{}, rb0, xb0); to_lanes(std::integral_constant<int, CS>{}, rb1, xb1); }
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = dense(0, v, std::integral_constant<int, 2 * CP>{}, acc);
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = c < C ? acc[c] + prmv(0, c) : 0.f;
      strow(a.pre_out, x);
    } else {
      if (a.in_folded) {  // (B, C, RT): the bottleneck's layout (N == 1)
        const float* q = a.in + (int64_t)b * C * RT + rs_c;
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = c < C ? q[(int64_t)c * RT] : 0.f;
      } else {
        ldrow(a.in, C, x, C);
      }
      float rb0[CS > 0 ? CS : 1], rb1[CS > 0 ? CS : 1];
      if constexpr (CS > 0) {
        if constexpr (N == 1) {
          const float* q0 = a.blk[0].inB + row * CS;
          const float* q1 = a.blk[1].inB + row * CS;
#pragma unroll
          for (int c = 0; c < CS; ++c) { rb0[c] = q0[c]; rb1[c] = q1[c]; }
        } else {
          lin_load(a.blk[0].inB, std::integral_constant<int, CS * N>{}, rb0); lin_load(a.blk[1].inB, std::integral_constant<int, CS * N>{}, rb1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CS > 0) { to_lanes(std::integral_constant<int, CS>{}, rb0, xb0); to_lanes(std::integral_constant<int, CS>{}, rb1, xb1); }
      strow(a.in_copy, x);
    }
    DQ_PSTAMP(PID, 2);
    // ---------------------------------------------------------------- the level's ResnetBlocks (unet1d.py:302-323)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const TinyBlkK& r = a.blk[bi];
      const int base = S_PRE + bi * BLK, pq = 1 + 7 * bi;
      const float* xs = bi == 0 ? xb0 : xb1;
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = dense(base, x, std::integral_constant<int, C>{}, acc);
      if constexpr (CS > 0) acc = dense(base + C, xs, std::integral_constant<int, CS>{}, acc);
      float u[16], h[16];
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { u[c] = acc[c] + prmv(pq + 0, c); ssq = fmaf(u[c], u[c], ssq); }
      strow(r.u1, u);
      {
        const float inv = rms_inv(ssq, sqC);
#pragma unroll
        for (int c = 0; c < 16; ++c) h[c] = c < C ? silu_f(fmaf(u[c] * inv * prmv(pq + 1, c), prmv(pq + 5, c), prmv(pq + 6, c))) : 0.f;
      }
      strow(r.a1, h);
      f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc2 = dense(base + S_C1, h, std::integral_constant<int, C>{}, acc2);
      ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { u[c] = acc2[c] + prmv(pq + 2, c); ssq = fmaf(u[c], u[c], ssq); }
      strow(r.u2, u);
      float res[16];
      if constexpr (WR) {
        f32x16 ar = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        ar = dense(base + S_C1 + S_C2, x, std::integral_constant<int, C>{}, ar);
        ar = dense(base + S_C1 + S_C2 + C, xs, std::integral_constant<int, CS>{}, ar);
#pragma unroll
        for (int c = 0; c < C; ++c) res[c] = ar[c] + prmv(pq + 4, c);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) res[c] = x[c];
      }
      {
        const float inv = rms_inv(ssq, sqC);
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = c < C ? silu_f(u[c] * inv * prmv(pq + 3, c)) + res[c] : 0.f;
      }
      strow(r.out, x);
      DQ_PSTAMP(PID, 3 + bi);
    }
    if constexpr (N == 1) {
      // ------------------------------------------------------------ Residual(PreNorm(LinearAttention)) over one position: linear
      if (a.la) {
        float xh[16];
        float ssq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) ssq = fmaf(x[c], x[c], ssq);
        const float inv = rms_inv(ssq, sqC);
#pragma unroll
        for (int c = 0; c < 16; ++c) xh[c] = c < C ? x[c] * inv * prmv(15, c) : 0.f;
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = dense(B_LA, xh, std::integral_constant<int, C>{}, acc);
        float yp[16];
        ssq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { yp[c] = acc[c] + prmv(16, c); ssq = fmaf(yp[c], yp[c], ssq); }
        strow(a.la_ypre, yp);
        const float inv2 = rms_inv(ssq, sqC);
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = c < C ? fmaf(yp[c] * inv2, prmv(17, c), x[c]) : 0.f;
        strow(a.la_y, x);
      }
      // ------------------------------------------------------------ the last down level's k3 conv, stored as (B, C, RT)
      if (a.post) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = dense(B_POST, x, std::integral_constant<int, C>{}, acc);
        if (live) {
          float* q = a.post_out + (int64_t)b * C * RT + rs_c;
#pragma unroll
          for (int c = 0; c < C; ++c) q[(int64_t)c * RT] = acc[c] + prmv(18, c);
        }
      }
    }
    DQ_PSTAMP(PID, 5);
  }
  DQ_PSTAMP(PID, 6);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Backward DATA path of a level with rows of ONE position, one launch (autograd of the chain above; reference: unet1d.py:302-323, 446-496,
// 1134-1158):  [the last down level's k3 conv] -> Residual(PreNorm(LinearAttention)) (linear at n = 1) -> ResnetBlock 1 -> ResnetBlock 0
// -> [Downsample that produced the level's input], every transposed conv a chain of dense layers exactly as in the forward (lane = row,
// register = channel), everything per position in-lane.  It writes what the weight-gradient kernels on the side stream read (d u1, d u2 of
// both blocks, d rs), the skip gradients, the level's input gradient (the first up level: straight into the bottleneck's (B, C, RT)
// layout), the blocks' [d g2 | d g1 | d scale | d shift] sums per workgroup (ResBwd::gpart format) and ONE LinearAttention slot per
// workgroup in k_linattn_bwd1's format (d Wq = d Wk = 0 at one position; d W2 the same for the four heads).  d W2 = sum_rows dYpre xhat^T
// runs over the lanes: both operands go through a wave-private LDS tile [c][row] and 32 MFMAs per tile.
// Replaces k_fold + k_conv_bwd_wg + k_linattn_bwd1 + 2 k_res_bwd_cp (+ the previous level's k_conv_bwd_wg data path) per level.
struct TinyBwdBlkK {
  int g1, g2, ss_off;                  // gains (offsets in P), this block's [scale | shift] in the sample's vector
  const float* u1; const float* u2;    // saved pre-norm conv outputs (rows, C)
  float* du1; float* du2;              // their gradients (written: the weight-gradient kernels read them)
  float* dB; int dB_acc;               // gradient of the skip input (rows, CS): stored, or accumulated into
  float* dout_st;                      // nullable: the block's OUTPUT gradient (rows, C), stored for the res_conv weight gradient
  float* gpart;                        // [b * gridDim.x + x][4 C] = [d g2 | d g1 | d scale | d shift]
};
struct TinyBwdK {
  const float* x; const float* ypre; const float* dy;  // r1.out (forward), la_pre (forward), d la (rows, C); with a post conv: its skip part
  int la_gpre, la_go;
  float* la_part;                                      // [workgroup][la_slot(C)]
  const float* dmid; float* drs_out;                   // post conv: d mid_in (B, C, RT); the same as (rows, C) for its weight gradient
  float* dfold;                                        // LEVEL_PRE_NONE: the level's input gradient as (B, C, RT)
  float* din_rows; float* dprev;                       // LEVEL_PRE_DOWN: the input gradient as (rows, C); the Downsample's data gradient += (rows, CP, 2)
  const float* r0out_g;                                // nullable: d r0.out already holds a skip gradient (down path): added to block 0's d out
  const float* dup;                                    // UPT: d rs (rows, C, 2) of the Upsample conv behind the level
  TinyBwdBlkK blk[2];                                  // [0] = ResnetBlock 0, [1] = ResnetBlock 1
  int rows_per_sample, ss_stride;
};
constexpr int tiny_la_slot(int C) { return 256 * C + 4 * C * C + 3 * C; }  // = la_slot(C) of k_la_bwd.hip

// UPT: the Upsample conv behind the level (its backward data path in front of everything else)
template <int C, int PRE, int CP, int CS, bool UPT>
__global__ void __launch_bounds__(256) k_tiny_bwd(TinyBwdK a, const float* __restrict__ P, const float* __restrict__ ssb, const float* __restrict__ img,
                                                 int tiles_ps) {
  static_assert(C == 16, "built for 16 channels");
  constexpr bool WR = CS > 0, POST = PRE == LEVEL_PRE_DOWN;
  constexpr int BLKT = 2 * C + (CS ? C : 0) + (WR ? 2 * C : 0);
  constexpr int B_POST = 0, B_LA = POST ? C : 0, B_R1 = B_LA + C, B_R0 = B_R1 + BLKT, B_ST = B_R0 + BLKT;
  constexpr int B_UP = B_ST + (PRE == LEVEL_PRE_DOWN ? 2 * C : 0);
  constexpr int TOTAL = B_UP + (UPT ? 2 * C : 0);
  constexpr int NLD = tiny_rounds(TOTAL);
  constexpr int NACC = 3 * C + 8 * C;  // per-lane sums: d g_out | d b_out | d g_pre | per block d g2 | d g1 | d scale | d shift
  constexpr int PIDB = 400000 + PRE * 10 + (CS ? 1 : 0);  // tools/probe_step.py id
  DQ_PSTAMP(PIDB, 0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                       // [TOTAL][64] padded to NLD rounds
  float* prm = lds + NLD * 1024;         // [16 vectors][16]: g_out | g_pre | blk0: g1 g2 scale shift | blk1: g1 g2 scale shift
  float* tr = prm + 256 + (threadIdx.x >> 6) * (2 * C * 65);   // per wave: dYpre | scale-free xhat as [c][row]
  float* comb = prm + 256 + 4 * (2 * C * 65);                  // [4 waves][4 lane rows][NACC], then [4 waves][C * C]
  float* w2c = comb + 16 * NACC;
  const int b = blockIdx.y;
  {
    const int i = threadIdx.x;
    const int what = i >> 4, c = i & 15;
    const int bi = what >= 6 ? 1 : 0, k = what >= 2 ? (what - 2) & 3 : 4;   // k: 0 g1, 1 g2, 2 scale, 3 shift
    const int g1o = bi ? a.blk[1].g1 : a.blk[0].g1, g2o = bi ? a.blk[1].g2 : a.blk[0].g2, sso = bi ? a.blk[1].ss_off : a.blk[0].ss_off;
    const bool isS = what >= 2 && what < 10 && (k == 2 || k == 3);
    int offP = what == 0 ? a.la_go : what == 1 ? a.la_gpre : (what < 10 ? (k == 0 ? g1o : k == 1 ? g2o : -1) : -1);
    const int offS = isS ? sso + (k == 3 ? C : 0) + c : 0;
    const float pv = P[offP >= 0 ? offP + c : 0];
    const float sv = ssb[(int64_t)b * a.ss_stride + offS];
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = reinterpret_cast<const float4*>(img)[u * 256 + (int)threadIdx.x];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) reinterpret_cast<float4*>(wl)[u * 256 + (int)threadIdx.x] = v[u];
    prm[i] = isS ? sv : (offP >= 0 ? pv : 0.f);
  }
  __syncthreads();
  DQ_PSTAMP(PIDB, 1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv), nwaves = gridDim.x * 4;
  const int RT = a.rows_per_sample;
  const float sqC = 4.0f;  // sqrt(16)
  const float* wlane = wl + lane;
  // (scheduling fences around every layer: left alone the scheduler hoists the operand reads of ALL ~180 layers' steps to the top of the tile --
  // 512 + 256 registers and 60-110 spilled ones)
  auto dense = [&](int base, const float* breg, f32x16 acc) __attribute__((always_inline)) -> f32x16 {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < C; ++s) acc = mfma_f32(wlane[(base + s) * 64], breg[s], acc);
    __builtin_amdgcn_sched_barrier(0);
    return acc;
  };
  auto zero16 = []() __attribute__((always_inline)) -> f32x16 { f32x16 z; for (int r = 0; r < 16; ++r) z[r] = 0.f; return z; };
  auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  // Sums over the rows (norm gains, bias, per-sample d scale / d shift): 176 values per tile.  Kept per lane across the tiles they cost 176
  // registers (the kernel sat at 512 + 256 with ~100 spilled); added into an LDS table value by value they were 64 dependent LDS round trips
  // per block (21,700 clocks per block, probe).  Now: four DPP adds give every lane its 16-lane row's sum of value i, and lane (i mod 16) of
  // each row keeps the running sum of value i -- 11 registers per lane, two VALU instructions per value, a fixed order, no LDS in the loop.
  float racc[NACC / 16];
#pragma unroll
  for (int i = 0; i < NACC / 16; ++i) racc[i] = 0.f;
  auto flush_vals = [&](const float* v, auto n_c, auto off_c) __attribute__((always_inline)) {
    constexpr int NV = decltype(n_c)::value, OFF = decltype(off_c)::value;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float t = v[i];
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x141, 0xF, 0xF, false));  // row_half_mirror
      t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x140, 0xF, 0xF, false));  // row_mirror
      racc[(OFF + i) / 16] += (lane & 15) == ((OFF + i) & 15) ? t : 0.f;
      if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from carrying dozens of row sums around)
    }
  };
  f32x16 w2acc = zero16();

#pragma unroll 1
  for (int tile = wid; tile < tiles_ps; tile += nwaves) {
    const int r_in_s = tile * 64 + lane;
    const bool live = r_in_s < RT;
    const int rs_c = live ? r_in_s : RT - 1;
    const int64_t row = (int64_t)b * RT + rs_c;
    auto ld16 = [&](const float* base, float* dst) __attribute__((always_inline)) {
      const float4* q = reinterpret_cast<const float4*>(base + row * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float4 t = q[u]; dst[4 * u] = t.x; dst[4 * u + 1] = t.y; dst[4 * u + 2] = t.z; dst[4 * u + 3] = t.w; }
    };
    auto st16 = [&](float* base, const float* v) __attribute__((always_inline)) {
      if (base && live) {
        float4* q = reinterpret_cast<float4*>(base + row * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = make_float4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
      }
    };
    // ---- loads of the whole tile up front
    float xv[16], uv[16], dv[16], dm[16], u2a[16], u1a[16], u2b[16], u1b[16], r0g[16];
    if constexpr (UPT) {
      // d la = Upsample^T d rs: this row's 2 x 16 values [c][p], then two dense layers (one per output position of the conv); on its own
      // memory round trip in front of the tile's other loads (32 more live registers at the peak otherwise)
      float d0[16], d1[16];
      const float4* q = reinterpret_cast<const float4*>(a.dup + row * (2 * C));
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float4 t = q[u]; d0[2 * u] = t.x; d1[2 * u] = t.y; d0[2 * u + 1] = t.z; d1[2 * u + 1] = t.w; }
      const f32x16 t = dense(B_UP + C, d1, dense(B_UP, d0, zero16()));
#pragma unroll
      for (int c = 0; c < C; ++c) dv[c] = t[c];
      __builtin_amdgcn_sched_barrier(0);
    }
    ld16(a.x, xv); ld16(a.ypre, uv);
    if constexpr (!UPT) ld16(a.dy, dv);
    if constexpr (POST) {
      const float* q = a.dmid + (int64_t)b * C * RT + rs_c;
#pragma unroll
      for (int c = 0; c < C; ++c) dm[c] = q[(int64_t)c * RT];
    }
    ld16(a.blk[1].u2, u2a); ld16(a.blk[1].u1, u1a);
    __builtin_amdgcn_sched_barrier(0);  // (block 0's tensors are requested behind the LinearAttention part: 48 registers less at the peak)
    // ---- the last down level's k3 conv (centre tap): d la += W^T d mid_in; d rs in row layout for its weight gradient
    if constexpr (POST) {
      st16(a.drs_out, dm);
      const f32x16 t = dense(B_POST, dm, zero16());
#pragma unroll
      for (int c = 0; c < C; ++c) dv[c] += t[c];
    }
    DQ_PSTAMP(PIDB, 2);
    // ---- Residual(PreNorm(LinearAttention)) at one position (arithmetic of k_linattn_bwd1)
    float dout[16];
    {
      float ssq = 0.f, usq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { ssq = fmaf(xv[c], xv[c], ssq); usq = fmaf(uv[c], uv[c], usq); }
      const float nrm = fast_sqrt(ssq), unrm = fast_sqrt(usq);
      const float inv = sqC * fast_rcp(fmaxf(nrm, RMS_EPS)), uinv = fast_rcp(fmaxf(unrm, RMS_EPS));
      float xh[16], DY[16], gdv[16], lacc[3 * C];
#pragma unroll
      for (int i = 0; i < 3 * C; ++i) lacc[i] = 0.f;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        xh[c] = xv[c] * inv * prm[16 + c];
        const float uh = uv[c] * uinv;
        if (live) lacc[c] = dv[c] * (uh * sqC);  // d g_out
        gdv[c] = dv[c] * prm[c] * sqC;
        uv[c] = uh;
        dot = fmaf(gdv[c], uh, dot);
      }
      const bool uclamped = unrm < RMS_EPS;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        DY[c] = uclamped ? gdv[c] * uinv : uinv * (gdv[c] - uv[c] * dot);
        DY[c] = live ? DY[c] : 0.f;
        lacc[C + c] = DY[c];  // d b_out
      }
      // d W2[c'][c] += sum_rows dYpre[c'][row] xhat[c][row] (the 32^-0.5 is applied at the flush): K = rows = lanes, through the [c][row] tile
      wsync();
#pragma unroll
      for (int c = 0; c < C; ++c) { tr[c * 65 + lane] = DY[c]; tr[(C + c) * 65 + lane] = live ? xh[c] : 0.f; }
      wsync();
#pragma unroll
      for (int sgm = 0; sgm < 32; ++sgm) {
        const float av = tr[(lane & 15) * 65 + 2 * sgm + (lane >> 5)];
        const float bv = tr[(C + (lane & 15)) * 65 + 2 * sgm + (lane >> 5)];
        w2acc = mfma_f32(av, bv, w2acc);
      }
      const f32x16 dr = dense(B_LA, DY, zero16());  // 32^-0.5 (Wo Wv)^T dYpre
      const float pinv = fast_rcp(fmaxf(nrm, RMS_EPS));
      float tot[16], uh2[16];
      float dot2 = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float dxh = dr[c];
        uh2[c] = xv[c] * pinv;
        lacc[2 * C + c] = dxh * (uh2[c] * sqC);  // d g_pre  (dxh is 0 for dead lanes: DY is)
        tot[c] = dxh * prm[16 + c] * sqC;
        dot2 = fmaf(tot[c], uh2[c], dot2);
      }
      const bool clamped = nrm < RMS_EPS;
#pragma unroll
      for (int c = 0; c < C; ++c) dout[c] = dv[c] + (clamped ? tot[c] * pinv : pinv * (tot[c] - uh2[c] * dot2));
      flush_vals(lacc, std::integral_constant<int, 3 * C>{}, std::integral_constant<int, 0>{});
    }
    DQ_PSTAMP(PIDB, 3);
    ld16(a.blk[0].u2, u2b); ld16(a.blk[0].u1, u1b);
    if (a.r0out_g) ld16(a.r0out_g, r0g);
    __builtin_amdgcn_sched_barrier(0);
    // ---- ResnetBlock 1, then ResnetBlock 0
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int bi = 1 - pass;
      const TinyBwdBlkK& r = a.blk[bi];
      const int base = bi ? B_R1 : B_R0, pq = 2 + 4 * bi;
      float bacc[4 * C];
#pragma unroll
      for (int i = 0; i < 4 * C; ++i) bacc[i] = 0.f;
      if (bi == 0 && a.r0out_g) {
#pragma unroll
        for (int c = 0; c < C; ++c) dout[c] += r0g[c];
      }
      if (!live) {
#pragma unroll
        for (int c = 0; c < C; ++c) dout[c] = 0.f;
      }
      st16(r.dout_st, dout);
      float d[16], dsum_dummy[16];
      float* u2 = bi ? u2a : u2b;
      float* u1 = bi ? u1a : u1b;
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] = dout[c];
      norm_act_bwd<C, false>(u2, d, prm + (pq + 1) * 16, nullptr, bacc, dsum_dummy, dsum_dummy);   // block2: d u2 ; d g2
      st16(r.du2, d);
      const f32x16 da1 = dense(base, d, zero16());                                               // conv2^T
      float e[16];
#pragma unroll
      for (int c = 0; c < C; ++c) e[c] = da1[c];
      norm_act_bwd<C, true>(u1, e, prm + pq * 16, prm + (pq + 2) * 16, bacc + C, bacc + 2 * C, bacc + 3 * C);  // block1: d u1 ; d g1, d scale, d shift
      st16(r.du1, e);
      if (bi) flush_vals(bacc, std::integral_constant<int, 4 * C>{}, std::integral_constant<int, 3 * C + 4 * C>{});
      else flush_vals(bacc, std::integral_constant<int, 4 * C>{}, std::integral_constant<int, 3 * C>{});
      f32x16 dxa = dense(base + C, e, zero16());                                                 // conv1^T, x part
      if constexpr (CS > 0) {
        f32x16 dsk = dense(base + 2 * C, e, zero16());                                           // conv1^T, skip part
        if constexpr (WR) {
          dxa = dense(base + 3 * C, dout, dxa);                                                  // res_conv^T
          dsk = dense(base + 4 * C, dout, dsk);
        }
        if (r.dB && live) {
          float4* q = reinterpret_cast<float4*>(r.dB + row * CS);
#pragma unroll
          for (int u = 0; u < CS / 4; ++u) {
            float4 o = make_float4(dsk[4 * u], dsk[4 * u + 1], dsk[4 * u + 2], dsk[4 * u + 3]);
            if (r.dB_acc) { const float4 p0 = q[u]; o.x += p0.x; o.y += p0.y; o.z += p0.z; o.w += p0.w; }
            q[u] = o;
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dout[c] = dxa[c];
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) dout[c] = dxa[c] + dout[c];                                  // identity residual
      }
    }
    DQ_PSTAMP(PIDB, 4);
    // ---- the level's input gradient
    if constexpr (PRE == LEVEL_PRE_NONE) {
      if (live) {
        float* q = a.dfold + (int64_t)b * C * RT + rs_c;
#pragma unroll
        for (int c = 0; c < C; ++c) q[(int64_t)c * RT] = dout[c];
      }
    } else {
      st16(a.din_rows, dout);
      // Downsample k4 s2 p1 from two positions: d in[ci][j] += sum_co w[co][ci][1 + j] d out[co]
      const f32x16 t0 = dense(B_ST, dout, zero16()), t1 = dense(B_ST + C, dout, zero16());
      if (live) {
        float2* q = reinterpret_cast<float2*>(a.dprev + row * (CP * 2));
#pragma unroll
        for (int ci = 0; ci < CP; ++ci) { float2 o = q[ci]; o.x += t0[ci]; o.y += t1[ci]; q[ci] = o; }
      }
    }
  }
  DQ_PSTAMP(PIDB, 5);
  // ---- flush: the waves' row tables and d W2 tiles -> workgroup sums (fixed order), then the block's slots
  {
    float* crow = comb + (wv * 4) * NACC;  // [lane row][value]: lane (row, j) holds the row sums of the values i = j (mod 16)
#pragma unroll
    for (int sl = 0; sl < NACC / 16; ++sl) crow[(lane >> 4) * NACC + sl * 16 + (lane & 15)] = racc[sl];
  }
  {
    // the d W2 tile: register j of lane (half, col) holds [c' = rmap(j, half)][c = col]; rows / columns 16..31 are duplicates
    const int half = lane >> 5, col = lane & 31;
    float* cw = w2c + wv * (C * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cp = rmap(j, half);
      if (col < C) cw[cp * C + col] = w2acc[j];
    }
  }
  __syncthreads();
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  float* slot = a.la_part + (int64_t)wg * tiny_la_slot(C);
  for (int i = threadIdx.x; i < 256 * C; i += 256) slot[i] = 0.f;  // d Wq | d Wk
  for (int i = threadIdx.x; i < NACC + C * C; i += 256) {
    float v = 0.f;
    if (i < NACC) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v += comb[q * NACC + i];  // (wave, lane row) order
    } else {
      const int e = i - NACC;
      v = (w2c[e] + w2c[C * C + e]) + (w2c[2 * C * C + e] + w2c[3 * C * C + e]);
    }
    if (i < 3 * C) slot[256 * C + 4 * C * C + i] = v;                                   // d g_out | d b_out | d g_pre
    else if (i < NACC) {
      const int bi = (i - 3 * C) / (4 * C), e = (i - 3 * C) % (4 * C), what = e / C, c = e % C;
      // order per block: d g2 | d g1 | d scale | d shift  == ResBwd::gpart order
      a.blk[bi].gpart[(int64_t)wg * (4 * C) + what * C + c] = v;
    } else {
      const float w = v * 0.17677669529663687f;
#pragma unroll
      for (int hd = 0; hd < 4; ++hd) slot[256 * C + hd * C * C + (i - NACC)] = w;
    }
  }
  DQ_PSTAMP(PIDB, 6);
}

// the layer list of a launch, in image order (the kernel's step bases follow the same formulas)
int tiny_layers(const TinyFwd& t, TinyLayer* L) {
  const LevelFwd& a = t.lv;
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - a.params) : -1; };
  int nl = 0;
  const int C = a.C, N = a.n;
  if (a.pre == LEVEL_PRE_DOWN) L[nl++] = TinyLayer{N == 2 ? TL_DOWN_N2 : TL_DOWN_N1, poff(a.pw), 0, a.cp, 4, 0, 2 * a.cp, C};
  if (a.pre == LEVEL_PRE_UP) L[nl++] = TinyLayer{TL_UP_N2, poff(a.pw), 0, a.cp, 3, 0, 2 * a.cp, C};
  for (int b = 0; b < 2; ++b) {
    const ResFwd& r = a.blk[b];
    const int cs = r.cinB, cin = C + cs, kind = N == 2 ? TL_CONV3_N2 : TL_DIAG;
    L[nl++] = TinyLayer{kind, poff(r.w1), 0, cin, 3, 0, C, C};
    if (cs) L[nl++] = TinyLayer{kind, poff(r.w1), 0, cin, 3, C, cs, C};
    L[nl++] = TinyLayer{kind, poff(r.w2), 0, C, 3, 0, C, C};
    if (cs) {
      L[nl++] = TinyLayer{TL_DIAG, poff(r.wr), 0, cin, 1, 0, C, C};
      L[nl++] = TinyLayer{TL_DIAG, poff(r.wr), 0, cin, 1, C, cs, C};
    }
  }
  if (t.la) L[nl++] = TinyLayer{TL_LA1, poff(t.w_qkv), poff(t.w_out), C, 1, 0, C, C};
  if (t.post_w) L[nl++] = TinyLayer{TL_DIAG, poff(t.post_w), 0, C, 3, 0, C, C};  // (only behind the LinearAttention: tiny_fwd_usable)
  return nl;
}


// the backward's transposed layers, in image order (k_tiny_bwd's step bases follow the same formulas)
int tiny_bwd_layers(const TinyBwd& t, TinyLayer* L) {
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - t.params) : -1; };
  int nl = 0;
  const int C = t.C, cs = t.cs, cin = C + cs;
  if (t.pre == LEVEL_PRE_DOWN) L[nl++] = TinyLayer{TL_DIAG_T, poff(t.post_w), 0, C, 3, 0, C, C};
  L[nl++] = TinyLayer{TL_LA1_T, poff(t.w_qkv), poff(t.w_out), C, 1, 0, C, C};
  for (int b = 1; b >= 0; --b) {
    const TinyBwd::Blk& r = t.blk[b];
    L[nl++] = TinyLayer{TL_DIAG_T, poff(r.w2), 0, C, 3, 0, C, C};
    L[nl++] = TinyLayer{TL_DIAG_T, poff(r.w1), 0, cin, 3, 0, C, C};
    if (cs) {
      L[nl++] = TinyLayer{TL_DIAG_T, poff(r.w1), 0, cin, 3, C, C, cs};
      L[nl++] = TinyLayer{TL_DIAG_T, poff(r.wr), 0, cin, 1, 0, C, C};
      L[nl++] = TinyLayer{TL_DIAG_T, poff(r.wr), 0, cin, 1, C, C, cs};
    }
  }
  if (t.pre == LEVEL_PRE_DOWN) {
    L[nl++] = TinyLayer{TL_DOWN_N1_T, poff(t.stage_w), 0, t.cp, 4, 0, C, t.cp};
    L[nl++] = TinyLayer{TL_DOWN_N1_T, poff(t.stage_w), 1, t.cp, 4, 0, C, t.cp};
  }
  if (t.up_w) {  // (last in the image: the other bases do not move)
    L[nl++] = TinyLayer{TL_UP_N2_T, poff(t.up_w), 0, C, 3, 0, C, C};
    L[nl++] = TinyLayer{TL_UP_N2_T, poff(t.up_w), 1, C, 3, 0, C, C};
  }
  return nl;
}

bool tiny_bwd_enabled() {
  const bool on = !DQ_DEV_FLAG("DQ_NO_TINY_BWD", '1');  // (dev switch)
  return on;
}

bool tiny_enabled() {
  const bool on = !DQ_DEV_FLAG("DQ_NO_TINY", '1');  // (dev switch)
  return on;
}

}  // namespace

bool tiny_fwd_usable(const TinyFwd& t) {
  const LevelFwd& a = t.lv;
  if (!tiny_enabled()) return false;
  if (a.nblocks != 2 || a.rows_per_sample <= 1 || !a.params) return false;
  const int cs = a.blk[0].cinB;
  if (a.blk[1].cinB != cs) return false;
  for (int b = 0; b < 2; ++b) if ((cs > 0) != (a.blk[b].wr != nullptr)) return false;
  const bool extras = t.la || t.post_w || t.in_folded;
  if (extras && a.n != 1) return false;
  if (t.post_w && !t.la) return false;  // (the post conv's steps sit behind the LinearAttention's in the image)
  if (a.C == 12 && a.n == 2 && a.pre == LEVEL_PRE_DOWN && a.cp == 12 && cs == 0) return true;
  if (a.C == 16 && a.n == 1 && a.pre == LEVEL_PRE_DOWN && a.cp == 12 && cs == 0) return true;
  if (a.C == 16 && a.n == 1 && a.pre == LEVEL_PRE_NONE && cs == 16) return true;
  if (a.C == 16 && a.n == 2 && a.pre == LEVEL_PRE_UP && a.cp == 16 && cs == 12) return true;
  return false;
}

int64_t tiny_img_floats(const TinyFwd& t) {
  TinyLayer L[TINY_MAX_LAYERS];
  const int nl = tiny_layers(t, L);
  int64_t steps = 0;
  for (int i = 0; i < nl; ++i) steps += L[i].steps;
  return steps * 64;
}

int launch_tiny_images(const TinyFwd* calls, int count, hipStream_t s) {
  if (count == 0) return 0;
  DQ_REQUIRE(count <= TINY_IMG_MAX, "tiny images: too many launches");
  TinyImgMulti mm;
  int64_t mx = 0;
  for (int i = 0; i < count; ++i) {
    const TinyFwd& t = calls[i];
    DQ_REQUIRE(t.img && ((uintptr_t)t.img & 15) == 0 && t.lv.params, "tiny images: missing / misaligned image buffer");
    DQ_REQUIRE(tiny_fwd_usable(t), "tiny images: unsupported shape");
    DQ_REQUIRE(t.lv.params == calls[0].lv.params, "tiny images: the launches must share one parameter buffer");
    TinyImgItem& it = mm.it[i];
    it.nl = tiny_layers(t, it.L);
    it.total_steps = 0;
    for (int l = 0; l < it.nl; ++l) it.total_steps += it.L[l].steps;
    it.dst = const_cast<float*>(t.img);
    DQ_REQUIRE((int64_t)it.total_steps * 64 <= TINY_IMG_FLOATS, "tiny images: image larger than its slot");
    mx = std::max<int64_t>(mx, (int64_t)it.total_steps * 64);
  }
  hipLaunchKernelGGL(k_tiny_images, dim3(cdiv(mx, 256), count), dim3(256), 0, s, mm, calls[0].lv.params);
  DQ_LAUNCH_CHECK();
  return 0;
}

static int tiny_num_cus() {
  static const int v = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
  return v;
}

int launch_tiny_fwd(const TinyFwd& t, hipStream_t s) {
  const LevelFwd& a = t.lv;
  DQ_REQUIRE(tiny_fwd_usable(t), "tiny_fwd: unsupported shape");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0 && a.in && t.img, "tiny_fwd: bad rows / missing input or operand image");
  DQ_REQUIRE(a.pre == LEVEL_PRE_NONE || (a.pw && a.pb), "tiny_fwd: the input stage needs its conv weight and bias");
  DQ_REQUIRE(!t.la || (t.w_qkv && t.w_out && t.b_out && t.g_pre && t.g_out && t.la_y), "tiny_fwd: incomplete LinearAttention operands");
  DQ_REQUIRE(!t.post_w || (t.post_b && t.post_out), "tiny_fwd: incomplete post conv");
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - a.params) : -1; };
  TinyFwdK k;
  k.in = a.in; k.pre_out = a.pre_out; k.in_copy = t.in_copy; k.pb = poff(a.pb); k.rows_per_sample = a.rows_per_sample; k.in_folded = t.in_folded;
  const float* ssb = a.blk[0].ss;
  k.ss_stride = a.blk[0].ss_stride;
  for (int b = 0; b < 2; ++b) {
    const ResFwd& r = a.blk[b];
    DQ_REQUIRE(r.w1 && r.b1 && r.g1 && r.w2 && r.b2 && r.g2 && r.ss && (r.cinB == 0 || (r.inB && r.wr && r.br)), "tiny_fwd: missing block operand");
    DQ_REQUIRE(r.ss_stride == k.ss_stride && r.ss >= ssb - (1 << 20) && r.ss <= ssb + (1 << 20), "tiny_fwd: the blocks' scale / shift vectors must share one buffer");
    for (const float* q : {r.w1, r.b1, r.g1, r.w2, r.b2, r.g2, r.wr, r.br})
      DQ_REQUIRE(!q || (q >= a.params && q - a.params < (1ll << 31)), "tiny_fwd: a parameter lies outside the flat parameter buffer");
    TinyBlkK& d = k.blk[b];
    d.b1 = poff(r.b1); d.g1 = poff(r.g1); d.b2 = poff(r.b2); d.g2 = poff(r.g2); d.br = poff(r.br);
    d.ss_off = (int)(r.ss - ssb);
    d.inB = r.inB; d.u1 = r.u1; d.a1 = r.a1; d.u2 = r.u2; d.out = r.out;
  }
  k.la = t.la; k.la_gpre = poff(t.g_pre); k.la_bo = poff(t.b_out); k.la_go = poff(t.g_out); k.la_y = t.la_y; k.la_ypre = t.la_ypre;
  k.post = t.post_w ? 1 : 0; k.post_b = poff(t.post_b); k.post_out = t.post_out;
  k.total_steps = (int)(tiny_img_floats(t) / 64);
  const int B = a.rows / a.rows_per_sample;
  const int rpt = 64 / a.n;
  const int tiles_ps = cdiv(a.rows_per_sample, rpt);
  // (image region padded to whole rounds of 256 threads x 16 bytes over the instantiation's largest image: steps + the optional LinearAttention / post conv)
  const int max_steps = k.total_steps + (t.la ? 0 : a.C) + (t.post_w ? 0 : a.C);
  const int e_in = a.pre == LEVEL_PRE_DOWN ? a.cp * 2 * a.n : (a.pre == LEVEL_PRE_UP ? a.cp : 0);
  const size_t lds = (size_t)tiny_rounds(max_steps) * 4096 + 32 * 16 * 4 + (size_t)4 * tiny_stg_floats(a.n, e_in, a.C, a.blk[0].cinB) * 4;
  DQ_REQUIRE((int64_t)tiny_rounds(max_steps) * 1024 <= TINY_IMG_FLOATS, "tiny_fwd: image slot too small for the padded copy");
  DQ_REQUIRE(lds <= 96 * 1024, "tiny_fwd: weight image too large");
  const int cs = a.blk[0].cinB;
#define DQ_TINY(CC, NN, PP, PC, SS)                                                                                           \
  if (a.C == CC && a.n == NN && a.pre == PP && (PP == LEVEL_PRE_NONE || a.cp == PC) && cs == SS) {                            \
    const int nb = occ_blocks_per_cu((const void*)k_tiny_fwd<CC, NN, PP, PC, SS>, 256, lds);                                  \
    if (nb < 0) return 1;                                                                                                     \
    const int gx = std::max(1, std::min(nb * tiny_num_cus() / B, (tiles_ps + 3) / 4));                                        \
    hipLaunchKernelGGL((k_tiny_fwd<CC, NN, PP, PC, SS>), dim3(gx, B), dim3(256), lds, s, k, a.params, ssb, t.img, tiles_ps);  \
    DQ_LAUNCH_CHECK();                                                                                                        \
    return 0;                                                                                                                 \
  }
  DQ_TINY(12, 2, LEVEL_PRE_DOWN, 12, 0)
  DQ_TINY(16, 1, LEVEL_PRE_DOWN, 12, 0)
  DQ_TINY(16, 1, LEVEL_PRE_NONE, 0, 16)
  DQ_TINY(16, 2, LEVEL_PRE_UP, 16, 12)
#undef DQ_TINY
  set_error("tiny_fwd: unsupported (C, n, stage, stage input width, skip width)");
  return 2;
}


static int tiny_bwd_gx(const TinyBwd& t);
bool tiny_bwd_usable(const TinyBwd& t) {
  if (!tiny_enabled() || !tiny_bwd_enabled()) return false;
  if (t.C != 16 || t.rows_per_sample <= 1 || !t.params || t.rows <= 0 || t.rows % t.rows_per_sample) return false;
  // one LinearAttention slot per workgroup: a layer's reservation (la_part_reserve: 1024 slots at 16 channels) must hold them -- batches
  // beyond ~1000 samples keep the per-kernel backward
  if ((int64_t)tiny_bwd_gx(t) * (t.rows / t.rows_per_sample) > 1000) return false;
  if (t.pre == LEVEL_PRE_DOWN) return t.cp == 12 && t.cs == 0;
  if (t.pre == LEVEL_PRE_NONE) return t.cs == 16 && !(t.dup && !t.up_w);
  return false;
}
static int tiny_bwd_gx(const TinyBwd& t) {
  const int B = t.rows / t.rows_per_sample, tiles_ps = cdiv(t.rows_per_sample, 64);
  return std::max(1, std::min(tiny_num_cus() / std::max(1, B), (tiles_ps + 3) / 4));
}
int tiny_bwd_slots(const TinyBwd& t) { return tiny_bwd_gx(t) * (t.rows / t.rows_per_sample); }
int64_t tiny_bwd_img_floats(const TinyBwd& t) {
  TinyLayer L[TINY_MAX_LAYERS + 4];
  const int nl = tiny_bwd_layers(t, L);
  int64_t steps = 0;
  for (int i = 0; i < nl; ++i) steps += L[i].steps;
  return steps * 64;
}
int launch_tiny_bwd_images(const TinyBwd* calls, int count, hipStream_t s) {
  if (count == 0) return 0;
  DQ_REQUIRE(count <= TINY_IMG_MAX, "tiny bwd images: too many launches");
  TinyImgMulti mm;
  int64_t mx = 0;
  for (int i = 0; i < count; ++i) {
    const TinyBwd& t = calls[i];
    DQ_REQUIRE(t.img && ((uintptr_t)t.img & 15) == 0 && tiny_bwd_usable(t), "tiny bwd images: missing image buffer / unsupported shape");
    DQ_REQUIRE(t.params == calls[0].params, "tiny bwd images: the launches must share one parameter buffer");
    TinyImgItem& it = mm.it[i];
    TinyLayer L[TINY_MAX_LAYERS + 4];
    it.nl = tiny_bwd_layers(t, L);
    DQ_REQUIRE(it.nl <= TINY_MAX_LAYERS, "tiny bwd images: too many layers");
    it.total_steps = 0;
    for (int l = 0; l < it.nl; ++l) { it.L[l] = L[l]; it.total_steps += L[l].steps; }
    it.dst = const_cast<float*>(t.img);
    DQ_REQUIRE((int64_t)tiny_rounds(it.total_steps) * 1024 <= TINY_IMG_FLOATS, "tiny bwd images: image larger than its slot");
    mx = std::max<int64_t>(mx, (int64_t)it.total_steps * 64);
  }
  hipLaunchKernelGGL(k_tiny_images, dim3(cdiv(mx, 256), count), dim3(256), 0, s, mm, calls[0].params);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_tiny_bwd(const TinyBwd& t, hipStream_t s) {
  DQ_REQUIRE(tiny_bwd_usable(t), "tiny_bwd: unsupported shape");
  DQ_REQUIRE(!t.up_w || (t.pre == LEVEL_PRE_NONE && t.dup), "tiny_bwd: the Upsample stage needs d rs and belongs to the up level");
  DQ_REQUIRE(t.rows % t.rows_per_sample == 0 && t.img && t.x && t.ypre && (t.dy || t.up_w) && t.w_qkv && t.w_out && t.g_pre && t.g_out && t.la_part, "tiny_bwd: missing operand");
  DQ_REQUIRE(t.pre != LEVEL_PRE_DOWN || (t.post_w && t.dmid && t.drs_out && t.din_rows && t.dprev && t.stage_w), "tiny_bwd: incomplete down-level operands");
  DQ_REQUIRE(t.pre != LEVEL_PRE_NONE || t.dfold, "tiny_bwd: missing input-gradient tensor");
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - t.params) : -1; };
  const int C = t.C, B = t.rows / t.rows_per_sample, gx = tiny_bwd_gx(t);
  TinyBwdK k;
  k.x = t.x; k.ypre = t.ypre; k.dy = t.dy; k.la_gpre = poff(t.g_pre); k.la_go = poff(t.g_out); k.la_part = t.la_part;
  k.dmid = t.dmid; k.drs_out = t.drs_out; k.dfold = t.dfold; k.din_rows = t.din_rows; k.dprev = t.dprev; k.r0out_g = t.r0out_g;
  k.rows_per_sample = t.rows_per_sample; k.dup = t.dup;
  const float* ssb = t.blk[0].ss;
  k.ss_stride = t.blk[0].ss_stride;
  DQ_REQUIRE(t.la_part_floats >= (int64_t)gx * B * tiny_la_slot(C) + 4 * C * C, "tiny_bwd: LinearAttention slot region too small");
  for (int b = 0; b < 2; ++b) {
    const TinyBwd::Blk& r = t.blk[b];
    DQ_REQUIRE(r.w1 && r.w2 && r.g1 && r.g2 && r.ss && r.u1 && r.u2 && r.du1 && r.du2 && r.gpart && (t.cs == 0 || r.wr), "tiny_bwd: missing block operand");
    DQ_REQUIRE(r.ss_stride == k.ss_stride && r.gpart_floats >= (int64_t)gx * B * 4 * C, "tiny_bwd: scale / shift stride or partial-sum slot");
    TinyBwdBlkK& d = k.blk[b];
    d.g1 = poff(r.g1); d.g2 = poff(r.g2); d.ss_off = (int)(r.ss - ssb);
    d.u1 = r.u1; d.u2 = r.u2; d.du1 = r.du1; d.du2 = r.du2; d.dB = r.dB; d.dB_acc = r.dB_acc; d.dout_st = r.dout_st; d.gpart = r.gpart;
  }
  if (t.gblocks) *t.gblocks = gx;
  const int tiles_ps = cdiv(t.rows_per_sample, 64);
  const int blkt = 2 * C + (t.cs ? C : 0) + (t.cs ? 2 * C : 0);
  const int total = (t.pre == LEVEL_PRE_DOWN ? C : 0) + C + 2 * blkt + (t.pre == LEVEL_PRE_DOWN ? 2 * C : 0) + (t.up_w ? 2 * C : 0);
  DQ_REQUIRE((int64_t)total * 64 == tiny_bwd_img_floats(t), "tiny_bwd: image layout mismatch");
  const size_t lds = (size_t)tiny_rounds(total) * 4096 + 256 * 4 + (size_t)4 * (2 * C * 65) * 4 + (size_t)(16 * 11 * C + 4 * C * C) * 4;
#define DQ_TINYB(PP, PC, SS, UU)                                                                                              \
  if (t.pre == PP && t.cs == SS && (t.up_w != nullptr) == UU) {                                                               \
    const int nb = occ_blocks_per_cu((const void*)k_tiny_bwd<16, PP, PC, SS, UU>, 256, lds);                                  \
    if (nb < 0) return 1;                                                                                                     \
    hipLaunchKernelGGL((k_tiny_bwd<16, PP, PC, SS, UU>), dim3(gx, B), dim3(256), lds, s, k, t.params, ssb, t.img, tiles_ps);  \
    DQ_LAUNCH_CHECK();                                                                                                        \
    return 0;                                                                                                                 \
  }
  DQ_TINYB(LEVEL_PRE_DOWN, 12, 0, false)
  DQ_TINYB(LEVEL_PRE_NONE, 0, 16, false)
  DQ_TINYB(LEVEL_PRE_NONE, 0, 16, true)
#undef DQ_TINYB
  set_error("tiny_bwd: unsupported (stage, skip width)");
  return 2;
}

}  // namespace dq
