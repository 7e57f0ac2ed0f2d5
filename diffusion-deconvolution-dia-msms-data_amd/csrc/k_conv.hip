// Conv-family kernels of the U-Net (K2, K3, K4, K6, K8 and the 1x1 projections of K7), forward and backward.
// Reference arithmetic: dquartic/model/unet1d.py:223-323 (Block / ResnetBlock: Conv1d(k3,p1) -> RMSNorm ->
// x*(scale+1)+shift -> SiLU, + res_conv), :82-110 (Upsample = nearest x2 + Conv1d(k3,p1); Downsample =
// Conv1d(k4,s2,p1)), :113-140 (RMSNorm = F.normalize(dim=1) * g * sqrt(C)).
//
// Layout: every activation is (rows, C, n) fp32 with n (m/z, or RT in the bottleneck) contiguous.
// Mapping: one thread per (row, output position); all output channels of that position live in registers, so the
// RMSNorm over channels, the scale/shift, the activation and the residual are in-thread.  Consecutive lanes
// are consecutive positions => coalesced 256-B wave accesses per channel.  Weights are read through
// wave-uniform addresses (scalar loads).  No LDS.
#include "dq_common.h"
#include "dq_kernels.h"
#include <cstdlib>

namespace dq {

// input position feeding output position p through tap k, or -1 (zero padding)
template <int MODE, int K>
__device__ __forceinline__ int tap_pos(int p, int k, int n_in) {
  if (MODE == CONV_S1) {
    const int q = p + k - (K - 1) / 2;
    return (q >= 0 && q < n_in) ? q : -1;
  } else if (MODE == CONV_DOWN) {  // k4 s2 p1
    const int q = 2 * p + k - 1;
    return (q >= 0 && q < n_in) ? q : -1;
  } else {  // nearest x2 then k3 p1
    const int j = p + k - 1;
    return (j >= 0 && j < 2 * n_in) ? (j >> 1) : -1;
  }
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <int COUT, int K, int MODE>
__global__ void __launch_bounds__(256) k_conv_fwd(ConvFwd a) {
  const int64_t item = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.rows * a.n_out;
  const int co_base = blockIdx.y * COUT;  // output-channel chunk (only for the norm-free 1x1 projections)
  const int cin = a.cinA + a.cinB;
  // this block's weights, transposed to [ci][co][k], in LDS: inside the input-channel loop they were COUT x K scattered scalar
  // loads per iteration with nothing to overlap them
  constexpr int WSH = 4096;
  __shared__ __attribute__((aligned(16))) float wsh[WSH];
  const bool use_lds = COUT * cin * K <= WSH;
  if (use_lds) {
    const float* wg = a.w + (int64_t)co_base * cin * K;
    // (256 threads; all of a thread's loads requested before its first store: as a loop striding by blockDim.x -- not unrollable -- the
    // staging was up to 16 memory round trips in a row)
    constexpr int NIT = WSH / 256;
    const int tot = COUT * cin * K;
    float v[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      v[u] = 0.f;
      if (u * 256 < tot) v[u] = wg[i < tot ? i : 0];  // (wave-uniform: a 56-float weight costs one load per thread, not sixteen)
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      if (i < tot) {
        const int co = i / (cin * K), r = i - co * (cin * K), ci = r / K, k = r - ci * K;
        wsh[(ci * COUT + co) * K + k] = v[u];
      }
    }
    __syncthreads();
  }
  if (item >= total) return;
  const int row = (int)(item / a.n_out), p = (int)(item % a.n_out);

  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = a.bias ? a.bias[co_base + co] : 0.f;

  int q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = tap_pos<MODE, K>(p, k, a.n_in);

  const float* w = a.w + (int64_t)co_base * cin * K;
  for (int ci = 0; ci < cin; ++ci) {
    const float* src = (ci < a.cinA) ? a.inA + ((int64_t)row * a.cinA + ci) * a.n_in
                                     : a.inB + ((int64_t)row * a.cinB + (ci - a.cinA)) * a.n_in;
    float xv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) xv[k] = q[k] >= 0 ? src[q[k]] : 0.f;
    if (use_lds) {
      const float* wl = wsh + ci * (COUT * K);
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[co] = fmaf(wl[co * K + k], xv[k], acc[co]);
      }
    } else {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[co] = fmaf(w[((int64_t)co * cin + ci) * K + k], xv[k], acc[co]);
      }
    }
  }

  const int cout_total = gridDim.y * COUT;
  if (a.u_out) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) a.u_out[((int64_t)row * cout_total + co_base + co) * a.n_out + p] = acc[co];
  }
  if (a.g) {  // RMSNorm over channels (chunking is never combined with a norm)
    float ssq = 0.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co) ssq = fmaf(acc[co], acc[co], ssq);
    const float inv = rms_inv(ssq, sqrtf((float)COUT));
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = acc[co] * inv * a.g[co];
  }
  if (a.ss) {
    const float* ss = a.ss + (int64_t)(row / a.rows_per_sample) * a.ss_stride;
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(acc[co], ss[co] + 1.0f, ss[COUT + co]);
  }
  if (a.act == ACT_SILU) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = silu_f(acc[co]);
  } else if (a.act == ACT_GELU) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = gelu_f(acc[co]);
  }
  if (a.resA) {
    if (a.res_w) {
      const int rcin = a.rcinA + a.rcinB;
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] += a.res_b ? a.res_b[co] : 0.f;
      for (int ci = 0; ci < rcin; ++ci) {
        const float xv = (ci < a.rcinA) ? a.resA[((int64_t)row * a.rcinA + ci) * a.n_out + p]
                                        : a.resB[((int64_t)row * a.rcinB + (ci - a.rcinA)) * a.n_out + p];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a.res_w[(int64_t)co * rcin + ci], xv, acc[co]);
      }
    } else {
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] += a.resA[((int64_t)row * cout_total + co_base + co) * a.n_out + p];
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) a.y_out[((int64_t)row * cout_total + co_base + co) * a.n_out + p] = acc[co];
}

template <int COUT>
static int conv_fwd_dispatch(const ConvFwd& a, int chunks, hipStream_t s) {
  const int64_t total = (int64_t)a.rows * a.n_out;
  if (total == 0) return 0;
  dim3 grid(cdiv(total, 256), chunks), block(256);
#define DQ_CF(KK, MM)                                                                   \
  if (a.K == KK && a.mode == MM) {                                                       \
    hipLaunchKernelGGL((k_conv_fwd<COUT, KK, MM>), grid, block, 0, s, a);                \
    DQ_LAUNCH_CHECK();                                                                   \
    return 0;                                                                            \
  }
  DQ_CF(1, CONV_S1)
  DQ_CF(3, CONV_S1)
  DQ_CF(7, CONV_S1)
  DQ_CF(4, CONV_DOWN)
  DQ_CF(3, CONV_UP)
#undef DQ_CF
  set_error("conv_fwd: unsupported (K, mode) = (" + std::to_string(a.K) + ", " + std::to_string(a.mode) + ")");
  return 2;
}

int launch_conv_fwd(const ConvFwd& a, hipStream_t s) {
  DQ_REQUIRE(a.inA && a.w && a.y_out && a.cout > 0 && a.cinA > 0, "conv_fwd: missing operand");
  DQ_REQUIRE(a.cinB == 0 || a.inB, "conv_fwd: cinB > 0 needs inB");
  if (a.mode == CONV_S1) DQ_REQUIRE(a.n_in == a.n_out && (a.K & 1), "conv_fwd: stride-1 conv needs n_in == n_out and odd K");
  if (a.mode == CONV_DOWN) DQ_REQUIRE(a.n_in == 2 * a.n_out && a.K == 4, "conv_fwd: downsample needs n_in == 2*n_out, K == 4");
  if (a.mode == CONV_UP) DQ_REQUIRE(2 * a.n_in == a.n_out && a.K == 3, "conv_fwd: upsample needs n_out == 2*n_in, K == 3");
  if (a.resA && !a.res_w) DQ_REQUIRE(a.rcinA == a.cout && a.rcinB == 0, "conv_fwd: identity residual needs cout channels");
  const bool chunkable = !a.g && !a.ss && !(a.resA && a.res_w);
  switch (a.cout) {
    case 1: return conv_fwd_dispatch<1>(a, 1, s);
    case 4: return conv_fwd_dispatch<4>(a, 1, s);
    case 8: return conv_fwd_dispatch<8>(a, 1, s);
    case 12: return conv_fwd_dispatch<12>(a, 1, s);
    case 16: return conv_fwd_dispatch<16>(a, 1, s);
    case 32: return conv_fwd_dispatch<32>(a, 1, s);
    // (64 output channels in one thread -- the 64-channel bottleneck of BASELINE configs[4] until round 4 -- went to the im2col + GEMM path: k_wide.hip)
    default:
      if (chunkable && a.cout % 32 == 0) return conv_fwd_dispatch<32>(a, a.cout / 32, s);
      set_error("conv_fwd: unsupported cout " + std::to_string(a.cout));
      return 2;
  }
}

// -------------------------------------------------------------------------------------------------
// pointwise backward of norm -> scale/shift -> act
// grid (blocks_per_sample, B): a thread only ever touches items of one sample, so the per-sample
// d(scale)/d(shift) sums are block-reduced and cost one atomic per (block, channel).
// -------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_block_bwd(BlockBwd a) {
  const int b = blockIdx.y;
  const int rows_here = min(a.rows_per_sample, a.rows - b * a.rows_per_sample);  // the last group may be ragged
  const int64_t per_sample = (int64_t)rows_here * a.n;                            // items of this sample / group
  const float sqC = sqrtf((float)C);
  float g[C], sc[C], sh[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    g[c] = a.g ? a.g[c] : 1.f;
    sc[c] = a.ss ? a.ss[(int64_t)b * a.ss_stride + c] + 1.0f : 1.f;
    sh[c] = a.ss ? a.ss[(int64_t)b * a.ss_stride + C + c] : 0.f;
  }
  float dg[C], dsc[C], dsh[C], dbs[C];
#pragma unroll
  for (int c = 0; c < C; ++c) dg[c] = dsc[c] = dsh[c] = dbs[c] = 0.f;

  for (int64_t it = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; it < per_sample; it += (int64_t)gridDim.x * blockDim.x) {
    const int row = b * a.rows_per_sample + (int)(it / a.n), p = (int)(it % a.n);
    const int64_t base = (int64_t)row * C * a.n + p;
    float u[C], d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { u[c] = a.u[base + (int64_t)c * a.n]; d[c] = a.dy[base + (int64_t)c * a.n]; }
    if (a.g) {
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) ssq = fmaf(u[c], u[c], ssq);
      const float nrm = fast_sqrt(ssq);
      const float inv = fast_rcp(fmaxf(nrm, RMS_EPS));
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float uh = u[c] * inv;            // normalised
        const float z = uh * g[c] * sqC;        // after gain
        const float w = fmaf(z, sc[c], sh[c]);  // after scale/shift
        float dw = d[c];
        if (a.act == ACT_SILU) dw *= silu_grad_f(w);
        else if (a.act == ACT_GELU) dw *= gelu_grad_f(w);
        dsh[c] += dw;
        dsc[c] = fmaf(dw, z, dsc[c]);
        const float dz = dw * sc[c];
        dg[c] = fmaf(dz, uh * sqC, dg[c]);
        const float gd = dz * g[c] * sqC;  // grad wrt the normalised value
        d[c] = gd;
        u[c] = uh;
        dot = fmaf(gd, uh, dot);
      }
      const bool clamped = nrm < RMS_EPS;  // F.normalize clamps the norm: below eps the map is linear
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] = clamped ? d[c] * inv : inv * (d[c] - u[c] * dot);
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float w = fmaf(u[c], sc[c], sh[c]);
        float dw = d[c];
        if (a.act == ACT_SILU) dw *= silu_grad_f(w);
        else if (a.act == ACT_GELU) dw *= gelu_grad_f(w);
        dsh[c] += dw;
        dsc[c] = fmaf(dw, u[c], dsc[c]);
        d[c] = dw * sc[c];
      }
    }
    float oldv[C];  // an accumulating launch reads all C old values before its first store
#pragma unroll
    for (int c = 0; c < C; ++c) oldv[c] = a.accumulate ? (a.add_src ? a.add_src : a.du)[base + (int64_t)c * a.n] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      dbs[c] += d[c];
      a.du[base + (int64_t)c * a.n] = oldv[c] + d[c];
    }
  }

  __shared__ float red[4][4 * C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float s0 = wave_sum(dg[c]), s1 = wave_sum(dsc[c]), s2 = wave_sum(dsh[c]), s3 = wave_sum(dbs[c]);
    if (lane == 0) { red[wv][c] = s0; red[wv][C + c] = s1; red[wv][2 * C + c] = s2; red[wv][3 * C + c] = s3; }
  }
  __syncthreads();
  // [dg | dscale | dshift | dbias] of this block into its own slot; launch_part_reduce sums the slots in block order
  if (a.part)
    for (int i = threadIdx.x; i < 4 * C; i += blockDim.x)
      a.part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * C) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// -------------------------------------------------------------------------------------------------
// Ordered reduction of per-block partial sums: part[(b * gx + x) * nv + i], b < B groups (samples), x < gx blocks per group.
//   "global" segments  [start, start + len) -> dst[j] += sum over all (b, x)      (norm gains, biases)
//   one per-group segment [s0, s0 + sn)     -> sdst[b * sstride + j] += sum over x (a sample's d(scale), d(shift))
// Every sum is taken in a fixed order (strided per thread, then an LDS tree in thread order): repeatable to the bit, which the
// float atomics this replaces were not.
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_part_reduce(PartReduce a, int nglobal) {
  __shared__ float red[256];
  if ((int)blockIdx.x < nglobal) {
    int j = blockIdx.x, seg = 0;
    while (j >= a.seg_len[seg]) { j -= a.seg_len[seg]; ++seg; }
    const int col = a.seg_start[seg] + j;
    const int blocks = a.B * a.gx;
    float s0 = 0.f, s1 = 0.f;
    int k = threadIdx.x;
    for (; k + 256 < blocks; k += 512) {
      s0 += a.part[(int64_t)k * a.nv + col];
      s1 += a.part[(int64_t)(k + 256) * a.nv + col];
    }
    if (k < blocks) s0 += a.part[(int64_t)k * a.nv + col];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16) {  // 16 threads x 16 consecutive partials, then one thread over the 16
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += red[threadIdx.x * 16 + q];
      red[threadIdx.x * 16] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += red[q * 16];
      a.seg_dst[seg][j] += t;
    }
    return;
  }
  // group b: thread (slice, i) sums x = slice, slice + S, ... of value s0 + i; the S slices meet in LDS in slice order
  const int b = blockIdx.x - nglobal;
  for (int i0 = 0; i0 < a.sn; i0 += 32) {
    const int i = i0 + (threadIdx.x & 31), slice = threadIdx.x >> 5;
    float t = 0.f;
    if (i < a.sn)
      for (int x = slice; x < a.gx; x += 8) t += a.part[((int64_t)b * a.gx + x) * a.nv + a.s0 + i];
    red[threadIdx.x] = t;
    __syncthreads();
    if (slice == 0 && i < a.sn) {
      float v = 0.f;
      for (int q = 0; q < 8; ++q) v += red[q * 32 + (threadIdx.x & 31)];
      a.sdst[(int64_t)b * a.sstride + i] += v;
    }
    __syncthreads();
  }
}

int launch_part_reduce(const PartReduce& a, hipStream_t s) {
  DQ_REQUIRE(a.part && a.B > 0 && a.gx > 0 && a.nv > 0 && a.nseg >= 0 && a.nseg <= 3, "part_reduce: bad descriptor");
  int nglobal = 0;
  for (int i = 0; i < a.nseg; ++i) nglobal += a.seg_len[i];
  const int groups = (a.sn > 0 && a.sdst) ? a.B : 0;
  if (nglobal + groups == 0) return 0;
  PartReduce k = a;
  if (!groups) k.sn = 0;
  hipLaunchKernelGGL(k_part_reduce, dim3(nglobal + groups), dim3(256), 0, s, k, nglobal);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_block_bwd(const BlockBwd& a, hipStream_t s) {
  DQ_REQUIRE(a.u && a.dy && a.du && a.rows > 0 && a.n > 0, "block_bwd: missing operand");
  DQ_REQUIRE(!a.ss || a.rows % a.rows_per_sample == 0, "block_bwd: rows must be a multiple of rows_per_sample");
  DQ_REQUIRE(!a.ss || a.dss, "block_bwd: scale/shift needs a gradient buffer");
  BlockBwd a2 = a;
  if (!a.ss) a2.rows_per_sample = std::max(1, std::min(a.rows, 16384 / std::max(1, a.n)));  // no per-sample state: regroup freely
  const BlockBwd& a_ = a2;
  const int B = cdiv(a_.rows, a_.rows_per_sample);
  const int64_t per_sample = (int64_t)a_.rows_per_sample * a_.n;
  // one item per thread while that stays within 64 blocks per sample (the bottleneck's PreNorm backward -- 12,800 items -- ran on 13
  // workgroups at four items per thread: 19 us, a latency floor on 5 % of the CUs); larger tensors: 64 blocks per sample, a strided loop
  int bps = std::max(1, std::min(64, cdiv(per_sample, 256)));
  const bool need_sums = a.dg || a.dss || a.dbias;
  // (never more blocks than the caller's partial-sum slot holds: the arena sizes it for the per-sample grouping, the regrouped launches of
  // short tensors can ask for more)
  if (need_sums && a.part && a.part_floats >= (int64_t)B * 4 * a.C) bps = (int)std::min<int64_t>(bps, a.part_floats / ((int64_t)B * 4 * a.C));
  dim3 grid(bps, B), block(256);
  DQ_REQUIRE(!need_sums || (a.part && a.part_floats >= (int64_t)bps * B * 4 * a.C), "block_bwd: partial-sum slot missing or too small");
  if (!need_sums) a2.part = nullptr;
#define DQ_BB(CC)                                                     \
  case CC:                                                            \
    hipLaunchKernelGGL((k_block_bwd<CC>), grid, block, 0, s, a_);     \
    break;
  switch (a.C) {
    DQ_BB(1) DQ_BB(4) DQ_BB(8) DQ_BB(12) DQ_BB(16) DQ_BB(32)
    default:
      set_error("block_bwd: unsupported channel count " + std::to_string(a.C));
      return 2;
  }
#undef DQ_BB
  DQ_LAUNCH_CHECK();
  if (!need_sums) return 0;
  // the ordered sums of the per-block partials, right behind the kernel on the same stream (the slot is free again afterwards)
  PartReduce r;
  r.part = a.part; r.B = B; r.gx = bps; r.nv = 4 * a.C;
  if (a.dg) { r.seg_start[r.nseg] = 0; r.seg_len[r.nseg] = a.C; r.seg_dst[r.nseg] = a.dg; ++r.nseg; }
  if (a.dbias) { r.seg_start[r.nseg] = 3 * a.C; r.seg_len[r.nseg] = a.C; r.seg_dst[r.nseg] = a.dbias; ++r.nseg; }
  if (a.dss) { r.s0 = a.C; r.sn = 2 * a.C; r.sdst = a.dss; r.sstride = a.ss_stride; }
  if (a.defer_reduce) { *a.defer_reduce = r; return 0; }
  return launch_part_reduce(r, s);
}

// -------------------------------------------------------------------------------------------------
// data gradient: thread = (row, input position m), NCI input channels per thread (grid.y = channel chunks)
//   dX[ci][m] (+)= sum_co sum_(p,k : tap(p,k) == m) W[co][ci][k] * dU[co][p]
// -------------------------------------------------------------------------------------------------
template <int NCI, int K, int MODE>
__global__ void __launch_bounds__(256) k_conv_bwd_data(ConvBwdData a) {
  const int64_t item = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.rows * a.n_in;
  const int cin = a.cinA + a.cinB;
  const int ci0 = blockIdx.y * NCI;
  // This block's weights W[co][ci0 .. ci0 + NCI)[k] staged in LDS once: read from memory inside the output-channel loop they are
  // a scalar load per iteration that nothing overlaps -- at the deep levels (a few hundred waves) the kernel was a chain of
  // cout such round trips, 30-50 us whatever the level's size
  constexpr int WSH = 4096;
  __shared__ float wsh[WSH];
  const bool use_lds = a.cout * NCI * K <= WSH;
  if (use_lds) {
    constexpr int NIT = WSH / 256;  // (as in k_conv_fwd: one memory round trip)
    const int tot = a.cout * NCI * K;
    float v[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int co = i / (NCI * K), r = i - co * (NCI * K), ci = ci0 + r / K, k = r % K;
      v[u] = 0.f;
      if (u * 256 < tot) v[u] = a.w[(i < tot && ci < cin) ? (co * cin + ci) * K + k : 0];  // (wave-uniform)
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int i = u * 256 + (int)threadIdx.x;
      const int r = i % (NCI * K), ci = ci0 + r / K;
      if (i < tot) wsh[i] = ci < cin ? v[u] : 0.f;
    }
    __syncthreads();
  }
  if (item >= total) return;
  const int row = (int)(item / a.n_in), m = (int)(item % a.n_in);

  // contributing (p, k) pairs of input position m
  constexpr int NT = (MODE == CONV_S1) ? K : (MODE == CONV_DOWN ? 2 : 6);
  int tp[NT], tk[NT];
  if (MODE == CONV_S1) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int p = m - k + (K - 1) / 2;
      tk[k] = k;
      tp[k] = (p >= 0 && p < a.n_out) ? p : -1;
    }
  } else if (MODE == CONV_DOWN) {  // 2p + k - 1 == m
    const int par = (m + 1) & 1;   // k has the parity of m+1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = par + 2 * j;
      const int p2 = m + 1 - k;  // = 2p, even by construction
      const int p = p2 >> 1;
      tk[j] = k;
      tp[j] = (p2 >= 0 && p < a.n_out) ? p : -1;
    }
  } else {  // upsampled positions j in {2m, 2m+1}; j == p + k - 1
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int p = 2 * m + jj + 1 - k;
        tk[jj * 3 + k] = k;
        tp[jj * 3 + k] = (p >= 0 && p < a.n_out) ? p : -1;
      }
    }
  }

  float acc[NCI];
#pragma unroll
  for (int i = 0; i < NCI; ++i) acc[i] = 0.f;
  if (use_lds) {
#pragma unroll 2
    for (int co = 0; co < a.cout; ++co) {
      const float* du = a.du + ((int64_t)row * a.cout + co) * a.n_out;
      const float* w = wsh + co * (NCI * K);
      float dv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) dv[t] = tp[t] >= 0 ? du[tp[t]] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < NCI; ++i) acc[i] = fmaf(w[i * K + tk[t]], dv[t], acc[i]);  // (zero weights beyond cin)
    }
  } else {
    for (int co = 0; co < a.cout; ++co) {
      const float* du = a.du + ((int64_t)row * a.cout + co) * a.n_out;
      const float* w = a.w + (int64_t)co * cin * K;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float dv = tp[t] >= 0 ? du[tp[t]] : 0.f;
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
          const int ci = ci0 + i;
          if (ci < cin) acc[i] = fmaf(w[(int64_t)ci * K + tk[t]], dv, acc[i]);
        }
      }
    }
  }
  // the old values of an accumulating launch are all requested before the first store (a `*dst += v` per channel is NCI serial
  // round trips: the next load may not pass the previous store)
  float* dsts[NCI];
  float oldv[NCI];
#pragma unroll
  for (int i = 0; i < NCI; ++i) {
    const int ci = ci0 + i;
    float* dst = nullptr;
    if (ci < a.cinA) { if (a.dinA) dst = a.dinA + ((int64_t)row * a.cinA + ci) * a.n_in + m; }
    else if (ci < cin) { if (a.dinB) dst = a.dinB + ((int64_t)row * a.cinB + (ci - a.cinA)) * a.n_in + m; }
    dsts[i] = dst;
    oldv[i] = (dst && a.accumulate) ? *dst : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NCI; ++i)
    if (dsts[i]) *dsts[i] = oldv[i] + acc[i];
}

int launch_conv_bwd_data(const ConvBwdData& a, hipStream_t s) {
  DQ_REQUIRE(a.du && a.w && a.cout > 0 && a.cinA > 0, "conv_bwd_data: missing operand");
  const int cin = a.cinA + a.cinB;
  const int64_t total = (int64_t)a.rows * a.n_in;
  if (total == 0) return 0;
  constexpr int NCI = 4;
  dim3 grid(cdiv(total, 256), cdiv(cin, NCI)), block(256);
#define DQ_BD(KK, MM)                                                                     \
  if (a.K == KK && a.mode == MM) {                                                         \
    hipLaunchKernelGGL((k_conv_bwd_data<NCI, KK, MM>), grid, block, 0, s, a);              \
    DQ_LAUNCH_CHECK();                                                                     \
    return 0;                                                                              \
  }
  DQ_BD(1, CONV_S1)
  DQ_BD(3, CONV_S1)
  DQ_BD(7, CONV_S1)
  DQ_BD(4, CONV_DOWN)
  DQ_BD(3, CONV_UP)
#undef DQ_BD
  set_error("conv_bwd_data: unsupported (K, mode)");
  return 2;
}

// -------------------------------------------------------------------------------------------------
// weight gradient, two deterministic stages (no atomics, fixed summation order => bitwise run-to-run repeatable):
//  (1) block (x, y): y owns a (COB x CIB x K) sub-block of dW, x a strided share of the (row, p) items; the sub-block
//      lives in registers, is reduced over the block (wave shuffles + LDS) and written to partials[x][element];
//  (2) k_wgrad_reduce sums the <= 64 partials of every element in order and adds them to dW / dbias.
// -------------------------------------------------------------------------------------------------
// block-level reduction of the per-thread (COB x CIB x K) sub-block + bias sums -> this block's partial slot
template <int COB, int CIB, int K>
__device__ __forceinline__ void wgrad_block_reduce(float (&acc)[COB][CIB][K], float (&accb)[COB], const ConvWgrad& a, int co0, int ci0,
                                                   int cin, int nelem_w) {
  constexpr int NACC = COB * CIB * K + COB;
  __shared__ float red[4][NACC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < COB; ++i) {
#pragma unroll
    for (int j = 0; j < CIB; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float v = wave_sum(acc[i][j][k]);
        if (lane == 0) red[wv][(i * CIB + j) * K + k] = v;
      }
    const float vb = wave_sum(accb[i]);
    if (lane == 0) red[wv][COB * CIB * K + i] = vb;
  }
  __syncthreads();
  float* part = a.scratch + (int64_t)blockIdx.x * (nelem_w + a.cout);
  for (int e = threadIdx.x; e < NACC; e += blockDim.x) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (e < COB * CIB * K) {
      const int i = e / (CIB * K), j = (e / K) % CIB, k = e % K;
      const int co = co0 + i, ci = ci0 + j;
      if (co < a.cout && ci < cin) part[((int64_t)co * cin + ci) * K + k] = v;
    } else if (ci0 == 0) {
      const int co = co0 + (e - COB * CIB * K);
      if (co < a.cout) part[nelem_w + co] = v;
    }
  }
}

template <int COB, int CIB, int K, int MODE>
__device__ __forceinline__ void wgrad_body(const ConvWgrad& a, int n_cib, int nelem_w, int gx, int by) {
  const int cin = a.cinA + a.cinB;
  const int co0 = (by / n_cib) * COB, ci0 = (by % n_cib) * CIB;
  const int total = a.rows * a.n_out;  // < 2^31 (checked on the host)
  float acc[COB][CIB][K];
  float accb[COB];
#pragma unroll
  for (int i = 0; i < COB; ++i) {
    accb[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CIB; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[i][j][k] = 0.f;
  }
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gx * blockDim.x) {
    const int row = it / a.n_out, p = it - row * a.n_out;
    float d[COB];
    const float* dub = a.du + ((int64_t)row * a.cout + co0) * a.n_out + p;
#pragma unroll
    for (int i = 0; i < COB; ++i) d[i] = (co0 + i < a.cout) ? dub[(int64_t)i * a.n_out] : 0.f;
    int q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = tap_pos<MODE, K>(p, k, a.n_in);
#pragma unroll
    for (int j = 0; j < CIB; ++j) {
      const int ci = ci0 + j;
      if (ci >= cin) continue;
      const float* src = (ci < a.cinA) ? a.inA + ((int64_t)row * a.cinA + ci) * a.n_in
                                       : a.inB + ((int64_t)row * a.cinB + (ci - a.cinA)) * a.n_in;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float xv = q[k] >= 0 ? src[q[k]] : 0.f;
#pragma unroll
        for (int i = 0; i < COB; ++i) acc[i][j][k] = fmaf(d[i], xv, acc[i][j][k]);
      }
    }
#pragma unroll
    for (int i = 0; i < COB; ++i) accb[i] += d[i];
  }
  wgrad_block_reduce<COB, CIB, K>(acc, accb, a, co0, ci0, cin, nelem_w);
}

// stride-1 'same' convs with n % 4 == 0 (K = 1 or 3): a thread takes 4 consecutive positions per step -- 16-byte loads of
// du and x (+ the two halo values), a quarter of the memory instructions and index arithmetic of the scalar form
template <int COB, int CIB, int K, int MODE>
__global__ void __launch_bounds__(256) k_conv_wgrad(ConvWgrad a, int n_cib, int nelem_w) {
  wgrad_body<COB, CIB, K, MODE>(a, n_cib, nelem_w, gridDim.x, blockIdx.y);
}

template <int COB, int CIB, int K>
__device__ __forceinline__ void wgrad_body_v4(const ConvWgrad& a, int n_cib, int nelem_w, int gx, int by) {
  static_assert(K == 1 || K == 3 || K == 7, "vectorised weight gradient: K = 1, 3 or 7");
  constexpr int H = (K - 1) / 2;
  const int cin = a.cinA + a.cinB;
  const int co0 = (by / n_cib) * COB, ci0 = (by % n_cib) * CIB;
  const int n = a.n_out, n4 = n >> 2;
  const int total4 = a.rows * n4;
  float acc[COB][CIB][K];
  float accb[COB];
#pragma unroll
  for (int i = 0; i < COB; ++i) {
    accb[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CIB; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[i][j][k] = 0.f;
  }
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < total4; it += gx * blockDim.x) {
    const int row = it / n4, p = (it - row * n4) << 2;
    float d[COB][4];
    const float* dub = a.du + ((int64_t)row * a.cout + co0) * n + p;
#pragma unroll
    for (int i = 0; i < COB; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (co0 + i < a.cout) v = *reinterpret_cast<const float4*>(dub + (int64_t)i * n);
      d[i][0] = v.x; d[i][1] = v.y; d[i][2] = v.z; d[i][3] = v.w;
      accb[i] += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int j = 0; j < CIB; ++j) {
      const int ci = ci0 + j;
      if (ci >= cin) continue;
      const float* src = (ci < a.cinA) ? a.inA + ((int64_t)row * a.cinA + ci) * n : a.inB + ((int64_t)row * a.cinB + (ci - a.cinA)) * n;
      const float4 xm = *reinterpret_cast<const float4*>(src + p);
      float win[4 + 2 * H];
      win[H + 0] = xm.x; win[H + 1] = xm.y; win[H + 2] = xm.z; win[H + 3] = xm.w;
#pragma unroll
      for (int h = 0; h < H; ++h) {  // the halo: H positions either side (zero padding at the row ends)
        const int ql = p - H + h, qr = p + 4 + h;
        win[h] = ql >= 0 ? src[ql] : 0.f;
        win[4 + H + h] = qr < n ? src[qr] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < COB; ++i) {
          float t = acc[i][j][k];
#pragma unroll
          for (int q = 0; q < 4; ++q) t = fmaf(d[i][q], win[q + k], t);
          acc[i][j][k] = t;
        }
    }
  }
  wgrad_block_reduce<COB, CIB, K>(acc, accb, a, co0, ci0, cin, nelem_w);
}

template <int COB, int CIB, int K>
__global__ void __launch_bounds__(256) k_conv_wgrad_v4(ConvWgrad a, int n_cib, int nelem_w) {
  wgrad_body_v4<COB, CIB, K>(a, n_cib, nelem_w, gridDim.x, blockIdx.y);
}

// Up to three stride-1 convs over the SAME (rows, n) in one launch (blockIdx.z picks the conv): the two convs and the
// residual 1x1 conv of a ResnetBlock.  The weight-gradient launches are short and latency-bound -- ~200 of them per train
// step on the side stream -- so a ResnetBlock now costs 2 launches (this + the merged reduce) instead of 6.
struct WgradMulti {
  ConvWgrad c[3];
  int n_cib[3], nelem_w[3], gx[3], tiles[3];
};
template <bool VEC4>
__global__ void __launch_bounds__(256) k_conv_wgrad_multi(WgradMulti m) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x >= m.gx[z] || (int)blockIdx.y >= m.tiles[z]) return;
  const ConvWgrad& a = m.c[z];
  if (VEC4) {
    if (a.K == 3) wgrad_body_v4<4, 4, 3>(a, m.n_cib[z], m.nelem_w[z], m.gx[z], blockIdx.y);
    else wgrad_body_v4<4, 4, 1>(a, m.n_cib[z], m.nelem_w[z], m.gx[z], blockIdx.y);
  } else {
    if (a.K == 3) wgrad_body<4, 4, 3, CONV_S1>(a, m.n_cib[z], m.nelem_w[z], m.gx[z], blockIdx.y);
    else wgrad_body<4, 4, 1, CONV_S1>(a, m.n_cib[z], m.nelem_w[z], m.gx[z], blockIdx.y);
  }
}

// dW[e] += sum over the per-block partials in a fixed order (deterministic).  A block owns 16 consecutive elements x 16
// partial groups: thread (e, g) sums partials g, g+16, ... (64-byte segments: every fetched sector is fully used), the 16
// group sums meet in LDS.
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ part, int nblk, int nelem_w, int cout,
                                                      float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const int nelem = nelem_w + cout;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem) {
    int b = g;
    for (; b + 16 < nblk; b += 32) {
      s0 += part[(int64_t)b * nelem + e];
      s1 += part[(int64_t)(b + 16) * nelem + e];
    }
    if (b < nblk) s0 += part[(int64_t)b * nelem + e];
  }
  red[g][el] = s0 + s1;
  __syncthreads();
  if (g == 0 && e < nelem) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    if (e < nelem_w) dw[e] += s;
    else if (dbias) dbias[e - nelem_w] += s;
  }
}

__global__ void __launch_bounds__(256) k_wgrad_reduce_multi(WgradMulti m) {
  const int z = blockIdx.z;
  const ConvWgrad& a = m.c[z];
  const int nelem_w = m.nelem_w[z], nelem = nelem_w + a.cout, nblk = m.gx[z];
  if ((int)blockIdx.x * 16 >= nelem) return;  // uniform per block
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem) {
    int b = g;
    for (; b + 16 < nblk; b += 32) {
      s0 += a.scratch[(int64_t)b * nelem + e];
      s1 += a.scratch[(int64_t)(b + 16) * nelem + e];
    }
    if (b < nblk) s0 += a.scratch[(int64_t)b * nelem + e];
  }
  red[g][el] = s0 + s1;
  __syncthreads();
  if (g == 0 && e < nelem) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    if (e < nelem_w) a.dw[e] += s;
    else if (a.dbias) a.dbias[e - nelem_w] += s;
  }
}

// ---- The same three weight gradients for the DEEP levels (m/z rows of 1 .. 8 positions, <= 16 output and <= 32 input channels) on the
// matrix pipe: dW[co][ci][k] = sum_(row, p) dU[co][p] x[ci][p + k - 1] is a product whose K index is the m/z ROW, so with
// v_mfma_f32_16x16x4_f32 (A: lane l supplies A[l % 16][l / 16], B: B[l / 16][l % 16]) a wave takes 16 rows per step and both operands are
// loaded from memory ALREADY in operand order -- lane (g, i) reads dU[row 4 s + g][channel i][p] resp. x[row 4 s + g][channel i][q] for K-step
// s: no staging, no transposes, every load of a 16-row tile in flight at once.  The four waves of a workgroup take the (conv, input-channel
// tile) jobs of the block: conv2 | conv1, channels 0..15 | conv1, channels 16..31 | res_conv (both tiles); a workgroup walks its tiles and
// leaves its sums in block `blockIdx.x` of each conv's scratch region, in the layout k_wgrad_reduce_multi sums.
// k_conv_wgrad_multi walks the same tensors once per (4 x 4) channel block -- 18 .. 32 passes, per-thread accumulators, a block reduction --
// and its per-lane 4-byte accesses at a row pitch of C n floats touch 32 cache lines per instruction: 30 .. 70 us per launch next to the main
// chain's kernels for 2.4 MB of operands.  With those launches SKIPPED the train step is 0.114 ms shorter (3.472 -> 3.358 ms,
// profiles/r05_ab_skip_deep_wgrad.log): that is what the side queue's largest tenant costs the main chain.
typedef float wr_f32x4 __attribute__((ext_vector_type(4)));
template <int N>
__global__ void __launch_bounds__(256) k_wgrad_rows(WgradMulti m, int count, int ntiles) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's jobs: (conv z, input-channel tile t); wave 3 takes both tiles of the res_conv
  int jz[2] = {-1, -1}, jt[2] = {0, 0};
  if (wv == 0) { jz[0] = 0; if (m.c[0].cinA + m.c[0].cinB > 16) { jz[1] = 0; jt[1] = 1; } }
  else if (wv == 1) { if (count > 1) jz[0] = 1; }
  else if (wv == 2) { if (count > 1 && m.c[1].cinA + m.c[1].cinB > 16) { jz[0] = 1; jt[0] = 1; } }
  else { if (count > 2) { jz[0] = 2; if (m.c[2].cinA + m.c[2].cinB > 16) { jz[1] = 2; jt[1] = 1; } } }
  wr_f32x4 acc[2][3];
  float bsum[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[j][k] = wr_f32x4{0.f, 0.f, 0.f, 0.f};
  const int rows = m.c[0].rows;
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (jz[j] < 0) continue;  // (wave-uniform)
      const ConvWgrad& a = m.c[jz[j]];
      const int cin = a.cinA + a.cinB, ci = 16 * jt[j] + i;
      const bool co_ok = i < a.cout, ci_ok = ci < cin;
      const bool fromA = ci < a.cinA || !ci_ok;  // (a lane beyond the conv's input channels reads channel 0 of inA -- inB may be null -- and is zeroed)
      float A[N][4], B[N][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = tile * 16 + 4 * s + g;
        const bool rok = row < rows;
        const int rc = rok ? row : rows - 1;
        // (no load is predicated: a clamped address, the select afterwards)
        const float* du = a.du + ((int64_t)rc * a.cout + (co_ok ? i : 0)) * N;
        const float* xs = fromA ? a.inA + ((int64_t)rc * a.cinA + (ci_ok ? ci : 0)) * N : a.inB + ((int64_t)rc * a.cinB + (ci - a.cinA)) * N;
#pragma unroll
        for (int p = 0; p < N; ++p) { A[p][s] = du[p]; B[p][s] = xs[p]; }
#pragma unroll
        for (int p = 0; p < N; ++p) { A[p][s] = (rok && co_ok) ? A[p][s] : 0.f; B[p][s] = (rok && ci_ok) ? B[p][s] : 0.f; }
      }
      if (a.K == 3) {
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int q = p + k - 1;
            if (q < 0 || q >= N) continue;  // (zero padding at the row ends; compile time)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p][s], B[q][s], acc[j][k], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p][s], B[p][s], acc[j][0], 0, 0, 0);
      }
      if (jt[j] == 0) {
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
          for (int s = 0; s < 4; ++s) bsum[j] += A[p][s];
      }
    }
  }
  // ---- this workgroup's block of each conv's partials: [dw (cout, cin, K) | dbias (cout)]; D register r of lane (g, j) = dW[co = 4 g + r][ci = 16 t + j]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (jz[j] < 0) continue;
    const ConvWgrad& a = m.c[jz[j]];
    const int cin = a.cinA + a.cinB, ci = 16 * jt[j] + i, nelem_w = m.nelem_w[jz[j]];
    float* part = a.scratch + (int64_t)blockIdx.x * (nelem_w + a.cout);
    if (ci < cin) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= a.K) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 4 * g + r;
          if (co < a.cout) part[((int64_t)co * cin + ci) * a.K + k] = acc[j][k][r];
        }
      }
    }
    if (jt[j] == 0) {  // bias: lane (g, i) holds the sum over its rows 4 s + g of channel i; the four lane groups meet through v_permlane swaps
      float t = bsum[j];
      const auto x1 = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
      t = __int_as_float(x1[0]) + __int_as_float(x1[1]);
      const auto x2 = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
      t = __int_as_float(x2[0]) + __int_as_float(x2[1]);
      if (g == 0 && i < a.cout) part[nelem_w + i] = t;
    }
  }
}
static bool wgrad_rows_usable(const ConvWgrad* w, int count) {
  const int n = w[0].n_out;
  if (!(n == 1 || n == 2 || n == 4 || n == 8)) return false;
  for (int z = 0; z < count; ++z)
    if (w[z].cout > 16 || w[z].cout < 12 || w[z].cinA + w[z].cinB > 32 || w[z].mode != CONV_S1 || !(w[z].K == 1 || w[z].K == 3) || (w[z].cinB > 0 && !w[z].inB)) return false;
  // the job table: conv 0 and conv 1 are the block's k3 convs, conv 2 (if any) its 1x1 residual conv
  return count >= 2 && w[0].K == 3 && w[1].K == 3 && (count < 3 || w[2].K == 1);
}

// blocks in flight per conv of a weight-gradient launch (2048: settled in round 2 over 512 .. 4096)
static int wgrad_blocks() { return 2048; }

// count <= 3 stride-1 convs with identical (rows, n_in == n_out); each descriptor carries its own scratch region
int launch_conv_wgrad_multi(const ConvWgrad* w, int count, hipStream_t s) {
  DQ_REQUIRE(count >= 1 && count <= 3, "conv_wgrad_multi: 1..3 convs");
  WgradMulti m;
  int gx_max = 1, tiles_max = 1, nelem_max = 1;
  const int64_t total = (int64_t)w[0].rows * w[0].n_out;
  if (total == 0) return 0;
  DQ_REQUIRE(total < (1ll << 31), "conv_wgrad: rows*n must be below 2^31");
  for (int i = 0; i < count; ++i)
    DQ_REQUIRE(w[i].du && w[i].inA && w[i].dw && w[i].scratch && w[i].cout > 0 && w[i].cinA > 0 && w[i].rows == w[0].rows &&
               w[i].n_in == w[i].n_out && w[i].n_out == w[0].n_out, "conv_wgrad_multi: missing operand / different rows");
  const bool vec4 = w[0].n_out % 4 == 0;
  for (int i = 0; i < count; ++i) {
    const ConvWgrad& a = w[i];
    DQ_REQUIRE(a.du && a.inA && a.dw && a.scratch && a.cout > 0 && a.cinA > 0, "conv_wgrad_multi: missing operand");
    DQ_REQUIRE(a.mode == CONV_S1 && (a.K == 1 || a.K == 3) && a.n_in == a.n_out && a.rows == w[0].rows && a.n_out == w[0].n_out,
               "conv_wgrad_multi: stride-1 k1/k3 convs over the same rows");
    const int cin = a.cinA + a.cinB;
    const int n_cob = cdiv(a.cout, 4), n_cib = cdiv(cin, 4);
    m.c[i] = a;
    m.n_cib[i] = n_cib;
    m.nelem_w[i] = a.cout * cin * a.K;
    m.tiles[i] = n_cob * n_cib;
    m.gx[i] = std::max(1, std::min({cdiv(total, 256 * (vec4 ? 8 : 4)), WGRAD_MAX_PARTS, std::max(1, wgrad_blocks() / m.tiles[i])}));
    DQ_REQUIRE((int64_t)m.gx[i] * (m.nelem_w[i] + a.cout) <= a.scratch_floats, "conv_wgrad_multi: scratch too small");
    gx_max = std::max(gx_max, m.gx[i]); tiles_max = std::max(tiles_max, m.tiles[i]);
    nelem_max = std::max(nelem_max, m.nelem_w[i] + a.cout);
  }
  for (int i = count; i < 3; ++i) { m.c[i] = w[0]; m.n_cib[i] = 1; m.nelem_w[i] = 0; m.gx[i] = 0; m.tiles[i] = 0; }
  if (wgrad_rows_usable(w, count)) {  // the deep levels: 16 rows per K-step on the matrix pipe (k_wgrad_rows)
    const int ntiles = cdiv(w[0].rows, 16);
    int gx = std::min(ntiles, 256);
    for (int i = 0; i < count; ++i) gx = (int)std::min<int64_t>(gx, w[i].scratch_floats / (m.nelem_w[i] + w[i].cout));
    DQ_REQUIRE(gx >= 1, "conv_wgrad_multi: scratch too small");
    for (int i = 0; i < count; ++i) m.gx[i] = gx;
    const int n = w[0].n_out;
    if (n == 1) hipLaunchKernelGGL(k_wgrad_rows<1>, dim3(gx), dim3(256), 0, s, m, count, ntiles);
    else if (n == 2) hipLaunchKernelGGL(k_wgrad_rows<2>, dim3(gx), dim3(256), 0, s, m, count, ntiles);
    else if (n == 4) hipLaunchKernelGGL(k_wgrad_rows<4>, dim3(gx), dim3(256), 0, s, m, count, ntiles);
    else hipLaunchKernelGGL(k_wgrad_rows<8>, dim3(gx), dim3(256), 0, s, m, count, ntiles);
    DQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3(cdiv(nelem_max, 16), 1, count), dim3(256), 0, s, m);
    DQ_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(gx_max, tiles_max, count), block(256);
  if (vec4) hipLaunchKernelGGL(k_conv_wgrad_multi<true>, grid, block, 0, s, m);
  else hipLaunchKernelGGL(k_conv_wgrad_multi<false>, grid, block, 0, s, m);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3(cdiv(nelem_max, 16), 1, count), dim3(256), 0, s, m);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_conv_wgrad(const ConvWgrad& a, hipStream_t s) {
  DQ_REQUIRE(a.du && a.inA && a.dw && a.scratch && a.cout > 0 && a.cinA > 0, "conv_wgrad: missing operand");
  const int cin = a.cinA + a.cinB;
  const int64_t total = (int64_t)a.rows * a.n_out;
  if (total == 0) return 0;
  DQ_REQUIRE(total < (1ll << 31), "conv_wgrad: rows*n must be below 2^31");
  constexpr int COB = 4, CIB = 4;
  const int n_cob = cdiv(a.cout, COB), n_cib = cdiv(cin, CIB);
  const int nelem_w = a.cout * cin * a.K;
  // <= WGRAD_MAX_PARTS partial blocks per element; ~4 items per thread, ~2048 blocks in flight where the problem allows it
  // (K = 7 -- init_conv and the first MS1 conv -- joined in round 4: as one position per thread and step the init_conv launch was six dependent
  // memory round trips per thread, 25 us at the tail of the side queue, which the end of the backward waits for)
  const bool vec4 = a.mode == CONV_S1 && (a.K == 1 || a.K == 3 || a.K == 7) && a.n_out % 4 == 0 && a.n_in == a.n_out;
  const int gx = std::max(1, std::min({cdiv(total, 256 * (vec4 ? 8 : 4)), WGRAD_MAX_PARTS, std::max(1, wgrad_blocks() / (n_cob * n_cib))}));
  DQ_REQUIRE((int64_t)gx * (nelem_w + a.cout) <= a.scratch_floats, "conv_wgrad: scratch too small");
  dim3 grid(gx, n_cob * n_cib), block(256);
  if (vec4) {
    if (a.K == 1) hipLaunchKernelGGL((k_conv_wgrad_v4<COB, CIB, 1>), grid, block, 0, s, a, n_cib, nelem_w);
    else if (a.K == 3) hipLaunchKernelGGL((k_conv_wgrad_v4<COB, CIB, 3>), grid, block, 0, s, a, n_cib, nelem_w);
    else hipLaunchKernelGGL((k_conv_wgrad_v4<COB, CIB, 7>), grid, block, 0, s, a, n_cib, nelem_w);
    DQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(nelem_w + a.cout, 16)), dim3(256), 0, s, a.scratch, gx, nelem_w, a.cout, a.dw, a.dbias);
    DQ_LAUNCH_CHECK();
    return 0;
  }
#define DQ_WG(KK, MM)                                                                                  \
  if (a.K == KK && a.mode == MM) {                                                                      \
    hipLaunchKernelGGL((k_conv_wgrad<COB, CIB, KK, MM>), grid, block, 0, s, a, n_cib, nelem_w);         \
    DQ_LAUNCH_CHECK();                                                                                  \
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(nelem_w + a.cout, 16)), dim3(256), 0, s, a.scratch, gx, nelem_w, a.cout, a.dw, \
                       a.dbias);                                                                        \
    DQ_LAUNCH_CHECK();                                                                                  \
    return 0;                                                                                           \
  }
  DQ_WG(1, CONV_S1)
  DQ_WG(3, CONV_S1)
  DQ_WG(7, CONV_S1)
  DQ_WG(4, CONV_DOWN)
  DQ_WG(3, CONV_UP)
#undef DQ_WG
  set_error("conv_wgrad: unsupported (K, mode)");
  return 2;
}

// -------------------------------------------------------------------------------------------------
// small helpers
// -------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_rmsnorm_fwd(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ y,
                                                     int rows, int n) {
  const int64_t item = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (item >= (int64_t)rows * n) return;
  const int row = (int)(item / n), p = (int)(item % n);
  const int64_t base = (int64_t)row * C * n + p;
  float v[C];
  float ssq = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { v[c] = x[base + (int64_t)c * n]; ssq = fmaf(v[c], v[c], ssq); }
  const float inv = rms_inv(ssq, sqrtf((float)C));
#pragma unroll
  for (int c = 0; c < C; ++c) y[base + (int64_t)c * n] = v[c] * inv * g[c];
}

int launch_rmsnorm_fwd(const float* x, const float* g, float* y, int C, int rows, int n, hipStream_t s) {
  const int64_t total = (int64_t)rows * n;
  if (total == 0) return 0;
  dim3 grid(cdiv(total, 256)), block(256);
  switch (C) {
    case 4: hipLaunchKernelGGL((k_rmsnorm_fwd<4>), grid, block, 0, s, x, g, y, rows, n); break;
    case 8: hipLaunchKernelGGL((k_rmsnorm_fwd<8>), grid, block, 0, s, x, g, y, rows, n); break;
    case 12: hipLaunchKernelGGL((k_rmsnorm_fwd<12>), grid, block, 0, s, x, g, y, rows, n); break;
    case 16: hipLaunchKernelGGL((k_rmsnorm_fwd<16>), grid, block, 0, s, x, g, y, rows, n); break;
    case 32: hipLaunchKernelGGL((k_rmsnorm_fwd<32>), grid, block, 0, s, x, g, y, rows, n); break;
    default: set_error("rmsnorm_fwd: unsupported channel count " + std::to_string(C)); return 2;
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

// in: to_mid ? (B*RT, cn) rows : (B, cn, P).  Small tensor (B*RT*cn floats); plain index transpose.
__global__ void __launch_bounds__(256) k_fold(const float* __restrict__ in, float* __restrict__ out, int B, int RT, int cn, int to_mid,
                                              int add, int P) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // i indexes the OUTPUT contiguously; the (B, cn, .) side has row pitch P >= RT
  if (to_mid) {  // out (B, cn, P)
    if (i >= (int64_t)B * P * cn) return;
    const int rt = (int)(i % P);
    const int c = (int)((i / P) % cn);
    const int b = (int)(i / ((int64_t)P * cn));
    if (rt >= RT) { out[i] = 0.f; return; }  // pad column
    const int64_t src = ((int64_t)b * RT + rt) * cn + c;
    out[i] = add ? out[i] + in[src] : in[src];
  } else {  // out (B*RT, cn)
    if (i >= (int64_t)B * RT * cn) return;
    const int c = (int)(i % cn);
    const int rt = (int)((i / cn) % RT);
    const int b = (int)(i / ((int64_t)RT * cn));
    const int64_t src = ((int64_t)b * cn + c) * P + rt;
    out[i] = add ? out[i] + in[src] : in[src];
  }
}

int launch_fold(const float* in, float* out, int B, int RT, int cn, int to_mid, int add, hipStream_t s, int pitch) {
  const int P = pitch > 0 ? pitch : RT;
  const int64_t total = (int64_t)B * (to_mid ? P : RT) * cn;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_fold, dim3(cdiv(total, 256)), dim3(256), 0, s, in, out, B, RT, cn, to_mid, add, P);
  DQ_LAUNCH_CHECK();
  return 0;
}

// cat0[row][0][mz] = (cond*cm+ca)*(scale_b+1)+shift_b ; cat0[row][1][mz] = x ; ms1n = ms1*cm+ca
// (reference unet1d.py:1107-1115 with the normalisation of model.py:310-311 / 350-351 folded in)
__global__ void __launch_bounds__(256) k_prep_inputs(const float* __restrict__ x, const float* __restrict__ cond,
                                                     const float* __restrict__ ms1, const float* __restrict__ ss, int ss_stride,
                                                     int ss_off, float cm, float ca, float* __restrict__ cat0,
                                                     float* __restrict__ ms1n, int B, int RT, int MZ) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * RT * MZ;
  if (ms1n && i < (int64_t)B * RT) ms1n[i] = fmaf(ms1[i], cm, ca);
  if (i >= total) return;
  const int mz = (int)(i % MZ);
  const int64_t row = i / MZ;
  const int b = (int)(row / RT);
  const float sc = ss[(int64_t)b * ss_stride + ss_off], sh = ss[(int64_t)b * ss_stride + ss_off + 1];
  const float cn = fmaf(cond[i], cm, ca);
  cat0[(row * 2 + 0) * MZ + mz] = cn * (sc + 1.0f) + sh;
  cat0[(row * 2 + 1) * MZ + mz] = x[i];
}

int launch_prep_inputs(const float* x, const float* cond, const float* ms1, const float* ss, int ss_stride, int ss_off, float cm,
                       float ca, float* cat0, float* ms1n, int B, int RT, int MZ, hipStream_t s) {
  const int64_t total = (int64_t)B * RT * MZ;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_prep_inputs, dim3(cdiv(total, 256)), dim3(256), 0, s, x, cond, ms1, ss, ss_stride, ss_off, cm, ca, cat0,
                     ms1n, B, RT, MZ);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ms1n = ms1 * cm + ca alone (the first layer's other work rides in the level kernel's input stage in inference)
__global__ void __launch_bounds__(256) k_ms1_norm(const float* __restrict__ ms1, float cm, float ca, float* __restrict__ ms1n, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) ms1n[i] = fmaf(ms1[i], cm, ca);
}
int launch_ms1_norm(const float* ms1, float cm, float ca, float* ms1n, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_ms1_norm, dim3(cdiv(n, 256)), dim3(256), 0, s, ms1, cm, ca, ms1n, n);
  DQ_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_prep_inputs_bwd(const float* __restrict__ dcat0, const float* __restrict__ cond, float cm,
                                                         float ca, float* __restrict__ part, int RT, int MZ) {
  const int b = blockIdx.y;
  const int64_t per = (int64_t)RT * MZ;
  float dsc = 0.f, dsh = 0.f;
  for (int64_t it = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; it < per; it += (int64_t)gridDim.x * blockDim.x) {
    const int mz = (int)(it % MZ);
    const int64_t row = (int64_t)b * RT + it / MZ;
    const float d = dcat0[(row * 2 + 0) * MZ + mz];
    dsh += d;
    dsc = fmaf(d, fmaf(cond[(int64_t)b * per + it], cm, ca), dsc);
  }
  __shared__ float red[4][2];
  dsc = wave_sum(dsc);
  dsh = wave_sum(dsh);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = dsc; red[threadIdx.x >> 6][1] = dsh; }
  __syncthreads();
  if (threadIdx.x < 2)  // [dscale, dshift] of this block; summed over the sample's blocks in order by launch_part_reduce
    part[((int64_t)b * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int launch_prep_inputs_bwd(const float* dcat0, const float* cond, float cm, float ca, float* dss, int ss_stride, int ss_off, int B,
                           int RT, int MZ, float* part, int64_t part_floats, hipStream_t s) {
  const int64_t per = (int64_t)RT * MZ;
  if (per == 0 || B == 0) return 0;
  dim3 grid(std::max(1, std::min(32, cdiv(per, 1024))), B);
  DQ_REQUIRE(part && part_floats >= (int64_t)grid.x * B * 2, "prep_inputs_bwd: partial-sum slot missing or too small");
  hipLaunchKernelGGL(k_prep_inputs_bwd, grid, dim3(256), 0, s, dcat0, cond, cm, ca, part, RT, MZ);
  DQ_LAUNCH_CHECK();
  PartReduce r;
  r.part = part; r.B = B; r.gx = (int)grid.x; r.nv = 2; r.s0 = 0; r.sn = 2; r.sdst = dss + ss_off; r.sstride = ss_stride;
  return launch_part_reduce(r, s);
}

}  // namespace dq
