// Training-batch formation on the GPU from a dataset that is resident in HBM: the step right before
// the hot path.  Replaces, for a batch of B (window 1, window 2) index pairs,
//   DIAMSDataset.__getitem__'s per-pair min-max normalisation   dquartic/utils/data_loader.py:70-79
//   the mixture  ms2_cond = w1*ms2_1 + w2*ms2_2                 dquartic/model/model_interface.py:1073-1075
// with the same fp32 arithmetic in the same order ((x - min) / (max - min), then a*w1 + b*w2, no FMA contraction), so the
// result is bit-identical to the reference's numpy/torch path on float32 data -- including NaN for a constant pair (0/0).
// MS2 min/max run over BOTH windows of the pair, MS1 min/max over window 1 only and are applied to both (:72-79).
//
// Two launches: (1) k_pair_minmax: grid (PAIR_CHUNKS, B), each block reduces its chunk of both windows to a partial
// (min, max); block 0 of a pair also reduces MS1 of window 1.  (2) k_pair_mix: float4 grid-stride over the batch; every
// thread folds its pair's PAIR_CHUNKS partials in fixed order, normalises and writes the outputs.
// HBM-bound; algorithmic bytes per pair: read 2 windows + write ms2_1, ms2_2, ms2_cond = 5 * RT*MZ*4 B (+ MS1 rows); the second
// read of the two windows in (2) comes from L2 / Infinity Cache.
#include "dq_common.h"
#include "dq_kernels.h"
#include "../../include/dq_hip.h"
#include <algorithm>

namespace dq {

constexpr int PAIR_CHUNKS = 8;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// idx: [idx1(B) | idx2(B)] on the device; a pair with an index outside [0, n_windows) is never dereferenced: all of its
// outputs become NaN (loud in the loss) instead of a wild read.  part: B * (PAIR_CHUNKS + 1) * 2 floats ((min,max) per chunk, last slot = MS1).
__global__ void __launch_bounds__(256) k_pair_minmax(const float* __restrict__ ms2, const float* __restrict__ ms1,
                                                     const int64_t* __restrict__ idx, int64_t n_windows, int B, int64_t per4,
                                                     int64_t ms1_per, float* __restrict__ part) {
  const int b = blockIdx.y, c = blockIdx.x;
  const int64_t i1 = idx[b], i2 = idx[B + b];
  if (i1 < 0 || i1 >= n_windows || i2 < 0 || i2 >= n_windows) {  // never read out of bounds: poison the pair instead
    if (threadIdx.x == 0) {
      float* pp = part + ((int64_t)b * (PAIR_CHUNKS + 1) + c) * 2;
      pp[0] = pp[1] = NAN;
      if (c == 0) pp[2 * PAIR_CHUNKS] = pp[2 * PAIR_CHUNKS + 1] = NAN;
    }
    return;
  }
  const float4* w1 = reinterpret_cast<const float4*>(ms2) + i1 * per4;
  const float4* w2 = reinterpret_cast<const float4*>(ms2) + i2 * per4;
  const int64_t lo_i = per4 * c / PAIR_CHUNKS, hi_i = per4 * (c + 1) / PAIR_CHUNKS;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = lo_i + threadIdx.x; i < hi_i; i += blockDim.x) {
    const float4 a = w1[i], d = w2[i];
    mn = fminf(fminf(fminf(a.x, a.y), fminf(a.z, a.w)), fminf(fminf(fminf(d.x, d.y), fminf(d.z, d.w)), mn));
    mx = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)), mx));
  }
  __shared__ float red[2][4];
  mn = wave_min(mn); mx = wave_max_f(mx);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mn; red[1][threadIdx.x >> 6] = mx; }
  __syncthreads();
  float* pp = part + ((int64_t)b * (PAIR_CHUNKS + 1) + c) * 2;
  if (threadIdx.x == 0) {
    pp[0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    pp[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
  if (c != 0) return;
  // MS1 of window 1 only (data_loader.py:74-75)
  __syncthreads();
  const float* m = ms1 + i1 * ms1_per;
  mn = INFINITY; mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < ms1_per; i += blockDim.x) { mn = fminf(mn, m[i]); mx = fmaxf(mx, m[i]); }
  mn = wave_min(mn); mx = wave_max_f(mx);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mn; red[1][threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p1 = part + ((int64_t)b * (PAIR_CHUNKS + 1) + PAIR_CHUNKS) * 2;
    p1[0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    p1[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

__device__ __forceinline__ float4 norm4(float4 v, float lo, float rng) {
  return make_float4((v.x - lo) / rng, (v.y - lo) / rng, (v.z - lo) / rng, (v.w - lo) / rng);
}

__global__ void __launch_bounds__(256) k_pair_mix(const float* __restrict__ ms2, const float* __restrict__ ms1,
                                                  const int64_t* __restrict__ idx, int64_t n_windows, int B, int64_t per4,
                                                  int64_t ms1_per, const float* __restrict__ part, float w1, float w2, float* __restrict__ o_ms2_1,
                                                  float* __restrict__ o_ms1_1, float* __restrict__ o_ms2_2,
                                                  float* __restrict__ o_ms1_2, float* __restrict__ o_cond) {
  const int64_t total = (int64_t)B * per4;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gid; i < total; i += gstride) {
    const int b = (int)(i / per4);
    const int64_t e = i - (int64_t)b * per4;
    const float* pp = part + (int64_t)b * (PAIR_CHUNKS + 1) * 2;
    float lo = pp[0], hi = pp[1];
#pragma unroll
    for (int c = 1; c < PAIR_CHUNKS; ++c) { lo = fminf(lo, pp[2 * c]); hi = fmaxf(hi, pp[2 * c + 1]); }
    const float rng = hi - lo;
    const int64_t i1 = idx[b], i2 = idx[B + b];
    const bool ok = i1 >= 0 && i1 < n_windows && i2 >= 0 && i2 < n_windows;
    const float4 nan4 = make_float4(NAN, NAN, NAN, NAN);
    const float4 a = ok ? norm4(reinterpret_cast<const float4*>(ms2)[i1 * per4 + e], lo, rng) : nan4;
    const float4 d = ok ? norm4(reinterpret_cast<const float4*>(ms2)[i2 * per4 + e], lo, rng) : nan4;
    reinterpret_cast<float4*>(o_ms2_1)[i] = a;
    if (o_ms2_2) reinterpret_cast<float4*>(o_ms2_2)[i] = d;
    if (o_cond)  // (ms2_1 * w1) + (ms2_2 * w2), two roundings per product like the reference's tensor ops
      reinterpret_cast<float4*>(o_cond)[i] =
          make_float4(a.x * w1 + d.x * w2, a.y * w1 + d.y * w2, a.z * w1 + d.z * w2, a.w * w1 + d.w * w2);
  }
  const int64_t total1 = (int64_t)B * ms1_per;
  for (int64_t i = gid; i < total1; i += gstride) {
    const int b = (int)(i / ms1_per);
    const int64_t e = i - (int64_t)b * ms1_per;
    const float* p1 = part + ((int64_t)b * (PAIR_CHUNKS + 1) + PAIR_CHUNKS) * 2;
    const float lo = p1[0], rng = p1[1] - p1[0];
    const int64_t i1 = idx[b], i2 = idx[B + b];
    const bool ok = i1 >= 0 && i1 < n_windows && i2 >= 0 && i2 < n_windows;
    o_ms1_1[i] = ok ? (ms1[i1 * ms1_per + e] - lo) / rng : NAN;
    if (o_ms1_2) o_ms1_2[i] = ok ? (ms1[i2 * ms1_per + e] - lo) / rng : NAN;
  }
}

}  // namespace dq

extern "C" {

int64_t dq_pair_batch_scratch_bytes(int B) {
  return B <= 0 ? 0 : (int64_t)B * (dq::PAIR_CHUNKS + 1) * 2 * (int64_t)sizeof(float);
}

int dq_pair_batch(const float* ms2_data, const float* ms1_data, int64_t n_windows, const int64_t* idx_dev, int B, int RT, int MZ,
                  int64_t ms1_per_window, float w1, float w2, float* ms2_1, float* ms1_1, float* ms2_2, float* ms1_2,
                  float* ms2_cond, void* scratch, int64_t scratch_bytes, void* stream) {
  using namespace dq;
  DQ_REQUIRE(ms2_data && ms1_data && idx_dev && ms2_1 && ms1_1 && scratch, "dq_pair_batch: null argument");
  DQ_REQUIRE(B > 0 && RT > 0 && MZ > 0 && ms1_per_window > 0 && n_windows > 0, "dq_pair_batch: sizes must be positive");
  const int64_t per = (int64_t)RT * MZ;
  DQ_REQUIRE(per % 4 == 0, "dq_pair_batch: RT*MZ must be a multiple of 4");
  DQ_REQUIRE(B <= 65535, "dq_pair_batch: at most 65535 pairs per call");
  DQ_REQUIRE(scratch_bytes >= dq_pair_batch_scratch_bytes(B), "dq_pair_batch: scratch too small (dq_pair_batch_scratch_bytes)");
  hipStream_t s = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(scratch);
  hipLaunchKernelGGL(k_pair_minmax, dim3(PAIR_CHUNKS, B), dim3(256), 0, s, ms2_data, ms1_data, idx_dev, n_windows, B, per / 4,
                     ms1_per_window, part);
  DQ_LAUNCH_CHECK();
  const int grid = (int)std::min<int64_t>(cdiv((int64_t)B * per / 4, 256), 4096);
  hipLaunchKernelGGL(k_pair_mix, dim3(grid), dim3(256), 0, s, ms2_data, ms1_data, idx_dev, n_windows, B, per / 4, ms1_per_window,
                     part, w1, w2, ms2_1, ms1_1, ms2_2, ms1_2, ms2_cond);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
