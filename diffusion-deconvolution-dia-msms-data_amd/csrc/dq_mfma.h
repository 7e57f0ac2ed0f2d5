// Wave-level MFMA helpers shared by the LinearAttention kernels (v_mfma_f32_32x32x2_f32, exact fp32).
//   A operand: lane l supplies A[i = l & 31][k = l >> 5] ; B operand: lane l supplies B[k = l >> 5][j = l & 31]
//   C/D: register r of lane l holds D[row = rmap(r, l >> 5)][col = l & 31]
// An accumulator X (rows in registers, column on the lane) feeds, register by register, a product that sums over X's
// ROW index:  sum_r mfma(X.r, Y.r) = X^T Y  with rows = X's columns, cols = Y's columns.
#pragma once
#include <hip/hip_runtime.h>

namespace dq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ constexpr int rmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// Which channel x-register j of a lane of half `half` holds in the LinearAttention kernels, and how many such registers a lane
// has.  C = 8 / 16: the accumulator row map (8 / 16 channel slots, all used).  C = 4: c = 2 * half + j over TWO registers, C = 12:
// c = 6 * half + j over SIX -- with the row map the lanes of half 1 would hold channels 4..7 (C = 4) resp. four of the sixteen slots
// would be empty (C = 12), i.e. part of every K = 2 slice of a 32x32x2 product would be zero padding: four MFMAs per K = 4
// projection instead of two, eight per K = 12 projection instead of six.
__host__ __device__ __forceinline__ constexpr int la_nj(int C) { return C == 4 ? 2 : (C == 12 ? 6 : (C <= 8 ? 4 : 8)); }
__device__ __forceinline__ constexpr int la_chan(int C, int j, int half) {
  return C == 4 ? 2 * half + j : (C == 12 ? 6 * half + j : rmap(j, half));
}
// k_la_small: the channel that half `half` supplies in K-step i of a projection (C / 2 steps).  C = 8 / 16: the accumulator row map, so
// that a projection operand register IS the residual / output register of the same index; C = 12: i + 6 half (six full steps instead of
// eight with a third of the slots empty; the residual is then read a second time in the row-map layout).
__host__ __device__ __forceinline__ constexpr int sm_chan(int C, int i, int half) { return C == 12 ? i + 6 * half : (i & 3) + 8 * (i >> 2) + 4 * half; }
// value of lane ^ 32 (ds_bpermute).  gfx950's v_permlane32_swap_b32 (tools/probe/permlane32.hip) was tried here: with the
// two register copies and the select it needs it is four VALU instructions, and the VALU-bound forward kernel got 4 % slower
// (the LDS pipe that serves ds_bpermute is otherwise idle there); the latency-bound backward did not change.
__device__ __forceinline__ float swap_half(float v) { return __shfl_xor(v, 32, 64); }

// X^T Y over all 16 registers
__device__ __forceinline__ f32x16 xty(const f32x16& x, const f32x16& y, f32x16 acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc = mfma_f32(x[r], y[r], acc);
  return acc;
}

// 32x32 transpose of an accumulator tile through a wave-private LDS tile [32][33] (conflict-free both ways)
__device__ __forceinline__ f32x16 transpose_tile(f32x16 a, float* tile, int col, int half) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[rmap(r, half) * 33 + col] = a[r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = tile[col * 33 + rmap(r, half)];
  return o;
}

// keep element (row in registers, column on the lane) only where row / N == col / N  (pairs inside one m/z row)
template <int N>
__device__ __forceinline__ f32x16 mask_same_row(f32x16 a, int col, int half) {
  const int cr = col / N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = half ? rmap(r, 1) / N : rmap(r, 0) / N;
    a[r] = rr == cr ? a[r] : 0.f;
  }
  return a;
}

}  // namespace dq
