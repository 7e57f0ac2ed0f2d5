// Device helpers shared by the ResnetBlock backward kernels (k_res.hip, k_res_wg.hip).
#pragma once
#include "dq_common.h"

namespace dq {

// pointwise backward of RMSNorm -> (scale+1, shift) -> SiLU at one position (reference dquartic/model/unet1d.py:113-140, 262-266):
// returns dU in d[], accumulates dg / dsc / dsh
template <int C, bool SS>
__device__ __forceinline__ void norm_act_bwd(const float* u, float* d, const float* __restrict__ g, const float* __restrict__ ss,
                                             float* dg, float* dsc, float* dsh) {
  const float sqC = sqrtf((float)C);
  float ssq = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) ssq = fmaf(u[c], u[c], ssq);
  const float nrm = fast_sqrt(ssq), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
  float uh[C];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    uh[c] = u[c] * inv;
    const float z = uh[c] * g[c] * sqC;
    const float sc = SS ? ss[c] + 1.0f : 1.0f, sh = SS ? ss[C + c] : 0.f;
    const float w = fmaf(z, sc, sh);
    const float dw = d[c] * silu_grad_f(w);
    if (SS) { dsh[c] += dw; dsc[c] = fmaf(dw, z, dsc[c]); }
    const float dz = dw * sc;
    dg[c] = fmaf(dz, uh[c] * sqC, dg[c]);
    d[c] = dz * g[c] * sqC;
    dot = fmaf(d[c], uh[c], dot);
  }
  const bool clamped = nrm < RMS_EPS;
#pragma unroll
  for (int c = 0; c < C; ++c) d[c] = clamped ? d[c] * inv : inv * (d[c] - uh[c] * dot);
}

// Sum of the 16 four-lane blocks' 4 x 4 MFMA results: afterwards every lane (in particular lanes 0..3 = block 0, column j = lane) holds
// the total in every register.  Two DPP row rotations (the 4 blocks of a 16-lane row), then gfx950's v_permlane16_swap /
// v_permlane32_swap (tools/probe/blocks_sum.hip): all VALU.  As two __shfl_xor (ds_bpermute) per value -- a dependent LDS round trip
// each -- the flush of a weight-gradient kernel's ~50 values per wave took 10-17,000 clocks per workgroup, as much as a tile's work.
// The additions pair the same operands in the same order as that form did: the sums are bit-identical.
typedef float f32x4_sum __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_sum blocks_sum(f32x4_sum v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = v[i];
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);  // rows 0 <-> 1, 2 <-> 3
    t = __int_as_float(a[0]) + __int_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);  // halves
    v[i] = __int_as_float(b[0]) + __int_as_float(b[1]);
  }
  return v;
}

}  // namespace dq
