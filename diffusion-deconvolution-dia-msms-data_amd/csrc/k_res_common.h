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

}  // namespace dq
