// Probe: v_mfma_f32_4x4x1 issue rate in the shape the conv kernels use it: 12 accumulator chains, A operands from a register ring,
// B operands produced by DPP shifts + selects, 1 or 4 waves per workgroup, 1..N workgroups.  Reports s_memtime ticks per MFMA and the
// wall time per MFMA (HIP events) so that ticks can be converted to ns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lane_m1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_p1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true)); }
template <int MODE>  // 0: constant operands; 1: varying A / B registers; 2: + DPP shifts and selects per 12 MFMAs
__global__ void rate(float* out, const float* in, int iters) {
  f4 c[4][3];
  for (int g = 0; g < 4; ++g) for (int k = 0; k < 3; ++k) c[g][k] = f4{0, 0, 0, 0};
  float x[16], w[16];
  for (int i = 0; i < 16; ++i) { x[i] = in[threadIdx.x + 64 * i]; w[i] = in[threadIdx.x * 3 + i]; }
  const bool hasL = (threadIdx.x & 7) != 0, hasR = (threadIdx.x & 7) != 7;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      float xm, xc, xp;
      if (MODE == 2) { const float tm = lane_m1(x[ch]), tp = lane_p1(x[ch]); xm = hasL ? tm : 0.f; xp = hasR ? tp : 0.f; xc = x[ch]; }
      else if (MODE == 1) { xm = x[(ch + 1) & 15]; xc = x[ch]; xp = x[(ch + 2) & 15]; }
      else { xm = xc = xp = x[0]; }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float a0 = MODE ? w[(ch + g) & 15] : w[0], a1 = MODE ? w[(ch + g + 5) & 15] : w[0], a2 = MODE ? w[(ch + g + 9) & 15] : w[0];
        c[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, xm, c[g][0], 0, 0, 0);
        c[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, xc, c[g][1], 0, 0, 0);
        c[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, xp, c[g][2], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int g = 0; g < 4; ++g) for (int k = 0; k < 3; ++k) s += c[g][k][0] + c[g][k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (192.0f * iters);
}
int main() {
  float *d, *in;
  (void)hipMalloc(&d, (1 << 20) * 4 + 64);
  (void)hipMalloc(&in, 1 << 16);
  (void)hipMemset(in, 0, 1 << 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400;
  auto run = [&](int mode, int blocks, int threads) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) rate<0><<<blocks, threads>>>(d, in, iters);
      if (mode == 1) rate<1><<<blocks, threads>>>(d, in, iters);
      if (mode == 2) rate<2><<<blocks, threads>>>(d, in, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
    }
    float ms = 0, ticks = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&ticks, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    printf("mode %d blocks %4d threads %3d: %.2f ticks per MFMA, %.2f ns per MFMA per wave (kernel %.1f us)\n", mode, blocks, threads, ticks,
           ms * 1e6 / (192.0 * iters), ms * 1e3);
  };
  for (int mode = 0; mode < 3; ++mode) {
    run(mode, 1, 64); run(mode, 1, 256); run(mode, 128, 256); run(mode, 256, 256); run(mode, 512, 256); run(mode, 1024, 256);
  }
  return 0;
}
