// Probe: cycles per v_mfma_f32_4x4x1_16b_f32 with 1, 2, 4, 8 independent accumulator chains (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void rate(float* out, int iters) {
  f4 c[NCH];
  for (int i = 0; i < NCH; ++i) c[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < NCH; ++k) c[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NCH; ++i) s += c[i][0] + c[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (8.0f * NCH * iters);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  float h[65];
#define RUN(N) rate<N><<<1, 64>>>(d, 2000); (void)hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("chains %d: %.2f ticks per MFMA\n", N, h[64]);
  RUN(1) RUN(2) RUN(4) RUN(8)
  return 0;
}
