// Probe: lane mapping and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (prints D for a = lane id, b = 100 + lane id).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(100 + l), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void rate(float* out, int iters) {
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = threadIdx.x * 0.001f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.0f * iters);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  probe<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" %8.0f", h[l * 4 + r]); printf("\n"); }
  rate<<<1, 64>>>(d, 10000);
  hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
  printf("clock64 ticks per 4x4x1 MFMA (4 independent chains): %.2f\n", h[64]);
  return 0;
}
