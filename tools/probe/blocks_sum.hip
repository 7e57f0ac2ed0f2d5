// Probe: sum over the 16 four-lane blocks of a wave without LDS round trips -- two DPP row rotations, then v_permlane16_swap /
// v_permlane32_swap (gfx950) -- against the __shfl_xor form; prints the permlane16 semantics and the cycles of both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float ror(float v, int c) {
  return c == 4 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false))
                : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum_shfl(float t) {
  t += ror(t, 4); t += ror(t, 8);
  t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
  return t;
}
__device__ __forceinline__ float sum_perm(float t) {
  t += ror(t, 4); t += ror(t, 8);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = __int_as_float(a[0]) + __int_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}
__global__ void k(float* out, int* sem, long long* cyc) {
  const int l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(100 + l, 200 + l, false, false);
  sem[2 * l] = r[0]; sem[2 * l + 1] = r[1];
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = (float)((l * 7 + i * 13) % 31) - 15.f;
  float a[32], b[32];
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = sum_shfl(v[i]);
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 32; ++i) b[i] = sum_perm(v[i]);
  long long t2 = clock64();
  float d = 0;
  for (int i = 0; i < 32; ++i) d = fmaxf(d, fabsf(a[i] - b[i]));
  out[l] = d; out[64 + l] = a[0]; out[128 + l] = b[0];
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  float* d; int* s; long long* c;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&s, 1024); (void)hipMalloc(&c, 64);
  k<<<1, 64>>>(d, s, c);
  float h[192]; int hs[128]; long long hc[2];
  (void)hipMemcpy(h, d, 768, hipMemcpyDeviceToHost); (void)hipMemcpy(hs, s, 512, hipMemcpyDeviceToHost); (void)hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  for (int l : {0, 15, 16, 31, 32, 47, 48, 63}) printf("lane %2d: permlane16_swap r0 = %d r1 = %d\n", l, hs[2 * l], hs[2 * l + 1]);
  float mx = 0; for (int l = 0; l < 64; ++l) mx = fmaxf(mx, h[l]);
  printf("max |shfl - perm| over lanes = %g (lane 0: %g vs %g; lane 5: %g vs %g)\n", mx, h[64], h[128], h[69], h[133]);
  printf("32 sums: shfl form %lld clocks, permlane form %lld clocks\n", hc[0], hc[1]);
  return 0;
}
