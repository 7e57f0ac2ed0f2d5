// Probe: semantics of v_permlane32_swap_b32 on gfx950 (builtin __builtin_amdgcn_permlane32_swap(old, src, fi, bc)).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2 + 0] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  int* d; (void)hipMalloc(&d, 1024);
  k<<<1, 64>>>(d);
  int h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %d  r1 = %d   (a = %d, b = %d)\n", l, h[2 * l], h[2 * l + 1], 100 + l, 200 + l);
  return 0;
}
