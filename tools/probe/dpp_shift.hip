// Probe: cycles per lane-shift-by-one of a 64-lane register: DPP wave_shr:1 / wave_shl:1, DPP row_shr:1 (+ row_bcast:15 fix-up),
// ds_bpermute, and a plain v_mov for scale (s_memtime ticks, one wave, dependent chain and 8 independent values).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__device__ __forceinline__ float sh(float v, int addr) {
  int x = __float_as_int(v);
  if (MODE == 0) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false));       // wave_shr:1
  if (MODE == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x130, 0xF, 0xF, false));       // wave_shl:1
  if (MODE == 2) {                                                                                       // row_bcast:15 then row_shr:1
    int t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xF, 0xF, false);
    return __int_as_float(__builtin_amdgcn_update_dpp(t, x, 0x111, 0xF, 0xF, false));
  }
  if (MODE == 3) return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, x));                           // ds_bpermute
  if (MODE == 4) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false));       // row_shr:1 only
  return v * 1.0001f;
}
template <int MODE>
__global__ void rate(float* out, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  const int addr = ((threadIdx.x + 63) & 63) * 4;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sh<MODE>(v[i], addr) + 1.0f;
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (8.0f * iters);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  float h[65];
#define RUN(M, name) rate<M><<<1, 64>>>(d, 4000); (void)hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("%-34s %.2f ticks per (shift + add)   lane1=%g lane17=%g\n", name, h[64], h[1], h[17]);
  RUN(5, "v_mul only")
  RUN(0, "dpp wave_shr:1") RUN(1, "dpp wave_shl:1") RUN(2, "dpp row_bcast:15 + row_shr:1") RUN(4, "dpp row_shr:1") RUN(3, "ds_bpermute")
  return 0;
}
