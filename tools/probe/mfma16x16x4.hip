// Probe: lane maps of v_mfma_f32_16x16x4_f32 on gfx950 (the layout csrc/k_la_rows_bwd.hip builds on):
//   A (16 x 4): lane l supplies A[i = l % 16][k = l / 16];  B (4 x 16): lane l supplies B[k = l / 16][j = l % 16];
//   D (16 x 16): register r of lane l holds D[i = 4 (l / 16) + r][j = l % 16].
// Prints the number of mismatches against a host product and the dependent / independent issue cadence.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D, long long* cyc) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
  f32x4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
  const float a = A[l], b = B[l];
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  long long t2 = clock64();
  D[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  float hA[64], hB[64], hD[320], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 13) - 6.f; hB[i] = (float)((i * 5) % 11) - 5.f; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD; long long* dc;
  (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 1280); (void)hipMalloc(&dc, 16);
  (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD, dc);
  long long hc[2];
  (void)hipMemcpy(hD, dD, 1280, hipMemcpyDeviceToHost); (void)hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x4f32 lane map: %d mismatches of 256\n", bad);
  printf("64 MFMAs on one accumulator: %lld clocks (%.1f each); on four accumulators: %lld clocks (%.1f each)\n", hc[0], hc[0] / 64.0, hc[1], hc[1] / 64.0);
  return bad != 0;
}
