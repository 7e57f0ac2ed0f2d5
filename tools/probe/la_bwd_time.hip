// Where does the time of one LinearAttention-backward launch go?  Builds the product kernel source with -DDQ_LA_PROBE (per-wave
// shader-clock stamps: 0 start, 1 after the weight staging, 2 after the unit loop, 10 / 11 inside the flush, 3 after it) and
// launches single instantiations on random data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DDQ_LA_PROBE \
//         -I include -I diffusion-deconvolution-dia-msms-data_amd/csrc tools/probe/la_bwd_time.hip -o gpurun_out/la_bwd_time
#include "k_la_bwd.hip"
#include <algorithm>
#include <vector>

namespace dq {
void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
int launch_axpy(float*, const float*, int64_t, hipStream_t) { return 1; }
int launch_block_bwd(const BlockBwd&, hipStream_t) { return 1; }
int launch_zero(float*, int64_t, hipStream_t) { return 1; }
int launch_linattn_bwd_long(const float*, const float*, float*, const float*, const float*, const float*, float*, int, int, int, int*,
                            hipStream_t) { return 1; }
void launch_linattn_bwd_big(const LinAttnBwdK&, int, int, hipStream_t) {}
}  // namespace dq

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f); }
  float* d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

template <int C, int N>
static void run(int rows) {
  using namespace dq;
  constexpr int RW = N >= 32 ? 1 : 32 / N;
  const size_t el = (size_t)rows * C * N;
  LinAttnBwdK k;
  k.x = dev_rand(el, 1.f, 1); k.ypre = dev_rand(el, 1.f, 2); k.dy = dev_rand(el, 1.f, 3);
  k.dx = dev_rand(el, 1.f, 4);
  k.w_qkv = dev_rand(384 * C, 0.3f, 6); k.w_out = dev_rand(128 * C, 0.3f, 7); k.g_pre = dev_rand(C, 1.f, 8); k.g_out = dev_rand(C, 1.f, 9);
  // the launcher's grid (linattn_bwd_n): a block of four waves (= heads) per unit range, one resident round; N = 1: a wave per range
  const int units = cdiv(rows, RW);
  int occ0 = 1;
  if constexpr (N != 1) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, k_linattn_bwd<C, N>, 256, 0));
  const int max_slots = N == 1 ? 1024 : 256 * occ0;
  k.units_per_wave = std::max(1, cdiv(units, max_slots));
  const int slots = cdiv(units, k.units_per_wave);
  const int blocks = N == 1 ? cdiv(slots, 4) : slots, waves = 4 * blocks;
  float* part; CK(hipMalloc(&part, (size_t)slots * la_slot(C) * sizeof(float)));
  k.part = part; k.rows = rows; k.prep = nullptr; k.dx_store = 0;
  unsigned long long* probe; CK(hipMalloc(&probe, (size_t)(waves + 4) * 16 * 8));
  CK(hipMemset(probe, 0, (size_t)(waves + 4) * 16 * 8));
  k.probe = nullptr;
  auto launch = [&] {
    if constexpr (N == 1) hipLaunchKernelGGL((k_linattn_bwd1<C>), dim3(blocks), dim3(256), 0, 0, k);
    else hipLaunchKernelGGL((k_linattn_bwd<C, N>), dim3(blocks), dim3(256), 0, 0, k);
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  k.probe = probe;
  launch();
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)waves * 16);
  CK(hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < waves; ++w) if (h[(size_t)w * 16]) t0 = std::min(t0, h[(size_t)w * 16]);
  int occ = 0;
  hipFuncAttributes fa;
  if constexpr (N == 1) {
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_linattn_bwd1<C>, 256, 0));
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_linattn_bwd1<C>)));
  } else {
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_linattn_bwd<C, N>, 256, 0));
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_linattn_bwd<C, N>)));
  }
  printf("C=%d N=%d rows=%d blocks=%d units/block=%d : %.1f us/launch  [occupancy %d blocks/CU, regs %d, lds %zu, scratch %zu]\n", C, N, rows, blocks,
         k.units_per_wave, 1e3 * ms / reps, occ, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  // stamps are shader-clock ticks (clock64 = s_memtime); raw ticks, min / median / max over the waves
  const char* names[12] = {"start", "staged", "loop", "flush", "", "", "", "", "", "", "qk stored", "w2 summed"};
  for (int i : {0, 1, 2, 10, 11, 3}) {
    std::vector<unsigned long long> v;
    for (int w = 0; w < waves; ++w) if (h[(size_t)w * 16 + i]) v.push_back(h[(size_t)w * 16 + i] - t0);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("   %-9s min %8llu  med %8llu  max %8llu ticks\n", names[i], v.front(), v[v.size() / 2], v.back());
  }
  fflush(stdout);
}

int main() {
  const int rows = 12800;  // batch 32 x 400 RT: the bench's train step
  // the eleven instantiations of the bench's train step, each on the launcher's grid
  run<4, 64>(rows); run<4, 32>(rows);
  run<8, 32>(rows); run<8, 16>(rows); run<8, 8>(rows);
  run<12, 8>(rows); run<12, 4>(rows); run<12, 2>(rows);
  run<16, 2>(rows); run<16, 1>(rows);
  return 0;
}
