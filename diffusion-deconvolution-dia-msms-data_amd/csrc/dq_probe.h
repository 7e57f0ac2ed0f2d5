// In-kernel interval probe (timing experiments only; not part of the product build).  A translation unit compiled with -DDQ_KPROBE
// (tools/build_variant.sh probe <file> -DDQ_KPROBE) records shader-clock stamps of the first wave of every workgroup of ONE selected
// kernel instantiation: DQ_PSTAMP(id, i) in the kernel, dq_kprobe_select(id) / dq_kprobe_read(buf) from the host (tools/probe_step.py).
#pragma once
#ifdef DQ_KPROBE
__device__ unsigned long long dq_kprobe_buf[4096 * 16];
__device__ int dq_kprobe_want;
#define DQ_PSTAMP(id, i)                                                                                        \
  do {                                                                                                          \
    if (threadIdx.x == 0 && dq_kprobe_want == (id))                                                             \
      dq_kprobe_buf[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 16 + (i)] = clock64();                      \
  } while (0)
extern "C" int dq_kprobe_select(int id) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(dq_kprobe_want), &id, sizeof(int)); }
extern "C" int dq_kprobe_read(unsigned long long* out) {
  const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dq_kprobe_buf), sizeof(unsigned long long) * 4096 * 16);
  return rc;
}
extern "C" int dq_kprobe_clear(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(dq_kprobe_buf)) != hipSuccess) return 1;
  return (int)hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 16);
}
#else
#define DQ_PSTAMP(id, i) do {} while (0)
#endif
