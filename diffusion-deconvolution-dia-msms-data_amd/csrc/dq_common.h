// Shared helpers for the gfx950 kernels of libdq_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace dq {

void set_error(const std::string& msg);

#define DQ_HIP_OK(expr)                                                                            \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      dq::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                    std::to_string(__LINE__) + ")");                                               \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

#define DQ_LAUNCH_CHECK()                                                                          \
  do {                                                                                             \
    hipError_t _e = hipGetLastError();                                                             \
    if (_e != hipSuccess) {                                                                        \
      dq::set_error(std::string("kernel launch failed: ") + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                    std::to_string(__LINE__) + ")");                                               \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

#define DQ_REQUIRE(cond, msg)                                                                      \
  do {                                                                                             \
    if (!(cond)) {                                                                                 \
      dq::set_error(std::string(msg) + " [" #cond "] (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
      return 2;                                                                                    \
    }                                                                                              \
  } while (0)

constexpr float RMS_EPS = 1e-12f;  // F.normalize eps (reference unet1d.py:140)

// v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the ~10-instruction correctly-rounded division / square root sequences: the
// pointwise ResnetBlock kernels are VALU-bound at sampling batch sizes, and 1 ulp is far inside the fp32 parity tolerance
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// sqrt(C) / max(||u||, eps) of RMSNorm / F.normalize
__device__ __forceinline__ float rms_inv(float ssq, float sqC) { return sqC * fast_rcp(fmaxf(fast_sqrt(ssq), RMS_EPS)); }

__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = fast_rcp(1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}


// Workgroup barrier that orders LDS accesses ONLY: this wave's LDS operations complete (lgkmcnt(0)), then s_barrier.  __syncthreads() is
// a full fence -- the compiler puts s_waitcnt vmcnt(0) in front of it, so every global load in flight (a tile requested ahead, the next
// unit's prefetch) is waited for at the barrier.  Use where only LDS data crosses waves.  (The "memory" clobber keeps the compiler from
// moving memory accesses across it.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// 64-lane wave reductions (all lanes receive the result)
// Sum over the 64 lanes, every lane receives it.  Inside a 16-lane row: four DPP adds (quad_perm swaps, then the two row
// mirrors) -- no LDS; across the four rows: v_readlane of one lane per row.  The __shfl_xor butterfly this replaces is six
// dependent ds_bpermute round trips per value and was the floor (~12 us) of every kernel that ends in a block reduction of
// dozens of values (the weight-gradient kernels: 52 per wave).
__device__ __forceinline__ float wave_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));  // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));  // row_mirror: every lane = its row's sum
  x = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Resident workgroups per CU of `fn` at (threads, dynamic LDS bytes) on the CURRENT device, cached per (function, LDS size, device): one
// instantiation is launched with different LDS footprints (down / up levels, the head), and a process may drive several devices.  Raises the
// function's dynamic-LDS limit when the launch needs more than 48 KB.  Returns < 0 after set_error().  (dq_unet.hip)
int occ_blocks_per_cu(const void* fn, int threads, size_t lds);

}  // namespace dq
