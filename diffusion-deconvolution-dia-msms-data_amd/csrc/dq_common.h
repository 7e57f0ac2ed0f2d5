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

// 64-lane wave reductions (all lanes receive the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace dq
