// Shared device/host helpers for the gfx950 kernels of the V2A flow-matching sampler.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/v2a_cfm.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define V2A_WAVE 64

// ---- error reporting (host) -------------------------------------------------------------
extern thread_local char v2a_err_buf[512];
static inline int v2a_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(v2a_err_buf, sizeof(v2a_err_buf), fmt, ap);
  va_end(ap);
  return code;
}
#define V2A_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return v2a_fail(V2A_ERR_ARG, __VA_ARGS__); \
  } while (0)
static inline int v2a_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return V2A_OK;
}

// More than 64 KB of dynamic LDS is opt-in per kernel AND per device: `done` (one static per call site = per kernel
// instantiation) holds one bit per device ordinal, so a process that launches on a second GPU sets the attribute there too, and a
// failing attribute call is reported as itself instead of as an opaque launch error later.
static inline int v2a_enable_lds(const void* kern, size_t bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return V2A_OK;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "%s: hipFuncSetAttribute(%zu bytes of LDS) on device %d: %s", what, bytes, dev, hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_release);
  return V2A_OK;
}

// ---- device helpers ---------------------------------------------------------------------
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
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_exp + one v_rcp instead of libm erff's
// branchy polynomial; used where the result is rounded to bf16 anyway.
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp); __frcp_rn is a ten-instruction IEEE division
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfz = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfz, x));
}

// Sum over the 8 lanes of an aligned lane octet, left in every lane of it: three DPP adds (swap neighbours, swap pairs, mirror the
// half row), no LDS traffic (a __shfl_xor is a ds_bpermute).
__device__ __forceinline__ float octet_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// step vector address: base + step[0]*step_stride + batch*batch_stride
__device__ __forceinline__ const float* step_vec(const float* base, const int32_t* step, int64_t step_stride,
                                                 int64_t batch_stride, int64_t batch) {
  int64_t s = step ? (int64_t)step[0] : 0;
  return base + s * step_stride + batch * batch_stride;
}

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = V2A_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = V2A_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
