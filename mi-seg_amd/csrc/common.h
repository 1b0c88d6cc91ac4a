// Shared device/host helpers for libmiseg_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/miseg_hip.h"

namespace miseg {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

int set_error(int code, const char* fmt, ...);

#define MISEG_REQUIRE(cond, code, ...)                      \
  do {                                                      \
    if (!(cond)) return miseg::set_error((code), __VA_ARGS__); \
  } while (0)

#define MISEG_LAUNCH_CHECK(name)                                                        \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess)                                                              \
      return miseg::set_error(MISEG_E_LAUNCH, "%s: %s", (name), hipGetErrorString(e__)); \
  } while (0)

// 16-byte vector of T (the unit every HBM / LDS access moves per lane)
template <class T> struct Vec16;
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16>  { typedef bf16x8 type; static constexpr int N = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

template <class F> static inline int dispatch_dtype(int dtype, F&& f) {
  if (dtype == MISEG_F32) return f((float*)nullptr);
  if (dtype == MISEG_BF16) return f((bf16*)nullptr);
  return set_error(MISEG_E_BADARG, "unknown dtype %d", dtype);
}

}  // namespace miseg
