// Shared helpers for the speechain_amd HIP library (gfx950 / CDNA4 only).
//
// Conventions of every exported entry point (include/speechain_amd.h):
//   * plain pointers + sizes, a caller-owned hipStream_t, int status (0 = ok);
//   * never allocates, frees or synchronises (graph-capturable);
//   * on error the message is kept per thread and returned by sc_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define SC_OK 0
#define SC_ERR_ARG 1
#define SC_ERR_LAUNCH 2
#define SC_ERR_WORKSPACE 3

#define SC_API extern "C" __attribute__((visibility("default")))

int sc_set_error(int code, const char* fmt, ...);

#define SC_CHECK_ARG(cond, ...)                                \
  do {                                                         \
    if (!(cond)) return sc_set_error(SC_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define SC_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return sc_set_error(SC_ERR_LAUNCH, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                          __FILE__, __LINE__);                                                    \
  } while (0)

#define SC_LAUNCH_CHECK(name)                                                                       \
  do {                                                                                              \
    hipError_t e_ = hipGetLastError();                                                              \
    if (e_ != hipSuccess)                                                                           \
      return sc_set_error(SC_ERR_LAUNCH, "launch of %s failed: %s", name, hipGetErrorString(e_));   \
  } while (0)

static inline int sc_cdiv(int a, int b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// 64-lane wavefront reductions (butterfly over DPP/ds_swizzle as the compiler chooses).
__device__ __forceinline__ float sc_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sc_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// bf16 <-> f32 (round-to-nearest-even through the hardware convert; keeps NaN a NaN).
__device__ __forceinline__ float sc_bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short sc_f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
