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

// 64-lane wavefront reductions on the DPP data path (v_add_f32_dpp / v_max_f32_dpp: quad permutes, half-row and row mirrors, then the
// row broadcasts 15 and 31 and a v_readlane of lane 63): ~8 VALU instructions.  The __shfl_xor butterfly compiles to six DEPENDENT
// ds_bpermute_b32 round trips through the LDS crossbar -- the serial part of every row kernel (two reductions per LayerNorm row).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sc_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float sc_wave_sum(float v) {
  v += sc_dpp<0xB1, 0xF>(0.f, v);   // quad_perm [1,0,3,2]
  v += sc_dpp<0x4E, 0xF>(0.f, v);   // quad_perm [2,3,0,1]
  v += sc_dpp<0x141, 0xF>(0.f, v);  // row_half_mirror
  v += sc_dpp<0x140, 0xF>(0.f, v);  // row_mirror: every lane holds the sum of its row of 16
  v += sc_dpp<0x142, 0xA>(0.f, v);  // row_bcast15 into rows 1 and 3
  v += sc_dpp<0x143, 0xC>(0.f, v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float sc_wave_max(float v) {
  v = fmaxf(v, sc_dpp<0xB1, 0xF>(v, v));
  v = fmaxf(v, sc_dpp<0x4E, 0xF>(v, v));
  v = fmaxf(v, sc_dpp<0x141, 0xF>(v, v));
  v = fmaxf(v, sc_dpp<0x140, 0xF>(v, v));
  v = fmaxf(v, sc_dpp<0x142, 0xA>(v, v));
  v = fmaxf(v, sc_dpp<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// bf16 <-> f32 (round-to-nearest-even through the hardware convert; keeps NaN a NaN).
__device__ __forceinline__ float sc_bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short sc_f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// ---- stateless dropout generator -------------------------------------------------------------------------------------
// One 32-bit hash (lowbias32 finaliser) per PAIR of consecutive elements; element i uses bits [16*(i&1), +16) of the
// hash of pair i>>1 and is dropped when that 16-bit value is below round(p * 65536).  Cheap enough (2 x v_mul_lo_u32
// per two elements) to live in GEMM epilogues; every kernel that regenerates a mask uses these two helpers.
__device__ __forceinline__ unsigned sc_rand_pair(unsigned long long seed, unsigned long long pair) {
  unsigned x = (unsigned)pair * 0x9E3779B1u;
  x ^= (unsigned)(pair >> 32) * 0x85EBCA77u;
  x ^= (unsigned)seed ^ ((unsigned)(seed >> 32) * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned sc_drop_threshold(float p) { return (unsigned)(p * 65536.f + 0.5f); }
__device__ __forceinline__ bool sc_dropped(unsigned long long seed, unsigned long long idx, unsigned thr16) {
  const unsigned h = sc_rand_pair(seed, idx >> 1);
  return ((h >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr16;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on v_exp/v_rcp -- ~4x cheaper than erff in a GEMM epilogue
__device__ __forceinline__ float sc_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));  // (v_rcp_f32, 1 ulp: __frcp_rn expands to the 10-instruction IEEE division)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.f - poly * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// 3-term variant (A&S 7.1.25, |error| <= 2.5e-5): used where the result is rounded to bf16 (eps 3.9e-3) anyway
__device__ __forceinline__ float sc_erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f, ax, 1.f));
  float poly = fmaf(0.7478556f, t, -0.0958798f);
  poly = fmaf(poly, t, 0.3480242f);
  const float r = 1.f - poly * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// the same two, also handing back exp(-x^2) (for GELU': with x = u / sqrt 2 that is the Gaussian factor exp(-u^2 / 2) of phi(u))
__device__ __forceinline__ float sc_erf_g(float x, float& gauss) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  gauss = __expf(-ax * ax);
  return copysignf(1.f - poly * t * gauss, x);
}
__device__ __forceinline__ float sc_erf_fast_g(float x, float& gauss) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f, ax, 1.f));
  float poly = fmaf(0.7478556f, t, -0.0958798f);
  poly = fmaf(poly, t, 0.3480242f);
  gauss = __expf(-ax * ax);
  return copysignf(1.f - poly * t * gauss, x);
}
template <bool FAST>
__device__ __forceinline__ float sc_gelu_t(float x) {
  return 0.5f * x * (1.f + (FAST ? sc_erf_fast(x * 0.70710678118654752f) : sc_erf(x * 0.70710678118654752f)));
}
template <bool FAST>
__device__ __forceinline__ float sc_gelu_grad_t(float x) {
  float g;
  const float e = FAST ? sc_erf_fast_g(x * 0.70710678118654752f, g) : sc_erf_g(x * 0.70710678118654752f, g);
  return 0.5f * (1.f + e) + x * 0.39894228040143268f * g;
}
__device__ __forceinline__ float sc_gelu(float x) { return 0.5f * x * (1.f + sc_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float sc_gelu_grad(float x) {
  float g;
  const float e = sc_erf_g(x * 0.70710678118654752f, g);
  return 0.5f * (1.f + e) + x * 0.39894228040143268f * g;
}
