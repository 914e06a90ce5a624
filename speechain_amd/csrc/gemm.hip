// Batched matrix product with fused epilogue on the gfx950 matrix cores -- the workhorse behind every
// nn.Linear / torch.matmul call site of the reference's transformer path (paths under /root/reference/):
//   speechain/module/transformer/attention.py:72-79,99-106,131   (Q/K/V/out projections, QK^T, PV)
//   speechain/module/transformer/feed_forward.py:93-101          (FFN in/out layers + activation + dropout)
//   speechain/module/postnet/token.py:49                         (vocabulary heads)
// and their autograd backward (dgrad / wgrad) products.
//
//   C[z][m][n] = epilogue( alpha * sum_k opA[z][m][k] * opB[z][n][k] )
//
// opA is M x K, opB is N x K ("NT" form: nn.Linear's y = x W^T with W stored (out, in)).  Either operand may be
// stored with its ROW index contiguous instead of k (ta / tb = 1), which covers dgrad-free wgrad (dW = dy^T x),
// P V and the attention backward products without materialising transposes in HBM.
//
// Two arithmetic modes, same tiling (128 x 128 output tile per 256-thread workgroup, 4 waves as 2 x 2, each wave
// 64 x 64 = 2 x 2 MFMA 32x32 accumulators):
//   bf16 in / f32 accumulate : v_mfma_f32_32x32x16_bf16, BK = 32   (training / bench path)
//   f32  in / f32 accumulate : v_mfma_f32_32x32x2_f32,   BK = 16   (exact-f32 parity path; a k-ordered fmaf chain)
// Global -> registers -> LDS staging with double-buffered LDS and one barrier per K tile; k-contiguous operands
// are staged with 16-byte loads/stores into rows padded to 80 B (bf16) / 68 B (f32) so the MFMA fragment reads
// are bank-conflict free; row-contiguous operands are transposed on their way into LDS.
#include "sc_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { AUX_NONE = 0, AUX_ADD = 1, AUX_ACT_GRAD = 2, AUX_MUL = 3 };
constexpr int ACT_PRE_GRAD = 0x100;  // flag bit of the `act` argument: pre_out receives act'(pre) * dropout factor instead of pre

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  long long lda, ldb, ldc;
  int nb2;
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  float alpha;
  const float* bias;
  int act;
  int pre_grad;        // pre_out <- act'(v) * (dropped ? 0 : keep) (what a later data-gradient product multiplies by: AUX_MUL) instead of v
  void* pre_out;
  const void* aux;
  long long ldaux;
  int aux_mode;
  float drop_p, drop_keep;
  const unsigned long long* seed;  // device cell holding the generator's current seed (graph replays draw fresh masks)
  unsigned long long offset;
  int out_f32;
  int accumulate;
  int a_vec, b_vec;  // 16-byte vector loads allowed for A / B
  int c_vec, c_vec32;  // vector accesses allowed for C/aux/pre_out in the compute dtype / for an f32 C
  int splitk;          // > 1: blockIdx.z is a K slice of ksplit_len, raw f32 partial tiles go to slab z of C
  int ksplit_len;
  float* colsum_part;  // split-K with a transposed A only: per-slice column sums of A over the contraction, (slices, M) f32, or null
  int exp_flags;       // experiment switches (sc_gemm_experiment; 0 in production): 4 = static priority for the younger waves
};

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const void* p, long long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct Elem<unsigned short> {
  static __device__ __forceinline__ float load(const void* p, long long i) { return sc_bf16_to_f32(((const unsigned short*)p)[i]); }
  static __device__ __forceinline__ void store(void* p, long long i, float v) { ((unsigned short*)p)[i] = sc_f32_to_bf16(v); }
};

constexpr int STG_LD = 68;  // f32 row stride of a wave's 32 x 64 staging tile (272 B: 16-byte aligned, bank-skewed)

__device__ __forceinline__ float epilogue_elem(const GemmArgs& p, float v, float bias, float aux, bool dropped, float keep) {
  v = v * p.alpha + bias;
  if (p.aux_mode == AUX_ACT_GRAD) {
    if (p.act == ACT_RELU) v = aux > 0.f ? v : 0.f;
    else if (p.act == ACT_GELU) v *= sc_gelu_grad(aux);
  } else {
    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == ACT_GELU) v = sc_gelu(v);
  }
  v = dropped ? 0.f : v * keep;
  if (p.aux_mode == AUX_ADD) v += aux;
  if (p.aux_mode == AUX_MUL) v *= aux;
  return v;
}

// Epilogue of one wave's 32 x 64 sub-tile held in its private LDS staging area: every lane owns runs of 8
// consecutive columns, so bias/aux loads and the C store are 16-byte (bf16) / 32-byte (f32) vector accesses.
template <typename T>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, const float* stage, long long zoffC, int z, int mbase,
                                              int nbase, int lane) {
  constexpr bool kBf16 = sizeof(T) == 2;
  for (int it = 0; it < 4; ++it) {
    const int q = lane + 64 * it;
    const int row = q >> 3, cg = q & 7;
    const int m = mbase + row, n = nbase + cg * 8;
    if (m >= p.M || n >= p.N) continue;
    const f32x4 s0 = *(const f32x4*)(stage + row * STG_LD + cg * 8);
    const f32x4 s1 = *(const f32x4*)(stage + row * STG_LD + cg * 8 + 4);
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    const int nv = min(8, p.N - n);
    const long long ci = zoffC + (long long)m * p.ldc + n;
    const long long ai = zoffC + (long long)m * p.ldaux + n;
    float aux[8];
    const bool has_aux = p.aux_mode != AUX_NONE;
    if (has_aux) {
      if (nv == 8 && p.c_vec) {
        if (kBf16) {
          const uint4 u = *(const uint4*)((const unsigned short*)p.aux + ai);
          const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { aux[2 * j] = __uint_as_float(w[j] << 16); aux[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
        } else {
          const float4 a0 = *(const float4*)((const float*)p.aux + ai), a1 = *(const float4*)((const float*)p.aux + ai + 4);
          aux[0] = a0.x; aux[1] = a0.y; aux[2] = a0.z; aux[3] = a0.w; aux[4] = a1.x; aux[5] = a1.y; aux[6] = a1.z; aux[7] = a1.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) aux[j] = j < nv ? Elem<T>::load(p.aux, ai + j) : 0.f;
      }
    }
    float pre[8];
    const bool drop = p.drop_p > 0.f;
    const unsigned long long idx0 = p.offset + ((unsigned long long)z * p.M + m) * p.N + n;
    const unsigned long long seed = drop ? *p.seed : 0ull;
    const unsigned thr = sc_drop_threshold(p.drop_p);
    const float keep = p.drop_keep;
    unsigned h = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float bias = (p.bias && j < nv) ? p.bias[n + j] : 0.f;
      pre[j] = v[j] * p.alpha + bias;
      bool dropped = false;
      if (drop) {
        const unsigned long long idx = idx0 + j;
        if (j == 0 || (idx & 1) == 0) h = sc_rand_pair(seed, idx >> 1);
        dropped = ((h >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr;
      }
      v[j] = epilogue_elem(p, v[j], bias, has_aux ? aux[j] : 0.f, dropped, keep);
      if (p.pre_grad) {
        const float x = pre[j];
        const float gprime = p.act == ACT_RELU ? (x > 0.f ? 1.f : 0.f) : (p.act == ACT_GELU ? sc_gelu_grad(x) : 1.f);
        pre[j] = dropped ? 0.f : gprime * keep;
      }
    }
    if (p.out_f32) {
      float* c = (float*)p.C + ci;
      if (nv == 8 && p.c_vec32) {
        float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
        if (p.accumulate) {
          const float4 c0 = *(const float4*)c, c1 = *(const float4*)(c + 4);
          o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w; o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
        }
        *(float4*)c = o0;
        *(float4*)(c + 4) = o1;
      } else {
        for (int j = 0; j < nv; ++j) c[j] = p.accumulate ? c[j] + v[j] : v[j];
      }
    } else if (kBf16) {
      unsigned short* c = (unsigned short*)p.C + ci;
      unsigned short* pr = p.pre_out ? (unsigned short*)p.pre_out + ci : nullptr;
      if (nv == 8 && p.c_vec) {
        uint4 o;
        o.x = sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
        o.y = sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
        o.z = sc_f32_to_bf16(v[4]) | ((unsigned)sc_f32_to_bf16(v[5]) << 16);
        o.w = sc_f32_to_bf16(v[6]) | ((unsigned)sc_f32_to_bf16(v[7]) << 16);
        *(uint4*)c = o;
        if (pr) {
          o.x = sc_f32_to_bf16(pre[0]) | ((unsigned)sc_f32_to_bf16(pre[1]) << 16);
          o.y = sc_f32_to_bf16(pre[2]) | ((unsigned)sc_f32_to_bf16(pre[3]) << 16);
          o.z = sc_f32_to_bf16(pre[4]) | ((unsigned)sc_f32_to_bf16(pre[5]) << 16);
          o.w = sc_f32_to_bf16(pre[6]) | ((unsigned)sc_f32_to_bf16(pre[7]) << 16);
          *(uint4*)pr = o;
        }
      } else {
        for (int j = 0; j < nv; ++j) {
          c[j] = sc_f32_to_bf16(v[j]);
          if (pr) pr[j] = sc_f32_to_bf16(pre[j]);
        }
      }
    } else {
      float* c = (float*)p.C + ci;
      float* pr = p.pre_out ? (float*)p.pre_out + ci : nullptr;
      if (nv == 8 && p.c_vec) {
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
        if (pr) {
          *(float4*)pr = make_float4(pre[0], pre[1], pre[2], pre[3]);
          *(float4*)(pr + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
        }
      } else {
        for (int j = 0; j < nv; ++j) {
          c[j] = v[j];
          if (pr) pr[j] = pre[j];
        }
      }
    }
  }
}

// Fast epilogue of a 32 x 64 staged sub-tile for the common cases: out = alpha * acc (+ bias[n]) stored in the compute dtype.
// No per-element mode branches: most GEMMs of a step (dgrad, attention products, projections without fused extras) land here.
template <typename T, bool BIAS>
__device__ __forceinline__ void epilogue_tile_fast(const GemmArgs& p, const float* stage, long long zoffC, int mbase, int nbase,
                                                   int lane) {
  constexpr bool kBf16 = sizeof(T) == 2;
  const int cg = lane & 7;
  const int n = nbase + cg * 8;
  if (n >= p.N) return;
  const int nv = min(8, p.N - n);
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = (BIAS && j < nv) ? p.bias[n + j] : 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it;
    const int m = mbase + row;
    if (m >= p.M) continue;
    const f32x4 s0 = *(const f32x4*)(stage + row * STG_LD + cg * 8);
    const f32x4 s1 = *(const f32x4*)(stage + row * STG_LD + cg * 8 + 4);
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], p.alpha, bias[j]);
    const long long ci = zoffC + (long long)m * p.ldc + n;
    if (kBf16) {
      unsigned short* c = (unsigned short*)p.C + ci;
      if (nv == 8 && p.c_vec) {
        uint4 o;
        o.x = sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
        o.y = sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
        o.z = sc_f32_to_bf16(v[4]) | ((unsigned)sc_f32_to_bf16(v[5]) << 16);
        o.w = sc_f32_to_bf16(v[6]) | ((unsigned)sc_f32_to_bf16(v[7]) << 16);
        *(uint4*)c = o;
      } else {
        for (int j = 0; j < nv; ++j) c[j] = sc_f32_to_bf16(v[j]);
      }
    } else {
      float* c = (float*)p.C + ci;
      if (nv == 8 && p.c_vec) {
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        for (int j = 0; j < nv; ++j) c[j] = v[j];
      }
    }
  }
}

// Fused epilogues with the feature set fixed at compile time (the fully run-time `epilogue_tile` stays as the fallback):
//   v = alpha*acc + bias;  [PRE: pre_out <- v];  ACTG ? v *= act'(aux) : v = act(v);  dropout (run-time p, hoisted);
//   [ADD: v += aux];  store in the compute dtype.   ACT: 0 none, 1 ReLU, 2 GELU.
template <typename T, bool BIAS, int ACT, bool ACTG, bool ADD, bool PRE, bool PREG = false>
__device__ __forceinline__ void epilogue_tile_fused(const GemmArgs& p, const float* stage, long long zoffC, int z, int mbase,
                                                    int nbase, int lane) {
  constexpr bool kBf16 = sizeof(T) == 2;
  constexpr bool AUX = ACTG || ADD;
  const int cg = lane & 7;
  const int n = nbase + cg * 8;
  if (n >= p.N) return;
  const int nv = min(8, p.N - n);
  const bool vec = nv == 8 && p.c_vec;
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = (BIAS && j < nv) ? p.bias[n + j] : 0.f;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long seed = drop ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
  const float keep = p.drop_keep;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it;
    const int m = mbase + row;
    if (m >= p.M) continue;
    const f32x4 s0 = *(const f32x4*)(stage + row * STG_LD + cg * 8);
    const f32x4 s1 = *(const f32x4*)(stage + row * STG_LD + cg * 8 + 4);
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    const long long ci = zoffC + (long long)m * p.ldc + n;
    float aux[8];
    if (AUX) {
      const long long ai = zoffC + (long long)m * p.ldaux + n;
      if (vec) {
        if (kBf16) {
          const uint4 u = *(const uint4*)((const unsigned short*)p.aux + ai);
          const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { aux[2 * j] = __uint_as_float(w[j] << 16); aux[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
        } else {
          const float4 a0 = *(const float4*)((const float*)p.aux + ai), a1 = *(const float4*)((const float*)p.aux + ai + 4);
          aux[0] = a0.x; aux[1] = a0.y; aux[2] = a0.z; aux[3] = a0.w; aux[4] = a1.x; aux[5] = a1.y; aux[6] = a1.z; aux[7] = a1.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) aux[j] = j < nv ? Elem<T>::load(p.aux, ai + j) : 0.f;
      }
    }
    float pre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = fmaf(v[j], p.alpha, bias[j]);
      pre[j] = v[j];
      if (ACTG) {
        if (ACT == ACT_RELU) v[j] = aux[j] > 0.f ? v[j] : 0.f;
        else if (ACT == ACT_GELU) v[j] *= sc_gelu_grad_t<kBf16>(aux[j]);
        else v[j] *= aux[j];  // AUX_MUL: the factor was prepared by the forward product (PREG)
      } else if (PREG && ACT == ACT_GELU) {
        // gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x): one erf serves both, and the erf approximation's own exp(-(x/sqrt 2)^2) IS
        // the Gaussian factor of phi(x)
        const float x = v[j];
        float g;  // exp(-x^2 / 2)
        const float e = kBf16 ? sc_erf_fast_g(x * 0.70710678118654752f, g) : sc_erf_g(x * 0.70710678118654752f, g);
        const float Phi = 0.5f * (1.f + e);
        v[j] = x * Phi;
        pre[j] = fmaf(x * 0.39894228040143268f, g, Phi);
      } else {
        if (PREG) pre[j] = (ACT == ACT_RELU) ? (v[j] > 0.f ? 1.f : 0.f) : 1.f;
        if (ACT == ACT_RELU) v[j] = fmaxf(v[j], 0.f);
        else if (ACT == ACT_GELU) v[j] = sc_gelu_t<kBf16>(v[j]);
      }
    }
    if (drop) {
      const unsigned long long idx0 = p.offset + ((unsigned long long)z * p.M + m) * p.N + n;
      unsigned hsh = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned long long idx = idx0 + j;
        if (j == 0 || (idx & 1) == 0) hsh = sc_rand_pair(seed, idx >> 1);
        const bool dropped = ((hsh >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr;
        v[j] = dropped ? 0.f : v[j] * keep;
        if (PREG) pre[j] = dropped ? 0.f : pre[j] * keep;
      }
    }
    if (ADD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += aux[j];
    }
    if (kBf16) {
      unsigned short* c = (unsigned short*)p.C + ci;
      unsigned short* pr = PRE ? (unsigned short*)p.pre_out + ci : nullptr;
      if (vec) {
        uint4 o;
        o.x = sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
        o.y = sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
        o.z = sc_f32_to_bf16(v[4]) | ((unsigned)sc_f32_to_bf16(v[5]) << 16);
        o.w = sc_f32_to_bf16(v[6]) | ((unsigned)sc_f32_to_bf16(v[7]) << 16);
        *(uint4*)c = o;
        if (PRE) {
          o.x = sc_f32_to_bf16(pre[0]) | ((unsigned)sc_f32_to_bf16(pre[1]) << 16);
          o.y = sc_f32_to_bf16(pre[2]) | ((unsigned)sc_f32_to_bf16(pre[3]) << 16);
          o.z = sc_f32_to_bf16(pre[4]) | ((unsigned)sc_f32_to_bf16(pre[5]) << 16);
          o.w = sc_f32_to_bf16(pre[6]) | ((unsigned)sc_f32_to_bf16(pre[7]) << 16);
          *(uint4*)pr = o;
        }
      } else {
        for (int j = 0; j < nv; ++j) {
          c[j] = sc_f32_to_bf16(v[j]);
          if (PRE) pr[j] = sc_f32_to_bf16(pre[j]);
        }
      }
    } else {
      float* c = (float*)p.C + ci;
      float* pr = PRE ? (float*)p.pre_out + ci : nullptr;
      if (vec) {
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
        if (PRE) {
          *(float4*)pr = make_float4(pre[0], pre[1], pre[2], pre[3]);
          *(float4*)(pr + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
        }
      } else {
        for (int j = 0; j < nv; ++j) {
          c[j] = v[j];
          if (PRE) pr[j] = pre[j];
        }
      }
    }
  }
}

// acc[MI][2] (a wave's (MI*32) x 64 result) -> private LDS staging slice -> epilogue functor, one 32 x 64 sub-tile at a
// time.  Instantiated once per epilogue variant so that every instance is a small, fully unrollable loop (a single loop
// that dispatched between variants per sub-tile was too big to unroll and pushed the accumulators into scratch).
template <int MI, typename F>
__device__ __forceinline__ void run_epilogue(float* stage, f32x16 (&acc)[MI][2], int lane, F&& f) {
  const int h = lane >> 5, li = lane & 31;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * STG_LD + j * 32 + li] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    f(i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
}

// Epilogue of the generic kernels (attention products, wgrad, f32 parity mode): plain / bias-only fast paths, else the
// run-time general epilogue.  mrow0 = first row of the wave's tile, ncol0 = first column.
template <typename T, int MI>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& p, float* stage, f32x16 (&acc)[MI][2], long long zoffC, int z,
                                              int mrow0, int ncol0, int lane) {
  const bool simple = p.act == ACT_NONE && p.aux_mode == AUX_NONE && p.drop_p == 0.f && !p.pre_out && !p.out_f32;
  if (simple && !p.bias)
    run_epilogue<MI>(stage, acc, lane, [&](int i) { epilogue_tile_fast<T, false>(p, stage, zoffC, mrow0 + i * 32, ncol0, lane); });
  else if (simple)
    run_epilogue<MI>(stage, acc, lane, [&](int i) { epilogue_tile_fast<T, true>(p, stage, zoffC, mrow0 + i * 32, ncol0, lane); });
  else
    run_epilogue<MI>(stage, acc, lane, [&](int i) { epilogue_tile<T>(p, stage, zoffC, z, mrow0 + i * 32, ncol0, lane); });
}

// Epilogue of the LDS-DMA kernel (forward projections / FFN layers / data gradients): additionally the compile-time fused
// variants used by the transformer layers.
template <typename T, int MI>
__device__ __forceinline__ void epilogue_wave_fused(const GemmArgs& p, float* stage, f32x16 (&acc)[MI][2], long long zoffC, int z,
                                                    int mrow0, int ncol0, int lane) {
#define SC_RUN(...) run_epilogue<MI>(stage, acc, lane, [&](int i) { __VA_ARGS__(p, stage, zoffC, z, mrow0 + i * 32, ncol0, lane); })
  if (!p.out_f32 && p.bias) {
    if (p.aux_mode == AUX_NONE && p.pre_out && p.pre_grad && p.act == ACT_GELU) { SC_RUN(epilogue_tile_fused<T, true, ACT_GELU, false, false, true, true>); return; }
    if (p.aux_mode == AUX_NONE && p.pre_out && p.pre_grad && p.act == ACT_RELU) { SC_RUN(epilogue_tile_fused<T, true, ACT_RELU, false, false, true, true>); return; }
    if (p.aux_mode == AUX_NONE && p.pre_out && !p.pre_grad && p.act == ACT_GELU) { SC_RUN(epilogue_tile_fused<T, true, ACT_GELU, false, false, true>); return; }
    if (p.aux_mode == AUX_NONE && p.pre_out && !p.pre_grad && p.act == ACT_RELU) { SC_RUN(epilogue_tile_fused<T, true, ACT_RELU, false, false, true>); return; }
    if (p.aux_mode == AUX_ADD && !p.pre_out && p.act == ACT_NONE) { SC_RUN(epilogue_tile_fused<T, true, ACT_NONE, false, true, false>); return; }
    if (p.aux_mode == AUX_NONE && !p.pre_out && p.act == ACT_NONE && p.drop_p > 0.f) { SC_RUN(epilogue_tile_fused<T, true, ACT_NONE, false, false, false>); return; }
  }
  if (!p.out_f32 && !p.bias && p.aux_mode == AUX_MUL && !p.pre_out && p.act == ACT_NONE && p.drop_p == 0.f) {
    SC_RUN(epilogue_tile_fused<T, false, ACT_NONE, true, false, false>);
    return;
  }
  if (!p.out_f32 && !p.bias && p.aux_mode == AUX_ACT_GRAD && !p.pre_out) {
    if (p.act == ACT_GELU) { SC_RUN(epilogue_tile_fused<T, false, ACT_GELU, true, false, false>); return; }
    if (p.act == ACT_RELU) { SC_RUN(epilogue_tile_fused<T, false, ACT_RELU, true, false, false>); return; }
  }
#undef SC_RUN
  epilogue_wave<T, MI>(p, stage, acc, zoffC, z, mrow0, ncol0, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 kernel
// ---------------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128;
constexpr int BK16 = 32, LD16 = 40;  // bf16: 32-deep K tile, LDS rows padded to 40 elements (80 B)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// One operand tile (128 rows x 32 k) global -> registers.  Non-transposed: chunk = 8 consecutive k of one row.
// Transposed: chunk = 8 consecutive rows of one k.
constexpr int LDT16 = 160;  // transposed-operand tile: 32 k-rows x 128 columns, rows padded to 160 elements (320 B) so the
                            // 32 lanes of a ds_read_b64_tr_b16 half-wave cover all 64 LDS banks exactly once

typedef short s16x4_t __attribute__((ext_vector_type(4)));

// One operand tile global -> registers.  Non-transposed (k contiguous in memory): tile = 128 rows x 32 k, chunk = 8
// consecutive k of one row.  Transposed (row index contiguous): tile = 32 k-rows x 128 rows, chunk = 8 consecutive rows of
// one k -- still a coalesced 16-byte load; the transposition happens in the LDS *read* (ds_read_b64_tr_b16).
template <bool TR>
__device__ __forceinline__ void load_tile16(const unsigned short* base, long long ld, int rows, int K, int row0, int k0,
                                            bool vec, int tid, uint4 (&reg)[2]) {
  // wave-uniform fast path: interior tile, 16-byte aligned -> two unconditional 16-byte loads per lane (keeps the
  // compiler's vmcnt bookkeeping counted, so three tiles really stay in flight)
  const bool full = vec && (row0 + BM <= rows) && (k0 + BK16 <= K);
  if (full) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      const unsigned short* src = !TR ? base + (long long)(row0 + (c >> 2)) * ld + k0 + (c & 3) * 8
                                      : base + (long long)(k0 + (c >> 4)) * ld + row0 + (c & 15) * 8;
      reg[i] = *(const uint4*)src;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    int r, k, nvalid;
    const unsigned short* src;
    if (!TR) {
      r = row0 + (c >> 2);
      k = k0 + (c & 3) * 8;
      nvalid = (r < rows) ? min(8, K - k) : 0;
      src = base + (long long)r * ld + k;
    } else {
      k = k0 + (c >> 4);
      r = row0 + (c & 15) * 8;
      nvalid = (k < K) ? min(8, rows - r) : 0;
      src = base + (long long)k * ld + r;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nvalid >= 8 && vec) {
      v = *(const uint4*)src;
    } else if (nvalid > 0) {
      unsigned short tmp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) tmp[j] = (j < nvalid) ? src[j] : (unsigned short)0;
      v.x = tmp[0] | ((unsigned)tmp[1] << 16); v.y = tmp[2] | ((unsigned)tmp[3] << 16);
      v.z = tmp[4] | ((unsigned)tmp[5] << 16); v.w = tmp[6] | ((unsigned)tmp[7] << 16);
    }
    reg[i] = v;
  }
}

template <bool TR>
__device__ __forceinline__ void store_tile16(unsigned short* lds, int tid, const uint4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if (!TR) *(uint4*)(lds + (c >> 2) * LD16 + (c & 3) * 8) = reg[i];
    else *(uint4*)(lds + (c >> 4) * LDT16 + (c & 15) * 8) = reg[i];
  }
}

// MFMA 32x32x16 operand fragment of lane l: 8 consecutive k (k = ks*16 + 8*(l>>5) + j) of tile row `rbase + (l & 31)`.
template <bool TR>
__device__ __forceinline__ bf16x8_t read_frag16(const unsigned short* tile, int rbase, int ks, int lane) {
  if (!TR) {
    return *(const bf16x8_t*)(tile + (rbase + (lane & 31)) * LD16 + ks * 16 + (lane >> 5) * 8);
  } else {
    // hardware-transposing LDS read: per 16-lane group a 4(k) x 16(row) block; lane 4q+p supplies the address of
    // k-row q, columns 4p..4p+3 and receives column (lane & 15) of the four k-rows.
    const int g = lane >> 4, ip = lane & 15, q = ip >> 2, pp = ip & 3;
    const unsigned short* a = tile + (ks * 16 + 8 * (g >> 1) + q) * LDT16 + rbase + 16 * (g & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * LDT16));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 2 * BM * LD16];
  unsigned short* As = smem;
  unsigned short* Bs = smem + 2 * BM * LD16;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int h = lane >> 5, li = lane & 31;
  // XCD-aware tile order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (private L2s); remap so
  // that each XCD works on a contiguous run of tiles -- the N-tiles sharing one A row-panel then hit the same L2.
  // (Speed only: any placement gives the same result.)  Bijective for every grid size.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y, orig = by * gx + bx;
    if (nwg >= 16) {
      const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
      const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      by = id / gx;
      bx = id - by * gx;
    }
  }
  const int n0 = bx * BN, m0 = by * BM;
  const int z = blockIdx.z, z1 = z / p.nb2, z2 = z - z1 * p.nb2;
  const unsigned short* A = (const unsigned short*)p.A + z1 * p.sA1 + z2 * p.sA2;
  const unsigned short* B = (const unsigned short*)p.B + z1 * p.sB1 + z2 * p.sB2;
  const long long zoffC = z1 * p.sC1 + z2 * p.sC2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Column sums of A over the contraction (the bias gradient of a wgrad product: A = dz read transposed).  The A tile
  // passes through registers on its way to LDS and thread t always holds the same 8 columns m0 + (t & 15) * 8 .. + 7
  // (rows (t >> 4) and (t >> 4) + 16 of each K tile), so the workgroups of the first N tile add them up on the way.
  const bool csum = TA && p.colsum_part != nullptr && n0 == 0;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#define SC_CSUM(RA)                                                                     \
  do {                                                                                  \
    if (csum) {                                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                \
        const unsigned w_[4] = {RA[i_].x, RA[i_].y, RA[i_].z, RA[i_].w};                \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                              \
          cs[2 * e_] += __uint_as_float(w_[e_] << 16);                                  \
          cs[2 * e_ + 1] += __uint_as_float(w_[e_] & 0xffff0000u);                      \
        }                                                                               \
      }                                                                                 \
    }                                                                                   \
  } while (0)

  // K range of this workgroup (split-K: slice z of the contraction)
  const int kbeg = p.splitk > 1 ? z * p.ksplit_len : 0;
  const int kend = p.splitk > 1 ? min(p.K, kbeg + p.ksplit_len) : p.K;
  const int nk = (kend - kbeg + BK16 - 1) / BK16;

  // 3-deep register prefetch ring (tiles t+1..t+3 in flight while tile t is multiplied) over double-buffered LDS:
  // the K loop is latency-bound at these problem sizes (~250 workgroups, 16-64 K tiles), not bandwidth-bound.
  uint4 ra0[2], rb0[2], ra1[2], rb1[2], ra2[2], rb2[2];
#define SC_LOAD(RA, RB, T)                                                              \
  do {                                                                                  \
    if ((T) < nk) {                                                                     \
      load_tile16<TA>(A, p.lda, p.M, kend, m0, kbeg + (T) * BK16, p.a_vec, tid, RA);    \
      load_tile16<TB>(B, p.ldb, p.N, kend, n0, kbeg + (T) * BK16, p.b_vec, tid, RB);    \
    }                                                                                   \
  } while (0)
#define SC_COMPUTE(CUR)                                                                                         \
  do {                                                                                                          \
    const unsigned short* as = As + (CUR) * BM * LD16;                                                          \
    const unsigned short* bs = Bs + (CUR) * BM * LD16;                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
      bf16x8_t a[2], b[2];                                                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
        a[i] = read_frag16<TA>(as, wr * 64 + i * 32, ks, lane);                                                 \
        b[i] = read_frag16<TB>(bs, wc * 64 + i * 32, ks, lane);                                                 \
      }                                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);                  \
    }                                                                                                           \
  } while (0)
#define SC_STEP(T, RLOAD_A, RLOAD_B, RNEXT_A, RNEXT_B)                                  \
  do {                                                                                  \
    SC_LOAD(RLOAD_A, RLOAD_B, (T) + 3);                                                 \
    SC_COMPUTE((T) & 1);                                                                \
    if ((T) + 1 < nk) {                                                                 \
      store_tile16<TA>(As + (((T) + 1) & 1) * BM * LD16, tid, RNEXT_A);                 \
      store_tile16<TB>(Bs + (((T) + 1) & 1) * BM * LD16, tid, RNEXT_B);                 \
      SC_CSUM(RNEXT_A);                                                                 \
    }                                                                                   \
    __syncthreads();                                                                    \
  } while (0)

  // interior, aligned workgroups run a branch-free steady state (unconditional 16-byte loads -> the compiler keeps
  // counted vmcnt waits, so two tiles stay in flight across every barrier); edges / tails take the guarded steps.
  const bool fast = p.a_vec && p.b_vec && (m0 + BM <= p.M) && (n0 + BN <= p.N);
#define SC_LOAD_FAST(RA, RB, T)                                                                                   \
  do {                                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                            \
      const int c_ = tid + 256 * i_;                                                                              \
      const int kk_ = kbeg + (T) * BK16;                                                                          \
      RA[i_] = *(const uint4*)(!TA ? A + (long long)(m0 + (c_ >> 2)) * p.lda + kk_ + (c_ & 3) * 8                 \
                                   : A + (long long)(kk_ + (c_ >> 4)) * p.lda + m0 + (c_ & 15) * 8);              \
      RB[i_] = *(const uint4*)(!TB ? B + (long long)(n0 + (c_ >> 2)) * p.ldb + kk_ + (c_ & 3) * 8                 \
                                   : B + (long long)(kk_ + (c_ >> 4)) * p.ldb + n0 + (c_ & 15) * 8);              \
    }                                                                                                             \
  } while (0)
#define SC_STEP_FAST(T, RLOAD_A, RLOAD_B, RNEXT_A, RNEXT_B)                             \
  do {                                                                                  \
    SC_LOAD_FAST(RLOAD_A, RLOAD_B, (T) + 3);                                            \
    SC_COMPUTE((T) & 1);                                                                \
    store_tile16<TA>(As + (((T) + 1) & 1) * BM * LD16, tid, RNEXT_A);                   \
    store_tile16<TB>(Bs + (((T) + 1) & 1) * BM * LD16, tid, RNEXT_B);                   \
    SC_CSUM(RNEXT_A);                                                                   \
    __syncthreads();                                                                    \
  } while (0)

  int t = 0;
  const int nfull = (kend - kbeg) / BK16;  // tiles that lie completely inside [kbeg, kend)
  if (fast && nfull >= 6) {
    SC_LOAD_FAST(ra0, rb0, 0);
    SC_LOAD_FAST(ra1, rb1, 1);
    SC_LOAD_FAST(ra2, rb2, 2);
    store_tile16<TA>(As, tid, ra0);
    store_tile16<TB>(Bs, tid, rb0);
    SC_CSUM(ra0);
    __syncthreads();
    for (; t + 6 <= nfull; t += 3) {  // steps t, t+1, t+2 load tiles t+3, t+4, t+5 (all full)
      SC_STEP_FAST(t, ra0, rb0, ra1, rb1);
      SC_STEP_FAST(t + 1, ra1, rb1, ra2, rb2);
      SC_STEP_FAST(t + 2, ra2, rb2, ra0, rb0);
    }
    // here: tile t is in LDS[t & 1], tiles t+1 / t+2 are in ring slots 1 / 2 (t is a multiple of 3)
  } else {
    SC_LOAD(ra0, rb0, 0);
    SC_LOAD(ra1, rb1, 1);
    SC_LOAD(ra2, rb2, 2);
    if (nk > 0) {
      store_tile16<TA>(As, tid, ra0);
      store_tile16<TB>(Bs, tid, rb0);
      SC_CSUM(ra0);
    }
    __syncthreads();
  }
  for (; t < nk; t += 3) {
    SC_STEP(t, ra0, rb0, ra1, rb1);
    if (t + 1 < nk) SC_STEP(t + 1, ra1, rb1, ra2, rb2);
    if (t + 2 < nk) SC_STEP(t + 2, ra2, rb2, ra0, rb0);
  }
#undef SC_LOAD_FAST
#undef SC_STEP_FAST
#undef SC_LOAD
#undef SC_COMPUTE
#undef SC_STEP
#undef SC_CSUM

  if (csum) {  // 16 row-groups x 128 columns through LDS (the tiles are dead: the loop ended on a barrier)
    float* red = (float*)smem;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < p.M) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += red[g * 128 + tid];
      p.colsum_part[(long long)z * p.M + m0 + tid] = v;
    }
    __syncthreads();  // the epilogue reuses smem as its staging tile
  }

  if (p.splitk > 1) {  // raw partial tile -> slab z (f32, no epilogue); the reduce kernel finishes the job
    float* stage = (float*)smem + wid * 32 * STG_LD;
    float* slab = (float*)p.C + (long long)z * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * STG_LD + j * 32 + li] = acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      for (int it = 0; it < 8; ++it) {
        const int q = lane + 64 * it;
        const int row = q >> 4, c4 = (q & 15) * 4;
        const int m = m0 + wr * 64 + i * 32 + row, n = n0 + wc * 64 + c4;
        if (m < p.M && n < p.N) {
          const f32x4 v = *(const f32x4*)(stage + row * STG_LD + c4);
          float* dst = slab + (long long)m * p.ldc + n;
          if (n + 4 <= p.N && p.c_vec32) *(f32x4*)dst = v * p.alpha;
          else for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e] * p.alpha;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    return;
  }
  epilogue_wave<unsigned short, 2>(p, (float*)smem + wid * 32 * STG_LD, acc, zoffC, z, m0 + wr * 64, n0 + wc * 64, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 NT kernel, LDS-DMA pipeline (both operands k-contiguous, K % 32 == 0, 16-byte aligned rows)
//
// global_load_lds_dwordx4 writes tiles straight into LDS (no VGPRs, no ds_write -- the VGPR->LDS store path was the
// bottleneck of the register-staged kernel: 16 KB of ds_write_b128 per K tile cost more LDS cycles than the MFMAs).
// Four 16 KB stages, three tiles in flight behind counted vmcnt waits and ONE raw s_barrier per K tile.
// An LDS-DMA wave-instruction writes 1 KB linearly (16 rows x 64 B), so rows cannot be padded; bank conflicts of the
// ds_read_b128 fragment reads are removed by XOR-swizzling the 16-byte chunk index with (row >> 2) & 3 -- applied to the
// per-lane SOURCE address when staging and to the read address (the same involution on both sides).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* sc_gptr_t;
typedef __attribute__((address_space(3))) void* sc_lptr_t;

// Tile configuration: WM x WN waves, each owning MI x 2 accumulators of 32 x 32  ->  tile (WM*MI*32) x (WN*64).
//   <2,2,2> 128 x 128, 4 waves, 4 stages (64 KB, 2 workgroups / CU)   -- many small tiles for narrow outputs
//   <2,4,4> 256 x 256, 8 waves, 4 stages (128 KB, 1 workgroup / CU)   -- half the L2->LDS traffic per flop; the mid-size
//           GEMMs of this model (M ~ 8000, K = 512) are bound by exactly that traffic, not by the matrix cores
// PF: how the MFMA fragments are fetched from LDS.
//   0  per K tile: wait + barrier, then read fragments, then multiply (the reads' latency is exposed twice per tile; with one
//      wave per SIMD -- every launch of at most 256 workgroups of 4 waves -- nothing else fills it: ~1000 cycles per tile for
//      256 cycles of MFMA work)
//   1  software-pipelined by half tiles: the fragments of (tile, ks+1) are read while (tile, ks) is multiplied; the barrier
//      that publishes tile t+1 sits between the two halves of tile t                       (2 x (MI + 2) fragments live)
//   2  software-pipelined by whole tiles: tile t+1's fragments are read while tile t is multiplied (4 x (MI + 2) fragments)
// In 1 and 2 the barrier of iteration t publishes tile t+1 (not t), all waves have tile t in registers by then, so the stage of
// tile t is free and receives tile t + STAGES: STAGES - 1 tiles in flight behind the one being read.
// KTAIL (PF 0 only): K is a multiple of 8 but not of 32 (the 5000-wide contraction of the vocabulary heads' data gradients).  The last
// K tile is partial: lanes whose 16-byte source chunk lies beyond K read chunk 0 of that tile instead (valid memory), and the fragments
// of the k-chunks beyond K are zeroed in registers before they reach the matrix cores.
// -DSC_GEMM_TIMING (instrumented builds only, scripts/gemm_probe.py): every workgroup records s_memrealtime (100 MHz, chip-wide) and s_memtime
// (shader cycles) at its phase boundaries, wave 0 lane 0; workgroup 0 also stamps four steady-state K tiles per wave.  Read back with
// sc_gemm_debug_read (not part of the ABI).
#ifdef SC_GEMM_TIMING
__device__ long long sc_gemm_probe[1024 * 8 + 8 * 4 * 4];
#define GT_WG(k)                                                                                                                \
  do {                                                                                                                          \
    const int id_ = blockIdx.y * gridDim.x + blockIdx.x;                                                                        \
    if (lane == 0 && wid == 0 && id_ < 1024) {                                                                                  \
      sc_gemm_probe[id_ * 8 + (k)] = __builtin_readcyclecounter();                                                              \
      if ((k) == 0 || (k) == 3) sc_gemm_probe[id_ * 8 + 4 + ((k) == 3)] = __builtin_amdgcn_s_memrealtime();                     \
      if ((k) == 0) {                                                                                                           \
        unsigned hw_, xcc_;                                                                                                     \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                     \
        sc_gemm_probe[id_ * 8 + 6] = ((long long)xcc_ << 32) | hw_;                                                             \
      }                                                                                                                         \
    }                                                                                                                           \
  } while (0)
#define GT_IT(t, k)                                                                                                             \
  do {                                                                                                                          \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (t) >= 8 && (t) < 12)                                                \
      sc_gemm_probe[1024 * 8 + (wid * 4 + ((t) - 8)) * 4 + (k)] = __builtin_readcyclecounter();                                 \
  } while (0)
#else
#define GT_WG(k) do {} while (0)
#define GT_IT(t, k) do {} while (0)
#endif

// HALFLOAD: only the first half of the waves (the older wave of every SIMD pair) issues the LDS-DMA instructions, twice as many each.  Two
// waves share a SIMD; the one dispatched first wins the issue arbitration, the younger one reaches the barrier later by about the cost
// of its DMA issue (scripts/gemm_probe.py): giving that work to the older half evens the two out.
template <int WM, int WN, int MI, int PF, bool KTAIL = false, bool HALFLOAD = false, int SG = 4>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1) void gemm_bf16_nt_dma_kernel(GemmArgs p) {
  static_assert(!KTAIL || PF == 0, "the K tail is implemented in the unpipelined loop only");
  constexpr int NW = WM * WN, TM = WM * MI * 32, TN = WN * 64;
  constexpr int NL = HALFLOAD ? NW / 2 : NW;  // waves that stage operands
  static_assert(!HALFLOAD || (NW == 8 && !KTAIL), "HALFLOAD is built for the 8-wave tiles");
  constexpr int STAGES = SG, STAGE_ELEMS = (TM + TN) * BK16;  // (SG: ring depth, a power of two; 4 everywhere but in the depth experiment)
  static_assert((SG & (SG - 1)) == 0 && SG >= 4 && SG <= 8, "ring depth");
  constexpr int ACH = TM / 16 / NL, BCH = TN / 16 / NL;  // 1 KB chunks (16 rows) of A / B each staging wave moves per tile
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsmem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid % WN;
  const int h = lane >> 5, li = lane & 31;
  const bool loader = !HALFLOAD || wid < NL;
  if (p.exp_flags & 4) {  // (experiment: static priority for the younger half of the waves)
    if (wid >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  GT_WG(0);
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y, orig = by * gx + bx;
    if (nwg >= 16) {
      const int xcd = ((orig & 7) + ((p.exp_flags >> 3) & 7)) & 7, q = nwg >> 3, r = nwg & 7;  // (experiment bits 3..5: rotate the labels)
      const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      by = id / gx;
      bx = id - by * gx;
    }
  }
  const int n0 = bx * TN, m0 = by * TM;
  const int z = blockIdx.z, z1 = z / p.nb2, z2 = z - z1 * p.nb2;
  const unsigned short* A = (const unsigned short*)p.A + z1 * p.sA1 + z2 * p.sA2;
  const unsigned short* B = (const unsigned short*)p.B + z1 * p.sB1 + z2 * p.sB2;
  const long long zoffC = z1 * p.sC1 + z2 * p.sC2;

  // per-lane source pointers of this wave's chunks: chunk c covers tile rows 16c .. 16c+15; lane l fills LDS slot
  // (row 16c + l/4, 16-byte slot l&3) from source chunk (l&3) ^ ((row >> 2) & 3).  Rows beyond the matrix are clamped to
  // the last valid row (their products are never stored).
  const unsigned short* asrc[ACH];
  const unsigned short* bsrc[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int row = ((wid % NL) + NL * i) * 16 + (lane >> 2);
    asrc[i] = A + (long long)min(m0 + row, p.M - 1) * p.lda + ((lane & 3) ^ ((row >> 2) & 3)) * 8;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int row = ((wid % NL) + NL * i) * 16 + (lane >> 2);
    bsrc[i] = B + (long long)min(n0 + row, p.N - 1) * p.ldb + ((lane & 3) ^ ((row >> 2) & 3)) * 8;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = KTAIL ? (p.K + BK16 - 1) / BK16 : p.K / BK16;
  const int ktail = KTAIL ? p.K % BK16 : 0;  // (8, 16 or 24 when KTAIL)
  // source chunk of this lane inside a K tile ((row >> 2) & 3 == (lane >> 4) & 3 for every chunk row of every wave) and how far to step
  // back in the last tile when that chunk lies beyond K
  const int tail_adj = (KTAIL && (((lane & 3) ^ ((lane >> 4) & 3)) * 8 >= ktail)) ? ((lane & 3) ^ ((lane >> 4) & 3)) * 8 : 0;
#define SC_DMA_ISSUE(T)                                                                                                \
  do {                                                                                                                 \
    if (loader) {                                                                                                      \
      unsigned short* st_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                                                      \
      const int koff_ = (T) * BK16 - ((KTAIL && (T) == nk - 1) ? tail_adj : 0);                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < ACH; ++i_)                                                               \
        __builtin_amdgcn_global_load_lds((sc_gptr_t)(asrc[i_] + koff_), (sc_lptr_t)(st_ + ((wid % NL) + NL * i_) * 512), 16, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < BCH; ++i_)                                                               \
        __builtin_amdgcn_global_load_lds((sc_gptr_t)(bsrc[i_] + koff_),                                                \
                                         (sc_lptr_t)(st_ + TM * BK16 + ((wid % NL) + NL * i_) * 512), 16, 0, 0);       \
    }                                                                                                                  \
  } while (0)

#pragma unroll
  for (int i = 0; i < (PF != 0 ? STAGES : STAGES - 1); ++i)
    if (nk > i) SC_DMA_ISSUE(i);

  // fragment read offsets (elements) inside an operand tile: row = base + li, 16-byte chunk (2*ks + h) swizzled
  int aoff[MI][2], boff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ra = (wr * MI + i) * 32 + li;
      aoff[i][ks] = ra * BK16 + (((2 * ks + h) ^ ((ra >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rb = (wc * 2 + j) * 32 + li;
      boff[j][ks] = rb * BK16 + (((2 * ks + h) ^ ((rb >> 2) & 3)) * 8);
    }
  }

  constexpr int LPT = ACH + BCH;  // DMA instructions per wave per tile
#define SC_WAIT_TILES(AHEAD)                                                                  \
  do {                                                                                        \
    const int ah_ = (AHEAD);                                                                  \
    if (!loader) break; /* (a wave that stages nothing has nothing in flight) */              \
    if (STAGES > 4 && ah_ >= 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * LPT) : "memory"); \
    else if (STAGES > 4 && ah_ == 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LPT) : "memory"); \
    else if (STAGES > 4 && ah_ == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LPT) : "memory"); \
    else if (STAGES > 4 && ah_ == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT) : "memory"); \
    else if (ah_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");         \
    else if (ah_ == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");         \
    else if (ah_ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");             \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
  } while (0)
#define SC_FRAGS(T, KS, FA, FB)                                                               \
  do {                                                                                        \
    const unsigned short* as_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                         \
    const unsigned short* bs_ = as_ + TM * BK16;                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) FA[i_] = *(const bf16x8_t*)(as_ + aoff[i_][KS]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) FB[j_] = *(const bf16x8_t*)(bs_ + boff[j_][KS]);  \
  } while (0)
#define SC_MMA(FA, FB)                                                                        \
  do {                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                        \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i_], FB[j_], acc[i_][j_], 0, 0, 0); \
  } while (0)
// tile T+1 has landed for this wave; after the barrier it is visible to all, and every wave holds tile T in registers
#define SC_PUBLISH_NEXT(T)                                                                    \
  do {                                                                                        \
    SC_WAIT_TILES(min(STAGES - 2, nk - 2 - (T)));                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                             \
    if ((T) + STAGES < nk) SC_DMA_ISSUE((T) + STAGES);                                        \
  } while (0)
#define SC_PUBLISH_STEADY(T)                                                                  \
  do {                                                                                        \
    GT_IT(T, 0);                                                                              \
    if (loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    GT_IT(T, 1);                                                                              \
    __builtin_amdgcn_s_barrier();                                                             \
    GT_IT(T, 2);                                                                              \
    SC_DMA_ISSUE((T) + STAGES);                                                               \
    GT_IT(T, 3);                                                                              \
  } while (0)
  if constexpr (PF == 0) {
  for (int t = 0; t < nk; ++t) {
    // tile t must have landed: every wave waits for its own DMA (counted: later tiles stay in flight), then the barrier
    // makes all waves' pieces visible and proves everyone is done reading the stage that tile t+3 overwrites.
    const int ahead = min(2, nk - 1 - t);
    if (!loader) {}
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 3 < nk) SC_DMA_ISSUE(t + 3);
    const unsigned short* as = dsmem + (t % STAGES) * STAGE_ELEMS;
    const unsigned short* bs = as + TM * BK16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[MI], b[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8_t*)(as + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8_t*)(bs + boff[j][ks]);
      if (KTAIL && t == nk - 1 && (2 * ks + h) * 8 >= ktail) {  // this lane's k-chunk of the partial tile lies beyond K
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)0.f;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  } else if constexpr (PF == 1) {
    bf16x8_t fa0[MI], fb0[2], fa1[MI], fb1[2];
    SC_WAIT_TILES(min(STAGES - 1, nk - 1));
    __builtin_amdgcn_s_barrier();
    SC_FRAGS(0, 0, fa0, fb0);
    int t = 0;
    // steady state, branch-free (a conditional around the reads makes the compiler wait lgkmcnt(0) at the merge point, which
    // serialises reads and MFMAs again): tile t + STAGES exists, so every wait is the full-depth one
    for (; t + STAGES < nk; ++t) {
      SC_FRAGS(t, 1, fa1, fb1);
      SC_MMA(fa0, fb0);
      SC_PUBLISH_STEADY(t);
      SC_FRAGS(t + 1, 0, fa0, fb0);
      SC_MMA(fa1, fb1);
    }
    for (; t < nk; ++t) {
      SC_FRAGS(t, 1, fa1, fb1);
      SC_MMA(fa0, fb0);
      if (t + 1 < nk) {
        SC_PUBLISH_NEXT(t);
        SC_FRAGS(t + 1, 0, fa0, fb0);
      }
      SC_MMA(fa1, fb1);
    }
  } else {
    bf16x8_t fa00[MI], fb00[2], fa01[MI], fb01[2], fa10[MI], fb10[2], fa11[MI], fb11[2];
    SC_WAIT_TILES(min(STAGES - 1, nk - 1));
    __builtin_amdgcn_s_barrier();
    GT_WG(1);
    SC_FRAGS(0, 0, fa00, fb00);
    SC_FRAGS(0, 1, fa01, fb01);
    int t = 0;
    for (; t + STAGES + 1 < nk; t += 2) {  // steady state, branch-free: tiles t + STAGES and t + 1 + STAGES exist
      SC_PUBLISH_STEADY(t);
      SC_FRAGS(t + 1, 0, fa10, fb10);
      SC_FRAGS(t + 1, 1, fa11, fb11);
      SC_MMA(fa00, fb00);
      SC_MMA(fa01, fb01);
      SC_PUBLISH_STEADY(t + 1);
      SC_FRAGS(t + 2, 0, fa00, fb00);
      SC_FRAGS(t + 2, 1, fa01, fb01);
      SC_MMA(fa10, fb10);
      SC_MMA(fa11, fb11);
    }
    for (; t < nk; t += 2) {
      if (t + 1 < nk) {
        SC_PUBLISH_NEXT(t);
        SC_FRAGS(t + 1, 0, fa10, fb10);
        SC_FRAGS(t + 1, 1, fa11, fb11);
      }
      SC_MMA(fa00, fb00);
      SC_MMA(fa01, fb01);
      if (t + 1 < nk) {
        if (t + 2 < nk) {
          SC_PUBLISH_NEXT(t + 1);
          SC_FRAGS(t + 2, 0, fa00, fb00);
          SC_FRAGS(t + 2, 1, fa01, fb01);
        }
        SC_MMA(fa10, fb10);
        SC_MMA(fa11, fb11);
      }
    }
  }
#undef SC_PUBLISH_STEADY
#undef SC_WAIT_TILES
#undef SC_FRAGS
#undef SC_MMA
#undef SC_PUBLISH_NEXT
#undef SC_DMA_ISSUE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves are done with the operand stages before they become epilogue staging
  GT_WG(2);

  if constexpr (KTAIL)  // (data gradients: plain / bias / run-time general epilogue; keeps the number of instantiations down)
    epilogue_wave<unsigned short, MI>(p, (float*)dsmem + wid * 32 * STG_LD, acc, zoffC, z, m0 + wr * MI * 32, n0 + wc * 64, lane);
  else
    epilogue_wave_fused<unsigned short, MI>(p, (float*)dsmem + wid * 32 * STG_LD, acc, zoffC, z, m0 + wr * MI * 32, n0 + wc * 64, lane);
#ifdef SC_GEMM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores have left the wave)
  GT_WG(3);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 TN split-K kernel on the same LDS-DMA pipeline: the wgrad product dW = dz^T x, both operands read TRANSPOSED
// (contraction index = token = slow dimension, rows contiguous along the features).
//
// An operand tile is 32 k-rows x 128 columns (256-byte rows); one LDS-DMA wave-instruction moves 4 k-rows (1 KB).
// MFMA fragments come from the transposing read ds_read_b64_tr_b16, whose 16-lane groups touch 4 consecutive k-rows at
// the same column offsets -- with linear 256-byte rows those four addresses share their banks (4-way conflict), and DMA
// rows cannot be padded.  So the 16-byte chunk index is XOR-swizzled with 4 * (k-row & 3) on the SOURCE side of the DMA
// (lane l of a granule fills LDS slot (row l >> 4, chunk l & 15) from source chunk (l & 15) ^ (4 (l >> 4))) and by the
// same involution in the read address: the 16 lanes of a group then cover 16 distinct 8-byte bank pairs.
// Split-K only (raw f32 partial tiles to slab z, summed by splitk_reduce_kernel); needs K % 32 == 0, M % 8 == N % 8 == 0
// and 16-byte aligned rows -- anything else takes the register-staged kernel.
// STATUS: correct (tests/test_gemm_gpu.py drives it through the C ABI) but NOT the default: on the train step it measured
// 16.03 ms vs 15.83 ms for the register-staged TN kernel (GEMM aggregate 233 vs 251 TFLOP/s, same box, same commit).  With a
// 32-token K tile a wave issues only 8 MFMAs per barrier, so the per-tile overhead (barrier, counted wait, tr reads, 64-bit
// source offsets) dominates in both kernels; the staging path was not the limiter here.  Enabled with SC_GEMM_TN_DMA=1.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t read_frag16_tr_swz(const unsigned short* tile, int rbase, int ks, int lane) {
  const int g = lane >> 4, ip = lane & 15, q = ip >> 2, pp = ip & 3;
  const int kr = ks * 16 + 8 * (g >> 1) + q;         // k-row of the `lo` read (kr & 3 == q); `hi` is k-row kr + 4: next granule
  const int col = rbase + 16 * (g & 1) + 4 * pp;
  const unsigned short* a = tile + (kr >> 2) * 512 + q * 128 + (((col >> 3) ^ (q << 2)) << 3) + (col & 7);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a + 512));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_dma_splitk_kernel(GemmArgs p) {
  constexpr int STAGES = 4, OPER_ELEMS = BM * BK16, STAGE_ELEMS = 2 * OPER_ELEMS;  // 8 KB per operand tile, 16 KB per stage
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsmem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int h = lane >> 5, li = lane & 31;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y, orig = by * gx + bx;
    if (nwg >= 16) {
      const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
      const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      by = id / gx;
      bx = id - by * gx;
    }
  }
  const int n0 = bx * BN, m0 = by * BM;
  const int z = blockIdx.z;
  const int kbeg = z * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int nk = (kend - kbeg) / BK16;  // whole tiles only (K % 32 == 0 is a launch condition)

  // Each wave stages granules wid and wid + 4 of both operands: granule gch = k-rows 4 gch .. 4 gch + 3.  Lane l fills LDS
  // slot (row l >> 4, chunk l & 15) from source chunk (l & 15) ^ (4 (l >> 4)); column chunks beyond the matrix are clamped to
  // the last valid chunk (their products are never stored).
  const unsigned short* A = (const unsigned short*)p.A;
  const unsigned short* B = (const unsigned short*)p.B;
  const int q = lane >> 4, schunk = (lane & 15) ^ (q << 2);
  const int acol = min(m0 + schunk * 8, p.M - 8), bcol = min(n0 + schunk * 8, p.N - 8);
  const unsigned short* asrc[2];
  const unsigned short* bsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int krow = (wid + 4 * i) * 4 + q;
    asrc[i] = A + (long long)(kbeg + krow) * p.lda + acol;
    bsrc[i] = B + (long long)(kbeg + krow) * p.ldb + bcol;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define SC_TN_ISSUE(T)                                                                                                  \
  do {                                                                                                                  \
    unsigned short* st_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                  \
      __builtin_amdgcn_global_load_lds((sc_gptr_t)(asrc[i_] + (long long)(T) * BK16 * p.lda),                           \
                                       (sc_lptr_t)(st_ + (wid + 4 * i_) * 512), 16, 0, 0);                              \
      __builtin_amdgcn_global_load_lds((sc_gptr_t)(bsrc[i_] + (long long)(T) * BK16 * p.ldb),                           \
                                       (sc_lptr_t)(st_ + OPER_ELEMS + (wid + 4 * i_) * 512), 16, 0, 0);                 \
    }                                                                                                                   \
  } while (0)

  if (nk > 0) SC_TN_ISSUE(0);
  if (nk > 1) SC_TN_ISSUE(1);
  if (nk > 2) SC_TN_ISSUE(2);
  if (nk > 3) SC_TN_ISSUE(3);

  // bias gradient: column sums of A over the contraction, taken from the landed A tiles by the workgroups of the first
  // N tile.  Thread t reads column chunk t & 15 of k-rows (t >> 4) and (t >> 4) + 16 -- the same ownership as in the
  // register-staged kernel, so the final cross-thread reduction is identical.
  const bool csum = p.colsum_part != nullptr && n0 == 0;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  int csoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kr = (tid >> 4) + 16 * i;
    csoff[i] = (kr >> 2) * 512 + (kr & 3) * 128 + (((tid & 15) ^ ((kr & 3) << 2)) << 3);
  }

  // Software-pipelined as the NT kernel (PF = 2): the barrier of iteration t publishes tile t+1, whose fragments are read
  // while tile t is multiplied; the stage of tile t then receives tile t + STAGES.
  constexpr int LPT = 4;  // DMA instructions per wave per tile
#define SC_WAIT_TILES(AHEAD)                                                                  \
  do {                                                                                        \
    const int ah_ = (AHEAD);                                                                  \
    if (ah_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");              \
    else if (ah_ == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");         \
    else if (ah_ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");             \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
  } while (0)
#define SC_FRAGS(T, FA, FB)                                                                   \
  do {                                                                                        \
    const unsigned short* as_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                         \
    const unsigned short* bs_ = as_ + OPER_ELEMS;                                             \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                      \
        FA[ks_][i_] = read_frag16_tr_swz(as_, wr * 64 + i_ * 32, ks_, lane);                  \
        FB[ks_][i_] = read_frag16_tr_swz(bs_, wc * 64 + i_ * 32, ks_, lane);                  \
      }                                                                                       \
  } while (0)
#define SC_CSUM(T)                                                                            \
  do {                                                                                        \
    if constexpr (decltype(with_csum)::value) {                                               \
      const unsigned short* as_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                      \
        const uint4 u_ = *(const uint4*)(as_ + csoff[i_]);                                    \
        const unsigned w_[4] = {u_.x, u_.y, u_.z, u_.w};                                      \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                    \
          cs[2 * e_] += __uint_as_float(w_[e_] << 16);                                        \
          cs[2 * e_ + 1] += __uint_as_float(w_[e_] & 0xffff0000u);                            \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define SC_MMA(FA, FB)                                                                        \
  do {                                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                      \
          acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[ks_][i_], FB[ks_][j_], acc[i_][j_], 0, 0, 0); \
  } while (0)
#define SC_PUBLISH_NEXT(T)                                                                    \
  do {                                                                                        \
    SC_WAIT_TILES(min(STAGES - 2, nk - 2 - (T)));                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                             \
    if ((T) + STAGES < nk) SC_TN_ISSUE((T) + STAGES);                                         \
  } while (0)
#define SC_PUBLISH_STEADY(T)                                                                  \
  do {                                                                                        \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                             \
    SC_TN_ISSUE((T) + STAGES);                                                                \
  } while (0)
  // (two instances of the loop, with and without the column sums: a run-time branch around LDS reads inside the steady state makes
  // the compiler wait for ALL outstanding reads where the paths merge, which serialises fragment reads and MFMAs again)
  auto main_loop = [&](auto with_csum) {
    bf16x8_t fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2];
    if (nk > 0) {
      SC_WAIT_TILES(min(STAGES - 1, nk - 1));
      __builtin_amdgcn_s_barrier();
      SC_FRAGS(0, fa0, fb0);
      SC_CSUM(0);
    }
    int t = 0;
    for (; t + STAGES + 1 < nk; t += 2) {  // steady state, branch-free
      SC_PUBLISH_STEADY(t);
      SC_FRAGS(t + 1, fa1, fb1);
      SC_CSUM(t + 1);
      SC_MMA(fa0, fb0);
      SC_PUBLISH_STEADY(t + 1);
      SC_FRAGS(t + 2, fa0, fb0);
      SC_CSUM(t + 2);
      SC_MMA(fa1, fb1);
    }
    for (; t < nk; t += 2) {
      if (t + 1 < nk) {
        SC_PUBLISH_NEXT(t);
        SC_FRAGS(t + 1, fa1, fb1);
        SC_CSUM(t + 1);
      }
      SC_MMA(fa0, fb0);
      if (t + 1 < nk) {
        if (t + 2 < nk) {
          SC_PUBLISH_NEXT(t + 1);
          SC_FRAGS(t + 2, fa0, fb0);
          SC_CSUM(t + 2);
        }
        SC_MMA(fa1, fb1);
      }
    }
  };
  if (csum) main_loop(std::true_type{});
  else main_loop(std::false_type{});
#undef SC_WAIT_TILES
#undef SC_FRAGS
#undef SC_CSUM
#undef SC_MMA
#undef SC_PUBLISH_NEXT
#undef SC_PUBLISH_STEADY
#undef SC_TN_ISSUE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves are done with the operand stages before they become staging

  if (csum) {
    float* red = (float*)dsmem;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < p.M) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += red[g * 128 + tid];
      p.colsum_part[(long long)z * p.M + m0 + tid] = v;
    }
    __syncthreads();
  }

  // raw partial tile -> slab z (f32); the reduce kernel finishes the job
  float* stage = (float*)dsmem + wid * 32 * STG_LD;
  float* slab = (float*)p.C + (long long)z * p.M * p.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * STG_LD + j * 32 + li] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < 8; ++it) {
      const int qq = lane + 64 * it;
      const int row = qq >> 4, c4 = (qq & 15) * 4;
      const int m = m0 + wr * 64 + i * 32 + row, n = n0 + wc * 64 + c4;
      if (m < p.M && n < p.N) {
        const f32x4 v = *(const f32x4*)(stage + row * STG_LD + c4);
        float* dst = slab + (long long)m * p.ldc + n;
        if (n + 4 <= p.N && p.c_vec32) *(f32x4*)dst = v * p.alpha;
        else for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e] * p.alpha;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
}

static int g_gemm_exp = 0;  // experiment switches, see sc_gemm_experiment

template <int WM, int WN, int MI, int PF, bool KTAIL = false, bool HALFLOAD = false, int SG = 4>
int launch_dma_pf(const GemmArgs& a_in, int nbatch, hipStream_t stream) {
  constexpr int TM = WM * MI * 32, TN = WN * 64;
  constexpr size_t smem_ring = (size_t)SG * (TM + TN) * BK16 * sizeof(unsigned short), smem_stg = (size_t)WM * WN * 32 * STG_LD * sizeof(float);
  size_t smem = smem_ring > smem_stg ? smem_ring : smem_stg;  // (the operand stages double as the epilogue staging)
  GemmArgs a = a_in;
  a.exp_flags = g_gemm_exp;
  dim3 grid(sc_cdiv(a.N, TN), sc_cdiv(a.M, TM), nbatch);
  // (experiment 1: a launch with at most one workgroup per CU asks for more than half the LDS, so that no two of them can share a CU)
  if ((g_gemm_exp & 1) && (long long)grid.x * grid.y * grid.z <= 256 && smem <= 80 * 1024) smem = 96 * 1024;
  auto kernel = gemm_bf16_nt_dma_kernel<WM, WN, MI, PF, KTAIL, HALFLOAD, SG>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return sc_set_error(SC_ERR_LAUNCH, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kernel, grid, dim3(64 * WM * WN), smem, stream, a);
  SC_LAUNCH_CHECK("gemm_bf16_nt_dma_kernel");
  return SC_OK;
}

// Fragment pipelining per tile configuration, from scripts/ab_gemm_pf.sh on the step's shapes (two alternating rounds, one box):
//   <2,2,2> 128 x 128, one wave per SIMD when the launch has <= 256 workgroups: PF 2   (M = 7968: N = K = 512 11.4 -> 10.4 us,
//           N = 512, K = 2048 26.9 -> 24.8 us; PF 1 in between)
//   <2,2,1> 64 x 128: PF 2 (M = 2016, N = 2048, K = 512: 10.8 -> 9.8 us);  <2,1,1> 64 x 64: PF 1 (M = 2016, N = K = 512: 6.8 -> 5.9 us)
//   <2,4,4> 256 x 256, two waves per SIMD (the partner wave already fills the read latency): PF 0 (7968 x 2048 x 512: 26.5 / 26.2 /
//           26.4 us for PF 0 / 1 / 2; 8192^3: 1182 / 1353 / 1265 us)
template <int WM, int WN, int MI, int PF>
int launch_dma(const GemmArgs& a, int nbatch, hipStream_t stream) {
  if (a.K % BK16 != 0) return launch_dma_pf<WM, WN, MI, 0, true>(a, nbatch, stream);  // partial last K tile
  return launch_dma_pf<WM, WN, MI, PF>(a, nbatch, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// f32 kernel (exact f32: each output is a k-ordered fmaf chain on the f32 matrix cores)
// ---------------------------------------------------------------------------------------------------------------
constexpr int BK32 = 16, LD32 = 17;

template <bool TR>
__device__ __forceinline__ void load_tile32(const float* base, long long ld, int rows, int K, int row0, int k0, bool vec,
                                            int tid, float4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    int r, k, nvalid;
    const float* src;
    if (!TR) {
      r = row0 + (c >> 2);
      k = k0 + (c & 3) * 4;
      nvalid = (r < rows) ? min(4, K - k) : 0;
      src = base + (long long)r * ld + k;
    } else {
      k = k0 + (c & 15);
      r = row0 + (c >> 4) * 4;
      nvalid = (k < K) ? min(4, rows - r) : 0;
      src = base + (long long)k * ld + r;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) {
      v = *(const float4*)src;
    } else if (nvalid > 0) {
      v.x = src[0];
      if (nvalid > 1) v.y = src[1];
      if (nvalid > 2) v.z = src[2];
      if (nvalid > 3) v.w = src[3];
    }
    reg[i] = v;
  }
}

template <bool TR>
__device__ __forceinline__ void store_tile32(float* lds, int tid, const float4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    const float w[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    if (!TR) {
      float* d = lds + (c >> 2) * LD32 + (c & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = w[j];
    } else {
      const int k = c & 15, r = (c >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(r + j) * LD32 + k] = w[j];
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BM * LD32];
  float* As = smem;
  float* Bs = smem + 2 * BM * LD32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int h = lane >> 5, li = lane & 31;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int z = blockIdx.z, z1 = z / p.nb2, z2 = z - z1 * p.nb2;
  const float* A = (const float*)p.A + z1 * p.sA1 + z2 * p.sA2;
  const float* B = (const float*)p.B + z1 * p.sB1 + z2 * p.sB2;
  const long long zoffC = z1 * p.sC1 + z2 * p.sC2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2], rb[2];
  const int nk = (p.K + BK32 - 1) / BK32;
  load_tile32<TA>(A, p.lda, p.M, p.K, m0, 0, p.a_vec, tid, ra);
  load_tile32<TB>(B, p.ldb, p.N, p.K, n0, 0, p.b_vec, tid, rb);
  store_tile32<TA>(As, tid, ra);
  store_tile32<TB>(Bs, tid, rb);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      load_tile32<TA>(A, p.lda, p.M, p.K, m0, (t + 1) * BK32, p.a_vec, tid, ra);
      load_tile32<TB>(B, p.ldb, p.N, p.K, n0, (t + 1) * BK32, p.b_vec, tid, rb);
    }
    const float* as = As + cur * BM * LD32;
    const float* bs = Bs + cur * BM * LD32;
#pragma unroll
    for (int ks = 0; ks < BK32 / 2; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = as[(wr * 64 + i * 32 + li) * LD32 + ks * 2 + h];
        b[i] = bs[(wc * 64 + i * 32 + li) * LD32 + ks * 2 + h];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) {
      store_tile32<TA>(As + (cur ^ 1) * BM * LD32, tid, ra);
      store_tile32<TB>(Bs + (cur ^ 1) * BM * LD32, tid, rb);
    }
    __syncthreads();
  }

  epilogue_wave<float, 2>(p, smem + wid * 32 * STG_LD, acc, zoffC, z, m0 + wr * 64, n0 + wc * 64, lane);
}

// out[i] (+)= sum_s slab[s][i]   (deterministic split-K second stage); optionally the same for a second, short segment
// (the per-slice column sums: out2[j] (+)= sum_s part2[s][j], j < n2)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int S, long long n, float* out, int accumulate,
                                                            const float* part2, int n2, float* out2) {
  if (out2 != nullptr) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n2; j += gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < S; ++k) s += part2[(long long)k * n2 + j];
      out2[j] = accumulate ? out2[j] + s : s;
    }
  }
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      // eight slice loads in flight per thread (the slabs sit in L2 / Infinity Cache: the kernel is latency-bound, and a loop
      // with a run-time trip count would keep one load in flight); the summation order stays k = 0, 1, 2, ...
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < S; k0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (k0 + u < S) ? *(const f32x4*)(slabs + (long long)(k0 + u) * n + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      if (accumulate) s += *(const f32x4*)(out + i);
      *(f32x4*)(out + i) = s;
    } else {
      for (long long e = i; e < n; ++e) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += slabs[(long long)k * n + e];
        out[e] = accumulate ? out[e] + s : s;
      }
    }
  }
}

template <typename K>
int launch(K kernel, const GemmArgs& a, int nbatch, hipStream_t stream, const char* name) {
  dim3 grid(sc_cdiv(a.N, BN), sc_cdiv(a.M, BM), nbatch);
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK(name);
  return SC_OK;
}

}  // namespace

// Which kernel the last sc_gemm / sc_gemm_splitk call of this thread dispatched (bench.py attributes its per-launch HIP-event
// times to kernels with it, so that its roofline line names a row of the rocprofv3 kernel statistics).
static thread_local int g_last_gemm_kernel = 0;
static const char* const kGemmKernelNames[] = {
    "none",
    "gemm_bf16_nt_dma_kernel<2, 2, 2, 2, false>",
    "gemm_bf16_nt_dma_kernel<2, 4, 4, 0, false>",
    "gemm_bf16_kernel<false, false>",
    "gemm_bf16_kernel<false, true>",
    "gemm_bf16_kernel<true, false>",
    "gemm_bf16_kernel<true, true>",
    "gemm_f32_kernel",
    "gemm_bf16_kernel<true, true> (split-K) + splitk_reduce_kernel",
    "gemm_bf16_tn_dma_splitk_kernel + splitk_reduce_kernel",
    "gemm_bf16_kernel (split-K, other layout) + splitk_reduce_kernel",
    "gemm_bf16_nt_dma_kernel<2, 2, 1, 2, false>",
    "gemm_bf16_nt_dma_kernel<2, 1, 1, 1, false>",
    "wgrad_grouped_kernel<2, 4, 4>",
    "wgrad_grouped_kernel<2, 2, 2>",
    "gemm_bf16_nt_dma_kernel<4, 2, 1, 2, false>",
    "gemm_bf16_nt_dma_kernel<4, 2, 1, 0, false>",
    "gemm_bf16_nt_dma_kernel<4, 2, 1, 1, false>",
};
SC_API int sc_gemm_last_kernel(void) { return g_last_gemm_kernel; }
// Experiment switches for A/B runs inside ONE process (scripts/ab_gemm_exp.py); not part of the ABI.  bit 0: launches of at most 256
// workgroups take a CU each; bit 1: the 8-wave 128 x 128 kernel lets its older four waves issue every LDS-DMA; bit 2: static priority 1 for
// the younger four waves.
SC_API void sc_gemm_experiment(int flags) { g_gemm_exp = flags; }
#ifdef SC_GEMM_TIMING
SC_API int sc_gemm_debug_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sc_gemm_probe), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif
SC_API const char* sc_gemm_kernel_name(int id) {
  return (id >= 0 && id < (int)(sizeof(kGemmKernelNames) / sizeof(kGemmKernelNames[0]))) ? kGemmKernelNames[id] : "unknown";
}

// dtype: 0 = f32, 1 = bf16 (A, B, aux, pre_out and -- unless out_f32 -- C share it).
SC_API int sc_gemm(int dtype, int ta, int tb, const void* A, const void* B, void* C, int M, int N, int K, long long lda,
                   long long ldb, long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1,
                   long long sB2, long long sC1, long long sC2, float alpha, const float* bias, int act, void* pre_out,
                   const void* aux, long long ldaux, int aux_mode, float drop_p, const unsigned long long* seed,
                   unsigned long long offset, int out_f32, int accumulate, hipStream_t stream) {
  SC_CHECK_ARG(A && B && C, "sc_gemm: null operand");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_gemm: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(M >= 1 && N >= 1 && K >= 1 && nb1 >= 1 && nb2 >= 1, "sc_gemm: bad sizes M=%d N=%d K=%d nb=%dx%d", M, N, K, nb1, nb2);
  SC_CHECK_ARG((long long)nb1 * nb2 <= 65535, "sc_gemm: batch %lld too large", (long long)nb1 * nb2);
  // (A not transposed: lda < K is allowed -- overlapping rows, the im2col-free 'same' Conv1d of functional._ConvLinear, see sc_pad_rows)
  SC_CHECK_ARG(lda >= (ta ? M : 1) && ldb >= (tb ? N : K) && ldc >= N, "sc_gemm: leading dimension too small");
  SC_CHECK_ARG(aux_mode >= AUX_NONE && aux_mode <= AUX_MUL && (aux_mode == AUX_NONE || aux), "sc_gemm: aux_mode %d (0..3) without aux or out of range", aux_mode);
  SC_CHECK_ARG((act & 0xFF) <= ACT_GELU && (act & ~(0xFF | ACT_PRE_GRAD)) == 0, "sc_gemm: bad act code %d", act);
  SC_CHECK_ARG(!(act & ACT_PRE_GRAD) || pre_out, "sc_gemm: the pre-gradient flag needs pre_out");
  SC_CHECK_ARG(!accumulate || out_f32, "sc_gemm: accumulate needs an f32 output");
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_gemm: dropout p=%f", drop_p);
  SC_CHECK_ARG(drop_p == 0.f || seed, "sc_gemm: dropout needs a seed cell");
  GemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.nb2 = nb2; a.sA1 = sA1; a.sA2 = sA2; a.sB1 = sB1; a.sB2 = sB2; a.sC1 = sC1; a.sC2 = sC2;
  a.alpha = alpha; a.bias = bias; a.act = act & 0xFF; a.pre_grad = (act & ACT_PRE_GRAD) ? 1 : 0; a.pre_out = pre_out; a.aux = aux; a.ldaux = ldaux; a.aux_mode = aux_mode;
  a.drop_p = drop_p; a.drop_keep = 1.f / (1.f - drop_p); a.seed = seed; a.offset = offset; a.out_f32 = out_f32; a.accumulate = accumulate;
  const long long esz = dtype == 0 ? 4 : 2;
  auto vec_ok = [&](const void* ptr, long long ld, long long s1, long long s2) {
    return ((uintptr_t)ptr % 16 == 0) && ((ld * esz) % 16 == 0) && ((s1 * esz) % 16 == 0) && ((s2 * esz) % 16 == 0);
  };
  a.a_vec = vec_ok(A, lda, sA1, sA2);
  a.b_vec = vec_ok(B, ldb, sB1, sB2);
  {
    const long long cesz = out_f32 ? 4 : esz;
    auto ok = [&](const void* ptr, long long ld, long long e) {
      return ptr == nullptr || (((uintptr_t)ptr % 16 == 0) && ((ld * e) % 16 == 0) && ((sC1 * e) % 16 == 0) && ((sC2 * e) % 16 == 0));
    };
    a.c_vec = ok(C, ldc, cesz) && ok(aux, ldaux, esz) && ok(pre_out, ldc, esz);
    a.c_vec32 = ok(C, ldc, 4) && ok(aux, ldaux, esz);
  }
  const int nb = nb1 * nb2;
  if (dtype == 1) {
    if (!ta && !tb && a.a_vec && a.b_vec && (K % 8 == 0) && K >= 2 * BK16) {
      // 256 x 256 tiles when they still give every CU work, else 128 x 128
      const long long big = (long long)sc_cdiv(M, 256) * sc_cdiv(N, 256) * nb;
      // Few 128 x 128 tiles (the decoder-sized products, M ~ 2000: 64 .. 192 workgroups on 256 CUs): 64 x 128 tiles double the
      // number of waves.  Measured on M = 2016: N = K = 512 18.0 -> 13.9 us, K = 2048 33.4 -> 27.8 us, K = 1536 25.2 -> 21.4 us
      // (DESIGN.md section 6).  At 252 tiles (M = 7968, N = 512) they do not pay: K = 1536 26.4 -> 30.9 us, K = 2048 44.8 -> 45.5 us,
      // K = 512 21.1 -> 20.3 us -- hence the threshold of 192.  [Tried and rejected: 128 x 128 instead of 256 x 256 tiles for the launches with a VALU-heavy
      // epilogue (FFN1 bias+GELU+dropout, act' dgrad) so that a second workgroup per CU covers the epilogue: 56.4 -> 61.7 us.]
      const long long t128 = (long long)sc_cdiv(M, 128) * sc_cdiv(N, 128) * nb;
      // (N <= 128, e.g. the 64-channel prenet convolution: a 256-wide tile would be three quarters empty)
      static const long long bigmin = [] { const char* e = getenv("SC_GEMM_BIG_MIN"); return e ? atoll(e) : 700LL; }();  // (A/B knob; 7968 x 1536 x 512, 192 tiles of 256 x 256: 27.4 us, as 756 tiles of 128 x 128: 26.3 us.
      // Round 3, measured inside the train step (three alternating runs per setting, one box): with the 256-tile (7968 x 2048) and 640-tile (7968 x 5000) products
      // on the 8-wave 128 x 128 kernel as well and that kernel unpipelined (SC_GEMM_W8=1) the step is 0.8 % faster than with 200 / pipelined (8.17 vs 8.24 ms),
      // neutral on a second box; the conv products of the [512,512] prenet (1250 tiles) stay on 256 x 256, which they prefer.)
      // (a batched product whose M is far from a multiple of 256 -- the per-utterance products of the im2col-free Conv1d: T = 640 frames
      //  would fill 3 row tiles of 256 to 83 %, T = 128 tokens one to 50 % -- takes 128-row tiles when those waste less)
      const bool ragged256 = (long long)sc_cdiv(M, 256) * 256 * 100 > (long long)sc_cdiv(M, 128) * 128 * 108;
      if (big >= bigmin && N > 128 && !ragged256) { g_last_gemm_kernel = 2; return launch_dma<2, 4, 4, 0>(a, nb, stream); }
      // Very few tiles (the decoder's M = 2016, N = 512 products: 64 tiles of 128 x 128, 128 of 64 x 128): 64 x 64 tiles, two waves each,
      // put a workgroup on every CU.  Step-level A/B (40 steps, alternating): 12.35 / 12.36 / 12.55 ms with them vs 12.60 / 12.61 / 12.69
      // without; thresholds of 80, 128 and 192 tiles measure the same (12.53 .. 12.64 on another box), 256 hurts (12.99).
      // SC_GEMM_T64=<max tiles> overrides (0 switches the variant off).
      static const long long t64max = [] { const char* e = getenv("SC_GEMM_T64"); return e ? atoll(e) : 80LL; }();
      if (t128 <= t64max) { g_last_gemm_kernel = 12; return launch_dma<2, 1, 1, 1>(a, nb, stream); }
      static const long long t64x128max = [] { const char* e = getenv("SC_GEMM_T64X128"); return e ? atoll(e) : 192LL; }();
      if (t128 <= t64x128max) { g_last_gemm_kernel = 11; return launch_dma<2, 2, 1, 2>(a, nb, stream); }
      // At most one 128 x 128 tile per CU (M = 7968, N = 512: 252 tiles): the tile runs on 8 waves of 32 x 64 -- two per SIMD, so one wave's
      // fragment reads and address arithmetic overlap the other's matrix-core work -- instead of 4 waves of 64 x 64.  Kernel A/B (one box):
      // K = 512 10.5 -> 9.9 us, K = 1024 15.2 -> 14.4, K = 2048 25.5 -> 25.2, K = 96 7.1 -> 6.4, K = 2048 + bias + residual 28.1 -> 26.0;
      // step A/B (alternating, 60 steps): 9.17 / 9.17 ms without, 8.94 / 9.01 ms with.  SC_GEMM_W8=0 switches it off, SC_GEMM_W8_MAX=<tiles>
      // moves the threshold.
      // Round 3: default 1, the unpipelined loop (was 3, fragments pipelined by whole tiles).  Launched back to back the pipelined form is 3-6 % faster; inside
      // the step -- every launch behind a different kernel, instructions and first tiles cold -- the short loop wins (see SC_GEMM_BIG_MIN above and DESIGN 6r3).
      static const int w8 = [] { const char* e = getenv("SC_GEMM_W8"); return e ? atoi(e) : 1; }();
      static const long long w8max = [] { const char* e = getenv("SC_GEMM_W8_MAX"); return e ? atoll(e) : (1LL << 60); }();
      if (w8 && t128 <= w8max && a.K % BK16 == 0) {
        g_last_gemm_kernel = w8 == 1 ? 16 : w8 == 2 ? 17 : 15;
        if (w8 == 1) return launch_dma_pf<4, 2, 1, 0>(a, nb, stream);
        if (w8 == 2) return launch_dma_pf<4, 2, 1, 1>(a, nb, stream);
        if (g_gemm_exp & 64) return launch_dma_pf<4, 2, 1, 2, false, false, 8>(a, nb, stream);  // (experiment 64: eight stages in the ring)
        if (g_gemm_exp & 2) return launch_dma_pf<4, 2, 1, 2, false, true>(a, nb, stream);  // (experiment 2: the older half stages)
        return launch_dma_pf<4, 2, 1, 2>(a, nb, stream);
      }
      g_last_gemm_kernel = 1;
      return launch_dma<2, 2, 2, 2>(a, nb, stream);
    }
    g_last_gemm_kernel = 3 + (ta ? 2 : 0) + (tb ? 1 : 0);
    if (!ta && !tb) return launch(gemm_bf16_kernel<false, false>, a, nb, stream, "gemm_bf16<NT>");
    if (!ta && tb) return launch(gemm_bf16_kernel<false, true>, a, nb, stream, "gemm_bf16<NN>");
    if (ta && !tb) return launch(gemm_bf16_kernel<true, false>, a, nb, stream, "gemm_bf16<TT>");
    return launch(gemm_bf16_kernel<true, true>, a, nb, stream, "gemm_bf16<TN>");
  }
  g_last_gemm_kernel = 7;
  if (!ta && !tb) return launch(gemm_f32_kernel<false, false>, a, nb, stream, "gemm_f32<NT>");
  if (!ta && tb) return launch(gemm_f32_kernel<false, true>, a, nb, stream, "gemm_f32<NN>");
  if (ta && !tb) return launch(gemm_f32_kernel<true, false>, a, nb, stream, "gemm_f32<TT>");
  return launch(gemm_f32_kernel<true, true>, a, nb, stream, "gemm_f32<TN>");
}

// Split-K product for wgrad-shaped problems (small M x N output, long contraction): C (f32, M x N, ldc == N)
// (+)= alpha * opA opB^T with the contraction cut into `splitk` slices whose partial tiles go to `workspace`
// (splitk * M * N floats) and are summed by a second, deterministic kernel.  bf16 operands only.
// (splitk * M * N slab floats, then splitk * M floats for the optional column sums)
SC_API long long sc_gemm_splitk_workspace(int M, int N, int splitk) {
  return ((long long)splitk * M * N + (long long)splitk * M) * (long long)sizeof(float);
}

SC_API int sc_gemm_splitk(int ta, int tb, const void* A, const void* B, float* C, int M, int N, int K, long long lda, long long ldb,
                          float alpha, int splitk, int accumulate, float* colsum, void* workspace, long long ws_bytes,
                          hipStream_t stream) {
  SC_CHECK_ARG(colsum == nullptr || ta, "sc_gemm_splitk: colsum needs a transposed A (its columns are the rows of C)");
  SC_CHECK_ARG(A && B && C && workspace, "sc_gemm_splitk: null pointer");
  SC_CHECK_ARG(M >= 1 && N >= 1 && K >= 1 && splitk >= 2 && splitk <= 64, "sc_gemm_splitk: bad sizes");
  SC_CHECK_ARG(lda >= (ta ? M : K) && ldb >= (tb ? N : K), "sc_gemm_splitk: leading dimension too small");
  SC_CHECK_ARG(ws_bytes >= sc_gemm_splitk_workspace(M, N, splitk), "sc_gemm_splitk: workspace too small");
  GemmArgs a{};
  a.A = A; a.B = B; a.C = workspace; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = N;
  a.nb2 = splitk; a.alpha = alpha; a.out_f32 = 1;
  a.a_vec = ((uintptr_t)A % 16 == 0) && ((lda * 2) % 16 == 0);
  a.b_vec = ((uintptr_t)B % 16 == 0) && ((ldb * 2) % 16 == 0);
  a.c_vec32 = ((uintptr_t)workspace % 16 == 0) && (N % 4 == 0);
  a.c_vec = a.c_vec32;
  a.splitk = splitk;
  a.ksplit_len = ((K + splitk - 1) / splitk + BK16 - 1) / BK16 * BK16;
  const int slices = (K + a.ksplit_len - 1) / a.ksplit_len;
  a.splitk = slices > 1 ? slices : 2;  // keep the split-K code path even if rounding left a single slice
  float* part2 = (float*)workspace + (long long)splitk * M * N;
  a.colsum_part = colsum ? part2 : nullptr;
  int rc;
  const bool tn_dma = ta && tb && a.a_vec && a.b_vec && K % BK16 == 0 && M % 8 == 0 && N % 8 == 0 && M >= 8 && N >= 8 &&
                      getenv("SC_GEMM_TN_DMA") != nullptr;  // opt-in: measured 1.3 % slower per train step than the register-staged kernel
  g_last_gemm_kernel = tn_dma ? 9 : (ta && tb ? 8 : 10);
  if (tn_dma) {
    constexpr size_t smem = (size_t)4 * 2 * BM * BK16 * sizeof(unsigned short);  // four 16 KB stages
    static_assert(smem >= (size_t)16 * 128 * sizeof(float) && smem >= (size_t)4 * 32 * STG_LD * sizeof(float), "stages must cover the staging tiles");
    dim3 grid(sc_cdiv(N, BN), sc_cdiv(M, BM), slices);
    hipLaunchKernelGGL(gemm_bf16_tn_dma_splitk_kernel, grid, dim3(256), smem, stream, a);
    hipError_t e = hipGetLastError();
    rc = e == hipSuccess ? SC_OK : sc_set_error(SC_ERR_LAUNCH, "gemm_bf16_tn_dma_splitk_kernel: %s", hipGetErrorString(e));
  } else
  if (!ta && !tb) rc = launch(gemm_bf16_kernel<false, false>, a, slices, stream, "gemm_bf16<NT> split-K");
  else if (!ta && tb) rc = launch(gemm_bf16_kernel<false, true>, a, slices, stream, "gemm_bf16<NN> split-K");
  else if (ta && !tb) rc = launch(gemm_bf16_kernel<true, false>, a, slices, stream, "gemm_bf16<TT> split-K");
  else rc = launch(gemm_bf16_kernel<true, true>, a, slices, stream, "gemm_bf16<TN> split-K");
  if (rc != SC_OK) return rc;
  const long long n = (long long)M * N;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)workspace, slices, n, C, accumulate,
                     (const float*)part2, M, colsum);
  SC_LAUNCH_CHECK("splitk_reduce_kernel");
  return SC_OK;
}
