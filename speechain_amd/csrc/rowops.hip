// Row-wise kernels of the transformer path for gfx950 (one 64-lane wavefront per row, shuffle reductions, fp32 math):
//   LayerNorm forward/backward   -- nn.LayerNorm(d, eps=1e-6) at speechain/module/transformer/encoder.py:97-98,115-141,
//                                   decoder.py:90-92,122-173 and the final LN encoder.py:291-292,346-347
//   masked softmax forward/backward -- masked_fill(-inf) + Softmax + Dropout at attention.py:88-95
//   label-smoothed cross entropy -- log_softmax + gather + smoothing at speechain/criterion/cross_entropy.py:132-146
//   column sums                  -- bias gradients of every Linear
// Activations are f32 or bf16 (template T); statistics, gammas, losses and reductions are always f32.
#include "sc_common.h"
#include <algorithm>

namespace {

template <typename T> struct Row;
template <> struct Row<float> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct Row<unsigned short> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return sc_bf16_to_f32(((const unsigned short*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((unsigned short*)p)[i] = sc_f32_to_bf16(v); }
};

// ---- LayerNorm ---------------------------------------------------------------------------------------------------
constexpr int LN_MAX = 32;  // elements per lane kept in registers: D <= 64 * 32 = 2048

template <typename T, int NI>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* x, const float* gamma, const float* beta, void* y,
                                                            float* mean, float* rstd, long long R, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const long long base = row * D;
  float v[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    v[i] = (k < D) ? Row<T>::ld(x, base + k) : 0.f;
    s += v[i];
  }
  const float mu = sc_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    const float d = (k < D) ? v[i] - mu : 0.f;
    q += d * d;
  }
  const float rs = rsqrtf(sc_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    if (k < D) Row<T>::st(y, base + k, (v[i] - mu) * rs * gamma[k] + beta[k]);
  }
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
}

// dx for a strip of rows; per-workgroup partial dgamma / dbeta go to part[(2*blockIdx.x + {0,1}) * D + k]
template <typename T, int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* dy, const void* x, const float* mean, const float* rstd,
                                                            const float* gamma, void* dx, float* part, long long R, int D,
                                                            int rows_per_block, const void* dres) {
  extern __shared__ float red[];  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float dg[NI], db[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) dg[i] = db[i] = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(R, r0 + rows_per_block);
  for (long long row = r0 + wid; row < r1; row += 4) {
    const long long base = row * D;
    const float mu = mean[row], rs = rstd[row];
    float g[NI], xh[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = lane + 64 * i;
      if (k < D) {
        const float d = Row<T>::ld(dy, base + k);
        xh[i] = (Row<T>::ld(x, base + k) - mu) * rs;
        g[i] = d * gamma[k];
        dg[i] += d * xh[i];
        db[i] += d;
        s1 += g[i];
        s2 += g[i] * xh[i];
      } else {
        g[i] = xh[i] = 0.f;
      }
    }
    s1 = sc_wave_sum(s1) / (float)D;
    s2 = sc_wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = lane + 64 * i;
      if (k < D) Row<T>::st(dx, base + k, rs * (g[i] - s1 - xh[i] * s2) + (dres ? Row<T>::ld(dres, base + k) : 0.f));
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    if (k < D) {
      red[(wid * 2 + 0) * D + k] = dg[i];
      red[(wid * 2 + 1) * D + k] = db[i];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < D; k += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * 2 + 0) * D + k];
      b += red[(w * 2 + 1) * D + k];
    }
    part[((long long)blockIdx.x * 2 + 0) * D + k] = a;
    part[((long long)blockIdx.x * 2 + 1) * D + k] = b;
  }
}

// ---- bf16 LayerNorm with 16-byte accesses: lane l owns the 8 consecutive elements 8 (l + 64 c) .. + 7 of chunk c < NV ----
// (the element-per-lane kernels above issue one 2-byte load per lane per instruction: instruction-bound, 1.5 TB/s).
// Needs D % 8 == 0, 16-byte aligned rows and D <= 512 NV.
__device__ __forceinline__ void ln_unpack8(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(w[e] << 16);
    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 ln_pack8(const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = (unsigned)sc_f32_to_bf16(f[2 * e]) | ((unsigned)sc_f32_to_bf16(f[2 * e + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void ln_load8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const unsigned short* x, const float* gamma, const float* beta,
                                                                unsigned short* y, float* mean, float* rstd, long long R, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const long long base = row * D;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k = 8 * (lane + 64 * c);
    if (k < D) {
      ln_unpack8(*reinterpret_cast<const uint4*>(x + base + k), v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mu = sc_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c)
    if (8 * (lane + 64 * c) < D)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[c][e] - mu;
        q += d * d;
      }
  const float rs = rsqrtf(sc_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k = 8 * (lane + 64 * c);
    if (k < D) {
      float g[8], b[8], o[8];
      ln_load8f(gamma + k, g);
      ln_load8f(beta + k, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mu) * rs * g[e] + b[e];
      *reinterpret_cast<uint4*>(y + base + k) = ln_pack8(o);
    }
  }
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
}

// dx for a strip of rows; per-workgroup partial dgamma / dbeta go to part[(2*blockIdx.x + {0,1}) * D + k]
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const unsigned short* dy, const unsigned short* x, const float* mean,
                                                                const float* rstd, const float* gamma, unsigned short* dx, float* part,
                                                                long long R, int D, int rows_per_block, const unsigned short* dres,
                                                                unsigned short* dz, float drop_p, const unsigned long long* seed_ptr,
                                                                unsigned long long offset) {
  extern __shared__ float red[];  // [4 waves][2][D]
  // optional second output: dz = dx * keepmask / (1 - p) of the dropout site whose output this LayerNorm normalised (element index =
  // offset + row * D + column, the convention of sc_dropout): what that site's backward pass would otherwise make from dx in its own launch
  const unsigned long long seed = dz ? *seed_ptr : 0ull;
  const unsigned thr = sc_drop_threshold(drop_p);
  const float keep = 1.f / (1.f - drop_p);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float gm[NV][8], dg[NV][8], db[NV][8];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k = 8 * (lane + 64 * c);
    if (k < D) ln_load8f(gamma + k, gm[c]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[c][e] = db[c][e] = 0.f;
      if (k >= D) gm[c][e] = 0.f;
    }
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(R, r0 + rows_per_block);
  // A wave's rows are a serial chain of (load, two reductions, store): the next row's pieces are requested before the current row is
  // reduced, so that a row costs its arithmetic instead of a memory round trip.
  uint4 nd[NV], nx[NV], nr[NV];
  float nmu = 0.f, nrs = 0.f;
#define LN_BWD_REQUEST(ROW)                                                             \
  do {                                                                                  \
    const long long nb_ = (ROW) * D;                                                    \
    nmu = mean[(ROW)];                                                                  \
    nrs = rstd[(ROW)];                                                                  \
    _Pragma("unroll") for (int c = 0; c < NV; ++c) {                                    \
      const int k = 8 * (lane + 64 * c);                                                \
      if (k < D) {                                                                      \
        nd[c] = *reinterpret_cast<const uint4*>(dy + nb_ + k);                          \
        nx[c] = *reinterpret_cast<const uint4*>(x + nb_ + k);                           \
        if (dres) nr[c] = *reinterpret_cast<const uint4*>(dres + nb_ + k);              \
      }                                                                                 \
    }                                                                                   \
  } while (0)
  if (r0 + wid < r1) LN_BWD_REQUEST(r0 + wid);
  for (long long row = r0 + wid; row < r1; row += 4) {
    const long long base = row * D;
    const float mu = nmu, rs = nrs;
    uint4 cd[NV], cx[NV], cr[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      cd[c] = nd[c];
      cx[c] = nx[c];
      cr[c] = nr[c];
    }
    if (row + 4 < r1) LN_BWD_REQUEST(row + 4);
    float g[NV][8], xh[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int k = 8 * (lane + 64 * c);
      if (k < D) {
        float d[8];
        ln_unpack8(cd[c], d);
        ln_unpack8(cx[c], xh[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[c][e] = (xh[c][e] - mu) * rs;
          g[c][e] = d[e] * gm[c][e];
          dg[c][e] += d[e] * xh[c][e];
          db[c][e] += d[e];
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[c][e] = xh[c][e] = 0.f;
      }
    }
    s1 = sc_wave_sum(s1) / (float)D;
    s2 = sc_wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int k = 8 * (lane + 64 * c);
      if (k < D) {
        float o[8], rv[8];
        if (dres) ln_unpack8(cr[c], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (g[c][e] - s1 - xh[c][e] * s2) + (dres ? rv[e] : 0.f);
        const uint4 packed = ln_pack8(o);
        *reinterpret_cast<uint4*>(dx + base + k) = packed;
        if (dz) {  // from the ROUNDED dx, so that the result equals sc_dropout applied to the stored tensor bit for bit
          float q[8];
          ln_unpack8(packed, q);
          const unsigned long long idx0 = offset + (unsigned long long)(base + k);
          unsigned hsh = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned long long idx = idx0 + e;
            if (e == 0 || (idx & 1) == 0) hsh = sc_rand_pair(seed, idx >> 1);
            const bool dropped = ((hsh >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr;
            q[e] = dropped ? 0.f : q[e] * keep;
          }
          *reinterpret_cast<uint4*>(dz + base + k) = ln_pack8(q);
        }
      }
    }
  }
#undef LN_BWD_REQUEST
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k = 8 * (lane + 64 * c);
    if (k < D)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wid * 2 + 0) * D + k + e] = dg[c][e];
        red[(wid * 2 + 1) * D + k + e] = db[c][e];
      }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < D; k += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * 2 + 0) * D + k];
      b += red[(w * 2 + 1) * D + k];
    }
    part[((long long)blockIdx.x * 2 + 0) * D + k] = a;
    part[((long long)blockIdx.x * 2 + 1) * D + k] = b;
  }
}

// out[j][k] = sum_p part[(p * nout + j) * D + k]   (deterministic second stage of column reductions).
// grid (ceil(D/16), nout); 1024 threads = 16 columns x 64 partial-lanes, combined through LDS in a fixed order.  The
// stage is a few MB read by few workgroups, i.e. latency-bound: every lane keeps up to 16 loads in flight.
__global__ __launch_bounds__(1024) void reduce_parts_kernel(const float* part, int P, int nout, int D, float* out0, float* out1) {
  __shared__ float red[64][17];
  const int c = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + c;
  const int j = blockIdx.y;
  float s = 0.f;
  if (k < D) {
    for (int p0 = pl; p0 < P; p0 += 64 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int pp = p0 + 64 * u;
        v[u] = pp < P ? part[((long long)pp * nout + j) * D + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && k < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) t += red[i][c];
    (j == 0 ? out0 : out1)[k] = t;
  }
}

// reduce_parts_kernel for a batch of independent (part, P, D, out0, out1) problems with nout = 2: blockIdx.z selects the problem
constexpr int LN_BATCH = 64;
struct LnFinishBatch {
  const float* part[LN_BATCH];
  float* out0[LN_BATCH];
  float* out1[LN_BATCH];
  int P[LN_BATCH];
  int D[LN_BATCH];
};

__global__ __launch_bounds__(1024) void reduce_parts_batched_kernel(LnFinishBatch b) {
  __shared__ float red[64][17];
  const int z = blockIdx.z;
  const float* part = b.part[z];
  const int P = b.P[z], D = b.D[z];
  const int c = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + c;
  const int j = blockIdx.y;
  float s = 0.f;
  if (k < D) {
    for (int p0 = pl; p0 < P; p0 += 64 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int pp = p0 + 64 * u;
        v[u] = pp < P ? part[((long long)pp * 2 + j) * D + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && k < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) t += red[i][c];
    (j == 0 ? b.out0[z] : b.out1[z])[k] = t;
  }
}

// column sums of an (R, N) matrix with row stride ld: part[blockIdx.y][n]; 256 threads = 64 columns x 4 row-lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const void* x, long long R, int N, long long ld, int rows_per_block, float* part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (long long r = r0 + rl; r < r1; r += 4) s += Row<T>::ld(x, r * ld + n);
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && n < N) part[(long long)blockIdx.y * N + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---- masked softmax ------------------------------------------------------------------------------------------------
constexpr int SM_MAX = 16;  // keys per lane in registers: Tk <= 1024

struct SoftmaxArgs {
  const void* S;   // (Z, Tq, ld) scores (already scaled)
  void* P;         // (Z, Tq, ld) probabilities (may alias S)
  void* Pd;        // (Z, Tq, ld) dropped probabilities or null
  int Z, H, Tq, Tk;
  long long ld;
  const unsigned char* mask;  // bool (B, Tq|1, Tk) or null; batch = z / H
  long long mask_b, mask_q;   // strides (mask_q = 0 broadcasts over queries)
  int causal;
  float drop_p;
  const unsigned long long* seed;
  unsigned long long offset;
};

template <typename T, int NI>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(SoftmaxArgs p) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)p.Z * p.Tq) return;
  const int z = (int)(row / p.Tq), q = (int)(row - (long long)z * p.Tq), b = z / p.H;
  const long long base = row * p.ld;
  const unsigned char* mrow = p.mask ? p.mask + b * p.mask_b + q * p.mask_q : nullptr;
  float v[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    bool ok = k < p.Tk;
    if (ok && mrow) ok = mrow[k] != 0;
    if (ok && p.causal) ok = k <= q;
    v[i] = ok ? Row<T>::ld(p.S, base + k) : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = sc_wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    v[i] = (v[i] == -INFINITY) ? 0.f : expf(v[i] - mx);
    s += v[i];
  }
  s = sc_wave_sum(s);
  const float inv = s > 0.f ? 1.f / s : 0.f;  // a fully masked row yields zeros (torch would give NaN)
  const float keep = 1.f / (1.f - p.drop_p);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    if (k < p.ld) {  // also zero the padding columns [Tk, ld) so they are inert as a GEMM operand
      const float pr = (k < p.Tk) ? v[i] * inv : 0.f;
      Row<T>::st(p.P, base + k, pr);
      if (p.Pd) {
        float d = pr;
        if (p.drop_p > 0.f) d = sc_dropped(*p.seed, p.offset + base + k, sc_drop_threshold(p.drop_p)) ? 0.f : pr * keep;
        Row<T>::st(p.Pd, base + k, d);
      }
    }
  }
}

// The same with 16-byte accesses (bf16, ld % 8 == 0, 16-byte aligned tensors, ld <= 512 NV): lane l owns the 8 consecutive keys
// 8 (l + 64 c) .. + 7 of chunk c.  The element-per-lane kernel moves 128 bytes per wave-instruction: 70 us = 2.2 TB/s on the FastSpeech2
// decoder's (32 x 2 x 640, 640) score tensors; same arithmetic, same dropout indices.
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(SoftmaxArgs p) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)p.Z * p.Tq) return;
  const int z = (int)(row / p.Tq), q = (int)(row - (long long)z * p.Tq), b = z / p.H;
  const long long base = row * p.ld;
  const unsigned char* mrow = p.mask ? p.mask + b * p.mask_b + q * p.mask_q : nullptr;
  const unsigned short* S = (const unsigned short*)p.S;
  float v[NV][8];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k0 = 8 * (lane + 64 * c);
    if (k0 < p.ld) {
      ln_unpack8(*reinterpret_cast<const uint4*>(S + base + k0), v[c]);
      unsigned long long mb = ~0ull;  // the 8 mask bytes of this chunk (a byte != 0 keeps the key)
      if (mrow) {
        if ((((uintptr_t)(mrow + k0)) & 7) == 0 && k0 + 8 <= p.Tk) {
          mb = *reinterpret_cast<const unsigned long long*>(mrow + k0);
        } else {
          mb = 0ull;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e < p.Tk) mb |= (unsigned long long)mrow[k0 + e] << (8 * e);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        bool ok = k < p.Tk && ((mb >> (8 * e)) & 0xFFull) != 0;
        if (p.causal) ok = ok && k <= q;
        v[c][e] = ok ? v[c][e] : -INFINITY;
        mx = fmaxf(mx, v[c][e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = -INFINITY;
    }
  }
  mx = sc_wave_max(mx);
  float s = 0.f;
  const float mxl = mx * 1.44269504088896341f;
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // exp(v - mx) as one v_exp_f32 on (v log2 e - mx log2 e); a masked key (-inf) gives exactly 0, a fully masked row (mx = -inf) is
      // caught below
      v[c][e] = (v[c][e] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(fmaf(v[c][e], 1.44269504088896341f, -mxl));
      s += v[c][e];
    }
  s = sc_wave_sum(s);
  const float inv = s > 0.f ? 1.f / s : 0.f;  // a fully masked row yields zeros (torch would give NaN)
  const float keep = 1.f / (1.f - p.drop_p);
  const unsigned long long seed = (p.Pd && p.drop_p > 0.f) ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k0 = 8 * (lane + 64 * c);
    if (k0 >= p.ld) continue;  // (columns [Tk, ld) are written as zeros: inert as a GEMM operand)
    float pr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pr[e] = v[c][e] * inv;
    const uint4 packed = ln_pack8(pr);
    *reinterpret_cast<uint4*>((unsigned short*)p.P + base + k0) = packed;
    if (p.Pd) {
      float d[8];
      const unsigned long long idx0 = p.offset + (unsigned long long)(base + k0);
      unsigned hsh = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // (from the value before rounding, as the element-per-lane kernel does; one hash serves the two elements of a pair)
        d[e] = pr[e];
        if (p.drop_p > 0.f) {
          const unsigned long long idx = idx0 + e;
          if (e == 0 || (idx & 1) == 0) hsh = sc_rand_pair(seed, idx >> 1);
          d[e] = (((hsh >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr) ? 0.f : pr[e] * keep;
        }
      }
      *reinterpret_cast<uint4*>((unsigned short*)p.Pd + base + k0) = ln_pack8(d);
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const unsigned short* P, const unsigned short* dPd, unsigned short* dS, long long rows,
                                                              int Tk, long long ld, float out_scale, float drop_p,
                                                              const unsigned long long* seed_ptr, unsigned long long offset) {
  const unsigned long long seed = drop_p > 0.f ? *seed_ptr : 0ull;
  const unsigned thr = sc_drop_threshold(drop_p);
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long base = row * ld;
  const float keep = 1.f / (1.f - drop_p);
  float pr[NV][8], dp[NV][8];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k0 = 8 * (lane + 64 * c);
    if (k0 < ld) {
      ln_unpack8(*reinterpret_cast<const uint4*>(P + base + k0), pr[c]);
      ln_unpack8(*reinterpret_cast<const uint4*>(dPd + base + k0), dp[c]);
      const unsigned long long idx0 = offset + (unsigned long long)(base + k0);
      unsigned hsh = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (k0 + e < Tk) {
          if (drop_p > 0.f) {
            const unsigned long long idx = idx0 + e;
            if (e == 0 || (idx & 1) == 0) hsh = sc_rand_pair(seed, idx >> 1);
            dp[c][e] = (((hsh >> (16u * (unsigned)(idx & 1))) & 0xFFFFu) < thr) ? 0.f : dp[c][e] * keep;
          }
          dot += dp[c][e] * pr[c][e];
        } else {
          pr[c][e] = dp[c][e] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) pr[c][e] = dp[c][e] = 0.f;
    }
  }
  dot = sc_wave_sum(dot);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int k0 = 8 * (lane + 64 * c);
    if (k0 >= ld) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (k0 + e < Tk) ? pr[c][e] * (dp[c][e] - dot) * out_scale : 0.f;
    *reinterpret_cast<uint4*>(dS + base + k0) = ln_pack8(o);
  }
}

// dS = P * (dP - sum_k dP * P) * out_scale with dP = dPd * keepmask / (1 - p)
template <typename T, int NI>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const void* P, const void* dPd, void* dS, long long rows, int Tk,
                                                          long long ld, float out_scale, float drop_p,
                                                          const unsigned long long* seed_ptr, unsigned long long offset) {
  const unsigned long long seed = drop_p > 0.f ? *seed_ptr : 0ull;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long base = row * ld;
  const float keep = 1.f / (1.f - drop_p);
  float pr[NI], dp[NI];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    if (k < Tk) {
      pr[i] = Row<T>::ld(P, base + k);
      float d = Row<T>::ld(dPd, base + k);
      if (drop_p > 0.f) d = sc_dropped(seed, offset + base + k, sc_drop_threshold(drop_p)) ? 0.f : d * keep;
      dp[i] = d;
      dot += d * pr[i];
    } else {
      pr[i] = dp[i] = 0.f;
    }
  }
  dot = sc_wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int k = lane + 64 * i;
    if (k < ld) Row<T>::st(dS, base + k, (k < Tk) ? pr[i] * (dp[i] - dot) * out_scale : 0.f);
  }
}

// ---- cross entropy ---------------------------------------------------------------------------------------------------
// one workgroup per row: online max / sum-exp / sum over V in a single read of the logits
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const void* logits, const long long* target, long long R, int V, long long ld,
                                                     float inv_temp, float smoothing, float* row_loss, float* lse_out) {
  __shared__ float sm[3][4];
  const long long row = blockIdx.x;
  const long long base = row * ld;
  float mx = -INFINITY, se = 0.f, sx = 0.f;
  for (int k = threadIdx.x; k < V; k += 256) {
    const float x = Row<T>::ld(logits, base + k) * inv_temp;
    sx += x;
    if (x > mx) {
      se = se * expf(mx - x) + 1.f;
      mx = x;
    } else {
      se += expf(x - mx);
    }
  }
  // wave then workgroup combine of (mx, se) pairs
  const float wmx = sc_wave_max(mx);
  se = sc_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
  sx = sc_wave_sum(sx);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = wmx; sm[1][wid] = se; sm[2][wid] = sx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float gm = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
    float gs = 0.f, gx = 0.f;
    for (int w = 0; w < 4; ++w) { gs += sm[1][w] * expf(sm[0][w] - gm); gx += sm[2][w]; }
    const float lse = gm + logf(gs);
    const long long t = target[row];
    const float lp_t = Row<T>::ld(logits, base + t) * inv_temp - lse;
    const float sum_lp = gx - (float)V * lse;
    float loss = lp_t;
    if (smoothing > 0.f) loss = lp_t * (1.f - smoothing) + sum_lp * (smoothing / (float)V);
    row_loss[row] = -loss;
    lse_out[row] = lse;
  }
}

// dlogits[r][v] = grow[r] * inv_temp * (softmax - (1 - eps) * onehot - eps / V)
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const void* logits, const long long* target, const float* lse, const float* grow,
                                                     long long R, int V, long long ld, float inv_temp, float smoothing, void* dlogits) {
  const long long row = blockIdx.x;
  const long long base = row * ld;
  const float g = grow[row] * inv_temp, l = lse[row];
  const long long t = target[row];
  const float sneg = smoothing / (float)V;
  for (int k = threadIdx.x; k < V; k += 256) {
    float d = 0.f;
    if (g != 0.f) {
      const float pr = expf(Row<T>::ld(logits, base + k) * inv_temp - l);
      d = g * (pr - ((k == t) ? (1.f - smoothing) : 0.f) - sneg);
    }
    Row<T>::st(dlogits, base + k, d);
  }
}

// row-wise logsumexp only (CTC): lse[r]
template <typename T>
__global__ __launch_bounds__(256) void row_lse_kernel(const void* logits, long long R, int V, long long ld, float* lse_out) {
  __shared__ float sm[2][4];
  const long long base = (long long)blockIdx.x * ld;
  float mx = -INFINITY, se = 0.f;
  for (int k = threadIdx.x; k < V; k += 256) {
    const float x = Row<T>::ld(logits, base + k);
    if (x > mx) { se = se * expf(mx - x) + 1.f; mx = x; } else { se += expf(x - mx); }
  }
  const float wmx = sc_wave_max(mx);
  se = sc_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = wmx; sm[1][wid] = se; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gm = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
    float gs = 0.f;
    for (int w = 0; w < 4; ++w) gs += sm[1][w] * expf(sm[0][w] - gm);
    lse_out[blockIdx.x] = gm + logf(gs);
  }
}

// bf16 rows in 16-byte pieces (V, ld multiples of 8): one running (max, sum) pair per lane, rescaled once per piece
__global__ __launch_bounds__(256) void row_lse_vec_kernel(const unsigned short* logits, int V, long long ld, float* lse_out) {
  __shared__ float sm[2][4];
  const uint4* row = reinterpret_cast<const uint4*>(logits + (long long)blockIdx.x * ld);
  float mx = -INFINITY, se = 0.f;
  for (int c = threadIdx.x; c < (V >> 3); c += 256) {
    float f[8];
    ln_unpack8(row[c], f);
    float cm = f[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
    if (cm > mx) {
      se *= expf(mx - cm);  // (exp(-inf) = 0 on the first piece)
      mx = cm;
    }
    if (mx != -INFINITY)
#pragma unroll
      for (int e = 0; e < 8; ++e) se += expf(f[e] - mx);
  }
  const float wmx = sc_wave_max(mx);
  se = sc_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = wmx; sm[1][wid] = se; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gm = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
    float gs = 0.f;
    for (int w = 0; w < 4; ++w) gs += sm[0][w] == -INFINITY ? 0.f : sm[1][w] * expf(sm[0][w] - gm);
    lse_out[blockIdx.x] = gm + logf(gs);
  }
}

// loss = mean_b( sum_{l < len[b]} rows[b][l] / (len[b] + 1e-10) )   (length_normalized; else without the division): the reduction of the
// per-token losses of speechain/criterion/cross_entropy.py:201-212 in ONE launch (one workgroup; (B, L) is a few thousand floats) instead of
// mask construction, masked_fill, sum, divide and mean -- eight small launches on the path between the forward and the backward pass, and
// eight more coming back.
__global__ __launch_bounds__(256) void seq_loss_fwd_kernel(const float* rows, const long long* len, int B, int L, int length_normalized, float* loss) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x >> 6; b < B; b += 4) {  // one wave per sentence
    const int n = (int)min((long long)L, max(0LL, len[b]));
    float s = 0.f;
    for (int l = threadIdx.x & 63; l < n; l += 64) s += rows[(long long)b * L + l];
    s = sc_wave_sum(s);
    if (length_normalized) s = s / ((float)len[b] + 1e-10f);
    if ((threadIdx.x & 63) == 0) acc += (double)s;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; i += 64) t += red[i];
    loss[0] = (float)(t / (double)B);
  }
}

// grow[b][l] = g / (B (len[b] + 1e-10)) for l < len[b], else 0
__global__ __launch_bounds__(256) void seq_loss_bwd_kernel(const float* g, const long long* len, int B, int L, int length_normalized, float* grow) {
  const float gv = g[0] / (float)B;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < (long long)B * L; i += 256LL * gridDim.x) {
    const int b = (int)(i / L), l = (int)(i - (long long)b * L);
    const float d = length_normalized ? (float)len[b] + 1e-10f : 1.f;
    grow[i] = l < len[b] ? gv / d : 0.f;
  }
}

// Masked L1 / L2 / L1+L2 between two (B, T, D) tracks (speechain/criterion/least_error.py:96-129): sum over the valid frames t < len[b] of
// mean_d f(pred - tgt), divided by the number of valid frames (normalised) or by B (per-utterance sums, batch mean).  The reference's
// form is ~10 index- and frame-sized torch launches each way per term, and FastSpeech2 has five terms: 113 consecutive launches between
// the forward and the backward pass of its step.  Here: partial sums per workgroup (f32 in a lane, double across lanes), a one-workgroup
// finish in a fixed order, and one elementwise launch for the gradient.  loss_type: 1 = L1, 2 = L2, 3 = L1 + L2.
template <typename T>
struct LeIo;
template <> struct LeIo<float> {
  static __device__ __forceinline__ float ld(const float* p, long long i) { return p[i]; }
  static __device__ __forceinline__ void st(float* p, long long i, float v) { p[i] = v; }
};
template <> struct LeIo<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short* p, long long i) { return sc_bf16_to_f32(p[i]); }
  static __device__ __forceinline__ void st(unsigned short* p, long long i, float v) { p[i] = sc_f32_to_bf16(v); }
};

template <typename T>
__global__ __launch_bounds__(256) void least_error_partial_kernel(const T* pred, const float* tgt, const long long* len, int B, int T_, int D,
                                                                  int loss_type, double* partial) {
  __shared__ double red[4];
  float acc = 0.f;
  // frame by frame (one 32-bit quotient per frame, none per element); narrow tracks (D < 256: pitch, energy, duration) take the frames
  // of a workgroup's pass side by side, 256 / D of them
  const int fpp = D >= 256 ? 1 : 256 / D;  // frames per pass of the workgroup
  for (int r0 = blockIdx.x * fpp; r0 < B * T_; r0 += gridDim.x * fpp) {
    const int r = r0 + (D >= 256 ? 0 : (int)threadIdx.x / D);
    const int d0 = D >= 256 ? (int)threadIdx.x : (int)threadIdx.x % D;
    if (r < B * T_ && (D >= 256 || (int)threadIdx.x < fpp * D)) {
      const int b = r / T_, t = r - b * T_;
      if (t < len[b]) {
        const long long base = (long long)r * D;
        for (int dd = d0; dd < D; dd += 256) {
          const float d = LeIo<T>::ld(pred, base + dd) - tgt[base + dd];
          acc += loss_type == 1 ? fabsf(d) : loss_type == 2 ? d * d : fabsf(d) + d * d;
        }
      }
    }
  }
  const double w = (double)sc_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// sum_{b,t,d} a[b][t][d] * tab[t][d]: the gradient of the scalar weight of ScaledPositionalEncoding (speechain/module/transformer/pos_enc.py:
// alpha * table added to the input) from the gradient of the sum -- a cast, a broadcast product and a sum in torch (42 us per site).
template <typename T>
__global__ __launch_bounds__(256) void dot_table_partial_kernel(const T* a, const float* tab, int B, int T_, int D, double* partial) {
  __shared__ double red[4];
  float acc = 0.f;
  for (int r = blockIdx.x; r < B * T_; r += gridDim.x) {  // (row by row: one 32-bit remainder per row, none per element)
    const long long ra = (long long)r * D, rt = (long long)(r % T_) * D;
    for (int d = threadIdx.x; d < D; d += 256) acc = fmaf(LeIo<T>::ld(a, ra + d), tab[rt + d], acc);
  }
  const double w = (double)sc_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void sum_partials_to_float_kernel(const double* partial, int np, float* out) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < np; ++i) s += partial[i];
    out[0] = (float)s;
  }
}

// denominator shared by both directions: D x (number of valid frames, or B)
__device__ __forceinline__ double least_error_denominator(const long long* len, int B, int T_, int D, int normalized) {
  if (!normalized) return (double)D * (double)B;
  long long c = 0;
  for (int b = 0; b < B; ++b) c += min((long long)T_, max(0LL, len[b]));
  return (double)D * (double)c;
}

__global__ __launch_bounds__(64) void least_error_finish_kernel(const double* partial, int np, const long long* len, int B, int T_, int D,
                                                                int normalized, float* loss) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < np; ++i) s += partial[i];
    loss[0] = (float)(s / least_error_denominator(len, B, T_, D, normalized));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void least_error_bwd_kernel(const T* pred, const float* tgt, const long long* len, int B, int T_, int D,
                                                              int loss_type, int normalized, const float* g, T* dpred) {
  __shared__ float s_scale;
  if (threadIdx.x == 0) s_scale = (float)((double)g[0] / least_error_denominator(len, B, T_, D, normalized));
  __syncthreads();
  const float scale = s_scale;
  const int fpp = D >= 256 ? 1 : 256 / D;
  for (int r0 = blockIdx.x * fpp; r0 < B * T_; r0 += gridDim.x * fpp) {
    const int r = r0 + (D >= 256 ? 0 : (int)threadIdx.x / D);
    const int d0 = D >= 256 ? (int)threadIdx.x : (int)threadIdx.x % D;
    if (r < B * T_ && (D >= 256 || (int)threadIdx.x < fpp * D)) {
      const int b = r / T_, t = r - b * T_;
      const bool valid = t < len[b];
      const long long base = (long long)r * D;
      for (int dd = d0; dd < D; dd += 256) {
        float v = 0.f;
        if (valid) {
          const float d = LeIo<T>::ld(pred, base + dd) - tgt[base + dd];
          const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // (torch: the gradient of |x| at 0 is 0)
          v = scale * (loss_type == 1 ? sg : loss_type == 2 ? 2.f * d : sg + 2.f * d);
        }
        LeIo<T>::st(dpred, base + dd, v);
      }
    }
  }
}

}  // namespace

#define SC_DISPATCH_TN(dtype, n, KERNEL, N0, N1, N2, ...)                                                  \
  do {                                                                                                     \
    if ((dtype) == 0) {                                                                                    \
      if ((n) <= 64 * N0) hipLaunchKernelGGL((KERNEL<float, N0>), __VA_ARGS__);                            \
      else if ((n) <= 64 * N1) hipLaunchKernelGGL((KERNEL<float, N1>), __VA_ARGS__);                       \
      else hipLaunchKernelGGL((KERNEL<float, N2>), __VA_ARGS__);                                           \
    } else {                                                                                               \
      if ((n) <= 64 * N0) hipLaunchKernelGGL((KERNEL<unsigned short, N0>), __VA_ARGS__);                   \
      else if ((n) <= 64 * N1) hipLaunchKernelGGL((KERNEL<unsigned short, N1>), __VA_ARGS__);              \
      else hipLaunchKernelGGL((KERNEL<unsigned short, N2>), __VA_ARGS__);                                  \
    }                                                                                                      \
  } while (0)

#define SC_DISPATCH_T(dtype, KERNEL, ...)                                        \
  do {                                                                           \
    if ((dtype) == 0) hipLaunchKernelGGL(KERNEL<float>, __VA_ARGS__);            \
    else hipLaunchKernelGGL(KERNEL<unsigned short>, __VA_ARGS__);                \
  } while (0)

SC_API int sc_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                            long long R, int D, float eps, hipStream_t stream) {
  SC_CHECK_ARG(x && gamma && beta && y, "sc_layernorm_fwd: null pointer");
  SC_CHECK_ARG(R >= 1 && D >= 1 && D <= 64 * LN_MAX, "sc_layernorm_fwd: need 1 <= D <= %d, got %d", 64 * LN_MAX, D);
  const bool vec = dtype == 1 && D % 8 == 0 && D <= 2048 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0);
  if (vec) {
    const dim3 grid((unsigned)((R + 3) / 4));
    if (D <= 512) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<1>, grid, dim3(256), 0, stream, (const unsigned short*)x, gamma, beta, (unsigned short*)y, mean, rstd, R, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<2>, grid, dim3(256), 0, stream, (const unsigned short*)x, gamma, beta, (unsigned short*)y, mean, rstd, R, D, eps);
    else hipLaunchKernelGGL(layernorm_fwd_vec_kernel<4>, grid, dim3(256), 0, stream, (const unsigned short*)x, gamma, beta, (unsigned short*)y, mean, rstd, R, D, eps);
  } else {
    SC_DISPATCH_TN(dtype, D, layernorm_fwd_kernel, 8, 16, 32, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, R, D, eps);
  }
  SC_LAUNCH_CHECK("layernorm_fwd_kernel");
  return SC_OK;
}

// The partition of sc_layernorm_bwd: rows per workgroup (4 waves, one row each at a time) and the number of workgroups, i.e. of partial
// dgamma / dbeta rows.  (A/B knobs: SC_LN_BWD_ROWS = rows per workgroup aimed for, SC_LN_BWD_CAP = most workgroups.)
static void ln_bwd_partition(long long R, int* rpb_out, long long* blocks_out) {
  static const long long rows = [] { const char* e = getenv("SC_LN_BWD_ROWS"); return e ? atoll(e) : 8LL; }();
  static const long long cap = [] { const char* e = getenv("SC_LN_BWD_CAP"); return e ? atoll(e) : 1024LL; }();
  long long blocks = (R + rows - 1) / rows;
  if (blocks > cap) blocks = cap;
  const int rpb = (int)((R + blocks - 1) / blocks);
  *rpb_out = rpb;
  *blocks_out = (R + rpb - 1) / rpb;
}

SC_API long long sc_layernorm_bwd_workspace(long long R, int D) {
  int rpb;
  long long blocks;
  ln_bwd_partition(R, &rpb, &blocks);
  return blocks * 2 * D * (long long)sizeof(float);
}

// 1 if sc_layernorm_bwd_dz can write its masked second output for this case (the 16-byte vector kernel: bf16, D % 8 == 0, D <= 2048)
SC_API int sc_layernorm_bwd_dz_supported(int dtype, int D) { return dtype == 1 && D % 8 == 0 && D <= 2048; }

static int layernorm_bwd_impl(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              void* dx, float* dgamma, float* dbeta, long long R, int D, const void* dres, void* workspace,
                              long long ws_bytes, int stages, void* dz, float drop_p, const unsigned long long* seed,
                              unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace, "sc_layernorm_bwd: null pointer");
  SC_CHECK_ARG(stages == 1 || stages == 3, "sc_layernorm_bwd: stages is 1 (dx + partial sums; finish with sc_layernorm_bwd_finish) or 3 (everything)");
  SC_CHECK_ARG(R >= 1 && D >= 1 && D <= 64 * LN_MAX, "sc_layernorm_bwd: need 1 <= D <= %d, got %d", 64 * LN_MAX, D);
  SC_CHECK_ARG(ws_bytes >= sc_layernorm_bwd_workspace(R, D), "sc_layernorm_bwd: workspace too small");
  int rpb;
  long long blocks;
  ln_bwd_partition(R, &rpb, &blocks);
  const size_t smem = (size_t)8 * D * sizeof(float);
  SC_CHECK_ARG(smem <= 64 * 1024, "sc_layernorm_bwd: D too large for the LDS reduction");
  const bool vec = dtype == 1 && D % 8 == 0 && D <= 2048 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)dres | (uintptr_t)dz) % 16 == 0);
  SC_CHECK_ARG(dz == nullptr || vec, "sc_layernorm_bwd_dz: the masked second output needs bf16, D %% 8 == 0, D <= 2048 and 16-byte aligned tensors");
  SC_CHECK_ARG(dz == nullptr || (drop_p > 0.f && drop_p < 1.f && seed != nullptr && offset % 2 == 0), "sc_layernorm_bwd_dz: 0 < p < 1, a seed cell and an even offset");
  if (vec) {
    const dim3 grid((unsigned)blocks);
    const unsigned short *dy16 = (const unsigned short*)dy, *x16 = (const unsigned short*)x;
    unsigned short* dz16 = (unsigned short*)dz;
    if (D <= 512) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<1>, grid, dim3(256), smem, stream, dy16, x16, mean, rstd, gamma, (unsigned short*)dx, (float*)workspace, R, D, rpb, (const unsigned short*)dres, dz16, drop_p, seed, offset);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<2>, grid, dim3(256), smem, stream, dy16, x16, mean, rstd, gamma, (unsigned short*)dx, (float*)workspace, R, D, rpb, (const unsigned short*)dres, dz16, drop_p, seed, offset);
    else hipLaunchKernelGGL(layernorm_bwd_vec_kernel<4>, grid, dim3(256), smem, stream, dy16, x16, mean, rstd, gamma, (unsigned short*)dx, (float*)workspace, R, D, rpb, (const unsigned short*)dres, dz16, drop_p, seed, offset);
  } else {
    SC_DISPATCH_TN(dtype, D, layernorm_bwd_kernel, 8, 16, 32, dim3((unsigned)blocks), dim3(256), smem, stream, dy, x, mean, rstd, gamma, dx,
                  (float*)workspace, R, D, rpb, dres);
  }
  SC_LAUNCH_CHECK("layernorm_bwd_kernel");
  if (stages & 2) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(sc_cdiv(D, 16), 2), dim3(1024), 0, stream, (const float*)workspace, (int)blocks, 2, D, dgamma, dbeta);
    SC_LAUNCH_CHECK("reduce_parts_kernel");
  }
  return SC_OK;
}

SC_API int sc_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                            void* dx, float* dgamma, float* dbeta, long long R, int D, const void* dres, void* workspace,
                            long long ws_bytes, int stages, hipStream_t stream) {
  return layernorm_bwd_impl(dtype, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, R, D, dres, workspace, ws_bytes, stages, nullptr, 0.f,
                            nullptr, 0ull, stream);
}

SC_API int sc_layernorm_bwd_dz(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                               void* dx, float* dgamma, float* dbeta, long long R, int D, const void* dres, void* workspace,
                               long long ws_bytes, int stages, void* dz, float drop_p, const unsigned long long* seed,
                               unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(dz, "sc_layernorm_bwd_dz: null dz");
  return layernorm_bwd_impl(dtype, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, R, D, dres, workspace, ws_bytes, stages, dz, drop_p, seed,
                            offset, stream);
}

// Second stage of up to LN_BATCH LayerNorm backward calls in ONE launch (grid.z = call): in a training step the 44 parameter-gradient
// reductions are not needed before the step ends, and as separate launches each costs its run time plus a dependency gap on the stream.
SC_API int sc_layernorm_bwd_finish(int n, const void* const* workspaces, const long long* R, const int* D, float* const* dgamma,
                                   float* const* dbeta, hipStream_t stream) {
  SC_CHECK_ARG(workspaces && R && D && dgamma && dbeta, "sc_layernorm_bwd_finish: null pointer");
  SC_CHECK_ARG(n >= 1, "sc_layernorm_bwd_finish: n=%d", n);
  for (int i0 = 0; i0 < n; i0 += LN_BATCH) {
    LnFinishBatch b{};
    const int m = std::min(LN_BATCH, n - i0);
    int dmax = 0;
    for (int i = 0; i < m; ++i) {
      const int k = i0 + i;
      SC_CHECK_ARG(workspaces[k] && dgamma[k] && dbeta[k] && R[k] >= 1 && D[k] >= 1, "sc_layernorm_bwd_finish: bad entry %d", k);
      int rpb;
      long long blocks;
      ln_bwd_partition(R[k], &rpb, &blocks);
      b.part[i] = (const float*)workspaces[k];
      b.out0[i] = dgamma[k];
      b.out1[i] = dbeta[k];
      b.P[i] = (int)blocks;
      b.D[i] = D[k];
      dmax = std::max(dmax, D[k]);
    }
    hipLaunchKernelGGL(reduce_parts_batched_kernel, dim3(sc_cdiv(dmax, 16), 2, m), dim3(1024), 0, stream, b);
    SC_LAUNCH_CHECK("reduce_parts_batched_kernel");
  }
  return SC_OK;
}

SC_API long long sc_colsum_workspace(long long R, int N) {
  long long strips = (R + 63) / 64;
  if (strips > 128) strips = 128;
  return strips * N * (long long)sizeof(float);
}

SC_API int sc_colsum(int dtype, const void* x, long long R, int N, long long ld, float* out, void* workspace, long long ws_bytes,
                     hipStream_t stream) {
  SC_CHECK_ARG(x && out && workspace, "sc_colsum: null pointer");
  SC_CHECK_ARG(R >= 1 && N >= 1 && ld >= N, "sc_colsum: bad sizes");
  SC_CHECK_ARG(ws_bytes >= sc_colsum_workspace(R, N), "sc_colsum: workspace too small");
  long long strips = (R + 63) / 64;
  if (strips > 128) strips = 128;
  const int rpb = (int)((R + strips - 1) / strips);
  strips = (R + rpb - 1) / rpb;
  SC_DISPATCH_T(dtype, colsum_kernel, dim3(sc_cdiv(N, 64), (unsigned)strips), dim3(256), 0, stream, x, R, N, ld, rpb, (float*)workspace);
  SC_LAUNCH_CHECK("colsum_kernel");
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(sc_cdiv(N, 16), 1), dim3(1024), 0, stream, (const float*)workspace, (int)strips, 1, N, out, out);
  SC_LAUNCH_CHECK("reduce_parts_kernel");
  return SC_OK;
}

SC_API int sc_softmax_fwd(int dtype, const void* S, void* P, void* Pd, int Z, int H, int Tq, int Tk, long long ld,
                          const unsigned char* mask, long long mask_b, long long mask_q, int causal, float drop_p,
                          const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(S && P, "sc_softmax_fwd: null pointer");
  SC_CHECK_ARG(Z >= 1 && H >= 1 && Tq >= 1 && Tk >= 1 && ld >= Tk, "sc_softmax_fwd: bad sizes");
  SC_CHECK_ARG(ld <= 64 * SM_MAX, "sc_softmax_fwd: row length %lld > %d keys", ld, 64 * SM_MAX);
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sc_softmax_fwd: dropout p=%f needs a seed cell", drop_p);
  SoftmaxArgs a{S, P, Pd, Z, H, Tq, Tk, ld, mask, mask_b, mask_q, causal, drop_p, seed, offset};
  const long long rows = (long long)Z * Tq;
  if (dtype == 1 && ld % 8 == 0 && ld <= 1024 && (((uintptr_t)S | (uintptr_t)P | (uintptr_t)Pd) % 16 == 0)) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (ld <= 512) hipLaunchKernelGGL(softmax_fwd_vec_kernel<1>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(softmax_fwd_vec_kernel<2>, grid, dim3(256), 0, stream, a);
    SC_LAUNCH_CHECK("softmax_fwd_vec_kernel");
    return SC_OK;
  }
  SC_DISPATCH_TN(dtype, ld, softmax_fwd_kernel, 4, 8, 16, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK("softmax_fwd_kernel");
  return SC_OK;
}

SC_API int sc_softmax_bwd(int dtype, const void* P, const void* dPd, void* dS, long long rows, int Tk, long long ld, float out_scale,
                          float drop_p, const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(P && dPd && dS, "sc_softmax_bwd: null pointer");
  SC_CHECK_ARG(rows >= 1 && Tk >= 1 && ld >= Tk && ld <= 64 * SM_MAX, "sc_softmax_bwd: bad sizes");
  if (dtype == 1 && ld % 8 == 0 && ld <= 1024 && (((uintptr_t)P | (uintptr_t)dPd | (uintptr_t)dS) % 16 == 0)) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (ld <= 512)
      hipLaunchKernelGGL(softmax_bwd_vec_kernel<1>, grid, dim3(256), 0, stream, (const unsigned short*)P, (const unsigned short*)dPd,
                         (unsigned short*)dS, rows, Tk, ld, out_scale, drop_p, seed, offset);
    else
      hipLaunchKernelGGL(softmax_bwd_vec_kernel<2>, grid, dim3(256), 0, stream, (const unsigned short*)P, (const unsigned short*)dPd,
                         (unsigned short*)dS, rows, Tk, ld, out_scale, drop_p, seed, offset);
    SC_LAUNCH_CHECK("softmax_bwd_vec_kernel");
    return SC_OK;
  }
  SC_DISPATCH_TN(dtype, ld, softmax_bwd_kernel, 4, 8, 16, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, P, dPd, dS, rows, Tk, ld, out_scale,
                drop_p, seed, offset);
  SC_LAUNCH_CHECK("softmax_bwd_kernel");
  return SC_OK;
}

SC_API int sc_ce_fwd(int dtype, const void* logits, const long long* target, long long R, int V, long long ld, float temperature,
                     float smoothing, float* row_loss, float* lse, hipStream_t stream) {
  SC_CHECK_ARG(logits && target && row_loss && lse, "sc_ce_fwd: null pointer");
  SC_CHECK_ARG(R >= 1 && V >= 1 && ld >= V && temperature > 0.f, "sc_ce_fwd: bad sizes");
  SC_DISPATCH_T(dtype, ce_fwd_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, target, R, V, ld, 1.f / temperature, smoothing, row_loss, lse);
  SC_LAUNCH_CHECK("ce_fwd_kernel");
  return SC_OK;
}

SC_API int sc_ce_bwd(int dtype, const void* logits, const long long* target, const float* lse, const float* grow, long long R, int V,
                     long long ld, float temperature, float smoothing, void* dlogits, hipStream_t stream) {
  SC_CHECK_ARG(logits && target && lse && grow && dlogits, "sc_ce_bwd: null pointer");
  SC_CHECK_ARG(R >= 1 && V >= 1 && ld >= V && temperature > 0.f, "sc_ce_bwd: bad sizes");
  SC_DISPATCH_T(dtype, ce_bwd_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, target, lse, grow, R, V, ld, 1.f / temperature, smoothing, dlogits);
  SC_LAUNCH_CHECK("ce_bwd_kernel");
  return SC_OK;
}

SC_API int sc_row_lse(int dtype, const void* logits, long long R, int V, long long ld, float* lse, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse, "sc_row_lse: null pointer");
  SC_CHECK_ARG(R >= 1 && V >= 1 && ld >= V, "sc_row_lse: bad sizes");
  if (dtype == 1 && (V & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)logits & 15) == 0)
    hipLaunchKernelGGL(row_lse_vec_kernel, dim3((unsigned)R), dim3(256), 0, stream, (const unsigned short*)logits, V, ld, lse);
  else
    SC_DISPATCH_T(dtype, row_lse_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, R, V, ld, lse);
  SC_LAUNCH_CHECK("row_lse_kernel");
  return SC_OK;
}

// Masked, length-normalised mean of per-token losses (speechain/criterion/cross_entropy.py:201-212 without token weights / confidence
// filtering): rows (B, L) f32, len (B) int64 -> loss (1) f32; and its gradient with respect to rows for an upstream scalar g (device).
SC_API int sc_seq_loss_fwd(const float* rows, const long long* len, int B, int L, int length_normalized, float* loss, hipStream_t stream) {
  SC_CHECK_ARG(rows && len && loss && B >= 1 && L >= 1, "sc_seq_loss_fwd: bad arguments");
  hipLaunchKernelGGL(seq_loss_fwd_kernel, dim3(1), dim3(256), 0, stream, rows, len, B, L, length_normalized, loss);
  SC_LAUNCH_CHECK("seq_loss_fwd_kernel");
  return SC_OK;
}
SC_API int sc_seq_loss_bwd(const float* g, const long long* len, int B, int L, int length_normalized, float* grow, hipStream_t stream) {
  SC_CHECK_ARG(g && len && grow && B >= 1 && L >= 1, "sc_seq_loss_bwd: bad arguments");
  hipLaunchKernelGGL(seq_loss_bwd_kernel, dim3((unsigned)sc_cdiv(B * L, 256)), dim3(256), 0, stream, g, len, B, L, length_normalized, grow);
  SC_LAUNCH_CHECK("seq_loss_bwd_kernel");
  return SC_OK;
}

// LeastError (speechain/criterion/least_error.py:96-129) of pred (B, T, D) f32 / bf16 against tgt (B, T, D) f32 over the frames t < len[b]:
// loss (1) f32 = sum_valid mean_d f(pred - tgt) / (valid frames | B); workspace: sc_least_error_workspace() bytes.  _bwd: dpred (B, T, D) in
// pred's dtype for an upstream scalar g (device), zero at the frames past the lengths.  dtype 0 = f32, 1 = bf16; loss_type 1 L1, 2 L2, 3 L1+L2.
static int least_error_blocks(long long frames, int D) { const int fpp = D >= 256 ? 1 : 256 / D; return (int)std::min<long long>(512, (frames + fpp - 1) / fpp); }
SC_API long long sc_least_error_workspace() { return 512 * (long long)sizeof(double); }
SC_API int sc_least_error_fwd(int dtype, const void* pred, const float* tgt, const long long* len, int B, int T, int D, int loss_type, int normalized,
                              void* workspace, float* loss, hipStream_t stream) {
  SC_CHECK_ARG(pred && tgt && len && workspace && loss && B >= 1 && T >= 1 && D >= 1 && (long long)B * T <= 0x3fffffffLL, "sc_least_error_fwd: bad arguments");
  SC_CHECK_ARG((dtype == 0 || dtype == 1) && loss_type >= 1 && loss_type <= 3, "sc_least_error_fwd: dtype 0 / 1, loss_type 1 .. 3");
  const int nb = least_error_blocks((long long)B * T, D);
  if (dtype == 0)
    hipLaunchKernelGGL(least_error_partial_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)pred, tgt, len, B, T, D, loss_type,
                       (double*)workspace);
  else
    hipLaunchKernelGGL(least_error_partial_kernel<unsigned short>, dim3(nb), dim3(256), 0, stream, (const unsigned short*)pred, tgt, len, B, T, D,
                       loss_type, (double*)workspace);
  SC_LAUNCH_CHECK("least_error_partial_kernel");
  hipLaunchKernelGGL(least_error_finish_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, nb, len, B, T, D, normalized, loss);
  SC_LAUNCH_CHECK("least_error_finish_kernel");
  return SC_OK;
}
// out (1) f32 = sum over (b, t, d) of a (B, T, D) f32 / bf16 times tab (T, D) f32 (deterministic: fixed partial order); workspace:
// sc_least_error_workspace() bytes.
SC_API int sc_dot_table(int dtype, const void* a, const float* tab, int B, int T, int D, void* workspace, float* out, hipStream_t stream) {
  SC_CHECK_ARG(a && tab && workspace && out && B >= 1 && T >= 1 && D >= 1 && (dtype == 0 || dtype == 1) && (long long)B * T <= 0x7fffffffLL,
               "sc_dot_table: bad arguments");
  const int nb = (int)std::min<long long>(512, (long long)B * T);
  if (dtype == 0)
    hipLaunchKernelGGL(dot_table_partial_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)a, tab, B, T, D, (double*)workspace);
  else
    hipLaunchKernelGGL(dot_table_partial_kernel<unsigned short>, dim3(nb), dim3(256), 0, stream, (const unsigned short*)a, tab, B, T, D,
                       (double*)workspace);
  SC_LAUNCH_CHECK("dot_table_partial_kernel");
  hipLaunchKernelGGL(sum_partials_to_float_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, nb, out);
  SC_LAUNCH_CHECK("sum_partials_to_float_kernel");
  return SC_OK;
}
SC_API int sc_least_error_bwd(int dtype, const void* pred, const float* tgt, const long long* len, int B, int T, int D, int loss_type, int normalized,
                              const float* g, void* dpred, hipStream_t stream) {
  SC_CHECK_ARG(pred && tgt && len && g && dpred && B >= 1 && T >= 1 && D >= 1 && (long long)B * T <= 0x3fffffffLL, "sc_least_error_bwd: bad arguments");
  SC_CHECK_ARG((dtype == 0 || dtype == 1) && loss_type >= 1 && loss_type <= 3, "sc_least_error_bwd: dtype 0 / 1, loss_type 1 .. 3");
  const int fpp = D >= 256 ? 1 : 256 / D;
  const int nb = (int)std::min<long long>(4096, ((long long)B * T + fpp - 1) / fpp);
  if (dtype == 0)
    hipLaunchKernelGGL(least_error_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)pred, tgt, len, B, T, D, loss_type, normalized, g,
                       (float*)dpred);
  else
    hipLaunchKernelGGL(least_error_bwd_kernel<unsigned short>, dim3(nb), dim3(256), 0, stream, (const unsigned short*)pred, tgt, len, B, T, D,
                       loss_type, normalized, g, (unsigned short*)dpred);
  SC_LAUNCH_CHECK("least_error_bwd_kernel");
  return SC_OK;
}

