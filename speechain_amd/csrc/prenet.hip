// Conv2d prenet for gfx950 -- the device work of speechain/module/prenet/conv2d.py:168-205,227-280 (reference path /root/reference/):
// [Conv2d(k, s, p) -> BatchNorm2d -> activation] blocks on a (B, T, F) feature image, then flattening for the Linear part.
//
// Layout: every intermediate is CHANNELS-LAST, (B, T', F', C) -- a (B*T'*F', C) matrix -- so that
//   * a block with C_in > 1 is im2col (KH contiguous runs of KW*C_in elements per output position) + ONE call of the library's GEMM
//     (z = col Wm^T, Wm[c_out][(kh, kw, c_in)]), its weight gradient the split-K GEMM dz^T col and its data gradient dz Wm followed by
//     col2im written as a gather (no atomics);
//   * BatchNorm statistics are column sums of that matrix, normalisation + activation one elementwise pass;
//   * the last block's output IS the (B*T', F'*C) operand of the Linear part (the host permutes the Linear weight's columns from the
//     reference's channel-major flattening, conv2d.py:251, once per step instead of transposing the activations).
// The FIRST block has one input channel and K = KH*KW (9) taps: far too thin for matrix cores and, with 512 output channels, its
// output (B*500*39*512 values) is the largest tensor of the prenet.  It is never materialised before normalisation: because
// z[pos][c] = w_c . patch[pos] is linear in the patch, the batch statistics of every channel follow from the KK x KK second-moment
// matrix S = sum_pos patch patch^T and P = sum_pos patch (accumulated once in float64, independent of C):
//   mean_c = w_c.P / N,  E[z_c^2] = w_c^T S w_c / N;
// the output pass recomputes z from the staged input rows and writes act(scale_c z + shift_c) once.  The same identities give the
// block's weight / gamma / beta gradients from G[c][k] = sum_pos g[pos][c] patch_k[pos] and sum_pos g (g = dy * act'), one pass over dy.
// All kernels here are HBM-bound byte movers; sums that decide statistics are finished in float64.
#include "sc_common.h"
#include <algorithm>

namespace {

enum { PACT_NONE = 0, PACT_RELU = 1, PACT_LEAKY = 2, PACT_TANH = 3 };

struct Geo {
  int B, Ti, Fi;          // input image (per channel)
  int To, Fo;             // output image
  int KH, KW, SH, SW, PH, PW;
};

__device__ __forceinline__ float pact(float v, int act, float slope) {
  if (act == PACT_RELU) return v > 0.f ? v : 0.f;
  if (act == PACT_LEAKY) return v > 0.f ? v : v * slope;
  if (act == PACT_TANH) return tanhf(v);
  return v;
}
__device__ __forceinline__ float pact_grad(float v, int act, float slope) {  // torch: derivative at 0 is 0 (ReLU) / slope (LeakyReLU)
  if (act == PACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (act == PACT_LEAKY) return v > 0.f ? 1.f : slope;
  if (act == PACT_TANH) { const float t = tanhf(v); return 1.f - t * t; }
  return 1.f;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short* p) { return sc_bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = sc_f32_to_bf16(v); }
};

// 4 consecutive elements <-> float[4] (8- or 16-byte access when `vec`, else element-wise with a bound)
template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&v)[4], bool vec, int n) {
  if (vec) {
    if constexpr (sizeof(T) == 4) {
      const float4 q = *reinterpret_cast<const float4*>(p);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      const uint2 q = *reinterpret_cast<const uint2*>(p);
      v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xFFFF0000u);
      v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xFFFF0000u);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = i < n ? Elem<T>::ld(p + i) : 0.f;
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const float (&v)[4], bool vec, int n) {
  if (vec) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2 q;
      q.x = (unsigned)sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
      q.y = (unsigned)sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
      *reinterpret_cast<uint2*>(p) = q;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n) Elem<T>::st(p + i, v[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// First block, statistics: S[i][j] = sum_pos p_i p_j, P[j] = sum_pos p_j over all output positions (padding taps read 0).
// grid = row groups; a workgroup stages the KH input rows of one output row in LDS, thread q < KK*KK + KK owns one entry and
// walks the Fo positions of the row.  Partials [group][KK*KK + KK] in float64; conv1_moments_finish sums them in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------------------
// out[n] = sum_g partial[g][n] in float64: 1024 threads = 32 columns x 32 row lanes, fixed order (deterministic)
template <typename TIn>
__global__ __launch_bounds__(1024) void sum_partials_kernel(const TIn* partial, int G, long long N, double* out) {
  __shared__ double red[32][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const long long n = (long long)blockIdx.x * 32 + cx;
  double s = 0.0;
  if (n < N)
    for (int g = ry; g < G; g += 32) s += (double)partial[(long long)g * N + n];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < N) {
    double t = 0.0;
#pragma unroll
    for (int l = 0; l < 32; ++l) t += red[l][cx];
    out[n] = t;
  }
}

// A tile = R consecutive output rows of one image; its (R-1)*SH + KH input rows are staged in LDS once.  A workgroup walks
// `tiles_per_group` tiles.
struct Tiling {
  int R, tiles_per_b, tiles_per_group, groups;
};

constexpr int MAX_TAPS = 15;  // KH*KW of the first block (9 in every recipe): KK*KK + KK <= 256 entries, one thread each

__global__ __launch_bounds__(256) void conv1_moments_kernel(const float* x, long long sb, long long st, Geo g, Tiling tl, double* partial) {
  extern __shared__ float rows[];  // [(R-1)*SH + KH][Fi + 2*PW] (zero borders)
  __shared__ double half[128];     // the second row-lane's sums (entries < 128 have two lanes)
  const int KK = g.KH * g.KW, NE = KK * KK + KK;
  const int W = g.Fi + 2 * g.PW;
  // entry q < NE of the moment set; with NE <= 128 two threads share an entry and split a tile's output rows (even / odd)
  const int split = NE <= 128 ? 2 : 1;
  const int q = threadIdx.x % 128, sub = threadIdx.x / 128;
  const bool live = split == 2 ? q < NE : (int)threadIdx.x < NE;
  const int e = split == 2 ? q : (int)threadIdx.x;
  int oi = 0, oj = 0;  // LDS offsets of the two taps of this entry, relative to the patch origin
  bool pair = false;
  if (e < KK * KK) {
    const int i = e / KK, j = e % KK;
    oi = (i / g.KW) * W + (i % g.KW);
    oj = (j / g.KW) * W + (j % g.KW);
    pair = true;
  } else if (e < NE) {
    const int j = e - KK * KK;
    oi = (j / g.KW) * W + (j % g.KW);
  }
  double acc = 0.0;
  const int ntiles = g.B * tl.tiles_per_b;
  const int nin = (tl.R - 1) * g.SH + g.KH;
  for (int tile = blockIdx.x * tl.tiles_per_group; tile < (blockIdx.x + 1) * tl.tiles_per_group && tile < ntiles; ++tile) {
    const int b = tile / tl.tiles_per_b, to0 = (tile % tl.tiles_per_b) * tl.R;
    const int nr = min(tl.R, g.To - to0);
    __syncthreads();
    for (int i = threadIdx.x; i < nin * W; i += blockDim.x) {
      const int r = i / W, f = i % W - g.PW;
      const int t = to0 * g.SH - g.PH + r;
      rows[i] = (t >= 0 && t < g.Ti && f >= 0 && f < g.Fi) ? x[(long long)b * sb + (long long)t * st + f] : 0.f;
    }
    __syncthreads();
    if (live)
      for (int ro = (split == 2 ? sub : 0); ro < nr; ro += split) {
        const float* base = rows + ro * g.SH * W;
        for (int fo = 0; fo < g.Fo; ++fo) {  // float64 throughout: the variance is a difference of these moments
          const double a = (double)base[fo * g.SW + oi];
          acc += pair ? a * (double)base[fo * g.SW + oj] : a;
        }
      }
  }
  if (split == 2) {  // fixed order: lane 0 + lane 1
    if (sub == 1) half[q] = acc;
    __syncthreads();
    if (sub == 0 && live) partial[(long long)blockIdx.x * NE + e] = acc + half[q];
  } else if (live) {
    partial[(long long)blockIdx.x * NE + e] = acc;
  }
}

// BatchNorm2d coefficients of channel c from the moments (training): scale = gamma * rstd, shift = beta - mean * scale; the
// running statistics move by `momentum` with the UNBIASED variance (torch.nn.BatchNorm2d).
__global__ void conv1_bn_coeffs_kernel(const double* moments, const float* w, int C, int KK, double count, const double* count_dev,
                                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                       float* running_var, float* scale, float* shift, float* mean, float* rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;  // (synchronised statistics: the positions of ALL ranks, summed on the device)
  const double* S = moments;
  const double* P = moments + KK * KK;
  double m = 0.0, e2 = 0.0;
  for (int i = 0; i < KK; ++i) {
    const double wi = (double)w[c * KK + i];
    m += wi * P[i];
    double row = 0.0;
    for (int j = 0; j < KK; ++j) row += S[i * KK + j] * (double)w[c * KK + j];
    e2 += wi * row;
  }
  m /= count;
  double var = e2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(ga * rs);
  shift[c] = (float)(be - m * ga * rs);
  mean[c] = (float)m;
  rstd[c] = (float)rs;
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
  if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (count > 1.0 ? count / (count - 1.0) : 1.0));
}

// First block, output: y[b][to][fo][c] = act(scale_c * (w_c . patch) + shift_c), channels-last.  A workgroup walks tiles of R output
// rows whose input rows are staged in LDS once; a thread owns 4 consecutive channels (their KK x 4 weights stay in registers)
// and strides over fo.  grid.y tiles the channels in slabs of 1024.
#define SC_CONV1_THREAD_SETUP(SLAB)                                                                                         \
  constexpr int NT = KKT ? KKT : MAX_TAPS; /* register arrays sized for the taps that exist */                              \
  const int KK = KKT ? KKT : g.KH * g.KW;  /* KKT = 9: the per-tap guards of the unrolled loops fold away */                \
  const int W = g.Fi + 2 * g.PW;                                                                                            \
  const int c0 = blockIdx.y * (SLAB);                                                                                       \
  const int CS = min((SLAB), C - c0);                                                                                       \
  const int CSp = (CS + 3) & ~3;                                                                                            \
  const int CQ = CSp >> 2;                                                                                                  \
  const int lanes = max(1, (int)blockDim.x / CQ);                                                                           \
  const int cq = threadIdx.x % CQ, lane = threadIdx.x / CQ;                                                                 \
  const bool live = threadIdx.x < CQ * lanes;                                                                               \
  const int c = c0 + cq * 4;                                                                                                \
  const int nc = max(0, min(4, C - c));                                                                                     \
  const bool vec = (C % 4 == 0);                                                                                            \
  float sc[4], sh[4];                                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                           \
    sc[i] = (live && i < nc) ? scale[c + i] : 0.f;                                                                          \
    sh[i] = (live && i < nc) ? shift[c + i] : 0.f;                                                                          \
  }                                                                                                                         \
  int toff[NT];                                                                                                       \
  float wr[NT][4];                                                                                                    \
  _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                                                    \
    toff[k] = k < KK ? (k / g.KW) * W + (k % g.KW) : 0;                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) wr[k][i] = (k < KK && live && i < nc) ? w[(long long)(c + i) * KK + k] : 0.f; \
  }

#define SC_CONV1_STAGE_TILE()                                                                                               \
  const int b = tile / tl.tiles_per_b, to0 = (tile % tl.tiles_per_b) * tl.R;                                                \
  const int npos = min(tl.R, g.To - to0) * g.Fo;                                                                            \
  __syncthreads();                                                                                                          \
  for (int i = threadIdx.x; i < nin * W; i += blockDim.x) {                                                                 \
    const int r = i / W, f = i % W - g.PW;                                                                                  \
    const int t = to0 * g.SH - g.PH + r;                                                                                    \
    rows[i] = (t >= 0 && t < g.Ti && f >= 0 && f < g.Fi) ? x[(long long)b * sb + (long long)t * st + f] : 0.f;              \
  }                                                                                                                         \
  __syncthreads();

template <typename T, int KKT>
__global__ __launch_bounds__(256) void conv1_apply_kernel(const float* x, long long sb, long long st, Geo g, const float* w, int C,
                                                          const float* scale, const float* shift, int act, float slope, T* y,
                                                          Tiling tl) {
  extern __shared__ float rows[];  // [(R-1)*SH + KH][W]
  SC_CONV1_THREAD_SETUP(1024)
  const int ntiles = g.B * tl.tiles_per_b;
  const int nin = (tl.R - 1) * g.SH + g.KH;
  for (int tile = blockIdx.x * tl.tiles_per_group; tile < (blockIdx.x + 1) * tl.tiles_per_group && tile < ntiles; ++tile) {
    SC_CONV1_STAGE_TILE()
    if (!live) continue;
    const long long r = (long long)b * g.To + to0;  // first output row of the tile
    for (int pos = lane; pos < npos; pos += lanes) {
      const int ro = pos / g.Fo, fo = pos - ro * g.Fo;
      const int origin = ro * g.SH * W + fo * g.SW;
      float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NT; ++k)
        if (k < KK) {
          const float p = rows[toff[k] + origin];
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = fmaf(p, wr[k][i], z[i]);
        }
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pact(fmaf(sc[i], z[i], sh[i]), act, slope);
      st4<T>(y + (r * g.Fo + pos) * (long long)C + c, o, vec, nc);
    }
  }
}

// First block, backward: partial[group][c][KK + 1] = (sum_pos g * patch_k (k < KK), sum_pos g) with g = dy * act'(scale z + shift),
// z recomputed from the staged rows.  Same work split as conv1_apply_kernel (256-channel slabs); the lanes of a quad meet in LDS.
template <typename T, int KKT>
__global__ __launch_bounds__(256) void conv1_bwd_reduce_kernel(const float* x, long long sb, long long st, Geo g, const float* w, int C,
                                                               const float* scale, const float* shift, int act, float slope,
                                                               const T* dy, Tiling tl, float* partial) {
  extern __shared__ float rows[];  // [(R-1)*SH + KH][W], then [lanes][CSp] scratch for the final reduction
  SC_CONV1_THREAD_SETUP(256)
  const int ntiles = g.B * tl.tiles_per_b;
  const int nin = (tl.R - 1) * g.SH + g.KH;
  float* red = rows + nin * W;
  float G[NT + 1][4];
#pragma unroll
  for (int k = 0; k <= NT; ++k) G[k][0] = G[k][1] = G[k][2] = G[k][3] = 0.f;
  for (int tile = blockIdx.x * tl.tiles_per_group; tile < (blockIdx.x + 1) * tl.tiles_per_group && tile < ntiles; ++tile) {
    SC_CONV1_STAGE_TILE()
    if (!live) continue;
    const long long r = (long long)b * g.To + to0;
    // four positions per trip, their dy loads issued before any arithmetic: with ~170 VGPRs only two waves share a SIMD, so a
    // load that is waited for immediately costs a full memory round trip per position
    for (int pos0 = lane; pos0 < npos; pos0 += 4 * lanes) {
      float d[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pos = pos0 + u * lanes;
        if (pos < npos) {
          ld4<T>(dy + (r * g.Fo + pos) * (long long)C + c, d[u], vec, nc);
        } else {
          d[u][0] = d[u][1] = d[u][2] = d[u][3] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pos = min(pos0 + u * lanes, npos - 1);  // (a clamped position contributes d = 0)
        const int ro = pos / g.Fo, fo = pos - ro * g.Fo;
        const int origin = ro * g.SH * W + fo * g.SW;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NT; ++k)
          if (k < KK) {
            const float p = rows[toff[k] + origin];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = fmaf(p, wr[k][i], z[i]);
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[u][i] *= pact_grad(fmaf(sc[i], z[i], sh[i]), act, slope);
#pragma unroll
        for (int k = 0; k < NT; ++k)
          if (k < KK) {
            const float p = rows[toff[k] + origin];
#pragma unroll
            for (int i = 0; i < 4; ++i) G[k][i] = fmaf(d[u][i], p, G[k][i]);
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) G[NT][i] += d[u][i];
      }
    }
  }
  // reduce the lanes of every quad through LDS, one tap at a time (fixed order)
  for (int k = 0; k <= KK; ++k) {
    const int kk = k < KK ? k : NT;
    __syncthreads();
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q <= NT; ++q)
          if (q == kk) v = G[q][i];
        red[lane * CSp + cq * 4 + i] = v;
      }
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < CS; cc += blockDim.x) {
      float s = 0.f;
      for (int l = 0; l < lanes; ++l) s += red[l * CSp + cc];
      partial[((long long)blockIdx.x * C + c0 + cc) * (KK + 1) + k] = s;
    }
  }
}

// dW[c][k], dgamma, dbeta of the first block from the reduced G, sum g and the forward moments (see the file header).
// Gsum: [C][KK + 1] float64 (sum_partials_kernel over the groups).
// Synchronised statistics (torch.nn.SyncBatchNorm: mean / variance over the batches of ALL ranks): `Gglob` holds the sums of all ranks and
// `count_dev` their position count; mean / rstd are the global ones.  What this rank contributes to the parameter gradients uses ITS sums
// (G, S, P: the gradient all-reduce averages the ranks afterwards), the terms that come from differentiating the statistics use the global
// ones -- dz = gamma rstd (g - sum_all g / N_all - xhat sum_all(g xhat) / N_all) for this rank's positions.
__global__ void conv1_bwd_finish_kernel(const double* Gsum, const double* Gglob, const double* moments, const float* w, int C, int KK,
                                        double count, const double* count_dev, const float* gamma, const float* mean, const float* rstd,
                                        int batch_stats, float* dw, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;
  const double* G = Gsum + (long long)c * (KK + 1);
  const double* Gg = (Gglob ? Gglob : Gsum) + (long long)c * (KK + 1);
  const double sg = G[KK], sg_all = Gg[KK];
  if (!mean) {  // no BatchNorm: z = w.p + bias
    for (int k = 0; k < KK; ++k) dw[c * KK + k] = (float)G[k];
    if (dbeta) dbeta[c] = (float)sg;  // the bias gradient
    return;
  }
  double sgz = 0.0, sgz_all = 0.0;
  for (int k = 0; k < KK; ++k) {
    sgz += (double)w[c * KK + k] * G[k];
    sgz_all += (double)w[c * KK + k] * Gg[k];
  }
  const double m = mean[c], rs = rstd[c], ga = gamma ? (double)gamma[c] : 1.0;
  const double dg = rs * (sgz - m * sg);              // sum g * xhat over this rank's positions
  const double dg_all = rs * (sgz_all - m * sg_all);  // ... over all ranks'
  if (dgamma) dgamma[c] = (float)dg;
  if (dbeta) dbeta[c] = (float)sg;
  const double* S = moments;
  const double* P = moments + KK * KK;
  for (int k = 0; k < KK; ++k) {
    double v = G[k];
    if (batch_stats) {
      double swS = 0.0;
      for (int i = 0; i < KK; ++i) swS += (double)w[c * KK + i] * S[i * KK + k];
      const double sxp = rs * (swS - m * P[k]);  // sum_pos xhat * p_k (this rank's positions)
      v -= (sg_all / count) * P[k] + (dg_all / count) * sxp;
    }
    dw[c * KK + k] = (float)(ga * rs * v);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Blocks with C_in > 1: im2col / col2im on channels-last images.  col[(b, to, fo)][(kh, kw, c)] = a[b][to*SH-PH+kh][fo*SW-PW+kw][c].
// One thread moves VEC consecutive channels (16 bytes) per step.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const T* a, Geo g, int C, T* col) {
  const int CV = C / VEC;
  const long long KK = (long long)g.KH * g.KW;
  const long long total = (long long)g.B * g.To * g.Fo * KK * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int k = (int)(r % KK);
    r /= KK;
    const int fo = (int)(r % g.Fo);
    r /= g.Fo;
    const int to = (int)(r % g.To), b = (int)(r / g.To);
    const int t = to * g.SH - g.PH + k / g.KW, f = fo * g.SW - g.PW + k % g.KW;
    T* dst = col + i * VEC;
    const bool in = t >= 0 && t < g.Ti && f >= 0 && f < g.Fi;
    const T* src = a + (((long long)b * g.Ti + t) * g.Fi + f) * C + cv * VEC;
    if constexpr (VEC * sizeof(T) == 16) {
      *reinterpret_cast<uint4*>(dst) = in ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[e] = in ? src[e] : (T)0;
    }
  }
}

// da[b][t][f][c] = sum over the taps (kh, kw) that reach (t, f): dcol[(b, (t+PH-kh)/SH, (f+PW-kw)/SW)][(kh, kw, c)]
// bf16 with C % 8 == 0: 8 channels (16 bytes) per thread
__global__ __launch_bounds__(256) void col2im_bf16x8_kernel(const unsigned short* dcol, Geo g, int C, unsigned short* da) {
  const int CV = C >> 3;
  const long long KKC = (long long)g.KH * g.KW * C;
  const long long total = (long long)g.B * g.Ti * g.Fi * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int f = (int)(r % g.Fi);
    r /= g.Fi;
    const int t = (int)(r % g.Ti), b = (int)(r / g.Ti);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < g.KH; ++kh) {
      const int tt = t + g.PH - kh;
      if (tt < 0 || tt % g.SH) continue;
      const int to = tt / g.SH;
      if (to >= g.To) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int ff = f + g.PW - kw;
        if (ff < 0 || ff % g.SW) continue;
        const int fo = ff / g.SW;
        if (fo >= g.Fo) continue;
        const uint4 u = *reinterpret_cast<const uint4*>(dcol + (((long long)b * g.To + to) * g.Fo + fo) * KKC + (long long)(kh * g.KW + kw) * C + cv * 8);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += __uint_as_float(w[j] << 16);
          acc[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u);
        }
      }
    }
    uint4 o;
    o.x = (unsigned)sc_f32_to_bf16(acc[0]) | ((unsigned)sc_f32_to_bf16(acc[1]) << 16);
    o.y = (unsigned)sc_f32_to_bf16(acc[2]) | ((unsigned)sc_f32_to_bf16(acc[3]) << 16);
    o.z = (unsigned)sc_f32_to_bf16(acc[4]) | ((unsigned)sc_f32_to_bf16(acc[5]) << 16);
    o.w = (unsigned)sc_f32_to_bf16(acc[6]) | ((unsigned)sc_f32_to_bf16(acc[7]) << 16);
    *reinterpret_cast<uint4*>(da + (((long long)b * g.Ti + t) * g.Fi + f) * C + cv * 8) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* dcol, Geo g, int C, T* da) {
  const int CV = (C + 3) / 4;
  const bool vec = (C % 4 == 0);
  const long long KKC = (long long)g.KH * g.KW * C;
  const long long total = (long long)g.B * g.Ti * g.Fi * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int f = (int)(r % g.Fi);
    r /= g.Fi;
    const int t = (int)(r % g.Ti), b = (int)(r / g.Ti);
    const int c = cv * 4, nc = min(4, C - c);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < g.KH; ++kh) {
      const int tt = t + g.PH - kh;
      if (tt < 0 || tt % g.SH) continue;
      const int to = tt / g.SH;
      if (to >= g.To) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int ff = f + g.PW - kw;
        if (ff < 0 || ff % g.SW) continue;
        const int fo = ff / g.SW;
        if (fo >= g.Fo) continue;
        float v[4];
        ld4<T>(dcol + (((long long)b * g.To + to) * g.Fo + fo) * KKC + (long long)(kh * g.KW + kw) * C + c, v, vec, nc);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
      }
    }
    st4<T>(da + (((long long)b * g.Ti + t) * g.Fi + f) * C + c, acc, vec, nc);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BatchNorm over the rows of an (M, C) matrix + activation.
// Two-stage column reductions: 256 threads = 32 column quads x 8 row lanes; grid (column slabs of 128, row slabs).
//   MODE 0 (forward statistics):  partial[slab][0][c] = sum z,      [1][c] = sum z^2
//   MODE 1 (backward):            partial[slab][0][c] = sum g,      [1][c] = sum g*z,   g = dy * act'(scale_c z + shift_c)
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_kernel(const T* z, const T* dy, long long M, int C, const float* scale,
                                                           const float* shift, int act, float slope, long long rows_per_slab,
                                                           float* partial) {
  __shared__ float red[2][8][128];
  const int qx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 128 + qx * 4;
  const int nc = max(0, min(4, C - c));
  const bool vec = (C % 4 == 0);
  float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nc) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
  }
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  const long long r0 = (long long)blockIdx.y * rows_per_slab;
  const long long r1 = min(M, r0 + rows_per_slab);
  if (nc > 0)
    for (long long r = r0 + ry; r < r1; r += 8) {
      float v[4];
      ld4<T>(z + r * C + c, v, vec, nc);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0[i] += v[i]; s1[i] = fmaf(v[i], v[i], s1[i]); }
      } else {
        float d[4];
        ld4<T>(dy + r * C + c, d, vec, nc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gg = d[i] * pact_grad(fmaf(sc[i], v[i], sh[i]), act, slope);
          s0[i] += gg;
          s1[i] = fmaf(gg, v[i], s1[i]);
        }
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][ry][qx * 4 + i] = s0[i]; red[1][ry][qx * 4 + i] = s1[i]; }
  __syncthreads();
  const int which = threadIdx.x >> 7, cc = threadIdx.x & 127;
  if (blockIdx.x * 128 + cc < C) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) s += red[which][l][cc];
    partial[((long long)blockIdx.y * 2 + which) * C + blockIdx.x * 128 + cc] = s;
  }
}

// forward finish from sums[0][c] = sum z, sums[1][c] = sum z^2 (float64): batch mean / biased variance -> scale, shift, saved
// mean / rstd; running statistics updated in place
__global__ void bn_stats_finish_kernel(const double* sums, int C, double count, const double* count_dev, const float* gamma,
                                       const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                       float* shift, float* mean, float* rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;
  const double m = sums[c] / count;
  double var = sums[C + c] / count - m * m;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(ga * rs);
  shift[c] = (float)(be - m * ga * rs);
  mean[c] = (float)m;
  rstd[c] = (float)rs;
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
  if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (count > 1.0 ? count / (count - 1.0) : 1.0));
}

// backward finish from sums[0][c] = sum g, sums[1][c] = sum g*z: dgamma = sum g*xhat, dbeta = sum g and the per-column coefficients of
// dz = ca*g + cb*z + cc
//   batch statistics: dz = gamma*rstd * (g - sum g / N - xhat * dgamma / N);  fixed (running) statistics or no BatchNorm: dz = scale * g
//   synchronised statistics: dgamma / dbeta from THIS rank's sums (`sums`), the coefficients from the sums of all ranks (`sums_all`, `count_dev`)
__global__ void bn_bwd_finish_kernel(const double* sums, const double* sums_all, int C, double count, const double* count_dev,
                                     const float* scale, const float* mean, const float* rstd, int batch_stats, float* dgamma, float* dbeta,
                                     float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;
  if (!sums_all) sums_all = sums;
  const double sg = sums[c], sgz = sums[C + c];
  const double sg_all = sums_all[c], sgz_all = sums_all[C + c];
  const double a = (double)scale[c];  // gamma * rstd (or 1 without BatchNorm)
  double dg_all = 0.0;
  if (mean) {
    const double m = mean[c], rs = rstd[c];
    dg_all = rs * (sgz_all - m * sg_all);
    if (dgamma) dgamma[c] = (float)(rs * (sgz - m * sg));
  }
  if (dbeta) dbeta[c] = (float)sg;
  double cb = 0.0, cc = 0.0;
  if (mean && batch_stats) {
    const double m = mean[c], rs = rstd[c];
    cb = -a * rs * dg_all / count;
    cc = -a * sg_all / count - cb * m;
  }
  coef[c] = (float)a;
  coef[C + c] = (float)cb;
  coef[2 * C + c] = (float)cc;
}

// MODE 0: y = act(scale_c z + shift_c);  MODE 1: dz = ca_c * g + cb_c * z + cc_c with g = dy * act'(scale_c z + shift_c)
// A thread owns one column quad -- its per-column coefficients stay in registers -- and walks rows: 256 threads = CV quads x 256/CV
// rows in flight (the first version re-read up to five coefficients per element from memory: 354 us for 465 MB at C = 512).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* z, const T* dy, long long M, int C, const float* scale, const float* shift,
                                                       const float* coef, int act, float slope, T* out, long long rows_per_block) {
  const int CV = (C + 3) / 4;
  const bool vec = (C % 4 == 0);
  const int lanes = max(1, 256 / min(CV, 256));
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  for (int cv = threadIdx.x % min(CV, 256); cv < CV; cv += 256) {  // (one pass unless C > 1024)
    const int rl = threadIdx.x / min(CV, 256);
    if (rl >= lanes) continue;
    const int c = cv * 4, nc = min(4, C - c);
    float sc[4], sh[4], ca[4], cb[4], cc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = e < nc;
      sc[e] = in ? scale[c + e] : 0.f;
      sh[e] = in ? shift[c + e] : 0.f;
      ca[e] = (MODE == 1 && in) ? coef[c + e] : 0.f;
      cb[e] = (MODE == 1 && in) ? coef[C + c + e] : 0.f;
      cc[e] = (MODE == 1 && in) ? coef[2 * C + c + e] : 0.f;
    }
    for (long long r = r0 + rl; r < r1; r += lanes) {
      float v[4], o[4];
      ld4<T>(z + r * C + c, v, vec, nc);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pact(fmaf(sc[e], v[e], sh[e]), act, slope);
      } else {
        float d[4];
        ld4<T>(dy + r * C + c, d, vec, nc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = d[e] * pact_grad(fmaf(sc[e], v[e], sh[e]), act, slope);
          o[e] = fmaf(ca[e], gg, fmaf(cb[e], v[e], cc[e]));
        }
      }
      st4<T>(out + r * C + c, o, vec, nc);
    }
  }
}

int geo_ok(const Geo& g) {
  return g.B >= 1 && g.Ti >= 1 && g.Fi >= 1 && g.To >= 1 && g.Fo >= 1 && g.KH >= 1 && g.KW >= 1 && g.SH >= 1 && g.SW >= 1 && g.PH >= 0 &&
         g.PW >= 0 && g.To == (g.Ti + 2 * g.PH - g.KH) / g.SH + 1 && g.Fo == (g.Fi + 2 * g.PW - g.KW) / g.SW + 1 &&
         g.Ti + 2 * g.PH >= g.KH && g.Fi + 2 * g.PW >= g.KW;
}

Tiling tiling(const Geo& g) {
  Tiling t;
  t.R = std::min(8, g.To);
  while (t.R > 1 && (size_t)((t.R - 1) * g.SH + g.KH) * (g.Fi + 2 * g.PW) * sizeof(float) > 32768) --t.R;
  t.tiles_per_b = (g.To + t.R - 1) / t.R;
  const long long tiles = (long long)g.B * t.tiles_per_b;
  t.tiles_per_group = (int)std::max(1LL, (tiles + 1023) / 1024);  // <= 1024 workgroups (4 per CU), <= 1024 partial rows
  t.groups = (int)((tiles + t.tiles_per_group - 1) / t.tiles_per_group);
  return t;
}

size_t tile_lds(const Geo& g, const Tiling& t) { return (size_t)((t.R - 1) * g.SH + g.KH) * (g.Fi + 2 * g.PW) * sizeof(float); }

Geo geo_of(int B, int Ti, int Fi, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW) { return Geo{B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW}; }

int col_slabs(long long M, long long* rows_per_slab) {
  long long slabs = (M + 63) / 64;  // >= 64 rows (8 per row lane) per slab
  if (slabs > 512) slabs = 512;
  if (slabs < 1) slabs = 1;
  *rows_per_slab = (M + slabs - 1) / slabs;
  return (int)((M + *rows_per_slab - 1) / *rows_per_slab);
}

// bn_apply_kernel: rows per workgroup so that ~4096 workgroups cover M rows (at least as many rows as are in flight per pass)
unsigned apply_grid(long long M, int C, long long* rows_per_block) {
  const int CV = (C + 3) / 4;
  const int lanes = std::max(1, 256 / std::min(CV, 256));
  long long rpb = (M + 4095) / 4096;
  rpb = (rpb + lanes - 1) / lanes * lanes;
  if (rpb < 4LL * lanes) rpb = 4LL * lanes;
  *rows_per_block = rpb;
  return (unsigned)((M + rpb - 1) / rpb);
}

unsigned flat_grid(long long items) {
  long long blocks = (items + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

#define SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW) \
  Geo g{B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW};       \
  SC_CHECK_ARG(geo_ok(g), "%s: inconsistent geometry B=%d in=%dx%d out=%dx%d k=%dx%d s=%dx%d p=%dx%d", __func__, B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW)

// ---- first block -------------------------------------------------------------------------------------------------------------------
// (workspace sizes depend on the tiling, which depends on the whole geometry: the *_workspace functions take it all)
SC_API long long sc_conv1_moments_workspace(int B, int Ti, int Fi, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW) {
  const Geo g = geo_of(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  if (!geo_ok(g)) return 0;
  const long long NE = (long long)KH * KW * KH * KW + KH * KW;
  return tiling(g).groups * NE * (long long)sizeof(double);
}

SC_API int sc_conv1_moments(const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH, int KW,
                            int SH, int SW, int PH, int PW, double* moments, void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(x && moments && workspace, "sc_conv1_moments: null pointer");
  SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  const int KK = KH * KW, NE = KK * KK + KK;
  SC_CHECK_ARG(KK <= MAX_TAPS, "sc_conv1_moments: at most %d taps (KH*KW = %d): one thread per entry of the moment matrix", MAX_TAPS, KK);
  SC_CHECK_ARG(stride_t >= Fi && stride_b >= (long long)Ti * stride_t, "sc_conv1_moments: strides too small");
  const Tiling tl = tiling(g);
  SC_CHECK_ARG(ws_bytes >= (long long)tl.groups * NE * (long long)sizeof(double), "sc_conv1_moments: workspace too small");
  const size_t lds = tile_lds(g, tl);
  SC_CHECK_ARG(lds <= 60000, "sc_conv1_moments: %d input rows of %d values do not fit LDS", KH, Fi + 2 * PW);
  hipLaunchKernelGGL(conv1_moments_kernel, dim3(tl.groups), dim3(256), lds, stream, x, stride_b, stride_t, g, tl, (double*)workspace);
  SC_LAUNCH_CHECK("conv1_moments_kernel");
  hipLaunchKernelGGL(sum_partials_kernel<double>, dim3(sc_cdiv(NE, 32)), dim3(1024), 0, stream, (const double*)workspace, tl.groups,
                     (long long)NE, moments);
  SC_LAUNCH_CHECK("sum_partials_kernel");
  return SC_OK;
}

SC_API int sc_conv1_bn_coeffs(const double* moments, const float* w, int C, int KK, long long count, const double* count_dev,
                              const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* scale, float* shift, float* mean, float* rstd, hipStream_t stream) {
  SC_CHECK_ARG(moments && w && scale && shift && mean && rstd, "sc_conv1_bn_coeffs: null pointer");
  SC_CHECK_ARG(C >= 1 && KK >= 1 && KK <= MAX_TAPS && count >= 1, "sc_conv1_bn_coeffs: bad sizes C=%d KK=%d count=%lld", C, KK, count);
  hipLaunchKernelGGL(conv1_bn_coeffs_kernel, dim3(sc_cdiv(C, 128)), dim3(128), 0, stream, moments, w, C, KK, (double)count, count_dev, gamma,
                     beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd);
  SC_LAUNCH_CHECK("conv1_bn_coeffs_kernel");
  return SC_OK;
}

SC_API int sc_conv1_apply(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH,
                          int KW, int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act,
                          float slope, void* y, hipStream_t stream) {
  SC_CHECK_ARG(x && w && scale && shift && y, "sc_conv1_apply: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_conv1_apply: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(act >= 0 && act <= 2 && C >= 1, "sc_conv1_apply: bad act %d / C %d", act, C);
  SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  SC_CHECK_ARG(KH * KW <= MAX_TAPS, "sc_conv1_apply: at most %d taps", MAX_TAPS);
  SC_CHECK_ARG(stride_t >= Fi && stride_b >= (long long)Ti * stride_t, "sc_conv1_apply: strides too small");
  const Tiling tl = tiling(g);
  const size_t lds = tile_lds(g, tl);
  SC_CHECK_ARG(lds <= 60000, "sc_conv1_apply: %d input rows of %d values do not fit LDS", KH, Fi + 2 * PW);
  const dim3 grid(tl.groups, sc_cdiv(C, 1024));
#define SC_LAUNCH_APPLY(T, KKT) \
  hipLaunchKernelGGL((conv1_apply_kernel<T, KKT>), grid, dim3(256), lds, stream, x, stride_b, stride_t, g, w, C, scale, shift, act, slope, (T*)y, tl)
  if (dtype == 0) {
    if (KH * KW == 9) SC_LAUNCH_APPLY(float, 9); else SC_LAUNCH_APPLY(float, 0);
  } else {
    if (KH * KW == 9) SC_LAUNCH_APPLY(unsigned short, 9); else SC_LAUNCH_APPLY(unsigned short, 0);
  }
#undef SC_LAUNCH_APPLY
  SC_LAUNCH_CHECK("conv1_apply_kernel");
  return SC_OK;
}

SC_API long long sc_conv1_bwd_workspace(int B, int Ti, int Fi, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW, int C) {
  const Geo g = geo_of(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  if (!geo_ok(g)) return 0;
  const long long n = (long long)C * (KH * KW + 1);
  return n * (long long)sizeof(double) + (long long)tiling(g).groups * n * (long long)sizeof(float);
}

SC_API int sc_conv1_bwd_reduce(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH,
                               int KW, int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act,
                               float slope, const void* dy, double* gsum, void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(x && w && scale && shift && dy && gsum && workspace, "sc_conv1_bwd_reduce: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_conv1_bwd_reduce: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(act >= 0 && act <= 2 && C >= 1, "sc_conv1_bwd_reduce: bad act %d / C %d", act, C);
  SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  const int KK = KH * KW;
  SC_CHECK_ARG(KK <= MAX_TAPS, "sc_conv1_bwd_reduce: at most %d taps", MAX_TAPS);
  SC_CHECK_ARG(stride_t >= Fi && stride_b >= (long long)Ti * stride_t, "sc_conv1_bwd_reduce: strides too small");
  const Tiling tl = tiling(g);
  const long long n = (long long)C * (KK + 1);
  SC_CHECK_ARG(ws_bytes >= (long long)tl.groups * n * (long long)sizeof(float), "sc_conv1_bwd_reduce: workspace too small");
  SC_CHECK_ARG((uintptr_t)workspace % 4 == 0 && (uintptr_t)gsum % 8 == 0, "sc_conv1_bwd_reduce: misaligned workspace / gsum");
  float* partial = (float*)workspace;
  const int CSp = (std::min(C, 256) + 3) & ~3;
  const int lanes = std::max(1, 256 / (CSp / 4));
  const size_t lds = tile_lds(g, tl) + (size_t)lanes * CSp * sizeof(float);
  SC_CHECK_ARG(lds <= 60000, "sc_conv1_bwd_reduce: input rows + reduction scratch (%zu bytes) do not fit LDS", lds);
  const dim3 grid(tl.groups, sc_cdiv(C, 256));
#define SC_LAUNCH_REDUCE(T, KKT)                                                                                                          \
  hipLaunchKernelGGL((conv1_bwd_reduce_kernel<T, KKT>), grid, dim3(256), lds, stream, x, stride_b, stride_t, g, w, C, scale, shift, act, slope, \
                     (const T*)dy, tl, partial)
  if (dtype == 0) {
    if (KK == 9) SC_LAUNCH_REDUCE(float, 9); else SC_LAUNCH_REDUCE(float, 0);
  } else {
    if (KK == 9) SC_LAUNCH_REDUCE(unsigned short, 9); else SC_LAUNCH_REDUCE(unsigned short, 0);
  }
#undef SC_LAUNCH_REDUCE
  SC_LAUNCH_CHECK("conv1_bwd_reduce_kernel");
  hipLaunchKernelGGL(sum_partials_kernel<float>, dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, stream, (const float*)partial, tl.groups, n, gsum);
  SC_LAUNCH_CHECK("sum_partials_kernel");
  return SC_OK;
}

SC_API int sc_conv1_bwd_finish(const double* gsum, const double* gsum_all, const double* moments, const float* w, int C, int KK, long long count,
                               const double* count_dev, const float* gamma, const float* mean, const float* rstd, int batch_stats, float* dw,
                               float* dgamma, float* dbeta, hipStream_t stream) {
  SC_CHECK_ARG(gsum && w && dw, "sc_conv1_bwd_finish: null pointer");
  SC_CHECK_ARG(C >= 1 && KK >= 1 && KK <= MAX_TAPS && count >= 1, "sc_conv1_bwd_finish: bad sizes C=%d KK=%d count=%lld", C, KK, count);
  SC_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "sc_conv1_bwd_finish: mean and rstd come together");
  SC_CHECK_ARG(!(mean && batch_stats) || moments, "sc_conv1_bwd_finish: batch statistics need the forward moments");
  SC_CHECK_ARG((gsum_all == nullptr) == (count_dev == nullptr), "sc_conv1_bwd_finish: the sums of all ranks and their count come together");
  hipLaunchKernelGGL(conv1_bwd_finish_kernel, dim3(sc_cdiv(C, 64)), dim3(64), 0, stream, gsum, gsum_all, moments, w, C, KK, (double)count,
                     count_dev, gamma, mean, rstd, batch_stats, dw, dgamma, dbeta);
  SC_LAUNCH_CHECK("conv1_bwd_finish_kernel");
  return SC_OK;
}

SC_API int sc_conv1_bwd(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH, int KW,
                        int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act, float slope,
                        const void* dy, const double* moments, const float* gamma, const float* mean, const float* rstd, int batch_stats,
                        float* dw, float* dgamma, float* dbeta, void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(dw && workspace, "sc_conv1_bwd: null pointer");
  SC_CHECK_ARG(ws_bytes >= sc_conv1_bwd_workspace(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW, C) && ws_bytes > 0, "sc_conv1_bwd: workspace too small");
  SC_CHECK_ARG((uintptr_t)workspace % 8 == 0, "sc_conv1_bwd: workspace must be 8-byte aligned");
  const long long n = (long long)C * (KH * KW + 1);
  double* gsum = (double*)workspace;
  int rc = sc_conv1_bwd_reduce(dtype, x, stride_b, stride_t, B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW, w, C, scale, shift, act, slope, dy, gsum,
                               gsum + n, ws_bytes - n * (long long)sizeof(double), stream);
  if (rc != SC_OK) return rc;
  return sc_conv1_bwd_finish(gsum, nullptr, moments, w, C, KH * KW, (long long)B * To * Fo, nullptr, gamma, mean, rstd, batch_stats, dw, dgamma,
                             dbeta, stream);
}

// ---- blocks with C_in > 1 ----------------------------------------------------------------------------------------------------------
SC_API int sc_im2col(int dtype, const void* a, int B, int Ti, int Fi, int C, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW,
                     void* col, hipStream_t stream) {
  SC_CHECK_ARG(a && col, "sc_im2col: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_im2col: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(C >= 1, "sc_im2col: C=%d", C);
  SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  const long long items = (long long)B * To * Fo * KH * KW;
  const bool al = ((uintptr_t)a % 16 == 0) && ((uintptr_t)col % 16 == 0);
  if (dtype == 0) {
    if (C % 4 == 0 && al)
      hipLaunchKernelGGL((im2col_kernel<float, 4>), dim3(flat_grid(items * (C / 4))), dim3(256), 0, stream, (const float*)a, g, C, (float*)col);
    else
      hipLaunchKernelGGL((im2col_kernel<float, 1>), dim3(flat_grid(items * C)), dim3(256), 0, stream, (const float*)a, g, C, (float*)col);
  } else {
    if (C % 8 == 0 && al)
      hipLaunchKernelGGL((im2col_kernel<unsigned short, 8>), dim3(flat_grid(items * (C / 8))), dim3(256), 0, stream, (const unsigned short*)a,
                         g, C, (unsigned short*)col);
    else
      hipLaunchKernelGGL((im2col_kernel<unsigned short, 1>), dim3(flat_grid(items * C)), dim3(256), 0, stream, (const unsigned short*)a, g, C,
                         (unsigned short*)col);
  }
  SC_LAUNCH_CHECK("im2col_kernel");
  return SC_OK;
}

SC_API int sc_col2im(int dtype, const void* dcol, int B, int Ti, int Fi, int C, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW,
                     void* da, hipStream_t stream) {
  SC_CHECK_ARG(dcol && da, "sc_col2im: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_col2im: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(C >= 1, "sc_col2im: C=%d", C);
  SC_GEO(B, Ti, Fi, To, Fo, KH, KW, SH, SW, PH, PW);
  SC_CHECK_ARG(((uintptr_t)dcol % 16 == 0) && ((uintptr_t)da % 16 == 0), "sc_col2im: buffers must be 16-byte aligned");
  const long long items = (long long)B * Ti * Fi * ((C + 3) / 4);
  if (dtype == 0)
    hipLaunchKernelGGL(col2im_kernel<float>, dim3(flat_grid(items)), dim3(256), 0, stream, (const float*)dcol, g, C, (float*)da);
  else if (C % 8 == 0)
    hipLaunchKernelGGL(col2im_bf16x8_kernel, dim3(flat_grid((long long)B * Ti * Fi * (C / 8))), dim3(256), 0, stream, (const unsigned short*)dcol, g, C,
                       (unsigned short*)da);
  else
    hipLaunchKernelGGL(col2im_kernel<unsigned short>, dim3(flat_grid(items)), dim3(256), 0, stream, (const unsigned short*)dcol, g, C,
                       (unsigned short*)da);
  SC_LAUNCH_CHECK("col2im_kernel");
  return SC_OK;
}

// ---- BatchNorm over the rows of an (M, C) matrix + activation ------------------------------------------------------------------------
SC_API long long sc_bn_workspace(long long M, int C) {  // [2][C] float64 sums, then [slabs][2][C] float partials
  long long rps;
  const int slabs = col_slabs(M, &rps);
  return 2LL * C * (long long)sizeof(double) + (long long)slabs * 2 * C * (long long)sizeof(float);
}

SC_API int sc_bn_sums(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, int act, float slope,
                      double* sums, void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(z && sums && workspace, "sc_bn_sums: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_bn_sums: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(M >= 1 && C >= 1 && act >= 0 && act <= 3, "sc_bn_sums: bad sizes M=%lld C=%d act=%d", M, C, act);
  SC_CHECK_ARG(!dy || (scale && shift), "sc_bn_sums: the backward sums need scale and shift");
  SC_CHECK_ARG((uintptr_t)z % 16 == 0 && (!dy || (uintptr_t)dy % 16 == 0), "sc_bn_sums: z / dy must be 16-byte aligned");
  long long rps;
  const int slabs = col_slabs(M, &rps);
  SC_CHECK_ARG(ws_bytes >= (long long)slabs * 2 * C * (long long)sizeof(float), "sc_bn_sums: workspace too small");
  SC_CHECK_ARG((uintptr_t)workspace % 4 == 0 && (uintptr_t)sums % 8 == 0, "sc_bn_sums: misaligned workspace / sums");
  float* partial = (float*)workspace;
  const dim3 grid(sc_cdiv(C, 128), slabs);
  if (!dy) {
    if (dtype == 0)
      hipLaunchKernelGGL((bn_colreduce_kernel<float, 0>), grid, dim3(256), 0, stream, (const float*)z, (const float*)nullptr, M, C, nullptr,
                         nullptr, 0, 0.f, rps, partial);
    else
      hipLaunchKernelGGL((bn_colreduce_kernel<unsigned short, 0>), grid, dim3(256), 0, stream, (const unsigned short*)z,
                         (const unsigned short*)nullptr, M, C, nullptr, nullptr, 0, 0.f, rps, partial);
    SC_LAUNCH_CHECK("bn_colreduce_kernel<0>");
  } else {
    if (dtype == 0)
      hipLaunchKernelGGL((bn_colreduce_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)z, (const float*)dy, M, C, scale, shift, act,
                         slope, rps, partial);
    else
      hipLaunchKernelGGL((bn_colreduce_kernel<unsigned short, 1>), grid, dim3(256), 0, stream, (const unsigned short*)z, (const unsigned short*)dy,
                         M, C, scale, shift, act, slope, rps, partial);
    SC_LAUNCH_CHECK("bn_colreduce_kernel<1>");
  }
  hipLaunchKernelGGL(sum_partials_kernel<float>, dim3(sc_cdiv(2 * C, 32)), dim3(1024), 0, stream, (const float*)partial, slabs, 2LL * C, sums);
  SC_LAUNCH_CHECK("sum_partials_kernel");
  return SC_OK;
}

SC_API int sc_bn_stats_finish(const double* sums, int C, long long count, const double* count_dev, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd,
                              hipStream_t stream) {
  SC_CHECK_ARG(sums && scale && shift && mean && rstd, "sc_bn_stats_finish: null pointer");
  SC_CHECK_ARG(C >= 1 && count >= 1, "sc_bn_stats_finish: bad sizes C=%d count=%lld", C, count);
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(sc_cdiv(C, 128)), dim3(128), 0, stream, sums, C, (double)count, count_dev, gamma, beta, eps,
                     momentum, running_mean, running_var, scale, shift, mean, rstd);
  SC_LAUNCH_CHECK("bn_stats_finish_kernel");
  return SC_OK;
}

SC_API int sc_bn_stats(int dtype, const void* z, long long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd, void* workspace,
                       long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(workspace, "sc_bn_stats: null pointer");
  SC_CHECK_ARG(ws_bytes >= sc_bn_workspace(M, C) && M >= 1 && C >= 1, "sc_bn_stats: workspace too small");
  SC_CHECK_ARG((uintptr_t)workspace % 8 == 0, "sc_bn_stats: workspace must be 8-byte aligned");
  double* sums = (double*)workspace;
  int rc = sc_bn_sums(dtype, z, nullptr, M, C, nullptr, nullptr, 0, 0.f, sums, sums + 2LL * C, ws_bytes - 2LL * C * (long long)sizeof(double), stream);
  if (rc != SC_OK) return rc;
  return sc_bn_stats_finish(sums, C, M, nullptr, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd, stream);
}

SC_API int sc_bn_act_fwd(int dtype, const void* z, long long M, int C, const float* scale, const float* shift, int act, float slope, void* y,
                         hipStream_t stream) {
  SC_CHECK_ARG(z && scale && shift && y, "sc_bn_act_fwd: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_bn_act_fwd: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(M >= 1 && C >= 1 && act >= 0 && act <= 3, "sc_bn_act_fwd: bad sizes M=%lld C=%d act=%d", M, C, act);
  SC_CHECK_ARG(((uintptr_t)z % 16 == 0) && ((uintptr_t)y % 16 == 0), "sc_bn_act_fwd: buffers must be 16-byte aligned");
  long long rpb;
  const unsigned grid = apply_grid(M, C, &rpb);
  if (dtype == 0)
    hipLaunchKernelGGL((bn_apply_kernel<float, 0>), dim3(grid), dim3(256), 0, stream, (const float*)z, (const float*)nullptr, M, C, scale, shift,
                       nullptr, act, slope, (float*)y, rpb);
  else
    hipLaunchKernelGGL((bn_apply_kernel<unsigned short, 0>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)z,
                       (const unsigned short*)nullptr, M, C, scale, shift, nullptr, act, slope, (unsigned short*)y, rpb);
  SC_LAUNCH_CHECK("bn_apply_kernel<0>");
  return SC_OK;
}

SC_API int sc_bn_bwd_finish(const double* sums, const double* sums_all, int C, long long count, const double* count_dev, const float* scale,
                            const float* mean, const float* rstd, int batch_stats, float* dgamma, float* dbeta, float* coef, hipStream_t stream) {
  SC_CHECK_ARG(sums && scale && coef, "sc_bn_bwd_finish: null pointer");
  SC_CHECK_ARG(C >= 1 && count >= 1, "sc_bn_bwd_finish: bad sizes C=%d count=%lld", C, count);
  SC_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "sc_bn_bwd_finish: mean and rstd come together");
  SC_CHECK_ARG((sums_all == nullptr) == (count_dev == nullptr), "sc_bn_bwd_finish: the sums of all ranks and their count come together");
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(sc_cdiv(C, 128)), dim3(128), 0, stream, sums, sums_all, C, (double)count, count_dev, scale, mean,
                     rstd, batch_stats, dgamma, dbeta, coef);
  SC_LAUNCH_CHECK("bn_bwd_finish_kernel");
  return SC_OK;
}

SC_API int sc_bn_bwd_apply(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, const float* coef,
                           int act, float slope, void* dz, hipStream_t stream) {
  SC_CHECK_ARG(z && dy && scale && shift && coef && dz, "sc_bn_bwd_apply: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_bn_bwd_apply: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(M >= 1 && C >= 1 && act >= 0 && act <= 3, "sc_bn_bwd_apply: bad sizes M=%lld C=%d act=%d", M, C, act);
  SC_CHECK_ARG(((uintptr_t)z % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dz % 16 == 0), "sc_bn_bwd_apply: buffers must be 16-byte aligned");
  long long rpb;
  const unsigned g2 = apply_grid(M, C, &rpb);
  if (dtype == 0)
    hipLaunchKernelGGL((bn_apply_kernel<float, 1>), dim3(g2), dim3(256), 0, stream, (const float*)z, (const float*)dy, M, C, scale, shift, coef,
                       act, slope, (float*)dz, rpb);
  else
    hipLaunchKernelGGL((bn_apply_kernel<unsigned short, 1>), dim3(g2), dim3(256), 0, stream, (const unsigned short*)z, (const unsigned short*)dy,
                       M, C, scale, shift, coef, act, slope, (unsigned short*)dz, rpb);
  SC_LAUNCH_CHECK("bn_apply_kernel<1>");
  return SC_OK;
}

SC_API int sc_bn_act_bwd(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, const float* mean,
                         const float* rstd, int batch_stats, int act, float slope, void* dz, float* dgamma, float* dbeta, void* workspace,
                         long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(workspace && M >= 1 && C >= 1, "sc_bn_act_bwd: null pointer / bad sizes");
  SC_CHECK_ARG(ws_bytes >= sc_bn_workspace(M, C) + 3LL * C * (long long)sizeof(float), "sc_bn_act_bwd: workspace too small");
  SC_CHECK_ARG((uintptr_t)workspace % 8 == 0, "sc_bn_act_bwd: workspace must be 8-byte aligned");
  long long rps;
  const int slabs = col_slabs(M, &rps);
  double* sums = (double*)workspace;
  float* partial = (float*)(sums + 2LL * C);
  float* coef = partial + (long long)slabs * 2 * C;
  int rc = sc_bn_sums(dtype, z, dy, M, C, scale, shift, act, slope, sums, partial, (long long)slabs * 2 * C * (long long)sizeof(float), stream);
  if (rc != SC_OK) return rc;
  rc = sc_bn_bwd_finish(sums, nullptr, C, M, nullptr, scale, mean, rstd, batch_stats, dgamma, dbeta, coef, stream);
  if (rc != SC_OK) return rc;
  return sc_bn_bwd_apply(dtype, z, dy, M, C, scale, shift, coef, act, slope, dz, stream);
}
