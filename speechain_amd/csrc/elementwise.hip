// Element-wise helpers of the transformer path for gfx950 (HBM-bound; 8 elements = 16/32 bytes per lane):
//   dtype casts (f32 master weights -> bf16 compute copies, plain and transposed)
//   positional-encoding add with scale / learnable alpha / dropout -- speechain/module/transformer/pos_enc.py:168-182
//   dropout apply (forward of residual dropout's backward: dy * keep / (1 - p)) -- encoder.py:119,134; decoder.py:126,150,166
#include "sc_common.h"
#include <algorithm>

namespace {

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct Io<unsigned short> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return sc_bf16_to_f32(((const unsigned short*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((unsigned short*)p)[i] = sc_f32_to_bf16(v); }
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const void* in, void* out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) Io<TO>::st(out, i, Io<TI>::ld(in, i));
}

// out[c][r] = in[r][c] for an (R, C) f32 matrix -> (C, R) matrix of TO, via a 32 x 33 LDS tile
template <typename TO>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* in, void* out, int R, int C, long long ld_out) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? in[(long long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) Io<TO>::st(out, (long long)c * ld_out + r, tile[tx][j]);
  }
}

// y[b][t][d] = dropout(x[b][t][d] * xscale + table[t][d] * alpha)
template <typename T>
__global__ __launch_bounds__(256) void posenc_kernel(const void* x, const float* table, const float* alpha, float xscale, void* y,
                                                     long long B, int Tn, int D, float drop_p, const unsigned long long* seed_ptr,
                                                     unsigned long long offset) {
  const unsigned long long seed = drop_p > 0.f ? *seed_ptr : 0ull;
  const long long n = B * Tn * D;
  const long long td = (long long)Tn * D;
  const float a = alpha ? *alpha : 1.f;
  const float keep = 1.f / (1.f - drop_p);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = Io<T>::ld(x, i) * xscale + table[i % td] * a;
    if (drop_p > 0.f) v = sc_dropped(seed, offset + i, sc_drop_threshold(drop_p)) ? 0.f : v * keep;
    Io<T>::st(y, i, v);
  }
}

// y = x * scale * keepmask / (1 - p)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const void* x, void* y, long long n, float scale, float drop_p,
                                                      const unsigned long long* seed_ptr, unsigned long long offset) {
  const unsigned long long seed = drop_p > 0.f ? *seed_ptr : 0ull;
  const float keep = scale / (1.f - drop_p);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = Io<T>::ld(x, i);
    v = (drop_p > 0.f && sc_dropped(seed, offset + i, sc_drop_threshold(drop_p))) ? 0.f : v * keep;
    Io<T>::st(y, i, v);
  }
}

// dpre = dy * act'(pre) * keepmask / (1 - p)     (act: 0 none, 1 ReLU, 2 exact GELU)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* dy, const void* pre, void* dpre, long long n, int act, float drop_p,
                                                      const unsigned long long* seed_ptr, unsigned long long offset) {
  const unsigned long long seed = drop_p > 0.f ? *seed_ptr : 0ull;
  const float keep = 1.f / (1.f - drop_p);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    float g = Io<T>::ld(dy, i);
    const float x = Io<T>::ld(pre, i);
    if (act == 1) g = x > 0.f ? g : 0.f;
    else if (act == 2) g *= sc_gelu_grad(x);
    if (drop_p > 0.f) g = sc_dropped(seed, offset + i, sc_drop_threshold(drop_p)) ? 0.f : g * keep;
    Io<T>::st(dpre, i, g);
  }
}

// (see sc_pad_rows) one 16-byte chunk per thread and round
__global__ __launch_bounds__(256) void pad_rows_kernel(const uint4* x, int B, int T, int C8, int r, uint4* out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int Tp = T + 2 * r;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long row = i / C8;
    const int c = (int)(i - row * C8);
    const long long b = row / Tp;
    const int t = (int)(row - b * Tp) - r;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (b < B && t >= 0 && t < T) v = x[(b * T + t) * C8 + c];
    out[i] = v;
  }
}

// (see sc_zero_pad_rows) the pad rows of a padded image only: 2 r rows per utterance + the rows behind the last utterance
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(int B, int T, int C8, int r, uint4* out, long long rows_total) {
  const int Tp = T + 2 * r;
  const long long npad = (long long)B * 2 * r + (rows_total - (long long)B * Tp);
  const long long n = npad * C8, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long pr = i / C8;
    const int c = (int)(i - pr * C8);
    long long row;
    if (pr < (long long)B * 2 * r) {
      const long long b = pr / (2 * r);
      const int j = (int)(pr - b * 2 * r);
      row = b * Tp + (j < r ? j : T + j);
    } else {
      row = (long long)B * Tp + (pr - (long long)B * 2 * r);
    }
    out[row * C8 + c] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// One launch refreshing many compute-dtype weight copies: descriptor d covers tiles [tile0, next.tile0) of a (rows, cols)
// f32 matrix; every 32 x 32 tile is written to dst (same layout, row stride ld_dst) and/or dst_t (transposed, row stride
// ld_dst_t).  Layout of sc_cast_desc: include/speechain_amd.h.
struct CastDesc {
  const float* src;
  void* dst;
  void* dst_t;
  int rows, cols;
  long long ld_dst, ld_dst_t;
  int tile0, tiles_x;
  int conv_k, tr_rev;  // conv_k > 1: src is a Conv1d weight (rows, cols / conv_k, conv_k); column c of the GEMM matrix = tap c / C, channel c % C
                       // tr_rev = 1 (with conv_k > 1): row (tap, channel) of the transposed copy is stored at row channel * conv_k + (conv_k - 1 - tap)
                       // -- the (C, k N) matrix [channel][j N + out] = w[out][channel][k - 1 - j] of the im2col-free Conv1d's data gradient
};

// 64 x 64 tiles, 256 threads: 16-byte f32 reads (4 columns per thread, 16 rows per pass), 8-byte bf16 stores; the transposed copy goes
// through an LDS image and leaves as 4 consecutive source rows = 4 contiguous elements of a destination row.  (The first version moved one
// element per thread on 32 x 32 tiles and reached 2.2 TB/s; the refresh of all weight copies is 0.9 GB per step.)
template <typename TO>
__global__ __launch_bounds__(256) void cast_weights_kernel(const CastDesc* desc, int n_desc) {
  __shared__ float tile[64][65];
  const int t = blockIdx.x;
  int lo = 0, hi = n_desc - 1;  // last descriptor with tile0 <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int lt = t - d.tile0;
  const int r0 = (lt / d.tiles_x) * 64, c0 = (lt % d.tiles_x) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 column quads x 16 rows
  const bool fast = d.conv_k <= 1 && (d.cols & 3) == 0 && ((uintptr_t)d.src & 15) == 0 && r0 + 64 <= d.rows && c0 + 64 <= d.cols;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = ty + 16 * i, r = r0 + rl, c = c0 + 4 * tx;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (fast) {
      const float4 q = *reinterpret_cast<const float4*>(d.src + (long long)r * d.cols + c);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if (r < d.rows) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ce = c + e;
        if (ce < d.cols) {
          int cs = ce;
          if (d.conv_k > 1) {
            const int C = d.cols / d.conv_k;
            cs = (ce % C) * d.conv_k + ce / C;
          }
          v[e] = d.src[(long long)r * d.cols + cs];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rl][4 * tx + e] = v[e];
    if (d.dst) {
      const long long o = (long long)r * d.ld_dst + c;
      if (fast && sizeof(TO) == 2 && (d.ld_dst & 3) == 0 && ((uintptr_t)d.dst & 7) == 0) {
        uint2 w;
        w.x = sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
        w.y = sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>((unsigned short*)d.dst + o) = w;
      } else if (r < d.rows) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < d.cols) Io<TO>::st(d.dst, o + e, v[e]);
      }
    }
  }
  if (!d.dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = ty + 16 * i, c = c0 + cl, r = r0 + 4 * tx;  // destination row c, 4 consecutive source rows
    if (c >= d.cols) continue;
    long long drow = c;
    if (d.tr_rev && d.conv_k > 1) {
      const int C = d.cols / d.conv_k;
      drow = (long long)(c % C) * d.conv_k + (d.conv_k - 1 - c / C);
    }
    const long long o = drow * d.ld_dst_t + r;
    if (sizeof(TO) == 2 && r + 4 <= d.rows && (d.ld_dst_t & 3) == 0 && ((uintptr_t)d.dst_t & 7) == 0) {
      uint2 w;
      w.x = sc_f32_to_bf16(tile[4 * tx][cl]) | ((unsigned)sc_f32_to_bf16(tile[4 * tx + 1][cl]) << 16);
      w.y = sc_f32_to_bf16(tile[4 * tx + 2][cl]) | ((unsigned)sc_f32_to_bf16(tile[4 * tx + 3][cl]) << 16);
      *reinterpret_cast<uint2*>((unsigned short*)d.dst_t + o) = w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (r + e < d.rows) Io<TO>::st(d.dst_t, o + e, tile[4 * tx + e][cl]);
    }
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// Gradient of an embedding lookup, deterministic and without atomics: one wave per token position.  The wave of position i owns row
// ids[i] of the table gradient iff no earlier position holds the same id; the owner then adds the rows dy[j] of ALL positions j >= i with
// that id in position order (matches found 64 at a time by ballot) and writes the table row once.  Rows nobody owns are zeroed by
// embedding_zero_kernel beforehand; the padding id receives no gradient (nn.Embedding(padding_idx=...)).
__global__ __launch_bounds__(256) void embedding_zero_kernel(float* dw, long long n4) {
  float4* p = reinterpret_cast<float4*>(dw);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// LDS: the ids are staged in LDS first (N ints), eight loads per lane in flight -- the two scans below then cost LDS reads; read from global
// memory they were chains of N / 64 dependent loads per scan (98 us for 2016 positions, all of it latency).
template <typename T, bool LDS>
__global__ __launch_bounds__(64) void embedding_bwd_kernel(const long long* ids, const T* dy, long long ld, int N, int D, long long padding_idx,
                                                           float* dw) {
  extern __shared__ int s_ids[];
  const int i = blockIdx.x, lane = threadIdx.x;
  if (LDS) {
    for (int j0 = 0; j0 < N; j0 += 64 * 8) {
      long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ids[min(j0 + 64 * u + lane, N - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + 64 * u + lane < N) s_ids[j0 + 64 * u + lane] = (int)v[u];  // (ids lie in [0, V), V an int)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes, then its own reads)
  }
  auto id_at = [&](int j) -> long long { return LDS ? (long long)s_ids[j] : ids[j]; };
  const long long my = LDS ? (long long)s_ids[i] : ids[i];
  if (my == padding_idx) return;
  for (int base = 0; base < i; base += 64) {
    const int j = base + lane;
    if (__ballot(j < i && id_at(j) == my)) return;  // an earlier position owns this row
  }
  // 2 x 4 columns per lane and pass (D <= 512: one pass).  An id that many positions hold -- <sos> leads every sentence -- makes its owner
  // sum that many rows, one per sentence, i.e. one per 64-position chunk: the owner first lists ALL its rows (LDS scans), then loads them
  // eight at a time; the additions stay in position order.  (Chunk by chunk that was a chain of 32 dependent row loads: 55-90 us.)
  int* s_rows = s_ids + N;
  int cnt = 0;
  if (LDS) {
    for (int base = i; base < N; base += 64) {
      const int j = base + lane;
      const bool hit = j < N && s_ids[j] == (int)my;
      const unsigned long long mask = __ballot(hit);
      if (hit) s_rows[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = j;
      cnt += __popcll(mask);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int c0 = 0; c0 < D; c0 += 64 * 8) {
    const int c = c0 + lane * 4;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto add_rows = [&](const int (&rows)[8], int nr) {
      float v[8][2][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = c + 256 * h + e;
            v[u][h][e] = cc < D ? Io<T>::ld(dy, (long long)rows[u] * ld + cc) : 0.f;
          }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < nr) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[h][e] += v[u][h][e];
        }
    };
    if (LDS) {
      for (int r0 = 0; r0 < cnt; r0 += 8) {
        int rows[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rows[u] = s_rows[min(r0 + u, cnt - 1)];
        add_rows(rows, min(8, cnt - r0));
      }
    } else {
      for (int base = i; base < N; base += 64) {
        const int j = base + lane;
        unsigned long long mask = __ballot(j < N && ids[j] == my);
        while (mask) {
          int rows[8], nr = 0;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (mask) {
              rows[u] = base + __builtin_ctzll(mask);
              mask &= mask - 1;
              nr = u + 1;
            } else {
              rows[u] = rows[0];
            }
          add_rows(rows, nr);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + 256 * h + e < D) dw[my * D + c + 256 * h + e] = acc[h][e];
  }
}

}  // namespace

// dW (V, D) f32 of out = table[ids] from dy (N, D) (row stride ld) -- the autograd of the token embedding in front of the decoder
// (speechain/module/prenet/embed.py:55-66).  ids must lie in [0, V).
SC_API int sc_embedding_bwd(int dtype, const long long* ids, const void* dy, long long ld, int N, int D, int V, long long padding_idx, float* dw,
                            hipStream_t stream) {
  SC_CHECK_ARG(ids && dy && dw, "sc_embedding_bwd: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_embedding_bwd: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(N >= 1 && D >= 1 && V >= 1 && ld >= D, "sc_embedding_bwd: bad sizes N=%d D=%d V=%d ld=%lld", N, D, V, ld);
  SC_CHECK_ARG((long long)V * D % 4 == 0 && (uintptr_t)dw % 16 == 0, "sc_embedding_bwd: the table gradient must be 16-byte aligned with V*D a multiple of 4");
  const long long n4 = (long long)V * D / 4;
  hipLaunchKernelGGL(embedding_zero_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, stream, dw, n4);
  SC_LAUNCH_CHECK("embedding_zero_kernel");
  const bool lds = N <= 7680;  // (60 KB: the ids and the owner's row list)
  const size_t smem = lds ? (size_t)2 * N * sizeof(int) : 0;
#define SC_LAUNCH_EMB(T, L) \
  hipLaunchKernelGGL((embedding_bwd_kernel<T, L>), dim3(N), dim3(64), smem, stream, ids, (const T*)dy, ld, N, D, padding_idx, dw)
  if (dtype == 0) {
    if (lds) SC_LAUNCH_EMB(float, true);
    else SC_LAUNCH_EMB(float, false);
  } else {
    if (lds) SC_LAUNCH_EMB(unsigned short, true);
    else SC_LAUNCH_EMB(unsigned short, false);
  }
#undef SC_LAUNCH_EMB
  SC_LAUNCH_CHECK("embedding_bwd_kernel");
  return SC_OK;
}

// dtype codes: 0 = f32, 1 = bf16
SC_API int sc_cast(int dtype_in, int dtype_out, const void* in, void* out, long long n, hipStream_t stream) {
  SC_CHECK_ARG(in && out && n >= 1, "sc_cast: bad arguments");
  if (dtype_in == 0 && dtype_out == 1) hipLaunchKernelGGL((cast_kernel<float, unsigned short>), dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  else if (dtype_in == 1 && dtype_out == 0) hipLaunchKernelGGL((cast_kernel<unsigned short, float>), dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  else if (dtype_in == 0 && dtype_out == 0) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  else hipLaunchKernelGGL((cast_kernel<unsigned short, unsigned short>), dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  SC_LAUNCH_CHECK("cast_kernel");
  return SC_OK;
}

// Zero-padded row image of a (B, T, C) bf16 tensor for a 'same' Conv1d evaluated WITHOUT an im2col copy (functional._ConvLinear):
// out holds rows_total rows of C; utterance b occupies rows b (T + 2 r) .. + T + 2 r - 1 = r zero rows, its T rows, r zero rows; rows
// behind the last utterance are zero.  The GEMM then reads "row" t of utterance b as the k C contiguous elements that start at padded
// row t (leading dimension C < K: rows overlap), i.e. [x[t - r], .., x[t + r]] -- the matrix sc_im2col would have written.
SC_API int sc_pad_rows(const void* x, int B, int T, int C, int r, void* out, long long rows_total, hipStream_t stream) {
  SC_CHECK_ARG(x && out && B >= 1 && T >= 1 && C >= 8 && C % 8 == 0 && r >= 0 && rows_total >= (long long)B * (T + 2 * r),
               "sc_pad_rows: bad arguments (B=%d T=%d C=%d r=%d rows=%lld; bf16, C %% 8 == 0)", B, T, C, r, rows_total);
  SC_CHECK_ARG(((uintptr_t)x | (uintptr_t)out) % 16 == 0, "sc_pad_rows: tensors must be 16-byte aligned");
  const long long n = rows_total * (C / 8);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const uint4*)x, B, T, C / 8, r, (uint4*)out, n);
  SC_LAUNCH_CHECK("pad_rows_kernel");
  return SC_OK;
}

// The zero rows of the image sc_pad_rows describes, and nothing else: for a producer (a GEMM with batch stride (T + 2 r) C and its output
// pointer r rows in) that writes the data rows itself.
SC_API int sc_zero_pad_rows(int B, int T, int C, int r, void* out, long long rows_total, hipStream_t stream) {
  SC_CHECK_ARG(out && B >= 1 && T >= 1 && C >= 8 && C % 8 == 0 && r >= 0 && rows_total >= (long long)B * (T + 2 * r) &&
                   (uintptr_t)out % 16 == 0,
               "sc_zero_pad_rows: bad arguments (B=%d T=%d C=%d r=%d rows=%lld; bf16, C %% 8 == 0, 16-byte aligned)", B, T, C, r, rows_total);
  const long long n = ((long long)B * 2 * r + (rows_total - (long long)B * (T + 2 * r))) * (C / 8);
  if (n == 0) return SC_OK;
  hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(grid_for(n)), dim3(256), 0, stream, B, T, C / 8, r, (uint4*)out, rows_total);
  SC_LAUNCH_CHECK("zero_pad_rows_kernel");
  return SC_OK;
}

SC_API int sc_cast_transpose(int dtype_out, const float* in, void* out, int R, int C, long long ld_out, hipStream_t stream) {
  SC_CHECK_ARG(in && out && R >= 1 && C >= 1 && ld_out >= R, "sc_cast_transpose: bad arguments");
  dim3 grid(sc_cdiv(C, 32), sc_cdiv(R, 32));
  if (dtype_out == 1) hipLaunchKernelGGL(cast_transpose_kernel<unsigned short>, grid, dim3(256), 0, stream, in, out, R, C, ld_out);
  else hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, stream, in, out, R, C, ld_out);
  SC_LAUNCH_CHECK("cast_transpose_kernel");
  return SC_OK;
}

SC_API int sc_posenc_fwd(int dtype, const void* x, const float* table, const float* alpha, float xscale, void* y, long long B,
                         int T, int D, float drop_p, const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(x && table && y && B >= 1 && T >= 1 && D >= 1, "sc_posenc_fwd: bad arguments");
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sc_posenc_fwd: dropout p=%f needs a seed cell", drop_p);
  const long long n = B * T * D;
  if (dtype == 0) hipLaunchKernelGGL(posenc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, x, table, alpha, xscale, y, B, T, D, drop_p, seed, offset);
  else hipLaunchKernelGGL(posenc_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, x, table, alpha, xscale, y, B, T, D, drop_p, seed, offset);
  SC_LAUNCH_CHECK("posenc_kernel");
  return SC_OK;
}

SC_API int sc_dropout(int dtype, const void* x, void* y, long long n, float scale, float drop_p, const unsigned long long* seed,
                      unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(x && y && n >= 1, "sc_dropout: bad arguments");
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sc_dropout: dropout p=%f needs a seed cell", drop_p);
  if (dtype == 0) hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n, scale, drop_p, seed, offset);
  else hipLaunchKernelGGL(dropout_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n, scale, drop_p, seed, offset);
  SC_LAUNCH_CHECK("dropout_kernel");
  return SC_OK;
}

SC_API int sc_act_bwd(int dtype, const void* dy, const void* pre, void* dpre, long long n, int act, float drop_p,
                      const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(dy && pre && dpre && n >= 1 && (drop_p == 0.f || seed), "sc_act_bwd: bad arguments");
  if (dtype == 0) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, dy, pre, dpre, n, act, drop_p, seed, offset);
  else hipLaunchKernelGGL(act_bwd_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, dy, pre, dpre, n, act, drop_p, seed, offset);
  SC_LAUNCH_CHECK("act_bwd_kernel");
  return SC_OK;
}

SC_API int sc_cast_weights(int dtype_out, const void* desc, int n_desc, int total_tiles, hipStream_t stream) {
  SC_CHECK_ARG(desc && n_desc >= 1 && total_tiles >= 1, "sc_cast_weights: bad arguments");
  if (dtype_out == 1) hipLaunchKernelGGL(cast_weights_kernel<unsigned short>, dim3(total_tiles), dim3(256), 0, stream, (const CastDesc*)desc, n_desc);
  else hipLaunchKernelGGL(cast_weights_kernel<float>, dim3(total_tiles), dim3(256), 0, stream, (const CastDesc*)desc, n_desc);
  SC_LAUNCH_CHECK("cast_weights_kernel");
  return SC_OK;
}
