// FastSpeech2 glue kernels for gfx950: the frame-level expansion of the token sequence by its durations (length regulation), its
// adjoint, and the per-token average of a frame-level scalar.
//
// What they replace in the reference (paths under /root/reference/):
//   speechain/module/decoder/nar_tts.py:439-472   a Python loop over the batch: repeat_interleave per utterance, pad / cut to the longest
//                                                  target, unsqueeze, cat -- B small launches, B host reads of .long() counts, one big copy
//   speechain/module/decoder/nar_tts.py:150-204   average_scalar_by_duration: a (B, L, T) index cube, a gather of the same size and two
//                                                  reductions over it (O(B L T) memory traffic for O(B T) work)
// Durations arrive as the float tensor the reference computes (proc_duration: rounded, non-negative); they are truncated to integers as
// `.long()` does.  Frame t of utterance b belongs to token k iff  sum(d[b][:k]) <= t < sum(d[b][:k+1]).
//
// Byte-moving kernels (HBM-bound): the expansion reads each token row once per frame it covers (rows are L2 resident: L x D) and
// writes (B, T, D) once; its adjoint reads (B, T, D) once.  Deterministic: no atomics, fixed summation order.
#include "sc_common.h"

namespace {

constexpr int MAXL = 2048;  // tokens per utterance the kernels take (the duration row and its prefix sums live in LDS)

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct Io<unsigned short> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return sc_bf16_to_f32(((const unsigned short*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((unsigned short*)p)[i] = sc_f32_to_bf16(v); }
};

// prefix sums of one utterance's durations into LDS: cum[k] = frames in front of token k, cum[L] = all frames.  256 threads.
__device__ __forceinline__ void load_cum(const float* dur, int L, int* cum) {
  // (L is at most a few hundred: one thread scans; the row is 4 L bytes)
  if (threadIdx.x == 0) {
    int s = 0;
    for (int k = 0; k < L; ++k) {
      cum[k] = s;
      const float d = dur[k];
      s += d > 0.f ? (int)d : 0;
    }
    cum[L] = s;
  }
  __syncthreads();
}

// out[b][t][:] = x[b][k(t)][:], zero for t >= total frames.  grid (ceil(T / 32), B); wave w of the workgroup writes rows 8 w .. 8 w + 7
// of its 32-row strip.
template <typename T>
__global__ __launch_bounds__(256) void length_regulate_fwd_kernel(const void* x, const float* dur, void* out, int L, int Tn, int D) {
  __shared__ int cum[MAXL + 1];
  const int b = blockIdx.y;
  load_cum(dur + (long long)b * L, L, cum);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int total = cum[L];
  for (int r = 0; r < 8; ++r) {
    const int t = blockIdx.x * 32 + wid * 8 + r;
    if (t >= Tn) break;
    const long long ob = ((long long)b * Tn + t) * D;
    if (t >= total) {
      for (int c = lane; c < D; c += 64) Io<T>::st(out, ob + c, 0.f);
      continue;
    }
    int lo = 0, hi = L - 1;  // last k with cum[k] <= t  (tokens of zero duration share their successor's start and are skipped)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (cum[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const long long xb = ((long long)b * L + lo) * D;
    if (sizeof(T) == 2 && D % 8 == 0 && (((uintptr_t)x | (uintptr_t)out) % 16 == 0)) {
      for (int c = lane * 8; c < D; c += 512) *(uint4*)((unsigned short*)out + ob + c) = *(const uint4*)((const unsigned short*)x + xb + c);
    } else {
      for (int c = lane; c < D; c += 64) Io<T>::st(out, ob + c, Io<T>::ld(x, xb + c));
    }
  }
}

// dx[b][k][:] = sum over the frames of token k that exist (t < Tn) of dy[b][t][:], in frame order.  grid (ceil(L / 4), B); one wave per token.
template <typename T>
__global__ __launch_bounds__(256) void length_regulate_bwd_kernel(const void* dy, const float* dur, void* dx, int L, int Tn, int D) {
  __shared__ int cum[MAXL + 1];
  const int b = blockIdx.y;
  load_cum(dur + (long long)b * L, L, cum);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wid;
  if (k >= L) return;
  const int t0 = cum[k], t1 = min(cum[k + 1], Tn);
  const long long xb = ((long long)b * L + k) * D;
  for (int c0 = 0; c0 < D; c0 += 256) {  // 4 columns per lane and pass
    const int c = c0 + lane * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t < t1; ++t) {
      const long long yb = ((long long)b * Tn + t) * D;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < D) acc[e] += Io<T>::ld(dy, yb + c + e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < D) Io<T>::st(dx, xb + c + e, acc[e]);
  }
}

// out[b][k] = mean of frame[b][t] over the frames of token k that exist (t < Tn); 0 when there are none.  grid (B); thread per token.
__global__ __launch_bounds__(256) void avg_by_duration_kernel(const float* frame, const float* dur, float* out, int L, int Tn) {
  __shared__ int cum[MAXL + 1];
  const int b = blockIdx.x;
  load_cum(dur + (long long)b * L, L, cum);
  for (int k = threadIdx.x; k < L; k += 256) {
    const int t0 = cum[k], t1 = min(cum[k + 1], Tn);
    float s = 0.f;
    for (int t = t0; t < t1; ++t) s += frame[(long long)b * Tn + t];
    const int n = max(t1 - t0, 0);
    out[(long long)b * L + k] = s / ((float)n + 1e-10f);  // (decoder/nar_tts.py:203: the 1e-10 keeps an empty token at 0)
  }
}

}  // namespace

// dtype: 0 = f32, 1 = bf16 (x and out).  dur (B, L) f32, non-negative, already rounded.
SC_API int sc_length_regulate_fwd(int dtype, const void* x, const float* dur, void* out, int B, int L, int T, int D, hipStream_t stream) {
  SC_CHECK_ARG(x && dur && out, "sc_length_regulate_fwd: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_length_regulate_fwd: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && L >= 1 && L <= MAXL && T >= 1 && D >= 1, "sc_length_regulate_fwd: bad sizes B=%d L=%d T=%d D=%d (L <= %d)", B, L, T, D, MAXL);
  const dim3 grid(sc_cdiv(T, 32), B);
  if (dtype == 0) hipLaunchKernelGGL(length_regulate_fwd_kernel<float>, grid, dim3(256), 0, stream, x, dur, out, L, T, D);
  else hipLaunchKernelGGL(length_regulate_fwd_kernel<unsigned short>, grid, dim3(256), 0, stream, x, dur, out, L, T, D);
  SC_LAUNCH_CHECK("length_regulate_fwd_kernel");
  return SC_OK;
}

SC_API int sc_length_regulate_bwd(int dtype, const void* dy, const float* dur, void* dx, int B, int L, int T, int D, hipStream_t stream) {
  SC_CHECK_ARG(dy && dur && dx, "sc_length_regulate_bwd: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_length_regulate_bwd: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && L >= 1 && L <= MAXL && T >= 1 && D >= 1, "sc_length_regulate_bwd: bad sizes B=%d L=%d T=%d D=%d (L <= %d)", B, L, T, D, MAXL);
  const dim3 grid(sc_cdiv(L, 4), B);
  if (dtype == 0) hipLaunchKernelGGL(length_regulate_bwd_kernel<float>, grid, dim3(256), 0, stream, dy, dur, dx, L, T, D);
  else hipLaunchKernelGGL(length_regulate_bwd_kernel<unsigned short>, grid, dim3(256), 0, stream, dy, dur, dx, L, T, D);
  SC_LAUNCH_CHECK("length_regulate_bwd_kernel");
  return SC_OK;
}

SC_API int sc_avg_by_duration(const float* frame, const float* dur, float* out, int B, int L, int T, hipStream_t stream) {
  SC_CHECK_ARG(frame && dur && out, "sc_avg_by_duration: null pointer");
  SC_CHECK_ARG(B >= 1 && L >= 1 && L <= MAXL && T >= 1, "sc_avg_by_duration: bad sizes B=%d L=%d T=%d (L <= %d)", B, L, T, MAXL);
  hipLaunchKernelGGL(avg_by_duration_kernel, dim3(B), dim3(256), 0, stream, frame, dur, out, L, T);
  SC_LAUNCH_CHECK("avg_by_duration_kernel");
  return SC_OK;
}
