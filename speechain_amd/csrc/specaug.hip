// SpecAugment for gfx950: bicubic time warp + frequency/time masking in ONE pass over the features.
//
// Replaces (paths under /root/reference/) speechain/module/augment/specaug.py:86-204:
//   two F.interpolate(mode='bicubic', align_corners=True) calls + in-place slice writes (:111-124),
//   four compare/any mask constructions (:160-199) and the masked_fill (:202-204).
// The random integers are drawn on the host side of the boundary in the reference's order and handed in.
//
// Time warp: F.interpolate on (B,1,T,F) with the F size unchanged is a 1-D cubic convolution along T
// (Keys kernel, A = -0.75, 4 taps, border-clamped indices, src = dst * fp32((T_in-1)/(T_out-1))).
// HBM-bound: each output element reads <= 4 input rows that neighbouring threads share through L1/L2.
#include "sc_common.h"

namespace {

struct SpecAugArgs {
  const float* in;
  float* out;
  int B, T, F;
  int tmin;
  int warp_center, warp_pos;  // warp_center < 0: no warp
  float scale_l, scale_r;
  const long long* freq_len;  // (B, nf) or null
  const long long* freq_pos;
  int nf;
  const long long* time_len;  // (B, nt) or null
  const long long* time_pos;
  int nt;
  const float* fill;          // device scalar or null (fill = 0)
};

__device__ __forceinline__ float cubic_inner(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic_outer(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

constexpr int kMaxMasks = 16;

__global__ __launch_bounds__(256) void specaug_kernel(SpecAugArgs p) {
  const int b = blockIdx.y;
  const int rows_per_block = blockDim.y;
  const int t = blockIdx.x * rows_per_block + threadIdx.y;
  if (t >= p.T) return;
  const float fillv = p.fill ? *p.fill : 0.f;

  // time mask for this row (uniform over the row's threads)
  bool tmask = false;
  for (int i = 0; i < p.nt; ++i) {
    const long long ps = p.time_pos[b * p.nt + i], ln = p.time_len[b * p.nt + i];
    tmask |= (ps <= t) && (t < ps + ln);
  }

  // warp source rows and taps
  int idx[4];
  float w[4];
  bool warp = p.warp_center >= 0 && t < p.tmin;
  if (warp) {
    int base, in_len, dst;
    float scale;
    if (t < p.warp_pos) { base = 0; in_len = p.warp_center; dst = t; scale = p.scale_l; }
    else { base = p.warp_center; in_len = p.tmin - p.warp_center; dst = t - p.warp_pos; scale = p.scale_r; }
    float real = scale * (float)dst;
    asm volatile("" : "+v"(real));  // aten rounds the product before taking floor / fraction: keep hipcc from
                                    // contracting it into fma(scale, dst, -floor) (HIP's __fmul_rn does not)
    int i0 = (int)floorf(real);
    if (i0 > in_len - 1) i0 = in_len - 1;
    float lam = real - (float)i0;
    lam = fminf(fmaxf(lam, 0.f), 1.f);
    const float A = -0.75f;
    w[0] = cubic_outer(lam + 1.f, A);
    w[1] = cubic_inner(lam, A);
    w[2] = cubic_inner(1.f - lam, A);
    w[3] = cubic_outer(2.f - lam, A);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int q = i0 - 1 + j;
      q = q < 0 ? 0 : (q > in_len - 1 ? in_len - 1 : q);
      idx[j] = base + q;
    }
  }

  const float* src = p.in + (long long)b * p.T * p.F;
  float* dst_row = p.out + ((long long)b * p.T + t) * p.F;
  for (int f = threadIdx.x; f < p.F; f += blockDim.x) {
    bool masked = tmask;
    for (int i = 0; i < p.nf; ++i) {
      const long long ps = p.freq_pos[b * p.nf + i], ln = p.freq_len[b * p.nf + i];
      masked |= (ps <= f) && (f < ps + ln);
    }
    float v;
    if (masked) {
      v = fillv;
    } else if (warp) {
      v = src[(long long)idx[0] * p.F + f] * w[0];
      v += src[(long long)idx[1] * p.F + f] * w[1];
      v += src[(long long)idx[2] * p.F + f] * w[2];
      v += src[(long long)idx[3] * p.F + f] * w[3];
    } else {
      v = src[(long long)t * p.F + f];
    }
    dst_row[f] = v;
  }
}

// Banded mel filterbank + clamp/log on an (N, NB) spectrogram: the stand-alone LinearSpec2MelSpec.forward
// (speechain/module/frontend/linear2mel.py:163-189).  One thread per (row, mel).
__global__ __launch_bounds__(256) void mel_log_kernel(const float* spec, long long N, int NB, const float* melw,
                                                       const int* band_lo, const int* band_hi, int n_mels, float clamp,
                                                       int logging, float log_div, float* out) {
  const long long total = N * n_mels;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / n_mels;
    const int m = (int)(e - row * n_mels);
    const float* x = spec + row * NB;
    const float* w = melw + (long long)m * NB;
    float s = 0.f;
    for (int k = band_lo[m]; k < band_hi[m]; ++k) s = fmaf(x[k], w[k], s);
    if (logging) s = logf(fmaxf(s, clamp)) / log_div;
    out[e] = s;
  }
}

}  // namespace

SC_API int sc_specaug_fwd(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos,
                          const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                          const long long* time_pos, int nt, const float* fill, hipStream_t stream) {
  SC_CHECK_ARG(in && out, "sc_specaug_fwd: null pointer");
  SC_CHECK_ARG(B >= 1 && T >= 1 && F >= 1, "sc_specaug_fwd: bad shape (%d,%d,%d)", B, T, F);
  SC_CHECK_ARG(nf == 0 || (freq_len && freq_pos), "sc_specaug_fwd: nf=%d without index tensors", nf);
  SC_CHECK_ARG(nt == 0 || (time_len && time_pos), "sc_specaug_fwd: nt=%d without index tensors", nt);
  SC_CHECK_ARG(nf <= kMaxMasks * 64 && nt <= kMaxMasks * 64, "sc_specaug_fwd: too many masks");
  SpecAugArgs a{};
  a.in = in; a.out = out; a.B = B; a.T = T; a.F = F; a.tmin = tmin;
  a.warp_center = warp_center; a.warp_pos = warp_pos;
  if (warp_center >= 0) {
    SC_CHECK_ARG(in != out, "sc_specaug_fwd: the time warp cannot run in place");
    SC_CHECK_ARG(warp_pos >= 1 && warp_pos < tmin && warp_center >= 1 && warp_center < tmin && tmin <= T,
                 "sc_specaug_fwd: bad warp (center=%d pos=%d tmin=%d T=%d)", warp_center, warp_pos, tmin, T);
    const int lo = warp_pos, ro = tmin - warp_pos;
    a.scale_l = lo > 1 ? (float)((double)(warp_center - 1) / (double)(lo - 1)) : 0.f;
    a.scale_r = ro > 1 ? (float)((double)(tmin - warp_center - 1) / (double)(ro - 1)) : 0.f;
  }
  a.freq_len = freq_len; a.freq_pos = freq_pos; a.nf = nf;
  a.time_len = time_len; a.time_pos = time_pos; a.nt = nt;
  a.fill = fill;
  const int bx = F >= 64 ? 64 : 32;
  dim3 block(bx, 256 / bx);
  dim3 grid(sc_cdiv(T, block.y), B);
  hipLaunchKernelGGL(specaug_kernel, grid, block, 0, stream, a);
  SC_LAUNCH_CHECK("specaug_kernel");
  return SC_OK;
}

SC_API int sc_mel_log_fwd(const float* spec, long long N, int NB, const float* melw, const int* band_lo, const int* band_hi,
                          int n_mels, float clamp, int logging, float log_div, float* out, hipStream_t stream) {
  SC_CHECK_ARG(spec && melw && band_lo && band_hi && out, "sc_mel_log_fwd: null pointer");
  SC_CHECK_ARG(N >= 1 && NB >= 1 && n_mels >= 1, "sc_mel_log_fwd: bad sizes");
  long long blocks = (N * n_mels + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mel_log_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, spec, N, NB, melw, band_lo, band_hi,
                     n_mels, clamp, logging, log_div, out);
  SC_LAUNCH_CHECK("mel_log_kernel");
  return SC_OK;
}
