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
  int warp_mode;              // 0 bicubic (4 taps, A = -0.75), 1 bilinear (2 taps): F.interpolate's modes that take align_corners=True
  float scale_l, scale_r;
  const long long* freq_len;  // (B, nf) or null
  const long long* freq_pos;
  int nf;
  const long long* time_len;  // (B, nt) or null
  const long long* time_pos;
  int nt;
  const float* fill;          // device scalar or null (fill = 0)
  double* partial_out;        // per-workgroup sum of the written elements (pass 1 of sc_specaug_fwd_meanfill) or null
  const double* partial_in;   // the partial sums of pass 1: fill = their total * inv_count (pass 2) or null
  int n_partial;
  double inv_count;
};

__device__ __forceinline__ float cubic_inner(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic_outer(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

constexpr int kMaxMasks = 16;

// One workgroup = kRowsPerBlock consecutive frames of one utterance.  The mask ranges of the utterance are staged in LDS once and the
// frequency mask is expanded to one flag per column once (the first version re-derived both per element from global memory in 8032
// workgroups of ~320 elements each: 29 us for 20 MB, all of it block prologue latency); a thread then owns 4 consecutive columns
// (16-byte loads / stores when F % 4 == 0) and walks the rows.  Per-element arithmetic is unchanged.
constexpr int kRowsPerBlock = 32;
constexpr int kMaxF = 1024;

__global__ __launch_bounds__(256) void specaug_kernel(SpecAugArgs p) {
  __shared__ long long tps[kMaxMasks * 64], tln[kMaxMasks * 64];
  __shared__ int fm[kMaxF];
  const int b = blockIdx.y;
  const int row0 = blockIdx.x * kRowsPerBlock;
  const int tid = threadIdx.x;
  for (int i = tid; i < p.nt; i += 256) { tps[i] = p.time_pos[b * p.nt + i]; tln[i] = p.time_len[b * p.nt + i]; }
  for (int f = tid; f < p.F; f += 256) {
    bool m = false;
    for (int i = 0; i < p.nf; ++i) {
      const long long ps = p.freq_pos[b * p.nf + i], ln = p.freq_len[b * p.nf + i];
      m |= (ps <= f) && (f < ps + ln);
    }
    fm[f] = m ? 1 : 0;
  }
  __shared__ double red[256];
  if (p.partial_in) {  // fill = mean of the warped features (specaug.py:203), from the per-workgroup sums of the first pass
    double s = 0.0;
    for (int i = tid; i < p.n_partial; i += 256) s += p.partial_in[i];
    red[tid] = s;
  }
  __syncthreads();
  if (p.partial_in) {
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
  }
  const float fillv = p.partial_in ? (float)(red[0] * p.inv_count) : (p.fill ? *p.fill : 0.f);
  float wsum = 0.f;
  const int FQ = (p.F + 3) >> 2;                 // column quads per row
  const int lanes = max(1, 256 / FQ);            // rows in flight
  const int q = tid % FQ, rl = tid / FQ;
  const bool vec = (p.F & 3) == 0 && (((uintptr_t)p.in | (uintptr_t)p.out) & 15) == 0;
  const float* src = p.in + (long long)b * p.T * p.F;
  for (int qq = q; qq < FQ && rl < lanes; qq += 256) {  // (FQ > 256 only for F > 1024, which the host refuses)
    const int f0 = qq * 4, nc = min(4, p.F - f0);
    for (int t = row0 + rl; t < min(row0 + kRowsPerBlock, p.T); t += lanes) {
      bool tmask = false;
      for (int i = 0; i < p.nt; ++i) tmask |= (tps[i] <= t) && (t < tps[i] + tln[i]);
      int idx[4];
      float w[4];
      const bool warp = p.warp_center >= 0 && t < p.tmin;
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
        if (p.warp_mode == 1) {  // bilinear: (1 - lam) x[i0] + lam x[i0 + 1] (aten upsample_bilinear2d, the width axis is the identity here)
          w[0] = 0.f;
          w[1] = 1.f - lam;
          w[2] = lam;
          w[3] = 0.f;
        } else {
          w[0] = cubic_outer(lam + 1.f, A);
          w[1] = cubic_inner(lam, A);
          w[2] = cubic_inner(1.f - lam, A);
          w[3] = cubic_outer(2.f - lam, A);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int qi = i0 - 1 + j;
          qi = qi < 0 ? 0 : (qi > in_len - 1 ? in_len - 1 : qi);
          idx[j] = base + qi;
        }
      }
      float v[4];
      if (warp) {
        float r[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* sp = src + (long long)idx[j] * p.F + f0;
          if (vec) { const float4 x = *reinterpret_cast<const float4*>(sp); r[j][0] = x.x; r[j][1] = x.y; r[j][2] = x.z; r[j][3] = x.w; }
          else { for (int e = 0; e < 4; ++e) r[j][e] = e < nc ? sp[e] : 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = r[0][e] * w[0];
          a += r[1][e] * w[1];
          a += r[2][e] * w[2];
          a += r[3][e] * w[3];
          v[e] = a;
        }
      } else {
        const float* sp = src + (long long)t * p.F + f0;
        if (vec) { const float4 x = *reinterpret_cast<const float4*>(sp); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
        else { for (int e = 0; e < 4; ++e) v[e] = e < nc ? sp[e] : 0.f; }
      }
      // (a select, not `if (tmask || (e < nc && fm[..])) v[e] = fillv`: hipcc 7.2 compiled that short-circuit form into code that
      //  computed the column flag and then ignored it -- frequency masks silently had no effect)
      int fl4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) fl4[e] = fm[min(f0 + e, p.F - 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool hit = tmask | ((e < nc) & (fl4[e] != 0));
        v[e] = hit ? fillv : v[e];
      }
      float* dp = p.out + ((long long)b * p.T + t) * p.F + f0;
      if (vec) *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
      else { for (int e = 0; e < nc; ++e) dp[e] = v[e]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) wsum += e < nc ? v[e] : 0.f;
    }
  }
  if (p.partial_out) {  // (<= 128 elements per thread in float, the workgroup's and the grid's totals in double)
    __syncthreads();
    red[tid] = (double)wsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) p.partial_out[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
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

SC_API int sc_specaug_fwd(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos, int warp_mode,
                          const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                          const long long* time_pos, int nt, const float* fill, hipStream_t stream) {
  SC_CHECK_ARG(warp_mode == 0 || warp_mode == 1, "sc_specaug_fwd: warp_mode must be 0 (bicubic) or 1 (bilinear)");
  SC_CHECK_ARG(in && out, "sc_specaug_fwd: null pointer");
  SC_CHECK_ARG(B >= 1 && T >= 1 && F >= 1, "sc_specaug_fwd: bad shape (%d,%d,%d)", B, T, F);
  SC_CHECK_ARG(nf == 0 || (freq_len && freq_pos), "sc_specaug_fwd: nf=%d without index tensors", nf);
  SC_CHECK_ARG(nt == 0 || (time_len && time_pos), "sc_specaug_fwd: nt=%d without index tensors", nt);
  SC_CHECK_ARG(nf <= kMaxMasks * 64 && nt <= kMaxMasks * 64, "sc_specaug_fwd: too many masks");
  SpecAugArgs a{};
  a.in = in; a.out = out; a.B = B; a.T = T; a.F = F; a.tmin = tmin;
  a.warp_center = warp_center; a.warp_pos = warp_pos; a.warp_mode = warp_mode;
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
  SC_CHECK_ARG(F <= kMaxF, "sc_specaug_fwd: at most %d feature columns", kMaxF);
  dim3 grid(sc_cdiv(T, kRowsPerBlock), B);
  hipLaunchKernelGGL(specaug_kernel, grid, dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK("specaug_kernel");
  return SC_OK;
}

SC_API long long sc_specaug_meanfill_workspace(int B, int T) { return (long long)sc_cdiv(T, kRowsPerBlock) * B * (long long)sizeof(double); }

// Warp, then mask with the mean of the warped features as fill value (specaug.py:111-124, :160-204 with feat_norm off) in one call: pass 1
// warps in -> out and leaves one sum per workgroup, pass 2 masks out in place and every workgroup totals the sums for its fill value.
SC_API int sc_specaug_fwd_meanfill(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos, int warp_mode,
                                   const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                                   const long long* time_pos, int nt, void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(in && out && workspace, "sc_specaug_fwd_meanfill: null pointer");
  SC_CHECK_ARG(in != out, "sc_specaug_fwd_meanfill: not in place");
  SC_CHECK_ARG(warp_mode == 0 || warp_mode == 1, "sc_specaug_fwd_meanfill: warp_mode must be 0 (bicubic) or 1 (bilinear)");
  SC_CHECK_ARG(B >= 1 && T >= 1 && F >= 1 && F <= kMaxF, "sc_specaug_fwd_meanfill: bad shape (%d,%d,%d)", B, T, F);
  SC_CHECK_ARG(nf == 0 || (freq_len && freq_pos), "sc_specaug_fwd_meanfill: nf=%d without index tensors", nf);
  SC_CHECK_ARG(nt == 0 || (time_len && time_pos), "sc_specaug_fwd_meanfill: nt=%d without index tensors", nt);
  SC_CHECK_ARG(nf <= kMaxMasks * 64 && nt <= kMaxMasks * 64, "sc_specaug_fwd_meanfill: too many masks");
  SC_CHECK_ARG(ws_bytes >= sc_specaug_meanfill_workspace(B, T), "sc_specaug_fwd_meanfill: workspace too small");
  SpecAugArgs a{};
  a.in = in; a.out = out; a.B = B; a.T = T; a.F = F; a.tmin = tmin;
  a.warp_center = warp_center; a.warp_pos = warp_pos; a.warp_mode = warp_mode;
  if (warp_center >= 0) {
    SC_CHECK_ARG(warp_pos >= 1 && warp_pos < tmin && warp_center >= 1 && warp_center < tmin && tmin <= T,
                 "sc_specaug_fwd_meanfill: bad warp (center=%d pos=%d tmin=%d T=%d)", warp_center, warp_pos, tmin, T);
    const int lo = warp_pos, ro = tmin - warp_pos;
    a.scale_l = lo > 1 ? (float)((double)(warp_center - 1) / (double)(lo - 1)) : 0.f;
    a.scale_r = ro > 1 ? (float)((double)(tmin - warp_center - 1) / (double)(ro - 1)) : 0.f;
  }
  dim3 grid(sc_cdiv(T, kRowsPerBlock), B);
  a.partial_out = (double*)workspace;
  hipLaunchKernelGGL(specaug_kernel, grid, dim3(256), 0, stream, a);
  SpecAugArgs m{};
  m.in = out; m.out = out; m.B = B; m.T = T; m.F = F; m.tmin = tmin;
  m.warp_center = m.warp_pos = -1;
  m.freq_len = freq_len; m.freq_pos = freq_pos; m.nf = nf;
  m.time_len = time_len; m.time_pos = time_pos; m.nt = nt;
  m.partial_in = (const double*)workspace;
  m.n_partial = (int)(grid.x * grid.y);
  m.inv_count = 1.0 / ((double)B * T * F);
  hipLaunchKernelGGL(specaug_kernel, grid, dim3(256), 0, stream, m);
  SC_LAUNCH_CHECK("specaug_kernel (mean fill)");
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
