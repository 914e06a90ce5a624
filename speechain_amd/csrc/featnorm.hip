// Feature normalisation for gfx950 -- the device work of speechain/module/norm/feat_norm.py (reference path /root/reference/):
//   * per-utterance statistics over the VALID frames, feat[i][:feat_len[i]].mean(0) and the unbiased .std(0) clamped from below
//     (feat_norm.py:115-133) -- a Python loop over the batch in the reference, one launch here;
//   * (feat - mean) / std over the WHOLE padded tensor with per-utterance or shared vectors (feat_norm.py:137-139, 280-299,
//     322-331; padding frames are normalised too, exactly as the reference does).
// The running-statistics bookkeeping ((B, F)-sized) stays in the host module.  f32 only: the frontend output is f32.
#include "sc_common.h"

namespace {

// grid (B, ceil(F / 64)); 1024 threads = 64 features x 16 row-lanes.  Two passes (mean, then squared deviations): the same
// formulation as torch's mean / std, so an utterance of one frame gives std = 0/0 = NaN and an empty one mean = NaN, as there.
__global__ __launch_bounds__(1024) void featnorm_stats_kernel(const float* feat, long long sb, long long st, const long long* feat_len,
                                                              int T, int F, float clamp, float* mean, float* stdv, int want_std) {
  __shared__ float red[16][65];
  const int b = blockIdx.x;
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.y * 64 + c;
  long long n = feat_len[b];
  n = n < 0 ? 0 : (n > T ? T : n);
  const float* x = feat + (long long)b * sb + f;
  float s = 0.f;
  if (f < F)
    for (long long t = rl; t < n; t += 16) s += x[t * st];
  red[rl][c] = s;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mu += red[i][c];
  mu = mu / (float)n;
  __syncthreads();
  if (rl == 0 && f < F && mean) mean[(long long)b * F + f] = mu;
  if (!want_std) return;
  float q = 0.f;
  if (f < F)
    for (long long t = rl; t < n; t += 16) {
      const float d = x[t * st] - mu;
      q += d * d;
    }
  red[rl][c] = q;
  __syncthreads();
  if (rl == 0 && f < F) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][c];
    v = sqrtf(v / (float)(n - 1));
    stdv[(long long)b * F + f] = (v != v) ? v : fmaxf(v, clamp);  // torch.clamp keeps NaN (fmaxf would drop it)
  }
}

// out[b][t][f] = (feat[b][t][f] - mean[b * mb + f]) / std[b * sb2 + f]; either vector may be null; true division
__global__ __launch_bounds__(256) void featnorm_apply_kernel(const float* feat, float* out, long long sb, long long st, long long ob,
                                                             long long ot, int B, int T, int F, const float* mean, long long mb,
                                                             const float* stdv, long long sb2) {
  const long long total = (long long)B * T * F;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const long long bt = i / F;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    float v = feat[(long long)b * sb + (long long)t * st + f];
    if (mean) v -= mean[(long long)b * mb + f];
    if (stdv) v /= stdv[(long long)b * sb2 + f];
    out[(long long)b * ob + (long long)t * ot + f] = v;
  }
}

}  // namespace

SC_API int sc_featnorm_stats(const float* feat, long long stride_b, long long stride_t, const long long* feat_len, int B, int T, int F,
                             float clamp, float* mean, float* std, hipStream_t stream) {
  SC_CHECK_ARG(feat && feat_len && (mean || std), "sc_featnorm_stats: null pointer");
  SC_CHECK_ARG(B >= 1 && T >= 1 && F >= 1 && B <= 65535, "sc_featnorm_stats: bad sizes B=%d T=%d F=%d", B, T, F);
  SC_CHECK_ARG(stride_t >= F && stride_b >= (long long)T * stride_t, "sc_featnorm_stats: strides too small");
  hipLaunchKernelGGL(featnorm_stats_kernel, dim3(B, sc_cdiv(F, 64)), dim3(1024), 0, stream, feat, stride_b, stride_t, feat_len, T, F, clamp,
                     mean, std, std ? 1 : 0);
  SC_LAUNCH_CHECK("featnorm_stats_kernel");
  return SC_OK;
}

SC_API int sc_featnorm_apply(const float* feat, float* out, long long stride_b, long long stride_t, long long out_stride_b,
                             long long out_stride_t, int B, int T, int F, const float* mean, long long mean_stride_b, const float* std,
                             long long std_stride_b, hipStream_t stream) {
  SC_CHECK_ARG(feat && out, "sc_featnorm_apply: null pointer");
  SC_CHECK_ARG(B >= 1 && T >= 1 && F >= 1, "sc_featnorm_apply: bad sizes B=%d T=%d F=%d", B, T, F);
  SC_CHECK_ARG(stride_t >= F && out_stride_t >= F, "sc_featnorm_apply: row strides smaller than F");
  SC_CHECK_ARG((mean_stride_b == 0 || mean_stride_b >= F) && (std_stride_b == 0 || std_stride_b >= F),
               "sc_featnorm_apply: a statistics stride is 0 (one shared vector) or >= F (one vector per utterance)");
  const long long total = (long long)B * T * F;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(featnorm_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, feat, out, stride_b, stride_t, out_stride_b,
                     out_stride_t, B, T, F, mean, mean_stride_b, std, std_stride_b);
  SC_LAUNCH_CHECK("featnorm_apply_kernel");
  return SC_OK;
}
