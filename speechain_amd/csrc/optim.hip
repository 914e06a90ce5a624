// Optimizer tail of the training step for gfx950: global-norm gradient clipping + Adam over the flat gradient buckets.
// What it replaces in the reference's step (OptimScheduler.step, speechain/optim_sche/abs.py:232-253: clip_grad_norm_ over all parameters, then
// torch.optim.Adam.step; betas / eps from the recipes, e.g. 100-bpe5k_transformer-large_lr2e-3.yaml:251-261): ~40 launches and three
// extra passes over the 312 MB of gradients (g*g temporaries for the norm, g *= coef) become two launches and ONE pass:
//   sc_grad_sumsq  sum of squares of every bucket, two-stage and deterministic, finished in float64; also advances the step counter
//   sc_adam_step   p, m, v updated in place from g * clip_coef, clip_coef = min(1, max_norm / (||g|| + 1e-6)) formed per thread from
//                  the scalar (the clipped gradient itself is never written); a step whose norm is inf / nan is skipped
// Parameters, first and second moments live in flat float32 buffers with the layout of the gradient buckets.  HBM-bound:
// 4 reads + 3 writes of 4 bytes per parameter.
#include "sc_common.h"

namespace {

constexpr int MAX_BUCKETS = 48;
// elements per workgroup.  78 M parameters: 65536 (1190 workgroups, 64 rounds of 4 loads per thread) 397 us for the Adam launch, 16384 357 us,
// 8192 320 us, 4096 354 us (rocprofv3 rows of the bench step)
constexpr long long CHUNK = 8192;

struct Buckets {
  const float* g[MAX_BUCKETS];
  float* p[MAX_BUCKETS];
  float* m[MAX_BUCKETS];
  float* v[MAX_BUCKETS];
  long long n[MAX_BUCKETS];
  int chunk0[MAX_BUCKETS + 1];  // first chunk of every bucket
  int nb;
};

__device__ __forceinline__ int bucket_of(const Buckets& b, int chunk) {
  int k = 0;
  while (k + 1 < b.nb && chunk >= b.chunk0[k + 1]) ++k;
  return k;
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(Buckets b, double* partial) {
  __shared__ float red[4];
  const int k = bucket_of(b, blockIdx.x);
  const long long start = (long long)(blockIdx.x - b.chunk0[k]) * CHUNK;
  const long long end = min(b.n[k], start + CHUNK);
  const float* g = b.g[k];
  float s = 0.f;
  // buckets start 16-byte aligned and chunks are multiples of 4: float4 body, scalar tail
  const long long n4 = (end - start) >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g + start);
  for (long long i = threadIdx.x; i < n4; i += 256) {
    const float4 q = g4[i];
    s = fmaf(q.x, q.x, s); s = fmaf(q.y, q.y, s); s = fmaf(q.z, q.z, s); s = fmaf(q.w, q.w, s);
  }
  for (long long i = start + (n4 << 2) + threadIdx.x; i < end; i += 256) s = fmaf(g[i], g[i], s);
  s = sc_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

// one workgroup: sums the partials in a fixed order; state[0] = ||g||^2, state[1] = clip coefficient, state[2] = step (advanced here).
// A norm that is not finite (an overflowed bf16 step) SKIPS the update as the reference does (abs.py:239-247: "the parameters are not
// updated in this step"): state[1] = -1 tells adam_step_kernel to return, and the step counter stays.
__global__ __launch_bounds__(256) void grad_sumsq_finish_kernel(const double* partial, int n, float max_norm, float* state) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nsq = red[0];
    state[0] = (float)nsq;
    if (!isfinite(nsq)) {
      state[1] = -1.f;
      return;
    }
    const double coef = max_norm > 0.f ? (double)max_norm / (sqrt(nsq) + 1e-6) : 1.0;
    state[1] = (float)(coef < 1.0 ? coef : 1.0);
    state[2] += 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_step_kernel(Buckets b, const float* lr_cell, float beta1, float beta2, float eps, const float* state) {
  const int k = bucket_of(b, blockIdx.x);
  const long long start = (long long)(blockIdx.x - b.chunk0[k]) * CHUNK;
  const long long end = min(b.n[k], start + CHUNK);
  const float coef = state[1], step = state[2], lr = *lr_cell;
  if (coef < 0.f) return;  // skipped step (gradient norm not finite): p, m, v untouched
  // torch.optim.Adam: p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1, rsq2 = 1.f / sqrtf(bc2);
  const float a1 = 1.f - beta1, a2 = 1.f - beta2;
  const float* g = b.g[k];
  float *p = b.p[k], *m = b.m[k], *v = b.v[k];
  const long long n4 = (end - start) >> 2;
  for (long long i = threadIdx.x; i < n4; i += 256) {
    const long long o = start + (i << 2);
    const float4 gq = *reinterpret_cast<const float4*>(g + o);
    float4 pq = *reinterpret_cast<float4*>(p + o), mq = *reinterpret_cast<float4*>(m + o), vq = *reinterpret_cast<float4*>(v + o);
#define SC_ADAM1(c)                                             \
  {                                                             \
    const float gg = gq.c * coef;                               \
    mq.c = fmaf(beta1, mq.c, a1 * gg);                          \
    vq.c = fmaf(beta2, vq.c, a2 * gg * gg);                     \
    pq.c -= step_size * mq.c / (sqrtf(vq.c) * rsq2 + eps);      \
  }
    SC_ADAM1(x) SC_ADAM1(y) SC_ADAM1(z) SC_ADAM1(w)
#undef SC_ADAM1
    *reinterpret_cast<float4*>(p + o) = pq;
    *reinterpret_cast<float4*>(m + o) = mq;
    *reinterpret_cast<float4*>(v + o) = vq;
  }
  for (long long i = start + (n4 << 2) + threadIdx.x; i < end; i += 256) {
    const float gg = g[i] * coef;
    const float mm = fmaf(beta1, m[i], a1 * gg), vv = fmaf(beta2, v[i], a2 * gg * gg);
    m[i] = mm;
    v[i] = vv;
    p[i] -= step_size * mm / (sqrtf(vv) * rsq2 + eps);
  }
}

int fill(Buckets& b, int nb, const float* const* g, float* const* p, float* const* m, float* const* v, const long long* n) {
  b.nb = nb;
  int c = 0;
  for (int k = 0; k < nb; ++k) {
    b.g[k] = g[k];
    b.p[k] = p ? p[k] : nullptr;
    b.m[k] = m ? m[k] : nullptr;
    b.v[k] = v ? v[k] : nullptr;
    b.n[k] = n[k];
    b.chunk0[k] = c;
    c += (int)((n[k] + CHUNK - 1) / CHUNK);
  }
  b.chunk0[nb] = c;
  return c;
}

}  // namespace

SC_API long long sc_grad_sumsq_workspace(int nb, const long long* sizes) {
  long long c = 0;
  for (int k = 0; k < nb; ++k) c += (sizes[k] + CHUNK - 1) / CHUNK;
  return c * (long long)sizeof(double);
}

SC_API int sc_grad_sumsq(int nb, const float* const* grads, const long long* sizes, float max_norm, float* state, void* workspace,
                         long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(grads && sizes && state && workspace, "sc_grad_sumsq: null pointer");
  SC_CHECK_ARG(nb >= 1 && nb <= MAX_BUCKETS, "sc_grad_sumsq: 1..%d buckets, got %d", MAX_BUCKETS, nb);
  for (int k = 0; k < nb; ++k)
    SC_CHECK_ARG(grads[k] && sizes[k] >= 1 && (uintptr_t)grads[k] % 16 == 0, "sc_grad_sumsq: bucket %d is null, empty or not 16-byte aligned", k);
  SC_CHECK_ARG(ws_bytes >= sc_grad_sumsq_workspace(nb, sizes), "sc_grad_sumsq: workspace too small");
  Buckets b;
  const int chunks = fill(b, nb, grads, nullptr, nullptr, nullptr, sizes);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(chunks), dim3(256), 0, stream, b, (double*)workspace);
  SC_LAUNCH_CHECK("grad_sumsq_kernel");
  hipLaunchKernelGGL(grad_sumsq_finish_kernel, dim3(1), dim3(256), 0, stream, (const double*)workspace, chunks, max_norm, state);
  SC_LAUNCH_CHECK("grad_sumsq_finish_kernel");
  return SC_OK;
}

SC_API int sc_adam_step(int nb, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const long long* sizes, const float* lr, float beta1, float beta2, float eps, const float* state,
                        hipStream_t stream) {
  SC_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes && lr && state, "sc_adam_step: null pointer");
  SC_CHECK_ARG(nb >= 1 && nb <= MAX_BUCKETS, "sc_adam_step: 1..%d buckets, got %d", MAX_BUCKETS, nb);
  SC_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "sc_adam_step: bad hyper-parameters");
  for (int k = 0; k < nb; ++k)
    SC_CHECK_ARG(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k] && sizes[k] >= 1 && (uintptr_t)params[k] % 16 == 0 &&
                     (uintptr_t)grads[k] % 16 == 0 && (uintptr_t)exp_avg[k] % 16 == 0 && (uintptr_t)exp_avg_sq[k] % 16 == 0,
                 "sc_adam_step: bucket %d has a null, empty or misaligned buffer", k);
  Buckets b;
  const int chunks = fill(b, nb, grads, params, exp_avg, exp_avg_sq, sizes);
  hipLaunchKernelGGL(adam_step_kernel, dim3(chunks), dim3(256), 0, stream, b, lr, beta1, beta2, eps, state);
  SC_LAUNCH_CHECK("adam_step_kernel");
  return SC_OK;
}
