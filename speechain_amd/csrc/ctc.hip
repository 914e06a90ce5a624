// CTC loss forward / backward for gfx950 -- replaces log_softmax + F.ctc_loss(reduction='none', zero_infinity) of
// speechain/criterion/ctc.py:48-65 (reference path /root/reference/).
//
// The (B, T, V) log-softmax is never materialised: a row-wise logsumexp (sc_row_lse) gives lse[b][t] and the
// recursions gather logit[b][t][label] - lse[b][t] for the <= 2L+1 extended labels only.
//   forward : one workgroup per utterance, alpha_t(s) in LDS (double buffered), one barrier per frame; all alphas are
//             kept in a (B, T, S) f32 workspace for the backward pass; nll[b] = -logsumexp(alpha_T(S-1), alpha_T(S-2))
//   backward: a dense pass writes g_b * softmax (zero beyond the utterance length); then one workgroup per utterance
//             runs the beta recursion and overwrites the label columns with
//               g_b * (softmax[t][v] - exp(logsumexp_{s: l'_s = v}(alpha_t(s) + beta_t(s)) - lp[t][v] + nll_b))
//             (beta includes the emission at t, as in torch's kernels).
#include "sc_common.h"

namespace {

template <typename T> struct Lg;
template <> struct Lg<float> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct Lg<unsigned short> {
  static __device__ __forceinline__ float ld(const void* p, long long i) { return sc_bf16_to_f32(((const unsigned short*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { ((unsigned short*)p)[i] = sc_f32_to_bf16(v); }
};

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

constexpr int kMaxS = 1024;  // extended label length 2L+1 handled per workgroup

struct CtcArgs {
  const void* logits;       // (B, Tmax, V)
  const float* lse;         // (B, Tmax)
  const long long* in_len;  // (B,)
  const long long* tgt;     // (B, tgt_stride) label ids
  long long tgt_stride;
  const long long* tgt_len; // (B,)
  int B, Tmax, V, Smax, blank;
  float* alpha;             // (B, Tmax, Smax)
  float* nll;               // (B,) raw negative log-likelihood (may be +inf)
  const float* gscale;      // (B,) backward only
  void* grad;               // (B, Tmax, V) backward only
};

template <typename T>
__global__ __launch_bounds__(256) void ctc_alpha_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ float buf[2][kMaxS];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = (int)p.tgt_len[b];
  const int S = 2 * L + 1;
  const int Tb = (int)min((long long)p.Tmax, p.in_len[b]);
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)p.tgt[b * p.tgt_stride + (s >> 1)] : p.blank;
  __syncthreads();
  if (Tb <= 0) {
    if (tid == 0) p.nll[b] = (L == 0) ? 0.f : INFINITY;
    return;
  }
  const long long lbase = (long long)b * p.Tmax * p.V;
  float* abase = p.alpha + (long long)b * p.Tmax * p.Smax;
  {
    const float l0 = p.lse[(long long)b * p.Tmax];
    for (int s = tid; s < S; s += 256) {
      float a = -INFINITY;
      if (s < 2) a = Lg<T>::ld(p.logits, lbase + ext[s]) - l0;
      buf[0][s] = a;
      abase[s] = a;
    }
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* prev = buf[(t - 1) & 1];
    float* cur = buf[t & 1];
    const float lt = p.lse[(long long)b * p.Tmax + t];
    for (int s = tid; s < S; s += 256) {
      const float a0 = prev[s];
      const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;
      const float a2 = (s >= 2 && (s & 1) && ext[s] != ext[s - 2]) ? prev[s - 2] : -INFINITY;
      const float v = lse3(a0, a1, a2) + (Lg<T>::ld(p.logits, lbase + (long long)t * p.V + ext[s]) - lt);
      cur[s] = v;
      abase[(long long)t * p.Smax + s] = v;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float* last = buf[(Tb - 1) & 1];
    const float ll = (S > 1) ? lse2(last[S - 1], last[S - 2]) : last[0];
    p.nll[b] = -ll;
  }
}

// grad[b][t][v] = gscale[b] * softmax(logits[b][t])[v] for t < in_len[b], else 0
template <typename T>
__global__ __launch_bounds__(256) void ctc_dense_grad_kernel(CtcArgs p) {
  const long long row = blockIdx.x;
  const int b = (int)(row / p.Tmax), t = (int)(row - (long long)b * p.Tmax);
  const float g = (t < p.in_len[b]) ? p.gscale[b] : 0.f;
  const long long base = row * p.V;
  const float l = p.lse[row];
  for (int v = threadIdx.x; v < p.V; v += 256) Lg<T>::st(p.grad, base + v, g != 0.f ? g * expf(Lg<T>::ld(p.logits, base + v) - l) : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void ctc_beta_grad_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ int nxt[kMaxS];      // next later position carrying the same (non-blank) label, or -1
  __shared__ unsigned char lead[kMaxS];
  __shared__ float buf[2][kMaxS];
  __shared__ float ab[kMaxS];
  __shared__ float red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float g = p.gscale[b];
  if (g == 0.f) return;  // infinite loss (zero_infinity) or zero upstream gradient: the dense pass already wrote zeros
  const int L = (int)p.tgt_len[b];
  const int S = 2 * L + 1;
  const int Tb = (int)min((long long)p.Tmax, p.in_len[b]);
  if (Tb <= 0) return;
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)p.tgt[b * p.tgt_stride + (s >> 1)] : p.blank;
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    int n = -1;
    unsigned char first = 1;
    if (s & 1) {
      for (int s2 = s + 2; s2 < S; s2 += 2)
        if (ext[s2] == ext[s]) { n = s2; break; }
      for (int s2 = 1; s2 < s; s2 += 2)
        if (ext[s2] == ext[s]) { first = 0; break; }
    }
    nxt[s] = n;
    lead[s] = first;
  }
  const float nll = p.nll[b];
  const long long lbase = (long long)b * p.Tmax * p.V;
  const float* abase = p.alpha + (long long)b * p.Tmax * p.Smax;
  __syncthreads();
  for (int t = Tb - 1; t >= 0; --t) {
    float* cur = buf[t & 1];
    const float* nextb = buf[(t + 1) & 1];
    const float lt = p.lse[(long long)b * p.Tmax + t];
    float blank_max = -INFINITY;
    for (int s = tid; s < S; s += 256) {
      const float lp = Lg<T>::ld(p.logits, lbase + (long long)t * p.V + ext[s]) - lt;
      float v;
      if (t == Tb - 1) {
        v = (s >= S - 2) ? lp : -INFINITY;
      } else {
        const float b0 = nextb[s];
        const float b1 = (s + 1 < S) ? nextb[s + 1] : -INFINITY;
        const float b2 = (s + 2 < S && (s & 1) && ext[s] != ext[s + 2]) ? nextb[s + 2] : -INFINITY;
        v = lse3(b0, b1, b2) + lp;
      }
      cur[s] = v;
      const float x = abase[(long long)t * p.Smax + s] + v;
      ab[s] = x;
      if (!(s & 1)) blank_max = fmaxf(blank_max, x);
    }
    // logsumexp over the blank positions (even s): workgroup max, then sum of exp
    red[tid] = blank_max;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
      __syncthreads();
    }
    const float bm = red[0];
    __syncthreads();
    float bs = 0.f;
    if (bm != -INFINITY)
      for (int s = tid * 2; s < S; s += 512) bs += expf(ab[s] - bm);
    red[tid] = bs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const long long gbase = lbase + (long long)t * p.V;
    if (tid == 0) {
      const float lpb = Lg<T>::ld(p.logits, gbase + p.blank) - lt;
      const float lc = (bm == -INFINITY) ? -INFINITY : bm + logf(red[0]);
      Lg<T>::st(p.grad, gbase + p.blank, g * (expf(lpb) - expf(lc - lpb + nll)));
    }
    // non-blank labels: the first occurrence gathers every later occurrence of the same label
    for (int s = tid; s < S; s += 256) {
      if ((s & 1) && lead[s]) {
        float lc = ab[s];
        for (int s2 = nxt[s]; s2 >= 0; s2 = nxt[s2]) lc = lse2(lc, ab[s2]);
        const float lpv = Lg<T>::ld(p.logits, gbase + ext[s]) - lt;
        Lg<T>::st(p.grad, gbase + ext[s], g * (expf(lpv) - expf(lc - lpv + nll)));
      }
    }
    __syncthreads();
  }
}

}  // namespace

SC_API int sc_ctc_fwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                      long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, float* alpha,
                      float* nll, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && nll, "sc_ctc_fwd: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_fwd: bad sizes (Smax=%d, limit %d)", Smax, kMaxS);
  SC_CHECK_ARG(blank >= 0 && blank < V, "sc_ctc_fwd: blank %d outside the vocabulary", blank);
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, alpha, nll, nullptr, nullptr};
  if (dtype == 0) hipLaunchKernelGGL(ctc_alpha_kernel<float>, dim3(B), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(ctc_alpha_kernel<unsigned short>, dim3(B), dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK("ctc_alpha_kernel");
  return SC_OK;
}

SC_API int sc_ctc_bwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                      long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank,
                      const float* alpha, const float* nll, const float* gscale, void* grad, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && nll && gscale && grad, "sc_ctc_bwd: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_bwd: bad sizes");
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, const_cast<float*>(alpha), const_cast<float*>(nll), gscale, grad};
  if (dtype == 0) {
    hipLaunchKernelGGL(ctc_dense_grad_kernel<float>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(ctc_beta_grad_kernel<float>, dim3(B), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(ctc_dense_grad_kernel<unsigned short>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(ctc_beta_grad_kernel<unsigned short>, dim3(B), dim3(256), 0, stream, a);
  }
  SC_LAUNCH_CHECK("ctc backward kernels");
  return SC_OK;
}
