// CTC loss forward / backward for gfx950 -- replaces log_softmax + F.ctc_loss(reduction='none', zero_infinity) of
// speechain/criterion/ctc.py:48-65 (reference path /root/reference/).
//
// The (B, T, V) log-softmax is never materialised: a row-wise logsumexp (sc_row_lse) gives lse[b][t] and the
// recursions gather logit[b][t][label] - lse[b][t] for the <= 2L+1 extended labels only.
//   forward : one workgroup per utterance, alpha_t(s) in LDS (double buffered), ONE barrier per frame; the emission
//             log-probabilities of the next frames are gathered from HBM two frames ahead (they do not depend on the
//             recursion), so a frame costs an LDS round trip instead of an HBM one.  All alphas are kept in a (B, T, S)
//             f32 workspace; nll[b] = -logsumexp(alpha_T(S-1), alpha_T(S-2)).
//   backward: the same recursion run backwards stores beta_t(s) (including the emission at t, as in torch's kernels) in a
//             second (B, T, S) workspace; then ONE WORKGROUP PER FRAME (B x T of them, fully parallel) writes
//               grad[t][v] = g_b * softmax[t][v]                                                       v not a label
//               grad[t][v] = g_b * (softmax[t][v] - exp(logsumexp_{s: l'_s = v}(alpha_t(s) + beta_t(s)) - lp[t][v] + nll_b))
//             and zeros beyond the utterance length.  (Before: the label columns were patched inside the sequential
//             beta kernel -- two tree reductions, ~20 barriers and un-prefetched gathers per frame, 735 us per step.)
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

// log-sum-exp on the hardware exp2 / log2 units (v_exp_f32 / v_log_f32, ~1 ulp): the recursions evaluate it once per extended label
// and frame on their serial path, where the full-precision expf / logf sequences were most of the per-frame time.
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
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
  float* beta;              // (B, Tmax, Smax) backward only
};

// Frame barrier of the recursions: only the LDS exchange of alpha / beta has to be ordered.  __syncthreads() would also wait
// for every outstanding global access (s_waitcnt vmcnt(0)) -- this frame's store of the alpha row and the emission gathers
// prefetched for the next frames -- i.e. one HBM round trip per frame on the serial path.
__device__ __forceinline__ void ctc_frame_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}


// Raw emission logits of frame t for this thread's positions s = tid + 256 i, plus the frame's logsumexp.  Deliberately free of
// arithmetic and of branches (index clamped; the kernels are instantiated for NSPT = ceil(Smax / 256) position groups): a value consumed
// inside an `if (s < S)` makes the compiler wait for the load inside that branch, i.e. at issue time -- which turns the
// two-frames-ahead prefetch into one memory round trip per frame on the serial path (seen in the ISA; 183 us per recursion).
template <typename T, int NSPT>
__device__ __forceinline__ void ctc_gather(const CtcArgs& p, long long lbase, int b, int t, const int* ext, int S, int tid,
                                           float (&raw)[NSPT], float& lt) {
  lt = p.lse[(long long)b * p.Tmax + t];
#pragma unroll
  for (int i = 0; i < NSPT; ++i) raw[i] = Lg<T>::ld(p.logits, lbase + (long long)t * p.V + ext[min(tid + 256 * i, S - 1)]);
}

template <typename T, int NSPT>
__device__ __forceinline__ void ctc_alpha_body(const CtcArgs& p, const int b, int* ext, float (*buf)[kMaxS]) {
  const int tid = threadIdx.x;
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
  bool skip2[NSPT];  // the s-2 -> s transition exists (odd s, different label)
#pragma unroll
  for (int i = 0; i < NSPT; ++i) {
    const int s = tid + 256 * i;
    skip2[i] = s < S && s >= 2 && (s & 1) && ext[s] != ext[s - 2];
  }
  // Raw logits three frames ahead in THREE register buffers used in rotation (frame t reads buffer t % 3 and, once it has consumed
  // it, loads frame t + 3 into the same buffer).  The rotation is by name -- the loop is unrolled three ways -- never by copying: a register-to-register
  // move of a value still in flight makes the compiler wait for the load in the same frame (seen in the ISA: s_waitcnt vmcnt(0)
  // in front of every frame barrier), which is one memory round trip per frame on the serial path.
  float r0[NSPT], r1[NSPT], r2[NSPT];
  float l0, l1 = 0.f, l2 = 0.f;
  ctc_gather<T, NSPT>(p, lbase, b, 0, ext, S, tid, r0, l0);
  if (Tb > 1) ctc_gather<T, NSPT>(p, lbase, b, 1, ext, S, tid, r1, l1);
  if (Tb > 2) ctc_gather<T, NSPT>(p, lbase, b, 2, ext, S, tid, r2, l2);
#pragma unroll
  for (int i = 0; i < NSPT; ++i) {
    const int s = tid + 256 * i;
    if (s < S) {
      const float a = s < 2 ? r0[i] - l0 : -INFINITY;
      buf[0][s] = a;
      abase[s] = a;
    }
  }
  if (Tb > 3) ctc_gather<T, NSPT>(p, lbase, b, 3, ext, S, tid, r0, l0);  // frame 0 is consumed: its buffer takes frame 3
  ctc_frame_barrier();
#define CTC_ALPHA_FRAME(T_, USE, LUSE, MORE)                                                   \
  do {                                                                                                \
    const int t_ = (T_);                                                                              \
    const float* prev = buf[(t_ - 1) & 1];                                                            \
    float* cur = buf[t_ & 1];                                                                         \
    float v_[NSPT];                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NSPT; ++i) {                                                \
      const int s = tid + 256 * i;                                                                    \
      v_[i] = -INFINITY;                                                                              \
      if (s < S) {                                                                                    \
        const float a0 = prev[s];                                                                     \
        const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;                                            \
        const float a2 = skip2[i] ? prev[s - 2] : -INFINITY;                                          \
        v_[i] = lse3(a0, a1, a2) + (USE[i] - LUSE);                                                   \
        cur[s] = v_[i];                                                                               \
        abase[(long long)t_ * p.Smax + s] = v_[i];                                                    \
      }                                                                                               \
    }                                                                                                 \
    if (MORE) ctc_gather<T, NSPT>(p, lbase, b, t_ + 3, ext, S, tid, USE, LUSE); /* consumed: reload */    \
    ctc_frame_barrier();                                                                              \
  } while (0)
  // steady state without a single branch (every frame and every prefetch of the triple exists: t + 5 < Tb), so that the compiler
  // keeps COUNTED vmcnt waits across the loop; the last frames run guarded
  int t = 1;
  for (; t + 5 < Tb; t += 3) {
    CTC_ALPHA_FRAME(t, r1, l1, true);
    CTC_ALPHA_FRAME(t + 1, r2, l2, true);
    CTC_ALPHA_FRAME(t + 2, r0, l0, true);
  }
  for (; t < Tb; t += 3) {
    CTC_ALPHA_FRAME(t, r1, l1, t + 3 < Tb);
    if (t + 1 < Tb) CTC_ALPHA_FRAME(t + 1, r2, l2, t + 4 < Tb);
    if (t + 2 < Tb) CTC_ALPHA_FRAME(t + 2, r0, l0, t + 5 < Tb);
  }
#undef CTC_ALPHA_FRAME
  if (tid == 0) {
    const float* last = buf[(Tb - 1) & 1];
    const float ll = (S > 1) ? lse2(last[S - 1], last[S - 2]) : last[0];
    p.nll[b] = -ll;
  }
}

// beta_t(s) (with the emission at t) for t = Tb-1 .. 0 into p.beta (B, Tmax, Smax); same structure as the alpha kernel
template <typename T, int NSPT>
__device__ __forceinline__ void ctc_beta_body(const CtcArgs& p, const int b, int* ext, float (*buf)[kMaxS]) {
  const int tid = threadIdx.x;
  const int L = (int)p.tgt_len[b];
  const int S = 2 * L + 1;
  const int Tb = (int)min((long long)p.Tmax, p.in_len[b]);
  if (Tb <= 0) return;
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)p.tgt[b * p.tgt_stride + (s >> 1)] : p.blank;
  __syncthreads();
  const long long lbase = (long long)b * p.Tmax * p.V;
  float* bbase = p.beta + (long long)b * p.Tmax * p.Smax;
  bool skip2[NSPT];  // the s -> s+2 transition exists
#pragma unroll
  for (int i = 0; i < NSPT; ++i) {
    const int s = tid + 256 * i;
    skip2[i] = s + 2 < S && (s & 1) && ext[s] != ext[s + 2];
  }
  // three register buffers rotated by name, three frames ahead (see ctc_alpha_kernel); frames run Tb-1, Tb-2, ...: frame
  // Tb-1-j reads buffer j % 3
  float r0[NSPT], r1[NSPT], r2[NSPT];
  float l0, l1 = 0.f, l2 = 0.f;
  const int tl = Tb - 1;
  ctc_gather<T, NSPT>(p, lbase, b, tl, ext, S, tid, r0, l0);
  if (tl >= 1) ctc_gather<T, NSPT>(p, lbase, b, tl - 1, ext, S, tid, r1, l1);
  if (tl >= 2) ctc_gather<T, NSPT>(p, lbase, b, tl - 2, ext, S, tid, r2, l2);
#pragma unroll
  for (int i = 0; i < NSPT; ++i) {
    const int s = tid + 256 * i;
    if (s < S) {
      const float v = s >= S - 2 ? r0[i] - l0 : -INFINITY;
      buf[tl & 1][s] = v;
      bbase[(long long)tl * p.Smax + s] = v;
    }
  }
  if (tl >= 3) ctc_gather<T, NSPT>(p, lbase, b, tl - 3, ext, S, tid, r0, l0);
  ctc_frame_barrier();
#define CTC_BETA_FRAME(T_, USE, LUSE, MORE)                                                                 \
  do {                                                                                                \
    const int t_ = (T_);                                                                              \
    const float* nextb = buf[(t_ + 1) & 1];                                                           \
    float* cur = buf[t_ & 1];                                                                         \
    _Pragma("unroll") for (int i = 0; i < NSPT; ++i) {                                                \
      const int s = tid + 256 * i;                                                                    \
      if (s < S) {                                                                                    \
        const float b0 = nextb[s];                                                                    \
        const float b1 = s + 1 < S ? nextb[s + 1] : -INFINITY;                                        \
        const float b2 = skip2[i] ? nextb[s + 2] : -INFINITY;                                         \
        const float v = lse3(b0, b1, b2) + (USE[i] - LUSE);                                           \
        cur[s] = v;                                                                                   \
        bbase[(long long)t_ * p.Smax + s] = v;                                                        \
      }                                                                                               \
    }                                                                                                 \
    if (MORE) ctc_gather<T, NSPT>(p, lbase, b, t_ - 3, ext, S, tid, USE, LUSE);                            \
    ctc_frame_barrier();                                                                              \
  } while (0)
  int t = tl - 1;
  for (; t - 5 >= 0; t -= 3) {  // branch-free steady state (see ctc_alpha_kernel)
    CTC_BETA_FRAME(t, r1, l1, true);
    CTC_BETA_FRAME(t - 1, r2, l2, true);
    CTC_BETA_FRAME(t - 2, r0, l0, true);
  }
  for (; t >= 0; t -= 3) {
    CTC_BETA_FRAME(t, r1, l1, t - 3 >= 0);
    if (t - 1 >= 0) CTC_BETA_FRAME(t - 1, r2, l2, t - 4 >= 0);
    if (t - 2 >= 0) CTC_BETA_FRAME(t - 2, r0, l0, t - 5 >= 0);
  }
#undef CTC_BETA_FRAME
}

template <typename T, int NSPT>
__global__ __launch_bounds__(256) void ctc_alpha_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ float buf[2][kMaxS];
  ctc_alpha_body<T, NSPT>(p, blockIdx.x, ext, buf);
}

template <typename T, int NSPT>
__global__ __launch_bounds__(256) void ctc_beta_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ float buf[2][kMaxS];
  if (p.gscale[blockIdx.x] == 0.f) return;  // infinite loss (zero_infinity) or zero upstream gradient: the gradient is all zeros
  ctc_beta_body<T, NSPT>(p, blockIdx.x, ext, buf);
}

// Both recursions of a training step in ONE launch of 2 B workgroups (they are independent and each is a serial chain of Tb frames on one
// workgroup: launched one after the other they cost two chains, together one).  Workgroups [0, B) run alpha, [B, 2B) beta.
template <typename T, int NSPT>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ float buf[2][kMaxS];
  if ((int)blockIdx.x < p.B) ctc_alpha_body<T, NSPT>(p, blockIdx.x, ext, buf);
  else ctc_beta_body<T, NSPT>(p, (int)blockIdx.x - p.B, ext, buf);
}

// One workgroup per frame (b, t): the whole gradient row.
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(CtcArgs p) {
  __shared__ int ext[kMaxS];
  __shared__ float ab[kMaxS];
  __shared__ unsigned islabel[(1 << 16) / 32];  // bitmap over the vocabulary (V <= 65536: checked on the host)
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const int b = (int)(row / p.Tmax), t = (int)(row - (long long)b * p.Tmax);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float g = (t < p.in_len[b]) ? p.gscale[b] : 0.f;
  const long long base = row * p.V;
  // bf16 rows of a multiple of 8 classes move as 16-byte pieces (a 2-byte access per lane and instruction is instruction-bound)
  const bool vec = sizeof(T) == 2 && (p.V & 7) == 0;
  if (g == 0.f) {
    if (vec) {
      uint4* gr = reinterpret_cast<uint4*>((unsigned short*)p.grad + base);
      for (int c = tid; c < (p.V >> 3); c += 256) gr[c] = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (int v = tid; v < p.V; v += 256) Lg<T>::st(p.grad, base + v, 0.f);
    }
    return;
  }
  const int L = (int)p.tgt_len[b];
  const int S = 2 * L + 1;
  const float lt = p.lse[row];
  const float nll = p.nll[b];
  for (int w = tid; w < (p.V + 31) / 32; w += 256) islabel[w] = 0u;
  const float* arow = p.alpha + ((long long)b * p.Tmax + t) * p.Smax;
  const float* brow = p.beta + ((long long)b * p.Tmax + t) * p.Smax;
  float blank_max = -INFINITY;
  for (int s = tid; s < S; s += 256) {
    ext[s] = (s & 1) ? (int)p.tgt[b * p.tgt_stride + (s >> 1)] : p.blank;
    const float x = arow[s] + brow[s];
    ab[s] = x;
    if (!(s & 1)) blank_max = fmaxf(blank_max, x);
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256) atomicOr(&islabel[ext[s] >> 5], 1u << (ext[s] & 31));
  // logsumexp over the blank positions (even s)
  blank_max = sc_wave_max(blank_max);
  if (lane == 0) red[wid] = blank_max;
  __syncthreads();
  const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float bs = 0.f;
  if (bm != -INFINITY)
    for (int s = tid * 2; s < S; s += 512) bs += expf(ab[s] - bm);
  bs = sc_wave_sum(bs);
  if (lane == 0) red[4 + wid] = bs;
  __syncthreads();
  if (tid == 0) {
    const float lpb = Lg<T>::ld(p.logits, base + p.blank) - lt;
    const float lc = (bm == -INFINITY) ? -INFINITY : bm + logf(red[4] + red[5] + red[6] + red[7]);
    Lg<T>::st(p.grad, base + p.blank, g * (expf(lpb) - expf(lc - lpb + nll)));
  }
  // non-blank labels: the first occurrence gathers every later occurrence of the same label
  for (int s = tid; s < S; s += 256) {
    if (s & 1) {
      bool first = true;
      for (int s2 = 1; s2 < s; s2 += 2)
        if (ext[s2] == ext[s]) { first = false; break; }
      if (first) {
        float lc = ab[s];
        for (int s2 = s + 2; s2 < S; s2 += 2)
          if (ext[s2] == ext[s]) lc = lse2(lc, ab[s2]);
        const float lpv = Lg<T>::ld(p.logits, base + ext[s]) - lt;
        Lg<T>::st(p.grad, base + ext[s], g * (expf(lpv) - expf(lc - lpv + nll)));
      }
    }
  }
  // every other class: g * softmax
  if (vec) {
    const uint4* lg = reinterpret_cast<const uint4*>((const unsigned short*)p.logits + base);
    unsigned short* gr = (unsigned short*)p.grad + base;
    for (int c = tid; c < (p.V >> 3); c += 256) {
      const uint4 u = lg[c];
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
      const unsigned bits = (islabel[c >> 2] >> ((c & 3) * 8)) & 0xFFu;  // the piece's 8 classes lie in one bitmap word
      unsigned short o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = sc_f32_to_bf16(g * expf(__uint_as_float(w[e] << 16) - lt));
        o[2 * e + 1] = sc_f32_to_bf16(g * expf(__uint_as_float(w[e] & 0xffff0000u) - lt));
      }
      if (bits == 0u) {
        *reinterpret_cast<uint4*>(gr + 8 * c) = make_uint4(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16),
                                                           o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16));
      } else {  // label columns are written by the threads above
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!((bits >> e) & 1u)) gr[8 * c + e] = o[e];
      }
    }
    return;
  }
  for (int v = tid; v < p.V; v += 256)
    if (!((islabel[v >> 5] >> (v & 31)) & 1u)) Lg<T>::st(p.grad, base + v, g * expf(Lg<T>::ld(p.logits, base + v) - lt));
}

}  // namespace

SC_API int sc_ctc_fwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                      long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, float* alpha,
                      float* nll, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && nll, "sc_ctc_fwd: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_fwd: bad sizes (Smax=%d, limit %d)", Smax, kMaxS);
  SC_CHECK_ARG(blank >= 0 && blank < V, "sc_ctc_fwd: blank %d outside the vocabulary", blank);
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, alpha, nll, nullptr, nullptr};
  // NSPT = extended-label positions per thread the recursion kernels are instantiated for (Smax <= 256 NSPT)
#define SC_CTC_RECURSION(KERNEL, NB)                                                                                           \
  do {                                                                                                                       \
    if (dtype == 0) {                                                                                                        \
      if (Smax <= 256) hipLaunchKernelGGL((KERNEL<float, 1>), dim3(NB), dim3(256), 0, stream, a);                              \
      else if (Smax <= 512) hipLaunchKernelGGL((KERNEL<float, 2>), dim3(NB), dim3(256), 0, stream, a);                         \
      else hipLaunchKernelGGL((KERNEL<float, 4>), dim3(NB), dim3(256), 0, stream, a);                                          \
    } else {                                                                                                                 \
      if (Smax <= 256) hipLaunchKernelGGL((KERNEL<unsigned short, 1>), dim3(NB), dim3(256), 0, stream, a);                     \
      else if (Smax <= 512) hipLaunchKernelGGL((KERNEL<unsigned short, 2>), dim3(NB), dim3(256), 0, stream, a);                \
      else hipLaunchKernelGGL((KERNEL<unsigned short, 4>), dim3(NB), dim3(256), 0, stream, a);                                 \
    }                                                                                                                        \
  } while (0)
  SC_CTC_RECURSION(ctc_alpha_kernel, B);
  SC_LAUNCH_CHECK("ctc_alpha_kernel");
  return SC_OK;
}

SC_API int sc_ctc_fwd_train(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                            long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, float* alpha,
                            float* beta, float* nll, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && beta && nll, "sc_ctc_fwd_train: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_fwd_train: bad sizes (Smax=%d, limit %d)", Smax, kMaxS);
  SC_CHECK_ARG(blank >= 0 && blank < V, "sc_ctc_fwd_train: blank %d outside the vocabulary", blank);
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, alpha, nll, nullptr, nullptr, beta};
  SC_CTC_RECURSION(ctc_alpha_beta_kernel, 2 * B);
  SC_LAUNCH_CHECK("ctc_alpha_beta_kernel");
  return SC_OK;
}

SC_API int sc_ctc_grad(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                       long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, const float* alpha,
                       const float* beta, const float* nll, const float* gscale, void* grad, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && nll && gscale && beta && grad, "sc_ctc_grad: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_grad: bad sizes");
  SC_CHECK_ARG(V <= (1 << 16), "sc_ctc_grad: vocabulary %d > 65536 (label bitmap in LDS)", V);
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, const_cast<float*>(alpha), const_cast<float*>(nll), gscale, grad,
            const_cast<float*>(beta)};
  if (dtype == 0) hipLaunchKernelGGL(ctc_grad_kernel<float>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(ctc_grad_kernel<unsigned short>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK("ctc_grad_kernel");
  return SC_OK;
}

SC_API int sc_ctc_bwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                      long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank,
                      const float* alpha, const float* nll, const float* gscale, float* beta, void* grad, hipStream_t stream) {
  SC_CHECK_ARG(logits && lse && in_len && tgt && tgt_len && alpha && nll && gscale && beta && grad, "sc_ctc_bwd: null pointer");
  SC_CHECK_ARG(B >= 1 && Tmax >= 1 && V >= 1 && Smax >= 1 && Smax <= kMaxS, "sc_ctc_bwd: bad sizes");
  SC_CHECK_ARG(V <= (1 << 16), "sc_ctc_bwd: vocabulary %d > 65536 (label bitmap in LDS)", V);
  CtcArgs a{logits, lse, in_len, tgt, tgt_stride, tgt_len, B, Tmax, V, Smax, blank, const_cast<float*>(alpha), const_cast<float*>(nll), gscale, grad, beta};
  if (dtype == 0) {
    SC_CTC_RECURSION(ctc_beta_kernel, B);
    hipLaunchKernelGGL(ctc_grad_kernel<float>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
  } else {
    SC_CTC_RECURSION(ctc_beta_kernel, B);
    hipLaunchKernelGGL(ctc_grad_kernel<unsigned short>, dim3((unsigned)((long long)B * Tmax)), dim3(256), 0, stream, a);
  }
  SC_LAUNCH_CHECK("ctc backward kernels");
  return SC_OK;
}
