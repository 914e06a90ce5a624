// Cached-key/value decoding for the autoregressive ASR decoder on gfx950.
//
// What it replaces in the reference (paths under /root/reference/): beam_searching (speechain/infer_func/beam_search.py:325-333) calls
// the whole decoder on the whole growing prefix of every hypothesis at every step -- O(L^2) decoder passes per utterance -- and re-orders
// the hypotheses by indexing every per-row tensor with beam_idx (:518-521).  Here a step feeds ONE position per hypothesis:
//   * the keys and values of earlier positions stay where they were written, (row, position) slots of a (N, Tmax, 2 d) cache per layer;
//   * a hypothesis is a list of slots: table[n][t] = the row whose slot t holds position t of hypothesis n.  Re-ordering the beams
//     copies table rows (N x Tmax ints), never keys or values;
//   * sc_attention_decode: one wave per (hypothesis, head) appends the step's key / value, scores the query against the history read
//     through the table, and forms the context.  Byte-moving (each history row is read once per step): HBM / L2 bound.
// Cross-attention needs none of this: its keys / values are per SENTENCE and the beams of a sentence are adjacent rows, so the
// ordinary fused attention kernel runs with (sentences, beam) as (batch, queries).
#include "sc_common.h"

namespace {

constexpr int DH = 64;  // head size (one lane per head dimension)

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  // dot of 64 contiguous elements with q (64 floats in LDS)
  static __device__ __forceinline__ float dot64(const float* row, const float* q) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
      const float4 a = *(const float4*)(row + j);
      const float4 b = *(const float4*)(q + j);
      s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return s;
  }
};
template <> struct Elem<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short* p) { return sc_bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = sc_f32_to_bf16(v); }
  static __device__ __forceinline__ float dot64(const unsigned short* row, const float* q) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j += 8) {
      const uint4 a = *(const uint4*)(row + j);
      const float4 b0 = *(const float4*)(q + j), b1 = *(const float4*)(q + j + 4);
      s += __uint_as_float(a.x << 16) * b0.x + __uint_as_float(a.x & 0xFFFF0000u) * b0.y;
      s += __uint_as_float(a.y << 16) * b0.z + __uint_as_float(a.y & 0xFFFF0000u) * b0.w;
      s += __uint_as_float(a.z << 16) * b1.x + __uint_as_float(a.z & 0xFFFF0000u) * b1.y;
      s += __uint_as_float(a.w << 16) * b1.z + __uint_as_float(a.w & 0xFFFF0000u) * b1.w;
    }
    return s;
  }
};

// table_out[n][t] = table_in[parent(n)][t] for t < pos, table_out[n][pos] = n.  grid (ceil(N / 4)); one wave per row.
__global__ __launch_bounds__(256) void beam_table_kernel(const int* tin, int* tout, const long long* beam_idx, int N, int Tmax, int pos) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const long long par = beam_idx ? beam_idx[n] : n;
  for (int t = lane; t < pos; t += 64) tout[(long long)n * Tmax + t] = tin[par * Tmax + t];
  if (lane == 0) tout[(long long)n * Tmax + pos] = n;
}

// The head of a decoding step in one launch (one wave per hypothesis row): the ancestry table as beam_table_kernel writes it, and the row's
// decoder input x[n] = round_T(emb[token[n]] * emb_scale) * xscale + pe * alpha -- nn.Embedding lookup (prenet/embed.py:40-47), the cast to
// the compute type, and the positional encoding of position `pos` (transformer/pos_enc.py:155-163, evaluation mode).  As five launches
// (lookup, scale, cast, encoding, table) on the branches of a replayed hipGraph these cost ~150 us of mostly idle device per position.
template <typename T>
__global__ __launch_bounds__(256) void decode_embed_kernel(const float* emb, const long long* tokens, float emb_scale, const float* pe, const float* alpha,
                                                           float xscale, T* x, const int* tin, int* tout, const long long* beam_idx, int N, int d,
                                                           int Tmax, int pos) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const long long par = beam_idx ? beam_idx[n] : n;
  for (int t = lane; t < pos; t += 64) tout[(long long)n * Tmax + t] = tin[par * Tmax + t];
  if (lane == 0) tout[(long long)n * Tmax + pos] = n;
  const float* erow = emb + tokens[n] * (long long)d;
  const float a = alpha ? *alpha : 1.f;
  for (int k = lane; k < d; k += 64) {
    float e = erow[k] * emb_scale;
    if (sizeof(T) == 2) e = sc_bf16_to_f32(sc_f32_to_bf16(e));
    const float v = e * xscale + pe[k] * a;
    if (sizeof(T) == 2) ((unsigned short*)x)[(long long)n * d + k] = sc_f32_to_bf16(v);
    else ((float*)x)[(long long)n * d + k] = v;
  }
}

// 16 bytes of a row as EPL floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int EPL = 4;
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    const float4 a = *(const float4*)p;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  static __device__ __forceinline__ void st(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
  static __device__ __forceinline__ uint4 raw(const float* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ void unpack(const uint4& a, float* v) {
    v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
  }
};
template <> struct Chunk<unsigned short> {
  static constexpr int EPL = 8;
  static __device__ __forceinline__ uint4 raw(const unsigned short* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ void unpack(const uint4& a, float* v) {
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(w[j] << 16);
      v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
  }
  static __device__ __forceinline__ void ld(const unsigned short* p, float* v) { unpack(raw(p), v); }
  static __device__ __forceinline__ void st(unsigned short* p, const float* v) {
    uint4 a;
    a.x = sc_f32_to_bf16(v[0]) | ((unsigned)sc_f32_to_bf16(v[1]) << 16);
    a.y = sc_f32_to_bf16(v[2]) | ((unsigned)sc_f32_to_bf16(v[3]) << 16);
    a.z = sc_f32_to_bf16(v[4]) | ((unsigned)sc_f32_to_bf16(v[5]) << 16);
    a.w = sc_f32_to_bf16(v[6]) | ((unsigned)sc_f32_to_bf16(v[7]) << 16);
    *(uint4*)p = a;
  }
};

// Sum over aligned groups of G lanes (8 or 16), DPP only; the group's last lane holds the total (every lane of a group of 8 does).
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += sc_dpp<0xB1, 0xF>(0.f, v);   // quad_perm [1,0,3,2]
  v += sc_dpp<0x4E, 0xF>(0.f, v);   // quad_perm [2,3,0,1]
  v += sc_dpp<0x141, 0xF>(0.f, v);  // row_half_mirror
  if (G == 16) v += sc_dpp<0x140, 0xF>(0.f, v);  // row_mirror
  return v;
}

// qkv: (N, 3 d) this step's projections [q | k | v]; cache: (N, Tmax, 2 d) [k | v]; table: (N, Tmax); out: (N, d).
// One wave per (row, head), 4 per workgroup (see the XCD placement below).  A key's 64 head dimensions are spread over LPK lanes of 16 bytes each, so that every
// load instruction reads KPI = 64 / LPK whole rows (8 for bf16, 4 for f32); the probabilities never leave the lanes that multiply the values.
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* qkv, long long ldqkv, T* cache, const int* table, T* out, long long ldo,
                                                          int N, int H, int Tmax, int pos, float scale) {
  constexpr int EPL = Chunk<T>::EPL, LPK = DH / EPL, KPI = 64 / LPK;
  extern __shared__ __attribute__((aligned(16))) float dsm_dec[];  // per wave: q[64], v_new[64], rows[Tmax]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // Workgroups go to the 8 XCDs round-robin.  The rows of one sentence's beam are neighbours and share most of their ancestry, so the 16 rows
  // n0 .. n0 + 15 of one head (4 workgroups) are placed on one XCD back to back: the shared cache rows then come out of that XCD's L2.
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int g = (seq >> 2) * 8 + xcd;  // (chunk of 16 rows, head)
  const int n = (g / H) * 16 + (seq & 3) * 4 + w, h = g % H;
  if (n >= N) return;
  float* qs = dsm_dec + (size_t)w * (2 * DH + Tmax);
  float* vs = qs + DH;
  int* rs = (int*)(vs + DH);
  const int d = H * DH;
  // The kernel is a chain of memory round trips, so there are as few as the data flow allows: [this step's q, k, v and the ancestry row],
  // then [keys and values of the first two batches], then one batch ahead of the arithmetic.
  const int* trow = table + (long long)n * Tmax;
  const int t0 = lane < pos ? trow[lane] : 0;
  const T* row = qkv + (long long)n * ldqkv + h * DH;
  const float q = Elem<T>::ld(row + lane), kn = Elem<T>::ld(row + d + lane), vn = Elem<T>::ld(row + 2 * d + lane);
  if (lane < pos) rs[lane] = t0;
  for (int t = lane + 64; t < pos; t += 64) rs[t] = trow[t];
  // append this position: slot (n, pos)
  T* slot = cache + ((long long)n * Tmax + pos) * (2 * d) + h * DH;
  Elem<T>::st(slot + lane, kn);
  Elem<T>::st(slot + d + lane, vn);
  qs[lane] = q;
  vs[lane] = vn;
  asm volatile("" ::: "memory");  // (the wave reads its own LDS rows below; LDS operations of one wave complete in order)
  const int sub = lane % LPK, grp = lane / LPK;
  // A key's 64 head dimensions are spread over LPK lanes of 16 bytes each: one load instruction reads KPI whole rows (8 for bf16, 4 for
  // f32), a batch is U instructions for the keys and U for the values.  Lanes past the history re-read key pos - 1 and give it the score
  // -inf, which keeps the loads free of branches.
  constexpr int U = 4, KPB = KPI * U;
  const T* kbase = cache + h * DH + sub * EPL;
  auto fetch = [&](int k0, uint4 (&kd)[U], uint4 (&vd)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = min(k0 + u * KPI + grp, pos - 1);
      const T* p = kbase + ((long long)rs[kc] * Tmax + kc) * (2 * d);
      kd[u] = Chunk<T>::raw(p);
      vd[u] = Chunk<T>::raw(p + d);
    }
  };
  uint4 kb[U], vb[U];
  if (pos > 0) fetch(0, kb, vb);
  float qc[EPL], acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    qc[e] = qs[sub * EPL + e];
    acc[e] = grp == 0 ? vs[sub * EPL + e] : 0.f;  // the new position (probability 1 relative to the running maximum, its own score)
  }
  float m = sc_wave_sum(q * kn) * scale;  // the new position's own score (its key is still in registers)
  float l = grp == 0 ? 1.f : 0.f;         // every lane of a group counts the group's keys: the wave sum below is LPK times the normaliser
  // ---- the history, one batch at a time with a running maximum (the probabilities stay in the lanes that multiply the values) ----
  for (int k0 = 0; k0 < pos; k0 += KPB) {
    uint4 kx[U], vx[U];
    const bool more = k0 + KPB < pos;
    if (more) fetch(k0 + KPB, kx, vx);
    float s[U], bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kv[EPL], part = 0.f;
      Chunk<T>::unpack(kb[u], kv);
#pragma unroll
      for (int e = 0; e < EPL; ++e) part = fmaf(kv[e], qc[e], part);
      part = group_sum<LPK>(part) * scale;
      s[u] = k0 + u * KPI + grp < pos ? part : -INFINITY;
      bm = fmaxf(bm, s[u]);
    }
    const float mn = fmaxf(m, sc_wave_max(bm));
    const float corr = __expf(m - mn);
    m = mn;
    l *= corr;
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] *= corr;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pk = __expf(s[u] - m);
      float vv[EPL];
      Chunk<T>::unpack(vb[u], vv);
      l += pk;
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        kb[u] = kx[u];
        vb[u] = vx[u];
      }
    }
  }
  l = sc_wave_sum(l) * (1.f / LPK);
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (grp == 0) {
    const float inv = 1.f / l;
    float o8[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o8[e] = acc[e] * inv;
    Chunk<T>::st(out + (long long)n * ldo + h * DH + sub * EPL, o8);
  }
}

}  // namespace

// Ancestry table of a beam-search step (speechain/infer_func/beam_search.py:518-521 re-orders rows by beam_idx instead):
// table_out[n][t] = table_in[beam_idx[n]][t] for t < pos (beam_idx NULL = identity), table_out[n][pos] = n.  Tables are (N, Tmax) int32.
SC_API int sc_beam_table_update(const int* table_in, int* table_out, const long long* beam_idx, int N, int Tmax, int pos,
                                hipStream_t stream) {
  SC_CHECK_ARG(table_out && (table_in || pos == 0), "sc_beam_table_update: null pointer");
  SC_CHECK_ARG(table_in != table_out, "sc_beam_table_update: the tables must be two buffers");
  SC_CHECK_ARG(N >= 1 && Tmax >= 1 && pos >= 0 && pos < Tmax, "sc_beam_table_update: bad sizes N=%d Tmax=%d pos=%d", N, Tmax, pos);
  hipLaunchKernelGGL(beam_table_kernel, dim3(sc_cdiv(N, 4)), dim3(256), 0, stream, table_in, table_out, beam_idx, N, Tmax, pos);
  SC_LAUNCH_CHECK("beam_table_kernel");
  return SC_OK;
}

// sc_beam_table_update plus the decoder input rows of the step (see decode_embed_kernel): emb (V, d) f32 embedding table, tokens (N) the
// newest token of every row, pe (d) the positional-encoding row of `pos`, alpha its learnt scale (device scalar) or null, x (N, d) of `dtype`.
SC_API int sc_decode_embed_step(int dtype, const float* emb, const long long* tokens, float emb_scale, const float* pe, const float* alpha,
                                float xscale, void* x, const int* table_in, int* table_out, const long long* beam_idx, int N, int d, int V, int Tmax,
                                int pos, hipStream_t stream) {
  SC_CHECK_ARG(emb && tokens && pe && x && table_out && (table_in || pos == 0), "sc_decode_embed_step: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_decode_embed_step: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(table_in != table_out, "sc_decode_embed_step: the tables must be two buffers");
  SC_CHECK_ARG(N >= 1 && d >= 1 && V >= 1 && Tmax >= 1 && pos >= 0 && pos < Tmax, "sc_decode_embed_step: bad sizes N=%d d=%d Tmax=%d pos=%d", N, d, Tmax, pos);
  if (dtype == 0)
    hipLaunchKernelGGL(decode_embed_kernel<float>, dim3(sc_cdiv(N, 4)), dim3(256), 0, stream, emb, tokens, emb_scale, pe, alpha, xscale, (float*)x, table_in,
                       table_out, beam_idx, N, d, Tmax, pos);
  else
    hipLaunchKernelGGL(decode_embed_kernel<unsigned short>, dim3(sc_cdiv(N, 4)), dim3(256), 0, stream, emb, tokens, emb_scale, pe, alpha, xscale,
                       (unsigned short*)x, table_in, table_out, beam_idx, N, d, Tmax, pos);
  SC_LAUNCH_CHECK("decode_embed_kernel");
  return SC_OK;
}

// One decoding step of causal self-attention with cached keys / values (heads of 64).  dtype: 0 = f32, 1 = bf16.
//   qkv   (N, >= 3 d) row stride ldqkv: this step's [q | k | v] projections of the N hypothesis rows;
//   cache (N, Tmax, 2 d) [k | v]: slot (n, pos) is written here, slots (table[n][t], t), t < pos, are read;
//   out   (N, d) row stride ldo: softmax(scale q K^T) V per head.
SC_API int sc_attention_decode(int dtype, const void* qkv, long long ldqkv, void* cache, const int* table, void* out, long long ldo, int N,
                               int H, int Tmax, int pos, float scale, hipStream_t stream) {
  SC_CHECK_ARG(qkv && cache && table && out, "sc_attention_decode: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_attention_decode: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(N >= 1 && H >= 1 && Tmax >= 1 && Tmax <= 512 && pos >= 0 && pos < Tmax, "sc_attention_decode: bad sizes N=%d H=%d Tmax=%d (<= 512) pos=%d",
               N, H, Tmax, pos);
  SC_CHECK_ARG((long long)(N + 15) * H <= 0x0fffffffLL, "sc_attention_decode: too many (row, head) pairs");
  const long long d = (long long)H * DH;
  SC_CHECK_ARG(ldqkv >= 3 * d && ldo >= d, "sc_attention_decode: row strides too small");
  SC_CHECK_ARG(((uintptr_t)cache | (uintptr_t)qkv) % 16 == 0 && ldqkv % 8 == 0, "sc_attention_decode: buffers must be 16-byte aligned");
  SC_CHECK_ARG(ldo % 8 == 0 && (uintptr_t)out % 16 == 0, "sc_attention_decode: the output must be 16-byte aligned with a row stride multiple of 8");
  const dim3 grid((unsigned)(sc_cdiv(sc_cdiv(N, 16) * H, 8) * 8 * 4));  // (chunks of 16 rows x heads) rounded up to the 8 XCDs, 4 workgroups each
  const size_t smem = (size_t)4 * (2 * DH + Tmax) * sizeof(float);
  if (dtype == 0)
    hipLaunchKernelGGL(attn_decode_kernel<float>, grid, dim3(256), smem, stream, (const float*)qkv, ldqkv, (float*)cache, table, (float*)out, ldo,
                       N, H, Tmax, pos, scale);
  else
    hipLaunchKernelGGL(attn_decode_kernel<unsigned short>, grid, dim3(256), smem, stream, (const unsigned short*)qkv, ldqkv,
                       (unsigned short*)cache, table, (unsigned short*)out, ldo, N, H, Tmax, pos, scale);
  SC_LAUNCH_CHECK("attn_decode_kernel");
  return SC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// CTC prefix scores (speechain/infer_func/ctc_decoding.py:76-160; Algorithm 2 of Watanabe et al. 2017).
//
// For hypothesis row n with prefix g and candidate token c (h = g + c), from the forward variables of g over the frames,
// r_prev[t][0 | 1][n] = log prob of g ending in (its last label | blank) at frame t:
//   r_sum[t] = logaddexp(r_prev[t][0], r_prev[t][1]);   phi[t] = (c == last label of g) ? r_prev[t][1] : r_sum[t]
//   psi(n, c) = log( exp(r^n_{start-1}(h)) + sum_{t = start}^{T-1} exp(phi[t-1] + x[t][c]) ),  start = max(1, |g|),
//   r^n_{start-1}(h) = x[0][c] for the empty prefix, impossible otherwise;  psi(n, <eos>) = r_sum[eos_frame[n]];  psi(n, blank) impossible.
// "Impossible" is the reference's finite -1e20, so that sums of such terms stay finite exactly as they do there.  Frames past an
// utterance count as certain blank (x = 0 for blank, -1e20 for labels, :49-57).
//
// The reference builds r(h) for ALL candidates -- a (T, 2, N, V) tensor per step -- because the next step needs the rows of the
// candidates that survive.  psi itself needs no recursion (only phi, which belongs to g), so ctc_psi_kernel is a masked logsumexp over the
// frames, one candidate per lane, x read once per sentence and frame with unit stride over the vocabulary; ctc_state_kernel then runs
// the two-line recursion (:132-142) for the N surviving (parent row, token) pairs only.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr float CTC_IMPOSSIBLE = -1e20f;

__device__ __forceinline__ float logaddexpf(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return m;  // (both empty: the identity of the scans below)
  return m + log1pf(__expf(fminf(a, b) - m));
}

// grid (ceil(V / 256), N / RPB); thread = candidate c of the RPB hypothesis rows RPB blockIdx.y .. + RPB - 1, all of ONE sentence
// (RPB divides the beam): x[t][c] is loaded once and serves the RPB rows, whose phi / r_sum rows sit in LDS.  Without that sharing the
// 16 beams of a sentence, running on other CUs, each pull the sentence's (T, V) block through the memory system again (measured: 866 us
// per step at N 512, V 5000, T 249 -- 2.5 GB of reads -- against the exp-bound ~200 us).
template <int RPB>
__global__ __launch_bounds__(256) void ctc_psi_kernel(const float* x, const int* enc_lens, const int* eos_frame, const float* r_prev,
                                                      const long long* last, float* psi, int N, int T, int V, int beam, int plen,
                                                      int blank, int eos) {
  extern __shared__ float s_phi[];  // [RPB][2][T]: r_sum, r_blank
  const int n0 = blockIdx.y * RPB, b = n0 / beam;
  for (int i = threadIdx.x; i < RPB * T; i += 256) {
    const int r = i / T, t = i - r * T;
    const float r0 = r_prev[((long long)t * 2 + 0) * N + n0 + r], r1 = r_prev[((long long)t * 2 + 1) * N + n0 + r];
    s_phi[(r * 2 + 0) * T + t] = logaddexpf(r0, r1);
    s_phi[(r * 2 + 1) * T + t] = r1;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= V) return;
  const int len = enc_lens[b];
  const float* xc = x + (long long)b * T * V + c;
  const int start = plen > 1 ? plen : 1;
  const float pad_x = (c == blank) ? 0.f : CTC_IMPOSSIBLE;
  const float* phi[RPB];
  float m[RPB], s[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const long long lastc = plen > 0 ? last[n0 + r] : 0;  // (the reference takes token 0 for the empty prefix, :84)
    phi[r] = s_phi + (r * 2 + (c == lastc ? 1 : 0)) * T;
    m[r] = (plen == 0) ? ((0 < len) ? xc[0] : pad_x) : CTC_IMPOSSIBLE;  // first term r^n_{start-1}(h)
  }
  // two sweeps over the frames instead of a running (max, sum) pair: the maximum first (add + max per term), then the sum of
  // exp(term - max) (add, sub, exp, add) -- about a third fewer vector instructions per term, and x comes out of L2 the second time
  for (int t0 = start; t0 < T; t0 += 8) {  // (eight frames' loads in flight per lane)
    float xl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xl[u] = (t0 + u < len) ? xc[(long long)(t0 + u) * V] : pad_x;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + u < T) {
#pragma unroll
        for (int r = 0; r < RPB; ++r) m[r] = fmaxf(m[r], phi[r][t0 + u - 1] + xl[u]);
      }
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const float first = (plen == 0) ? ((0 < len) ? xc[0] : pad_x) : CTC_IMPOSSIBLE;
    s[r] = __expf(first - m[r]);
  }
  for (int t0 = start; t0 < T; t0 += 8) {
    float xl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xl[u] = (t0 + u < len) ? xc[(long long)(t0 + u) * V] : pad_x;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + u < T) {
#pragma unroll
        for (int r = 0; r < RPB; ++r) s[r] += __expf(phi[r][t0 + u - 1] + xl[u] - m[r]);
      }
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    float out = m[r] + __logf(s[r]);
    if (c == eos) out = s_phi[(r * 2 + 0) * T + eos_frame[n0 + r]];
    if (c == blank) out = CTC_IMPOSSIBLE;
    psi[(long long)(n0 + r) * V + c] = out;
  }
}

// The same scores with the exponential factorised: exp(phi[t-1] + x[t][c]) = exp(phi[t-1] - a) * exp(x[t][c] - bmax[c]), the second factor
// E (B, T, V) computed ONCE per batch (sc_ctc_prefix_scores_fast), the first (RPB x 2 x T values) once per workgroup -- the sweep over the frames is
// one multiply-add per (row, candidate, frame) instead of add / sub / exp / add after a separate maximum sweep.
//   psi = a + bmax[c] + log( exp(first - a - bmax[c]) + sum_t P[t-1] E[t][c] ),   a = max of the phi row over the frames the sum reads.
// A sum that underflows (its largest term more than ~87 below a + bmax[c]: hopeless candidates, or rows that are impossible altogether)
// is redone for that (row, candidate) by the exact two-sweep form, so every score is the exact kernel's up to f32 rounding of the sum.
template <int RPB>
__global__ __launch_bounds__(256) void ctc_psi_fast_kernel(const float* x, const float* E, const float* bmax, const int* enc_lens,
                                                           const int* eos_frame, const float* r_prev, const long long* last, float* psi, int N,
                                                           int T, int V, int beam, int plen, int blank, int eos) {
  extern __shared__ float s_all[];  // [RPB][2][T] phi (r_sum, r_blank), then [RPB][2][T] P = exp(phi - a), then a[RPB][2]
  float* s_phi = s_all;
  float* s_P = s_all + RPB * 2 * T;
  float* s_a = s_P + RPB * 2 * T;
  const int n0 = blockIdx.y * RPB, b = n0 / beam;
  const int start = plen > 1 ? plen : 1;
  for (int i = threadIdx.x; i < RPB * T; i += 256) {
    const int r = i / T, t = i - r * T;
    const float r0 = r_prev[((long long)t * 2 + 0) * N + n0 + r], r1 = r_prev[((long long)t * 2 + 1) * N + n0 + r];
    s_phi[(r * 2 + 0) * T + t] = logaddexpf(r0, r1);
    s_phi[(r * 2 + 1) * T + t] = r1;
  }
  __syncthreads();
  {  // a = max over the frames t - 1 = start - 1 .. T - 2 of each of the 2 RPB rows: one wave per row, rows dealt round-robin
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int row = w; row < 2 * RPB; row += 4) {
      float mx = -INFINITY;
      for (int t = start - 1 + lane; t < T - 1; t += 64) mx = fmaxf(mx, s_phi[row * T + t]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if (lane == 0) s_a[row] = (mx == -INFINITY) ? 0.f : mx;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * RPB * T; i += 256) s_P[i] = __expf(s_phi[i] - s_a[i / T]);
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= V) return;
  const int len = enc_lens[b];
  const float* xc = x + (long long)b * T * V + c;
  const float* Ec = E + (long long)b * T * V + c;
  const float bc = bmax[(long long)b * V + c];
  const float pad_x = (c == blank) ? 0.f : CTC_IMPOSSIBLE;
  const float e_pad = __expf(pad_x - bc);
  const float first = (plen == 0) ? ((0 < len) ? xc[0] : pad_x) : CTC_IMPOSSIBLE;
  int var[RPB];
  float S[RPB];
  unsigned exact = 0;  // rows of this candidate that take the exact form
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const long long lastc = plen > 0 ? last[n0 + r] : 0;
    var[r] = r * 2 + (c == lastc ? 1 : 0);
    const float f0 = first - s_a[var[r]] - bc;  // (the empty prefix's first term is not scaled by phi: it may tower over a + bmax[c])
    if (f0 > 80.f) exact |= 1u << r;
    S[r] = f0 > 80.f ? 0.f : __expf(f0);
  }
  for (int t0 = start; t0 < T; t0 += 8) {
    float el[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) el[u] = (t0 + u < len) ? Ec[(long long)(t0 + u) * V] : e_pad;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + u < T) {
#pragma unroll
        for (int r = 0; r < RPB; ++r) S[r] = fmaf(s_P[var[r] * T + t0 + u - 1], el[u], S[r]);
      }
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    float out;
    if (S[r] > 1e-30f && !((exact >> r) & 1u)) {
      out = s_a[var[r]] + bc + __logf(S[r]);
    } else {  // underflow (or an impossible row): the exact form for this (row, candidate)
      const float* phi = s_phi + var[r] * T;
      float m = first;
      for (int t = start; t < T; ++t) m = fmaxf(m, phi[t - 1] + ((t < len) ? xc[(long long)t * V] : pad_x));
      float sum = __expf(first - m);
      for (int t = start; t < T; ++t) sum += __expf(phi[t - 1] + ((t < len) ? xc[(long long)t * V] : pad_x) - m);
      out = m + __logf(sum);
    }
    if (c == eos) out = s_phi[(r * 2 + 0) * T + eos_frame[n0 + r]];
    if (c == blank) out = CTC_IMPOSSIBLE;
    psi[(long long)(n0 + r) * V + c] = out;
  }
}

// The factorised scores of one sentence's 16 rows as an exact-f32 matrix product on the matrix cores: S (16 x V) = P (16 x T) E_b (T x V),
// v_mfma_f32_16x16x4_f32 (a chain of f32 multiply-adds: the vector kernel's arithmetic, summed in another order).  grid (ceil(V / 256), B ceil(beam / 16)),
// 4 waves of 64 columns each.  A lane reads 16 bytes of a frame's row of E -- 4 columns, one for each of its wave's 4 accumulator tiles, so
// tile j holds columns c0 + 4 (lane & 15) + j and a row of the result leaves as 16-byte stores -- and one value of P^T (T x 16, LDS) per 4
// frames.  The variant of P for a row's own last token (r_blank instead of r_sum) differs in ONE column per row: a wave whose 64 columns hold
// such a token runs a second set of accumulators on the same loads.  Frames past the sentence's length have E = 0 for every token but the
// blank, whose column is overwritten: the sweep stops at the length.  phi is kept only as P: the rare exact form re-reads r_prev.
template <bool PREP>
__global__ __launch_bounds__(256) void ctc_psi_mfma_kernel(const float* x, const float* E, const float* bmax, const int* enc_lens,
                                                           const int* eos_frame, const float* r_prev, const long long* last, float* psi, int N,
                                                           int T, int V, int beam, int plen, int blank, int eos, const float* pt,
                                                           const float* pa) {
  extern __shared__ __attribute__((aligned(16))) float s_all[];  // P^T [2][T][16] (r_sum, r_blank variants), a[2][16], partial maxima [8][32]
  float* s_P0 = s_all;
  float* s_P1 = s_all + 16 * T;
  float* s_a = s_all + 32 * T;
  float* s_red = s_a + 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // group blockIdx.y = rows 16 gi .. of sentence b (G = ceil(beam / 16) groups per sentence; the last one may hold fewer than 16 rows:
  // its idle rows repeat row nrows - 1 in every load and are never stored)
  const int G = (beam + 15) >> 4, b = blockIdx.y / G, gi = blockIdx.y - b * G;
  const int n0 = b * beam + gi * 16, nrows = min(16, beam - gi * 16);
  const int start = plen > 1 ? plen : 1;
  // the first frames of E are on their way while the workgroup prepares P (they depend on nothing in LDS)
  const int c0 = blockIdx.x * 256 + w * 64;
  const int rr = lane & 15, kq = lane >> 4;
  const int cl = c0 + 4 * rr;  // this lane's 4 columns (V % 4 == 0: all inside or all outside)
  const bool col_ok = cl < V;
  const int len = enc_lens[b];
  const int tend = min(T, len);
  // Every load is unconditional (the compiler then counts them: waits of the form vmcnt(n), not vmcnt(0)): frames past the end re-read the
  // last frame and meet a zero of P, columns past the vocabulary re-read its last four and are never stored.
  const float* Ec = E + (long long)b * T * V + min(cl, V - 4);
  constexpr int UL = 4, D = 4;  // batches of UL k-steps (4 frames each); D batches in flight per wave (16 KB)
  auto fetch = [&](int t0, float4 (&e)[UL]) {
#pragma unroll
    for (int u = 0; u < UL; ++u) e[u] = *(const float4*)(Ec + (long long)max(min(t0 + 4 * u + kq, tend - 1), 0) * V);
  };
  // phi of the 16 rows: 8 pairs of r_prev per thread and pass.  The first pass's loads go out BEFORE the first D batches of E, so that its
  // arithmetic waits for them only (loads retire in order) while E travels.
  float4 ring[D][UL];
  auto phi_load = [&](int i0, float (&r0)[8], float (&r1)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = min(i0 + 256 * q, 16 * T - 1);
      r0[q] = r_prev[((long long)(i >> 4) * 2 + 0) * N + n0 + min(i & 15, nrows - 1)];
      r1[q] = r_prev[((long long)(i >> 4) * 2 + 1) * N + n0 + min(i & 15, nrows - 1)];
    }
  };
  auto phi_store = [&](int i0, const float (&r0)[8], const float (&r1)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + 256 * q;
      if (i < 16 * T) {
        s_P0[i] = logaddexpf(r0[q], r1[q]);
        s_P1[i] = r1[q];
      }
    }
  };
  if constexpr (PREP) {  // P^T and a of this group of 16 rows, made by ctc_state_kernel with the rows themselves: a copy instead of the block below
    const f32x4* src = (const f32x4*)(pt + (long long)blockIdx.y * 32 * T);
    f32x4* dst = (f32x4*)s_all;
    f32x4 pv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pv[q] = src[min(tid + 256 * q, 8 * T - 1)];
    asm volatile("" ::: "memory");  // (keeps the two groups of loads in this order)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) fetch(start + dd * 4 * UL, ring[dd]);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[min(tid + 256 * q, 8 * T - 1)] = pv[q];  // (past the end: the last element once more)
    for (int i = tid + 256 * 8; i < 8 * T; i += 256) dst[i] = src[i];
    if (tid < 32) s_a[tid] = pa[(long long)blockIdx.y * 32 + tid];
  } else {
  {
    float r0[8], r1[8];
    phi_load(tid, r0, r1);
    asm volatile("" ::: "memory");  // (keeps the two groups of loads in this order)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) fetch(start + dd * 4 * UL, ring[dd]);
    asm volatile("" ::: "memory");
    phi_store(tid, r0, r1);
  }
  for (int i0 = tid + 256 * 8; i0 < 16 * T; i0 += 256 * 8) {
    float r0[8], r1[8];
    phi_load(i0, r0, r1);
    phi_store(i0, r0, r1);
  }
  __syncthreads();
  {  // a = max over the frames t - 1 = start - 1 .. T - 2 of each of the 32 (variant, row) columns
    const int col = tid & 31, part = tid >> 5;
    const float* src = (col < 16 ? s_P0 : s_P1) + (col & 15);
    float mx = -INFINITY;
    for (int t = start - 1 + part; t < T - 1; t += 8) mx = fmaxf(mx, src[t * 16]);
    s_red[part * 32 + col] = mx;
  }
  __syncthreads();
  if (tid < 32) {
    float mx = s_red[tid];
#pragma unroll
    for (int q = 1; q < 8; ++q) mx = fmaxf(mx, s_red[q * 32 + tid]);
    s_a[tid] = (mx == -INFINITY) ? 0.f : mx;
  }
  __syncthreads();
  for (int i = tid; i < 32 * T; i += 256) s_all[i] = __expf(s_all[i] - s_a[(i >= 16 * T ? 16 : 0) + (i & 15)]);
  }
  __syncthreads();
  if (c0 >= V) return;
  const long long lc_lane = plen > 0 ? last[n0 + min(rr, nrows - 1)] : 0;
  const bool dual = __any(lc_lane >= c0 && lc_lane < c0 + 64);
  f32x4 acc0[4], acc1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc0[j] = acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto consume = [&](const float4 (&eb)[UL], int tb) {
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int t = tb + 4 * u + kq;
      const int ta = min(t, tend) - 1;
      const float a0 = t < tend ? s_P0[ta * 16 + rr] : 0.f;
      const float4 e = eb[u];
      acc0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, e.x, acc0[0], 0, 0, 0);
      acc0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, e.y, acc0[1], 0, 0, 0);
      acc0[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, e.z, acc0[2], 0, 0, 0);
      acc0[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, e.w, acc0[3], 0, 0, 0);
      if (dual) {
        const float a1 = t < tend ? s_P1[ta * 16 + rr] : 0.f;
        acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, e.x, acc1[0], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, e.y, acc1[1], 0, 0, 0);
        acc1[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, e.z, acc1[2], 0, 0, 0);
        acc1[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, e.w, acc1[3], 0, 0, 0);
      }
    }
  };
  // Rounds of D batches.  While another round follows, every slot is refilled right after it is consumed, with no condition around the
  // loads (conditional loads make the compiler wait for vmcnt(0)); the last round only consumes.
  constexpr int R = 4 * UL * D;
  int t0 = start;
  {
    for (; t0 + R < tend; t0 += R) {
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        consume(ring[dd], t0 + dd * 4 * UL);
        fetch(t0 + R + dd * 4 * UL, ring[dd]);
      }
    }
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
      if (t0 + dd * 4 * UL < tend) consume(ring[dd], t0 + dd * 4 * UL);  // (uniform)
  }
  if (!col_ok) return;
  // ---- epilogue: lane = columns cl .. cl + 3 (tile j = column cl + j) of rows 4 kq + v ----
  const float4 bc4 = *(const float4*)(bmax + (long long)b * V + cl);
  const float bc[4] = {bc4.x, bc4.y, bc4.z, bc4.w};
  const float* xc = x + (long long)b * T * V + cl;
  float first[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float pad_x = (cl + j == blank) ? 0.f : CTC_IMPOSSIBLE;
    first[j] = (plen == 0) ? ((0 < len) ? xc[j] : pad_x) : CTC_IMPOSSIBLE;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 4 * kq + v, n = n0 + row;
    if (row >= nrows) break;
    const long long lastc = plen > 0 ? last[n] : 0;  // (the reference takes token 0 for the empty prefix, ctc_decoding.py:84)
    const float a0r = s_a[row], a1r = s_a[16 + row];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cl + j;
      const bool sel = (c == lastc);
      const float a = sel ? a1r : a0r;
      float S = sel ? acc1[j][v] : acc0[j][v];
      const float f0 = first[j] - a - bc[j];  // (the empty prefix's first term is not scaled by phi: it may tower over a + bmax[c])
      const bool exact = f0 > 80.f;
      S += exact ? 0.f : __expf(f0);
      float out;
      if (S > 1e-30f && !exact) {
        out = a + bc[j] + __logf(S);
      } else {  // underflow (or an impossible row): the exact two-sweep form for this (row, candidate)
        const float pad_x = (c == blank) ? 0.f : CTC_IMPOSSIBLE;
        const float* rp = r_prev + n;
        float m = first[j];
        for (int t = start; t < T; ++t) {
          const float r0 = rp[((long long)(t - 1) * 2 + 0) * N], r1 = rp[((long long)(t - 1) * 2 + 1) * N];
          m = fmaxf(m, (sel ? r1 : logaddexpf(r0, r1)) + ((t < len) ? xc[(long long)t * V + j] : pad_x));
        }
        float sum = __expf(first[j] - m);
        for (int t = start; t < T; ++t) {
          const float r0 = rp[((long long)(t - 1) * 2 + 0) * N], r1 = rp[((long long)(t - 1) * 2 + 1) * N];
          sum += __expf((sel ? r1 : logaddexpf(r0, r1)) + ((t < len) ? xc[(long long)t * V + j] : pad_x) - m);
        }
        out = m + __logf(sum);
      }
      if (c == eos) {
        const long long ef = eos_frame[n];
        out = logaddexpf(r_prev[(ef * 2 + 0) * N + n], r_prev[(ef * 2 + 1) * N + n]);
      }
      if (c == blank) out = CTC_IMPOSSIBLE;
      o[j] = out;
    }
    *(float4*)(psi + (long long)n * V + cl) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// An affine map r -> logaddexp(a + r, b) in the log domain (R -> A R + B in the probability domain), and "g after f".
struct LogAffine {
  float a, b;
};
// logaddexp on the hardware exponential and logarithm (v_exp_f32 / v_log_f32: ~1 ulp each).  log1pf's polynomial (~150 instructions with
// two divisions) buys RELATIVE accuracy of log(1 + e) for tiny e; the sum m + log(1 + e) is rounded to ulp(m) anyway, so for forward
// variables (log-probabilities, |m| from ~1e-3 up to 1e20) the absolute error of this form, <= ~1.2e-7, is below the rounding of the exact
// one.  The scans below are chains of ~30 DEPENDENT logaddexps per row at one wave per SIMD: instruction count is their whole cost.
__device__ __forceinline__ float logaddexp_hw(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return m;
  return m + __logf(1.f + __expf(fminf(a, b) - m));
}
__device__ __forceinline__ LogAffine la_then(const LogAffine f, const LogAffine g) { return {f.a + g.a, logaddexp_hw(g.a + f.b, g.b)}; }

// Inclusive scan over frames t = start .. T - 1 of the maps (a[t], b[t]) applied to r0, one wave: out[t] = (f_t o ... o f_start)(r0).
// Lane l owns the CH consecutive frames start + l CH ...; a / b are read from LDS, the result is written over `out` (LDS).
__device__ __forceinline__ void la_wave_scan(const float* a, const float* b, float* out, float r0, int start, int T, int lane) {
  const int n = T - start;
  if (n <= 0) return;
  const int CH = (n + 63) >> 6;
  const int t0 = start + lane * CH, t1 = min(t0 + CH, T);
  LogAffine tot = {0.f, -INFINITY};  // identity
  for (int t = t0; t < t1; ++t) tot = la_then(tot, LogAffine{a[t], b[t]});
  // exclusive scan of the lanes' totals
  LogAffine inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    LogAffine up = {__shfl_up(inc.a, o, 64), __shfl_up(inc.b, o, 64)};
    if (lane >= o) inc = la_then(up, inc);
  }
  LogAffine pre = {__shfl_up(inc.a, 1, 64), __shfl_up(inc.b, 1, 64)};
  if (lane == 0) pre = {0.f, -INFINITY};
  float r = logaddexp_hw(pre.a + r0, pre.b);  // the value just before this lane's first frame
  for (int t = t0; t < t1; ++t) {
    r = logaddexp_hw(a[t] + r, b[t]);
    out[t] = r;
  }
}

// One wave per new row n (2 or 4 per workgroup): the forward variables of (parent row beam_idx[n]) + (token tokens[n]) over all frames.
// The two-line recursion (:132-142) is a pair of first-order linear recurrences in the probability domain,
//   R^n_t = x_t (R^n_{t-1} + Phi_{t-1}),   R^b_t = xb_t (R^b_{t-1} + R^n_{t-1}),
// so each is a prefix scan of affine maps: the lanes gather phi / x of the whole utterance into LDS (independent loads), then two
// wave scans replace the T sequential steps (measured: 282 us per launch with one lane walking the frames).
__global__ __launch_bounds__(256) void ctc_state_kernel(const float* x, const int* enc_lens, const float* r_prev, const long long* last,
                                                       const long long* beam_idx, const long long* tokens, float* r_new, int N, int T, int V,
                                                       int beam, int plen, int blank, const float* psi_prev, float* psi_sel, float* pt,
                                                       float* pa) {
  extern __shared__ float s_st[];  // [4 waves][4][T]: a (x label), b (x label + phi), then r^n; x blank
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * (blockDim.x >> 6) + w;
  if (n >= N) return;
  float* sa = s_st + (w * 4 + 0) * T;
  float* sb = s_st + (w * 4 + 1) * T;
  float* rn = s_st + (w * 4 + 2) * T;
  float* xb = s_st + (w * 4 + 3) * T;
  const int par = (int)beam_idx[n], b = par / beam;
  const long long c = tokens[n];
  if (psi_sel && lane == 0) psi_sel[n] = psi_prev[(long long)par * V + c];  // the prefix score this row was chosen with (ctc_decoding.py:63)
  const int len = enc_lens[b];
  const long long lastc = plen > 0 ? last[par] : 0;
  const float* xc = x + (long long)b * T * V;
  const int start = plen > 1 ? plen : 1;
  for (int t0 = lane; t0 < T; t0 += 64 * 4) {  // (four frames per lane and pass: their 16 gathers -- one cache line per lane each -- fly together)
    float xl[4], xk[4], p0[4], p1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(t0 + 64 * u, T - 1), tp = max(t - 1, 0);
      xl[u] = xc[(long long)min(t, max(len - 1, 0)) * V + c];
      xk[u] = xc[(long long)min(t, max(len - 1, 0)) * V + blank];
      p0[u] = r_prev[((long long)tp * 2 + 0) * N + par];
      p1[u] = r_prev[((long long)tp * 2 + 1) * N + par];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + 64 * u;
      if (t < T) {
        const float xv = (t < len) ? xl[u] : (c == blank ? 0.f : CTC_IMPOSSIBLE);
        float phi = CTC_IMPOSSIBLE;
        if (t >= 1) phi = (c == lastc) ? p1[u] : logaddexpf(p0[u], p1[u]);
        sa[t] = xv;
        sb[t] = xv + phi;  // b_t = x_t + phi_{t-1}
        xb[t] = (t < len) ? xk[u] : 0.f;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its own LDS writes, then its own reads)
  const float rn0 = (plen == 0) ? sa[0] : CTC_IMPOSSIBLE;  // r^n at frame start - 1 (:116-117); r^b there is impossible
  for (int t = lane; t < start && t < T; t += 64) rn[t] = (t == 0 && plen == 0) ? rn0 : CTC_IMPOSSIBLE;
  la_wave_scan(sa, sb, rn, rn0, start, T, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // second recurrence: a_t = xb_t, b_t = xb_t + r^n_{t-1}
  for (int t = start + lane; t < T; t += 64) sb[t] = xb[t] + rn[t - 1];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  la_wave_scan(xb, sb, sa, CTC_IMPOSSIBLE, start, T, lane);  // r^b over sa
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int t = lane; t < T; t += 64) {
    r_new[((long long)t * 2 + 0) * N + n] = rn[t];
    r_new[((long long)t * 2 + 1) * N + n] = t < start ? CTC_IMPOSSIBLE : sa[t];
  }
  if (pt) {
    // What ctc_psi_mfma_kernel derives from this row at the NEXT position (prefix length plen + 1), made here once instead of in every
    // one of its workgroups: phi (r_sum, r_blank), a = its maximum over the frames the sums read, P = exp(phi - a) in the kernel's P^T layout.
    const int sb_ = n / beam, j = n - sb_ * beam;  // (groups of 16 rows never straddle two sentences)
    const int g = sb_ * ((beam + 15) >> 4) + (j >> 4), r = j & 15;
    float m0 = -INFINITY, m1 = -INFINITY;
    for (int t = lane; t < T; t += 64) {  // (every lane reads back its own frames only)
      const float rb = t < start ? CTC_IMPOSSIBLE : sa[t];
      const float p0 = logaddexpf(rn[t], rb);
      sb[t] = p0;
      xb[t] = rb;
      if (t >= plen && t < T - 1) {
        m0 = fmaxf(m0, p0);
        m1 = fmaxf(m1, rb);
      }
    }
    m0 = sc_wave_max(m0);
    m1 = sc_wave_max(m1);
    const float a0 = (m0 == -INFINITY) ? 0.f : m0, a1 = (m1 == -INFINITY) ? 0.f : m1;
    float* p0 = pt + ((long long)(g * 2 + 0) * T) * 16 + r;
    float* p1 = pt + ((long long)(g * 2 + 1) * T) * 16 + r;
    for (int t = lane; t < T; t += 64) {
      p0[(long long)t * 16] = __expf(sb[t] - a0);
      p1[(long long)t * 16] = __expf(xb[t] - a1);
    }
    if (lane == 0) {
      pa[g * 32 + r] = a0;
      pa[g * 32 + 16 + r] = a1;
    }
  }
}

}  // namespace

// psi (N, V) of every candidate of every hypothesis row.  x (B, T, V) f32 log-probabilities, one copy per sentence (row n belongs to
// sentence n / beam); enc_lens (B); eos_frame (N): the frame whose r_sum is the <eos> score of a row; r_prev (T, 2, N); last (N): the last
// label of each prefix (ignored when plen == 0); plen: the prefix length without <sos>.
SC_API int sc_ctc_prefix_scores(const float* x, const int* enc_lens, const int* eos_frame, const float* r_prev, const long long* last,
                                float* psi, int B, int beam, int T, int V, int plen, int blank, int eos, hipStream_t stream) {
  SC_CHECK_ARG(x && enc_lens && eos_frame && r_prev && psi && (last || plen == 0), "sc_ctc_prefix_scores: null pointer");
  SC_CHECK_ARG(B >= 1 && beam >= 1 && T >= 1 && V >= 1 && plen >= 0 && blank >= 0 && blank < V && eos >= 0 && eos < V &&
                   (long long)B * beam <= 65535, "sc_ctc_prefix_scores: bad sizes B=%d beam=%d T=%d V=%d plen=%d", B, beam, T, V, plen);
  const int N = B * beam;
  int rpb = beam % 8 == 0 ? 8 : beam % 4 == 0 ? 4 : beam % 2 == 0 ? 2 : 1;  // rows of one sentence per workgroup
  while (rpb > 1 && (size_t)rpb * 2 * T * sizeof(float) > 64 * 1024) rpb >>= 1;  // (long utterances: fewer rows share a workgroup)
  const size_t smem = (size_t)rpb * 2 * T * sizeof(float);
  SC_CHECK_ARG(smem <= 64 * 1024, "sc_ctc_prefix_scores: T too large for the LDS rows (%d frames x %d rows)", T, rpb);
  const dim3 grid(sc_cdiv(V, 256), N / rpb);
#define SC_LAUNCH_PSI(R) \
  hipLaunchKernelGGL(ctc_psi_kernel<R>, grid, dim3(256), smem, stream, x, enc_lens, eos_frame, r_prev, last, psi, N, T, V, beam, plen, blank, eos)
  if (rpb == 8) SC_LAUNCH_PSI(8);
  else if (rpb == 4) SC_LAUNCH_PSI(4);
  else if (rpb == 2) SC_LAUNCH_PSI(2);
  else SC_LAUNCH_PSI(1);
#undef SC_LAUNCH_PSI
  SC_LAUNCH_CHECK("ctc_psi_kernel");
  return SC_OK;
}

// The same scores from the factorised exponential: E (B, T, V) = exp(x - bmax) and bmax (B, V) = max over the frames of x, made once per
// batch by the caller (two torch launches).  Results equal sc_ctc_prefix_scores' up to the f32 rounding of a 250-term sum.
SC_API int sc_ctc_prefix_scores_fast(const float* x, const float* E, const float* bmax, const int* enc_lens, const int* eos_frame,
                                     const float* r_prev, const long long* last, float* psi, int B, int beam, int T, int V, int plen, int blank,
                                     int eos, const float* pt, const float* pa, hipStream_t stream) {
  SC_CHECK_ARG(x && E && bmax && enc_lens && eos_frame && r_prev && psi && (last || plen == 0), "sc_ctc_prefix_scores_fast: null pointer");
  SC_CHECK_ARG((pt == nullptr) == (pa == nullptr) && (!pt || (plen >= 1 && (uintptr_t)pt % 16 == 0)),
               "sc_ctc_prefix_scores_fast: pt / pa come together, from sc_ctc_prefix_state at prefix length plen - 1");
  SC_CHECK_ARG(B >= 1 && beam >= 1 && T >= 2 && V >= 1 && plen >= 0 && blank >= 0 && blank < V && eos >= 0 && eos < V &&
                   (long long)B * beam <= 65535, "sc_ctc_prefix_scores_fast: bad sizes B=%d beam=%d T=%d V=%d plen=%d", B, beam, T, V, plen);
  const int N = B * beam;
  // The matrix-core form, groups of up to 16 rows of a sentence (SC_CTC_PSI_MFMA=0: the vector kernel below).
  static const bool use_mfma = [] { const char* e = getenv("SC_CTC_PSI_MFMA"); return !(e && e[0] == '0'); }();
  const size_t smem_m = ((size_t)32 * T + 32 + 256) * sizeof(float);
  const int groups = B * sc_cdiv(beam, 16);
  if (use_mfma && V % 4 == 0 && smem_m <= 64 * 1024 && groups <= 65535 &&
      (((uintptr_t)E | (uintptr_t)bmax | (uintptr_t)psi) % 16) == 0) {
    if (pt)
      hipLaunchKernelGGL(ctc_psi_mfma_kernel<true>, dim3(sc_cdiv(V, 256), groups), dim3(256), smem_m, stream, x, E, bmax, enc_lens, eos_frame,
                         r_prev, last, psi, N, T, V, beam, plen, blank, eos, pt, pa);
    else
      hipLaunchKernelGGL(ctc_psi_mfma_kernel<false>, dim3(sc_cdiv(V, 256), groups), dim3(256), smem_m, stream, x, E, bmax, enc_lens, eos_frame,
                         r_prev, last, psi, N, T, V, beam, plen, blank, eos, pt, pa);
    SC_LAUNCH_CHECK("ctc_psi_mfma_kernel");
    return SC_OK;
  }
  // (16 rows per workgroup when the beam allows: the (B, T, V) factor is then read once per sentence and step)
  int rpb = beam % 16 == 0 ? 16 : beam % 8 == 0 ? 8 : beam % 4 == 0 ? 4 : beam % 2 == 0 ? 2 : 1;
  while (rpb > 1 && (size_t)rpb * (4 * T + 2) * sizeof(float) > 64 * 1024) rpb >>= 1;
  const size_t smem = (size_t)rpb * (4 * T + 2) * sizeof(float);
  SC_CHECK_ARG(smem <= 64 * 1024, "sc_ctc_prefix_scores_fast: T too large for the LDS rows (%d frames x %d rows)", T, rpb);
  const dim3 grid(sc_cdiv(V, 256), N / rpb);
#define SC_LAUNCH_PSI(R)                                                                                                              \
  hipLaunchKernelGGL(ctc_psi_fast_kernel<R>, grid, dim3(256), smem, stream, x, E, bmax, enc_lens, eos_frame, r_prev, last, psi, N, T, V, beam, \
                     plen, blank, eos)
  if (rpb == 16) SC_LAUNCH_PSI(16);
  else if (rpb == 8) SC_LAUNCH_PSI(8);
  else if (rpb == 4) SC_LAUNCH_PSI(4);
  else if (rpb == 2) SC_LAUNCH_PSI(2);
  else SC_LAUNCH_PSI(1);
#undef SC_LAUNCH_PSI
  SC_LAUNCH_CHECK("ctc_psi_fast_kernel");
  return SC_OK;
}

// r_new (T, 2, N): the forward variables of the surviving candidates, new row n = prefix of row beam_idx[n] extended by tokens[n]
// (what ctc_decoding.py:162-196 gathers out of its (T, 2, N, V) tensor).
SC_API int sc_ctc_prefix_state(const float* x, const int* enc_lens, const float* r_prev, const long long* last, const long long* beam_idx,
                               const long long* tokens, float* r_new, int B, int beam, int T, int V, int plen, int blank, const float* psi_prev,
                               float* psi_sel, float* pt, float* pa, hipStream_t stream) {
  SC_CHECK_ARG(x && enc_lens && r_prev && beam_idx && tokens && r_new && (last || plen == 0), "sc_ctc_prefix_state: null pointer");
  SC_CHECK_ARG((psi_prev == nullptr) == (psi_sel == nullptr), "sc_ctc_prefix_state: psi_prev and psi_sel come together");
  SC_CHECK_ARG(r_prev != r_new, "sc_ctc_prefix_state: r_prev and r_new must be two buffers");
  SC_CHECK_ARG((pt == nullptr) == (pa == nullptr), "sc_ctc_prefix_state: pt / pa come together");
  SC_CHECK_ARG(B >= 1 && beam >= 1 && T >= 1 && V >= 1 && plen >= 0 && blank >= 0 && blank < V, "sc_ctc_prefix_state: bad sizes");
  // (the rows' gathers -- one cache line per lane -- queue in their CU's address unit: two rows per workgroup spread 512 rows over all CUs)
  const int wpb = (B * beam <= 1024) ? 2 : 4;
  const size_t smem = (size_t)wpb * 4 * T * sizeof(float);
  SC_CHECK_ARG(smem <= 64 * 1024, "sc_ctc_prefix_state: T too large for the LDS rows (%d frames)", T);
  hipLaunchKernelGGL(ctc_state_kernel, dim3(sc_cdiv(B * beam, wpb)), dim3(64 * wpb), smem, stream, x, enc_lens, r_prev, last, beam_idx, tokens, r_new,
                     B * beam, T, V, beam, plen, blank, psi_prev, psi_sel, pt, pa);
  SC_LAUNCH_CHECK("ctc_state_kernel");
  return SC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-wise top-K (K <= 64) of a (R, L) f32 matrix, sorted descending, ties by the lower index: the candidate selection of a beam-search
// step (speechain/infer_func/beam_search.py:419-421 calls torch.topk on the (batch, beam x vocabulary) score matrix; with 80 000
// columns that is a chain of multi-block radix kernels with host-side bookkeeping -- 5 launches, and not reliably replayable from a
// hipGraph).  One workgroup per row: three histogram rounds over the order-preserving integer image of the floats (11 + 11 + 10 bits, LDS
// atomics, one per RUN of equal bins a lane meets: scores share an exponent and would otherwise hit one address) give the K-th largest key
// exactly; the row is then walked once more for what lies above the threshold and the elements equal to it (lowest indices first when
// there are more of those than places left); K threads rank the K survivors against each other.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ unsigned topk_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float <=> larger key (-0 < +0; NaN sorts above +inf)
}

// finds, scanning bins from the top, the bin in which the running count reaches `need`; returns it and the count above it
// The bin in which the running count from the TOP reaches `need`, by all 1024 threads: thread t owns bins nbins - 1 - t per - i (per = 1
// or 2); wave scans, the waves' totals through wtot[16], one barrier inside.  (One wave walking 32 bins per lane read the histogram with a
// 64-way bank conflict: ~1.2 us per round, three rounds per selection.)
__device__ __forceinline__ void topk_find_bin(const unsigned* hist, int nbins, unsigned need, int* sh_bin, unsigned* sh_above, unsigned* wtot) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = nbins >> 10;  // (nbins is 2048 or 1024, the workgroup 1024 threads)
  const int b0 = nbins - 1 - tid * per;
  const unsigned h0 = hist[b0], h1 = per == 2 ? hist[b0 - 1] : 0u;
  const unsigned mine = h0 + h1;
  unsigned inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  unsigned base = 0;
  for (int i = 0; i < w; ++i) base += wtot[i];
  inc += base;
  const unsigned before = inc - mine;  // count in the bins above this thread's
  if (before < need && inc >= need) {  // the crossing is in this thread's bins
    if (before + h0 >= need) {
      *sh_bin = b0;
      *sh_above = before;
    } else {
      *sh_bin = b0 - 1;
      *sh_above = before + h0;
    }
  }
}

// (`row` may point into LDS -- the fused candidate kernel below stages its row there -- or into global memory: flat loads)
__device__ __forceinline__ void topk_rows_body(const float* row, int L, int K, float* vals, long long* idx) {
  __shared__ unsigned hist[2048];
  __shared__ int sh_bin;
  __shared__ unsigned sh_above, sh_cnt, sh_eq_total, sh_eq_base[1025];
  __shared__ unsigned sel_key[64];
  __shared__ int sel_idx[64];
  const int tid = threadIdx.x;
  unsigned prefix = 0, need = (unsigned)K;  // key bits fixed so far (from the top), elements still wanted among those matching the prefix
  // ---- three rounds: bits [31:21], [20:10], [9:0].  Scores cluster (same exponent): a lane adds a RUN of equal bins with one atomic ----
  const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
  unsigned mask_hi = 0;  // which top bits are fixed
  for (int rd = 0; rd < 3; ++rd) {
    const int nb = 1 << widths[rd];
    for (int b = tid; b < nb; b += 1024) hist[b] = 0;
    __syncthreads();
    int cur = -1;
    unsigned run = 0;
    for (int i = tid; i < L; i += 8 * 1024) {  // eight loads in flight per lane (the row comes out of L2: latency, not bytes, is the cost)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i + u * 1024 < L) ? row[i + u * 1024] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned k = topk_key(v[u]);
        if (i + u * 1024 < L && (k & mask_hi) == prefix) {
          const int b = (int)((k >> shifts[rd]) & (unsigned)(nb - 1));
          if (b == cur) {
            ++run;
          } else {
            if (run) atomicAdd(&hist[cur], run);
            cur = b;
            run = 1;
          }
        }
      }
    }
    {  // scores of one row share their exponent: in the first round a whole wave usually holds ONE bin -- one atomic for the wave then
      const int c0 = __builtin_amdgcn_readfirstlane(cur);
      if (__all(run == 0 || cur == c0)) {
        const unsigned tot = (unsigned)sc_wave_sum((float)run);  // (< 2^24: exact)
        if ((tid & 63) == 0 && tot) atomicAdd(&hist[c0], tot);
      } else if (run) {
        atomicAdd(&hist[cur], run);
      }
    }
    __syncthreads();
    topk_find_bin(hist, nb, need, &sh_bin, &sh_above, sel_key);  // (sel_key is free until the collection pass)
    __syncthreads();
    prefix |= (unsigned)sh_bin << shifts[rd];
    mask_hi |= (unsigned)(nb - 1) << shifts[rd];
    need -= sh_above;  // (no barrier here: sh_bin / sh_above are next written two barriers on, the histogram is read by nobody any more)
  }
  const unsigned thr = prefix;  // the K-th largest key; `need` (>= 1) elements equal to it are wanted, lowest indices first
  if (tid == 0) {
    sh_cnt = 0;
    sh_eq_total = 0;
  }
  __syncthreads();
  // ---- everything above the threshold; count of the elements equal to it ----
  {
    unsigned eq = 0;
    for (int i = tid; i < L; i += 8 * 1024) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i + u * 1024 < L) ? row[i + u * 1024] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i + u * 1024 >= L) continue;
        const unsigned k = topk_key(v[u]);
        if (k == thr) {  // (an equal element is kept for now: at most 64 slots are ever used; the tie path below starts over)
          ++eq;
          const unsigned slot = atomicAdd(&sh_cnt, 1u);
          if (slot < 64) {
            sel_key[slot] = k;
            sel_idx[slot] = i + u * 1024;
          }
        } else if (k > thr) {
          const unsigned slot = atomicAdd(&sh_cnt, 1u);
          if (slot < 64) {
            sel_key[slot] = k;
            sel_idx[slot] = i + u * 1024;
          }
        }
      }
    }
    if (eq) atomicAdd(&sh_eq_total, eq);
  }
  __syncthreads();
  if (sh_eq_total != need) {  // a tie across the last place (rare): start the collection over
    __syncthreads();
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    for (int i = tid; i < L; i += 1024) {
      const unsigned k = topk_key(row[i]);
      if (k > thr) {
        const unsigned slot = atomicAdd(&sh_cnt, 1u);
        sel_key[slot] = k;  // (fewer than K of them)
        sel_idx[slot] = i;
      }
    }
    __syncthreads();
  }
  if (sh_eq_total == need) {  // the usual case: no tie across the last place -- everything was collected in the pass above
  } else {  // a tie across the last place: the lowest indices win -- threads take contiguous index ranges and rank their equals in order
    const int chunk = (L + 1023) / 1024, i0 = tid * chunk, i1 = min(i0 + chunk, L);
    unsigned eq = 0;
    for (int i = i0; i < i1; ++i) eq += topk_key(row[i]) == thr;
    sh_eq_base[tid + 1] = eq;
    if (tid == 0) sh_eq_base[0] = 0;
    __syncthreads();
    if (tid == 0)
      for (int t = 1; t <= 1024; ++t) sh_eq_base[t] += sh_eq_base[t - 1];
    __syncthreads();
    unsigned eq_rank = sh_eq_base[tid];
    for (int i = i0; i < i1 && eq_rank < need; ++i)
      if (topk_key(row[i]) == thr) {
        ++eq_rank;
        const unsigned slot = atomicAdd(&sh_cnt, 1u);
        sel_key[slot] = thr;
        sel_idx[slot] = i;
      }
  }
  __syncthreads();
  // ---- order the K survivors: descending key, ascending index ----
  if (tid < K) {
    const unsigned k = sel_key[tid];
    const int id = sel_idx[tid];
    int rank = 0;
    for (int j = 0; j < K; ++j) rank += (sel_key[j] > k) || (sel_key[j] == k && sel_idx[j] < id);
    vals[rank] = row[id];
    idx[rank] = id;
  }
}

__global__ __launch_bounds__(1024) void topk_rows_kernel(const float* x, long long ld, int L, int K, float* vals, long long* idx) {
  topk_rows_body(x + (long long)blockIdx.x * ld, L, K, vals + (long long)blockIdx.x * K, idx + (long long)blockIdx.x * K);
}

// Second stage of a two-stage top-K over G groups of `parts` rows each (see sc_topk_merge): candidate j of part p sits at column p K + j.
// One workgroup per group, one thread per candidate; a candidate's rank is the number of candidates that beat it (higher key, or the same
// key at a lower column: the parts are in index order and each part's candidates are sorted with ties by the lower index, so column order
// IS index order among equal keys).
// Every part is sorted, so a candidate's rank is its position in its own part plus, for every other part, the length of that part's prefix
// that beats it -- a binary search per part (parts x log2 K LDS reads per thread instead of parts x K compares: 17 -> 5 us at 16 x 32).
__device__ __forceinline__ int merge_rank(const unsigned* key, int parts, int K, int p, int j) {
  const unsigned k = key[p * K + j];
  int rank = j;
  int top = 1;
  while (top * 2 <= K) top *= 2;  // largest power of two <= K
  // Per other part q: how many of its (descending) keys beat the candidate -- equal keys of an earlier part do.  Branch-free searches, four
  // parts abreast: the chains of dependent LDS reads of different parts overlap (15 parts x 6 reads one after the other took ~3 us).
  for (int q0 = 0; q0 < parts; q0 += 4) {
    int lo[4] = {0, 0, 0, 0};
    for (int s = top; s > 0; s >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(q0 + u, parts - 1);
        const int m = lo[u] + s;
        const unsigned kv = key[q * K + min(m, K) - 1];
        const bool beats = q < p ? kv >= k : kv > k;
        lo[u] = (m <= K && beats) ? m : lo[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) rank += (q0 + u < parts && q0 + u != p) ? lo[u] : 0;
  }
  return rank;
}

__global__ __launch_bounds__(1024) void topk_merge_kernel(const float* vals1, const long long* idx1, int parts, int K, int L, float* vals,
                                                          long long* idx) {
  __shared__ unsigned key[1024];
  const int n = parts * K, tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * n;
  if (tid < n) key[tid] = topk_key(vals1[base + tid]);
  __syncthreads();
  if (tid >= n) return;
  const int rank = merge_rank(key, parts, K, tid / K, tid % K);
  if (rank < K) {
    vals[(long long)blockIdx.x * K + rank] = vals1[base + tid];
    idx[(long long)blockIdx.x * K + rank] = (long long)(tid / K) * L + idx1[base + tid];
  }
}

// topk_merge_kernel and beam_advance_kernel in one launch (one workgroup per sentence): the merged 2 * beam candidates go to vals / idx AND
// to the step's record, and the next rows are chosen from them (see beam_advance_kernel).
__global__ __launch_bounds__(1024) void topk_merge_advance_kernel(const float* vals1, const long long* idx1, int parts, int K, int L, float* vals,
                                                                  long long* idx, const float* eos, const float* other, double* pack, int width,
                                                                  int eos_token, float* scores, long long* tokens, long long* prev_tokens,
                                                                  long long* beam_idx) {
  __shared__ unsigned key[1024];
  __shared__ float sv[64];
  __shared__ long long si[64];
  __shared__ int sp[64], st[64];  // part (= parent beam) and token of the survivors, kept beside si so that nobody divides by L
  const int n = parts * K, tid = threadIdx.x, b = blockIdx.x, beam = parts;
  const long long base = (long long)b * n;
  float myv = 0.f;
  long long tok = 0;
  if (tid < n) {  // (value and index in one trip: the ranks below then touch LDS only)
    myv = vals1[base + tid];
    tok = idx1[base + tid];
    key[tid] = topk_key(myv);
  }
  if (tid < beam) prev_tokens[(long long)b * beam + tid] = tokens[(long long)b * beam + tid];
  __syncthreads();
  if (tid < n) {
    const int rank = merge_rank(key, parts, K, tid / K, tid % K);
    if (rank < K) {
      sv[rank] = myv;
      si[rank] = (long long)(tid / K) * L + tok;
      sp[rank] = tid / K;
      st[rank] = (int)tok;
    }
  }
  __syncthreads();
  double* row = pack + (long long)b * width;
  for (int i = tid; i < width; i += 1024) {
    double v;
    if (i < K) v = (double)sv[i];
    else if (i < 2 * K) v = (double)si[i - K];
    else if (i < 2 * K + beam) v = (double)eos[(long long)b * beam + i - 2 * K];
    else v = (double)other[(long long)b * beam + i - 2 * K - beam];
    row[i] = v;
  }
  if (tid < K) {
    vals[(long long)b * K + tid] = sv[tid];
    idx[(long long)b * K + tid] = si[tid];
  }
  if (tid < 64) {  // the next rows: the first `beam` survivors that are not <eos>, in order -- one wave, a ballot instead of a walk
    const int tok = tid < K ? st[tid] : eos_token;
    const bool keep = tok != eos_token;
    const unsigned long long m = __ballot(keep);
    const int at = __popcll(m & ((1ull << tid) - 1ull));
    if (keep && at < beam) {
      const long long r = (long long)b * beam + at;
      scores[r] = sv[tid];
      tokens[r] = tok;
      beam_idx[r] = (long long)b * beam + sp[tid];
    }
  }
}

}  // namespace

// The K largest of every GROUP of `parts` consecutive rows of length L (a sentence's beams: one row of parts * L columns) from the
// per-row results of sc_topk_rows on the (G * parts, L) matrix: vals1 / idx1 (G * parts, K) -> vals / idx (G, K), idx = row-in-group * L +
// column, descending, ties by the lower index -- exactly what sc_topk_rows returns on the (G, parts * L) view, with G * parts workgroups
// scanning L columns each instead of G scanning parts * L.  parts * K <= 1024.
SC_API int sc_topk_merge(const float* vals1, const long long* idx1, int G, int parts, int K, int L, float* vals, long long* idx,
                         hipStream_t stream) {
  SC_CHECK_ARG(vals1 && idx1 && vals && idx, "sc_topk_merge: null pointer");
  SC_CHECK_ARG(G >= 1 && G <= 65535 && parts >= 1 && K >= 1 && K <= 64 && parts * K <= 1024 && L >= K,
               "sc_topk_merge: bad sizes G=%d parts=%d K=%d L=%d (parts * K <= 1024)", G, parts, K, L);
  hipLaunchKernelGGL(topk_merge_kernel, dim3(G), dim3(1024), 0, stream, vals1, idx1, parts, K, L, vals, idx);
  SC_LAUNCH_CHECK("topk_merge_kernel");
  return SC_OK;
}

// sc_topk_merge over the B sentences' beams followed by sc_beam_advance, in one launch: vals1 / idx1 (B * beam, K) from sc_topk_rows on the
// (B * beam, V) candidate scores; vals / idx (B, K) receive the merged selection as well.  K >= 2 * beam, K <= 64, beam * K <= 1024.
SC_API int sc_beam_merge_advance(const float* vals1, const long long* idx1, int B, int beam, int K, int V, const float* eos, const float* other,
                                 int eos_token, float* vals, long long* idx, double* pack, float* scores, long long* tokens,
                                 long long* prev_tokens, long long* beam_idx, hipStream_t stream) {
  SC_CHECK_ARG(vals1 && idx1 && vals && idx && pack && scores && tokens && prev_tokens && beam_idx && ((eos == nullptr) == (other == nullptr)),
               "sc_beam_merge_advance: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && beam >= 1 && K >= 2 * beam && K <= 64 && beam * K <= 1024 && V >= K && eos_token >= 0 && eos_token < V,
               "sc_beam_merge_advance: bad sizes B=%d beam=%d K=%d V=%d", B, beam, K, V);
  hipLaunchKernelGGL(topk_merge_advance_kernel, dim3(B), dim3(1024), 0, stream, vals1, idx1, beam, K, V, vals, idx, eos, other, pack,
                     2 * K + (eos ? 2 * beam : 0), eos_token, scores, tokens, prev_tokens, beam_idx);
  SC_LAUNCH_CHECK("topk_merge_advance_kernel");
  return SC_OK;
}

namespace {
// row b of pack = [vals (K) | idx (K) | eos (beam) | other (beam)] as float64 (the last two blocks only when eos is given)
__global__ __launch_bounds__(256) void beam_pack_kernel(const float* vals, const long long* idx, const float* eos, const float* other, int K, int beam,
                                                        double* pack, int width) {
  const int b = blockIdx.x;
  double* row = pack + (long long)b * width;
  for (int i = threadIdx.x; i < width; i += 256) {
    double v;
    if (i < K) v = (double)vals[(long long)b * K + i];
    else if (i < 2 * K) v = (double)idx[(long long)b * K + i - K];
    else if (i < 2 * K + beam) v = (double)eos[(long long)b * beam + i - 2 * K];
    else v = (double)other[(long long)b * beam + i - 2 * K - beam];
    row[i] = v;
  }
}
}  // namespace

// The device-to-host record of a beam-search step in one launch: pack (B, 2 K [+ 2 beam]) float64, row b = [the K candidate scores | their
// indices | the beams' <eos> scores | the beams' best other scores] (the last two blocks with eos != null: <eos> filtering,
// infer_func/beam_search.py:455-462).  Indices up to 2^53 are exact in float64.
SC_API int sc_beam_pack(const float* vals, const long long* idx, const float* eos, const float* other, int B, int K, int beam, double* pack,
                        hipStream_t stream) {
  SC_CHECK_ARG(vals && idx && pack && B >= 1 && B <= 65535 && K >= 1 && beam >= 1 && ((eos == nullptr) == (other == nullptr)),
               "sc_beam_pack: bad arguments");
  hipLaunchKernelGGL(beam_pack_kernel, dim3(B), dim3(256), 0, stream, vals, idx, eos, other, K, beam, pack, 2 * K + (eos ? 2 * beam : 0));
  SC_LAUNCH_CHECK("beam_pack_kernel");
  return SC_OK;
}

namespace {
// beam_pack_kernel plus the next step's rows of sentence b: the first `beam` candidates in rank order whose token is not <sos/eos>
// (beam_search.py:423-509 keeps exactly those for a running sentence; a row holds one <sos/eos> column, so at most `beam` of the 2 beam
// candidates are skipped).  prev_tokens <- tokens before tokens is overwritten.
__global__ __launch_bounds__(256) void beam_advance_kernel(const float* vals, const long long* idx, const float* eos, const float* other, int K, int beam,
                                                           double* pack, int width, int V, int eos_token, float* scores, long long* tokens,
                                                           long long* prev_tokens, long long* beam_idx) {
  __shared__ float sv[128];
  __shared__ long long si[128];
  const int b = blockIdx.x, tid = threadIdx.x;
  double* row = pack + (long long)b * width;
  for (int i = tid; i < width; i += 256) {
    double v;
    if (i < K) v = (double)vals[(long long)b * K + i];
    else if (i < 2 * K) v = (double)idx[(long long)b * K + i - K];
    else if (i < 2 * K + beam) v = (double)eos[(long long)b * beam + i - 2 * K];
    else v = (double)other[(long long)b * beam + i - 2 * K - beam];
    row[i] = v;
  }
  if (tid < K) {
    sv[tid] = vals[(long long)b * K + tid];
    si[tid] = idx[(long long)b * K + tid];
  }
  if (tid < beam) prev_tokens[(long long)b * beam + tid] = tokens[(long long)b * beam + tid];
  __syncthreads();
  if (tid == 0) {
    int kept = 0;
    for (int i = 0; i < K && kept < beam; ++i) {
      const long long id = si[i];
      const int tok = (int)(id % V);
      if (tok == eos_token) continue;
      const long long r = (long long)b * beam + kept;
      scores[r] = sv[i];
      tokens[r] = tok;
      beam_idx[r] = (long long)b * beam + id / V;
      ++kept;
    }
  }
}
}  // namespace

// sc_beam_pack and, in the same launch, the rows of the next step written on the device: scores / tokens / beam_idx (B * beam) <- the first
// `beam` non-<sos/eos> candidates of every sentence in rank order, prev_tokens <- the tokens of this step.  The host reads the packed record
// for its n-best lists and stopping rules while the device is already running the next position.  K <= 128, K >= 2 * beam.
SC_API int sc_beam_advance(const float* vals, const long long* idx, const float* eos, const float* other, int B, int K, int beam, int V,
                           int eos_token, double* pack, float* scores, long long* tokens, long long* prev_tokens, long long* beam_idx,
                           hipStream_t stream) {
  SC_CHECK_ARG(vals && idx && pack && scores && tokens && prev_tokens && beam_idx && ((eos == nullptr) == (other == nullptr)), "sc_beam_advance: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && beam >= 1 && K >= 2 * beam && K <= 128 && V >= 2 && eos_token >= 0 && eos_token < V,
               "sc_beam_advance: bad sizes (B=%d K=%d beam=%d V=%d)", B, K, beam, V);
  hipLaunchKernelGGL(beam_advance_kernel, dim3(B), dim3(256), 0, stream, vals, idx, eos, other, K, beam, pack, 2 * K + (eos ? 2 * beam : 0), V,
                     eos_token, scores, tokens, prev_tokens, beam_idx);
  SC_LAUNCH_CHECK("beam_advance_kernel");
  return SC_OK;
}

// vals / idx (R, K): the K largest elements of every row of x (R, L; row stride ld) in descending order, ties by the lower index.
SC_API int sc_topk_rows(const float* x, long long ld, int R, int L, int K, float* vals, long long* idx, hipStream_t stream) {
  SC_CHECK_ARG(x && vals && idx, "sc_topk_rows: null pointer");
  SC_CHECK_ARG(R >= 1 && R <= 65535 && L >= 1 && K >= 1 && K <= 64 && K <= L && ld >= L, "sc_topk_rows: bad sizes R=%d L=%d K=%d (K <= 64, K <= L)", R, L, K);
  hipLaunchKernelGGL(topk_rows_kernel, dim3(R), dim3(1024), 0, stream, x, ld, L, K, vals, idx);
  SC_LAUNCH_CHECK("topk_rows_kernel");
  return SC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Candidate scores of a beam-search step in one pass over the token head's logits (speechain/infer_func/beam_search.py:334-345,401-410):
//   cand[n][v] = (1 - w) * log_softmax(logits[n] / temperature)[v] + w * (psi[n][v] - psi_prev[n]) + beam_score[n]
// with the blank column of the attention term forced to -1e20 when CTC scores take part (:339), plus -- for <eos> filtering (:455-462) --
// the row's <eos> score and the best other score (both of the mixed token scores, before beam_score is added).
// One workgroup per hypothesis row; the row is read three times out of L1 / L2 (maximum, sum of exponentials, output).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void beam_candidates_kernel(const T* logits, long long ld, const float* psi, const float* psi_prev,
                                                              const float* beam_score, float* cand, float* eos_score, float* other_max,
                                                              int V, float inv_temp, float w, int blank, int eos) {
  __shared__ float red[8];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T* row = logits + (long long)n * ld;
  float m = -INFINITY;
  for (int v = tid; v < V; v += 256) m = fmaxf(m, Elem<T>::ld(row + v) * inv_temp);
  m = sc_wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int v = tid; v < V; v += 256) s += __expf(Elem<T>::ld(row + v) * inv_temp - m);
  s = sc_wave_sum(s);
  if (lane == 0) red[4 + wv] = s;
  __syncthreads();
  const float lse = m + __logf(red[4] + red[5] + red[6] + red[7]);
  const float bs = beam_score[n];
  const float pp = (psi && psi_prev) ? psi_prev[n] : 0.f;
  float best_other = -INFINITY;
  for (int v = tid; v < V; v += 256) {
    float t = Elem<T>::ld(row + v) * inv_temp - lse;
    if (psi) {
      if (v == blank) t = CTC_IMPOSSIBLE;
      t = (1.f - w) * t + w * (psi[(long long)n * V + v] - pp);
    }
    cand[(long long)n * V + v] = t + bs;
    if (v == eos) {
      if (eos_score) eos_score[n] = t;
    } else {
      best_other = fmaxf(best_other, t);
    }
  }
  if (other_max) {
    __syncthreads();
    best_other = sc_wave_max(best_other);
    if (lane == 0) red[wv] = best_other;
    __syncthreads();
    if (tid == 0) other_max[n] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// beam_candidates_kernel and the row's own top K (topk_rows_body) in one launch: the candidate row lives in LDS only -- the (N, V) score
// matrix (10 MB at N = 512, V = 5000) is never written, and the selection's four sweeps read LDS instead of L2.  1024 threads per row.
template <typename T, int PER>
__global__ __launch_bounds__(1024) void beam_candidates_topk_kernel(const T* logits, long long ld, const float* psi, const float* psi_prev,
                                                                   const float* beam_score, float* eos_score, float* other_max, int V,
                                                                   float inv_temp, float w, int blank, int eos, int K, float* vals,
                                                                   long long* idx) {
  extern __shared__ float candrow[];  // [V]
  __shared__ float red[32];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T* row = logits + (long long)n * ld;
  // the row's logits and prefix scores: PER of each per thread, all loads in flight at once (the three sweeps below read registers)
  float x[PER], ps[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int v = min(tid + 1024 * u, V - 1);
    x[u] = Elem<T>::ld(row + v) * inv_temp;
  }
  if (psi) {
#pragma unroll
    for (int u = 0; u < PER; ++u) ps[u] = psi[(long long)n * V + min(tid + 1024 * u, V - 1)];
  }
  float m = -INFINITY;
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (tid + 1024 * u < V) m = fmaxf(m, x[u]);
  m = sc_wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (tid + 1024 * u < V) s += __expf(x[u] - m);
  s = sc_wave_sum(s);
  if (lane == 0) red[16 + wv] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += red[16 + i];
  const float lse = m + __logf(tot);
  const float bs = beam_score[n];
  const float pp = (psi && psi_prev) ? psi_prev[n] : 0.f;
  float best_other = -INFINITY;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int v = tid + 1024 * u;
    if (v < V) {
      float t = x[u] - lse;
      if (psi) {
        if (v == blank) t = CTC_IMPOSSIBLE;
        t = (1.f - w) * t + w * (ps[u] - pp);
      }
      candrow[v] = t + bs;
      if (v == eos) {
        if (eos_score) eos_score[n] = t;
      } else {
        best_other = fmaxf(best_other, t);
      }
    }
  }
  __syncthreads();
  if (other_max) {
    best_other = sc_wave_max(best_other);
    if (lane == 0) red[wv] = best_other;
    __syncthreads();
    if (tid == 0) {
      float o = red[0];
      for (int i = 1; i < 16; ++i) o = fmaxf(o, red[i]);
      other_max[n] = o;
    }
  }
  topk_rows_body(candrow, V, K, vals + (long long)n * K, idx + (long long)n * K);
}

}  // namespace

// cand (N, V) f32 as above.  dtype of logits: 0 = f32, 1 = bf16, row stride ld.  psi (N, V) / psi_prev (N) may be NULL (no CTC term / first
// step); eos_score, other_max (N) may be NULL.
// sc_beam_candidates followed by sc_topk_rows on its result, in one launch and without the (N, V) score matrix: vals / idx (N, K), every
// row's K best candidate scores (descending, ties by the lower index) and their columns.  V * 4 bytes of LDS per workgroup: V <= 12288.
SC_API int sc_beam_candidates_topk(int dtype, const void* logits, long long ld, const float* psi, const float* psi_prev, const float* beam_score,
                                   float* eos_score, float* other_max, int N, int V, float temperature, float ctc_weight, int blank, int eos,
                                   int K, float* vals, long long* idx, hipStream_t stream) {
  SC_CHECK_ARG(logits && beam_score && vals && idx, "sc_beam_candidates_topk: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_beam_candidates_topk: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(N >= 1 && N <= 65535 && V >= 1 && V <= 12288 && ld >= V && temperature > 0.f && blank >= 0 && blank < V && eos >= 0 && eos < V &&
                   K >= 1 && K <= 64 && K <= V,
               "sc_beam_candidates_topk: bad sizes N=%d V=%d K=%d (V <= 12288, K <= 64)", N, V, K);
  const size_t smem = (size_t)V * sizeof(float);
#define SC_LAUNCH_CT(T, PER)                                                                                                            \
  hipLaunchKernelGGL((beam_candidates_topk_kernel<T, PER>), dim3(N), dim3(1024), smem, stream, (const T*)logits, ld, psi, psi_prev, beam_score, \
                     eos_score, other_max, V, 1.f / temperature, ctc_weight, blank, eos, K, vals, idx)
  // (elements per thread: 6 covers V <= 6144, 12 the LDS limit)
  if (dtype == 0) {
    if (V <= 6144) SC_LAUNCH_CT(float, 6);
    else SC_LAUNCH_CT(float, 12);
  } else {
    if (V <= 6144) SC_LAUNCH_CT(unsigned short, 6);
    else SC_LAUNCH_CT(unsigned short, 12);
  }
#undef SC_LAUNCH_CT
  SC_LAUNCH_CHECK("beam_candidates_topk_kernel");
  return SC_OK;
}

SC_API int sc_beam_candidates(int dtype, const void* logits, long long ld, const float* psi, const float* psi_prev, const float* beam_score,
                              float* cand, float* eos_score, float* other_max, int N, int V, float temperature, float ctc_weight, int blank,
                              int eos, hipStream_t stream) {
  SC_CHECK_ARG(logits && beam_score && cand, "sc_beam_candidates: null pointer");
  SC_CHECK_ARG(dtype == 0 || dtype == 1, "sc_beam_candidates: dtype must be 0 (f32) or 1 (bf16)");
  SC_CHECK_ARG(N >= 1 && N <= 0x7fffffff && V >= 1 && ld >= V && temperature > 0.f && blank >= 0 && blank < V && eos >= 0 && eos < V,
               "sc_beam_candidates: bad sizes N=%d V=%d", N, V);
  if (dtype == 0)
    hipLaunchKernelGGL(beam_candidates_kernel<float>, dim3(N), dim3(256), 0, stream, (const float*)logits, ld, psi, psi_prev, beam_score, cand,
                       eos_score, other_max, V, 1.f / temperature, ctc_weight, blank, eos);
  else
    hipLaunchKernelGGL(beam_candidates_kernel<unsigned short>, dim3(N), dim3(256), 0, stream, (const unsigned short*)logits, ld, psi, psi_prev,
                       beam_score, cand, eos_score, other_max, V, 1.f / temperature, ctc_weight, blank, eos);
  SC_LAUNCH_CHECK("beam_candidates_kernel");
  return SC_OK;
}
