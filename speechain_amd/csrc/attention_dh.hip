// Fused scaled-dot-product attention for head sizes other than 64 (bf16, any head size that is a multiple of 8 up to 256; any number of
// keys; key-padding masks and the causal flag) -- matmul(q, k^T) * scale -> masked_fill(-inf) -> softmax -> dropout -> matmul(attn, v) of
// speechain/module/transformer/attention.py:88-106,131 without the (B, H, Tq, Tk) tensors in HBM, forward and backward.  The reference
// is head-size agnostic (attention.py:53-65); FastSpeech2 runs two heads of 192 (recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:
// 40-43).  csrc/attention.hip holds the kernels tuned for the ASR models' head size 64; these are their generalisation:
//
// The head dimension is cut into NC chunks of 64 (zero-padded: a chunk column beyond the head size is read from a valid address of the same
// row -- finite numbers -- and met by ZERO query / gradient elements, or produces output columns that are never stored).  On chip every
// chunk of a K / V / Q / dO tile has the layout of the 64-wide kernels (128-byte rows, 16-byte slot index XOR-ed with row & 7 on the source
// side of the LDS-DMA and in the read addresses), so the fragment addressing is theirs with a chunk offset.
//
//   attn_dh_fwd_kernel      S^T = K Q^T per 64-key tile with ONE QUERY PER LANE (online softmax in registers), O^T += V^T P~^T; K / V tiles
//                           stream through a two-stage LDS ring by LDS-DMA.  Saves the row logsumexp.
//   attn_dh_bwd_dq_kernel   the same sweep with P = exp(scale S - lse) from the saved logsumexp and dP^T = V dO^T beside S^T:
//                           dS^T = P (keep-mask dP / (1 - p) - delta) scale, dQ^T += K^T dS^T.  One workgroup per 128 queries.
//   attn_dh_bwd_dkv_kernel  ONE KEY PER LANE: a wave keeps K / V of its 32 keys as B fragments and dK^T, dV^T as accumulators and sweeps the
//                           queries 64 at a time (Q / dO tiles staged through LDS with their row constants -lse / scale and -delta (1 - p) as
//                           the initial accumulators): dV^T += dO^T P~, dK^T += Q^T dS.  One workgroup per 128 keys.
// The backward pass recomputes S and dP in both kernels (7 products instead of 5) and in exchange needs no atomics, no partial-sum
// workspace and no cross-workgroup order: deterministic, and parallel over (batch, head, 128-row block) -- 320 workgroups for the
// FastSpeech2 decoder's 32 x 2 heads x 640 frames, where a (batch, head)-per-workgroup sweep would occupy a quarter of the chip.
// Dropout indices, the logsumexp definition and the zero output of an all-masked row are those of csrc/attention.hip.
#include "sc_common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((address_space(1))) const void* dh_gptr_t;
typedef __attribute__((address_space(3))) void* dh_lptr_t;

struct DhArgs {
  const unsigned short* Q;
  const unsigned short* K;
  const unsigned short* V;
  unsigned short* O;        // forward: output; backward: the forward's output (read)
  float* LSE;
  long long q_b, ldq, kv_b, ldk, o_b, ldo;
  const unsigned char* mask;  // (B, 1, Tk) key mask or null
  long long mask_b;
  int causal;
  int B, H, Tq, Tk, dh, ldp;
  float scale, drop_p, drop_keep;
  const unsigned long long* seed;
  unsigned long long offset;
  const unsigned short* dO;
  unsigned short* dQ;
  unsigned short* dK;
  unsigned short* dV;
  long long dq_b, lddq, dkv_b, lddk;
  // backward workspace, three (B H, Tqp) f32 planes with Tqp = Tq rounded up to 64, written by attn_dh_delta_kernel:
  //   delta = rowsum(dO * O);  nl = -lse / scale (-inf for rows without a visible key and for the padding rows);  nd = -delta (1 - p_drop)
  float* delta;
  float* nl;
  float* nd;
  int Tqp;
};

constexpr int KT = 64;          // keys (or queries) per streamed tile
constexpr int CH = KT * 64;     // elements of one 64 x 64 chunk (8 KB)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// row-major MFMA operand fragment of a swizzled 64 x 64 chunk: row `row`, elements 16 ks + 8 h .. + 7
__device__ __forceinline__ bf16x8_t frag_rows(const unsigned short* chunk, int row, int ks, int h) {
  return *(const bf16x8_t*)(chunk + row * 64 + (((2 * ks + h) ^ (row & 7)) << 3));
}
// transposed operand fragment (contraction over the ROWS of the chunk): this lane's column block, rows kbase and kbase + 8 (4 rows each)
__device__ __forceinline__ bf16x8_t frag_cols(const unsigned short* chunk, int kbase, int col) {
  const unsigned short* a0 = chunk + kbase * 64 + (((col >> 3) ^ (kbase & 7)) << 3) + (col & 7);
  const unsigned short* a1 = chunk + (kbase + 8) * 64 + (((col >> 3) ^ ((kbase + 8) & 7)) << 3) + (col & 7);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a1);
  const s16x8_t v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v8);
}

// The same read as inline asm: behind a global_load_lds hipcc puts s_waitcnt vmcnt(0) in front of every transposing LDS read it can see (it
// cannot tell which DMA a read depends on) -- in a ring with tiles in flight that drains the prefetch once per tile.  The reads are issued
// in batches and waited for by an s_waitcnt that takes the registers it guards as in/out operands (a consumer can then not be scheduled in
// front of the wait: for inline asm the compiler believes the registers are delivered when the statement ends).
struct TrFrag {
  s16x4_t lo, hi;
};
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void frag_cols_issue(const unsigned short* chunk, int kbase, int col, TrFrag& f) {
  const unsigned a0 = lds_addr(chunk + kbase * 64 + (((col >> 3) ^ (kbase & 7)) << 3) + (col & 7));
  const unsigned a1 = lds_addr(chunk + (kbase + 8) * 64 + (((col >> 3) ^ ((kbase + 8) & 7)) << 3) + (col & 7));
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0), "v"(a1));
}
template <int N>
__device__ __forceinline__ void frag_cols_wait(TrFrag (&f)[N]) {
  if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi));
  else if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi));
  else if constexpr (N == 6)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi),
                 "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi),
                 "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi), "+v"(f[6].lo), "+v"(f[6].hi), "+v"(f[7].lo), "+v"(f[7].hi));
}
__device__ __forceinline__ bf16x8_t frag_of(const TrFrag& f) {
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// A tile of 64 rows x (64 NC) columns of a strided bf16 matrix -> NC swizzled chunks at `dst`, by LDS-DMA: wave w of 4 moves granules
// (8 rows) 2 w and 2 w + 1 of every chunk.  Rows beyond `nrows` are clamped (the caller masks them), columns beyond `dh` are redirected to
// column 0 of the same row (see the file header).
template <int NC>
__device__ __forceinline__ void dma_tile(unsigned short* dst, const unsigned short* src, long long ld, int row0, int nrows, int dh, int wid,
                                         int lane) {
  const int grow = lane >> 3, gch = (lane & 7) ^ (grow & 7);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = c * 64 + gch * 8;
    col = col < dh ? col : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wid + i;
      const long long off = (long long)min(row0 + g * 8 + grow, nrows - 1) * ld + col;
      __builtin_amdgcn_global_load_lds((dh_gptr_t)(src + off), (dh_lptr_t)(dst + c * CH + g * 512), 16, 0, 0);
    }
  }
}

// this lane's row of a strided matrix as B fragments: elements 64 c + 16 ks + 8 h .. + 7 (zero beyond dh)
template <int NC>
__device__ __forceinline__ void load_row_frags(bf16x8_t (&f)[4 * NC], const unsigned short* row, int dh, int h) {
#pragma unroll
  for (int i = 0; i < 4 * NC; ++i) {
    const int d0 = i * 16 + h * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (d0 < dh) v = *(const uint4*)(row + d0);
    f[i] = __builtin_bit_cast(bf16x8_t, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// forward (MODE 0) and dQ (MODE 1): one 256-thread workgroup per (batch, head, 128 queries); a lane owns one query
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int NC, bool DROP, int MODE>
__global__ __launch_bounds__(256, 1) void attn_dh_q_kernel(DhArgs p) {
  constexpr int STAGE = 2 * NC * CH;  // K chunks, then V chunks
  constexpr int NS = NC <= 3 ? 3 : 2;  // ring stages (48 KB each at NC = 3): two tiles in flight behind the one being multiplied where LDS allows
  constexpr int LPT = 4 * NC;          // LDS-DMA instructions per wave and tile
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int z = blockIdx.y, qt = blockIdx.x;
  const int b = z / p.H, hd = z - b * p.H;
  const int qg = qt * 128 + wid * 32 + li;
  const int qc = min(qg, p.Tq - 1);
  const unsigned short* Kg = p.K + b * p.kv_b + (long long)hd * p.dh;
  const unsigned short* Vg = p.V + b * p.kv_b + (long long)hd * p.dh;
  const int nt = (p.Tk + KT - 1) / KT;

  bf16x8_t qf[4 * NC];
  load_row_frags<NC>(qf, p.Q + b * p.q_b + (long long)hd * p.dh + (long long)qc * p.ldq, p.dh, h);
  bf16x8_t gf[4 * NC];             // dO fragments (dQ kernel; unused and optimised away in the forward)
  float delta = 0.f, lse2 = 0.f;   // rowsum(dO * O) and the saved logsumexp in log2 units (dQ kernel)
  if constexpr (MODE == 1) {
    load_row_frags<NC>(gf, p.dO + b * p.o_b + (long long)hd * p.dh + (long long)qc * p.ldo, p.dh, h);
    delta = p.delta[(long long)z * p.Tqp + qc];
    const float lse = p.LSE[(long long)z * p.Tq + qc];
    // a query beyond Tq or a row without a visible key (lse = -inf): every probability is zero
    lse2 = (qg < p.Tq && lse != -INFINITY) ? lse * 1.4426950408889634f : INFINITY;
  }
  const unsigned char* mrow = p.mask ? p.mask + b * p.mask_b : nullptr;
  // key validity, lane i answers for key 64 t + i.  The mask byte of tile t + 1 is loaded in iteration t by an inline-asm load (a load the
  // compiler can see makes it wait vmcnt(0) at the use and drain the ring) right BEFORE the DMA of tile t + 2 is issued, so the counted wait
  // at the top of iteration t + 1 covers it; that wait takes the register as an in/out operand (no use can be scheduled in front of it)
  auto key_byte = [&](int t_) -> unsigned {
    unsigned v = 1u;
    if (mrow) {
      const unsigned char* a = mrow + min(t_ * KT + lane, p.Tk - 1);
      asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(a) : "memory");
    }
    return v;
  };
  const float c1 = p.scale * 1.4426950408889634f;
  const unsigned long long seed = DROP ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
  const unsigned long long pairq = ((p.offset + ((unsigned long long)z * p.Tq + qc) * p.ldp) >> 1) + (unsigned)(2 * h);
  const int qwave0 = qt * 128 + wid * 32;

  f32x16 o[2 * NC];  // forward: O^T; dQ kernel: dQ^T (head dims x queries)
#pragma unroll
  for (int j = 0; j < 2 * NC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  unsigned kbyte = key_byte(0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(kbyte)::"memory");  // nothing of the prologue stays in flight into the counted waits of the loop
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nt) {
      dma_tile<NC>(dsm + i * STAGE, Kg, p.ldk, i * KT, p.Tk, p.dh, wid, lane);
      dma_tile<NC>(dsm + i * STAGE + NC * CH, Vg, p.ldk, i * KT, p.Tk, p.dh, wid, lane);
    }
  const int g = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
  for (int t = 0; t < nt; ++t) {
    // tile t and this tile's mask byte have landed for this wave; tile t + 1 may stay in flight (in-order queue: DMA(t), byte(t), DMA(t + 1)) ...
    if (NS == 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kbyte) : "n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(kbyte)::"memory");
    __builtin_amdgcn_s_barrier();  // ... for everybody; and everybody is done reading the stage that tile t + NS - 1 goes to
    const bool kok = t * KT + lane < p.Tk && kbyte != 0u;
    kbyte = key_byte(t + 1);  // (a clamped address; before the DMA below)
    if (t + NS - 1 < nt) {
      unsigned short* nx = dsm + ((t + NS - 1) % NS) * STAGE;
      dma_tile<NC>(nx, Kg, p.ldk, (t + NS - 1) * KT, p.Tk, p.dh, wid, lane);
      dma_tile<NC>(nx + NC * CH, Vg, p.ldk, (t + NS - 1) * KT, p.Tk, p.dh, wid, lane);
    }
    const unsigned short* ks_ = dsm + (t % NS) * STAGE;
    const unsigned short* vs_ = ks_ + NC * CH;
    const unsigned long long bits = __ballot(kok);

    // ---- S^T = K Q^T (and dP^T = V dO^T): two 32-key blocks x 32 queries ----
    f32x16 s2[2], dp2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s2[j][r] = dp2[j][r] = 0.f;
      const int krow = 32 * j + li;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(ks_ + c * CH, krow, ks, h), qf[c * 4 + ks], s2[j], 0, 0, 0);
          if constexpr (MODE == 1)
            dp2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vs_ + c * CH, krow, ks, h), gf[c * 4 + ks], dp2[j], 0, 0, 0);
        }
    }
    // ---- masks (wave-uniform test: a tile with an invisible key, or one the causal diagonal runs through) ----
    if (bits != ~0ull || (p.causal && t * KT + KT - 1 > qwave0)) {
      const int dq = p.causal ? qg - t * KT - 4 * h : KT;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned wj = (unsigned)(bits >> (32 * j)) >> (4 * h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kb = (r & 3) + 8 * (r >> 2);
          const bool ok = (wj & (1u << kb)) != 0u && 32 * j + kb <= dq;
          s2[j][r] = ok ? s2[j][r] : -INFINITY;
        }
      }
    }
    bf16x8_t pf[4];  // B operand of the second product: k-step ks = block ks >> 1, registers 8 (ks & 1) .. + 7
    if constexpr (MODE == 0) {
      float mloc = -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s2[j][r]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float m_new = fmaxf(m_run, mloc * c1);
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      const float nmsub = (m_new == -INFINITY) ? 0.f : -m_new;
      float lsum = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned pw[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p0 = __builtin_amdgcn_exp2f(fmaf(s2[j][r], c1, nmsub)), p1 = __builtin_amdgcn_exp2f(fmaf(s2[j][r + 1], c1, nmsub));
          lsum += p0 + p1;
          if (DROP) {
            const int kb = 32 * j + (r & 3) + 8 * (r >> 2);
            const unsigned hsh = sc_rand_pair(seed, pairq + (unsigned)(t * (KT / 2) + kb / 2));
            p0 = ((hsh & 0xFFFFu) < thr) ? 0.f : p0;
            p1 = ((hsh >> 16) < thr) ? 0.f : p1;
          }
          pw[r >> 1] = (unsigned)sc_f32_to_bf16(p0) | ((unsigned)sc_f32_to_bf16(p1) << 16);
        }
        pf[2 * j] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[0], pw[1], pw[2], pw[3]));
        pf[2 * j + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[4], pw[5], pw[6], pw[7]));
      }
      l_run = l_run * alpha + lsum;
      m_run = m_new;
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int j = 0; j < 2 * NC; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
      }
    } else {
      // dS^T = P (keep-mask dP / (1 - p) - delta) scale with P = exp2(c1 S - lse2)
      const float nsub = -lse2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned pw[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(s2[j][r], c1, nsub)), p1 = __builtin_amdgcn_exp2f(fmaf(s2[j][r + 1], c1, nsub));
          float d0 = dp2[j][r], d1 = dp2[j][r + 1];
          if (DROP) {
            const int kb = 32 * j + (r & 3) + 8 * (r >> 2);
            const unsigned hsh = sc_rand_pair(seed, pairq + (unsigned)(t * (KT / 2) + kb / 2));
            d0 = ((hsh & 0xFFFFu) < thr) ? 0.f : d0 * p.drop_keep;
            d1 = ((hsh >> 16) < thr) ? 0.f : d1 * p.drop_keep;
          }
          const float e0 = p0 * (d0 - delta) * p.scale, e1 = p1 * (d1 - delta) * p.scale;
          pw[r >> 1] = (unsigned)sc_f32_to_bf16(e0) | ((unsigned)sc_f32_to_bf16(e1) << 16);
        }
        pf[2 * j] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[0], pw[1], pw[2], pw[3]));
        pf[2 * j + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[4], pw[5], pw[6], pw[7]));
      }
    }
    // ---- O^T += V^T P~^T  /  dQ^T += K^T dS^T: k-step ks covers keys 32 (ks >> 1) + 16 (ks & 1) + 4 h + {0..3, 8..11} ----
    {
      const unsigned short* os_ = MODE == 0 ? vs_ : ks_;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kbase = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * (g >> 1) + tq;
        TrFrag tf[2 * NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) frag_cols_issue(os_ + c * CH, kbase, 32 * jj + 16 * (g & 1) + 4 * tp, tf[2 * c + jj]);
        frag_cols_wait<2 * NC>(tf);
#pragma unroll
        for (int j = 0; j < 2 * NC; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(tf[j]), pf[ks], o[j], 0, 0, 0);
      }
    }
  }

  // ---- store: lane (query, half h) holds head dims 32 j + 8 (r >> 2) + 4 h + (r & 3) of block j ----
  float inv = 1.f;
  if constexpr (MODE == 0) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    inv = l > 0.f ? p.drop_keep / l : 0.f;
    if (qg < p.Tq && h == 0) p.LSE[(long long)z * p.Tq + qg] = l > 0.f ? (m_run + log2f(l)) * 0.6931471805599453f : -INFINITY;
  }
  if (qg < p.Tq) {
    unsigned short* orow = MODE == 0 ? p.O + b * p.o_b + (long long)hd * p.dh + (long long)qg * p.ldo
                                     : p.dQ + b * p.dq_b + (long long)hd * p.dh + (long long)qg * p.lddq;
#pragma unroll
    for (int j = 0; j < 2 * NC; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * j + 8 * r4 + 4 * h;
        if (d0 < p.dh) {
          const unsigned w0 = sc_f32_to_bf16(o[j][4 * r4] * inv) | ((unsigned)sc_f32_to_bf16(o[j][4 * r4 + 1] * inv) << 16);
          const unsigned w1 = sc_f32_to_bf16(o[j][4 * r4 + 2] * inv) | ((unsigned)sc_f32_to_bf16(o[j][4 * r4 + 3] * inv) << 16);
          *(uint2*)(orow + d0) = make_uint2(w0, w1);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// dK, dV: one 256-thread workgroup per (batch, head, 128 keys); a lane owns one key
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int NC, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_dh_bwd_dkv_kernel(DhArgs p) {
  constexpr int STAGE = 2 * NC * CH;  // Q chunks, then dO chunks
  constexpr int NS = NC <= 3 ? 3 : 2;
  constexpr int LPT = 4 * NC;
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsm[];
  float* rowc = (float*)(dsm + NS * STAGE);  // [NS stages][2][64]: -lse / scale, -delta (1 - p_drop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int z = blockIdx.y, b = z / p.H, hd = z - b * p.H;
  const int key = blockIdx.x * 128 + wid * 32 + li;
  const int kc = min(key, p.Tk - 1);
  const unsigned short* Qg = p.Q + b * p.q_b + (long long)hd * p.dh;
  const unsigned short* dOg = p.dO + b * p.o_b + (long long)hd * p.dh;
  const int nt = (p.Tq + KT - 1) / KT;
  const float c1 = p.scale * 1.4426950408889634f;

  bf16x8_t kf[4 * NC], vf[4 * NC];
  load_row_frags<NC>(kf, p.K + b * p.kv_b + (long long)hd * p.dh + (long long)kc * p.ldk, p.dh, h);
  load_row_frags<NC>(vf, p.V + b * p.kv_b + (long long)hd * p.dh + (long long)kc * p.ldk, p.dh, h);
  const bool kvalid = key < p.Tk && (p.mask == nullptr || p.mask[b * p.mask_b + kc] != 0);
  const bool need_poison = !__all(kvalid);
  const unsigned long long seed = DROP ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
  const int odd = li & 1;
  const unsigned wsel = odd ? 0x07060302u : 0x01000504u;
  const unsigned lterm = (unsigned)(key >> 1) + (unsigned)odd * (unsigned)(p.ldp >> 1) + (unsigned)h * (unsigned)(2 * p.ldp);

  f32x16 dk[2 * NC], dv[2 * NC];
#pragma unroll
  for (int d = 0; d < 2 * NC; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

  // row constants of a 64-query tile, the initial accumulators: two 4-byte-per-lane LDS-DMAs from the workspace planes (padded to whole
  // tiles: no clamping), issued by every wave alike so that all waves count the same number of DMA operations
  auto consts_dma = [&](int t_) {
    const long long off = (long long)z * p.Tqp + t_ * KT + lane;
    __builtin_amdgcn_global_load_lds((dh_gptr_t)(p.nl + off), (dh_lptr_t)(rowc + (t_ % NS) * 128), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((dh_gptr_t)(p.nd + off), (dh_lptr_t)(rowc + (t_ % NS) * 128 + 64), 4, 0, 0);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prologue's loads are not part of the counted waits below
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nt) {
      dma_tile<NC>(dsm + i * STAGE, Qg, p.ldq, i * KT, p.Tq, p.dh, wid, lane);
      dma_tile<NC>(dsm + i * STAGE + NC * CH, dOg, p.ldo, i * KT, p.Tq, p.dh, wid, lane);
      consts_dma(i);
    }
  const int g = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
  for (int t = 0; t < nt; ++t) {
    if (NS == 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT + 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int tn = t + NS - 1;  // the tile whose stage is free now
    if (tn < nt) {
      unsigned short* nx = dsm + (tn % NS) * STAGE;
      dma_tile<NC>(nx, Qg, p.ldq, tn * KT, p.Tq, p.dh, wid, lane);
      dma_tile<NC>(nx + NC * CH, dOg, p.ldo, tn * KT, p.Tq, p.dh, wid, lane);
      consts_dma(tn);
    }
    const unsigned short* qs_ = dsm + (t % NS) * STAGE;
    const unsigned short* ds_ = qs_ + NC * CH;
    const float* nlc = rowc + (t % NS) * 128;
    const int q0 = t * KT;
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
      float ndk[16];
      {
        // (asm reads: the constants arrived by LDS-DMA, and a read the compiler can see behind one makes it wait vmcnt(0))
        f32x4 ca[4], cd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          asm volatile("ds_read_b128 %0, %1" : "=&v"(ca[j]) : "v"(lds_addr(nlc + qb * 32 + 8 * j + 4 * h)));
          asm volatile("ds_read_b128 %0, %1" : "=&v"(cd[j]) : "v"(lds_addr(nlc + 64 + qb * 32 + 8 * j + 4 * h)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ca[0]), "+v"(ca[1]), "+v"(ca[2]), "+v"(ca[3]), "+v"(cd[0]), "+v"(cd[1]), "+v"(cd[2]), "+v"(cd[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ndk[4 * j + i] = cd[j][i];
            s[4 * j + i] = ca[j][i];
            dp[4 * j + i] = cd[j][i];
          }
      }
      if (need_poison) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = kvalid ? s[r] : -INFINITY;
      }
      if (p.causal) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qg = q0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          s[r] = (key <= qg) ? s[r] : -INFINITY;
        }
      }
      const int qrow = qb * 32 + li;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(qs_ + c * CH, qrow, ks, h), kf[c * 4 + ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(ds_ + c * CH, qrow, ks, h), vf[c * 4 + ks], dp, 0, 0, 0);
        }
      unsigned pw[8], dw[8];
      const unsigned long long rowbase0 = (p.offset + ((unsigned long long)z * p.Tq + q0 + qb * 32) * p.ldp) >> 1;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[r] * c1), p1 = __builtin_amdgcn_exp2f(s[r + 1] * c1);
        float pd0 = p0, pd1 = p1, sel0 = dp[r], sel1 = dp[r + 1];
        if (DROP) {
          const unsigned long long rowbase = rowbase0 + (unsigned long long)(((r & 3) + 8 * (r >> 2)) * (p.ldp >> 1));
          const unsigned hs = sc_rand_pair(seed, rowbase + lterm);
          const unsigned gs = (unsigned)__builtin_amdgcn_update_dpp((int)hs, (int)hs, 0xB1, 0xF, 0xF, false);
          const unsigned w = __builtin_amdgcn_perm(hs, gs, wsel);
          const bool d0 = (w & 0xFFFFu) < thr, d1 = (w >> 16) < thr;
          pd0 = d0 ? 0.f : p0;
          pd1 = d1 ? 0.f : p1;
          sel0 = d0 ? ndk[r] : sel0;
          sel1 = d1 ? ndk[r + 1] : sel1;
        }
        pw[r >> 1] = (unsigned)sc_f32_to_bf16(pd0) | ((unsigned)sc_f32_to_bf16(pd1) << 16);
        dw[r >> 1] = (unsigned)sc_f32_to_bf16(p0 * sel0) | ((unsigned)sc_f32_to_bf16(p1 * sel1) << 16);
      }
      bf16x8_t pf[2], dsf[2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        pf[ss] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]));
        dsf[ss] = __builtin_bit_cast(bf16x8_t, make_uint4(dw[4 * ss], dw[4 * ss + 1], dw[4 * ss + 2], dw[4 * ss + 3]));
      }
      // dV^T += dO^T P~', dK^T += Q^T dS': k-step ss covers queries 32 qb + 16 ss + 4 h + {0..3, 8..11}
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int rbase = qb * 32 + 16 * ss + 4 * (g >> 1) + tq;
        constexpr int G = NC == 4 ? 1 : NC;  // chunks per batch of reads (four chunks at once do not fit the register file beside 256 accumulators)
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += G) {
          TrFrag td[2 * G], tqf[2 * G];
#pragma unroll
          for (int c = 0; c < G; ++c)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) frag_cols_issue(ds_ + (c0 + c) * CH, rbase, 32 * jj + 16 * (g & 1) + 4 * tp, td[2 * c + jj]);
#pragma unroll
          for (int c = 0; c < G; ++c)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) frag_cols_issue(qs_ + (c0 + c) * CH, rbase, 32 * jj + 16 * (g & 1) + 4 * tp, tqf[2 * c + jj]);
          frag_cols_wait<2 * G>(td);
          frag_cols_wait<2 * G>(tqf);
#pragma unroll
          for (int j = 0; j < 2 * G; ++j) {
            dv[2 * c0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(td[j]), pf[ss], dv[2 * c0 + j], 0, 0, 0);
            dk[2 * c0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(tqf[j]), dsf[ss], dk[2 * c0 + j], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- dK (x scale / (1 - p)), dV (x 1 / (1 - p)): lane (key, half h) holds head dims 32 j + 8 (r >> 2) + 4 h + (r & 3) of block j ----
  if (key < p.Tk) {
    const float fk = p.scale * p.drop_keep, fv = p.drop_keep;
    unsigned short* krow = p.dK + b * p.dkv_b + (long long)hd * p.dh + (long long)key * p.lddk;
    unsigned short* vrow = p.dV + b * p.dkv_b + (long long)hd * p.dh + (long long)key * p.lddk;
#pragma unroll
    for (int j = 0; j < 2 * NC; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * j + 8 * r4 + 4 * h;
        if (d0 < p.dh) {
          *(uint2*)(krow + d0) = make_uint2(cvt_pk_bf16(dk[j][4 * r4] * fk, dk[j][4 * r4 + 1] * fk), cvt_pk_bf16(dk[j][4 * r4 + 2] * fk, dk[j][4 * r4 + 3] * fk));
          *(uint2*)(vrow + d0) = make_uint2(cvt_pk_bf16(dv[j][4 * r4] * fv, dv[j][4 * r4 + 1] * fv), cvt_pk_bf16(dv[j][4 * r4 + 2] * fv, dv[j][4 * r4 + 3] * fv));
        }
      }
  }
}

// Workspace planes of the backward pass, rows (z, q) with q padded to Tqp: delta = sum_d dO[q][d] O[q][d], nl = -lse / scale (-inf for a row
// without a visible key and for padding rows), nd = -delta (1 - p_drop).  32 lanes per row (8 elements each), 8 rows per 256-thread workgroup.
__global__ __launch_bounds__(256) void attn_dh_delta_kernel(DhArgs p) {
  const int tid = threadIdx.x, sub = tid & 31;
  const long long row = (long long)blockIdx.x * 8 + (tid >> 5);  // over (z, padded q)
  const long long nrows = (long long)p.B * p.H * p.Tqp;
  float dot = 0.f, lse = -INFINITY;
  if (row < nrows) {
    const int z = (int)(row / p.Tqp), q = (int)(row - (long long)z * p.Tqp);
    if (q < p.Tq) {
      const int b = z / p.H, hd = z - b * p.H;
      const int d0 = sub * 8;
      if (d0 < p.dh) {
        const uint4 vd = *(const uint4*)(p.dO + b * p.o_b + (long long)hd * p.dh + (long long)q * p.ldo + d0);
        const uint4 vo = *(const uint4*)(p.O + b * p.o_b + (long long)hd * p.dh + (long long)q * p.ldo + d0);
        const unsigned wd[4] = {vd.x, vd.y, vd.z, vd.w}, wo[4] = {vo.x, vo.y, vo.z, vo.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dot += __uint_as_float(wd[j] << 16) * __uint_as_float(wo[j] << 16);
          dot += __uint_as_float(wd[j] & 0xFFFF0000u) * __uint_as_float(wo[j] & 0xFFFF0000u);
        }
      }
      if (sub == 0) lse = p.LSE[(long long)z * p.Tq + q];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (row < nrows && sub == 0) {
    p.delta[row] = dot;
    p.nl[row] = (lse == -INFINITY) ? -INFINITY : -lse / p.scale;
    p.nd[row] = -dot * (1.f - p.drop_p);
  }
}

int fill_args(DhArgs& a, const void* Q, const void* K, const void* V, const void* O, float* LSE, int B, int H, int Tq, int Tk, int dh,
              long long q_b, long long ldq, long long kv_b, long long ldk, long long o_b, long long ldo, const unsigned char* mask,
              long long mask_b, int causal, float scale, float drop_p, const unsigned long long* seed, unsigned long long offset,
              const char* who) {
  SC_CHECK_ARG(Q && K && V && O && LSE, "%s: null pointer", who);
  SC_CHECK_ARG(B >= 1 && H >= 1 && Tq >= 1 && Tk >= 1 && (long long)B * H <= 65535, "%s: bad sizes B=%d H=%d Tq=%d Tk=%d", who, B, H, Tq, Tk);
  SC_CHECK_ARG(dh >= 8 && dh <= 256 && dh % 8 == 0, "%s: head size %d (a multiple of 8 up to 256)", who, dh);
  SC_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0 && q_b % 8 == 0 && kv_b % 8 == 0 && o_b % 8 == 0 && (uintptr_t)Q % 16 == 0 &&
                   (uintptr_t)K % 16 == 0 && (uintptr_t)V % 16 == 0 && (uintptr_t)O % 16 == 0,
               "%s: operands must be 16-byte aligned with strides multiple of 8", who);
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed) && offset % 2 == 0, "%s: bad dropout arguments", who);
  a.Q = (const unsigned short*)Q; a.K = (const unsigned short*)K; a.V = (const unsigned short*)V; a.O = (unsigned short*)const_cast<void*>(O);
  a.LSE = LSE; a.q_b = q_b; a.ldq = ldq; a.kv_b = kv_b; a.ldk = ldk; a.o_b = o_b; a.ldo = ldo; a.mask = mask; a.mask_b = mask_b;
  a.causal = causal; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.dh = dh; a.ldp = (Tk + 7) / 8 * 8; a.scale = scale; a.drop_p = drop_p;
  a.drop_keep = 1.f / (1.f - drop_p); a.seed = seed; a.offset = offset;
  return SC_OK;
}

template <int NC, int MODE>
int launch_q(const DhArgs& a, hipStream_t stream) {
  const size_t smem = (size_t)(NC <= 3 ? 3 : 2) * 2 * NC * CH * sizeof(unsigned short);
  const void* fn = a.drop_p > 0.f ? (const void*)attn_dh_q_kernel<NC, true, MODE> : (const void*)attn_dh_q_kernel<NC, false, MODE>;
  SC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  DhArgs args = a;
  void* kargs[] = {&args};
  SC_HIP(hipLaunchKernel(fn, dim3(sc_cdiv(a.Tq, 128), a.B * a.H), dim3(256), kargs, smem, stream));
  return SC_OK;
}

template <int NC>
int launch_dkv(const DhArgs& a, hipStream_t stream) {
  const size_t smem = (size_t)(NC <= 3 ? 3 : 2) * 2 * NC * CH * sizeof(unsigned short) + 3 * 128 * sizeof(float);
  const void* fn = a.drop_p > 0.f ? (const void*)attn_dh_bwd_dkv_kernel<NC, true> : (const void*)attn_dh_bwd_dkv_kernel<NC, false>;
  SC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  DhArgs args = a;
  void* kargs[] = {&args};
  SC_HIP(hipLaunchKernel(fn, dim3(sc_cdiv(a.Tk, 128), a.B * a.H), dim3(256), kargs, smem, stream));
  return SC_OK;
}

}  // namespace

SC_API int sc_attention_dh_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Tq, int Tk, int head_dim,
                               long long q_b, long long ldq, long long kv_b, long long ldk, long long o_b, long long ldo,
                               const unsigned char* mask, long long mask_b, int causal, float scale, float drop_p,
                               const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  DhArgs a{};
  int rc = fill_args(a, Q, K, V, O, LSE, B, H, Tq, Tk, head_dim, q_b, ldq, kv_b, ldk, o_b, ldo, mask, mask_b, causal, scale, drop_p, seed, offset,
                     "sc_attention_dh_fwd");
  if (rc != SC_OK) return rc;
  const int nc = (head_dim + 63) / 64;
  if (nc == 1) return launch_q<1, 0>(a, stream);
  if (nc == 2) return launch_q<2, 0>(a, stream);
  if (nc == 3) return launch_q<3, 0>(a, stream);
  return launch_q<4, 0>(a, stream);
}

SC_API long long sc_attention_dh_bwd_workspace(int B, int H, int Tq) {
  return 3LL * B * H * ((Tq + 63) / 64 * 64) * (long long)sizeof(float);
}

SC_API int sc_attention_dh_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                               void* dV, int B, int H, int Tq, int Tk, int head_dim, long long q_b, long long ldq, long long kv_b,
                               long long ldk, long long o_b, long long ldo, long long dq_b, long long lddq, long long dkv_b, long long lddk,
                               const unsigned char* mask, long long mask_b, int causal, float scale, float drop_p,
                               const unsigned long long* seed, unsigned long long offset, float* workspace, long long ws_bytes,
                               hipStream_t stream) {
  SC_CHECK_ARG(dO && dQ && dK && dV && workspace, "sc_attention_dh_bwd: null pointer");
  SC_CHECK_ARG(ws_bytes >= sc_attention_dh_bwd_workspace(B, H, Tq), "sc_attention_dh_bwd: workspace too small (sc_attention_dh_bwd_workspace)");
  DhArgs a{};
  int rc = fill_args(a, Q, K, V, O, const_cast<float*>(LSE), B, H, Tq, Tk, head_dim, q_b, ldq, kv_b, ldk, o_b, ldo, mask, mask_b, causal, scale,
                     drop_p, seed, offset, "sc_attention_dh_bwd");
  if (rc != SC_OK) return rc;
  SC_CHECK_ARG(lddq % 4 == 0 && dq_b % 4 == 0 && lddk % 4 == 0 && dkv_b % 4 == 0 && (uintptr_t)dQ % 8 == 0 && (uintptr_t)dK % 8 == 0 &&
                   (uintptr_t)dV % 8 == 0 && (uintptr_t)dO % 16 == 0,
               "sc_attention_dh_bwd: gradients must be 8-byte aligned with strides multiple of 4 (dO: 16-byte aligned)");
  a.dO = (const unsigned short*)dO; a.dQ = (unsigned short*)dQ; a.dK = (unsigned short*)dK; a.dV = (unsigned short*)dV;
  a.dq_b = dq_b; a.lddq = lddq; a.dkv_b = dkv_b; a.lddk = lddk;
  a.Tqp = (Tq + 63) / 64 * 64;
  a.delta = workspace;
  a.nl = workspace + (long long)B * H * a.Tqp;
  a.nd = workspace + 2LL * B * H * a.Tqp;
  SC_CHECK_ARG((uintptr_t)workspace % 4 == 0, "sc_attention_dh_bwd: misaligned workspace");
  hipLaunchKernelGGL(attn_dh_delta_kernel, dim3((unsigned)(((long long)B * H * a.Tqp + 7) / 8)), dim3(256), 0, stream, a);
  SC_LAUNCH_CHECK("attn_dh_delta_kernel");
  const int nc = (head_dim + 63) / 64;
  if (nc == 1) { rc = launch_dkv<1>(a, stream); if (rc != SC_OK) return rc; return launch_q<1, 1>(a, stream); }
  if (nc == 2) { rc = launch_dkv<2>(a, stream); if (rc != SC_OK) return rc; return launch_q<2, 1>(a, stream); }
  if (nc == 3) { rc = launch_dkv<3>(a, stream); if (rc != SC_OK) return rc; return launch_q<3, 1>(a, stream); }
  rc = launch_dkv<4>(a, stream);
  if (rc != SC_OK) return rc;
  return launch_q<4, 1>(a, stream);
}
