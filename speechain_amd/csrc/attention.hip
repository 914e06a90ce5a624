// Fused scaled-dot-product attention for gfx950 (bf16, head size 64, up to 256 keys) -- the training-path replacement of
// matmul(q, k^T) * scale -> masked_fill(-inf) -> softmax -> dropout -> matmul(attn, v) at
// speechain/module/transformer/attention.py:88-106,131 (reference path /root/reference/) when the caller does not ask for
// the (B, H, Tq, Tk) probability tensor.  Scores never touch HBM; the backward pass recomputes them from Q, K and the
// saved per-row logsumexp.
//
// Forward: one 512-thread workgroup per (batch, head, 64-query tile).  Q tile, all keys and all values of the head live
// in LDS (padded 144-byte rows); S = scale * Q K^T on the bf16 matrix cores into an f32 LDS image; row softmax with 8 lanes
// per row (mask byte / causal / length, dropout from the device seed cell); P is written back in place as bf16; O = P V with
// V fragments read through the transposing LDS read, the two key halves combined through LDS.
#include "sc_common.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

constexpr int DH = 64;        // head size
constexpr int TKMAX = 256;    // keys held in LDS
constexpr int LDR = 72;       // bf16 row stride of the Q / K / V / dO tiles (64 + 8 pad: 144 B, 16-byte aligned)
constexpr int LDS_S = 264;    // f32 row stride of the score image (its bf16 view has row stride 528)
constexpr int LDP = 2 * LDS_S;

struct AttnArgs {
  const unsigned short* Q;
  const unsigned short* K;
  const unsigned short* V;
  unsigned short* O;
  float* LSE;
  long long q_b, ldq, kv_b, ldk, o_b, ldo;
  const unsigned char* mask;
  long long mask_b, mask_q;
  int causal;
  int B, H, Tq, Tk, ldp;  // ldp: row pitch of the (virtual) probability tensor in the dropout index = Tk rounded up to 8
  float scale, drop_p, drop_keep;
  const unsigned long long* seed;
  unsigned long long offset;
  // backward only
  const unsigned short* dO;
  unsigned short* dQ;
  unsigned short* dK;
  unsigned short* dV;
  long long dq_b, lddq, dkv_b, lddk;
  // key-block sweep of the key-on-the-lane backward (more than 256 keys): this launch owns keys [key0, key0 + 32 NW)
  int key0;
  int dq_mode;   // 0: dQ final (bf16); 1: f32 partial -> dq_ws; 2: dq_ws + partial -> bf16 final; 3: dq_ws += partial
  float* dq_ws;  // (B H, Tq, 64) f32
};

// k-contiguous MFMA 32x32x16 operand fragment: row `row`, elements k0 + 8*(lane>>5) .. +7 of a [rows][ld] bf16 tile
__device__ __forceinline__ bf16x8_t frag_rowmajor(const unsigned short* tile, int ld, int row, int k0, int lane) {
  return *(const bf16x8_t*)(tile + row * ld + k0 + (lane >> 5) * 8);
}
// transposed operand fragment from a [k][col] tile: column col_base + (lane & 31), rows k0 + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_transposed(const unsigned short* tile, int ld, int col_base, int k0, int lane) {
  const int g = lane >> 4, ip = lane & 15, q = ip >> 2, pp = ip & 3;
  const unsigned short* a = tile + (k0 + 8 * (g >> 1) + q) * ld + col_base + 16 * (g & 1) + 4 * pp;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 4 * ld));
  const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// two f32 -> packed bf16 pair in ONE instruction.  Inline asm, so (a) never feed it the result of a transcendental (v_exp, v_rcp ...)
// directly: the wait state such a result needs before its first use is not inserted for asm; (b) it pins the order "select, then
// convert" that the compiler otherwise turns into "convert each, select 16-bit halves, v_perm".
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// rows [r0, r0+nrows) x 64 columns of a strided bf16 matrix -> LDS tile with row stride LDR; rows >= valid are zero
__device__ __forceinline__ void load_rows(unsigned short* tile, const unsigned short* src, long long ld, int r0, int nrows,
                                          int valid, int tid, int nthreads) {
  for (int c = tid; c < nrows * 8; c += nthreads) {
    const int r = c >> 3, ch = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < valid) v = *(const uint4*)(src + (long long)(r0 + r) * ld + ch * 8);
    *(uint4*)(tile + r * LDR + ch * 8) = v;
  }
}

__global__ __launch_bounds__(512, 1) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qs = (unsigned short*)smem_raw;            // [64][72]
  unsigned short* Ks = Qs + 64 * LDR;                         // [256][72]
  unsigned short* Vs = Ks + TKMAX * LDR;                      // [256][72]
  float* Ss = (float*)(Vs + TKMAX * LDR);                     // [64][264] f32 (bf16 view: row stride 528)
  unsigned short* Ps = (unsigned short*)Ss;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int z = blockIdx.y, b = z / p.H, hd = z - b * p.H;
  const int q0 = blockIdx.x * 64;
  const unsigned short* Qg = p.Q + b * p.q_b + hd * DH;
  const unsigned short* Kg = p.K + b * p.kv_b + hd * DH;
  const unsigned short* Vg = p.V + b * p.kv_b + hd * DH;

  load_rows(Qs, Qg, p.ldq, q0, 64, p.Tq, tid, 512);
  load_rows(Ks, Kg, p.ldk, 0, TKMAX, p.Tk, tid, 512);
  load_rows(Vs, Vg, p.ldk, 0, TKMAX, p.Tk, tid, 512);
  __syncthreads();

  // ---- S = scale * Q K^T : wave (qs, kq) owns queries qs*32.., keys kq*64.. (two 32-key tiles) ----
  {
    const int qs = wid & 1, kq = wid >> 1;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = frag_rowmajor(Qs, LDR, qs * 32 + li, ks * 16, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8_t bb = frag_rowmajor(Ks, LDR, kq * 64 + t * 32 + li, ks * 16, lane);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Ss[(qs * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDS_S + kq * 64 + t * 32 + li] = acc[t][r] * p.scale;
  }
  __syncthreads();

  // ---- row softmax: 8 consecutive lanes per query row, 32 keys each; P (with dropout) written back in place as bf16 ----
  {
    const int row = tid >> 3, seg = tid & 7;
    const int qg = q0 + row;
    const unsigned char* mrow = p.mask ? p.mask + b * p.mask_b + (long long)min(qg, p.Tq - 1) * p.mask_q : nullptr;
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int k = seg * 32 + j;
      bool ok = k < p.Tk;
      if (ok && mrow) ok = mrow[k] != 0;
      if (ok && p.causal) ok = k <= qg;
      v[j] = ok ? Ss[row * LDS_S + k] : -INFINITY;
      mx = fmaxf(mx, v[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      v[j] = (v[j] == -INFINITY) ? 0.f : __expf(v[j] - mx);
      s += v[j];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    if (seg == 0 && qg < p.Tq) p.LSE[(long long)z * p.Tq + qg] = s > 0.f ? mx + __logf(s) : -INFINITY;
    const bool drop = p.drop_p > 0.f;
    const unsigned long long seed = drop ? *p.seed : 0ull;
    const unsigned thr = sc_drop_threshold(p.drop_p);
    const unsigned long long idx0 = p.offset + ((unsigned long long)z * p.Tq + qg) * p.ldp + seg * 32;
    // all 32 values of this lane are in registers: the in-place bf16 store below cannot clobber unread f32 scores of the
    // row (its other 7 lanes belong to the same wave and have issued their loads already)
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
      float p0 = v[j] * inv, p1 = v[j + 1] * inv;
      if (drop) {
        const unsigned hsh = sc_rand_pair(seed, (idx0 + j) >> 1);
        p0 = ((hsh & 0xFFFFu) < thr) ? 0.f : p0 * p.drop_keep;
        p1 = ((hsh >> 16) < thr) ? 0.f : p1 * p.drop_keep;
      }
      *(unsigned*)(Ps + row * LDP + seg * 32 + j) = sc_f32_to_bf16(p0) | ((unsigned)sc_f32_to_bf16(p1) << 16);
    }
  }
  __syncthreads();

  // ---- O = P V : wave (qs, dt, kh) owns queries qs*32.., head dims dt*32.., keys kh*128.. ----
  {
    const int qs = wid & 1, dt = (wid >> 1) & 1, kh = wid >> 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k0 = kh * 128 + kk * 16;
      const bf16x8_t a = frag_rowmajor(Ps, LDP, qs * 32 + li, k0, lane);
      const bf16x8_t bb = frag_transposed(Vs, LDR, dt * 32, k0, lane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
    }
    float* red = (float*)Ks + (qs * 2 + dt) * 32 * 33;  // K is dead: 4 x (32 x 33) f32 reduction slabs
    __syncthreads();
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + li] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
      unsigned short* Og = p.O + b * p.o_b + hd * DH + dt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int qg = q0 + qs * 32 + row;
        if (qg < p.Tq) Og[(long long)qg * p.ldo] = sc_f32_to_bf16(acc[r] + red[row * 33 + li]);
      }
    }
  }
}



// ---------------------------------------------------------------------------------------------------------------
// Streaming ("flash") forward for key masks that do not depend on the query (none, (B, 1, Tk), causal flag): any Tk.
//
// One 256-thread workgroup per (batch, head, 128-query tile); wave w owns queries 32 w .. 32 w + 31 and keeps its Q rows
// in registers.  K and V stream through LDS in 64-key tiles (LDS-DMA, three 16 KB stages, one barrier per tile; rows are
// 128 bytes and cannot be padded, so the 16-byte chunk index is XOR-ed with (key & 7) on the source side and in the
// read addresses: conflict-free for the row-major K fragments and the transposing V reads alike).
//
// The score tile is computed TRANSPOSED, S^T = K Q^T (keys x queries): in the 32x32 accumulator layout a lane then holds
// ONE query (column lane & 31) and 16 keys per 32-key block in its registers, so the online softmax (running max / sum,
// rescale) is register arithmetic plus one v_permlane32_swap per tile -- no score image in LDS, no shuffles per row.
// The same registers, packed to bf16, are the B operand of O^T += V^T P~^T (head dims x queries): lane half h supplies
// keys 4h + {0..3, 8..11} (+16, +32, +48) of the tile, and the V fragments are read with the matching key order.
// Dropout indices, the LSE definition (natural log) and the zero output of an all-masked row are those of attn_fwd_kernel,
// so attn_bwd_kernel consumes either forward.
// ---------------------------------------------------------------------------------------------------------------
// -DSC_ATTN_TIMING (instrumented builds only): workgroup 0 records s_memtime at phase boundaries, per wave; read back with
// sc_attention_debug_read (not part of the ABI)
#ifdef SC_ATTN_TIMING
__device__ long long sc_attn_probe[8 * 8 * 8];
#define AT_PROBE(round, point)                                                                                            \
  do {                                                                                                                    \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (round) < 8)                                                   \
      sc_attn_probe[(wid * 8 + (round)) * 8 + (point)] = __builtin_readcyclecounter();                                     \
  } while (0)
#else
#define AT_PROBE(round, point) do {} while (0)
#endif
typedef __attribute__((address_space(1))) const void* at_gptr_t;
typedef __attribute__((address_space(3))) void* at_lptr_t;

// SPLIT (at most 32 queries per (batch, head): the encoder-decoder attention of a decoding position, beams as queries): all four waves
// hold the SAME 32 queries and take every fourth key tile each -- wave w stages tiles w, w + 4, ... into a region of its own (no barrier in
// the sweep) -- and wave 0 merges the four partial (maximum, sum, O) triples at the end.  The unsplit form gives such a launch to one wave:
// four dependent key tiles, ~12 us for 0.26 GFLOP.  No causal mask, no dropout in this form (checked on the host).
template <bool DROP, bool SPLIT = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_stream_kernel(AttnArgs p) {
  constexpr int KT = 64, STAGES = 3, TILE = KT * DH, STAGE_ELEMS = 2 * TILE;  // K tile + V tile per stage (16 KB)
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  // workgroups are dealt to the 8 XCDs round-robin by linear id: the query tiles of one (batch, head) are mapped to ids 8 apart, i.e.
  // to ONE XCD and to consecutive slots of it, so that the head's K and V come out of that XCD's L2 from the second tile on
  int z = blockIdx.y, qt = blockIdx.x;
  {
    const int nq = gridDim.x, nz = gridDim.y;
    if (nq > 1 && (nz & 7) == 0) {
      const int lin = blockIdx.x + nq * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
      z = (slot / nq) * 8 + xcd;
      qt = slot % nq;
    }
  }
  const int b = z / p.H, hd = z - b * p.H;
  const int qg = SPLIT ? qt * 32 + li : qt * 128 + wid * 32 + li;  // this lane's query
  const int qc = min(qg, p.Tq - 1);
  const unsigned short* Kg = p.K + b * p.kv_b + hd * DH;
  const unsigned short* Vg = p.V + b * p.kv_b + hd * DH;
  const int nt = (p.Tk + KT - 1) / KT;
  AT_PROBE(7, 0);

  // DMA sources: a granule is 8 key rows x 128 B; wave w stages granules 2w, 2w+1 of K and of V.  Lane l fills LDS slot
  // (row l >> 3, chunk l & 7) from source chunk (l & 7) ^ (row & 7); rows beyond Tk are clamped (masked out below).
  const int grow = lane >> 3, gch = (lane & 7) ^ (grow & 7);
#define AT_ISSUE(T)                                                                                                      \
  do {                                                                                                                   \
    unsigned short* st_ = dsm + (SPLIT ? wid : (T) % STAGES) * STAGE_ELEMS; /* (SPLIT: the wave's own region, all 8 granules) */ \
    _Pragma("unroll") for (int i_ = 0; i_ < (SPLIT ? 8 : 2); ++i_) {                                                     \
      const int g_ = SPLIT ? i_ : 2 * wid + i_;                                                                          \
      const long long src_ = (long long)min((T) * KT + g_ * 8 + grow, p.Tk - 1) * p.ldk + gch * 8;                       \
      __builtin_amdgcn_global_load_lds((at_gptr_t)(Kg + src_), (at_lptr_t)(st_ + g_ * 512), 16, 0, 0);                   \
      __builtin_amdgcn_global_load_lds((at_gptr_t)(Vg + src_), (at_lptr_t)(st_ + TILE + g_ * 512), 16, 0, 0);            \
    }                                                                                                                    \
  } while (0)
  // Q^T operand of this lane's query: head dims 16 ks + 8 h .. + 7
  bf16x8_t qf[4];
  {
    const unsigned short* qrow = p.Q + b * p.q_b + hd * DH + (long long)qc * p.ldq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qrow + ks * 16 + h * 8);
  }
  const unsigned char* mrow = p.mask ? p.mask + b * p.mask_b : nullptr;
  // key validity, lane i answers for key 64 t + i; the byte of tile t + 1 is fetched before the DMA of tile t + 2 is issued,
  // so that waiting for it never drains the prefetched tiles (vmcnt counts in order)
  auto key_ok = [&](int t_) {
    const int kk = t_ * KT + lane;
    return kk < p.Tk && (mrow == nullptr || mrow[min(kk, p.Tk - 1)] != 0);
  };
  const float c1 = p.scale * 1.4426950408889634f;  // running maximum and sums are kept in log2 units: p = exp2(c1 s - m)
  const unsigned long long seed = DROP ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
  // pair index of this lane's (query, key 64 t + kb + 4 h), kb even: pairq + 32 t + kb / 2
  const unsigned long long pairq = ((p.offset + ((unsigned long long)z * p.Tq + qc) * p.ldp) >> 1) + (unsigned)(2 * h);
  const int qwave0 = SPLIT ? qt * 32 : qt * 128 + wid * 32;  // first query of the wave

  f32x16 o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this lane half's share of the row sum

  bool kok = key_ok(SPLIT ? wid : 0);
  if (!SPLIT) {
    if (nt > 0) AT_ISSUE(0);
    if (nt > 1) AT_ISSUE(1);
  }
  for (int t = SPLIT ? wid : 0; t < nt; t += SPLIT ? 4 : 1) {
    AT_PROBE(t, 0);
    bool kok_next;
    if (SPLIT) {  // this wave's own tile: request, wait, no barrier (nobody else touches the region)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads of the wave's previous tile are done)
      AT_ISSUE(t);
      kok_next = (t + 4 < nt) ? key_ok(t + 4) : false;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      AT_PROBE(t, 1);
      kok_next = (t + 1 < nt) ? key_ok(t + 1) : false;
      if (t + 2 < nt) AT_ISSUE(t + 2);
    }
    const unsigned short* ks_ = dsm + (SPLIT ? wid : t % STAGES) * STAGE_ELEMS;
    const unsigned short* vs_ = ks_ + TILE;
    // key validity of the tile as a 64-bit lane mask
    const unsigned long long bits = __ballot(kok);
    kok = kok_next;

    // ---- S^T = K Q^T: two 32-key blocks x 32 queries ----
    f32x16 s2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s2[j][r] = 0.f;
      const int krow = 32 * j + li;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t a = *(const bf16x8_t*)(ks_ + krow * DH + (((2 * ks + h) ^ (krow & 7)) << 3));
        s2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s2[j], 0, 0, 0);
      }
    }
    AT_PROBE(t, 2);
    // ---- masks: only for a tile with an invisible key or one the causal diagonal runs through (wave-uniform test) ----
    if (bits != ~0ull || (p.causal && t * KT + KT - 1 > qwave0)) {
      const int dq = p.causal ? qg - t * KT - 4 * h : KT;  // register (j, r) is visible up to key index 32 j + (r & 3) + 8 (r >> 2) <= dq
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
    // ---- online softmax for this lane's query (the maximum is taken over the raw scores: c1 > 0) ----
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s2[j][r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
      mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // both halves: max over all 64 keys
    }
    const float m_new = fmaxf(m_run, mloc * c1);
    const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
    const float nmsub = (m_new == -INFINITY) ? 0.f : -m_new;
    float lsum = 0.f;
    bf16x8_t pf[4];  // P~^T operand: k-step ks = block ks >> 1, registers 8 (ks & 1) .. + 7
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned pw[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        // exp2(-inf) = 0 for masked keys.  The factor 1 / (1 - p_drop) of the kept ones is applied to the row at the end.
        float p0 = __builtin_amdgcn_exp2f(fmaf(s2[j][r], c1, nmsub)), p1 = __builtin_amdgcn_exp2f(fmaf(s2[j][r + 1], c1, nmsub));
        lsum += p0 + p1;
        if (DROP) {
          const int kb = 32 * j + (r & 3) + 8 * (r >> 2);  // (+ 4 h) even key: (kb, kb + 1) share one hash word
          const unsigned hsh = sc_rand_pair(seed, pairq + (unsigned)(t * (KT / 2) + kb / 2));
          p0 = ((hsh & 0xFFFFu) < thr) ? 0.f : p0;
          p1 = ((hsh >> 16) < thr) ? 0.f : p1;
          pw[r >> 1] = cvt_pk_bf16(p0, p1);
        } else {
          pw[r >> 1] = (unsigned)sc_f32_to_bf16(p0) | ((unsigned)sc_f32_to_bf16(p1) << 16);
        }
      }
      pf[2 * j] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[0], pw[1], pw[2], pw[3]));
      pf[2 * j + 1] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[4], pw[5], pw[6], pw[7]));
    }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
    if (__any(alpha != 1.f)) {  // (the running maximum of most rows settles in the first tiles)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
    }
    AT_PROBE(t, 3);
    // ---- O^T += V^T P~^T: two 32-head-dim blocks; k-step ks covers keys 32 (ks >> 1) + 16 (ks & 1) + 4 h + {0..3, 8..11} ----
    {
      const int g = lane >> 4, ip = lane & 15, q = ip >> 2, pp = ip & 3;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kbase = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * (g >> 1) + q;  // this lane's key row of the `lo` read; `hi` is + 8
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = 32 * j + 16 * (g & 1) + 4 * pp;
          const unsigned short* a0 = vs_ + kbase * DH + (((col >> 3) ^ (kbase & 7)) << 3) + (col & 7);
          const unsigned short* a1 = vs_ + (kbase + 8) * DH + (((col >> 3) ^ ((kbase + 8) & 7)) << 3) + (col & 7);
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a1);
          const s16x8_t v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, v8), pf[ks], o[j], 0, 0, 0);
        }
      }
    }
  }
#undef AT_ISSUE
  AT_PROBE(7, 1);

  if (SPLIT) {
    // ---- merge: every wave leaves (m, l, O) of its key tiles in its own region, lane by lane (the four waves hold the same queries in the
    //      same registers); wave 0 rescales them to the common maximum and sums ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* part = (float*)(dsm + wid * STAGE_ELEMS);  // [34][64]: m, l, o[0][0..15], o[1][0..15], lane-contiguous (conflict-free)
    part[0 * 64 + lane] = m_run;
    part[1 * 64 + lane] = l_run;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(2 + 16 * j + r) * 64 + lane] = o[j][r];
    __syncthreads();
    if (wid != 0) return;
    float mw[4], lw[4];
    float mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* pw = (const float*)(dsm + w * STAGE_ELEMS);
      mw[w] = pw[0 * 64 + lane];
      lw[w] = pw[1 * 64 + lane];
      mstar = fmaxf(mstar, mw[w]);
    }
    float lsum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (mw[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[w] - mstar);
      const float* pw = (const float*)(dsm + w * STAGE_ELEMS);
      lsum += lw[w] * f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = fmaf(pw[(2 + 16 * j + r) * 64 + lane], f, o[j][r]);
    }
    m_run = mstar;
    l_run = lsum;
  }

  // ---- normalise and store: lane (query, half h) holds head dims 32 j + 8 (r >> 2) + 4 h + (r & 3) ----
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const float inv = l > 0.f ? p.drop_keep / l : 0.f;
    if (qg < p.Tq) {
      if (h == 0) p.LSE[(long long)z * p.Tq + qg] = l > 0.f ? (m_run + log2f(l)) * 0.6931471805599453f : -INFINITY;
      unsigned short* orow = p.O + b * p.o_b + hd * DH + (long long)qg * p.ldo;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const unsigned w0 = sc_f32_to_bf16(o[j][4 * r4] * inv) | ((unsigned)sc_f32_to_bf16(o[j][4 * r4 + 1] * inv) << 16);
          const unsigned w1 = sc_f32_to_bf16(o[j][4 * r4 + 2] * inv) | ((unsigned)sc_f32_to_bf16(o[j][4 * r4 + 3] * inv) << 16);
          *(uint2*)(orow + 32 * j + 8 * r4 + 4 * h) = make_uint2(w0, w1);
        }
    }
  }
  AT_PROBE(7, 2);
}

// Backward: one 512-thread workgroup per (batch, head) sweeping its 64-query tiles.  K and V of the head stay in LDS; per
// tile S = Q K^T and dP = dO V^T are formed side by side in registers, P = exp(scale*S - lse) (masked), the dropout mask is
// regenerated, dS = P * (dP - delta) * scale with delta = rowsum(dO * O); P~ (dropped P) and dS go to LDS as bf16 and feed
// dV += P~^T dO, dK += dS^T Q (accumulated in registers across tiles: every wave owns 32 keys) and dQ = dS K.
constexpr int LDPB = 264;  // bf16 row stride of the P~ / dS tiles

__global__ __launch_bounds__(512, 1) void attn_bwd_kernel(AttnArgs p, const unsigned short* Og) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qs = (unsigned short*)smem_raw;   // [64][72]
  unsigned short* dOs = Qs + 64 * LDR;              // [64][72]
  unsigned short* Ks = dOs + 64 * LDR;              // [256][72]
  unsigned short* Vs = Ks + TKMAX * LDR;            // [256][72]
  unsigned short* Pd = Vs + TKMAX * LDR;            // [64][264]
  unsigned short* dSs = Pd + 64 * LDPB;             // [64][264]
  float* rowc = (float*)(dSs + 64 * LDPB);          // [2][64]: lse, delta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int z = blockIdx.x, b = z / p.H, hd = z - b * p.H;
  const unsigned short* Qg = p.Q + b * p.q_b + hd * DH;
  const unsigned short* Kg = p.K + b * p.kv_b + hd * DH;
  const unsigned short* Vg = p.V + b * p.kv_b + hd * DH;
  const unsigned short* dOg = p.dO + b * p.o_b + hd * DH;
  const unsigned short* Ogh = Og + b * p.o_b + hd * DH;

  load_rows(Ks, Kg, p.ldk, 0, TKMAX, p.Tk, tid, 512);
  load_rows(Vs, Vg, p.ldk, 0, TKMAX, p.Tk, tid, 512);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;

  const bool drop = p.drop_p > 0.f;
  const unsigned long long seed = drop ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);

  for (int q0 = 0; q0 < p.Tq; q0 += 64) {
    // K / V fragments are loop-invariant, and hoisting them out of the tile loop (64+ VGPRs) is what pushes this kernel
    // into scratch: launder the tile bases once per iteration so the reads stay inside it
    const unsigned short* Kt = Ks;
    const unsigned short* Vt = Vs;
    asm volatile("" : "+v"(Kt), "+v"(Vt));
    // ---- stage Q, dO; delta = rowsum(dO * O); lse ----
    load_rows(Qs, Qg, p.ldq, q0, 64, p.Tq, tid, 512);
    {
      const int r = tid >> 3, ch = tid & 7;
      uint4 vd = make_uint4(0, 0, 0, 0), vo = make_uint4(0, 0, 0, 0);
      if (q0 + r < p.Tq) {
        vd = *(const uint4*)(dOg + (long long)(q0 + r) * p.ldo + ch * 8);
        vo = *(const uint4*)(Ogh + (long long)(q0 + r) * p.ldo + ch * 8);
      }
      *(uint4*)(dOs + r * LDR + ch * 8) = vd;
      const unsigned wd[4] = {vd.x, vd.y, vd.z, vd.w}, wo[4] = {vo.x, vo.y, vo.z, vo.w};
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dot += __uint_as_float(wd[j] << 16) * __uint_as_float(wo[j] << 16);
        dot += __uint_as_float(wd[j] & 0xFFFF0000u) * __uint_as_float(wo[j] & 0xFFFF0000u);
      }
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      if (ch == 0) {
        rowc[64 + r] = dot;
        rowc[r] = (q0 + r < p.Tq) ? p.LSE[(long long)z * p.Tq + q0 + r] : -INFINITY;
      }
    }
    __syncthreads();

    // ---- S and dP tiles of wave (qs, kq); P, P~, dS ----
    {
      const int qs = wid & 1, kq = wid >> 1;
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {  // one 32-key tile at a time keeps the register footprint below the spill point
        f32x16 as, ap;
#pragma unroll
        for (int r = 0; r < 16; ++r) as[r] = ap[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8_t aq = frag_rowmajor(Qs, LDR, qs * 32 + li, ks * 16, lane);
          const bf16x8_t ad = frag_rowmajor(dOs, LDR, qs * 32 + li, ks * 16, lane);
          const bf16x8_t bk = frag_rowmajor(Kt, LDR, kq * 64 + t * 32 + li, ks * 16, lane);
          const bf16x8_t bv = frag_rowmajor(Vt, LDR, kq * 64 + t * 32 + li, ks * 16, lane);
          as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bk, as, 0, 0, 0);
          ap = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bv, ap, 0, 0, 0);
        }
        const int key = kq * 64 + t * 32 + li;
        const bool kvalid = key < p.Tk && (p.mask == nullptr || p.mask[b * p.mask_b + key] != 0);  // (B, 1, Tk) masks only
        const unsigned long long didx = p.offset + ((unsigned long long)z * p.Tq + q0 + qs * 32) * p.ldp + key;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = qs * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int qg = q0 + ql;
          bool ok = kvalid && qg < p.Tq;
          if (ok && p.causal) ok = key <= qg;
          const float lse = rowc[ql];
          const float pr = (ok && lse != -INFINITY) ? __expf(as[r] * p.scale - lse) : 0.f;
          float pd = pr, dp = ap[r];
          if (drop) {
            const bool dropped = sc_dropped(seed, didx + (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * h) * p.ldp), thr);
            pd = dropped ? 0.f : pr * p.drop_keep;
            dp = dropped ? 0.f : dp * p.drop_keep;
          }
          const float ds = pr * (dp - rowc[64 + ql]) * p.scale;
          Pd[ql * LDPB + key] = sc_f32_to_bf16(pd);
          dSs[ql * LDPB + key] = sc_f32_to_bf16(ds);
          __builtin_amdgcn_sched_barrier(0);  // keep the 16 element bodies sequential: short live ranges, no spills
        }
      }
    }
    __syncthreads();

    // ---- dV += P~^T dO, dK += dS^T Q for this wave's 32 keys; partial dQ of unit (qs, dt, kh) ----
    {
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {  // contraction over the 64 queries of the tile
        const bf16x8_t apd = frag_transposed(Pd, LDPB, wid * 32, ks * 16, lane);
        const bf16x8_t ads = frag_transposed(dSs, LDPB, wid * 32, ks * 16, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8_t bdo = frag_transposed(dOs, LDR, t * 32, ks * 16, lane);
          const bf16x8_t bq = frag_transposed(Qs, LDR, t * 32, ks * 16, lane);
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apd, bdo, dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ads, bq, dk[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int qs = wid & 1, dt = (wid >> 1) & 1, kh = wid >> 2;
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll 2
      for (int kk = 0; kk < 8; ++kk) {
        const int k0 = kh * 128 + kk * 16;
        const bf16x8_t a = frag_rowmajor(dSs, LDPB, qs * 32 + li, k0, lane);
        const bf16x8_t bb = frag_transposed(Kt, LDR, dt * 32, k0, lane);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, dq, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();  // every wave is done reading P~: its storage becomes the dQ reduction slabs
      float* red = (float*)Pd + (qs * 2 + dt) * 32 * 33;
      if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + li] = dq[r];
      }
      __syncthreads();
      if (kh == 0) {
        unsigned short* dQg = p.dQ + b * p.dq_b + hd * DH + dt * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const int qg = q0 + qs * 32 + row;
          if (qg < p.Tq) dQg[(long long)qg * p.lddq] = sc_f32_to_bf16(dq[r] + red[row * 33 + li]);
        }
      }
    }
    __syncthreads();  // slabs / Q / dO tiles are rewritten by the next tile
  }

  // ---- dK, dV of this wave's keys ----
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    unsigned short* dKg = p.dK + b * p.dkv_b + hd * DH + t * 32 + li;
    unsigned short* dVg = p.dV + b * p.dkv_b + hd * DH + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (key < p.Tk) {
        dKg[(long long)key * p.lddk] = sc_f32_to_bf16(dk[t][r]);
        dVg[(long long)key * p.lddk] = sc_f32_to_bf16(dv[t][r]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward, "key on the lane": one workgroup of NW waves per (batch, head); wave w owns keys 32 w .. 32 w + 31 and keeps their
// K and V rows as MFMA B fragments and their dK^T, dV^T (head dims x keys) as accumulators for the whole kernel.  NW = 4 (one
// wave per SIMD) for up to 128 keys, NW = 8 (two waves per SIMD: one wave's exp / hash / select work runs beside the other's
// matrix products) for up to 256.  The workgroup sweeps the queries 64 at a time (two 32-query blocks):
//   S' = Q K^T - lse / scale and dP' = dO V^T - delta (1 - p_drop): the row constants are the INITIAL accumulators, so that
//        p = exp2(c1 S') needs no subtraction and no row maximum; a -inf constant switches a row (query >= Tq, all-masked
//        row) or a key (padding mask, causal) off;
//   the accumulators hold one key per lane and 16 queries per 32 x 32 block, which is the B-operand layout of the next two
//        products: dV^T += dO^T P~ and dK^T += Q^T dS take P~ and dS straight from registers (the A operands are transposing
//        reads of the one Q / dO image in LDS, in the matching query order);
//   only dS crosses LDS, once, as a [key][query] image (8-byte stores of four consecutive queries) for
//        dQ^T = K^T dS^T (head dims x queries): wave w takes (head-dim block w & 1, query block (w >> 1) & 1) and, with eight
//        waves, the key half w >> 2; the upper half's partial sum crosses LDS and is added by its partner after the next
//        barrier the loop has anyway; 8-byte stores per register quad.
// Dropout: the mask word of a key pair is hashed once, by the even lane for the even query row and by the odd lane for
// the odd row, exchanged with one DPP move and put in (row, row + 1) order by one v_perm.  The factors scale and
// 1 / (1 - p_drop) are applied to the sums at the end (dS' = p sel, dS = keep scale dS'; P~' = p or 0, dV = keep sum).
// Q, dO and O of the next 64 queries are fetched into registers before the current ones are used (two LDS stages): two
// barriers per 64 queries.  Every global load of the prologue is in flight before the first one is waited for.
// ---------------------------------------------------------------------------------------------------------------
// LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier): the loads and stores in flight stay in flight
#define KL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define KL_PROBE(round, point) AT_PROBE(round, point)
template <int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_kl_kernel(AttnArgs p, const unsigned short* Og) {
  constexpr int NK = 32 * NW;    // keys the workgroup covers
  constexpr int NT = 64 * NW;    // threads
  constexpr int SPT = 512 / NT;  // 16-byte chunks of a 64 x 64 stage per thread and matrix
  constexpr int TPR = 8 / SPT;   // threads per stage row
  constexpr int KPT = NK * 8 / NT;  // 16-byte chunks of the K image per thread (4)
  constexpr int NKS = NK / 16 / (NW / 4);  // k-steps of one wave's dQ product (8)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Kimg = (unsigned short*)smem_raw;  // [NK][72]
  unsigned short* Timg = Kimg + NK * LDR;            // [NK][72]   dS'^T
  unsigned short* QT = Timg + NK * LDR;              // [2][64][72]
  unsigned short* DT = QT + 2 * 64 * LDR;            // [2][64][72]
  float* rowc = (float*)(DT + 2 * 64 * LDR);         // [2][2][64]: -lse / scale, -delta (1 - p_drop)
  float* red = rowc + 256;                           // [4][4][64][4] partial dQ^T of the upper key half (NW = 8)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int z = blockIdx.x, b = z / p.H, hd = z - b * p.H;
  const unsigned short* Qg = p.Q + b * p.q_b + hd * DH;
  const unsigned short* Kg = p.K + b * p.kv_b + hd * DH;
  const unsigned short* Vg = p.V + b * p.kv_b + hd * DH;
  const unsigned short* dOg = p.dO + b * p.o_b + hd * DH;
  const unsigned short* Ogh = Og + b * p.o_b + hd * DH;
  const int wkey0 = wid * 32;
  const bool wave_on = p.key0 + wkey0 < p.Tk;  // a wave without keys only takes part in the dQ product
  KL_PROBE(7, 0);
  const float inv_scale = 1.f / p.scale, c1 = p.scale * 1.4426950408889634f, unkeep = 1.f - p.drop_p;

  // ---- this thread's share of a 64-query stage: row tid / TPR, 16-byte chunks (tid % TPR) + TPR i ----
  const int sr = tid / TPR, sc = tid % TPR;
  uint4 rq[SPT], rd[SPT], ro[SPT];
  float rlse;
  auto fetch = [&](int q0) {  // loads only: their latency is covered by the round in between
    const int q = q0 + sr;
#pragma unroll
    for (int i = 0; i < SPT; ++i) rq[i] = rd[i] = ro[i] = make_uint4(0, 0, 0, 0);
    rlse = -INFINITY;
    if (q < p.Tq) {
#pragma unroll
      for (int i = 0; i < SPT; ++i) {
        rq[i] = *(const uint4*)(Qg + (long long)q * p.ldq + (sc + TPR * i) * 8);
        rd[i] = *(const uint4*)(dOg + (long long)q * p.ldo + (sc + TPR * i) * 8);
        ro[i] = *(const uint4*)(Ogh + (long long)q * p.ldo + (sc + TPR * i) * 8);
      }
      rlse = p.LSE[(long long)z * p.Tq + q];
    }
  };
  auto stash = [&](int buf) {
    float dot = 0.f;  // delta = rowsum(dO * O): this thread's elements; the row's other threads are the neighbouring lanes
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      const unsigned wd[4] = {rd[i].x, rd[i].y, rd[i].z, rd[i].w}, wo[4] = {ro[i].x, ro[i].y, ro[i].z, ro[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dot += __uint_as_float(wd[j] << 16) * __uint_as_float(wo[j] << 16);
        dot += __uint_as_float(wd[j] & 0xFFFF0000u) * __uint_as_float(wo[j] & 0xFFFF0000u);
      }
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    if (TPR == 8) dot += __shfl_xor(dot, 4, 64);
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      *(uint4*)(QT + (buf * 64 + sr) * LDR + (sc + TPR * i) * 8) = rq[i];
      *(uint4*)(DT + (buf * 64 + sr) * LDR + (sc + TPR * i) * 8) = rd[i];
    }
    // (computed by every thread, so that the lse load is waited for on every path: left pending in most threads, it makes the
    // compiler drain vmcnt -- the dQ stores included -- before the next fetch may overwrite the register)
    const float nl = (rlse == -INFINITY) ? -INFINITY : -rlse * inv_scale;
    asm volatile("" ::"v"(nl));
    if (sc == 0) {
      rowc[buf * 128 + sr] = nl;
      rowc[buf * 128 + 64 + sr] = -dot * unkeep;
    }
  };

  // ---- prologue: every load first (stage 0, the K image, this wave's V fragments, its mask bytes), then the LDS writes ----
  fetch(0);
  uint4 kreg[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int c = tid + i * NT, r = c >> 3, ch = c & 7;
    kreg[i] = make_uint4(0, 0, 0, 0);
    if (p.key0 + r < p.Tk) kreg[i] = *(const uint4*)(Kg + (long long)(p.key0 + r) * p.ldk + ch * 8);
  }
  bf16x8_t kf[4], vf[4];
  const int key = p.key0 + wkey0 + li;  // (absolute; the LDS images are indexed by the key inside this launch's block, lkey)
  const int lkey = wkey0 + li;
  const int kc = min(key, p.Tk - 1);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) vf[ks] = *(const bf16x8_t*)(Vg + (long long)kc * p.ldk + ks * 16 + h * 8);
  const bool kvalid = key < p.Tk && (p.mask == nullptr || p.mask[b * p.mask_b + kc] != 0);
  for (int c = tid; c < NK * 9; c += NT) ((uint4*)Timg)[c] = make_uint4(0, 0, 0, 0);  // keys of idle waves stay zero rows of dS'^T
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int c = tid + i * NT, r = c >> 3, ch = c & 7;
    *(uint4*)(Kimg + r * LDR + ch * 8) = kreg[i];
  }
  stash(0);
  const bool need_poison = !__all(kvalid);
  const unsigned long long seed = DROP ? *p.seed : 0ull;
  const unsigned thr = sc_drop_threshold(p.drop_p);
  const int odd = li & 1;
  // the word of (row r, row r + 1) in that order from this lane's hash x and the neighbour's y: even lanes take the low halves
  // (x, y), odd lanes the high halves (y, x)
  const unsigned wsel = odd ? 0x07060302u : 0x01000504u;
  // pair index of (query row, key) = rowbase + lane term; the lane term carries the key pair, the odd lanes' +1 row and the
  // upper lane half's +4 rows
  const unsigned lterm = (unsigned)(key >> 1) + (unsigned)odd * (unsigned)(p.ldp >> 1) + (unsigned)h * (unsigned)(2 * p.ldp);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

  // nothing of the prologue may be left in flight: the compiler would wait for it INSIDE the loop, with counts that, from the second
  // round on, drain the prefetched rows instead
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  KL_BARRIER();  // K image, stage 0
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_rowmajor(Kimg, LDR, lkey, ks * 16, lane);

  const int g = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
  const int ud = wid & 1, uq = (wid >> 1) & 1, ukh = wid >> 2;  // dQ unit of this wave: head-dim block, query block, key half
  f32x16 dq_prev;  // (NW = 8, lower key half) the previous round's partial sum, completed after the next barrier
#pragma unroll
  for (int r = 0; r < 16; ++r) dq_prev[r] = 0.f;
  auto store_dq = [&](const f32x16& acc, int q0) {
    const int q = q0 + uq * 32 + li;
    if (q < p.Tq) {
      const float f = p.scale * p.drop_keep;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[r] * f;
      float* ws = p.dq_ws + ((long long)z * p.Tq + q) * DH + ud * 32 + 4 * h;  // (only touched in modes 1-3)
      if (p.dq_mode >= 2) {  // this launch continues the sum of the earlier key blocks
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = *(const float4*)(ws + 8 * j);
          v[4 * j] += t.x; v[4 * j + 1] += t.y; v[4 * j + 2] += t.z; v[4 * j + 3] += t.w;
        }
      }
      if (p.dq_mode == 1 || p.dq_mode == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4*)(ws + 8 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
      } else {
        unsigned short* dQg = p.dQ + b * p.dq_b + hd * DH + (long long)q * p.lddq + ud * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned w0 = sc_f32_to_bf16(v[4 * j]) | ((unsigned)sc_f32_to_bf16(v[4 * j + 1]) << 16);
          const unsigned w1 = sc_f32_to_bf16(v[4 * j + 2]) | ((unsigned)sc_f32_to_bf16(v[4 * j + 3]) << 16);
          *(uint2*)(dQg + 8 * j) = make_uint2(w0, w1);
        }
      }
    }
  };
  auto finish_prev = [&](int q0_prev) {  // lower key half: add the partner's partial sum of the previous round and store
    float4* rp = (float4*)red + (ud * 2 + uq) * 256 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = rp[j * 64];
      dq_prev[4 * j] += t.x; dq_prev[4 * j + 1] += t.y; dq_prev[4 * j + 2] += t.z; dq_prev[4 * j + 3] += t.w;
    }
    store_dq(dq_prev, q0_prev);
  };

  int buf = 0;
  KL_PROBE(0, 7);
  for (int q0 = 0; q0 < p.Tq; q0 += 64, buf ^= 1) {
    KL_PROBE(q0 >> 6, 0);
    fetch(q0 + 64);  // unconditionally (rows beyond Tq are zeros without a load), and stashed unconditionally below: a fetch that is
                     // consumed on only some paths makes the compiler drain vmcnt at the top of the loop
    if (q0 > 0) KL_BARRIER();  // stage `buf` is visible; every wave is done with the dS' image and the partial sums of the last round
    KL_PROBE(q0 >> 6, 1);
    if (NW == 8 && q0 > 0 && ukh == 0) finish_prev(q0 - 64);
    const unsigned short* Qs = QT + buf * 64 * LDR;
    const unsigned short* Ds = DT + buf * 64 * LDR;
    const float* nlc = rowc + buf * 128;

    if (wave_on) {
#pragma unroll 1
      for (int qb = 0; qb < 2; ++qb) {  // (a loop, not two copies: register pressure)
        f32x16 s, dp;
        float ndk[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 a = *(const float4*)(nlc + qb * 32 + 8 * j + 4 * h);
          const float4 d4 = *(const float4*)(nlc + 64 + qb * 32 + 8 * j + 4 * h);
          const float av[4] = {a.x, a.y, a.z, a.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ndk[4 * j + i] = dvv[i];
            s[4 * j + i] = av[i];
            dp[4 * j + i] = dvv[i];
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
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8_t aq = frag_rowmajor(Qs, LDR, qb * 32 + li, ks * 16, lane);
          const bf16x8_t ad = frag_rowmajor(Ds, LDR, qb * 32 + li, ks * 16, lane);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp, 0, 0, 0);
        }
        // ---- P~', dS' as the B operands of the two key-side products ----
        unsigned pw[8], dw[8];  // packed pairs (rows r, r + 1) of P~' and dS'
        const unsigned long long rowbase0 = (p.offset + ((unsigned long long)z * p.Tq + q0 + qb * 32) * p.ldp) >> 1;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[r] * c1), p1 = __builtin_amdgcn_exp2f(s[r + 1] * c1);
          float pd0 = p0, pd1 = p1, sel0 = dp[r], sel1 = dp[r + 1];
          if (DROP) {
            const unsigned long long rowbase = rowbase0 + (unsigned long long)(((r & 3) + 8 * (r >> 2)) * (p.ldp >> 1));
            const unsigned hs = sc_rand_pair(seed, rowbase + lterm);
            const unsigned gs = (unsigned)__builtin_amdgcn_update_dpp((int)hs, (int)hs, 0xB1, 0xF, 0xF, false);  // the neighbour lane's word
            const unsigned w = __builtin_amdgcn_perm(hs, gs, wsel);
            const bool d0 = (w & 0xFFFFu) < thr, d1 = (w >> 16) < thr;
            pd0 = d0 ? 0.f : p0;
            pd1 = d1 ? 0.f : p1;
            sel0 = d0 ? ndk[r] : sel0;
            sel1 = d1 ? ndk[r + 1] : sel1;
          }
          // (asm after the selects: left to itself the compiler converts first and selects the 16-bit halves afterwards.  Never asm
          // on the v_exp result itself: the one wait state a transcendental's consumer needs is not inserted for inline asm)
          if (DROP) pw[r >> 1] = cvt_pk_bf16(pd0, pd1);
          else pw[r >> 1] = (unsigned)sc_f32_to_bf16(pd0) | ((unsigned)sc_f32_to_bf16(pd1) << 16);
          dw[r >> 1] = cvt_pk_bf16(p0 * sel0, p1 * sel1);
        }
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          pf[ss] = __builtin_bit_cast(bf16x8_t, make_uint4(pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]));
          dsf[ss] = __builtin_bit_cast(bf16x8_t, make_uint4(dw[4 * ss], dw[4 * ss + 1], dw[4 * ss + 2], dw[4 * ss + 3]));
        }
        // ---- dV^T += dO^T P~', dK^T += Q^T dS': k-step ss covers queries 32 qb + 16 ss + 4 h + {0..3, 8..11} ----
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int rbase = qb * 32 + 16 * ss + 4 * (g >> 1) + tq;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const int col = d * 32 + 16 * (g & 1) + 4 * tp;
            const s16x4_t dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Ds + rbase * LDR + col));
            const s16x4_t dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Ds + (rbase + 8) * LDR + col));
            const s16x4_t qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Qs + rbase * LDR + col));
            const s16x4_t qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Qs + (rbase + 8) * LDR + col));
            const bf16x8_t ado = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7));
            const bf16x8_t aqt = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7));
            dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ado, pf[ss], dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aqt, dsf[ss], dk[d], 0, 0, 0);
          }
        }
        // ---- dS'^T image: registers 4 j .. 4 j + 3 are queries 32 qb + 8 j + 4 h + 0..3 of this lane's key ----
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const uint4 w = __builtin_bit_cast(uint4, dsf[ss]);
          unsigned short* trow = Timg + lkey * LDR + qb * 32 + 16 * ss + 4 * h;
          *(uint2*)(trow) = make_uint2(w.x, w.y);
          *(uint2*)(trow + 8) = make_uint2(w.z, w.w);
        }
      }
    }
    KL_PROBE(q0 >> 6, 2);
    KL_BARRIER();  // the dS' image of this round is complete
    KL_PROBE(q0 >> 6, 3);

    // ---- dQ^T = K^T dS'^T for (head-dim block, query block, key half) of this wave: two chains of NKS / 2 products ----
    {
      f32x16 dq0, dq1;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq0[r] = dq1[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NKS; kk += 2) {
        const int k0 = (ukh * NKS + kk) * 16;
        const bf16x8_t a0 = frag_transposed(Kimg, LDR, ud * 32, k0, lane);
        const bf16x8_t b0 = frag_transposed(Timg, LDR, uq * 32, k0, lane);
        const bf16x8_t a1 = frag_transposed(Kimg, LDR, ud * 32, k0 + 16, lane);
        const bf16x8_t b1 = frag_transposed(Timg, LDR, uq * 32, k0 + 16, lane);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dq1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dq0[r] += dq1[r];
      KL_PROBE(q0 >> 6, 4);
      stash(buf ^ 1);  // before the dQ stores: waits for the fetched rows then never include those stores
      KL_PROBE(q0 >> 6, 5);
      if (NW == 4) {
        store_dq(dq0, q0);
      } else if (ukh == 1) {
        float4* rp = (float4*)red + (ud * 2 + uq) * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) rp[j * 64] = make_float4(dq0[4 * j], dq0[4 * j + 1], dq0[4 * j + 2], dq0[4 * j + 3]);
      } else {
        dq_prev = dq0;
      }
    }
  }
  KL_BARRIER();  // every wave is done with the dS' image (it stages dK / dV below); the last round's partial sums are complete
  if (NW == 8 && ukh == 0) finish_prev((p.Tq - 1) / 64 * 64);

  // ---- dK, dV of this wave's keys: registers 4 j .. 4 j + 3 are head dims 32 d + 8 j + 4 h + 0..3.  Through the wave's own rows of
  // the (now idle) dS' image, so that the stores are whole 128-byte rows instead of 16-byte pieces of 32 rows each ----
  KL_PROBE(7, 1);
  if (wave_on) {
    const float fk = p.scale * p.drop_keep, fv = p.drop_keep;
    unsigned short* stg = Timg + wkey0 * LDR;  // [32][72], this wave's
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float f = m == 0 ? fk : fv;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const f32x16& acc = m == 0 ? dk[d] : dv[d];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(uint2*)(stg + li * LDR + d * 32 + 8 * j + 4 * h) =
              make_uint2(cvt_pk_bf16(acc[4 * j] * f, acc[4 * j + 1] * f), cvt_pk_bf16(acc[4 * j + 2] * f, acc[4 * j + 3] * f));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes; nobody else touches these rows
      unsigned short* dst = (m == 0 ? p.dK : p.dV) + b * p.dkv_b + hd * DH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), kk = p.key0 + wkey0 + r;
        const uint4 v = *(const uint4*)(stg + r * LDR + (lane & 7) * 8);
        if (kk < p.Tk) *(uint4*)(dst + (long long)kk * p.lddk + (lane & 7) * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the second matrix overwrites the rows
    }
  }
  KL_PROBE(7, 2);
}
#undef KL_BARRIER
#undef KL_PROBE
#undef AT_PROBE

}  // namespace

SC_API int sc_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Tq, int Tk,
                            long long q_b, long long ldq, long long kv_b, long long ldk, long long o_b, long long ldo,
                            const unsigned char* mask, long long mask_b, long long mask_q, int causal, float scale, float drop_p,
                            const unsigned long long* seed, unsigned long long offset, hipStream_t stream) {
  SC_CHECK_ARG(Q && K && V && O && LSE, "sc_attention_fwd: null pointer");
  const bool stream_ok = (mask == nullptr || mask_q == 0) && ldo % 4 == 0 && o_b % 4 == 0 && (uintptr_t)O % 8 == 0 &&
                         getenv("SC_ATTN_NO_STREAM") == nullptr;
  SC_CHECK_ARG(B >= 1 && H >= 1 && Tq >= 1 && Tk >= 1 && (stream_ok || Tk <= TKMAX),
               "sc_attention_fwd: a query-dependent mask needs 1 <= Tk <= %d (got %d)", TKMAX, Tk);
  SC_CHECK_ARG((long long)B * H <= 65535, "sc_attention_fwd: too many (batch, head) pairs");
  SC_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && q_b % 8 == 0 && kv_b % 8 == 0 && (uintptr_t)Q % 16 == 0 && (uintptr_t)K % 16 == 0 &&
                   (uintptr_t)V % 16 == 0, "sc_attention_fwd: operands must be 16-byte aligned with strides multiple of 8");
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sc_attention_fwd: dropout p=%f needs a seed cell", drop_p);
  SC_CHECK_ARG(offset % 2 == 0, "sc_attention_fwd: the dropout offset must be even");
  AttnArgs a{};
  a.Q = (const unsigned short*)Q; a.K = (const unsigned short*)K; a.V = (const unsigned short*)V; a.O = (unsigned short*)O; a.LSE = LSE;
  a.q_b = q_b; a.ldq = ldq; a.kv_b = kv_b; a.ldk = ldk; a.o_b = o_b; a.ldo = ldo;
  a.mask = mask; a.mask_b = mask_b; a.mask_q = mask_q; a.causal = causal;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldp = (Tk + 7) / 8 * 8; a.scale = scale; a.drop_p = drop_p; a.drop_keep = 1.f / (1.f - drop_p);
  a.seed = seed; a.offset = offset;
  static const bool split_ok = getenv("SC_ATTN_NO_SPLIT") == nullptr;  // (A/B knob)
  if (stream_ok && split_ok && Tq <= 32 && Tk > 64 && !causal && drop_p == 0.f) {
    // few queries against many keys (a decoding position's encoder-decoder attention): the four waves split the key tiles
    const size_t smem4 = (size_t)4 * 2 * 64 * DH * sizeof(unsigned short);  // one 16 KB region per wave
    hipLaunchKernelGGL((attn_fwd_stream_kernel<false, true>), dim3(1, B * H), dim3(256), smem4, stream, a);
    SC_LAUNCH_CHECK("attn_fwd_stream_kernel<split>");
    return SC_OK;
  }
  if (stream_ok) {
    const size_t smem_s = (size_t)3 * 2 * 64 * DH * sizeof(unsigned short);  // three 16 KB stages
    if (drop_p > 0.f) hipLaunchKernelGGL(attn_fwd_stream_kernel<true>, dim3(sc_cdiv(Tq, 128), B * H), dim3(256), smem_s, stream, a);
    else hipLaunchKernelGGL(attn_fwd_stream_kernel<false>, dim3(sc_cdiv(Tq, 128), B * H), dim3(256), smem_s, stream, a);
    SC_LAUNCH_CHECK("attn_fwd_stream_kernel");
    return SC_OK;
  }
  const size_t smem = (size_t)(64 + 2 * TKMAX) * LDR * 2 + (size_t)64 * LDS_S * 4;
  SC_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(sc_cdiv(Tq, 64), B * H), dim3(512), smem, stream, a);
  SC_LAUNCH_CHECK("attn_fwd_kernel");
  return SC_OK;
}

// bytes of the f32 dQ workspace sc_attention_bwd needs (0 up to 256 keys)
SC_API long long sc_attention_bwd_workspace(int B, int H, int Tq, int Tk) {
  return Tk <= TKMAX ? 0 : (long long)B * H * Tq * DH * (long long)sizeof(float);
}

SC_API int sc_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                            void* dK, void* dV, int B, int H, int Tq, int Tk, long long q_b, long long ldq, long long kv_b,
                            long long ldk, long long o_b, long long ldo, long long dq_b, long long lddq, long long dkv_b,
                            long long lddk, const unsigned char* mask, long long mask_b, long long mask_q, int causal, float scale,
                            float drop_p, const unsigned long long* seed, unsigned long long offset, float* dq_workspace,
                            long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(Q && K && V && O && dO && LSE && dQ && dK && dV, "sc_attention_bwd: null pointer");
  SC_CHECK_ARG(B >= 1 && H >= 1 && Tq >= 1 && Tk >= 1, "sc_attention_bwd: bad sizes");
  SC_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0 && q_b % 8 == 0 && kv_b % 8 == 0 && o_b % 8 == 0 &&
                   (uintptr_t)Q % 16 == 0 && (uintptr_t)K % 16 == 0 && (uintptr_t)V % 16 == 0 && (uintptr_t)O % 16 == 0 &&
                   (uintptr_t)dO % 16 == 0, "sc_attention_bwd: operands must be 16-byte aligned with strides multiple of 8");
  SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed) && offset % 2 == 0, "sc_attention_bwd: bad dropout arguments");
  SC_CHECK_ARG(mask == nullptr || mask_q == 0, "sc_attention_bwd: only (B, 1, Tk) key masks (mask_q == 0) are supported; use the causal flag");
  AttnArgs a{};
  a.Q = (const unsigned short*)Q; a.K = (const unsigned short*)K; a.V = (const unsigned short*)V; a.LSE = const_cast<float*>(LSE);
  a.q_b = q_b; a.ldq = ldq; a.kv_b = kv_b; a.ldk = ldk; a.o_b = o_b; a.ldo = ldo;
  a.mask = mask; a.mask_b = mask_b; a.mask_q = mask_q; a.causal = causal;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldp = (Tk + 7) / 8 * 8; a.scale = scale; a.drop_p = drop_p; a.drop_keep = 1.f / (1.f - drop_p);
  a.seed = seed; a.offset = offset;
  a.dO = (const unsigned short*)dO; a.dQ = (unsigned short*)dQ; a.dK = (unsigned short*)dK; a.dV = (unsigned short*)dV;
  a.dq_b = dq_b; a.lddq = lddq; a.dkv_b = dkv_b; a.lddk = lddk;
  const bool kl_ok = lddq % 4 == 0 && dq_b % 4 == 0 && lddk % 8 == 0 && dkv_b % 8 == 0 && (uintptr_t)dQ % 8 == 0 && (uintptr_t)dK % 16 == 0 &&
                     (uintptr_t)dV % 16 == 0 && getenv("SC_ATTN_BWD_OLD") == nullptr;
  if (kl_ok) {
    // 256 keys per launch; more keys: one launch per key block, each sweeping all queries, dQ summed through the caller's f32 workspace
    // in launch order (deterministic: the same workgroup adds a (batch, head)'s blocks one after the other)
    const int nblk = sc_cdiv(Tk, TKMAX);
    SC_CHECK_ARG(nblk == 1 || (dq_workspace && ws_bytes >= sc_attention_bwd_workspace(B, H, Tq, Tk)),
                 "sc_attention_bwd: %d keys need a dQ workspace of %lld bytes (sc_attention_bwd_workspace)", Tk, sc_attention_bwd_workspace(B, H, Tq, Tk));
    const unsigned short* Oc = (const unsigned short*)O;
    for (int blk = 0; blk < nblk; ++blk) {
      a.key0 = blk * TKMAX;
      a.dq_mode = nblk == 1 ? 0 : (blk == 0 ? 1 : (blk == nblk - 1 ? 2 : 3));
      a.dq_ws = dq_workspace;
      const int nw = (Tk - a.key0) > 128 ? 8 : 4;
      const size_t smem = (size_t)(2 * 32 * nw + 4 * 64) * LDR * 2 + 4 * 64 * sizeof(float) + (nw == 8 ? 4 * 4 * 64 * 4 * sizeof(float) : 0);
      const void* fn = nw == 8 ? (drop_p > 0.f ? (const void*)attn_bwd_kl_kernel<8, true> : (const void*)attn_bwd_kl_kernel<8, false>)
                               : (drop_p > 0.f ? (const void*)attn_bwd_kl_kernel<4, true> : (const void*)attn_bwd_kl_kernel<4, false>);
      SC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      void* kargs[] = {&a, &Oc};
      SC_HIP(hipLaunchKernel(fn, dim3(B * H), dim3(64 * nw), kargs, smem, stream));
      SC_LAUNCH_CHECK("attn_bwd_kl_kernel");
    }
    return SC_OK;
  }
  SC_CHECK_ARG(Tk <= TKMAX, "sc_attention_bwd: the previous kernel holds at most %d keys (got %d)", TKMAX, Tk);
  const size_t smem = (size_t)(2 * 64 + 2 * TKMAX) * LDR * 2 + (size_t)2 * 64 * LDPB * 2 + 2 * 64 * sizeof(float);
  SC_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(512), smem, stream, a, (const unsigned short*)O);
  SC_LAUNCH_CHECK("attn_bwd_kernel");
  return SC_OK;
}

#ifdef SC_ATTN_TIMING
SC_API int sc_attention_debug_read(long long* host, int n) {
  SC_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(sc_attn_probe), (size_t)n * sizeof(long long)));
  return SC_OK;
}
#endif
