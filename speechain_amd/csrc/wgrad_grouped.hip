// Grouped weight-gradient products for gfx950: dW_g (n_out x n_in, f32) = dz_g^T x_g for a LIST of layers in ONE launch.
//
// What it replaces: the `grad_weight = grad_output^T input` (+ `grad_bias = column sums of grad_output`) half of every nn.Linear
// backward of the reference's transformer path (paths under /root/reference/):
//   speechain/module/transformer/attention.py:53-59      (q / k / v / output projections)
//   speechain/module/transformer/feed_forward.py:62-84   (in / out layers)
//   speechain/module/postnet/token.py:35-37, speechain/model/ar_asr.py:224-228   (vocabulary and CTC heads)
//   speechain/module/prenet/conv2d.py:171-180, 196-203   (convolutions as im2col products, the Linear part)
// which torch runs as one GEMM (+ one reduction) per layer as autograd reaches it.
//
// Why grouped: one layer's gradient is a small output (512 x 512 ... 2048 x 512) over a long contraction (the ~8000 tokens of the
// batch).  Alone it has 4-64 output tiles for 256 CUs, so it was cut along the contraction (split-K: f32 partial slabs + a second
// launch to add them, 94 times per step).  All layers of a backward pass together have ~1000 tiles of 256 x 256: every
// workgroup runs a FULL contraction for one tile, nothing is split, no partial sums travel, and the launch count drops from 188 to 2-3.
// The products are deferred by the host (speechain_amd/functional.py: _WgradQueue) until the backward pass -- or the piece of it that a
// gradient bucket is waiting for -- has run; only gradients that land in a flat bucket slot are deferred (nothing reads those earlier).
//
// Kernel: both operands are read TRANSPOSED (contraction index = token = slow dimension).  An operand sub-tile is 32 tokens x 128
// features (256-byte rows), staged by LDS-DMA in 1 KB granules of 4 token rows with the 16-byte chunk index XOR-swizzled by
// 4 * (row & 3) on the SOURCE side, so that the transposing fragment reads (ds_read_b64_tr_b16) are bank-conflict free on unpadded
// rows.  Four stages; the barrier of iteration t publishes tile t+1, whose fragments are read while tile t is multiplied
// (software pipelining by half tiles); three tiles stay in flight behind counted vmcnt waits.
//   <2,4,4>: 256 x 256 tile, 8 waves of 128 x 64, 128 KB LDS (32 B of L2->LDS traffic per MFMA cycle and CU)
//   <2,2,2>: 128 x 128 tile, 4 waves of  64 x 64,  64 KB LDS (64 B: twice the traffic per flop, for small groups)
// Deterministic: every output element is one k-ordered accumulation chain.
#include "sc_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BK = 32;     // tokens per K tile
constexpr int SUB = 128 * BK;  // elements of one 32 x 128 operand sub-tile (8 KB)
constexpr int STG_LD = 68;     // f32 row stride of a wave's 32 x 64 epilogue staging tile

struct Prob {
  const unsigned short* dz;  // (tokens, n_out) bf16, row stride ldz
  const unsigned short* x;   // (tokens, n_in)  bf16, row stride ldx
  float* dW;                 // (n_out, n_in) f32, dense
  float* db;                 // (n_out) f32 column sums of dz, or null
  int n_out, n_in, tokens, tiles_n;
  int ldz, ldx;
  int conv_k, pad_;          // conv_k > 1: dW is an nn.Conv1d weight (n_out, n_in / conv_k, conv_k); GEMM column = tap * C + channel
};
// The schedule lives in a device table (caller-provided workspace): header, per-unit tile offsets, per-unit descriptors.
// Units are dealt to the 8 XCDs (each with its own L2) so that every XCD gets about the same amount of work AND all tiles of a unit
// -- which share its dz / x panels -- run on one XCD: units first[x] .. first[x+1] belong to XCD x, tile0[k] = tiles of that XCD in
// front of unit k.  Workgroup b serves XCD b % 8 (how the hardware deals workgroups; speed only), slot b / 8.
// The table is written by small launches that carry the entries as kernel arguments (<= 4 KB each): no host-to-device copy, so the
// whole call can be captured into a hipGraph.
struct Header {
  int first[9];
  int n_units;
  int pad[6];
};
constexpr int CHUNK = 48;  // table entries per writer launch
struct Chunk {
  Prob p[CHUNK];
  int tile0[CHUNK];
  Header hdr;
  int count, offset;
  int* tile0_dst;
  Prob* p_dst;
  Header* hdr_dst;  // written by the first chunk only
};
static_assert(sizeof(Chunk) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(64) void wgrad_table_write_kernel(Chunk c) {
  const int t = threadIdx.x;
  if (c.hdr_dst != nullptr && t == 0) *c.hdr_dst = c.hdr;
  if (t < c.count) {
    c.tile0_dst[c.offset + t] = c.tile0[t];
    c.p_dst[c.offset + t] = c.p[t];
  }
}

// LDS reads are issued by inline asm and waited for by hand.  Reason: for any LDS access the compiler can see (a plain load, the
// ds_read_tr builtin) behind an LDS-DMA it inserts s_waitcnt vmcnt(0) -- it cannot tell which DMA the read depends on -- and that
// drains the whole prefetch ring on every K tile (the loop then ran at 2.7x its MFMA time).  The asm reads are invisible to that pass;
// the data dependence is carried by counted s_waitcnt lgkmcnt + sched_barrier in front of the MFMAs that consume them.
struct Frag { s16x4_t lo, hi; };

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// Byte offset (inside a swizzled 32 x 128 sub-tile) of the first of the two transposing 8-byte reads that make lane l's MFMA 32x32x16
// operand fragment: 8 consecutive k (k = ks*16 + 8*(l>>5) + j) of column `rbase + (l & 31)`; the second read is 4 token rows (1 KB) on.
__device__ __forceinline__ int frag_tr_offset(int rbase, int ks, int lane) {
  const int g = lane >> 4, ip = lane & 15, q = ip >> 2, pp = ip & 3;
  const int kr = ks * 16 + 8 * (g >> 1) + q;
  const int col = rbase + 16 * (g & 1) + 4 * pp;
  return 2 * ((kr >> 2) * 512 + q * 128 + (((col >> 3) ^ (q << 2)) << 3) + (col & 7));
}
__device__ __forceinline__ void frag_tr_issue(unsigned addr, Frag& f) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr));
}
// s_waitcnt lgkmcnt(N) that the fragments "pass through": the compiler believes an asm read has delivered its registers when the
// statement ends, so anything computed from them (the MFMAs, the column-sum adds) may legally be moved in front of a plain wait
// statement -- across basic blocks too, where sched_barrier does not reach (it did: the adds common to both arms of a branch were
// hoisted above both waits and summed stale registers).  With the registers as in/out operands of the wait, every consumer depends
// on the wait itself.
// Only the B fragments are tied (every MFMA of a block consumes one of them, so every MFMA depends on the wait); tying all twelve
// fragments of the 256 x 256 kernel cost 20 VGPRs and pushed it into scratch.
template <int N>
__device__ __forceinline__ void frag_wait(Frag (&fb)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fb[0].lo), "+v"(fb[0].hi), "+v"(fb[1].lo), "+v"(fb[1].hi) : [n] "n"(N));
}

__device__ __forceinline__ bf16x8_t frag_value(const Frag& f) {
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// -DSC_WGRAD_TIMING (instrumented builds only, scripts/wgrad_probe.py): every workgroup records its physical XCC, its start and end on the
// chip-wide 100 MHz clock and its K-tile count; read back with sc_wgrad_debug_read (not part of the ABI).
#ifdef SC_WGRAD_TIMING
__device__ long long sc_wgrad_probe[4096 * 4];
#endif

template <int WM, int WN, int MI>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1) void wgrad_grouped_kernel(const Header* __restrict__ hdr, const int* __restrict__ tile0,
                                                                                          const Prob* __restrict__ probs) {
  constexpr int NW = WM * WN, TM = WM * MI * 32, TN = WN * 64;
  constexpr int SA = TM / 128, SB = TN / 128;
  static_assert(TM % 128 == 0 && TN % 128 == 0 && NW == 4 * SA, "tile = whole 128-column sub-tiles; 16 threads per sub-tile row pair");
  constexpr int STAGES = 4, STAGE_ELEMS = (SA + SB) * SUB;
  constexpr int LPT = 8 * (SA + SB) / NW;  // DMA granules per wave per tile
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsmem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid % WN;
  const int h = lane >> 5, li = lane & 31;

  // tile of this workgroup: slot blockIdx / 8 of XCD blockIdx % 8's list
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int g = hdr->first[xcd];
  const int gend = hdr->first[xcd + 1];
  if (g >= gend) return;
  while (g + 1 < gend && slot >= tile0[g + 1]) ++g;
  const Prob P = probs[g];
  const int lt = slot - tile0[g];
  if (lt >= ((P.n_out + TM - 1) / TM) * P.tiles_n) return;  // this XCD has fewer tiles than the busiest one
  const int tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int nk = P.tokens / BK;
#ifdef SC_WGRAD_TIMING
  if (tid == 0 && blockIdx.x < 4096) {
    unsigned xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    sc_wgrad_probe[blockIdx.x * 4 + 0] = xcc_ & 15;
    sc_wgrad_probe[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    sc_wgrad_probe[blockIdx.x * 4 + 3] = nk;
  }
#endif

  // per-lane source pointers of this wave's granules: granule index gi = wid + NW i covers sub-tile gi / 8 (A sub-tiles first),
  // token rows 4 (gi % 8) .. + 3; lane l fills LDS slot (row l >> 4, chunk l & 15) from source chunk (l & 15) ^ (4 (l >> 4)).
  // Column chunks beyond the matrix are clamped to the last valid chunk (their products are never stored).
  const int q = lane >> 4, schunk = (lane & 15) ^ (q << 2);
  const unsigned short* src[LPT];
  long long step[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int gi = wid + NW * i, s = gi >> 3, gr = gi & 7;
    if (s < SA) {
      src[i] = P.dz + (long long)(gr * 4 + q) * P.ldz + min(m0 + s * 128 + schunk * 8, P.n_out - 8);
      step[i] = (long long)BK * P.ldz;
    } else {
      src[i] = P.x + (long long)(gr * 4 + q) * P.ldx + min(n0 + (s - SA) * 128 + schunk * 8, P.n_in - 8);
      step[i] = (long long)BK * P.ldx;
    }
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define SC_ISSUE(T)                                                                                      \
  do {                                                                                                   \
    unsigned short* st_ = dsmem + ((T) % STAGES) * STAGE_ELEMS;                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < LPT; ++i_)                                                   \
    {                                                                                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)src[i_], (lptr_t)(st_ + (wid + NW * i_) * 512), 16, 0, 0); \
      src[i_] += step[i_];  /* tiles are issued in order: running pointers, no 64-bit multiply per tile */ \
    }                                                                                                    \
  } while (0)

  if (nk > 0) SC_ISSUE(0);
  if (nk > 1) SC_ISSUE(1);
  if (nk > 2) SC_ISSUE(2);
  if (nk > 3) SC_ISSUE(3);

  // bias gradient: column sums of dz over the tokens, taken from the landed dz tiles by the workgroups of the first n tile.
  // Thread t reads column chunk t & 15 of sub-tile (t >> 4) % SA, token rows kg and kg + 16 with kg = (t >> 4) / SA.
  const bool csum = P.db != nullptr && tn == 0;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  const int cs_sa = (tid >> 4) % SA, cs_kg = (tid >> 4) / SA;
  int csoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kr = cs_kg + 16 * i;
    csoff[i] = 2 * (cs_sa * SUB + (kr >> 2) * 512 + (kr & 3) * 128 + (((tid & 15) ^ ((kr & 3) << 2)) << 3));  // bytes
  }

  // fragment sources (byte offsets inside a stage): A rows (features of dz) wr*MI*32 + i*32 .., B rows (features of x) wc*64 + j*32 ..
  const unsigned lds0 = lds_addr(dsmem);
  unsigned a_off[MI][2], b_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ra = (wr * MI + i) * 32;
      a_off[i][ks] = lds0 + 2 * (ra >> 7) * SUB + frag_tr_offset(ra & 127, ks, lane);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rb = (wc * 2 + j) * 32;
      b_off[j][ks] = lds0 + 2 * (SA + (rb >> 7)) * SUB + frag_tr_offset(rb & 127, ks, lane);
    }
  }
  constexpr int RPH = 2 * (MI + 2);  // LDS read instructions per half tile of fragments

#define SC_WAIT_TILES(AHEAD)                                                                  \
  do {                                                                                        \
    const int ah_ = (AHEAD);                                                                  \
    if (ah_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");              \
    else if (ah_ == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");         \
    else if (ah_ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");             \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
  } while (0)
#define SC_FRAGS(T, KS, FA, FB)                                                               \
  do {                                                                                        \
    const unsigned st_ = ((T) % STAGES) * (2 * STAGE_ELEMS);                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) frag_tr_issue(st_ + a_off[i_][KS], FA[i_]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) frag_tr_issue(st_ + b_off[j_][KS], FB[j_]);  \
  } while (0)
// column sums: two 16-byte reads per thread and tile, issued with the fragments, added once the next lgkmcnt(0) has passed
#define SC_CSUM_ISSUE(T)                                                                      \
  do {                                                                                        \
    if constexpr (decltype(with_csum)::value) {                                               \
      const unsigned st_ = lds0 + ((T) % STAGES) * (2 * STAGE_ELEMS);                         \
      asm volatile("ds_read_b128 %0, %1" : "=&v"(csu[0]) : "v"(st_ + csoff[0]));             \
      asm volatile("ds_read_b128 %0, %1" : "=&v"(csu[1]) : "v"(st_ + csoff[1]));             \
    }                                                                                         \
  } while (0)
#define SC_CSUM_ADD()                                                                         \
  do {                                                                                        \
    if constexpr (decltype(with_csum)::value) {                                               \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(csu[0]), "+v"(csu[1]));  /* the adds depend on this wait (see frag_wait) */ \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                        \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                    \
          cs[2 * e_] += __uint_as_float(csu[i_][e_] << 16);                                   \
          cs[2 * e_ + 1] += __uint_as_float(csu[i_][e_] & 0xffff0000u);                       \
        }                                                                                     \
    }                                                                                         \
  } while (0)
// the MFMAs consume the outputs of the wait in front of them (frag_wait); the sched_barriers keep the block where it is: not above
// the reads it should overlap with, not below the next barrier (that would expose the reads just issued)
#define SC_MMA(FA, FB, NWAIT)                                                                 \
  do {                                                                                        \
    frag_wait<NWAIT>(FB);  /* all but the NWAIT youngest LDS reads have delivered */          \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                        \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_value(FA[i_]), frag_value(FB[j_]), acc[i_][j_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);  /* ... nor sink below the next wait / barrier (that would expose the reads just issued) */ \
  } while (0)
// tile T+1 has landed for this wave; after the barrier it is visible to all, every wave holds tile T in registers, and the stage of
// tile T is free for tile T + STAGES
#define SC_PUBLISH_NEXT(T)                                                                    \
  do {                                                                                        \
    SC_WAIT_TILES(min(STAGES - 2, nk - 2 - (T)));                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                             \
    if ((T) + STAGES < nk) SC_ISSUE((T) + STAGES);                                            \
  } while (0)
#define SC_PUBLISH_STEADY(T)                                                                  \
  do {                                                                                        \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                             \
    SC_ISSUE((T) + STAGES);                                                                   \
  } while (0)
  // (two instances of the loop, with and without the column sums: a run-time branch around LDS reads inside the steady state makes
  // the compiler wait for ALL outstanding reads where the paths merge, which serialises fragment reads and MFMAs again)
  auto main_loop = [&](auto with_csum) {
    constexpr int CSR = decltype(with_csum)::value ? 2 : 0;  // extra reads per tile for the column sums
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 csu[2];
    Frag fa0[MI], fb0[2], fa1[MI], fb1[2];
    if (nk > 0) {
      SC_WAIT_TILES(min(STAGES - 1, nk - 1));
      __builtin_amdgcn_s_barrier();
      SC_FRAGS(0, 0, fa0, fb0);
    }
    int t = 0;
    for (; t + STAGES < nk; ++t) {  // steady state, branch-free
      SC_FRAGS(t, 1, fa1, fb1);
      SC_CSUM_ISSUE(t);
      SC_MMA(fa0, fb0, RPH + CSR);  // the first half's fragments (issued one half earlier) are in; this half's reads stay in flight
      SC_PUBLISH_STEADY(t);         // (contains lgkmcnt(0): the second half's fragments and the column-sum reads are in)
      SC_CSUM_ADD();
      SC_FRAGS(t + 1, 0, fa0, fb0);
      SC_MMA(fa1, fb1, RPH);
    }
    for (; t < nk; ++t) {
      SC_FRAGS(t, 1, fa1, fb1);
      SC_CSUM_ISSUE(t);
      SC_MMA(fa0, fb0, RPH + CSR);
      if (t + 1 < nk) {
        SC_PUBLISH_NEXT(t);
        SC_CSUM_ADD();
        SC_FRAGS(t + 1, 0, fa0, fb0);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SC_CSUM_ADD();
      }
      SC_MMA(fa1, fb1, RPH);  // (its operands were waited for in either arm; this wait only carries the dependence)
    }
  };
  if (csum) main_loop(std::true_type{});
  else main_loop(std::false_type{});
#undef SC_WAIT_TILES
#undef SC_FRAGS
#undef SC_CSUM_ISSUE
#undef SC_CSUM_ADD
#undef SC_MMA
#undef SC_PUBLISH_NEXT
#undef SC_PUBLISH_STEADY
#undef SC_ISSUE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all waves are done with the operand stages before they become staging

  if (csum) {  // 16 token-row groups x TM columns through LDS
    float* red = (float*)dsmem;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[cs_kg * TM + cs_sa * 128 + (tid & 15) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < TM && m0 + tid < P.n_out) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += red[k * TM + tid];
      P.db[m0 + tid] = v;
    }
    __syncthreads();
  }

  // accumulators -> the wave's private LDS staging slice -> 16-byte f32 stores into dW
  float* stage = (float*)dsmem + wid * 32 * STG_LD;
  const bool vec = (P.n_in & 3) == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * STG_LD + j * 32 + li] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int qq = lane + 64 * it;
      const int row = qq >> 4, c4 = (qq & 15) * 4;
      const int m = m0 + (wr * MI + i) * 32 + row, n = n0 + wc * 64 + c4;
      if (m < P.n_out && n < P.n_in) {
        const f32x4 v = *(const f32x4*)(stage + row * STG_LD + c4);
        float* dst = P.dW + (long long)m * P.n_in + n;
        if (P.conv_k > 1) {  // Conv1d weight layout: GEMM column tap * C + c -> (c, tap)
          const int C = P.n_in / P.conv_k;
          for (int e = 0; e < 4 && n + e < P.n_in; ++e) P.dW[(long long)m * P.n_in + ((n + e) % C) * P.conv_k + (n + e) / C] = v[e];
        } else if (n + 4 <= P.n_in && vec) *(f32x4*)dst = v;
        else for (int e = 0; e < 4 && n + e < P.n_in; ++e) dst[e] = v[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
#ifdef SC_WGRAD_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && blockIdx.x < 4096) sc_wgrad_probe[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int WM, int WN, int MI>
int launch(const Header* hdr, const int* tile0, const Prob* probs, int grid, hipStream_t stream) {
  constexpr int TM = WM * MI * 32, TN = WN * 64;
  constexpr size_t smem = (size_t)4 * (TM / 128 + TN / 128) * SUB * sizeof(unsigned short);
  static_assert(smem >= (size_t)WM * WN * 32 * STG_LD * sizeof(float) && smem >= (size_t)16 * TM * sizeof(float),
                "operand stages must cover the epilogue staging and the column-sum scratch");
  auto kernel = wgrad_grouped_kernel<WM, WN, MI>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return sc_set_error(SC_ERR_LAUNCH, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * WM * WN), smem, stream, hdr, tile0, probs);
  SC_LAUNCH_CHECK("wgrad_grouped_kernel");
  return SC_OK;
}

// scheduling units of one problem for a tile edge: pieces of at most 16 tiles along n_out (whole tile rows)
int rows_per_unit(int n_in, int tile) { return std::max(1, 16 / sc_cdiv(n_in, tile)) * tile; }
int units_of(int n_out, int n_in, int tile) { return sc_cdiv(n_out, rows_per_unit(n_in, tile)); }
long long table_bytes(long long units) {
  return (long long)sizeof(Header) + ((units * (long long)sizeof(int) + 15) / 16) * 16 + units * (long long)sizeof(Prob);
}

}  // namespace

#ifdef SC_WGRAD_TIMING
extern "C" __attribute__((visibility("default"))) int sc_wgrad_debug_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sc_wgrad_probe), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif


static thread_local int g_last_tile = 0;
// output tile edge (128 / 256) of the calling thread's last sc_wgrad_grouped call (measurement aid for bench.py)
SC_API int sc_wgrad_grouped_last_tile(void) { return g_last_tile; }

// 1 if a product of this shape can go through sc_wgrad_grouped (whole 32-token K tiles, 16-byte aligned rows on both sides)
SC_API int sc_wgrad_grouped_supported(int n_out, int n_in, int tokens, long long ldz, long long ldx) {
  return n_out >= 8 && n_in >= 8 && n_out % 8 == 0 && n_in % 8 == 0 && tokens >= BK && tokens % BK == 0 && ldz % 8 == 0 && ldx % 8 == 0 &&
         ldz >= n_out && ldx >= 8 && ldz < (1LL << 31) && ldx < (1LL << 31);  // (ldx < n_in: overlapping rows, see sc_pad_rows)
}

// bytes of device workspace sc_wgrad_grouped needs for its schedule table (for either tile size)
SC_API long long sc_wgrad_grouped_workspace(int n, const int* n_out, const int* n_in) {
  long long u128 = 0, u256 = 0;
  for (int g = 0; g < n; ++g) {
    u128 += units_of(n_out[g], n_in[g], 128);
    u256 += units_of(n_out[g], n_in[g], 256);
  }
  return table_bytes(std::max(u128, u256));
}

// dW[g] (n_out[g] x n_in[g], f32, dense) = dz[g]^T x[g]; db[g] (n_out[g], f32; entries may be null) = column sums of dz[g].
// dz[g]: (tokens[g], n_out[g]) bf16 with row stride ldz[g]; x[g]: (tokens[g], n_in[g]) bf16 with row stride ldx[g].
// All arrays are HOST arrays of n entries holding device pointers / sizes; every problem must satisfy sc_wgrad_grouped_supported.
// tile: 0 = choose (256 x 256 tiles when they give every CU work, else 128 x 128), 128 or 256 = force.
// workspace: device memory, sc_wgrad_grouped_workspace(n, n_out, n_in) bytes, 16-byte aligned; it must stay untouched until the
// launches of this call have run.
SC_API int sc_wgrad_grouped(int n, const void* const* dz, const void* const* x, float* const* dW, float* const* db, const int* n_out,
                            const int* n_in, const int* tokens, const long long* ldz, const long long* ldx, const int* conv_k, int tile,
                            void* workspace, long long ws_bytes, hipStream_t stream) {
  SC_CHECK_ARG(n >= 1 && dz && x && dW && db && n_out && n_in && tokens && ldz && ldx, "sc_wgrad_grouped: null array or empty list");
  SC_CHECK_ARG(tile == 0 || tile == 128 || tile == 256, "sc_wgrad_grouped: tile must be 0, 128 or 256");
  SC_CHECK_ARG(workspace && (uintptr_t)workspace % 16 == 0, "sc_wgrad_grouped: workspace is null or not 16-byte aligned");
  long long t256 = 0;
  for (int g = 0; g < n; ++g) {
    SC_CHECK_ARG(dz[g] && x[g] && dW[g], "sc_wgrad_grouped: problem %d has a null pointer", g);
    SC_CHECK_ARG(sc_wgrad_grouped_supported(n_out[g], n_in[g], tokens[g], ldz[g], ldx[g]),
                 "sc_wgrad_grouped: problem %d (n_out=%d n_in=%d tokens=%d ldz=%lld ldx=%lld) is not supported", g, n_out[g], n_in[g],
                 tokens[g], ldz[g], ldx[g]);
    SC_CHECK_ARG(!conv_k || conv_k[g] <= 1 || n_in[g] % conv_k[g] == 0, "sc_wgrad_grouped: problem %d: n_in %d is not a multiple of conv_k %d", g,
                 n_in[g], conv_k ? conv_k[g] : 0);
    SC_CHECK_ARG((uintptr_t)dz[g] % 16 == 0 && (uintptr_t)x[g] % 16 == 0 && (uintptr_t)dW[g] % 16 == 0,
                 "sc_wgrad_grouped: problem %d has a pointer that is not 16-byte aligned", g);
    t256 += (long long)sc_cdiv(n_out[g], 256) * sc_cdiv(n_in[g], 256);
  }
  static const int forced = [] { const char* e = getenv("SC_WGRAD_TILE"); return e ? atoi(e) : 0; }();
  if (tile == 0) tile = forced;
  if (tile == 0) tile = t256 >= 224 ? 256 : 128;
  g_last_tile = tile;
  // Scheduling units: a problem is cut along n_out into pieces of at most 16 tiles (whole tile rows: the tiles of a row share their dz
  // panel), so that no unit is more than half an XCD's worth of workgroups -- the 5000-row vocabulary heads are 40 tiles.
  struct Unit { int g, m_base, rows; long long work; int tiles; };
  std::vector<Unit> units;
  for (int g = 0; g < n; ++g) {
    const int tiles_n = sc_cdiv(n_in[g], tile), rpu = rows_per_unit(n_in[g], tile);
    for (int mb = 0; mb < n_out[g]; mb += rpu) {
      const int rows = std::min(rpu, n_out[g] - mb);
      const int t = sc_cdiv(rows, tile) * tiles_n;
      units.push_back({g, mb, rows, (long long)t * tokens[g], t});
    }
  }
  const int nu = (int)units.size();
  SC_CHECK_ARG(ws_bytes >= table_bytes(nu), "sc_wgrad_grouped: workspace too small (%lld < %lld bytes)", ws_bytes, table_bytes(nu));
  // Largest units first, each dealt to the XCD with the least work so far.  Inside an XCD the order stays largest-first: the workgroups
  // that start last are the short ones.
  std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& c) { return a.work > c.work; });
  std::vector<int> lists[8];
  long long work[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < nu; ++k) {
    int best = 0;
    for (int xc = 1; xc < 8; ++xc)
      if (work[xc] < work[best]) best = xc;
    lists[best].push_back(k);
    work[best] += units[k].work;
  }
  Header hdr{};
  std::vector<Prob> probs(nu);
  std::vector<int> tile0(nu);
  int k = 0, most = 0;
  static const bool debug_ld0 = getenv("SC_WGRAD_DEBUG_LD0") != nullptr;  // timing experiment only (wrong results): every K tile re-reads the first rows
  for (int xc = 0; xc < 8; ++xc) {
    hdr.first[xc] = k;
    int t0 = 0;
    for (int ui : lists[xc]) {
      const Unit& u = units[ui];
      const int g = u.g;
      Prob& P = probs[k];
      P.dz = (const unsigned short*)dz[g] + u.m_base;
      P.x = (const unsigned short*)x[g];
      P.dW = dW[g] + (long long)u.m_base * n_in[g];
      P.db = db[g] ? db[g] + u.m_base : nullptr;
      P.n_out = u.rows; P.n_in = n_in[g]; P.tokens = tokens[g];
      P.ldz = debug_ld0 ? 0 : (int)ldz[g]; P.ldx = debug_ld0 ? 0 : (int)ldx[g];
      P.conv_k = (conv_k && conv_k[g] > 1) ? conv_k[g] : 1;
      P.tiles_n = sc_cdiv(n_in[g], tile);
      tile0[k++] = t0;
      t0 += u.tiles;
    }
    most = std::max(most, t0);
  }
  hdr.first[8] = k;
  hdr.n_units = nu;
  // device table: header | tile0[nu] (padded to 16 bytes) | probs[nu]
  Header* d_hdr = (Header*)workspace;
  int* d_tile0 = (int*)((char*)workspace + sizeof(Header));
  Prob* d_probs = (Prob*)((char*)d_tile0 + (((long long)nu * sizeof(int) + 15) / 16) * 16);
  for (int first = 0; first < nu; first += CHUNK) {
    Chunk c{};
    c.count = std::min(CHUNK, nu - first);
    c.offset = first;
    for (int i = 0; i < c.count; ++i) { c.p[i] = probs[first + i]; c.tile0[i] = tile0[first + i]; }
    c.hdr = hdr;
    c.hdr_dst = first == 0 ? d_hdr : nullptr;
    c.tile0_dst = d_tile0;
    c.p_dst = d_probs;
    hipLaunchKernelGGL(wgrad_table_write_kernel, dim3(1), dim3(64), 0, stream, c);
    SC_LAUNCH_CHECK("wgrad_table_write_kernel");
  }
  return tile == 256 ? launch<2, 4, 4>(d_hdr, d_tile0, d_probs, 8 * most, stream) : launch<2, 2, 2>(d_hdr, d_tile0, d_probs, 8 * most, stream);
}
