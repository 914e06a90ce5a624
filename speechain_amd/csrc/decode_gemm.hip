// Linear layers of a decoding step (bf16): out = act(LN(x) W^T + bias) + residual for a FEW HUNDRED rows -- the 512 hypothesis rows of a beam
// search step (speechain/infer_func/beam_search.py:325-333 re-runs the decoder per step; one step of the cached form here runs 6 layers x
// (q|k|v, out, q, out, ffn-in, ffn-out) + the token head = 37 such products, each 0.27-1.1 GFLOP: launch-latency sized).
//
// The general NT kernel (csrc/gemm.hip) gives such a product 64 workgroups of 64 x 64 outputs, each pulling 128 KB through ONE CU's fill
// path behind a 4-stage ring of 32-deep K tiles: ~10 us for 0.27 GFLOP, and a LayerNorm launch (~5 us + a dependency gap) in front of half of
// them.  Here a workgroup owns a 32 x 32 output tile (256 workgroups for N = 512: every CU pulls 64 KB) and asks for its WHOLE contraction
// at once (K <= 512; longer contractions, the ffn-out product, run super-tiles of 512 through two stages).  The four waves split K: wave w
// stages k-tiles 4 w .. 4 w + 3 of both operands and multiplies exactly those -- it waits for its own LDS-DMA only, no barrier in the product
// loop -- and the four partial 32 x 32 sums meet in LDS for the epilogue (bias, ReLU / GELU, residual, bf16 store).
//
// LayerNorm in front of the product (pre-LN layers: nn.LayerNorm(d, eps) on the rows the product consumes; decoder.py:90-92 in the
// reference) is folded in when the whole row is on chip (K <= 512): once the rows have landed the workgroup normalises them IN PLACE in LDS --
// 8 lanes per row, two-pass variance in f32, gamma / beta staged in LDS, bf16 result: the arithmetic of layernorm_fwd_vec_kernel -- and the
// products read the normalised rows.  Every column tile of a row block repeats that (16-157 times 32 KB of LDS traffic): nothing next to the
// launch it replaces (19 LayerNorm launches per step).
//
// The cheaper form of the same fold (LNM = 2; what the decoder's steps use): LayerNorm is affine per row, so
//   LN(x) W^T + b = rstd (x (gamma . W)^T - mean colsum) + (W beta + b),   colsum[n] = sum_k bf16(gamma[k] W[n][k]).
// The product runs on the RAW rows against the pre-scaled weight (made once per weight version on the host side: functional.py), the waves
// start multiplying as their own k-tiles land exactly as without a LayerNorm, and only the epilogue needs the row statistics: one pass
// over the staged rows (8 lanes per row, two-pass variance) AFTER the product loop, no write-back, no gamma / beta on chip.
//
// LDS image of an operand's super-tile: 16 k-tiles of [32 rows][32 k] (64-byte rows; one LDS-DMA instruction = 16 rows), 16-byte chunk index
// XOR-ed with (row >> 2) & 3 on the source side and in the read addresses, as in gemm_bf16_nt_dma_kernel.
#include "sc_common.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* dg_gptr_t;
typedef __attribute__((address_space(3))) void* dg_lptr_t;

struct DecArgs {
  const unsigned short* X;   // (M, K) rows, leading dimension ldx
  const unsigned short* W;   // (N, K) row-major weight (nn.Linear layout), leading dimension ldw
  unsigned short* Y;         // (M, N), leading dimension ldy
  const float* bias;         // (N) or null
  const float* gamma;        // LayerNorm weight / bias (K) or null: no LayerNorm
  const float* beta;
  const float* colsum;       // LNM = 2: row sums of the pre-scaled weight (N); bias then holds W beta + b
  const unsigned short* res; // residual (M, N), leading dimension ldr, or null
  long long ldx, ldw, ldy, ldr;
  int M, N, K;
  int act;                   // 0 none, 1 ReLU, 2 GELU
  float eps;
};

constexpr int KS = 512;  // super-tile depth: 16 k-tiles of 32

// sum over each group of 8 consecutive lanes, in every lane of the group (DPP: two quad permutes and the half-row mirror)
__device__ __forceinline__ float oct_sum(float v) {
  v += sc_dpp<0xB1, 0xF>(0.f, v);
  v += sc_dpp<0x4E, 0xF>(0.f, v);
  v += sc_dpp<0x141, 0xF>(0.f, v);
  return v;
}

// TR: the output tile is (32 TR) x (32 TR).  TR = 1 puts a workgroup on every CU for the narrow products (N = 512: 256 workgroups of 64 KB
// each); TR = 2 (64 x 64, 128 KB of operands: K <= 512 only) halves the bytes per output for the wide ones (q|k|v, ffn-in, the token head),
// where 32 x 32 tiles would re-read every row of x N / 32 times.
template <int LNM, int TR>
__global__ __launch_bounds__(256, TR == 1 ? 2 : 1) void decode_linear_kernel(DecArgs p) {
  constexpr bool LN = LNM == 1;  // in-place normalisation of the staged rows; LNM == 2: statistics for the epilogue only
  constexpr int TROWS = 32 * TR;         // rows of x / of w per workgroup
  constexpr int KTE = TROWS * 32;        // elements of one operand k-tile
  constexpr int OPER = 16 * KTE;         // elements of one operand's super-tile
  constexpr int STAGE = 2 * OPER;        // A then B (64 KB x TR)
  constexpr int DPW = 16 * TR;           // LDS-DMA instructions per wave and super-tile
  constexpr int RED_LD = TROWS + 4;      // f32 row stride of a wave's partial-sum tile
  extern __shared__ __attribute__((aligned(1024))) unsigned short dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int n0 = blockIdx.x * TROWS, m0 = blockIdx.y * TROWS;
  const int nst = (p.K + KS - 1) / KS;
  const int nstages = nst > 1 ? 2 : 1;
  float* gb = (float*)(dsm + nstages * STAGE);  // LN only: gamma (K floats, padded to 512) then beta

  // this wave stages (and later multiplies) k-tiles 4 wid .. 4 wid + 3 of a super-tile: instruction j covers k-tile 4 wid + j / (2 TR), rows
  // 16 (j % (2 TR)) .. + 15 of A and of B.  k-tiles beyond K are redirected to the last valid one (never multiplied).
  auto issue = [&](int st) {
    unsigned short* dst = dsm + (st & 1) * STAGE;
    const int kt_n = min(16, (p.K - st * KS) / 32);
#pragma unroll
    for (int j = 0; j < 8 * TR; ++j) {
      const int kt = 4 * wid + j / (2 * TR), rg = j % (2 * TR);
      const int row = rg * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const long long koff = (long long)st * KS + min(kt, kt_n - 1) * 32 + chunk * 8;
      __builtin_amdgcn_global_load_lds((dg_gptr_t)(p.X + (long long)min(m0 + row, p.M - 1) * p.ldx + koff), (dg_lptr_t)(dst + kt * KTE + rg * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((dg_gptr_t)(p.W + (long long)min(n0 + row, p.N - 1) * p.ldw + koff),
                                       (dg_lptr_t)(dst + OPER + kt * KTE + rg * 512), 16, 0, 0);
    }
  };
  if (LN) {  // gamma | beta -> LDS: 16 bytes per lane, waves 0 / 1 the two halves of gamma, waves 2 / 3 of beta (K <= 512: 2 KB each)
    const float* src = (wid < 2 ? p.gamma : p.beta) + min((wid & 1) * 256 + lane * 4, p.K - 4);
    __builtin_amdgcn_global_load_lds((dg_gptr_t)src, (dg_lptr_t)(gb + (wid >> 1) * 512 + (wid & 1) * 256), 16, 0, 0);
  }
  issue(0);
  if (nst > 1) issue(1);

  f32x16 acc[TR][TR];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TR; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (LN) {
    // the rows are complete once EVERY wave's k-tiles have landed: wait, barrier, then 8 lanes per row normalise it in place
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int sub = lane & 7;
    const int nch = p.K >> 3;  // 16-byte chunks per row
    for (int rep = 0; rep < TR; ++rep) {
      const int row = wid * 8 * TR + rep * 8 + (lane >> 3);
      const int sw = (row >> 2) & 3;
      uint4 raw[8];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = sub + 8 * i;  // chunk of the row: k = 8 c, k-tile c >> 2, slot (c & 3) ^ sw
        raw[i] = make_uint4(0, 0, 0, 0);
        if (c < nch) raw[i] = *(const uint4*)(dsm + (c >> 2) * KTE + row * 32 + (((c & 3) ^ sw) << 3));
        const unsigned w4[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) s += __uint_as_float(w4[j] << 16) + __uint_as_float(w4[j] & 0xFFFF0000u);
      }
      s = oct_sum(s);
      const float mean = s / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (sub + 8 * i < nch) {
          const unsigned w4[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d0 = __uint_as_float(w4[j] << 16) - mean, d1 = __uint_as_float(w4[j] & 0xFFFF0000u) - mean;
            q = fmaf(d0, d0, fmaf(d1, d1, q));
          }
        }
      q = oct_sum(q);
      const float rstd = rsqrtf(q / (float)p.K + p.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = sub + 8 * i;
        if (c < nch) {
          const f32x4 g0 = *(const f32x4*)(gb + 8 * c), g1 = *(const f32x4*)(gb + 8 * c + 4);
          const f32x4 b0 = *(const f32x4*)(gb + 512 + 8 * c), b1 = *(const f32x4*)(gb + 512 + 8 * c + 4);
          const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
          const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          const unsigned w4[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
          unsigned o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float y0 = fmaf((__uint_as_float(w4[j] << 16) - mean) * rstd, gg[2 * j], bb[2 * j]);
            const float y1 = fmaf((__uint_as_float(w4[j] & 0xFFFF0000u) - mean) * rstd, gg[2 * j + 1], bb[2 * j + 1]);
            o4[j] = (unsigned)sc_f32_to_bf16(y0) | ((unsigned)sc_f32_to_bf16(y1) << 16);
          }
          *(uint4*)(dsm + (c >> 2) * KTE + row * 32 + (((c & 3) ^ sw) << 3)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  for (int st = 0; st < nst; ++st) {
    if (!LN) {  // this wave's own k-tiles of super-tile st have landed (the next super-tile's may stay in flight): no barrier needed
      if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned short* as = dsm + (st & 1) * STAGE;
    const unsigned short* bs = as + OPER;
    const int kt_n = min(16, (p.K - st * KS) / 32);
    const int so = ((2 * 0 + h) ^ ((li >> 2) & 3)) << 3, s1 = ((2 * 1 + h) ^ ((li >> 2) & 3)) << 3;  // ((32 i + li) >> 2) & 3 == (li >> 2) & 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kt = 4 * wid + j;
      if (kt < kt_n) {
        bf16x8_t a0[TR], a1[TR], b0[TR], b1[TR];
#pragma unroll
        for (int i = 0; i < TR; ++i) {
          a0[i] = *(const bf16x8_t*)(as + kt * KTE + (32 * i + li) * 32 + so);
          a1[i] = *(const bf16x8_t*)(as + kt * KTE + (32 * i + li) * 32 + s1);
          b0[i] = *(const bf16x8_t*)(bs + kt * KTE + (32 * i + li) * 32 + so);
          b1[i] = *(const bf16x8_t*)(bs + kt * KTE + (32 * i + li) * 32 + s1);
        }
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int jj = 0; jj < TR; ++jj) {
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[jj], acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[jj], acc[i][jj], 0, 0, 0);
          }
      }
    }
    if (st + 2 < nst) {  // refill this wave's own quarter of the stage: nobody else reads it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(st + 2);
    }
  }

  // ---- the four partial sums meet in LDS (over the operand stage, once every wave is done reading it); epilogue by all 256 threads:
  //      4 consecutive columns each: bias, activation, residual, bf16 store ----
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* rowstat = gb;  // LNM == 2: [TROWS] mean, [TROWS] rstd (behind the operand stage)
  if (LNM == 2) {
    // every wave's k-tiles have landed (each waited for its own before multiplying them): the rows are complete
    const int sub = lane & 7;
    const int nch = p.K >> 3;
    for (int rep = 0; rep < TR; ++rep) {
      const int row = wid * 8 * TR + rep * 8 + (lane >> 3);
      const int sw = (row >> 2) & 3;
      uint4 raw[8];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = sub + 8 * i;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (c < nch) raw[i] = *(const uint4*)(dsm + (c >> 2) * KTE + row * 32 + (((c & 3) ^ sw) << 3));
        const unsigned w4[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += __uint_as_float(w4[j] << 16) + __uint_as_float(w4[j] & 0xFFFF0000u);
      }
      sm = oct_sum(sm);
      const float mean = sm / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (sub + 8 * i < nch) {
          const unsigned w4[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d0 = __uint_as_float(w4[j] << 16) - mean, d1 = __uint_as_float(w4[j] & 0xFFFF0000u) - mean;
            q = fmaf(d0, d0, fmaf(d1, d1, q));
          }
        }
      q = oct_sum(q);
      if (sub == 0) {
        rowstat[row] = mean;
        rowstat[TROWS + row] = rsqrtf(q / (float)p.K + p.eps);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (the partial sums below overwrite the rows)
  }
  float* red = (float*)dsm;  // [4 waves][TROWS][RED_LD]
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TR; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[wid * TROWS * RED_LD + (32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * RED_LD + 32 * j + li] = acc[i][j][r];
  __syncthreads();
  constexpr int CPT = TROWS * TROWS / 256;  // consecutive columns per thread: 4 (32 x 32) or 16 (64 x 64)
  {
    const int row = tid / (TROWS / CPT), c0 = (tid % (TROWS / CPT)) * CPT;
    const int m = m0 + row, n = n0 + c0;
    if (m < p.M && n < p.N) {
      float o[CPT];
#pragma unroll
      for (int q4 = 0; q4 < CPT / 4; ++q4) {
        f32x4 v = *(const f32x4*)(red + row * RED_LD + c0 + 4 * q4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red + w * TROWS * RED_LD + row * RED_LD + c0 + 4 * q4);
        o[4 * q4] = v[0]; o[4 * q4 + 1] = v[1]; o[4 * q4 + 2] = v[2]; o[4 * q4 + 3] = v[3];
      }
      const int nv = min(CPT, p.N - n);
      if (LNM == 2) {
        const float mean = rowstat[row], rstd = rowstat[TROWS + row];
#pragma unroll
        for (int e = 0; e < CPT; ++e)
          if (e < nv) o[e] = rstd * (o[e] - mean * p.colsum[n + e]);
      }
#pragma unroll
      for (int e = 0; e < CPT; ++e) {
        if (p.bias && e < nv) o[e] += p.bias[n + e];
        if (p.act == 1) o[e] = fmaxf(o[e], 0.f);
        else if (p.act == 2) o[e] = sc_gelu_t<true>(o[e]);
      }
      unsigned short* yrow = p.Y + (long long)m * p.ldy + n;
      const unsigned short* rrow = p.res ? p.res + (long long)m * p.ldr + n : nullptr;
      if (nv == CPT && ((uintptr_t)yrow % 8 == 0) && (rrow == nullptr || (uintptr_t)rrow % 8 == 0)) {
#pragma unroll
        for (int q4 = 0; q4 < CPT / 4; ++q4) {
          float* oo = o + 4 * q4;
          if (rrow) {
            const uint2 u = *(const uint2*)(rrow + 4 * q4);
            oo[0] += __uint_as_float(u.x << 16); oo[1] += __uint_as_float(u.x & 0xFFFF0000u);
            oo[2] += __uint_as_float(u.y << 16); oo[3] += __uint_as_float(u.y & 0xFFFF0000u);
          }
          *(uint2*)(yrow + 4 * q4) = make_uint2(sc_f32_to_bf16(oo[0]) | ((unsigned)sc_f32_to_bf16(oo[1]) << 16),
                                                sc_f32_to_bf16(oo[2]) | ((unsigned)sc_f32_to_bf16(oo[3]) << 16));
        }
      } else {
        for (int e = 0; e < nv; ++e) {
          float t = o[e];
          if (rrow) t += sc_bf16_to_f32(rrow[e]);
          yrow[e] = sc_f32_to_bf16(t);
        }
      }
    }
  }
}

}  // namespace

SC_API int sc_decode_linear(const void* x, long long ldx, int M, int K, const void* w, long long ldw, int N, const float* bias,
                            const float* ln_gamma, const float* ln_beta, float ln_eps, int act, const void* residual, long long ldr, void* y,
                            long long ldy, hipStream_t stream) {
  SC_CHECK_ARG(x && w && y, "sc_decode_linear: null pointer");
  SC_CHECK_ARG(M >= 1 && N >= 1 && K >= 32 && K % 32 == 0, "sc_decode_linear: bad sizes M=%d N=%d K=%d (K a multiple of 32)", M, N, K);
  SC_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N && ldx % 8 == 0 && ldw % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0,
               "sc_decode_linear: rows of x and w must be 16-byte aligned");
  SC_CHECK_ARG(residual == nullptr || ldr >= N, "sc_decode_linear: residual rows shorter than N");
  SC_CHECK_ARG(act >= 0 && act <= 2, "sc_decode_linear: act %d (0 none, 1 ReLU, 2 GELU)", act);
  SC_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "sc_decode_linear: LayerNorm weight and bias come together");
  SC_CHECK_ARG(ln_gamma == nullptr || K <= KS, "sc_decode_linear: the folded LayerNorm needs the whole row on chip (K <= %d, got %d)", KS, K);
  SC_CHECK_ARG((long long)sc_cdiv(M, 32) <= 65535, "sc_decode_linear: too many rows");
  SC_CHECK_ARG(ln_gamma == nullptr || ((uintptr_t)ln_gamma % 16 == 0 && (uintptr_t)ln_beta % 16 == 0), "sc_decode_linear: LayerNorm vectors must be 16-byte aligned");
  DecArgs a{};
  a.X = (const unsigned short*)x; a.W = (const unsigned short*)w; a.Y = (unsigned short*)y; a.bias = bias; a.gamma = ln_gamma; a.beta = ln_beta;
  a.res = (const unsigned short*)residual; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.act = act; a.eps = ln_eps;
  // 32 x 32 tiles (SC_DECODE_TR=2: 64 x 64 when the contraction fits one stage).
  // LDS: one stage (64 KB x TR) when the whole contraction fits it, else two; + gamma | beta for the folded LayerNorm
  static const int force_tr = [] { const char* e = getenv("SC_DECODE_TR"); return e ? atoi(e) : 0; }();  // (A/B knob: 1 / 2 force the tile size)
  // (64 x 64 tiles only on request: launched back to back the two tile sizes tie on the wide products, inside a decoding step -- every launch
  //  behind a different kernel, weights cold -- the 32 x 32 tiles, two workgroups per CU, win: 60.7 -> 58.7 ms per batch)
  const bool big = K <= KS && force_tr == 2;
  const int tr = big ? 2 : 1;
  const size_t smem = (size_t)(K > KS ? 2 : 1) * (size_t)tr * 2 * 16 * 32 * 32 * sizeof(unsigned short) + (ln_gamma ? 2 * 512 * sizeof(float) : 0);
  const void* fn = ln_gamma ? (big ? (const void*)decode_linear_kernel<1, 2> : (const void*)decode_linear_kernel<1, 1>)
                            : (big ? (const void*)decode_linear_kernel<0, 2> : (const void*)decode_linear_kernel<0, 1>);
  SC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  void* kargs[] = {&a};
  SC_HIP(hipLaunchKernel(fn, dim3(sc_cdiv(N, 32 * tr), sc_cdiv(M, 32 * tr)), dim3(256), kargs, smem, stream));
  return SC_OK;
}

// act(LayerNorm(x) W^T + b) + residual from the RAW rows x and the pre-scaled weight (see the top of the file):
//   w_scaled (N, K) = bf16(gamma[k] W[n][k]),  colsum (N) = sum_k w_scaled[n][k],  shift (N) = W beta + b.
SC_API int sc_decode_linear_lnstats(const void* x, long long ldx, int M, int K, const void* w_scaled, long long ldw, int N, const float* colsum,
                                    const float* shift, float ln_eps, int act, const void* residual, long long ldr, void* y, long long ldy,
                                    hipStream_t stream) {
  SC_CHECK_ARG(x && w_scaled && y && colsum && shift, "sc_decode_linear_lnstats: null pointer");
  SC_CHECK_ARG(M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && K <= KS, "sc_decode_linear_lnstats: bad sizes M=%d N=%d K=%d (K a multiple of 32, <= %d)", M, N, K, KS);
  SC_CHECK_ARG(ldx >= K && ldw >= K && ldy >= N && ldx % 8 == 0 && ldw % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w_scaled % 16 == 0,
               "sc_decode_linear_lnstats: rows of x and w must be 16-byte aligned");
  SC_CHECK_ARG(residual == nullptr || ldr >= N, "sc_decode_linear_lnstats: residual rows shorter than N");
  SC_CHECK_ARG(act >= 0 && act <= 2, "sc_decode_linear_lnstats: act %d (0 none, 1 ReLU, 2 GELU)", act);
  SC_CHECK_ARG((long long)sc_cdiv(M, 32) <= 65535, "sc_decode_linear_lnstats: too many rows");
  DecArgs a{};
  a.X = (const unsigned short*)x; a.W = (const unsigned short*)w_scaled; a.Y = (unsigned short*)y; a.bias = shift; a.colsum = colsum;
  a.res = (const unsigned short*)residual; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.act = act; a.eps = ln_eps;
  static const int force_tr = [] { const char* e = getenv("SC_DECODE_TR"); return e ? atoi(e) : 0; }();
  const bool big = force_tr == 2;  // (see sc_decode_linear)
  const int tr = big ? 2 : 1;
  const size_t smem = (size_t)tr * 2 * 16 * 32 * 32 * sizeof(unsigned short) + 2 * 64 * sizeof(float);
  const void* fn = big ? (const void*)decode_linear_kernel<2, 2> : (const void*)decode_linear_kernel<2, 1>;
  SC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  void* kargs[] = {&a};
  SC_HIP(hipLaunchKernel(fn, dim3(sc_cdiv(N, 32 * tr), sc_cdiv(M, 32 * tr)), dim3(256), kargs, smem, stream));
  return SC_OK;
}
