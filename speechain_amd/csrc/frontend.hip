// Fused acoustic frontend for gfx950:
//   waveform -> pre-emphasis -> reflect-padded framing -> windowed real DFT -> |.|^2 (-> sqrt)
//            -> mel filterbank -> clamp + log   (+ frame energy, + optional linear spectrogram)
// one kernel, the waveform is read from HBM once and the (log-)mel written once.
//
// Replaces the op chain of the reference (paths under /root/reference/):
//   speechain/module/frontend/speech2linear.py:142-250  (Conv1d pre-emphasis, torch.stft, power, energy, masking)
//   speechain/module/frontend/linear2mel.py:163-189     (bias-free Linear with the mel matrix, clamp, log)
//
// Arithmetic: the windowed DFT is evaluated as an exact-f32 matrix product on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32): frames(32 per wave) x n_fft  times  basis(n_fft x columns), where each
// 32-column tile holds [16 cos | 16 sin] columns of 16 consecutive bins, so re and im of one bin sit
// in lanes l and l^16 of the same accumulator register.  The mel stage is a banded (sparse) product on
// the VALU with one lane per frame.
//
// Work decomposition: one 256-thread workgroup = FT*32 consecutive frames of one utterance.  The
// waveform span of those frames is staged once in LDS (pre-emphasis, reflect padding and tail zeroing
// applied on the way in) with one pad word per hop so that the 32 frame rows of an MFMA A-operand hit
// 32 different banks.  Wave w owns frame tile (w % FT) and column group (w / FT).
#include "sc_common.h"
#include <cstdlib>

namespace {

constexpr int kMaxTiles = 7;  // accumulator tiles (16 VGPRs each) a wave carries per pass

struct FrontendArgs {
  const float* wave;
  const long long* wave_len;
  long long wave_stride;
  int T_w;
  int use_preemph;
  float preemph;
  int n_fft, krows, hop, pad, T;
  const float* basis;  // (krows, NT*32)
  int NT, NB;
  const float* melw;   // (n_mels, NB) or null
  const int* band_lo;
  const int* band_hi;
  int n_mels;
  int mag_spec;
  float clamp;
  int logging;
  float log_div;
  float log_scale;     // ln 2 / log_div
  float* out_feat;     // (B, T, n_mels) or null
  float* out_energy;   // (B, T) or null
  float* out_linear;   // (B, T, NB) or null
  float lin_clamp;
  int lin_logging;
  float lin_log_div;
  long long* feat_len; // (B,)
  int padflag;         // 1 when hop is even: one pad word per hop in the LDS sample image
  int PS, OS;          // row strides (floats) of the power image and the output staging image
  int off_P, off_O;    // LDS offsets (floats)
  int overlap;         // power image overlaps the sample image (single-pass configuration)
  int tiles_per_group, tiles_per_pass, npasses;
  const float* pfa;    // tables of the 16 x 25 two-stage DFT (n_fft = 400) or null
  unsigned hop_magic;  // ceil(2^20 / hop): (n * hop_magic) >> 20 == n / hop for n < 800
  const float* mel_image;  // 16-slot-per-filter weight image (sc_frontend_mel_image_init) or null
  int off_img, img_floats; // its LDS offset and size (floats)
};

__device__ __forceinline__ long long floordiv_ll(long long a, long long b) {
  long long q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// Stages the waveform span of frames [t0, t0 + FTF) in LDS: pre-emphasis, reflect padding and tail zeroing applied on the
// way in; one pad word per hop (padflag) so that consecutive frames start on different banks.
template <int FTF, int NTHR, int U>
__device__ __forceinline__ void stage_samples(const FrontendArgs& p, float* smem, int b, int t0, long long wlen, int tid) {
  const float* x = p.wave + (long long)b * p.wave_stride;
  const int count = (FTF - 1) * p.hop + p.krows;
  const int padded_len = p.T_w + 2 * p.pad;
  const int base = t0 * p.hop - p.pad;  // sample index of the span's first element (32-bit: T_w < 2^31)
  const int len = (int)(wlen < p.T_w ? wlen : p.T_w);
  const int limit = p.use_preemph ? len : p.T_w;  // the pre-emphasis path zeroes the tail beyond wave_len
  const float negc = p.use_preemph ? -p.preemph : 0.f;
  const float inv_hop = 1.f / (float)p.hop;
  // U loads in flight per lane
  for (int i0 = tid; i0 < count; i0 += NTHR * U) {
    float cur[U], prev[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + NTHR * u;
      int n = base + i;
      const bool inside = i < count && n + p.pad < padded_len;
      n = n < 0 ? -n : n;
      n = n >= p.T_w ? 2 * (p.T_w - 1) - n : n;
      const bool take = inside && n < limit;
      n = take ? n : 0;  // branch-free: always a valid address
      const float c = x[n];
      const float q = x[n > 0 ? n - 1 : 0];
      cur[u] = take ? c : 0.f;
      prev[u] = (p.use_preemph && take && n > 0) ? q : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + NTHR * u;
      if (i < count) smem[i + (int)(((float)i + 0.5f) * inv_hop) * p.padflag] = fmaf(negc, prev[u], cur[u]);  // + one pad word per completed hop
    }
  }
}

// value of lane l ^ 16, valid in the lanes with (l & 16) == 0: one v_permlane16_swap instead of an LDS bpermute
__device__ __forceinline__ float upper_row_partner(float v) {
  const unsigned u = __float_as_uint(v);
  return __uint_as_float(__builtin_amdgcn_permlane16_swap(u, u, false, false)[1]);
}

// Per-frame epilogue on the power image P (FTF rows, stride p.PS): banded mel + clamp/log with one lane per frame, frame
// energy, coalesced copy-out of the mel rows (through the O staging image) and of the optional linear spectrogram.
template <int FTF>
__device__ __forceinline__ void frontend_epilogue(const FrontendArgs& p, const float* P, float* O, int b, int t0, int flen,
                                                  int tid, int lane, int wid, bool mel_done = false) {
  const int f = lane & (FTF - 1);
  const int t = t0 + f;
  const bool live = t < flen;
  const float* prow = P + f * p.PS;

  if (p.out_feat && !mel_done) {
    for (int m = wid; m < p.n_mels; m += 4) {
      const int lo = p.band_lo[m], hi = p.band_hi[m];
      const float* wrow = p.melw + (long long)m * p.NB;
      float s = 0.f;
      for (int k = lo; k < hi; ++k) s = fmaf(prow[k], wrow[k], s);
      if (!live) s = 0.f;
      if (p.logging) s = logf(fmaxf(s, p.clamp)) / p.log_div;
      if (lane < FTF) O[f * p.OS + m] = s;
    }
  }
  if (p.out_energy && wid == 3 && lane < FTF && t < p.T) {
    float e = 0.f;
    for (int k = 0; k < p.NB; ++k) {
      const float v = prow[k];
      e += p.mag_spec ? v * v : v;
    }
    p.out_energy[(long long)b * p.T + t] = live ? sqrtf(fmaxf(e, 1e-10f)) : 0.f;
  }
  __syncthreads();

  const int nvalid = min(FTF, p.T - t0);
  if (p.out_feat) {
    float* dst = p.out_feat + ((long long)b * p.T + t0) * p.n_mels;
    const int total = nvalid * p.n_mels;
    const float inv_nm = 1.f / (float)p.n_mels;
    for (int e = tid; e < total; e += 256) {
      const int ff = (int)(((float)e + 0.5f) * inv_nm);  // e / n_mels (e < 2^13)
      const int m = e - ff * p.n_mels;
      dst[e] = O[ff * p.OS + m];
    }
  }
  if (p.out_linear) {
    float* dst = p.out_linear + ((long long)b * p.T + t0) * p.NB;
    const int total = nvalid * p.NB;
    for (int e = tid; e < total; e += 256) {
      const int ff = e / p.NB;
      const int k = e - ff * p.NB;
      float v = (t0 + ff < flen) ? P[ff * p.PS + k] : 0.f;
      if (p.lin_logging) v = logf(fmaxf(v, p.lin_clamp)) / p.lin_log_div;
      dst[e] = v;
    }
  }
}

template <int FT>
__global__ __launch_bounds__(256, 2) void frontend_kernel(FrontendArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int FTF = FT * 32;  // frames per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FTF;

  const long long wlen = p.wave_len[b];
  long long flen_ll = floordiv_ll(wlen + 2LL * p.pad - p.n_fft, p.hop) + 1;
  if (blockIdx.x == 0 && tid == 0) p.feat_len[b] = flen_ll;
  const int flen = (int)(flen_ll < 0 ? 0 : (flen_ll > p.T ? p.T : flen_ll));

  stage_samples<FTF, 256, 8>(p, smem, b, t0, wlen, tid);
  __syncthreads();

  // ---- windowed DFT on the f32 matrix cores ------------------------------------------------------
  const int ft = wid % FT;
  const int cg = wid / FT;
  const int h = lane >> 5;
  const int li = lane & 31;
  const int fstride = p.hop + p.padflag;
  const float* arow = smem + (ft * 32 + li) * fstride;
  const int NC = p.NT * 32;
  const int group_begin = cg * p.tiles_per_group;
  const int group_end = min(p.NT, group_begin + p.tiles_per_group);
  float* P = smem + p.off_P;

  for (int pass = 0; pass < p.npasses; ++pass) {
    const int tile0 = group_begin + pass * p.tiles_per_pass;
    int ntile = min(group_end, tile0 + p.tiles_per_pass) - tile0;
    ntile = __builtin_amdgcn_readfirstlane(ntile < 0 ? 0 : ntile);
    f32x16 acc[kMaxTiles];
#pragma unroll
    for (int t = 0; t < kMaxTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (ntile > 0) {
      const float* bcol = p.basis + (long long)h * NC + tile0 * 32 + li;
      // segment loop: inside one hop-long segment the LDS offset of sample k is k + seg*padflag
      const int seglen = p.padflag ? p.hop : p.krows;
      for (int seg0 = 0, seg = 0; seg0 < p.krows; seg0 += seglen, ++seg) {
        const int kend = min(p.krows, seg0 + seglen);
        const float* ap = arow + seg * p.padflag + h;
        float a_n = ap[seg0];
        float b_n[kMaxTiles];
        const float* bp = bcol + (long long)seg0 * NC;
#pragma unroll
        for (int t = 0; t < kMaxTiles; ++t) b_n[t] = (t < ntile) ? bp[t * 32] : 0.f;
        for (int k0 = seg0; k0 < kend; k0 += 2) {
          const float a_c = a_n;
          float b_c[kMaxTiles];
#pragma unroll
          for (int t = 0; t < kMaxTiles; ++t) b_c[t] = b_n[t];
          if (k0 + 2 < kend) {
            a_n = ap[k0 + 2];
            bp += 2 * NC;
#pragma unroll
            for (int t = 0; t < kMaxTiles; ++t)
              if (t < ntile) b_n[t] = bp[t * 32];
          }
#pragma unroll
          for (int t = 0; t < kMaxTiles; ++t)
            if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[t], acc[t], 0, 0, 0);
        }
      }
    }

    if (p.overlap) __syncthreads();  // every wave is done reading samples before P overwrites them
    // power: re^2 + im^2 with re in lane l (l&16 == 0) and im in lane l^16
#pragma unroll
    for (int t = 0; t < kMaxTiles; ++t) {
      if (t < ntile) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[t][r] * acc[t][r];
          v += upper_row_partner(v);
          if ((li & 16) == 0) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int bin = (tile0 + t) * 16 + li;
            P[(ft * 32 + row) * p.PS + bin] = p.mag_spec ? sqrtf(v) : v;
          }
        }
      }
    }
  }
  __syncthreads();

  frontend_epilogue<FTF>(p, P, smem + p.off_O, b, t0, flen, tid, lane, wid);
}


// ---- wide transforms (n_fft >= 512) as a matrix product of the library: frames -> sc_gemm (exact f32) -> finish -------------------
// frontend_kernel<FT> above reads its B operand (the basis) one float per lane and MFMA from L2: 32 TFLOP/s at n_fft = 1024 (1.35 ms for the
// FastSpeech2 batch), where the LDS-staged exact-f32 GEMM kernel of csrc/gemm.hip runs the same product at 76 TFLOP/s.  The three steps:
//   frontend_frames_kernel       frames (B T, n_fft): pre-emphasis, reflect padding, tail zeroing -- the sample image of stage_samples, written out
//   sc_gemm (f32)                spec (B T, 32 NT) = frames x basis  ([16 cos | 16 sin] column tiles, the window folded into the basis)
//   frontend_spec_finish_kernel  power / magnitude of 32 frames into the LDS image the shared epilogue reads (mel, log, energy, copy-out)
__global__ __launch_bounds__(256) void frontend_frames_kernel(FrontendArgs p, float* frames) {
  const int b = blockIdx.y, t = blockIdx.x;
  const float* x = p.wave + (long long)b * p.wave_stride;
  const long long wlen = p.wave_len[b];
  const int len = (int)(wlen < p.T_w ? wlen : p.T_w);
  const int limit = p.use_preemph ? len : p.T_w;
  const float negc = p.use_preemph ? -p.preemph : 0.f;
  float* dst = frames + ((long long)b * p.T + t) * p.krows;
  const int base = t * p.hop - p.pad;
  for (int k = threadIdx.x; k < p.krows; k += 256) {
    int n = base + k;
    const bool inside = k < p.n_fft;
    n = n < 0 ? -n : n;
    n = n >= p.T_w ? 2 * (p.T_w - 1) - n : n;
    const bool take = inside && n < limit;
    n = take ? n : 0;
    const float c = x[n], q = x[n > 0 ? n - 1 : 0];
    dst[k] = fmaf(negc, (p.use_preemph && take && n > 0) ? q : 0.f, take ? c : 0.f);
  }
}

__global__ __launch_bounds__(256) void frontend_spec_finish_kernel(FrontendArgs p, const float* spec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const long long wlen = p.wave_len[b];
  long long flen_ll = floordiv_ll(wlen + 2LL * p.pad - p.n_fft, p.hop) + 1;
  if (blockIdx.x == 0 && tid == 0) p.feat_len[b] = flen_ll;
  const int flen = (int)(flen_ll < 0 ? 0 : (flen_ll > p.T ? p.T : flen_ll));
  float* P = smem + p.off_P;
  const int NC = p.NT * 32;
  // column tile j of spec holds cos (re) of bins 16 j .. 16 j + 15 in its first 16 columns and sin (im) in the other 16
  for (int e = tid; e < 32 * p.NT * 16; e += 256) {
    const int f = e / (p.NT * 16), c = e - f * (p.NT * 16);
    const int tile = c >> 4, i = c & 15;
    float v = 0.f;
    if (t0 + f < p.T) {
      const float* srow = spec + ((long long)b * p.T + t0 + f) * NC + tile * 32;
      const float re = srow[i], im = srow[16 + i];
      v = re * re + im * im;
    }
    P[f * p.PS + c] = p.mag_spec ? sqrtf(v) : v;
  }
  __syncthreads();
  frontend_epilogue<32>(p, P, smem + p.off_O, b, t0, flen, tid, lane, wid);
}

// ---- the same in radix-4 form (n_fft a multiple of 4): a quarter of the product's flops ---------------------------------------------
//   x[4 m + r] (r = 0..3) are four real sequences of Q = n_fft / 4 samples; with Y_r their Q-point DFTs,
//     X[k] = sum_r e^{-2 pi i r k / n_fft} Y_r[k mod Q],       Y_r[Q - q] = conj(Y_r[q])  (real input: bins 0 .. Q / 2 suffice).
//   frontend_frames4_kernel       frames4 (4 B T, Q): row 4 f + r = window[4 m + r] x sample(4 m + r) of frame f (the window can no
//                                 longer be folded into the basis: the four sequences share it)
//   sc_gemm (f32)                 spec4 (4 B T, 32 NTq) = frames4 x basis_Q (the unwindowed Q-point basis of sc_frontend_basis_init)
//   frontend_spec_finish4_kernel  the twiddled sum over r, then power / magnitude into the image the shared epilogue reads
// 4 x Q x (Q / 2 + 1) x 2 multiply-adds per frame instead of n_fft x (n_fft / 2 + 1) x 2: 0.59 instead of 2.16 MFLOP at n_fft = 1024.
__global__ __launch_bounds__(256) void frontend_frames4_kernel(FrontendArgs p, const float* window, float* frames4) {
  const int b = blockIdx.y, t = blockIdx.x;
  const float* x = p.wave + (long long)b * p.wave_stride;
  const long long wlen = p.wave_len[b];
  const int len = (int)(wlen < p.T_w ? wlen : p.T_w);
  const int limit = p.use_preemph ? len : p.T_w;
  const float negc = p.use_preemph ? -p.preemph : 0.f;
  const int Q = p.n_fft >> 2;
  float* dst = frames4 + ((long long)b * p.T + t) * 4 * Q;
  const int base = t * p.hop - p.pad;
  for (int k = threadIdx.x; k < p.n_fft; k += 256) {
    int n = base + k;
    n = n < 0 ? -n : n;
    n = n >= p.T_w ? 2 * (p.T_w - 1) - n : n;
    const bool take = n < limit;
    n = take ? n : 0;
    const float c = x[n], q = x[n > 0 ? n - 1 : 0];
    const float v = fmaf(negc, (p.use_preemph && take && n > 0) ? q : 0.f, take ? c : 0.f);
    dst[(k & 3) * Q + (k >> 2)] = window[k] * v;
  }
}

// twiddle (4, NB, 2) f64: cos, sin of 2 pi r k / n_fft.  The sum over r runs in double: a bin 100 dB below the frame's energy is the small
// difference of four large terms, and float twiddles / float adds would put their own 6e-8 x |Y| on top of the transform's (measured on the
// fixture's tone-plus-floor utterance: 7.8e-3 in log-mel with a float sum against the reference's own 6.3e-3).
__global__ __launch_bounds__(256) void frontend_spec_finish4_kernel(FrontendArgs p, const float* spec4, const double* twiddle) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const long long wlen = p.wave_len[b];
  long long flen_ll = floordiv_ll(wlen + 2LL * p.pad - p.n_fft, p.hop) + 1;
  if (blockIdx.x == 0 && tid == 0) p.feat_len[b] = flen_ll;
  const int flen = (int)(flen_ll < 0 ? 0 : (flen_ll > p.T ? p.T : flen_ll));
  float* P = smem + p.off_P;
  const int Q = p.n_fft >> 2;
  const int NCq = ((Q / 2 + 1 + 15) / 16) * 32;  // columns of spec4: [16 cos | 16 sin] tiles of the Q-point transform's bins 0 .. Q / 2
  for (int e = tid; e < 32 * p.NT * 16; e += 256) {
    const int f = e / (p.NT * 16), k = e - f * (p.NT * 16);
    float v = 0.f;
    if (t0 + f < p.T && k < p.NB) {
      const int q = k % Q;
      const bool mirrored = q > Q / 2;
      const int qq = mirrored ? Q - q : q;
      const float* srow = spec4 + ((long long)b * p.T + t0 + f) * 4 * NCq + (qq >> 4) * 32 + (qq & 15);
      double re = 0.0, im = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the basis columns hold cos and -sin: Re Y_r and Im Y_r; the mirrored bins are the conjugates)
        const double yr = (double)srow[r * NCq], yi = mirrored ? -(double)srow[r * NCq + 16] : (double)srow[r * NCq + 16];
        const double ct = twiddle[(r * p.NB + k) * 2], st = twiddle[(r * p.NB + k) * 2 + 1];
        re += yr * ct + yi * st;  // (cos - i sin)(yr + i yi)
        im += yi * ct - yr * st;
      }
      v = (float)(re * re + im * im);
    }
    P[f * p.PS + k] = p.mag_spec ? sqrtf(v) : v;
  }
  __syncthreads();
  frontend_epilogue<32>(p, P, smem + p.off_O, b, t0, flen, tid, lane, wid);
}

// ---- n_fft = 400: Good-Thomas (prime-factor) two-stage DFT, 400 = 16 x 25 with gcd 1 -> no twiddle factors --------------
//   input index  n = (25 n1 + 16 n2) mod 400,   output index  k = (225 k1 + 176 k2) mod 400   (CRT maps)
//   stage 1: 25-point real-input DFTs along n2 for each n1 (k2 = 0..12 suffice: conjugate symmetry)
//   stage 2: 16-point complex DFTs along n1 for each k2
// both as exact-f32 MFMA products (v_mfma_f32_32x32x2_f32) with the tiny DFT matrices as the A operand (rows = output
// components) and the 32 frames of the tile as the N dimension -- 408 MFMAs per 32 frames instead of the 2592 of the
// direct 400 x 402 product, and FFT-like (short) accumulation chains.
//
// ONE WAVE per 32-frame tile, nothing but registers between the stages: a 32x32 accumulator holds row rho in lane half
// (rho >> 2) & 1, register (rho & 3) + 4 (rho >> 3).  The stage-1 matrix puts Re Y(k2) in half-0 row r = k2 and
// Im Y(k2) in the half-1 row of the same register, so register k2 of accumulator n1 IS the stage-2 B operand for
// (k = (n1, re|im), frame): lane half 0 supplies the real part, half 1 the imaginary part.  Stage 2 does the same with
// k1, so |Z|^2 is one v_permlane32_swap + add for two bins.  16 accumulators x 16 registers = 256 of the wave's 512.
// The 208 computed (k1, k2) values cover the 201 bins after folding k > 200 to 400 - k.  The power image (32 x 201,
// wave-private LDS, overlapping the dead sample image) feeds the banded mel product on the VALU, whose weights sit in
// LDS as a 16-slot-per-filter image prepared once by sc_frontend_mel_image_init.
constexpr int PFA_T25 = 26 * 32;   // [n2 (25 + zero pad row)][row rho]
constexpr int PFA_T16 = 32 * 32;   // [(n1, re|im)][row rho]
constexpr int PFA_WIN = 512;       // window, zero-padded to 400 samples, then to two 1 KB DMA granules
constexpr int PFA_FLOATS = PFA_T25 + PFA_T16 + PFA_WIN;
constexpr int PFA_PS = 201;        // power-image row stride: exactly the 201 bins (odd: conflict-free frame-major access)
constexpr int kMelSlots = 16;

typedef __attribute__((address_space(1))) const void* fe_gptr_t;
typedef __attribute__((address_space(3))) void* fe_lptr_t;

static inline int mel_image_floats(int n_mels) { return (n_mels * (kMelSlots + 2) + 2 + 255) & ~255; }

// image: [n_mels x 16] slot weights | [n_mels x 2] ints (first bin read, slots needed) | wide flag | pairs8
//   slot j of filter m = melw[m][lo' + j] for lo <= lo' + j < hi (else 0), lo' = min(lo, NB - 16): reads stay inside a row
//   wide   = some filter does not fit 16 slots (the kernel then reads the dense matrix)
//   pairs8 = number of leading filter pairs (2i, 2i+1) that both fit 8 slots
__global__ void mel_image_kernel(const float* melw, const int* band_lo, const int* band_hi, int n_mels, int NB, float* img, int total) {
  int* iimg = reinterpret_cast<int*>(img);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (i < n_mels * kMelSlots) {
      const int m = i >> 4, j = i & 15;
      const int lo = band_lo[m], hi = band_hi[m];
      const int k = max(min(lo, NB - kMelSlots), 0) + j;
      img[i] = (k >= lo && k < hi && k < NB) ? melw[(long long)m * NB + k] : 0.f;
    } else if (i < n_mels * (kMelSlots + 2)) {
      const int e = i - n_mels * kMelSlots;
      const int m = e >> 1;
      const int lo2 = max(min(band_lo[m], NB - kMelSlots), 0);
      iimg[i] = (e & 1) ? band_hi[m] - lo2 : lo2;
    } else if (i == n_mels * (kMelSlots + 2)) {
      int wide = NB < kMelSlots ? 1 : 0;
      for (int m = 0; m < n_mels; ++m) wide |= (band_hi[m] - max(min(band_lo[m], NB - kMelSlots), 0) > kMelSlots) ? 1 : 0;
      iimg[i] = wide;
    } else if (i == n_mels * (kMelSlots + 2) + 1) {
      int pairs = 0;
      for (int m = 0; m + 1 < n_mels; m += 2) {
        const int n0 = band_hi[m] - max(min(band_lo[m], NB - kMelSlots), 0);
        const int n1 = band_hi[m + 1] - max(min(band_lo[m + 1], NB - kMelSlots), 0);
        if (n0 > 8 || n1 > 8) break;
        ++pairs;
      }
      iimg[i] = pairs;
    } else {
      img[i] = 0.f;
    }
  }
}

// sum over NS slots of power x weight; the weight row is wave-half-uniform (LDS broadcast reads)
template <int NS>
__device__ __forceinline__ float mel_dot(const float* prow, const float* wrow) {
  float pv[NS], wv[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) pv[j] = prow[j];
#pragma unroll
  for (int j = 0; j < NS; j += 4) {
    const float4 w4 = *reinterpret_cast<const float4*>(wrow + j);
    wv[j] = w4.x; wv[j + 1] = w4.y; wv[j + 2] = w4.z; wv[j + 3] = w4.w;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) s = fmaf(pv[j], wv[j], s);
  return s;
}

// clamp + log of a mel value: v_log_f32 (log2, ~1 ulp) times ln2 / log_div -- far inside the 1e-3 log-mel budget
__device__ __forceinline__ float mel_log(const FrontendArgs& p, float s) {
  return p.logging ? __log2f(fmaxf(s, p.clamp)) * p.log_scale : s;
}

// four filter pairs (m0, m0 + 2, m0 + 4, m0 + 6; lane half h takes filter m + h of each pair) at a time: first every band
// record, then every power / weight read, then the arithmetic -- two LDS latencies per group instead of per filter.
// Only pairs below m_end are stored (the caller's other loop owns the rest).
template <int NS>
__device__ __forceinline__ void mel_group(const FrontendArgs& p, const float* prow0, const float* img, const int* bands, float* orow,
                                          int m0, int m_end, int h, bool live, bool valid) {
  int lo2[4], mm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + 2 * q + h;
    mm[q] = m < p.n_mels ? m : p.n_mels - 1;
    lo2[q] = bands[2 * mm[q]];
  }
  float s[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] = mel_dot<NS>(prow0 + lo2[q], img + mm[q] * kMelSlots);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + 2 * q + h;
    const float v = mel_log(p, live ? s[q] : 0.f);
    if (valid && m0 + 2 * q < m_end && m < p.n_mels) orow[m] = v;
  }
}

// One 256-sample chunk of the sample image per call; lane l owns elements 4l .. 4l+3 of the chunk.  A chunk that lies
// wholly inside the waveform (no reflection, no tail) takes two coalesced loads per lane; any other chunk evaluates the
// padding rules per element.  Loads only -- the caller stores once every chunk's loads are in flight.
struct ChunkRegs {
  float4 cur;
  float4 prev;  // prev.x only on the fast path (the other predecessors are cur.x .. cur.z)
};

__device__ __forceinline__ ChunkRegs chunk_load(const FrontendArgs& p, const float* x, int base, int i0, int count, int limit,
                                                bool fast, int lane) {
  ChunkRegs r;
  const int i = i0 + 4 * lane;
  if (fast) {
    const float* src = x + base + i;
    r.cur = *reinterpret_cast<const float4*>(src);
    r.prev.x = p.use_preemph ? src[-1] : 0.f;
    r.prev.y = r.prev.z = r.prev.w = 0.f;
  } else {
    const int padded_len = p.T_w + 2 * p.pad;
    float c[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int n = base + i + k;
      const bool inside = i + k < count && n + p.pad < padded_len;
      n = n < 0 ? -n : n;
      n = n >= p.T_w ? 2 * (p.T_w - 1) - n : n;
      const bool take = inside && n < limit;
      n = take ? n : 0;
      const float cv = x[n];
      const float qv = x[n > 0 ? n - 1 : 0];
      c[k] = take ? cv : 0.f;
      q[k] = (p.use_preemph && take && n > 0) ? qv : 0.f;
    }
    r.cur = make_float4(c[0], c[1], c[2], c[3]);
    r.prev = make_float4(q[0], q[1], q[2], q[3]);
  }
  return r;
}

template <int HOP, bool MAG>
__global__ __launch_bounds__(64) void frontend_pfa400_kernel(FrontendArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const int h = lane >> 5, f = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 32;
  const int hop = HOP > 0 ? HOP : p.hop;
  const int padflag = (hop % 2 == 0) ? 1 : 0;
  const unsigned hop_magic = HOP > 0 ? ((1u << 20) + HOP - 1) / HOP : p.hop_magic;

  const long long wlen = p.wave_len[b];
  long long flen_ll = floordiv_ll(wlen + 2LL * p.pad - p.n_fft, hop) + 1;
  if (blockIdx.x == 0 && lane == 0) p.feat_len[b] = flen_ll;
  const int flen = (int)(flen_ll < 0 ? 0 : (flen_ll > p.T ? p.T : flen_ll));
  const int t = t0 + f;
  const bool live = t < flen, valid = t < p.T;

  const float* T25 = p.pfa;
  const float* T16 = T25 + PFA_T25;
  const float* gwtab = T16 + PFA_T16;
  float* P = smem;                  // power image; the sample image lives here first
  float* img = smem + p.off_img;    // mel slot image
  float* wtab = img + p.img_floats; // window in operand order: [(n1, s)][h]
  float* t16 = wtab + PFA_WIN;      // stage-2 matrix (read into registers only once the stage-1 operands are dead)

  if (t0 >= flen) {
    // every frame of the tile lies beyond the utterance: all-zero power spectrum (speech2linear.py:226-228)
    if (p.out_feat && valid) {
      const float v = mel_log(p, 0.f);
      float* orow = p.out_feat + ((long long)b * p.T + t) * p.n_mels;
      for (int m = h; m < p.n_mels; m += 2) orow[m] = v;
    }
    if (p.out_energy && h == 0 && valid) p.out_energy[(long long)b * p.T + t] = 0.f;
    if (p.out_linear) {
      const int total = min(32, p.T - t0) * p.NB;
      float* dst = p.out_linear + ((long long)b * p.T + t0) * p.NB;
      const float v = p.lin_logging ? logf(fmaxf(0.f, p.lin_clamp)) / p.lin_log_div : 0.f;
      for (int e = lane; e < total; e += 64) dst[e] = v;
    }
    return;
  }

  // tables into LDS by DMA (no registers, nothing waits until the samples are in as well)
  for (int c = 0; c < p.img_floats; c += 256)
    __builtin_amdgcn_global_load_lds((fe_gptr_t)(p.mel_image + c + lane * 4), (fe_lptr_t)(img + c), 16, 0, 0);
#pragma unroll
  for (int c = 0; c < PFA_WIN; c += 256)
    __builtin_amdgcn_global_load_lds((fe_gptr_t)(gwtab + c + lane * 4), (fe_lptr_t)(wtab + c), 16, 0, 0);
#pragma unroll
  for (int c = 0; c < PFA_T16; c += 256)
    __builtin_amdgcn_global_load_lds((fe_gptr_t)(T16 + c + lane * 4), (fe_lptr_t)(t16 + c), 16, 0, 0);
  float a1[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) a1[s] = T25[(2 * s + h) * 32 + f];

  // ---- sample image: pre-emphasis, reflect padding and tail zeroing applied on the way in, one pad word per hop ----
  {
    const float* x = p.wave + (long long)b * p.wave_stride;
    const int count = 31 * hop + 400;
    const int base = t0 * hop - p.pad;
    const int len = (int)(wlen < p.T_w ? wlen : p.T_w);
    const int limit = p.use_preemph ? len : p.T_w;
    const float negc = p.use_preemph ? -p.preemph : 0.f;
    const bool aligned = (hop % 4 == 0) && ((reinterpret_cast<unsigned long long>(x + base) & 15ull) == 0);
    // 256-sample chunks a tile spans (count <= 6144: checked on the host); every chunk's loads are in flight together.
    // Whether a chunk may take the two-load path depends on T_w only, so nothing waits for the wave_len load; the
    // pre-emphasis path's tail zeroing (n >= wave_len) is applied to the loaded values afterwards.
    constexpr int NCH = HOP > 0 ? (31 * HOP + 400 + 255) / 256 : 24;
    ChunkRegs r[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int i0 = u * 256;
      if (i0 < count) {
        const bool fast = aligned && base + i0 >= 1 && base + i0 + 256 <= p.T_w;
        r[u] = chunk_load(p, x, base, i0, count, limit, fast, lane);
        if (fast) {
          r[u].prev.y = r[u].cur.x; r[u].prev.z = r[u].cur.y; r[u].prev.w = r[u].cur.z;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int i0 = u * 256;
      if (i0 < count) {
        const int i = i0 + 4 * lane;
        const int pc = (int)(((unsigned)i * hop_magic) >> 20) * padflag;
        float* dst = smem + i + pc;
        float y0 = fmaf(negc, r[u].prev.x, r[u].cur.x), y1 = fmaf(negc, r[u].prev.y, r[u].cur.y);
        float y2 = fmaf(negc, r[u].prev.z, r[u].cur.z), y3 = fmaf(negc, r[u].prev.w, r[u].cur.w);
        const bool fast = aligned && base + i0 >= 1 && base + i0 + 256 <= p.T_w;
        if (fast && base + i0 + 256 > limit) {  // (wave-uniform) a two-load chunk reaching into the zeroed tail
          const int n = base + i;               // no reflection here: n is the sample index
          y0 = n < limit ? y0 : 0.f; y1 = n + 1 < limit ? y1 : 0.f; y2 = n + 2 < limit ? y2 : 0.f; y3 = n + 3 < limit ? y3 : 0.f;
        }
        if (hop % 4 == 0) {  // the four elements sit inside one hop
          dst[0] = y0; dst[1] = y1; dst[2] = y2; dst[3] = y3;
        } else {
          smem[i + (int)(((unsigned)i * hop_magic) >> 20) * padflag] = y0;
          smem[i + 1 + (int)(((unsigned)(i + 1) * hop_magic) >> 20) * padflag] = y1;
          smem[i + 2 + (int)(((unsigned)(i + 2) * hop_magic) >> 20) * padflag] = y2;
          smem[i + 3 + (int)(((unsigned)(i + 3) * hop_magic) >> 20) * padflag] = y3;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA'd tables have landed

  // ---- stage 1: Y(n1, k2) for all 16 n1 ----
  f32x16 acc[16];
  {
    const int fs = hop + padflag;
    const float* frow = smem + f * fs;
    const float* xb = frow + h * 16;
    const float* wt = wtab + h;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    // Scheduling rules of a LONE wave on its SIMD (scripts/mb/mfma_overlap.hip, cycles per 32x32x2 f32 MFMA: alone 64,
    // + 1 VALU op 80.5, + 4 VALU ops 92.5, + 2 LDS reads 67.5, + 12 SALU ops 67): the wave's own VALU work does not
    // overlap its MFMAs and every MFMA -> VALU -> MFMA switch costs ~12 cycles, LDS reads and SALU are nearly free.
    // Hence: two interleaved accumulation chains (n1, n1 + 1) so that no MFMA waits for its predecessor; between the
    // MFMAs of chain pair P only the LDS reads of pair P + 2; the window x sample products of pair P + 1 in ONE burst
    // of 13 packed multiplies before the pair; no per-lane address arithmetic (one base pointer per value of
    // pos(h = 1) - pos(h = 0): 16 normally, 17 across a hop's pad word, wrapd across the mod-400 wrap).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xs[2][13], ws[2][13], pr[2][13];
    constexpr bool kConstWrap = HOP > 0 && (384 / (HOP > 0 ? HOP : 1) == 399 / (HOP > 0 ? HOP : 1));
    const int wrapd = -384 - (HOP > 0 ? (384 / (HOP > 0 ? HOP : 1)) * ((HOP % 2 == 0) ? 1 : 0) : 0);
    const float* xb17 = frow + h * 17;
    const float* xbw = frow + h * wrapd;
    struct XW { float x, w; };
    auto load_step = [&](int n0, int s, int d) -> XW {
      float xo, wo;
      const int n1 = n0 + d;
      // n = (25 n1 + 16 n2) mod 400 with n2 = 2s + h for both h; the padded k-step (n2 = 25) multiplies a zero matrix
      // column (its sample is read at the h = 0 position).  pos adds one pad word per completed hop.
      int na = 25 * n1 + 32 * s, nb = na + 16;
      na = na >= 400 ? na - 400 : na;
      nb = nb >= 400 ? nb - 400 : nb;
      const int pa = na + (int)(((unsigned)na * hop_magic) >> 20) * padflag;
      const int pb = nb + (int)(((unsigned)nb * hop_magic) >> 20) * padflag;
      const int delta = pb - pa;
      if (2 * s + 1 >= 25) xo = frow[pa];
      else if (HOP > 0 && delta == 16) xo = xb[pa];
      else if (HOP > 0 && delta == 17) xo = xb17[pa];
      else if (kConstWrap && delta == wrapd) xo = xbw[pa];
      else xo = frow[h ? pb : pa];
      wo = wt[(n1 * 13 + s) * 2];
      return {xo, wo};
    };
    auto load_pair = [&](int n0, f32x2* xo, f32x2* wo) {
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const XW e0 = load_step(n0, s, 0), e1 = load_step(n0, s, 1);
        xo[s].x = e0.x; wo[s].x = e0.w;
        xo[s].y = e1.x; wo[s].y = e1.w;
      }
    };
    load_pair(0, xs[0], ws[0]);
    load_pair(2, xs[1], ws[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      const int c = P & 1;
      // burst: products of this pair (its operands were read during pair P - 2)
#pragma unroll
      for (int s = 0; s < 13; ++s) pr[c][s] = ws[c][s] * xs[c][s];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 13; ++s) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int n1 = 2 * P + d;
          acc[n1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], d ? pr[c][s].y : pr[c][s].x, s == 0 ? zero : acc[n1], 0, 0, 0);
          if (P + 2 < 8) {
            const XW e = load_step(2 * (P + 2), s, d);
            if (d) { xs[c][s].y = e.x; ws[c][s].y = e.w; } else { xs[c][s].x = e.x; ws[c][s].x = e.w; }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- stage 2 + power: Z(k1, k2) = sum_n1 Y(n1, k2) w16^(n1 k1); lane half 0 ends with k1 = r, half 1 with k1 = r + 8 ----
  // (one wave, LDS operations retire in order: the power rows may overwrite the sample image from here on)
  {
    float a2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a2[s] = t16[(2 * s + h) * 32 + f];
    float* prow = P + f * PFA_PS;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    // Chain pairs (k2, k2 + 1): 32 MFMAs back to back, then ONE burst of packed VALU work for the previous pair
    // (squares, half swap, sums, store addresses) -- see the scheduling rules above.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int hoff = h;
    auto power_chain = [&](const f32x16& zz, int k2) {
      f32x2 sq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x2 v = {zz[2 * q], zz[2 * q + 1]};
        sq[q] = v * v;  // half 0: re^2, half 1: im^2
      }
      f32x2 pw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // registers r = 2q, 2q + 1 against r + 8: after the swap the lower half holds both addends of k1 = r, the upper
        // half both addends of k1 = r + 8
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq[q].x), __float_as_uint(sq[q + 4].x), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq[q].y), __float_as_uint(sq[q + 4].y), false, false);
        const f32x2 lo = {__uint_as_float(s0[0]), __uint_as_float(s1[0])}, hi = {__uint_as_float(s0[1]), __uint_as_float(s1[1])};
        pw[q] = lo + hi;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int b0 = (225 * r + 176 * k2) % 400, b1 = (225 * (r + 8) + 176 * k2) % 400;
        b0 = b0 > 200 ? 400 - b0 : b0;
        b1 = b1 > 200 ? 400 - b1 : b1;
        const float v = (r & 1) ? pw[r >> 1].y : pw[r >> 1].x;
        prow[b0 + hoff * (b1 - b0)] = MAG ? sqrtf(v) : v;
      }
    };
    f32x16 z[2][2];
#pragma unroll
    for (int k0 = 0; k0 < 14; k0 += 2) {
      const int cur = (k0 >> 1) & 1, prv = cur ^ 1;
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (k0 + d < 13) z[cur][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], acc[s][k0 + d], s == 0 ? zero : z[cur][d], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (k0 >= 2) {
        power_chain(z[prv][0], k0 - 2);
        power_chain(z[prv][1], k0 - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    power_chain(z[0][0], 12);  // k2 = 12 was the lone chain of the last pair
  }

  // ---- mel / energy / linear outputs of frame t0 + f ----
  const float* prow = P + f * PFA_PS;
  if (p.out_feat) {
    float* orow = p.out_feat + ((long long)b * p.T + (valid ? t : 0)) * p.n_mels;
    const int* bands = reinterpret_cast<const int*>(img + p.n_mels * kMelSlots);
    const bool slots = p.mel_image != nullptr && bands[2 * p.n_mels] == 0;
    if (slots) {
      const int split = 2 * bands[2 * p.n_mels + 1];
      for (int m0 = 0; m0 < split; m0 += 8) mel_group<8>(p, prow, img, bands, orow, m0, split, h, live, valid);
      for (int m0 = split; m0 < p.n_mels; m0 += 8) mel_group<16>(p, prow, img, bands, orow, m0, p.n_mels, h, live, valid);
    } else {
      for (int m = h; m < p.n_mels; m += 2) {  // dense-matrix loop (filters wider than 16 bins, or no image)
        const int lo = p.band_lo[m], hi = p.band_hi[m];
        const float* wrow = p.melw + (long long)m * p.NB;
        float s = 0.f;
        for (int k = lo; k < hi; ++k) s = fmaf(prow[k], wrow[k], s);
        s = mel_log(p, live ? s : 0.f);
        if (valid) orow[m] = s;
      }
    }
  }
  if (p.out_energy) {
    float e = 0.f;
    for (int k = h; k < p.NB; k += 2) {
      const float v = prow[k];
      e += MAG ? v * v : v;
    }
    e += __shfl_xor(e, 32, 64);
    if (h == 0 && valid) p.out_energy[(long long)b * p.T + t] = live ? sqrtf(fmaxf(e, 1e-10f)) : 0.f;
  }
  if (p.out_linear) {
    const int nvalid = min(32, p.T - t0);
    float* dst = p.out_linear + ((long long)b * p.T + t0) * p.NB;
    const int total = nvalid * p.NB;
    const float inv_nb = 1.f / (float)p.NB;
    for (int e = lane; e < total; e += 64) {
      const int ff = (int)(((float)e + 0.5f) * inv_nb);
      const int k = e - ff * p.NB;
      float v = (t0 + ff < flen) ? P[ff * PFA_PS + k] : 0.f;
      if (p.lin_logging) v = logf(fmaxf(v, p.lin_clamp)) / p.lin_log_div;
      dst[e] = v;
    }
  }
}

// tables of the two-stage DFT (row rho of a 32-row MFMA tile: lane half (rho >> 2) & 1, register (rho & 3) + 4 (rho >> 3))
__global__ void pfa400_tables_kernel(const float* window, int win_length, float* out) {
  float* T25 = out;
  float* T16 = T25 + PFA_T25;
  float* wtab = T16 + PFA_T16;
  const int left = (400 - win_length) / 2;
  // window in stage-1 operand order: entry ((n1 * 13 + s) * 2 + h) = window[(25 n1 + 16 (2s + h)) mod 400], 0 for the pad step
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PFA_WIN; i += gridDim.x * blockDim.x) {
    const int hh = i & 1, j = i >> 1;
    const int n1 = j / 13, n2 = 2 * (j % 13) + hh;
    float v = 0.f;
    if (n1 < 16 && n2 < 25) {
      const int n = (25 * n1 + 16 * n2) % 400;
      v = (n >= left && n < left + win_length) ? window[n - left] : 0.f;
    }
    wtab[i] = v;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * 32; i += gridDim.x * blockDim.x) {
    const int k = i >> 5, rho = i & 31;
    const int half = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    double s, c;
    if (k < 26) {  // T25[n2 = k][rho]: Re Y(k2 = r) = sum x cos, Im Y(k2 = r) = -sum x sin
      float v = 0.f;
      if (k < 25 && r < 13) {
        sincospi(2.0 * ((k * r) % 25) / 25.0, &s, &c);
        v = (float)(half ? -s : c);
      }
      T25[i] = v;
    }
    {  // T16[(n1, ri) = k][rho]: Re Z(k1 = r) = Yre c + Yim s, Im Z(k1 = r) = Yim c - Yre s
      const int n1 = k >> 1, ri = k & 1;
      sincospi(2.0 * ((n1 * r) % 16) / 16.0, &s, &c);
      T16[i] = (float)(half ? (ri ? c : -s) : (ri ? s : c));
    }
  }
}

// basis[k][tile*32 + j]: j < 16 -> w[k]*cos(2 pi k bin / n_fft), j >= 16 -> -w[k]*sin(...), bin = tile*16 + (j & 15);
// zero for bin >= NB and for the padding row k >= n_fft.  The window is centred in n_fft (torch.stft).
__global__ void basis_kernel(const float* window, int win_length, int n_fft, int krows, int NT, int NB, float* basis) {
  const int NC = NT * 32;
  const long long total = (long long)krows * NC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i / NC);
    const int c = (int)(i - (long long)k * NC);
    const int j = c & 31;
    const int bin = (c >> 5) * 16 + (j & 15);
    float v = 0.f;
    const int left = (n_fft - win_length) / 2;
    if (k < n_fft && bin < NB && k >= left && k < left + win_length) {
      const long long m = ((long long)k * bin) % n_fft;
      double s, c2;
      sincospi(2.0 * (double)m / (double)n_fft, &s, &c2);
      const double w = (double)window[k - left];
      v = (float)((j < 16) ? w * c2 : -w * s);
    }
    basis[i] = v;
  }
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------

SC_API int sc_frontend_num_tiles(int n_fft) { return (n_fft / 2 + 1 + 15) / 16; }

static long long direct_basis_floats(int n_fft) {
  const long long krows = (n_fft + 1) & ~1;
  return krows * sc_frontend_num_tiles(n_fft) * 32;
}

SC_API long long sc_frontend_basis_floats(int n_fft) { return direct_basis_floats(n_fft) + (n_fft == 400 ? PFA_FLOATS : 0); }

SC_API int sc_frontend_basis_init(const float* window, int win_length, int n_fft, int hop, float* basis, hipStream_t stream) {
  SC_CHECK_ARG(window && basis, "sc_frontend_basis_init: null pointer");
  SC_CHECK_ARG(n_fft >= 2 && win_length >= 1 && win_length <= n_fft, "sc_frontend_basis_init: need 1 <= win_length <= n_fft, got %d, %d", win_length, n_fft);
  const int NT = sc_frontend_num_tiles(n_fft);
  const int krows = (n_fft + 1) & ~1;
  hipLaunchKernelGGL(basis_kernel, dim3(256), dim3(256), 0, stream, window, win_length, n_fft, krows, NT, n_fft / 2 + 1, basis);
  SC_LAUNCH_CHECK("basis_kernel");
  if (n_fft == 400) {
    SC_CHECK_ARG(hop >= 1, "sc_frontend_basis_init: hop must be positive");
    hipLaunchKernelGGL(pfa400_tables_kernel, dim3(4), dim3(256), 0, stream, window, win_length, basis + direct_basis_floats(n_fft));
    SC_LAUNCH_CHECK("pfa400_tables_kernel");
  }
  return SC_OK;
}

SC_API long long sc_frontend_mel_image_floats(int n_mels) { return n_mels >= 1 ? mel_image_floats(n_mels) : 0; }

SC_API int sc_frontend_mel_image_init(const float* melw, const int* band_lo, const int* band_hi, int n_mels, int n_bins, float* image,
                                      hipStream_t stream) {
  SC_CHECK_ARG(melw && band_lo && band_hi && image, "sc_frontend_mel_image_init: null pointer");
  SC_CHECK_ARG(n_mels >= 1 && n_bins >= 1, "sc_frontend_mel_image_init: bad sizes n_mels=%d n_bins=%d", n_mels, n_bins);
  const int total = mel_image_floats(n_mels);
  hipLaunchKernelGGL(mel_image_kernel, dim3(sc_cdiv(total, 256)), dim3(256), 0, stream, melw, band_lo, band_hi, n_mels, n_bins, image, total);
  SC_LAUNCH_CHECK("mel_image_kernel");
  return SC_OK;
}

SC_API int sc_frontend_logmel_fwd(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride,
                                  int use_preemph, float preemph, int n_fft, int hop, int center, const float* basis,
                                  const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                                  float clamp, int logging, float log_div, float* out_feat, float* out_energy,
                                  float* out_linear, float lin_clamp, int lin_logging, float lin_log_div,
                                  long long* feat_len, int T, const float* mel_image, hipStream_t stream) {
  SC_CHECK_ARG(wave && wave_len && basis && feat_len, "sc_frontend_logmel_fwd: null pointer");
  SC_CHECK_ARG(B >= 1 && T_w >= 1 && n_fft >= 2 && hop >= 1, "sc_frontend_logmel_fwd: bad sizes B=%d T_w=%d n_fft=%d hop=%d", B, T_w, n_fft, hop);
  SC_CHECK_ARG(wave_stride >= T_w, "sc_frontend_logmel_fwd: wave_stride %lld < T_w %d", wave_stride, T_w);
  const int pad = center ? n_fft / 2 : 0;
  SC_CHECK_ARG(!center || pad < T_w, "sc_frontend_logmel_fwd: reflect padding %d needs a longer waveform (%d)", pad, T_w);
  SC_CHECK_ARG(T_w + 2 * pad >= n_fft, "sc_frontend_logmel_fwd: waveform (%d) shorter than one frame (%d)", T_w + 2 * pad, n_fft);
  const int T_expect = (T_w + 2 * pad - n_fft) / hop + 1;
  SC_CHECK_ARG(T == T_expect, "sc_frontend_logmel_fwd: T=%d but the padded batch holds %d frames", T, T_expect);
  SC_CHECK_ARG(out_feat == nullptr || (melw && band_lo && band_hi && n_mels >= 1), "sc_frontend_logmel_fwd: mel output needs melw/bands");
  SC_CHECK_ARG(out_feat || out_linear || out_energy, "sc_frontend_logmel_fwd: no output requested");

  FrontendArgs a{};
  a.wave = wave; a.wave_len = wave_len; a.wave_stride = wave_stride; a.T_w = T_w;
  a.use_preemph = use_preemph; a.preemph = preemph;
  a.n_fft = n_fft; a.krows = (n_fft + 1) & ~1; a.hop = hop; a.pad = pad; a.T = T;
  a.basis = basis; a.NT = sc_frontend_num_tiles(n_fft); a.NB = n_fft / 2 + 1;
  a.melw = melw; a.band_lo = band_lo; a.band_hi = band_hi; a.n_mels = out_feat ? n_mels : 0;
  a.mag_spec = mag_spec; a.clamp = clamp; a.logging = logging; a.log_div = log_div;
  a.log_scale = (float)(0.69314718055994530942 / (double)log_div);
  a.out_feat = out_feat; a.out_energy = out_energy; a.out_linear = out_linear;
  a.lin_clamp = lin_clamp; a.lin_logging = lin_logging; a.lin_log_div = lin_log_div;
  a.feat_len = feat_len;
  a.padflag = (hop % 2 == 0) ? 1 : 0;
  a.hop_magic = ((1u << 20) + (unsigned)hop - 1) / (unsigned)hop;
  a.mel_image = nullptr;

  // frames per workgroup: 64 (2 frame tiles x 2 column groups) while one pass of <= 7 tiles per wave covers all
  // bins, else 32 (1 frame tile x 4 column groups, several passes)
  const int FT = (sc_cdiv(a.NT, 2) <= kMaxTiles) ? 2 : 1;
  const int FTF = FT * 32;
  const int groups = 4 / FT;
  a.tiles_per_group = sc_cdiv(a.NT, groups);
  a.npasses = sc_cdiv(a.tiles_per_group, kMaxTiles);
  a.tiles_per_pass = sc_cdiv(a.tiles_per_group, a.npasses);
  a.PS = a.NT * 16 + 1;
  a.OS = (n_mels > 0 ? n_mels : 1) | 1;
  const int count = (FTF - 1) * hop + a.krows;
  const int sample_floats = count + count / hop + 8;
  const int P_floats = FTF * a.PS;
  const int O_floats = out_feat ? FTF * a.OS : 0;
  a.overlap = (a.npasses == 1) ? 1 : 0;
  if (a.overlap) {
    a.off_P = 0;
    a.off_O = (sample_floats > P_floats ? sample_floats : P_floats);
  } else {
    a.off_P = sample_floats;
    a.off_O = sample_floats + P_floats;
  }
  a.off_O = (a.off_O + 3) & ~3;
  const size_t smem = (size_t)(a.off_O + O_floats) * sizeof(float);
  SC_CHECK_ARG(smem <= 160 * 1024, "sc_frontend_logmel_fwd: n_fft=%d hop=%d needs %zu B of LDS (> 160 KiB)", n_fft, hop, smem);

  if (n_fft == 400 && !getenv("SC_FRONTEND_DIRECT")) {
    // two-stage 16 x 25 DFT, one wave per 32 frames: sample image / power image (32 x 201) | mel slot image | window
    a.pfa = basis + direct_basis_floats(n_fft);
    const int count32 = 31 * hop + a.krows;
    const int nm = out_feat ? n_mels : 0;
    a.mel_image = nm > 0 ? mel_image : nullptr;
    a.img_floats = a.mel_image ? mel_image_floats(nm) : 0;
    a.off_img = (32 * PFA_PS + 3) & ~3;
    const size_t smem_pfa = (size_t)(a.off_img + a.img_floats + PFA_WIN + PFA_T16) * sizeof(float);
    const int span32 = (count32 + 255) & ~255;  // the sample image is filled in whole 256-sample chunks
    if (span32 + span32 / hop + 8 <= 32 * PFA_PS && count32 <= 24 * 256 && smem_pfa <= 64 * 1024) {
      auto kern = hop == 160 ? (mag_spec ? frontend_pfa400_kernel<160, true> : frontend_pfa400_kernel<160, false>)
                             : (mag_spec ? frontend_pfa400_kernel<0, true> : frontend_pfa400_kernel<0, false>);
      SC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_pfa));
      hipLaunchKernelGGL(kern, dim3(sc_cdiv(T, 32), B), dim3(64), smem_pfa, stream, a);
      SC_LAUNCH_CHECK("frontend_pfa400_kernel");
      return SC_OK;
    }
  }
  dim3 grid(sc_cdiv(T, FTF), B);
  if (FT == 2) {
    SC_HIP(hipFuncSetAttribute((const void*)frontend_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(frontend_kernel<2>, grid, dim3(256), smem, stream, a);
  } else {
    SC_HIP(hipFuncSetAttribute((const void*)frontend_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(frontend_kernel<1>, grid, dim3(256), smem, stream, a);
  }
  SC_LAUNCH_CHECK("frontend_kernel");
  return SC_OK;
}


// ---- wide transforms through the library's f32 GEMM (see frontend_frames_kernel) ---------------------------------------------------
// frames (B T, krows) f32 with krows = n_fft rounded up to even: what sc_gemm multiplies with the basis of sc_frontend_basis_init.
SC_API int sc_frontend_frames(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride, int use_preemph,
                              float preemph, int n_fft, int hop, int center, int T, float* frames, hipStream_t stream) {
  SC_CHECK_ARG(wave && wave_len && frames, "sc_frontend_frames: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && T_w >= 1 && n_fft >= 2 && hop >= 1 && wave_stride >= T_w, "sc_frontend_frames: bad sizes");
  const int pad = center ? n_fft / 2 : 0;
  SC_CHECK_ARG((!center || pad < T_w) && T_w + 2 * pad >= n_fft && T == (T_w + 2 * pad - n_fft) / hop + 1, "sc_frontend_frames: T=%d does not match the batch", T);
  FrontendArgs a{};
  a.wave = wave; a.wave_len = wave_len; a.wave_stride = wave_stride; a.T_w = T_w;
  a.use_preemph = use_preemph; a.preemph = preemph;
  a.n_fft = n_fft; a.krows = (n_fft + 1) & ~1; a.hop = hop; a.pad = pad; a.T = T;
  hipLaunchKernelGGL(frontend_frames_kernel, dim3(T, B), dim3(256), 0, stream, a, frames);
  SC_LAUNCH_CHECK("frontend_frames_kernel");
  return SC_OK;
}

// frames4 (4 B T, n_fft / 4): row 4 f + r holds window[4 m + r] x sample(4 m + r) of frame f (radix-4 form of the wide transform; `window`
// is the n_fft-long analysis window, zero-padded around win_length as torch.stft pads it).  n_fft a multiple of 4.
SC_API int sc_frontend_frames4(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride, int use_preemph,
                               float preemph, int n_fft, int hop, int center, int T, const float* window, float* frames4, hipStream_t stream) {
  SC_CHECK_ARG(wave && wave_len && frames4 && window, "sc_frontend_frames4: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && T_w >= 1 && n_fft >= 8 && n_fft % 4 == 0 && hop >= 1 && wave_stride >= T_w, "sc_frontend_frames4: bad sizes (n_fft a multiple of 4)");
  const int pad = center ? n_fft / 2 : 0;
  SC_CHECK_ARG((!center || pad < T_w) && T_w + 2 * pad >= n_fft && T == (T_w + 2 * pad - n_fft) / hop + 1, "sc_frontend_frames4: T=%d does not match the batch", T);
  FrontendArgs a{};
  a.wave = wave; a.wave_len = wave_len; a.wave_stride = wave_stride; a.T_w = T_w;
  a.use_preemph = use_preemph; a.preemph = preemph;
  a.n_fft = n_fft; a.krows = n_fft; a.hop = hop; a.pad = pad; a.T = T;
  hipLaunchKernelGGL(frontend_frames4_kernel, dim3(T, B), dim3(256), 0, stream, a, window, frames4);
  SC_LAUNCH_CHECK("frontend_frames4_kernel");
  return SC_OK;
}

// spec4 (4 B T, 32 * sc_frontend_num_tiles(n_fft / 4)) f32 = frames4 x the unwindowed (n_fft / 4)-point basis -> the outputs of
// sc_frontend_logmel_fwd; twiddle (4, n_fft / 2 + 1, 2) f64 = cos, sin of 2 pi r k / n_fft.  Other arguments as sc_frontend_spec_finish.
SC_API int sc_frontend_spec_finish4(const float* spec4, const double* twiddle, const long long* wave_len, int B, int T_w, int n_fft, int hop,
                                    int center, const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                                    float clamp, int logging, float log_div, float* out_feat, float* out_energy, float* out_linear,
                                    float lin_clamp, int lin_logging, float lin_log_div, long long* feat_len, int T, hipStream_t stream) {
  SC_CHECK_ARG(spec4 && twiddle && wave_len && feat_len, "sc_frontend_spec_finish4: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && T >= 1 && n_fft >= 8 && n_fft % 4 == 0 && hop >= 1, "sc_frontend_spec_finish4: bad sizes");
  SC_CHECK_ARG(out_feat == nullptr || (melw && band_lo && band_hi && n_mels >= 1), "sc_frontend_spec_finish4: mel output needs melw/bands");
  SC_CHECK_ARG(out_feat || out_linear || out_energy, "sc_frontend_spec_finish4: no output requested");
  FrontendArgs a{};
  a.wave_len = wave_len; a.T_w = T_w;
  a.n_fft = n_fft; a.krows = n_fft; a.hop = hop; a.pad = center ? n_fft / 2 : 0; a.T = T;
  a.NT = sc_frontend_num_tiles(n_fft); a.NB = n_fft / 2 + 1;
  a.melw = melw; a.band_lo = band_lo; a.band_hi = band_hi; a.n_mels = out_feat ? n_mels : 0;
  a.mag_spec = mag_spec; a.clamp = clamp; a.logging = logging; a.log_div = log_div;
  a.log_scale = (float)(0.69314718055994530942 / (double)log_div);
  a.out_feat = out_feat; a.out_energy = out_energy; a.out_linear = out_linear;
  a.lin_clamp = lin_clamp; a.lin_logging = lin_logging; a.lin_log_div = lin_log_div;
  a.feat_len = feat_len;
  a.PS = a.NT * 16 + 1;
  a.OS = (n_mels > 0 ? n_mels : 1) | 1;
  a.off_P = 0;
  a.off_O = (32 * a.PS + 3) & ~3;
  const size_t smem = (size_t)(a.off_O + (out_feat ? 32 * a.OS : 0)) * sizeof(float);
  SC_CHECK_ARG(smem <= 160 * 1024, "sc_frontend_spec_finish4: n_fft=%d needs %zu B of LDS (> 160 KiB)", n_fft, smem);
  SC_HIP(hipFuncSetAttribute((const void*)frontend_spec_finish4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(frontend_spec_finish4_kernel, dim3(sc_cdiv(T, 32), B), dim3(256), smem, stream, a, spec4, twiddle);
  SC_LAUNCH_CHECK("frontend_spec_finish4_kernel");
  return SC_OK;
}

// spec (B T, 32 * sc_frontend_num_tiles(n_fft)) f32 = frames x basis -> the outputs of sc_frontend_logmel_fwd (same arguments, same
// meaning; mel_image unused).
SC_API int sc_frontend_spec_finish(const float* spec, const long long* wave_len, int B, int T_w, int n_fft, int hop, int center,
                                   const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec, float clamp,
                                   int logging, float log_div, float* out_feat, float* out_energy, float* out_linear, float lin_clamp,
                                   int lin_logging, float lin_log_div, long long* feat_len, int T, hipStream_t stream) {
  SC_CHECK_ARG(spec && wave_len && feat_len, "sc_frontend_spec_finish: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && T >= 1 && n_fft >= 2 && hop >= 1, "sc_frontend_spec_finish: bad sizes");
  SC_CHECK_ARG(out_feat == nullptr || (melw && band_lo && band_hi && n_mels >= 1), "sc_frontend_spec_finish: mel output needs melw/bands");
  SC_CHECK_ARG(out_feat || out_linear || out_energy, "sc_frontend_spec_finish: no output requested");
  FrontendArgs a{};
  a.wave_len = wave_len; a.T_w = T_w;
  a.n_fft = n_fft; a.krows = (n_fft + 1) & ~1; a.hop = hop; a.pad = center ? n_fft / 2 : 0; a.T = T;
  a.NT = sc_frontend_num_tiles(n_fft); a.NB = n_fft / 2 + 1;
  a.melw = melw; a.band_lo = band_lo; a.band_hi = band_hi; a.n_mels = out_feat ? n_mels : 0;
  a.mag_spec = mag_spec; a.clamp = clamp; a.logging = logging; a.log_div = log_div;
  a.log_scale = (float)(0.69314718055994530942 / (double)log_div);
  a.out_feat = out_feat; a.out_energy = out_energy; a.out_linear = out_linear;
  a.lin_clamp = lin_clamp; a.lin_logging = lin_logging; a.lin_log_div = lin_log_div;
  a.feat_len = feat_len;
  a.PS = a.NT * 16 + 1;
  a.OS = (n_mels > 0 ? n_mels : 1) | 1;
  a.off_P = 0;
  a.off_O = (32 * a.PS + 3) & ~3;
  const size_t smem = (size_t)(a.off_O + (out_feat ? 32 * a.OS : 0)) * sizeof(float);
  SC_CHECK_ARG(smem <= 160 * 1024, "sc_frontend_spec_finish: n_fft=%d needs %zu B of LDS (> 160 KiB)", n_fft, smem);
  SC_HIP(hipFuncSetAttribute((const void*)frontend_spec_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(frontend_spec_finish_kernel, dim3(sc_cdiv(T, 32), B), dim3(256), smem, stream, a, spec);
  SC_LAUNCH_CHECK("frontend_spec_finish_kernel");
  return SC_OK;
}
