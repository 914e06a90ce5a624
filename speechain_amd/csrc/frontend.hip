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
};

__device__ __forceinline__ long long floordiv_ll(long long a, long long b) {
  long long q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// Stages the waveform span of frames [t0, t0 + FTF) in LDS: pre-emphasis, reflect padding and tail zeroing applied on the
// way in; one pad word per hop (padflag) so that consecutive frames start on different banks.
template <int FTF>
__device__ __forceinline__ void stage_samples(const FrontendArgs& p, float* smem, int b, int t0, long long wlen, int tid) {
  const float* x = p.wave + (long long)b * p.wave_stride;
  const int count = (FTF - 1) * p.hop + p.krows;
  const int padded_len = p.T_w + 2 * p.pad;
  const int base = t0 * p.hop - p.pad;  // sample index of the span's first element (32-bit: T_w < 2^31)
  const int len = (int)(wlen < p.T_w ? wlen : p.T_w);
  const float negc = -p.preemph;
  const float inv_hop = 1.f / (float)p.hop;
  for (int i = tid; i < count; i += 256) {
    int n = base + i;
    const bool inside = n + p.pad < padded_len;
    n = n < 0 ? -n : n;
    n = n >= p.T_w ? 2 * (p.T_w - 1) - n : n;
    float v = 0.f;
    if (inside) {
      if (p.use_preemph) {
        if (n < len) v = fmaf(negc, n > 0 ? x[n - 1] : 0.f, x[n]);
      } else {
        v = x[n];
      }
    }
    smem[i + (int)(((float)i + 0.5f) * inv_hop) * p.padflag] = v;  // + one pad word per completed hop
  }
}

// Per-frame epilogue on the power image P (FTF rows, stride p.PS): banded mel + clamp/log with one lane per frame, frame
// energy, coalesced copy-out of the mel rows (through the O staging image) and of the optional linear spectrogram.
template <int FTF>
__device__ __forceinline__ void frontend_epilogue(const FrontendArgs& p, const float* P, float* O, int b, int t0, int flen,
                                                  int tid, int lane, int wid) {
  const int f = lane & (FTF - 1);
  const int t = t0 + f;
  const bool live = t < flen;
  const float* prow = P + f * p.PS;

  if (p.out_feat) {
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
    for (int e = tid; e < total; e += 256) {
      const int ff = e / p.n_mels;
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

  stage_samples<FTF>(p, smem, b, t0, wlen, tid);
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
          v += __shfl_xor(v, 16, 64);
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


// ---- n_fft = 400: Good-Thomas (prime-factor) two-stage DFT, 400 = 16 x 25 with gcd 1 -> no twiddle factors --------------
//   input index  n = (25 n1 + 16 n2) mod 400,   output index  k = (225 k1 + 176 k2) mod 400   (CRT maps)
//   stage 1: 25-point real-input DFTs along n2 for each n1 (k2 = 0..12 suffice: conjugate symmetry)   [32 frames x 26] x [26 x 32]
//   stage 2: 16-point complex DFTs along n1 for each k2                                              [32 frames x 32] x [32 x 32]
// both as exact-f32 MFMA products with the (tiny) DFT matrices held in registers; 13 MFMAs per frame instead of the 81 of the
// direct 400 x 402 product, and FFT-like (short) accumulation chains.  The 208 computed (k1, k2) values cover the 201 bins
// after folding k > 200 to 400 - k.
constexpr int PFA_FLOATS = 26 * 32 + 32 * 32 + 16 * 32 + 400;
constexpr int PFA_YS = 401;  // Y row stride per frame: 16 * 25 + 1 (odd: conflict-free frame-major reads)

__global__ __launch_bounds__(256, 2) void frontend_pfa400_kernel(FrontendArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int FTF = 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FTF;
  const long long wlen = p.wave_len[b];
  long long flen_ll = floordiv_ll(wlen + 2LL * p.pad - p.n_fft, p.hop) + 1;
  if (blockIdx.x == 0 && tid == 0) p.feat_len[b] = flen_ll;
  const int flen = (int)(flen_ll < 0 ? 0 : (flen_ll > p.T ? p.T : flen_ll));

  stage_samples<FTF>(p, smem, b, t0, wlen, tid);

  const float* W25 = p.pfa;
  const float* W16c = W25 + 26 * 32;
  const float* W16r = W16c + 32 * 32;
  const float* gwin = W16r + 16 * 32;  // window, zero-padded to 400 samples
  float* Y = smem;              // overlaps the sample image (rewritten only after every wave finished stage 1)
  float* P = smem + p.off_P;
  float* win = P + FTF * p.PS;  // 400-entry window table in LDS
  for (int i = tid; i < 400; i += 256) win[i] = gwin[i];

  // ---- stage 1: wave w transforms n1 = 4w .. 4w+3 ----
  {
    float b1[13];
#pragma unroll
    for (int s = 0; s < 13; ++s) b1[s] = W25[(2 * s + h) * 32 + li];
    f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    __syncthreads();  // samples and window table are staged
    const float* frow = smem + li * (p.hop + p.padflag);
    const float inv_hop = 1.f / (float)p.hop;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n1 = wid * 4 + u;
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        // n = (25 n1 + 16 n2) mod 400 with n2 = 2s + h; the padded k-step (n2 = 25) multiplies a zero basis row
        int n = 25 * n1 + 16 * (2 * s + h);
        n = n >= 400 ? n - 400 : n;
        n = (2 * s + h) < 25 ? n : 0;
        const int pos = n + (int)(((float)n + 0.5f) * inv_hop) * p.padflag;  // + one pad word per completed hop
        const float a = win[n] * frow[pos];
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[s], acc[u], 0, 0, 0);
      }
    }
    __syncthreads();
    if (li < 25) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Y[((r & 3) + 8 * (r >> 2) + 4 * h) * PFA_YS + (wid * 4 + u) * 25 + li] = acc[u][r];
    }
  }
  __syncthreads();

  // ---- stage 2: k2 = 0 (real input, 8 MFMAs) and k2 = 1..12 (16 MFMAs each): waves take {0,1,2,3} {4,5,6} {7,8,9} {10,11,12}
  {
    float b2[16], b0[8];
#pragma unroll
    for (int s = 0; s < 16; ++s) b2[s] = W16c[(2 * s + h) * 32 + li];
#pragma unroll
    for (int s = 0; s < 8; ++s) b0[s] = W16r[(2 * s + h) * 32 + li];
    const int kbeg = wid == 0 ? 0 : 1 + 3 * wid, kend = wid == 0 ? 4 : 4 + 3 * wid;
    const float* yrow = Y + li * PFA_YS;
    for (int k2 = kbeg; k2 < kend; ++k2) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (k2 == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yrow[(2 * s + h) * 25], b0[s], acc, 0, 0, 0);
      } else {
        const int comp = h ? 12 + k2 : k2;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yrow[s * 25 + comp], b2[s], acc, 0, 0, 0);
      }
      int bin = (225 * (li & 15) + 176 * k2) % 400;
      if (bin > 200) bin = 400 - bin;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r] * acc[r];
        v += __shfl_xor(v, 16, 64);
        if ((li & 16) == 0) P[((r & 3) + 8 * (r >> 2) + 4 * h) * p.PS + bin] = p.mag_spec ? sqrtf(v) : v;
      }
    }
  }
  __syncthreads();
  frontend_epilogue<FTF>(p, P, smem + p.off_O, b, t0, flen, tid, lane, wid);
}

// tables of the two-stage DFT; tpos holds LDS offsets (one pad word per hop when hop is even)
__global__ void pfa400_tables_kernel(const float* window, int win_length, int hop, float* out) {
  float* W25 = out;
  float* W16c = W25 + 26 * 32;
  float* W16r = W16c + 32 * 32;
  float* gwin = W16r + 16 * 32;
  const int left = (400 - win_length) / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 400; i += gridDim.x * blockDim.x)
    gwin[i] = (i >= left && i < left + win_length) ? window[i - left] : 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * 32; i += gridDim.x * blockDim.x) {
    const int k = i >> 5, j = i & 31;
    double s, c;
    if (k < 26) {  // W25[n2][comp]
      float v = 0.f;
      if (k < 25 && j < 25) {
        const int kk = j < 13 ? j : j - 12;
        sincospi(2.0 * ((k * kk) % 25) / 25.0, &s, &c);
        v = (float)(j < 13 ? c : -s);
      }
      W25[i] = v;
    }
    {  // W16c[(n1, ri)][(k1 | 16 + k1)]
      const int n1 = k >> 1, ri = k & 1, k1 = j & 15;
      sincospi(2.0 * ((n1 * k1) % 16) / 16.0, &s, &c);
      W16c[i] = (float)(j < 16 ? (ri ? s : c) : (ri ? c : -s));
    }
    if (k < 16) {  // W16r[n1][(k1 | 16 + k1)]
      const int k1 = j & 15;
      sincospi(2.0 * ((k * k1) % 16) / 16.0, &s, &c);
      W16r[i] = (float)(j < 16 ? c : -s);
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
    hipLaunchKernelGGL(pfa400_tables_kernel, dim3(4), dim3(256), 0, stream, window, win_length, hop, basis + direct_basis_floats(n_fft));
    SC_LAUNCH_CHECK("pfa400_tables_kernel");
  }
  return SC_OK;
}

SC_API int sc_frontend_logmel_fwd(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride,
                                  int use_preemph, float preemph, int n_fft, int hop, int center, const float* basis,
                                  const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                                  float clamp, int logging, float log_div, float* out_feat, float* out_energy,
                                  float* out_linear, float lin_clamp, int lin_logging, float lin_log_div,
                                  long long* feat_len, int T, hipStream_t stream) {
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
  a.out_feat = out_feat; a.out_energy = out_energy; a.out_linear = out_linear;
  a.lin_clamp = lin_clamp; a.lin_logging = lin_logging; a.lin_log_div = lin_log_div;
  a.feat_len = feat_len;
  a.padflag = (hop % 2 == 0) ? 1 : 0;

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
    // two-stage 16 x 25 DFT: 32 frames per workgroup; Y (32 x 401) overlaps the sample image, P follows, O overlaps Y
    a.pfa = basis + direct_basis_floats(n_fft);
    const int count32 = 31 * hop + a.krows;
    const int samples32 = count32 + count32 / hop + 8;
    const int region0 = samples32 > 32 * PFA_YS ? samples32 : 32 * PFA_YS;
    a.off_P = (region0 + 3) & ~3;
    a.off_O = 0;
    a.PS = 201;  // exactly the 201 bins (odd stride: conflict-free frame-major reads)
    const size_t smem_pfa = (size_t)(a.off_P + 32 * a.PS + 400) * sizeof(float);
    if (smem_pfa <= 160 * 1024 && 32 * a.OS <= region0) {
      SC_HIP(hipFuncSetAttribute((const void*)frontend_pfa400_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_pfa));
      hipLaunchKernelGGL(frontend_pfa400_kernel, dim3(sc_cdiv(T, 32), B), dim3(256), smem_pfa, stream, a);
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
