// Griffin-Lim waveform recovery for gfx950 -- the kernels behind Speech2LinearSpec.recover (speechain/module/frontend/speech2linear.py:
// 252-351; reference path /root/reference/), SURVEY.md row f4.
//
// The reference delegates the iteration to torchaudio.transforms.GriffinLim (speech2linear.py:279-293; torchaudio is a third-party
// dependency absent from the image and from /root/reference: requirements.txt:17 pins 0.13.0, environment.yaml:154 2.0.2).  Its published
// algorithm (torchaudio.functional.griffinlim) is, for a magnitude spectrogram S and momentum m:
//     angles <- initial phases;  tprev <- 0
//     repeat n_iter times:  y <- istft(S * angles);  R <- stft(y);  A <- R - tprev * m / (1 + m);  angles <- A / (|A| + 1e-16);  tprev <- R
//     return istft(S * angles)
// Both transforms are products of the library's exact-f32 GEMM (csrc/gemm.hip) with a basis that has the window folded in:
//     stft :  frames (sc_frontend_frames) x basis (sc_frontend_basis_init)            -> spectrum, [16 re | 16 im] column tiles
//     istft:  spectrum x inverse basis (sc_istft_basis_init) -> windowed frames -> sc_istft_overlap_add (gather form: every output sample
//             sums the <= n_fft / hop frames that cover it, divides by the window envelope: deterministic, no atomics)
// sc_gl_phase_update is the element-wise step between them, sc_fir_causal the inverse pre-emphasis (:301-345: a 100-tap FIR of powers of
// the coefficient).
#include "sc_common.h"

namespace {

// inv[c][n], c = tile * 32 + j (j < 16: real part of bin tile * 16 + j, else imaginary part), n < krows:
//   x[n] = (1 / N) sum_k w_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)),  w_k = 1 for k = 0 and k = N / 2, else 2 (irfft),
// times the synthesis window centred in n_fft (torch.istft multiplies the frames by the window before the overlap-add).
__global__ void istft_basis_kernel(const float* window, int win_length, int n_fft, int krows, int NT, int NB, float* inv) {
  const long long total = (long long)NT * 32 * krows;
  const int left = (n_fft - win_length) / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / krows), n = (int)(i - (long long)c * krows);
    const int j = c & 31, bin = (c >> 5) * 16 + (j & 15);
    float v = 0.f;
    if (n < n_fft && bin < NB && n >= left && n < left + win_length) {
      const long long m = ((long long)bin * n) % n_fft;
      double s, c2;
      sincospi(2.0 * (double)m / (double)n_fft, &s, &c2);
      const double wk = (bin == 0 || 2 * bin == n_fft) ? 1.0 : 2.0;
      const double w = (double)window[n - left] * wk / (double)n_fft;
      v = (float)((j < 16) ? w * c2 : -w * s);
    }
    inv[i] = v;
  }
}

// out[b][n'] = sum_t frames[b][t][n' + pad - t hop] / sum_t win^2[n' + pad - t hop - left]   (center = true: n' + pad indexes the padded signal)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* frames, const float* window, int win_length, int n_fft, int krows,
                                                        int hop, int T, float* out, int L, long long out_stride) {
  const int b = blockIdx.y;
  const int np = blockIdx.x * 256 + threadIdx.x;
  if (np >= L) return;
  const int m = np + n_fft / 2;
  const int left = (n_fft - win_length) / 2;
  int t_hi = m / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  float acc = 0.f, env = 0.f;
  for (int t = t_hi; t >= 0; --t) {
    const int k = m - t * hop;
    if (k >= n_fft) break;
    acc += frames[((long long)b * T + t) * krows + k];
    if (k >= left && k < left + win_length) {
      const float w = window[k - left];
      env = fmaf(w, w, env);
    }
  }
  out[(long long)b * out_stride + np] = env > 1e-11f ? acc / env : 0.f;
}

// spectrum layout: (M, NT * 32), tile j of a row = [16 re | 16 im] of bins 16 j .. 16 j + 15.
// mode 0: spec = mag * (angle_re + i angle_im) from the (M, NB) planes `are`, `aim` (the initial phases)
// mode 1: A = rebuilt - coef * tprev;  spec = mag * A / (|A| + 1e-16);  tprev <- rebuilt
__global__ __launch_bounds__(256) void gl_phase_kernel(int mode, const float* mag, const float* are, const float* aim, const float* rebuilt,
                                                       float* tprev, float* spec, long long M, int NT, int NB, float coef) {
  const long long total = M * NT * 16;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / (NT * 16);
    const int c = (int)(i - row * (NT * 16)), tile = c >> 4, jj = c & 15, bin = c;
    const long long o = row * (NT * 32) + tile * 32 + jj;
    float re = 0.f, im = 0.f;
    if (bin < NB) {
      const float s = mag[row * NB + bin];
      if (mode == 0) {
        re = s * are[row * NB + bin];
        im = s * aim[row * NB + bin];
      } else {
        const float rr = rebuilt[o], ri = rebuilt[o + 16];
        const float ar = rr - coef * tprev[o], ai = ri - coef * tprev[o + 16];
        tprev[o] = rr;
        tprev[o + 16] = ri;
        const float inv = 1.f / (sqrtf(ar * ar + ai * ai) + 1e-16f);
        re = s * ar * inv;
        im = s * ai * inv;
      }
    }
    spec[o] = re;
    spec[o + 16] = im;
  }
}

// y[b][n] = sum_{k < taps, k <= n} c^k x[b][n - k]
__global__ __launch_bounds__(256) void fir_causal_kernel(const float* x, long long xs, float* y, long long ys, int L, float c, int taps) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const float* xr = x + (long long)b * xs;
  const int kmax = min(taps - 1, n);
  float acc = 0.f;
  for (int k = kmax; k >= 0; --k) acc = fmaf(acc, c, xr[n - k]);  // Horner over the taps, oldest sample first
  y[(long long)b * ys + n] = acc;
}

}  // namespace

// inverse basis (32 * sc_frontend_num_tiles(n_fft), krows) f32, krows = n_fft rounded up to even.
SC_API int sc_istft_basis_init(const float* window, int win_length, int n_fft, float* inv_basis, hipStream_t stream) {
  SC_CHECK_ARG(window && inv_basis, "sc_istft_basis_init: null pointer");
  SC_CHECK_ARG(n_fft >= 2 && win_length >= 1 && win_length <= n_fft, "sc_istft_basis_init: need 1 <= win_length <= n_fft, got %d, %d", win_length, n_fft);
  const int NT = (n_fft / 2 + 1 + 15) / 16, krows = (n_fft + 1) & ~1;
  hipLaunchKernelGGL(istft_basis_kernel, dim3(256), dim3(256), 0, stream, window, win_length, n_fft, krows, NT, n_fft / 2 + 1, inv_basis);
  SC_LAUNCH_CHECK("istft_basis_kernel");
  return SC_OK;
}

// frames (B T, krows): the windowed inverse-transformed frames (spectrum x inverse basis); out (B, L), L = hop (T - 1) (center = true).
SC_API int sc_istft_overlap_add(const float* frames, const float* window, int win_length, int n_fft, int hop, int B, int T, float* out,
                                int L, long long out_stride, hipStream_t stream) {
  SC_CHECK_ARG(frames && window && out, "sc_istft_overlap_add: null pointer");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && T >= 1 && n_fft >= 2 && hop >= 1 && win_length >= 1 && win_length <= n_fft && L >= 1 &&
                   L <= hop * (T - 1) + n_fft - 2 * (n_fft / 2) && out_stride >= L,
               "sc_istft_overlap_add: bad sizes B=%d T=%d n_fft=%d hop=%d L=%d", B, T, n_fft, hop, L);
  hipLaunchKernelGGL(istft_ola_kernel, dim3(sc_cdiv(L, 256), B), dim3(256), 0, stream, frames, window, win_length, n_fft, (n_fft + 1) & ~1, hop,
                     T, out, L, out_stride);
  SC_LAUNCH_CHECK("istft_ola_kernel");
  return SC_OK;
}

// mode 0: spec <- mag * initial phases (angle_re, angle_im: (M, NB));  mode 1: the Griffin-Lim phase step (rebuilt, tprev: (M, 32 NT))
SC_API int sc_gl_phase_update(int mode, const float* mag, const float* angle_re, const float* angle_im, const float* rebuilt, float* tprev,
                              float* spec, long long M, int n_fft, float coef, hipStream_t stream) {
  SC_CHECK_ARG(mag && spec && M >= 1 && n_fft >= 2, "sc_gl_phase_update: bad arguments");
  SC_CHECK_ARG(mode == 0 ? (angle_re && angle_im) : (mode == 1 && rebuilt && tprev), "sc_gl_phase_update: mode %d lacks its inputs", mode);
  const int NT = (n_fft / 2 + 1 + 15) / 16, NB = n_fft / 2 + 1;
  const long long total = M * NT * 16;
  const int blocks = (int)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(gl_phase_kernel, dim3(blocks), dim3(256), 0, stream, mode, mag, angle_re, angle_im, rebuilt, tprev, spec, M, NT, NB, coef);
  SC_LAUNCH_CHECK("gl_phase_kernel");
  return SC_OK;
}

// y[b][n] = sum_{k < taps} c^k x[b][n - k] (samples before the start count as zero): the inverse pre-emphasis window of
// speech2linear.py:301-345 (F.pad(taps - 1, 0) + Conv1d with weights c^(taps-1) .. c^0).
SC_API int sc_fir_causal(const float* x, long long x_stride, float* y, long long y_stride, int B, int L, float c, int taps, hipStream_t stream) {
  SC_CHECK_ARG(x && y && x != y, "sc_fir_causal: null or aliased pointers");
  SC_CHECK_ARG(B >= 1 && B <= 65535 && L >= 1 && taps >= 1 && x_stride >= L && y_stride >= L, "sc_fir_causal: bad sizes");
  hipLaunchKernelGGL(fir_causal_kernel, dim3(sc_cdiv(L, 256), B), dim3(256), 0, stream, x, x_stride, y, y_stride, L, c, taps);
  SC_LAUNCH_CHECK("fir_causal_kernel");
  return SC_OK;
}
