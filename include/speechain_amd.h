/* speechain_amd -- C ABI of the MI355X (gfx950) kernels behind the SpeeChain module surface.
 *
 * The reference (bagustris/speechain) is pure Python on PyTorch and has no FFI; these entry points are the
 * layer UNDER the reference's `speechain.module.abs.Module` plugin surface.  Each one replaces a chain of stock
 * torch ops in the reference, cited as /root/reference-relative file:line.
 *
 * Conventions (all functions):
 *   - raw device pointers, explicit sizes/strides, scalar hyper-parameters, caller-owned hipStream_t;
 *   - return 0 on success, non-zero on error (message: sc_last_error(), thread-local);
 *   - never allocate, free or synchronise; safe to capture into a hipGraph;
 *   - tensors are dense row-major unless a stride argument is given; lengths are int64 ("long long") on device.
 */
#ifndef SPEECHAIN_AMD_H
#define SPEECHAIN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

const char* sc_last_error(void);
int sc_version(void);

/* ---- acoustic frontend --------------------------------------------------------------------------------
 * Replaces Speech2LinearSpec.forward (speechain/module/frontend/speech2linear.py:142-250) and
 * LinearSpec2MelSpec.forward (speechain/module/frontend/linear2mel.py:163-189), i.e. Speech2MelSpec.forward
 * (speechain/module/frontend/speech2mel.py:158-189) in one launch.                                        */

/* number of 16-bin column tiles and float count of the windowed-DFT basis the kernel consumes */
int sc_frontend_num_tiles(int n_fft);
long long sc_frontend_basis_floats(int n_fft);

/* basis <- window (win_length floats on device, centred in n_fft as torch.stft does) x DFT twiddles, built in
 * double precision on the device.  Replaces the lazy torch.<window>_window + the FFT plan of torch.stft
 * (speech2linear.py:192-198). */
int sc_frontend_basis_init(const float* window, int win_length, int n_fft, float* basis, hipStream_t stream);

/* wave (B, T_w) f32 with row stride wave_stride, wave_len (B,) -> any of
 *   out_feat   (B, T, n_mels)  mel filterbank output, clamp+log when `logging` (value = ln(max(x, clamp)) / log_div)
 *   out_energy (B, T)          sqrt(clamp(sum_f power, 1e-10)), zero beyond feat_len   (speech2linear.py:216-233)
 *   out_linear (B, T, n_fft/2+1) power (or magnitude when mag_spec) spectrogram with its own clamp/log options
 *   feat_len   (B,)            (len + 2*pad - n_fft) // hop + 1                          (speech2linear.py:201-212)
 * T must equal (T_w + 2*pad - n_fft)/hop + 1 with pad = center ? n_fft/2 : 0.  Frames >= feat_len are computed from
 * an all-zero power spectrum exactly as the reference does (speech2linear.py:226-228).
 * melw is the reference's (n_mels, n_fft/2+1) buffer `linear2mel.mel_fbanks.weight`; band_lo/band_hi (n_mels,) int32
 * give for each mel row the half-open range of bins holding its non-zero weights. */
int sc_frontend_logmel_fwd(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride,
                           int use_preemph, float preemph, int n_fft, int hop, int center, const float* basis,
                           const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                           float clamp, int logging, float log_div, float* out_feat, float* out_energy,
                           float* out_linear, float lin_clamp, int lin_logging, float lin_log_div,
                           long long* feat_len, int T, hipStream_t stream);

/* Stand-alone mel filterbank + clamp/log over an (N, NB) spectrogram: LinearSpec2MelSpec.forward
 * (speechain/module/frontend/linear2mel.py:163-189) when it is called on its own. */
int sc_mel_log_fwd(const float* spec, long long N, int NB, const float* melw, const int* band_lo, const int* band_hi,
                   int n_mels, float clamp, int logging, float log_div, float* out, hipStream_t stream);

/* ---- SpecAugment -----------------------------------------------------------------------------------------
 * Replaces SpecAugment.forward after its random draws (speechain/module/augment/specaug.py:111-124 time warp,
 * :160-199 mask construction, :202-204 masked_fill).  in/out (B, T, F) f32.  warp_center < 0 disables the warp;
 * otherwise rows [0, warp_center) are resampled (bicubic, align_corners) onto [0, warp_pos) and rows
 * [warp_center, tmin) onto [warp_pos, tmin).  freq_/time_ len & pos are the (B, nf) / (B, nt) int64 tensors the
 * reference draws with torch.randint; a position is masked when pos <= x < pos + len for any of them.  Masked
 * elements get *fill (device scalar) or 0.0 when fill is null. */
int sc_specaug_fwd(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos,
                   const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                   const long long* time_pos, int nt, const float* fill, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHAIN_AMD_H */
