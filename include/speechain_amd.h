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
 * double precision on the device; for n_fft = 400 it also holds the tables of the 16 x 25 prime-factor DFT the kernel then
 * uses (they depend on hop through the LDS padding).  Replaces the lazy torch.<window>_window + the FFT plan of torch.stft
 * (speech2linear.py:192-198). */
int sc_frontend_basis_init(const float* window, int win_length, int n_fft, int hop, float* basis, hipStream_t stream);

/* Band-packed image of the mel matrix `linear2mel.mel_fbanks.weight` (linear2mel.py:135-140): 16 weight slots per
 * filter + per-filter first bin / width, sc_frontend_mel_image_floats(n_mels) floats on the device.  Built once per
 * (matrix, device); replaces nothing in the reference (it is the layout the fused kernel consumes). */
long long sc_frontend_mel_image_floats(int n_mels);
int sc_frontend_mel_image_init(const float* melw, const int* band_lo, const int* band_hi, int n_mels, int n_bins,
                               float* image, hipStream_t stream);

/* wave (B, T_w) f32 with row stride wave_stride, wave_len (B,) -> any of
 *   out_feat   (B, T, n_mels)  mel filterbank output, clamp+log when `logging` (value = ln(max(x, clamp)) / log_div)
 *   out_energy (B, T)          sqrt(clamp(sum_f power, 1e-10)), zero beyond feat_len   (speech2linear.py:216-233)
 *   out_linear (B, T, n_fft/2+1) power (or magnitude when mag_spec) spectrogram with its own clamp/log options
 *   feat_len   (B,)            (len + 2*pad - n_fft) // hop + 1                          (speech2linear.py:201-212)
 * T must equal (T_w + 2*pad - n_fft)/hop + 1 with pad = center ? n_fft/2 : 0.  Frames >= feat_len are computed from
 * an all-zero power spectrum exactly as the reference does (speech2linear.py:226-228).
 * melw is the reference's (n_mels, n_fft/2+1) buffer `linear2mel.mel_fbanks.weight`; band_lo/band_hi (n_mels,) int32
 * give for each mel row the half-open range of bins holding its non-zero weights.  mel_image (nullable) is the
 * band-packed copy of melw made by sc_frontend_mel_image_init; with it the n_fft = 400 kernel keeps the filter weights
 * in LDS, without it (or when a filter is wider than 16 bins) it reads the dense matrix. */
int sc_frontend_logmel_fwd(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride,
                           int use_preemph, float preemph, int n_fft, int hop, int center, const float* basis,
                           const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                           float clamp, int logging, float log_div, float* out_feat, float* out_energy,
                           float* out_linear, float lin_clamp, int lin_logging, float lin_log_div,
                           long long* feat_len, int T, const float* mel_image, hipStream_t stream);

/* Wide transforms (n_fft >= 512; the FastSpeech2 preset has 1024) as three launches instead of sc_frontend_logmel_fwd's one:
 *   sc_frontend_frames       frames (B T, krows) f32, krows = n_fft rounded up to even: pre-emphasis, reflect padding, tail zeroing
 *                            (speech2linear.py:167-173 and the framing inside torch.stft, :198)
 *   sc_gemm (dtype 0, tb 1)  spec (B T, 32 * sc_frontend_num_tiles(n_fft)) = frames x the basis of sc_frontend_basis_init
 *   sc_frontend_spec_finish  power / magnitude, mel, clamp / log, energy, lengths: the outputs of sc_frontend_logmel_fwd
 * The LDS-staged exact-f32 product runs the 1024-point transform of a 32 x 7.4 s batch in 0.6 ms where the one-launch kernel takes 1.35 ms. */
int sc_frontend_frames(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride, int use_preemph, float preemph,
                       int n_fft, int hop, int center, int T, float* frames, hipStream_t stream);
int sc_frontend_spec_finish(const float* spec, const long long* wave_len, int B, int T_w, int n_fft, int hop, int center, const float* melw,
                            const int* band_lo, const int* band_hi, int n_mels, int mag_spec, float clamp, int logging, float log_div,
                            float* out_feat, float* out_energy, float* out_linear, float lin_clamp, int lin_logging, float lin_log_div,
                            long long* feat_len, int T, hipStream_t stream);
/* The wide transform in radix-4 form (n_fft a multiple of 4; a quarter of the product's flops): X[k] = sum_r e^{-2 pi i r k / n_fft}
 * Y_r[k mod Q], Y_r the Q = n_fft / 4-point transforms of the four decimated, windowed sequences x[4 m + r].
 *   sc_frontend_frames4: frames4 (4 B T, Q), row 4 f + r = window[4 m + r] x sample(4 m + r) of frame f (window: n_fft floats, zero-padded
 *     around win_length as torch.stft pads it);
 *   sc_gemm (f32): spec4 (4 B T, 32 * sc_frontend_num_tiles(Q)) = frames4 x the UNWINDOWED Q-point basis (sc_frontend_basis_init with a
 *     window of ones, win_length = n_fft = Q);
 *   sc_frontend_spec_finish4: the twiddled sum (twiddle (4, n_fft / 2 + 1, 2) f64 = cos, sin of 2 pi r k / n_fft; the sum runs in double), then as
 *     sc_frontend_spec_finish. */
int sc_frontend_frames4(const float* wave, const long long* wave_len, int B, int T_w, long long wave_stride, int use_preemph,
                        float preemph, int n_fft, int hop, int center, int T, const float* window, float* frames4, hipStream_t stream);
int sc_frontend_spec_finish4(const float* spec4, const double* twiddle, const long long* wave_len, int B, int T_w, int n_fft, int hop,
                             int center, const float* melw, const int* band_lo, const int* band_hi, int n_mels, int mag_spec,
                             float clamp, int logging, float log_div, float* out_feat, float* out_energy, float* out_linear,
                             float lin_clamp, int lin_logging, float lin_log_div, long long* feat_len, int T, hipStream_t stream);

/* ---- Griffin-Lim waveform recovery (csrc/griffinlim.hip): Speech2LinearSpec.recover (speechain/module/frontend/speech2linear.py:252-351).
 * The reference delegates the iteration to torchaudio.transforms.GriffinLim (:279-293; third party, absent); its published algorithm runs
 * here as products of sc_gemm (dtype 0) with two bases plus these element-wise / gather kernels:
 *   sc_istft_basis_init   inverse basis (32 * sc_frontend_num_tiles(n_fft), krows): irfft weights / n_fft times the centred synthesis window
 *   sc_istft_overlap_add  frames (B T, krows) = spectrum x inverse basis -> out (B, L): overlap-add divided by the window envelope, the
 *                         n_fft / 2 reflect padding of center = true trimmed (L <= hop (T - 1))
 *   sc_gl_phase_update    mode 0: spec = mag * initial phases; mode 1: A = rebuilt - coef * tprev, spec = mag * A / (|A| + 1e-16),
 *                         tprev <- rebuilt (coef = momentum / (1 + momentum)); spectra in the [16 re | 16 im] column-tile layout (M, 32 NT)
 *   sc_fir_causal         y[n] = sum_{k < taps} c^k x[n - k]: the sliding-window inverse pre-emphasis (:301-345) */
int sc_istft_basis_init(const float* window, int win_length, int n_fft, float* inv_basis, hipStream_t stream);
int sc_istft_overlap_add(const float* frames, const float* window, int win_length, int n_fft, int hop, int B, int T, float* out, int L,
                         long long out_stride, hipStream_t stream);
int sc_gl_phase_update(int mode, const float* mag, const float* angle_re, const float* angle_im, const float* rebuilt, float* tprev,
                       float* spec, long long M, int n_fft, float coef, hipStream_t stream);
int sc_fir_causal(const float* x, long long x_stride, float* y, long long y_stride, int B, int L, float c, int taps, hipStream_t stream);

/* Stand-alone mel filterbank + clamp/log over an (N, NB) spectrogram: LinearSpec2MelSpec.forward
 * (speechain/module/frontend/linear2mel.py:163-189) when it is called on its own. */
int sc_mel_log_fwd(const float* spec, long long N, int NB, const float* melw, const int* band_lo, const int* band_hi,
                   int n_mels, float clamp, int logging, float log_div, float* out, hipStream_t stream);

/* ---- feature normalisation (speechain/module/norm/feat_norm.py) ------------------------------------------------------------
 * sc_featnorm_stats: per-utterance statistics over the valid frames feat[b][:feat_len[b]] of a (B, T, F) f32 tensor with strides
 * (stride_b, stride_t, 1): mean (B, F) and the unbiased standard deviation clamped from below (B, F) -- the Python loops at
 * feat_norm.py:115-133.  Either output may be null.  One frame gives std = NaN and no frame mean = NaN, as torch does.
 * sc_featnorm_apply: out = (feat - mean) / std over the WHOLE padded tensor (feat_norm.py:137-139, 280-299, 322-331); a
 * statistics stride of 0 means one shared (F,) vector, otherwise one vector per utterance; a null vector skips that step. */
int sc_featnorm_stats(const float* feat, long long stride_b, long long stride_t, const long long* feat_len, int B, int T, int F,
                      float clamp, float* mean, float* std, hipStream_t stream);
int sc_featnorm_apply(const float* feat, float* out, long long stride_b, long long stride_t, long long out_stride_b,
                      long long out_stride_t, int B, int T, int F, const float* mean, long long mean_stride_b, const float* std,
                      long long std_stride_b, hipStream_t stream);

/* ---- Conv2d prenet (speechain/module/prenet/conv2d.py:168-205 construction, :227-280 forward) ---------------------------------
 * [Conv2d(kernel KH x KW, stride SH x SW, zero padding PH x PW) -> BatchNorm2d -> activation] blocks on a (B, T, F) feature image.
 * Every intermediate is channels-last: (B, To, Fo, C), i.e. a (B*To*Fo, C) matrix in `dtype` (0 = f32, 1 = bf16).  Geometry
 * arguments must satisfy To = (Ti + 2 PH - KH) / SH + 1 (same for Fo).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 Tanh (the Conv1d
 * postnet of the TTS recipes, postnet/conv1d.py:117-127, runs its blocks through the same kernels as (B, T, 1, C) images).
 *
 * First block (one input channel, KK = KH*KW <= 15 taps; replaces nn.Conv2d(1, C) + nn.BatchNorm2d + activation at conv2d.py:171-190):
 *   sc_conv1_moments  moments[KK*KK + KK] (float64) = sum over all output positions of patch patch^T, then of patch; x is the (B, Ti, Fi)
 *                     f32 image with strides (stride_b, stride_t, 1).
 *   sc_conv1_bn_coeffs  per channel, from the moments and w (C, KK): batch mean / biased variance of z = w.patch, scale = gamma*rstd,
 *                     shift = beta - mean*scale, saved mean / rstd; running_mean / running_var (nullable) move by `momentum` with the
 *                     unbiased variance, as torch.nn.BatchNorm2d in training mode.
 *   sc_conv1_apply    y[b][to][fo][c] = act(scale[c] * (w[c] . patch) + shift[c])   (no BatchNorm: scale = 1, shift = conv bias).
 *   sc_conv1_bwd      from dy (same layout as y): dw (C, KK), dgamma (C), dbeta (C; the bias gradient when mean == rstd == null).
 *                     batch_stats = 1: gradients flow through the batch statistics (training mode), needs `moments`; 0: fixed statistics. */
long long sc_conv1_moments_workspace(int B, int Ti, int Fi, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW);
int sc_conv1_moments(const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH, int KW,
                     int SH, int SW, int PH, int PW, double* moments, void* workspace, long long ws_bytes, hipStream_t stream);
int sc_conv1_bn_coeffs(const double* moments, const float* w, int C, int KK, long long count, const double* count_dev,
                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* scale, float* shift, float* mean, float* rstd, hipStream_t stream);
int sc_conv1_apply(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH,
                   int KW, int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act,
                   float slope, void* y, hipStream_t stream);
long long sc_conv1_bwd_workspace(int B, int Ti, int Fi, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW, int C);
int sc_conv1_bwd(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH, int KW,
                 int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act, float slope,
                 const void* dy, const double* moments, const float* gamma, const float* mean, const float* rstd, int batch_stats,
                 float* dw, float* dgamma, float* dbeta, void* workspace, long long ws_bytes, hipStream_t stream);

/* Synchronised batch statistics (torch.nn.SyncBatchNorm, which the reference's DDP set-up turns every BatchNorm into:
 * speechain/runner.py:303-310 `enable_syncbatchnorm` default True, :1856-1858 convert_sync_batchnorm before the DDP wrap).  Mean and
 * variance are those of the batches of ALL ranks: the per-rank sums (the float64 moment / sum vectors below) are all-reduced by the
 * caller between the two phases of each entry point above, which exist separately for that purpose:
 *   forward, first block:  sc_conv1_moments -> [all-reduce moments, count] -> sc_conv1_bn_coeffs(count_dev = the summed count)
 *   backward, first block: sc_conv1_bwd_reduce (gsum[C][KK + 1] = sum g.patch_k, sum g) -> [all-reduce] -> sc_conv1_bwd_finish
 *   forward, other blocks: sc_bn_sums(dy = null) (sums[2][C] = sum z, sum z^2) -> [all-reduce] -> sc_bn_stats_finish
 *   backward:              sc_bn_sums(dy) (sum g, sum g z) -> [all-reduce] -> sc_bn_bwd_finish (coef[3][C]) -> sc_bn_bwd_apply
 * The *_finish kernels take the rank's own sums for what it contributes to dgamma / dbeta / dw (the gradient all-reduce averages the ranks
 * afterwards, as DDP does for SyncBatchNorm) and the all-rank sums (`*_all`, `count_dev`: one float64 on the device) for the terms that
 * come from differentiating the statistics; null `*_all` / `count_dev` = one rank (`count` is used).  `moments` of sc_conv1_bwd_finish are
 * THIS rank's.  sc_conv1_bwd / sc_bn_stats / sc_bn_act_bwd are the one-rank compositions.  Workspaces: sc_conv1_bwd_reduce needs
 * sc_conv1_bwd_workspace(..) - C*(KK+1)*8 bytes, sc_bn_sums needs sc_bn_workspace(M, C) - 16*C bytes. */
int sc_conv1_bwd_reduce(int dtype, const float* x, long long stride_b, long long stride_t, int B, int Ti, int Fi, int To, int Fo, int KH,
                        int KW, int SH, int SW, int PH, int PW, const float* w, int C, const float* scale, const float* shift, int act,
                        float slope, const void* dy, double* gsum, void* workspace, long long ws_bytes, hipStream_t stream);
int sc_conv1_bwd_finish(const double* gsum, const double* gsum_all, const double* moments, const float* w, int C, int KK, long long count,
                        const double* count_dev, const float* gamma, const float* mean, const float* rstd, int batch_stats, float* dw,
                        float* dgamma, float* dbeta, hipStream_t stream);
int sc_bn_sums(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, int act, float slope,
               double* sums, void* workspace, long long ws_bytes, hipStream_t stream);
int sc_bn_stats_finish(const double* sums, int C, long long count, const double* count_dev, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd,
                       hipStream_t stream);
int sc_bn_bwd_finish(const double* sums, const double* sums_all, int C, long long count, const double* count_dev, const float* scale,
                     const float* mean, const float* rstd, int batch_stats, float* dgamma, float* dbeta, float* coef, hipStream_t stream);
int sc_bn_bwd_apply(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, const float* coef,
                    int act, float slope, void* dz, hipStream_t stream);

/* Blocks with C input channels (nn.Conv2d(C, C_out) at conv2d.py:171-180 as im2col + sc_gemm; its autograd as sc_gemm_splitk / sc_gemm +
 * sc_col2im):  col[(b, to, fo)][(kh, kw, c)] = a[b][to*SH - PH + kh][fo*SW - PW + kw][c] (0 outside the image);
 * sc_col2im is its adjoint, da[b][t][f][c] = sum of the dcol entries that read a[b][t][f][c] (a gather: deterministic). */
int sc_im2col(int dtype, const void* a, int B, int Ti, int Fi, int C, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW,
              void* col, hipStream_t stream);
int sc_col2im(int dtype, const void* dcol, int B, int Ti, int Fi, int C, int To, int Fo, int KH, int KW, int SH, int SW, int PH, int PW,
              void* da, hipStream_t stream);

/* nn.BatchNorm2d (+ activation) on a channels-last tensor = per-column statistics of an (M, C) matrix z (conv2d.py:184-190).
 *   sc_bn_stats    batch mean / biased variance per column -> scale, shift, mean, rstd; running statistics updated as above.
 *   sc_bn_act_fwd  y = act(scale[c] * z + shift[c])                (eval mode / no BatchNorm: the caller supplies scale and shift).
 *   sc_bn_act_bwd  dz, dgamma (nullable), dbeta (nullable) from dy; batch_stats as in sc_conv1_bwd; mean == rstd == null: no BatchNorm
 *                  (dz = scale * dy * act').  Workspace: sc_bn_workspace(M, C) bytes (+ 12*C for sc_bn_act_bwd). */
long long sc_bn_workspace(long long M, int C);
int sc_bn_stats(int dtype, const void* z, long long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd, void* workspace,
                long long ws_bytes, hipStream_t stream);
int sc_bn_act_fwd(int dtype, const void* z, long long M, int C, const float* scale, const float* shift, int act, float slope, void* y,
                  hipStream_t stream);
int sc_bn_act_bwd(int dtype, const void* z, const void* dy, long long M, int C, const float* scale, const float* shift, const float* mean,
                  const float* rstd, int batch_stats, int act, float slope, void* dz, float* dgamma, float* dbeta, void* workspace,
                  long long ws_bytes, hipStream_t stream);

/* ---- SpecAugment -----------------------------------------------------------------------------------------
 * Replaces SpecAugment.forward after its random draws (speechain/module/augment/specaug.py:111-124 time warp,
 * :160-199 mask construction, :202-204 masked_fill).  in/out (B, T, F) f32.  warp_center < 0 disables the warp;
 * otherwise rows [0, warp_center) are resampled (bicubic, align_corners) onto [0, warp_pos) and rows
 * [warp_center, tmin) onto [warp_pos, tmin); warp_mode 0 = bicubic (the default of the reference), 1 = bilinear (`time_warp_mode`, :23).  freq_/time_ len & pos are the (B, nf) / (B, nt) int64 tensors the
 * reference draws with torch.randint; a position is masked when pos <= x < pos + len for any of them.  Masked
 * elements get *fill (device scalar) or 0.0 when fill is null. */
int sc_specaug_fwd(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos, int warp_mode,
                   const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                   const long long* time_pos, int nt, const float* fill, hipStream_t stream);

/* The same with the mean of the WARPED features as fill value (specaug.py:203, feat_norm off) in one call of two launches: the first warps
 * and leaves one sum per workgroup in `workspace` (sc_specaug_meanfill_workspace bytes), the second masks in place with their total / (B T F). */
long long sc_specaug_meanfill_workspace(int B, int T);
int sc_specaug_fwd_meanfill(const float* in, float* out, int B, int T, int F, int tmin, int warp_center, int warp_pos, int warp_mode,
                            const long long* freq_len, const long long* freq_pos, int nf, const long long* time_len,
                            const long long* time_pos, int nt, void* workspace, long long ws_bytes, hipStream_t stream);

/* ---- transformer primitives -------------------------------------------------------------------------------
 * `dtype` selects the activation type of a call: 0 = float32 (exact-f32 matrix cores, parity path),
 * 1 = bfloat16 (bf16 matrix cores with f32 accumulation, training path).  Statistics, biases, gammas, losses and
 * gradients of parameters are always float32. */

/* C[z][m][n] = epilogue(alpha * sum_k opA[z][m][k] * opB[z][n][k]),  z = z1 * nb2 + z2 over nb1 x nb2 batches.
 * opA is M x K: element (m,k) at A[m*lda + k] (ta = 0) or A[k*lda + m] (ta = 1); opB is N x K likewise (tb).
 * Replaces nn.Linear / torch.matmul and their autograd products at speechain/module/transformer/attention.py:72-79,
 * 99-106,131, feed_forward.py:93-101 and speechain/module/postnet/token.py:49.
 * Epilogue, in order: v = alpha*acc + bias[n];  pre_out <- v (optional);
 *   aux_mode 2: v *= act'(aux[m][n])  else  v = act(v)   (act: 0 none, 1 ReLU, 2 exact GELU);
 *   dropout(drop_p) with the stateless generator keyed (*seed, offset + linear index of (z,m,n)); `seed` points to a
 *   device cell so that a captured hipGraph draws fresh masks on every replay once the host bumps the cell;
 *   aux_mode 1: v += aux[m][n];  aux_mode 3: v *= aux[m][n];  store as `dtype`, or as float32 when out_f32 (optionally accumulating).
 * act | 0x100: pre_out receives act'(pre) * (dropped ? 0 : 1 / (1 - drop_p)) instead of the pre-activation -- exactly the factor the
 * data-gradient product of the same layer multiplies by (aux_mode 3), so that launch neither re-evaluates act' nor the dropout mask. */
int sc_gemm(int dtype, int ta, int tb, const void* A, const void* B, void* C, int M, int N, int K, long long lda,
            long long ldb, long long ldc, int nb1, int nb2, long long sA1, long long sA2, long long sB1, long long sB2,
            long long sC1, long long sC2, float alpha, const float* bias, int act, void* pre_out, const void* aux,
            long long ldaux, int aux_mode, float drop_p, const unsigned long long* seed, unsigned long long offset, int out_f32,
            int accumulate, hipStream_t stream);

/* Measurement aid: id / name of the kernel the calling thread's last sc_gemm or sc_gemm_splitk dispatched (names as they appear
 * in rocprofv3's kernel statistics).  bench.py attributes its per-launch HIP-event times to kernels with it. */
int sc_gemm_last_kernel(void);
const char* sc_gemm_kernel_name(int id);

/* Split-K form for wgrad-shaped products (small M x N, long contraction K), bf16 operands, f32 output with ldc == N:
 * C (+)= alpha * opA opB^T.  `workspace` holds splitk partial M x N tiles that a second deterministic kernel sums.
 * colsum (nullable, needs ta): colsum[m] (+)= sum_k A[k][m], the column sums of the transposed A over the contraction --
 * for dW = dz^T x this is the bias gradient of the same nn.Linear (autograd of attention.py:72-79, feed_forward.py:93-101),
 * taken from the tiles the product streams anyway and reduced by the same second kernel (fixed order: deterministic). */
long long sc_gemm_splitk_workspace(int M, int N, int splitk);
int sc_gemm_splitk(int ta, int tb, const void* A, const void* B, float* C, int M, int N, int K, long long lda, long long ldb,
                   float alpha, int splitk, int accumulate, float* colsum, void* workspace, long long ws_bytes,
                   hipStream_t stream);

/* Grouped weight gradients: for g < n, dW[g] (n_out[g] x n_in[g], f32, dense) = dz[g]^T x[g] and -- where db[g] is not null --
 * db[g] (n_out[g], f32) = column sums of dz[g]: the grad_weight / grad_bias halves of the nn.Linear backward of every layer of the
 * transformer path (attention.py:53-59, feed_forward.py:62-84, postnet/token.py:35-37, model/ar_asr.py:224-228, prenet/conv2d.py:171-203),
 * which the reference's autograd runs one layer at a time, computed together: every workgroup owns one output tile of one layer and runs
 * the whole contraction over the tokens (no split-K, no partial sums, deterministic).  dz[g] is (tokens[g], n_out[g]) bf16 with row stride
 * ldz[g], x[g] is (tokens[g], n_in[g]) bf16 with row stride ldx[g].  The nine arrays are HOST arrays of n entries (device pointers and
 * sizes); every problem must satisfy sc_wgrad_grouped_supported (n_out, n_in, ldz, ldx multiples of 8, tokens a multiple of 32,
 * 16-byte aligned pointers).  tile: 0 = choose, 128 / 256 = force the output tile edge.  `workspace` (device memory,
 * sc_wgrad_grouped_workspace(n, n_out, n_in) bytes, 16-byte aligned) receives the schedule: the problems cut into units of at most 16
 * tiles, dealt to the 8 XCDs by work; it is filled by small launches that carry the entries as kernel arguments, so the call makes
 * no host-to-device copy and can be captured into a hipGraph.  conv_k (nullable; entries 0 / 1 = plain): problem g is the im2col product
 * of a 'same' Conv1d with conv_k[g] taps (x[g] columns = tap * C_in + channel) and dW[g] is the nn.Conv1d weight gradient in ITS layout,
 * (n_out, n_in / conv_k, conv_k): the epilogue stores element (o, tap * C + c) at (o, c, tap). */
int sc_wgrad_grouped_supported(int n_out, int n_in, int tokens, long long ldz, long long ldx);
int sc_wgrad_grouped_last_tile(void); /* measurement aid: tile edge the calling thread's last sc_wgrad_grouped call used */
long long sc_wgrad_grouped_workspace(int n, const int* n_out, const int* n_in);
int sc_wgrad_grouped(int n, const void* const* dz, const void* const* x, float* const* dW, float* const* db, const int* n_out,
                     const int* n_in, const int* tokens, const long long* ldz, const long long* ldx, const int* conv_k, int tile,
                     void* workspace, long long ws_bytes, hipStream_t stream);

/* ---- FastSpeech2 glue (speechain/module/decoder/nar_tts.py) ------------------------------------------------------------------
 * Length regulation (nar_tts.py:439-472, a Python loop of repeat_interleave / pad / cat over the batch): out[b][t][:] = x[b][k][:] for the
 * token k whose frames [sum(dur[b][:k]), sum(dur[b][:k+1])) contain t, zeros beyond the last frame; and its adjoint, dx[b][k][:] = sum of
 * dy[b][t][:] over the frames of token k that exist (t < T), in frame order (deterministic).  dur (B, L) f32 holds the rounded,
 * non-negative frame counts the reference computes (proc_duration, :206-230), truncated as its `.long()` does.  L <= 2048.
 * sc_avg_by_duration: FastSpeech2Decoder.average_scalar_by_duration (:150-204): out[b][k] = mean of frame[b][t] over the existing frames
 * of token k, 0 when it has none -- without the reference's (B, L, T) index cube. */
int sc_length_regulate_fwd(int dtype, const void* x, const float* dur, void* out, int B, int L, int T, int D, hipStream_t stream);
int sc_length_regulate_bwd(int dtype, const void* dy, const float* dur, void* dx, int B, int L, int T, int D, hipStream_t stream);
int sc_avg_by_duration(const float* frame, const float* dur, float* out, int B, int L, int T, hipStream_t stream);

/* nn.LayerNorm over the last dim D of an (R, D) matrix (encoder.py:97-98,291-292; decoder.py:90-92; pos_enc.py:94-95);
 * mean/rstd (R,) are saved for the backward pass (may be null).  Backward returns dx and the parameter gradients;
 * `workspace` needs sc_layernorm_bwd_workspace(R, D) bytes.  dres (nullable, (R, D) in `dtype`) is added to dx before the store: in a
 * pre-LN block the input of the LayerNorm also feeds the residual connection (encoder.py:115-141), and its second gradient is summed
 * here instead of in a separate elementwise launch.  `stages`: 3 = everything; 1 = dx and the per-workgroup partial sums only -- the
 * parameter gradients of `n` such calls are then produced together by ONE launch of sc_layernorm_bwd_finish (host arrays of the
 * calls' workspaces, R, D and outputs): nothing downstream reads them before the step ends. */
int sc_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     long long R, int D, float eps, hipStream_t stream);
long long sc_layernorm_bwd_workspace(long long R, int D);
int sc_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                     void* dx, float* dgamma, float* dbeta, long long R, int D, const void* dres, void* workspace,
                     long long ws_bytes, int stages, hipStream_t stream);
/* sc_layernorm_bwd with a second output: dz (R, D) = dx * keepmask / (1 - drop_p), the mask of the dropout site whose output this
 * LayerNorm normalised (stateless generator keyed (*seed, offset + r * D + c), as sc_dropout).  In a pre-LN stack the gradient a
 * LayerNorm backward returns goes straight into the backward pass of the preceding sublayer (encoder.py:115-141), which starts by
 * masking it; produced here, that costs one extra store instead of a launch that re-reads dx.  dz is computed from the ROUNDED dx, so
 * it equals sc_dropout(dx) bit for bit.  bf16, D % 8 == 0, D <= 2048 only (sc_layernorm_bwd_dz_supported). */
int sc_layernorm_bwd_dz_supported(int dtype, int D);
int sc_layernorm_bwd_dz(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        void* dx, float* dgamma, float* dbeta, long long R, int D, const void* dres, void* workspace,
                        long long ws_bytes, int stages, void* dz, float drop_p, const unsigned long long* seed,
                        unsigned long long offset, hipStream_t stream);
int sc_layernorm_bwd_finish(int n, const void* const* workspaces, const long long* R, const int* D, float* const* dgamma,
                            float* const* dbeta, hipStream_t stream);

/* out[n] = sum_r x[r][n] (bias gradients); deterministic two-stage reduction. */
long long sc_colsum_workspace(long long R, int N);
int sc_colsum(int dtype, const void* x, long long R, int N, long long ld, float* out, void* workspace, long long ws_bytes,
              hipStream_t stream);

/* Masked softmax over the last dim of (Z, Tq, ld) scores, Z = B*H (attention.py:88-95): positions where the bool
 * mask (B, Tq|1, Tk; strides mask_b, mask_q; may be null) is false, or k > q when `causal`, get probability 0.
 * P receives the probabilities (the reference's returned `score_soft`), Pd (optional) the dropped-out copy fed to
 * P.V.  Columns [Tk, ld) are written as zeros.  Backward: dS = P * (dP - sum_k dP*P) * out_scale with
 * dP = dPd * keepmask / (1 - p) regenerated from (seed, offset). */
int sc_softmax_fwd(int dtype, const void* S, void* P, void* Pd, int Z, int H, int Tq, int Tk, long long ld,
                   const unsigned char* mask, long long mask_b, long long mask_q, int causal, float drop_p,
                   const unsigned long long* seed, unsigned long long offset, hipStream_t stream);
int sc_softmax_bwd(int dtype, const void* P, const void* dPd, void* dS, long long rows, int Tk, long long ld, float out_scale,
                   float drop_p, const unsigned long long* seed, unsigned long long offset, hipStream_t stream);

/* Fused attention forward (bf16, head size 64, Tk <= 256): O = dropout(softmax(scale * Q K^T + mask)) V without writing
 * scores to HBM; replaces attention.py:88-106,131 when the probability tensor is not requested.  Q/K/V element
 * (b, h, t, d) = ptr[b*x_b + t*ldx + h*64 + d]; O likewise with (o_b, ldo); LSE (B*H, Tq) f32 receives the row
 * logsumexp for the backward pass.  mask / causal / dropout as in sc_softmax_fwd, with the dropout index of element
 * (z, q, k) being offset + (z*Tq + q)*round_up(Tk, 8) + k (identical to the unfused path; offset must be even). */
int sc_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Tq, int Tk,
                     long long q_b, long long ldq, long long kv_b, long long ldk, long long o_b, long long ldo,
                     const unsigned char* mask, long long mask_b, long long mask_q, int causal, float scale, float drop_p,
                     const unsigned long long* seed, unsigned long long offset, hipStream_t stream);

/* Backward of sc_attention_fwd: recomputes the probabilities from Q, K and LSE, regenerates the dropout mask and returns
 * dQ (element (b,h,t,d) at dQ[b*dq_b + t*lddq + h*64 + d]) and dK / dV (same with dkv_b, lddk).  O is the forward output,
 * dO its gradient (both (o_b, ldo)). */
int sc_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                     void* dK, void* dV, int B, int H, int Tq, int Tk, long long q_b, long long ldq, long long kv_b,
                     long long ldk, long long o_b, long long ldo, long long dq_b, long long lddq, long long dkv_b, long long lddk,
                     const unsigned char* mask, long long mask_b, long long mask_q, int causal, float scale, float drop_p,
                     const unsigned long long* seed, unsigned long long offset, float* dq_workspace, long long ws_bytes,
                     hipStream_t stream);
/* bytes of the f32 workspace sc_attention_bwd needs for Tk keys: 0 up to 256 keys; beyond, the keys are swept in blocks of 256, one launch
 * each, and dQ is summed through (B H, Tq, 64) floats in launch order (dq_workspace may be NULL otherwise). */
long long sc_attention_bwd_workspace(int B, int H, int Tq, int Tk);

/* Linear layers of ONE decoding step (csrc/decode_gemm.hip): y (M, N) = act(LayerNorm(x) w^T + bias) + residual in bf16 for the few hundred
 * hypothesis rows of a beam-search step (the reference re-runs its whole decoder per step: speechain/infer_func/beam_search.py:325-333; the
 * layers are those of module/transformer/decoder.py:90-148).  w is an nn.Linear weight (N, K) row-major; bias (N) f32 or null; ln_gamma /
 * ln_beta (K) f32 or both null -- with them, nn.LayerNorm(K, eps) of every row of x is folded in front of the product (needs K <= 512: the
 * whole row on chip); act: 0 none, 1 ReLU, 2 GELU (applied before the residual is added); residual (M, N) bf16 or null.  K a multiple of 32,
 * 16-byte aligned rows.  A 64 x 64 output tile's whole contraction is requested at once instead of in 32-deep pipelined tiles. */
int sc_decode_linear(const void* x, long long ldx, int M, int K, const void* w, long long ldw, int N, const float* bias,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, int act, const void* residual, long long ldr, void* y,
                     long long ldy, hipStream_t stream);
/* The same with the LayerNorm applied through its row statistics: LN(x) W^T + b = rstd (x w_scaled^T - mean colsum) + shift with
 * w_scaled[n][k] = bf16(gamma[k] W[n][k]), colsum[n] = sum_k w_scaled[n][k], shift = W beta + b (made once per weight version by the caller).
 * The product reads the raw rows; the statistics (two-pass variance over the staged rows) enter in the epilogue only.  K <= 512. */
int sc_decode_linear_lnstats(const void* x, long long ldx, int M, int K, const void* w_scaled, long long ldw, int N, const float* colsum,
                             const float* shift, float ln_eps, int act, const void* residual, long long ldr, void* y, long long ldy,
                             hipStream_t stream);

/* Fused attention for any head size that is a multiple of 8 up to 256 (csrc/attention_dh.hip): the reference's attention is head-size
 * agnostic (speechain/module/transformer/attention.py:53-65,88-106,131); FastSpeech2 runs 2 heads of 192
 * (recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:40-43).  Same semantics, dropout indices and saved logsumexp as
 * sc_attention_fwd / sc_attention_bwd with `head_dim` explicit (head h of a row starts at element h * head_dim); masks are (B, 1, Tk) key
 * masks (mask_b = batch stride in bytes, null = none) and / or the causal flag.  The backward pass is three launches -- rowsum(dO * O) into
 * `workspace` (sc_attention_dh_bwd_workspace bytes), dK / dV by key blocks, dQ by query blocks -- with no atomics: deterministic. */
int sc_attention_dh_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Tq, int Tk, int head_dim,
                        long long q_b, long long ldq, long long kv_b, long long ldk, long long o_b, long long ldo,
                        const unsigned char* mask, long long mask_b, int causal, float scale, float drop_p,
                        const unsigned long long* seed, unsigned long long offset, hipStream_t stream);
long long sc_attention_dh_bwd_workspace(int B, int H, int Tq);
int sc_attention_dh_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                        void* dV, int B, int H, int Tq, int Tk, int head_dim, long long q_b, long long ldq, long long kv_b,
                        long long ldk, long long o_b, long long ldo, long long dq_b, long long lddq, long long dkv_b, long long lddk,
                        const unsigned char* mask, long long mask_b, int causal, float scale, float drop_p,
                        const unsigned long long* seed, unsigned long long offset, float* workspace, long long ws_bytes,
                        hipStream_t stream);

/* ---- cached-key/value decoding (csrc/decode.hip): the inference callers of SURVEY.md row f3 --------------------------------
 * Replaces, inside speechain/infer_func/beam_search.py:325-333 + :518-521, the re-run of the whole decoder on the whole prefix at
 * every step and the re-ordering of per-row tensors by beam_idx.
 *
 * sc_beam_table_update: table_out[n][t] = table_in[beam_idx[n]][t] for t < pos (beam_idx NULL: identity), table_out[n][pos] = n;
 *   (N, Tmax) int32 tables, two distinct buffers.  table[n][t] names the cache row whose slot t holds position t of hypothesis n.
 * sc_attention_decode: one step of causal self-attention, heads of 64; dtype 0 = f32, 1 = bf16.  qkv (N, >= 3 d), row stride ldqkv:
 *   this step's [q | k | v]; cache (N, Tmax, 2 d) [k | v]: slot (n, pos) is written, slots (table[n][t], t), t < pos, are read;
 *   out (N, d), row stride ldo: softmax(scale q K^T) V per head (speechain/module/transformer/attention.py:88-106 on one query row). */
int sc_beam_table_update(const int* table_in, int* table_out, const long long* beam_idx, int N, int Tmax, int pos, hipStream_t stream);
/* sc_beam_table_update plus the decoder input rows of the step in the same launch: x[n] = round(emb[tokens[n]] * emb_scale) * xscale +
 * pe * alpha (speechain/module/prenet/embed.py:40-47 lookup and scaling, the cast to the compute type `dtype`, transformer/pos_enc.py:155-163
 * for the one position `pos`: pe is that position's row of the encoding table, alpha the learnt scale or null). */
int sc_decode_embed_step(int dtype, const float* emb, const long long* tokens, float emb_scale, const float* pe, const float* alpha,
                         float xscale, void* x, const int* table_in, int* table_out, const long long* beam_idx, int N, int d, int V, int Tmax,
                         int pos, hipStream_t stream);
int sc_attention_decode(int dtype, const void* qkv, long long ldqkv, void* cache, const int* table, void* out, long long ldo, int N, int H,
                        int Tmax, int pos, float scale, hipStream_t stream);

/* CTC prefix scores for joint CTC / attention decoding (speechain/infer_func/ctc_decoding.py:76-196; Algorithm 2 of Watanabe et al.
 * 2017).  x (B, T, V) f32 CTC log-probabilities, ONE copy per sentence: hypothesis row n belongs to sentence n / beam; enc_lens (B)
 * int32; frames past an utterance count as certain blank (:49-57); "impossible" is the reference's finite -1e20.
 *   sc_ctc_prefix_scores: psi (N = B * beam, V) for every candidate token of every row from the forward variables r_prev (T, 2, N) of
 *     the rows' prefixes; last (N): last label of each prefix (unused when plen == 0); eos_frame (N) int32: frame whose
 *     logaddexp(r_prev[.][0], r_prev[.][1]) is the row's <eos> score (:151-155); psi(., blank) = -1e20 (:158).  T <= 8192 (the rows' phi / r_sum live in LDS:
 *     2 T floats per row, up to 8 rows of a sentence per workgroup).
 *   sc_ctc_prefix_state: r_new (T, 2, N), the forward variables of the surviving candidates -- new row n = prefix of row beam_idx[n]
 *     extended by tokens[n] -- recomputed for those N pairs (the reference gathers them out of a (T, 2, N, V) tensor, :162-196).  With
 *     psi_prev (N, V) and psi_sel (N) the launch also gathers psi_sel[n] = psi_prev[beam_idx[n]][tokens[n]], the score the row was chosen
 *     with (:63; both NULL: not wanted). */
int sc_ctc_prefix_scores(const float* x, const int* enc_lens, const int* eos_frame, const float* r_prev, const long long* last, float* psi,
                         int B, int beam, int T, int V, int plen, int blank, int eos, hipStream_t stream);
/* The same scores with the exponential factorised (exp(phi + x) = exp(phi - a) * exp(x - bmax)): E (B, T, V) = exp(x - bmax) and bmax (B, V) =
 * max over the frames of x are made once per batch by the caller; a step is then one multiply-add per (row, token, frame) instead of
 * one v_exp.  Sums that underflow are redone exactly per (row, token): the result is sc_ctc_prefix_scores' up to f32 rounding of the sum.
 * With V % 4 == 0 the sums run as exact-f32 matrix-core products, up to 16 rows of a sentence per tile (G = ceil(beam / 16) groups per
 * sentence; a beam of 16 fills its tile).
 * pt (B G, 2, T, 16) / pa (B G, 2, 16), both or neither: P = exp(phi - a) and a of the rows' r_sum / r_blank, group by group, as
 * sc_ctc_prefix_state wrote them at prefix length plen - 1 for THESE rows (r_prev is that call's r_new; rows a group does not hold are never
 * read into a result); NULL: derived from r_prev inside the launch (same bits). */
int sc_ctc_prefix_scores_fast(const float* x, const float* E, const float* bmax, const int* enc_lens, const int* eos_frame, const float* r_prev,
                              const long long* last, float* psi, int B, int beam, int T, int V, int plen, int blank, int eos, const float* pt,
                              const float* pa, hipStream_t stream);
/* (pt / pa, both or neither: see sc_ctc_prefix_scores_fast -- the next position's launch then copies them instead of
 *  deriving them in each of its workgroups) */
int sc_ctc_prefix_state(const float* x, const int* enc_lens, const float* r_prev, const long long* last, const long long* beam_idx,
                        const long long* tokens, float* r_new, int B, int beam, int T, int V, int plen, int blank, const float* psi_prev, float* psi_sel,
                        float* pt, float* pa, hipStream_t stream);

/* Row-wise top-K of a (R, L) f32 matrix, row stride ld: vals / idx (R, K) in descending order, ties by the lower index; K <= 64.
 * The candidate selection of a beam-search step (speechain/infer_func/beam_search.py:419-421: torch.topk over (batch, beam * vocab)). */
int sc_topk_rows(const float* x, long long ld, int R, int L, int K, float* vals, long long* idx, hipStream_t stream);
/* Second stage of the same selection over G groups of `parts` consecutive rows (beam_search.py:401-410 takes the top 2 * beam of a
 * sentence's beam * vocabulary candidates): from sc_topk_rows' per-row results (G * parts, K) to (G, K) with idx = row-in-group * L +
 * column -- identical to sc_topk_rows on the (G, parts * L) view, ties included.  parts * K <= 1024. */
/* One launch for the step's device-to-host record: pack (B, 2 K [+ 2 beam]) float64 = [scores | indices | <eos> scores | best other scores]
 * per sentence (infer_func/beam_search.py:401-410,455-462 read these on the host). */
int sc_beam_pack(const float* vals, const long long* idx, const float* eos, const float* other, int B, int K, int beam, double* pack,
                 hipStream_t stream);
/* sc_beam_pack plus the next step's rows made on the device (beam_search.py:423-509: a running sentence keeps its first `beam` candidates in
 * rank order whose token is not <sos/eos>): scores / tokens / beam_idx (B * beam) are overwritten, prev_tokens receives this step's tokens.
 * The host still reads the packed record for the n-best lists and the stopping rules, one step behind the device.  2 beam <= K <= 128. */
int sc_beam_advance(const float* vals, const long long* idx, const float* eos, const float* other, int B, int K, int beam, int V,
                    int eos_token, double* pack, float* scores, long long* tokens, long long* prev_tokens, long long* beam_idx,
                    hipStream_t stream);
/* sc_topk_merge over the B sentences (parts = beam) and sc_beam_advance in ONE launch: vals1 / idx1 (B * beam, K) are sc_topk_rows' results on
 * the (B * beam, V) candidate scores; vals / idx (B, K) receive the merged selection too.  2 beam <= K <= 64, beam * K <= 1024. */
int sc_beam_merge_advance(const float* vals1, const long long* idx1, int B, int beam, int K, int V, const float* eos, const float* other,
                          int eos_token, float* vals, long long* idx, double* pack, float* scores, long long* tokens,
                          long long* prev_tokens, long long* beam_idx, hipStream_t stream);
int sc_topk_merge(const float* vals1, const long long* idx1, int G, int parts, int K, int L, float* vals, long long* idx, hipStream_t stream);

/* Candidate scores of a beam-search step (speechain/infer_func/beam_search.py:334-345,401-410) in one pass over the token head's logits:
 *   cand[n][v] = (1 - ctc_weight) * log_softmax(logits[n] / temperature)[v] + ctc_weight * (psi[n][v] - psi_prev[n]) + beam_score[n],
 * the blank column of the attention term forced to -1e20 when psi is given (:339).  logits (N, V) f32 / bf16 (dtype 0 / 1), row stride ld;
 * psi (N, V), psi_prev (N) may be NULL; eos_score / other_max (N) (the mixed score of <eos> and the best other one, for <eos> filtering
 * :455-462) may be NULL. */
int sc_beam_candidates(int dtype, const void* logits, long long ld, const float* psi, const float* psi_prev, const float* beam_score,
                       float* cand, float* eos_score, float* other_max, int N, int V, float temperature, float ctc_weight, int blank, int eos,
                       hipStream_t stream);
/* sc_beam_candidates + sc_topk_rows on its result in ONE launch, the (N, V) score matrix kept in LDS row by row and never written:
 * vals / idx (N, K) = every row's K best candidate scores, descending, ties by the lower column.  V <= 12288, K <= 64. */
int sc_beam_candidates_topk(int dtype, const void* logits, long long ld, const float* psi, const float* psi_prev, const float* beam_score,
                            float* eos_score, float* other_max, int N, int V, float temperature, float ctc_weight, int blank, int eos,
                            int K, float* vals, long long* idx, hipStream_t stream);

/* Label-smoothed token cross entropy over (R, V) logits (speechain/criterion/cross_entropy.py:132-146):
 * row_loss[r] = -((1-eps) * lp[target] + eps/V * sum_v lp[v]) with lp = log_softmax(logits / temperature);
 * lse[r] is kept for the backward pass, which writes dlogits = grow[r]/temperature * (softmax - (1-eps)*onehot - eps/V). */
int sc_ce_fwd(int dtype, const void* logits, const long long* target, long long R, int V, long long ld, float temperature,
              float smoothing, float* row_loss, float* lse, hipStream_t stream);
int sc_ce_bwd(int dtype, const void* logits, const long long* target, const float* lse, const float* grow, long long R, int V,
              long long ld, float temperature, float smoothing, void* dlogits, hipStream_t stream);

/* The reduction of the per-token losses to the criterion's scalar (cross_entropy.py:201-212, no token weights, no confidence filter) in one
 * launch: loss = mean_b( sum_{l < len[b]} rows[b][l] / (len[b] + 1e-10) ) (length_normalized = 0: without the division); and
 * grow[b][l] = g / (B (len[b] + 1e-10)) for l < len[b], else 0, for an upstream scalar g on the device. */
int sc_seq_loss_fwd(const float* rows, const long long* len, int B, int L, int length_normalized, float* loss, hipStream_t stream);
int sc_seq_loss_bwd(const float* g, const long long* len, int B, int L, int length_normalized, float* grow, hipStream_t stream);

/* LeastError (speechain/criterion/least_error.py:96-129): masked L1 / L2 / L1+L2 of pred (B, T, D) f32 / bf16 against tgt (B, T, D) f32
 * over the frames t < len[b] (int64): loss (1) f32 = sum_valid mean_d f(pred - tgt) / (number of valid frames when `normalized`, else B).
 * workspace: sc_least_error_workspace() bytes.  _bwd: dpred (B, T, D) in pred's dtype for an upstream scalar g (device), zero past the
 * lengths.  dtype 0 = f32, 1 = bf16; loss_type 1 = L1, 2 = L2, 3 = L1 + L2. */
long long sc_least_error_workspace(void);
int sc_least_error_fwd(int dtype, const void* pred, const float* tgt, const long long* len, int B, int T, int D, int loss_type, int normalized,
                       void* workspace, float* loss, hipStream_t stream);
int sc_least_error_bwd(int dtype, const void* pred, const float* tgt, const long long* len, int B, int T, int D, int loss_type, int normalized,
                       const float* g, void* dpred, hipStream_t stream);
/* out (1) f32 = sum_{b,t,d} a[b][t][d] * tab[t][d] for a (B, T, D) f32 / bf16 and tab (T, D) f32: the gradient of the scalar weight alpha of
 * ScaledPositionalEncoding (speechain/module/transformer/pos_enc.py: x + alpha * table) from the gradient of the sum.  Deterministic.
 * workspace: sc_least_error_workspace() bytes. */
int sc_dot_table(int dtype, const void* a, const float* tab, int B, int T, int D, void* workspace, float* out, hipStream_t stream);

/* lse[r] = logsumexp_v logits[r][v] */
int sc_row_lse(int dtype, const void* logits, long long R, int V, long long ld, float* lse, hipStream_t stream);

/* CTC negative log-likelihood per utterance and its gradient w.r.t. the logits (speechain/criterion/ctc.py:48-65 =
 * log_softmax + F.ctc_loss(reduction='none')).  logits (B, Tmax, V); lse (B, Tmax) from sc_row_lse; tgt holds label
 * ids with row stride tgt_stride; alpha (forward, kept) and beta (backward scratch) are (B, Tmax, Smax) float workspaces,
 * Smax >= 2*max(tgt_len)+1; V <= 65536.
 * nll may be +inf (infeasible alignment); the caller applies zero_infinity through gscale[b] (= upstream/B, or 0). */
int sc_ctc_fwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
               long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, float* alpha,
               float* nll, hipStream_t stream);
int sc_ctc_bwd(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
               long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, const float* alpha,
               const float* nll, const float* gscale, float* beta, void* grad, hipStream_t stream);

/* The same in two launches for a training step: sc_ctc_fwd_train runs the alpha AND the beta recursion in one launch of 2 B workgroups (they
 * are independent, and each is one serial chain of frames per utterance), sc_ctc_grad then only writes the gradient rows from both. */
int sc_ctc_fwd_train(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                     long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, float* alpha,
                     float* beta, float* nll, hipStream_t stream);
int sc_ctc_grad(int dtype, const void* logits, const float* lse, const long long* in_len, const long long* tgt,
                long long tgt_stride, const long long* tgt_len, int B, int Tmax, int V, int Smax, int blank, const float* alpha,
                const float* beta, const float* nll, const float* gscale, void* grad, hipStream_t stream);

/* dtype casts (0 = f32, 1 = bf16) and the transposed cast out[c][r] = in[r][c] of an (R, C) f32 matrix. */
int sc_cast(int dtype_in, int dtype_out, const void* in, void* out, long long n, hipStream_t stream);
/* Zero-padded row image of a (B, T, C) bf16 tensor: rows_total rows of C, utterance b at rows b (T + 2 r) .. with r zero rows on either side,
 * zero rows behind the last utterance.  With it a 'same' Conv1d (speechain/module/prenet/conv1d.py:Conv1dEv, the conv-type feed-forward of
 * speechain/module/transformer/feed_forward.py:71-95) is ONE sc_gemm over overlapping rows (lda = C < K = k C) instead of im2col + GEMM. */
int sc_pad_rows(const void* x, int B, int T, int C, int r, void* out, long long rows_total, hipStream_t stream);
/* Only the zero rows of that image (2 r per utterance + the tail): for a producer that writes the data rows itself -- an sc_gemm with
 * C = out + r * C_row, batch stride sC1 = (T + 2 r) * C_row. */
int sc_zero_pad_rows(int B, int T, int C, int r, void* out, long long rows_total, hipStream_t stream);
/* dW (V, D) f32 of out = table[ids] from dy (N, D; row stride ld): autograd of the token embedding (speechain/module/prenet/embed.py:55-66).
 * Deterministic, no atomics (the first position holding an id sums all of that id's rows in position order); the row of padding_idx
 * (nn.Embedding's) stays zero; pass a negative padding_idx for none.  ids in [0, V). */
int sc_embedding_bwd(int dtype, const long long* ids, const void* dy, long long ld, int N, int D, int V, long long padding_idx, float* dw,
                     hipStream_t stream);
int sc_cast_transpose(int dtype_out, const float* in, void* out, int R, int C, long long ld_out, hipStream_t stream);

/* Batched refresh of compute-dtype weight copies in ONE launch.  `desc` is a device array of n_desc records
 *   struct sc_cast_desc { const float* src; void* dst; void* dst_t; int rows, cols; long long ld_dst, ld_dst_t; int tile0, tiles_x;
 *                         int conv_k, tr_rev; };
 * (64 bytes): record i owns tiles [tile0_i, tile0_{i+1}) of 64 x 64 elements of its (rows, cols) float32 matrix,
 * tiles_x = ceil(cols/64); each tile is written as `dtype_out` to dst (row stride ld_dst) and/or transposed to dst_t
 * (row stride ld_dst_t); null pointers are skipped.  total_tiles = sum of all records' tiles.  conv_k > 1: src is an nn.Conv1d weight
 * (rows, cols / conv_k, conv_k) (prenet/conv1d.py:99-108) read as the GEMM matrix of the 'same' convolution, column = tap * C_in + channel;
 * tr_rev = 1: row (tap, channel) of the transposed copy goes to row channel * conv_k + (conv_k - 1 - tap) (the data-gradient matrix of
 * the im2col-free convolution, functional._ConvLinear). */
int sc_cast_weights(int dtype_out, const void* desc, int n_desc, int total_tiles, hipStream_t stream);

/* y = dropout(x * xscale + table[t] * (*alpha)) for x (B, T, D), table (>= T, D) f32 (pos_enc.py:168-182). */
int sc_posenc_fwd(int dtype, const void* x, const float* table, const float* alpha, float xscale, void* y, long long B, int T,
                  int D, float drop_p, const unsigned long long* seed, unsigned long long offset, hipStream_t stream);

/* y = x * scale * keepmask / (1 - p) with the stateless generator (seed, offset + index). */
int sc_dropout(int dtype, const void* x, void* y, long long n, float scale, float drop_p, const unsigned long long* seed,
               unsigned long long offset, hipStream_t stream);

/* dpre = dy * act'(pre) * keepmask / (1 - p): backward of dropout(act(.)) when the upstream gradient is not a GEMM result. */
int sc_act_bwd(int dtype, const void* dy, const void* pre, void* dpre, long long n, int act, float drop_p,
               const unsigned long long* seed, unsigned long long offset, hipStream_t stream);

/* ---- optimizer tail: global-norm clipping + Adam over flat buffers -------------------------------------------------------------------
 * Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step of the reference's training step (speechain/optim_sche/abs.py:232-253; Adam
 * betas / eps from the recipes, e.g. recipes/asr/librispeech/train-clean-100/exp_cfg/100-bpe5k_transformer-large_lr2e-3.yaml:251-261).
 * `nb` flat float32 buffers of `sizes[k]` elements each (the gradient buckets; parameters and both moments in buffers of the same
 * layout); the pointer arrays are HOST arrays of device pointers, read at call time.
 * state (device, 3 floats): [0] = ||g||^2 over all buffers, [1] = clip coefficient min(1, max_norm / (||g|| + 1e-6)) (1 when
 * max_norm <= 0), [2] = step count -- sc_grad_sumsq writes [0], [1] and adds 1 to [2]; sc_adam_step reads [1], [2] and *lr:
 *   g' = g * state[1];  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). */
long long sc_grad_sumsq_workspace(int nb, const long long* sizes);
int sc_grad_sumsq(int nb, const float* const* grads, const long long* sizes, float max_norm, float* state, void* workspace,
                  long long ws_bytes, hipStream_t stream);
int sc_adam_step(int nb, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const long long* sizes, const float* lr, float beta1, float beta2, float eps, const float* state, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHAIN_AMD_H */
