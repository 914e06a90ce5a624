"""Waveform -> linear (power / magnitude) spectrogram on the MI355X.

Drop-in for ``speechain.module.frontend.speech2linear.Speech2LinearSpec`` (reference file
speechain/module/frontend/speech2linear.py): same constructor kwargs (:20-37), same buffer
``preemph_filter.weight`` (1,1,2) (:103-116), same ``forward(speech, speech_len)`` returns (:142-250).
All arithmetic is one launch of ``sc_frontend_logmel_fwd`` (csrc/frontend.hip).

Differences kept on purpose: ``speech_len`` is NOT mutated in place (the reference adds the centre padding to the
caller's tensor, :202); the output does not carry autograd history (the frontend has no trainable parameter).
"""
import math
import os

import torch

from speechain_amd._lib import lib, ptr, require_gpu, stream
from speechain_amd.module.abs import Module


class _BufferOnly(torch.nn.Module):
    """Holds one non-trainable tensor under the name ``weight`` (state_dict key ``<name>.weight``)."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.register_buffer("weight", weight)


def _to_samples(value, sr: int) -> int:
    return int(value * sr) if isinstance(value, float) else value


class StftPlan:
    """Device-resident windowed-DFT basis for (window, win_length, n_fft); built lazily per device."""

    def __init__(self, window: str, win_length: int, n_fft: int, hop: int):
        self.window, self.win_length, self.n_fft, self.hop = window, win_length, n_fft, hop
        self._basis = {}

    def basis(self, device: torch.device) -> torch.Tensor:
        key = (device.type, device.index)
        if key not in self._basis:
            win = getattr(torch, f"{self.window}_window")(self.win_length, dtype=torch.float32, device=device)
            basis = torch.empty(lib.value("sc_frontend_basis_floats", self.n_fft), dtype=torch.float32, device=device)
            lib.call("sc_frontend_basis_init", ptr(win), self.win_length, self.n_fft, self.hop, ptr(basis), stream())
            self._basis[key] = basis
        return self._basis[key]

    def radix4(self, device: torch.device):
        """(window padded to n_fft as torch.stft pads it, unwindowed (n_fft / 4)-point basis, twiddles (4, n_fft / 2 + 1, 2)) for the radix-4
        form of the wide transform (csrc/frontend.hip: sc_frontend_frames4 / sc_frontend_spec_finish4), built lazily per device."""
        key = ("r4", device.type, device.index)
        if key not in self._basis:
            n, q = self.n_fft, self.n_fft // 4
            win = getattr(torch, f"{self.window}_window")(self.win_length, dtype=torch.float32, device=device)
            left = (n - self.win_length) // 2
            wpad = torch.nn.functional.pad(win, (left, n - self.win_length - left)).contiguous()
            ones = torch.ones(q, dtype=torch.float32, device=device)
            bq = torch.empty(lib.value("sc_frontend_basis_floats", q), dtype=torch.float32, device=device)
            lib.call("sc_frontend_basis_init", ptr(ones), q, q, self.hop, ptr(bq), stream())
            r = torch.arange(4, dtype=torch.float64)[:, None]
            k = torch.arange(n // 2 + 1, dtype=torch.float64)[None, :]
            ang = 2.0 * math.pi * ((r * k) % n) / n
            tw = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(device).contiguous()  # (float64: the finish kernel sums in double)
            self._basis[key] = (wpad, bq, tw)
        return self._basis[key]

    def synthesis(self, device: torch.device):
        """(window, inverse basis) for the overlap-add inverse transform (csrc/griffinlim.hip), built lazily per device."""
        key = ("inv", device.type, device.index)
        if key not in self._basis:
            win = getattr(torch, f"{self.window}_window")(self.win_length, dtype=torch.float32, device=device)
            nc = 32 * lib.value("sc_frontend_num_tiles", self.n_fft)
            inv = torch.empty(nc, (self.n_fft + 1) & ~1, dtype=torch.float32, device=device)
            lib.call("sc_istft_basis_init", ptr(win), self.win_length, self.n_fft, ptr(inv), stream())
            self._basis[key] = (win, inv)
        return self._basis[key]


def run_frontend(speech: torch.Tensor, speech_len: torch.Tensor, *, plan: StftPlan, hop: int, center: bool,
                 preemphasis, mag_spec: bool, want_linear: bool, lin_clamp: float = 1e-10, lin_logging: bool = False,
                 lin_log_base=None, want_energy: bool = False, melw: torch.Tensor = None, bands=None,
                 clamp: float = 1e-10, logging: bool = True, log_base=10.0):
    """Shared launcher of the fused frontend kernel.  Returns (mel|None, linear|None, energy|None, feat_len)."""
    if speech.dim() == 3:
        if speech.size(-1) != 1:
            raise RuntimeError(
                "Currently, we don't support multi-channel speech waveforms. If the speech is given in 3D vectors, "
                f"the last dimension must be 1. But got speech.shape={tuple(speech.shape)}.")
        speech = speech.squeeze(-1)
    if speech.dim() != 2:
        raise RuntimeError(f"speech must be (batch, samples[, 1]) but got shape {tuple(speech.shape)}")
    require_gpu(speech, "the acoustic frontend")
    wave = speech if (speech.dtype == torch.float32 and speech.stride(1) == 1) else speech.float().contiguous()
    B, T_w = wave.shape
    dev = wave.device
    wave_len = speech_len.to(device=dev, dtype=torch.int64)
    n_fft = plan.n_fft
    pad = n_fft // 2 if center else 0
    if T_w + 2 * pad < n_fft or (center and pad >= T_w):
        raise RuntimeError(f"waveforms of {T_w} samples are too short for n_fft={n_fft}")
    T = (T_w + 2 * pad - n_fft) // hop + 1
    n_bins = n_fft // 2 + 1
    mel = lin = energy = None
    n_mels = 0
    lo = hi = image = None
    if melw is not None:
        n_mels = melw.size(0)
        mel = torch.empty(B, T, n_mels, dtype=torch.float32, device=dev)
        lo, hi, image = bands if len(bands) == 3 else (*bands, None)
    if want_linear:
        lin = torch.empty(B, T, n_bins, dtype=torch.float32, device=dev)
    if want_energy:
        energy = torch.empty(B, T, dtype=torch.float32, device=dev)
    feat_len = torch.empty(B, dtype=torch.int64, device=dev)
    log_div = 1.0 if log_base is None else math.log(log_base)
    lin_log_div = 1.0 if lin_log_base is None else math.log(lin_log_base)
    if n_fft >= 512 and n_fft % 64 == 0 and os.environ.get("SC_FRONTEND_GEMM", "1") != "0" and os.environ.get("SC_FRONTEND_RADIX4", "0") == "1":
        # OPT-IN (SC_FRONTEND_RADIX4=1): the wide transform in radix-4 form -- four (n_fft / 4)-point transforms per frame as ONE exact-f32 GEMM,
        # twiddled sum (in double) in the finish kernel: a quarter of the flops of the direct product below, FastSpeech2 step 9.92 -> 9.67 ms.
        # Not the default because it has an FFT's accuracy, not the direct product's: a bin 100 dB below the frame's energy becomes the small
        # difference of four large float32 sub-transforms -- 8.1e-3 in log-mel on the fixture's tone-plus-floor utterance, where torch's own
        # float32 FFT is 6.3e-3 from the exact value and the direct product 1e-4; the golden test holds the default path to 1e-3 + the
        # reference's own deviation (tests/test_frontend_gpu.py).
        q = n_fft // 4
        ncq = 32 * lib.value("sc_frontend_num_tiles", q)
        wpad, bq, tw = plan.radix4(dev)
        frames4 = torch.empty(B * T * 4, q, dtype=torch.float32, device=dev)
        lib.call("sc_frontend_frames4", ptr(wave), ptr(wave_len), B, T_w, wave.stride(0), 1 if preemphasis is not None else 0,
                 float(preemphasis or 0.0), n_fft, hop, 1 if center else 0, T, ptr(wpad), ptr(frames4), stream())
        spec4 = torch.empty(B * T * 4, ncq, dtype=torch.float32, device=dev)
        lib.call("sc_gemm", 0, 0, 1, ptr(frames4), ptr(bq), ptr(spec4), B * T * 4, ncq, q, q, ncq, ncq, 1, 1, 0, 0, 0, 0, 0, 0, 1.0,
                 None, 0, None, None, 0, 0, 0.0, None, 0, 0, 0, stream())
        lib.call("sc_frontend_spec_finish4", ptr(spec4), ptr(tw), ptr(wave_len), B, T_w, n_fft, hop, 1 if center else 0, ptr(melw), ptr(lo), ptr(hi),
                 n_mels, 1 if mag_spec else 0, float(clamp), 1 if logging else 0, log_div, ptr(mel), ptr(energy), ptr(lin), float(lin_clamp),
                 1 if lin_logging else 0, lin_log_div, ptr(feat_len), T, stream())
        return mel, lin, energy, feat_len
    if n_fft >= 512 and n_fft % 2 == 0 and os.environ.get("SC_FRONTEND_GEMM", "1") != "0":
        # wide transform: framing, the library's exact-f32 GEMM against the windowed-DFT basis, finish (csrc/frontend.hip)
        nc = 32 * lib.value("sc_frontend_num_tiles", n_fft)
        frames = torch.empty(B * T, n_fft, dtype=torch.float32, device=dev)
        lib.call("sc_frontend_frames", ptr(wave), ptr(wave_len), B, T_w, wave.stride(0), 1 if preemphasis is not None else 0,
                 float(preemphasis or 0.0), n_fft, hop, 1 if center else 0, T, ptr(frames), stream())
        spec = torch.empty(B * T, nc, dtype=torch.float32, device=dev)
        lib.call("sc_gemm", 0, 0, 1, ptr(frames), ptr(plan.basis(dev)), ptr(spec), B * T, nc, n_fft, n_fft, nc, nc, 1, 1, 0, 0, 0, 0, 0, 0, 1.0,
                 None, 0, None, None, 0, 0, 0.0, None, 0, 0, 0, stream())
        lib.call("sc_frontend_spec_finish", ptr(spec), ptr(wave_len), B, T_w, n_fft, hop, 1 if center else 0, ptr(melw), ptr(lo), ptr(hi),
                 n_mels, 1 if mag_spec else 0, float(clamp), 1 if logging else 0, log_div, ptr(mel), ptr(energy), ptr(lin), float(lin_clamp),
                 1 if lin_logging else 0, lin_log_div, ptr(feat_len), T, stream())
        return mel, lin, energy, feat_len
    lib.call("sc_frontend_logmel_fwd", ptr(wave), ptr(wave_len), B, T_w, wave.stride(0),
             1 if preemphasis is not None else 0, float(preemphasis or 0.0), n_fft, hop, 1 if center else 0,
             ptr(plan.basis(dev)), ptr(melw), ptr(lo), ptr(hi), n_mels, 1 if mag_spec else 0, float(clamp),
             1 if logging else 0, 1.0 if log_base is None else math.log(log_base), ptr(mel), ptr(energy), ptr(lin),
             float(lin_clamp), 1 if lin_logging else 0, 1.0 if lin_log_base is None else math.log(lin_log_base),
             ptr(feat_len), T, ptr(image), stream())
    return mel, lin, energy, feat_len


class Speech2LinearSpec(Module):
    def module_init(self, hop_length, win_length, sr: int = 16000, n_fft: int = None, preemphasis: float = None,
                    pre_stft_norm: str = None, window: str = "hann", center: bool = True, normalized: bool = False,
                    onesided: bool = True, mag_spec: bool = False, return_energy: bool = False, clamp: float = 1e-10,
                    logging: bool = False, log_base: float = None):
        hop_length, win_length = _to_samples(hop_length, sr), _to_samples(win_length, sr)
        if n_fft is None:
            n_fft = win_length
        if not onesided or normalized:
            raise NotImplementedError("speechain_amd frontend implements the onesided, un-normalised STFT every "
                                      "SpeeChain recipe uses (onesided=True, normalized=False).")
        if pre_stft_norm not in (None, "min_max"):
            # 'mean_std' only broadcasts for batch size 1 in the reference (speech2linear.py:179); not carried over
            raise ValueError(f"pre_stft_norm must be None or 'min_max', got {pre_stft_norm}")
        self.output_size = n_fft // 2 + 1
        self.win_length, self.hop_length, self.n_fft, self.sr = win_length, hop_length, n_fft, sr
        self.preemphasis = preemphasis
        if preemphasis is not None:
            self.preemph_filter = _BufferOnly(torch.tensor([-preemphasis, 1.0]).reshape(1, 1, 2))
        self.pre_stft_norm = pre_stft_norm
        self.stft_config = dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=center,
                                normalized=normalized, onesided=onesided, return_complex=True)
        self.mag_spec, self.return_energy = mag_spec, return_energy
        self.clamp, self.logging, self.log_base = clamp, logging, log_base
        self._plan = StftPlan(window, win_length, n_fft, hop_length)

    def _prepare(self, speech: torch.Tensor, speech_len: torch.Tensor):
        """Optional min-max pre-normalisation needs the pre-emphasised signal first (speech2linear.py:167-185);
        that rare path runs un-fused.  Returns (speech, preemphasis coefficient still to be applied)."""
        if self.pre_stft_norm is None:
            return speech, self.preemphasis
        require_gpu(speech, "the acoustic frontend")
        x = speech.squeeze(-1) if speech.dim() == 3 else speech
        if self.preemphasis is not None:
            prev = torch.nn.functional.pad(x, (1, 0))[:, :-1]
            x = x - self.preemphasis * prev
            idx = torch.arange(x.size(1), device=x.device)[None, :]
            x = x.masked_fill(idx >= speech_len.to(x.device).reshape(-1, 1), 0.0)
        lo, hi = x.min(dim=1, keepdim=True)[0], x.max(dim=1, keepdim=True)[0]
        return (x - lo) / (hi - lo) * 2 - 1, None

    def frame_count(self, speech_len: torch.Tensor) -> torch.Tensor:
        """Number of frames of utterances of `speech_len` samples, on the tensor's own device (speech2linear.py:201-212) -- for callers
        that keep the lengths on the host as well (the kernel writes the same numbers to the device-side ``feat_len`` it returns)."""
        n = speech_len + (2 * (self.n_fft // 2) if self.stft_config["center"] else 0)
        return torch.div(n - self.n_fft, self.hop_length, rounding_mode="floor") + 1

    def forward(self, speech: torch.Tensor, speech_len: torch.Tensor):
        speech, preemph = self._prepare(speech, speech_len)
        _, lin, energy, feat_len = run_frontend(
            speech, speech_len, plan=self._plan, hop=self.hop_length, center=self.stft_config["center"],
            preemphasis=preemph, mag_spec=self.mag_spec, want_linear=True, lin_clamp=self.clamp,
            lin_logging=self.logging, lin_log_base=self.log_base, want_energy=self.return_energy)
        if self.return_energy:
            return lin, feat_len, energy, feat_len
        return lin, feat_len

    def recover(self, feat: torch.Tensor, feat_len: torch.Tensor, inv_preemph_winlen: int = 100, n_iter: int = 32,
                momentum: float = 0.99, rand_init: bool = True, _init_angles: torch.Tensor = None):
        """Waveforms from (un-logged) power / magnitude spectrograms by the Griffin-Lim iteration (speech2linear.py:252-351): feat
        (B, T, n_fft // 2 + 1) -> (wav (B, hop (T - 1), 1), wav_len = (feat_len - 1) * hop), then the sliding-window inverse pre-emphasis
        and silence beyond each utterance's length.  The reference calls ``torchaudio.transforms.GriffinLim`` with its defaults (32
        iterations, momentum 0.99, random initial phases: re, im uniform in [0, 1)); ``n_iter`` / ``momentum`` / ``rand_init`` are those
        defaults, ``_init_angles`` (B, T, F) complex fixes the start for tests."""
        assert self.pre_stft_norm is None, "waveform pre-stft normalization cannot be recovered for TTS synthesis."
        wav = self._griffin_lim(feat, n_iter, momentum, rand_init, _init_angles)
        B, L = wav.shape
        dev, hop = wav.device, self.hop_length
        wav_len = (feat_len.to(dev) - 1) * hop
        assert int(wav_len.max()) == wav.size(1), "Something wrong happens when calculating the length of synthetic utterances."
        if self.preemphasis is not None:  # X[n] = sum_k 0.97^k Y[n - k] over a window of inv_preemph_winlen samples (:301-345)
            out = torch.empty_like(wav)
            lib.call("sc_fir_causal", ptr(wav), wav.stride(0), ptr(out), out.stride(0), B, L, float(self.preemphasis), int(inv_preemph_winlen),
                     stream())
            wav = out
        idx = torch.arange(L, device=dev)[None, :]
        wav = wav.masked_fill(idx >= wav_len[:, None], 0.0)
        return wav.unsqueeze(-1), wav_len

    def _griffin_lim(self, feat: torch.Tensor, n_iter: int, momentum: float, rand_init: bool, _init_angles: torch.Tensor = None):
        """(B, hop (T - 1)) waveforms whose magnitude spectrograms approach ``feat`` (torchaudio.functional.griffinlim's iteration)."""
        require_gpu(feat, "Griffin-Lim recovery")
        if not self.stft_config["center"]:
            raise NotImplementedError("speechain_amd: Griffin-Lim recovery is built for center=True (every SpeeChain preset)")
        B, T, F = feat.shape
        n_fft, hop, win = self.n_fft, self.hop_length, self.win_length
        assert F == n_fft // 2 + 1 and n_fft % 2 == 0, (F, n_fft)
        dev = feat.device
        mag = feat.float().clamp_min(0.0)
        mag = (mag if self.mag_spec else mag.sqrt()).reshape(B * T, F).contiguous()  # specgram.pow(1 / power)
        L, M, nc = hop * (T - 1), B * T, 32 * lib.value("sc_frontend_num_tiles", n_fft)
        basis = self._plan.basis(dev)
        window, inv_basis = self._plan.synthesis(dev)
        if _init_angles is not None:
            a_re, a_im = _init_angles.real.float().reshape(M, F).contiguous(), _init_angles.imag.float().reshape(M, F).contiguous()
        elif rand_init:
            a = torch.rand(2, M, F, device=dev)
            a_re, a_im = a[0], a[1]
        else:
            a_re, a_im = torch.ones(M, F, device=dev), torch.zeros(M, F, device=dev)
        spec = torch.empty(M, nc, dtype=torch.float32, device=dev)
        rebuilt, tprev = torch.empty_like(spec), torch.zeros_like(spec)
        frames = torch.empty(M, n_fft, dtype=torch.float32, device=dev)
        wav = torch.empty(B, L, dtype=torch.float32, device=dev)
        full_len = torch.full((B,), L, dtype=torch.int64, device=dev)
        coef = momentum / (1.0 + momentum)

        def gemm(a, b, c, n, k):  # c (M, n) = a (M, k) x b (k, n), exact f32
            lib.call("sc_gemm", 0, 0, 1, ptr(a), ptr(b), ptr(c), M, n, k, k, n, n, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, None, None, 0, 0, 0.0,
                     None, 0, 0, 0, stream())

        def istft():
            gemm(spec, inv_basis, frames, n_fft, nc)
            lib.call("sc_istft_overlap_add", ptr(frames), ptr(window), win, n_fft, hop, B, T, ptr(wav), L, wav.stride(0), stream())

        lib.call("sc_gl_phase_update", 0, ptr(mag), ptr(a_re), ptr(a_im), None, None, ptr(spec), M, n_fft, 0.0, stream())
        for _ in range(n_iter):
            istft()
            lib.call("sc_frontend_frames", ptr(wav), ptr(full_len), B, L, wav.stride(0), 0, 0.0, n_fft, hop, 1, T, ptr(frames), stream())
            gemm(frames, basis, rebuilt, nc, n_fft)
            lib.call("sc_gl_phase_update", 1, ptr(mag), None, None, ptr(rebuilt), ptr(tprev), ptr(spec), M, n_fft, float(coef), stream())
        istft()
        return wav

    def get_sample_rate(self):
        return self.sr

    def __repr__(self) -> str:
        s = (f"{self.__class__.__name__}(\nwin_length={self.win_length}, hop_length={self.hop_length}, "
             f"n_fft={self.n_fft}, mag_spec={self.mag_spec}, ")
        if self.preemphasis is not None:
            s += f"preemphasis={self.preemphasis}, "
        if self.pre_stft_norm is not None:
            s += f"pre_stft_norm={self.pre_stft_norm}, "
        return s + "\n)"
