"""Linear spectrogram -> (log-)mel spectrogram.

Drop-in for ``speechain.module.frontend.linear2mel.LinearSpec2MelSpec`` (speechain/module/frontend/linear2mel.py):
same kwargs (:20-33), the filterbank kept as the buffer ``mel_fbanks.weight`` (n_mels, n_fft//2+1) (:146-155),
``forward(feat, feat_len)`` = filterbank, clamp, log / ln(base) (:163-189).
"""
import math

import torch

from speechain_amd._lib import lib, ptr, require_gpu, stream
from speechain_amd.module.abs import Module
from speechain_amd.module.frontend.melscale import mel_bands, mel_filterbank
from speechain_amd.module.frontend.speech2linear import _BufferOnly


class LinearSpec2MelSpec(Module):
    def module_init(self, n_fft: int, n_mels: int, sr: int = 16000, fmin: float = 0.0, fmax: float = None,
                    clamp: float = 1e-10, logging: bool = True, log_base: float = 10.0, mel_scale: str = "slaney",
                    mel_norm: bool = True, mag_spec: bool = False):
        self.sr = sr
        self.stft_dim = n_fft // 2 + 1
        self.n_mels = n_mels
        self.fmin = fmin
        self.fmax = fmax if fmax is not None else sr // 2
        assert self.fmax > self.fmin, f"fmax must be larger than fmin, but got fmax={self.fmax} and fmin={self.fmin}!"
        assert mel_scale in ["htk", "slaney"], f"mel_scale must be either 'htk' or 'slaney', but got mel_scale={mel_scale}"
        self.mel_scale, self.mel_norm = mel_scale, mel_norm
        self.output_size = n_mels
        self.mel_fbanks = _BufferOnly(mel_filterbank(self.stft_dim, n_mels, sr, self.fmin, self.fmax, mel_scale, mel_norm))
        self.clamp, self.logging, self.log_base, self.mag_spec = clamp, logging, log_base, mag_spec
        self._bands = None

    def bands(self):
        """(lo, hi, image): int32 device tensors of each filter's support and the band-packed weight image; recomputed
        when the buffer is replaced or moved."""
        w = self.mel_fbanks.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._bands is None or self._bands[0] != key:
            lo, hi = mel_bands(w)
            lo, hi = lo.contiguous(), hi.contiguous()
            image = None
            if w.is_cuda:  # band-packed copy the fused frontend kernel keeps in LDS
                image = torch.empty(lib.value("sc_frontend_mel_image_floats", self.n_mels), dtype=torch.float32, device=w.device)
                lib.call("sc_frontend_mel_image_init", ptr(w), ptr(lo), ptr(hi), self.n_mels, w.size(1), ptr(image), stream())
            self._bands = (key, lo, hi, image)
        return self._bands[1], self._bands[2], self._bands[3]

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        require_gpu(feat, "LinearSpec2MelSpec")
        x = feat.float().contiguous()
        out = torch.empty(*x.shape[:-1], self.n_mels, dtype=torch.float32, device=x.device)
        lo, hi, _ = self.bands()
        lib.call("sc_mel_log_fwd", ptr(x), x.numel() // x.size(-1), x.size(-1), ptr(self.mel_fbanks.weight), ptr(lo), ptr(hi),
                 self.n_mels, float(self.clamp), 1 if self.logging else 0,
                 1.0 if self.log_base is None else math.log(self.log_base), ptr(out), stream())
        return out, feat_len

    def recover(self, feat: torch.Tensor, feat_len: torch.Tensor):
        """(log-)mel -> linear spectrogram (linear2mel.py:191-224): undo the logarithm, solve ``mel_fbanks x = mel`` in the least-squares
        sense, silence the padded frames of the shorter utterances, clamp at 1e-10.  The reference calls ``torch.linalg.lstsq(driver=
        'gels')`` on the (n_mels, n_bins) matrix per call: the system is under-determined and of full row rank, so its solution is the
        minimum-norm one, W^T (W W^T)^-1 mel -- one fixed (n_bins, n_mels) matrix (computed once, in float64) applied by the HIP GEMM."""
        require_gpu(feat, "LinearSpec2MelSpec.recover")
        x = feat.float()
        if self.logging:
            x = torch.exp(x * math.log(math.e if self.log_base is None else self.log_base))
        w = self.mel_fbanks.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if getattr(self, "_pinv", None) is None or self._pinv[0] != key:
            w64 = w.double().cpu()
            self._pinv = (key, (w64.t() @ torch.linalg.inv(w64 @ w64.t())).float().to(w.device).contiguous())  # (n_bins, n_mels)
        pinv = self._pinv[1]
        B, T, M = x.shape
        x2 = x.reshape(B * T, M).contiguous()
        out = torch.empty(B * T, self.stft_dim, dtype=torch.float32, device=x.device)
        lib.call("sc_gemm", 0, 0, 0, ptr(x2), ptr(pinv), ptr(out), B * T, self.stft_dim, M, M, M, self.stft_dim, 1, 1, 0, 0, 0, 0, 0, 0, 1.0,
                 None, 0, None, None, 0, 0, 0.0, None, 0, 0, 0, stream())
        out = out.view(B, T, self.stft_dim)
        flen = feat_len.to(x.device)
        idx = torch.arange(T, device=x.device)[None, :, None]
        out = out.masked_fill((idx >= flen[:, None, None]) & (flen != flen.max())[:, None, None], 0.0)
        return torch.clamp(out, min=1e-10)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(\nstft_dim={self.stft_dim}, n_mels={self.n_mels}, fmin={self.fmin}, "
                f"fmax={self.fmax}, mel_scale={self.mel_scale}, mel_norm={self.mel_norm}\n)")
