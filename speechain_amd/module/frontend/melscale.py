"""Mel filterbank matrix for LinearSpec2MelSpec (host-side, float32 torch ops).

The reference takes it from ``torchaudio.functional.melscale_fbanks(...).T``
(speechain/module/frontend/linear2mel.py:135-143); torchaudio is not part of this stack, so its published
construction is implemented here: triangular filters between n_mels+2 points equally spaced on the mel axis,
``fb = max(0, min(down_slope, up_slope))``, optional Slaney area normalisation ``2 / (f[m+2] - f[m])``.
"""
import math

import torch


def _hz_to_mel(freq: float, mel_scale: str) -> float:
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_sp = 200.0 / 3
    if freq >= 1000.0:
        return 1000.0 / f_sp + math.log(freq / 1000.0) / (math.log(6.4) / 27.0)
    return freq / f_sp


def _mel_to_hz(mels: torch.Tensor, mel_scale: str) -> torch.Tensor:
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = 1000.0 * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float, f_max: float, mel_scale: str = "slaney",
                   slaney_norm: bool = True) -> torch.Tensor:
    """(n_mels, n_freqs) float32 -- the layout of the reference buffer ``mel_fbanks.weight``."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, mel_scale), _hz_to_mel(f_max, mel_scale), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    if slaney_norm:
        fb = fb * (2.0 / (f_pts[2: n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.T.contiguous()


def mel_bands(melw: torch.Tensor):
    """Per mel row the half-open [lo, hi) bin range that holds its non-zero weights (int32, host)."""
    nz = melw != 0
    n_bins = melw.size(1)
    idx = torch.arange(n_bins, device=melw.device)
    lo = torch.where(nz, idx, torch.full_like(idx, n_bins)).min(dim=1).values
    hi = torch.where(nz, idx + 1, torch.zeros_like(idx)).max(dim=1).values
    lo = torch.minimum(lo, hi)
    return lo.to(torch.int32), hi.to(torch.int32)
