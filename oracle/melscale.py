"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``torchaudio.functional.melscale_fbanks``.

The reference obtains its mel filterbank from torchaudio
(``speechain/module/frontend/linear2mel.py:135-143``); torchaudio (pins: ``requirements.txt:17``
0.13.0, ``environment.yaml:154`` 2.0.2) is NOT present in this image nor under /root/reference, so
its published algorithm is restated here.  PARITY UNPINNED at this boundary: no reference test or
fixture holds the matrix values.

Published algorithm (float32 throughout, as torchaudio computes with default-dtype tensors):
  all_freqs = linspace(0, sr // 2, n_freqs)
  m_pts     = linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2);  f_pts = mel2hz(m_pts)
  slopes    = f_pts[None, :] - all_freqs[:, None]
  down      = -slopes[:, :-2] / (f_pts[1:] - f_pts[:-1])[:-1]
  up        =  slopes[:, 2:]  / (f_pts[1:] - f_pts[:-1])[1:]
  fb        = max(0, min(down, up))                                   # (n_freqs, n_mels)
  slaney norm: fb *= 2 / (f_pts[2:] - f_pts[:-2])
"""
import math

import numpy as np

_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = (_MIN_LOG_HZ - 0.0) / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(freq: float, mel_scale: str = "slaney") -> float:
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    mels = (freq - 0.0) / _F_SP
    if freq >= _MIN_LOG_HZ:
        mels = _MIN_LOG_MEL + math.log(freq / _MIN_LOG_HZ) / _LOGSTEP
    return mels


def mel_to_hz(mels: np.ndarray, mel_scale: str = "slaney") -> np.ndarray:
    mels = mels.astype(np.float32)
    if mel_scale == "htk":
        return (np.float32(700.0) * (np.float32(10.0) ** (mels / np.float32(2595.0)) - np.float32(1.0))).astype(np.float32)
    freqs = (np.float32(0.0) + np.float32(_F_SP) * mels).astype(np.float32)
    log_t = mels >= np.float32(_MIN_LOG_MEL)
    freqs[log_t] = (np.float32(_MIN_LOG_HZ) * np.exp(np.float32(_LOGSTEP) * (mels[log_t] - np.float32(_MIN_LOG_MEL)))).astype(np.float32)
    return freqs


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int,
                    norm: str = None, mel_scale: str = "htk") -> np.ndarray:
    """Returns the (n_freqs, n_mels) float32 triangular filterbank."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_min, m_max = hz_to_mel(f_min, mel_scale), hz_to_mel(f_max, mel_scale)
    m_pts = np.linspace(m_min, m_max, n_mels + 2, dtype=np.float32)
    f_pts = mel_to_hz(m_pts, mel_scale)
    f_diff = (f_pts[1:] - f_pts[:-1]).astype(np.float32)
    slopes = (f_pts[None, :] - all_freqs[:, None]).astype(np.float32)
    down = (np.float32(-1.0) * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(np.float32(0.0), np.minimum(down, up)).astype(np.float32)
    if norm == "slaney":
        enorm = (np.float32(2.0) / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).astype(np.float32)
        fb = (fb * enorm[None, :]).astype(np.float32)
    return fb
