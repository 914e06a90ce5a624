"""Waveform -> log-mel spectrogram in ONE fused HIP launch.

Drop-in for ``speechain.module.frontend.speech2mel.Speech2MelSpec`` (speechain/module/frontend/speech2mel.py):
same kwargs (:24-48), sub-modules ``speech2linear`` / ``linear2mel`` (so the state_dict keys
``speech2linear.preemph_filter.weight`` and ``linear2mel.mel_fbanks.weight`` are unchanged), ``forward`` returns
``(feat, feat_len)`` or ``(feat, feat_len, energy, energy_len)`` when ``return_energy`` (:158-189).
"""
import torch

from speechain_amd.module.abs import Module
from speechain_amd.module.frontend.delta_feat import DeltaFeature
from speechain_amd.module.frontend.linear2mel import LinearSpec2MelSpec
from speechain_amd.module.frontend.speech2linear import Speech2LinearSpec, _to_samples, run_frontend


class Speech2MelSpec(Module):
    def module_init(self, n_mels: int, hop_length, win_length, n_fft: int = None, sr: int = 16000,
                    preemphasis: float = None, pre_stft_norm: str = None, window: str = "hann", center: bool = True,
                    normalized: bool = False, onesided: bool = True, mag_spec: bool = False, return_energy: bool = False,
                    fmin: float = 0.0, fmax: float = None, clamp: float = 1e-10, logging: bool = True,
                    log_base: float = 10.0, mel_scale: str = "slaney", mel_norm: bool = True, delta_order: int = None,
                    delta_N: int = 2):
        hop_length, win_length = _to_samples(hop_length, sr), _to_samples(win_length, sr)
        if n_fft is None:
            n_fft = win_length
        self.output_size = n_mels if delta_order is None else n_mels * (delta_order + 1)
        self.return_energy = return_energy
        self.speech2linear = Speech2LinearSpec(
            n_fft=n_fft, sr=sr, hop_length=hop_length, win_length=win_length, preemphasis=preemphasis,
            pre_stft_norm=pre_stft_norm, window=window, center=center, normalized=normalized, onesided=onesided,
            mag_spec=mag_spec, return_energy=return_energy, logging=False)
        self.linear2mel = LinearSpec2MelSpec(
            sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, clamp=clamp, logging=logging, log_base=log_base,
            mel_scale=mel_scale, mel_norm=mel_norm, mag_spec=mag_spec)
        self.delta_order = delta_order
        if delta_order is not None:
            self.delta_N = delta_N
            self.delta = DeltaFeature(delta_order=delta_order, delta_N=delta_N)

    def forward(self, speech: torch.Tensor, speech_len: torch.Tensor):
        s2l, l2m = self.speech2linear, self.linear2mel
        speech, preemph = s2l._prepare(speech, speech_len)
        feat, _, energy, feat_len = run_frontend(
            speech, speech_len, plan=s2l._plan, hop=s2l.hop_length, center=s2l.stft_config["center"],
            preemphasis=preemph, mag_spec=s2l.mag_spec, want_linear=False, want_energy=self.return_energy,
            melw=l2m.mel_fbanks.weight, bands=l2m.bands(), clamp=l2m.clamp, logging=l2m.logging, log_base=l2m.log_base)
        if self.delta_order is not None:
            feat, feat_len = self.delta(feat, feat_len)
        if self.return_energy:
            return feat, feat_len, energy, feat_len
        return feat, feat_len

    def recover(self, feat: torch.Tensor, feat_len: torch.Tensor):
        assert self.delta_order is None
        feat = self.linear2mel.recover(feat, feat_len)
        return self.speech2linear.recover(feat, feat_len)

    def get_sample_rate(self):
        return self.speech2linear.sr

    def __repr__(self):
        s = f"{self.__class__.__name__}(\n" + str(self.speech2linear) + "\n" + str(self.linear2mel)
        if self.delta_order is not None:
            s += "\n" + str(self.delta)
        return s + "\n)"
