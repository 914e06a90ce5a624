"""GPU: recover() of the frontend -- mel -> linear by the minimum-norm least-squares matrix on the HIP GEMM, Griffin-Lim
as products of the f32 GEMM with the forward / inverse bases + csrc/griffinlim.hip, inverse pre-emphasis -- against the reference fixture
(g12_recover) where the reference is deterministic and against the oracle's restatement of torchaudio's published iteration elsewhere."""
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


def test_linear2mel_recover_against_reference(golden):
    from speechain_amd.module.frontend.linear2mel import LinearSpec2MelSpec

    g = golden("g12_recover")
    for base, key in ((10.0, "lin"), (None, "lin_ln")):
        m = LinearSpec2MelSpec(n_fft=64, n_mels=12, sr=16000, log_base=base).cuda()
        got = m.recover(g["mel"].cuda(), g["mel_len"].cuda()).cpu()
        want = g[key]
        assert torch.allclose(got, want, atol=2e-4, rtol=2e-3), float((got - want).abs().max())
        assert float(got[1, 6:].max()) == pytest.approx(1e-10)  # the shorter utterance's padded frames are silenced, then clamped


def test_recover_postprocessing_against_reference(golden):
    from speechain_amd.module.frontend.speech2linear import Speech2LinearSpec

    g = golden("g12_recover")
    m = Speech2LinearSpec(hop_length=16, win_length=64, sr=16000, preemphasis=0.97).cuda()
    m._griffin_lim = lambda feat, *a, **k: g["gl_wav"].cuda().clone()
    wav, wav_len = m.recover(torch.rand(2, 11, 33).cuda(), g["feat_len"].cuda(), inv_preemph_winlen=20)
    assert torch.equal(wav_len.cpu(), g["wav_len"])
    assert torch.allclose(wav.cpu(), g["wav"], atol=2e-6), float((wav.cpu() - g["wav"]).abs().max())


@pytest.mark.parametrize("n_fft,hop,win,mag_spec,preemph", [(64, 16, 64, True, None), (128, 32, 100, False, 0.97), (400, 160, 400, False, 0.97)])
def test_griffin_lim_against_the_restated_iteration(n_fft, hop, win, mag_spec, preemph):
    """Same initial phases, 6 iterations with momentum: the HIP path (framing, two f32 GEMMs per iteration, overlap-add, phase step) equals
    the oracle (torch.stft / torch.istft on the CPU) sample by sample."""
    from speechain_amd.module.frontend.speech2linear import Speech2LinearSpec

    gen = torch.Generator().manual_seed(n_fft)
    B, T = 2, 24
    wave = torch.randn(B, hop * (T - 1), generator=gen).cumsum(-1) * 0.02
    window = torch.hann_window(win)
    mag = torch.stft(wave, n_fft, hop, win, window, return_complex=True).abs().transpose(1, 2)
    assert mag.size(1) == T
    feat = mag if mag_spec else mag ** 2
    init = torch.rand(B, T, n_fft // 2 + 1, generator=gen) + 1j * torch.rand(B, T, n_fft // 2 + 1, generator=gen)
    feat_len = torch.tensor([T, T - 5])
    want_gl = R.griffin_lim(mag, window, n_fft, hop, win, 6, 0.99, init)
    want, want_len = R.speech2linear_recover_post(want_gl, feat_len, hop, preemph)
    m = Speech2LinearSpec(hop_length=hop, win_length=win, n_fft=n_fft, sr=16000, preemphasis=preemph, mag_spec=mag_spec).cuda()
    got, got_len = m.recover(feat.cuda(), feat_len.cuda(), n_iter=6, _init_angles=init.cuda())
    assert torch.equal(got_len.cpu(), want_len)
    scale = float(want.abs().max())
    assert float((got.cpu() - want).abs().max()) < 2e-4 * max(scale, 1.0) + 1e-5, (float((got.cpu() - want).abs().max()), scale)


def test_speech2mel_recover_at_the_tts_preset_converges():
    """Speech2MelSpec.recover at the FastSpeech2 preset (22.05 kHz, n_fft 1024, hop 256, 80 mels, magnitude, natural log): the mel
    spectrogram of the recovered waveform is close to the one it was recovered from (32 iterations, random initial phases)."""
    from speechain_amd.module.frontend.speech2mel import Speech2MelSpec

    torch.manual_seed(0)
    m = Speech2MelSpec(n_mels=80, hop_length=256, win_length=1024, n_fft=1024, sr=22050, fmin=0, fmax=8000, mag_spec=True, clamp=1e-5,
                       log_base=None).cuda()
    t = torch.arange(256 * 60) / 22050.0
    wave = torch.stack([0.3 * torch.sin(2 * torch.pi * 440 * t) + 0.1 * torch.sin(2 * torch.pi * 1320 * t + 1.0),
                        0.2 * torch.sin(2 * torch.pi * (200 + 300 * t) * t)]).cuda()
    wave = wave + 0.01 * torch.randn_like(wave)
    wlen = torch.tensor([wave.size(1), wave.size(1) - 256 * 9]).cuda()
    feat, feat_len = m(wave.unsqueeze(-1), wlen.clone())
    wav, wav_len = m.recover(feat, feat_len)
    assert wav.shape == (2, 256 * (feat.size(1) - 1), 1) and torch.equal(wav_len, (feat_len - 1) * 256)
    assert float(wav[1, int(wav_len[1]):].abs().max()) == 0.0
    feat2, _ = m(wav, wav_len.clone())
    n = int(feat_len[1]) - 2
    err = float((feat2[:, 2:n] - feat[:, 2:n]).abs().mean())
    assert err < 0.35, err  # mean |delta ln mel| over the frames both hold (mel inversion + phase retrieval are both lossy)
