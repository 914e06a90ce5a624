"""GPU parity: fused HIP frontend + SpecAugment (through the C ABI) vs golden fixtures and the CPU oracle."""
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu

ASR = dict(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
TTS = dict(sr=22050, hop_length=256, win_length=1024, n_fft=1024, n_mels=80, fmin=0, fmax=8000, mag_spec=True,
           return_energy=True, clamp=1e-5, log_base=None)
ODD = dict(sr=16000, hop_length=128, win_length=400, n_fft=512, n_mels=40, mel_scale="htk", mel_norm=False, log_base=None)
LOGMEL_TOL = 1e-3  # north-star bound on log-mel max-abs error; observed values are printed


def _assert_logmel(out, ref, wave, lens, conf, melw=None):
    """|HIP - reference| <= 1e-3 per utterance, plus the reference's OWN fp32 deviation from a float64 evaluation
    of the same formulas where that is not negligible: on signals with bins ~100 dB below the frame energy (the
    fixture's tone-plus-floor utterance) torch's fp32 FFT is itself 1.5e-3 .. 6e-3 away from the exact value, so two
    correct fp32 evaluations cannot agree to 1e-3 there."""
    kw = dict(conf)
    kw.pop("return_energy", None)
    truth, _ = R.speech2mel(wave.double().squeeze(-1), lens, explicit=True,
                            mel_w=None if melw is None else melw.double(), **kw)
    out = out.cpu()
    worst = 0.0
    for i in range(out.size(0)):
        ref_noise = float((ref[i].double() - truth[i]).abs().max())
        e = float((out[i] - ref[i]).abs().max())
        e64 = float((out[i].double() - truth[i]).abs().max())
        print(f"  utt {i}: |hip-ref| {e:.2e}  |hip-f64| {e64:.2e}  |ref-f64| {ref_noise:.2e}")
        assert e < LOGMEL_TOL + ref_noise, (i, e, ref_noise)
        assert e64 < LOGMEL_TOL + ref_noise, (i, e64, ref_noise)
        worst = max(worst, e)
    return worst


def _mod(conf, melw=None):
    from speechain_amd.module.frontend.speech2mel import Speech2MelSpec

    m = Speech2MelSpec(**conf)
    if melw is not None:
        m.linear2mel.mel_fbanks.weight.copy_(melw)
    return m.cuda()


def _err(a, b):
    return float((a.cpu() - b).abs().max())


@pytest.mark.parametrize("tag,conf,n", [("tts", TTS, 3), ("odd", ODD, 2)])
def test_radix4_transform_opt_in(golden, tag, conf, n, monkeypatch):
    """SC_FRONTEND_RADIX4=1 (csrc/frontend.hip: four quarter-length transforms + twiddled sum, a quarter of the flops; off by default): the
    same outputs as the default path to float32 accuracy on ordinary signals, and an FFT's accuracy -- within 1.5 x the reference's own
    float32 deviation from the float64 evaluation, + 1e-3 -- on the tone-plus-floor utterance whose bins lie 100 dB below the frame energy."""
    g = golden("g1_frontend")
    m = _mod(conf, g[f"{tag}_melw"])
    wave, wl = g["wave"][:n].cuda(), g["wave_len"][:n].cuda()
    base = m(wave, wl)
    monkeypatch.setenv("SC_FRONTEND_RADIX4", "1")
    seen = []
    from speechain_amd._lib import lib
    orig = lib.call
    monkeypatch.setattr(lib, "call", lambda name, *a: (seen.append(name), orig(name, *a))[1])
    out = m(wave, wl)
    assert "sc_frontend_spec_finish4" in seen and torch.equal(out[1], base[1])
    kw = dict(conf)
    kw.pop("return_energy", None)
    truth, _ = R.speech2mel(g["wave"][:n].double().squeeze(-1), g["wave_len"][:n], explicit=True, mel_w=g[f"{tag}_melw"].double(), **kw)
    for i in range(n):
        ref_noise = float((g[f"{tag}_logmel"][i].double() - truth[i]).abs().max())
        e64 = float((out[0][i].cpu().double() - truth[i]).abs().max())
        print(f"  utt {i}: radix-4 |hip-f64| {e64:.2e}  |ref-f64| {ref_noise:.2e}  |radix4-default| {float((out[0][i] - base[0][i]).abs().max()):.2e}")
        assert e64 < LOGMEL_TOL + 1.5 * ref_noise, (i, e64, ref_noise)


@pytest.mark.parametrize("tag,conf,n", [("asr", ASR, 4), ("tts", TTS, 3), ("odd", ODD, 2)])
def test_golden_logmel(golden, tag, conf, n):
    g = golden("g1_frontend")
    m = _mod(conf, g[f"{tag}_melw"])
    wl = g["wave_len"][:n].clone()
    out = m(g["wave"][:n].cuda(), wl.cuda())
    assert torch.equal(wl, g["wave_len"][:n])
    assert torch.equal(out[1].cpu(), g[f"{tag}_len"])
    e = _assert_logmel(out[0], g[f"{tag}_logmel"], g["wave"][:n], g["wave_len"][:n], conf, g[f"{tag}_melw"])
    print(f"{tag} log-mel max-abs err {e:.3e}")
    if tag == "tts":
        ee = _err(out[2], g["tts_energy"])
        print(f"tts energy max-abs err {ee:.3e}")
        assert torch.allclose(out[2].cpu(), g["tts_energy"], atol=1e-4, rtol=1e-4)
        assert torch.equal(out[3].cpu(), g["tts_len"])


def test_golden_linear_power(golden):
    from speechain_amd.module.frontend.speech2linear import Speech2LinearSpec

    g = golden("g1_frontend")
    m = Speech2LinearSpec(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025).cuda()
    lin, flen = m(g["wave"].cuda(), g["wave_len"].cuda())
    assert lin.shape == (4, 151, 201)
    assert torch.allclose(lin[:2, :40].cpu(), g["asr_power"], atol=1e-6, rtol=2e-4)
    full, _ = R.speech2linear(g["wave"], g["wave_len"], sr=16000, preemph=0.97, hop_length=0.010, win_length=0.025)
    scale = float(full.abs().max())
    assert _err(lin, full) < 2e-6 * scale
    # frames beyond feat_len are exactly zero, as the reference leaves them
    for i in range(4):
        assert float(lin[i, int(flen[i]):].abs().max() if flen[i] < 151 else 0.0) == 0.0


def test_default_filterbank_and_standalone_mel(golden):
    from speechain_amd.module.frontend.linear2mel import LinearSpec2MelSpec

    g = golden("g1_frontend")
    m = _mod(ASR)  # own filterbank construction
    feat, _ = m(g["wave"].cuda(), g["wave_len"].cuda())
    _assert_logmel(feat, g["asr_logmel"], g["wave"], g["wave_len"], ASR)
    lin, _ = R.speech2linear(g["wave"], g["wave_len"], sr=16000, preemph=0.97, hop_length=0.010, win_length=0.025)
    l2m = LinearSpec2MelSpec(n_fft=400, n_mels=80).cuda()
    l2m.mel_fbanks.weight.copy_(g["asr_melw"])
    out, _ = l2m(lin.cuda(), None)
    assert _err(out, R.linear2mel(lin, g["asr_melw"])) < 1e-4


@pytest.mark.parametrize("B,T_w,seed", [(1, 400, 0), (3, 4567, 1), (5, 16001, 2), (2, 200, 3)])
def test_ragged_and_tiny_inputs_vs_oracle(B, T_w, seed):
    gen = torch.Generator().manual_seed(seed)
    wave = torch.randn(B, T_w, 1, generator=gen) * 0.1
    lens = torch.randint(1, T_w + 1, (B,), generator=gen)
    lens[0] = T_w
    for i in range(B):
        wave[i, lens[i]:] = 0
    if T_w <= 200:  # reflect padding needs pad < T_w: the reference (torch.stft) errors too
        with pytest.raises(RuntimeError):
            _mod(ASR)(wave.cuda(), lens.cuda())
        return
    ref, rlen = R.speech2mel(wave, lens, **ASR)
    out, olen = _mod(ASR)(wave.cuda(), lens.cuda())
    assert torch.equal(olen.cpu(), rlen)
    assert _err(out, ref) < LOGMEL_TOL


def test_no_center_no_preemph_vs_oracle():
    gen = torch.Generator().manual_seed(7)
    wave = torch.randn(3, 5000, generator=gen)
    lens = torch.tensor([5000, 4000, 777])
    conf = dict(sr=16000, hop_length=100, win_length=320, n_fft=320, n_mels=23, center=False, fmin=20.0, fmax=7600.0)
    ref, rlen = R.speech2mel(wave, lens, **conf)
    out, olen = _mod(conf)(wave.cuda(), lens.cuda())
    assert torch.equal(olen.cpu(), rlen)
    assert _err(out, ref) < LOGMEL_TOL


def test_full_size_batch_vs_oracle():
    """BASELINE config 2: 32 x 10 s @ 16 kHz, N(0, 0.1^2), ragged lengths (seed 1)."""
    gen = torch.Generator().manual_seed(1)
    wave = torch.randn(32, 160000, 1, generator=gen) * 0.1
    lens = torch.randint(80000, 160001, (32,), generator=gen)
    lens[0] = 160000
    ref, rlen = R.speech2mel(wave, lens, **ASR)
    out, olen = _mod(ASR)(wave.cuda(), lens.cuda())
    assert out.shape == (32, 1001, 80)
    assert torch.equal(olen.cpu(), rlen)
    e = _err(out, ref)
    print(f"full-size log-mel max-abs err {e:.3e}")
    assert e < LOGMEL_TOL
    # size-independent property: padded frames hold exactly log10(clamp) = -10
    for i in range(32):
        if rlen[i] < 1001:
            assert float((out[i, int(rlen[i]):] + 10.0).abs().max()) < 1e-5


def _rand_batch(B, T_w, seed):
    gen = torch.Generator().manual_seed(seed)
    wave = torch.randn(B, T_w, generator=gen) * 0.1
    lens = torch.randint(T_w // 2, T_w + 1, (B,), generator=gen)
    lens[0] = T_w
    for i in range(B):
        wave[i, lens[i]:] = 0
    return wave, lens


@pytest.mark.parametrize("conf", [
    dict(ASR, n_mels=10),                                    # filters wider than the 16 LDS slots: dense-matrix loop
    dict(ASR, hop_length=200),                               # n_fft = 400 with a hop the kernel is not specialised for
    dict(ASR, hop_length=150, preemphasis=None),             # hop % 4 != 0: element-wise sample placement
    dict(ASR, n_mels=41, mel_scale="htk", log_base=None),   # odd filter count, natural log
], ids=["wide", "hop200", "hop150", "odd41"])
def test_two_stage_dft_kernel_variants_vs_oracle(conf):
    wave, lens = _rand_batch(3, 23456, 11)
    ref, rlen = R.speech2mel(wave, lens, **conf)
    out, olen = _mod(conf)(wave.cuda(), lens.cuda())
    assert torch.equal(olen.cpu(), rlen)
    assert _err(out, ref) < LOGMEL_TOL


def test_mel_image_and_dense_matrix_paths_agree():
    """The band-packed LDS image (sc_frontend_mel_image_init) is only a layout: without it the kernel reads the dense
    mel matrix and must produce the same log-mel (log via v_log_f32 in both, sums in a different order: 1e-5)."""
    from speechain_amd.module.frontend.speech2linear import run_frontend

    m = _mod(ASR)
    wave, lens = _rand_batch(2, 40000, 5)
    s2l, l2m = m.speech2linear, m.linear2mel
    lo, hi, image = l2m.bands()
    assert image is not None
    kw = dict(plan=s2l._plan, hop=s2l.hop_length, center=True, preemphasis=0.97, mag_spec=False, want_linear=False,
              melw=l2m.mel_fbanks.weight, clamp=l2m.clamp, logging=True, log_base=10.0)
    a = run_frontend(wave.cuda(), lens.cuda(), bands=(lo, hi, image), **kw)[0]
    b = run_frontend(wave.cuda(), lens.cuda(), bands=(lo, hi), **kw)[0]
    assert float((a - b).abs().max()) < 1e-5


def test_linear_energy_magnitude_outputs_n_fft_400():
    """Speech2LinearSpec on the two-stage kernel: magnitude spectrogram + energy, vs the oracle.  Compared in the linear
    domain against the spectrum's scale (a near-cancelled bin has no meaningful log)."""
    from speechain_amd.module.frontend.speech2linear import Speech2LinearSpec

    conf = dict(sr=16000, hop_length=160, win_length=400, mag_spec=True, return_energy=True, logging=False)
    wave, lens = _rand_batch(3, 20000, 9)
    ref = R.speech2linear(wave, lens, preemph=0.97, **conf)
    out = Speech2LinearSpec(preemphasis=0.97, **conf).cuda()(wave.cuda(), lens.cuda())
    assert torch.equal(out[1].cpu(), ref[1])
    assert _err(out[0], ref[0]) < 1e-5 * float(ref[0].abs().max())
    assert float(((out[2].cpu() - ref[2]).abs() / (ref[2].abs() + 1e-3)).max()) < 1e-4


SPECAUG_CASES = dict(
    recipe=dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5),
    default=dict(),
    nowarp_mean=dict(time_warp=False, feat_norm=False, freq_mask_width=[2, 9], time_mask_width=[3, 12]),
    warponly=dict(freq_mask=False, time_mask=False),
    bilinear=dict(time_warp_mode="bilinear", freq_mask_width=15, freq_mask_num=2, time_mask_width=0.05, time_mask_num=2),
)


def _golden_draws(g, name, conf):
    keys = []
    if conf.get("time_warp", True):
        keys += ["warp_center", "warp_pos"]
    if conf.get("freq_mask", True):
        keys += ["freq_len", "freq_pos"]
    if conf.get("time_mask", True):
        keys += ["time_len", "time_pos"]
    d = dict(time_minlen=int(g["feat_len"].min()), time_maxlen=int(g["feat_len"].max()), warp_center=None, warp_pos=None,
             freq_len=None, freq_pos=None, time_len=None, time_pos=None)
    assert len(keys) == int(g[f"{name}_ndraw"])
    for j, k in enumerate(keys):
        v = g[f"{name}_draw{j}"]
        d[k] = int(v.reshape(-1)[0]) if k.startswith("warp") else v
    return d


@pytest.mark.parametrize("name", list(SPECAUG_CASES))
def test_specaug_golden(golden, name):
    from speechain_amd.module.augment.specaug import SpecAugment

    g = golden("g2_specaug")
    conf = SPECAUG_CASES[name]
    mod = SpecAugment(input_size=80, **conf)
    d = _golden_draws(g, name, conf)
    x = g["feat"].cuda()
    out = mod.apply(x, d)
    assert torch.equal(x.cpu(), g["feat"])  # input untouched
    e = _err(out, g[f"{name}_out"])
    print(f"specaug {name} max-abs err {e:.3e}")
    assert e < 1e-4
    # masked positions are bit-exact: the mask indices are integers
    mask = R.specaug_mask(d, 3, 120, 80)
    if mask is not None and conf.get("feat_norm", True):
        assert float(out.cpu()[mask].abs().max()) == 0.0


def test_specaug_forward_full_size_properties():
    from speechain_amd.module.augment.specaug import SpecAugment

    torch.manual_seed(11)
    mod = SpecAugment(input_size=80, **SPECAUG_CASES["recipe"]).cuda()
    feat = torch.randn(32, 1001, 80).cuda()
    flen = torch.randint(600, 1002, (32,))
    torch.manual_seed(5)
    out, olen = mod(feat, flen)
    torch.manual_seed(5)
    d = R.specaug_draw(flen, 32, 80, device="cuda", **SPECAUG_CASES["recipe"])  # same generator stream, same integers
    ref = R.specaug_apply(feat.cpu(), flen, d)
    assert olen is flen
    assert _err(out, ref) < 1e-4
    assert torch.equal(out[:, int(flen.min()):].cpu()[~R.specaug_mask(d, 32, 1001, 80)[:, int(flen.min()):]],
                       feat[:, int(flen.min()):].cpu()[~R.specaug_mask(d, 32, 1001, 80)[:, int(flen.min()):]])
