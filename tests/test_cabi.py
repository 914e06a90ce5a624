"""CPU: the C-ABI library loads and exports every symbol include/speechain_amd.h declares; host-side contracts."""
import ctypes
import os

import pytest
import torch

from speechain_amd import _lib


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m speechain_amd.build` first"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    protos = _lib.parse_header()
    assert len(protos) >= 8
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in the header but not exported"


def test_header_is_plain_c_abi():
    import re

    text = open(_lib.HEADER_PATH).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code


def test_loader_binds_and_reports_errors():
    _lib.lib.load()
    assert _lib.lib.value("sc_version") >= 1
    assert _lib.lib.value("sc_frontend_num_tiles", 400) == 13
    assert _lib.lib.value("sc_frontend_basis_floats", 400) >= 400 * 13 * 32
    assert _lib.lib.value("sc_frontend_basis_floats", 1024) == 1024 * 33 * 32
    with pytest.raises(RuntimeError, match="null pointer"):
        _lib.lib.call("sc_frontend_basis_init", None, 400, 400, 160, None, None)


def test_modules_refuse_cpu_tensors_and_keep_state_dict_keys():
    from speechain_amd.module.augment.specaug import SpecAugment
    from speechain_amd.module.frontend.speech2mel import Speech2MelSpec

    m = Speech2MelSpec(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
    assert m.output_size == 80 and m.get_sample_rate() == 16000
    assert list(m.state_dict()) == ["speech2linear.preemph_filter.weight", "linear2mel.mel_fbanks.weight"]
    assert m.state_dict()["speech2linear.preemph_filter.weight"].shape == (1, 1, 2)
    assert m.state_dict()["linear2mel.mel_fbanks.weight"].shape == (80, 201)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4000, 1), torch.tensor([4000]))
    with pytest.raises(RuntimeError, match="multi-channel"):
        m(torch.zeros(1, 4000, 2), torch.tensor([4000]))
    s = SpecAugment(input_size=80)
    assert s.output_size == 80
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s(torch.zeros(2, 50, 80), torch.tensor([50, 40]))


def test_mel_filterbank_matches_fixture(golden):
    from speechain_amd.module.frontend.melscale import mel_bands, mel_filterbank

    g = golden("g1_frontend")
    w = mel_filterbank(201, 80, 16000, 0.0, 8000, "slaney", True)
    assert torch.allclose(w, g["asr_melw"], atol=1e-6, rtol=1e-5)
    w2 = mel_filterbank(513, 80, 22050, 0, 8000, "slaney", True)
    assert torch.allclose(w2, g["tts_melw"], atol=1e-6, rtol=1e-5)
    w3 = mel_filterbank(257, 40, 16000, 0.0, 8000, "htk", False)
    assert torch.allclose(w3, g["odd_melw"], atol=1e-5, rtol=1e-5)
    lo, hi = mel_bands(w)
    for m in range(80):
        nz = torch.nonzero(w[m]).flatten()
        assert lo[m] == nz.min() and hi[m] == nz.max() + 1


SPECAUG_CASES = dict(
    recipe=dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5),
    default=dict(),
    nowarp_mean=dict(time_warp=False, feat_norm=False, freq_mask_width=[2, 9], time_mask_width=[3, 12]),
    warponly=dict(freq_mask=False, time_mask=False),
)


@pytest.mark.parametrize("name", list(SPECAUG_CASES))
def test_specaug_draws_are_bit_exact(golden, name):
    """Host-side draw logic of the product module == the integers the reference drew under the same seed."""
    from speechain_amd.module.augment.specaug import SpecAugment

    g = golden("g2_specaug")
    mod = SpecAugment(input_size=80, **SPECAUG_CASES[name])
    torch.manual_seed(int(g[f"{name}_seed"]))
    d = mod.draw(g["feat_len"], 3, 80, torch.device("cpu"))
    drawn = [d[k] for k in ("warp_center", "warp_pos", "freq_len", "freq_pos", "time_len", "time_pos") if d[k] is not None]
    assert len(drawn) == int(g[f"{name}_ndraw"])
    for j, x in enumerate(drawn):
        ref = g[f"{name}_draw{j}"]
        assert torch.equal(torch.as_tensor(x).reshape(ref.shape), ref)
