"""CPU: host logic of the Conv2dPrenet / LinearPrenet drop-ins -- constructor keywords, state_dict keys and shapes against the ones captured
from the imported reference (tests/golden/g9_prenet.npz), output_size, the minimum input sizes, registration under the reference's module
paths and the refusals.  (The arithmetic is GPU-only: tests/test_prenet_gpu.py.)"""
import sys

import pytest
import torch

from prenet_cases import CASES, state


@pytest.mark.parametrize("tag", list(CASES))
def test_state_dict_matches_the_reference_layout(golden, tag):
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet

    g = golden("g9_prenet")
    ref = state(g, tag, requires_grad=False)
    m = Conv2dPrenet(input_size=int(g[f"{tag}.x"].size(-1)), **CASES[tag])
    mine = m.state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype, k
    assert m.output_size == int(g[f"{tag}.output_size"])
    m.load_state_dict(ref, strict=True)


def test_minimum_input_sizes_and_keyword_forms():
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet
    from speechain_amd.module.prenet.linear import LinearPrenet

    m = Conv2dPrenet(input_size=80, conv_dims=[64, 64], conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=512)
    assert (m.min_height, m.min_width) == (7, 7) and m.output_size == 512 and m.out_width == 19  # (80 -> 39 -> 19 mel bins)
    assert Conv2dPrenet(feat_dim=80, conv_dims=32, lnr_dims=None).output_size == 32 * 39        # an int for one layer; no Linear part
    assert Conv2dPrenet(input_size=80, conv_dims=[8], conv_kernel=[3, 5], conv_stride=[2, 1], lnr_dims=None).output_size == 8 * 76
    lp = LinearPrenet(feat_dim=40, lnr_dims=[64, 32], lnr_activation="ReLU", lnr_dropout=[0.1, 0.2], zero_centered=True)
    assert lp.output_size == 32 and [k for k, _ in lp.named_parameters()] == ["linear.0.weight", "linear.0.bias", "linear.3.weight", "linear.3.bias"]
    assert [type(x).__name__ for x in lp.linear] == ["Linear", "ReLU", "Dropout", "Linear", "Dropout"]  # no ReLU after the last layer


def test_refusals():
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet
    from speechain_amd.module.prenet.linear import LinearPrenet

    with pytest.raises(RuntimeError):
        Conv2dPrenet(conv_dims=[4])                                   # neither input_size nor feat_dim (conv2d.py:129-133)
    with pytest.raises(AssertionError):
        Conv2dPrenet(input_size=80, conv_dims="64")
    with pytest.raises(NotImplementedError):
        Conv2dPrenet(input_size=80, conv_dims=[4], conv_activation="Tanh")
    with pytest.raises(NotImplementedError):
        Conv2dPrenet(input_size=80, conv_dims=[4], conv_kernel=5)     # 25 taps: beyond the single-channel first block's 15
    with pytest.raises(NotImplementedError):
        LinearPrenet(feat_dim=8, lnr_dims=[4], lnr_activation="Tanh")
    m = Conv2dPrenet(input_size=20, conv_dims=[4], lnr_dims=None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 37, 20), torch.tensor([37, 30]))


def test_registered_under_the_reference_paths():
    import speechain_amd.register as reg

    reg.register()
    try:
        import importlib

        mod = importlib.import_module("speechain.module.prenet.conv2d")
        assert mod.Conv2dPrenet.__module__ == "speechain_amd.module.prenet.conv2d"
        assert importlib.import_module("speechain.module.prenet.linear").LinearPrenet.__module__ == "speechain_amd.module.prenet.linear"
    finally:
        for k in [k for k in sys.modules if k == "speechain" or k.startswith("speechain.")]:
            del sys.modules[k]
