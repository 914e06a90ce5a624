"""``Conv1dEv`` padding modes (speechain/module/prenet/conv1d.py:74-93,124-128): 'valid', 'full', 'same' and 'causal' of the HIP path against
``torch.nn.functional.conv1d`` on the same weights with the reference's padding arithmetic -- output, input gradient, weight gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["valid", "full", "same", "causal"])
@pytest.mark.parametrize("k", [1, 3, 9])
def test_conv1dev_padding_modes_against_torch(mode, k):
    from speechain_amd.module.prenet.conv1d import Conv1dEv

    torch.manual_seed(k)
    B, C, N, T = 3, 24, 40, 37
    m = Conv1dEv(C, N, k, padding_mode=mode)
    x = torch.randn(B, C, T)
    w = torch.randn(B, N, {"valid": T - k + 1, "full": T + k - 1, "same": T, "causal": T}[mode])
    # the reference's forward on the same holder (conv1d.py:124-128; odd kernels: no cut-off)
    xr = x.clone().requires_grad_()
    ref = m.conv_lyr(torch.nn.functional.pad(xr, (m.causal_padding, 0)) if m.causal_padding > 0 else xr)
    (ref * w).sum().backward()
    gw_ref, gx_ref = m.conv_lyr.weight.grad.clone(), xr.grad.clone()
    m.conv_lyr.weight.grad = None
    m.conv_lyr.bias.grad = None
    g = m.cuda()
    xg = x.cuda().requires_grad_()
    out = g(xg)
    assert out.shape == ref.shape, (out.shape, ref.shape)
    (out * w.cuda()).sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=2e-4, rtol=1e-4), float((out.detach().cpu() - ref.detach()).abs().max())
    assert torch.allclose(xg.grad.cpu(), gx_ref, atol=2e-4, rtol=1e-4)
    assert torch.allclose(g.conv_lyr.weight.grad.cpu(), gw_ref, atol=2e-3, rtol=1e-3)


def test_conv1dev_refusals():
    from speechain_amd.module.prenet.conv1d import Conv1dEv

    for kw in (dict(kernel_size=4), dict(kernel_size=3, stride=2, padding_mode="valid"), dict(kernel_size=3, dilation=2), dict(kernel_size=3, groups=2)):
        with pytest.raises(NotImplementedError):
            Conv1dEv(8, 8, **kw)
    with pytest.raises(ValueError):
        Conv1dEv(8, 8, 3, padding_mode="reflect")
