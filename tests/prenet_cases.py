"""The Conv2dPrenet configurations of tests/golden/g9_prenet.npz (tests/golden/make_golden.py:g9_prenet), shared by the oracle test
(CPU) and the HIP parity test (GPU)."""
import torch

CASES = dict(
    bn=dict(conv_dims=[8, 16], conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=32),
    relu=dict(conv_dims=[8, 8], lnr_dims=[24, 16], lnr_activation="ReLU"),
    asym=dict(conv_dims=[8], conv_kernel=[3, 5], conv_stride=[2, 1], conv_batchnorm=True, lnr_dims=None),
    zc=dict(conv_dims=[8, 8], conv_activation="ReLU", lnr_dims=None, zero_centered=True),
    narrow=dict(conv_dims=[4, 4], conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=8),
)


def state(g, tag, requires_grad=True):
    pre = f"{tag}.sd."
    sd = {k[len(pre):]: v.clone() for k, v in g.items() if k.startswith(pre)}
    if requires_grad:
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_()
    return sd


def weighted_sum(y):
    """The scalar whose gradients the fixture holds (make_golden._grads)."""
    return (y * torch.linspace(-1, 1, y.numel(), device=y.device, dtype=y.dtype).reshape(y.shape)).sum()
