"""``Conv1dEv``, drop-in for ``speechain.module.prenet.conv1d.Conv1dEv`` (speechain/module/prenet/conv1d.py:20-128): an ``nn.Conv1d``
held as ``conv_lyr`` (parameters ``conv_lyr.{weight,bias}``, weight (out, in, k)).  HIP path: the 'same'-padded, stride-1, odd-kernel
case every recipe uses (variance predictors, conv feed-forward, postnet) evaluated as ONE GEMM over the k time-shifted copies of the
input; the other padding modes, strides, dilation, groups and weight norm raise NotImplementedError."""
import torch
import torch.nn as nn

from speechain_amd import functional as Fn


def shifted_cat(x: torch.Tensor, k: int) -> torch.Tensor:
    """(B, T, d) -> (B, T, k*d): [x[t-r], ..., x[t], ..., x[t+r]] with zero padding (the im2col of a 'same' Conv1d); one HIP launch."""
    return Fn.shifted_cat(x, k)


class Conv1dEv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1, padding_mode: str = "same",
                 bias: bool = True, use_weight_norm: bool = False, groups: int = 1):
        super().__init__()
        if padding_mode not in ("valid", "full", "same", "causal"):
            raise ValueError("Unsupported padding mode. Supported modes are 'valid', 'full', 'same' and 'causal'.")
        if padding_mode != "same" or kernel_size % 2 == 0 or stride != 1 or dilation != 1 or groups != 1 or use_weight_norm:
            raise NotImplementedError("speechain_amd Conv1dEv implements odd-kernel, stride-1, 'same'-padded, ungrouped convolutions "
                                      f"(got kernel {kernel_size}, stride {stride}, dilation {dilation}, {padding_mode}, groups {groups})")
        self.cutoff, self.causal_padding, self.dilation = False, 0, dilation
        self.kernel_size = kernel_size
        self.conv_lyr = nn.Conv1d(in_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2, bias=bias)

    def gemm_weight(self) -> torch.Tensor:
        w = self.conv_lyr.weight  # (out, in, k) -> (out, k*in) matching shifted_cat's column order
        return w.permute(0, 2, 1).reshape(w.size(0), -1)

    def forward_btc(self, x: torch.Tensor, drop_p: float = 0.0, residual=None) -> torch.Tensor:
        """(B, T, C_in) -> (B, T, C_out): the layout the HIP path works in."""
        return Fn.conv_linear(Fn.to_compute(x), self.conv_lyr.weight, self.conv_lyr.bias, None, drop_p, residual)

    def forward(self, feat: torch.Tensor) -> torch.Tensor:
        """The reference's layout: (batch, feat_dim, feat_maxlen) -> (batch, out_channels, feat_maxlen)."""
        return self.forward_btc(feat.transpose(1, 2)).transpose(1, 2)
