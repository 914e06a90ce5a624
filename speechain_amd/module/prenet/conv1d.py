"""``Conv1dEv``, drop-in for ``speechain.module.prenet.conv1d.Conv1dEv`` (speechain/module/prenet/conv1d.py:20-128): an ``nn.Conv1d``
held as ``conv_lyr`` (parameters ``conv_lyr.{weight,bias}``, weight (out, in, k)).  HIP path: stride-1, undilated, ungrouped, odd-kernel
convolutions evaluated as ONE GEMM over the k time-shifted copies of the input -- 'same' padding (what every recipe uses: variance
predictors, conv feed-forward, postnet) directly, 'valid' / 'full' / 'causal' (:74-93) as the same product on a zero-padded input and a crop
of its output; strides, dilation, groups, even kernels and weight norm raise NotImplementedError."""
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
        if kernel_size % 2 == 0 or stride != 1 or dilation != 1 or groups != 1 or use_weight_norm:
            raise NotImplementedError("speechain_amd Conv1dEv implements odd-kernel, stride-1, undilated, ungrouped convolutions "
                                      f"(got kernel {kernel_size}, stride {stride}, dilation {dilation}, {padding_mode}, groups {groups})")
        self.cutoff, self.dilation = False, dilation
        self.causal_padding = kernel_size - 1 if padding_mode == "causal" else 0  # (conv1d.py:90-93)
        self.kernel_size, self.padding_mode = kernel_size, padding_mode
        pad = {"valid": 0, "full": kernel_size - 1, "same": (kernel_size - 1) // 2, "causal": 0}[padding_mode]  # (conv1d.py:74-93)
        self.conv_lyr = nn.Conv1d(in_channels, out_channels, kernel_size, padding=pad, bias=bias)

    def gemm_weight(self) -> torch.Tensor:
        w = self.conv_lyr.weight  # (out, in, k) -> (out, k*in) matching shifted_cat's column order
        return w.permute(0, 2, 1).reshape(w.size(0), -1)

    def forward_btc(self, x: torch.Tensor, drop_p: float = 0.0, residual=None) -> torch.Tensor:
        """(B, T, C_in) -> (B, T, C_out): the layout the HIP path works in."""
        x = Fn.to_compute(x)
        mode, r, T = self.padding_mode, (self.kernel_size - 1) // 2, x.size(1)
        if mode == "same":
            return Fn.conv_linear(x, self.conv_lyr.weight, self.conv_lyr.bias, None, drop_p, residual)
        # the other modes as the 'same' product on a zero-padded input, cropped (odd kernel, stride 1): a 'same' convolution of the
        # input with p_l / p_r zero frames in front / behind yields out[u] = sum_j w[j] x_pad[u - r + j]
        assert residual is None, "a residual needs an output as long as the input: 'same' padding"
        if mode == "valid":  # T - k + 1 frames: same[r : T - r]
            if T < self.kernel_size:
                raise RuntimeError(f"Conv1dEv('valid'): {T} frames are fewer than the kernel's {self.kernel_size}")
            return Fn.conv_linear(x, self.conv_lyr.weight, self.conv_lyr.bias, None, drop_p, None)[:, r: T - r]
        if mode == "full":  # T + k - 1 frames
            xp = torch.nn.functional.pad(x, (0, 0, r, r))
            return Fn.conv_linear(xp, self.conv_lyr.weight, self.conv_lyr.bias, None, drop_p, None)
        xp = torch.nn.functional.pad(x, (0, 0, 2 * r, 0))  # causal: k - 1 zero frames in front (conv1d.py:124-125), T frames out
        return Fn.conv_linear(xp, self.conv_lyr.weight, self.conv_lyr.bias, None, drop_p, None)[:, r: r + T]

    def forward(self, feat: torch.Tensor) -> torch.Tensor:
        """The reference's layout: (batch, feat_dim, feat_maxlen) -> (batch, out_channels, feat_maxlen)."""
        return self.forward_btc(feat.transpose(1, 2)).transpose(1, 2)
