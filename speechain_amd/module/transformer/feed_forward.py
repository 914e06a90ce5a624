"""Position-wise feed-forward, drop-in for ``speechain.module.transformer.feed_forward.PositionwiseFeedForward``
(speechain/module/transformer/feed_forward.py): ``in_layer`` / ``out_layer`` are ``nn.Linear`` (or the k-tap 'same'
Conv1d of the TTS recipes, parameters ``{in,out}_layer.conv_lyr.{weight,bias}``), activation by ``torch.nn`` name, dropout
after the activation (:77-106).  HIP path: two GEMMs with bias + activation + dropout (+ the caller's residual) fused in
their epilogues; the backward's activation derivative is fused into the dgrad GEMM."""
from typing import Any, Dict

import torch
import torch.nn as nn

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


class Conv1dSame(nn.Module):
    """Odd-kernel stride-1 zero-padded Conv1d holder with the reference's parameter names (prenet/conv1d.py:73-90,
    ``conv_lyr.weight`` (out, in, k)).  Evaluated as a GEMM over the k time-shifted copies of the input."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, **kw):
        super().__init__()
        if kernel_size % 2 == 0 or kw.get("stride", 1) != 1 or kw.get("dilation", 1) != 1 or kw.get("padding_mode", "same") != "same":
            raise NotImplementedError("speechain_amd conv feed-forward implements odd-kernel, stride-1, 'same' convolutions")
        self.conv_lyr = nn.Conv1d(in_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2, bias=kw.get("bias", True))
        self.kernel_size = kernel_size

    def gemm_weight(self) -> torch.Tensor:
        w = self.conv_lyr.weight  # (out, in, k) -> (out, k*in) matching shifted_cat's column order
        return w.permute(0, 2, 1).reshape(w.size(0), -1)


def shifted_cat(x: torch.Tensor, k: int) -> torch.Tensor:
    """(B, T, d) -> (B, T, k*d): [x[t-r], ..., x[t], ..., x[t+r]] with zero padding (the im2col of a 'same' Conv1d)."""
    r = (k - 1) // 2
    xp = torch.nn.functional.pad(x, (0, 0, r, r))
    return torch.cat([xp[:, j: j + x.size(1)] for j in range(k)], dim=-1)


class PositionwiseFeedForward(Module):
    def module_init(self, d_model: int = 512, fdfwd_dim: int = 2048, fdfwd_type: str = "linear", fdfwd_activation: str = "ReLU",
                    fdfwd_args: Dict[str, Any] = {}, dropout=0.1):
        if len(fdfwd_args) == 0 and fdfwd_type == "conv":
            fdfwd_args = dict(kernel_size=3)
        if fdfwd_type == "linear":
            self.in_layer = nn.Linear(d_model, fdfwd_dim, **fdfwd_args)
            self.out_layer = nn.Linear(fdfwd_dim, d_model, **fdfwd_args)
        elif fdfwd_type == "conv":
            self.in_layer = Conv1dSame(d_model, fdfwd_dim, **fdfwd_args)
            self.out_layer = Conv1dSame(fdfwd_dim, d_model, **fdfwd_args)
        else:
            raise NotImplementedError(f"Currently, fdfwd_type can only be one of 'linear' and 'conv'. But got {fdfwd_type}!")
        self.activation_name = fdfwd_activation
        self.activation = getattr(torch.nn, fdfwd_activation)()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, *, _residual=None, _res_drop: float = 0.0):
        x = Fn.to_compute(x)
        p = self.dropout.p if self.training else 0.0
        res = Fn.to_compute(_residual, x.dtype) if _residual is not None else None
        if isinstance(self.in_layer, nn.Linear):
            return Fn.ffn(x, self.in_layer.weight, self.in_layer.bias, self.out_layer.weight, self.out_layer.bias,
                          self.activation_name, p, _res_drop, res)
        # conv type: each layer is a GEMM over the time-shifted concatenation of its input
        cin, cout = self.in_layer, self.out_layer
        h = Fn.linear_act(shifted_cat(x, cin.kernel_size), cin.gemm_weight(), cin.conv_lyr.bias, self.activation_name, p)
        return Fn.linear(shifted_cat(h, cout.kernel_size), [cout.gemm_weight()], [cout.conv_lyr.bias], drop_p=_res_drop, residual=res)
