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
from speechain_amd.module.prenet.conv1d import Conv1dEv, shifted_cat


def Conv1dSame(in_channels: int, out_channels: int, kernel_size: int = 3, **kw) -> Conv1dEv:
    """The k-tap 'same' Conv1d of the conv feed-forward (feed_forward.py:62-75 builds Conv1dEv(..., **fdfwd_args))."""
    return Conv1dEv(in_channels, out_channels, kernel_size, **kw)


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
        # (the Conv1d weights go in as they are, (out, in, k): functional reads them as the (out, k * in) matrices of these products and
        # writes their gradients back in the weights' own layout -- cached casts, gradient-bucket slots, the grouped wgrad launch)
        # (Fn.conv_ffn: with gradient sinks registered -- the training step -- products over overlapping rows of zero-padded images, no
        #  im2col / col2im / activation-backward launches; otherwise shifted_cat + linear.  Same forward arithmetic, bit for bit.)
        return Fn.conv_ffn(x, cin.conv_lyr.weight, cin.conv_lyr.bias, cout.conv_lyr.weight, cout.conv_lyr.bias, self.activation_name, p,
                           _res_drop, res)
