"""``Conv1dPostnet``, drop-in for ``speechain.module.postnet.conv1d.Conv1dPostnet`` (speechain/module/postnet/conv1d.py:16-166): blocks of
Conv1dEv -> [BatchNorm1d] -> [activation] -> [Dropout] in an ``nn.Sequential`` named ``conv`` (same entry order, hence the same
``state_dict`` keys ``conv.{i}.conv_lyr.weight``, ``conv.{i+1}.{weight,bias,running_mean,running_var,num_batches_tracked}``).

HIP path: every block runs through the Conv2d-prenet kernels (csrc/prenet.hip) as a (B, T, 1, C) channels-last image with a (k, 1)
kernel: im2col + GEMM, column statistics, normalise + activation (Tanh added for this module) -- statistics over ALL B*T positions of the
padded batch, as nn.BatchNorm1d on the reference's (B, C, T) tensor takes them."""
from typing import List

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.prenet.conv1d import Conv1dEv


class _As2d:
    """The view of a Conv1dEv that functional.conv_block expects of an nn.Conv2d."""

    def __init__(self, ev: Conv1dEv):
        k = ev.kernel_size
        self.weight = ev.conv_lyr.weight.unsqueeze(-1)  # (out, in, k) -> (out, in, k, 1): a view, gradients reach the parameter
        self.bias = ev.conv_lyr.bias
        self.kernel_size, self.stride, self.padding = (k, 1), (1, 1), ((k - 1) // 2, 0)


class Conv1dPostnet(Module):
    def module_init(self, feat_dim: int = None, conv_dims: int or List[int] = [512, 512, 512, 512, 0], conv_kernel: int = 5, conv_stride: int = 1,
                    conv_padding_mode: str = "same", conv_batchnorm: bool = True, conv_activation: str = "Tanh",
                    conv_dropout: float or List[float] = None, zero_centered: bool = False):
        assert isinstance(conv_dims, (List, int)), "The dimensions of convolutional layers must be given as a list of integers or an integer!"
        assert isinstance(conv_kernel, int), "The sizes of convolutional kernels must be given as an integer!"
        assert isinstance(conv_stride, int), "The lengths of convolutional strides must be given as an integer!"
        if conv_dropout is not None:
            assert isinstance(conv_dropout, (List, float)), \
                "The dropout rates of convolutional layers must be given as a list of integers or an integer!"
        if conv_activation not in Fn.PACT:
            raise NotImplementedError(f"speechain_amd Conv1dPostnet implements Tanh, ReLU, LeakyReLU and no activation, got {conv_activation}")
        if self.input_size is not None:
            feat_dim = self.input_size
        else:
            assert feat_dim is not None
        self.conv_dims = list(conv_dims) if isinstance(conv_dims, List) else [conv_dims]
        self.conv_kernel, self.conv_stride, self.conv_padding_mode, self.conv_dropout = conv_kernel, conv_stride, conv_padding_mode, conv_dropout
        prev, layers, self.plan = feat_dim, [], []
        for i in range(len(self.conv_dims)):
            if self.conv_dims[i] == 0:
                self.conv_dims[i] = feat_dim
            elif self.conv_dims[i] == -1:
                self.conv_dims[i] = self.conv_dims[i - 1]
            blk = dict(conv=len(layers), bn=None, act=None, drop=None)
            layers.append(Conv1dEv(in_channels=prev, out_channels=self.conv_dims[i], kernel_size=conv_kernel, stride=conv_stride,
                                   padding_mode=conv_padding_mode, bias=not conv_batchnorm))
            if conv_batchnorm:
                blk["bn"] = len(layers)
                layers.append(torch.nn.BatchNorm1d(self.conv_dims[i]))
            if conv_activation is not None and (i != len(self.conv_dims) - 1 or not (zero_centered and "ReLU" in conv_activation)):
                blk["act"] = conv_activation
                layers.append(getattr(torch.nn, conv_activation)())
            if conv_dropout is not None:
                blk["drop"] = len(layers)
                layers.append(torch.nn.Dropout(p=conv_dropout if not isinstance(conv_dropout, List) else conv_dropout[i]))
            self.plan.append(blk)
            prev = self.conv_dims[i]
        self.conv = torch.nn.Sequential(*layers)
        self.output_size = prev

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        """feat (batch, feat_maxlen, feat_dim) -> (batch, feat_maxlen, conv_dims[-1]) (postnet/conv1d.py:146-166; feat_len is not used there)."""
        x = Fn.to_compute(feat).unsqueeze(2)  # (B, T, 1, C): channels-last image, one frequency column
        for blk in self.plan:
            ev = self.conv[blk["conv"]]
            bn = self.conv[blk["bn"]] if blk["bn"] is not None else None
            slope = getattr(self.conv[blk["conv"] + (2 if bn is not None else 1)], "negative_slope", 0.0) if blk["act"] == "LeakyReLU" else 0.0
            x = Fn.conv_block(x, _As2d(ev), bn, blk["act"], slope, self.training, first=False)
            if blk["drop"] is not None:
                x = Fn.dropout(x, self.conv[blk["drop"]].p, self.training)
        return x.squeeze(2)
