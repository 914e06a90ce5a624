"""Conv2d prenet (the 4x time-subsampling front of the ASR encoder), drop-in for ``speechain.module.prenet.conv2d.Conv2dPrenet``
(speechain/module/prenet/conv2d.py:15-280): same constructor keywords (:34-48), the same ``nn.Sequential`` of parameter holders --
``conv.{i}.weight`` (C_out, C_in, kh, kw), ``conv.{i}.bias`` only without BatchNorm (:177), BatchNorm2d ``conv.{i}.{weight, bias,
running_mean, running_var, num_batches_tracked}`` -- and ``linear.linear.{i}.*`` of the LinearPrenet behind it, the same
``forward(feat, feat_len) -> (feat, feat_len)`` (:227-280), ``output_size``, ``min_height`` / ``min_width`` padding of short inputs.

HIP path (csrc/prenet.hip, ``functional.conv_block``): every intermediate is channels-last; the first block (one input channel) takes
its BatchNorm statistics from the 9 x 9 patch moments and writes its normalised, activated output once; the following blocks are
im2col + one bf16 GEMM + column statistics + one normalise/activate pass; backward is the adjoint of each.  The reference flattens
(B, C, T', F') channel-major for the Linear part (:251); here the Linear weight's columns are permuted instead of the activations.

Kept differences: ``feat_len`` is not modified in place when a short input is padded (:239); the reference's consistency check
``max(feat_len) != feat.size(1)`` (:266-271, a device-to-host copy every step) is evaluated from the shapes alone: it raises, as the
reference does, whenever the length formula disagrees with the convolution (any conv_padding > 0 with stride 2) -- a batch that is
not padded to its longest utterance is not detected.  The first block supports kernels of at most 15 taps (3 x 3 in every recipe).
"""
from typing import List

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.prenet.linear import LinearPrenet


class Conv2dPrenet(Module):
    def module_init(self, feat_dim: int = None, conv_dims: int or List[int] = [64, 64], conv_kernel: int or List[int] = 3,
                    conv_stride: int or List[int] = 2, conv_padding: int or List[int] = 0, conv_batchnorm: bool = False,
                    conv_activation: str = "ReLU", conv_dropout: float or List[float] = None, lnr_dims: int or List[int] = 512,
                    lnr_activation: str = None, lnr_dropout: float or List[float] = None, zero_centered: bool = False):
        assert isinstance(conv_dims, (List, int)), \
            "The dimensions of convolutional layers must be given as a list of integers or an integer!"
        assert isinstance(conv_kernel, (List, int)), \
            "The sizes of convolutional kernels must be given as a list of integers or an integer!"
        assert isinstance(conv_stride, (List, int)), \
            "The lengths of convolutional strides must be given as a list of integers or an integer!"
        assert isinstance(conv_padding, (List, int)), \
            "The lengths of convolutional paddings must be given as a list of integers, an integer, or a string!"
        if conv_dropout is not None:
            assert isinstance(conv_dropout, (List, float)), \
                "The dropout rates of convolutional layers must be given as a list of integers or an integer!"
        if lnr_dropout is not None:
            assert isinstance(lnr_dropout, (List, float)), \
                "The dropout rates of linear layers must be given as a list of integers or an integer!"
        if lnr_dims is not None:
            assert isinstance(lnr_dims, (List, int)), "The dimensions of linear layers must be given as a list of integers or an integer!"
        if self.input_size is not None:
            feat_dim = self.input_size
        elif feat_dim is None:
            raise RuntimeError
        if conv_activation not in (None, "ReLU", "LeakyReLU"):  # (the first block's fused kernels know these; Tanh exists for the Conv1d postnet only)
            raise NotImplementedError(f"speechain_amd Conv2dPrenet implements ReLU, LeakyReLU and no activation, got {conv_activation}")
        pair = lambda v: tuple(v) if isinstance(v, List) else (v, v)
        self.conv_dims = conv_dims if isinstance(conv_dims, List) else [conv_dims]
        self.conv_kernel, self.conv_stride, self.conv_padding = pair(conv_kernel), pair(conv_stride), pair(conv_padding)
        self.conv_dropout = conv_dropout
        if self.conv_kernel[0] * self.conv_kernel[1] > 15:
            raise NotImplementedError("speechain_amd Conv2dPrenet: the single-channel first block handles kernels of at most 15 taps")

        # smallest input the stack accepts (conv2d.py:141-152; both extents use element [0] of kernel / stride / padding there)
        self.min_height, self.min_width = 1, 1
        for _ in self.conv_dims:
            self.min_height = (self.min_height - 1) * self.conv_stride[0] + self.conv_kernel[0] - 2 * self.conv_padding[0]
            self.min_width = (self.min_width - 1) * self.conv_stride[0] + self.conv_kernel[0] - 2 * self.conv_padding[0]

        layers, self.plan, prev = [], [], 1
        for i, dim in enumerate(self.conv_dims):
            blk = dict(conv=len(layers), bn=None, act=None, p=0.0)
            # no bias in a convolution that is followed by BatchNorm (conv2d.py:177)
            layers.append(torch.nn.Conv2d(in_channels=prev, out_channels=dim, kernel_size=self.conv_kernel, stride=self.conv_stride,
                                          padding=self.conv_padding, bias=not conv_batchnorm))
            if conv_batchnorm:
                blk["bn"] = len(layers)
                layers.append(torch.nn.BatchNorm2d(dim))
            if conv_activation is not None:
                # the last block keeps a ReLU-series activation unless (no Linear part and zero_centered) (conv2d.py:186-190)
                if not (i == len(self.conv_dims) - 1 and lnr_dims is None) or not (zero_centered and "ReLU" in conv_activation):
                    blk["act"] = conv_activation
                    layers.append(getattr(torch.nn, conv_activation)())
            if conv_dropout is not None:
                blk["p"] = conv_dropout if not isinstance(conv_dropout, List) else conv_dropout[i]
                layers.append(torch.nn.Dropout(p=blk["p"]))
            self.plan.append(blk)
            prev = dim
        self.conv = torch.nn.Sequential(*layers)  # parameter / buffer holders; forward() launches the HIP blocks instead

        for _ in self.conv_dims:  # feature width after the stack (conv2d.py:208-209; uses element [-1], without the padding)
            feat_dim = (feat_dim - self.conv_kernel[-1]) // self.conv_stride[-1] + 1
        self.out_width = feat_dim
        self.output_size = prev * feat_dim
        if lnr_dims is not None:
            self.linear = LinearPrenet(feat_dim=self.output_size, lnr_dims=lnr_dims, lnr_activation=lnr_activation,
                                       lnr_dropout=lnr_dropout, zero_centered=zero_centered)
            self.output_size = self.linear.output_size

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        """(batch, feat_maxlen, feat_dim), (batch,) -> (batch, feat_maxlen', output_size) in the compute dtype, (batch,)."""
        if feat.size(1) < self.min_height:
            d = self.min_height - feat.size(1)
            feat = torch.nn.functional.pad(feat, (0, 0, int(d / 2), d - int(d / 2)))
            feat_len = feat_len + d
        if feat.size(2) < self.min_width:
            d = self.min_width - feat.size(2)
            feat = torch.nn.functional.pad(feat, (int(d / 2), d - int(d / 2)))

        x, len_max = feat, feat.size(1)
        # The blocks' length maps len -> floor((len - k) / s) + 1 = floor((len + s - k) / s) compose into ONE floor((len + C) / S)
        # (floor(floor((x + C) / S) + c) / s) = floor((x + C + S c) / (S s)) for integers): two tiny launches per forward instead of three
        # per block, each a ~5 us node on the captured step's only stream.
        len_c, len_s = 0, 1
        for n, blk in enumerate(self.plan):
            bn = self.conv[blk["bn"]] if blk["bn"] is not None else None
            slope = self.conv[blk["bn"] + 1 if bn is not None else blk["conv"] + 1].negative_slope if blk["act"] == "LeakyReLU" else 0.0
            x = Fn.conv_block(x, self.conv[blk["conv"]], bn, blk["act"], slope, self.training, first=(n == 0))
            x = Fn.dropout(x, blk["p"], self.training)
            len_c += len_s * (self.conv_stride[0] - self.conv_kernel[0])
            len_s *= self.conv_stride[0]
            len_max = (len_max - self.conv_kernel[0]) // self.conv_stride[0] + 1
        if self.plan:
            feat_len = torch.div(feat_len + len_c, len_s, rounding_mode="floor")
        B, T, Fo, C = x.shape
        if len_max != T:  # (conv2d.py:266-271)
            raise RuntimeError(f"There is a bug in the {self.__class__.__name__}."
                               f"The calculation of the feature lengths has something wrong.")
        if hasattr(self, "linear"):
            w0 = self.linear.linear[0].weight  # (N, C * Fo) channel-major -> (N, Fo * C) to match the channels-last features
            out = self.linear.run(x.view(B, T, Fo * C), first_weight=w0.view(-1, C, Fo).transpose(1, 2).reshape(-1, Fo * C))
        else:
            out = x.transpose(2, 3).reshape(B, T, C * Fo)
        return out, feat_len
