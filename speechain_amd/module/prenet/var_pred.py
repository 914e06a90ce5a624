"""``Conv1dVarPredictor``, drop-in for ``speechain.module.prenet.var_pred.Conv1dVarPredictor`` (speechain/module/prenet/var_pred.py:42-233):
the duration / pitch / energy predictors of FastSpeech2.  [Conv1dEv -> ReLU -> LayerNorm over channels (eps 1e-12) -> Dropout] blocks in an
``nn.Sequential`` named ``conv`` (so ``conv.{4i}.conv_lyr.*`` and ``conv.{4i+2}.{weight,bias}`` as in the reference), ``linear`` (C -> 1),
optional ``gate_linear``, and ``conv_emb`` that embeds a scalar track back into d_model.

HIP path per block: one GEMM over the k shifted copies of the input with bias + ReLU in its epilogue, one LayerNorm kernel, one dropout
kernel; the heads and the scalar embedding are GEMMs too (N = 1 / K = kernel)."""
from typing import List

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.prenet.conv1d import Conv1dEv, shifted_cat


class LayerNorm(torch.nn.LayerNorm):
    """LayerNorm over dimension `dim` of a (B, C, T) tensor, eps 1e-12 (var_pred.py:9-39)."""

    def __init__(self, nout, dim=-1):
        super().__init__(nout, eps=1e-12)
        self.dim = dim

    def forward(self, x):
        if self.dim == -1:
            return Fn.layer_norm(x, self.weight, self.bias, self.eps)
        return Fn.layer_norm(x.transpose(self.dim, -1), self.weight, self.bias, self.eps).transpose(self.dim, -1)


class Conv1dVarPredictor(Module):
    def module_init(self, feat_dim: int = None, conv_dims: int or List[int] = [256, 256], conv_kernel: int = 3, conv_stride: int = 1,
                    use_gate: bool = False, conv_dropout: float or List[float] = 0.5, use_conv_emb: bool = True, conv_emb_kernel: int = 1,
                    conv_emb_dropout: float = 0.0):
        assert isinstance(conv_dims, (List, int)), "The dimensions of convolutional layers must be given as a list of integers or an integer!"
        assert isinstance(conv_kernel, int), "The sizes of convolutional kernels must be given as an integer!"
        assert isinstance(conv_stride, int), "The lengths of convolutional strides must be given as an integer!"
        if conv_dropout is not None:
            assert isinstance(conv_dropout, (List, float)), \
                "The dropout rates of convolutional layers must be given as a list of integers or an integer!"
        if self.input_size is not None:
            feat_dim = self.input_size
        else:
            assert feat_dim is not None
            self.input_size = feat_dim
        self.conv_dims = list(conv_dims) if isinstance(conv_dims, List) else [conv_dims]
        self.conv_kernel, self.conv_stride, self.conv_dropout = conv_kernel, conv_stride, conv_dropout
        prev, layers = feat_dim, []
        for i in range(len(self.conv_dims)):
            if self.conv_dims[i] == 0:  # 0: back to the input feat_dim; -1: as the previous layer
                self.conv_dims[i] = feat_dim
            elif self.conv_dims[i] == -1:
                self.conv_dims[i] = self.conv_dims[i - 1]
            layers.append(Conv1dEv(in_channels=prev, out_channels=self.conv_dims[i], kernel_size=conv_kernel, stride=conv_stride,
                                   padding_mode="same"))
            layers.append(torch.nn.ReLU())
            layers.append(LayerNorm(self.conv_dims[i], dim=1))
            if conv_dropout is not None:
                layers.append(torch.nn.Dropout(p=conv_dropout if not isinstance(conv_dropout, List) else conv_dropout[i]))
            prev = self.conv_dims[i]
        self.conv = torch.nn.Sequential(*layers)
        self.linear = torch.nn.Linear(prev, 1)
        if use_gate:
            self.gate_linear = torch.nn.Linear(prev, 1)
        if use_conv_emb:
            emb = [Conv1dEv(in_channels=1, out_channels=self.input_size, kernel_size=conv_emb_kernel, padding_mode="same")]
            if conv_emb_dropout > 0:
                emb.append(torch.nn.Dropout(p=conv_emb_dropout))
            self.conv_emb = torch.nn.Sequential(*emb)
        self.output_size = self.input_size

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        """feat (batch, feat_maxlen, feat_dim) -> (batch, feat_maxlen) predictions [+ gate], feat_len unchanged (var_pred.py:185-215)."""
        x = Fn.to_compute(feat)
        mods = list(self.conv)
        i = 0
        while i < len(mods):
            conv, ln = mods[i], mods[i + 2]
            x = Fn.conv_linear(x, conv.conv_lyr.weight, conv.conv_lyr.bias, "ReLU", 0.0)  # (im2col-free when training with gradient sinks)
            x = Fn.layer_norm(x, ln.weight, ln.bias, ln.eps)
            i += 3
            if i < len(mods) and isinstance(mods[i], torch.nn.Dropout):
                x = Fn.dropout(x, mods[i].p, self.training)
                i += 1
        pred = Fn.linear(x, [self.linear.weight], [self.linear.bias]).squeeze(-1)
        if not hasattr(self, "gate_linear"):
            return pred, feat_len
        return pred, feat_len, Fn.linear(x, [self.gate_linear.weight], [self.gate_linear.bias]).squeeze(-1)

    def emb_pred_scalar(self, pred_scalar: torch.Tensor):
        """(batch, feat_maxlen[, 1]) scalar track -> (batch, feat_maxlen, d_model) (var_pred.py:217-233)."""
        assert hasattr(self, "conv_emb"), "Please set the argument 'use_conv_emb' to True if you want to embed the predicted scalar!"
        if pred_scalar.dim() == 2:
            pred_scalar = pred_scalar.unsqueeze(-1)
        else:
            assert pred_scalar.dim() == 3 and pred_scalar.size(-1) == 1
        mods = list(self.conv_emb)
        out = mods[0].forward_btc(pred_scalar)
        for m in mods[1:]:
            out = Fn.dropout(out, m.p, self.training)
        return out
