"""Transformer encoder, drop-in for ``speechain.module.transformer.encoder`` (speechain/module/transformer/encoder.py):
``TransformerEncoderLayer`` (:19-143; sub-modules ``multihead_att``, ``feed_forward``, ``att_layernorm``,
``fdfwd_layernorm`` with eps 1e-6) and ``TransformerEncoder`` (:146-349; ``posenc``, ``trfm_layers``, ``layernorm``;
``forward(src, mask) -> (out, mask, attmat[L], hidden[L])``).

HIP path per pre-LN layer: LayerNorm kernel -> fused QKV GEMM -> QK^T / masked softmax / PV -> output projection with
residual + residual-dropout epilogue -> LayerNorm kernel -> FFN (two GEMMs, fused bias/activation/dropout/residual).
Activations run in the compute dtype (bf16 under autocast / ``set_compute_dtype``, else f32) and are returned in it.
The per-layer ``hidden`` entries are the layer outputs themselves (the reference clones them, :343; nothing here
mutates them, so the copy is skipped).
"""
from typing import Any, Dict

import torch
from torch import nn

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.transformer.attention import MultiHeadedAttention
from speechain_amd.module.transformer.feed_forward import PositionwiseFeedForward
from speechain_amd.module.transformer.pos_enc import PositionalEncoding


def _ln(mod: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    return Fn.layer_norm(x, mod.weight, mod.bias, mod.eps)


def _ln_res(mod: nn.LayerNorm, x: torch.Tensor):
    """(LayerNorm(x), x for the residual connection): both gradients of x are summed inside the LayerNorm backward kernel."""
    return Fn.layer_norm_with_residual(x, mod.weight, mod.bias, mod.eps)


class TransformerEncoderLayer(Module):
    def module_init(self, d_model: int = 512, num_heads: int = 8, scale_dp_by_head: bool = False, att_dropout: float = 0.1,
                    fdfwd_dim: int = 2048, fdfwd_type: str = "linear", fdfwd_activation: str = "ReLU",
                    fdfwd_args: Dict[str, Any] = {}, fdfwd_dropout: float = 0.1, res_dropout: float = 0.1,
                    layernorm_first: bool = True):
        self.multihead_att = MultiHeadedAttention(d_model=d_model, num_heads=num_heads, dropout=att_dropout,
                                                  scale_dp_by_head=scale_dp_by_head)
        self.feed_forward = PositionwiseFeedForward(d_model=d_model, fdfwd_dim=fdfwd_dim, fdfwd_type=fdfwd_type,
                                                    fdfwd_activation=fdfwd_activation, fdfwd_args=fdfwd_args, dropout=fdfwd_dropout)
        self.dropout = nn.Dropout(res_dropout)
        self.layernorm_first = layernorm_first
        self.att_layernorm = nn.LayerNorm(d_model, eps=1e-6)
        self.fdfwd_layernorm = nn.LayerNorm(d_model, eps=1e-6)

    def forward(self, src: torch.Tensor, mask: torch.Tensor, *, _causal: bool = False):
        src = Fn.to_compute(src)
        p = self.dropout.p if self.training else 0.0
        # pre-LN: the sublayer input feeds the norm AND the residual connection; _ln_res hands it back so that both of its
        # gradients meet inside the LayerNorm backward kernel
        src_norm, src = _ln_res(self.att_layernorm, src) if self.layernorm_first else (src, src)
        att_output, attmat = self.multihead_att(src_norm, src_norm, src_norm, mask, _residual=src, _res_drop=p, _causal=_causal)
        if not self.layernorm_first:
            att_output = _ln(self.att_layernorm, att_output)
        att_norm, att_output = _ln_res(self.fdfwd_layernorm, att_output) if self.layernorm_first else (att_output, att_output)
        out = self.feed_forward(att_norm, _residual=att_output, _res_drop=p)
        if not self.layernorm_first:
            out = _ln(self.fdfwd_layernorm, out)
        return out, attmat


class TransformerEncoder(Module):
    def module_init(self, posenc_type: str = "mix", posenc_maxlen: int = 5000, posenc_dropout: float = 0.1,
                    posenc_scale: bool = False, posenc_init_alpha: float = 1.0, emb_layernorm: bool = False,
                    emb_scale: bool = False, d_model: int = 512, num_heads: int = 4, num_layers: int = 8,
                    scale_dp_by_head: bool = False, att_dropout: float = 0.1, fdfwd_dim: int = 2048, fdfwd_type: str = "linear",
                    fdfwd_activation: str = "ReLU", fdfwd_args: Dict[str, Any] = {}, fdfwd_dropout: float = 0.1,
                    res_dropout: float = 0.1, layernorm_first: bool = True, uni_direction: bool = False):
        if self.input_size is not None:
            d_model = self.input_size
        self.output_size = d_model
        self.d_model, self.num_layers, self.num_heads = d_model, num_layers, num_heads
        self.layernorm_first, self.uni_direction = layernorm_first, uni_direction
        self.posenc = PositionalEncoding(posenc_type=posenc_type, d_model=d_model, emb_scale=emb_scale, emb_layernorm=emb_layernorm,
                                         posenc_scale=posenc_scale, init_alpha=posenc_init_alpha, max_len=posenc_maxlen,
                                         dropout=posenc_dropout)
        self.trfm_layers = torch.nn.ModuleList([
            TransformerEncoderLayer(d_model=d_model, num_heads=num_heads, scale_dp_by_head=scale_dp_by_head, att_dropout=att_dropout,
                                    fdfwd_dim=fdfwd_dim, fdfwd_type=fdfwd_type, fdfwd_activation=fdfwd_activation,
                                    fdfwd_args=fdfwd_args, fdfwd_dropout=fdfwd_dropout, res_dropout=res_dropout,
                                    layernorm_first=layernorm_first) for _ in range(num_layers)])
        if self.layernorm_first:
            self.layernorm = nn.LayerNorm(d_model, eps=1e-6)

    @staticmethod
    def subsequent_mask(batch_size, maxlen: int) -> torch.Tensor:
        return ~torch.triu(torch.ones(batch_size, maxlen, maxlen, dtype=torch.bool), diagonal=1)

    def forward(self, src: torch.Tensor, mask: torch.Tensor):
        src = self.posenc(src)
        pad_mask = mask
        if self.uni_direction:
            # the returned mask is the reference's dense (B, T, T) one; the kernels take padding mask + causal flag
            batch_size, _, src_maxlen = mask.size()
            mask = torch.logical_and(mask.repeat(1, src_maxlen, 1), self.subsequent_mask(batch_size, src_maxlen).to(mask.device))
        attmat, hidden = [], []
        for layer in self.trfm_layers:
            src, _att = layer(src, pad_mask, _causal=self.uni_direction)
            # A metadata-only autograd node (no kernel, nothing saved) between the layers: a trainer that runs backward in pieces
            # (torch.autograd.backward(..., inputs=[hidden[k]])) needs its cut tensors to come from such a node -- autograd releases the
            # saved tensors of a custom Function whose output is a capture point, and the next piece then cannot run it.
            src = src.view_as(src)
            attmat.append(_att)
            hidden.append(src)
        if self.layernorm_first:
            src = _ln(self.layernorm, src).view_as(src)  # (same: the encoder output is the natural first cut)
        return src, mask, attmat, hidden
