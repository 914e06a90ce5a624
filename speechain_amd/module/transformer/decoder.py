"""Transformer decoder, drop-in for ``speechain.module.transformer.decoder`` (speechain/module/transformer/decoder.py):
``TransformerDecoderLayer`` (:16-175; ``self_att``, ``encdec_att``, ``feed_forward``, ``self_att_ln``,
``encdec_att_ln``, ``fdfwd_ln``) and ``TransformerDecoder`` (:178-362; ``forward(tgt, src, tgt_mask, src_mask,
return_att, return_hidden) -> (out, self_attmat[L], encdec_attmat[L], hidden[L])``).

Quirks kept: the enc-dec attention never scales by head (:76-78), the decoder never forwards ``layernorm_first`` to its
layers (:271-285), ``emb_scale`` defaults to True (:187).  The causal mask is not materialised: the softmax kernel
combines the (B,1,L) padding mask with ``k <= q``.
"""
import torch
from torch import nn

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.transformer.attention import MultiHeadedAttention
from speechain_amd.module.transformer.encoder import _ln, _ln_res
from speechain_amd.module.transformer.feed_forward import PositionwiseFeedForward
from speechain_amd.module.transformer.pos_enc import PositionalEncoding


class TransformerDecoderLayer(Module):
    def module_init(self, d_model: int = 512, num_heads: int = 8, scale_dp_by_head: bool = False, att_dropout: float = 0.1,
                    fdfwd_dim: int = 0, fdfwd_activation: str = "ReLU", fdfwd_dropout: float = 0.1, res_dropout: float = 0.1,
                    layernorm_first: bool = True):
        self.self_att = MultiHeadedAttention(num_heads=num_heads, d_model=d_model, dropout=att_dropout, scale_dp_by_head=scale_dp_by_head)
        self.encdec_att = MultiHeadedAttention(num_heads=num_heads, d_model=d_model, dropout=att_dropout)
        self.feed_forward = PositionwiseFeedForward(d_model=d_model, fdfwd_dim=fdfwd_dim, fdfwd_activation=fdfwd_activation,
                                                    dropout=fdfwd_dropout)
        self.layernorm_first = layernorm_first
        self.self_att_ln = nn.LayerNorm(d_model, eps=1e-6)
        self.encdec_att_ln = nn.LayerNorm(d_model, eps=1e-6)
        self.fdfwd_ln = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout = nn.Dropout(res_dropout)

    def forward(self, tgt: torch.Tensor, src: torch.Tensor, tgt_mask: torch.Tensor, src_mask: torch.Tensor, *, _causal: bool = False):
        """``tgt_mask`` is the reference's dense (B, L, L) mask, or -- with ``_causal=True`` -- the (B, 1, L) padding mask."""
        tgt, src = Fn.to_compute(tgt), Fn.to_compute(src)
        p = self.dropout.p if self.training else 0.0
        first = self.layernorm_first
        # (pre-LN: _ln_res returns the sublayer input again for the residual connection, see encoder.py)
        tgt_norm, tgt = _ln_res(self.self_att_ln, tgt) if first else (tgt, tgt)
        x, self_attmat = self.self_att(tgt_norm, tgt_norm, tgt_norm, mask=tgt_mask, _residual=tgt, _res_drop=p, _causal=_causal)
        if not first:
            x = _ln(self.self_att_ln, x)
        x_norm, x = _ln_res(self.encdec_att_ln, x) if first else (x, x)
        y, encdec_attmat = self.encdec_att(src, src, x_norm, mask=src_mask, _residual=x, _res_drop=p)
        if not first:
            y = _ln(self.encdec_att_ln, y)
        y_norm, y = _ln_res(self.fdfwd_ln, y) if first else (y, y)
        out = self.feed_forward(y_norm, _residual=y, _res_drop=p)
        if not first:
            out = _ln(self.fdfwd_ln, out)
        return out, self_attmat, encdec_attmat


class TransformerDecoder(Module):
    def module_init(self, posenc_type: str = "mix", posenc_maxlen: int = 5000, posenc_dropout: float = 0.1,
                    posenc_scale: bool = False, posenc_init_alpha: float = 1.0, emb_layernorm: bool = False,
                    emb_scale: bool = True, d_model: int = 512, num_heads: int = 4, num_layers: int = 8,
                    scale_dp_by_head: bool = False, fdfwd_dim: int = 2048, fdfwd_activation: str = "ReLU",
                    fdfwd_dropout: float = 0.1, att_dropout: float = 0.1, res_dropout: float = 0.1, layernorm_first: bool = True):
        if self.input_size is not None:
            d_model = self.input_size
        self.output_size = d_model
        self.d_model, self.num_layers, self.num_heads, self.layernorm_first = d_model, num_layers, num_heads, layernorm_first
        self.posenc = PositionalEncoding(posenc_type=posenc_type, d_model=d_model, emb_scale=emb_scale, emb_layernorm=emb_layernorm,
                                         posenc_scale=posenc_scale, init_alpha=posenc_init_alpha, max_len=posenc_maxlen,
                                         dropout=posenc_dropout)
        self.trfm_layers = torch.nn.ModuleList([
            TransformerDecoderLayer(d_model=d_model, num_heads=num_heads, scale_dp_by_head=scale_dp_by_head, att_dropout=att_dropout,
                                    fdfwd_dim=fdfwd_dim, fdfwd_activation=fdfwd_activation, fdfwd_dropout=fdfwd_dropout,
                                    res_dropout=res_dropout) for _ in range(num_layers)])
        if self.layernorm_first:
            self.layernorm = nn.LayerNorm(d_model, eps=1e-6)

    @staticmethod
    def subsequent_mask(batch_size, maxlen: int) -> torch.Tensor:
        return ~torch.triu(torch.ones(batch_size, maxlen, maxlen, dtype=torch.bool), diagonal=1)

    def forward(self, tgt: torch.Tensor, src: torch.Tensor, tgt_mask: torch.Tensor, src_mask: torch.Tensor, return_att: bool = False,
                return_hidden: bool = False):
        assert tgt_mask is not None, "tgt_mask is required for Transformer!"
        tgt = self.posenc(tgt)
        src = Fn.to_compute(src)
        self_attmat, encdec_attmat, hidden = [], [], []
        for layer in self.trfm_layers:
            tgt, _self, _encdec = layer(tgt=tgt, tgt_mask=tgt_mask, src=src, src_mask=src_mask, _causal=True)
            self_attmat.append(_self)
            encdec_attmat.append(_encdec)
            hidden.append(tgt)
        if self.layernorm_first:
            tgt = _ln(self.layernorm, tgt)
        return tgt, self_attmat, encdec_attmat, hidden
