"""Sinusoidal positional encoding, drop-in for ``speechain.module.transformer.pos_enc.PositionalEncoding``
(speechain/module/transformer/pos_enc.py): same kwargs (:24-34), buffer ``posenc`` (1, max_len, d) (:115-146), optional
parameter ``alpha`` and sub-module ``emb_layernorm`` (:94-101), ``forward`` = [LN] -> [* sqrt(d)] -> + table*[alpha] ->
dropout (:148-182) in one HIP launch (plus the LayerNorm kernel when enabled)."""
import math
from typing import Dict

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


def sinusoid_table(max_len: int, d_model: int, posenc_type: str) -> torch.Tensor:
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * (math.log(10000.0) / d_model))
    table = torch.zeros(max_len, d_model)
    if posenc_type == "mix":
        table[:, 0::2] = torch.sin(position / div_term)
        table[:, 1::2] = torch.cos(position / div_term)
    else:
        div_ext = torch.exp(torch.arange(d_model, d_model * 2, 2, dtype=torch.float) * (math.log(10000.0) / d_model))
        table[:, : d_model // 2] = torch.sin(position / div_term)
        table[:, d_model // 2:] = torch.cos(position / div_ext)
    return table.unsqueeze(0)


class PositionalEncoding(Module):
    def module_init(self, posenc_type: str = "mix", d_model: int = 512, emb_scale: bool = False, emb_layernorm: bool = False,
                    posenc_scale: bool = False, init_alpha: float = 1.0, max_len: int = 5000, dropout: float = 0.0):
        assert posenc_type in ["mix", "sep"], f"The type of PositionalEncoding layer must be either 'mix' or 'sep', but got type={posenc_type}!"
        assert d_model % 2 == 0, f"Cannot apply sin/cos positional encoding to the vectors with odd dimensions (got d_model={d_model:d})."
        self.posenc_type, self.d_model, self.emb_scale = posenc_type, d_model, emb_scale
        if emb_layernorm:
            self.emb_layernorm = torch.nn.LayerNorm(d_model)
        self.init_alpha = float(init_alpha)
        if posenc_scale:
            self.alpha = torch.nn.Parameter(torch.tensor(self.init_alpha))
        self.update_posenc(max_len)
        self.dropout = torch.nn.Dropout(p=dropout)

    def reset_parameters(self):
        if hasattr(self, "alpha"):
            self.alpha.data = torch.tensor(self.init_alpha, device=self.alpha.device)

    def update_posenc(self, max_len: int):
        table = sinusoid_table(max_len, self.d_model, self.posenc_type)
        if hasattr(self, "posenc"):
            table = table.to(self.posenc.device)
        self.register_buffer("posenc", table)

    def forward(self, emb_feat: torch.Tensor):
        if emb_feat.size(1) > self.posenc.size(1):
            self.update_posenc(emb_feat.size(1))
        x = Fn.to_compute(emb_feat)
        if hasattr(self, "emb_layernorm"):
            ln = self.emb_layernorm
            x = Fn.layer_norm(x, ln.weight, ln.bias, ln.eps)
        p = self.dropout.p if self.training else 0.0
        return Fn.posenc_add(x, self.posenc, self.alpha if hasattr(self, "alpha") else None,
                             math.sqrt(self.d_model) if self.emb_scale else 1.0, p)

    def forward_step(self, emb: torch.Tensor, pos: int):
        """:meth:`forward` (evaluation mode) for the single position ``pos`` of every row: emb (N, d) -> (N, d)."""
        if pos >= self.posenc.size(1):
            self.update_posenc(pos + 1)
        x = Fn.to_compute(emb).unsqueeze(1)
        if hasattr(self, "emb_layernorm"):
            ln = self.emb_layernorm
            x = Fn.layer_norm(x, ln.weight, ln.bias, ln.eps)
        return Fn.posenc_add(x, self.posenc[:, pos:pos + 1], self.alpha if hasattr(self, "alpha") else None,
                             math.sqrt(self.d_model) if self.emb_scale else 1.0, 0.0).squeeze(1)

    def get_recordable_para(self) -> Dict or None:
        return dict(alpha=self.alpha) if hasattr(self, "alpha") else None

    def extra_repr(self) -> str:
        return f"emb_scale={self.emb_scale}\nposenc_scale={hasattr(self, 'alpha')}"
