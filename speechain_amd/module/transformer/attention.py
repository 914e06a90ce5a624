"""Multi-head attention, drop-in for ``speechain.module.transformer.attention.MultiHeadedAttention``
(speechain/module/transformer/attention.py): parameters ``{k,v,q,output}_layer.{weight,bias}`` (:53-59), argument order
``forward(k, v, q, mask)`` returning ``(output, probs)`` (:110-133), scores scaled by 1/sqrt(d_model) unless
``scale_dp_by_head`` (:61-65), probabilities returned pre-dropout (:94).

HIP path: the Q/K/V projections are one fused GEMM when their inputs coincide (self-attention: one (B,T,3d) buffer;
enc-dec: q and a fused (B,T,2d) k|v buffer), QK^T / softmax / PV and the output projection run through
``speechain_amd.functional``; the residual add and residual dropout of the calling layer are fused into the output
projection's epilogue through the private ``_residual`` / ``_res_drop`` keywords.
"""
import math

import torch
import torch.nn as nn

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


class MultiHeadedAttention(Module):
    def module_init(self, num_heads: int, d_model: int, dropout: float = 0.1, scale_dp_by_head: bool = False):
        assert d_model % num_heads == 0, "d_model is not divisible by num_heads!"
        self.head_size = d_model // num_heads
        self.d_model, self.num_heads = d_model, num_heads
        self.k_layer = nn.Linear(d_model, num_heads * self.head_size)
        self.v_layer = nn.Linear(d_model, num_heads * self.head_size)
        self.q_layer = nn.Linear(d_model, num_heads * self.head_size)
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.output_layer = nn.Linear(d_model, d_model)
        self.scale = 1 / math.sqrt(self.head_size) if scale_dp_by_head else 1 / math.sqrt(self.d_model)

    def _project(self, k, v, q):
        """Returns (q_buf, kv_buf, iq, ik, iv): projection buffers and the column-block index of q / k / v in them."""
        K, V, Q = self.k_layer, self.v_layer, self.q_layer
        if (k is v) and (v is q):
            return Fn.linear(q, [Q.weight, K.weight, V.weight], [Q.bias, K.bias, V.bias]), None, 0, 1, 2
        qp = Fn.linear(q, [Q.weight], [Q.bias])
        if k is v:
            return qp, Fn.linear(k, [K.weight, V.weight], [K.bias, V.bias]), 0, 0, 1
        kv = torch.cat([Fn.linear(k, [K.weight], [K.bias]), Fn.linear(v, [V.weight], [V.bias])], dim=-1)
        return qp, kv, 0, 0, 1

    def kvq_forward(self, k: torch.Tensor, v: torch.Tensor, q: torch.Tensor):
        """(B, H, T, dh) projections (attention.py:67-81); kept for subclasses such as the Conformer's relative-position
        attention, which calls it and then :meth:`attention_forward`."""
        B = k.size(0)
        x = Fn.to_compute
        kp = Fn.linear(x(k), [self.k_layer.weight], [self.k_layer.bias]).view(B, -1, self.num_heads, self.head_size).transpose(1, 2)
        vp = Fn.linear(x(v), [self.v_layer.weight], [self.v_layer.bias]).view(B, -1, self.num_heads, self.head_size).transpose(1, 2)
        qp = Fn.linear(x(q), [self.q_layer.weight], [self.q_layer.bias]).view(B, -1, self.num_heads, self.head_size).transpose(1, 2)
        return kp, vp, qp

    def attention_forward(self, v: torch.Tensor, scores: torch.Tensor, mask: torch.Tensor):
        """Masked softmax -> dropout -> context -> output projection from externally computed scores (attention.py:83-108).
        Subclass hook only (not on the accelerated path): softmax/matmul here are device torch ops, the projection is HIP."""
        if mask is not None:
            scores = scores.masked_fill(~mask.unsqueeze(1), float("-inf"))
        attention = torch.softmax(scores.float(), dim=-1).to(v.dtype)
        score_soft = attention.clone()
        attention = self.dropout(attention)
        context = torch.matmul(attention, v).transpose(1, 2).contiguous().view(v.size(0), -1, self.num_heads * self.head_size)
        return Fn.linear(context, [self.output_layer.weight], [self.output_layer.bias]), score_soft

    def forward(self, k: torch.Tensor, v: torch.Tensor, q: torch.Tensor, mask: torch.Tensor = None, *, _residual=None,
                _res_drop: float = 0.0, _causal: bool = False, _kv=None):
        """``_kv``: ``(keys | values buffer (B, Tk, 2 d) already projected from k / v, callable to run before it is read)`` -- the decoder
        projects the encoder output for all its layers up front on a second stream (decoder.py)."""
        same_kv, same_all = k is v, (k is v) and (v is q)
        qc = Fn.to_compute(q)
        if _kv is not None:
            Q = self.q_layer
            q_buf, kv_buf, iq, ik, iv = Fn.linear(qc, [Q.weight], [Q.bias]), _kv[0], 0, 0, 1
            _kv[1]()
        else:
            kc = qc if same_all else Fn.to_compute(k)
            vc = kc if same_kv else Fn.to_compute(v)
            q_buf, kv_buf, iq, ik, iv = self._project(kc, vc, qc)
        p = self.dropout.p if self.training else 0.0
        context, probs = Fn.attention_core(q_buf, kv_buf, iq, ik, iv, self.num_heads, self.d_model, mask, _causal, self.scale, p)
        res = Fn.to_compute(_residual, context.dtype) if _residual is not None else None
        out = Fn.linear(context, [self.output_layer.weight], [self.output_layer.bias], drop_p=_res_drop, residual=res)
        return out, probs
