"""Transformer decoder, drop-in for ``speechain.module.transformer.decoder`` (speechain/module/transformer/decoder.py):
``TransformerDecoderLayer`` (:16-175; ``self_att``, ``encdec_att``, ``feed_forward``, ``self_att_ln``,
``encdec_att_ln``, ``fdfwd_ln``) and ``TransformerDecoder`` (:178-362; ``forward(tgt, src, tgt_mask, src_mask,
return_att, return_hidden) -> (out, self_attmat[L], encdec_attmat[L], hidden[L])``).

Quirks kept: the enc-dec attention never scales by head (:76-78), the decoder never forwards ``layernorm_first`` to its
layers (:271-285), ``emb_scale`` defaults to True (:187).  The causal mask is not materialised: the softmax kernel
combines the (B,1,L) padding mask with ``k <= q``.
"""
import math
import os

import torch
from torch import nn

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.transformer.attention import MultiHeadedAttention
from speechain_amd.module.transformer.encoder import _ln, _ln_res
from speechain_amd.module.transformer.feed_forward import PositionwiseFeedForward
from speechain_amd.module.transformer.pos_enc import PositionalEncoding


_CROSSKV_BRANCH = os.environ.get("SC_CROSSKV_BRANCH", "1") == "1"  # (A/B knob)


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

    def forward(self, tgt: torch.Tensor, src: torch.Tensor, tgt_mask: torch.Tensor, src_mask: torch.Tensor, *, _causal: bool = False,
                _cross_kv=None):
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
        y, encdec_attmat = self.encdec_att(src, src, x_norm, mask=src_mask, _residual=x, _res_drop=p, _kv=_cross_kv)
        if not first:
            y = _ln(self.encdec_att_ln, y)
        y_norm, y = _ln_res(self.fdfwd_ln, y) if first else (y, y)
        out = self.feed_forward(y_norm, _residual=y, _res_drop=p)
        if not first:
            out = _ln(self.fdfwd_ln, out)
        return out, self_attmat, encdec_attmat


    # ---- cached decoding (inference) ----
    def cross_kv(self, src: torch.Tensor) -> torch.Tensor:
        """(B, Tk, 2 d) keys | values of the encoder-decoder attention, computed once per utterance."""
        a = self.encdec_att
        return Fn.linear(Fn.to_compute(src), [a.k_layer.weight, a.v_layer.weight], [a.k_layer.bias, a.v_layer.bias])

    def forward_step(self, x: torch.Tensor, self_cache: torch.Tensor, table: torch.Tensor, pos: int, cross_kv: torch.Tensor,
                     src_mask: torch.Tensor, beam: int) -> torch.Tensor:
        """One position per hypothesis row: x (N, d) -> (N, d).  Same arithmetic as :meth:`forward` on the last position of a prefix,
        with the earlier positions' keys / values taken from ``self_cache`` (N, Tmax, 2 d) through the ancestry ``table`` and the
        encoder side from ``cross_kv`` (B, Tk, 2 d), shared by the ``beam`` adjacent rows of a sentence."""
        first = self.layernorm_first
        sa, ea, ff = self.self_att, self.encdec_att, self.feed_forward
        d, H = sa.d_model, sa.num_heads
        # every product of the step is one launch of csrc/decode_gemm.hip (Fn.decode_linear): a pre-LN layer's LayerNorm is folded in front of the
        # product that consumes it, the residual connection behind the one that closes a sublayer
        pre = lambda ln: ln if first else None
        post = lambda ln, v: v if first else _ln(ln, v)
        qkv = Fn.decode_linear(x, [sa.q_layer.weight, sa.k_layer.weight, sa.v_layer.weight], [sa.q_layer.bias, sa.k_layer.bias, sa.v_layer.bias],
                               ln=pre(self.self_att_ln))
        ctx = Fn.attention_decode(qkv, self_cache, table, pos, H, sa.scale)
        x = post(self.self_att_ln, Fn.decode_linear(ctx, [sa.output_layer.weight], [sa.output_layer.bias], residual=x))
        q = Fn.decode_linear(x, [ea.q_layer.weight], [ea.q_layer.bias], ln=pre(self.encdec_att_ln))
        B = cross_kv.size(0)
        ctx, _ = Fn.attention_core(q.view(B, beam, d), cross_kv, 0, 0, 1, H, d, src_mask, False, ea.scale, 0.0)
        x = post(self.encdec_att_ln, Fn.decode_linear(ctx.reshape(B * beam, d), [ea.output_layer.weight], [ea.output_layer.bias], residual=x))
        if isinstance(ff.in_layer, nn.Linear):
            hmid = Fn.decode_linear(x, [ff.in_layer.weight], [ff.in_layer.bias], ln=pre(self.fdfwd_ln), activation=ff.activation_name)
            x = Fn.decode_linear(hmid, [ff.out_layer.weight], [ff.out_layer.bias], residual=x)
        else:  # (convolutional feed-forward: no single-position form; as before)
            xn = _ln(self.fdfwd_ln, x) if first else x
            x = ff(xn, _residual=x, _res_drop=0.0)
        return post(self.fdfwd_ln, x)


class DecodeCache:
    """State of a cached decoding run over N = sentences * beam hypothesis rows (see csrc/decode.hip): per layer the self-attention
    key | value slots (N, Tmax, 2 d) and the encoder-side keys | values (B, Tk, 2 d); the two ancestry tables; the next position."""

    def __init__(self, self_kv, cross_kv, src_mask, beam: int, max_len: int, device):
        self.self_kv, self.cross_kv, self.src_mask, self.beam, self.max_len = self_kv, cross_kv, src_mask, beam, max_len
        N = self_kv[0].size(0)
        self.tables = [torch.zeros(N, max_len, dtype=torch.int32, device=device) for _ in range(2)]
        self.pos = 0


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
        # The encoder-side keys | values of every layer depend on the encoder output alone: they are projected up front on a second stream,
        # next to the first layer's self-attention -- and, since autograd runs a node's backward on its forward's stream, their data
        # gradients (products over all the encoder frames) run next to the backward of the decoder's own small products.
        # SC_CROSSKV_BRANCH=0: projected inside each layer, on one stream.
        cross = [None] * len(self.trfm_layers)
        if _CROSSKV_BRANCH and src.is_cuda:
            side = Fn._Side(src, tag="crosskv")
            with side:
                s_in = side.entry(src)
                kvs = [layer.cross_kv(s_in) for layer in self.trfm_layers]
            nop = lambda: None
            cross = [(kv, side.join if i == 0 else nop) for i, kv in enumerate(kvs)]
        for layer, ckv in zip(self.trfm_layers, cross):
            tgt, _self, _encdec = layer(tgt=tgt, tgt_mask=tgt_mask, src=src, src_mask=src_mask, _causal=True, _cross_kv=ckv)
            self_attmat.append(_self)
            encdec_attmat.append(_encdec)
            hidden.append(tgt)
        if self.layernorm_first:
            tgt = _ln(self.layernorm, tgt)
        return tgt, self_attmat, encdec_attmat, hidden

    # ---- cached decoding (inference): the reference re-runs forward() on the whole prefix at every step of its
    # beam search (speechain/infer_func/beam_search.py:325-333) ----
    def init_cache(self, src: torch.Tensor, src_mask: torch.Tensor, beam: int, max_len: int, reuse: DecodeCache = None) -> DecodeCache:
        """src (B, Tk, d) encoder output of the B sentences (NOT repeated per beam), src_mask (B, 1, Tk).  ``reuse``: a cache of the same
        geometry whose buffers are refilled in place (a decoding loop replayed from hipGraphs holds their addresses)."""
        src = Fn.to_compute(src)
        B = src.size(0)
        N, d = B * beam, self.d_model
        with torch.no_grad():
            cross = [layer.cross_kv(src) for layer in self.trfm_layers]
        if reuse is not None and reuse.beam == beam and reuse.max_len == max_len and reuse.cross_kv[0].shape == cross[0].shape \
                and reuse.cross_kv[0].dtype == cross[0].dtype and reuse.src_mask.shape == src_mask.shape:
            with torch.no_grad():
                for dst, new in zip(reuse.cross_kv, cross):
                    dst.copy_(new)
                reuse.src_mask.copy_(src_mask)
            reuse.pos = 0
            return reuse
        self_kv = [torch.zeros(N, max_len, 2 * d, dtype=src.dtype, device=src.device) for _ in self.trfm_layers]
        return DecodeCache(self_kv, cross, src_mask.clone(), beam, max_len, src.device)

    def forward_step(self, tgt: torch.Tensor, cache: DecodeCache, beam_idx: torch.Tensor = None, _skip_final_ln: bool = False,
                     _embed=None) -> torch.Tensor:
        """tgt (N, d): the embedded token every hypothesis row feeds at position ``cache.pos``; ``beam_idx`` (N,) int64: the row each
        hypothesis continues (the selection of the previous step; None = unchanged order).  Returns the (N, d) decoder output of that
        position -- what ``forward`` gives at the last position of the prefix.  ``_skip_final_ln``: the caller folds the stack's final
        LayerNorm (``self.layernorm``, pre-LN stacks) into the product that follows (``token_head_step``)."""
        pos = cache.pos
        if pos >= cache.max_len:
            raise RuntimeError(f"speechain_amd: the decoding cache holds {cache.max_len} positions")
        with torch.no_grad():
            t_in, t_out = cache.tables[pos & 1 ^ 1], cache.tables[pos & 1]
            pe = self.posenc
            if _embed is not None and Fn._DECODE_EMBED and not hasattr(pe, "emb_layernorm") and Fn.compute_dtype() in (torch.float32, torch.bfloat16):
                # tgt holds TOKENS: lookup, cast, positional encoding and the ancestry table in one launch (_embed = (table, scale))
                if pos >= pe.posenc.size(1):
                    pe.update_posenc(pos + 1)
                x = Fn.decode_embed_step(_embed[0], tgt, _embed[1], pe.posenc[0, pos], pe.alpha if hasattr(pe, "alpha") else None,
                                         math.sqrt(pe.d_model) if pe.emb_scale else 1.0, t_in if pos > 0 else None, t_out, beam_idx, pos)
            else:
                if _embed is not None:
                    tgt = _embed[2](tgt.unsqueeze(1)).squeeze(1)
                Fn.beam_table_update(t_in if pos > 0 else None, t_out, beam_idx, pos)
                x = pe.forward_step(tgt, pos)
            for layer, kv, ckv in zip(self.trfm_layers, cache.self_kv, cache.cross_kv):
                x = layer.forward_step(x, kv, t_out, pos, ckv, cache.src_mask, cache.beam)
            if self.layernorm_first and not _skip_final_ln:
                x = _ln(self.layernorm, x)
        cache.pos = pos + 1
        return x

    def token_head_step(self, tgt: torch.Tensor, cache: DecodeCache, beam_idx: torch.Tensor, head: nn.Linear, embedding=None) -> torch.Tensor:
        """``head(forward_step(...))`` with the final LayerNorm of a pre-LN stack folded into the head's product (one launch).  With
        ``embedding`` (an ``EmbedPrenet``) ``tgt`` holds the rows' newest TOKENS and the lookup joins the step's first launch."""
        emb = None
        if embedding is not None:
            w = embedding.embed.weight
            emb = (w.detach(), math.sqrt(w.size(1)) if getattr(embedding, "scale", False) else 1.0, embedding)  # (prenet/embed.py:46: x sqrt(dim))
            if w.dtype != torch.float32 or not w.is_contiguous():
                tgt, emb = embedding(tgt.unsqueeze(1)).squeeze(1), None
        x = self.forward_step(tgt, cache, beam_idx, _skip_final_ln=True, _embed=emb)
        with torch.no_grad():
            return Fn.decode_linear(x, [head.weight], [head.bias], ln=self.layernorm if self.layernorm_first else None)
