"""The callers' sequencing, restated as a thin harness around the speechain_amd modules (SURVEY.md row a-harness).

SpeeChain's own ``ASREncoder`` / ``ARASRDecoder`` / ``ARASR`` (module/encoder/asr.py:103-131, module/decoder/ar_asr.py:49-78,
model/ar_asr.py:386-388,453-458,550) are not part of the hot path and are not re-implemented; with
``speechain_amd.register.register()`` they run unmodified on top of these modules.  This class wires the same modules in the
same order for parity tests, ``smoke()`` and ``bench.py``: [frontend -> SpecAugment -> Conv2dPrenet ->] encoder -> embedding ->
decoder -> token heads -> ``(1 - w) * CE + w * CTC``.
"""
import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.cross_entropy import CrossEntropy
from speechain_amd.criterion.ctc import CTCLoss
from speechain_amd.module.postnet.token import TokenPostnet
from speechain_amd.module.transformer.decoder import TransformerDecoder
from speechain_amd.module.transformer.encoder import TransformerEncoder
from speechain_amd.utilbox.train_util import make_mask_from_len


class Embed(torch.nn.Module):
    """Token embedding with the reference's parameter name ``embed.weight`` (module/prenet/embed.py:14-66).  The lookup is a torch gather;
    the table gradient comes from ``sc_embedding_bwd`` (deterministic, written straight into its gradient-bucket slot)."""

    def __init__(self, vocab_size: int, dim: int, padding_idx: int = 0):
        super().__init__()
        self.embed = torch.nn.Embedding(vocab_size, dim, padding_idx=padding_idx)  # parameter holder (and its initialisation)

    def forward(self, text):
        return Fn.embedding(self.embed.weight, text, self.embed.padding_idx)


class ASRTransformer(torch.nn.Module):
    def __init__(self, d_model=512, num_heads=8, enc_layers=12, dec_layers=6, fdfwd_dim=2048, vocab_size=5000, dropout=0.1,
                 activation="GELU", posenc_maxlen=5000, label_smoothing=0.2, ctc_weight=0.3):
        super().__init__()
        common = dict(d_model=d_model, num_heads=num_heads, fdfwd_dim=fdfwd_dim, posenc_dropout=dropout, att_dropout=dropout,
                      fdfwd_dropout=dropout, res_dropout=dropout, fdfwd_activation=activation, posenc_maxlen=posenc_maxlen,
                      emb_scale=False, layernorm_first=True)
        self.encoder = TransformerEncoder(num_layers=enc_layers, **common)
        self.embed = Embed(vocab_size, d_model)
        self.decoder = TransformerDecoder(num_layers=dec_layers, **common)
        self.postnet = TokenPostnet(input_size=d_model, vocab_size=vocab_size)
        self.ctc_layer = TokenPostnet(input_size=d_model, vocab_size=vocab_size)
        self.ce_loss = CrossEntropy(label_smoothing=label_smoothing)
        self.ctc_loss = CTCLoss()
        self.ctc_weight = ctc_weight

    def forward(self, feat, feat_len, text, text_len):
        """feat (B, T', d_model) encoder input; text (B, L) with <sos/eos> at both ends of every sentence."""
        enc_mask = make_mask_from_len(feat_len, max_len=feat.size(1))
        enc, enc_mask, _, enc_hidden = self.encoder(feat, enc_mask)
        # the trailing <sos/eos> of every sentence becomes padding, then the last column is dropped (model/ar_asr.py:386-388)
        pos = torch.arange(text.size(1), device=text.device)[None, :]
        dec_in = text.masked_fill(pos == (text_len - 1)[:, None], 0)  # (no indexed assignment: that would sync the host)
        dec_in, dec_len = dec_in[:, :-1], text_len - 1
        tgt_mask = make_mask_from_len(dec_len, max_len=dec_in.size(1))
        dec, _, _, _ = self.decoder(tgt=self.embed(dec_in), src=enc, tgt_mask=tgt_mask, src_mask=enc_mask)
        logits = self.postnet(dec)
        ctc_logits = self.ctc_layer(enc)
        ce = self.ce_loss(logits=logits, text=text, text_len=text_len)
        ctc = self.ctc_loss(ctc_logits, feat_len, text, text_len)
        loss = (1 - self.ctc_weight) * ce + self.ctc_weight * ctc
        # (enc_hidden: the per-layer encoder outputs, encoder.py:339-343 -- a trainer that cuts backward into pieces uses them as cut points)
        return dict(enc=enc, logits=logits, ctc_logits=ctc_logits, ce=ce, ctc=ctc, loss=loss, enc_hidden=enc_hidden)
