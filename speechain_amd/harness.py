"""The callers' sequencing, restated as a thin harness around the speechain_amd modules (SURVEY.md row a-harness).

SpeeChain's own ``ASREncoder`` / ``ARASRDecoder`` / ``ARASR`` (module/encoder/asr.py:103-131, module/decoder/ar_asr.py:49-78,
model/ar_asr.py:386-388,453-458,550) are not part of the hot path and are not re-implemented; with
``speechain_amd.register.register()`` they run unmodified on top of these modules.  This class wires the same modules in the
same order for parity tests, ``smoke()`` and ``bench.py``: [frontend -> SpecAugment -> Conv2dPrenet ->] encoder -> embedding ->
decoder -> token heads -> ``(1 - w) * CE + w * CTC``.
"""
import os

import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.cross_entropy import CrossEntropy
from speechain_amd.criterion.ctc import CTCLoss
from speechain_amd.module.postnet.token import TokenPostnet
from speechain_amd.module.transformer.decoder import TransformerDecoder
from speechain_amd.module.transformer.encoder import TransformerEncoder
from speechain_amd.utilbox.train_util import make_mask_from_len


_CTC_BRANCH = os.environ.get("SC_CTC_BRANCH", "1") == "1"  # (A/B knob)


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
        # The CTC head and loss read only the encoder output: they run (forward AND backward) on a second stream next to the decoder --
        # the CTC recursions occupy 32 workgroups for ~150 us, the decoder's launches fill the other CUs meanwhile.  SC_CTC_BRANCH=0: one stream.
        side = Fn._Side(enc, tag="ctc") if _CTC_BRANCH else None
        if side is not None:
            with side:
                ctc_logits = self.ctc_layer(side.entry(enc))
                ctc = self.ctc_loss(ctc_logits, feat_len, text, text_len)
        dec, _, _, _ = self.decoder(tgt=self.embed(dec_in), src=enc, tgt_mask=tgt_mask, src_mask=enc_mask)
        logits = self.postnet(dec)
        ce = self.ce_loss(logits=logits, text=text, text_len=text_len)
        if side is None:
            ctc_logits = self.ctc_layer(enc)
            ctc = self.ctc_loss(ctc_logits, feat_len, text, text_len)
        else:
            side.join()
        loss = (1 - self.ctc_weight) * ce + self.ctc_weight * ctc
        # (enc_hidden: the per-layer encoder outputs, encoder.py:339-343 -- a trainer that cuts backward into pieces uses them as cut points)
        return dict(enc=enc, logits=logits, ctc_logits=ctc_logits, ce=ce, ctc=ctc, loss=loss, enc_hidden=enc_hidden)


    # ---- inference: ARASR.inference step 1 (model/ar_asr.py:854-878) ----
    def decode_step_fn(self):
        """The decoder as ``beam_searching`` wants it: the stateless call of the reference plus cached ``start_decoding`` / ``forward_step``.
        One object per model: the replayed decoding session (buffers, hipGraphs) is kept on it between calls."""
        if getattr(self, "_decode_fn", None) is None:
            object.__setattr__(self, "_decode_fn", _HarnessDecoder(self))
        return self._decode_fn

    @torch.no_grad()
    def inference(self, infer_conf: dict, feat: torch.Tensor, feat_len: torch.Tensor):
        """Encoder -> beam search (ctc_layer as the CTC head) -> ``hypo_text``, ``hypo_text_len``, ``feat_token_len_ratio``,
        ``hypo_text_confid``.  feat (B, T', d_model) is the encoder input, as in :meth:`forward`."""
        from speechain_amd.infer_func.beam_search import beam_searching

        enc_mask = make_mask_from_len(feat_len, max_len=feat.size(1))
        enc, enc_mask, _, _ = self.encoder(feat, enc_mask)
        V = self.postnet.linear.out_features
        return beam_searching(enc_feat=enc, enc_feat_mask=enc_mask, asr_decode_fn=self.decode_step_fn(), ctc_decode_fn=self.ctc_layer,
                              vocab_size=V, sos_eos=V - 1, padding_idx=0, **infer_conf)


class _HarnessDecoder:
    """embedding -> decoder -> postnet of an :class:`ASRTransformer`, with the calling conventions of ``ARASRDecoder``."""

    def __init__(self, model: ASRTransformer):
        self.m = model

    def __call__(self, enc_feat, enc_feat_mask, text, text_len):
        m = self.m
        tgt_mask = make_mask_from_len(text_len, max_len=text.size(1))
        dec, sa, ea, hid = m.decoder(tgt=m.embed(text), src=enc_feat, tgt_mask=tgt_mask, src_mask=enc_feat_mask)
        return m.postnet(dec), sa, ea, hid

    def start_decoding(self, enc_feat, enc_feat_mask, beam_size: int, max_len: int, reuse=None):
        return self.m.decoder.init_cache(enc_feat, enc_feat_mask, beam_size, max_len, reuse)

    def forward_step(self, tokens, cache, beam_idx=None):
        m = self.m
        with torch.no_grad():
            if isinstance(getattr(m.embed, "embed", None), torch.nn.Embedding):  # (the lookup joins the step's first launch)
                return m.decoder.token_head_step(tokens, cache, beam_idx, m.postnet.linear, embedding=m.embed)
            emb = m.embed(tokens.unsqueeze(1)).squeeze(1)
            return m.decoder.token_head_step(emb, cache, beam_idx, m.postnet.linear)


class TTSFastSpeech2(torch.nn.Module):
    """FastSpeech2 training step as SpeeChain's ``FastSpeech2`` model wires it (model/nar_tts.py:240-470 module_forward, :472-560
    criterion_forward): TTSEncoder -> FastSpeech2Decoder -> feat L1 (before + after the postnet) + pitch / energy / log-duration L2.
    `conf`: the ``module_conf`` of a recipe (e.g. recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:150-249) -- ``enc_emb``,
    ``encoder``, the three predictors, ``feat_frontend``, the normalisation switches, ``decoder``, ``dec_postnet``."""

    def __init__(self, vocab_size: int, conf: dict, reduction_factor: int = 1):
        super().__init__()
        from speechain_amd.criterion.least_error import LeastError
        from speechain_amd.module.decoder.nar_tts import FastSpeech2Decoder
        from speechain_amd.module.encoder.tts import TTSEncoder

        self.encoder = TTSEncoder(vocab_size=vocab_size, embedding=conf["enc_emb"], encoder=conf["encoder"], prenet=conf.get("enc_prenet"))
        self.decoder = FastSpeech2Decoder(
            input_size=self.encoder.output_size, decoder=conf["decoder"], postnet=conf["dec_postnet"], pitch_predictor=conf["pitch_predictor"],
            energy_predictor=conf["energy_predictor"], duration_predictor=conf["duration_predictor"], feat_frontend=conf.get("feat_frontend"),
            feat_normalize=conf.get("feat_normalize", True), pitch_normalize=conf.get("pitch_normalize", True),
            energy_normalize=conf.get("energy_normalize", True), reduction_factor=reduction_factor)
        self.feat_loss = LeastError(loss_type="L1")
        self.pitch_loss, self.energy_loss, self.duration_loss = (LeastError(loss_type="L2") for _ in range(3))

    def prepare(self, duration, duration_len, feat, feat_len, pitch, pitch_len, epoch: int = 1):
        """The weight-independent head of a step (frontend, normalisations, frame counts, per-token targets): see
        FastSpeech2Decoder.prepare_targets.  Its result goes to ``forward(..., targets=...)``."""
        with torch.no_grad():
            return self.decoder.prepare_targets(duration=duration, duration_len=duration_len, pitch=pitch, pitch_len=pitch_len, feat=feat,
                                                feat_len=feat_len, epoch=epoch, min_frame_num=0)

    def forward(self, text, text_len, duration=None, duration_len=None, feat=None, feat_len=None, pitch=None, pitch_len=None, epoch: int = 1,
                targets=None):
        """text (B, L) WITHOUT <sos/eos> (module_forward strips them, model/nar_tts.py:366-368); duration (B, L) in any unit (rescaled to
        the frame count); feat (B, T_w, 1) waveforms; pitch (B, T) frame-level -- or `targets` from :meth:`prepare`."""
        enc_text, enc_mask, _, _ = self.encoder(text=text, text_len=text_len)
        (before, after, pred_len, tgt_feat, tgt_feat_len, pred_pitch, tgt_pitch, tgt_pitch_len, pred_energy, tgt_energy, tgt_energy_len,
         pred_dur, pred_gate, used_dur, used_dur_len, _, _) = self.decoder(
            enc_text=enc_text, enc_text_mask=enc_mask, duration=duration, duration_len=duration_len, pitch=pitch, pitch_len=pitch_len,
            feat=feat, feat_len=feat_len, epoch=epoch, min_frame_num=0, _targets=targets)
        terms = dict(feat_before=self.feat_loss(pred=before, tgt=tgt_feat, tgt_len=tgt_feat_len),
                     feat_after=self.feat_loss(pred=after, tgt=tgt_feat, tgt_len=tgt_feat_len),
                     pitch=self.pitch_loss(pred=pred_pitch, tgt=tgt_pitch, tgt_len=tgt_pitch_len),
                     energy=self.energy_loss(pred=pred_energy, tgt=tgt_energy, tgt_len=tgt_energy_len),
                     duration=self.duration_loss(pred=pred_dur, tgt=torch.log(used_dur.float() + 1), tgt_len=used_dur_len))
        out = dict(enc_text=enc_text, before=before, after=after, tgt_feat=tgt_feat, tgt_feat_len=tgt_feat_len, pred_pitch=pred_pitch,
                   tgt_pitch=tgt_pitch, pred_energy=pred_energy, tgt_energy=tgt_energy, pred_dur=pred_dur, used_dur=used_dur,
                   loss=sum(terms.values()))
        out.update({"loss_" + k: v for k, v in terms.items()})
        return out
