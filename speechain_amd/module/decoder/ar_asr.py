"""``ARASRDecoder``, drop-in for ``speechain.module.decoder.ar_asr.ARASRDecoder`` (speechain/module/decoder/ar_asr.py:16-78): token
embedding -> TransformerDecoder -> TokenPostnet, built from the same ``embedding`` / ``decoder`` configuration dictionaries, same
parameter names (``embedding.*``, ``decoder.*``, ``postnet.*``), same ``forward(enc_feat, enc_feat_mask, text, text_len)``.

On top of that surface: ``start_decoding`` / ``forward_step`` feed one token per hypothesis against cached keys and
values; ``speechain_amd.infer_func.beam_search.beam_searching`` uses them when the decode function offers them, where the reference
re-runs ``forward`` on the whole prefix at every step (speechain/infer_func/beam_search.py:325-333)."""
from typing import Dict

import torch

from speechain_amd.module.abs import Module
from speechain_amd.module.postnet.token import TokenPostnet
from speechain_amd.utilbox.import_util import import_class
from speechain_amd.utilbox.train_util import make_mask_from_len


class ARASRDecoder(Module):
    def module_init(self, embedding: Dict, decoder: Dict, vocab_size: int = None):
        embedding_class = import_class("speechain.module." + embedding["type"])
        self.embedding = embedding_class(vocab_size=vocab_size, **embedding.get("conf", dict()))
        decoder_class = import_class("speechain.module." + decoder["type"])
        self.decoder = decoder_class(input_size=self.embedding.output_size, **decoder.get("conf", dict()))
        self.postnet = TokenPostnet(input_size=self.decoder.output_size, vocab_size=vocab_size)

    def forward(self, enc_feat: torch.Tensor, enc_feat_mask: torch.Tensor, text: torch.Tensor, text_len: torch.Tensor):
        emb_text = self.embedding(text)
        text_mask = make_mask_from_len(text_len, max_len=text.size(1))
        if text.is_cuda:
            text_mask = text_mask.to(text.device)
        dec_feat, self_attmat, encdec_attmat, hidden = self.decoder(src=enc_feat, src_mask=enc_feat_mask, tgt=emb_text, tgt_mask=text_mask)
        return self.postnet(dec_feat), self_attmat, encdec_attmat, hidden

    # ---- cached decoding ----
    def start_decoding(self, enc_feat: torch.Tensor, enc_feat_mask: torch.Tensor, beam_size: int, max_len: int, reuse=None):
        """enc_feat (B, T, d), enc_feat_mask (B, 1, T) of the B sentences (one copy each, not one per beam)."""
        return self.decoder.init_cache(enc_feat, enc_feat_mask, beam_size, max_len, reuse)

    def forward_step(self, tokens: torch.Tensor, cache, beam_idx: torch.Tensor = None) -> torch.Tensor:
        """tokens (N,) int64: the newest token of every hypothesis row; returns the (N, vocab) logits of the next one."""
        with torch.no_grad():
            if hasattr(self.decoder, "token_head_step"):
                fused = isinstance(getattr(self.embedding, "embed", None), torch.nn.Embedding)  # (EmbedPrenet: the lookup joins the step's first launch)
                if fused:
                    return self.decoder.token_head_step(tokens, cache, beam_idx, self.postnet.linear, embedding=self.embedding)
                emb = self.embedding(tokens.unsqueeze(1)).squeeze(1)
                return self.decoder.token_head_step(emb, cache, beam_idx, self.postnet.linear)
            emb = self.embedding(tokens.unsqueeze(1)).squeeze(1)
            return self.postnet(self.decoder.forward_step(emb, cache, beam_idx))
