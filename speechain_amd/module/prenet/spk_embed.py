"""``SpeakerEmbedPrenet``, drop-in for ``speechain.module.prenet.spk_embed.SpeakerEmbedPrenet`` (speechain/module/prenet/spk_embed.py:7-230):
closed-set speaker lookup table and / or pretrained (open-set) speaker vectors, L2-normalised, then combined with the TTS encoder output (and
optionally the decoder input) by addition or concatenation and one projection back to ``d_model``.  Same keywords, same parameter names
(``spk_lookup.embed.weight``, ``pre_add_proj_lookup.*``, ``pre_add_proj_pretrain.*``, ``final_proj_enc.*``, ``final_proj_dec.*``), same two
calls: ``forward(spk_ids, spk_feat) -> (spk_feat_lookup, spk_feat)`` and ``combine_spk_feat(...) -> (enc_output, dec_input)``.

What runs where: the table lookup is ``functional.embedding`` (deterministic HIP gradient), every projection -- including the
(B L, d_model + speaker dims) -> d_model one of the 'concat' recipes (recipes/tts/vctk/exp_cfg/16khz_ecapa_mfa_fastspeech2_mic2.yaml:228-230:
384 + 192 -> 384) -- is ``functional.linear`` on the HIP GEMMs with the weight gradient in the grouped launch; the L2 normalisation of the
(B, dim) speaker vectors and the broadcast add / concat are (B, dim)- and (B, L, d)-sized torch element-wise ops."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.prenet.embed import EmbedPrenet


class SpeakerEmbedPrenet(Module):
    def module_init(self, d_model: int, spk_emb_dim_lookup: int = None, spk_emb_dim_pretrained: int = None, spk_num: int = None,
                    spk_emb_comb: str = "concat", dec_comb: bool = False, encdec_same_proj: bool = True):
        self.use_lookup, self.use_pretrain = spk_emb_dim_lookup is not None, spk_emb_dim_pretrained is not None
        assert self.use_lookup or self.use_pretrain, (
            "spk_emb_dim_lookup and spk_emb_dim_pretrained cannot be None at the same time! "
            "Please specify the value of at least one of them.")
        assert spk_emb_comb in ["add", "concat"], f"spk_emb_comb must be either 'add' or 'concat', but got {spk_emb_comb}."
        if self.use_lookup:
            self.spk_emb_dim_lookup = spk_emb_dim_lookup
            self.spk_lookup = EmbedPrenet(embedding_dim=self.spk_emb_dim_lookup, vocab_size=spk_num)
            if spk_emb_comb == "add" and self.spk_emb_dim_lookup != d_model:
                self.pre_add_proj_lookup = torch.nn.Linear(self.spk_emb_dim_lookup, d_model)
        self.spk_emb_comb = spk_emb_comb
        self.spk_emb_dim_pretrained = spk_emb_dim_pretrained
        if self.use_pretrain and spk_emb_comb == "add" and self.spk_emb_dim_pretrained != d_model:
            self.pre_add_proj_pretrain = torch.nn.Linear(self.spk_emb_dim_pretrained, d_model)
        proj_in_size = d_model
        if self.use_lookup and spk_emb_comb == "concat":
            proj_in_size += self.spk_emb_dim_lookup
        if self.use_pretrain and spk_emb_comb == "concat":
            proj_in_size += self.spk_emb_dim_pretrained
        self.final_proj_enc = torch.nn.Linear(proj_in_size, d_model)
        self.dec_comb, self.encdec_same_proj = dec_comb, encdec_same_proj
        if self.dec_comb and not self.encdec_same_proj:
            self.final_proj_dec = torch.nn.Linear(proj_in_size, d_model)
        self.output_size = d_model

    @staticmethod
    def _proj(x: torch.Tensor, layer: torch.nn.Linear) -> torch.Tensor:
        return Fn.linear(Fn.to_compute(x), [layer.weight], [layer.bias])

    def forward(self, spk_ids: torch.Tensor = None, spk_feat: torch.Tensor = None):
        if self.use_lookup:
            assert spk_ids is not None, "For lookup-based close-set TTS, you must pass spk_ids to SpeakerEmbedPrenet."
            spk_feat_lookup = self.spk_lookup(spk_ids)
            if hasattr(self, "pre_add_proj_lookup"):
                spk_feat_lookup = self._proj(spk_feat_lookup, self.pre_add_proj_lookup)
        else:
            spk_feat_lookup = None
        if self.use_pretrain:
            assert spk_feat is not None, "For pretrained-bsed open-set TTS, you must pass spk_feat to SpeakerEmbedPrenet."
            if hasattr(self, "pre_add_proj_pretrain"):
                spk_feat = self._proj(spk_feat, self.pre_add_proj_pretrain)
        else:
            spk_feat = None
        # unit vectors (spk_embed.py:140-151); float32 whatever the compute dtype: (B, dim) numbers
        if spk_feat_lookup is not None:
            spk_feat_lookup = torch.nn.functional.normalize(spk_feat_lookup.float(), dim=-1)
        if spk_feat is not None:
            spk_feat = torch.nn.functional.normalize(spk_feat.float(), dim=-1)
        return spk_feat_lookup, spk_feat

    def combine_spk_feat(self, spk_feat: torch.Tensor, spk_feat_lookup: torch.Tensor, enc_output: torch.Tensor, dec_input: torch.Tensor = None):
        if spk_feat is not None:
            if spk_feat.dim() == 2:
                spk_feat = spk_feat.unsqueeze(1)
            else:
                assert spk_feat.dim() == 3 and spk_feat.size(1) == 1, \
                    f"Something wrong happens to the dimension of the input spk_feat. Its dimension is {spk_feat.shape}."
        if spk_feat_lookup is not None:
            if spk_feat_lookup.dim() == 2:
                spk_feat_lookup = spk_feat_lookup.unsqueeze(1)
            else:
                assert spk_feat_lookup.dim() == 3 and spk_feat_lookup.size(1) == 1, (
                    f"Something wrong happens to the dimension of the input spk_feat_lookup. "
                    f"Its dimension is {spk_feat_lookup.shape}.")

        def combine(tgt: torch.Tensor, layer: torch.nn.Linear) -> torch.Tensor:
            tgt = Fn.to_compute(tgt)
            parts = [v.to(tgt.dtype) for v in (spk_feat, spk_feat_lookup) if v is not None]  # (pretrained first, then lookup: :204-213)
            if self.spk_emb_comb == "add":
                for v in parts:
                    tgt = tgt + v
            elif self.spk_emb_comb == "concat":
                tgt = torch.cat([tgt] + [v.expand(-1, tgt.size(1), -1) for v in parts], dim=-1)
            else:
                raise RuntimeError
            return Fn.linear(tgt, [layer.weight], [layer.bias])

        enc_output = combine(enc_output, self.final_proj_enc)
        if self.dec_comb:
            assert dec_input is not None, (
                "If you want to combine speaker embeddings with decoder input vectors, "
                "please give the decoder input vectors by the argument 'dec_input' in combine_spk_feat()")
            dec_input = combine(dec_input, self.final_proj_enc if self.encdec_same_proj else self.final_proj_dec)
        return enc_output, dec_input

    def extra_repr(self) -> str:
        return f"dec_comb={self.dec_comb}"
