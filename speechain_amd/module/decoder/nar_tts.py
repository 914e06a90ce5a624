"""``FastSpeech2Decoder``, drop-in for ``speechain.module.decoder.nar_tts.FastSpeech2Decoder`` (speechain/module/decoder/nar_tts.py:21-502):
variance predictors on the encoded text, ground-truth (training) or predicted (inference) durations / pitch / energy, length
regulation, decoder stack, ``feat_pred`` and the residual postnet.  Same constructor dictionaries, same 17-tuple, same parameter names.

What runs where: the frontend, the three normalisations, the predictors, the decoder stack, ``feat_pred`` and the postnet are the HIP-backed
drop-ins of this package; the two pieces of the reference that are Python loops over the batch become one launch each (csrc/tts.hip):
the expansion of the token rows by their durations (:439-472) and ``average_scalar_by_duration`` (:150-204, a (B, L, T) index cube
there).  Lengths and durations are (B,)- / (B, L)-sized device arithmetic with the reference's formulas.  Speaker embeddings
(``spk_emb``: multi-speaker recipes, nar_tts.py:42-52, :257-263) go through ``speechain_amd.module.prenet.spk_embed.SpeakerEmbedPrenet``."""
import copy
import warnings
from typing import Dict

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module
from speechain_amd.module.norm.feat_norm import FeatureNormalization
from speechain_amd.module.prenet.spk_embed import SpeakerEmbedPrenet
from speechain_amd.utilbox.import_util import import_class
from speechain_amd.utilbox.train_util import make_len_from_mask, make_mask_from_len


class FastSpeech2Decoder(Module):
    def module_init(self, decoder: Dict, postnet: Dict, pitch_predictor: Dict, energy_predictor: Dict, duration_predictor: Dict,
                    spk_emb: Dict = None, feat_frontend: Dict = None, feat_normalize: Dict or bool = True, pitch_normalize: Dict or bool = True,
                    energy_normalize: Dict or bool = True, reduction_factor: int = 1):
        self.reduction_factor = reduction_factor
        if spk_emb is not None:  # (nar_tts.py:42-52: d_model is forced to the decoder's input size)
            spk_emb = copy.deepcopy(spk_emb)
            if "d_model" in spk_emb.keys() and spk_emb["d_model"] != self.input_size:
                warnings.warn(f"Your input d_model ({spk_emb['d_model']}) is different from the one automatically "
                              f"generated during model initialization ({self.input_size})."
                              f"Currently, the automatically generated one is used.")
            spk_emb["d_model"] = self.input_size
            self.spk_emb = SpeakerEmbedPrenet(**spk_emb)
        decoder, postnet, feat_frontend = copy.deepcopy(decoder), copy.deepcopy(postnet), copy.deepcopy(feat_frontend)

        def build(conf, **forced):
            conf = copy.deepcopy(conf)
            c = conf.get("conf", {})
            c.update(forced)
            return import_class("speechain.module." + conf["type"])(input_size=self.input_size, **c)

        self.duration_predictor = build(duration_predictor, use_conv_emb=False)
        self.pitch_predictor = build(pitch_predictor, use_conv_emb=True)
        self.energy_predictor = build(energy_predictor, use_conv_emb=True)
        if feat_frontend is not None:
            fc = feat_frontend.get("conf", {})
            fc["return_energy"] = True
            self.feat_frontend = import_class("speechain.module." + feat_frontend["type"])(**fc)
            feat_dim = self.feat_frontend.output_size * self.reduction_factor
        else:
            feat_dim = None
        for name, conf, size in (("feat_normalize", feat_normalize, feat_dim), ("energy_normalize", energy_normalize, 1),
                                 ("pitch_normalize", pitch_normalize, 1)):
            if conf is not None and conf is not False:
                setattr(self, name, FeatureNormalization(input_size=size, distributed=self.distributed, **(dict() if conf is True else conf)))
        self.decoder = import_class("speechain.module." + decoder["type"])(input_size=self.input_size, **decoder.get("conf", {}))
        self.feat_pred = torch.nn.Linear(in_features=self.decoder.output_size, out_features=feat_dim)
        self.postnet = import_class("speechain.module." + postnet["type"])(input_size=feat_dim, **postnet.get("conf", {}))

    @staticmethod
    def average_scalar_by_duration(frame_scalar: torch.Tensor, duration: torch.Tensor, duration_len: torch.Tensor):
        return Fn.avg_by_duration(frame_scalar, duration), duration_len

    def proc_duration(self, duration: torch.Tensor, min_frame_num: int = 0, max_frame_num: int = None, duration_alpha: torch.Tensor = None):
        if not self.training and duration_alpha is not None:
            duration = duration * duration_alpha
        duration = torch.clamp(torch.round(duration), min=0)
        zero = duration == 0
        duration = torch.clamp(duration, min=round(min_frame_num / self.reduction_factor),
                               max=None if max_frame_num is None else round(max_frame_num / self.reduction_factor))
        return torch.where(zero, torch.zeros_like(duration), duration)

    def _reduce(self, x, x_len, is_feat):
        """Frame grouping by the reduction factor (nar_tts.py:284-299 features, :304-319 / :324-339 scalar tracks)."""
        r = self.reduction_factor
        if r == 1:
            return x, x_len
        res = x.size(1) % r
        if res != 0:
            x = x[:, :-res]
        if is_feat:
            B, T, F = x.shape
            x = x.reshape(B, T // r, F * r)
        else:
            B, T = x.shape
            x = x.reshape(B, T // r, r).mean(dim=-1)
        return x, torch.div(x_len, r, rounding_mode="floor").type(torch.long)

    def prepare_targets(self, duration: torch.Tensor = None, duration_len: torch.Tensor = None, pitch: torch.Tensor = None,
                        pitch_len: torch.Tensor = None, feat: torch.Tensor = None, feat_len: torch.Tensor = None, energy: torch.Tensor = None,
                        energy_len: torch.Tensor = None, spk_ids: torch.Tensor = None, epoch: int = None, min_frame_num: int = None,
                        max_frame_num: int = None, duration_alpha: torch.Tensor = None):
        """The part of ``forward`` that depends on the batch only, not on the weights (nar_tts.py:266-340, :377-386, :404-407, :421-424):
        frontend, the three normalisations (with their running statistics), frame grouping, the frame counts per token and the per-token
        pitch / energy targets.  A trainer that captures the weight-dependent part into a hipGraph runs this eagerly (the normalisations
        re-create their buffers on every call) and feeds the result to ``forward(..., _targets=...)``."""
        if feat is not None:
            if feat.size(-1) == 1:
                assert hasattr(self, "feat_frontend"), \
                    f"Currently, {self.__class__.__name__} doesn't support time-domain TTS. Please specify a feature extraction frontend."
                feat, feat_len, energy, energy_len = self.feat_frontend(feat, feat_len)  # (float32 inside, whatever the compute dtype)
            if hasattr(self, "feat_normalize"):
                feat, feat_len = self.feat_normalize(feat, feat_len, group_ids=spk_ids, epoch=epoch)
            feat, feat_len = self._reduce(feat, feat_len, True)
        if pitch is not None:
            if hasattr(self, "pitch_normalize"):
                pitch, pitch_len = self.pitch_normalize(pitch, pitch_len, group_ids=spk_ids, epoch=epoch)
            pitch, pitch_len = self._reduce(pitch, pitch_len, False)
        if energy is not None:
            if hasattr(self, "energy_normalize"):
                energy, energy_len = self.energy_normalize(energy, energy_len, group_ids=spk_ids, epoch=epoch)
            energy, energy_len = self._reduce(energy, energy_len, False)
        used_duration = used_duration_len = None
        if duration is not None:
            # ground-truth durations (seconds or any unit), rescaled to the utterance's number of frames (nar_tts.py:377-386)
            used_duration = self.proc_duration(duration / duration.sum(dim=-1, keepdims=True) * feat_len.unsqueeze(-1),
                                               min_frame_num=min_frame_num, max_frame_num=max_frame_num, duration_alpha=duration_alpha)
            used_duration_len = duration_len
            if pitch is not None:
                pitch, pitch_len = self.average_scalar_by_duration(pitch, used_duration, used_duration_len)
            if energy is not None:
                energy, energy_len = self.average_scalar_by_duration(energy, used_duration, used_duration_len)
        return dict(feat=feat, feat_len=feat_len, pitch=pitch, pitch_len=pitch_len, energy=energy, energy_len=energy_len,
                    used_duration=used_duration, used_duration_len=used_duration_len)

    def forward(self, enc_text: torch.Tensor, enc_text_mask: torch.Tensor, duration: torch.Tensor = None, duration_len: torch.Tensor = None,
                pitch: torch.Tensor = None, pitch_len: torch.Tensor = None, feat: torch.Tensor = None, feat_len: torch.Tensor = None,
                energy: torch.Tensor = None, energy_len: torch.Tensor = None, spk_feat: torch.Tensor = None, spk_ids: torch.Tensor = None,
                epoch: int = None, min_frame_num: int = None, max_frame_num: int = None, duration_alpha: torch.Tensor = None,
                energy_alpha: torch.Tensor = None, pitch_alpha: torch.Tensor = None, _targets: Dict = None):
        if hasattr(self, "spk_emb"):  # (nar_tts.py:257-263) speaker vectors into the encoded text, before everything else reads it
            spk_feat_lookup, spk_feat = self.spk_emb(spk_ids=spk_ids, spk_feat=spk_feat)
            enc_text, _ = self.spk_emb.combine_spk_feat(spk_feat=spk_feat, spk_feat_lookup=spk_feat_lookup, enc_output=enc_text)
        tg = _targets if _targets is not None else self.prepare_targets(
            duration=duration, duration_len=duration_len, pitch=pitch, pitch_len=pitch_len, feat=feat, feat_len=feat_len, energy=energy,
            energy_len=energy_len, spk_ids=spk_ids, epoch=epoch, min_frame_num=min_frame_num, max_frame_num=max_frame_num,
            duration_alpha=duration_alpha)
        feat, feat_len, pitch, pitch_len, energy, energy_len = (tg[k] for k in ("feat", "feat_len", "pitch", "pitch_len", "energy", "energy_len"))
        used_duration, used_duration_len = tg["used_duration"], tg["used_duration_len"]
        if self.training:
            assert energy is not None and energy_len is not None and pitch is not None and pitch_len is not None and used_duration is not None \
                and used_duration_len is not None, "During training, please give the ground-truth of energy, pitch, and duration."

        outs = self.duration_predictor(enc_text, make_len_from_mask(enc_text_mask))
        if len(outs) == 2:
            (pred_duration, enc_text_len), pred_duration_gate = outs, None
        elif len(outs) == 3:
            pred_duration, enc_text_len, pred_duration_gate = outs
        else:
            raise RuntimeError
        if used_duration is None:  # inference without ground-truth durations: the predictor's (nar_tts.py:387-402)
            pd = pred_duration.float()
            if pred_duration_gate is not None:
                pd = pd.masked_fill(pred_duration_gate > 0, 0.0)
            used_duration = self.proc_duration(torch.exp(pd) - 1, min_frame_num=min_frame_num, max_frame_num=max_frame_num,
                                               duration_alpha=duration_alpha)
            used_duration_len = enc_text_len
            used_duration = used_duration.masked_fill(~make_mask_from_len(enc_text_len, max_len=used_duration.size(1), return_3d=False), 0.0)
            if pitch is not None:
                pitch, pitch_len = self.average_scalar_by_duration(pitch, used_duration, used_duration_len)
            if energy is not None:
                energy, energy_len = self.average_scalar_by_duration(energy, used_duration, used_duration_len)

        pred_pitch, enc_text_len = self.pitch_predictor(enc_text, enc_text_len)
        used_pitch = pitch if self.training else pred_pitch
        if not self.training and pitch_alpha is not None:
            used_pitch = used_pitch * pitch_alpha
        emb_pitch = self.pitch_predictor.emb_pred_scalar(used_pitch)
        pred_energy, enc_text_len = self.energy_predictor(enc_text, enc_text_len)
        used_energy = energy if self.training else pred_energy
        if not self.training and energy_alpha is not None:
            used_energy = used_energy * energy_alpha
        emb_energy = self.energy_predictor.emb_pred_scalar(used_energy)
        enc_text = Fn.to_compute(enc_text) + emb_pitch + emb_energy

        # length regulation: one launch instead of the per-utterance repeat_interleave / pad / cat loop (nar_tts.py:439-472)
        if feat_len is not None:
            expand_len = feat_len
        else:
            expand_len = used_duration.sum(dim=-1).long()
        expand_maxlen = int(feat.size(1)) if feat is not None else int(expand_len.max().item())
        expanded = Fn.length_regulate(enc_text, used_duration, expand_maxlen)

        dec_feat, dec_feat_mask, dec_attmat, dec_hidden = self.decoder(expanded, make_mask_from_len(expand_len, max_len=expand_maxlen))
        pred_feat_before = Fn.linear(dec_feat, [self.feat_pred.weight], [self.feat_pred.bias])
        dec_len = make_len_from_mask(dec_feat_mask)
        pred_feat_after = pred_feat_before + self.postnet(pred_feat_before, dec_len)
        return (pred_feat_before, pred_feat_after, dec_len, feat, feat_len, pred_pitch, pitch, pitch_len, pred_energy, energy, energy_len,
                pred_duration, pred_duration_gate, used_duration, used_duration_len, dec_attmat, dec_hidden)
