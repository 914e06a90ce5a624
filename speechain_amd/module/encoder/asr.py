"""``ASREncoder``, drop-in for ``speechain.module.encoder.asr.ASREncoder`` (speechain/module/encoder/asr.py:19-131): [frontend] ->
[FeatureNormalization] -> [SpecAugment, training only] -> [prenet] -> encoder stack, each part resolved from its YAML ``type:`` string,
member names ``frontend`` / ``normalize`` / ``specaug`` / ``prenet`` / ``encoder``.  Returns the encoder's 4-tuple
(hidden, mask, attention maps, per-layer hidden states)."""
import copy
from typing import Dict

import torch

from speechain_amd.module.abs import Module
from speechain_amd.module.augment.specaug import SpecAugment
from speechain_amd.module.norm.feat_norm import FeatureNormalization
from speechain_amd.utilbox.import_util import import_class
from speechain_amd.utilbox.train_util import make_mask_from_len


class ASREncoder(Module):
    def module_init(self, encoder: Dict, frontend: Dict = None, normalize: Dict or bool = None, specaug: Dict or bool = None,
                    prenet: Dict = None):
        encoder, frontend, prenet = copy.deepcopy(encoder), copy.deepcopy(frontend), copy.deepcopy(prenet)
        prev = None
        if frontend is not None:
            self.frontend = import_class("speechain.module." + frontend["type"])(**frontend.get("conf", {}))
            prev = self.frontend.output_size
        has_norm = normalize is not None and normalize is not False
        if has_norm:
            self.normalize = FeatureNormalization(input_size=prev, distributed=self.distributed, **(dict() if normalize is True else normalize))
            prev = self.normalize.output_size
        if specaug is not None and specaug is not False:
            self.specaug = SpecAugment(input_size=prev, feat_norm=has_norm, **(dict() if specaug is True else specaug))
            prev = self.specaug.output_size
        if prenet is not None:
            self.prenet = import_class("speechain.module." + prenet["type"])(input_size=prev, **prenet.get("conf", {}))
            prev = self.prenet.output_size
        self.encoder = import_class("speechain.module." + encoder["type"])(input_size=prev, **encoder.get("conf", {}))
        self.output_size = self.encoder.output_size

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor, epoch: int = None, domain: str = None):
        if feat.size(-1) == 1:  # a waveform: the frontend always computes in f32 (asr.py:105-109)
            assert hasattr(self, "frontend"), "Currently, we don't support time-domain ASR. Please specify a feature extraction frontend!"
            feat, feat_len = self.frontend(feat, feat_len)
        if hasattr(self, "normalize"):
            feat, feat_len = self.normalize(feat, feat_len, epoch=epoch, group_ids=domain)
        if self.training and hasattr(self, "specaug"):
            feat, feat_len = self.specaug(feat, feat_len)
        if hasattr(self, "prenet"):
            feat, feat_len = self.prenet(feat, feat_len)
        feat_mask = make_mask_from_len(feat_len, max_len=feat.size(1))
        if feat.is_cuda:
            feat_mask = feat_mask.to(feat.device)
        return self.encoder(feat, feat_mask)
