"""Label-smoothed, length-masked cross entropy, drop-in for ``speechain.criterion.cross_entropy.CrossEntropy``
(speechain/criterion/cross_entropy.py:17-212), token re-weighting (:81-95, :148-149) and confidence thresholding (:160-199) included.
The (B*L, V) log-softmax / gather / smoothing pass is one HIP kernel (forward) + one (backward); the (B, L)-sized masks, weights,
sentence sums and the batch mean are tiny device-side torch ops on its per-token output."""
import math

import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.abs import Criterion
from speechain_amd.utilbox.train_util import make_mask_from_len


class CrossEntropy(Criterion):
    def criterion_init(self, length_normalized: bool = False, label_smoothing: float = 0.0, temperature: float = 1.0,
                       confid_threshold: float = 0.0, confid_level: str = "sentence", token_vocab: str = None, new_weights=None):
        assert 0 <= label_smoothing < 1.0, f"The value of label_smoothing should be a float number in [0, 1), but got {label_smoothing}!"
        assert temperature > 0.0, f"The value of temperature should be a positive float number, but got {temperature}"
        assert 0 <= confid_threshold < 1.0, f"The value of confid_threshold should be a float number in [0, 1), but got {confid_threshold}!"
        assert confid_level in ["sentence", "token"], f"confid_level must be one of ['sentence', 'token'], but got {confid_level}"
        self.length_normalized, self.label_smoothing, self.temperature = length_normalized, label_smoothing, temperature
        self.confid_threshold, self.confid_level, self.token_weights = confid_threshold, confid_level, None
        if new_weights is not None:  # one weight per vocabulary line, 1 unless the dictionary names the token (:85-95)
            assert token_vocab is not None, "Please specify a token dictionary by 'token_vocab' if you want to customize the token weights."
            with open(token_vocab, encoding="utf-8") as f:
                tokens = [line.rstrip("\n") for line in f if line.rstrip("\n") != ""]
            index = {tok: i for i, tok in enumerate(tokens)}
            self.token_weights = torch.ones(len(tokens))
            for token, weight in new_weights.items():
                self.token_weights[index[token]] = weight

    def __call__(self, logits: torch.Tensor, text: torch.Tensor, text_len: torch.Tensor):
        if logits.size(1) == text.size(1) - 1:
            if not (text_len.is_cuda and torch.cuda.is_current_stream_capturing()):  # a host sync: skipped under capture
                assert text_len.max() == text.size(1), (
                    "There is a mismatch of the sentence length between text and text_len. "
                    f"got text_len.max()={text_len.max()} and text.size(1)={text.size(1)}.")
            text = text[:, 1:]
            text_len = text_len - 1
        elif logits.size(1) != text.size(1):
            raise RuntimeError
        batch, seq_maxlen, vocab_size = logits.size()
        dev = logits.device
        target = text.reshape(-1).to(dev)
        logits2 = logits.reshape(batch * seq_maxlen, vocab_size)
        rows = Fn.cross_entropy_rows(logits2, target, self.label_smoothing, self.temperature)  # -(smoothed log-probability) per token
        if self.token_weights is not None:
            if self.token_weights.device != dev:
                self.token_weights = self.token_weights.to(dev)
            rows = rows * self.token_weights.index_select(0, target)
        text_len = text_len.to(dev)
        if self.token_weights is None and self.confid_threshold == 0 and rows.is_cuda and rows.dtype == torch.float32:
            # the plain case (every recipe): mask, per-sentence sums, length normalisation and the batch mean in one launch each way
            return Fn.seq_loss(rows.reshape(batch, seq_maxlen), text_len, self.length_normalized)
        valid = make_mask_from_len(text_len, max_len=seq_maxlen, return_3d=False)
        drop = ~valid
        valid_sent = None
        if self.confid_threshold > 0:
            # the target's log-probability, for the masks only (no gradient flows through a comparison)
            with torch.no_grad():
                lp_t = torch.log_softmax(logits2.float() / self.temperature, dim=-1).gather(1, target[:, None]).reshape(batch, seq_maxlen)
            if self.confid_level == "token":
                low = lp_t <= math.log(self.confid_threshold)
                drop = drop | low
                text_len = (~low).masked_fill(~valid, False).sum(dim=-1)
                valid_sent = text_len > 0
            else:
                sent_lp = lp_t.masked_fill(~valid, 0.0).sum(dim=-1, keepdim=True)
                low = sent_lp <= (text_len * math.log(self.confid_threshold)).unsqueeze(-1)
                drop = drop | low
                valid_sent = ~low.squeeze(-1)
        loss = rows.reshape(batch, seq_maxlen).masked_fill(drop, 0.0).sum(dim=-1)
        if self.length_normalized:
            loss = loss / (text_len + 1e-10)
        return loss.mean() if self.confid_threshold == 0 else loss.sum() / (valid_sent.sum() + 1e-10)
