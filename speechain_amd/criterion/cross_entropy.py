"""Label-smoothed, length-masked cross entropy, drop-in for ``speechain.criterion.cross_entropy.CrossEntropy``
(speechain/criterion/cross_entropy.py:17-212).  The (B*L, V) log-softmax / gather / smoothing pass is one HIP kernel
(forward) + one (backward); the (B, L) masking, sentence sums and batch mean are tiny device-side torch ops.
Confidence thresholding and token re-weighting (:81-95, :148-199) are not on the accelerated path."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.abs import Criterion
from speechain_amd.utilbox.train_util import make_mask_from_len


class CrossEntropy(Criterion):
    def criterion_init(self, length_normalized: bool = False, label_smoothing: float = 0.0, temperature: float = 1.0,
                       confid_threshold: float = 0.0, confid_level: str = "sentence", token_vocab: str = None, new_weights=None):
        assert 0 <= label_smoothing < 1.0, f"The value of label_smoothing should be a float number in [0, 1), but got {label_smoothing}!"
        assert temperature > 0.0, f"The value of temperature should be a positive float number, but got {temperature}"
        if confid_threshold != 0.0 or new_weights is not None:
            raise NotImplementedError("speechain_amd CrossEntropy implements the recipes' path (confid_threshold=0, no token re-weighting)")
        self.length_normalized, self.label_smoothing, self.temperature = length_normalized, label_smoothing, temperature
        self.confid_threshold, self.confid_level, self.token_weights = confid_threshold, confid_level, None

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
        rows = Fn.cross_entropy_rows(logits.reshape(batch * seq_maxlen, vocab_size), text.reshape(-1).to(logits.device),
                                     self.label_smoothing, self.temperature)
        text_len = text_len.to(logits.device)
        valid = make_mask_from_len(text_len, max_len=seq_maxlen, return_3d=False)
        loss = rows.reshape(batch, seq_maxlen).masked_fill(~valid, 0.0).sum(dim=-1)
        if self.length_normalized:
            loss = loss / (text_len + 1e-10)
        return loss.mean()
