"""CTC loss, drop-in for ``speechain.criterion.ctc.CTCLoss`` (speechain/criterion/ctc.py:6-80): log-softmax over the
vocabulary, targets stripped of their leading/trailing <sos/eos>, per-utterance negative log-likelihood with
``zero_infinity``, mean over the batch -- HIP kernels (row logsumexp, alpha recursion, beta + gradient)."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.abs import Criterion


class CTCLoss(Criterion):
    def criterion_init(self, blank: int = 0, zero_infinity: bool = True):
        self.blank, self.zero_infinity = blank, zero_infinity

    def __call__(self, ctc_logits: torch.Tensor, enc_feat_len: torch.Tensor, text: torch.Tensor, text_len: torch.Tensor):
        tgt, tgt_len = text[:, 1:-1], text_len - 2
        if tgt.size(1) == 0:
            tgt = text.new_zeros(text.size(0), 1)
        nll = Fn.ctc_nll(ctc_logits, enc_feat_len, tgt.contiguous(), tgt_len, self.blank, self.zero_infinity)
        return nll.mean()

    def recover(self, ctc_text: torch.Tensor, ctc_text_len: torch.Tensor):
        if ctc_text.dim() == 3:
            ctc_text = torch.argmax(ctc_text, dim=-1)
        ctc_text = ctc_text.cpu()
        out = [[] for _ in range(ctc_text.size(0))]
        for i in range(ctc_text.size(0)):
            prev = None
            for j in range(int(ctc_text_len[i])):
                tok = int(ctc_text[i][j])
                if tok != prev and tok != self.blank:
                    out[i].append(tok)
                prev = tok
        return out
