"""``LeastError``, drop-in for ``speechain.criterion.least_error.LeastError`` (speechain/criterion/least_error.py:15-129): masked L1 / L2 /
L1+L2 between predicted and target feature tracks, normalised by the number of valid frames (or summed per utterance and averaged).
On the GPU one fused reduction (``sc_least_error_fwd`` / ``_bwd``); the elementwise form below serves the CPU, ``update_range`` and targets with a gradient."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.criterion.abs import Criterion
from speechain_amd.utilbox.train_util import make_mask_from_len


class LeastError(Criterion):
    def criterion_init(self, loss_type: str = "L2", is_normalized: bool = True, update_range: int or float = None):
        assert loss_type in ["L1", "L2", "L1+L2"], \
            f"You input loss_type must be one of 'L1', 'L2', and 'L1+L2'. But got loss_type={loss_type}."
        if isinstance(update_range, int):
            assert update_range < 0, f"For setting absolute updating range, you must input a negative integer, but got {update_range}."
        elif isinstance(update_range, float):
            assert 0 < update_range < 1, \
                f"For setting relative updating range, you must input a postive float number between 0 and 1, but got {update_range}."
        elif update_range is not None:
            raise TypeError
        self.loss_type, self.is_normalized, self.update_range = loss_type, is_normalized, update_range

    def __call__(self, pred: torch.Tensor, tgt: torch.Tensor, tgt_len: torch.Tensor):
        if pred.dim() == 2:
            pred = pred.unsqueeze(-1)
        elif pred.dim() != 3:
            raise RuntimeError
        if tgt.dim() == 2:
            tgt = tgt.unsqueeze(-1)
        elif tgt.dim() != 3:
            raise RuntimeError
        # one fused reduction on the GPU (Fn.least_error: two launches forward, one backward); everything below is the reference's
        # sequence of elementwise steps, kept for the CPU, for sliced feature ranges and for targets that carry a gradient
        if (pred.is_cuda and self.update_range is None and pred.shape == tgt.shape and pred.is_contiguous() and not tgt.requires_grad
                and pred.dtype in (torch.float32, torch.bfloat16)):
            return Fn.least_error(pred, tgt, tgt_len, self.loss_type, self.is_normalized)
        pred, tgt = pred.float(), tgt.float()
        batch_size, feat_maxlen, feat_dim = pred.shape
        if self.update_range is not None:
            ndim = int(feat_dim * self.update_range) if self.update_range > 0 else -self.update_range
            assert ndim < feat_dim
            pred, tgt = pred[:, :, :ndim], tgt[:, :, :ndim]
        mask = make_mask_from_len(tgt_len, max_len=feat_maxlen, return_3d=False)
        d = pred - tgt
        loss = {"L1": d.abs, "L2": lambda: d * d, "L1+L2": lambda: d.abs() + d * d}[self.loss_type]().mean(dim=-1)
        loss = loss.masked_fill(~mask, 0.0)
        if self.is_normalized:
            return loss.sum() / mask.sum()
        return loss.sum(dim=-1).mean()

    def extra_repr(self) -> str:
        return f"loss_type={self.loss_type}, is_normalized={self.is_normalized}"
