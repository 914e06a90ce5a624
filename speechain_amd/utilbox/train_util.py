"""Length -> mask helpers (speechain/utilbox/train_util.py:31-90), computed on the lengths' own device without the
reference's per-utterance Python loop and host->device copy."""
import torch


def make_mask_from_len(data_len: torch.Tensor, max_len: int = None, mask_type: torch.dtype = torch.bool, return_3d: bool = True):
    """(B, 1, max_len) [or (B, max_len)] mask, True/1 for positions < len (train_util.py:31-75)."""
    if max_len is None:
        max_len = int(data_len.max())
    elif not (data_len.is_cuda and torch.cuda.is_current_stream_capturing()):  # the check syncs with the host
        assert max_len >= data_len.max(), "max_len should be larger than the maximum of data_len!"
    mask = (torch.arange(max_len, device=data_len.device)[None, :] < data_len[:, None]).to(mask_type)
    return mask[:, None, :] if return_3d else mask


def make_len_from_mask(data_mask: torch.Tensor):
    if data_mask.dim() == 3:
        data_mask = data_mask.squeeze(1)
    return data_mask.sum(dim=-1)
