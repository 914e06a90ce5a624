"""``speechain.module.norm.feat_norm.FeatureNormalization`` on gfx950.

Drop-in for speechain/module/norm/feat_norm.py: same kwargs (:53-77), the same lazily created buffers ``<prefix>_batch`` /
``<prefix>_mean`` / ``<prefix>_std`` / ``aver_mean`` / ``aver_std`` (re-registered on every update, :472-509), the same lazy
``_load_from_state_dict`` (:606-632), ``forward(feat, feat_len, group_ids=None, epoch=None)`` and ``recover``.

Device work -- the per-utterance statistics (a Python loop of B ``.mean`` / ``.std`` calls in the reference, :115-133) and the
normalisation of the padded tensor -- is two launches (``sc_featnorm_stats``, ``sc_featnorm_apply``); everything (B, F)-sized
(batch average, running average, DDP gathers) is torch arithmetic on the device with the reference's formulas, without host
synchronisation.  Only the per-utterance group ids (``norm_type='group'`` with a tensor of ids) are read on the host, as the
reference does, because they select buffers by name.

Mirrored on purpose: statistics over the valid frames but normalisation of the whole padded tensor; ``epoch=None`` restarts the
batch counter at every call; the average over groups includes the previous ``aver_*`` buffer (:519-528).
Deviations: the input tensor is never modified (the reference's group branch works in place, :216-230); ``mean_norm=False`` with
the global / group types, which raises AttributeError inside the reference (:475), is rejected at construction; there is no
gradient through the statistics (the module sits behind the frontend, whose input is the waveform): a ``feat`` that requires
grad is refused instead of silently detached.
"""
from typing import Optional

import torch

from speechain_amd._lib import lib, ptr, require_gpu, stream
from speechain_amd.module.abs import Module


def utterance_stats(feat: torch.Tensor, feat_len: torch.Tensor, clamp: float, want_mean: bool, want_std: bool):
    """(B, F) mean and clamped unbiased std over feat[i][:feat_len[i]] -- one launch (feat_norm.py:115-133)."""
    B, T, F = feat.shape
    mean = torch.empty(B, F, dtype=torch.float32, device=feat.device) if want_mean else None
    std = torch.empty(B, F, dtype=torch.float32, device=feat.device) if want_std else None
    lens = feat_len.to(device=feat.device, dtype=torch.int64)
    lib.call("sc_featnorm_stats", ptr(feat), feat.stride(0), feat.stride(1), ptr(lens), B, T, F, float(clamp), ptr(mean), ptr(std), stream())
    return mean, std


def apply_stats(feat: torch.Tensor, mean: Optional[torch.Tensor], std: Optional[torch.Tensor]) -> torch.Tensor:
    """(feat - mean) / std over the whole padded tensor; mean / std are (F,) shared or (B, F) per utterance, or None."""
    B, T, F = feat.shape
    out = torch.empty(B, T, F, dtype=torch.float32, device=feat.device)

    def prep(v):
        if v is None:
            return None, 0
        v = v.to(device=feat.device, dtype=torch.float32).contiguous()
        assert v.shape in ((F,), (B, F)), f"statistics of shape {tuple(v.shape)} for features {tuple(feat.shape)}"
        return v, (F if v.dim() == 2 else 0)

    m, mb = prep(mean)
    s, sb = prep(std)
    lib.call("sc_featnorm_apply", ptr(feat), ptr(out), feat.stride(0), feat.stride(1), out.stride(0), out.stride(1), B, T, F, ptr(m), mb,
             ptr(s), sb, stream())
    return out


def ddp_batch_statistics(curr_means: Optional[torch.Tensor], curr_stds: Optional[torch.Tensor], batch_size: int, group=None):
    """Batch-level mean / std across all processes: sum of the per-utterance vectors of every rank divided by the total number of
    utterances (feat_norm.py:236-265 with gather_scalars :341-351 and gather_vectors :353-394).  Pure torch + torch.distributed
    on the tensors' own device (RCCL on the GPU, gloo in the CPU test)."""
    import torch.distributed as dist

    ref = curr_means if curr_means is not None else curr_stds
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=ref.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([batch_size], dtype=torch.int64, device=ref.device), group=group)
    total = torch.cat(sizes).sum()

    def reduce(v):
        if v is None:
            return None
        mine = v.sum(dim=0)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        return torch.stack(parts).sum(dim=0) / total

    return reduce(curr_means), reduce(curr_stds)


class FeatureNormalization(Module):
    def module_init(self, norm_type: str = "global", mean_norm: bool = True, std_norm: bool = True, clamp: float = 1e-10,
                    max_epoch_num: int = 4):
        if norm_type not in ("utterance", "batch", "group", "global"):
            raise ValueError(f"norm_type can only be one of 'utterance', 'batch', 'group', 'global', but got norm_type={norm_type}!")
        if not mean_norm and norm_type in ("global", "group"):
            raise ValueError("mean_norm=False with norm_type 'global' / 'group' raises inside the reference "
                             "(feat_norm.py:475 reads curr_aver_mean.device of None); it is not supported here either.")
        self.norm_type, self.mean_norm, self.std_norm = norm_type, mean_norm, std_norm
        self.clamp, self.max_epoch_num = clamp, max_epoch_num
        if self.input_size is not None:
            self.output_size = self.input_size

    # ---- running statistics (feat_norm.py:452-531) --------------------------------------------------------------------------
    def register_mean_std_batch(self, curr_aver_mean, curr_aver_std, prefix, epoch):
        if epoch is None or not hasattr(self, f"{prefix}_batch"):
            self.register_buffer(f"{prefix}_batch", torch.ones(1, dtype=torch.int64, device=curr_aver_mean.device))
        elif epoch <= self.max_epoch_num:
            self.register_buffer(f"{prefix}_batch", self.get_buffer(f"{prefix}_batch") + 1)
        if epoch is None or epoch <= self.max_epoch_num:
            curr_weight = 1 / self.get_buffer(f"{prefix}_batch")
            if self.mean_norm:
                if not hasattr(self, f"{prefix}_mean"):
                    self.register_buffer(f"{prefix}_mean", curr_aver_mean)
                else:
                    self.register_buffer(f"{prefix}_mean", curr_weight * curr_aver_mean + (1 - curr_weight) * self.get_buffer(f"{prefix}_mean"))
            if self.std_norm:
                if not hasattr(self, f"{prefix}_std"):
                    self.register_buffer(f"{prefix}_std", curr_aver_std)
                else:
                    self.register_buffer(f"{prefix}_std", curr_weight * curr_aver_std + (1 - curr_weight) * self.get_buffer(f"{prefix}_std"))

    def update_aver_mean_std(self, epoch):
        if epoch is None or epoch <= self.max_epoch_num:
            n_mean = n_std = 0
            aver_mean = aver_std = None
            for name, buff in self.named_buffers():
                if name.endswith("_mean"):
                    n_mean += 1
                    aver_mean = buff.clone() if aver_mean is None else aver_mean + buff
                elif name.endswith("_std"):
                    n_std += 1
                    aver_std = buff.clone() if aver_std is None else aver_std + buff
            self.register_buffer("aver_mean", aver_mean / n_mean)
            self.register_buffer("aver_std", aver_std / n_std)

    # ---- forward (feat_norm.py:78-338) ----------------------------------------------------------------------------------------
    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor, group_ids=None, epoch: int = None):
        if self.norm_type == "group":
            assert group_ids is not None, ("You are using group-level feature normalization, but group_ids is not given. "
                                           "Please check 'data_cfg' in your configuration.")
        squeeze_flag = False
        if feat.dim() == 2:
            feat, squeeze_flag = feat.unsqueeze(-1), True
        elif feat.dim() != 3:
            raise RuntimeError(f"{self.__class__.__name__} only accepts the input vectors in the shape of "
                               f"(batch, length, channel) or (batch, length), but got shape={feat.shape}!")
        require_gpu(feat, "FeatureNormalization")
        if feat.requires_grad:
            raise NotImplementedError("speechain_amd FeatureNormalization does not differentiate through its statistics; it "
                                      "normalises frontend features, which carry no gradient.")
        x = feat if feat.dtype == torch.float32 and feat.stride(2) == 1 else feat.float().contiguous()
        batch_size = x.size(0)
        curr_means, curr_stds = utterance_stats(x, feat_len, self.clamp, self.mean_norm, self.std_norm)

        if self.norm_type == "utterance":
            out = apply_stats(x, curr_means, curr_stds)
        elif self.norm_type == "group" and isinstance(group_ids, torch.Tensor):
            out = self._forward_group_tensor(x, curr_means, curr_stds, group_ids, batch_size, epoch)
        else:
            batch_mean = batch_std = None
            if self.training:
                if self.distributed:
                    batch_mean, batch_std = ddp_batch_statistics(curr_means, curr_stds, batch_size)
                else:
                    batch_mean = curr_means.mean(dim=0) if curr_means is not None else None
                    batch_std = curr_stds.mean(dim=0) if curr_stds is not None else None
            if self.norm_type == "batch":
                out = apply_stats(x, batch_mean, batch_std) if self.training else apply_stats(x, curr_means, curr_stds)
            else:
                if self.norm_type == "group":
                    assert isinstance(group_ids, (str, int)), (
                        "If all the utterances in a single batch belong to the same group, you should give group_ids as a string "
                        f"or integer. But got type(group_ids)={type(group_ids)}.")
                prefix = "global" if self.norm_type == "global" else group_ids
                if self.training:
                    self.register_mean_std_batch(batch_mean, batch_std, prefix, epoch)
                out = apply_stats(x, self.get_buffer(f"{prefix}_mean") if curr_means is not None else None,
                                  self.get_buffer(f"{prefix}_std") if curr_stds is not None else None)
        return (out.squeeze(-1) if squeeze_flag else out), feat_len

    def _forward_group_tensor(self, x, curr_means, curr_stds, group_ids, batch_size, epoch):
        """Per-utterance group ids (feat_norm.py:148-230): ids select buffers by name, so they are read on the host once."""
        ids = [int(g) for g in group_ids.tolist()]
        if self.training:
            all_ids, all_means, all_stds = ids, curr_means, curr_stds
            if self.distributed:
                all_ids, all_means, all_stds = self._gather_groups(group_ids, curr_means, curr_stds, batch_size)
            order = []
            for g in all_ids:
                if g not in order:
                    order.append(g)
            id_t = torch.tensor(all_ids, device=x.device)
            for g in order:
                sel = id_t == g
                self.register_mean_std_batch(all_means[sel].mean(dim=0) if all_means is not None else None,
                                             all_stds[sel].mean(dim=0) if all_stds is not None else None, g, epoch)
            self.update_aver_mean_std(epoch)
        mean = std = None
        if self.mean_norm:
            mean = torch.stack([self.get_buffer(f"{g}_mean") if hasattr(self, f"{g}_mean") else self.get_buffer("aver_mean") for g in ids])
        if self.std_norm:
            std = torch.stack([self.get_buffer(f"{g}_std") if hasattr(self, f"{g}_std") else self.get_buffer("aver_std") for g in ids])
        return apply_stats(x, mean, std)

    @staticmethod
    def _gather_groups(group_ids, curr_means, curr_stds, batch_size):
        """Ids and per-utterance vectors of every rank, ragged batches padded to the largest (feat_norm.py:162-176, 353-424)."""
        import torch.distributed as dist

        world, dev = dist.get_world_size(), group_ids.device
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([batch_size], dtype=torch.int64, device=dev))
        sizes = [int(s) for s in torch.cat(sizes).tolist()]
        mx = max(sizes)

        def gather(t):
            if t is None:
                return None
            pad = torch.zeros((mx - t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            mine = torch.cat([t, pad])
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            return torch.cat([p[:n] for p, n in zip(parts, sizes)])

        return [int(g) for g in gather(group_ids.to(torch.int64)).tolist()], gather(curr_means), gather(curr_stds)

    # ---- recover (feat_norm.py:533-604) -----------------------------------------------------------------------------------------
    def recover(self, feat: torch.Tensor, group_ids=None):
        assert self.norm_type not in ["utterance", "batch"], \
            "If norm_type is either 'utterance' or 'batch', the normalized features cannot be recovered."
        if self.norm_type == "global" or (self.norm_type == "group" and isinstance(group_ids, (str, int))):
            prefix = "global" if self.norm_type == "global" else str(group_ids)
            feat = feat * self.get_buffer(f"{prefix}_std") if self.std_norm else feat
            feat = feat + self.get_buffer(f"{prefix}_mean") if self.mean_norm else feat
        elif self.norm_type == "group" and isinstance(group_ids, torch.Tensor):
            ids = [int(g) for g in group_ids.tolist()]
            if self.std_norm:
                feat = feat * torch.stack([self.get_buffer(f"{g}_std") if hasattr(self, f"{g}_std") else self.get_buffer("aver_std")
                                           for g in ids], dim=0).unsqueeze(1)
            if self.mean_norm:
                feat = feat + torch.stack([self.get_buffer(f"{g}_mean") if hasattr(self, f"{g}_mean") else self.get_buffer("aver_mean")
                                           for g in ids], dim=0).unsqueeze(1)
        elif self.norm_type == "group" and group_ids is None:
            feat = feat * self.get_buffer("aver_std").expand(1, 1, -1) if self.std_norm else feat
            feat = feat + self.get_buffer("aver_mean").expand(1, 1, -1) if self.mean_norm else feat
        else:
            raise RuntimeError
        return feat

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Buffers named ``*_batch`` / ``*_std`` / ``*_mean`` are created from the checkpoint (feat_norm.py:606-632)."""
        for key in state_dict.keys():
            if key.startswith(prefix):
                input_name = key[len(prefix):].split(".", 1)[0]
                if "_" in input_name and input_name.split("_")[-1] in ["batch", "std", "mean"]:
                    self.register_buffer(input_name, state_dict[key])
                else:
                    unexpected_keys.append(key)

    def extra_repr(self) -> str:
        return f"norm_type={self.norm_type}, mean_norm={self.mean_norm}, std_norm={self.std_norm}"
