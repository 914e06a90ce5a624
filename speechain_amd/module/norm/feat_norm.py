"""``speechain.module.norm.feat_norm.FeatureNormalization`` on gfx950.

Drop-in for speechain/module/norm/feat_norm.py: same kwargs (:53-77), the same lazily created buffers ``<prefix>_batch`` /
``<prefix>_mean`` / ``<prefix>_std`` / ``aver_mean`` / ``aver_std`` (re-created on every update, :472-509), checkpoints that bring
their own buffers (:606-632), ``forward(feat, feat_len, group_ids=None, epoch=None)`` and ``recover``.  The module is organised
differently from the reference: two kernels for the tensor work, a DDP helper, and a small "statistics book" (_stat / _put /
_fold / _refresh_average / _vectors_for) for the running statistics.

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
    on the tensors' own device (RCCL on the GPU, gloo in the CPU test).

    The reference gathers the batch sizes, the means and the stds in three collectives; here ONE all-gather carries
    [batch size | sum of means | sum of stds] per rank (the sizes travel as float32, exact below 2^24): this sits on the host-bound
    head of the training step (between the frontend and SpecAugment's host-side draws), where every collective costs its launch latency
    in full.  The size is written by a device-side fill, not copied from the host (a pageable host-to-device copy waits for the stream)."""
    import torch.distributed as dist

    ref = curr_means if curr_means is not None else curr_stds
    world = dist.get_world_size(group)
    feats = ref.size(-1)
    parts = [torch.full((1,), float(batch_size), dtype=torch.float32, device=ref.device)]
    parts += [v.sum(dim=0).float() for v in (curr_means, curr_stds) if v is not None]
    mine = torch.cat(parts)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    tot = torch.stack(gathered).sum(dim=0)
    total = tot[0]
    out, off = [], 1
    for v in (curr_means, curr_stds):
        if v is None:
            out.append(None)
        else:
            out.append((tot[off: off + feats] / total).to(v.dtype))
            off += feats
    return out[0], out[1]


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

    # ---- the statistics book: every statistic is a buffer named <key>_mean / <key>_std / <key>_batch -------------------------------
    # (the names and update rules are the checkpoint contract of feat_norm.py:452-531; the bookkeeping itself is organised here
    #  around three small primitives: _stat / _put / _fold)
    def _stat(self, key, kind: str) -> Optional[torch.Tensor]:
        return self._buffers.get(f"{key}_{kind}")

    def _put(self, key, kind: str, value: torch.Tensor):
        self.register_buffer(f"{key}_{kind}", value)  # a fresh tensor each time, as the reference re-registers its buffers

    def _fold(self, key, new_mean, new_std, epoch):
        """One more batch for `key`: the stored vectors become the average over the batches seen so far.  While `epoch` is not
        given the count restarts at one (the stored vectors then equal the newest batch); past `max_epoch_num` nothing moves."""
        frozen = epoch is not None and epoch > self.max_epoch_num
        seen = self._stat(key, "batch")
        if epoch is None or seen is None:
            anchor = new_mean if new_mean is not None else new_std
            self._put(key, "batch", torch.ones(1, dtype=torch.int64, device=anchor.device))
        elif not frozen:
            self._put(key, "batch", seen + 1)
        if frozen:
            return
        share = 1 / self._stat(key, "batch")
        for kind, fresh, wanted in (("mean", new_mean, self.mean_norm), ("std", new_std, self.std_norm)):
            if not wanted:
                continue
            old = self._stat(key, kind)
            self._put(key, kind, fresh if old is None else share * fresh + (1 - share) * old)

    def _refresh_average(self, epoch):
        """aver_mean / aver_std: the plain average of every stored *_mean / *_std vector -- the previous average included, which is
        what iterating over all buffers by suffix amounts to in the reference (feat_norm.py:519-528)."""
        if epoch is not None and epoch > self.max_epoch_num:
            return
        for kind in ("mean", "std"):
            pool = [v for name, v in self.named_buffers() if name.endswith("_" + kind)]
            total = pool[0].clone()
            for v in pool[1:]:
                total = total + v
            self.register_buffer("aver_" + kind, total / len(pool))

    def _vectors_for(self, ids, kind: str) -> torch.Tensor:
        """(len(ids), F): the stored vector of every id, the average one for ids never seen."""
        return torch.stack([self._stat(g, kind) if self._stat(g, kind) is not None else self.get_buffer("aver_" + kind) for g in ids])

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
                    self._fold(prefix, batch_mean, batch_std, epoch)
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
                self._fold(g, all_means[sel].mean(dim=0) if all_means is not None else None,
                           all_stds[sel].mean(dim=0) if all_stds is not None else None, epoch)
            self._refresh_average(epoch)
        return apply_stats(x, self._vectors_for(ids, "mean") if self.mean_norm else None,
                           self._vectors_for(ids, "std") if self.std_norm else None)

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

    # ---- recover (feat_norm.py:533-604): the inverse map x * std + mean with the statistics the forward pass would pick ---------
    def recover(self, feat: torch.Tensor, group_ids=None):
        assert self.norm_type not in ["utterance", "batch"], \
            "If norm_type is either 'utterance' or 'batch', the normalized features cannot be recovered."
        whole_batch = self.norm_type == "global" or isinstance(group_ids, (str, int))
        if whole_batch:
            key = "global" if self.norm_type == "global" else str(group_ids)
            pick = lambda kind: self.get_buffer(f"{key}_{kind}")
        elif isinstance(group_ids, torch.Tensor):
            ids = [int(g) for g in group_ids.tolist()]
            pick = lambda kind: self._vectors_for(ids, kind).unsqueeze(1)
        elif group_ids is None:
            pick = lambda kind: self.get_buffer("aver_" + kind).expand(1, 1, -1)
        else:
            raise RuntimeError(f"group_ids of type {type(group_ids)} cannot select statistics")
        if self.std_norm:
            feat = feat * pick("std")
        if self.mean_norm:
            feat = feat + pick("mean")
        return feat

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """The statistics exist only once observed, so a checkpoint brings its own buffers: every entry of this module whose name
        ends in _batch / _std / _mean becomes a buffer; anything else is reported as unexpected (behaviour of feat_norm.py:606-632)."""
        import re

        ours = re.compile(r"^[^.]*_(?:batch|std|mean)$")
        for key, value in state_dict.items():
            if not key.startswith(prefix):
                continue
            name = key[len(prefix):].split(".", 1)[0]
            if ours.match(name):
                self.register_buffer(name, value)
            else:
                unexpected_keys.append(key)

    def extra_repr(self) -> str:
        return f"norm_type={self.norm_type}, mean_norm={self.mean_norm}, std_norm={self.std_norm}"
