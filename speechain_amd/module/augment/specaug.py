"""Batch-level SpecAugment: host-side index draws (bit-identical to the reference's) + one HIP pass.

Drop-in for ``speechain.module.augment.specaug.SpecAugment`` (speechain/module/augment/specaug.py): same kwargs
(:19-32) and ``forward(feat, feat_len) -> (feat, feat_len)`` (:73-206).  The random integers are drawn with the very
same ``torch.randint`` calls in the very same order (warp centre and position on the default CPU generator,
:98-108; mask lengths and positions on ``feat.device``, :145-158 and :178-191), so under one seed the indices are
bit-identical to the reference's wherever it runs.  Warp + mask + fill are then one launch of ``sc_specaug_fwd``.

Difference kept on purpose: the input ``feat`` is not modified (the reference writes the warped rows back into its
argument, :123-124, before ``masked_fill`` makes the returned copy).

Host round trips.  The reference's draw order forces three device-to-host reads per call when everything lives on the GPU: the shortest
length (:87) and the largest drawn mask length of each kind (:155, :188: the next ``randint``'s bound).  A caller that has the lengths on
the host -- every data loader does -- may pass ``feat_len`` as a CPU tensor (it is returned as it came) and set ``host_draws = True`` on the
module: all six draws then use the default CPU generator (the stream the reference itself uses when its features are CPU tensors: the
golden fixtures), nothing is read back from the device, and the four index tensors travel in ONE pinned host-to-device copy.
"""
from typing import Dict, List, Optional, Union

import torch

from speechain_amd._lib import lib, ptr, require_gpu, stream
from speechain_amd.module.abs import Module


class SpecAugment(Module):
    def module_init(self, time_warp: bool = True, time_warp_window: int = 5, time_warp_mode: str = "bicubic",
                    freq_mask: bool = True, freq_mask_width: Union[int, List[int]] = 30, freq_mask_num: int = 2,
                    time_mask: bool = True, time_mask_width: Union[int, float, List] = 0.05, time_mask_num: int = 2,
                    time_mask_ratio: float = 1.0, feat_norm: bool = True):
        assert time_warp or freq_mask or time_mask, "You must specify at least one type of augmentation in SpecAugment!"
        if time_warp and time_warp_mode not in ("bicubic", "bilinear"):
            # (the reference hands the mode to F.interpolate(..., align_corners=True) on a 4-D tensor, which accepts exactly these two)
            raise ValueError(f"time_warp_mode must be 'bicubic' or 'bilinear' (F.interpolate with align_corners=True), got {time_warp_mode!r}")
        self.feat_dim = None
        if self.input_size is not None:
            self.feat_dim = self.input_size
            self.output_size = self.input_size
        self.time_warp, self.time_warp_window, self.time_warp_mode = time_warp, time_warp_window, time_warp_mode
        self.freq_mask = freq_mask
        if isinstance(freq_mask_width, int):
            freq_mask_width = [0, freq_mask_width]
        elif not isinstance(freq_mask_width, List):
            raise ValueError
        if self.feat_dim is not None:
            assert freq_mask_width[1] < self.feat_dim, (
                "The number of maximum frequency masking bins cannot be larger than the feature dimension! "
                f"freq_mask_width[1]={freq_mask_width[1]} and self.feat_dim={self.feat_dim}.")
        self.freq_mask_width, self.freq_mask_num = freq_mask_width, freq_mask_num
        self.time_mask = time_mask
        if isinstance(time_mask_width, (int, float)):
            time_mask_width = [0, time_mask_width]
        elif not isinstance(time_mask_width, List):
            raise ValueError
        self.time_mask_width, self.time_mask_num = time_mask_width, time_mask_num
        self.feat_norm = feat_norm
        self.host_draws = False  # (see the module docstring; an attribute, not a keyword: the constructor stays the reference's)

    def draw(self, feat_len: torch.Tensor, batch_size: int, feat_dim: int, device) -> Dict:
        """The random integers of one call, drawn exactly as specaug.py:87-191 does (host logic, no kernel)."""
        tmin = int(feat_len.min().item())  # (specaug.py:87 -- one of the three device-to-host reads the draw order forces)
        d = dict(time_minlen=tmin, warp_center=None, warp_pos=None, freq_len=None, freq_pos=None, time_len=None, time_pos=None)
        W = self.time_warp_window
        if self.time_warp and tmin > 2 * W + 1:
            d["warp_center"] = torch.randint(low=W + 1, high=tmin - W, size=(1,))[0].item()
            d["warp_pos"] = torch.randint(low=d["warp_center"] - W, high=d["warp_center"] + W, size=(1,))[0].item()
        if self.freq_mask:
            if self.feat_dim is None:
                assert self.freq_mask_width[1] < feat_dim, (
                    "The number of maximum frequency masking bins cannot be larger than the feature dimension! "
                    f"self.freq_mask_width[1]={self.freq_mask_width[1]} and feat_dim={feat_dim}.")
            d["freq_len"] = torch.randint(self.freq_mask_width[0], self.freq_mask_width[1] + 1,
                                          size=(batch_size, self.freq_mask_num), device=device)
            d["freq_pos"] = torch.randint(0, max(1, feat_dim - d["freq_len"].max().item()),
                                          size=(batch_size, self.freq_mask_num), device=device)
        if self.time_mask:
            lo, hi = self.time_mask_width
            if isinstance(lo, float):
                lo = int(lo * tmin)
            if isinstance(hi, float):
                hi = int(hi * tmin)
            hi = min(hi, tmin)
            d["time_len"] = torch.randint(lo, hi + 1, size=(batch_size, self.time_mask_num), device=device)
            d["time_pos"] = torch.randint(0, max(1, tmin - d["time_len"].max().item()),
                                          size=(batch_size, self.time_mask_num), device=device)
        return d

    def _staging(self, n: int):
        """(pinned int64 buffer of >= n elements, its event) from a ring of eight: the host runs ahead of the device, so a buffer is reused
        only once the copy that read it last has completed."""
        ring = self.__dict__.get("_ring")
        if ring is None or ring[0][0].numel() < n:
            ring = [(torch.empty(max(n, 256), dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self.__dict__["_ring"], self.__dict__["_ring_at"] = ring, 0
        at = self.__dict__["_ring_at"] = (self.__dict__["_ring_at"] + 1) % len(ring)
        pin, ev = ring[at]
        if not ev.query():  # (True for an event never recorded or long complete)
            ev.synchronize()
        return pin, ev

    def apply(self, feat: torch.Tensor, draws: Dict) -> torch.Tensor:
        """Warp + mask + fill on the device for given indices (``draws`` as returned by :meth:`draw`)."""
        require_gpu(feat, "SpecAugment")
        x = feat.float().contiguous()
        B, T, Fd = x.shape
        dev = x.device
        idx = [draws["freq_len"], draws["freq_pos"], draws["time_len"], draws["time_pos"]]
        host = [t for t in idx if t is not None and not t.is_cuda]
        if len(host) > 1:
            # host-side draws: the four tensors go into one pinned staging buffer (from a ring: the kernels run later) which the kernels read
            # in place -- pinned host memory is device-addressable, and ~1 KB read once per workgroup costs less than an asynchronous copy call
            n = sum(t.numel() for t in host)
            pin, done = self._staging(n)
            torch.cat([t.reshape(-1) if t.dtype == torch.int64 else t.reshape(-1).to(torch.int64) for t in host], out=pin[:n])
            base, off, staged = pin.data_ptr(), 0, []
            for t in idx:
                if t is not None and not t.is_cuda:
                    staged.append((base + 8 * off, t.size(1)))
                    off += t.numel()
                else:
                    staged.append(None)
        else:
            staged, done = [None] * 4, None
        i64 = lambda t: t if t is None or (t.device == dev and t.dtype == torch.int64 and t.is_contiguous()) else t.to(device=dev, dtype=torch.int64).contiguous()
        keep = [None if st is not None else i64(t) for t, st in zip(idx, staged)]  # (device-side draws: the tensors themselves)
        addr = [st[0] if st is not None else ptr(k) for k, st in zip(keep, staged)]
        width = [st[1] if st is not None else (0 if k is None else k.size(1)) for k, st in zip(keep, staged)]
        fl, fp, tl, tp = addr
        nf, nt = width[0], width[2]
        c = -1 if draws["warp_center"] is None else int(draws["warp_center"])
        p = -1 if draws["warp_pos"] is None else int(draws["warp_pos"])
        out = torch.empty_like(x)
        wm = 1 if self.time_warp_mode == "bilinear" else 0
        tmin = int(draws["time_minlen"])
        masked = nf > 0 or nt > 0
        if masked and not self.feat_norm:
            # fill value = mean of the (warped) features (specaug.py:203): warp + per-workgroup sums, then masks in place (one call)
            ws = self.__dict__.get("_meanfill_ws")
            if ws is None or ws[0] != (B, T, dev):
                ws = ((B, T, dev), torch.empty(lib.value("sc_specaug_meanfill_workspace", B, T), dtype=torch.uint8, device=dev))
                self.__dict__["_meanfill_ws"] = ws  # (reused: the calls of one module are ordered on its stream)
            lib.call("sc_specaug_fwd_meanfill", ptr(x), ptr(out), B, T, Fd, tmin, c, p, wm, fl, fp, nf, tl, tp, nt, ptr(ws[1]), ws[1].numel(), stream())
        else:
            lib.call("sc_specaug_fwd", ptr(x), ptr(out), B, T, Fd, tmin, c, p, wm, fl, fp, nf, tl, tp, nt, None, stream())
        if done is not None:
            done.record()  # (the staging buffer is free once these kernels have run)
        return out

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        require_gpu(feat, "SpecAugment")
        draws = self.draw(feat_len, feat.size(0), feat.size(-1), torch.device("cpu") if self.host_draws else feat.device)
        return self.apply(feat, draws), feat_len

    def extra_repr(self) -> str:
        out = ""
        if self.time_warp:
            out += f"time_warp_window={self.time_warp_window}, time_warp_mode={self.time_warp_mode}, "
        if self.freq_mask:
            out += f"\nfreq_mask_width={self.freq_mask_width}, freq_mask_num={self.freq_mask_num}, "
        if self.time_mask:
            out += f"\ntime_mask_width={self.time_mask_width}, time_mask_num={self.time_mask_num}"
        return out
