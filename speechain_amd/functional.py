"""Autograd-aware wrappers over the C ABI (include/speechain_amd.h) -- the ONLY place where kernels are launched for
the transformer path.  Every function here requires device tensors and calls the HIP library; there is no torch
fallback.  Activations are float32 (exact parity mode) or bfloat16 (training mode); parameters stay float32 masters
and gradients of parameters are produced in float32.

Stateless dropout: each dropout site draws (seed, offset) from :class:`DropoutState`; the backward pass regenerates the
identical keep-mask from the saved pair, so no mask tensor is stored.
"""
import math
import struct
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from speechain_amd._lib import lib, require_gpu, stream

ACT = {None: 0, "none": 0, "ReLU": 1, "GELU": 2}
F32, BF16 = 0, 1


def dcode(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"speechain_amd kernels take float32 or bfloat16 activations, got {t.dtype}")


def _p(t: Optional[torch.Tensor], elem_offset: int = 0):
    if t is None:
        return None
    return t.data_ptr() + elem_offset * t.element_size()


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class DropoutState:
    """Counter-based generator state.  The seed lives in a DEVICE cell (one per GPU) that kernels dereference, so a
    captured hipGraph draws fresh masks on every replay after :meth:`advance`; offsets are handed out from a host counter
    that ``advance`` resets (every step then uses the same offsets under a new seed).  Without ``advance`` the offsets
    simply keep growing, which is just as valid."""

    base_seed = None
    offset = 0
    cells = {}
    advances = 0  # advance() calls since the seed was set: a cell created later starts from the same value as the existing ones

    @classmethod
    def _cell(cls, device: torch.device) -> torch.Tensor:
        key = device.index
        if key not in cls.cells:
            if cls.base_seed is None:
                cls.base_seed = torch.initial_seed() & 0x3FFFFFFFFFFFFFFF
            v = (cls.base_seed + cls.advances * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            cls.cells[key] = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v], dtype=torch.int64, device=device)
        return cls.cells[key]

    @classmethod
    def next(cls, numel: int, device: torch.device = None) -> Tuple[int, int]:
        """(address of the device seed cell, offset) for a dropout site covering `numel` elements."""
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        off = cls.offset
        cls.offset += (int(numel) + 1) & ~1  # offsets stay even: masks are generated for pairs of elements
        return cls._cell(device).data_ptr(), off

    @classmethod
    def advance(cls):
        """New seed for the next step (a device-side add: capturable), offsets restart from zero."""
        for c in cls.cells.values():
            c.add_(0x9E3779B97F4A7C15 - (1 << 64))  # odd constant: the int64 cell walks all 2^64 values
        cls.advances += 1
        cls.offset = 0
        _MaskedGrad.sites.clear()  # (sites noted in the previous step whose output never reached a LayerNorm)

    @classmethod
    def reset(cls, seed: int):
        cls.base_seed, cls.offset, cls.advances = seed & 0x3FFFFFFFFFFFFFFF, 0, 0
        for c in cls.cells.values():
            c.fill_(cls.base_seed)


# ---------------------------------------------------------------------------------------------------------------------
# compute dtype policy and cached low-precision copies of the float32 master weights
# ---------------------------------------------------------------------------------------------------------------------

_compute_dtype = None  # None: follow torch autocast (bf16 when autocast is on, else f32)


def set_compute_dtype(dtype: Optional[torch.dtype]):
    """Force the activation dtype of the speechain_amd modules (torch.float32 / torch.bfloat16); None = follow autocast."""
    global _compute_dtype
    assert dtype in (None, torch.float32, torch.bfloat16)
    _compute_dtype = dtype


def compute_dtype() -> torch.dtype:
    if _compute_dtype is not None:
        return _compute_dtype
    return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32


def to_compute(x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Differentiable cast of an activation to the compute dtype (a HIP cast kernel; identity when already there)."""
    dtype = dtype or compute_dtype()
    if x.dtype == dtype:
        return x
    return _Cast.apply(x, dtype)


def raw_cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    x = x.contiguous()
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    lib.call("sc_cast", dcode(x), BF16 if dtype == torch.bfloat16 else F32, _p(x), _p(out), x.numel(), stream())
    return out


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        require_gpu(x, "cast")
        ctx.in_dtype = x.dtype
        return raw_cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return raw_cast(g, ctx.in_dtype), None


class _WeightCache:
    """(fused) compute-dtype copies of parameters, refreshed when a parameter's version counter moves.

    An entry is current while every source parameter still has the (version counter, address) it was made from AND the cache's
    `epoch` has not moved.  An optimizer that writes parameters with a raw kernel (``speechain_amd.optim.FusedAdam``: torch's
    version counters do not see it) calls :meth:`mark_stale` after its update; :meth:`refresh` rebuilds EVERY entry -- bf16
    and float32 matrices, their transposes, and the concatenated bias vectors -- in place (same buffers, so a captured hipGraph
    that holds their addresses keeps working) in at most two launches and marks them current."""

    def __init__(self):
        self.store = {}
        self.epoch = 0
        self._desc = {}  # per output dtype: (signature, device descriptor table, records, tiles) of the last refresh

    def _ver(self, params):
        return (self.epoch,) + tuple((p._version, p.data_ptr()) for p in params)

    def mark_stale(self):
        """Every cached copy is rebuilt on its next use (or by the next :meth:`refresh`).  Host-side flag only."""
        self.epoch += 1

    def _lookup(self, key, params):
        hit = self.store.get(key)
        if hit is not None and not all(r() is p for r, p in zip(hit[2], params)):
            return None  # an id() was recycled by a different (newer) parameter object: the entry is dead
        return hit

    def _put(self, key, params, buf):
        if len(self.store) > 4096:  # entries of dead parameters (models created and dropped) are not worth keeping
            self.store = {k: v for k, v in self.store.items() if all(r() is not None for r in v[2])}
        self.store[key] = (self._ver(params), buf, tuple(weakref.ref(p) for p in params))

    @staticmethod
    def _matrix(p: torch.Tensor) -> torch.Tensor:
        """The float32 (rows, K) GEMM matrix of a parameter: itself, or -- for an nn.Conv1d weight (out, in, k) of a 'same' convolution
        evaluated over the k shifted copies of its input (shifted_cat) -- the (out, k * in) matrix whose column is tap * C_in + channel."""
        p = p.detach()
        if p.dim() == 3:
            return p.permute(0, 2, 1).reshape(p.size(0), -1)
        return p.contiguous()

    def fused(self, params: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
        """Rows of all `params` (each (n_i, K) float32, or a Conv1d weight (n_i, C, k) read as (n_i, k * C)) stacked into one
        (sum n_i, K) matrix of `dtype`."""
        if len(params) == 1 and params[0].dtype == dtype and params[0].is_contiguous() and params[0].dim() == 2:
            return params[0].detach()
        cacheable = all(p.is_leaf for p in params)  # derived tensors (e.g. reshaped conv weights) are cast on the fly
        key = (tuple(id(p) for p in params), dtype)
        hit = self._lookup(key, params) if cacheable else None
        if hit is not None and hit[0] == self._ver(params):
            return hit[1]
        K = params[0].numel() // params[0].size(0)
        rows = sum(p.size(0) for p in params)
        buf = hit[1] if hit is not None else torch.empty(rows, K, dtype=dtype, device=params[0].device)
        r = 0
        for p in params:
            src = self._matrix(p).contiguous()
            lib.call("sc_cast", F32, BF16 if dtype == torch.bfloat16 else F32, _p(src), _p(buf, r * K), src.numel(), stream())
            r += p.size(0)
        if cacheable:
            self._put(key, params, buf)
        return buf

    def fused_t(self, params: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
        """(K, sum n_i) TRANSPOSED compute-dtype copy of the stacked weights: lets the data-gradient product dx = dz W run in
        the fast k-contiguous ("NT") GEMM form instead of reading W with its contraction index strided."""
        cacheable = all(p.is_leaf for p in params)
        key = (tuple(id(p) for p in params), dtype, "T")
        hit = self._lookup(key, params) if cacheable else None
        if hit is not None and hit[0] == self._ver(params):
            return hit[1]
        K = params[0].numel() // params[0].size(0)
        rows = sum(p.size(0) for p in params)
        buf = hit[1] if hit is not None else torch.empty(K, rows, dtype=dtype, device=params[0].device)
        r = 0
        for p in params:
            src = self._matrix(p).float().contiguous()
            lib.call("sc_cast_transpose", BF16 if dtype == torch.bfloat16 else F32, _p(src), _p(buf, r), p.size(0), K, rows, stream())
            r += p.size(0)
        if cacheable:
            self._put(key, params, buf)
        return buf

    def conv_dgrad(self, w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """(C, k N) compute-dtype matrix [channel][j N + out] = w[out][channel][k - 1 - j] of an nn.Conv1d weight (N, C, k): the operand of the
        im2col-free convolution's data gradient (`_ConvLinear.backward`).  Kept current by `refresh()` like the other copies."""
        cacheable = w.is_leaf
        key = ((id(w),), dtype, "D")
        hit = self._lookup(key, [w]) if cacheable else None
        if hit is not None and hit[0] == self._ver([w]):
            return hit[1]
        N, C, k = w.shape
        src = w.detach().permute(1, 2, 0).flip(1).reshape(C, k * N).to(dtype)  # [c][j][o] = w[o][c][k - 1 - j]
        if hit is not None:
            buf = hit[1].copy_(src)  # in place: a captured graph may hold the buffer's address
        else:
            buf = src.contiguous()
        if cacheable:
            self._put(key, [w], buf)
        return buf

    def vector(self, params: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
        """float32 concatenation of bias vectors (None if there is no bias).  The biases of one fused call are all present or all
        absent (the drop-in modules never mix them; `linear` checks)."""
        if all(p is None for p in params):
            return None
        if any(p is None for p in params):
            raise ValueError("a fused linear call needs a bias on every stacked weight or on none")
        if len(params) == 1:
            return params[0].detach()
        cacheable = all(p.is_leaf for p in params)
        key = (tuple(id(p) for p in params), "vec")
        hit = self._lookup(key, params) if cacheable else None
        if hit is not None and hit[0] == self._ver(params):
            return hit[1]
        if hit is not None:  # rebuilt in place: a captured graph may hold the buffer's address
            buf = torch.cat([p.detach() for p in params], out=hit[1])
        else:
            buf = torch.cat([p.detach() for p in params])
        if cacheable:
            self._put(key, params, buf)
        return buf

    def ln_folded(self, ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]], ln: torch.nn.LayerNorm):
        """(w_scaled bf16 (N, K), colsum f32 (N), shift f32 (N)) for `sc_decode_linear_lnstats`: LN(x) W^T + b = rstd (x w_scaled^T - mean colsum)
        + shift with w_scaled = bf16(gamma . W), colsum = its row sums (of the ROUNDED weights: the subtraction then cancels exactly what the
        product accumulated), shift = W beta + b.  Made once per version of the parameters, rebuilt in place (a captured decoding graph holds
        the addresses); inference only."""
        params = list(ws) + [b for b in bs if b is not None] + [ln.weight, ln.bias]
        key = tuple(id(p) for p in params)
        store = self.__dict__.setdefault("ln_store", {})
        hit = store.get(key)
        if hit is not None and not all(r() is p for r, p in zip(hit[2], params)):
            hit = None
        ver = self._ver(params)
        if hit is not None and hit[0] == ver:
            return hit[1]
        new = self._ln_build(ws, bs, ln.weight, ln.bias, hit[1] if hit is not None else None)
        if len(store) > 1024:
            store.clear()
        none_ref = lambda: None
        store[key] = (ver, new, tuple(weakref.ref(p) for p in params),
                      (tuple(weakref.ref(w) for w in ws), tuple(weakref.ref(b) if b is not None else none_ref for b in bs)))
        return new

    def _ln_build(self, ws, bs, gamma, beta, into):
        with torch.no_grad():
            Wf = torch.cat([self._matrix(w).float() for w in ws])
            Wp = (Wf * gamma.detach().float()[None, :]).to(torch.bfloat16)
            colsum = Wp.float().sum(dim=1)
            shift = Wf @ beta.detach().float()
            if any(b is not None for b in bs):
                shift = shift + torch.cat([b.detach().float() if b is not None else torch.zeros(w.size(0), device=w.device) for w, b in zip(ws, bs)])
            new = (Wp.contiguous(), colsum.contiguous(), shift.contiguous())
            if into is not None:
                for dst, src in zip(into, new):
                    dst.copy_(src)
                new = into
        return new

    def sync(self):
        """Brings every cached copy whose parameters have changed since it was made up to date IN PLACE, without a lookup by the layers that
        use it: for code that replays captured launches (the beam search's per-position hipGraphs hold the copies' addresses and never
        run the host-side lookups again).  No launches when nothing changed."""
        live = [(k, v) for k, v in self.store.items() if all(r() is not None for r in v[2])]
        if any(v[0] != self._ver([r() for r in v[2]]) for _, v in live):
            self.refresh()
        for key, hit in list(self.__dict__.get("ln_store", {}).items()):
            params = [r() for r in hit[2]]
            if any(p is None for p in params):
                del self.ln_store[key]
                continue
            ver = self._ver(params)
            if hit[0] != ver:
                ws, bs = [r() for r in hit[3][0]], [r() for r in hit[3][1]]
                self._ln_build(ws, bs, params[-2], params[-1], hit[1])
                self.ln_store[key] = (ver,) + hit[1:]

    def refresh(self):
        """Rebuilds EVERY cached copy of live parameters in place -- one launch for the bf16 matrices (plain, fused, transposed), one
        for the float32 ones and the concatenated bias vectors -- and marks them current.  Call once per training step after the
        optimizer (or at the top of a captured step): the per-call lookups then all hit, replacing ~250 tiny launches."""
        live = [(k, v) for k, v in self.store.items() if all(r() is not None for r in v[2])]
        groups = {BF16: [(k, v) for k, v in live if k[1] == torch.bfloat16],
                  F32: [(k, v) for k, v in live if k[1] != torch.bfloat16]}
        for code, ents in groups.items():
            if not ents:
                continue
            esz = 2 if code == BF16 else 4
            sig = tuple((k, v[1].data_ptr()) + tuple(r().data_ptr() for r in v[2]) for k, v in ents)
            cur = self._desc.get(code)
            if cur is None or cur[0] != sig:
                recs, tile0 = [], 0
                bykey = dict(ents)
                for k, v in ents:
                    buf, params = v[1], [r() for r in v[2]]
                    transposed = len(k) == 3 and k[2] in ("T", "D")  # ("D": the transposed copy with its (tap, channel) rows re-ordered)
                    # a plain copy and the transposed copy of the same parameters share their records: the source is read once
                    if len(k) == 3 and k[2] == "T" and (k[0], k[1]) in bykey:
                        continue
                    partner = bykey.get((k[0], k[1], "T")) if (len(k) == 2 and k[1] != "vec") else None
                    r = 0
                    for p in params:
                        assert p.is_contiguous() and p.dtype == torch.float32
                        if k[1] == "vec":  # a bias: one row of p.numel() columns at element offset r of the concatenation
                            rows, cols = 1, p.numel()
                            rec = (p.data_ptr(), buf.data_ptr() + esz * r, 0, rows, cols, cols, 0, tile0, (cols + 63) // 64)
                            r += cols
                        else:
                            rows, cols = p.size(0), p.numel() // p.size(0)
                            if transposed:
                                ldt = rows if k[2] == "D" else buf.size(1)  # ("D" is stored as (C, k N): its (C k, N) rows are N long)
                                rec = (p.data_ptr(), 0, buf.data_ptr() + esz * r, rows, cols, 0, ldt, tile0, (cols + 63) // 64)
                            elif partner is not None:
                                bt = partner[1]
                                rec = (p.data_ptr(), buf.data_ptr() + esz * r * cols, bt.data_ptr() + esz * r, rows, cols, cols, bt.size(1), tile0,
                                       (cols + 63) // 64)
                            else:
                                rec = (p.data_ptr(), buf.data_ptr() + esz * r * cols, 0, rows, cols, cols, 0, tile0, (cols + 63) // 64)
                            r += rows
                        conv_k = p.size(2) if (p.dim() == 3 and k[1] != "vec") else 1
                        recs.append(struct.pack("<QQQiiqqiiii", *rec, conv_k, 1 if (len(k) == 3 and k[2] == "D") else 0))
                        tile0 += ((cols + 63) // 64) * ((rows + 63) // 64)
                dev = ents[0][1][1].device
                cur = (sig, torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev), len(recs), tile0)
                self._desc[code] = cur
            lib.call("sc_cast_weights", code, _p(cur[1]), cur[2], cur[3], stream())
        for k, v in live:
            self.store[k] = (self._ver([r() for r in v[2]]), v[1], v[2])

    def invalidate(self):
        """Drops every cached copy (buffers included: not for use between the capture and the replays of a hipGraph)."""
        self.store.clear()
        self.__dict__.get("ln_store", {}).clear()
        self._desc = {}


weights = _WeightCache()


# ---------------------------------------------------------------------------------------------------------------------
# raw GEMM launcher
# ---------------------------------------------------------------------------------------------------------------------

def gemm(code: int, A, B, C, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, *, ta=0, tb=0, nb=(1, 1), sA=(0, 0),
         sB=(0, 0), sC=(0, 0), alpha=1.0, bias=None, act=0, pre_out=None, aux=None, ldaux=0, aux_mode=0, drop=(0.0, 0, 0),
         out_f32=False, accumulate=False):
    """A, B, C, pre_out, aux: raw device addresses (ints); see sc_gemm in include/speechain_amd.h."""
    lib.call("sc_gemm", code, ta, tb, A, B, C, M, N, K, lda, ldb, ldc, nb[0], nb[1], sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
             float(alpha), _p(bias), act, pre_out, aux, ldaux, aux_mode, float(drop[0]), int(drop[1]), int(drop[2]),
             1 if out_f32 else 0, 1 if accumulate else 0, stream())


class GradSinks:
    """Optional destination registry for parameter gradients.  A trainer (``speechain_amd.ddp.GradBucketReducer``) registers
    for every parameter a slot inside a flat float32 bucket; the backward kernels then write dW / db / dgamma straight into
    that slot and hand autograd a view of it, so no per-parameter ``grad += new`` kernel and no gradient copy is left.
    A parameter that receives a second contribution in the same step falls back to a fresh tensor (autograd adds it)."""

    table = {}
    used = {}
    step = 0

    @classmethod
    def register(cls, param: torch.Tensor, flat: torch.Tensor, offset: int):
        cls.table[id(param)] = (weakref.ref(param), flat, offset, param.numel())

    @classmethod
    def clear(cls):
        cls.table.clear()
        cls.used.clear()

    @classmethod
    def begin_step(cls):
        cls.step += 1
        _LnFinish.reset()  # (entries of a backward call that raised must not survive into the next step)
        _WgradQueue.reset()
        _MaskedGrad.reset()

    @classmethod
    def peek(cls, params: Sequence[torch.Tensor]) -> bool:
        """Would `take(params)` succeed now?  (Nothing is marked.)"""
        end = None
        for p in params:
            e = cls.table.get(id(p))
            if e is None or e[0]() is not p or cls.used.get(id(p)) == cls.step or (end is not None and (e[1] is not flat or e[2] != end)):
                return False
            flat, end = e[1], e[2] + e[3]
        return bool(params)

    @classmethod
    def take(cls, params: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
        """A flat float32 view covering the slots of `params` if they are registered, adjacent in this order and not yet
        written in this step; else None."""
        ents = []
        for p in params:
            e = cls.table.get(id(p))
            if e is None or e[0]() is not p or cls.used.get(id(p)) == cls.step:
                return None
            ents.append(e)
        flat, off = ents[0][1], ents[0][2]
        end = off
        for e in ents:
            if e[1] is not flat or e[2] != end:
                return None
            end += e[3]
        for p in params:
            cls.used[id(p)] = cls.step
        return flat[off:end]


class _Side:
    """Runs independent kernels of one backward step (e.g. wgrad next to dgrad) on a second stream and joins before the
    results are handed back to autograd.  The GEMMs here are latency-bound (~250 workgroups, 1 per CU), so two of them
    overlap almost perfectly.  Capture-safe (plain event fork/join)."""

    streams = {}
    enabled = True

    def __init__(self, ref: torch.Tensor, wgrad_only: bool = False, tag: str = "side"):
        """`wgrad_only`: the side work consists of weight-gradient products only -- no fork when those are deferred to the grouped
        launch anyway (gradient sinks registered, bf16).  `tag`: one stream per (device, tag); a forward branch (see `entry`) takes a
        tag of its own so that the fork/joins inside its own and the main path's backward nodes do not serialise the two."""
        self.on = _Side.enabled and ref.is_cuda
        if wgrad_only and _WgradQueue.enabled and GradSinks.table and ref.dtype == torch.bfloat16:
            self.on = False
        if self.on and tag == "side" and any(t != "side" and d == ref.device.index and s == torch.cuda.current_stream(ref.device)
                                             for (d, t), s in _Side.streams.items()):
            # Inside a forward branch (or its backward): the branch already runs next to the main path.  [A second-level fork that shares
            # the side stream with the main path's forks makes hipStreamEndCapture crash on ROCm 7.2 (tests/test_transformer_gpu.py:
            # test_hipgraph_replay_matches_eager_steps without gradient sinks).]
            self.on = False
        if self.on:
            key = (ref.device.index, tag)
            if key not in _Side.streams:
                _Side.streams[key] = torch.cuda.Stream(device=ref.device)
            self.side = _Side.streams[key]
            self.main = torch.cuda.current_stream(ref.device)

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)

    def entry(self, x: torch.Tensor) -> torch.Tensor:
        """Call inside the ``with`` block on the tensor a whole forward BRANCH starts from (e.g. the CTC head + loss next to the decoder):
        autograd runs a node's backward on the stream its forward ran on, so the branch's backward runs on the side stream too, next to
        the backward of the main path; the returned identity's backward is the branch's last node and joins the side stream back."""
        return _BranchEntry.apply(x, self.main) if (self.on and x.requires_grad) else x


class _BranchEntry(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, main):
        ctx.main = main
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # everything the branch's backward launched (gradient-slot writes and operands queued for the grouped weight-gradient launch
        # included) is ordered before whatever the main stream launches from here on
        ctx.main.wait_stream(torch.cuda.current_stream(g.device))
        return g, None


import os as _os
# Workgroups a split-K weight gradient aims for.  Measured on the bench step (scripts/ab_wgrad.sh, two alternating rounds on one box,
# 40 steps each): 128 -> 14.21 / 14.27 ms, 192 -> 13.69 / 13.83, 256 -> 13.46 / 13.55, 320 -> 13.55 / 13.56, 384 -> 13.74 / 13.65,
# 512 -> 14.03 / 13.84, 768 -> 14.42 / 14.12: one workgroup per CU -- more slices only add partial slabs for the reduce kernel to stream.
_WGRAD_WG_TARGET = int(_os.environ.get("SC_WGRAD_WG_TARGET", "256"))


class _WgradQueue:
    """Weight-gradient products whose outputs are gradient-bucket slots, deferred and computed TOGETHER by one grouped launch
    (``sc_wgrad_grouped``, csrc/wgrad_grouped.hip) when the running backward call finishes (autograd engine callback) or when the
    reducer is about to ship a bucket.  One layer's dW = dz^T x is a small output over a long contraction -- 4 to 64 tiles for 256
    CUs, which is why it used to be cut along the contraction (partial slabs + a second launch, 94 times per step); all layers of a
    backward pass together are ~1000 tiles of 256 x 256, each a full contraction.  Nothing reads a bucket slot before the step's
    backward has ended, so the result is the same.  Entries hold raw addresses of the outputs (a lingering reference to the slot
    views would make autograd's accumulation clone them before they are written) and keep the operands alive."""

    pending = []   # (dz address, x address, dW address, db address or 0, n_out, n_in, tokens, ldz, ldx, conv_k)
    keep = []      # operand tensors
    armed = False
    enabled = _os.environ.get("SC_WGRAD_GROUPED", "1") == "1"  # (A/B knob)

    @classmethod
    def add(cls, dz: torch.Tensor, x2: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor], conv_k: int = 1):
        cls.pending.append((dz.data_ptr(), x2.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, dz.size(1), x2.size(1),
                            dz.size(0), dz.stride(0), x2.stride(0), conv_k))
        cls.keep.append((dz, x2))
        if not cls.armed:
            cls.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(cls.flush)

    @classmethod
    def reset(cls):
        cls.pending.clear()
        cls.keep.clear()
        cls.armed = False

    @classmethod
    def flush(cls):
        """Launches the pending products on the current stream."""
        cls.armed = False
        if not cls.pending:
            return
        import ctypes

        n = len(cls.pending)
        col = lambda i, ct: (ct * n)(*[e[i] for e in cls.pending])
        if lib.timing is not None and "sc_wgrad_grouped" in lib.timing and lib.flop_log is not None:
            lib.flop_log.append(sum(2.0 * e[4] * e[5] * e[6] for e in cls.pending))
        n_out, n_in = col(4, ctypes.c_int), col(5, ctypes.c_int)
        ws = torch.empty(lib.value("sc_wgrad_grouped_workspace", n, n_out, n_in), dtype=torch.uint8, device=cls.keep[0][0].device)
        lib.call("sc_wgrad_grouped", n, col(0, ctypes.c_void_p), col(1, ctypes.c_void_p), col(2, ctypes.c_void_p), col(3, ctypes.c_void_p),
                 n_out, n_in, col(6, ctypes.c_int), col(7, ctypes.c_longlong), col(8, ctypes.c_longlong), col(9, ctypes.c_int), 0, ws.data_ptr(),
                 ws.numel(), stream())
        cls.pending.clear()
        cls.keep.clear()


def wgrad(code: int, dz: torch.Tensor, x2: torch.Tensor, params: Sequence[torch.Tensor] = (), bias: bool = False,
          bias_params: Sequence[torch.Tensor] = ()):
    """dW (N, K) float32 = dz^T x for dz (M, N), x (M, K), written into the registered gradient slots of `params` when they
    have adjacent ones.  The output is small and the contraction long, so in bf16 mode the contraction is split across
    workgroups (sc_gemm_splitk) to fill the chip.  With `bias` the bias gradient db (N,) = column sums of dz comes back
    too: the split-K kernel takes it from the dz tiles it streams anyway, otherwise a separate column-sum launch does."""
    M, N = dz.shape
    K = x2.size(1)
    # a Conv1d weight (N, C, k) behind a product over the k shifted copies of the input: the gradient is wanted in the weight's own layout
    conv_k = params[0].size(2) if (len(params) == 1 and params[0].dim() == 3) else 1
    sink = GradSinks.take(params) if params else None
    if (code == BF16 and sink is not None and _WgradQueue.enabled and dz.stride(1) == 1 and x2.stride(1) == 1
            and lib.value("sc_wgrad_grouped_supported", N, K, M, dz.stride(0), x2.stride(0))
            and dz.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and sink.data_ptr() % 16 == 0):
        bsink = GradSinks.take(bias_params) if (bias and bias_params) else None
        if not bias or bsink is not None:
            # both outputs are bucket slots: the product joins the grouped launch at the end of this backward call
            dW = sink.view(N, K)
            _WgradQueue.add(dz, x2, dW, bsink, conv_k)
            return (dW, bsink) if bias else dW
        dW = sink.view(N, K)  # (a bias outside the buckets: its column sums are taken now)
        _WgradQueue.add(dz, x2, dW, None, conv_k)
        return dW, _colsum(dz, ())
    if conv_k > 1:  # immediate path: the product in GEMM layout, then one permuting copy into the weight's layout
        g2 = wgrad(code, dz, x2, (), bias=bias, bias_params=bias_params)
        g2d, db = g2 if bias else (g2, None)
        dW = sink.view(N, K) if sink is not None else torch.empty(N, K, dtype=torch.float32, device=dz.device)
        dW.view(N, K // conv_k, conv_k).copy_(g2d.view(N, conv_k, K // conv_k).permute(0, 2, 1))
        return (dW, db) if bias else dW
    dW = sink.view(N, K) if sink is not None else torch.empty(N, K, dtype=torch.float32, device=dz.device)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    # (at most 16 slices for the transformer's products, M ~ 8000; up to 64 for the prenet's conv weight gradient, whose contraction
    # runs over all B*T'*F' = 151 392 output positions while its output is as small as 64 x 576)
    # (rounded, not floored: the vocabulary heads' 5000 x 512 gradients are 160 tiles -- one slice would send them to the generic kernel)
    splitk = min(16 if M < 32768 else 64, max(1, (_WGRAD_WG_TARGET + tiles // 2) // tiles), max(1, M // 256))
    db = None
    if code == BF16 and splitk >= 2:
        if bias:
            bsink = GradSinks.take(bias_params) if bias_params else None
            db = bsink if bsink is not None else torch.empty(N, dtype=torch.float32, device=dz.device)
        ws = torch.empty(lib.value("sc_gemm_splitk_workspace", N, K, splitk), dtype=torch.uint8, device=dz.device)
        lib.call("sc_gemm_splitk", 1, 1, _p(dz), _p(x2), _p(dW), N, K, M, dz.stride(0), x2.stride(0), 1.0, splitk, 0, _p(db),
                 _p(ws), ws.numel(), stream())
    else:
        gemm(code, _p(dz), _p(x2), _p(dW), N, K, M, dz.stride(0), x2.stride(0), K, ta=1, tb=1, out_f32=True)
        if bias:
            db = _colsum(dz, bias_params)
    return (dW, db) if bias else dW


def dgrad(code: int, dz: torch.Tensor, ws: Sequence[torch.Tensor], out: torch.Tensor, **epi):
    """out (M, K) = dz (M, N) @ W (N, K) [+ fused epilogue].  bf16: k-contiguous product against the cached transposed
    weights; f32 (parity mode): the exact-f32 kernel reads W directly with its contraction index strided."""
    M, N = dz.shape
    K = out.size(1)
    if code == BF16:
        Wt = weights.fused_t(ws, dz.dtype)
        gemm(code, _p(dz), _p(Wt), _p(out), M, K, N, dz.stride(0), N, K, **epi)
    else:
        W = weights.fused(ws, dz.dtype)
        gemm(code, _p(dz), _p(W), _p(out), M, K, N, dz.stride(0), K, K, tb=1, **epi)
    return out


def _colsum(x2d: torch.Tensor, params: Sequence[torch.Tensor] = ()) -> torch.Tensor:
    R, N = x2d.shape
    sink = GradSinks.take(params) if params else None
    out = sink if sink is not None else torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(lib.value("sc_colsum_workspace", R, N), dtype=torch.uint8, device=x2d.device)
    lib.call("sc_colsum", dcode(x2d), _p(x2d), R, N, x2d.stride(0), _p(out), _p(ws), ws.numel(), stream())
    return out


def _dropout_apply(x: torch.Tensor, p: float, seed: int, offset: int) -> torch.Tensor:
    if p <= 0.0:
        return x
    y = torch.empty_like(x)
    lib.call("sc_dropout", dcode(x), _p(x), _p(y), x.numel(), 1.0, float(p), seed, offset, stream())
    return y


class _MaskedGrad:
    """Hands the masked gradient of a dropout site from the LayerNorm backward that produced the gradient to the site's own backward.

    In a pre-LN stack y = x + dropout(sublayer(LN(x))) the gradient that arrives at a sublayer is what the NEXT LayerNorm's backward
    kernel wrote, and the sublayer's backward starts by masking it (dz = dy * keepmask / (1 - p)) before two GEMMs consume it -- a
    launch per sublayer that re-reads 8 MB to write 8 MB.  The LayerNorm backward kernel can write dz as a second output
    (``sc_layernorm_bwd_dz``) if it knows the site's (p, seed, offset).  The link is made through addresses, not tensor objects
    (views and autograd's own wrappers change those):
      forward:   the site notes (p, seed, offset) under the address of its output; a LayerNorm whose input has a noted address
                 remembers the site;
      backward:  that LayerNorm writes dz next to dx and files it under dx's address, HOLDING dx -- while it is held autograd cannot
                 add another gradient into that memory in place, so a gradient that arrives at the site with this address IS dx,
                 unmodified; the site takes dz if address, shape and (p, seed, offset) all agree, else it masks the gradient itself.
    Everything is dropped at the start of a step."""

    enabled = _os.environ.get("SC_FUSE_DROPOUT_BWD", "1") == "1"  # (A/B knob)
    sites = {}    # address of a site's output -> (p, seed, offset, numel)
    ready = {}    # address of dx -> (dx, dz, (p, seed, offset))
    armed = False
    hits = 0      # masked gradients handed over so far (tests)

    @classmethod
    def note_site(cls, y: torch.Tensor, p: float, seed: int, off: int):
        if cls.enabled and p > 0.0 and y.dtype == torch.bfloat16:
            if len(cls.sites) > 4096:
                cls.sites.clear()
            cls.sites[y.data_ptr()] = (p, seed, off, y.numel())

    @classmethod
    def site_of(cls, x2: torch.Tensor):
        e = cls.sites.pop(x2.data_ptr(), None) if cls.enabled else None
        return e[:3] if e is not None and e[3] == x2.numel() else None

    @classmethod
    def put(cls, dx: torch.Tensor, dz: torch.Tensor, site):
        cls.ready[dx.data_ptr()] = (dx, dz, site)
        if not cls.armed:  # whatever nobody took is released when this backward call ends
            cls.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(cls._end_of_backward)

    @classmethod
    def _end_of_backward(cls):
        cls.armed = False
        cls.ready.clear()

    @classmethod
    def take(cls, dy2: torch.Tensor, p: float, seed: int, off: int) -> Optional[torch.Tensor]:
        e = cls.ready.get(dy2.data_ptr())
        if e is None or e[2] != (p, seed, off) or e[0].shape != dy2.shape or e[0].dtype != dy2.dtype or not dy2.is_contiguous():
            return None
        del cls.ready[dy2.data_ptr()]
        cls.hits += 1
        return e[1]

    @classmethod
    def reset(cls):
        cls.sites.clear()
        cls.ready.clear()
        cls.armed = False


def _masked_grad(dy2: torch.Tensor, p: float, seed: int, off: int) -> torch.Tensor:
    """dz = dy * keepmask / (1 - p) of a dropout site: taken from the LayerNorm backward that already made it, or made here."""
    if p <= 0.0:
        return dy2
    dz = _MaskedGrad.take(dy2, p, seed, off)
    return dz if dz is not None else _dropout_apply(dy2, p, seed, off)


# ---------------------------------------------------------------------------------------------------------------------
# Linear:  y = dropout(x W^T + b) [+ residual]      (W may be several parameters stacked along the output dim)
# ---------------------------------------------------------------------------------------------------------------------

class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, drop_p, nw, *wb):
        ws, bs = list(wb[:nw]), list(wb[nw:])
        require_gpu(x, "linear")
        x2 = x.reshape(-1, x.size(-1))
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        code = dcode(x2)
        W = weights.fused(ws, x2.dtype)
        b = weights.vector(bs)
        N = W.size(0)
        y = torch.empty(M, N, dtype=x2.dtype, device=x2.device)
        seed, off = DropoutState.next(M * N, x2.device) if drop_p > 0 else (0, 0)
        res2 = None
        if residual is not None:
            res2 = residual.reshape(M, N)
            if not res2.is_contiguous():
                res2 = res2.contiguous()
        gemm(code, _p(x2), _p(W), _p(y), M, N, K, K, K, N, bias=b, aux=_p(res2), ldaux=N, aux_mode=1 if res2 is not None else 0,
             drop=(drop_p, seed, off))
        ctx.save_for_backward(x2, *ws)
        ctx.bias_params = [bb for bb in bs if bb is not None]
        ctx.meta = (x.shape, drop_p, seed, off, [w.size(0) for w in ws], [bb is not None for bb in bs], residual is not None)
        _MaskedGrad.note_site(y, drop_p, seed, off)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, *ws = ctx.saved_tensors
        xshape, drop_p, seed, off, rows, has_b, has_res = ctx.meta
        M, K = x2.shape
        N = sum(rows)
        code = dcode(x2)
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres = dy if has_res else None
        dz = _masked_grad(dy2, drop_p, seed, off)
        W = weights.fused(ws, x2.dtype)
        side = _Side(x2, wgrad_only=True)
        with side:  # parameter gradients next to the data gradient
            # dW = dz^T x, and db = column sums of dz out of the same launch
            dW, db = wgrad(code, dz, x2, ws, bias=True, bias_params=ctx.bias_params if all(has_b) else ()) if any(has_b) \
                else (wgrad(code, dz, x2, ws), None)
            dws = [g.view_as(w) for g, w in zip(dW.split(rows, dim=0), ws)]  # (a Conv1d weight gets its (out, in, k) shape back)
            dbs = [None] * len(rows)
            if any(has_b):
                dbs = [g if hb else None for g, hb in zip(db.split(rows), has_b)]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(code, dz, ws, torch.empty(M, K, dtype=x2.dtype, device=x2.device)).reshape(xshape)  # dx = dz W
        side.join()
        return (dx, dres, None, None, *dws, *dbs)


def linear(x: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]], drop_p: float = 0.0,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _Linear.apply(x, residual, float(drop_p), len(ws), *ws, *bs)


class _LinearAct(torch.autograd.Function):
    """h = dropout(act(x W^T + b)) with the pre-activation kept for the backward pass (conv-type FFN in-layer)."""

    @staticmethod
    def forward(ctx, x, w, b, act, drop_p):
        require_gpu(x, "linear")
        x2 = x.reshape(-1, x.size(-1))
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        code = dcode(x2)
        W = weights.fused([w], x2.dtype)
        N = W.size(0)
        pre = torch.empty(M, N, dtype=x2.dtype, device=x2.device)
        h = torch.empty(M, N, dtype=x2.dtype, device=x2.device)
        seed, off = DropoutState.next(M * N, x2.device) if drop_p > 0 else (0, 0)
        gemm(code, _p(x2), _p(W), _p(h), M, N, K, K, K, N, bias=b.detach() if b is not None else None, act=act, pre_out=_p(pre),
             drop=(drop_p, seed, off))
        ctx.save_for_backward(x2, pre, w)
        ctx.bias_params = [b] if b is not None else []
        ctx.meta = (x.shape, act, drop_p, seed, off, b is not None)
        return h.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dh):
        x2, pre, w = ctx.saved_tensors
        xshape, act, drop_p, seed, off, has_b = ctx.meta
        M, K = x2.shape
        N = pre.size(1)
        code = dcode(x2)
        dh2 = dh.reshape(M, N)
        if not dh2.is_contiguous():
            dh2 = dh2.contiguous()
        dpre = torch.empty_like(pre)
        lib.call("sc_act_bwd", code, _p(dh2), _p(pre), _p(dpre), dpre.numel(), act, float(drop_p), seed, off, stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(code, dpre, [w], torch.empty(M, K, dtype=x2.dtype, device=x2.device)).reshape(xshape)
        dW, db = wgrad(code, dpre, x2, [w], bias=True, bias_params=ctx.bias_params) if has_b else (wgrad(code, dpre, x2, [w]), None)
        return dx, dW.view_as(w), db, None, None


def linear_act(x, w, b, activation: str, drop_p: float = 0.0):
    if activation not in ACT:
        raise NotImplementedError(f"speechain_amd feed-forward implements ReLU and GELU, got {activation}")
    return _LinearAct.apply(x, w, b, ACT[activation], float(drop_p))


# ---------------------------------------------------------------------------------------------------------------------
# 'same' Conv1d as ONE product over overlapping rows of a zero-padded image (no im2col copy, no col2im):
#     y = dropout(act(conv1d(x, w) + b)) [+ residual]        x (B, T, C), w (N, C, k) -- an nn.Conv1d weight --, k odd
# ---------------------------------------------------------------------------------------------------------------------

_CONV_DIRECT = _os.environ.get("SC_CONV_DIRECT", "1") == "1"  # (A/B knob: 0 = shifted_cat (im2col) + linear everywhere)


def _pad_rows(x3: torch.Tensor, r: int) -> torch.Tensor:
    """(B, T, C) bf16 -> (rows, C) image: utterance b at rows b (T + 2 r) .. with r zero rows on either side, zero rows behind the last
    utterance (csrc/elementwise.hip:sc_pad_rows).  rows = B (T + 2 r) rounded up to 32 (the grouped weight-gradient launch contracts over
    whole 32-row tiles) + 32 (the last windows read up to 2 r rows further)."""
    B, T, C = x3.shape
    rows = (B * (T + 2 * r) + 31) // 32 * 32 + 32
    out = torch.empty(rows, C, dtype=x3.dtype, device=x3.device)
    lib.call("sc_pad_rows", _p(x3), B, T, C, r, _p(out), rows, stream())
    return out


def _padded_empty(B: int, T: int, C: int, r: int, like: torch.Tensor) -> torch.Tensor:
    """The image of `_pad_rows` with its zero rows written and its data rows left to a GEMM (output pointer r rows in, batch stride
    (T + 2 r) C)."""
    rows = (B * (T + 2 * r) + 31) // 32 * 32 + 32
    out = torch.empty(rows, C, dtype=like.dtype, device=like.device)
    lib.call("sc_zero_pad_rows", B, T, C, r, _p(out), rows, stream())
    return out


def conv_direct_ok(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> bool:
    """The im2col-free path serves the training configuration it was built for: bf16 compute on the GPU, channel counts in whole 16-byte
    chunks, odd taps, and the weight's (and bias') gradient written into bucket slots by the grouped launch (GradSinks registered) --
    everything else keeps shifted_cat + linear."""
    return (_CONV_DIRECT and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and w.dim() == 3 and w.size(2) > 1 and w.size(2) % 2 == 1
            and x.size(2) == w.size(1) and x.size(2) % 8 == 0 and w.size(0) % 8 == 0 and w.is_leaf and _WgradQueue.enabled
            and id(w) in GradSinks.table and (b is None or id(b) in GradSinks.table))


class _ConvLinear(torch.autograd.Function):
    """Row t of utterance b of the im2col matrix [x[t - r], .., x[t + r]] is the k C contiguous elements that start at padded row t of the
    image `_pad_rows` writes, so the GEMM reads it in place with leading dimension C (rows overlap), batched over the utterances; the
    output is the contiguous (B, T, N) tensor.  Backward: the same trick on the zero-padded gradient gives the data gradient as one
    product against the taps in reverse order (dx[t] = sum_j dz[t - r + j] W[k - 1 - j]) -- no (M, k C) column gradient, no col2im --
    and the weight gradient joins the grouped launch with both operands as overlapping-row views.  Arithmetic and dropout indices are
    those of `linear(shifted_cat(x, k), ...)`: the results are identical bit for bit."""

    @staticmethod
    def forward(ctx, x, residual, w, b, act, drop_p):
        require_gpu(x, "conv1d")
        x = x if x.is_contiguous() else x.contiguous()
        B, T, C = x.shape
        N, _, k = w.shape
        r = (k - 1) // 2
        Tp, M = T + 2 * r, B * T
        xp = _pad_rows(x, r)
        W = weights.fused([w], x.dtype)  # (N, k C), column = tap * C + channel
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        pre = torch.empty(M, N, dtype=x.dtype, device=x.device) if act else None
        seed, off = DropoutState.next(M * N, x.device) if drop_p > 0 else (0, 0)
        res2 = None
        if residual is not None:
            res2 = residual.reshape(M, N)
            res2 = res2 if res2.is_contiguous() else res2.contiguous()
        gemm(BF16, _p(xp), _p(W), _p(y), T, N, k * C, C, k * C, N, nb=(B, 1), sA=(Tp * C, 0), sC=(T * N, 0),
             bias=b.detach() if b is not None else None, act=act, pre_out=_p(pre), aux=_p(res2), ldaux=N,
             aux_mode=1 if res2 is not None else 0, drop=(drop_p, seed, off))
        ctx.save_for_backward(xp, pre, w)
        ctx.bias_params = [b] if b is not None else []
        ctx.meta = (B, T, C, N, k, act, drop_p, seed, off, residual is not None)
        if not act:
            _MaskedGrad.note_site(y, drop_p, seed, off)
        return y.view(B, T, N)

    @staticmethod
    def backward(ctx, dy):
        xp, pre, w = ctx.saved_tensors
        B, T, C, N, k, act, drop_p, seed, off, has_res = ctx.meta
        r = (k - 1) // 2
        Tp, M = T + 2 * r, B * T
        dy2 = dy.reshape(M, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        if act:
            dz = torch.empty_like(pre)
            lib.call("sc_act_bwd", BF16, _p(dy2), _p(pre), _p(dz), dz.numel(), act, float(drop_p), seed, off, stream())
        else:
            dz = _masked_grad(dy2, drop_p, seed, off)
        dzp = _pad_rows(dz.view(B, T, N), r)
        bp = ctx.bias_params
        dzv, x2v = _conv_views(xp, dzp, B, Tp, r, k * C, C, N, [w] + bp)
        dW, db = wgrad(BF16, dzv, x2v, [w], bias=True, bias_params=bp) if bp else (wgrad(BF16, dzv, x2v, [w]), None)
        dx = None
        if ctx.needs_input_grad[0]:
            D = weights.conv_dgrad(w, dz.dtype)  # (C, k N): [channel][j * N + out] = w[out][channel][k - 1 - j]
            dx = torch.empty(M, C, dtype=dz.dtype, device=dz.device)
            gemm(BF16, _p(dzp), _p(D), _p(dx), T, C, k * N, N, k * N, C, nb=(B, 1), sA=(Tp * N, 0), sC=(T * C, 0))
            dx = dx.view(B, T, C)
        return dx, (dy if has_res else None), dW.view_as(w), db, None, None


def _conv_views(img: torch.Tensor, gimg: torch.Tensor, B: int, Tp: int, r: int, kc: int, c: int, n: int, params: Sequence[torch.Tensor]):
    """Operands of the weight gradient over the padded images (see `_ConvLinear.backward`): window i of `img` (rows i .. i + k - 1) belongs to
    the output position whose gradient sits at row i + r of `gimg`; windows that straddle two utterances meet zero rows there."""
    tokens = (B * Tp + 31) // 32 * 32
    x2v = img.as_strided((tokens, kc), (c, 1))
    dzv = gimg.as_strided((tokens, n), (n, 1), r * n)
    if not all(GradSinks.peek([p]) for p in params):
        x2v = x2v.contiguous()  # (not the configuration this path is for: the immediate weight-gradient kernels want distinct rows)
    return dzv, x2v


class _ConvFFN(torch.autograd.Function):
    """The conv-type position-wise feed-forward (speechain/module/transformer/feed_forward.py:71-95: Conv1d -> activation -> dropout ->
    Conv1d -> dropout [+ residual]) on the im2col-free products of `_ConvLinear`, with what `_FFN` does for the linear type: the first
    product stores act'(pre) x dropout factor next to the hidden activation, and the second convolution's data-gradient product
    multiplies by it in its epilogue -- no separate activation-backward launch."""

    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, act, p_h, p_res):
        require_gpu(x, "conv feed-forward")
        x = x if x.is_contiguous() else x.contiguous()
        B, T, C = x.shape
        Fd, _, k1 = w1.shape
        C2, _, k2 = w2.shape
        r1, r2 = (k1 - 1) // 2, (k2 - 1) // 2
        M = B * T
        xp = _pad_rows(x, r1)
        W1, W2 = weights.fused([w1], x.dtype), weights.fused([w2], x.dtype)
        s1, o1 = DropoutState.next(M * Fd, x.device) if p_h > 0 else (0, 0)
        if r1 == r2:
            # the first product writes the hidden activation straight into the second convolution's padded image (and the factor into an
            # image of the same geometry: the epilogue addresses both through the output's batch stride) -- no pad pass over (M, Fd)
            hp = _padded_empty(B, T, Fd, r2, x)
            fac = torch.empty_like(hp)
            gemm(BF16, _p(xp), _p(W1), _p(hp, r2 * Fd), T, Fd, k1 * C, C, k1 * C, Fd, nb=(B, 1), sA=((T + 2 * r1) * C, 0),
                 sC=((T + 2 * r2) * Fd, 0), bias=b1.detach() if b1 is not None else None, act=act | 0x100, pre_out=_p(fac, r2 * Fd),
                 drop=(p_h, s1, o1))
        else:
            fac = torch.empty(M, Fd, dtype=x.dtype, device=x.device)
            h = torch.empty(M, Fd, dtype=x.dtype, device=x.device)
            gemm(BF16, _p(xp), _p(W1), _p(h), T, Fd, k1 * C, C, k1 * C, Fd, nb=(B, 1), sA=((T + 2 * r1) * C, 0), sC=(T * Fd, 0),
                 bias=b1.detach() if b1 is not None else None, act=act | 0x100, pre_out=_p(fac), drop=(p_h, s1, o1))
            hp = _pad_rows(h.view(B, T, Fd), r2)
            del h
        y = torch.empty(M, C2, dtype=x.dtype, device=x.device)
        s2, o2 = DropoutState.next(M * C2, x.device) if p_res > 0 else (0, 0)
        res2 = None
        if residual is not None:
            res2 = residual.reshape(M, C2)
            res2 = res2 if res2.is_contiguous() else res2.contiguous()
        gemm(BF16, _p(hp), _p(W2), _p(y), T, C2, k2 * Fd, Fd, k2 * Fd, C2, nb=(B, 1), sA=((T + 2 * r2) * Fd, 0), sC=(T * C2, 0),
             bias=b2.detach() if b2 is not None else None, aux=_p(res2), ldaux=C2, aux_mode=1 if res2 is not None else 0,
             drop=(p_res, s2, o2))
        ctx.save_for_backward(xp, fac, hp, w1, w2)
        ctx.bias_params = [[b1] if b1 is not None else [], [b2] if b2 is not None else []]
        ctx.meta = (B, T, C, Fd, C2, k1, k2, p_res, s2, o2, residual is not None)
        _MaskedGrad.note_site(y, p_res, s2, o2)
        return y.view(B, T, C2)

    @staticmethod
    def backward(ctx, dy):
        xp, fac, hp, w1, w2 = ctx.saved_tensors
        B, T, C, Fd, C2, k1, k2, p_res, s2, o2, has_res = ctx.meta
        r1, r2 = (k1 - 1) // 2, (k2 - 1) // 2
        M = B * T
        bp1, bp2 = ctx.bias_params
        dy2 = dy.reshape(M, C2)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dz2p = _pad_rows(_masked_grad(dy2, p_res, s2, o2).view(B, T, C2), r2)
        dzv, x2v = _conv_views(hp, dz2p, B, T + 2 * r2, r2, k2 * Fd, Fd, C2, [w2] + bp2)
        dW2, db2 = wgrad(BF16, dzv, x2v, [w2], bias=True, bias_params=bp2) if bp2 else (wgrad(BF16, dzv, x2v, [w2]), None)
        # dz1 = (second convolution's data gradient) * [act'(pre) * keepmask_h / (1 - p_h)]: the bracket is `fac`
        if r1 == r2:  # (fac is an image of the geometry dz1p gets: the product writes dz1p's data rows directly)
            dz1p = _padded_empty(B, T, Fd, r1, dy2)
            gemm(BF16, _p(dz2p), _p(weights.conv_dgrad(w2, dy2.dtype)), _p(dz1p, r1 * Fd), T, Fd, k2 * C2, C2, k2 * C2, Fd, nb=(B, 1),
                 sA=((T + 2 * r2) * C2, 0), sC=((T + 2 * r1) * Fd, 0), aux=_p(fac, r1 * Fd), ldaux=Fd, aux_mode=3)
        else:
            dz1 = torch.empty(M, Fd, dtype=dy2.dtype, device=dy2.device)
            gemm(BF16, _p(dz2p), _p(weights.conv_dgrad(w2, dy2.dtype)), _p(dz1), T, Fd, k2 * C2, C2, k2 * C2, Fd, nb=(B, 1),
                 sA=((T + 2 * r2) * C2, 0), sC=(T * Fd, 0), aux=_p(fac), ldaux=Fd, aux_mode=3)
            dz1p = _pad_rows(dz1.view(B, T, Fd), r1)
            del dz1
        dzv, x2v = _conv_views(xp, dz1p, B, T + 2 * r1, r1, k1 * C, C, Fd, [w1] + bp1)
        dW1, db1 = wgrad(BF16, dzv, x2v, [w1], bias=True, bias_params=bp1) if bp1 else (wgrad(BF16, dzv, x2v, [w1]), None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, C, dtype=dy2.dtype, device=dy2.device)
            gemm(BF16, _p(dz1p), _p(weights.conv_dgrad(w1, dy2.dtype)), _p(dx), T, C, k1 * Fd, Fd, k1 * Fd, C, nb=(B, 1),
                 sA=((T + 2 * r1) * Fd, 0), sC=(T * C, 0))
            dx = dx.view(B, T, C)
        return dx, (dy if has_res else None), dW1.view_as(w1), db1, dW2.view_as(w2), db2, None, None, None


def conv_ffn(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor], activation: str,
             p_hidden: float = 0.0, p_res: float = 0.0, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Conv1d -> activation -> dropout -> Conv1d -> dropout [+ residual] over (B, T, C) with nn.Conv1d weights: `_ConvFFN` when both
    convolutions qualify for the im2col-free path, else two `conv_linear` calls.  Same forward arithmetic either way."""
    if activation not in ACT or not ACT[activation]:
        raise NotImplementedError(f"speechain_amd conv feed-forward implements ReLU and GELU, got {activation}")
    if conv_direct_ok(x, w1, b1) and w2.dim() == 3 and w2.size(2) > 1 and w2.size(2) % 2 == 1 and w2.size(1) == w1.size(0) and \
            w2.size(0) % 8 == 0 and w2.is_leaf and id(w2) in GradSinks.table and (b2 is None or id(b2) in GradSinks.table):
        return _ConvFFN.apply(x, residual, w1, b1, w2, b2, ACT[activation], float(p_hidden), float(p_res))
    h = conv_linear(x, w1, b1, activation, p_hidden)
    return conv_linear(h, w2, b2, None, p_res, residual)


def conv_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], activation: Optional[str] = None, drop_p: float = 0.0,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """'same' Conv1d over (B, T, C) with an nn.Conv1d weight (N, C, k): the im2col-free path when `conv_direct_ok`, else
    `linear[_act](shifted_cat(x, k), ...)` -- the same arithmetic either way."""
    if activation not in ACT:
        raise NotImplementedError(f"speechain_amd conv feed-forward implements ReLU and GELU, got {activation}")
    if conv_direct_ok(x, w, b) and (residual is None or ACT[activation] == 0):
        return _ConvLinear.apply(x, residual, w, b, ACT[activation], float(drop_p))
    col = shifted_cat(x, w.size(2))
    if ACT[activation]:
        return linear_act(col, w, b, activation, drop_p)
    return linear(col, [w], [b], drop_p=drop_p, residual=residual)


# ---------------------------------------------------------------------------------------------------------------------
# Position-wise feed-forward:  y = dropout_res( dropout_h(act(x W1^T + b1)) W2^T + b2 ) [+ residual]
# ---------------------------------------------------------------------------------------------------------------------

_FFN_PREGRAD = _os.environ.get("SC_FFN_PREGRAD", "1") == "1"


class _FFN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, act, p_h, p_res):
        require_gpu(x, "feed-forward")
        x2 = x.reshape(-1, x.size(-1))
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        code = dcode(x2)
        W1, W2 = weights.fused([w1], x2.dtype), weights.fused([w2], x2.dtype)
        Fd = W1.size(0)
        h_pre = torch.empty(M, Fd, dtype=x2.dtype, device=x2.device)
        h = torch.empty(M, Fd, dtype=x2.dtype, device=x2.device)
        s1, o1 = DropoutState.next(M * Fd, x2.device) if p_h > 0 else (0, 0)
        # (act | 0x100: the second output is act'(pre) * dropout factor, the multiplier of this layer's data gradient -- see backward)
        gemm(code, _p(x2), _p(W1), _p(h), M, Fd, K, K, K, Fd, bias=b1.detach() if b1 is not None else None,
             act=(act | 0x100) if _FFN_PREGRAD else act, pre_out=_p(h_pre), drop=(p_h, s1, o1))
        y = torch.empty(M, K, dtype=x2.dtype, device=x2.device)
        s2, o2 = DropoutState.next(M * K, x2.device) if p_res > 0 else (0, 0)
        res2 = None
        if residual is not None:
            res2 = residual.reshape(M, K)
            if not res2.is_contiguous():
                res2 = res2.contiguous()
        gemm(code, _p(h), _p(W2), _p(y), M, K, Fd, Fd, Fd, K, bias=b2.detach() if b2 is not None else None, aux=_p(res2), ldaux=K,
             aux_mode=1 if res2 is not None else 0, drop=(p_res, s2, o2))
        ctx.save_for_backward(x2, h_pre, h, w1, w2)
        ctx.bias_params = [[b1] if b1 is not None else [], [b2] if b2 is not None else []]
        ctx.meta = (x.shape, act, p_h, s1, o1, p_res, s2, o2, b1 is not None, b2 is not None, residual is not None)
        _MaskedGrad.note_site(y, p_res, s2, o2)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, h_pre, h, w1, w2 = ctx.saved_tensors
        xshape, act, p_h, s1, o1, p_res, s2, o2, has_b1, has_b2, has_res = ctx.meta
        M, K = x2.shape
        Fd = h.size(1)
        code = dcode(x2)
        dy2 = dy.reshape(M, K)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dz = _masked_grad(dy2, p_res, s2, o2)
        W1, W2 = weights.fused([w1], x2.dtype), weights.fused([w2], x2.dtype)
        side = _Side(x2, wgrad_only=True)
        with side:
            dW2, db2 = wgrad(code, dz, h, [w2], bias=True, bias_params=ctx.bias_params[1]) if has_b2 \
                else (wgrad(code, dz, h, [w2]), None)
        # dpre = (dz W2) * [act'(pre) * keepmask_h / (1 - p_h)]: the bracket was stored by the forward product (h_pre), so the dgrad
        # epilogue is one multiply (it used to re-evaluate erf / exp and the dropout hash for every element)
        if _FFN_PREGRAD:
            dpre = dgrad(code, dz, [w2], torch.empty(M, Fd, dtype=x2.dtype, device=x2.device), aux=_p(h_pre), ldaux=Fd, aux_mode=3)
        else:  # (A/B knob SC_FFN_PREGRAD=0: h_pre holds the pre-activation, act' and the mask are re-evaluated here)
            dpre = dgrad(code, dz, [w2], torch.empty(M, Fd, dtype=x2.dtype, device=x2.device), act=act, aux=_p(h_pre), ldaux=Fd,
                         aux_mode=2, drop=(p_h, s1, o1))
        with side:
            dW1, db1 = wgrad(code, dpre, x2, [w1], bias=True, bias_params=ctx.bias_params[0]) if has_b1 \
                else (wgrad(code, dpre, x2, [w1]), None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(code, dpre, [w1], torch.empty(M, K, dtype=x2.dtype, device=x2.device)).reshape(xshape)
        side.join()
        return dx, (dy if has_res else None), dW1, db1, dW2, db2, None, None, None


def ffn(x, w1, b1, w2, b2, activation: str, p_hidden: float = 0.0, p_res: float = 0.0, residual=None):
    if activation not in ACT:
        raise NotImplementedError(f"speechain_amd feed-forward implements ReLU and GELU, got {activation}")
    return _FFN.apply(x, residual, w1, b1, w2, b2, ACT[activation], float(p_hidden), float(p_res))


# ---------------------------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------------------------

class _LnFinish:
    """Pending second stages of LayerNorm backward calls (dgamma / dbeta from the partial sums) whose outputs are gradient-bucket slots:
    nothing reads those before the step ends, so they are produced together by ONE launch when the running backward call finishes
    (autograd engine callback) -- 43 launches and as many dependency gaps fewer per training step.  Entries are raw addresses; only the
    workspaces are kept alive (a lingering reference to the gradient views would make autograd's accumulation clone them right away,
    before they are written)."""

    pending = []   # (workspace address, R, D, dgamma address, dbeta address)
    keep = []      # the workspace tensors
    armed = False
    enabled = _os.environ.get("SC_LN_BATCH_FINISH", "1") == "1"  # (A/B knob)

    @classmethod
    def add(cls, ws: torch.Tensor, R: int, D: int, dg: torch.Tensor, db: torch.Tensor):
        cls.pending.append((ws.data_ptr(), R, D, dg.data_ptr(), db.data_ptr()))
        cls.keep.append(ws)
        if not cls.armed:
            cls.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(cls.flush)

    @classmethod
    def reset(cls):
        cls.pending.clear()
        cls.keep.clear()
        cls.armed = False

    @classmethod
    def flush(cls):
        """Launches the pending reductions on the current stream (also called by GradBucketReducer before it ships a bucket)."""
        cls.armed = False
        if not cls.pending:
            return
        import ctypes

        n = len(cls.pending)
        col = lambda i, ct: (ct * n)(*[e[i] for e in cls.pending])
        lib.call("sc_layernorm_bwd_finish", n, col(0, ctypes.c_void_p), col(1, ctypes.c_longlong), col(2, ctypes.c_int),
                 col(3, ctypes.c_void_p), col(4, ctypes.c_void_p), stream())
        cls.pending.clear()
        cls.keep.clear()


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last dim.  With `with_residual` the input is also returned unchanged as a second output for the residual
    connection of a pre-LN block to use: the block input then reaches autograd through this function only, backward receives both of
    its gradients (through the norm and through the residual path) and the kernel adds them before the store -- instead of a
    separate elementwise sum launched by autograd for every sublayer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, with_residual):
        require_gpu(x, "layer norm")
        D = x.size(-1)
        x2 = x.reshape(-1, D)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        R = x2.size(0)
        y = torch.empty_like(x2)
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        lib.call("sc_layernorm_fwd", dcode(x2), _p(x2), _p(gamma.detach()), _p(beta.detach()), _p(y), _p(mean), _p(rstd), R, D,
                 float(eps), stream())
        ctx.save_for_backward(x2, mean, rstd, gamma)
        ctx.beta_param = beta
        ctx.xshape = x.shape
        ctx.with_residual = with_residual
        # the input is the output of a dropout site (out-projection / feed-forward with residual dropout): backward also writes that
        # site's masked gradient (see _MaskedGrad)
        ctx.site = _MaskedGrad.site_of(x2) if lib.value("sc_layernorm_bwd_dz_supported", dcode(x2), D) else None
        if with_residual:
            return y.reshape(x.shape), x.view_as(x)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dres=None):
        x2, mean, rstd, gamma = ctx.saved_tensors
        R, D = x2.shape
        if dy is None:  # the normalised output was not used: only the residual path carries a gradient
            return dres, None, None, None, None
        dy2 = dy.reshape(R, D)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dr2 = None
        if dres is not None:
            dr2 = dres.reshape(R, D).to(dy2.dtype)
            if not dr2.is_contiguous():
                dr2 = dr2.contiguous()
        dx = torch.empty_like(x2)
        sg, sb = GradSinks.take([gamma]), GradSinks.take([ctx.beta_param])
        dg = sg if sg is not None else torch.empty(D, dtype=torch.float32, device=x2.device)
        db = sb if sb is not None else torch.empty(D, dtype=torch.float32, device=x2.device)
        ws = torch.empty(lib.value("sc_layernorm_bwd_workspace", R, D), dtype=torch.uint8, device=x2.device)
        batched = _LnFinish.enabled and sg is not None and sb is not None  # both outputs are bucket slots: see _LnFinish
        site = ctx.site
        if site is not None and all(t is None or t.data_ptr() % 16 == 0 for t in (dy2, x2, dx, dr2)):
            dz = torch.empty_like(x2)
            lib.call("sc_layernorm_bwd_dz", dcode(x2), _p(dy2), _p(x2), _p(mean), _p(rstd), _p(gamma.detach()), _p(dx), _p(dg), _p(db), R, D,
                     _p(dr2), _p(ws), ws.numel(), 1 if batched else 3, _p(dz), float(site[0]), int(site[1]), int(site[2]), stream())
            _MaskedGrad.put(dx, dz, site)
        else:
            lib.call("sc_layernorm_bwd", dcode(x2), _p(dy2), _p(x2), _p(mean), _p(rstd), _p(gamma.detach()), _p(dx), _p(dg), _p(db), R, D,
                     _p(dr2), _p(ws), ws.numel(), 1 if batched else 3, stream())
        if batched:
            _LnFinish.add(ws, R, D, dg, db)
        return dx.reshape(ctx.xshape), dg, db, None, None


def layer_norm(x, gamma, beta, eps: float):
    return _LayerNorm.apply(x, gamma, beta, eps, False)


def layer_norm_with_residual(x, gamma, beta, eps: float):
    """(LayerNorm(x), x): use the second output for the residual connection (see _LayerNorm)."""
    return _LayerNorm.apply(x, gamma, beta, eps, True)


# ---------------------------------------------------------------------------------------------------------------------
# Scaled dot-product attention core on head-interleaved buffers
# ---------------------------------------------------------------------------------------------------------------------

def _mask_bytes(mask: torch.Tensor) -> torch.Tensor:
    """The attention mask as contiguous bytes.  A bool tensor already IS one byte per element holding 0 / 1: it is reinterpreted, not
    converted (a `.to(torch.uint8)` here cost one copy kernel per attention call, 24 per training step)."""
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    elif mask.dtype != torch.uint8:
        mask = mask.to(torch.uint8)
    return mask if mask.is_contiguous() else mask.contiguous()


class _AttnCore(torch.autograd.Function):
    """q_buf (B, Tq, nq*d) and kv_buf (B, Tk, nkv*d) hold projections side by side along the last dim; q/k/v are the
    d-wide column blocks iq / ik / iv of those buffers (for self-attention q_buf IS kv_buf).  Heads are the dh-wide
    column groups inside a block.  Returns the merged-head context (B, Tq, d) and the probabilities (B, H, Tq, Tk)."""

    @staticmethod
    def forward(ctx, q_buf, kv_buf, same, iq, ik, iv, H, d, mask, causal, scale, drop_p):
        require_gpu(q_buf, "attention")
        if same:
            kv_buf = q_buf
        B, Tq = q_buf.shape[:2]
        Tk = kv_buf.size(1)
        dh = d // H
        code = dcode(q_buf)
        dt, dev = q_buf.dtype, q_buf.device
        ldq, ldk = q_buf.stride(1), kv_buf.stride(1)
        sqb, skb = q_buf.stride(0), kv_buf.stride(0)
        ld = _up(Tk, 8)
        Z = B * H
        S = torch.empty(Z, Tq, ld, dtype=dt, device=dev)
        # S = scale * Q K^T
        gemm(code, _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(S), Tq, Tk, dh, ldq, ldk, ld, nb=(B, H), sA=(sqb, dh), sB=(skb, dh),
             sC=(H * Tq * ld, Tq * ld), alpha=scale)
        m8 = mb = mq = None
        if mask is not None:
            m8 = _mask_bytes(mask)
            mb = m8.stride(0)
            mq = m8.stride(1) if m8.size(1) > 1 else 0
        Pd = None
        seed = off = 0
        if drop_p > 0:
            Pd = torch.empty_like(S)
            seed, off = DropoutState.next(S.numel(), dev)
        lib.call("sc_softmax_fwd", code, _p(S), _p(S), _p(Pd), Z, H, Tq, Tk, ld, _p(m8), mb or 0, mq or 0, 1 if causal else 0,
                 float(drop_p), seed, off, stream())
        P = S
        Pv = Pd if Pd is not None else P
        out = torch.empty(B, Tq, d, dtype=dt, device=dev)
        # ctx = Pd V  (V rows are keys: contraction index strided -> tb=1)
        gemm(code, _p(Pv), _p(kv_buf, iv * d), _p(out), Tq, dh, Tk, ld, ldk, d, tb=1, nb=(B, H), sA=(H * Tq * ld, Tq * ld),
             sB=(skb, dh), sC=(Tq * d, dh))
        ctx.save_for_backward(q_buf, kv_buf, P, Pv)
        ctx.meta = (same, iq, ik, iv, H, d, scale, drop_p, seed, off, ld)
        probs = P.view(B, H, Tq, ld)[..., :Tk]
        ctx.mark_non_differentiable(probs)
        return out, probs

    @staticmethod
    def backward(ctx, dout, _dprobs):
        q_buf, kv_buf, P, Pv = ctx.saved_tensors
        same, iq, ik, iv, H, d, scale, drop_p, seed, off, ld = ctx.meta
        B, Tq = q_buf.shape[:2]
        Tk = kv_buf.size(1)
        dh = d // H
        code = dcode(q_buf)
        dt, dev = q_buf.dtype, q_buf.device
        Z = B * H
        ldq, ldk = q_buf.stride(1), kv_buf.stride(1)
        sqb, skb = q_buf.stride(0), kv_buf.stride(0)
        dout = dout.contiguous()
        dq_buf = torch.zeros_like(q_buf) if q_buf.size(-1) > (3 * d if same else d) else torch.empty_like(q_buf)
        dkv_buf = dq_buf if same else torch.empty_like(kv_buf)
        zs = (H * Tq * ld, Tq * ld)
        # dPd = dout V^T
        dP = torch.empty(Z, Tq, ld, dtype=dt, device=dev)
        side = _Side(q_buf)
        with side:  # dV = Pd^T dout, independent of the dP -> dS -> dQ/dK chain
            gemm(code, _p(Pv), _p(dout), _p(dkv_buf, iv * d), Tk, dh, Tq, ld, d, dkv_buf.stride(1), ta=1, tb=1, nb=(B, H), sA=zs,
                 sB=(Tq * d, dh), sC=(dkv_buf.stride(0), dh))
        gemm(code, _p(dout), _p(kv_buf, iv * d), _p(dP), Tq, Tk, dh, d, ldk, ld, nb=(B, H), sA=(Tq * d, dh), sB=(skb, dh), sC=zs)
        # dS (w.r.t. the unscaled products) = scale * P * (dP - rowsum(dP * P)), written in place over dP
        lib.call("sc_softmax_bwd", code, _p(P), _p(dP), _p(dP), Z * Tq, Tk, ld, float(scale), float(drop_p), seed, off, stream())
        # dQ = dS K ;  dK = dS^T Q
        with side:
            gemm(code, _p(dP), _p(q_buf, iq * d), _p(dkv_buf, ik * d), Tk, dh, Tq, ld, ldq, dkv_buf.stride(1), ta=1, tb=1, nb=(B, H),
                 sA=zs, sB=(sqb, dh), sC=(dkv_buf.stride(0), dh))
        gemm(code, _p(dP), _p(kv_buf, ik * d), _p(dq_buf, iq * d), Tq, dh, Tk, ld, ldk, dq_buf.stride(1), tb=1, nb=(B, H), sA=zs,
             sB=(skb, dh), sC=(dq_buf.stride(0), dh))
        side.join()
        return dq_buf, (None if same else dkv_buf), None, None, None, None, None, None, None, None, None, None


_want_probs = True
_ATTN_ANY_DH = _os.environ.get("SC_ATTN_ANY_DH", "1") == "1"  # (A/B knob: 0 = head sizes other than 64 take the unfused GEMM + softmax path)


def set_attention_probs(flag: bool):
    """True (default, the reference's contract): attention modules return the (B, H, Tq, Tk) probabilities.  False: they
    return None instead, which lets bf16 attention with head size 64 and <= 256 keys run the fused kernel that never
    writes scores to HBM (training harnesses that do not read the maps)."""
    global _want_probs
    _want_probs = bool(flag)


class _AttnFused(torch.autograd.Function):
    """Fused forward (sc_attention_fwd); the backward pass recomputes the probabilities from the saved logsumexp."""

    @staticmethod
    def forward(ctx, q_buf, kv_buf, same, iq, ik, iv, H, d, mask, causal, scale, drop_p):
        if same:
            kv_buf = q_buf
        B, Tq = q_buf.shape[:2]
        Tk = kv_buf.size(1)
        dev = q_buf.device
        out = torch.empty(B, Tq, d, dtype=q_buf.dtype, device=dev)
        lse = torch.empty(B * H, Tq, dtype=torch.float32, device=dev)
        m8 = None
        mb = mq = 0
        if mask is not None:
            m8 = _mask_bytes(mask)
            mb, mq = m8.stride(0), (m8.stride(1) if m8.size(1) > 1 else 0)
        ld = _up(Tk, 8)
        seed, off = DropoutState.next(B * H * Tq * ld, dev) if drop_p > 0 else (0, 0)
        if d // H != 64:  # any other head size (a multiple of 8 up to 256; key masks / causal flag only): csrc/attention_dh.hip
            lib.call("sc_attention_dh_fwd", _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(kv_buf, iv * d), _p(out), _p(lse), B, H, Tq, Tk, d // H,
                     q_buf.stride(0), q_buf.stride(1), kv_buf.stride(0), kv_buf.stride(1), out.stride(0), out.stride(1), _p(m8), mb,
                     1 if causal else 0, float(scale), float(drop_p), seed, off, stream())
        else:
            lib.call("sc_attention_fwd", _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(kv_buf, iv * d), _p(out), _p(lse), B, H, Tq, Tk,
                     q_buf.stride(0), q_buf.stride(1), kv_buf.stride(0), kv_buf.stride(1), out.stride(0), out.stride(1), _p(m8), mb, mq,
                     1 if causal else 0, float(scale), float(drop_p), seed, off, stream())
        ctx.save_for_backward(q_buf, kv_buf, m8, lse, out)
        ctx.meta = (same, iq, ik, iv, H, d, causal, scale, drop_p, seed, off, ld, mb, mq)
        return out

    @staticmethod
    def backward(ctx, dout):
        q_buf, kv_buf, m8, lse, out = ctx.saved_tensors
        same, iq, ik, iv, H, d, causal, scale, drop_p, seed, off, ld, mb, mq = ctx.meta
        B, Tq = q_buf.shape[:2]
        Tk = kv_buf.size(1)
        dh = d // H
        code = dcode(q_buf)
        dt, dev = q_buf.dtype, q_buf.device
        Z = B * H
        ldq, ldk = q_buf.stride(1), kv_buf.stride(1)
        sqb, skb = q_buf.stride(0), kv_buf.stride(0)
        zs = (H * Tq * ld, Tq * ld)
        dout = dout.contiguous()
        if mq == 0 and dh != 64:  # (head sizes other than 64: rowsum(dO * O), then dK / dV by key blocks and dQ by query blocks)
            dq_buf = torch.empty_like(q_buf)
            dkv_buf = dq_buf if same else torch.empty_like(kv_buf)
            ws_bytes = lib.value("sc_attention_dh_bwd_workspace", B, H, Tq)
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
            lib.call("sc_attention_dh_bwd", _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(kv_buf, iv * d), _p(out), _p(dout), _p(lse),
                     _p(dq_buf, iq * d), _p(dkv_buf, ik * d), _p(dkv_buf, iv * d), B, H, Tq, Tk, dh, sqb, ldq, skb, ldk, out.stride(0),
                     out.stride(1), dq_buf.stride(0), dq_buf.stride(1), dkv_buf.stride(0), dkv_buf.stride(1), _p(m8), mb,
                     1 if causal else 0, float(scale), float(drop_p), seed, off, _p(ws), ws_bytes, stream())
            return dq_buf, (None if same else dkv_buf), None, None, None, None, None, None, None, None, None, None
        if mq == 0:  # fused backward: probabilities are recomputed on chip from Q, K and the saved logsumexp
            dq_buf = torch.empty_like(q_buf)
            dkv_buf = dq_buf if same else torch.empty_like(kv_buf)
            ws_bytes = lib.value("sc_attention_bwd_workspace", B, H, Tq, Tk)
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev) if ws_bytes else None
            lib.call("sc_attention_bwd", _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(kv_buf, iv * d), _p(out), _p(dout), _p(lse),
                     _p(dq_buf, iq * d), _p(dkv_buf, ik * d), _p(dkv_buf, iv * d), B, H, Tq, Tk, sqb, ldq, skb, ldk, out.stride(0),
                     out.stride(1), dq_buf.stride(0), dq_buf.stride(1), dkv_buf.stride(0), dkv_buf.stride(1), _p(m8), mb, mq,
                     1 if causal else 0, float(scale), float(drop_p), seed, off, _p(ws), ws_bytes, stream())
            return dq_buf, (None if same else dkv_buf), None, None, None, None, None, None, None, None, None, None
        # dense (B, Tq, Tk) mask: recompute P (and its dropped copy) exactly as the unfused forward does
        S = torch.empty(Z, Tq, ld, dtype=dt, device=dev)
        gemm(code, _p(q_buf, iq * d), _p(kv_buf, ik * d), _p(S), Tq, Tk, dh, ldq, ldk, ld, nb=(B, H), sA=(sqb, dh), sB=(skb, dh),
             sC=zs, alpha=scale)
        Pd = torch.empty_like(S) if drop_p > 0 else None
        lib.call("sc_softmax_fwd", code, _p(S), _p(S), _p(Pd), Z, H, Tq, Tk, ld, _p(m8), mb, mq, 1 if causal else 0, float(drop_p),
                 seed, off, stream())
        P, Pv = S, (Pd if Pd is not None else S)
        dq_buf = torch.empty_like(q_buf)
        dkv_buf = dq_buf if same else torch.empty_like(kv_buf)
        dP = torch.empty(Z, Tq, ld, dtype=dt, device=dev)
        side = _Side(q_buf)
        with side:
            gemm(code, _p(Pv), _p(dout), _p(dkv_buf, iv * d), Tk, dh, Tq, ld, d, dkv_buf.stride(1), ta=1, tb=1, nb=(B, H), sA=zs,
                 sB=(Tq * d, dh), sC=(dkv_buf.stride(0), dh))
        gemm(code, _p(dout), _p(kv_buf, iv * d), _p(dP), Tq, Tk, dh, d, ldk, ld, nb=(B, H), sA=(Tq * d, dh), sB=(skb, dh), sC=zs)
        lib.call("sc_softmax_bwd", code, _p(P), _p(dP), _p(dP), Z * Tq, Tk, ld, float(scale), float(drop_p), seed, off, stream())
        with side:
            gemm(code, _p(dP), _p(q_buf, iq * d), _p(dkv_buf, ik * d), Tk, dh, Tq, ld, ldq, dkv_buf.stride(1), ta=1, tb=1, nb=(B, H),
                 sA=zs, sB=(sqb, dh), sC=(dkv_buf.stride(0), dh))
        gemm(code, _p(dP), _p(kv_buf, ik * d), _p(dq_buf, iq * d), Tq, dh, Tk, ld, ldk, dq_buf.stride(1), tb=1, nb=(B, H), sA=zs,
             sB=(skb, dh), sC=(dq_buf.stride(0), dh))
        side.join()
        return dq_buf, (None if same else dkv_buf), None, None, None, None, None, None, None, None, None, None


def attention_core(q_buf, kv_buf, iq, ik, iv, num_heads, d_model, mask, causal, scale, drop_p):
    same = kv_buf is None or kv_buf is q_buf
    Tk = (q_buf if same else kv_buf).size(1)
    # (any number of keys: the streaming forward tiles them, the backward sweeps them in blocks of 256.  A query-dependent mask -- only
    #  Conformer-style subclasses pass one -- takes the forward kernel that holds a head's keys in LDS: at most 256)
    dense_mask = mask is not None and mask.dim() == 3 and mask.size(1) > 1
    dh = d_model // num_heads
    fused_ok = (dh == 64 and (Tk <= 256 or not dense_mask)) or (_ATTN_ANY_DH and dh != 64 and dh % 8 == 0 and dh <= 256 and not dense_mask)
    if (not _want_probs and q_buf.dtype == torch.bfloat16 and fused_ok
            and q_buf.stride(1) % 8 == 0 and (same or kv_buf.stride(1) % 8 == 0)):
        out = _AttnFused.apply(q_buf, None if same else kv_buf, same, iq, ik, iv, num_heads, d_model, mask, bool(causal), float(scale),
                               float(drop_p))
        return out, None
    return _AttnCore.apply(q_buf, None if same else kv_buf, same, iq, ik, iv, num_heads, d_model, mask, bool(causal), float(scale),
                           float(drop_p))


# ---------------------------------------------------------------------------------------------------------------------
# Positional encoding add
# ---------------------------------------------------------------------------------------------------------------------

_POSENC_DOT = _os.environ.get("SC_POSENC_DOT", "1") == "1"


class _PosEnc(torch.autograd.Function):
    _ws = {}

    @staticmethod
    def forward(ctx, x, table, alpha, xscale, drop_p):
        require_gpu(x, "positional encoding")
        x = x.contiguous()
        B, T, D = x.shape
        y = torch.empty_like(x)
        seed, off = DropoutState.next(x.numel(), x.device) if drop_p > 0 else (0, 0)
        tab = table.reshape(-1, D)
        lib.call("sc_posenc_fwd", dcode(x), _p(x), _p(tab), _p(alpha.detach()) if alpha is not None else None, float(xscale), _p(y),
                 B, T, D, float(drop_p), seed, off, stream())
        ctx.meta = (xscale, drop_p, seed, off, alpha is not None, T, D)
        if alpha is not None:
            ctx.save_for_backward(tab)
        return y

    @staticmethod
    def backward(ctx, dy):
        xscale, drop_p, seed, off, has_alpha, T, D = ctx.meta
        dy = dy.contiguous()
        dz = _dropout_apply(dy, drop_p, seed, off)  # gradient w.r.t. (x * xscale + table * alpha)
        dalpha = None
        if has_alpha:
            (tab,) = ctx.saved_tensors
            # scalar parameter of the TTS recipes: one fused reduction (sc_dot_table) instead of a cast, a broadcast product and a sum
            if not _POSENC_DOT:  # (A/B knob SC_POSENC_DOT=0: the torch form)
                return (dz if xscale == 1.0 else dz * xscale), None, (dz.float() * tab[:T]).sum(), None, None
            B = dz.numel() // (T * D)
            ws = _PosEnc._ws.get(dz.device)
            if ws is None:
                ws = _PosEnc._ws[dz.device] = torch.empty(lib.value("sc_least_error_workspace"), dtype=torch.uint8, device=dz.device)
            dalpha = torch.empty(1, dtype=torch.float32, device=dz.device)
            tabc = tab[:T].float().contiguous()
            lib.call("sc_dot_table", dcode(dz), _p(dz), _p(tabc), B, T, D, _p(ws), _p(dalpha), stream())
            dalpha = dalpha.reshape(())
        if xscale != 1.0:
            dx = torch.empty_like(dz)
            lib.call("sc_dropout", dcode(dz), _p(dz), _p(dx), dz.numel(), float(xscale), 0.0, 0, 0, stream())
        else:
            dx = dz
        return dx, None, dalpha, None, None


def posenc_add(x, table, alpha, xscale: float, drop_p: float):
    return _PosEnc.apply(x, table, alpha, float(xscale), float(drop_p))


# ---------------------------------------------------------------------------------------------------------------------
# Losses
# ---------------------------------------------------------------------------------------------------------------------

class _CrossEntropyRows(torch.autograd.Function):
    """Per-token losses (R,) float32 of (R, V) logits; masking / sentence sums / means stay tiny torch ops on (R,)."""

    @staticmethod
    def forward(ctx, logits2, target, smoothing, temperature):
        require_gpu(logits2, "cross entropy")
        logits2 = logits2.contiguous()
        R, V = logits2.shape
        target = target.contiguous()
        row_loss = torch.empty(R, dtype=torch.float32, device=logits2.device)
        lse = torch.empty(R, dtype=torch.float32, device=logits2.device)
        lib.call("sc_ce_fwd", dcode(logits2), _p(logits2), _p(target), R, V, V, float(temperature), float(smoothing), _p(row_loss),
                 _p(lse), stream())
        ctx.save_for_backward(logits2, target, lse)
        ctx.meta = (smoothing, temperature)
        return row_loss

    @staticmethod
    def backward(ctx, grow):
        logits2, target, lse = ctx.saved_tensors
        smoothing, temperature = ctx.meta
        R, V = logits2.shape
        d = torch.empty_like(logits2)
        grow = grow.float().contiguous()
        lib.call("sc_ce_bwd", dcode(logits2), _p(logits2), _p(target), _p(lse), _p(grow), R, V, V, float(temperature), float(smoothing),
                 _p(d), stream())
        return d, None, None, None


def cross_entropy_rows(logits2, target, smoothing: float, temperature: float = 1.0):
    return _CrossEntropyRows.apply(logits2, target, float(smoothing), float(temperature))


class _SeqLoss(torch.autograd.Function):
    """mean over the sentences of the (length-normalised) sum of their valid per-token losses, one launch each way (sc_seq_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, rows, text_len, length_normalized):
        require_gpu(rows, "sequence loss")
        rows = rows.float().contiguous()
        B, L = rows.shape
        tl = text_len.to(device=rows.device, dtype=torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=rows.device)
        lib.call("sc_seq_loss_fwd", _p(rows), _p(tl), B, L, 1 if length_normalized else 0, _p(loss), stream())
        ctx.save_for_backward(tl)
        ctx.meta = (B, L, length_normalized)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (tl,) = ctx.saved_tensors
        B, L, ln = ctx.meta
        g1 = g.float().reshape(1).contiguous()
        grow = torch.empty(B, L, dtype=torch.float32, device=g.device)
        lib.call("sc_seq_loss_bwd", _p(g1), _p(tl), B, L, 1 if ln else 0, _p(grow), stream())
        return grow, None, None


def seq_loss(rows: torch.Tensor, text_len: torch.Tensor, length_normalized: bool) -> torch.Tensor:
    """Scalar criterion value from (B, L) per-token losses and the sentence lengths (cross_entropy.py:201-212 without its options)."""
    return _SeqLoss.apply(rows, text_len, bool(length_normalized))


class _LeastError(torch.autograd.Function):
    """Masked L1 / L2 / L1+L2 of two (B, T, D) tracks over the frames t < len[b] as a scalar (sc_least_error_fwd / _bwd): two launches
    forward, one backward, instead of ~10 torch launches each way (criterion/least_error.py)."""

    _ws = {}

    @staticmethod
    def forward(ctx, pred, tgt, tgt_len, loss_type, normalized):
        B, T, D = pred.shape
        tl = tgt_len.to(device=pred.device, dtype=torch.int64).contiguous()
        ws = _LeastError._ws.get(pred.device)
        if ws is None:
            ws = _LeastError._ws[pred.device] = torch.empty(lib.value("sc_least_error_workspace"), dtype=torch.uint8, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        lib.call("sc_least_error_fwd", dcode(pred), _p(pred), _p(tgt), _p(tl), B, T, D, loss_type, 1 if normalized else 0, _p(ws), _p(loss), stream())
        ctx.save_for_backward(pred, tgt, tl)
        ctx.meta = (loss_type, normalized)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, tgt, tl = ctx.saved_tensors
        loss_type, normalized = ctx.meta
        B, T, D = pred.shape
        g1 = g.float().reshape(1).contiguous()
        dpred = torch.empty_like(pred)
        lib.call("sc_least_error_bwd", dcode(pred), _p(pred), _p(tgt), _p(tl), B, T, D, loss_type, 1 if normalized else 0, _p(g1), _p(dpred), stream())
        return dpred, None, None, None, None


def least_error(pred: torch.Tensor, tgt: torch.Tensor, tgt_len: torch.Tensor, loss_type: str, normalized: bool) -> torch.Tensor:
    """LeastError of pred (B, T, D) f32 / bf16, contiguous, against tgt (B, T, D) (no gradient) over the frames t < tgt_len[b]."""
    require_gpu(pred, "least_error")
    assert pred.dim() == 3 and pred.shape == tgt.shape and pred.is_contiguous() and not tgt.requires_grad
    return _LeastError.apply(pred, tgt.detach().float().contiguous(), tgt_len, {"L1": 1, "L2": 2, "L1+L2": 3}[loss_type], bool(normalized))


class _CTC(torch.autograd.Function):
    """Per-utterance CTC negative log-likelihood (B,) float32 with zero_infinity handling."""

    @staticmethod
    def forward(ctx, logits, in_len, tgt, tgt_len, blank, zero_infinity):
        require_gpu(logits, "CTC")
        logits = logits.contiguous()
        B, T, V = logits.shape
        dev = logits.device
        in_len = in_len.to(device=dev, dtype=torch.int64).contiguous()
        tgt_len = tgt_len.to(device=dev, dtype=torch.int64).contiguous()
        tgt = tgt.to(device=dev, dtype=torch.int64)
        assert tgt.stride(1) == 1
        Smax = 2 * tgt.size(1) + 1
        lse = torch.empty(B * T, dtype=torch.float32, device=dev)
        lib.call("sc_row_lse", dcode(logits), _p(logits), B * T, V, V, _p(lse), stream())
        alpha = torch.empty(B, T, Smax, dtype=torch.float32, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        if ctx.needs_input_grad[0]:  # a training step: both recursions in one launch (each is one serial chain of frames per utterance)
            beta = torch.empty_like(alpha)
            lib.call("sc_ctc_fwd_train", dcode(logits), _p(logits), _p(lse), _p(in_len), _p(tgt), tgt.stride(0), _p(tgt_len), B, T, V, Smax,
                     blank, _p(alpha), _p(beta), _p(nll), stream())
            ctx.save_for_backward(logits, lse, in_len, tgt, tgt_len, alpha, nll, beta)
        else:
            lib.call("sc_ctc_fwd", dcode(logits), _p(logits), _p(lse), _p(in_len), _p(tgt), tgt.stride(0), _p(tgt_len), B, T, V, Smax,
                     blank, _p(alpha), _p(nll), stream())
            ctx.save_for_backward(logits, lse, in_len, tgt, tgt_len, alpha, nll)
        ctx.meta = (blank, zero_infinity, Smax)
        if zero_infinity:
            return torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
        return nll.clone()

    @staticmethod
    def backward(ctx, g):
        logits, lse, in_len, tgt, tgt_len, alpha, nll = ctx.saved_tensors[:7]
        blank, zero_infinity, Smax = ctx.meta
        B, T, V = logits.shape
        gscale = g.float().contiguous()
        if zero_infinity:
            gscale = torch.where(torch.isinf(nll), torch.zeros_like(gscale), gscale)
        grad = torch.empty_like(logits)
        if len(ctx.saved_tensors) == 8:  # beta came with the forward pass (sc_ctc_fwd_train)
            lib.call("sc_ctc_grad", dcode(logits), _p(logits), _p(lse), _p(in_len), _p(tgt), tgt.stride(0), _p(tgt_len), B, T, V, Smax, blank,
                     _p(alpha), _p(ctx.saved_tensors[7]), _p(nll), _p(gscale), _p(grad), stream())
            return grad, None, None, None, None, None
        beta = torch.empty_like(alpha)  # (B, T, Smax) workspace of the backward recursion
        lib.call("sc_ctc_bwd", dcode(logits), _p(logits), _p(lse), _p(in_len), _p(tgt), tgt.stride(0), _p(tgt_len), B, T, V, Smax, blank,
                 _p(alpha), _p(nll), _p(gscale), _p(beta), _p(grad), stream())
        return grad, None, None, None, None, None


def ctc_nll(logits, in_len, tgt, tgt_len, blank: int = 0, zero_infinity: bool = True):
    return _CTC.apply(logits, in_len, tgt, tgt_len, int(blank), bool(zero_infinity))


# ---------------------------------------------------------------------------------------------------------------------
# Conv2d prenet blocks (csrc/prenet.hip): channels-last (B, T', F', C) images in the compute dtype
# ---------------------------------------------------------------------------------------------------------------------
PACT = {None: 0, "ReLU": 1, "LeakyReLU": 2, "Tanh": 3}  # (Tanh: blocks after the first only -- the Conv1d postnet of the TTS recipes)


def conv_out_size(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def _bn_eval_coeffs(gamma, beta, rm, rv, eps):
    rstd = torch.rsqrt(rv.detach() + eps)
    scale = gamma.detach() * rstd
    return scale, beta.detach() - rm.detach() * scale, rm.detach().clone(), rstd


def _sink_or_new(params, shape, device):
    sink = GradSinks.take(params)
    return sink.view(shape) if sink is not None else torch.empty(shape, dtype=torch.float32, device=device)


class _RankSums:
    """The exchange step of ``torch.nn.SyncBatchNorm``: per-rank float64 sum vectors (and position counts) summed over a process group.
    The reference turns every BatchNorm of a DDP model into its synchronised form (speechain/runner.py:303-310 ``enable_syncbatchnorm``,
    default True; :1856-1858), so under DDP the prenet / postnet normalise with the statistics of the batches of ALL ranks.  Here the HIP
    kernels produce the rank's sums, one small all-reduce (C-sized, on the current stream) makes them global, and the finish kernels read
    both (include/speechain_amd.h, "Synchronised batch statistics")."""

    def __init__(self, group):
        self.group = group

    def sum_with_count(self, local: torch.Tensor, count: int):
        buf = torch.cat([local.reshape(-1), torch.full((1,), float(count), dtype=torch.float64, device=local.device)])
        torch.distributed.all_reduce(buf, group=self.group)
        return buf[:-1], buf[-1:]

    def sum(self, local: torch.Tensor):
        buf = local.clone()
        torch.distributed.all_reduce(buf, group=self.group)
        return buf

    @staticmethod
    def of(bn, use_batch: bool, training: bool):
        """The exchange object for a BatchNorm module, or None: only a ``SyncBatchNorm`` (what ``convert_sync_batchnorm`` leaves in place of
        the ``nn.BatchNorm*d`` holders) in training mode with a live process group of more than one rank synchronises, as torch's does."""
        if not (isinstance(bn, torch.nn.SyncBatchNorm) and use_batch and training):
            return None
        import torch.distributed as dist

        if not dist.is_initialized():
            return None
        group = bn.process_group
        if dist.get_world_size(group) < 2 and _os.environ.get("SC_DDP_FORCE_COLLECTIVES", "0") != "1":
            return None
        return _RankSums(group)


class _ConvFirst(torch.autograd.Function):
    """Conv2d(1, C, kernel, stride, padding) [+ BatchNorm2d] [+ activation] on a (B, T, F) float32 image -> (B, To, Fo, C).
    Training-mode BatchNorm takes its statistics from the patch moments (no pre-normalisation tensor exists); the running
    statistics are updated in place, as torch does."""

    @staticmethod
    def forward(ctx, feat, w, bias, gamma, beta, rm, rv, geom, act, slope, training, momentum, eps, dtype, sync=None):
        require_gpu(feat, "the Conv2d prenet")
        (kh, kw), (sh, sw), (ph, pw) = geom
        x = feat.detach()
        if x.dtype != torch.float32 or x.stride(2) != 1:
            x = x.float().contiguous()
        B, Ti, Fi = x.shape
        To, Fo = conv_out_size(Ti, kh, sh, ph), conv_out_size(Fi, kw, sw, pw)
        C, KK = w.size(0), kh * kw
        dev = x.device
        w2 = w.detach().reshape(C, KK).contiguous()
        g6 = (B, Ti, Fi, To, Fo, kh, kw, sh, sw, ph, pw)
        moments = mean = rstd = None
        bn = gamma is not None
        if bn and training:
            ws = torch.empty(lib.value("sc_conv1_moments_workspace", *g6), dtype=torch.uint8, device=dev)
            moments = torch.empty(KK * KK + KK, dtype=torch.float64, device=dev)
            lib.call("sc_conv1_moments", _p(x), x.stride(0), x.stride(1), *g6, _p(moments), _p(ws), ws.numel(), stream())
            scale, shift, mean, rstd = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
            # synchronised statistics: the moments of all ranks' patches and their count (this rank's moments are kept for backward)
            mom_all, count_all = sync.sum_with_count(moments, B * To * Fo) if sync is not None else (moments, None)
            lib.call("sc_conv1_bn_coeffs", _p(mom_all), _p(w2), C, KK, B * To * Fo, _p(count_all), _p(gamma.detach()), _p(beta.detach()),
                     float(eps), float(momentum), _p(rm), _p(rv), _p(scale), _p(shift), _p(mean), _p(rstd), stream())
        elif bn:
            scale, shift, mean, rstd = _bn_eval_coeffs(gamma, beta, rm, rv, eps)
        else:
            scale = torch.ones(C, dtype=torch.float32, device=dev)
            shift = bias.detach().float() if bias is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        y = torch.empty(B, To, Fo, C, dtype=dtype, device=dev)
        lib.call("sc_conv1_apply", dcode(y), _p(x), x.stride(0), x.stride(1), *g6, _p(w2), C, _p(scale), _p(shift), act, float(slope), _p(y),
                 stream())
        ctx.save_for_backward(x, w, scale, shift, moments, gamma, mean, rstd)
        ctx.params = (w, bias, gamma, beta)
        ctx.meta = (g6, act, slope, bn, bn and training)
        ctx.sync = (sync, count_all) if (sync is not None and bn and training) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("speechain_amd Conv2d prenet: no gradient w.r.t. the input features (the frontend in front of it "
                                      "has no trainable parameter, module/encoder/asr.py:103-131)")
        x, w, scale, shift, moments, gamma, mean, rstd = ctx.saved_tensors
        g6, act, slope, bn, batch_stats = ctx.meta
        pw_, pb, pg, pbe = ctx.params
        B, To, kh, kw = g6[0], g6[3], g6[5], g6[6]
        C, KK = w.size(0), kh * kw
        dev = x.device
        dy = dy if dy.is_contiguous() else dy.contiguous()
        w2 = w.detach().reshape(C, KK).contiguous()
        dw = _sink_or_new([pw_], (C, KK), dev)
        dgamma = _sink_or_new([pg], (C,), dev) if bn else None
        dbeta = _sink_or_new([pbe], (C,), dev) if bn else (_sink_or_new([pb], (C,), dev) if pb is not None else None)
        ws = torch.empty(lib.value("sc_conv1_bwd_workspace", *g6, C), dtype=torch.uint8, device=dev)
        if ctx.sync is None:
            lib.call("sc_conv1_bwd", dcode(dy), _p(x), x.stride(0), x.stride(1), *g6, _p(w2), C, _p(scale), _p(shift), act, float(slope), _p(dy),
                     _p(moments), _p(gamma.detach()) if bn else None, _p(mean), _p(rstd), 1 if batch_stats else 0, _p(dw), _p(dgamma), _p(dbeta),
                     _p(ws), ws.numel(), stream())
        else:  # reduce -> all-reduce of (sum g.patch, sum g) -> finish with this rank's and all ranks' sums
            sync, count_all = ctx.sync
            gsum = torch.empty(C * (KK + 1), dtype=torch.float64, device=dev)
            lib.call("sc_conv1_bwd_reduce", dcode(dy), _p(x), x.stride(0), x.stride(1), *g6, _p(w2), C, _p(scale), _p(shift), act, float(slope),
                     _p(dy), _p(gsum), _p(ws), ws.numel(), stream())
            gsum_all = sync.sum(gsum)
            lib.call("sc_conv1_bwd_finish", _p(gsum), _p(gsum_all), _p(moments), _p(w2), C, KK, B * To * g6[4], _p(count_all), _p(gamma.detach()),
                     _p(mean), _p(rstd), 1, _p(dw), _p(dgamma), _p(dbeta), stream())
        return (None, dw.view_as(w), None if bn else dbeta, dgamma, dbeta if bn else None) + (None,) * 10


def _conv1d_base(w: torch.Tensor):
    """The nn.Conv1d parameter (out, in, k) behind an (out, in, k, 1) image-kernel view of it, or None."""
    b = getattr(w, "_base", None)
    if w.dim() == 4 and w.size(3) == 1 and b is not None and b.is_leaf and b.dim() == 3 and tuple(b.shape) == tuple(w.shape[:3]):
        return b
    return None


class _ConvNext(torch.autograd.Function):
    """Conv2d(Cin, Cout, kernel, stride, padding) [+ BatchNorm2d] [+ activation] on a channels-last (B, Ti, Fi, Cin) image:
    im2col -> one GEMM against Wm[c_out][(kh, kw, c_in)] -> column statistics -> normalise + activation."""

    @staticmethod
    def forward(ctx, a, w, bias, gamma, beta, rm, rv, geom, act, slope, training, momentum, eps, sync=None):
        require_gpu(a, "the Conv2d prenet")
        (kh, kw), (sh, sw), (ph, pw) = geom
        a = a if a.is_contiguous() else a.contiguous()
        B, Ti, Fi, Cin = a.shape
        To, Fo = conv_out_size(Ti, kh, sh, ph), conv_out_size(Fi, kw, sw, pw)
        Cout, K, M = w.size(0), kh * kw * Cin, B * To * Fo
        dev, code = a.device, dcode(a)
        g = (B, Ti, Fi, Cin, To, Fo, kh, kw, sh, sw, ph, pw)
        col = torch.empty(M, K, dtype=a.dtype, device=dev)
        lib.call("sc_im2col", code, _p(a), *g, _p(col), stream())
        # (c_out, (kh, kw, c_in)) view of the weight in the compute dtype.  A Conv1d weight seen as an image kernel (the (out, in, k, 1)
        # view the Conv1d postnet passes) has exactly the weight cache's Conv1d layout, column = tap * C + channel: its copies then come
        # from the one refresh launch of the step instead of a permute + cast per block and direction.  Other kernels: cast here (the
        # permuted copy is a new tensor every step).
        base = _conv1d_base(w)
        Wc = weights.fused([base], a.dtype) if base is not None else w.detach().permute(0, 2, 3, 1).reshape(Cout, K).to(a.dtype)
        bn = gamma is not None
        z = torch.empty(M, Cout, dtype=a.dtype, device=dev)
        gemm(code, _p(col), _p(Wc), _p(z), M, Cout, K, K, K, Cout, bias=bias.detach() if (bias is not None and not bn) else None)
        mean = rstd = None
        if bn and training:
            scale, shift, mean, rstd = (torch.empty(Cout, dtype=torch.float32, device=dev) for _ in range(4))
            ws = torch.empty(lib.value("sc_bn_workspace", M, Cout), dtype=torch.uint8, device=dev)
            if sync is None:
                lib.call("sc_bn_stats", code, _p(z), M, Cout, _p(gamma.detach()), _p(beta.detach()), float(eps), float(momentum), _p(rm), _p(rv),
                         _p(scale), _p(shift), _p(mean), _p(rstd), _p(ws), ws.numel(), stream())
            else:  # column sums -> all-reduce (sum z, sum z^2, rows) -> statistics of the batches of all ranks
                sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
                lib.call("sc_bn_sums", code, _p(z), None, M, Cout, None, None, 0, 0.0, _p(sums), _p(ws), ws.numel(), stream())
                sums_all, count_all = sync.sum_with_count(sums, M)
                lib.call("sc_bn_stats_finish", _p(sums_all), Cout, M, _p(count_all), _p(gamma.detach()), _p(beta.detach()), float(eps),
                         float(momentum), _p(rm), _p(rv), _p(scale), _p(shift), _p(mean), _p(rstd), stream())
                ctx.sync = (sync, count_all)
        elif bn:
            scale, shift, mean, rstd = _bn_eval_coeffs(gamma, beta, rm, rv, eps)
        else:
            scale = torch.ones(Cout, dtype=torch.float32, device=dev)
            shift = torch.zeros(Cout, dtype=torch.float32, device=dev)
        if bn or act:
            y = torch.empty_like(z)
            lib.call("sc_bn_act_fwd", code, _p(z), M, Cout, _p(scale), _p(shift), act, float(slope), _p(y), stream())
        else:
            y = z
        ctx.save_for_backward(col, z, w, scale, shift, mean, rstd)
        ctx.params = (w, bias, gamma, beta)
        ctx.base = base
        ctx.meta = (g, act, slope, bn, bn and training)
        if not (bn and training and sync is not None):
            ctx.sync = None
        return y.view(B, To, Fo, Cout)

    @staticmethod
    def backward(ctx, dy):
        col, z, w, scale, shift, mean, rstd = ctx.saved_tensors
        g, act, slope, bn, batch_stats = ctx.meta
        pw_, pb, pg, pbe = ctx.params
        B, Ti, Fi, Cin, To, Fo, kh, kw = g[:8]
        M, Cout = z.shape
        K = col.size(1)
        dev, code = z.device, dcode(z)
        dy2 = dy.reshape(M, Cout)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dgamma = dbeta = dbias = None
        if bn or act:
            dz = torch.empty_like(z)
            if bn:
                dgamma, dbeta = _sink_or_new([pg], (Cout,), dev), _sink_or_new([pbe], (Cout,), dev)
            ws = torch.empty(lib.value("sc_bn_workspace", M, Cout) + 12 * Cout, dtype=torch.uint8, device=dev)
            if ctx.sync is None:
                lib.call("sc_bn_act_bwd", code, _p(z), _p(dy2), M, Cout, _p(scale), _p(shift), _p(mean), _p(rstd), 1 if batch_stats else 0, act,
                         float(slope), _p(dz), _p(dgamma), _p(dbeta), _p(ws), ws.numel(), stream())
            else:  # (sum g, sum g z) -> all-reduce -> coefficients of dz from all ranks' sums, dgamma / dbeta from this rank's
                sync, count_all = ctx.sync
                sums = torch.empty(2 * Cout, dtype=torch.float64, device=dev)
                coef = torch.empty(3 * Cout, dtype=torch.float32, device=dev)
                lib.call("sc_bn_sums", code, _p(z), _p(dy2), M, Cout, _p(scale), _p(shift), act, float(slope), _p(sums), _p(ws), ws.numel(), stream())
                sums_all = sync.sum(sums)
                lib.call("sc_bn_bwd_finish", _p(sums), _p(sums_all), Cout, M, _p(count_all), _p(scale), _p(mean), _p(rstd), 1, _p(dgamma),
                         _p(dbeta), _p(coef), stream())
                lib.call("sc_bn_bwd_apply", code, _p(z), _p(dy2), M, Cout, _p(scale), _p(shift), _p(coef), act, float(slope), _p(dz), stream())
        else:
            dz = dy2
        side = _Side(z)
        with side:
            if pb is not None and not bn:
                dWm, dbias = wgrad(code, dz, col, bias=True, bias_params=[pb])
            else:
                dWm = wgrad(code, dz, col)
        da = None
        if ctx.needs_input_grad[0]:
            dcol = torch.empty(M, K, dtype=z.dtype, device=dev)
            if code == BF16:  # k-contiguous product against the transposed weight ((kh, kw, c_in), c_out)
                Wt = (weights.fused_t([ctx.base], z.dtype) if ctx.base is not None
                      else w.detach().permute(2, 3, 1, 0).reshape(K, Cout).to(z.dtype))
                gemm(code, _p(dz), _p(Wt), _p(dcol), M, K, Cout, Cout, Cout, K)
            else:
                Wm = w.detach().permute(0, 2, 3, 1).reshape(Cout, K)
                gemm(code, _p(dz), _p(Wm), _p(dcol), M, K, Cout, Cout, K, K, tb=1)
            da = torch.empty(B, Ti, Fi, Cin, dtype=z.dtype, device=dev)
            lib.call("sc_col2im", code, _p(dcol), *g, _p(da), stream())
        side.join()
        dw = dWm.view(Cout, kh, kw, Cin).permute(0, 3, 1, 2)
        return (da, dw, dbias, dgamma, dbeta) + (None,) * 9


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = x if x.is_contiguous() else x.contiguous()
        seed, off = DropoutState.next(x.numel(), x.device)
        ctx.meta = (p, seed, off)
        return _dropout_apply(x, p, seed, off)

    @staticmethod
    def backward(ctx, dy):
        p, seed, off = ctx.meta
        return _dropout_apply(dy if dy.is_contiguous() else dy.contiguous(), p, seed, off), None


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """nn.Dropout on a compute-dtype tensor with the stateless generator (the mask is regenerated in the backward pass)."""
    return _Dropout.apply(x, float(p)) if (training and p > 0.0) else x


def conv_block(x, conv: torch.nn.Conv2d, bn: Optional[torch.nn.BatchNorm2d], activation: Optional[str], slope: float, training: bool,
               first: bool):
    """One [Conv2d -> BatchNorm2d -> activation] block of the prenet.  `first`: x is the (B, T, F) float32 feature image (one input
    channel) and the result has the compute dtype; otherwise x is a channels-last (B, T', F', C) image."""
    if activation not in PACT:
        raise NotImplementedError(f"speechain_amd Conv2d prenet implements ReLU, LeakyReLU and no activation, got {activation}")
    geom = (tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))
    use_batch = bn is not None and (training or not bn.track_running_stats)
    if bn is not None:
        if bn.momentum is None:
            raise NotImplementedError("speechain_amd Conv2d prenet: BatchNorm2d(momentum=None) (cumulative average) is not implemented")
        args = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        mom, eps = bn.momentum, bn.eps
        if use_batch and bn.track_running_stats and training:
            bn.num_batches_tracked += 1
    else:
        args, mom, eps = (None, None, None, None), 0.0, 0.0
    sync = _RankSums.of(bn, use_batch, training)  # (a SyncBatchNorm holder under a live process group: statistics over all ranks)
    if first:
        return _ConvFirst.apply(x, conv.weight, conv.bias, *args, geom, PACT[activation], slope, use_batch, mom, eps, compute_dtype(), sync)
    return _ConvNext.apply(x, conv.weight, conv.bias, *args, geom, PACT[activation], slope, use_batch, mom, eps, sync)


# ---------------------------------------------------------------------------------------------------------------------
# Token embedding:  out = table[ids]   (gather left to torch; the table gradient by csrc/elementwise.hip, deterministic)
# ---------------------------------------------------------------------------------------------------------------------

class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, padding_idx):
        require_gpu(table, "the token embedding")
        ctx.save_for_backward(ids)
        ctx.table = table
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        return torch.nn.functional.embedding(ids, table.detach())

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        table = ctx.table
        V, D = table.shape
        dy2 = dy.reshape(-1, D)
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        flat = ids.reshape(-1).to(torch.int64)
        flat = flat if flat.is_contiguous() else flat.contiguous()
        sink = GradSinks.take([table])
        dw = sink.view(V, D) if sink is not None else torch.empty(V, D, dtype=torch.float32, device=table.device)
        lib.call("sc_embedding_bwd", dcode(dy2), _p(flat), _p(dy2), dy2.stride(0), flat.numel(), D, V, ctx.padding_idx, _p(dw), stream())
        return dw, None, None


def embedding(table: torch.Tensor, ids: torch.Tensor, padding_idx: Optional[int] = None) -> torch.Tensor:
    """nn.Embedding lookup whose table gradient is produced without atomics (and straight into its gradient-bucket slot)."""
    return _Embedding.apply(table, ids, padding_idx)


# ---------------------------------------------------------------------------------------------------------------------
# im2col of a 'same' Conv1d: (B, T, C) -> (B, T, k*C) = [x[t-r], ..., x[t], ..., x[t+r]]  (zero rows outside the sequence)
# ---------------------------------------------------------------------------------------------------------------------

class _ShiftedCat(torch.autograd.Function):
    """One launch of the prenet's im2col kernel on the (B, T, 1, C) image with a (k, 1) kernel -- and its gather-form adjoint -- instead of
    pad + k slices + cat (three torch launches forward, 2k + 2 backward, per convolution)."""

    @staticmethod
    def forward(ctx, x, k):
        require_gpu(x, "conv1d im2col")
        x = x if x.is_contiguous() else x.contiguous()
        B, T, C = x.shape
        g = (B, T, 1, C, T, 1, k, 1, 1, 1, (k - 1) // 2, 0)
        col = torch.empty(B, T, k * C, dtype=x.dtype, device=x.device)
        lib.call("sc_im2col", dcode(x), _p(x), *g, _p(col), stream())
        ctx.g = g
        return col

    @staticmethod
    def backward(ctx, dcol):
        g = ctx.g
        dcol = dcol if dcol.is_contiguous() else dcol.contiguous()
        dx = torch.empty(g[0], g[1], g[3], dtype=dcol.dtype, device=dcol.device)
        lib.call("sc_col2im", dcode(dcol), _p(dcol), *g, _p(dx), stream())
        return dx, None


def shifted_cat(x: torch.Tensor, k: int) -> torch.Tensor:
    return x if k == 1 else _ShiftedCat.apply(x, int(k))


# ---------------------------------------------------------------------------------------------------------------------
# FastSpeech2 glue (csrc/tts.hip): length regulation and per-token averages
# ---------------------------------------------------------------------------------------------------------------------

class _LengthRegulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dur, out_len):
        require_gpu(x, "length regulation")
        B, L, D = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        dur = dur.detach().to(torch.float32).contiguous()
        out = torch.empty(B, out_len, D, dtype=x.dtype, device=x.device)
        lib.call("sc_length_regulate_fwd", dcode(x), _p(x), _p(dur), _p(out), B, L, out_len, D, stream())
        ctx.save_for_backward(dur)
        ctx.meta = (B, L, D, out_len)
        return out

    @staticmethod
    def backward(ctx, dy):
        (dur,) = ctx.saved_tensors
        B, L, D, T = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty(B, L, D, dtype=dy.dtype, device=dy.device)
        lib.call("sc_length_regulate_bwd", dcode(dy), _p(dy), _p(dur), _p(dx), B, L, T, D, stream())
        return dx, None, None


def length_regulate(x: torch.Tensor, duration: torch.Tensor, out_len: int) -> torch.Tensor:
    """(B, L, D) token rows -> (B, out_len, D) frame rows: row k of utterance b repeated int(duration[b][k]) times, zero rows behind the
    last frame, cut at out_len (speechain/module/decoder/nar_tts.py:439-472)."""
    return _LengthRegulate.apply(x, duration, int(out_len))


def avg_by_duration(frame_scalar: torch.Tensor, duration: torch.Tensor) -> torch.Tensor:
    """(B, T) frame-level scalar -> (B, L) per-token means over each token's frames, 0 for a token without frames
    (FastSpeech2Decoder.average_scalar_by_duration, decoder/nar_tts.py:150-204).  Targets only: not differentiated."""
    require_gpu(frame_scalar, "average_scalar_by_duration")
    f = frame_scalar.detach().to(torch.float32).contiguous()
    d = duration.detach().to(torch.float32).contiguous()
    out = torch.empty(d.shape, dtype=torch.float32, device=f.device)
    lib.call("sc_avg_by_duration", _p(f), _p(d), _p(out), f.size(0), d.size(1), f.size(1), stream())
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Cached-key/value decoding (inference; csrc/decode.hip)
# ---------------------------------------------------------------------------------------------------------------------

def beam_table_update(table_in: Optional[torch.Tensor], table_out: torch.Tensor, beam_idx: Optional[torch.Tensor], pos: int):
    """table_out[n, :pos] = table_in[beam_idx[n], :pos] (identity without beam_idx), table_out[n, pos] = n; (N, Tmax) int32."""
    require_gpu(table_out, "beam_table_update")
    N, Tmax = table_out.shape
    lib.call("sc_beam_table_update", _p(table_in), _p(table_out), _p(beam_idx), N, Tmax, int(pos), stream())


def decode_embed_step(emb: torch.Tensor, tokens: torch.Tensor, emb_scale: float, pe_row: torch.Tensor, alpha: Optional[torch.Tensor], xscale: float,
                      table_in: Optional[torch.Tensor], table_out: torch.Tensor, beam_idx: Optional[torch.Tensor], pos: int) -> torch.Tensor:
    """`beam_table_update` and the decoder input of the step -- embedding lookup (x emb_scale), cast to the compute type, positional encoding of
    position `pos` -- in one launch (sc_decode_embed_step).  emb (V, d) f32, tokens (N,) int64, pe_row (d,) f32; returns x (N, d)."""
    require_gpu(emb, "decode_embed_step")
    N, Tmax = table_out.shape
    V, d = emb.shape
    dt = compute_dtype()
    assert emb.dtype == torch.float32 and emb.is_contiguous() and tokens.dtype == torch.long and tokens.numel() == N and tokens.is_contiguous()
    assert pe_row.dtype == torch.float32 and pe_row.numel() == d and pe_row.is_contiguous() and dt in (torch.float32, torch.bfloat16)
    x = torch.empty(N, d, dtype=dt, device=emb.device)
    lib.call("sc_decode_embed_step", BF16 if dt == torch.bfloat16 else F32, _p(emb), _p(tokens), float(emb_scale), _p(pe_row),
             _p(alpha.detach()) if alpha is not None else None, float(xscale), _p(x), _p(table_in), _p(table_out), _p(beam_idx), N, d, V, Tmax,
             int(pos), stream())
    return x


_DECODE_EMBED = _os.environ.get("SC_DECODE_EMBED", "1") == "1"  # (A/B knob: 0 = lookup, cast, encoding and table as their own launches)
_DECODE_LINEAR = _os.environ.get("SC_DECODE_LINEAR", "1") == "1"  # (A/B knob: 0 = LayerNorm launch + the general GEMM kernel)


_DECODE_STATS_MAXN = int(_os.environ.get("SC_DECODE_STATS_MAXN", "2048"))  # (A/B knob: widest product that takes sc_decode_linear_lnstats)
_DECODE_LN_STATS = _os.environ.get("SC_DECODE_LN_STATS", "1") == "1"  # (A/B knob: 0 = LayerNorm folded in place / as its own launch)


def decode_linear(x: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]], ln: Optional[torch.nn.LayerNorm] = None,
                  activation: Optional[str] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inference-only linear layer for the rows of one decoding step: ``act(LayerNorm(x) W^T + b) + residual`` in ONE launch
    (csrc/decode_gemm.hip: the whole contraction of a 64 x 64 tile requested at once; the LayerNorm folded in when the row fits on chip).
    Falls back to ``layer_norm`` + ``linear`` / ``linear_act`` where the kernel does not apply (f32 parity mode, odd sizes)."""
    x2 = x.reshape(-1, x.size(-1))
    K = x2.size(1)
    n_out = sum(w.size(0) for w in ws)
    # Measured per product at 512 rows (scripts/bench_decode_linear.py, launches back to back): the narrow products (N = 512: out 3.7 vs
    # 6.5 us, LayerNorm + q 6.5 vs 8.7) and the long contraction (ffn-out, K = 2048: 7.3 vs 13.4) win; the wide ones (q|k|v 12.5 vs 9.0,
    # ffn-in 14.3 vs 10.2, the 5000-wide head 39 vs 16) lose -- every column tile re-stages and re-normalises its rows -- and stay on the
    # LayerNorm launch + the general kernel.
    # With the LayerNorm applied through its row statistics in the epilogue (sc_decode_linear_lnstats: the product reads the raw rows) the
    # wide products behind a LayerNorm (q|k|v, ffn-in) lose their LayerNorm launch as well and take this kernel too.
    # (not the 5000-wide token head: at that width the general kernel wins by more than the LayerNorm launch costs)
    stats = ln is not None and K <= 512 and n_out <= _DECODE_STATS_MAXN and _DECODE_LN_STATS
    wide = n_out > 1024 and K <= 512 and not stats
    ok = (_DECODE_LINEAR and not wide and x2.dtype == torch.bfloat16 and x2.is_cuda and K % 32 == 0 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0
          and (ln is None or K <= 512) and (activation is None or activation in ("ReLU", "GELU")) and not torch.is_grad_enabled())
    if not ok:
        if ln is not None:
            x = layer_norm(x, ln.weight, ln.bias, ln.eps)
        if activation is not None:
            assert len(ws) == 1
            y = linear_act(x, ws[0], bs[0], activation)
            return y if residual is None else y + residual
        return linear(x, ws, bs, residual=residual)
    M, N = x2.size(0), n_out
    y = torch.empty(M, N, dtype=x2.dtype, device=x2.device)
    res2 = None
    if residual is not None:
        res2 = residual.reshape(M, N)
        if res2.stride(1) != 1:
            res2 = res2.contiguous()
    act = 0 if activation is None else ACT[activation]
    if stats:
        Wp, colsum, shift = weights.ln_folded(ws, bs, ln)
        lib.call("sc_decode_linear_lnstats", _p(x2), x2.stride(0), M, K, _p(Wp), Wp.stride(0), N, _p(colsum), _p(shift), float(ln.eps), act, _p(res2),
                 res2.stride(0) if res2 is not None else 0, _p(y), N, stream())
        return y.reshape(*x.shape[:-1], N)
    W = weights.fused(ws, x2.dtype)
    b = weights.vector(bs)
    lib.call("sc_decode_linear", _p(x2), x2.stride(0), M, K, _p(W), W.stride(0), N, _p(b), _p(ln.weight.detach()) if ln is not None else None,
             _p(ln.bias.detach()) if ln is not None else None, float(ln.eps) if ln is not None else 0.0, act, _p(res2),
             res2.stride(0) if res2 is not None else 0, _p(y), N, stream())
    return y.reshape(*x.shape[:-1], N)


def attention_decode(qkv: torch.Tensor, cache: torch.Tensor, table: torch.Tensor, pos: int, num_heads: int, scale: float) -> torch.Tensor:
    """One step of causal self-attention over cached keys / values: qkv (N, 3 d) this step's projections, cache (N, Tmax, 2 d) (slot
    (n, pos) is written), table (N, Tmax) int32 ancestry.  Returns the (N, d) context.  Heads of 64 only."""
    require_gpu(qkv, "attention_decode")
    N, d = qkv.size(0), qkv.size(1) // 3
    if d // num_heads != 64:
        raise NotImplementedError(f"speechain_amd: cached decoding is built for heads of 64 (got d_model={d}, num_heads={num_heads})")
    assert qkv.is_contiguous() and cache.is_contiguous() and cache.dtype == qkv.dtype and cache.shape[0] == N and cache.shape[2] == 2 * d
    out = torch.empty(N, d, dtype=qkv.dtype, device=qkv.device)
    lib.call("sc_attention_decode", dcode(qkv), _p(qkv), qkv.stride(0), _p(cache), _p(table), _p(out), out.stride(0), N, num_heads,
             cache.size(1), int(pos), float(scale), stream())
    return out


def ctc_prefix_init(x: torch.Tensor, enc_lens: torch.Tensor, beam: int, blank: int) -> torch.Tensor:
    """Forward variables (T, 2, N) of the empty prefix: only blanks so far (ctc_decoding.py:88-99).  x (B, T, V) f32."""
    B, T, _ = x.shape
    t = torch.arange(T, device=x.device)[None, :]
    xb = torch.where(t < enc_lens[:, None], x[:, :, blank], torch.zeros((), device=x.device))  # past the utterance: blank is certain
    r = torch.full((T, 2, B * beam), -1e20, dtype=torch.float32, device=x.device)
    r[:, 1] = torch.cumsum(xb, dim=1).t().repeat_interleave(beam, dim=1)
    return r


def ctc_prefix_factors(x: torch.Tensor):
    """(E, bmax) for `ctc_prefix_scores(..., factors=...)`: bmax (B, V) = max over the frames of the log-probabilities x (B, T, V),
    E = exp(x - bmax) -- made once per batch, they turn a step's scores into multiply-adds (sc_ctc_prefix_scores_fast)."""
    bmax = x.max(dim=1).values.contiguous()
    return torch.exp(x - bmax[:, None, :]).contiguous(), bmax


def ctc_prefix_prepared(B: int, beam: int, T: int, device):
    """Buffers (pt, pa) that `ctc_prefix_state(..., prepared=)` fills and the next position's `ctc_prefix_scores(..., prepared=)` reads:
    P = exp(phi - a) of the rows in the matrix-core kernel's layout, (B G, 2, T, 16), and a, (B G, 2, 16), for the G = ceil(beam / 16)
    groups of up to 16 rows of every sentence (zeros where a group holds fewer rows)."""
    G = B * ((beam + 15) // 16)
    return torch.zeros(G, 2, T, 16, dtype=torch.float32, device=device), torch.zeros(G, 2, 16, dtype=torch.float32, device=device)


def ctc_prefix_scores(x, enc_lens, eos_frame, r_prev, last, plen: int, beam: int, blank: int, eos: int, out=None, factors=None,
                      prepared=None) -> torch.Tensor:
    """psi (N, V) of every candidate token of every hypothesis row (sc_ctc_prefix_scores); r_prev None = the empty prefix.  `prepared`:
    the (pt, pa) pair that `ctc_prefix_state` filled when it made r_prev (with `factors` only; plen >= 1)."""
    require_gpu(x, "ctc_prefix_scores")
    B, T, V = x.shape
    if r_prev is None:
        r_prev = ctc_prefix_init(x, enc_lens, beam, blank)
    psi = out if out is not None else torch.empty(B * beam, V, dtype=torch.float32, device=x.device)
    if factors is not None and T >= 2:
        E, bmax = factors
        assert E.shape == x.shape and E.is_contiguous() and bmax.shape == (B, V) and bmax.is_contiguous()
        pt, pa = prepared if (prepared is not None and plen >= 1) else (None, None)
        if pt is not None:
            G = B * ((beam + 15) // 16)
            assert pt.shape == (G, 2, T, 16) and pt.is_contiguous() and pa.shape == (G, 2, 16) and pa.is_contiguous()
        lib.call("sc_ctc_prefix_scores_fast", _p(x), _p(E), _p(bmax), _p(enc_lens), _p(eos_frame), _p(r_prev.contiguous()), _p(last), _p(psi), B,
                 beam, T, V, int(plen), int(blank), int(eos), _p(pt), _p(pa), stream())
        return psi
    lib.call("sc_ctc_prefix_scores", _p(x), _p(enc_lens), _p(eos_frame), _p(r_prev.contiguous()), _p(last), _p(psi), B, beam, T, V, int(plen),
             int(blank), int(eos), stream())
    return psi


def ctc_prefix_state(x, enc_lens, r_prev, last, plen: int, beam_idx, tokens, beam: int, blank: int, out=None, psi_prev=None, psi_sel=None,
                     prepared=None) -> torch.Tensor:
    """Forward variables (T, 2, N) of the rows that continue row beam_idx[n] with tokens[n] (sc_ctc_prefix_state).  `prepared`: a
    (pt, pa) pair of `ctc_prefix_prepared` to fill for the next position's `ctc_prefix_scores`."""
    B, T, V = x.shape
    if r_prev is None:
        r_prev = ctc_prefix_init(x, enc_lens, beam, blank)
    r_new = out if out is not None else torch.empty(T, 2, B * beam, dtype=torch.float32, device=x.device)
    # (psi_prev (N, V) / psi_sel (N): the same launch also gathers psi_sel[n] = psi_prev[beam_idx[n], tokens[n]], ctc_decoding.py:63)
    lib.call("sc_ctc_prefix_state", _p(x), _p(enc_lens), _p(r_prev.contiguous()), _p(last), _p(beam_idx.contiguous()), _p(tokens), _p(r_new),
             B, beam, T, V, int(plen), int(blank), _p(psi_prev), _p(psi_sel), _p(prepared[0] if prepared is not None else None),
             _p(prepared[1] if prepared is not None else None), stream())
    return r_new


def topk_rows(x: torch.Tensor, k: int):
    """(values, indices) of the k largest elements of every row of the 2-D f32 tensor x, descending, ties by the lower index
    (sc_topk_rows; k <= 64)."""
    require_gpu(x, "topk_rows")
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R, L = x.shape
    vals = torch.empty(R, k, dtype=torch.float32, device=x.device)
    idx = torch.empty(R, k, dtype=torch.long, device=x.device)
    lib.call("sc_topk_rows", _p(x), x.stride(0), R, L, int(k), _p(vals), _p(idx), stream())
    return vals, idx


def beam_pack(vals: torch.Tensor, idx: torch.Tensor, eos: Optional[torch.Tensor], other: Optional[torch.Tensor], beam: int, pack: torch.Tensor):
    """pack (B, 2 K [+ 2 beam]) float64 <- [vals | idx | eos | other] per sentence (sc_beam_pack)."""
    B, K = vals.shape
    assert pack.dtype == torch.float64 and pack.is_contiguous() and pack.shape == (B, 2 * K + (2 * beam if eos is not None else 0))
    assert idx.dtype == torch.long and (eos is None or (eos.dtype == torch.float32 and other.dtype == torch.float32 and eos.is_contiguous()
                                                        and other.is_contiguous() and eos.numel() == B * beam == other.numel()))
    lib.call("sc_beam_pack", _p(vals), _p(idx), _p(eos), _p(other), B, K, int(beam), _p(pack), stream())


def beam_advance(vals, idx, eos, other, beam: int, V: int, eos_token: int, pack, scores, tokens, prev_tokens, beam_idx):
    """`beam_pack` and the next step's rows in one launch (sc_beam_advance): scores / tokens / beam_idx <- the first `beam` candidates of every
    sentence that are not <sos/eos>, prev_tokens <- tokens."""
    B, K = vals.shape
    assert pack.dtype == torch.float64 and pack.is_contiguous() and pack.shape == (B, 2 * K + (2 * beam if eos is not None else 0))
    assert idx.dtype == torch.long and vals.dtype == torch.float32 and vals.is_contiguous() and idx.is_contiguous()
    assert eos is None or (eos.dtype == torch.float32 and other.dtype == torch.float32 and eos.is_contiguous() and other.is_contiguous()
                           and eos.numel() == B * beam == other.numel())
    for t_, dt in ((scores, torch.float32), (tokens, torch.long), (prev_tokens, torch.long), (beam_idx, torch.long)):
        assert t_.dtype == dt and t_.is_contiguous() and t_.numel() == B * beam
    lib.call("sc_beam_advance", _p(vals), _p(idx), _p(eos), _p(other), B, K, int(beam), int(V), int(eos_token), _p(pack), _p(scores), _p(tokens),
             _p(prev_tokens), _p(beam_idx), stream())


_TOPK_TWO_STAGE = _os.environ.get("SC_TOPK_TWO_STAGE", "1") == "1"  # (A/B knob)


def topk_groups(x: torch.Tensor, parts: int, k: int):
    """`topk_rows(x.view(G, parts * L), k)` for x (G * parts, L) in two stages -- every row's own top k (G * parts workgroups over L columns
    each), then the merge of a group's parts * k candidates (sc_topk_merge) -- with the same result, ties included."""
    R, L = x.shape
    G = R // parts
    if parts == 1 or parts * k > 1024 or L < k or not _TOPK_TWO_STAGE:
        return topk_rows(x.view(G, parts * L), k)
    v1, i1 = topk_rows(x, k)
    vals = torch.empty(G, k, dtype=torch.float32, device=x.device)
    idx = torch.empty(G, k, dtype=torch.long, device=x.device)
    lib.call("sc_topk_merge", _p(v1), _p(i1), G, int(parts), int(k), L, _p(vals), _p(idx), stream())
    return vals, idx


def beam_select_advance(cand: torch.Tensor, beam: int, k: int, eos, other, eos_token: int, pack, scores, tokens, prev_tokens, beam_idx):
    """The selection of a beam-search step from its (B * beam, V) candidate scores in two launches: every row's own top k (sc_topk_rows),
    then merge per sentence + the step's record + the next rows (sc_beam_merge_advance).  Returns the merged (B, k) scores and indices.
    Needs 2 * beam <= k <= 64 and beam * k <= 1024 (the caller falls back to `topk_groups` + `beam_advance` otherwise)."""
    R, V = cand.shape
    B = R // beam
    v1, i1 = topk_rows(cand, k)
    vals = torch.empty(B, k, dtype=torch.float32, device=cand.device)
    idx = torch.empty(B, k, dtype=torch.long, device=cand.device)
    assert pack.dtype == torch.float64 and pack.is_contiguous() and pack.shape == (B, 2 * k + (2 * beam if eos is not None else 0))
    for t_, dt in ((scores, torch.float32), (tokens, torch.long), (prev_tokens, torch.long), (beam_idx, torch.long)):
        assert t_.dtype == dt and t_.is_contiguous() and t_.numel() == B * beam
    lib.call("sc_beam_merge_advance", _p(v1), _p(i1), B, int(beam), int(k), V, _p(eos), _p(other), int(eos_token), _p(vals), _p(idx), _p(pack),
             _p(scores), _p(tokens), _p(prev_tokens), _p(beam_idx), stream())
    return vals, idx


def beam_step_select(logits: torch.Tensor, temperature: float, psi, psi_prev, ctc_weight: float, blank: int, eos_token: int, want_eos_stats: bool,
                     beam: int, k: int, pack, scores, tokens, prev_tokens, beam_idx):
    """The tail of a beam-search step in TWO launches: candidate scores (log-softmax of the head's logits, CTC mixing, running scores) and
    every row's own top k without the (N, V) score matrix (sc_beam_candidates_topk), then merge per sentence + the step's record + the
    next rows (sc_beam_merge_advance).  `scores` is read (the rows' running scores) and then overwritten.  V <= 12288, 2 beam <= k <= 64,
    beam * k <= 1024."""
    require_gpu(logits, "beam_step_select")
    N, V = logits.shape
    B = N // beam
    assert logits.stride(1) == 1 and scores.dtype == torch.float32
    v1 = torch.empty(N, k, dtype=torch.float32, device=logits.device)
    i1 = torch.empty(N, k, dtype=torch.long, device=logits.device)
    es = torch.empty(N, dtype=torch.float32, device=logits.device) if want_eos_stats else None
    om = torch.empty(N, dtype=torch.float32, device=logits.device) if want_eos_stats else None
    lib.call("sc_beam_candidates_topk", dcode(logits), _p(logits), logits.stride(0), _p(psi), _p(psi_prev), _p(scores), _p(es), _p(om), N, V,
             float(temperature), float(ctc_weight), int(blank), int(eos_token), int(k), _p(v1), _p(i1), stream())
    vals = torch.empty(B, k, dtype=torch.float32, device=logits.device)
    idx = torch.empty(B, k, dtype=torch.long, device=logits.device)
    assert pack.dtype == torch.float64 and pack.is_contiguous() and pack.shape == (B, 2 * k + (2 * beam if want_eos_stats else 0))
    lib.call("sc_beam_merge_advance", _p(v1), _p(i1), B, int(beam), int(k), V, _p(es), _p(om), int(eos_token), _p(vals), _p(idx), _p(pack),
             _p(scores), _p(tokens), _p(prev_tokens), _p(beam_idx), stream())
    return vals, idx


def beam_candidates(logits: torch.Tensor, beam_score: torch.Tensor, temperature: float, psi=None, psi_prev=None, ctc_weight: float = 0.0,
                    blank: int = 0, eos: int = 0, want_eos_stats: bool = False):
    """(N, V) f32 candidate scores of a beam-search step from the token head's logits (sc_beam_candidates): length-V log-softmax at
    `temperature`, mixed with the CTC prefix scores `psi - psi_prev` when given, plus the rows' running scores.  With `want_eos_stats`
    also the mixed <eos> score and the best other score of every row."""
    require_gpu(logits, "beam_candidates")
    N, V = logits.shape
    assert logits.stride(1) == 1 and beam_score.dtype == torch.float32
    cand = torch.empty(N, V, dtype=torch.float32, device=logits.device)
    es = torch.empty(N, dtype=torch.float32, device=logits.device) if want_eos_stats else None
    om = torch.empty(N, dtype=torch.float32, device=logits.device) if want_eos_stats else None
    lib.call("sc_beam_candidates", dcode(logits), _p(logits), logits.stride(0), _p(psi), _p(psi_prev), _p(beam_score), _p(cand), _p(es), _p(om),
             N, V, float(temperature), float(ctc_weight), int(blank), int(eos), stream())
    return cand, es, om
