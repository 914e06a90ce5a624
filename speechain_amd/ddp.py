"""Single-node data parallelism for the train step: bucketed gradient all-reduce over RCCL/xGMI, overlapped with backward.

What it replaces: ``torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu])`` at speechain/runner.py:1855-1862
(NCCL backend, :220-224) -- the only data-path collective of the reference.  Differences, on purpose:
  * no per-forward buffer broadcast (the reference re-broadcasts ~20 MB of constant posenc tables every step);
  * gradients live in a few flat buckets laid out in reverse registration order (decoder head first ... encoder layer 0
    last), each ``param.grad`` being a view into its bucket, so a bucket is reduced in place with ONE collective as soon
    as its last gradient has been accumulated;
  * collectives are issued on a side stream behind an event recorded on the backward stream; the optimizer waits on them.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 312 MB fp32 gradient set costs ~3.6 ms as one ring but each
~25-50 MB bucket is hidden behind the remaining backward compute.
Works with any process group (``nccl`` = RCCL on GPUs; ``gloo`` on CPU for the tests).
"""
import contextlib
import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def attention_fusion_groups(model: torch.nn.Module):
    """Parameter groups of a model whose gradients one fused kernel produces: q/k/v weights (and biases) of every
    speechain_amd attention block, in the order the fused projection stacks them."""
    from speechain_amd.module.transformer.attention import MultiHeadedAttention

    groups = []
    for m in model.modules():
        if isinstance(m, MultiHeadedAttention):
            groups.append([m.q_layer.weight, m.k_layer.weight, m.v_layer.weight])
            groups.append([m.q_layer.bias, m.k_layer.bias, m.v_layer.bias])
    return groups


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0, process_group=None,
                 adjacent: Iterable[Iterable[torch.nn.Parameter]] = (), sinks: bool = True):
        """`adjacent`: groups of parameters to lay out back to back, in the given order (e.g. the q/k/v weights of an attention
        block, so that their fused gradient GEMM can write one contiguous slot).  `sinks`: register the slots with
        ``speechain_amd.functional.GradSinks`` so the backward kernels write gradients in place."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # `multi`: run the collectives.  True for more than one rank -- or when a process group exists and the environment
        # asks for it (SC_DDP_FORCE_COLLECTIVES=1): the multi-rank control flow on a one-rank group, which is how the N > 1
        # path is rehearsed on a single GPU (bench.py BENCH_FORCE_DIST=1).
        self.multi = self.world > 1 or (dist.is_initialized() and os.environ.get("SC_DDP_FORCE_COLLECTIVES", "0") == "1")
        params = [p for p in params if p.requires_grad]
        order = list(reversed(params))  # backward produces gradients roughly in reverse registration order
        group_of = {}
        for g in adjacent:
            g = [p for p in g if p.requires_grad]
            for p in g:
                group_of[id(p)] = g
        self.params, seen = [], set()
        for p in order:
            if id(p) in seen:
                continue
            for q in group_of.get(id(p), [p]):
                if id(q) not in seen:
                    seen.add(id(q))
                    self.params.append(q)
        self.use_sinks = sinks
        assert self.params, "no trainable parameters"
        self.device = self.params[0].device
        self.cuda = self.device.type == "cuda"
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[dict] = []
        # Slots start on 16-byte boundaries (the sink-writing kernels use 16-byte stores): a parameter whose element count is not a
        # multiple of 4 -- a scalar such as the positional encoding's alpha, an odd vocabulary -- is followed by zero padding.  The members
        # of an `adjacent` group stay back to back (one fused kernel writes them as ONE matrix, so only the group's start matters).
        self._glued = {id(q) for g in group_of.values() for q in g[1:]}
        cur, cur_n = [], 0
        for p in self.params:
            assert p.dtype == torch.float32, "master parameters are float32"
            if cur and cur_n + p.numel() > cap and id(p) not in self._glued:
                self._make_bucket(cur)
                cur, cur_n = [], 0
            if id(p) not in self._glued:
                cur_n = (cur_n + 3) // 4 * 4
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._make_bucket(cur)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._pending = []
        # gradient accumulation (no_sync): micro-steps whose gradients stay on this rank.  With gradient sinks the kernels OVERWRITE their
        # slots, so a held micro-step's buckets are added to accumulators of the same layout, which are folded back into the buckets right
        # before they are reduced (or, on one rank, in wait()).  Without sinks autograd accumulates into the slot views by itself.
        self._hold = False
        self._acc, self._acc_live = None, False
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_bucket(self, plist):
        offsets, off = [], 0
        for p in plist:
            if id(p) not in self._glued:
                off = (off + 3) // 4 * 4
            offsets.append(off)
            off += p.numel()
        flat = torch.zeros(off, dtype=torch.float32, device=self.device)  # (padding stays zero: no kernel writes it)
        views = []
        for p, o in zip(plist, offsets):
            v = flat[o: o + p.numel()].view_as(p)
            p.grad = v
            views.append(v)
            if self.use_sinks:
                from speechain_amd.functional import GradSinks

                GradSinks.register(p, flat, o)
        self.buckets.append(dict(params=plist, flat=flat, ready=0, views=views, offsets=offsets, id=len(self.buckets),
                                 index={id(p): i for i, p in enumerate(plist)}))

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            if self.use_sinks and self.multi:  # the slot must hold the gradient before the bucket is reduced
                v = b["views"][b["index"][id(param)]]
                if param.grad is not None and param.grad.data_ptr() != v.data_ptr():
                    v.copy_(param.grad)
                    param.grad = v
            b["ready"] += 1
            if b["ready"] == len(b["params"]):
                self._launch(b)
        return hook

    def _fold(self, b):
        """Adds what earlier micro-steps of this update left in the accumulator to a bucket whose slots now hold the last micro-step's
        gradients (once per update)."""
        if self._acc_live and not b.get("folded", False):
            b["flat"].add_(self._acc[b["id"]])
            b["folded"] = True

    def _launch(self, b):
        if self._hold:  # a micro-step inside no_sync(): nothing leaves this rank
            return
        if not self.multi:  # (one rank: accumulators are folded in wait(), after every deferred gradient launch of backward)
            return
        if self.cuda and torch.cuda.is_current_stream_capturing():
            # Backward is being captured into a hipGraph: a collective issued from a hook would fork the side stream off the
            # capturing stream without a join ("capturing stream has unjoined work") and hand the RCCL watchdog a captured
            # event.  The caller reduces every bucket eagerly after the replay instead (reduce_all_now).
            return
        if self.cuda:
            from speechain_amd.functional import _Side

            # Gradients of a forward branch (harness: the CTC head) are written on a stream of their own, and the hook that completes a
            # bucket may fire on either: the stream the bucket leaves from waits for the step's main stream and every side stream.
            cur = torch.cuda.current_stream(self.device)
            others = [s for (dev, _), s in _Side.streams.items() if dev == self.device.index]
            if getattr(self, "step_stream", None) is not None:
                others.append(self.step_stream)
            for s in others:
                if s != cur:
                    cur.wait_stream(s)
            if self.use_sinks:  # LayerNorm parameter gradients queued for the end of backward belong to the slots too
                from speechain_amd.functional import _LnFinish, _WgradQueue

                _LnFinish.flush()
                _WgradQueue.flush()  # (weight gradients deferred to one grouped launch: they must be in the slots now)
            self._fold(b)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                work = dist.all_reduce(b["flat"], op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self._pending.append(work)
        else:
            self._fold(b)
            work = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append((work, b["flat"]))

    def reduce_all_now(self):
        """Launches every bucket's all-reduce (used when backward ran inside a captured hipGraph, where the per-parameter
        hooks do not fire) and waits for them."""
        if self.use_sinks:
            self._settle()
        for b in self.buckets:
            b["ready"] = -1
            self._launch(b)
        self.wait()

    def flatten_parameters(self):
        """Moves every parameter's storage into a flat float32 buffer per bucket (same layout as the gradients) and returns the list of
        those buffers: what a fused optimizer (``speechain_amd.optim.FusedAdam``) steps over.  Values are preserved; ``param.data``
        becomes a view, so modules, ``state_dict`` and checkpoints see no difference."""
        if getattr(self, "flat_params", None) is None:
            self.flat_params = []
            with torch.no_grad():
                for b in self.buckets:
                    flat = torch.zeros_like(b["flat"])  # (padding elements: zero parameter, zero gradient, zero moments for ever)
                    for p, off in zip(b["params"], b["offsets"]):
                        v = flat[off: off + p.numel()].view_as(p)
                        v.copy_(p.data)
                        p.data = v
                    self.flat_params.append(flat)
            if self.use_sinks:
                from speechain_amd.functional import weights

                weights.invalidate()  # cached compute-dtype copies are keyed by the old storage
        return self.flat_params

    def completed_buckets(self, taken: set) -> List[int]:
        """Indices of the buckets whose every gradient has been accumulated since ``zero_grad()`` and that are not in `taken` (which is
        updated).  For a trainer that CAPTURES backward in several hipGraphs: called after each piece is captured (the hooks run at capture
        time and count, but launch nothing), it tells which buckets that piece completes -- and hence which all-reduces may be issued
        right after that piece's replay, overlapping the next one."""
        out = [i for i, b in enumerate(self.buckets) if i not in taken and b["ready"] == len(b["params"])]
        taken.update(out)
        return out

    def launch_bucket(self, index: int):
        """Issues the all-reduce of one bucket now (behind an event on the current stream); ``wait()`` will not launch it again."""
        b = self.buckets[index]
        b["ready"] = -1
        self._launch(b)

    def broadcast_parameters(self, src: int = 0):
        """One-off parameter (and nothing else) broadcast at start-up: one collective per flat parameter buffer (nine for the 78 M-parameter
        ASR model) instead of one per tensor; parameters are moved into the flat buffers for it (see ``flatten_parameters``)."""
        if not self.multi:
            return
        for flat in self.flatten_parameters():
            dist.broadcast(flat, src=src, group=self.pg)

    def settle(self):
        """Public form of the step below for callers that capture backward into a hipGraph: called INSIDE the captured region,
        right after ``backward()``, the copies become part of the graph (a replay writes torch-produced gradients into static
        tensors of the capture, not into the slots) and every ``param.grad`` is a slot view by the time the optimizer is captured."""
        if self.use_sinks:
            self._settle()

    def _settle(self):
        """Makes every ``param.grad`` the view of its bucket slot again: gradients produced outside the sink-aware kernels
        (torch ops, second contributions, cloned tensors) are copied in; parameters without a gradient get zeros."""
        for b in self.buckets:
            for p, v in zip(b["params"], b["views"]):
                g = p.grad
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr():
                    v.copy_(g)
                p.grad = v

    def wait(self):
        """Blocks the optimizer's stream until every bucket is reduced; buckets whose hooks never fired (unused
        parameters) are reduced here so all ranks issue the same collectives."""
        if self.use_sinks:
            self._settle()
        for b in self.buckets:
            if b["ready"] != len(b["params"]) and b["ready"] != -1:
                self._launch(b)
        if not self.multi:
            for b in self.buckets:
                self._fold(b)
        for w in self._pending:
            if self.cuda:
                w.wait()
            else:
                work, flat = w
                work.wait()
                flat.div_(self.world)
        self._pending.clear()
        for b in self.buckets:
            b["ready"] = 0

    @contextlib.contextmanager
    def no_sync(self):
        """A micro-step of gradient accumulation: ``with reducer.no_sync(): loss.backward()`` keeps this backward pass's gradients on the
        rank, exactly like ``DistributedDataParallel.no_sync`` in the reference's ``OptimScheduler.step`` (speechain/optim_sche/abs.py:180-188:
        every step but the last of ``accum_grad`` runs under it, the loss being divided by ``accum_grad`` by the caller, :205-207).  The last
        micro-step runs outside the context; its buckets leave as the SUM of all micro-steps since ``zero_grad()``.  Eager steps only (a
        captured step replays one fixed micro-step)."""
        assert not (self.cuda and torch.cuda.is_current_stream_capturing()), "no_sync(): gradient accumulation is not captured into hipGraphs"
        self._hold = True
        try:
            yield
        finally:
            self._hold = False
            self._stash()

    def _stash(self):
        if self.use_sinks:
            from speechain_amd.functional import GradSinks, _LnFinish, _WgradQueue

            if self.cuda:
                _LnFinish.flush()
                _WgradQueue.flush()
            self._settle()
            if self._acc is None:
                self._acc = [torch.zeros_like(b["flat"]) for b in self.buckets]
            with torch.no_grad():
                for a, b in zip(self._acc, self.buckets):
                    if self._acc_live:
                        a.add_(b["flat"])
                    else:
                        a.copy_(b["flat"])
            self._acc_live = True
            GradSinks.begin_step()  # the next micro-step's kernels write the slots afresh
            for b in self.buckets:
                for p in b["params"]:
                    p.grad = None
        for b in self.buckets:  # (without sinks autograd keeps adding into the slot views: nothing to move)
            b["ready"] = 0

    def zero_grad(self):
        """Start of a step.  With gradient sinks the kernels overwrite their slots, so nothing is zeroed: ``param.grad`` is
        dropped (autograd then adopts the slot view a kernel returns instead of adding into it).  Without sinks the flat
        buckets are zeroed and gradients stay views into them."""
        self.step_stream = torch.cuda.current_stream(self.device) if self.cuda else None  # (the stream the step's main path runs on)
        self._acc_live = False
        for b in self.buckets:
            b["folded"] = False
        if self.use_sinks:
            from speechain_amd.functional import GradSinks

            GradSinks.begin_step()
            for b in self.buckets:
                for p in b["params"]:
                    p.grad = None
                b["ready"] = 0
            return
        for b in self.buckets:
            b["flat"].zero_()
            b["ready"] = 0

    def grad_bytes(self) -> int:
        return sum(b["flat"].numel() for b in self.buckets) * 4
