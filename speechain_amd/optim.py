"""Optimizer tail of the training step on the flat gradient buckets: global-norm clipping + Adam in two launches (csrc/optim.hip).

What it replaces: ``torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)`` followed by ``torch.optim.Adam.step()`` in the
reference's ``OptimScheduler.step`` (speechain/optim_sche/abs.py:232-253) -- with stock torch about 40 launches and three extra passes
over the gradients per step.  Same arithmetic: the norm over ALL gradients (finished in float64), ``coef = min(1, max_norm / (norm +
1e-6))``, Adam without weight decay / amsgrad, bias corrections from a step counter that lives on the device (so a captured hipGraph
advances it on every replay), learning rate read from a device cell the scheduler fills.  The clipped gradient is not written back:
``clip_coef`` holds the factor the update used.  As in the reference (abs.py:239-247) a step whose gradient norm is not finite is
SKIPPED: parameters, both moments and the step counter stay as they were, and ``skipped`` reads 1.

The update is a raw kernel, so torch's version counters do not move: ``step()`` therefore marks the cached compute-dtype weight
copies (``functional.weights``) stale, and the next forward pass rebuilds them -- all of them in two launches if the training loop
calls ``functional.weights.refresh()`` once per step (a captured step must: its replays never run ``step()`` on the host).
"""
import ctypes
from typing import Optional, Tuple

import torch

from speechain_amd._lib import lib, stream


class FusedAdam:
    def __init__(self, reducer, lr: torch.Tensor, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = None):
        """`reducer`: the ``GradBucketReducer`` that owns the gradient buckets (its parameters are flattened here).  `lr`: a one-element
        float32 device tensor, read by the kernel at every step."""
        assert lr.dtype == torch.float32 and lr.numel() == 1 and lr.is_cuda, "lr must be a one-element float32 device tensor"
        self.reducer, self.lr, self.betas, self.eps = reducer, lr, betas, eps
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self.grads = [b["flat"] for b in reducer.buckets]
        self.params = reducer.flatten_parameters()
        self.exp_avg = [torch.zeros_like(f) for f in self.params]
        self.exp_avg_sq = [torch.zeros_like(f) for f in self.params]
        dev = self.grads[0].device
        self.state = torch.zeros(3, dtype=torch.float32, device=dev)  # ||g||^2, clip coefficient, step count
        n = len(self.grads)
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._g, self._p, self._m, self._v = ptrs(self.grads), ptrs(self.params), ptrs(self.exp_avg), ptrs(self.exp_avg_sq)
        self._n = (ctypes.c_longlong * n)(*[t.numel() for t in self.grads])
        self._ws = torch.empty(lib.value("sc_grad_sumsq_workspace", n, self._n), dtype=torch.uint8, device=dev)

    @property
    def grad_norm(self) -> torch.Tensor:
        return self.state[0].sqrt()

    @property
    def clip_coef(self) -> torch.Tensor:
        """The factor the last update multiplied the gradients by (0 for a skipped step)."""
        return self.state[1].clamp(min=0.0)

    @property
    def skipped(self) -> torch.Tensor:
        """1.0 if the last step was skipped because its gradient norm was not finite (device scalar; reading it synchronises)."""
        return (self.state[1] < 0).float()

    def step(self):
        n = len(self.grads)
        lib.call("sc_grad_sumsq", n, self._g, self._n, self.max_grad_norm, self.state.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream())
        lib.call("sc_adam_step", n, self._p, self._g, self._m, self._v, self._n, self.lr.data_ptr(), float(self.betas[0]), float(self.betas[1]),
                 float(self.eps), self.state.data_ptr(), stream())
        from speechain_amd.functional import weights
        weights.mark_stale()  # the kernel wrote the parameters behind torch's version counters

    def state_of(self, param: torch.Tensor):
        """(exp_avg, exp_avg_sq) views of one parameter, shaped like it (for checkpoints in torch.optim.Adam's layout)."""
        for b, m, v in zip(self.reducer.buckets, self.exp_avg, self.exp_avg_sq):
            if id(param) in b["index"]:
                off = b["offsets"][b["index"][id(param)]]
                return m[off: off + param.numel()].view_as(param), v[off: off + param.numel()].view_as(param)
        raise KeyError("parameter is not managed by this optimizer")
